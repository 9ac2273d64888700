"""The shipped liborbhip.so reads four environment switches and nothing else (csrc/orbhip_internal.h, ORB_SWITCH / ORB_TUNE):
a stray ORBHIP_* variable in a SLAM process must not be able to change what the library computes, and no timing-ablation stop
("phases") may exist in it.  The ablation build, which tests/ and tools/ use to reach the fallback paths, has them all."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vi-orb-slam-icra2018_amd", "csrc")
SHIPPED = {"ORBHIP_KNN2_MFMA", "ORBHIP_FAST_FIX", "ORBHIP_NO_GRAPH", "ORBHIP_NO_CHAIN"}


def _env_names(lib):
    out = subprocess.run(["strings", os.path.join(CSRC, lib)], capture_output=True, text=True, check=True).stdout
    return [l for l in out.splitlines() if "ORBHIP_" in l]


def test_shipped_library_has_only_the_four_switches():
    lines = _env_names("liborbhip.so")
    assert len(lines) <= 10, lines
    names = {l for l in lines if re.fullmatch(r"ORBHIP_[A-Z0-9_]+", l)}
    assert names == SHIPPED, names
    for banned in ("PHASES", "PADLDS", "LDS_PAD", "LISTCAP"):
        assert not any(banned in l for l in lines), (banned, lines)


def test_ablation_library_has_the_knobs():
    names = {l for l in _env_names("liborbhip_ablation.so") if re.fullmatch(r"ORBHIP_[A-Z0-9_]+", l)}
    for k in ("ORBHIP_FAST_PHASES", "ORBHIP_FAST_LISTCAP", "ORBHIP_DESCRIBE_PHASES", "ORBHIP_PROJ_K", "ORBHIP_FAST_XCD"):
        assert k in names, (k, names)


def test_binding_picks_the_ablation_build_only_when_asked():
    code = "import sys; sys.path.insert(0, %r); from orbhip import capi; print(capi.LIB_PATH)" % os.path.join(ROOT, "vi-orb-slam-icra2018_amd")
    env = {k: v for k, v in os.environ.items() if not k.startswith("ORBHIP_")}
    plain = subprocess.run(["python3", "-c", code], env=env, capture_output=True, text=True, check=True).stdout.strip()
    assert plain.endswith("liborbhip.so")
    forced = subprocess.run(["python3", "-c", code], env=dict(env, ORBHIP_FAST_LISTCAP="8"), capture_output=True, text=True, check=True).stdout.strip()
    assert forced.endswith("liborbhip_ablation.so")
    # the four shipped switches do not need it
    kept = subprocess.run(["python3", "-c", code], env=dict(env, ORBHIP_FAST_FIX="0"), capture_output=True, text=True, check=True).stdout.strip()
    assert kept.endswith("liborbhip.so")


def test_set_fingerprint_is_a_pure_function_of_what_it_hashes():
    """orbhip_set_fingerprint / _rows (include/orbhip.h): host arithmetic, no device -- the identity check of the resident sets."""
    import ctypes as C
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "vi-orb-slam-icra2018_amd"))
    from orbhip import capi
    L = C.CDLL(capi.LIB_PATH)
    for f in (L.orbhip_set_fingerprint, L.orbhip_set_fingerprint_rows):
        f.restype = C.c_uint64
    L.orbhip_set_fingerprint.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.orbhip_set_fingerprint_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    rng = np.random.default_rng(5)
    n = 200
    kps = np.zeros(n, capi.KP_DTYPE)
    kps["x"], kps["y"] = rng.random(n) * 640, rng.random(n) * 480
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    fp = L.orbhip_set_fingerprint(p(kps), p(desc), n)
    assert fp != 0 and fp == L.orbhip_set_fingerprint(p(kps), p(desc), n)
    assert fp == L.orbhip_set_fingerprint_rows(p(kps), p(desc[0]), p(desc[n - 1]), n)          # rows that are not contiguous
    assert fp != L.orbhip_set_fingerprint(p(kps), p(desc), n - 1)                                # another count
    d2 = desc.copy(); d2[n - 1, 7] ^= 1
    assert fp != L.orbhip_set_fingerprint(p(kps), p(d2), n)                                      # another last descriptor
    d3 = desc.copy(); d3[0, 0] ^= 128
    assert fp != L.orbhip_set_fingerprint(p(kps), p(d3), n)                                      # another first descriptor
    k2 = kps.copy(); k2["x"][0] += 1
    assert fp != L.orbhip_set_fingerprint(p(k2), p(desc), n)                                     # another first keypoint
    d4 = desc.copy(); d4[50] ^= 255
    assert fp == L.orbhip_set_fingerprint(p(kps), p(d4), n)                                      # (the middle is not hashed: documented)
    assert L.orbhip_set_fingerprint(None, None, 0) != 0
