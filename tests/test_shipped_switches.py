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
