"""bench.py's whole-batch check on the host (torch CPU tensors): every tiled copy must equal its original in what belongs to the
frame (count, keypoints, descriptors -- original = frame b % U) and in what belongs to the pair (b - 1, b) (match count, match12,
match21 -- original = row ((b - 1) % U) + 1); entries beyond a frame's count are not compared; a difference ends the run with
exit code 3.  (The check found a once-in-thousands fault of k_bow_lane in round 5 that 33 frames against the oracle and a soak
of 4 000 configurations had passed: it must itself be tested.)"""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _batch(U=4, R=5, cap=12, seed=3):
    import torch
    rng = np.random.default_rng(seed)
    B = U * R
    cnt_u = rng.integers(5, cap + 1, U)
    kps_u = rng.integers(0, 1 << 30, (U, cap, 7))
    desc_u = rng.integers(0, 256, (U, cap, 32))
    # pair rows 1 .. U (row r: frame r % U against frame (r - 1) % U)
    pair = {r: (int(rng.integers(0, 9)), rng.integers(-1, cap, cap), rng.integers(-1, cap, cap)) for r in range(1, U + 1)}
    cnt = np.zeros(B, np.int32)
    kps = rng.integers(0, 1 << 30, (B, cap, 7)).astype(np.int32)          # garbage beyond the counts, different in every row
    desc = rng.integers(0, 256, (B, cap, 32)).astype(np.uint8)
    nm = rng.integers(0, 99, B).astype(np.int32)
    m12 = rng.integers(-1, cap, (B, cap)).astype(np.int32)
    m21 = rng.integers(-1, cap, (B, cap)).astype(np.int32)
    for b in range(B):
        u = b % U
        n = int(cnt_u[u])
        cnt[b] = n
        kps[b, :n] = kps_u[u, :n]
        desc[b, :n] = desc_u[u, :n]
        if b >= 1:
            r = (b - 1) % U + 1
            nprev = int(cnt_u[(b - 1) % U])
            nm[b] = pair[r][0]
            m12[b, :nprev] = pair[r][1][:nprev]
            m21[b, :n] = pair[r][2][:n]
    t = lambda a: torch.from_numpy(a.copy())          # noqa: E731
    return dict(cnt=t(cnt), kps=t(kps), desc=t(desc), nm=t(nm), m12=t(m12), m21=t(m21)), U, B, cap


def test_tiled_copies_equal_their_originals_and_garbage_beyond_the_counts_is_ignored():
    m = _bench()
    bufs, U, B, cap = _batch()
    assert m.verify_tiled_copies(bufs, U, B, cap, "bow") == B - (U + 1)
    assert m.verify_tiled_copies(bufs, U, U + 1, cap, "bow") == 0          # nothing but originals


@pytest.mark.parametrize("field", ["cnt", "kps", "desc", "nm", "m12", "m21"])
def test_a_single_differing_value_in_a_copy_is_found(field, capsys):
    m = _bench()
    bufs, U, B, cap = _batch()
    b = 2 * U + 3                                                          # a copy (its originals: frame 3, pair row 3)
    if field == "cnt":
        bufs["cnt"][b] += 1
    elif field == "nm":
        bufs["nm"][b] += 1
    elif field == "m12":
        bufs["m12"][b, 0] += 1                                             # (inside frame b - 1's count: >= 5 features)
    else:
        bufs[field][b].view(-1)[0] += 1                                    # first live entry of the row
    with pytest.raises(SystemExit) as e:
        m.verify_tiled_copies(bufs, U, B, cap, "bow")
    assert e.value.code == 3
    assert "frame %d of the timed batch differs from its original" % b in capsys.readouterr().err


def test_a_difference_beyond_a_frames_count_is_not_a_difference():
    m = _bench()
    bufs, U, B, cap = _batch()
    b = 3 * U + 1
    n = int(bufs["cnt"][b])
    if n < cap:
        bufs["desc"][b, n] ^= 0xFF
        bufs["m21"][b, n] += 7
    assert m.verify_tiled_copies(bufs, U, B, cap, "bow") == B - (U + 1)


def test_stream_frames_by_forked_workers_and_oracle_by_worker_processes(oracle):
    """Round 6: the timed batch is a stream of distinct frames drawn by forked workers, and every frame is verified against oracle
    outputs computed by worker PROCESSES on slices of the batch.  Both must equal what one process computes: the frames of
    orbhip.synth.make_frames, and the oracle's keypoints / descriptors / SearchByBoW of every pair -- also across the slice bounds
    (a worker recomputes the frame in front of its slice for the pair)."""
    from orbhip import distributed as D, synth
    m = _bench()
    frames = m.make_stream_frames(77, 16, 3)
    assert frames.shape == (16, m.H, m.W) and np.array_equal(frames, synth.make_frames(77, m.W, m.H, 16))
    assert np.array_equal(m.make_stream_frames(77, 9, 1), frames[:9])               # (one worker: in-process)
    blob = D.make_synthetic_vocabulary(5, k=10, L=3)
    n = 9
    par = m.oracle_outputs_parallel(frames[:n], "bow", blob, 3)                      # 3 slices: 0-2, 3-5, 6-8
    one = m.cpu_baseline(frames[:n], n, "bow", blob, keep=n)[1]
    assert len(par) == len(one) == n
    for b in range(n):
        assert par[b]["k"].tobytes() == one[b]["k"].tobytes() and np.array_equal(par[b]["d"], one[b]["d"]), b
        assert ("bow" in par[b]) == ("bow" in one[b]) == (b > 0)
        if b:
            assert par[b]["bow"][0] == one[b]["bow"][0] > 0
            assert np.array_equal(par[b]["bow"][1], one[b]["bow"][1]) and np.array_equal(par[b]["bow"][2], one[b]["bow"][2])
