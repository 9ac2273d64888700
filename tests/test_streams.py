"""Streams mode = BASELINE.json config 4 ("batched EuRoC V1_01/V1_02/V2_01/MH_02 streams sharded 1 seq/GPU"): whole
sequences assigned to ranks, one extractor context per stream, the vocabulary through one broadcast, sampled frames of
every stream (incl. the pair that straddles a batch boundary) bit-exact against the oracle."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = os.path.join(ROOT, "tests", "streams_check.py")
# the four sequences' lengths 2912 / 1710 / 2280 / 3040, shortened 100x
LENGTHS = "29,17,23,30"


def test_batch_plan_covers_every_consecutive_pair_once():
    from orbhip.streams import batch_plan, default_samples, pair_location
    for n in (1, 2, 7, 8, 9, 29, 64, 65):
        for B in (2, 3, 8, 16):
            plan = batch_plan(n, B)
            seen = {}
            for k, (s, nb) in enumerate(plan):
                assert 1 <= nb <= B and s + nb <= n
                for b in range(1, nb):
                    assert (s + b) not in seen
                    seen[s + b] = (k, b)
            assert sorted(seen) == list(range(1, n))          # every pair (t - 1, t), t = 1..n-1, exactly once
            assert all(pair_location(t, B) == seen[t] for t in seen)
            assert plan[0][0] == 0 and plan[-1][0] + plan[-1][1] == n
            assert all(0 <= t < n for t in default_samples(n, B))
    assert batch_plan(0, 8) == []
    with pytest.raises(ValueError):
        batch_plan(5, 1)


def test_assignment_of_the_four_euroc_streams():
    from orbhip import distributed as D
    from orbhip.streams import EUROC_STREAMS
    lengths = [n for _, n in EUROC_STREAMS]
    assert D.assign_streams(lengths, 4) == [[3], [0], [2], [1]]          # one sequence per GPU, longest first
    two = D.assign_streams(lengths, 2)
    assert sorted(sum(two, [])) == [0, 1, 2, 3] and abs(sum(lengths[i] for i in two[0]) - sum(lengths[i] for i in two[1])) < 1400
    eight = D.assign_streams(lengths * 2, 8)
    assert all(len(r) == 1 for r in eight)


def _json_line(out):
    return json.loads([l for l in out.splitlines() if l.strip().startswith("{")][-1])


@pytest.mark.gpu
def test_config4_streams_on_one_rank_with_the_rccl_broadcast():
    """Four streams with the V101 / V102 / V201 / MH02 length ratios on per-stream contexts of one GPU; the vocabulary goes
    through orbhip_comm_init + orbhip_bcast_blob_device (one-rank RCCL communicator) into a device buffer."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    out = subprocess.run([sys.executable, CHECK, "--lengths", LENGTHS, "--batch", "8", "--unique", "12", "--voc-l", "5"],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _json_line(out.stdout)
    assert d["mode"] == "streams" and d["n_ranks"] == 1 and d["rccl_ranks"] == 1 and d["backend"] == "nccl"
    assert d["assignment"] == [[3, 0, 2, 1]] and [s["frames"] for s in d["streams"]] == [29, 17, 23, 30]
    assert d["verified_frames"] == 4 * 5 and d["per_rank"][0]["frames"] == 99 and d["value"] > 100


@pytest.mark.gpu
def test_config4_streams_two_ranks_sharing_device_0():
    """Multi-rank orchestration on the GPU path with the one GPU a box has: two processes, both on device 0 (RCCL refuses
    duplicate devices, so the blob travels over gloo), streams assigned longest-first, every rank verifies its own."""
    from orbhip import distributed as D
    rc, out = D.launch_ranks([sys.executable, CHECK, "--lengths", LENGTHS, "--batch", "8", "--unique", "12", "--voc-l", "5",
                              "--backend", "gloo"], 2, timeout=900, local_ranks=[0, 0])
    assert rc == 0, out[-3000:]
    d = _json_line(out)
    assert d["n_ranks"] == 2 and d["backend"] == "gloo" and d["rccl_ranks"] is None
    assert d["assignment"] == [[3, 1], [0, 2]]
    assert [r["frames"] for r in d["per_rank"]] == [47, 52] and [r["verified_frames"] for r in d["per_rank"]] == [10, 10]
    assert d["verified_frames"] == 20
