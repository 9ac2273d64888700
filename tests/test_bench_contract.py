"""bench.py keeps the driver's contract: ONE JSON line, last on stdout, with the required keys, `roofline` and
`cpu_baseline` (small batch / few steps so that it runs in seconds)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_roofline_and_cpu_baseline():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "64",
                          "--cpu-frames", "40", "--pipelined", "0"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    d = json.loads(lines[-1])                                   # the JSON line is the LAST thing on stdout
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "u8" and d["data"] == "synthetic"
    assert d["unit"] == "frames/s" and d["value"] > 1000 and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["achieved"] > 0
    assert r["traffic"] is None or r["traffic"] > 0              # committed PMC figure only for the default batch
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] == 1 and c["value"] > 1 and c["unit"] == "frames/s" and c["sample"]
    assert d["cpu_baseline_all_cores"]["cores"] >= 1
    # r06: the timed batch holds DISTINCT frames (consecutive frames of one stream), every one of them verified against the oracle;
    # the figures that bracket the headline travel inside `config` (what the driver records): host-fed frames/s -- the reference's
    # own interface -- and the tiled batch of rounds 1-5
    assert d["verified_frames"] == 64 and d["verified_vs_oracle"] == 64 and d["verified_copies_vs_original"] == 0
    cfg = d["config"]
    assert cfg["unique_frames"] == 64 and cfg["frames_per_step_per_gpu"] == 64 and cfg["verified_frames"] == 64
    assert cfg["host_fed_frames_per_s"] == d["host_fed"]["value"] > 1000
    assert cfg["tiled_32_frames_per_s"] == d["tiled_check"]["value"] > 1000
    assert 0.5 < d["tiled_check"]["ratio_to_headline"] < 2.0 and d["tiled_check"]["copies_equal_their_originals"] == 64 - 33
    assert 0 < d["step_hbm_frac"] < 1 and d["step_algorithmic_bytes"] > 64 * 4e6
    # r04: one roofline entry per streaming kernel of the step, measured in this run; the matrix-pipe roofline of the 1M query;
    # the content classes, each verified; the committed counters flagged when they no longer describe the kernel
    names = [e["kernel"] for e in d["rooflines"]]
    assert any(n.startswith("k_resize") for n in names) and "k_fast" in names and any(n.startswith("k_describe_blur") for n in names)
    assert d["stage_ms"]["blur"] == 0 and d["stage_ms"]["describe"] > 0          # r06: no k_blur launch in a batch
    for e in d["rooflines"]:
        assert e["ms"] > 0 and e["algorithmic_bytes"] > 0 and abs(e["frac"] - e["achieved_GBps"] / 8000.0) < 1e-3
    m = d["roofline_mfma"]
    assert m["bound"] == "mfma" and 0 < m["frac"] < 1 and abs(m["frac"] - m["achieved"] / m["peak"]) < 1e-3
    from orbhip import synth
    for kind in ("textured", "indoor_sparse", "white_noise", "low_contrast") + (("photographs",) if synth.load_photographs() else ()):
        assert d["content"][kind]["verified_frames"] == 256 and d["content"][kind]["value"] > 1000     # 9 vs the oracle + 247 copies
    assert d["roofline"]["traffic_stale"] in (True, False)
    # r05: the same step at two and three times the batch (child processes), every frame verified there too
    for b in ("128", "192"):
        assert d["batch_sweep"][b]["value"] > 1000 and d["batch_sweep"][b]["verified_frames"] == int(b), d["batch_sweep"]


def _bench(args, env=None, timeout=600):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          timeout=timeout, env=e)


def test_gpus_n_starts_n_ranks_itself_cpu_gloo():
    """`bench.py --gpus 2` without a launcher starts two fresh rank processes (RANK / WORLD_SIZE / MASTER_* set,
    127.0.0.1 rendezvous) before touching any GPU; checked here on CPU with the launcher self-test (gloo)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ORBHIP_BENCH_LAUNCH_SELFTEST"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip()][-1])
    assert d == {"selftest": "launcher", "n_gpus": 2, "ranks_seen": 2, "gpus_arg": 2}


def test_gpus_must_match_the_launchers_world_size():
    out = _bench(["--gpus", "4"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "ORBHIP_BENCH_LAUNCH_SELFTEST": "1"},
                 timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr


def test_launch_ranks_reports_a_failing_rank():
    from orbhip import distributed as D
    rc, out = D.launch_ranks([sys.executable, "-c", "import os,sys; print('r'+os.environ['RANK']); sys.exit(int(os.environ['RANK']))"], 2,
                             timeout=60)
    assert rc == 1 and out.strip() == "r0"
    env = D.rank_env(1, 2, 12345, base={})
    assert env["RANK"] == "1" and env["WORLD_SIZE"] == "2" and env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "12345"


@pytest.mark.gpu
def test_bench_distributed_code_path_on_one_gpu():
    """ORBHIP_BENCH_FORCE_DIST=1: process group (nccl = RCCL, world 1), vocabulary broadcast into a device buffer and
    the vocabulary load from that buffer -- the N > 1 code path of bench.py on the one GPU the box has."""
    out = _bench(["--steps", "2", "--warmup", "1", "--batch", "64", "--cpu-frames", "0", "--pipelined", "0", "--configs", "0",
                  "--content", "0", "--streams-config", "1"], env={"ORBHIP_BENCH_FORCE_DIST": "1"})
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip()][-1])
    assert d["n_gpus"] == 1 and d["value"] > 1000 and d["verified_frames"] == 64 and d["bow_matches_per_frame"] > 50
    assert d["rccl_ranks"] == 1 and d["vocabulary_broadcast"]["bytes"] > 40e6
    _check_multi_rank_fields(d, 1)


def _check_multi_rank_fields(d, world):
    """r05: what an N > 1 run reports beyond the device-resident step -- every rank fed from host memory at the same time (the
    mode that can fail to scale), with where each rank sits on the host; and config 4 proper, whole streams per rank."""
    hf = d["host_fed"]
    assert hf["ranks"] == world and len(hf["per_rank"]) == world and hf["value"] > 1000 and hf["h2d_GBps_total"] > 0.3
    assert hf["host"]["nproc"] >= 1 and hf["host"]["numa_nodes"] >= 0 and "numa_node_cpus" in hf["host"]
    for i, r in enumerate(hf["per_rank"]):
        assert r["rank"] == i and r["frames_per_s"] > 500 and r["h2d_GBps"] > 0.1 and r["seconds"] > 0
        assert "gpu_numa_node" in r["numa"] and r["numa"]["bound"] in (True, False) and r["numa"]["cpus"] >= 1
    assert hf["value"] <= sum(r["frames_per_s"] for r in hf["per_rank"]) * 1.02         # all frames / the SLOWEST rank's time
    c4 = d["configs"]["4_euroc_streams_one_per_rank"]
    assert c4["ranks"] == world and c4["frames"] == 2912 + 1710 + 2280 + 3040 and c4["value"] > 1000
    assert len(c4["per_rank"]) == world and sum(r["frames"] for r in c4["per_rank"]) == c4["frames"]
    assert all(r["verified"] >= 4 for r in c4["per_rank"]) and c4["verified"] == sum(r["verified"] for r in c4["per_rank"])


@pytest.mark.gpu
def test_bench_two_ranks_sharing_device_0():
    """The N > 1 path of bench.py with two real rank processes on the one GPU a box has (gloo: RCCL refuses duplicate devices,
    the vocabulary travels through host memory): sharded frames, max-over-ranks time, every rank's whole batch verified, the
    host-fed mode on both ranks at once, config 4's streams split over the ranks."""
    from orbhip import distributed as D
    rc, out = D.launch_ranks([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2",
                              "--warmup", "1", "--batch", "64", "--pipelined", "0", "--streams-config", "1"], 2, timeout=900,
                             local_ranks=[0, 0])
    assert rc == 0, out[-3000:]
    d = json.loads([l for l in out.splitlines() if l.strip().startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["rccl_ranks"] is None and len(d["per_rank_frames_per_s"]) == 2
    assert d["verified_frames_per_rank"] == [64, 64] and d["verified_frames"] == 64
    assert "cpu_baseline" not in d and "content" not in d            # rank 0 at N = 1 only
    _check_multi_rank_fields(d, 2)


@pytest.mark.gpu
def test_bench_brute_match_is_verified_too():
    out = _bench(["--steps", "1", "--warmup", "1", "--batch", "16", "--cpu-frames", "0", "--pipelined", "0", "--match", "both",
                  "--verify", "6"])
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip()][-1])
    assert d["verified_frames"] == 6


def test_launch_ranks_ends_the_job_when_a_nonzero_rank_dies_first():
    """Rank 1 fails at start-up while rank 0 would sit in its rendezvous for a minute: the launcher must report the failure
    at once and kill rank 0, not wait for rank 0 first."""
    import time
    from orbhip import distributed as D
    prog = "import os,sys,time; r=int(os.environ['RANK']); print('up', flush=True); sys.exit(7) if r else time.sleep(60)"
    t0 = time.monotonic()
    rc, out = D.launch_ranks([sys.executable, "-c", prog], 2, timeout=50)
    assert rc == 7 and out.strip() == "up" and time.monotonic() - t0 < 20
    # a rank that never finishes: the timeout is reported as 124
    rc, _ = D.launch_ranks([sys.executable, "-c", "import time; time.sleep(30)"], 2, timeout=1.0)
    assert rc == 124
