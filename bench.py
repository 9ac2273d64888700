#!/usr/bin/env python3
"""bench.py -- ORB extract + match throughput on MI355X (BASELINE.json metric).

A step = one pass of the hot path over one batch of B synthetic frames resident in HBM:
  * ORBextractor on all B frames (8-level pyramid, FAST-9/16 per cell, quadtree, IC angle, 7x7 blur,
    rBRIEF), then
  * the reference's frame-to-frame matching (Tracking::TrackReferenceKeyFrame): vocabulary transform of
    every descriptor (Frame::ComputeBoW, levelsup 4) and ORBmatcher::SearchByBoW of every frame against
    its predecessor (--match bow, default), or brute-force best/second Hamming (--match brute), or both.
The timed region runs the whole batch on ONE extractor context (--contexts 1), so that every kernel owns the
GPU while it runs and the per-stage HIP-event times / the roofline are clean.  A supplementary figure
(`pipelined`) reports the free-running throughput when the same batch is split over several contexts (the
reference itself runs two extractor instances side by side for stereo, src/Frame.cc:422-425): their streams let
the latency-bound stages of one part overlap the VALU-bound stages of another.
N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); frames are sharded, there is no
per-frame collective; the ORB vocabulary blob is broadcast once at start-up over xGMI (not timed).

Prints ONE JSON line on rank 0, last on stdout (driver contract), with `roofline` (FAST kernel) and,
at N=1, `cpu_baseline` (the CPU oracle doing the same operations on this box's host cores, bounded
sample).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "vi-orb-slam-icra2018_amd"))

import numpy as np  # noqa: E402

W, H, NFEAT = 640, 480, 1000          # the size BASELINE.json's metric is quoted on
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VOC_K, VOC_L, LEVELSUP = 10, 6, 4     # stock ORBvoc shape; Frame::ComputeBoW uses levelsup 4 (src/Frame.cc:744)
NNRATIO = 0.7                         # TrackReferenceKeyFrame: ORBmatcher matcher(0.7,true) (src/Tracking.cc:1881)


def fast_algorithmic_bytes(w, h, nlevels, level_size):
    """sum_l (w_l - 32)(h_l - 32) bytes read once per frame."""
    tot = 0
    for l in range(nlevels):
        lw, lh = level_size(w, h, l)
        tot += (lw - 32) * (lh - 32)
    return tot


def cpu_baseline(frames, nsample, match, blob, keep=0):
    """The oracle (CPU restatement, 1 thread) on a bounded sample of the same workload.  The outputs of the first
    `keep` frames (keypoints, descriptors, SearchByBoW / knn2 results against the previous frame) are returned too:
    main() compares the GPU's outputs for the same frames with them (`verified_frames`)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orb_oracle_py as oracle
    ex = oracle.Extractor(NFEAT, 1.2, 8, 20, 7)
    voc = oracle.Vocabulary(blob) if match in ("bow", "both") else None
    ex(frames[0])  # warm up (page in)
    kept = []
    t0 = time.perf_counter()
    prev = None
    for i in range(nsample):
        k, d = ex(frames[i % len(frames)])
        cur = {"k": k, "d": d}
        rec = {"k": k, "d": d} if i < keep else None
        if voc is not None:
            w, wt, nid = voc.transform(d, LEVELSUP)
            cur["fv"] = oracle.feature_vector(nid, wt)
        if prev is not None:
            if voc is not None:
                r = oracle.search_by_bow(prev["d"], np.ones(len(prev["d"]), np.uint8), prev["k"]["angle"], prev["fv"], d,
                                         None, k["angle"], cur["fv"], th=50, th_mode=0, nnratio=NNRATIO, check_ori=True)
                if rec is not None:
                    rec["bow"] = r
            if match in ("brute", "both"):
                r = oracle.knn2(d, prev["d"])
                if rec is not None:
                    rec["knn2"] = r
        if rec is not None:
            kept.append(rec)
        prev = cur
    dt = time.perf_counter() - t0
    what = {"bow": "vocabulary transform + SearchByBoW", "brute": "brute-force knn2",
            "both": "vocabulary transform + SearchByBoW + brute-force knn2"}[match]
    res = {"value": round(nsample / dt, 2), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d frames %dx%d, %d features, extract + %s vs previous frame, oracle/liborb_oracle.so "
                     "(gcc -O3 -march=x86-64-v3), %.1f s" % (nsample, W, H, NFEAT, what, dt)}
    return (res, kept) if keep else res


def verify_against_oracle(kept, bufs, cap, match):
    """GPU outputs of the first len(kept) frames of the timed batch against the oracle's, bit for bit: the 28-byte
    keypoint records, the descriptors, SearchByBoW's match12 / match21 / count and the brute-force triples.
    Returns the number of frames verified; raises SystemExit(3) on the first difference."""
    n = len(kept)
    cnt = bufs["cnt"][:n].cpu().numpy()
    kps = bufs["kps"][:n].cpu().numpy()
    desc = bufs["desc"][:n].cpu().numpy()
    get = {name: bufs[name][:n].cpu().numpy() for name in ("m12", "m21", "nm", "bi", "bd", "sd") if name in bufs}

    def bad(what, b):
        print("bench.py: GPU output differs from the oracle: %s of frame %d" % (what, b), file=sys.stderr)
        raise SystemExit(3)
    for b, rec in enumerate(kept):
        k, d = rec["k"], rec["d"]
        if cnt[b] != len(k):
            bad("keypoint count (%d vs %d)" % (cnt[b], len(k)), b)
        if kps[b, :len(k)].tobytes() != k.tobytes():
            bad("keypoints", b)
        if not np.array_equal(desc[b, :len(k)], d):
            bad("descriptors", b)
        if "bow" in rec:
            nm, m12, m21 = rec["bow"]
            if int(get["nm"][b]) != nm or not np.array_equal(get["m12"][b, :len(m12)], m12) or \
                    not np.array_equal(get["m21"][b, :len(m21)], m21):
                bad("SearchByBoW matches", b)
        if "knn2" in rec:
            bi, bd, sd = rec["knn2"]
            if not (np.array_equal(get["bi"][b, :len(bi)], bi) and np.array_equal(get["bd"][b, :len(bd)], bd)
                    and np.array_equal(get["sd"][b, :len(sd)], sd)):
                bad("brute-force best / second", b)
    return n


def cpu_worker(path, nsample, match):
    """Child process of cpu_baseline_all_cores: never touches the GPU; prints its own frame count and seconds."""
    z = np.load(path, allow_pickle=False)
    blob = z["blob"].tobytes() if z["blob"].size else None
    r = cpu_baseline(z["frames"], nsample, match, blob)
    print(json.dumps({"frames": nsample, "fps": r["value"]}), flush=True)


def cpu_baseline_all_cores(frames, nsample, match, blob):
    """The same oracle loop on every host core at once: one child process per core (started as children, the
    GPU process is never replaced), each on its own sample; value = sum of frames / slowest child's time."""
    import subprocess
    import tempfile
    cores = min(len(os.sched_getaffinity(0)), 32)     # bounded: at most 32 processes
    per = max(nsample // 4, 50)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "cpu_sample.npz")
        np.savez(path, frames=frames, blob=np.frombuffer(blob or b"", np.uint8))
        env = dict(os.environ, OMP_NUM_THREADS="1")
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", path, "--cpu-frames", str(per),
                                   "--match", match], stdout=subprocess.PIPE, env=env) for _ in range(cores)]
        outs = [p.communicate()[0] for p in procs]
        wall = time.perf_counter() - t0
    rates = [json.loads(o.decode().strip().splitlines()[-1])["fps"] for o in outs]
    slowest = per / min(rates)
    return {"value": round(cores * per / slowest, 1), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d processes x %d frames (one oracle loop per host core, same operations), %.1f s wall incl. start-up"
                      % (cores, per, wall)}


def committed_traffic(batch, contexts):
    """HBM-side bytes per k_fast launch from the committed rocprofv3 PMC passes (profiles/traffic.json, written by
    tools/profile_gpu.sh + tools/summarize_prof.py for the default configuration), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            t = json.load(fh)
        if int(t.get("batch", -1)) == batch and int(t.get("contexts", 1)) == contexts:
            return int(t["traffic_bytes_per_launch"]), t.get("source", "profiles/traffic.json")
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def pipelined_throughput(args, d_img, blob, device, cap):
    """Free-running multi-context throughput of the same step (no per-step readout): supplementary figure."""
    import torch
    from orbhip.extractor import ORBextractor
    from orbhip.vocabulary import ORBVocabulary
    NC, B = args.pipelined, args.batch
    Bc = B // NC
    i32 = dict(dtype=torch.int32, device="cuda")
    ctxs = []
    for c in range(NC):
        ex = ORBextractor(NFEAT, 1.2, 8, 20, 7, max_w=W, max_h=H, max_batch=Bc, device=device)
        b = {"img": d_img[c * Bc:(c + 1) * Bc], "kps": torch.empty((Bc, cap, 7), **i32),
             "desc": torch.empty((Bc, cap, 32), dtype=torch.uint8, device="cuda"), "cnt": torch.zeros(Bc, **i32),
             "wt": torch.empty((Bc, cap), dtype=torch.float32, device="cuda"), "nm": torch.zeros(Bc, **i32)}
        for name in ("word", "node", "m12", "m21"):
            b[name] = torch.empty((Bc, cap), **i32)
        if blob is not None:
            ORBVocabulary(ex).loadFromBinaryBlob(blob)
        ctxs.append((ex, b))
    L = ctxs[0][0]._L

    def step():
        for ex, b in ctxs:
            ex.extract_batch_device(b["img"].data_ptr(), Bc, W, H, W, H * W, b["kps"].data_ptr(), b["desc"].data_ptr(), cap,
                                    b["cnt"].data_ptr())
            if blob is not None:
                L.orbhip_vocab_transform_device(ex.handle, b["desc"].data_ptr(), Bc * cap, LEVELSUP, b["word"].data_ptr(),
                                                b["wt"].data_ptr(), b["node"].data_ptr())
                L.orbhip_search_by_bow_seq_device(ex.handle, b["desc"].data_ptr(), b["kps"].data_ptr(), b["cnt"].data_ptr(),
                                                  b["node"].data_ptr(), b["wt"].data_ptr(), None, cap, Bc, 1, 0,
                                                  C.c_float(NNRATIO), 1, b["m12"].data_ptr(), b["m21"].data_ptr(),
                                                  b["nm"].data_ptr())
    for _ in range(args.warmup):
        step()
    for ex, _ in ctxs:
        ex.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    for ex, _ in ctxs:
        ex.sync()
    dt = time.perf_counter() - t0
    for ex, _ in ctxs:
        ex.close()
    return {"value": round(B * args.steps / dt, 1), "unit": "frames/s", "contexts": NC, "frames_per_launch": Bc,
            "note": "same step, batch split over independent contexts / HIP streams, no per-step stage readout"}


def host_fed_throughput(args, uniq, blob, device):
    """The same step fed from HOST memory (the reference's frames always arrive from host memory): pinned frame buffers,
    orbhip_pipe_* ring -- batch n + 1 crosses PCIe while batch n computes and batch n - 1's keypoints / descriptors /
    matches come back.  Supplementary figure, never `value`."""
    from orbhip.extractor import ORBextractor
    from orbhip.vocabulary import ORBVocabulary
    Bp, depth = args.host_batch, 3
    ex = ORBextractor(NFEAT, 1.2, 8, 20, 7, max_w=W, max_h=H, max_batch=Bp, device=device)
    if blob is not None:
        ORBVocabulary(ex).loadFromBinaryBlob(blob)
    ex.pipe_create(depth, Bp, W, H)
    if blob is not None:
        ex.pipe_enable_bow(LEVELSUP, NNRATIO, True)
    reps = (Bp + len(uniq) - 1) // len(uniq)
    src = np.concatenate([uniq] * reps)[:Bp]
    pinned = [ex.host_frames((Bp, H, W)) for _ in range(depth)]
    for b in pinned:
        b[:] = src
    nb = max(2 * depth, (args.steps * args.batch + Bp - 1) // Bp)

    def run(nbatches):
        nkp = 0
        for n in range(nbatches):
            if n >= depth:
                _, _, cnt = ex.pipe_wait(copy=False)
                nkp += int(cnt.sum())
            ex.pipe_submit(pinned[n % depth])
        for _ in range(min(depth, nbatches)):
            _, _, cnt = ex.pipe_wait(copy=False)
            nkp += int(cnt.sum())
        return nkp
    run(depth)                                             # warm up: first-touch of the slots, clocks
    t0 = time.perf_counter()
    nkp = run(nb)
    dt = time.perf_counter() - t0
    for b in pinned:
        ex.host_free(b)
    ex.close()
    fps = nb * Bp / dt
    return {"value": round(fps, 1), "unit": "frames/s", "pinned": True, "frames_per_batch": Bp, "ring_depth": depth,
            "batches": nb, "h2d_GBps": round(fps * W * H / 1e9, 2),
            "d2h_GBps": round(fps * (ex.cap * (28 + 32 + (8 if blob is not None else 0)) + 8) / 1e9, 2),
            "keypoints_per_frame": round(nkp / (nb * Bp), 1),
            "note": "orbhip_pipe_submit / orbhip_pipe_wait: frames in pinned host memory, copy-in, kernels (extract"
                    + (" + vocabulary transform + SearchByBoW" if blob is not None else "") + ") and copy-out of every "
                    "batch overlapped with its neighbours'; PCIe Gen5 x16 is 63 GB/s per direction by specification"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="frames per step per GPU")
    ap.add_argument("--unique", type=int, default=32, help="distinct synthetic frames (tiled to the batch)")
    ap.add_argument("--match", choices=["bow", "brute", "both"], default="bow")
    ap.add_argument("--contexts", type=int, default=1, help="extractor contexts the batch is split over in the timed region")
    ap.add_argument("--pipelined", type=int, default=2, help="also report the free-running throughput with this many "
                    "contexts (0 = skip); supplementary, never `value`")
    ap.add_argument("--cpu-frames", type=int, default=800, help="frames of the CPU baseline sample (0 = skip)")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--host-batch", type=int, default=256, help="frames per batch of the host-fed pipeline figure (0 = skip)")
    ap.add_argument("--verify", type=int, default=8, help="frames of the timed batch whose GPU outputs are compared with "
                    "the oracle outside the timed region (0 = skip); a difference ends the run with exit code 3")
    args = ap.parse_args()
    if args.cpu_worker:
        cpu_worker(args.cpu_worker, args.cpu_frames, args.match)
        return

    # ---- one process per GPU ----
    # Under a launcher (torch.distributed.run sets RANK / WORLD_SIZE) this process is one rank.  Without one,
    # `--gpus N` starts the N ranks itself as fresh child processes -- before this process imports torch or touches
    # the GPU (a process that has initialised HIP must not be replaced or forked into another program on this pool).
    from orbhip import distributed as D
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            rc, out0 = D.launch_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus, timeout=float(os.environ.get("ORBHIP_BENCH_TIMEOUT", "1800")))
            sys.stdout.write(out0)
            sys.stdout.flush()
            raise SystemExit(rc)
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, os.environ["WORLD_SIZE"]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus

    if os.environ.get("ORBHIP_BENCH_LAUNCH_SELFTEST"):
        # CPU test of the launcher / rank plumbing (tests/test_bench_contract.py): gloo, no GPU, no product code
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(t)
        dist.barrier()
        dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"selftest": "launcher", "n_gpus": world, "ranks_seen": int(t.item()), "gpus_arg": args.gpus}), flush=True)
        return

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no GPU (%d visible)" % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("ORBHIP_BENCH_FORCE_DIST"):   # the env var exercises the N>1 code path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from orbhip import synth
    from orbhip.extractor import ORBextractor
    from orbhip.vocabulary import ORBVocabulary

    B = args.batch
    NC = max(1, min(args.contexts, B))
    while B % NC:
        NC -= 1
    Bc = B // NC                                           # frames per context
    # independent streams per rank (weak scaling: per-GPU work fixed)
    uniq = synth.make_frames(1000 + rank, W, H, min(args.unique, B))
    reps = (B + len(uniq) - 1) // len(uniq)
    frames = np.concatenate([uniq] * reps)[:B]
    d_img = torch.from_numpy(np.ascontiguousarray(frames)).cuda()          # (B, H, W), stride W (multiple of 16)

    # ORB vocabulary: reference binary format (TemplatedVocabulary.h:1727-1751), synthetic tree of the
    # stock shape (k=10, L=6, 1.11 M nodes, 45.6 MB).  Rank 0 builds it; N>1: one RCCL broadcast over xGMI.
    use_bow = args.match in ("bow", "both")
    blob = None
    d_blob = None
    if use_bow or dist is not None:
        blob = D.make_synthetic_vocabulary(4242, VOC_K, VOC_L) if rank == 0 else b""
        if dist is not None:
            d_blob = D.broadcast_blob(blob, src=0, device="cuda")
            torch.cuda.synchronize()

    i32 = dict(dtype=torch.int32, device="cuda")
    ctxs = []
    for c in range(NC):
        ex = ORBextractor(NFEAT, 1.2, 8, 20, 7, max_w=W, max_h=H, max_batch=Bc, device=local_rank)
        cap = ex.cap
        bufs = {"img": d_img[c * Bc:(c + 1) * Bc], "kps": torch.empty((Bc, cap, 7), **i32),
                "desc": torch.empty((Bc, cap, 32), dtype=torch.uint8, device="cuda"), "cnt": torch.zeros(Bc, **i32),
                "wt": torch.empty((Bc, cap), dtype=torch.float32, device="cuda"), "nm": torch.zeros(Bc, **i32)}
        for name in ("bi", "bd", "sd", "word", "node", "m12", "m21"):
            bufs[name] = torch.empty((Bc, cap), **i32)
        if use_bow:
            if d_blob is not None:
                ORBVocabulary(ex).loadFromDeviceBlob(d_blob.data_ptr(), d_blob.numel())
            else:
                ORBVocabulary(ex).loadFromBinaryBlob(blob)
        ctxs.append((ex, bufs))
    del d_blob
    ex0 = ctxs[0][0]
    cap = ex0.cap
    L = ex0._L

    def step():
        for ex, b in ctxs:
            ex.extract_batch_device(b["img"].data_ptr(), Bc, W, H, W, H * W, b["kps"].data_ptr(), b["desc"].data_ptr(), cap,
                                    b["cnt"].data_ptr())
            if use_bow:
                rc = L.orbhip_vocab_transform_device(ex.handle, b["desc"].data_ptr(), Bc * cap, LEVELSUP, b["word"].data_ptr(),
                                                     b["wt"].data_ptr(), b["node"].data_ptr())
                assert rc == 0
                rc = L.orbhip_search_by_bow_seq_device(ex.handle, b["desc"].data_ptr(), b["kps"].data_ptr(), b["cnt"].data_ptr(),
                                                       b["node"].data_ptr(), b["wt"].data_ptr(), None, cap, Bc, 1, 0,
                                                       C.c_float(NNRATIO), 1, b["m12"].data_ptr(), b["m21"].data_ptr(),
                                                       b["nm"].data_ptr())
                assert rc == 0
            if args.match in ("brute", "both"):
                rc = L.orbhip_hamming_knn2_seq_device(ex.handle, b["desc"].data_ptr(), b["cnt"].data_ptr(), cap, Bc, 1,
                                                      b["bi"].data_ptr(), b["bd"].data_ptr(), b["sd"].data_ptr())
                assert rc == 0

    def barrier():
        for ex, _ in ctxs:
            ex.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    stage = np.zeros(6, np.float64)          # per step: summed over the contexts
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        # stage device times (HIP events on the context streams; reading them waits for the step)
        for ex, _ in ctxs:
            ms = (C.c_float * 6)()
            assert L.orbhip_get_stage_times(ex.handle, ms) == 0
            stage += np.array(list(ms))
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    stage /= max(args.steps, 1)

    counts = np.concatenate([b["cnt"].cpu().numpy() for _, b in ctxs])
    nmatch = float(np.mean([b["nm"].cpu().numpy()[1:].mean() for _, b in ctxs])) if (use_bow and Bc > 1) else None
    nbrute = float(np.mean([(b["bd"].cpu().numpy()[1:] <= 50).sum() / max(Bc - 1, 1) for _, b in ctxs])) \
        if args.match in ("brute", "both") else None

    out = None
    if rank == 0:
        fps = world * B * args.steps / dt
        alg = fast_algorithmic_bytes(W, H, 8, ex0.level_size) * Bc        # bytes per FAST launch (one per context)
        fast_ms = float(stage[1]) / NC                                    # average duration of one launch
        achieved = alg / (fast_ms * 1e-3) / 1e9 if fast_ms > 0 else 0.0
        match_desc = {"bow": "vocabulary transform (k=10, L=6, levelsup 4) + ORBmatcher::SearchByBoW(0.7, checkOri)",
                      "brute": "Hamming best/second brute force",
                      "both": "vocabulary transform + SearchByBoW + Hamming brute force"}[args.match]
        traffic, traffic_src = committed_traffic(B, NC)
        out = {
            "metric": "ORB extract+match frames/sec @640x480/1000 feat",
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "640x480 frames, 1000 features, 8 levels, scale 1.2, FAST 20/7; batched "
                                   "ORBextractor + " + match_desc + " of every frame vs its predecessor",
                       "frames_per_step_per_gpu": B, "contexts": NC, "frames_per_launch": Bc,
                       "unique_frames": int(len(uniq)), "match": args.match,
                       "parallelism": "frames sharded, 1 process per GPU, no per-frame collective"},
            "roofline": {"bound": "hbm", "kernel": "k_fast", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": int(alg), "launch_ms": round(fast_ms, 4),
                         "note": "k_fast is integer-VALU bound (VALU issue ~100 % busy), not HBM bound (DESIGN.md section 4)"
                                 + ("; launch_ms is measured while the other contexts' kernels share the GPU" if NC > 1 else "")},
            "stage_ms": {"pyramid": round(float(stage[0]), 4), "fast": round(float(stage[1]), 4),
                         "quadtree": round(float(stage[2]), 4), "blur": round(float(stage[3]), 4),
                         "describe": round(float(stage[4]), 4), "last_match_kernel": round(float(stage[5]), 4)},
            "keypoints_per_frame": round(float(counts.mean()), 1),
            "bow_matches_per_frame": None if nmatch is None else round(nmatch, 1),
            "brute_matches_le_TH_LOW_per_frame": None if nbrute is None else round(nbrute, 1),
        }
        if world == 1 and args.pipelined > 1 and B % args.pipelined == 0:
            out["pipelined"] = pipelined_throughput(args, d_img, blob if use_bow else None, local_rank, cap)
        if world == 1 and args.host_batch > 0:
            out["host_fed"] = host_fed_throughput(args, uniq, blob if use_bow else None, local_rank)
        # the GPU's outputs for the first frames of the timed batch against the oracle, outside the timed region
        nver = max(0, min(args.verify, Bc))
        kept = []
        if world == 1 and args.cpu_frames > 0:
            nver = min(nver, args.cpu_frames)
            res = cpu_baseline(uniq, args.cpu_frames, args.match, blob, keep=nver)
            out["cpu_baseline"], kept = res if nver else (res, [])
            out["speedup_vs_cpu_1core"] = round(fps / out["cpu_baseline"]["value"], 1)
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(uniq, args.cpu_frames, args.match, blob)
        elif nver:
            kept = cpu_baseline(uniq, nver, args.match, blob, keep=nver)[1]
        out["verified_frames"] = verify_against_oracle(kept, ctxs[0][1], cap, args.match) if kept else 0
        out["verified_against"] = "oracle/liborb_oracle.so: keypoints (28-byte records), descriptors, " + \
            {"bow": "SearchByBoW match12/match21/count", "brute": "brute-force best/second",
             "both": "SearchByBoW and brute-force results"}[args.match] + " of frames 0..n-1 of the timed batch, bit for bit"
    for ex, _ in ctxs:
        ex.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST thing on stdout (RCCL prints its own banner lines earlier)
        C.CDLL(None).fflush(None)      # libc-buffered banner text of RCCL / the HIP runtime goes out first
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
