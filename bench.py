#!/usr/bin/env python3
"""bench.py -- ORB extract + match throughput on MI355X (BASELINE.json metric).

A step = one pass of the hot path over one batch of B synthetic frames resident in HBM -- B DISTINCT frames, the
consecutive frames of one synthetic stream (round 6; rounds 1-5 tiled 32 frames to the batch, which is now the supplementary
`tiled_check`), every one of them verified against the oracle outside the timed region:
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
I8_MFMA_PEAK_OPS = 5.0e15             # MI355X_MICROARCH.md: dense I8 MFMA = 2 x BF16 per clock, BF16 ~2.5 PFLOP/s dense
FP4_MFMA_PEAK_OPS = 10.0e15           # ... block-scaled FP4 (v_mfma_scale_f32_32x32x64_f8f6f4) = 4 x BF16 per clock
VOC_K, VOC_L, LEVELSUP = 10, 6, 4     # stock ORBvoc shape; Frame::ComputeBoW uses levelsup 4 (src/Frame.cc:744)
FAST_CEILING_FRAC = 0.276             # floor of the k_fast_fix formulation (242 M wave-instructions at full issue) as a fraction of HBM
NNRATIO = 0.7                         # TrackReferenceKeyFrame: ORBmatcher matcher(0.7,true) (src/Tracking.cc:1881)


def fast_algorithmic_bytes(w, h, nlevels, level_size):
    """sum_l (w_l - 32)(h_l - 32) bytes read once per frame."""
    tot = 0
    for l in range(nlevels):
        lw, lh = level_size(w, h, l)
        tot += (lw - 32) * (lh - 32)
    return tot


def stage_algorithmic_bytes(w, h, nlevels, level_size, keypoints):
    """SURVEY.md section 8d, per frame: pyramid = every level but the last read + every level but the first written; describe
    (round 6: k_describe_blur, the 7 x 7 blur inside the describe kernel -- no blurred pyramid is written or read) = the 43 x 43 raw
    pixels a keypoint's blurred 37 x 37 patch and its orientation disc depend on, read once, and 28 + 32 bytes written per
    keypoint.  (Rounds 1-5: k_blur read and wrote every level, k_describe read 749 disc bytes + 512 test bytes per keypoint.)"""
    px = [level_size(w, h, l) for l in range(nlevels)]
    px = [a * b for a, b in px]
    return {"k_resize (7 launches)": sum(px[:-1]) + sum(px[1:]),
            "k_describe_blur (2 launches)": int(round(keypoints * (43 * 43 + 60))),
            "k_fast": fast_algorithmic_bytes(w, h, nlevels, level_size)}


def file_sha16(path):
    import hashlib
    try:
        with open(path, "rb") as fh:
            return hashlib.sha256(fh.read()).hexdigest()[:16]
    except OSError:
        return None


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


def make_stream_frames(seed, count, workers):
    """`count` consecutive frames of the synthetic stream `seed` (orbhip.synth.make_frames: one scene under a slowly varying
    warp), drawn by a pool of forked workers -- 32 ms a frame on one core.  Call BEFORE anything touches the GPU."""
    from orbhip import synth
    return synth.make_frames_parallel(seed, W, H, count, workers)


def oracle_worker(path, first, last, match, out_path):
    """Child process of oracle_outputs_parallel: frames [first, last) of the batch through the oracle (frame first - 1 too, for the
    pair), results to an .npz.  Never touches the GPU."""
    z = np.load(path + ".meta.npz", allow_pickle=False)
    frames = np.load(path, mmap_mode="r")
    blob = z["blob"].tobytes() if z["blob"].size else None
    lo = max(first - 1, 0)
    kept = cpu_baseline(frames[lo:last], last - lo, match, blob, keep=last - lo)[1]
    out = {}
    for b in range(first, last):
        rec = kept[b - lo]
        out["k%d" % b], out["d%d" % b] = rec["k"], rec["d"]
        if "bow" in rec:
            out["nm%d" % b], out["m12_%d" % b], out["m21_%d" % b] = np.int32(rec["bow"][0]), rec["bow"][1], rec["bow"][2]
        if "knn2" in rec:
            out["bi%d" % b], out["bd%d" % b], out["sd%d" % b] = rec["knn2"]
    np.savez(out_path, **out)


def oracle_outputs_parallel(frames, match, blob, workers):
    """The oracle's outputs for EVERY frame of the batch (and every pair (b - 1, b)), computed by `workers` child processes on
    contiguous slices; the list verify_against_oracle takes.  ~11 ms of one core per frame."""
    import subprocess
    import tempfile
    n = len(frames)
    workers = max(1, min(workers, n // 4))
    bounds = [n * i // workers for i in range(workers + 1)]
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "batch.npy")
        np.save(path, frames)
        np.savez(path + ".meta.npz", blob=np.frombuffer(blob or b"", np.uint8))
        env = dict(os.environ, OMP_NUM_THREADS="1")
        procs = []
        for i in range(workers):
            outp = os.path.join(tmp, "out%d.npz" % i)
            procs.append((subprocess.Popen([sys.executable, os.path.abspath(__file__), "--oracle-worker", path, str(bounds[i]), str(bounds[i + 1]),
                                            outp, "--match", match], env=env), outp, bounds[i], bounds[i + 1]))
        kept = []
        for pr, outp, a, b in procs:
            if pr.wait() != 0:
                raise SystemExit("bench.py: an oracle worker failed")
            z = np.load(outp, allow_pickle=False)
            for f in range(a, b):
                rec = {"k": z["k%d" % f], "d": z["d%d" % f]}
                if "nm%d" % f in z.files:
                    rec["bow"] = (int(z["nm%d" % f]), z["m12_%d" % f], z["m21_%d" % f])
                if "bi%d" % f in z.files:
                    rec["knn2"] = (z["bi%d" % f], z["bd%d" % f], z["sd%d" % f])
                kept.append(rec)
    return kept


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


def verify_tiled_copies(bufs, U, B, cap, match):
    """Every frame of the batch beyond the first U + 1 is a copy of one of them (frame b = distinct frame b % U, its predecessor
    = distinct frame (b - 1) % U): its keypoint count, keypoints, descriptors and match results must equal, bit for bit, those of
    its ORIGINAL -- frame b % U for what belongs to the frame, row ((b - 1) % U) + 1 for what belongs to the pair (b - 1, b).  The
    originals are rows 0 .. U, which verify_against_oracle has compared with the oracle.  Runs on the device (a few gathers and
    compares of ~100 MB); returns the number of copies checked, raises SystemExit(3) at the first difference.  Also proves that
    the batch path is deterministic across the batch: U outputs repeated B / U times by different workgroups at different times."""
    import torch
    if B <= U + 1:
        return 0
    cnt = bufs["cnt"][:B]
    dev = cnt.device
    b = torch.arange(B, device=dev)
    orig = b % U
    porig = torch.where(b >= 1, (b - 1) % U + 1, b)
    prev = (b - 1).clamp(min=0)

    def bad(what, mask_rows):
        row = int(torch.nonzero(mask_rows)[0].item())
        print("bench.py: frame %d of the timed batch differs from its original (frame %d): %s" % (row, row % U, what), file=sys.stderr)
        raise SystemExit(3)
    d = cnt != cnt[orig]
    if bool(d.any()):
        bad("keypoint count", d)
    col = torch.arange(cap, device=dev)[None, :]
    live = col < cnt[:, None]                                  # (B, cap): features of frame b
    live_prev = col < cnt[prev][:, None]                       # features of frame b - 1
    d = ((bufs["kps"][:B] != bufs["kps"][:B][orig]).any(-1) & live).any(-1)
    if bool(d.any()):
        bad("keypoints", d)
    d = ((bufs["desc"][:B] != bufs["desc"][:B][orig]).any(-1) & live).any(-1)
    if bool(d.any()):
        bad("descriptors", d)
    pair = b >= 1
    if match in ("bow", "both"):
        d = (bufs["nm"][:B] != bufs["nm"][:B][porig]) & pair
        if bool(d.any()):
            bad("SearchByBoW match count", d)
        d = ((bufs["m21"][:B] != bufs["m21"][:B][porig]) & live).any(-1) & pair
        if bool(d.any()):
            bad("SearchByBoW match21", d)
        d = ((bufs["m12"][:B] != bufs["m12"][:B][porig]) & live_prev).any(-1) & pair
        if bool(d.any()):
            bad("SearchByBoW match12", d)
    if match in ("brute", "both"):
        for name in ("bi", "bd", "sd"):
            d = ((bufs[name][:B] != bufs[name][:B][porig]) & live).any(-1) & pair
            if bool(d.any()):
                bad("brute-force " + name, d)
    return B - (U + 1)


def gather_objects(dist, obj, world):
    """One Python object per rank on every rank (any backend); [obj] without a process group."""
    if dist is None or world == 1:
        return [obj]
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out


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


def committed_counters(batch, contexts):
    """Per-launch figures of k_fast from the committed rocprofv3 PMC passes (profiles/traffic.json, written by
    tools/summarize_prof.py from the separate --pmc passes of the round's profile script, default configuration): HBM-side bytes
    (FETCH_SIZE x2 + WRITE_SIZE as MI355X_MICROARCH.md prescribes), vector wave-instructions (SQ_INSTS_VALU), the issue weight
    of the kernel's instruction mix (tools/update_traffic_meta.py) and the sha256 of the kernel source they were taken from.  {} when the
    file does not match the run's configuration.  The counters cannot be collected inside a timed bench run (a --pmc pass
    serialises the kernels); main() marks them stale when the kernel source or the launch time has moved since."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            t = json.load(fh)
        if int(t.get("batch", -1)) == batch and int(t.get("contexts", 1)) == contexts:
            return t
    except (OSError, ValueError, KeyError):
        pass
    return {}


def batch_sweep(args):
    """The same step at larger batches, each in a child process of its own (fresh contexts and buffers): launch gaps and the last,
    partly filled round of workgroups of every kernel weigh less.  Supplementary: the headline, the profiles and every per-launch
    figure stay at --batch."""
    import subprocess
    res = {}
    for mult in (2, 3):
        b = args.batch * mult
        cmd = [sys.executable, os.path.abspath(__file__), "--batch", str(b), "--steps", "8", "--warmup", "2", "--match", args.match,
               "--cpu-frames", "0", "--pipelined", "0", "--host-batch", "0", "--configs", "0", "--content", "0", "--tiled-check", "0",
               "--verify", "-1", "--batch-sweep", "0"]
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            d = json.loads(p.stdout.strip().splitlines()[-1])
            res[str(b)] = {"value": d["value"], "unit": "frames/s", "ms_per_step": d["ms_per_step"], "verified_frames": d["verified_frames"]}
        except Exception as e:   # a supplementary figure must not take the line with it
            res[str(b)] = {"error": repr(e)[:200]}
    res["note"] = "same step, frames_per_step_per_gpu as the key; every frame of each batch verified as in the headline"
    return res


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
            "batches": nb, "seconds": round(dt, 4), "h2d_GBps": round(fps * W * H / 1e9, 2),
            "d2h_GBps": round(fps * (ex.cap * (28 + 32 + (8 if blob is not None else 0)) + 8) / 1e9, 2),
            "keypoints_per_frame": round(nkp / (nb * Bp), 1),
            "note": "orbhip_pipe_submit / orbhip_pipe_wait: frames in pinned host memory, copy-in, kernels (extract"
                    + (" + vocabulary transform + SearchByBoW" if blob is not None else "") + ") and copy-out of every "
                    "batch overlapped with its neighbours'; PCIe Gen5 x16 is 63 GB/s per direction by specification"}


# ---- BASELINE.json's other configurations, timed in the same run (outside the headline's timed region) -------------
def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orb_oracle_py as oracle
    return oracle


def _same_frame(rec_k, rec_d, k, d):
    return len(rec_k) == len(k) and rec_k.tobytes() == k.tobytes() and np.array_equal(rec_d, d)


def config_sequence(device, blob, name, lengths, W_, H_, nfeat, B, uniq=16, seed=50):
    """Configs 2 and 4: whole sequences through orbhip.streams.StreamRunner (extract + vocabulary transform + SearchByBoW of
    every frame against its predecessor, batches overlapping by one frame), one context per sequence, back to back on this
    GPU.  Verified: the sampled frames of every sequence (first, first pair, the pair across the first batch boundary, last)
    against the oracle, bit for bit."""
    from orbhip import streams, synth
    oracle = _oracle()
    runners = [streams.StreamRunner(device, B, synth.make_frames(seed + i, W_, H_, uniq), blob=blob, w=W_, h=H_, nfeat=nfeat)
               for i in range(len(lengths))]
    for r, n in zip(runners, lengths):
        r.run(min(n, B))                                   # warm-up
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r, n in zip(runners, lengths):
        r.run(n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # verification pass (not timed): sampled frames
    ref, V = oracle.Extractor(nfeat, 1.2, 8, 20, 7), oracle.Vocabulary(blob)
    verified = 0
    for r, n in zip(runners[:2], lengths[:2]):             # two sequences bound the oracle's share of the run
        got = r.run(n, streams.default_samples(n, B))
        cache = {}

        def rf(t, r=r, cache=cache):
            if t not in cache:
                k, d = ref(r.frame(t))
                _, wt, nid = V.transform(d, streams.LEVELSUP)
                cache[t] = (k, d, oracle.feature_vector(nid, wt))
            return cache[t]
        for t, rec in sorted(got.items()):
            k, d, fv = rf(t)
            if rec["n"] != len(k) or rec["kps"] != k.tobytes() or not np.array_equal(rec["desc"], d):
                raise SystemExit("bench.py: config %s: frame %d differs from the oracle" % (name, t))
            if t >= 1:
                pk, pd, pfv = rf(t - 1)
                nm, m12, m21 = oracle.search_by_bow(pd, np.ones(len(pd), np.uint8), pk["angle"], pfv, d, None, k["angle"], fv,
                                                    th=50, th_mode=0, nnratio=NNRATIO, check_ori=True)
                if rec["nm"] != nm or not np.array_equal(rec["m12"], m12) or not np.array_equal(rec["m21"], m21):
                    raise SystemExit("bench.py: config %s: SearchByBoW of frame %d differs from the oracle" % (name, t))
            verified += 1
    for r in runners:
        r.close()
    total = int(sum(lengths))
    return {"workload": "%s: %s frames %dx%d, %d features, extract + vocabulary transform + SearchByBoW vs previous frame, "
                        "batches of %d overlapping by one frame, one context per sequence" % (
                            name, "+".join(str(n) for n in lengths), W_, H_, nfeat, B),
            "value": round(total / dt, 1), "unit": "frames/s", "frames": total, "seconds": round(dt, 4), "verified": verified,
            "verified_what": "sampled frames (keypoints, descriptors, SearchByBoW match12 / match21 / count) vs oracle"}


def config_stereo(device, blob, B=128, steps=6):
    """Config 3: KITTI 00 stereo, 1241x376, 2000 features (Examples/Stereo/KITTI00-02.yaml): left + right extraction on two
    contexts, Frame::ComputeStereoMatches on the resident pyramids, vocabulary transform + SearchByBoW of every left frame
    against its predecessor.  Verified: pairs 0 and 1 (keypoints, descriptors, mvuRight / mvDepth as bit patterns, the match
    of left frame 1 against left frame 0) against the oracle."""
    import torch
    from orbhip import synth
    from orbhip.extractor import ORBextractor
    from orbhip.vocabulary import ORBVocabulary
    oracle = _oracle()
    W_, H_, NF, MB, MBF = 1241, 376, 2000, 0.53716, 386.1448
    uniq = 8
    pairs = [synth.make_stereo_pair(300 + i, W_, H_, disparity=10 + 3 * i) for i in range(uniq)]
    stride = (W_ + 15) // 16 * 16
    host = np.zeros((2, B, H_, stride), np.uint8)
    for b in range(B):
        host[0, b, :, :W_], host[1, b, :, :W_] = pairs[b % uniq]
    d_img = torch.from_numpy(host).cuda(device)
    exs = [ORBextractor(NF, 1.2, 8, 20, 7, max_w=W_, max_h=H_, max_batch=B, device=device) for _ in range(2)]
    ORBVocabulary(exs[0]).loadFromBinaryBlob(blob)
    cap = exs[0].cap
    dev = torch.device("cuda", device)
    i32 = dict(dtype=torch.int32, device=dev)
    d_kps = torch.empty((2, B, cap, 7), **i32)
    d_desc = torch.empty((2, B, cap, 32), dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros((2, B), **i32)
    d_u = torch.empty((B, cap), dtype=torch.float32, device=dev)
    d_z = torch.empty_like(d_u)
    d_ns = torch.zeros(B, **i32)
    d_word, d_node, d_m12, d_m21 = (torch.empty((B, cap), **i32) for _ in range(4))
    d_wt = torch.empty((B, cap), dtype=torch.float32, device=dev)
    d_nm = torch.zeros(B, **i32)
    L = exs[0]._L

    def step():
        for s in range(2):
            exs[s].extract_batch_device(d_img[s].data_ptr(), B, W_, H_, stride, H_ * stride, d_kps[s].data_ptr(),
                                        d_desc[s].data_ptr(), cap, d_cnt[s].data_ptr())
        assert L.orbhip_stereo_match_device(exs[0].handle, exs[1].handle, d_kps[0].data_ptr(), d_desc[0].data_ptr(),
                                            d_cnt[0].data_ptr(), d_kps[1].data_ptr(), d_desc[1].data_ptr(), d_cnt[1].data_ptr(), cap,
                                            B, MB, MBF, d_u.data_ptr(), d_z.data_ptr(), d_ns.data_ptr()) == 0
        assert L.orbhip_vocab_transform_device(exs[0].handle, d_desc[0].data_ptr(), B * cap, LEVELSUP, d_word.data_ptr(),
                                               d_wt.data_ptr(), d_node.data_ptr()) == 0
        assert L.orbhip_search_by_bow_seq_device(exs[0].handle, d_desc[0].data_ptr(), d_kps[0].data_ptr(), d_cnt[0].data_ptr(),
                                                 d_node.data_ptr(), d_wt.data_ptr(), None, cap, B, 1, 0, C.c_float(NNRATIO), 1,
                                                 d_m12.data_ptr(), d_m21.data_ptr(), d_nm.data_ptr()) == 0
    for _ in range(2):
        step()
    for e in exs:
        e.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    for e in exs:
        e.sync()
    dt = time.perf_counter() - t0
    # verification (pairs 0 and 1 of the batch)
    oL, oR, V = oracle.Extractor(NF, 1.2, 8, 20, 7), oracle.Extractor(NF, 1.2, 8, 20, 7), oracle.Vocabulary(blob)
    cnt = d_cnt.cpu().numpy()
    prev = None
    verified = 0
    for b in range(2):
        kL, dL = oL(pairs[b][0])
        kR, dR = oR(pairs[b][1])
        for s, (k, d) in enumerate(((kL, dL), (kR, dR))):
            n = int(cnt[s, b])
            gk = d_kps[s, b, :n].cpu().numpy().tobytes()
            if n != len(k) or gk != k.tobytes() or not np.array_equal(d_desc[s, b, :n].cpu().numpy(), d):
                raise SystemExit("bench.py: config 3: keypoints / descriptors of pair %d differ from the oracle" % b)
        ru, rz, rn = oracle.stereo_matches(oL, kL, dL, oR, kR, dR, MB, MBF)
        if int(d_ns[b].item()) != rn or d_u[b, :len(kL)].cpu().numpy().tobytes() != ru.tobytes() or \
                d_z[b, :len(kL)].cpu().numpy().tobytes() != rz.tobytes():
            raise SystemExit("bench.py: config 3: ComputeStereoMatches of pair %d differs from the oracle" % b)
        _, wt, nid = V.transform(dL, LEVELSUP)
        fv = oracle.feature_vector(nid, wt)
        if prev is not None:
            pk, pd, pfv = prev
            nm, m12, m21 = oracle.search_by_bow(pd, np.ones(len(pd), np.uint8), pk["angle"], pfv, dL, None, kL["angle"], fv, th=50,
                                                th_mode=0, nnratio=NNRATIO, check_ori=True)
            if int(d_nm[b].item()) != nm or not np.array_equal(d_m12[b, :len(pk)].cpu().numpy(), m12) or \
                    not np.array_equal(d_m21[b, :len(kL)].cpu().numpy(), m21):
                raise SystemExit("bench.py: config 3: SearchByBoW of left frame %d differs from the oracle" % b)
        prev = (kL, dL, fv)
        verified += 1
    depth_pts = float((d_u >= 0).sum().item()) / B
    for e in exs:
        e.close()
    return {"workload": "KITTI 00 stereo: 1241x376 pairs, 2000 features, batches of %d pairs: extract left + right (two contexts), "
                        "ComputeStereoMatches on the resident pyramids, vocabulary transform + SearchByBoW of consecutive left "
                        "frames" % B,
            "value": round(steps * B / dt, 1), "unit": "stereo pairs/s", "images_per_s": round(2 * steps * B / dt, 1),
            "pairs": steps * B, "seconds": round(dt, 4), "depth_points_per_pair": round(depth_pts, 1), "verified": verified,
            "verified_what": "pairs 0 and 1: keypoints, descriptors, mvuRight / mvDepth bit patterns, SearchByBoW vs oracle"}


def config_relocalisation(device, B=256, steps=6, nq=4000, ndb=1000000, nver=32):
    """Config 5: TUM fr1_desk geometry (640x480) at 4000 features, extraction + brute-force best / second against the previous
    frame, and ONE relocalisation query of 4000 descriptors against a database of 1 000 000 (orbhip_hamming_knn2_device).
    Verified: frames 0 and 1 of the batch (keypoints, descriptors, brute-force triples) and `nver` of the 4000 queries against
    the full database, all against the oracle."""
    import torch
    from orbhip import synth
    from orbhip.extractor import ORBextractor
    oracle = _oracle()
    NF = 4000
    uniq = synth.make_frames(3, W, H, 8)
    frames = np.concatenate([uniq] * (B // 8))
    dev = torch.device("cuda", device)
    d_img = torch.from_numpy(np.ascontiguousarray(frames)).cuda(device)
    ex = ORBextractor(NF, 1.2, 8, 20, 7, max_w=W, max_h=H, max_batch=B, device=device)
    cap = ex.cap
    i32 = dict(dtype=torch.int32, device=dev)
    d_kps = torch.empty((B, cap, 7), **i32)
    d_desc = torch.empty((B, cap, 32), dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(B, **i32)
    d_bi, d_bd, d_sd = (torch.empty((B, cap), **i32) for _ in range(3))
    L = ex._L

    def step():
        ex.extract_batch_device(d_img.data_ptr(), B, W, H, W, H * W, d_kps.data_ptr(), d_desc.data_ptr(), cap, d_cnt.data_ptr())
        assert L.orbhip_hamming_knn2_seq_device(ex.handle, d_desc.data_ptr(), d_cnt.data_ptr(), cap, B, 1, d_bi.data_ptr(),
                                                d_bd.data_ptr(), d_sd.data_ptr()) == 0
    for _ in range(2):
        step()
    ex.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ex.sync()
    dt = time.perf_counter() - t0
    ref = oracle.Extractor(NF, 1.2, 8, 20, 7)
    cnt = d_cnt.cpu().numpy()
    prev, verified = None, 0
    for b in range(2):
        k, d = ref(frames[b])
        n = int(cnt[b])
        if n != len(k) or d_kps[b, :n].cpu().numpy().tobytes() != k.tobytes() or not np.array_equal(d_desc[b, :n].cpu().numpy(), d):
            raise SystemExit("bench.py: config 5: frame %d differs from the oracle" % b)
        if prev is not None:
            bi, bd, sd = oracle.knn2(d, prev)
            if not (np.array_equal(d_bi[b, :n].cpu().numpy(), bi) and np.array_equal(d_bd[b, :n].cpu().numpy(), bd)
                    and np.array_equal(d_sd[b, :n].cpu().numpy(), sd)):
                raise SystemExit("bench.py: config 5: brute-force match of frame %d differs from the oracle" % b)
        prev = d
        verified += 1
    kp = float(cnt.mean())
    # the 1M-descriptor query
    gen = torch.Generator(device=dev)
    gen.manual_seed(77)
    db = torch.randint(0, 256, (ndb, 32), dtype=torch.uint8, device=dev, generator=gen)
    idx = torch.randint(0, ndb, (nq,), device=dev, generator=gen)
    q = db[idx].clone()
    flips = torch.randint(0, 256, (nq, 4), dtype=torch.uint8, device=dev, generator=gen)
    q[:, :4] ^= flips & 0x11                                  # a few bit flips: best / second structure is not trivial
    bi = torch.empty(nq, **i32)
    bd, sd = torch.empty_like(bi), torch.empty_like(bi)
    torch.cuda.synchronize()
    for _ in range(3):
        assert L.orbhip_hamming_knn2_device(ex.handle, q.data_ptr(), nq, db.data_ptr(), ndb, bi.data_ptr(), bd.data_ptr(),
                                            sd.data_ptr()) == 0
    ex.sync()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        L.orbhip_hamming_knn2_device(ex.handle, q.data_ptr(), nq, db.data_ptr(), ndb, bi.data_ptr(), bd.data_ptr(), sd.data_ptr())
    ex.sync()
    qdt = (time.perf_counter() - t0) / reps
    hq, hdb = q[:nver].cpu().numpy(), db.cpu().numpy()
    rbi, rbd, rsd = oracle.knn2(hq, hdb)
    if not (np.array_equal(bi[:nver].cpu().numpy(), rbi) and np.array_equal(bd[:nver].cpu().numpy(), rbd)
            and np.array_equal(sd[:nver].cpu().numpy(), rsd)):
        raise SystemExit("bench.py: config 5: the 1M-descriptor query differs from the oracle")
    ex.close()
    return {"workload": "TUM fr1_desk geometry: 640x480 at 4000 features, batches of %d: extract + brute-force best/second vs the "
                        "previous frame; and one %d x %d Hamming relocalisation query" % (B, nq, ndb),
            "value": round(steps * B / dt, 1), "unit": "frames/s", "keypoints_per_frame": round(kp, 1),
            "query_ms": round(qdt * 1e3, 3), "query_pair_evals_per_s": round(nq * ndb / qdt, 0),
            # k_knn2_mfma computes a pair's distance as a 256-term FP4 dot product on the matrix pipe: 512 operations per pair
            "roofline_mfma": {"bound": "mfma", "kernel": "k_knn2_mfma", "achieved": round(nq * ndb / qdt * 512 / 1e12, 1),
                              "peak": FP4_MFMA_PEAK_OPS / 1e12, "unit": "TOP/s (FP4, exact on 0 / 1 / -1 operands)",
                              "frac": round(nq * ndb / qdt * 512 / FP4_MFMA_PEAK_OPS, 4),
                              "frac_of_int8_rate": round(nq * ndb / qdt * 512 / I8_MFMA_PEAK_OPS, 4),
                              "note": "pair evaluations/s x 512 operations / the dense FP4 MFMA rate (4 x BF16 per clock, "
                                      "MI355X_MICROARCH.md) the kernel's v_mfma_scale_f32_32x32x64_f8f6f4 runs at; frac_of_int8_rate "
                                      "prices the same work at the int8 rate of the r03 kernel (v_mfma_i32_32x32x32_i8); "
                                      "the query's wall time incl. the merge kernel"},
            "query_database_GBps": round(32.0 * ndb / qdt / 1e9, 1), "verified": verified + nver,
            "verified_what": "frames 0 and 1 (keypoints, descriptors, brute-force triples) and %d of the %d queries against the "
                             "full database vs oracle" % (nver, nq)}


def content_classes(device, blob, B=256, steps=4, verify=True):
    """The headline step (extract + vocabulary transform + SearchByBoW) on each synthetic content class of orbhip/synth.py: FAST's
    cost depends on what it looks at (the dense fallbacks of its lists, the second pass of empty cells), so the headline is a
    property of its texture.  Per class: frames/s, the FAST launch time, and frames 0 and 1 verified against the oracle."""
    import torch
    from orbhip import synth
    from orbhip.extractor import ORBextractor
    from orbhip.vocabulary import ORBVocabulary
    oracle = _oracle()
    out = {}
    t_all = time.perf_counter()
    dev = torch.device("cuda", device)
    i32 = dict(dtype=torch.int32, device=dev)
    ex = ORBextractor(NFEAT, 1.2, 8, 20, 7, max_w=W, max_h=H, max_batch=B, device=device)
    ORBVocabulary(ex).loadFromBinaryBlob(blob)
    cap = ex.cap
    L = ex._L
    d_kps = torch.empty((B, cap, 7), **i32)
    d_desc = torch.empty((B, cap, 32), dtype=torch.uint8, device=dev)
    d_cnt, d_nm = torch.zeros(B, **i32), torch.zeros(B, **i32)
    d_word, d_node, d_m12, d_m21 = (torch.empty((B, cap), **i32) for _ in range(4))
    d_wt = torch.empty((B, cap), dtype=torch.float32, device=dev)
    ref = oracle.Extractor(NFEAT, 1.2, 8, 20, 7)
    voc = oracle.Vocabulary(blob)
    for kind in synth.CONTENT_CLASSES + ("photographs",):
        # (photographs: the sample pictures scikit-learn / matplotlib install in this image, read where they lie -- the one class
        # that is not orbhip/synth.py's own drawing; left out when they are absent)
        uniq = synth.photograph_frames(W, H, 8) if kind == "photographs" else synth.make_frames_class(kind, 2000, W, H, 8)
        if uniq is None:
            continue
        frames = np.concatenate([uniq] * (B // 8))
        d_img = torch.from_numpy(np.ascontiguousarray(frames)).cuda(device)

        def step():
            ex.extract_batch_device(d_img.data_ptr(), B, W, H, W, H * W, d_kps.data_ptr(), d_desc.data_ptr(), cap, d_cnt.data_ptr())
            assert L.orbhip_vocab_transform_device(ex.handle, d_desc.data_ptr(), B * cap, LEVELSUP, d_word.data_ptr(), d_wt.data_ptr(),
                                                   d_node.data_ptr()) == 0
            assert L.orbhip_search_by_bow_seq_device(ex.handle, d_desc.data_ptr(), d_kps.data_ptr(), d_cnt.data_ptr(), d_node.data_ptr(),
                                                     d_wt.data_ptr(), None, cap, B, 1, 0, C.c_float(NNRATIO), 1, d_m12.data_ptr(),
                                                     d_m21.data_ptr(), d_nm.data_ptr()) == 0
        step()
        ex.sync()
        t0 = time.perf_counter()
        fast = 0.0
        for _ in range(steps):
            step()
            ms = (C.c_float * 6)()
            assert L.orbhip_get_stage_times(ex.handle, ms) == 0
            fast += ms[1] / steps
        ex.sync()
        dt = time.perf_counter() - t0
        cnt = d_cnt.cpu().numpy()
        prev = None
        nv = min(len(uniq) + 1, B) if verify else 0   # every distinct frame and the pair across the first tile boundary
        for b in range(nv):                           # (--verify 0: timing ablations, whose results are invalid by design)
            k, d = ref(frames[b])
            n = int(cnt[b])
            if n != len(k) or d_kps[b, :n].cpu().numpy().tobytes() != k.tobytes() or not np.array_equal(d_desc[b, :n].cpu().numpy(), d):
                raise SystemExit("bench.py: content class %s: frame %d differs from the oracle" % (kind, b))
            _, wt, nid = voc.transform(d, LEVELSUP)
            cur = (k, d, oracle.feature_vector(nid, wt))
            if prev is not None:
                nm, m12, m21 = oracle.search_by_bow(prev[1], np.ones(len(prev[1]), np.uint8), prev[0]["angle"], prev[2], d, None,
                                                    k["angle"], cur[2], th=50, th_mode=0, nnratio=NNRATIO, check_ori=True)
                if int(d_nm[b].item()) != nm or not np.array_equal(d_m21[b, :len(m21)].cpu().numpy(), m21):
                    raise SystemExit("bench.py: content class %s: SearchByBoW of frame %d differs from the oracle" % (kind, b))
            prev = cur
        # ... and every tiled copy against its original, on the device
        ncopies = verify_tiled_copies({"cnt": d_cnt, "kps": d_kps, "desc": d_desc, "nm": d_nm, "m12": d_m12, "m21": d_m21}, len(uniq), B, cap,
                                      "bow") if verify else 0
        out[kind] = {"value": round(steps * B / dt, 1), "unit": "frames/s", "k_fast_ms_per_1024_frames": round(fast * 1024.0 / B, 4),
                     "keypoints_per_frame": round(float(cnt.mean()), 1), "bow_matches_per_frame": round(float(d_nm.cpu().numpy()[1:].mean()), 1),
                     "verified_frames": nv + ncopies}
        del d_img
    ex.close()
    out["frames_per_step"] = B
    out["note"] = ("batches of %d frames (8 distinct, tiled), %d timed steps each; every distinct frame and the pair across the first tile boundary "
                   "against the oracle, every copy against its original on the device; the headline runs the textured class at 1024" % (B, steps))
    out["seconds_total"] = round(time.perf_counter() - t_all, 1)
    return out


def config_streams_ranks(device, blob, rank, world, dist, barrier, B=512, uniq=16):
    """BASELINE config 4 proper under N ranks: the four EuRoC sequences (V1_01 / V1_02 / V2_01 / MH_02; twice for eight ranks), whole
    streams assigned to ranks longest first (orbhip.distributed.assign_streams: one stream per rank from N = 4 on), every rank runs
    its streams through orbhip.streams.StreamRunner on its own GPU -- no exchange per frame; the vocabulary has already travelled.
    value = all frames / the slowest rank's time.  Every rank checks the sampled frames of its first stream against the oracle."""
    from orbhip import distributed as D
    from orbhip import streams, synth
    oracle = _oracle()
    lengths = [n for _, n in streams.EUROC_STREAMS] * max(1, (world + 3) // 4)
    lengths = lengths[:max(4, world)]
    plan = D.assign_streams(lengths, world)
    mine = plan[rank]
    runners = [(si, streams.StreamRunner(device, B, synth.make_frames(2000 + si, streams.W, streams.H, uniq), blob=blob)) for si in mine]
    for si, r in runners:
        r.run(min(lengths[si], B))                              # warm-up
    barrier()
    t0 = time.perf_counter()
    for si, r in runners:
        r.run(lengths[si])
    barrier()
    dt = time.perf_counter() - t0
    verified = 0
    if runners:
        si, r = runners[0]
        n = lengths[si]
        got = r.run(n, streams.default_samples(n, B))
        ref, V = oracle.Extractor(streams.NFEAT, 1.2, 8, 20, 7), oracle.Vocabulary(blob)
        cache = {}

        def rf(t):
            if t not in cache:
                k, d = ref(r.frame(t))
                _, wt, nid = V.transform(d, streams.LEVELSUP)
                cache[t] = (k, d, oracle.feature_vector(nid, wt))
            return cache[t]
        for t, rec in sorted(got.items()):
            k, d, fv = rf(t)
            if rec["n"] != len(k) or rec["kps"] != k.tobytes() or not np.array_equal(rec["desc"], d):
                raise SystemExit("bench.py: config 4, rank %d: frame %d of stream %d differs from the oracle" % (rank, t, si))
            if t >= 1:
                pk, pd, pfv = rf(t - 1)
                nm, m12, m21 = oracle.search_by_bow(pd, np.ones(len(pd), np.uint8), pk["angle"], pfv, d, None, k["angle"], fv,
                                                    th=50, th_mode=0, nnratio=NNRATIO, check_ori=True)
                if rec["nm"] != nm or not np.array_equal(rec["m12"], m12) or not np.array_equal(rec["m21"], m21):
                    raise SystemExit("bench.py: config 4, rank %d: SearchByBoW of frame %d differs from the oracle" % (rank, t))
            verified += 1
    for _, r in runners:
        r.close()
    allr = gather_objects(dist, (dt, sum(lengths[si] for si in mine), verified, len(mine)), world)
    tmax = max(a[0] for a in allr)
    total = sum(a[1] for a in allr)
    return {"workload": "EuRoC %s: %s frames %dx%d, %d features, one stream per rank from 4 ranks on (assignment %s), extract + "
                        "vocabulary transform + SearchByBoW vs previous frame, batches of %d overlapping by one frame" % (
                            " / ".join(nm for nm, _ in streams.EUROC_STREAMS), "+".join(str(n) for n in lengths), streams.W, streams.H,
                            streams.NFEAT, plan, B),
            "value": round(total / tmax, 1), "unit": "frames/s", "frames": int(total), "seconds": round(tmax, 4), "ranks": world,
            "per_rank": [{"rank": i, "streams": a[3], "frames": a[1], "seconds": round(a[0], 4),
                          "frames_per_s": round(a[1] / a[0], 1) if a[0] > 0 else None, "verified": a[2]} for i, a in enumerate(allr)],
            "verified": sum(a[2] for a in allr),
            "verified_what": "sampled frames of every rank's first stream (keypoints, descriptors, SearchByBoW) vs oracle"}


def secondary_configs(device, blob):
    """BASELINE.json configs 2-5 on this GPU, each with its own check against the oracle; ~40 s in total."""
    from orbhip.streams import EUROC_STREAMS
    out = {}
    t0 = time.perf_counter()
    out["2_euroc_mh01_sequence"] = config_sequence(device, blob, "EuRoC MH_01", [3682], 752, 480, 1000, 512)
    out["3_kitti00_stereo"] = config_stereo(device, blob)
    out["4_euroc_streams_one_gpu"] = config_sequence(device, blob, "V1_01 / V1_02 / V2_01 / MH_02 back to back on one GPU "
                                                     "(one sequence per GPU needs four)", [n for _, n in EUROC_STREAMS], 752, 480,
                                                     1000, 512, seed=60)
    out["5_tum_4000feat_1M_query"] = config_relocalisation(device)
    out["seconds_total"] = round(time.perf_counter() - t0, 1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="frames per step per GPU")
    ap.add_argument("--unique", type=int, default=0, help="distinct synthetic frames; 0 = the whole batch (default since round 6: every "
                    "frame of the timed batch is a different frame of one synthetic stream); a smaller number is tiled to the batch")
    ap.add_argument("--match", choices=["bow", "brute", "both"], default="bow")
    ap.add_argument("--contexts", type=int, default=1, help="extractor contexts the batch is split over in the timed region")
    ap.add_argument("--pipelined", type=int, default=2, help="also report the free-running throughput with this many "
                    "contexts (0 = skip); supplementary, never `value`")
    ap.add_argument("--cpu-frames", type=int, default=800, help="frames of the CPU baseline sample (0 = skip)")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--oracle-worker", nargs=4, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--host-batch", type=int, default=256, help="frames per batch of the host-fed pipeline figure (0 = skip)")
    ap.add_argument("--configs", type=int, default=1, help="also time BASELINE.json's configs 2-5 (outside the headline's timed "
                    "region, each verified against the oracle); 0 = skip")
    ap.add_argument("--content", type=int, default=1, help="also run the step on the synthetic content classes of orbhip/synth.py "
                    "(textured, indoor_sparse, white_noise, low_contrast), each verified against the oracle; 0 = skip")
    ap.add_argument("--verify", type=int, default=-1, help="frames of the timed batch whose GPU outputs are compared with "
                    "the oracle outside the timed region (0 = skip; -1 = every distinct frame, by oracle processes on the host's "
                    "cores, and the pair across the first tile boundary when the batch is tiled); a difference ends the run with exit "
                    "code 3.  With one context every tiled copy is then compared ON THE DEVICE with its original, so that the whole "
                    "timed batch is verified")
    ap.add_argument("--tiled-check", type=int, default=1, help="also time a few steps on the first 32 frames TILED to the batch "
                    "(the headline of rounds 1-5): what tiling gains; 0 = skip")
    ap.add_argument("--no-tiling", type=int, default=None, help=argparse.SUPPRESS)   # rounds 1-5's name of the opposite check (tools/*.sh)
    ap.add_argument("--backend", choices=["nccl", "gloo"], default=os.environ.get("ORBHIP_BENCH_BACKEND", "nccl"),
                    help="torch.distributed backend of an N > 1 run; gloo only for ranks that share one device (tests): RCCL "
                    "refuses duplicate devices, the vocabulary then travels through host memory")
    ap.add_argument("--batch-sweep", type=int, default=1, help="also time the same step at 2 and 3 times the batch in child processes "
                    "(the headline stays at --batch: the profiles and every per-launch figure refer to it)")
    ap.add_argument("--streams-config", type=int, default=-1, help="N > 1: also run BASELINE config 4 proper, one EuRoC stream per "
                    "rank (-1 = when N >= 4; 1 = always; 0 = never)")
    args = ap.parse_args()
    if args.no_tiling is not None:
        args.tiled_check = args.no_tiling
    if args.cpu_worker:
        cpu_worker(args.cpu_worker, args.cpu_frames, args.match)
        return
    if args.oracle_worker:
        oracle_worker(args.oracle_worker[0], int(args.oracle_worker[1]), int(args.oracle_worker[2]), args.match, args.oracle_worker[3])
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

    # the batch's frames: drawn by forked workers BEFORE this process touches the GPU (a process that has initialised HIP does
    # not fork); independent streams per rank (weak scaling: per-GPU work fixed)
    host_workers = max(1, min(len(os.sched_getaffinity(0)) // max(world, 1), 32))
    n_uniq = args.batch if args.unique <= 0 else min(args.unique, args.batch)
    t_gen = time.perf_counter()
    uniq = make_stream_frames(1000 + rank, n_uniq, host_workers)
    t_gen = time.perf_counter() - t_gen

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
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    # N > 1: every rank next to its GPU (CPUs of the GPU's NUMA node), before anything page-locked is allocated
    numa = D.numa_bind(local_rank, bind=os.environ.get("ORBHIP_BENCH_NUMA_BIND", "1") != "0") if dist is not None else None

    from orbhip import synth
    from orbhip.extractor import ORBextractor
    from orbhip.vocabulary import ORBVocabulary

    B = args.batch
    NC = max(1, min(args.contexts, B))
    while B % NC:
        NC -= 1
    Bc = B // NC                                           # frames per context
    reps = (B + len(uniq) - 1) // len(uniq)
    frames = uniq if reps == 1 else np.concatenate([uniq] * reps)[:B]
    d_img = torch.from_numpy(np.ascontiguousarray(frames)).cuda()          # (B, H, W), stride W (multiple of 16)

    # ORB vocabulary: reference binary format (TemplatedVocabulary.h:1727-1751), synthetic tree of the
    # stock shape (k=10, L=6, 1.11 M nodes, 45.6 MB).  Rank 0 builds it; N>1: one RCCL broadcast over xGMI.
    use_bow = args.match in ("bow", "both")
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
        ctxs.append((ex, bufs))
    # ORB vocabulary: reference binary format (TemplatedVocabulary.h:1727-1751), synthetic tree of the stock shape (k=10,
    # L=6, 1.11 M nodes, 45.6 MB).  Rank 0 builds it; N > 1: ONE broadcast over xGMI through the C ABI's own RCCL
    # communicator (orbhip_comm_init + orbhip_bcast_blob_device; the 128-byte unique id travels through torch's group).
    blob = None
    d_blob = None
    rccl_ranks, bcast_ms = None, None
    if use_bow or dist is not None:
        blob = D.make_synthetic_vocabulary(4242, VOC_K, VOC_L) if rank == 0 else b""
        if dist is not None and args.backend != "nccl":
            buf = D.broadcast_blob(blob, src=0, device="cpu")
            blob = bytes(buf.numpy().tobytes())
        elif dist is not None:
            from orbhip import streams

            def exchange(u):
                t = torch.frombuffer(bytearray(u), dtype=torch.uint8).cuda()
                dist.broadcast(t, src=0)
                return bytes(t.cpu().numpy().tobytes())
            torch.cuda.synchronize()
            tb = time.perf_counter()
            d_blob, rccl_ranks = streams.comm_broadcast_vocabulary(ctxs[0][0], blob, rank, world, exchange)
            bcast_ms = (time.perf_counter() - tb) * 1e3
            if rccl_ranks != world:
                raise SystemExit("bench.py: the RCCL communicator reports %d ranks, WORLD_SIZE is %d" % (rccl_ranks, world))
    if use_bow:
        for ex, _ in ctxs:
            if d_blob is not None:
                ORBVocabulary(ex).loadFromDeviceBlob(d_blob.data_ptr(), d_blob.numel())
            else:
                ORBVocabulary(ex).loadFromBinaryBlob(blob)
    if d_blob is not None and rank != 0:
        blob = bytes(d_blob.cpu().numpy().tobytes())      # (every rank checks its own outputs against the oracle)
    del d_blob
    ex0 = ctxs[0][0]
    cap = ex0.cap
    L = ex0._L

    def step(img=None):
        for ci, (ex, b) in enumerate(ctxs):
            src = b["img"] if img is None else img[ci * Bc:(ci + 1) * Bc]
            ex.extract_batch_device(src.data_ptr(), Bc, W, H, W, H * W, b["kps"].data_ptr(), b["desc"].data_ptr(), cap,
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
    # Timed region: the library records only the two HIP events around the FAST launch (orbhip_set_stage_timing 1 -- the
    # roofline needs that kernel's duration; every further event record between two kernels of a stream costs ~4 us of device
    # time, 0.5 % of a step for the full set).  The other stages are timed afterwards, in a short pass outside the timed region.
    for ex, _ in ctxs:
        assert L.orbhip_set_stage_timing(ex.handle, 1) == 0
    stage = np.zeros(6, np.float64)          # per step: summed over the contexts
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        # FAST launch time (HIP events on the launch stream; reading them waits for the step)
        for ex, _ in ctxs:
            ms = (C.c_float * 6)()
            assert L.orbhip_get_stage_times(ex.handle, ms) == 0
            stage[1] += ms[1]
    barrier()
    dt = time.perf_counter() - t0
    for ex, _ in ctxs:
        assert L.orbhip_set_stage_timing(ex.handle, 2) == 0
    stage_pass = max(1, min(5, args.steps))
    fast_instrumented = 0.0
    for _ in range(stage_pass):
        step()
        for ex, _ in ctxs:
            ms = (C.c_float * 6)()
            assert L.orbhip_get_stage_times(ex.handle, ms) == 0
            for i in (0, 2, 3, 4, 5):
                stage[i] += ms[i] * max(args.steps, 1) / stage_pass
            fast_instrumented += ms[1] / stage_pass
    barrier()
    per_rank_dt = [float(x) for x in gather_objects(dist, dt, world)]
    dt = max(per_rank_dt)                                      # the MAX over ranks is the job's time
    stage /= max(args.steps, 1)

    counts = np.concatenate([b["cnt"].cpu().numpy() for _, b in ctxs])
    nmatch = float(np.mean([b["nm"].cpu().numpy()[1:].mean() for _, b in ctxs])) if (use_bow and Bc > 1) else None
    nbrute = float(np.mean([(b["bd"].cpu().numpy()[1:] <= 50).sum() / max(Bc - 1, 1) for _, b in ctxs])) \
        if args.match in ("brute", "both") else None

    out = None
    if rank == 0:
        fps = world * B * args.steps / dt if dt > 0 else 0.0
        alg = fast_algorithmic_bytes(W, H, 8, ex0.level_size) * Bc        # bytes per FAST launch (one per context)
        fast_ms = float(stage[1]) / NC                                    # average duration of one launch
        achieved = alg / (fast_ms * 1e-3) / 1e9 if fast_ms > 0 else 0.0
        match_desc = {"bow": "vocabulary transform (k=10, L=6, levelsup 4) + ORBmatcher::SearchByBoW(0.7, checkOri)",
                      "brute": "Hamming best/second brute force",
                      "both": "vocabulary transform + SearchByBoW + Hamming brute force"}[args.match]
        ctr = committed_counters(B, NC)
        traffic, traffic_src = ctr.get("traffic_bytes_per_launch"), ctr.get("source")
        valu = ctr.get("valu_wave_insts_per_launch")
        # the committed counters describe THIS kernel only if its source has not changed since and the launch takes what it took then
        src_now = file_sha16(os.path.join(ROOT, "vi-orb-slam-icra2018_amd", "csrc", "k_fast.hip"))
        src_moved = bool(ctr) and ctr.get("kernel_source_sha16") != src_now
        time_moved = bool(ctr) and fast_ms > 0 and abs(float(ctr.get("avg_launch_us", 0.0)) / 1e3 - fast_ms) > 0.05 * fast_ms
        ctr_stale = src_moved or time_moved
        VALU_ISSUE_WEIGHT = float(ctr.get("issue_weight", 0.835))
        kp_mean = float(counts.mean())
        alg_stage = stage_algorithmic_bytes(W, H, 8, ex0.level_size, kp_mean)
        stage_of = {"k_resize (7 launches)": 0, "k_describe_blur (2 launches)": 4, "k_fast": 1}
        rooflines = []
        for kname, by in alg_stage.items():
            ms_k = float(stage[stage_of[kname]]) / NC
            rooflines.append({"kernel": kname, "algorithmic_bytes": int(by * Bc), "ms": round(ms_k, 4),
                              "achieved_GBps": round(by * Bc / (ms_k * 1e-3) / 1e9, 1) if ms_k > 0 else 0.0,
                              "frac": round(by * Bc / (ms_k * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms_k > 0 else 0.0})
        # the whole step against the HBM roofline: the streaming kernels' algorithmic bytes + what the matching stage reads and writes
        # per keypoint (descriptor 32 B read by the transform and twice by SearchByBoW -- as side 1 and as side 2 -- word / weight /
        # node 12 B written and read, match12 / match21 8 B written)
        step_bytes = (sum(alg_stage.values()) + (kp_mean * (3 * 32 + 2 * 12 + 8) if use_bow else 0.0)) * B
        # vector-issue roofline: 1024 SIMDs, one wave64 instruction per 4 cycles each, at the clock the counters saw
        issue_peak = 1024 / 4.0 * float(ctr.get("clock_ghz", 2.4)) * 1e9
        out = {
            "metric": "ORB extract+match frames/sec @640x480/1000 feat",
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "640x480 frames, 1000 features, 8 levels, scale 1.2, FAST 20/7; batched "
                                   "ORBextractor + " + match_desc + " of every frame vs its predecessor",
                       "frames_per_step_per_gpu": B, "contexts": NC, "frames_per_launch": Bc,
                       "unique_frames": int(len(uniq)),
                       "frames": ("frames 0..%d of one synthetic stream (orbhip.synth.make_frames: a scene under a slowly varying "
                                  "similarity + shear warp), all distinct" % (len(uniq) - 1)) if len(uniq) == B else
                                 ("%d distinct frames tiled to the batch" % len(uniq)),
                       "inputs": "resident in HBM when the timed region starts, outputs stay there (the reference's own interface -- "
                                 "host cv::Mat in, host keypoints out, src/Frame.cc:594-596 -- is host_fed_frames_per_s)",
                       "match": args.match,
                       "parallelism": "frames sharded, 1 process per GPU, no per-frame collective"},
            # `bound` is what the counters say limits the kernel (vector-instruction issue); achieved / peak / frac are the HBM
            # figures the metric asks for, roofline_valu is the roofline of the resource that actually binds
            "roofline": {"bound": "hbm", "kernel": "k_fast", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_src, "traffic_stale": ctr_stale,
                         "traffic_stale_why": None if not ctr_stale else ("k_fast.hip has changed since the counter pass" if src_moved
                                                                            else "launch time differs by more than 5 % from the counter pass"),
                         "limited_by": "vector and LDS instruction issue (roofline_valu), not bytes: exact cv::FAST in this formulation "
                                       "takes at least 242 M vector wave-instructions per 1024 frames (tools/fast_bound.py, DESIGN "
                                       "section 4) = %.3f of the HBM roofline at full issue -- the ceiling; this run reaches %.2f of "
                                       "that ceiling (frozen in round 6: the remaining structural ideas cost more instructions than "
                                       "they remove, tools/fast_append_model.py)" % (FAST_CEILING_FRAC, achieved / HBM_PEAK_GBS / FAST_CEILING_FRAC),
                         "ceiling_frac": FAST_CEILING_FRAC, "frac_of_ceiling": round(achieved / HBM_PEAK_GBS / FAST_CEILING_FRAC, 3),
                         "algorithmic_bytes_per_launch": int(alg), "launch_ms": round(fast_ms, 4),
                         "note": "achieved = algorithmic bytes / launch time (HIP events on the launch stream, k_fast alone on the "
                                 "device); the kernel is bound by vector-instruction issue, see roofline_valu"
                                 + ("; launch_ms is measured while the other contexts' kernels share the GPU" if NC > 1 else "")},
            # one entry per streaming kernel of the step: SURVEY section 8d's algorithmic bytes per launch / the stage's HIP-event time of this
            # run (the instrumented pass behind the timed region) / 8 TB/s
            "rooflines": rooflines,
            "step_hbm_frac": round(step_bytes / (dt / max(args.steps, 1)) / 1e9 / HBM_PEAK_GBS, 4) if dt > 0 else None,
            "step_algorithmic_bytes": int(step_bytes),
            "roofline_valu": None if not valu or fast_ms <= 0 else {
                "kernel": "k_fast", "wave_insts": int(valu), "stale": ctr_stale, "achieved": round(valu / (fast_ms * 1e-3) / 1e9, 1),
                "issue_peak": round(issue_peak / 1e9, 1), "unit": "G wave-instructions/s",
                "frac": round(valu / (fast_ms * 1e-3) / issue_peak, 4),
                "issue_weight": VALU_ISSUE_WEIGHT,
                "frac_weighted": round(VALU_ISSUE_WEIGHT * valu / (fast_ms * 1e-3) / issue_peak, 4),
                "source": ctr.get("valu_source"),
                "note": "SQ_INSTS_VALU per launch (committed counter pass) / launch time of this run; peak = 1024 SIMDs x 1 "
                        "wave64 instruction per 4 cycles x %.2f GHz.  frac counts every instruction as 4 cycles; measured on the "
                        "device (profiles/r03/valu_rates.txt) 16 simple opcodes (add / sub / and / or / xor / mov / right shifts, f32 "
                        "add / mul / fma, v_bitop3) issue in 2, everything else (v_lerp_u8, v_pk_minimum3_f16, v_perm, compares, "
                        "selects, left shifts ...) in 4: frac_weighted = frac x issue_weight, the static share of the two classes in "
                        "the kernel's ISA (tools/update_traffic_meta.py), is the fraction of the issue cycles really taken"
                        % float(ctr.get("clock_ghz", 2.4))},
            "stage_ms": {"pyramid": round(float(stage[0]), 4), "fast": round(float(stage[1]), 4),
                         "quadtree": round(float(stage[2]), 4), "blur": round(float(stage[3]), 4),
                         "describe": round(float(stage[4]), 4), "last_match_kernel": round(float(stage[5]), 4)},
            "stage_ms_note": "fast: HIP events around the launch in every timed step; the other stages: %d further steps after the "
                             "timed region with every stage's events recorded (fast there: %.4f ms)" % (stage_pass, fast_instrumented),
            "rccl_ranks": rccl_ranks,
            "per_rank_frames_per_s": [round(B * args.steps / t, 1) for t in per_rank_dt],
            "vocabulary_broadcast": None if rccl_ranks is None else {
                "path": "orbhip_comm_init + orbhip_bcast_blob_device (RCCL ncclBroadcast on the library's own communicator)",
                "bytes": len(blob) if blob else None, "ms_incl_comm_init": round(bcast_ms, 2)},
            "keypoints_per_frame": round(float(counts.mean()), 1),
            "bow_matches_per_frame": None if nmatch is None else round(nmatch, 1),
            "brute_matches_le_TH_LOW_per_frame": None if nbrute is None else round(nbrute, 1),
        }
        if world == 1 and args.pipelined > 1 and B % args.pipelined == 0:
            out["pipelined"] = pipelined_throughput(args, d_img, blob if use_bow else None, local_rank, cap)
        if world == 1 and args.host_batch > 0:
            out["host_fed"] = host_fed_throughput(args, uniq, blob if use_bow else None, local_rank)
            out["config"]["host_fed_frames_per_s"] = out["host_fed"]["value"]
            out["config"]["host_fed_h2d_GBps"] = out["host_fed"]["h2d_GBps"]
    # ---- the GPU's outputs for the timed batch against the oracle, outside the timed region; every rank checks its own ----
    # Every distinct frame (by default: every frame of the batch) and its pair against the oracle, computed by oracle processes on
    # this rank's share of the host cores; a tiled batch: rows 0 .. U against the oracle (U = the pair across the first tile
    # boundary), every further row -- a tiled copy -- against its original on the device.
    U = int(len(uniq))
    nver = min(U + 1, Bc) if args.verify < 0 else max(0, min(args.verify, Bc))
    if rank == 0 and world == 1 and args.cpu_frames > 0:
        out["cpu_baseline"] = cpu_baseline(uniq, args.cpu_frames, args.match, blob)
        out["speedup_vs_cpu_1core"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(uniq[:64], args.cpu_frames, args.match, blob)
    kept = []
    t_ver = time.perf_counter()
    if nver:
        kept = oracle_outputs_parallel(frames[:nver], args.match, blob, host_workers)
    v_oracle = verify_against_oracle(kept, ctxs[0][1], cap, args.match) if kept else 0
    v_copies = verify_tiled_copies(ctxs[0][1], U, Bc, cap, args.match) if (NC == 1 and U < Bc and v_oracle >= min(U + 1, Bc)) else 0
    t_ver = time.perf_counter() - t_ver
    v_all = gather_objects(dist, (v_oracle, v_copies), world)
    if out is not None:
        out["verified_frames"] = v_oracle + v_copies
        out["verified_vs_oracle"] = v_oracle
        out["verified_copies_vs_original"] = v_copies
        out["verify_seconds"] = round(t_ver, 1)
        out["frame_generation_seconds"] = round(t_gen, 1)
        out["host_worker_processes"] = host_workers
        if world > 1:
            out["verified_frames_per_rank"] = [a + c for a, c in v_all]
        out["verified_against"] = "oracle/liborb_oracle.so (%d processes): keypoints (28-byte records), descriptors, " % host_workers + \
            {"bow": "SearchByBoW match12/match21/count", "brute": "brute-force best/second",
             "both": "SearchByBoW and brute-force results"}[args.match] + " of frames 0..%d of the timed batch, bit for bit" % max(v_oracle - 1, 0) + \
            ("; frames %d..%d are tiled copies, each compared on the device with its original among those (count, keypoints, "
             "descriptors, match results of its pair)" % (v_oracle, v_oracle + v_copies - 1) if v_copies else "")
        out["config"]["verified_frames"] = v_oracle + v_copies
    # ---- the step of rounds 1-5: the first 32 frames TILED to the batch -- what does tiling gain? ----
    if args.tiled_check and NC == 1 and world == 1 and U == B and B > 32:
        TU = 32
        d_img2 = d_img[:TU].repeat((B + TU - 1) // TU, 1, 1)[:B].contiguous()
        assert L.orbhip_set_stage_timing(ex0.handle, 1) == 0    # as in the timed region: the FAST launch's two events only
        step(d_img2)
        barrier()
        nt_steps = max(1, min(5, args.steps))
        t0 = time.perf_counter()
        for _ in range(nt_steps):
            step(d_img2)
            ms = (C.c_float * 6)()
            assert L.orbhip_get_stage_times(ex0.handle, ms) == 0     # (reading them waits for the step, as the timed loop does)
        barrier()
        nt = (time.perf_counter() - t0) / nt_steps
        assert L.orbhip_set_stage_timing(ex0.handle, 2) == 0
        step(d_img2)
        barrier()
        ms = (C.c_float * 6)()
        assert L.orbhip_get_stage_times(ex0.handle, ms) == 0
        nt_stage = {k: round(float(ms[i]), 4) for i, k in enumerate(("pyramid", "fast", "quadtree", "blur", "describe", "last_match_kernel"))}
        # rows 0 .. 31 against the oracle's records of the same frames, every further row against its original on the device (the
        # pair across the tile boundary, frame 31 -> frame 0, is compared between the copies only)
        bufs0 = ctxs[0][1]
        tiled_copies = 0
        if len(kept) >= TU:
            verify_against_oracle(kept[:TU], bufs0, cap, args.match)
            tiled_copies = verify_tiled_copies(bufs0, TU, B, cap, args.match)
        out["tiled_check"] = {"unique_frames": TU, "value": round(B / nt, 1), "unit": "frames/s", "ms_per_step": round(nt * 1e3, 3),
                              "steps": nt_steps, "ratio_to_headline": round((B / nt) / out["value"], 4), "stage_ms": nt_stage,
                              "copies_equal_their_originals": tiled_copies,
                              "note": "frames 0..31 of the headline's stream repeated to the batch (the headline of rounds 1-5): copies of a "
                                      "frame descend the same branches of the vocabulary and find the same corners, so caches and "
                                      "branch divergence favour it; same context, same buffers"}
        out["config"]["tiled_32_frames_per_s"] = out["tiled_check"]["value"]
        del d_img2
        step()                                                   # the headline batch's results back into the buffers
        barrier()
    # ---- N > 1: the mode that can fail to scale -- every rank fed from host memory at once ----
    if dist is not None and args.host_batch > 0:
        barrier()
        hf = host_fed_throughput(args, uniq, blob if use_bow else None, local_rank)
        hf_all = gather_objects(dist, (hf, numa), world)
        if out is not None:
            topo = D.host_topology()
            topo.pop("_cpus", None)
            secs = [h["seconds"] for h, _ in hf_all]
            frames_all = sum(h["batches"] * h["frames_per_batch"] for h, _ in hf_all)
            out["host_fed"] = {
                "value": round(frames_all / max(secs), 1), "unit": "frames/s", "ranks": world,
                "h2d_GBps_total": round(frames_all / max(secs) * W * H / 1e9, 2),
                "per_rank": [{"rank": i, "frames_per_s": h["value"], "h2d_GBps": h["h2d_GBps"], "d2h_GBps": h["d2h_GBps"],
                              "seconds": h["seconds"], "numa": n} for i, (h, n) in enumerate(hf_all)],
                "frames_per_batch": hf["frames_per_batch"], "ring_depth": hf["ring_depth"], "pinned": True, "host": topo,
                "note": "every rank runs its own orbhip_pipe_* ring at the same time (barrier in front): pinned frame buffers "
                        "allocated after the rank was bound to the CPUs of its GPU's NUMA node, copy-in / kernels / copy-out "
                        "overlapped; value = all ranks' frames / the slowest rank's time.  This, not the device-resident "
                        "headline, is what can bend a 1 -> N curve: N x ~50 GB/s out of one host's memory"}
    # ---- N >= 4: BASELINE config 4 proper, one EuRoC stream per rank ----
    want_streams = args.streams_config == 1 or (args.streams_config < 0 and world >= 4)
    if dist is not None and want_streams and use_bow:
        cfg4 = config_streams_ranks(local_rank, blob, rank, world, dist, barrier)
        if out is not None:
            out.setdefault("configs", {})["4_euroc_streams_one_per_rank"] = cfg4
    for ex, _ in ctxs:
        ex.close()
    del ctxs, d_img
    if out is not None and world == 1 and args.configs:
        torch.cuda.empty_cache()
        if blob is None:
            blob = D.make_synthetic_vocabulary(4242, VOC_K, VOC_L)
        out.setdefault("configs", {}).update(secondary_configs(local_rank, blob))
        out["roofline_mfma"] = out["configs"]["5_tum_4000feat_1M_query"].get("roofline_mfma")
    if out is not None and world == 1 and args.content:
        torch.cuda.empty_cache()
        if blob is None:
            blob = D.make_synthetic_vocabulary(4242, VOC_K, VOC_L)
        out["content"] = content_classes(local_rank, blob, verify=args.verify != 0)
    if out is not None and world == 1 and args.batch_sweep:
        out["batch_sweep"] = batch_sweep(args)
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
