"""Streams mode (BASELINE.json config 4): independent image sequences, one per GPU, each on its own extractor
context; the ORB vocabulary reaches every rank by ONE broadcast at start-up, nothing is exchanged per frame
(SURVEY.md section 8e).

A stream is processed like Tracking processes a sequence (src/Tracking.cc:816-822, 1881-1885): every frame goes through
ORBextractor::operator(), Frame::ComputeBoW (vocabulary transform, levelsup 4) and ORBmatcher(0.7, true)::SearchByBoW
against its predecessor.  Frames are taken in batches of B that overlap by one frame, so that every consecutive pair of
the sequence -- also the pair that straddles two batches -- is matched exactly once by the batched kernel
(orbhip_search_by_bow_seq_device, lag 1).

Run as a rank of a one-node job:  python -m orbhip.streams --lengths 29,17,23,30 ...   (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* from the environment, as torch.distributed.run sets them); rank 0 prints one JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

from . import distributed as D

# the four sequences BASELINE.json names and their lengths (EuRoC time-stamp files of the reference,
# Examples/Monocular/EuRoC_TimeStamps/{V101,V102,V201,MH02}.txt)
EUROC_STREAMS = (("V1_01", 2912), ("V1_02", 1710), ("V2_01", 2280), ("MH_02", 3040))
W, H, NFEAT = 752, 480, 1000          # EuRoC cam0 (Examples/Monocular/EuRoC.yaml)
LEVELSUP, NNRATIO = 4, 0.7


def batch_plan(n_frames, B):
    """Batches (first frame, frame count) covering frames 0..n-1 with one frame of overlap: consecutive pairs (t - 1, t)
    each fall inside exactly one batch."""
    if n_frames <= 0:
        return []
    if B < 2:
        raise ValueError("streams need batches of at least two frames")
    out, s = [], 0
    while True:
        nb = min(B, n_frames - s)
        out.append((s, nb))
        if s + nb >= n_frames:
            return out
        s += B - 1


def pair_location(t, B):
    """(batch index, frame slot inside the batch) that holds frame t together with its predecessor t - 1 (t >= 1)."""
    k = (t - 1) // (B - 1)
    return k, t - k * (B - 1)


class StreamRunner:
    """One sequence on one extractor context of `device`.  The stream's frames are `ring[t % U]` of a device-resident set
    of U distinct frames (there are no datasets on either box; frame content is synthetic, orbhip.synth)."""

    def __init__(self, device, B, uniq_frames, blob=None, d_blob=None, w=W, h=H, nfeat=NFEAT):
        import torch
        from .extractor import ORBextractor
        from .vocabulary import ORBVocabulary
        self.B, self.w, self.h = B, w, h
        self.ex = ORBextractor(nfeat, 1.2, 8, 20, 7, max_w=w, max_h=h, max_batch=B, device=device)
        voc = ORBVocabulary(self.ex)
        if d_blob is not None:
            voc.loadFromDeviceBlob(d_blob.data_ptr(), d_blob.numel())
        else:
            voc.loadFromBinaryBlob(blob)
        self.U = len(uniq_frames)
        # any window [s % U, s % U + B) of the cyclic stream is a contiguous slice of `ring`
        reps = 1 + (B + self.U - 1) // self.U
        self.ring = torch.from_numpy(np.ascontiguousarray(np.concatenate([uniq_frames] * reps))).cuda(device)
        cap = self.cap = self.ex.cap
        dev = dict(device=torch.device("cuda", device))
        i32 = dict(dtype=torch.int32, **dev)
        self.kps = torch.empty((B, cap, 7), **i32)
        self.desc = torch.empty((B, cap, 32), dtype=torch.uint8, **dev)
        self.cnt = torch.zeros(B, **i32)
        self.word, self.node, self.m12, self.m21 = (torch.empty((B, cap), **i32) for _ in range(4))
        self.wt = torch.empty((B, cap), dtype=torch.float32, **dev)
        self.nm = torch.zeros(B, **i32)

    def run(self, n_frames, sample=()):
        """Process frames 0..n_frames-1.  `sample`: frame indices whose outputs (keypoints, descriptors, and for t >= 1
        the SearchByBoW result against frame t - 1) are copied back: {t: dict}.  Returns (samples, matches summed)."""
        ex, L, B, cap, w, h = self.ex, self.ex._L, self.B, self.cap, self.w, self.h
        want = {}
        for t in sample:
            if t == 0:
                want.setdefault(0, []).append((0, 0))
            else:
                k, b = pair_location(t, B)
                want.setdefault(k, []).append((t, b))
        got, nmatch = {}, 0
        for k, (s, nb) in enumerate(batch_plan(n_frames, B)):
            img = self.ring[s % self.U:s % self.U + nb]
            ex.extract_batch_device(img.data_ptr(), nb, w, h, w, h * w, self.kps.data_ptr(), self.desc.data_ptr(), cap,
                                    self.cnt.data_ptr())
            rc = L.orbhip_vocab_transform_device(ex.handle, self.desc.data_ptr(), nb * cap, LEVELSUP, self.word.data_ptr(),
                                                 self.wt.data_ptr(), self.node.data_ptr())
            assert rc == 0, rc
            rc = L.orbhip_search_by_bow_seq_device(ex.handle, self.desc.data_ptr(), self.kps.data_ptr(), self.cnt.data_ptr(),
                                                   self.node.data_ptr(), self.wt.data_ptr(), None, cap, nb, 1, 0,
                                                   C.c_float(NNRATIO), 1, self.m12.data_ptr(), self.m21.data_ptr(),
                                                   self.nm.data_ptr())
            assert rc == 0, rc
            if k in want:
                ex.sync()
                cnt = self.cnt.cpu().numpy()
                for t, b in want[k]:
                    n = int(cnt[b])
                    rec = {"n": n, "kps": self.kps[b, :n].cpu().numpy().tobytes(), "desc": self.desc[b, :n].cpu().numpy()}
                    if t >= 1:
                        rec.update(n_prev=int(cnt[b - 1]), nm=int(self.nm[b].item()),
                                   m12=self.m12[b, :int(cnt[b - 1])].cpu().numpy(), m21=self.m21[b, :n].cpu().numpy())
                    got[t] = rec
        ex.sync()
        return got

    def frame(self, t):
        """Host copy of frame t of the stream (for the oracle)."""
        return self.ring[t % self.U].cpu().numpy()

    def close(self):
        self.ex.close()


def default_samples(n, B):
    """Frames checked against the oracle: the first, the first pair, the pair straddling the first batch boundary, the last."""
    s = {0, 1, n - 1}
    if n > B:
        s.update({B - 1, B})
    return sorted(t for t in s if 0 <= t < n)


def comm_broadcast_vocabulary(ex, blob, rank, world, exchange_uid):
    """The vocabulary broadcast through the C ABI's RCCL path (orbhip_comm_* + orbhip_bcast_blob_device): unique id from
    rank 0 (distributed by `exchange_uid`, a callable bytes -> bytes that returns rank 0's value on every rank), one
    ncclBroadcast of the length and one of the blob into a device buffer.  Returns (device tensor, ranks reported by the
    communicator itself)."""
    import torch
    L = ex._L
    uid = (C.c_uint8 * 128)()
    if rank == 0:
        assert L.orbhip_comm_unique_id(uid) == 0
    uid = (C.c_uint8 * 128).from_buffer_copy(exchange_uid(bytes(uid)))
    rc = L.orbhip_comm_init(ex.handle, rank, world, uid)
    if rc != 0:
        from .capi import last_error
        raise RuntimeError("orbhip_comm_init: " + last_error(ex.handle))
    dev = torch.device("cuda", torch.cuda.current_device())
    n = torch.tensor([len(blob) if rank == 0 else 0], dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    assert L.orbhip_bcast_blob_device(ex.handle, n.data_ptr(), 8, 0) == 0
    ex.sync()
    nbytes = int(n.item())
    if rank == 0:
        buf = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    else:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    assert L.orbhip_bcast_blob_device(ex.handle, buf.data_ptr(), nbytes, 0) == 0
    ex.sync()
    r, nr = C.c_int(), C.c_int()
    assert L.orbhip_comm_info(ex.handle, C.byref(r), C.byref(nr)) == 0
    assert r.value == rank
    return buf, nr.value


def run_rank(args, verifier=None):
    """One rank of the streams job.  `verifier(runner, samples, blob) -> frames checked` is supplied by the tests (it runs the
    CPU oracle, which is test infrastructure and never imported from the product package)."""
    import torch
    import torch.distributed as dist
    from . import synth
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if not torch.cuda.is_available():
        raise SystemExit("orbhip.streams needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    backend = args.backend
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29519")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    lengths = [int(x) for x in args.lengths.split(",")]
    mine = D.assign_streams(lengths, world)[rank]
    blob0 = D.make_synthetic_vocabulary(4242, args.voc_k, args.voc_l) if rank == 0 else b""

    from .extractor import ORBextractor
    rccl_ranks, d_blob, blob, bcast = None, None, None, None
    t0 = time.perf_counter()
    if backend == "nccl":
        # the C ABI's own RCCL communicator (also for one rank); the 128-byte unique id travels through torch's group
        def exchange(u):
            if world == 1:
                return u
            t = torch.frombuffer(bytearray(u), dtype=torch.uint8).cuda()
            dist.broadcast(t, src=0)
            return bytes(t.cpu().numpy().tobytes())
        cx = ORBextractor(NFEAT, 1.2, 8, 20, 7, max_w=W, max_h=H, max_batch=1, device=local)
        d_blob, rccl_ranks = comm_broadcast_vocabulary(cx, blob0, rank, world, exchange)
        blob = bytes(d_blob.cpu().numpy().tobytes()) if verifier else None
        bcast = "orbhip_bcast_blob_device (RCCL ncclBroadcast, communicator of %d ranks)" % rccl_ranks
    else:
        # ranks that share a device (RCCL refuses duplicate devices) or CPU-side rendezvous: the blob over gloo
        buf = D.broadcast_blob(blob0, src=0, device="cpu") if world > 1 else None
        blob = blob0 if world == 1 else bytes(buf.numpy().tobytes())
        cx = None
        bcast = "torch.distributed gloo broadcast (host memory)"
    bcast_s = time.perf_counter() - t0

    runners = []
    for si in mine:
        uniq = synth.make_frames(2000 + si, W, H, args.unique)
        runners.append((si, StreamRunner(local, args.batch, uniq, blob=blob, d_blob=d_blob)))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
    for si, r in runners:                              # warm-up: one short batch per stream
        r.run(min(lengths[si], args.batch))
    barrier()
    t0 = time.perf_counter()
    got = {}
    for si, r in runners:
        got[si] = r.run(lengths[si], default_samples(lengths[si], args.batch) if verifier else ())
    barrier()
    dt = time.perf_counter() - t0
    verified = 0
    if verifier:
        for si, r in runners:
            verified += verifier(r, got[si], blob)
    my_frames = sum(lengths[si] for si in mine)
    rec = torch.tensor([dt, float(my_frames), float(verified), float(len(mine))], dtype=torch.float64)
    if world > 1:
        if backend == "nccl":
            rec = rec.cuda()
        allr = [torch.empty_like(rec) for _ in range(world)]
        dist.all_gather(allr, rec)
        allr = [a.cpu().numpy() for a in allr]
    else:
        allr = [rec.numpy()]
    for _, r in runners:
        r.close()
    if cx is not None:
        cx._L.orbhip_comm_destroy(cx.handle)
        cx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        tmax = max(float(a[0]) for a in allr)
        total = sum(int(a[1]) for a in allr)
        out = {"mode": "streams", "metric": "ORB extract+match frames/sec, 1 sequence per context (BASELINE config 4)",
               "value": round(total / tmax, 1), "unit": "frames/s", "n_ranks": world, "backend": backend,
               "rccl_ranks": rccl_ranks, "vocabulary_broadcast": bcast, "vocabulary_bytes": len(blob0),
               "vocabulary_broadcast_s": round(bcast_s, 4),
               "streams": [{"name": EUROC_STREAMS[i % 4][0], "frames": lengths[i]} for i in range(len(lengths))],
               "assignment": D.assign_streams(lengths, world), "frames_per_batch": args.batch, "frame_size": [W, H],
               "per_rank": [{"rank": i, "streams": int(a[3]), "frames": int(a[1]), "seconds": round(float(a[0]), 4),
                             "frames_per_s": round(float(a[1]) / float(a[0]), 1) if a[0] > 0 else None,
                             "verified_frames": int(a[2])} for i, a in enumerate(allr)],
               "verified_frames": sum(int(a[2]) for a in allr)}
        C.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


def main(argv=None, verifier=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--lengths", default=",".join(str(n) for _, n in EUROC_STREAMS), help="frames per stream")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--unique", type=int, default=16, help="distinct synthetic frames per stream (cycled)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl")
    ap.add_argument("--voc-k", type=int, default=10)
    ap.add_argument("--voc-l", type=int, default=6)
    run_rank(ap.parse_args(argv), verifier)


if __name__ == "__main__":
    main()
