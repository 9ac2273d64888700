"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend nccl = RCCL on the GPU
box, gloo in the CPU tests).  The ORB front end shards by independent units -- frames of one
stream, or whole streams -- so there is NO per-frame collective; the only exchange steps are the
one-off broadcast of the ORB vocabulary and (brute-force relocalisation) a min-merge of per-shard
(best, index, second) triples.  SURVEY.md section 8e.
"""
import os
import struct

import numpy as np

# ---- ORB vocabulary blob: the reference's binary format --------------------------------------
# Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1727-1751 (saveToBinaryFile) / :1680-1721 (load):
#   u32 nb_nodes (= m_nodes.size(), root included), u32 size_node (= 4 + 32 + 4 + 1 = 41),
#   i32 k, i32 L, i32 scoring, i32 weighting, then for node id 1..nb_nodes-1:
#   i32 parent, u8 descriptor[32], f32 weight, u8 is_leaf.
VOC_HEADER = struct.Struct("<IIiiii")
VOC_NODE_SIZE = 41
VOC_NODE_DTYPE = np.dtype([("parent", "<i4"), ("desc", "u1", 32), ("weight", "<f4"), ("leaf", "u1")])
assert VOC_NODE_DTYPE.itemsize == VOC_NODE_SIZE


def pack_vocabulary(k, L, scoring, weighting, parent, desc, weight, leaf):
    """Arrays over node ids 1..n (root excluded) -> the binary blob of saveToBinaryFile."""
    n = len(parent)
    nodes = np.zeros(n, VOC_NODE_DTYPE)
    nodes["parent"] = parent
    nodes["desc"] = desc
    nodes["weight"] = weight
    nodes["leaf"] = leaf
    return VOC_HEADER.pack(n + 1, VOC_NODE_SIZE, k, L, scoring, weighting) + nodes.tobytes()


def unpack_vocabulary(blob):
    """Inverse of pack_vocabulary: dict(k, L, scoring, weighting, nodes=structured array 1..n)."""
    blob = bytes(blob)
    nb, size_node, k, L, scoring, weighting = VOC_HEADER.unpack_from(blob, 0)
    if size_node != VOC_NODE_SIZE:
        raise ValueError("unexpected vocabulary node size %d" % size_node)
    n = (len(blob) - VOC_HEADER.size) // VOC_NODE_SIZE
    if n != nb - 1:
        raise ValueError("vocabulary blob truncated: %d nodes in header, %d present" % (nb - 1, n))
    nodes = np.frombuffer(blob, VOC_NODE_DTYPE, n, VOC_HEADER.size)
    return {"k": k, "L": L, "scoring": scoring, "weighting": weighting, "nodes": nodes}


def vocabulary_to_text(blob):
    """The text form of a vocabulary (ref: saveToTextFile, TemplatedVocabulary.h:1651-1672): "k L  scoring weighting"
    (the reference writes two spaces there), then per node "parent leaf d0 .. d31 weight"; the weight goes through an
    ostream at its default precision (6 significant digits, %g)."""
    v = unpack_vocabulary(blob)
    lines = ["%d %d  %d %d" % (v["k"], v["L"], v["scoring"], v["weighting"])]
    for nd in v["nodes"]:
        lines.append("%d %d %s %s" % (nd["parent"], 1 if nd["leaf"] else 0, " ".join(str(int(b)) for b in nd["desc"]) + " ",
                                      "%g" % float(nd["weight"])))
    return ("\n".join(lines) + "\n").encode()


def make_synthetic_vocabulary(seed, k=10, L=3):
    """A complete k-ary tree of depth L with random 256-bit node descriptors, nodes numbered level by
    level (the stock ORBvoc is k=10, L=6, ~1.08 M nodes, ~44 MB; the file is not in the reference
    mirror)."""
    rng = np.random.default_rng(seed)
    parents, leaves = [], []
    start_prev, n_prev, next_id = 0, 1, 1          # previous level = the root
    for depth in range(1, L + 1):
        n = n_prev * k
        parents.append(start_prev + np.arange(n, dtype=np.int64) // k)
        leaves.append(np.full(n, 1 if depth == L else 0, np.uint8))
        start_prev, n_prev, next_id = next_id, n, next_id + n
    parent = np.concatenate(parents).astype(np.int32)
    leaf = np.concatenate(leaves)
    n = len(parent)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    weight = rng.random(n).astype(np.float32)
    return pack_vocabulary(k, L, 0, 0, parent, desc, weight, leaf)


# ---- sharding -------------------------------------------------------------------------------
def shard_frames(n_frames, rank, world):
    """Contiguous block of frames for `rank` (single-stream mode): sizes differ by at most one,
    every frame belongs to exactly one rank."""
    base, extra = divmod(n_frames, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def assign_streams(lengths, world):
    """Streams mode (BASELINE config 4: one sequence per GPU): longest-first greedy assignment
    of whole streams to ranks.  Returns a list of stream-index lists, one per rank."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    load = [0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda j: (load[j], j))
        out[r].append(i)
        load[r] += lengths[i]
    return out


# ---- exchange steps -------------------------------------------------------------------------
def broadcast_blob(blob, src=0, device=None):
    """Broadcast a byte blob (the vocabulary) from `src` to every rank.  Two collectives: the
    length, then the bytes.  With backend nccl this is RCCL over xGMI: the root has a direct
    link to each peer, one ~44 MB send per link."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    n = torch.tensor([len(blob) if rank == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(n, src=src)
    if rank == src:
        buf = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    else:
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    dist.broadcast(buf, src=src)
    return buf


def merge_knn2_shards(best_idx, best_d, second_d, shard_offsets):
    """Min-merge of per-shard brute-force results in shard order (lower database rows first), with
    the reference's tie rule (strict '<': the lowest global index wins).  Inputs are lists of
    arrays, one per shard, indices local to the shard."""
    bi = np.full_like(best_idx[0], -1)
    bd = np.full_like(best_d[0], 256)
    sd = np.full_like(second_d[0], 256)
    for li, ld, ls, off in zip(best_idx, best_d, second_d, shard_offsets):
        gi = np.where(li >= 0, li + off, -1)
        better = ld < bd
        sd = np.where(better, np.minimum(bd, ls), np.minimum(sd, ld))
        bi = np.where(better, gi, bi)
        bd = np.where(better, ld, bd)
    return bi, bd, sd


def allgather_knn2(best_idx, best_d, second_d, shard_offset):
    """The one exchange step of sharded brute force with host arrays (gloo / CPU tests): all-gather Q x 3 int32
    and merge on the host.  The GPU path is allgather_knn2_device."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    mine = torch.from_numpy(np.stack([best_idx, best_d, second_d,
                                      np.full_like(best_idx, shard_offset)]).astype(np.int32)).to(dev)
    outs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(outs, mine)
    arrs = [o.cpu().numpy() for o in outs]
    return merge_knn2_shards([a[0] for a in arrs], [a[1] for a in arrs], [a[2] for a in arrs],
                             [int(a[3][0]) if a.shape[1] else 0 for a in arrs])


def allgather_knn2_device(ex, best_idx, best_d, second_d, shard_offset):
    """The same exchange step with everything resident on the GPU: per-rank results are torch int32 CUDA tensors
    [nq] (outputs of orbhip_hamming_knn2_device on this rank's database rows); one all-gather of 3 * nq + 1 int32 per
    rank (RCCL), then the min-merge kernel of liborbhip (orbhip_knn2_merge_device) -- nothing returns to the host.
    Works without an initialised process group (world 1).  Returns three CUDA tensors with global indices."""
    import torch
    import torch.distributed as dist
    nq = int(best_idx.numel())
    dev = best_idx.device
    part = torch.cat([best_idx.to(torch.int32).reshape(-1), best_d.to(torch.int32).reshape(-1),
                      second_d.to(torch.int32).reshape(-1),
                      torch.tensor([int(shard_offset)], dtype=torch.int32, device=dev)])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        world = dist.get_world_size()
        allp = torch.empty((world, part.numel()), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(allp, part)
    else:
        world, allp = 1, part.reshape(1, -1).contiguous()
    out = [torch.empty(nq, dtype=torch.int32, device=dev) for _ in range(3)]
    torch.cuda.current_stream().synchronize()      # torch's stream -> the context's stream
    rc = ex._L.orbhip_knn2_merge_device(ex.handle, allp.data_ptr(), world, nq, out[0].data_ptr(), out[1].data_ptr(),
                                        out[2].data_ptr())
    if rc != 0:
        raise RuntimeError("orbhip_knn2_merge_device failed (%d)" % rc)
    ex.sync()
    return out


# ---- host placement ---------------------------------------------------------------------------
def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.extend(range(int(a), int(b or a) + 1))
    return cpus


def host_topology():
    """What the host looks like to a rank: logical CPUs, CPUs this process may run on, NUMA nodes with their CPU counts
    (sysfs; an empty node list where the kernel exposes none)."""
    nodes = {}
    base = "/sys/devices/system/node"
    try:
        for name in sorted(os.listdir(base)):
            if name.startswith("node") and name[4:].isdigit():
                with open(os.path.join(base, name, "cpulist")) as fh:
                    nodes[int(name[4:])] = _parse_cpulist(fh.read())
    except OSError:
        pass
    return {"nproc": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "numa_nodes": len(nodes),
            "numa_node_cpus": {str(k): len(v) for k, v in sorted(nodes.items())}, "_cpus": nodes}


def gpu_numa_node(pci_domain, pci_bus, pci_device):
    """NUMA node of the GPU at this PCI address (sysfs numa_node; None when unknown or -1)."""
    for fn in range(8):
        path = "/sys/bus/pci/devices/%04x:%02x:%02x.%d/numa_node" % (pci_domain, pci_bus, pci_device, fn)
        try:
            with open(path) as fh:
                n = int(fh.read().strip())
            return n if n >= 0 else None
        except (OSError, ValueError):
            continue
    return None


def numa_bind(local_rank, bind=True):
    """Run this rank -- every thread it has at this point, and through inheritance every later one -- on the CPUs of its GPU's NUMA
    node (sched_setaffinity per thread id of /proc/self/task, before any page-locked ring is allocated: pages
    are placed on the node of the thread that first touches them, and the threads that fill the rings then run next to
    them).  Eight ranks each pull ~50 GB/s out of host memory over their own PCIe link; a ring on the other socket crosses
    the inter-socket fabric twice per frame.  Best effort: returns what it found and what it did, never raises."""
    info = {"gpu_numa_node": None, "bound": False, "cpus": len(os.sched_getaffinity(0))}
    try:
        import torch
        pr = torch.cuda.get_device_properties(local_rank)
        node = gpu_numa_node(int(pr.pci_domain_id), int(pr.pci_bus_id), int(pr.pci_device_id))
        info["pci"] = "%04x:%02x:%02x" % (int(pr.pci_domain_id), int(pr.pci_bus_id), int(pr.pci_device_id))
        info["gpu_numa_node"] = node
        topo = host_topology()
        if bind and node is not None and topo["numa_nodes"] > 1:
            allowed = sorted(set(topo["_cpus"].get(node, [])) & os.sched_getaffinity(0))
            if allowed:
                # every thread the process has by now (the HIP runtime's, RCCL's and OpenMP's were started by the set_device /
                # init_process_group calls in front of this one and would keep the old mask; threads created later inherit the
                # caller's) -- ADVICE r05
                tids = [0]
                try:
                    tids = [int(t) for t in os.listdir("/proc/self/task")]
                except OSError:
                    pass
                done = 0
                for t in tids:
                    try:
                        os.sched_setaffinity(t, allowed)
                        done += 1
                    except OSError:                  # a thread that ended meanwhile
                        pass
                os.sched_setaffinity(0, allowed)
                info["bound"] = True
                info["cpus"] = len(allowed)
                info["threads_bound"] = done
    except Exception as exc:                      # noqa: BLE001 -- placement is an optimisation, never a reason to fail
        info["error"] = str(exc)[:200]
    return info


# ---- launching one process per GPU ------------------------------------------------------------
def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_env(rank, world, port, base=None):
    """Environment of rank `rank` of a one-node job (what torch.distributed.run would set)."""
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on these hosts (RCCL needs it)
    return env


def launch_ranks(argv, world, timeout=None, env=None, local_ranks=None):
    """Start `world` fresh child processes of `argv` (one per GPU), rank r with rank_env(r, ...), and wait for them.
    The caller must not have touched the GPU: a process that has initialised HIP must never be replaced or forked
    into another program on this pool, so the launcher runs BEFORE anything imports torch.cuda state.  Rank 0's
    stdout is returned (its last line is the JSON line); the other ranks' stdout is discarded, stderr is inherited.
    All ranks are polled together: the first rank that exits non-zero ends the job at once (the others -- exactly the
    processes started here -- are killed) instead of leaving rank 0 in the rendezvous until torch's own timeout.
    `local_ranks` (optional list) overrides LOCAL_RANK per rank, e.g. [0, 0] runs two ranks on device 0.
    Returns (return code: 0, the first failing rank's, or 124 on timeout; rank-0 stdout)."""
    import subprocess
    import threading
    import time
    port = free_port()
    procs = []
    for r in range(world):
        e = rank_env(r, world, port, env)
        if local_ranks is not None:
            e["LOCAL_RANK"] = str(local_ranks[r])
        procs.append(subprocess.Popen(argv, env=e, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    chunks = []

    def drain():                       # rank 0's pipe must be emptied while it runs, or it blocks on a full pipe
        for block in iter(lambda: procs[0].stdout.read(65536), b""):
            chunks.append(block)

    t = threading.Thread(target=drain, daemon=True)
    t.start()
    rc = 0
    deadline = None if timeout is None else time.monotonic() + timeout
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                rc = abs(bad[0])
                break
            if all(c == 0 for c in codes):
                break
            if deadline is not None and time.monotonic() > deadline:
                rc = 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()          # exactly the processes started here
        for p in procs:
            p.wait()
        t.join(timeout=10)
    return rc, b"".join(chunks).decode(errors="replace")
