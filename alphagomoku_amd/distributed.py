"""Multi-GPU plumbing of the benchmark / launcher: one process per GPU, each with its own independent game pool.

There is NO collective on the data path (games never talk to each other, exactly like the reference's one-thread-per-device
GeneratorManager, src/selfplay/GeneratorManager.cpp:146-152).  torch.distributed is used only for the start/stop barrier and to
combine the per-rank measurements: wall time = MAX over ranks, work counters = SUM over ranks."""
import os
import sys


def env_ranks():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None, device=None):
    """returns the torch.distributed module (initialised) or None for a single process"""
    rank, local_rank, world = env_ranks()
    if world <= 1 and not os.environ.get("AGX_DIST_FORCE"):  # AGX_DIST_FORCE=1: exercise the rendezvous / barrier path with one rank
        return None
    import torch
    import torch.distributed as dist
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kwargs = {}
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        kwargs["device_id"] = torch.device("cuda", local_rank)
    # gloo announces its connections on stdout ("[Gloo] Rank 0 is connected to ..."); bench.py's stdout is ONE JSON line, so the process's
    # stdout goes to stderr while the group is being set up
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        dist.init_process_group(backend=backend, **kwargs)
        dist.barrier()  # (gloo connects lazily: the first collective makes the announcements)
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)
    return dist


def combine(dist, elapsed_seconds, counters):
    """elapsed -> MAX over ranks, counters (list of numbers) -> SUM over ranks; identity for a single process"""
    if dist is None:
        return float(elapsed_seconds), [float(c) for c in counters]
    import torch
    device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([elapsed_seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([float(x) for x in counters], dtype=torch.float64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), [float(x) for x in c.tolist()]


def gather(dist, values):
    """every rank's list of numbers, as a list of lists indexed by rank (identity wrapper for a single process)"""
    if dist is None:
        return [[float(v) for v in values]]
    import torch
    device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    mine = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [[float(x) for x in t.tolist()] for t in out]


def rank_seed_base(rank):
    """disjoint opening seeds per rank so that the ranks play different games"""
    return rank * 1000003


def _kfd_gpu_pci_addresses(sysfs_root):
    """PCI addresses ("dddd:bb:dd.f") of the GPUs in KFD topology order — the order in which the HIP runtime numbers its devices (every
    /sys/class/kfd/kfd/topology/nodes/<n>/properties with simd_count > 0 is a GPU; `domain` and `location_id` = bus << 8 | device << 3 | function
    give its PCI function).  None when the topology cannot be read."""
    base = os.path.join(sysfs_root, "sys/class/kfd/kfd/topology/nodes")
    try:
        nodes = sorted((int(n) for n in os.listdir(base) if n.isdigit()))
    except OSError:
        return None
    out = []
    for n in nodes:
        try:
            props = dict(line.split()[:2] for line in open(os.path.join(base, str(n), "properties")) if len(line.split()) >= 2)
        except OSError:
            continue   # (a node this user may not read: not one of its devices)
        if int(props.get("simd_count", "0")) <= 0:
            continue   # a CPU node
        loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
        out.append("%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 0x7))
    return out or None


def _id_list(var):
    """the integer list of a *_VISIBLE_DEVICES variable: None when unset, False when it cannot be resolved here (UUID form, garbage)"""
    spec = os.environ.get(var, "").strip()
    if not spec:
        return None
    try:
        return [int(x) for x in spec.split(",") if x.strip() != ""]
    except ValueError:
        return False


def _visible_index(local_device):
    """The runtime numbers only the VISIBLE devices, and the filters COMPOSE: ROCR_VISIBLE_DEVICES selects from the topology's GPUs, then
    HIP_VISIBLE_DEVICES (or CUDA_VISIBLE_DEVICES, which HIP honours as well) selects from what ROCR left — ROCR "2,3" + HIP "1" makes device 0
    the topology's GPU 3.  None whenever a case cannot be resolved (UUID forms, both HIP_ and CUDA_ set to different lists, an index out of
    range): the rank then stays unpinned instead of being pinned to another GPU's NUMA node."""
    rocr, hip, cuda = _id_list("ROCR_VISIBLE_DEVICES"), _id_list("HIP_VISIBLE_DEVICES"), _id_list("CUDA_VISIBLE_DEVICES")
    if rocr is False or hip is False or cuda is False:
        return None
    if hip is not None and cuda is not None and hip != cuda:
        return None
    inner = hip if hip is not None else cuda
    index = local_device
    if inner is not None:
        if index >= len(inner):
            return None
        index = inner[index]
    if rocr is not None:
        if index < 0 or index >= len(rocr):
            return None
        index = rocr[index]
    return index if index >= 0 else None


def pin_to_gpu_numa_node(local_device, sysfs_root="/"):
    """Binds this process's host threads to the CPUs of the NUMA node its GPU hangs off (one generator thread per device in the reference,
    GeneratorManager.cpp:146-152; SURVEY 8e: the host side of a rank is a launch loop, it must not wander to the other socket).  Best effort
    from sysfs: the GPU is the local_device-th VISIBLE one in KFD topology order (the HIP runtime's numbering; fallback: the AMD display /
    processing-accelerator PCI functions in address order), its PCI function's numa_node, that node's cpulist intersected with the current
    affinity.  Returns (numa_node, cpus pinned to) or (None, 0) when the topology cannot be read."""
    import glob
    try:
        index = _visible_index(local_device)
        if index is None:
            return None, 0
        addresses = _kfd_gpu_pci_addresses(sysfs_root)
        if addresses is not None:
            if index >= len(addresses):
                return None, 0
            device_dir = os.path.join(sysfs_root, "sys/bus/pci/devices", addresses[index])
        else:
            gpus = []
            for dev in glob.glob(os.path.join(sysfs_root, "sys/bus/pci/devices/*")):
                try:
                    vendor = open(os.path.join(dev, "vendor")).read().strip()
                    cls = open(os.path.join(dev, "class")).read().strip()
                except OSError:
                    continue
                if vendor == "0x1002" and (cls.startswith("0x03") or cls.startswith("0x12")):
                    gpus.append(dev)
            gpus.sort()
            if index >= len(gpus):
                return None, 0
            device_dir = gpus[index]
        node = int(open(os.path.join(device_dir, "numa_node")).read().strip())
        if node < 0:
            return None, 0
        cpus = set()
        for part in open(os.path.join(sysfs_root, "sys/devices/system/node/node%d/cpulist" % node)).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = cpus & set(os.sched_getaffinity(0))
        if not allowed:
            return node, 0
        os.sched_setaffinity(0, allowed)
        return node, len(allowed)
    except (OSError, ValueError, AttributeError):
        return None, 0
