"""Multi-GPU story of the hot path: stereo pairs are independent units, so the
batch is split contiguously across one-process-per-GPU ranks (same rule as the reference's
InferenceSampler, nmrf/utils/evaluation.py:61-69); weights are replicated; the ONLY collective is the
gather of the finished disparity maps (RCCL all-gather over xGMI on GPUs, gloo on CPU in the tests).
"""
import os

import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous [begin, end) of `total` units for `rank`; the first total%world ranks get one extra."""
    base, extra = divmod(total, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def gather_disparity(disp_local, total=None, group=None):
    """disp_local [b_local,H,W] on every rank -> [sum b_local, H, W] on every rank, in rank order.
    Equal shards use one all_gather_into_tensor (a single RCCL ring/direct all-gather); ragged shards
    (total % world != 0) fall back to all_gather with padding to the largest shard."""
    if not (dist.is_available() and dist.is_initialized()):
        return disp_local
    world = dist.get_world_size(group)
    if world == 1:
        return disp_local
    sizes = None
    if total is not None:
        sizes = [shard_range(total, r, world) for r in range(world)]
        sizes = [e - b for b, e in sizes]
    if sizes is None or len(set(sizes)) == 1:
        out = disp_local.new_empty((world * disp_local.shape[0],) + tuple(disp_local.shape[1:]))
        if dist.get_backend(group) == "gloo":
            parts = list(out.chunk(world, 0))
            dist.all_gather(parts, disp_local.contiguous(), group=group)
            return torch.cat(parts, 0)
        dist.all_gather_into_tensor(out, disp_local.contiguous(), group=group)
        return out
    mx = max(sizes)
    pad = disp_local.new_zeros((mx,) + tuple(disp_local.shape[1:]))
    pad[: disp_local.shape[0]] = disp_local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], 0)


class OverlappedGather:
    """The result gather of a batch-sharded job, taken OFF the compute stream ("completely overlappable with the
    next batch").  `submit(disp)` is called on the compute stream right after a step: a side stream waits for that point, copies
    the rank's disparities into a staging slot (so the next step -- a hipGraph replay writing the same static output buffer --
    may start as soon as the 2 MB copy is done) and issues ONE RCCL all_gather_into_tensor from there; the compute stream only
    waits for the copy.  `depth` slots are cycled; a slot's previous collective is awaited (stream-side) before it is reused.
    `finish()` makes the calling stream wait for everything in flight.  Without a process group (or world 1) it is the identity.
    On CPU / gloo there are no streams: the gather runs synchronously (tests, launcher plumbing)."""

    def __init__(self, depth=2, group=None, single_rank_too=False):
        self.depth, self.group = depth, group
        # (single_rank_too: run the collective on a 1-rank group as well -- the RCCL smoke path of bench.py --force-dist)
        self.active = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or single_rank_too)
        self.slots = [None] * depth
        self.i = 0
        self.side = None

    def submit(self, disp):
        if not self.active:
            return disp
        if not disp.is_cuda:
            return gather_disparity(disp, group=self.group)
        world = dist.get_world_size(self.group)
        k = self.i % self.depth
        self.i += 1
        if self.side is None:
            self.side = torch.cuda.Stream(device=disp.device)
        slot = self.slots[k]
        if slot is None or slot["stage"].shape != disp.shape:
            slot = self.slots[k] = {"stage": torch.empty_like(disp, memory_format=torch.contiguous_format),
                                    "out": disp.new_empty((world * disp.shape[0],) + tuple(disp.shape[1:])), "work": None}
        main = torch.cuda.current_stream(disp.device)
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            if slot["work"] is not None:
                slot["work"].wait()                      # the slot's previous collective (stream-side wait, no host block)
            slot["stage"].copy_(disp, non_blocking=True)
            copied = torch.cuda.Event()
            copied.record(self.side)
            slot["work"] = dist.all_gather_into_tensor(slot["out"], slot["stage"], group=self.group, async_op=True)
        main.wait_event(copied)
        return slot["out"]

    def finish(self):
        for slot in self.slots:
            if slot is not None and slot.get("work") is not None:
                slot["work"].wait()
                slot["work"] = None
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)


def _parse_cpulist(text):
    """'0-23,96-119' -> sorted list of CPU ids (the format of /sys/devices/system/node/node*/cpulist)."""
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return sorted(cpus)


def pin_to_gpu_numa(device_index, sysfs="/sys", pci=None):
    """Bind this rank's host threads to the NUMA node its GPU hangs off (one process per GPU: the pinned H2D / D2H rings of
    nmrf_amd.driver.StereoStream are then allocated and touched on the memory next to the device, and the launch thread does not
    migrate across sockets).  Call BEFORE the pinned allocations.  The node comes from the GPU's PCI address
    (/sys/bus/pci/devices/<domain:bus:dev.fn>/numa_node), its CPUs from /sys/devices/system/node/node<N>/cpulist, intersected with
    the CPUs this process may already use (a container's cpuset).  Best effort and never fatal: returns a record of what was done
    ({"pinned": False, "why": ...} when the topology is not exposed -- single-socket hosts report numa_node -1)."""
    rec = {"pinned": False, "device": int(device_index)}
    try:
        if not hasattr(os, "sched_setaffinity"):
            return dict(rec, why="no sched_setaffinity on this platform")
        if pci is None:                                       # (pci: the address given by the caller -- tests)
            p = torch.cuda.get_device_properties(device_index)
            pci = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        addr = pci
        rec["pci"] = addr
        with open(os.path.join(sysfs, "bus/pci/devices", addr, "numa_node")) as f:
            node = int(f.read().strip())
        rec["numa_node"] = node
        if node < 0:
            return dict(rec, why="numa_node -1 (single node / not exposed)")
        with open(os.path.join(sysfs, "devices/system/node/node%d/cpulist" % node)) as f:
            cpus = set(_parse_cpulist(f.read()))
        allowed = cpus & set(os.sched_getaffinity(0))
        if not allowed:
            return dict(rec, why="no allowed CPU on node %d" % node)
        os.sched_setaffinity(0, allowed)
        return dict(rec, pinned=True, cpus=len(allowed))
    except Exception as e:                                   # missing sysfs entry, no such attribute, permission: leave the affinity alone
        return dict(rec, why="%s: %s" % (type(e).__name__, e))
