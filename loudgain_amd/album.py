"""Album mode across GPUs: tracks sharded one process per GPU, album result by
small collectives (RCCL over xGMI on the GPU box; gloo in the CPU tests).

What it replaces: the in-process walk over all ebur128 states in
ebur128_loudness_global_multiple / ebur128_loudness_range_multiple and the
album-peak loop (/root/reference/src/scan.c:383-391, :359-378).

Exact formulation (the reference gates over an exact
block list, not a histogram, so a single histogram all-reduce would not be
exact):
  1. all-reduce SUM {sum_abs, n_abs, n_st} and all-reduce MAX {peak}
     -> relative threshold  Gamma_r = 0.1 * sum_abs / n_abs  on every rank
  2. every rank re-counts its own blocks against Gamma_r;
     all-reduce SUM {sum_rel, n_rel}          -> album loudness
  3. all-gather of the listed 3 s energies (<= a few MB) -> every rank selects
     the two LRA order statistics itself (deterministic, identical everywhere).
Track -> rank assignment is round-robin (t mod world_size); per-track results
are final on the owning rank, no PCM ever crosses a link.
"""
import numpy as np


def shard_indices(n_tracks, rank, world):
    """Round-robin track ownership: rank r scans tracks {t : t mod world == r}."""
    return list(range(rank, n_tracks, world))


class _DevArray:
    """Zero-copy view of engine-owned HBM for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr, n, typestr="<f8"):
        self.__cuda_array_interface__ = {
            "shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2,
            "strides": None,
        }


def wrap_device_doubles(ptr, n, device):
    import torch
    if n == 0:
        return torch.zeros(0, dtype=torch.float64, device=device)
    return torch.as_tensor(_DevArray(ptr, n), device=device)


class DeviceShard:
    """This rank's part of an album on one GPU (wraps a DeviceScanner)."""

    def __init__(self, scanner, tracks, rates, true_peak=True):
        import torch
        self.sc = scanner
        self.device = torch.device("cuda", scanner.device)
        self.sc.plan(tracks, rates, true_peak=true_peak, album="part1")
        p1, p2, st, n_st = self.sc.album_part_ptrs()
        self.part1 = wrap_device_doubles(p1, 4, self.device)
        self.part2 = wrap_device_doubles(p2, 2, self.device)
        self.st = wrap_device_doubles(st, n_st, self.device)
        self.stream = None

    def run_tracks(self, stream=None):
        self.stream = stream
        self.sc.execute(stream)  # per-track results + part1

    def stage2(self):
        self.sc.album_stage2(self.stream)

    def st_energies(self):
        return self.st

    def finish(self, st_all):
        self.sc.album_stage3(st_all.data_ptr() if st_all.numel() else None, st_all.numel(),
                             self.stream)

    def fetch(self):
        return self.sc.fetch()


def _on_wire(t, group):
    """RCCL moves HBM tensors directly.  Any other backend (gloo in the tests)
    gets a host copy of the few bytes being exchanged."""
    import torch.distributed as dist
    return t if (not t.is_cuda or dist.get_backend(group) == "nccl") else t.cpu()


def _all_reduce(t, op, group):
    import torch.distributed as dist
    w = _on_wire(t, group)
    dist.all_reduce(w, op=op, group=group)
    if w is not t:
        t.copy_(w)


def reduce_album(shard, group=None, st_slots_max=None):
    """Run the three-step protocol over `shard` (DeviceShard, or any object with
    part1 / part2 tensors and stage2() / st_energies() / finish()).  Collectives
    are enqueued on the current stream; returns the gathered short-term array."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    p1, p2 = shard.part1, shard.part2
    if world > 1:
        peak = p1[2:3].clone()
        _all_reduce(p1, dist.ReduceOp.SUM, group)       # sum_abs, n_abs, (peak), n_st
        _all_reduce(peak, dist.ReduceOp.MAX, group)
        p1[2:3].copy_(peak)
    shard.stage2()
    if world > 1:
        _all_reduce(p2, dist.ReduceOp.SUM, group)       # sum_rel, n_rel
    st = shard.st_energies()
    if world > 1:
        if st_slots_max is None:
            m = torch.tensor([st.numel()], dtype=torch.int64, device=st.device)
            _all_reduce(m, dist.ReduceOp.MAX, group)
            st_slots_max = int(m.item())
        pad = torch.zeros(st_slots_max, dtype=torch.float64, device=st.device)  # 0.0 == unlisted
        pad[:st.numel()].copy_(st)
        st_all = torch.empty(st_slots_max * world, dtype=torch.float64, device=st.device)
        wire_in = _on_wire(pad, group)
        if wire_in is pad:
            dist.all_gather_into_tensor(st_all, pad, group=group)
        else:
            parts = [torch.empty_like(wire_in) for _ in range(world)]
            dist.all_gather(parts, wire_in, group=group)
            st_all.copy_(torch.cat(parts))
    else:
        st_all = st
    shard.finish(st_all)
    return st_all


class DistributedAlbumScanner:
    """scan.c's album mode over `world` GPUs: plan once, execute many times."""

    def __init__(self, scanner, tracks, rates, true_peak=True, group=None):
        import torch
        import torch.distributed as dist
        self.group = group
        self.shard = DeviceShard(scanner, tracks, rates, true_peak)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # slot counts are static for a plan: exchange them once, not per scan
        n = self.shard.st.numel()
        if self.world > 1:
            m = torch.tensor([n], dtype=torch.int64, device=self.shard.device)
            _all_reduce(m, dist.ReduceOp.MAX, group)
            n = int(m.item())
        self.st_slots_max = n

    def execute(self, stream=None):
        import torch
        self.shard.run_tracks(stream)
        if stream is not None:
            with torch.cuda.stream(stream):
                reduce_album(self.shard, self.group, self.st_slots_max)
        else:
            reduce_album(self.shard, self.group, self.st_slots_max)
        return self

    def fetch(self):
        return self.shard.fetch()


# ---- host restatement of the album finish, used by the gloo tests and by
# callers that already hold block energies on the host ------------------------
def album_from_partials(sum_rel, n_rel, st_all, peak):
    """Final album numbers from fully reduced partials (numpy, float64)."""
    st = np.asarray(st_all, dtype=np.float64)
    st = np.sort(st[st > 0.0])
    loud = 10.0 * (np.log(sum_rel / n_rel) / np.log(10.0)) - 0.691 if n_rel > 0 else -np.inf
    lra = 0.0
    if st.size:
        thr = 10.0 ** (-20.0 / 10.0) * (st.sum() / st.size)
        rel = st[~(st < thr)]
        if rel.size:
            hi = rel[int((rel.size - 1) * 0.95 + 0.5)]
            lo = rel[int((rel.size - 1) * 0.1 + 0.5)]
            lra = (10.0 * (np.log(hi) / np.log(10.0)) - 0.691) - (10.0 * (np.log(lo) / np.log(10.0)) - 0.691)
    return dict(loudness=float(loud), lra=float(lra), peak=float(peak))
