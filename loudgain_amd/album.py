"""Album mode across GPUs: tracks sharded one process per GPU, album result by
two small all-gathers (RCCL over xGMI on the GPU box; gloo in the CPU tests).

What it replaces: the in-process walk over all ebur128 states in
ebur128_loudness_global_multiple / ebur128_loudness_range_multiple and the
album-peak loop (/root/reference/src/scan.c:383-391, :359-378).

Exact formulation (the reference gates over an exact
block list, not a histogram, so a single histogram all-reduce would not be
exact):
  1. every rank publishes record 1 = {sum_abs, n_abs, peak, n_st | its listed 3 s
     energies | 0-padding}; ALL-GATHER -> every rank folds the heads in rank order
     (same bits everywhere): relative threshold Gamma_r = 0.1 * sum_abs / n_abs,
     album peak = max; the rest of the gathered buffer is the album's short-term list
  2. every rank re-counts its own blocks against Gamma_r -> record 2 = {sum_rel, n_rel};
     ALL-GATHER -> album loudness
  3. every rank selects the two LRA order statistics from the gathered list itself
     (deterministic, identical everywhere).
Track -> rank assignment is round-robin (t mod world_size); per-track results
are final on the owning rank, no PCM ever crosses a link (<= a few MB do).

Back-to-back album scans pipeline: the exchange of scan k runs on its own stream
(ordered behind scan k by lgd_album_join) while the kernels of the following scans
already run; the engine's workspaces are used in turn and one is not scanned into
again before the stage 3 that read it has finished.
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


def common_slots(n_slots, group=None, device=None):
    """Record 1 must have the same length on every rank: the largest short-term slot
    count of any rank (one tiny all-reduce, once per plan)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return int(n_slots)
    m = torch.tensor([int(n_slots)], dtype=torch.int64, device=device)
    _all_reduce(m, dist.ReduceOp.MAX, group)
    return int(m.item())


class DeviceShard:
    """This rank's part of an album on one GPU (wraps a DeviceScanner).

    The engine uses several workspaces in turn, so the records of the most recent
    scan live at changing addresses: `rec1` / `rec2` always refer to the most
    recent `run_tracks`."""

    def __init__(self, scanner, tracks, rates, true_peak=True, slots=None, group=None):
        import torch
        self.sc = scanner
        self.device = torch.device("cuda", scanner.device)
        self.sc.set_param("album_slots", 0)
        self.sc.plan(tracks, rates, true_peak=true_peak, album="part1")
        if slots is None:
            slots = common_slots(self.sc.album_records()[1] - 4, group, self.device)
        self.sc.set_param("album_slots", slots)
        self.sc.plan(tracks, rates, true_peak=true_peak, album="part1")
        self._views = {}
        self.stream = None          # stream the scans are handed to
        self.reduce_stream = None   # stream the exchange + stages 2/3 run on (None: same)
        self.index = 0              # which workspace the most recent scan used
        self._select()

    def _select(self):
        r1, n, r2 = self.sc.album_records()
        if r1 not in self._views:
            self._views[r1] = (len(self._views), wrap_device_doubles(r1, n, self.device),
                               wrap_device_doubles(r2, 2, self.device))
        self.index, self.rec1, self.rec2 = self._views[r1]

    def _rs(self):
        return self.reduce_stream if self.reduce_stream is not None else self.stream

    def run_tracks(self, stream=None):
        self.stream = stream
        self.sc.execute(stream)  # per-track results + record 1
        self._select()
        self.sc.album_join(self._rs())  # the scan may run on a stream of the engine

    def stage2(self, all1, world):
        self.sc.album_stage2(all1.data_ptr() if all1 is not None else None, world, self._rs())

    def finish(self, all2, world):
        self.sc.album_stage3(all2.data_ptr() if all2 is not None else None, world, self._rs())

    def fetch(self):
        return self.sc.fetch()


def _gather(t, out, group):
    """all-gather `t` (same length on every rank) into `out` = [world * len].  RCCL moves
    HBM tensors directly; any other backend (gloo in the tests) gets host copies of
    the few bytes being exchanged."""
    import torch
    import torch.distributed as dist
    if not t.is_cuda or dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(out, t, group=group)
    else:
        src = t.cpu()
        parts = [torch.empty_like(src) for _ in range(dist.get_world_size(group))]
        dist.all_gather(parts, src, group=group)
        out.copy_(torch.cat(parts))
    return out


def _all_reduce(t, op, group):
    import torch.distributed as dist
    w = t if (not t.is_cuda or dist.get_backend(group) == "nccl") else t.cpu()
    dist.all_reduce(w, op=op, group=group)
    if w is not t:
        t.copy_(w)


def reduce_album(shard, group=None, always_exchange=False, buffers=None):
    """Run the protocol over `shard` (DeviceShard, or any object with rec1 / rec2
    tensors, stage2(all1, world) and finish(all2, world)).  Collectives are enqueued on
    the current stream.  `buffers`: optional dict the gather buffers are kept in,
    keyed by shard.index (a pipelined caller must not reuse one while it is read)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (always_exchange and dist.is_initialized()):
        shard.stage2(None, 1)
        shard.finish(None, 1)
        return None
    r1, r2 = shard.rec1, shard.rec2
    key = getattr(shard, "index", 0)
    if buffers is not None and key in buffers:
        all1, all2 = buffers[key]
    else:
        all1 = torch.empty(world * r1.numel(), dtype=torch.float64, device=r1.device)
        all2 = torch.empty(world * 2, dtype=torch.float64, device=r1.device)
        if buffers is not None:
            buffers[key] = (all1, all2)
    _gather(r1, all1, group)
    shard.stage2(all1, world)
    _gather(r2, all2, group)
    shard.finish(all2, world)
    return all1


class DistributedAlbumScanner:
    """scan.c's album mode over `world` GPUs: plan once, execute many times.

    `execute` returns as soon as everything is enqueued: the scan on `stream` (or the
    engine's second stream), the album exchange on this object's own stream."""

    def __init__(self, scanner, tracks, rates, true_peak=True, group=None, always_exchange=False):
        import torch
        self.group = group
        self.always_exchange = always_exchange
        # back-to-back album scans pipeline on the engine's two streams (the track buffers
        # stay untouched until fetch(): the contract of "overlap" 1)
        scanner.set_param("overlap", 1)
        self.shard = DeviceShard(scanner, tracks, rates, true_peak, group=group)
        self.shard.reduce_stream = torch.cuda.Stream(device=self.shard.device)
        self.buffers = {}
        torch.cuda.synchronize(self.shard.device)

    def execute(self, stream=None):
        import torch
        self.shard.run_tracks(stream)
        with torch.cuda.stream(self.shard.reduce_stream):
            reduce_album(self.shard, self.group, self.always_exchange, self.buffers)
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
