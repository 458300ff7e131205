"""CPU, world_size 2, gloo: the album reduction protocol of loudgain_amd.album
(the N > 1 path) gives the same album loudness / LRA / peak as the oracle's
ebur128_loudness_global_multiple / _range_multiple over all tracks."""
import math
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleShard:
    """Stands in for DeviceShard on the CPU: records from the oracle's block lists."""

    def __init__(self, states, group=None):
        import torch
        from loudgain_amd.album import common_slots
        self.gate = [s.gating_blocks() for s in states]
        stb = [s.shortterm_blocks() for s in states]
        allb = np.concatenate(self.gate) if self.gate else np.zeros(0)
        st = np.concatenate(stb) if stb else np.zeros(0)
        peak = max([s.peak() for s in states], default=0.0)
        slots = common_slots(len(st), group)
        rec1 = np.zeros(4 + slots)
        rec1[:4] = [allb.sum(), float(len(allb)), peak, float(len(st))]
        rec1[4:4 + len(st)] = st
        self.rec1 = torch.from_numpy(rec1)
        self.rec2 = torch.zeros(2, dtype=torch.float64)
        self.index = 0
        self.result = None

    def stage2(self, all1, world):
        if all1 is None:
            all1, world = self.rec1, 1
        heads = all1.view(world, -1)[:, :4]
        sum_abs = n_abs = n_st = 0.0
        peak = 0.0
        for r in range(world):   # rank order, like lgd_album_part2_kernel
            sum_abs += float(heads[r, 0]); n_abs += float(heads[r, 1]); n_st += float(heads[r, 3])
            peak = max(peak, float(heads[r, 2]))
        heads.zero_()            # what is left is the album's short-term list
        thr = sum_abs / n_abs * 10.0 ** (-10.0 / 10.0) if n_abs > 0 else 0.0
        allb = np.concatenate(self.gate) if self.gate else np.zeros(0)
        sel = allb[allb >= thr]
        self.rec2[0], self.rec2[1] = float(sel.sum()), float(len(sel))
        self.peak, self.st_all = peak, all1

    def finish(self, all2, world):
        from loudgain_amd.album import album_from_partials
        if all2 is None:
            all2, world = self.rec2, 1
        p = all2.view(world, 2)
        sum_rel = n_rel = 0.0
        for r in range(world):
            sum_rel += float(p[r, 0]); n_rel += float(p[r, 1])
        self.result = album_from_partials(sum_rel, n_rel, self.st_all.numpy(), self.peak)


def _tracks():
    from loudgain_amd import synth
    specs = [(48000, 2, 14.0, 1, 1.0), (48000, 2, 9.0, 2, 0.04), (44100, 1, 11.0, 3, 1.0),
             (48000, 2, 6.5, 4, 0.5), (96000, 2, 5.0, 5, 1.0)]
    out = []
    for rate, ch, secs, seed, g in specs:
        pcm = synth.snap_s16_numpy(synth.track_numpy(int(rate * secs), ch, rate, seed=seed, step_s=1.5) * g)
        out.append((pcm, rate))
    return out


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from loudgain_amd.album import reduce_album, shard_indices
    from oracle import lgoracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tracks = _tracks()
    mine = shard_indices(len(tracks), rank, world)
    states = [lgoracle.State(tracks[i][0].shape[1], tracks[i][1]).add(tracks[i][0]) for i in mine]
    shard = OracleShard(states)
    reduce_album(shard)
    q.put((rank, mine, shard.result))
    dist.destroy_process_group()


def test_album_protocol_world2_gloo():
    import torch.multiprocessing as mp
    from oracle import lgoracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tracks = _tracks()
    states = [lgoracle.State(p.shape[1], r).add(p) for p, r in tracks]
    want_l, want_r = lgoracle.album_loudness(states), lgoracle.album_lra(states)
    want_p = max(s.peak() for s in states)
    owned = sorted(i for _, mine, _ in res for i in mine)
    assert owned == list(range(len(tracks)))          # every track scanned exactly once
    for rank, mine, got in res:
        assert abs(got["loudness"] - want_l) <= 1e-9, (rank, got, want_l)
        assert abs(got["lra"] - want_r) <= 1e-9
        assert got["peak"] == want_p


def test_shard_indices_round_robin():
    from loudgain_amd.album import shard_indices
    assert shard_indices(10, 0, 4) == [0, 4, 8] and shard_indices(10, 3, 4) == [3, 7]
    assert sorted(sum((shard_indices(1000, r, 8) for r in range(8)), [])) == list(range(1000))
    assert shard_indices(2, 5, 8) == []


def test_album_from_partials_empty():
    from loudgain_amd.album import album_from_partials
    r = album_from_partials(0.0, 0.0, np.zeros(0), 0.0)
    assert r["loudness"] == -math.inf and r["lra"] == 0.0
