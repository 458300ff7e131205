"""GPU: the sharded album path end to end.  Two ranks share the one GPU of the
test box (RCCL needs one device per rank, so the exchange runs over gloo with
host copies of the few exchanged bytes; the kernels and stages are the real
ones).  The 8-GPU RCCL run itself is the driver's scaling bench."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _many_tracks(n=300):
    """BASELINE.json configs[3] in shape (many stereo tracks, one album, sharded), short tracks."""
    from loudgain_amd import synth
    rng = np.random.default_rng(303)
    proto = synth.track_numpy(48000 * 5, 2, 48000, seed=77, step_s=0.6)
    out = []
    for i in range(n):
        frames = int(48000 * rng.uniform(0.4, 3.0))
        off = int(rng.integers(0, proto.shape[0] - frames + 1))
        g = float(10.0 ** (rng.uniform(-30.0, 0.0) / 20.0))
        out.append((synth.snap_s16_numpy(proto[off:off + frames] * g), 48000))
    return out


def _tracks(kind="mixed"):
    if kind == "many":
        return _many_tracks()
    from loudgain_amd import synth
    if kind in ("long", "long1"):   # > 8192 listed 3 s blocks in the gathered album list: the multi-workgroup LRA kernels
        # ("long1": with scratch sized for ONE rank -- "album_world" 1 -- on two: the single-workgroup streaming select)
        return [(synth.track_numpy(8000 * 60 * 31, 1, 8000, seed=40 + i, step_s=13.0 + i), 8000) for i in range(6)]
    specs = [(48000, 2, 14.0, 1, 1.0), (48000, 2, 9.0, 2, 0.04), (44100, 1, 11.0, 3, 1.0),
             (48000, 2, 6.5, 4, 0.5), (96000, 2, 5.0, 5, 1.0), (48000, 6, 4.5, 6, 0.8), (48000, 2, 0.2, 7, 1.0)]
    out = []
    for rate, ch, secs, seed, g in specs:
        pcm = synth.snap_s16_numpy(synth.track_numpy(int(rate * secs), ch, rate, seed=seed, step_s=1.5) * g)
        out.append((pcm, rate))
    return out


def _worker(rank, world, port, q, backend="gloo", kind="mixed"):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from loudgain_amd.album import DistributedAlbumScanner, shard_indices
    from loudgain_amd.device import DeviceScanner
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    tracks = _tracks(kind)
    mine = shard_indices(len(tracks), rank, world)
    dev = [torch.from_numpy(tracks[i][0]).cuda() for i in mine]
    sc = DeviceScanner(0)
    if kind == "long1":
        sc.set_param("album_world", 1)
    job = DistributedAlbumScanner(sc, dev, [tracks[i][1] for i in mine],
                                  always_exchange=(backend == "nccl"))
    stream = torch.cuda.Stream()
    # five pipelined scans (both workspaces, scan k+1 under the exchange of scan k): the
    # last one must still be right, partials are rebuilt every time
    for _ in range(5):
        job.execute(stream)
    tr, album = job.fetch()
    q.put((rank, mine, tr, album))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,backend,kind", [(1, "gloo", "mixed"), (2, "gloo", "mixed"), (3, "gloo", "mixed"),
                                                (1, "nccl", "mixed"), (3, "gloo", "many"), (2, "gloo", "long"), (2, "gloo", "long1")])
def test_sharded_album_matches_oracle(oracle, world, backend, kind):
    """(1, "nccl"): the RCCL calls themselves (all-reduce SUM/MAX on engine-owned HBM,
    all-gather into a tensor) with one rank, where every collective is an identity."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 2000) + world + (7 if backend == "nccl" else 0) + (13 if kind == "many" else 0) \
        + (29 if kind == "long" else 0) + (41 if kind == "long1" else 0)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, backend, kind)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tracks = _tracks(kind)
    refs = [oracle.scan_track(p, r) for p, r in tracks]
    states = [r["state"] for r in refs]
    want_l, want_r = oracle.album_loudness(states), oracle.album_lra(states)
    det = oracle.gating_detail(states)
    for rank, mine, tr, album in res:
        assert abs(album["loudness"] - want_l) <= 1e-6
        assert abs(album["lra"] - want_r) <= 1e-6
        assert abs(album["peak"] - max(r["peak"] for r in refs)) <= 1e-4
        assert album["n_abs"] == det["n_abs"] and album["n_rel"] == det["n_rel"]
        for i, got in zip(mine, tr):   # per-track results are final on the owning rank
            assert got["n_abs"] == refs[i]["n_abs"] and abs(got["loudness"] - refs[i]["loudness"]) <= 1e-6 \
                or refs[i]["n_rel"] == 0
