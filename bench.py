#!/usr/bin/env python3
"""Headline benchmark: Msamples/s scanned (48 kHz stereo f32), BASELINE.json.

  python bench.py [--gpus N --steps K --warmup W] [--workload c2|c3|c4|c5]

N > 1 may be started either way: `python bench.py --gpus N` spawns its N ranks itself (fresh
child processes, before this process touches a GPU), and under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the ranks read
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.  One rank per GPU, RCCL.

Workloads (BASELINE.json configs; all synthetic, generated in HBM, every
value on the S16 grid; a "step" = one full scan of the rank's tracks incl. gating / LRA epilogue
and, in album mode, the album reduction):
  c2  configs[1]  one 60 min 48 kHz stereo f32 buffer per GPU (1 382 400 000 B), true peak OFF
                  (a build-only switch).  The contract `value` at N = 1; the same line carries a
                  "c3" object, because the reference always has true peak on (scan.c:203-207).
                  N > 1: every rank scans its own buffer as one track of an N-track album (weak).
  c3  configs[2]  the same buffer with the 4x true-peak interpolator = the reference's semantics.
  c4  configs[3]  album of 1000 stereo 48 kHz tracks, track t = 180 s + (t mod 7) * 30 s, seed t,
                  dealt round-robin (t mod N), true peak on, album result over all 1000 (strong).
                  The default workload for N > 1.
  c5  configs[4]  album of 64 tracks cycling 44.1 / 48 / 96 / 192 kHz x mono / stereo / 5.1, 120 s
                  each, round-robin over the ranks, true peak on (4x / 2x / none by rate).
`value` = samples of all ranks per step / max-over-ranks wall time per step.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak
METRIC = "Msamples/s scanned (48 kHz stereo f32)"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="0 = the workload's default")
    ap.add_argument("--warmup", type=int, default=-1, help="-1 = the workload's default")
    ap.add_argument("--workload", default="auto", choices=["auto", "c2", "c3", "c4", "c5"],
                    help="auto: c2 (with the c3 object) on one GPU, c4 on several")
    ap.add_argument("--minutes", type=float, default=60.0, help="c2 / c3: buffer length")
    ap.add_argument("--tracks", type=int, default=0, help="c4 / c5: number of tracks (0 = 1000 / 64)")
    ap.add_argument("--track-scale", type=float, default=1.0, help="c4 / c5: scales every track length (tests)")
    ap.add_argument("--material", default="steps", choices=["steps", "adversarial", "silence", "noise", "limited"],
                    help="c2 / c3: steps = SURVEY 8d programme material; adversarial = constant-amplitude fs/4 "
                         "sine sampled on its peaks (no true-peak output can be pruned)")
    ap.add_argument("--no-c3", action="store_true", help="c2 on one GPU: skip the c3 object")
    ap.add_argument("--pcm", default="f32", choices=["f32", "s16"],
                    help="element format of the resident PCM: f32 (BASELINE.json's configs) or the same samples as the "
                         "interleaved int16 the reference feeds libebur128 (scan.c:442-448), read by the S16 kernel variants")
    ap.add_argument("--no-h2d", action="store_true", help="skip the host-buffer (PCIe-inclusive) figures")
    ap.add_argument("--no-c4", action="store_true", help="c2 on one GPU: skip the c4 object (the N = 1 anchor of the scaling series)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=int, default=1800, help="audio seconds of the 1-thread CPU leg")
    ap.add_argument("--cpu-procs", type=int, default=0, help="worker processes of the CPU leg (0 = host cores, <= 16)")
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--seg-subblocks", type=int, default=0)
    ap.add_argument("--waves-per-cu", type=int, default=0)
    ap.add_argument("--warm-subblocks", type=int, default=-1)
    ap.add_argument("--no-tp-prune", action="store_true")
    ap.add_argument("--param", action="append", default=[], metavar="NAME=VALUE",
                    help="any engine parameter (lgd_set_param), e.g. merge_launches=0")
    ap.add_argument("--debug", type=int, default=0, help="kernel floor measurement: 1 no loads, 2 no arithmetic")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal of the N>1 path on one GPU: RCCL group of one rank, album exchange every step")
    ap.add_argument("--serial", action="store_true",
                    help="no stream pipelining of consecutive scans (the mode rocprofv3 kernel durations are quoted in)")
    ap.add_argument("--settle", type=int, default=-1, help="serial launches before the kernel timing (-1 = default)")
    ap.add_argument("--launches", type=int, default=-1, help="serial launches timed for the roofline (-1 = default)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsals (the launcher on the CPU; with --same-gpu the whole N > 1 path on one GPU)")
    ap.add_argument("--same-gpu", action="store_true",
                    help="every rank uses GPU 0 (needs --backend gloo: RCCL wants one device per rank)")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="no GPU work: ranks rendezvous, all-reduce one number, rank 0 prints a JSON line")
    ap.add_argument("--selftest-die", type=int, default=-1, metavar="RANK",
                    help="launcher selftest: this rank exits with code 3 after the rendezvous, the others go on into another collective")
    ap.add_argument("--launch-timeout", type=float, default=540.0,
                    help="self-launched ranks (--gpus N without a launcher): seconds until the launcher gives up and ends them")
    return ap.parse_args(argv)


# --------------------------------------------------------------------------- launcher --
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_children(n, argv, timeout_s=540.0):
    """`python bench.py --gpus N` without a launcher: N fresh processes, one per GPU.  Nothing in
    this (parent) process has touched a GPU; it only relays rank 0's line and the exit codes.
    All ranks are watched: the first one that exits non-zero (or the deadline, kept under the
    driver's own 600 s) ends the others -- a rank that dies after the rendezvous would otherwise
    leave its peers inside a collective for good -- and the launcher exits non-zero."""
    import threading
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), LGD_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out = []
    reader = threading.Thread(target=lambda: out.extend(procs[0].stdout.readlines()), daemon=True)
    reader.start()
    deadline = time.monotonic() + timeout_s
    why = None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            why = "ranks failed: %s" % bad
            break
        if all(rc == 0 for rc in rcs):
            break
        if time.monotonic() > deadline:
            why = "no result after %.0f s (ranks still running: %s)" % (
                timeout_s, [r for r, rc in enumerate(rcs) if rc is None])
            break
        time.sleep(0.05)
    if why:  # end exactly the processes started here
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_kill = time.monotonic() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    reader.join(timeout=5.0)
    for ln in out:   # stdout carries rank 0's JSON line only (libraries chat on stdout too)
        (sys.stdout if ln.startswith("{") and not why else sys.stderr).write(ln if ln.endswith("\n") else ln + "\n")
    sys.stdout.flush()
    if why:
        sys.stderr.write("bench.py: %s\n" % why)
        return 1
    return 0


# ----------------------------------------------------------------------- CPU baseline --
def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_worker(args):
    """One worker of the multi-process leg: its own track, as rgbpm2 hands out albums
    (/root/reference/bin/rgbpm2:170-175).  Returns (samples, seconds of the scan alone)."""
    idx, seconds, rate, start_at = args
    from loudgain_amd import synth
    from oracle import lgoracle
    lgoracle.lib()
    pcm = synth.track_numpy(seconds * rate, 2, rate, seed=1000 + idx)
    while time.time() < start_at:   # all workers scan at the same time
        time.sleep(0.001)
    t0 = time.perf_counter()
    st = lgoracle.State(2, rate).add(pcm, chunk=4096)
    st.loudness(), st.lra(), st.peak()
    return pcm.size, time.perf_counter() - t0, t0


def cpu_baseline(args, rate=48000):
    """CPU restatement of the reference path (oracle/: scan.c + libebur128 1.2.4 semantics, all five
    modes on as scan.c:203-207, i.e. INCLUDING the 4x true peak -- the GPU figure it stands next to
    is c3).  Leg (i): 1 process, 1 thread -- how loudgain runs.  Leg (ii): P processes, one track
    each -- how bin/rgbpm2 parallelises.  Runs before this process touches the GPU."""
    import multiprocessing as mp
    from loudgain_amd import synth
    from oracle import lgoracle
    lgoracle.lib()
    secs = int(min(args.cpu_seconds, args.minutes * 60))
    pcm = synth.track_numpy(secs * rate, 2, rate, seed=0)
    t0 = time.perf_counter()
    st = lgoracle.State(2, rate).add(pcm, chunk=4096)
    loud = st.loudness()
    st.lra(), st.peak()
    dt = time.perf_counter() - t0
    one = pcm.size / dt / 1e6
    del pcm
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    nproc = args.cpu_procs or min(cores, 16)   # (a GPU box gives one GPU's job a 16-core share)
    wsecs = max(10, min(secs, 600))
    multi = None
    if nproc > 1:
        ctx = mp.get_context("fork")           # (no GPU state exists in this process yet)
        with ctx.Pool(nproc) as pool:
            start_at = time.time() + 1.5 + 0.012 * wsecs
            res = pool.map(_cpu_worker, [(i, wsecs, rate, start_at) for i in range(nproc)])
        wall = max(t0 + d for _, d, t0 in res) - min(t0 for _, _, t0 in res)
        v = sum(n for n, _, _ in res) / wall / 1e6
        multi = dict(value=round(v, 2), unit="Msamples/s", cores=nproc,
                     model=_cpu_model(), host_cores=cores,
                     # SURVEY 8d asks for P = host cores; a GPU box of this pool gives one GPU's job a 16-core share of
                     # its host (worker pools are to be sized to that), so P is capped -- both figures are printed
                     cap="P = min(host cores, 16): one GPU's CPU share on this pool" if not args.cpu_procs else "--cpu-procs",
                     at_host_cores=dict(cores=cores, value=round(v / nproc * cores, 1), unit="Msamples/s",
                                        kind="EXTRAPOLATED linearly from the measured %d processes (not run: the job's share "
                                             "is %d cores)" % (nproc, nproc)) if cores > nproc else None,
                     sample="%d processes, one %d s 48 kHz stereo track each (as bin/rgbpm2 hands out work), "
                            "%.2f s wall" % (nproc, wsecs, wall))
    return dict(value=round(one, 2), unit="Msamples/s", cores=1, kind="port", model=_cpu_model(),
                stands_next_to="c3 (the reference always computes the 4x true peak, scan.c:203-207)",
                sample="first %d s of a c2/c3-style track (numpy Philox twin of the GPU generator), 1 thread, oracle "
                       "-O2, all modes incl. 4x true peak (CPU restatement of reference path); %.2f s wall; %.3f LUFS"
                       % (secs, dt, loud),
                multi_process=multi)


# -------------------------------------------------------------------------- workloads --
def c4_track_frames(t, rate=48000, scale=1.0):
    return int((180 + (t % 7) * 30) * rate * scale)


C5_RATES = (44100, 48000, 96000, 192000)
C5_CHANNELS = (1, 2, 6)


def c5_track_spec(t, scale=1.0):
    rate = C5_RATES[t % 4]
    ch = C5_CHANNELS[(t // 4) % 3]
    return rate, ch, int(120 * rate * scale)


def build_tracks(args, workload, rank, world, dev):
    """-> (list of device tensors, list of rates, description, true_peak, album)"""
    import torch
    from loudgain_amd import synth
    if workload in ("c2", "c3"):
        rate, ch = 48000, 2
        frames = int(round(args.minutes * 60 * rate))
        if args.material == "adversarial":
            pcm = synth.adversarial_torch(frames, ch, device=dev)
        elif args.material == "silence":
            pcm = torch.zeros((frames, ch), dtype=torch.float32, device=dev)
        elif args.material == "noise":
            pcm = synth.track_torch(frames, ch, rate, seed=rank, step_s=1e9, device=dev, sine=False)
        elif args.material == "limited":
            pcm = synth.limited_torch(frames, ch, rate, seed=rank, device=dev)
        else:
            pcm = synth.track_torch(frames, ch, rate, seed=rank, device=dev)
        return [pcm], [rate]
    if workload == "c4":
        n = args.tracks or 1000
        mine = range(rank, n, world)
        return ([synth.track_torch(c4_track_frames(t, scale=args.track_scale), 2, 48000, seed=t, device=dev)
                 for t in mine], [48000] * len(mine))
    n = args.tracks or 64
    tracks, rates = [], []
    for t in range(rank, n, world):
        rate, ch, frames = c5_track_spec(t, args.track_scale)
        tracks.append(synth.track_torch(frames, ch, rate, seed=t, device=dev))
        rates.append(rate)
    return tracks, rates


def describe(args, workload, world, distributed):
    d = _describe(args, workload, world, distributed)
    return d if args.pcm == "f32" else d.replace("stereo f32", "stereo") + " [the samples as interleaved int16 in HBM, LGD_PCM_S16]"


def _describe(args, workload, world, distributed):
    if workload in ("c2", "c3"):
        tp = " + 4x true peak" if workload == "c3" else ", no true peak"
        mat = "" if args.material == "steps" else " [%s material]" % args.material
        alb = "; %d-track album, RCCL album reduce per step" % world if distributed else ""
        return "%s: %g min 48 kHz stereo f32 per GPU, K-filter + gated loudness + LRA%s%s%s" % (
            workload.upper(), args.minutes, tp, mat, alb)
    if workload == "c4":
        return ("C4: album of %d stereo 48 kHz tracks of 180 + (t mod 7) * 30 s%s, round-robin over %d GPU(s), "
                "K-filter + gated loudness + LRA + 4x true peak per track, album loudness / range / peak over all"
                % (args.tracks or 1000, "" if args.track_scale == 1.0 else " (x %g)" % args.track_scale, world))
    return ("C5: album of %d tracks cycling 44.1/48/96/192 kHz x mono/stereo/5.1, 120 s each%s, round-robin over "
            "%d GPU(s), per-channel K-weighting (libebur128 default map), 4x / 2x / no interpolator by rate"
            % (args.tracks or 64, "" if args.track_scale == 1.0 else " (x %g)" % args.track_scale, world))


# ------------------------------------------------------------------------- measurement --
class Runner:
    def __init__(self, args, rank, world, local_rank):
        import torch
        from loudgain_amd.device import DeviceScanner
        self.torch = torch
        self.args, self.rank, self.world = args, rank, world
        self.dev = torch.device("cuda", local_rank)
        self.sc = DeviceScanner(local_rank)
        a, sc = args, self.sc
        if a.chunk:
            sc.set_param("chunk", a.chunk)
        if a.seg_subblocks:
            sc.set_param("seg_subblocks", a.seg_subblocks)
        if a.waves_per_cu:
            sc.set_param("waves_per_cu", a.waves_per_cu)
        if a.warm_subblocks >= 0:
            sc.set_param("warm_subblocks", a.warm_subblocks)
        if a.debug:
            sc.set_param("debug", a.debug)
        if a.no_tp_prune:
            sc.set_param("tp_prune", 0)
        for kv in a.param:
            name, value = kv.split("=")
            sc.set_param(name, int(value))
        self.stream = torch.cuda.Stream(device=self.dev)

    def kernel_stats(self, tracks, rates, true_peak, album, launches=64, settle=300):
        """The kernels that read PCM (scan kernels + the true-peak kernels behind them), timed with
        hipEvents on the launch stream around every launch, launches strictly serial -- the mode the
        committed rocprofv3 kernel traces (`bench.py --serial`) are taken in.  In the timed region of
        `run` consecutive scans pipeline on two streams, where a per-launch bracket would include
        queueing behind the previous scan."""
        sc = self.sc
        sc.set_param("overlap", 0)
        sc.set_param("timing", 1)         # hipEvent brackets around the kernels that read PCM: for this section only
        sc.plan(tracks, rates, true_peak=true_peak, album=album)
        for _ in range(settle):           # clocks and caches in the steady state of that mode
            sc.execute(self.stream)
        sc.fetch()
        for _ in range(launches):
            sc.execute(self.stream)
        self.last_results = sc.fetch()
        ks = sc.kernel_ms_stats(launches)
        sc.set_param("timing", 0)         # (the timed regions run without the instrumentation's event packets)
        return ks

    def run(self, job, steps, warmup, distributed):
        torch = self.torch
        import torch.distributed as dist

        def barrier():
            if distributed:
                dist.barrier()
            torch.cuda.synchronize()

        for _ in range(warmup):
            job.execute(self.stream)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            job.execute(self.stream)
        results = job.fetch()  # synchronises the streams, copies the numbers out
        barrier()
        dt = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([dt], dtype=torch.float64, device=self.dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, results

    def step_times(self, job, n):
        """median / min over n individually timed (synchronised) steps"""
        ts = []
        for _ in range(n):
            self.torch.cuda.synchronize()
            t0 = time.perf_counter()
            job.execute(self.stream)
            job.fetch()
            ts.append((time.perf_counter() - t0) * 1e3)
        return dict(median=round(statistics.median(ts), 4), min=round(min(ts), 4), n=n,
                    note="one scan at a time incl. fetch (host launch + D2H of the results included)")


def roofline_block(algo_bytes, ks, dt_step, traffic, timing, kernels):
    achieved = algo_bytes / (ks["scan_mean_ms"] * 1e-3) / 1e9
    return {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
        # (HBM bytes per launch from the committed rocprofv3 --pmc passes -- FETCH_SIZE x 2 + WRITE_SIZE, collected
        # counters-only in their own runs -- not a counter read during this run)
        "traffic_source": None if traffic is None else "profiles/traffic_*.json of this commit (rocprofv3 --pmc passes, tools/pmc.sh)",
        "kernel": kernels, "kernel_ms_mean": round(ks["scan_mean_ms"], 4),
        "kernel_ms_min": round(ks["scan_min_ms"], 4),
        "launches_timed": ks["n"],
        "algorithmic_bytes_per_launch": algo_bytes,
        # the same bytes over the wall time of one step of the timed region (scans pipelined,
        # epilogue and launch overheads included)
        "sustained_GBs": round(algo_bytes / dt_step / 1e9, 1),
        "sustained_frac": round(algo_bytes / dt_step / 1e9 / HBM_PEAK_GBS, 4),
        "timing": timing,
    }


def traffic_from_profiles(tag):
    p = os.path.join(ROOT, "profiles", "traffic_%s.json" % tag)
    try:
        return json.load(open(p)).get("hbm_bytes_per_launch")
    except Exception:
        return None


def valu_bound_from_profiles(tag):
    """the VALU issue bound beside the HBM one.  Wave-instructions per launch from the committed
    PMC pass (SQ_INSTS_VALU), priced at one instruction per 4 cycles and SIMD, 1024 SIMDs, 2.4 GHz."""
    p = os.path.join(ROOT, "profiles", "traffic_%s.json" % tag)
    try:
        n = json.load(open(p)).get("valu_wave_instructions_per_launch")
        return None if not n else {"wave_instructions_per_launch": n,
                                   "ms_at_2400MHz": round(n / (1024 * 0.6e9) * 1e3, 4),
                                   "source": "profiles/traffic_%s.json (rocprofv3 --pmc SQ_INSTS_VALU)" % tag}
    except Exception:
        return None


def h2d_inclusive(run, pcm, rate, true_peak):
    """Host-buffer entry (what scan_pcm_* / scan_file pay): pinned host -> HBM copy + scan + fetch.
    f32 upload, and S16 upload scanned as S16 (the grid scan.c:414 puts every input on)."""
    torch = run.torch
    sc = run.sc
    out = {}
    n = pcm.numel()
    host32 = torch.empty(pcm.shape, dtype=torch.float32).pin_memory()
    host32.copy_(pcm)
    host16 = torch.empty(pcm.shape, dtype=torch.int16).pin_memory()
    host16.copy_((pcm * 32768.0).to(torch.int16))
    dev32 = torch.empty_like(pcm)
    dev16 = torch.empty(pcm.shape, dtype=torch.int16, device=pcm.device)
    sc.set_param("overlap", 0)
    for name in ("f32", "s16"):
        # (S16: the int16 buffer is scanned where it lands -- LGD_PCM_S16, no widening pass)
        sc.plan([dev32 if name == "f32" else dev16], rate, true_peak=true_peak, album=False)
        ts = []
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.cuda.stream(run.stream):
                if name == "f32":
                    dev32.copy_(host32, non_blocking=True)
                else:
                    dev16.copy_(host16, non_blocking=True)
            sc.execute(run.stream)
            sc.fetch()
            ts.append(time.perf_counter() - t0)
        t = min(ts[1:])
        out[name] = dict(msamples_per_s=round(n / t / 1e6, 1), ms=round(t * 1e3, 3),
                         pcie_GBs=round(n * (4 if name == "f32" else 2) / t / 1e9, 1))
    out["note"] = ("pinned host buffer -> HBM -> scan -> results, one buffer at a time, no overlap of copy and "
                   "scan; PCIe-bound; never the bench `value`")
    return out


def main():
    args = parse_args()
    argv = sys.argv[1:]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1 and not os.environ.get("LGD_BENCH_CHILD"):
        sys.exit(launch_children(args.gpus, argv, args.launch_timeout))   # before anything here touches a GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")

    if args.launcher_selftest:
        import torch
        import torch.distributed as dist
        dist.init_process_group(args.backend, rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        if args.selftest_die >= 0:
            if rank == args.selftest_die:
                os._exit(3)   # dies after the rendezvous ...
            u = torch.tensor([1.0])
            dist.all_reduce(u)  # ... while its peers wait for it in the next collective
            time.sleep(3600)
        if rank == 0:
            print(json.dumps({"launcher_selftest": True, "world_size": dist.get_world_size(),
                              "backend": dist.get_backend(), "sum": float(t.item())}), flush=True)
        dist.destroy_process_group()
        return

    # Exactly ONE line may reach stdout (rank 0's JSON): libraries write banners there too (RCCL prints
    # its version when the first communicator is made), so fd 1 is pointed at stderr from here on and
    # the line goes out through a duplicate of the real stdout.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    workload = args.workload
    if workload == "auto":
        workload = "c2" if world == 1 else "c4"

    # the CPU legs first: they fork worker processes, which must not happen after GPU initialisation
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)

    import torch
    import torch.distributed as dist
    from loudgain_amd.album import DistributedAlbumScanner

    if args.same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = world > 1 or args.force_dist
    if distributed:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")   # where the few reduced scalars live

    run = Runner(args, rank, world, local_rank)
    sc = run.sc
    tracks, rates = build_tracks(args, workload, rank, world, dev)
    if args.pcm == "s16":
        tracks = [torch.round(t * 32768.0).to(torch.int16) for t in tracks]
    torch.cuda.synchronize()
    true_peak = workload != "c2"
    album = workload in ("c4", "c5")
    my_samples = sum(int(t.numel()) for t in tracks)
    algo_bytes = my_samples * 4           # 4 B read per sample, writes ~ 0
    if distributed:
        t = torch.tensor([my_samples], dtype=torch.int64, device=red_dev)
        dist.all_reduce(t)
        total_samples = int(t.item())
    else:
        total_samples = my_samples
    steps = args.steps or {"c2": 2000, "c3": 2000, "c4": max(10, 20 * world), "c5": 200}[workload]
    warmup = args.warmup if args.warmup >= 0 else {"c2": 20, "c3": 20, "c4": 3, "c5": 5}[workload]

    overlapped = not args.serial
    settle = args.settle if args.settle >= 0 else {"c2": 300, "c3": 300, "c4": 3, "c5": 20}[workload]
    launches = args.launches if args.launches > 0 else {"c2": 64, "c3": 64, "c4": 8, "c5": 32}[workload]
    ks = run.kernel_stats(tracks, rates, true_peak, album and not distributed, launches, settle)
    sc.set_param("overlap", 1 if overlapped else 0)
    if distributed:
        job = DistributedAlbumScanner(sc, tracks, rates, true_peak=true_peak, always_exchange=True)
        if not overlapped:
            sc.set_param("overlap", 0)
    else:
        job = sc.plan(tracks, rates, true_peak=true_peak, album=album)
    dt, results = run.run(job, steps, warmup, distributed)
    value = total_samples * steps / dt / 1e6
    info = sc.plan_info()

    collective = None
    if distributed:
        r1n = job.shard.rec1.numel()
        collective = {"backend": dist.get_backend(),
                      "library": "RCCL (torch.distributed 'nccl' on ROCm)" if dist.get_backend() == "nccl" else "gloo (rehearsal)",
                      "world_size": dist.get_world_size(),
                      # read back from the gathered records on the device (rank 0's album result of the last step):
                      # record-1 heads folded in stage 2, how many of them held blocks, records 2 folded in stage 3
                      "ranks_folded": None,
                      "per_step": "all_gather of album record 1 (%d doubles per rank) + all_gather of record 2 "
                                  "(2 doubles per rank), on a side stream behind the scan" % r1n,
                      "bytes_gathered_per_rank_per_step": 8 * (r1n + 2) * dist.get_world_size()}

    line = None
    if rank == 0:
        tr = results[0][0]
        kernels = ("lgd_scan_kernel" if not true_peak else
                   "lgd_scan_kernel + lgd_peak_reduce_kernel + lgd_tp_kernel (everything that reads PCM)")
        timing = ("%d serial launches before the timed region (the region itself pipelines consecutive scans on "
                  "two streams: see sustained_frac)" % launches if overlapped else "serial launches, as the timed region")
        line = {
            "metric": METRIC, "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 6),
            "higher_is_better": True, "scaling": "strong" if workload in ("c4", "c5") else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": describe(args, workload, world, distributed),
                "tracks_this_rank": len(tracks), "samples_per_step_all_ranks": total_samples,
                "chunk": info["chunk"], "segments": info["segments"], "pcm": args.pcm,
            },
            "roofline": dict(roofline_block(algo_bytes, ks, dt / steps,
                                            traffic_from_profiles(workload) if args.pcm == "f32" else None, timing, kernels),
                             valu_bound=valu_bound_from_profiles(workload)),
            "result": {"loudness": tr["loudness"], "lra": tr["lra"], "peak": tr["peak"],
                       "n_abs": tr["n_abs"], "n_rel": tr["n_rel"], "n_st": tr["n_st"]},
        }
        if workload in ("c2", "c3"):
            frames = tracks[0].shape[0]
            line["config"].update(frames_per_gpu=frames, channels=2, rate=48000,
                                  x_realtime=round(value * 1e6 / (48000 * 2), 0),
                                  mframes_per_s=round(value / 2, 1))
        if results[1] is not None:
            al = results[1] if isinstance(results[1], dict) else results[1][0]
            line["result"]["album"] = {"loudness": al["loudness"], "lra": al["lra"], "peak": al["peak"],
                                       "n_abs": al["n_abs"], "n_rel": al["n_rel"], "n_st": al["n_st"]}
        if collective:
            al_ = results[1] if isinstance(results[1], dict) else results[1][0]
            collective["ranks_folded"] = {"stage2_heads": al_["ranks_stage2"], "with_content": al_["ranks_with_content"],
                                          "stage3_records": al_["ranks_stage3"]}
            line["collective"] = collective

    # one GPU, c2: the reference's own semantics (true peak on) beside it, standard and adversarial material
    if rank == 0 and world == 1 and workload == "c2" and not args.no_c3 and not distributed and args.pcm == "f32":
        from loudgain_amd import synth
        c3steps = steps
        ks3 = run.kernel_stats(tracks, rates, True, False, 64, 300)
        sc.set_param("overlap", 1 if overlapped else 0)
        job3 = sc.plan(tracks, rates, true_peak=True, album=False)
        dt3, res3 = run.run(job3, c3steps, warmup, False)
        c3 = {"workload": describe(args, "c3", 1, False),
              "value": round(my_samples * c3steps / dt3 / 1e6, 1), "unit": "Msamples/s",
              "steps": c3steps, "ms_per_step": round(dt3 / c3steps * 1e3, 6),
              "roofline": dict(roofline_block(algo_bytes, ks3, dt3 / c3steps, traffic_from_profiles("c3"),
                                              "64 serial launches; pipelined timed region",
                                              "lgd_scan_kernel + lgd_peak_reduce_kernel + lgd_tp_kernel"),
                               valu_bound=valu_bound_from_profiles("c3")),
              "peak": res3[0][0]["peak"], "true_peak_pruning": "exact; data dependent -- see adversarial"}
        c3["dtype"] = "f64 K-filter + f32 interpolator (the reference accumulates the interpolator in double)"
        if args.material == "steps" and args.minutes >= 1:
            # the pruning is data dependent: beside the SURVEY 8d material the two that defeat it
            def other(name, pcm, what):
                ksm = run.kernel_stats([pcm], rates, True, False, 32, 100)
                ach = algo_bytes / (ksm["scan_mean_ms"] * 1e-3) / 1e9
                c3[name] = {"material": what, "kernel_ms_mean": round(ksm["scan_mean_ms"], 4),
                            "kernel_ms_min": round(ksm["scan_min_ms"], 4), "achieved": round(ach, 1),
                            "frac": round(ach / HBM_PEAK_GBS, 4)}
            noi = synth.track_torch(tracks[0].shape[0], 2, 48000, seed=0, step_s=1e9, device=dev, sine=False)
            other("noise", noi, "stationary Gaussian noise at one level (sample peak ~6 sigma): isolated peaks, where the "
                                "adjacent-pair bound prunes what L1 * max|x| cannot")
            del noi
            lim = synth.limited_torch(tracks[0].shape[0], 2, 48000, seed=0, device=dev)
            other("limited", lim, "loud, heavily limited programme: noise + tones through a hard limiter at 0.8 FS, crest "
                                  "factor ~8 dB, the sample peak is reached everywhere: (nearly) every interpolator output "
                                  "is evaluated")
            del lim
            adv = synth.adversarial_torch(tracks[0].shape[0], 2, device=dev)
            other("adversarial", adv, "constant-amplitude fs/4 sine sampled on its peaks: sample peak == true peak in every "
                                      "window, no interpolator output can be pruned")
            del adv
        line["c3"] = c3
        if args.material == "steps" and args.minutes >= 1:
            # the same samples as the interleaved S16 the reference feeds libebur128 (scan.c:442-448), resident in HBM and
            # read as they are (LGD_PCM_S16): 2 B per sample of HBM traffic; the algorithmic 4 B per sample of SURVEY.md 8d
            # stay the numerator of `frac` (its storage-format rule), `frac_of_own_bytes` prices the 2 B that are moved
            s16 = torch.round(tracks[0] * 32768.0).to(torch.int16)
            legs = {}
            for name, tp_on in (("no_true_peak", False), ("true_peak", True)):
                k16 = run.kernel_stats([s16], rates, tp_on, False, 32, 100)
                r16 = run.last_results[0][0]
                ach = algo_bytes / (k16["scan_mean_ms"] * 1e-3) / 1e9
                legs[name] = {"kernel_ms_mean": round(k16["scan_mean_ms"], 4), "kernel_ms_min": round(k16["scan_min_ms"], 4),
                              "msamples_per_s": round(my_samples / (k16["scan_mean_ms"] * 1e-3) / 1e6, 1),
                              "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4),
                              "frac_of_own_bytes": round(ach / 2 / HBM_PEAK_GBS, 4),
                              "loudness": r16["loudness"], "peak": r16["peak"]}
            line["s16"] = dict(workload="the C2 / C3 buffer as interleaved int16 in HBM (691 200 000 B for 60 min): "
                                        "lgd_scan_kernel<..., S16> / lgd_tp_kernel<..., S16>",
                               dtype="f64 K-filter on int16 samples (x / 32768, ebur128_add_frames_short)",
                               hbm_bytes_per_sample=2, **legs,
                               identical_to_f32=bool(legs["no_true_peak"]["loudness"] == tr["loudness"] and
                                                     legs["true_peak"]["peak"] == c3["peak"]))
            del s16
        line["step_ms"] = run.step_times(sc.plan(tracks, rates, true_peak=False, album=False), 20)
        if not args.no_h2d:
            line["h2d_inclusive"] = h2d_inclusive(run, tracks[0], rates[0], True)
    # one GPU, default line: config 4 too -- the workload `--gpus N` runs for N > 1 -- so that the driver's 1 / 2 / 4 / 8
    # series has its N = 1 point on the same workload (1000 tracks, 104 GB: fits one GPU's 288 GB)
    if rank == 0 and world == 1 and workload == "c2" and not args.no_c4 and not distributed and args.workload == "auto" \
            and args.pcm == "f32":
        del tracks
        sc.plan([], [], true_peak=False, album=False)   # (the engine lets go of the C2 buffer)
        torch.cuda.empty_cache()
        tr4, rt4 = build_tracks(args, "c4", 0, 1, dev)
        torch.cuda.synchronize()
        s4 = sum(int(t.numel()) for t in tr4)
        ks4 = run.kernel_stats(tr4, rt4, True, True, 8, 3)
        sc.set_param("overlap", 1 if overlapped else 0)
        job4 = sc.plan(tr4, rt4, true_peak=True, album=True)
        steps4 = 20
        dt4, res4 = run.run(job4, steps4, 3, False)
        al4 = res4[1] if isinstance(res4[1], dict) else res4[1][0]
        line["c4"] = {"workload": describe(args, "c4", 1, False), "n_gpus": 1, "scaling": "strong",
                      "value": round(s4 * steps4 / dt4 / 1e6, 1), "unit": "Msamples/s", "steps": steps4,
                      "ms_per_step": round(dt4 / steps4 * 1e3, 6), "samples_per_step": s4,
                      "segments": sc.plan_info()["segments"],
                      "roofline": roofline_block(s4 * 4, ks4, dt4 / steps4, None, "8 serial launches; pipelined timed region",
                                                 "lgd_scan_kernel + lgd_peak_reduce_kernel + lgd_tp_kernel"),
                      "album": {"loudness": al4["loudness"], "lra": al4["lra"], "peak": al4["peak"],
                                "n_abs": al4["n_abs"], "n_rel": al4["n_rel"], "n_st": al4["n_st"]},
                      "note": "the N = 1 point of the `--gpus N` series (N > 1 runs this workload, tracks dealt t mod N)"}
        del tr4, job4, res4
    if rank == 0:
        if cpu is not None:
            line["cpu_baseline"] = cpu
        real_stdout.write(json.dumps(line) + "\n")
        real_stdout.flush()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
