"""Segment length for small (rate, channels) groups (C5-shaped: a handful of 120 s tracks): scan kernel
time against "seg_subblocks" (0 = the planner's choice)."""
import os, sys
sys.path.insert(0, "/root/repo")
import torch
from loudgain_amd import synth
from loudgain_amd.device import DeviceScanner
cases = [(int(r), int(c)) for r, c in (x.split("x") for x in os.environ.get("PROBE_CASES", "192000x1,192000x2,96000x1,48000x1,44100x1,48000x2,44100x2,48000x6").split(","))]
ntr = int(os.environ.get("PROBE_TRACKS", 5))
tp = bool(int(os.environ.get("PROBE_TP", 1)))
for rate, ch in cases:
    tracks = [synth.track_torch(rate * 120, ch, rate, seed=7 + i, device="cuda") for i in range(ntr)]
    nbytes = sum(t.numel() for t in tracks) * 4
    out = []
    for seg in [int(x) for x in os.environ.get("PROBE_SEGS", "0,2,3,4,5,6,8,12").split(",")]:
        sc = DeviceScanner(0); sc.set_param("overlap", 0); sc.set_param("timing", 2)
        if seg: sc.set_param("seg_subblocks", seg)
        sc.plan(tracks, rate, true_peak=tp and rate < 192000)
        s = torch.cuda.Stream()
        for _ in range(150): sc.execute(s)
        sc.fetch()
        for _ in range(40): sc.execute(s)
        sc.fetch()
        ks = sc.kernel_ms_stats(40); info = sc.plan_info()
        out.append("seg %2d: %4d segs %.4f ms (%.0f %%)" % (seg, info["segments"], ks["scan_only_mean_ms"], nbytes / ks["scan_only_mean_ms"] / 1e6 / 80.0))
        sc.close()
    print(rate, ch, " | ".join(out), flush=True)
    del tracks; torch.cuda.empty_cache()
