"""7 / 5 channel layouts: scan-kernel time of 345.6 M samples under the channel-set options of the planner
("strided" 1 = default, 2 = pairs, 0 = planar / run-time-channel kernel)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from loudgain_amd import synth
from loudgain_amd.device import DeviceScanner
for ch in [int(c) for c in os.environ.get("PROBE_CH", "7,5").split(",")]:
    frames = int(172800000 * 2 / ch)
    pcm = synth.track_torch(frames, ch, 48000, seed=1, device="cuda")
    for params in ({}, {"strided": 2}):
        sc = DeviceScanner(0); sc.set_param("overlap", 0); sc.set_param("timing", 2)
        for k, v in params.items(): sc.set_param(k, v)
        sc.plan([pcm], 48000, true_peak=False)
        s = torch.cuda.Stream()
        for _ in range(100): sc.execute(s)
        sc.fetch()
        for _ in range(30): sc.execute(s)
        (r,), _ = sc.fetch()
        ks = sc.kernel_ms_stats(30)
        print(os.environ.get("LOUDSCAN_LIB", "default")[-12:], ch, params, sc.plan_info()["segments"], sc.plan_info()["chunk"], round(ks["scan_only_mean_ms"], 4), r["loudness"], flush=True)
        sc.close()
    del pcm
