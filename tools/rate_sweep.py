"""Kernel time at the common sample rates / layouts: the same number of samples as C2 (345.6 M) per
case, serial launches, hipEvents on the launch stream (scan kernels alone, and with the true-peak
kernels behind them).  Output is committed as profiles/rNN_rate_sweep.txt.
  python tools/rate_sweep.py [--chunk C] [--only 5.1]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from loudgain_amd import synth
from loudgain_amd.device import DeviceScanner

ap = argparse.ArgumentParser()
ap.add_argument("--chunk", type=int, default=0)
ap.add_argument("--only", default="")
ap.add_argument("--pcm", default="f32", choices=["f32", "s16"], help="s16: the samples as interleaved int16 in HBM (LGD_PCM_S16)")
a = ap.parse_args()
CASES = [(48000, 2), (44100, 2), (96000, 2), (192000, 2), (22050, 2), (32000, 2), (48000, 1), (44100, 1),
         (48000, 3), (48000, 4), (48000, 5), (44100, 5), (48000, 6), (44100, 6), (96000, 6), (192000, 6),
         (48000, 7), (48000, 8), (48000, 12), (48000, 24)]
if a.only == "5.1":
    CASES = [c for c in CASES if c[1] == 6]
elif a.only == "multi":
    CASES = [c for c in CASES if c[1] > 2]
print("PCM resident as %s" % a.pcm)
print("%-7s %-3s %-5s %-6s | %-28s | %-28s" % ("rate", "ch", "chunk", "segs", "no true peak: ms, % of 8 TB/s",
                                               "true peak: scan ms, scan+tp ms, %"))
for rate, ch in CASES:
    frames = int(172800000 * 2 / ch)
    pcm = synth.track_torch(frames, ch, rate, seed=1, device="cuda")
    if a.pcm == "s16":
        pcm = torch.round(pcm * 32768.0).to(torch.int16)
    nbytes = frames * ch * 4  # (algorithmic: 4 B per sample whatever the element format)
    row = []
    for tp, timing in ((False, 1), (True, 1), (True, 2)):
        sc = DeviceScanner(0)
        # (the marker behind the scan kernels -- "timing" 2, for scan_only_* -- costs ~5 us inside the bracket: the totals are
        # taken without it, the split of the true-peak case in a third run)
        sc.set_param("timing", timing)
        sc.set_param("overlap", 0)
        if a.chunk:
            sc.set_param("chunk", a.chunk)
        sc.plan([pcm], rate, true_peak=tp)
        s = torch.cuda.Stream()
        for _ in range(150):  # (clocks and caches settle: 20 launches read 10 % slow)
            sc.execute(s)
        sc.fetch()
        for _ in range(40):
            sc.execute(s)
        sc.fetch()
        ks = sc.kernel_ms_stats(40)
        info = sc.plan_info()
        row.append((ks, info))
        sc.close()
    (k0, info), (k1, _), (k2, _) = row
    k1 = dict(k1, scan_only_mean_ms=k2["scan_only_mean_ms"])
    print("%-7d %-3d %-5d %-6d | %7.4f ms  %5.1f %%           | %7.4f  %7.4f ms  %5.1f %%" % (
        rate, ch, info["chunk"], info["segments"], k0["scan_mean_ms"], nbytes / k0["scan_mean_ms"] / 1e6 / 80.0,
        k1["scan_only_mean_ms"], k1["scan_mean_ms"], nbytes / k1["scan_mean_ms"] / 1e6 / 80.0), flush=True)
    del pcm
    torch.cuda.empty_cache()
