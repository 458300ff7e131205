#!/usr/bin/env python3
"""Kernel time of a C4-shaped album (tracks of 180 + (t mod 7) * 30 s, seed t) on one GPU: scan + peak-reduce + true-peak
kernels, serial launches.   python tools/c4_probe.py [--tracks 1000] [--params a=1,b=2]"""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from loudgain_amd import synth
from loudgain_amd.device import DeviceScanner
ap = argparse.ArgumentParser()
ap.add_argument("--tracks", type=int, default=1000)
ap.add_argument("--params", default="")
a = ap.parse_args()
tr = [synth.track_torch((180 + (t % 7) * 30) * 48000, 2, 48000, seed=t, device="cuda") for t in range(a.tracks)]
torch.cuda.synchronize()
ap_sets = [x for x in a.params.split(";")] if a.params else [""]
for pset in ap_sets:   # (several parameter sets on the same tracks: --params "seg_subblocks=36;seg_subblocks=24,warm_subblocks=2")
    sc = DeviceScanner(0); sc.set_param("timing", 2)
    for kv in filter(None, pset.split(",")):
        k, v = kv.split("="); sc.set_param(k, int(v))
    sc.plan(tr, 48000, true_peak=True, album=True)
    st = torch.cuda.Stream()
    for _ in range(3): sc.execute(st)
    sc.fetch()
    for _ in range(8): sc.execute(st)
    sc.fetch()
    ks = sc.kernel_ms_stats(8)
    n = sum(int(t.numel()) for t in tr)
    print(os.environ.get("LOUDSCAN_LIB", "default")[-14:], "[%s]" % pset, "tracks", a.tracks, "segments", sc.plan_info()["segments"],
          "scan+tp %.3f ms  scan %.3f  tp+reduce %.3f  frac %.4f" % (
              ks["scan_mean_ms"], ks["scan_only_mean_ms"], ks["scan_mean_ms"] - ks["scan_only_mean_ms"], n * 4 / ks["scan_mean_ms"] / 1e6 / 8000), flush=True)
    sc.close()
