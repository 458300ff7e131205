#!/usr/bin/env python3
"""Kernel time of the C3 scan (scan + peak-reduce + true-peak kernels, serial launches) per programme
material and true-peak kernel setting:   python tools/tp_probe.py [--minutes 60] [--materials steps,noise,...]
                                                                  [--dense-min 24,65,1] [--params a=1,b=2]
Prints one line per (material, tp_dense_min): total / scan-only / true-peak share in ms, the peak found."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from loudgain_amd import synth  # noqa: E402
from loudgain_amd.device import DeviceScanner  # noqa: E402


def material(name, frames, ch, rate, dev):
    if name == "adversarial":
        return synth.adversarial_torch(frames, ch, device=dev)
    if name == "silence":
        return torch.zeros((frames, ch), dtype=torch.float32, device=dev)
    if name == "noise":
        return synth.track_torch(frames, ch, rate, seed=0, step_s=1e9, device=dev, sine=False)
    if name == "limited":
        return synth.limited_torch(frames, ch, rate, seed=0, device=dev)
    return synth.track_torch(frames, ch, rate, seed=0, device=dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=60.0)
    ap.add_argument("--rate", type=int, default=48000)
    ap.add_argument("--channels", type=int, default=2)
    ap.add_argument("--materials", default="steps,noise,limited,adversarial,silence")
    ap.add_argument("--dense-min", default="24")
    ap.add_argument("--params", default="")
    ap.add_argument("--settle", type=int, default=150)
    ap.add_argument("--launches", type=int, default=48)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    sc = DeviceScanner(0)
    sc.set_param("timing", 2)  # the marker behind the scan kernels (scan_only_*)
    st = torch.cuda.Stream(device=dev)
    for kv in filter(None, a.params.split(",")):
        k, v = kv.split("=")
        sc.set_param(k, int(v))
    frames = int(a.minutes * 60 * a.rate)
    for m in a.materials.split(","):
        pcm = material(m, frames, a.channels, a.rate, dev)
        torch.cuda.synchronize()
        ref = None
        for dm in [int(x) for x in a.dense_min.split(",")]:
            sc.set_param("tp_dense_min", dm)
            sc.plan([pcm], a.rate, true_peak=True, album=False)
            for _ in range(a.settle):
                sc.execute(st)
            sc.fetch()
            for _ in range(a.launches):
                sc.execute(st)
            (r,), _ = sc.fetch()
            ks = sc.kernel_ms_stats(a.launches)
            sp, tp = sc.channel_peaks(0, a.channels)
            same = "" if ref is None else ("  same" if list(tp) == ref else "  DIFFERENT from first setting: %r vs %r" % (list(tp), ref))
            if ref is None:
                ref = list(tp)
            print("%-12s dense_min %2d: total %.4f ms (min %.4f)  scan %.4f  tp+reduce %.4f   peak %.6f%s" % (
                m, dm, ks["scan_mean_ms"], ks["scan_min_ms"], ks["scan_only_mean_ms"],
                ks["scan_mean_ms"] - ks["scan_only_mean_ms"], r["peak"], same), flush=True)
        del pcm
    sc.close()


if __name__ == "__main__":
    main()
