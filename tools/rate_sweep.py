"""Kernel time per sample at the common sample rates / layouts (serial launches, hipEvents)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from loudgain_amd import synth
from loudgain_amd.device import DeviceScanner
for rate, ch, tp in [(48000, 2, False), (44100, 2, False), (44100, 2, True), (48000, 2, True), (96000, 2, False), (96000, 2, True),
                     (192000, 2, True), (22050, 2, False), (32000, 2, False), (48000, 1, False), (44100, 1, True), (48000, 6, True)]:
    frames = int(172800000 * 2 / ch)          # same number of samples as C2
    frames -= frames % (rate // 10) if False else 0
    pcm = synth.track_torch(frames, ch, rate, seed=1, device="cuda")
    sc = DeviceScanner(0)
    sc.set_param("overlap", 0)
    sc.plan([pcm], rate, true_peak=tp)
    s = torch.cuda.Stream()
    for _ in range(3): sc.execute(s)
    sc.fetch()
    for _ in range(30): sc.execute(s)
    sc.fetch()
    ks = sc.kernel_ms_stats(30)
    info = sc.plan_info()
    nbytes = frames * ch * 4
    print("rate %6d ch %d tp %d: chunk %2d segs %5d kernel %.4f ms  %.1f GB/s (%.1f %% of 8 TB/s)  %.0f Msamples/s" % (
        rate, ch, tp, info["chunk"], info["segments"], ks["scan_mean_ms"], nbytes / ks["scan_mean_ms"] / 1e6,
        nbytes / ks["scan_mean_ms"] / 1e6 / 80.0, frames * ch / ks["scan_mean_ms"] / 1e3), flush=True)
    sc.close(); del pcm; torch.cuda.empty_cache()
