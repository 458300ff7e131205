import os, sys
sys.path.insert(0, "/root/repo")
import torch
from loudgain_amd import synth
from loudgain_amd.device import DeviceScanner
for rate, ch, tp, wpc in [(48000, 3, True, 8), (48000, 4, True, 8), (48000, 5, True, 8), (44100, 5, True, 8), (48000, 6, True, 8), (44100, 6, True, 8), (48000, 7, True, 8), (48000, 8, True, 8), (48000, 12, True, 8), (48000, 24, True, 8)]:
    frames = int(172800000 * 2 / ch)
    pcm = synth.track_torch(frames, ch, rate, seed=1, device="cuda")
    sc = DeviceScanner(0); sc.set_param("overlap", 0); sc.set_param("waves_per_cu", wpc); sc.plan([pcm], rate, true_peak=tp)
    s = torch.cuda.Stream()
    for _ in range(3): sc.execute(s)
    sc.fetch()
    for _ in range(20): sc.execute(s)
    sc.fetch()
    ks = sc.kernel_ms_stats(20); info = sc.plan_info(); nbytes = frames * ch * 4
    print("wpc %2d rate %6d ch %d tp %d: chunk %2d segs %5d kernel %.4f ms  %.1f %% of 8 TB/s" % (wpc, rate, ch, tp, info["chunk"], info["segments"], ks["scan_mean_ms"], nbytes / ks["scan_mean_ms"] / 1e6 / 80.0), flush=True)
    sc.close(); del pcm; torch.cuda.empty_cache()
