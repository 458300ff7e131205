import os, sys
sys.path.insert(0, os.getcwd())
import torch
from loudgain_amd import synth
from loudgain_amd.device import DeviceScanner
for ch in [int(c) for c in os.environ.get("PROBE_CH", "7").split(",")]:
    frames = int(172800000 * 2 / ch)
    pcm = synth.track_torch(frames, ch, 48000, seed=1, device="cuda")
    for st in [int(v) for v in os.environ.get("PROBE_STRIDED", "1,4").split(",")]:
        out = []
        for tp in (False, True):
            sc = DeviceScanner(0); sc.set_param("overlap", 0); sc.set_param("strided", st)
            sc.plan([pcm], 48000, true_peak=tp)
            s = torch.cuda.Stream()
            for _ in range(100): sc.execute(s)
            sc.fetch()
            for _ in range(30): sc.execute(s)
            (r,), _ = sc.fetch()
            ks = sc.kernel_ms_stats(30)
            out.append("%.4f ms %4.1f %%" % (ks["scan_mean_ms"], frames * ch * 4 / ks["scan_mean_ms"] / 1e6 / 80.0))
            sc.close()
        print(ch, "strided", st, "chunk", sc.plan_info()["chunk"] if False else "", "| no tp", out[0], "| tp", out[1], flush=True)
    del pcm
