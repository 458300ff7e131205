"""S16-resident PCM against f32 (same samples): kernel time of the scan, C2 / C3 size.  PROBE_RATE, PROBE_CH, PROBE_MAT."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from loudgain_amd import synth
from loudgain_amd.device import DeviceScanner
rate = int(os.environ.get("PROBE_RATE", "48000"))
for ch in [int(c) for c in os.environ.get("PROBE_CH", "2").split(",")]:
    frames = int(172800000 * 2 / ch * rate / 48000) if rate < 48000 else int(172800000 * 2 / ch)
    mat = os.environ.get("PROBE_MAT", "steps")
    pcm = (synth.limited_torch if mat == "limited" else synth.track_torch)(frames, ch, rate, seed=1, device="cuda")
    s16 = torch.round(pcm * 32768.0).to(torch.int16)
    for name, buf in (("f32", pcm), ("s16", s16)):
        out = []
        for tp in (False, True):
            sc = DeviceScanner(0); sc.set_param("overlap", 0)
            for kv in [x for x in os.environ.get("PROBE_PARAMS", "").split(",") if x]:
                sc.set_param(kv.split("=")[0], int(kv.split("=")[1]))
            sc.plan([buf], rate, true_peak=tp)
            s = torch.cuda.Stream()
            for _ in range(100): sc.execute(s)
            sc.fetch()
            for _ in range(30): sc.execute(s)
            (r,), _ = sc.fetch()
            ks = sc.kernel_ms_stats(30)
            info = sc.plan_info()["segments"], sc.plan_info()["chunk"]
            out.append("%.4f ms %4.1f %% (%.6f LUFS, peak %.6f)" % (ks["scan_mean_ms"], frames * ch * 4 / ks["scan_mean_ms"] / 1e6 / 80.0, r["loudness"], r["peak"]))
            sc.close()
        print(os.environ.get("PROBE_PARAMS", ""), rate, ch, "ch", mat, name, "segs", info, "| no tp", out[0], "| tp", out[1], flush=True)
    del pcm, s16
