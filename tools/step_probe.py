"""Sustained step time of back-to-back scans of one plan (the bench's timed region) under the
engine's stream / event options."""
import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from loudgain_amd import synth
from loudgain_amd.device import DeviceScanner
rate, ch = 48000, 2
tp = bool(int(os.environ.get("PROBE_TP", 0)))
pcm = synth.track_torch(172800000, ch, rate, seed=1, device="cuda")
for overlap in (0, 1):
    for timing in (1, 0):
        sc = DeviceScanner(0); sc.set_param("overlap", overlap); sc.set_param("timing", timing)
        [sc.set_param(k, int(v)) for k, v in (kv.split("=") for kv in os.environ.get("PROBE_PARAMS", "").split(",") if kv)]
        job = sc.plan([pcm], rate, true_peak=tp)
        s = torch.cuda.Stream()
        for _ in range(300): job.execute(s)
        job.fetch(); torch.cuda.synchronize()
        best = []
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(1000): job.execute(s)
            job.fetch(); torch.cuda.synchronize()
            best.append((time.perf_counter() - t0))
        if timing:
            ks = sc.kernel_ms_stats(40)
            print("   last 40 launches: scan_only mean %.4f min %.4f | scan..tp mean %.4f" % (ks["scan_only_mean_ms"], ks["scan_only_min_ms"], ks["scan_mean_ms"]))
        print("skip_epi", os.environ.get("LGD_SKIP_EPI", "0"), "tp", tp, "overlap", overlap, "timing", timing,
              "ms/step", [round(b, 4) for b in best], flush=True)
        sc.close()
