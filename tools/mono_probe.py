import os, sys
sys.path.insert(0, "/root/repo")
import torch
N_WARM = int(os.environ.get('PROBE_WARM', 100)); N_RUN = int(os.environ.get('PROBE_RUN', 40))
LAYOUTS = [(int(os.environ.get("PROBE_RATE", 48000)), int(c)) for c in os.environ.get('PROBE_CH', '1,2').split(',')]
ORDER = [x == 'T' for x in (sys.argv[1] if len(sys.argv) > 1 else 'FT')]
from loudgain_amd import synth
from loudgain_amd.device import DeviceScanner
for rate, ch in LAYOUTS:
    frames = int(172800000 * 2 / ch)
    pcm = synth.track_torch(frames, ch, rate, seed=1, device="cuda")
    for tp in ORDER:
        sc = DeviceScanner(0); sc.set_param("overlap", 0); sc.set_param("timing", 2); [sc.set_param(k, int(v)) for k, v in (kv.split("=") for kv in os.environ.get("PROBE_PARAMS", "").split(",") if kv)]; sc.plan([pcm], rate, true_peak=tp)
        s = torch.cuda.Stream()
        for _ in range(N_WARM): sc.execute(s)
        sc.fetch()
        for _ in range(N_RUN): sc.execute(s)
        sc.fetch()
        ks = sc.kernel_ms_stats(N_RUN)
        print(os.environ.get("LOUDSCAN_LIB", "base")[-16:], ch, tp, round(ks["scan_only_mean_ms"], 4), round(ks["scan_only_min_ms"], 4), flush=True)
        sc.close()
