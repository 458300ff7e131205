"""Does a scan on the engine's second stream wait for work enqueued earlier on the caller's stream?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from loudgain_amd import synth
from loudgain_amd.device import DeviceScanner
dev = torch.device("cuda", 0)
rate, ch = 48000, 2
real = synth.track_torch(20 * rate, ch, rate, seed=3, device=dev)
sc = DeviceScanner(0)
want = sc.scan([real], rate, true_peak=False)[0][0]
pcm = torch.zeros_like(real)
sc.plan([pcm], rate, true_peak=False)
s = torch.cuda.Stream()
big = torch.randn(8192, 8192, device=dev)
for trial in range(4):
    pcm.zero_(); torch.cuda.synchronize()
    n_exec = trial + 1           # trial 0: last execute is k=0 (caller stream); trial 1: k=1 (second stream) ...
    with torch.cuda.stream(s):
        for _ in range(n_exec - 1):
            sc.execute(s)
        for _ in range(10):
            big = big @ big * 1e-4   # ~ms of work on the caller's stream
        pcm.copy_(real)
        sc.execute(s)
    got = sc.fetch()[0][0]
    print("executes", n_exec, "loudness", got["loudness"], "want", want["loudness"], "OK" if abs(got["loudness"] - want["loudness"]) < 1e-9 else "STALE INPUT")
