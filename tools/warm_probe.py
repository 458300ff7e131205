"""How much does the K-filter warm-up length matter? Compare block energies for warm = 1..5
sub-blocks against warm = 8 on material with 0 dBFS LF content followed by -80 dB noise."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from loudgain_amd import synth
from loudgain_amd.device import DeviceScanner
fs = 48000
t = np.arange(fs * 6) / fs
rng = np.random.default_rng(1)
x = np.concatenate([0.95 * np.sin(2 * np.pi * 25 * t), rng.standard_normal(fs * 6) * 1e-4,
                    rng.standard_normal(fs * 6) * 0.2, rng.standard_normal(fs * 6) * 1e-3])
pcm = synth.snap_s16_numpy(np.stack([x, x * 0.7], 1))
dev = torch.from_numpy(pcm).cuda()
def run(warm):
    s = DeviceScanner(0); s.set_param("seg_subblocks", 1); s.set_param("warm_subblocks", warm)
    s.scan([dev], fs); e = s.subblock_energies(0); s.close(); return e
ref = run(8)
for w in (1, 2, 3, 4, 5):
    e = run(w)
    rel = np.abs(e - ref) / ref
    print("warm", w, "max rel diff vs warm 8:", rel.max(), "at sub-block", int(rel.argmax()))
