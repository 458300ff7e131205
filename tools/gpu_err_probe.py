"""Probe: GPU sub-block energies, and the oracle's (= libebur128's double DF-II) 400 ms
block energies, against a long-double evaluation of the same filter.
    gpu_err_probe.py [rate] [noise|step|tone]"""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from oracle import lgoracle as o
from loudgain_amd.device import DeviceScanner

def ref_ld(x, a, b):
    al = a.astype(np.longdouble); bl = b.astype(np.longdouble)
    v = np.zeros(5, np.longdouble); y = np.zeros(len(x), np.longdouble)
    for n in range(len(x)):
        v[0] = x[n] - al[1]*v[1] - al[2]*v[2] - al[3]*v[3] - al[4]*v[4]
        y[n] = bl[0]*v[0] + bl[1]*v[1] + bl[2]*v[2] + bl[3]*v[3] + bl[4]*v[4]
        v[4] = v[3]; v[3] = v[2]; v[2] = v[1]; v[1] = v[0]
    return y

fs = int(sys.argv[1]) if len(sys.argv) > 1 else 48000
kind = sys.argv[2] if len(sys.argv) > 2 else "noise"
N = fs * 4
rng = np.random.default_rng(0)
t = np.arange(N) / fs
if kind == "noise":
    x = rng.standard_normal(N) * 0.1
elif kind == "tone":
    x = 0.7 * np.sin(2 * np.pi * 37.0 * t)
else:
    x = rng.standard_normal(N) * 0.1 * np.where(t < 2, 1.0, 1e-4) + np.where(t < 2, 0.9*np.sin(2*np.pi*30*t), 0)
x = (np.clip(np.round(x * 32768), -32768, 32767) / 32768).astype(np.float32)
b, a = o.design_filter(fs)
y = ref_ld(x.astype(np.float64), a, b)
s100 = (fs + 5) // 10
nsb = N // s100
Eld = np.array([float((y[k*s100:(k+1)*s100]**2).sum()) for k in range(nsb)])
st = o.State(1, fs).add(x.reshape(-1, 1))
Zld = (Eld[:-3] + Eld[1:-2] + Eld[2:-1] + Eld[3:]) / (4.0 * s100)
Zo = st.gating_blocks()
if len(Zo) == len(Zld):
    print("fs", fs, kind, "ORACLE (double DF-II) vs long double, 400 ms blocks: max rel", (np.abs(Zo - Zld) / Zld).max())
for seg in (1000000, 3):
    for chunk in (0, 25, 75):
        s = DeviceScanner(0)
        s.set_param("seg_subblocks", seg); s.set_param("chunk", chunk)
        try:
            s.scan([torch.from_numpy(x.reshape(-1, 1)).cuda()], fs)
        except Exception as e:
            print("skip", seg, chunk, e); continue
        E = s.subblock_energies(0)
        rel = np.abs(E - Eld) / Eld
        print("fs", fs, kind, "seg", seg, "chunk", s.plan_info()["chunk"], "max rel", rel.max(), "median", np.median(rel),
              "first8", np.array2string(rel[:8], precision=2))
        s.close()
