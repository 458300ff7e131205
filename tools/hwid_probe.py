"""Where do the scan kernel's waves land?  Needs the placement-probe build, in which every wave leaves HW_ID,
XCC_ID and its start / end time in its channel's first chunk-maxima row:
    make -C loudgain_amd/csrc libloudscan_hip_hwid.so
    LOUDSCAN_LIB=$PWD/loudgain_amd/csrc/libloudscan_hip_hwid.so PROBE_CH=1 python tools/hwid_probe.py
Prints waves per SIMD, workgroups per CU, start skew and wave duration by SIMD load (wave placement)."""
import os, sys, ctypes, collections
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from loudgain_amd import synth, _lib
from loudgain_amd.device import DeviceScanner
rate, ch = 48000, int(os.environ.get("PROBE_CH", 1))
frames = 172800000 * 2 // ch
pcm = synth.track_torch(frames, ch, rate, seed=1, device="cuda")
sc = DeviceScanner(0); sc.set_param("overlap", 0); sc.set_param("timing", 2)
[sc.set_param(k, int(v)) for k, v in (kv.split("=") for kv in os.environ.get("PROBE_PARAMS", "").split(",") if kv)]
sc.plan([pcm], rate, true_peak=True)
s = torch.cuda.Stream()
for _ in range(30): sc.execute(s)
sc.fetch()
ks = sc.kernel_ms_stats(20)
lib = _lib.load()
info = sc.plan_info()
n_seg = int(info["segments"])
wg_ch = int(os.environ.get("PROBE_WG_CH", 3 if ch == 6 else (2 if ch in (5, 7) or ch > 16 else ch)))  # waves per workgroup (5.1: triples)
n_sets = -(-ch // wg_ch) if wg_ch != ch else 1   # workgroup sets per segment (pairs / triples of a wider stream)
seg_sb = -(-int(info["subblocks"]) * n_sets // n_seg)
tiles = -(-seg_sb * (rate // 10) // (64 * int(info["chunk"])))
per = -(-tiles // 8) * 1024 * wg_ch
print("segments", n_seg, "sub-blocks each", seg_sb, "chunk", info["chunk"], "tiles", tiles)
buf = np.zeros(n_seg * per // 4, np.uint32)
lib.lgd_debug_rows.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
rc = lib.lgd_debug_rows(sc.ctx, buf.ctypes.data, buf.nbytes)
d = buf.reshape(n_seg, per // 4)[:, :4 * wg_ch].reshape(n_seg * wg_ch, 4)
hw, xcc, t0, t1 = d[:, 0], d[:, 1] & 15, d[:, 2].astype(np.int64), d[:, 3].astype(np.int64)
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
print("skip_tp", bool(os.environ.get("LGD_SKIP_TP")), "scan_only", round(ks["scan_only_mean_ms"], 4), "rc", rc)
key = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist(), simd.tolist()))
per_simd = collections.Counter(key.values()); print("waves per SIMD histogram", sorted(per_simd.items()))
cuk = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
print("CUs used", len(cuk), "WGs per CU histogram", sorted(collections.Counter(cuk.values()).items()))
print("per XCC", sorted(collections.Counter(xcc.tolist()).items()))
dur = ((t1 - t0) % (1 << 32)) / 100.0
st = ((t0 - t0.min()) % (1 << 32)) / 100.0
print("start us: p50 %.1f p90 %.1f max %.1f | duration us: min %.1f p50 %.1f p90 %.1f max %.1f" % (
    np.median(st), np.percentile(st, 90), st.max(), dur.min(), np.median(dur), np.percentile(dur, 90), dur.max()))
# duration by number of waves sharing the SIMD
share = np.array([key[k] for k in zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist(), simd.tolist())])
for n in sorted(set(share.tolist())):
    print("  waves on the SIMD", n, "count", int((share == n).sum()), "mean duration us %.1f" % dur[share == n].mean())
late = st > 20
print("late starters", int(late.sum()), "mean duration %.1f" % (dur[late].mean() if late.any() else 0))
