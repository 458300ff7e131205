"""ctypes binding of the CPU oracle (oracle/liblgoracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (loudgain_amd) never imports
this module.  PARITY UNPINNED by reference fixtures -- see oracle/lg_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class ScanResult(C.Structure):
    # field order of /root/reference/src/scan.h:35-53
    _fields_ = [
        ("file", C.c_char_p), ("container", C.c_char_p), ("codec_id", C.c_int),
        ("track_gain", C.c_double), ("track_peak", C.c_double),
        ("track_loudness", C.c_double), ("track_loudness_range", C.c_double),
        ("album_gain", C.c_double), ("album_peak", C.c_double),
        ("album_loudness", C.c_double), ("album_loudness_range", C.c_double),
        ("loudness_reference", C.c_double),
    ]


def build(force=False):
    so = os.path.join(_HERE, "liblgoracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("lg_oracle.c", "lg_scan_oracle.c", "lg_oracle.h")]
    if force or not os.path.exists(so) or any(
            os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    L = C.CDLL(build())
    vp, dp, sz = C.c_void_p, C.POINTER(C.c_double), C.c_size_t
    L.lgo_create.restype = vp
    L.lgo_create.argtypes = [C.c_uint, C.c_ulong]
    L.lgo_destroy.argtypes = [vp]
    L.lgo_add_frames_short.argtypes = [vp, vp, sz]
    L.lgo_add_frames_float.argtypes = [vp, vp, sz]
    for n in ("lgo_loudness_global", "lgo_loudness_range"):
        getattr(L, n).argtypes = [vp, dp]
    for n in ("lgo_loudness_global_multiple", "lgo_loudness_range_multiple"):
        getattr(L, n).argtypes = [C.POINTER(vp), sz, dp]
    L.lgo_true_peak.argtypes = [vp, C.c_uint, dp]
    L.lgo_sample_peak.argtypes = [vp, C.c_uint, dp]
    L.lgo_channels.argtypes = [vp]
    L.lgo_channels.restype = C.c_uint
    for n in ("lgo_gating_block_count", "lgo_shortterm_block_count"):
        getattr(L, n).argtypes = [vp]
        getattr(L, n).restype = sz
    for n in ("lgo_gating_blocks", "lgo_shortterm_blocks"):
        getattr(L, n).argtypes = [vp]
        getattr(L, n).restype = dp
    L.lgo_gating_detail.argtypes = [C.POINTER(vp), sz, C.POINTER(sz), dp, dp, C.POINTER(sz), dp]
    L.lgo_design_filter.argtypes = [C.c_ulong, dp, dp]
    L.lgo_design_interp.argtypes = [C.c_ulong, C.POINTER(C.c_uint), C.POINTER(C.c_uint),
                                    C.POINTER(C.c_uint), dp]
    L.lgo_scan_init.argtypes = [C.c_uint]
    L.lgo_scan_file.argtypes = [C.c_char_p, C.c_uint]
    L.lgo_scan_pcm_s16.argtypes = [vp, sz, C.c_uint, C.c_ulong, C.c_uint]
    L.lgo_scan_pcm_f32.argtypes = [vp, sz, C.c_uint, C.c_ulong, C.c_uint]
    L.lgo_scan_get_track_result.restype = C.POINTER(ScanResult)
    L.lgo_scan_get_track_result.argtypes = [C.c_uint, C.c_double]
    L.lgo_scan_get_album_peak.restype = C.c_double
    L.lgo_scan_set_album_result.argtypes = [C.POINTER(ScanResult), C.c_double]
    L.lgo_scan_state.restype = vp
    L.lgo_scan_state.argtypes = [C.c_uint]
    _LIB = L
    return L


def design_filter(rate):
    b = (C.c_double * 5)()
    a = (C.c_double * 5)()
    lib().lgo_design_filter(rate, b, a)
    return np.array(b), np.array(a)


def design_interp(rate):
    """-> (factor, delay, [(index[], coeff[]) per phase])"""
    delay = C.c_uint()
    count = (C.c_uint * 4)()
    index = (C.c_uint * 100)()
    coeff = (C.c_double * 100)()
    f = lib().lgo_design_interp(rate, C.byref(delay), count, index, coeff)
    phases = []
    for p in range(f):
        n = count[p]
        phases.append((np.array(index[p * 25:p * 25 + n]), np.array(coeff[p * 25:p * 25 + n])))
    return f, delay.value, phases


class State:
    """One ebur128_state as scan.c:203-207 creates it (all five modes on)."""

    def __init__(self, channels, rate):
        self.L = lib()
        self.h = self.L.lgo_create(channels, rate)
        if not self.h:
            raise ValueError("lgo_create failed")
        self.channels, self.rate = channels, rate

    def close(self):
        if self.h:
            self.L.lgo_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def add(self, pcm, chunk=None):
        """pcm: [frames, channels] int16 or float32, C-contiguous."""
        pcm = np.ascontiguousarray(pcm)
        if pcm.ndim == 1:
            pcm = pcm.reshape(-1, self.channels)
        assert pcm.shape[1] == self.channels
        fn = {np.dtype(np.int16): self.L.lgo_add_frames_short,
              np.dtype(np.float32): self.L.lgo_add_frames_float}[pcm.dtype]
        n = pcm.shape[0]
        step = chunk or n or 1
        for off in range(0, n, step):
            part = pcm[off:off + step]
            fn(self.h, part.ctypes.data, part.shape[0])
        return self

    def loudness(self):
        out = C.c_double()
        self.L.lgo_loudness_global(self.h, C.byref(out))
        return out.value

    def lra(self):
        out = C.c_double()
        self.L.lgo_loudness_range(self.h, C.byref(out))
        return out.value

    def true_peak(self, ch):
        out = C.c_double()
        self.L.lgo_true_peak(self.h, ch, C.byref(out))
        return out.value

    def sample_peak(self, ch):
        out = C.c_double()
        self.L.lgo_sample_peak(self.h, ch, C.byref(out))
        return out.value

    def peak(self):
        return max(self.true_peak(c) for c in range(self.channels))

    def gating_blocks(self):
        n = self.L.lgo_gating_block_count(self.h)
        return np.ctypeslib.as_array(self.L.lgo_gating_blocks(self.h), (n,)).copy() if n else np.zeros(0)

    def shortterm_blocks(self):
        n = self.L.lgo_shortterm_block_count(self.h)
        return np.ctypeslib.as_array(self.L.lgo_shortterm_blocks(self.h), (n,)).copy() if n else np.zeros(0)


def _handles(states):
    arr = (C.c_void_p * len(states))(*[s.h for s in states])
    return arr


def gating_detail(states):
    n_abs, n_rel = C.c_size_t(), C.c_size_t()
    s_abs, thr, s_rel = C.c_double(), C.c_double(), C.c_double()
    lib().lgo_gating_detail(_handles(states), len(states), C.byref(n_abs), C.byref(s_abs),
                            C.byref(thr), C.byref(n_rel), C.byref(s_rel))
    return dict(n_abs=n_abs.value, sum_abs=s_abs.value, rel_threshold=thr.value,
                n_rel=n_rel.value, sum_rel=s_rel.value)


def album_loudness(states):
    out = C.c_double()
    lib().lgo_loudness_global_multiple(_handles(states), len(states), C.byref(out))
    return out.value


def album_lra(states):
    out = C.c_double()
    lib().lgo_loudness_range_multiple(_handles(states), len(states), C.byref(out))
    return out.value


def scan_track(pcm, rate):
    """Full per-track answer the GPU path is compared against."""
    pcm = np.ascontiguousarray(pcm)
    st = State(pcm.shape[1], rate).add(pcm, chunk=4096)
    d = gating_detail([st])
    return dict(loudness=st.loudness(), lra=st.lra(), peak=st.peak(),
                true_peak=[st.true_peak(c) for c in range(st.channels)],
                sample_peak=[st.sample_peak(c) for c in range(st.channels)],
                n_st=len(st.shortterm_blocks()), state=st, **d)
