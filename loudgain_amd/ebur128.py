"""ctypes mirror of include/loudscan_ebur128.h: libebur128's API subset that
loudgain's scan.c calls (/root/reference/src/scan.c:102,203,294,297,303,371,383,
388,448), served by the HIP scanner.  Same names, argument meaning and error
codes as libebur128 1.2.4, so tests read like code written against ebur128.h."""
import ctypes as C

import numpy as np

from . import _lib

MODE_M = 1 << 0
MODE_S = (1 << 1) | MODE_M
MODE_I = (1 << 2) | MODE_M
MODE_LRA = (1 << 3) | MODE_S
MODE_SAMPLE_PEAK = (1 << 4) | MODE_M
MODE_TRUE_PEAK = (1 << 5) | MODE_M | MODE_SAMPLE_PEAK
MODE_ALL = MODE_S | MODE_I | MODE_LRA | MODE_SAMPLE_PEAK | MODE_TRUE_PEAK  # scan.c:204-207

SUCCESS, ERROR_NOMEM, ERROR_INVALID_MODE, ERROR_INVALID_CHANNEL_INDEX, ERROR_NO_CHANGE = range(5)

EBUR128_SYMBOLS = [
    "ebur128_get_version", "ebur128_init", "ebur128_destroy", "ebur128_add_frames_short",
    "ebur128_add_frames_float", "ebur128_loudness_global", "ebur128_loudness_global_multiple",
    "ebur128_loudness_range", "ebur128_loudness_range_multiple", "ebur128_sample_peak",
    "ebur128_true_peak", "loudscan_ebur128_set_device", "loudscan_ebur128_plan_count",
]


class Ebur128State(C.Structure):
    """ebur128_state: scan.c reads ->channels directly (scan.c:300,368)."""
    _fields_ = [("mode", C.c_int), ("channels", C.c_uint), ("samplerate", C.c_ulong), ("d", C.c_void_p)]


_bound = False


def lib():
    global _bound
    L = _lib.load()
    if not _bound:
        P = C.POINTER(Ebur128State)
        L.ebur128_get_version.argtypes = [C.POINTER(C.c_int)] * 3
        L.ebur128_get_version.restype = None
        L.ebur128_init.argtypes = [C.c_uint, C.c_ulong, C.c_int]
        L.ebur128_init.restype = P
        L.ebur128_destroy.argtypes = [C.POINTER(P)]
        L.ebur128_destroy.restype = None
        L.ebur128_add_frames_short.argtypes = [P, C.c_void_p, C.c_size_t]
        L.ebur128_add_frames_float.argtypes = [P, C.c_void_p, C.c_size_t]
        for n in ("ebur128_loudness_global", "ebur128_loudness_range"):
            getattr(L, n).argtypes = [P, C.POINTER(C.c_double)]
        for n in ("ebur128_loudness_global_multiple", "ebur128_loudness_range_multiple"):
            getattr(L, n).argtypes = [C.POINTER(P), C.c_size_t, C.POINTER(C.c_double)]
        for n in ("ebur128_sample_peak", "ebur128_true_peak"):
            getattr(L, n).argtypes = [P, C.c_uint, C.POINTER(C.c_double)]
        L.loudscan_ebur128_set_device.argtypes = [C.c_int]
        L.loudscan_ebur128_plan_count.restype = C.c_ulonglong
        _bound = True
    return L


def plan_count():
    """scans (plans) the shim has run so far in this process"""
    return int(lib().loudscan_ebur128_plan_count())


def get_version():
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    lib().ebur128_get_version(C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


class Ebur128Error(RuntimeError):
    def __init__(self, code, what):
        super().__init__("%s: libebur128 error %d" % (what, code))
        self.code = code


class State:
    """One ebur128_state (one track).  Raises Ebur128Error with libebur128's code."""

    def __init__(self, channels, samplerate, mode=MODE_ALL):
        self.L = lib()
        self.p = self.L.ebur128_init(channels, samplerate, mode)
        if not self.p:
            raise Ebur128Error(ERROR_NOMEM, "ebur128_init")  # scan.c:209 "Could not initialize EBU R128 scanner"

    @property
    def channels(self):
        return self.p.contents.channels

    def close(self):
        if getattr(self, "p", None):
            self.L.ebur128_destroy(C.byref(self.p))
            self.p = None

    def __del__(self):
        self.close()

    def add_frames(self, pcm):
        """pcm: numpy [frames, channels] int16 (ebur128_add_frames_short) or float32 (_float)."""
        a = np.ascontiguousarray(pcm)
        fn = {np.dtype(np.int16): self.L.ebur128_add_frames_short,
              np.dtype(np.float32): self.L.ebur128_add_frames_float}[a.dtype]
        rc = fn(self.p, a.ctypes.data, a.shape[0])
        if rc:
            raise Ebur128Error(rc, "ebur128_add_frames")
        return self

    def _q(self, fn, *args):
        out = C.c_double()
        rc = fn(self.p, *args, C.byref(out))
        if rc:
            raise Ebur128Error(rc, fn.__name__)
        return out.value

    def loudness_global(self):
        return self._q(self.L.ebur128_loudness_global)

    def loudness_range(self):
        return self._q(self.L.ebur128_loudness_range)

    def sample_peak(self, ch):
        return self._q(self.L.ebur128_sample_peak, ch)

    def true_peak(self, ch):
        return self._q(self.L.ebur128_true_peak, ch)


def _multi(fn, states):
    arr = (C.POINTER(Ebur128State) * len(states))(*[s.p for s in states])
    out = C.c_double()
    rc = fn(arr, len(states), C.byref(out))
    if rc:
        raise Ebur128Error(rc, fn.__name__)
    return out.value


def loudness_global_multiple(states):
    return _multi(lib().ebur128_loudness_global_multiple, states)


def loudness_range_multiple(states):
    return _multi(lib().ebur128_loudness_range_multiple, states)
