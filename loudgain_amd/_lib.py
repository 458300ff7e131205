"""Loads libloudscan_hip.so and declares the C ABI of include/loudscan_device.h."""
import ctypes as C
import os

from .build import LIB as _DEFAULT_LIB

# measurement builds only (e.g. csrc/libloudscan_hip_dbg.so with the kernel floor modes)
LIB = os.environ.get("LOUDSCAN_LIB", _DEFAULT_LIB)

_L = None


class LgdTrack(C.Structure):
    _fields_ = [("pcm", C.c_void_p), ("frames", C.c_uint64), ("channels", C.c_uint32),
                ("rate", C.c_uint32)]


class LgdTrackResult(C.Structure):
    _fields_ = [("loudness", C.c_double), ("lra", C.c_double), ("peak", C.c_double),
                ("sample_peak", C.c_double), ("true_peak", C.c_double),
                ("rel_threshold", C.c_double), ("sum_abs", C.c_double), ("sum_rel", C.c_double),
                ("n_blocks", C.c_uint64), ("n_abs", C.c_uint64), ("n_rel", C.c_uint64),
                ("n_st_blocks", C.c_uint64), ("n_st", C.c_uint64),
                ("max_momentary", C.c_double), ("max_shortterm", C.c_double)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class LgdAlbumResult(C.Structure):
    _fields_ = [("loudness", C.c_double), ("lra", C.c_double), ("peak", C.c_double),
                ("rel_threshold", C.c_double), ("sum_abs", C.c_double), ("sum_rel", C.c_double),
                ("n_abs", C.c_uint64), ("n_rel", C.c_uint64), ("n_st", C.c_uint64),
                ("ranks_stage2", C.c_uint32), ("ranks_with_content", C.c_uint32), ("ranks_stage3", C.c_uint32),
                ("reserved", C.c_uint32)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


FLAG_TRUE_PEAK = 1
FLAG_ALBUM = 2
FLAG_ALBUM_PART1 = 4

# every symbol include/loudscan_device.h declares
PCM_F32, PCM_S16 = 0, 1  # include/loudscan_device.h LGD_PCM_*

DEVICE_SYMBOLS = [
    "lgd_create", "lgd_destroy", "lgd_last_error", "lgd_set_param", "lgd_plan", "lgd_execute",
    "lgd_fetch", "lgd_album_record1", "lgd_album_record2", "lgd_album_stage2",
    "lgd_album_stage3", "lgd_copy_subblock_energies", "lgd_last_kernel_ms",
    "lgd_kernel_ms_stats", "lgd_plan_info", "lgd_album_join", "lgd_copy_channel_peaks", "lgd_plan_albums", "lgd_convert_s16", "lgd_plan_formats",
    "lgd_join", "lgd_scan_only_ms_stats",
]


def load():
    """The product path has no fallback: a missing library is a hard error."""
    global _L
    if _L is not None:
        return _L
    if not os.path.exists(LIB):
        raise RuntimeError(
            "libloudscan_hip.so is not built (%s). Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'`; there is no CPU fallback." % LIB)
    # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64
    # (same soname as /opt/rocm's).  Whichever is mapped first is shared by the
    # other only if torch comes first, so import torch before our library; a
    # second runtime in the process would see no GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB)
    vp = C.c_void_p
    L.lgd_create.restype = vp
    L.lgd_create.argtypes = [C.c_int]
    L.lgd_destroy.argtypes = [vp]
    L.lgd_destroy.restype = None
    L.lgd_last_error.restype = C.c_char_p
    L.lgd_set_param.argtypes = [vp, C.c_char_p, C.c_long]
    L.lgd_plan.argtypes = [vp, C.POINTER(LgdTrack), C.c_uint32, C.c_uint32]
    L.lgd_plan_albums.argtypes = [vp, C.POINTER(LgdTrack), C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32,
                                  C.c_uint32]
    L.lgd_convert_s16.argtypes = [vp, vp, C.c_uint64, vp]
    L.lgd_plan_formats.argtypes = [vp, C.POINTER(C.c_uint8), C.c_uint32]
    L.lgd_execute.argtypes = [vp, vp]
    L.lgd_fetch.argtypes = [vp, C.POINTER(LgdTrackResult), C.POINTER(LgdAlbumResult)]
    L.lgd_album_record1.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint64)]
    L.lgd_album_record2.argtypes = [vp, C.POINTER(vp)]
    L.lgd_album_stage2.argtypes = [vp, vp, C.c_uint32, vp]
    L.lgd_album_join.argtypes = [vp, vp]
    L.lgd_join.argtypes = [vp, vp]
    L.lgd_copy_channel_peaks.argtypes = [vp, C.c_uint32, vp, vp, C.c_uint32]
    L.lgd_album_stage3.argtypes = [vp, vp, C.c_uint32, vp]
    L.lgd_copy_subblock_energies.argtypes = [vp, C.c_uint32, vp, C.c_uint64,
                                             C.POINTER(C.c_uint64)]
    L.lgd_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.lgd_kernel_ms_stats.argtypes = [vp, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                      C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
    L.lgd_scan_only_ms_stats.argtypes = [vp, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.lgd_plan_info.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
                                C.POINTER(C.c_uint64)]
    _L = L
    return L
