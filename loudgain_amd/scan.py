"""Python mirror of loudgain's scan module interface (/root/reference/src/scan.h:35-65).

Same names, argument meaning and error behaviour as the C functions exported by
libloudscan_hip.so (include/loudscan.h); this module only binds them with ctypes.
Fatal conditions terminate the process like the reference's fail_printf does, so
tests that provoke them run in a child process.
"""
import ctypes as C

from . import _lib


class ScanResult(C.Structure):
    # field order of scan.h:35-53
    _fields_ = [
        ("file", C.c_char_p), ("container", C.c_char_p), ("codec_id", C.c_int),
        ("track_gain", C.c_double), ("track_peak", C.c_double),
        ("track_loudness", C.c_double), ("track_loudness_range", C.c_double),
        ("album_gain", C.c_double), ("album_peak", C.c_double),
        ("album_loudness", C.c_double), ("album_loudness_range", C.c_double),
        ("loudness_reference", C.c_double),
    ]


# every symbol include/loudscan.h declares
SCAN_SYMBOLS = [
    "scan_init", "scan_deinit", "scan_album_has_different_codecs",
    "scan_album_has_different_containers", "scan_album_has_opus", "scan_file",
    "scan_get_track_result", "scan_get_album_peak", "scan_set_album_result",
    "scan_get_album_result", "scan_set_device", "scan_pcm_s16", "scan_pcm_f32",
    "scan_pcm_f32_device", "scan_pcm_s16_device", "scan_set_codec", "scan_wav_probe", "scan_wav_read_s16", "scan_set_devices", "scan_get_channel_peaks",
]


class ScanWavInfo(C.Structure):
    _fields_ = [("codec_id", C.c_int), ("channels", C.c_uint), ("rate", C.c_uint), ("bits", C.c_uint),
                ("frames", C.c_size_t)]

_L = None
_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def _lib_scan():
    global _L
    if _L is None:
        L = _lib.load()
        L.scan_init.argtypes = [C.c_uint]
        L.scan_deinit.restype = None
        L.scan_file.argtypes = [C.c_char_p, C.c_uint]
        L.scan_get_track_result.restype = C.POINTER(ScanResult)
        L.scan_get_track_result.argtypes = [C.c_uint, C.c_double]
        L.scan_get_album_peak.restype = C.c_double
        L.scan_set_album_result.argtypes = [C.POINTER(ScanResult), C.c_double]
        L.scan_set_album_result.restype = None
        L.scan_get_album_result.argtypes = [C.POINTER(ScanResult), C.c_double]
        L.scan_get_album_result.restype = None
        L.scan_set_device.argtypes = [C.c_int]
        L.scan_set_devices.argtypes = [C.c_int]
        L.scan_pcm_s16.argtypes = [C.c_void_p, C.c_size_t, C.c_uint, C.c_uint, C.c_uint]
        L.scan_pcm_f32.argtypes = [C.c_void_p, C.c_size_t, C.c_uint, C.c_uint, C.c_uint]
        L.scan_pcm_f32_device.argtypes = [C.c_void_p, C.c_size_t, C.c_uint, C.c_uint, C.c_uint]
        L.scan_pcm_s16_device.argtypes = [C.c_void_p, C.c_size_t, C.c_uint, C.c_uint, C.c_uint]
        L.scan_set_codec.argtypes = [C.c_uint, C.c_int, C.c_char_p]
        L.scan_wav_probe.argtypes = [C.c_char_p, C.POINTER(ScanWavInfo)]
        L.scan_wav_read_s16.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t]
        L.scan_wav_read_s16.restype = C.c_longlong
        L.scan_get_channel_peaks.argtypes = [C.c_uint, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_uint]
        _L = L
    return _L


def scan_set_device(device):
    return _lib_scan().scan_set_device(int(device))


def scan_init(nb_files):
    return _lib_scan().scan_init(int(nb_files))


def scan_deinit():
    _lib_scan().scan_deinit()


def scan_file(path, index):
    return _lib_scan().scan_file(str(path).encode(), int(index))


def scan_pcm(pcm, rate, index):
    """numpy [frames, channels] int16 / float32 host array, or a device torch tensor."""
    L = _lib_scan()
    if hasattr(pcm, "is_cuda"):
        if not pcm.is_cuda:
            pcm = pcm.numpy()
        else:
            assert pcm.is_contiguous() and pcm.element_size() in (2, 4)
            fn = L.scan_pcm_s16_device if pcm.element_size() == 2 else L.scan_pcm_f32_device
            return fn(pcm.data_ptr(), pcm.shape[0], pcm.shape[1], int(rate), int(index))
    import numpy as np
    pcm = np.ascontiguousarray(pcm)
    fn = L.scan_pcm_s16 if pcm.dtype == np.int16 else L.scan_pcm_f32
    assert pcm.dtype in (np.int16, np.float32)
    return fn(pcm.ctypes.data, pcm.shape[0], pcm.shape[1], int(rate), int(index))


WAV_ERRORS = {-1: "Could not open input", -2: "Could not find stream info (not RIFF/WAVE)",
              -3: "Could not find the codec", -4: "Could not find audio stream"}


def scan_wav_probe(path):
    """Header of a RIFF/WAVE file: dict(codec_id, channels, rate, bits, frames); OSError otherwise."""
    wi = ScanWavInfo()
    rc = _lib_scan().scan_wav_probe(str(path).encode(), C.byref(wi))
    if rc:
        raise OSError("%s: %s" % (WAV_ERRORS.get(rc, "error %d" % rc), path))
    return dict(codec_id=wi.codec_id, channels=wi.channels, rate=wi.rate, bits=wi.bits, frames=wi.frames)


def scan_wav_read_s16(path, out_ptr, cap_frames):
    """Reads the data chunk as interleaved S16 into host memory at out_ptr; returns frames read."""
    n = _lib_scan().scan_wav_read_s16(str(path).encode(), out_ptr, int(cap_frames))
    if n < 0:
        raise OSError("%s: %s" % (WAV_ERRORS.get(int(n), "error %d" % n), path))
    return int(n)


def scan_set_codec(index, codec_id, container=None):
    return _lib_scan().scan_set_codec(int(index), int(codec_id), container.encode() if container else None)


def scan_get_channel_peaks(index):
    """(sample_peak[ch], true_peak[ch]) of a scanned file, per channel (the values scan.c:300-307 folds)."""
    sp = (C.c_double * 64)()
    tp = (C.c_double * 64)()
    n = _lib_scan().scan_get_channel_peaks(int(index), sp, tp, 64)
    if n < 0:
        raise IndexError("scan_get_channel_peaks(%d)" % index)
    return list(sp[:n]), list(tp[:n])


class _OwnedResult:
    """A scan_result malloc'd by the library; freed like loudgain.c:651 does."""

    def __init__(self, ptr):
        self._ptr = ptr
        self.contents = ptr.contents

    def __getattr__(self, name):
        return getattr(self.contents, name)

    def free(self):
        if self._ptr:
            _libc.free(C.cast(self._ptr, C.c_void_p))
            self._ptr = None

    def __del__(self):
        self.free()


def scan_get_track_result(index, pre_gain=0.0):
    p = _lib_scan().scan_get_track_result(int(index), float(pre_gain))
    return _OwnedResult(p) if p else None


def scan_get_album_peak():
    return _lib_scan().scan_get_album_peak()


def scan_set_album_result(result, pre_gain=0.0):
    _lib_scan().scan_set_album_result(result._ptr, float(pre_gain))


def scan_album_has_different_codecs():
    return _lib_scan().scan_album_has_different_codecs()


def scan_album_has_different_containers():
    return _lib_scan().scan_album_has_different_containers()


def scan_album_has_opus():
    return _lib_scan().scan_album_has_opus()
