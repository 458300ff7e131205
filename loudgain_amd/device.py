"""ctypes binding of the device-level C ABI (include/loudscan_device.h).

PyTorch is plumbing only here: it owns the HBM buffers and the HIP stream that
the C ABI is handed as plain pointers.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (FLAG_ALBUM, FLAG_ALBUM_PART1, FLAG_TRUE_PEAK, LgdAlbumResult, LgdTrack,
                   LgdTrackResult)


class LoudscanError(RuntimeError):
    pass


def _stream_handle(stream):
    if stream is None:
        return None
    if isinstance(stream, int):
        return stream
    return stream.cuda_stream  # torch.cuda.Stream


class DeviceScanner:
    """One lgd_ctx: plan -> execute (async) -> fetch."""

    def __init__(self, device=0):
        self.L = _lib.load()
        self.ctx = self.L.lgd_create(int(device))
        if not self.ctx:
            raise LoudscanError(self.L.lgd_last_error().decode())
        self.device = int(device)
        self._keep = None
        self.n_tracks = 0
        self.n_albums = None
        self.flags = 0

    def close(self):
        if getattr(self, "ctx", None):
            self.L.lgd_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        self.close()

    def _chk(self, rc):
        if rc != 0:
            raise LoudscanError("%s (code %d)" % (self.L.lgd_last_error().decode(), rc))

    def set_param(self, name, value):
        self._chk(self.L.lgd_set_param(self.ctx, name.encode(), int(value)))

    def plan(self, tracks, rates, true_peak=True, album=False, albums=None):
        """tracks: device torch tensors [frames, channels] float32 (or int16: mono / stereo, read as they
        are by the S16 kernels) contiguous, or (ptr, frames, channels[, LGD_PCM_*]) tuples; rates: int or list.
        albums: optional album index per track (non-decreasing): one launch scans all
        tracks and reduces every album; fetch() then returns a list of album results."""
        if isinstance(rates, int):
            rates = [rates] * len(tracks)
        arr = (LgdTrack * max(1, len(tracks)))()
        fmts = (C.c_uint8 * max(1, len(tracks)))()
        for i, (t, r) in enumerate(zip(tracks, rates)):
            if isinstance(t, tuple):
                ptr, frames, ch = t[:3]
                fmts[i] = int(t[3]) if len(t) > 3 else _lib.PCM_F32
            else:
                # int16 tensors are the reference's own feed (ebur128_add_frames_short, scan.c:448): read as they are
                s16 = t.dtype.is_floating_point is False and t.element_size() == 2 and t.dtype.is_signed
                f32 = t.dtype.is_floating_point and t.element_size() == 4
                if t.dim() != 2 or not t.is_contiguous() or not (s16 or f32):
                    raise LoudscanError("track %d: need a contiguous [frames, channels] float32 or int16 tensor" % i)
                if not t.is_cuda:
                    raise LoudscanError("track %d: PCM must be resident in HBM (device tensor)" % i)
                ptr, frames, ch = t.data_ptr(), t.shape[0], t.shape[1]
                fmts[i] = _lib.PCM_S16 if s16 else _lib.PCM_F32
            arr[i] = LgdTrack(ptr, frames, ch, int(r))
        self._chk(self.L.lgd_plan_formats(self.ctx, fmts if any(fmts[i] for i in range(len(tracks))) else None,
                                          len(tracks)))
        self._keep = tracks
        self.n_tracks = len(tracks)
        # album: False | True (all stages on this GPU) | "part1" (multi-GPU: the
        # caller exchanges the partials and drives stages 2 and 3)
        aflag = FLAG_ALBUM_PART1 if album == "part1" else (FLAG_ALBUM if (album or albums is not None) else 0)
        self.flags = (FLAG_TRUE_PEAK if true_peak else 0) | aflag
        if albums is None:
            self.n_albums = None
            self._chk(self.L.lgd_plan(self.ctx, arr, len(tracks), self.flags))
        else:
            if len(albums) != len(tracks):
                raise LoudscanError("albums: one index per track")
            self.n_albums = (max(albums) + 1) if len(albums) else 1
            idx = (C.c_uint32 * max(1, len(albums)))(*[int(a) for a in albums])
            self._chk(self.L.lgd_plan_albums(self.ctx, arr, len(tracks), idx, self.n_albums, self.flags))
        return self

    def execute(self, stream=None):
        self._chk(self.L.lgd_execute(self.ctx, _stream_handle(stream)))
        return self

    def join(self, stream=None):
        """Order `stream` behind every scan enqueued so far ("overlap" 1: they may run on the
        engine's own stream); needed before `stream` overwrites a PCM buffer being scanned."""
        self._chk(self.L.lgd_join(self.ctx, _stream_handle(stream)))

    def fetch(self):
        res = (LgdTrackResult * max(1, self.n_tracks))()
        na = self.n_albums or 1
        alb = (LgdAlbumResult * na)()
        want_album = bool(self.flags & (FLAG_ALBUM | FLAG_ALBUM_PART1))
        self._chk(self.L.lgd_fetch(self.ctx, res, alb if want_album else None))
        tracks = [res[i].asdict() for i in range(self.n_tracks)]
        if not want_album:
            return tracks, None
        return tracks, (alb[0].asdict() if self.n_albums is None else [alb[i].asdict() for i in range(na)])

    def scan(self, tracks, rates, true_peak=True, album=False, stream=None, albums=None):
        return self.plan(tracks, rates, true_peak, album, albums).execute(stream).fetch()

    # -- introspection -----------------------------------------------------------
    def subblock_energies(self, track):
        n = C.c_uint64()
        self._chk(self.L.lgd_copy_subblock_energies(self.ctx, track, None, 0, C.byref(n)))
        out = np.zeros(n.value, np.float64)
        if n.value:
            self._chk(self.L.lgd_copy_subblock_energies(self.ctx, track, out.ctypes.data, n.value,
                                                        C.byref(n)))
        return out

    def channel_peaks(self, track, channels):
        sp = np.zeros(channels, np.float64)
        tp = np.zeros(channels, np.float64)
        self._chk(self.L.lgd_copy_channel_peaks(self.ctx, track, sp.ctypes.data, tp.ctypes.data, channels))
        return sp, tp

    def last_kernel_ms(self):
        a, b = C.c_float(-1), C.c_float(-1)
        self._chk(self.L.lgd_last_kernel_ms(self.ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def kernel_ms_stats(self, last_n=0):
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        n = C.c_uint32()
        self._chk(self.L.lgd_kernel_ms_stats(self.ctx, last_n, C.byref(a), C.byref(b), C.byref(c),
                                             C.byref(n)))
        so_mean, so_min = C.c_float(float("nan")), C.c_float(float("nan"))
        # (the marker behind the scan kernels is recorded with set_param("timing", 2) only: it costs ~5 us of
        # queue time inside the bracket scan_mean_ms is taken over)
        if self.L.lgd_scan_only_ms_stats(self.ctx, last_n, C.byref(so_mean), C.byref(so_min)) != 0:
            so_mean, so_min = C.c_float(float("nan")), C.c_float(float("nan"))
        # scan_*: every kernel that reads PCM (scan kernels + the true-peak kernel behind them);
        # scan_only_*: the scan kernels alone
        return dict(scan_mean_ms=a.value, scan_min_ms=b.value, total_mean_ms=c.value, n=n.value,
                    scan_only_mean_ms=so_mean.value, scan_only_min_ms=so_min.value)

    def plan_info(self):
        ns, nb, pb, wb = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        ck = C.c_uint32()
        self._chk(self.L.lgd_plan_info(self.ctx, C.byref(ns), C.byref(nb), C.byref(ck),
                                       C.byref(pb), C.byref(wb)))
        return dict(segments=ns.value, subblocks=nb.value, chunk=ck.value, pcm_bytes=pb.value,
                    warm_bytes=wb.value)

    # -- multi-GPU album plumbing (device pointers of the exchanged records) ---------
    def album_records(self):
        """(record 1 pointer, its length in doubles, record 2 pointer) of the most recent execute."""
        r1, r2 = C.c_void_p(), C.c_void_p()
        n = C.c_uint64()
        self._chk(self.L.lgd_album_record1(self.ctx, C.byref(r1), C.byref(n)))
        self._chk(self.L.lgd_album_record2(self.ctx, C.byref(r2)))
        return r1.value, n.value, r2.value

    def album_join(self, stream=None):
        """Order `stream` behind the most recent execute (which may run on an internal stream)."""
        self._chk(self.L.lgd_album_join(self.ctx, _stream_handle(stream)))

    def album_stage2(self, all_rec1_ptr=None, world=1, stream=None):
        self._chk(self.L.lgd_album_stage2(self.ctx, all_rec1_ptr, world, _stream_handle(stream)))

    def album_stage3(self, all_rec2_ptr=None, world=1, stream=None):
        self._chk(self.L.lgd_album_stage3(self.ctx, all_rec2_ptr, world, _stream_handle(stream)))
