"""GPU: scan_file on files larger than one pinned staging piece (64 MiB of S16) whose channel count does
not divide the piece -- the packet loop of /root/reference/src/scan.c:225-250 feeding scan.c:436-448,
here a sequential WAV reader feeding the double-buffered upload (scan_api.cpp `upload`).

Round 2 cut pieces at 2^25 samples whatever the channel count: for 3 / 5 / 6 / 7 channels every piece
after the first started 2 samples late (channels rotated, the file's last samples lost).  Every case
below is compared (a) with the oracle's restatement of scan.c per channel and (b) bit for bit with the
same samples handed over as one array (scan_pcm_s16: memcpy staging, never affected)."""
import ctypes as C
import math

import numpy as np
import pytest

from loudgain_amd import synth
from test_gpu_scan_api import write_wav

pytestmark = pytest.mark.gpu


def _material(frames, ch, rate, seed):
    """Every channel at its own level and with its own late burst: a rotation of the channels or a lost
    tail changes every per-channel peak and (weights 1 / 1 / 1 / 0 / 1.41 / 1.41) the loudness."""
    pcm = synth.track_numpy(frames, ch, rate, seed=seed, step_s=7.0)
    pcm *= (1.0 - 0.09 * np.arange(ch, dtype=np.float32))[None, :]
    n = np.arange(rate // 50)
    for c in range(ch):  # 20 ms bursts inside the last 100 ms, the very last one ending on the last frame
        t1 = frames - c * (rate // 100)
        t0 = t1 - len(n)
        pcm[t0:t1, c] += (0.3 + 0.05 * c) * np.sin(2 * np.pi * n * (0.11 + 0.01 * c)).astype(np.float32)
    return synth.snap_s16_numpy(pcm)


def _oracle_file(oracle, path):
    L = oracle.lib()
    L.lgo_scan_init(1)
    assert L.lgo_scan_file(path.encode(), 0) == 0
    o = L.lgo_scan_get_track_result(0, 0.0).contents
    st = L.lgo_scan_state(0)
    nch = L.lgo_channels(st)
    tp, sp = [], []
    for c in range(nch):
        v = C.c_double()
        L.lgo_true_peak(st, c, C.byref(v)); tp.append(v.value)
        L.lgo_sample_peak(st, c, C.byref(v)); sp.append(v.value)
    out = dict(loudness=o.track_loudness, lra=o.track_loudness_range, peak=o.track_peak, gain=o.track_gain,
               true_peak=tp, sample_peak=sp)
    L.lgo_scan_deinit()
    return out


@pytest.mark.parametrize("ch,kind,min_samples", [
    (3, "s16", (1 << 26) + 1000), (5, "s16", (1 << 25) + 1000), (6, "s16", (1 << 26) + 1000),
    (7, "s16", (1 << 25) + 1000), (3, "s24", (1 << 25) + 1000), (5, "s24", (1 << 26) + 1000),
    (6, "s24", (1 << 25) + 1000), (7, "s24", (1 << 26) + 1000)])
def test_multi_piece_odd_channel_wav(oracle, tmp_path, ch, kind, min_samples):
    from loudgain_amd import scan
    rate = 48000
    frames = min_samples // ch + 4801 + ch  # ragged: not a whole sub-block, not a multiple of 8
    pcm = _material(frames, ch, rate, seed=100 + ch)
    path = write_wav(str(tmp_path / "m.wav"), pcm, rate, kind)
    ref = _oracle_file(oracle, path)
    s16 = np.round(pcm * 32768).astype(np.int16)
    scan.scan_init(2)
    assert scan.scan_file(path, 0) == 0
    assert scan.scan_pcm(s16, rate, 1) == 0
    a, b = scan.scan_get_track_result(0, 0.0), scan.scan_get_track_result(1, 0.0)
    sp_a, tp_a = scan.scan_get_channel_peaks(0)
    sp_b, tp_b = scan.scan_get_channel_peaks(1)
    scan.scan_deinit()
    # (b) file == array, bit for bit (same samples, same plan shape per track)
    for f in ("track_loudness", "track_loudness_range", "track_peak", "track_gain"):
        assert getattr(a, f) == getattr(b, f), (f, getattr(a, f), getattr(b, f))
    assert sp_a == sp_b and tp_a == tp_b
    # (a) oracle, per channel
    assert sp_a == ref["sample_peak"], (sp_a, ref["sample_peak"])
    assert len(set(sp_a)) == ch  # the material really tells the channels apart
    for c in range(ch):
        assert abs(tp_a[c] - ref["true_peak"][c]) <= 1e-4, (c, tp_a[c], ref["true_peak"][c])
    assert abs(a.track_loudness - ref["loudness"]) <= 1e-6
    assert abs(a.track_loudness_range - ref["lra"]) <= 1e-6
    assert abs(a.track_peak - ref["peak"]) <= 1e-4
    assert abs(a.track_gain - ref["gain"]) <= 1e-6


def test_staging_pieces_are_whole_frames():
    """Host model of the piece arithmetic for every channel count the scanner accepts."""
    stage = 64 << 20
    for ch in range(1, 65):
        for elem in (2, 4):
            piece = stage // elem
            piece -= piece % (8 * ch)
            assert piece > 0 and piece % ch == 0 and piece % 8 == 0 and piece * elem <= stage


def test_config5_layouts_as_wav_session(oracle, tmp_path):
    """BASELINE.json config 5's twelve (rate, layout) pairs, 120 s each, written as S16 WAV and scanned as ONE
    album through the scan.h session the way loudgain.c:299-340 drives it; the 5.1 files at 48 / 96 / 192 kHz are
    2 / 3 / 5 staging pieces.  Per track and per channel against the oracle's scan.c restatement, album too."""
    from loudgain_amd import scan
    L = oracle.lib()
    specs = [(r, c) for r in (44100, 48000, 96000, 192000) for c in (1, 2, 6)]
    paths = []
    for i, (rate, ch) in enumerate(specs):
        pcm = _material(rate * 120 + 17 * i, ch, rate, seed=500 + i)
        paths.append(write_wav(str(tmp_path / ("t%02d.wav" % i)), pcm, rate, "s16"))
        del pcm
    n = len(paths)
    L.lgo_scan_init(n)
    scan.scan_init(n)
    for i, p in enumerate(paths):
        assert L.lgo_scan_file(p.encode(), i) == 0
        assert scan.scan_file(p, i) == 0
    for i, (rate, ch) in enumerate(specs):
        o = L.lgo_scan_get_track_result(i, 0.0).contents
        L.lgo_scan_set_album_result(C.byref(o), 0.0)
        g = scan.scan_get_track_result(i, 0.0)
        scan.scan_set_album_result(g, 0.0)
        st = L.lgo_scan_state(i)
        sp, tp = scan.scan_get_channel_peaks(i)
        for c in range(ch):
            v = C.c_double()
            L.lgo_sample_peak(st, c, C.byref(v))
            assert sp[c] == v.value, (i, c)
            L.lgo_true_peak(st, c, C.byref(v))
            assert abs(tp[c] - v.value) <= 1e-4, (i, c, tp[c], v.value)
        # low-frequency-free broadband material: 1e-6 LU up to 96 kHz, the oracle's own (rate/48k)^4 noise above
        lu = 1e-6 if rate <= 96000 else 2e-5
        assert abs(g.track_loudness - o.track_loudness) <= lu, (i, g.track_loudness, o.track_loudness)
        assert abs(g.track_loudness_range - o.track_loudness_range) <= lu
        assert abs(g.track_peak - o.track_peak) <= 1e-4
        assert abs(g.album_loudness - o.album_loudness) <= 2e-5
        assert abs(g.album_loudness_range - o.album_loudness_range) <= 2e-5
        assert abs(g.album_peak - o.album_peak) <= 1e-4
        assert not math.isinf(g.track_loudness)
    assert abs(scan.scan_get_album_peak() - L.lgo_scan_get_album_peak()) <= 1e-4
    scan.scan_deinit()
    L.lgo_scan_deinit()
