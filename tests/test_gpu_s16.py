"""Interleaved S16 PCM resident in HBM, read as it is (LGD_PCM_S16, lgd_plan_formats).

The reference never sees anything else: scan_frame converts every decoded frame to interleaved S16
(/root/reference/src/scan.c:414,442) and feeds ebur128_add_frames_short (scan.c:448), which scales by 1/32768.
The S16 variants of lgd_scan_kernel / lgd_tp_kernel widen to the integer-valued float at staging time and carry
the power-of-two scale to the values they store; every operation in between is linear, so their results must be
BIT-IDENTICAL to the f32 variants fed with k/32768 -- every sub-block energy, every peak, every count is
compared with `==` here -- and within the north-star tolerances of the oracle.
"""
import numpy as np
import pytest

from loudgain_amd import synth
from tests.gpu_util import check_track, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scanner():
    from loudgain_amd.device import DeviceScanner
    s = DeviceScanner(0)
    yield s
    s.close()


def _as_s16(x):
    q = np.rint(x.astype(np.float64) * 32768.0)
    assert np.array_equal(q / 32768.0, x.astype(np.float64)), "material must lie on the S16 grid"
    assert q.size == 0 or (q.min() >= -32768 and q.max() <= 32767)
    return q.astype(np.int16)


def _material(kind, frames, nch, rate, seed):
    if kind == "limited":
        return synth.limited_numpy(frames, nch, rate, seed=seed)
    if kind == "adversarial":
        n = np.arange(frames)
        x = 0.8 * np.sin(2 * np.pi * n / 4.0 + np.pi / 2)
        return synth.snap_s16_numpy(np.repeat(x[:, None], nch, 1).astype(np.float32))
    if kind == "fullscale":  # the grid's ends: -32768 and 32767
        rng = np.random.default_rng(seed)
        x = rng.integers(-32768, 32768, size=(frames, nch)).astype(np.float32) / 32768.0
        x[frames // 3] = -1.0
        x[frames // 2] = 32767.0 / 32768.0
        return x
    return synth.track_numpy(frames, nch, rate, seed=seed, step_s=1.1)


def _same(a, b):
    for k in a:
        va, vb = a[k], b[k]
        if isinstance(va, float) and np.isnan(va) and np.isnan(vb):
            continue
        assert va == vb, (k, va, vb)


CASES = [
    # rate, channels, seconds, material
    (48000, 2, 9.37, "steps"), (48000, 1, 7.03, "steps"), (44100, 2, 8.21, "steps"), (44100, 1, 6.4, "steps"),
    (96000, 2, 5.3, "steps"), (96000, 1, 4.1, "limited"), (192000, 2, 3.7, "steps"), (88200, 2, 4.4, "steps"),
    (22050, 2, 9.9, "steps"), (11025, 2, 12.3, "steps"), (11025, 1, 12.3, "limited"), (8000, 2, 14.0, "steps"),
    (32000, 1, 9.0, "steps"), (48000, 2, 6.11, "limited"), (48000, 2, 5.5, "adversarial"), (48000, 2, 5.2, "fullscale"),
    (44100, 2, 5.2, "fullscale"), (48000, 2, 0.35, "steps"), (48000, 1, 0.0, "steps"), (48000, 2, 3.05, "steps"),
]


@pytest.mark.parametrize("true_peak", [True, False])
@pytest.mark.parametrize("rate,nch,secs,kind", CASES)
def test_s16_bit_identical_to_f32(scanner, oracle, rate, nch, secs, kind, true_peak):
    frames = int(rate * secs)
    x = _material(kind, max(frames, 1), nch, rate, seed=rate % 1000 + nch)[:frames]
    xf = to_dev(x.astype(np.float32).reshape(frames, nch))
    xs = to_dev(_as_s16(x).reshape(frames, nch))
    rs, _ = scanner.scan([xs], rate, true_peak=true_peak)
    es = scanner.subblock_energies(0)
    ps = scanner.channel_peaks(0, nch)
    # bit for bit only on the same summation tree: the S16 plan may prefer another chunk length than the f32 plan
    # (44.1 kHz: C = 63 at three waves per SIMD instead of C = 70 at two)
    scanner.set_param("chunk", scanner.plan_info()["chunk"])
    rf, _ = scanner.scan([xf], rate, true_peak=true_peak)
    ef = scanner.subblock_energies(0)
    pf = scanner.channel_peaks(0, nch)
    scanner.set_param("chunk", 0)
    _same(rf[0], rs[0])
    assert np.array_equal(ef, es)
    assert np.array_equal(pf[0], ps[0]) and np.array_equal(pf[1], ps[1])
    if frames and true_peak:
        ref = oracle.scan_track(x.astype(np.float32).reshape(frames, nch), rate)
        check_track(rs[0], ref, tp=True, rate=rate)


def test_mixed_formats_in_one_plan(scanner, oracle):
    """S16 and f32 tracks of several rates and layouts in one plan with an album."""
    specs = [(48000, 2, 6.3, True), (48000, 2, 5.1, False), (44100, 1, 7.7, True), (48000, 6, 4.2, False), (48000, 6, 3.9, True),
             (96000, 2, 3.3, True), (44100, 2, 4.9, False), (48000, 1, 5.5, True), (11025, 2, 8.0, True)]
    pcm, devs = [], []
    for i, (rate, nch, secs, s16) in enumerate(specs):
        frames = int(rate * secs)
        x = synth.track_numpy(frames, nch, rate, seed=40 + i, step_s=0.9)
        pcm.append(x)
        devs.append(to_dev(_as_s16(x) if s16 else x))
    rates = [s[0] for s in specs]
    got, alb = scanner.scan(devs, rates, true_peak=True, album=True)
    ref_f32, alb_f32 = scanner.scan([to_dev(x) for x in pcm], rates, true_peak=True, album=True)
    for g, r in zip(got, ref_f32):
        _same(g, r)
    _same(alb, alb_f32)
    for g, x, rate in zip(got, pcm, rates):
        check_track(g, oracle.scan_track(x, rate), tp=True, rate=rate)


def test_s16_segments_of_a_long_track(scanner):
    """Many segments, warm-up tiles in front of each, the 44.1 kHz family's odd sub-block offsets (a segment's first
    sample is then only 4-byte aligned in S16)."""
    for rate, nch in ((44100, 2), (44100, 1), (48000, 2), (22050, 1)):
        frames = int(rate * 171.3)
        x = synth.track_numpy(frames, nch, rate, seed=7, step_s=3.1)
        a, _ = scanner.scan([to_dev(x)], rate, true_peak=True)
        ea = scanner.subblock_energies(0)
        b, _ = scanner.scan([to_dev(_as_s16(x))], rate, true_peak=True)
        eb = scanner.subblock_energies(0)
        _same(a[0], b[0])
        assert np.array_equal(ea, eb)


WIDE = [(48000, 3), (48000, 4), (48000, 5), (48000, 6), (44100, 6), (96000, 6), (48000, 7), (48000, 8), (48000, 9),
        (48000, 12), (48000, 16), (44100, 21), (48000, 24), (11025, 6), (22050, 3)]


@pytest.mark.parametrize("strided", [1, 0, 2])
@pytest.mark.parametrize("rate,nch", WIDE)
def test_s16_wide_layouts(scanner, oracle, rate, nch, strided):
    """3 .. 24 channels: planar kernels, channel pairs / triples / quads of the interleaved stream (aligned and
    unaligned sets, overlapping sets), the run-time-channel kernel and its channel groups -- S16 against f32, bit for bit."""
    frames = int(rate * 3.37)
    x = synth.track_numpy(frames, nch, rate, seed=nch, step_s=0.7)
    scanner.set_param("strided", strided)
    try:
        rs, _ = scanner.scan([to_dev(_as_s16(x))], rate, true_peak=True)
        es, ps = scanner.subblock_energies(0), scanner.channel_peaks(0, nch)
        rf, _ = scanner.scan([to_dev(x)], rate, true_peak=True)
        ef, pf = scanner.subblock_energies(0), scanner.channel_peaks(0, nch)
    finally:
        scanner.set_param("strided", 1)
    _same(rf[0], rs[0])
    assert np.array_equal(ef, es)
    assert np.array_equal(pf[0], ps[0]) and np.array_equal(pf[1], ps[1])
    if strided == 1:
        check_track(rs[0], oracle.scan_track(x, rate), tp=True, rate=rate)


def test_scan_h_sessions_keep_s16(tmp_path):
    """scan.h level: S16 input (host, device, a 16-bit RIFF/WAVE file through scan_file) of mono / stereo tracks stays
    S16 in HBM; a session fed the same samples as f32 must report the same bits, track by track and for the album.
    5.1 likewise (channel triples of the S16 stream)."""
    import torch
    from loudgain_amd import scan
    from tests.test_gpu_scan_api import write_wav
    specs = [(48000, 2, 7.3), (44100, 2, 6.1), (48000, 1, 5.2), (96000, 2, 3.4), (48000, 6, 4.0)]
    pcm = [synth.track_numpy(int(r * s), c, r, seed=60 + i, step_s=1.2) for i, (r, c, s) in enumerate(specs)]
    n = len(specs)

    def session(feed):
        scan.scan_init(n)
        keep = [feed(i) for i in range(n)]
        out = []
        for i in range(n):
            r = scan.scan_get_track_result(i, 0.0)
            scan.scan_set_album_result(r, 0.0)
            out.append((r.track_loudness, r.track_loudness_range, r.track_peak, r.album_loudness, r.album_loudness_range,
                        r.album_peak) + tuple(np.concatenate(scan.scan_get_channel_peaks(i)).tolist()))
        out.append(scan.scan_get_album_peak())
        scan.scan_deinit()
        del keep
        return out

    def feed_f32(i):
        assert scan.scan_pcm(pcm[i], specs[i][0], i) == 0

    def feed_s16_host(i):
        assert scan.scan_pcm(_as_s16(pcm[i]), specs[i][0], i) == 0

    def feed_s16_device(i):
        t = torch.from_numpy(_as_s16(pcm[i])).cuda()
        assert scan.scan_pcm(t, specs[i][0], i) == 0
        return t

    def feed_wav(i):
        p = write_wav(str(tmp_path / ("t%d.wav" % i)), pcm[i], specs[i][0], "s16")
        assert scan.scan_file(p, i) == 0

    ref = session(feed_f32)
    for feed in (feed_s16_host, feed_s16_device, feed_wav):
        assert session(feed) == ref, feed.__name__


def test_ebur128_shim_short_frames_stay_s16(oracle):
    """ebur128_add_frames_short: scanned as S16, same bits as _float with k / 32768."""
    from loudgain_amd import ebur128
    rate = 48000
    x = synth.track_numpy(rate * 6, 2, rate, seed=91, step_s=1.0)
    a, b = ebur128.State(2, rate), ebur128.State(2, rate)
    for k in range(0, x.shape[0], 8192):
        a.add_frames(_as_s16(x[k:k + 8192]))
        b.add_frames(x[k:k + 8192].astype(np.float32))
    assert a.loudness_global() == b.loudness_global()
    assert a.loudness_range() == b.loudness_range()
    for c in range(2):
        assert a.true_peak(c) == b.true_peak(c) and a.sample_peak(c) == b.sample_peak(c)
    assert ebur128.loudness_global_multiple([a, b]) == ebur128.loudness_global_multiple([b, a])
    ref = oracle.scan_track(x, rate)
    assert abs(a.loudness_global() - ref["loudness"]) <= 1e-6
    a.close()
    b.close()


def test_plan_formats_argument_checks(scanner):
    import ctypes as C
    from loudgain_amd.device import LoudscanError
    L, ctx = scanner.L, scanner.ctx
    bad = (C.c_uint8 * 1)(7)
    assert L.lgd_plan_formats(ctx, bad, 1) != 0                      # unknown format code
    two = (C.c_uint8 * 2)(1, 1)
    assert L.lgd_plan_formats(ctx, two, 2) == 0
    x = to_dev(np.zeros((4800, 2), np.int16))
    from loudgain_amd._lib import LgdTrack
    arr = (LgdTrack * 1)(LgdTrack(x.data_ptr(), 4800, 2, 48000))
    assert L.lgd_plan(ctx, arr, 1, 0) != 0                           # announced for 2 tracks, planned 1
    assert b"announced" in L.lgd_last_error()
    assert L.lgd_plan_formats(ctx, None, 0) == 0                     # back to all-f32
    with pytest.raises(LoudscanError):
        scanner.plan([(x.data_ptr() + 2, 100, 2, 1)], 48000)         # misaligned pointer
