"""GPU: the scan.h drop-in (include/loudscan.h) driven the way loudgain.c's main
drives the reference (loudgain.c:299-340,651-654), compared field by field with
the oracle's restatement of scan.c."""
import ctypes as C
import math
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from loudgain_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_wav(path, pcm, rate, kind="s16"):
    """pcm: float32 [frames, ch] on the S16 grid."""
    ch = pcm.shape[1]
    if kind == "s16":
        data, tag, bits = np.round(pcm * 32768).astype("<i2").tobytes(), 1, 16
    elif kind == "u8":
        data, tag, bits = (np.round(pcm * 128).astype(np.int16) + 128).clip(0, 255).astype(np.uint8).tobytes(), 1, 8
    elif kind == "s24":
        v = np.round(pcm.astype(np.float64) * 8388608).astype(np.int64).clip(-8388608, 8388607)
        b = np.zeros((v.size, 3), np.uint8)
        vv = (v.reshape(-1) & 0xFFFFFF)
        b[:, 0], b[:, 1], b[:, 2] = vv & 255, (vv >> 8) & 255, (vv >> 16) & 255
        data, tag, bits = b.tobytes(), 1, 24
    elif kind == "s32":
        data, tag, bits = np.round(pcm.astype(np.float64) * 2147483648).clip(-2**31, 2**31 - 1).astype("<i4").tobytes(), 1, 32
    elif kind == "f32":
        data, tag, bits = pcm.astype("<f4").tobytes(), 3, 32
    elif kind == "f64":
        data, tag, bits = pcm.astype("<f8").tobytes(), 3, 64
    elif kind == "ext16":  # WAVE_FORMAT_EXTENSIBLE carrying PCM16
        data, tag, bits = np.round(pcm * 32768).astype("<i2").tobytes(), 0xFFFE, 16
    ba = ch * bits // 8
    with open(path, "wb") as f:
        if tag == 0xFFFE:
            fmt = struct.pack("<HHIIHHHHIH14s", tag, ch, rate, rate * ba, ba, bits, 22, bits, 0, 1,
                              b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71")
        else:
            fmt = struct.pack("<HHIIHH", tag, ch, rate, rate * ba, ba, bits)
        f.write(b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(data)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<I", len(fmt)) + fmt)
        f.write(b"LIST" + struct.pack("<I", 4) + b"INFO")  # a chunk to skip
        f.write(b"data" + struct.pack("<I", len(data)) + data)
    return path


FIELDS = ["track_gain", "track_peak", "track_loudness", "track_loudness_range", "album_gain",
          "album_peak", "album_loudness", "album_loudness_range", "loudness_reference"]


def close(a, b, tol):
    return (a == b) or (math.isinf(a) and math.isinf(b) and (a > 0) == (b > 0)) or abs(a - b) <= tol


def test_album_session_like_loudgain_main(oracle, tmp_path):
    from loudgain_amd import scan
    L = oracle.lib()
    specs = [("a.wav", 48000, 2, 21.3, "s16"), ("b.wav", 48000, 2, 12.0, "s24"), ("c.wav", 44100, 1, 9.5, "f32"),
             ("d.wav", 48000, 6, 8.2, "s16"), ("e.wav", 48000, 2, 0.3, "u8"), ("f.wav", 96000, 2, 6.1, "s32"),
             ("g.wav", 48000, 2, 7.7, "ext16"), ("h.wav", 48000, 1, 5.5, "f64")]
    paths = []
    for i, (name, rate, ch, secs, kind) in enumerate(specs):
        pcm = synth.track_numpy(int(rate * secs), ch, rate, seed=40 + i, step_s=1.9)
        if i == 1:
            pcm = synth.snap_s16_numpy(pcm * 0.05)
        paths.append(write_wav(str(tmp_path / name), pcm, rate, kind))
    n = len(paths)
    # reference-side restatement
    L.lgo_scan_init(n)
    for i, p in enumerate(paths):
        assert L.lgo_scan_file(p.encode(), i) == 0
    # product
    assert scan.scan_init(n) == 0
    for i, p in enumerate(paths):
        assert scan.scan_file(p, i) == 0       # loudgain.c:304 (return value ignored there)
    assert scan.scan_file(paths[0], n) == -1   # scan.c:132-135
    assert scan.scan_album_has_different_containers() == L.lgo_scan_album_has_different_containers() == 0
    assert scan.scan_album_has_different_codecs() == L.lgo_scan_album_has_different_codecs() == 1
    assert scan.scan_album_has_opus() == 0
    for pre in (0.0, -5.0):
        for i in range(n):                     # loudgain.c:323-340
            r = scan.scan_get_track_result(i, pre)
            o = L.lgo_scan_get_track_result(i, pre).contents
            scan.scan_set_album_result(r, pre)
            L.lgo_scan_set_album_result(C.byref(o), pre)
            assert r.codec_id == o.codec_id and r.container == o.container == b"wav"
            assert r.file == o.file
            for f in FIELDS:
                tol = 1e-4 if "peak" in f else 1e-6
                assert close(getattr(r, f), getattr(o, f), tol), (i, f, getattr(r, f), getattr(o, f))
            r.free()                            # loudgain.c:651
    assert abs(scan.scan_get_album_peak() - L.lgo_scan_get_album_peak()) <= 1e-4
    assert scan.scan_get_track_result(n, 0.0) is None    # "Index too high", scan.c:283-286
    scan.scan_deinit()
    L.lgo_scan_deinit()


def test_pcm_entry_points_and_opus_rule(oracle):
    import torch
    from loudgain_amd import scan
    rate = 48000
    pcm = synth.track_numpy(rate * 7, 2, rate, seed=77, step_s=2.0)
    s16 = np.round(pcm * 32768).astype(np.int16)
    ref = oracle.scan_track(pcm, rate)
    scan.scan_init(3)
    assert scan.scan_pcm(s16, rate, 0) == 0
    assert scan.scan_pcm(pcm, rate, 1) == 0
    dev = torch.from_numpy(pcm).cuda()
    assert scan.scan_pcm(dev, rate, 2) == 0
    assert scan.scan_pcm(pcm, rate, 3) == -1
    scan.scan_set_codec(2, 0x1503C, "ogg")      # Opus: -23 LUFS reference (scan.c:309-311)
    rs = [scan.scan_get_track_result(i, 0.0) for i in range(3)]
    for r in rs:
        assert abs(r.track_loudness - ref["loudness"]) <= 1e-6
        assert abs(r.track_peak - ref["peak"]) <= 1e-4
    assert rs[0].track_gain == pytest.approx(-18.0 - ref["loudness"], abs=1e-6)
    assert rs[2].track_gain == pytest.approx(-23.0 - ref["loudness"], abs=1e-6)
    assert rs[2].loudness_reference == -23.0 and rs[0].loudness_reference == -18.0
    assert scan.scan_album_has_opus() == 1 and scan.scan_album_has_different_containers() == 1
    scan.scan_set_album_result(rs[0], 0.0)      # album with Opus present: pre_gain - 5 (scan.c:397-398)
    assert rs[0].album_gain == pytest.approx(-23.0 - rs[0].album_loudness, abs=1e-9)
    scan.scan_deinit()


def test_short_and_silent_files(oracle, tmp_path):
    from loudgain_amd import scan
    p1 = write_wav(str(tmp_path / "short.wav"), synth.track_numpy(4000, 2, 48000, seed=1), 48000)
    p2 = write_wav(str(tmp_path / "silent.wav"), np.zeros((48000 * 2, 2), np.float32), 48000)
    scan.scan_init(2)
    scan.scan_file(p1, 0)
    scan.scan_file(p2, 1)
    a, b = scan.scan_get_track_result(0, 0.0), scan.scan_get_track_result(1, 0.0)
    # < 400 ms and digital silence: loudness -inf -> gain +inf (SURVEY 8a semantics)
    assert a.track_loudness == -math.inf and a.track_gain == math.inf and a.track_peak > 0
    assert b.track_loudness == -math.inf and b.track_peak == 0.0 and b.track_loudness_range == 0.0
    scan.scan_set_album_result(a, 0.0)
    assert a.album_loudness == -math.inf
    scan.scan_deinit()


@pytest.mark.parametrize("code,msg", [
    ("scan.scan_init(1); scan.scan_file('/nonexistent/x.wav', 0)", "Could not open input"),
    ("open(p,'wb').write(b'not a wave file at all'); scan.scan_init(1); scan.scan_file(p, 0)", "Could not find stream info"),
])
def test_fatal_errors_exit_like_fail_printf(tmp_path, code, msg):
    """fail_printf prints and _exit(EXIT_FAILURE)s (printf.c:94-102): check in a child."""
    src = ("import sys; sys.path.insert(0, %r)\nfrom loudgain_amd import scan\np=%r\n%s\nprint('survived')\n"
           % (ROOT, str(tmp_path / "bad.wav"), code))
    r = subprocess.run([sys.executable, "-c", src], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1, (r.returncode, r.stderr[-300:])
    assert msg in r.stderr and "survived" not in r.stdout


def _session(paths, scan):
    n = len(paths)
    assert scan.scan_init(n) == 0
    for i, p in enumerate(paths):
        assert scan.scan_file(p, i) == 0
    out = []
    for i in range(n):
        r = scan.scan_get_track_result(i, 0.0)
        scan.scan_set_album_result(r, 0.0)
        out.append({f: getattr(r, f) for f in FIELDS})
        r.free()
    peak = scan.scan_get_album_peak()
    scan.scan_deinit()
    return out, peak


@pytest.mark.parametrize("virtual", [2, 3, 5])
def test_album_over_several_devices_of_one_process(oracle, tmp_path, virtual, monkeypatch):
    """scan_set_devices / LOUDSCAN_DEVICES: the files of a session dealt round-robin over several GPUs
    of ONE process, album values from the exchange of partial sums (what loudgain.c:299-340 gets
    without knowing).  A one-GPU box rehearses it with LOUDSCAN_VIRTUAL_DEVICES: that many engine
    contexts, all stages of the exchange, one device.  Must equal the single-context session bit for
    bit in every count-derived field and the oracle within the usual bars."""
    from loudgain_amd import scan
    L = oracle.lib()
    specs = [(48000, 2, 14.0, 1.0), (48000, 2, 9.0, 0.04), (44100, 1, 11.0, 1.0), (48000, 2, 33.0, 0.5),
             (96000, 2, 5.0, 1.0), (48000, 6, 4.5, 0.8), (48000, 2, 0.2, 1.0), (48000, 2, 17.0, 0.25)]
    paths = []
    for i, (rate, ch, secs, g) in enumerate(specs):
        pcm = synth.snap_s16_numpy(synth.track_numpy(int(rate * secs), ch, rate, seed=60 + i, step_s=1.5) * g)
        paths.append(write_wav(str(tmp_path / ("t%d.wav" % i)), pcm, rate))
    monkeypatch.delenv("LOUDSCAN_VIRTUAL_DEVICES", raising=False)
    one, peak1 = _session(paths, scan)
    monkeypatch.setenv("LOUDSCAN_VIRTUAL_DEVICES", str(virtual))
    many, peakn = _session(paths, scan)
    monkeypatch.delenv("LOUDSCAN_VIRTUAL_DEVICES")
    L.lgo_scan_init(len(paths))
    for i, p in enumerate(paths):
        L.lgo_scan_file(p.encode(), i)
    for i, (a, b) in enumerate(zip(one, many)):
        o = L.lgo_scan_get_track_result(i, 0.0).contents
        L.lgo_scan_set_album_result(C.byref(o), 0.0)
        for f in FIELDS:
            if f.startswith("track") or f == "loudness_reference":
                assert a[f] == b[f] or (math.isinf(a[f]) and math.isinf(b[f]))       # same kernels, same plan per track
            else:
                assert close(a[f], b[f], 1e-9), (f, a[f], b[f])                        # album: another summation order
            assert close(b[f], getattr(o, f), 1e-4 if "peak" in f else 1e-6), (i, f)
    assert peak1 == peakn
    L.lgo_scan_deinit()


def test_rescan_reinit_and_big_file(oracle, tmp_path):
    """What scan.c tolerates: scanning an index again after results were read, scan_init without a
    scan_deinit in between.  And a file larger than one staging piece (64 MB of S16), i.e. the
    double-buffered pinned upload path with several pieces."""
    from loudgain_amd import scan
    rate = 48000
    big = synth.track_numpy(rate * 60 * 7, 2, rate, seed=5, step_s=11.0)          # 40.3 M samples = 80 MB as S16
    small = synth.track_numpy(rate * 6, 2, rate, seed=6)
    pb, ps = write_wav(str(tmp_path / "big.wav"), big, rate), write_wav(str(tmp_path / "small.wav"), small, rate)
    ref_b, ref_s = oracle.scan_track(big, rate), oracle.scan_track(small, rate)
    scan.scan_init(2)
    scan.scan_file(ps, 0)
    scan.scan_file(ps, 1)
    a = scan.scan_get_track_result(1, 0.0)
    assert abs(a.track_loudness - ref_s["loudness"]) <= 1e-6
    scan.scan_file(pb, 1)                       # index 1 again, after its result was read
    b = scan.scan_get_track_result(1, 0.0)
    assert abs(b.track_loudness - ref_b["loudness"]) <= 1e-6 and abs(b.track_peak - ref_b["peak"]) <= 1e-4
    assert abs(b.track_loudness_range - ref_b["lra"]) <= 1e-6
    scan.scan_init(1)                           # no scan_deinit: the old session is dropped
    scan.scan_file(ps, 0)
    c = scan.scan_get_track_result(0, 0.0)
    assert abs(c.track_loudness - ref_s["loudness"]) <= 1e-6
    scan.scan_deinit()


@pytest.mark.parametrize("announced", [0, 0xFFFFFFFF, 10**9])
def test_wav_with_unreliable_data_size(oracle, tmp_path, announced):
    """Streamed WAVs announce 0 or 0xFFFFFFFF bytes of data, truncated ones too many: the reader goes by
    the file (FFmpeg's demuxer reads to the end of the file as well, scan.c:225)."""
    from loudgain_amd import scan
    rate = 48000
    pcm = synth.track_numpy(rate * 5, 2, rate, seed=9, step_s=1.0)
    p = write_wav(str(tmp_path / "s.wav"), pcm, rate)
    raw = bytearray(open(p, "rb").read())
    off = raw.index(b"data") + 4
    raw[off:off + 4] = struct.pack("<I", announced)
    open(p, "wb").write(raw)
    ref = oracle.scan_track(pcm, rate)
    scan.scan_init(1)
    scan.scan_file(p, 0)
    r = scan.scan_get_track_result(0, 0.0)
    assert abs(r.track_loudness - ref["loudness"]) <= 1e-6 and abs(r.track_peak - ref["peak"]) <= 1e-4
    scan.scan_deinit()


def test_more_devices_than_files(oracle, tmp_path, monkeypatch):
    """Devices without a single file take part in the album exchange with empty records."""
    from loudgain_amd import scan
    paths = []
    for i in range(2):
        pcm = synth.track_numpy(48000 * (6 + 3 * i), 2, 48000, seed=90 + i, step_s=1.1)
        paths.append(write_wav(str(tmp_path / ("m%d.wav" % i)), pcm, 48000))
    one, peak1 = _session(paths, scan)
    monkeypatch.setenv("LOUDSCAN_VIRTUAL_DEVICES", "5")
    many, peakn = _session(paths, scan)
    monkeypatch.delenv("LOUDSCAN_VIRTUAL_DEVICES")
    assert peak1 == peakn
    for a, b in zip(one, many):
        for f in FIELDS:
            assert close(a[f], b[f], 1e-9), (f, a[f], b[f])
