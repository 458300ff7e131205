"""CPU: the RIFF/WAVE reader behind scan_file (scan_wav_probe / scan_wav_read_s16, host
code of libloudscan_hip.so) against a numpy statement of what FFmpeg's decoders +
swr_convert hand to scan.c:442 (interleaved S16), and against the oracle's own reader --
random formats, channel counts, lengths, extra chunks with odd sizes."""
import os
import struct

import numpy as np
import pytest


def _write(path, raw, tag, bits, ch, rate, rng, extensible=False):
    ba = ch * bits // 8
    with open(path, "wb") as f:
        if extensible:
            fmt = struct.pack("<HHIIHHHHIH14s", 0xFFFE, ch, rate, rate * ba, ba, bits, 22, bits, 0, tag,
                              b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71")
        else:
            fmt = struct.pack("<HHIIHH", tag, ch, rate, rate * ba, ba, bits)
        body = b"fmt " + struct.pack("<I", len(fmt)) + fmt
        for _ in range(int(rng.integers(0, 3))):          # chunks to skip, odd sizes are padded
            n = int(rng.integers(0, 40))
            body += rng.choice([b"LIST", b"fact", b"junk"]) + struct.pack("<I", n) + bytes(n) + (b"\0" if n & 1 else b"")
        body += b"data" + struct.pack("<I", len(raw)) + raw
        f.write(b"RIFF" + struct.pack("<I", 4 + len(body)) + b"WAVE" + body)


def _encode(x, kind):
    """x: float64 [frames, ch] in [-1, 1) -> (raw bytes, tag, bits, expected S16)."""
    if kind == "u8":
        v = np.clip(np.rint(x * 128) + 128, 0, 255).astype(np.uint8)
        return v.tobytes(), 1, 8, ((v.astype(np.int32) - 128) * 256).astype(np.int16)
    if kind == "s16":
        v = np.clip(np.rint(x * 32768), -32768, 32767).astype("<i2")
        return v.tobytes(), 1, 16, v.astype(np.int16)
    if kind == "s24":
        v = np.clip(np.rint(x * 8388608), -8388608, 8388607).astype(np.int64)
        b = np.zeros(v.shape + (3,), np.uint8)
        u = v & 0xFFFFFF
        b[..., 0], b[..., 1], b[..., 2] = u & 255, (u >> 8) & 255, (u >> 16) & 255
        return b.tobytes(), 1, 24, (v >> 8).astype(np.int16)          # (s24 << 8) >> 16
    if kind == "s32":
        v = np.clip(np.rint(x * 2147483648.0), -2**31, 2**31 - 1).astype("<i4")
        return v.tobytes(), 1, 32, (v >> 16).astype(np.int16)
    if kind == "f32":
        v = (x * 1.3).astype("<f4")                                     # some values clip
        return v.tobytes(), 3, 32, np.clip(np.rint(v * np.float32(32768.0)), -32768, 32767).astype(np.int16)
    v = (x * 1.3).astype("<f8")
    return v.tobytes(), 3, 64, np.clip(np.rint(v * 32768.0), -32768, 32767).astype(np.int16)


@pytest.mark.parametrize("seed", range(6))
def test_wav_reader_random_files(tmp_path, oracle, seed):
    from loudgain_amd import scan
    rng = np.random.default_rng(seed)
    L = oracle.lib()
    codec = {"u8": 0x10005, "s16": 0x10000, "s24": 0x1000C, "s32": 0x10008, "f32": 0x10015, "f64": 0x10017}
    for i in range(8):
        kind = str(rng.choice(list(codec)))
        ch = int(rng.integers(1, 9))
        rate = int(rng.choice([8000, 22050, 44100, 48000, 96000]))
        frames = int(rng.choice([0, 1, 7, 4799, 4800, 50001]))
        x = rng.uniform(-1.0, 1.0, size=(frames, ch))
        raw, tag, bits, want = _encode(x, kind)
        ext = bool(rng.integers(2)) and kind != "u8"
        p = str(tmp_path / ("f%d_%d.wav" % (seed, i)))
        _write(p, raw, tag, bits, ch, rate, rng, extensible=ext)
        wi = scan.scan_wav_probe(p)
        assert (wi["channels"], wi["rate"], wi["bits"], wi["frames"], wi["codec_id"]) == (ch, rate, bits, frames, codec[kind])
        out = np.full((frames + 3, ch), 12345, np.int16)
        assert scan.scan_wav_read_s16(p, out.ctypes.data, frames + 3) == frames
        assert np.array_equal(out[:frames], want.reshape(frames, ch)), (kind, ch, frames)
        assert (out[frames:] == 12345).all()                            # nothing written past the data
        if frames > 10:                                                # a smaller buffer gets a prefix
            part = np.zeros((10, ch), np.int16)
            assert scan.scan_wav_read_s16(p, part.ctypes.data, 10) == 10
            assert np.array_equal(part, want.reshape(frames, ch)[:10])
        # the oracle's reader (separate C code) sees the same samples: same loudness state
        L.lgo_scan_init(1)
        assert L.lgo_scan_file(p.encode(), 0) == 0
        st = oracle.State(ch, rate)
        st.add(want.reshape(frames, ch)) if frames else None
        r = L.lgo_scan_get_track_result(0, 0.0).contents
        a, b = r.track_loudness, st.loudness()
        assert a == b or abs(a - b) <= 1e-12
        L.lgo_scan_deinit()
