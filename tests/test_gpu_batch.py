"""GPU: library scan (loudgain_amd.batch) -- a folder tree of WAV albums scanned in
batches, every album compared with the oracle's restatement of scan.c run over
exactly that folder's files (one loudgain -a call per folder in bin/rgbpm2)."""
import ctypes as C
import os

import numpy as np
import pytest

from loudgain_amd import synth
from tests.test_gpu_scan_api import close, write_wav

pytestmark = pytest.mark.gpu


def _library(root):
    albums = {
        "artist1/album1": [(48000, 2, 9.0, "s16", 1.0), (48000, 2, 7.0, "s16", 0.05), (48000, 2, 11.5, "s24", 0.6)],
        "artist1/album2": [(44100, 2, 8.0, "s16", 1.0), (44100, 1, 5.0, "f32", 0.8)],
        "artist2/single": [(48000, 2, 6.0, "s16", 1.0)],
        "artist2/hires": [(96000, 2, 4.0, "s32", 0.9), (96000, 2, 5.0, "s32", 0.4)],
        "artist2/short": [(48000, 2, 0.2, "u8", 1.0), (48000, 2, 3.5, "ext16", 1.0)],
        "va [compilations]": [(48000, 2, 3.0, "s16", 1.0)],          # excluded by rgbpm2's rule
    }
    seed = 100
    for folder, tracks in albums.items():
        os.makedirs(os.path.join(root, folder), exist_ok=True)
        for k, (rate, ch, secs, kind, g) in enumerate(tracks):
            seed += 1
            pcm = synth.snap_s16_numpy(synth.track_numpy(int(rate * secs), ch, rate, seed=seed, step_s=1.3) * g)
            write_wav(os.path.join(root, folder, "%02d.wav" % (k + 1)), pcm, rate, kind)
    open(os.path.join(root, "artist1/album1/cover.jpg"), "wb").close()
    return albums


@pytest.mark.parametrize("batch_samples", [1 << 29, 1_500_000])   # one batch / several batches
def test_library_scan_matches_oracle_sessions(oracle, tmp_path, batch_samples):
    from loudgain_amd import batch
    root = str(tmp_path)
    albums = _library(root)
    res, st = batch.scan_library([root], batch_samples=batch_samples, reader_threads=2)
    assert st["excluded_folders"] == 1 and st["tasks"] == len(albums) - 1 == len(res)
    assert st["tracks"] == sum(len(v) for k, v in albums.items() if "compilations" not in k)
    L = oracle.lib()
    for album in res:
        files = [t["file"] for t in album["tracks"]]
        assert files == sorted(files) and all(os.path.dirname(f) == album["folder"] for f in files)
        L.lgo_scan_init(len(files))
        for i, f in enumerate(files):
            assert L.lgo_scan_file(f.encode(), i) == 0
        for i, t in enumerate(album["tracks"]):
            o = L.lgo_scan_get_track_result(i, 0.0).contents
            L.lgo_scan_set_album_result(C.byref(o), 0.0)
            assert close(t["loudness"], o.track_loudness, 1e-6), (t["file"], t["loudness"], o.track_loudness)
            assert close(t["lra"], o.track_loudness_range, 1e-6)
            assert close(t["peak"], o.track_peak, 1e-4)
            assert t["codec_id"] == o.codec_id
            assert close(album["album"]["loudness"], o.album_loudness, 1e-6)
            assert close(album["album"]["lra"], o.album_loudness_range, 1e-6)
            assert close(album["album"]["peak"], o.album_peak, 1e-4)
            if np.isfinite(o.track_loudness):
                # loudgain -a -k: gains after clip prevention at -1 dBTP (loudgain.c:323-379)
                from loudgain_amd.gain import apply_clip_logic
                want = apply_clip_logic(o.track_gain, o.track_peak, o.album_gain, o.album_peak, do_album=True,
                                        no_clip=True, max_true_peak_level=-1.0)
                assert abs(t["gain"] - want["track_gain"]) <= 1e-4
                assert abs(album["album"]["gain"] - want["album_gain"]) <= 1e-4
        L.lgo_scan_deinit()


def test_wav_probe_and_read(tmp_path):
    from loudgain_amd import scan
    rate, ch = 44100, 2
    pcm = synth.snap_s16_numpy(synth.track_numpy(rate * 2, ch, rate, seed=5))
    p = write_wav(str(tmp_path / "x.wav"), pcm, rate, "s24")
    wi = scan.scan_wav_probe(p)
    assert (wi["channels"], wi["rate"], wi["bits"], wi["frames"], wi["codec_id"]) == (ch, rate, 24, rate * 2, 0x1000C)
    out = np.zeros((wi["frames"], ch), np.int16)
    assert scan.scan_wav_read_s16(p, out.ctypes.data, wi["frames"]) == wi["frames"]
    assert np.array_equal(out, np.round(pcm * 32768).astype(np.int16))
    # truncated file: silently shortened like the packet loop of scan.c:229-240
    data = open(p, "rb").read()
    open(p, "wb").write(data[:-6 * 1000])
    assert scan.scan_wav_read_s16(p, out.ctypes.data, wi["frames"]) == wi["frames"] - 1000
    with pytest.raises(OSError):
        scan.scan_wav_probe(str(tmp_path / "missing.wav"))
    open(str(tmp_path / "junk.wav"), "wb").write(b"not a wave file at all")
    with pytest.raises(OSError):
        scan.scan_wav_probe(str(tmp_path / "junk.wav"))
