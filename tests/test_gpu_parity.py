"""GPU parity: HIP path (through the C ABI) vs the CPU oracle, same inputs.

Bars (BASELINE.json north_star): integer block counts bit-exact; loudness and
LRA within +-0.01 LU; true peak within +-0.0001.
"""
import json
import math
import os

import numpy as np
import pytest

from loudgain_amd import synth
from tests.gpu_util import (ENERGY_RTOL, check_track, energy_rtol, gating_blocks_from_subblocks,
                            to_dev)

pytestmark = pytest.mark.gpu

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tracks.json")))
ABS_GATE = 10 ** ((-70 + 0.691) / 10)


@pytest.fixture(scope="module")
def scanner():
    from loudgain_amd.device import DeviceScanner
    s = DeviceScanner(0)
    yield s
    s.close()


@pytest.mark.parametrize("g", GOLD, ids=[g["name"] for g in GOLD])
def test_golden_and_oracle(scanner, oracle, g):
    pcm = synth.track_numpy(g["frames"], g["channels"], g["rate"], seed=g["seed"], step_s=g["step_s"])
    ref = oracle.scan_track(pcm, g["rate"])
    (got,), _ = scanner.scan([to_dev(pcm)], g["rate"])
    check_track(got, ref, rate=g["rate"])
    # committed golden vector
    assert (got["n_abs"], got["n_rel"], got["n_st"]) == (g["n_abs"], g["n_rel"], g["n_st"])
    if g["loudness"] != "-inf":
        assert abs(got["loudness"] - g["loudness"]) <= 1e-6
    assert abs(got["lra"] - g["lra"]) <= 1e-6
    assert abs(got["peak"] - g["peak"]) <= 1e-4
    # block energies one by one
    s100 = (g["rate"] + 5) // 10
    z = gating_blocks_from_subblocks(scanner.subblock_energies(0), s100)
    listed = z[z >= ABS_GATE]
    refb = ref["state"].gating_blocks()
    assert len(listed) == len(refb)
    np.testing.assert_allclose(listed, refb, rtol=energy_rtol(g["rate"]))


@pytest.mark.parametrize("frames", [0, 1, 11, 12, 13, 4799, 4800, 19199, 19200, 19201, 24000,
                                    143999, 144000, 144001, 1600 * 64 + 5])
def test_ragged_lengths(scanner, oracle, frames):
    pcm = synth.track_numpy(max(frames, 1), 2, 48000, seed=frames, step_s=0.7)[:frames]
    pcm = np.ascontiguousarray(pcm)
    ref = oracle.scan_track(pcm, 48000)
    import torch
    dev = to_dev(pcm) if frames else torch.zeros((0, 2), dtype=torch.float32, device="cuda")
    (got,), _ = scanner.scan([dev], 48000)
    check_track(got, ref)


@pytest.mark.parametrize("rate,nch", [(48000, 3), (48000, 4), (48000, 5), (48000, 6), (44100, 6),
                                      (48000, 8), (96000, 6), (48000, 16)])
def test_multichannel(scanner, oracle, rate, nch):
    """libebur128 default channel map by index: 4ch L R Ls Rs; 5ch L R C Ls Rs;
    otherwise L R C UNUSED Ls Rs UNUSED...; surrounds weigh 1.41; peaks include
    the unused channels."""
    frames = int(rate * 9.3) + 17
    pcm = synth.track_numpy(frames, nch, rate, seed=100 + nch, step_s=1.7)
    # different level per channel so that a wrong weight or mapping shows
    gains = np.array([1.0, 0.7, 0.5, 1.3, 0.9, 0.6, 1.1, 0.8] * 2)[:nch].astype(np.float32)
    pcm = synth.snap_s16_numpy(pcm * gains[None, :])
    ref = oracle.scan_track(pcm, rate)
    (got,), _ = scanner.scan([to_dev(pcm)], rate)
    check_track(got, ref, rate=rate)
    assert got["sample_peak"] == max(ref["sample_peak"])


@pytest.mark.parametrize("rate,nch", [(11025, 2), (11025, 1), (44056, 2), (8000, 2), (22050, 1),
                                      (11025, 6), (4000, 1)])
def test_odd_rates(scanner, oracle, rate, nch):
    """Rates whose 100 ms sub-block (1103 frames at 11 025 Hz: prime) no compiled
    chunk length divides run on the generic kernel: sub-block boundaries fall
    inside lanes' chunks."""
    frames = int(rate * 21.7) + 3
    pcm = synth.track_numpy(frames, nch, rate, seed=200 + nch, step_s=2.3)
    ref = oracle.scan_track(pcm, rate)
    (got,), _ = scanner.scan([to_dev(pcm)], rate)
    check_track(got, ref, rate=rate)
    s100 = (rate + 5) // 10
    z = gating_blocks_from_subblocks(scanner.subblock_energies(0), s100)
    w = {1: [1], 2: [1, 1], 6: [1, 1, 1, 0, 1.41, 1.41]}[nch]
    listed = z[z >= ABS_GATE]
    refb = ref["state"].gating_blocks()
    assert len(listed) == len(refb)
    np.testing.assert_allclose(listed, refb, rtol=1e-9)


@pytest.mark.parametrize("nch", [17, 24, 33, 64])
def test_wide_streams(scanner, oracle, nch):
    """17..64 channels (ebur128_init accepts up to 64): scanned as groups of 16
    channels; channels >= 6 are EBUR128_UNUSED for loudness but count for peaks."""
    rate = 48000
    frames = int(rate * 5.3)
    pcm = synth.track_numpy(frames, nch, rate, seed=300 + nch, step_s=1.1)
    gains = (0.3 + 0.7 * np.abs(np.sin(np.arange(nch) * 1.7))).astype(np.float32)
    gains[nch - 1] = 1.2   # the loudest samples sit in the last (unused) channel
    pcm = synth.snap_s16_numpy(pcm * gains[None, :])
    ref = oracle.scan_track(pcm, rate)
    (got,), _ = scanner.scan([to_dev(pcm)], rate)
    check_track(got, ref, rate=rate)
    assert got["sample_peak"] == max(ref["sample_peak"])


def test_rate_below_floor(scanner):
    import torch
    from loudgain_amd.device import LoudscanError
    x = torch.zeros((4800, 2), dtype=torch.float32, device="cuda")
    with pytest.raises(LoudscanError):
        scanner.plan([x], 3999)          # ebur128_init accepts it, but its K-filter design is
                                         # meaningless below ~3.4 kHz (1682 Hz shelf vs Nyquist)


def test_silence_and_full_scale(scanner, oracle):
    sil = np.zeros((48000 * 5, 2), np.float32)
    (got,), _ = scanner.scan([to_dev(sil)], 48000)
    assert got["loudness"] == -math.inf and got["lra"] == 0.0 and got["peak"] == 0.0
    assert got["n_abs"] == 0 and got["n_blocks"] == 47
    fs = np.full((48000 * 4, 1), -1.0, np.float32)
    fs[::2] = 32767 / 32768
    ref = oracle.scan_track(fs, 48000)
    (got,), _ = scanner.scan([to_dev(fs)], 48000)
    check_track(got, ref)
    assert got["sample_peak"] == 1.0


def test_true_peak_switch(scanner, oracle):
    pcm = synth.track_numpy(48000 * 8, 2, 48000, seed=5)
    ref = oracle.scan_track(pcm, 48000)
    (got,), _ = scanner.scan([to_dev(pcm)], 48000, true_peak=False)
    check_track(got, ref, tp=False)
    assert got["true_peak"] == 0.0 and got["peak"] == got["sample_peak"]


@pytest.mark.parametrize("seg,warm,chunk", [(1, 2, 0), (3, 2, 25), (7, 3, 50), (1000000, 2, 75),
                                            (5, 2, 0)])
def test_segmentation_invariance(oracle, seg, warm, chunk):
    """Results must not depend on how the track is cut into wave segments."""
    from loudgain_amd.device import DeviceScanner
    pcm = synth.track_numpy(48000 * 47 + 321, 2, 48000, seed=77, step_s=3.0)
    ref = oracle.scan_track(pcm, 48000)
    s = DeviceScanner(0)
    s.set_param("seg_subblocks", seg)
    s.set_param("warm_subblocks", warm)
    s.set_param("chunk", chunk)
    (got,), _ = s.scan([to_dev(pcm)], 48000)
    check_track(got, ref)
    z = gating_blocks_from_subblocks(s.subblock_energies(0), 4800)
    listed = z[z >= ABS_GATE]
    np.testing.assert_allclose(listed, ref["state"].gating_blocks(), rtol=ENERGY_RTOL)
    s.close()


def test_loud_to_quiet_transition_warmup(oracle):
    """Worst case for the segment warm-up: -80 dB material right after 0 dBFS
    low-frequency content, one sub-block per wave."""
    from loudgain_amd.device import DeviceScanner
    fs = 48000
    t = np.arange(fs * 4) / fs
    loud = 0.95 * np.sin(2 * np.pi * 30 * t)
    rng = np.random.default_rng(1)
    quiet = rng.standard_normal(fs * 6) * 1e-4
    x = np.concatenate([loud, quiet])
    pcm = synth.snap_s16_numpy(np.stack([x, x], 1))
    ref = oracle.scan_track(pcm, fs)
    s = DeviceScanner(0)
    s.set_param("seg_subblocks", 1)
    (got,), _ = s.scan([to_dev(pcm)], fs)
    check_track(got, ref)
    z = gating_blocks_from_subblocks(s.subblock_energies(0), 4800)
    refz = ref["state"].gating_blocks()
    listed = z[z >= ABS_GATE]
    assert len(listed) == len(refz)
    np.testing.assert_allclose(listed, refz, rtol=1e-7)
    s.close()


def test_album_parity(scanner, oracle):
    specs = [(48000, 2, 33.3, 11), (48000, 2, 20.0, 12), (44100, 2, 25.5, 13), (48000, 1, 18.2, 14),
             (96000, 2, 8.0, 15), (48000, 2, 0.2, 16)]
    pcms = [synth.track_numpy(int(r * s), c, r, seed=sd, step_s=2.5) for r, c, s, sd in specs]
    # make track 1 much quieter so the album relative gate differs from the per-track ones
    pcms[1] = synth.snap_s16_numpy(pcms[1] * 0.03)
    refs = [oracle.scan_track(p, sp[0]) for p, sp in zip(pcms, specs)]
    states = [r["state"] for r in refs]
    tracks, album = scanner.scan([to_dev(p) for p in pcms], [sp[0] for sp in specs], album=True)
    for got, ref, sp in zip(tracks, refs, specs):
        check_track(got, ref, rate=sp[0])
    det = oracle.gating_detail(states)
    assert album["n_abs"] == det["n_abs"] and album["n_rel"] == det["n_rel"]
    assert abs(album["loudness"] - oracle.album_loudness(states)) <= 1e-6
    assert abs(album["lra"] - oracle.album_lra(states)) <= 1e-6
    assert abs(album["peak"] - max(r["peak"] for r in refs)) <= 1e-4
    assert album["n_st"] == sum(r["n_st"] for r in refs)


def test_errors(scanner):
    import torch
    from loudgain_amd.device import LoudscanError
    x = torch.zeros((4800, 2), dtype=torch.float32, device="cuda")
    with pytest.raises(LoudscanError):
        scanner.plan([x], 15)            # rate < 16 (ebur128_init check)
    with pytest.raises(LoudscanError):
        scanner.plan([(x.data_ptr(), 4800, 0)], 48000)   # 0 channels
    with pytest.raises(LoudscanError):
        scanner.plan([(x.data_ptr(), 4800, 65)], 48000)  # > 64 channels
    with pytest.raises(LoudscanError):
        scanner.plan([(x.data_ptr() + 4, 100, 2)], 48000)  # misaligned
    with pytest.raises(LoudscanError):
        scanner.plan([x.cpu()], 48000)   # host memory is not accepted


def test_ten_minutes_vs_oracle(scanner, oracle):
    pcm = synth.track_numpy(48000 * 600, 2, 48000, seed=99)
    ref = oracle.scan_track(pcm, 48000)
    (got,), _ = scanner.scan([to_dev(pcm)], 48000)
    check_track(got, ref)


def test_many_albums_in_one_plan(scanner, oracle):
    """lgd_plan_albums: one launch, several albums (the library-scan shape of bin/rgbpm2):
    every album equals the oracle's _multiple result over exactly its tracks; albums may be
    empty, hold one short track, or mix rates and channel counts."""
    from loudgain_amd import synth
    specs = [  # (album, rate, ch, seconds, seed, gain)
        (0, 48000, 2, 8.0, 1, 1.0), (0, 48000, 2, 6.5, 2, 0.03), (0, 44100, 2, 9.0, 3, 0.6),
        (1, 48000, 1, 5.0, 4, 1.0),
        # album 2 is empty
        (3, 96000, 2, 4.0, 5, 0.9), (3, 48000, 6, 5.5, 6, 0.2), (3, 48000, 2, 0.25, 7, 1.0),
        (4, 22050, 2, 7.0, 8, 0.5), (4, 48000, 2, 12.0, 9, 1.0),
    ]
    pcms = [synth.snap_s16_numpy(synth.track_numpy(int(r * s), c, r, seed=sd, step_s=1.5) * g)
            for _, r, c, s, sd, g in specs]
    albums = [a for a, *_ in specs]
    tracks, res = scanner.scan([to_dev(p) for p in pcms], [sp[1] for sp in specs], albums=albums)
    assert len(res) == 5
    refs = [oracle.scan_track(p, sp[1]) for p, sp in zip(pcms, specs)]
    for got, ref, sp in zip(tracks, refs, specs):
        check_track(got, ref, rate=sp[1])
    for a in range(5):
        mine = [r for r, sp in zip(refs, specs) if sp[0] == a]
        if not mine:
            assert res[a]["loudness"] == -np.inf and res[a]["lra"] == 0.0 and res[a]["n_abs"] == 0
            continue
        states = [r["state"] for r in mine]
        det = oracle.gating_detail(states)
        assert res[a]["n_abs"] == det["n_abs"] and res[a]["n_rel"] == det["n_rel"]
        assert abs(res[a]["loudness"] - oracle.album_loudness(states)) <= 1e-6
        assert abs(res[a]["lra"] - oracle.album_lra(states)) <= 1e-6
        assert abs(res[a]["peak"] - max(r["peak"] for r in mine)) <= 1e-4
        assert res[a]["n_st"] == sum(r["n_st"] for r in mine)
    # bad album arrays are refused
    from loudgain_amd.device import LoudscanError
    with pytest.raises(LoudscanError):
        scanner.plan([to_dev(pcms[0]), to_dev(pcms[1])], 48000, albums=[1, 0])


def test_album_of_1000_mixed_tracks(scanner, oracle):
    """BASELINE.json configs[3]/[4] in shape: one album of 1000 tracks, mixed rates
    (44.1 / 48 / 96 / 192 kHz) and layouts (mono / stereo / 5.1), durations 0.3 .. 4 s so
    that the oracle finishes in seconds.  Every track and the album against the oracle."""
    rng = np.random.default_rng(1000)
    rates = [44100, 48000, 96000, 192000]
    chans = [1, 2, 2, 2, 6]
    protos = {}
    specs = []
    for i in range(1000):
        rate = rates[int(rng.integers(len(rates)))]
        ch = chans[int(rng.integers(len(chans)))]
        secs = float(rng.uniform(0.3, 4.0)) * (0.5 if rate >= 96000 else 1.0)
        gain = float(10.0 ** (rng.uniform(-40.0, 0.0) / 20.0))
        specs.append((rate, ch, int(rate * secs), gain))
    pcms = []
    for rate, ch, frames, gain in specs:
        key = (rate, ch)
        if key not in protos:   # one 4 s prototype per (rate, layout); tracks are gain-scaled cuts of it
            protos[key] = synth.track_numpy(int(rate * 4.0), ch, rate, seed=rate % 1000 + ch, step_s=0.7)
        off = int(rng.integers(0, protos[key].shape[0] - frames + 1))
        pcms.append(synth.snap_s16_numpy(protos[key][off:off + frames] * gain))
    tracks, album = scanner.scan([to_dev(p) for p in pcms], [s[0] for s in specs], album=True)
    refs = [oracle.scan_track(p, s[0]) for p, s in zip(pcms, specs)]
    for got, ref, s in zip(tracks, refs, specs):
        check_track(got, ref, rate=s[0])
    states = [r["state"] for r in refs]
    det = oracle.gating_detail(states)
    assert album["n_abs"] == det["n_abs"] and album["n_rel"] == det["n_rel"]
    assert abs(album["loudness"] - oracle.album_loudness(states)) <= 1e-6
    assert abs(album["lra"] - oracle.album_lra(states)) <= 1e-6
    assert abs(album["peak"] - max(r["peak"] for r in refs)) <= 1e-4
    assert album["n_st"] == sum(r["n_st"] for r in refs)


def test_max_momentary_and_shortterm(scanner):
    """Loudest 400 ms / 3 s window on the 100 ms grid: against a numpy sliding sum over the
    (oracle-checked) 100 ms sub-block energies, and EBU Tech 3341's -23 dBFS 1 kHz stereo
    sine, for which M = S = I = -23.0 LUFS."""
    rate = 48000
    pcm = synth.track_numpy(rate * 21, 2, rate, seed=31, step_s=2.5)
    short = synth.track_numpy(int(rate * 1.7), 2, rate, seed=32)            # has 400 ms windows, no 3 s window
    t = np.arange(rate * 8)
    sine = np.repeat((10 ** (-23 / 20) * np.sin(2 * np.pi * 1000 * t / rate)).astype(np.float32)[:, None], 2, 1)
    tracks, _ = scanner.scan([to_dev(pcm), to_dev(short), to_dev(sine)], rate)
    for i, (x, tr) in enumerate(zip((pcm, short, sine), tracks)):
        e = scanner.subblock_energies(i)                                   # sum_c w_c sum y^2 per 100 ms
        s100 = rate // 10
        m = np.convolve(e, np.ones(4), "valid") / (4 * s100) if e.size >= 4 else np.zeros(0)
        s = np.convolve(e, np.ones(30), "valid") / (30 * s100) if e.size >= 30 else np.zeros(0)
        want_m = 10 * np.log10(m.max()) - 0.691 if m.size else -np.inf
        want_s = 10 * np.log10(s.max()) - 0.691 if s.size else -np.inf
        assert abs(tr["max_momentary"] - want_m) <= 1e-9 or tr["max_momentary"] == want_m
        assert abs(tr["max_shortterm"] - want_s) <= 1e-9 or tr["max_shortterm"] == want_s
    assert tracks[1]["max_shortterm"] == -np.inf and np.isfinite(tracks[1]["max_momentary"])
    assert abs(tracks[2]["max_momentary"] + 23.0) <= 0.1 and abs(tracks[2]["max_shortterm"] + 23.0) <= 0.1
    assert abs(tracks[2]["loudness"] + 23.0) <= 0.1


def test_full_size_properties():
    """BASELINE.json's full size (60 min, 48 kHz stereo f32 = 1.38 GB), where the oracle
    would take minutes: size-independent properties of the measurement instead.
      * the first 90 s agree with the oracle (same buffer, block energies are causal);
      * halving the signal (exact in binary floating point) scales every 100 ms energy by
        exactly 1/4 (bit for bit), the ungated maxima by exactly 10 log10(1/4), the peaks by 1/2;
      * block energies do not depend on how the track is cut into segments / tiles
        (sub-block energies within 1e-10 relative for three other segmentations / tile lengths);
      * an album of the track with itself has the track's loudness and range;
      * 4x true peak on/off changes nothing but the peak fields."""
    import torch
    from loudgain_amd.device import DeviceScanner
    from oracle import lgoracle
    rate, ch = 48000, 2
    frames = 60 * 60 * rate
    pcm = synth.track_torch(frames, ch, rate, seed=2026, device="cuda")
    sc = DeviceScanner(0)
    (full,), _ = sc.scan([pcm], rate, true_peak=True)
    e_ref = sc.subblock_energies(0)
    assert full["n_blocks"] == 36000 - 3 and full["n_st_blocks"] == (36000 - 30) // 10 + 1
    # prefix against the oracle
    head = pcm[: 90 * rate].cpu().numpy()
    ref = lgoracle.scan_track(head, rate)
    (hd,), _ = sc.scan([pcm[: 90 * rate]], rate, true_peak=True)
    check_track(hd, ref)
    # causal: the prefix has the same energies (another segmentation: equal to rounding)
    np.testing.assert_allclose(sc.subblock_energies(0), e_ref[:900], rtol=1e-12, atol=0)
    # exact scaling
    half = pcm * 0.5
    (hf,), _ = sc.scan([half], rate, true_peak=True)
    # (the absolute gate is absolute: gate counts and the integrated value may change, the
    # block energies and the ungated maxima may not)
    assert abs((full["max_momentary"] - hf["max_momentary"]) - 10 * np.log10(4.0)) <= 1e-12
    assert abs((full["max_shortterm"] - hf["max_shortterm"]) - 10 * np.log10(4.0)) <= 1e-12
    assert hf["sample_peak"] == full["sample_peak"] * 0.5 and abs(hf["true_peak"] - full["true_peak"] * 0.5) <= 1e-7
    np.testing.assert_array_equal(sc.subblock_energies(0), e_ref * 0.25)
    del half
    # segmentation independence, bit for bit
    for seg, chunk in ((9, 75), (100, 25), (1000000, 50)):
        s2 = DeviceScanner(0)
        s2.set_param("seg_subblocks", seg)
        s2.set_param("chunk", chunk)
        (g2,), _ = s2.scan([pcm], rate, true_peak=False)
        e2 = s2.subblock_energies(0)
        # different tilings round differently: <= ~1e-11 relative, on quiet blocks right after loud ones
        np.testing.assert_allclose(e2, e_ref, rtol=1e-10, atol=0)
        assert g2["n_abs"] == full["n_abs"] and g2["n_rel"] == full["n_rel"]
        assert abs(g2["loudness"] - full["loudness"]) <= 1e-10 and abs(g2["lra"] - full["lra"]) <= 1e-10
        assert g2["sample_peak"] == full["sample_peak"] and g2["true_peak"] == 0.0
        s2.close()
    # album of the track with itself
    (a, b), alb = sc.scan([pcm, pcm], rate, true_peak=False, album=True)
    assert a == b and a["n_abs"] == full["n_abs"]
    assert alb["n_abs"] == 2 * full["n_abs"] and alb["n_rel"] == 2 * full["n_rel"]
    assert abs(alb["loudness"] - full["loudness"]) <= 1e-12 and abs(alb["lra"] - full["lra"]) <= 1e-12
    assert alb["peak"] == a["peak"]
    sc.close()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("seed", range(11, 11 + int(os.environ.get("LGD_FUZZ_SEEDS", "8"))))
def test_fuzz_mixed_plans(scanner, oracle, seed):
    """Random plans: 12 tracks each with random rate, layout (1..8 channels, now and then a
    wide stream), length (from a few frames to ~25 s, clustered around the block-length
    edges 100 ms / 400 ms / 3 s), gain and content (noise, LF tone, impulses, silence),
    true peak on or off, cut into a random number of albums (some empty)."""
    rng = np.random.default_rng(seed)
    rates = [8000, 11025, 22050, 32000, 44100, 48000, 88200, 96000, 176400, 192000]
    specs, pcms = [], []
    for i in range(12):
        rate = rates[int(rng.integers(len(rates)))]
        ch = int(rng.choice([1, 2, 2, 2, 3, 5, 6, 8, 2, 2, 12, 17, 33]))
        edge = float(rng.choice([0.0, 0.1, 0.4, 3.0, 3.1]))
        secs = edge + float(rng.choice([0.0, 1.0 / rate, 0.003, 0.0999, 0.1001, 0.37, 1.234, 7.7, 25.0])) \
            * (0.3 if rate > 96000 or ch > 2 else 1.0) * (0.3 if ch > 8 else 1.0)
        frames = max(0, int(round(secs * rate)))
        kind = int(rng.integers(4))
        t = np.arange(frames)[:, None] / rate
        if kind == 0:
            x = rng.standard_normal((frames, ch)) * 0.2
        elif kind == 1:
            x = 0.7 * np.sin(2 * np.pi * float(rng.uniform(20, 200)) * t + np.arange(ch)[None, :])
        elif kind == 2:
            x = np.zeros((frames, ch))
            if frames:
                x[rng.integers(0, frames, size=min(frames, 20)), :] = rng.uniform(-1, 1, size=(min(frames, 20), 1))
        else:
            x = np.zeros((frames, ch))
        x = x * float(10.0 ** (rng.uniform(-60.0, 0.0) / 20.0))
        pcms.append(synth.snap_s16_numpy(x.astype(np.float32).reshape(frames, ch)))
        specs.append((rate, ch, frames))
    tp = bool(rng.integers(2))
    n_albums = int(rng.integers(1, 6))
    albums = sorted(int(a) for a in rng.integers(0, n_albums, size=12))
    albums[-1] = max(albums[-1], n_albums - 1) if rng.integers(2) else albums[-1]   # sometimes trailing empties
    # every track resident as f32 or, at random, as the interleaved int16 the reference feeds libebur128 (LGD_PCM_S16)
    as_s16 = [bool(b) for b in np.random.default_rng(1000 + seed).integers(0, 2, size=12)]

    def dev(p, s16):
        if not p.size:
            return torch_empty(p.shape[1], s16)
        return to_dev(np.rint(p.astype(np.float64) * 32768.0).astype(np.int16)) if s16 else to_dev(p)
    tracks, res = scanner.scan([dev(p, f) for p, f in zip(pcms, as_s16)],
                               [s[0] for s in specs], true_peak=tp, albums=albums)
    refs = [oracle.scan_track(p, s[0]) for p, s in zip(pcms, specs)]
    for got, ref, s in zip(tracks, refs, specs):
        check_track(got, ref, tp=tp, rate=s[0], lf_tones=True)
    assert len(res) == max(albums) + 1
    if tp:   # the pruning of the interpolator is exact on every one of these plans too: same floats with it off
        scanner.set_param("tp_prune", 0)
        try:
            tracks0, res0 = scanner.scan([dev(p, not f) for p, f in zip(pcms, as_s16)],   # (and the other format each)
                                         [s[0] for s in specs], true_peak=True, albums=albums)
        finally:
            scanner.set_param("tp_prune", 1)
        for a, b in zip(tracks, tracks0):
            assert a["peak"] == b["peak"] and a["true_peak"] == b["true_peak"] and a["sample_peak"] == b["sample_peak"]
        for a, b in zip(res, res0):
            assert a["peak"] == b["peak"]
    for a, album in enumerate(res):
        states = [r["state"] for r, al in zip(refs, albums) if al == a]
        if not states:
            assert album["n_abs"] == 0 and album["loudness"] == -np.inf and album["lra"] == 0.0
            continue
        det = oracle.gating_detail(states)
        assert album["n_abs"] == det["n_abs"] and album["n_rel"] == det["n_rel"]
        want = oracle.album_loudness(states)
        # (tones of tens of Hz at 176 / 192 kHz: see gpu_util.energy_rtol; the bar is 0.01 LU)
        assert album["loudness"] == want or abs(album["loudness"] - want) <= 2e-5
        assert abs(album["lra"] - oracle.album_lra(states)) <= 2e-5
        if tp:
            assert abs(album["peak"] - max(r["peak"] for r, al in zip(refs, albums) if al == a)) <= 1e-4


def torch_empty(ch, s16=False):
    import torch
    return torch.zeros((0, ch), dtype=torch.int16 if s16 else torch.float32, device="cuda")


def test_large_album_range_selection(scanner):
    """The album's loudness range is an exact order statistic over ALL listed 3 s blocks
    (tens of thousands here: far more than the LRA kernel stages at once).  40 ten-minute
    tracks = one base track at 40 power-of-two-free gains; expected values from numpy over the
    100 ms energies of each scanned track."""
    import torch
    from loudgain_amd.album import album_from_partials
    rate = 48000
    base = synth.track_torch(10 * 60 * rate, 2, rate, seed=404, step_s=7.0, device="cuda")
    gains = [0.25 + 0.018 * i for i in range(40)]
    pcms = [base * g for g in gains]
    tracks, album = scanner.scan(pcms, rate, true_peak=False, album=True)
    st_all, zsum, zn, z_all = [], 0.0, 0, []
    for i in range(len(pcms)):
        e = scanner.subblock_energies(i)
        z = (e[:-3] + e[1:-2] + e[2:-1] + e[3:]) / (4.0 * 4800)
        z_all.append(z[z >= ABS_GATE])
        k = np.arange((len(e) - 30) // 10 + 1)
        st = np.array([e[10 * j:10 * j + 30].sum() for j in k]) / (30.0 * 4800)
        st_all.append(st[st >= ABS_GATE])
    z_all = np.concatenate(z_all)
    thr = 0.1 * z_all.sum() / len(z_all)
    sel = z_all[z_all >= thr]
    want = album_from_partials(sel.sum(), len(sel), np.concatenate(st_all), max(t["peak"] for t in tracks))
    assert album["n_st"] == sum(len(s) for s in st_all) > 15000
    assert album["n_abs"] == len(z_all) and album["n_rel"] == len(sel)
    assert abs(album["loudness"] - want["loudness"]) <= 1e-9
    assert abs(album["lra"] - want["lra"]) <= 1e-9
    del pcms, base
    torch.cuda.empty_cache()


def test_long_track_range_selection(scanner, oracle):
    """One track with more than 8192 listed 3 s blocks (2.4 h at 8 kHz mono): its own loudness range
    goes through the multi-workgroup selection kernels (lgd_lra_big_*), like the album of a large
    plan; same exact order statistics as the oracle's sort (ebur128_loudness_range)."""
    rate = 8000
    pcm = synth.track_numpy(rate * 8700, 1, rate, seed=88, step_s=41.0)
    ref = oracle.scan_track(pcm, rate)
    (got,), album = scanner.scan([to_dev(pcm)], rate, album=True)
    assert got["n_st"] == ref["n_st"] and got["n_st_blocks"] > 8192 and ref["n_st"] > 1000
    check_track(got, ref, rate=rate)
    assert abs(got["lra"] - ref["lra"]) <= 1e-9
    assert abs(album["lra"] - ref["lra"]) <= 1e-9 and album["n_st"] == ref["n_st"]


@pytest.mark.parametrize("overlap", [0, 1])
def test_pcm_buffer_reuse_between_executes(oracle, overlap):
    """A caller that refills ONE device buffer between scans (the ingest loop that replaces
    scan.c:225-250).  Default ("overlap" 0): every scan runs on the caller's stream, so a copy
    enqueued on that stream behind lgd_execute is ordered behind the scan like behind any kernel.
    "overlap" 1: scans may run on the engine's own stream; the writer must be ordered behind them
    with lgd_join first."""
    import torch
    from loudgain_amd.device import DeviceScanner
    rate = 48000
    n = rate * 240
    src = [synth.track_numpy(n, 2, rate, seed=300 + i, step_s=3.0 + i) for i in range(4)]
    refs = [oracle.scan_track(p, rate) for p in src]
    host = [torch.from_numpy(p).pin_memory() for p in src]
    sc = DeviceScanner(0)
    sc.set_param("overlap", overlap)
    stream = torch.cuda.Stream()
    buf = torch.empty((n, 2), dtype=torch.float32, device="cuda")
    sc.plan([buf], rate, true_peak=True)
    got = []
    with torch.cuda.stream(stream):
        for i in range(4):
            if overlap:
                sc.join(stream)                      # the scans of the previous rounds have read `buf`
            buf.copy_(host[i], non_blocking=True)    # refill, no host synchronisation
            sc.execute(stream)
            if i % 2 == 1:                           # results are read only every other round
                (r,), _ = sc.fetch()
                got.append((i, r))
    for i, r in got:
        check_track(r, refs[i])
    sc.close()


@pytest.mark.parametrize("rate,nch", [(48000, 3), (48000, 4), (44100, 6), (48000, 6), (96000, 8), (48000, 12)])
def test_channel_pair_workgroups_on_any_layout(oracle, rate, nch):
    """"strided" 2: every 3+ channel stream as stereo-shaped workgroups, one per channel pair of the
    interleaved stream (the default uses them for 5 / 7 / 17+ channels only, where they are faster).
    Same results as the many-plane kernels: both against the oracle, block energies against each other."""
    from loudgain_amd.device import DeviceScanner
    frames = int(rate * 7.3) + 11
    pcm = synth.track_numpy(frames, nch, rate, seed=500 + nch, step_s=1.3)
    gains = np.array([1.0, 0.7, 0.5, 1.3, 0.9, 0.6, 1.1, 0.8] * 2)[:nch].astype(np.float32)
    pcm = synth.snap_s16_numpy(pcm * gains[None, :])
    ref = oracle.scan_track(pcm, rate)
    out = []
    for mode in (2, 0):
        sc = DeviceScanner(0)
        sc.set_param("strided", mode)
        (got,), _ = sc.scan([to_dev(pcm)], rate)
        check_track(got, ref, rate=rate)
        sp, tp = sc.channel_peaks(0, nch)
        np.testing.assert_allclose(tp, np.asarray(ref["true_peak"]), atol=1e-4, rtol=0)
        out.append((got, sc.subblock_energies(0), sp, tp))
        sc.close()
    (a, ea, spa, tpa), (b, eb, spb, tpb) = out
    np.testing.assert_allclose(ea, eb, rtol=energy_rtol(rate))   # (other chunk length: other rounding)
    assert np.array_equal(spa, spb) and np.array_equal(tpa, tpb)
    assert (a["n_abs"], a["n_rel"], a["n_st"]) == (b["n_abs"], b["n_rel"], b["n_st"])


@pytest.mark.parametrize("rate,nch", [(48000, 3), (48000, 6), (44100, 6), (96000, 6), (192000, 6), (48000, 9), (32000, 12),
                                      (48000, 7), (48000, 8)])
def test_channel_triple_workgroups(oracle, rate, nch):
    """"strided" 3: streams whose channel count divides by three as three-wave workgroups, one per channel
    TRIPLE of the interleaved stream (the default does this for 5.1: two triples, the LFE in the second);
    other counts fall back to what the default does.  Same results as the many-plane kernels ("strided" 0)."""
    from loudgain_amd.device import DeviceScanner
    frames = int(rate * 6.1) + 29
    pcm = synth.track_numpy(frames, nch, rate, seed=700 + nch, step_s=1.1)
    gains = np.array([1.0, 0.7, 0.5, 1.3, 0.9, 0.6, 1.1, 0.8] * 2)[:nch].astype(np.float32)
    pcm = synth.snap_s16_numpy(pcm * gains[None, :])
    ref = oracle.scan_track(pcm, rate)
    out = []
    for mode in (3, 0):
        sc = DeviceScanner(0)
        sc.set_param("strided", mode)
        (got,), _ = sc.scan([to_dev(pcm)], rate)
        check_track(got, ref, rate=rate)
        sp, tp = sc.channel_peaks(0, nch)
        np.testing.assert_allclose(tp, np.asarray(ref["true_peak"]), atol=1e-4, rtol=0)
        out.append((got, sc.subblock_energies(0), sp, tp))
        sc.close()
    (a, ea, spa, tpa), (b, eb, spb, tpb) = out
    np.testing.assert_allclose(ea, eb, rtol=energy_rtol(rate))
    assert np.array_equal(spa, spb) and np.array_equal(tpa, tpb)
    assert (a["n_abs"], a["n_rel"], a["n_st"]) == (b["n_abs"], b["n_rel"], b["n_st"])


def test_merged_launches_of_a_mixed_plan(oracle):
    """"merge_launches" 1 (default): the (rate, channels) groups of a plan that run the same kernel instance
    -- 48 / 96 / 192 kHz stereo at C = 75, the 5.1 triples at C = 50 ... -- go out as ONE launch, constants per
    segment, 192 kHz segments (no interpolator) beside interpolating ones.  Same results as one launch per
    group ("merge_launches" 0), track by track, and both equal the oracle."""
    from loudgain_amd.device import DeviceScanner
    layout = [(48000, 2, 7.3), (96000, 2, 5.1), (192000, 2, 3.7), (44100, 2, 6.9), (48000, 1, 9.1), (192000, 1, 4.3),
              (96000, 1, 6.1), (48000, 6, 4.9), (96000, 6, 3.1), (192000, 6, 2.3), (44100, 6, 3.3), (48000, 2, 0.35),
              (32000, 2, 5.5), (22050, 1, 8.1), (11025, 2, 6.0), (48000, 3, 4.1), (96000, 3, 2.9)]
    pcms, rates = [], []
    for i, (rate, nch, secs) in enumerate(layout):
        pcms.append(synth.track_numpy(int(rate * secs) + 7 * i, nch, rate, seed=900 + i, step_s=0.9))
        rates.append(rate)
    devs = [to_dev(p) for p in pcms]
    out = []
    for merge in (1, 0):
        sc = DeviceScanner(0)
        sc.set_param("merge_launches", merge)
        res, _ = sc.scan(devs, rates)
        peaks = [sc.channel_peaks(i, p.shape[1]) for i, p in enumerate(pcms)]
        energies = [sc.subblock_energies(i) for i in range(len(pcms))]
        out.append((res, peaks, energies))
        sc.close()
    (ra, pa, ea), (rb, pb, eb) = out
    for i, (p, rate) in enumerate(zip(pcms, rates)):
        check_track(ra[i], oracle.scan_track(p, rate), rate=rate)
        for k in ("loudness", "lra", "peak", "true_peak", "sample_peak", "n_abs", "n_rel", "n_st"):
            assert ra[i][k] == rb[i][k] or (ra[i][k] != ra[i][k] and rb[i][k] != rb[i][k]), (i, k)
        assert np.array_equal(pa[i][0], pb[i][0]) and np.array_equal(pa[i][1], pb[i][1])
        assert np.array_equal(ea[i], eb[i])      # sub-block energies do not depend on the segmentation


def test_non_finite_samples_stay_in_their_track(oracle):
    """The reference never sees NaN / Inf (scan.c:414 resamples everything to S16), the f32 entry point can:
    such a track's own numbers are unspecified, but the scan must return and the other tracks of the plan
    must come out exactly as they do alone."""
    from loudgain_amd.device import DeviceScanner
    rate = 48000
    good = [synth.track_numpy(rate * 11 + 5, 2, rate, seed=41, step_s=1.1), synth.track_numpy(rate * 7, 2, rate, seed=42, step_s=0.8)]
    bad = synth.track_numpy(rate * 9, 2, rate, seed=43, step_s=1.0).copy()
    bad[rate * 2 + 17, 0] = np.nan
    bad[rate * 5 + 3, 1] = np.inf
    bad[rate * 6 + 1, 0] = -np.inf
    sc = DeviceScanner(0)
    res, _ = sc.scan([to_dev(good[0]), to_dev(bad), to_dev(good[1])], rate)
    solo = [sc.scan([to_dev(g)], rate)[0][0] for g in good]
    sc.close()
    for got, want, pcm in ((res[0], solo[0], good[0]), (res[2], solo[1], good[1])):
        for k in ("loudness", "lra", "peak", "true_peak", "sample_peak", "n_abs", "n_rel", "n_st"):
            assert got[k] == want[k], k
        check_track(got, oracle.scan_track(pcm, rate))
    assert res[1]["n_blocks"] == 87       # 9 s: the block grid does not depend on the samples


@pytest.mark.parametrize("rate,nch", [(48000, 2), (44100, 2), (96000, 2), (48000, 1), (48000, 6), (88200, 1)])
def test_limited_programme_dense_true_peak_rows(scanner, oracle, rate, nch):
    """Loud, hard-limited material: the sample peak is reached everywhere, (nearly) every chunk's bound exceeds it and
    lgd_tp_kernel walks whole rows (dense path: 4x at 44.1 / 48 kHz with window steps of 7 / 5 frames, 2x at 88.2 / 96 kHz) --
    per channel against the oracle's interp_process restatement (oracle/lg_oracle.c:156-179), and against the same scan with
    the dense path switched off ("tp_dense_min" 65: chunk by chunk) bit for bit."""
    from loudgain_amd.device import DeviceScanner
    pcm = synth.limited_numpy(rate * 9 + 1234, nch, rate, seed=rate % 89 + nch)
    ref = oracle.scan_track(pcm, rate)
    sc = DeviceScanner(0)
    out = []
    for dm in (32, 65):
        sc.set_param("tp_dense_min", dm)
        (got,), _ = sc.scan([to_dev(pcm)], rate)
        sp, tp = sc.channel_peaks(0, nch)
        out.append((got, list(sp), list(tp)))
    sc.close()
    (got, sp, tp), (got2, sp2, tp2) = out
    check_track(got, ref, rate=rate)
    assert sp == ref["sample_peak"]
    for c in range(nch):
        assert abs(tp[c] - ref["true_peak"][c]) <= 1e-4, (c, tp[c], ref["true_peak"][c])
    assert tp == tp2 and sp == sp2 and got["peak"] == got2["peak"]
    assert max(tp) > max(sp) * 1.05        # the limiter's flat tops overshoot between the samples: the interpolator matters here


@pytest.mark.parametrize("nch,width", [(5, 3), (7, 4), (8, 4), (9, 4), (16, 4), (21, 4)])
def test_overlapping_channel_sets(oracle, nch, width):
    """Layouts the planar kernels are weak on go out as channel triples (5 ch: 0-2 | 2-4) or quads (7: 0-3 | 3-6; 7.1: 0-3 | 4-7;
    9: 0-3 | 4-7 | 5-8; 16+), sibling workgroups of one launch; a set that overlaps its neighbour leaves the shared channels to
    it where it keeps two to filter (LgdSeg::skip_mask), else computes them again.  Per channel against the oracle, and equal to
    the pairs form ("strided" 2) and the planar / run-time-channel kernels ("strided" 0): energies and peaks are per channel,
    so all three must agree exactly."""
    from loudgain_amd.device import DeviceScanner
    rate = 48000
    pcm = synth.track_numpy(rate * 7 + 321, nch, rate, seed=70 + nch, step_s=1.3)
    pcm *= (1.0 - 0.03 * np.arange(nch, dtype=np.float32))[None, :]
    pcm = synth.snap_s16_numpy(pcm)
    ref = oracle.scan_track(pcm, rate)
    res = {}
    for st in (1, 2, 0):
        sc = DeviceScanner(0)
        sc.set_param("strided", st)
        (got,), _ = sc.scan([to_dev(pcm)], rate)
        sp, tp = sc.channel_peaks(0, nch)
        res[st] = (got, list(sp), list(tp), sc.plan_info()["chunk"], sc.subblock_energies(0))
        sc.close()
    assert res[1][3] == 50 and res[2][3] == 75                                  # triples / quads at C = 50, pairs at C = 75
    for st, (got, sp, tp, _, _) in res.items():
        check_track(got, ref, rate=rate)
        assert sp == ref["sample_peak"], (st, sp, ref["sample_peak"])
        for c in range(nch):
            assert abs(tp[c] - ref["true_peak"][c]) <= 1e-4, (st, c)
    assert res[1][1] == res[2][1] == res[0][1] and res[1][2] == res[2][2] == res[0][2]
    np.testing.assert_allclose(res[1][4], res[0][4], rtol=energy_rtol(rate))
    assert width in (3, 4)
