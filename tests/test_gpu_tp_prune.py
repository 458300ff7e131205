"""True-peak pruning is exact: the 4x / 2x interpolator (ebur128_check_true_peak / interp_process,
reached from /root/reference/src/scan.c:448) is evaluated only for chunks whose bound
L1 * max|x| exceeds the channel's sample peak over the whole track (lgd_peak_reduce_kernel ->
lgd_tp_kernel); what is reported is max(true peak, sample peak) as ebur128_true_peak does.  These
tests scan the same PCM with the pruning on ("tp_prune" 1, default) and off (every chunk
evaluated) and demand bit-identical peaks -- per track and per channel -- on material chosen to
make pruning easy (quiet passages beside a loud one), useless (constant full-scale amplitude) and
treacherous (the loudest inter-sample peak sits in an otherwise quiet window, far from the loudest
sample).
"""
import numpy as np
import pytest

from loudgain_amd import synth
from tests.gpu_util import PEAK_TOL, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scanner():
    from loudgain_amd.device import DeviceScanner
    s = DeviceScanner(0)
    yield s
    s.set_param("tp_prune", 1)
    s.close()


def _material(kind, frames, nch, rate, seed):
    rng = np.random.default_rng(seed)
    n = np.arange(frames)
    if kind == "steps":            # the benchmark's programme material
        return synth.track_numpy(frames, nch, rate, seed=seed, step_s=1.3)
    if kind == "limited":          # hard-limited programme: the sample peak is reached everywhere (dense rows)
        return synth.limited_numpy(frames, nch, rate, seed=seed)
    if kind == "adversarial":      # fs/4 sine sampled on its peaks: sample peak == true peak in
        x = 0.8 * np.sin(2 * np.pi * n / 4.0 + np.pi / 2)   # every window, nothing can be skipped
        return synth.snap_s16_numpy(np.repeat(x[:, None], nch, 1).astype(np.float32))
    if kind == "square":           # full-scale square wave: large overshoot everywhere
        x = np.where((n // 7) % 2 == 0, 0.999, -0.999)
        return synth.snap_s16_numpy(np.repeat(x[:, None], nch, 1).astype(np.float32))
    if kind == "hidden":
        # a loud but smooth passage (true peak ~ sample peak 0.5) and, elsewhere, a short fs/4
        # 45-degree burst whose samples reach only 0.45 but whose inter-sample peak is 0.636:
        # the track's true peak hides in a window whose samples are BELOW the loudest sample
        x = 0.02 * rng.standard_normal((frames, nch))
        a, b = frames // 5, frames // 5 + rate // 2
        x[a:b] += (0.5 * np.sin(2 * np.pi * 440.0 * n[a:b] / rate))[:, None]
        c = (3 * frames) // 4
        x[c:c + 64] = (0.636 * np.sin(2 * np.pi * n[:64] / 4.0 + np.pi / 4))[:, None]
        return synth.snap_s16_numpy(x.astype(np.float32))
    if kind == "music":            # slow dynamics + transients, different per channel
        env = 0.05 + 0.45 * (0.5 + 0.5 * np.sin(2 * np.pi * n / (rate * 2.7)))[:, None]
        x = env * rng.standard_normal((frames, nch)) * 0.5
        hits = rng.integers(0, frames - 8, 40)
        for h in hits:
            x[h:h + 4, rng.integers(0, nch)] += rng.uniform(-0.9, 0.9)
        return synth.snap_s16_numpy(np.clip(x, -1, 1).astype(np.float32))
    if kind == "impulses":         # isolated full-scale samples in silence
        x = np.zeros((frames, nch), np.float32)
        pos = rng.integers(0, frames, 25)
        x[pos, rng.integers(0, nch, 25)] = rng.choice([-1.0, 32767 / 32768], 25)
        return x
    raise ValueError(kind)


def _scan_both(scanner, pcm, rate):
    nch = pcm.shape[1]
    dev = to_dev(pcm)
    out = []
    for prune in (1, 0):
        scanner.set_param("tp_prune", prune)
        (r,), _ = scanner.scan([dev], rate)
        sp, tp = scanner.channel_peaks(0, nch)
        out.append((r, sp.copy(), tp.copy()))
    # the unpruned scan walks every row as a whole (dense rows); once more unpruned but chunk by chunk
    # (sparse rows only): every output is computed by the same instructions whichever way it is reached
    scanner.set_param("tp_dense_min", 65)
    (r3,), _ = scanner.scan([dev], rate)
    sp3, tp3 = scanner.channel_peaks(0, nch)
    scanner.set_param("tp_dense_min", 32)
    scanner.set_param("tp_prune", 1)
    assert r3["peak"] == out[1][0]["peak"] and np.array_equal(tp3, out[1][2]) and np.array_equal(sp3, out[1][1])
    return out


CASES = [
    ("steps", 48000, 2, 31.7), ("steps", 44100, 2, 17.3), ("steps", 96000, 2, 9.1),
    ("steps", 48000, 1, 12.9), ("steps", 48000, 6, 9.7), ("steps", 44100, 5, 7.9),
    ("steps", 48000, 12, 4.3), ("steps", 11025, 2, 25.0), ("steps", 88200, 6, 5.1),
    ("adversarial", 48000, 2, 11.3), ("adversarial", 96000, 1, 6.7), ("square", 44100, 2, 8.9),
    ("hidden", 48000, 2, 23.9), ("hidden", 96000, 2, 12.1), ("hidden", 44100, 6, 9.3),
    ("music", 48000, 2, 29.9), ("music", 48000, 8, 6.1), ("music", 32000, 3, 11.7),
    ("impulses", 48000, 2, 14.1), ("impulses", 96000, 4, 6.3),
    # round 3: dense rows on every true-peak kernel instance -- window steps of 5 / 7 frames x the 4x / 2x interpolator,
    # slabs that divide the row (C = 75, 45) and that reach past it (C = 50, 70, 63, 25) -- and the channel-set forms
    ("limited", 48000, 2, 9.3), ("limited", 44100, 2, 8.1), ("limited", 22050, 2, 14.7), ("limited", 32000, 1, 9.9),
    ("limited", 96000, 2, 5.3), ("limited", 88200, 2, 5.9), ("limited", 176400, 2, 3.1), ("limited", 44100, 6, 5.7),
    ("limited", 48000, 5, 5.1), ("limited", 48000, 7, 4.9), ("limited", 11025, 2, 19.0), ("limited", 48000, 20, 2.9),
]


@pytest.mark.parametrize("kind,rate,nch,secs", CASES,
                         ids=["%s-%d-%dch" % (k, r, c) for k, r, c, _ in CASES])
def test_pruned_equals_unpruned(scanner, oracle, kind, rate, nch, secs):
    frames = int(rate * secs) + 37
    pcm = _material(kind, frames, nch, rate, seed=rate // 100 + nch)
    (a, spa, tpa), (b, spb, tpb) = _scan_both(scanner, pcm, rate)
    # bit-identical: every float compared with ==
    assert a["peak"] == b["peak"]
    assert a["true_peak"] == b["true_peak"]
    assert a["sample_peak"] == b["sample_peak"]
    assert np.array_equal(spa, spb) and np.array_equal(tpa, tpb)
    for k in ("loudness", "lra", "n_abs", "n_rel", "n_st", "sum_abs", "sum_rel"):
        assert a[k] == b[k] or (a[k] != a[k] and b[k] != b[k])
    # and both against the oracle (reference semantics: max(true peak, sample peak) per channel)
    ref = oracle.scan_track(pcm, rate)
    assert abs(a["peak"] - ref["peak"]) <= PEAK_TOL
    np.testing.assert_allclose(tpa, np.asarray(ref["true_peak"]), atol=PEAK_TOL, rtol=0)
    # regression bar far inside the contract: the fp32 interpolator against the reference's double
    # accumulation has measured <= 2e-7 (loudgain prints peaks with %.6f)
    np.testing.assert_allclose(tpa, np.asarray(ref["true_peak"]), atol=5e-6, rtol=0)


def test_hidden_peak_is_interpolated(scanner):
    """The 'hidden' material really has its true peak above every sample (otherwise the test
    above would not exercise the bound)."""
    pcm = _material("hidden", 48000 * 10, 2, 48000, seed=1)
    (r,), _ = scanner.scan([to_dev(pcm)], 48000)
    assert r["true_peak"] > r["sample_peak"] + 0.05


def test_hints_do_not_outlive_their_pcm(scanner, oracle):
    """The per-channel bound the pruning uses belongs to one scan: re-executing a plan after its
    buffer was overwritten with QUIETER audio must give the quiet audio's peaks (a stale loud
    bound would prune every chunk)."""
    import torch
    rate = 48000
    frames = rate * 20
    loud = synth.snap_s16_numpy((0.9 * np.sin(2 * np.pi * np.arange(frames) / 4.0 + np.pi / 4))[:, None]
                                .repeat(2, 1).astype(np.float32))
    quiet = _material("hidden", frames, 2, rate, seed=9) * np.float32(0.25)
    quiet = synth.snap_s16_numpy(quiet)
    buf = to_dev(loud)
    scanner.set_param("tp_prune", 1)
    scanner.plan([buf], rate, true_peak=True)
    scanner.execute()
    (r1,), _ = scanner.fetch()
    buf.copy_(torch.from_numpy(quiet))
    torch.cuda.synchronize()
    scanner.execute()
    (r2,), _ = scanner.fetch()
    ref = oracle.scan_track(quiet, rate)
    assert abs(r2["peak"] - ref["peak"]) <= PEAK_TOL
    assert r2["true_peak"] > r2["sample_peak"]          # the hidden inter-sample peak was found
    scanner.set_param("tp_prune", 0)
    (r3,), _ = scanner.scan([buf], rate)
    assert r2["peak"] == r3["peak"]
    assert r1["peak"] > 0.89


def test_album_of_tracks_shares_nothing_across_tracks(scanner, oracle):
    """The bound is per track and channel: a loud track must not prune a quiet one of the same plan."""
    rate = 48000
    loud = _material("adversarial", rate * 6, 2, rate, seed=2)
    quiet = synth.snap_s16_numpy(_material("hidden", rate * 9, 2, rate, seed=3) * np.float32(0.2))
    tracks = [to_dev(loud), to_dev(quiet), to_dev(loud[: rate * 3])]
    scanner.set_param("tp_prune", 1)
    got, _ = scanner.scan(tracks, rate)
    scanner.set_param("tp_prune", 0)
    want, _ = scanner.scan(tracks, rate)
    scanner.set_param("tp_prune", 1)
    for g, w, pcm in zip(got, want, (loud, quiet, loud[: rate * 3])):
        assert g["peak"] == w["peak"] and g["true_peak"] == w["true_peak"]
        assert abs(g["peak"] - oracle.scan_track(pcm, rate)["peak"]) <= PEAK_TOL
