"""BASELINE.json configs 4 and 5 at SPEC SIZE on one GPU (the reference's album
walk they replace: /root/reference/src/scan.c:359-405): a 1000-track stereo album of 180..360 s
tracks (~104 GB of f32 PCM) and a 64-track album cycling 44.1/48/96/192 kHz x mono/stereo/5.1 at
120 s.  The oracle would need an hour for them, so:
  * a handful of sampled tracks are checked against the oracle one by one (same buffers);
  * the album numbers are recomputed with numpy from the 100 ms energies of ALL tracks (gating
    over the exact block list, order statistics for the range): counts exact, values to 1e-9;
  * invariants: the album's counts are the sums of the tracks', its peak their maximum; scanning
    the tracks in another order changes no count and not the range (an order statistic), and the
    loudness only by summation order; halving every sample (exact in binary floating point) halves
    the peaks exactly and shifts the ungated maxima by exactly 10 log10(1/4).
"""
import os
import sys

import numpy as np
import pytest

from tests.gpu_util import check_track

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ABS_GATE = 10 ** ((-70 + 0.691) / 10)


def _album_numpy(scanner, n_tracks, s100s):
    """Album loudness / range / counts from every track's 100 ms energies (channel-weighted sums as
    lgd_gate_pass1 forms them), the way libebur128's *_multiple functions walk all states."""
    from loudgain_amd.album import album_from_partials
    z_all, st_all = [], []
    for i in range(n_tracks):
        e = scanner.subblock_energies(i)
        s100 = s100s[i]
        if len(e) >= 4:
            z = (e[:-3] + e[1:-2] + e[2:-1] + e[3:]) / (4.0 * s100)
            z_all.append(z[z >= ABS_GATE])
        if len(e) >= 30:
            c = np.concatenate(([0.0], np.cumsum(e)))
            k = np.arange((len(e) - 30) // 10 + 1)
            # (cumsum differences round differently from the kernel's direct 30-term sums: take the
            # direct sums, exact order, for the values that are compared to 1e-9)
            st = np.array([e[10 * j:10 * j + 30].sum() for j in k]) / (30.0 * s100)
            st_all.append(st[st >= ABS_GATE])
            del c
    z_all = np.concatenate(z_all)
    st_all = np.concatenate(st_all) if st_all else np.zeros(0)
    thr = 0.1 * z_all.sum() / len(z_all)
    sel = z_all[z_all >= thr]
    return dict(n_abs=len(z_all), n_rel=len(sel), n_st=len(st_all), thr=thr,
                **album_from_partials(sel.sum(), len(sel), st_all, 0.0))


def _check_album(album, tracks, want):
    assert album["n_abs"] == want["n_abs"] == sum(t["n_abs"] for t in tracks)
    assert album["n_rel"] == want["n_rel"]
    assert album["n_st"] == want["n_st"] == sum(t["n_st"] for t in tracks)
    assert abs(album["loudness"] - want["loudness"]) <= 1e-9
    assert abs(album["lra"] - want["lra"]) <= 1e-9
    assert album["peak"] == max(t["peak"] for t in tracks)


def _run_config(build, sample_idx):
    import torch
    from loudgain_amd.device import DeviceScanner
    from oracle import lgoracle
    pcms, rates = build()
    n = len(pcms)
    sc = DeviceScanner(0)
    tracks, album = sc.scan(pcms, rates, true_peak=True, album=True)
    s100s = [(r + 5) // 10 for r in rates]
    # sampled tracks against the oracle
    for i in sample_idx:
        ref = lgoracle.scan_track(pcms[i].cpu().numpy(), rates[i])
        check_track(tracks[i], ref, rate=rates[i])
    want = _album_numpy(sc, n, s100s)
    _check_album(album, tracks, want)
    # another order of the same tracks: counts and the order statistics are order-free
    perm = list(np.random.default_rng(7).permutation(n))
    tr2, al2 = sc.scan([pcms[i] for i in perm], [rates[i] for i in perm], true_peak=True, album=True)
    for j, i in enumerate(perm):
        assert tr2[j] == tracks[i]
    assert (al2["n_abs"], al2["n_rel"], al2["n_st"]) == (album["n_abs"], album["n_rel"], album["n_st"])
    assert al2["lra"] == album["lra"] and al2["peak"] == album["peak"]
    assert abs(al2["loudness"] - album["loudness"]) <= 1e-11
    # exact scaling, in place (x 1/2 is exact in binary floating point)
    for p in pcms:
        p.mul_(0.5)
    tr3, al3 = sc.scan(pcms, rates, true_peak=True, album=True)
    for a, b in zip(tracks, tr3):
        assert b["sample_peak"] == a["sample_peak"] * 0.5 and abs(b["peak"] - a["peak"] * 0.5) <= 1e-7
        if np.isfinite(a["max_momentary"]):
            assert abs((a["max_momentary"] - b["max_momentary"]) - 10 * np.log10(4.0)) <= 1e-11
    assert al3["peak"] == max(t["peak"] for t in tr3)
    sc.close()
    del pcms
    torch.cuda.empty_cache()


def test_c4_album_of_1000_tracks_at_spec_size():
    sys.path.insert(0, ROOT)
    import bench
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 150e9:
        pytest.skip("needs ~110 GB of free HBM")
    args = bench.parse_args(["--workload", "c4"])

    def build():
        tr, rates = bench.build_tracks(args, "c4", 0, 1, torch.device("cuda", 0))
        assert len(tr) == 1000 and sum(t.numel() for t in tr) * 4 > 100e9
        return tr, rates
    _run_config(build, [0, 6, 501, 999])


def test_c5_mixed_album_at_spec_size():
    sys.path.insert(0, ROOT)
    import bench
    import torch
    args = bench.parse_args(["--workload", "c5"])

    def build():
        tr, rates = bench.build_tracks(args, "c5", 0, 1, torch.device("cuda", 0))
        assert len(tr) == 64 and {(r, t.shape[1]) for r, t in zip(rates, tr)} == \
            {(r, c) for r in (44100, 48000, 96000, 192000) for c in (1, 2, 6)}
        return tr, rates
    _run_config(build, list(range(12)))   # one track of every (rate, layout) pair
