"""CPU: the oracle reproduces the committed golden vectors (guards both the
oracle and the synthetic-input generator against drift)."""
import json
import math
import os

import pytest

from loudgain_amd import synth

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tracks.json")))


@pytest.mark.parametrize("g", GOLD, ids=[g["name"] for g in GOLD])
def test_oracle_matches_golden(oracle, g):
    pcm = synth.track_numpy(g["frames"], g["channels"], g["rate"], seed=g["seed"], step_s=g["step_s"])
    r = oracle.scan_track(pcm, g["rate"])
    exp_l = -math.inf if g["loudness"] == "-inf" else g["loudness"]
    assert r["loudness"] == pytest.approx(exp_l, abs=1e-10)
    assert r["lra"] == pytest.approx(g["lra"], abs=1e-10)
    assert r["peak"] == g["peak"]
    assert (r["n_abs"], r["n_rel"], r["n_st"]) == (g["n_abs"], g["n_rel"], g["n_st"])


# ---- the fixtures keep their distance from the gates --------------------------------------
# The HIP path sums in another order than the reference, so block energies differ in their last
# bits (<= 1e-9 relative, tests/gpu_util.py), yet the integer block counts must agree EXACTLY.
# That can only be demanded of inputs none of whose blocks sits within that distance of a gate:
# the absolute gate (-70 LUFS) for 400 ms and 3 s blocks, the relative gate (-10 LU) for the
# listed 400 ms blocks, the -20 LU cut of the loudness range for the listed 3 s blocks.
# All block energies (listed or not) come from an independent evaluation: scipy's lfilter with
# the oracle's own coefficients + brute-force block sums, as tests/test_oracle_kat.py does.
ABS_GATE = 10 ** ((-70 + 0.691) / 10)
MARGIN = 1e-9


def _all_block_energies(oracle, pcm, rate):
    import numpy as np
    from scipy import signal
    b, a = oracle.design_filter(rate)
    y = signal.lfilter(b, a, pcm.astype(np.float64), axis=0)
    s100 = (rate + 5) // 10
    nch = pcm.shape[1]
    w = np.array([[1.0], [1.0, 1.0], [1.0, 1.0, 1.0], [1.0, 1.0, 1.41, 1.41], [1.0, 1.0, 1.0, 1.41, 1.41]][nch - 1]
                 if nch <= 5 else [1.0, 1.0, 1.0, 0.0, 1.41, 1.41] + [0.0] * (nch - 6))
    nsb = pcm.shape[0] // s100
    e = ((y[: nsb * s100] ** 2).reshape(nsb, s100, nch).sum(axis=1) * w[None, :]).sum(axis=1)
    z = np.array([e[j:j + 4].sum() for j in range(max(0, nsb - 3))]) / (4.0 * s100)
    st = np.array([e[10 * k:10 * k + 30].sum() for k in range((nsb - 30) // 10 + 1 if nsb >= 30 else 0)]) / (30.0 * s100)
    return z, st


def _rel_gap(values, threshold):
    import numpy as np
    if len(values) == 0 or threshold <= 0:
        return np.inf
    return float(np.min(np.abs(np.asarray(values) / threshold - 1.0)))


@pytest.mark.parametrize("g", GOLD, ids=[g["name"] for g in GOLD])
def test_no_block_near_a_gate(oracle, g):
    import numpy as np
    pcm = synth.track_numpy(g["frames"], g["channels"], g["rate"], seed=g["seed"], step_s=g["step_s"])
    z, st = _all_block_energies(oracle, pcm, g["rate"])
    assert _rel_gap(z, ABS_GATE) > MARGIN and _rel_gap(st, ABS_GATE) > MARGIN
    zl, stl = z[z >= ABS_GATE], st[st >= ABS_GATE]
    assert (len(zl), len(stl)) == (g["n_abs"], g["n_st"])     # the independent evaluation lists the same blocks
    if len(zl):
        gamma_r = 0.1 * zl.sum() / len(zl)
        assert _rel_gap(zl, gamma_r) > MARGIN
        assert int((zl >= gamma_r).sum()) == g["n_rel"]
    if len(stl):
        assert _rel_gap(stl, 0.01 * stl.sum() / len(stl)) > MARGIN
