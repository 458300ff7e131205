"""CPU: the adjacent-pair bound the true-peak pruning rests on (loudgain_amd/csrc/lgd_engine.cpp interp_pair_bound,
DESIGN.md 3.2), checked as mathematics against the oracle's interpolator taps (oracle/lg_oracle.c:156-179 restates
libebur128's interp_process): for every window, every non-trivial phase,
    |y| <= min(a, b) * S2 + (|a - b| + R) * M,
a, b = the |taps| at the two centre positions, R = the sum of the other |taps|, M = the window's largest |x|,
S2 = its largest |x[j]| + |x[j-1]| under the centre taps.  Also that it is never looser than L1 * M and what it gives
for an isolated peak."""
import numpy as np
import pytest


def _phases(oracle, rate):
    f, delay, ph = oracle.design_interp(rate)
    out = []
    for p in range(1, f):
        idx, c = ph[p]
        taps = np.zeros(delay)
        taps[idx] = c            # tap k multiplies x[n - k]
        out.append(taps[np.nonzero(taps)[0].min():np.nonzero(taps)[0].max() + 1])
    return f, out


def _pair_coeffs(taps):
    n = len(taps)
    ka, kb = n // 2 - 1, n // 2
    a, b = abs(taps[ka]), abs(taps[kb])
    rest = np.abs(taps).sum() - a - b
    return min(a, b), abs(a - b) + rest, ka, kb


@pytest.mark.parametrize("rate", [48000, 44100, 96000])
def test_pair_bound_holds_and_is_tighter_than_l1(oracle, rate):
    f, phases = _phases(oracle, rate)
    assert f in (2, 4) and all(len(t) == (12 if f == 4 else 24) for t in phases)
    rng = np.random.default_rng(rate)
    n = len(phases[0])
    mats = [rng.standard_normal((4000, n)),                                   # noise
            np.where(rng.random((4000, n)) < 0.08, rng.standard_normal((4000, n)) * 4, rng.standard_normal((4000, n)) * 0.2),
            np.clip(rng.standard_normal((4000, n)) * 2.0, -1, 1),             # limited: flat tops
            np.sin(np.arange(n)[None, :] * rng.uniform(0.05, 3.1, (4000, 1)) + rng.uniform(0, 6.3, (4000, 1)))]
    for taps in phases:
        al, be, ka, kb = _pair_coeffs(taps)
        l1 = np.abs(taps).sum()
        assert al * 2 + be == pytest.approx(l1, rel=1e-12)                    # flat top (S2 = 2 M): the L1 bound again
        for x in mats:
            y = np.abs(x[:, ::-1] @ taps)                                     # window w[i] = x[n - (n_taps - 1) + i]
            xa = np.abs(x[:, ::-1])                                           # xa[:, k] = |x[n - k]|
            m = xa.max(axis=1)
            s2 = xa[:, ka] + xa[:, kb]
            bound = al * s2 + be * m
            assert np.all(y <= bound * (1 + 1e-12))
            assert np.all(bound <= l1 * m * (1 + 1e-12))
    if f == 4:   # an isolated peak with neighbours at 0.3 of it: ~1.4 M where L1 M is 1.86 M
        worst = max(al * 1.3 + be for al, be, _, _ in map(_pair_coeffs, phases))
        assert 1.35 < worst < 1.5 and max(np.abs(t).sum() for t in phases) == pytest.approx(1.8642, abs=2e-3)
