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
