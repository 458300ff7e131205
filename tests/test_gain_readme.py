"""CPU: the caller arithmetic (loudgain_amd/gain.py, mirroring src/loudgain.c:323-379
and the -O format :586-612) against the only known-answer data the reference
holds: the three README tables (README.md:632-681, docs/images/test-{1,2,3}.csv.png),
transcribed into tests/golden/readme_tables.json.  The tables print 2 / 6
decimals, so inputs are rounded: gains must agree to 0.011 dB, peaks to 2e-3 relative."""
import json
import os

import pytest

from loudgain_amd import gain as G

DOC = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "readme_tables.json")))


@pytest.mark.parametrize("tab", DOC["tables"], ids=[t["args"] for t in DOC["tables"]])
def test_readme_table(tab):
    pre = tab["pre_gain"]
    alb = DOC["album"]
    album_gain = -18.0 - alb["loudness"] + pre                   # scan.c:400
    n = len(DOC["files"])
    for i in range(n):
        L, tp = DOC["loudness"][i], DOC["true_peak"][i]
        track_gain = -18.0 - L + pre                              # scan.c:317
        assert -18.0 + pre == tab["reference"]                   # scan.c:327
        c = G.apply_clip_logic(track_gain, tp, album_gain, alb["peak"], do_album=True,
                               no_clip=tab["no_clip"])
        assert ("Y" if c["will_clip"] else "N") == tab["will_clip"][i], DOC["files"][i]
        assert ("Y" if c["tclip"] else "N") == tab["clip_prevent"][i], DOC["files"][i]
        assert abs(c["track_gain"] - tab["gain"][i]) <= 0.011, (DOC["files"][i], c["track_gain"])
        assert abs(c["tnew"] / tab["new_peak"][i] - 1.0) <= 2e-3, (DOC["files"][i], c["tnew"])
        if i == n - 1:
            a = tab["album"]
            assert abs(c["album_gain"] - a["gain"]) <= 0.011
            assert abs(c["anew"] / a["new_peak"] - 1.0) <= 2e-3
            assert ("Y" if c["aclip"] else "N") == a["clip_prevent"]
            assert ("Y" if (not c["aclip"] and c["again"] > c["apeak"]) else "N") == a["will_clip"]


def test_output_new_row_format():
    scan = dict(file="a.flac", track_loudness=-17.03, track_loudness_range=6.44, track_peak=1.000862,
                loudness_reference=-18.0, album_loudness=-18.12, album_loudness_range=9.57,
                album_peak=1.01761)
    c = G.apply_clip_logic(-0.97, 1.000862, 0.12, 1.01761, do_album=True, no_clip=True)
    rows = G.output_new_rows(scan, c, last=True, do_album=True)
    assert rows[0] == "a.flac\t-17.03 LUFS\t6.44 dB\t1.000862\t0.01 dBTP\t-18.00 LUFS\tN\tY\t-1.01 dB\t0.891251\t-1.00 dBTP"
    assert rows[1] == "Album\t-18.12 LUFS\t9.57 dB\t1.017610\t0.15 dBTP\t-18.00 LUFS\tN\tY\t-1.15 dB\t0.891251\t-1.00 dBTP"
    assert G.output_new_header().count("\t") == 10


def test_q78():
    assert G.gain_to_q78num(-1.01) == -259 and G.gain_to_q78num(0.5) == 128
    assert G.gain_to_q78num(0.001953125) == 1 and G.gain_to_q78num(-0.001953125) == -1  # round half away
