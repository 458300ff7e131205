"""Generates tests/golden/tracks.json: seeded synthetic inputs (loudgain_amd.synth,
regenerated from the recorded parameters, never stored) and the outputs of the
CPU oracle on them.

The reference (/root/reference) holds no fixture for the scan path and cannot be
built or run here (libebur128 + FFmpeg absent), so these vectors come from the
repo's own oracle (oracle/lg_oracle.c, pinned by tests/test_oracle_kat.py):
PARITY UNPINNED with respect to reference-held data.

Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from loudgain_amd import synth  # noqa: E402
from oracle import lgoracle  # noqa: E402

CASES = [
    # name, rate, channels, seconds, seed, step_s
    ("48k_stereo_64s", 48000, 2, 64.37, 1, 10.0),
    ("48k_mono_31s", 48000, 1, 31.2, 2, 4.0),
    ("44k1_stereo_40s", 44100, 2, 40.01, 3, 5.0),
    ("44k1_mono_12s", 44100, 1, 12.5, 4, 2.0),
    ("96k_stereo_21s", 96000, 2, 21.3, 5, 3.0),
    ("192k_stereo_9s", 192000, 2, 9.05, 6, 1.5),
    ("32k_stereo_15s", 32000, 2, 15.0, 7, 2.0),
    ("48k_stereo_short_0s35", 48000, 2, 0.35, 8, 10.0),
    ("48k_stereo_0s4", 48000, 2, 0.4, 9, 10.0),
    ("48k_stereo_3s", 48000, 2, 3.0, 10, 10.0),
]


def main():
    out = []
    for name, rate, ch, secs, seed, step in CASES:
        frames = int(round(secs * rate))
        pcm = synth.track_numpy(frames, ch, rate, seed=seed, step_s=step)
        r = lgoracle.scan_track(pcm, rate)
        st = r.pop("state")
        out.append(dict(name=name, rate=rate, channels=ch, frames=frames, seed=seed, step_s=step,
                        loudness=r["loudness"] if r["n_rel"] else "-inf", lra=r["lra"],
                        peak=r["peak"], true_peak=r["true_peak"], sample_peak=r["sample_peak"],
                        n_abs=r["n_abs"], n_rel=r["n_rel"], n_st=r["n_st"],
                        sum_abs=r["sum_abs"], sum_rel=r["sum_rel"],
                        rel_threshold=r["rel_threshold"],
                        first_blocks=[float(x) for x in st.gating_blocks()[:4]]))
    # one album over the first three stereo 48k/44k1 cases is rate-mixed on purpose
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tracks.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, len(out), "cases")


if __name__ == "__main__":
    main()
