"""The drop-in boundary from C: tests/c/link_scan.c (a caller written like loudgain.c's
main and like scan.c's libebur128 usage) is built with plain gcc against
libloudscan_hip.so.  CPU: it compiles and links.  GPU: it runs, and both boundaries
agree with the oracle on the same synthetic input."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "link_scan.c")


def _build(out):
    from loudgain_amd import _lib
    _lib.load()  # library present
    libdir = os.path.join(ROOT, "loudgain_amd", "csrc")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["gcc", "-O1", "-Wall", "-Werror", "-std=gnu99", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "include", "compat"), SRC, "-o", out, "-L" + libdir, "-lloudscan_hip",
           "-Wl,-rpath," + libdir, "-L" + os.path.join(rocm, "lib"), "-Wl,-rpath," + os.path.join(rocm, "lib"),
           "-lamdhip64", "-lm"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return out


def test_c_caller_compiles_and_links(tmp_path):
    exe = _build(str(tmp_path / "link_scan"))
    assert os.path.exists(exe)
    syms = subprocess.run(["nm", "-u", exe], capture_output=True, text=True).stdout
    for s in ("scan_init", "scan_get_track_result", "scan_set_album_result", "ebur128_init",
              "ebur128_add_frames_short", "ebur128_loudness_global_multiple", "ebur128_true_peak"):
        assert s in syms


def _lcg_pcm(frames, seed):
    s = np.uint64((seed * 2654435761 + 1) & 0xFFFFFFFF)
    n = frames * 2
    # s_{k+1} = a s_k + c (mod 2^32), vectorised through the closed form of the LCG powers
    a, c = 1664525, 1013904223
    out = np.empty(n, np.uint32)
    cur = int(s)
    for i in range(n):
        cur = (cur * a + c) & 0xFFFFFFFF
        out[i] = cur
    v = ((out >> 16).astype(np.int64) - 32768) / 32768.0
    amp = np.where(np.arange(n) < frames, 8000.0, 1000.0)
    return np.rint(v * amp).astype(np.int16).reshape(frames, 2)   # lrint: round half to even, like np.rint


@pytest.mark.gpu
def test_c_caller_matches_oracle(oracle, tmp_path):
    exe = _build(str(tmp_path / "link_scan"))
    frames, seed = 48000 * 6 + 123, 7
    r = subprocess.run([exe, str(frames), str(seed)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout.strip().splitlines()[-1])
    pcm = _lcg_pcm(frames, seed)
    a = oracle.scan_track(pcm.astype(np.float32) / 32768.0, 48000)
    b = oracle.scan_track(pcm[: frames // 2].astype(np.float32) / 32768.0, 48000)
    states = [a["state"], b["state"]]
    want_al, want_ar = oracle.album_loudness(states), oracle.album_lra(states)
    assert got["version"] == [1, 2, 4]
    for side in ("scan", "ebur128"):
        g = got[side]
        assert abs(g["l0"] - a["loudness"]) <= 1e-6 and abs(g["l1"] - b["loudness"]) <= 1e-6
        assert abs(g["lra0"] - a["lra"]) <= 1e-6 and abs(g["peak0"] - a["peak"]) <= 1e-4
        assert abs(g["album_l"] - want_al) <= 1e-6 and abs(g["album_lra"] - want_ar) <= 1e-6
    assert abs(got["scan"]["gain0"] - (-18.0 - a["loudness"])) <= 1e-6 and got["scan"]["ref"] == -18.0
