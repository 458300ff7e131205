"""GPU: bench.py prints exactly one JSON line with the fields the driver reads
(metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better /
scaling / vs_baseline / dtype / data / config.workload, plus roofline and cpu_baseline),
for the default workload (c2 with the c3 object beside it) and for configs 4 and 5."""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    return json.loads(lines[0])


def _check_roofline(rf, algo_bytes):
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["kernel_ms_mean"] * 1e-3) / 1e9) / rf["achieved"] < 1e-2
    assert rf["algorithmic_bytes_per_launch"] == algo_bytes
    assert rf["kernel_ms_min"] <= rf["kernel_ms_mean"] * 1.001


def test_bench_line_contract():
    d = _bench("--steps", "30", "--warmup", "3", "--minutes", "2", "--cpu-seconds", "20", "--cpu-procs", "2",
               "--tracks", "40", "--track-scale", "0.05")
    assert d["metric"].startswith("Msamples/s scanned") and d["unit"] == "Msamples/s"
    assert d["n_gpus"] == 1 and d["steps"] == 30 and d["warmup"] == 3
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["workload"].startswith("C2")
    assert d["value"] > 0 and d["ms_per_step"] > 0
    # value = samples per step / time per step
    samples = d["config"]["frames_per_gpu"] * d["config"]["channels"]
    assert samples == d["config"]["samples_per_step_all_ranks"]
    assert abs(d["value"] - samples / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3
    _check_roofline(d["roofline"], samples * 4)
    # the reference's semantics (true peak on) beside it, and the material that defeats the pruning
    c3 = d["c3"]
    assert c3["workload"].startswith("C3") and c3["value"] > 0 and c3["unit"] == "Msamples/s"
    _check_roofline(c3["roofline"], samples * 4)
    assert c3["peak"] >= d["result"]["peak"]
    assert 0 < c3["adversarial"]["frac"] <= c3["roofline"]["frac"] * 1.05
    assert 0 < c3["limited"]["frac"] <= c3["roofline"]["frac"] * 1.05 and "f32 interpolator" in c3["dtype"]
    assert 0 < c3["limited"]["frac"] <= c3["noise"]["frac"] * 1.05   # nothing prunable vs nearly everything
    assert d["step_ms"]["min"] <= d["step_ms"]["median"]
    # the N = 1 anchor of the scaling series: config 4 (here 40 short tracks) on the same line
    # the same samples resident as interleaved int16 (LGD_PCM_S16): same results, reported beside the f32 line
    s16 = d["s16"]
    assert s16["identical_to_f32"] is True and s16["hbm_bytes_per_sample"] == 2
    for leg in ("no_true_peak", "true_peak"):
        assert 0 < s16[leg]["kernel_ms_min"] <= s16[leg]["kernel_ms_mean"]
        assert abs(s16[leg]["frac"] - 2 * s16[leg]["frac_of_own_bytes"]) < 2e-4
    assert s16["no_true_peak"]["loudness"] == d["result"]["loudness"] and s16["true_peak"]["peak"] == c3["peak"]
    assert d["config"]["pcm"] == "f32"
    c4 = d["c4"]
    assert c4["workload"].startswith("C4") and c4["scaling"] == "strong" and c4["n_gpus"] == 1 and c4["value"] > 0
    assert abs(c4["value"] - c4["samples_per_step"] / (c4["ms_per_step"] * 1e-3) / 1e6) / c4["value"] < 1e-3
    _check_roofline(c4["roofline"], c4["samples_per_step"] * 4)
    assert c4["album"]["n_abs"] > 0
    for k in ("f32", "s16"):
        assert d["h2d_inclusive"][k]["msamples_per_s"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["unit"] == "Msamples/s" and cb["value"] > 0 and cb["sample"]
    assert "c3" in cb["stands_next_to"] and cb["model"]
    mp = cb["multi_process"]
    assert mp["cores"] == 2 and mp["value"] > 0 and mp["model"] and mp["unit"] == "Msamples/s"


@pytest.mark.parametrize("pcm", ["f32", "s16"])
@pytest.mark.parametrize("wl,tracks", [("c4", 21), ("c5", 24)])
def test_album_workloads(wl, tracks, pcm):
    """configs 4 / 5 at reduced size (track lengths x 0.05): one GPU, album mode, true peak on; f32 and S16-resident PCM."""
    d = _bench("--workload", wl, "--tracks", str(tracks), "--track-scale", "0.05", "--steps", "5", "--warmup", "1",
               "--no-cpu-baseline", "--pcm", pcm)
    assert d["config"]["pcm"] == pcm
    assert d["config"]["workload"].startswith(wl.upper()) and d["scaling"] == "strong" and d["n_gpus"] == 1
    assert d["config"]["tracks_this_rank"] == tracks
    samples = d["config"]["samples_per_step_all_ranks"]
    assert abs(d["value"] - samples / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3
    _check_roofline(d["roofline"], samples * 4)
    al = d["result"]["album"]
    assert al["n_abs"] >= d["result"]["n_abs"] and al["peak"] >= d["result"]["peak"]


def test_one_rank_rccl_rehearsal():
    """The N > 1 code path (RCCL process group, album exchange every step) with one rank."""
    d = _bench("--workload", "c4", "--tracks", "6", "--track-scale", "0.05", "--steps", "5", "--warmup", "1",
               "--no-cpu-baseline", "--force-dist")
    c = d["collective"]
    assert c["backend"] == "nccl" and c["world_size"] == 1 and c["bytes_gathered_per_rank_per_step"] > 0
    assert c["ranks_folded"] == {"stage2_heads": 1, "with_content": 1, "stage3_records": 1}
    assert d["result"]["album"]["n_abs"] > 0


def test_two_ranks_rehearsal_on_one_gpu():
    """`python bench.py --gpus 2` end to end: the launcher spawns two ranks, default workload for N > 1 is
    config 4 (album tracks dealt round-robin), album exchange every step, max-over-ranks time, rank 0
    prints.  Both ranks share the one GPU of the box, so the exchange runs over gloo (RCCL needs one
    device per rank); the 8-GPU RCCL run is the driver's."""
    d = _bench("--gpus", "2", "--backend", "gloo", "--same-gpu", "--tracks", "11", "--track-scale", "0.05",
               "--steps", "4", "--warmup", "1")
    assert d["n_gpus"] == 2 and d["config"]["workload"].startswith("C4") and d["scaling"] == "strong"
    assert d["config"]["tracks_this_rank"] == 6                       # tracks 0, 2, 4, 6, 8, 10
    import bench
    assert d["config"]["samples_per_step_all_ranks"] == sum(bench.c4_track_frames(t, scale=0.05) * 2 for t in range(11))
    c = d["collective"]
    assert c["world_size"] == 2 and c["backend"] == "gloo" and c["bytes_gathered_per_rank_per_step"] > 0
    # counted by the album kernels on the device and read back with the album result: both ranks' records were folded
    assert c["ranks_folded"] == {"stage2_heads": 2, "with_content": 2, "stage3_records": 2}
    assert abs(d["value"] - d["config"]["samples_per_step_all_ranks"] / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3
    assert d["result"]["album"]["n_abs"] > d["result"]["n_abs"] > 0
