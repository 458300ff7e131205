"""`python bench.py --gpus N` must start its N ranks itself (the driver runs exactly that command):
fresh child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, spawned before the parent
touches a GPU, rank 0's JSON line relayed, non-zero exit if a rank fails.  Rehearsed here on the
CPU with the gloo backend (world size 2 and 3); the workload definitions of configs 4 and 5
 are checked as data."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LGD_BENCH_CHILD"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=300, env=e)


def test_self_launch_world_2_and_3():
    for n in (2, 3):
        r = _run(["--gpus", str(n), "--launcher-selftest", "--backend", "gloo"])
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(lines) == 1, r.stdout            # exactly one line on stdout: rank 0's
        d = json.loads(lines[0])
        assert d["world_size"] == n and d["backend"] == "gloo"
        assert d["sum"] == n * (n + 1) / 2          # every rank took part in the all-reduce


def test_under_an_external_launcher_no_respawn():
    """With WORLD_SIZE already set (torch.distributed.run) bench.py must not spawn again."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    e0 = dict(os.environ, **env, RANK="0", LOCAL_RANK="0")
    e1 = dict(os.environ, **env, RANK="1", LOCAL_RANK="1")
    cmd = [sys.executable, BENCH, "--gpus", "2", "--launcher-selftest", "--backend", "gloo"]
    p1 = subprocess.Popen(cmd, env=e1, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    p0 = subprocess.run(cmd, env=e0, capture_output=True, text=True, timeout=300)
    out1, _ = p1.communicate(timeout=300)
    assert p0.returncode == 0 and p1.returncode == 0, p0.stderr[-1000:]
    d = json.loads([l for l in p0.stdout.splitlines() if l.startswith("{")][0])
    assert d["world_size"] == 2
    assert not [l for l in out1.splitlines() if l.startswith("{")]   # only rank 0 prints


def test_a_failing_rank_fails_the_launcher():
    r = _run(["--gpus", "2", "--launcher-selftest", "--backend", "nosuchbackend"])
    assert r.returncode != 0


def test_workload_definitions():
    sys.path.insert(0, ROOT)
    import bench
    # config 4: track t = 180 s + (t mod 7) * 30 s at 48 kHz
    assert bench.c4_track_frames(0) == 180 * 48000 and bench.c4_track_frames(6) == 360 * 48000
    assert bench.c4_track_frames(7) == 180 * 48000
    total = sum(bench.c4_track_frames(t) for t in range(1000))
    assert 1000 * 180 * 48000 < total < 1000 * 360 * 48000
    # config 5: 64 tracks cycle the 4 rates x 3 layouts, 120 s each
    specs = [bench.c5_track_spec(t) for t in range(64)]
    assert {(r, c) for r, c, _ in specs} == {(r, c) for r in (44100, 48000, 96000, 192000) for c in (1, 2, 6)}
    assert all(f == 120 * r for r, _, f in specs)
    # round-robin ownership covers every track exactly once
    for world in (1, 2, 3, 8):
        owned = sorted(t for r in range(world) for t in range(r, 1000, world))
        assert owned == list(range(1000))


def test_a_rank_that_dies_after_the_rendezvous_ends_the_others():
    """Rank 1 exits (code 3) once the process group is up; rank 0 then waits for it inside a collective.
    The launcher has to notice, end rank 0 and exit non-zero -- promptly, not at some outer limit."""
    import time
    t0 = time.time()
    r = _run(["--gpus", "2", "--launcher-selftest", "--backend", "gloo", "--selftest-die", "1"])
    assert r.returncode != 0
    assert "ranks failed" in r.stderr and "(1, 3)" in r.stderr, r.stderr[-1000:]
    assert time.time() - t0 < 120
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]   # no result line from a failed run


def test_the_launcher_gives_up_at_its_deadline():
    """Every rank alive but stuck (rank 1 never returns from its sleep): the launcher's own limit ends them."""
    import time
    t0 = time.time()
    r = _run(["--gpus", "2", "--launcher-selftest", "--backend", "gloo", "--selftest-die", "7", "--launch-timeout", "20"])
    assert r.returncode != 0 and "no result after" in r.stderr, r.stderr[-1000:]
    assert time.time() - t0 < 120
