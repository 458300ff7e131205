"""End-to-end library scan measurement: a synthetic WAV library in RAM-backed storage,
scanned by loudgain_amd.batch (files -> S16 -> PCIe -> HBM -> one launch per batch).
    python tools/library_bench.py [--albums 40 --tracks 12 --minutes 3.5 --dir /dev/shm/lglib]
Prints one JSON line; PCIe and file reading are inside the timed region (this is NOT the
bench.py metric, which starts with PCM resident in HBM)."""
import argparse, json, os, shutil, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--albums", type=int, default=40)
    ap.add_argument("--tracks", type=int, default=12)
    ap.add_argument("--minutes", type=float, default=3.5)
    ap.add_argument("--rate", type=int, default=44100)
    ap.add_argument("--dir", default="/dev/shm/lglib")
    ap.add_argument("--batch-samples", type=int, default=1 << 30)
    ap.add_argument("--threads", type=int, default=16, help="file reader threads (8: 1.75 s, 16: 1.44 s, 32: 1.69 s for the 17.8 GB library)")
    ap.add_argument("--cpu-albums", type=int, default=1, help="albums timed through the CPU oracle for comparison")
    args = ap.parse_args()
    from loudgain_amd import batch, synth
    from tests.test_gpu_scan_api import write_wav
    shutil.rmtree(args.dir, ignore_errors=True)
    frames = int(args.minutes * 60 * args.rate)
    t0 = time.perf_counter()
    protos = []
    for k in range(args.tracks):   # distinct tracks of one album, then copied with per-album gains folded in the name
        pcm = synth.snap_s16_numpy(synth.track_numpy(frames, 2, args.rate, seed=k + 1, step_s=3.0) * (0.3 + 0.7 * k / args.tracks))
        protos.append(write_wav(os.path.join(args.dir, "proto", "%02d.wav" % k), pcm, args.rate, "s16")
                      if os.makedirs(os.path.join(args.dir, "proto"), exist_ok=True) is None else None)
    for a in range(args.albums):
        d = os.path.join(args.dir, "lib", "artist%03d" % (a // 4), "album%03d" % a)
        os.makedirs(d)
        for k in range(args.tracks):
            shutil.copyfile(protos[(k + a) % args.tracks], os.path.join(d, "%02d.wav" % (k + 1)))
    gen_s = time.perf_counter() - t0
    lib = os.path.join(args.dir, "lib")
    nbytes = sum(os.path.getsize(os.path.join(r, f)) for r, _, fs in os.walk(lib) for f in fs)

    import torch
    torch.cuda.init()
    ls_kw = dict(batch_samples=args.batch_samples, reader_threads=args.threads)
    batch.scan_library([os.path.join(lib, "artist000")], **ls_kw)      # warm-up: library load, kernels, pinned pools
    t0 = time.perf_counter()
    res, st = batch.scan_library([lib], **ls_kw)
    dt = time.perf_counter() - t0
    line = dict(metric="library scan, files in RAM -> results", albums=len(res), tracks=st["tracks"],
                seconds=round(dt, 3), albums_per_s=round(len(res) / dt, 1), tracks_per_s=round(st["tracks"] / dt, 1),
                msamples_per_s=round(st["samples"] / dt / 1e6, 1), file_gb_per_s=round(nbytes / dt / 1e9, 2),
                x_realtime=round(st["samples"] / 2 / args.rate / dt, 0), read_s=round(st["read_s"], 3),
                gpu_wait_s=round(st["gpu_s"], 3), library_gb=round(nbytes / 1e9, 2), generate_s=round(gen_s, 1),
                example=dict(album=res[0]["album"], track0={k: res[0]["tracks"][0][k] for k in ("loudness", "lra", "peak", "gain")}))
    if args.cpu_albums:
        from oracle import lgoracle
        L = lgoracle.lib()
        t0 = time.perf_counter()
        n_s = 0
        for album in res[: args.cpu_albums]:
            files = [t["file"] for t in album["tracks"]]
            L.lgo_scan_init(len(files))
            for i, f in enumerate(files):
                L.lgo_scan_file(f.encode(), i)
            import ctypes as C
            for i in range(len(files)):
                o = L.lgo_scan_get_track_result(i, 0.0).contents
                L.lgo_scan_set_album_result(C.byref(o), 0.0)
            assert abs(o.album_loudness - album["album"]["loudness"]) < 1e-6
            L.lgo_scan_deinit()
            n_s += sum(t["frames"] * t["channels"] for t in album["tracks"])
        cdt = time.perf_counter() - t0
        line["cpu_oracle"] = dict(albums=args.cpu_albums, seconds=round(cdt, 2), msamples_per_s=round(n_s / cdt / 1e6, 1),
                                  note="CPU restatement of scan.c + libebur128, 1 thread, same files")
    print(json.dumps(line))
    shutil.rmtree(args.dir, ignore_errors=True)


if __name__ == "__main__":
    main()
