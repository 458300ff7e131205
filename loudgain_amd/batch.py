"""Library scan: folders -> albums -> batched GPU scans.

The reference's bin/rgbpm2 (/root/reference/bin/rgbpm2:120-175) walks folder trees,
calls files of one type in one folder an album, and runs one `loudgain -a` process
per album on a pool of CPU cores.  Here the same clustering feeds a GPU queue:
albums are packed into batches that fit a pinned staging buffer, a reader thread
fills batch k+1 (files -> interleaved S16, the grid scan.c:442 puts everything on)
while batch k crosses PCIe as S16, is widened on the device and scanned -- every
track and every album of the batch in ONE launch (lgd_plan_albums).  Across GPUs
whole albums are dealt out by size (no exchange: albums are independent); only a
single album larger than a GPU would need loudgain_amd.album.

Without FFmpeg only RIFF/WAVE is readable (scan_wav_probe / scan_wav_read_s16);
the clustering itself knows rgbpm2's extension table.
"""
import fnmatch
import os
import queue
import threading

import numpy as np

from . import _lib
from . import gain as _gain

# rgbpm2's table (bin/rgbpm2:57-75): extension -> loudgain options; READABLE = what
# this build can decode itself
EXTENSIONS = {
    ".flac": "-a -k -s e", ".ogg": "-a -k -s e", ".oga": "-a -k -s e", ".spx": "-a -k -s e",
    ".opus": "-a -k -s e", ".mp2": "-I 3 -S -L -a -k -s e", ".mp3": "-I 3 -S -L -a -k -s e",
    ".m4a": "-L -a -k -s e", ".wma": "-L -a -k -s e", ".asf": "-L -a -k -s e",
    ".wav": "-I 3 -L -a -k -s e", ".aif": "-I 3 -L -a -k -s e", ".aiff": "-I 3 -L -a -k -s e",
    ".wv": "-S -a -k -s e", ".ape": "-S -a -k -s e",
}
READABLE = (".wav",)
EXCLUDES = ("*[[]compilations[]]",)  # bin/rgbpm2:79


def cluster_folders(folders, extensions=READABLE, follow_links=False, excludes=EXCLUDES):
    """{folder: {ext: [file names]}} with rgbpm2's rules (bin/rgbpm2:120-141): recursive
    walk, excluded directory patterns pruned, extension compared in lower case.
    Returns (cluster, number of excluded folders)."""
    cluster, excluded = {}, 0
    for folder in folders:
        for root, dirs, files in os.walk(os.path.abspath(folder), followlinks=follow_links, topdown=True):
            keep = []
            for d in dirs:
                if any(fnmatch.fnmatch(d, pat) for pat in excludes):
                    excluded += 1
                else:
                    keep.append(d)
            dirs[:] = sorted(keep)
            for f in sorted(files):
                ext = os.path.splitext(f)[1].lower()
                if ext in extensions:
                    cluster.setdefault(root, {}).setdefault(ext, []).append(f)
    return cluster, excluded


def album_tasks(cluster):
    """One task = one album = the files of one type in one folder (bin/rgbpm2:150-157)."""
    return [dict(folder=folder, ext=ext, files=[os.path.join(folder, f) for f in cluster[folder][ext]])
            for folder in sorted(cluster) for ext in sorted(cluster[folder])]


def deal_albums(sizes, world):
    """Whole albums to ranks, largest first onto the least loaded rank (LPT): returns, per
    rank, the album indices in their original order.  Deterministic, no communication."""
    load = [0] * world
    owner = [0] * len(sizes)
    for i in sorted(range(len(sizes)), key=lambda i: (-sizes[i], i)):
        r = min(range(world), key=lambda r: (load[r], r))
        owner[i] = r
        load[r] += sizes[i]
    return [[i for i in range(len(sizes)) if owner[i] == r] for r in range(world)]


def pack_batches(sizes, capacity):
    """Consecutive albums into batches of at most `capacity` samples (an album larger than
    that gets a batch of its own and is handled by growing the staging buffer)."""
    batches, cur, used = [], [], 0
    for i, s in enumerate(sizes):
        if cur and used + s > capacity:
            batches.append(cur)
            cur, used = [], 0
        cur.append(i)
        used += s
    if cur:
        batches.append(cur)
    return batches


class _Staging:
    """One pinned host buffer + its device twin (S16: scanned as it is)."""

    def __init__(self, device):
        self.device = device
        self.host = self.dev16 = None
        self.cap = 0

    def ensure(self, n):
        import torch
        if n > self.cap:
            cap = max(n, 1)
            self.host = torch.empty(cap, dtype=torch.int16).pin_memory()
            self.dev16 = torch.empty(cap, dtype=torch.int16, device=self.device)
            self.cap = cap


class LibraryScanner:
    """Scans album tasks (album_tasks) on one GPU; results as loudgain reports them.

    for album in LibraryScanner(0).scan(tasks): album["tracks"][i] has file, loudness,
    lra, peak (true peak) and gain/clip fields of loudgain's -k -s e run;
    album["album"] the album values."""

    def __init__(self, device=0, batch_samples=1 << 29, true_peak=True, pre_gain=0.0,
                 max_true_peak_level=-1.0, clip_prevention=True, reader_threads=4):
        import torch
        from .device import DeviceScanner
        self.torch = torch
        self.device = torch.device("cuda", device)
        self.sc = DeviceScanner(device)
        self.batch_samples = int(batch_samples)
        self.true_peak = true_peak
        self.pre_gain = pre_gain
        self.max_tpl = max_true_peak_level
        self.clip = clip_prevention
        self.reader_threads = reader_threads
        self.stage = [_Staging(self.device), _Staging(self.device)]
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.scan_stream = torch.cuda.Stream(device=self.device)
        self.stats = dict(albums=0, tracks=0, samples=0, read_s=0.0, gpu_s=0.0)

    # -- host side: probe + read one batch into pinned memory -------------------------
    def _probe(self, tasks):
        from . import scan as S
        infos = []
        for t in tasks:
            infos.append([S.scan_wav_probe(f) for f in t["files"]])
        return infos

    def _read_batch(self, tasks, infos, idxs, st):
        """Fills st.host; returns the track table [(album_in_batch, file, info, offset, frames)]."""
        from concurrent.futures import ThreadPoolExecutor
        from . import scan as S
        table, off = [], 0
        for b, i in enumerate(idxs):
            for f, wi in zip(tasks[i]["files"], infos[i]):
                n = wi["frames"] * wi["channels"]
                off = (off + 7) & ~7          # 16-byte aligned S16 tracks
                table.append([b, f, wi, off, wi["frames"]])
                off += n
        st.ensure(off)
        base = st.host.data_ptr()

        def rd(row):
            row[4] = S.scan_wav_read_s16(row[1], base + 2 * row[3], row[2]["frames"])  # may be short
        with ThreadPoolExecutor(max(1, self.reader_threads)) as ex:   # ctypes releases the GIL
            list(ex.map(rd, table))
        return table, off

    # -- device side -------------------------------------------------------------------
    def _launch(self, st, table, n_samples, n_albums):
        torch = self.torch
        with torch.cuda.stream(self.copy_stream):
            st.dev16[:n_samples].copy_(st.host[:n_samples], non_blocking=True)
        self.scan_stream.wait_stream(self.copy_stream)
        # the tracks are scanned as the S16 they are (LGD_PCM_S16: what the reference feeds libebur128, scan.c:442-448)
        base16 = st.dev16.data_ptr()
        tracks = [(base16 + 2 * off, frames, wi["channels"], _lib.PCM_S16) for _, _, wi, off, frames in table]
        rates = [wi["rate"] for _, _, wi, _, _ in table]
        albums = [b for b, *_ in table]
        self.sc.plan(tracks, rates, true_peak=self.true_peak, albums=albums)
        self.sc.n_albums = n_albums        # trailing albums without readable tracks still get a record
        self.sc.execute(self.scan_stream)

    def _collect(self, tasks, idxs, table):
        tr, al = self.sc.fetch()
        out = []
        for b, i in enumerate(idxs):
            rows = [(row, r) for row, r in zip(table, tr) if row[0] == b]
            a = al[b]
            album = dict(folder=tasks[i]["folder"], ext=tasks[i]["ext"], tracks=[],
                         album=dict(loudness=a["loudness"], lra=a["lra"], peak=a["peak"]))
            have_album = bool(np.isfinite(a["loudness"]))
            a_gain = -18.0 - a["loudness"] + self.pre_gain if have_album else 0.0   # scan.c:393
            for row, r in rows:
                t = dict(file=row[1], frames=row[4], channels=row[2]["channels"], rate=row[2]["rate"],
                         codec_id=row[2]["codec_id"], loudness=r["loudness"], lra=r["lra"], peak=r["peak"],
                         max_momentary=r["max_momentary"], max_shortterm=r["max_shortterm"])
                if np.isfinite(r["loudness"]):
                    # loudgain.c:323-379, the -k (clip prevention) / -K logic of an album run
                    g = _gain.apply_clip_logic(-18.0 - r["loudness"] + self.pre_gain, r["peak"], a_gain,
                                               a["peak"], do_album=have_album, no_clip=self.clip,
                                               max_true_peak_level=self.max_tpl)
                    t.update(gain=g["track_gain"], will_clip=g["will_clip"], clip_prevented=g["tclip"])
                    if have_album and "gain" not in album["album"]:
                        album["album"].update(gain=g["album_gain"], clip_prevented=g["aclip"])
                album["tracks"].append(t)
            out.append(album)
        return out

    def scan(self, tasks):
        """Generator of album results, in task order.  Reading batch k+1 overlaps the
        upload + scan of batch k (two staging buffers, one reader thread)."""
        import time
        infos = self._probe(tasks)
        sizes = [sum(wi["frames"] * wi["channels"] + 8 for wi in inf) for inf in infos]
        batches = pack_batches(sizes, self.batch_samples)
        q = queue.Queue(maxsize=1)
        free = queue.Queue()
        for st in self.stage:
            free.put(st)

        def reader():
            try:
                for idxs in batches:
                    st = free.get()
                    t0 = time.perf_counter()
                    table, n = self._read_batch(tasks, infos, idxs, st)
                    self.stats["read_s"] += time.perf_counter() - t0
                    q.put((idxs, st, table, n))
                q.put(None)
            except BaseException as e:  # surface reader errors in the consumer
                q.put(e)

        th = threading.Thread(target=reader, daemon=True)
        th.start()
        pending = None
        while True:
            item = q.get()
            if isinstance(item, BaseException):
                raise item
            if item is not None:
                idxs, st, table, n = item
                t0 = time.perf_counter()
                self._launch(st, table, n, len(idxs))      # async: returns once enqueued
            if pending is not None:                         # results of the previous batch
                p_idxs, p_st, p_table, p_t0, p_res = pending
                for album in p_res:
                    yield album
            if item is None:
                break
            res = self._collect(tasks, idxs, table)         # syncs this batch
            self.stats["gpu_s"] += time.perf_counter() - t0
            self.stats["albums"] += len(idxs)
            self.stats["tracks"] += len(table)
            self.stats["samples"] += n
            free.put(st)                                     # reader may refill it now
            pending = (idxs, st, table, t0, res)
        th.join()


def scan_library(folders, device=0, rank=0, world=1, **kw):
    """rgbpm2's job for this rank: cluster, deal whole albums to ranks, scan.  Returns
    (list of album results of this rank, statistics)."""
    cluster, excluded = cluster_folders(folders)
    tasks = album_tasks(cluster)
    if world > 1:
        sizes = [sum(os.path.getsize(f) for f in t["files"]) for t in tasks]
        tasks = [tasks[i] for i in deal_albums(sizes, world)[rank]]
    ls = LibraryScanner(device, **kw)
    res = list(ls.scan(tasks))
    st = dict(ls.stats, excluded_folders=excluded, folders=len(cluster), tasks=len(tasks))
    return res, st
