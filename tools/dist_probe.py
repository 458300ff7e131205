"""Host-enqueue vs GPU time of the pipelined album path (RCCL group of one rank)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from loudgain_amd import synth
from loudgain_amd.device import DeviceScanner
from loudgain_amd.album import DistributedAlbumScanner
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
rate, ch = 48000, 2
pcm = synth.track_torch(60 * 60 * rate, ch, rate, seed=0, device=dev)
sc = DeviceScanner(0)
job = DistributedAlbumScanner(sc, [pcm], rate, true_peak=False, always_exchange=True)
s = torch.cuda.Stream()
for _ in range(5): job.execute(s)
job.fetch()
for n in (20, 100):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): job.execute(s)
    t1 = time.perf_counter(); job.fetch(); t2 = time.perf_counter()
    print("n=%d host enqueue %.3f ms/step, total %.3f ms/step" % (n, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
# pieces
t0 = time.perf_counter()
for _ in range(100): job.shard.run_tracks(s)
t1 = time.perf_counter(); job.fetch(); print("run_tracks only: host %.3f ms/step" % ((t1 - t0) / 100 * 1e3))
dist.destroy_process_group()
