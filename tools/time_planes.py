"""Wall time of the label-plane pass alone (ckl_encoder_stats: k_label_planes_fast + k_planes_reduce + one report back)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crackle_amd import synth, distributed as ckd
dev = torch.device("cuda:0")
shape = (1024, 1024, 512)
vol = synth.voronoi_labels(shape, np.uint32, seed=2, device=dev)
be = ckd.HipBackend(0)
ts = []
for i in range(30):
  torch.cuda.synchronize(); t0 = time.perf_counter()
  be.stats(vol, shape)
  torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("stats pass ms: min %.3f median %.3f" % (min(ts), sorted(ts)[len(ts)//2]))
# a plain read of the same bytes for comparison
x = vol.view(torch.int32)
for _ in range(3): x.sum()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): x.max()
torch.cuda.synchronize(); print("torch max over the volume ms: %.3f" % ((time.perf_counter() - t0) * 100))
