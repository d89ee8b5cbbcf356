"""Timing of point_cloud on a BASELINE-shaped volume: device path vs the CPU checker."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crackle_amd
from crackle_amd import operations, synth

shape = tuple(int(v) for v in (sys.argv[1:4] or (1024, 1024, 64)))
import torch
arr = synth.as_numpy_f(synth.voronoi_labels(shape, np.uint32, seed=4, cell=(32, 32, 8), device=torch.device("cuda", 0)))      # generated on the device: a host core needs minutes for the C2 volume
binary = crackle_amd.compress(arr)
os.environ["CKL_PROFILE"] = "1"
for i in range(3):
  t0 = time.time()
  ptc = operations._point_cloud_raw(binary, 0, -1, None, False, 0)
  t1 = time.time()
  print(f"point_cloud {shape}: {t1 - t0:.3f} s, {len(ptc)} labels, {sum(v.size for v in ptc.values()) // 3} points", flush=True)
if "--cpu" in sys.argv:
  from oracle import oracle
  ck = oracle.best()
  t0 = time.time()
  want = ck.point_cloud(binary, 0, -1, None, False)
  print(f"cpu checker ({type(ck).__name__}, 1 thread): {time.time() - t0:.3f} s; equal: {sorted(want) == sorted(ptc) and all(np.array_equal(want[k], ptc[k]) for k in want)}")
