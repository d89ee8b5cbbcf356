"""Times crackle_amd.reencode on the bench workload: python tools/time_reencode.py [sx sy sz]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import crackle_amd
from crackle_amd import synth
from crackle_amd import distributed as ckd


def main():
  shape = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (1024, 1024, 512)
  vol = synth.voronoi_labels(shape, np.dtype(np.uint32), seed=2, device=torch.device("cuda", 0))
  torch.cuda.synchronize()
  backend = ckd.HipBackend(0)
  b0 = bytes(backend.encode(vol, shape))
  b5 = bytes(backend.encode(vol, shape, markov_model_order=5))
  for src, order, want, tag in ((b0, 5, b5, "0 -> 5"), (b5, 0, b0, "5 -> 0")):
    for it in range(3):
      t0 = time.perf_counter()
      got = crackle_amd.reencode(src, order)
      t1 = time.perf_counter()
      print(f"reencode {tag}: {1e3 * (t1 - t0):.2f} ms  ({len(src)} -> {len(got)} bytes) equal to a fresh encode: {got == want}", flush=True)


main()
