"""Times ckl_decoder_label_stats on the bench workload (C2 by default) and prints the
stage timings of the run.  Usage: python tools/time_stats.py [sx sy sz]"""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import crackle_amd
from crackle_amd import _lib, synth
from crackle_amd import distributed as ckd


def main():
  shape = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (1024, 1024, 512)
  dev = torch.device("cuda", 0)
  vol = synth.voronoi_labels(shape, np.dtype(np.uint32), seed=2, device=dev)
  torch.cuda.synchronize()
  backend = ckd.HipBackend(0)
  binary = bytes(backend.encode(vol, shape))
  L = _lib.lib()
  h = C.c_void_p()
  assert L.ckl_decoder_create(binary, len(binary), 0, -1, 0, C.byref(h)) == 0
  cap = crackle_amd.num_labels(binary)
  lab = np.zeros(cap, np.uint64); cnt = np.zeros(cap, np.uint64)
  sums = np.zeros((cap, 3), np.uint64); box = np.zeros((cap, 6), np.uint32)
  n = C.c_uint64()
  for it in range(4):
    t0 = time.perf_counter()
    rc = L.ckl_decoder_label_stats(h, cap, lab.ctypes.data, cnt.ctypes.data, sums.ctypes.data, box.ctypes.data, C.byref(n))
    t1 = time.perf_counter()
    assert rc == 0, _lib.last_error()
    print(f"run {it}: {1e3 * (t1 - t0):.2f} ms wall, labels={n.value}, voxels={int(cnt.sum())}", flush=True)
  i = 0
  name, ms = C.c_char_p(), C.c_float()
  while L.ckl_decoder_stage_timing(h, i, C.byref(name), C.byref(ms)) == 0:
    print(f"  {name.value.decode():24s} {ms.value:.3f} ms")
    i += 1
  L.ckl_decoder_destroy(h)
  u, c = torch.unique(vol.view(torch.int32), return_counts=True)
  want = dict(zip(u.cpu().tolist(), c.cpu().tolist()))
  got = {int(a): int(b) for a, b in zip(lab[:n.value], cnt[:n.value]) if b}
  print("counts equal torch.unique:", want == got)


main()
