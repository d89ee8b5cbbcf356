import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crackle_amd import synth
from crackle_amd import distributed as ckd
dev = torch.device("cuda:0")
vol = synth.voronoi_labels((1024, 1024, 512), np.uint32, seed=2, device=dev)
be = ckd.HipBackend(0)
codec = ckd.ShardedCodec(be, device=dev)
b = codec.compress(vol, (1024, 1024, 512))
out = torch.empty_like(vol)
for ch in ("1", "2", "3", "4", "8"):
  os.environ["CKL_DECODE_CHUNKS"] = ch
  s = codec.open_decoder(b, (1024, 1024, 512))
  ts = []
  for it in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s.run(out)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
  print(f"chunks={ch}: wall {min(ts):.3f} ms  device pipeline {s.timing()[0]:.3f} ms", flush=True)
  s.close()
assert torch.equal(out, vol)
