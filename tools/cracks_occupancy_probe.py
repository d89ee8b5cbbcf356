"""Does k_decode_cracks gain from two workgroups per CU?  Slices with few enough control symbols
fit LDS tables of half the size (CKL_LDS_CONTROLS), which lets two workgroups share a CU:
  python tools/cracks_occupancy_probe.py CELL [SX SY SZ]
prints the kernel's stage time; run once per setting of CKL_LDS_CONTROLS."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crackle_amd import synth
from crackle_amd import distributed as ckd

cell = int(sys.argv[1]) if len(sys.argv) > 1 else 56
sx, sy, sz = (int(v) for v in (sys.argv[2:5] or (1024, 1024, 512)))
dev = torch.device("cuda:0")
vol = synth.voronoi_labels((sx, sy, sz), np.uint32, seed=2, device=dev, cell=(cell, cell, 8))
codec = ckd.ShardedCodec(ckd.HipBackend(0), device=dev)
b = codec.compress(vol, (sx, sy, sz))
out = torch.empty_like(vol)
s = codec.open_decoder(b, (sx, sy, sz))
acc = {}
for it in range(6):
  s.run(out)
  for n, ms in s.stages():
    acc.setdefault(n, []).append(ms)
torch.cuda.synchronize()
assert torch.equal(out, vol)
print(f"cell {cell}, CKL_LDS_CONTROLS={os.environ.get('CKL_LDS_CONTROLS', 'default')}, bytes {len(b)}: " +
      "  ".join(f"{n}={np.mean(v[2:]):.3f}" for n, v in acc.items()) + f"  pipeline={s.timing()[0]:.3f}")
