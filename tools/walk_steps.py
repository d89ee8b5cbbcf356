"""How serial is the crack trail?  The step kinds of k_trail_walk per slice (ckl_encoder_walk_step_kinds) for the
BASELINE volumes' slice sizes and for over-segmented / adversarial slices -> stdout (profiles/r05_walk_steps.txt).
  python tools/walk_steps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crackle_amd import synth, distributed as ckd

dev = torch.device("cuda:0")
cases = [
  ("C2 slices 1024x1024 u32, cell 32x32x8", (1024, 1024, 64), np.uint32, dict(cell=(32, 32, 8))),
  ("C4 slices 2048x2048 u32, cell 32x32x8", (2048, 2048, 16), np.uint32, dict(cell=(32, 32, 8))),
  ("watershed-like 1024x1024 u64, cell 8x8x4", (1024, 1024, 16), np.uint64, dict(cell=(8, 8, 4), offset=1 << 40)),
  ("binary noise 1024x1024", (1024, 1024, 8), np.uint32, None),
]
for name, shape, dt, kw in cases:
  if kw is None:
    vol = synth.random_labels_device(shape, dt, seed=2, high=2, device=dev)
  else:
    vol = synth.voronoi_labels(shape, dt, seed=2, device=dev, **kw)
  be = ckd.HipBackend(0)
  codec = ckd.ShardedCodec(be, device=dev)
  codec.compress(vol, shape)
  k = be.walk_step_kinds().astype(np.float64)
  tot = k[:, :4].sum(axis=1)
  print(f"{name}: {len(k)} slices, steps per slice mean {tot.mean():.0f} (min {tot.min():.0f}, max {tot.max():.0f})")
  print(f"   along the only remaining edge {100 * k[:, 0].sum() / tot.sum():.1f} %, branch + lowest edge {100 * k[:, 1].sum() / tot.sum():.1f} %, "
        f"dead ends {100 * k[:, 2].sum() / tot.sum():.1f} %, chain ends {100 * k[:, 3].sum() / tot.sum():.2f} %")
  print(f"   longest run of steps between two branch steps: mean {k[:, 4].mean():.1f}, max {k[:, 4].max():.0f}; mean run {(tot / np.maximum(k[:, 1], 1)).mean():.2f} steps per branch step")
  hist = np.bincount(np.minimum(k[:, 4].astype(np.int64), 63), minlength=64)
  print("   histogram of the longest decision-free run per slice (run length: slices): " + ", ".join(f"{i}: {c}" for i, c in enumerate(hist) if c))
