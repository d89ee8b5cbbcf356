"""Per-stage / per-phase timing of the C2 workload (device resident), for kernel work.
  python tools/stage_diag.py [SX SY SZ] [--markov N]
Env: CKL_DECODE_DIAG=1 adds the phase cycle counters of k_decode_cracks,
     CKL_WALK_DIAG=1 those of the encoder's walk kernel."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crackle_amd import synth
from crackle_amd import distributed as ckd

args = [a for a in sys.argv[1:] if not a.startswith("--")]
markov = int(sys.argv[sys.argv.index("--markov") + 1]) if "--markov" in sys.argv else 0
sx, sy, sz = (int(v) for v in (args[:3] or (1024, 1024, 512)))
dev = torch.device("cuda:0")
vol = synth.voronoi_labels((sx, sy, sz), np.uint32, seed=2, device=dev)
be = ckd.HipBackend(0)
codec = ckd.ShardedCodec(be, device=dev)
out = torch.empty_like(vol)
for it in range(3):
  torch.cuda.synchronize(); t0 = time.perf_counter()
  b = codec.compress(vol, (sx, sy, sz), markov_model_order=markov)
  torch.cuda.synchronize(); t1 = time.perf_counter()
  s = codec.open_decoder(b, (sx, sy, sz))
  torch.cuda.synchronize(); t2 = time.perf_counter()
  try:
    s.run(out)
  except RuntimeError as exc:
    if not os.environ.get("CKL_ABLATE_NOCHECK"):
      raise
  torch.cuda.synchronize(); t3 = time.perf_counter()
  print(f"iter {it}: encode {1e3*(t1-t0):.2f} ms  decode {1e3*(t3-t2):.2f} ms  (device pipeline {s.timing()[0]:.2f} ms)  bytes {len(b)}")
  print("   stages: " + "  ".join(f"{n}={ms:.3f}" for n, ms in s.stages()))
  s.close()
assert os.environ.get("CKL_ABLATE_NOCHECK") or torch.equal(out, vol)
