import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from crackle_amd import synth, distributed as ckd
dev = torch.device("cuda:0")
vol = synth.voronoi_labels((1024,1024,512), np.uint32, seed=2, device=dev)
be = ckd.HipBackend(0, zero_copy=True)
be.keep_device_stream((1024,1024,512), 4, True)
codec = ckd.ShardedCodec(be, device=dev)
out = torch.empty_like(vol)
for it in range(6):
    b = codec.compress(vol, (1024,1024,512))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s = be.open_decoder(be.device_stream(), 0, 512)
    t1 = time.perf_counter()
    s.run(out)
    t2 = time.perf_counter()
    print(f"create {1e3*(t1-t0):.3f} run {1e3*(t2-t1):.3f} device {s.timing()[0]:.3f}", file=sys.stderr)
    s.close()
