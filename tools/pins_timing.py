"""Timing of the pin label path on a larger volume (device passes vs host cover)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crackle_amd
from crackle_amd import synth

shape = tuple(int(v) for v in (sys.argv[1:4] or (512, 512, 256)))
arr = synth.as_numpy_f(synth.voronoi_labels(shape, np.uint32, seed=4, cell=(64, 64, 64)))
os.environ["CKL_PROFILE"] = "1"
for i in range(2):
  t0 = time.time()
  b = crackle_amd.compress(arr, allow_pins=1)
  t1 = time.time()
  print(f"pins compress {shape}: {t1 - t0:.3f} s, {len(b)} bytes, fmt={crackle_amd.header(b).label_format}", flush=True)
t0 = time.time()
f = crackle_amd.compress(arr, allow_pins=0)
print(f"flat compress: {time.time() - t0:.3f} s, {len(f)} bytes")
t0 = time.time()
back = crackle_amd.decompress(b)
print(f"pins decompress: {time.time() - t0:.3f} s", np.array_equal(back, arr))
