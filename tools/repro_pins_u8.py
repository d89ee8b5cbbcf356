"""KNOWN DEFECT (found by tools/random_parity.py 600 606, case 24, at the end of round 5; present since round 4 at least):
a pin-label stream of a 1024 x 248 x 4 uint8 volume of tiny Voronoi cells (4 x 4 x 2 voxels: ~65 k components, 250 label
values with ~130 pins each) decodes to WRONG labels in most voxels, differently from run to run, on every decoder path
(strip kernels, general run pipeline, k_decode_cracks) — the encoder's bytes are the reference's.  It is a window: 1024 x 244 ... 256 x 4 fail, 1024 x 248 x 3, x 6 and 1024 x 320 x 4 decode correctly, and so do uint16 /
uint32 labels of the same geometry and cells one slice deep (the table this script prints) — ~57 - 65 k components and pin work
items, 2-byte component ids.  Not yet located: the common stage is the pin label map (k_label_map_pins / k_label_map_ccids and
the pin tables of decoder_build)."""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import crackle_amd
from crackle_amd import synth
from oracle import oracle
chk = oracle.best()
def check(tag, arr, **kw):
  want = chk.compress(arr, allow_pins=True, **kw)
  h = crackle_amd.header(want)
  back = crackle_amd.decompress(want)
  print(tag, arr.shape, arr.dtype.name, "label_format", h.label_format, "bytes", len(want), "| ours:", "ok" if np.array_equal(back, arr) else f"WRONG ({int((back != arr).sum())} voxels)", flush=True)
for shape, dt, cell in [((1024, 248, 4), np.uint8, (4, 4, 2)), ((1024, 248, 4), np.uint16, (4, 4, 2)), ((1024, 248, 3), np.uint8, (4, 4, 2)), ((1024, 244, 4), np.uint8, (4, 4, 2)), ((1024, 252, 4), np.uint8, (4, 4, 2)), ((1024, 256, 4), np.uint8, (4, 4, 2)), ((1024, 320, 4), np.uint8, (4, 4, 2)), ((1024, 248, 6), np.uint8, (4, 4, 2))]:
  arr = synth.as_numpy_f(synth.voronoi_labels(shape, dt, seed=11, cell=cell))
  check(str(cell), arr)
