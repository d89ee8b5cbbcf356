"""An UPSTREAM defect that this repository reproduces faithfully (found by tools/random_parity.py 600 606, case 24).

The reference's pin-label encoder (src/labels.hpp:209-229) sizes the per-label count fields from the labels' PIN counts
(num_pins_width = compute_byte_width(max pins of a label)) and writes the count of a label's single-component ids with the
same width (labels.hpp: num_cc_labels, read back at :571).  A label with more than 255 single-component ids in a volume
whose labels all have fewer than 256 pins overflows that one-byte count: the reference writes a pin section that its own
decoder parses out of step, and crackle.decompress(crackle.compress(x, allow_pins=True)) != x.  It takes few label values
with many small components each: uint8 volumes of tiny Voronoi cells two slices deep, 1024 x 244 ... 256 x 4.

crackle_amd's encoder is bit-exact against the reference's (the requirement), so it writes the same bytes, and its decoder,
like the reference's, returns wrong labels for them.  This script shows both: our bytes equal the reference's, the reference
decodes its own stream wrongly, and where the count field overflows.  (Run on a GPU box; the first half needs none.)"""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from crackle_amd import synth
from oracle import oracle

chk = oracle.best()


def bw(x):
  return 1 if x < 256 else 2 if x < 65536 else 4 if x < 2 ** 32 else 8


def first_overflow(b, shape):
  """walks the pin section the way labels.hpp:556-590 does; returns the first label whose record is out of step"""
  sx, sy, sz = shape
  sw = 1 << (int.from_bytes(b[5:7], "little") >> 2 & 3)
  nlb = int.from_bytes(b[20:28], "little")
  lb = b[29 + 4 * (sz + 1): 29 + 4 * (sz + 1) + nlb]
  nu = int.from_bytes(lb[sw:sw + 8], "little")
  off = 8 + sw * (nu + 1) + bw(sx * sy) * sz
  comb = lb[off]; off += 1
  npw, dw, ccw = 1 << (comb & 3), 1 << ((comb >> 2) & 3), 1 << ((comb >> 4) & 3)
  iw, vol, i = bw(sx * sy * sz), sx * sy * sz, off
  for label in range(nu):
    n = int.from_bytes(lb[i:i + npw], "little"); i += npw
    idx = 0
    for j in range(n):
      idx += int.from_bytes(lb[i + j * iw: i + (j + 1) * iw], "little")
      if idx >= vol:
        return f"label {label}: a pin outside the volume (count fields of {npw} byte, the record before it was read short)"
    i += n * (iw + dw)
    c = int.from_bytes(lb[i:i + npw], "little"); i += npw + c * ccw
  return None if i == len(lb) else f"the records end at byte {i} of {len(lb)}"


CASES = [((1024, 248, 3), "uint8"), ((1024, 248, 4), "uint8"), ((1024, 248, 4), "uint16"), ((1024, 320, 4), "uint8")]
if len(sys.argv) == 1:      # one process per case: the reference may crash on the section it wrote
  import subprocess
  for k in range(len(CASES)):
    r = subprocess.run([sys.executable, __file__, str(k)], capture_output=True, text=True)
    print((r.stdout.strip().splitlines() or [f"{CASES[k]}: the process died (return code {r.returncode})"])[-1], flush=True)
  sys.exit(0)
shape, dt = CASES[int(sys.argv[1])]
dt = np.dtype(dt).type
arr = synth.as_numpy_f(synth.voronoi_labels(shape, dt, seed=11, cell=(4, 4, 2)))
want = chk.compress(arr, allow_pins=True)
line = f"{shape} {np.dtype(dt).name}: section: {first_overflow(want, shape) or 'in step'}"
print(line, flush=True)
try:
  import crackle_amd
  got = crackle_amd.compress(arr, allow_pins=True)
  line += f"; our bytes equal the reference's: {got == want}"
  print(line, flush=True)
  try:
    back = crackle_amd.decompress(want)
    line += f"; our decode {'correct' if np.array_equal(back, arr) else 'wrong (%d voxels)' % int((back != arr).sum())}"
  except RuntimeError as exc:
    line += f"; our decoder refuses the stream: {str(exc)[:90]}"
except (OSError, ImportError) as exc:      # no GPU here
  line += f"; (HIP path not run: {type(exc).__name__})"
print(line, flush=True)
ref = chk.decompress(want).reshape(shape, order="F")      # (last: the compiled reference may crash on the section it wrote)
line += f"; reference decodes its own stream {'correctly' if np.array_equal(ref, arr) else 'WRONGLY (%d voxels)' % int((ref != arr).sum())}"
print(line, flush=True)
