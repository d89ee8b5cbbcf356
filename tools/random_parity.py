"""Randomised parity sweep on the GPU: random shapes / dtypes / label patterns / options; the
encoder's bytes must equal the checker's, the decoder must give the volume back, and the
consumers (statistics, VCG, reencode, point clouds, array_equal, mode pooling) must agree with
numpy / the checker.
usage: python tools/random_parity.py [cases] [seed]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import crackle_amd
from crackle_amd import operations, synth
from oracle import oracle


def make(rng):
  sx, sy = int(rng.integers(1, 300)), int(rng.integers(1, 300))
  if rng.random() < 0.25:
    sx = int(rng.choice([1, 2, 3, 4, 31, 32, 33, 64, 127, 128, 129, 256, 1024, 1025]))
  sz = int(rng.integers(1, 7)) if rng.random() < 0.8 else int(rng.integers(7, 80))      # deep volumes: pin runs, several label registers
  if sz > 6:
    sx, sy = min(sx, 96), min(sy, 96)
  dt = [np.uint8, np.uint16, np.uint32, np.uint64][int(rng.integers(0, 4))]
  kind = int(rng.integers(0, 6))
  hi = int(min(np.iinfo(dt).max, [3, 50, 2000, 1 << 20][int(rng.integers(0, 4))]))
  if kind == 0:
    cell = tuple(int(rng.integers(2, 40)) for _ in range(2)) + (int(rng.integers(1, 5)),)
    arr = synth.as_numpy_f(synth.voronoi_labels((sx, sy, sz), dt, seed=int(rng.integers(0, 1 << 30)), cell=cell))
  elif kind == 1:
    arr = synth.random_labels((sx, sy, sz), dt, seed=int(rng.integers(0, 1 << 30)), high=max(hi, 2))
  elif kind == 2:
    arr = np.full((sx, sy, sz), int(rng.integers(0, hi + 1)), dt, order="F")
  elif kind == 3:      # stripes / checker patterns
    x, y, z = np.meshgrid(np.arange(sx), np.arange(sy), np.arange(sz), indexing="ij")
    p = int(rng.integers(1, 5))
    arr = np.asfortranarray((((x // p) + (y // p) * int(rng.integers(0, 2)) + z) % max(2, min(hi, 7))).astype(dt))
  elif kind == 4:      # blobs on background 0
    arr = np.zeros((sx, sy, sz), dt, order="F")
    for _ in range(int(rng.integers(1, 12))):
      x0, y0 = int(rng.integers(0, sx)), int(rng.integers(0, sy))
      arr[x0:x0 + int(rng.integers(1, 40)), y0:y0 + int(rng.integers(1, 40)), int(rng.integers(0, sz)):] = int(rng.integers(1, hi + 1))
  else:                # sparse single pixels
    arr = np.zeros((sx, sy, sz), dt, order="F")
    n = int(rng.integers(1, 200))
    arr[rng.integers(0, sx, n), rng.integers(0, sy, n), rng.integers(0, sz, n)] = rng.integers(1, hi + 1, n).astype(dt)
  kw = dict(markov_model_order=int(rng.choice([0, 0, 0, 1, 3, 5, 6])))
  if rng.random() < 0.3:
    kw["allow_pins"] = True
  return arr, kw


def main():
  cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
  rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
  chk = oracle.best()
  bad = 0
  t0 = time.time()
  for i in range(cases):
    arr, kw = make(rng)
    tag = f"case {i}: {arr.shape} {arr.dtype.name} {kw}"
    want = chk.compress(arr, **kw)
    got = crackle_amd.compress(arr, **{k: (1 if k == "allow_pins" else v) for k, v in kw.items()})
    ok = got == want
    try:
      back = crackle_amd.decompress(want)
    except RuntimeError as exc:      # (the reference's own pin-section overflow is refused: tools/repro_pins_u8.py)
      print("REFUSED", tag, str(exc)[:100], flush=True)
      bad += 1
      continue
    ok = ok and back.shape == arr.shape and np.array_equal(back, arr)
    if i % 3 == 0 and arr.size:
      u, c = np.unique(arr, return_counts=True)
      ok = ok and crackle_amd.voxel_counts(want) == {int(a): int(b) for a, b in zip(u, c)}
      ok = ok and np.array_equal(crackle_amd.voxel_connectivity_graph(want, 6), chk.voxel_connectivity_graph(want, 6))
      order = 2 if kw["markov_model_order"] != 2 else 0
      ok = ok and crackle_amd.reencode(want, order) == chk.reencode(want, order)
    if i % 4 == 1 and arr.size:
      z0 = int(rng.integers(0, arr.shape[2]))
      args = (z0, int(rng.integers(z0 + 1, arr.shape[2] + 1)), None, bool(rng.integers(0, 2)))
      a, b = operations._point_cloud_raw(want, args[0], args[1], args[2], args[3], 0), chk.point_cloud(want, *args)
      ok = ok and sorted(a) == sorted(b) and all(np.array_equal(a[k], b[k]) for k in b)
    if i % 7 == 2 and arr.size:
      ok = ok and crackle_amd.array_equal(want, got)
      pooled = chk.mode_pooling_2x2x1(want)
      if len({(crackle_amd.header(b).crack_format, crackle_amd.header(b).stored_data_width >= 0) for b in pooled}) == 1:
        ok = ok and crackle_amd.mode_pooling_2x2x1(want) == crackle_amd.zstack(pooled)
      else:      # pooled slices of different crack formats do not stack (crackle/operations.py:474-475 raises too)
        try:
          crackle_amd.mode_pooling_2x2x1(want)
          ok = False
        except ValueError:
          pass
    if not ok:
      bad += 1
      print("MISMATCH", tag, flush=True)
    if i % 20 == 19:
      print(f"{i + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
  print(f"done: {cases} cases, {bad} mismatches")
  sys.exit(1 if bad else 0)


main()
