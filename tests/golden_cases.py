"""Definitions of the golden-vector grid.

Shared by tests/gen_golden.py (which runs the compiled reference, here only) and by
the tests (which replay the inputs through the oracle restatement and the HIP path).
Inputs are rebuilt from these definitions; the *expected bytes* live in
tests/golden/golden.npz (small cases, exact .ckl bytes) and
tests/golden/manifest.json (larger cases: sha256 + length + section hashes).
"""
import numpy as np

from crackle_amd import synth


def _rand(shape, hi, dtype, seed, order):
  # counter-hash generator (not numpy's RNG) so inputs are reproducible everywhere
  a = synth.random_labels(shape, dtype=dtype, seed=seed, high=hi)
  return a if order == "F" else np.ascontiguousarray(a)


def spurious_a():
  # input data of the reference's test_spurious_branch_elimination (automated_test.py:908-935)
  return np.array([
    [0,0,0,0,0,0,0,0,0,0],
    [0,0,0,0,0,0,0,0,0,0],
    [0,0,1,1,2,2,0,0,0,0],
    [0,0,1,1,2,2,0,0,0,0],
    [0,0,4,4,3,3,0,0,0,0],
    [0,0,4,4,3,3,0,0,0,0],
    [0,0,0,0,0,0,0,0,0,0],
    [0,0,0,0,0,0,0,0,0,0],
    [0,0,0,0,0,0,0,0,0,0],
  ], dtype=np.uint8).T


def spurious_b():
  return np.array([
    [  0, 139, 139, 139, 139],
    [  0, 139,   0, 139, 139],
    [  0, 161,   0,   0, 161],
    [161, 161, 161, 161, 161],
  ], dtype=np.uint8).T


def _checker(n, dtype=np.uint8):
  x, y = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
  return np.asfortranarray(((x + y) & 1).astype(dtype)[:, :, None])


def _spiral(n):
  a = np.zeros((n, n), dtype=np.uint8)
  x = y = 0
  dx, dy = 1, 0
  lo_x, hi_x, lo_y, hi_y = 0, n - 1, 0, n - 1
  # draw a 1-pixel spiral wall with 1-pixel gaps
  k = 0
  while lo_x <= hi_x and lo_y <= hi_y and k < 4 * n:
    for i in range(lo_x, hi_x + 1): a[i, lo_y] = 1
    for j in range(lo_y, hi_y + 1): a[hi_x, j] = 1
    for i in range(lo_x + 2, hi_x + 1): a[i, hi_y] = 1
    for j in range(lo_y + 2, hi_y + 1): a[lo_x + 2 if lo_x + 2 <= hi_x else hi_x, j] = 1
    lo_x += 4; lo_y += 2; hi_x -= 2; hi_y -= 2
    k += 1
  return np.asfortranarray(a[:, :, None])


def _dots(n):
  a = np.zeros((n, n, 2), dtype=np.uint16)
  a[1::3, 1::3, 0] = np.arange(1, len(range(1, n, 3)) ** 2 + 1).reshape(len(range(1, n, 3)), -1)
  a[2::4, 2::4, 1] = 7
  return np.asfortranarray(a)


def _kat4():
  a = np.zeros((4, 4, 1), np.uint8, order="F")
  a[1:3, 1:3, 0] = 7
  return a


def _nlabels(n, dtype):
  # exactly n distinct labels, one 2x2 block each, to pin key-width flips (255/256/257)
  side = int(np.ceil(np.sqrt(n)))
  a = np.zeros((2 * side, 2 * side, 1), dtype=dtype, order="F")
  k = 0
  for i in range(side):
    for j in range(side):
      a[2*i:2*i+2, 2*j:2*j+2, 0] = k if k < n else n - 1
      k += 1
  return a


def small_cases():
  """name -> (labels ndarray, kwargs for fastcrackle-style compress)"""
  cases = {}

  def add(name, arr, **kw):
    kw.setdefault("allow_pins", False)
    kw.setdefault("markov_model_order", 0)
    cases[name] = (arr, kw)

  add("kat_4x4", _kat4())
  add("kat_4x4_m1", _kat4(), markov_model_order=1)
  add("kat_4x4_c", np.ascontiguousarray(_kat4()))
  add("kat_ones_300", np.ones((300, 300, 2), np.uint32, order="F"))
  add("spurious_a", np.asfortranarray(spurious_a()[:, :, None]))
  add("spurious_b", np.asfortranarray(spurious_b()[:, :, None]))
  add("spurious_a_m2", np.asfortranarray(spurious_a()[:, :, None]), markov_model_order=2)
  for dt in (np.uint8, np.uint16, np.uint32, np.uint64):
    for order in ("F", "C"):
      for mk in (0, 1, 2, 3, 5):
        for pins in (False, True):
          nm = f"rand_17x13x5_{np.dtype(dt).name}_{order}_m{mk}_p{int(pins)}"
          add(nm, _rand((17, 13, 5), 4, dt, seed=len(cases), order=order), markov_model_order=mk, allow_pins=pins)
  for mk in (0, 3):
    add(f"rand_4x4x1_m{mk}", _rand((4, 4, 1), 5, np.uint8, 7, "F"), markov_model_order=mk)
    add(f"rand_64x63x3_m{mk}", _rand((64, 63, 3), 40, np.uint16, 8, "F"), markov_model_order=mk)
    add(f"rand_256x255x1_m{mk}", _rand((256, 255, 1), 40, np.uint8, 9, "F"), markov_model_order=mk)
    add(f"rand_254x257x2_m{mk}", _rand((254, 257, 2), 3, np.uint8, 10, "F"), markov_model_order=mk)
  add("rand_100x100x10", _rand((100, 100, 10), 40, np.uint32, 11, "F"))
  add("noise_2000", _rand((64, 64, 4), 2000, np.uint32, 12, "F"))
  add("noise_2000_m5", _rand((64, 64, 4), 2000, np.uint32, 12, "F"), markov_model_order=5)
  add("binary_noise", _rand((64, 64, 4), 2, np.uint8, 13, "F"))
  add("checker_16", _checker(16))
  add("checker_16_m3", _checker(16), markov_model_order=3)
  add("spiral_40", _spiral(40))
  add("dots_30", _dots(30))
  add("dots_30_pins", _dots(30), allow_pins=True)
  add("single_voxel", np.full((1, 1, 1), 9, np.uint16, order="F"))
  add("row_33", _rand((33, 1, 7), 3, np.uint8, 14, "F"))
  add("col_29", _rand((1, 29, 4), 3, np.uint8, 15, "F"))
  add("empty_000", np.zeros((0, 0, 0), np.uint8, order="F"))
  add("empty_503", np.zeros((5, 0, 3), np.uint32, order="F"), markov_model_order=3, allow_pins=True)
  add("zeros_50", np.zeros((50, 50, 5), np.uint32, order="F"), markov_model_order=3)
  add("u32max", np.full((20, 20, 3), 2**32 - 1, np.uint32, order="F"))
  add("u64_2p32", np.full((20, 20, 3), 2**32, np.uint64, order="F"))
  add("maxlabel_255", np.asfortranarray(np.array([[[0, 255]]], dtype=np.uint16)))
  add("maxlabel_256", np.asfortranarray(np.array([[[0, 256]]], dtype=np.uint16)))
  add("maxlabel_65535", np.asfortranarray(np.array([[[0, 65535]]], dtype=np.uint32)))
  add("maxlabel_65536", np.asfortranarray(np.array([[[0, 65536]]], dtype=np.uint32)))
  add("arange_16", np.arange(16 * 16 * 4, dtype=np.uint32).reshape((16, 16, 4), order="F"))
  add("arange_16_pins", np.arange(16 * 16 * 4, dtype=np.uint32).reshape((16, 16, 4), order="F"), allow_pins=True)
  add("arange_2d", np.arange(16 * 16, dtype=np.uint32).reshape((16, 16, 1), order="F"))
  for n in (255, 256, 257):
    add(f"nlabels_{n}", _nlabels(n, np.uint16))
  add("pins_sz1", _rand((17, 13, 1), 4, np.uint8, 16, "F"), allow_pins=True)
  v = synth.as_numpy_f(synth.voronoi_labels((64, 64, 16), np.uint8, seed=0, cell=(8, 8, 4)))
  add("c0_voronoi_u8", v)
  add("c0_voronoi_u8_m5", v, markov_model_order=5)
  add("c0_voronoi_u8_pins", v, allow_pins=True)
  add("c0_voronoi_u8_pins_m5", v, allow_pins=True, markov_model_order=5)
  add("c0_voronoi_u8_c", np.ascontiguousarray(v))
  return cases


def large_cases():
  """name -> (generator thunk, kwargs); expected = sha256 manifest"""
  def vor(shape, dt, seed, cell, **kw):
    return lambda: synth.as_numpy_f(synth.voronoi_labels(shape, dt, seed=seed, cell=cell, **kw))
  cases = {
    "vor_128x128x16_u32": (vor((128, 128, 16), np.uint32, 1, (16, 16, 4)), dict()),
    "vor_128x128x16_u32_m5": (vor((128, 128, 16), np.uint32, 1, (16, 16, 4)), dict(markov_model_order=5)),
    "vor_128x128x16_u32_pins": (vor((128, 128, 16), np.uint32, 1, (16, 16, 4)), dict(allow_pins=True)),
    "vor_300x270x4_u16": (vor((300, 270, 4), np.uint16, 2, (32, 32, 8)), dict()),
    "vor_512x512x8_u32": (vor((512, 512, 8), np.uint32, 1, (32, 32, 8)), dict()),
    "vor_512x512x8_u32_m5": (vor((512, 512, 8), np.uint32, 1, (32, 32, 8)), dict(markov_model_order=5)),
    "vor_257x255x6_u64": (vor((257, 255, 6), np.uint64, 3, (32, 32, 8), offset=1 << 40), dict()),
    "vor_1024x1024x2_u32": (vor((1024, 1024, 2), np.uint32, 2, (32, 32, 8)), dict()),
    "noise_256x256x4_u32": (lambda: synth.random_labels((256, 256, 4), np.uint32, 5, 2000), dict()),
  }
  for v in cases.values():
    v[1].setdefault("allow_pins", False)
    v[1].setdefault("markov_model_order", 0)
  return cases


def xl_cases():
  """BASELINE.json configurations at (or near) their full per-GPU size; expected = sha256
  manifest written by `python tests/gen_golden.py --xl` (tests/golden/manifest_xl.json).
  name -> (generator thunk, kwargs, slab description used by bench.py / the GPU tests)"""
  def vor(shape, dt, seed, **kw):
    return lambda: synth.as_numpy_f(synth.voronoi_labels(shape, dt, seed=seed, cell=(32, 32, 8), **kw))
  cases = {
    # C1: 512 x 512 x 128 uint32
    "c1_512x512x128_u32": (vor((512, 512, 128), np.uint32, 2), dict()),
    # C2: the bench workload itself (bench.py asserts this sha on its own output)
    "c2_1024x1024x512_u32": (vor((1024, 1024, 512), np.uint32, 2), dict()),
    # C3: one 16-slice slab of the uint64 volume (stored width 8)
    "c3_1024x1024x16_u64": (vor((1024, 1024, 16), np.uint64, 2, offset=1 << 40), dict()),
    # C3 whole: 1024^3 uint64 (8.6 GB of labels; a quarter of an hour on the build container's cores)
    "c3_1024x1024x1024_u64": (vor((1024, 1024, 1024), np.uint64, 2, offset=1 << 40), dict()),
    # C4: 2048-wide slices, pins + markov order 5
    "c4_2048x2048x8_u32_pins_m5": (vor((2048, 2048, 8), np.uint32, 2), dict(allow_pins=True, markov_model_order=5)),
    "c4_2048x2048x8_u32_m5": (vor((2048, 2048, 8), np.uint32, 2), dict(markov_model_order=5)),
    # C4 whole: 2048 x 2048 x 256 uint32, pins + markov order 5 (the reference's pin solver over 1.07 G voxels)
    "c4_2048x2048x256_u32_pins_m5": (vor((2048, 2048, 256), np.uint32, 2), dict(allow_pins=True, markov_model_order=5)),
  }
  for v in cases.values():
    v[1].setdefault("allow_pins", False)
    v[1].setdefault("markov_model_order", 0)
  return cases
