"""GPU tests of point_cloud (crackle_amd/csrc/ckl_contours.hpp: k_contour_dirs, k_trace_contours,
k_contour_components, k_contour_emit) against the oracle's restatement of
operations::point_cloud + dual_graph::extract_contours (src/operations.hpp:183-262,
src/dual_graph.hpp:133-275), which tests/test_oracle.py pins to the compiled reference.
Equality is exact: the same labels, and for every label the same (x, y, z) triples in the order
the reference appends them with parallel = 1."""
import numpy as np
import pytest

import crackle_amd
from crackle_amd import operations, synth

pytestmark = pytest.mark.gpu


def _same(got, want, what):
  assert sorted(got) == sorted(want), what
  for k in want:
    assert np.array_equal(np.asarray(got[k]).ravel(), want[k]), (what, k)


def _rings(sx, sy, sz):
  """Nested rings, one-pixel lines, isolated pixels, a component inside the hole of another one,
  structures touching the border."""
  v = np.zeros((sx, sy, sz), dtype=np.uint16)
  for z in range(sz):
    o = z % 3
    v[2 + o:sx - 2, 2:sy - 2, z] = 1
    v[4 + o:sx - 4, 4:sy - 4, z] = 0            # ring of width 2
    v[6 + o:sx - 6, 6:sy - 6, z] = 2            # a block inside the hole
    v[8 + o:sx - 8, 8:sy - 8, z] = 2 if z % 2 else 3
    v[sx // 2, :, z] = 4                        # a one-pixel line across everything, border to border
    v[0, 0, z] = 5                              # isolated corner pixel
    v[sx - 1, sy // 3, z] = 6
    v[3, sy - 1, z] = 6
    v[1::4, sy - 4, z] = 7                      # a row of isolated pixels
  return np.asfortranarray(v)


def _checker(sx, sy, sz, dtype=np.uint8):
  x, y, z = np.meshgrid(np.arange(sx), np.arange(sy), np.arange(sz), indexing="ij")
  return np.asfortranarray((((x // 2 + y // 3 + z) % 2) + 1).astype(dtype))


def _spiral(n):
  v = np.zeros((n, n, 2), dtype=np.uint32)
  x0, y0, x1, y1 = 0, 0, n - 1, n - 1
  while x1 - x0 > 3 and y1 - y0 > 3:
    v[x0:x1 + 1, y0, :] = 9
    v[x1, y0:y1 + 1, :] = 9
    v[x0 + 2:x1 + 1, y1, :] = 9
    v[x0 + 2, y0 + 2:y1 + 1, :] = 9
    x0, y0, x1, y1 = x0 + 2, y0 + 2, x1 - 2, y1 - 2
  v[:, :, 1] = v[::-1, :, 1]
  return np.asfortranarray(v)


VOLUMES = {
  "flat_zero": lambda: np.zeros((8, 13, 1), dtype=np.uint32, order="F"),      # the reference's own test (automated_test.py:677-700)
  "voronoi_u32": lambda: synth.as_numpy_f(synth.voronoi_labels((96, 80, 7), np.uint32, seed=11, cell=(16, 16, 4))),
  "voronoi_u64": lambda: synth.as_numpy_f(synth.voronoi_labels((70, 45, 5), np.uint64, seed=12, cell=(12, 12, 4), offset=1 << 40)),
  "voronoi_mod": lambda: synth.as_numpy_f(synth.voronoi_labels((128, 96, 6), np.uint16, seed=13, cell=(10, 10, 3), modulus=5)),
  "noise3": lambda: synth.random_labels((40, 33, 4), np.uint8, seed=9, high=3),          # PERMISSIBLE crack format
  "noise_many": lambda: synth.random_labels((33, 29, 3), np.uint32, seed=10, high=1000),
  "rings": lambda: _rings(40, 36, 4),
  "checker": lambda: _checker(37, 41, 3),
  "spiral": lambda: _spiral(45),
  "one_row": lambda: np.asfortranarray((np.arange(77) // 5 % 3).astype(np.uint8).reshape(77, 1, 1)),
  "one_col": lambda: np.asfortranarray((np.arange(50) // 3 % 2).astype(np.uint8).reshape(1, 50, 1)),
  "one_voxel": lambda: np.asfortranarray(np.full((1, 1, 1), 3, np.uint8)),
  "wide": lambda: synth.as_numpy_f(synth.voronoi_labels((1100, 40, 2), np.uint32, seed=14, cell=(32, 32, 8))),
}


@pytest.mark.parametrize("name", sorted(VOLUMES))
def test_point_cloud_equals_oracle(name, checker):
  arr = VOLUMES[name]()
  for kw in (dict(), dict(markov_model_order=3), dict(allow_pins=True)):
    binary = checker.compress(arr, **kw)
    for skip in (False, True):
      want = checker.point_cloud(binary, 0, -1, None, skip)
      got = operations._point_cloud_raw(binary, 0, -1, None, skip, 0)
      _same(got, want, (name, kw, skip))


def test_point_cloud_ranges_and_label_selection(checker):
  arr = VOLUMES["voronoi_u32"]()
  binary = checker.compress(arr)
  labels = [int(arr[3, 3, 0]), int(arr[50, 50, 3]), 99999]
  for z0, z1 in ((0, -1), (2, 5), (6, 7), (-1, -1), (3, 100)):
    _same(operations._point_cloud_raw(binary, z0, z1, None, False, 0), checker.point_cloud(binary, z0, z1, None, False), (z0, z1))
    _same(operations._point_cloud_raw(binary, z0, z1, labels, True, 0), checker.point_cloud(binary, z0, z1, labels, True), (z0, z1, "sel"))
  # the reference's Python surface (crackle/codec.py:804-872): shapes and errors
  one = crackle_amd.point_cloud(binary, labels[0])
  assert one.dtype == np.uint16 and one.ndim == 2 and one.shape[1] == 3
  assert np.array_equal(one.ravel(), checker.point_cloud(binary, 0, arr.shape[2], [labels[0]], True)[labels[0]])
  both = crackle_amd.point_cloud(binary, labels[:2])
  assert sorted(both) == sorted(labels[:2])
  everything = crackle_amd.point_cloud(binary, skip_background=False)
  assert sorted(everything) == sorted(int(v) for v in np.unique(arr))
  with pytest.raises(ValueError):
    crackle_amd.point_cloud(binary, 99999)
  with pytest.raises(RuntimeError, match="Invalid range"):
    operations._point_cloud_raw(binary, 5, 5, None, False, 0)
  with pytest.raises(RuntimeError, match="Invalid range"):
    checker.point_cloud(binary, 5, 5, None, False)


def test_point_cloud_signed_labels(checker):
  """point_cloud<LABEL> keys its map by the unsigned type of the data width (operations.hpp:274-300)."""
  arr = np.asfortranarray((synth.as_numpy_f(synth.voronoi_labels((50, 40, 3), np.uint32, seed=15, cell=(10, 10, 2), modulus=7)).astype(np.int64) - 3).astype(np.int32))
  binary = checker.compress(arr)
  want = checker.point_cloud(binary, 0, -1, None, False)
  assert any(k >= 1 << 31 for k in want)
  _same(operations._point_cloud_raw(binary, 0, -1, None, False, 0), want, "signed")


def test_point_cloud_paths(checker, monkeypatch):
  """The visited bits in HBM (slices of more than ~1.2 M pixels use it), the second pass with
  worst-case buffers after an overflow, several z-chunks."""
  arr = VOLUMES["rings"]()
  binary = checker.compress(arr)
  want = checker.point_cloud(binary, 0, -1, None, False)
  monkeypatch.setenv("CKL_CONTOUR_HBM_VISITED", "1")
  _same(operations._point_cloud_raw(binary, 0, -1, None, False, 0), want, "hbm visited")
  monkeypatch.delenv("CKL_CONTOUR_HBM_VISITED")
  monkeypatch.setenv("CKL_CONTOUR_SMALL", "64")
  _same(operations._point_cloud_raw(binary, 0, -1, None, False, 0), want, "second pass")
  monkeypatch.delenv("CKL_CONTOUR_SMALL")
  for name in ("rings", "checker", "spiral", "noise3", "voronoi_mod"):      # every start walked, as the reference does (no walked-loop memo)
    b = checker.compress(VOLUMES[name]())
    w = checker.point_cloud(b, 0, -1, None, False)
    monkeypatch.setenv("CKL_CONTOUR_NO_MEMO", "1")
    _same(operations._point_cloud_raw(b, 0, -1, None, False, 0), w, ("no memo", name))
    monkeypatch.delenv("CKL_CONTOUR_NO_MEMO")
    monkeypatch.setenv("CKL_CONTOUR_HBM_VISITED", "1")
    _same(operations._point_cloud_raw(b, 0, -1, None, False, 0), w, ("hbm visited", name))
    monkeypatch.delenv("CKL_CONTOUR_HBM_VISITED")
  big = synth.as_numpy_f(synth.voronoi_labels((2048, 1024, 2), np.uint32, seed=16, cell=(48, 48, 8)))      # 2 M pixels per slice
  b2 = checker.compress(big)
  _same(operations._point_cloud_raw(b2, 0, -1, None, True, 0), checker.point_cloud(b2, 0, -1, None, True), "2048 x 1024")


def test_point_cloud_c2_slice_shape(checker):
  """BASELINE.json's C2 slice shape (1024 x 1024), a few slices."""
  arr = synth.as_numpy_f(synth.voronoi_labels((1024, 1024, 4), np.uint32, seed=17, cell=(32, 32, 8)))
  binary = checker.compress(arr)
  _same(operations._point_cloud_raw(binary, 0, -1, None, False, 0), checker.point_cloud(binary, 0, -1, None, False), "c2")


def test_point_cloud_golden_streams_against_the_reference_fixture():
  """Every small golden stream (C order, pins, markov, PERMISSIBLE noise, single voxels, rows, the
  empty stream) through the device path, against tests/golden/point_cloud.json — digests of the
  compiled reference's own output (tests/gen_golden.py --ops)."""
  import json
  import os
  from gen_golden import POINT_CLOUD_ARGS, point_cloud_digest
  from util import golden
  with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "point_cloud.json")) as f:
    want = json.load(f)
  g = golden()
  for name in sorted(g):
    for tag, (z0, z1, labels, skip) in POINT_CLOUD_ARGS.items():
      try:
        got = point_cloud_digest(operations._point_cloud_raw(g[name], z0, z1, labels, skip, 0))
      except RuntimeError as exc:
        got = "error: " + str(exc)
      assert got == want[name][tag], (name, tag)


def test_point_cloud_c1_full_size_against_the_reference_fixture():
  """BASELINE.json configs[1] (512 x 512 x 128 uint32) whole: the device's point cloud against the
  digests of the compiled reference's own output (tests/golden/point_cloud_xl.json, written by
  tests/gen_golden.py --ops): whole range, background skipped, slices [1, 65)."""
  import json
  import os
  import torch
  from crackle_amd import distributed as ckd
  from gen_golden import POINT_CLOUD_ARGS, point_cloud_digest
  with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "point_cloud_xl.json")) as f:
    want = json.load(f)["c1_512x512x128_u32"]
  vol = synth.voronoi_labels((512, 512, 128), np.uint32, seed=2, device=torch.device("cuda:0"))
  binary = bytes(ckd.HipBackend(0).encode(vol, (512, 512, 128), False, True, 0, None))
  for tag, (z0, z1, labels, skip) in POINT_CLOUD_ARGS.items():
    got = point_cloud_digest(operations._point_cloud_raw(binary, z0, 65 if tag == "z1" else z1, labels, skip, 0))
    assert got == want[tag], tag
