"""The `fastcrackle`-compatible pybind11 module (crackle_amd/csrc/fastcrackle.cpp): what the
reference's crackle/codec.py imports, with the same positional signatures
(src/fastcrackle.cpp:84-210, 641-669), on top of libcrackle_amd.so."""
import importlib.util
import os
import sys

import numpy as np
import pytest

from crackle_amd import build as ckl_build, synth
import golden_cases
from util import golden, label_format, flat_1d

SMALL = golden_cases.small_cases()


def _module():
  path = ckl_build.fastcrackle_path()
  assert os.path.exists(path), "build the module first (python -m crackle_amd.build)"
  spec = importlib.util.spec_from_file_location("fastcrackle", path)
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


def test_module_imports_and_exposes_the_reference_names():
  m = _module()
  for name in ("compress", "decompress", "reencode_markov", "voxel_counts", "centroids", "bounding_boxes", "voxel_connectivity_graph", "array_equal", "mode_pooling_2x2x1", "point_cloud"):
    assert callable(getattr(m, name)), name
  # header problems surface without a device, like the reference's CrackleHeader constructor
  with pytest.raises(RuntimeError):
    m.decompress(b"nope" + bytes(40), 0, -1, 1, None)
  with pytest.raises(RuntimeError, match="1D"):
    m.decompress(np.zeros((4, 8), np.uint8), 0, -1, 1, None)


@pytest.mark.gpu
def test_goldens_through_the_module():
  """crackle/codec.py:670,729 call it positionally: do the same."""
  m = _module()
  names = sorted(SMALL)[::5]
  for name in names:
    arr, kw = SMALL[name]
    want = golden()[name]
    got = m.compress(np.asfortranarray(arr), bool(kw["allow_pins"]), arr.flags.f_contiguous, kw["markov_model_order"], False, True, 0, 0)
    assert got == want, name
    back = m.decompress(want, 0, -1, 0, None)
    assert back.ndim == 1 and back.dtype == arr.dtype and back.size == arr.size, name
    if arr.size:
      assert np.array_equal(back, flat_1d(arr)), name


@pytest.mark.gpu
def test_ranges_labels_and_statistics_through_the_module():
  m = _module()
  arr, _ = SMALL["c0_voronoi_u8"]
  b = golden()["c0_voronoi_u8_pins_m5"]
  part = m.decompress(b, 3, 9, 1, None)
  assert np.array_equal(part, flat_1d(np.asfortranarray(arr[:, :, 3:9])))
  lbl = int(arr[10, 10, 5])
  img = m.decompress(b, 0, -1, 1, lbl)
  assert img.dtype == np.uint8 and np.array_equal(img.astype(bool), flat_1d(arr) == lbl)
  cts = m.voxel_counts(b, 0, -1, 1)
  vals, n = np.unique(arr, return_counts=True)
  assert cts == {int(v): int(c) for v, c in zip(vals, n)}
  cen = m.centroids(b, 0, -1, 1)
  bbx = m.bounding_boxes(b, 0, -1, 1)
  x, y, z = np.nonzero(arr == lbl)
  assert np.allclose(cen[lbl], [x.mean(), y.mean(), z.mean()])
  assert bbx[lbl].tolist() == [x.min(), y.min(), z.min(), x.max(), y.max(), z.max()]
  re0 = m.reencode_markov(b, 0, 1)
  assert np.array_equal(m.decompress(re0, 0, -1, 1, None), flat_1d(arr))
  vcg = m.voxel_connectivity_graph(golden()["c0_voronoi_u8"], 2, 7, 1, 4)
  assert vcg.shape == (64, 64, 5) and vcg.flags.f_contiguous
  same_x = arr[1:, :, 2:7] == arr[:-1, :, 2:7]
  assert np.array_equal((vcg[1:, :, :] & 2) != 0, same_x)


@pytest.mark.gpu
def test_fastcrackle_point_cloud(checker):
  """fastcrackle.point_cloud (src/fastcrackle.cpp:315-345): dict label -> flat uint16 triples."""
  fc = _module()
  arr = synth.as_numpy_f(synth.voronoi_labels((64, 48, 5), np.uint32, seed=21, cell=(12, 12, 3)))
  binary = checker.compress(arr)
  want = checker.point_cloud(binary, 1, 4, None, True)
  got = fc.point_cloud(binary, 1, 4, None, True, 1)
  assert sorted(got) == sorted(want)
  for k in want:
    assert got[k].dtype == np.uint16 and np.array_equal(got[k], want[k])
  some = sorted(want)[:2]
  got = fc.point_cloud(binary, labels=some)
  assert sorted(got) == some
