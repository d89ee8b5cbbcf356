"""Stream surgery (crackle_amd.operations: zstack / zsplit / zshatter, host only) against the
oracle: sub-streams decode to the right slices, shattering and stacking again reproduces the
original bytes, stacking slab streams equals compressing the whole (automated_test.py:448-487)."""
import numpy as np
import pytest

import crackle_amd
from crackle_amd import synth


def _vol(kind):
  if kind == "voronoi":
    return synth.as_numpy_f(synth.voronoi_labels((72, 60, 9), np.uint16, seed=14, cell=(12, 12, 3)))
  if kind == "wide":
    v = synth.as_numpy_f(synth.voronoi_labels((40, 33, 6), np.uint32, seed=15, cell=(10, 10, 3), modulus=150)).copy(order="F")
    v[2:9, 3:8, 4] = 3_000_000      # one slice needs 4 stored bytes, the others 1
    return v
  if kind == "noise":
    return synth.random_labels((30, 28, 5), np.uint32, seed=16, high=500)
  raise ValueError(kind)


@pytest.mark.parametrize("kind,order", [("voronoi", 0), ("voronoi", 3), ("wide", 0), ("noise", 2)])
def test_zsplit_zshatter_zstack(port, kind, order):
  vol = _vol(kind)
  whole = port.compress(vol, markov_model_order=order)
  sz = vol.shape[2]
  parts = crackle_amd.zshatter(whole)
  assert len(parts) == sz
  for z, p in enumerate(parts):
    got = port.decompress(p).reshape(vol.shape[:2] + (1,), order="F")
    assert np.array_equal(got[:, :, 0], vol[:, :, z]), (kind, z)
  assert crackle_amd.zstack(parts) == whole
  before, middle, after = crackle_amd.zsplit(whole, 2)
  assert np.array_equal(port.decompress(before).reshape(vol.shape[:2] + (2,), order="F"), vol[:, :, :2])
  assert np.array_equal(port.decompress(middle).reshape(vol.shape[:2] + (1,), order="F")[:, :, 0], vol[:, :, 2])
  assert np.array_equal(port.decompress(after).reshape(vol.shape[:2] + (sz - 3,), order="F"), vol[:, :, 3:])
  assert crackle_amd.zstack([before, middle, after]) == whole
  b0, m0, a0 = crackle_amd.zsplit(whole, 0)
  assert b0 == b"" and crackle_amd.zstack([m0, a0]) == whole


def test_zstack_of_slab_streams_equals_whole(port):
  vol = _vol("voronoi")
  slabs = [port.compress(np.asfortranarray(vol[:, :, a:b])) for a, b in ((0, 4), (4, 5), (5, 9))]
  assert crackle_amd.zstack(slabs) == port.compress(vol)


def test_zsplit_argument_checks(port):
  whole = port.compress(_vol("noise"))
  with pytest.raises(ValueError):
    crackle_amd.zsplit(whole, 99)
  pins = port.compress(_vol("voronoi"), allow_pins=True)
  if crackle_amd.header(pins).label_format == 2:
    with pytest.raises(ValueError):
      crackle_amd.zshatter(pins)
