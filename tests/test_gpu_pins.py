"""GPU tests of the pin label encoder's device passes (crackle_amd/csrc/ckl_pins_dev.hpp:
k_pin_dedup, k_pin_columns, k_pin_choice, k_pin_extent, k_pin_ids) against the oracle's
pins::compute + encode_condensed_pins (src/pins.hpp:95-403, src/labels.hpp:157-344).

The volumes are built to reach every branch of add_pin's dedup (a run covered by the previous
column's pin, a run covering it, partial overlap, a label returning within one column, a dropped
run followed by another run of the same label) and the quirks of the cover (the deepest-pin
search that never updates its maximum, ties in the background colour)."""
import numpy as np
import pytest

import crackle_amd
from crackle_amd import synth
from util import label_format

pytestmark = pytest.mark.gpu


def _blocks(shape, block, high, seed, dtype=np.uint32, offset=0):
  """Noise over blocks: few labels, so labels return along z within a column."""
  rng = np.random.default_rng(seed)
  small = tuple(-(-s // b) for s, b in zip(shape, block))
  v = rng.integers(0, high, size=small, dtype=np.int64)
  for ax, b in enumerate(block):
    v = np.repeat(v, b, axis=ax)
  v = v[:shape[0], :shape[1], :shape[2]].astype(dtype) + dtype(offset)
  return np.asfortranarray(v)


def _staircase(sx, sy, sz, dtype=np.uint16):
  """One label whose z-range grows, shrinks and shifts from column to column over a second label."""
  v = np.ones((sx, sy, sz), dtype=dtype)
  rng = np.random.default_rng(5)
  for y in range(sy):
    lo, hi = sz // 3, 2 * sz // 3
    for x in range(sx):
      step = rng.integers(0, 5)
      if step == 0: lo = max(0, lo - 1)                     # covers the neighbour: replaces it
      elif step == 1: lo = min(hi, lo + 1)                  # covered by the neighbour: dropped
      elif step == 2: lo, hi = min(lo + 1, sz - 1), min(hi + 1, sz - 1)      # shifted: appended
      elif step == 3: hi = min(sz - 1, hi + 1)
      lo = min(lo, hi)
      v[x, y, lo:hi + 1] = 7
      if x % 11 == 3 and hi + 2 < sz:
        v[x, y, hi + 2:] = 7                                # the label returns higher up in the same column
  return np.asfortranarray(v)


CASES = {
  "blocks_3labels": lambda: _blocks((48, 40, 19), (3, 2, 2), 3, 1),
  "blocks_2labels_thin": lambda: _blocks((37, 21, 33), (1, 1, 1), 2, 2, np.uint8),
  "blocks_u64": lambda: _blocks((40, 24, 12), (4, 3, 2), 5, 3, np.uint64, offset=(1 << 63) + 11),
  "blocks_many_labels": lambda: _blocks((40, 32, 9), (2, 2, 1), 400, 4),
  "staircase": lambda: _staircase(64, 9, 21),
  "staircase_tall": lambda: _staircase(33, 5, 70),
  "one_label": lambda: np.asfortranarray(np.full((20, 10, 6), 9, np.uint8)),
  "two_slices": lambda: _blocks((50, 30, 2), (5, 5, 1), 4, 6),
  "one_row": lambda: _blocks((300, 1, 8), (7, 1, 3), 3, 7),
  "one_column": lambda: _blocks((1, 1, 40), (1, 1, 3), 3, 8),
  "one_x": lambda: _blocks((1, 64, 12), (1, 5, 2), 3, 9),
  "tall_k4": lambda: _blocks((9, 7, 200), (2, 2, 3), 3, 12),             # label registers per lane: 4, 8, 16
  "tall_k8": lambda: _blocks((6, 5, 400), (2, 1, 2), 3, 13, np.uint64, offset=1 << 50),
  "tall_k8_groups": lambda: _blocks((8, 5, 400), (2, 1, 2), 3, 16, np.uint64, offset=1 << 50),      # sx a multiple of 4: four columns per load
  "wide_groups_u8": lambda: _blocks((64, 9, 70), (3, 2, 4), 5, 17, np.uint8),
  "tall_k16": lambda: _blocks((6, 6, 900), (2, 2, 5), 4, 14, np.uint8),
  "tall_thread_rows": lambda: _blocks((4, 3, 1100), (2, 1, 3), 3, 15),    # above 1024 slices: the thread-per-row kernel
  "voronoi_deep": lambda: synth.as_numpy_f(synth.voronoi_labels((96, 64, 80), np.uint32, seed=10, cell=(24, 24, 12))),
  "voronoi_modulus": lambda: synth.as_numpy_f(synth.voronoi_labels((128, 96, 24), np.uint16, seed=11, cell=(12, 12, 5), modulus=6)),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_pin_section_bytes_equal_oracle(name, checker):
  arr = CASES[name]()
  for kw in (dict(), dict(auto_bgcolor=False, manual_bgcolor=1)):
    want = checker.compress(arr, allow_pins=True, **kw)
    akw = dict(bgcolor=kw["manual_bgcolor"]) if kw else {}
    got = crackle_amd.compress(arr, allow_pins=1, **akw)
    assert label_format(want) == 2
    assert got == want, (name, kw)
  assert np.array_equal(crackle_amd.decompress(got), arr)


def test_pin_passes_reused_session_and_c_order(checker):
  """The encoder session keeps its pin buffers between volumes of different sizes; C order goes
  through the same passes after the transpose."""
  a = CASES["blocks_3labels"]()
  b = CASES["voronoi_modulus"]()
  for arr in (a, b, a):
    assert crackle_amd.compress(arr, allow_pins=1) == checker.compress(arr, allow_pins=True)
  c = np.ascontiguousarray(a)
  assert crackle_amd.compress(c, allow_pins=1) == checker.compress(c, allow_pins=True)


@pytest.mark.parametrize("name", ["blocks_3labels", "blocks_u64", "staircase", "staircase_tall", "one_column", "voronoi_modulus"])
def test_thread_per_row_dedup_equals_oracle(name, checker, monkeypatch):
  """k_pin_dedup (label tables in global memory, used above 1024 slices) on the ordinary cases."""
  monkeypatch.setenv("CKL_PINS_ROW_THREADS", "1")
  arr = CASES[name]()
  assert crackle_amd.compress(arr, allow_pins=1) == checker.compress(arr, allow_pins=True)


@pytest.mark.parametrize("name", ["voronoi_deep", "voronoi_modulus", "tall_k8"])
def test_dedup_one_column_per_load_equals_oracle(name, checker, monkeypatch):
  """k_pin_dedup_wave fetches four columns per load when sx is a multiple of four; the one-column form
  on the same volumes (and the host-side bookkeeping of the chosen runs instead of the device's)."""
  monkeypatch.setenv("CKL_PINS_COLUMN_LOADS", "1")
  arr = CASES[name]()
  want = checker.compress(arr, allow_pins=True)
  assert crackle_amd.compress(arr, allow_pins=1) == want
  monkeypatch.setenv("CKL_PINS_HOST_BOOKKEEPING", "1")
  assert crackle_amd.compress(arr, allow_pins=1) == want


@pytest.mark.parametrize("name", sorted(CASES))
def test_distinct_pin_path_equals_oracle(name, checker, monkeypatch):
  """The chosen runs reduced to the distinct ones before their ids are gathered (what volumes whose
  per-component id lists would pass the budget take: CKL_PIN_IDS_BUDGET=0 forces it)."""
  monkeypatch.setenv("CKL_PIN_IDS_BUDGET", "0")
  arr = CASES[name]()
  assert crackle_amd.compress(arr, allow_pins=1) == checker.compress(arr, allow_pins=True), name


def test_whole_volume_pin_stage_on_device(checker):
  """ckl_encoder_components_device + ckl_encoder_pin_labels (the sharded encoder's whole-volume
  stage on rank 0): ids of two slabs painted into one device volume, the section computed from
  device-resident labels and ids equals the label section of the oracle's stream; ids out of
  range are refused."""
  import ctypes as C
  import torch
  from crackle_amd import _lib
  from crackle_amd.headers import CrackleHeader
  L = _lib.lib()
  dev = torch.device("cuda:0")
  for name in ("blocks_3labels", "voronoi_modulus", "blocks_u64"):
    arr = CASES[name]()
    sx, sy, sz = arr.shape
    signed = {1: np.uint8, 2: np.int16, 4: np.int32, 8: np.int64}[arr.dtype.itemsize]
    vol = torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 1, 0)).view(signed)).to(dev)
    half = sz // 2
    cc = torch.zeros(sx * sy * sz, dtype=torch.int32, device=dev)
    nc = np.zeros(sz, dtype=np.uint32)
    base = 0
    for z0, z1 in ((0, half), (half, sz)):
      enc = C.c_void_p()
      assert L.ckl_encoder_create(sx, sy, z1 - z0, arr.dtype.itemsize, 0, C.byref(enc)) == 0
      slab = vol[z0:z1].contiguous()
      part = np.zeros(z1 - z0, dtype=np.uint32)
      torch.cuda.synchronize()
      assert L.ckl_encoder_components_device(enc, slab.data_ptr(), sx, sy, z1 - z0, base, cc[sx * sy * z0:].data_ptr(), part.ctypes.data) == 0, _lib.last_error()
      nc[z0:z1] = part
      base += int(part.sum())
      if z1 != sz:
        L.ckl_encoder_destroy(enc)
    want = checker.compress(arr, allow_pins=True)
    out, n = C.c_void_p(), C.c_uint64()
    h = CrackleHeader.frombytes(want)
    sw = int(h.stored_data_width)
    first = 29 + 4 * (sz + 1)
    want_section = want[first:first + h.num_label_bytes]
    assert L.ckl_encoder_pin_labels(enc, vol.data_ptr(), cc.data_ptr(), sx, sy, sz, nc.ctypes.data, sw, 1, 0, C.byref(out), C.byref(n)) == 0, _lib.last_error()
    got = C.string_at(out.value, n.value)
    L.ckl_free(out)
    assert got == want_section, name
    cc[7] = base + 5      # an id no slice accounts for
    assert L.ckl_encoder_pin_labels(enc, vol.data_ptr(), cc.data_ptr(), sx, sy, sz, nc.ctypes.data, sw, 1, 0, C.byref(out), C.byref(n)) != 0
    L.ckl_encoder_destroy(enc)
