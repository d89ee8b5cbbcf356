"""Host stage of the pin label encoder (ckl_pin_labels_host) against the oracle: given the
labels and the global component ids, the pin section must equal the label section of the
stream the reference algorithm writes.  Runs without a GPU: this is the order-sensitive
host logic (robin-hood slot order, libstdc++ unordered_map order, greedy cover)."""
import ctypes as C

import numpy as np
import pytest

from crackle_amd import _lib
from crackle_amd.headers import CrackleHeader


def _pin_section(L, checker, arr, auto=True, bg=0):
  arr = np.asfortranarray(arr)
  cc, per, _ = checker.connected_components(arr)
  sx, sy, sz = arr.shape
  ccf = np.ascontiguousarray(cc.ravel(order="F").astype(np.uint32))
  nc = np.ascontiguousarray(per.astype(np.uint32))
  mx = int(arr.max())
  sw = 1 if mx < 2**8 else 2 if mx < 2**16 else 4 if mx < 2**32 else 8
  lab = np.ascontiguousarray(arr.ravel(order="F"))
  out, n = C.c_void_p(), C.c_uint64()
  rc = L.ckl_pin_labels_host(lab.ctypes.data, arr.dtype.itemsize, ccf.ctypes.data, sx, sy, sz,
                             nc.ctypes.data, sw, int(auto), bg, C.byref(out), C.byref(n))
  assert rc == 0, _lib.last_error()
  b = C.string_at(out.value, n.value)
  L.ckl_free(out)
  return b


def _want_section(checker, arr, **kw):
  b = checker.compress(arr, allow_pins=True, **kw)
  h = CrackleHeader.frombytes(b)
  if int(h.label_format) != 2:
    return None
  off = 29 + 4 * (arr.shape[2] + 1)
  return b[off:off + h.num_label_bytes]


def test_pin_section_matches_oracle(checker):
  L = _lib.lib()
  rng = np.random.default_rng(0)
  n_ok = 0
  for t in range(48):
    sx, sy, sz = int(rng.integers(3, 40)), int(rng.integers(3, 40)), int(rng.integers(2, 20))
    k = int(rng.integers(2, 8))
    small = rng.integers(0, rng.integers(2, 300), size=((sx + k - 1) // k, (sy + k - 1) // k, (sz + 1) // 2 + 1))
    arr = np.kron(small, np.ones((k, k, 2), dtype=np.int64))[:sx, :sy, :sz]
    dt = [np.uint8, np.uint16, np.uint32, np.uint64][t % 4]
    arr = np.asfortranarray((arr % 256 if dt == np.uint8 else arr).astype(dt))
    want = _want_section(checker, arr)
    if want is None:
      continue
    assert _pin_section(L, checker, arr) == want, (t, arr.shape)
    bg = int(rng.integers(0, 5))
    assert _pin_section(L, checker, arr, False, bg) == _want_section(checker, arr, auto_bgcolor=False, manual_bgcolor=bg), ("manual", t)
    n_ok += 1
  assert n_ok >= 30


def test_pin_section_ties_and_many_labels(checker):
  """Many labels with equal pin counts: background colour ties follow the unordered_map
  iteration order; > 128 labels forces rehashes of both containers."""
  L = _lib.lib()
  sx, sy, sz = 40, 36, 6
  x, y, z = np.meshgrid(np.arange(sx), np.arange(sy), np.arange(sz), indexing="ij")
  arr = np.asfortranarray(((x // 2) + (sx // 2) * (y // 2) + 1000 * (z // 3)).astype(np.uint32))
  want = _want_section(checker, arr)
  assert want is not None
  assert _pin_section(L, checker, arr) == want


def test_pin_section_tens_of_thousands_of_labels(checker):
  """45 k labels: the host stage sorts in pieces on its worker threads and writes the labels' records
  side by side at precomputed places; with and without the unordered_map of the background colour."""
  import os
  L = _lib.lib()
  rng = np.random.default_rng(5)
  small = rng.integers(1, 2**31, size=(150, 150, 2), dtype=np.int64)
  arr = np.asfortranarray(np.kron(small, np.ones((2, 2, 3), dtype=np.int64)).astype(np.uint32))
  arr[:, :, 0] = 7      # one label with many pins: a background colour without a tie
  want = _want_section(checker, arr)
  assert want is not None
  assert _pin_section(L, checker, arr) == want
  os.environ["CKL_PINS_BGCOLOR_MAP"] = "1"
  try:
    assert _pin_section(L, checker, arr) == want
  finally:
    del os.environ["CKL_PINS_BGCOLOR_MAP"]
  os.environ["CKL_PINS_THREADS"] = "1"
  try:
    assert _pin_section(L, checker, arr) == want
  finally:
    del os.environ["CKL_PINS_THREADS"]


def test_pin_host_argument_checks():
  L = _lib.lib()
  out, n = C.c_void_p(), C.c_uint64()
  assert L.ckl_pin_labels_host(None, 1, None, 1, 1, 1, None, 1, 1, 0, C.byref(out), C.byref(n)) != 0
  a = np.zeros(4, np.uint8)
  c = np.zeros(4, np.uint32)
  nc = np.ones(1, np.uint32)
  assert L.ckl_pin_labels_host(a.ctypes.data, 3, c.ctypes.data, 2, 2, 1, nc.ctypes.data, 1, 1, 0, C.byref(out), C.byref(n)) != 0
  assert "dtype" in _lib.last_error()
