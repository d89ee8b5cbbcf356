"""Stream surgery on .ckl bytes without decoding — the slab merge the sharded encoder is built
on, exposed with the reference's names (crackle/operations.py:424-662: zstack, zsplit,
zshatter).  Host only (native: ckl_zstack / ckl_zsplit); FLAT label streams."""
import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib
from .codec import contains, header, labels, num_labels


def _take(out: C.c_void_p, n: C.c_uint64) -> bytes:
  try:
    return C.string_at(out.value, n.value)
  finally:
    _lib.lib().ckl_free(out)


def zstack(images: Sequence[bytes]) -> bytes:
  """Concatenates the streams of consecutive z-slabs into the stream of the whole volume
  (crackle/operations.py:424-548); byte-identical to compress() of the whole when the slabs
  agree on the crack format (automated_test.py:468-487)."""
  bufs = [bytes(b) for b in images]
  if not bufs:
    raise ValueError("zstack needs at least one stream")
  # The reference brings every input to markov order 0 first (operations.py:447-457).  Slabs
  # that share one model (the sharded encoder's, or the parts of one stream) are merged as they
  # are; otherwise the codes are re-encoded to order 0 on the device, like the reference.
  heads = [header(b) for b in bufs]
  def _model(b, h):
    off = h.header_bytes + h.grid_index_bytes + h.num_label_bytes
    return b[off:off + h.markov_model_bytes]
  orders = {h.markov_model_order for h in heads}
  if orders != {0} and (len(orders) > 1 or len({_model(b, h) for b, h in zip(bufs, heads)}) > 1):
    from .codec import reencode
    bufs = [reencode(b, 0) for b in bufs]
  L = _lib.lib()
  arr = (C.c_char_p * len(bufs))(*bufs)
  lens = (C.c_uint64 * len(bufs))(*[len(b) for b in bufs])
  out, n = C.c_void_p(), C.c_uint64()
  if L.ckl_zstack(arr, lens, len(bufs), C.byref(out), C.byref(n)) != _lib.CKL_OK:
    raise ValueError(_lib.last_error())
  return _take(out, n)


def _zrange(binary: bytes, z_start: int, z_end: int) -> bytes:
  out, n = C.c_void_p(), C.c_uint64()
  if _lib.lib().ckl_zsplit(binary, len(binary), z_start, z_end, C.byref(out), C.byref(n)) != _lib.CKL_OK:
    raise ValueError(_lib.last_error())
  return _take(out, n)


def zsplit(binary: bytes, z: int) -> Tuple[bytes, bytes, bytes]:
  """(before, middle, after) streams around slice z (crackle/operations.py:626-647); empty
  ranges come back as b''."""
  head = header(binary)
  if z < 0 or z >= head.sz:
    raise ValueError(f"{z} is outside the range 0 to {head.sz}.")
  before = _zrange(binary, 0, z) if z > 0 else b""
  middle = _zrange(binary, z, z + 1)
  after = _zrange(binary, z + 1, head.sz) if z + 1 < head.sz else b""
  return before, middle, after


def zshatter(binary: bytes) -> List[bytes]:
  """One stream per z-slice (crackle/operations.py:649-662)."""
  head = header(binary)
  return [_zrange(binary, z, z + 1) for z in range(head.sz)]


def array_equal(binary1: bytes, binary2: bytes, parallel: int = 0, device: int = 0) -> bool:
  """Do the two streams hold the same array, whatever their encoding (crackle/operations.py:966-994:
  shapes, number of labels and label sets are compared on the host, then
  fastcrackle.array_equal = src/operations.hpp:1039-1184 -> ckl_array_equal)."""
  b1, b2 = bytes(binary1), bytes(binary2)
  h1, h2 = header(b1), header(b2)
  if h1.sx != h2.sx or h1.sy != h2.sy or h1.sz != h2.sz:
    return False
  if num_labels(b1) != num_labels(b2):
    return False
  if np.any(labels(b1) != labels(b2)):
    return False
  eq = C.c_int(0)
  if _lib.lib().ckl_array_equal(b1, len(b1), b2, len(b2), int(device), C.byref(eq)) != _lib.CKL_OK:
    raise RuntimeError(_lib.last_error())
  return bool(eq.value)


def _mode_pooling_slices(binary: bytes, z_start: int = 0, z_end: int = -1, device: int = 0) -> List[bytes]:
  """fastcrackle.mode_pooling_2x2x1 (src/fastcrackle.cpp:620-639): one pooled stream per slice."""
  b = bytes(binary)
  out, n, lens, cnt = C.c_void_p(), C.c_uint64(), C.c_void_p(), C.c_uint64()
  L = _lib.lib()
  if L.ckl_mode_pooling_2x2x1(b, len(b), int(z_start), int(z_end), int(device), C.byref(out), C.byref(n), C.byref(lens), C.byref(cnt)) != _lib.CKL_OK:
    raise RuntimeError(_lib.last_error())
  try:
    sizes = list((C.c_uint64 * cnt.value).from_address(lens.value)) if cnt.value else []
    blob = C.string_at(out.value, n.value) if n.value else b""
  finally:
    if out.value:
      L.ckl_free(out)
    if lens.value:
      L.ckl_free(lens)
  res, at = [], 0
  for m in sizes:
    res.append(blob[at:at + m])
    at += m
  return res


def mode_pooling_2x2x1(binary: bytes, parallel: int = 0, device: int = 0) -> bytes:
  """Downsamples a segmentation 2 x 2 x 1 by the reference's pooling rule
  (crackle/operations.py:1023-1026: the per-slice streams of fastcrackle.mode_pooling_2x2x1, stacked)."""
  return zstack(_mode_pooling_slices(binary, 0, -1, device))


def _point_cloud_raw(binary: bytes, z_start: int, z_end: int, label_list, skip_background: bool, device: int) -> Dict[int, np.ndarray]:
  """fastcrackle.point_cloud (src/fastcrackle.cpp:315-345 -> ckl_point_cloud): dict label -> flat
  uint16 array of (x, y, z) triples."""
  b = bytes(binary)
  sel = None if label_list is None else np.ascontiguousarray(label_list, dtype=np.uint64)
  lab_p, off_p, pts_p, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64()
  L = _lib.lib()
  rc = L.ckl_point_cloud(b, len(b), int(z_start), int(z_end), None if sel is None else sel.ctypes.data,
                         0 if sel is None else sel.size, int(sel is not None), int(bool(skip_background)), int(device),
                         C.byref(lab_p), C.byref(off_p), C.byref(pts_p), C.byref(n))
  if rc != _lib.CKL_OK:
    raise RuntimeError(_lib.last_error())
  try:
    k = int(n.value)
    if k == 0:
      return {}
    labs = np.ctypeslib.as_array(C.cast(lab_p, C.POINTER(C.c_uint64)), shape=(k,)).tolist()
    offs = (3 * np.ctypeslib.as_array(C.cast(off_p, C.POINTER(C.c_uint64)), shape=(k + 1,))).tolist()
    # one copy out of the library's buffer; the labels' arrays are views of it
    pts = np.ctypeslib.as_array(C.cast(pts_p, C.POINTER(C.c_uint16)), shape=(offs[k],)).copy() if offs[k] else np.zeros(0, np.uint16)
    return {labs[i]: pts[offs[i]:offs[i + 1]] for i in range(k)}
  finally:
    for p in (lab_p, off_p, pts_p):
      if p.value:
        L.ckl_free(p)


def point_cloud(
  binary: bytes, label: Optional[Union[int, List[int]]] = None, parallel: int = 0,
  z_start: int = -1, z_end: int = -1, skip_background: bool = True, device: int = 0,
) -> Union[np.ndarray, Dict[int, np.ndarray]]:
  """Surface point clouds of the labels without decompressing the image (crackle/codec.py:804-872).

  Without `label`: dict label -> (N, 3) uint16 array of (x, y, z).  With an int: that label's
  array; with a list: the dict restricted to those labels.  A label the image does not contain
  raises ValueError.  The reference narrows an unspecified z-range to the slices that hold the
  labels (z_range_for_label); slices without them contribute no points, so the whole range is
  traced here instead.  Points come in the order the reference gives with parallel = 1."""
  scalar_input = False
  if isinstance(label, (int, np.integer)):
    scalar_input = True
    label = [int(label)]
  head = header(binary)
  if isinstance(label, (list, tuple)):
    for lbl in label:
      if not contains(binary, lbl):
        raise ValueError(f"Label {lbl} not contained in image.")
  if z_start == -1:
    z_start = 0
  if z_end == -1:
    z_end = head.sz
  ptc = _point_cloud_raw(binary, z_start, z_end, label, skip_background, device)
  if len(ptc) == 0:
    if label:
      return np.zeros([0, 3], dtype=np.uint16, order="C")
    return {}
  for lbl, pts in ptc.items():
    ptc[lbl] = np.asarray(pts, dtype=np.uint16, order="C").reshape([len(pts) // 3, 3], order="C")
  if scalar_input:
    return ptc[label[0]]
  return ptc
