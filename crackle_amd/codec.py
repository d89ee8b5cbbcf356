"""crackle.compress / crackle.decompress on an MI355X.

Mirrors the reference's Python surface for this path
(/root/reference/crackle/codec.py:616-733): same names, arguments, return types and
error behaviour, with one addition — ``device`` selects the HIP device ordinal.
All compute goes through the C-ABI of libcrackle_amd.so; there is no CPU fallback.
"""
import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from .headers import CrackleHeader, FormatError, LabelFormat


def _raise(rc: int):
  msg = _lib.last_error()
  if rc == _lib.CKL_ERR_FORMAT:
    raise FormatError(msg)
  if rc == _lib.CKL_ERR_ARG:
    raise ValueError(msg)
  raise RuntimeError(msg)


def header(binary: bytes) -> CrackleHeader:
  """Decode the header from a Crackle bytestream (codec.py:13-15)."""
  return CrackleHeader.frombytes(binary)


def _label_section(binary: bytes, head: CrackleHeader):
  """(number of stored unique labels, their byte offset) of the label section
  (codec.py:17-50, 82-96)."""
  off = head.header_bytes + head.grid_index_bytes
  if head.label_format != LabelFormat.FLAT:
    off += head.stored_data_width
  n = int.from_bytes(binary[off:off + 8], "little")
  return n, off + 8


def num_labels(binary: bytes) -> int:
  """Number of unique labels in the stream (codec.py:82-96; pin streams count the
  background color too)."""
  head = header(binary)
  if head.voxels() == 0:
    return 0
  n, _ = _label_section(binary, head)
  if head.label_format != LabelFormat.FLAT:
    n += 1
  return n


def labels(binary: bytes) -> np.ndarray:
  """Unique labels of the stream (codec.py:17-50)."""
  head = header(binary)
  if head.voxels() == 0:
    return np.zeros((0,), dtype=head.dtype)
  n, off = _label_section(binary, head)
  uniq = np.frombuffer(binary, dtype=head.stored_dtype, offset=off, count=n)
  if head.label_format != LabelFormat.FLAT:
    base = head.header_bytes + head.grid_index_bytes
    bg = np.frombuffer(binary, dtype=head.stored_dtype, offset=base, count=1)
    uniq = np.sort(np.concatenate((bg, uniq)))
  return uniq.astype(head.dtype, copy=False)


def contains(binary: bytes, label: int) -> bool:
  """codec.py:98-132 (sorted search on the unique label list)."""
  head = header(binary)
  if head.voxels() == 0:
    return False
  u = labels(binary)
  if label < 0 or label > np.iinfo(u.dtype).max:
    return False
  i = np.searchsorted(u, u.dtype.type(label))
  return bool(i < u.size and u[i] == label)


def compress(
  labels: np.ndarray,
  allow_pins: int = 0,
  markov_model_order: int = 0,
  bgcolor: Optional[int] = None,
  parallel: int = 0,
  device: int = 0,
) -> bytes:
  """Compress the 3D labels array into a Crackle bytestream (codec.py:689-733).

  ``parallel`` is accepted for signature compatibility; the device schedules all
  z-slices itself."""
  if np.issubdtype(labels.dtype, np.signedinteger):
    raise TypeError("Signed integer data types are not currently supported.")
  if labels.dtype.kind not in "ub" or labels.dtype.itemsize not in (1, 2, 4, 8):
    raise TypeError(f"Unsupported data type: {labels.dtype}")
  if labels.ndim > 3:
    raise ValueError("labels must have at most 3 dimensions")

  f_order = bool(labels.flags.f_contiguous)
  labels = np.asfortranarray(labels)
  shape = list(labels.shape) + [1] * (3 - labels.ndim)   # fastcrackle.cpp:141-147
  optimize_pins = (allow_pins == 2)
  auto_bgcolor = (bgcolor is None)
  manual_bgcolor = 0 if bgcolor is None else int(bgcolor)

  out = C.c_void_p()
  n = C.c_uint64()
  rc = _lib.lib().ckl_compress(
    labels.ctypes.data, _lib.MEM_HOST, labels.dtype.itemsize, 0,
    shape[0], shape[1], shape[2],
    int(bool(allow_pins)), int(f_order), int(markov_model_order),
    int(optimize_pins), int(auto_bgcolor), manual_bgcolor,
    int(device), C.byref(out), C.byref(n),
  )
  if rc != _lib.CKL_OK:
    _raise(rc)
  try:
    return C.string_at(out.value, n.value)
  finally:
    _lib.lib().ckl_free(out)


def _device_decompress(binary: bytes, z_start: int, z_end: int, label: Optional[int], device: int, head: CrackleHeader) -> np.ndarray:
  """fastcrackle.decompress equivalent (fastcrackle.cpp:40-128): 1-D array."""
  zs = max(z_start, 0)
  ze = head.sz if z_end == -1 else min(max(z_end, 0), head.sz)
  voxels = head.sx * head.sy * max(ze - zs, 0)
  dtype = np.uint8 if label is not None else head.dtype
  arr = np.empty((voxels,), dtype=dtype)
  rc = _lib.lib().ckl_decompress(
    binary, len(binary), arr.ctypes.data, arr.nbytes, _lib.MEM_HOST,
    z_start, z_end, int(label is not None), int(label or 0), int(device),
  )
  if rc != _lib.CKL_OK:
    _raise(rc)
  return arr


def decompress_range(
  binary: bytes,
  z_start: Optional[int],
  z_end: Optional[int],
  parallel: int = 0,
  label: Optional[int] = None,
  device: int = 0,
) -> np.ndarray:
  """Decompress a z-range of a Crackle binary into a numpy array (codec.py:632-687)."""
  binary = bytes(binary)
  head = CrackleHeader.frombytes(binary)
  sx, sy, sz = head.sx, head.sy, head.sz
  z_start = 0 if z_start is None else int(z_start)
  z_end = sz if z_end is None else int(z_end)
  order = "F" if head.fortran_order else "C"
  shape = (sx, sy, z_end - z_start)

  if sx * sy * sz == 0:
    out = np.zeros((0,), dtype=head.dtype)
  elif label is not None and not contains(binary, label):
    out = np.zeros(shape, order=order, dtype=head.dtype)
  elif label is None and num_labels(binary) == 1:
    single = labels(binary)[0]
    out = np.zeros(shape, order=order, dtype=head.dtype) if single == 0 else np.full(shape, single, order=order, dtype=head.dtype)
  else:
    out = _device_decompress(binary, z_start, z_end, label, device, head)

  out = out.reshape(shape, order=order)
  if label is not None:
    return out.view(bool)
  if head.signed:
    out = out.view({1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}[head.data_width])
  return out


def decompress(
  binary: bytes,
  label: Optional[int] = None,
  parallel: int = 0,
  crop: bool = False,
  device: int = 0,
) -> np.ndarray:
  """Decompress a Crackle binary into a numpy array (codec.py:616-630).
  With ``label``, returns the boolean image of that label."""
  if label is None:
    return decompress_range(binary, None, None, parallel, device=device)
  # binary-image mode: the reference restricts the decode to the label's z-range
  # (codec.py:588-614); the full-range decode gives the same image
  img = decompress_range(binary, None, None, parallel, label=label, device=device)
  if not crop:
    return img
  nz = np.flatnonzero(img.any(axis=(0, 1)))
  if nz.size == 0:
    head = CrackleHeader.frombytes(binary)
    return np.zeros([0, 0, 0], dtype=bool, order="F" if head.fortran_order else "C")
  return img[:, :, nz[0]:nz[-1] + 1]


def _label_stats(binary: bytes, device: int, want_boxes: bool):
  """One device pass over the stream's runs: (labels, counts, sums[n,3], boxes[n,6]).
  Label values are the unsigned bit patterns the reference keys its dicts with
  (fastcrackle.cpp:346-420)."""
  binary = bytes(binary)
  head = header(binary)
  L = _lib.lib()
  handle = C.c_void_p()
  rc = L.ckl_decoder_create(binary, len(binary), 0, -1, int(device), C.byref(handle))
  if rc != _lib.CKL_OK:
    _raise(rc)
  try:
    cap = num_labels(binary)
    lab = np.zeros((cap,), dtype=np.uint64)
    cnt = np.zeros((cap,), dtype=np.uint64)
    sums = np.zeros((cap, 3), dtype=np.uint64)
    box = np.zeros((cap, 6), dtype=np.uint32) if want_boxes else None
    n = C.c_uint64()
    rc = L.ckl_decoder_label_stats(
      handle, cap, lab.ctypes.data, cnt.ctypes.data, sums.ctypes.data,
      box.ctypes.data if want_boxes else None, C.byref(n))
    if rc != _lib.CKL_OK:
      _raise(rc)
  finally:
    L.ckl_decoder_destroy(handle)
  n = int(n.value)
  if head.data_width < 8:
    lab &= np.uint64((1 << (8 * head.data_width)) - 1)
  return lab[:n], cnt[:n], sums[:n], (box[:n] if want_boxes else None)


def _require_label(binary: bytes, label: Optional[int]):
  if label is not None and not contains(binary, label):
    raise ValueError(f"Label {label} not contained in image.")


def voxel_counts(binary: bytes, label: Optional[int] = None, parallel: int = 0, device: int = 0):
  """Number of voxels per label (codec.py:949-980, operations.hpp:321-372), computed on the
  device from the decoded runs; the volume is never materialised.  With ``label`` returns
  that label's count (the reference narrows the z-range first; the count is the same)."""
  _require_label(binary, label)
  head = header(binary)
  if head.voxels() == 0:
    vcts = {}
  elif num_labels(binary) == 1:
    vcts = {int(labels(binary)[0]): head.voxels()}
  else:
    lab, cnt, _, _ = _label_stats(binary, device, False)
    vcts = {int(l): int(c) for l, c in zip(lab, cnt) if c}
  if label is not None:
    return vcts[label]
  return vcts


def centroids(binary: bytes, label: Optional[int] = None, parallel: int = 0, device: int = 0):
  """Centroid (mean x, y, z as float64) per label (codec.py:982-1006,
  operations.hpp:422-492: integer coordinate sums divided once)."""
  _require_label(binary, label)
  head = header(binary)
  if head.voxels() == 0:
    out = {}
  else:
    lab, cnt, sums, _ = _label_stats(binary, device, False)
    out = {
      int(l): s.astype(np.float64) / np.float64(c)
      for l, c, s in zip(lab, cnt, sums) if c
    }
  if label is not None:
    return out[label]
  return out


def bounding_boxes(binary: bytes, label: Optional[int] = None, parallel: int = 0, no_slice_conversion: bool = False, device: int = 0):
  """Axis-aligned bounding box per label (codec.py:1008-1067, operations.hpp:541-618):
  uint32 [xmin, ymin, zmin, xmax, ymax, zmax] (inclusive), or slices unless
  ``no_slice_conversion``.  A single-label stream reports [0, 0, 0, sx, sy, sz] like the
  reference (codec.py:1038-1043)."""
  _require_label(binary, label)
  head = header(binary)
  if head.voxels() == 0:
    bbxes = {}
  elif num_labels(binary) == 1:
    bbxes = {int(labels(binary)[0]): np.array([0, 0, 0, head.sx, head.sy, head.sz], dtype=np.uint32)}
  else:
    lab, cnt, _, box = _label_stats(binary, device, True)
    # every label of the unique list has an entry (absent ones keep the initial box); a label outside
    # that list that is absent from the range comes back as a zero box with count 0 and has none
    bbxes = {int(l): b.copy() for l, c, b in zip(lab, cnt, box) if c or b[0]}
  if no_slice_conversion:
    return bbxes[label] if label is not None else bbxes
  if label is not None:
    bbxes = {label: bbxes[label]}
  for lbl, b in bbxes.items():
    bbxes[lbl] = (slice(int(b[0]), int(b[3]) + 1), slice(int(b[1]), int(b[4]) + 1), slice(int(b[2]), int(b[5]) + 1))
  return bbxes[label] if label is not None else bbxes


def reencode(binary: bytes, markov_model_order: int, parallel: int = 0, device: int = 0) -> bytes:
  """The stream with its crack codes stored under another markov model order
  (codec.py:877-881 -> reencode_with_markov_order, crackle.hpp:858-984); the label section is
  untouched.  Runs on the device: crack decoder -> crack planes -> the encoder's trail."""
  binary = bytes(binary)
  head = header(binary)
  if head.markov_model_order == markov_model_order:
    return binary
  out, n = C.c_void_p(), C.c_uint64()
  rc = _lib.lib().ckl_reencode_markov(binary, len(binary), int(markov_model_order), int(device), C.byref(out), C.byref(n))
  if rc != _lib.CKL_OK:
    _raise(rc)
  try:
    return C.string_at(out.value, n.value)
  finally:
    _lib.lib().ckl_free(out)


def voxel_connectivity_graph(binary: bytes, connectivity: int = 6, parallel: int = 0, device: int = 0) -> np.ndarray:
  """The voxel connectivity graph of the image as a uint8 array, shape (sx, sy, sz), F order
  (crackle/operations.py:936-954 -> operations.hpp:667-826).
  bitset (right hand side is LSB): 00-z+z-y+y-x+x"""
  if connectivity not in (4, 6):
    raise ValueError(f"Only 4 and 6 connected are supported. Got: {connectivity}")
  binary = bytes(binary)
  head = header(binary)
  vcg = np.zeros((head.sx, head.sy, head.sz), dtype=np.uint8, order="F")
  rc = _lib.lib().ckl_voxel_connectivity_graph(binary, len(binary), int(connectivity), int(device), vcg.ctypes.data, vcg.nbytes)
  if rc != _lib.CKL_OK:
    _raise(rc)
  return vcg


def crack_crcs(binary: bytes) -> Optional[np.ndarray]:
  """The stored per-slice crc32c of the component images (codec.py:215-223)."""
  head = header(binary)
  if head.format_version == 0:
    return None
  crcl = head.sz * 4
  return np.frombuffer(bytes(binary)[-crcl:] if crcl else b"", dtype=np.uint32)


def structure_equal(binary1: bytes, binary2: bytes, parallel: int = 0, device: int = 0) -> bool:
  """Whether two streams partition the volume the same way, whatever the labels
  (crackle/operations.py:996-1021): same shape, same stored crcs of the component images, same
  4-connected voxel connectivity graph (computed on the device)."""
  h1, h2 = header(binary1), header(binary2)
  if h1.sx != h2.sx or h1.sy != h2.sy or h1.sz != h2.sz:
    return False
  if h1.format_version > 0 and h2.format_version > 0:
    if not np.all(crack_crcs(binary1) == crack_crcs(binary2)):
      return False
  vcg1 = voxel_connectivity_graph(binary1, connectivity=4, parallel=parallel, device=device)
  vcg2 = voxel_connectivity_graph(binary2, connectivity=4, parallel=parallel, device=device)
  return bool(np.all(vcg1 == vcg2))


def labels_crc(binary: bytes) -> Optional[int]:
  """The stored crc32c of the label section (codec.py:205-213)."""
  head = header(binary)
  if head.format_version == 0:
    return None
  crcl = head.sz * 4 + 4
  return int.from_bytes(bytes(binary)[-crcl:len(binary) - crcl + 4], "little")


def check(binary: bytes, device: int = 0) -> dict:
  """Test for file corruption, reporting which sections are damaged (codec.py:900-948): header
  (crc8), crack index (its crc32c and that it stays inside the stream), label section (crc32c)
  and the z of every slice whose crack code does not decode, whose component count disagrees
  with the label section or whose component-image crc32c differs.  The slices are verified on
  the device by the decode pipeline up to the component ids; no volume is produced."""
  binary = bytes(binary)
  sections = {"header": None, "crack_index": None, "labels": None, "z": None}
  try:
    head = header(binary)
  except FormatError:
    sections["header"] = False
    return sections
  sections["header"] = True
  L = _lib.lib()
  hb, gib = head.header_bytes, head.grid_index_bytes
  if len(binary) < hb + gib:
    sections["crack_index"] = False
    return sections
  zidx = np.frombuffer(binary, dtype="<u4", offset=hb, count=head.sz)
  if head.format_version > 0:
    stored = int.from_bytes(binary[hb + 4 * head.sz: hb + 4 * head.sz + 4], "little")
    if stored != int(L.ckl_crc32c(binary[hb:hb + 4 * head.sz], 4 * head.sz)):
      sections["crack_index"] = False
      return sections
  # the reference's offsets leave the markov model out here (codec.py:277-281); kept
  if hb + gib + head.num_label_bytes + int(zidx.astype(np.uint64).sum()) >= len(binary):
    sections["crack_index"] = False
    return sections
  sections["crack_index"] = True
  if head.format_version == 0:
    return sections
  lab = binary[hb + gib: hb + gib + head.num_label_bytes]
  sections["labels"] = labels_crc(binary) == int(L.ckl_crc32c(lab, len(lab)))
  sections["z"] = []
  if head.voxels() == 0:
    return sections
  handle = C.c_void_p()
  rc = L.ckl_decoder_create(binary, len(binary), 0, -1, int(device), C.byref(handle))
  if rc != _lib.CKL_OK:
    # the stream's layout cannot even be parsed: every slice is unreadable
    sections["z"] = list(range(head.sz))
    return sections
  try:
    errs = np.zeros(head.sz, dtype=np.uint32)
    rc = L.ckl_decoder_check(handle, errs.ctypes.data, errs.size)
    if rc != _lib.CKL_OK:
      _raise(rc)
  finally:
    L.ckl_decoder_destroy(handle)
  sections["z"] = [int(z) for z in np.flatnonzero(errs)]
  return sections


def ok(binary: bytes, device: int = 0) -> bool:
  """Whether the stream is intact as a whole (codec.py:883-898)."""
  report = check(binary, device=device)
  if report["header"] is False or report["crack_index"] is False or report["labels"] is False:
    return False
  return not (report["z"] is not None and len(report["z"]) > 0)
