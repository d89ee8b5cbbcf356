"""TEST INFRASTRUCTURE ONLY — ctypes loaders for the CPU checkers.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (crackle_amd/) never does.

Two checkers share one calling convention:
  * ``ref``    — oracle/_ref/libcrackle_ref.so: the reference's own header-only C++
                 compiled in place by ``make -C oracle ref`` (binary travels to the
                 GPU box; sources do not).
  * ``port``   — oracle/libckl_oracle.so: this repo's plain-C restatement
                 (oracle/ckl_oracle.c), pinned byte-for-byte against ``ref`` and the
                 committed golden fixtures.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_REF_SO = os.path.join(HERE, "_ref", "libcrackle_ref.so")
_PORT_SO = os.path.join(HERE, "libckl_oracle.so")


def build(which=("oracle", "ref")):
  for target in which:
    subprocess.run(["make", "-s", "-C", HERE, target], check=True)


class _Checker:
  def __init__(self, path, prefix, kind):
    self.path = path
    self.kind = kind
    self.lib = C.CDLL(path)
    self.prefix = prefix
    L = self.lib
    f = getattr(L, prefix + "compress")
    f.restype = C.c_int
    f.argtypes = [
      C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64,
      C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int64, C.c_uint64,
      C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
    ]
    self._compress = f
    f = getattr(L, prefix + "decompress")
    f.restype = C.c_int
    f.argtypes = [
      C.c_char_p, C.c_uint64, C.c_void_p, C.c_int64, C.c_int64, C.c_uint64,
      C.c_int, C.c_uint64,
    ]
    self._decompress = f
    f = getattr(L, prefix + "free")
    f.restype = None
    f.argtypes = [C.c_void_p]
    self._free = f
    f = getattr(L, prefix + "last_error")
    f.restype = C.c_char_p
    self._err = f
    f = getattr(L, prefix + "connected_components")
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64,
                  C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    self._cc = f
    f = getattr(L, prefix + "slice_vcg")
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_uint64, C.c_int64, C.c_void_p]
    self._vcg = f
    f = getattr(L, prefix + "crc32c")
    f.restype = C.c_uint32
    f.argtypes = [C.c_char_p, C.c_uint64]
    self._crc = f
    f = getattr(L, prefix + "voxel_connectivity_graph")
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_uint64, C.c_void_p]
    self._vcg3d = f
    f = getattr(L, prefix + "array_equal")
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_int)]
    self._array_equal = f
    f = getattr(L, prefix + "mode_pooling")
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_uint64, C.c_int64, C.c_int64, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64)]
    self._mode_pooling = f
    f = getattr(L, prefix + "point_cloud")
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_uint64, C.c_int64, C.c_int64, C.c_void_p, C.c_uint64, C.c_int, C.c_int,
                  C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    self._point_cloud = f
    f = getattr(L, prefix + "reencode")
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    self._reencode = f
    f = getattr(L, prefix + "label_stats")
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_int64, C.c_int64, C.c_uint64,
                  C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    self._label_stats = f

  # -- reference-shaped surface (fastcrackle.compress/decompress semantics) --
  def compress(self, labels, allow_pins=False, fortran_order=None, markov_model_order=0,
               optimize_pins=False, auto_bgcolor=True, manual_bgcolor=0, parallel=1):
    """labels: any-order ndarray; converted like crackle/codec.py:723-724."""
    if fortran_order is None:
      fortran_order = bool(labels.flags.f_contiguous)
    labels = np.asfortranarray(labels)
    shape = list(labels.shape) + [1] * (3 - labels.ndim)
    out = C.c_void_p()
    n = C.c_uint64()
    rc = self._compress(
      labels.ctypes.data, labels.dtype.itemsize, int(labels.dtype.kind == "i"),
      shape[0], shape[1], shape[2], int(bool(allow_pins)), int(fortran_order),
      int(markov_model_order), int(optimize_pins), int(auto_bgcolor),
      int(manual_bgcolor), int(parallel), C.byref(out), C.byref(n))
    if rc != 0:
      raise RuntimeError(self._err().decode())
    try:
      return C.string_at(out.value, n.value)
    finally:
      self._free(out)

  def decompress(self, binary, z_start=0, z_end=-1, parallel=1, label=None):
    """Returns the 1-D array fastcrackle.decompress would (x fastest for F streams)."""
    binary = bytes(binary)
    fmt = int.from_bytes(binary[5:7], "little")
    dw = 1 << (fmt & 3)
    sx = int.from_bytes(binary[7:11], "little")
    sy = int.from_bytes(binary[11:15], "little")
    sz = int.from_bytes(binary[15:19], "little")
    zs = max(z_start, 0)
    ze = sz if z_end == -1 else min(max(z_end, 0), sz)
    dtype = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[dw]
    if label is not None:
      dtype = np.uint8
    out = np.zeros(sx * sy * max(ze - zs, 0), dtype=dtype)
    rc = self._decompress(binary, len(binary), out.ctypes.data, z_start, z_end,
                          int(parallel), int(label is not None), int(label or 0))
    if rc != 0:
      raise RuntimeError(self._err().decode())
    return out

  def connected_components(self, labels):
    labels = np.asfortranarray(labels)
    sx, sy, sz = labels.shape
    cc = np.zeros(sx * sy * sz, dtype=np.uint32)
    per = np.zeros(sz, dtype=np.uint64)
    N = C.c_uint64()
    rc = self._cc(labels.ctypes.data, labels.dtype.itemsize, sx, sy, sz,
                  cc.ctypes.data, per.ctypes.data, C.byref(N))
    if rc != 0:
      raise RuntimeError(self._err().decode())
    return cc.reshape((sx, sy, sz), order="F"), per, N.value

  def slice_vcg(self, binary, z):
    binary = bytes(binary)
    sx = int.from_bytes(binary[7:11], "little")
    sy = int.from_bytes(binary[11:15], "little")
    out = np.zeros(sx * sy, dtype=np.uint8)
    rc = self._vcg(binary, len(binary), z, out.ctypes.data)
    if rc != 0:
      raise RuntimeError(self._err().decode())
    return out.reshape((sx, sy), order="F")

  def voxel_connectivity_graph(self, binary, connectivity=6, parallel=1):
    """crackle.voxel_connectivity_graph (operations.py:936-954): uint8 (sx, sy, sz), F order."""
    import numpy as np
    binary = bytes(binary)
    sx, sy, sz = (int.from_bytes(binary[o:o + 4], "little") for o in (7, 11, 15))
    out = np.zeros((sx, sy, sz), dtype=np.uint8, order="F")
    rc = self._vcg3d(binary, len(binary), int(connectivity), int(parallel), out.ctypes.data)
    if rc != 0:
      raise RuntimeError(self._err().decode())
    return out

  def reencode(self, binary, markov_model_order, parallel=1):
    """crackle.reencode (codec.py:877-881 -> crackle.hpp:858-984)."""
    binary = bytes(binary)
    out, n = C.c_void_p(), C.c_uint64()
    rc = self._reencode(binary, len(binary), int(markov_model_order), int(parallel), C.byref(out), C.byref(n))
    if rc != 0:
      raise RuntimeError(self._err().decode())
    try:
      return C.string_at(out.value, n.value)
    finally:
      self._free(out)

  def array_equal(self, binary1, binary2, parallel=1):
    """fastcrackle.array_equal (operations.hpp:1039-1184), without the host-side pre-checks of
    crackle/operations.py:976-992."""
    b1, b2 = bytes(binary1), bytes(binary2)
    eq = C.c_int(0)
    rc = self._array_equal(b1, len(b1), b2, len(b2), int(parallel), C.byref(eq))
    if rc != 0:
      raise RuntimeError(self._err().decode())
    return bool(eq.value)

  def mode_pooling_2x2x1(self, binary, z_start=0, z_end=-1, parallel=1):
    """fastcrackle.mode_pooling_2x2x1 (operations.hpp:1201-1340): list of per-slice streams."""
    binary = bytes(binary)
    sz = int.from_bytes(binary[15:19], "little")
    lens = np.zeros(max(sz, 1), dtype=np.uint64)
    out, n, cnt = C.c_void_p(), C.c_uint64(), C.c_uint64()
    rc = self._mode_pooling(binary, len(binary), int(z_start), int(z_end), int(parallel), C.byref(out), C.byref(n), lens.ctypes.data, C.byref(cnt))
    if rc != 0:
      raise RuntimeError(self._err().decode())
    try:
      blob = C.string_at(out.value, n.value) if n.value else b""
    finally:
      if out.value:
        self._free(out)
    res, at = [], 0
    for m in lens[:cnt.value]:
      res.append(blob[at:at + int(m)])
      at += int(m)
    return res

  def point_cloud(self, binary, z_start=0, z_end=-1, labels=None, skip_background=False):
    """fastcrackle.point_cloud (operations.hpp:183-262, dual_graph.hpp:133-275) with parallel = 1:
    dict label -> flat uint16 array of (x, y, z) triples, in the reference's append order."""
    binary = bytes(binary)
    sel = None if labels is None else np.ascontiguousarray(labels, dtype=np.uint64)
    lab_p, off_p, pts_p, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64()
    rc = self._point_cloud(binary, len(binary), int(z_start), int(z_end), None if sel is None else sel.ctypes.data,
                           0 if sel is None else sel.size, int(sel is not None), int(bool(skip_background)),
                           C.byref(lab_p), C.byref(off_p), C.byref(pts_p), C.byref(n))
    if rc != 0:
      raise RuntimeError(self._err().decode())
    try:
      k = int(n.value)
      if k == 0:
        return {}
      labs = np.frombuffer(C.string_at(lab_p.value, 8 * k), dtype=np.uint64)
      offs = np.frombuffer(C.string_at(off_p.value, 8 * (k + 1)), dtype=np.uint64)
      pts = np.frombuffer(C.string_at(pts_p.value, 6 * int(offs[k])), dtype=np.uint16)
      return {int(labs[i]): pts[3 * int(offs[i]):3 * int(offs[i + 1])].copy() for i in range(k)}
    finally:
      for p in (lab_p, off_p, pts_p):
        if p.value:
          self._free(p)

  def _stats(self, binary, which, z_start, z_end, parallel):
    binary = bytes(binary)
    lab_p, val_p, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
    rc = self._label_stats(binary, len(binary), which, int(z_start), int(z_end), int(parallel), C.byref(lab_p), C.byref(val_p), C.byref(n))
    if rc != 0:
      raise RuntimeError(self._err().decode())
    try:
      k = int(n.value)
      labs = np.frombuffer(C.string_at(lab_p.value, 8 * k), dtype=np.uint64).copy()
      dt, per = ((np.uint64, 1), (np.float64, 3), (np.uint32, 6))[which]
      vals = np.frombuffer(C.string_at(val_p.value, np.dtype(dt).itemsize * per * k), dtype=dt).reshape(k, per).copy()
      return labs, vals
    finally:
      for p in (lab_p, val_p):
        if p.value:
          self._free(p)

  def voxel_counts(self, binary, z_start=0, z_end=-1, parallel=1):
    """fastcrackle.voxel_counts (src/fastcrackle.cpp:346-365, operations.hpp:321-371): dict label -> count."""
    labs, vals = self._stats(binary, 0, z_start, z_end, parallel)
    return {int(l): int(v[0]) for l, v in zip(labs, vals)}

  def centroids(self, binary, z_start=0, z_end=-1, parallel=1):
    """fastcrackle.centroids (src/fastcrackle.cpp:367-392, operations.hpp:421-491): dict label -> float64[3]."""
    labs, vals = self._stats(binary, 1, z_start, z_end, parallel)
    return {int(l): v for l, v in zip(labs, vals)}

  def bounding_boxes(self, binary, z_start=0, z_end=-1, parallel=1):
    """fastcrackle.bounding_boxes (src/fastcrackle.cpp:394-420, operations.hpp:541-617): dict label -> uint32[6]."""
    labs, vals = self._stats(binary, 2, z_start, z_end, parallel)
    return {int(l): v for l, v in zip(labs, vals)}

  def crc32c(self, data):
    data = bytes(data)
    return int(self._crc(data, len(data)))


_cache = {}


def ref():
  """The compiled reference (None if oracle/_ref was never built)."""
  if "ref" not in _cache:
    _cache["ref"] = (
      _Checker(_REF_SO, "ckl_ref_", "reference") if os.path.exists(_REF_SO) else None
    )
  return _cache["ref"]


def port():
  """This repo's plain-C restatement (built on demand)."""
  if "port" not in _cache:
    if not os.path.exists(_PORT_SO):
      build(("oracle",))
    _cache["port"] = _Checker(_PORT_SO, "ckl_oracle_", "port")
  return _cache["port"]


def best():
  """Strongest available checker: the compiled reference, else the restatement."""
  return ref() or port()
