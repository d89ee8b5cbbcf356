"""TEST INFRASTRUCTURE: a compute backend for crackle_amd.distributed.ShardedCodec that
runs the per-slab stages on the CPU oracle (oracle/ckl_oracle.c), so the sharding /
collective / merge orchestration can be exercised under gloo without a GPU.
Volumes are F-ordered numpy arrays (sx, sy, sz_local)."""
import ctypes as C

import numpy as np

from oracle import oracle


class OracleDecodeSession:
  def __init__(self, port, binary, z_start, z_end):
    self.port, self.binary, self.z_start, self.z_end = port, binary, z_start, z_end

  def run(self, out: np.ndarray, label=None):
    flat = self.port.decompress(self.binary, self.z_start, self.z_end, parallel=2, label=label)
    out[...] = flat.reshape(out.shape, order="F")

  def timing(self):
    return 0.0, 0.0

  def close(self):
    pass


class OracleBackend:
  def __init__(self):
    self.port = oracle.port()
    L = self.port.lib
    L.ckl_oracle_compress_ex.restype = C.c_int
    L.ckl_oracle_compress_ex.argtypes = [
      C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64,
      C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int64, C.c_uint64,
      C.c_int, C.c_int, C.c_int, C.c_void_p,
      C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.ckl_oracle_stats.restype = C.c_int
    L.ckl_oracle_stats.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64] + [C.POINTER(C.c_uint64)] * 4
    L.ckl_oracle_markov_hist.restype = C.c_int
    L.ckl_oracle_markov_hist.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_uint64, C.c_void_p]
    self.L = L

  def stats(self, vol, shape):
    vol = np.asfortranarray(vol)
    mx, pairs, first, last = (C.c_uint64() for _ in range(4))
    rc = self.L.ckl_oracle_stats(vol.ctypes.data, vol.dtype.itemsize, *shape, C.byref(mx), C.byref(pairs), C.byref(first), C.byref(last))
    assert rc == 0
    return mx.value, pairs.value, first.value, last.value

  def markov_hist(self, vol, shape, crack_format, order):
    vol = np.asfortranarray(vol)
    hist = np.zeros((4 ** order) * 4, dtype=np.uint32)
    rc = self.L.ckl_oracle_markov_hist(vol.ctypes.data, vol.dtype.itemsize, *shape, int(crack_format), int(order), hist.ctypes.data)
    assert rc == 0
    return hist

  def encode(self, vol, shape, allow_pins=False, fortran_order=True, markov_model_order=0, overrides=None):
    vol = np.asfortranarray(vol)
    ov = overrides or {}
    model = ov.get("model")
    keep = np.ascontiguousarray(model, dtype=np.uint8) if model is not None else None
    out, n = C.c_void_p(), C.c_uint64()
    rc = self.L.ckl_oracle_compress_ex(
      vol.ctypes.data, vol.dtype.itemsize, 0, *shape,
      int(bool(allow_pins)), int(fortran_order), int(markov_model_order), 0, 1, 0, 2,
      int(ov.get("crack_format", -1)), int(ov.get("label_format", -1)), int(ov.get("stored_width", 0)),
      keep.ctypes.data if keep is not None else None,
      C.byref(out), C.byref(n))
    if rc != 0:
      raise RuntimeError(self.port._err().decode())
    try:
      return C.string_at(out.value, n.value)
    finally:
      self.port._free(out)

  def itemsize(self, vol) -> int:
    return int(np.asarray(vol).dtype.itemsize)

  def volume_to_host(self, vol, out: np.ndarray):
    out[...] = np.asfortranarray(vol).ravel(order="F").view(out.dtype)

  def components(self, vol, shape, id_base, cc_out, ncomp_out):
    cc, per, _ = self.port.connected_components(np.asfortranarray(vol))
    cc_out[...] = cc.ravel(order="F").astype(np.uint32) + np.uint32(id_base)
    ncomp_out[...] = per.astype(np.uint32)

  def open_decoder(self, binary, z_start, z_end):
    return OracleDecodeSession(self.port, binary, z_start, z_end)

  def encoder_timing(self):
    return 0.0, 0.0
