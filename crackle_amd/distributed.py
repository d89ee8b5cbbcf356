"""Z-slab sharded encode/decode: one process per GPU, `torch.distributed` over
RCCL/xGMI (backend "nccl") on a node, or gloo in CPU tests of the orchestration.

Slices of one volume are independent on this path (reference: one thread-pool task
per z-slice, src/crackcodes.hpp:510-518, src/labels.hpp:56-88, src/crackle.hpp:584-660),
so rank r owns slices [r*sz, (r+1)*sz).  What is *not* per-slice in the reference are
three whole-volume decisions, each an all-reduce / all-gather of a few
bytes — latency-bound, nothing bulky ever crosses xGMI except the compressed slabs
(~1 % of the input) gathered to the rank that writes the file:

  1. pixel_pairs + max_label      -> crack format, stored width   (src/lib.hpp:224-256, src/crackle.hpp:48-64, 233-235)
  2. order-N context histogram    -> one markov model for all     (src/markov.hpp:193-266)
  3. per-slab streams -> rank 0, merged like crackle.operations.zstack
     (crackle/operations.py:424-548; native: ckl_zstack)

Decode needs no communication: every rank decodes its own z-range of the stream.

The compute backend is injected.  The product backend is HipBackend (C-ABI of
libcrackle_amd.so, device-resident volumes); tests/ substitutes the CPU oracle to
exercise this orchestration under gloo.
"""
import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import _lib

FLAT = 0
IMPERMISSIBLE, PERMISSIBLE = 0, 1


def _byte_width(x: int) -> int:
  return 1 if x <= 0xFF else 2 if x <= 0xFFFF else 4 if x <= 0xFFFFFFFF else 8


def stats_to_model(hist: np.ndarray) -> np.ndarray:
  """markov::stats_to_model (src/markov.hpp:222-266): per context row, symbol -> rank by
  count descending, ties towards the larger symbol (libstdc++ insertion
  sort under the reference's `>=` comparator)."""
  hist = np.asarray(hist, dtype=np.uint64).reshape(-1, 4)
  # sort key: (-count, -symbol)
  order = np.lexsort((-np.arange(4)[None, :].repeat(hist.shape[0], 0), -hist.astype(np.int64)), axis=1)
  model = np.empty_like(order, dtype=np.uint8)
  rows = np.arange(hist.shape[0])[:, None]
  model[rows, order] = np.arange(4, dtype=np.uint8)[None, :]
  return np.ascontiguousarray(model)


def zstack(bufs: Sequence[bytes]) -> bytes:
  """ckl_zstack: host-only merge of FLAT slab streams."""
  L = _lib.lib()
  n = len(bufs)
  arr = (C.c_char_p * n)(*bufs)
  lens = (C.c_uint64 * n)(*[len(b) for b in bufs])
  out, m = C.c_void_p(), C.c_uint64()
  rc = L.ckl_zstack(arr, lens, n, C.byref(out), C.byref(m))
  if rc != _lib.CKL_OK:
    raise RuntimeError(_lib.last_error())
  try:
    return C.string_at(out.value, m.value)
  finally:
    L.ckl_free(out)



_W2DT = {1: "u1", 2: "<u2", 4: "<u4", 8: "<u8"}


def _unpack(raw: np.ndarray, width: int, n: int) -> np.ndarray:
  """n little-endian integers of `width` bytes -> int64 array."""
  return np.frombuffer(raw, dtype=_W2DT[width], count=n).astype(np.int64)


def _pack(values: np.ndarray, width: int) -> np.ndarray:
  return np.ascontiguousarray(values.astype(_W2DT[width])).view(np.uint8).reshape(-1)


def _crc8(data: bytes) -> int:
  """crc.hpp:23-37: reflected, poly 0xe7, init 0xFF."""
  crc = 0xFF
  for b in data:
    crc ^= b
    for _ in range(8):
      crc = ((crc >> 1) ^ 0xE7) if (crc & 1) else (crc >> 1)
  return crc


class _SlabSections:
  """Views of the sections of one FLAT .ckl stream (header.hpp:284-297, labels.hpp:123-152)."""

  def __init__(self, stream):
    if isinstance(stream, _lib.HostStream):
      a = np.frombuffer(stream.view(), dtype=np.uint8)
    else:
      a = np.frombuffer(stream, dtype=np.uint8)
    info = _lib.HeaderInfo()
    head = bytes(a[:64])
    if _lib.lib().ckl_header_info_from_bytes(head, len(head), info) != _lib.CKL_OK:
      raise RuntimeError(_lib.last_error())
    if info.label_format != FLAT or info.format_version != 1:
      raise RuntimeError("sharded merge needs version-1 FLAT slab streams")
    self.header = head[:29]
    sx, sy, sz = int(info.sx), int(info.sy), int(info.sz)
    self.sz = sz
    self.stored_width = int(info.stored_data_width)
    self.comp_width = _byte_width(sx * sy)
    hb = 29
    self.zidx = a[hb:hb + 4 * sz]
    off = hb + 4 * (sz + 1)
    nlb = int(info.num_label_bytes)
    lab = a[off:off + nlb]
    nu = int(np.frombuffer(lab[:8], dtype="<u8")[0])
    sw, cw = self.stored_width, self.comp_width
    self.uniq = _unpack(lab[8:8 + nu * sw], sw, nu)
    self.comp = lab[8 + nu * sw: 8 + nu * sw + sz * cw]
    ncomp = int(_unpack(self.comp, cw, sz).sum())
    kw = _byte_width(nu)
    k0 = 8 + nu * sw + sz * cw
    self.key_width = kw
    self.n_keys = ncomp
    self.keys_raw = lab[k0:k0 + ncomp * kw]        # re-keyed on the device, never unpacked here
    order = int(info.markov_model_order)
    mb = 0 if order == 0 else ((4 ** order) * 5 + 4) // 8      # header.hpp:284-297
    self.model = a[off + nlb: off + nlb + mb]
    crack_bytes = int(np.frombuffer(self.zidx, dtype="<u4").astype(np.int64).sum())
    c0 = off + nlb + mb
    self.cracks = a[c0:c0 + crack_bytes]
    self.crcs = a[len(a) - 4 * sz:]


class SharedStream:
  """The merged stream in the node-local shared mapping (rank 0's view): bytes-like through
  .view() / bytes(); .ptr / .n for the C-ABI."""

  def __init__(self, mm, n: int):
    self._mm = mm
    self.n = int(n)
    self.ptr = C.addressof(C.c_ubyte.from_buffer(mm))

  def __len__(self):
    return self.n

  def view(self) -> memoryview:
    return memoryview(self._mm)[:self.n]

  def tobytes(self) -> bytes:
    return bytes(self.view())

  __bytes__ = tobytes

  def __eq__(self, other):
    if isinstance(other, (bytes, bytearray, memoryview)):
      return self.tobytes() == bytes(other)
    if hasattr(other, "tobytes"):
      return self.tobytes() == other.tobytes()
    return NotImplemented


class _SharedOutput:
  """One file in /dev/shm mapped by every rank of the node: each rank writes its slab's
  sections at their final offsets (the compressed bytes go device ->
  host on every GPU's own link, not GPU <-> GPU).  Grows geometrically, reused across calls."""

  def __init__(self, rank: int, world: int, register: bool = False):
    import mmap
    import os
    self._mmap, self._os = mmap, os
    self.rank = rank
    self.register = bool(register)
    self.registered = False
    self._reg_base = None
    obj = [None]
    if rank == 0:
      # a name nobody else has (two mappings of one codec, or two codecs, must never share a file)
      import tempfile
      fd, obj[0] = tempfile.mkstemp(prefix="ckl_amd_", dir="/dev/shm")
      os.close(fd)
    dist.broadcast_object_list(obj, src=0)
    self.path = obj[0]
    self.cap = 0
    self.mm = None

  def ensure(self, nbytes: int):
    """Collective: every rank calls it with the same size."""
    if nbytes <= self.cap:
      return
    if self.mm is not None:
      self._unregister()
      self.mm.close()
      self.mm = None
    cap = max(1 << 20, int(nbytes * 1.5))
    if self.rank == 0:
      with open(self.path, "wb") as f:
        f.truncate(cap)
    dist.barrier()
    with open(self.path, "r+b") as f:
      self.mm = self._mmap.mmap(f.fileno(), cap)
    self.cap = cap
    if self.register:
      # page-locked: the ranks' crack codes come straight from their GPUs into the mapping
      base = C.addressof(C.c_ubyte.from_buffer(self.mm))
      self.registered = _lib.lib().ckl_host_register(base, cap) == _lib.CKL_OK
      self._reg_base = base if self.registered else None
    dist.barrier()

  def array(self) -> np.ndarray:
    return np.frombuffer(self.mm, dtype=np.uint8)

  def _unregister(self):
    if self._reg_base is not None:
      _lib.lib().ckl_host_unregister(self._reg_base)
      self._reg_base = None
      self.registered = False

  def close(self):
    try:
      if self.mm is not None:
        self._unregister()
        self.mm.close()
        self.mm = None
      if self.rank == 0 and self._os.path.exists(self.path):
        self._os.unlink(self.path)
    except Exception:
      pass

  def __del__(self):
    self.close()


class DeviceStream:
  """A .ckl stream resident in HBM: pointer and length (the encoder session's own copy of its last
  stream, ckl_encoder_device_stream, or any device buffer a caller keeps alive through `owner`)."""

  def __init__(self, ptr: int, n: int, owner=None, generation=None):
    self.ptr, self.n, self.owner = int(ptr), int(n), owner
    # an encoder session's stream lives until that session's next encode (or its end): the backend counts
    # its encodes and a stream of an earlier count refuses to be decoded from
    self.generation = generation

  def valid(self) -> bool:
    if self.generation is None or self.owner is None:
      return True
    return getattr(self.owner, "_stream_generation", None) == self.generation

  def check(self):
    if not self.valid():
      raise RuntimeError("crackle_amd: the device-resident stream was overwritten by a later encode of its session")

  def __len__(self):
    return self.n


class HipDecodeSession:
  def __init__(self, binary, z_start: int, z_end: int, device_index: int):
    self._L = _lib.lib()
    self._h = C.c_void_p()
    self._binary = binary   # keep alive
    if isinstance(binary, DeviceStream):
      binary.check()
      rc = self._L.ckl_decoder_create_device(binary.ptr, binary.n, z_start, z_end, device_index, C.byref(self._h))
    else:
      ptr, n = _lib.as_pointer(binary)
      rc = self._L.ckl_decoder_create(ptr, n, z_start, z_end, device_index, C.byref(self._h))
    if rc != _lib.CKL_OK:
      raise RuntimeError(_lib.last_error())

  def run(self, out: torch.Tensor, label: Optional[int] = None):
    if isinstance(self._binary, DeviceStream):
      self._binary.check()      # a session kept across an encode would read freed or rewritten bytes
    rc = self._L.ckl_decoder_run(self._h, out.data_ptr(), out.numel() * out.element_size(), int(label is not None), int(label or 0))
    if rc != _lib.CKL_OK:
      raise RuntimeError(_lib.last_error())

  def timing(self) -> Tuple[float, float]:
    p, k = C.c_float(), C.c_float()
    self._L.ckl_decoder_last_timing(self._h, C.byref(p), C.byref(k))
    return p.value, k.value

  def stage_events(self, on: bool):
    """HIP events between the kernels of the following runs (the per-stage table of stages()) on or off; the
    events around the whole pipeline (timing()) are always recorded."""
    self._L.ckl_decoder_set_stage_events(self._h, int(bool(on)))

  def stages(self):
    """[(stage name, ms)] of the last run, in launch order."""
    out = []
    i = 0
    while True:
      name, ms = C.c_char_p(), C.c_float()
      if self._L.ckl_decoder_stage_timing(self._h, i, C.byref(name), C.byref(ms)) != _lib.CKL_OK:
        break
      out.append((name.value.decode(), ms.value))
      i += 1
    return out

  def close(self):
    if self._h:
      self._L.ckl_decoder_destroy(self._h)
      self._h = C.c_void_p()

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass


class HipBackend:
  """Device-resident volumes: torch tensors of shape (sz, sy, sx) on the rank's GPU."""
  merges_unique_in_encode = True      # encode() accepts overrides["merge_unique"]

  def __init__(self, device_index: int = 0, zero_copy: bool = False):
    self.device_index = int(device_index)
    self.zero_copy = bool(zero_copy)   # encode() returns _lib.HostStream (the library's buffer) instead of bytes
    self._L = _lib.lib()
    self._enc = None
    self._enc_key = None
    self._stream_generation = 0       # DeviceStream objects of earlier encodes are dead (DeviceStream.check)

  def _encoder(self, shape, itemsize):
    key = (tuple(shape), itemsize)
    if self._enc_key != key:
      self._stream_generation += 1
      if self._enc:
        self._L.ckl_encoder_destroy(self._enc)
      h = C.c_void_p()
      rc = self._L.ckl_encoder_create(shape[0], shape[1], shape[2], itemsize, self.device_index, C.byref(h))
      if rc != _lib.CKL_OK:
        raise RuntimeError(_lib.last_error())
      self._enc, self._enc_key = h, key
    return self._enc

  def itemsize(self, vol) -> int:
    return vol.element_size()

  def stats(self, vol, shape):
    e = self._encoder(shape, vol.element_size())
    mx, pairs, first, last = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    rc = self._L.ckl_encoder_stats(e, vol.data_ptr(), shape[0], shape[1], shape[2], C.byref(mx), C.byref(pairs), C.byref(first), C.byref(last))
    if rc != _lib.CKL_OK:
      raise RuntimeError(_lib.last_error())
    return mx.value, pairs.value, first.value, last.value

  def markov_hist(self, vol, shape, crack_format: int, order: int) -> np.ndarray:
    e = self._encoder(shape, vol.element_size())
    hist = np.zeros((4 ** order) * 4, dtype=np.uint32)
    rc = self._L.ckl_encoder_markov_stats(e, vol.data_ptr(), shape[0], shape[1], shape[2], int(crack_format), int(order), hist.ctypes.data)
    if rc != _lib.CKL_OK:
      raise RuntimeError(_lib.last_error())
    return hist

  def encode(self, vol, shape, allow_pins=False, fortran_order=True, markov_model_order=0, overrides=None) -> bytes:
    e = self._encoder(shape, vol.element_size())
    ov_ptr = None
    keep = None
    if overrides is not None:
      ov = _lib.EncodeOverrides()
      ov.force_crack_format = int(overrides.get("crack_format", -1))
      ov.force_label_format = int(overrides.get("label_format", -1))
      ov.force_stored_width = int(overrides.get("stored_width", 0))
      model = overrides.get("model")
      ov.has_model = int(model is not None)
      if model is not None:
        keep = np.ascontiguousarray(model, dtype=np.uint8)
        ov.model = keep.ctypes.data
      merge = overrides.get("merge_unique")
      if merge is not None:
        # merge(local uint64 array) -> sorted uint64 array of all slabs' labels; runs while the
        # crack trail executes (ckl_encode_overrides.merge_unique)
        state = {"merged": None, "error": None}
        def _cb(ctx, local_ptr, n_local, merged_out, n_out):
          try:
            import time as _time
            _t0 = _time.perf_counter()
            local = np.ctypeslib.as_array(local_ptr, shape=(int(n_local),)).copy() if n_local else np.zeros(0, np.uint64)
            _t1 = _time.perf_counter()
            state["merged"] = np.ascontiguousarray(merge(local), dtype=np.uint64)
            import os as _os2
            if _os2.environ.get("CKL_PROFILE"):
              import sys as _sys
              print(f"[ckl merge callback ms] view+copy={(_t1 - _t0) * 1e3:.2f} merge={(_time.perf_counter() - _t1) * 1e3:.2f}", file=_sys.stderr)
            merged_out[0] = state["merged"].ctypes.data
            n_out[0] = state["merged"].size
            return 0
          except BaseException as exc:      # must not propagate through the C frames
            state["error"] = exc
            return 1
        cb = _lib.MERGE_UNIQUE_FN(_cb)
        ov.merge_unique = cb
        keep = (keep, cb, state)
      ov_ptr = C.byref(ov)
    out, n = C.c_void_p(), C.c_uint64()
    self._stream_generation += 1      # the session's stream buffer in HBM is rewritten (and may move)
    rc = self._L.ckl_encoder_run(
      e, vol.data_ptr(), shape[0], shape[1], shape[2],
      int(bool(allow_pins)), int(fortran_order), int(markov_model_order), 0, 1, 0,
      ov_ptr, C.byref(out), C.byref(n))
    if rc != _lib.CKL_OK:
      if isinstance(keep, tuple) and keep[2]["error"] is not None:
        raise keep[2]["error"]
      raise RuntimeError(_lib.last_error())
    del keep
    if self.zero_copy:
      return _lib.HostStream(out.value, n.value)
    try:
      # with async_host_copy the crack codes may still be on their way into `out`
      # (include/crackle_amd.h: neither read nor free before ckl_encoder_host_wait)
      self.host_wait()
      return C.string_at(out.value, n.value)
    finally:
      self._L.ckl_free(out)

  def components(self, vol, shape, id_base: int, cc_out: np.ndarray, ncomp_out: np.ndarray):
    """Per-voxel component ids of the slab (ids continue from id_base over its slices)
    written into the host arrays cc_out (uint32, x fastest) / ncomp_out (uint32 per slice)."""
    e = self._encoder(shape, vol.element_size())
    rc = self._L.ckl_encoder_components(e, vol.data_ptr(), shape[0], shape[1], shape[2], int(id_base),
                                        cc_out.ctypes.data, ncomp_out.ctypes.data)
    if rc != _lib.CKL_OK:
      raise RuntimeError(_lib.last_error())

  def volume_to_host(self, vol, out: np.ndarray):
    """Copies the slab's labels (x fastest) into the host array `out`."""
    out[...] = vol.reshape(-1).cpu().numpy().view(out.dtype)

  def components_device(self, vol, shape, id_base: int, cc_out: torch.Tensor) -> np.ndarray:
    """Per-voxel component ids of the slab written into the device tensor cc_out (int32 bit
    patterns of uint32 ids, x fastest); returns the component count of every slice."""
    e = self._encoder(shape, vol.element_size())
    nc = np.zeros(max(int(shape[2]), 1), dtype=np.uint32)
    rc = self._L.ckl_encoder_components_device(e, vol.data_ptr(), shape[0], shape[1], shape[2], int(id_base),
                                               cc_out.data_ptr(), nc.ctypes.data)
    if rc != _lib.CKL_OK:
      raise RuntimeError(_lib.last_error())
    return nc[:int(shape[2])]

  def pin_labels_device(self, labels_all: torch.Tensor, cc_all: torch.Tensor, shape_all, ncomp_all: np.ndarray, stored_width: int) -> np.ndarray:
    """The pin label section of the whole volume from labels and global component ids resident on
    this rank's GPU (ckl_encoder_pin_labels); uses the slab session's stream and scratch."""
    if self._enc is None:
      raise RuntimeError("pin_labels_device follows an encode of this backend")
    nc = np.ascontiguousarray(ncomp_all, dtype=np.uint32)
    out_p, out_n = C.c_void_p(), C.c_uint64()
    rc = self._L.ckl_encoder_pin_labels(self._enc, labels_all.data_ptr(), cc_all.data_ptr(), shape_all[0], shape_all[1], shape_all[2],
                                        nc.ctypes.data, int(stored_width), 1, 0, C.byref(out_p), C.byref(out_n))
    if rc != _lib.CKL_OK:
      raise RuntimeError(_lib.last_error())
    try:
      return np.frombuffer(C.string_at(out_p.value, out_n.value), dtype=np.uint8)
    finally:
      self._L.ckl_free(out_p)

  # -- the pin stage sharded by rows (ckl_pins_rows_*): every array is a device tensor of the caller ----------
  def _rows_call(self, fn, *args):
    if fn(self._enc, *args) != _lib.CKL_OK:
      raise RuntimeError(_lib.last_error())

  def pins_rows_first(self, lab_rows, cc_rows, sx, rows, sz, y0, n, first_any, first_kept, comp_label):
    self._rows_call(self._L.ckl_pins_rows_first, lab_rows.data_ptr(), cc_rows.data_ptr(), sx, rows, sz, y0, n,
                    first_any.data_ptr(), first_kept.data_ptr(), comp_label.data_ptr())

  def pins_rows_best(self, lab_rows, cc_rows, sx, rows, sz, y0, n, first_kept, best):
    self._rows_call(self._L.ckl_pins_rows_best, lab_rows.data_ptr(), cc_rows.data_ptr(), sx, rows, sz, y0, n, first_kept.data_ptr(), best.data_ptr())

  def pins_rows_extent(self, lab_rows, cc_rows, sx, rows, sz, y0, n, first_kept, best, choice, ze_plus1):
    self._rows_call(self._L.ckl_pins_rows_extent, lab_rows.data_ptr(), cc_rows.data_ptr(), sx, rows, sz, y0, n,
                    first_kept.data_ptr(), best.data_ptr(), choice.data_ptr(), ze_plus1.data_ptr())

  def pins_rows_ids(self, cc_rows, sx, rows, sz, y0, n, choice, ze_plus1, offsets, ids):
    self._rows_call(self._L.ckl_pins_rows_ids, cc_rows.data_ptr(), sx, rows, sz, y0, n, choice.data_ptr(), ze_plus1.data_ptr(), offsets.data_ptr(), ids.data_ptr())

  def pins_rows_section(self, sx, sy, sz, n, ncomp_all: np.ndarray, comp_label, first_any, choice, ze_plus1, offsets, ids, stored_width: int) -> np.ndarray:
    nc = np.ascontiguousarray(ncomp_all, dtype=np.uint32)
    out_p, out_n = C.c_void_p(), C.c_uint64()
    self._rows_call(self._L.ckl_pins_rows_section, sx, sy, sz, n, nc.ctypes.data, comp_label.data_ptr(), first_any.data_ptr(), choice.data_ptr(),
                    ze_plus1.data_ptr(), offsets.data_ptr(), ids.data_ptr(), int(stored_width), 1, 0, C.byref(out_p), C.byref(out_n))
    try:
      return np.frombuffer(C.string_at(out_p.value, out_n.value), dtype=np.uint8)
    finally:
      self._L.ckl_free(out_p)

  def keep_device_stream(self, shape, itemsize: int, keep: bool = True):
    """Following encodes of this shape also leave their whole stream in HBM (device_stream())."""
    self._L.ckl_encoder_keep_device_stream(self._encoder(shape, itemsize), int(bool(keep)))

  def async_host_copy(self, shape, itemsize: int, on: bool = True):
    """Following encodes of this shape return when their stream is complete in HBM; the crack codes and the
    flat label section (with its crc32c, computed on the device) reach the returned host buffer in the
    background (needs keep_device_stream): host_wait() before its bytes are read or released."""
    self._L.ckl_encoder_async_host_copy(self._encoder(shape, itemsize), int(bool(on)))

  def host_wait(self):
    """Blocks until the last encode's host buffer is complete (ckl_encoder_host_wait)."""
    if self._enc and self._L.ckl_encoder_host_wait(self._enc) != _lib.CKL_OK:
      raise RuntimeError(_lib.last_error())

  def device_stream(self) -> DeviceStream:
    """The last encode's stream in HBM (valid until the session's next encode)."""
    p, n = C.c_void_p(), C.c_uint64()
    rc = self._L.ckl_encoder_device_stream(self._enc, C.byref(p), C.byref(n))
    if rc != _lib.CKL_OK:
      raise RuntimeError(_lib.last_error())
    return DeviceStream(p.value, n.value, owner=self, generation=self._stream_generation)

  def defer_codes(self, shape, itemsize: int, defer: bool):
    """Following encodes of this shape leave the crack codes in HBM for codes_to_host."""
    self._L.ckl_encoder_defer_codes(self._encoder(shape, itemsize), int(bool(defer)))

  def codes_to_host(self, dst_ptr: int, capacity: int) -> int:
    """Copies the last encode's crack codes (slice order) to host address dst_ptr."""
    n = C.c_uint64()
    rc = self._L.ckl_encoder_codes_to_host(self._enc, dst_ptr, capacity, C.byref(n))
    if rc != _lib.CKL_OK:
      raise RuntimeError(_lib.last_error())
    return int(n.value)

  def encoder_timing(self) -> Tuple[float, float]:
    p, k = C.c_float(), C.c_float()
    if self._enc:
      self._L.ckl_encoder_last_timing(self._enc, C.byref(p), C.byref(k))
    return p.value, k.value

  def walk_paths(self) -> Tuple[int, int]:
    """(slices walked by the hand-scheduled loop, slices walked by the compiled one) of the last encode."""
    f, c = C.c_uint32(), C.c_uint32()
    if self._enc and self._L.ckl_encoder_walk_paths(self._enc, C.byref(f), C.byref(c)) != _lib.CKL_OK:
      raise RuntimeError(_lib.last_error())
    return f.value, c.value

  def walk_step_kinds(self, max_slices: int = 1 << 16):
    """(slices, 5) array of the last encode's serial trail by kind of step: along the only remaining edge, branch +
    lowest edge, dead end, chain end, longest run without a branch (ckl_encoder_walk_step_kinds)."""
    buf = (C.c_uint32 * (5 * max_slices))()
    n = C.c_uint32()
    if not self._enc:
      return np.zeros((0, 5), np.uint32)
    if self._L.ckl_encoder_walk_step_kinds(self._enc, buf, max_slices, C.byref(n)) != _lib.CKL_OK:
      raise RuntimeError(_lib.last_error())
    return np.frombuffer(buf, dtype=np.uint32, count=5 * n.value).reshape(n.value, 5).copy()

  def open_decoder(self, binary: bytes, z_start: int, z_end: int):
    return HipDecodeSession(binary, z_start, z_end, self.device_index)

  def __del__(self):
    try:
      if self._enc:
        self._L.ckl_encoder_destroy(self._enc)
    except Exception:
      pass


class ShardedCodec:
  """Sharded compress / decode over the default process group (or single process)."""

  def __init__(self, backend, rank: int = 0, world: int = 1, device="cpu", compute_device=None, force_sharded: bool = False):
    """device: where the tensors of the collectives live (cuda for RCCL, cpu for gloo);
    compute_device: where the small tensor work of the label merge runs (default: device);
    force_sharded: a single rank takes the sharded path too (every collective runs over a process
    group of one: how the RCCL calls are exercised on a one-GPU box)."""
    self.force_sharded = bool(force_sharded)
    self.backend = backend
    self.rank, self.world = int(rank), int(world)
    self.device = torch.device(device)
    self.compute_device = torch.device(compute_device) if compute_device is not None else self.device
    self._shared = None     # node-local output mapping of the sharded encoder
    self._shared_vol = None # node-local labels + component ids of the whole volume (pin encoding)

  # -- encode -------------------------------------------------------------------
  def _gather_rows(self, mine: np.ndarray) -> np.ndarray:
    """Every rank's short int64 vector as the rows of one array: one collective into ONE tensor where the backend
    has it (RCCL), one transfer back to the host."""
    t = torch.from_numpy(np.ascontiguousarray(mine, dtype=np.int64)).to(self.device)
    import os
    if dist.get_backend() == "nccl" and not os.environ.get("CKL_SHARDED_LIST_GATHER"):
      out = torch.empty(self.world * t.numel(), dtype=torch.int64, device=self.device)
      dist.all_gather_into_tensor(out, t)
      return out.cpu().numpy().reshape(self.world, t.numel())
    parts = [torch.empty_like(t) for _ in range(self.world)]
    dist.all_gather(parts, t)
    return torch.stack(parts).cpu().numpy()

  def compress(self, vol, slab_shape, markov_model_order: int = 0, allow_pins: bool = False, fortran_order: bool = True, defer: bool = False):
    """Every rank passes its own z-slab (slab_shape = (sx, sy, sz_local)).  Returns the
    stream of the whole volume on rank 0, None elsewhere.
    defer=True returns a callable instead, as soon as this rank's slab stream is complete in HBM: with a
    backend whose crack codes travel to the host in the background (HipBackend.async_host_copy) the
    caller can decode from the resident stream meanwhile; calling it waits for the copy, joins the
    ranks and returns what compress() returns."""
    be = self.backend
    if self.world == 1 and not self.force_sharded:
      whole = be.encode(vol, slab_shape, allow_pins, fortran_order, markov_model_order, None)
      def _done():
        if hasattr(be, "host_wait"):
          be.host_wait()
        return whole
      return _done if defer else _done()
    import os, time
    prof = os.environ.get("CKL_PROFILE") is not None
    marks = []
    t_last = [time.perf_counter()]
    def mark(name):
      if prof:
        now = time.perf_counter()
        marks.append((name, (now - t_last[0]) * 1e3))
        t_last[0] = now

    sx, sy, sz = slab_shape
    voxels_local = sx * sy * sz
    # 1. whole-volume reductions: pixel_pairs (linear, crosses slab boundaries) and max label
    mx, pairs, first, last = be.stats(vol, slab_shape)
    # uint64 labels travel as their int64 bit patterns (labels >= 2^63 do not fit torch.int64 as values)
    mine_np = np.array([pairs, mx, first, last, voxels_local], dtype=np.uint64).view(np.int64)
    table = self._gather_rows(mine_np).view(np.uint64)
    tot_pairs = int(table[:, 0].sum())
    nonempty = [r for r in range(self.world) if table[r, 4] > 0]
    for a, b in zip(nonempty[:-1], nonempty[1:]):
      tot_pairs += int(table[a, 3] == table[b, 2])     # the pair straddling a slab boundary
    tot_voxels = int(table[:, 4].sum())
    max_label = int(table[:, 1].max())
    crack_format = PERMISSIBLE if tot_pairs < tot_voxels // 2 else IMPERMISSIBLE   # crackle.hpp:50-55
    # pins only with the IMPERMISSIBLE crack format and more than one slice (crackle.hpp:50-64)
    sz_all = int(sum(int(table[r, 4]) for r in range(self.world)) // max(sx * sy, 1))
    use_pins = bool(allow_pins) and crack_format == IMPERMISSIBLE and sz_all > 1
    overrides = dict(crack_format=crack_format, label_format=FLAT, stored_width=_byte_width(max_label))
    mark("stats+allgather")

    # 2. one markov model for all slabs
    order = int(markov_model_order)
    if order > 0:
      hist = be.markov_hist(vol, slab_shape, crack_format, order)
      h = torch.from_numpy(hist.astype(np.int64)).to(self.device)
      dist.all_reduce(h, op=dist.ReduceOp.SUM)
      hist = h.cpu().numpy()
      if int(hist.sum()) == 0:
        order = 0                                      # no chains anywhere (crackle.hpp:107-118)
      else:
        overrides["model"] = stats_to_model(hist).reshape(-1)

    # 3. every rank encodes its slab and places the sections at their final offsets of one
    #    node-local shared buffer.  Crack codes, z-index entries, component counts and crcs
    #    are concatenated verbatim (crackle/operations.py:508-548); only the flat label keys
    #    are re-keyed against the merged, sorted unique-label list (labels.hpp:92-152).
    import os as _os
    early_merge = bool(getattr(be, "merges_unique_in_encode", False)) and not _os.environ.get("CKL_SHARDED_LEGACY") and not getattr(self, "_legacy_merge", False)
    if early_merge:
      # the slabs' unique labels are exchanged from inside the encode, under the crack trail: the
      # label sections then share one unique list and one key width and concatenate as they are
      cdev0 = self.compute_device
      def _merge_unique(local: np.ndarray) -> np.ndarray:
        if _os.environ.get("CKL_TEST_MERGE_FAIL"):      # testing: exercises the fallback below
          raise RuntimeError("forced failure of the in-encode merge")
        # two collectives (sizes, then the padded lists), each into ONE tensor, and one sort: this runs on
        # the label stream's host thread under the crack trail and should stay shorter than the trail
        # (1.0 ms as a group of one over RCCL at C2: a dozen small torch operations; putting them on a
        # high-priority stream changed nothing)
        dev_c = self.device
        _t = [time.perf_counter()] if prof else None
        def _m(name):
          if prof:
            now = time.perf_counter(); marks.append(("merge:" + name, (now - _t[0]) * 1e3)); _t[0] = now
        flat_gather = dist.get_backend() == "nccl"       # gloo has no all_gather_into_tensor
        mine_n = torch.tensor([local.size], dtype=torch.int64, device=dev_c)
        if flat_gather:
          sizes_t = torch.empty(self.world, dtype=torch.int64, device=dev_c)
          dist.all_gather_into_tensor(sizes_t, mine_n)
        else:
          parts = [torch.empty_like(mine_n) for _ in range(self.world)]
          dist.all_gather(parts, mine_n)
          sizes_t = torch.cat(parts)
        sizes_h = sizes_t.cpu()                          # one transfer, not one per rank
        _m("sizes")
        maxn = max(int(sizes_h.max()), 1)
        pad = torch.zeros(maxn, dtype=torch.int64, device=dev_c)
        pad[:local.size] = torch.from_numpy(local.view(np.int64)).to(dev_c)
        if flat_gather:
          allv = torch.empty(self.world * maxn, dtype=torch.int64, device=dev_c)
          dist.all_gather_into_tensor(allv, pad)
        else:
          parts = [torch.empty_like(pad) for _ in range(self.world)]
          dist.all_gather(parts, pad)
          allv = torch.cat(parts)
        _m("lists")
        allv = allv.view(self.world, maxn).to(cdev0)
        keep = torch.arange(maxn, device=cdev0)[None, :] < sizes_h.to(cdev0)[:, None]
        sign = -(1 << 63)                                # flipping the sign bit maps the unsigned order onto the signed one
        m = torch.unique(allv[keep] ^ sign) ^ sign
        out = m.cpu().numpy().view(np.uint64)
        _m("unique")
        return out
      overrides["merge_unique"] = _merge_unique
    direct = hasattr(be, "codes_to_host")      # the crack codes go from HBM straight to their place in the shared buffer
    if direct:
      be.defer_codes(slab_shape, be.itemsize(vol), True)
    try:
      try:
        slab = be.encode(vol, slab_shape, False, fortran_order, order, overrides)
      except Exception as exc:
        if not early_merge:
          raise
        # insurance: should the exchange from inside the encode fail on this installation (it
        # would on every rank alike), this and all later calls merge after the encode instead
        import sys as _sys
        print(f"crackle_amd: merging unique labels inside the encode failed ({exc!r}); falling back to the merge after the encode", file=_sys.stderr)
        self._legacy_merge = True
        early_merge = False
        overrides.pop("merge_unique", None)
        slab = be.encode(vol, slab_shape, False, fortran_order, order, overrides)
    finally:
      if direct:
        be.defer_codes(slab_shape, be.itemsize(vol), False)
    mark("encode")
    sec = _SlabSections(slab)
    mark("sections")
    dev = self.device
    table = self._gather_rows(np.array([len(sec.uniq), sec.n_keys, len(sec.cracks), sec.sz], dtype=np.int64))
    packed_dev = None
    cdev = self.compute_device
    if early_merge:
      # every slab's section already holds the merged list and keys of the final width
      mark("gathers")
      merged = None
      n_merged = len(sec.uniq)
      kw = sec.key_width
      if int(table[:, 0].min()) != int(table[:, 0].max()):
        raise RuntimeError("the slabs' label sections disagree on the merged unique list")
    else:
      max_u = int(table[:, 0].max())
      mine_u = torch.zeros(max(max_u, 1), dtype=torch.int64, device=dev)
      mine_u[:len(sec.uniq)] = torch.from_numpy(sec.uniq).to(dev)
      all_u = [torch.empty_like(mine_u) for _ in range(self.world)]
      dist.all_gather(all_u, mine_u)
      mark("gathers")
      # labels are uint64 bit patterns in int64 tensors: flipping the sign bit maps the unsigned
      # order onto the signed one (labels >= 2^63 must sort last, labels.hpp:92-121)
      sign = torch.tensor(-(1 << 63), dtype=torch.int64, device=cdev)
      merged_t = torch.unique(torch.cat([all_u[r][:int(table[r, 0])].to(cdev) for r in range(self.world)]) ^ sign)   # sorted as unsigned
      merged = merged_t ^ sign
      n_merged = int(merged.numel())
      kw = _byte_width(n_merged)
    if not early_merge and not use_pins and sec.n_keys:
      # keys -> positions in the merged list, packed at their new width, all on `dev`: the only
      # host work is the copy of the packed bytes into the shared buffer
      signed = {1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}
      remap = torch.searchsorted(merged_t, mine_u[:len(sec.uniq)].to(cdev) ^ sign)
      raw = torch.from_numpy(np.array(sec.keys_raw, copy=True)).to(cdev)
      old = raw.view(signed[sec.key_width]).to(torch.int64)
      if sec.key_width in (2, 4):
        old &= (1 << (8 * sec.key_width)) - 1
      packed_dev = remap[old].to(signed[kw]).contiguous().view(torch.uint8)      # collected after the host copies below
    mark("labels")

    sw, cw = sec.stored_width, sec.comp_width
    sz_tot = int(table[:, 3].sum())
    z_before = int(table[:self.rank, 3].sum())
    pins_section = None
    import os as _os3
    rows_done = False
    if use_pins and hasattr(be, "pins_rows_first") and not _os3.environ.get("CKL_PINS_ON_ROOT"):
      # every rank works on its own rows of the whole volume; rank 0 only writes the section
      got = self._pins_by_rows(be, vol, slab_shape, table, sec, cw, sw, mark)
      if got is not None:
        rows_done = True
        pins_section = got if self.rank == 0 else None
    if rows_done:
      pass
    elif use_pins and hasattr(be, "pin_labels_device"):
      # Pin labels (pins.hpp:348-403, labels.hpp:157-344): candidate pins are z-runs per (x,y)
      # column over the WHOLE volume and the greedy cover is order sensitive.  The slabs' labels
      # and component ids (numbered continuously over all slices) are collected in rank 0's HBM
      # — device to device over the collective backend when it carries device tensors (RCCL) —
      # and rank 0 runs the pin passes there (ckl_encoder_pin_labels); nothing volumetric
      # touches the host.
      ncomp_mine = _unpack(sec.comp, cw, sec.sz)
      id_base = int(table[:self.rank, 1].sum())
      sxy = sx * sy
      cc_mine = torch.empty(max(sxy * sec.sz, 1), dtype=torch.int32, device=vol.device)
      nc_mine = be.components_device(vol, slab_shape, id_base, cc_mine)
      if not np.array_equal(nc_mine.astype(np.int64), ncomp_mine):
        raise RuntimeError("component counts of the slab stream and of the component pass differ")
      counts = [int(table[r, 3]) * sxy for r in range(self.world)]
      lab_all = self._collect_on_root(vol.reshape(-1), counts)
      cc_all = self._collect_on_root(cc_mine[:sxy * sec.sz], counts)
      max_sz = max(int(table[:, 3].max()), 1)
      nc_pad = torch.zeros(max_sz, dtype=torch.int64, device=dev)
      nc_pad[:sec.sz] = torch.from_numpy(nc_mine.astype(np.int64)).to(dev)
      nc_every = [torch.empty_like(nc_pad) for _ in range(self.world)]
      dist.all_gather(nc_every, nc_pad)
      if self.rank == 0:
        nc_all = np.concatenate([nc_every[r][:int(table[r, 3])].cpu().numpy() for r in range(self.world)]).astype(np.uint32)
        pins_section = be.pin_labels_device(lab_all, cc_all, (sx, sy, sz_tot), nc_all, sw)
        del lab_all, cc_all
    elif use_pins:
      # backends without a device stage (the CPU orchestration tests): the slabs' labels and
      # component ids are laid side by side in a second node-local mapping and rank 0 runs the
      # host statement of the stage once.
      ncomp_mine = _unpack(sec.comp, cw, sec.sz)
      id_base = int(table[:self.rank, 1].sum())
      sxy = sx * sy
      item = be.itemsize(vol)
      vol_bytes = sxy * sz_tot * item
      cc_off = (vol_bytes + 15) // 16 * 16
      nc_off = cc_off + sxy * sz_tot * 4
      if self._shared_vol is None:
        self._shared_vol = _SharedOutput(self.rank, self.world)
      self._shared_vol.ensure(nc_off + 4 * sz_tot)
      big = self._shared_vol.array()
      lab_all = big[:vol_bytes].view({1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[item])
      cc_all = big[cc_off:cc_off + sxy * sz_tot * 4].view(np.uint32)
      nc_all = big[nc_off:nc_off + 4 * sz_tot].view(np.uint32)
      mine = slice(sxy * z_before, sxy * (z_before + sec.sz))
      be.volume_to_host(vol, lab_all[mine])
      nc_mine = np.zeros(sec.sz, dtype=np.uint32)
      cc_mine = np.zeros(sxy * sec.sz, dtype=np.uint32)
      be.components(vol, slab_shape, id_base, cc_mine, nc_mine)
      if not np.array_equal(nc_mine.astype(np.int64), ncomp_mine):
        raise RuntimeError("component counts of the slab stream and of the component pass differ")
      cc_all[mine] = cc_mine
      nc_all[z_before:z_before + sec.sz] = nc_mine
      dist.barrier()
      if self.rank == 0:
        L = _lib.lib()
        out_p, out_n = C.c_void_p(), C.c_uint64()
        rc = L.ckl_pin_labels_host(lab_all.ctypes.data, item, cc_all.ctypes.data, sx, sy, sz_tot, nc_all.ctypes.data,
                                   sw, 1, 0, C.byref(out_p), C.byref(out_n))
        if rc != _lib.CKL_OK:
          raise RuntimeError(_lib.last_error())
        pins_section = np.frombuffer(C.string_at(out_p.value, out_n.value), dtype=np.uint8)
        L.ckl_free(out_p)
    if use_pins:
      n_pin = torch.tensor([0 if pins_section is None else len(pins_section)], dtype=torch.int64, device=dev)
      dist.broadcast(n_pin, src=0)
      label_bytes = int(n_pin.item())
      kw = 1
      mark("pins")
    else:
      n_keys = int(table[:, 1].sum())
      label_bytes = 8 + n_merged * sw + sz_tot * cw + n_keys * kw
    o_zidx = 29
    o_labels = o_zidx + 4 * (sz_tot + 1)
    o_comp = o_labels + 8 + n_merged * sw
    o_keys = o_comp + sz_tot * cw
    o_model = o_labels + label_bytes
    o_cracks = o_model + len(sec.model)
    o_tail = o_cracks + int(table[:, 2].sum())
    total = o_tail + 4 * (sz_tot + 1)
    keys_before = int(table[:self.rank, 1].sum())
    cracks_before = int(table[:self.rank, 2].sum())

    if self._shared is None:
      self._shared = _SharedOutput(self.rank, self.world, register=direct)
    o_part = (total + 63) // 64 * 64            # behind the stream: every rank's partial crcs of the label section
    self._shared.ensure(o_part + 16 * self.world)
    out = self._shared.array()
    L = _lib.lib()
    base = C.addressof(C.c_ubyte.from_buffer(self._shared.mm))
    # the big host copies first: the re-keying above is still running on the device
    out[o_zidx + 4 * z_before: o_zidx + 4 * (z_before + sec.sz)] = sec.zidx
    if direct:
      got = be.codes_to_host(base + o_cracks + cracks_before, len(sec.cracks))
      if got != len(sec.cracks):
        raise RuntimeError("crack code bytes of the slab stream and of the encoder session differ")
    else:
      out[o_cracks + cracks_before: o_cracks + cracks_before + len(sec.cracks)] = sec.cracks
    out[o_tail + 4 + 4 * z_before: o_tail + 4 + 4 * (z_before + sec.sz)] = sec.crcs
    if not use_pins:
      if early_merge:
        packed_keys = sec.keys_raw
      else:
        packed_keys = packed_dev.cpu().numpy() if packed_dev is not None else np.zeros(0, np.uint8)
      c_at, c_len = o_comp + cw * z_before, cw * sec.sz
      k_at, k_len = o_keys + kw * keys_before, len(packed_keys)
      out[c_at:c_at + c_len] = sec.comp
      out[k_at:k_at + k_len] = packed_keys
      # the label section's crc32c is put together from the ranks' own parts (ckl_crc32c_combine)
      part = np.array([L.ckl_crc32c(base + c_at, c_len), c_len, L.ckl_crc32c(base + k_at, k_len), k_len], dtype="<u4")
      out[o_part + 16 * self.rank: o_part + 16 * (self.rank + 1)] = part.view(np.uint8)
    if self.rank == 0:
      if use_pins:
        out[o_labels:o_labels + label_bytes] = pins_section
      else:
        uniq_g = sec.uniq if early_merge else merged.cpu().numpy()      # int64 bit patterns, ascending as uint64
        out[o_labels:o_labels + 8] = np.frombuffer(np.array([len(uniq_g)], dtype="<u8").tobytes(), dtype=np.uint8)
        out[o_labels + 8:o_comp] = _pack(uniq_g, sw)
      out[o_model:o_cracks] = sec.model
      head = bytearray(sec.header)
      if use_pins:
        fmt = int.from_bytes(head[5:7], "little")
        fmt = (fmt & ~(3 << 5)) | (2 << 5)          # label_format = PINS_VARIABLE_WIDTH (header.hpp:216-224)
        head[5:7] = fmt.to_bytes(2, "little")
      head[15:19] = int(sz_tot).to_bytes(4, "little")
      head[20:28] = int(label_bytes).to_bytes(8, "little")
      head[28] = _crc8(bytes(head[5:28]))
      out[0:29] = np.frombuffer(bytes(head), dtype=np.uint8)
    mark("place")
    def _finish():
      if direct and hasattr(be, "host_wait"):
        be.host_wait()                      # the crack codes are in their place in the shared buffer
      dist.barrier()
      if self.rank != 0:
        return None
      return self._seal(out, base, L, o_zidx, sz_tot, use_pins, o_labels, label_bytes, o_part, o_comp, o_tail, total, mark, marks, prof)
    return _finish if defer else _finish()

  def _seal(self, out, base, L, o_zidx, sz_tot, use_pins, o_labels, label_bytes, o_part, o_comp, o_tail, total, mark, marks, prof):
    """Rank 0, after the barrier: z-index crc and the label section's crc32c from the ranks' parts."""
    out[o_zidx + 4 * sz_tot: o_zidx + 4 * sz_tot + 4] = np.frombuffer(int(L.ckl_crc32c(base + o_zidx, 4 * sz_tot)).to_bytes(4, "little"), dtype=np.uint8)
    if use_pins:
      labels_crc = int(L.ckl_crc32c(base + o_labels, label_bytes))
    else:
      parts = np.frombuffer(bytes(out[o_part:o_part + 16 * self.world]), dtype="<u4").reshape(self.world, 4)
      labels_crc = int(L.ckl_crc32c(base + o_labels, o_comp - o_labels))        # count + unique labels
      for col in (0, 2):                                                          # component counts, then keys, in rank order
        for r in range(self.world):
          labels_crc = int(L.ckl_crc32c_combine(labels_crc, int(parts[r, col]), int(parts[r, col + 1])))
    out[o_tail:o_tail + 4] = np.frombuffer(labels_crc.to_bytes(4, "little"), dtype=np.uint8)
    mark("crc")
    if prof:
      import sys
      print("[ckl sharded compress ms] " + " ".join(f"{n}={v:.2f}" for n, v in marks), file=sys.stderr)
    return SharedStream(self._shared.mm, total)

  # -- pins, sharded by rows ------------------------------------------------------
  def _rows_of(self, sy: int):
    """[(y0, rows)] of every rank: the rows dealt out as evenly as they go (the first ranks take the remainder)."""
    base, rem = divmod(int(sy), self.world)
    out, y = [], 0
    for r in range(self.world):
      n = base + (1 if r < rem else 0)
      out.append((y, n))
      y += n
    return out

  def _to_row_slabs(self, t3: torch.Tensor, sz_list, rows):
    """This rank's z-slab (sz_local, sy, sx) -> its rows of EVERY slice (sz_total, rows_mine, sx): one
    all_to_all_single (over xGMI with RCCL; staged through the host under gloo).  Unsigned dtypes travel as the
    signed type of the same width (bit patterns)."""
    orig = t3.dtype
    item = t3.element_size()
    sz_l, sy, sx = (int(v) for v in t3.shape)
    y0_me, rows_me = rows[self.rank]
    # as bytes: neither RCCL nor gloo knows every integer width (no 16-bit integers, no unsigned 32 / 64 under gloo)
    send = torch.cat([t3[:, y0:y0 + n, :].reshape(-1) for (y0, n) in rows]).view(torch.uint8) if sz_l else torch.empty(0, dtype=torch.uint8, device=t3.device)
    in_split = [sz_l * n * sx * item for (_, n) in rows]
    out_split = [int(z) * rows_me * sx * item for z in sz_list]
    direct = self.device.type == t3.device.type
    src = send if direct else send.to(self.device)
    recv = torch.empty(sum(out_split), dtype=torch.uint8, device=src.device)
    if self.world > 1:
      dist.all_to_all_single(recv, src, out_split, in_split)
    else:
      recv.copy_(src)
    if not direct:
      recv = recv.to(t3.device)
    return recv.view(orig).view(int(sum(sz_list)), rows_me, sx)

  def _reduce(self, t: torch.Tensor, op: str, unsigned: bool):
    """all_reduce of a device tensor of integer bit patterns in place; unsigned 64-bit order through the sign flip."""
    if self.world == 1:
      return
    sign = -(1 << 63)
    if unsigned:
      t ^= sign
    direct = self.device.type == t.device.type
    w = t if direct else t.to(self.device)
    dist.all_reduce(w, op={"min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX, "sum": dist.ReduceOp.SUM}[op])
    if not direct:
      t.copy_(w.to(t.device))
    if unsigned:
      t ^= sign

  def _agree(self, exc):
    """Do all ranks get past their rank-local step?  One word, all-reduced: if any rank failed, EVERY rank raises here
    (its own error, or a note that another rank failed) instead of the healthy ones blocking in the next collective."""
    if self.world > 1:
      flag = torch.tensor([1 if exc is not None else 0], dtype=torch.int32, device=self.device)
      dist.all_reduce(flag, op=dist.ReduceOp.MAX)
      if int(flag.item()) and exc is None:
        raise RuntimeError("sharded pin stage: another rank failed in its local pass")
    if exc is not None:
      raise exc

  @staticmethod
  def _local(fn):
    """Runs a rank-local step; returns the exception it raised (for _agree) or None."""
    try:
      fn()
      return None
    except Exception as exc:      # noqa: BLE001 (whatever it is, the other ranks must hear of it)
      return exc

  def _pins_by_rows(self, be, vol, slab_shape, table, sec, cw, sw, mark):
    """The pin label section with every rank working on its own ROWS of the whole volume (ckl_pins_rows_*,
    include/crackle_amd.h): labels and component ids are transposed from z-slabs to row slabs, the per-component
    extrema are reduced over the ranks between the device passes, rank 0 writes the section (ordered cover).
    Returns the section on rank 0, b"" elsewhere; None when the stage has to fall back to the collection on
    rank 0 (id lists beyond the budget)."""
    import os
    sx, sy, sz = slab_shape
    sxy = sx * sy
    dev = vol.device
    sz_list = [int(table[r, 3]) for r in range(self.world)]
    sz_tot = int(sum(sz_list))
    N = int(table[:, 1].sum())
    ncomp_mine = _unpack(sec.comp, cw, sec.sz)
    id_base = int(table[:self.rank, 1].sum())
    cc_mine = torch.empty(max(sxy * sec.sz, 1), dtype=torch.int32, device=dev)
    got = {}
    def _components():
      got["nc"] = be.components_device(vol, slab_shape, id_base, cc_mine)
      if not np.array_equal(got["nc"].astype(np.int64), ncomp_mine):
        raise RuntimeError("component counts of the slab stream and of the component pass differ")
    self._agree(self._local(_components))      # before the transposes: a rank that raised would leave the others in all_to_all_single
    nc_mine = got["nc"]
    rows = self._rows_of(sy)
    y0, nrows = rows[self.rank]
    lab_rows = self._to_row_slabs(vol.reshape(sec.sz, sy, sx), sz_list, rows)
    cc_rows = self._to_row_slabs(cc_mine[:sxy * sec.sz].view(sec.sz, sy, sx), sz_list, rows)
    mark("pins:transpose")
    i64 = dict(dtype=torch.int64, device=dev)
    first_any = torch.full((N,), -1, **i64)
    first_kept = torch.full((N,), -1, **i64)
    comp_label = torch.zeros(N, **i64)
    self._agree(self._local(lambda: be.pins_rows_first(lab_rows, cc_rows, sx, nrows, sz_tot, y0, N, first_any, first_kept, comp_label)) if nrows else None)
    self._reduce(first_any, "min", True)
    self._reduce(first_kept, "min", True)
    self._reduce(comp_label, "max", True)
    best = torch.zeros(N, **i64)
    self._agree(self._local(lambda: be.pins_rows_best(lab_rows, cc_rows, sx, nrows, sz_tot, y0, N, first_kept, best)) if nrows else None)
    self._reduce(best, "max", False)
    choice = torch.full((N,), -1, **i64)
    ze = torch.zeros(N, dtype=torch.int32, device=dev)
    self._agree(self._local(lambda: be.pins_rows_extent(lab_rows, cc_rows, sx, nrows, sz_tot, y0, N, first_kept, best, choice, ze)) if nrows else None)
    self._reduce(ze, "max", False)
    if not nrows:      # (a rank without rows: the choice is a function of the reduced arrays alone)
      choice = torch.where(best != 0, best - 1, torch.where(first_kept == -1, first_kept, first_kept >> 16))
    counts = torch.where((choice != -1) & (ze > 0), ze.to(torch.int64) - (choice % sz_tot), torch.zeros_like(choice))
    offsets = torch.zeros(N + 1, **i64)
    torch.cumsum(counts, 0, out=offsets[1:])
    total = int(offsets[-1].item())
    budget = int(os.environ.get("CKL_PIN_IDS_BUDGET", str(1 << 26)))
    if total > budget:      # volumes with long z-runs: the whole-volume stage makes the chosen pins distinct first
      # (decided per call — `total` is the same on every rank, they all take this branch together — and said aloud:
      # it puts the whole volume on rank 0 again)
      if self.rank == 0:
        import sys
        print(f"[crackle_amd] sharded pin stage: id lists of {total} entries exceed CKL_PIN_IDS_BUDGET={budget}; this volume's pin stage runs on rank 0", file=sys.stderr)
      return None
    ids = torch.zeros(max(total, 1), dtype=torch.int32, device=dev)
    self._agree(self._local(lambda: be.pins_rows_ids(cc_rows, sx, nrows, sz_tot, y0, N, choice, ze, offsets, ids)) if (nrows and total) else None)
    self._reduce(ids, "sum", False)      # one rank holds each run: the others add zeros
    mark("pins:passes")
    # component counts of every slice
    max_sz = max(max(sz_list), 1)
    nc_pad = torch.zeros(max_sz, dtype=torch.int64, device=self.device)
    nc_pad[:sec.sz] = torch.from_numpy(nc_mine.astype(np.int64)).to(self.device)
    nc_every = [torch.empty_like(nc_pad) for _ in range(self.world)]
    dist.all_gather(nc_every, nc_pad)
    if self.rank != 0:
      return np.zeros(0, np.uint8)
    nc_all = np.concatenate([nc_every[r][:sz_list[r]].cpu().numpy() for r in range(self.world)]).astype(np.uint32)
    section = be.pins_rows_section(sx, sy, sz_tot, N, nc_all, comp_label, first_any, choice, ze, offsets, ids, sw)
    mark("pins:section")
    return section

  def _collect_on_root(self, mine: torch.Tensor, counts):
    """Concatenates every rank's 1-D device tensor (counts[r] elements, rank order) in rank 0's
    device memory with point-to-point transfers: device to device when the process group carries
    device tensors (RCCL over xGMI), through host staging otherwise (gloo).  None off rank 0."""
    import os
    if os.environ.get("CKL_TEST_NO_COLLECT"):      # testing: the pin stage must not fall back to the collection on rank 0
      raise RuntimeError("the whole volume was asked for on rank 0 (CKL_TEST_NO_COLLECT)")
    direct = self.device.type == mine.device.type
    # the unsigned 16 / 32 / 64-bit dtypes have no entry in the collective backends' type tables: the
    # labels travel as the signed type of the same width (bit patterns; the caller gets `mine`'s dtype back)
    # (16-bit integers, which neither RCCL nor gloo carries, go as bytes)
    as_signed = {torch.uint16: torch.uint8, torch.int16: torch.uint8, torch.uint32: torch.int32, torch.uint64: torch.int64}
    orig_dtype = mine.dtype
    scale = 2 if orig_dtype in (torch.uint16, torch.int16) else 1
    if orig_dtype in as_signed:
      mine = mine.view(as_signed[orig_dtype])
      counts = [c * scale for c in counts]
    if self.rank != 0:
      if counts[self.rank]:
        dist.send(mine.contiguous() if direct else mine.to(self.device), dst=0)
      return None
    whole = torch.empty(max(sum(counts), 1), dtype=mine.dtype, device=mine.device)
    whole[:counts[0]] = mine
    off = counts[0]
    for r in range(1, self.world):
      if not counts[r]:
        continue
      if direct:
        dist.recv(whole[off:off + counts[r]], src=r)
      else:
        stage = torch.empty(counts[r], dtype=mine.dtype, device=self.device)
        dist.recv(stage, src=r)
        whole[off:off + counts[r]] = stage.to(mine.device)
      off += counts[r]
    return whole.view(orig_dtype) if orig_dtype in as_signed else whole

  # -- decode -------------------------------------------------------------------
  def open_decoder(self, binary: Optional[bytes], slab_shape):
    """Makes the stream resident on every rank (broadcast from rank 0) and returns a
    session that decodes this rank's z-range into a caller-provided volume."""
    if self.world > 1 or self.force_sharded:
      n = torch.tensor([len(binary) if self.rank == 0 else 0], dtype=torch.int64, device=self.device)
      dist.broadcast(n, src=0)
      if self.rank == 0:
        buf = torch.frombuffer(bytearray(bytes(binary)), dtype=torch.uint8).to(self.device)
      else:
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=self.device)
      dist.broadcast(buf, src=0)
      binary = bytes(buf.cpu().numpy().tobytes())
    sz = slab_shape[2]
    return self.backend.open_decoder(binary, self.rank * sz, (self.rank + 1) * sz)
