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


class HipDecodeSession:
  def __init__(self, binary, z_start: int, z_end: int, device_index: int):
    self._L = _lib.lib()
    self._h = C.c_void_p()
    self._binary = binary   # keep alive
    ptr, n = _lib.as_pointer(binary)
    rc = self._L.ckl_decoder_create(ptr, n, z_start, z_end, device_index, C.byref(self._h))
    if rc != _lib.CKL_OK:
      raise RuntimeError(_lib.last_error())

  def run(self, out: torch.Tensor, label: Optional[int] = None):
    rc = self._L.ckl_decoder_run(self._h, out.data_ptr(), out.numel() * out.element_size(), int(label is not None), int(label or 0))
    if rc != _lib.CKL_OK:
      raise RuntimeError(_lib.last_error())

  def timing(self) -> Tuple[float, float]:
    p, k = C.c_float(), C.c_float()
    self._L.ckl_decoder_last_timing(self._h, C.byref(p), C.byref(k))
    return p.value, k.value

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

  def __init__(self, device_index: int = 0, zero_copy: bool = False):
    self.device_index = int(device_index)
    self.zero_copy = bool(zero_copy)   # encode() returns _lib.HostStream (the library's buffer) instead of bytes
    self._L = _lib.lib()
    self._enc = None
    self._enc_key = None

  def _encoder(self, shape, itemsize):
    key = (tuple(shape), itemsize)
    if self._enc_key != key:
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
      ov_ptr = C.byref(ov)
    out, n = C.c_void_p(), C.c_uint64()
    rc = self._L.ckl_encoder_run(
      e, vol.data_ptr(), shape[0], shape[1], shape[2],
      int(bool(allow_pins)), int(fortran_order), int(markov_model_order), 0, 1, 0,
      ov_ptr, C.byref(out), C.byref(n))
    del keep
    if rc != _lib.CKL_OK:
      raise RuntimeError(_lib.last_error())
    if self.zero_copy:
      return _lib.HostStream(out.value, n.value)
    try:
      return C.string_at(out.value, n.value)
    finally:
      self._L.ckl_free(out)

  def encoder_timing(self) -> Tuple[float, float]:
    p, k = C.c_float(), C.c_float()
    if self._enc:
      self._L.ckl_encoder_last_timing(self._enc, C.byref(p), C.byref(k))
    return p.value, k.value

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

  def __init__(self, backend, rank: int = 0, world: int = 1, device="cpu"):
    self.backend = backend
    self.rank, self.world = int(rank), int(world)
    self.device = torch.device(device)

  # -- encode -------------------------------------------------------------------
  def compress(self, vol, slab_shape, markov_model_order: int = 0, allow_pins: bool = False, fortran_order: bool = True) -> Optional[bytes]:
    """Every rank passes its own z-slab (slab_shape = (sx, sy, sz_local)).  Returns the
    stream of the whole volume on rank 0, None elsewhere."""
    be = self.backend
    if self.world == 1:
      return be.encode(vol, slab_shape, allow_pins, fortran_order, markov_model_order, None)
    if allow_pins:
      raise NotImplementedError("pins do not shard by z alone (SURVEY.md section 8e); encode on one GPU")

    sx, sy, sz = slab_shape
    voxels_local = sx * sy * sz
    # 1. whole-volume reductions: pixel_pairs (linear, crosses slab boundaries) and max label
    mx, pairs, first, last = be.stats(vol, slab_shape)
    mine = torch.tensor([pairs, mx, first, last, voxels_local], dtype=torch.int64, device=self.device)
    everyone = [torch.empty_like(mine) for _ in range(self.world)]
    dist.all_gather(everyone, mine)
    table = torch.stack(everyone).cpu().numpy().astype(np.uint64)
    tot_pairs = int(table[:, 0].sum())
    nonempty = [r for r in range(self.world) if table[r, 4] > 0]
    for a, b in zip(nonempty[:-1], nonempty[1:]):
      tot_pairs += int(table[a, 3] == table[b, 2])     # the pair straddling a slab boundary
    tot_voxels = int(table[:, 4].sum())
    max_label = int(table[:, 1].max())
    crack_format = PERMISSIBLE if tot_pairs < tot_voxels // 2 else IMPERMISSIBLE   # crackle.hpp:50-55
    overrides = dict(crack_format=crack_format, label_format=FLAT, stored_width=_byte_width(max_label))

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

    # 3. per-slab streams, gathered to rank 0 and merged
    slab = bytes(be.encode(vol, slab_shape, False, fortran_order, order, overrides))
    n_mine = torch.tensor([len(slab)], dtype=torch.int64, device=self.device)
    sizes = [torch.empty_like(n_mine) for _ in range(self.world)]
    dist.all_gather(sizes, n_mine)
    sizes = [int(s.item()) for s in sizes]
    cap = max(sizes)
    buf = torch.zeros(cap, dtype=torch.uint8)
    buf[:len(slab)] = torch.frombuffer(bytearray(slab), dtype=torch.uint8)
    buf = buf.to(self.device)
    gathered = [torch.empty_like(buf) for _ in range(self.world)] if self.rank == 0 else None
    dist.gather(buf, gathered, dst=0)
    if self.rank != 0:
      return None
    slabs = [bytes(gathered[r][:sizes[r]].cpu().numpy().tobytes()) for r in range(self.world)]
    return zstack(slabs)

  # -- decode -------------------------------------------------------------------
  def open_decoder(self, binary: Optional[bytes], slab_shape):
    """Makes the stream resident on every rank (broadcast from rank 0) and returns a
    session that decodes this rank's z-range into a caller-provided volume."""
    if self.world > 1:
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
