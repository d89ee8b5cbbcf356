"""ctypes binding of libcrackle_amd.so (the C-ABI in include/crackle_amd.h).

The product path has no CPU fallback: if the shared library is missing, loading it
raises, and every compute entry point fails with CKL_ERR_NO_DEVICE when no HIP
device is usable.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# CKL_TUNING_LIB=1 (kernel work only): the -DCKL_TUNING build with cycle stamps and ablation switches
# CKL_LIB_AB=name (kernel work only): libcrackle_amd_<name>.so, an earlier build kept beside the current one for an A/B inside one GPU call
LIB_PATH = os.path.join(HERE, "libcrackle_amd_tuning.so" if os.environ.get("CKL_TUNING_LIB") else
                        ("libcrackle_amd_%s.so" % os.environ["CKL_LIB_AB"]) if os.environ.get("CKL_LIB_AB") else "libcrackle_amd.so")

CKL_OK, CKL_ERR_FORMAT, CKL_ERR_RUNTIME, CKL_ERR_ARG, CKL_ERR_NO_DEVICE, CKL_ERR_CRC = range(6)
MEM_HOST, MEM_DEVICE = 0, 1


class HeaderInfo(C.Structure):
  _fields_ = [
    ("format_version", C.c_uint32), ("label_format", C.c_uint32), ("crack_format", C.c_uint32),
    ("is_signed", C.c_uint32), ("data_width", C.c_uint32), ("stored_data_width", C.c_uint32),
    ("sx", C.c_uint32), ("sy", C.c_uint32), ("sz", C.c_uint32),
    ("fortran_order", C.c_uint32), ("markov_model_order", C.c_uint32), ("is_sorted", C.c_uint32),
    ("num_label_bytes", C.c_uint64), ("header_bytes", C.c_uint64),
  ]


MERGE_UNIQUE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64))


class EncodeOverrides(C.Structure):
  _fields_ = [
    ("force_crack_format", C.c_int32), ("force_label_format", C.c_int32),
    ("force_stored_width", C.c_int32), ("has_model", C.c_int32),
    ("model", C.c_void_p),
    ("merge_unique", MERGE_UNIQUE_FN), ("merge_ctx", C.c_void_p),
  ]


EXPORTS = {
  # name: (restype, argtypes) — one entry per declaration in include/crackle_amd.h
  "ckl_last_error": (C.c_char_p, []),
  "ckl_abi_version": (C.c_int, []),
  "ckl_device_count": (C.c_int, []),
  "ckl_header_info_from_bytes": (C.c_int, [C.c_char_p, C.c_uint64, C.POINTER(HeaderInfo)]),
  "ckl_compress": (C.c_int, [
    C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64,
    C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int64,
    C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
  "ckl_free": (None, [C.c_void_p]),
  "ckl_decompress": (C.c_int, [
    C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int,
    C.c_int64, C.c_int64, C.c_int, C.c_uint64, C.c_int]),
  "ckl_decoder_create": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]),
  "ckl_decoder_create_device": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]),
  "ckl_encoder_keep_device_stream": (C.c_int, [C.c_void_p, C.c_int]),
  "ckl_encoder_async_host_copy": (C.c_int, [C.c_void_p, C.c_int]),
  "ckl_encoder_host_wait": (C.c_int, [C.c_void_p]),
  "ckl_encoder_device_stream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
  "ckl_decoder_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_uint64]),
  "ckl_decoder_label_stats": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]),
  "ckl_decoder_last_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
  "ckl_decoder_stage_timing": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_float)]),
  "ckl_decoder_set_stage_events": (C.c_int, [C.c_void_p, C.c_int]),
  "ckl_decoder_destroy": (None, [C.c_void_p]),
  "ckl_encoder_create": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
  "ckl_encoder_run": (C.c_int, [
    C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
    C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int64,
    C.POINTER(EncodeOverrides), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
  "ckl_encoder_stats": (C.c_int, [
    C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
    C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
  "ckl_encoder_markov_stats": (C.c_int, [
    C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_uint64, C.c_void_p]),
  "ckl_encoder_components": (C.c_int, [
    C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_uint32, C.c_void_p, C.c_void_p]),
  "ckl_encoder_components_device": (C.c_int, [
    C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_uint32, C.c_void_p, C.c_void_p]),
  "ckl_encoder_pin_labels": (C.c_int, [
    C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
    C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
  "ckl_pins_rows_first": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
  "ckl_pins_rows_best": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_void_p, C.c_void_p]),
  "ckl_pins_rows_extent": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
  "ckl_pins_rows_ids": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
  "ckl_pins_rows_section": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
  "ckl_encoder_walk_step_kinds": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint32)]),
  "ckl_encoder_last_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
  "ckl_encoder_walk_paths": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
  "ckl_encoder_destroy": (None, [C.c_void_p]),
  "ckl_pin_labels_host": (C.c_int, [
    C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
    C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
  "ckl_zstack": (C.c_int, [C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
  "ckl_decoder_check": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
  "ckl_decoder_vcg": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]),
  "ckl_voxel_connectivity_graph": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_uint64]),
  "ckl_voxel_connectivity_graph_range": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_uint64]),
  "ckl_point_cloud": (C.c_int, [
    C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int,
    C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
  "ckl_reencode_markov": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
  "ckl_decoder_crack_planes": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
  "ckl_zsplit": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
  "ckl_encoder_defer_codes": (C.c_int, [C.c_void_p, C.c_int]),
  "ckl_encoder_codes_to_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
  "ckl_host_register": (C.c_int, [C.c_void_p, C.c_uint64]),
  "ckl_host_unregister": (C.c_int, [C.c_void_p]),
  "ckl_array_equal": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_int)]),
  "ckl_mode_pooling_2x2x1": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
  "ckl_crc32c": (C.c_uint32, [C.c_void_p, C.c_uint64]),
  "ckl_crc32c_combine": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint64]),
}

_lib = None


def lib():
  """Loads libcrackle_amd.so, failing loudly when it has not been built."""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise ImportError(
        f"{LIB_PATH} is missing: build the HIP extension first "
        "(python -m crackle_amd.build, or __graft_entry__.build()). "
        "crackle_amd has no CPU fallback."
      )
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in EXPORTS.items():
      f = getattr(L, name)   # AttributeError here = ABI drift: fail loudly
      f.restype = res
      f.argtypes = args
    _lib = L
  return _lib


def last_error() -> str:
  return lib().ckl_last_error().decode("utf-8", "replace")


class HostStream:
  """A .ckl stream in the library's own (pinned) host buffer, as ckl_encoder_run hands it
  over: no copy into a Python bytes object.  bytes(stream) / stream.tobytes() copy;
  the buffer goes back to the library (ckl_free) when the object dies."""

  def __init__(self, ptr: int, n: int):
    self.ptr, self.n = int(ptr), int(n)

  def __len__(self):
    return self.n

  def view(self) -> memoryview:
    return memoryview((C.c_ubyte * self.n).from_address(self.ptr)).cast("B")

  def tobytes(self) -> bytes:
    return C.string_at(self.ptr, self.n)

  __bytes__ = tobytes

  def __eq__(self, other):
    if isinstance(other, HostStream):
      return self.tobytes() == other.tobytes()
    if isinstance(other, (bytes, bytearray, memoryview)):
      return self.tobytes() == bytes(other)
    return NotImplemented

  def __del__(self):
    try:
      if self.ptr:
        lib().ckl_free(self.ptr)
        self.ptr = 0
    except Exception:
      pass


def as_pointer(binary):
  """(address-or-bytes, length) of a stream given as bytes, HostStream or any object with
  .ptr / .n (distributed.SharedStream), for c_void_p arguments."""
  if hasattr(binary, "ptr") and hasattr(binary, "n"):
    return binary.ptr, binary.n
  return binary, len(binary)
