"""CPU-side checks of the product library: it loads, exports every symbol declared in
include/crackle_amd.h, refuses to compute without a device, and its host-only logic
(header parse, checksums, zstack merge, Python-surface shortcuts) is right.
No compute entry point is exercised here without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import crackle_amd
from crackle_amd import _lib
import golden_cases
from util import golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = golden_cases.small_cases()


def declared_symbols():
  text = open(os.path.join(ROOT, "include", "crackle_amd.h")).read()
  text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
  return sorted(set(re.findall(r"\b(ckl_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
  L = C.CDLL(_lib.LIB_PATH)
  names = declared_symbols()
  assert len(names) >= 19
  for n in names:
    assert hasattr(L, n), f"{n} declared in include/crackle_amd.h but not exported"
  assert set(names) == set(_lib.EXPORTS), "ctypes binding and header disagree"
  assert _lib.lib().ckl_abi_version() == 1


def test_header_info_matches_reference_layout():
  b = golden()["c0_voronoi_u8_pins_m5"]
  h = crackle_amd.header(b)
  assert (h.sx, h.sy, h.sz) == (64, 64, 16)
  assert h.data_width == 1 and h.stored_data_width == 1
  assert h.label_format == crackle_amd.LabelFormat.PINS_VARIABLE_WIDTH
  assert h.crack_format == crackle_amd.CrackFormat.IMPERMISSIBLE
  assert h.markov_model_order == 5 and h.fortran_order and h.is_sorted and not h.signed
  hc = crackle_amd.header(golden()["kat_4x4_c"])
  assert not hc.fortran_order


def test_header_crc8_fault_injection():
  # every single-bit flip in the protected bytes 5..27 must be detected
  # (reference: automated_test.py:731-826, crc8 Hamming distance)
  good = bytearray(golden()["kat_4x4"])
  for byte in range(5, 28):
    for bit in range(8):
      bad = bytearray(good)
      bad[byte] ^= (1 << bit)
      with pytest.raises(crackle_amd.FormatError):
        crackle_amd.header(bytes(bad))
  with pytest.raises(crackle_amd.FormatError):
    crackle_amd.header(b"nope" + bytes(good[4:]))
  with pytest.raises(crackle_amd.FormatError):
    crackle_amd.header(bytes(good[:10]))


def test_crc32c_host(port):
  L = _lib.lib()
  assert L.ckl_crc32c(b"123456789", 9) == 0xE3069283
  data = bytes(np.random.default_rng(0).integers(0, 256, 100003, dtype=np.uint8))
  assert L.ckl_crc32c(data, len(data)) == port.crc32c(data)
  # lengths around the switch to three interleaved chains and their remainders
  for n in (0, 1, 7, 8, 9, 3071, 3072, 3073, 3095, 3096, 24005):
    assert L.ckl_crc32c(data[:n], n) == port.crc32c(data[:n]), n


def test_crc32c_combine_host(port):
  """ckl_crc32c_combine: the label-section crc of a sharded encode is put together from the
  ranks' parts."""
  L = _lib.lib()
  rng = np.random.default_rng(5)
  parts = [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for n in (0, 1, 17, 4096, 100003, 0, 5)]
  crc = L.ckl_crc32c(parts[0], len(parts[0]))
  for p in parts[1:]:
    crc = L.ckl_crc32c_combine(crc, L.ckl_crc32c(p, len(p)), len(p))
  assert crc == port.crc32c(b"".join(parts))


def test_signed_input_rejected():
  with pytest.raises(TypeError):
    crackle_amd.compress(np.zeros((4, 4, 4), dtype=np.int32))


def test_empty_volume_is_header_only(port):
  for shape, kw in [((0, 0, 0), {}), ((5, 0, 3), dict(markov_model_order=3, allow_pins=1))]:
    arr = np.zeros(shape, np.uint32, order="F")
    b = crackle_amd.compress(arr, **kw)
    assert len(b) == 29
    assert b == port.compress(arr, allow_pins=bool(kw.get("allow_pins", 0)), markov_model_order=kw.get("markov_model_order", 0))
    assert crackle_amd.decompress(b).size == 0
  assert golden()["empty_000"] == crackle_amd.compress(np.zeros((0, 0, 0), np.uint8, order="F"))


def test_single_label_shortcut_needs_no_device():
  # crackle/codec.py:659-668: streams with one label never reach the native decoder
  out = crackle_amd.decompress(golden()["kat_ones_300"])
  assert out.shape == (300, 300, 2) and out.dtype == np.uint32 and (out == 1).all() and out.flags.f_contiguous
  out = crackle_amd.decompress(golden()["zeros_50"])
  assert out.shape == (50, 50, 5) and not out.any()
  assert crackle_amd.num_labels(golden()["c0_voronoi_u8"]) == len(np.unique(SMALL["c0_voronoi_u8"][0]))
  assert np.array_equal(crackle_amd.labels(golden()["c0_voronoi_u8_pins"]), np.unique(SMALL["c0_voronoi_u8"][0]))
  assert crackle_amd.contains(golden()["kat_4x4"], 7) and not crackle_amd.contains(golden()["kat_4x4"], 3)
  absent = crackle_amd.decompress(golden()["kat_4x4"], label=3)
  assert absent.dtype == bool and not absent.any()


def test_compute_refuses_to_run_without_a_device():
  if _lib.lib().ckl_device_count() > 0:
    pytest.skip("a HIP device is present")
  with pytest.raises(RuntimeError, match="no usable HIP device"):
    crackle_amd.compress(SMALL["kat_4x4"][0])
  with pytest.raises(RuntimeError, match="no usable HIP device"):
    crackle_amd.decompress(golden()["kat_4x4"])


def _zstack(bufs):
  n = len(bufs)
  arr = (C.c_char_p * n)(*bufs)
  lens = (C.c_uint64 * n)(*[len(b) for b in bufs])
  out, m = C.c_void_p(), C.c_uint64()
  rc = _lib.lib().ckl_zstack(arr, lens, n, C.byref(out), C.byref(m))
  if rc != 0:
    raise RuntimeError(_lib.last_error())
  try:
    return C.string_at(out.value, m.value)
  finally:
    _lib.lib().ckl_free(out)


@pytest.mark.parametrize("order", [0, 3])
def test_zstack_equals_whole_volume_compress(port, order):
  # reference property: zstack(compress(slabs)) == compress(whole) (automated_test.py:448-487).
  # Holds when the slabs agree on crack format / stored width / markov model; the slabs here
  # are encoded with the whole volume's model through the oracle's override hook.
  from crackle_amd import synth
  vol = synth.as_numpy_f(synth.voronoi_labels((96, 80, 12), np.uint16, seed=4, cell=(16, 16, 4)))
  whole = port.compress(vol, markov_model_order=order)
  if order == 0:
    slabs = [port.compress(np.asfortranarray(vol[:, :, a:b])) for a, b in ((0, 5), (5, 6), (6, 12))]
    assert _zstack(slabs) == whole
  else:
    # different slabs have different statistics -> different models -> refuse to merge
    slabs = [port.compress(np.asfortranarray(vol[:, :, a:b]), markov_model_order=order) for a, b in ((0, 6), (6, 12))]
    with pytest.raises(RuntimeError, match="markov"):
      _zstack(slabs)


def test_zstack_rejects_mismatched_slabs(port):
  a = port.compress(np.zeros((8, 8, 2), np.uint8, order="F") + np.arange(8, dtype=np.uint8)[:, None, None] // 4)
  b = port.compress(np.ones((8, 9, 2), np.uint8, order="F"))
  with pytest.raises(RuntimeError):
    _zstack([a, b])


def _forge(stream: bytes, num_label_bytes: int) -> bytes:
  """The stream with a forged num_label_bytes and a matching crc8 (crc.hpp:23-37: the header's
  own checksum is no protection against a hostile stream)."""
  b = bytearray(stream)
  b[20:28] = int(num_label_bytes & (2**64 - 1)).to_bytes(8, "little")
  crc = 0xFF
  for x in b[5:28]:
    crc ^= x
    for _ in range(8):
      crc = ((crc >> 1) ^ 0xE7) if (crc & 1) else (crc >> 1)
  b[28] = crc
  return bytes(b)


@pytest.mark.parametrize("nlb", [2**64 - 1, 2**64 - 29, 2**64 - 200, 2**63, 2**40, 10**6])
def test_forged_label_section_length_is_refused_by_the_host_parsers(nlb):
  """Offsets taken from the header must not be summed until each is known to lie inside the
  buffer: a num_label_bytes close to 2^64 made the old `a + b + c > n` checks wrap."""
  import crackle_amd.operations as ops
  good = golden()["c0_voronoi_u8"]
  bad = _forge(good, nlb)
  assert crackle_amd.header(bad).num_label_bytes == nlb      # the forged header itself parses
  with pytest.raises(ValueError, match="past end of buffer|malformed"):
    ops.zstack([bad, good])
  with pytest.raises(ValueError, match="past end of buffer|malformed"):
    ops.zsplit(bad, 3)
