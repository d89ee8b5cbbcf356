"""GPU tests of the library's behaviour on hostile or unusual inputs: forged header fields (the
crc8 is no protection), and encoder overrides that force nothing after ckl_encoder_stats cached the
label planes."""
import ctypes as C

import numpy as np
import pytest

import crackle_amd
from crackle_amd import _lib, synth
from util import golden
from test_abi_cpu import _forge

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nlb", [2**64 - 1, 2**64 - 29, 2**64 - 200, 2**63, 2**40, 10**6, 3])
def test_forged_label_section_length_is_refused(nlb):
  for name in ("c0_voronoi_u8", "c0_voronoi_u8_pins_m5"):
    bad = _forge(golden()[name], nlb)
    with pytest.raises(RuntimeError):
      crackle_amd.decompress(bad)
    with pytest.raises((RuntimeError, ValueError)):
      crackle_amd.reencode(bad, 3 if "m5" not in name else 0)
    with pytest.raises((RuntimeError, ValueError)):
      crackle_amd.voxel_connectivity_graph(bad, connectivity=4)
    with pytest.raises((RuntimeError, ValueError)):
      crackle_amd.voxel_counts(bad)


def test_forged_unique_count_and_component_counts_are_refused():
  good = golden()["c0_voronoi_u8"]
  off = 29 + 4 * 17      # header + z-index of 16 slices
  for value in (2**64 - 1, 2**63, 2**32, 10**9):
    bad = bytearray(good)
    bad[off:off + 8] = int(value).to_bytes(8, "little")      # num_unique of the flat label section
    with pytest.raises(RuntimeError):
      crackle_amd.decompress(bytes(bad))
  nu = int.from_bytes(good[off:off + 8], "little")
  comp = off + 8 + nu * 1      # components per slice, 2 bytes each (64 x 64 pixels)
  bad = bytearray(good)
  bad[comp:comp + 2] = (0xFFFF).to_bytes(2, "little")
  with pytest.raises(RuntimeError):
    crackle_amd.decompress(bytes(bad))


def test_overrides_that_force_nothing_use_the_cached_volume_statistics(checker):
  """ckl_encoder_stats leaves the label planes in the session; a following ckl_encoder_run with an
  override block that forces neither crack format nor stored width must still decide them from
  the volume (include/crackle_amd.h: -1 / 0 = "decide from this volume")."""
  import torch
  L = _lib.lib()
  dev = torch.device("cuda:0")
  for dt, mod in ((np.uint32, 70000), (np.uint16, 300)):
    vol = synth.voronoi_labels((96, 80, 6), dt, seed=51, cell=(16, 16, 4), modulus=mod, device=dev)
    arr = synth.as_numpy_f(vol)
    want = checker.compress(arr)
    enc = C.c_void_p()
    assert L.ckl_encoder_create(96, 80, 6, arr.itemsize, 0, C.byref(enc)) == 0
    mx, pairs, first, last = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    assert L.ckl_encoder_stats(enc, vol.data_ptr(), 96, 80, 6, C.byref(mx), C.byref(pairs), C.byref(first), C.byref(last)) == 0
    assert mx.value == int(arr.max())
    ov = _lib.EncodeOverrides()
    ov.force_crack_format = -1; ov.force_label_format = -1; ov.force_stored_width = 0; ov.has_model = 0
    out, n = C.c_void_p(), C.c_uint64()
    rc = L.ckl_encoder_run(enc, vol.data_ptr(), 96, 80, 6, 0, 1, 0, 0, 1, 0, C.byref(ov), C.byref(out), C.byref(n))
    assert rc == 0, _lib.last_error()
    got = C.string_at(out.value, n.value)
    L.ckl_free(out)
    L.ckl_encoder_destroy(enc)
    assert got == want, np.dtype(dt).name


@pytest.mark.gpu
def test_the_references_overflowed_pin_section_is_refused(checker):
  """The reference's pin encoder writes the count of a label's single-component ids in a field sized from the labels' pin
  counts (src/labels.hpp:209-229): 1024 x 248 x 4 uint8 voxels of 4 x 4 x 2 cells give a label 261 such ids beside fewer
  than 256 pins per label, the one-byte count overflows and the reference decodes its own stream to wrong labels
  (tools/repro_pins_u8.py).  The encoder here writes the same bytes (bit-exact); the decoder refuses them."""
  shape = (1024, 248, 4)
  arr = synth.as_numpy_f(synth.voronoi_labels(shape, np.uint8, seed=11, cell=(4, 4, 2)))
  want = checker.compress(arr, allow_pins=True)
  assert crackle_amd.compress(arr, allow_pins=True) == want
  with pytest.raises(RuntimeError, match="pin section is malformed"):
    crackle_amd.decompress(want)
  # one slice less and every count fits: the same geometry round-trips
  arr3 = np.asfortranarray(arr[:, :, :3])
  want3 = checker.compress(arr3, allow_pins=True)
  assert crackle_amd.compress(arr3, allow_pins=True) == want3
  assert np.array_equal(crackle_amd.decompress(want3), arr3)
