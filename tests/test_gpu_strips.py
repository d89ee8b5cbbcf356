"""GPU tests of the decoder's strip path (crackle_amd/csrc/ckl_strips.hpp: k_strip_ccl,
k_slice_resolve, k_paint_strips) against the oracle, of its hand-over to the general run
pipeline when a strip or a slice overflows the LDS tables, of the z-chunked launches, and of the
BASELINE.json configurations at full slice size against the reference's own bytes
(tests/golden/manifest_xl.json, written by tests/gen_golden.py --xl from the compiled reference)."""
import json
import os

import numpy as np
import pytest

import crackle_amd
from crackle_amd import synth
import golden_cases
from util import sha

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _vol(shape, dt, seed, cell=(16, 16, 4), **kw):
  return synth.as_numpy_f(synth.voronoi_labels(shape, dt, seed=seed, cell=cell, **kw))


STRIP_SHAPES = [
  ((320, 288, 5), np.uint32, (16, 16, 4)),      # several strips of 102 rows, tiles of 12 rows
  ((1024, 96, 3), np.uint16, (32, 32, 8)),      # 32 words per row: strips of 32 rows
  ((64, 64, 16), np.uint8, (8, 8, 4)),          # one strip per slice
  ((2048, 40, 2), np.uint32, (32, 32, 8)),      # 64 words per row, strips of 16 rows
  ((4100, 8, 2), np.uint32, (32, 32, 8)),       # rows wider than a paint tile (one row per tile)
  ((36, 300, 3), np.uint64, (8, 8, 4)),         # two words per row, ragged last word, 8-byte labels
  ((4, 4, 2), np.uint8, (2, 2, 1)),
]


@pytest.mark.parametrize("shape,dt,cell", STRIP_SHAPES)
def test_strip_path_matches_general_path_and_oracle(shape, dt, cell, checker, monkeypatch):
  arr = _vol(shape, dt, seed=31, cell=cell, offset=(1 << 40) if dt == np.uint64 else 0)
  for kw in (dict(markov_model_order=0), dict(markov_model_order=3), dict(allow_pins=True)):
    b = checker.compress(arr, **kw)
    monkeypatch.delenv("CKL_DECODE_GENERAL", raising=False)
    got = crackle_amd.decompress(b)
    assert np.array_equal(got, arr), f"strip path {shape} {kw}"
    lbl = int(arr[shape[0] // 2, shape[1] // 2, 0])
    img = crackle_amd.decompress(b, label=lbl)
    assert np.array_equal(img, arr == lbl), f"strip path label= {shape} {kw}"
    if shape[2] > 2:
      part = crackle_amd.decompress_range(b, 1, shape[2] - 1)
      assert np.array_equal(part, arr[:, :, 1:shape[2] - 1]), f"strip path z-range {shape} {kw}"
    monkeypatch.setenv("CKL_DECODE_GENERAL", "1")
    assert np.array_equal(crackle_amd.decompress(b), arr), f"general path {shape} {kw}"
    monkeypatch.delenv("CKL_DECODE_GENERAL", raising=False)
    # the strip path behind the rasterising kernel (k_decode_cracks) instead of the crack records
    monkeypatch.setenv("CKL_DECODE_RASTER", "1")
    assert np.array_equal(crackle_amd.decompress(b), arr), f"raster front end {shape} {kw}"
    monkeypatch.delenv("CKL_DECODE_RASTER", raising=False)
    # record lists too small: the session hands over to the rasterising kernel
    monkeypatch.setenv("CKL_REC_CAP", "3")
    assert np.array_equal(crackle_amd.decompress(b), arr), f"record list overflow {shape} {kw}"
    monkeypatch.delenv("CKL_REC_CAP", raising=False)
    # control tables of k_crack_match in HBM
    monkeypatch.setenv("CKL_LDS_CONTROLS", "64")
    assert np.array_equal(crackle_amd.decompress(b), arr), f"global control tables {shape} {kw}"
    monkeypatch.delenv("CKL_LDS_CONTROLS", raising=False)
  monkeypatch.delenv("CKL_DECODE_GENERAL", raising=False)


def test_strip_overflow_hands_over_to_the_general_pipeline(checker, monkeypatch):
  # noise: every strip of 32 Ki pixels has far more than 3072 runs
  noise = synth.random_labels((256, 256, 4), np.uint32, seed=5, high=2000)
  b = checker.compress(noise)
  assert np.array_equal(crackle_amd.decompress(b), noise)
  bits = synth.random_labels((512, 64, 3), np.uint8, seed=6, high=2)
  b = checker.compress(bits, markov_model_order=2)
  assert np.array_equal(crackle_amd.decompress(b), bits)
  # a slice with more strip components than the resolve table (forced small)
  arr = _vol((320, 288, 5), np.uint32, seed=32)
  b = checker.compress(arr)
  monkeypatch.setenv("CKL_RESOLVE_CAP", "7")
  assert np.array_equal(crackle_amd.decompress(b), arr)
  b = checker.compress(arr, allow_pins=True)
  assert np.array_equal(crackle_amd.decompress(b), arr)


@pytest.mark.parametrize("chunks", ["1", "2", "3", "8"])
def test_z_chunked_launches(chunks, checker, monkeypatch):
  monkeypatch.setenv("CKL_DECODE_CHUNKS", chunks)
  arr = _vol((256, 192, 11), np.uint32, seed=33)
  for kw in (dict(), dict(markov_model_order=4), dict(allow_pins=True)):
    b = checker.compress(arr, **kw)
    assert np.array_equal(crackle_amd.decompress(b), arr), f"chunks={chunks} {kw}"
    assert np.array_equal(crackle_amd.decompress_range(b, 2, 9), arr[:, :, 2:9])


def test_strip_path_reports_corruption(checker):
  arr = _vol((256, 192, 6), np.uint32, seed=34)
  good = checker.compress(arr)
  bad = bytearray(good)
  bad[-5] ^= 0x10      # crc32c of a slice's component image
  with pytest.raises(RuntimeError, match="crc"):
    crackle_amd.decompress(bytes(bad))
  # a flipped crack code bit changes the components: count or crc must object
  info = crackle_amd.header(good)
  hb = 29 + 4 * (6 + 1) + int(info.num_label_bytes)
  hits = 0
  for off in range(60, 400, 37):
    bad = bytearray(good)
    bad[hb + off] ^= 0x04
    try:
      out = crackle_amd.decompress(bytes(bad))
      hits += int(np.array_equal(out, arr))
    except RuntimeError:
      hits += 1
  assert hits == len(range(60, 400, 37))      # every damaged stream is either refused or (index bytes) decodes identically


def _manifest_xl():
  with open(os.path.join(HERE, "golden", "manifest_xl.json")) as f:
    return json.load(f)


@pytest.mark.parametrize("name", ["c3_1024x1024x16_u64", "c4_2048x2048x8_u32_m5", "c4_2048x2048x8_u32_pins_m5"])
def test_baseline_configs_at_full_slice_size(name):
  """C3 (uint64, stored width 8) and C4 (2048 x 2048, pins + markov order 5): bytes against the
  reference encoder's sha256, then decoded back."""
  thunk, kw = golden_cases.xl_cases()[name]
  arr = thunk()
  want = _manifest_xl()[name]
  got = crackle_amd.compress(arr, allow_pins=int(kw["allow_pins"]), markov_model_order=kw["markov_model_order"])
  assert len(got) == want["length"] and sha(got) == want["sha256"], name
  assert np.array_equal(crackle_amd.decompress(got), arr)


def test_c2_full_size_bytes_against_the_reference():
  """BASELINE.json configs[2] at its full size, device resident: the encoder's stream has the
  sha256 of the reference encoder's output; the decoder returns the volume."""
  import torch
  from crackle_amd import distributed as ckd
  dev = torch.device("cuda:0")
  vol = synth.voronoi_labels((1024, 1024, 512), np.uint32, seed=2, device=dev)
  be = ckd.HipBackend(0)
  b = be.encode(vol, (1024, 1024, 512), False, True, 0, None)
  want = _manifest_xl()["c2_1024x1024x512_u32"]
  assert len(b) == want["length"] and sha(b) == want["sha256"]
  s = be.open_decoder(b, 0, 512)
  out = torch.empty_like(vol)
  s.run(out)
  torch.cuda.synchronize()
  assert torch.equal(out.view(torch.int32), vol.view(torch.int32))
  s.close()


def test_c3_full_size_bytes_against_the_reference():
  """BASELINE.json configs[3] whole (1024 x 1024 x 1024 uint64, labels from 2^40 up), device
  resident: the encoder's stream has the sha256 of the reference encoder's output (written into
  manifest_xl.json by `tests/gen_golden.py --xl --only=c3_1024x1024x1024_u64`); decoded back."""
  import torch
  from crackle_amd import distributed as ckd
  want = _manifest_xl().get("c3_1024x1024x1024_u64")
  assert want is not None, "manifest_xl.json has no entry for the whole C3 volume"
  dev = torch.device("cuda:0")
  vol = synth.voronoi_labels((1024, 1024, 1024), np.uint64, seed=2, device=dev, offset=1 << 40)
  be = ckd.HipBackend(0, zero_copy=True)
  b = be.encode(vol, (1024, 1024, 1024), False, True, 0, None)
  assert len(b) == want["length"] and sha(b.view()) == want["sha256"]
  s = be.open_decoder(b, 0, 1024)
  out = torch.empty_like(vol)
  s.run(out)
  torch.cuda.synchronize()
  assert torch.equal(out, vol)
  s.close()


def test_c1_full_size_bytes_against_the_reference():
  """BASELINE.json configs[1] (512 x 512 x 128 uint32), device resident, against the reference's sha256."""
  import torch
  from crackle_amd import distributed as ckd
  want = _manifest_xl()["c1_512x512x128_u32"]
  dev = torch.device("cuda:0")
  vol = synth.voronoi_labels((512, 512, 128), np.uint32, seed=2, device=dev)
  be = ckd.HipBackend(0)
  b = be.encode(vol, (512, 512, 128), False, True, 0, None)
  assert len(b) == want["length"] and sha(b) == want["sha256"]
  assert np.array_equal(crackle_amd.decompress(b), synth.as_numpy_f(vol))
