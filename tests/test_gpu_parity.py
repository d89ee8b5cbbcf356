"""GPU parity tests: the HIP path (through the C-ABI) against the golden fixtures
produced by the reference, against the oracle on seeded inputs, and — at sizes the
oracle cannot cover quickly — through size-independent properties.
Bit-exact: this is integer/byte work."""
import ctypes as C

import numpy as np
import pytest

import crackle_amd
from crackle_amd import _lib, synth
import golden_cases
from util import golden, manifest, label_format, sha

pytestmark = pytest.mark.gpu

SMALL = golden_cases.small_cases()
LARGE = golden_cases.large_cases()


def _kw(kw):
  return dict(allow_pins=int(kw["allow_pins"]), markov_model_order=kw["markov_model_order"])


def test_extension_loaded_and_device_present():
  assert _lib.lib().ckl_device_count() >= 1


@pytest.mark.parametrize("name", sorted(SMALL))
def test_decode_golden(name):
  arr, _ = SMALL[name]
  got = crackle_amd.decompress(golden()[name])
  if arr.size == 0:
    assert got.size == 0
    return
  assert got.dtype == arr.dtype and got.shape == arr.shape
  assert got.flags.f_contiguous == arr.flags.f_contiguous or arr.ndim < 2
  assert np.array_equal(got, arr)


@pytest.mark.parametrize("name", sorted(n for n in SMALL if label_format(golden()[n]) == 0))
def test_encode_golden_flat(name):
  arr, kw = SMALL[name]
  assert crackle_amd.compress(arr, **_kw(kw)) == golden()[name]


@pytest.mark.parametrize("name", sorted(n for n in SMALL if label_format(golden()[n]) == 2 and SMALL[n][0].size))
def test_encode_golden_pins(name):
  arr, kw = SMALL[name]
  assert crackle_amd.compress(arr, **_kw(kw)) == golden()[name]


@pytest.mark.parametrize("name", sorted(LARGE))
def test_large_manifest(name):
  thunk, kw = LARGE[name]
  arr = thunk()
  m = manifest()[name]
  b = crackle_amd.compress(arr, **_kw(kw))
  assert len(b) == m["length"] and sha(b) == m["sha256"]
  assert np.array_equal(crackle_amd.decompress(b), arr)


def test_z_range_label_and_c_order():
  arr, _ = SMALL["c0_voronoi_u8"]
  for key in ("c0_voronoi_u8", "c0_voronoi_u8_pins", "c0_voronoi_u8_m5", "c0_voronoi_u8_pins_m5"):
    b = golden()[key]
    got = crackle_amd.decompress_range(b, 3, 9)
    assert got.shape == (64, 64, 6) and np.array_equal(got, arr[:, :, 3:9])
    lbl = int(arr[10, 10, 5])
    img = crackle_amd.decompress(b, label=lbl)
    assert img.dtype == bool and np.array_equal(img, arr == lbl)
  c = crackle_amd.decompress(golden()["c0_voronoi_u8_c"])
  assert c.flags.c_contiguous and np.array_equal(c, arr)


def test_oracle_agreement_on_seeded_volumes(checker):
  cases = [
    ((200, 150, 9), np.uint8, 5, (16, 16, 4), dict(markov_model_order=0)),
    ((333, 257, 5), np.uint16, 6, (32, 32, 8), dict(markov_model_order=4)),
    ((512, 512, 6), np.uint32, 7, (32, 32, 8), dict(markov_model_order=0)),
    ((256, 256, 8), np.uint64, 8, (32, 32, 8), dict(markov_model_order=7)),
  ]
  for shape, dt, seed, cell, kw in cases:
    arr = synth.as_numpy_f(synth.voronoi_labels(shape, dt, seed=seed, cell=cell, offset=(1 << 40) if dt == np.uint64 else 0))
    want = checker.compress(arr, parallel=4, **kw)
    got = crackle_amd.compress(arr, **kw)
    assert got == want, f"{shape} {dt} {kw}"
    assert np.array_equal(crackle_amd.decompress(want), arr)
  # pin label encoding (device CCL + host cover), auto and manual background colour
  for shape, dt, seed, cell in [((96, 80, 24), np.uint8, 11, (16, 16, 8)), ((130, 70, 33), np.uint32, 12, (20, 20, 10)), ((64, 64, 40), np.uint64, 13, (16, 16, 16))]:
    arr = synth.as_numpy_f(synth.voronoi_labels(shape, dt, seed=seed, cell=cell))
    for kw in (dict(), dict(markov_model_order=3), dict(bgcolor=2)):
      okw = dict(kw)
      if "bgcolor" in okw:
        okw.update(auto_bgcolor=False, manual_bgcolor=okw.pop("bgcolor"))
      want = checker.compress(arr, allow_pins=True, **okw)
      assert label_format(want) == 2
      assert crackle_amd.compress(arr, allow_pins=1, **kw) == want, f"pins {shape} {dt} {kw}"
      assert np.array_equal(crackle_amd.decompress(want), arr)
  noise = synth.random_labels((128, 128, 3), np.uint32, seed=9, high=2000)
  assert crackle_amd.compress(noise) == checker.compress(noise)
  bits = synth.random_labels((96, 96, 2), np.uint8, seed=10, high=2)
  assert crackle_amd.compress(bits, markov_model_order=2) == checker.compress(bits, markov_model_order=2)


def test_corruption_is_reported():
  b = bytearray(golden()["c0_voronoi_u8"])
  b[-1] ^= 0xFF   # last slice's crc32c
  with pytest.raises(RuntimeError, match="crc"):
    crackle_amd.decompress(bytes(b))
  b = bytearray(golden()["c0_voronoi_u8"])
  b[29] ^= 0x01   # z-index
  with pytest.raises(RuntimeError):
    crackle_amd.decompress(bytes(b))
  with pytest.raises(crackle_amd.FormatError):
    crackle_amd.decompress(b"crkl" + b"\x07" + bytes(40))   # unknown format version
  with pytest.raises(crackle_amd.FormatError):
    crackle_amd.decompress(b"nope" + golden()["c0_voronoi_u8"][4:])
  trunc = golden()["c0_voronoi_u8"][:200]
  with pytest.raises(RuntimeError):
    crackle_amd.decompress(trunc)


def test_device_resident_sessions_full_size_properties():
  """1024x1024x32 uint32 entirely on device: encode -> decode round trip equality,
  determinism, and agreement between the per-slice crc32c the encoder stored and the
  one the decoder recomputes (a checksum of checksums over 32 slices)."""
  import torch
  L = _lib.lib()
  dev = torch.device("cuda:0")
  vol = synth.voronoi_labels((1024, 1024, 32), np.uint32, seed=2, device=dev)
  sz, sy, sx = vol.shape
  enc = C.c_void_p()
  assert L.ckl_encoder_create(sx, sy, sz, 4, 0, C.byref(enc)) == 0
  outs = []
  for _ in range(2):
    out, n = C.c_void_p(), C.c_uint64()
    rc = L.ckl_encoder_run(enc, vol.data_ptr(), sx, sy, sz, 0, 1, 0, 0, 1, 0, None, C.byref(out), C.byref(n))
    assert rc == 0, _lib.last_error()
    outs.append(C.string_at(out.value, n.value))
    L.ckl_free(out)
  L.ckl_encoder_destroy(enc)
  assert outs[0] == outs[1]
  binary = outs[0]
  assert len(binary) < vol.numel() * 4 // 50
  dec = C.c_void_p()
  assert L.ckl_decoder_create(binary, len(binary), 0, -1, 0, C.byref(dec)) == 0, _lib.last_error()
  back = torch.empty_like(vol)
  for _ in range(2):
    back.zero_()
    torch.cuda.synchronize()   # the session runs on its own stream: the buffer must be ready before the call
    assert L.ckl_decoder_run(dec, back.data_ptr(), back.numel() * 4, 0, 0) == 0, _lib.last_error()
    torch.cuda.synchronize()
    assert torch.equal(back.view(torch.int32), vol.view(torch.int32))
  L.ckl_decoder_destroy(dec)


@pytest.mark.parametrize("shape,dtype,cell,markov", [
  ((8, 8, 2), np.uint8, (4, 4, 1), 0),            # a label section of a few bytes: one workgroup, most of the frame is padding
  ((36, 20, 3), np.uint16, (6, 6, 2), 0),
  ((100, 60, 5), np.uint64, (5, 5, 2), 0),        # wide labels, a section whose length is no multiple of anything
  ((256, 256, 8), np.uint32, (8, 8, 2), 3),       # with a markov model between the label section and the codes
  ((512, 512, 16), np.uint32, (4, 4, 2), 0),      # 130 k labels: several workgroups of pieces
])
def test_label_section_crc_on_device(checker, shape, dtype, cell, markov):
  """Streams that stay in HBM take the label section's crc32c on the device (k_crc32c_pieces / k_crc32c_fold) and bring the
  section to the host in the background: the host bytes after the wait are the reference's, for sections from a few bytes on."""
  import torch
  from crackle_amd import distributed as ckd
  dev = torch.device("cuda:0")
  be = ckd.HipBackend(0, zero_copy=True)
  vol = synth.voronoi_labels(shape, dtype, seed=91, device=dev, cell=cell)
  want = checker.compress(synth.as_numpy_f(vol), markov_model_order=markov)
  item = vol.element_size()
  be.keep_device_stream(shape, item, True)
  be.async_host_copy(shape, item, True)
  try:
    for _ in range(2):
      stream = be.encode(vol, shape, False, True, markov, None)
      be.host_wait()
      assert bytes(stream) == want
  finally:
    be.async_host_copy(shape, item, False)
    be.keep_device_stream(shape, item, False)


def test_encoder_async_host_copy(checker):
  """ckl_encoder_async_host_copy / ckl_encoder_host_wait: the call returns with the stream complete in HBM
  (a decoder runs from it at once), the host buffer is complete after the wait, and the next run waits for
  a copy still in flight by itself."""
  import torch
  from crackle_amd import distributed as ckd
  dev = torch.device("cuda:0")
  be = ckd.HipBackend(0, zero_copy=True)
  shape = (512, 384, 24)
  vols = [synth.voronoi_labels(shape, np.uint32, seed=s, device=dev, cell=(16, 16, 4)) for s in (81, 82)]
  want = [checker.compress(synth.as_numpy_f(v)) for v in vols]
  be.keep_device_stream(shape, 4, True)
  be.async_host_copy(shape, 4, True)
  out = torch.empty_like(vols[0])
  for it in range(3):
    v, w = vols[it % 2], want[it % 2]
    stream = be.encode(v, shape, False, True, 0, None)
    s = be.open_decoder(be.device_stream(), 0, shape[2])      # before the host bytes are there
    s.run(out)
    s.close()
    assert torch.equal(out, v)
    if it != 1:      # the second round leaves the wait to the next run
      be.host_wait()
      assert bytes(stream) == w
  be.host_wait()
  be.async_host_copy(shape, 4, False)
  assert bytes(be.encode(vols[0], shape, False, True, 0, None)) == want[0]


def test_decoder_from_device_resident_stream(checker):
  """ckl_encoder_keep_device_stream / ckl_encoder_device_stream / ckl_decoder_create_device: the encoder's
  HBM copy of its stream is byte for byte the host stream, and a decoder session created from it (no
  upload; header, z-index, label head and crcs read back) decodes the volume — flat, markov, pins,
  uint64 labels, z-ranges, and a stream a caller put into HBM itself."""
  import torch
  from crackle_amd import distributed as ckd
  hip = C.CDLL("libamdhip64.so")
  dev = torch.device("cuda:0")
  be = ckd.HipBackend(0)
  cases = [
    ((160, 128, 6), np.uint32, dict(), 0),
    ((160, 128, 6), np.uint32, dict(markov_model_order=4), 0),
    ((96, 80, 7), np.uint16, dict(allow_pins=True), 0),
    ((64, 64, 5), np.uint64, dict(), 1 << 40),
    ((1024, 1024, 3), np.uint32, dict(), 0),
  ]
  for shape, dt, kw, offset in cases:
    vol = synth.voronoi_labels(shape, dt, seed=71, device=dev, cell=(16, 16, 4), offset=offset)
    sx, sy, sz = shape
    be.keep_device_stream(shape, vol.element_size(), True)
    host = bytes(be.encode(vol, shape, bool(kw.get("allow_pins")), True, int(kw.get("markov_model_order", 0)), None))
    assert host == checker.compress(synth.as_numpy_f(vol), **kw), (shape, kw)
    ds = be.device_stream()
    assert len(ds) == len(host)
    back = np.zeros(len(host), dtype=np.uint8)
    assert hip.hipMemcpy(C.c_void_p(back.ctypes.data), C.c_void_p(ds.ptr), C.c_size_t(len(host)), 2) == 0
    assert back.tobytes() == host, (shape, kw)
    out = torch.empty_like(vol)
    s = be.open_decoder(ds, 0, sz)
    s.run(out)
    s.close()
    assert torch.equal(out, vol), (shape, kw)
    if sz > 3:
      part = torch.empty_like(vol[1:sz - 1])
      s = be.open_decoder(ds, 1, sz - 1)
      s.run(part)
      s.close()
      assert torch.equal(part, vol[1:sz - 1]), (shape, kw)
    # a stream the caller uploaded itself
    mine = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(dev)
    s = be.open_decoder(ckd.DeviceStream(mine.data_ptr(), mine.numel(), owner=mine), 0, sz)
    out.zero_()
    s.run(out)
    s.close()
    assert torch.equal(out, vol), (shape, kw)
    be.keep_device_stream(shape, vol.element_size(), False)
  # a corrupted resident stream is refused like a host one
  bad = bytearray(host)
  bad[30] ^= 0x40      # z-index
  mine = torch.frombuffer(bad, dtype=torch.uint8).to(dev)
  with pytest.raises(RuntimeError):
    be.open_decoder(ckd.DeviceStream(mine.data_ptr(), mine.numel(), owner=mine), 0, sz)


def test_large_slices_and_dense_graphs(checker):
  """Slices of the C4 shape (2048 x 2048: node tables of the trail near the LDS limit),
  a PERMISSIBLE volume big enough to leave the LDS tables (dense crack graph, loops,
  dead ends everywhere) and binary noise; bytes against the oracle."""
  arr = synth.as_numpy_f(synth.voronoi_labels((2048, 2048, 3), np.uint32, seed=21, cell=(32, 32, 8)))
  for kw in (dict(markov_model_order=0), dict(markov_model_order=5)):
    want = checker.compress(arr, parallel=8, **kw)
    assert crackle_amd.compress(arr, **kw) == want, f"2048x2048x3 {kw}"
  assert np.array_equal(crackle_amd.decompress(want), arr)
  small_cells = synth.as_numpy_f(synth.voronoi_labels((1024, 768, 2), np.uint16, seed=22, cell=(6, 6, 2)))
  assert crackle_amd.compress(small_cells) == checker.compress(small_cells, parallel=8)
  noise = synth.random_labels((512, 384, 2), np.uint32, seed=23, high=2000)
  want = checker.compress(noise, parallel=8)
  assert crackle_amd.compress(noise) == want
  assert np.array_equal(crackle_amd.decompress(want), noise)
  bits = synth.random_labels((640, 512, 2), np.uint8, seed=24, high=2)
  for kw in (dict(), dict(markov_model_order=3)):
    want = checker.compress(bits, parallel=8, **kw)
    assert crackle_amd.compress(bits, **kw) == want, f"binary noise {kw}"
  assert np.array_equal(crackle_amd.decompress(want), bits)


@pytest.mark.parametrize("env", [{"CKL_TRAIL_LDS": "4096"}, {"CKL_TRAIL_WALK": "plain"}, {"CKL_TRAIL_WALK": "lds"}, {"CKL_TRAIL_WALK": "lds", "CKL_TRAIL_WALK_STACK": "8"}, {"CKL_BITONIC_SINGLE_STEPS": "1"}, {"CKL_TRAIL_WALK_STACK": "8"}, {"CKL_LABELS_AT_WALK": "0"}, {"CKL_PLANES_GENERIC": "1"}, {"CKL_NO_OVERLAP": "1"}, {"CKL_SMALL_ESTIMATE": "1"}, {"CKL_STRIP_RUNS": "64"}])
def test_encoder_fallback_paths(env, checker, monkeypatch):
  """The encoder's alternate code paths (node tables / union-find in global memory instead
  of LDS, the compiled walk instead of the hand-scheduled one, a walk that starts over because its
  branch stack ran out, the label stream in front of the graph kernel instead of beside the walk, the
  generic label-plane kernel, no stream overlap) produce the same bytes."""
  for k, v in env.items():
    monkeypatch.setenv(k, v)
  arr = synth.as_numpy_f(synth.voronoi_labels((320, 288, 5), np.uint32, seed=31, cell=(16, 16, 4)))
  for kw in (dict(markov_model_order=0), dict(markov_model_order=4), dict(allow_pins=1)):
    okw = dict(kw)
    if "allow_pins" in okw:
      okw["allow_pins"] = True
    assert crackle_amd.compress(arr, **kw) == checker.compress(arr, **okw), f"{env} {kw}"
  noise = synth.random_labels((160, 144, 3), np.uint16, seed=32, high=300)
  assert crackle_amd.compress(noise) == checker.compress(noise), f"{env} noise"
  for name in ("c0_voronoi_u8", "c0_voronoi_u8_m5"):
    arr8, kw8 = SMALL[name]
    assert crackle_amd.compress(arr8, **_kw(kw8)) == golden()[name]


@pytest.mark.parametrize("env", [{"CKL_MARKOV_SERIAL": "1"}, {"CKL_LDS_CONTROLS": "64"}, {"CKL_NO_LDS_RASTER": "1"}, {"CKL_STRIP_RUNS": "64"}, {"CKL_PAINT_NOSTAGE": "1"}])
def test_decoder_fallback_paths(env, checker, monkeypatch):
  """The decoder's alternate code paths (one-thread markov expansion, control tables in HBM,
  rasterisation with HBM atomics) decode the same volumes."""
  for k, v in env.items():
    monkeypatch.setenv(k, v)
  arr = synth.as_numpy_f(synth.voronoi_labels((320, 288, 5), np.uint32, seed=41, cell=(16, 16, 4)))
  for kw in (dict(markov_model_order=0), dict(markov_model_order=2), dict(markov_model_order=6), dict(allow_pins=True, markov_model_order=3)):
    b = checker.compress(arr, **kw)
    assert np.array_equal(crackle_amd.decompress(b), arr), f"{env} {kw}"
  for name in sorted(n for n in SMALL if "_m" in n)[:12]:
    arr8, _ = SMALL[name]
    got = crackle_amd.decompress(golden()[name])
    assert got.size == arr8.size and (arr8.size == 0 or np.array_equal(got, arr8)), name


def test_encoder_components_api(checker):
  """ckl_encoder_components (per-voxel component ids of a device-resident slab, numbered from
  id_base) against the oracle's cc3d restatement."""
  import torch
  L = _lib.lib()
  dev = torch.device("cuda:0")
  for shape, dt in (((96, 80, 7), np.uint32), ((130, 67, 5), np.uint16)):
    vol = synth.voronoi_labels(shape, dt, seed=51, device=dev, cell=(16, 16, 4))
    sz, sy, sx = vol.shape
    enc = C.c_void_p()
    assert L.ckl_encoder_create(sx, sy, sz, vol.element_size(), 0, C.byref(enc)) == 0
    cc = np.zeros(sx * sy * sz, dtype=np.uint32)
    nc = np.zeros(sz, dtype=np.uint32)
    torch.cuda.synchronize()
    assert L.ckl_encoder_components(enc, vol.data_ptr(), sx, sy, sz, 1000, cc.ctypes.data, nc.ctypes.data) == 0, _lib.last_error()
    L.ckl_encoder_destroy(enc)
    want_cc, want_per, _ = checker.connected_components(synth.as_numpy_f(vol))
    assert np.array_equal(nc, want_per.astype(np.uint32))
    assert np.array_equal(cc, want_cc.ravel(order="F").astype(np.uint32) + 1000)


def test_big_slice_markov_uses_global_scratch(checker):
  """A slice whose markov tables exceed the LDS (2048 x 2048, ~400 k codes) still expands in
  parallel, with payload and ranks in global scratch."""
  arr = synth.as_numpy_f(synth.voronoi_labels((2048, 2048, 2), np.uint32, seed=61, cell=(32, 32, 8)))
  for order in (1, 5):
    b = checker.compress(arr, parallel=8, markov_model_order=order)
    assert np.array_equal(crackle_amd.decompress(b), arr), order


def test_corrupted_streams_return_an_answer(checker):
  """Bit flips anywhere behind the header (z-index, labels, model, crack codes, crcs): every
  decode returns a volume or raises; no fault and no hang (tools/fuzz_decode.py runs more)."""
  rng = np.random.default_rng(17)
  vol = synth.as_numpy_f(synth.voronoi_labels((160, 128, 3), np.uint16, seed=81, cell=(16, 16, 3)))
  streams = [golden()["c0_voronoi_u8"], golden()["c0_voronoi_u8_pins_m5"],
             checker.compress(vol), checker.compress(vol, markov_model_order=4)]
  outcomes = 0
  for t in range(120):
    b = bytearray(streams[t % len(streams)])
    for _ in range(int(rng.integers(1, 4))):
      b[int(rng.integers(29, len(b)))] ^= 1 << int(rng.integers(0, 8))
    try:
      crackle_amd.decompress(bytes(b))
    except (RuntimeError, ValueError, crackle_amd.FormatError):
      pass
    outcomes += 1
  assert outcomes == 120


def _numpy_label_stats(arr):
  """What the reference's per-pixel loops compute (operations.hpp:321-618), with numpy."""
  arr = np.asarray(arr)
  mask = (1 << (8 * arr.dtype.itemsize)) - 1
  flat = arr.ravel(order="F")
  uniq, inv, cts = np.unique(flat, return_inverse=True, return_counts=True)
  sx, sy, sz = arr.shape
  idx = np.arange(flat.size, dtype=np.int64)
  coords = (idx % sx, (idx // sx) % sy, idx // (sx * sy))
  # float64 weights are exact here: every sum stays far below 2**53
  sums = np.stack([np.bincount(inv, weights=c.astype(np.float64), minlength=uniq.size) for c in coords], axis=1)
  counts, cents, boxes = {}, {}, {}
  for k, u in enumerate(uniq):
    key = int(u) & mask
    counts[key] = int(cts[k])
    cents[key] = sums[k] / float(cts[k])
    sel = inv == k
    boxes[key] = tuple(int(np.min(c[sel])) for c in coords) + tuple(int(np.max(c[sel])) for c in coords)
  return counts, cents, boxes


def _reference_box_quirk(binary, boxes):
  """The reference seeds its box map from the unique list only (operations.hpp:561-567): a pin
  stream's background colour outside that list starts from a default-constructed box, so its three
  minima are 0 (tests/golden/label_stats.json pins the same behaviour to the reference itself)."""
  if label_format(binary) == 0:
    return boxes
  sz = int.from_bytes(binary[15:19], "little")
  sw = 1 << ((int.from_bytes(binary[5:7], "little") >> 2) & 3)
  lb = binary[29 + 4 * (sz + 1):]
  bg = int.from_bytes(lb[:sw], "little")
  n = int.from_bytes(lb[sw:sw + 8], "little")
  uniq = {int.from_bytes(lb[sw + 8 + i * sw: sw + 8 + (i + 1) * sw], "little") for i in range(n)}
  if bg not in uniq and bg in boxes:
    boxes = dict(boxes)
    boxes[bg] = (0, 0, 0) + tuple(boxes[bg][3:])
  return boxes


def test_label_statistics_match_numpy(checker):
  """voxel_counts / centroids / bounding_boxes (codec.py:949-1067) from the device runs."""
  cases = [
    (synth.as_numpy_f(synth.voronoi_labels((96, 80, 12), np.uint16, seed=4, cell=(16, 16, 4))), dict()),
    (synth.as_numpy_f(synth.voronoi_labels((130, 70, 33), np.uint32, seed=12, cell=(20, 20, 10))), dict(allow_pins=True)),
    (synth.as_numpy_f(synth.voronoi_labels((64, 64, 10), np.uint64, seed=8, cell=(16, 16, 4), offset=1 << 40)), dict(markov_model_order=3)),
    (synth.random_labels((50, 41, 3), np.uint32, seed=9, high=200), dict()),
    (SMALL["c0_voronoi_u8"][0], dict(allow_pins=True, markov_model_order=5)),
  ]
  for arr, kw in cases:
    binary = checker.compress(arr, **kw)
    counts, cents, boxes = _numpy_label_stats(arr)
    boxes = _reference_box_quirk(binary, boxes)
    assert crackle_amd.voxel_counts(binary) == counts
    got_c = crackle_amd.centroids(binary)
    assert sorted(got_c) == sorted(cents)
    for k in cents:
      assert got_c[k].dtype == np.float64 and np.array_equal(got_c[k], cents[k]), (k, got_c[k], cents[k])
    got_b = crackle_amd.bounding_boxes(binary, no_slice_conversion=True)
    assert sorted(got_b) == sorted(boxes)
    for k in boxes:
      assert got_b[k].dtype == np.uint32 and tuple(int(v) for v in got_b[k]) == boxes[k], (k, got_b[k], boxes[k])
    some = next(iter(boxes))
    x0, y0, z0, x1, y1, z1 = boxes[some]
    assert crackle_amd.bounding_boxes(binary, label=some) == (slice(x0, x1 + 1), slice(y0, y1 + 1), slice(z0, z1 + 1))
    assert crackle_amd.voxel_counts(binary, label=some) == counts[some]
    assert np.array_equal(crackle_amd.centroids(binary, label=some), cents[some])
  with pytest.raises(ValueError):
    crackle_amd.voxel_counts(checker.compress(cases[0][0]), label=60000)
  # the reference's shortcuts for single-label streams (codec.py:968-973, 1038-1043)
  const = np.full((40, 30, 4), 5, np.uint8, order="F")
  b = checker.compress(const)
  assert crackle_amd.voxel_counts(b) == {5: 4800}
  assert tuple(crackle_amd.bounding_boxes(b, no_slice_conversion=True)[5]) == (0, 0, 0, 40, 30, 4)
  assert np.array_equal(crackle_amd.centroids(b)[5], [19.5, 14.5, 1.5])
  empty = checker.compress(np.zeros((0, 0, 0), np.uint8, order="F"))
  assert crackle_amd.voxel_counts(empty) == {} and crackle_amd.centroids(empty) == {} and crackle_amd.bounding_boxes(empty) == {}


def test_label_statistics_against_the_reference_fixture():
  """voxel_counts / centroids / bounding_boxes of every small golden stream (whole range and one
  slice) against digests of the compiled reference's own output (tests/golden/label_stats.json,
  tests/gen_golden.py --ops): counts exact, centroids bit for bit as float64, boxes with the
  reference's initial values for absent labels and its zero minima for a pin stream's background
  colour.  The z-ranges go through the fastcrackle module (src/fastcrackle.cpp:346-420 take them)."""
  import json
  import os
  from gen_golden import STATS_RANGES, stats_digest
  from crackle_amd import fastcrackle as m
  with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "label_stats.json")) as f:
    want = json.load(f)
  g = golden()
  assert sorted(want) == sorted(g)
  for name in sorted(g):
    for tag, (z0, z1) in STATS_RANGES.items():
      for fn in ("voxel_counts", "centroids", "bounding_boxes"):
        w = want[name][f"{fn}.{tag}"]
        try:
          got = stats_digest(getattr(m, fn)(g[name], z0, z1, 1))
        except RuntimeError as exc:
          got = "error: " + str(exc)
        assert got == w, (name, fn, tag)      # the error texts of the empty stream included
    # the Python surface (codec.py:949-1067) over the whole stream
    if not want[name]["voxel_counts.all"].startswith("error") and crackle_amd.num_labels(g[name]) > 1:
      assert stats_digest(crackle_amd.voxel_counts(g[name])) == want[name]["voxel_counts.all"], name
      assert stats_digest(crackle_amd.centroids(g[name])) == want[name]["centroids.all"], name
      assert stats_digest(crackle_amd.bounding_boxes(g[name], no_slice_conversion=True)) == want[name]["bounding_boxes.all"], name


def test_label_statistics_run_by_run_merge(checker, monkeypatch):
  """Slices with more components than the LDS accumulators hold merge run by run."""
  monkeypatch.setenv("CKL_STATS_LDS_COMPS", "8")
  arr = synth.as_numpy_f(synth.voronoi_labels((130, 70, 33), np.uint32, seed=12, cell=(20, 20, 10)))
  counts, cents, boxes = _numpy_label_stats(arr)
  for kw in (dict(), dict(allow_pins=True)):
    binary = checker.compress(arr, **kw)
    assert crackle_amd.voxel_counts(binary) == counts
    got_b = crackle_amd.bounding_boxes(binary, no_slice_conversion=True)
    assert {k: tuple(int(v) for v in b) for k, b in got_b.items()} == _reference_box_quirk(binary, boxes)
    got_c = crackle_amd.centroids(binary)
    assert all(np.array_equal(got_c[k], cents[k]) for k in cents)


def test_label_statistics_full_size_checksums(checker):
  """At a large size the statistics are checked through what they must add up to: the
  counts to the voxel total, the coordinate sums to those of the whole grid, the union of
  the boxes to the volume."""
  arr = synth.as_numpy_f(synth.voronoi_labels((512, 512, 64), np.uint32, seed=21, cell=(32, 32, 16)))
  binary = crackle_amd.compress(arr)
  sx, sy, sz = arr.shape
  vc = crackle_amd.voxel_counts(binary)
  assert sum(vc.values()) == arr.size
  u, c = np.unique(arr, return_counts=True)
  assert vc == {int(a): int(b) for a, b in zip(u, c)}
  cents = crackle_amd.centroids(binary)
  total = sum(np.asarray(cents[k]) * vc[k] for k in vc)
  want = np.array([(sx - 1) / 2, (sy - 1) / 2, (sz - 1) / 2]) * arr.size
  assert np.allclose(total, want, rtol=1e-12)
  bb = crackle_amd.bounding_boxes(binary, no_slice_conversion=True)
  lo = np.min([b[:3] for b in bb.values()], axis=0)
  hi = np.max([b[3:] for b in bb.values()], axis=0)
  assert tuple(lo) == (0, 0, 0) and tuple(hi) == (sx - 1, sy - 1, sz - 1)


def test_reencode_on_device(checker):
  """crackle.reencode (codec.py:877-881): golden fixtures of one volume at two markov orders must
  turn into each other; seeded volumes are checked against the oracle's restatement of
  reencode_with_markov_order (crackle.hpp:858-984), flat and pin label sections alike."""
  g = golden()
  for m0, m5 in (("c0_voronoi_u8", "c0_voronoi_u8_m5"), ("c0_voronoi_u8_pins", "c0_voronoi_u8_pins_m5"), ("noise_2000", "noise_2000_m5")):
    assert crackle_amd.reencode(g[m5], 0) == g[m0], m5
    assert crackle_amd.reencode(g[m0], 5) == g[m5], m0
    assert crackle_amd.reencode(g[m0], 0) == g[m0]
  cases = [
    (synth.as_numpy_f(synth.voronoi_labels((200, 150, 9), np.uint8, seed=5, cell=(16, 16, 4))), dict()),
    (synth.as_numpy_f(synth.voronoi_labels((333, 257, 5), np.uint16, seed=6, cell=(32, 32, 8))), dict(markov_model_order=4)),
    (synth.as_numpy_f(synth.voronoi_labels((130, 70, 33), np.uint32, seed=12, cell=(20, 20, 10))), dict(allow_pins=True, markov_model_order=2)),
    (synth.random_labels((128, 128, 3), np.uint32, seed=9, high=2000), dict()),             # PERMISSIBLE crack format
    (np.full((40, 30, 4), 5, np.uint8, order="F"), dict()),                                  # no cracks at all
    (synth.as_numpy_f(synth.voronoi_labels((1100, 1060, 2), np.uint32, seed=17, cell=(8, 8, 2))), dict(markov_model_order=1)),   # multi-tile slices
  ]
  for arr, kw in cases:
    binary = checker.compress(arr, **kw)
    for order in (0, 1, 3, 6):
      want = checker.reencode(binary, order)
      got = crackle_amd.reencode(binary, order)
      assert got == want, (arr.shape, kw, order)
      assert np.array_equal(crackle_amd.decompress(got), arr)
  # zstack of streams with different models re-encodes them to order 0 first (operations.py:447-457)
  vol = synth.as_numpy_f(synth.voronoi_labels((96, 80, 12), np.uint16, seed=4, cell=(16, 16, 4)))
  a = checker.compress(np.asfortranarray(vol[:, :, :5]), markov_model_order=3)
  b = checker.compress(np.asfortranarray(vol[:, :, 5:]), markov_model_order=0)
  assert crackle_amd.zstack([a, b]) == checker.compress(vol)
  with pytest.raises(ValueError):
    crackle_amd.reencode(binary, 14)


def test_voxel_connectivity_graph(checker):
  """crackle.voxel_connectivity_graph (operations.py:936-954) against the oracle's restatement of
  operations.hpp:667-826: golden streams (flat, pins, markov, C order, PERMISSIBLE noise, single
  slices) and seeded volumes, connectivity 4 and 6."""
  g = golden()
  names = [n for n in sorted(g) if len(g[n]) >= 29 and SMALL.get(n, (np.zeros(0),))[0].size][::4]
  names += ["c0_voronoi_u8_c", "noise_2000", "c0_voronoi_u8_pins_m5"]
  for name in names:
    for conn in (4, 6):
      want = checker.voxel_connectivity_graph(g[name], conn)
      got = crackle_amd.voxel_connectivity_graph(g[name], conn)
      assert got.dtype == np.uint8 and got.shape == want.shape and got.flags.f_contiguous
      assert np.array_equal(got, want), (name, conn, int((got != want).sum()))
  for shape, dt, kw in [((200, 150, 9), np.uint8, dict()), ((333, 257, 5), np.uint16, dict(markov_model_order=4)), ((130, 70, 33), np.uint64, dict(allow_pins=True))]:
    arr = synth.as_numpy_f(synth.voronoi_labels(shape, dt, seed=5, cell=(16, 16, 4)))
    binary = checker.compress(arr, **kw)
    for conn in (4, 6):
      assert np.array_equal(crackle_amd.voxel_connectivity_graph(binary, conn), checker.voxel_connectivity_graph(binary, conn)), (shape, conn)
  # the +z bit is what numpy says about the labels
  v6 = crackle_amd.voxel_connectivity_graph(binary, 6)
  assert np.array_equal((v6[:, :, :-1] & 0x10) != 0, arr[:, :, :-1] == arr[:, :, 1:])
  with pytest.raises(ValueError):
    crackle_amd.voxel_connectivity_graph(binary, 8)


def test_structure_equal(checker):
  """structure_equal (operations.py:996-1021): encodings of one partition agree whatever the
  labels, markov order or label format; a moved boundary does not."""
  arr = synth.as_numpy_f(synth.voronoi_labels((96, 80, 12), np.uint16, seed=4, cell=(16, 16, 4)))
  a = checker.compress(arr)
  relabelled = checker.compress((arr.astype(np.uint32) * 7 + 3).astype(np.uint32))
  assert crackle_amd.structure_equal(a, checker.compress(arr, markov_model_order=3))
  assert crackle_amd.structure_equal(a, checker.compress(arr, allow_pins=True))
  assert crackle_amd.structure_equal(a, relabelled)
  moved = arr.copy(order="F")
  moved[10:14, 20:23, 5] = moved[9, 19, 5] + 1
  assert not crackle_amd.structure_equal(a, checker.compress(moved))
  assert not crackle_amd.structure_equal(a, checker.compress(np.asfortranarray(arr[:, :, :6])))
  assert crackle_amd.crack_crcs(a).dtype == np.uint32 and crackle_amd.crack_crcs(a).size == 12


def test_check_and_ok(checker):
  """crackle.check / crackle.ok (codec.py:883-948): damage is attributed to the section it hits."""
  arr = synth.as_numpy_f(synth.voronoi_labels((96, 80, 12), np.uint16, seed=4, cell=(16, 16, 4)))
  for kw in (dict(), dict(markov_model_order=3), dict(allow_pins=True)):
    good = checker.compress(arr, **kw)
    assert crackle_amd.ok(good)
    assert crackle_amd.check(good) == {"header": True, "crack_index": True, "labels": True, "z": []}
    head = crackle_amd.header(good)
    b = bytearray(good); b[8] ^= 1
    assert crackle_amd.check(bytes(b))["header"] is False and not crackle_amd.ok(bytes(b))
    b = bytearray(good); b[29] ^= 1                                   # first z-index entry
    assert crackle_amd.check(bytes(b))["crack_index"] is False
    b = bytearray(good); b[29 + 4 * 13 + 9] ^= 0x10                   # inside the label section
    rep = crackle_amd.check(bytes(b))
    assert rep["labels"] is False and not crackle_amd.ok(bytes(b))
    b = bytearray(good); b[-4 * 5 + 1] ^= 0x40                        # stored crc of slice 7
    assert crackle_amd.check(bytes(b))["z"] == [7]
    # a flipped bit in the crack codes of slice 3
    zidx = np.frombuffer(good, dtype="<u4", offset=29, count=12).astype(np.int64)
    start = 29 + 4 * 13 + head.num_label_bytes + head.markov_model_bytes + int(zidx[:3].sum())
    b = bytearray(good); b[start + int(zidx[3]) // 2] ^= 0x04
    rep = crackle_amd.check(bytes(b))
    assert rep["header"] and rep["crack_index"] and rep["labels"] and rep["z"] == [3], rep


def test_array_equal_on_device(checker):
  """ckl_array_equal against the oracle's restatement of operations::array_equal
  (src/operations.hpp:1039-1184, pinned to the compiled reference in tests/test_oracle.py),
  the reference's quirk included."""
  from crackle_amd import operations as ops
  L = _lib.lib()
  arr, _ = SMALL["c0_voronoi_u8"]
  g = golden()
  same = [g["c0_voronoi_u8"], g["c0_voronoi_u8_m5"], g["c0_voronoi_u8_pins"], g["c0_voronoi_u8_pins_m5"], g["c0_voronoi_u8_c"]]
  other = arr.copy(order="F"); other[10:14, 20:25, 3] = 251
  relabel = np.asfortranarray(((arr.astype(np.uint16) * 7 + 3) % 251).astype(np.uint8))
  moved = np.asfortranarray(np.roll(arr, 1, axis=0))
  cases = same + [checker.compress(other), checker.compress(relabel), checker.compress(moved), checker.compress(arr[:, :, :8].copy(order="F"))]
  def raw(a, b):
    eq = C.c_int(0)
    rc = L.ckl_array_equal(a, len(a), b, len(b), 0, C.byref(eq))
    assert rc == 0, _lib.last_error()
    return bool(eq.value)
  for a in cases[:7]:
    for b in cases:
      assert raw(a, b) == checker.array_equal(a, b)
  assert ops.array_equal(same[0], same[3]) and ops.array_equal(same[4], same[1])
  assert not ops.array_equal(same[0], cases[5]) and not ops.array_equal(same[0], cases[6])
  big = synth.as_numpy_f(synth.voronoi_labels((320, 288, 5), np.uint32, seed=61, cell=(16, 16, 4)))
  b0, b5 = checker.compress(big), checker.compress(big, markov_model_order=5, allow_pins=True)
  assert ops.array_equal(b0, b5)
  big2 = big.copy(order="F"); big2[100, 100, 2] += 1
  assert not ops.array_equal(b0, checker.compress(big2))


def test_mode_pooling_on_device(checker):
  """ckl_mode_pooling_2x2x1 against the oracle's restatement of operations::mode_pooling_2x2x1
  (src/operations.hpp:1201-1340): the per-slice streams byte for byte, and the stacked result."""
  from crackle_amd import operations as ops
  for name in ("c0_voronoi_u8", "c0_voronoi_u8_pins_m5", "c0_voronoi_u8_c", "rand_17x13x5_uint32_F_m0_p0", "rand_17x13x5_uint64_C_m2_p1", "rand_254x257x2_m0", "single_voxel", "row_33", "col_29"):
    b = golden()[name]
    want = checker.mode_pooling_2x2x1(b)
    got = ops._mode_pooling_slices(b)
    assert got == want, name
  b = golden()["c0_voronoi_u8"]
  assert ops._mode_pooling_slices(b, 2, 5) == checker.mode_pooling_2x2x1(b, 2, 5)
  arr, _ = SMALL["c0_voronoi_u8"]
  small = crackle_amd.decompress(ops.mode_pooling_2x2x1(b))
  assert small.shape == (32, 32, 16)
  a, bb, c, d = arr[0::2, 0::2], arr[1::2, 0::2], arr[0::2, 1::2], arr[1::2, 1::2]
  want = np.where(a == bb, a, np.where(a == c, a, np.where(bb == c, bb, d)))
  assert np.array_equal(small, want)


def test_label_table_hash_pass_and_full_sort(checker, monkeypatch):
  """The flat label table: distinct labels found by the hash pass before the sort (more than 8192
  components) or all component labels sorted (CKL_LABEL_SORT_ALL); the all-ones uint64 label is the
  hash table's empty marker and travels in a flag of its own."""
  vols = [
    synth.as_numpy_f(synth.voronoi_labels((512, 384, 4), np.uint32, seed=71, cell=(8, 8, 2))),                       # ~12 k components, ~6 k labels
    synth.as_numpy_f(synth.voronoi_labels((384, 256, 6), np.uint64, seed=72, cell=(6, 6, 3), offset=1 << 41)).copy(order="F"),
    synth.random_labels((200, 160, 3), np.uint16, seed=73, high=40000),                                               # nearly every component its own label
  ]
  vols[1][5:9, 7:11, 2] = np.uint64((1 << 64) - 1)
  vols[1][100:130, 90:95, 4] = np.uint64((1 << 64) - 2)
  for arr in vols:
    want = checker.compress(arr)
    assert crackle_amd.compress(arr) == want
    monkeypatch.setenv("CKL_LABEL_SORT_ALL", "1")
    assert crackle_amd.compress(arr) == want
    monkeypatch.delenv("CKL_LABEL_SORT_ALL")


def test_consumers_on_the_whole_c1_volume_against_the_reference_fixture():
  """reencode, mode pooling and the voxel connectivity graph of BASELINE.json configs[1] (512 x 512 x
  128 uint32) against digests of the compiled reference's own output (tests/golden/ops_xl.json,
  written by tests/gen_golden.py --ops)."""
  import hashlib
  import json
  import os
  import torch
  from crackle_amd import distributed as ckd
  with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ops_xl.json")) as f:
    want = json.load(f)["c1_512x512x128_u32"]
  vol = synth.voronoi_labels((512, 512, 128), np.uint32, seed=2, device=torch.device("cuda:0"))
  binary = bytes(ckd.HipBackend(0).encode(vol, (512, 512, 128), False, True, 0, None))
  h = lambda b: hashlib.sha256(b).hexdigest()
  m3 = crackle_amd.reencode(binary, 3)
  assert h(m3) == want["reencode_m3"]
  assert h(crackle_amd.reencode(m3, 0)) == want["reencode_m0_of_m3"] == h(binary)
  from crackle_amd import operations
  assert h(b"".join(operations._mode_pooling_slices(binary))) == want["mode_pooling_2x2x1"]
  for conn in (4, 6):
    assert h(np.ascontiguousarray(crackle_amd.voxel_connectivity_graph(binary, conn)).tobytes()) == want[f"vcg{conn}"], conn
  # voxel_counts / centroids / bounding_boxes (operations.hpp:321-665) of the same volume
  from gen_golden import STATS_RANGES, stats_digest
  from crackle_amd import fastcrackle as m
  for tag, (z0, z1) in STATS_RANGES.items():
    for fn in ("voxel_counts", "centroids", "bounding_boxes"):
      assert stats_digest(getattr(m, fn)(binary, z0, z1, 1)) == want[f"{fn}.{tag}"], (fn, tag)
