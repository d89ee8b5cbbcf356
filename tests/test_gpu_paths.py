"""Which kernels answered: every BASELINE.json configuration (one GPU's slab, full slice size) must be
decoded by the strip path's kernels (k_crack_match -> k_strip_ccl2 -> k_slice_resolve -> k_paint_strips,
crackle_amd/csrc/ckl_strips.hpp, ckl_strips2.hpp, ckl_crack_records.hpp) and not by a hand-over to the rasterising kernel or to
the general run pipeline — those are silent and session-sticky (ckl_decode.hip: decoder_run), which is how C4
once ran at 2 % of the roofline without a test noticing.  The stage names come from ckl_decoder_stage_timing,
the encoder's walk from ckl_encoder_walk_paths.  Outputs are checked as well (round trip on the device)."""
import numpy as np
import pytest
import torch

from crackle_amd import synth
from crackle_amd import distributed as ckd

pytestmark = pytest.mark.gpu

# k_strip_ccl2 (ckl_strips2.hpp) is the strip kernel for rows of a power of two of plane words: every shape below
FAST_FLAT = ["k_crack_match", "k_strip_ccl2", "k_slice_resolve", "k_paint_strips"]
FAST_PINS = ["k_crack_match", "k_strip_ccl2", "k_slice_resolve", "k_label_map_pins", "k_strip_labels", "k_paint_strips"]
SLOW = {"k_decode_cracks", "k_run_index", "k_run_union_strips", "k_run_assign", "k_paint_runs"}

# name, slab shape, dtype, encoder options, offset added to the labels
CONFIGS = [
  ("C1", (512, 512, 128), np.uint32, dict(), 0),
  ("C2 slab", (1024, 1024, 32), np.uint32, dict(), 0),
  ("C2 slab markov 5", (1024, 1024, 32), np.uint32, dict(markov_model_order=5), 0),
  ("C3 slab", (1024, 1024, 16), np.uint64, dict(), 1 << 40),
  ("C4 slab markov 5", (2048, 2048, 8), np.uint32, dict(markov_model_order=5), 0),
  ("C4 slab pins + markov 5", (2048, 2048, 8), np.uint32, dict(markov_model_order=5, allow_pins=True), 0),
]


@pytest.mark.parametrize("name,shape,dt,opts,offset", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_baseline_configurations_stay_on_the_fast_kernels(name, shape, dt, opts, offset):
  dev = torch.device("cuda:0")
  vol = synth.voronoi_labels(shape, dt, seed=2, device=dev, offset=offset)
  be = ckd.HipBackend(0)
  codec = ckd.ShardedCodec(be, device=dev)
  binary = codec.compress(vol, shape, **opts)
  fast, compiled = be.walk_paths()
  assert fast + compiled == shape[2], (name, fast, compiled)
  # (2048 x 2048 slices, 13 k nodes, take the hand-scheduled loop on the 9-byte tables: trail_walk_chain_wide)
  assert compiled == 0, f"{name}: {compiled} slices fell back to the compiled walk"
  out = torch.empty_like(vol)
  sess = codec.open_decoder(binary, shape)
  for _ in range(2):      # the hand-overs are sticky: a second run would show them too
    sess.run(out)
    names = [n for n, _ in sess.stages()]
    assert not (set(names) & SLOW), f"{name}: decoded by {names}"
    assert names == (FAST_PINS if opts.get("allow_pins") else FAST_FLAT), f"{name}: decoded by {names}"
  sess.close()
  assert torch.equal(out, vol), name


def test_fused_strip_kernel_is_bit_exact(monkeypatch):
  """k_strip_fused (opt-in: strips, resolve and paint in one ticketed launch) against the
  input, for 4- and 8-byte labels and a shape with ragged rows."""
  monkeypatch.setenv("CKL_DECODE_FUSED", "1")
  dev = torch.device("cuda:0")
  cases = [((1024, 1024, 24), np.uint32, 0, (32, 32, 8)), ((1024, 1024, 9), np.uint64, 1 << 40, (32, 32, 8)),
           ((320, 288, 5), np.uint16, 0, (16, 16, 4)), ((36, 300, 3), np.uint8, 0, (8, 8, 4))]
  for shape, dt, offset, cell in cases:
    vol = synth.voronoi_labels(shape, dt, seed=7, device=dev, offset=offset, cell=cell)
    be = ckd.HipBackend(0)
    codec = ckd.ShardedCodec(be, device=dev)
    binary = codec.compress(vol, shape)
    out = torch.empty_like(vol)
    sess = codec.open_decoder(binary, shape)
    sess.run(out)
    names = [n for n, _ in sess.stages()]
    sess.close()
    assert names == ["k_crack_match", "k_strip_fused"], names
    assert torch.equal(out, vol), (shape, dt)
  # a slice with more strip components than the resolver's LDS table holds: the session hands over to the three launches
  vol = synth.voronoi_labels((1024, 1024, 4), np.uint32, seed=8, device=dev, cell=(12, 12, 4))
  be = ckd.HipBackend(0)
  codec = ckd.ShardedCodec(be, device=dev)
  binary = codec.compress(vol, (1024, 1024, 4))
  out = torch.empty_like(vol)
  sess = codec.open_decoder(binary, (1024, 1024, 4))
  sess.run(out)
  assert "k_strip_fused" not in [n for n, _ in sess.stages()]
  sess.close()
  assert torch.equal(out, vol)


def test_stage_events_can_be_switched_off():
  """ckl_decoder_set_stage_events(0): the run records events around the pipeline only (bench.py's timed runs: the
  events between the kernels cost device time); the labels and the pipeline time are there, the stage table is empty,
  and a corrupted stream is still refused (the error words reach the host through k_slice_resolve)."""
  dev = torch.device("cuda:0")
  shape = (1024, 1024, 4)
  vol = synth.voronoi_labels(shape, np.uint32, seed=3, device=dev)
  be = ckd.HipBackend(0)
  codec = ckd.ShardedCodec(be, device=dev)
  binary = codec.compress(vol, shape)
  out = torch.empty_like(vol)
  sess = codec.open_decoder(binary, shape)
  sess.stage_events(False)
  sess.run(out)
  assert sess.stages() == []
  assert sess.timing()[0] > 0
  assert torch.equal(out, vol)
  sess.stage_events(True)
  sess.run(out)
  assert [n for n, _ in sess.stages()] == FAST_FLAT
  sess.close()
  bad = bytearray(bytes(binary))
  bad[len(bad) // 2] ^= 0x5A      # inside the crack codes of a middle slice
  sess = codec.open_decoder(bytes(bad), shape)
  sess.stage_events(False)
  with pytest.raises(RuntimeError):
    sess.run(out)
  sess.close()


@pytest.mark.parametrize("cell,dt,offset", [((8, 8, 4), np.uint64, 1 << 40), ((12, 12, 4), np.uint32, 0)], ids=["cell8x8x4-u64", "cell12x12x4-u32"])
def test_over_segmented_slices_stay_on_the_strip_kernels(cell, dt, offset):
  """The reference's second benchmark is a severely over-segmented uint64 watershed (benchmarks/README.md:284-318):
  cells of 8 x 8 give a 1024 x 1024 slice ~16 k segments and ~30 k strip components.  The decoder must keep such
  slices on k_crack_match -> k_strip_ccl2 -> k_slice_resolve -> k_paint_strips (k_slice_resolve with a CU's LDS to
  itself: ckl_decoder::resolve_cap_max), also when the resolver's table was sized too small at first (forced here
  with a decode whose first attempt overflows), instead of handing the session to the general run pipeline."""
  dev = torch.device("cuda:0")
  shape = (1024, 1024, 6)
  vol = synth.voronoi_labels(shape, dt, seed=11, device=dev, offset=offset, cell=cell)
  be = ckd.HipBackend(0)
  codec = ckd.ShardedCodec(be, device=dev)
  binary = codec.compress(vol, shape)
  out = torch.empty_like(vol)
  sess = codec.open_decoder(binary, shape)
  for _ in range(2):
    sess.run(out)
    names = [n for n, _ in sess.stages()]
    assert names == FAST_FLAT, f"cell {cell}: decoded by {names}"
  sess.close()
  assert torch.equal(out, vol)
