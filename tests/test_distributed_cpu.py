"""World-size-2 / 4 / 8 gloo tests of the sharded codec orchestration
(crackle_amd/distributed.py): the reductions, the model agreement, the gather and the
zstack merge must reproduce, byte for byte, what a single encoder produces for the
whole volume.  The per-slab compute is injected: here the CPU oracle (the product
backend is the HIP library; its sharded path runs under RCCL in bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from crackle_amd import synth
from crackle_amd import distributed as ckd

WORLD = 2


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _volume(kind):
  if kind == "voronoi":
    return synth.as_numpy_f(synth.voronoi_labels((96, 80, 12), np.uint16, seed=4, cell=(16, 16, 4)))
  if kind == "voronoi16":
    # 16 slices: slabs of 4 / 2 slices at world 4 / 8; labels of the upper half only appear there
    v = synth.as_numpy_f(synth.voronoi_labels((80, 64, 16), np.uint32, seed=14, cell=(16, 16, 4))).copy(order="F")
    v[:, :, 8:] += 100000
    return v
  if kind == "wide_labels":
    # max label lives in the second slab only: stored width must come from the all-gather
    v = synth.as_numpy_f(synth.voronoi_labels((64, 48, 8), np.uint32, seed=6, cell=(16, 16, 4), modulus=200)).copy(order="F")
    v[3:9, 4:7, 6] = 70000
    return v
  if kind == "noise":
    # PERMISSIBLE crack format decided by the summed pixel_pairs
    return synth.random_labels((48, 40, 6), np.uint32, seed=3, high=2000)
  if kind == "constant":
    return np.full((40, 30, 4), 5, np.uint8, order="F")
  if kind == "u64":
    # C3's label type: stored width 8, labels above 2^40 and two above 2^63 (one per slab):
    # nothing on the way may treat them as signed values
    v = synth.as_numpy_f(synth.voronoi_labels((64, 48, 8), np.uint64, seed=8, cell=(16, 16, 4), offset=1 << 40)).copy(order="F")
    v[5:11, 7:12, 1] = (1 << 63) + 5
    v[20:26, 30:33, 6] = (1 << 64) - 1
    return v
  raise ValueError(kind)


def _worker(rank, port, kind, order, q, pins=False, world=WORLD):
  import sys
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  from oracle_backend import OracleBackend
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  os.environ.setdefault("OMP_NUM_THREADS", "1")
  dist.init_process_group("gloo", rank=rank, world_size=world)
  try:
    vol = _volume(kind)
    sx, sy, sz = vol.shape
    szl = sz // world
    slab = np.asfortranarray(vol[:, :, rank * szl:(rank + 1) * szl])
    codec = ckd.ShardedCodec(OracleBackend(), rank=rank, world=world, device="cpu")
    binary = codec.compress(slab, (sx, sy, szl), markov_model_order=order, allow_pins=pins)
    session = codec.open_decoder(binary, (sx, sy, szl))
    back = np.zeros_like(slab)
    session.run(back)
    q.put((rank, None if binary is None else bytes(binary), bool(np.array_equal(back, slab))))
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,order", [("voronoi", 0), ("voronoi", 3), ("wide_labels", 0), ("noise", 2), ("constant", 4), ("u64", 0), ("u64", 5)])
def test_sharded_compress_equals_whole_volume(port, kind, order):
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  p = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, p, kind, order, q)) for r in range(WORLD)]
  for pr in procs:
    pr.start()
  results = {}
  for _ in range(WORLD):
    rank, binary, ok = q.get(timeout=120)
    results[rank] = (binary, ok)
  for pr in procs:
    pr.join(timeout=60)
    assert pr.exitcode == 0
  vol = _volume(kind)
  whole = port.compress(vol, markov_model_order=order)
  assert results[0][0] == whole, "merged slab streams differ from the whole-volume stream"
  assert results[1][0] is None
  assert results[0][1] and results[1][1], "a rank decoded its z-range wrongly"


@pytest.mark.parametrize("kind,order", [("voronoi", 0), ("voronoi", 2), ("wide_labels", 0), ("noise", 0), ("u64", 5)])
def test_sharded_pins_equal_whole_volume(port, kind, order):
  """allow_pins across slabs: columns cross the slab boundary, component ids are numbered over
  the whole volume and the host cover runs once on rank 0; PERMISSIBLE volumes fall back to
  flat labels exactly like the reference (crackle.hpp:50-64)."""
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  p = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, p, kind, order, q, True)) for r in range(WORLD)]
  for pr in procs:
    pr.start()
  results = {}
  for _ in range(WORLD):
    rank, binary, ok = q.get(timeout=180)
    results[rank] = (binary, ok)
  for pr in procs:
    pr.join(timeout=60)
    assert pr.exitcode == 0
  vol = _volume(kind)
  whole = port.compress(vol, allow_pins=True, markov_model_order=order)
  assert results[0][0] == whole, "sharded pin stream differs from the whole-volume stream"
  assert results[0][1] and results[1][1], "a rank decoded its z-range wrongly"


def test_stats_to_model_matches_reference_tie_rule(port):
  # count descending, ties towards the larger symbol; an all-zero row is (3,2,1,0)
  hist = np.array([[0, 0, 0, 0], [5, 5, 1, 9], [7, 7, 7, 2], [1, 2, 3, 4]], dtype=np.uint32)
  m = ckd.stats_to_model(hist)
  # symbol -> rank
  assert m[0].tolist() == [3, 2, 1, 0]
  assert m[1].tolist() == [2, 1, 3, 0]
  assert m[2].tolist() == [2, 1, 0, 3]
  assert m[3].tolist() == [3, 2, 1, 0]


def _run_world(port_fixture, kind, order, pins, world):
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  p = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, p, kind, order, q, pins, world)) for r in range(world)]
  for pr in procs:
    pr.start()
  results = {}
  for _ in range(world):
    rank, binary, ok = q.get(timeout=300)
    results[rank] = (binary, ok)
  for pr in procs:
    pr.join(timeout=120)
    assert pr.exitcode == 0
  vol = _volume(kind)
  whole = port_fixture.compress(vol, allow_pins=pins, markov_model_order=order)
  assert results[0][0] == whole, f"world {world}: merged slab streams differ from the whole-volume stream"
  assert all(results[r][0] is None for r in range(1, world))
  assert all(results[r][1] for r in range(world)), "a rank decoded its z-range wrongly"


@pytest.mark.parametrize("world,kind,order,pins", [(4, "voronoi16", 0, False), (4, "voronoi16", 2, True), (8, "voronoi16", 3, False)])
def test_sharded_compress_at_world_4_and_8(port, kind, order, pins, world):
  """The same orchestration with 4 and 8 ranks (the node size north_star names): every collective,
  the label merge, the per-rank crc parts and the placement of the sections with more than two
  slabs; pins through the whole-volume stage on rank 0."""
  _run_world(port, kind, order, pins, world)


def _agree_worker(rank, port, q):
  import sys
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  from oracle_backend import OracleBackend
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=WORLD)
  try:
    codec = ckd.ShardedCodec(OracleBackend(), rank=rank, world=WORLD, device="cpu")
    seen = []
    codec._agree(None)      # nobody failed: nobody raises
    seen.append("ok")
    # rank 1 fails its local step: BOTH ranks must raise, the healthy one too (it would otherwise sit in the next collective)
    exc = codec._local(lambda: (_ for _ in ()).throw(ValueError("rank-local failure"))) if rank == 1 else codec._local(lambda: None)
    try:
      codec._agree(exc)
      seen.append("passed")
    except ValueError as e:
      seen.append("own:" + str(e))
    except RuntimeError as e:
      seen.append("other:" + str(e))
    # and the group is still usable afterwards
    t = torch.tensor([rank + 1])
    dist.all_reduce(t)
    seen.append(int(t.item()))
    q.put((rank, seen))
  finally:
    dist.destroy_process_group()


def test_ranks_agree_on_a_rank_local_failure():
  """ShardedCodec._agree (the row-sharded pin stage): a rank whose local pass raised takes every rank with it."""
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_agree_worker, args=(r, port, q)) for r in range(WORLD)]
  for p in procs:
    p.start()
  got = dict(q.get(timeout=120) for _ in range(WORLD))
  for p in procs:
    p.join(timeout=60)
  assert got[0] == ["ok", "other:sharded pin stage: another rank failed in its local pass", 3]
  assert got[1] == ["ok", "own:rank-local failure", 3]
