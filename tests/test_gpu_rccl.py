"""The sharded codec over RCCL (torch.distributed backend "nccl") with device tensors, on the one GPU
of the test box: a process group of ONE rank takes the sharded path (`force_sharded`), so every
collective of ShardedCodec.compress / open_decoder — all_gather of the slab statistics, all_reduce of
the markov histogram, the unique-label exchange from inside the encode, the section table, barrier,
broadcast of the stream, the point-to-point collection of the pin stage — runs through RCCL on cuda
tensors.  More ranks need more GPUs (the driver's 8-GPU run); the orchestration with two ranks is
covered over gloo in tests/test_distributed_cpu.py and tests/test_gpu_sharded.py."""
import os
import socket

import numpy as np
import pytest

from crackle_amd import synth

pytestmark = pytest.mark.gpu


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _run(q):
  import torch
  import torch.distributed as dist
  from crackle_amd import distributed as ckd
  import sys
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  from oracle import oracle
  checker = oracle.best()
  dev = torch.device("cuda", 0)
  torch.cuda.set_device(dev)
  dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
  results = []
  try:
    for dt, order, pins in ((np.uint32, 0, False), (np.uint32, 5, False), (np.uint64, 3, True), (np.uint16, 0, True)):
      vol = synth.voronoi_labels((192, 160, 10), dt, seed=91, device=dev, cell=(16, 16, 4), offset=(1 << 40) if dt == np.uint64 else 0)
      sz, sy, sx = vol.shape
      codec = ckd.ShardedCodec(ckd.HipBackend(0, zero_copy=True), rank=0, world=1, device=dev, compute_device=dev, force_sharded=True)
      binary = bytes(codec.compress(vol, (sx, sy, sz), markov_model_order=order, allow_pins=pins))
      want = checker.compress(synth.as_numpy_f(vol), markov_model_order=order, allow_pins=pins)
      session = codec.open_decoder(binary, (sx, sy, sz))
      back = torch.zeros_like(vol)
      session.run(back)
      torch.cuda.synchronize()
      results.append((str(np.dtype(dt)), order, pins, binary == want, bool(torch.equal(back, vol))))
      session.close()
    dist.barrier()
  finally:
    dist.destroy_process_group()
  q.put(results)


def test_sharded_codec_over_rccl_process_group_of_one():
  import torch.multiprocessing as mp
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(_free_port())
  ctx = mp.get_context("spawn")      # RCCL in a child: the test process keeps no communicator
  q = ctx.Queue()
  p = ctx.Process(target=_run, args=(q,))
  p.start()
  results = q.get(timeout=300)
  p.join(timeout=60)
  assert p.exitcode == 0
  assert len(results) == 4
  for dt, order, pins, same, roundtrip in results:
    assert same, f"stream differs from the reference's ({dt}, markov {order}, pins {pins})"
    assert roundtrip, f"decode differs ({dt}, markov {order}, pins {pins})"


# ---- two ranks on two GPUs: RCCL moving bytes between devices ------------------------------------------
# Skips on a one-GPU box (this pool's test boxes) and fires on the first box with two: every collective and
# point-to-point transfer of the sharded codec between two devices, the unsigned labels as signed views
# (uint16 / uint32 / uint64 >= 2^63), the pin stage's collection on rank 0, and bench.py's own --gpus 2 path.

def _gpu_count():
  import torch
  return torch.cuda.device_count()      # does not initialise a GPU on this image


def _run_two(rank, port, q):
  import sys
  import torch
  import torch.distributed as dist
  from crackle_amd import distributed as ckd
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  from oracle import oracle
  checker = oracle.best()
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dev = torch.device("cuda", rank)
  torch.cuda.set_device(dev)
  dist.init_process_group(backend="nccl", rank=rank, world_size=2, device_id=dev)
  results = []
  try:
    cases = ((np.uint32, 0, False, 0), (np.uint32, 5, False, 0), (np.uint64, 5, True, 1 << 40), (np.uint64, 0, False, 1 << 40), (np.uint16, 2, True, 0))
    for dt, order, pins, offset in cases:
      sx, sy, sz_total = 192, 160, 12
      sz = sz_total // 2
      vol = synth.voronoi_labels((sx, sy, sz_total), dt, seed=93, device=dev, cell=(16, 16, 4), offset=offset, z_range=(rank * sz, (rank + 1) * sz))
      if dt == np.uint64:      # labels above 2^63, one block per slab (they travel as signed views: RCCL has no uint64)
        vol.view(torch.int64)[1, 5:11, 7:12] = -(1 << 63) + 5 + rank
      codec = ckd.ShardedCodec(ckd.HipBackend(rank, zero_copy=True), rank=rank, world=2, device=dev, compute_device=dev)
      binary = codec.compress(vol, (sx, sy, sz), markov_model_order=order, allow_pins=pins)
      same = None
      if rank == 0:
        whole = synth.as_numpy_f(synth.voronoi_labels((sx, sy, sz_total), dt, seed=93, cell=(16, 16, 4), offset=offset)).copy(order="F")
        if dt == np.uint64:
          whole[7:12, 5:11, 1] = (1 << 63) + 5
          whole[7:12, 5:11, sz + 1] = (1 << 63) + 6
        same = bytes(binary) == checker.compress(whole, markov_model_order=order, allow_pins=pins)
      session = codec.open_decoder(binary, (sx, sy, sz))
      back = torch.zeros_like(vol)
      session.run(back)
      torch.cuda.synchronize()
      results.append((str(np.dtype(dt)), order, pins, same, bool(torch.equal(back, vol))))
      session.close()
    dist.barrier()
  finally:
    dist.destroy_process_group()
  q.put((rank, results))


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs: RCCL between devices")
def test_sharded_codec_over_rccl_two_ranks():
  import torch.multiprocessing as mp
  port = _free_port()
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  procs = [ctx.Process(target=_run_two, args=(r, port, q)) for r in range(2)]
  for p in procs:
    p.start()
  got = dict(q.get(timeout=600) for _ in range(2))
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  for dt, order, pins, same, roundtrip in got[0]:
    assert same, f"merged stream differs from the reference's ({dt}, markov {order}, pins {pins})"
  for r in (0, 1):
    for dt, order, pins, _, roundtrip in got[r]:
      assert roundtrip, f"rank {r} decodes something else ({dt}, markov {order}, pins {pins})"


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs: bench.py --gpus 2 over RCCL")
def test_bench_two_gpus_over_rccl():
  import json
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  for extra in (["--shape", "512x512x32"], ["--shape", "512x512x32", "--dtype", "uint64", "--markov", "5"], ["--shape", "256x256x16", "--pins", "1", "--markov", "5"],
                ["--shape", "512x512x64", "--scaling", "strong"]):
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + extra,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["roundtrip_ok"] is True, line
