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
