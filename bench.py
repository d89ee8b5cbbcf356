#!/usr/bin/env python3
"""Throughput of the crackle encode+decode hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: encode the
label volume resident in HBM into .ckl bytes, then decode those bytes (resident in
HBM) back into a label volume in HBM.  At N=1 the workload is BASELINE.json
configs[2]: 1024x1024x512 uint32.  At N>1 every rank holds one such z-slab of a
1024x1024x(512*N) volume (weak scaling): format-deciding reductions and the gather
of per-slab streams go over RCCL, rank 0 merges them into one .ckl, every rank
decodes its own z-range.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field meanings).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The encoder overlaps its crack and label streams; the HIP runtime multiplexes streams onto 4
# hardware queues by default and torch / RCCL take some: ask for 8 before the runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def parse_args():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=10)
  ap.add_argument("--warmup", type=int, default=3)
  ap.add_argument("--shape", type=str, default="1024x1024x512", help="per-GPU slab, SXxSYxSZ")
  ap.add_argument("--dtype", type=str, default="uint32")
  ap.add_argument("--markov", type=int, default=0)
  ap.add_argument("--pins", type=int, default=0, help="allow_pins (parity / rehearsal runs; the metric is quoted on flat labels)")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--cpu-sample-slices", type=int, default=128)
  return ap.parse_args()


def measured_copy_bandwidth(torch, dev, nbytes=1 << 30, reps=5):
  """Device-to-device copy rate on this box (read + write bytes per second), for context
  next to the 8 TB/s spec figure."""
  a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
  b = torch.empty_like(a)
  b.copy_(a)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    b.copy_(a)
  e1.record()
  torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / reps
  del a, b
  return 2.0 * nbytes / (ms * 1e-3) / 1e9


def pmc_traffic(kernel, workload_key):
  """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc pass
  (profiles/pmc_traffic.json, produced by tools/pmc_summary.py from the same workload);
  None when no counters were collected for this workload."""
  path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
  try:
    with open(path) as f:
      t = json.load(f)
    if t.get("workload") != workload_key:
      return None
    for name, row in t["kernels"].items():
      # decode kernels are instantiated for the output type: "k_paint_runs<u32, true>"
      if name == kernel or (name.startswith(kernel + "<") and not name.endswith("false>")):
        return row.get("hbm_bytes_per_launch")
    return None
  except (OSError, ValueError, KeyError):
    return None


def cpu_baseline(vol_np_slab, markov, hip_bytes_for_slab=None):
  """Times the CPU checker (the compiled reference when oracle/_ref travelled here,
  else the C restatement) on a bounded z-slab of the same workload."""
  from oracle import oracle
  chk = oracle.best()
  cores = os.cpu_count() or 1
  best_e, best_d, binary = None, None, None
  for _ in range(2):
    t = time.perf_counter()
    binary = chk.compress(vol_np_slab, markov_model_order=markov, parallel=cores)
    te = time.perf_counter() - t
    t = time.perf_counter()
    out = chk.decompress(binary, parallel=cores)
    td = time.perf_counter() - t
    best_e = te if best_e is None else min(best_e, te)
    best_d = td if best_d is None else min(best_d, td)
  ok = bool(np.array_equal(out.reshape(vol_np_slab.shape, order="F"), vol_np_slab))
  vox = vol_np_slab.size
  # one-thread row on a quarter of the sample
  q = np.asfortranarray(vol_np_slab[:, :, :max(1, vol_np_slab.shape[2] // 4)])
  t = time.perf_counter()
  b1 = chk.compress(q, markov_model_order=markov, parallel=1)
  chk.decompress(b1, parallel=1)
  t1 = time.perf_counter() - t
  res = {
    "value": vox / (best_e + best_d),
    "unit": "voxels/s",
    "cores": cores,
    "kind": chk.kind,
    "sample": f"{vol_np_slab.shape[0]}x{vol_np_slab.shape[1]}x{vol_np_slab.shape[2]} {vol_np_slab.dtype} z-slab of the same synthetic volume, encode+decode, parallel={cores}, best of 2",
    "encode_voxels_per_s": vox / best_e,
    "decode_voxels_per_s": vox / best_d,
    "single_thread_voxels_per_s": q.size / t1,
    "roundtrip_ok": ok,
  }
  if hip_bytes_for_slab is not None:
    res["hip_bytes_equal_cpu_bytes_on_sample"] = bool(hip_bytes_for_slab == binary)
  return res


def main():
  args = parse_args()
  import torch
  import torch.distributed as dist
  from crackle_amd import _lib, synth
  from crackle_amd import distributed as ckd

  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  # rehearsal of the multi-rank path on a box with fewer GPUs than ranks (never for reported
  # numbers): CKL_BENCH_REHEARSAL=1 maps the ranks onto the available devices and uses gloo
  rehearsal = os.environ.get("CKL_BENCH_REHEARSAL") == "1"
  n_dev = torch.cuda.device_count()
  dev_index = (local_rank % max(n_dev, 1)) if (world > 1 and rehearsal) else (local_rank if world > 1 else 0)
  if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(dev_index)
    if rehearsal:
      dist.init_process_group(backend="gloo")
    else:
      dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{dev_index}"))
  dev = torch.device(f"cuda:{dev_index}")
  torch.cuda.set_device(dev)
  L = _lib.lib()
  assert L.ckl_device_count() > dev_index, "no HIP device for this rank"

  sx, sy, sz = (int(v) for v in args.shape.lower().split("x"))
  np_dtype = np.dtype(args.dtype)
  voxels_local = sx * sy * sz
  voxels_total = voxels_local * world

  # synthetic connectomics-style labels, generated on device;
  # rank r holds slices [r*sz, (r+1)*sz) of one global volume
  offset = (1 << 40) if np_dtype.itemsize == 8 else 0
  vol = synth.voronoi_labels((sx, sy, sz * world), np_dtype, seed=2, device=dev, offset=offset,
                             z_range=(rank * sz, (rank + 1) * sz)) if world > 1 else \
        synth.voronoi_labels((sx, sy, sz), np_dtype, seed=2, device=dev, offset=offset)
  torch.cuda.synchronize()

  # the stream stays in the library's pinned host buffer (no copy into a Python bytes object)
  backend = ckd.HipBackend(dev_index, zero_copy=True)
  coll_dev = torch.device("cpu") if (world > 1 and rehearsal) else dev    # gloo: collectives on host tensors
  codec = ckd.ShardedCodec(backend, rank=rank, world=world, device=coll_dev, compute_device=dev)

  def barrier():
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  out = torch.empty_like(vol)
  enc_ms, dec_ms, dec_kernel_ms, dec_pipe_ms, enc_pipe_ms, enc_kernel_ms = [], [], [], [], [], []
  binary = None
  dec_stages = []
  total_s = 0.0
  for step in range(args.warmup + args.steps):
    timed = step >= args.warmup
    barrier()
    t0 = time.perf_counter()
    binary = codec.compress(vol, (sx, sy, sz), markov_model_order=args.markov, allow_pins=bool(args.pins))   # merged stream on rank 0
    barrier()
    t1 = time.perf_counter()
    # decode leg: the stream is made resident first (not timed), then every rank decodes its z-range
    session = codec.open_decoder(binary, (sx, sy, sz))
    barrier()
    t2 = time.perf_counter()
    session.run(out)
    barrier()
    t3 = time.perf_counter()
    if timed:
      enc_ms.append((t1 - t0) * 1e3)
      dec_ms.append((t3 - t2) * 1e3)
      total_s += (t1 - t0) + (t3 - t2)
      p, k = session.timing()
      dec_pipe_ms.append(p); dec_kernel_ms.append(k)
      dec_stages.append(session.stages())
      p, k = backend.encoder_timing()
      enc_pipe_ms.append(p); enc_kernel_ms.append(k)
    session.close()

  ok_local = bool(torch.equal(out.view(torch.uint8), vol.view(torch.uint8)))
  # max over ranks of the timed wall clock
  t = torch.tensor([total_s, sum(enc_ms), sum(dec_ms), 0.0 if ok_local else 1.0], dtype=torch.float64, device=coll_dev)
  if world > 1:
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
  total_s, enc_sum_ms, dec_sum_ms, any_bad = (float(v) for v in t.tolist())

  if rank == 0:
    K = args.steps
    ms_per_step = total_s * 1e3 / K
    ckl_len = len(binary)
    item = np_dtype.itemsize
    # roofline of the dominant decode kernel: algorithmic bytes per launch =
    # label bytes written + stream bytes read, this rank's slab
    alg_bytes = voxels_local * item + ckl_len / world
    # per-stage means over the timed steps; the dominant decode kernel is the slowest stage
    stage_names = [n for n, _ in dec_stages[0]]
    stage_ms = {n: float(np.mean([dict(s)[n] for s in dec_stages])) for n in stage_names}
    dom = max(stage_ms, key=stage_ms.get)
    k_ms = stage_ms[dom]
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    copy_gbs = measured_copy_bandwidth(torch, dev)
    workload_key = f"{sx}x{sy}x{sz} {np_dtype.name} markov {args.markov}"
    res = {
      "metric": "voxels/s encode+decode, 1024x1024x512 uint32; bit-exact .ckl bytes",
      "value": voxels_total * K / total_s,
      "unit": "voxels/s",
      "n_gpus": world,
      "steps": K,
      "warmup": args.warmup,
      "ms_per_step": ms_per_step,
      "higher_is_better": True,
      "scaling": "weak",
      "vs_baseline": None,
      "dtype": {1: "u8", 2: "u16", 4: "u32", 8: "u64"}[item],
      "data": "synthetic",
      "config": {
        "workload": f"{sx}x{sy}x{sz * world} {np_dtype.name} jittered-Voronoi labels (cell 32x32x8), encode+decode, {'pin' if args.pins else 'flat'} labels, markov {args.markov}",
        "per_gpu_slab": f"{sx}x{sy}x{sz}",
        "parallelism": f"z-slab x{world}",
      },
      "roundtrip_ok": any_bad == 0.0,
      "compressed_bytes": ckl_len,
      "compression_ratio_pct": 100.0 * ckl_len / (voxels_total * item),
      "encode_voxels_per_s": voxels_total * K / (enc_sum_ms * 1e-3),
      "decode_voxels_per_s": voxels_total * K / (dec_sum_ms * 1e-3),
      "encode_ms": float(np.mean(enc_ms)),
      "decode_ms": float(np.mean(dec_ms)),
      "decode_device_pipeline_ms": float(np.mean(dec_pipe_ms)),
      "encode_device_pipeline_ms": float(np.mean(enc_pipe_ms)),
      "encode_dfs_kernel_ms": float(np.mean(enc_kernel_ms)),
      "roofline": {
        "bound": "hbm",
        "kernel": dom + " (decode)",
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "traffic": pmc_traffic(dom, workload_key),
        "measured_copy_GBs": copy_gbs,
        "frac_of_measured_copy": achieved / copy_gbs,
        "algorithmic_bytes_per_launch": alg_bytes,
        "kernel_ms": k_ms,
        "decode_pipeline_frac": alg_bytes / (float(np.mean(dec_pipe_ms)) * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "decode_stage_ms": stage_ms,
      },
    }
    # the same figure for the encoder's longest kernel (the serial trail over nodes: one
    # wavefront per slice, bound by dependent instruction latency, not by bytes)
    enc_k = float(np.mean(enc_kernel_ms))
    if enc_k > 0:
      res["roofline_encode"] = {
        "bound": "hbm", "kernel": "k_trail_dfs (encode)",
        "achieved": alg_bytes / (enc_k * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": alg_bytes / (enc_k * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "traffic": pmc_traffic("k_trail_dfs", workload_key),
        "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": enc_k,
        "encode_pipeline_frac": alg_bytes / (float(np.mean(enc_pipe_ms)) * 1e-3) / 1e9 / HBM_PEAK_GBS,
      }
    if not args.no_cpu_baseline:
      ns = min(args.cpu_sample_slices, sz)
      slab = synth.as_numpy_f(vol[:ns])
      res["cpu_baseline"] = cpu_baseline(np.asfortranarray(slab), args.markov)
    print(json.dumps(res), flush=True)

  if world > 1:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
