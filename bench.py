#!/usr/bin/env python3
"""Throughput of the crackle encode+decode hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]

A "step" is one pass of the hot path over one batch of synthetic input: encode the
label volume resident in HBM into .ckl bytes, then decode those bytes (resident in
HBM) back into a label volume in HBM.  At N=1 the workload is BASELINE.json
configs[2]: 1024x1024x512 uint32.

N>1: one process per GPU.  Under `torch.distributed.run` (RANK / WORLD_SIZE in the
environment) this process is one rank; started plainly with --gpus N it spawns the N
rank processes itself — before anything touches the GPU — and relays rank 0's line.
  weak scaling (default): every rank holds one --shape slab of a volume N times as deep;
  strong scaling: --shape is the whole volume, its slices are dealt out over the ranks
  (BASELINE.json configs[3] is `--scaling strong --shape 1024x1024x1024 --dtype uint64`).
Format-deciding reductions and the small per-slab tables go over RCCL (backend "nccl"),
every rank writes its slab's sections at their final offsets of one node-local buffer,
every rank decodes its own z-range.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field meanings).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The encoder overlaps its crack and label streams and the decoder pipelines z-chunks; the
# HIP runtime multiplexes streams onto 4 hardware queues by default and torch / RCCL take
# some: ask for 8 before the runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


# BASELINE.json's configurations by name (bench.py --config): c2 is the metric's line (one slab per GPU, weak); the others
# are FIXED volumes, so that N ranks share them (strong scaling) — C3 and C4 are the 8-GPU configurations of BASELINE.json
CONFIG_PRESETS = {
  "c1": dict(shape="512x512x128", dtype="uint32", scaling="strong"),
  "c2": dict(shape="1024x1024x512", dtype="uint32", scaling="weak"),
  "c3": dict(shape="1024x1024x1024", dtype="uint64", scaling="strong"),
  "c4": dict(shape="2048x2048x256", dtype="uint32", markov=5, scaling="strong"),
  "c4pins": dict(shape="2048x2048x256", dtype="uint32", markov=5, pins=1, scaling="strong"),
}


def parse_args(argv=None):
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=10)
  ap.add_argument("--warmup", type=int, default=3)
  ap.add_argument("--shape", type=str, default="1024x1024x512", help="SXxSYxSZ: per-GPU slab (weak) or whole volume (strong)")
  ap.add_argument("--dtype", type=str, default="uint32")
  ap.add_argument("--scaling", type=str, default="weak", choices=("weak", "strong"))
  ap.add_argument("--markov", type=int, default=0)
  ap.add_argument("--pins", type=int, default=0, help="allow_pins (parity / rehearsal runs; the metric is quoted on flat labels)")
  ap.add_argument("--data", type=str, default="voronoi", choices=("voronoi", "noise2000", "binary"),
                  help="voronoi: the connectomics-style volume the metric is quoted on; noise2000 / binary: the reference's adversarial "
                       "inputs (uniform-random labels in [0, 2000) / in {0, 1}: benchmarks/README.md:108-114, 193-227), reported for honesty")
  ap.add_argument("--cell", type=str, default="32x32x8", help="cell of the jittered-Voronoi generator (8x8x4: severely over-segmented, the reference's watershed benchmark, benchmarks/README.md:284-318)")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--sync-host-copy", action="store_true", help="the encoder call returns only when its host bytes are complete (default at N = 1: the codes' PCIe copy overlaps the decode leg and is waited for inside the step)")
  ap.add_argument("--cpu-sample-slices", type=int, default=0, help="slices of the CPU baseline's sample (0: the whole slab)")
  ap.add_argument("--config", type=str, default=None, choices=sorted(CONFIG_PRESETS),
                  help="a BASELINE.json configuration by name: c2 = the default (1024x1024x512 uint32 per GPU, weak scaling: the metric's line); "
                       "c1, c3, c4, c4pins = the fixed volumes BASELINE.json names, dealt out over --gpus ranks (strong scaling: what an 8-GPU run "
                       "of those configurations measures); explicit --shape / --dtype / --markov / --pins / --scaling still win")
  args = ap.parse_args(argv)
  if args.config:
    given = {a.split("=")[0] for a in (argv if argv is not None else sys.argv[1:]) if a.startswith("--")}
    for key, val in CONFIG_PRESETS[args.config].items():
      if "--" + key not in given:
        setattr(args, key, val)
  return args


# ------------------------------------------------------------------------------------
# multi-rank launcher (no GPU call may precede it: the children are fresh processes)
# ------------------------------------------------------------------------------------
_json_out = sys.stdout


def _free_port():
  import socket
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    return s.getsockname()[1]


def visible_gpus():
  """GPUs this process may use, counted without touching a GPU runtime: the KFD topology nodes that have
  SIMDs (CPU nodes have none), capped by the device lists of HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES /
  CUDA_VISIBLE_DEVICES.  None when the topology cannot be read (the ranks then check for themselves)."""
  import glob
  n = 0
  paths = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
  if not paths:
    return None
  for path in paths:
    try:
      with open(path) as f:
        for line in f:
          k, _, v = line.partition(" ")
          if k == "simd_count" and int(v.strip() or 0) > 0:
            n += 1
    except (OSError, ValueError):
      return None
  for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
    v = os.environ.get(var)
    if v is not None:
      n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
  return n


def spawn_ranks(args):
  """`python bench.py --gpus N` without a launcher: start N rank processes of this script
  and relay rank 0's JSON line.  Returns the exit code."""
  n = args.gpus
  rehearsal = os.environ.get("CKL_BENCH_REHEARSAL", "")
  if not rehearsal:
    have = visible_gpus()      # from sysfs: the launcher makes no HIP / torch call at all
    if have is not None and have < n:
      print(f"bench.py: --gpus {n} asked for but only {have} GPU(s) are visible", file=sys.stderr)
      return 2
  env = dict(os.environ)
  env["MASTER_ADDR"] = "127.0.0.1"
  env["MASTER_PORT"] = str(_free_port())
  env["WORLD_SIZE"] = str(n)
  env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
  procs = []
  for r in range(n):
    e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
    out = subprocess.PIPE if r == 0 else subprocess.DEVNULL
    procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e, stdout=out))
  # rank 0's stdout is drained by a thread; the ranks are polled, and when one of them fails the others
  # (which would wait for it in the rendezvous or a collective for ever) are ended: exactly the
  # processes started above
  import threading
  chunks = []
  reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
  reader.start()
  failed = False
  while any(p.poll() is None for p in procs):
    if any(p.poll() not in (None, 0) for p in procs):
      failed = True
      break
    time.sleep(0.05)
  if failed:
    time.sleep(1.0)      # let the others fail by themselves first (their own messages are the better diagnosis)
    for p in procs:
      if p.poll() is None:
        p.terminate()
    for p in procs:
      try:
        p.wait(timeout=10)
      except subprocess.TimeoutExpired:
        p.kill()
  rcs = [p.wait() for p in procs]
  reader.join(timeout=10)
  text = (chunks[0] if chunks and chunks[0] else b"").decode("utf-8", "replace")
  if any(rcs):
    sys.stderr.write(text)
    print(f"bench.py: rank exit codes {rcs}", file=sys.stderr)
    return 1
  try:
    last = text.strip().splitlines()[-1]
    res = json.loads(last)
  except (ValueError, IndexError):
    sys.stderr.write(text)
    print("bench.py: rank 0 printed no JSON line", file=sys.stderr)
    return 1
  sys.stdout.write(last + "\n")      # ONE line: whatever else reached rank 0's stdout is dropped
  sys.stdout.flush()
  if res.get("n_gpus") != n:
    print(f"bench.py: the line reports n_gpus={res.get('n_gpus')}, --gpus was {n}", file=sys.stderr)
    return 1
  return 0


def dry_run(args, world, rank):
  """CKL_BENCH_REHEARSAL=dry: rendezvous, barriers and the max-over-ranks reduction of the
  real run over gloo, no compute — covers the launcher on a box without GPUs."""
  if os.environ.get("CKL_BENCH_TEST_FAIL_RANK") == str(rank):      # testing: a rank that dies before the rendezvous
    sys.exit(3)
  import torch
  import torch.distributed as dist
  if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo")
  t = torch.tensor([float(rank + 1)], dtype=torch.float64)
  if world > 1:
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
  if rank == 0:
    print(json.dumps({"metric": metric_name(args), "value": None, "unit": "voxels/s", "n_gpus": world, "steps": args.steps,
                      "warmup": args.warmup, "dry_run": True, "max_over_ranks": float(t.item()), "scaling": args.scaling}), file=_json_out, flush=True)
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()


# ------------------------------------------------------------------------------------
def metric_name(args):
  sx, sy, sz = (int(v) for v in args.shape.lower().split("x"))
  extra = "" if args.data == "voronoi" else f" ({args.data} labels)"
  return f"voxels/s encode+decode, {sx}x{sy}x{sz} {args.dtype}{extra}; bit-exact .ckl bytes"


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


def lib_sha16():
  """Digest of the library's sources (crackle_amd.build.source_digest): ties a PMC profile to a build."""
  try:
    from crackle_amd import build as ckl_build
    return ckl_build.source_digest()
  except OSError:
    return None


def pmc_traffic(kernels, workload_key):
  """HBM bytes per launch, summed over `kernels` (name prefixes), from the committed rocprofv3
  --pmc passes of this workload (profiles/r03_pmc_traffic.json, written by tools/pmc_summary.py:
  separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled as the guide prescribes for gfx950).
  None when no counters were collected for this workload, a kernel is missing from them, or the
  profile was taken from another build of the library (its recorded lib_sha16 differs)."""
  for fn in ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json"):
    path = os.path.join(ROOT, "profiles", fn)
    try:
      with open(path) as f:
        t = json.load(f)
    except (OSError, ValueError):
      continue
    if t.get("workload") != workload_key or t.get("lib_sha16") != lib_sha16():
      continue
    total, per = 0.0, {}
    for want in kernels:
      got = None
      for name, row in t.get("kernels", {}).items():
        if name.split("<")[0] == want:      # the instantiations of a template that ran (names carry their arguments)
          got = (got or 0.0) + float(row.get("hbm_bytes_per_launch", 0.0))
      if got is None:
        return None, {}
      per[want] = got
      total += got
    return total, per
  return None, {}


DECODE_SIDE_KERNELS = ("k_crack_match", "k_strip_ccl", "k_strip_ccl2", "k_slice_resolve", "k_paint_strips", "k_fetch_to_host", "k_fetch_flat_counts",
                       "k_build_geom_table", "k_decode_cracks", "k_flags_to_host", "k_strip_labels", "k_label_map", "k_label_map_pins", "k_paint_runs")


def pmc_traffic_encode(workload_key):
  """HBM bytes of ONE encode: every kernel of the committed PMC passes that is not one of the decoder's, its bytes per launch
  times its launches per encode (launches over those of the label-plane kernel, which runs once per encode).  None under the
  conditions of pmc_traffic()."""
  path = os.path.join(ROOT, "profiles", "r05_pmc_traffic.json")
  try:
    with open(path) as f:
      t = json.load(f)
  except (OSError, ValueError):
    return None
  if t.get("workload") != workload_key or t.get("lib_sha16") != lib_sha16():
    return None
  rows = t.get("kernels", {})
  encodes = sum(float(r.get("launches", 0)) for n, r in rows.items() if n.split("<")[0].startswith("k_label_planes"))
  if encodes <= 0:
    return None
  total = 0.0
  for name, r in rows.items():
    if name.split("<")[0] in DECODE_SIDE_KERNELS:
      continue
    total += float(r.get("hbm_bytes_per_launch", 0.0)) * float(r.get("launches", 0)) / encodes
  return total


def cpu_model():
  try:
    with open("/proc/cpuinfo") as f:
      for line in f:
        if line.lower().startswith("model name"):
          return line.split(":", 1)[1].strip()
  except OSError:
    pass
  return "unknown"


def cpu_baseline(np, vol_np_slab, markov, hip_bytes_for_slab=None, whole=False):
  """Times the CPU checker (the compiled reference when oracle/_ref travelled here,
  else the C restatement) on the same workload: the whole volume by default (its pool runs one
  thread per slice at most, src/crackle.hpp:66-69, so a slab would leave cores idle), best of 3."""
  from oracle import oracle
  chk = oracle.best()
  cores = os.cpu_count() or 1
  nz = vol_np_slab.shape[2]
  best_e, best_d, binary = None, None, None
  for _ in range(3):
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
  # one-thread row on 16 slices of the sample
  q = np.asfortranarray(vol_np_slab[:, :, :max(1, min(nz, 16))])
  t = time.perf_counter()
  b1 = chk.compress(q, markov_model_order=markov, parallel=1)
  chk.decompress(b1, parallel=1)
  t1 = time.perf_counter() - t
  # the reference's pool never runs more threads than slices (src/crackle.hpp:66-69, 570-573)
  threads = min(cores, nz)
  res = {
    "value": vox / (best_e + best_d),
    "unit": "voxels/s",
    "cores": threads,
    "host_cores": cores,
    "cpu_model": cpu_model(),
    "kind": chk.kind,
    "sample": f"{vol_np_slab.shape[0]}x{vol_np_slab.shape[1]}x{nz} {vol_np_slab.dtype} {'(the whole volume)' if whole else 'z-slab'} of the same synthetic volume, encode+decode, parallel={cores} asked, {threads} threads effective (one per slice at most), best of 3; single_thread row: {q.shape[2]} slices, parallel=1",
    "encode_voxels_per_s": vox / best_e,
    "decode_voxels_per_s": vox / best_d,
    "single_thread_voxels_per_s": q.size / t1,
    "roundtrip_ok": ok,
  }
  if hip_bytes_for_slab is not None:
    res["hip_bytes_equal_cpu_bytes_on_sample"] = bool(bytes(hip_bytes_for_slab) == bytes(binary))
  return res


def reference_manifest(name):
  """sha256 / length of the reference encoder's bytes for a full-size configuration
  (tests/golden/manifest_xl.json, written here by tests/gen_golden.py --xl)."""
  try:
    with open(os.path.join(ROOT, "tests", "golden", "manifest_xl.json")) as f:
      return json.load(f).get(name)
  except (OSError, ValueError):
    return None


DECODE_KERNELS_FOR_TRAFFIC = None   # filled from the stage names of the run


def main():
  args = parse_args()
  world_env = os.environ.get("WORLD_SIZE")
  if world_env is None and args.gpus > 1:
    sys.exit(spawn_ranks(args))
  # The contract is ONE JSON line on stdout.  RCCL prints a version banner to stdout when its first
  # communicator is made, and other libraries may chat there too: from here on file descriptor 1 is
  # stderr, and the JSON line goes out through a duplicate of the real stdout.
  global _json_out
  sys.stdout.flush()
  _json_out = os.fdopen(os.dup(1), "w")
  os.dup2(2, 1)
  world = int(world_env or "1")
  if world != args.gpus:
    print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    sys.exit(2)
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  rehearsal_mode = os.environ.get("CKL_BENCH_REHEARSAL", "")
  if rehearsal_mode == "dry":
    return dry_run(args, world, rank)

  import hashlib
  import numpy as np
  import torch
  import torch.distributed as dist
  from crackle_amd import _lib, synth
  from crackle_amd import distributed as ckd

  # rehearsal of the multi-rank path on a box with fewer GPUs than ranks (never for reported
  # numbers): CKL_BENCH_REHEARSAL=1 maps the ranks onto the available devices and uses gloo
  rehearsal = rehearsal_mode == "1"
  # CKL_BENCH_REHEARSAL=group1: one rank, but through the multi-rank code path (RCCL process group
  # of one, sharded codec, barriers, max-over-ranks): what a one-GPU box can rehearse of --gpus N
  group1 = rehearsal_mode == "group1" and world == 1
  n_dev = torch.cuda.device_count()
  if world > 1 and not rehearsal and n_dev < world:
    print(f"bench.py: {world} ranks but {n_dev} GPU(s)", file=sys.stderr)
    sys.exit(2)
  dev_index = (local_rank % max(n_dev, 1)) if (world > 1 and rehearsal) else (local_rank if world > 1 else 0)
  if world > 1 or group1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if group1:
      os.environ.setdefault("MASTER_PORT", str(_free_port()))
      os.environ.setdefault("RANK", "0")
      os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(dev_index)
    if rehearsal:
      dist.init_process_group(backend="gloo")
    elif group1:
      dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device(f"cuda:{dev_index}"))
    else:
      dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{dev_index}"))
  dev = torch.device(f"cuda:{dev_index}")
  torch.cuda.set_device(dev)
  L = _lib.lib()
  assert L.ckl_device_count() > dev_index, "no HIP device for this rank"

  sx, sy, sz_arg = (int(v) for v in args.shape.lower().split("x"))
  np_dtype = np.dtype(args.dtype)
  if args.scaling == "strong":
    if sz_arg % world:
      print(f"bench.py: strong scaling needs the slice count {sz_arg} to divide over {world} ranks", file=sys.stderr)
      sys.exit(2)
    sz = sz_arg // world
  else:
    sz = sz_arg
  sz_total = sz * world
  voxels_local = sx * sy * sz
  voxels_total = voxels_local * world

  # synthetic connectomics-style labels, generated on device;
  # rank r holds slices [r*sz, (r+1)*sz) of one global volume
  offset = (1 << 40) if np_dtype.itemsize == 8 else 0
  cell = tuple(int(v) for v in args.cell.lower().split("x"))
  if args.data == "voronoi":
    vol = synth.voronoi_labels((sx, sy, sz_total), np_dtype, seed=2, device=dev, offset=offset, cell=cell,
                               z_range=(rank * sz, (rank + 1) * sz)) if world > 1 else \
          synth.voronoi_labels((sx, sy, sz), np_dtype, seed=2, device=dev, offset=offset, cell=cell)
  else:
    vol = synth.random_labels_device((sx, sy, sz_total), np_dtype, seed=2, high=2000 if args.data == "noise2000" else 2,
                                     device=dev, z_range=(rank * sz, (rank + 1) * sz))
  torch.cuda.synchronize()

  # the stream stays in the library's pinned host buffer (no copy into a Python bytes object)
  backend = ckd.HipBackend(dev_index, zero_copy=True)
  coll_dev = torch.device("cpu") if (world > 1 and rehearsal) else dev    # gloo: collectives on host tensors
  codec = ckd.ShardedCodec(backend, rank=rank, world=world, device=coll_dev, compute_device=dev, force_sharded=group1)

  def barrier():
    torch.cuda.synchronize()
    if world > 1 or group1:
      dist.barrier()
    torch.cuda.synchronize()

  out = torch.empty_like(vol)
  enc_ms, dec_ms, open_ms, dec_pipe_ms, enc_pipe_ms, enc_kernel_ms = [], [], [], [], [], []
  binary = None
  total_s = 0.0
  # The encoder leaves its stream in HBM as well (ckl_encoder_keep_device_stream): the decode leg starts
  # from those resident bytes — every rank from its own slab's stream, which carries the merged label
  # table: header + z-index + labels + its own crack codes, what SURVEY.md section 8e gives a rank —
  # and its whole cost, session set-up included, is inside the timed region.  Pin streams are merged on
  # rank 0 only: with --pins the ranks take the broadcast stream and its set-up stays outside.
  resident = not (args.pins and (world > 1 or group1))
  if resident:
    backend.keep_device_stream((sx, sy, sz), np_dtype.itemsize, True)
  # The encoder returns when the (slab's) stream is complete in HBM and its crack codes (the bulk of the
  # host copy, 16 MB at C2) cross PCIe — into the host buffer, or into their place in the ranks' shared
  # buffer — while the decode leg runs from the resident stream (ckl_encoder_async_host_copy); the step
  # ends only when the host bytes have arrived and, with several ranks, the merged stream is sealed
  # (`pending()` below, inside the timed region).  --sync-host-copy restores the synchronous call.
  # (not with a markov model: the decoder's set-up then reads the model back as well, and its small round
  # trips queue behind the codes on the link: 1.3 instead of 0.08 ms at C2)
  overlap_copy = resident and not args.sync_host_copy and args.markov == 0
  if overlap_copy:
    backend.async_host_copy((sx, sy, sz), np_dtype.itemsize, True)
  copy_wait_ms = []
  for step in range(args.warmup + args.steps):
    timed = step >= args.warmup
    barrier()
    t0 = time.perf_counter()
    if overlap_copy:
      pending = codec.compress(vol, (sx, sy, sz), markov_model_order=args.markov, allow_pins=bool(args.pins), defer=True)
    else:
      binary = codec.compress(vol, (sx, sy, sz), markov_model_order=args.markov, allow_pins=bool(args.pins))   # merged stream on rank 0
      barrier()
    t1 = time.perf_counter()
    if os.environ.get("CKL_BENCH_PENDING"):      # measuring: device work still pending when the encoder returns (shifts it into encode_ms)
      torch.cuda.synchronize()
      print(f"[bench] device work pending after the encode: {(time.perf_counter() - t1) * 1e3:.2f} ms", file=sys.stderr)
      t1 = time.perf_counter()
    # decode leg: compressed bytes resident in HBM -> labels resident in HBM,
    # ckl_decoder_create_device (header / z-index / label-section head read back, descriptors, scratch)
    # + ckl_decoder_run
    if resident:
      session = backend.open_decoder(backend.device_stream(), 0, sz)
    else:
      session = codec.open_decoder(binary, (sx, sy, sz))
      barrier()
      t1 = time.perf_counter()
    if hasattr(session, "stage_events"):
      session.stage_events(False)      # the timed runs record events around the pipeline only; the per-kernel table comes from the runs below
    if not overlap_copy:
      torch.cuda.synchronize()
    t2 = time.perf_counter()
    session.run(out)
    if overlap_copy:
      tw = time.perf_counter()
      binary = pending()      # waits for the codes' copy; several ranks: barrier + the merged stream's crcs on rank 0
      if timed:
        copy_wait_ms.append((time.perf_counter() - tw) * 1e3)
    barrier()
    t3 = time.perf_counter()
    if timed:
      enc_ms.append((t1 - t0) * 1e3)
      open_ms.append((t2 - t1) * 1e3)
      dec_ms.append((t3 - t2) * 1e3)
      total_s += (t3 - t0) if resident else (t1 - t0) + (t3 - t2)
      p, _ = session.timing()
      dec_pipe_ms.append(p)
      p, k = backend.encoder_timing()
      enc_pipe_ms.append(p); enc_kernel_ms.append(k)
    session.close()

  ok_local = bool(torch.equal(out.view(torch.uint8), vol.view(torch.uint8)))
  # The timed decode leg reads each rank's resident slab stream; the stream the job RETURNS is the merged one
  # in the host buffer (async code copy, _finish, _seal): it is decoded here once, outside the timed region,
  # through the broadcast path (every rank its own z-range), and has to give the same labels.
  if resident:
    out.zero_()
    session = codec.open_decoder(binary, (sx, sy, sz))
    session.run(out)
    session.close()
    ok_local = ok_local and bool(torch.equal(out.view(torch.uint8), vol.view(torch.uint8)))

  # one more decode with the z-chunks serialised on one stream, for the per-kernel table:
  # HIP events between the kernels on the decoder's own stream (ckl_decoder_stage_timing)
  stage_ms = {}
  if rank == 0 or world > 1:
    os.environ["CKL_DECODE_CHUNKS"] = "1"
    session = backend.open_decoder(backend.device_stream(), 0, sz) if resident else codec.open_decoder(binary, (sx, sy, sz))
    acc = {}
    for _ in range(3):
      session.run(out)
      for n, ms in session.stages():
        acc.setdefault(n, []).append(ms)
    serial_pipe = session.timing()[0]
    session.close()
    del os.environ["CKL_DECODE_CHUNKS"]
    stage_ms = {n: float(np.mean(v[1:])) for n, v in acc.items()}

  # max over ranks of the timed wall clock
  t = torch.tensor([total_s, sum(enc_ms), sum(dec_ms) + (sum(open_ms) if resident else 0.0), 0.0 if ok_local else 1.0], dtype=torch.float64, device=coll_dev)
  if world > 1 or group1:
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
  total_s, enc_sum_ms, dec_sum_ms, any_bad = (float(v) for v in t.tolist())

  if rank == 0:
    K = args.steps
    ms_per_step = total_s * 1e3 / K
    ckl_len = len(binary)
    item = np_dtype.itemsize
    # algorithmic bytes of one decode of this rank's slab = label bytes written + stream bytes
    # read (4 B/voxel @u32 plus ~1 % for the stream)
    alg_bytes = voxels_local * item + ckl_len / world
    pipe_ms = float(np.mean(dec_pipe_ms))
    achieved = alg_bytes / (pipe_ms * 1e-3) / 1e9
    copy_gbs = measured_copy_bandwidth(torch, dev)
    workload_key = f"{sx}x{sy}x{sz} {np_dtype.name} markov {args.markov}"
    kernels = [n for n in stage_ms if n.startswith("k_")]
    traffic, traffic_per = pmc_traffic(kernels, workload_key)
    dom = max(stage_ms, key=stage_ms.get) if stage_ms else None
    res = {
      "metric": metric_name(args),
      "value": voxels_total * K / total_s,
      "unit": "voxels/s",
      "n_gpus": world,
      "steps": K,
      "warmup": args.warmup,
      "ms_per_step": ms_per_step,
      "higher_is_better": True,
      "scaling": args.scaling,
      "vs_baseline": None,
      "dtype": {1: "u8", 2: "u16", 4: "u32", 8: "u64"}[item],
      "data": "synthetic",
      "config": {
        "workload": f"{sx}x{sy}x{sz_total} {np_dtype.name} " + {"voronoi": f"jittered-Voronoi labels (cell {args.cell})", "noise2000": "uniform-random labels in [0, 2000)", "binary": "uniform-random labels in {0, 1}"}[args.data] + f", encode+decode, {'pin' if args.pins else 'flat'} labels, markov {args.markov}",
        "preset": args.config or "c2 (default)" if (args.shape, args.dtype, args.markov, args.pins, args.data, args.cell) == ("1024x1024x512", "uint32", 0, 0, "voronoi", "32x32x8") or args.config else "custom",
        "scaling_mode": "weak: every rank holds one --shape slab of a volume N times as deep" if args.scaling == "weak" else "strong: the slices of the one --shape volume are dealt out over the ranks",
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
      "decoder_create_ms": float(np.mean(open_ms)),
      "per_step_ms": {"encode": [round(v, 3) for v in enc_ms], "decoder_create": [round(v, 3) for v in open_ms], "decode": [round(v, 3) for v in dec_ms]},
      "decode_total_ms": float(np.mean(open_ms)) + float(np.mean(dec_ms)),
      "decode_setup_in_value": bool(resident),
      # encode_ms ends when the stream is complete in HBM and the call has returned; with the overlapped
      # host copy the crack codes reach the host buffer during the decode leg, host_copy_wait_ms is what the
      # step still waits for them after the decode (all of it inside `value`)
      "encode_host_copy": "overlapped with the decode leg, completed inside the step" if overlap_copy else "inside the encoder call",
      "host_copy_wait_ms": float(np.mean(copy_wait_ms)) if copy_wait_ms else 0.0,
      "decode_device_pipeline_ms": pipe_ms,
      "encode_device_pipeline_ms": float(np.mean(enc_pipe_ms)),
      "encode_dfs_kernel_ms": float(np.mean(enc_kernel_ms)),
      # the decode direction as one unit (north_star states its target for it): every kernel of one
      # decode, first launch to last completion, HIP events on the decoder's streams
      "roofline": {
        "bound": "hbm",
        "kernel": "decode pipeline: " + " + ".join(kernels),
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic,
        "measured_copy_GBs": copy_gbs,
        "frac_of_measured_copy": achieved / copy_gbs,
        "algorithmic_bytes_per_launch": alg_bytes,
        "kernel_ms": pipe_ms,
        "serialized_pipeline_ms": serial_pipe,
        "decode_stage_ms": stage_ms,
        "decode_stage_traffic": traffic_per,
        "slowest_stage": None if dom is None else {
          "kernel": dom, "kernel_ms": stage_ms[dom],
          "achieved": alg_bytes / (stage_ms[dom] * 1e-3) / 1e9,
          "frac": alg_bytes / (stage_ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS,
        },
      },
    }
    # the same figure for the encode direction as one unit: the slab's label bytes read + its stream bytes
    # written over the encoder's device pipeline (first kernel to last, HIP events on its own streams).  Its
    # longest kernel, the serial trail over nodes (one wavefront per slice, bound by dependent instruction
    # latency, 44 MB of traffic), is a detail field: bytes over ITS time alone are not a roofline fraction.
    enc_k = float(np.mean(enc_kernel_ms))
    enc_p = float(np.mean(enc_pipe_ms))
    if enc_p > 0:
      res["roofline_encode"] = {
        "bound": "hbm", "kernel": "encode pipeline (planes, trail, label stream: two streams)",
        "achieved": alg_bytes / (enc_p * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": alg_bytes / (enc_p * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "traffic": pmc_traffic_encode(workload_key) if (args.pins == 0 and args.data == "voronoi" and args.cell == "32x32x8") else None,
        "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": enc_p,
        "longest_kernel": {"kernel": "k_trail_walk", "kernel_ms": enc_k, "share_of_pipeline": (enc_k / enc_p) if enc_p > 0 else None},
      }
    # bytes against the reference encoder's own output at full size (sha256 of the whole stream)
    if world == 1 and not args.pins:
      name = f"c2_{sx}x{sy}x{sz}_u32" if (np_dtype.itemsize == 4 and args.markov == 0 and args.data == "voronoi" and args.cell == "32x32x8") else None
      ref = reference_manifest(name) if name else None
      if ref is not None:
        sha = hashlib.sha256(bytes(binary.view()) if hasattr(binary, "view") else bytes(binary)).hexdigest()
        res["bytes_match_reference"] = bool(sha == ref["sha256"] and ckl_len == ref["length"])
        res["reference_sha256"] = ref["sha256"]
    if not args.no_cpu_baseline:
      ns = min(args.cpu_sample_slices, sz) if args.cpu_sample_slices > 0 else sz
      slab = np.asfortranarray(synth.as_numpy_f(vol[:ns]))
      hip_slab = None
      if world == 1:
        hip_slab = ckd.HipBackend(dev_index).encode(vol[:ns].contiguous(), (sx, sy, ns), False, True, args.markov, None)
      res["cpu_baseline"] = cpu_baseline(np, slab, args.markov, hip_slab, whole=(ns == sz and world == 1))
    print(json.dumps(res), file=_json_out, flush=True)

  if world > 1 or group1:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
