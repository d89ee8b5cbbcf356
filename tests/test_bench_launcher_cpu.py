"""bench.py's own multi-rank launcher (`python bench.py --gpus N` without torchrun): N rank
processes are spawned before anything touches a GPU, rendezvous over 127.0.0.1, and rank 0's
line reports n_gpus = N.  CKL_BENCH_REHEARSAL=dry swaps the compute for a gloo barrier +
all-reduce, so the launcher is covered on a box without GPUs."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra):
  env = dict(os.environ, **env_extra)
  for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
    env.pop(k, None)
  return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True, timeout=300)


def test_gpus2_spawns_two_ranks():
  p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"CKL_BENCH_REHEARSAL": "dry"})
  assert p.returncode == 0, p.stderr
  line = json.loads(p.stdout.strip().splitlines()[-1])
  assert line["n_gpus"] == 2 and line["dry_run"] is True
  assert line["max_over_ranks"] == 2.0      # the reduction saw both ranks
  assert "1024x1024x512 uint32" in line["metric"]


def test_metric_follows_shape_and_dtype():
  p = _run(["--gpus", "2", "--shape", "1024x1024x1024", "--dtype", "uint64", "--scaling", "strong"], {"CKL_BENCH_REHEARSAL": "dry"})
  assert p.returncode == 0, p.stderr
  line = json.loads(p.stdout.strip().splitlines()[-1])
  assert "1024x1024x1024 uint64" in line["metric"] and line["scaling"] == "strong" and line["n_gpus"] == 2


def test_world_size_mismatch_is_refused():
  env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", CKL_BENCH_REHEARSAL="dry")
  p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
  assert p.returncode != 0 and "WORLD_SIZE" in p.stderr


def test_too_few_gpus_is_refused():
  # no rehearsal: the launcher counts devices (none here, or fewer than 64 anywhere) and refuses
  p = _run(["--gpus", "64"], {})
  assert p.returncode != 0 and "GPU" in p.stderr


def test_a_failed_rank_ends_the_others():
  """One rank dies before the rendezvous: the launcher ends the rank that would wait for it for ever and
  reports the failure, instead of hanging on rank 0's output."""
  import time
  t0 = time.time()
  p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"CKL_BENCH_REHEARSAL": "dry", "CKL_BENCH_TEST_FAIL_RANK": "1"})
  assert p.returncode != 0 and "exit codes" in p.stderr, p.stderr
  assert time.time() - t0 < 120


def test_config_presets_select_the_baseline_shapes_and_strong_scaling():
  """bench.py --config c3 | c4 | c4pins: the fixed volumes of BASELINE.json dealt out over the ranks (strong scaling: what
  an 8-GPU run of those configurations measures); c2 / no preset: the metric's line, one slab per GPU (weak); explicit
  flags win over the preset."""
  import importlib
  import sys
  sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  bench = importlib.import_module("bench")
  a = bench.parse_args([])
  assert (a.shape, a.dtype, a.scaling, a.markov, a.pins) == ("1024x1024x512", "uint32", "weak", 0, 0)
  a = bench.parse_args(["--config", "c3", "--gpus", "8"])
  assert (a.shape, a.dtype, a.scaling, a.gpus) == ("1024x1024x1024", "uint64", "strong", 8)
  a = bench.parse_args(["--config", "c4"])
  assert (a.shape, a.dtype, a.scaling, a.markov, a.pins) == ("2048x2048x256", "uint32", "strong", 5, 0)
  a = bench.parse_args(["--config", "c4pins"])
  assert (a.markov, a.pins, a.scaling) == (5, 1, "strong")
  a = bench.parse_args(["--config", "c4", "--scaling", "weak", "--markov", "0"])
  assert (a.shape, a.scaling, a.markov) == ("2048x2048x256", "weak", 0)
