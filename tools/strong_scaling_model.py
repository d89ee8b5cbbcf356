#!/usr/bin/env python3
"""Strong-scaling budget of the fixed-size BASELINE.json volumes (C3: 1024 x 1024 x 1024 uint64; C4: 2048 x 2048 x 256
uint32, markov order 5, with and without pins) from ONE GPU: the whole volume through the plain path (N = 1), and a
rank's share of it — a slab of sz / N slices — through the sharded path as a process group of one over RCCL
(CKL_BENCH_REHEARSAL=group1: every collective, the node-local buffer, the sealing).  Slabs are independent between
the ranks of a z-sharded encode / decode, so T(N) ~ T_group1(sz / N) (+ the all-gathers' latency over N ranks, which a
group of one cannot show) and the projected speed-up is T(1) / T_group1(sz / N).

  python tools/strong_scaling_model.py [TAG]     -> gpurun_out/TAG/strong_scaling.json / .txt

Not a measurement of N GPUs: the driver's SCALE_rNN.json is."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench(extra, group1):
  env = dict(os.environ)
  if group1:
    env["CKL_BENCH_REHEARSAL"] = "group1"
    env["CKL_PROFILE"] = "1"      # the sharded encoder's stage times on stderr (rank 0's pin section is read from them)
  else:
    env.pop("CKL_BENCH_REHEARSAL", None)
  cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "5", "--warmup", "2"] + extra
  r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
  if r.returncode != 0:
    raise RuntimeError(" ".join(cmd) + "\n" + r.stderr[-1500:])
  d = json.loads(r.stdout.strip().splitlines()[-1])
  sections = [float(m) for m in re.findall(r"pins:section=([0-9.]+)", r.stderr)]
  d["pins_section_ms"] = sorted(sections)[len(sections) // 2] if sections else None      # median over the steps
  return d


CONFIGS = [
  ("C3 1024x1024x1024 uint64", 1024, 1024, 1024, ["--dtype", "uint64"]),
  ("C4 2048x2048x256 uint32 markov 5", 2048, 2048, 256, ["--markov", "5"]),
  ("C4 2048x2048x256 uint32 markov 5 pins", 2048, 2048, 256, ["--markov", "5", "--pins", "1"]),
]


def main():
  tag = sys.argv[1] if len(sys.argv) > 1 else "strong_scaling"
  out_dir = os.path.join(ROOT, "gpurun_out", tag)
  os.makedirs(out_dir, exist_ok=True)
  only = os.environ.get("CKL_SCALING_ONLY")
  result, lines = {}, []
  for name, sx, sy, sz, extra in CONFIGS:
    if only and only not in name:
      continue
    rows = {}
    whole = bench(["--shape", f"{sx}x{sy}x{sz}"] + extra, group1=False)
    t1 = whole["ms_per_step"]
    rows["1"] = {"slab": sz, "ms_per_step": t1, "encode_ms": whole["encode_ms"], "decode_ms": whole["decode_total_ms"], "path": "plain", "roundtrip_ok": whole["roundtrip_ok"]}
    lines.append(f"{name}: whole volume on one GPU (plain path) {t1:.2f} ms per step (encode {whole['encode_ms']:.2f}, decode {whole['decode_total_ms']:.2f})")
    for n in (1, 2, 4, 8):
      slab = sz // n
      g = bench(["--shape", f"{sx}x{sy}x{slab}"] + extra, group1=True)
      tn = g["ms_per_step"]
      rows[f"group1_{n}"] = {"slab": slab, "ms_per_step": tn, "encode_ms": g["encode_ms"], "decode_ms": g["decode_total_ms"], "path": "sharded, group of one over RCCL",
                             "projected_speedup": t1 / tn, "roundtrip_ok": g["roundtrip_ok"]}
      lines.append(f"  N = {n}: slab of {slab:4d} slices through the sharded path {tn:7.2f} ms per step (encode {g['encode_ms']:.2f}, decode {g['decode_total_ms']:.2f})"
                   f" -> projected speed-up {t1 / tn:.2f}x, efficiency {t1 / tn / n:.2f}")
      if g.get("pins_section_ms") is not None:
        # A group of one sees its own slab's pin stage only.  On N ranks the device passes work on 1 / N of the rows of
        # the WHOLE volume (the same number of voxels as the slab), but rank 0 writes the section of the whole volume:
        # the slab's section is replaced by the whole volume's (the N = 1 row of this table).
        if n == 1:
          whole_section = g["pins_section_ms"]
        corrected = tn - g["pins_section_ms"] + whole_section
        rows[f"group1_{n}"].update({"pins_section_ms": g["pins_section_ms"], "whole_volume_section_ms": whole_section, "ms_per_step_with_whole_section": corrected,
                                    "projected_speedup_with_whole_section": t1 / corrected})
        lines.append(f"         rank 0's pin section: {g['pins_section_ms']:.1f} ms for this slab, {whole_section:.1f} ms for the whole volume -> {corrected:7.2f} ms per step,"
                     f" projected speed-up {t1 / corrected:.2f}x, efficiency {t1 / corrected / n:.2f}")
    result[name] = rows
    print("\n".join(lines[-9:]), flush=True)
  # C2, the metric's line, scales WEAKLY (bench.py's default: one 1024 x 1024 x 512 slab per GPU): what N ranks lose against
  # N independent GPUs is the sharded path itself — measured here as a group of one against the plain path on the same slab
  if not only or only in "C2 weak":
    plain = bench(["--shape", "1024x1024x512"], group1=False)
    g1 = bench(["--shape", "1024x1024x512"], group1=True)
    eff = plain["ms_per_step"] / g1["ms_per_step"]
    result["C2 1024x1024x512 uint32 per GPU (weak)"] = {
      "plain_ms_per_step": plain["ms_per_step"], "group1_ms_per_step": g1["ms_per_step"], "sharded_path_overhead_pct": 100.0 * (g1["ms_per_step"] / plain["ms_per_step"] - 1.0),
      "projected_weak_efficiency": eff, "plain_value": plain["value"], "group1_value": g1["value"],
      "projected_value": {str(n): (plain["value"] if n == 1 else n * g1["value"]) for n in (1, 2, 4, 8)},
      "projected_speedup": {str(n): (1.0 if n == 1 else n * eff) for n in (1, 2, 4, 8)},
      "roundtrip_ok": bool(plain["roundtrip_ok"] and g1["roundtrip_ok"]),
    }
    lines.append(f"C2 1024x1024x512 uint32 per GPU, weak scaling: plain path {plain['ms_per_step']:.2f} ms per step, sharded path as a group of one {g1['ms_per_step']:.2f} ms"
                 f" (+{100.0 * (g1['ms_per_step'] / plain['ms_per_step'] - 1.0):.1f} %) -> projected efficiency {eff:.2f} from 2 ranks on: "
                 + ", ".join(f"N = {n}: {n * eff:.2f}x" for n in (2, 4, 8)))
    print(lines[-1], flush=True)
  try:
    sys.path.insert(0, ROOT)
    from crackle_amd import build as ckl_build
    sha = ckl_build.source_digest()
  except Exception:      # noqa: BLE001
    sha = None
  model = {"lib_sha16": sha, "note": "projections from ONE GPU (a rank's share through the sharded path as a process group of one over RCCL): not a measurement of N GPUs", "configs": result}
  with open(os.path.join(out_dir, "scaling_model.json"), "w") as f:
    json.dump(model, f, indent=1)
  with open(os.path.join(out_dir, "strong_scaling.json"), "w") as f:
    json.dump(result, f, indent=1)
  with open(os.path.join(out_dir, "strong_scaling.txt"), "w") as f:
    f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
  main()
