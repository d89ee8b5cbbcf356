#!/usr/bin/env python3
"""Folds rocprofv3 --pmc passes of SQ counters (tools/sq_profile.sh) into one JSON:
per ckl kernel the mean counter values per launch (warm-up launch dropped) and a few ratios.

  python3 tools/sq_summary.py OUT.json DIR [DIR ...]

Units (MI355X_MICROARCH.md, cycle constants): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles summed over waves; SQ_BUSY_CYCLES is summed over the shader engines.
WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
  n = name.split("(")[0]
  for p in ("void ", "ckl::dev::", "ckl::", "(anonymous namespace)::"):
    n = n.replace(p, "")
  return n.replace("unsigned char", "u8").replace("unsigned short", "u16").replace("unsigned int", "u32").replace("unsigned long", "u64").strip()


def main():
  out = sys.argv[1]
  kernels = defaultdict(lambda: defaultdict(list))
  for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
      disp = defaultdict(lambda: defaultdict(float))
      names = {}
      with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
          key = row.get("Dispatch_Id") or row.get("Correlation_Id")
          disp[key][row["Counter_Name"]] += float(row["Counter_Value"])
          names[key] = row["Kernel_Name"]
      for key in sorted(disp, key=lambda k: int(k)):
        if "ckl" not in names[key]:
          continue
        for c, v in disp[key].items():
          kernels[short(names[key])][c].append(v)
  res = {}
  for k, cs in kernels.items():
    row = {}
    for c, vals in cs.items():
      use = vals[1:] if len(vals) > 1 else vals
      row[c] = sum(use) / len(use)
    wc = row.get("SQ_WAVE_CYCLES")
    if wc:
      for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA"):
        if c in row:
          row["frac_" + c[3:].lower()] = row[c] / wc
    if row.get("SQ_LDS_IDX_ACTIVE"):
      row["lds_conflict_frac"] = row.get("SQ_LDS_BANK_CONFLICT", 0.0) / row["SQ_LDS_IDX_ACTIVE"]
    if row.get("SQ_WAVES"):
      for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM"):
        if c in row:
          row[c[3:].lower() + "_per_wave"] = row[c] / row["SQ_WAVES"]
    res[k] = row
  with open(out, "w") as fh:
    json.dump({"source": "rocprofv3 --kernel-trace --pmc <SQ group> of bench.py --steps 1 --warmup 1 (tools/sq_profile.sh)", "kernels": res}, fh, indent=1, sort_keys=True)
  want = ("k_decode_cracks", "k_crack", "k_strip", "k_slice_resolve", "k_paint_strips", "k_trail_walk", "k_trail_items", "k_trail_segments", "k_trail_expand", "k_trail_graph", "k_trail_nodes", "k_label_planes", "k_trail_components", "k_finish")
  for k in sorted(res):
    if not k.startswith(want):
      continue
    r = res[k]
    print(k)
    print("   " + "  ".join(f"{c}={r[c]:.4g}" for c in sorted(r)))


if __name__ == "__main__":
  main()
