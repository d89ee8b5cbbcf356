#!/usr/bin/env python3
"""Folds two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter per pass, see
/opt/skills/guides/MI355X_MICROARCH.md, HBM / PMC slots) of `bench.py` into
profiles/r03_pmc_traffic.json: memory-side bytes per launch of every ckl kernel.

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
  python3 tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write "1024x1024x512 uint32 markov 0" profiles/r03_pmc_traffic.json

Units and corrections (the guide's HBM section): both counters are reported in KiB
(TCC_EA0_RDREQ / WRREQ scaled by the request size); on gfx950 FETCH_SIZE tallies the
128-byte requests of wide coalesced reads at 64 bytes, so it is doubled; WRITE_SIZE is
exact for 16-byte-per-lane streaming stores.  Narrow / scattered accesses are
uncalibrated: treat those rows as lower bounds.  Infinity-Cache hits are counted.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(dirname, counter):
  per = defaultdict(list)
  files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
  if not files:
    raise SystemExit(f"no counter_collection.csv under {dirname}")
  for f in files:
    disp = defaultdict(float)
    names = {}
    with open(f, newline="") as fh:
      for row in csv.DictReader(fh):
        if row.get("Counter_Name") != counter:
          continue
        key = row.get("Dispatch_Id") or row.get("Correlation_Id")
        disp[key] += float(row["Counter_Value"])
        names[key] = row["Kernel_Name"]
    for key, v in disp.items():
      per[names[key]].append(v)
  return per


def short(name):
  n = name.split("(")[0]
  for p in ("void ", "ckl::dev::", "ckl::"):
    n = n.replace(p, "")
  # keep the template arguments: the encoder and the decoder instantiate some kernels differently
  return n.replace("unsigned char", "u8").replace("unsigned short", "u16").replace("unsigned int", "u32").replace("unsigned long", "u64").strip()


def main():
  fetch_dir, write_dir, workload, out = sys.argv[1:5]
  fetch = load(fetch_dir, "FETCH_SIZE")
  write = load(write_dir, "WRITE_SIZE")
  kernels = {}
  for name in sorted(set(fetch) | set(write)):
    if "ckl" not in name:
      continue
    f = fetch.get(name, [])
    w = write.get(name, [])
    # the warm-up launch is dropped when there is more than one
    fm = sum(f[1:]) / len(f[1:]) if len(f) > 1 else (f[0] if f else 0.0)
    wm = sum(w[1:]) / len(w[1:]) if len(w) > 1 else (w[0] if w else 0.0)
    kernels.setdefault(short(name), {"launches": 0, "fetch_KiB_raw": 0.0, "write_KiB_raw": 0.0})
    k = kernels[short(name)]
    k["launches"] = max(k["launches"], len(f), len(w))
    k["fetch_KiB_raw"] += fm
    k["write_KiB_raw"] += wm
  for k in kernels.values():
    k["hbm_bytes_per_launch"] = (2.0 * k["fetch_KiB_raw"] + k["write_KiB_raw"]) * 1024.0
  sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  from crackle_amd import build as ckl_build
  sha16 = ckl_build.source_digest()
  res = {
    "workload": workload,
    "lib_sha16": sha16,      # digest of the library's sources: bench.py takes roofline.traffic from this file only while they are unchanged
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py --steps 1 --warmup 1",
    "correction": "bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB: gfx950 counts 128-byte read requests at 64 bytes",
    "kernels": kernels,
  }
  with open(out, "w") as fh:
    json.dump(res, fh, indent=1, sort_keys=True)
  for n, k in sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]):
    print(f"{n:28s} {k['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch")


if __name__ == "__main__":
  main()
