"""Prints the kernel timeline of the last encode in a rocprofv3 --kernel-trace csv:
start offset, duration and queue of every kernel between the last k_label_planes* launch and the
next k_decode_cracks, with the idle gaps of the queue that carries the crack trail."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if "k_label_planes" in n]
first = starts[-1]
last = next(i for i in range(first, len(rows)) if ("k_crack_match" in names[i] or "k_decode_cracks" in names[i]))
t0 = int(rows[first]["Start_Timestamp"])
trail_q = next(r["Queue_Id"] for r in rows[first:last] if "k_trail_walk" in r["Kernel_Name"])
prev_end = None
for r in rows[first:last]:
  s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
  q = r["Queue_Id"]
  gap = ""
  if q == trail_q:
    if prev_end is not None:
      gap = f"gap {1e-3 * (s - prev_end):7.1f} us"
    prev_end = e
  n = r["Kernel_Name"].split("(")[0][:60]
  print(f"{1e-6 * (s - t0):8.3f} ms  {1e-3 * (e - s):8.1f} us  q{q:>3s}  {n:60s} {gap}")
