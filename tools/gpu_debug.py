"""Diagnostic sweep on the GPU box: every golden case through the HIP encode and
decode, reporting mismatches instead of stopping at the first (not a test)."""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import crackle_amd  # noqa: E402
import golden_cases  # noqa: E402
from util import golden, manifest, label_format, flat_1d, sha  # noqa: E402


def first_diff(a: bytes, b: bytes):
  n = min(len(a), len(b))
  for i in range(n):
    if a[i] != b[i]:
      return i
  return n if len(a) != len(b) else -1


def section_of(binary: bytes, off: int):
  sz = int.from_bytes(binary[15:19], "little")
  nlb = int.from_bytes(binary[20:28], "little")
  fmt = int.from_bytes(binary[5:7], "little")
  order = (fmt >> 9) & 15
  mb = ((4 ** order) * 5 + 4) // 8 if order else 0
  bounds = [("header", 29), ("z_index", 4 * (sz + 1)), ("labels", nlb), ("model", mb)]
  o = 0
  for name, ln in bounds:
    if off < o + ln:
      return f"{name}+{off - o}"
    o += ln
  tail = len(binary) - 4 * (sz + 1)
  if off < tail:
    return f"cracks+{off - o}"
  return f"crcs+{off - tail}"


def main():
  only = sys.argv[1:] 
  small = golden_cases.small_cases()
  G = golden()
  dec_bad, enc_bad, n = [], [], 0
  t0 = time.time()
  for name in sorted(small):
    if only and not any(o in name for o in only):
      continue
    arr, kw = small[name]
    want = G[name]
    n += 1
    # decode the reference's bytes
    if arr.size:
      try:
        got = crackle_amd.decompress(want)
        exp = arr
        if got.shape != exp.shape or not np.array_equal(got, exp):
          bad = int(np.count_nonzero(got != exp)) if got.shape == exp.shape else -1
          dec_bad.append((name, f"{bad} voxels differ"))
      except Exception as e:
        dec_bad.append((name, f"{type(e).__name__}: {e}"))
    # encode and compare bytes
    try:
      b = crackle_amd.compress(arr, allow_pins=int(kw["allow_pins"]), markov_model_order=kw["markov_model_order"])
      if b != want:
        d = first_diff(b, want)
        enc_bad.append((name, f"len {len(b)} vs {len(want)}, first diff @{d} ({section_of(want, d) if d >= 0 else ''})"))
    except Exception as e:
      enc_bad.append((name, f"{type(e).__name__}: {e}"))
  print(f"cases {n}  decode failures {len(dec_bad)}  encode failures {len(enc_bad)}  ({time.time() - t0:.1f}s)")
  for nm, msg in dec_bad[:40]:
    print("  DEC", nm, msg)
  for nm, msg in enc_bad[:60]:
    print("  ENC", nm, msg)

  if not only:
    for name, (thunk, kw) in sorted(golden_cases.large_cases().items()):
      arr = thunk()
      m = manifest()[name]
      try:
        t = time.time()
        b = crackle_amd.compress(arr, allow_pins=int(kw["allow_pins"]), markov_model_order=kw["markov_model_order"])
        te = time.time() - t
        ok = (sha(b) == m["sha256"])
        t = time.time()
        back = crackle_amd.decompress(b)
        td = time.time() - t
        print(f"  LARGE {name}: bytes {'OK' if ok else 'MISMATCH'} ({len(b)} vs {m['length']}), roundtrip {'OK' if np.array_equal(back, arr) else 'MISMATCH'}  enc {te:.3f}s dec {td:.3f}s")
      except Exception as e:
        print(f"  LARGE {name}: {type(e).__name__}: {e}")


if __name__ == "__main__":
  try:
    main()
  except Exception:
    traceback.print_exc()
    sys.exit(1)
