"""Generate the golden fixtures by running the REFERENCE ITSELF (compiled in place
into oracle/_ref by `make -C oracle ref`).  Runs only in the build container, where
/root/reference exists; its outputs (data, not code) are committed:

  tests/golden/golden.npz      exact .ckl bytes for every small case
  tests/golden/manifest.json   sha256/length/section hashes for larger generated cases

usage: python tests/gen_golden.py [--xl]
  --xl  writes tests/golden/manifest_xl.json (full-size BASELINE.json configurations)
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from oracle import oracle  # noqa: E402
import golden_cases  # noqa: E402


def sections(binary: bytes):
  """Split a v1 stream into its sections for per-section hashes."""
  sz = int.from_bytes(binary[15:19], "little")
  if len(binary) == 29:
    return {"header": binary}
  nlb = int.from_bytes(binary[20:28], "little")
  fmt = int.from_bytes(binary[5:7], "little")
  order = (fmt >> 9) & 15
  mb = ((4 ** order) * 5 + 4) // 8 if order else 0
  o = 29
  out = {"header": binary[:29]}
  out["z_index"] = binary[o:o + 4 * (sz + 1)]; o += 4 * (sz + 1)
  out["labels"] = binary[o:o + nlb]; o += nlb
  out["model"] = binary[o:o + mb]; o += mb
  tail = 4 * (sz + 1)
  out["cracks"] = binary[o:len(binary) - tail]
  out["crcs"] = binary[len(binary) - tail:]
  return out


def manifest_entry(arr, b):
  return {
    "length": len(b),
    "sha256": hashlib.sha256(b).hexdigest(),
    "input_sha256": hashlib.sha256(np.asfortranarray(arr).tobytes(order="F")).hexdigest(),
    "sections": {k: hashlib.sha256(v).hexdigest() for k, v in sections(b).items()},
  }


def main_xl(ref):
  """Full-size BASELINE.json configurations (minutes of CPU time, gigabytes of memory)."""
  manifest = {}
  for name, (thunk, kw) in golden_cases.xl_cases().items():
    arr = thunk()
    b = ref.compress(arr, parallel=8, **kw)
    manifest[name] = manifest_entry(arr, b)
    print(name, len(b), manifest[name]["sha256"], flush=True)
    del arr, b
  with open(os.path.join(HERE, "golden", "manifest_xl.json"), "w") as f:
    json.dump(manifest, f, indent=1, sort_keys=True)


def main():
  ref = oracle.ref()
  assert ref is not None, "build oracle/_ref first (make -C oracle ref)"
  if "--xl" in sys.argv:
    return main_xl(ref)
  blobs = {}
  for name, (arr, kw) in golden_cases.small_cases().items():
    blobs[name] = np.frombuffer(ref.compress(arr, parallel=2, **kw), dtype=np.uint8)
  np.savez_compressed(os.path.join(HERE, "golden", "golden.npz"), **blobs)
  manifest = {}
  for name, (thunk, kw) in golden_cases.large_cases().items():
    arr = thunk()
    b = ref.compress(arr, parallel=4, **kw)
    manifest[name] = manifest_entry(arr, b)
  with open(os.path.join(HERE, "golden", "manifest.json"), "w") as f:
    json.dump(manifest, f, indent=1, sort_keys=True)
  print("small:", len(blobs), "cases,", sum(v.size for v in blobs.values()), "bytes; large:", len(manifest))


if __name__ == "__main__":
  main()
