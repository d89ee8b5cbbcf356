import hashlib
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

_golden = None
_manifest = None


def golden():
  global _golden
  if _golden is None:
    z = np.load(os.path.join(HERE, "golden", "golden.npz"))
    _golden = {k: z[k].tobytes() for k in z.files}
  return _golden


def manifest():
  global _manifest
  if _manifest is None:
    with open(os.path.join(HERE, "golden", "manifest.json")) as f:
      _manifest = json.load(f)
  return _manifest


def label_format(binary: bytes) -> int:
  return (int.from_bytes(binary[5:7], "little") >> 5) & 3


def flat_1d(arr: np.ndarray) -> np.ndarray:
  """What fastcrackle.decompress returns for a stream made from `arr` (x fastest for F inputs)."""
  return arr.reshape(-1, order="F" if arr.flags.f_contiguous else "C")


def sha(b: bytes) -> str:
  return hashlib.sha256(b).hexdigest()
