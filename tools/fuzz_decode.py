"""GPU robustness aid: decodes corrupted copies of golden / seeded streams; every call must
return (a volume or an error) — no fault, no hang.  Run under `timeout`."""
import sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import crackle_amd
from crackle_amd import synth
from util import golden
from oracle import oracle


def main():
  trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
  rng = np.random.default_rng(7)
  chk = oracle.best()
  streams = [golden()[k] for k in sorted(golden()) if k.startswith("c0_") or "spur" in k or "checker" in k][:12]
  vol = synth.as_numpy_f(synth.voronoi_labels((256, 192, 4), np.uint32, seed=71, cell=(16, 16, 4)))
  streams += [chk.compress(vol, markov_model_order=m) for m in (0, 3)]
  streams += [chk.compress(synth.random_labels((96, 96, 2), np.uint8, seed=72, high=2), markov_model_order=2)]
  pin_vol = synth.as_numpy_f(synth.voronoi_labels((128, 96, 24), np.uint32, seed=73, cell=(8, 8, 4)))
  pin_streams = [chk.compress(pin_vol, allow_pins=True, markov_model_order=m) for m in (0, 2)]
  assert all(crackle_amd.header(b).label_format == 2 for b in pin_streams)
  streams += pin_streams
  if len(sys.argv) > 2 and sys.argv[2] == "pins":      # only pin streams, damaged inside their label section (the decoder's parallel parse)
    streams = pin_streams
  ok = err = ok2 = err2 = 0
  wide = len(sys.argv) > 2 and sys.argv[2] == "wide"
  for t in range(trials):
    b = bytearray(streams[t % len(streams)])
    n = len(b)
    for _ in range(int(rng.integers(1, 4))):
      # keep the header intact (its crc8 rejects damage early): hit index, labels, codes, crcs
      pos = int(rng.integers(29, n))
      if len(sys.argv) > 2 and sys.argv[2] == "pins":
        h = crackle_amd.header(bytes(streams[t % len(streams)]))
        lo = 29 + 4 * (h.sz + 1)
        pos = int(rng.integers(lo, lo + h.num_label_bytes))
      b[pos] ^= 1 << int(rng.integers(0, 8))
    try:
      crackle_amd.decompress(bytes(b))
      ok += 1
    except (RuntimeError, ValueError, crackle_amd.FormatError):
      err += 1
    if wide:
      # the consumers of the decode path must come back as well (an answer or an error)
      for fn in (lambda: crackle_amd.voxel_counts(bytes(b)), lambda: crackle_amd.bounding_boxes(bytes(b)),
                 lambda: crackle_amd.voxel_connectivity_graph(bytes(b), 6),
                 lambda: crackle_amd.reencode(bytes(b), 1 if crackle_amd.header(bytes(b)).markov_model_order != 1 else 0),
                 lambda: crackle_amd.point_cloud(bytes(b), skip_background=False),
                 lambda: crackle_amd.array_equal(bytes(b), streams[t % len(streams)])):
        try:
          fn()
          ok2 += 1
        except (RuntimeError, ValueError, KeyError, crackle_amd.FormatError):
          err2 += 1
    if t % 50 == 49:
      print(f"trial {t + 1}: decoded {ok}, rejected {err}", flush=True)
  print(f"done: {trials} corrupted streams, decoded {ok}, rejected {err}, no faults" + (f"; statistics / vcg / reencode / point_cloud / array_equal calls: {ok2} answered, {err2} rejected" if wide else ""))


if __name__ == "__main__":
  main()
