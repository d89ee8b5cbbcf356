"""Generate the golden fixtures by running the REFERENCE ITSELF (compiled in place
into oracle/_ref by `make -C oracle ref`).  Runs only in the build container, where
/root/reference exists; its outputs (data, not code) are committed:

  tests/golden/golden.npz      exact .ckl bytes for every small case
  tests/golden/manifest.json   sha256/length/section hashes for larger generated cases

usage: python tests/gen_golden.py [--xl | --ops | --v0]
  --xl   writes tests/golden/manifest_xl.json (full-size BASELINE.json configurations)
  --ops  writes tests/golden/point_cloud.json (the reference's point_cloud on the small golden
         streams: per stream and argument set a sha256 over labels, offsets and points),
         tests/golden/label_stats.json (voxel_counts / centroids / bounding_boxes digests of the
         same streams) and the whole-C1 digests of ops_xl.json / point_cloud_xl.json
  --v0   writes tests/golden/v0.npz: golden streams rewritten as format version 0 (24-byte header with a
         4-byte num_label_bytes and no crc8, z-index without its crc32c, no crc tail: src/header.hpp:113-129,
         168-183, src/crackle.hpp:276, 566, 599) together with the sha256 of what the REFERENCE decodes them to
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
  path = os.path.join(HERE, "golden", "manifest_xl.json")
  only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]      # --only=NAME: (re)generate one entry, keep the others
  manifest = {}
  if only and os.path.exists(path):
    with open(path) as f:
      manifest = json.load(f)
  for name, (thunk, kw) in golden_cases.xl_cases().items():
    if only and name not in only:
      continue
    arr = thunk()
    b = ref.compress(arr, parallel=8, **kw)
    manifest[name] = manifest_entry(arr, b)
    print(name, len(b), manifest[name]["sha256"], flush=True)
    del arr, b
  with open(path, "w") as f:
    json.dump(manifest, f, indent=1, sort_keys=True)


def point_cloud_digest(ptc) -> str:
  """sha256 over the labels (ascending) and each label's flat uint16 (x, y, z) triples."""
  h = hashlib.sha256()
  for k in sorted(ptc):
    h.update(int(k).to_bytes(8, "little"))
    h.update(int(ptc[k].size).to_bytes(8, "little"))
    h.update(np.ascontiguousarray(ptc[k], dtype="<u2").tobytes())
  return h.hexdigest()


def stats_digest(m) -> str:
  """sha256 over a label -> value map of voxel_counts / centroids / bounding_boxes: labels ascending,
  each followed by its value's little-endian bytes (uint64 count, 3 float64, 6 uint32)."""
  h = hashlib.sha256()
  for k in sorted(m):
    h.update(int(k).to_bytes(8, "little"))
    v = m[k]
    h.update(int(v).to_bytes(8, "little") if np.isscalar(v) or isinstance(v, int) else np.ascontiguousarray(v).astype(np.asarray(v).dtype.newbyteorder("<")).tobytes())
  return h.hexdigest()


STATS_RANGES = {"all": (0, -1), "z1": (1, 2)}


def label_stats_entry(chk, stream):
  """digests of the three statistics over each z-range of STATS_RANGES (or the error text)"""
  entry = {}
  for tag, (z0, z1) in STATS_RANGES.items():
    for fn in ("voxel_counts", "centroids", "bounding_boxes"):
      try:
        entry[f"{fn}.{tag}"] = stats_digest(getattr(chk, fn)(stream, z0, z1))
      except RuntimeError as exc:
        entry[f"{fn}.{tag}"] = "error: " + str(exc)
  return entry


POINT_CLOUD_ARGS = {"all": (0, -1, None, False), "skip0": (0, -1, None, True), "z1": (1, 2, None, False)}


def main_ops(ref):
  with np.load(os.path.join(HERE, "golden", "golden.npz")) as z:
    streams = {k: z[k].tobytes() for k in z.files}
  out = {}
  for name in sorted(streams):
    entry = {}
    for tag, (z0, z1, labels, skip) in POINT_CLOUD_ARGS.items():
      try:
        entry[tag] = point_cloud_digest(ref.point_cloud(streams[name], z0, z1, labels, skip))
      except RuntimeError as exc:
        entry[tag] = "error: " + str(exc)
    out[name] = entry
  with open(os.path.join(HERE, "golden", "point_cloud.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
  print("point_cloud:", len(out), "streams")
  # voxel_counts / centroids / bounding_boxes (operations.hpp:321-665) of the same streams
  stats = {name: label_stats_entry(ref, streams[name]) for name in sorted(streams)}
  with open(os.path.join(HERE, "golden", "label_stats.json"), "w") as f:
    json.dump(stats, f, indent=1, sort_keys=True)
  print("label_stats:", len(stats), "streams")
  # the same for BASELINE.json configs[1] at full size (512 x 512 x 128 uint32)
  thunk, kw = golden_cases.xl_cases()["c1_512x512x128_u32"]
  stream = ref.compress(thunk(), parallel=8, **kw)
  big = {tag: point_cloud_digest(ref.point_cloud(stream, z0, z1 if tag != "z1" else 65, labels, skip)) for tag, (z0, z1, labels, skip) in POINT_CLOUD_ARGS.items()}
  with open(os.path.join(HERE, "golden", "point_cloud_xl.json"), "w") as f:
    json.dump({"c1_512x512x128_u32": big}, f, indent=1, sort_keys=True)
  print("point_cloud xl:", big)
  # the other consumers on the same stream: reencode, mode pooling, voxel connectivity graph
  ops = {
    "reencode_m3": hashlib.sha256(ref.reencode(stream, 3)).hexdigest(),
    "reencode_m0_of_m3": hashlib.sha256(ref.reencode(ref.reencode(stream, 3), 0)).hexdigest(),
    "mode_pooling_2x2x1": hashlib.sha256(b"".join(ref.mode_pooling_2x2x1(stream))).hexdigest(),
    "vcg4": hashlib.sha256(np.ascontiguousarray(ref.voxel_connectivity_graph(stream, 4)).tobytes()).hexdigest(),
    "vcg6": hashlib.sha256(np.ascontiguousarray(ref.voxel_connectivity_graph(stream, 6)).tobytes()).hexdigest(),
  }
  ops.update(label_stats_entry(ref, stream))
  with open(os.path.join(HERE, "golden", "ops_xl.json"), "w") as f:
    json.dump({"c1_512x512x128_u32": ops}, f, indent=1, sort_keys=True)
  print("ops xl:", ops)


def to_v0(binary: bytes) -> bytes:
  """The same stream as the reference's version 0 layout (pure byte surgery, no re-encoding)."""
  sec = sections(binary)
  h = sec["header"]
  assert h[4] == 1
  nlb = int.from_bytes(h[20:28], "little")
  assert nlb < (1 << 32)
  head0 = h[:4] + b"\x00" + h[5:20] + nlb.to_bytes(4, "little")
  sz = int.from_bytes(h[15:19], "little")
  return head0 + sec["z_index"][:4 * sz] + sec["labels"] + sec["model"] + sec["cracks"]


V0_CASES = ["c0_voronoi_u8", "c0_voronoi_u8_m5", "c0_voronoi_u8_pins_m5", "c0_voronoi_u8_c", "rand_17x13x5_uint64_F_m3_p1", "noise_2000", "kat_4x4", "single_voxel"]


def main_v0(ref):
  with np.load(os.path.join(HERE, "golden", "golden.npz")) as z:
    names = [n for n in V0_CASES if n in z.files] or sorted(z.files)[:6]
    streams = {k: z[k].tobytes() for k in names}
  out = {}
  for name, b1 in streams.items():
    b0 = to_v0(b1)
    a1 = ref.decompress(b1)
    a0 = ref.decompress(b0)      # the reference itself reads the version 0 stream
    assert np.array_equal(a0, a1), name
    out[name] = np.frombuffer(b0, dtype=np.uint8)
    out[name + ".decoded_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(a0).tobytes()).hexdigest().encode(), dtype=np.uint8)
    try:      # what the reference's reencode makes of it (src/crackle.hpp:947-984)
      r = ref.reencode(b0, 0)
      out[name + ".ref_reencode_m0"] = np.frombuffer(bytes(r), dtype=np.uint8)
    except RuntimeError as exc:
      out[name + ".ref_reencode_error"] = np.frombuffer(str(exc).encode(), dtype=np.uint8)
    print(name, len(b1), "->", len(b0))
  np.savez_compressed(os.path.join(HERE, "golden", "v0.npz"), **out)


def main():
  ref = oracle.ref()
  assert ref is not None, "build oracle/_ref first (make -C oracle ref)"
  if "--xl" in sys.argv:
    return main_xl(ref)
  if "--v0" in sys.argv:
    return main_v0(ref)
  if "--ops" in sys.argv:
    return main_ops(ref)
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
