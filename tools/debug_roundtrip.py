"""GPU debugging aid: device-resident encode -> decode of a seeded volume, mismatch report per slice."""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from crackle_amd import _lib, synth

def main():
  shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1024x1024x32").split("x"))
  L = _lib.lib()
  dev = torch.device("cuda:0")
  vol = synth.voronoi_labels(shape, np.uint32, seed=2, device=dev)
  sz, sy, sx = vol.shape
  enc = C.c_void_p()
  assert L.ckl_encoder_create(sx, sy, sz, 4, 0, C.byref(enc)) == 0
  outs = []
  for _ in range(2):
    out, n = C.c_void_p(), C.c_uint64()
    rc = L.ckl_encoder_run(enc, vol.data_ptr(), sx, sy, sz, 0, 1, 0, 0, 1, 0, None, C.byref(out), C.byref(n))
    assert rc == 0, _lib.last_error()
    outs.append(C.string_at(out.value, n.value))
    L.ckl_free(out)
  L.ckl_encoder_destroy(enc)
  print("encodes equal:", outs[0] == outs[1], len(outs[0]))
  binary = outs[0]
  try:
    from oracle import oracle
    chk = oracle.best()
    want = chk.compress(synth.as_numpy_f(vol), markov_model_order=0)
    print("bytes equal to", chk.kind, ":", want == binary)
    if want != binary:
      compare_streams(binary, want, sz)
  except Exception as ex:
    print("oracle unavailable:", ex)
  dec = C.c_void_p()
  assert L.ckl_decoder_create(binary, len(binary), 0, -1, 0, C.byref(dec)) == 0, _lib.last_error()
  back = torch.empty_like(vol)
  for it in range(3):
    back.zero_()
    torch.cuda.synchronize()
    rc = L.ckl_decoder_run(dec, back.data_ptr(), back.numel() * 4, 0, 0)
    torch.cuda.synchronize()
    bad = (back.view(torch.int32) != vol.view(torch.int32))
    per = bad.reshape(sz, -1).sum(dim=1).tolist()
    print("iter", it, "rc", rc, "mismatching voxels per slice:", {z: c for z, c in enumerate(per) if c})
    if bad.any():
      z = [z for z, c in enumerate(per) if c][0]
      rows = bad[z].sum(dim=1).nonzero().flatten().tolist()
      print("  slice", z, "rows with mismatches:", rows[:5], "...", rows[-5:], "count", len(rows))
      zero = (back[z] == 0).sum().item()
      print("  zeros in that slice:", zero)
  L.ckl_decoder_destroy(dec)



def compare_streams(a: bytes, b: bytes, sz: int):
  """a: ours, b: reference.  Prints the first section that differs."""
  import struct
  print("lengths", len(a), len(b), "header equal", a[:29] == b[:29])
  nlb_a = struct.unpack_from("<Q", a, 20)[0]; nlb_b = struct.unpack_from("<Q", b, 20)[0]
  print("num_label_bytes", nlb_a, nlb_b)
  za = np.frombuffer(a, dtype="<u4", count=sz, offset=29); zb = np.frombuffer(b, dtype="<u4", count=sz, offset=29)
  print("z-index equal", np.array_equal(za, zb), "first diff", np.nonzero(za != zb)[0][:8], za[:4], zb[:4])
  la = a[29 + 4 * (sz + 1): 29 + 4 * (sz + 1) + nlb_a]; lb = b[29 + 4 * (sz + 1): 29 + 4 * (sz + 1) + nlb_b]
  print("labels equal", la == lb)
  if la != lb:
    n = min(len(la), len(lb))
    d = np.nonzero(np.frombuffer(la[:n], np.uint8) != np.frombuffer(lb[:n], np.uint8))[0]
    print("  first label byte diffs at", d[:10], "uniq counts", struct.unpack_from("<Q", la, 0)[0], struct.unpack_from("<Q", lb, 0)[0])
  oa = 29 + 4 * (sz + 1) + nlb_a; ob = 29 + 4 * (sz + 1) + nlb_b
  for z in range(sz):
    ca = a[oa: oa + za[z]]; cb = b[ob: ob + zb[z]]
    if ca != cb:
      print("  crack code differs first at slice", z, "len", len(ca), len(cb))
      break
    oa += za[z]; ob += zb[z]
  ta = np.frombuffer(a[-4 * sz:], "<u4"); tb = np.frombuffer(b[-4 * sz:], "<u4")
  print("slice crcs equal", np.array_equal(ta, tb), np.nonzero(ta != tb)[0][:8])


if __name__ == "__main__":
  main()
