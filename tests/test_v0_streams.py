"""Format version 0 streams (24-byte header with a 4-byte num_label_bytes and no crc8, z-index without its crc32c,
no crc tail: /root/reference/src/header.hpp:113-129, 168-183; src/crackle.hpp:276, 566, 599).
tests/golden/v0.npz holds golden streams rewritten to that layout by tests/gen_golden.py --v0 (byte surgery), the
sha256 of what the compiled REFERENCE decodes each of them to, and what the reference's reencode returns for them.

The reference's reencode of a version 0 stream under a new markov order is broken: CrackleHeader::tobytes()
allocates 29 bytes and tochars() writes 24 of them (src/header.hpp:277-281), so five zero bytes follow the header
of the stream it returns (src/crackle.hpp:956).  The fixture pins that, and that the stream is the intended one once
the five bytes are taken out — which is what ckl_reencode_markov writes."""
import hashlib
import os

import numpy as np
import pytest

import crackle_amd

HERE = os.path.dirname(os.path.abspath(__file__))


def _fixtures():
  with np.load(os.path.join(HERE, "golden", "v0.npz")) as z:
    names = [k for k in z.files if "." not in k]
    return {k: (z[k].tobytes(), z[k + ".decoded_sha256"].tobytes().decode(),
                z[k + ".ref_reencode_m0"].tobytes() if k + ".ref_reencode_m0" in z.files else None) for k in names}


FIX = _fixtures()


def _sha(a):
  return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _unstray(r: bytes) -> bytes:
  """the reference's reencoded version 0 stream without the five stray bytes behind its header"""
  assert r[4] == 0 and r[24:29] == b"\x00" * 5
  return r[:24] + r[29:]


@pytest.mark.parametrize("name", sorted(FIX))
def test_checkers_decode_version_0(name, checker):
  b0, want, _ = FIX[name]
  assert b0[4] == 0
  head = crackle_amd.header(b0)
  assert head.format_version == 0
  assert _sha(checker.decompress(b0)) == want


@pytest.mark.parametrize("name", sorted(FIX))
def test_reference_reencode_of_version_0_leaves_five_stray_bytes(name, checker):
  b0, want, r = FIX[name]
  if r is None:
    pytest.skip("no reencode sample")
  order0 = (int.from_bytes(b0[5:7], "little") >> 9) & 15
  if order0 == 0:
    assert r == b0      # same order: returned as it came (src/crackle.hpp:887-889)
    return
  fixed = _unstray(r)
  assert crackle_amd.header(fixed).markov_model_order == 0
  assert _sha(checker.decompress(fixed)) == want


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(FIX))
def test_device_decodes_version_0(name, checker):
  b0, want, _ = FIX[name]
  got = crackle_amd.decompress(b0)
  head = crackle_amd.header(b0)
  ref = np.asarray(checker.decompress(b0)).ravel()
  assert _sha(ref) == want
  assert np.array_equal(got.ravel(order="F" if head.fortran_order else "C"), ref)
  if head.sz > 2:
    part = crackle_amd.decompress_range(b0, 1, head.sz - 1)
    assert np.array_equal(part, got[:, :, 1:head.sz - 1])


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(FIX))
def test_device_reencode_keeps_version_0(name, checker):
  b0, want, r = FIX[name]
  order0 = (int.from_bytes(b0[5:7], "little") >> 9) & 15
  assert crackle_amd.reencode(b0, order0) == b0
  for order in (0, 2, 5):
    if order == order0:
      continue
    out = crackle_amd.reencode(b0, order)
    head = crackle_amd.header(out)
    assert head.format_version == 0 and head.markov_model_order == order
    assert _sha(checker.decompress(out)) == want, (name, order)
    if order == 0 and r is not None:
      assert out == _unstray(r), "differs from the reference's stream (its five stray bytes taken out)"
