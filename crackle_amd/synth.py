"""Seeded synthetic segmentation volumes.

Integer-only torch arithmetic with a counter-based hash, so the same call
returns bit-identical labels on CPU (this container) and on an MI355X.

Connectomics-style labels = jittered-grid 3-D Voronoi: one seed per lattice
cell of ``cell`` voxels, anisotropic metric (z distance x4), label =
1 + hash(cell) % modulus (+ offset).

Memory layout: tensors are returned with shape (sz, sy, sx), C-contiguous, i.e.
x fastest — byte-identical to a Fortran-ordered numpy array of shape
(sx, sy, sz), which is what crackle.compress consumes
(/root/reference/crackle/codec.py:723-724).
"""
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

_M32 = 0xFFFFFFFF


def _hash32(x: torch.Tensor) -> torch.Tensor:
  """32-bit integer mixer on int64 tensors (multiplier < 2^27: no int64 overflow)."""
  x = x & _M32
  x = (((x >> 16) ^ x) * 0x45D9F3B) & _M32
  x = (((x >> 16) ^ x) * 0x45D9F3B) & _M32
  x = ((x >> 16) ^ x) & _M32
  return x


def _torch_dtype(dtype) -> Tuple[torch.dtype, torch.dtype]:
  """(storage dtype used for arithmetic-free views, final dtype)"""
  dt = np.dtype(dtype)
  table = {
    np.dtype(np.uint8): (torch.uint8, torch.uint8),
    np.dtype(np.uint16): (torch.int16, torch.uint16),
    np.dtype(np.uint32): (torch.int32, torch.uint32),
    np.dtype(np.uint64): (torch.int64, torch.uint64),
  }
  if dt not in table:
    raise TypeError(f"unsupported dtype {dt}")
  return table[dt]


def voronoi_labels(
  shape: Sequence[int],
  dtype=np.uint32,
  seed: int = 0,
  cell: Sequence[int] = (32, 32, 8),
  modulus: Optional[int] = None,
  offset: int = 0,
  device="cpu",
  z_chunk: int = 8,
  z_range: Optional[Tuple[int, int]] = None,
) -> torch.Tensor:
  """Jittered-grid anisotropic Voronoi labels, tensor shape (sz, sy, sx).

  ``z_range=(z0, z1)`` returns only slices [z0, z1) of the (sx, sy, sz) volume —
  bit-identical to slicing the full result — so that each GPU of a sharded run can
  generate its own slab."""
  sx, sy, sz = (int(s) for s in shape)
  z_lo, z_hi = (0, sz) if z_range is None else (int(z_range[0]), int(z_range[1]))
  cx, cy, cz = (int(c) for c in cell)
  store_dt, final_dt = _torch_dtype(dtype)
  itemsize = np.dtype(dtype).itemsize
  if modulus is None:
    modulus = {1: 250, 2: 60000, 4: 1 << 30, 8: 1 << 30}[itemsize]
  dev = torch.device(device)
  out = torch.empty((max(z_hi - z_lo, 0), sy, sx), dtype=store_dt, device=dev)
  if sx * sy * max(z_hi - z_lo, 0) == 0:
    return out.view(final_dt)

  # lattice of cells covering the volume plus a one-cell halo
  nx, ny, nz = sx // cx + 3, sy // cy + 3, sz // cz + 3
  X = torch.arange(-1, nx - 1, device=dev, dtype=torch.int64)
  Y = torch.arange(-1, ny - 1, device=dev, dtype=torch.int64)
  Z = torch.arange(-1, nz - 1, device=dev, dtype=torch.int64)
  cid = (
    ((Z + 7)[:, None, None] * 1000003)
    ^ ((Y + 7)[None, :, None] * 10007)
    ^ ((X + 7)[None, None, :] * 101)
  ) + (int(seed) & 0xFFFF) * 7919
  h = _hash32(cid)
  seed_x = (X[None, None, :] * cx + (_hash32(h + 1) % cx)).to(torch.int32)
  seed_y = (Y[None, :, None] * cy + (_hash32(h + 2) % cy)).to(torch.int32)
  seed_z = (Z[:, None, None] * cz + (_hash32(h + 3) % cz)).to(torch.int32)
  seed_x = seed_x.expand(nz, ny, nx).contiguous()
  seed_y = seed_y.expand(nz, ny, nx).contiguous()
  seed_z = seed_z.expand(nz, ny, nx).contiguous()
  tie = (h & 0xFFF).to(torch.int64)
  lab = (1 + (_hash32(h + 4) % int(modulus)) + int(offset)).to(torch.int64)

  xs = torch.arange(sx, device=dev, dtype=torch.int32)
  ys = torch.arange(sy, device=dev, dtype=torch.int32)
  cxi = (xs // cx + 1).to(torch.int64)   # +1: halo offset into the lattice
  cyi = (ys // cy + 1).to(torch.int64)

  for z0 in range(z_lo, z_hi, z_chunk):
    z1 = min(z_hi, z0 + z_chunk)
    zs = torch.arange(z0, z1, device=dev, dtype=torch.int32)
    czi = (zs // cz + 1).to(torch.int64)
    best_key = None
    best_lab = None
    for dz in (-1, 0, 1):
      for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
          idx = (
            ((czi + dz) * ny)[:, None, None] + (cyi + dy)[None, :, None]
          ) * nx + (cxi + dx)[None, None, :]
          ddx = (xs[None, None, :] - seed_x.view(-1)[idx]).to(torch.int64)
          ddy = (ys[None, :, None] - seed_y.view(-1)[idx]).to(torch.int64)
          ddz = (zs[:, None, None] - seed_z.view(-1)[idx]).to(torch.int64) * 4
          key = ((ddx * ddx + ddy * ddy + ddz * ddz) << 12) | tie.view(-1)[idx]
          cand = lab.view(-1)[idx]
          if best_key is None:
            best_key, best_lab = key, cand
          else:
            take = key < best_key
            best_key = torch.where(take, key, best_key)
            best_lab = torch.where(take, cand, best_lab)
    out[z0 - z_lo:z1 - z_lo] = best_lab.to(store_dt)
  return out.view(final_dt)


def as_numpy_f(t: torch.Tensor) -> np.ndarray:
  """(sz, sy, sx) C-contiguous tensor -> Fortran-ordered numpy (sx, sy, sz) view."""
  signed = {torch.uint16: torch.int16, torch.uint32: torch.int32, torch.uint64: torch.int64}
  np_dt = {torch.uint8: np.uint8, torch.uint16: np.uint16, torch.uint32: np.uint32, torch.uint64: np.uint64}[t.dtype]
  tt = t.detach().cpu()
  if tt.dtype in signed:
    tt = tt.view(signed[tt.dtype])
  return tt.numpy().view(np_dt).transpose(2, 1, 0)


def random_labels_device(shape, dtype=np.uint32, seed=0, high=2000, device="cpu", z_range=None) -> torch.Tensor:
  """Uniform-random labels in [0, high) as a tensor of shape (sz, sy, sx) on `device`: the same values as
  random_labels (voxel i = x + sx * (y + sy * z) hashed), generated where they are needed (bench.py --data
  noise2000 / binary: the reference's adversarial inputs, benchmarks/README.md:108-114, 193-227)."""
  sx, sy, sz = (int(s) for s in shape)
  z0, z1 = (0, sz) if z_range is None else (int(z_range[0]), int(z_range[1]))
  store_dt, final_dt = _torch_dtype(dtype)
  out = torch.empty((z1 - z0, sy, sx), dtype=store_dt, device=device)
  for z in range(z0, z1):      # slice by slice: the int64 hash of a whole volume would take 8 bytes per voxel
    i = torch.arange(z * sx * sy, (z + 1) * sx * sy, dtype=torch.int64, device=device)
    v = _hash32(i * 3 + (int(seed) & 0xFFFF) * 7919) % int(high)
    out[z - z0] = v.reshape(sy, sx).to(store_dt)
  return out.view(final_dt)


def random_labels(shape, dtype=np.uint32, seed=0, high=2000) -> np.ndarray:
  """Uniform-random labels in [0, high) (adversarial PERMISSIBLE case), F-ordered numpy."""
  sx, sy, sz = (int(s) for s in shape)
  i = torch.arange(sx * sy * sz, dtype=torch.int64)
  v = _hash32(i * 3 + (int(seed) & 0xFFFF) * 7919) % int(high)
  return v.numpy().astype(dtype).reshape((sx, sy, sz), order="F")
