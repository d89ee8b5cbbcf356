"""Write-only and copy bandwidth of the device, for reading k_paint_runs' roofline fraction."""
import torch

def timed(fn, reps=10):
  fn(); torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(reps):
    fn()
  b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b) / reps

n = 1 << 31
x = torch.empty(n // 4, dtype=torch.int32, device="cuda")
y = torch.empty_like(x)
ms = timed(lambda: x.fill_(7))
print(f"fill 2 GiB: {ms:.3f} ms  {n / ms / 1e6:.0f} GB/s written")
ms = timed(lambda: x.zero_())
print(f"zero 2 GiB: {ms:.3f} ms  {n / ms / 1e6:.0f} GB/s written")
ms = timed(lambda: y.copy_(x))
print(f"copy 2 GiB: {ms:.3f} ms  {2 * n / ms / 1e6:.0f} GB/s read+written")
ms = timed(lambda: x.sum())
print(f"sum  2 GiB: {ms:.3f} ms  {n / ms / 1e6:.0f} GB/s read")
