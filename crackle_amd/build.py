"""Builds crackle_amd/libcrackle_amd.so (HIP kernels + C-ABI) for gfx950 with hipcc.

  python -m crackle_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is built in-tree so that it travels
with the repository snapshot to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcrackle_amd.so")
SOURCES = ["ckl_common.hip", "ckl_decode.hip", "ckl_encode.hip", "ckl_pins.hip", "ckl_zstack.hip", "ckl_upload.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc"))) + [os.path.join("..", "..", "include", "crackle_amd.h")]
ARCH = os.environ.get("CKL_OFFLOAD_ARCH", "gfx950")


def source_digest():
  """sha256 (first 16 hex digits) over the library's sources: names a build whatever machine compiled it
  (profiles/r03_pmc_traffic.json records it, bench.py takes roofline.traffic from that file only while
  the sources are the ones that were profiled)."""
  import hashlib
  h = hashlib.sha256()
  for name in sorted(SOURCES) + sorted(os.path.basename(x) for x in HEADERS):
    path = os.path.join(CSRC, name) if not name.endswith("crackle_amd.h") else os.path.join(HERE, "..", "include", "crackle_amd.h")
    with open(path, "rb") as f:
      h.update(name.encode() + b"\0" + f.read())
  return h.hexdigest()[:16]


def _hipcc():
  for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
    if c and (os.path.sep not in c or os.path.exists(c)):
      return c
  return "hipcc"


def _stale(target, deps):
  if not os.path.exists(target):
    return True
  t = os.path.getmtime(target)
  return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


TUNING_LIB = os.path.join(HERE, "libcrackle_amd_tuning.so")


def build(force=False, verbose=True, tuning=False):
  """tuning=True builds libcrackle_amd_tuning.so with -DCKL_TUNING (in-kernel cycle stamps and the
  ablation switches of tools/ablate.sh; selected at run time with CKL_TUNING_LIB=1).  The shipped
  library has neither."""
  srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
  hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
  objdir = os.path.join(HERE, "build_tuning" if tuning else "build")
  os.makedirs(objdir, exist_ok=True)
  objs = []
  flags = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]
  if tuning:
    flags.append("-DCKL_TUNING")
  LIB = TUNING_LIB if tuning else globals()["LIB"]
  for s in srcs:
    o = os.path.join(objdir, os.path.basename(s) + ".o")
    objs.append(o)
    if force or _stale(o, [s] + hdrs):
      cmd = [_hipcc(), *flags, "-c", s, "-o", o]
      if verbose:
        print(" ".join(cmd), flush=True)
      subprocess.run(cmd, check=True)
  if force or _stale(LIB, objs):
    cmd = [_hipcc(), "-shared", "-fPIC", "-pthread", f"--offload-arch={ARCH}", *objs, "-o", LIB]
    if verbose:
      print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
  if not tuning:
    build_fastcrackle(force=force, verbose=verbose)
  return LIB


def fastcrackle_path():
  import sysconfig
  return os.path.join(HERE, "fastcrackle" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_fastcrackle(force=False, verbose=True):
  """crackle_amd/fastcrackle*.so: the reference's pybind11 module (src/fastcrackle.cpp:641-669,
  same names and positional arguments) on top of the C-ABI; host code only (g++), linked against
  libcrackle_amd.so next to it ($ORIGIN rpath)."""
  import sysconfig
  import pybind11
  src = os.path.join(CSRC, "fastcrackle.cpp")
  out = fastcrackle_path()
  deps = [src, os.path.normpath(os.path.join(CSRC, "..", "..", "include", "crackle_amd.h")), LIB]
  if not (force or _stale(out, deps)):
    return out
  cmd = [
    os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden",
    f"-I{pybind11.get_include()}", f"-I{sysconfig.get_paths()['include']}",
    src, "-o", out, f"-L{HERE}", "-lcrackle_amd", "-Wl,-rpath,$ORIGIN",
  ]
  if verbose:
    print(" ".join(cmd), flush=True)
  subprocess.run(cmd, check=True)
  return out


if __name__ == "__main__":
  build(force="--force" in sys.argv, tuning="--tuning" in sys.argv)
