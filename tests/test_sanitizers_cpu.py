"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU-side code: the oracle's C restatement
and the host-only stages of the product library (ckl_common / ckl_zstack / ckl_pins: no kernels in
them), driven by tests/sanitize/host_sanitize.cpp over valid, ragged and hostile streams.  GPU
sanitizers are not available on this pool; the device code is covered by the parity tests."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "crackle_amd", "csrc")


@pytest.mark.skipif(shutil.which("g++") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime.h"), reason="needs g++ and the HIP headers")
def test_host_stages_and_oracle_under_asan_ubsan(tmp_path):
  exe = str(tmp_path / "host_sanitize")
  san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
  cobj = str(tmp_path / "oracle.o")
  subprocess.run(["gcc", "-std=c11", "-msse4.2", "-c", os.path.join(ROOT, "oracle", "ckl_oracle.c"), "-o", cobj] + san, check=True)
  cmd = ["g++", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-msse4.2", "-mpclmul"] + san
  for src in ("ckl_common.hip", "ckl_zstack.hip", "ckl_pins.hip"):
    cmd += ["-x", "c++", os.path.join(CSRC, src)]
  cmd += ["-x", "c++", os.path.join(ROOT, "tests", "sanitize", "host_sanitize.cpp"), "-x", "none", cobj,
          "-o", exe, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-lm"]
  subprocess.run(cmd, check=True)
  env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
  p = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=600)
  assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-6000:]
  assert "host_sanitize: 0 failures" in p.stdout
