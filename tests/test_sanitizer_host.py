"""AddressSanitizer + UBSan over the HOST side of the HIP extension (SURVEY section 5 "sanitizers"; GPU ASan / xnack+ is not
available on this pool).  tools/build_asan.sh builds swat_amd/lib/libswat_hip_asan.so (about six minutes, so it is not part of
build()); when it exists this test drives it through argument checks, dispatch / tile planning and launch-failure paths without
a GPU and fails on any sanitizer report.  The C oracle has its own sanitizer test (tests/test_oracle_c.py)."""
import glob
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "..", "swat_amd", "lib", "libswat_hip_asan.so")


def test_host_side_under_asan_ubsan():
    if not os.path.exists(LIB):
        pytest.skip("libswat_hip_asan.so not built (tools/build_asan.sh)")
    srcs = glob.glob(os.path.join(HERE, "..", "swat_amd", "csrc", "*.h*")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    if max(os.path.getmtime(f) for f in srcs) > os.path.getmtime(LIB):
        pytest.skip("libswat_hip_asan.so is older than the sources: rebuild it with tools/build_asan.sh")
    rt = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not rt:
        pytest.skip("no libclang_rt.asan in this toolchain")
    env = dict(os.environ, LD_PRELOAD=rt[0], ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:verify_asan_link_order=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, os.path.join(HERE, "helpers", "asan_host_driver.py"), os.path.abspath(LIB)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "no sanitizer report" in r.stdout
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
