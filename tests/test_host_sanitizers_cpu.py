"""AddressSanitizer + UndefinedBehaviorSanitizer run of the host-only translation unit
(csrc/vrx_host.cpp: MatrixMarket parser, count merge, threaded text / VCF writers, the MT19937
continuation and jump, NumPy's float32 sum) -- SURVEY.md section 5 / VERDICT r3 housekeeping.

The unit is compiled on its own with g++ -fsanitize=address,undefined (no GPU, no hipcc) and
driven in a child process (tests/_san_driver.py) by the CPU tests of the I/O layer and by
malformed MatrixMarket inputs; any sanitizer report fails the test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vireo_amd", "csrc")
GXX = os.environ.get("CXX", "g++")


def test_host_translation_unit_under_asan_ubsan(tmp_path):
    try:
        libasan = subprocess.run([GXX, "-print-file-name=libasan.so"], capture_output=True, text=True,
                                 check=True).stdout.strip()
    except (OSError, subprocess.CalledProcessError):
        pytest.skip("no g++")
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("g++ has no libasan")
    lib = str(tmp_path / "libvrx_host_san.so")
    subprocess.run([GXX, "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer",
                    "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"), "vrx_host.cpp",
                    "-o", lib, "-lz", "-lpthread"], cwd=CSRC, check=True)
    # (libstdc++ next to libasan: the interceptor of __cxa_throw must find the real one when a
    #  C++ extension of SciPy throws, else ASan aborts on its own CHECK)
    libstdcxx = subprocess.run([GXX, "-print-file-name=libstdc++.so.6"], capture_output=True, text=True,
                               check=True).stdout.strip()
    preload = libasan + (":" + libstdcxx if os.path.isabs(libstdcxx) else "")
    env = dict(os.environ, LD_PRELOAD=preload, PYTHONPATH=ROOT,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=97",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98")
    scratch = tmp_path / "scratch"
    scratch.mkdir()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_san_driver.py"), lib, str(scratch)],
                       env=env, capture_output=True, text=True, timeout=900)
    report = p.stdout[-3000:] + "\n" + p.stderr[-6000:]
    assert "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, report
    assert p.returncode == 0, report
    assert "mtx fuzz:" in p.stdout and "balance tiles: 6 cases" in p.stdout
