"""CPU: the host side under AddressSanitizer + UndefinedBehaviorSanitizer
(SURVEY.md section 5 asked for this target; the reference has none, and its own
reader carries missing-return UB).

`make -C hgaprec_amd/csrc asan` builds libhgaprec_host_asan.so and
host_selftest_asan.  The native driver walks the id maps, the TSV reader on
hostile input, the dataset-cache parser on truncated / bit-flipped images, the
threaded writer and the TCP star of `hgaprec -ngpus N` (with a stranger
connecting first); then tests/test_host_side.py -- the reference-fixture and
oracle comparisons of the host library -- runs again with the sanitized
library loaded.  Any sanitizer report aborts the process and fails the test.
"""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "hgaprec_amd" / "csrc"
SAN_ENV = {
    "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1:halt_on_error=1",
    "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1",
}


@pytest.fixture(scope="module")
def asan_build():
    r = subprocess.run(["make", "-C", str(CSRC), "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return ROOT / "hgaprec_amd" / "libhgaprec_host_asan.so", ROOT / "hgaprec_amd" / "host_selftest_asan"


def test_native_selftest_under_asan_ubsan(asan_build, tmp_path):
    _, exe = asan_build
    env = dict(os.environ, **SAN_ENV)
    env["ASAN_OPTIONS"] = "detect_leaks=1:abort_on_error=1:halt_on_error=1"     # native code: leaks count too
    r = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    assert "host_selftest ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


def test_host_side_suite_under_asan_ubsan(asan_build):
    lib, _ = asan_build
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    libubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not (os.path.isabs(libasan) and os.path.exists(libasan)):
        pytest.skip("libasan.so not installed")
    env = dict(os.environ, **SAN_ENV)
    env["LD_PRELOAD"] = libasan + (":" + libubsan if os.path.exists(libubsan) else "")
    env["HGAPREC_HOST_LIB"] = str(lib)
    r = subprocess.run([sys.executable, "-m", "pytest", str(ROOT / "tests" / "test_host_side.py"), "-x", "-q",
                        "-p", "no:cacheprovider"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=1200)
    tail = r.stdout[-3000:] + r.stderr[-6000:]
    assert r.returncode == 0, tail
    assert "runtime error" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr, tail
    assert " passed" in r.stdout


def test_native_selftest_under_thread_sanitizer(tmp_path):
    """the threaded host paths of round 4 -- ratings files parsed in pieces, the CSR built per user range, the start
    state's expectations, three matrix writers side by side -- and the TCP star, under -fsanitize=thread: no data
    race (the first run of this found one: a counter read by the thread next to the one writing it)"""
    r = subprocess.run(["make", "-C", str(CSRC), "tsan"], capture_output=True, text=True)
    if r.returncode != 0 and "tsan" in (r.stdout + r.stderr).lower() and "cannot find" in (r.stdout + r.stderr):
        pytest.skip("libtsan not installed")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    exe = ROOT / "hgaprec_amd" / "host_selftest_tsan"
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1")
    r = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    assert "host_selftest ok" in r.stdout and "ThreadSanitizer" not in r.stderr
