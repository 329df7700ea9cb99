"""shim/sgdnet_shim.c + tests/rmock/rmock.c under AddressSanitizer and UndefinedBehaviorSanitizer (CPU build:
GPU sanitizers are not available on the pool).  The shim is the code that runs inside R's process, and its
error paths -- a missing control field, an unknown family, shapes that do not match, an S4 object without
slots, a backend failure after the RNG scope was opened -- leave through Rf_error, i.e. a longjmp over the
shim's frames: exactly where a leak of PROTECTs, a use after free or a read past a coerced vector would
hide.  tests/test_shim_mock.py is run again, in a child process, against the sanitizer build of the same
two sources."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
ASAN_SO = os.path.join(HERE, "rmock", "libsgdnet_shim_mock_asan.so")


def test_shim_error_paths_are_clean_under_asan_and_ubsan():
    if not os.path.exists(ASAN_SO):
        pytest.fail(f"{ASAN_SO} missing: run ./build.sh")
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan next to gcc")
    env = dict(os.environ)
    env.update(LD_PRELOAD=libasan, SGDNET_SHIM_MOCK_SO=ASAN_SO,
               # the interpreter and the HIP runtime never free everything at exit; the longjmp of the mock's
               # Rf_error crosses instrumented frames
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(HERE, "test_shim_mock.py")], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=600)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-4000:]
    assert r.returncode == 0, out[-4000:]
    assert " passed" in r.stdout
