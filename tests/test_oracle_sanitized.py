"""the CPU oracles under AddressSanitizer + UndefinedBehaviorSanitizer.  `make -C oracle asan` builds both
oracles instrumented; the oracle-only tests of the CPU suite (known answers, golden vectors, agreement of the two oracles, the
KLD and FastSLAM 2.0 restatements, the reference's invariants) then run in a child interpreter that preloads the ASan runtime
and loads the instrumented libraries.  Any heap overflow, use after free or undefined operation aborts the child."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_TESTS = ["tests/test_known_answers.py", "tests/test_golden.py", "tests/test_oracle_agreement.py", "tests/test_kld_oracles.py",
                "tests/test_fs2_oracles.py", "tests/test_reference_invariants.py", "tests/test_detmath.py"]


def test_oracle_suite_is_clean_under_asan_and_ubsan():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    assert os.path.exists(libasan), "no libasan.so beside gcc"
    env = dict(os.environ, RR_ORACLE_DIR=os.path.join(ROOT, "oracle", "_build_asan"), LD_PRELOAD=libasan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-s", "-m", "not gpu", "-p", "no:cacheprovider"] + ORACLE_TESTS,  # -s: a sanitizer report must reach stderr
                       capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
