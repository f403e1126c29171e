"""Host side of the library under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers run on the CPU
build only): pack.cpp and the planning / validation code of the API files are rebuilt with -fsanitize=address,undefined and the
existing vstab_host_* CPU tests (forward schedule geometry, weight packer, XCD map, workspace layout, error paths) are run
against that build in a child interpreter.  Any sanitizer report fails the child (halt_on_error)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_tests_under_asan_ubsan():
    from coupe.optical_flow_based_deep_video_stabilization_amd import build
    lib = build.build_sanitized()
    env = dict(os.environ)
    env.update({"VSTAB_LIB": lib, "LD_PRELOAD": build.asan_runtime(),
                "ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1:abort_on_error=0:detect_odr_violation=0",
                "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"})
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_host_plan.py"), os.path.join(ROOT, "tests", "test_host_misc.py")],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, tail
    assert " passed" in r.stdout, tail
