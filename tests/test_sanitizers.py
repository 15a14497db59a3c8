"""Sanitizer tier (CPU only; GPU AddressSanitizer is not available on this pool).

  * the CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer, through every batch entry point;
  * the lane emulator of the HIP kernel source (64 host threads per wavefront, TSAT_SYNC / TSAT_SYNC_LDS = std::barrier)
    under ThreadSanitizer, every build of the solve kernel (wide, dense, packed with four / eight trajectories per wavefront —
    record ring, hand-over tables, joint backward passes — and the float builds): the independent check that every cross-lane LDS / HBM hand-off of the
    kernels sits behind one of the two sync macros;
  * both run the same workload (tests/sanitize/synth.hpp) and print checksums, which must agree between oracle and
    emulated kernel.
Standalone drivers rather than preloaded runtimes under Python: the sanitizer runtime must be the first DSO of the process.
"""
import os
import re
import subprocess

import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sanitize")


def _build(target):
    subprocess.check_call(["make", "-C", HERE, target], stdout=subprocess.DEVNULL)


def _run(exe, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([os.path.join(HERE, exe)], capture_output=True, text=True, timeout=timeout, env=e)
    return p.returncode, p.stdout, p.stderr


def _numbers(out):
    rows = {}
    for line in out.splitlines():
        key = " ".join(w for w in line.split() if re.fullmatch(r"[A-Za-z=0-9]*[A-Za-z][A-Za-z=0-9]*", w))
        rows[key] = [float(x) for x in re.findall(r"-?\d+\.\d+e[+-]\d+", line)] + [int(x) for x in re.findall(r"(?<![\w.+-])\d+(?![\w.])", line)]
    return rows


@pytest.fixture(scope="module")
def oracle_run():
    _build("oracle_asan")
    rc, out, err = _run("oracle_asan", {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})
    return rc, out, err


def test_oracle_is_clean_under_asan_and_ubsan(oracle_run):
    rc, out, err = oracle_run
    assert rc == 0, err[-2000:]
    assert "AddressSanitizer" not in err and "runtime error" not in err and "LeakSanitizer" not in err, err[-2000:]
    assert "mpc X" in out


@pytest.mark.parametrize("exe", ["emu_tsan", "emu_tsan_dense", "emu_tsan_packed", "emu_tsan_packed8", "emu_tsan_packed8w", "emu_tsan_packed_mixed", "emu_tsan_mixed"])
def test_emulated_kernels_are_race_free_under_tsan(exe, oracle_run):
    _build(exe)
    rc, out, err = _run(exe, {"TSAN_OPTIONS": "halt_on_error=0:report_signal_unsafe=0"})
    if "unexpected memory mapping" in err:       # ASLR layout TSan cannot shadow: environment, not a finding
        rc, out, err = subprocess.run(["setarch", "x86_64", "-R", os.path.join(HERE, exe)], capture_output=True, text=True,
                                      timeout=900).returncode, None, None
        pytest.skip("ThreadSanitizer cannot map its shadow in this environment") if rc else None
    assert "ThreadSanitizer" not in err, err[-3000:]
    assert rc == 0
    assert out.count("solve es=") == 4        # both state-difference modes x both integrators ran to the end
    if "packed" in exe:
        # the endgame of a packed launch (tsat_set_endgame): counters, the parked list and the states go from the first launch's
        # wavefronts to the second's through HBM — same checksums, no report, and something was parked
        rc2, out2, err2 = _run(exe, {"TSAN_OPTIONS": "halt_on_error=0:report_signal_unsafe=0", "TSAT_EMU_SUSPEND_AT": "3"})
        assert "ThreadSanitizer" not in err2, err2[-3000:]
        assert rc2 == 0 and out2 == out
        parked = [int(x) for x in re.findall(r"endgame parked (\d+)", err2)]
        assert len(parked) == 4 and max(parked) > 0, err2[-500:]
    if exe == "emu_tsan":                        # same workload as the oracle run: the checksums agree
        a, b = _numbers(oracle_run[1]), _numbers(out)
        assert a.keys() == b.keys() and len(a) >= 7
        for k in a:
            for x, y in zip(a[k], b[k]):
                assert abs(x - y) <= 1e-9 * max(1.0, abs(x)), (k, x, y)
