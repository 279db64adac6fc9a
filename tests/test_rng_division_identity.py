"""FloatProducer.toDouble (Float.fs:29) divides by UInt32.MaxValue; the HIP path uses a division-free identity instead.
This runs the exhaustive C proof over all 2^32 inputs (about 10 s on one core)."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _cpu_has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read().replace("\n", " ")
    except OSError:
        return False


@pytest.mark.skipif(not _cpu_has_fma(), reason="needs a hardware fma (4.3e9 software fma calls would take minutes)")
def test_three_term_fma_equals_division_for_every_uint32(tmp_path):
    exe = str(tmp_path / "rng_division_identity")
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-o", exe, os.path.join(HERE, "c", "rng_division_identity.c"), "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout
    assert "three_term_mismatches=0 " in out.stdout
    assert "two_term_mismatches=0" not in out.stdout.split(" ", 1)[1]  # the shorter form really is inexact
