"""The chip-domain form of the plain multicorrelator loop (gnss-sdr-1_amd/csrc/trk_chips.hpp, GNSSCORR_TRK_LOOP=chips).

It is not the default -- it measured slower than the per-sample loop in every mode of the bench -- but it is
a complete second implementation of the hot loop with the reference's exact chip walk (volk_gnsssdr_32f_xn_resampler_32f_xn.h:77-94),
so it is held to the same parity suite: the switch is read once per process, hence the child interpreter."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_parity_suite_with_the_chip_domain_loop():
    env = dict(os.environ, GNSSCORR_TRK_LOOP="chips")
    files = ["tests/test_tracking_gpu.py", "tests/test_tracking_variants_gpu.py", "tests/test_fuzz_gpu.py::test_randomised_level1_call_sequences", "tests/test_fuzz_gpu.py::test_randomised_ring_addressing",
        "tests/test_fuzz_gpu.py::test_randomised_open_loop_parity"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + files, cwd=ROOT, env=env,
        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout
