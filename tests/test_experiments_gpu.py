"""The EXPERIMENTS build (make -C gnss-sdr-1_amd/csrc exp -> libgnsscorr_exp.so, -DGNSSCORR_EXPERIMENTS): kernel variants that were
built, held to the parity suite and measured slower than the product's, kept out of libgnsscorr.so:
  * the chip-domain form of the plain multicorrelator loop (csrc/trk_chips.hpp, GNSSCORR_TRK_LOOP=chips);
  * the inverse transform of a cell kept on its CU (acq_inv_fused_kernel, GNSSCORR_ACQ_ONCHIP=1);
  * row and column passes as roles of one launch (GNSSCORR_ACQ_ROLES=1);
  * closed-loop code periods cut into slices, one launch per period (gc_trk_loop_set_geometry(slices > 1); tests/exp_sliced_loop.py).
Opt-in: GNSSCORR_TEST_EXPERIMENTS=1 and the experiments library present; the default `-m gpu` run skips them (they re-run whole
suites on slower kernels).  The switches are read once per process, hence the child interpreters."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP_LIB = os.path.join(ROOT, "gnss-sdr-1_amd", "libgnsscorr_exp.so")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("GNSSCORR_TEST_EXPERIMENTS") != "1" or not os.path.exists(EXP_LIB),
    reason="opt-in: GNSSCORR_TEST_EXPERIMENTS=1 and `make -C gnss-sdr-1_amd/csrc exp`")]


def _suite(env_extra, files):
    env = dict(os.environ, GNSSCORR_LIB=EXP_LIB, **env_extra)
    env.pop("GNSSCORR_TEST_EXPERIMENTS", None)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + files, cwd=ROOT, env=env,
        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout


def test_parity_suite_with_the_chip_domain_loop():
    """A complete second implementation of the hot loop with the reference's exact chip walk
    (volk_gnsssdr_32f_xn_resampler_32f_xn.h:77-94), held to the same parity suite."""
    _suite({"GNSSCORR_TRK_LOOP": "chips"}, ["tests/test_tracking_gpu.py", "tests/test_tracking_variants_gpu.py",
        "tests/test_fuzz_gpu.py::test_randomised_level1_call_sequences", "tests/test_fuzz_gpu.py::test_randomised_ring_addressing",
        "tests/test_fuzz_gpu.py::test_randomised_open_loop_parity"])


def test_acquisition_suite_with_the_on_chip_inverse_transform():
    """pcps_acquisition.cc:724-739 with a cell's values kept on its CU: same oracle gates, pair path bit-identical to per-dwell."""
    _suite({"GNSSCORR_ACQ_ONCHIP": "1"}, ["tests/test_acquisition_gpu.py"])


def test_acquisition_suite_with_row_and_column_roles():
    _suite({"GNSSCORR_ACQ_ROLES": "1", "GNSSCORR_ACQ_Q_MB": "8"}, ["tests/test_acquisition_gpu.py::test_cfg4_full_width_32_prns_41_bins_2_dwells",
        "tests/test_acquisition_gpu.py::test_dwell_pairs_equal_per_dwell_processing"])


def test_sliced_closed_loop_periods():
    _suite({}, ["tests/exp_sliced_loop.py"])
