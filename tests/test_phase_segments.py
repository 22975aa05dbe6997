"""CPU test: the closed-form float32 running phase of the Doppler wipe-off tables (gnss-sdr-1_amd/csrc/acq_phase_segments.h) equals the
sequential `_phase += phase_inc` of volk_gnsssdr_s32f_sincos_32fc (pcps_acquisition.cc:296-310) bit for bit -- for every increment of
the acquisition grids the tests and the bench use (four sampling rates, FDMA / IF offsets, +-10 kHz in 250 Hz) and 3000 random ones
incl. tie-prone mantissas (tests/phase_segments_selftest.cpp).  The table built from it on the GPU is compared with the oracle's
reference-pinned table in tests/test_acquisition_gpu.py (wipe-off rows through gc_acq_peek)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_phase_segments_equal_the_sequential_float32_sum(tmp_path):
    exe = str(tmp_path / "phase_segments_selftest")
    subprocess.check_call(["g++", "-O2", "-std=c++14", "-ffp-contract=off", "-Wall", "-I", os.path.join(ROOT, "gnss-sdr-1_amd", "csrc"),
        os.path.join(ROOT, "tests", "phase_segments_selftest.cpp"), "-o", exe])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(p.stdout, p.stderr)
    assert p.returncode == 0 and "agree with the sequential float32 sum bit for bit" in p.stdout
