"""GPU test: the plain-C example (examples/receiver_flow.c) builds against include/gnsscorr.h with gcc -std=c99 and
runs the whole flow -- ring push, PCPS search of 8 PRNs, hand-over, closed-loop tracking -- through the C ABI alone."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(out):
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
        os.path.join(ROOT, "examples", "receiver_flow.c"), "-L", os.path.join(ROOT, "gnss-sdr-1_amd"), "-lgnsscorr",
        "-Wl,-rpath," + os.path.join(ROOT, "gnss-sdr-1_amd"), "-lm", "-o", out])


def test_c_example_compiles_as_c99(tmp_path):
    """CPU part: the example is valid C99 against the public header and links against the library."""
    _build(str(tmp_path / "receiver_flow"))


@pytest.mark.gpu
def test_c_example_runs(tmp_path):
    exe = str(tmp_path / "receiver_flow")
    _build(exe)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(p.stdout, p.stderr)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "receiver flow ok" in p.stdout and p.stdout.count("<- acquired") == 4 and p.stdout.count("locked") >= 4
    assert "NOT LOCKED" not in p.stdout
