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


@pytest.mark.gpu
def test_python_receiver_front_end_example():
    """examples/receiver_bench.py: ring push -> 32-PRN PCPS search -> hand-over into free slots of one closed-loop engine ->
    tracking, at 4 Msps for half a second: every satellite of the stream found (no false alarm) and tracked to within 5 Hz."""
    import json
    import sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "receiver_bench.py"), "--fs", "4000000", "--seconds", "0.5"], capture_output=True,
        text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["all_found_and_tracked"] and d["detected"] == d["satellites_in_stream"] and len(d["detected"]) == 8
