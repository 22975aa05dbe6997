"""CPU test of the stream ring's reader bookkeeping (gnss-sdr-1_amd/csrc/gc_reader_table.h): producer and consumer threads with
host-side events; a launch must never read a ring cell that a push has overwritten since the launch was validated.  Also run
under ThreadSanitizer when the toolchain has it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "reader_table_selftest.cpp")
INC = os.path.join(ROOT, "gnss-sdr-1_amd", "csrc")


def _run(tmp_path, flags, name):
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-g", *flags, "-I", INC, SRC, "-o", exe, "-lpthread"])
    return subprocess.run([exe], capture_output=True, text=True, timeout=600)


def test_reservation_protocol_threads(tmp_path):
    p = _run(tmp_path, [], "rt")
    assert p.returncode == 0, p.stdout + p.stderr
    assert " 0 stale" in p.stdout, p.stdout


def test_reservation_protocol_under_tsan(tmp_path):
    probe = subprocess.run(["g++", "-fsanitize=thread", "-x", "c++", "-", "-o", str(tmp_path / "probe")], input="int main(){return 0;}", text=True, capture_output=True)
    if probe.returncode != 0:
        pytest.skip("ThreadSanitizer runtime not installed")
    p = _run(tmp_path, ["-fsanitize=thread"], "rt_tsan")
    assert p.returncode == 0 and "ThreadSanitizer" not in p.stderr, p.stdout + p.stderr[-4000:]
