"""Tracking kernel time against the epoch length at a fixed number of bytes (how much of a workgroup's life is prologue):
python profiles/tools/trk_epoch_length.py <samples per epoch> <epochs per channel>"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
n, e = int(sys.argv[1]), int(sys.argv[2])
bench.N_EPOCH = n
sys.argv = ['bench.py', '--no-cpu', '--no-acq', '--no-shared', '--epochs', str(e), '--steps', '20']
try:
    bench.main()
except AssertionError as ex:
    print("assert", str(ex)[:100])
