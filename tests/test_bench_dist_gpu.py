"""The driver's multi-GPU launch shape on the one GPU a test box has: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 1
--master-addr 127.0.0.1 ... bench.py --gpus 1` with the process group forced on (BENCH_FORCE_DIST=1), so that bench.py's RCCL branch
(`init_process_group("nccl", device_id=...)`, the barriers, the MAX all-reduce of the elapsed time on the device) executes for real.
The N > 1 sharding logic itself is covered by the 2-rank gloo test (tests/test_sharding.py); 8 GPUs are the driver's to launch."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_under_torchrun_with_rccl_process_group():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
        os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--epochs", "16", "--no-cpu", "--no-acq", "--no-shared"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["scaling"] == "weak" and j["value"] > 0
    assert j["roofline"]["traffic"] is None  # another workload than the profiled one: no borrowed counter figure


def test_two_ranks_report_per_gpu_figures():
    """Two ranks (gloo, both on the one GPU of the box: BENCH_FORCE_DEVICE) through bench.py's N > 1 path: the line carries one
    per_gpu entry per rank -- own Msamples/s, own kernel time and HBM fraction, own hybrid share (BASELINE configs[4]) and own PRNs of
    the acquisition search (configs[3]) -- and names the slowest rank."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, BENCH_FORCE_DEVICE="0", BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--epochs", "16", "--segments", "4", "--no-cpu"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    j = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and [g["rank"] for g in j["per_gpu"]] == [0, 1] and j["slowest_rank"] in (0, 1)
    for g in j["per_gpu"]:
        assert g["msamples_s"] > 0 and 0 < g["hbm_frac"] < 1 and g["kernel_ms"] > 0 and g["steady_state_kernel_ms"] > 0
    # whole-job value = all ranks' units / the slowest rank's time: never above the sum of the per-GPU rates
    assert j["value"] <= sum(g["msamples_s"] for g in j["per_gpu"]) * (1 + 1e-9)
    assert len(j["roofline"]["segments_ms"]) == 4
    # every rank timed ITS share of BASELINE configs[4] (16 GPS + 8 Galileo 5-tap + 8 BeiDou, open loop) and its PRNs of the cfg4 search
    # (channel / PRN i on GPU i mod G; gnss_flowgraph.cc:496-499)
    for g in j["per_gpu"]:
        assert g["hybrid_msamples_s"] > 0 and 0 < g["hybrid_frac"] < 1.2 and g["acq_dwells_per_s"] > 0 and 0 < g["acq_hbm_frac"] < 1
    assert j["per_gpu"][0]["acq_prns"] == list(range(1, 33, 2)) and j["per_gpu"][1]["acq_prns"] == list(range(2, 33, 2))
    assert j["acquisition_all_gpus"]["prns_partition_1_to_32"] is True
    assert j["acquisition_all_gpus"]["dwells_per_s"] == pytest.approx(sum(g["acq_dwells_per_s"] for g in j["per_gpu"]))
    assert j["hybrid_all_gpus"]["channels"] == 64 and j["hybrid_all_gpus"]["value"] == pytest.approx(sum(g["hybrid_msamples_s"] for g in j["per_gpu"]))
