"""bench.py --gpus N without a GPU: the control plane of a multi-rank run on gloo ranks.

At N > 1 bench.py runs every transport (rccl, p2p-gather, p2p-consume) as its own leg in a child
process per rank and ranks the legs that worked.  Here no HIP device exists, so every leg's children
fail at once ("no HIP device visible" -- the product has no CPU fallback); what this checks is that
the parents neither hang nor crash: rank 0 still prints ONE JSON line naming every leg with its exit
codes, `value` is null, and the run exits non-zero.  (The legs themselves are exercised on the GPU
box: tests/test_gpu_p2p.py::test_bench_legs_on_one_gpu.)
"""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_multi_rank_bench_reports_every_leg_when_no_device_is_there(B):
    if B.device_count() >= 1:
        import pytest
        pytest.skip("a HIP device is visible: the legs would run (covered by the gpu-marked test)")
    env = dict(os.environ, L2Z_BENCH_LEG_TIMEOUT_S="120")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode != 0
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:] + p.stderr.decode()[-2000:]
    out = json.loads(lines[0])
    assert out["value"] is None and out["n_gpus"] == 2
    legs = out["comm"]["legs"]
    assert [l["transport"] for l in legs] == ["rccl", "p2p-gather", "p2p-consume", "rccl-allreduce", "p2p-allreduce"]
    for l in legs:
        assert l["ok"] is False and l["exit_codes"] == [1, 1], l
    assert out["comm"]["rccl"]["initialised"] is False
