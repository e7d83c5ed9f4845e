"""-m gpu: the RCCL code path (torch.distributed backend "nccl") exercised on the GPU box.
The box has ONE GPU, so these run world size 1 -- the point is that communicator creation, device-tensor collectives,
sot_amd.distributed on HIP tensors and bench.py's N > 1 branch have all run under test before the 8-GPU driver run.
Each job is a fresh child process tree (`python -m torch.distributed.run`): the permitted way to start another GPU program."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _torchrun(script_and_args, extra_env=None, timeout=420):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), *script_and_args]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.gpu
def test_sharded_sot_loss_over_rccl_equals_the_single_process_module():
    r = _torchrun([os.path.join(ROOT, "tests", "rccl_child.py")])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RCCL_CHILD ")]
    assert line, r.stdout[-2000:]
    out = json.loads(line[-1][len("RCCL_CHILD "):])
    assert out["backend"] == "nccl" and out["world"] == 1 and out["allreduce"] == 1.0
    for tag in ("mean", "dims1"):
        got, want = out[tag]["got"], out[tag]["want"]
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert abs(a - b) <= 2e-6 * abs(b), (tag, a, b)
        assert out[tag]["grad_err"] <= 2e-6 * out[tag]["grad_max"], (tag, out[tag])


@pytest.mark.gpu
def test_bench_rank_process_runs_the_collective_branch():
    """bench.py as the driver launches it for N > 1 (a rank of torch.distributed.run), with SOT_BENCH_FORCE_DIST=1 so that ONE
    rank takes the RCCL branch: two alternating streams, one all-reduce(SUM) of the fp64 partial sum per step."""
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "40", "--warmup", "5", "--prewarm", "40", "--rows", "2048",
                   "--no-extras", "--no-cpu-baseline"], {"SOT_BENCH_FORCE_DIST": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # exactly one JSON line on stdout (RCCL's banner goes to stderr)
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["steps"] == 40 and rec["scaling"] == "weak" and rec["unit"] == "rows/s"
    assert "all-reduce" in rec["config"]["collective"] and rec["config"]["streams"] == 2
    assert rec["value"] > 0 and rec["ms_per_step"] > 0
    assert rec["extras"]["ms_per_step_without_collective"] > 0
    assert 0.0 < rec["roofline"]["frac"] < 1.0
