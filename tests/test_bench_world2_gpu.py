"""GPU: `bench.py`'s world > 1 code path, executed before any multi-GPU node sees it (reference
semantics `traineval.py:130`).  Two processes launched exactly as the driver launches a scaling run
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 ... bench.py --gpus 2 ...`) share the one GPU of the test box
through `--backend gloo` (RCCL refuses two ranks on one device): the settled-precondition agreement (all_reduce MIN), the
barriers around the timed region, the MAX over ranks of the elapsed time, `all_gather_object` of the per-rank device reports and
the single JSON line of rank 0 must all come out."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_two_ranks_print_one_line_with_both_ranks():
    env = dict(os.environ, OBMAN_MANO_SYNTHETIC="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--backend", "gloo", "--batch", "8", "--image-size", "128", "--precondition-max", "9"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=REPO)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # rank 0 only
    assert p.stdout.strip().splitlines()[-1] == lines[0]  # and it is the LAST line of stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 16 and out["config"]["per_gpu_batch"] == 8 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and abs(out["value"] - 16 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]  # whole-job rate
    d = out["dist"]
    assert d["world_size"] == 2 and d["launcher_world_size"] == 2 and d["backend"] == "gloo"
    assert sorted(r["rank"] for r in d["ranks"]) == [0, 1] and len({r["pid"] for r in d["ranks"]}) == 2
    assert d["buckets"]["enabled"] and d["buckets"]["world_size"] == 2 and d["buckets"]["reduce_op"] == "SUM + 1/world"
    assert "selftest" in out and "cpu_baseline" not in out and out["precondition_steps"] >= 8
    assert out["config"]["final_loss"] == out["config"]["final_loss"]  # finite


def test_bench_graph_mode_refuses_a_host_driven_backend():
    """`bench.py --graph` records the data-parallel step only when its collectives are stream work (RCCL)."""
    env = dict(os.environ, OBMAN_MANO_SYNTHETIC="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--batch", "4", "--image-size", "64", "--precondition-max", "0", "--graph"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=REPO)
    assert p.returncode != 0 and "--graph" in (p.stdout + p.stderr) and "RCCL" in (p.stdout + p.stderr), (p.stdout[-1500:], p.stderr[-1500:])


def test_bench_one_rank_rccl_graph_mode_is_the_fused_data_parallel_graph():
    """The RCCL form on the one device there is: `--force-dist --graph` records the collectives INTO the step's graph."""
    env = dict(os.environ, OBMAN_MANO_SYNTHETIC="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "8",
           "--image-size", "128", "--precondition-max", "9", "--force-dist", "--graph", "--no-cpu-baseline", "--secondary-steps", "0"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=REPO)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["host_enqueue_ms"]["hipgraph_mode"] == "fused" and out["dist"]["backend"] == "nccl" and out["dist"]["buckets"]["enabled"]
    assert out["value"] > 0 and out["host_enqueue_ms"]["median"] < 2.0  # one launch call per step
