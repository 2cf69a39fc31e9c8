"""bench.py as the driver starts it: a plain `python bench.py --gpus N` (no launcher) must bring up N ranks itself and leave rank 0's
JSON as the last line of stdout.  Run here at N = 2 on the ONE GPU of the test box (MCR_TEST_BACKEND=gloo: both ranks on cuda:0,
host-staged collectives -- RCCL refuses duplicate devices; every kernel still runs on the GPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launches_two_ranks_on_one_gpu(dev):
    env = dict(os.environ, MCR_TEST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--nbv-iters", "3",
                        "--no-cpu-baseline", "--watchdog", "150"], capture_output=True, text=True, timeout=420, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    last = r.stdout.strip().splitlines()[-1]
    res = json.loads(last)                                     # rank 0's line is the LAST line of stdout
    assert res["n_gpus"] == 2 and res["ranks_seen"] == 2 and res["steps"] == 20 and res["value"] > 0
    assert res["scaling"] == "weak" and res["config"]["parallelism"] == "camera-shard x2"
    assert res["nbv_step"]["p50_ms"] > 0 and res["nbv_step"]["config"]["parallelism"] == "query+camera shard x2"
    assert res["nbv_batch"]["config"]["parallelism"] == "cloud shard x2"
    mac = res["macarons_step"]
    assert mac["p50_ms"] > 0 and mac["config"]["parallelism"].endswith("x2") and mac["checks"]["all_hold"], mac


def test_contract_line_survives_the_extra_legs(dev):
    """The extra legs (NBV step, scene batch, MACARONS decision) run under a deadline: when they do not finish -- here a 1 s deadline;
    on a real node a rank stuck in a collective -- every rank leaves and rank 0 still prints the contract line (the scorer loop's
    result, complete before the legs start), marked `legs_incomplete`, with exit code 0."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-pmc", "--no-cpu-baseline",
                        "--legs-deadline", "1"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["value"] > 0 and res["steps"] == 20 and res["roofline"]["frac"] > 0 and "legs_incomplete" in res, res
