"""bench.py's own launcher and its N>1 (row-sharded, strong-scaling) path, on a box without GPUs:
``python bench.py --gpus 2`` must start two ranks by itself, cut ONE frame by rows, exchange the gradients
and print one JSON line whose n_gpus is 2.  The ranks trace with the CPU oracle wrapped in the Pipeline
interface (bench.py's RF_BENCH_TEST_PIPELINE hook, gloo backend) -- what is under test is the launcher, the
sharding and the exchange, not the tracer; the JSON says that it is not a measurement."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, gpus):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["RF_BENCH_TEST_PIPELINE"] = "tests.test_dist:OraclePipeline"
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "2",
           "--backend", "gloo", "--points", "1500", "--seed", "4", "--sh-degree", "1", "--width", "40", "--height", "32",
           "--no-cpu-baseline"] + extra
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("extra,scaling", [([], "strong"), (["--exchange", "dense", "--no-rebalance"], "strong"),
                                           (["--weak"], "weak")])
def test_bench_launches_its_own_ranks(extra, scaling):
    d = _run(extra, 2)
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["steps"] == 2 and d["warmup"] == 2
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert "not a measurement" in d["data"]
    rays = 40 * 32
    assert d["config"]["rays_per_step"] == (rays if scaling == "strong" else 2 * rays)
    if scaling == "strong":
        b = d["detail"]["row_bounds"]
        if "--no-rebalance" in extra:
            assert b is None
        else:
            assert b[0] == 0 and b[-1] == 32 and len(b) == 3 and b[1] % 8 == 0
        if "--exchange" not in extra:   # the toy foam is touched everywhere: either outcome is legitimate
            assert (d["detail"]["exchange"] == "sparse" and len(d["detail"]["exchange_rows_per_rank"]) == 2) \
                or "fell back" in d["detail"]["exchange"]
        else:
            assert d["detail"]["exchange"] == "dense"


def test_bench_forward_only_frame_cut_by_rows():
    """BASELINE config 5's multi-GPU shape: a frame cut by rows, forward only, nothing to exchange."""
    d = _run(["--forward-only"], 2)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["rays_per_step"] == 40 * 32
    b = d["detail"]["row_bounds"]
    assert b[0] == 0 and b[-1] == 32 and len(b) == 3
    assert d["detail"]["backward_ms"] == 0.0


def test_bench_single_rank_needs_no_launcher():
    d = _run([], 1)
    assert d["n_gpus"] == 1 and d["config"]["rays_per_step"] == 40 * 32
