"""bench.py end to end on the GPU box: the N=1 contract line, and a 2-rank rehearsal of the N>1 launch (both ranks
on the single GPU of the test box, gloo instead of RCCL -- AVSEP_BENCH_REHEARSAL) so the sharding / barrier /
max-over-ranks / JSON path the driver runs at N=2,4,8 is exercised before it gets there."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline")


def _run(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # ONE JSON line
    return json.loads(lines[0])


def test_bench_single_gpu_contract_line():
    out = _run(["bench.py", "--steps", "20", "--warmup", "3", "--cpu-seconds", "2"])
    for k in CONTRACT + ("cpu_baseline",):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 20 and out["scaling"] == "weak" and out["dtype"] == "f32"
    assert out["config"]["workload"].startswith("cfg2") and out["config"]["global_batch"] == 32
    rf = out["roofline"]
    assert rf["bound"] == "mfma" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert abs(out["value"] - 32 / (out["ms_per_step"] * 1e-3)) < 1e-2 * out["value"]
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["gpu_masks_max_abs_err_vs_cpu"] < 1e-4


@pytest.mark.parametrize("mode,launcher", [("forward", "self"), ("forward", "torchrun"), ("train", "self")])
def test_bench_two_rank_rehearsal(mode, launcher):
    """`self`: plain `python bench.py --gpus 2` (the driver's observed form) -- bench.py spawns its own ranks;
    `torchrun`: launched under torch.distributed.run (the contract's documented form)."""
    args = ["bench.py", "--gpus", "2", "--steps", "4", "--warmup", "2"]
    if launcher == "torchrun":
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        args = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port)] + args
    if mode == "train":
        args += ["--mode", "train", "--batch", "2"]
    out = _run(args, {"AVSEP_BENCH_REHEARSAL": "1"})
    for k in CONTRACT:
        assert k in out, k
    assert out["n_gpus"] == 2 and "cpu_baseline" not in out
    per = 2 if mode == "train" else 32
    assert out["config"]["global_batch"] == 2 * per
    assert abs(out["value"] - 2 * per / (out["ms_per_step"] * 1e-3)) < 1e-2 * out["value"]
