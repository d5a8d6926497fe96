"""N>1 path of bench.py on CPU: world_size-2 gloo processes check the clip sharding (disjoint, covering,
no data-path collective needed) and the barrier + MAX-over-ranks timing reduction bench.py uses."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, batch, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    mine = torch.tensor(list(bench.shard_range(rank, world, batch)), dtype=torch.int64)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)              # test-only: the bench itself moves no clip data between ranks
    dist.barrier()
    t = torch.tensor([0.010 * (rank + 1)], dtype=torch.float64)     # pretend rank r took 10*(r+1) ms
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        allc = torch.cat(gathered)
        torch.save({"clips": allc, "tmax": float(t.item())}, out)
    dist.destroy_process_group()


def test_two_rank_sharding_and_max_time(tmp_path):
    out = str(tmp_path / "r0.pt")
    world, batch = 2, 32
    mp.spawn(_worker, args=(world, _free_port(), batch, out), nprocs=world, join=True)
    r = torch.load(out)
    assert sorted(r["clips"].tolist()) == list(range(world * batch))       # disjoint and covering
    assert abs(r["tmax"] - 0.020) < 1e-12                                    # slowest rank defines the step time
    # whole-job value = all ranks' clips / max time, as bench.py computes it
    assert abs(world * batch * 1 / r["tmax"] - 3200.0) < 1e-6


def _bench(args, env_extra=None, env_full=None):
    """env_full: the child's whole environment (the way to REMOVE variables); env_extra: additions to this one's."""
    import json
    import subprocess
    env = dict(os.environ) if env_full is None else dict(env_full)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r.returncode, [json.loads(ln) for ln in lines], r.stderr


def test_plain_bench_gpus2_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun around it (how the driver calls it): the parent spawns one worker
    per rank before any GPU call and relays rank 0's single JSON line; --selftest-launch stops short of the GPU work."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    rc, lines, err = _bench(["--gpus", "2", "--selftest-launch"], env_full=env)
    assert rc == 0, err[-2000:]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2
    assert abs(lines[0]["elapsed_max"] - 0.020) < 1e-12 and lines[0]["rank0_clips"] == [0, 1, 2, 3]
    # per-rank reporting of the N > 1 line: every rank's own time and device identity, gathered on rank 0
    assert lines[0]["per_rank_ms"] == [10.0, 20.0] and lines[0]["world_size"] == 2
    assert [d["pci"] for d in lines[0]["devices"]] == ["0000:01:00", "0000:02:00"]
    assert lines[0]["devices"][0]["pid"] != lines[0]["devices"][1]["pid"]


def test_two_ranks_on_one_device_is_an_error():
    rc, lines, err = _bench(["--gpus", "2", "--selftest-launch"], {"AVSEP_SELFTEST_SAME_DEVICE": "1"})
    assert rc != 0 and not lines and "share a device" in err


def test_duplicate_device_detection():
    import bench
    a = {"uuid": "GPU-1", "pci": "0000:05:00"}
    b = {"uuid": "GPU-2", "pci": "0000:06:00"}
    assert bench.duplicate_devices([a, b]) == []
    assert bench.duplicate_devices([a, b, dict(a)]) == [(0, 2, "0000:05:00")]
    assert bench.duplicate_devices([{"uuid": None, "pci": "0000:05:00"}, {"uuid": None, "pci": "0000:05:00"}]) == [(0, 1, "0000:05:00")]
    # nothing is concluded from a missing or all-zero address, nor from equal uuids alone
    assert bench.duplicate_devices([{"uuid": None, "pci": None}, {"uuid": None, "pci": None}]) == []
    assert bench.duplicate_devices([{"uuid": "X", "pci": "0000:00:00"}, {"uuid": "X", "pci": "0000:00:00"}]) == []
    assert bench.duplicate_devices([{"uuid": "X", "pci": "0000:05:00"}, {"uuid": "X", "pci": "0000:06:00"}]) == []
    assert bench.median([3.0, 1.0, 2.0]) == 2.0 and bench.median([4.0, 1.0, 2.0, 3.0]) == 2.5


def test_plain_bench_propagates_a_worker_failure():
    rc, lines, _ = _bench(["--gpus", "2", "--selftest-launch"], {"AVSEP_SELFTEST_FAIL_RANK": "1"})
    assert rc != 0 and not lines


def test_world_size_mismatch_is_an_error():
    rc, lines, err = _bench(["--gpus", "2", "--selftest-launch"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and "disagree" in err


def test_flops_per_clip_matches_survey_table():
    import bench
    f = bench.flops_per_clip
    assert abs(f(257, 63, 50, 32, 32, 256, 2, 2, 2) / 1e9 - 0.9209) < 5e-4       # cfg 1/2
    assert abs(f(257, 251, 50, 32, 32, 512, 6, 4, 2) / 1e9 - 20.370) < 5e-3      # cfg 3
    assert abs(f(257, 251, 75, 32, 32, 512, 6, 4, 3) / 1e9 - 21.609) < 5e-3      # cfg 4
    assert abs(f(257, 501, 50, 48, 48, 512, 6, 8, 2) / 1e9 - 56.033) < 5e-3      # cfg 5
