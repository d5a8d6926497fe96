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


def test_flops_per_clip_matches_survey_table():
    import bench
    f = bench.flops_per_clip
    assert abs(f(257, 63, 50, 32, 32, 256, 2, 2, 2) / 1e9 - 0.9209) < 5e-4       # cfg 1/2
    assert abs(f(257, 251, 50, 32, 32, 512, 6, 4, 2) / 1e9 - 20.370) < 5e-3      # cfg 3
    assert abs(f(257, 251, 75, 32, 32, 512, 6, 4, 3) / 1e9 - 21.609) < 5e-3      # cfg 4
    assert abs(f(257, 501, 50, 48, 48, 512, 6, 8, 2) / 1e9 - 56.033) < 5e-3      # cfg 5
