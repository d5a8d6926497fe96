"""Data-parallel training exchange steps (SURVEY.md §8(e) "Training") on CPU: world_size-2 gloo processes check that
the host logic of av_separation.parallel reproduces the single-process full-batch result -- bucketed gradient
averaging (several buckets, an unused parameter, two consecutive steps), the BatchNorm statistics merge, and the
batch-global PIT permutation choice.  The kernels themselves are covered by the -m gpu tests (test_train_gpu.py runs
the same two-rank job on the HIP path against the reference's gradients)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _net():
    torch.manual_seed(5)
    net = torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.Tanh(), torch.nn.Linear(32, 32), torch.nn.Tanh(),
                              torch.nn.Linear(32, 3))
    net.unused = torch.nn.Parameter(torch.ones(7))          # never reaches the loss: its bucket must still reduce
    return net


def _data():
    g = torch.Generator().manual_seed(9)
    return torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from av_separation import parallel
    from av_separation.losses import SeparationLoss
    res = {}
    # ---- 1. gradient buckets: 2 steps of SGD must track the single-process full-batch run
    net = _net()
    if rank == 1:
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)                                  # DataParallel must overwrite this with rank 0's values
    dp = parallel.DataParallel(net, bucket_mb=0.002, first_bucket_mb=0.0005, sync_bn=False)
    res["buckets"] = dp.buckets.bucket_sizes
    x, y = _data()
    idx = list(parallel.shard_range(rank, world, 8))
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    for _ in range(2):
        dp.zero_grad()
        loss = ((net(x[idx]) - y[idx]) ** 2).mean()
        loss.backward()
        dp.reduce_gradients()
        res.setdefault("grads", []).append([p.grad.clone() for p in net.parameters()])
        opt.step()
    res["params"] = [p.detach().clone() for p in net.parameters()]
    # ---- 2. BatchNorm statistics merge (ragged row counts)
    g = torch.Generator().manual_seed(3)
    rows = torch.randn(37, 5, generator=g) * 3 + 1
    mine = rows[:11] if rank == 0 else rows[11:]
    st = torch.cat([mine.mean(0), mine.var(0, unbiased=False), torch.tensor([float(mine.shape[0])])])
    allst = parallel.all_gather_rows(st)
    res["bn"] = parallel.combine_bn_stats(allst[:, :5], allst[:, 5:10], allst[:, 10])
    # ---- 3. PIT: rank-local choices disagree, the global choice must follow the whole batch
    tg = torch.rand(4, 2, 6, 5, generator=g)
    est = tg.clone()
    est[3] = tg[3].flip(0)                                    # clip 3 alone prefers the swapped order
    est = est + 0.05 * torch.rand(4, 2, 6, 5, generator=g)
    sl = slice(0, 2) if rank == 0 else slice(2, 4)
    crit = SeparationLoss(0.5)
    local = crit(est[sl], tg[sl], group=dist.group.WORLD)
    tot = local.detach().clone()
    dist.all_reduce(tot)
    res["pit"] = float(tot / world)
    res["pit_full"] = float(crit(est, tg))
    res["pit_local_only"] = float(crit(est[sl], tg[sl]))
    torch.save(res, f"{out}.{rank}")
    dist.destroy_process_group()


def test_two_rank_training_exchanges(tmp_path):
    out = str(tmp_path / "dp")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert len(r0["buckets"]) >= 3, r0["buckets"]                      # really bucketed
    # reference: one process, whole batch
    net = _net()
    x, y = _data()
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    for step in range(2):
        opt.zero_grad()
        ((net(x) - y) ** 2).mean().backward()
        for got0, got1, p in zip(r0["grads"][step], r1["grads"][step], net.parameters()):
            want = p.grad if p.grad is not None else torch.zeros_like(p)
            assert torch.allclose(got0, want, atol=1e-6) and torch.equal(got0, got1)
        opt.step()
    for a, b, p in zip(r0["params"], r1["params"], net.parameters()):
        assert torch.equal(a, b) and torch.allclose(a, p.detach(), atol=1e-6)
    # BatchNorm merge == statistics of all rows
    g = torch.Generator().manual_seed(3)
    rows = torch.randn(37, 5, generator=g) * 3 + 1
    mean, var, total = r0["bn"]
    assert float(total) == 37.0
    assert torch.allclose(mean, rows.mean(0), atol=1e-6) and torch.allclose(var, rows.var(0, unbiased=False), atol=1e-5)
    # PIT
    assert abs(r0["pit"] - r0["pit_full"]) < 1e-5 and abs(r1["pit"] - r0["pit_full"]) < 1e-5
    assert abs(0.5 * (r0["pit_local_only"] + r1["pit_local_only"]) - r0["pit_full"]) > 1e-3   # the case is not vacuous
