"""Data-parallel training around the HIP training path: one process per MI355X, torch.distributed over RCCL
(backend "nccl" on ROCm), SURVEY.md §8(e) "Training" row.  The reference has no distributed code (single
process, demo.py:121), so this is new functionality whose contract is: G ranks x B/G clips produce the SAME
parameter update as the reference's single process on the B-clip batch.  That needs three exchanges per step:

  1. gradients     : sum/G of every parameter gradient as reduce-scatter + all-gather over flat buckets, the
                     reduce-scatters launched from autograd hooks while the rest of the backward still runs
                     (``GradBuckets``);
  2. BatchNorm     : per-channel statistics over the rows of all ranks, forward (mean/var/count all-gather) and
                     backward (two sums all-reduce) -- ``_train.SyncBatchNormReluFn``, 3 layers x 224 channels total;
  3. PIT loss      : the permutation is chosen for the WHOLE batch from the batch-mean loss (losses.py:65-71), so the
                     S! candidate losses are all-reduced before the argmin (``losses.SeparationLoss(group=...)``).

``clip_grad_norm_`` (demo.py:103) needs no exchange of its own: after (1) every rank holds the same gradients.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): small all-reduces are latency-bound and a ring only ever
drives one link per hop, so buckets are LARGE (default 64 MB; cfg 4/5 have 212/262 MB of fp32 gradients -> 4-5
collectives per step) with a smaller first bucket so the first collective starts early in the backward.

gloo (the CPU tests, and >1 rank sharing one GPU) has no device collectives here, so device tensors are staged
through the host for that backend only.
"""
from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn as nn


# ------------------------------------------------------------------------------------------- small collectives
def _staged(t: torch.Tensor, group) -> bool:
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce_sum_(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum over the ranks of ``group``."""
    if _staged(t, group):
        h = t.detach().cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, group=group)
    return t


def all_gather_rows(t: torch.Tensor, group=None) -> torch.Tensor:
    """[n] on every rank -> [G, n] (row r = rank r's vector)."""
    world = dist.get_world_size(group)
    src = t.detach().cpu() if _staged(t, group) else t.detach().contiguous()
    out = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(out, src, group=group)
    return torch.stack(out).to(t.device)


def combine_bn_stats(means: torch.Tensor, variances: torch.Tensor, counts: torch.Tensor):
    """Merge per-rank (mean, biased variance, row count) into the statistics of the union of all rows:
    mean = sum n_r mean_r / N;  var = sum n_r (var_r + (mean_r - mean)^2) / N.  means/variances [G, C], counts [G]."""
    n = counts.reshape(-1, 1)
    total = counts.sum()
    mean = (means * n).sum(0) / total
    var = ((variances + (means - mean) ** 2) * n).sum(0) / total
    return mean, var, total


# ------------------------------------------------------------------------------------------- gradient buckets
class GradBuckets:
    """Flat gradient buckets, exchanged as reduce-scatter + all-gather, overlapped with the backward pass.

    Parameters are taken in REVERSE registration order (roughly the order the backward produces their gradients);
    each parameter's ``.grad`` is a view into its bucket's flat buffer, so there is no gather/scatter copy: autograd
    accumulates in place, the post-accumulate hook counts arrivals and the bucket's REDUCE-SCATTER is launched
    (``async_op``) the moment its last gradient lands.  ``finish()`` launches any bucket that did not fill (unused
    parameters contribute zeros), waits for the reduce-scatters, scales each rank's 1/G shard by 1/G (1/G of the
    elementwise work of scaling after an all-reduce) and ALL-GATHERS the shards back into the flat buffers.
    Why the two-step form: on the fully connected xGMI node every rank sends a different 1/G slice to
    each of its 7 peers at once -- all links busy in both phases -- where a ring all-reduce moves the whole buffer
    round one link at a time; and the shard is where a sharded optimizer / gradient norm would hook in.
    Buckets are padded to a multiple of G elements.  Use ``zero_grad()`` of this object (it keeps the views)."""

    def __init__(self, params, bucket_mb: float = 64.0, first_bucket_mb: float = 8.0, group=None, timing: bool = False):
        self.group = group
        self.timing = timing                   # record per-bucket spans of the two collectives (bucket_times_ms())
        self._times = []
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.params = [p for p in reversed(list(params)) if p.requires_grad]
        self.buckets = []                      # dict(flat, params, pending, work)
        self._where = {}                       # param -> (bucket index, byte offset in its flat buffer)
        cap = int(first_bucket_mb * (1 << 20)) // 4
        cur, used = [], 0
        for p in self.params:
            if cur and used + p.numel() > cap:
                self._close(cur)
                cur, used, cap = [], 0, int(bucket_mb * (1 << 20)) // 4
            cur.append(p)
            used += p.numel()
        if cur:
            self._close(cur)
        self._hooks = [p.register_post_accumulate_grad_hook(self._arrived) for p in self.params]
        self.exposed_wait_s = 0.0              # host time finish() spent waiting on collectives (last call)
        self._begin()

    def _close(self, plist):
        n = sum(p.numel() for p in plist)
        quantum = 4 * self.world                                   # shards of whole float4s
        npad = (n + quantum - 1) // quantum * quantum
        flat = torch.zeros(npad, dtype=plist[0].dtype, device=plist[0].device)
        off = 0
        for p in plist:
            p.grad = flat[off:off + p.numel()].view_as(p)
            self._where[p] = (len(self.buckets), off * p.element_size())
            off += p.numel()
        shard = flat.new_empty(npad // self.world)
        self.buckets.append(dict(flat=flat, shard=shard, used=n, params=plist, pending=len(plist), work=None,
                                 launched=False))

    def _begin(self):
        for b in self.buckets:
            b["pending"], b["work"], b["launched"] = len(b["params"]), None, False

    def _mark(self, b, key):
        """Timing mark: a device event on the current stream (collectives are stream-ordered through Work.wait()), or
        the host clock when the exchange is host-staged."""
        if not self.timing:
            return
        if b["flat"].is_cuda and not _staged(b["flat"], self.group):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            b[key] = ev
        else:
            import time
            b[key] = time.perf_counter()

    def _launch(self, b):
        b["launched"] = True
        if self.world == 1:
            return
        self._mark(b, "t_rs0")
        if _staged(b["flat"], self.group):                  # gloo + device tensor (CPU-hosted tests): synchronous
            h, hs = b["flat"].detach().cpu(), torch.empty(b["shard"].numel())
            dist.reduce_scatter_tensor(hs, h, group=self.group)
            b["shard"].copy_(hs)
        else:
            b["work"] = dist.reduce_scatter_tensor(b["shard"], b["flat"], group=self.group, async_op=True)

    def _arrived(self, p):
        bi, off = self._where[p]
        b = self.buckets[bi]
        if p.grad is None or p.grad.data_ptr() != b["flat"].data_ptr() + off:
            raise RuntimeError("a parameter's .grad no longer aliases its bucket: use GradBuckets.zero_grad(), "
                               "not zero_grad(set_to_none=True)")
        b["pending"] -= 1
        if b["pending"] == 0 and not b["launched"]:
            self._launch(b)

    def finish(self):
        """Call after ``backward()``: every gradient is the mean over ranks when this returns."""
        import time
        for b in self.buckets:
            if not b["launched"]:
                self._launch(b)
        if self.world == 1:
            self.exposed_wait_s = 0.0
            self._begin()
            return
        t0 = time.perf_counter()
        gathers = []
        for b in self.buckets:                              # in launch order: the early buckets are done first
            if b["work"] is not None:
                b["work"].wait()
            self._mark(b, "t_rs1")
            b["shard"].mul_(1.0 / self.world)
            self._mark(b, "t_ag0")
            if _staged(b["flat"], self.group):
                hs, h = b["shard"].detach().cpu(), torch.empty(b["flat"].numel())
                dist.all_gather_into_tensor(h, hs, group=self.group)
                b["flat"].copy_(h)
                self._mark(b, "t_ag1")
            else:
                gathers.append((b, dist.all_gather_into_tensor(b["flat"], b["shard"], group=self.group, async_op=True)))
        for b, w in gathers:
            w.wait()
            self._mark(b, "t_ag1")
        self.exposed_wait_s = time.perf_counter() - t0
        if self.timing:
            self._times = [{k: b.get(k) for k in ("t_rs0", "t_rs1", "t_ag0", "t_ag1")} | {"bytes": b["flat"].numel() * 4}
                           for b in self.buckets]
        self._begin()

    def bucket_times_ms(self):
        """Per bucket of the LAST finish() (timing=True): span from the launch of its reduce-scatter (inside the backward)
        to its completion on the compute stream, and the span of its all-gather.  Device-event times for RCCL, host
        times when the exchange is host-staged.  Synchronises the events it reads."""
        out = []
        for t in self._times:
            def span(a, b):
                if a is None or b is None:
                    return None
                if isinstance(a, float):
                    return round((b - a) * 1e3, 4)
                b.synchronize()
                return round(a.elapsed_time(b), 4)
            out.append({"bytes": t["bytes"], "reduce_scatter_span_ms": span(t["t_rs0"], t["t_rs1"]),
                        "all_gather_span_ms": span(t["t_ag0"], t["t_ag1"])})
        return out

    def zero_grad(self):
        for b in self.buckets:
            b["flat"].zero_()
        self._begin()

    def close(self):
        """Detach from the parameters (remove the hooks; ``.grad`` tensors stay as they are)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []

    @property
    def bucket_sizes(self):
        return [b["flat"].numel() * 4 for b in self.buckets]


# ------------------------------------------------------------------------------------------- module wrapper
class DataParallel(nn.Module):
    """``DataParallel(AVSeparationTransformer(...).to(dev))``: broadcasts rank 0's parameters and buffers, makes the
    BatchNorm statistics span all ranks, and owns the gradient buckets.  A training step reads like the reference's
    (demo.py:96-106) plus one line::

        dp.zero_grad(); sep, _ = dp(mixed_shard, lips_shard)
        loss = criterion(sep, targets_shard, group=dp.group); loss.backward()
        dp.reduce_gradients()                      # <- the exchange step
        torch.nn.utils.clip_grad_norm_(dp.parameters(), 1.0); optimizer.step()
    """

    def __init__(self, module: nn.Module, group=None, bucket_mb: float = 64.0, first_bucket_mb: float = 8.0,
                 sync_bn: bool = True, timing: bool = False):
        super().__init__()
        self.module = module
        group = group if group is not None else dist.group.WORLD
        self.group = group
        self.world = dist.get_world_size(group)
        src = dist.get_global_rank(group, 0)
        with torch.no_grad():
            for t in list(module.parameters()) + [b for b in module.buffers() if b.dtype.is_floating_point]:
                if _staged(t, group):
                    h = t.detach().cpu()
                    dist.broadcast(h, src, group=group)
                    t.copy_(h)
                else:
                    dist.broadcast(t, src, group=group)    # into the tensor itself: bumps its version counter, which
                                                           # is what the fused path's weight-change check reads
        if hasattr(module, "invalidate_weights"):
            module.invalidate_weights()                    # ranks > 0 may have run a forward before being wrapped
        object.__setattr__(module, "_dp_group", group if (sync_bn and self.world > 1) else None)
        old = getattr(module, "_dp_buckets", None)
        if old is not None:               # re-wrapping the same module: the previous hooks would count arrivals twice
            old.close()
        self.buckets = GradBuckets(module.parameters(), bucket_mb, first_bucket_mb, group, timing=timing)
        object.__setattr__(module, "_dp_buckets", self.buckets)

    def forward(self, mixed_spec, lip_frames):
        return self.module(mixed_spec, lip_frames)

    def reduce_gradients(self):
        self.buckets.finish()

    def zero_grad(self, set_to_none: bool = False):
        self.buckets.zero_grad()


def shard_range(rank: int, world: int, batch: int) -> range:
    """Clips [rank*B/G, (rank+1)*B/G) of a global batch; the batch must divide evenly so that the
    mean of per-rank mean losses equals the reference's batch mean."""
    if batch % world:
        raise ValueError(f"global batch {batch} does not divide over {world} ranks")
    per = batch // world
    return range(rank * per, (rank + 1) * per)
