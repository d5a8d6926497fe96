"""SI-SNR / L1 permutation-invariant loss, host-side PyTorch as BASELINE.json's north_star asks
("Host code stays Python on PyTorch-ROCm for ... the SI-SNR/L1 loss").  Semantics of
``/root/reference/src/av_separation/losses.py``: ``si_snr`` 14-42, ``SeparationLoss`` 45-73.

Details that matter for parity: all non-batch dims are flattened TOGETHER (for a
4-D input the speakers are part of one long vector), both vectors are mean-centred, eps=1e-8 enters in
three places, and ONE permutation is chosen for the whole batch from the batch-mean loss.
"""
from itertools import permutations

import torch
import torch.nn as nn


def si_snr(estimate: torch.Tensor, target: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    """Mean scale-invariant SNR in dB over the batch; inputs (B, ...) of equal shape."""
    b = estimate.shape[0]
    e = estimate.reshape(b, -1)
    t = target.reshape(b, -1)
    e = e - e.mean(dim=-1, keepdim=True)
    t = t - t.mean(dim=-1, keepdim=True)
    scale = (e * t).sum(dim=-1, keepdim=True) / ((t * t).sum(dim=-1, keepdim=True) + eps)
    s_target = scale * t
    e_noise = e - s_target
    ratio = (s_target * s_target).sum(dim=-1) / ((e_noise * e_noise).sum(dim=-1) + eps)
    return (10 * torch.log10(ratio + eps)).mean()


class SeparationLoss(nn.Module):
    """min over speaker permutations of  -si_snr + l1_weight * L1  (batch-global permutation)."""

    def __init__(self, l1_weight: float = 0.5):
        super().__init__()
        self.l1_weight = l1_weight

    def forward(self, separated: torch.Tensor, targets: torch.Tensor, group=None) -> torch.Tensor:
        """``group``: data-parallel process group (equal shards).  The reference picks the permutation from the mean
        over the WHOLE batch, so the candidates' values are averaged over ranks before comparing; the returned
        (differentiable) loss is this rank's term of the chosen permutation -- its mean over ranks is the
        reference's batch loss."""
        n_spk = separated.shape[1]
        cands = []
        for order in permutations(range(n_spk)):
            cand = separated[:, list(order)]
            cands.append(self.l1_weight * (cand - targets).abs().mean() - si_snr(cand, targets))
        ranking = cands
        if group is not None:
            import torch.distributed as dist
            from .parallel import all_reduce_sum_
            if dist.get_world_size(group) > 1:
                ranking = all_reduce_sum_(torch.stack([c.detach() for c in cands]), group) / dist.get_world_size(group)
        best = 0
        for i in range(1, len(cands)):
            # strict "<" like the reference (losses.py:70): ties keep the earlier permutation
            if ranking[i] < ranking[best]:
                best = i
        return cands[best]
