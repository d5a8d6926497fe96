"""si_snr / SeparationLoss against known answers computed by the reference (tests/golden/losses.npz) and
the identities its own tests check (tests/test_model.py:297-325)."""
import numpy as np
import torch

from av_separation.losses import SeparationLoss, si_snr


def test_known_answers(golden):
    g = golden("losses")
    est, tgt = torch.from_numpy(g["est"]), torch.from_numpy(g["tgt"])
    assert abs(float(si_snr(est, tgt)) - g["si_snr_4d"]) < 1e-5
    assert abs(float(si_snr(est[:, 0], tgt[:, 0])) - g["si_snr_3d"]) < 1e-5
    assert abs(float(si_snr(tgt, tgt)) - g["si_snr_self"]) < 1e-3
    for s in (2, 3):
        for w in (0.5, 0.0):
            got = float(SeparationLoss(l1_weight=w)(est[:, :s], tgt[:, :s]))
            assert abs(got - g[f"sep_loss_S{s}_w{w}"]) < 1e-5
    assert abs(float(SeparationLoss(0.5)(tgt[:, [2, 0, 1]] * 0.9, tgt)) - g["sep_loss_perm"]) < 1e-3


def test_gradient_matches_reference(golden):
    g = golden("losses")
    e = torch.from_numpy(g["est"]).clone().requires_grad_(True)
    SeparationLoss(0.5)(e, torch.from_numpy(g["tgt"])).backward()
    assert np.abs(e.grad.numpy() - g["sep_loss_grad"]).max() < 1e-7


def test_identities():
    x = torch.rand(2, 65, 32) + 0.1
    assert float(si_snr(x, x)) > 20
    a = torch.zeros(1, 4, 4); a[0, 0, :] = 1
    b = torch.zeros(1, 4, 4); b[0, 1, :] = 1
    assert float(si_snr(a, b)) < 0
    loss = SeparationLoss()(torch.rand(2, 2, 65, 32), torch.rand(2, 2, 65, 32))
    assert loss.dim() == 0 and not torch.isnan(loss)


def test_pairwise_ranking_equals_candidate_by_candidate_evaluation():
    """SeparationLoss ranks the S! speaker orders from pairwise statistics (one pass) instead of S! evaluations of the
    loss; the ranking values must agree with the direct ones and pick the same order (reference: losses.py:61-73)."""
    from itertools import permutations
    torch.manual_seed(0)
    crit = SeparationLoss(0.5)
    for n_spk in (2, 3, 4):
        orders = list(permutations(range(n_spk)))
        for trial in range(4):
            tgt = torch.rand(3, n_spk, 17, 12) * 3
            est = tgt[:, list(orders[(trial * 5 + 1) % len(orders)])] * (0.9 + 0.03 * trial) + 0.05 * torch.randn_like(tgt)
            table = crit._order_table(n_spk, est.device)
            fast = crit._ranking(est, tgt, table)
            slow = torch.stack([crit._value(est[:, list(o)], tgt) for o in orders])
            assert float((fast - slow).abs().max()) < 1e-3 and int(fast.argmin()) == int(slow.argmin())
            assert float(crit(est, tgt)) == float(slow.min())
