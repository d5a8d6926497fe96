"""Training path on the GPU against the reference's own train-mode forward/backward
(tests/golden/train_*.npz, made by make_golden.py::make_train with dropout 0): forward outputs with BatchNorm
batch statistics, the PIT loss, EVERY parameter gradient and the updated BatchNorm buffers."""
import numpy as np
import pytest
import torch

from helpers import maxabs
from oracle import seeded

pytestmark = pytest.mark.gpu


def _build(g, dev):
    import av_separation as av
    c = g["config"]
    m = av.AVSeparationTransformer(c["F"], c["d"], c["h"], c["Le"], c["Lf"], c["S"], dropout=0.0)
    shapes = seeded.model_shapes(c["F"], c["d"], c["h"], c["Le"], c["Lf"], c["S"])
    state = seeded.fill_state(shapes, c["seed"], gain=float(g["gain"]))
    sd = m.state_dict()
    for k, v in state.items():
        sd[k] = torch.from_numpy(np.ascontiguousarray(v))
    m.load_state_dict(sd)
    return m.to(dev).train()


@pytest.mark.parametrize("name", ["train_tiny", "train_odd"])
def test_train_forward_backward_matches_reference(golden, name):
    from av_separation.losses import SeparationLoss
    g = golden(name)
    c = g["config"]
    dev = torch.device("cuda:0")
    m = _build(g, dev)
    mx, lp = seeded.inputs(c["seed"], c["B"], c["F"], c["T"], c["N"], c["H"], c["W"])
    sep, masks = m(torch.from_numpy(mx).to(dev), torch.from_numpy(lp).to(dev))
    assert sep.requires_grad and masks.shape == (c["B"], c["S"], c["F"], c["T"])
    assert maxabs(masks.detach().cpu().numpy(), g["masks"]) < 1e-5
    assert maxabs(sep.detach().cpu().numpy(), g["separated"]) < 1e-5 * max(1.0, float(np.abs(mx).max()))
    loss = SeparationLoss(l1_weight=0.5)(sep, torch.from_numpy(g["targets"]).to(dev))
    assert abs(float(loss) - float(g["loss"])) < 2e-5
    loss.backward()
    worst, bad = 0.0, []
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        got = p.grad.detach().cpu().numpy()
        if "g." + k in g:
            ref = g["g." + k]
            scale = max(1e-3, float(np.abs(ref).max()))
            err = maxabs(got, ref) / scale
        else:
            ref = g["g." + k + ".slice"]
            scale = max(1e-3, float(np.abs(ref).max()))
            err = maxabs(got.reshape(-1)[::5], ref) / scale
            assert abs(np.linalg.norm(got.astype(np.float64)) - g["g." + k + ".norm"]) < 1e-3 * max(1e-3, g["g." + k + ".norm"])
        worst = max(worst, err)
        if err >= 2e-3:
            bad.append((k, round(err, 4)))
    assert not bad, bad                      # errors are relative to the largest gradient entry of each tensor
    # BatchNorm buffers after one training forward (momentum 0.1, unbiased variance)
    sd = m.state_dict()
    for k in g:
        if k.startswith("buf."):
            got = sd[k[4:]].cpu().numpy()
            assert maxabs(got, g[k]) < 1e-5, k
    print(f"{name}: worst relative gradient error {worst:.2e}")


def test_training_step_reduces_loss_and_eval_sees_new_weights():
    """A few Adam steps on the HIP training path (the reference's quick_train recipe, demo.py:83-113) reduce the
    loss, and the inference path re-packs the updated weights/BN statistics afterwards."""
    import av_separation as av
    from av_separation.losses import SeparationLoss
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = av.AVSeparationTransformer(freq_bins=65, d_model=64, nhead=4, num_encoder_layers=1, num_fusion_layers=1,
                                   num_speakers=2, dropout=0.0).to(dev)
    ds = av.SyntheticAVDataset(num_samples=16, sample_rate=8000, duration=0.496, n_fft=128, hop_length=128, num_frames=5,
                               frame_h=16, frame_w=16)
    items = [ds[i] for i in range(8)]
    mixed = torch.stack([x["mixed_spec"] for x in items]).to(dev)
    lips = torch.stack([x["lip_frames"] for x in items]).to(dev)
    tg = torch.stack([x["clean_specs"] for x in items]).to(dev)
    m.eval()
    with torch.no_grad():
        _, masks_before = m(mixed, lips)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=2e-3)
    crit = SeparationLoss(0.5)
    losses = []
    for _ in range(12):
        opt.zero_grad()
        sep, _ = m(mixed, lips)
        loss = crit(sep, tg)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < losses[0] - 1.0, losses
    m.eval()
    with torch.no_grad():
        _, masks_after = m(mixed, lips)
    assert float((masks_after - masks_before).abs().max()) > 1e-2
