"""Seeded shape fuzzing of the drop-in on the GPU: random (freq_bins, d_model, heads, layers, speakers, T, N, H, W, B)
within the ABI's stated limits, (1) the fused inference path against the numpy oracle, (2) the train-mode autograd
path (outputs, loss and every parameter gradient, dropout 0) against oracle/torch_cpu.forward_train -- both pinned on the
reference's own outputs/gradients in tests/test_oracle.py.  Catches what hand-picked edge cases miss: ragged tiles,
odd frame sizes on either conv path, head dims between the specialised ones, T < N, single-frame clips."""
import math
import os
import random

import numpy as np
import pytest
import torch

from helpers import maxabs
from oracle import numpy_forward as onp
from oracle import seeded, torch_cpu

pytestmark = pytest.mark.gpu


def _draw(rng, train):
    while True:
        h = rng.choice([1, 2, 4, 8])
        dh = rng.choice([16, 32, 48, 64] if train else [4, 8, 12, 16, 24, 32, 64, 128])
        d = h * dh
        if d % 32 == 0 and d <= (128 if train else 256):
            break
    cfg = dict(freq_bins=rng.randint(3, 70), d_model=d, nhead=h, num_encoder_layers=rng.randint(0, 2),
               num_fusion_layers=rng.randint(0, 2), num_speakers=rng.randint(1, 3))
    dims = dict(B=rng.randint(1, 5), T=rng.randint(1, 70), N=rng.randint(1, 12), H=rng.randint(3, 40), W=rng.randint(3, 40))
    return cfg, dims


def _model(dev, cfg, seed, dropout=0.0):
    import av_separation as av
    torch.manual_seed(seed)
    m = av.AVSeparationTransformer(dropout=dropout, **cfg)
    g = torch.Generator().manual_seed(seed + 1)
    sd = m.state_dict()
    for k, v in sd.items():
        if k.endswith("running_mean"):
            v.copy_(torch.rand(v.shape, generator=g) * 0.6 - 0.3)
        elif k.endswith("running_var"):
            v.copy_(torch.rand(v.shape, generator=g) + 0.5)
    return m.to(dev)


@pytest.mark.parametrize("case", range(int(os.environ.get("AVSEP_FUZZ_INFER", "24"))))     # more with the env knob
def test_fuzz_inference_against_numpy_oracle(case):
    rng = random.Random(1000 + case)
    cfg, dm = _draw(rng, train=False)
    dev = torch.device("cuda:0")
    m = _model(dev, cfg, case).eval()
    mixed, lips = seeded.inputs(500 + case, dm["B"], cfg["freq_bins"], dm["T"], dm["N"], dm["H"], dm["W"])
    x, l = torch.from_numpy(mixed).to(dev), torch.from_numpy(lips).to(dev)
    with torch.no_grad():
        sep, masks = m(x, l)
        for b in sorted({0, dm["B"] - 1}):                           # a clip alone: the bits it has inside the batch
            s1, m1 = m(x[b:b + 1], l[b:b + 1])
            assert torch.equal(m1, masks[b:b + 1]) and torch.equal(s1, sep[b:b + 1]), (b, cfg, dm)
    state = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    rs, rm = onp.forward(state, mixed, lips, cfg["nhead"], cfg["num_speakers"])
    assert masks.shape == (dm["B"], cfg["num_speakers"], cfg["freq_bins"], dm["T"]), (cfg, dm)
    assert maxabs(masks.cpu().numpy(), rm) < 4e-6, (cfg, dm)
    assert maxabs(sep.cpu().numpy(), rs) < 4e-6 * max(1.0, float(np.abs(mixed).max())), (cfg, dm)


def _draw_wide(rng):
    """d_model >= 512: the split-precision path (two-term fp16 GEMMs with static exponents, three-term bf16 where there is no static
    bound, split-precision attention at dh = 64 and 128+ keys), incl. head dims that keep the fp32 attention, odd S*F (fp32 mask
    head), zero layers, and row counts on both sides of the 256 x 128 kernels' threshold"""
    d = rng.choice([512, 512, 512, 640, 768, 1024])
    h = rng.choice([x for x in (4, 8, 16) if d % x == 0 and (d // x) % 4 == 0 and d // x <= 128])
    cfg = dict(freq_bins=rng.choice([257, rng.randint(40, 300)]), d_model=d, nhead=h, num_encoder_layers=rng.randint(0, 2),
               num_fusion_layers=rng.randint(0, 2), num_speakers=rng.randint(1, 3))
    big = rng.random() < 0.4
    dims = dict(B=rng.randint(10, 14) if big else rng.randint(1, 4), T=rng.randint(200, 300) if big else rng.randint(1, 300),
                N=rng.randint(1, 30), H=rng.randint(8, 40), W=rng.randint(8, 40))
    return cfg, dims


@pytest.mark.parametrize("case", range(int(os.environ.get("AVSEP_FUZZ_WIDE", "12"))))
def test_fuzz_wide_models_against_torch_cpu_port(case):
    rng = random.Random(31000 + case)
    cfg, dm = _draw_wide(rng)
    dev = torch.device("cuda:0")
    m = _model(dev, cfg, 300 + case).eval()
    mixed, lips = seeded.inputs(1500 + case, dm["B"], cfg["freq_bins"], dm["T"], dm["N"], dm["H"], dm["W"])
    x, l = torch.from_numpy(mixed).to(dev), torch.from_numpy(lips).to(dev)
    with torch.no_grad():
        sep, masks = m(x, l)
        for b in sorted({0, 1 % dm["B"], dm["B"] // 2, dm["B"] - 1}):  # a clip alone: the bits it has inside the batch (odd N: both frame parities)
            s1, m1 = m(x[b:b + 1], l[b:b + 1])
            assert torch.equal(m1, masks[b:b + 1]) and torch.equal(s1, sep[b:b + 1]), (b, cfg, dm)
    state = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    rs, rm = torch_cpu.forward(state, torch.from_numpy(mixed), torch.from_numpy(lips), cfg["nhead"], cfg["num_speakers"])
    assert masks.shape == (dm["B"], cfg["num_speakers"], cfg["freq_bins"], dm["T"]), (cfg, dm)
    assert torch.isfinite(masks).all() and torch.isfinite(sep).all(), (cfg, dm)
    assert maxabs(masks.cpu().numpy(), rm.contiguous().numpy()) < 4e-6, (cfg, dm)
    assert maxabs(sep.cpu().numpy(), rs.contiguous().numpy()) < 4e-6 * max(1.0, float(np.abs(mixed).max())), (cfg, dm)


def test_two_term_bounds_follow_weight_updates():
    """The static exponents of the two-term path are derived from the weights when they are packed (avsep_finalize_weights).  A weight
    update in eval mode -- in place under no_grad, which the per-call change check sees -- must re-derive them: LayerNorm gains and
    biases grown 300x, in-projection rows grown 40x and an out-projection shrunk 1e-4x between two forwards of the same module; the
    second forward stays finite and at the oracle's distance (an exponent left over from the first would overflow fp16 or lose the
    small operand)."""
    dev = torch.device("cuda:0")
    cfg = dict(freq_bins=129, d_model=512, nhead=8, num_encoder_layers=1, num_fusion_layers=1, num_speakers=2)
    B, T, N, H, W = 2, 160, 7, 16, 16
    m = _model(dev, cfg, 91).eval()
    mixed, lips = seeded.inputs(777, B, cfg["freq_bins"], T, N, H, W)
    x, l = torch.from_numpy(mixed).to(dev), torch.from_numpy(lips).to(dev)

    def check():
        with torch.no_grad():
            _, masks = m(x, l)
        assert torch.isfinite(masks).all()
        state = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        _, r32 = torch_cpu.forward(state, torch.from_numpy(mixed), torch.from_numpy(lips), cfg["nhead"], cfg["num_speakers"])
        s64 = {k: (v.double() if v.is_floating_point() else v) for k, v in state.items()}
        _, r64 = torch_cpu.forward(s64, torch.from_numpy(mixed).double(), torch.from_numpy(lips).double(), cfg["nhead"], cfg["num_speakers"])
        e32 = maxabs(r32.contiguous().numpy().astype(np.float64), r64.contiguous().numpy())
        e = maxabs(masks.cpu().numpy().astype(np.float64), r64.contiguous().numpy())
        assert e < 2.0 * e32 + 4e-6, (e, e32)

    check()
    sd = dict(m.named_parameters())
    with torch.no_grad():
        for k in ("audio_encoder.transformer.layers.0.norm1.weight", "audio_encoder.transformer.layers.0.norm1.bias",
                  "fusion.layers.0.norm2.weight", "fusion.layers.0.norm2.bias", "fusion.norm.weight"):
            sd[k].mul_(300.0)
        sd["audio_encoder.transformer.layers.0.self_attn.in_proj_weight"].mul_(40.0 / 300.0)   # q, k, v bounds move, the scores stay sane
        sd["fusion.layers.0.cross_attn.out_proj.weight"].mul_(1e-4)
        sd["fusion.layers.0.ff.0.weight"].mul_(1.0 / 300.0)
        sd["decoder.decoder.0.weight"].mul_(1.0 / 300.0)
    check()


@pytest.mark.parametrize("scale", [1e-5, 1e-2, 1.0, 3e2, 1e5])
def test_visual_stream_magnitude_sweep_on_the_two_term_path(scale):
    """The cross-attention of a d_model >= 512 model runs on two fp16 terms with exponents taken PER CLIP from the magnitude of the
    resized visual stream (clip_exp_kernel), the self-attention with static bounds of its in-projection.  Nothing normalises the visual
    residual stream before it is resized, so its magnitude is whatever frame_proj makes it: swept here over ten decades (one clip of
    the batch a further 1000x louder than the others).  Outputs stay finite and as close to the float64 oracle as the fp32 oracle is
    (2x its error + MASK_TOL), and the loud clip alone has the bits it has inside the batch."""
    dev = torch.device("cuda:0")
    cfg = dict(freq_bins=129, d_model=512, nhead=8, num_encoder_layers=1, num_fusion_layers=2, num_speakers=2)
    B, T, N, H, W = 3, 200, 9, 16, 16
    m = _model(dev, cfg, 77).eval()
    with torch.no_grad():
        m.visual_encoder.frame_proj.weight.mul_(scale)
        m.visual_encoder.frame_proj.bias.mul_(scale)
    mixed, lips = seeded.inputs(4242, B, cfg["freq_bins"], T, N, H, W)
    lips[1] *= 1000.0
    x, l = torch.from_numpy(mixed).to(dev), torch.from_numpy(lips).to(dev)
    with torch.no_grad():
        sep, masks = m(x, l)
        s1, m1 = m(x[1:2], l[1:2])
    assert torch.isfinite(masks).all() and torch.isfinite(sep).all()
    assert torch.equal(m1, masks[1:2]) and torch.equal(s1, sep[1:2])
    state = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    _, r32 = torch_cpu.forward(state, torch.from_numpy(mixed), torch.from_numpy(lips), cfg["nhead"], cfg["num_speakers"])
    s64 = {k: (v.double() if v.is_floating_point() else v) for k, v in state.items()}
    _, r64 = torch_cpu.forward(s64, torch.from_numpy(mixed).double(), torch.from_numpy(lips).double(), cfg["nhead"], cfg["num_speakers"])
    e32 = maxabs(r32.contiguous().numpy().astype(np.float64), r64.contiguous().numpy())
    e = maxabs(masks.cpu().numpy().astype(np.float64), r64.contiguous().numpy())
    assert e < 2.0 * e32 + 4e-6, (scale, e, e32)


@pytest.mark.parametrize("case", range(int(os.environ.get("AVSEP_FUZZ_TRAIN", "10"))))
def test_fuzz_training_gradients_against_torch_cpu_port(case):
    from av_separation.losses import SeparationLoss
    rng = random.Random(7000 + case)
    cfg, dm = _draw(rng, train=True)
    dm["B"] = max(dm["B"], 2)                      # BatchNorm batch statistics need more than one value per channel
    dev = torch.device("cuda:0")
    m = _model(dev, cfg, 100 + case).train()
    mixed, lips = seeded.inputs(900 + case, dm["B"], cfg["freq_bins"], dm["T"], dm["N"], dm["H"], dm["W"])
    tg = torch.rand(dm["B"], cfg["num_speakers"], cfg["freq_bins"], dm["T"], generator=torch.Generator().manual_seed(case))
    tg = tg * torch.from_numpy(mixed).unsqueeze(1)
    state = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    for k, v in state.items():
        if v.is_floating_point() and "running_" not in k and not k.endswith(".pe"):
            v.requires_grad_()
    torch_cpu.RELU_PROBE = []
    try:
        rsep, _ = torch_cpu.forward_train(state, torch.from_numpy(mixed), torch.from_numpy(lips), cfg["nhead"],
                                          cfg["num_speakers"])
        nearest_kink = min(torch_cpu.RELU_PROBE)
    finally:
        torch_cpu.RELU_PROBE = None
    # A ReLU input within rounding of 0: that unit's mask bit is not determined at fp32, and one flipped bit moves a
    # whole row of a weight gradient by O(1) of its size (case 98: one hidden unit, one row of linear1.weight off by
    # 5 %, everything else at 1e-3).  Such cases are compared in the L2 norm instead of element by element.
    kinky = nearest_kink < 2e-6
    rloss = SeparationLoss(0.5)(rsep, tg)
    rloss.backward()
    sep, _ = m(torch.from_numpy(mixed).to(dev), torch.from_numpy(lips).to(dev))
    loss = SeparationLoss(0.5)(sep, tg.to(dev))
    loss.backward()
    assert abs(float(loss.detach()) - float(rloss.detach())) < 1e-4 * max(1.0, abs(float(rloss.detach()))), (cfg, dm)
    worst = 0.0
    for k, p in m.named_parameters():
        ref = state[k].grad
        if ref is None:                             # e.g. no layers -> parameter unused on both sides
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        if kinky:
            err = float((p.grad.cpu() - ref).norm()) / max(1e-3, float(ref.norm()))
            assert err < 5e-2, (k, err, "L2", cfg, dm)
        else:
            err = maxabs(p.grad.cpu().numpy(), ref.numpy()) / max(1e-3, float(ref.abs().max()))
            assert err < 5e-3, (k, err, cfg, dm)
        worst = max(worst, err)
    for k, v in m.state_dict().items():
        if "running_" in k:
            assert maxabs(v.cpu().numpy(), state[k].numpy()) < 1e-4, (k, cfg, dm)
    assert math.isfinite(worst)
