"""Pins the CPU oracle (oracle/numpy_forward.py) against the golden vectors produced by the reference
(tests/golden/make_golden.py).  Tolerances: the reference's own fp32 noise floor is ~2e-7 on masks
(SURVEY.md §8(c)); the oracle must sit within 2e-6 of the fp32 goldens and 1e-6 of the fp64 goldens
(float64 mode), far inside the 1e-4 hard gate of BASELINE.json."""
import os
import sys

import numpy as np
import pytest

from oracle import numpy_forward as onp
from oracle import seeded
from helpers import golden_state, golden_inputs, sliced, maxabs

FULL = ["fwd_tiny", "fwd_odd", "fwd_down", "fwd_t1", "trained_tiny"]
BIG = ["fwd_cfg1", "fwd_cfg3", "fwd_cfg4", "fwd_cfg5"]


def _run(g, dtype, taps=None):
    c = g["config"]
    state = golden_state(g)
    if "pe" in g:   # the reference's own pe rows (torch sin/cos), so numpy-vs-torch trig does not enter
        L, d = g["pe"].shape
        pe = np.zeros((1, L, d), np.float32)
        pe[0] = g["pe"]
        state["audio_encoder.pos_enc.pe"] = pe
        state["visual_encoder.pos_enc.pe"] = pe
    mixed, lips = golden_inputs(g)
    return onp.forward(state, mixed, lips, c["h"], c["S"], dtype=dtype, taps=taps)


@pytest.mark.parametrize("name", FULL)
def test_oracle_full_fixtures_fp32(golden, name):
    g = golden(name)
    taps = {}
    sep, masks = _run(g, np.float32, taps)
    assert masks.shape == g["masks"].shape
    assert maxabs(masks, g["masks"]) < 2e-6
    scale = max(1.0, float(np.abs(golden_inputs(g)[0]).max()))
    assert maxabs(sep, g["separated"]) < 2e-6 * scale
    for k, v in taps.items():
        ref = g["tap." + k]
        tol = 2e-5 * max(1.0, float(np.abs(ref).max()))
        assert maxabs(v.reshape(ref.shape), ref) < tol, k


@pytest.mark.parametrize("name", FULL)
def test_oracle_full_fixtures_fp64(golden, name):
    g = golden(name)
    sep, masks = _run(g, np.float64)
    # pe is float32 in both; everything else float64 -> agreement to ~1e-12 would need identical op
    # order; 1e-9 shows the formulas are the same ones
    assert maxabs(masks, g["masks64"]) < 1e-9
    assert maxabs(sep, g["separated64"]) < 1e-7


@pytest.mark.parametrize("name", BIG)
def test_oracle_baseline_configs(golden, name):
    g = golden(name)
    taps = {}
    sep, masks = _run(g, np.float32, taps)
    c = g["config"]
    assert masks.shape == (c["B"], c["S"], c["F"], c["T"])
    assert maxabs(sliced(masks, 7), g["masks.slice"]) < 5e-6
    assert maxabs(sliced(masks, 7), g["masks64.slice"]) < 5e-6
    scale = max(1.0, float(np.abs(golden_inputs(g)[0]).max()))
    assert maxabs(sliced(sep, 7), g["separated.slice"]) < 5e-6 * scale
    assert abs(masks.astype(np.float64).sum() - g["masks.sum"]) < 1e-6 * g["masks.abssum"]
    for k, v in taps.items():
        ref = g["tap." + k + ".slice"]
        tol = 5e-5 * max(1.0, float(np.abs(ref).max()))
        assert maxabs(sliced(v, 97), ref) < tol, k


def test_oracles_on_the_reference_trained_config1_model(golden):
    """trained_cfg1: BASELINE config 1's model (d = 256, 2 + 2 layers) after the reference's own quick_train
    (demo.py:83-113), evaluated by the reference on SyntheticAVDataset items 0, 1: both CPU oracles against the reference's
    full outputs and every stage tap."""
    import torch
    from oracle import torch_cpu
    g = golden("trained_cfg1")
    c = g["config"]
    state = golden_state(g)
    mixed, lips = golden_inputs(g)
    scale = max(1.0, float(np.abs(mixed).max()))
    taps = {}
    sep, masks = onp.forward(state, mixed, lips, c["h"], c["S"], dtype=np.float32, taps=taps)
    assert maxabs(masks, g["masks"]) < 5e-6 and maxabs(sep, g["separated"]) < 5e-6 * scale
    for k, v in taps.items():
        ref = g["tap." + k + ".slice"]
        assert maxabs(sliced(v, 97), ref) < 5e-5 * max(1.0, float(np.abs(ref).max())), k
    _, m64 = onp.forward(state, mixed, lips, c["h"], c["S"], dtype=np.float64)
    assert maxabs(m64, g["masks64"]) < 1e-6          # float32 sin/cos of the pe table (numpy vs torch) is all that differs
    st = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in state.items()}
    sp, mk = torch_cpu.forward(st, torch.from_numpy(mixed), torch.from_numpy(lips), c["h"], c["S"])
    assert maxabs(mk.contiguous().numpy(), g["masks"]) < 2e-6
    assert maxabs(sp.contiguous().numpy(), g["separated"]) < 2e-6 * scale
    assert float(g["eval.out_snr"]) - float(g["eval.in_snr"]) > 35.0     # the README's +37 dB recipe (README.md:61-65)


def test_oracle_on_the_reference_trained_d512_model(golden):
    """trained_d512: BASELINE config 3's shapes with 1 + 1 layers after the reference's own quick_train, the
    fixture that brings TRAINED weights to the d_model >= 512 split-precision kernels: the torch CPU port (the checker of the GPU test)
    against the reference's full outputs, the loss trajectory and the masks' saturation for the record."""
    import torch
    from oracle import torch_cpu
    g = golden("trained_d512")
    c = g["config"]
    assert c["d"] == 512 and c["h"] == 8 and c["Le"] == 1 and c["Lf"] == 1 and c["T"] == 251
    state = golden_state(g)
    mixed, lips = golden_inputs(g)
    scale = max(1.0, float(np.abs(mixed).max()))
    st = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in state.items()}
    sp, mk = torch_cpu.forward(st, torch.from_numpy(mixed), torch.from_numpy(lips), c["h"], c["S"])
    assert maxabs(mk.contiguous().numpy(), g["masks"]) < 2e-6
    assert maxabs(sp.contiguous().numpy(), g["separated"]) < 2e-6 * scale
    assert maxabs(g["masks"], g["masks64"]) < 2e-6
    assert float(g["losses"][0]) > -2.0 and float(g["losses"][-1]) < -30.0
    assert float(g["masks"].min()) < 1e-3 and float(g["masks"].max()) > 0.999


def test_seeded_generator_known_answers():
    # splitmix64 reference values (seed 0): published test vector of the algorithm
    z = seeded.splitmix64(0, 3)
    assert [int(x) for x in z] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]
    assert seeded.fnv1a64("") == 0xCBF29CE484222325
    assert seeded.fnv1a64("a") == 0xAF63DC4C8601EC8C
    t = seeded.tensor(5, "x", (4,), -1.0, 1.0)
    assert t.dtype == np.float32 and np.all(np.abs(t) <= 1)


def test_interp_matches_definition():
    # upsample 2 -> 4 with align_corners=False: src = (i+0.5)*0.5-0.5 -> [-0.25->0, 0.25, 0.75, 1.25->clamped]
    x = np.array([[[0.0], [1.0]]], dtype=np.float32)
    y = onp.interp_linear(x, 4)[0, :, 0]
    np.testing.assert_allclose(y, [0.0, 0.25, 0.75, 1.0], atol=1e-7)


@pytest.mark.parametrize("N,T", [(50, 63), (50, 501), (75, 251), (10, 32), (12, 5)])
def test_interp_index_is_fma(N, T):
    """torch's F.interpolate (the reference's model.py:115) rounds scale*(i+0.5)-0.5 once (fused); with two
    roundings a few output rows move by ~4e-6.  The oracle must agree with torch on the host to 1 ulp."""
    import torch
    x = seeded.tensor(4, "x", (2, N, 16), -3, 3)
    want = torch.nn.functional.interpolate(torch.from_numpy(x).permute(0, 2, 1), size=T, mode="linear",
                                           align_corners=False).permute(0, 2, 1).numpy()
    assert maxabs(onp.interp_linear(x, T), want) < 5e-7


@pytest.mark.parametrize("name", ["train_tiny", "train_odd"])
def test_torch_cpu_train_port_matches_reference_gradients(golden, name):
    """oracle/torch_cpu.forward_train (the CPU baseline of `bench.py --mode train`) against the reference's own
    train-mode forward/backward: loss, outputs, every parameter gradient, updated BatchNorm buffers."""
    import json
    import torch
    from oracle import seeded, torch_cpu
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "av-separation-transformer_amd"))
    from av_separation.losses import SeparationLoss
    g = golden(name)
    c = g["config"]
    shapes = seeded.model_shapes(c["F"], c["d"], c["h"], c["Le"], c["Lf"], c["S"])
    state = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in
             seeded.fill_state(shapes, c["seed"], gain=float(g["gain"])).items()}
    for k, t in state.items():
        if t.is_floating_point() and "running_" not in k and not k.endswith(".pe"):
            t.requires_grad_()
    mx, lp = seeded.inputs(c["seed"], c["B"], c["F"], c["T"], c["N"], c["H"], c["W"])
    sep, masks = torch_cpu.forward_train(state, torch.from_numpy(mx), torch.from_numpy(lp), c["h"], c["S"])
    loss = SeparationLoss(0.5)(sep, torch.from_numpy(g["targets"]))
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    assert maxabs(masks.detach().numpy(), g["masks"]) < 1e-6
    for k, t in state.items():
        if t.requires_grad:
            got = t.grad.numpy()
            if "g." + k in g:
                ref = g["g." + k]
                assert maxabs(got, ref) < 1e-5 * max(1e-3, float(np.abs(ref).max())) + 1e-7, k
            else:
                ref = g["g." + k + ".slice"]
                assert maxabs(got.reshape(-1)[::5], ref) < 1e-5 * max(1e-3, float(np.abs(ref).max())) + 1e-7, k
        elif "running_" in k:
            assert maxabs(t.numpy(), g["buf." + k]) < 1e-6, k


@pytest.mark.parametrize("name", FULL + ["fwd_cfg1"])
def test_torch_cpu_port_matches_reference_outputs(golden, name):
    """oracle/torch_cpu.forward (bench.py's cpu_baseline, kind "port") reproduces the reference's eval forward: it
    issues the same ATen kernels, so it sits at the reference's own noise floor."""
    import torch
    from oracle import torch_cpu
    g = golden(name)
    c = g["config"]
    state = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in golden_state(g).items()}
    mixed, lips = golden_inputs(g)
    for fast in (True, False):
        sep, masks = torch_cpu.forward(state, torch.from_numpy(mixed), torch.from_numpy(lips), c["h"], c["S"], fast=fast)
        sep, masks = sep.contiguous().numpy(), masks.contiguous().numpy()
        scale = max(1.0, float(np.abs(mixed).max()))
        if "masks" in g:
            assert maxabs(masks, g["masks"]) < 2e-6, (name, fast)
            assert maxabs(sep, g["separated"]) < 2e-6 * scale, (name, fast)
        else:
            assert maxabs(sliced(masks, 7), g["masks.slice"]) < 2e-6, (name, fast)
            assert maxabs(sliced(sep, 7), g["separated.slice"]) < 2e-6 * scale, (name, fast)


def test_fixture_low_bits_stream_known_answers():
    """tests/helpers.py::low_bits (the seeded low 16 bits of trained_d512's weight words) is part of a committed fixture: pin the stream
    (numpy's PCG64 with a SeedSequence of [seed, crc32(name)]) so that a numpy upgrade that changed it would fail HERE, not as a parity
    mystery."""
    from helpers import low_bits
    got = low_bits("audio_encoder.input_proj.0.weight", (2, 3), 20251005)
    assert got.dtype == np.uint32 and got.tolist() == [[47434, 37588, 18345], [48786, 49516, 49537]]
    assert int(low_bits("x", (1000,), 1).max()) < 1 << 16


def test_oracle_relu_forcing_takes_the_given_decisions():
    """oracle/torch_cpu.RELU_FORCE (the kink-aware gradient gate's hook): ReLU i returns x * mask_i; forcing the decisions the oracle
    makes by itself changes nothing, forcing a flipped unit changes exactly that unit's contribution; RELU_RECORD sees every ReLU."""
    import torch
    from oracle import torch_cpu
    F_, d, h, S = 9, 32, 4, 2
    shapes = seeded.model_shapes(F_, d, h, 1, 1, S)
    state = {k: torch.from_numpy(np.ascontiguousarray(v)).double() if v.dtype.kind == "f" else torch.from_numpy(np.ascontiguousarray(v))
             for k, v in seeded.fill_state(shapes, 5).items()}
    mixed, lips = seeded.inputs(5, 2, F_, 6, 3, 8, 8)
    x, l = torch.from_numpy(mixed).double(), torch.from_numpy(lips).double()
    torch_cpu.RELU_RECORD = rec = []
    try:
        _, m0 = torch_cpu.forward_train({k: v.clone() for k, v in state.items()}, x, l, h, S)
    finally:
        torch_cpu.RELU_RECORD = None
    assert len(rec) == 2 + 1 + 3 + 1                     # input_proj.0 / .2, audio linear1, three Conv2d blocks, visual linear1
    torch_cpu.RELU_FORCE = [r.clone() for r in rec]
    try:
        _, m1 = torch_cpu.forward_train({k: v.clone() for k, v in state.items()}, x, l, h, S)
        flipped = [r.clone() for r in rec]
        flipped[2][0, 0, 0] = ~flipped[2][0, 0, 0]
        torch_cpu.RELU_FORCE = flipped
        _, m2 = torch_cpu.forward_train({k: v.clone() for k, v in state.items()}, x, l, h, S)
    finally:
        torch_cpu.RELU_FORCE = None
    assert torch.equal(m0, m1)
    assert not torch.equal(m0[0], m2[0]) and torch.equal(m0[1], m2[1])   # clip 0's unit flipped: only clip 0 moves
