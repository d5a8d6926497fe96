"""Op-level parity of the TRAINING ops (include/avsep.h "training ops") at production-like shapes against plain
PyTorch fp32 autograd on the same device: the end-to-end gradient goldens (test_train_gpu.py) only reach tiny
shapes, these reach the split-K weight gradient, ragged column reductions, long-sequence attention backward and
the 300k-row BatchNorm of the conv layers.  Tolerances are relative to the largest reference entry."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(got, ref):
    return float((got - ref).abs().max()) / max(1e-6, float(ref.abs().max()))


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("M,N,K,act,res", [
    (4016, 512, 512, 0, False), (4016, 2048, 512, 1, False), (2016, 256, 1024, 0, True),
    (19200, 64, 288, 0, False), (307200, 32, 32, 0, False), (1200, 512, 128, 0, False), (37, 771, 1024, 0, False),
])
def test_linear_fn_forward_backward(dev, M, N, K, act, res):
    """LinearFn: y, dX = dY W, dW = dY^T X (split over rows when N*K is small), db -- vs torch."""
    from av_separation import _train as tr
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(dev).requires_grad_()
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev).requires_grad_()
    b = torch.randn(N, generator=g).to(dev).requires_grad_()
    r = torch.randn(M, N, generator=g).to(dev).requires_grad_() if res else None
    dy = torch.randn(M, N, generator=g).to(dev)
    y = tr.LinearFn.apply(x, w, b, tr.ACT_RELU if act else tr.ACT_NONE, r, 0)
    y.backward(dy)
    got = [y.detach(), x.grad, w.grad, b.grad] + ([r.grad] if res else [])
    x2, w2, b2 = (t.detach().clone().requires_grad_() for t in (x, w, b))
    r2 = r.detach().clone().requires_grad_() if res else None
    y2 = F.linear(x2, w2, b2)
    if act:     # the ReLU decisions of the op under test (a pre-activation within rounding of 0 may fall on either side in two correct
        y2 = y2 * (y.detach() > 0)   # GEMMs -- the split-precision forward GEMM is the default); the VALUES are compared below
        assert float((torch.relu(F.linear(x2, w2, b2)).detach() - y.detach()).abs().max()) < 2e-5 * float(y.detach().abs().max())
    if res:
        y2 = y2 + r2
    y2.backward(dy)
    ref = [y2.detach(), x2.grad, w2.grad, b2.grad] + ([r2.grad] if res else [])
    for name, a_, b_ in zip(("y", "dx", "dw", "db", "dres"), got, ref):
        assert _rel(a_, b_) < 2e-5, (name, _rel(a_, b_))


@pytest.mark.parametrize("M,C", [(307200, 32), (4016, 2048), (4016, 771), (5, 7), (1, 1), (70000, 128), (257, 300)])
def test_column_reductions(dev, M, C):
    from av_separation import _train as tr
    g = torch.Generator(device="cpu").manual_seed(M * 7 + C)
    a = torch.randn(M, C, generator=g).to(dev)
    b = torch.randn(M, C, generator=g).to(dev)
    s0, _ = tr._colsum(a)
    t0, t1 = tr._colsum(a, b)
    ref0 = a.double().sum(0)
    ref1 = (a.double() * b.double()).sum(0)
    scale = math.sqrt(M)
    assert float((s0.double() - ref0).abs().max()) < 2e-5 * scale
    assert torch.equal(s0, t0)                                     # same stage-1 partition either way
    assert float((t1.double() - ref1).abs().max()) < 2e-5 * scale
    again, _ = tr._colsum(a)
    assert torch.equal(s0, again)                                  # deterministic (no atomics)


@pytest.mark.parametrize("M,C", [(307200, 32), (76800, 64), (19200, 128), (40, 32)])
def test_batchnorm_relu_fn(dev, M, C):
    """BatchNormReluFn (batch statistics, running-stat update, backward) vs F.batch_norm(training=True)+relu."""
    from av_separation import _train as tr
    g = torch.Generator(device="cpu").manual_seed(M + C)
    x = (torch.randn(M, C, generator=g) * 2 + 0.5).to(dev)
    gam = (torch.rand(C, generator=g) + 0.5).to(dev).requires_grad_()
    bet = torch.randn(C, generator=g).to(dev).requires_grad_()
    # keep every pre-activation away from the ReLU kink: with 10^7 elements one of them lands within rounding of 0,
    # and the two implementations may then legitimately disagree on its mask bit (an O(1) error at that element)
    for _ in range(2):
        xd = x.double()
        pre = (xd - xd.mean(0)) / torch.sqrt(xd.var(0, unbiased=False) + 1e-5) * gam.detach().double() + bet.detach().double()
        x = torch.where(pre.abs() < 1e-3, x + 0.02 * torch.sign(pre).float() + 0.02 * (pre == 0).float(), x)
    x.requires_grad_()
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    dy = torch.randn(M, C, generator=g).to(dev)
    y = tr.BatchNormReluFn.apply(x, gam, bet, rm, rv, 1e-5, 0.1)
    y.backward(dy)
    x2, g2, b2 = (t.detach().clone().requires_grad_() for t in (x, gam, bet))
    rm2, rv2 = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    y2 = torch.relu(F.batch_norm(x2, rm2, rv2, g2, b2, True, 0.1, 1e-5))
    y2.backward(dy)
    for name, a_, b_ in (("y", y.detach(), y2.detach()), ("dx", x.grad, x2.grad), ("dgamma", gam.grad, g2.grad),
                         ("dbeta", bet.grad, b2.grad), ("running_mean", rm, rm2), ("running_var", rv, rv2)):
        assert _rel(a_, b_) < 5e-5, (name, _rel(a_, b_))


@pytest.mark.parametrize("B,h,dh,Lq,Lk,cross", [(2, 8, 64, 251, 251, False), (1, 8, 64, 501, 501, True),
                                                 (3, 2, 32, 19, 50, True), (2, 4, 16, 32, 32, False), (1, 4, 16, 1, 1, False)])
def test_attention_fn_forward_backward(dev, B, h, dh, Lq, Lk, cross):
    """AttentionFn on the packed in_proj layout vs torch scaled_dot_product_attention autograd."""
    from av_separation import _train as tr
    d = h * dh
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + Lq + Lk)
    if cross:
        q = torch.randn(B * Lq, d, generator=g).to(dev).requires_grad_()
        kv = torch.randn(B * Lk, 2 * d, generator=g).to(dev).requires_grad_()
        o = tr.AttentionFn.apply(q, kv, 0, 0, d, B, h, dh, Lq, Lk, 1.0 / math.sqrt(dh))
    else:
        qkv = torch.randn(B * Lq, 3 * d, generator=g).to(dev).requires_grad_()
        o = tr.AttentionFn.apply(qkv, qkv, 0, d, 2 * d, B, h, dh, Lq, Lk, 1.0 / math.sqrt(dh))
    do = torch.randn(B * Lq, d, generator=g).to(dev)
    o.backward(do)

    def heads(t, L):
        return t.reshape(B, L, h, dh).permute(0, 2, 1, 3)

    if cross:
        q2, kv2 = q.detach().clone().requires_grad_(), kv.detach().clone().requires_grad_()
        qq, kk, vv = heads(q2, Lq), heads(kv2[:, :d], Lk), heads(kv2[:, d:], Lk)
    else:
        qkv2 = qkv.detach().clone().requires_grad_()
        qq, kk, vv = heads(qkv2[:, :d], Lq), heads(qkv2[:, d:2 * d], Lk), heads(qkv2[:, 2 * d:], Lk)
    p = torch.softmax((qq @ kk.transpose(-1, -2)) / math.sqrt(dh), dim=-1)
    o2 = (p @ vv).permute(0, 2, 1, 3).reshape(B * Lq, d)
    o2.backward(do)
    assert _rel(o.detach(), o2.detach()) < 2e-5
    if cross:
        assert _rel(q.grad, q2.grad) < 5e-5 and _rel(kv.grad, kv2.grad) < 5e-5
    else:
        assert _rel(qkv.grad, qkv2.grad) < 5e-5


@pytest.mark.parametrize("M,d", [(4016, 512), (2016, 256), (70, 96), (3, 64)])
def test_layernorm_fn(dev, M, d):
    from av_separation import _train as tr
    g = torch.Generator(device="cpu").manual_seed(M + d)
    x = (torch.randn(M, d, generator=g) * 3 + 1).to(dev).requires_grad_()
    gam = (torch.rand(d, generator=g) + 0.5).to(dev).requires_grad_()
    bet = torch.randn(d, generator=g).to(dev).requires_grad_()
    dy = torch.randn(M, d, generator=g).to(dev)
    y = tr.LayerNormFn.apply(x, gam, bet, 1e-5)
    y.backward(dy)
    x2, g2, b2 = (t.detach().clone().requires_grad_() for t in (x, gam, bet))
    y2 = F.layer_norm(x2, (d,), g2, b2, 1e-5)
    y2.backward(dy)
    for name, a_, b_ in (("y", y.detach(), y2.detach()), ("dx", x.grad, x2.grad), ("dg", gam.grad, g2.grad),
                         ("db", bet.grad, b2.grad)):
        assert _rel(a_, b_) < 3e-5, (name, _rel(a_, b_))


def test_conv_layers_as_im2col_gemm(dev):
    """Conv1d(k3,p1) and Conv2d(k3,s2,p1) forward/backward through Im2col*Fn + LinearFn vs torch convolutions
    (the layout conventions of train_forward: channels-last rows, weights permuted tap-major)."""
    from av_separation import _train as tr
    g = torch.Generator(device="cpu").manual_seed(3)
    B, T, Ci, Co = 3, 37, 64, 96
    x = torch.randn(B * T, Ci, generator=g).to(dev).requires_grad_()
    w = (torch.randn(Co, Ci, 3, generator=g) * 0.1).to(dev).requires_grad_()
    b = torch.randn(Co, generator=g).to(dev).requires_grad_()
    dy = torch.randn(B * T, Co, generator=g).to(dev)
    y = tr.LinearFn.apply(tr.Im2col1dFn.apply(x, T), w.permute(0, 2, 1).reshape(Co, 3 * Ci), b, tr.ACT_NONE, None, 0)
    y.backward(dy)
    x2, w2, b2 = (t.detach().clone().requires_grad_() for t in (x, w, b))
    y2 = F.conv1d(x2.reshape(B, T, Ci).permute(0, 2, 1), w2, b2, padding=1).permute(0, 2, 1).reshape(B * T, Co)
    y2.backward(dy)
    for a_, b_ in ((y.detach(), y2.detach()), (x.grad, x2.grad), (w.grad, w2.grad), (b.grad, b2.grad)):
        assert _rel(a_, b_) < 3e-5
    I, H, W, Ci, Co = 5, 15, 13, 32, 64
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    x = torch.randn(I * H * W, Ci, generator=g).to(dev).requires_grad_()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.1).to(dev).requires_grad_()
    b = torch.randn(Co, generator=g).to(dev).requires_grad_()
    dy = torch.randn(I * Ho * Wo, Co, generator=g).to(dev)
    y = tr.LinearFn.apply(tr.Im2col2dFn.apply(x, I, H, W, 9 * Ci), w.permute(0, 2, 3, 1).reshape(Co, 9 * Ci), b,
                          tr.ACT_NONE, None, 0)
    y.backward(dy)
    x2, w2, b2 = (t.detach().clone().requires_grad_() for t in (x, w, b))
    y2 = F.conv2d(x2.reshape(I, H, W, Ci).permute(0, 3, 1, 2), w2, b2, stride=2, padding=1)
    y2 = y2.permute(0, 2, 3, 1).reshape(I * Ho * Wo, Co)
    y2.backward(dy)
    for a_, b_ in ((y.detach(), y2.detach()), (x.grad, x2.grad), (w.grad, w2.grad), (b.grad, b2.grad)):
        assert _rel(a_, b_) < 3e-5


def test_attention_backward_rejects_unsupported_head_dim(dev):
    """The backward kernels tile the head dimension in 16-wide MFMA blocks: dh % 16 != 0 (legal for the forward, which
    takes any dh % 4 == 0) must fail loudly, not silently compute something else."""
    from av_separation import _train as tr
    qkv = torch.randn(4, 3 * 32, device=dev, requires_grad=True)
    o = tr.AttentionFn.apply(qkv, qkv, 0, 32, 64, 1, 4, 8, 4, 4, 1.0 / math.sqrt(8))
    with pytest.raises(RuntimeError, match="multiple of 16"):
        o.sum().backward()


@pytest.mark.parametrize("B,N,T,d", [(16, 75, 251, 512), (2, 50, 63, 256), (3, 12, 5, 32), (2, 1, 7, 64), (2, 9, 1, 64),
                                     (1, 50, 501, 512), (4, 10, 10, 32)])
def test_interp_fn_backward_is_the_adjoint_of_f_interpolate(dev, B, N, T, d):
    """InterpFn backward (gather formulation) against torch autograd of F.interpolate(mode="linear",
    align_corners=False) on the HOST (model.py:114-116), up- and down-sampling, single-row edge cases."""
    from av_separation import _train as tr
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + N * 10 + T)
    x = torch.randn(B * N, d, generator=g)
    dy = torch.randn(B * T, d, generator=g)
    xd = x.to(dev).requires_grad_()
    y = tr.InterpFn.apply(xd, B, N, T)
    y.backward(dy.to(dev))
    xr = x.clone().requires_grad_()
    yr = F.interpolate(xr.view(B, N, d).permute(0, 2, 1), size=T, mode="linear", align_corners=False).permute(0, 2, 1)
    yr.reshape(B * T, d).backward(dy)
    assert _rel(y.detach().cpu(), yr.detach().reshape(B * T, d)) < 2e-6
    assert _rel(xd.grad.cpu(), xr.grad) < 3e-6


@pytest.mark.parametrize("R,N,K", [(4016, 512, 512), (4016, 1536, 512), (4016, 2048, 512), (4016, 512, 2048), (3990, 516, 1028),
                                   (8032, 1536, 512), (100, 2048, 2048), (33, 512, 512), (1200, 64, 288)])
def test_wgrad_direct_split_against_float64(dev, R, N, K):
    """The split-precision weight gradient (csrc/wgrad_split.hip: both operands cut into three bf16 terms on their way to LDS,
    transposed by the LDS read, six bf16 MFMA products per fp32 product) against float64 at the fp32 kernel's error level -- at
    most 2x its measured error + 2e-7 on the same operands -- with ragged row counts / tile edges, sliced and unsliced row
    ranges, with and without the bias gradient, which must be the fp32 op's bit for bit.  (1200, 64, 288) takes the fp32
    plan's 32-wide tile: the op then runs the fp32 kernel itself.)"""
    import ctypes as C
    from av_separation import _native
    lib = _native.load()
    g = torch.Generator(device="cpu").manual_seed(R + N + K)
    dy = torch.randn(R, N, generator=g).to(dev)
    x = (torch.randn(R, K, generator=g) * 1.5 + 0.2).to(dev)
    ref_w = (dy.double().T @ x.double())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ns = lib.avsep_op_wgrad_bias_direct_scratch_floats(N, K, R)
    scratch = torch.empty(max(ns, 1), device=dev)
    out0 = torch.full((N * K + N,), float("nan"), device=dev)
    assert lib.avsep_op_wgrad_bias_direct(dy.data_ptr(), N, x.data_ptr(), K, out0.data_ptr(), scratch.data_ptr() if ns else None,
                                          N, K, R, st) == 0, lib.avsep_last_error()
    sc = float(ref_w.abs().max())
    e0 = float((out0[:N * K].view(N, K).double() - ref_w).abs().max()) / sc
    for with_bias in (1, 0):
        out1 = torch.full((N * K + N,), float("nan"), device=dev)
        assert lib.avsep_op_wgrad_direct_split(dy.data_ptr(), N, x.data_ptr(), K, out1.data_ptr(), scratch.data_ptr() if ns else None,
                                               N, K, R, with_bias, st) == 0, lib.avsep_last_error()
        gw = out1[:N * K].view(N, K)
        assert torch.isfinite(gw).all()
        e1 = float((gw.double() - ref_w).abs().max()) / sc
        assert e1 < 2.0 * e0 + 2e-7 and e1 < 3e-7 * math.sqrt(R) + 1e-6, (with_bias, e0, e1)
        if with_bias:
            assert torch.equal(out1[N * K:], out0[N * K:])
        else:
            assert torch.isnan(out1[N * K:]).all()                  # nothing written behind the weight gradients


@pytest.mark.parametrize("R,N,K", [(4016, 512, 512), (4016, 1536, 512), (4016, 2048, 512), (4016, 512, 2048),
                                   (4016, 512, 128), (1200, 64, 288), (37, 772, 1024)])
def test_wgrad_bias_direct_against_float64(dev, R, N, K):
    """dW = dY^T X and db = column sums of dY from ONE launch (avsep_op_wgrad_bias_direct: the bias gradient rides the
    weight-gradient kernel) at the config-4 training shapes, against float64 -- the library's own tile / slice choice, and
    through the developer build every combination of the 32- and 64-wide tile with an unsliced and a sliced row range
    (the slices are summed in a fixed order: all variants of one tile width agree bit for bit with each other's sums only
    up to the slice boundaries, so each is held to the float64 result)."""
    import ctypes as C
    import os
    from av_separation import _native
    g = torch.Generator(device="cpu").manual_seed(R + N + K)
    dy = torch.randn(R, N, generator=g).to(dev)
    x = torch.randn(R, K, generator=g).to(dev)
    ref_w = (dy.double().T @ x.double())
    ref_b = dy.double().sum(0)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(lib):
        out = torch.full((N * K + N,), float("nan"), device=dev)
        ns = lib.avsep_op_wgrad_bias_direct_scratch_floats(N, K, R)
        scratch = torch.empty(max(ns, 1), device=dev)
        rc = lib.avsep_op_wgrad_bias_direct(dy.data_ptr(), N, x.data_ptr(), K, out.data_ptr(), scratch.data_ptr() if ns else None,
                                            N, K, R, st)
        assert rc == 0, lib.avsep_last_error()
        return out[:N * K].view(N, K), out[N * K:]

    variants = {"product": run(_native.load())}
    dev_lib = _native.load_dev()
    try:
        for tile in ("32", "64"):
            for slices in ("1", "4"):
                os.environ["AVSEP_WGRAD_TILE"], os.environ["AVSEP_WGRAD_SLICES"] = tile, slices
                variants[f"tile{tile}/slices{slices}"] = run(dev_lib)
    finally:
        os.environ.pop("AVSEP_WGRAD_TILE", None)
        os.environ.pop("AVSEP_WGRAD_SLICES", None)
    # the in-launch slice merge of the developer build (the tile's last-arriving workgroup sums the slices; measured slower,
    # not in the product): the SAME bits as the two-launch form, every time (a race between a slice's stores and the
    # reducer's loads would show as a run-to-run difference)
    lib = dev_lib
    ns = lib.avsep_op_wgrad_bias_direct_scratch_floats(N, K, R)
    cnt = torch.zeros(max(int(lib.avsep_op_wgrad_tiles(N, K, R)), 1), dtype=torch.int32, device=dev)
    for with_bias in (1, 0):
        for _ in range(6):
            out = torch.full((N * K + N,), float("nan"), device=dev)
            scratch = torch.full((max(ns, 1),), float("nan"), device=dev)
            rc = lib.avsep_op_wgrad_merged(dy.data_ptr(), N, x.data_ptr(), K, out.data_ptr(), scratch.data_ptr() if ns else None,
                                           cnt.data_ptr(), N, K, R, with_bias, st)
            assert rc == 0, lib.avsep_last_error()
            assert torch.equal(out[:N * K].view(N, K), variants["product"][0])
            if with_bias:
                assert torch.equal(out[N * K:], variants["product"][1])
            assert int(cnt.abs().sum()) == 0                    # every counter is back at zero
    tol = 3e-7 * math.sqrt(R)                              # fp32 accumulation over R rows, relative to the largest entry
    for name, (gw, gb) in variants.items():
        assert torch.isfinite(gw).all() and torch.isfinite(gb).all(), name
        assert _rel(gw.double(), ref_w) < tol, (name, _rel(gw.double(), ref_w))
        assert _rel(gb.double(), ref_b) < tol, (name, _rel(gb.double(), ref_b))


@pytest.mark.parametrize("I,H,W,C,Kp", [(1200, 32, 32, 1, 32), (1200, 16, 16, 32, 288), (1200, 8, 8, 64, 576),
                                        (7, 13, 9, 3, 32), (5, 48, 48, 1, 32), (3, 24, 24, 32, 288), (2, 5, 7, 8, 96)])
def test_im2col2d_and_its_adjoint_against_unfold(dev, I, H, W, C, Kp):
    """im2col2d (Conv2d k3 s2 p1 as a GEMM on a channels-last image, K = 9 C zero-padded to Kp) and col2im2d, its adjoint,
    against F.unfold / F.fold -- exactly: both only move (and, in the adjoint, add up to four) floats.  Round 3 replaced the
    per-element integer divisions by multiply-highs and moves four channels per thread; the odd shapes take the one-channel
    instance."""
    import ctypes as C_
    from av_separation import _native
    lib = _native.load()
    st = C_.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cpu").manual_seed(I + H + W + C)
    x = torch.randn(I, H, W, C, generator=g).to(dev)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    col = torch.full((I * Ho * Wo, Kp), float("nan"), device=dev)
    _native.check(lib.avsep_op_im2col2d(x.data_ptr(), col.data_ptr(), I, H, W, C, Kp, st))
    unf = F.unfold(x.permute(0, 3, 1, 2), kernel_size=3, padding=1, stride=2)            # (I, C*9, L), channel-major
    ref = unf.view(I, C, 9, Ho * Wo).permute(0, 3, 2, 1).reshape(I * Ho * Wo, 9 * C)     # row (img, y, x), column tap*C + c
    assert torch.equal(col[:, :9 * C], ref)
    assert (col[:, 9 * C:] == 0).all()
    dcol = torch.randn(I * Ho * Wo, Kp, generator=g).to(dev)
    dx = torch.full((I, H, W, C), float("nan"), device=dev)
    _native.check(lib.avsep_op_col2im2d(dcol.data_ptr(), dx.data_ptr(), I, H, W, C, Kp, st))
    cols = dcol[:, :9 * C].view(I, Ho * Wo, 9, C).permute(0, 3, 2, 1).reshape(I, C * 9, Ho * Wo)
    fold = F.fold(cols, output_size=(H, W), kernel_size=3, padding=1, stride=2).permute(0, 2, 3, 1)
    assert float((dx - fold).abs().max()) <= 4e-6 * float(fold.abs().max())            # <= 4 terms, any order
