"""CPU baseline: the reference's PyTorch CPU forward restated with torch functional ops.

TEST INFRASTRUCTURE (see oracle/numpy_forward.py header) -- used only by bench.py's `cpu_baseline` leg
(kind "port") and by tests as a second checker.  /root/reference does not exist on the GPU box, so the
reference's CPU path (eval, no_grad, fp32, torch's fused encoder fast path) is timed
there through this restatement, which issues the same ATen kernels the reference's nn.Modules dispatch
to: mkldnn convolution, addmm, native_layer_norm, native_batch_norm,
_transformer_encoder_layer_fwd (the eval fast path of nn.TransformerEncoderLayer, torch
nn/modules/transformer.py:842-921), the explicit bmm/softmax/bmm cross-attention of
F.multi_head_attention_forward with need_weights=True (torch nn/functional.py:6576-6594), exact-erf GELU,
upsample_linear1d, sigmoid.  Pinned against tests/golden/ in tests/test_oracle.py.

`forward_train` is the same restatement with autograd recording and TRAIN-mode semantics (BatchNorm batch statistics
with the running-stat update, dropout at the reference's sites: PositionalEncoding model.py:301, the encoder layers'
attention/dropout1/dropout/dropout2, fusion 159/164 + attention dropout 155, decoder 197); pinned against the
reference's own gradients (tests/golden/train_*.npz) in tests/test_oracle.py and timed as the CPU baseline of
`bench.py --mode train`.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


# When a list is assigned here, every ReLU of forward_train appends min|pre-activation|: a value within rounding of 0
# means two correct implementations may disagree on that unit's mask bit, and with it on a whole row of gradients.
RELU_PROBE = None


# When a list is assigned here, ReLU number i of a forward (call order: audio input_proj.0, input_proj.2, audio encoder linear1 per layer,
# the three Conv2d blocks, visual encoder linear1 per layer) takes its DECISIONS from entry i (a bool / 0-1 tensor of the
# pre-activation's shape; None = decide by sign as usual): relu(x) := x * mask.  The kink-aware gradient gate of
# tests/test_train_gpu.py forces the decisions the HIP step made, so that what is compared is the derivative of the SAME piecewise-
# linear branch of the function -- a pre-activation within rounding of 0 may fall on either side in two correct implementations.
RELU_FORCE = None
RELU_RECORD = None      # a list: every ReLU appends the decisions it made by sign (x > 0)
_relu_calls = [0]


def _relu(x):
    i = _relu_calls[0]
    _relu_calls[0] = i + 1
    if RELU_PROBE is not None:
        RELU_PROBE.append(float(x.detach().abs().min()))
    if RELU_RECORD is not None:
        RELU_RECORD.append(x.detach() > 0)
    if RELU_FORCE is not None and i < len(RELU_FORCE) and RELU_FORCE[i] is not None:
        m = RELU_FORCE[i]
        assert m.shape == x.shape, (i, tuple(m.shape), tuple(x.shape))
        return x * m.to(x.dtype)
    return F.relu(x)


def _to_torch(state, dtype=torch.float32):
    out = {}
    for k, v in state.items():
        t = v if isinstance(v, torch.Tensor) else torch.from_numpy(__import__("numpy").ascontiguousarray(v))
        out[k] = t.to(dtype) if t.is_floating_point() else t
    return out


def _count(state, prefix):
    n = 0
    while f"{prefix}{n}.norm1.weight" in state:
        n += 1
    return n


def _encoder_layer(x, W, p, nhead, fast, drop=0.0):
    """pre-norm TransformerEncoderLayer, relu, ff=4d  (model.py:48-52); `drop` > 0 = train-mode dropout."""
    if fast and drop == 0.0 and not torch.is_grad_enabled() and hasattr(torch, "_transformer_encoder_layer_fwd") and x.dtype == torch.float32:
        return torch._transformer_encoder_layer_fwd(
            x, x.shape[-1], nhead, W[p + "self_attn.in_proj_weight"], W[p + "self_attn.in_proj_bias"],
            W[p + "self_attn.out_proj.weight"], W[p + "self_attn.out_proj.bias"], False, True, 1e-5,
            W[p + "norm1.weight"], W[p + "norm1.bias"], W[p + "norm2.weight"], W[p + "norm2.bias"],
            W[p + "linear1.weight"], W[p + "linear1.bias"], W[p + "linear2.weight"], W[p + "linear2.bias"],
            None, None)
    d = x.shape[-1]
    n = F.layer_norm(x, (d,), W[p + "norm1.weight"], W[p + "norm1.bias"], 1e-5)
    a = _mha(n, n, W[p + "self_attn.in_proj_weight"], W[p + "self_attn.in_proj_bias"],
             W[p + "self_attn.out_proj.weight"], W[p + "self_attn.out_proj.bias"], nhead, drop)
    x = x + F.dropout(a, drop, drop > 0)
    n = F.layer_norm(x, (d,), W[p + "norm2.weight"], W[p + "norm2.bias"], 1e-5)
    f = F.dropout(_relu(F.linear(n, W[p + "linear1.weight"], W[p + "linear1.bias"])), drop, drop > 0)
    return x + F.dropout(F.linear(f, W[p + "linear2.weight"], W[p + "linear2.bias"]), drop, drop > 0)


def _mha(q_in, kv_in, w_in, b_in, w_out, b_out, nhead, drop=0.0):
    B, Lq, d = q_in.shape
    Lk = kv_in.shape[1]
    dh = d // nhead
    q = F.linear(q_in, w_in[:d], b_in[:d]) * (1.0 / math.sqrt(dh))
    k = F.linear(kv_in, w_in[d:2 * d], b_in[d:2 * d])
    v = F.linear(kv_in, w_in[2 * d:], b_in[2 * d:])
    q = q.view(B, Lq, nhead, dh).transpose(1, 2).reshape(B * nhead, Lq, dh)
    k = k.view(B, Lk, nhead, dh).transpose(1, 2).reshape(B * nhead, Lk, dh)
    v = v.view(B, Lk, nhead, dh).transpose(1, 2).reshape(B * nhead, Lk, dh)
    p = F.dropout(torch.softmax(torch.bmm(q, k.transpose(1, 2)), dim=-1), drop, drop > 0)
    o = torch.bmm(p, v).view(B, nhead, Lq, dh).transpose(1, 2).reshape(B, Lq, d)
    return F.linear(o, w_out, b_out)


@torch.no_grad()
def forward(state, mixed, lips, nhead, num_speakers, fast=True):
    """(separated, masks), logical (B,S,F,T) like AVSeparationTransformer.forward (model.py:268-276) in eval mode.
    `state`: reference state_dict (torch tensors or numpy arrays); pe buffers optional."""
    W = state if all(isinstance(v, torch.Tensor) for v in state.values()) else _to_torch(state, mixed.dtype)
    return _forward(W, mixed, lips, nhead, num_speakers, fast, False, 0.0)


def forward_train(state, mixed, lips, nhead, num_speakers, dropout=0.0):
    """Train-mode forward with autograd recording: `state` must hold torch tensors (parameters with requires_grad as
    the caller wishes; BatchNorm running statistics are updated in place, num_batches_tracked incremented)."""
    return _forward(state, mixed, lips, nhead, num_speakers, False, True, float(dropout))


def _forward(W, mixed, lips, nhead, num_speakers, fast, train, drop):
    B, Fq, T = mixed.shape
    _relu_calls[0] = 0
    # ---- AudioEncoder (model.py:54-60)
    h = _relu(F.conv1d(mixed, W["audio_encoder.input_proj.0.weight"], W["audio_encoder.input_proj.0.bias"], padding=1))
    h = _relu(F.conv1d(h, W["audio_encoder.input_proj.2.weight"], W["audio_encoder.input_proj.2.bias"], padding=1))
    a = h.permute(0, 2, 1)
    d = a.shape[-1]
    a = F.dropout(a + _pe(W, "audio_encoder.pos_enc.pe", T, d, a.dtype), drop, drop > 0)
    for i in range(_count(W, "audio_encoder.transformer.layers.")):
        a = _encoder_layer(a, W, f"audio_encoder.transformer.layers.{i}.", nhead, fast, drop)
    # ---- VisualEncoder (model.py:103-117)
    _, N, H, Wd = lips.shape
    x = lips.reshape(B * N, 1, H, Wd)
    for ci, bi in ((0, 1), (3, 4), (6, 7)):
        c, b = f"visual_encoder.conv.{ci}.", f"visual_encoder.conv.{bi}."
        x = F.conv2d(x, W[c + "weight"], W[c + "bias"], stride=2, padding=1)
        x = F.batch_norm(x, W[b + "running_mean"], W[b + "running_var"], W[b + "weight"], W[b + "bias"], train, 0.1, 1e-5)
        if train and b + "num_batches_tracked" in W:
            W[b + "num_batches_tracked"] += 1
        x = _relu(x)
    v = F.adaptive_avg_pool2d(x, 1).flatten(1)
    v = F.linear(v, W["visual_encoder.frame_proj.weight"], W["visual_encoder.frame_proj.bias"]).view(B, N, d)
    v = F.dropout(v + _pe(W, "visual_encoder.pos_enc.pe", N, d, v.dtype), drop, drop > 0)
    for i in range(_count(W, "visual_encoder.transformer.layers.")):
        v = _encoder_layer(v, W, f"visual_encoder.transformer.layers.{i}.", nhead, fast, drop)
    v = F.interpolate(v.permute(0, 2, 1), size=T, mode="linear", align_corners=False).permute(0, 2, 1)
    # ---- CrossModalFusion (model.py:145-173)
    for i in range(_count(W, "fusion.layers.")):
        p = f"fusion.layers.{i}."
        n = F.layer_norm(a, (d,), W[p + "norm1.weight"], W[p + "norm1.bias"], 1e-5)
        a = a + F.dropout(_mha(n, v, W[p + "cross_attn.in_proj_weight"], W[p + "cross_attn.in_proj_bias"],
                               W[p + "cross_attn.out_proj.weight"], W[p + "cross_attn.out_proj.bias"], nhead, drop),
                          drop, drop > 0)
        n = F.layer_norm(a, (d,), W[p + "norm2.weight"], W[p + "norm2.bias"], 1e-5)
        f = F.dropout(F.gelu(F.linear(n, W[p + "ff.0.weight"], W[p + "ff.0.bias"])), drop, drop > 0)
        a = a + F.dropout(F.linear(f, W[p + "ff.3.weight"], W[p + "ff.3.bias"]), drop, drop > 0)
    a = F.layer_norm(a, (d,), W["fusion.norm.weight"], W["fusion.norm.bias"], 1e-5)
    # ---- SeparationDecoder (model.py:201-220)
    z = F.linear(F.dropout(F.gelu(F.linear(a, W["decoder.decoder.0.weight"], W["decoder.decoder.0.bias"])), drop, drop > 0),
                 W["decoder.decoder.3.weight"], W["decoder.decoder.3.bias"])
    masks = torch.sigmoid(z.view(B, T, num_speakers, Fq).permute(0, 2, 3, 1))
    return masks * mixed.unsqueeze(1), masks


def _pe(W, key, L, d, dtype):
    if key in W:
        return W[key][:, :L].to(dtype)
    pos = torch.arange(0, L).unsqueeze(1).float()
    div = torch.exp(torch.arange(0, d, 2).float() * (-math.log(10000.0) / d))
    pe = torch.zeros(L, d)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.unsqueeze(0).to(dtype)
