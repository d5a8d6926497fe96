"""CPU oracle: numpy restatement of the AVSeparationTransformer forward pass.

TEST INFRASTRUCTURE -- only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import anything under oracle/.  The product path (av-separation-transformer_amd/) never does, and
fails loudly when its HIP library is missing.

Parity status: PINNED.  tests/test_oracle.py checks every function here against the golden vectors
under tests/golden/, which were produced by importing the reference (`/root/reference/src`) in the
build container with tests/golden/make_golden.py (the reference's own tests hold no numeric
goldens).

The arithmetic of the reference lives in PyTorch (torch>=2.0, requirements.txt:2; the run that made
the fixtures used torch 2.10.0 CPU kernels).  This file restates the published semantics of the
torch.nn modules the reference composes, each function citing the reference call site it follows.
It takes the reference's state_dict (numpy arrays, reference key names) and works in float32 or
float64 (`dtype`), float64 being the tie-breaker oracle of SURVEY.md §8(c).
"""
from __future__ import annotations

import math
import numpy as np

try:  # scipy is in the image; keep a slow exact fallback so the oracle never silently changes formula
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    _erf = np.vectorize(math.erf)

LN_EPS = 1e-5   # nn.LayerNorm default (model.py:143,162-163; TransformerEncoderLayer layer_norm_eps)
BN_EPS = 1e-5   # nn.BatchNorm2d default (model.py:83,86,89)


# ----------------------------------------------------------------------------- primitives
def linear(x, w, b=None):
    """y = x @ w.T + b   (nn.Linear; model.py:93,155-161,195,198)."""
    y = x @ w.T
    return y if b is None else y + b


def layer_norm(x, g, b, eps=LN_EPS):
    """nn.LayerNorm over the last dim, biased variance (model.py:143,162-163)."""
    mu = x.mean(axis=-1, keepdims=True)
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True)
    return xc / np.sqrt(var + x.dtype.type(eps)) * g + b


def relu(x):
    return np.maximum(x, 0)


def gelu_erf(x):
    """nn.GELU() default = exact erf form (model.py:158,196)."""
    t = x.dtype.type
    return t(0.5) * x * (t(1.0) + _erf(x * t(1.0 / math.sqrt(2.0))).astype(x.dtype))


def sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x))).astype(x.dtype)


def softmax_last(x):
    m = x.max(axis=-1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=-1, keepdims=True)


def conv1d_k3_p1(x, w, b):
    """nn.Conv1d(k=3, padding=1): cross-correlation, zero padding (model.py:38,40).
    x (B,Ci,T), w (Co,Ci,3) -> (B,Co,T)."""
    B, Ci, T = x.shape
    xp = np.zeros((B, Ci, T + 2), dtype=x.dtype)
    xp[:, :, 1:T + 1] = x
    # im2col: col[b,t,ci,k] = xp[b,ci,t+k]
    col = np.stack([xp[:, :, k:k + T] for k in range(3)], axis=-1)      # (B,Ci,T,3)
    col = col.transpose(0, 2, 1, 3).reshape(B, T, Ci * 3)
    y = col @ w.reshape(w.shape[0], Ci * 3).T + b                         # (B,T,Co)
    return y.transpose(0, 2, 1)


def conv2d_k3_s2_p1(x, w, b):
    """nn.Conv2d(k=3, stride=2, padding=1) (model.py:82,85,88). x (M,Ci,H,W), w (Co,Ci,3,3).
    Output size floor((H+2-3)/2)+1 = floor((H-1)/2)+1."""
    M, Ci, H, W = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    xp = np.zeros((M, Ci, H + 2, W + 2), dtype=x.dtype)
    xp[:, :, 1:H + 1, 1:W + 1] = x
    cols = []
    for ky in range(3):
        for kx in range(3):
            cols.append(xp[:, :, ky:ky + 2 * Ho:2, kx:kx + 2 * Wo:2])   # (M,Ci,Ho,Wo)
    col = np.stack(cols, axis=-1)                                         # (M,Ci,Ho,Wo,9)
    col = col.transpose(0, 2, 3, 1, 4).reshape(M * Ho * Wo, Ci * 9)
    y = col @ w.reshape(w.shape[0], Ci * 9).T + b
    return y.reshape(M, Ho, Wo, w.shape[0]).transpose(0, 3, 1, 2)


def batchnorm2d(x, g, b, mean, var, eps=BN_EPS, train=False):
    """nn.BatchNorm2d. eval: running stats (model.py:83 in .eval()); train: biased batch stats over
    (M,H,W) -- SURVEY.md §8(a) a5."""
    if train:
        mean = x.mean(axis=(0, 2, 3))
        var = x.var(axis=(0, 2, 3))
    sh = (1, -1, 1, 1)
    return (x - mean.reshape(sh)) / np.sqrt(var.reshape(sh) + x.dtype.type(eps)) * g.reshape(sh) + b.reshape(sh)


def interp_linear(x, out_len):
    """F.interpolate(mode='linear', align_corners=False) along axis 1 of (B,N,d)  (model.py:114-116).
    Index arithmetic in the tensor dtype (float32 on the product path) like ATen's area_pixel_compute_source_index:
    src = fma(N/T, i+0.5, -0.5) clamped at 0; i0=floor(src); i1=min(i0+1,N-1); w=src-i0.
    The multiply-add is FUSED in ATen's compiled CPU kernel (one rounding): emulated here in float64, where the
    float32 x float32 product is exact.  Pinned by tests/test_oracle.py::test_interp_index_is_fma."""
    B, N, d = x.shape
    ft = x.dtype.type                      # ATen does the index arithmetic in the tensor's opmath type
    scale = ft(N) / ft(out_len)
    i = np.arange(out_len).astype(x.dtype)
    src = (scale.astype(np.float64) * (i + ft(0.5)).astype(np.float64) - 0.5).astype(x.dtype)
    src = np.maximum(src, ft(0.0))
    i0 = np.floor(src).astype(np.int64)
    i0 = np.minimum(i0, N - 1)
    i1 = np.minimum(i0 + 1, N - 1)
    w1 = (src - i0.astype(x.dtype)).astype(x.dtype)[None, :, None]
    w0 = ft(1.0) - w1
    return w0 * x[:, i0, :] + w1 * x[:, i1, :]


def mha(q_in, kv_in, w_in, b_in, w_out, b_out, nhead):
    """nn.MultiheadAttention(batch_first=True) forward with packed in_proj (rows [Wq;Wk;Wv]),
    heads = contiguous dh chunks, q scaled by 1/sqrt(dh) before QK^T, softmax over keys, out_proj.
    Self-attention: model.py:48-52,97-101;  cross-attention: model.py:155,169."""
    B, Lq, d = q_in.shape
    Lk = kv_in.shape[1]
    dh = d // nhead
    q = linear(q_in, w_in[:d], b_in[:d])
    k = linear(kv_in, w_in[d:2 * d], b_in[d:2 * d])
    v = linear(kv_in, w_in[2 * d:], b_in[2 * d:])
    q = q * q.dtype.type(1.0 / math.sqrt(dh))
    q = q.reshape(B, Lq, nhead, dh).transpose(0, 2, 1, 3)
    k = k.reshape(B, Lk, nhead, dh).transpose(0, 2, 1, 3)
    v = v.reshape(B, Lk, nhead, dh).transpose(0, 2, 1, 3)
    p = softmax_last(q @ k.transpose(0, 1, 3, 2))
    o = (p @ v).transpose(0, 2, 1, 3).reshape(B, Lq, d)
    return linear(o, w_out, b_out)


# ----------------------------------------------------------------------------- model stages
def _get(state, dtype):
    return lambda k: np.asarray(state[k]).astype(dtype)


def encoder_layer(x, P, prefix, nhead):
    """nn.TransformerEncoderLayer(norm_first=True, activation=relu, ff=4d)  (model.py:48-52)."""
    n = layer_norm(x, P(prefix + "norm1.weight"), P(prefix + "norm1.bias"))
    x = x + mha(n, n, P(prefix + "self_attn.in_proj_weight"), P(prefix + "self_attn.in_proj_bias"),
                P(prefix + "self_attn.out_proj.weight"), P(prefix + "self_attn.out_proj.bias"), nhead)
    n = layer_norm(x, P(prefix + "norm2.weight"), P(prefix + "norm2.bias"))
    h = relu(linear(n, P(prefix + "linear1.weight"), P(prefix + "linear1.bias")))
    return x + linear(h, P(prefix + "linear2.weight"), P(prefix + "linear2.bias"))


def _num_layers(state, prefix):
    n = 0
    while f"{prefix}{n}.norm1.weight" in state:
        n += 1
    return n


def _pe(state, key, L, d, dtype):
    if key in state:
        return np.asarray(state[key]).astype(dtype)[0, :L]
    from .seeded import sinusoid_pe
    return sinusoid_pe(L, d).astype(dtype)[0]


def audio_encoder(state, mixed, nhead, dtype=np.float32, taps=None):
    """AudioEncoder.forward (model.py:54-60)."""
    P = _get(state, dtype)
    x = mixed.astype(dtype)
    h = relu(conv1d_k3_p1(x, P("audio_encoder.input_proj.0.weight"), P("audio_encoder.input_proj.0.bias")))
    if taps is not None:
        taps["a_conv1"] = h.transpose(0, 2, 1)
    h = relu(conv1d_k3_p1(h, P("audio_encoder.input_proj.2.weight"), P("audio_encoder.input_proj.2.bias")))
    h = h.transpose(0, 2, 1)                                     # (B,T,d)  model.py:57
    if taps is not None:
        taps["a_conv2"] = h
    T, d = h.shape[1], h.shape[2]
    h = h + _pe(state, "audio_encoder.pos_enc.pe", T, d, dtype)  # no sqrt(d) scaling, model.py:300
    if taps is not None:
        taps["a_pe"] = h
    for i in range(_num_layers(state, "audio_encoder.transformer.layers.")):
        h = encoder_layer(h, P, f"audio_encoder.transformer.layers.{i}.", nhead)
        if taps is not None:
            taps[f"a_enc{i}"] = h
    return h                                                     # no final norm (norm=None)


def visual_encoder(state, lips, target_len, nhead, dtype=np.float32, taps=None, bn_train=False):
    """VisualEncoder.forward (model.py:103-117)."""
    P = _get(state, dtype)
    B, N, H, W = lips.shape
    x = lips.astype(dtype).reshape(B * N, 1, H, W)
    for conv_i, bn_i in ((0, 1), (3, 4), (6, 7)):
        c, b = f"visual_encoder.conv.{conv_i}.", f"visual_encoder.conv.{bn_i}."
        x = conv2d_k3_s2_p1(x, P(c + "weight"), P(c + "bias"))
        x = batchnorm2d(x, P(b + "weight"), P(b + "bias"), P(b + "running_mean"), P(b + "running_var"),
                        train=bn_train)
        x = relu(x)
        if taps is not None:
            taps[f"v_conv{conv_i // 3}"] = x
    feat = x.mean(axis=(2, 3))                                   # AdaptiveAvgPool2d(1)  model.py:91
    if taps is not None:
        taps["v_pool"] = feat
    feat = linear(feat, P("visual_encoder.frame_proj.weight"), P("visual_encoder.frame_proj.bias"))
    d = feat.shape[-1]
    feat = feat.reshape(B, N, d)
    if taps is not None:
        taps["v_proj"] = feat
    feat = feat + _pe(state, "visual_encoder.pos_enc.pe", N, d, dtype)
    for i in range(_num_layers(state, "visual_encoder.transformer.layers.")):
        feat = encoder_layer(feat, P, f"visual_encoder.transformer.layers.{i}.", nhead)
        if taps is not None:
            taps[f"v_enc{i}"] = feat
    out = interp_linear(feat, target_len)
    if taps is not None:
        taps["v_interp"] = out
    return out


def fusion(state, audio, visual, nhead, dtype=np.float32, taps=None):
    """CrossModalFusion.forward + CrossAttentionLayer.forward (model.py:145-149,166-173):
    visual is NOT normalised and feeds every layer unchanged; exact-erf GELU in ff; final LayerNorm."""
    P = _get(state, dtype)
    h = audio
    for i in range(_num_layers(state, "fusion.layers.")):
        p = f"fusion.layers.{i}."
        n = layer_norm(h, P(p + "norm1.weight"), P(p + "norm1.bias"))
        h = h + mha(n, visual, P(p + "cross_attn.in_proj_weight"), P(p + "cross_attn.in_proj_bias"),
                    P(p + "cross_attn.out_proj.weight"), P(p + "cross_attn.out_proj.bias"), nhead)
        n = layer_norm(h, P(p + "norm2.weight"), P(p + "norm2.bias"))
        f = linear(gelu_erf(linear(n, P(p + "ff.0.weight"), P(p + "ff.0.bias"))),
                   P(p + "ff.3.weight"), P(p + "ff.3.bias"))
        h = h + f
        if taps is not None:
            taps[f"f_layer{i}"] = h
    h = layer_norm(h, P("fusion.norm.weight"), P("fusion.norm.bias"))
    if taps is not None:
        taps["f_norm"] = h
    return h


def decoder(state, fused, mixed, num_speakers, dtype=np.float32, taps=None):
    """SeparationDecoder.forward + .separate (model.py:201-220): output channel c = s*F + f."""
    P = _get(state, dtype)
    B, T, _ = fused.shape
    F = mixed.shape[1]
    h = gelu_erf(linear(fused, P("decoder.decoder.0.weight"), P("decoder.decoder.0.bias")))
    logits = linear(h, P("decoder.decoder.3.weight"), P("decoder.decoder.3.bias"))     # (B,T,S*F)
    if taps is not None:
        taps["d_logits"] = logits
    masks = sigmoid(logits.reshape(B, T, num_speakers, F).transpose(0, 2, 3, 1))       # (B,S,F,T)
    separated = masks * mixed.astype(dtype)[:, None]
    return separated, masks


def forward(state, mixed, lips, nhead, num_speakers, dtype=np.float32, taps=None, bn_train=False):
    """AVSeparationTransformer.forward (model.py:268-276) -> (separated, masks), each (B,S,F,T)."""
    T = mixed.shape[-1]
    a = audio_encoder(state, mixed, nhead, dtype, taps)
    v = visual_encoder(state, lips, T, nhead, dtype, taps, bn_train)
    f = fusion(state, a, v, nhead, dtype, taps)
    return decoder(state, f, mixed, num_speakers, dtype, taps)
