"""Framework-independent seeded tensor generator  --  TEST INFRASTRUCTURE, not product code.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

Everything here is pure 64-bit integer arithmetic (splitmix64 + FNV-1a) followed by one exact
int->float conversion, so the same (seed, key, shape) yields bit-identical float32 tensors in the
survey container (where the reference was run to make tests/golden/*) and on the GPU box (where the
reference does not exist).  That lets the large BASELINE configs be pinned by committing only the
reference's *outputs*: the weights and inputs are regenerated from the seed.

The fill rules follow the reference's state_dict layout:
  * ``*.pe``                 -> not generated (the module computes it, model.py:289-297)
  * ``*.num_batches_tracked``-> int64 constant 3
  * ``*.running_var``        -> U(0.5, 1.5)     ``*.running_mean`` -> U(-0.2, 0.2)
  * 1-D ``*.weight``         -> 1 + U(-0.2, 0.2)   (LayerNorm / BatchNorm scale)
  * 1-D ``*.bias`` / in_proj_bias -> U(-0.1, 0.1)
  * >=2-D weights            -> U(-a, a), a = gain / sqrt(fan_in), fan_in = prod(shape[1:])
"""
from __future__ import annotations

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def fnv1a64(text: str) -> int:
    h = 0xCBF29CE484222325
    for b in text.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def splitmix64(seed: int, n: int) -> np.ndarray:
    """n outputs of the splitmix64 stream started at `seed` (vectorised: output i uses state
    seed + (i+1)*GOLDEN, exactly the sequential definition)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, n: int) -> np.ndarray:
    """float64 in [0,1) from the top 24 bits (exactly representable in float32 too)."""
    return (splitmix64(seed, n) >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))


def tensor(seed: int, key: str, shape, lo: float, hi: float) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(seed ^ fnv1a64(key), n)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def fill_state(shapes: dict, seed: int, gain: float = 1.7) -> dict:
    """shapes: {state_dict key: tuple shape}.  Returns {key: np.ndarray} for every key except *.pe."""
    out = {}
    for key in sorted(shapes):
        shape = tuple(int(s) for s in shapes[key])
        if key.endswith(".pe"):
            continue
        if key.endswith("num_batches_tracked"):
            out[key] = np.array(3, dtype=np.int64)
        elif key.endswith("running_var"):
            out[key] = tensor(seed, key, shape, 0.5, 1.5)
        elif key.endswith("running_mean"):
            out[key] = tensor(seed, key, shape, -0.2, 0.2)
        elif len(shape) == 1 and key.endswith("weight"):
            out[key] = tensor(seed, key, shape, 0.8, 1.2)
        elif len(shape) == 1:
            out[key] = tensor(seed, key, shape, -0.1, 0.1)
        else:
            a = gain / float(np.sqrt(np.prod(shape[1:])))
            out[key] = tensor(seed, key, shape, -a, a)
    return out


def inputs(seed: int, B: int, F: int, T: int, N: int, H: int, W: int):
    """Synthetic (mixed_spec, lip_frames) with the value ranges of SyntheticAVDataset
    (dataset.py:70-151: spectrogram magnitudes >= 0 with a heavy tail, lips in [0,1])."""
    u = tensor(seed, "in.mixed.u", (B, F, T), 0.0, 1.0).astype(np.float64)
    v = tensor(seed, "in.mixed.v", (B, F, T), 0.0, 1.0).astype(np.float64)
    mixed = (40.0 * u ** 6 + 0.5 * v).astype(np.float32)
    lips = tensor(seed, "in.lips", (B, N, H, W), 0.0, 1.0)
    return mixed, lips


def model_shapes(freq_bins=257, d_model=256, nhead=4, num_encoder_layers=2, num_fusion_layers=2,
                 num_speakers=2):
    """state_dict key -> shape for the reference model (measured there;
    model.py:37-52, 81-101, 139-164, 194-199).  `pe` buffers are listed but never generated."""
    d, F, S = d_model, freq_bins, num_speakers
    sh = {}

    def enc_layer(prefix):
        sh[prefix + "self_attn.in_proj_weight"] = (3 * d, d)
        sh[prefix + "self_attn.in_proj_bias"] = (3 * d,)
        sh[prefix + "self_attn.out_proj.weight"] = (d, d)
        sh[prefix + "self_attn.out_proj.bias"] = (d,)
        sh[prefix + "linear1.weight"] = (4 * d, d)
        sh[prefix + "linear1.bias"] = (4 * d,)
        sh[prefix + "linear2.weight"] = (d, 4 * d)
        sh[prefix + "linear2.bias"] = (d,)
        for n in ("norm1", "norm2"):
            sh[prefix + n + ".weight"] = (d,)
            sh[prefix + n + ".bias"] = (d,)

    sh["audio_encoder.input_proj.0.weight"] = (d, F, 3)
    sh["audio_encoder.input_proj.0.bias"] = (d,)
    sh["audio_encoder.input_proj.2.weight"] = (d, d, 3)
    sh["audio_encoder.input_proj.2.bias"] = (d,)
    sh["audio_encoder.pos_enc.pe"] = (1, 5000, d)
    for i in range(num_encoder_layers):
        enc_layer(f"audio_encoder.transformer.layers.{i}.")
    cin = 1
    for conv_i, bn_i, cout in ((0, 1, 32), (3, 4, 64), (6, 7, 128)):
        sh[f"visual_encoder.conv.{conv_i}.weight"] = (cout, cin, 3, 3)
        sh[f"visual_encoder.conv.{conv_i}.bias"] = (cout,)
        sh[f"visual_encoder.conv.{bn_i}.weight"] = (cout,)
        sh[f"visual_encoder.conv.{bn_i}.bias"] = (cout,)
        sh[f"visual_encoder.conv.{bn_i}.running_mean"] = (cout,)
        sh[f"visual_encoder.conv.{bn_i}.running_var"] = (cout,)
        sh[f"visual_encoder.conv.{bn_i}.num_batches_tracked"] = ()
        cin = cout
    sh["visual_encoder.frame_proj.weight"] = (d, 128)
    sh["visual_encoder.frame_proj.bias"] = (d,)
    sh["visual_encoder.pos_enc.pe"] = (1, 5000, d)
    for i in range(num_encoder_layers):
        enc_layer(f"visual_encoder.transformer.layers.{i}.")
    for i in range(num_fusion_layers):
        p = f"fusion.layers.{i}."
        sh[p + "cross_attn.in_proj_weight"] = (3 * d, d)
        sh[p + "cross_attn.in_proj_bias"] = (3 * d,)
        sh[p + "cross_attn.out_proj.weight"] = (d, d)
        sh[p + "cross_attn.out_proj.bias"] = (d,)
        sh[p + "ff.0.weight"] = (4 * d, d)
        sh[p + "ff.0.bias"] = (4 * d,)
        sh[p + "ff.3.weight"] = (d, 4 * d)
        sh[p + "ff.3.bias"] = (d,)
        for n in ("norm1", "norm2"):
            sh[p + n + ".weight"] = (d,)
            sh[p + n + ".bias"] = (d,)
    sh["fusion.norm.weight"] = (d,)
    sh["fusion.norm.bias"] = (d,)
    sh["decoder.decoder.0.weight"] = (2 * d, d)
    sh["decoder.decoder.0.bias"] = (2 * d,)
    sh["decoder.decoder.3.weight"] = (F * S, 2 * d)
    sh["decoder.decoder.3.bias"] = (F * S,)
    return sh


def sinusoid_pe(max_len: int, d_model: int) -> np.ndarray:
    """pe[p,2i]=sin(p*exp(-2i*ln(1e4)/d)), pe[p,2i+1]=cos(same), float32 (model.py:289-297)."""
    pos = np.arange(max_len, dtype=np.float32)[:, None]
    div = np.exp(np.arange(0, d_model, 2, dtype=np.float32) * np.float32(-np.log(10000.0) / d_model))
    pe = np.zeros((max_len, d_model), dtype=np.float32)
    pe[:, 0::2] = np.sin(pos * div)
    pe[:, 1::2] = np.cos(pos * div)
    return pe[None]
