"""Host-side mirror of the reference's module API for the forward path, backed by libavsep_hip.so.

Same public names, constructor keywords/defaults, call signatures and ``state_dict`` keys/shapes as
``/root/reference/src/av_separation/model.py`` (class -> reference lines):

    AudioEncoder            model.py:22-60       VisualEncoder       model.py:67-117
    CrossModalFusion        model.py:124-149     SeparationDecoder   model.py:180-220
    AVSeparationTransformer model.py:227-276     PositionalEncoding  model.py:283-301

so ``load_state_dict`` of a reference checkpoint works unchanged.  The modules only own
parameters (torch tensors on the ROCm device = device-memory plumbing); all arithmetic of ``forward`` is
done by the hand-written HIP kernels behind the C ABI in ``include/avsep.h``.  There is no eager-PyTorch or
CPU fallback: calling a module that is not on a ROCm device, or without the built library, raises.

The parameter tree is built from a key/shape/initialiser table instead of composing torch.nn layers; the
initialisers reproduce the distributions torch.nn gives the reference's layers (kaiming-uniform(a=sqrt 5)
for Conv/Linear, xavier-uniform for MultiheadAttention.in_proj_weight, zeros for its biases).

Training: ``AVSeparationTransformer.forward`` in ``.train()`` mode runs the op-by-op
HIP training path of ``_train.py`` (autograd wrappers whose forward and backward are HIP kernels, BatchNorm batch
statistics, stateless-mask dropout); the stand-alone stage modules do the same for their stage, and an eval-mode
module called while autograd is recording takes the same differentiable path with eval semantics.
"""
from __future__ import annotations

import ctypes as C
import math
import warnings

import torch
import torch.nn as nn

from . import _native

_MAX_LEN = 5000


# --------------------------------------------------------------------------------------- parameter table
def _encoder_layer_entries(prefix, d):
    """nn.TransformerEncoderLayer(d, nhead, 4d, norm_first=True) parameters (model.py:48-52)."""
    return [
        (prefix + "self_attn.in_proj_weight", (3 * d, d), "xavier"),
        (prefix + "self_attn.in_proj_bias", (3 * d,), "zeros"),
        (prefix + "self_attn.out_proj.weight", (d, d), "linear_w"),
        (prefix + "self_attn.out_proj.bias", (d,), "zeros"),
        (prefix + "linear1.weight", (4 * d, d), "linear_w"),
        (prefix + "linear1.bias", (4 * d,), ("linear_b", d)),
        (prefix + "linear2.weight", (d, 4 * d), "linear_w"),
        (prefix + "linear2.bias", (d,), ("linear_b", 4 * d)),
        (prefix + "norm1.weight", (d,), "ones"), (prefix + "norm1.bias", (d,), "zeros"),
        (prefix + "norm2.weight", (d,), "ones"), (prefix + "norm2.bias", (d,), "zeros"),
    ]


def _audio_entries(F, d, layers):
    e = [("input_proj.0.weight", (d, F, 3), "linear_w"), ("input_proj.0.bias", (d,), ("linear_b", 3 * F)),
         ("input_proj.2.weight", (d, d, 3), "linear_w"), ("input_proj.2.bias", (d,), ("linear_b", 3 * d)),
         ("pos_enc.pe", (1, _MAX_LEN, d), "pe")]
    for i in range(layers):
        e += _encoder_layer_entries(f"transformer.layers.{i}.", d)
    return e


def _visual_entries(d, layers):
    e, cin = [], 1
    for conv_i, cout in ((0, 32), (3, 64), (6, 128)):
        bn = conv_i + 1
        e += [(f"conv.{conv_i}.weight", (cout, cin, 3, 3), "linear_w"),
              (f"conv.{conv_i}.bias", (cout,), ("linear_b", 9 * cin)),
              (f"conv.{bn}.weight", (cout,), "ones"), (f"conv.{bn}.bias", (cout,), "zeros"),
              (f"conv.{bn}.running_mean", (cout,), "buf_zeros"), (f"conv.{bn}.running_var", (cout,), "buf_ones"),
              (f"conv.{bn}.num_batches_tracked", (), "buf_long")]
        cin = cout
    e += [("frame_proj.weight", (d, 128), "linear_w"), ("frame_proj.bias", (d,), ("linear_b", 128)),
          ("pos_enc.pe", (1, _MAX_LEN, d), "pe")]
    for i in range(layers):
        e += _encoder_layer_entries(f"transformer.layers.{i}.", d)
    return e


def _fusion_entries(d, layers):
    e = []
    for i in range(layers):
        p = f"layers.{i}."
        e += [(p + "cross_attn.in_proj_weight", (3 * d, d), "xavier"),
              (p + "cross_attn.in_proj_bias", (3 * d,), "zeros"),
              (p + "cross_attn.out_proj.weight", (d, d), "linear_w"),
              (p + "cross_attn.out_proj.bias", (d,), "zeros"),
              (p + "ff.0.weight", (4 * d, d), "linear_w"), (p + "ff.0.bias", (4 * d,), ("linear_b", d)),
              (p + "ff.3.weight", (d, 4 * d), "linear_w"), (p + "ff.3.bias", (d,), ("linear_b", 4 * d)),
              (p + "norm1.weight", (d,), "ones"), (p + "norm1.bias", (d,), "zeros"),
              (p + "norm2.weight", (d,), "ones"), (p + "norm2.bias", (d,), "zeros")]
    e += [("norm.weight", (d,), "ones"), ("norm.bias", (d,), "zeros")]
    return e


def _decoder_entries(d, F, S):
    return [("decoder.0.weight", (2 * d, d), "linear_w"), ("decoder.0.bias", (2 * d,), ("linear_b", d)),
            ("decoder.3.weight", (F * S, 2 * d), "linear_w"), ("decoder.3.bias", (F * S,), ("linear_b", 2 * d))]


def _sinusoid_table(d_model, max_len=_MAX_LEN):
    """pe[p,2i] = sin(p * exp(-2i ln(1e4)/d)), pe[p,2i+1] = cos(same); no sqrt(d) scaling (model.py:289-297)."""
    pos = torch.arange(0, max_len).unsqueeze(1).float()
    freq = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    table = torch.zeros(max_len, d_model)
    table[:, 0::2] = torch.sin(pos * freq)
    table[:, 1::2] = torch.cos(pos * freq)
    return table.unsqueeze(0)


# Every structural change of a module tree built here (a parameter / buffer / sub-module assigned, registered, deleted,
# or moved by .to() / .float() / ..., which REPLACES buffer tensors) bumps this counter; the engines cache their flat
# (key, tensor) lists against it, so the per-call "did a weight change?" check is a 15 us walk over cached tensors
# (data_ptr + version) instead of a 120 us state_dict() traversal -- a third of the host cost of a graph-replayed step.
_STRUCT_EPOCH = [0]


class _Tracked(nn.Module):
    def __setattr__(self, name, value):
        if isinstance(value, (torch.Tensor, nn.Module)) or name in self.__dict__.get("_parameters", ()) or \
                name in self.__dict__.get("_buffers", ()) or name in self.__dict__.get("_modules", ()):
            _STRUCT_EPOCH[0] += 1
        super().__setattr__(name, value)

    def __delattr__(self, name):
        _STRUCT_EPOCH[0] += 1
        super().__delattr__(name)

    def register_parameter(self, name, param):
        _STRUCT_EPOCH[0] += 1
        super().register_parameter(name, param)

    def register_buffer(self, name, tensor, persistent=True):
        _STRUCT_EPOCH[0] += 1
        super().register_buffer(name, tensor, persistent=persistent)

    def add_module(self, name, module):
        _STRUCT_EPOCH[0] += 1
        super().add_module(name, module)

    def _apply(self, fn, *args, **kwargs):
        _STRUCT_EPOCH[0] += 1
        return super()._apply(fn, *args, **kwargs)

    def _load_from_state_dict(self, *args, **kwargs):
        _STRUCT_EPOCH[0] += 1          # copies in place (version bump) -- belt and braces for assign=True loads
        return super()._load_from_state_dict(*args, **kwargs)

    def train(self, mode: bool = True):
        # A train() <-> eval() transition re-packs the weights on the next fused forward.  The per-call change check
        # compares (data_ptr, _version) of every tensor, and writes through ``.data`` (EMA, clamps, hand-rolled
        # loaders: ``p.data.mul_()``, ``p.data.copy_()``) do not bump ``_version`` -- the usual place for such writes
        # is a training loop, which ends with ``.eval()``.
        if bool(mode) != self.training:
            _STRUCT_EPOCH[0] += 1
        return super().train(mode)

    def invalidate_weights(self):
        """Force the next fused forward to re-pack the weights.  Needed only after in-place writes through ``.data``
        (or raw pointers) while the module stays in eval mode: those bypass autograd's version counter, which is what
        the per-call change check reads.  ``optimizer.step()``, ``load_state_dict``, ``.to()`` and in-place ops under
        ``torch.no_grad()`` are detected without it."""
        _STRUCT_EPOCH[0] += 1
        return self


class _Node(_Tracked):
    """Bare container: only there so dotted state_dict keys match the reference's module tree."""


def _attach(root: nn.Module, entries):
    for key, shape, init in entries:
        *path, leaf = key.split(".")
        node = root
        for name in path:
            if name not in node._modules:
                node.add_module(name, _Node())
            node = node._modules[name]
        if init == "pe":
            node.register_buffer(leaf, _sinusoid_table(shape[2], shape[1]))
            continue
        if init == "buf_zeros":
            node.register_buffer(leaf, torch.zeros(shape))
            continue
        if init == "buf_ones":
            node.register_buffer(leaf, torch.ones(shape))
            continue
        if init == "buf_long":
            node.register_buffer(leaf, torch.tensor(0, dtype=torch.long))
            continue
        t = torch.empty(shape)
        if init == "ones":
            nn.init.ones_(t)
        elif init == "zeros":
            nn.init.zeros_(t)
        elif init == "xavier":
            nn.init.xavier_uniform_(t)
        elif init == "linear_w":      # kaiming_uniform_(a=sqrt(5))  ==  U(+-1/sqrt(fan_in))
            bound = 1.0 / math.sqrt(math.prod(shape[1:]))
            nn.init.uniform_(t, -bound, bound)
        elif isinstance(init, tuple) and init[0] == "linear_b":
            bound = 1.0 / math.sqrt(init[1])
            nn.init.uniform_(t, -bound, bound)
        else:  # pragma: no cover
            raise AssertionError(init)
        node.register_parameter(leaf, nn.Parameter(t))


# --------------------------------------------------------------------------------------- native engine
class _Engine:
    """One avsep_ctx + workspace per (module, device).  Re-packs weights when any tensor changed."""

    def __init__(self, owner: nn.Module, prefix: str, F, d, h, Le, Lf, S):
        self._owner_ref = owner
        self.prefix = prefix
        self.cfg = (int(F), int(d), int(h), int(Le), int(Lf), int(S))
        self.ctx = None
        self.device = None
        self.sig = None
        self.ws = None
        self.taps = False
        self.schedule = (0, 8, 0.0)   # avsep_set_schedule arguments, re-applied to every native context this engine creates
        self.split_precision = True   # avsep_set_split_precision, likewise
        self.static = None     # graph-replay buffers
        self._flat, self._flat_epoch = None, -1

    # -- lifetime
    def _create(self, device):
        self.close()
        lib = _native.load()
        cfg = _native.AvsepConfig(*self.cfg)
        ctx = C.c_void_p()
        with torch.cuda.device(device):
            _native.check(lib.avsep_create(C.byref(cfg), C.byref(ctx)), "avsep_create")
        self.ctx, self.device, self.sig, self.ws = ctx, device, None, None
        if self.taps:
            _native.check(lib.avsep_set_debug_taps(self.ctx, 1))
        if self.schedule[0]:
            _native.check(lib.avsep_set_schedule(self.ctx, int(self.schedule[0]), int(self.schedule[1]), float(self.schedule[2])),
                          "avsep_set_schedule")
        if not self.split_precision:
            _native.check(lib.avsep_set_split_precision(self.ctx, 0), "avsep_set_split_precision")

    def close(self):
        if self.ctx is not None:
            _native.load().avsep_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __getstate__(self):   # copy.deepcopy / pickle of a module: the copy gets a fresh native context
        state = self.__dict__.copy()
        state.update(ctx=None, device=None, sig=None, ws=None, static=None, _flat=None, _flat_epoch=-1)
        return state

    # -- weights
    def _tensors(self):
        if self._flat is None or self._flat_epoch != _STRUCT_EPOCH[0]:
            owner = self._owner_ref
            self._flat = [(self.prefix + k, v) for k, v in owner.state_dict(keep_vars=True).items()
                          if not k.endswith("num_batches_tracked")]
            self._flat_epoch = _STRUCT_EPOCH[0]
        return self._flat

    def sync_weights(self, device, stream):
        tensors = self._tensors()
        sig = (self._flat_epoch,) + tuple((v.data_ptr(), v._version) for _, v in tensors)
        if self.ctx is None or self.device != device:
            self._create(device)
        if sig == self.sig:
            return
        lib = _native.load()
        keep = []
        for k, v in tensors:
            if v.device != device:
                raise RuntimeError(f"parameter {k} is on {v.device}, input is on {device}")
            t = v.detach()
            if t.dtype != torch.float32 or not t.is_contiguous():
                t = t.float().contiguous()
            keep.append(t)
            shape = (C.c_int64 * max(1, t.dim()))(*t.shape)
            _native.check(lib.avsep_set_weight(self.ctx, k.encode(), t.data_ptr(), shape, t.dim()),
                          f"avsep_set_weight({k})")
        _native.check(lib.avsep_finalize_weights(self.ctx, stream), "avsep_finalize_weights")
        # the packer reads `keep` asynchronously on `stream`; temporaries stay alive until it is done
        if any(t.data_ptr() != v.data_ptr() for t, (_, v) in zip(keep, tensors)):
            torch.cuda.current_stream(device).synchronize()
        self.sig = sig

    def workspace(self, B, T, N, H, W, device):
        lib = _native.load()
        need = int(lib.avsep_workspace_bytes(self.ctx, B, T, N, H, W))
        if need == 0:
            raise RuntimeError("avsep_workspace_bytes: invalid sizes")
        if self.ws is None or self.ws.numel() < need or self.ws.device != device:
            self.ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self.ws, need

    def set_schedule(self, schedule: int, group: int = 8, skew: float = 0.0):
        if schedule and not hasattr(_native.load(), "avsep_set_schedule"):
            raise RuntimeError("launch schedules 1 / 2 (chained encoder layers) exist in the developer build of the library only "
                               "(AVSEP_LIB=dev): they were measured slower than the launch-per-op schedule")
        self.schedule = (int(schedule), int(group), float(skew))
        if self.ctx is not None and hasattr(_native.load(), "avsep_set_schedule"):
            _native.check(_native.load().avsep_set_schedule(self.ctx, *self.schedule[:2], C.c_float(self.schedule[2])),
                          "avsep_set_schedule")

    def set_split_precision(self, on: bool):
        self.split_precision = bool(on)
        if self.ctx is not None:
            _native.check(_native.load().avsep_set_split_precision(self.ctx, int(self.split_precision)), "avsep_set_split_precision")

    def set_taps(self, on: bool):
        self.taps = bool(on)
        if self.ctx is not None:
            _native.check(_native.load().avsep_set_debug_taps(self.ctx, int(self.taps)))
            self.ws = None


def _prep(x: torch.Tensor, what: str, ndim: int) -> torch.Tensor:
    if not isinstance(x, torch.Tensor) or x.dim() != ndim:
        raise RuntimeError(f"{what}: expected a {ndim}-D tensor, got {tuple(getattr(x, 'shape', ()))}")
    if x.device.type != "cuda":
        raise RuntimeError(
            f"{what} is on {x.device}: the MI355X path needs tensors and module on a ROCm device "
            "(`.to('cuda')`); there is no CPU fallback in this package")
    if x.dtype != torch.float32:
        x = x.float()
    return x.contiguous()             # both differentiable: an input that requires grad keeps its graph


_warned_grad = False


def _wants_autograd(module: nn.Module, *inputs) -> bool:
    """The reference's modules are plain torch.nn: in train mode they apply dropout and BatchNorm batch statistics,
    and whenever autograd is recording they build a graph (tests/test_model.py:90-97, 116-122, 210-217 backprop
    through freshly constructed modules).  Mirror that: the op-by-op HIP autograd path (``_train.py``) runs when the
    module is in train mode, or when grad mode is on and a parameter or an input requires grad; the fused inference
    path (one C-ABI call, hipGraph-able) runs otherwise -- i.e. under ``torch.no_grad()`` as demo.py:42 does."""
    global _warned_grad
    if module.training:
        return True
    if not torch.is_grad_enabled():
        return False
    if any(isinstance(t, torch.Tensor) and t.requires_grad for t in inputs) or \
            any(p.requires_grad for p in module.parameters()):
        if not _warned_grad:
            _warned_grad = True
            warnings.warn("av_separation (MI355X): autograd is recording, so this eval-mode call runs the op-by-op "
                          "differentiable path; wrap inference in torch.no_grad() for the fused forward",
                          stacklevel=3)
        return True
    return False


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


# --------------------------------------------------------------------------------------- public modules
class PositionalEncoding(_Tracked):
    """x + pe[:, :L] (model.py:283-301).  Stand-alone it is pure tensor plumbing (one broadcast add); inside
    the encoders the add is fused into the producing GEMM's epilogue."""

    def __init__(self, d_model: int, dropout: float = 0.1, max_len: int = 5000):
        super().__init__()
        self.p = float(dropout)
        self.register_buffer("pe", _sinusoid_table(d_model, max_len))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.size(1) > self.pe.size(1):
            raise RuntimeError(f"sequence length {x.size(1)} exceeds max_len {self.pe.size(1)}")
        y = x + self.pe[:, :x.size(1)]
        if self.training and self.p > 0:
            y = torch.nn.functional.dropout(y, self.p, True)
        return y


class AudioEncoder(_Tracked):
    """(B, freq_bins, T) -> (B, T, d_model): Conv1d-ReLU-Conv1d-ReLU, PE, pre-norm encoder layers."""

    def __init__(self, freq_bins: int = 257, d_model: int = 256, nhead: int = 4,
                 num_layers: int = 2, dropout: float = 0.1):
        super().__init__()
        self.freq_bins, self.d_model, self.nhead, self.num_layers = freq_bins, d_model, nhead, num_layers
        self.dropout_p = float(dropout)
        _attach(self, _audio_entries(freq_bins, d_model, num_layers))
        self._engine = _Engine(self, "audio_encoder.", freq_bins, d_model, nhead, num_layers, 0, 1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = _prep(x, "mixed_spec", 3)
        B, F, T = x.shape
        if F != self.freq_bins:
            raise RuntimeError(f"expected input with {self.freq_bins} channels (freq_bins), got {F}")
        if T > _MAX_LEN:
            raise RuntimeError(f"sequence length {T} exceeds PositionalEncoding max_len {_MAX_LEN}")
        if _wants_autograd(self, x):
            from ._train import audio_encoder_forward
            return audio_encoder_forward(self, x)
        eng, dev = self._engine, x.device
        with torch.cuda.device(dev):
            st = _stream(dev)
            eng.sync_weights(dev, st)
            ws, nbytes = eng.workspace(B, T, 1, 1, 1, dev)
            out = torch.empty(B, T, self.d_model, device=dev)
            _native.check(_native.load().avsep_audio_encoder(eng.ctx, x.data_ptr(), out.data_ptr(), ws.data_ptr(),
                                                             nbytes, B, T, st), "avsep_audio_encoder")
        return out


class VisualEncoder(_Tracked):
    """(B, N, H, W) lip frames -> (B, target_len, d_model)."""

    def __init__(self, d_model: int = 256, nhead: int = 4, num_layers: int = 2, dropout: float = 0.1):
        super().__init__()
        self.d_model, self.nhead, self.num_layers = d_model, nhead, num_layers
        self.dropout_p = float(dropout)
        _attach(self, _visual_entries(d_model, num_layers))
        self._engine = _Engine(self, "visual_encoder.", 1, d_model, nhead, num_layers, 0, 1)

    def forward(self, frames: torch.Tensor, target_len: int) -> torch.Tensor:
        frames = _prep(frames, "lip_frames", 4)
        B, N, H, W = frames.shape
        T = int(target_len)
        if T <= 0:
            raise RuntimeError("target_len must be positive")
        if N > _MAX_LEN:
            raise RuntimeError(f"sequence length {N} exceeds PositionalEncoding max_len {_MAX_LEN}")
        if _wants_autograd(self, frames):
            from ._train import visual_encoder_forward
            return visual_encoder_forward(self, frames, T)
        eng, dev = self._engine, frames.device
        with torch.cuda.device(dev):
            st = _stream(dev)
            eng.sync_weights(dev, st)
            ws, nbytes = eng.workspace(B, T, N, H, W, dev)
            out = torch.empty(B, T, self.d_model, device=dev)
            _native.check(_native.load().avsep_visual_encoder(eng.ctx, frames.data_ptr(), out.data_ptr(),
                                                              ws.data_ptr(), nbytes, B, N, H, W, T, st),
                          "avsep_visual_encoder")
        return out


class CrossModalFusion(_Tracked):
    """audio (B,T,d) queries visual (B,T,d) keys/values -> (B,T,d)."""

    def __init__(self, d_model: int = 256, nhead: int = 4, num_layers: int = 2, dropout: float = 0.1):
        super().__init__()
        self.d_model, self.nhead, self.num_layers = d_model, nhead, num_layers
        self.dropout_p = float(dropout)
        _attach(self, _fusion_entries(d_model, num_layers))
        self._engine = _Engine(self, "fusion.", 1, d_model, nhead, 0, num_layers, 1)

    def forward(self, audio: torch.Tensor, visual: torch.Tensor) -> torch.Tensor:
        audio = _prep(audio, "audio", 3)
        visual = _prep(visual, "visual", 3)
        if audio.shape != visual.shape or audio.shape[2] != self.d_model:
            raise RuntimeError(f"audio {tuple(audio.shape)} / visual {tuple(visual.shape)} must both be (B,T,{self.d_model})")
        if _wants_autograd(self, audio, visual):
            from ._train import fusion_forward
            return fusion_forward(self, audio, visual)
        B, T, _ = audio.shape
        eng, dev = self._engine, audio.device
        with torch.cuda.device(dev):
            st = _stream(dev)
            eng.sync_weights(dev, st)
            ws, nbytes = eng.workspace(B, T, 1, 1, 1, dev)
            out = torch.empty(B, T, self.d_model, device=dev)
            _native.check(_native.load().avsep_fusion(eng.ctx, audio.data_ptr(), visual.data_ptr(), out.data_ptr(),
                                                      ws.data_ptr(), nbytes, B, T, st), "avsep_fusion")
        return out


class SeparationDecoder(_Tracked):
    """fused (B,T,d) -> masks (B,S,F,T) in [0,1]; ``separate`` applies them to the mixture."""

    def __init__(self, d_model: int = 256, freq_bins: int = 257, num_speakers: int = 2, dropout: float = 0.1):
        super().__init__()
        self.d_model, self.freq_bins, self.num_speakers = d_model, freq_bins, num_speakers
        self.dropout_p = float(dropout)
        _attach(self, _decoder_entries(d_model, freq_bins, num_speakers))
        # nhead is irrelevant to this stage; d_model/32 heads keep the ctx's head-dim check happy
        self._engine = _Engine(self, "decoder.", freq_bins, d_model, max(1, d_model // 32), 0, 0, num_speakers)

    def forward(self, fused: torch.Tensor) -> torch.Tensor:
        fused = _prep(fused, "fused", 3)
        B, T, d = fused.shape
        if d != self.d_model:
            raise RuntimeError(f"expected last dim {self.d_model}, got {d}")
        if _wants_autograd(self, fused):
            from ._train import decoder_forward
            return decoder_forward(self, fused)
        eng, dev = self._engine, fused.device
        with torch.cuda.device(dev):
            st = _stream(dev)
            eng.sync_weights(dev, st)
            ws, nbytes = eng.workspace(B, T, 1, 1, 1, dev)
            masks = torch.empty(B, T, self.num_speakers, self.freq_bins, device=dev)
            _native.check(_native.load().avsep_decoder(eng.ctx, fused.data_ptr(), None, masks.data_ptr(), None,
                                                       ws.data_ptr(), nbytes, B, T, st), "avsep_decoder")
        return masks.permute(0, 2, 3, 1)      # view with the reference's strides (S*F*T, F, 1, S*F)

    def separate(self, masks: torch.Tensor, mixed_spec: torch.Tensor) -> torch.Tensor:
        # stand-alone use is one broadcast multiply (tensor plumbing); inside AVSeparationTransformer the
        # product is fused into the mask GEMM's epilogue
        return masks * mixed_spec.unsqueeze(1)


class AVSeparationTransformer(_Tracked):
    """``model(mixed_spec (B,F,T), lip_frames (B,N,H,W)) -> (separated, masks)``, each (B,S,F,T)."""

    def __init__(self, freq_bins: int = 257, d_model: int = 256, nhead: int = 4, num_encoder_layers: int = 2,
                 num_fusion_layers: int = 2, num_speakers: int = 2, dropout: float = 0.1, *, split_precision: bool = True):
        """The reference's constructor (model.py:240-249) plus ONE keyword-only option of the MI355X path:

        split_precision (d_model >= 512 only; no effect below): True (default) runs the Linear layers and the long-sequence
        attention of the fused eval forward as split-precision products on the 16-bit matrix pipe -- fp32-equivalent results
        (every golden inside the same tolerance), 1.9x the fp32-MFMA kernels' throughput at BASELINE config 3 (11.9 k against
        5.9 k clips/s), the same bits at every batch size; False keeps the fp32 MFMA kernels, which are still ~20 % faster for
        ONE clip at a time (config 3: 0.84 ms against 1.05 ms per forward; profiles/r05_ab_small_batches.txt) -- the setting
        for a latency deployment that calls the model with 1-2 clips.  ``set_split_precision`` switches later."""
        super().__init__()
        self.freq_bins, self.d_model, self.nhead = freq_bins, d_model, nhead
        self.num_speakers = num_speakers
        self.audio_encoder = AudioEncoder(freq_bins=freq_bins, d_model=d_model, nhead=nhead,
                                          num_layers=num_encoder_layers, dropout=dropout)
        self.visual_encoder = VisualEncoder(d_model=d_model, nhead=nhead, num_layers=num_encoder_layers,
                                            dropout=dropout)
        self.fusion = CrossModalFusion(d_model=d_model, nhead=nhead, num_layers=num_fusion_layers, dropout=dropout)
        self.decoder = SeparationDecoder(d_model=d_model, freq_bins=freq_bins, num_speakers=num_speakers,
                                         dropout=dropout)
        self._engine = _Engine(self, "", freq_bins, d_model, nhead, num_encoder_layers, num_fusion_layers,
                               num_speakers)
        self._engine.split_precision = bool(split_precision)
        self._graph = False

    # -- options of the MI355X path (not part of the reference API)
    def enable_graph_replay(self, on: bool = True):
        """Replay the whole forward from one hipGraph (launch-bound at small batch).  Inputs are copied into
        static buffers and the returned tensors are views of static output buffers that the NEXT call
        overwrites -- the usual device-graph contract."""
        self._graph = bool(on)
        self._engine.static = None
        return self

    def enable_debug_taps(self, on: bool = True):
        self._engine.set_taps(on)
        return self

    def set_schedule(self, schedule: int = 0, group: int = 8, skew: float = 0.0):
        """Launch schedule of the fused eval forward (include/avsep.h avsep_set_schedule, DEVELOPER build of the library only):
        0 = one launch per op (default), 1 / 2 = the encoder layers of each branch as one dependency-driven persistent launch
        (one queue / XCD-local queues); same output bits either way, measured slower."""
        self._engine.set_schedule(schedule, group, skew)
        for eng in self.__dict__.get("_slot_engines", {}).values():
            eng.set_schedule(schedule, group, skew)
        return self

    def set_split_precision(self, on: bool = True):
        """The split-precision GEMM / attention kernels of the fused eval forward on (default) or off (include/avsep.h
        avsep_set_split_precision; the constructor's ``split_precision`` keyword sets the initial value).  They only ever apply to
        d_model >= 512 models and are fp32-equivalent; they win from 2-4 clips per forward on and are ~20 % slower than the fp32
        MFMA kernels for ONE clip at a time; the library never looks at the batch size by itself (same bits at every batch size
        under either setting): a latency deployment of a d_model >= 512 model passes ``split_precision=False``."""
        self._engine.set_split_precision(on)
        for eng in self.__dict__.get("_slot_engines", {}).values():
            eng.set_split_precision(on)
        return self

    def chain_status(self):
        """Raises if a chained launch (schedule 1) of the last forward gave up waiting for a producer tile."""
        eng = self._engine
        _native.check(_native.load().avsep_chain_status(eng.ctx, _stream(eng.device)), "avsep_chain_status")

    def read_tap(self, name: str, shape):
        eng = self._engine
        B, T, N, H, W = eng.last_dims
        out = torch.empty(*shape, device=eng.device)
        n = _native.load().avsep_read_tap(eng.ctx, name.encode(), out.data_ptr(), out.numel(), eng.ws.data_ptr(),
                                          B, T, N, H, W, _stream(eng.device))
        _native.check(n, f"avsep_read_tap({name})")
        if n != out.numel():
            raise RuntimeError(f"tap {name}: {n} floats, expected {out.numel()}")
        return out

    def run_static(self, mixed: torch.Tensor, lips: torch.Tensor, masks_btsf: torch.Tensor, sep_btsf: torch.Tensor,
                   graph: bool = False, slot: int = 0):
        """Lowest-level call: caller-owned contiguous float32 device buffers in, (B,T,S,F) buffers out, nothing
        allocated or copied here.  With ``graph=True`` the launch sequence is replayed from a hipGraph keyed on
        these exact buffers (bench.py's timed loop).  ``slot``: forwards that are in flight at the same time (different
        streams, different input / output buffers) must use different slots; every slot > 0 is a native context of its
        own (own packed-weight arena, workspace and graphs) -- measured: two graphs of ONE context replayed on two
        streams do not overlap (0.466 ms per 32-clip step), two contexts do (0.375)."""
        B, F, T = mixed.shape
        _, N, H, W = lips.shape
        eng, dev = self._engine, mixed.device
        if slot:
            extra = self.__dict__.setdefault("_slot_engines", {})
            eng = extra.get(slot)
            if eng is None:
                eng = extra[slot] = _Engine(self, "", *self._engine.cfg)
                eng.schedule = self._engine.schedule
                eng.split_precision = self._engine.split_precision
        lib = _native.load()
        with torch.cuda.device(dev):
            st = _stream(dev)
            eng.sync_weights(dev, st)
            ws, nbytes = eng.workspace(B, T, N, H, W, dev)
            eng.last_dims = (B, T, N, H, W)
            fn = lib.avsep_forward_graph if graph else lib.avsep_forward
            _native.check(fn(eng.ctx, mixed.data_ptr(), lips.data_ptr(), masks_btsf.data_ptr(), sep_btsf.data_ptr(),
                             ws.data_ptr(), nbytes, B, T, N, H, W, st),
                          "avsep_forward_graph" if graph else "avsep_forward")

    def profile_begin(self):
        _native.check(_native.load().avsep_profile_begin(self._engine.ctx), "avsep_profile_begin")

    def profile_end(self):
        """-> list of {"name","calls","ms","flops","bytes"} per kernel instance (HIP-event timed)."""
        import json
        buf = C.create_string_buffer(1 << 16)
        n = _native.load().avsep_profile_end(self._engine.ctx, buf, len(buf))
        _native.check(n, "avsep_profile_end")
        return json.loads(buf.value.decode())

    def forward(self, mixed_spec: torch.Tensor, lip_frames: torch.Tensor):
        mixed = _prep(mixed_spec, "mixed_spec", 3)
        lips = _prep(lip_frames, "lip_frames", 4)
        B, F, T = mixed.shape
        if F != self.freq_bins:
            raise RuntimeError(f"expected input with {self.freq_bins} channels (freq_bins), got {F}")
        if lips.shape[0] != B:
            raise RuntimeError(f"batch mismatch: mixed_spec {B}, lip_frames {lips.shape[0]}")
        if T > _MAX_LEN or lips.shape[1] > _MAX_LEN:
            raise RuntimeError(f"sequence length exceeds PositionalEncoding max_len {_MAX_LEN}")
        if _wants_autograd(self, mixed, lips) and (self.training or not self._engine.taps):
            # training / autograd path: same HIP library, op by op (av_separation/_train.py)
            from ._train import train_forward
            return train_forward(self, mixed, lips)
        S, dev = self.num_speakers, mixed.device
        eng = self._engine
        if self._graph and not eng.taps:
            key = (tuple(mixed.shape), tuple(lips.shape), dev)
            if eng.static is None or eng.static[0] != key:
                eng.static = (key, torch.empty_like(mixed), torch.empty_like(lips),
                              torch.empty(B, T, S, F, device=dev), torch.empty(B, T, S, F, device=dev))
            _, s_mixed, s_lips, masks, sep = eng.static
            s_mixed.copy_(mixed)
            s_lips.copy_(lips)
            self.run_static(s_mixed, s_lips, masks, sep, graph=True)
        else:
            masks = torch.empty(B, T, S, F, device=dev)
            sep = torch.empty(B, T, S, F, device=dev)
            self.run_static(mixed, lips, masks, sep, graph=False)
        # logical (B,S,F,T) views over (B,T,S,F) memory: the reference's output strides
        return sep.permute(0, 2, 3, 1), masks.permute(0, 2, 3, 1)
