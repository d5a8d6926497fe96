"""Host-side mirror of the reference API (no GPU needed): names, constructor signatures, state_dict
keys/shapes/order, load_state_dict of a reference-format checkpoint, and LOUD failure when the HIP path
cannot run (CPU tensors, train mode) -- mirrors the structural checks of the reference's
tests/test_model.py:58-230."""
import copy
import inspect

import numpy as np
import pytest
import torch

import av_separation
from av_separation.model import (AudioEncoder, VisualEncoder, CrossModalFusion, SeparationDecoder,
                                 AVSeparationTransformer, PositionalEncoding)
from oracle import seeded
from helpers import golden_state


def test_public_names_match_reference_init():
    assert av_separation.__all__ == ["AudioEncoder", "VisualEncoder", "CrossModalFusion", "SeparationDecoder",
                                     "AVSeparationTransformer", "SyntheticAVDataset"]


def test_constructor_signatures():
    def sig(cls):   # the reference's parameters: everything that can be passed positionally
        return [(p.name, p.default) for p in inspect.signature(cls.__init__).parameters.values()
                if p.kind is not inspect.Parameter.KEYWORD_ONLY][1:]
    assert sig(AVSeparationTransformer) == [("freq_bins", 257), ("d_model", 256), ("nhead", 4),
                                            ("num_encoder_layers", 2), ("num_fusion_layers", 2),
                                            ("num_speakers", 2), ("dropout", 0.1)]
    # the one option of the MI355X path is keyword-only and defaults to the fast setting (a constructor argument, not a
    # method the caller has to discover)
    extra = [(p.name, p.default) for p in inspect.signature(AVSeparationTransformer.__init__).parameters.values()
             if p.kind is inspect.Parameter.KEYWORD_ONLY]
    assert extra == [("split_precision", True)]
    assert AVSeparationTransformer(split_precision=False)._engine.split_precision is False
    assert sig(AudioEncoder) == [("freq_bins", 257), ("d_model", 256), ("nhead", 4), ("num_layers", 2), ("dropout", 0.1)]
    assert sig(VisualEncoder) == [("d_model", 256), ("nhead", 4), ("num_layers", 2), ("dropout", 0.1)]
    assert sig(CrossModalFusion) == [("d_model", 256), ("nhead", 4), ("num_layers", 2), ("dropout", 0.1)]
    assert sig(SeparationDecoder) == [("d_model", 256), ("freq_bins", 257), ("num_speakers", 2), ("dropout", 0.1)]
    assert sig(PositionalEncoding) == [("d_model", inspect._empty), ("dropout", 0.1), ("max_len", 5000)]


@pytest.mark.parametrize("kw", [dict(), dict(freq_bins=65, d_model=64, nhead=4, num_encoder_layers=1,
                                             num_fusion_layers=3, num_speakers=3)])
def test_state_dict_keys_and_shapes(kw):
    m = AVSeparationTransformer(**kw)
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    want = seeded.model_shapes(kw.get("freq_bins", 257), kw.get("d_model", 256), kw.get("nhead", 4),
                               kw.get("num_encoder_layers", 2), kw.get("num_fusion_layers", 2),
                               kw.get("num_speakers", 2))
    assert got == want
    assert m.state_dict()["visual_encoder.conv.1.num_batches_tracked"].dtype == torch.long


def test_parameter_count_matches_reference():
    # 5,654,978 parameters at d=256; README.md:60: 1,612,738 at d=128
    assert sum(p.numel() for p in AVSeparationTransformer().parameters()) == 5_654_978
    assert sum(p.numel() for p in AVSeparationTransformer(d_model=128).parameters()) == 1_612_738


def test_state_dict_order_is_module_order():
    keys = list(AVSeparationTransformer(num_encoder_layers=1, num_fusion_layers=1).state_dict())
    assert keys[:5] == ["audio_encoder.input_proj.0.weight", "audio_encoder.input_proj.0.bias",
                        "audio_encoder.input_proj.2.weight", "audio_encoder.input_proj.2.bias",
                        "audio_encoder.pos_enc.pe"]
    assert keys[5] == "audio_encoder.transformer.layers.0.self_attn.in_proj_weight"
    assert keys[-4:] == ["decoder.decoder.0.weight", "decoder.decoder.0.bias", "decoder.decoder.3.weight",
                         "decoder.decoder.3.bias"]


def test_load_reference_format_checkpoint(golden):
    g = golden("trained_tiny")
    c = g["config"]
    m = AVSeparationTransformer(c["F"], c["d"], c["h"], c["Le"], c["Lf"], c["S"])
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in golden_state(g).items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert unexpected == [] and sorted(missing) == ["audio_encoder.pos_enc.pe", "visual_encoder.pos_enc.pe"]
    assert torch.equal(m.fusion.norm.weight, sd["fusion.norm.weight"])
    # the module's own pe equals the reference's pe rows stored with the fixture (same torch ops)
    assert np.array_equal(m.audio_encoder.pos_enc.pe[0, :c["T"]].numpy(), g["pe"])


def test_submodule_attributes_and_deepcopy():
    m = AVSeparationTransformer(freq_bins=65, d_model=64)
    assert isinstance(m.audio_encoder, AudioEncoder) and isinstance(m.visual_encoder, VisualEncoder)
    assert isinstance(m.fusion, CrossModalFusion) and isinstance(m.decoder, SeparationDecoder)
    m2 = copy.deepcopy(m)
    assert torch.equal(m2.decoder.decoder[0].weight if hasattr(m2.decoder.decoder, "__getitem__") else
                       m2.state_dict()["decoder.decoder.0.weight"], m.state_dict()["decoder.decoder.0.weight"])


def test_default_init_distributions():
    torch.manual_seed(0)
    m = AVSeparationTransformer()
    sd = m.state_dict()
    w = sd["decoder.decoder.0.weight"]                     # Linear(256,512): U(+-1/16)
    assert abs(float(w.abs().max()) - 1 / 16) < 1e-3 and abs(float(w.std()) - (1 / 16) / 3 ** 0.5) < 2e-3
    ipw = sd["fusion.layers.0.cross_attn.in_proj_weight"]  # xavier: bound sqrt(6/(256+768))
    assert abs(float(ipw.abs().max()) - (6 / 1024) ** 0.5) < 1e-3
    assert float(sd["fusion.layers.0.cross_attn.in_proj_bias"].abs().max()) == 0
    assert float(sd["audio_encoder.transformer.layers.0.self_attn.out_proj.bias"].abs().max()) == 0
    assert torch.all(sd["visual_encoder.conv.1.running_var"] == 1) and torch.all(sd["fusion.norm.weight"] == 1)


def test_positional_encoding_module():
    pe = PositionalEncoding(64, dropout=0.0)
    out = pe(torch.zeros(2, 32, 64))
    assert out.shape == (2, 32, 64) and not torch.all(out == 0)
    assert torch.allclose(out[0, :, 0], torch.sin(torch.arange(32.0)), atol=1e-6)
    with pytest.raises(RuntimeError):
        pe(torch.zeros(1, 5001, 64))


def test_cpu_tensors_fail_loudly_no_fallback():
    m = AVSeparationTransformer(freq_bins=65, d_model=64).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(2, 65, 32), torch.zeros(2, 10, 16, 16))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.audio_encoder(torch.zeros(2, 65, 32))


def test_train_mode_never_fakes_eval_semantics():
    m = AVSeparationTransformer(freq_bins=65, d_model=64).train()
    with pytest.raises(RuntimeError, match="no CPU fallback"):      # the training path is HIP-only as well
        m(torch.zeros(2, 65, 32), torch.zeros(2, 10, 16, 16))
    with pytest.raises(RuntimeError, match="no CPU fallback"):      # ... and so are the stand-alone stages
        m.audio_encoder(torch.zeros(2, 65, 32))


def test_separate_is_broadcast_multiply():
    dec = SeparationDecoder(d_model=64, freq_bins=65, num_speakers=2)
    masks, mixed = torch.rand(2, 2, 65, 32), torch.rand(2, 65, 32)
    assert torch.equal(dec.separate(masks, mixed), masks * mixed.unsqueeze(1))
