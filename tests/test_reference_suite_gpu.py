"""The reference's own test suite (tests/test_model.py: shapes, mask range, gradient flow, one training step, the
DataLoader batch) restated against the drop-in on the ROCm device: same constants, same module calls, same default
modes -- the reference builds its modules and calls them WITHOUT .eval(), i.e. in train mode with dropout 0.1, and
back-propagates through the stand-alone stages -- only the device differs (the reference's fixture says "cpu",
tests/test_model.py:39-41; the HIP path has no CPU fallback).  Values are pinned elsewhere (test_gpu_parity.py,
test_train_gpu.py); this file is about the drop-in behaving like the original under the original's usage."""
import pytest
import torch

pytestmark = pytest.mark.gpu

FREQ_BINS, T, D_MODEL, NHEAD, BATCH, NUM_FRAMES, H, W, NUM_SPEAKERS = 65, 32, 64, 4, 2, 10, 16, 16, 2


@pytest.fixture
def device():
    return torch.device("cuda:0")


@pytest.fixture
def audio_batch(device):
    return torch.randn(BATCH, FREQ_BINS, T, device=device)


@pytest.fixture
def visual_batch(device):
    return torch.randn(BATCH, NUM_FRAMES, H, W, device=device)


def _all_grads(module):
    for name, p in module.named_parameters():
        if p.requires_grad:
            assert p.grad is not None, f"No grad for {name}"
            assert torch.isfinite(p.grad).all(), name


class TestPositionalEncoding:                                   # tests/test_model.py:58-70
    def test_output_shape_and_adds_encoding(self, device):
        from av_separation.model import PositionalEncoding
        pe = PositionalEncoding(D_MODEL, dropout=0.0).to(device)
        out = pe(torch.zeros(BATCH, T, D_MODEL, device=device))
        assert out.shape == (BATCH, T, D_MODEL) and not torch.all(out == 0)


class TestAudioEncoder:                                         # tests/test_model.py:77-97
    def _enc(self, device):
        from av_separation.model import AudioEncoder
        return AudioEncoder(freq_bins=FREQ_BINS, d_model=D_MODEL, nhead=NHEAD, num_layers=1).to(device)

    def test_output_shape(self, audio_batch, device):
        assert self._enc(device)(audio_batch).shape == (BATCH, T, D_MODEL)

    def test_different_T(self, device):
        enc = self._enc(device)
        for t in (16, 32, 64):
            assert enc(torch.randn(BATCH, FREQ_BINS, t, device=device)).shape == (BATCH, t, D_MODEL)

    def test_gradient_flow(self, audio_batch, device):
        enc = self._enc(device)
        enc(audio_batch).sum().backward()
        _all_grads(enc)


class TestVisualEncoder:                                        # tests/test_model.py:104-122
    def _enc(self, device):
        from av_separation.model import VisualEncoder
        return VisualEncoder(d_model=D_MODEL, nhead=NHEAD, num_layers=1).to(device)

    def test_output_shape(self, visual_batch, device):
        assert self._enc(device)(visual_batch, target_len=T).shape == (BATCH, T, D_MODEL)

    def test_different_target_len(self, visual_batch, device):
        enc = self._enc(device)
        for tlen in (20, 32, 50):
            assert enc(visual_batch, target_len=tlen).shape == (BATCH, tlen, D_MODEL)

    def test_gradient_flow(self, visual_batch, device):
        enc = self._enc(device)
        enc(visual_batch, target_len=T).sum().backward()
        _all_grads(enc)


class TestCrossModalFusion:                                     # tests/test_model.py:129-148
    def test_output_shape_and_visual_dependence(self, device):
        from av_separation.model import CrossModalFusion
        fusion = CrossModalFusion(d_model=D_MODEL, nhead=NHEAD, num_layers=1).to(device)
        audio = torch.randn(BATCH, T, D_MODEL, device=device)
        v1, v2 = torch.randn(BATCH, T, D_MODEL, device=device), torch.randn(BATCH, T, D_MODEL, device=device)
        out1, out2 = fusion(audio, v1), fusion(audio, v2)
        assert out1.shape == (BATCH, T, D_MODEL)
        assert not torch.allclose(out1, out2, atol=1e-5)


class TestSeparationDecoder:                                    # tests/test_model.py:155-179
    def test_shapes_and_mask_range(self, device):
        from av_separation.model import SeparationDecoder
        dec = SeparationDecoder(d_model=D_MODEL, freq_bins=FREQ_BINS, num_speakers=NUM_SPEAKERS).to(device)
        masks = dec(torch.randn(BATCH, T, D_MODEL, device=device))
        assert masks.shape == (BATCH, NUM_SPEAKERS, FREQ_BINS, T)
        assert masks.min() >= 0.0 and masks.max() <= 1.0
        sep = dec.separate(masks, torch.randn(BATCH, FREQ_BINS, T, device=device))
        assert sep.shape == (BATCH, NUM_SPEAKERS, FREQ_BINS, T)


class TestAVSeparationTransformer:                              # tests/test_model.py:186-230
    def _build(self, device):
        from av_separation import AVSeparationTransformer
        return AVSeparationTransformer(freq_bins=FREQ_BINS, d_model=D_MODEL, nhead=NHEAD, num_encoder_layers=1,
                                       num_fusion_layers=1, num_speakers=NUM_SPEAKERS, dropout=0.0).to(device)

    def test_output_shapes_and_bounds(self, audio_batch, visual_batch, device):
        separated, masks = self._build(device)(audio_batch, visual_batch)
        assert separated.shape == masks.shape == (BATCH, NUM_SPEAKERS, FREQ_BINS, T)
        assert masks.min() >= 0.0 and masks.max() <= 1.0

    def test_backward_pass(self, audio_batch, visual_batch, device):
        model = self._build(device)
        separated, masks = model(audio_batch, visual_batch)
        (separated.sum() + masks.sum()).backward()
        _all_grads(model)

    def test_eval_mode_no_error(self, audio_batch, visual_batch, device):
        model = self._build(device).eval()
        with torch.no_grad():
            separated, _ = model(audio_batch, visual_batch)
        assert separated.shape == (BATCH, NUM_SPEAKERS, FREQ_BINS, T)

    def test_parameter_count(self, device):
        assert 10_000 < sum(p.numel() for p in self._build(device).parameters()) < 100_000_000


class TestIntegration:                                          # tests/test_model.py:332-364
    def test_one_training_step(self, device):
        from av_separation import AVSeparationTransformer
        from av_separation.losses import SeparationLoss
        model = AVSeparationTransformer(freq_bins=FREQ_BINS, d_model=D_MODEL, nhead=NHEAD, num_encoder_layers=1,
                                        num_fusion_layers=1, num_speakers=NUM_SPEAKERS, dropout=0.0).to(device)
        optimizer = torch.optim.Adam(model.parameters(), lr=1e-3)
        before = [p.detach().clone() for p in model.parameters()]
        optimizer.zero_grad()
        separated, _ = model(torch.randn(BATCH, FREQ_BINS, T, device=device),
                             torch.randn(BATCH, NUM_FRAMES, H, W, device=device))
        loss = SeparationLoss()(separated, torch.randn(BATCH, NUM_SPEAKERS, FREQ_BINS, T, device=device))
        loss.backward()
        optimizer.step()
        assert not torch.isnan(loss)
        assert any(not torch.equal(a, b.detach()) for a, b in zip(before, model.parameters()))

    def test_dataloader_batch(self, device):
        from torch.utils.data import DataLoader
        from av_separation import AVSeparationTransformer, SyntheticAVDataset
        ds = SyntheticAVDataset(num_samples=8, n_fft=256, hop_length=64, num_frames=10, frame_h=16, frame_w=16)
        batch = next(iter(DataLoader(ds, batch_size=4)))
        assert batch["mixed_spec"].shape[0] == batch["lip_frames"].shape[0] == batch["clean_specs"].shape[0] == 4
        model = AVSeparationTransformer(freq_bins=batch["mixed_spec"].shape[1], d_model=D_MODEL, nhead=NHEAD,
                                        num_encoder_layers=1, num_fusion_layers=1).to(device).eval()
        with torch.no_grad():
            sep, _ = model(batch["mixed_spec"].to(device), batch["lip_frames"].to(device))
        assert sep.shape == batch["clean_specs"].shape


# ---- beyond the reference's suite: the stage modules' autograd path agrees with the fused inference path and with
# ---- torch's own gradients of the same stage
def test_eval_autograd_path_equals_fused_inference_path(device):
    from av_separation import AVSeparationTransformer
    torch.manual_seed(4)
    model = AVSeparationTransformer(freq_bins=FREQ_BINS, d_model=D_MODEL, nhead=NHEAD, num_encoder_layers=1,
                                    num_fusion_layers=1, num_speakers=NUM_SPEAKERS).to(device).eval()
    for bn in ("visual_encoder.conv.1.", "visual_encoder.conv.4.", "visual_encoder.conv.7."):   # non-trivial statistics
        sd = model.state_dict()
        sd[bn + "running_mean"].uniform_(-0.3, 0.3)
        sd[bn + "running_var"].uniform_(0.5, 1.5)
    mixed, lips = torch.rand(BATCH, FREQ_BINS, T, device=device), torch.rand(BATCH, NUM_FRAMES, H, W, device=device)
    with torch.no_grad():
        sep0, masks0 = model(mixed, lips)
    with pytest.warns(UserWarning, match="autograd is recording"):
        sep1, masks1 = model(mixed, lips)                      # grad mode on, parameters require grad
    assert masks1.requires_grad
    assert float((masks0 - masks1.detach()).abs().max()) < 2e-6 and float((sep0 - sep1.detach()).abs().max()) < 2e-5
    (sep1.sum() + masks1.sum()).backward()
    _all_grads(model)
    stages = ((model.audio_encoder, (mixed,)), (model.visual_encoder, (lips, T)),
              (model.fusion, (torch.randn(BATCH, T, D_MODEL, device=device), torch.randn(BATCH, T, D_MODEL, device=device))),
              (model.decoder, (torch.randn(BATCH, T, D_MODEL, device=device),)))
    for mod, args in stages:
        with torch.no_grad():
            fused = mod(*args)
        diff = mod(*args)
        assert diff.requires_grad and float((fused - diff.detach()).abs().max()) < 5e-6, type(mod).__name__


def test_stage_input_gradients_match_torch(device):
    """d out / d input of the stand-alone fusion and decoder stages (what an upstream module would receive) against a
    plain-torch functional restatement of the same stage on the same weights (eval semantics, dropout off)."""
    import torch.nn.functional as F
    from av_separation.model import SeparationDecoder
    torch.manual_seed(9)
    dec = SeparationDecoder(d_model=D_MODEL, freq_bins=FREQ_BINS, num_speakers=NUM_SPEAKERS, dropout=0.0).to(device)
    x = torch.randn(BATCH, T, D_MODEL, device=device, requires_grad=True)
    w = torch.randn(BATCH, NUM_SPEAKERS, FREQ_BINS, T, device=device)
    (dec(x) * w).sum().backward()
    x2 = x.detach().clone().requires_grad_()
    sd = dec.state_dict()
    hmid = F.gelu(F.linear(x2, sd["decoder.0.weight"], sd["decoder.0.bias"]))
    logits = F.linear(hmid, sd["decoder.3.weight"], sd["decoder.3.bias"])
    ref = torch.sigmoid(logits.view(BATCH, T, NUM_SPEAKERS, FREQ_BINS).permute(0, 2, 3, 1))
    (ref * w).sum().backward()
    assert float((x.grad - x2.grad).abs().max()) < 1e-4 * max(1.0, float(x2.grad.abs().max()))
    assert dec.get_parameter("decoder.0.weight").grad is not None
