"""N4: the HIP STFT-magnitude front-end (avsep_op_stft_mag through av_separation.stft) against the
REFERENCE's spectrograms -- tests/golden/dataset.npz holds `mixed_spec` / `clean_specs` produced by the reference's
SyntheticAVDataset._stft (dataset.py:122-135) -- and against a float64 restatement on edge shapes.
Tolerance: the reference runs a float32 FFT (1e-5 from float64 at |spec| ~ 84), the HIP path a float32 DFT-as-GEMM
(observed 4e-5 at the same scale = 4.5e-7 relative): gate 5e-6 * max|spec|."""
import json

import numpy as np
import pytest
import torch

from helpers import maxabs

pytestmark = pytest.mark.gpu
REL_TOL = 5e-6


def _kwargs(g, tag):
    kw = json.loads(str(g[f"{tag}.kwargs"]))
    if "speaker_freqs" in kw:
        kw["speaker_freqs"] = tuple(kw["speaker_freqs"])
    return kw


def _stft64(audio, n_fft, hop):
    """float64 restatement of dataset.py:122-135 (zero-padded tail frames, symmetric Hann, |rfft|)."""
    L = len(audio)
    T = 1 + L // hop
    pad = np.concatenate([audio.astype(np.float64), np.zeros(n_fft + hop)])
    frames = np.stack([pad[i * hop:i * hop + n_fft] for i in range(T)])
    return np.abs(np.fft.rfft(frames * np.hanning(n_fft), axis=1)).T


@pytest.mark.parametrize("tag", ["small", "cfg1", "cfg4", "cfg5"])
def test_stft_matches_reference_spectrograms(golden, tag):
    from av_separation import SyntheticAVDataset
    from av_separation.stft import stft_magnitude
    g = golden("dataset")
    ds = SyntheticAVDataset(**_kwargs(g, tag))
    dev = torch.device("cuda:0")
    waves = []
    for idx in (0, 1, 3):
        mix, clean = ds.waveforms(idx)
        waves += [mix] + list(clean)
    spec = stft_magnitude(torch.stack(waves).to(dev), ds.n_fft, ds.hop_length).cpu().numpy()
    assert spec.shape == (len(waves), ds.freq_bins, ds.T) and spec.dtype == np.float32
    S, row = ds.num_speakers, 0
    for idx in (0, 1, 3):
        got = {"mixed_spec": spec[row], "clean_specs": spec[row + 1:row + 1 + S]}
        row += 1 + S
        for key, a in got.items():
            full = f"{tag}.{idx}.{key}"
            scale = float(np.abs(a).max())
            if full in g:
                assert maxabs(a, g[full]) < REL_TOL * scale, full
            else:
                assert list(a.shape) == list(g[full + ".shape"])
                assert maxabs(np.ascontiguousarray(a).reshape(-1)[::13], g[full + ".slice"]) < REL_TOL * scale, full
                assert abs(a.astype(np.float64).sum() - float(g[full + ".sum"])) < REL_TOL * scale * a.size


@pytest.mark.parametrize("B,L,n_fft,hop", [(3, 4000, 128, 32), (1, 8000, 512, 128), (2, 100, 64, 4), (5, 36, 128, 128),
                                           (1, 64000, 512, 128), (7, 1000, 256, 100), (130, 512, 32, 8)])
def test_stft_edge_shapes_against_float64(B, L, n_fft, hop):
    from av_separation.stft import stft_magnitude
    rng = np.random.default_rng(L + n_fft)
    audio = (rng.standard_normal((B, L)) * rng.uniform(0.1, 2.0, (B, 1))).astype(np.float32)
    got = stft_magnitude(torch.from_numpy(audio).cuda(), n_fft, hop).cpu().numpy()
    assert got.shape == (B, n_fft // 2 + 1, 1 + L // hop)
    ref = np.stack([_stft64(a, n_fft, hop) for a in audio])
    assert maxabs(got, ref) < REL_TOL * float(np.abs(ref).max())
    one = stft_magnitude(torch.from_numpy(audio[0]).cuda(), n_fft, hop)                # 1-D input -> (F, T)
    assert torch.equal(one.cpu(), torch.from_numpy(got[0]))


def test_stft_rejects_what_it_cannot_do():
    from av_separation.stft import stft_magnitude
    with pytest.raises(RuntimeError, match="no CPU"):
        stft_magnitude(torch.zeros(2, 512), 128, 32)
    with pytest.raises(RuntimeError, match="multiple of 32"):
        stft_magnitude(torch.zeros(2, 512, device="cuda"), 100, 20)
    with pytest.raises(RuntimeError, match="multiples of 4"):
        stft_magnitude(torch.zeros(2, 510, device="cuda"), 128, 32)
