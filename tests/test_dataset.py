"""SyntheticAVDataset must be BIT-identical to the reference's generator: the golden
items were produced by the reference (tests/golden/make_golden.py::make_dataset).  Also mirrors the
property tests of the reference suite (tests/test_model.py:237-290)."""
import json

import numpy as np
import pytest
import torch

from av_separation import SyntheticAVDataset


def _kwargs(g, tag):
    kw = json.loads(str(g[f"{tag}.kwargs"]))
    if "speaker_freqs" in kw:
        kw["speaker_freqs"] = tuple(kw["speaker_freqs"])
    return kw


@pytest.mark.parametrize("tag", ["small", "cfg1", "cfg4", "cfg5"])
def test_items_bit_identical_to_reference(golden, tag):
    g = golden("dataset")
    ds = SyntheticAVDataset(**_kwargs(g, tag))
    assert [ds.freq_bins, ds.T, len(ds)] == list(g[f"{tag}.dims"])
    for idx in (0, 1, 3):
        it = ds[idx]
        for key in ("mixed_spec", "lip_frames", "clean_specs"):
            a = it[key].numpy()
            full = f"{tag}.{idx}.{key}"
            if full in g:
                assert a.dtype == np.float32 and np.array_equal(a, g[full]), full
            else:
                assert list(a.shape) == list(g[full + ".shape"])
                assert np.array_equal(a.reshape(-1)[::13], g[full + ".slice"]), full
                assert a.astype(np.float64).sum() == g[full + ".sum"]


def test_shapes_ranges_determinism():
    ds = SyntheticAVDataset(num_samples=10, sample_rate=8000, duration=0.5, n_fft=128, hop_length=32,
                            num_frames=10, frame_h=16, frame_w=16)
    assert len(ds) == 10
    a, b, c = ds[3], ds[3], ds[4]
    assert a["mixed_spec"].shape == (65, 1 + 4000 // 32)
    assert a["lip_frames"].shape == (20, 16, 16) and a["clean_specs"].shape == (2, 65, 126)
    assert float(a["lip_frames"].min()) >= 0 and float(a["lip_frames"].max()) <= 1
    assert torch.equal(a["mixed_spec"], b["mixed_spec"]) and not torch.equal(a["mixed_spec"], c["mixed_spec"])
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=4)))
    assert batch["mixed_spec"].shape[0] == 4 and batch["lip_frames"].dim() == 4


def test_waveforms_are_what_the_items_are_made_of():
    """`waveforms(idx)` (the input of the device STFT, av_separation/stft.py) replays the item's random stream: the
    host STFT of those signals is the item's spectrogram bit for bit."""
    ds = SyntheticAVDataset(num_samples=6, sample_rate=8000, duration=0.5, n_fft=128, hop_length=32, num_frames=5,
                            frame_h=8, frame_w=8, speaker_freqs=(220.0, 440.0, 660.0))
    for idx in (0, 4):
        mix, clean = ds.waveforms(idx)
        it = ds[idx]
        assert mix.shape == (4000,) and clean.shape == (3, 4000) and mix.dtype == torch.float32
        assert np.array_equal(ds._stft(mix.numpy()), it["mixed_spec"].numpy())
        assert np.array_equal(np.stack([ds._stft(c.numpy()) for c in clean]), it["clean_specs"].numpy())
