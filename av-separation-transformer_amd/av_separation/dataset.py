"""SyntheticAVDataset -- bit-compatible generator of the reference's synthetic clips (host side, numpy).

Follows ``/root/reference/src/av_separation/dataset.py`` (constructor 33-65, item 70-120, STFT 122-135,
lip patch 137-151) call for call, because the random stream is part of the contract: one
``np.random.default_rng(idx)`` per item, drawn in this order -- S amplitudes U(0.3,1); per speaker a
frequency jitter U(0.95,1.05) then a phase U(0,2pi); then, speakers outer / frames inner, one
N(0,0.05) patch of the centre half of the frame.  tests/test_dataset.py pins it against golden items
produced by the reference.
"""
import math

import numpy as np
import torch
from torch.utils.data import Dataset


class SyntheticAVDataset(Dataset):
    """item -> {"mixed_spec": (F,T), "lip_frames": (S*num_frames,H,W), "clean_specs": (S,F,T)} float32."""

    def __init__(self, num_samples: int = 1000, sample_rate: int = 8000, duration: float = 1.0,
                 n_fft: int = 512, hop_length: int = 128, num_frames: int = 25, frame_h: int = 32,
                 frame_w: int = 32, speaker_freqs: tuple = (220.0, 440.0), seed: int = 42):
        self.num_samples = num_samples
        self.sample_rate = sample_rate
        self.duration = duration
        self.n_fft = n_fft
        self.hop_length = hop_length
        self.num_frames = num_frames
        self.frame_h = frame_h
        self.frame_w = frame_w
        self.speaker_freqs = speaker_freqs
        self.num_speakers = len(speaker_freqs)
        self.rng = np.random.default_rng(seed)           # kept for API parity; items use rng(idx)
        self.num_samples_audio = int(sample_rate * duration)
        self.t = np.linspace(0, duration, self.num_samples_audio, endpoint=False)
        self.freq_bins = n_fft // 2 + 1
        self.T = 1 + self.num_samples_audio // hop_length
        self._window = np.hanning(n_fft)                 # symmetric Hann, float64

    def __len__(self) -> int:
        return self.num_samples

    def _stft(self, audio: np.ndarray) -> np.ndarray:
        """|rFFT| of Hann-windowed frames, hop `hop_length`, tail frames zero-padded -> (F, T) float32."""
        n = len(audio)
        cols = np.empty((self.freq_bins, self.T), dtype=np.float32)
        buf = np.empty(self.n_fft, dtype=np.float32)
        for i in range(self.T):
            lo = i * self.hop_length
            seg = audio[lo:lo + self.n_fft] if lo + self.n_fft <= n else audio[lo:]
            buf[:] = 0
            buf[:len(seg)] = seg
            buf *= self._window                          # float32 *= float64, rounded back to float32
            cols[:, i] = np.abs(np.fft.rfft(buf))
        return cols

    def _lip_patch(self, energy: float, rng: np.random.Generator) -> np.ndarray:
        h0, h1 = self.frame_h // 4, 3 * self.frame_h // 4
        w0, w1 = self.frame_w // 4, 3 * self.frame_w // 4
        level = min(1.0, energy * 20.0)
        jitter = rng.normal(0, 0.05, (h1 - h0, w1 - w0)).astype(np.float32)
        patch = np.zeros((self.frame_h, self.frame_w), dtype=np.float32)
        patch[h0:h1, w0:w1] = np.clip(level + jitter, 0, 1)
        return patch

    def _voices(self, rng: np.random.Generator):
        amps = rng.uniform(0.3, 1.0, size=self.num_speakers)
        voices = []
        for f0, amp in zip(self.speaker_freqs, amps):
            f = f0 * rng.uniform(0.95, 1.05)
            phi = rng.uniform(0, 2 * math.pi)
            voices.append((amp * np.sin(2 * math.pi * f * self.t + phi)).astype(np.float32))
        mix = voices[0]
        for v in voices[1:]:
            mix = mix + v
        return voices, mix.astype(np.float32)

    def waveforms(self, idx: int):
        """The time-domain signals behind item ``idx`` -- (mixed (L,), clean (S, L)) float32 -- i.e. what the reference
        feeds to ``_stft`` (dataset.py:74-89).  For producing spectrograms on the device with
        ``av_separation.stft.stft_magnitude`` instead of on the host in ``__getitem__``."""
        voices, mix = self._voices(np.random.default_rng(idx))
        return torch.from_numpy(mix), torch.from_numpy(np.stack(voices, axis=0))

    def __getitem__(self, idx: int):
        rng = np.random.default_rng(idx)
        voices, mix = self._voices(rng)

        mixed_spec = self._stft(mix)
        clean = np.stack([self._stft(v) for v in voices], axis=0)

        step = self.num_samples_audio // self.num_frames
        lips = np.empty((self.num_speakers * self.num_frames, self.frame_h, self.frame_w), dtype=np.float32)
        k = 0
        for v in voices:
            for fi in range(self.num_frames):
                lo = fi * step
                hi = min(lo + step, self.num_samples_audio)
                lips[k] = self._lip_patch(float(np.mean(v[lo:hi] ** 2)), rng)
                k += 1
        return {"mixed_spec": torch.from_numpy(mixed_spec), "lip_frames": torch.from_numpy(lips),
                "clean_specs": torch.from_numpy(clean)}
