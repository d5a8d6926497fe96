"""STFT magnitude front-end on the device.

``stft_magnitude(audio, n_fft, hop_length)`` computes what ``SyntheticAVDataset._stft`` computes per clip on the host
(/root/reference/src/av_separation/dataset.py:122-135: symmetric Hann window ``np.hanning(n_fft)``, hop ``hop_length``,
``T = 1 + L // hop_length`` frames, tail frames zero-padded, ``|np.fft.rfft|``) for a whole batch of waveforms that
already live in HBM, as ONE fp32-MFMA GEMM launch of libavsep_hip.so (``avsep_op_stft_mag``): overlapping rows of the
waveform against a windowed real-DFT basis, magnitude in the epilogue, output (B, n_fft//2+1, T) like the reference.
The reference notes the inverse transform (phase reconstruction) as absent (README.md:140); so is it here.

No CPU fallback: CPU tensors raise.  ``SyntheticAVDataset.waveforms(idx)`` gives the host-side waveforms of an item so
that spectrograms can be produced on the device instead of in ``__getitem__``.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _native

_BASIS = {}     # (device, n_fft) -> basis tensor


def _basis(device, n_fft: int) -> torch.Tensor:
    key = (device, int(n_fft))
    b = _BASIS.get(key)
    if b is None:
        lib = _native.load()
        n = int(lib.avsep_stft_basis_floats(n_fft))
        _native.check(n, "avsep_stft_basis_floats")
        b = torch.empty(n, dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            _native.check(lib.avsep_stft_basis(b.data_ptr(), n_fft,
                                               C.c_void_p(torch.cuda.current_stream(device).cuda_stream)),
                          "avsep_stft_basis")
        _BASIS[key] = b
    return b


def stft_magnitude(audio: torch.Tensor, n_fft: int = 512, hop_length: int = 128) -> torch.Tensor:
    """audio (B, L) or (L,) float32 on a ROCm device -> magnitude spectrogram (B, n_fft//2+1, 1 + L//hop) / (F, T)."""
    if not isinstance(audio, torch.Tensor) or audio.dim() not in (1, 2):
        raise RuntimeError("audio: expected a (B, L) or (L,) tensor")
    if audio.device.type != "cuda":
        raise RuntimeError(f"audio is on {audio.device}: the MI355X path needs a ROCm device tensor; there is no CPU "
                           "fallback in this package")
    squeeze = audio.dim() == 1
    x = audio.reshape(1, -1) if squeeze else audio
    x = x.float().contiguous()
    B, L = x.shape
    if L <= 0 or B <= 0:
        raise RuntimeError("audio: empty tensor")
    F, T = n_fft // 2 + 1, 1 + L // hop_length
    spec = torch.empty(B, F, T, dtype=torch.float32, device=x.device)
    dev = x.device
    with torch.cuda.device(dev):
        basis = _basis(dev, n_fft)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _native.check(_native.load().avsep_op_stft_mag(x.data_ptr(), basis.data_ptr(), spec.data_ptr(), B, L, n_fft,
                                                      hop_length, st), "avsep_op_stft_mag")
    return spec[0] if squeeze else spec
