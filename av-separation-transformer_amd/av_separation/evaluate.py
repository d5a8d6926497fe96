"""Evaluation harness of the reference's demo: plain spectrogram-domain SNR of the separated
outputs with permutation-invariant speaker matching.  Semantics of ``/root/reference/demo.py``:
``snr_db`` 24-28, ``evaluate_separation`` 31-64, ``_permutation_snr`` 67-80; "SNR improvement" = output SNR after
minus input SNR before (demo.py:177).  Host-side numpy on the model's outputs; the forward itself runs on the
HIP path.  Pinned against the reference's numbers in tests/test_evaluate.py.
"""
import math
from itertools import permutations

import numpy as np
import torch


def snr_db(signal: np.ndarray, noise: np.ndarray, eps: float = 1e-8) -> float:
    """10 log10( mean(signal^2) / (mean(noise^2) + eps) + eps ).  The powers stay in the arrays' dtype (float32
    for dataset items), as in the reference: numpy's weak-scalar promotion keeps `+ eps` in float32 too."""
    ratio = np.mean(np.square(signal)) / (np.mean(np.square(noise)) + eps)
    return 10 * math.log10(ratio + eps)


def permutation_snr(separated: np.ndarray, targets: np.ndarray) -> float:
    """Best mean-over-speakers SNR over all assignments of outputs to targets; arrays (S, F, T)."""
    n = separated.shape[0]
    scores = [np.mean([snr_db(targets[t], separated[s] - targets[t]) for t, s in enumerate(order)])
              for order in permutations(range(n))]
    return float(max(scores))


def evaluate_separation(model, dataset, device, num_eval: int = 20):
    """(mean input SNR of the mixture w.r.t. each clean source, mean best-permutation output SNR) over the first
    ``num_eval`` items, one clip per forward like the reference."""
    model.eval()
    snr_in, snr_out = [], []
    with torch.no_grad():
        for i in range(min(num_eval, len(dataset))):
            item = dataset[i]
            mixed = item["mixed_spec"].unsqueeze(0).to(device)
            lips = item["lip_frames"].unsqueeze(0).to(device)
            clean = item["clean_specs"].numpy()
            separated, _ = model(mixed, lips)
            separated = separated.squeeze(0).cpu().numpy()
            mix = item["mixed_spec"].numpy()
            snr_in.extend(snr_db(clean[s], mix - clean[s]) for s in range(clean.shape[0]))
            snr_out.append(permutation_snr(separated, clean))
    return float(np.mean(snr_in)), float(np.mean(snr_out))
