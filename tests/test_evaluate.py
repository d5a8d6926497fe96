"""Eval harness against numbers produced by the reference's demo.py functions
(tests/golden/make_golden.py::make_eval)."""
import json

import numpy as np
import pytest
import torch

from av_separation import SyntheticAVDataset, AVSeparationTransformer
from av_separation.evaluate import snr_db, permutation_snr, evaluate_separation
from helpers import golden_state


def test_snr_known_answers(golden):
    g = golden("eval")
    assert abs(snr_db(g["a"], g["b"]) - g["snr_db"]) < 1e-9
    assert abs(permutation_snr(g["a"], g["b"]) - g["perm_snr"]) < 1e-9
    assert abs(permutation_snr(g["b"][[2, 0, 1]] * 1.01, g["b"]) - g["perm_snr_shuffled"]) < 1e-9
    assert g["perm_snr_shuffled"] > 30          # the shuffled copy is found by the permutation search


@pytest.mark.gpu
def test_evaluate_separation_matches_reference_on_gpu(golden):
    g, t = golden("eval"), golden("trained_tiny")
    c = t["config"]
    m = AVSeparationTransformer(c["F"], c["d"], c["h"], c["Le"], c["Lf"], c["S"], dropout=0.0)
    sd = m.state_dict()
    for k, v in golden_state(t).items():
        sd[k] = torch.from_numpy(np.ascontiguousarray(v))
    m.load_state_dict(sd)
    dev = torch.device("cuda:0")
    m.to(dev)
    ds = SyntheticAVDataset(num_samples=64, sample_rate=8000, duration=0.496, n_fft=128, hop_length=128,
                            num_frames=5, frame_h=16, frame_w=16)
    in_snr, out_snr = evaluate_separation(m, ds, dev, num_eval=6)
    assert abs(in_snr - g["in_snr"]) < 1e-6
    assert abs(out_snr - g["out_snr"]) < 1e-3     # dB; masks agree to ~1e-6
    assert out_snr - in_snr > 10                  # the trained model separates (demo.py:177 "SNR improvement")
