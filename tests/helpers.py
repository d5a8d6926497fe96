"""Shared helpers for the parity tests (test infrastructure; may import oracle/)."""
import numpy as np

from oracle import seeded


def golden_state(g):
    """Weights for a golden fixture: committed (trained_*) or regenerated from the seed (fwd_*)."""
    c = g["config"]
    if any(k.startswith("w.") for k in g):
        state = {k[2:]: g[k] for k in g if k.startswith("w.")}
    else:
        shapes = seeded.model_shapes(c["F"], c["d"], c["h"], c["Le"], c["Lf"], c["S"])
        state = seeded.fill_state(shapes, c["seed"])
    return state


def golden_inputs(g):
    c = g["config"]
    if "in.mixed" in g:
        return g["in.mixed"], g["in.lips"]
    return seeded.inputs(c["seed"], c["B"], c["F"], c["T"], c["N"], c["H"], c["W"])


def sliced(a, step):
    return np.ascontiguousarray(a).reshape(-1)[::step]


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))
