"""Shared helpers for the parity tests (test infrastructure; may import oracle/)."""
import numpy as np

from oracle import seeded


def low_bits(name, shape, seed):
    """16 seeded low-order bits per float32 word of matrix `name` (numpy's PCG64 stream, stable across versions); make_golden.py uses
    the same function, so fixture and test agree on every bit of the weights."""
    import zlib
    rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
    return rng.integers(0, 1 << 16, size=shape, dtype=np.uint32)


def golden_state(g):
    """Weights for a golden fixture: committed (trained_*) or regenerated from the seed (fwd_*)."""
    c = g["config"]
    if any(k.startswith("w.") for k in g):
        state = {k[2:]: g[k] for k in g if k.startswith("w.")}
        # trained_cfg1: matrices are committed as the upper 16 bits of their float32 words (the reference produced the
        # fixture's outputs with exactly these rounded weights, tests/golden/make_golden.py::make_trained_cfg1)
        # trained_d512 (config "lowbits"): the low 16 bits of every matrix word come from a seeded stream (low_bits below), so the
        # weights the reference ran with are full 24-bit float32 values -- their low split terms are not identically zero
        for k in g:
            if k.startswith("wh."):
                u = g[k].astype(np.uint32) << 16
                if c.get("lowbits"):
                    u = u | low_bits(k[3:], u.shape, int(c["lowbits"]))
                state[k[3:]] = u.view(np.float32)
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
