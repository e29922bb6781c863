"""Element-wise comparison used by the feature / gradient parity tests.

north_star: "within 1e-3 rel on bbox/feature tensors".  Every element must satisfy |a - b| <= rtol*|b| + atol_c, where atol_c is set PER
CHANNEL (last axis) from that channel's own scale: atol_c = atol_frac * max|b[..., c]|.  A channel whose values are all small is therefore
held to its own magnitude (a norm-wise max|a-b| / max|b| over the whole tensor would let it be 100 % wrong).  Defaults: rtol 1e-3,
atol_frac 1e-4 (fp32 accumulation noise of a 27 x 64-term dot product relative to the channel's largest value)."""
import numpy as np


def assert_close_per_channel(a, b, rtol=1e-3, atol_frac=1e-4, name=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    if b.size == 0:
        return
    b2 = b.reshape(-1, b.shape[-1]) if b.ndim > 1 else b.reshape(-1, 1)
    a2 = a.reshape(b2.shape)
    atol = atol_frac * np.abs(b2).max(axis=0, keepdims=True)
    err = np.abs(a2 - b2) - (rtol * np.abs(b2) + atol)
    if (err > 0).any():
        r, c = np.unravel_index(np.argmax(err), err.shape)
        raise AssertionError(f"{name}: element ({r},{c}) got {a2[r, c]!r} want {b2[r, c]!r} (|diff| {abs(a2[r, c] - b2[r, c]):.3e} > "
                             f"{rtol:g}*|want| + {atol[0, c]:.3e}); {(err > 0).sum()} of {err.size} elements out of tolerance")
