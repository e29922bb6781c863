"""TEST INFRASTRUCTURE (like everything under oracle/): element-wise comparison used by the feature / gradient parity tests and smoke().

north_star: "within 1e-3 rel on bbox/feature tensors".  Every element must satisfy |a - b| <= rtol*|b| + atol_c, where atol_c is set PER
CHANNEL (last axis) from that channel's own scale: atol_c = atol_frac * max|b[..., c]|.  A channel whose values are all small is therefore
held to its own magnitude (a norm-wise max|a-b| / max|b| over the whole tensor would let it be 100 % wrong).  Defaults: rtol 1e-3,
atol_frac 1e-4 (fp32 accumulation noise of a 27 x 64-term dot product relative to the channel's largest value)."""
import os

import numpy as np


def assert_close_per_channel(a, b, rtol=1e-3, atol_frac=1e-4, name="", channel_axis=-1):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if channel_axis != -1:
        a, b = np.moveaxis(a, channel_axis, -1), np.moveaxis(b, channel_axis, -1)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    if b.size == 0:
        return
    b2 = b.reshape(-1, b.shape[-1]) if b.ndim > 1 else b.reshape(-1, 1)
    a2 = a.reshape(b2.shape)
    scale = np.abs(b2).max(axis=0, keepdims=True)
    atol = atol_frac * scale
    err = np.abs(a2 - b2) - (rtol * np.abs(b2) + atol)
    report = os.environ.get("SEEVCN_TOL_REPORT")
    if report:                                                  # measurement aid: the atol_frac this comparison would need at its rtol
        need = float((np.maximum(np.abs(a2 - b2) - rtol * np.abs(b2), 0.0) / np.maximum(scale, 1e-30)).max())
        with open(report, "a") as f:
            f.write(f"{name or '?'}\tshape={b.shape}\trtol={rtol:g}\tatol_frac={atol_frac:g}\tneeded={need:.3e}\n")
    if (err > 0).any():
        r, c = np.unravel_index(np.argmax(err), err.shape)
        raise AssertionError(f"{name}: element ({r},{c}) got {a2[r, c]!r} want {b2[r, c]!r} (|diff| {abs(a2[r, c] - b2[r, c]):.3e} > "
                             f"{rtol:g}*|want| + {atol[0, c]:.3e}); {(err > 0).sum()} of {err.size} elements out of tolerance")
