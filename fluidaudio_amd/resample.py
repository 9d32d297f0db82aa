"""Host-side mirror of the resampling arithmetic of ``AudioConverter`` (reference:
Sources/FluidAudio/Shared/AudioConverter.swift) over the HIP C ABI (csrc/resample.hip).

``linear_resample`` is ``AudioConverter.linearResample`` (:388-442), the only resampling arithmetic present in the
reference tree (used for > 2 channels); ``resample_poly`` is an extension with its own specification (the default path
of the reference is Apple's closed-source AVAudioConverter: parity unpinned)."""
from __future__ import annotations

import ctypes as C
from math import gcd

import numpy as np

from . import _lib as L

TARGET_SAMPLE_RATE = 16000.0  # AudioConverter's target format (:24-32)


def linear_resample(channel_data, sample_rate: float, target_rate: float = TARGET_SAMPLE_RATE, ctx: L.Context | None = None) -> np.ndarray:
    """channel_data: [channels, frames] float32 (planar, like floatChannelData) -> mono float32 at target_rate."""
    x = np.ascontiguousarray(channel_data, np.float32)
    if x.ndim == 1:
        x = x[None, :]
    ch, frames = x.shape
    n_out = L.lib().fa_resample_linear_frames(frames, float(sample_rate), float(target_rate))
    out = np.zeros(max(n_out, 1), np.float32)
    if frames == 0 or n_out == 0:
        return out[:0]
    ctx = ctx or L.default_context()
    got = C.c_int64()
    ctx.check(L.lib().fa_resample_linear(ctx.handle, x.ctypes.data, ch, frames, float(sample_rate), float(target_rate),
                                         out.ctypes.data, out.size, C.byref(got)), "fa_resample_linear")
    return out[:got.value]


def resample_poly(x, up: int, down: int, ctx: L.Context | None = None) -> np.ndarray:
    """EXTENSION (parity unpinned): polyphase FIR resampling by up/down, the specification of scipy.signal.resample_poly."""
    x = np.ascontiguousarray(x, np.float32)
    g = gcd(int(up), int(down))
    up, down = int(up) // g, int(down) // g
    n_out = L.lib().fa_resample_poly_frames(x.size, up, down)
    out = np.zeros(max(n_out, 1), np.float32)
    if x.size == 0:
        return out[:0]
    ctx = ctx or L.default_context()
    got = C.c_int64()
    ctx.check(L.lib().fa_resample_poly(ctx.handle, x.ctypes.data, x.size, up, down, out.ctypes.data, out.size, C.byref(got)),
              "fa_resample_poly")
    return out[:got.value]


def poly_taps(up: int, down: int):
    """The FIR the kernel uses (host computation, no GPU needed): (taps float32, pre_remove)."""
    n, pre = C.c_int64(), C.c_int64()
    L.lib().fa_resample_poly_taps(up, down, None, 0, C.byref(n), C.byref(pre))
    taps = np.zeros(n.value, np.float32)
    L.lib().fa_resample_poly_taps(up, down, taps.ctypes.data, taps.size, C.byref(n), C.byref(pre))
    return taps, pre.value
