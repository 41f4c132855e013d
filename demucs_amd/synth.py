"""Deterministic synthetic audio for parity fixtures and benchmarks (no datasets offline).

`synth_mix` is a pure function of its arguments on every machine: noise comes from the
counter-based generator of `weights.py`; the tonal variant adds a few sinusoids, a chirp and a
silent gap (exercises the per-segment normalisation with near-zero variance and the reflect
padding, SURVEY.md §8d "Value distribution").
"""
from __future__ import annotations

import numpy as np

from .weights import counter_normal

__all__ = ["synth_mix"]


def synth_mix(seed: int, length: int, kind: str = "noise", channels: int = 2, scale: float = 0.1,
              offset: int = 0) -> np.ndarray:
    """(channels, length) float32.  `offset` lets a rank generate only its slice of a long track."""
    out = np.empty((channels, length), dtype=np.float64)
    for c in range(channels):
        out[c] = counter_normal(seed * 16 + c, length, offset) * scale
    if kind == "noise":
        return out.astype(np.float32)
    if kind != "tones":
        raise ValueError(kind)
    t = (np.arange(offset, offset + length, dtype=np.float64)) / 44100.0
    tone = 0.3 * np.sin(2 * np.pi * 110.0 * t) + 0.2 * np.sin(2 * np.pi * 1760.0 * t + 0.5)
    chirp = 0.25 * np.sin(2 * np.pi * (200.0 + 900.0 * t) * t)
    sig = np.stack([tone + chirp, 0.8 * tone - 0.6 * chirp])[:channels] + 0.2 * out
    # a 0.5 s silent gap every 3 s starting at 1.2 s, a DC step after 5 s on channel 0
    gap = ((t - 1.2) % 3.0) < 0.5
    sig[:, gap] = 0.0
    sig[0, t > 5.0] += 0.05
    return sig.astype(np.float32)
