"""Fit behind `gelu_exact` (demucs_amd/csrc/common.h): erf(t) = 1 - 2^(t Q(t)), Q of degree 8, weighted least squares on
Chebyshev nodes so that the ABSOLUTE error of erf is what is minimised; coefficients folded to take t = |x| (not |x|/sqrt 2).
Prints the coefficients and the float32-emulated error of the resulting GELU against float64.  Needs scipy (CPU only)."""
import numpy as np
from scipy.special import erf, erfc

TMAX = 3.95
t = (np.cos(np.pi * (np.arange(8000) + 0.5) / 8000) * 0.5 + 0.5) * TMAX
t = t[t > 1e-4]
w = erfc(t) * t
coef, *_ = np.linalg.lstsq(np.vander(t, 9, increasing=True) * w[:, None], np.log2(erfc(t)) / t * w, rcond=None)
r2 = np.sqrt(0.5)
cg = np.array([c * r2 ** (k + 1) for k, c in enumerate(coef)]).astype(np.float32)
xmax = np.float32(TMAX / r2)

x = np.concatenate([np.linspace(-9, 9, 1800001), np.logspace(-8, 0, 2000), -np.logspace(-8, 0, 2000)]).astype(np.float32)
ax = np.minimum(np.abs(x), xmax)
q = np.full_like(x, cg[-1])
for c in cg[-2::-1]:
    q = (q * ax + c).astype(np.float32)
e = np.exp2((q * ax).astype(np.float32)).astype(np.float32)
g = ((np.float32(0.5) * x).astype(np.float32) * (np.float32(1) + np.copysign((np.float32(1) - e).astype(np.float32), x)).astype(np.float32))
x64 = x.astype(np.float64)
want = 0.5 * x64 * (1 + erf(x64 * r2))
err = np.abs(g - want)
print("clamp |x| at", float(xmax))
print("coefficients (constant term first):", ", ".join(f"{float(c):.9e}f" for c in cg))
print(f"gelu max abs err {err.max():.3e}; max err / |x| {(err / np.maximum(np.abs(x64), 1e-30)).max():.3e}")
