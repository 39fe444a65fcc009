#!/usr/bin/env python3
"""Coefficients of the sin^2 polynomial in csrc/device_common.h (SINSQ_C1..C5) and its fp32 error check."""
import numpy as np
f32 = np.float32
Z = (np.pi / 2 * 1.0005) ** 2
N = 4000
zz = (np.cos(np.pi * (np.arange(N) + 0.5) / N) + 1) / 2 * Z
tgt = np.where(zz > 1e-12, np.sin(np.sqrt(zz)) ** 2 / zz, 1.0)
A1 = np.vander(zz, 6, increasing=True)[:, 1:]
coef, *_ = np.linalg.lstsq(A1 * zz[:, None], (tgt - 1) * zz, rcond=None)
c = [1.0] + [float(f32(x)) for x in coef]
print("coefficients:", c[1:])

def fma(a, b, cc):
    return (a.astype(np.float64) * b.astype(np.float64) + cc.astype(np.float64)).astype(f32)

def sin_sq(t):
    n = np.rint(t * f32(1 / np.pi)).astype(f32)
    r = t
    for k in (3.140625, 9.675025939941406e-4, 1.5099580252808664e-7):
        r = fma(n, np.full_like(t, f32(-k)), r)
    z = (r * r).astype(f32)
    p = np.full_like(t, f32(c[-1]))
    for ck in c[-2::-1]:
        p = fma(p, z, np.full_like(t, f32(ck)))
    return (z * p).astype(f32)

rng = np.random.default_rng(0)
for span in (1, 8, 64, 1000, 25000):
    t = rng.uniform(-span, span, 2_000_000).astype(f32)
    print(f"|t| <= {span}: max abs error {np.abs(sin_sq(t) - np.sin(t.astype(np.float64)) ** 2).max():.2e}")
