#!/usr/bin/env python3
"""VERDICT r02 next #6(a), the numerical half: fp32 Winograd F(4x4,3x3) against F(2x2,3x3) and the direct sum at the decoder
conv's channel counts (304 -> 256, reference blocks.py:117), all three held against the float64 direct convolution.  CPU only
(NumPy; float32 transforms, float32 accumulation emulated by float32 matmuls); spatial size reduced, channels full: the error of a
Winograd output depends on the reduction length and the transform constants, not on the image size.
Output: max and rms error relative to the output's max magnitude (the unit of the parity tests: 2e-5 of scale).
Result recorded in profiles/r03_winograd_f4x4_fp32_error.txt."""
import numpy as np

rng = np.random.default_rng(1993)
N, H, W, CIN, COUT = 2, 24, 32, 304, 256
w = (rng.normal(0, 1, (3, 3, CIN, COUT)) / np.sqrt(9 * CIN)).astype(np.float32)


def direct(x, dt):
    xp = np.zeros((N, H + 2, W + 2, CIN), dt)
    xp[:, 1:-1, 1:-1] = x
    y = np.zeros((N, H, W, COUT), dt)
    for kh in range(3):
        for kw in range(3):
            y += (xp[:, kh:kh + H, kw:kw + W].reshape(-1, CIN) @ w[kh, kw].astype(dt)).reshape(N, H, W, COUT)
    return y


def winograd(x, m, BT, G, AT):
    """F(m x m, 3 x 3), everything in float32"""
    t = m + 2
    BT, G, AT = BT.astype(np.float32), G.astype(np.float32), AT.astype(np.float32)
    U = np.einsum("ik,klcn,jl->ijcn", G, w, G).astype(np.float32)                     # (t, t, cin, cout)
    xp = np.zeros((N, H + 2, W + 2, CIN), np.float32)
    xp[:, 1:-1, 1:-1] = x
    y = np.zeros((N, H, W, COUT), np.float32)
    for a in range(H // m):
        for b in range(W // m):
            d = xp[:, a * m:a * m + t, b * m:b * m + t]                               # (N, t, t, cin)
            V = np.einsum("ik,nklc,jl->nijc", BT, d, BT).astype(np.float32)
            M = np.einsum("nijc,ijco->nijo", V, U).astype(np.float32)                  # float32 accumulation over cin
            y[:, a * m:(a + 1) * m, b * m:(b + 1) * m] = np.einsum("ik,nklo,jl->nijo", AT, M, AT).astype(np.float32)
    return y


BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], np.float64)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], np.float64)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)

operands = {"forward operand (a ReLU6 output in [0, 6])": np.clip(rng.normal(1.5, 2.0, (N, H, W, CIN)), 0, 6).astype(np.float32),
            "input-gradient operand (zero mean, unbounded)": rng.normal(0, 1, (N, H, W, CIN)).astype(np.float32)}
for what, x in operands.items():
    ref = direct(x.astype(np.float64), np.float64)
    scale = np.abs(ref).max()
    print(what)
    for name, y in (("direct fp32", direct(x, np.float32)), ("winograd F(2x2,3x3) fp32", winograd(x, 2, BT2, G2, AT2)),
                    ("winograd F(4x4,3x3) fp32", winograd(x, 4, BT4, G4, AT4))):
        e = y.astype(np.float64) - ref
        print(f"  {name:26s} max {np.abs(e).max() / scale:.2e}  rms {np.sqrt((e ** 2).mean()) / scale:.2e}   of the output's max magnitude")
