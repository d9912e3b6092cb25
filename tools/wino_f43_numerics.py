#!/usr/bin/env python
"""Error of Winograd F(4x4,3x3) against F(2x2,3x3) and the direct form, all with float32 products and float32 accumulation
over the input channels (what the MFMA pipe does), against a float64 direct conv.  Decides whether F(4x4,3x3) -- 36
multiplies per 16 outputs instead of 64, i.e. 1.78x fewer MFMAs than the shipped F(2x2,3x3) kernel -- can meet the
parity bar (residual branch within 5e-6 relative, tests/test_hip_parity.py).  CPU only.

    python tools/wino_f43_numerics.py
"""
import numpy as np


def transforms(m):
    if m == 2:
        BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
        G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
        AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
    else:   # Lavin & Gray F(4x4,3x3), points 0, +-1, +-2, inf
        BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                       [0, 4, 0, -5, 0, 1]], np.float64)
        G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
                     np.float64)
        AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)
    return BT, G, AT


def wino(x, w, m):
    """x [H, W, C] float32 (H, W multiples of m, zero padded outside), w [3, 3, C, K]; f32 everywhere but the filter transform."""
    BT, G, AT = transforms(m)
    t = m + 2
    H, W, C = x.shape
    K = w.shape[3]
    U = np.einsum("ai,ijck,bj->abck", G, w.astype(np.float64), G).astype(np.float32)          # host, float64, rounded once
    xp = np.zeros((H + 2, W + 2, C), np.float32)
    xp[1:-1, 1:-1] = x
    out = np.zeros((H, W, K), np.float32)
    BT32, AT32 = BT.astype(np.float32), AT.astype(np.float32)
    for ty in range(0, H, m):
        for tx in range(0, W, m):
            d = xp[ty:ty + t, tx:tx + t]                                                    # [t, t, C]
            V = np.einsum("ai,ijc,bj->abc", BT32, d, BT32).astype(np.float32)               # f32 adds (exact order differs, same class)
            M = np.zeros((t, t, K), np.float32)
            for c in range(C):                                                              # f32 accumulation over channels, in order
                M += V[:, :, c, None] * U[:, :, c, :]
            out[ty:ty + m, tx:tx + m] = np.einsum("ai,ijk,bj->abk", AT32, M, AT32).astype(np.float32)
    return out


def direct(x, w, dtype):
    H, W, C = x.shape
    xp = np.zeros((H + 2, W + 2, C), dtype)
    xp[1:-1, 1:-1] = x
    out = np.zeros((H, W, w.shape[3]), dtype)
    for dy in range(3):
        for dx in range(3):
            for c in range(C):
                out += xp[dy:dy + H, dx:dx + W, c, None].astype(dtype) * w[dy, dx, c].astype(dtype)
    return out


def main():
    rng = np.random.default_rng(0)
    for C, K in ((196, 32), (96, 32), (32, 16)):
        x = np.maximum(rng.normal(0, 40, (16, 16, C)), -5).astype(np.float32)               # PReLU-like activations of the bench model
        w = (rng.normal(0, 1, (3, 3, C, K)) * np.sqrt(2.0 / (9 * C))).astype(np.float32)
        ref = direct(x, w, np.float64)
        scale = np.max(np.abs(ref))
        e_dir = np.max(np.abs(direct(x, w, np.float32) - ref)) / scale
        e_f2 = np.max(np.abs(wino(x, w, 2) - ref)) / scale
        e_f4 = np.max(np.abs(wino(x, w, 4) - ref)) / scale
        print("cin %3d: max|ref| %.1f   relative max error: direct f32 %.2e   F(2x2,3x3) %.2e   F(4x4,3x3) %.2e   (F4/F2 = %.1f)"
              % (C, scale, e_dir, e_f2, e_f4, e_f4 / e_f2))


if __name__ == "__main__":
    main()
