"""Numerics of emulating an f32 dot product with bf16 matrix instructions (next-round candidate).

x = x1 + x2 + x3 with bf16 pieces (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)); a product a*b is taken as
a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1 (the dropped terms are <= 2^-24 relative), every bf16 x bf16 product is exact in
f32, blocks of 32 products are summed inside the instruction and added to an f32 accumulator.  K = 1764 (CNN2's 9 x 196),
activations ~ N(0, 50), weights ~ N(0, 0.03):

    f32 fma chain       max err 2.7e-4   rms 4.6e-5     (what conv_igemm does today)
    bf16x3, 6 products  max err 5.6e-5   rms 8.7e-6     (block sums taken as exact -- optimistic, but not worse than f32)
    bf16x3, 3 products  max err 1.1e-3   rms 2.8e-4     (too coarse for the 1e-4 parity bar)
"""
import numpy as np


def to_bf16(x):
    u = x.astype(np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).view(np.float32)


def split3(x):
    a1 = to_bf16(x)
    r = (x - a1).astype(np.float32)
    a2 = to_bf16(r)
    return a1, a2, to_bf16((r - a2).astype(np.float32))


def main():
    rng = np.random.default_rng(0)
    K, M = 1764, 4000
    a = (rng.standard_normal((M, K)) * 50).astype(np.float32)
    b = (rng.standard_normal((K,)) * 0.03).astype(np.float32)
    truth = a.astype(np.float64) @ b.astype(np.float64)
    acc = np.zeros(M, np.float32)
    for k in range(K):
        acc = (acc + a[:, k] * b[k]).astype(np.float32)
    report = {"f32 chain": np.abs(acc - truth)}
    a1, a2, a3 = split3(a)
    b1, b2, b3 = split3(b)
    pairs = [(a1, b1), (a1, b2), (a2, b1), (a2, b2), (a1, b3), (a3, b1)]
    for n in (6, 3):
        acc = np.zeros(M, np.float32)
        for k0 in range(0, K, 32):
            blk = np.zeros(M, np.float64)
            for x, y in pairs[:n][::-1]:
                blk += (x[:, k0:k0 + 32].astype(np.float64) * y[k0:k0 + 32].astype(np.float64)).sum(1)
            acc = (acc + blk.astype(np.float32)).astype(np.float32)
        report["bf16x3, %d products" % n] = np.abs(acc - truth)
    print("max|truth| %.1f" % np.abs(truth).max())
    for k, e in report.items():
        print("%-20s max err %.3g  rms %.3g" % (k, e.max(), np.sqrt((e ** 2).mean())))


if __name__ == "__main__":
    main()
