"""Numerics of f32-accurate contractions on the 16-bit matrix pipe (CPU only).

An f32 operand is split into two f16 pieces, x*s = hi + lo with hi = f16(x*s), lo = f16(x*s - hi) (s a power of two), and
a product a*b is taken as ah*bh + ah*bl + al*bh (3 MFMA products; the dropped al*bl is <= 2^-22 relative).  With round to
nearest the pair (hi, lo) carries 22-23 significant bits, every f16 x f16 product is exact in f32, and the hardware adds
blocks of 32 products to an f32 accumulator.  Compared here against

    * the f32 fma chain the shipped kernels run (v_mfma_f32_16x16x4_f32 == a k-ordered fmaf chain),
    * bf16 x 3 pieces / 6 products (tools/bf16x3_numerics.py, r01's candidate),

(1) on single dot products of the bench model's longest contractions (direct 3x3: K = 9*196 = 1764, the Winograd domain of
the same layer: 16 frequencies x K = 196, and A1 || B1: K = 1301), and (2) through the whole L12_F196to48 x2 chain with
the oracle's seeded weights: the attenuated output (1e-4 max-abs bar) and the un-attenuated residual branch of
tests/test_hip_parity.py::test_residual_branch_relative_error (5e-6 relative bar).

    python tools/f16x3_numerics.py            # prints the table committed as profiles/r03_f16x3_numerics.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dcscn_oracle as O  # noqa: E402

F32 = np.float32


def f16_split(x, scale=1.0, flush=False):
    """(hi, lo) as float64 arrays holding f16-representable values; flush=True models an MFMA that flushes f16 subnormals."""
    xs = (x.astype(F32) * F32(scale)).astype(F32)
    hi = xs.astype(np.float16)
    lo = (xs - hi.astype(F32)).astype(F32).astype(np.float16)
    if flush:
        tiny = np.float16(2.0 ** -14)
        hi = np.where(np.abs(hi) < tiny, np.float16(0), hi)
        lo = np.where(np.abs(lo) < tiny, np.float16(0), lo)
    return hi.astype(np.float64), lo.astype(np.float64)


def to_bf16(x):
    u = x.astype(F32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).view(F32)


def bf16_split3(x):
    a1 = to_bf16(x)
    r = (x.astype(F32) - a1).astype(F32)
    a2 = to_bf16(r)
    a3 = to_bf16((r - a2).astype(F32))
    return a1.astype(np.float64), a2.astype(np.float64), a3.astype(np.float64)


def pow2_scale(x, top):
    """power of two s with max|x| * s in [top / 2, top)."""
    m = float(np.max(np.abs(x)))
    if m == 0.0:
        return 1.0
    return 2.0 ** (np.floor(np.log2(top / m)))


def gemm_f32_chain(a, b):
    """k-ordered f32 fma chain: what v_mfma_f32_16x16x4_f32 computes (cdna_hip_programming.md section 3)."""
    acc = np.zeros((a.shape[0], b.shape[1]), F32)
    a = a.astype(F32)
    b = b.astype(F32)
    for k in range(a.shape[1]):
        acc = (acc.astype(np.float64) + a[:, k:k + 1].astype(np.float64) * b[k:k + 1, :].astype(np.float64)).astype(F32)
    return acc


def gemm_blocks(pairs, K, inv_scale, acc=None, shape=None):
    """MFMA model: per block of 32 k and per product pair one instruction D = C + sum_k a_k b_k, rounded to f32 once."""
    if acc is None:
        acc = np.zeros(shape, F32)
    for k0 in range(0, K, 32):
        for x, y in pairs:                       # small terms first
            acc = (acc.astype(np.float64) + x[:, k0:k0 + 32] @ y[k0:k0 + 32, :]).astype(F32)
    return acc


def gemm_f16x3(a, b, sa=None, sb=None, flush=False, acc=None):
    sa = pow2_scale(a, 2.0 ** 15) if sa is None else sa
    sb = pow2_scale(b, 2.0 ** 15) if sb is None else sb
    ah, al = f16_split(a, sa, flush)
    bh, bl = f16_split(b, sb, flush)
    out = gemm_blocks([(al, bh), (ah, bl), (ah, bh)], a.shape[1], None, acc, (a.shape[0], b.shape[1]))
    return out, 1.0 / (sa * sb)


def gemm_bf16(a, b, n_products):
    a1, a2, a3 = bf16_split3(a)
    b1, b2, b3 = bf16_split3(b)
    pairs = [(a1, b1), (a1, b2), (a2, b1), (a2, b2), (a1, b3), (a3, b1)][:n_products][::-1]
    return gemm_blocks(pairs, a.shape[1], None, None, (a.shape[0], b.shape[1]))


def report(name, got, truth):
    e = np.abs(got.astype(np.float64) - truth)
    print("    %-34s max err %.3g   rms %.3g   (max|truth| %.4g)" % (name, e.max(), np.sqrt((e ** 2).mean()), np.abs(truth).max()))


# ---------------------------------------------------------------------------------------------------------------------
# (1) single contractions
# ---------------------------------------------------------------------------------------------------------------------

def single_contractions():
    rng = np.random.default_rng(0)
    M, N = 512, 64
    print("(1) single contractions, %d x %d outputs, activations ~ N(0, 50) rectified by PReLU(0.2), weights ~ He" % (M, N))
    for label, K, wstd in (("direct 3x3, K = 9*196 = 1764", 1764, np.sqrt(2.0 / 1764)), ("A1 || B1, K = 1301", 1301, np.sqrt(2.0 / 1301)),
                           ("Winograd domain, one frequency, K = 196", 196, np.sqrt(2.0 / 1764))):
        a = rng.standard_normal((M, K)) * 50
        a = np.where(a > 0, a, 0.2 * a)
        if "Winograd" in label:                  # B^T d B sums four input values with signs
            a = a + rng.standard_normal((M, K)) * 50 - rng.standard_normal((M, K)) * 50 - rng.standard_normal((M, K)) * 50
        a = a.astype(F32)
        b = (rng.standard_normal((K, N)) * wstd).astype(F32)
        truth = a.astype(np.float64) @ b.astype(np.float64)
        print("  " + label)
        report("f32 fma chain (shipped)", gemm_f32_chain(a, b), truth)
        got, inv = gemm_f16x3(a, b)
        report("f16 hi/lo, 3 products", got.astype(np.float64) * inv, truth)
        got, inv = gemm_f16x3(a, b, flush=True)
        report("f16 hi/lo, 3 products, flush subnormals", got.astype(np.float64) * inv, truth)
        got, inv = gemm_f16x3(a, b, sa=1.0)
        report("f16 hi/lo, 3 products, act scale 1", got.astype(np.float64) * inv, truth)
        got, inv = gemm_f16x3(a, b, sa=1.0, flush=True)
        report("  ... and flushed subnormals", got.astype(np.float64) * inv, truth)
        report("bf16 x 3 pieces, 6 products", gemm_bf16(a, b, 6), truth)
        report("bf16 x 3 pieces, 3 products", gemm_bf16(a, b, 3), truth)


# ---------------------------------------------------------------------------------------------------------------------
# (2) the whole chain
# ---------------------------------------------------------------------------------------------------------------------

class ConvModel:
    """Replaces oracle.conv2d_same for the layers the 16-bit kernels would take (3x3 with >= 24 input channels and >= 2
    output tiles; 1x1 with >= 32 input channels); everything else, and bias / PReLU / adds, runs in float32."""

    def __init__(self, mode, act_scale=None, flush=False):
        self.mode, self.act_scale, self.flush = mode, act_scale, flush
        self.max_act = {}

    def eligible(self, w):
        kh, kw, cin, cout = w.shape
        if kh == 3:
            return cin >= 24 and cout > 16
        return kh == 1 and cin >= 32

    def __call__(self, x, w):
        kh, kw, cin, cout = w.shape
        n, h, wd, _ = x.shape
        x = x.astype(F32)
        w = w.astype(F32)
        ph, pw = kh // 2, kw // 2
        xp = np.zeros((n, h + 2 * ph, wd + 2 * pw, cin), F32)
        xp[:, ph:ph + h, pw:pw + wd, :] = x
        key = "%dx%d %d->%d" % (kh, kw, cin, cout)
        self.max_act[key] = max(self.max_act.get(key, 0.0), float(np.abs(x).max()))
        if self.mode == "f32" or not self.eligible(w):
            acc = np.zeros((n * h * wd, cout), F32)
            for dy in range(kh):
                for dx in range(kw):
                    acc = gemm_f32_chain_fast(xp[:, dy:dy + h, dx:dx + wd, :].reshape(-1, cin), w[dy, dx], acc)
            return acc.reshape(n, h, wd, cout)
        sb = pow2_scale(w, 2.0 ** 15)
        sa = self.act_scale if self.act_scale is not None else pow2_scale(x, 2.0 ** 15)
        acc = np.zeros((n * h * wd, cout), F32)
        for dy in range(kh):
            for dx in range(kw):
                a = xp[:, dy:dy + h, dx:dx + wd, :].reshape(-1, cin)
                if self.mode == "f16x3":
                    acc, _ = gemm_f16x3(a, w[dy, dx], sa, sb, self.flush, acc)
                else:
                    raise ValueError(self.mode)
        return (acc.astype(np.float64) / (sa * sb)).astype(F32).reshape(n, h, wd, cout)


def gemm_f32_chain_fast(a, b, acc):
    """f32 accumulation in blocks of 4 k (one v_mfma_f32_16x16x4_f32 each, summed as the instruction does: sequential fma).
    A full per-k Python loop over 1764 terms and 4608 pixels is slow; numpy f32 matmul per 4-block keeps f32 roundoff class."""
    for k0 in range(0, a.shape[1], 4):
        acc = acc + a[:, k0:k0 + 4] @ b[k0:k0 + 4, :]
    return acc.astype(F32)


def chain():
    cfg = O.make_config()
    rng = np.random.default_rng(1)
    x = rng.uniform(0, 255, (1, 32, 32, 1)).astype(F32)
    x2 = np.stack([O.pil_bicubic(x[i], 2) for i in range(x.shape[0])]).astype(F32)
    print("\n(2) L12_F196to48 x2, oracle weights, one %dx%d patch, all eligible layers on the modelled pipe" % x.shape[1:3])
    results = {}
    for tag, seed, bare in (("attenuated output (bar: 1e-4 max-abs on 0-255)", 0, False), ("bare residual branch (bar: 5e-6 relative)", 7, True)):
        weights = O.synthetic_weights(cfg, seed=seed)
        xx2 = x2
        if bare:
            weights = dict(weights)
            weights["R-CNN1/conv_W"] = weights["R-CNN1/conv_W"] * 100.0
            xx2 = np.zeros_like(x2)
        truth = O.forward(cfg, weights, x, xx2, dtype=np.float64)
        print("  " + tag)
        real = O.conv2d_same
        try:
            for name, model in (("f32 chain (shipped arithmetic)", ConvModel("f32")),
                                ("f16 hi/lo x3, per-tensor pow2 scale", ConvModel("f16x3")),
                                ("f16 hi/lo x3, fixed act scale 1", ConvModel("f16x3", act_scale=1.0)),
                                ("f16 hi/lo x3, fixed act scale 16", ConvModel("f16x3", act_scale=16.0)),
                                ("f16 hi/lo x3, scale 1, flush subnormals", ConvModel("f16x3", act_scale=1.0, flush=True)),
                                ("f16 hi/lo x3, scale 16, flush subnormals", ConvModel("f16x3", act_scale=16.0, flush=True))):
                O.conv2d_same = model
                y = O.forward(cfg, weights, x, xx2, dtype=F32)
                e = np.abs(y.astype(np.float64) - truth)
                print("    %-42s max-abs %.3g   relative to max|y| %.3g" % (name, e.max(), e.max() / np.abs(truth).max()))
                results[(tag, name)] = e.max()
                last = model
        finally:
            O.conv2d_same = real
        if not bare:
            print("  largest |activation| entering each conv (seed-0 weights, U(0,255) input): " +
                  ", ".join("%s: %.0f" % kv for kv in last.max_act.items()))


if __name__ == "__main__":
    single_contractions()
    chain()
