// What does the f16 matrix pipe sustain when every MFMA takes its operands from DIFFERENT registers, as conv3_h's tap does?
// One "tap" = NT channel tiles x 4 pixel rows x 3 products: 2 NT A fragments, 8 B fragments (4 rows x hi / lo), 4 NT accumulators,
// all held in registers (no LDS, no memory traffic).  ORDER 0: conv3_h's (per tile, per row: wl*xh, wh*xl, wh*xh back to back on one
// accumulator); ORDER 1: per tile the three products outermost (an accumulator is touched every 4th instruction); ORDER 2: every MFMA
// re-reads ONE operand pair (the bare-pipe reference of tools/mfma_two_waves.hip).  Random (seeded) operand bits: the pipe's power
// state is data dependent.  Reports ns per MFMA per SIMD for 1 and 2 waves per SIMD (8.0 ns = 16 cycles at 2.0 GHz).
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/mfma_operands.hip -o tools/mfma_operands
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int NT, int ORDER>
__global__ __launch_bounds__(512, 2) void probe(const h8* src, float* out, int iters) {
    h8 wh[NT], wl[NT], xh[4], xl[4];
    const h8* p = src + threadIdx.x % 64;
    for (int n = 0; n < NT; ++n) { wh[n] = p[(2 * n) * 64]; wl[n] = p[(2 * n + 1) * 64]; }
    for (int m = 0; m < 4; ++m) { xh[m] = p[(2 * NT + 2 * m) * 64]; xl[m] = p[(2 * NT + 2 * m + 1) * 64]; }
    f32x4 acc[4][NT];
    for (int m = 0; m < 4; ++m) for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            if (ORDER == 0) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[n], xh[m], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[n], xl[m], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[n], xh[m], acc[m][n], 0, 0, 0);
                }
            } else if (ORDER == 1) {
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[n], xh[m], acc[m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[n], xl[m], acc[m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[n], xh[m], acc[m][n], 0, 0, 0);
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int k = 0; k < 3; ++k) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[0], xh[0], acc[m][n], 0, 0, 0);
            }
        }
        // keep the operands opaque (no hoisting / folding across iterations)
        for (int n = 0; n < NT; ++n) asm volatile("" : "+v"(wh[n]), "+v"(wl[n]));
        for (int m = 0; m < 4; ++m) asm volatile("" : "+v"(xh[m]), "+v"(xl[m]));
    }
    f32x4 s = {0, 0, 0, 0};
    for (int m = 0; m < 4; ++m) for (int n = 0; n < NT; ++n) s += acc[m][n];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s.z + s.w;
}

// The same tap (96 channels x 64 pixels x K = 32 x 3 products) on v_mfma_f32_32x32x16_f16: 3 channel tiles x 2 pixel tiles x 2 K halves x 3
// products = 36 instructions of twice the work, 12 A + 8 B fragments, 6 x 16 accumulator registers -- half the operand reads per MAC.
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512, 2) void probe32(const h8* src, float* out, int iters) {
    h8 wh[3][2], wl[3][2], xh[2][2], xl[2][2];
    const h8* p = src + threadIdx.x % 64;
    for (int n = 0; n < 3; ++n) for (int k = 0; k < 2; ++k) { wh[n][k] = p[(4 * n + 2 * k) * 64]; wl[n][k] = p[(4 * n + 2 * k + 1) * 64]; }
    for (int m = 0; m < 2; ++m) for (int k = 0; k < 2; ++k) { xh[m][k] = p[(12 + 4 * m + 2 * k) * 64]; xl[m][k] = p[(12 + 4 * m + 2 * k + 1) * 64]; }
    f32x16 acc[3][2];
    for (int n = 0; n < 3; ++n) for (int m = 0; m < 2; ++m) for (int i = 0; i < 16; ++i) acc[n][m][i] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[n][k], xh[m][k], acc[n][m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[n][k], xl[m][k], acc[n][m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[n][k], xh[m][k], acc[n][m], 0, 0, 0);
            }
        for (int n = 0; n < 3; ++n) for (int k = 0; k < 2; ++k) asm volatile("" : "+v"(wh[n][k]), "+v"(wl[n][k]));
        for (int m = 0; m < 2; ++m) for (int k = 0; k < 2; ++k) asm volatile("" : "+v"(xh[m][k]), "+v"(xl[m][k]));
    }
    float s = 0.0f;
    for (int n = 0; n < 3; ++n) for (int m = 0; m < 2; ++m) for (int i = 0; i < 16; ++i) s += acc[n][m][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static void run32(int threads, const h8* src, float* out) {
    const int blocks = 256, iters = 300;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe32<<<blocks, threads>>>(src, out, 10);
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0); probe32<<<blocks, threads>>>(src, out, iters); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const int wps = threads / 256;
    const double eq = (double)iters * 72 * wps;                 // in 16x16x32 instructions of the same work
    printf("32x32x16, same tap      %d wave(s)/SIMD: %.3f ms  -> %.2f ns per 16x16x32-equivalent MFMA per SIMD\n", wps, best, best * 1e6 / eq);
}

template <int NT, int ORDER>
static void run(int threads, const h8* src, float* out) {
    const int blocks = 256, iters = 300;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NT, ORDER><<<blocks, threads>>>(src, out, 10);
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0); probe<NT, ORDER><<<blocks, threads>>>(src, out, iters); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const int wps = threads / 256;
    const double mfma_per_simd = (double)iters * 12 * NT * wps;
    printf("NT %d order %d  %d wave(s)/SIMD: %.3f ms  -> %.2f ns per MFMA per SIMD\n", NT, ORDER, wps, best, best * 1e6 / mfma_per_simd);
}

int main() {
    const int n = 64 * 32;
    std::vector<unsigned short> h(n * 8);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; const unsigned e = 10 + ((s >> 20) % 12); v = (unsigned short)(((s >> 31) << 15) | (e << 10) | ((s >> 8) & 0x3ff)); }   // finite f16, |x| ~ 2^-5 .. 2^6
    h8* src; float* out;
    hipMalloc(&src, n * 16); hipMalloc(&out, 256 * 512 * 4);
    hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice);
    for (int t : {256, 512}) {
        run<6, 2>(t, src, out); run<6, 0>(t, src, out); run<6, 1>(t, src, out); run32(t, src, out);
        run<3, 0>(t, src, out); run<3, 1>(t, src, out);      // (these overflow to inf: MFMAs on non-finite accumulators take 50x longer)
    }
    return 0;
}
