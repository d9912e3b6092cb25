// How well do TWO waves of one SIMD share the f16 matrix pipe?  One workgroup per CU; 256 / 512 / 768 threads = 1 / 2 / 3 waves per SIMD,
// every wave runs bare v_mfma_f32_16x16x32_f16 on ACC rotating accumulators (ACC = 1: a dependent chain).  Reports cycles per MFMA
// per SIMD (16 = the pipe's rate).     hipcc --offload-arch=gfx950 -O3 tools/mfma_two_waves.hip -o tools/mfma_two_waves
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int ACC>
__global__ void probe(float* out, long long* cyc, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x4 acc[ACC];
    for (int i = 0; i < ACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k % ACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k % ACC], 0, 0, 0);
    }
    const long long t1 = clock64();
    f32x4 s = acc[0];
    for (int i = 1; i < ACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s.z + s.w;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int ACC>
static void run(int threads) {
    const int blocks = 256, iters = 2000;
    float* out; long long* cyc;
    hipMalloc(&out, blocks * threads * 4); hipMalloc(&cyc, blocks * 16 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<ACC><<<blocks, threads>>>(out, cyc, 10);
    hipEventRecord(e0); probe<ACC><<<blocks, threads>>>(out, cyc, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c[16]; hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
    const int wps = threads / 256;
    // clock64 ticks at 100 MHz on this chip?  report both the wall-time rate (assuming all SIMDs alike) and the raw counter
    const double mfma_per_simd = (double)iters * 16 * wps;
    printf("ACC %d  %d wave(s)/SIMD: %.3f ms  -> %.2f ns per MFMA per SIMD  (wave 0 counter %lld)\n", ACC, wps, ms, ms * 1e6 / mfma_per_simd, c[0]);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int t : {256, 512, 768}) { run<1>(t); run<2>(t); run<4>(t); run<8>(t); }
    return 0;
}
