// Does the matrix pipe overlap with VALU work of OTHER waves on the same SIMD (gfx950)?  One workgroup of 16 waves per CU
// (4 per SIMD); per SIMD, `nm` waves run a dependent-free MFMA stream, `nv` waves a v_pk_fma stream, the rest exit.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o tools/mfma_valu_overlap && tools/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// the same with v_mfma_f32_16x16x32_bf16 (16 cycles each: 48 per iteration = 768 matrix cycles)
__global__ __launch_bounds__(1024) void probe_bf16(float* out, int nm, int nv, int iters) {
    const int wave = threadIdx.x >> 6, slot = wave >> 2;
    if (slot < nm) {
        f32x4 acc[6];
        for (int i = 0; i < 6; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f + i); b[i] = (__bf16)(1.0f + i); }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 48; ++i) acc[i % 6] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i % 6], 0, 0, 0);
        float s = 0;
        for (int i = 0; i < 6; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
        out[blockIdx.x * 1024 + threadIdx.x] = s;
    } else if (slot < nm + nv) {
        f32x2 v[8];
        for (int i = 0; i < 8; ++i) v[i] = f32x2{threadIdx.x * 1e-4f + i, 1.0f};
        const f32x2 m = {0.999f, 1.001f}, c = {1e-3f, -1e-3f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 192; ++i) v[i % 8] = v[i % 8] * m + c;
        float s = 0;
        for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
        out[blockIdx.x * 1024 + threadIdx.x] = s;
    }
}

__global__ __launch_bounds__(1024) void probe(float* out, int nm, int nv, int iters) {
    const int wave = threadIdx.x >> 6, slot = wave >> 2;        // wave w runs on SIMD w & 3; slot = its index on that SIMD
    if (slot < nm) {
        f32x4 acc[6];
        for (int i = 0; i < 6; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float a = threadIdx.x * 1e-3f, b = 1.0f + a;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 24; ++i) acc[i % 6] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i % 6], 0, 0, 0);
        float s = 0;
        for (int i = 0; i < 6; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
        out[blockIdx.x * 1024 + threadIdx.x] = s;
    } else if (slot < nm + nv) {
        f32x2 v[8];
        for (int i = 0; i < 8; ++i) v[i] = f32x2{threadIdx.x * 1e-4f + i, 1.0f};
        const f32x2 m = {0.999f, 1.001f}, c = {1e-3f, -1e-3f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 192; ++i) v[i % 8] = v[i % 8] * m + c;      // 192 v_pk_fma = 768 issue cycles ~ 24 MFMAs
        float s = 0;
        for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
        out[blockIdx.x * 1024 + threadIdx.x] = s;
    }
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000;
    const int cases[][2] = {{1, 0}, {0, 1}, {1, 1}, {2, 0}, {0, 2}, {2, 2}, {1, 3}, {3, 1}, {4, 0}, {0, 4}};
    for (int bf = 0; bf < 2; ++bf)
    for (auto& c : cases) {
        auto kern = bf ? probe_bf16 : probe;
        if (bf && c[0] + c[1] > 2) continue;
        hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 0, 0, out, c[0], c[1], 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 0, 0, out, c[0], c[1], iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        // per iteration: an MFMA wave needs 24 * 32 = 768 matrix cycles, a VALU wave 192 * 4 = 768 issue cycles
        printf("%s %d MFMA waves + %d VALU waves per SIMD: %.3f ms  = %.0f cycles per iteration at 2.2 GHz (768 per wave-iteration of work)\n",
               bf ? "bf16 16x16x32" : "f32 16x16x4  ", c[0], c[1], ms, ms * 1e-3 * 2.2e9 / iters);
    }
    return 0;
}
