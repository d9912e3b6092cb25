// HBM read ceilings for conv_nin_h's access pattern: what a kernel that only READS can reach on this chip.
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_read_probe.hip -o tools/abl/hbm_read_probe && tools/abl/hbm_read_probe
// Variants (all 256-thread workgroups, 12.6 GB read per launch, nothing written but one word per thread):
//   copy        float4 copy (the guide's 6.3 TB/s figure counts read + write bytes)
//   write       float4 stores only
//   read        global_load_dwordx4, grid-stride over the whole buffer, 8 loads in flight per thread
//   dma         global_load_lds_dwordx4 into a 3 x 16 KB ring per workgroup, counted vmcnt, one barrier per 16 KB (conv_nin_h's staging, no consumer)
//   dma-planes  the same, but consecutive 16 KB pieces of a workgroup come from 41 planes 300 MB apart (conv_nin_h's real address stream)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_copy(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}

__global__ __launch_bounds__(256) void k_write(f32x4* __restrict__ out, size_t n, float v) {
    const f32x4 x = {v, v + 1.f, v + 2.f, v + 3.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = x;
}

__global__ __launch_bounds__(256) void k_read(const f32x4* __restrict__ in, float* __restrict__ out, size_t n) {
    f32x4 s = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n; i += 8 * stride) {
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(&in[i + k * stride]);
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[k];
    }
    for (; i < n; i += stride) s += in[i];
    if (s.x + s.y + s.z + s.w == 12345.678f) out[threadIdx.x] = s.x;
}

__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}

__device__ __forceinline__ void glds16v(const void* src, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_dst) : "memory", "m0");
}

// persistent workgroups; item = 16 KB; PLANES = 1: items are consecutive 16 KB pieces of the buffer dealt round-robin to workgroups;
// PLANES = 41: a workgroup's block b covers 16 KB at offset b * 16 KB of EACH of 41 planes (plane stride = bytes / 41), visited plane by plane
// MIS = 1: the 128-byte record a lane group of 8 fetches straddles two planes -- 32 bytes (2 lanes) from the tail of plane p - 1's record,
// 96 bytes from the head of plane p's: the skip-concat's K chunks when they are cut every 4 octets regardless of tensor boundaries
template <int S, int PLANES, int MIS = 0>
__global__ __launch_bounds__(256, 2) void k_dma(const char* __restrict__ in, float* __restrict__ out, size_t bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t plane = (bytes / PLANES) & ~(size_t)16383;
    const size_t blocks = plane / 16384;                       // 16 KB blocks per plane
    const size_t my_blocks = blockIdx.x < blocks ? (blocks - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const size_t T = my_blocks * PLANES;
    auto issue = [&](size_t q, int stage) {
        const size_t b = blockIdx.x + (q / PLANES) * gridDim.x, p = q % PLANES;
        const char* src = in + p * plane + b * 16384;
        if constexpr (MIS) {
            const char* prev = in + (p ? p - 1 : PLANES - 1) * plane + b * 16384;
            const int u = lane & 7;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t rec = (size_t)(wave + 4 * r) * 1024 + (lane >> 3) * 128;
                glds16v(u < 2 ? prev + rec + 96 + u * 16 : src + rec + (u - 2) * 16, lds0 + stage * 16384 + (wave + 4 * r) * 1024);
            }
            return;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) glds16(src, (unsigned)((wave + 4 * r) * 1024 + lane * 16), lds0 + stage * 16384 + (wave + 4 * r) * 1024);
    };
    for (int s = 0; s < S; ++s) if ((size_t)s < T) issue(s, s);
    float acc = 0.f;
    int st = 0;
    for (size_t q = 0; q < T; ++q) {
        // wait for item q (issued S items ago): at most S - 1 younger items x 4 pieces in flight
        if (q + S - 1 < T) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (S - 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc += *reinterpret_cast<const float*>(smem + st * 16384 + threadIdx.x * 64);
        __syncthreads();
        if (q + S < T) issue(q + S, st);
        st = st + 1 == S ? 0 : st + 1;
    }
    if (acc == 12345.678f) out[threadIdx.x] = acc;
}

int main() {
    const size_t bytes = (size_t)41 * 2359296 * 128;           // 12.38 GB: the bench model's skip-concat as P16 planes
    char* in; char* outb; float* out;
    CHECK(hipMalloc(&in, bytes));
    CHECK(hipMalloc(&outb, bytes / 2));
    CHECK(hipMalloc(&out, 4096));
    CHECK(hipMemset(in, 1, bytes));
    CHECK(hipMemset(outb, 0, bytes / 2));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto time = [&](const char* name, double gb, auto fn) {
        fn(); CHECK(hipDeviceSynchronize());
        float best = 1e9f, sum = 0.f;
        for (int i = 0; i < 5; ++i) {
            CHECK(hipEventRecord(e0)); fn(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; sum += ms;
        }
        CHECK(hipGetLastError());
        printf("%-28s %8.3f ms avg %8.3f ms best   %6.2f TB/s (best)\n", name, sum / 5, best, gb / best);
    };
    const double gb = bytes / 1e9;
    time("copy (6.2 GB -> 6.2 GB)", gb, [&] { hipLaunchKernelGGL(k_copy, dim3(256 * 16), dim3(256), 0, 0, (const f32x4*)in, (f32x4*)outb, bytes / 32); });
    time("write only, 8 WG/CU (6.2 GB)", gb / 2, [&] { hipLaunchKernelGGL(k_write, dim3(256 * 8), dim3(256), 0, 0, (f32x4*)outb, bytes / 32, 1.0f); });
    time("write only, 16 WG/CU (6.2 GB)", gb / 2, [&] { hipLaunchKernelGGL(k_write, dim3(256 * 16), dim3(256), 0, 0, (f32x4*)outb, bytes / 32, 2.0f); });
    for (int wgs : {256 * 4, 256 * 8, 256 * 16})
        time(wgs == 1024 ? "read x4 WG/CU" : wgs == 2048 ? "read x8 WG/CU" : "read x16 WG/CU", gb, [&] { hipLaunchKernelGGL(k_read, dim3(wgs), dim3(256), 0, 0, (const f32x4*)in, out, bytes / 16); });
    CHECK(hipFuncSetAttribute((const void*)&k_dma<3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 16384 + 27 * 1024));
    CHECK(hipFuncSetAttribute((const void*)&k_dma<4, 41>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 16384 + 11 * 1024));
    CHECK(hipFuncSetAttribute((const void*)&k_dma<2, 41>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 16384 + 60 * 1024));
    CHECK(hipFuncSetAttribute((const void*)&k_dma<3, 41>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 16384 + 40 * 1024));
    // LDS sized like conv_nin_h<6>: 75 KB -> two workgroups per CU
    time("dma S3 linear, 2 WG/CU", gb, [&] { hipLaunchKernelGGL((k_dma<3, 1>), dim3(512), dim3(256), 3 * 16384 + 27 * 1024, 0, in, out, bytes); });
    time("dma S3 41 planes, 2 WG/CU", gb, [&] { hipLaunchKernelGGL((k_dma<3, 41>), dim3(512), dim3(256), 3 * 16384 + 27 * 1024, 0, in, out, bytes); });
    CHECK(hipFuncSetAttribute((const void*)&k_dma<3, 41, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 16384 + 27 * 1024));
    time("dma S3 41 planes straddling", gb, [&] { hipLaunchKernelGGL((k_dma<3, 41, 1>), dim3(512), dim3(256), 3 * 16384 + 27 * 1024, 0, in, out, bytes); });
    time("dma S4 41 planes, 2 WG/CU", gb, [&] { hipLaunchKernelGGL((k_dma<4, 41>), dim3(512), dim3(256), 4 * 16384 + 11 * 1024, 0, in, out, bytes); });
    time("dma S2 41 planes, 2 WG/CU", gb, [&] { hipLaunchKernelGGL((k_dma<2, 41>), dim3(512), dim3(256), 2 * 16384 + 43 * 1024, 0, in, out, bytes); });
    time("dma S2 41 planes, 1 WG/CU", gb, [&] { hipLaunchKernelGGL((k_dma<2, 41>), dim3(256), dim3(256), 2 * 16384 + 60 * 1024, 0, in, out, bytes); });
    time("dma S3 41 planes, 1 WG/CU", gb, [&] { hipLaunchKernelGGL((k_dma<3, 41>), dim3(256), dim3(256), 3 * 16384 + 40 * 1024, 0, in, out, bytes); });
    time("dma S2 41 planes, 3 WG/CU", gb, [&] { hipLaunchKernelGGL((k_dma<2, 41>), dim3(768), dim3(256), 2 * 16384 + 16 * 1024, 0, in, out, bytes); });
    return 0;
}
