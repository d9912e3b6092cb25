// Two-process triage of the cross-process corruption recorded in DESIGN 6: does a process that runs
// conv3_h perturb OTHER processes on the same GPU whatever they run (platform defect), or only victims that use LDS-DMA /
// counted-vmcnt pipelines (our kernels are not wave save / restore safe)?
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/xproc_triage.hip -o tools/xproc_triage -Ldcscn-super-resolution_amd -ldcscn_hip '-Wl,-rpath,$ORIGIN/../dcscn-super-resolution_amd'
//   tools/xproc_triage aggressor <c3h | mfma | copy> <seconds>     keeps the GPU busy with that kernel
//   tools/xproc_triage victim <copy | memcpy | lds | ldsdma | regs | c3h> <seconds>
//       copy    global -> global copy kernel + a check kernel (no LDS, no MFMA)
//       memcpy  hipMemcpyAsync device -> device + the check kernel
//       lds     workgroups keep 64 KB of a pattern in LDS (ds_write) for ~1 ms and re-verify it continuously (ds_read)
//       ldsdma  the same LDS image filled by global_load_lds_dwordx4 behind a counted s_waitcnt vmcnt
//       regs    waves keep 128 VGPRs + 64 MFMA accumulators of known values live for ~1 ms and verify them
//       c3h     conv3_h<6> on a CNN2-shaped layer, every run compared bit for bit with the first
//       cin1<cs>  conv_cin1 (the first feature layer, 1 -> cs channels; VALU + LDS only) through the product library, 96 patches
//       pkfma / fma   chains of v_pk_fma_f32 / v_fma_f32 with an exactly known result;  ldssmall   1.6 KB of LDS per workgroup, pattern re-read
//       igemm   a 32 -> 32 channel 3x3 layer on conv_igemm (f32 MFMA, register-staged LDS), every run compared bit for bit with the first
//       chain0..3  a dependent chain of MFMAs of ones whose exact sum is known (0 f32 16x16x4, 1 f32 32x32x2, 2 f16 16x16x32, 3 f16 32x32x16)
//       xcd / xcdsmall   producer kernel -> consumer kernel on one stream, the consumer block reads what a block on ANOTHER XCD wrote
//               (256 MB / 16 MB per pass, a new pattern every pass)
//   every victim prints the number of mismatching elements it saw; tools/xproc_triage.sh runs the matrix.
#include <hip/hip_runtime.h>

#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../dcscn-super-resolution_amd/csrc/conv3_h.hpp"
#include "../dcscn-super-resolution_amd/csrc/split16_pack.hpp"
#include "../dcscn-super-resolution_amd/csrc/conv_variants.hpp"

using namespace dcscn;

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__host__ __device__ inline unsigned pat(unsigned long long i, unsigned seed) {
    unsigned x = (unsigned)i * 2654435761u ^ (unsigned)(i >> 32) ^ seed;
    x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
    return x;
}

__global__ void fill_pat(unsigned* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = pat(i, seed);
}
__global__ void copy_u4(const uint4* a, uint4* b, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void check_pat(const unsigned* p, size_t n, unsigned seed, unsigned long long* bad) {
    unsigned long long c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += p[i] != pat(i, seed);
    if (c) atomicAdd(bad, c);
}
// producer / consumer pair with a cross-XCD mapping: block b writes chunk b, the check block b reads chunk b + 3 (another XCD wrote it)
__global__ void fill_chunks(unsigned* p, int chunk, unsigned seed) {
    unsigned* q = p + (size_t)blockIdx.x * chunk;
    for (int i = threadIdx.x; i < chunk; i += blockDim.x) q[i] = pat((size_t)blockIdx.x * chunk + i, seed);
}
__global__ void check_chunks(const unsigned* p, int chunk, unsigned seed, unsigned long long* bad) {
    const size_t c = (blockIdx.x + 3) % gridDim.x;
    const unsigned* q = p + c * chunk;
    unsigned long long n = 0;
    for (int i = threadIdx.x; i < chunk; i += blockDim.x) n += q[i] != pat(c * chunk + i, seed);
    if (n) atomicAdd(bad, n);
}
__global__ void diff_u32(const unsigned* a, const unsigned* b, size_t n, unsigned long long* bad) {
    unsigned long long c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
    if (c) atomicAdd(bad, c);
}

// 64 KB of LDS per workgroup holding pat(); filled with ds_write (DMA == 0) or by LDS-DMA behind a counted wait (DMA == 1), then
// re-read and compared `iters` times (a few thousand cycles each) -- the image must survive whatever happens to the wave meanwhile
template <int DMA>
__global__ __launch_bounds__(256) void lds_hold(const unsigned* src, int iters, unsigned seed, unsigned long long* bad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned* mine = src + (size_t)blockIdx.x * 16384;             // 64 KB of the pattern per workgroup
    if constexpr (DMA) {
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
        // 64 pieces of 1 KB, 16 per wave; the wait leaves the last 4 in flight first, as the product kernels' counted waits do
        for (int r = 0; r < 16; ++r) glds16(reinterpret_cast<const char*>(mine) + (size_t)(wave * 16 + r) * 1024, (unsigned)(lane * 16), lds0 + (unsigned)(wave * 16 + r) * 1024u);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        for (int r = 0; r < 16; ++r) reinterpret_cast<uint4*>(smem)[r * 256 + tid] = reinterpret_cast<const uint4*>(mine)[r * 256 + tid];
    }
    __syncthreads();
    unsigned long long c = 0;
    for (int it = 0; it < iters; ++it) {
        for (int r = 0; r < 16; ++r) {
            const u32x4 v = *reinterpret_cast<volatile u32x4*>(smem + (size_t)(r * 256 + tid) * 16);
            const unsigned long long i0 = (unsigned long long)blockIdx.x * 16384 + (size_t)(r * 256 + tid) * 4;
            c += (v.x != pat(i0, seed)) + (v.y != pat(i0 + 1, seed)) + (v.z != pat(i0 + 2, seed)) + (v.w != pat(i0 + 3, seed));
        }
        __builtin_amdgcn_s_sleep(8);
    }
    if (c) atomicAdd(bad, c);
}

// 128 VGPRs of pat() + 16 MFMA accumulators (64 registers) that zero-operand MFMAs must leave unchanged, kept live for `iters` rounds
__global__ __launch_bounds__(256, 1) void regs_hold(const unsigned* src, int iters, unsigned seed, unsigned long long* bad) {
    unsigned v[128];
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 128;
    for (int i = 0; i < 128; ++i) v[i] = src[base + i];
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{(float)(i + 1), (float)(threadIdx.x), 3.0f * i, -1.0f};
    h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    asm volatile("" : "+v"(z));
    for (int it = 0; it < iters; ++it) {
        for (int i = 0; i < 128; ++i) asm volatile("" : "+v"(v[i]));
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(z, z, acc[i], 0, 0, 0);
        __builtin_amdgcn_s_sleep(4);
    }
    unsigned long long c = 0;
    for (int i = 0; i < 128; ++i) c += v[i] != pat(base + i, seed);
    for (int i = 0; i < 16; ++i) c += (acc[i].x != (float)(i + 1)) + (acc[i].y != (float)threadIdx.x) + (acc[i].z != 3.0f * i) + (acc[i].w != -1.0f);
    if (c) atomicAdd(bad, c);
}

// A dependent chain of MFMAs with operands of ones: every instruction adds exactly K to every accumulator element, so after n of them
// the accumulator must be n * K (exact in f32 below 2^24).  A contribution lost anywhere -- e.g. an instruction that was in flight when
// the wave was saved for a context switch -- shows as a smaller value.  V: 0 v_mfma_f32_16x16x4_f32 (8 passes), 1 v_mfma_f32_32x32x2_f32
// (16 passes), 2 v_mfma_f32_16x16x32_f16 (4 passes), 3 v_mfma_f32_32x32x16_f16 (8 passes)
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int V>
__global__ __launch_bounds__(256) void mfma_chain(int iters, unsigned long long* bad, unsigned long long* lost) {
    float one = 1.0f;
    asm volatile("" : "+v"(one));
    h8 oh = {1, 1, 1, 1, 1, 1, 1, 1};
    asm volatile("" : "+v"(oh));
    unsigned long long c = 0, l = 0;
    for (int rep = 0; rep < 4; ++rep) {
        if constexpr (V == 0 || V == 2) {
            f32x4 acc = {0, 0, 0, 0};
            for (int it = 0; it < iters; ++it)
                acc = V == 0 ? __builtin_amdgcn_mfma_f32_16x16x4f32(one, one, acc, 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x32_f16(oh, oh, acc, 0, 0, 0);
            const float want = (float)iters * (V == 0 ? 4.0f : 32.0f);
            for (int i = 0; i < 4; ++i) if (acc[i] != want) { ++c; l += (unsigned long long)((want - acc[i]) / (V == 0 ? 4.0f : 32.0f)); }
        } else {
            f32x16 acc;
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
            for (int it = 0; it < iters; ++it)
                acc = V == 1 ? __builtin_amdgcn_mfma_f32_32x32x2f32(one, one, acc, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_f16(oh, oh, acc, 0, 0, 0);
            const float want = (float)iters * (V == 1 ? 2.0f : 16.0f);
            for (int i = 0; i < 16; ++i) if (acc[i] != want) { ++c; l += (unsigned long long)((want - acc[i]) / (V == 1 ? 2.0f : 16.0f)); }
        }
    }
    if (c) { atomicAdd(bad, c); atomicAdd(lost, l); }
}

// VALU chains with an exactly known result: x <- x * 1 + 1, `iters` times, as v_pk_fma_f32 (PK = 1) or v_fma_f32 (PK = 0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int PK>
__global__ __launch_bounds__(256) void valu_chain(int iters, unsigned long long* bad) {
    float one = 1.0f;
    asm volatile("" : "+v"(one));
    unsigned long long c = 0;
    if constexpr (PK) {
        f32x2 x[4], o = {one, one};
        for (int i = 0; i < 4; ++i) x[i] = f32x2{0.0f, 0.0f};
        for (int it = 0; it < iters; ++it)
            for (int i = 0; i < 4; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(o));
        for (int i = 0; i < 4; ++i) c += (x[i].x != (float)iters) + (x[i].y != (float)iters);
    } else {
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = 0.0f;
        for (int it = 0; it < iters; ++it)
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(one));
        for (int i = 0; i < 8; ++i) c += x[i] != (float)iters;
    }
    if (c) atomicAdd(bad, c);
}
// ONE dependent chain of v_pk_fma_f32 (x <- x * 1 + 1), NOPS wait states between consecutive instructions (hipcc itself puts one,
// `s_nop 0`, between dependent packed-f32 operations: conv_cin1's inner loop)
template <int NOPS>
__global__ __launch_bounds__(256) void pk_dep_chain(int iters, unsigned long long* bad) {
    float one = 1.0f;
    asm volatile("" : "+v"(one));
    f32x2 x = {0.0f, 0.0f}, o = {one, one};
    for (int it = 0; it < iters; ++it) {
        if constexpr (NOPS == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(o));
        else if constexpr (NOPS == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n\ts_nop 0\n\tv_pk_fma_f32 %0, %0, %1, %1\n\ts_nop 0\n\tv_pk_fma_f32 %0, %0, %1, %1\n\ts_nop 0\n\tv_pk_fma_f32 %0, %0, %1, %1\n\ts_nop 0" : "+v"(x) : "v"(o));
        else asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n\ts_nop 7\n\tv_pk_fma_f32 %0, %0, %1, %1\n\ts_nop 7\n\tv_pk_fma_f32 %0, %0, %1, %1\n\ts_nop 7\n\tv_pk_fma_f32 %0, %0, %1, %1\n\ts_nop 7" : "+v"(x) : "v"(o));
    }
    const float want = 4.0f * iters;
    const unsigned long long c = (x.x != want) + (x.y != want);
    if (c) atomicAdd(bad, c);
}
// small-LDS workgroups at high occupancy (conv_cin1's shape: 1.6 KB per workgroup): write a pattern, barrier, read it back many times
__global__ __launch_bounds__(256) void lds_small(int iters, unsigned seed, unsigned long long* bad) {
    __shared__ unsigned sm[412];
    for (int i = threadIdx.x; i < 412; i += 256) sm[i] = pat((size_t)blockIdx.x * 412 + i, seed);
    __syncthreads();
    unsigned long long c = 0;
    for (int it = 0; it < iters; ++it) {
        const int i = (threadIdx.x * 7 + it * 13) % 412;
        c += reinterpret_cast<volatile unsigned*>(sm)[i] != pat((size_t)blockIdx.x * 412 + i, seed);
    }
    if (c) atomicAdd(bad, c);
}

// bare MFMA stream (the aggressor `mfma`): 8 rotating accumulators, ~1 ms per launch
__global__ __launch_bounds__(256) void mfma_stream(float* out, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    h8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {(_Float16)0.5f, 1, 2, 1, 1, 3, 1, 1};
    asm volatile("" : "+v"(a), "+v"(b));
    for (int it = 0; it < iters; ++it)
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 12345.678f) out[0] = s;
}

// a CNN2-shaped conv3_h<6> launch on random data (aggressor `c3h`, victim `c3h`)
struct C3 {
    ConvArgs a{};
    dim3 grid;
    float *d_in, *d_out;
    size_t out_floats;
    void init(int N) {
        const int cin = 196, cout = 166, H = 48, W = 48, cin_phys = 196, in_stride = 196, out_stride = 168;
        const size_t in_floats = (size_t)N * H * W * in_stride;
        out_floats = (size_t)N * H * W * out_stride;
        CK(hipMalloc(&d_in, in_floats * 4 + 256)); CK(hipMalloc(&d_out, out_floats * 4));
        std::vector<float> hin(in_floats);
        for (size_t i = 0; i < in_floats; ++i) hin[i] = ((int)(pat(i, 7u) >> 8 & 0xffff) - 32768) * (200.0f / 32768.0f);
        CK(hipMemcpy(d_in, hin.data(), in_floats * 4, hipMemcpyHostToDevice));
        CK(hipMemset(d_out, 0, out_floats * 4));
        const int ng = 2, nt = 6, nfull = 1, n_chunks = 7, ctot = ng * nt * 16;
        std::vector<float> dense((size_t)9 * n_chunks * 32 * ctot, 0.0f), bp(ctot, 0.1f), ap(ctot, 0.2f);
        for (int t = 0; t < 9; ++t)
            for (int c = 0; c < cin; ++c)
                for (int o = 0; o < cout; ++o) {
                    const int tl = o / 16, wide = nfull * nt;
                    const int g = tl < wide ? tl / nt : nfull + (tl - wide) / (nt - 1), tg = tl < wide ? tl % nt : (tl - wide) % (nt - 1);
                    dense[((size_t)t * n_chunks * 32 + c) * ctot + (g * nt + tg) * 16 + o % 16] = ((int)(pat(((size_t)t * cin + c) * cout + o, 9u) & 0xffff) - 32768) * (0.05f / 32768.0f);
                }
        const int e = split16_scale_exp(dense.data(), dense.size());
        const int octs = c3h_tail_octs(cin_phys);
        std::vector<uint16_t> p16 = pack_conv16(dense, 9, n_chunks * 32, ctot, ng, nt, n_chunks, e, octs);
        void* d_p16; float *d_bp, *d_ap; int* d_redo;
        CK(hipMalloc(&d_p16, p16.size() * 2)); CK(hipMalloc(&d_bp, ctot * 4)); CK(hipMalloc(&d_ap, ctot * 4));
        CK(hipMemcpy(d_p16, p16.data(), p16.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_bp, bp.data(), ctot * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_ap, ap.data(), ctot * 4, hipMemcpyHostToDevice));
        a.in = d_in; a.in_stride = in_stride; a.in_off = 0; a.cin_phys = cin_phys; a.act = ACT_ALPHA;
        a.N = N; a.H = H; a.W = W; a.split = 1 << 30; a.ps = 1; a.ps_c = 1; a.vec4 = 1; a.res = nullptr; a.res_stride = 1;
        a.tiles_x = 3; a.tiles_y = 3;
        CK(hipMalloc(&d_redo, (size_t)N * 9 * 4)); CK(hipMemset(d_redo, 0, (size_t)N * 9 * 4));
        a.wpack16 = d_p16; a.inv_scale = std::ldexp(1.0f, -e); a.n_chunks = n_chunks; a.n_full = nfull; a.bias = d_bp; a.alpha = d_ap; a.redo = d_redo; a.tail_octs = octs;
        a.out0 = OutDesc{d_out, out_stride, 0, out_stride}; a.out1 = a.out0;
        a.n_groups = ng; a.group_span = ng;
        grid = dim3((unsigned)((((size_t)N * 9 + 7) / 8) * 8 * ng));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_h<6>), hipFuncAttributeMaxDynamicSharedMemorySize, C3HGeom<6>::LDS_BYTES));
    }
    void launch() { hipLaunchKernelGGL((conv3_h<6>), grid, dim3(256), C3HGeom<6>::LDS_BYTES, 0, a); }
};

// a 3x3 layer on conv_igemm (f32 MFMA, register-staged LDS, no LDS-DMA): 32 -> 32 channels, random finite data and filters
struct IG {
    ConvArgs a{};
    float* d_out;
    size_t out_floats;
    using V = Variant<3, 2>;
    void init(int N) {
        const int H = 48, W = 48, cin = 32, cs = 32;
        const size_t in_floats = (size_t)N * H * W * cin;
        out_floats = (size_t)N * H * W * cs;
        float *d_in, *d_w, *d_b;
        CK(hipMalloc(&d_in, in_floats * 4 + 256)); CK(hipMalloc(&d_out, out_floats * 4));
        std::vector<float> hin(in_floats);
        for (size_t i = 0; i < in_floats; ++i) hin[i] = ((int)(pat(i, 17u) >> 8 & 0xffff) - 32768) * (100.0f / 32768.0f);
        CK(hipMemcpy(d_in, hin.data(), in_floats * 4, hipMemcpyHostToDevice));
        using G = ConvGeom<3, V::MT, 2, V::KC>;
        const int n_chunks = cin / V::KC;
        std::vector<float> w((size_t)n_chunks * G::B_FLOATS), b(64, 0.25f);
        for (size_t i = 0; i < w.size(); ++i) w[i] = ((int)(pat(i, 23u) & 0xffff) - 32768) * (0.1f / 32768.0f);
        CK(hipMalloc(&d_w, w.size() * 4)); CK(hipMalloc(&d_b, 64 * 4));
        CK(hipMemcpy(d_w, w.data(), w.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_b, b.data(), 64 * 4, hipMemcpyHostToDevice));
        a.in = d_in; a.in_stride = cin; a.in_off = 0; a.cin_phys = cin; a.n_chunks = n_chunks; a.wpack = d_w; a.bias = d_b; a.alpha = d_b; a.act = ACT_ALPHA;
        a.N = N; a.H = H; a.W = W; a.tiles_x = 3; a.tiles_y = (H + 4 * V::MT - 1) / (4 * V::MT);
        a.out0 = OutDesc{d_out, cs, 0, cs}; a.out1 = a.out0; a.split = 1 << 30; a.ps = 1; a.ps_c = 1; a.vec4 = 1; a.res = nullptr; a.res_stride = 1;
        CK(V::set_attr());
    }
    void launch() { CK(V::launch(a, 1, 0)); }
};

// conv_cin1<3>'s arithmetic (kernels.hip) with the accumulation written on float4 vectors (PK = 1: hipcc emits v_pk_fma_f32 with op_sel
// broadcasts, as in the product kernel) or on four separate floats (PK = 0: v_fma_f32 / v_fmac_f32 only)
template <int PK>
__global__ __launch_bounds__(256) void cin1_local(const Cin1Args a, int tpp_log2) {
    constexpr int TAPS = 9, T = 16, HT = 18;
    extern __shared__ __attribute__((aligned(16))) float smem_l[];
    const int cs = a.cs, c4n = cs >> 2;
    float* xs = smem_l;
    float* ws = smem_l + ((HT * HT + 3) & ~3);
    float* bs = ws + TAPS * cs;
    float* as = bs + cs;
    int bid = blockIdx.x;
    const int tiles_x = (a.W + T - 1) / T, tiles_y = (a.H + T - 1) / T;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int img = bid / tiles_y;
    const int y0 = ty * T, x0 = tx * T, tid = threadIdx.x;
    const float* xin = a.x + (size_t)img * a.H * a.W;
    for (int i = tid; i < HT * HT; i += 256) {
        const int hy = i / HT, hx = i - hy * HT, gy = y0 + hy - 1, gx = x0 + hx - 1;
        xs[i] = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? xin[(size_t)gy * a.W + gx] : 0.0f;
    }
    for (int i = tid; i < TAPS * cs; i += 256) ws[i] = a.w[i];
    for (int i = tid; i < cs; i += 256) { bs[i] = a.bias[i]; as[i] = a.alpha[i]; }
    __syncthreads();
    const int tpp = 1 << tpp_log2, cl = tid & (tpp - 1), pl = tid >> tpp_log2, ppi = 256 >> tpp_log2;
    for (int c4 = cl; c4 < c4n; c4 += tpp) {
        f32x4 wr[TAPS];
#pragma unroll
        for (int t = 0; t < TAPS; ++t) wr[t] = *reinterpret_cast<const f32x4*>(ws + t * cs + 4 * c4);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bs + 4 * c4), av = *reinterpret_cast<const f32x4*>(as + 4 * c4);
        for (int p = pl; p < T * T; p += ppi) {
            const int py = p >> 4, px = p & 15, gy = y0 + py, gx = x0 + px;
            if (gy >= a.H || gx >= a.W) continue;
            f32x4 v;
            if constexpr (PK) {
                f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int t = 0; t < TAPS; ++t) s += wr[t] * xs[(py + t / 3) * HT + px + t % 3];
                v = bv + s;
            } else {
                float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    float xv = xs[(py + t / 3) * HT + px + t % 3];
                    asm volatile("" : "+v"(xv));               // one value, four separate FMAs (no packing)
                    s0 = __builtin_fmaf(wr[t].x, xv, s0); s1 = __builtin_fmaf(wr[t].y, xv, s1);
                    s2 = __builtin_fmaf(wr[t].z, xv, s2); s3 = __builtin_fmaf(wr[t].w, xv, s3);
                }
                v = f32x4{bv.x + s0, bv.y + s1, bv.z + s2, bv.w + s3};
            }
            v.x = v.x > 0.0f ? v.x : av.x * v.x; v.y = v.y > 0.0f ? v.y : av.y * v.y;
            v.z = v.z > 0.0f ? v.z : av.z * v.z; v.w = v.w > 0.0f ? v.w : av.w * v.w;
            const size_t pix = ((size_t)img * a.H + gy) * a.W + gx;
            *reinterpret_cast<f32x4*>(a.out.ptr + pix * a.out.stride + a.out.off + 4 * c4) = v;
        }
    }
}

// the first feature layer alone: conv_cin1 (1 -> cs channels, VALU + LDS only, no MFMA, no DMA) through the product library's launcher
struct C1 {
    Cin1Args a{};
    float* d_out;
    size_t out_floats;
    void init(int N, int cs) {
        const int H = 48, W = 48;
        out_floats = (size_t)N * H * W * cs;
        float *d_x, *d_w, *d_b;
        CK(hipMalloc(&d_x, (size_t)N * H * W * 4)); CK(hipMalloc(&d_out, out_floats * 4));
        std::vector<float> hx((size_t)N * H * W), w(11 * cs);
        for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)(pat(i, 31u) & 0xffff) * (255.0f / 65535.0f);
        for (size_t i = 0; i < w.size(); ++i) w[i] = ((int)(pat(i, 37u) & 0xffff) - 32768) * (0.5f / 32768.0f);
        CK(hipMalloc(&d_w, w.size() * 4)); CK(hipMemcpy(d_w, w.data(), w.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        d_b = d_w + 9 * cs;
        a.x = d_x; a.w = d_w; a.bias = d_b; a.alpha = d_b + cs; a.act = ACT_ALPHA; a.ks = 3; a.N = N; a.H = H; a.W = W; a.cs = cs;
        a.out = OutDesc{d_out, cs, 0, cs};
    }
    int local = -1;                                      // 0 / 1: the local copy without / with packed f32 arithmetic instead of the library's kernel
    void launch() {
        if (local < 0) { CK(cin1_launch(a, 0)); return; }
        int tpp_log2 = 0;
        while ((1 << tpp_log2) < a.cs / 4 && tpp_log2 < 6) ++tpp_log2;
        const size_t lds = (size_t)(324 + 11 * a.cs) * 4;
        const dim3 grid((unsigned)(a.N * 9));
        if (local) hipLaunchKernelGGL((cin1_local<1>), grid, dim3(256), lds, 0, a, tpp_log2);
        else hipLaunchKernelGGL((cin1_local<0>), grid, dim3(256), lds, 0, a, tpp_log2);
    }
};

int main(int argc, char** argv) {
    if (argc < 4) { printf("usage: xproc_triage <aggressor|victim> <kind> <seconds>\n"); return 2; }
    const bool victim = !strcmp(argv[1], "victim");
    const char* kind = argv[2];
    const double secs = atof(argv[3]);
    const unsigned seed = 0x5eed0000u + (unsigned)getpid();
    unsigned long long* d_bad;
    CK(hipMalloc(&d_bad, 8)); CK(hipMemset(d_bad, 0, 8));
    const size_t n = 64u << 20;                                  // 256 MB of pattern
    unsigned *d_a = nullptr, *d_b = nullptr;
    auto need_ab = [&]() {
        CK(hipMalloc(&d_a, n * 4)); CK(hipMalloc(&d_b, n * 4));
        hipLaunchKernelGGL(fill_pat, dim3(4096), dim3(256), 0, 0, d_a, n, seed);
        CK(hipDeviceSynchronize());
    };
    long long launches = 0;
    const double t0 = now_s();
    if (!strcmp(kind, "copy") || !strcmp(kind, "memcpy")) {
        need_ab();
        while (now_s() - t0 < secs) {
            for (int i = 0; i < 8; ++i) {
                CK(hipMemsetAsync(d_b, 0, n * 4, 0));
                if (!strcmp(kind, "copy")) hipLaunchKernelGGL(copy_u4, dim3(8192), dim3(256), 0, 0, reinterpret_cast<const uint4*>(d_a), reinterpret_cast<uint4*>(d_b), n / 4);
                else CK(hipMemcpyAsync(d_b, d_a, n * 4, hipMemcpyDeviceToDevice, 0));
                if (victim) hipLaunchKernelGGL(check_pat, dim3(8192), dim3(256), 0, 0, d_b, n, seed, d_bad);
                ++launches;
            }
            CK(hipDeviceSynchronize());
        }
    } else if (!strcmp(kind, "xcd") || !strcmp(kind, "xcdsmall")) {
        // every launch pair writes a NEW pattern; the check reads what blocks on other XCDs wrote in the kernel before it (same stream)
        const bool small = !strcmp(kind, "xcdsmall");
        const int nb = small ? 2048 : 4096, chunk = small ? 2048 : 16384;       // 16 MB (stays in the L2s) / 256 MB
        CK(hipMalloc(&d_b, (size_t)nb * chunk * 4));
        unsigned it = 0;
        while (now_s() - t0 < secs) {
            for (int i = 0; i < 8; ++i) {
                ++it;
                hipLaunchKernelGGL(fill_chunks, dim3(nb), dim3(256), 0, 0, d_b, chunk, seed + it);
                hipLaunchKernelGGL(check_chunks, dim3(nb), dim3(256), 0, 0, d_b, chunk, seed + it, d_bad);
                ++launches;
            }
            CK(hipDeviceSynchronize());
        }
    } else if (!strcmp(kind, "lds") || !strcmp(kind, "ldsdma")) {
        need_ab();
        const bool dma = !strcmp(kind, "ldsdma");
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_hold<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_hold<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        while (now_s() - t0 < secs) {
            for (int i = 0; i < 8; ++i) {
                if (dma) hipLaunchKernelGGL((lds_hold<1>), dim3(2048), dim3(256), 65536, 0, d_a, 100, seed, d_bad);
                else hipLaunchKernelGGL((lds_hold<0>), dim3(2048), dim3(256), 65536, 0, d_a, 100, seed, d_bad);
                ++launches;
            }
            CK(hipDeviceSynchronize());
        }
    } else if (!strcmp(kind, "regs")) {
        need_ab();
        while (now_s() - t0 < secs) {
            for (int i = 0; i < 8; ++i) { hipLaunchKernelGGL(regs_hold, dim3(1024), dim3(256), 0, 0, d_a, 2000, seed, d_bad); ++launches; }
            CK(hipDeviceSynchronize());
        }
    } else if (!strncmp(kind, "pkdep", 5) || !strncmp(kind, "samepkdep", 9)) {
        // samepkdep*: the MFMA stream runs in THIS process on a second stream (no second process needed)
        const bool same = kind[0] == 's';
        const int v = kind[same ? 9 : 5] - '0';
        hipStream_t s2 = nullptr;
        float* d_o = nullptr;
        if (same) { CK(hipStreamCreate(&s2)); CK(hipMalloc(&d_o, 64)); }
        while (now_s() - t0 < secs) {
            if (same) hipLaunchKernelGGL(mfma_stream, dim3(2048), dim3(256), 0, s2, d_o, 20000);
            for (int i = 0; i < 8; ++i) {
                if (v == 0) hipLaunchKernelGGL((pk_dep_chain<0>), dim3(4096), dim3(256), 0, 0, 20000, d_bad);
                else if (v == 1) hipLaunchKernelGGL((pk_dep_chain<1>), dim3(4096), dim3(256), 0, 0, 20000, d_bad);
                else hipLaunchKernelGGL((pk_dep_chain<2>), dim3(4096), dim3(256), 0, 0, 20000, d_bad);
                ++launches;
            }
            CK(hipDeviceSynchronize());
        }
    } else if (!strcmp(kind, "pkfma") || !strcmp(kind, "fma") || !strcmp(kind, "ldssmall")) {
        while (now_s() - t0 < secs) {
            for (int i = 0; i < 8; ++i) {
                if (!strcmp(kind, "pkfma")) hipLaunchKernelGGL((valu_chain<1>), dim3(4096), dim3(256), 0, 0, 4000, d_bad);
                else if (!strcmp(kind, "fma")) hipLaunchKernelGGL((valu_chain<0>), dim3(4096), dim3(256), 0, 0, 4000, d_bad);
                else hipLaunchKernelGGL(lds_small, dim3(8192), dim3(256), 0, 0, 2000, seed, d_bad);
                ++launches;
            }
            CK(hipDeviceSynchronize());
        }
    } else if (!strncmp(kind, "chain", 5)) {
        const int v = kind[5] ? kind[5] - '0' : 0;
        unsigned long long* d_lost;
        CK(hipMalloc(&d_lost, 8)); CK(hipMemset(d_lost, 0, 8));
        while (now_s() - t0 < secs) {
            for (int i = 0; i < 8; ++i) {
                const int iters = 20000;
                if (v == 0) hipLaunchKernelGGL((mfma_chain<0>), dim3(1024), dim3(256), 0, 0, iters, d_bad, d_lost);
                else if (v == 1) hipLaunchKernelGGL((mfma_chain<1>), dim3(1024), dim3(256), 0, 0, iters, d_bad, d_lost);
                else if (v == 2) hipLaunchKernelGGL((mfma_chain<2>), dim3(1024), dim3(256), 0, 0, iters, d_bad, d_lost);
                else hipLaunchKernelGGL((mfma_chain<3>), dim3(1024), dim3(256), 0, 0, iters, d_bad, d_lost);
                ++launches;
            }
            CK(hipDeviceSynchronize());
        }
        unsigned long long lost = 0;
        CK(hipMemcpy(&lost, d_lost, 8, hipMemcpyDeviceToHost));
        printf("  [%s: MFMA contributions missing from the accumulators, summed over elements: %llu]\n", kind, lost);
    } else if (!strcmp(kind, "mfma")) {
        float* d_o;
        CK(hipMalloc(&d_o, 64));
        while (now_s() - t0 < secs) {
            for (int i = 0; i < 8; ++i) { hipLaunchKernelGGL(mfma_stream, dim3(2048), dim3(256), 0, 0, d_o, 20000); ++launches; }
            CK(hipDeviceSynchronize());
        }
    } else if (!strncmp(kind, "cin1", 4)) {
        C1 c;
        // cin1<cs>: the library's kernel; cin1p<cs> / cin1s<cs>: the local copy with packed / scalar f32 arithmetic
        const int skip = kind[4] == 'p' || kind[4] == 's' ? 5 : 4;
        c.local = kind[4] == 'p' ? 1 : kind[4] == 's' ? 0 : -1;
        c.init(96, kind[skip] ? atoi(kind + skip) : 8);
        c.launch();
        CK(hipDeviceSynchronize());
        float* d_first;
        CK(hipMalloc(&d_first, c.out_floats * 4));
        CK(hipMemcpy(d_first, c.d_out, c.out_floats * 4, hipMemcpyDeviceToDevice));
        bool shown = false;
        std::vector<float> h0(c.out_floats), h1(c.out_floats);
        CK(hipMemcpy(h0.data(), d_first, c.out_floats * 4, hipMemcpyDeviceToHost));
        while (now_s() - t0 < secs) {
            for (int i = 0; i < 8; ++i) {
                c.launch();
                hipLaunchKernelGGL(diff_u32, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const unsigned*>(d_first), reinterpret_cast<const unsigned*>(c.d_out), c.out_floats, d_bad);
                ++launches;
                if (!shown) {                                  // where and what: the first launch that differs is listed
                    unsigned long long b = 0;
                    CK(hipMemcpy(&b, d_bad, 8, hipMemcpyDeviceToHost));
                    if (b) {
                        shown = true;
                        CK(hipMemcpy(h1.data(), c.d_out, c.out_floats * 4, hipMemcpyDeviceToHost));
                        const int cs = c.a.cs;
                        int n = 0; size_t total = 0;
                        for (size_t e = 0; e < c.out_floats; ++e) total += memcmp(&h0[e], &h1[e], 4) != 0;
                        printf("  first differing launch: %zu elements differ; (img, y, x, c): first run -> this run\n", total);
                        for (size_t e = 0; e < c.out_floats && n < 40; ++e)
                            if (memcmp(&h0[e], &h1[e], 4)) {
                                const size_t px = e / cs;
                                unsigned u0, u1; memcpy(&u0, &h0[e], 4); memcpy(&u1, &h1[e], 4);
                                printf("    (%3zu, %2zu, %2zu, %3zu): %12.6g -> %12.6g   %08x -> %08x\n", px / 2304, px % 2304 / 48, px % 48, e % cs, h0[e], h1[e], u0, u1);
                                ++n;
                            }
                    }
                }
            }
            CK(hipDeviceSynchronize());
        }
    } else if (!strcmp(kind, "igemm")) {
        IG c;
        c.init(256);
        c.launch();
        CK(hipDeviceSynchronize());
        float* d_first;
        CK(hipMalloc(&d_first, c.out_floats * 4));
        CK(hipMemcpy(d_first, c.d_out, c.out_floats * 4, hipMemcpyDeviceToDevice));
        while (now_s() - t0 < secs) {
            for (int i = 0; i < 8; ++i) {
                c.launch();
                hipLaunchKernelGGL(diff_u32, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const unsigned*>(d_first), reinterpret_cast<const unsigned*>(c.d_out), c.out_floats, d_bad);
                ++launches;
            }
            CK(hipDeviceSynchronize());
        }
    } else if (!strcmp(kind, "c3h")) {
        C3 c;
        c.init(256);
        c.launch();
        CK(hipDeviceSynchronize());
        float* d_first;
        CK(hipMalloc(&d_first, c.out_floats * 4));
        CK(hipMemcpy(d_first, c.d_out, c.out_floats * 4, hipMemcpyDeviceToDevice));
        while (now_s() - t0 < secs) {
            for (int i = 0; i < 8; ++i) {
                c.launch();
                if (victim) hipLaunchKernelGGL(diff_u32, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const unsigned*>(d_first), reinterpret_cast<const unsigned*>(c.d_out), c.out_floats, d_bad);
                ++launches;
            }
            CK(hipDeviceSynchronize());
        }
    } else { printf("unknown kind %s\n", kind); return 2; }
    unsigned long long bad = 0;
    CK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
    printf("%s %-7s pid %d: %lld launches in %.1f s", argv[1], kind, (int)getpid(), launches, now_s() - t0);
    if (victim) printf(", %llu MISMATCHING elements%s", bad, bad ? "  <-- CORRUPTED" : "");
    printf("\n");
    return 0;
}
