// Correctness + timing harness for the split16 kernels (f32-accurate contraction on the f16 matrix pipe) against the f32
// kernels they stand in for and a float64 naive reference.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/h16_tune.hip -o tools/h16_tune
//   tools/h16_tune nin      A1 || B1 of the bench model (1316 -> 96 over 1024 patches): conv_nin vs conv_nin_h, one tensor / 12 sources,
//                           ragged pixel counts and channel tails, and the overflow -> redo -> f32 fallback path
//   tools/h16_tune conv3    the 3x3 layers of the bench model: conv_wino2 (f32 Winograd) vs conv3_h, edge cases, fallback
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <vector>

#include "../dcscn-super-resolution_amd/csrc/conv_nin_h.hpp"
#include "../dcscn-super-resolution_amd/csrc/split16_pack.hpp"
#ifdef H16_CONV3
#include "../dcscn-super-resolution_amd/csrc/conv3_h.hpp"
#ifdef H16_H8
#include "../dcscn-super-resolution_amd/csrc/conv3_h8.hpp"
#endif
#endif

using namespace dcscn;

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

static std::vector<float> rand_vec(size_t n, unsigned seed, float scale) {
    std::vector<float> h(n);
    unsigned s = seed;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (((s >> 8) & 0xffff) / 65536.0f - 0.5f) * scale; }
    return h;
}

// activations like the net's: ~N(0, 50)-ish (sum of uniforms) through PReLU(0.2), a sprinkle of tiny and of large values
__global__ void fill_act(float* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned s = (unsigned)(i * 2654435761u) ^ seed;
        float acc = 0.0f;
        for (int k = 0; k < 4; ++k) { s = s * 1664525u + 1013904223u; acc += ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
        float v = acc * 90.0f;
        s = s * 1664525u + 1013904223u;
        const unsigned sel = (s >> 10) & 1023;
        if (sel == 0) v *= 8.0f;            // up to ~1400
        else if (sel < 8) v *= 1e-4f;       // tiny values: lo pieces go subnormal
        p[i] = v > 0.0f ? v : 0.2f * v;
#ifdef H16_CONST_ACT                        // every activation the same value: what the clock does when the operands do not toggle
        p[i] = 1.0f;
#endif
    }
}

// naive 1x1 conv on sampled pixels: the K axis is a list of sources (pointer, pixel stride in floats, channels)
struct Src { const float* p; int stride; int ch; };
struct SrcList { Src s[16]; int n; };
__global__ void naive_nin(SrcList srcs, const float* w, int cin, int cout, const float* bias, const float* alpha, long long npix, long long step,
                          float* out, int out_stride) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nsamp = (npix + step - 1) / step;
    if (idx >= nsamp * cout) return;
    const int co = (int)(idx % cout);
    const long long p = (idx / cout) * step;
    double s = 0.0;
    int k = 0;
    for (int i = 0; i < srcs.n; ++i) {
        const float* ip = srcs.s[i].p + (size_t)p * srcs.s[i].stride;
        for (int c = 0; c < srcs.s[i].ch; ++c, ++k) s += (double)ip[c] * (double)w[(size_t)k * cout + co];
    }
    float v = (float)(s + bias[co]);
    v = v > 0.0f ? v : alpha[co] * v;
    out[(size_t)p * out_stride + co] = v;
}

template <typename K>
static float time_kernel(K kern, dim3 grid, size_t lds, const ConvArgs& a, int reps = 5) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, a);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, a);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    CK(hipGetLastError());
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return best;
}

// even spread of 16-channel tiles over groups of at most max_nt (api: finalize_op)
static void group_plan(int cout, int max_nt, int* n_groups, int* nt, int* n_full) {
    const int tiles16 = (cout + 15) / 16;
    *n_groups = (tiles16 + max_nt - 1) / max_nt;
    *nt = (tiles16 + *n_groups - 1) / *n_groups;
    *n_full = tiles16 - *n_groups * (*nt - 1);
}
static int padded_col(int cc, int nt, int n_full) {
    const int t = cc / 16, wide = n_full * nt;
    const int g = t < wide ? t / nt : n_full + (t - wide) / (nt - 1);
    const int tg = t < wide ? t % nt : (t - wide) % (nt - 1);
    return (g * nt + tg) * 16 + cc % 16;
}

// ----------------------------------------------------------------------------------------------------------------------
// conv_nin vs conv_nin_h
// ----------------------------------------------------------------------------------------------------------------------
static int g_S = 3;       // input stages of conv_nin_h: argv "nin 2" / "nin 3"
struct NinCase { const char* name; long long npix; std::vector<int> widths; int cout; bool multi; };

static int run_nin(const NinCase& C, bool timing, int overflow_test) {
    // sources: dense tensors of pad4(width) channels, or one tensor holding all slices (multi == false)
    const int nsrc = (int)C.widths.size();
    int cin = 0, cin_phys = 0;
    for (int wd : C.widths) { cin += wd; cin_phys += (wd + 3) & ~3; }
    const long long npix = C.npix;
    float* d_one = nullptr;
    std::vector<float*> d_src(nsrc);
    SrcList sl{};
    sl.n = nsrc;
    std::vector<int> off(nsrc);
    if (!C.multi) {
        CK(hipMalloc(&d_one, (size_t)npix * cin_phys * sizeof(float) + 256));
        hipLaunchKernelGGL(fill_act, dim3(4096), dim3(256), 0, 0, d_one, (size_t)npix * cin_phys, 17u);
        int o = 0;
        for (int i = 0; i < nsrc; ++i) { off[i] = o; sl.s[i] = Src{d_one + o, cin_phys, C.widths[i]}; o += (C.widths[i] + 3) & ~3; }
    } else {
        for (int i = 0; i < nsrc; ++i) {
            const int st = (C.widths[i] + 3) & ~3;
            CK(hipMalloc(&d_src[i], (size_t)npix * st * sizeof(float) + 256));
            hipLaunchKernelGGL(fill_act, dim3(4096), dim3(256), 0, 0, d_src[i], (size_t)npix * st, 17u + i);
            sl.s[i] = Src{d_src[i], st, C.widths[i]};
        }
    }
    CK(hipDeviceSynchronize());
    // logical channel k -> physical channel kp of the virtual concat (each source padded to 4)
    std::vector<int> kmap(cin);
    {
        int k = 0, kp = 0;
        for (int i = 0; i < nsrc; ++i) { for (int c = 0; c < C.widths[i]; ++c) kmap[k++] = kp + c; kp += (C.widths[i] + 3) & ~3; }
    }
    const int cout = C.cout;
    std::vector<float> w = rand_vec((size_t)cin * cout, 4242 + cin, 2.0f * std::sqrt(6.0f / cin));
    std::vector<float> bias = rand_vec(cout, 5, 0.4f), alpha = rand_vec(cout, 6, 0.25f);
    for (auto& v : alpha) v += 0.175f;
    int ng, nt, nfull;
    group_plan(cout, kNinMaxNT, &ng, &nt, &nfull);
    const int n_chunks = (cin_phys + 15) / 16, ctot = ng * nt * 16, ns = conv_ns(nt);
    std::vector<float> dense((size_t)n_chunks * 16 * ctot, 0.0f), bp(ctot, 0.0f), ap(ctot, 0.0f);
    for (int k = 0; k < cin; ++k)
        for (int co = 0; co < cout; ++co) dense[(size_t)kmap[k] * ctot + padded_col(co, nt, nfull)] = w[(size_t)k * cout + co];
    for (int co = 0; co < cout; ++co) { bp[padded_col(co, nt, nfull)] = bias[co]; ap[padded_col(co, nt, nfull)] = alpha[co]; }
    // f32 image of conv_nin: [group][chunk][(c & 3) * 4 + (c >> 2)][NS]
    std::vector<float> p32((size_t)ng * n_chunks * 16 * ns, 0.0f);
    for (int kp = 0; kp < n_chunks * 16; ++kp)
        for (int pc = 0; pc < ctot; ++pc) {
            const int chunk = kp / 16, c16 = kp % 16, row = (c16 & 3) * 4 + (c16 >> 2), grp = pc / (nt * 16), jn = pc % (nt * 16);
            p32[((size_t)grp * n_chunks + chunk) * 16 * ns + (size_t)row * ns + jn] = dense[(size_t)kp * ctot + pc];
        }
    const int e = split16_scale_exp(dense.data(), dense.size());
    const int n_chunks32 = (cin_phys + 31) / 32;              // conv_nin_h2: 32-channel chunks, pack_conv16 image with one tap
    std::vector<uint16_t> p16b = pack_conv16(dense, 1, n_chunks * 16, ctot, ng, nt, n_chunks32, e);
    void* d_p16b;
    CK(hipMalloc(&d_p16b, p16b.size() * 2));
    CK(hipMemcpy(d_p16b, p16b.data(), p16b.size() * 2, hipMemcpyHostToDevice));
    float *d_w, *d_p32, *d_bias, *d_alpha, *d_bp, *d_ap, *d_ref, *d_o32, *d_o16;
    int* d_redo;
    const int out_stride = (cout + 3) & ~3;
    const long long nblocks = (npix + 255) / 256;
    CK(hipMalloc(&d_w, w.size() * 4)); CK(hipMalloc(&d_p32, p32.size() * 4)); 
    CK(hipMalloc(&d_bias, cout * 4)); CK(hipMalloc(&d_alpha, cout * 4)); CK(hipMalloc(&d_bp, ctot * 4)); CK(hipMalloc(&d_ap, ctot * 4));
    CK(hipMalloc(&d_ref, (size_t)npix * out_stride * 4)); CK(hipMalloc(&d_o32, (size_t)npix * out_stride * 4)); CK(hipMalloc(&d_o16, (size_t)npix * out_stride * 4));
    CK(hipMalloc(&d_redo, nblocks * 4));
    CK(hipMemcpy(d_w, w.data(), w.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_p32, p32.data(), p32.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_bias, bias.data(), cout * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_alpha, alpha.data(), cout * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_bp, bp.data(), ctot * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_ap, ap.data(), ctot * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_ref, 0, (size_t)npix * out_stride * 4)); CK(hipMemset(d_o32, 0, (size_t)npix * out_stride * 4)); CK(hipMemset(d_o16, 0, (size_t)npix * out_stride * 4));
    CK(hipMemset(d_redo, 0, nblocks * 4));

    // source table (api.hip: densify_features): one entry per 16-byte quad of the virtual concat
    NinSrcQuad* d_tab = nullptr;
    if (C.multi) {
        std::vector<NinSrcQuad> tab((size_t)n_chunks32 * 8, NinSrcQuad{0, 0, 0});   // >= n_chunks * 4: both kernels read this table
        int q = 0;
        for (int i = 0; i < nsrc; ++i) {
            const int st = (C.widths[i] + 3) & ~3;
            for (int j = 0; j < st / 4; ++j, ++q) tab[q] = NinSrcQuad{(unsigned long long)(uintptr_t)(d_src[i] + 4 * j), (unsigned)(st * 4), 1u};
        }
        for (; q < n_chunks32 * 8; ++q) tab[q] = NinSrcQuad{(unsigned long long)(uintptr_t)d_src[0], 0u, 0u};
        CK(hipMalloc(&d_tab, tab.size() * sizeof(NinSrcQuad)));
        CK(hipMemcpy(d_tab, tab.data(), tab.size() * sizeof(NinSrcQuad), hipMemcpyHostToDevice));
    }
    long long poke_pix = -1;
    if (overflow_test) {                       // one activation beyond the f16 range
        poke_pix = npix / 3;
        const float big = 1.0e5f;
        float* dst = C.multi ? d_src[nsrc / 2] + (size_t)poke_pix * (((C.widths[nsrc / 2] + 3) & ~3)) + 1 : d_one + (size_t)poke_pix * cin_phys + 5;
        CK(hipMemcpy(dst, &big, 4, hipMemcpyHostToDevice));
    }

    const long long step = timing ? 97 : 1;
    {
        const long long nsamp = (npix + step - 1) / step;
        hipLaunchKernelGGL(naive_nin, dim3((unsigned)((nsamp * cout + 255) / 256)), dim3(256), 0, 0, sl, d_w, cin, cout, d_bias, d_alpha, npix, step, d_ref, out_stride);
        CK(hipDeviceSynchronize());
    }
    ConvArgs a{};
    a.in = C.multi ? d_src[0] : d_one; a.in_stride = cin_phys; a.in_off = 0; a.cin_phys = cin_phys; a.n_chunks = n_chunks;
    a.wpack = d_p32; a.bias = d_bp; a.alpha = d_ap; a.act = ACT_ALPHA;
    a.N = 1; a.H = 1; a.W = (int)npix;
    a.n_full = nfull; a.split = 1 << 30; a.ps = 1; a.ps_c = 1; a.vec4 = 1;
    a.srctab = d_tab;
    a.wpack16 = d_p16b; a.inv_scale = std::ldexp(1.0f, -e); a.redo = d_redo; a.redo_check = 0;
    const dim3 grid((unsigned)nblocks, (unsigned)ng);
    float ms32 = 0, ms16 = 0;
    const size_t tab_bytes = C.multi ? (size_t)n_chunks * 64 : 0;
    auto launch = [&](auto nt_c, bool h16, float* outp, bool redo_check) {
        constexpr int NTc = decltype(nt_c)::value;
        ConvArgs b = a;
        b.out0 = OutDesc{outp, out_stride, 0, out_stride};
        b.out1 = b.out0;
        b.redo_check = redo_check ? 1 : 0;
        float ms;
        const int reps = timing ? 5 : 1;
        if (h16) {                             // conv_nin_h (K = 32 chunks, 128-pixel blocks): "nin 2" = 2 input stages, "nin 3" = 3
            b.wpack16 = d_p16b; b.n_chunks = n_chunks32;
            const dim3 grid2((unsigned)((npix + 127) / 128), (unsigned)ng);
            const size_t tab2 = C.multi ? (size_t)n_chunks32 * 128 : 0;
            if (g_S == 2) {
                const size_t lds = NinHGeom<NTc, 2>::LDS_BYTES + tab2;
                if (C.multi) { auto k = conv_nin_h<NTc, true, 2>; CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); ms = time_kernel(k, grid2, lds, b, reps); }
                else { auto k = conv_nin_h<NTc, false, 2>; CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); ms = time_kernel(k, grid2, lds, b, reps); }
            } else {
                const size_t lds = NinHGeom<NTc, 3>::LDS_BYTES + tab2;
                if (C.multi) { auto k = conv_nin_h<NTc, true, 3>; CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); ms = time_kernel(k, grid2, lds, b, reps); }
                else { auto k = conv_nin_h<NTc, false, 3>; CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); ms = time_kernel(k, grid2, lds, b, reps); }
            }
        } else {
            const size_t lds = NinGeom<NTc>::LDS_BYTES + tab_bytes;
            if (C.multi) { auto k = conv_nin<NTc, true>; CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); ms = time_kernel(k, grid, lds, b, reps); }
            else { auto k = conv_nin<NTc, false>; CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); ms = time_kernel(k, grid, lds, b, reps); }
        }
        return ms;
    };
    auto dispatch = [&](bool h16, float* outp, bool redo_check) {
        switch (nt) {
            case 1: return launch(std::integral_constant<int, 1>{}, h16, outp, redo_check);
            case 2: return launch(std::integral_constant<int, 2>{}, h16, outp, redo_check);
            case 3: return launch(std::integral_constant<int, 3>{}, h16, outp, redo_check);
            case 4: return launch(std::integral_constant<int, 4>{}, h16, outp, redo_check);
            case 5: return launch(std::integral_constant<int, 5>{}, h16, outp, redo_check);
            default: return launch(std::integral_constant<int, 6>{}, h16, outp, redo_check);
        }
    };
    ms32 = dispatch(false, d_o32, false);
    ms16 = dispatch(true, d_o16, false);
    int bad = 0;
    std::vector<int> redo(nblocks);
    CK(hipMemcpy(redo.data(), d_redo, nblocks * 4, hipMemcpyDeviceToHost));
    long long nflag = 0;
    for (int v : redo) nflag += v != 0;
    if (overflow_test) {
        if (nflag != 1 || !redo[poke_pix / 256]) { printf("  ** redo flags wrong: %lld set, block of the poked pixel %d\n", nflag, redo[poke_pix / 256]); ++bad; }
        dispatch(false, d_o16, true);          // the fallback launch: only the flagged block is recomputed (into the split16 output)
    } else if (nflag) { printf("  ** %lld redo flags set without an overflow\n", nflag); ++bad; }

    std::vector<float> rf((size_t)npix * out_stride), o32(rf.size()), o16(rf.size());
    CK(hipMemcpy(rf.data(), d_ref, rf.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(o32.data(), d_o32, rf.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(o16.data(), d_o16, rf.size() * 4, hipMemcpyDeviceToHost));
    double e32 = 0, e16 = 0, d3216 = 0, mx = 0, s32 = 0, s16 = 0;
    long long cnt = 0;
    for (long long p = 0; p < npix; ++p)
        for (int c = 0; c < cout; ++c) {
            const size_t i = (size_t)p * out_stride + c;
            const double dd = std::fabs((double)o32[i] - o16[i]);
            if (!(dd <= d3216)) d3216 = dd;
            if (p % step == 0) {
                const double a32 = std::fabs((double)o32[i] - rf[i]), a16 = std::fabs((double)o16[i] - rf[i]);
                if (!(a32 <= e32)) e32 = a32;
                if (!(a16 <= e16)) e16 = a16;
                s32 += a32 * a32; s16 += a16 * a16; ++cnt;
                mx = std::fmax(mx, std::fabs((double)rf[i]));
            }
        }
    const double flop = 2.0 * cin * (double)cout * npix, bytes = 4.0 * (cin_phys + out_stride) * (double)npix;
    const bool ok = e16 <= 2.0 * e32 + 1e-6 * mx && std::isfinite(e16);
    printf("%-14s %8lld px  %4d -> %-3d %s NT%d groups %d chunks %d  f32 %7.3f ms (%6.1f TFLOP/s, %4.2f TB/s)  f16x3 %7.3f ms (%6.1f TFLOP/s, %4.2f TB/s)  "
           "max err f32 %.3g  f16x3 %.3g (rms %.3g / %.3g, max|ref| %.4g)  max|f32 - f16x3| %.3g%s\n",
           C.name, npix, cin, cout, C.multi ? "multi " : "single", nt, ng, n_chunks, ms32, flop / ms32 * 1e-9, bytes / ms32 * 1e-9, ms16, flop / ms16 * 1e-9,
           bytes / ms16 * 1e-9, e32, e16, std::sqrt(s32 / cnt), std::sqrt(s16 / cnt), mx, d3216, ok ? "" : "  ** MISMATCH **");
    bad += !ok;
    fflush(stdout);
    for (float* p : d_src) if (p) CK(hipFree(p));
    if (d_one) CK(hipFree(d_one));
    if (d_tab) CK(hipFree(d_tab));
    CK(hipFree(d_w)); CK(hipFree(d_p32)); CK(hipFree(d_p16b)); CK(hipFree(d_bias)); CK(hipFree(d_alpha)); CK(hipFree(d_bp)); CK(hipFree(d_ap));
    CK(hipFree(d_ref)); CK(hipFree(d_o32)); CK(hipFree(d_o16)); CK(hipFree(d_redo));
    return bad;
}

#ifdef H16_CONV3
#include "h16_tune_conv3.inc"
#endif

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "nin";
    int bad = 0;
    if (!strcmp(mode, "nin")) {
        if (argc > 2) g_S = atoi(argv[2]);
        printf("conv_nin_h input stages: %d\n", g_S);
        const std::vector<int> l12 = {196, 166, 148, 133, 120, 108, 97, 86, 76, 66, 57, 48};
        const std::vector<int> l8 = {96, 82, 75, 68, 62, 57, 52, 48};
        // edge cases first (full comparison against the float64 reference)
        bad += run_nin(NinCase{"ragged-1", 1, {33}, 16, false}, false, 0);
        bad += run_nin(NinCase{"ragged-2", 255, {40, 7}, 20, false}, false, 0);
        bad += run_nin(NinCase{"ragged-3", 1000, {37, 50, 9}, 96, true}, false, 0);
        bad += run_nin(NinCase{"ragged-4", 2305, {32}, 33, true}, false, 0);
        bad += run_nin(NinCase{"ragged-5", 5000, l8, 48, true}, false, 0);
        bad += run_nin(NinCase{"ragged-6", 777, l12, 97, false}, false, 0);
        bad += run_nin(NinCase{"overflow-s", 3000, l8, 96, false}, false, 1);
        bad += run_nin(NinCase{"overflow-m", 3000, l8, 96, true}, false, 1);
        // the bench layer
        bad += run_nin(NinCase{"A1||B1 L12", 1024LL * 2304, l12, 96, true}, true, 0);
        bad += run_nin(NinCase{"A1||B1 L12", 1024LL * 2304, l12, 96, false}, true, 0);
        bad += run_nin(NinCase{"A1||B1 L8", 256LL * 2304, l8, 96, true}, true, 0);
        printf("nin: %d failures\n", bad);
    }
#ifdef H16_CONV3
    else if (!strcmp(mode, "conv3")) bad = conv3_main(argc, argv);
#endif
    return bad != 0;
}
