// Correctness + timing harness for conv_wino against conv_igemm (same layer, same data).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wino_tune.hip -o tools/wino_tune && ./tools/wino_tune
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "conv_wino_r01.hpp"   // the shipped kernel: dcscn::conv_wino<NT, KC, WPS>
#include "conv_wino_fs.hpp"                                     // lab variants: dcscn_lab::...

using namespace dcscn;
namespace lab = dcscn_lab;

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

static int N = 1024, H = 48, W = 48;
static int g_lds_mul = 1;

struct Layer { const char* name; int cin, cout, in_stride, in_off, out_stride, out_off; };

static std::vector<float> rand_vec(size_t n, unsigned seed, float scale) {
    std::vector<float> h(n);
    unsigned s = seed;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (((s >> 8) & 0xffff) / 65536.0f - 0.5f) * scale; }
    return h;
}

// direct pack: [chunk][tap][kk][NS]
static std::vector<float> pack_direct(const std::vector<float>& w, int cin, int cout, int cin_phys, int kc, int nt, int* n_chunks) {
    const int ns = conv_ns(nt);
    *n_chunks = (cin_phys + kc - 1) / kc;
    std::vector<float> p((size_t)*n_chunks * 9 * kc * ns, 0.0f);
    for (int t = 0; t < 9; ++t)
        for (int c = 0; c < cin; ++c)
            for (int o = 0; o < cout; ++o)
                p[((size_t)(c / kc) * 9 + t) * kc * ns + (size_t)(c % kc) * ns + o] = w[((size_t)t * cin + c) * cout + o];
    return p;
}

// winograd pack: [ntile][chunk][f][kk][NS], U = G g G^T in double
static std::vector<float> pack_wino(const std::vector<float>& w, int cin, int cout, int cin_phys, int kc, int nt, int* n_chunks, int* n_tiles, int* nt_last) {
    const int ns = lab::wino_glb_ns(nt);
    const int tiles16 = (cout + 15) / 16;
    *n_tiles = (tiles16 + nt - 1) / nt;
    *nt_last = tiles16 - (*n_tiles - 1) * nt;
    *n_chunks = (cin_phys + kc - 1) / kc;
    const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    std::vector<float> p((size_t)*n_tiles * *n_chunks * 16 * kc * ns + 2048, 0.0f);
    for (int c = 0; c < cin; ++c)
        for (int o = 0; o < cout; ++o) {
            double g[3][3];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) g[i][j] = w[((size_t)(i * 3 + j) * cin + c) * cout + o];
            const int tile = o / (nt * 16), jn = o % (nt * 16);
            for (int xi = 0; xi < 4; ++xi)
                for (int nu = 0; nu < 4; ++nu) {
                    double u = 0;
                    for (int i = 0; i < 3; ++i)
                        for (int j = 0; j < 3; ++j) u += G[xi][i] * g[i][j] * G[nu][j];
                    const int f = xi * 4 + nu;
                    p[(((size_t)tile * *n_chunks + c / kc) * 16 + f) * kc * ns + (size_t)(c % kc) * ns + lab::wino_glb_col(nt, jn)] = (float)u;
                }
        }
    return p;
}

template <typename K>
static float time_kernel(K kern, dim3 grid, size_t lds, const ConvArgs& a, int reps = 5, int threads = 256) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, grid, dim3(threads), lds, 0, a);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, grid, dim3(threads), lds, 0, a);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    CK(hipGetLastError());
    return best;
}

static float *g_in, *g_ref, *g_out, *g_w, *g_bias;
static long long* g_dbg;

template <int MT, int NTD, int NTW, int KCW, int WPSW, bool DBW = false, int ABL = 0, int WAVES = 4, bool DMA = false, int PRIO = 0, int PF = 0, bool VPIPE = false>
void run(const Layer& L) {
    const int cin_phys = (L.cin + 3) & ~3;
    std::vector<float> w = rand_vec((size_t)9 * L.cin * L.cout, 777, 0.2f);
    ConvArgs a{};
    a.in = g_in; a.in_stride = L.in_stride; a.in_off = L.in_off; a.cin_phys = cin_phys;
    a.bias = g_bias; a.alpha = g_bias; a.act = ACT_ALPHA;
    a.N = N; a.H = H; a.W = W;
    a.split = 1 << 30; a.ps = 1; a.ps_c = 1; a.vec4 = 1; a.res = nullptr; a.res_stride = 1;
    const double flop = 2.0 * 9 * L.cin * (double)L.cout * N * H * W;

    // reference: direct kernel
    {
        using Gd = ConvGeom<3, MT, NTD, 4>;
        int nch;
        std::vector<float> p = pack_direct(w, L.cin, L.cout, cin_phys, 4, NTD, &nch);
        CK(hipMemcpy(g_w, p.data(), p.size() * sizeof(float), hipMemcpyHostToDevice));
        a.wpack = g_w; a.n_chunks = nch;
        a.tiles_x = (W + 15) / 16; a.tiles_y = (H + Gd::TH - 1) / Gd::TH;
        a.out0 = OutDesc{g_ref, L.out_stride, L.out_off, (L.cout + 3) & ~3};
        a.out1 = a.out0;
        auto kern = conv_igemm<3, MT, NTD, 4, false, 3>;
        const size_t lds = (size_t)Gd::BUF * sizeof(float);
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const float ms = time_kernel(kern, dim3(N * a.tiles_y * a.tiles_x, 1), lds, a);
        printf("%-8s %4d->%-4d direct MT%d NT%-2d            %8.3f ms  %7.2f TFLOP/s\n", L.name, L.cin, L.cout, MT, NTD, ms, flop / (ms * 1e-3) / 1e12);
    }
    // winograd
    {
        using Gw = lab::WinoGeom<NTW, KCW, WAVES>;
        int nch, ntiles, ntlast;
        std::vector<float> p = pack_wino(w, L.cin, L.cout, cin_phys, KCW, NTW, &nch, &ntiles, &ntlast);
        CK(hipMemcpy(g_w, p.data(), p.size() * sizeof(float), hipMemcpyHostToDevice));
        a.wpack = g_w; a.n_chunks = nch; a.n_full = ntlast;
        a.tiles_x = (W + 15) / 16; a.tiles_y = (H + Gw::TH - 1) / Gw::TH;
        a.out0 = OutDesc{g_out, L.out_stride, L.out_off, (L.cout + 3) & ~3};
        a.out1 = a.out0;
        auto kern = lab::conv_wino<NTW, KCW, WPSW, DBW, ABL, WAVES, DMA, PRIO, PF, VPIPE>;
        if (ABL == 7) { a.act = ACT_NONE; a.alpha = reinterpret_cast<const float*>(g_dbg); CK(hipMemset(g_dbg, 0, (size_t)(1024 + 4096 * 16 + 8 * 65536) * 8)); }
        const size_t lds = ((DBW || DMA) ? 2 : 1) * (size_t)Gw::BUF * sizeof(float) * (size_t)g_lds_mul;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int occ = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64 * WAVES, lds));
        CK(hipMemset(g_out, 0, (size_t)N * H * W * L.out_stride * sizeof(float)));
        const float ms = time_kernel(kern, dim3(N * a.tiles_y * a.tiles_x, ntiles), lds, a, 5, 64 * WAVES);
        // compare on a sample of images
        const size_t cnt = (size_t)8 * H * W * L.out_stride;
        std::vector<float> r(cnt), o(cnt);
        CK(hipMemcpy(r.data(), g_ref, cnt * sizeof(float), hipMemcpyDeviceToHost));
        CK(hipMemcpy(o.data(), g_out, cnt * sizeof(float), hipMemcpyDeviceToHost));
        double maxd = 0, maxv = 0;
        for (size_t px = 0; px < (size_t)8 * H * W; ++px)
            for (int c = 0; c < L.cout; ++c) {
                const size_t i = px * L.out_stride + L.out_off + c;
                maxd = std::fmax(maxd, std::fabs((double)r[i] - o[i]));
                maxv = std::fmax(maxv, std::fabs((double)r[i]));
            }
        if (ABL == 7) {
            std::vector<long long> d(1024 + 4096 * 16 + 8 * 65536);
            CK(hipMemcpy(d.data(), g_dbg, d.size() * 8, hipMemcpyDeviceToHost));
            double sum[4] = {0, 0, 0, 0};
            int cnt = 0;
            for (int b = 0; b < 4096; ++b)
                for (int w = 0; w < 4; ++w) {
                    const long long* e = &d[1024 + ((size_t)b * 4 + w) * 4];
                    if (e[2] == 0) continue;
                    for (int i = 0; i < 4; ++i) sum[i] += (double)e[i];
                    ++cnt;
                }
            double ct = 0, rt = 0;
            for (int b = 0; b < 512; ++b) { ct += (double)d[2 * b]; rt += (double)d[2 * b + 1]; }
            printf("   shader clock while this kernel runs: %.0f MHz (s_memtime ticks per s_memrealtime 100 MHz tick)\n", ct / rt * 100.0);
            {
                const long long* life = &d[1024 + 4096 * 16];
                long long tmin = -1, tmax = 0;
                double occ_ticks = 0, cyc = 0, pro = 0, loop = 0, epi = 0, drain = 0;
                int nwg = 0;
                for (int g = 0; g < 65536; ++g) {
                    const long long* e = life + 8 * (size_t)g;
                    if (e[1] == 0) continue;
                    if (tmin < 0 || e[0] < tmin) tmin = e[0];
                    if (e[1] > tmax) tmax = e[1];
                    occ_ticks += (double)(e[1] - e[0]);
                    cyc += (double)(e[3] - e[2]);
                    pro += (double)(e[4] - e[2]); loop += (double)(e[5] - e[4]); epi += (double)(e[6] - e[5]); drain += (double)(e[3] - e[6]);
                    ++nwg;
                }
                const double span = (double)(tmax - tmin);
                printf("   %d workgroups: device span %.3f ms, mean lifetime %.1f us = %.0f cycles (%.0f MHz), slot occupancy %.1f%% of 512\n",
                       nwg, span / 1e5, occ_ticks / nwg / 100.0, cyc / nwg, cyc / occ_ticks * 100.0, occ_ticks / (span * 512.0) * 100.0);
                {
                    printf("   chunk-loop cycles by dispatch-order decile:");
                    const int tot = nwg;
                    for (int dcl = 0; dcl < 10; ++dcl) {
                        double sm = 0; int cn = 0;
                        for (int g = dcl * tot / 10; g < (dcl + 1) * tot / 10; ++g) {
                            const long long* e = life + 8 * (size_t)g;
                            if (e[1] == 0) continue;
                            sm += (double)(e[5] - e[4]); ++cn;
                        }
                        printf(" %.0f", cn ? sm / cn : 0.0);
                    }
                    printf("\n");
                }
                printf("   wave 0 cycles: prologue %.0f  chunk loop %.0f  output transform + store issue %.0f  store drain %.0f\n",
                       pro / nwg, loop / nwg, epi / nwg, drain / nwg);
            }
            printf("   phases per chunk (cycles, avg over %d waves, %d chunks): store %.0f  barrier1 %.0f  load+compute %.0f  barrier2 %.0f  total %.0f\n",
                   cnt, nch, sum[0] / cnt / nch, sum[1] / cnt / nch, sum[2] / cnt / nch, sum[3] / cnt / nch,
                   (sum[0] + sum[1] + sum[2] + sum[3]) / cnt / nch);
        }
        printf("%-8s %4d->%-4d wino W%d DB%d DMA%d P%d PF%d VP%d ABL%d NT%d KC%d WPS%d tiles%d lds %5.1f KB occ %d  %8.3f ms  %7.2f TFLOP/s(alg)  max|diff| %.3g (max|ref| %.3g)\n",
               L.name, L.cin, L.cout, WAVES, (int)DBW, (int)DMA, PRIO, PF, (int)VPIPE, ABL, NTW, KCW, WPSW, ntiles, lds / 1024.0, occ, ms, flop / (ms * 1e-3) / 1e12, maxd, maxv);
    }
    fflush(stdout);
}

// the frequency-split kernel (conv_wino_fs.hpp) against the direct kernel
template <int MT, int NTD, int NTW, int WPSW>
void run_fs(const Layer& L) {
    constexpr int KCW = kWinoKC, WAVES = 4, ABL = 0, PRIO = 0, PF = 0;
    constexpr bool DBW = false, DMA = false, VPIPE = false;
    const int cin_phys = (L.cin + 3) & ~3;
    std::vector<float> w = rand_vec((size_t)9 * L.cin * L.cout, 777, 0.2f);
    ConvArgs a{};
    a.in = g_in; a.in_stride = L.in_stride; a.in_off = L.in_off; a.cin_phys = cin_phys;
    a.bias = g_bias; a.alpha = g_bias; a.act = ACT_ALPHA;
    a.N = N; a.H = H; a.W = W;
    a.split = 1 << 30; a.ps = 1; a.ps_c = 1; a.vec4 = 1; a.res = nullptr; a.res_stride = 1;
    const double flop = 2.0 * 9 * L.cin * (double)L.cout * N * H * W;

    // reference: direct kernel
    {
        using Gd = ConvGeom<3, MT, NTD, 4>;
        int nch;
        std::vector<float> p = pack_direct(w, L.cin, L.cout, cin_phys, 4, NTD, &nch);
        CK(hipMemcpy(g_w, p.data(), p.size() * sizeof(float), hipMemcpyHostToDevice));
        a.wpack = g_w; a.n_chunks = nch;
        a.tiles_x = (W + 15) / 16; a.tiles_y = (H + Gd::TH - 1) / Gd::TH;
        a.out0 = OutDesc{g_ref, L.out_stride, L.out_off, (L.cout + 3) & ~3};
        a.out1 = a.out0;
        auto kern = conv_igemm<3, MT, NTD, 4, false, 3>;
        const size_t lds = (size_t)Gd::BUF * sizeof(float);
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const float ms = time_kernel(kern, dim3(N * a.tiles_y * a.tiles_x, 1), lds, a);
        printf("%-8s %4d->%-4d direct MT%d NT%-2d            %8.3f ms  %7.2f TFLOP/s\n", L.name, L.cin, L.cout, MT, NTD, ms, flop / (ms * 1e-3) / 1e12);
    }
    // winograd
    {
        using Gw = lab::WinoGeom<NTW, KCW, WAVES>;
        int nch, ntiles, ntlast;
        std::vector<float> p = pack_wino(w, L.cin, L.cout, cin_phys, KCW, NTW, &nch, &ntiles, &ntlast);
        CK(hipMemcpy(g_w, p.data(), p.size() * sizeof(float), hipMemcpyHostToDevice));
        a.wpack = g_w; a.n_chunks = nch; a.n_full = ntlast;
        a.tiles_x = (W + 15) / 16; a.tiles_y = (H + Gw::TH - 1) / Gw::TH;
        a.out0 = OutDesc{g_out, L.out_stride, L.out_off, (L.cout + 3) & ~3};
        a.out1 = a.out0;
        auto kern = lab::conv_wino_fs<NTW, WPSW>;
        if (ABL == 7) { a.act = ACT_NONE; a.alpha = reinterpret_cast<const float*>(g_dbg); CK(hipMemset(g_dbg, 0, (size_t)(1024 + 4096 * 16 + 8 * 65536) * 8)); }
        const size_t lds = lab::wino_fs_lds_bytes<NTW>() * (size_t)g_lds_mul;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int occ = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64 * WAVES, lds));
        CK(hipMemset(g_out, 0, (size_t)N * H * W * L.out_stride * sizeof(float)));
        const float ms = time_kernel(kern, dim3(N * a.tiles_y * a.tiles_x, ntiles), lds, a, 5, 64 * WAVES);
        // compare on a sample of images
        const size_t cnt = (size_t)8 * H * W * L.out_stride;
        std::vector<float> r(cnt), o(cnt);
        CK(hipMemcpy(r.data(), g_ref, cnt * sizeof(float), hipMemcpyDeviceToHost));
        CK(hipMemcpy(o.data(), g_out, cnt * sizeof(float), hipMemcpyDeviceToHost));
        double maxd = 0, maxv = 0;
        for (size_t px = 0; px < (size_t)8 * H * W; ++px)
            for (int c = 0; c < L.cout; ++c) {
                const size_t i = px * L.out_stride + L.out_off + c;
                maxd = std::fmax(maxd, std::fabs((double)r[i] - o[i]));
                maxv = std::fmax(maxv, std::fabs((double)r[i]));
            }
        if (ABL == 7) {
            std::vector<long long> d(1024 + 4096 * 16 + 8 * 65536);
            CK(hipMemcpy(d.data(), g_dbg, d.size() * 8, hipMemcpyDeviceToHost));
            double sum[4] = {0, 0, 0, 0};
            int cnt = 0;
            for (int b = 0; b < 4096; ++b)
                for (int w = 0; w < 4; ++w) {
                    const long long* e = &d[1024 + ((size_t)b * 4 + w) * 4];
                    if (e[2] == 0) continue;
                    for (int i = 0; i < 4; ++i) sum[i] += (double)e[i];
                    ++cnt;
                }
            double ct = 0, rt = 0;
            for (int b = 0; b < 512; ++b) { ct += (double)d[2 * b]; rt += (double)d[2 * b + 1]; }
            printf("   shader clock while this kernel runs: %.0f MHz (s_memtime ticks per s_memrealtime 100 MHz tick)\n", ct / rt * 100.0);
            {
                const long long* life = &d[1024 + 4096 * 16];
                long long tmin = -1, tmax = 0;
                double occ_ticks = 0, cyc = 0, pro = 0, loop = 0, epi = 0, drain = 0;
                int nwg = 0;
                for (int g = 0; g < 65536; ++g) {
                    const long long* e = life + 8 * (size_t)g;
                    if (e[1] == 0) continue;
                    if (tmin < 0 || e[0] < tmin) tmin = e[0];
                    if (e[1] > tmax) tmax = e[1];
                    occ_ticks += (double)(e[1] - e[0]);
                    cyc += (double)(e[3] - e[2]);
                    pro += (double)(e[4] - e[2]); loop += (double)(e[5] - e[4]); epi += (double)(e[6] - e[5]); drain += (double)(e[3] - e[6]);
                    ++nwg;
                }
                const double span = (double)(tmax - tmin);
                printf("   %d workgroups: device span %.3f ms, mean lifetime %.1f us = %.0f cycles (%.0f MHz), slot occupancy %.1f%% of 512\n",
                       nwg, span / 1e5, occ_ticks / nwg / 100.0, cyc / nwg, cyc / occ_ticks * 100.0, occ_ticks / (span * 512.0) * 100.0);
                {
                    printf("   chunk-loop cycles by dispatch-order decile:");
                    const int tot = nwg;
                    for (int dcl = 0; dcl < 10; ++dcl) {
                        double sm = 0; int cn = 0;
                        for (int g = dcl * tot / 10; g < (dcl + 1) * tot / 10; ++g) {
                            const long long* e = life + 8 * (size_t)g;
                            if (e[1] == 0) continue;
                            sm += (double)(e[5] - e[4]); ++cn;
                        }
                        printf(" %.0f", cn ? sm / cn : 0.0);
                    }
                    printf("\n");
                }
                printf("   wave 0 cycles: prologue %.0f  chunk loop %.0f  output transform + store issue %.0f  store drain %.0f\n",
                       pro / nwg, loop / nwg, epi / nwg, drain / nwg);
            }
            printf("   phases per chunk (cycles, avg over %d waves, %d chunks): store %.0f  barrier1 %.0f  load+compute %.0f  barrier2 %.0f  total %.0f\n",
                   cnt, nch, sum[0] / cnt / nch, sum[1] / cnt / nch, sum[2] / cnt / nch, sum[3] / cnt / nch,
                   (sum[0] + sum[1] + sum[2] + sum[3]) / cnt / nch);
        }
        printf("%-8s %4d->%-4d wiFS W%d DB%d DMA%d P%d PF%d VP%d ABL%d NT%d KC%d WPS%d tiles%d lds %5.1f KB occ %d  %8.3f ms  %7.2f TFLOP/s(alg)  max|diff| %.3g (max|ref| %.3g)\n",
               L.name, L.cin, L.cout, WAVES, (int)DBW, (int)DMA, PRIO, PF, (int)VPIPE, ABL, NTW, KCW, WPSW, ntiles, lds / 1024.0, occ, ms, flop / (ms * 1e-3) / 1e12, maxd, maxv);
    }
    fflush(stdout);
}

// the SHIPPED kernel (csrc/conv_wino.hpp) against the direct kernel
template <int MT, int NTD, int NTW, int KCS, int WPSW, int PFS = 3>
void run_shipped(const Layer& L) {
    constexpr int KCW = KCS, WAVES = 4, ABL = 0, PRIO = 0, PF = 0;
    constexpr bool DBW = false, DMA = false, VPIPE = false;
    const int cin_phys = (L.cin + 3) & ~3;
    std::vector<float> w = rand_vec((size_t)9 * L.cin * L.cout, 777, 0.2f);
    ConvArgs a{};
    a.in = g_in; a.in_stride = L.in_stride; a.in_off = L.in_off; a.cin_phys = cin_phys;
    a.bias = g_bias; a.alpha = g_bias; a.act = ACT_ALPHA;
    a.N = N; a.H = H; a.W = W;
    a.split = 1 << 30; a.ps = 1; a.ps_c = 1; a.vec4 = 1; a.res = nullptr; a.res_stride = 1;
    const double flop = 2.0 * 9 * L.cin * (double)L.cout * N * H * W;

    // reference: direct kernel
    {
        using Gd = ConvGeom<3, MT, NTD, 4>;
        int nch;
        std::vector<float> p = pack_direct(w, L.cin, L.cout, cin_phys, 4, NTD, &nch);
        CK(hipMemcpy(g_w, p.data(), p.size() * sizeof(float), hipMemcpyHostToDevice));
        a.wpack = g_w; a.n_chunks = nch;
        a.tiles_x = (W + 15) / 16; a.tiles_y = (H + Gd::TH - 1) / Gd::TH;
        a.out0 = OutDesc{g_ref, L.out_stride, L.out_off, (L.cout + 3) & ~3};
        a.out1 = a.out0;
        auto kern = conv_igemm<3, MT, NTD, 4, false, 3>;
        const size_t lds = (size_t)Gd::BUF * sizeof(float);
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const float ms = time_kernel(kern, dim3(N * a.tiles_y * a.tiles_x, 1), lds, a);
        printf("%-8s %4d->%-4d direct MT%d NT%-2d            %8.3f ms  %7.2f TFLOP/s\n", L.name, L.cin, L.cout, MT, NTD, ms, flop / (ms * 1e-3) / 1e12);
    }
    // winograd
    {
        using Gw = lab::WinoGeom<NTW, KCW, WAVES>;
        int nch, ntiles, ntlast;
        std::vector<float> p = pack_wino(w, L.cin, L.cout, cin_phys, KCW, NTW, &nch, &ntiles, &ntlast);
        CK(hipMemcpy(g_w, p.data(), p.size() * sizeof(float), hipMemcpyHostToDevice));
        a.wpack = g_w; a.n_chunks = nch; a.n_full = ntlast;
        a.tiles_x = (W + 15) / 16; a.tiles_y = (H + Gw::TH - 1) / Gw::TH;
        a.out0 = OutDesc{g_out, L.out_stride, L.out_off, (L.cout + 3) & ~3};
        a.out1 = a.out0;
        auto kern = dcscn::conv_wino<NTW, KCW, WPSW, PFS>;
        if (ABL == 7) { a.act = ACT_NONE; a.alpha = reinterpret_cast<const float*>(g_dbg); CK(hipMemset(g_dbg, 0, (size_t)(1024 + 4096 * 16 + 8 * 65536) * 8)); }
        const size_t lds = (size_t)dcscn::WinoGeom<NTW, KCW>::BUF * sizeof(float) * (size_t)g_lds_mul;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int occ = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64 * WAVES, lds));
        CK(hipMemset(g_out, 0, (size_t)N * H * W * L.out_stride * sizeof(float)));
        a.n_groups = ntiles; a.group_span = ntiles < 3 ? ntiles : 3;
        const int tiles8 = (N * a.tiles_y * a.tiles_x + 7) / 8;
        const float ms = time_kernel(kern, dim3(tiles8 * 8 * a.group_span * ((ntiles + a.group_span - 1) / a.group_span)), lds, a, 5, 64 * WAVES);
        // compare on a sample of images
        const size_t cnt = (size_t)8 * H * W * L.out_stride;
        std::vector<float> r(cnt), o(cnt);
        CK(hipMemcpy(r.data(), g_ref, cnt * sizeof(float), hipMemcpyDeviceToHost));
        CK(hipMemcpy(o.data(), g_out, cnt * sizeof(float), hipMemcpyDeviceToHost));
        double maxd = 0, maxv = 0;
        for (size_t px = 0; px < (size_t)8 * H * W; ++px)
            for (int c = 0; c < L.cout; ++c) {
                const size_t i = px * L.out_stride + L.out_off + c;
                maxd = std::fmax(maxd, std::fabs((double)r[i] - o[i]));
                maxv = std::fmax(maxv, std::fabs((double)r[i]));
            }
        if (ABL == 7) {
            std::vector<long long> d(1024 + 4096 * 16 + 8 * 65536);
            CK(hipMemcpy(d.data(), g_dbg, d.size() * 8, hipMemcpyDeviceToHost));
            double sum[4] = {0, 0, 0, 0};
            int cnt = 0;
            for (int b = 0; b < 4096; ++b)
                for (int w = 0; w < 4; ++w) {
                    const long long* e = &d[1024 + ((size_t)b * 4 + w) * 4];
                    if (e[2] == 0) continue;
                    for (int i = 0; i < 4; ++i) sum[i] += (double)e[i];
                    ++cnt;
                }
            double ct = 0, rt = 0;
            for (int b = 0; b < 512; ++b) { ct += (double)d[2 * b]; rt += (double)d[2 * b + 1]; }
            printf("   shader clock while this kernel runs: %.0f MHz (s_memtime ticks per s_memrealtime 100 MHz tick)\n", ct / rt * 100.0);
            {
                const long long* life = &d[1024 + 4096 * 16];
                long long tmin = -1, tmax = 0;
                double occ_ticks = 0, cyc = 0, pro = 0, loop = 0, epi = 0, drain = 0;
                int nwg = 0;
                for (int g = 0; g < 65536; ++g) {
                    const long long* e = life + 8 * (size_t)g;
                    if (e[1] == 0) continue;
                    if (tmin < 0 || e[0] < tmin) tmin = e[0];
                    if (e[1] > tmax) tmax = e[1];
                    occ_ticks += (double)(e[1] - e[0]);
                    cyc += (double)(e[3] - e[2]);
                    pro += (double)(e[4] - e[2]); loop += (double)(e[5] - e[4]); epi += (double)(e[6] - e[5]); drain += (double)(e[3] - e[6]);
                    ++nwg;
                }
                const double span = (double)(tmax - tmin);
                printf("   %d workgroups: device span %.3f ms, mean lifetime %.1f us = %.0f cycles (%.0f MHz), slot occupancy %.1f%% of 512\n",
                       nwg, span / 1e5, occ_ticks / nwg / 100.0, cyc / nwg, cyc / occ_ticks * 100.0, occ_ticks / (span * 512.0) * 100.0);
                {
                    printf("   chunk-loop cycles by dispatch-order decile:");
                    const int tot = nwg;
                    for (int dcl = 0; dcl < 10; ++dcl) {
                        double sm = 0; int cn = 0;
                        for (int g = dcl * tot / 10; g < (dcl + 1) * tot / 10; ++g) {
                            const long long* e = life + 8 * (size_t)g;
                            if (e[1] == 0) continue;
                            sm += (double)(e[5] - e[4]); ++cn;
                        }
                        printf(" %.0f", cn ? sm / cn : 0.0);
                    }
                    printf("\n");
                }
                printf("   wave 0 cycles: prologue %.0f  chunk loop %.0f  output transform + store issue %.0f  store drain %.0f\n",
                       pro / nwg, loop / nwg, epi / nwg, drain / nwg);
            }
            printf("   phases per chunk (cycles, avg over %d waves, %d chunks): store %.0f  barrier1 %.0f  load+compute %.0f  barrier2 %.0f  total %.0f\n",
                   cnt, nch, sum[0] / cnt / nch, sum[1] / cnt / nch, sum[2] / cnt / nch, sum[3] / cnt / nch,
                   (sum[0] + sum[1] + sum[2] + sum[3]) / cnt / nch);
        }
        printf("%-8s %4d->%-4d SHIP W%d DB%d DMA%d P%d PF%d VP%d ABL%d NT%d KC%d WPS%d tiles%d lds %5.1f KB occ %d  %8.3f ms  %7.2f TFLOP/s(alg)  max|diff| %.3g (max|ref| %.3g)\n",
               L.name, L.cin, L.cout, WAVES, (int)DBW, (int)DMA, PRIO, PF, (int)VPIPE, ABL, NTW, KCW, WPSW, ntiles, lds / 1024.0, occ, ms, flop / (ms * 1e-3) / 1e12, maxd, maxv);
    }
    fflush(stdout);
}

int main(int argc, char** argv) {
    if (argc > 1) N = atoi(argv[1]);
    if (getenv("WINO_LDS_MUL")) g_lds_mul = atoi(getenv("WINO_LDS_MUL"));
    const size_t act = (size_t)N * H * W * 1316;
    CK(hipMalloc(&g_in, act * sizeof(float)));
    CK(hipMalloc(&g_ref, act * sizeof(float)));
    CK(hipMalloc(&g_out, act * sizeof(float)));
    CK(hipMalloc(&g_w, (size_t)(32u << 20) * sizeof(float)));
    CK(hipMalloc(&g_bias, 4096 * sizeof(float)));
    CK(hipMalloc(&g_dbg, (size_t)(1024 + 4096 * 16 + 8 * 65536) * 8));
    {
        std::vector<float> h = rand_vec(16u << 20, 4242, 100.0f);
        for (size_t off = 0; off < act; off += h.size())
            CK(hipMemcpy(g_in + off, h.data(), std::min(h.size(), act - off) * sizeof(float), hipMemcpyHostToDevice));
        std::vector<float> b = rand_vec(4096, 99, 0.5f);
        for (auto& v : b) v = std::fabs(v) * 0.5f;
        CK(hipMemcpy(g_bias, b.data(), 4096 * sizeof(float), hipMemcpyHostToDevice));
    }
    CK(hipMemset(g_ref, 0, act * sizeof(float)));
    const Layer cnn2{"CNN2", 196, 166, 1316, 0, 1316, 196};
    const Layer cnn5{"CNN5", 133, 120, 1316, 512, 1316, 648};
    const Layer cnn8{"CNN8", 97, 86, 1316, 984, 1316, 1084};
    const Layer cnn12{"CNN12", 57, 48, 1316, 1248, 1316, 1268};
    const Layer upps{"Up-PS", 96, 384, 96, 0, 384, 0};
    run_shipped<2, 11, 3, 4, 2>(cnn2);                          // csrc/conv_wino.hpp as built into the library
    run<2, 11, 3, 4, 2, false, 0, 4, false, 0, 3>(cnn2);        // the same loop in the lab kernel (2-D grid, no XCD-aware ids)
    run_shipped<2, 8, 3, 4, 2>(cnn5);
    run<2, 8, 3, 4, 2, false, 0, 4, false, 0, 3>(cnn5);
    run_shipped<4, 1, 1, 8, 4>(Layer{"tail16", 166, 16, 1316, 196, 1316, 364});
    return 0;
}
