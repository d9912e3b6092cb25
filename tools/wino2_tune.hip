// Correctness + timing harness for conv_wino2 (LDS-DMA staged Winograd) against a naive direct conv.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wino2_tune.hip -o tools/wino2_tune
//   ./tools/wino2_tune bench      every 3x3 layer of the bench model (L12_F196to48 x2), 1024 patches of 48x48
//   ./tools/wino2_tune edge       ragged sizes, channel tails, narrow last groups, depth_to_space -- against the naive conv
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../dcscn-super-resolution_amd/csrc/conv_wino2.hpp"

using namespace dcscn;

#ifndef W2_WPS
#define W2_WPS 2
#endif
#ifndef W2_PF
#define W2_PF 3
#endif
#ifndef W2_ABL
#define W2_ABL 0
#endif

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

struct Layer { const char* name; int cin, cout, in_stride, in_off, out_stride, out_off, ps; };

static std::vector<float> rand_vec(size_t n, unsigned seed, float scale) {
    std::vector<float> h(n);
    unsigned s = seed;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (((s >> 8) & 0xffff) / 65536.0f - 0.5f) * scale; }
    return h;
}

// naive reference: one thread per (pixel, channel), float64 accumulation, HWIO weights, bias + PReLU, optional depth_to_space
__global__ void naive_conv(const float* in, int in_stride, int in_off, int cin, const float* w, int cout, const float* bias,
                           const float* alpha, int n, int H, int W, float* out, int out_stride, int out_off, int ps) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)n * H * W * cout;
    if (idx >= total) return;
    const int co = (int)(idx % cout);
    long long p = idx / cout;
    const int x = (int)(p % W); p /= W;
    const int y = (int)(p % H);
    const int img = (int)(p / H);
    double s = 0.0;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
            const float* ip = in + ((size_t)(img * H + yy) * W + xx) * in_stride + in_off;
            const float* wp = w + ((size_t)((dy + 1) * 3 + (dx + 1)) * cin) * cout + co;
            for (int c = 0; c < cin; ++c) s += (double)ip[c] * (double)wp[(size_t)c * cout];
        }
    float v = (float)(s + bias[co]);
    v = v > 0.0f ? v : alpha[co] * v;
    if (ps == 1) out[((size_t)(img * H + y) * W + x) * out_stride + out_off + co] = v;
    else {
        const int psc = cout / (ps * ps);
        const int sub = co / psc, ch = co - sub * psc, ay = sub / ps, bx = sub - ay * ps;
        out[((size_t)(img * H * ps + y * ps + ay) * (W * ps) + (x * ps + bx)) * out_stride + out_off + ch] = v;
    }
}

// r01 winograd pack: [group][chunk][f][kk][NS], KC = 4
static std::vector<float> pack_wino1(const std::vector<float>& w, int cin, int cout, int cin_phys, int nt, int* n_chunks, int* n_groups, int* nt_last) {
    const int kc = 4, ns = conv_ns(nt);
    const int tiles16 = (cout + 15) / 16;
    *n_groups = (tiles16 + nt - 1) / nt;
    *nt_last = tiles16 - (*n_groups - 1) * nt;
    *n_chunks = (cin_phys + kc - 1) / kc;
    const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    std::vector<float> p((size_t)*n_groups * *n_chunks * 16 * kc * ns + 2048, 0.0f);
    for (int c = 0; c < cin; ++c)
        for (int o = 0; o < cout; ++o) {
            double g[3][3];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) g[i][j] = w[((size_t)(i * 3 + j) * cin + c) * cout + o];
            const int grp = o / (nt * 16), jn = o % (nt * 16);
            for (int xi = 0; xi < 4; ++xi)
                for (int nu = 0; nu < 4; ++nu) {
                    double u = 0;
                    for (int i = 0; i < 3; ++i)
                        for (int j = 0; j < 3; ++j) u += G[xi][i] * g[i][j] * G[nu][j];
                    p[(((size_t)grp * *n_chunks + c / kc) * 16 + xi * 4 + nu) * kc * ns + (size_t)(c % kc) * ns + jn] = (float)u;
                }
        }
    return p;
}

// conv_wino2 pack (mirrors api.hip finalize_op): tiles spread evenly over ceil(tiles / 3) groups of which the first n_full
// hold nt tiles and the others nt - 1; [group][chunk of 8][f][(c & 1) * 4 + (c >> 1)][NS]
static void wino2_plan(int cout, int* n_groups, int* nt, int* n_full) {
    const int tiles16 = (cout + 15) / 16;
    *n_groups = (tiles16 + 2) / 3;
    *nt = (tiles16 + *n_groups - 1) / *n_groups;
    *n_full = tiles16 - *n_groups * (*nt - 1);
}
static int wino2_padded(int cc, int nt, int n_full) {
    const int t = cc / 16, wide = n_full * nt;
    const int g = t < wide ? t / nt : n_full + (t - wide) / (nt - 1);
    const int tg = t < wide ? t % nt : (t - wide) % (nt - 1);
    return (g * nt + tg) * 16 + cc % 16;
}
static std::vector<float> pack_wino2(const std::vector<float>& w, int cin, int cout, int cin_phys, int nt, int n_groups, int n_full, int* n_chunks) {
    const int kc = 8, ns = conv_ns(nt);
    *n_chunks = (cin_phys + kc - 1) / kc;
    const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    std::vector<float> p((size_t)n_groups * *n_chunks * 16 * kc * ns, 0.0f);
    for (int c = 0; c < cin; ++c)
        for (int o = 0; o < cout; ++o) {
            double g[3][3];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) g[i][j] = w[((size_t)(i * 3 + j) * cin + c) * cout + o];
            const int pc = wino2_padded(o, nt, n_full);
            const int grp = pc / (nt * 16), jn = pc % (nt * 16);
            const int cc = c % kc, row = (cc & 1) * 4 + (cc >> 1);
            for (int xi = 0; xi < 4; ++xi)
                for (int nu = 0; nu < 4; ++nu) {
                    double u = 0;
                    for (int i = 0; i < 3; ++i)
                        for (int j = 0; j < 3; ++j) u += G[xi][i] * g[i][j] * G[nu][j];
                    p[(((size_t)grp * *n_chunks + c / kc) * 16 + xi * 4 + nu) * kc * ns + (size_t)row * ns + jn] = (float)u;
                }
        }
    return p;
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

static float *g_in, *g_ref, *g_out, *g_w, *g_bias, *g_alpha, *g_bias2, *g_alpha2, *g_wraw;

static dim3 wino_grid(ConvArgs& a, int n_groups) {
    a.n_groups = n_groups;
    a.group_span = n_groups < 3 ? n_groups : 3;
    const long long tiles = (long long)a.N * a.tiles_y * a.tiles_x;
    const int phases = (n_groups + a.group_span - 1) / a.group_span;
    return dim3((unsigned)(((tiles + 7) / 8) * 8 * a.group_span * phases));
}

struct Result { double maxd, maxv; float ms_old, ms_new; };

// NT: channel tiles per group of the new kernel (1..3)
template <int NT>
static Result run(const Layer& L, int N, int H, int W, int n_check, bool time_old, bool quiet = false) {
    const int cin_phys = (L.cin + 3) & ~3;
    std::vector<float> w = rand_vec((size_t)9 * L.cin * L.cout, 777 + L.cin, 0.2f);
    CK(hipMemcpy(g_wraw, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
    ConvArgs a{};
    a.in = g_in; a.in_stride = L.in_stride; a.in_off = L.in_off; a.cin_phys = cin_phys;
    a.bias = g_bias; a.alpha = g_alpha; a.act = ACT_ALPHA;
    a.N = N; a.H = H; a.W = W;
    a.split = 1 << 30; a.ps = L.ps; a.ps_c = L.cout / (L.ps * L.ps); a.vec4 = 1; a.res = nullptr; a.res_stride = 1;
    a.tiles_x = (W + 15) / 16; a.tiles_y = (H + 15) / 16;
    const int owidth = L.ps == 1 ? ((L.cout + 3) & ~3) : L.cout;
    Result r{};
    const double flop = 2.0 * 9 * L.cin * (double)L.cout * N * H * W;
    const size_t out_floats = (size_t)N * H * L.ps * W * L.ps * L.out_stride;

    // naive reference on the first n_check images
    CK(hipMemset(g_ref, 0, out_floats * sizeof(float)));
    {
        const long long total = (long long)n_check * H * W * L.cout;
        hipLaunchKernelGGL(naive_conv, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, g_in, L.in_stride, L.in_off, L.cin, g_wraw, L.cout,
                           g_bias, g_alpha, n_check, H, W, g_ref, L.out_stride, L.out_off, L.ps);
        CK(hipDeviceSynchronize());
    }
    (void)time_old;       // (the r01 kernel this leg timed is gone from tools/; its numbers: profiles/r01_wino_tune_log.txt)
    {
        using G2 = Wino2Geom<NT>;
        int nch, ng, nt, nfull;
        wino2_plan(L.cout, &ng, &nt, &nfull);
        if (nt != NT) { printf("plan mismatch\n"); exit(1); }
        const int ntl = nfull;
        std::vector<float> p = pack_wino2(w, L.cin, L.cout, cin_phys, NT, ng, nfull, &nch);
        CK(hipMemcpy(g_w, p.data(), p.size() * sizeof(float), hipMemcpyHostToDevice));
        {   // bias / slope in the padded layout of the plan
            std::vector<float> bh(4096), ah(4096), bp(4096, 0.0f), ap(4096, 0.0f);
            CK(hipMemcpy(bh.data(), g_bias, 4096 * sizeof(float), hipMemcpyDeviceToHost));
            CK(hipMemcpy(ah.data(), g_alpha, 4096 * sizeof(float), hipMemcpyDeviceToHost));
            for (int o = 0; o < L.cout; ++o) { bp[wino2_padded(o, NT, nfull)] = bh[o]; ap[wino2_padded(o, NT, nfull)] = ah[o]; }
            CK(hipMemcpy(g_bias2, bp.data(), 4096 * sizeof(float), hipMemcpyHostToDevice));
            CK(hipMemcpy(g_alpha2, ap.data(), 4096 * sizeof(float), hipMemcpyHostToDevice));
        }
        ConvArgs b = a;
        b.wpack = g_w; b.n_chunks = nch; b.n_full = nfull; b.bias = g_bias2; b.alpha = g_alpha2;
        b.out0 = OutDesc{g_out, L.out_stride, L.out_off, owidth};
        b.out1 = b.out0;
        const dim3 grid = wino_grid(b, ng);
        auto kern = conv_wino2<NT, W2_WPS, W2_PF, W2_ABL>;
        const size_t lds = G2::LDS_BYTES;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipMemset(g_out, 0, out_floats * sizeof(float)));
        r.ms_new = time_kernel(kern, grid, lds, b, (time_old || W2_ABL || N > 64) ? 5 : 1);
        const size_t cnt = (size_t)n_check * H * L.ps * W * L.ps * L.out_stride;
        std::vector<float> rf(cnt), o(cnt);
        CK(hipMemcpy(rf.data(), g_ref, cnt * sizeof(float), hipMemcpyDeviceToHost));
        CK(hipMemcpy(o.data(), g_out, cnt * sizeof(float), hipMemcpyDeviceToHost));
        const int cstore = L.ps == 1 ? L.cout : L.cout / (L.ps * L.ps);
        for (size_t px = 0; px < (size_t)n_check * H * L.ps * W * L.ps; ++px)
            for (int c = 0; c < cstore; ++c) {
                const size_t i = px * L.out_stride + L.out_off + c;
                const double d = std::fabs((double)rf[i] - o[i]);
                if (!(d <= r.maxd)) r.maxd = d;               // NaN-propagating max
                r.maxv = std::fmax(r.maxv, std::fabs((double)rf[i]));
            }
        // channels outside the written slice must stay untouched (zero)
        double stray = 0;
        for (size_t px = 0; px < (size_t)n_check * H * L.ps * W * L.ps; ++px)
            for (int c = 0; c < L.out_stride; ++c)
                if (c < L.out_off || c >= L.out_off + ((cstore + 3) & ~3)) stray = std::fmax(stray, std::fabs((double)o[px * L.out_stride + c]));
        if (!quiet || !(r.maxd < 2e-3) || stray != 0)
            printf("%-10s %4d->%-4d %dx%dx%d ps%d NT%d groups %d (%d wide) chunks %d  old %7.3f ms  new %7.3f ms %7.2f TFLOP/s(alg)  max|diff| %.3g (max|ref| %.3g)%s%s\n",
                   L.name, L.cin, L.cout, N, H, W, L.ps, NT, ng, ntl, nch, r.ms_old, r.ms_new, flop / (r.ms_new * 1e-3) / 1e12, r.maxd, r.maxv,
                   r.maxd < 2e-3 ? "" : "  ** MISMATCH **", stray != 0 ? "  ** STRAY WRITE **" : "");
    }
    fflush(stdout);
    return r;
}

static Result run_layer(const Layer& L, int N, int H, int W, int n_check, bool time_old, bool quiet = false) {
    int ng, nt, nfull;
    wino2_plan(L.cout, &ng, &nt, &nfull);
    return nt == 1 ? run<1>(L, N, H, W, n_check, time_old, quiet) : nt == 2 ? run<2>(L, N, H, W, n_check, time_old, quiet) : run<3>(L, N, H, W, n_check, time_old, quiet);
}

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "bench";
    const int N = 1024, H = 48, W = 48;
    const size_t in_floats = (size_t)N * H * W * 1316 + 64;
    const size_t out_floats = (size_t)N * H * 2 * W * 2 * 96 + 64;          // Up-PS writes [N, 96, 96, 96]
    CK(hipMalloc(&g_in, in_floats * sizeof(float)));
    CK(hipMalloc(&g_ref, std::max(in_floats, out_floats) * sizeof(float)));
    CK(hipMalloc(&g_out, std::max(in_floats, out_floats) * sizeof(float)));
    CK(hipMalloc(&g_w, (size_t)64 << 20));
    CK(hipMalloc(&g_wraw, (size_t)16 << 20));
    CK(hipMalloc(&g_bias, 4096 * sizeof(float)));
    CK(hipMalloc(&g_alpha, 4096 * sizeof(float)));
    CK(hipMalloc(&g_bias2, 4096 * sizeof(float)));
    CK(hipMalloc(&g_alpha2, 4096 * sizeof(float)));
    {
        std::vector<float> x = rand_vec(in_floats, 1, 2.0f);
        CK(hipMemcpy(g_in, x.data(), in_floats * sizeof(float), hipMemcpyHostToDevice));
        std::vector<float> b = rand_vec(4096, 2, 0.2f), al = rand_vec(4096, 3, 0.5f);
        CK(hipMemcpy(g_bias, b.data(), 4096 * sizeof(float), hipMemcpyHostToDevice));
        CK(hipMemcpy(g_alpha, al.data(), 4096 * sizeof(float), hipMemcpyHostToDevice));
    }
    if (!strcmp(mode, "abl")) {            // a few layers only (ablated builds give wrong results by design)
        const Layer layers[] = {{"CNN2", 196, 166, 1316, 0, 1316, 196, 1}, {"CNN3", 166, 148, 1316, 196, 1316, 364, 1}, {"CNN4", 148, 133, 1316, 364, 1316, 512, 1},
                                {"CNN7", 108, 97, 1316, 768, 1316, 876, 1}, {"CNN3t", 166, 16, 1316, 196, 1316, 364, 1}};
        printf("ABL %d PF %d\n", W2_ABL, W2_PF);
        for (const Layer& L : layers) {
            run_layer(L, N, H, W, 1, false);
        }
    } else if (!strcmp(mode, "bench")) {
        // the 3x3 layers of L12_F196to48 x2 as the plan lays them out (concat stride 1316, slices at 4-channel boundaries)
        const Layer layers[] = {
            {"CNN2", 196, 166, 1316, 0, 1316, 196, 1},   {"CNN3", 166, 148, 1316, 196, 1316, 364, 1},  {"CNN4", 148, 133, 1316, 364, 1316, 512, 1},
            {"CNN5", 133, 120, 1316, 512, 1316, 648, 1}, {"CNN6", 120, 108, 1316, 648, 1316, 768, 1},  {"CNN7", 108, 97, 1316, 768, 1316, 876, 1},
            {"CNN8", 97, 86, 1316, 876, 1316, 976, 1},   {"CNN9", 86, 76, 1316, 976, 1316, 1064, 1},   {"CNN10", 76, 66, 1316, 1064, 1316, 1140, 1},
            {"CNN11", 66, 57, 1316, 1140, 1316, 1208, 1}, {"CNN12", 57, 48, 1316, 1208, 1316, 1268, 1}, {"B2", 32, 32, 32, 0, 96, 0, 1},
            {"Up-PS", 96, 384, 96, 0, 96, 0, 2},
        };
        double so = 0, sn = 0;
        for (const Layer& L : layers) {
            Result r = run_layer(L, N, H, W, 2, true);
            so += r.ms_old; sn += r.ms_new;
        }
        printf("sum of 3x3 layers: old %.3f ms  new %.3f ms\n", so, sn);
    } else if (!strcmp(mode, "edge")) {
        int bad = 0, n = 0;
        const int sizes[][2] = {{1, 1}, {2, 3}, {15, 17}, {16, 16}, {17, 33}, {31, 5}, {48, 48}, {50, 21}};
        const int cins[] = {32, 36, 40, 57, 100};
        const int couts[] = {16, 20, 33, 48, 52, 64, 80, 97, 112, 160};
        for (auto& sz : sizes)
            for (int cin : cins)
                for (int cout : couts) {
                    const Layer L{"edge", cin, cout, 140, 8, 172, 4, 1};
                    Result r1 = run_layer(L, 3, sz[0], sz[1], 3, false, true);
                    ++n; bad += !(r1.maxd < 2e-3);
                }
        // depth_to_space epilogue (x2: 4*C channels -> C), odd image
        for (int ps : {2, 3}) {
            const Layer L{"edge-ps", 40, ps * ps * 8, 40, 0, 8, 0, ps};
            Result r = run_layer(L, 2, 19, 23, 2, false, true);
            ++n; bad += !(r.maxd < 2e-3);
        }
        printf("edge: %d cases, %d mismatches\n", n, bad);
        return bad != 0;
    }
    return 0;
}
