// Variant exploration for conv_igemm: times chosen (MT, KC, DB, WPS) instantiations on the layer shapes
// of the benchmark model.  Not part of the product; build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/conv_tune.hip -o /tmp/conv_tune && /tmp/conv_tune [set]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../dcscn-super-resolution_amd/csrc/conv_igemm.hpp"

using namespace dcscn;

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

struct Layer {
    const char* name;
    int ks, cin, cout, in_stride, in_off, out_stride, out_off;
};

static float* g_in = nullptr;
static float* g_out = nullptr;
static float* g_w = nullptr;
static float* g_bias = nullptr;
static int N = 1024, H = 48, W = 48;

template <int KS, int MT, int NT, int KC, bool DB, int WPS>
void run(const Layer& L, int n_tiles) {
    using G = ConvGeom<KS, MT, NT, KC>;
    if (KS != L.ks) return;
    const int cin_phys = (L.cin + 3) & ~3;
    ConvArgs a{};
    a.in = g_in;
    a.in_stride = L.in_stride;
    a.in_off = L.in_off;
    a.cin_phys = cin_phys;
    a.n_chunks = (cin_phys + KC - 1) / KC;
    a.wpack = g_w;
    a.bias = g_bias;
    a.alpha = g_bias;
    a.act = ACT_ALPHA;
    a.N = N; a.H = H; a.W = W;
    a.tiles_x = (W + 15) / 16;
    a.tiles_y = (H + G::TH - 1) / G::TH;
    a.out0 = OutDesc{g_out, L.out_stride, L.out_off, ((L.cout + 3) & ~3)};
    a.out1 = a.out0;
    a.split = 1 << 30;
    a.ps = 1; a.ps_c = 1; a.vec4 = 1;
    a.res = nullptr; a.res_stride = 1;
    const size_t lds = (DB ? 2 : 1) * (size_t)G::BUF * sizeof(float);
    auto kern = conv_igemm<KS, MT, NT, KC, DB, WPS>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, lds));
    const dim3 grid((unsigned)(N * a.tiles_y * a.tiles_x), (unsigned)n_tiles);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, a);
    CK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0;
    const int reps = 5;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, a);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
        sum += ms;
    }
    CK(hipGetLastError());
    const double flop = 2.0 * KS * KS * L.cin * (double)L.cout * N * H * W;
    printf("%-10s k%d %4d->%-4d  MT%d NT%-2d KC%-2d DB%d WPS%d tiles%d  lds %6.1f KB occ %d  best %8.3f ms avg %8.3f  %7.2f TFLOP/s\n",
           L.name, KS, L.cin, L.cout, MT, NT, KC, (int)DB, WPS, n_tiles, lds / 1024.0, occ, best, sum / reps,
           flop / (best * 1e-3) / 1e12);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const char* which = argc > 1 ? argv[1] : "all";
    const size_t in_floats = (size_t)N * H * W * 1316;
    const size_t out_floats = (size_t)N * H * W * 1316;
    CK(hipMalloc(&g_in, in_floats * sizeof(float)));
    CK(hipMalloc(&g_out, out_floats * sizeof(float)));
    const size_t w_floats = 16u << 20;
    CK(hipMalloc(&g_w, w_floats * sizeof(float)));
    CK(hipMalloc(&g_bias, 4096 * sizeof(float)));
    {
        std::vector<float> h(w_floats);
        unsigned s = 12345;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
        CK(hipMemcpy(g_w, h.data(), w_floats * sizeof(float), hipMemcpyHostToDevice));
        CK(hipMemcpy(g_bias, h.data(), 4096 * sizeof(float), hipMemcpyHostToDevice));
        // activations: tile the random block over the input
        for (size_t off = 0; off < in_floats; off += w_floats) {
            const size_t n = std::min(w_floats, in_floats - off);
            CK(hipMemcpy(g_in + off, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    CK(hipMemset(g_out, 0, out_floats * sizeof(float)));

    const Layer cnn2{"CNN2", 3, 196, 166, 1316, 0, 1316, 196};
    const Layer cnn3{"CNN3", 3, 166, 148, 1316, 196, 1316, 364};
    const Layer cnn5{"CNN5", 3, 133, 120, 1316, 512, 1316, 648};
    const Layer cnn7{"CNN7", 3, 108, 97, 1316, 876, 1316, 984};
    const Layer cnn10{"CNN10", 3, 76, 66, 1316, 1172, 1316, 1248};
    const Layer cnn12{"CNN12", 3, 57, 48, 1316, 1248, 1316, 1268};
    const Layer upps{"Up-PS", 3, 96, 384, 96, 0, 384, 0};
    const Layer nin{"B1+A1", 1, 1316, 96, 1316, 0, 96, 0};

    const Layer cnn4{"CNN4", 3, 148, 133, 1316, 364, 1316, 512};
    const Layer cnn6{"CNN6", 3, 120, 108, 1316, 648, 1316, 768};
    const Layer cnn8{"CNN8", 3, 97, 86, 1316, 984, 1316, 1084};
    const Layer cnn9{"CNN9", 3, 86, 76, 1316, 1084, 1316, 1172};
    const Layer cnn11{"CNN11", 3, 66, 57, 1316, 1248, 1316, 1188};
    const Layer b2{"B2", 3, 32, 32, 32, 0, 96, 0};
    auto want = [&](const char* k) { return !strcmp(which, "all") || !strcmp(which, k); };
    (void)cnn2; (void)cnn3; (void)cnn5; (void)cnn7; (void)cnn10; (void)upps; (void)nin;
    if (want("cnn4")) {
        run<3, 2, 9, 4, false, 3>(cnn4, 1);
        run<3, 3, 9, 4, false, 3>(cnn4, 1);
    }
    if (want("cnn6")) {
        run<3, 2, 7, 4, false, 4>(cnn6, 1);
        run<3, 3, 7, 4, false, 3>(cnn6, 1);
    }
    if (want("cnn8")) {
        run<3, 2, 6, 4, false, 4>(cnn8, 1);
        run<3, 3, 6, 4, false, 4>(cnn8, 1);
        run<3, 4, 6, 4, false, 3>(cnn8, 1);
        run<3, 2, 6, 8, false, 4>(cnn8, 1);
    }
    if (want("cnn9")) {
        run<3, 2, 5, 4, false, 4>(cnn9, 1);
        run<3, 3, 5, 4, false, 4>(cnn9, 1);
        run<3, 4, 5, 4, false, 3>(cnn9, 1);
        run<3, 4, 5, 4, false, 4>(cnn9, 1);
    }
    if (want("cnn11")) {
        run<3, 2, 4, 4, false, 4>(cnn11, 1);
        run<3, 2, 4, 8, false, 4>(cnn11, 1);
        run<3, 3, 4, 4, false, 4>(cnn11, 1);
        run<3, 4, 4, 4, false, 4>(cnn11, 1);
        run<3, 4, 4, 8, false, 3>(cnn11, 1);
    }
    if (want("cnn12")) {
        run<3, 2, 3, 4, false, 4>(cnn12, 1);
        run<3, 2, 3, 8, false, 4>(cnn12, 1);
        run<3, 3, 3, 4, false, 4>(cnn12, 1);
        run<3, 4, 3, 4, false, 4>(cnn12, 1);
    }
    if (want("b2")) {
        run<3, 2, 2, 4, false, 4>(b2, 1);
        run<3, 2, 2, 8, false, 4>(b2, 1);
        run<3, 4, 2, 4, false, 4>(b2, 1);
        run<3, 4, 2, 8, false, 4>(b2, 1);
        run<3, 4, 2, 8, true, 2>(b2, 1);
    }
    return 0;
}
