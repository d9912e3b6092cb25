// Bicubic resize of single-channel float images, bit-compatible with Pillow's mode-"F" path
// (Image.resize(..., BICUBIC), libImaging/Resample.c: precompute_coeffs + ImagingResampleHorizontal_32bpc /
// ImagingResampleVertical_32bpc), which is what the reference uses for the bicubic residual input x2 and for
// building LR images (helper/utilty.py:211-239 resize_image_by_pil, DCSCN.py:552-554, 682-683).
//
// Pillow: per output index a window [xmin, xmin + n) of input pixels and n <= ksize normalised float64 weights
// (a = -0.5 cubic, support 2 * max(1, in/out): antialiased when shrinking); horizontal pass into a float32
// image, then the vertical pass; each output is a float64 sum  ss += (double)pixel * k[i]  in window order,
// rounded once to float32.  The tables are built on the host with the same expressions; the kernels keep the
// summation order, and this translation unit is compiled with FP contraction OFF: hipcc's device default
// (-ffp-contract=fast) would fuse `ss + p * k` into an FMA -- HIP's __dmul_rn / __dadd_rn are plain * and + and
// do not prevent it -- while Pillow's x86-64 builds have no FMA; the difference is one float32 ulp on ~0.1 % of
// the pixels for non-dyadic scales (x3), none for x2 / x4.
#include <hip/hip_runtime.h>
#pragma clang fp contract(off)

#include <cmath>
#include <vector>

#include "kernels.h"

namespace dcscn {

namespace {

inline double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

}  // namespace

// precompute_coeffs(inSize, 0, inSize, outSize, BICUBIC): bounds[2 * xx] = first input index, [2 * xx + 1] = count
int resample_coeffs(int in_size, int out_size, std::vector<int>* bounds, std::vector<double>* kk) {
    const double in0 = 0.0, in1 = (double)in_size;
    double scale, filterscale;
    filterscale = scale = (in1 - in0) / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale;
    const int ksize = (int)std::ceil(support) * 2 + 1;
    bounds->assign((size_t)out_size * 2, 0);
    kk->assign((size_t)out_size * ksize, 0.0);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = in0 + (xx + 0.5) * scale;
        double ww = 0.0;
        const double ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double* k = &(*kk)[(size_t)xx * ksize];
        for (int x = 0; x < xmax; ++x) {
            const double w = bicubic_filter((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x)
            if (ww != 0.0) k[x] /= ww;
        (*bounds)[(size_t)xx * 2] = xmin;
        (*bounds)[(size_t)xx * 2 + 1] = xmax;
    }
    return ksize;
}

// out[img][y][xx] = (float) sum_i (double) in[img][y][xmin + i] * k[xx][i]
__global__ __launch_bounds__(256) void resample_h_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           const int* __restrict__ bounds, const double* __restrict__ kk,
                                                           int ksize, long long rows, int w, int ow) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * ow) return;
    const int xx = (int)(idx % ow);
    const long long row = idx / ow;
    const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
    const float* p = in + row * w + xmin;
    const double* k = kk + (size_t)xx * ksize;
    double ss = 0.0;
    for (int i = 0; i < n; ++i) ss += (double)p[i] * k[i];
    out[idx] = (float)ss;
}

// out[img][yy][x] = (float) sum_i (double) in[img][ymin + i][x] * k[yy][i]
__global__ __launch_bounds__(256) void resample_v_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           const int* __restrict__ bounds, const double* __restrict__ kk,
                                                           int ksize, int n_img, int h, int oh, int w) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)n_img * oh * w) return;
    const int x = (int)(idx % w);
    const long long t = idx / w;
    const int yy = (int)(t % oh);
    const long long img = t / oh;
    const int ymin = bounds[2 * yy], n = bounds[2 * yy + 1];
    const float* p = in + (img * h + ymin) * w + x;
    const double* k = kk + (size_t)yy * ksize;
    double ss = 0.0;
    for (int i = 0; i < n; ++i) ss += (double)p[(size_t)i * w] * k[i];
    out[idx] = (float)ss;
}

hipError_t resample_h_launch(const float* in, float* out, const int* bounds, const double* kk, int ksize,
                             long long rows, int w, int ow, hipStream_t stream) {
    const long long total = rows * ow;
    hipLaunchKernelGGL(resample_h_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, in, out, bounds, kk, ksize, rows, w, ow);
    return hipGetLastError();
}

hipError_t resample_v_launch(const float* in, float* out, const int* bounds, const double* kk, int ksize,
                             int n_img, int h, int oh, int w, hipStream_t stream) {
    const long long total = (long long)n_img * oh * w;
    hipLaunchKernelGGL(resample_v_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, in, out, bounds, kk, ksize, n_img, h, oh, w);
    return hipGetLastError();
}

}  // namespace dcscn
