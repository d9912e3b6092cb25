// Colour conversions of the evaluation / sr.py path on the device (helper/utilty.py:142-193 of the reference):
//   convert_rgb_to_y          Y  = [65.738 129.057 25.064] / 256 . RGB + 16          (float64, not rounded)
//   convert_rgb_to_ycbcr      YCbCr = M . RGB + [16 128 128]
//   convert_y_and_cbcr_to_rgb RGB = R . ([Y Cb Cr] - [16 128 128])
// The reference evaluates them as numpy `image.dot(M.T)` in float64.  numpy's matmul on this stack (OpenBLAS, FMA
// kernels) computes each 3-term dot product as fma(c2, b, fma(c1, g, c0 * r)); the kernels below use exactly that
// chain, so device and host results are bit-identical (tests/test_color_hip.py) -- and differ from any other
// summation order by at most one ulp of float64, far below what the following float32 cast / rint can see.
// HBM-bound elementwise work: one thread per pixel, 3-byte reads, 8-byte stores.
#include "kernels.h"

namespace dcscn {

__constant__ double kYCbCr[3][3] = {{65.738 / 256.0, 129.057 / 256.0, 25.064 / 256.0},
                                    {-37.945 / 256.0, -74.494 / 256.0, 112.439 / 256.0},
                                    {112.439 / 256.0, -94.154 / 256.0, -18.285 / 256.0}};
__constant__ double kRgb[3][3] = {{298.082 / 256.0, 0.0, 408.583 / 256.0},
                                  {298.082 / 256.0, -100.291 / 256.0, -208.120 / 256.0},
                                  {298.082 / 256.0, 516.412 / 256.0, 0.0}};

__device__ __forceinline__ double dot3(const double (&m)[3], double a, double b, double c) {
    return __fma_rn(c, m[2], __fma_rn(b, m[1], a * m[0]));
}

__global__ __launch_bounds__(256) void rgb_to_y_kernel(const uint8_t* __restrict__ rgb, double* __restrict__ y64, float* __restrict__ y32, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double r = rgb[3 * i], g = rgb[3 * i + 1], b = rgb[3 * i + 2];
    const double y = dot3(kYCbCr[0], r, g, b) + 16.0;
    if (y64) y64[i] = y;
    if (y32) y32[i] = (float)y;
}

__global__ __launch_bounds__(256) void rgb_to_ycbcr_kernel(const uint8_t* __restrict__ rgb, double* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double r = rgb[3 * i], g = rgb[3 * i + 1], b = rgb[3 * i + 2];
    out[3 * i + 0] = dot3(kYCbCr[0], r, g, b) + 16.0;
    out[3 * i + 1] = dot3(kYCbCr[1], r, g, b) + 128.0;
    out[3 * i + 2] = dot3(kYCbCr[2], r, g, b) + 128.0;
}

// y: one value per pixel (float64, or float32 when y32 != nullptr -- the network output); cbcr: [n][2] float64, or
// taken from `rgb8` (uint8 RGB, e.g. the Pillow-upscaled colour image) when cbcr == nullptr
__global__ __launch_bounds__(256) void y_cbcr_to_rgb_kernel(const double* __restrict__ y64, const float* __restrict__ y32,
                                                           const double* __restrict__ cbcr, const uint8_t* __restrict__ rgb8,
                                                           double* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double y = y32 ? (double)y32[i] : y64[i];
    double cb, cr;
    if (cbcr) {
        cb = cbcr[2 * i];
        cr = cbcr[2 * i + 1];
    } else {
        const double r = rgb8[3 * i], g = rgb8[3 * i + 1], b = rgb8[3 * i + 2];
        cb = dot3(kYCbCr[1], r, g, b) + 128.0;
        cr = dot3(kYCbCr[2], r, g, b) + 128.0;
    }
    const double s0 = y - 16.0, s1 = cb - 128.0, s2 = cr - 128.0;
    out[3 * i + 0] = dot3(kRgb[0], s0, s1, s2);
    out[3 * i + 1] = dot3(kRgb[1], s0, s1, s2);
    out[3 * i + 2] = dot3(kRgb[2], s0, s1, s2);
}

static dim3 grid_for(long long n) { return dim3((unsigned)((n + 255) / 256)); }

hipError_t rgb_to_y_launch(const uint8_t* rgb, double* y64, float* y32, long long n, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(rgb_to_y_kernel, grid_for(n), dim3(256), 0, stream, rgb, y64, y32, n);
    return hipGetLastError();
}

hipError_t rgb_to_ycbcr_launch(const uint8_t* rgb, double* out, long long n, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(rgb_to_ycbcr_kernel, grid_for(n), dim3(256), 0, stream, rgb, out, n);
    return hipGetLastError();
}

hipError_t y_cbcr_to_rgb_launch(const double* y64, const float* y32, const double* cbcr, const uint8_t* rgb8, double* out, long long n,
                                hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(y_cbcr_to_rgb_kernel, grid_for(n), dim3(256), 0, stream, y64, y32, cbcr, rgb8, out, n);
    return hipGetLastError();
}

}  // namespace dcscn
