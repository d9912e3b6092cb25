// Self-ensemble on the device (DCSCN.py:559-573, util.flip helper/utilty.py:595-617): the n <= 8 flipped / rotated
// copies of an image are gathered by one kernel, and the restored outputs are summed in float64 in the reference's
// order (output += restored_i for i ascending, then output /= n) by another -- the host never touches a pixel.
//
// Flip types: 0 identity, 1 flipud, 2 fliplr, 3 both, 4 rot90(+1), 5 rot90(-1), 6 flipud(rot90(+1)) = transpose,
// 7 flipud(rot90(-1)) = anti-transpose.  Types 0-3 keep the shape [h, w], types 4-7 are [w, h]; the two groups are
// two batches for the network.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace dcscn {

namespace {

// transformed[t][r][c] = image[sr][sc]   (h, w: shape of the untransformed image)
__device__ __forceinline__ void flip_src_dev(int t, int h, int w, int r, int c, int* sr, int* sc) {
    switch (t) {
        case 0: *sr = r; *sc = c; break;
        case 1: *sr = h - 1 - r; *sc = c; break;
        case 2: *sr = r; *sc = w - 1 - c; break;
        case 3: *sr = h - 1 - r; *sc = w - 1 - c; break;
        case 4: *sr = c; *sc = w - 1 - r; break;
        case 5: *sr = h - 1 - c; *sc = r; break;
        case 6: *sr = c; *sc = r; break;
        default: *sr = h - 1 - c; *sc = w - 1 - r; break;
    }
}

// position (r, c) in transformed image t that holds original pixel (R, C): the inverse of the map above
__device__ __forceinline__ void flip_dst_dev(int t, int h, int w, int R, int C, int* r, int* c) {
    switch (t) {
        case 0: *r = R; *c = C; break;
        case 1: *r = h - 1 - R; *c = C; break;
        case 2: *r = R; *c = w - 1 - C; break;
        case 3: *r = h - 1 - R; *c = w - 1 - C; break;
        case 4: *r = w - 1 - C; *c = R; break;
        case 5: *r = C; *c = h - 1 - R; break;
        case 6: *r = C; *c = R; break;
        default: *r = w - 1 - C; *c = h - 1 - R; break;
    }
}

}  // namespace

// out: n images of h*w pixels each, image t in the shape of its flip type (all the same pixel count)
__global__ __launch_bounds__(256) void ensemble_gather_kernel(const float* __restrict__ in, float* __restrict__ out, int h, int w, int n) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long per = (long long)h * w;
    if (idx >= per * n) return;
    const int t = (int)(idx / per);
    const int p = (int)(idx - t * per);
    const int tw = t >= 4 ? h : w;
    const int r = p / tw, c = p - r * tw;
    int sr, sc;
    flip_src_dev(t, h, w, r, c, &sr, &sc);
    out[idx] = in[(size_t)sr * w + sc];
}

// out[R][C] = (sum_t (double) y_t[inverse position]) / n, t ascending; y: n images of h*w floats as above
__global__ __launch_bounds__(256) void ensemble_reduce_kernel(const float* __restrict__ y, double* __restrict__ out, int h, int w, int n) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long per = (long long)h * w;
    if (idx >= per) return;
    const int R = (int)(idx / w), C = (int)(idx - (long long)R * w);
    double acc = 0.0;
    for (int t = 0; t < n; ++t) {
        int r, c;
        flip_dst_dev(t, h, w, R, C, &r, &c);
        const int tw = t >= 4 ? h : w;
        acc += (double)y[(size_t)t * per + (size_t)r * tw + c];
    }
    out[idx] = n > 1 ? acc / (double)n : acc;
}

hipError_t ensemble_gather_launch(const float* in, float* out, int h, int w, int n, hipStream_t stream) {
    const long long total = (long long)h * w * n;
    hipLaunchKernelGGL(ensemble_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, in, out, h, w, n);
    return hipGetLastError();
}

hipError_t ensemble_reduce_launch(const float* y, double* out, int h, int w, int n, hipStream_t stream) {
    const long long total = (long long)h * w;
    hipLaunchKernelGGL(ensemble_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, y, out, h, w, n);
    return hipGetLastError();
}

}  // namespace dcscn
