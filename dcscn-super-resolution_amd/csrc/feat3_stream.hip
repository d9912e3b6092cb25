// feat3_stream (feat3_stream.hpp): CNN1 .. CNNL of the non-separable narrow nets as one row-streamed launch.
#include "feat3_stream.hpp"

namespace dcscn {

hipError_t stream3_launch(const Stream3Args& a, int grid, hipStream_t stream) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&feat3_stream), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) return attr;
    if (a.n_waves < 2 || a.n_waves > kS3MaxWaves || a.ring_bytes > 160 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(feat3_stream, dim3(grid), dim3(a.n_waves * 64), (size_t)a.ring_bytes, stream, a);
    return hipGetLastError();
}

}  // namespace dcscn
