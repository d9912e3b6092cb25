// Launchers of the row-streamed kernels of the separable narrow nets (feat_stream.hpp, tail_stream.hpp).
#include "tail_stream.hpp"

namespace dcscn {

void stream_init_kernels() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&feat_stream), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_stream), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t tail_launch(const TailArgs& a, int grid, hipStream_t stream) {
    const size_t lds = (size_t)a.ring_bytes + a.ldsw_bytes;
    hipLaunchKernelGGL(tail_stream, dim3(grid), dim3(640), lds, stream, a);
    return hipGetLastError();
}

hipError_t stream_launch(const StreamArgs& a, int grid, hipStream_t stream) {
    const int threads = (1 + a.n_conv + a.L) * 64;
    const size_t lds = (size_t)a.ring_bytes + a.ldsw_bytes;
    hipLaunchKernelGGL(feat_stream, dim3(grid), dim3(threads), lds, stream, a);
    return hipGetLastError();
}

}  // namespace dcscn
