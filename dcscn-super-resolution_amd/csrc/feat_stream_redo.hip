// The gated float32 instantiations of the streamed kernels (feat_stream.hpp / tail_stream.hpp: GATE): the float32 plan behind a pass
// whose F16 launch flagged an image.  A translation unit of its own: these may spill (rare path), the kernels of every pass may not.
#include "tail_stream.hpp"

namespace dcscn {

void stream_redo_init_kernels() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&feat_stream<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_stream<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t tail_redo_launch(const TailArgs& a, int grid, hipStream_t stream) {
    const size_t lds = (size_t)a.ring_bytes + a.ldsw_bytes;
    hipLaunchKernelGGL((tail_stream<false, true>), dim3(grid), dim3(640), lds, stream, a);
    return hipGetLastError();
}

hipError_t stream_redo_launch(const StreamArgs& a, int grid, hipStream_t stream) {
    const int threads = (1 + a.n_conv + a.L) * 64;
    const size_t lds = (size_t)a.ring_bytes + a.ldsw_bytes;
    hipLaunchKernelGGL((feat_stream<false, true>), dim3(grid), dim3(threads), lds, stream, a);
    return hipGetLastError();
}

}  // namespace dcscn
