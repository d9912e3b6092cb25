// feat_stream / tail_stream variants (feat_stream.hpp, tail_stream.hpp), one translation unit.
#include "tail_stream.hpp"

namespace dcscn {

// feat_stream_redo.hip: the gated float32 instantiations (the float32 plan of flagged images)
void stream_redo_init_kernels();
hipError_t tail_redo_launch(const TailArgs& a, int grid, hipStream_t stream);
hipError_t stream_redo_launch(const StreamArgs& a, int grid, hipStream_t stream);

void stream_init_kernels() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&feat_stream<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_stream<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&feat_stream<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_stream<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    stream_redo_init_kernels();
}

hipError_t tail_launch(const TailArgs& a, int grid, bool f16, hipStream_t stream) {
    if (a.redo_check) return tail_redo_launch(a, grid, stream);
    const size_t lds = (size_t)a.ring_bytes + a.ldsw_bytes;
    if (f16) hipLaunchKernelGGL(tail_stream<true>, dim3(grid), dim3(640), lds, stream, a);
    else hipLaunchKernelGGL(tail_stream<false>, dim3(grid), dim3(640), lds, stream, a);
    return hipGetLastError();
}

hipError_t stream_launch(const StreamArgs& a, int grid, bool f16, hipStream_t stream) {
    const int threads = (1 + a.n_conv + a.L) * 64;
    const size_t lds = (size_t)a.ring_bytes + a.ldsw_bytes;
    if (a.redo_check) return stream_redo_launch(a, grid, stream);
    if (f16) hipLaunchKernelGGL(feat_stream<true>, dim3(grid), dim3(threads), lds, stream, a);
    else hipLaunchKernelGGL(feat_stream<false>, dim3(grid), dim3(threads), lds, stream, a);
    return hipGetLastError();
}

}  // namespace dcscn
