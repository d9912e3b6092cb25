// feat3_stream (feat3_stream.hpp): CNN1 .. CNNL of the non-separable narrow nets as one row-streamed launch.
#include "feat3_stream.hpp"

namespace dcscn {

hipError_t stream3_launch(const Stream3Args& a, int grid, hipStream_t stream) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&feat3_stream), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) return attr;
    if (a.n_waves < 2 || a.n_waves > kS3MaxWaves || a.ring_bytes > 160 * 1024) return hipErrorInvalidValue;
#ifdef S3_DBG
    static long long* d_dbg = nullptr;
    Stream3Args b = a;
    if (!d_dbg) (void)hipMalloc((void**)&d_dbg, 8 * 4 * sizeof(long long));
    b.dbg = d_dbg;
    hipLaunchKernelGGL(feat3_stream, dim3(grid), dim3(a.n_waves * 64), (size_t)a.ring_bytes + 256, stream, b);
    if (getenv("DCSCN_S3_DBG")) {
        long long host[32];
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(host, d_dbg, sizeof host, hipMemcpyDeviceToHost);
        for (int w = 0; w < a.n_waves; ++w)
            fprintf(stderr, "S3_DBG wave %d (conv %d): steps %lld, per step: compute %.0f cycles, barrier wait %.0f\n", w, (int)a.role_conv[w], host[w * 4 + 2],
                    (double)host[w * 4] / (double)host[w * 4 + 2], (double)host[w * 4 + 1] / (double)host[w * 4 + 2]);
    }
    return hipGetLastError();
#endif
    hipLaunchKernelGGL(feat3_stream, dim3(grid), dim3(a.n_waves * 64), (size_t)a.ring_bytes + 256, stream, a);
    return hipGetLastError();
}

}  // namespace dcscn
