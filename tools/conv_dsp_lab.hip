// Launcher of the conv_dsp lab kernel (tools/conv_dsp_lab.hpp); not part of the library build.
#include "conv_dsp_lab.hpp"

namespace dcscn {

template <int NT, int CK>
static hipError_t dsp_set_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_dsp<NT, CK>), hipFuncAttributeMaxDynamicSharedMemorySize, DspGeom<NT, CK>::LDS_BYTES);
}

#define DCSCN_DSP_ALL(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8)

hipError_t dsp_init_kernels() {
    hipError_t e = hipSuccess;
#define X(NT) \
    if (e == hipSuccess) e = dsp_set_attr<NT, 16>(); \
    if (e == hipSuccess) e = dsp_set_attr<NT, 32>();
    DCSCN_DSP_ALL(X)
#undef X
    return e;
}

static int compute_units() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

template <int NT, int CK>
static hipError_t dsp_launch_one(const ConvArgs& a, hipStream_t stream) {
    using G = DspGeom<NT, CK>;
    const long long tiles = (long long)a.N * a.tiles_y * a.tiles_x;
    int per_cu = (160 * 1024) / G::LDS_BYTES;                 // resident workgroups per CU (persistent: one tile stream each)
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
    long long grid = (long long)compute_units() * per_cu;
    if (grid > tiles) grid = tiles;
    hipLaunchKernelGGL((conv_dsp<NT, CK>), dim3((unsigned)grid), dim3(256), G::LDS_BYTES, stream, a);
    return hipGetLastError();
}

hipError_t dsp_launch(int nt, int ck, const ConvArgs& a, hipStream_t stream) {
    if (a.dww == nullptr || a.dwk != 3 || (ck != 16 && ck != 32) || a.cin_phys > ck) return hipErrorInvalidValue;
    switch (nt) {
#define X(NT) case NT: return ck == 16 ? dsp_launch_one<NT, 16>(a, stream) : dsp_launch_one<NT, 32>(a, stream);
        DCSCN_DSP_ALL(X)
#undef X
        default: return hipErrorInvalidValue;
    }
}

}  // namespace dcscn
