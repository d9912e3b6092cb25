// 1x1 (pointwise) implicit-GEMM variants, with and without the fused depthwise stage.
#include "conv_variants.hpp"

namespace dcscn {

hipError_t conv_init_k1() {
    hipError_t e;
#define X(KS, NT) if ((e = Variant<KS, NT>::set_attr()) != hipSuccess) return e;
    DCSCN_FOR_NT(X, 1)
#undef X
#define X(KS, NT, DWK) if ((e = Variant<KS, NT, DWK>::set_attr()) != hipSuccess) return e;
    DCSCN_FOR_NT_DW(X, 3)      // 1x1 depthwise halves are folded into the pointwise weights (api.hip: ColSeg::dw1)
#undef X
    return hipSuccess;
}

hipError_t conv_launch_k1(int nt, int dwk, const ConvArgs& a, int n_tiles, hipStream_t stream) {
    if (dwk != 0) {
        switch (dwk * 100 + nt) {
#define X(KS, NT, DWK) case DWK * 100 + NT: return Variant<KS, NT, DWK>::launch(a, n_tiles, stream);
            DCSCN_FOR_NT_DW(X, 3)
#undef X
            default: return hipErrorInvalidValue;
        }
    }
    switch (nt) {
#define X(KS, NT) case NT: return Variant<KS, NT>::launch(a, n_tiles, stream);
        DCSCN_FOR_NT(X, 1)
#undef X
        default: return hipErrorInvalidValue;
    }
}

}  // namespace dcscn
