// 7x7 direct implicit-GEMM variants (conv_igemm<7, ...>): --cnn_size=7 models.
#include "conv_variants.hpp"

namespace dcscn {

hipError_t conv_init_k7() {
    hipError_t e;
#define X(KS, NT) if ((e = Variant<KS, NT>::set_attr()) != hipSuccess) return e;
    DCSCN_FOR_NT_K7(X)
#undef X
    return hipSuccess;
}

hipError_t conv_launch_k7(int nt, const ConvArgs& a, int n_tiles, hipStream_t stream) {
    switch (nt) {
#define X(KS, NT) case NT: return Variant<KS, NT>::launch(a, n_tiles, stream);
        DCSCN_FOR_NT_K7(X)
#undef X
        default: return hipErrorInvalidValue;
    }
}

}  // namespace dcscn
