// 5x5 direct implicit-GEMM variants (conv_igemm<5, ...>): --cnn_size=5 models, and the folded linear tail of
// the network (pixel-shuffler conv -> depth_to_space -> last reconstruction conv as ONE 5x5 conv, api.hip:
// fold_linear_tail).
#include "conv_variants.hpp"

namespace dcscn {

hipError_t conv_init_k5() {
    hipError_t e;
#define X(KS, NT) if ((e = Variant<KS, NT>::set_attr()) != hipSuccess) return e;
    DCSCN_FOR_NT(X, 5)
#undef X
    return hipSuccess;
}

hipError_t conv_launch_k5(int nt, const ConvArgs& a, int n_tiles, hipStream_t stream) {
    switch (nt) {
#define X(KS, NT) case NT: return Variant<KS, NT>::launch(a, n_tiles, stream);
        DCSCN_FOR_NT(X, 5)
#undef X
        default: return hipErrorInvalidValue;
    }
}

}  // namespace dcscn
