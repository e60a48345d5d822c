// Launchers of the counted-ring (NST = 3 / 4) and loader-wave implicit-GEMM tile instantiations (kernel: conv_kernels.inc).  Split off conv_igemm.hip in r06 (build time);
// conv_igemm.hip owns the table and calls these through conv_cfg_list.h.
#include "conv_kernels.inc"
#include "conv_cfg_list.h"

namespace {
#include "conv_launch_igemm.inc"
}  // namespace

namespace vghcfg {
#define VGH_DEF_R(BP, BC, WP, WC, KBS, NST)                                                                                        \
    void lr_##BP##_##BC##_##WP##_##WC##_##KBS##_##NST(const ConvArgs& a, int ntc, int total, int chunk, int lds, hipStream_t st) { \
        launch_cfg<BP, BC, WP, WC, KBS, NST>(a, ntc, total, chunk, lds, st);                                                       \
    }
#define VGH_DEF_L(BP, BC, WP, WC, KBS, NST, LF)                                                                                             \
    void ll_##BP##_##BC##_##WP##_##WC##_##KBS##_##NST##_##LF(const ConvArgs& a, int ntc, int total, int chunk, int lds, hipStream_t st) { \
        launch_cfg_lf<BP, BC, WP, WC, KBS, NST, LF>(a, ntc, total, chunk, lds, st);                                                        \
    }
VGH_RCFG_LIST(VGH_DEF_R)
VGH_LCFG_LIST(VGH_DEF_L)
}  // namespace vghcfg
