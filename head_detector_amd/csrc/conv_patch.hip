// Launchers of the halo-patch ("p" / "q") and streaming 1x1 ("t") tile instantiations (kernels: conv_kernels.inc).  Split off conv_igemm.hip in r06 so that the two halves of
// the tile table compile side by side; conv_igemm.hip owns the table (names, eligibility, dispatch) and calls these through conv_cfg_list.h.
#include "conv_kernels.inc"
#include "conv_cfg_list.h"

namespace {
template <int TW, int TH, int BC, int NWP, int NWC>
void launch_patch_cfg(const ConvArgs& a, int ntc, int ntx, int nty, int total, int chunk, int lds, hipStream_t st) {
    static std::atomic<int> per_cu[kMaxDevices];
    const int n = patch_blocks_per_cu(conv3x3_patch_kernel<TW, TH, BC, NWP, NWC>, NWP * NWC * 64, lds, per_cu);
    const int gpx = vgh_conv_persistent_blocks_per_xcd(a, chunk, n);
    hipLaunchKernelGGL((conv3x3_patch_kernel<TW, TH, BC, NWP, NWC>), dim3(gpx * 8), dim3(NWP * NWC * 64), lds, st, a, ntc, ntx, nty, total, chunk);
}

template <int TW, int TH, int BC, int NWP, int NWC>
void launch_patch3_cfg(const ConvArgs& a, int ntc, int ntx, int nty, int total, int chunk, int lds, hipStream_t st) {
    static std::atomic<int> per_cu[kMaxDevices];
    const int n = patch_blocks_per_cu(conv3x3_patch3_kernel<TW, TH, BC, NWP, NWC>, NWP * NWC * 64, lds, per_cu);
    const int gpx = vgh_conv_persistent_blocks_per_xcd(a, chunk, n);
    hipLaunchKernelGGL((conv3x3_patch3_kernel<TW, TH, BC, NWP, NWC>), dim3(gpx * 8), dim3(NWP * NWC * 64), lds, st, a, ntc, ntx, nty, total, chunk);
}

template <int BP, int BC, int WP, int WC, int KBS, int NST>
void launch_stream_cfg(const ConvArgs& a, int ntc, int total, int chunk, int lds, hipStream_t st) {
    static std::atomic<int> per_cu[kMaxDevices];
    constexpr int threads = (BP / WP) * (BC / WC) * 64;
    const int n = patch_blocks_per_cu(conv1x1_stream_kernel<BP, BC, WP, WC, KBS, NST>, threads, lds, per_cu);
    const int gpx = vgh_conv_persistent_blocks_per_xcd(a, chunk, n);
    hipLaunchKernelGGL((conv1x1_stream_kernel<BP, BC, WP, WC, KBS, NST>), dim3(gpx * 8), dim3(threads), lds, st, a, ntc, total, chunk);
}
}  // namespace

namespace vghcfg {
#define VGH_DEF_P(TW, TH, BC, NWP, NWC)                                                                                                               \
    void lp_##TW##_##TH##_##BC##_##NWP##_##NWC(const ConvArgs& a, int ntc, int ntx, int nty, int total, int chunk, int lds, hipStream_t st) {        \
        launch_patch_cfg<TW, TH, BC, NWP, NWC>(a, ntc, ntx, nty, total, chunk, lds, st);                                                              \
    }
#define VGH_DEF_Q(TW, TH, BC, NWP, NWC)                                                                                                               \
    void lq_##TW##_##TH##_##BC##_##NWP##_##NWC(const ConvArgs& a, int ntc, int ntx, int nty, int total, int chunk, int lds, hipStream_t st) {        \
        launch_patch3_cfg<TW, TH, BC, NWP, NWC>(a, ntc, ntx, nty, total, chunk, lds, st);                                                             \
    }
#define VGH_DEF_T(BP, BC, WP, WC, KBS, NST)                                                                                        \
    void lt_##BP##_##BC##_##WP##_##WC##_##KBS##_##NST(const ConvArgs& a, int ntc, int total, int chunk, int lds, hipStream_t st) { \
        launch_stream_cfg<BP, BC, WP, WC, KBS, NST>(a, ntc, total, chunk, lds, st);                                                \
    }
VGH_PCFG_LIST(VGH_DEF_P)
VGH_QCFG_LIST(VGH_DEF_Q)
VGH_TCFG_LIST(VGH_DEF_T)
}  // namespace vghcfg
