// The tile instantiations that live in conv_patch.hip (halo-patch, streaming) and conv_rings.hip (counted-ring and loader-wave implicit-GEMM tiles) -- r06: conv_igemm.hip
// alone was 80 s of a 90-s build.
// ONE list for both translation units: conv_patch.hip defines a plain launcher per entry, conv_igemm.hip's table points at it (a tuple missing here fails at link time).
#pragma once
#include "vgh_internal.h"

#define VGH_PCFG_LIST(X) \
    X(16, 16, 64, 4, 1) \
    X(16, 16, 128, 4, 1) \
    X(16, 16, 128, 4, 2) \
    X(16, 16, 96, 4, 1) \
    X(40, 8, 64, 5, 1) \
    X(40, 8, 128, 5, 2) \
    X(40, 8, 96, 5, 1) \
    X(32, 8, 128, 4, 2) \
    X(32, 8, 64, 4, 1) \
    X(16, 16, 32, 4, 1) \
    X(32, 8, 256, 4, 4) \
    X(32, 16, 128, 8, 2) \
    X(32, 8, 96, 4, 1) \
    X(32, 8, 64, 4, 2) \
    X(16, 16, 256, 4, 4) \
    X(20, 8, 128, 5, 2) \
    X(32, 4, 128, 4, 2) \
    X(32, 16, 128, 4, 2) \
    X(32, 16, 64, 8, 1) \
    X(16, 8, 64, 4, 1) \
    X(16, 8, 128, 4, 2) \
    X(16, 8, 256, 4, 4) \
    X(32, 16, 128, 4, 1) \
    X(40, 16, 128, 4, 1) \
    X(16, 4, 64, 2, 2) \
    X(16, 4, 128, 2, 2) \
    X(16, 4, 128, 2, 4) \
    X(16, 4, 96, 2, 3) \
    X(16, 4, 64, 2, 1)

#define VGH_QCFG_LIST(X) \
    X(16, 16, 64, 4, 1) \
    X(16, 16, 128, 4, 2) \
    X(16, 16, 96, 4, 1) \
    X(32, 8, 96, 4, 1) \
    X(32, 8, 64, 4, 1) \
    X(32, 8, 128, 4, 2) \
    X(40, 8, 64, 5, 1) \
    X(40, 8, 128, 5, 2) \
    X(16, 16, 256, 4, 4) \
    X(32, 8, 256, 4, 4) \
    X(16, 16, 32, 4, 1) \
    X(20, 8, 128, 5, 2) \
    X(32, 16, 128, 4, 2) \
    X(16, 16, 128, 4, 1) \
    X(32, 8, 64, 4, 2) \
    X(32, 4, 128, 4, 2) \
    X(40, 8, 96, 5, 1) \
    X(16, 8, 64, 4, 1) \
    X(16, 8, 128, 4, 2)

#define VGH_TCFG_LIST(X) \
    X(128, 96, 32, 96, 1, 3) \
    X(128, 64, 32, 64, 1, 3) \
    X(128, 64, 32, 64, 1, 4) \
    X(128, 128, 64, 64, 1, 3) \
    X(64, 64, 32, 32, 1, 3) \
    X(64, 96, 32, 96, 1, 3) \
    X(128, 192, 64, 96, 1, 3) \
    X(128, 32, 32, 32, 1, 3) \
    X(128, 96, 32, 96, 2, 3) \
    X(128, 64, 32, 64, 2, 3) \
    X(256, 64, 64, 64, 1, 3) \
    X(128, 96, 32, 96, 1, 4)

#define VGH_RCFG_LIST(X) \
    X(128, 128, 64, 64, 1, 3) \
    X(128, 128, 64, 64, 1, 4) \
    X(128, 64, 32, 64, 1, 4) \
    X(256, 64, 64, 64, 1, 3) \
    X(128, 96, 32, 96, 1, 4) \
    X(64, 128, 32, 64, 1, 4) \
    X(64, 64, 32, 32, 1, 4) \
    X(128, 128, 64, 64, 2, 3) \
    X(128, 64, 32, 64, 2, 3) \
    X(128, 32, 32, 32, 1, 4) \
    X(256, 128, 64, 64, 1, 3) \
    X(256, 128, 64, 64, 1, 2) \
    X(256, 128, 64, 64, 2, 2) \
    X(256, 256, 64, 64, 1, 2) \
    X(256, 256, 64, 64, 1, 3) \
    X(512, 128, 64, 64, 1, 2) \
    X(256, 64, 32, 64, 1, 2) \
    X(256, 64, 32, 64, 1, 3) \
    X(256, 96, 32, 96, 1, 2) \
    X(128, 128, 32, 64, 1, 2) \
    X(128, 128, 32, 64, 1, 3) \
    X(512, 64, 64, 64, 1, 2) \
    X(256, 128, 128, 64, 1, 2) \
    X(256, 128, 128, 64, 1, 3) \
    X(256, 128, 64, 128, 1, 2) \
    X(256, 64, 128, 64, 1, 2) \
    X(512, 128, 128, 64, 1, 2) \
    X(256, 256, 128, 64, 1, 2) \
    X(256, 256, 64, 128, 1, 2) \
    X(256, 192, 64, 96, 1, 3) \
    X(128, 192, 32, 96, 1, 3) \
    X(64, 64, 32, 32, 4, 3) \
    X(64, 64, 32, 32, 4, 4) \
    X(64, 64, 32, 32, 2, 4) \
    X(64, 32, 32, 32, 4, 4) \
    X(128, 32, 32, 32, 4, 3) \
    X(128, 64, 32, 64, 2, 4) \
    X(128, 96, 32, 96, 1, 3) \
    X(128, 192, 32, 96, 1, 4)

#define VGH_LCFG_LIST(X) \
    X(64, 64, 32, 32, 4, 3, 2) \
    X(64, 64, 32, 32, 4, 3, 4) \
    X(64, 64, 32, 32, 4, 4, 4) \
    X(64, 64, 32, 32, 2, 4, 4) \
    X(64, 32, 32, 32, 4, 3, 3) \
    X(64, 32, 32, 32, 4, 3, 6) \
    X(32, 64, 32, 32, 4, 3, 3) \
    X(32, 64, 32, 32, 4, 3, 6) \
    X(128, 64, 32, 64, 2, 3, 3)

namespace vghcfg {
#define VGH_DECL_P(TW, TH, BC, NWP, NWC) __attribute__((visibility("hidden"))) void lp_##TW##_##TH##_##BC##_##NWP##_##NWC(const ConvArgs&, int, int, int, int, int, int, hipStream_t);
#define VGH_DECL_Q(TW, TH, BC, NWP, NWC) __attribute__((visibility("hidden"))) void lq_##TW##_##TH##_##BC##_##NWP##_##NWC(const ConvArgs&, int, int, int, int, int, int, hipStream_t);
#define VGH_DECL_T(BP, BC, WP, WC, KBS, NST) __attribute__((visibility("hidden"))) void lt_##BP##_##BC##_##WP##_##WC##_##KBS##_##NST(const ConvArgs&, int, int, int, int, hipStream_t);
#define VGH_DECL_R(BP, BC, WP, WC, KBS, NST) __attribute__((visibility("hidden"))) void lr_##BP##_##BC##_##WP##_##WC##_##KBS##_##NST(const ConvArgs&, int, int, int, int, hipStream_t);
#define VGH_DECL_L(BP, BC, WP, WC, KBS, NST, LF) __attribute__((visibility("hidden"))) void ll_##BP##_##BC##_##WP##_##WC##_##KBS##_##NST##_##LF(const ConvArgs&, int, int, int, int, hipStream_t);
VGH_RCFG_LIST(VGH_DECL_R)
VGH_LCFG_LIST(VGH_DECL_L)
#undef VGH_DECL_R
#undef VGH_DECL_L
VGH_PCFG_LIST(VGH_DECL_P)
VGH_QCFG_LIST(VGH_DECL_Q)
VGH_TCFG_LIST(VGH_DECL_T)
#undef VGH_DECL_P
#undef VGH_DECL_Q
#undef VGH_DECL_T
}  // namespace vghcfg

// Per-device launch state: the >64 KiB dynamic-LDS opt-in and the occupancy query act on the CURRENT device, so they are cached
// per device id (a second engine on another GPU of the same process needs its own opt-in).  Racing threads compute the same value.
namespace {
constexpr int kMaxDevices = 16;
inline int current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) d = 0;
    return d;
}

template <typename K>
inline int patch_blocks_per_cu(K kernel, int threads, int lds, std::atomic<int> (&cache)[kMaxDevices]) {
    const int dev = current_device();
    int n = cache[dev].load(std::memory_order_acquire);
    if (n == 0) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)kernel, threads, lds) != hipSuccess || n < 1) n = 1;
        cache[dev].store(n, std::memory_order_release);
    }
    return n;
}
}  // namespace
