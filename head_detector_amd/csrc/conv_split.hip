// Split-precision convolutions: the MATRIX-CORE parity path (north_star: "bbox IoU >= 0.999, FLAME params / vertices within 1e-4"
// against the reference's fp32 CPU network, head_detector/detector.py:58-59).
//
// bf16 storage (8 significand bits) cannot meet that bar and gfx950 has no fast fp32 matrix format (v_mfma_f32_32x32x2_f32 runs at the
// VALU rate, 1/16 of the 16-bit MFMAs).  These kernels keep every activation and weight as TWO 16-bit planes, v = hi + lo / L, and
// compute each product with three 16-bit MFMAs,
//        x * w  ~=  x_hi * w_hi  +  (x_hi * w_lo + x_lo * w_hi) / L          (the dropped x_lo * w_lo is 2^-2p relative),
// accumulated in fp32 -- the error-corrected tensor-core GEMM of Ootomo & Yokota restated for MFMA:
//   VGH_FMT_F16X2   fp16 planes, L = 2^11: 22 significand bits per operand, dropped term 2^-22  -> fp32-class results;
//                   per-op power-of-two weight prescale keeps the weight planes inside fp16's normal range (undone on the accumulator);
//   VGH_FMT_BF16X2  bf16 planes, L = 1:    16 significand bits, dropped term 2^-16              -> kept for the comparison table in DESIGN.md.
// The correction products are accumulated FIRST (two K segments), scaled once by 1/L (exact), and the main products accumulate on
// top, so one accumulator set serves both and the fp32 rounding behaviour of the main sum is the ordinary one.
//
// The kernels are the tiles of conv_kernels.inc (implicit-GEMM + 3x3 halo-patch) instantiated with SP != 0: same LDS-DMA loaders, same
// swizzles, same epilogue; the K loop is three times as long and the epilogue stores / reads two planes.  Tile choice is a small
// size rule (no tuning table: this is the parity mode, its throughput is reported, not tuned per layer).
#include <math.h>

#include <vector>

#include "conv_kernels.inc"

namespace {

constexpr int kMaxDev = 16;
int cur_dev() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDev) d = 0;
    return d;
}

constexpr int igemm_lds(int BP, int BC, int WP, int WC, int KBS, int NST) {
    const int nw = (BP / WP) * (BC / WC);
    const int loop = NST * KBS * (BP + BC) * 64 + (NST > 2 ? nw * 1024 : 0), epi = nw * 32 * (WC + 4) * 4;
    return loop > epi ? loop : epi;
}

struct SplitCfg {
    const char* name;
    int BP, BC, lds, patch, TW, TH, KBS;
    void (*launch)(const ConvArgs&, int sp, int ntc, int ntx, int nty, int total, int chunk, int lds, hipStream_t);
};

template <int BP, int BC, int WP, int WC, int KBS, int NST, int SP>
void launch_igemm_sp(const ConvArgs& a, int ntc, int total, int chunk, int lds, hipStream_t st) {
    static std::atomic<int> attr_done[kMaxDev];
    if (lds > 64 * 1024) {
        const int dev = cur_dev();
        if (!attr_done[dev].load(std::memory_order_acquire)) {
            (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<BP, BC, WP, WC, KBS, 0, NST, SP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<BP, BC, WP, WC, KBS, 1, NST, SP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            attr_done[dev].store(1, std::memory_order_release);
        }
    }
    const dim3 grid(chunk * 8), block((BP / WP) * (BC / WC) * 64);
    if (a.fast_epi)
        hipLaunchKernelGGL((conv_igemm_kernel<BP, BC, WP, WC, KBS, 1, NST, SP>), grid, block, lds, st, a, ntc, total, chunk);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<BP, BC, WP, WC, KBS, 0, NST, SP>), grid, block, lds, st, a, ntc, total, chunk);
}
template <int BP, int BC, int WP, int WC, int KBS, int NST>
void launch_igemm(const ConvArgs& a, int sp, int ntc, int, int, int total, int chunk, int lds, hipStream_t st) {
    if (sp == VGH_FMT_F16X2)
        launch_igemm_sp<BP, BC, WP, WC, KBS, NST, VGH_FMT_F16X2>(a, ntc, total, chunk, lds, st);
    else
        launch_igemm_sp<BP, BC, WP, WC, KBS, NST, VGH_FMT_BF16X2>(a, ntc, total, chunk, lds, st);
}

template <int TW, int TH, int BC, int NWP, int NWC, int SP>
void launch_patch_sp(const ConvArgs& a, int ntc, int ntx, int nty, int total, int chunk, int lds, hipStream_t st) {
    static std::atomic<int> per_cu[kMaxDev];
    const int dev = cur_dev();
    int n = per_cu[dev].load(std::memory_order_acquire);
    if (n == 0) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)conv3x3_patch_kernel<TW, TH, BC, NWP, NWC, SP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)conv3x3_patch_kernel<TW, TH, BC, NWP, NWC, SP>, NWP * NWC * 64, lds) != hipSuccess || n < 1) n = 1;
        per_cu[dev].store(n, std::memory_order_release);
    }
    const int gpx = vgh_conv_persistent_blocks_per_xcd(a, chunk, n);  // as the bf16 patch tiles: lane share + vgh_conv_set_max_blocks_per_xcd (multi-tile tests)
    hipLaunchKernelGGL((conv3x3_patch_kernel<TW, TH, BC, NWP, NWC, SP>), dim3(gpx * 8), dim3(NWP * NWC * 64), lds, st, a, ntc, ntx, nty, total, chunk);
}
template <int TW, int TH, int BC, int NWP, int NWC>
void launch_patch(const ConvArgs& a, int sp, int ntc, int ntx, int nty, int total, int chunk, int lds, hipStream_t st) {
    if (sp == VGH_FMT_F16X2)
        launch_patch_sp<TW, TH, BC, NWP, NWC, VGH_FMT_F16X2>(a, ntc, ntx, nty, total, chunk, lds, st);
    else
        launch_patch_sp<TW, TH, BC, NWP, NWC, VGH_FMT_BF16X2>(a, ntc, ntx, nty, total, chunk, lds, st);
}

#define SCFG(BP, BC, WP, WC) \
    { "s" #BP "x" #BC "_w" #WP "x" #WC, BP, BC, igemm_lds(BP, BC, WP, WC, 1, 2), 0, 0, 0, 1, launch_igemm<BP, BC, WP, WC, 1, 2> }
#define SCFGR(BP, BC, WP, WC, KBS, NST) \
    { "s" #BP "x" #BC "_w" #WP "x" #WC "_k" #KBS "_r" #NST, BP, BC, igemm_lds(BP, BC, WP, WC, KBS, NST), 0, 0, 0, KBS, launch_igemm<BP, BC, WP, WC, KBS, NST> }
#define SPCFG(TW, TH, BC, NWP, NWC) \
    { "sp" #TH "x" #TW "x" #BC "_n" #NWP "x" #NWC, (TW) * (TH), BC, patch_lds<TW, TH, BC, NWP, NWC>(), 1, TW, TH, 1, launch_patch<TW, TH, BC, NWP, NWC> }

// indices 0..15 are what pick_split_cfg returns; the rest are tuner candidates (tools/tune_conv.py --precision fp16x3)
const SplitCfg g_scfgs[] = {
    SCFG(128, 128, 64, 64),    // 0
    SCFG(128, 96, 32, 96),     // 1
    SCFG(128, 64, 32, 64),     // 2
    SCFG(128, 32, 32, 32),     // 3
    SCFG(64, 128, 32, 64),     // 4
    SCFG(64, 96, 32, 96),      // 5
    SCFG(64, 64, 32, 32),      // 6
    SCFG(64, 32, 32, 32),      // 7
    SPCFG(16, 16, 128, 4, 2),  // 8
    SPCFG(16, 16, 96, 4, 1),   // 9
    SPCFG(16, 16, 64, 4, 1),   // 10
    SPCFG(16, 16, 32, 4, 1),   // 11
    SPCFG(40, 8, 128, 5, 2),   // 12
    SPCFG(40, 8, 96, 5, 1),    // 13
    SPCFG(40, 8, 64, 5, 1),    // 14
    SCFG(256, 128, 64, 64),    // 15  8 waves
    SCFGR(256, 128, 64, 64, 1, 3),  // 16  the bf16 table's favourite on 40 / 20-wide maps
    SCFGR(128, 128, 64, 64, 1, 3),  // 17
    SCFGR(128, 128, 64, 64, 2, 2),  // 18
    SCFGR(128, 64, 32, 64, 2, 2),   // 19
    SCFGR(256, 64, 64, 64, 1, 2),   // 20
    SCFGR(256, 256, 64, 64, 1, 3),  // 21  16 waves
    SCFGR(128, 128, 32, 64, 1, 3),  // 22  8 waves
    SPCFG(32, 8, 128, 4, 2),   // 23
    SPCFG(32, 8, 64, 4, 1),    // 24
    SPCFG(32, 8, 96, 4, 1),    // 25
    SPCFG(20, 8, 128, 5, 2),   // 26  20-wide maps
    SPCFG(32, 16, 128, 4, 2),  // 27  512 px x 128
    SPCFG(16, 16, 256, 4, 4),  // 28
    SPCFG(16, 8, 64, 4, 1),    // 29  128 px x 64: three blocks per CU
    SPCFG(16, 8, 128, 4, 2),   // 30
};
constexpr int kNumSplitCfgs = sizeof(g_scfgs) / sizeof(g_scfgs[0]);

// single-plane fp16 (VGH_FMT_F16, r05): besides the tiles above, the 8-wave ping-pong 3x3 tiles of conv_pp.hip in their fp16 variant, addressed as the three
// pseudo-indices kNumSplitCfgs + {0, 1, 2} = 128 / 96 / 64 couts per workgroup
constexpr int kPPBC[3] = {128, 96, 64};
const char* const kPPNames[3] = {"g8x8x128_n8", "g8x8x96_n8", "g8x8x64_n8"};
bool pp16_ok(int cfg, const ConvArgs& a) {
    const int i = cfg - kNumSplitCfgs;
    return i >= 0 && i < 3 && a.nseg == 1 && a.split == VGH_FMT_F16X2 && a.ksize == 3 && a.stride == 1 && a.fast_epi && !a.out_f32 && !a.shuffle && !a.grp_cout && a.act != VGH_ACT_SILU &&
           a.cout_pad % kPPBC[i] == 0 && a.cout_store == a.cout_pad && vgh_conv_pp_fits(a);
}

bool scfg_ok(int cfg, const ConvArgs& a) {
    if (cfg >= kNumSplitCfgs) return pp16_ok(cfg, a);
    if (cfg < 0 || cfg >= kNumSplitCfgs) return false;
    const SplitCfg& e = g_scfgs[cfg];
    if (a.cout_pad % e.BC) return false;
    if (a.grp_cout && a.grp_cout % e.BC) return false;
    if (e.patch && !(a.ksize == 3 && a.stride == 1 && a.fast_epi && !a.out_f32 && !a.shuffle)) return false;
    // the accumulators are rescaled between two steps of the K loop: the boundary (2 x the k-blocks of a segment) must fall on a step.  The
    // ABI-level query has no channel count (cblocks = 0): there only KBS <= 2 is promised, which every segment length satisfies
    if (e.KBS > 2 && (a.cblocks == 0 || (2 * a.ksize * a.ksize * a.cblocks) % e.KBS)) return false;
    return true;
}

int pick_split_cfg(const ConvArgs& a) {
    const int n = a.grp_cout ? a.grp_cout : a.cout_pad;
    const int bc = n % 128 == 0 ? 128 : n % 96 == 0 ? 96 : n % 64 == 0 ? 64 : 32;
    // single-plane fp16: the ping-pong tiles where the bf16 table runs them -- 3x3 / stride-1 layers of the maps at least 40 pixels a side (8 x 8 sub-patches tile
    // a 20-wide map badly)
    if (a.nseg == 1 && bc >= 64 && a.W >= 40 && a.H >= 40) {
        const int c = kNumSplitCfgs + (bc == 128 ? 0 : bc == 96 ? 1 : 2);
        if (pp16_ok(c, a)) return c;
    }
    const bool patch_ok = a.ksize == 3 && a.stride == 1 && a.fast_epi && !a.out_f32 && !a.shuffle;
    if (patch_ok && a.W % 16 == 0 && a.H % 16 == 0) return bc == 128 ? 8 : bc == 96 ? 9 : bc == 64 ? 10 : 11;
    if (patch_ok && a.W % 40 == 0 && a.H % 8 == 0 && bc >= 64) return bc == 128 ? 12 : bc == 96 ? 13 : 14;
    // implicit GEMM: 128-pixel tiles while they still give every CU a few blocks, else 64-pixel tiles
    const int64_t tiles128 = ((int64_t)a.P + 127) / 128 * (a.cout_pad / bc);
    if (tiles128 >= 1024) return bc == 128 ? (tiles128 >= 4096 ? 15 : 0) : bc == 96 ? 1 : bc == 64 ? 2 : 3;
    return bc == 128 ? 4 : bc == 96 ? 5 : bc == 64 ? 6 : 7;
}

// ---- host-side 16-bit conversions (round to nearest even) ----
uint16_t f32_to_f16_host(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x47800000u) return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));  // >= 65536: inf (NaN stays NaN)
    if (x < 0x38800000u) {  // below 2^-14: subnormal half = rn(|f| * 2^24)
        float m;
        memcpy(&m, &x, 4);
        return (uint16_t)(sign | (uint32_t)lrintf(m * 16777216.0f));
    }
    const uint32_t mant = x & 0x7fffffu;
    uint32_t h = (((x >> 23) - 112u) << 10) | (mant >> 13);
    const uint32_t rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;  // a carry into the exponent is the correct result (up to inf)
    return (uint16_t)(sign | h);
}
float f16_to_f32_host(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 1023u;
    float out;
    if (e == 0) {
        out = (float)m * 5.9604644775390625e-08f;  // m * 2^-24
        uint32_t b;
        memcpy(&b, &out, 4);
        b |= sign;
        memcpy(&out, &b, 4);
        return out;
    }
    const uint32_t b = sign | (e == 31 ? 0x7f800000u | (m << 13) : ((e + 112u) << 23) | (m << 13));
    memcpy(&out, &b, 4);
    return out;
}
float bf16_to_f32_host(uint16_t h) {
    const uint32_t b = (uint32_t)h << 16;
    float out;
    memcpy(&out, &b, 4);
    return out;
}

// [cout_pad][taps][cin] 16-bit values -> the kernel image [kb][cout][slot ^ ((cout>>2)&3)][8] of vgh_pack_conv_weights_host
void pack_image16(const uint16_t* w16, int cout_pad, int ksize, int cin, uint16_t* dst) {
    const int cblocks = cin / 32, taps = ksize * ksize;
    for (int tap = 0; tap < taps; ++tap)
        for (int cb = 0; cb < cblocks; ++cb) {
            const int kb = tap * cblocks + cb;
            for (int co = 0; co < cout_pad; ++co) {
                const uint16_t* src = w16 + ((size_t)co * taps + tap) * cin + cb * 32;
                uint16_t* d = dst + ((size_t)kb * cout_pad + co) * 32;
                const int sw = (co >> 2) & 3;
                for (int chunk = 0; chunk < 4; ++chunk) memcpy(d + (chunk ^ sw) * 8, src + chunk * 8, 16);
            }
        }
}

}  // namespace

int vgh_conv_split_pick(const ConvArgs& a) { return pick_split_cfg(a); }

extern "C" int vgh_conv_split_num_cfgs(void) { return kNumSplitCfgs + 3; }
extern "C" const char* vgh_conv_split_cfg_name(int cfg) { return (cfg >= 0 && cfg < kNumSplitCfgs) ? g_scfgs[cfg].name : (cfg >= kNumSplitCfgs && cfg < kNumSplitCfgs + 3) ? kPPNames[cfg - kNumSplitCfgs] : "?"; }
extern "C" int vgh_conv_split_cfg_ok(int cfg, int ksize, int stride, int cout_pad, int fast_epilogue, int shuffle, int grp_cout) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.ksize = ksize;
    a.stride = stride;
    a.cout_pad = cout_pad;
    a.fast_epi = fast_epilogue;
    a.shuffle = shuffle;
    a.grp_cout = grp_cout;
    if (cfg >= kNumSplitCfgs) {  // the fp16 ping-pong tiles: single-plane fp16 nets only (the ABI-level query has no tensor sizes: P = 0 passes the 2 GiB rule)
        a.nseg = 1;
        a.split = VGH_FMT_F16X2;
        a.cout_store = cout_pad;
        a.W = 1;
        a.in_pitch = 8;
    }
    return scfg_ok(cfg, a) ? 1 : 0;
}

void vgh_pack_conv_weights_split_host(const float* w, int cout_pad, int ksize, int cin, int fmt, uint16_t* dst, float* out_scale) {
    const size_t n = (size_t)cout_pad * ksize * ksize * cin;
    std::vector<uint16_t> hi(n), lo(n);
    float scale = 1.0f;
    if (fmt == VGH_FMT_F16X2 || fmt == VGH_FMT_F16) {
        float mx = 0.0f;
        for (size_t i = 0; i < n; ++i) mx = fmaxf(mx, fabsf(w[i]));
        if (mx > 0.0f && isfinite(mx)) {
            int e;
            (void)frexpf(mx, &e);          // mx = f * 2^e, f in [0.5, 1)
            int s = 10 - e;                // max |w| * 2^s in [512, 1024)
            s = s < -40 ? -40 : s > 40 ? 40 : s;
            scale = ldexpf(1.0f, s);
        }
        for (size_t i = 0; i < n; ++i) {
            const float v = w[i] * scale;  // exact (power of two)
            const uint16_t h = fabsf(v) < 6.103515625e-05f ? (uint16_t)0 : f32_to_f16_host(v);
            hi[i] = h;
            lo[i] = f32_to_f16_host((v - f16_to_f32_host(h)) * 2048.0f);
        }
    } else {
        for (size_t i = 0; i < n; ++i) {
            const uint16_t h = vgh_f32_to_bf16_host(w[i]);
            hi[i] = h;
            lo[i] = vgh_f32_to_bf16_host(w[i] - bf16_to_f32_host(h));
        }
    }
    if (fmt == VGH_FMT_F16) {  // single-plane fp16: one segment, the rounded (and prescaled) weights themselves
        for (size_t i = 0; i < n; ++i) hi[i] = f32_to_f16_host(w[i] * scale);  // (no flush of the sub-normal range here: there is no lo plane to carry it)
        pack_image16(hi.data(), cout_pad, ksize, cin, dst);
        if (out_scale) *out_scale = 1.0f / scale;
        return;
    }
    // K segments in the order the kernels walk them: x_hi * w_lo, x_lo * w_hi, x_hi * w_hi
    pack_image16(lo.data(), cout_pad, ksize, cin, dst);
    pack_image16(hi.data(), cout_pad, ksize, cin, dst + n);
    pack_image16(hi.data(), cout_pad, ksize, cin, dst + 2 * n);
    if (out_scale) *out_scale = 1.0f / scale;
}

int vgh_launch_conv_split(const ConvArgs& a0, int force_cfg, hipStream_t stream) {
    VGH_REQUIRE(a0.split == VGH_FMT_BF16X2 || a0.split == VGH_FMT_F16X2, "conv_split: format %d is not a split format", a0.split);
    ConvArgs a = a0;
    const bool single = a.nseg == 1;  // VGH_FMT_F16: the caller has set split = F16X2 and the plane strides to 0
    VGH_REQUIRE(!single || (a.split == VGH_FMT_F16X2 && a.in_plane == 0 && a.out_plane == 0 && a.res_plane == 0), "conv_split: a single-plane launch rides the fp16 kernels with plane strides 0");
    if (!single) a.nseg = 3;
    a.seg_kb = a.ksize * a.ksize * a.cblocks;
    a.nkb = a.nseg * a.seg_kb;
    a.acc_scale = single ? 0.0f : a.split == VGH_FMT_F16X2 ? 1.0f / 2048.0f : 1.0f;  // single plane: the residual's "lo" term (a second read of its hi plane) is multiplied away
    a.lo_scale = a.split == VGH_FMT_F16X2 ? 2048.0f : 1.0f;
    if (a.split != VGH_FMT_F16X2) a.out_scale = 1.0f;
    VGH_REQUIRE(a.out_scale > 0.0f, "conv_split: out_scale must come from vgh_pack_conv_weights_split");
    VGH_REQUIRE(a.in_plane % 8 == 0 && (a.out_f32 || a.out_plane % 4 == 0) && (!a.res || a.res_plane % 4 == 0), "conv_split: plane strides must keep the vector alignment");
    VGH_REQUIRE((int64_t)a.B * a.H * a.W * a.in_pitch * 2 < (1ll << 31), "conv_split: input tensor (both planes) must stay below 2 GiB; run the batch in chunks");
    int cfg = (force_cfg >= 0 && scfg_ok(force_cfg, a)) ? force_cfg : (a.fallback_cfg1 > 0 && scfg_ok(a.fallback_cfg1 - 1, a)) ? a.fallback_cfg1 - 1 : pick_split_cfg(a);
    VGH_REQUIRE(scfg_ok(cfg, a), "conv_split: no tile for cout_pad=%d k=%d s=%d grp=%d", a.cout_pad, a.ksize, a.stride, a.grp_cout);
    if (cfg >= kNumSplitCfgs) return vgh_launch_conv_pp(a, kPPBC[cfg - kNumSplitCfgs], 1, vgh_conv_max_blocks_per_xcd(), stream);  // fp16 ping-pong tiles (conv_pp.hip)
    const SplitCfg& e = g_scfgs[cfg];
    const int ntc = a.cout_pad / e.BC;
    if (e.patch) {
        const int ntx = (a.W + e.TW - 1) / e.TW, nty = (a.H + e.TH - 1) / e.TH;
        const int64_t total = (int64_t)a.B * nty * ntx * ntc;
        VGH_REQUIRE(total < (1ll << 30), "conv_split: too many tiles");
        e.launch(a, a.split, ntc, ntx, nty, (int)total, (int)((total + 7) / 8), e.lds, stream);
    } else {
        const int64_t total = (((int64_t)a.P + e.BP - 1) / e.BP) * ntc;
        VGH_REQUIRE(total < (1ll << 30), "conv_split: too many tiles");
        e.launch(a, a.split, ntc, 0, 0, (int)total, (int)((total + 7) / 8), e.lds, stream);
    }
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}

extern "C" int vgh_pack_conv_weights_split(const float* w_host, int cout_pad, int ksize, int cin, int fmt, uint16_t* wpack_host, float* out_scale) {
    VGH_REQUIRE(w_host && wpack_host && out_scale, "pack_split: null argument");
    VGH_REQUIRE(cin % 32 == 0 && cout_pad % 32 == 0 && (ksize == 1 || ksize == 3), "pack_split: cin/cout_pad must be multiples of 32, ksize 1 or 3");
    VGH_REQUIRE(fmt == VGH_FMT_BF16X2 || fmt == VGH_FMT_F16X2 || fmt == VGH_FMT_F16, "pack_split: fmt must be VGH_FMT_BF16X2, VGH_FMT_F16X2 or VGH_FMT_F16 (one segment: cout_pad*k*k*cin u16)");
    vgh_pack_conv_weights_split_host(w_host, cout_pad, ksize, cin, fmt, wpack_host, out_scale);
    return VGH_OK;
}
