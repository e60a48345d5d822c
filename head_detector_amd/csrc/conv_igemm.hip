// Implicit-GEMM convolution for gfx950 (MI355X): NHWC bf16 activations, MFMA 32x32x16 bf16 with
// fp32 accumulation, fused epilogue {bias, ReLU/SiLU, alpha*residual, concat-by-offset store,
// two-segment store, fp32 store, ConvTranspose pixel-shuffle store}.
//
// Replaces (reference: every conv inside the TorchScript blob called at head_detector/detector.py:58-59;
// definition yolo_head_training/configs/arch_params/yolo_heads_{m,l}_arch_params.yaml:4-137 +
// yolo_head_training/yolo_head/yolo_head_dfl_head.py:74-135): the eval-mode QARepVGG / Conv+BN+ReLU /
// ConvBNReLU / ConvTranspose2d blocks, after folding to one conv + bias.
//
// GEMM view:  D[cout][pixel] = sum_k Wt[cout][k] * X[pixel][k],  k = (ky, kx, cin) in 32-channel k-blocks.
//   A operand (MFMA rows i)  = weights  -> every lane ends up holding 4 *consecutive couts* of one pixel
//   B operand (MFMA cols j)  = pixels      per accumulator quad => 8-byte bf16x4 stores, NHWC-contiguous.
// Staging: both tiles go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction).
//   LDS tiles are [rows][32 bf16] = 64 B rows; the four 16-B chunks of a row are XOR-swizzled with
//   ((row>>2)&3) so that ds_read_b128 fragment reads (lane = row) are bank-conflict free.  Because an
//   LDS-DMA destination is lane-linear, the swizzle is applied on the SOURCE side: per-lane gather
//   address for activations (which also implements im2col + zero padding), host pre-swizzled image for
//   weights (vgh_pack_conv_weights_host).
#include "conv_kernels.inc"
#include "conv_cfg_list.h"

namespace {

struct CfgEntry {
    const char* name;
    int BP, BC, threads, lds;
    void (*launch)(const ConvArgs&, int, int, int, int, hipStream_t);
    int patch, TW, TH;  // 7: conv3x3_pp_kernel with two 4 x 8 sub-patches per wave; 5: conv3x3_pp_kernel (conv_pp.hip: 8-wave ping-pong, 8 x 8 sub-patch per wave); 1 / 2: conv3x3_patch_kernel / conv3x3_patch3_kernel (3x3, stride 1, tile TH x TW); 3: conv1x1_stream_kernel; 4: conv3x3_patch_kernel stride 2 (output tile TH x TW); fast epilogue only
    void (*launch_patch)(const ConvArgs&, int, int, int, int, int, int, hipStream_t);
};

static std::atomic<int> g_max_blocks_per_xcd{0};
static std::atomic<int> g_nt_store{0};

static int persistent_blocks_per_xcd(const ConvArgs& a, int chunk, int per_cu) { return vgh_conv_persistent_blocks_per_xcd(a, chunk, per_cu); }

}  // namespace
// persistent workgroups per XCD of a tile-loop kernel (also used by conv_split.hip): resident slots, shared between lanes, capped by the test knob
int vgh_conv_persistent_blocks_per_xcd(const ConvArgs& a, int chunk, int per_cu) {
    // as many blocks per XCD as its 32 CUs keep resident, each looping over tiles; with the batch split over lane streams every
    // lane's kernel takes its share of the slots so that kernels of different lanes are co-resident (and out of phase)
    int slots = 32 * per_cu / (a.grid_share > 1 ? a.grid_share : 1);
    if (slots < 32) slots = 32;
    const int cap = g_max_blocks_per_xcd.load(std::memory_order_relaxed);
    if (cap > 0 && slots > cap) slots = cap;
#ifdef VGH_EXPERIMENTS
    static const int persist_env = getenv("VGH_PATCH_PERSIST") ? atoi(getenv("VGH_PATCH_PERSIST")) : 1;  // 0: one tile per block
    if (!persist_env) return chunk;
#endif
    return chunk < slots ? chunk : slots;
}
namespace {


#ifdef VGH_EXPERIMENTS
template <int TW, int TH, int BC, int NWP, int NWC>
void launch_patch_s2_cfg(const ConvArgs& a, int ntc, int ntx, int nty, int total, int chunk, int lds, hipStream_t st) {
    static std::atomic<int> per_cu[kMaxDevices];
    const int n = patch_blocks_per_cu(conv3x3_patch_kernel<TW, TH, BC, NWP, NWC, 0, 1>, NWP * NWC * 64, lds, per_cu);
    const int gpx = persistent_blocks_per_xcd(a, chunk, n);
    hipLaunchKernelGGL((conv3x3_patch_kernel<TW, TH, BC, NWP, NWC, 0, 1>), dim3(gpx * 8), dim3(NWP * NWC * 64), lds, st, a, ntc, ntx, nty, total, chunk);
}
#endif

#include "conv_launch_igemm.inc"

constexpr int lds_bytes(int BP, int BC, int WP, int WC, int KBS, int NST) {
    const int nw = (BP / WP) * (BC / WC);
    const int loop = NST * KBS * (BP + BC) * 64 + (NST > 2 ? nw * 1024 : 0), epi = nw * 32 * (WC + 4) * 4;
    return loop > epi ? loop : epi;
}
#define CFG(BP, BC, WP, WC, KBS) \
    { #BP "x" #BC "_w" #WP "x" #WC "_k" #KBS, BP, BC, (BP / WP) * (BC / WC) * 64, lds_bytes(BP, BC, WP, WC, KBS, 2), launch_cfg<BP, BC, WP, WC, KBS, 2>, 0, 0, 0, nullptr }
#define CFGR(BP, BC, WP, WC, KBS, NST) \
    { #BP "x" #BC "_w" #WP "x" #WC "_k" #KBS "_r" #NST, BP, BC, (BP / WP) * (BC / WC) * 64, lds_bytes(BP, BC, WP, WC, KBS, NST), vghcfg::lr_##BP##_##BC##_##WP##_##WC##_##KBS##_##NST, 0, 0, 0, nullptr }
#define LCFG(BP, BC, WP, WC, KBS, NST, LF) \
    { #BP "x" #BC "_w" #WP "x" #WC "_k" #KBS "_r" #NST "_l" #LF, BP, BC, (BP / WP) * (BC / WC) * 64 * LF, lds_bytes(BP, BC, WP, WC, KBS, NST), vghcfg::ll_##BP##_##BC##_##WP##_##WC##_##KBS##_##NST##_##LF, 0, 0, 0, nullptr }
#define PCFG(TW, TH, BC, NWP, NWC) \
    { "p" #TH "x" #TW "x" #BC "_n" #NWP "x" #NWC, (TW) * (TH), BC, (NWP) * (NWC) * 64, patch_lds<TW, TH, BC, NWP, NWC>(), nullptr, 1, TW, TH, vghcfg::lp_##TW##_##TH##_##BC##_##NWP##_##NWC }

#define QCFG(TW, TH, BC, NWP, NWC) \
    { "q" #TH "x" #TW "x" #BC "_n" #NWP "x" #NWC, (TW) * (TH), BC, (NWP) * (NWC) * 64, Patch3<TW, TH, BC, NWP, NWC>::LDS, nullptr, 2, TW, TH, vghcfg::lq_##TW##_##TH##_##BC##_##NWP##_##NWC }

#ifdef VGH_EXPERIMENTS
#define DCFG(TW, TH, BC, NWP, NWC) \
    { "d" #TH "x" #TW "x" #BC "_n" #NWP "x" #NWC, (TW) * (TH), BC, (NWP) * (NWC) * 64, patch_lds<TW, TH, BC, NWP, NWC, 1>(), nullptr, 4, TW, TH, launch_patch_s2_cfg<TW, TH, BC, NWP, NWC> }
#endif
#define TCFG(BP, BC, WP, WC, KBS, NST) \
    { "t" #BP "x" #BC "_w" #WP "x" #WC "_k" #KBS "_r" #NST, BP, BC, (BP / WP) * (BC / WC) * 64, Stream1<BP, BC, WP, WC, KBS, NST>::LDS, vghcfg::lt_##BP##_##BC##_##WP##_##WC##_##KBS##_##NST, 3, 0, 0, nullptr }

#define GCFG(BC) \
    { "g8x8x" #BC "_n8", 512, BC, 512, 0, nullptr, 5, 8, 8, nullptr }
#define HCFG(BC) \
    { "h8x8x" #BC "_n8", 512, BC, 512, 0, nullptr, 6, 8, 8, nullptr }
#define SCFG(BC) \
    { "s4x8x" #BC "_n8", 512, BC, 512, 0, nullptr, 7, 8, 4, nullptr }

const CfgEntry g_cfgs[] = {
    CFG(128, 128, 64, 64, 1),  // 0
    CFG(256, 64, 64, 64, 1),   // 1
    CFG(256, 96, 64, 96, 1),   // 2
    CFG(128, 64, 32, 64, 1),   // 3
    CFG(256, 32, 64, 32, 1),   // 4
    CFG(64, 128, 32, 64, 1),   // 5
    CFG(64, 64, 32, 32, 1),    // 6
    CFG(128, 128, 64, 64, 2),  // 7
    CFG(256, 64, 64, 64, 2),   // 8
    CFG(256, 96, 64, 96, 2),   // 9
    CFG(128, 96, 32, 96, 1),   // 10
    CFG(128, 32, 32, 32, 1),   // 11
    CFG(64, 32, 32, 32, 1),    // 12 (2 waves)
    CFG(256, 128, 64, 128, 1), // 13
    CFG(128, 64, 32, 64, 2),   // 14
    PCFG(16, 16, 64, 4, 1),    // 15  256 px x 64: 4 waves x (64 px x 64)
    PCFG(16, 16, 128, 4, 1),   // 16  256 px x 128, 1 wave/SIMD
    PCFG(16, 16, 128, 4, 2),   // 17  256 px x 128: 8 waves x (64 px x 64)
    PCFG(16, 16, 96, 4, 1),    // 18
    PCFG(40, 8, 64, 5, 1),     // 19  320 px (full-width rows of a 40-wide map)
    PCFG(40, 8, 128, 5, 2),    // 20  10 waves x (64 px x 64)
    PCFG(40, 8, 96, 5, 1),     // 21
    PCFG(32, 8, 128, 4, 2),    // 22  8 rows x 32 cols: conflict-free fragment reads, 8 waves x (64 px x 64)
    PCFG(32, 8, 64, 4, 1),     // 23  4 waves x (64 px x 64)
    PCFG(16, 16, 32, 4, 1),    // 24
    CFGR(128, 128, 64, 64, 1, 3),  // 25  counted-vmcnt rings
    CFGR(128, 128, 64, 64, 1, 4),  // 26
    CFGR(128, 64, 32, 64, 1, 4),   // 27
    CFGR(256, 64, 64, 64, 1, 3),   // 28
    CFGR(128, 96, 32, 96, 1, 4),   // 29
    CFGR(64, 128, 32, 64, 1, 4),   // 30
    CFGR(64, 64, 32, 32, 1, 4),    // 31
    CFGR(128, 128, 64, 64, 2, 3),  // 32
    CFGR(128, 64, 32, 64, 2, 3),   // 33
    CFGR(128, 32, 32, 32, 1, 4),   // 34
    CFGR(256, 128, 64, 64, 1, 3),  // 35  8 waves
    CFGR(256, 128, 64, 64, 1, 2),  // 36  8 waves, 2-stage
    CFGR(256, 128, 64, 64, 2, 2),  // 37
    CFGR(256, 256, 64, 64, 1, 2),  // 38  16 waves
    CFGR(256, 256, 64, 64, 1, 3),  // 39
    CFGR(512, 128, 64, 64, 1, 2),  // 40  16 waves
    CFGR(256, 64, 32, 64, 1, 2),   // 41  8 waves, 32x64 wave tiles
    CFGR(256, 64, 32, 64, 1, 3),   // 42
    CFGR(256, 96, 32, 96, 1, 2),   // 43  8 waves
    CFGR(128, 128, 32, 64, 1, 2),  // 44  8 waves
    CFGR(128, 128, 32, 64, 1, 3),  // 45
    CFGR(512, 64, 64, 64, 1, 2),   // 46  8 waves
    CFGR(256, 128, 128, 64, 1, 2), // 47  4 waves, 128x64 wave tiles (8 MFMA tiles / wave)
    CFGR(256, 128, 128, 64, 1, 3), // 48
    CFGR(256, 128, 64, 128, 1, 2), // 49  4 waves, 64x128 wave tiles
    CFGR(256, 64, 128, 64, 1, 2),  // 50  2 waves
    CFGR(512, 128, 128, 64, 1, 2), // 51  8 waves, 128x64 wave tiles
    CFGR(256, 256, 128, 64, 1, 2), // 52  8 waves
    CFGR(256, 256, 64, 128, 1, 2), // 53  8 waves
    PCFG(32, 8, 256, 4, 4),    // 54  256 px x 256: 16 waves x (64 px x 64)
    PCFG(32, 16, 128, 8, 2),   // 55  512 px x 128: 16 waves
    PCFG(32, 8, 96, 4, 1),     // 56
    PCFG(32, 8, 64, 4, 2),     // 57  8 waves x (64 px x 32)
    PCFG(16, 16, 256, 4, 4),   // 58
    PCFG(20, 8, 128, 5, 2),    // 59  160 px (20-wide maps), 10 waves
    PCFG(32, 4, 128, 4, 2),    // 60  128 px x 128, 8 waves x (32 px x 64)
    CFG(128, 128, 64, 64, 3),  // 61  deep single-shot stages for short K (1x1 convs): K = 96 / 128 in ONE load round trip
    CFG(128, 128, 64, 64, 4),  // 62
    CFG(128, 64, 32, 64, 3),   // 63
    CFG(128, 64, 32, 64, 4),   // 64
    CFG(128, 96, 32, 96, 3),   // 65
    CFG(128, 96, 32, 96, 4),   // 66
    CFG(64, 64, 32, 32, 3),    // 67
    CFG(64, 64, 32, 32, 4),    // 68
    CFG(256, 64, 64, 64, 3),   // 69
    CFG(64, 128, 32, 64, 3),   // 70
    CFG(64, 128, 32, 64, 4),   // 71
    CFG(128, 32, 32, 32, 4),   // 72
    CFG(64, 96, 32, 96, 3),    // 73
    CFG(256, 128, 64, 64, 3),  // 74  8 waves
    PCFG(32, 16, 128, 4, 2),   // 75  512 px x 128: 8 waves x (128 px x 64): 6 fragment reads per 8 MFMAs
    PCFG(32, 16, 64, 8, 1),    // 76  512 px x 64: 8 waves x (64 px x 64)
    // cross-tile pipelined patch kernels (v3), same tile shapes as their "p" twins
    QCFG(16, 16, 64, 4, 1),    // 77
    QCFG(16, 16, 128, 4, 2),   // 78
    QCFG(16, 16, 96, 4, 1),    // 79
    QCFG(32, 8, 96, 4, 1),     // 80
    QCFG(32, 8, 64, 4, 1),     // 81
    QCFG(32, 8, 128, 4, 2),    // 82
    QCFG(40, 8, 64, 5, 1),     // 83
    QCFG(40, 8, 128, 5, 2),    // 84
    QCFG(16, 16, 256, 4, 4),   // 85
    QCFG(32, 8, 256, 4, 4),    // 86
    QCFG(16, 16, 32, 4, 1),    // 87
    QCFG(20, 8, 128, 5, 2),    // 88
    QCFG(32, 16, 128, 4, 2),   // 89  512 px x 128: 8 waves x (128 px x 64)
    QCFG(16, 16, 128, 4, 1),   // 90
    QCFG(32, 8, 64, 4, 2),     // 91
    QCFG(32, 4, 128, 4, 2),    // 92
    QCFG(40, 8, 96, 5, 1),     // 93
    PCFG(16, 8, 64, 4, 1),     // 94  128 px x 64: 48 KiB of LDS -> three blocks per CU (3 waves per SIMD), 1.5 KiB of fragments per MFMA
    QCFG(16, 8, 64, 4, 1),     // 95
    PCFG(16, 8, 128, 4, 2),    // 96  128 px x 128, 8 waves: 72 KiB -> two blocks = 4 waves per SIMD
    QCFG(16, 8, 128, 4, 2),    // 97
    PCFG(16, 8, 256, 4, 4),    // 98  128 px x 256, 16 waves
    // big register tiles, one wave per SIMD (the hipBLASLt recipe: 0.5 KB of fragment reads per MFMA instead of 1 KB; one block per CU)
    PCFG(32, 16, 128, 4, 1),   // 99   512 px x 128: 4 waves x (128 px x 128): 256 accumulator registers per lane
    PCFG(40, 16, 128, 4, 1),   // 100  640 px x 128: 4 waves x (160 px x 128): tiles 80-wide maps exactly
    // persistent streaming tiles for 1x1 / stride-1 convs (conv1x1_stream_kernel: the ring of stages runs across tiles)
    TCFG(128, 96, 32, 96, 1, 3),   // 101
    TCFG(128, 64, 32, 64, 1, 3),   // 102
    TCFG(128, 64, 32, 64, 1, 4),   // 103
    TCFG(128, 128, 64, 64, 1, 3),  // 104
    TCFG(64, 64, 32, 32, 1, 3),    // 105
    TCFG(64, 96, 32, 96, 1, 3),    // 106  2 waves
    TCFG(128, 192, 64, 96, 1, 3),  // 107  all 192 couts of the merged conv1|conv2 launches in one pass over the pixels
    TCFG(128, 32, 32, 32, 1, 3),   // 108
    TCFG(128, 96, 32, 96, 2, 3),   // 109  128 B per pixel per stage: whole lines per request
    TCFG(128, 64, 32, 64, 2, 3),   // 110
    TCFG(256, 64, 64, 64, 1, 3),   // 111
    TCFG(128, 96, 32, 96, 1, 4),   // 112
#ifdef VGH_EXPERIMENTS
    // halo-patch tiles for 3x3 / stride-2 convs (conv3x3_patch_kernel<..., S2 = 1>: the input patch de-interleaved into four parity planes).
    // Measured (profiles/r03_tune_d_*.json): correct and conflict-free, but 10 - 60 % SLOWER than the implicit-GEMM rings on every stride-2 layer
    // (stage-1 downsample 439 vs 368 us): a 64-pixel tile has 0.3 us of MFMAs per (channel block, kernel row) step against a 1 us L2 round trip for
    // the next step's weights.  Kept as tuner candidates (the four best shapes; eight more were measured and dropped).
    DCFG(16, 4, 96, 1, 3),    // 113  64 output px x 96: 3 waves x (64 px x 32)
    DCFG(16, 4, 64, 2, 2),    // 114  4 waves x (32 px x 32)
    DCFG(16, 8, 128, 4, 2),   // 115  128 output px: 8 waves x (32 px x 64)
    DCFG(32, 4, 128, 4, 2),   // 116
#endif
    // 64-pixel halo-patch tiles for the 20^2 / 40^2 maps at small batch (M b32: 12 800 pixels = 100 tiles of 128 px per cout tile, i.e. one partial
    // round of the chip; twice the tiles of half the length fill it better)
    PCFG(16, 4, 64, 2, 2),    // 117  4 waves x (32 px x 32)
    PCFG(16, 4, 128, 2, 2),   // 118  4 waves x (32 px x 64)
    PCFG(16, 4, 128, 2, 4),   // 119  8 waves x (32 px x 32)
    PCFG(16, 4, 96, 2, 3),    // 120  6 waves x (32 px x 32)
    PCFG(16, 4, 64, 2, 1),    // 121  2 waves x (32 px x 64)
    // 8-wave ping-pong tiles (conv_pp.hip, r04): 8 sub-patches of 8 x 8 pixels x BC couts per workgroup, one workgroup per CU
    GCFG(128),  // 122
    GCFG(96),   // 123
    GCFG(64),   // 124
    // the same tiles with ONE barrier per tap (conv3x3_pp_kernel<TI, 2>: the groups run M -> L / L -> M inside a slot, 4-stage weight ring)
    HCFG(128),  // 125
    HCFG(96),   // 126
    HCFG(64),   // 127
    // the g tiles with TWO 4 x 8 sub-patches per wave (conv3x3_pp_kernel<TI, 1, ..., GEO = 1>, r06): 20-wide maps tile as 8 + 8 + 8 (the last one overlapping four
    // columns) x five 4-row bands -- 1.2 x the pixels where the 8 x 8 patches cover 1.44 x -- and small maps split into twice as many independent sub-patches
    SCFG(128),  // 128
    SCFG(96),   // 129
    SCFG(64),   // 130
    // 192-cout tiles (r06): the stride-2 3x3 layers with 192 couts ran two 96-cout tiles per pixel tile, i.e. loaded every activation k-block twice; what bounds the
    // implicit-GEMM family below the MFMA rate is the bytes a CU pulls through its LDS-DMA path (~12 - 16 B / clk / CU measured against 64 nominal), so flop per byte counts
    CFG(256, 192, 64, 96, 1),      // 131  8 waves x (64 px x 96)
    CFGR(256, 192, 64, 96, 1, 3),  // 132
    CFG(128, 192, 32, 96, 1),      // 133  8 waves x (32 px x 96)
    CFGR(128, 192, 32, 96, 1, 3),  // 134
    // deep rings on the small tiles (r06, single-image latency): at B = 1 a 20^2 / 40^2 layer is 56 - 150 blocks, each walking K = 1728 - 4608 alone; a step of the
    // 2-stage loop costs one L2 / HBM round trip (~0.85 us for 128 k: 31 us per stage-4 layer with 2 us of MFMAs in it) -- three stages of LDS-DMA in flight instead of one
    CFGR(64, 64, 32, 32, 4, 3),    // 135
    CFGR(64, 64, 32, 32, 4, 4),    // 136  128 KiB of stages: one block per CU (there are fewer blocks than CUs)
    CFGR(64, 64, 32, 32, 2, 4),    // 137
    CFGR(64, 32, 32, 32, 4, 4),    // 138  2 waves
    CFGR(128, 32, 32, 32, 4, 3),   // 139
    CFGR(128, 64, 32, 64, 2, 4),   // 140
    // loader waves (r06): the same small tiles with 2 - 6 x the waves, all of them staging tiles (one wave loads ~14 GB/s whatever it keeps in flight)
    LCFG(64, 64, 32, 32, 4, 3, 2),   // 141  8 waves
    LCFG(64, 64, 32, 32, 4, 3, 4),   // 142  16 waves
    LCFG(64, 64, 32, 32, 4, 4, 4),   // 143
    LCFG(64, 64, 32, 32, 2, 4, 4),   // 144
    LCFG(64, 32, 32, 32, 4, 3, 3),   // 145  6 waves
    LCFG(64, 32, 32, 32, 4, 3, 6),   // 146  12 waves
    LCFG(32, 64, 32, 32, 4, 3, 3),   // 147  32-pixel tiles: 13 instead of 7 pixel tiles over a 20^2 map
    LCFG(32, 64, 32, 32, 4, 3, 6),   // 148
    LCFG(128, 64, 32, 64, 2, 3, 3),  // 149  12 waves
    CFGR(128, 96, 32, 96, 1, 3),     // 150  three stages = 46 KiB < the 51 KiB of epilogue strips: still three blocks per CU, two stages in flight each
    CFGR(128, 192, 32, 96, 1, 4),    // 151
    // "r" tile (r06, ds_b2b.hip): 3x3 / stride-2 convs with 96 input channels on the persistent 4-wave structure of the stage-1 pair's tile -- the input patch fetched once
    // into parity planes, the wave's 32-cout slice of the weights (K = 864: 54 fragments) resident in registers
    { "r8x8x96_n4", 64, 96, 256, 0, nullptr, 8, 8, 8, nullptr },  // 152
    // "w" tiles (r06, ds_b2b.hip): 3x3 / stride-1 convs with 96 / 128 input channels, the weights resident in registers (K = 864 / 1 152: 54 / 72 fragments per wave), one
    // wave per cout group and SIMD, no barrier inside a tile
    { "w8x8x96_n3", 64, 96, 192, 0, nullptr, 9, 8, 8, nullptr },   // 153
    { "w8x8x128_n4", 64, 128, 256, 0, nullptr, 9, 8, 8, nullptr },  // 154
    // (measured and dropped: one-block 16-wave shapes p8x32x128_n8x2 / p16x16x128_n8x2 700 / 650 TFLOP/s where two 8-wave blocks reach 840-880;
    //  p8x16x96_n4x1 654 vs 781 for p8x32x96)
};
constexpr int kNumCfgs = sizeof(g_cfgs) / sizeof(g_cfgs[0]);

}  // namespace

#ifdef VGH_EXPERIMENTS
static unsigned long long* g_trace = nullptr;
extern "C" int vgh_conv_set_trace(void* dev_buffer) {
    g_trace = (unsigned long long*)dev_buffer;
    return VGH_OK;
}
#endif
int vgh_conv_num_cfgs() { return kNumCfgs; }
#ifdef VGH_EXPERIMENTS
int vgh_conv_set_nt_store(int on) {
    g_nt_store.store(on ? 1 : 0, std::memory_order_relaxed);
    return VGH_OK;
}
#endif

int vgh_conv_max_blocks_per_xcd() { return g_max_blocks_per_xcd.load(std::memory_order_relaxed); }
int vgh_conv_set_max_blocks_per_xcd(int blocks) {
    VGH_REQUIRE(blocks >= 0, "conv_set_max_blocks_per_xcd: negative");
    g_max_blocks_per_xcd.store(blocks, std::memory_order_relaxed);
    return VGH_OK;
}
int vgh_conv_cfg_ok(int cfg, int ksize, int stride, int cout_pad, int fast_epilogue, int shuffle) {
    if (cfg < 0 || cfg >= kNumCfgs) return 0;
    const CfgEntry& e = g_cfgs[cfg];
    if (cout_pad % e.BC) return 0;
    if ((e.patch == 1 || e.patch == 2 || e.patch == 5 || e.patch == 6 || e.patch == 7) && !(ksize == 3 && stride == 1 && fast_epilogue && !shuffle)) return 0;
    if (e.patch == 7 && cout_pad > 1024) return 0;
    if (e.patch == 3 && !(ksize == 1 && stride == 1 && fast_epilogue && !shuffle)) return 0;
    if (e.patch == 4 && !(ksize == 3 && stride == 2 && fast_epilogue && !shuffle)) return 0;
    if (e.patch == 8 && !(ksize == 3 && stride == 2 && fast_epilogue && !shuffle && cout_pad <= 192)) return 0;
    if (e.patch == 9 && !(ksize == 3 && stride == 1 && fast_epilogue && !shuffle && cout_pad <= 1024)) return 0;
    return 1;
}
const char* vgh_conv_cfg_name(int cfg) { return (cfg >= 0 && cfg < kNumCfgs) ? g_cfgs[cfg].name : "?"; }
int vgh_conv_cfg_cout_tile(int cfg) { return (cfg >= 0 && cfg < kNumCfgs) ? g_cfgs[cfg].BC : 0; }
// as vgh_conv_cfg_ok, for a concrete launch: a grouped conv additionally needs cout tiles that do not straddle groups
static int cfg_ok_for(int cfg, const ConvArgs& a) {
    if (!vgh_conv_cfg_ok(cfg, a.ksize, a.stride, a.cout_pad, a.fast_epi && !a.out_f32, a.shuffle)) return 0;
    if (a.grp_cout && a.grp_cout % g_cfgs[cfg].BC) return 0;
    if ((a.in_fp8 || a.out_fp8) && g_cfgs[cfg].patch != 5) return 0;  // e4m3 links: g tiles only (conv_pp.hip)
    if (a.dvec && g_cfgs[cfg].BC == 128) return 0;
    if ((g_cfgs[cfg].patch == 5 || g_cfgs[cfg].patch == 6 || g_cfgs[cfg].patch == 7) && (a.grp_cout || a.act == VGH_ACT_SILU || a.out_f32 || !vgh_conv_pp_fits(a))) return 0;  // ping-pong tiles: dense bf16 -> bf16, ReLU / none
    if (g_cfgs[cfg].patch == 7 && (a.Wo < 8 || a.Ho < 4)) return 0;  // 4 x 8 sub-patches are moved back inside the map, never cut
    if (g_cfgs[cfg].patch == 3 && (a.res || a.grp_cout || a.act == VGH_ACT_SILU || a.out_f32 || a.pad)) return 0;  // streaming 1x1 tiles: plain bf16 -> bf16 only
    if (g_cfgs[cfg].patch == 8 && !vgh_conv_ds_ok(a)) return 0;  // r tile: 96 input channels, whole 8 x 8 tiles
    if (g_cfgs[cfg].patch == 9 && !(vgh_conv_w_ok(a) && a.cin == g_cfgs[cfg].BC)) return 0;  // w tiles: cin = the tile's 96 / 128, whole 8 x 8 tiles
    return 1;
}

// for net.hip: can tile `cfg` run the PREPARED launch `a` (the only batch-dependent term is the ping-pong tiles' 2 GiB rule, vgh_conv_pp_fits)
int vgh_conv_cfg_ok_for(int cfg, const ConvArgs& a) { return a.split ? 1 : cfg_ok_for(cfg, a); }

namespace {
// Tile choice for shapes without a measured entry in tuning/conv_cfg.json: a class table distilled from the per-op tuning
// reports (tools/gen_heuristic.py).  Class = (kernel size, stride, pixel-count bucket, largest of {128,96,64,32} dividing
// cout, cout > 128, K bucket, output width divisible by 40 / 32 / 16).
struct HeurRow {
    int ks, st, pb, nd, nb, kb, wo;
    const char* name;
};
const HeurRow g_heur[] = {
#include "conv_heur_table.inc"
};
constexpr int kNumHeur = sizeof(g_heur) / sizeof(g_heur[0]);

int cfg_by_name(const char* name) {
    for (int i = 0; i < kNumCfgs; ++i)
        if (strcmp(g_cfgs[i].name, name) == 0) return i;
    return -1;
}

int pick_size_only(const ConvArgs& a) {  // last resort: largest plain implicit-GEMM tile that still fills the chip
    const int cp = a.grp_cout ? a.grp_cout : a.cout_pad;  // grouped: tiles must divide the group width
    const int64_t P = a.P;
    auto tiles = [&](int cfg) { return ((P + g_cfgs[cfg].BP - 1) / g_cfgs[cfg].BP) * (int64_t)(cp / g_cfgs[cfg].BC); };
    int cand[4];
    int n = 0;
    if (cp % 128 == 0) cand[n++] = 0;
    if (cp % 96 == 0) cand[n++] = 2;
    if (cp % 64 == 0) cand[n++] = 1;
    cand[n++] = 4;
    for (int i = 0; i < n; ++i)
        if (tiles(cand[i]) >= 512) return cand[i];
    if (cp % 128 == 0 && tiles(5) >= 256) return 5;
    if (cp % 64 == 0) return 6;
    if (cp % 32 == 0) return (P >= 128 * 512) ? 4 : 11;
    return cand[0];
}
}  // namespace

int vgh_conv_pick_cfg(const ConvArgs& a) {
    if (a.in_fp8 || a.out_fp8) {  // e4m3 links run on the g tiles: the widest cout tile that divides the layer
        static const int g128 = cfg_by_name("g8x8x128_n8"), g96 = cfg_by_name("g8x8x96_n8"), g64 = cfg_by_name("g8x8x64_n8");
        // (the 128-cout variant of the diagonal bypass spills ~100 registers: 96 or 64 couts per workgroup there)
        return (a.cout_pad % 128 == 0 && !a.dvec) ? g128 : a.cout_pad % 96 == 0 ? g96 : g64;
    }
    // A measured per-layer table (tuning/*.json) overrides this through force_cfg.
    static int row_cfg[kNumHeur];
    static bool resolved = false;
    if (!resolved) {
        for (int i = 0; i < kNumHeur; ++i) row_cfg[i] = cfg_by_name(g_heur[i].name);
        resolved = true;
    }
    const int64_t P = a.P;
    const int N = a.cout_pad, K = a.nkb * 32;
    const int pb = P < 8192 ? 0 : P < 40000 ? 1 : P < 160000 ? 2 : P < 600000 ? 3 : 4;
    const int nd = N % 128 == 0 ? 128 : N % 96 == 0 ? 96 : N % 64 == 0 ? 64 : 32;
    const int nb = N > 128, kb = K <= 128 ? 0 : K <= 512 ? 1 : 2;
    const int wo = (a.ksize == 3 && a.stride == 1) ? ((a.Wo % 40 == 0 && a.Wo % 32 != 0) | ((a.Wo % 32 == 0) << 1) | ((a.Wo % 16 == 0) << 2)) : 0;
    int best = -1, best_d = 1 << 30;
    for (int i = 0; i < kNumHeur; ++i) {
        const HeurRow& r = g_heur[i];
        if (r.ks != a.ksize || r.st != a.stride || row_cfg[i] < 0) continue;
        const int d = 4 * abs(r.pb - pb) + 8 * (r.nd != nd) + 2 * (r.nb != nb) + abs(r.kb - kb) + 3 * (r.wo != wo);
        if (d < best_d && cfg_ok_for(row_cfg[i], a)) {
            best_d = d;
            best = row_cfg[i];
        }
    }
    return best >= 0 ? best : pick_size_only(a);
}

int vgh_conv_prepare(ConvArgs& a) {
    VGH_REQUIRE(a.cin % 32 == 0 && a.cin > 0, "conv: cin=%d must be a positive multiple of 32", a.cin);
    VGH_REQUIRE(!(a.in_fp8 || a.out_fp8) || (a.ksize == 3 && a.stride == 1 && !a.split && !a.out_f32), "conv: e4m3 tensors link 3x3 / stride-1 convs only");
    VGH_REQUIRE(a.cout_pad % 32 == 0 && a.cout_pad > 0, "conv: cout_pad=%d must be a multiple of 32", a.cout_pad);
    VGH_REQUIRE(a.ksize == 1 || a.ksize == 3, "conv: ksize=%d unsupported", a.ksize);
    VGH_REQUIRE(a.in_coff % 8 == 0 && a.in_pitch % 8 == 0, "conv: input channel offset / pitch must keep 16-byte alignment");
    VGH_REQUIRE(a.out_f32 || (a.out_split % 4 == 0 && a.out_coff % 4 == 0 && a.out_coff2 % 4 == 0 && a.cout_store % 4 == 0 && a.out_pitch % 4 == 0),
                "conv: bf16 output needs 8-byte aligned channel offsets and cout_store %% 4 == 0");
    VGH_REQUIRE(!a.res || (a.res_coff % 4 == 0 && a.res_pitch % 4 == 0), "conv: residual alignment");
    VGH_REQUIRE((int64_t)a.B * a.H * a.W * a.in_pitch * 2 < (1ll << 31), "conv: input tensor must stay below 2 GiB (32-bit buffer offsets); run the batch in chunks");
    VGH_REQUIRE(!a.shuffle || (a.shuffle_c % 4 == 0 && a.cout_pad >= 4 * a.shuffle_c && a.ksize == 1 && a.stride == 1), "conv: bad shuffle");
    VGH_REQUIRE(a.grp_cout == 0 || (a.grp_cout % 32 == 0 && a.cout_pad % a.grp_cout == 0 && a.grp_in_stride % 8 == 0), "conv: bad group geometry (grp_cout=%d)", a.grp_cout);
    VGH_REQUIRE((int64_t)a.B * a.Ho * a.Wo < (1ll << 30), "conv: too many output pixels for one launch");
#ifdef VGH_EXPERIMENTS
    static const int ablate = getenv("VGH_CONV_ABLATE") ? atoi(getenv("VGH_CONV_ABLATE")) : 0;
    static const int stagger_env = getenv("VGH_STAGGER") ? atoi(getenv("VGH_STAGGER")) : -1;
    static const int share_env = getenv("VGH_GRID_SHARE") ? atoi(getenv("VGH_GRID_SHARE")) : -1;
    a.trace = g_trace;
    if (stagger_env >= 0) a.stagger = stagger_env;
    if (share_env >= 1) a.grid_share = share_env;
#else
    constexpr int ablate = 0;
#endif
    a.ablate = ablate;
    a.nt_out = g_nt_store.load(std::memory_order_relaxed);
    vgh_fastdiv_magic((unsigned)(a.Ho * a.Wo > 0 ? a.Ho * a.Wo : 1), &a.div_howo_m, &a.div_howo_s);
    vgh_fastdiv_magic((unsigned)(a.Wo > 0 ? a.Wo : 1), &a.div_wo_m, &a.div_wo_s);
    const bool al8 = a.out_coff % 8 == 0 && a.out_coff2 % 8 == 0 && a.out_split % 8 == 0 && a.cout_store % 8 == 0 && a.out_pitch % 8 == 0 &&
                     (!a.res || (a.res_coff % 8 == 0 && a.res_pitch % 8 == 0)) && (!a.shuffle || a.shuffle_c % 8 == 0) &&
                     (!a.split || (a.out_plane % 8 == 0 && (!a.res || a.res_plane % 8 == 0)));
    // fp32 outputs (prediction buffers) take the transposed epilogue too when every pixel row starts 16-byte aligned
    const bool al4f = a.out_f32 && a.out_coff % 4 == 0 && a.out_pitch % 4 == 0 && a.out_split >= a.cout_store && !a.res && !a.shuffle;
    a.fast_epi = (((!a.out_f32 && al8) || al4f) && !(ablate & 4)) ? 1 : 0;
    return VGH_OK;
}

int vgh_conv_pick_auto(const ConvArgs& a) { return a.split ? vgh_conv_split_pick(a) : vgh_conv_pick_cfg(a); }

int vgh_launch_conv(const ConvArgs& a0, int force_cfg, hipStream_t stream) {
    ConvArgs a = a0;
    if (int rc = vgh_conv_prepare(a)) return rc;
    if (a.P == 0) return VGH_OK;
    if (a.split) return vgh_launch_conv_split(a, force_cfg, stream);  // parity modes: own tile set (conv_split.hip)
    int cfg = force_cfg >= 0 ? force_cfg : vgh_conv_pick_cfg(a);
    VGH_REQUIRE(cfg < kNumCfgs, "conv: cfg %d out of range", cfg);
    if (!cfg_ok_for(cfg, a)) {
        VGH_REQUIRE(force_cfg < 0 || a.fallback_cfg1 > 0, "conv: cfg %s cannot run this conv (cout_pad=%d k=%d s=%d grp=%d)", g_cfgs[cfg].name, a.cout_pad, a.ksize, a.stride, a.grp_cout);
        cfg = force_cfg >= 0 ? a.fallback_cfg1 - 1 : (a.grp_cout ? pick_size_only(a) : 4);
        VGH_REQUIRE(cfg >= 0 && cfg < kNumCfgs, "conv: fallback cfg %d out of range", cfg);
        VGH_REQUIRE(cfg_ok_for(cfg, a), "conv: no tile for this conv (cout_pad=%d k=%d s=%d grp=%d)", a.cout_pad, a.ksize, a.stride, a.grp_cout);
    }
    if (g_cfgs[cfg].patch == 2 && !(a.cout_store == a.cout_pad && (a.out_split >= a.cout_pad || a.out_split % g_cfgs[cfg].BC == 0))) {
        // the pipelined kernel stores whole cout tiles into ONE output segment: ragged channel counts run on its "p" twin
        char twin[48];
        snprintf(twin, sizeof(twin), "p%s", g_cfgs[cfg].name + 1);
        const int t = cfg_by_name(twin);
        VGH_REQUIRE(t >= 0, "conv: cfg %s has no fallback tile", g_cfgs[cfg].name);
        cfg = t;
    }
    const CfgEntry& e = g_cfgs[cfg];
    if (e.patch == 8) return vgh_launch_conv_ds(a, stream);
    if (e.patch == 9) return vgh_launch_conv_w(a, stream);
    if (e.patch == 5 || e.patch == 6 || e.patch == 7) return vgh_launch_conv_pp(a, e.BC, e.patch == 7 ? 3 : e.patch == 6 ? 2 : 1, g_max_blocks_per_xcd.load(std::memory_order_relaxed), stream);
    if (e.patch == 1 || e.patch == 2 || e.patch == 4) {
        const int ntc = a.cout_pad / e.BC, ntx = (a.Wo + e.TW - 1) / e.TW, nty = (a.Ho + e.TH - 1) / e.TH;  // (Ho, Wo) = (H, W) for the stride-1 tiles
        const int64_t total = (int64_t)a.B * nty * ntx * ntc;
        VGH_REQUIRE(total < (1ll << 30), "conv: too many tiles");
        const int chunk = (int)((total + 7) / 8);
        e.launch_patch(a, ntc, ntx, nty, (int)total, chunk, e.lds, stream);
        VGH_HIP(hipGetLastError());
        return VGH_OK;
    }
    const int ntc = a.cout_pad / e.BC;
    const int64_t ntp = ((int64_t)a.P + e.BP - 1) / e.BP;
    const int64_t total = ntp * ntc;
    VGH_REQUIRE(total < (1ll << 30), "conv: too many tiles");
    const int chunk = (int)((total + 7) / 8);
    e.launch(a, ntc, (int)total, chunk, e.lds, stream);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}

// ---- back-to-back GEMM (conv_kernels.inc, T2 > 0): a conv whose 96 output channels fit ONE cout tile, fused with the 1x1 conv that is its only reader ----
namespace {
template <int T2, int NST>
int launch_b2b_96(const ConvArgs& a, hipStream_t st) {
    constexpr int BP = 128, BC = 96, WP = 32, WC = 96, KBS = 1;
    constexpr int lds0 = lds_bytes(BP, BC, WP, WC, KBS, NST), w2 = 3 * T2 * 2048;
    constexpr int lds = lds0 > w2 ? lds0 : w2;
    static_assert(lds <= 64 * 1024, "b2b: within the default dynamic-LDS limit");
    const int64_t total = ((int64_t)a.P + BP - 1) / BP;
    VGH_REQUIRE(total < (1ll << 30), "conv b2b: too many tiles");
    const int chunk = (int)((total + 7) / 8);
    hipLaunchKernelGGL((conv_igemm_kernel<BP, BC, WP, WC, KBS, 1, NST, 0, T2>), dim3(chunk * 8), dim3((BP / WP) * 64), lds, st, a, 1, (int)total, chunk);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}
template <int T2>
int launch_b2b_96_any(const ConvArgs& a, hipStream_t st) {
#ifdef VGH_EXPERIMENTS
    static const int nst = getenv("VGH_B2B_NST") ? atoi(getenv("VGH_B2B_NST")) : 2;
    if (nst == 4) return launch_b2b_96<T2, 4>(a, st);
    if (nst == 3) return launch_b2b_96<T2, 3>(a, st);
#endif
    return launch_b2b_96<T2, 2>(a, st);
}
}  // namespace

int vgh_conv_b2b_ok(int ksize, int stride, int cout_pad, int cout2_pad) {
    (void)stride;
    return (ksize == 1 || ksize == 3) && cout_pad == 96 && (cout2_pad == 256 || cout2_pad == 192 || cout2_pad == 128);
}

int vgh_launch_conv_b2b(const ConvArgs& a0, hipStream_t stream) {
    ConvArgs a = a0;
    if (int rc = vgh_conv_prepare(a)) return rc;
    if (a.P == 0) return VGH_OK;
    VGH_REQUIRE(a.w2pack && a.bias2 && a.out2, "conv b2b: the second conv's weights / bias / output are missing");
    VGH_REQUIRE(vgh_conv_b2b_ok(a.ksize, a.stride, a.cout_pad, a.cout2_pad), "conv b2b: no tile for %d -> %d channels", a.cout_pad, a.cout2_pad);
    VGH_REQUIRE(!a.split && !a.res && !a.shuffle && !a.grp_cout && !a.in_fp8 && !a.out_fp8 && !a.out_f32 && a.act != VGH_ACT_SILU && a.act2 != VGH_ACT_SILU,
                "conv b2b: plain bf16 convs with ReLU / no activation only");
    VGH_REQUIRE(a.out2_pitch % 8 == 0 && a.out2_coff % 8 == 0 && a.out2_coff2 % 8 == 0 && a.out2_split % 8 == 0 && a.cout2_store % 8 == 0 && a.cout2_store <= a.cout2_pad,
                "conv b2b: the second output needs 16-byte aligned channel offsets");
    if (!a.b2b_igemm && vgh_conv_ds_b2b_ok(a)) return vgh_launch_conv_ds_b2b(a, stream);  // the stage-1 pair: persistent t tile (ds_b2b.hip)
    return a.cout2_pad == 256 ? launch_b2b_96_any<8>(a, stream) : a.cout2_pad == 192 ? launch_b2b_96_any<6>(a, stream) : launch_b2b_96_any<4>(a, stream);
}

void vgh_pack_conv_weights_host(const float* w, int cout_pad, int ksize, int cin, uint16_t* dst) {
    // dst[kb][cout][slot][8] with slot = chunk ^ ((cout>>2)&3), kb = (ky*ks + kx)*(cin/32) + cb
    const int cblocks = cin / 32, taps = ksize * ksize;
    for (int tap = 0; tap < taps; ++tap)
        for (int cb = 0; cb < cblocks; ++cb) {
            const int kb = tap * cblocks + cb;
            for (int co = 0; co < cout_pad; ++co) {
                const float* src = w + ((size_t)co * taps + tap) * cin + cb * 32;
                uint16_t* d = dst + ((size_t)kb * cout_pad + co) * 32;
                const int sw = (co >> 2) & 3;
                for (int chunk = 0; chunk < 4; ++chunk) {
                    const int slot = chunk ^ sw;
                    for (int e = 0; e < 8; ++e) d[slot * 8 + e] = vgh_f32_to_bf16_host(src[chunk * 8 + e]);
                }
            }
        }
}
