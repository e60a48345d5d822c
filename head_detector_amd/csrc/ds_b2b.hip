// "t" tile (r06): the stage-1 downsample (QARepVGG 3x3 / stride 2, 48 -> 96 channels, yolo_heads_{m,l}_arch_params.yaml:11-17) and the CSP layer's merged conv1|conv2
// (1x1, 96 -> 128 / 192) behind it -- the first back-to-back pair of the TorchScript blob called at head_detector/detector.py:58-59 -- as ONE PERSISTENT launch whose
// weights never move: every wave keeps its 32-cout slice of BOTH convs in registers for the whole launch.
//
// Why (profiles/r06_igemm_trace.txt, r06_family_table_l64.md): on the implicit-GEMM b2b tile this pair is the slowest op of the forward after the 512-cout head conv
// (484 us per 64 images at 1.97 TB/s).  A K step of that tile is one loaded memory round trip (~1 500 cycles for 192 cycles of MFMAs), it has 18 of them per tile, and the
// 3 x 3 taps pull every input line through the LDS-DMA path up to nine times (3.2 GB of L2 -> LDS traffic for 0.63 GB of input).  Here
//   * a tile = 8 x 8 output pixels of one image; its 17 x 17 x 48-channel input patch is fetched ONCE (1.13 x the input bytes), all 34 LDS-DMA pieces of it in flight
//     together and two tiles ahead (three buffers), so a tile costs one round trip, hidden under the previous tile;
//   * the patch lands de-interleaved into the four parity planes [iy & 1][ix & 1] (the LDS-DMA source address is free per lane), so tap (ky, kx) of the stride-2 conv
//     reads plane (ky & 1, kx & 1) at a stride-1 offset: a plane row = one 1-KiB piece = 9 pixels x 6 sixteen-byte chunks, rotated by one chunk on odd rows -- the 16 lanes
//     of every ds_read_b128 lane group (MI355X_MICROARCH.md, LDS: {0-3, 12-15, 20-27}, ...) then cover two adjacent rows x eight pixels = 16 distinct 16-byte slots;
//   * compute wave w owns couts 32 w .. + 31 of the 3x3 conv for both 32-pixel groups: 27 A fragments (9 taps x three 16-channel steps; the padded
//     channels 48 .. 63 of the implicit-GEMM launch carry zero weights and are skipped: they only ever added exact zeros) = 108 VGPRs, loaded once;
//   * the three compute waves trade their bf16 results through LDS in fragment order (what the b2b tile's one wave holds in registers) and run the 1x1 conv with cout
//     groups 2 w and 2 w + 1 from 48 more resident fragment registers each; only the second conv's output is stored.
// Same instructions (v_mfma_f32_32x32x16_bf16), same operand slots, same k order (tap major, channels ascending; then k-block, half) and same roundings as the two
// implicit-GEMM launches: BIT-IDENTICAL outputs (tests/test_gpu_parity.py::test_b2b_pairs_equal_their_two_launches).
// Measured (profiles/r06_ds_tile_ablation.txt): 487 -> 347 us per 64 images; without its stores the launch takes 165 us -- what is left is the write
// path: 629 MB of 192-byte segments inside 768-byte pixels (EXPERIMENTS 8c: such a copy runs at 4.27 TB/s).
// This file also holds the tile's relatives: "u" (STEM = 1: the stem conv inside the same launch, the default for u8 images), "r" (ds_conv_kernel: plain 3x3 / stride-2
// convs with 96 input channels) and the "w" candidates (w_conv_kernel: stride-1, 96 / 128 input channels).
#include <atomic>
#include <type_traits>

#include "vgh_internal.h"

#define AS3 __attribute__((address_space(3)))
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

namespace {

constexpr int DT_ROW = 1024;               // bytes of one plane row (one LDS-DMA piece): 9 pixels x 96 B + the rotation chunk, padded to 64 chunks
constexpr int DT_PLANE = 9 * DT_ROW;       // planes (0, x) hold 9 rows, planes (1, x) 8 (their ninth row is filled with zeros and never read)
constexpr int DT_BUF = 4 * DT_PLANE;       // one patch
constexpr int DT_EX = 3 * DT_BUF;          // two exchange areas (by tile parity): [pixel group][producer wave][half][lane] x 16 B
constexpr int DT_EXB = 12 * 1024;
constexpr int DT_LDS = DT_EX + 2 * DT_EXB;
constexpr int DT_PIECES = 34;              // LDS-DMA pieces of a patch (the ninth row of planes (1, x) does not exist)
constexpr unsigned DT_OOB = 0xC0000000u;   // out of the descriptors' 2-GiB range, and not wrapped past 2^32 by the scalar offsets added to it
static_assert(DT_LDS <= 160 * 1024, "one workgroup per CU");
// STEM = 1 ("u" tile): the stem conv (YoloNASStem, QARepVGG 3 -> 48, 3x3 / stride 2, arch yaml :8-10) in the same launch -- the stem tensor never exists.  A tile's 17 x 17
// stem pixels are computed from the 35 x 35 x 3 image patch straight into the parity planes the LDS-DMA loader fills otherwise.
// v2 (bf16 x 3): the first version ran stem_kernel's exact fp32 chain on v_mfma_f32_32x32x2_f32 -- bit-identical, and 846 us against 654 for the two launches: 280 fp32 MFMAs
// of 64 cycles per tile.  A u8 pixel is an exact bf16, so the stem is ALSO an exact-product bf16 GEMM once the weights are split: w / 255 = hi + mid + lo, three bf16 values
// carrying 24 significant bits.  Per 16 pixels: 3 cout tiles x 3 splits = 9 v_mfma_f32_16x16x32_bf16 of 16 cycles (K = 27 in ONE instruction, no cout padding), fp32 accumulate,
// lo -> mid -> hi.  Every product is exact; what differs from the reference's fmaf chain is the order of the fp32 additions and where / 255 is rounded (the weights instead
// of the pixels): ~1e-7 relative before the ONE rounding to bf16, i.e. a flipped bf16 ulp in ~1e-4 of the stem values (tests/test_gpu_parity.py::test_stem_in_the_pair_launch...).
// K slots are assigned so that a lane's 8 B values are 16 contiguous bytes of the patch (a kernel row's first 8 of 9 (kx, ci) values); the three ninth values ride in the
// fourth k group.  The image patch sits in LDS as bf16 NHWC rows (35 x 105 values at a 216-byte pitch), converted by the loader wave (v_cvt_f32_ubyte, upper half stored).
constexpr int ST_IW = 35, ST_RP = 216;      // image patch: 35 x 35 pixels x 3 bf16 channels, row pitch in bytes
constexpr int ST_IMGB = 7680;                // bytes of one image patch (35 x 216 = 7 560)
constexpr int ST_EX = 2 * DT_BUF;
constexpr int ST_IMG = ST_EX + 2 * DT_EXB;
constexpr int ST_LDS = ST_IMG + 2 * ST_IMGB + 256;
static_assert(ST_LDS <= 160 * 1024 && ST_IW * ST_RP <= ST_IMGB, "one workgroup per CU");
struct StemArgs {
    const uint8_t* image;  // u8 NHWC
    const float* w;        // [27][48], k = (ky * 3 + kx) * 3 + ci
    const float* b;        // [48]
    int Hi, Wi;            // image size (the stem map is Hi / 2 x Wi / 2 = the 3x3 conv's input)
};

// experiments build: s_memtime marks per (workgroup, wave, tile) behind the 8192 rows the other conv kernels use of the tools' trace buffer (tools/ds_trace.py)
#ifdef VGH_EXPERIMENTS
#define DT_MARK(k)                                                                                                                                         \
    do {                                                                                                                                                   \
        if (a.trace && lane == 0 && tno < 8) a.trace[((size_t)8192 * VGH_TRACE_TILES * VGH_TRACE_MARKS) + ((size_t)(blockIdx.x * 4 + w) * 8 + tno) * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define DT_MARK(k) \
    do {           \
    } while (0)
#endif

// work-skipping switches of this kernel: runtime branches inside the K loop's slots cost the experiments build 2 x (measured), so these are compile-time
#ifdef VGH_DT_ABLATE  // compile-time mask (tools: python -m head_detector_amd.build -DVGH_DT_ABLATE=8 ...): 1 no patch loads, 2 no K-loop MFMAs, 8 no stores
#define DT_ABLATE(a, bit) ((VGH_DT_ABLATE) & (bit))
#else
#define DT_ABLATE(a, bit) 0
#endif

#ifndef DT_FULL_WIDTH
#define DT_FULL_WIDTH 0  // 1: every lane's launch takes all CUs (A/B knob) instead of its share
#endif
#ifndef DT_STORE_AUX
#define DT_STORE_AUX 0  // cache-policy bits of the t tile's output stores (A/B knob: 2 = nt, 16 = sc1)
#endif

struct DtDiv {
    unsigned m_per, s_per, m_nsx, s_nsx;
};
__device__ __forceinline__ int dt_div(int n, unsigned m, unsigned s) { return (int)((__umulhi((unsigned)n, m) + (unsigned)n) >> s); }
template <int N>
__device__ __forceinline__ void dt_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void dt_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ unsigned dt_pk(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    const f32x2_t f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf2));
}
__device__ __forceinline__ void dt_sw32(unsigned& x, unsigned& y) {  // lanes 32-63 of x trade places with lanes 0-31 of y
    const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    x = r[0];
    y = r[1];
}

// 8 accumulator values (runs q = 2 m, 2 m + 1 of a 32-cout group) -> this lane's 16 bytes: + bias (packed fp32 adds), bf16, ReLU on the packed bf16 values (v_pk_max_i16
// against 0: a bf16 is negative as int16 exactly when the float is; bound INT16_MIN = no activation) -- the same bits as max(x + b, 0) -> bf16 --, half-wave exchange
__device__ __forceinline__ u32x4_t dt_epi8(const f32x16_t& acc, int m, const f32x4_t b0, const f32x4_t b1, unsigned bound) {
    typedef __attribute__((ext_vector_type(2))) short s16x2;
    const f32x2_t a0 = {acc[8 * m + 0], acc[8 * m + 1]}, a1 = {acc[8 * m + 2], acc[8 * m + 3]}, a2 = {acc[8 * m + 4], acc[8 * m + 5]}, a3 = {acc[8 * m + 6], acc[8 * m + 7]};
    const f32x2_t s0 = a0 + f32x2_t{b0[0], b0[1]}, s1 = a1 + f32x2_t{b0[2], b0[3]}, s2 = a2 + f32x2_t{b1[0], b1[1]}, s3 = a3 + f32x2_t{b1[2], b1[3]};
    auto relu = [&](unsigned d) { return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, d), __builtin_bit_cast(s16x2, bound))); };
    unsigned pa0 = relu(dt_pk(s0[0], s0[1])), pa1 = relu(dt_pk(s1[0], s1[1])), pb0 = relu(dt_pk(s2[0], s2[1])), pb1 = relu(dt_pk(s3[0], s3[1]));
    dt_sw32(pa0, pb0);
    dt_sw32(pa1, pb1);
    return u32x4_t{pa0, pa1, pb0, pb1};
}

// T2 = cout groups (of 32) of the second conv: 4 (M) or 6 (L)
//
// Execution structure (v2; tools/ds_trace.py showed v1 -- three waves doing everything in sequence, two workgroups per CU -- issue-bound: 7 000 cycles of one wave's
// instruction stream per tile of which 2 500 are MFMAs, and two workgroups on a CU serialise because six such waves share four SIMDs):
//   * ONE workgroup of FOUR waves per CU, one per SIMD, up to 512 registers each: waves 0-2 compute, wave 3 only LOADS -- it issues the 34 LDS-DMA pieces of a patch two
//     tiles ahead (three patch buffers) and is the only wave that ever waits on vmcnt; the compute waves' stores drain on their own;
//   * a compute wave runs the 1x1 conv, the second epilogue and the stores of tile k - 1 INSIDE the K loop of tile k: the 24 MFMAs of the second GEMM join the 54 of the
//     3x3 conv in the matrix pipe and the VALU / store work of the epilogue sits in the shadow of those MFMAs (27 slots of two MFMAs each);
//   * ONE barrier per tile: it publishes the loader's patch k AND the exchange area the compute waves wrote at the end of tile k - 1 (two exchange areas, by parity).
template <int T2, int STEM = 0>
__global__ __launch_bounds__(256, 1) void ds_b2b_kernel(const ConvArgs a, const int nsx, const int per, const int total_tiles, const int chunk, const DtDiv dv, const StemArgs sa) {
    constexpr int EXOFF = STEM ? ST_EX : DT_EX;  // exchange areas behind the patch buffers (two with the stem in the launch, three without)
    constexpr int NT2 = 2;  // cout groups of the second conv per compute wave: wave w takes groups 2 w, 2 w + 1 = 64 consecutive couts = one whole 128-byte line of an output
                        // pixel wherever the segment allows (T2 = 4: wave 2 has none)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n32 = lane & 31, hi = lane >> 5, half4 = hi * 4;
    const int xcd = blockIdx.x & 7, gpx = gridDim.x >> 3;
    const int pixb = (int)a.in_pitch * 2;  // bytes per input pixel (96)
    const int rowb = a.W * pixb;
    // this workgroup's tiles: xcd * chunk + first + k * gpx, k < n_my
    const int first = blockIdx.x >> 3;
    const int lim = (total_tiles - xcd * chunk) < chunk ? (total_tiles - xcd * chunk) : chunk;
    const int n_my = first < lim ? (lim - first + gpx - 1) / gpx : 0;
    if (n_my <= 0) return;
    const int tile0 = xcd * chunk + first;
    int tno = 0;
    (void)tno;


    // ---- the stem's share of every wave (STEM = 1): unit g = stem pixels 16 g .. + 15 of the tile's 17 x 17 (row major), all 48 couts; lane = (pixel l % 16, k group l / 16) ----
    bf16x8_t SAw[3][3];  // A operands [cout tile][split hi / mid / lo]: lane (cout 16 t + l % 16, k group kg): k slots 8 kg .. + 7 = kernel row kg, values (kx, ci) 0 .. 7;
                         // kg = 3: the ninth value of rows 0, 1, 2, then zeros
    f32x4_t sbq[3];      // bias of this lane's output channels 16 t + 4 kg .. + 3
    if constexpr (STEM) {
        const int c16 = lane & 15, kg = lane >> 4;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            float wv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ky = kg < 3 ? kg : e, j = kg < 3 ? e : 8;
                const bool live = kg < 3 || e < 3;
                wv[e] = live ? sa.w[((live ? ky : 0) * 9 + j) * 48 + 16 * t + c16] / 255.0f : 0.0f;
            }
            bf16x8_t h, m, l;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                h[e] = (__bf16)wv[e];
                const float r1 = wv[e] - (float)h[e];
                m[e] = (__bf16)r1;
                l[e] = (__bf16)(r1 - (float)m[e]);
            }
            SAw[t][0] = h;
            SAw[t][1] = m;
            SAw[t][2] = l;
            sbq[t] = *(const f32x4_t*)(sa.b + 16 * t + 4 * kg);
        }
    }
#ifdef VGH_DT_DEBUG_STEM
    int dbg_b = 0;
#endif
    // per-lane geometry of this wave's units (unit i = pixel group g0 + i), fixed for the whole launch: stem pixel (sy, sx), its byte offset in an image patch and in a patch's planes
    int su_sy[5], su_sx[5];
    unsigned su_img[5], su_xp[5];
    unsigned su_rd[4];  // the lane's four dword offsets from its patch pixel: k groups 0 - 2 the 16 contiguous bytes of their kernel row; group 3 the ninth value of each row
    if constexpr (STEM) {
        const int c16 = lane & 15, kg = lane >> 4;
        const int g0 = w == 3 ? 15 : w * 5;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int p = (g0 + i) * 16 + c16;
            const int sp = p < 289 ? p : 288;
            const int sy = (sp * 241) >> 12, sx = sp - sy * 17;  // sp / 17 for sp < 289
            su_sy[i] = p < 289 ? sy : -100;                       // (a lane without a pixel: never inside the stem map)
            su_sx[i] = sx;
            su_img[i] = (unsigned)((2 * sy) * ST_RP + 12 * sx);
            const int row = sy >> 1, col = sx >> 1;
            su_xp[i] = (unsigned)(((sy & 1) * 2 + (sx & 1)) * DT_PLANE + row * DT_ROW + (col * 6 + (row & 1)) * 16 + 8 * kg);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) su_rd[i] = (unsigned)(kg < 3 ? kg * ST_RP + 4 * i : (i < 3 ? i : 0) * ST_RP + 16);
    }
    // a wave's units in three PHASES -- every unit's four dwords, then every unit's nine MFMAs, then every unit's epilogue and plane writes: unit by unit, hipcc keeps a unit's
    // LDS reads behind the previous unit's LDS writes (it cannot prove image patch and planes apart) and every unit paid its own read + MFMA-chain latency (~650 cycles each)
    auto stem_units = [&](int tyi, int txi, int ibuf, int xbuf, auto NU) __attribute__((always_inline)) {
        constexpr int N = decltype(NU)::value;
        const int kg = lane >> 4;
        const char* const img0 = smem + ST_IMG + ibuf * ST_IMGB;
        unsigned d[N][4];
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) d[i][r] = *(const unsigned*)(img0 + su_img[i] + su_rd[r]);
        f32x4_t acc[N][3];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const u32x4_t bv = kg < 3 ? u32x4_t{d[i][0], d[i][1], d[i][2], d[i][3]} : u32x4_t{(d[i][0] & 0xffffu) | (d[i][1] << 16), d[i][2] & 0xffffu, 0u, 0u};
            const bf16x8_t bfr = __builtin_bit_cast(bf16x8_t, bv);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                acc[i][t] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int sp3 = 2; sp3 >= 0; --sp3) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(SAw[t][sp3], bfr, acc[i][t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const bool has = su_sy[i] >= 0;
            const int gy = 16 * tyi - 1 + su_sy[i], gx = 16 * txi - 1 + su_sx[i];  // position in the stem map; outside it: the 3x3 conv's zero padding
            const unsigned keepm = (has && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W) ? 0xffffffffu : 0u;
            if (has) {
                typedef __attribute__((ext_vector_type(2))) short s16x2;
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
                char* const xp = smem + xbuf * DT_BUF + su_xp[i];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    // + bias (packed fp32), ONE rounding to bf16, ReLU on the packed values (a bf16 is negative as int16 exactly when the float is), zero outside the stem map (a mask)
                    const f32x2_t s0 = f32x2_t{acc[i][t][0], acc[i][t][1]} + f32x2_t{sbq[t][0], sbq[t][1]}, s1 = f32x2_t{acc[i][t][2], acc[i][t][3]} + f32x2_t{sbq[t][2], sbq[t][3]};
                    const s16x2 z = {0, 0};
                    const unsigned o0 = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, dt_pk(s0[0], s0[1])), z)) & keepm;
                    const unsigned o1 = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, dt_pk(s1[0], s1[1])), z)) & keepm;
                    *(u32x2_t*)(xp + 32 * t) = u32x2_t{o0, o1};
#ifdef VGH_DT_DEBUG_STEM  // diagnostic build: every in-map stem pixel also goes to the stem tensor (a.in)
                    if (keepm) *(u32x2_t*)((uint16_t*)a.in + (((size_t)dbg_b * a.H + gy) * a.W + gx) * a.in_pitch + a.in_coff + 16 * t + 4 * kg) = u32x2_t{o0, o1};
#endif
                }
            }
        }
    };
    auto stem_share = [&](int tile, int ibuf, int xbuf) __attribute__((always_inline)) {
        const int b = dt_div(tile, dv.m_per, dv.s_per);
        const int rem = tile - b * per;
        const int tyi = dt_div(rem, dv.m_nsx, dv.s_nsx), txi = rem - tyi * nsx;
#ifdef VGH_DT_DEBUG_STEM
        dbg_b = b;
#endif
        if (w == 3)
            stem_units(tyi, txi, ibuf, xbuf, std::integral_constant<int, 4>{});
        else
            stem_units(tyi, txi, ibuf, xbuf, std::integral_constant<int, 5>{});
    };

    if (w == 3 && STEM) {
        // =========================== loader wave, stem in the launch: image patches ===========================
        // a patch row = 105 bytes from image byte 3 ix0 (= 3 mod 4 in every tile: rows are multiples of 96 bytes): dword d of the row covers row bytes 4 d - 3 .. 4 d;
        // lane = (row parity, dword): 18 row pairs
        const int d = lane & 31, rpar = lane >> 5;
        int pos[4];  // byte offset of byte t of this lane's dword inside a patch row (bf16 values), or -1
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int j = 4 * d - 3 + t;
            pos[t] = (d <= 26 && j >= 0 && j <= 104) ? 2 * j : -1;
        }
        auto load_img = [&](int tile, unsigned (&q)[18]) __attribute__((always_inline)) {
            const int b = dt_div(tile, dv.m_per, dv.s_per);
            const int rem = tile - b * per;
            const int tyi = dt_div(rem, dv.m_nsx, dv.s_nsx), txi = rem - tyi * nsx;
            const int iy0 = 32 * tyi - 3, ix0 = 32 * txi - 3;
            const int64_t rowbytes = (int64_t)sa.Wi * 3;
            const uint8_t* const base = sa.image + ((int64_t)b * sa.Hi + iy0) * rowbytes + (int64_t)ix0 * 3 - 3 + 4 * d;
#pragma unroll
            for (int i = 0; i < 18; ++i) {
                const int r = 2 * i + rpar;
                const bool ok = r < ST_IW && d <= 26 && iy0 + r >= 0 && !(txi == 0 && d < 3);  // rows above the image, the three columns left of it: zeros (the stem's padding)
                q[i] = ok ? *(const unsigned*)(base + r * rowbytes) : 0u;
            }
        };
        auto store_img = [&](int ibuf, const unsigned (&q)[18]) __attribute__((always_inline)) {
            char* const img = smem + ST_IMG + ibuf * ST_IMGB;
#pragma unroll
            for (int i = 0; i < 18; ++i) {
                const int r = 2 * i + rpar;
                if (r < ST_IW) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (pos[t] >= 0) *(uint16_t*)(img + r * ST_RP + pos[t]) = (uint16_t)(__builtin_bit_cast(unsigned, (float)((q[i] >> (8 * t)) & 255u)) >> 16);  // an integer <= 255 is an exact bf16
                }
            }
        };
        unsigned q[18];
        load_img(tile0, q);
        store_img(0, q);
        dt_barrier();  // image patch 0
        if (n_my > 1) load_img(tile0 + gpx, q);
        stem_share(tile0, 0, 0);
        if (n_my > 1) store_img(1, q);
        for (int k = 0; k < n_my; ++k) {
            DT_MARK(0);
            dt_barrier();  // publishes stem patch k (every wave's share), the exchange area of tile k - 1 and image patch k + 1
            DT_MARK(2);
            if (k + 2 < n_my) load_img(tile0 + (k + 2) * gpx, q);
            if (k + 1 < n_my) stem_share(tile0 + (k + 1) * gpx, (k + 1) & 1, (k + 1) & 1);
            DT_MARK(3);
            if (k + 2 < n_my) store_img(k & 1, q);
            ++tno;
        }
        dt_barrier();
        return;
    }
    if (w == 3) {
        // =========================== loader wave ===========================
        // LDS-DMA source offsets inside a plane row, by the row's parity (odd rows are rotated by one chunk): lane l holds chunk s = l - parity: pixel s / 6, chunk s % 6
        unsigned rel0[2], rel1[2], relL[2];  // planes (y, 0) / planes (y, 1): eight pixels / planes (y, 0) of a tile on the left image edge (pixel 0 = column -1: zeros)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int s = lane - q;
            const bool valid = (unsigned)s < 54u;
            const int col = s / 6, chn = s - col * 6;
            const unsigned r = (unsigned)(col * 2 * pixb + chn * 16);
            rel0[q] = valid ? r : DT_OOB;
            rel1[q] = (valid && col < 8) ? r : DT_OOB;
            relL[q] = (valid && col > 0) ? r : DT_OOB;
        }
        auto issue_patch = [&](int tile, int buf) {
            const int b = dt_div(tile, dv.m_per, dv.s_per);
            const int rem = tile - b * per;
            const int tyi = dt_div(rem, dv.m_nsx, dv.s_nsx), txi = rem - tyi * nsx;
            const int iy0 = 16 * tyi - 1, ix0 = 16 * txi - 1;  // input pixel of patch position (0, 0)
            const bool top = tyi == 0, left = txi == 0;
            const char* const base = (const char*)a.in + (int64_t)a.in_coff * 2 + ((int64_t)(b * a.H + iy0) * a.W + ix0) * pixb;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x80000000, 0x00020000);
            char* const dst = smem + buf * DT_BUF;
#pragma unroll
            for (int plane = 0; plane < 4; ++plane)
#pragma unroll
                for (int r = 0; r < 9; ++r) {
                    const int py = plane >> 1, px = plane & 1, q = r & 1, lr = 2 * r + py;
                    if (lr > 16) continue;  // the ninth row of planes (1, x) is never read
                    unsigned vo = px ? rel1[q] : (left ? relL[q] : rel0[q]);
                    if (lr == 0) vo = top ? DT_OOB : vo;  // row -1 of the image: zeros
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (AS3 void*)(dst + (plane * 9 + r) * DT_ROW), 16, vo, (unsigned)(lr * rowb + px * pixb), 0, 0);
                }
        };
        if (!DT_ABLATE(a, 1)) {
            issue_patch(tile0, 0);
            if (n_my > 1) issue_patch(tile0 + gpx, 1);
        }
        int nb = 2;  // buffer of patch k + 2
        for (int k = 0; k < n_my; ++k) {
            DT_MARK(0);
            if (k + 1 < n_my)
                dt_wait_vm<DT_PIECES>();  // patch k has landed (patch k + 1 may be in flight)
            else
                dt_wait_vm<0>();
            DT_MARK(1);
            dt_barrier();  // publishes patch k; every compute wave is done with patch k - 1 (whose buffer patch k + 2 takes)
            DT_MARK(2);
            if (k + 2 < n_my && !DT_ABLATE(a, 1)) issue_patch(tile0 + (k + 2) * gpx, nb);
            DT_MARK(3);
            nb = nb == 2 ? 0 : nb + 1;
            ++tno;
        }
        dt_barrier();  // the compute waves' last barrier (before they drain the last tile's second GEMM)
        return;
    }

    // =========================== compute waves ===========================
    // ---- once: weights and biases -> registers ----
    bf16x8_t W1[27];  // A fragments of the 3x3 conv: step s = tap * 3 + c covers channels 16 c .. + 15 of tap (ky, kx); lane (n32, hi) = cout 32 w + n32, k = 8 hi .. + 7
    {
        const int co = w * 32 + n32, sw = (co >> 2) & 3;
#pragma unroll
        for (int s = 0; s < 27; ++s) {
            const int tap = s / 3, c = s % 3, kb = tap * 2 + (c >> 1), ch = 2 * (c & 1) + hi;
            W1[s] = *(const bf16x8_t*)(a.wpack + ((size_t)kb * 96 + co) * 32 + ((ch ^ sw) * 8));
        }
    }
    bf16x8_t W2[NT2][3][2];  // A fragments of the 1x1 conv: cout group 2 w + tt, k-block i (32 channels), half m
    f32x4_t BV1[2][2], BV2[NT2][2][2];  // bias of this lane's channels: [m][run]: channels 32 g + 8 (2 m + run) + 4 hi .. + 3
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int u = 0; u < 2; ++u) BV1[m][u] = *(const f32x4_t*)(a.bias + w * 32 + (2 * m + u) * 8 + half4);
#pragma unroll
    for (int tt = 0; tt < NT2; ++tt) {
        const int t = 2 * w + tt;
        const int tc = t < T2 ? t : 0;
        const int co = tc * 32 + n32, sw = (co >> 2) & 3;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int m = 0; m < 2; ++m) W2[tt][i][m] = *(const bf16x8_t*)(a.w2pack + ((size_t)i * (32 * T2) + co) * 32 + (((2 * m + hi) ^ sw) * 8));
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int u = 0; u < 2; ++u) BV2[tt][m][u] = *(const f32x4_t*)(a.bias2 + tc * 32 + (2 * m + u) * 8 + half4);
    }
    // ---- per-lane constants ----
    // this lane's output pixel inside a pixel group: rows / columns chosen per ds_read_b128 lane group (two adjacent rows x eight columns each)
    int ty4, tx;
    if (n32 < 4) ty4 = 0, tx = n32;
    else if (n32 < 12) ty4 = 2, tx = n32 - 4;
    else if (n32 < 16) ty4 = 0, tx = n32 - 8;
    else if (n32 < 20) ty4 = 3, tx = n32 - 16;
    else if (n32 < 28) ty4 = 1, tx = n32 - 20;
    else ty4 = 3, tx = n32 - 24;
    unsigned rb[2];  // fragment read offset of the lane for taps with ky >> 1 = 0 / 1 (the row parity moves the rotation)
#pragma unroll
    for (int kyh = 0; kyh < 2; ++kyh) rb[kyh] = (unsigned)(ty4 * DT_ROW + (tx * 6 + hi + ((ty4 + kyh) & 1)) * 16);
    const unsigned bound1 = a.act == VGH_ACT_RELU ? 0u : 0x80008000u, bound2 = a.act2 == VGH_ACT_RELU ? 0u : 0x80008000u;
    unsigned ovo[2];  // byte offset of the lane's output pixel (+ its 8-channel half) of pixel group j from the tile's first pixel
#pragma unroll
    for (int j = 0; j < 2; ++j) ovo[j] = (unsigned)(((4 * j + ty4) * a.Wo + tx) * (int)a.out2_pitch * 2 + hi * 16);
    uint16_t* const out2 = (uint16_t*)a.out2;
    unsigned ochan2[NT2][2];  // byte offset of the 16 couts (group 2 w + tt, half m) in an output pixel: the second conv's two output segments
#pragma unroll
    for (int tt = 0; tt < NT2; ++tt)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int c = (2 * w + tt) * 32 + 16 * m;
            ochan2[tt][m] = (unsigned)((c >= a.out2_split) ? a.out2_coff2 + (c - a.out2_split) : a.out2_coff + c) * 2u;
        }
    auto out_rsrc = [&](int tile) {  // the tile's output through a buffer descriptor based at its first pixel (per-lane 32-bit offsets, scalar channel offsets)
        const int b = dt_div(tile, dv.m_per, dv.s_per);
        const int rem = tile - b * per;
        const int tyi = dt_div(rem, dv.m_nsx, dv.s_nsx), txi = rem - tyi * nsx;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(out2 + ((size_t)(b * a.Ho + tyi * 8) * a.Wo + txi * 8) * a.out2_pitch), 0, 0x80000000, 0x00020000);
    };

    // ---- the work of the PREVIOUS tile, cut into the slots of a K loop: slot 0 / 6 read pixel group 0 / 1's six B fragments from the exchange area, slots 1-6 / 7-12
    //      run its 6 k steps x NT2 cout groups of the 1x1 conv, slots 8-11 / 14-17 its 2 NT2 epilogue blocks (bias, bf16, ReLU, exchange, one 16-byte store each) ----
    bf16x8_t XF[2][6];
    f32x16_t acc2[2][NT2];
    __amdgpu_buffer_rsrc_t prsrc = out_rsrc(tile0);
    unsigned povo[2] = {DT_OOB, DT_OOB};  // (no previous tile yet: the stores of the first pass are out of range)
    const char* pex = smem + EXOFF;
    auto pend = [&](int s) {
        if (s == 0 || s == 6) {
            const int j = s / 6;
#pragma unroll
            for (int f = 0; f < 6; ++f) XF[j][f] = *(const bf16x8_t*)(pex + (j * 6 + f) * 1024 + lane * 16);
#pragma unroll
            for (int tt = 0; tt < NT2; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[j][tt][r] = 0.0f;
        }
        if ((s >= 1 && s <= 6) || (s >= 7 && s <= 12)) {
            const int j = s >= 7 ? 1 : 0, f = s >= 7 ? s - 7 : s - 1;  // f = 2 i + m: k-block i, half m
#pragma unroll
            for (int tt = 0; tt < NT2; ++tt) acc2[j][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2[tt][f >> 1][f & 1], XF[j][f], acc2[j][tt], 0, 0, 0);
        }
        if ((s >= 8 && s < 8 + 2 * NT2) || (s >= 14 && s < 14 + 2 * NT2)) {
            const int j = s >= 14 ? 1 : 0, e = s >= 14 ? s - 14 : s - 8, tt = e >> 1, m = e & 1;
            if (T2 == 6 || 2 * w + tt < T2) {
                const u32x4_t v = dt_epi8(acc2[j][tt], m, BV2[tt][m][0], BV2[tt][m][1], bound2);
                // (the channel offset rides in the VECTOR offset: with a REGISTER scalar offset hipcc assumes that a 16-byte buffer store has no data hazard and may
                // schedule a VALU write of the data registers right behind it -- measured on gfx950: v_pk_add_f32 into v[0:1] one instruction after the store, and lanes
                // 12-15 / 28-31 stored the NEW value; with a constant scalar offset the compiler inserts the wait state itself)
                if (!DT_ABLATE(a, 8)) __builtin_amdgcn_raw_buffer_store_b128(v, prsrc, povo[j] + ochan2[tt][m], 0, DT_STORE_AUX);
            }
        }
    };
    static_assert(14 + 2 * NT2 <= 27, "the previous tile's work fits the K loop's slots");

    if constexpr (STEM) {
        dt_barrier();  // image patch 0
        stem_share(tile0, 0, 0);
    }
    int pb = 0;  // patch buffer of tile k
    for (int k = 0; k < n_my; ++k) {
        const int tile = tile0 + k * gpx;
        DT_MARK(0);
        dt_barrier();  // patch k is there, and so is the exchange area of tile k - 1
        DT_MARK(2);
        const char* const xb = smem + pb * DT_BUF;
        // ---- 3x3 / stride-2 conv of tile k: 27 k steps x two pixel groups; fragment reads run two steps ahead of the MFMAs ----
        f32x16_t acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
        {
            auto frag = [&](int s, int j) -> bf16x8_t {
                const int ky = s / 9, kx = (s / 3) % 3, c = s % 3;
                const int imm = ((ky & 1) * 2 + (kx & 1)) * DT_PLANE + (4 * j + (ky >> 1)) * DT_ROW + ((kx >> 1) * 6 + c * 2) * 16;
                return *(const bf16x8_t*)(xb + rb[ky >> 1] + imm);
            };
            bf16x8_t F[3][2];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int j = 0; j < 2; ++j) F[p][j] = frag(p, j);
#pragma unroll
            for (int s = 0; s < 27; ++s) {
                if (s + 2 < 27) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) F[(s + 2) % 3][j] = frag(s + 2, j);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!DT_ABLATE(a, 2)) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W1[s], F[s % 3][j], acc[j], 0, 0, 0);
                }
                pend(s);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        DT_MARK(4);
        // ---- its epilogue in registers: bias, bf16, ReLU, half-wave exchange -> lane (pixel n, half hi) holds channels 32 w + 16 m + 8 hi .. + 7: the B operand of
        //      k step (w, m) of the second GEMM; into this tile's exchange area [j][producer wave][m][lane] ----
        char* const ex = smem + EXOFF + (k & 1) * DT_EXB;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int m = 0; m < 2; ++m) *(u32x4_t*)(ex + ((j * 3 + w) * 2 + m) * 1024 + lane * 16) = dt_epi8(acc[j], m, BV1[m][0], BV1[m][1], bound1);
        DT_MARK(5);
        if constexpr (STEM) {
            if (k + 1 < n_my) stem_share(tile0 + (k + 1) * gpx, (k + 1) & 1, (k + 1) & 1);  // this wave's share of the next tile's stem pixels
        }
        // tile k becomes the previous tile
        prsrc = out_rsrc(tile);
        povo[0] = ovo[0];
        povo[1] = ovo[1];
        pex = ex;
        pb = STEM ? (pb ^ 1) : (pb == 2 ? 0 : pb + 1);
        DT_MARK(7);
        ++tno;
    }
    dt_barrier();  // the last tile's exchange area
#pragma unroll
    for (int s = 0; s < 14 + 2 * NT2; ++s) {
        pend(s);
        __builtin_amdgcn_sched_barrier(0);
    }
#ifdef VGH_EXPERIMENTS
    if (a.trace && lane == 0) {  // kernel end + the workgroup's tile count, in unused mark slots of trace tile 7
        unsigned long long* const t7 = a.trace + ((size_t)8192 * VGH_TRACE_TILES * VGH_TRACE_MARKS) + ((size_t)(blockIdx.x * 4 + w) * 8 + 7) * 8;
        t7[6] = __builtin_amdgcn_s_memtime();
        t7[3] = (unsigned long long)n_my;
    }
#endif
}

// ---- "r" tile (r06): the same execution structure for a PLAIN 3x3 / stride-2 conv with 96 input channels (stage-2 / neck-2 downsample: K = 864, HBM-bound on the
//      implicit-GEMM rings at 2.0 - 2.4 TB/s): 54 resident A fragments (216 VGPRs) per compute wave, a pixel = 12 chunks at a 14-chunk pitch (even chunk offsets for the
//      eight pixels of a row, the odd ones for the row below: one rotation chunk as before), a plane row = two LDS-DMA pieces, two patch buffers of 34 rows.  A workgroup
//      owns 96 couts; a 192-cout conv runs as two workgroup classes (cout half = tile parity, fixed per workgroup), the second reading its patches from L2.  The first
//      epilogue stores directly.  Bit-identical to the implicit-GEMM tiles (same instructions, k order, roundings). ----
constexpr int DR_PP = 14, DR_ROW = 2048, DR_NROWS = 34, DR_BUF = DR_NROWS * DR_ROW, DR_LDS = 2 * DR_BUF;
static_assert(DR_LDS <= 160 * 1024, "one workgroup per CU");
__device__ __forceinline__ constexpr int dr_plane_row0(int plane) { return plane == 0 ? 0 : plane == 1 ? 9 : plane == 2 ? 18 : 26; }

__global__ __launch_bounds__(256, 1) void ds_conv_kernel(const ConvArgs a, const int nsx, const int per, const int total_tiles, const int chunk, const DtDiv dv, const int nh) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n32 = lane & 31, hi = lane >> 5, half4 = hi * 4;
    const int xcd = blockIdx.x & 7, gpx = gridDim.x >> 3;
    const int pixb = (int)a.in_pitch * 2;
    const int rowb = a.W * pixb;
    const int first = blockIdx.x >> 3;
    const int lim = (total_tiles - xcd * chunk) < chunk ? (total_tiles - xcd * chunk) : chunk;
    const int n_my = first < lim ? (lim - first + gpx - 1) / gpx : 0;
    if (n_my <= 0) return;
    const int tile0 = xcd * chunk + first;  // work item = (pixel tile, cout half): item / nh, item % nh -- the half is the same for every item of a workgroup (gpx is even)
    const int half = nh > 1 ? tile0 % nh : 0;

    if (w == 3) {
        // =========================== loader wave ===========================
        unsigned rel0[2][2], rel1[2][2], relL[2][2];  // [row parity][piece]: planes (y, 0) / planes (y, 1) / planes (y, 0) of a tile on the left image edge
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int s = h * 64 + lane - q;
                const int col = s / DR_PP, chn = s - col * DR_PP;
                const bool valid = s >= 0 && col < 9 && chn < 12;
                const unsigned r = (unsigned)(col * 2 * pixb + chn * 16);
                rel0[q][h] = valid ? r : DT_OOB;
                rel1[q][h] = (valid && col < 8) ? r : DT_OOB;
                relL[q][h] = (valid && col > 0) ? r : DT_OOB;
            }
        auto issue_patch = [&](int item, int buf) __attribute__((always_inline)) {
            const int tile = nh > 1 ? item / nh : item;
            const int b = dt_div(tile, dv.m_per, dv.s_per);
            const int rem = tile - b * per;
            const int tyi = dt_div(rem, dv.m_nsx, dv.s_nsx), txi = rem - tyi * nsx;
            const int iy0 = 16 * tyi - 1, ix0 = 16 * txi - 1;
            const bool top = tyi == 0, left = txi == 0;
            const char* const base = (const char*)a.in + (int64_t)a.in_coff * 2 + ((int64_t)(b * a.H + iy0) * a.W + ix0) * pixb;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x80000000, 0x00020000);
            char* const dst = smem + buf * DR_BUF;
#pragma unroll
            for (int plane = 0; plane < 4; ++plane)
#pragma unroll
                for (int r = 0; r < 9; ++r) {
                    const int py = plane >> 1, px = plane & 1, q = r & 1, lr = 2 * r + py;
                    if (lr > 16) continue;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        unsigned vo = px ? rel1[q][h] : (left ? relL[q][h] : rel0[q][h]);
                        if (lr == 0) vo = top ? DT_OOB : vo;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (AS3 void*)(dst + (dr_plane_row0(plane) + r) * DR_ROW + h * 1024), 16, vo, (unsigned)(lr * rowb + px * pixb), 0, 0);
                    }
                }
        };
        issue_patch(tile0, 0);
        for (int k = 0; k < n_my; ++k) {
            dt_wait_vm<0>();  // patch k (the only one in flight)
            dt_barrier();     // publishes it; every compute wave is done with patch k - 1, whose buffer patch k + 1 takes
            if (k + 1 < n_my) issue_patch(tile0 + (k + 1) * gpx, (k + 1) & 1);
        }
        return;
    }

    // =========================== compute waves ===========================
    const int cg = half * 3 + w;  // this wave's cout group
    bf16x8_t W1[54];  // step s = tap * 6 + c: channels 16 c .. + 15 of tap (ky, kx)
    {
        const int co = cg * 32 + n32, sw = (co >> 2) & 3;
#pragma unroll
        for (int s = 0; s < 54; ++s) {
            const int tap = s / 6, c = s % 6, kb = tap * 3 + (c >> 1), ch = 2 * (c & 1) + hi;
            W1[s] = *(const bf16x8_t*)(a.wpack + ((size_t)kb * a.cout_pad + co) * 32 + ((ch ^ sw) * 8));
        }
    }
    f32x4_t BV1[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int u = 0; u < 2; ++u) BV1[m][u] = *(const f32x4_t*)(a.bias + cg * 32 + (2 * m + u) * 8 + half4);
    int ty4, tx;
    if (n32 < 4) ty4 = 0, tx = n32;
    else if (n32 < 12) ty4 = 2, tx = n32 - 4;
    else if (n32 < 16) ty4 = 0, tx = n32 - 8;
    else if (n32 < 20) ty4 = 3, tx = n32 - 16;
    else if (n32 < 28) ty4 = 1, tx = n32 - 20;
    else ty4 = 3, tx = n32 - 24;
    unsigned rb[2];
#pragma unroll
    for (int kyh = 0; kyh < 2; ++kyh) rb[kyh] = (unsigned)(ty4 * DR_ROW + (tx * DR_PP + hi + ((ty4 + kyh) & 1)) * 16);
    const unsigned bound1 = a.act == VGH_ACT_RELU ? 0u : 0x80008000u;
    unsigned ovo[2][2];  // byte offset of the lane's output pixel + its 8 couts of (pixel group j, half m) from the tile's first pixel
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int c = cg * 32 + 16 * m;
            const int oc = (c >= a.out_split) ? a.out_coff2 + (c - a.out_split) : a.out_coff + c;
            ovo[j][m] = (unsigned)(((4 * j + ty4) * a.Wo + tx) * (int)a.out_pitch * 2 + hi * 16 + oc * 2);
        }
    uint16_t* const outp = (uint16_t*)a.out;

    for (int k = 0; k < n_my; ++k) {
        const int item = tile0 + k * gpx;
        const int tile = nh > 1 ? item / nh : item;
        dt_barrier();
        const char* const xb = smem + (k & 1) * DR_BUF;
        f32x16_t acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
        {
            auto frag = [&](int s, int j) __attribute__((always_inline)) -> bf16x8_t {
                const int ky = s / 18, kx = (s / 6) % 3, c = s % 6;
                const int imm = (dr_plane_row0((ky & 1) * 2 + (kx & 1)) + 4 * j + (ky >> 1)) * DR_ROW + ((kx >> 1) * DR_PP + c * 2) * 16;
                return *(const bf16x8_t*)(xb + rb[ky >> 1] + imm);
            };
            bf16x8_t F[3][2];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int j = 0; j < 2; ++j) F[p][j] = frag(p, j);
#pragma unroll
            for (int s = 0; s < 54; ++s) {
                if (s + 2 < 54) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) F[(s + 2) % 3][j] = frag(s + 2, j);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W1[s], F[s % 3][j], acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const int b = dt_div(tile, dv.m_per, dv.s_per);
        const int rem = tile - b * per;
        const int tyi = dt_div(rem, dv.m_nsx, dv.s_nsx), txi = rem - tyi * nsx;
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(outp + ((size_t)(b * a.Ho + tyi * 8) * a.Wo + txi * 8) * a.out_pitch), 0, 0x80000000, 0x00020000);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int m = 0; m < 2; ++m) __builtin_amdgcn_raw_buffer_store_b128(dt_epi8(acc[j], m, BV1[m][0], BV1[m][1], bound1), orsrc, ovo[j][m], 0, 0);
    }
}

// ---- "w" tile (r06): 3x3 / STRIDE-1 convs with 96 / 128 input channels and the weights resident in registers.  The ping-pong g / h tiles stream a layer's weights
//      through an LDS ring (two barriers per tap, eight waves in two phase-shifted groups); with K = 9 x 96 = 864 or 9 x 128 = 1 152 a wave's 32-cout slice of the weights
//      is 54 / 72 A fragments = 216 / 288 of its 512 registers, so they can simply stay: NCG = 3 / 4 waves per workgroup (one per cout group, one per SIMD), each running
//      ds_read_b128 -> MFMA over the tile's two 32-pixel groups with NO barrier inside a tile and nothing but B fragments coming out of LDS.  A tile = 8 x 8 output pixels,
//      its 10 x 10 halo patch fetched once by LDS-DMA (every wave issues its share, a tile ahead, two buffers; a pixel = CP chunks at a PP-chunk pitch with the one-chunk
//      row rotation of the stride-2 tiles: the 16 lanes of a ds_read_b128 group cover two rows x eight pixels = 16 distinct slots).  Epilogue in registers as above; RES:
//      + alpha * residual in fp32 before the one rounding.  Same instruction, operand slots, k order and roundings as the implicit-GEMM tiles. ----
template <int CIN16>
struct WGeo {
    static constexpr int CP = 2 * CIN16;               // 16-byte chunks per input pixel
    static constexpr int PP = CIN16 == 6 ? 14 : 18;    // pixel pitch in chunks: = 2 mod 4 (eight neighbours on distinct even slots), >= CP
    static constexpr int PIECES = 3, ROWB = PIECES * 1024;  // a patch row: 10 pixels x PP + 1 <= 192 chunks
    static constexpr int BUF = 10 * ROWB;
    static constexpr int DUMMY = 2 * BUF;
    static constexpr int LDS = DUMMY + 1024;
    static constexpr unsigned DIVM = 65536 / PP + 1;   // s / PP == (s * DIVM) >> 16 for s < 192
    static_assert(10 * PP + 1 <= 64 * PIECES && PP >= CP && PP % 4 == 2, "patch row layout");
};

template <int CIN16, int NCG, int RES>
__global__ __launch_bounds__(64 * NCG, 1) void w_conv_kernel(const ConvArgs a, const int nsx, const int per, const int total_tiles, const int chunk, const DtDiv dv, const int nh) {
    using G = WGeo<CIN16>;
    constexpr int NS = 9 * CIN16, PP = G::PP, ROWB = G::ROWB;
    constexpr int UPW = (30 + NCG - 1) / NCG;  // LDS-DMA pieces per wave and patch (30 real ones; the last waves' extras fill a dummy unit with zeros)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n32 = lane & 31, hi = lane >> 5, half4 = hi * 4;
    const int xcd = blockIdx.x & 7, gpx = gridDim.x >> 3;
    const int pixb = (int)a.in_pitch * 2;
    const int rowb = a.W * pixb;
    const int first = blockIdx.x >> 3;
    const int lim = (total_tiles - xcd * chunk) < chunk ? (total_tiles - xcd * chunk) : chunk;
    const int n_my = first < lim ? (lim - first + gpx - 1) / gpx : 0;
    if (n_my <= 0) return;
    const int tile0 = xcd * chunk + first;  // work item = (pixel tile, cout part): item / nh, item % nh; the part is the same for every item of a workgroup (gpx % nh == 0)
    const int part = nh > 1 ? tile0 % nh : 0;
    const int cg = part * NCG + w;  // this wave's cout group

    // ---- once: weights, bias -> registers ----
    bf16x8_t W1[NS];  // step s = tap * CIN16 + c: channels 16 c .. + 15 of tap (ky, kx)
    {
        const int co = cg * 32 + n32, sw = (co >> 2) & 3;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int tap = s / CIN16, c = s % CIN16, kb = tap * (CIN16 / 2) + (c >> 1), ch = 2 * (c & 1) + hi;
            W1[s] = *(const bf16x8_t*)(a.wpack + ((size_t)kb * a.cout_pad + co) * 32 + ((ch ^ sw) * 8));
        }
    }
    f32x4_t BV1[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int u = 0; u < 2; ++u) BV1[m][u] = *(const f32x4_t*)(a.bias + cg * 32 + (2 * m + u) * 8 + half4);
    int ty4, tx;
    if (n32 < 4) ty4 = 0, tx = n32;
    else if (n32 < 12) ty4 = 2, tx = n32 - 4;
    else if (n32 < 16) ty4 = 0, tx = n32 - 8;
    else if (n32 < 20) ty4 = 3, tx = n32 - 16;
    else if (n32 < 28) ty4 = 1, tx = n32 - 20;
    else ty4 = 3, tx = n32 - 24;
    unsigned rb[2];  // fragment read offset of the lane for taps with ky & 1 = 0 / 1 (the patch row's parity moves the rotation)
#pragma unroll
    for (int par = 0; par < 2; ++par) rb[par] = (unsigned)(ty4 * ROWB + (tx * PP + hi + ((ty4 + par) & 1)) * 16);
    const unsigned bound1 = a.act == VGH_ACT_RELU ? 0u : 0x80008000u;
    const float act_lo = a.act == VGH_ACT_RELU ? 0.0f : -3.0e38f;
    unsigned ovo[2][2];   // byte offset of the lane's output pixel + its 8 couts of (pixel group j, half m) from the tile's first pixel
    unsigned rvo[2][2][2];  // RES: byte offset of the lane's 4 residual channels of (j, m, run u) -- its accumulator rows, before the half-wave exchange
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int c = cg * 32 + 16 * m;
            const int oc = (c >= a.out_split) ? a.out_coff2 + (c - a.out_split) : a.out_coff + c;
            ovo[j][m] = (unsigned)(((4 * j + ty4) * a.Wo + tx) * (int)a.out_pitch * 2 + hi * 16 + oc * 2);
#pragma unroll
            for (int u = 0; u < 2; ++u) rvo[j][m][u] = RES ? (unsigned)(((4 * j + ty4) * a.Wo + tx) * (int)a.res_pitch * 2 + (a.res_coff + c + 8 * u + half4) * 2) : 0u;
        }
    uint16_t* const outp = (uint16_t*)a.out;

    // Everything that is not an MFMA or a fragment read rides in the SLOTS of the K loop (one slot = one k step = two MFMAs = 64 matrix-pipe cycles): the LDS-DMA pieces
    // of the NEXT tile's patch (slots 0 .. UPW - 1: ~12 VALU + one piece each), the epilogue blocks and stores of the PREVIOUS tile (slots EB0, EB0 + 2, ...: its
    // accumulators were copied aside), the residual loads of the CURRENT tile (slot NS - 10).  One wave per SIMD has nobody to hide its non-MFMA instructions behind
    // but its own MFMAs (measured without this: 9 300 cycles per tile for 4 608 cycles of MFMAs).
    constexpr int EB0 = UPW + 2;
    static_assert(EB0 + 8 < NS - 10, "slots");
    f32x16_t acc[2], pacc[2];
    bf16x4_t rres[2][2][2], pres[2][2][2];
    unsigned pov[2][2] = {{DT_OOB, DT_OOB}, {DT_OOB, DT_OOB}};  // (no previous tile yet: the stores of the first pass are out of range)
    __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc((void*)outp, 0, 0x80000000, 0x00020000);
    // next tile's patch: wave-uniform part, set before the K loop that carries its pieces
    bool has_next = false;
    int n_iy0 = 0, n_ix0 = 0, n_buf = 0;
    __amdgpu_buffer_rsrc_t nrsrc = prsrc, rrsrc = prsrc;
    auto next_setup = [&](int item, int buf) __attribute__((always_inline)) {
        const int tile = nh > 1 ? item / nh : item;
        const int b = dt_div(tile, dv.m_per, dv.s_per);
        const int rem = tile - b * per;
        const int tyi = dt_div(rem, dv.m_nsx, dv.s_nsx), txi = rem - tyi * nsx;
        n_iy0 = 8 * tyi - 1;
        n_ix0 = 8 * txi - 1;
        n_buf = buf;
        nrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in + (int64_t)a.in_coff * 2 + ((int64_t)(b * a.H + n_iy0) * a.W + n_ix0) * pixb), 0, 0x80000000, 0x00020000);
    };
    // piece u = w + NCG i of the patch set up by next_setup -> patch row u / 3, third u % 3 (the last waves' extras fill a dummy unit with zeros)
    auto issue_piece = [&](int i) __attribute__((always_inline)) {
        const int u = w + NCG * i;
        const bool real = u < 30;
        const int r = real ? u / 3 : 0, h = real ? u - 3 * r : 0;
        const int sl = h * 64 + lane - (r & 1);
        const int col = (int)(((unsigned)(sl < 0 ? 0 : sl) * G::DIVM) >> 16), chn = sl - col * PP;
        const bool ok = real && sl >= 0 && col < 10 && chn < G::CP && (unsigned)(n_iy0 + r) < (unsigned)a.H && (unsigned)(n_ix0 + col) < (unsigned)a.W;
        const unsigned vo = ok ? (unsigned)(col * pixb + chn * 16) : DT_OOB;
        char* const dst = real ? smem + n_buf * G::BUF + r * ROWB + h * 1024 : smem + G::DUMMY;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(nrsrc, (AS3 void*)dst, 16, vo, real ? (unsigned)(r * rowb) : 0u, 0, 0);
    };
    // epilogue block (pixel group j, half m) of the tile whose accumulators sit in pacc
    auto epi_block = [&](int j, int m) __attribute__((always_inline)) {
        if constexpr (RES) {
            // max(acc + bias, lo) + alpha * residual in fp32 (an fma, as the implicit-GEMM epilogue compiles), ONE rounding, then the half-wave exchange
            unsigned pk[4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(a.alpha, (float)pres[j][m][u][e], fmaxf(pacc[j][8 * m + 4 * u + e] + BV1[m][u][e], act_lo));
                pk[2 * u] = dt_pk(v[0], v[1]);
                pk[2 * u + 1] = dt_pk(v[2], v[3]);
            }
            dt_sw32(pk[0], pk[2]);
            dt_sw32(pk[1], pk[3]);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{pk[0], pk[1], pk[2], pk[3]}, prsrc, pov[j][m], 0, 0);
        } else {
            __builtin_amdgcn_raw_buffer_store_b128(dt_epi8(pacc[j], m, BV1[m][0], BV1[m][1], bound1), prsrc, pov[j][m], 0, 0);
        }
    };
    auto pend = [&](int s) __attribute__((always_inline)) {
        if (s < UPW) {
            if (has_next) issue_piece(s);
        }
        if (s >= EB0 && s < EB0 + 8 && ((s - EB0) & 1) == 0) epi_block((s - EB0) >> 2, ((s - EB0) >> 1) & 1);
        if constexpr (RES) {
            if (s == NS - 10) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
                            rres[j][m][u] = __builtin_bit_cast(bf16x4_t, (u32x2_t)__builtin_amdgcn_raw_buffer_load_b64(rrsrc, rvo[j][m][u], 0, 0));
                        }
            }
        }
    };

    next_setup(tile0, 0);
#pragma unroll
    for (int i = 0; i < UPW; ++i) issue_piece(i);
    for (int k = 0; k < n_my; ++k) {
        const int item = tile0 + k * gpx;
        const int tile = nh > 1 ? item / nh : item;
        if (k == 0)
            dt_wait_vm<0>();
        else
            dt_wait_vm<4 + 8 * RES>();  // this wave's pieces of patch k have landed (behind them in the queue: the stores of tile k - 2, the residual loads of tile k - 1)
        dt_barrier();                   // everybody's pieces; everybody is done with patch k - 1
        has_next = k + 1 < n_my;
        if (has_next) next_setup(tile0 + (k + 1) * gpx, (k + 1) & 1);
        const int b = dt_div(tile, dv.m_per, dv.s_per);
        const int rem = tile - b * per;
        const int tyi = dt_div(rem, dv.m_nsx, dv.s_nsx), txi = rem - tyi * nsx;
        const size_t pix0 = (size_t)(b * a.Ho + tyi * 8) * a.Wo + txi * 8;
        if constexpr (RES) rrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res + pix0 * a.res_pitch), 0, 0x80000000, 0x00020000);
        const char* const xb = smem + (k & 1) * G::BUF;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
        {
            auto frag = [&](int s, int j) __attribute__((always_inline)) -> bf16x8_t {
                const int tap = s / CIN16, c = s % CIN16, ky = tap / 3, kx = tap % 3;
                const int imm = (4 * j + ky) * ROWB + (kx * PP + c * 2) * 16;
                return *(const bf16x8_t*)(xb + rb[ky & 1] + imm);
            };
            constexpr int D = 3;  // fragment reads run D steps ahead of the MFMAs
            bf16x8_t F[D + 1][2];
#pragma unroll
            for (int p = 0; p < D; ++p)
#pragma unroll
                for (int j = 0; j < 2; ++j) F[p][j] = frag(p, j);
#pragma clang loop unroll(full)
            for (int s = 0; s < NS; ++s) {
                if (s + D < NS) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) F[(s + D) % (D + 1)][j] = frag(s + D, j);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W1[s], F[s % (D + 1)][j], acc[j], 0, 0, 0);
                pend(s);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // tile k becomes the previous tile: its accumulators (and residual values) move aside, its epilogue rides in the next K loop
#pragma unroll
        for (int j = 0; j < 2; ++j) pacc[j] = acc[j];
        if constexpr (RES) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int u = 0; u < 2; ++u) pres[j][m][u] = rres[j][m][u];
        }
        prsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(outp + pix0 * a.out_pitch), 0, 0x80000000, 0x00020000);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int m = 0; m < 2; ++m) pov[j][m] = ovo[j][m];
    }
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) epi_block(blk >> 1, blk & 1);  // the last tile's epilogue
}

constexpr int kMaxDev = 16;
template <int T2, int STEM = 0>
int launch_dt(const ConvArgs& a, const StemArgs& sa, hipStream_t st) {
    constexpr int LDS = STEM ? ST_LDS : DT_LDS;
    static std::atomic<int> done[kMaxDev];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) dev = 0;
    if (!done[dev].load(std::memory_order_acquire)) {
        VGH_HIP(hipFuncSetAttribute((const void*)ds_b2b_kernel<T2, STEM>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        done[dev].store(1, std::memory_order_release);
    }
    const int nsx = a.Wo / 8, nsy = a.Ho / 8, per = nsx * nsy;
    const int64_t total = (int64_t)a.B * per;
    VGH_REQUIRE(total < (1ll << 30), "conv b2b: too many tiles");
    const int chunk = (int)((total + 7) / 8);
    int gpx = DT_FULL_WIDTH ? 32 : 32 / (a.grid_share > 1 ? a.grid_share : 1);  // one workgroup per CU; the executor's lane streams take their share of the CUs each
    if (gpx < 8) gpx = 8;
    const int cap = vgh_conv_max_blocks_per_xcd();
    if (cap > 0 && gpx > cap) gpx = cap;
    if (gpx > chunk) gpx = chunk;
    DtDiv dv;
    vgh_fastdiv_magic((unsigned)per, &dv.m_per, &dv.s_per);
    vgh_fastdiv_magic((unsigned)nsx, &dv.m_nsx, &dv.s_nsx);
    hipLaunchKernelGGL((ds_b2b_kernel<T2, STEM>), dim3(gpx * 8), dim3(256), LDS, st, a, nsx, per, (int)total, chunk, dv, sa);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}

}  // namespace

// the pair a (3x3 / stride 2 / pad 1, 48 real input channels at a 48-channel pitch declared as cin = 64, 96 couts) -> 1x1 with 128 / 192 couts, on a map of whole 8 x 8 tiles
int vgh_conv_ds_b2b_ok(const ConvArgs& a) {
    return a.ksize == 3 && a.stride == 2 && a.pad == 1 && a.cin == 64 && a.in_pitch == 48 && a.in_coff % 8 == 0 && a.cout_pad == 96 && a.cout_store == 96 && (a.cout2_pad == 128 || a.cout2_pad == 192) && a.cout2_store == a.cout2_pad &&
           a.H == 2 * a.Ho && a.W == 2 * a.Wo && a.Ho % 8 == 0 && a.Wo % 8 == 0 && (int64_t)a.W * a.in_pitch * 2 * 20 < (1ll << 30) && !a.split && !a.res && !a.shuffle && !a.grp_cout &&
           !a.in_fp8 && !a.out_fp8 && !a.out_f32 && a.act != VGH_ACT_SILU && a.act2 != VGH_ACT_SILU;
}


// plain 3x3 / stride-2 / pad-1 conv, 96 input channels, whole 96-cout workgroup tiles, a map of whole 8 x 8 tiles ("r" tile)
int vgh_conv_ds_ok(const ConvArgs& a) {
    return a.ksize == 3 && a.stride == 2 && a.pad == 1 && a.cin == 96 && a.in_pitch % 8 == 0 && a.in_pitch >= 96 && a.in_coff % 8 == 0 && a.cout_pad % 96 == 0 && a.cout_pad <= 192 &&
           a.cout_store == a.cout_pad && (a.out_split >= a.cout_pad || a.out_split % 16 == 0) && a.out_coff % 8 == 0 && a.out_coff2 % 8 == 0 && a.out_pitch % 8 == 0 && a.H == 2 * a.Ho &&
           a.W == 2 * a.Wo && a.Ho % 8 == 0 && a.Wo % 8 == 0 && (int64_t)a.W * a.in_pitch * 2 * 20 < (1ll << 30) && (int64_t)a.Wo * a.out_pitch * 2 * 9 < (1ll << 30) && !a.split && !a.res &&
           !a.shuffle && !a.grp_cout && !a.in_fp8 && !a.out_fp8 && !a.out_f32 && a.act != VGH_ACT_SILU && !a.w2pack;
}

int vgh_launch_conv_ds(const ConvArgs& a, hipStream_t st) {
    VGH_REQUIRE(vgh_conv_ds_ok(a), "conv: not a 96-channel stride-2 conv on whole 8 x 8 tiles (the r tile)");
    static std::atomic<int> done[kMaxDev];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) dev = 0;
    if (!done[dev].load(std::memory_order_acquire)) {
        VGH_HIP(hipFuncSetAttribute((const void*)ds_conv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DR_LDS));
        done[dev].store(1, std::memory_order_release);
    }
    const int nsx = a.Wo / 8, nsy = a.Ho / 8, per = nsx * nsy, nh = a.cout_pad / 96;
    const int64_t total = (int64_t)a.B * per * nh;
    VGH_REQUIRE(total < (1ll << 30), "conv: too many tiles");
    int chunk = (int)((total + 7) / 8);
    if (nh > 1) chunk += chunk & 1;  // even chunks: item parity = cout half, fixed per workgroup
    int gpx = DT_FULL_WIDTH ? 32 : 32 / (a.grid_share > 1 ? a.grid_share : 1);
    if (gpx < 8) gpx = 8;
    const int cap = vgh_conv_max_blocks_per_xcd();
    if (cap > 0 && gpx > cap) gpx = cap;
    if (gpx > chunk) gpx = chunk;
    if (nh > 1) gpx -= gpx & 1;
    if (gpx < nh) gpx = nh;
    DtDiv dv;
    vgh_fastdiv_magic((unsigned)per, &dv.m_per, &dv.s_per);
    vgh_fastdiv_magic((unsigned)nsx, &dv.m_nsx, &dv.s_nsx);
    hipLaunchKernelGGL(ds_conv_kernel, dim3(gpx * 8), dim3(256), DR_LDS, st, a, nsx, per, (int)total, chunk, dv, nh);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}

// 3x3 / stride-1 / pad-1 conv with 96 (-> cout_pad a multiple of 96) or 128 (-> a multiple of 128) input channels on a map of whole 8 x 8 tiles ("w" tile)
int vgh_conv_w_ok(const ConvArgs& a) {
    const int ncg = a.cin == 96 ? 3 : a.cin == 128 ? 4 : 0;
    return ncg && a.ksize == 3 && a.stride == 1 && a.pad == 1 && a.in_pitch % 8 == 0 && a.in_coff % 8 == 0 && a.cout_pad % (32 * ncg) == 0 && a.cout_pad <= 1024 && a.cout_store == a.cout_pad &&
           (a.out_split >= a.cout_pad || a.out_split % 16 == 0) && a.out_coff % 8 == 0 && a.out_coff2 % 8 == 0 && a.out_pitch % 8 == 0 && a.Ho == a.H && a.Wo == a.W && a.H % 8 == 0 && a.W % 8 == 0 &&
           (int64_t)a.W * a.in_pitch * 2 * 12 < (1ll << 30) && (int64_t)a.Wo * a.out_pitch * 2 * 9 < (1ll << 30) && (!a.res || ((int64_t)a.Wo * a.res_pitch * 2 * 9 < (1ll << 30) && a.res_pitch % 4 == 0 && a.res_coff % 4 == 0)) &&
           !a.split && !a.shuffle && !a.grp_cout && !a.in_fp8 && !a.out_fp8 && !a.out_f32 && a.act != VGH_ACT_SILU && !a.w2pack;
}

template <int CIN16, int NCG, int RES>
static int launch_w(const ConvArgs& a, hipStream_t st) {
    using G = WGeo<CIN16>;
    static std::atomic<int> done[kMaxDev];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) dev = 0;
    if (!done[dev].load(std::memory_order_acquire)) {
        VGH_HIP(hipFuncSetAttribute((const void*)w_conv_kernel<CIN16, NCG, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        done[dev].store(1, std::memory_order_release);
    }
    const int nsx = a.Wo / 8, nsy = a.Ho / 8, per = nsx * nsy, nh = a.cout_pad / (32 * NCG);
    const int64_t total = (int64_t)a.B * per * nh;
    VGH_REQUIRE(total < (1ll << 30), "conv: too many tiles");
    int chunk = (int)((total + 7) / 8);
    chunk = (chunk + nh - 1) / nh * nh;  // chunks of whole pixel tiles: item % nh = cout part, fixed per workgroup
    int gpx = DT_FULL_WIDTH ? 32 : 32 / (a.grid_share > 1 ? a.grid_share : 1);
    if (gpx < 8) gpx = 8;
    const int cap = vgh_conv_max_blocks_per_xcd();
    if (cap > 0 && gpx > cap) gpx = cap;
    if (gpx > chunk) gpx = chunk;
    gpx = gpx / nh * nh;
    if (gpx < nh) gpx = nh;
    DtDiv dv;
    vgh_fastdiv_magic((unsigned)per, &dv.m_per, &dv.s_per);
    vgh_fastdiv_magic((unsigned)nsx, &dv.m_nsx, &dv.s_nsx);
    hipLaunchKernelGGL((w_conv_kernel<CIN16, NCG, RES>), dim3(gpx * 8), dim3(64 * NCG), G::LDS, st, a, nsx, per, (int)total, chunk, dv, nh);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}

int vgh_launch_conv_w(const ConvArgs& a, hipStream_t st) {
    VGH_REQUIRE(vgh_conv_w_ok(a), "conv: not a 96 / 128-channel 3x3 stride-1 conv on whole 8 x 8 tiles (the w tile)");
    if (a.cin == 96) return a.res ? launch_w<6, 3, 1>(a, st) : launch_w<6, 3, 0>(a, st);
    return a.res ? launch_w<8, 4, 1>(a, st) : launch_w<8, 4, 0>(a, st);
}

// `a` prepared, with its b2b fields set and checked by vgh_launch_conv_b2b
int vgh_launch_conv_ds_b2b(const ConvArgs& a, hipStream_t stream) {
    VGH_REQUIRE(vgh_conv_ds_b2b_ok(a), "conv b2b: not a stage-1 downsample pair (the t tile)");
    const StemArgs none{nullptr, nullptr, nullptr, 0, 0};
    return a.cout2_pad == 192 ? launch_dt<6>(a, none, stream) : launch_dt<4>(a, none, stream);
}

// the same pair with the stem conv in front of it in the launch ("u" tile): `a` describes the pair as above (its input tensor -- the stem's output -- is neither written nor
// read: a.in may be null), the stem comes as its fp32 weights [27][48] / bias [48] and the u8 NHWC image batch (Hi x Wi = 2 a.H x 2 a.W)
int vgh_launch_stem_ds_b2b(const ConvArgs& a0, const void* image_u8, int Hi, int Wi, const float* wstem, const float* bstem, hipStream_t stream) {
    ConvArgs a = a0;
    if (int rc = vgh_conv_prepare(a)) return rc;
    if (a.P == 0) return VGH_OK;
    VGH_REQUIRE(a.w2pack && a.bias2 && a.out2 && vgh_conv_ds_b2b_ok(a), "stem + conv b2b: not a stage-1 downsample pair");
    VGH_REQUIRE(a.out2_pitch % 8 == 0 && a.out2_coff % 8 == 0 && a.out2_coff2 % 8 == 0 && a.out2_split % 8 == 0, "stem + conv b2b: the second output needs 16-byte aligned channel offsets");
    VGH_REQUIRE(image_u8 && wstem && bstem && Hi == 2 * a.H && Wi == 2 * a.W && Wi % 32 == 0 && a.act == VGH_ACT_RELU, "stem + conv b2b: image %d x %d for a %d x %d stem map", Hi, Wi, a.H, a.W);
    const StemArgs sa{(const uint8_t*)image_u8, wstem, bstem, Hi, Wi};
    return a.cout2_pad == 192 ? launch_dt<6, 1>(a, sa, stream) : launch_dt<4, 1>(a, sa, stream);
}
