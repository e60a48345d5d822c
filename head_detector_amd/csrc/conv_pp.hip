// "g" tiles: 3x3 / stride-1 convolution as an 8-wave PING-PONG kernel for gfx950 (r04).
//
// Replaces (reference: the 3x3 QARepVGG / ConvBNReLU blocks of the TorchScript blob called at head_detector/detector.py:58-59; definitions
// yolo_head_training/configs/arch_params/yolo_heads_{m,l}_arch_params.yaml:4-137, yolo_head_training/yolo_head/yolo_head_dfl_head.py:74-135)
// the same layers as conv3x3_patch_kernel ("p" / "q" tiles in conv_kernels.inc), with another execution structure:
//
//   * one 512-thread workgroup per CU, two wave GROUPS (waves 0-3 / 4-7: one wave of each group per SIMD) that run the same instruction
//     stream ONE BARRIER APART: while a group issues its 16 MFMAs of a tap ("M phase", s_setprio 1), the other group reads the next tap's
//     fragments from LDS, issues its LDS-DMA prefetches and waits ("L phase").  The matrix pipe of a SIMD always has one wave in its M phase;
//     the non-MFMA work of a wave is hidden behind its partner's MFMAs instead of colliding with it (two independent blocks per CU, as the
//     "p" tiles run, meet in random phase: PMC showed 37 % MFMA-busy with 27 % of the wave cycles parked at waitcnt / barrier).
//   * a wave owns 64 pixels (one 8 x 8 output sub-patch, any image) x BC = 32*TI couts: 128 accumulator registers for TI = 4 and
//     2*TI + 4 fragment reads per 4*TI MFMAs (0.75 KB of LDS reads per MFMA at TI = 4; the 64 x 64 wave tiles of the "p" kernels read 1 KB).
//   * the 10 x 10 halo of a wave's sub-patch is PRIVATE to the wave (7 LDS-DMA units per 32-channel block, double buffered, no cross-wave
//     hand-off); the weights of a tap (BC x 32 channels) are shared by the 8 waves through a 3-stage ring, each wave staging one 1-KiB unit
//     two taps ahead.  8 x 8 sub-patches tile the 160 / 80 / 40-wide maps exactly; a block takes 8 consecutive sub-patches of the batch.
//   * the operand stream runs ACROSS tiles: the last channel block of a tile prefetches the halo and the first two taps of the next one.
//   * epilogue entirely in registers: bias enters as the accumulator's initial value, ReLU, residual, v_cvt_pk_bf16_f32, and
//     v_permlane32_swap pairs so that every lane stores 16 bytes (32 contiguous bytes per pixel per instruction) -- no LDS round trip.
//     The two groups' epilogues fall into different barrier slots, each beside the other group's M phase.
//
// Barrier slots (g0 = waves 0-3, g1 = waves 4-7; B = s_barrier):
//     g0:      L(0) B M(0) B L(1) B M(1) B ...            M(n) B [B]
//     g1:  [B]      B L(0) B M(0) B L(1) B ...  L(n) B M(n) B
// LDS-DMA rules (MI355X_MICROARCH.md, "Two waves per SIMD" item 7): data staged by another wave is read one barrier AFTER the issuing
// wave's counted vmcnt; a stage is re-filled only after a barrier that follows the lgkmcnt(0) of its last readers.
#include <atomic>
#include <type_traits>

#include "vgh_internal.h"

#define AS3 __attribute__((address_space(3)))
#ifndef PP_RES_PREFETCH
#define PP_RES_PREFETCH 0  // 1: touch the residual lines of a tile through the LDS-DMA path during its last channel block (A/B knob; measured +-1 %)
#endif
#ifndef PP_XFRONT
#define PP_XFRONT 1  // 1: the 7 halo units of the next channel block are issued in taps 0-3 (2, 2, 2, 1) instead of one per tap in taps 0-6, so that at the
                     // epilogue nothing young and slow (an HBM halo fetch) sits in the in-order vmcnt queue ahead of the residual loads
#endif
#ifndef PP_STAGGER
#define PP_STAGGER 1  // workgroups that own fewer tiles than the busiest one start late by a pseudo-random share of the tile time they have to spare: the launch is
                      // otherwise in lockstep and every CU stores its 128-KB output tile in the same microseconds (an HBM write burst at ~4.8 TB/s that costs
                      // ~7 us per tile round on the 80^2 x 128 layers, r04_power_per_phase.txt); the kernel's makespan is set by the busiest workgroups, which do not wait
#endif
#ifndef PP_BAR_TAIL
#define PP_BAR_TAIL 0  // g tiles: MFMAs of an M phase issued AFTER the slot's closing barrier (0: the barrier follows the whole run)
#endif
#ifndef PP_RELAX_FIRST_WAIT
#define PP_RELAX_FIRST_WAIT 0  // 1: the first counted wait of a tile allows for the previous tile's 4 TI output stores.  Measured r04 (pp_ab r5e): 3 - 5 % SLOWER and the
                               // tile top unchanged in the trace -- the stall behind the epilogue is not this vmcnt wait but the CU's one vector-memory path: the next
                               // tile's LDS-DMA loads queue behind 128 KB of stores at 16 B/clk whatever the counters allow
#endif
#ifndef PP_INIT_IN_EPILOGUE
#define PP_INIT_IN_EPILOGUE 1  // the next tile's bias enters the accumulators between the last stores of the epilogue (1) or at the tile top (0) (A/B knob)
#endif
#ifndef PP_EPI_SAME_SLOT
#define PP_EPI_SAME_SLOT 1  // g tiles: both groups' epilogues in ONE barrier slot (group 1 defers its end-of-tile barrier) instead of one slot each (A/B knob)
#endif
#ifndef PP_PRIO
#define PP_PRIO 1  // s_setprio 1 around: 1 the M phase (MFMAs), 2 the L phase (fragment reads + LDS-DMA issue), 0 nothing (A/B knob)
#endif
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
typedef __attribute__((ext_vector_type(8))) int i32x8_t;

namespace {

__device__ __forceinline__ void dma16(const void* base, unsigned voffset, unsigned soffset, char* lds_wave_base) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x80000000, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (AS3 void*)lds_wave_base, 16, voffset, soffset, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void barrier_raw() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ float bf_lo(unsigned d) { return __builtin_bit_cast(float, d << 16); }
__device__ __forceinline__ float bf_hi(unsigned d) { return __builtin_bit_cast(float, d & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    const f32x2_t f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf2));  // ONE v_cvt_pk_bf16_f32 (element-wise casts became two converts + a v_perm)
}
// max of two packed bf16 values against a packed int16 bound (v_pk_max_i16): bound 0 = ReLU (bf16 as int16 is negative exactly when the float is),
// bound INT16_MIN = identity
__device__ __forceinline__ unsigned max_pk(unsigned d, unsigned bound) {
    typedef __attribute__((ext_vector_type(2))) short s16x2;
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, d), __builtin_bit_cast(s16x2, bound)));
}
// ---- fp8 (OCP e4m3fn) variants (r05; SURVEY 8(f) N4, the reference's own exporter offers INT8 / FP16: yolo_head_training/yolo_head/exportable_mesh_model.py:175-178,398-411) ----
// F8 = 1: the INPUT tensor and the weights are e4m3 bytes.  A 64-byte halo record / weight row then holds 64 channels instead of 32, a "channel block" is 64
// channels, and one v_mfma_f32_32x32x64_f8f6f4 (16 passes: twice the bf16 FLOPs per cycle) replaces the two 32x32x16 bf16 MFMAs of a (cout group, pixel group)
// pair -- same LDS geometry, same fragment reads (a lane's two 16-byte chunks of a row are the low and the high half of the instruction's 32-byte operand; the
// K order inside a block is irrelevant as long as weights and activations use the same one, which the shared chunk -> lane map guarantees), same barrier slots.
// O8 = 1: the OUTPUT tensor is e4m3 (no residual): 32 couts of a pixel = 32 contiguous bytes = one 16-byte store of each half-wave.
// Either way the epilogue multiplies by a per-cout factor g[c] staged in LDS beside the bias: F8: g = weight_scale[c] * input_scale (/ output_scale), the bias
// vector arrives pre-divided by it (host, vgh_net_create / vgh_conv2d); bf16 -> fp8: g = 1 / output_scale.
__device__ __forceinline__ i32x8_t cat8(bf16x8_t lo, bf16x8_t hi) {
    return __builtin_shufflevector(__builtin_bit_cast(i32x4_t, lo), __builtin_bit_cast(i32x4_t, hi), 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ f32x16_t mfma_f8(bf16x8_t alo, bf16x8_t ahi, bf16x8_t blo, bf16x8_t bhi, f32x16_t c) {
    // cbsz = blgp = 0: both operands e4m3; scale operands 0 -> hipcc selects the unscaled v_mfma_f32_32x32x64_f8f6f4
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cat8(alo, ahi), cat8(blo, bhi), c, 0, 0, 0, 0, 0, 0);
}
// four floats -> four e4m3 bytes (element 0 in the low byte), clamped to [lo, 448] first: the convert itself does not saturate in every mode
__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d, float lo) {
    a = __builtin_amdgcn_fmed3f(a, lo, 448.0f);
    b = __builtin_amdgcn_fmed3f(b, lo, 448.0f);
    c = __builtin_amdgcn_fmed3f(c, lo, 448.0f);
    d = __builtin_amdgcn_fmed3f(d, lo, 448.0f);
    int r = 0;
    r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, r, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
    return (unsigned)r;
}
// ---- int8 variants (F8 = 2 / O8 = 2; VGH_FMT_I8, the reference exporter's QuantizationMode.INT8): the same geometry as e4m3 -- 64 channels per 64-byte record -- with two
//      v_mfma_i32_32x32x32_i8 per (cout group, pixel group) pair, one per 16-byte chunk of a lane, exactly where the bf16 tile issues its two MFMAs.  The accumulator
//      registers hold int32 bit patterns (exact sums) that START AT THE BIAS in accumulator units -- the bias vector of such an op holds int32 bit patterns,
//      rn(bias[c] / (wscale[c] * scale(in))): half a unit is ~1e-5 of a typical output -- and the epilogue computes act(float(acc) * g[c]): the e4m3 epilogue plus one
//      conversion (a separate bias vector in the epilogue cost the 128-cout variant 170 - 1700 bytes of scratch per lane, r05_int8_links.txt) ----
__device__ __forceinline__ f32x16_t mfma_i8(bf16x8_t a, bf16x8_t b, f32x16_t c) {
    typedef __attribute__((ext_vector_type(16))) int i32x16;
    return __builtin_bit_cast(f32x16_t, __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4_t, a), __builtin_bit_cast(i32x4_t, b), __builtin_bit_cast(i32x16, c), 0, 0, 0));
}
// four floats -> four int8 bytes (element 0 in the low byte): clamp to [lo, 127], round to nearest even
__device__ __forceinline__ unsigned pack_i8x4(float a, float b, float c, float d, float lo) {
    const int ia = (int)__builtin_rintf(__builtin_amdgcn_fmed3f(a, lo, 127.0f)), ib = (int)__builtin_rintf(__builtin_amdgcn_fmed3f(b, lo, 127.0f));
    const int ic = (int)__builtin_rintf(__builtin_amdgcn_fmed3f(c, lo, 127.0f)), id = (int)__builtin_rintf(__builtin_amdgcn_fmed3f(d, lo, 127.0f));
    return ((unsigned)ia & 255u) | (((unsigned)ib & 255u) << 8) | (((unsigned)ic & 255u) << 16) | ((unsigned)id << 24);
}
// ---- single-plane fp16 variant (H16 = 1; VGH_FMT_F16, r05: the reference's own FP16 export, exportable_mesh_model.py:177,299,409): same tile, same bytes, same MFMA
//      count as bf16 with v_mfma_f32_32x32x16_f16; the weights carry a per-op power-of-two prescale, so the accumulator starts at bias / out_scale and is multiplied by
//      out_scale at the end; stores saturate at +-65504 ----
template <int H16>
__device__ __forceinline__ f32x16_t mfma16pp(bf16x8_t a, bf16x8_t b, f32x16_t c) {
    typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
    if constexpr (H16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <int H16>
__device__ __forceinline__ unsigned pack16(float lo, float hi) {
    if constexpr (H16) {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
        const f32x2_t f = {__builtin_amdgcn_fmed3f(lo, -65504.0f, 65504.0f), __builtin_amdgcn_fmed3f(hi, -65504.0f, 65504.0f)};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(f, h2));
    } else {
        return pack_bf16(lo, hi);
    }
}
template <int H16>
__device__ __forceinline__ float unpack_lo(unsigned d) {
    if constexpr (H16) {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
        return (float)__builtin_bit_cast(h2, d)[0];
    } else {
        return bf_lo(d);
    }
}
template <int H16>
__device__ __forceinline__ float unpack_hi(unsigned d) {
    if constexpr (H16) {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
        return (float)__builtin_bit_cast(h2, d)[1];
    } else {
        return bf_hi(d);
    }
}
// lanes 32-63 of `a` trade places with lanes 0-31 of `b`
__device__ __forceinline__ void swap32(unsigned& a, unsigned& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}

// GEO = 1 ("s" tiles, r06): the wave's 64 pixels are TWO 4-row x 8-column sub-patches with origins of their own (any image, any position) instead of one 8 x 8 patch:
// pixel group j of the MFMAs = sub-patch j, its 6 x 10 halo = records 60 j .. 60 j + 59 of the wave's stage (120 of 128 records = 8 LDS-DMA units).  20-wide maps
// tile as 8 + 8 + (8 overlapping the previous four columns) x five 4-row bands: 1.2 x the pixels where the 8 x 8 patches cover 24 x 24 = 1.44 x, and twice as many
// independent sub-patches for the small maps (20^2 at batch 32 per lane: 480 instead of 288 -> 30 workgroup tiles per cout tile instead of 36 half-empty ones).
// The overlap columns / rows are computed twice with the same summation order and stored twice with the same bits.
template <int TI, int V = 1, int SC = 0, int GEO = 0>
struct PPGeo {
    static constexpr int BC = 32 * TI, NW = 8;
    static constexpr int XU = GEO ? 8 : 7;  // 16-pixel LDS-DMA units of a wave's halo: 10 x 10 (100 of 112 records used) / two 6 x 10 (120 of 128)
    static constexpr int JOFF = GEO ? 60 * 64 : 4 * 10 * 64;  // bytes from pixel group 0's records to pixel group 1's
    static constexpr int XST = XU * 1024;  // bytes of one halo stage of one wave
    static constexpr int WST = BC * 64;    // bytes of one weight stage (one tap, one 32-channel block, BC couts)
    static constexpr int WU = BC / 16;     // 1-KiB units per weight stage (waves >= WU stage into the dummy unit)
    static constexpr int NWS = V == 2 ? 4 : 3;  // weight ring (v2: prefetch distance 3 taps, one barrier per tap)
    static constexpr int WOFF = NW * 2 * XST;
    static constexpr int DUMMY = WOFF + NWS * WST;
    static constexpr int BIAS = DUMMY + 1024;  // the layer's bias vector (<= 2048 couts; GEO = 1: <= 1024, its halo stages take 16 KB more), staged once per workgroup
    static constexpr int BIASB = GEO ? 4096 : 8192;
    static constexpr int GS = BIAS + BIASB;    // fp8 variants: the per-cout output factors g[c] (same shape as the bias vector)
    static constexpr int LDS = GS + (SC ? 8192 : 0);
};

struct PPDiv {  // n / d == (umulhi(n, m) + n) >> s for n < 2^30 (vgh_fastdiv_magic): the three divisions of a tile decode
    unsigned m_ntc, s_ntc, m_per, s_per, m_nsx, s_nsx;
};
__device__ __forceinline__ int pp_div(int n, unsigned m, unsigned s) { return (int)((__umulhi((unsigned)n, m) + (unsigned)n) >> s); }
struct PPHalo {  // wave-uniform part of a sub-patch's halo addressing
    int base;        // byte offset of halo record (0, 0) = input pixel (y0 - 1, x0 - 1) in the input view
    int y0m1, x0m1;  // its pixel coordinates (range checks)
};
struct PPTile {
    int valid;  // the workgroup has this tile
    int c0;     // first cout of the tile
    int spok;   // this wave's sub-patch exists
    int b, y0, x0;
    int spok1, b1, y01, x01;  // GEO = 1: the wave's second 4 x 8 sub-patch (pixel group 1); (spok, b, y0, x0) is the first
};

// DG = 1 (int8 input, bf16 output): the DIAGONAL BYPASS.  A re-parameterised RepVGG block folds its identity branch into the centre tap, so row c of the folded kernel
// holds one weight w[c][centre][c] 15 - 30 x its rms -- a per-cout int8 grid sized for it leaves ~23 dB for everything else (e4m3: 32 dB).  The host takes that one
// element out of the int8 image (which then reaches ~41 dB) and the epilogue adds it back exactly: + dvec[c] * code(input pixel, channel c), dvec = w[c][centre][c] * scale(in),
// from one dword load per four couts (the pixel's own input bytes: L2 hits, the halo has just been read).  dvec sits in the upper half of the factor region: cout_pad <= 1024.
template <int TI, int V, int F8 = 0, int O8 = 0, int H16 = 0, int DG = 0, int GEO = 0>
__global__ __launch_bounds__(512, 2) void conv3x3_pp_kernel(const ConvArgs a, const int ntc, const int nsx, const int nsy, const int nsp, const int total_tiles, const int chunk, const PPDiv dv) {
    constexpr int SC = (F8 || O8) ? 1 : 0;  // per-cout output factors in the epilogue
    constexpr int ES = F8 ? 1 : 2;          // bytes per input element
    static_assert(!GEO || (V == 1 && !F8 && !O8 && !DG && PP_RES_PREFETCH == 0 && PP_XFRONT == 1), "4 x 8 sub-patches: 16-bit g tiles only");
    static_assert(!(F8 || O8 || H16) || (V == 1 && PP_BAR_TAIL == 0 && PP_RES_PREFETCH == 0), "fp8 / fp16 variants: g tiles only");
    static_assert(!(H16 && (F8 || O8)), "one storage format per variant");
    static_assert(!(F8 && O8) || F8 == O8, "an 8-bit input and an 8-bit output share one format");
    static_assert(!DG || (F8 == 2 && O8 == 0), "the diagonal bypass belongs to the int8 -> bf16 variant");
    // fp16: the accumulator runs in prescaled-weight units: it starts at bias * a.bias_scale (= 1 / out_scale, host-computed: a kernel argument, no register)
    using G = PPGeo<TI, V, SC, GEO>;
    constexpr int BC = G::BC, XST = G::XST, WST = G::WST, WU = G::WU, XU = G::XU, JOFF = G::JOFF;
    // out-of-range marker for buffer offsets (descriptor range 2 GiB): still out of range, and not wrapped past 2^32, after the immediate / scalar
    // offsets the instructions add (channel offsets of the epilogue, weight k-block offsets < 2^30)
    constexpr unsigned OOB = 0xC0000000u;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = w >> 2;
    const int n32 = lane & 31, hi = lane >> 5;
    const int xcd = blockIdx.x & 7, gpx = gridDim.x >> 3;
    const int ncb = a.cblocks;
    const unsigned wkstride = (unsigned)a.cout_pad * 64u;
    const int in_pitch = (int)a.in_pitch;

    // a tile's wave-uniform part: cout tile, this wave's sub-patch, and what the per-lane halo offsets are built from (PPHalo)
    auto decode_tile = [&](int local, PPTile& t, PPHalo& hb, PPHalo& hb1, unsigned& wv) {
        const int tile = xcd * chunk + local;
        t.valid = (local < chunk && tile < total_tiles) ? 1 : 0;
        const int tl = t.valid ? tile : 0;
        const int g8 = pp_div(tl, dv.m_ntc, dv.s_ntc);  // tl / ntc
        t.c0 = (tl - g8 * ntc) * BC;
        const int per = nsy * nsx;
        if constexpr (GEO) {  // two 4 x 8 sub-patches per wave: 2 (8 g8 + w) and the next one; the last sub-patch of a row / the last band is moved back inside the map
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int sp = (g8 * 8 + w) * 2 + j;
                const int ok = (t.valid && sp < nsp) ? 1 : 0;
                const int spc = ok ? sp : 0;
                const int b = pp_div(spc, dv.m_per, dv.s_per);
                const int rem = spc - b * per;
                const int sy = pp_div(rem, dv.m_nsx, dv.s_nsx);
                int y0 = sy * 4, x0 = (rem - sy * nsx) * 8;
                y0 = y0 + 4 > a.H ? a.H - 4 : y0;
                x0 = x0 + 8 > a.W ? a.W - 8 : x0;
                PPHalo& h = j ? hb1 : hb;
                h.base = ES * (((b * a.H + y0 - 1) * a.W + x0 - 1) * in_pitch + a.in_coff);
                h.y0m1 = ok ? y0 - 1 : -64;
                h.x0m1 = x0 - 1;
                if (j == 0)
                    t.spok = ok, t.b = b, t.y0 = y0, t.x0 = x0;
                else
                    t.spok1 = ok, t.b1 = b, t.y01 = y0, t.x01 = x0;
            }
        } else {
            const int sp = g8 * 8 + w;
            t.spok = (t.valid && sp < nsp) ? 1 : 0;
            const int spc = t.spok ? sp : 0;
            t.b = pp_div(spc, dv.m_per, dv.s_per);  // spc / per
            const int rem = spc - t.b * per;
            const int sy = pp_div(rem, dv.m_nsx, dv.s_nsx);  // rem / nsx
            t.y0 = sy * 8;
            t.x0 = (rem - sy * nsx) * 8;
            // byte offset of halo record (0, 0) = input pixel (y0 - 1, x0 - 1): wave-uniform (scalar) and possibly "negative" -- only in-range records add up to an
            // offset that is used
            hb.base = ES * (((t.b * a.H + t.y0 - 1) * a.W + t.x0 - 1) * in_pitch + a.in_coff);
            hb.y0m1 = t.spok ? t.y0 - 1 : -64;  // a missing sub-patch: every record out of range
            hb.x0m1 = t.x0 - 1;
        }
        wv = (t.valid && w < WU) ? (unsigned)(lane * 16 + w * 1024) : OOB;
    };
    // source offset of this lane's 16 bytes of LDS-DMA unit u of the halo: record hp = 16 u + lane / 4 = halo pixel (hp / 10, hp % 10), two 24-bit multiply-adds
    // (hy < 12, hx < 10, row pitch < 2^24 bytes: vgh_conv_pp_fits) instead of the 32-bit multiplies of a full pixel address.  Recomputed per tile -- kept, the
    // lane geometry would cost registers across the whole K loop; the opaque lane copy stops hipcc from hoisting it -- and, for the NEXT tile, just in time in the
    // L phase that issues the unit (r04 trace: decoding all seven units at the top of the last channel block cost every SIMD ~2 000 cycles per tile with all
    // eight waves in it at once; inside an L phase the other group's MFMAs cover it)
    const unsigned rowb = (unsigned)ES * (unsigned)(a.W * in_pitch), pixb = (unsigned)ES * (unsigned)in_pitch;
    auto unit_off = [&](int u, const PPHalo& hb, const PPHalo& hb1) -> unsigned {
        int lane_t = lane;
        asm volatile("" : "+v"(lane_t));
        if constexpr (GEO) {  // record hp = 16 u + lane / 4 < 120: sub-patch hp / 60, halo pixel ((hp % 60) / 10, hp % 10) of its 6 x 10 halo
            const unsigned hp = (unsigned)(u * 16) + ((unsigned)lane_t >> 2);
            const bool j1 = hp >= 60u;
            const unsigned hl = j1 ? hp - 60u : hp;
            const unsigned hy = __umul24(hl, 205u) >> 11;
            const unsigned hx = hl - hy * 10u;
            const int y0m1 = j1 ? hb1.y0m1 : hb.y0m1, x0m1 = j1 ? hb1.x0m1 : hb.x0m1;
            const bool ok = hp < 120u && (unsigned)(y0m1 + (int)hy) < (unsigned)a.H && (unsigned)(x0m1 + (int)hx) < (unsigned)a.W;
            const unsigned rel = __umul24(hy, rowb) + __umul24(hx, pixb) + ((((unsigned)lane_t & 3u) ^ (hy & 3u)) << 4);
            return ok ? (unsigned)(j1 ? hb1.base : hb.base) + rel : OOB;
        }
        const unsigned hp = (unsigned)(u * 16) + ((unsigned)lane_t >> 2);
        const unsigned hy = __umul24(hp, 205u) >> 11;  // hp / 10 for hp < 1029
        const unsigned hx = hp - hy * 10u;
        const bool ok = hp < 100u && (unsigned)(hb.y0m1 + (int)hy) < (unsigned)a.H && (unsigned)(hb.x0m1 + (int)hx) < (unsigned)a.W;
        // source-side swizzle: LDS slot (lane & 3) of halo record (hy, hx) holds channel chunk slot ^ (hy & 3)
        const unsigned rel = __umul24(hy, rowb) + __umul24(hx, pixb) + ((((unsigned)lane_t & 3u) ^ (hy & 3u)) << 4);
        return ok ? (unsigned)hb.base + rel : OOB;
    };

    PPTile cur, nxt;
    PPHalo hb_nxt, hb1_nxt;  // (hb1_nxt: the second sub-patch of a GEO = 1 wave; unused otherwise)
    unsigned xo[XU], wv_cur, wv_nxt;  // xo: halo source offsets of the tile whose halo is being prefetched (the current one, the next one in the last channel block)
    int local = blockIdx.x >> 3;
    decode_tile(local, cur, hb_nxt, hb1_nxt, wv_cur);
#pragma unroll
    for (int u = 0; u < XU; ++u) xo[u] = unit_off(u, hb_nxt, hb1_nxt);
    if (!cur.valid) return;  // workgroup-uniform: no barrier has been executed yet

    if constexpr (PP_STAGGER != 0) {
        const int my_tiles = (chunk - 1 - (int)(blockIdx.x >> 3)) / gpx + 1, max_tiles = (chunk + gpx - 1) / gpx;
        if (my_tiles < max_tiles) {
            // PP_STAGGER == 2: one nap per PIXEL GROUP -- the ntc workgroups that run the cout tiles of the same eight sub-patches (consecutive tiles, same XCD)
            // start together, so that the halo one of them pulls into the XCD's L2 is still there when its siblings ask for it
            const unsigned key = PP_STAGGER == 2 ? (unsigned)((xcd * chunk + (int)(blockIdx.x >> 3)) / ntc) * 8u + (unsigned)xcd : blockIdx.x;
            const unsigned h = (key * 2654435761u) >> 24;  // 0 .. 255
            // ~0.85 of one tile time (ncb * 9 taps * two ~350-ns slots ~ 0.63 us per channel block), in s_sleep units of 1024 cycles (~0.5 us)
            const int naps = (int)((h * (unsigned)(ncb * 11 * (max_tiles - my_tiles))) >> 8);
            for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(16);
        }
    }
    char* const xw = smem + w * (2 * XST);                                          // this wave's two halo stages
    char* const wdst = (w < WU) ? smem + G::WOFF + w * 1024 : smem + G::DUMMY - 0;  // + stage * WST for real units
    const int wdst_step = (w < WU) ? WST : 0;

    // fragment byte offsets (LDS): B operand = halo records of this wave, A operand = weight rows
    //   halo record (hy, hx) at (hy*10 + hx)*64, its 16-byte chunk c at slot c ^ (hy & 3); pixel n32 of MFMA group j sits at sub-patch row 4j + (n32 >> 3),
    //   column n32 & 7; tap (ky, kx) reads record (row + ky, col + kx): kx and j are immediate offsets (+64, +2560), ky changes the swizzle -> own register
    int bofs[3][2];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = (n32 >> 3) + ky;
            bofs[ky][h] = w * (2 * XST) + (r * 10 + (n32 & 7)) * 64 + (((2 * h + hi) ^ (r & 3)) * 16);  // (GEO = 1: the same formula inside a 6 x 10 halo)
        }
    int aofs[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) aofs[h] = G::WOFF + n32 * 64 + (((2 * h + hi) ^ ((n32 >> 2) & 3)) * 16);

    const char* wbase_cur = (const char*)a.wpack + (int64_t)cur.c0 * 64;

    // ---- prologue: halo of channel block 0 and the first two taps of the first tile ----
#pragma unroll
    for (int u = 0; u < XU; ++u) dma16(a.in, xo[u], 0, xw + u * 1024);
    dma16(wbase_cur, wv_cur, (unsigned)(0 * ncb) * wkstride, wdst + 0 * wdst_step);
    dma16(wbase_cur, wv_cur, (unsigned)(1 * ncb) * wkstride, wdst + 1 * wdst_step);
    if constexpr (V == 2) dma16(wbase_cur, wv_cur, (unsigned)(2 * ncb) * wkstride, wdst + 2 * wdst_step);
    // (GEO = 1: the bias region is 4 KB = cout_pad <= 1024; waves 4 .. 7 land their all-zero unit in the dummy)
    dma16(a.bias, (lane * 16 + w * 1024 < a.cout_pad * 4) ? (unsigned)(lane * 16 + w * 1024) : OOB, 0, (w * 1024 < G::BIASB) ? smem + G::BIAS + w * 1024 : smem + G::DUMMY);
    if constexpr (SC) {
        if (!DG || w < 4) dma16(a.gscale, (lane * 16 + w * 1024 < a.cout_pad * 4) ? (unsigned)(lane * 16 + w * 1024) : OOB, 0, smem + G::GS + w * 1024);
    }
    if constexpr (DG) {  // waves 4 .. 7 stage dvec behind the (<= 4 KB: cout_pad <= 1024) factor vector
        if (w >= 4) dma16(a.dvec, (lane * 16 + (w - 4) * 1024 < a.cout_pad * 4) ? (unsigned)(lane * 16 + (w - 4) * 1024) : OOB, 0, smem + G::GS + w * 1024);
    }
    wait_vm<0>();
    barrier_raw();
    if constexpr (V == 1) {
        if (grp) barrier_raw();  // the stagger: group 1 runs one barrier behind group 0
    }

    f32x16_t acc[TI][2];
    int xs = 0;  // halo stage being read (0 / 1): bofs point into it
    const float act_lo = a.act == VGH_ACT_RELU ? 0.0f : __builtin_nanf("");

    // ---- v2 ("h" tiles): ONE barrier per tap.  Per barrier slot s (tap T of the running channel block) the groups run the two halves in opposite order:
    //         g0 (A):  M(s)  [16 MFMAs on the fragments loaded in slot s-1]  ->  LDS-DMA issue  ->  L(s+1)  [fragment reads for the next slot]
    //         g1 (B):  L(s)  ->  LDS-DMA issue  ->  M(s)
    //      so a SIMD's two waves still alternate between the matrix pipe and the LDS / DMA work, but the hand-over inside a slot needs no barrier (a wave's own
    //      L -> M order is its lgkmcnt); the barrier at the slot's end carries the weight-ring hand-off only.  Weights run THREE taps ahead through a 4-stage ring
    //      (tap g lives in stage g & 3; 9 = 1 mod 4, so the stage of tap T of the c-th channel block since launch is (c + T) & 3): W(s+3), issued in slot s, is
    //      waited for at the end of slot s+1 and first read (by A) in slot s+2.  The first slot of a tile has both groups start with their L half. ----
    bf16x8_t fa0[TI], fa1[TI], fb0[2], fb1[2];  // v2: group A's fragments live across the slot barrier
    int cbcount = 0;
    auto load_frags = [&](auto tlc, int sb) __attribute__((always_inline)) {
        constexpr int TL = decltype(tlc)::value;  // 0 .. 9 (9 = tap 0 of the next channel block: the other halo stage)
        constexpr int TT = TL % 9, ky = TT / 3, kx = TT % 3;
        const int rd = ((sb + TL) & 3) * WST;
        const int xd = TL == 9 ? (xs ? -XST : XST) : 0;
        if (!VGH_ABLATE(a, 32)) {
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                fa0[i] = *(const bf16x8_t*)(smem + aofs[0] + rd + i * 2048);
                fa1[i] = *(const bf16x8_t*)(smem + aofs[1] + rd + i * 2048);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                fb0[j] = *(const bf16x8_t*)(smem + bofs[ky][0] + xd + j * 2560 + kx * 64);
                fb1[j] = *(const bf16x8_t*)(smem + bofs[ky][1] + xd + j * 2560 + kx * 64);
            }
        } else {
#pragma unroll
            for (int i = 0; i < TI; ++i) asm volatile("" : "=v"(fa0[i]), "=v"(fa1[i]));
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" : "=v"(fb0[j]), "=v"(fb1[j]));
        }
    };
    auto mma = [&]() __attribute__((always_inline)) {
        if constexpr (PP_PRIO == 1) __builtin_amdgcn_s_setprio(1);
        if (!VGH_ABLATE(a, 2)) {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[i], fb0[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[i], fb1[j], acc[i][j], 0, 0, 0);
        }
        if constexpr (PP_PRIO == 1) __builtin_amdgcn_s_setprio(0);
    };

    // accumulators start at the bias: lane (n32, hi) holds couts c0 + 32 i + 8 q + 4 hi + e of its pixels (register r = 4 q + e).  From LDS (no vmcnt wait behind
    // the previous tile's stores); the first tile's here, every later tile's between the stores of the previous tile's epilogue (r04 trace: 128 v_mov + 16 LDS
    // reads per wave at the tile top were ~1 200 cycles per SIMD and tile with the matrix pipe idle; the epilogue waits on the store path anyway)
    // (a macro, not a lambda: an accumulator array captured by a lambda with several call sites loses its registers -- hipcc puts it in scratch)
#define PP_INIT_ROWS(i, q, c0n)                                                                                     \
    do {                                                                                                            \
        const f32x4_t bv_ = *(const f32x4_t*)(smem + G::BIAS + ((c0n) + (i) * 32 + (q) * 8 + hi * 4) * 4);         \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                                                          \
            const float b0_ = H16 ? bv_[e_] * a.bias_scale : bv_[e_]; /* int8: the vector holds int32 bit patterns */ \
            acc[i][0][(q) * 4 + e_] = b0_;                                                                          \
            acc[i][1][(q) * 4 + e_] = b0_;                                                                          \
        }                                                                                                           \
    } while (0)
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) PP_INIT_ROWS(i, q, cur.c0);
    int trace_no = -1;  // tile counter of this workgroup; -DVGH_EXPERIMENTS: per-tile marks of thread 0 (group 0) -- tile top, first channel block done, K loop done, epilogue done
    while (true) {
        const char* wbase_nxt = wbase_cur;
        ++trace_no;
        VGH_MARK(a, trace_no, 0);
        if constexpr (!PP_INIT_IN_EPILOGUE) {
            if (trace_no > 0) {
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) PP_INIT_ROWS(i, q, cur.c0);
            }
        }
        if constexpr (V == 2) {
            if (!grp) load_frags(std::integral_constant<int, 0>{}, cbcount & 3);  // group A enters its first slot with the fragments of tap 0
        }

        for (int cb = 0; cb < ncb; ++cb) {
            const bool last = cb == ncb - 1;
            // prefetch targets of this channel block: halo of (cb + 1) or of the next tile's block 0; weights two taps ahead
            unsigned rpo = OOB;  // byte offset of this lane's pixel (lane = pixel of the 8 x 8 sub-patch) in the residual tensor, for the L2 touches
            if (last) {
                decode_tile(local + gpx, nxt, hb_nxt, hb1_nxt, wv_nxt);  // its halo offsets: unit by unit in the L phases below
                wbase_nxt = (const char*)a.wpack + (int64_t)nxt.c0 * 64;
                if constexpr (PP_RES_PREFETCH != 0) {
                    const int y = cur.y0 + (lane >> 3), x = cur.x0 + (lane & 7);
                    const int64_t ro = ((((int64_t)cur.b * a.Ho + y) * a.Wo + x) * a.res_pitch + a.res_coff + cur.c0) * 2;
                    if (a.res && cur.spok && y < a.Ho && x < a.Wo && ro + 2 * BC <= 0x7fffffff) rpo = (unsigned)ro;
                }
            }
            const char* const xsrc = last ? (const char*)a.in : (const char*)a.in + (cb + 1) * 64;
            const char* const wb_n = last ? wbase_nxt : wbase_cur;
            const unsigned wv_n = last ? wv_nxt : wv_cur;
            const int cb_n = last ? 0 : cb + 1;
            char* const xpre = xw + (xs ^ 1) * XST;

            if constexpr (V == 2) {
                const int sb = cbcount & 3;
                auto issue = [&](auto tc) __attribute__((always_inline)) {
                    constexpr int T = decltype(tc)::value;
                    constexpr bool LAST = PP_RES_PREFETCH != 0;
                    if (VGH_ABLATE(a, 1)) return;
                    constexpr int TT = T + 3;
                    char* const wd = wdst + ((sb + TT) & 3) * wdst_step;
                    if constexpr (TT < 9)
                        dma16(wbase_cur, wv_cur, (unsigned)(TT * ncb + cb) * wkstride, wd);
                    else
                        dma16(wb_n, wv_n, (unsigned)((TT - 9) * ncb + cb_n) * wkstride, wd);
                    if constexpr (PP_XFRONT) {
                        if constexpr (T < 3) {
                            if (last) xo[2 * T] = unit_off(2 * T, hb_nxt, hb1_nxt), xo[2 * T + 1] = unit_off(2 * T + 1, hb_nxt, hb1_nxt);
                            dma16(xsrc, xo[2 * T], 0, xpre + (2 * T) * 1024);
                            dma16(xsrc, xo[2 * T + 1], 0, xpre + (2 * T + 1) * 1024);
                        } else if constexpr (T == 3) {
                            if (last) xo[6] = unit_off(6, hb_nxt, hb1_nxt);
                            dma16(xsrc, xo[6], 0, xpre + 6 * 1024);
                        }
                    } else {
                        if constexpr (T < 7) {
                            if (last) xo[T] = unit_off(T, hb_nxt, hb1_nxt);
                            dma16(xsrc, xo[T], 0, xpre + T * 1024);
                        }
                    }
                    if constexpr (LAST && T < 3) dma16(a.res, rpo, (unsigned)(T == 0 ? 0 : T == 1 ? 128 : BC * 2 - 16), smem + G::DUMMY);
                };
                auto slot = [&](auto tc) __attribute__((always_inline)) {
                    constexpr int T = decltype(tc)::value;
                    constexpr bool LAST = PP_RES_PREFETCH != 0;
                    __builtin_amdgcn_sched_barrier(0);
                    if (grp) {  // group B: fragments of this tap first ...
                        load_frags(tc, sb);
                        issue(tc);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    mma();  // ONE copy of the 16 MFMAs for both groups: no accumulator merge at a control-flow join
                    __builtin_amdgcn_sched_barrier(0);
                    if (!grp) {  // ... group A: the next tap's fragments afterwards
                        issue(tc);
                        if (T < 8 || !last) load_frags(std::integral_constant<int, T + 1>{}, sb);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    wait_lgkm0();  // this wave's fragment reads are complete before the barrier that releases the stages for re-filling
                    // the weight unit issued first in the previous slot has landed; younger and allowed in flight: the rest of that slot's units and all of this slot's
                    constexpr int XN_T = PP_XFRONT ? (T < 3 ? 2 : T == 3 ? 1 : 0) : (T < 7 ? 1 : 0);
                    constexpr int TP = (T + 8) % 9;
                    constexpr int XN_P = PP_XFRONT ? (TP < 3 ? 2 : TP == 3 ? 1 : 0) : (TP < 7 ? 1 : 0);
                    constexpr int PN_T = (LAST && T < 3) ? 1 : 0, PN_P = (LAST && TP < 3) ? 1 : 0;
                    wait_vm<XN_P + PN_P + 1 + XN_T + PN_T>();
                    barrier_raw();
                };
                slot(std::integral_constant<int, 0>{});
                slot(std::integral_constant<int, 1>{});
                slot(std::integral_constant<int, 2>{});
                slot(std::integral_constant<int, 3>{});
                slot(std::integral_constant<int, 4>{});
                slot(std::integral_constant<int, 5>{});
                slot(std::integral_constant<int, 6>{});
                slot(std::integral_constant<int, 7>{});
                slot(std::integral_constant<int, 8>{});
                ++cbcount;
            } else {
                auto phase = [&](auto tc) {
                    constexpr int T = decltype(tc)::value;
                    constexpr bool LAST = PP_RES_PREFETCH != 0;  // the residual touches are issued in every channel block (out of range = no access except in the last one): uniform vmcnt counts
                    constexpr int ky = T / 3, kx = T % 3, st = T % 3;
                    bf16x8_t a0[TI], a1[TI], b0[2], b1[2];
                    // ---- L phase: fragments of tap T, prefetches, counted waits ----
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (PP_PRIO == 2) __builtin_amdgcn_s_setprio(1);
                    if (!VGH_ABLATE(a, 32)) {
#pragma unroll
                        for (int i = 0; i < TI; ++i) {
                            a0[i] = *(const bf16x8_t*)(smem + aofs[0] + st * WST + i * 2048);
                            a1[i] = *(const bf16x8_t*)(smem + aofs[1] + st * WST + i * 2048);
                        }
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            b0[j] = *(const bf16x8_t*)(smem + bofs[ky][0] + j * JOFF + kx * 64);
                            b1[j] = *(const bf16x8_t*)(smem + bofs[ky][1] + j * JOFF + kx * 64);
                        }
                    } else {  // experiments build: the loop without its fragment traffic (MFMAs on whatever the registers hold)
#pragma unroll
                        for (int i = 0; i < TI; ++i) asm volatile("" : "=v"(a0[i]), "=v"(a1[i]));
#pragma unroll
                        for (int j = 0; j < 2; ++j) asm volatile("" : "=v"(b0[j]), "=v"(b1[j]));
                    }
                    if (!VGH_ABLATE(a, 1)) {
                        constexpr int TT = T + 2, ws = TT % 3;
                        if constexpr (TT < 9)
                            dma16(wbase_cur, wv_cur, (unsigned)(TT * ncb + cb) * wkstride, wdst + ws * wdst_step);
                        else
                            dma16(wb_n, wv_n, (unsigned)((TT - 9) * ncb + cb_n) * wkstride, wdst + ws * wdst_step);
                        if constexpr (PP_XFRONT) {
                            if constexpr (T < (GEO ? 4 : 3)) {  // (GEO = 1: eight units, two per tap in taps 0 - 3)
                                if (last) xo[2 * T] = unit_off(2 * T, hb_nxt, hb1_nxt), xo[2 * T + 1] = unit_off(2 * T + 1, hb_nxt, hb1_nxt);
                                dma16(xsrc, xo[2 * T], 0, xpre + (2 * T) * 1024);
                                dma16(xsrc, xo[2 * T + 1], 0, xpre + (2 * T + 1) * 1024);
                            } else if constexpr (T == 3) {
                                if (last) xo[6] = unit_off(6, hb_nxt, hb1_nxt);
                                dma16(xsrc, xo[6], 0, xpre + 6 * 1024);
                            }
                        } else {
                            if constexpr (T < 7) {
                                if (last) xo[T] = unit_off(T, hb_nxt, hb1_nxt);
                                dma16(xsrc, xo[T], 0, xpre + T * 1024);
                            }
                        }
                        // last channel block: pull this wave's residual lines into L2 (one lane per pixel, 16 bytes of every 128-byte line, into the
                        // dummy LDS unit) so that the epilogue's residual loads are L2 hits instead of two exposed HBM round trips per tile
                        if constexpr (LAST && T < 3) dma16(a.res, rpo, (unsigned)(T == 0 ? 0 : T == 1 ? 128 : BC * 2 - 16), smem + G::DUMMY);
                    }
                    wait_lgkm0();  // this wave's reads of stage st / of its halo are complete before the barrier that releases them for re-filling
                    // the weight unit issued first in L(T-1) (tap T+1) has landed; younger and allowed in flight: the rest of L(T-1) and all of L(T)
                    {
                        constexpr int XN_T = GEO ? (T < 4 ? 2 : 0) : PP_XFRONT ? (T < 3 ? 2 : T == 3 ? 1 : 0) : (T < 7 ? 1 : 0);
                        constexpr int TP = (T + 8) % 9;  // the previous phase
                        constexpr int XN_P = GEO ? (TP < 4 ? 2 : 0) : PP_XFRONT ? (TP < 3 ? 2 : TP == 3 ? 1 : 0) : (TP < 7 ? 1 : 0);
                        constexpr int PN_T = (LAST && T < 3) ? 1 : 0, PN_P = (LAST && TP < 3) ? 1 : 0;
                        constexpr int NW = XN_P + PN_P + 1 + XN_T + PN_T;
                        if constexpr (T == 0 && PP_RELAX_FIRST_WAIT != 0) {
                            // first phase of a tile that follows another one: the awaited unit (tap 1, issued in the previous tile's last phase) is also older
                            // than that tile's 4 TI output stores -- vmcnt is one in-order counter, so without them in the count this wait sat out the whole
                            // store drain of the epilogue (~5 000 cycles, r04_pp_trace.txt) before the tile's first MFMA
                            // (a workgroup's first tile takes the relaxed count too: its prologue has waited for everything)
                            if (cb == 0 && !VGH_ABLATE(a, 8 | 128))
                                wait_vm<NW + 4 * TI>();
                            else
                                wait_vm<NW>();
                        } else {
                            wait_vm<NW>();
                        }
                    }
                    if constexpr (PP_PRIO == 2) __builtin_amdgcn_s_setprio(0);
                    if (!VGH_ABLATE(a, 16)) barrier_raw();
                    // ---- M phase ----
                    if constexpr (PP_PRIO == 1) __builtin_amdgcn_s_setprio(1);
                    // the slot's closing barrier sits PP_BAR_TAIL MFMAs before the end of the run: the wave arrives while the matrix pipe still has work queued, the
                    // other group (long since waiting) is released ~one barrier latency later and starts its own run while this wave issues its tail -- the hand-over
                    // no longer drains the pipe (skeleton without loads / reads / epilogue: 672 cycles per 16-MFMA slot against 512, r04_power_per_phase.txt ablate 41).
                    // The tile's last barrier: group 1 goes straight from its last MFMAs into its epilogue and arrives at this barrier AFTER it, so that
                    // its epilogue shares a barrier slot with group 0's (which follows group 0's side of this barrier) instead of taking a slot of its own
                    const bool closing = !(PP_EPI_SAME_SLOT && T == 8 && last && grp);
                    constexpr int NM = TI * 4, BAR_AT = NM - (PP_BAR_TAIL < NM ? PP_BAR_TAIL : NM - 1);
                    if constexpr (F8 == 1) {
                        if (!VGH_ABLATE(a, 2)) {
#pragma unroll
                            for (int i = 0; i < TI; ++i)
#pragma unroll
                                for (int j = 0; j < 2; ++j) acc[i][j] = mfma_f8(a0[i], a1[i], b0[j], b1[j], acc[i][j]);
                        }
                    } else if (!VGH_ABLATE(a, 2)) {
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int i = 0; i < TI; ++i)
#pragma unroll
                                for (int j = 0; j < 2; ++j) {
                                    if constexpr (F8 == 2)
                                        acc[i][j] = mfma_i8(h ? a1[i] : a0[i], h ? b1[j] : b0[j], acc[i][j]);
                                    else
                                        acc[i][j] = mfma16pp<H16>(h ? a1[i] : a0[i], h ? b1[j] : b0[j], acc[i][j]);
                                    if (PP_BAR_TAIL > 0 && (h * TI + i) * 2 + j + 1 == BAR_AT) {
                                        __builtin_amdgcn_sched_barrier(0);
                                        if (closing && !VGH_ABLATE(a, 64)) barrier_raw();
                                        __builtin_amdgcn_sched_barrier(0);
                                    }
                                }
                    } else if (PP_BAR_TAIL > 0) {
                        if (closing && !VGH_ABLATE(a, 64)) barrier_raw();
                    }
                    if constexpr (PP_PRIO == 1) __builtin_amdgcn_s_setprio(0);
                    if constexpr (PP_BAR_TAIL == 0) {
                        if (closing && !VGH_ABLATE(a, 64)) barrier_raw();
                    }
                };
                phase(std::integral_constant<int, 0>{});
                phase(std::integral_constant<int, 1>{});
                phase(std::integral_constant<int, 2>{});
                phase(std::integral_constant<int, 3>{});
                phase(std::integral_constant<int, 4>{});
                phase(std::integral_constant<int, 5>{});
                phase(std::integral_constant<int, 6>{});
                phase(std::integral_constant<int, 7>{});
                phase(std::integral_constant<int, 8>{});
            }
            // flip the halo stage
            const int d = xs ? -XST : XST;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int h = 0; h < 2; ++h) bofs[ky][h] += d;
            xs ^= 1;
            if (cb == 0) VGH_MARK(a, trace_no, 1);
        }
        VGH_MARK(a, trace_no, 2);

        if constexpr (V == 2) {
            // the fragment registers are dead here (group A reloads them after the accumulators are re-initialised, group B at the top of its next slot); hipcc
            // cannot see that through the two group conditions and would carry 48 registers across the epilogue: fresh (undefined) values end the live ranges
#pragma unroll
            for (int i = 0; i < TI; ++i) asm volatile("" : "=v"(fa0[i]), "=v"(fa1[i]));
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" : "=v"(fb0[j]), "=v"(fb1[j]));
        }
        // ---- epilogue (registers only): ReLU, + alpha * residual, bf16, half-wave exchange, 16-byte stores through buffer descriptors (32-bit offsets,
        //      out-of-range offset = no access: no branch around a load or a store).  ALL residual vectors are requested before the first store: vmcnt is
        //      one in-order counter, so a residual load issued behind a store could only be awaited together with that store's write acknowledgement ----
        if (!VGH_ABLATE(a, 8)) {
            const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, 0x80000000, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, 0x80000000, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, 0x80000000, 0x00020000);
            const int cbase = cur.c0;
            constexpr int OS = O8 ? 1 : 2;                                     // bytes per output element
            const int dsplit = (a.out_coff2 - a.out_split - a.out_coff) * OS;  // byte shift of the second output segment
            unsigned ovb[2], rvb[2];  // byte offsets of this lane's pixel + channel (cbase + 8 hi) in the output / residual tensor
            unsigned ivb[2];          // DG: byte offset of the pixel's own int8 input, channel cbase + 16 hi
            // DG: one 16-byte load per cout group = the pixel's codes of channels cbase + 32 i + 16 hi .. + 15; two half-wave exchanges (diag_run) turn them into the
            // accumulator's runs q = 0 .. 3 (channels 32 i + 8 q + 4 hi .. + 3) -- the same trade the bf16 residual makes
            u32x4_t dq[TI];
            auto load_diag = [&](int j) {
                if constexpr (DG) {
#pragma unroll
                    for (int i = 0; i < TI; ++i) dq[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, ivb[j] + i * 32, 0, 0);
                }
            };
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int y = (GEO && j ? cur.y01 : cur.y0 + 4 * j) + (n32 >> 3), x = (GEO && j ? cur.x01 : cur.x0) + (n32 & 7);
                const bool okpx = (GEO && j ? cur.spok1 : cur.spok) && y < a.Ho && x < a.Wo;
                const int opix = ((GEO && j ? cur.b1 : cur.b) * a.Ho + y) * a.Wo + x;
                // bf16: a lane ends up with couts cbase + 32 i + 16 m + 8 hi .. + 7; e4m3: with couts cbase + 32 i + 16 hi .. + 15
                ovb[j] = okpx ? (unsigned)((opix * (int)a.out_pitch + a.out_coff + cbase + (O8 ? 16 : 8) * hi) * OS) : OOB;
                rvb[j] = okpx ? (unsigned)((opix * (int)a.res_pitch + a.res_coff + cbase + 8 * hi) * 2) : OOB;
                if constexpr (DG) ivb[j] = okpx ? (unsigned)(opix * in_pitch + a.in_coff + cbase + 16 * hi) : OOB;  // 3x3 / stride 1: input pixel = output pixel
            }
            // order: residual loads of pixel group 0 -> arithmetic of group 0 (results held) -> residual loads of group 1 -> stores of group 0 ->
            // arithmetic + stores of group 1: every load is issued ahead of every store, and group 1's vectors land in group 0's dead accumulators
            u32x4_t rv[TI][2], ov[TI][2];
            auto load_res = [&](int j) {
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const int cv = cbase + i * 32 + 16 * m + 8 * hi;
                        rv[i][m] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, (cv < a.cout_store ? rvb[j] : OOB) + (i * 32 + 16 * m) * 2, 0, 0);
                    }
            };
            // No runtime condition inside the per-vector code (r04: `if (a.res)` / `if (a.act == RELU)` / `if (a.nt_out)` per vector cut the epilogue into ~50 tiny
            // basic blocks of dependent VALU chains -- cvt -> max -> swap + s_nop -- that hipcc cannot interleave: ~11 cycles per instruction): the residual case and
            // the address form are wave-uniform branches AROUND the loops, ReLU is a max against a per-launch bound in both number formats.
            const unsigned relu_lo = a.act == VGH_ACT_RELU ? 0u : 0x80008000u;  // packed int16 bound: 0 = ReLU, INT16_MIN = identity
            auto arith = [&](auto jc, auto rc) {
                constexpr int j = decltype(jc)::value;
                constexpr bool RES = decltype(rc)::value;
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    unsigned dr[4] = {0u, 0u, 0u, 0u};  // DG: input codes of run q (see load_diag)
                    if constexpr (DG) {
                        unsigned x0 = dq[i][0], x1 = dq[i][1], x2 = dq[i][2], x3 = dq[i][3];
                        swap32(x0, x1);  // lower half-wave: own channels 0-3 | the partner's 16-19; upper: the partner's 4-7 | own 20-23
                        swap32(x2, x3);  //                  own 8-11 | 24-27;                              12-15 | 28-31
                        dr[0] = x0, dr[1] = x2, dr[2] = x1, dr[3] = x3;
                    }
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        // after the exchange this lane holds couts cv .. cv + 7; before it, runs q = 2m and q = 2m + 1 (couts 32 i + 8 q + 4 hi + e)
                        unsigned pa0, pa1, pb0, pb1;
                        if constexpr (SC) {  // 8-bit input: accumulator units -> real units, per cout (g > 0: commutes with the ReLU below)
#pragma unroll
                            for (int qq = 0; qq < 2; ++qq) {
                                const f32x4_t gv = *(const f32x4_t*)(smem + G::GS + (cbase + i * 32 + (2 * m + qq) * 8 + hi * 4) * 4);
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    if constexpr (F8 == 2) {  // the exact int32 sum (it started at the bias in the same units)
                                        // (through a scalar: __builtin_bit_cast applied to a vector-element lvalue reads element 0 -- hipcc 7.2, seen in the ISA)
                                        const float raw = acc[i][j][(2 * m + qq) * 4 + e];
                                        acc[i][j][(2 * m + qq) * 4 + e] = (float)__builtin_bit_cast(int, raw) * gv[e];
                                    } else {
                                        acc[i][j][(2 * m + qq) * 4 + e] *= gv[e];
                                    }
                                }
                            }
                        }
                        if constexpr (DG) {  // the diagonal bypass
#pragma unroll
                            for (int qq = 0; qq < 2; ++qq) {
                                const f32x4_t dv = *(const f32x4_t*)(smem + G::GS + 4096 + (cbase + i * 32 + (2 * m + qq) * 8 + hi * 4) * 4);
                                const unsigned xw = dr[2 * m + qq];
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[i][j][(2 * m + qq) * 4 + e] += dv[e] * (float)(int)(signed char)(xw >> (8 * e));
                            }
                        }
                        if constexpr (H16) {  // prescaled-weight units -> real units (one factor per op)
#pragma unroll
                            for (int e = 0; e < 8; ++e) acc[i][j][8 * m + e] *= a.out_scale;
                        }
                        if constexpr (RES) {
                            float va[4], vb[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                va[e] = fmaxf(acc[i][j][(2 * m) * 4 + e], act_lo);
                                vb[e] = fmaxf(acc[i][j][(2 * m + 1) * 4 + e], act_lo);
                            }
                            unsigned d0 = rv[i][m][0], d1 = rv[i][m][1], d2 = rv[i][m][2], d3 = rv[i][m][3];
                            swap32(d0, d2);  // back to the accumulator layout: (d0, d1) = run 2m, (d2, d3) = run 2m + 1 of this lane
                            swap32(d1, d3);
                            va[0] += a.alpha * unpack_lo<H16>(d0);
                            va[1] += a.alpha * unpack_hi<H16>(d0);
                            va[2] += a.alpha * unpack_lo<H16>(d1);
                            va[3] += a.alpha * unpack_hi<H16>(d1);
                            vb[0] += a.alpha * unpack_lo<H16>(d2);
                            vb[1] += a.alpha * unpack_hi<H16>(d2);
                            vb[2] += a.alpha * unpack_lo<H16>(d3);
                            vb[3] += a.alpha * unpack_hi<H16>(d3);
                            pa0 = pack16<H16>(va[0], va[1]), pa1 = pack16<H16>(va[2], va[3]);
                            pb0 = pack16<H16>(vb[0], vb[1]), pb1 = pack16<H16>(vb[2], vb[3]);
                        } else {
                            // round first, ReLU on the packed pairs: bf16 as int16 is negative exactly when the float is (rounding keeps the sign)
                            pa0 = max_pk(pack16<H16>(acc[i][j][8 * m + 0], acc[i][j][8 * m + 1]), relu_lo), pa1 = max_pk(pack16<H16>(acc[i][j][8 * m + 2], acc[i][j][8 * m + 3]), relu_lo);
                            pb0 = max_pk(pack16<H16>(acc[i][j][8 * m + 4], acc[i][j][8 * m + 5]), relu_lo), pb1 = max_pk(pack16<H16>(acc[i][j][8 * m + 6], acc[i][j][8 * m + 7]), relu_lo);
                        }
                        swap32(pa0, pb0);
                        swap32(pa1, pb1);
                        ov[i][m] = u32x4_t{pa0, pa1, pb0, pb1};
                    }
                }
            };
            // the whole cout tile stored, into ONE output segment: the offset of a vector is this lane's base + an immediate
            const bool simple = cbase + BC <= a.cout_store && (a.out_split >= cbase + BC || a.out_split <= cbase);
            const unsigned seg = a.out_split <= cbase ? (unsigned)dsplit : 0u;
            auto store_out = [&](int j, auto sc) {
                constexpr bool SIMPLE = decltype(sc)::value;
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        if (VGH_ABLATE(a, 128)) {  // experiments: the epilogue's arithmetic without its stores
                            asm volatile("" ::"v"(ov[i][m]));
                        } else if constexpr (SIMPLE) {
                            __builtin_amdgcn_raw_buffer_store_b128(ov[i][m], rs_out, ovb[j] + seg + (i * 32 + 16 * m) * 2, 0, 0);
                        } else {
                            const int cv = cbase + i * 32 + 16 * m + 8 * hi;
                            const unsigned vo = (cv < a.cout_store ? ovb[j] : OOB) + (unsigned)(cv >= a.out_split ? dsplit : 0) + (i * 32 + 16 * m) * 2;
                            __builtin_amdgcn_raw_buffer_store_b128(ov[i][m], rs_out, vo, 0, 0);
                        }
                    }
            };
            auto stores = [&](int j) {
                if (simple)
                    store_out(j, std::true_type{});
                else
                    store_out(j, std::false_type{});
            };
            // e4m3 output (no residual): runs q = 0 .. 3 of a cout group -> four packed dwords; two half-wave exchanges leave every lane with 16 consecutive couts
            const float lo8 = a.act == VGH_ACT_RELU ? 0.0f : O8 == 2 ? -127.0f : -448.0f;
            auto arith8 = [&](auto jc) {
                constexpr int j = decltype(jc)::value;
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    unsigned R[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4_t gv = *(const f32x4_t*)(smem + G::GS + (cbase + i * 32 + q * 8 + hi * 4) * 4);
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float raw = acc[i][j][q * 4 + e];  // (scalar first: see arith)
                            v[e] = (F8 == 2 ? (float)__builtin_bit_cast(int, raw) : raw) * gv[e];
                        }
                        R[q] = O8 == 2 ? pack_i8x4(v[0], v[1], v[2], v[3], lo8) : pack_fp8x4(v[0], v[1], v[2], v[3], lo8);
                    }
                    swap32(R[0], R[2]);  // lower half-wave: couts 0-3, 4-7, 8-11, 12-15 in R0, R2, R1, R3; upper: 16-19, 20-23, 24-27, 28-31
                    swap32(R[1], R[3]);
                    ov[i][0] = u32x4_t{R[0], R[2], R[1], R[3]};
                }
            };
            auto stores8 = [&](int j) {
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    if (VGH_ABLATE(a, 128))
                        asm volatile("" ::"v"(ov[i][0]));
                    else
                        __builtin_amdgcn_raw_buffer_store_b128(ov[i][0], rs_out, ovb[j] + seg + i * 32, 0, 0);
                }
            };
            if constexpr (O8) {
                arith8(std::integral_constant<int, 0>{});
                __builtin_amdgcn_sched_barrier(0);
                stores8(0);
                __builtin_amdgcn_sched_barrier(0);
                arith8(std::integral_constant<int, 1>{});
            } else if (a.res) {
                load_res(0);
                load_diag(0);
                __builtin_amdgcn_sched_barrier(0);
                arith(std::integral_constant<int, 0>{}, std::true_type{});
                __builtin_amdgcn_sched_barrier(0);
                load_res(1);
                load_diag(1);
                __builtin_amdgcn_sched_barrier(0);
                stores(0);
                __builtin_amdgcn_sched_barrier(0);
                arith(std::integral_constant<int, 1>{}, std::true_type{});
            } else {
                load_diag(0);
                arith(std::integral_constant<int, 0>{}, std::false_type{});
                __builtin_amdgcn_sched_barrier(0);
                load_diag(1);
                __builtin_amdgcn_sched_barrier(0);
                stores(0);
                __builtin_amdgcn_sched_barrier(0);
                arith(std::integral_constant<int, 1>{}, std::false_type{});
            }
            // the tile's last stores (pixel group 1): both groups' accumulators are dead and every store instruction waits its turn on the CU's store path, so
            // the next tile's bias goes into accumulator rows 32 i + 16 m .. + 15 between the stores (plain code, not inside store_out: see PP_INIT_ROWS)
            __builtin_amdgcn_sched_barrier(0);
#define PP_TAIL_STORES(SIMPLE)                                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < TI; ++i) _Pragma("unroll") for (int m = 0; m < 2; ++m) {                                               \
        const int cv = cbase + i * 32 + 16 * m + 8 * hi;                                                                                          \
        const unsigned vo = (SIMPLE) ? ovb[1] + seg : (cv < a.cout_store ? ovb[1] : OOB) + (unsigned)(cv >= a.out_split ? dsplit : 0);            \
        if (VGH_ABLATE(a, 128))                                                                                                                   \
            asm volatile("" ::"v"(ov[i][m]));                                                                                                     \
        else                                                                                                                                      \
            __builtin_amdgcn_raw_buffer_store_b128(ov[i][m], rs_out, vo + (i * 32 + 16 * m) * 2, 0, 0);                                           \
        if constexpr (PP_INIT_IN_EPILOGUE) {                                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                                                                    \
            PP_INIT_ROWS(i, 2 * m, nxt.c0);                                                                                                       \
            PP_INIT_ROWS(i, 2 * m + 1, nxt.c0);                                                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                                                                    \
        }                                                                                                                                         \
    }
            if constexpr (O8) {  // one 16-byte store per cout group, the next tile's bias for that group behind it
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    if (VGH_ABLATE(a, 128))
                        asm volatile("" ::"v"(ov[i][0]));
                    else
                        __builtin_amdgcn_raw_buffer_store_b128(ov[i][0], rs_out, ovb[1] + seg + i * 32, 0, 0);
                    if constexpr (PP_INIT_IN_EPILOGUE) {
                        __builtin_amdgcn_sched_barrier(0);
                        PP_INIT_ROWS(i, 0, nxt.c0);
                        PP_INIT_ROWS(i, 1, nxt.c0);
                        PP_INIT_ROWS(i, 2, nxt.c0);
                        PP_INIT_ROWS(i, 3, nxt.c0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else if (simple) {  // wave-uniform branch AROUND the loop (no runtime condition per vector)
                PP_TAIL_STORES(true)
            } else {
                PP_TAIL_STORES(false)
            }
#undef PP_TAIL_STORES
        } else if constexpr (PP_INIT_IN_EPILOGUE) {  // experiments build, epilogue ablated: the next tile still starts from its bias
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) PP_INIT_ROWS(i, q, nxt.c0);
        }
        VGH_MARK(a, trace_no, 3);
        if constexpr (V == 1 && PP_EPI_SAME_SLOT) {
            if (grp) barrier_raw();  // group 1's deferred end-of-tile barrier (see the last phase)
        }
        if (!nxt.valid) break;
        cur = nxt;
        wv_cur = wv_nxt;
        wbase_cur = wbase_nxt;
        local += gpx;
    }
    if constexpr (V == 1) {
        if (!grp) barrier_raw();  // group 0's share of the last barrier
    }
}

constexpr int kMaxDev = 16;
template <int TI, int V, int F8 = 0, int O8 = 0, int H16 = 0, int DG = 0, int GEO = 0>
int launch_pp(const ConvArgs& a, int ntc, int nsx, int nsy, int nsp, int total, int chunk, int max_blocks_per_xcd, hipStream_t st) {
    using G = PPGeo<TI, V, (F8 || O8) ? 1 : 0, GEO>;
    static std::atomic<int> done[kMaxDev];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) dev = 0;
    if (!done[dev].load(std::memory_order_acquire)) {
        VGH_HIP(hipFuncSetAttribute((const void*)conv3x3_pp_kernel<TI, V, F8, O8, H16, DG, GEO>, hipFuncAttributeMaxDynamicSharedMemorySize, (G::LDS)));
        done[dev].store(1, std::memory_order_release);
    }
    int gpx = 32;  // one workgroup per CU, 32 CUs per XCD
    if (max_blocks_per_xcd > 0 && gpx > max_blocks_per_xcd) gpx = max_blocks_per_xcd;
    if (gpx > chunk) gpx = chunk;
    PPDiv dv;
    vgh_fastdiv_magic((unsigned)ntc, &dv.m_ntc, &dv.s_ntc);
    vgh_fastdiv_magic((unsigned)(nsy * nsx), &dv.m_per, &dv.s_per);
    vgh_fastdiv_magic((unsigned)nsx, &dv.m_nsx, &dv.s_nsx);
    hipLaunchKernelGGL((conv3x3_pp_kernel<TI, V, F8, O8, H16, DG, GEO>), dim3(gpx * 8), dim3(512), (G::LDS), st, a, ntc, nsx, nsy, nsp, total, chunk, dv);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}

}  // namespace

// the epilogue addresses the output and residual tensors through buffer descriptors: 32-bit byte offsets
int vgh_conv_pp_fits(const ConvArgs& a) {
    const int64_t lim = (1ll << 31) - 4096;
    return (int64_t)a.P * a.out_pitch * (a.out_fp8 ? 1 : 2) < lim && (!a.res || (int64_t)a.P * a.res_pitch * 2 < lim) && a.cout_pad <= 2048 &&
           (int64_t)a.W * a.in_pitch * 2 < (1 << 24);  // one input row in 24 bits: the halo offsets are 24-bit multiply-adds
}
int vgh_conv_pp_lds(int bc) { return bc == 128 ? PPGeo<4, 2>::LDS : bc == 96 ? PPGeo<3, 2>::LDS : bc == 64 ? PPGeo<2, 2>::LDS : 0; }
static_assert(PPGeo<4, 1, 1>::LDS <= 160 * 1024, "e4m3 g tiles: bias + factor vectors must fit the 160 KB LDS");
static_assert(PPGeo<4, 1, 0, 1>::LDS <= 160 * 1024, "4 x 8 sub-patch tiles: two 8-KB halo stages per wave + the weight ring must fit the 160 KB LDS");

int vgh_launch_conv_pp(const ConvArgs& a, int bc, int version, int max_blocks_per_xcd, hipStream_t stream) {
    const bool h16 = a.split == VGH_FMT_F16X2 && a.nseg == 1;  // single-plane fp16 (VGH_FMT_F16) through conv_split.hip's pseudo-tiles
    VGH_REQUIRE(a.ksize == 3 && a.stride == 1 && a.fast_epi && !a.out_f32 && !a.shuffle && !a.grp_cout && (!a.split || h16) && a.act != VGH_ACT_SILU, "conv: the ping-pong tiles run plain 3x3 / stride-1 bf16 / fp16 / e4m3 convs only");
    VGH_REQUIRE(!h16 || (version == 1 && !a.in_fp8 && !a.out_fp8 && a.out_scale > 0.0f && a.cout_store == a.cout_pad), "conv: the fp16 ping-pong variant takes whole cout tiles and the op's weight prescale");
    if (a.in_fp8 || a.out_fp8) {
        VGH_REQUIRE(version == 1, "conv: the e4m3 / int8 variants exist for the g tiles only");
        VGH_REQUIRE(a.in_fp8 >= 0 && a.in_fp8 <= 2 && a.out_fp8 >= 0 && a.out_fp8 <= 2 && (!a.in_fp8 || !a.out_fp8 || a.in_fp8 == a.out_fp8), "conv: 8-bit formats are 1 (e4m3) or 2 (int8), the same on both sides");
        VGH_REQUIRE(a.gscale, "conv: an e4m3 conv needs its per-cout output factors (gscale)");
        VGH_REQUIRE(!a.in_fp8 || (a.cin % 64 == 0 && a.in_coff % 16 == 0 && a.in_pitch % 16 == 0), "conv: an e4m3 input view needs cin %% 64 == 0 and 16-byte aligned offset / pitch (cin=%d)", a.cin);
        VGH_REQUIRE(!a.out_fp8 || (!a.res && a.cout_store == a.cout_pad && (a.out_split >= a.cout_pad || a.out_split % bc == 0) && a.out_coff % 16 == 0 && a.out_coff2 % 16 == 0 && a.out_pitch % 16 == 0),
                    "conv: an e4m3 output takes whole cout tiles, 16-byte aligned segments and no residual");
    }
    VGH_REQUIRE(a.cout_pad % bc == 0 && a.cout_pad <= 2048, "conv: cout_pad %d is not a multiple of the %d-cout ping-pong tile (or above 2048)", a.cout_pad, bc);
    VGH_REQUIRE(vgh_conv_pp_fits(a), "conv: output / residual tensor above 2 GiB (32-bit buffer offsets), or an input row above 16 MiB");
    const bool geo = version == 3;  // "s" tiles: two 4 x 8 sub-patches per wave
    VGH_REQUIRE(!geo || (a.Wo >= 8 && a.Ho >= 4 && a.cout_pad <= 1024 && !a.in_fp8 && !a.out_fp8 && !h16), "conv: the 4 x 8 sub-patch tiles need a bf16 map of at least 4 x 8 pixels and cout_pad <= 1024");
    const int nsx = (a.Wo + 7) / 8, nsy = geo ? (a.Ho + 3) / 4 : (a.Ho + 7) / 8, ntc = a.cout_pad / bc;
    const int64_t nsp = (int64_t)a.B * nsy * nsx;
    const int64_t total = (nsp + (geo ? 15 : 7)) / (geo ? 16 : 8) * ntc;
    VGH_REQUIRE(total < (1ll << 30), "conv: too many tiles");
    const int chunk = (int)((total + 7) / 8);
    if (a.in_fp8 || a.out_fp8) {
        const int f = a.in_fp8 ? 1 : 0, o = a.out_fp8 ? 1 : 0;
        const bool i8 = a.in_fp8 == 2 || a.out_fp8 == 2;
#define PP_F8_CASE(TI_, Q_)                                                                                            \
    if (bc == 32 * TI_) {                                                                                              \
        if (f && o) return launch_pp<TI_, 1, Q_, Q_>(a, ntc, nsx, nsy, (int)nsp, (int)total, chunk, max_blocks_per_xcd, stream); \
        if (f) return launch_pp<TI_, 1, Q_, 0>(a, ntc, nsx, nsy, (int)nsp, (int)total, chunk, max_blocks_per_xcd, stream);       \
        return launch_pp<TI_, 1, 0, Q_>(a, ntc, nsx, nsy, (int)nsp, (int)total, chunk, max_blocks_per_xcd, stream);              \
    }
        if (a.dvec) {
            VGH_REQUIRE(a.in_fp8 == 2 && !a.out_fp8 && a.cout_pad <= 1024 && a.cin >= a.cout_pad && a.Ho == a.H && a.Wo == a.W,
                        "conv: the diagonal bypass (dvec) belongs to an int8 -> bf16 conv with cout_pad <= min(cin, 1024)");
            switch (bc) {  // (no 128-cout variant: it spills ~90 registers; vgh_conv_pick_cfg / cfg_ok_for keep such ops on 96 / 64)
                case 96: return launch_pp<3, 1, 2, 0, 0, 1>(a, ntc, nsx, nsy, (int)nsp, (int)total, chunk, max_blocks_per_xcd, stream);
                case 64: return launch_pp<2, 1, 2, 0, 0, 1>(a, ntc, nsx, nsy, (int)nsp, (int)total, chunk, max_blocks_per_xcd, stream);
            }
        }
        if (i8) {
            PP_F8_CASE(4, 2)
            PP_F8_CASE(3, 2)
            PP_F8_CASE(2, 2)
        } else {
            PP_F8_CASE(4, 1)
            PP_F8_CASE(3, 1)
            PP_F8_CASE(2, 1)
        }
#undef PP_F8_CASE
        VGH_REQUIRE(false, "conv: no e4m3 ping-pong tile with %d couts", bc);
    }
    if (h16) {
        ConvArgs ah = a;
        ah.bias_scale = 1.0f / a.out_scale;
        switch (bc) {
            case 128: return launch_pp<4, 1, 0, 0, 1>(ah, ntc, nsx, nsy, (int)nsp, (int)total, chunk, max_blocks_per_xcd, stream);
            case 96: return launch_pp<3, 1, 0, 0, 1>(ah, ntc, nsx, nsy, (int)nsp, (int)total, chunk, max_blocks_per_xcd, stream);
            case 64: return launch_pp<2, 1, 0, 0, 1>(ah, ntc, nsx, nsy, (int)nsp, (int)total, chunk, max_blocks_per_xcd, stream);
        }
        VGH_REQUIRE(false, "conv: no fp16 ping-pong tile with %d couts", bc);
    }
    if (geo) {
        switch (bc) {
            case 128: return launch_pp<4, 1, 0, 0, 0, 0, 1>(a, ntc, nsx, nsy, (int)nsp, (int)total, chunk, max_blocks_per_xcd, stream);
            case 96: return launch_pp<3, 1, 0, 0, 0, 0, 1>(a, ntc, nsx, nsy, (int)nsp, (int)total, chunk, max_blocks_per_xcd, stream);
            case 64: return launch_pp<2, 1, 0, 0, 0, 0, 1>(a, ntc, nsx, nsy, (int)nsp, (int)total, chunk, max_blocks_per_xcd, stream);
        }
        VGH_REQUIRE(false, "conv: no 4 x 8 ping-pong tile with %d couts", bc);
    }
    switch (bc + (version == 2 ? 1 : 0)) {
        case 128: return launch_pp<4, 1>(a, ntc, nsx, nsy, (int)nsp, (int)total, chunk, max_blocks_per_xcd, stream);
        case 96: return launch_pp<3, 1>(a, ntc, nsx, nsy, (int)nsp, (int)total, chunk, max_blocks_per_xcd, stream);
        case 64: return launch_pp<2, 1>(a, ntc, nsx, nsy, (int)nsp, (int)total, chunk, max_blocks_per_xcd, stream);
        case 129: return launch_pp<4, 2>(a, ntc, nsx, nsy, (int)nsp, (int)total, chunk, max_blocks_per_xcd, stream);
        case 97: return launch_pp<3, 2>(a, ntc, nsx, nsy, (int)nsp, (int)total, chunk, max_blocks_per_xcd, stream);
        case 65: return launch_pp<2, 2>(a, ntc, nsx, nsy, (int)nsp, (int)total, chunk, max_blocks_per_xcd, stream);
    }
    VGH_REQUIRE(false, "conv: no ping-pong tile with %d couts", bc);
    return VGH_OK;
}

// ---- e4m3 host side ------------------------------------------------------------------------------------------------------------------------
// fp32 -> OCP e4m3fn (1-4-3, bias 7, no infinities, 0x7f / 0xff = NaN), round to nearest even, SATURATING at +-448 (what v_cvt_pk_fp8_f32 does to a value
// the epilogue has clamped; torch.float8_e4m3fn rounds the same way below the clamp)
uint8_t vgh_f32_to_e4m3_host(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint8_t sign = (uint8_t)((u >> 24) & 0x80);
    if ((u & 0x7fffffffu) > 0x7f800000u) return sign | 0x7f;
    float a = f < 0 ? -f : f;
    if (a >= 448.0f) return sign | 0x7e;
    if (a < 0.015625f) {  // below 2^-6: subnormal grid of 2^-9 (8 * 2^-9 = the smallest normal, whose code is 8 as well)
        const float q = __builtin_nearbyintf(a * 512.0f);
        return sign | (uint8_t)q;
    }
    int e;
    const float m = frexpf(a, &e);  // a = m * 2^e, m in [0.5, 1)
    int E = e - 1;                  // a = (2m) * 2^E, 2m in [1, 2)
    int r = (int)__builtin_nearbyintf((2.0f * m - 1.0f) * 8.0f);
    if (r == 8) {
        r = 0;
        ++E;
    }
    if (E > 8 || (E == 8 && r == 7)) return sign | 0x7e;
    return sign | (uint8_t)(((E + 7) << 3) | r);
}

// dense [cout_pad][ks][ks][cin] f32 -> e4m3 image [tap * (cin / 64) + cb][cout][64 B] (16-byte chunks swizzled like the bf16 image: slot = chunk ^ ((cout >> 2) & 3))
// with one POWER-OF-TWO scale per cout: stored = rn_e4m3(w / wscale[c]), wscale[c] = 2^ceil(log2(max|w[c]| / 448)) (1 for an all-zero row): the division is exact,
// every weight is rounded once
void vgh_pack_conv_weights_fp8_host(const float* w, int cout_pad, int ksize, int cin, uint8_t* dst, float* wscale) {
    const int cblocks = cin / 64, taps = ksize * ksize;
    const size_t row = (size_t)taps * cin;
    for (int co = 0; co < cout_pad; ++co) {
        float mx = 0.0f;
        for (size_t i = 0; i < row; ++i) {
            const float v = fabsf(w[(size_t)co * row + i]);
            if (v > mx) mx = v;  // (a NaN weight never compares greater: it is converted to the NaN code below)
        }
        float sc = 1.0f;
        if (mx > 0.0f && mx <= 3.0e38f) {
            int e;
            frexpf(mx / 448.0f, &e);  // mx / 448 = m * 2^e, m in [0.5, 1): 2^e >= mx / 448
            sc = ldexpf(1.0f, e);
            if (mx / 448.0f == ldexpf(1.0f, e - 1)) sc = ldexpf(1.0f, e - 1);  // an exact power of two
        }
        wscale[co] = sc;
    }
    for (int tap = 0; tap < taps; ++tap)
        for (int cb = 0; cb < cblocks; ++cb) {
            const int kb = tap * cblocks + cb;
            for (int co = 0; co < cout_pad; ++co) {
                const float* src = w + ((size_t)co * taps + tap) * cin + cb * 64;
                uint8_t* d = dst + ((size_t)kb * cout_pad + co) * 64;
                const int sw = (co >> 2) & 3;
                const float inv = 1.0f / wscale[co];
                for (int chunk = 0; chunk < 4; ++chunk)
                    for (int e = 0; e < 16; ++e) d[(chunk ^ sw) * 16 + e] = vgh_f32_to_e4m3_host(src[chunk * 16 + e] * inv);
            }
        }
}

// dense [cout_pad][ks][ks][cin] f32 -> int8 image, same layout as the e4m3 one; one scale per cout: wscale[c] = max|w[c]| / 127 (1 for an all-zero or non-finite row),
// stored = clamp(rn(w / wscale[c]), -127, 127) (-128 is never produced: the grid is symmetric)
void vgh_pack_conv_weights_i8_host(const float* w, int cout_pad, int ksize, int cin, uint8_t* dst, float* wscale) {
    const int cblocks = cin / 64, taps = ksize * ksize;
    const size_t row = (size_t)taps * cin;
    for (int co = 0; co < cout_pad; ++co) {
        float mx = 0.0f;
        for (size_t i = 0; i < row; ++i) {
            const float v = fabsf(w[(size_t)co * row + i]);
            if (v > mx) mx = v;
        }
        wscale[co] = (mx > 0.0f && mx <= 3.0e38f) ? mx / 127.0f : 1.0f;
    }
    for (int tap = 0; tap < taps; ++tap)
        for (int cb = 0; cb < cblocks; ++cb) {
            const int kb = tap * cblocks + cb;
            for (int co = 0; co < cout_pad; ++co) {
                const float* src = w + ((size_t)co * taps + tap) * cin + cb * 64;
                uint8_t* d = dst + ((size_t)kb * cout_pad + co) * 64;
                const int sw = (co >> 2) & 3;
                for (int chunk = 0; chunk < 4; ++chunk)
                    for (int e = 0; e < 16; ++e) {
                        float q = __builtin_nearbyintf(src[chunk * 16 + e] / wscale[co]);
                        q = q != q ? 0.0f : q > 127.0f ? 127.0f : q < -127.0f ? -127.0f : q;
                        d[(chunk ^ sw) * 16 + e] = (uint8_t)(int8_t)(int)q;
                    }
            }
        }
}
