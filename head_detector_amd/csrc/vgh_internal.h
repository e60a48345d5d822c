// Internal declarations shared by the libvgh.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/vgh.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

// ---- error plumbing: never throw across the C ABI --------------------------------------
void vgh_set_error(const char* fmt, ...);

#define VGH_HIP(expr)                                                                       \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            vgh_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return VGH_ERR_HIP;                                                             \
        }                                                                                   \
    } while (0)

#define VGH_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            vgh_set_error(__VA_ARGS__);   \
            return VGH_ERR_INVALID;       \
        }                                 \
    } while (0)

// ---- conv launch descriptor (device pointers resolved) ---------------------------------
struct ConvArgs {
    const uint16_t* in;    // bf16 NHWC, pixel pitch in_pitch elements, first channel in_coff
    const uint16_t* wpack; // packed weights [nkb][cout_pad][32] bf16, 16B chunks pre-swizzled
    const float* bias;     // [cout_pad]
    void* out;             // bf16 (or f32 when out_f32) NHWC
    const uint16_t* res;   // residual, same spatial dims as out (or nullptr)
    const uint16_t* zeros; // >= 64 B of zeros (im2col padding source)
    int64_t in_pitch, out_pitch, res_pitch;
    int in_coff, out_coff, out_coff2, out_split, res_coff;
    int B, H, W, Ho, Wo;
    int cin;        // multiple of 32
    int cout_pad;   // rows in wpack (multiple of 32)
    int cout_store; // channels actually written (<= cout_pad)
    int ksize, stride, pad;
    int act;        // VGH_ACT_*
    int out_f32;    // 1: write float
    int shuffle;    // 1: ConvTranspose2d(k=2,s=2) pixel-shuffle store, cout_pad = 4*C
    int shuffle_c;  // C of the transposed conv
    float alpha;    // residual scale
    int P;          // B*Ho*Wo output pixels
    int nkb;        // ksize*ksize*cin/32
    int cblocks;    // cin/32
    unsigned div_howo_m, div_howo_s, div_wo_m, div_wo_s;  // n / d == (umulhi(n, m) + n) >> s for n < 2^30 (filled by vgh_launch_conv)
    int fast_epi;   // 1: LDS-transposed 16-byte epilogue (bf16 out, 8-channel aligned offsets)
    int stagger;    // persistent patch kernels: odd resident-slot blocks start this many ~0.5 us sleeps late (de-phases co-resident blocks)
    int grid_share; // persistent patch kernels: take 1/grid_share of the CU slots (the executor's other lane streams own the rest)
    // ---- grouped convolution (block-diagonal launch of sibling branches): cout tile c0 belongs to group c0 / grp_cout and reads its
    //      input channels at in_coff + group * grp_in_stride; every row of wpack holds that group's cin = a.cin channels ----
    int grp_cout, grp_in_stride;  // 0: dense
    // ---- split-precision parity modes (conv_split.hip): activations are two 16-bit planes per pixel [hi C | lo C] and the K loop runs
    //      three segments x_hi*w_lo, x_lo*w_hi (corrections), then -- after one multiply of the accumulators by acc_scale -- x_hi*w_hi ----
    int split;        // VGH_FMT_BF16 (0): plain; VGH_FMT_BF16X2 / VGH_FMT_F16X2: hi|lo planes
    int in_plane, out_plane, res_plane;  // elements from a pixel's hi plane to its lo plane (the buffer's logical pitch)
    int seg_kb;       // k-blocks per segment = ksize^2 * cblocks (nkb = nseg * seg_kb)
    int nseg;         // 3: the split modes' x_hi*w_lo, x_lo*w_hi, x_hi*w_hi; 1: the single-plane fp16 format (VGH_FMT_F16, r05) rides the F16X2 kernels with its one
                      //    plane as "hi" (split = VGH_FMT_F16X2, in/out/res_plane = 0, acc_scale = 0: no lo plane is read or written) and the [w_hi] weight image
    float acc_scale;  // 1/2048 (fp16: lo planes are stored scaled by 2^11) or 1 (bf16)
    float lo_scale;   // lo = (v - hi) * lo_scale
    float out_scale;  // undoes the per-op power-of-two weight prescale (fp16), applied to the accumulator before the bias
    // ---- e4m3 links (conv_pp.hip, r05): the input view / the output tensor hold OCP e4m3 bytes (pitches and channel offsets then count bytes); the residual stays
    //      bf16.  gscale[cout_pad]: per-cout output factor applied to the accumulator (which starts at bias / gscale when the input is e4m3) ----
    //      in_fp8 / out_fp8: 0 = 16-bit, 1 = e4m3, 2 = int8 (VGH_FMT_I8: the `bias` vector of an int8-INPUT conv holds int32 BIT PATTERNS in accumulator units,
    //      rn(bias[c] / (wscale[c] * scale(in))); the int32 accumulator starts there and out = act(float(acc) * gscale[c]) -- include/vgh.h, conv_pp.hip PP_INIT_ROWS) ----
    int in_fp8, out_fp8;
    const float* gscale;
    const float* dvec;  // int8 -> bf16 only (or nullptr): the diagonal bypass, out[c] += dvec[c] * code(input pixel, channel c) before the activation (conv_pp.hip DG)
    float bias_scale;  // fp16 ping-pong variant: 1 / out_scale (set by vgh_launch_conv_pp)
    int fallback_cfg1;  // 0: a forced tile that cannot run this conv is an error; k + 1: it falls back to tile k (network executor: a table may be stale)
    int nt_out;          // bf16 output stores carry the non-temporal hint (set by vgh_conv_prepare from vgh_conv_set_nt_store)
    // ---- back-to-back GEMM (r06, conv_kernels.inc T2 > 0): this conv's output tile feeds a 1x1 / stride-1 conv inside the same launch; only that conv's output is stored ----
    const uint16_t* w2pack;  // the second conv's packed weights [cout_pad / 32][cout2_pad][32] (vgh_pack_conv_weights_host), or nullptr
    const float* bias2;      // [cout2_pad]
    void* out2;              // the second conv's output tensor (bf16 NHWC), pitch / offsets / split / stored channels as for `out`
    int64_t out2_pitch;
    int out2_coff, out2_coff2, out2_split, cout2_pad, cout2_store, act2;
    int b2b_igemm;           // 1: keep the pair on the implicit-GEMM b2b tile even where the persistent t tile (ds_b2b.hip) could run it (vgh_net_set_b2b(n, 2): A/B, tests)
    int ablate;     // -DVGH_EXPERIMENTS builds only: bit0 skip tile loads, bit1 skip MFMAs, bit3 skip the epilogue (results are garbage)
    unsigned long long* trace;  // -DVGH_EXPERIMENTS builds only: per-(block, tile) phase timestamps (s_memtime), or nullptr
};
#define VGH_TRACE_TILES 16  // tiles recorded per block
#define VGH_TRACE_MARKS 4   // marks per tile: tile top, operands landed, K loop done, epilogue done

// Work-skipping switches exist only in the experiments build (tools/, never the shipped libvgh.so)
#ifdef VGH_EXPERIMENTS
#define VGH_ABLATE(a, bit) ((a).ablate & (bit))
#define VGH_MARK(a, tile_no, k)                                                                                                      \
    do {                                                                                                                             \
        if ((a).trace && threadIdx.x == 0 && (tile_no) < VGH_TRACE_TILES)                                                            \
            (a).trace[((size_t)blockIdx.x * VGH_TRACE_TILES + (tile_no)) * VGH_TRACE_MARKS + (k)] = __builtin_amdgcn_s_memtime();   \
    } while (0)
// one tile per block (the implicit-GEMM kernel): row = blockIdx.x, tile 0; the tool's buffer holds 8192 rows
#define VGH_MARK_BLOCK(a, k)                                                                                                  \
    do {                                                                                                                      \
        if ((a).trace && threadIdx.x == 0 && blockIdx.x < 8192)                                                               \
            (a).trace[((size_t)blockIdx.x * VGH_TRACE_TILES) * VGH_TRACE_MARKS + (k)] = __builtin_amdgcn_s_memtime();        \
    } while (0)
struct VghMarkAtExit {  // "epilogue done" on every return path of a kernel body
    unsigned long long* t;
    __device__ ~VghMarkAtExit() {
        if (t && threadIdx.x == 0 && blockIdx.x < 8192) t[((size_t)blockIdx.x * VGH_TRACE_TILES) * VGH_TRACE_MARKS + 3] = __builtin_amdgcn_s_memtime();
    }
};
#define VGH_MARK_EXIT(a) VghMarkAtExit vgh_mark_at_exit_{(a).trace}
#else
#define VGH_ABLATE(a, bit) 0
#define VGH_MARK(a, tile_no, k) \
    do {                        \
    } while (0)
#define VGH_MARK_BLOCK(a, k) \
    do {                     \
    } while (0)
#define VGH_MARK_EXIT(a) \
    do {                 \
    } while (0)
#endif

// postproc.hip (r06): vgh_nms + vgh_compact + the detector's head list as ONE launch (block per image; the last block to finish builds the head list; head_row_dev may be
// null: no head list).  ticket_dev: one zero-initialised int32 the kernel leaves at zero.
int vgh_nms_select(const float* boxes_dev, const float* scores_dev, const float* flame_dev, int B, int n_in, float conf_thr, float iou_thr, int keep_k,
                   int32_t* keep_idx_dev, int32_t* counts_dev, float* out_boxes_dev, float* out_scores_dev, float* out_flame_dev, int capacity, int32_t* head_row_dev,
                   int32_t* head_image_dev, int32_t* n_heads_dev, int32_t* ticket_dev, const vgh_head_level* lazy_levels, int n_levels, const int32_t* lazy_idx_dev,
                   int shape_c, int expr_c, void* stream);  // lazy_idx_dev != null: flame_dev is not read, the survivors' vectors come from the prediction buffers
int vgh_launch_conv(const ConvArgs& a, int force_cfg, hipStream_t stream);
// conv_igemm.hip: `a` with its b2b fields set (w2pack ...): the fused launch, or VGH_ERR_INVALID when the pair does not fit a b2b tile (vgh_conv_b2b_ok says so beforehand)
int vgh_launch_conv_b2b(const ConvArgs& a, hipStream_t stream);
int vgh_conv_b2b_ok(int ksize, int stride, int cout_pad, int cout2_pad);
// ds_b2b.hip ("t" tile): the stage-1 downsample + conv1|conv2 pair as one persistent launch with register-resident weights; `a` prepared, b2b fields set
int vgh_conv_ds_b2b_ok(const ConvArgs& a);
int vgh_launch_conv_ds_b2b(const ConvArgs& a, hipStream_t stream);
// ("r" tile) a plain 3x3 / stride-2 conv with 96 input channels in the same execution structure; `a` prepared
int vgh_conv_ds_ok(const ConvArgs& a);
int vgh_launch_conv_ds(const ConvArgs& a, hipStream_t stream);
// ("w" tile) 3x3 / stride-1 convs with 96 / 128 input channels, weights resident in registers; `a` prepared
int vgh_conv_w_ok(const ConvArgs& a);
int vgh_launch_conv_w(const ConvArgs& a, hipStream_t stream);
// ("u" tile) the same with the stem conv in the launch: u8 NHWC images in, the stem tensor never exists; `a` = the pair (b2b fields set, not yet prepared)
int vgh_launch_stem_ds_b2b(const ConvArgs& a, const void* image_u8, int Hi, int Wi, const float* wstem, const float* bstem, hipStream_t stream);
int vgh_launch_conv_split(const ConvArgs& a, int force_cfg, hipStream_t stream);  // conv_split.hip: a.split = VGH_FMT_BF16X2 / VGH_FMT_F16X2
// dense [cout_pad][ks][ks][cin] f32 -> the three-segment 16-bit image [w_lo | w_hi | w_hi] (each segment laid out like
// vgh_pack_conv_weights_host); fmt = VGH_FMT_BF16X2 / VGH_FMT_F16X2.  Returns through *out_scale the factor the kernel multiplies the
// accumulators by (fp16: 2^-s with 2^s the power-of-two prescale that brings max|w| to ~2^9; bf16: 1).
void vgh_pack_conv_weights_split_host(const float* w, int cout_pad, int ksize, int cin, int fmt, uint16_t* dst, float* out_scale);
static inline int vgh_fmt_bytes(int fmt) { return (fmt == VGH_FMT_FP8 || fmt == VGH_FMT_I8) ? 1 : (fmt == 0 || fmt == VGH_FMT_F16) ? 2 : 4; }  // bytes per logical element of an activation buffer
static inline int vgh_fmt_planes(int fmt) { return (fmt == VGH_FMT_BF16X2 || fmt == VGH_FMT_F16X2) ? 2 : 1; }
// conv_pp.hip (host): OCP e4m3fn conversion and the e4m3 weight image of the ping-pong tiles
uint8_t vgh_f32_to_e4m3_host(float f);
void vgh_pack_conv_weights_fp8_host(const float* w, int cout_pad, int ksize, int cin, uint8_t* dst, float* wscale);
void vgh_pack_conv_weights_i8_host(const float* w, int cout_pad, int ksize, int cin, uint8_t* dst, float* wscale);
static inline int vgh_fmt_is_q8(int fmt) { return fmt == VGH_FMT_FP8 || fmt == VGH_FMT_I8; }    // an 8-bit link format
static inline int vgh_fmt_q8_kind(int fmt) { return fmt == VGH_FMT_FP8 ? 1 : fmt == VGH_FMT_I8 ? 2 : 0; }  // ConvArgs::in_fp8 / out_fp8 code
// conv_pp.hip: the 8-wave ping-pong 3x3 / stride-1 tiles ("g" tiles; bc = 128 / 96 / 64 couts per workgroup); `a` must be prepared
int vgh_launch_conv_pp(const ConvArgs& a, int bc, int version /* 1: "g" (two barriers per tap), 2: "h" (one), 3: "s" (g with two 4 x 8 sub-patches per wave) */, int max_blocks_per_xcd, hipStream_t stream);
int vgh_conv_pp_lds(int bc);
int vgh_conv_max_blocks_per_xcd();  // the process-wide cap of vgh_conv_set_max_blocks_per_xcd (0 = none)
int vgh_conv_pp_fits(const ConvArgs& a);
int vgh_conv_persistent_blocks_per_xcd(const ConvArgs& a, int chunk, int blocks_per_cu);
int vgh_conv_pick_cfg(const ConvArgs& a);
int vgh_conv_cfg_ok_for(int cfg, const ConvArgs& a);  // `a` prepared; bf16 tiles only (split launches validate in conv_split.hip)
// validates `a` and fills its derived fields (fast-division constants, fast_epi); vgh_launch_conv calls it itself
int vgh_conv_prepare(ConvArgs& a);
// automatic tile of a PREPARED descriptor: index into the bf16 table (a.split == 0) or into conv_split.hip's table
int vgh_conv_pick_auto(const ConvArgs& a);
int vgh_conv_split_pick(const ConvArgs& a);
// host-side weight packing: dense [cout_pad][ks][ks][cin] f32 -> wpack bf16 image
void vgh_pack_conv_weights_host(const float* w, int cout_pad, int ksize, int cin, uint16_t* dst);
static inline size_t vgh_wpack_elems(int cout_pad, int ksize, int cin) { return (size_t)cout_pad * ksize * ksize * cin; }

static inline void vgh_fastdiv_magic(unsigned d, unsigned* m, unsigned* s) {
    unsigned sh = 0;
    while ((1ull << sh) < d) ++sh;
    *m = (unsigned)((((1ull << 32) * ((1ull << sh) - d)) / d) + 1);
    *s = sh;
}

static inline uint16_t vgh_f32_to_bf16_host(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40); // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// ---- other launchers --------------------------------------------------------------------
// fmt / plane: VGH_FMT_BF16 (plane unused) or a split format: then out_pitch is the physical pitch and `plane` the elements from hi to lo
int vgh_launch_stem(const void* image, int image_fmt, int B, int H, int W, const float* w /*[64][27] dev*/,
                    const float* bias /*[64] dev*/, uint16_t* out, int64_t out_pitch, int out_coff, int store_ch /*64, or 48 (bf16 only)*/, int fmt, int plane,
                    hipStream_t stream);
// stem_ds.hip: stem + stage-1 downsample in one kernel (bit-identical to vgh_launch_stem + the 3x3 / stride-2 / 64 -> 96 implicit-GEMM conv)
int vgh_launch_stem_ds(const void* image, int image_fmt, int B, int H, int W, const float* wstem, const float* bstem, const uint16_t* wds, const float* bds, uint16_t* out,
                       int64_t out_pitch, int out_coff, hipStream_t stream);
void vgh_pack_stem_ds_weights_host(const float* w /*[96][3][3][64]*/, uint16_t* dst /*[9][3][96][16]*/);
int vgh_launch_conv_f32(const ConvArgs& a, const float* wdense, hipStream_t stream);
int vgh_launch_stem_f32(const void* image, int image_fmt, int B, int H, int W, const float* w, const float* bias, float* out, int64_t out_pitch, int out_coff,
                        hipStream_t stream);
int vgh_launch_spp_pool_f32(float* buf, int64_t pitch, int coff, int C, int B, int H, int W, hipStream_t stream);
int vgh_launch_spp_pool(uint16_t* buf, int64_t pitch, int coff, int C, int B, int H, int W, int fmt, int plane, hipStream_t stream);

// streams.hip: a stream MEASURED to run side by side with every stream in `avoid` (earlier entries matter more when the hardware
// queues do not suffice for all of them); released streams are parked, never destroyed
int vgh_stream_acquire_internal(int device, const hipStream_t* avoid, int n_avoid, hipStream_t* out, bool low_priority = false);
void vgh_stream_release_internal(int device, hipStream_t s, bool low_priority = false);
struct vgh_net;
// net.hip: the net's lane streams for work entering on `main` (picked on first use, re-picked when `main` changes); out[3]
int vgh_net_lane_streams(vgh_net* n, hipStream_t main, hipStream_t* out);
extern "C" __attribute__((visibility("hidden"))) int vgh_net_device(vgh_net* n);  // library-internal
// net.hip, library-internal since r06 (was exported): the first op of the next forwards that writes a prediction buffer waits for `event` (or nullptr) on its stream
extern "C" __attribute__((visibility("hidden"))) int vgh_net_set_pred_guard(vgh_net* n, void* event);
