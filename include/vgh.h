/* libvgh.so -- C ABI of the MI355X-native VGGHeads forward path (gfx950 only).
 *
 * The reference (KupynOrest/head_detector) is pure Python: its hot path is three pluggable
 * callables inside HeadDetector (head_detector/detector.py:58-59,92-95).  Each entry point below
 * names the reference interface it replaces.  Conventions:
 *   - return 0 (VGH_OK) or a negative VGH_ERR_*; nothing is thrown across the ABI;
 *     vgh_last_error() returns a thread-local message for the last failure;
 *   - every `*_dev` pointer is caller-owned DEVICE memory (e.g. a torch-ROCm tensor's data_ptr());
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls are asynchronous
 *     with respect to the host unless stated otherwise;
 *   - handles (vgh_net, vgh_flame) are NOT thread-safe: one handle per (device, stream);
 *   - no hidden device allocation after *_create (arenas are planned up front).
 */
#ifndef VGH_H_
#define VGH_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VGH_OK 0
#define VGH_ERR_INVALID (-1)
#define VGH_ERR_HIP (-2)
#define VGH_ERR_NOMEM (-3)

#define VGH_ACT_NONE 0
#define VGH_ACT_RELU 1 /* every backbone/neck/head block: activation_type: relu (arch yaml :15..86) */
#define VGH_ACT_SILU 2

#define VGH_IMG_F32_NCHW 0 /* what HeadDetector._transform_image hands the model (detector.py:51) */
#define VGH_IMG_U8_NHWC 1  /* raw letterboxed image before .permute().float()/255 (detector.py:48-51) */

#define VGH_NUM_FLAME_PARAMS 413 /* FLAME_CONSTS, head_detector/head_info.py:12-21 */

const char* vgh_version(void);
const char* vgh_last_error(void);

/* ABI revision of this header: bumped whenever a struct below grows or a function changes meaning (r03 -> 3: vgh_conv_call / vgh_op_desc gained
 * grp_cout, grp_in_stride, fmt, out_scale and vgh_flame_set_matrix_path became a 0..4 mode; r04 -> 4: this call; -> 5: modes 0..7 of
 * vgh_flame_set_matrix_path; r05 -> 6: VGH_FMT_FP8, vgh_buf_desc.scale, vgh_conv_call.out_fp8 / gscale_dev, vgh_pack_conv_weights_fp8;
 * -> 7: VGH_FMT_I8, vgh_conv_call.out_fp8 = 2 / diag_dev, vgh_pack_conv_weights_i8, vgh_net_set_i8_diag).  A client built against another
 * revision passes structs of another size: compare before the first call that takes one (head_detector_amd/_lib.py and tests/c_abi_smoke.c do). */
#define VGH_ABI_VERSION 7
int vgh_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Network: replaces `self.model(image)` -- the TorchScript blob called at detector.py:58-59 whose
 * graph is YoloHeads.forward (yolo_heads.py:89-112, arch yaml :4-137) + YoloHeadsNDFLHeads.forward
 * (yolo_head_ndfl_heads.py:117-175) + VGGHeadDecodingModule (yolo_heads.py:44-86).
 * The host describes the folded network as a flat op program over NHWC bf16 buffers.
 * ---------------------------------------------------------------------------------------------- */
#define VGH_OP_STEM 0     /* 3x3 s2 conv on the raw image, fused /255 + bias + ReLU (YoloNASStem)      */
#define VGH_OP_CONV 1     /* implicit-GEMM conv k in {1,3}, stride in {1,2}, fused bias/act/residual    */
#define VGH_OP_SPP_POOL 2 /* SPP max-pool 5/9/13 written next to its input (concat-by-offset)          */
#define VGH_OP_FORK 3     /* fork point: ops with lane > 0 issued after it run on side HIP streams that wait for */
                          /* everything enqueued on the main stream BEFORE this point; all lanes join at the end */

/* activation storage formats (vgh_buf_desc.is_f32 -- the field keeps its historical name) */
#define VGH_FMT_BF16 0   /* throughput mode: bf16 NHWC */
#define VGH_FMT_F32 1    /* fp32 NHWC: head prediction outputs in every mode; every buffer in the fp32 (VALU) parity mode */
#define VGH_FMT_BF16X2 2 /* split parity mode: per pixel [hi C | lo C] bf16, value = hi + lo (16 significand bits) */
#define VGH_FMT_F16X2 3  /* split parity mode: per pixel [hi C | lo C] fp16, value = hi + lo / 2048 (22 significand bits) */
#define VGH_FMT_F16 5    /* r05, "fp16" throughput mode: ONE fp16 plane per value (the reference's own FP16 export: exportable_mesh_model.py:177,299,409); same bytes and MFMA */
                         /*   count as bf16, 11 instead of 8 significand bits; weights carry a per-op power-of-two prescale (undone on the accumulator), values beyond       */
                         /*   +-65504 saturate                                                                                                                              */
#define VGH_FMT_FP8 4    /* r05, "fp8" throughput mode: OCP e4m3fn bytes, value = stored * vgh_buf_desc.scale.  Only on links between two 3x3 / stride-1  */
                         /*   convs that run on the ping-pong tiles (csrc/conv_pp.hip): written by one, read by one; every other tensor stays bf16     */
#define VGH_FMT_I8 6     /* r05, "int8" throughput mode (the reference exporter's QuantizationMode.INT8: exportable_mesh_model.py:175-178,398-411): signed bytes */
                         /*   in [-127, 127], value = stored * vgh_buf_desc.scale, on the same links as VGH_FMT_FP8; v_mfma_i32_32x32x32_i8, exact int32 accumulation */

typedef struct vgh_buf_desc {
    int32_t h, w;   /* spatial size per image */
    int32_t pitch;  /* LOGICAL channels per pixel (concat width); 16-bit formats: multiple of 8; the two-plane formats occupy 2*pitch */
    int32_t is_f32; /* VGH_FMT_* */
    float scale;    /* VGH_FMT_FP8 / VGH_FMT_I8: value = stored * scale (> 0; from a calibration forward: max|activation| / 448 resp. / 127 with head-room); other formats: ignored */
} vgh_buf_desc;

typedef struct vgh_op_desc {
    int32_t kind;
    int32_t in_buf, in_coff, cin;        /* cin padded to a multiple of 32 (zero channels).  in_coff + cin may exceed the buffer's pitch (bf16 buffers,   */
                                         /*   dense convs): the window then runs on into the next pixel, and vgh_net_create REQUIRES every weight of the   */
                                         /*   input channels past the pitch to be exactly zero (the 48-channel stem tensor read as two 32-channel K blocks) */
    int32_t out_buf, out_coff, cout_pad; /* rows of the weight matrix, multiple of 32                  */
    int32_t cout_store;                  /* channels written; [cout_real, cout_store) are exact zeros (VGH_OP_STEM: 64, or 48 = no padding, bf16 only) */
    int32_t out_split, out_coff2;        /* channels >= out_split go to out_coff2 + (c - out_split)    */
    int32_t res_buf, res_coff;           /* residual added AFTER the activation (YoloNASBottleneck),   */
    float alpha;                         /*   out = act(conv + b) + alpha * res;  res_buf < 0: none    */
    int32_t ksize, stride, act;
    int32_t shuffle;                     /* 1: ConvTranspose2d(k=2,s=2) as 4 pointwise GEMMs + shuffle */
    int64_t w_off;                       /* offset (floats) of [cout_pad][k][k][cin] in `weights`      */
    int64_t b_off;                       /* offset (floats) of [cout_pad] in `biases`                  */
    int32_t force_cfg;                   /* -1: heuristic tile choice; else kernel config index        */
    int32_t lane;                        /* bits 0-7: 0 = main stream, 1..3 = side stream (independent branch, see VGH_OP_FORK); bits 8-31 (r06, additive): 0, or 1 + the index of */
                                         /*   the ONE earlier op on another lane this op waits for (an event recorded behind that op) -- arch.schedule_latency                      */
    int32_t grp_cout, grp_in_stride;     /* grouped conv (sibling branches in one launch): output channels [g*grp_cout, (g+1)*grp_cout) read  */
                                         /*   the `cin` input channels starting at in_coff + g*grp_in_stride; 0: dense                      */
} vgh_op_desc;

typedef struct vgh_net vgh_net;

int vgh_net_create(int device, int image_size, int max_batch, const vgh_buf_desc* bufs, int n_bufs, const vgh_op_desc* ops, int n_ops,
                   const float* weights_host, int64_t n_weights, const float* biases_host, int64_t n_biases, vgh_net** out);
void vgh_net_destroy(vgh_net* net);
/* image_dev: [B,3,S,S] f32 (VGH_IMG_F32_NCHW) or [B,S,S,3] u8 (VGH_IMG_U8_NHWC). Runs every op. */
int vgh_net_forward(vgh_net* net, const void* image_dev, int image_fmt, int B, void* stream);
/* As vgh_net_forward but host-synchronous, bracketing every op with HIP events on `stream`;
 * op_ms[n_ops] receives each op's device time in milliseconds (tuning / roofline reporting). */
int vgh_net_profile(vgh_net* net, const void* image_dev, int image_fmt, int B, void* stream, float* op_ms);
/* Capture vgh_net_forward into a hipGraph (replayed by vgh_net_forward_graph with the same B / image_dev).  VGH_ERR_INVALID for a program in which two SIDE
 * lanes wait for each other's ops (vgh_op_desc.lane bits 8+): the HIP runtime cannot end such a capture; waits on the main stream's ops are always fine. */
int vgh_net_capture(vgh_net* net, const void* image_dev, int image_fmt, int B, void* stream);
int vgh_net_forward_graph(vgh_net* net, void* stream);
void* vgh_net_buffer(vgh_net* net, int buf_id);         /* device pointer of an activation buffer   */
int64_t vgh_net_buffer_bytes(vgh_net* net, int buf_id); /* bytes for max_batch                       */
int vgh_net_set_cfg(vgh_net* net, int op_index, int cfg);
/* Batch split (1..4, default 1): vgh_net_forward runs the batch as `nsplit` independent sub-batches on net-owned lane streams
 * (forked from / joined into `stream`), so that the fixed cost of every launch -- dispatch, tile prologue and first-load
 * latency, epilogue store burst, tail -- of one sub-batch hides under the main loops of the others.  Results are identical
 * (images are independent; every op keeps its tile configuration). */
int vgh_net_set_split(vgh_net* net, int nsplit);
/* r06: BACK-TO-BACK GEMM.  A conv whose whole output (all channels in one 96-cout tile) is read by exactly one op -- the 1x1 conv behind it: a backbone stage's downsample
 * and the CSP layer's merged conv1|conv2 -- runs with that op as ONE launch: the first conv's accumulators become, in registers, the B operands of the second GEMM, and the
 * tensor between them is never written (csrc/conv_kernels.inc, T2 > 0).  vgh_net_create finds the pairs; they run fused by default.  enable = 0: the two launches (the same
 * output bits when the second conv runs on an implicit-GEMM tile -- the fused kernel's arithmetic; a tuning table that gives it a streaming "t" tile, whose accumulators
 * start at the bias, differs in the last fp32 rounding before the bf16 store; the intermediate tensor then exists in the arena, as the per-op parity tests need).  vgh_net_b2b_pairs: how many pairs the program has.
 * Late r06 (additive): with enable = 1 the stage-1 pair (3x3 / stride 2, 48 -> 96 channels + its 1x1) runs on a persistent tile of its own (csrc/ds_b2b.hip, "t" tile: input
 * patch fetched once into parity planes, both convs' weights resident in registers; the same output bits), and for VGH_IMG_U8_NHWC images the STEM conv runs inside that launch
 * too ("u" tile: a u8 pixel is an exact bf16, the weights / 255 are split into three bf16 values -- exact products, fp32 accumulation in another order than the stem kernel's
 * fmaf chain: a flipped bf16 ulp in ~4e-5 of the stem values; the stem tensor is not written).  enable = 3: the t tile fed by the stem launch (bit-identical to enable = 2);
 * enable = 2: fused, but every pair on the implicit-GEMM tile (A/B, tests). */
int vgh_net_set_b2b(vgh_net* net, int enable);
/* 1 when forwards of VGH_IMG_U8_NHWC images run the stem conv inside the stage-1 pair's launch (mode 1 and an eligible program): the stem tensor is then not written. */
int vgh_net_stem_fused(vgh_net* net);
int vgh_net_b2b_pairs(vgh_net* net);
int vgh_net_max_batch(vgh_net* net);  /* images the activation arena was planned for */
int vgh_net_image_size(vgh_net* net);

/* Stand-alone conv launch on caller-owned device tensors (per-layer parity tests, micro-benchmarks).
 * `wpack_dev` must come from vgh_pack_conv_weights. */
typedef struct vgh_conv_call {
    const void* in_dev;
    int64_t in_pitch;
    int32_t in_coff, cin;
    int32_t B, H, W;
    const void* wpack_dev;
    const float* bias_dev;
    void* out_dev;
    int64_t out_pitch;
    int32_t out_coff, cout_pad, cout_store, out_split, out_coff2, out_f32;
    const void* res_dev;
    int64_t res_pitch;
    int32_t res_coff;
    float alpha;
    int32_t ksize, stride, act, shuffle;
    int32_t force_cfg;
    int32_t grp_cout, grp_in_stride;     /* grouped conv, see vgh_op_desc (0: dense) */
    int32_t fmt;                         /* VGH_FMT_BF16 (0) or a split format: then in/out/res pitches are the LOGICAL pitches, the lo planes   */
                                         /*   sit `pitch` elements behind the hi planes, wpack_dev comes from vgh_pack_conv_weights_split        */
    float out_scale;                     /*   and out_scale is what that call returned                                                           */
                                         /* fmt = VGH_FMT_FP8 (3x3 / stride 1, ping-pong tiles only): in_dev holds e4m3 bytes (in_pitch / in_coff count   */
                                         /*   bytes, cin % 64 == 0), wpack_dev comes from vgh_pack_conv_weights_fp8, res_dev stays bf16, and bias_dev     */
                                         /*   holds bias[c] / gscale[c] (the accumulator starts there and is multiplied by gscale[c] at the end)          */
                                         /* fmt = VGH_FMT_I8: the same with int8 bytes and vgh_pack_conv_weights_i8, except that the accumulator is an      */
                                         /*   exact int32 sum: bias_dev holds INT32 values (bit patterns in the float array), rn(bias[c] / (wscale[c] *      */
                                         /*   input scale)); out = act(float(acc) * gscale[c])                                                               */
    int32_t out_fp8;                     /* 1: out_dev receives e4m3 bytes, 2: int8 bytes (out_pitch / offsets count bytes; whole cout tiles, no residual; */
                                         /*   an 8-bit input and an 8-bit output must be the same format)                                                   */
    const float* gscale_dev;             /* [cout_pad] per-cout output factor: e4m3 in: wscale[c] * input scale (/ output scale when out_fp8);           */
                                         /*   bf16 in, e4m3 out: 1 / output scale.  NULL otherwise                                                       */
    const float* diag_dev;               /* fmt = VGH_FMT_I8, bf16 output, cout_pad <= min(cin, 1024) (or NULL): the diagonal bypass -- before the         */
                                         /*   activation, out[c] += diag_dev[c] * code(input pixel, channel c): the caller has taken w[c][centre][c]         */
                                         /*   (the folded identity branch of a RepVGG block) out of the int8 image, diag_dev[c] = that weight * input scale  */
} vgh_conv_call;
int vgh_conv2d(const vgh_conv_call* c, void* stream);
/* Split-precision weight image (parity modes): dense [cout_pad][k][k][cin] f32 -> 3*cout_pad*k*k*cin u16 ([w_lo | w_hi | w_hi] segments);
 * *out_scale receives the accumulator scale of the op (see vgh_conv_call.out_scale). */
int vgh_pack_conv_weights_split(const float* w_host, int cout_pad, int ksize, int cin, int fmt, uint16_t* wpack_host, float* out_scale);
/* e4m3 weight image of the ping-pong tiles: dense [cout_pad][k][k][cin] f32 -> cout_pad*k*k*cin bytes + one power-of-two scale per cout
 * (stored = rn_e4m3(w / wscale[c])); cin % 64 == 0, k = 3. */
int vgh_pack_conv_weights_fp8(const float* w_host, int cout_pad, int ksize, int cin, uint8_t* wpack_host, float* wscale_host);
/* int8 weight image of the ping-pong tiles (same layout): stored = clamp(rn(w / wscale[c]), -127, 127) with wscale[c] = max|w[c]| / 127 (1 for an all-zero row) */
int vgh_pack_conv_weights_i8(const float* w_host, int cout_pad, int ksize, int cin, uint8_t* wpack_host, float* wscale_host);
/* dense [cout_pad][k][k][cin] f32 (host) -> kernel-private bf16 image (host, cout_pad*k*k*cin u16) */
int vgh_pack_conv_weights(const float* w_host, int cout_pad, int ksize, int cin, uint16_t* wpack_host);
int vgh_conv_num_cfgs(void);
const char* vgh_conv_cfg_name(int cfg);
int vgh_conv_cfg_cout_tile(int cfg); /* output channels per tile: a grouped conv (vgh_op_desc.grp_cout) needs grp_cout % tile == 0 */
/* 1 if tile configuration `cfg` can run a conv of this kind (tuning tools). */
int vgh_conv_cfg_ok(int cfg, int ksize, int stride, int cout_pad, int fast_epilogue, int shuffle);
/* The tile set of the split-precision parity modes (csrc/conv_split.hip) is a table of its own: vgh_net_set_cfg indexes it for a net whose
 * activation buffers are VGH_FMT_BF16X2 / VGH_FMT_F16X2. */
int vgh_conv_split_num_cfgs(void);
const char* vgh_conv_split_cfg_name(int cfg);
/* (The last three entries, g8x8x{128,96,64}_n8, are the fp16 ping-pong tiles: the answer is for a SINGLE-PLANE fp16 net (VGH_FMT_F16); a two-plane net cannot run them
 * and falls back to another tile at launch.) */
int vgh_conv_split_cfg_ok(int cfg, int ksize, int stride, int cout_pad, int fast_epilogue, int shuffle, int grp_cout);
/* Cap on the persistent 3x3 kernels' grid: at most `blocks` workgroups per XCD (0 = as many as stay resident, the default).
 * Process-wide.  Leaves CUs to other work; the parity tests use it to drive many tiles through one workgroup. */
int vgh_conv_set_max_blocks_per_xcd(int blocks);
/* Process-wide, read by vgh_net_create (default on): an int8 -> bf16 3x3 conv whose rows are dominated by w[c][centre][c] -- at least half of the live rows have it as
 * their largest weight: the identity branch a RepVGG block folds into its kernel -- keeps that element out of the int8 image and applies it in fp32 in the epilogue
 * (vgh_conv_call.diag_dev).  Off: the plain per-cout int8 grid (for the comparison; ~17 dB less weight precision on such rows). */
int vgh_net_set_i8_diag(int on);
int vgh_net_op_has_diag(vgh_net* net, int op_index); /* 1: the op of a created network runs with the diagonal bypass */

/* ------------------------------------------------------------------------------------------------
 * Head decode: replaces YoloHeadsNDFLHeads.forward's tail (yolo_head_ndfl_heads.py:143-172) and the
 * activations of YoloHeadsDFLHead.forward (yolo_head_dfl_head.py:162-184).  `levels` describe the
 * fp32 NHWC prediction buffers written by the *_pred 1x1 convs, channel order per pixel:
 *   [reg 68 (side*17+bin) | cls 1 | 3 unused | shape S | expr E | rot 6 | jaw 3 | trans 3 | scale 1]
 * (VGH_PRED_FLAME_OFF = 72: the three unused floats keep every segment 16-byte aligned so the *_pred convs store float4s).
 * ---------------------------------------------------------------------------------------------- */
#define VGH_PRED_CLS_OFF 68
#define VGH_PRED_FLAME_OFF 72
typedef struct vgh_head_level {
    const float* pred_dev; /* [B, h*w, pitch] */
    int32_t h, w, pitch, stride;
} vgh_head_level;

/* boxes_dev [B,A,4] xyxy px, scores_dev [B,A] = sigmoid(cls).  A = sum(h*w), level-major, row-major. */
int vgh_head_decode(const vgh_head_level* levels, int n_levels, int B, float* boxes_dev, float* scores_dev, void* stream);

/* VGGHeadDecodingModule.forward (yolo_heads.py:63-86): per image top-k (sorted descending, ties by
 * ascending anchor index).  idx_dev [B,k] int32 anchor indices, topk_scores_dev [B,k] (may be NULL). */
int vgh_topk(const float* scores_dev, int B, int A, int k, int32_t* idx_dev, float* topk_scores_dev, void* stream);

/* Gather boxes + build the 413-vector for the selected anchors only (never materialises [B,A,413]):
 * tanh*3 / exp/0.05 activations, zero padding to 300/100, translation[:2] += cell centre,
 * scale *= stride, and the from_3dmm -> to_3dmm_tensor channel permutation
 * (yolo_head_ndfl_heads.py:167-172; head_info.py:44-109).  shape_c / expr_c = live channels S / E. */
int vgh_gather_candidates(const vgh_head_level* levels, int n_levels, int B, int A, int shape_c, int expr_c, const float* boxes_dev,
                          const int32_t* idx_dev, int k, float* out_boxes_dev /*[B,k,4]*/, float* out_flame_dev /*[B,k,413]*/, void* stream);

/* nms() of head_detector/utils.py:159-194 for EVERY image (the batched twin,
 * yolo_heads_post_prediction_callback.py:55-84): conf filter (>=), greedy NMS with torchvision
 * semantics (suppress when IoU > iou_thr, float32 IEEE arithmetic), first keep_k survivors.
 * Inputs must be sorted by descending score per image (vgh_topk output order), n_in <= 1024.
 * keep_idx_dev [B,keep_k] int32 positions into the n_in candidates (-1 padded), counts_dev [B]. */
int vgh_nms(const float* boxes_dev /*[B,n_in,4]*/, const float* scores_dev /*[B,n_in]*/, int B, int n_in, float conf_thr, float iou_thr,
            int keep_k, int32_t* keep_idx_dev, int32_t* counts_dev, void* stream);

/* Compact survivors into fixed-capacity slabs: boxes [B,keep_k,4], scores [B,keep_k], flame [B,keep_k,413]. */
int vgh_compact(const float* boxes_dev, const float* scores_dev, const float* flame_dev, int B, int n_in, const int32_t* keep_idx_dev,
                int keep_k, float* out_boxes_dev, float* out_scores_dev, float* out_flame_dev, void* stream);

/* The whole of nms() (head_detector/utils.py:159-194; batched twin yolo_heads_post_prediction_callback.py:55-84) for EVERY image
 * on arbitrary (unsorted) inputs: conf filter, top-k (pre_k), greedy NMS, first keep_k survivors, gathered from the ORIGINAL
 * tensors -- boxes [B,n,4], scores [B,n], flame [B,n,flame_width] (or NULL) -> slabs [B,keep_k,{4,1,flame_width}] + counts [B].
 * min(pre_k, n) <= 1024.  workspace_dev: caller-owned, 16-byte aligned, vgh_topk_nms_workspace_bytes(...) bytes. */
int64_t vgh_topk_nms_workspace_bytes(int B, int n, int pre_k, int keep_k);
int vgh_topk_nms(const float* boxes_dev, const float* scores_dev, const float* flame_dev, int flame_width, int B, int n, float conf_thr, float iou_thr,
                 int pre_k, int keep_k, void* workspace_dev, float* out_boxes_dev, float* out_scores_dev, float* out_flame_dev, int32_t* counts_dev,
                 void* stream);

/* ------------------------------------------------------------------------------------------------
 * FLAME: replaces FLAMELayer (head_detector/flame.py:37-169) + smplx.lbs.lbs + rot_mat_from_6dof
 * (utils.py:120-128) + reproject_spatial_vertices (flame.py:179-208) + the vertex un-pad/un-scale of
 * HeadDetector._parse_predictions (detector.py:67-69).
 * vgh_flame_create takes exactly the buffers FLAMELayer.__init__ registers (flame.py:75-95), host f32:
 *   v_template [V,3], shapedirs [V,3,NB], posedirs [(NJ-1)*9, 3V], J_regressor [NJ,V] dense,
 *   parents [NJ] (parents[0] = -1), lbs_weights [V,NJ].
 * A handle owns one set of scratch buffers (per-head coefficients and packed transforms): decodes are asynchronous on the
 * stream they are given, and a decode queued on a different stream than the previous one is ordered after it on the device
 * (event wait, no host sync) -- calls from several streams are safe, they just do not overlap.  Not safe for concurrent
 * calls from several host threads; use one handle per thread.
 * Device memory of a handle: two copies of the blend basis (K x 3 x V floats each: the planar one and, since r04, a k-interleaved one for
 * the component-split tiles; 2 x 26.5 MB for FLAME) + the scratch (max_heads x (K + 128) floats).
 * ---------------------------------------------------------------------------------------------- */
typedef struct vgh_flame vgh_flame;
int vgh_flame_create(int device, int V, int NB, int NJ, const float* v_template, const float* shapedirs, const float* posedirs,
                     const float* J_regressor, const int32_t* parents, const float* lbs_weights, int max_heads, vgh_flame** out);
void vgh_flame_destroy(vgh_flame* f);

/* reproject_spatial_vertices(flame, params, to_2d=False): params_dev [n,413] (network OUTPUT layout,
 * read as from_3dmm does: shape300|expr100|jaw3|rot6|trans3|scale1).  Any output may be NULL.
 *   verts_dev [n,V,3]  FLAMELayer.forward(zero_rot=True) (incl. z += 0.05)
 *   rot_dev   [n,3,3]  rot_mat_from_6dof
 *   proj_dev  [n,V,3]  (R v) * clamp(scale,1e-8) + translation, then if unpad_dev != NULL
 *                      (x - pad_x, y - pad_y, z) / scale_factor with unpad_dev [n,3] = (pad_x,pad_y,scale)
 * shape_live / expr_live: number of leading shape / expression coefficients that can be non-zero
 * (300/100 = no assumption). Coefficients beyond them MUST be exactly 0 (the detector zero-pads them,
 * yolo_head_dfl_head.py:170-182); skipping them is then bit-exact. */
int vgh_flame_decode(vgh_flame* f, const float* params_dev, int n, int shape_live, int expr_live, const float* unpad_dev, float* verts_dev,
                     float* rot_dev, float* proj_dev, void* stream);

/* vgh_flame_decode for a head list that lives on the device (no host round trip for the data-dependent n):
 * head i (i < *n_heads_dev <= capacity <= max_heads) reads params_dev[head_row_dev[i]] and, if unpad_dev != NULL,
 * unpad_dev[head_image_dev[i]] ([images,3]); outputs are compact rows i.  Kernels are launched at `capacity`
 * and exit beyond the live count.  rpy_dev [capacity,3] = calculate_rpy (utils.py:146-151: scipy
 * Rotation.from_matrix(R^T).as_euler("xyz", degrees) in closed form + limit_angle) as (roll, pitch, yaw). */
int vgh_flame_decode_indirect(vgh_flame* f, const float* params_dev, const int32_t* head_row_dev, const int32_t* head_image_dev,
                              const int32_t* n_heads_dev, int capacity, int shape_live, int expr_live, const float* unpad_dev, float* verts_dev,
                              float* rot_dev, float* rpy_dev, float* proj_dev, void* stream);

/* General FLAMELayer.forward core = smplx lbs(betas, full_pose): betas_dev [n,NB], pose_dev [n,3*NJ]
 * axis-angle per joint -> verts_dev [n,V,3] (NO z offset, NO global rotation), joints_dev [n,NJ,3] or NULL. */
int vgh_flame_lbs(vgh_flame* f, const float* betas_dev, const float* pose_dev, int n, float* verts_dev, float* joints_dev, void* stream);
/* The vertex stage has interchangeable kernels that produce bit-identical vertices (each output is the same fp32 fmaf chain in ascending
 * k): VALU FMAs, FP32 matrix cores fed from registers (v_mfma_f32_32x32x2_f32), LDS-staged matrix-core tiles, and (r04) component-split
 * matrix-core tiles (one coordinate plane per wave, coefficient tile in LDS; prologue waves inside the block up to 8 heads).  mode
 * (process-wide, atomic; for parity tests and A/B measurements): 0 VALU only, 1 automatic by batch size (default), 2 register-fed matrix
 * cores, 3 / 4 / 5 the LDS-staged tiles (128- / 64- / 32-head blocks) where their tables fit, else automatic, 6 / 7 the component-split
 * tiles always (6: with the prologue kernel at every batch size; 7: fused up to 8 heads) where the model fits them (even ranges, NB a
 * multiple of 8), else automatic.  The meaning of 1 changed in r04 (it now picks the component-split tiles up to 112 heads); the set of
 * values grew from 0..5 to 0..7: VGH_ABI_VERSION 5. */
int vgh_flame_set_matrix_path(int mode);

/* ------------------------------------------------------------------------------------------------
 * Fused detector: HeadDetector._process + the device-side arithmetic of _parse_predictions
 * (head_detector/detector.py:54-90) for a whole batch behind one asynchronous call.
 *   vgh_detector_candidates = `self.model(image)` (detector.py:58-59; VGGHeadDecodingModule.forward, yolo_heads.py:44-86):
 *       fills the detector-owned candidate buffers boxes [max_batch,pre_k,4], scores [max_batch,pre_k], flame [max_batch,pre_k,413]
 *       (vgh_detector_candidate_buffers returns their device pointers)
 *   vgh_detector_select     = nms() for EVERY image (utils.py:159-194 / yolo_heads_post_prediction_callback.py:55-84) into the
 *       caller's fixed-capacity slabs + the image-major head list + reproject_spatial_vertices / un-pad / calculate_rpy of
 *       every survivor (detector.py:66-69,87)
 *   vgh_detect              = both.
 * Nothing is allocated after vgh_detector_create; all outputs are caller-owned device memory; every call is asynchronous
 * on `stream`; a detector (like the net and FLAME handles it borrows) serves one stream at a time.
 * ---------------------------------------------------------------------------------------------- */
#define VGH_MAX_LEVELS 4
typedef struct vgh_detect_cfg {
    int32_t n_levels;
    int32_t level_buf[VGH_MAX_LEVELS];    /* net buffer id of each level's fp32 prediction tensor (see vgh_head_level) */
    int32_t level_h[VGH_MAX_LEVELS], level_w[VGH_MAX_LEVELS], level_pitch[VGH_MAX_LEVELS], level_stride[VGH_MAX_LEVELS];
    int32_t shape_live, expr_live;        /* live shape / expression channels of the heads (L: 128/64, M: 64/32) */
    int32_t pre_k, keep_k;                /* 1000 / 100 in the reference */
    int32_t max_batch;                    /* may exceed the net's arena batch: processed in arena-sized chunks */
} vgh_detect_cfg;

typedef struct vgh_detect_out {
    float* boxes_dev;        /* [B,keep_k,4] xyxy px in the padded S-space (rows >= count are zero) */
    float* scores_dev;       /* [B,keep_k] */
    float* flame_dev;        /* [B,keep_k,413] network layout */
    int32_t* counts_dev;     /* [B] survivors per image */
    /* per-head outputs, image-major compact rows; all optional, n_heads_dev mandatory if any is set */
    int32_t* n_heads_dev;    /* [1] min(sum(counts), head_capacity) */
    int32_t* head_image_dev; /* [head_capacity] image of each head or NULL */
    int32_t head_capacity;   /* rows available in the per-head outputs; <= 0: B*keep_k */
    const float* unpad_dev;  /* [B,3] (pad_x, pad_y, scale) per IMAGE or NULL (detector.py:67-69) */
    float* verts_dev;        /* [head_capacity,5023,3] FLAMELayer.forward(zero_rot=True) or NULL */
    float* rot_dev;          /* [head_capacity,3,3] or NULL */
    float* rpy_dev;          /* [head_capacity,3] roll, pitch, yaw degrees or NULL */
    float* proj_dev;         /* [head_capacity,5023,3] projected (and un-padded) vertices or NULL */
} vgh_detect_out;

typedef struct vgh_detector vgh_detector;
/* `flame` may be NULL (then the FLAME outputs of vgh_detect_out must be NULL). The handles are borrowed, not owned. */
int vgh_detector_create(vgh_net* net, vgh_flame* flame, const vgh_detect_cfg* cfg, vgh_detector** out);
void vgh_detector_destroy(vgh_detector* d);
int vgh_detector_candidates(vgh_detector* d, const void* images_dev, int image_fmt, int B, void* stream);
/* The post-network half of vgh_detector_candidates alone (box/score decode, top-k, gather) for the n images currently in the
 * net's prediction buffers -> candidate rows [at, at+n).  With vgh_net_forward before it, this is what vgh_detector_candidates
 * does per arena-sized chunk; separate so a caller can bracket the network with its own events. */
int vgh_detector_decode_candidates(vgh_detector* d, int n, int at, void* stream);
int vgh_detector_candidate_buffers(vgh_detector* d, float** boxes_dev, float** scores_dev, float** flame_dev);
int vgh_detector_set_flame(vgh_detector* d, vgh_flame* flame);
/* detector-owned intermediates (parity tests / debugging): dense boxes [max_batch,A,4], dense scores [max_batch,A],
 * top-k anchor indices [max_batch,pre_k] i32, NMS keep positions [max_batch,keep_k] i32, head rows [max_batch*keep_k] i32 */
#define VGH_SCRATCH_BOXES_ALL 0
#define VGH_SCRATCH_SCORES_ALL 1
#define VGH_SCRATCH_TOPK_IDX 2
#define VGH_SCRATCH_KEEP_IDX 3
#define VGH_SCRATCH_HEAD_ROW 4
void* vgh_detector_scratch(vgh_detector* d, int which);
int vgh_detector_select(vgh_detector* d, int B, float conf_thr, float iou_thr, vgh_detect_out* out, void* stream);
int vgh_detect(vgh_detector* d, const void* images_dev, int image_fmt, int B, float conf_thr, float iou_thr, vgh_detect_out* out, void* stream);
/* Throughput mode.  With overlap enabled the select half (NMS, compaction, head list, FLAME decode: small latency-bound
 * kernels, ~0.3 ms per batch) AND the candidate half (decode, top-k, gather) are queued on a detector-owned side stream, so
 * they run underneath the NETWORK OF THE NEXT BATCH queued on the caller's stream; the next forward waits (vgh_net_set_pred_guard)
 * only before it overwrites the prediction buffers.  Ordering rules: the candidates / outputs of call s are complete once
 * `vgh_detector_join(det, stream)` has been ordered after it (it makes `stream` wait for everything queued on the side
 * stream); the caller must not reuse the vgh_detect_out buffers of call s for call s+1 unless it joined in between (or does
 * not read them). */
int vgh_detector_set_overlap(vgh_detector* d, int enable);
/* r06 (additive): LAZY FLAME GATHER.  enable = 1: vgh_detector_decode_candidates / vgh_detector_candidates gather the candidates' boxes only; the next
 * vgh_detector_select builds the 413-vectors of the SURVIVORS straight from the prediction buffers (the [B, pre_k, 413] candidate tensor -- 106 MB per 64 images for a handful of
 * survivors per image -- is neither written nor read: VGH_SCRATCH_CAND_FLAME is stale).  Contract: that select is queued before the next forward overwrites the prediction
 * buffers (vgh_detect does; in overlap mode the prediction guard moves behind the select by itself).  A batch that runs in several arena chunks gathers eagerly whatever the
 * flag says.  The detections are the same bits.  Default 0. */
int vgh_detector_set_lazy_flame(vgh_detector* d, int enable);
int vgh_detector_join(vgh_detector* d, void* stream);
/* Records the caller's HIP event behind everything queued so far for the post-network stages (overlap mode: on the detector's side stream; else on `stream`)
 * WITHOUT making any stream wait for it: a host that synchronises on the event of an EARLIER batch can queue that batch's consumers (e.g. the N>1 exchange) with no
 * device-side wait -- a hardware queue parked behind the low-priority side stream costs the network ~7 % (r05, tools/exchange_probe.py). */
int vgh_detector_record(vgh_detector* d, void* event, void* stream);
int vgh_detector_streams(vgh_detector* d, void* main_stream, void** out /*[4]*/);

/* ------------------------------------------------------------------------------------------------
 * Whole-pipeline context from ONE pack file: replaces HeadDetector.__init__ (head_detector/detector.py:19-30: hub download,
 * torch.jit.load of the .trcd, FLAMELayer()) and, with vgh_ctx_detect, HeadDetector._process + the device arithmetic of
 * _parse_predictions for a whole batch (detector.py:54-90) -- no Python at run time.  The pack (.vghpack, written by
 * `python -m head_detector_amd.pack <variant> <weights.trcd | seed:N> <generic_model.pkl | seed:N> out.vghpack`) holds the
 * lowered op program, the folded fp32 weights, the per-op tile choices (by name) and the FLAME constants, behind a versioned
 * header.  A context owns its net / FLAME / detector handles; it is NOT thread-safe: one context per (device, stream).
 * vgh_ctx_last_error(ctx) returns the message of the last failed call ON THAT CONTEXT (vgh_last_error() stays thread-local).
 * ---------------------------------------------------------------------------------------------- */
typedef struct vgh_config {
    int32_t device;
    const char* pack_path;
    int32_t max_batch;                 /* images per vgh_ctx_detect call (processed in arena-sized chunks when tensors would pass 2 GiB) */
    int32_t pre_nms_top_k, keep_top_k; /* 0: the reference's 1000 / 100 */
    int32_t max_heads;                 /* FLAME decode capacity; 0: max_batch * keep_top_k */
    int32_t batch_split;               /* vgh_net_set_split (0 / 1: off) */
    int32_t overlap;                   /* vgh_detector_set_overlap: results complete after vgh_ctx_join */
} vgh_config;
typedef struct vgh_ctx_info {
    char variant[32];
    int32_t image_size, max_batch, arena_batch, num_anchors, pre_nms_top_k, keep_top_k, num_vertices, shape_live, expr_live, precision;
    double flops_per_image;
} vgh_ctx_info;
typedef struct vgh_ctx vgh_ctx;
int vgh_create(const vgh_config* cfg, vgh_ctx** out);
void vgh_destroy(vgh_ctx* ctx);
const char* vgh_ctx_last_error(const vgh_ctx* ctx);
int vgh_ctx_get_info(const vgh_ctx* ctx, vgh_ctx_info* info);
/* = vgh_detect on the context's detector (outputs: caller-owned device slabs, see vgh_detect_out). */
int vgh_ctx_detect(vgh_ctx* ctx, const void* images_dev, int image_fmt, int B, float conf_thr, float iou_thr, vgh_detect_out* out, void* stream);
int vgh_ctx_join(vgh_ctx* ctx, void* stream);
/* borrowed handles (for callers that mix the context with the fine-grained entry points above) */
vgh_net* vgh_ctx_net(vgh_ctx* ctx);
vgh_flame* vgh_ctx_flame(vgh_ctx* ctx);
vgh_detector* vgh_ctx_detector(vgh_ctx* ctx);

/* ------------------------------------------------------------------------------------------------
 * Result-side consumers of the decoded meshes.
 * vgh_rasterize = Sim3DR.rasterize(vertices, triangles, colors, bg=image, reverse) (head_detector/Sim3DR/Sim3DR.py:17-38 ->
 *   _rasterize, head_detector/Sim3DR/lib/rasterize_kernel.cpp:219-293) with the binding's defaults alpha = 1 and a fresh
 *   -1e8 depth buffer: z-buffer rasterisation of one mesh INTO image_dev (uint8 [H,W,channels], in/out), triangle order
 *   semantics preserved (strictly deeper wins, ties to the earliest triangle); bit-identical to the reference's C++.
 *   verts_dev [V,3] f32 (x, y in pixels, z = depth), tri_dev [ntri,3] i32, colors_dev [V,channels] f32,
 *   zbuf_dev: caller-owned scratch of H*W uint64.
 * vgh_pncc_render = PNCCProcessor.__call__ (head_detector/pncc_processor.py:66-73): zeroes image_dev [H,W,3], then paints
 *   the heads in order (each with z negated and its own depth buffer; pixels whose painted colour is all-zero keep the
 *   earlier content).  verts_dev [n_heads,V,3] are read, NOT modified (the reference negates z in place on the host array;
 *   the Python facade reproduces that side effect).
 * vgh_refined_head_bbox = refined_head_bbox (head_detector/utils.py:26-35) for every head: out_dev [n_heads,4] i32 (x,y,w,h).
 * ---------------------------------------------------------------------------------------------- */
int vgh_rasterize(const float* verts_dev, const int32_t* tri_dev, int ntri, const float* colors_dev, int channels, uint8_t* image_dev, int H, int W,
                  int reverse, uint64_t* zbuf_dev, void* stream);
int vgh_pncc_render(const float* verts_dev, int n_heads, int V, const int32_t* tri_dev, int ntri, const float* colors_dev, uint8_t* image_dev, int H, int W,
                    uint64_t* zbuf_dev, void* stream);
int vgh_refined_head_bbox(const float* verts_dev, int n_heads, int V, const int32_t* idx_dev, int n_idx, int32_t* out_dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Letterbox: HeadDetector._transform_image (head_detector/detector.py:40-52) on the device:
 * cv2.resize(..., INTER_LANCZOS4) in OpenCV's 8-bit fixed-point arithmetic (8x8 taps, short weights scaled by 2048, int32
 * accumulation, (v + 2^21) >> 22, edge replication) + cv2.copyMakeBorder(BORDER_CONSTANT) into the u8 NHWC canvas [S,S,3] that
 * vgh_net_forward(VGH_IMG_U8_NHWC) consumes.  The caller supplies the per-axis tables exactly as resize.cpp builds them
 * (head_detector_amd/letterbox.py: xofs/yofs = floor source coordinate, alpha/beta = [n][8] fixed-point weights) and the
 * placement of the resized image; src is u8 [src_h, src_w, src_channels >= 3] with the given row pitch; pad_rgb is HOST memory.
 * ---------------------------------------------------------------------------------------------- */
int vgh_letterbox(const uint8_t* src_dev, int src_h, int src_w, int src_channels, int64_t src_pitch_bytes, const int32_t* xofs_dev,
                  const int16_t* alpha_dev, const int32_t* yofs_dev, const int16_t* beta_dev, int new_w, int new_h, int pad_x, int pad_y,
                  const uint8_t* pad_rgb, uint8_t* dst_dev, int S, void* stream);

/* ------------------------------------------------------------------------------------------------
 * HIP-event helpers so Python can time work on the stream the kernels actually run on.
 * ---------------------------------------------------------------------------------------------- */
int vgh_stream_create(int device, void** stream_out);
/* Streams that really run side by side.  HIP multiplexes a process's streams onto a few hardware queues (4 by default) and two
 * streams on one queue execute serially; which queue a new stream gets depends on the process's creation history (measured:
 * the same two-lane forward 5.1 ms vs 7.1 ms, profiles/r02_stream_queues.txt).  vgh_stream_acquire returns a stream MEASURED
 * (150 us spin-kernel pairs) to overlap with every stream in avoid[0..n_avoid) -- earlier entries win when the queues do not
 * suffice -- taking it from a per-device park of earlier candidates / released streams before creating new ones; it
 * synchronises the streams in `avoid`.  The net's lane streams and the detector's side stream are picked this way on first use.
 * vgh_stream_release parks a stream (never destroyed).  vgh_streams_overlap: the measurement itself (1 = side by side).
 * vgh_detector_streams: the streams a detector uses for work entering on main_stream: out[0..2] net lanes, out[3] the
 * overlap-mode side stream (NULL when overlap is off) -- e.g. to acquire a communication stream that avoids them. */
int vgh_stream_acquire(int device, void* const* avoid, int n_avoid, void** stream_out);
int vgh_stream_release(int device, void* stream);
int vgh_streams_overlap(void* stream_a, void* stream_b);
/* r06: 1 when a kernel queued on stream_b cannot START while stream_a's dispatches are being placed, although the two streams sit on different hardware queues:
 * the queues share a compute pipe (queues are spread over the pipes in creation order; a pipe places one dispatch at a time and serves its higher-priority queue
 * first).  This is what starved the detector's low-priority side stream in r05 (csrc/streams.hip, profiles/r06_starved_side_stream_classes.txt); the side stream is
 * now acquired clear of the pipes of the caller's stream and the first lane, by this measurement. */
int vgh_stream_blocked_behind(void* stream_a, void* stream_b);
/* One wave busy-waiting for `microseconds` on `stream`: the probe behind vgh_streams_overlap, exported so that a host can test
 * streams it does not own (e.g. a communication library's internal stream) for a shared hardware queue. */
int vgh_stream_spin(void* stream, int microseconds);
int vgh_stream_destroy(void* stream);
int vgh_stream_sync(void* stream);
int vgh_event_create(void** ev_out);
int vgh_event_destroy(void* ev);
int vgh_event_record(void* ev, void* stream);
int vgh_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out); /* synchronises on ev_stop */

/* ------------------------------------------------------------------------------------------------
 * EXPERIMENT KNOBS (r06): declared and compiled ONLY in the -DVGH_EXPERIMENTS build (head_detector_amd/libvgh_exp.so, `python -m head_detector_amd.build --experiments`),
 * which tools/ drive through VGH_LIB_PATH.  Every one is an A/B switch for something that was measured and did not pay (EXPERIMENTS.md); the product library neither
 * exports them nor contains the kernels behind them (the fused stem + downsample kernel, the matrix-core stem, the stride-2 parity-plane "d" tiles).
 * ------------------------------------------------------------------------------------------------ */
#ifdef VGH_EXPERIMENTS
/* Process-wide: with a batch split, lane l starts when lane l-1 has finished its first `ops` ops and stays that far behind, so that different layers
 * run side by side (0 = the lanes advance together).  Results do not change. */
int vgh_net_set_lane_lag(int ops);
/* Opt-in: the stem (3 -> 48, stride 2) and the first backbone downsample (48 -> 96, stride 2) as ONE kernel (csrc/stem_ds.hip) when the program has that
 * pair in bf16: the 48-channel stem activation then never goes to HBM and its arena buffer is not written.  Results are bit-identical either way.
 * Default off: measured (r03) it removes 1.5 GB of traffic per 64-image forward but is no faster than the two launches (latency-bound small tiles). */
int vgh_net_set_fuse_stem(vgh_net* net, int enable);
/* Process-wide opt-in (default 0): in the bf16 mode the stem of a u8 image runs as a K = 27 bf16 GEMM on the matrix cores (csrc/stem_pool.hip::stem_mfma_kernel: pixel
 * values are exact in bf16, /255 folded into bf16-rounded weights, fp32 accumulate) instead of the exact-fp32 VALU kernel.  Correct (2^-9 relative on a weight, below
 * the bf16 rounding of the stem's output) but measured SLOWER in r05 (442 vs 306 us per 64 images: per-block latency chain), hence off; float images and the parity
 * modes always use the exact kernel. */
int vgh_stem_set_mfma(int on);
/* Process-wide: bf16 conv outputs are stored with the non-temporal hint (evict-first in L2, so the input lines neighbouring tiles re-read survive).
 * Results do not change. */
int vgh_conv_set_nt_store(int on);
/* the priority class of the side stream the overlapped post stages run on (1 = lowest, the default; 0 = the caller's); between batches (after vgh_detector_join) */
int vgh_detector_set_side_priority(vgh_detector* d, int low);
/* first-kernel completion / pair completion (per mille) of two multi-round kernels launched back to back on a and b: ~1000 = their workgroups interleave */
int vgh_streams_interleave_permille(void* stream_a, void* stream_b);
#endif

#ifdef __cplusplus
}
#endif
#endif /* VGH_H_ */
