// Host side of libvgh.so: static-schedule network executor (pre-planned arena, concat-by-offset, one
// stream, optional hipGraph replay), stand-alone conv entry point, error/stream/event helpers.
//
// Replaces the TorchScript interpreter call `self.model(image)` (head_detector/detector.py:58-59).
#include <stdarg.h>

#include <atomic>
#include <vector>

#include "vgh_internal.h"

static thread_local char g_err[1024] = "";

void vgh_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

struct NetOp {
    vgh_op_desc d;
    uint16_t* wpack = nullptr;  // device (conv)
    float* wf32 = nullptr;      // device (stem [27][48])
    float* bias = nullptr;      // device
    float out_scale = 1.0f;     // split parity modes: accumulator scale of the op (vgh_pack_conv_weights_split_host)
    // automatic tile of the op, resolved ONCE for the arena batch: every forward -- any batch size, any chunk, any lane -- then runs the op
    // on the same tile, so its fp32 summation order (hence every output bit) does not depend on how many images ride along
    int auto_cfg = -1;  // the automatic tile of this op, resolved ONCE at vgh_net_create for the arena batch (chunk- and lane-independent bits)
    uint16_t* wds = nullptr;    // stage-1 downsample only: its weights in the fused stem + downsample kernel's layout (stem_ds.hip)
    // the input view is WIDER than the buffer's pitch (the 48-channel stem tensor read as 64-channel K blocks): channels past the pitch are the next
    // pixel's first ones; vgh_net_create has verified that every weight row is exactly zero there (finite x 0 adds +-0 to the accumulator)
    int overhang_ok = 0;
    float* gscale = nullptr;  // device [cout_pad]: per-cout output factors of an op that reads or writes a VGH_FMT_FP8 buffer (conv_pp.hip), else nullptr
    // back-to-back GEMM (r06, conv_kernels.inc T2 > 0): b2b = 1: this conv and the NEXT op (a 1x1 / stride-1 conv, the only reader of this conv's output tensor) run as ONE
    // launch that never writes the tensor in between; b2b = 2: this op is that second conv (nothing to launch when the net fuses)
    int b2b = 0;
    // exact cross-lane dependencies (r06, arch.schedule_latency): vgh_op_desc.lane = lane | (1 + producer op index) << 8.  ev_done: recorded behind this op when a later op
    // on ANOTHER lane names it as its producer (created at vgh_net_create), else nullptr
    hipEvent_t ev_done = nullptr;
    float* dvec = nullptr;    // device [cout_pad]: the diagonal bypass of an int8 -> bf16 conv whose rows are dominated by w[c][centre][c] (i8_peel_diag), else nullptr
};

static std::atomic<int> g_i8_diag{1};
// An int8 -> bf16 3x3 conv takes the diagonal bypass (conv_pp.hip DG) when at least half of its live rows have w[c][centre][c] as their largest weight -- the folded
// identity branch of a RepVGG block.  Exact either way (one element of the sum is evaluated in fp32 instead of through the int8 image); what it buys is the int8 grid
// of the remaining weights.
static bool i8_peel_diag(const float* w, const vgh_op_desc& d) {
    if (!g_i8_diag.load(std::memory_order_relaxed) || d.ksize != 3 || d.stride != 1 || d.grp_cout || d.cin < d.cout_pad || d.cout_pad > 1024) return false;
    const size_t row = (size_t)9 * d.cin;
    int live = 0, dom = 0;
    for (int c = 0; c < d.cout_pad; ++c) {
        float mx = 0.0f;
        for (size_t i = 0; i < row; ++i) mx = fmaxf(mx, fabsf(w[(size_t)c * row + i]));
        if (mx > 0.0f) {
            ++live;
            if (fabsf(w[(size_t)c * row + (size_t)4 * d.cin + c]) >= mx) ++dom;
        }
    }
    return live > 0 && 2 * dom >= live;
}

struct vgh_net {
    int device = 0, image_size = 0, max_batch = 0;
    std::vector<vgh_buf_desc> bufs;
    std::vector<void*> buf_ptr;
    std::vector<int64_t> buf_bytes;
    std::vector<NetOp> ops;
    char* arena = nullptr;
    int64_t arena_bytes = 0;
    char* wblob = nullptr;  // all packed weights + biases
    uint16_t* zeros = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    static constexpr int kLanes = 4;  // lane 0 = the caller's stream
    hipStream_t side[kLanes] = {nullptr, nullptr, nullptr, nullptr};  // picked by ensure_lanes for `lanes_main`
    hipStream_t lanes_main = nullptr;
    bool lanes_ready = false;
    hipEvent_t ev_fork = nullptr, ev_join[kLanes] = {nullptr, nullptr, nullptr, nullptr}, ev_lag[kLanes] = {nullptr, nullptr, nullptr, nullptr};
    // optional guard (borrowed event): the first op that writes an fp32 prediction buffer waits for it, so a consumer of the
    // PREVIOUS forward's predictions may still be running on another stream while this forward's backbone / neck execute
    hipEvent_t pred_guard = nullptr;
    // batch split: the batch runs as `nsplit` independent sub-batches on the lane streams, so the fixed cost of every launch
    // (dispatch, tile prologue, first-load latency, epilogue store burst, tail) of one sub-batch hides under the main loops of
    // the others (measured on the M net at B = 32: 5.75 ms -> 5.35 ms with 4 lanes)
    int nsplit = 1;
    // stem + stage-1 downsample as ONE kernel (stem_ds.hip: the 48-channel stem activation stays in LDS): index of the stem op when the pair
    // qualifies (bf16 mode, the architecture's 3x3 / stride-2 / 64 -> 96 conv as the stem tensor's only reader), else -1; results are bit-identical
    int stem_pair = -1;
    // two SIDE lanes that wait for each other's ops (lane bits 8+): fine eagerly, but under stream capture ROCm 7.0's runtime links a non-origin stream to the stream of
    // every event it waits for and hipStreamEndCapture then recurses without end (measured r06: a stack overflow inside hip::Stream::EndCapture) -> vgh_net_capture refuses
    bool lane_wait_cycle = false;
    int fuse_b2b = 1;   // vgh_net_set_b2b: the back-to-back pairs found at create time run fused (default) or as their two launches (every intermediate tensor then exists)
    int stem3_ok = -1;  // the stem + stage-1 pair as ONE launch (ds_b2b.hip, "u" tile; u8 images, fuse_b2b == 1): -1 not yet checked (first forward), 0 / 1
    int fuse_stem = 0;  // opt-in (vgh_net_set_fuse_stem): measured r03, the fused kernel saves 1.5 GB of HBM traffic per L b64 forward but no time (EXPERIMENTS.md 8c)
};

static inline int64_t buf_image_bytes(const vgh_buf_desc& b) { return (int64_t)b.h * b.w * b.pitch * vgh_fmt_bytes(b.is_f32); }

static int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

static int net_conv_args(vgh_net* n, const NetOp& op, int B, int at, ConvArgs* a) {
    const vgh_op_desc& d = op.d;
    const vgh_buf_desc& ib = n->bufs[d.in_buf];
    const vgh_buf_desc& ob = n->bufs[d.out_buf];
    memset(a, 0, sizeof(*a));
    a->in = (const uint16_t*)((const char*)n->buf_ptr[d.in_buf] + at * buf_image_bytes(ib));
    // split parity modes: a pixel is [hi plane | lo plane]; the kernels address it with the physical pitch and the plane stride
    const int ipl = vgh_fmt_planes(ib.is_f32), opl = vgh_fmt_planes(ob.is_f32);
    const bool h16 = ib.is_f32 == VGH_FMT_F16;  // single-plane fp16 (r05): the fp16 split kernels with ONE K segment and no lo plane (ConvArgs::nseg)
    a->split = h16 ? VGH_FMT_F16X2 : ipl > 1 ? ib.is_f32 : 0;
    a->nseg = h16 ? 1 : 3;
    a->in_plane = h16 ? 0 : ib.pitch;
    a->out_plane = h16 ? 0 : ob.pitch;
    a->out_scale = op.out_scale;
    a->grp_cout = d.grp_cout;
    a->grp_in_stride = d.grp_in_stride;
    a->in_pitch = (int64_t)ib.pitch * ipl;
    a->in_coff = d.in_coff;
    a->cin = d.cin;
    a->B = B;
    a->H = ib.h;
    a->W = ib.w;
    a->ksize = d.ksize;
    a->stride = d.stride;
    a->pad = d.ksize / 2;
    a->Ho = (ib.h + 2 * a->pad - d.ksize) / d.stride + 1;
    a->Wo = (ib.w + 2 * a->pad - d.ksize) / d.stride + 1;
    a->wpack = op.wpack;
    a->bias = op.bias;
    a->out = (char*)n->buf_ptr[d.out_buf] + at * buf_image_bytes(ob);
    a->out_pitch = (int64_t)ob.pitch * opl;
    a->out_coff = d.out_coff;
    a->out_coff2 = d.out_coff2;
    a->out_split = d.out_split;
    a->cout_pad = d.cout_pad;
    a->cout_store = d.cout_store;
    a->out_f32 = ob.is_f32 == VGH_FMT_F32;
    a->res = d.res_buf >= 0 ? (const uint16_t*)((const char*)n->buf_ptr[d.res_buf] + at * buf_image_bytes(n->bufs[d.res_buf])) : nullptr;
    a->res_pitch = d.res_buf >= 0 ? (int64_t)n->bufs[d.res_buf].pitch * vgh_fmt_planes(n->bufs[d.res_buf].is_f32) : 0;
    a->res_plane = (d.res_buf >= 0 && !h16) ? n->bufs[d.res_buf].pitch : 0;
    // e4m3 links (r05): bf16 -> e4m3, e4m3 -> bf16 and e4m3 -> e4m3 are legal pairs (3x3 / stride-1 convs on the ping-pong tiles); the residual of such an op is bf16
    const bool in8 = vgh_fmt_is_q8(ib.is_f32), out8 = vgh_fmt_is_q8(ob.is_f32);
    a->in_fp8 = vgh_fmt_q8_kind(ib.is_f32);
    a->out_fp8 = vgh_fmt_q8_kind(ob.is_f32);
    a->gscale = op.gscale;
    a->dvec = op.dvec;
    VGH_REQUIRE(a->out_f32 || ob.is_f32 == ib.is_f32 || (in8 && ob.is_f32 == VGH_FMT_BF16) || (out8 && ib.is_f32 == VGH_FMT_BF16), "net: op writes buffer %d in another format than it reads", d.out_buf);
    VGH_REQUIRE(d.res_buf < 0 || n->bufs[d.res_buf].is_f32 == (in8 ? VGH_FMT_BF16 : ib.is_f32), "net: residual buffer %d has another format than the input", d.res_buf);
    VGH_REQUIRE(!(in8 || out8) || (d.ksize == 3 && d.stride == 1 && !d.grp_cout && !d.shuffle && (!in8 || d.cin % 64 == 0)), "net: an e4m3 buffer can only link plain 3x3 / stride-1 convs (cin %% 64 == 0)");
    a->res_coff = d.res_coff;
    a->alpha = d.alpha;
    a->act = d.act;
    a->shuffle = d.shuffle;
    a->shuffle_c = d.shuffle ? d.cout_pad / 4 : 0;
    a->zeros = n->zeros;
    a->P = B * a->Ho * a->Wo;
    a->cblocks = d.cin / (in8 ? 64 : 32);  // 64-byte channel blocks
    a->nkb = d.ksize * d.ksize * a->cblocks;
    const int eh = d.shuffle ? 2 * a->Ho : a->Ho, ew = d.shuffle ? 2 * a->Wo : a->Wo;
    VGH_REQUIRE(ob.h == eh && ob.w == ew, "net: op output buffer %d is %dx%d, conv produces %dx%d", d.out_buf, ob.h, ob.w, eh, ew);
    VGH_REQUIRE(op.overhang_ok || d.in_coff + d.cin + (d.grp_cout ? (d.cout_pad / d.grp_cout - 1) * d.grp_in_stride : 0) <= ib.pitch, "net: conv reads past the input pitch (buf %d)", d.in_buf);
    return VGH_OK;
}

static void net_b2b_fields(ConvArgs& a, const ConvArgs& a2) {  // the second conv of a back-to-back pair rides in the first one's descriptor
    a.w2pack = a2.wpack;
    a.bias2 = a2.bias;
    a.out2 = a2.out;
    a.out2_pitch = a2.out_pitch;
    a.out2_coff = a2.out_coff;
    a.out2_coff2 = a2.out_coff2;
    a.out2_split = a2.out_split;
    a.cout2_pad = a2.cout_pad;
    a.cout2_store = a2.cout_store;
    a.act2 = a2.act;
}

// runs one op for the `B` images starting at batch row `at` (image pointer and every activation buffer offset accordingly)
static int net_run_op(vgh_net* n, const NetOp& op, const void* image0, int fmt, int B, int at, hipStream_t st, int share = 1) {
    const vgh_op_desc& d = op.d;
    const void* image = (const char*)image0 + (int64_t)at * n->image_size * n->image_size * 3 * (fmt == VGH_IMG_F32_NCHW ? 4 : 1);
    auto bp = [&](int id) { return (char*)n->buf_ptr[id] + at * buf_image_bytes(n->bufs[id]); };
    const int op_index = (int)(&op - n->ops.data());
    const bool fused = n->fuse_stem && n->stem_pair >= 0;
    // stem + downsample + conv1|conv2 as one launch (r06, ds_b2b.hip): u8 images, the default b2b mode, the pair on its persistent tile
    // (u8 images, default mode: the stem is then a bf16 x 3 split GEMM on the matrix cores -- exact products, another fp32 summation order: a flipped bf16 ulp in ~4e-5 of the
    // stem values; vgh_net_set_b2b(n, 3) keeps the stem launch and with it the bit-identity of every mode)
    const bool stem3 = n->fuse_b2b == 1 && !fused && n->stem_pair >= 0 && fmt == VGH_IMG_U8_NHWC && n->stem3_ok == 1 && n->ops[n->stem_pair + 1].b2b == 1;
    switch (d.kind) {
        case VGH_OP_STEM: {
            if (stem3 && op_index == n->stem_pair) {
                const NetOp &ds = n->ops[op_index + 1], &nx = n->ops[op_index + 2];
                ConvArgs a, a2;
                if (int rc = net_conv_args(n, ds, B, at, &a)) return rc;
                if (int rc = net_conv_args(n, nx, B, at, &a2)) return rc;
                a.grid_share = share;
                net_b2b_fields(a, a2);
                return vgh_launch_stem_ds_b2b(a, image, n->image_size, n->image_size, op.wf32, op.bias, st);
            }
#ifdef VGH_EXPERIMENTS  // (stem_ds.hip is part of the experiments build only)
            if (fused && op_index == n->stem_pair) {
                const NetOp& ds = n->ops[op_index + 1];
                const vgh_buf_desc& db = n->bufs[ds.d.out_buf];
                return vgh_launch_stem_ds(image, fmt, B, n->image_size, n->image_size, op.wf32, op.bias, ds.wds, ds.bias, (uint16_t*)bp(ds.d.out_buf), db.pitch, ds.d.out_coff, st);
            }
#endif
            const vgh_buf_desc& ob = n->bufs[d.out_buf];
            if (ob.is_f32 == VGH_FMT_F32)  // fp32 parity mode
                return vgh_launch_stem_f32(image, fmt, B, n->image_size, n->image_size, op.wf32, op.bias, (float*)bp(d.out_buf), ob.pitch, d.out_coff, st);
            if (ob.is_f32 == VGH_FMT_F16)  // single-plane fp16: the fp16 split stem with plane stride 0 (split_store then writes the hi plane only)
                return vgh_launch_stem(image, fmt, B, n->image_size, n->image_size, op.wf32, op.bias, (uint16_t*)bp(d.out_buf), ob.pitch, d.out_coff, d.cout_store, VGH_FMT_F16X2, 0, st);
            return vgh_launch_stem(image, fmt, B, n->image_size, n->image_size, op.wf32, op.bias, (uint16_t*)bp(d.out_buf), (int64_t)ob.pitch * vgh_fmt_planes(ob.is_f32), d.out_coff,
                                   d.cout_store, ob.is_f32, ob.pitch, st);
        }
        case VGH_OP_CONV: {
            if ((fused || stem3) && op_index == n->stem_pair + 1) return VGH_OK;  // ran inside the stem's launch
            if (n->fuse_b2b && op.b2b == 2) return VGH_OK;             // ran inside the previous conv's launch
            ConvArgs a;
            if (int rc = net_conv_args(n, op, B, at, &a)) return rc;
            a.grid_share = share;
            if (n->fuse_b2b && op.b2b == 1) {
                const NetOp& nx = n->ops[op_index + 1];
                ConvArgs a2;
                if (int rc = net_conv_args(n, nx, B, at, &a2)) return rc;
                net_b2b_fields(a, a2);
                a.b2b_igemm = n->fuse_b2b == 2;
                return vgh_launch_conv_b2b(a, st);
            }
            if (n->bufs[d.in_buf].is_f32 != VGH_FMT_F32) {
                a.fallback_cfg1 = op.auto_cfg + 1;
                return vgh_launch_conv(a, d.force_cfg >= 0 ? d.force_cfg : op.auto_cfg, st);
            }
            if (n->bufs[d.in_buf].is_f32 == VGH_FMT_F32) {  // fp32 parity mode: dense fp32 weights, FMA kernel
                VGH_REQUIRE(a.out_f32 && (d.res_buf < 0 || n->bufs[d.res_buf].is_f32 == VGH_FMT_F32), "net: fp32 conv needs fp32 output / residual buffers");
                VGH_REQUIRE(!d.grp_cout, "net: the fp32 FMA kernel has no grouped mode");
                return vgh_launch_conv_f32(a, op.wf32, st);
            }
            return vgh_launch_conv(a, d.force_cfg, st);
        }
        case VGH_OP_SPP_POOL: {
            const vgh_buf_desc& ib = n->bufs[d.in_buf];
            if (ib.is_f32 == VGH_FMT_F32) return vgh_launch_spp_pool_f32((float*)bp(d.in_buf), ib.pitch, d.in_coff, d.cin, B, ib.h, ib.w, st);
            if (ib.is_f32 == VGH_FMT_F16) return vgh_launch_spp_pool((uint16_t*)bp(d.in_buf), ib.pitch, d.in_coff, d.cin, B, ib.h, ib.w, VGH_FMT_F16X2, 0, st);  // plane stride 0: single plane
            return vgh_launch_spp_pool((uint16_t*)bp(d.in_buf), (int64_t)ib.pitch * vgh_fmt_planes(ib.is_f32), d.in_coff, d.cin, B, ib.h, ib.w, ib.is_f32, ib.pitch, st);
        }
        case VGH_OP_FORK:
            return VGH_OK;  // handled by the executor
        default:
            VGH_REQUIRE(false, "net: unknown op kind %d", d.kind);
    }
    return VGH_OK;
}

// Lane streams that are measured to overlap with the caller's stream and with each other (streams.hip: a fresh hipStreamCreate may
// share a hardware queue with `main`, which serialises the lanes and costs ~40 % of the forward).  Picked on the first forward that
// enters on `main`; a different caller stream re-picks.  Not callable while `main` is capturing (vgh_net_capture picks first).
static int ensure_lanes(vgh_net* n, hipStream_t main) {
    if (n->lanes_ready && n->lanes_main == main) return VGH_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(main, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
        VGH_REQUIRE(n->lanes_ready, "net: the lane streams cannot be picked while the caller's stream is capturing; run vgh_net_capture (or one forward) first");
        return VGH_OK;  // captured with the lanes picked for another caller stream: still correct, possibly not side by side
    }
    hipStream_t avoid[vgh_net::kLanes] = {main};
    for (int l = 1; l < vgh_net::kLanes; ++l) {
        if (n->side[l]) {
            VGH_HIP(hipStreamSynchronize(n->side[l]));
            vgh_stream_release_internal(n->device, n->side[l]);
            n->side[l] = nullptr;
        }
    }
    for (int l = 1; l < vgh_net::kLanes; ++l) {
        if (int rc = vgh_stream_acquire_internal(n->device, avoid, l, &n->side[l])) return rc;  // same priority as the caller's stream: a lower-priority lane starves (16.6 vs 13.6 ms)
        avoid[l] = n->side[l];
    }
    n->lanes_main = main;
    n->lanes_ready = true;
    return VGH_OK;
}

int vgh_net_lane_streams(vgh_net* n, hipStream_t main, hipStream_t* out) {
    if (int rc = ensure_lanes(n, main)) return rc;
    for (int l = 1; l < vgh_net::kLanes; ++l) out[l - 1] = n->side[l];
    return VGH_OK;
}

// The batch as nsplit independent sub-batches, one per lane stream (lane 0 = the caller's stream); launches are interleaved
// op by op.  lane_lag = 0: the lanes advance together.  lane_lag = k > 0: lane l starts when lane l-1 has finished its first k ops and
// stays k ops behind, so that the kernels running side by side are DIFFERENT layers (an HBM-bound 1x1 conv of one lane next to an
// MFMA-bound 3x3 conv of the other) instead of two copies of the same one.
static std::atomic<int> g_lane_lag{0};

static int net_forward_split(vgh_net* n, const void* image_dev, int image_fmt, int B, hipStream_t main) {
    const int L = n->nsplit < B ? n->nsplit : B;
    int at[vgh_net::kLanes + 1];
    at[0] = 0;
    for (int l = 0; l < L; ++l) at[l + 1] = at[l] + B / L + (l < B % L ? 1 : 0);
    VGH_HIP(hipEventRecord(n->ev_fork, main));
    for (int l = 1; l < L; ++l) VGH_HIP(hipStreamWaitEvent(n->side[l], n->ev_fork, 0));
    std::vector<const NetOp*> seq;
    seq.reserve(n->ops.size());
    for (const NetOp& op : n->ops)
        if (op.d.kind != VGH_OP_FORK) seq.push_back(&op);  // head lanes are not combined with the batch split
    const int N = (int)seq.size();
    int lag = g_lane_lag.load(std::memory_order_relaxed);
    if (lag < 0) lag = 0;
    if (lag > N) lag = N;
    bool guard_pending[vgh_net::kLanes];
    for (int l = 0; l < L; ++l) guard_pending[l] = n->pred_guard != nullptr;
    for (int s = 0; s < N + lag * (L - 1); ++s) {
        for (int l = 0; l < L; ++l) {
            const int i = s - l * lag;
            if (i < 0 || i >= N) continue;
            const NetOp& op = *seq[i];
            hipStream_t st = l == 0 ? main : n->side[l];
            if (lag > 0 && l > 0 && i == 0) VGH_HIP(hipStreamWaitEvent(st, n->ev_lag[l - 1], 0));  // recorded below, `lag` ops into lane l-1
            if (guard_pending[l] && op.d.kind == VGH_OP_CONV && n->bufs[op.d.out_buf].is_f32 == VGH_FMT_F32) {
                VGH_HIP(hipStreamWaitEvent(st, n->pred_guard, 0));
                guard_pending[l] = false;
            }
            if (int rc = net_run_op(n, op, image_dev, image_fmt, at[l + 1] - at[l], at[l], st)) return rc;
            if (lag > 0 && l + 1 < L && i == lag - 1) VGH_HIP(hipEventRecord(n->ev_lag[l], st));
        }
    }
    for (int l = 1; l < L; ++l) {
        VGH_HIP(hipEventRecord(n->ev_join[l], n->side[l]));
        VGH_HIP(hipStreamWaitEvent(main, n->ev_join[l], 0));
    }
    return VGH_OK;
}

extern "C" {

const char* vgh_version(void) { return "vgh 0.1.0 (gfx950)"; }
const char* vgh_last_error(void) { return g_err; }
int vgh_abi_version(void) { return VGH_ABI_VERSION; }

int vgh_net_create(int device, int image_size, int max_batch, const vgh_buf_desc* bufs, int n_bufs, const vgh_op_desc* ops, int n_ops,
                   const float* weights_host, int64_t n_weights, const float* biases_host, int64_t n_biases, vgh_net** out) {
    VGH_REQUIRE(out && bufs && ops && weights_host && biases_host, "net_create: null argument");
    VGH_REQUIRE(max_batch >= 1 && image_size >= 32 && image_size % 32 == 0, "net_create: bad max_batch / image_size");
    VGH_HIP(hipSetDevice(device));
    vgh_net* n = new vgh_net();
    n->device = device;
    n->image_size = image_size;
    n->max_batch = max_batch;
    n->bufs.assign(bufs, bufs + n_bufs);
    // ---- arena plan: one slab, every buffer live for the whole forward (288 GB HBM: no aliasing games) ----
    int64_t off = 0;
    std::vector<int64_t> offs(n_bufs);
    for (int i = 0; i < n_bufs; ++i) {
        VGH_REQUIRE(bufs[i].is_f32 >= VGH_FMT_BF16 && bufs[i].is_f32 <= VGH_FMT_I8, "net_create: buffer %d has unknown format %d", i, bufs[i].is_f32);
        VGH_REQUIRE(!vgh_fmt_is_q8(bufs[i].is_f32) || (bufs[i].scale > 0.0f && bufs[i].scale < 1e30f && bufs[i].pitch % 16 == 0), "net_create: 8-bit buffer %d needs a positive scale and a pitch that is a multiple of 16", i);
        const int64_t bytes = (int64_t)max_batch * bufs[i].h * bufs[i].w * bufs[i].pitch * vgh_fmt_bytes(bufs[i].is_f32);
        offs[i] = off;
        n->buf_bytes.push_back(bytes);
        off += align_up(bytes + 256, 256);  // +256: slack so 16-byte gathers at the very end stay in-bounds
    }
    n->arena_bytes = off;
    VGH_HIP(hipMalloc((void**)&n->arena, off));
    VGH_HIP(hipMemset(n->arena, 0, off));  // padding channels that no op writes must be exact zeros
    for (int i = 0; i < n_bufs; ++i) n->buf_ptr.push_back(n->arena + offs[i]);
    VGH_HIP(hipMalloc((void**)&n->zeros, 256));
    VGH_HIP(hipMemset(n->zeros, 0, 256));
    VGH_HIP(hipEventCreateWithFlags(&n->ev_fork, hipEventDisableTiming));
    for (int l = 1; l < vgh_net::kLanes; ++l) VGH_HIP(hipEventCreateWithFlags(&n->ev_join[l], hipEventDisableTiming));  // lane streams: ensure_lanes
    for (int l = 0; l < vgh_net::kLanes; ++l) VGH_HIP(hipEventCreateWithFlags(&n->ev_lag[l], hipEventDisableTiming));
    // ---- weights: pack on the host, one upload ----
    int64_t wbytes = 0;
    std::vector<int64_t> woff(n_ops, 0), boff(n_ops, 0), goff(n_ops, -1), doff(n_ops, -1);
    for (int i = 0; i < n_ops; ++i) {
        const vgh_op_desc& d = ops[i];
        if (d.kind == VGH_OP_CONV) {
            VGH_REQUIRE(d.cin % 32 == 0 && d.cout_pad % 32 == 0, "net_create: op %d channel padding", i);
            VGH_REQUIRE(d.in_buf >= 0 && d.in_buf < n_bufs && d.out_buf >= 0 && d.out_buf < n_bufs && d.res_buf < n_bufs, "net_create: op %d buffer ids", i);
            const int64_t we = (int64_t)d.cout_pad * d.ksize * d.ksize * d.cin;
            VGH_REQUIRE(d.w_off >= 0 && d.w_off + we <= n_weights && d.b_off >= 0 && d.b_off + d.cout_pad <= n_biases, "net_create: op %d weight range", i);
            if (d.in_coff + d.cin > bufs[d.in_buf].pitch && !d.grp_cout) {
                // a K window wider than the pitch is legal only over all-zero weight columns, in the single-plane bf16 format (a split pixel's lo plane follows its hi plane)
                const int live = bufs[d.in_buf].pitch - d.in_coff;
                VGH_REQUIRE(live > 0 && bufs[d.in_buf].is_f32 == VGH_FMT_BF16, "net_create: op %d reads past the pitch of buffer %d", i, d.in_buf);
                const float* w = weights_host + d.w_off;
                for (int64_t r = 0; r < (int64_t)d.cout_pad * d.ksize * d.ksize; ++r)
                    for (int c = live; c < d.cin; ++c)
                        VGH_REQUIRE(w[r * d.cin + c] == 0.0f, "net_create: op %d reads %d channels of a %d-channel pitch with a non-zero weight at input channel %d", i, d.cin, bufs[d.in_buf].pitch, c);
            }
            woff[i] = wbytes;
            const int wf = bufs[d.in_buf].is_f32;  // weight image: bf16 (2 B), dense fp32 (4 B) or the three 16-bit segments of the split modes (6 B)
            wbytes += align_up(we * (vgh_fmt_is_q8(wf) ? 1 : (wf == VGH_FMT_BF16 || wf == VGH_FMT_F16) ? 2 : wf == VGH_FMT_F32 ? 4 : 6), 256);
            boff[i] = wbytes;
            wbytes += align_up((int64_t)d.cout_pad * 4, 256);
            if (vgh_fmt_is_q8(wf) || vgh_fmt_is_q8(bufs[d.out_buf].is_f32)) {
                goff[i] = wbytes;
                wbytes += align_up((int64_t)d.cout_pad * 4, 256);
                if (wf == VGH_FMT_I8 && bufs[d.out_buf].is_f32 == VGH_FMT_BF16 && i8_peel_diag(weights_host + d.w_off, d)) {
                    doff[i] = wbytes;
                    wbytes += align_up((int64_t)d.cout_pad * 4, 256);
                }
            }
        } else if (d.kind == VGH_OP_STEM) {
            VGH_REQUIRE(d.w_off >= 0 && d.w_off + 27 * 48 <= n_weights && d.b_off + 48 <= n_biases, "net_create: stem weight range");
            woff[i] = wbytes;
            wbytes += align_up(27 * 48 * 4, 256);
            boff[i] = wbytes;
            wbytes += align_up(64 * 4, 256);
        }
    }
    // stem + stage-1 downsample pair (see vgh_net::stem_pair)
    int64_t wds_off = -1;
    const int n_pair_scan = n_ops;  // (the pair is what the "u" tile of ds_b2b.hip fuses; the experiments build's stem_ds.hip needs a second weight image on top)
    for (int i = 0; i + 1 < n_pair_scan; ++i) {
        const vgh_op_desc &a = ops[i], &d = ops[i + 1];
        if (a.kind != VGH_OP_STEM || d.kind != VGH_OP_CONV) continue;
        bool ok = image_size % 4 == 0 /* vgh_launch_stem_ds tiles 4 x 16 outputs of the 1/4-resolution map */ && d.in_buf == a.out_buf && d.in_coff == a.out_coff && d.ksize == 3 && d.stride == 2 && d.cin == 64 && d.cout_pad == 96 && d.cout_store == 96 && d.res_buf < 0 && !d.shuffle &&
                  d.out_split >= 96 && d.act == VGH_ACT_RELU && d.grp_cout == 0 && bufs[a.out_buf].is_f32 == VGH_FMT_BF16 && bufs[d.out_buf].is_f32 == VGH_FMT_BF16 && d.out_coff % 8 == 0 &&
                  bufs[d.out_buf].pitch % 8 == 0;
        for (int j = 0; ok && j < n_ops; ++j)  // nobody else may read (or write) the stem tensor
            if (j != i && j != i + 1 && (ops[j].kind == VGH_OP_CONV || ops[j].kind == VGH_OP_SPP_POOL) && (ops[j].in_buf == a.out_buf || ops[j].out_buf == a.out_buf || ops[j].res_buf == a.out_buf)) ok = false;
        if (ok) {
            n->stem_pair = i;
#ifdef VGH_EXPERIMENTS
            wds_off = wbytes;
            wbytes += align_up((int64_t)9 * 3 * 96 * 16 * 2, 256);
#endif
        }
        break;
    }
    std::vector<char> host(wbytes > 0 ? wbytes : 1, 0);
    std::vector<float> oscale(n_ops, 1.0f);
#ifdef VGH_EXPERIMENTS
    if (n->stem_pair >= 0) vgh_pack_stem_ds_weights_host(weights_host + ops[n->stem_pair + 1].w_off, (uint16_t*)(host.data() + wds_off));
#endif
    for (int i = 0; i < n_ops; ++i) {
        const vgh_op_desc& d = ops[i];
        if (d.kind == VGH_OP_CONV) {
            const int wf = bufs[d.in_buf].is_f32;
            if (wf == VGH_FMT_F32)
                memcpy(host.data() + woff[i], weights_host + d.w_off, (size_t)d.cout_pad * d.ksize * d.ksize * d.cin * 4);
            else if (wf == VGH_FMT_BF16)
                vgh_pack_conv_weights_host(weights_host + d.w_off, d.cout_pad, d.ksize, d.cin, (uint16_t*)(host.data() + woff[i]));
            else if (vgh_fmt_is_q8(wf))
                ;  // below: the 8-bit image comes with per-cout scales that also enter the bias
            else
                vgh_pack_conv_weights_split_host(weights_host + d.w_off, d.cout_pad, d.ksize, d.cin, wf, (uint16_t*)(host.data() + woff[i]), &oscale[i]);
            memcpy(host.data() + boff[i], biases_host + d.b_off, (size_t)d.cout_pad * 4);
            if (goff[i] >= 0) {
                // e4m3 links: out = act(acc * g + bias) with g[c] = wscale[c] * scale(in) for an e4m3 input (1 for bf16), divided by scale(out) for an e4m3 output.
                // The kernel starts its accumulator at the bias, so the bias is stored in accumulator units: bias / (wscale[c] * scale(in)).
                float* g = (float*)(host.data() + goff[i]);
                float* bq = (float*)(host.data() + boff[i]);
                const float s_out = vgh_fmt_is_q8(bufs[d.out_buf].is_f32) ? bufs[d.out_buf].scale : 1.0f;
                if (wf == VGH_FMT_I8) {
                    // int8: exact int32 accumulator that starts at the bias in accumulator units (int32 bit patterns in the bias vector); out = act(acc * g) with
                    // g = wscale[c] * scale(in) / scale(out)
                    std::vector<float> ws(d.cout_pad);
                    const float* wsrc = weights_host + d.w_off;
                    std::vector<float> peeled;
                    if (doff[i] >= 0) {  // diagonal bypass: w[c][centre][c] leaves the int8 image and is applied in fp32 by the epilogue (s_out = 1: bf16 output)
                        peeled.assign(wsrc, wsrc + (size_t)d.cout_pad * 9 * d.cin);
                        float* dv = (float*)(host.data() + doff[i]);
                        for (int c = 0; c < d.cout_pad; ++c) {
                            float& wd = peeled[(size_t)c * 9 * d.cin + (size_t)4 * d.cin + c];
                            dv[c] = wd * bufs[d.in_buf].scale;
                            wd = 0.0f;
                        }
                        wsrc = peeled.data();
                    }
                    vgh_pack_conv_weights_i8_host(wsrc, d.cout_pad, d.ksize, d.cin, (uint8_t*)(host.data() + woff[i]), ws.data());
                    for (int c = 0; c < d.cout_pad; ++c) {
                        const float acc_unit = ws[c] * bufs[d.in_buf].scale;
                        g[c] = acc_unit / s_out;
                        const float bi = __builtin_nearbyintf(bq[c] / acc_unit);
                        const int32_t b32 = bi != bi ? 0 : bi > 2.0e9f ? 2000000000 : bi < -2.0e9f ? -2000000000 : (int32_t)bi;  // (|bias| / unit is ~1e4 - 1e6 in practice)
                        memcpy(&bq[c], &b32, 4);
                    }
                } else if (wf == VGH_FMT_FP8) {
                    std::vector<float> ws(d.cout_pad);
                    vgh_pack_conv_weights_fp8_host(weights_host + d.w_off, d.cout_pad, d.ksize, d.cin, (uint8_t*)(host.data() + woff[i]), ws.data());
                    for (int c = 0; c < d.cout_pad; ++c) {
                        const float acc_unit = ws[c] * bufs[d.in_buf].scale;
                        bq[c] = bq[c] / acc_unit;
                        g[c] = acc_unit / s_out;
                    }
                } else {
                    for (int c = 0; c < d.cout_pad; ++c) g[c] = 1.0f / s_out;
                }
            }
        } else if (d.kind == VGH_OP_STEM) {
            // host gives [48][3(ky)][3(kx)][3(ci)] -> device [27][48]
            float* dst = (float*)(host.data() + woff[i]);
            for (int co = 0; co < 48; ++co)
                for (int k = 0; k < 27; ++k) dst[k * 48 + co] = weights_host[d.w_off + co * 27 + k];
            memcpy(host.data() + boff[i], biases_host + d.b_off, 48 * 4);
        }
    }
    VGH_HIP(hipMalloc((void**)&n->wblob, host.size()));
    VGH_HIP(hipMemcpy(n->wblob, host.data(), host.size(), hipMemcpyHostToDevice));
    for (int i = 0; i < n_ops; ++i) {
        NetOp op;
        op.d = ops[i];
        op.out_scale = oscale[i];
        if (ops[i].kind == VGH_OP_CONV) {
            op.wpack = (uint16_t*)(n->wblob + woff[i]);
            op.wf32 = (float*)(n->wblob + woff[i]);  // same storage: bf16 image (throughput mode) or dense fp32 (parity mode)
            op.bias = (float*)(n->wblob + boff[i]);
            op.overhang_ok = (!ops[i].grp_cout && ops[i].in_coff + ops[i].cin > bufs[ops[i].in_buf].pitch) ? 1 : 0;  // zero weight columns verified above
            if (goff[i] >= 0) op.gscale = (float*)(n->wblob + goff[i]);
            if (doff[i] >= 0) op.dvec = (float*)(n->wblob + doff[i]);
        } else if (ops[i].kind == VGH_OP_STEM) {
            op.wf32 = (float*)(n->wblob + woff[i]);
            op.bias = (float*)(n->wblob + boff[i]);
        }
        if (n->stem_pair >= 0 && i == n->stem_pair + 1) op.wds = (uint16_t*)(n->wblob + wds_off);
        n->ops.push_back(op);
    }
    // the automatic tile of every 16-bit conv, for the arena batch: one choice per op whatever batch / chunk / lane later runs it (bit-identical results across
    // chunkings), resolved here so that a max_batch whose tensors break the 2 GiB rule of the loaders fails at creation, and no forward ever writes the net
    for (NetOp& op : n->ops) {
        if (op.d.kind != VGH_OP_CONV || n->bufs[op.d.in_buf].is_f32 == VGH_FMT_F32) continue;
        ConvArgs ref;
        int rc = net_conv_args(n, op, n->max_batch, 0, &ref);
        if (!rc) rc = vgh_conv_prepare(ref);
        if (rc) {
            vgh_net_destroy(n);
            return rc;
        }
        op.auto_cfg = vgh_conv_pick_auto(ref);
    }
    // events of the ops that a later op on another lane waits for (arch.schedule_latency)
    for (int i = 0; i < (int)n->ops.size(); ++i) {
        const int dep = (n->ops[i].d.lane >> 8) - 1;
        if (dep < 0) continue;
        if (dep >= i) {
            vgh_net_destroy(n);
            vgh_set_error("net_create: op %d waits for op %d, which does not precede it", i, dep);
            return VGH_ERR_INVALID;
        }
        if (!n->ops[dep].ev_done && hipEventCreateWithFlags(&n->ops[dep].ev_done, hipEventDisableTiming) != hipSuccess) {
            vgh_net_destroy(n);
            vgh_set_error("net_create: hipEventCreate failed");
            return VGH_ERR_HIP;
        }
    }
    {
        bool w[vgh_net::kLanes][vgh_net::kLanes] = {};  // w[a][b]: side lane a waits for an op of side lane b (directly, then transitively)
        for (const NetOp& op : n->ops) {
            const int dep = (op.d.lane >> 8) - 1, a = op.d.lane & 0xff;
            if (dep < 0) continue;
            const int b = n->ops[dep].d.lane & 0xff;
            if (a > 0 && a < vgh_net::kLanes && b > 0 && b < vgh_net::kLanes && a != b) w[a][b] = true;
        }
        for (int k = 1; k < vgh_net::kLanes; ++k)
            for (int a = 1; a < vgh_net::kLanes; ++a)
                for (int b = 1; b < vgh_net::kLanes; ++b) w[a][b] = w[a][b] || (w[a][k] && w[k][b]);
        for (int a = 1; a < vgh_net::kLanes; ++a) n->lane_wait_cycle = n->lane_wait_cycle || w[a][a];
    }
    // back-to-back pairs (r06): a plain bf16 conv whose whole output tensor -- all of its channels in ONE cout tile -- is read by exactly one op, the next one, a plain
    // 1x1 / stride-1 bf16 conv (the architecture's stage downsample -> the CSP layer's merged conv1|conv2); nothing else touches that tensor
    for (int i = 0; i + 1 < (int)n->ops.size(); ++i) {
        const vgh_op_desc &a = n->ops[i].d, &b = n->ops[i + 1].d;
        if (a.kind != VGH_OP_CONV || b.kind != VGH_OP_CONV) continue;
        const vgh_buf_desc &ab = n->bufs[a.out_buf], &ai = n->bufs[a.in_buf], &bo = n->bufs[b.out_buf];
        bool ok = ab.is_f32 == VGH_FMT_BF16 && ai.is_f32 == VGH_FMT_BF16 && bo.is_f32 == VGH_FMT_BF16 && vgh_conv_b2b_ok(a.ksize, a.stride, a.cout_pad, b.cout_pad) && a.cout_store == a.cout_pad &&
                  a.out_coff == 0 && a.out_split >= a.cout_pad && ab.pitch == a.cout_pad && a.res_buf < 0 && !a.shuffle && !a.grp_cout && a.act != VGH_ACT_SILU && b.ksize == 1 && b.stride == 1 &&
                  b.in_buf == a.out_buf && b.in_coff == 0 && b.cin == a.cout_pad && b.res_buf < 0 && !b.shuffle && !b.grp_cout && b.act != VGH_ACT_SILU && b.out_coff % 8 == 0 && b.out_coff2 % 8 == 0 &&
                  b.out_split % 8 == 0 && b.cout_store % 8 == 0 && bo.pitch % 8 == 0 && n->ops[i].b2b == 0;
        for (int j = 0; ok && j < (int)n->ops.size(); ++j) {
            const vgh_op_desc& o = n->ops[j].d;
            if (j != i && j != i + 1 && (o.kind == VGH_OP_CONV || o.kind == VGH_OP_SPP_POOL) && (o.in_buf == a.out_buf || o.out_buf == a.out_buf || o.res_buf == a.out_buf)) ok = false;
        }
        if (ok) {
            n->ops[i].b2b = 1;
            n->ops[i + 1].b2b = 2;
        }
    }
    // the stem + the stage-1 pair as one launch ("u" tile): decided once, from the pair's descriptor
    n->stem3_ok = 0;
    if (n->stem_pair >= 0 && n->stem_pair + 2 < (int)n->ops.size() && n->ops[n->stem_pair + 1].b2b == 1 && image_size % 32 == 0) {
        ConvArgs a, a2;
        if (!net_conv_args(n, n->ops[n->stem_pair + 1], 1, 0, &a) && !net_conv_args(n, n->ops[n->stem_pair + 2], 1, 0, &a2)) {
            net_b2b_fields(a, a2);
            if (!vgh_conv_prepare(a)) n->stem3_ok = vgh_conv_ds_b2b_ok(a) && a.act == VGH_ACT_RELU ? 1 : 0;
        }
    }
    *out = n;
    return VGH_OK;
}

void vgh_net_destroy(vgh_net* n) {
    if (!n) return;
    if (n->graph_exec) hipGraphExecDestroy(n->graph_exec);
    if (n->graph) hipGraphDestroy(n->graph);
    for (int l = 1; l < vgh_net::kLanes; ++l) {
        if (n->side[l]) {
            hipStreamSynchronize(n->side[l]);
            vgh_stream_release_internal(n->device, n->side[l]);
        }
        if (n->ev_join[l]) hipEventDestroy(n->ev_join[l]);
    }
    for (int l = 0; l < vgh_net::kLanes; ++l)
        if (n->ev_lag[l]) hipEventDestroy(n->ev_lag[l]);
    if (n->ev_fork) hipEventDestroy(n->ev_fork);
    for (NetOp& op : n->ops)
        if (op.ev_done) hipEventDestroy(op.ev_done);
    hipFree(n->arena);
    hipFree(n->wblob);
    hipFree(n->zeros);
    delete n;
}

int vgh_net_forward(vgh_net* n, const void* image_dev, int image_fmt, int B, void* stream) {
    VGH_REQUIRE(n && image_dev, "net_forward: null argument");
    VGH_REQUIRE(B >= 0 && B <= n->max_batch, "net_forward: B=%d exceeds max_batch=%d", B, n->max_batch);
    // Independent branches (the three detection heads) run on side streams: a FORK op records an event on the main stream,
    // the first op of a lane after it makes that lane's stream wait for the event, and every used lane is joined back into
    // the main stream at the end (also valid under stream capture: the graph gets parallel branches).
    hipStream_t main = (hipStream_t)stream;
    if (int rc = ensure_lanes(n, main)) return rc;
    if (n->nsplit > 1 && B > 1) return net_forward_split(n, image_dev, image_fmt, B, main);
    bool pending[vgh_net::kLanes] = {false, false, false, false}, used[vgh_net::kLanes] = {false, false, false, false};
    bool guard_pending = n->pred_guard != nullptr;
    for (const NetOp& op : n->ops) {
        if (guard_pending && op.d.kind == VGH_OP_CONV && n->bufs[op.d.out_buf].is_f32 == VGH_FMT_F32) {
            // every stream that may run a prediction conv waits (main; the lanes inherit it through the fork / their own wait)
            VGH_HIP(hipStreamWaitEvent(main, n->pred_guard, 0));
            for (int l = 1; l < vgh_net::kLanes; ++l)
                if (n->side[l]) VGH_HIP(hipStreamWaitEvent(n->side[l], n->pred_guard, 0));
            guard_pending = false;
        }
        if (op.d.kind == VGH_OP_FORK) {
            VGH_HIP(hipEventRecord(n->ev_fork, main));
            for (int l = 1; l < vgh_net::kLanes; ++l) pending[l] = true;
            continue;
        }
        const int lane_f = op.d.lane & 0xff, dep = (op.d.lane >> 8) - 1;
        const int lane = (lane_f > 0 && lane_f < vgh_net::kLanes) ? lane_f : 0;
        hipStream_t st = main;
        if (lane > 0) {
            st = n->side[lane];
            if (pending[lane]) {
                VGH_HIP(hipStreamWaitEvent(st, n->ev_fork, 0));
                pending[lane] = false;
            }
            used[lane] = true;
        }
        if (dep >= 0 && dep < (int)n->ops.size() && n->ops[dep].ev_done) VGH_HIP(hipStreamWaitEvent(st, n->ops[dep].ev_done, 0));  // the ONE op this op waits for, on another lane
        if (int rc = net_run_op(n, op, image_dev, image_fmt, B, 0, st)) return rc;
        if (op.ev_done) VGH_HIP(hipEventRecord(op.ev_done, st));
    }
    for (int l = 1; l < vgh_net::kLanes; ++l)
        if (used[l]) {
            VGH_HIP(hipEventRecord(n->ev_join[l], n->side[l]));
            VGH_HIP(hipStreamWaitEvent(main, n->ev_join[l], 0));
        }
    return VGH_OK;
}

int vgh_net_profile(vgh_net* n, const void* image_dev, int image_fmt, int B, void* stream, float* op_ms) {
    VGH_REQUIRE(n && image_dev && op_ms, "net_profile: null argument");
    VGH_REQUIRE(B >= 0 && B <= n->max_batch, "net_profile: B=%d exceeds max_batch=%d", B, n->max_batch);
    hipStream_t st = (hipStream_t)stream;
    if (int rc = ensure_lanes(n, st)) return rc;
    const size_t m = n->ops.size();
    std::vector<hipEvent_t> ev(m + 1);
    for (auto& e : ev) VGH_HIP(hipEventCreate(&e));
    VGH_HIP(hipEventRecord(ev[0], st));
    const int L = (n->nsplit > 1 && B > 1) ? (n->nsplit < B ? n->nsplit : B) : 1;
    for (size_t i = 0; i < m; ++i) {
        if (L == 1) {
            if (int rc = net_run_op(n, n->ops[i], image_dev, image_fmt, B, 0, st)) return rc;
        } else {
            // batch-split mode: the op's sub-batches run concurrently on the lane streams, as in vgh_net_forward
            VGH_HIP(hipEventRecord(n->ev_fork, st));
            int at = 0;
            for (int l = 0; l < L; ++l) {
                const int nb = B / L + (l < B % L ? 1 : 0);
                hipStream_t ls = l == 0 ? st : n->side[l];
                if (l > 0) VGH_HIP(hipStreamWaitEvent(ls, n->ev_fork, 0));
                if (int rc = net_run_op(n, n->ops[i], image_dev, image_fmt, nb, at, ls)) return rc;
                if (l > 0) {
                    VGH_HIP(hipEventRecord(n->ev_join[l], ls));
                    VGH_HIP(hipStreamWaitEvent(st, n->ev_join[l], 0));
                }
                at += nb;
            }
        }
        VGH_HIP(hipEventRecord(ev[i + 1], st));
    }
    VGH_HIP(hipEventSynchronize(ev[m]));
    for (size_t i = 0; i < m; ++i) VGH_HIP(hipEventElapsedTime(&op_ms[i], ev[i], ev[i + 1]));
    for (auto& e : ev) hipEventDestroy(e);
    return VGH_OK;
}

int vgh_net_capture(vgh_net* n, const void* image_dev, int image_fmt, int B, void* stream) {
    VGH_REQUIRE(n && stream, "net_capture: needs a non-null stream");
    VGH_REQUIRE(!n->pred_guard, "net_capture: a prediction guard event is set (detector overlap mode); graph replay cannot honour it");
    VGH_REQUIRE(!n->lane_wait_cycle, "net_capture: two side lanes of this program wait for each other's ops; the HIP runtime cannot end the capture of such a program (run it eagerly, "
                "or lay the lanes out so that the wait-for relation among lanes 1..3 is acyclic, as arch.schedule_latency does)");
    hipStream_t st = (hipStream_t)stream;
    if (n->graph_exec) {
        hipGraphExecDestroy(n->graph_exec);
        n->graph_exec = nullptr;
    }
    if (n->graph) {
        hipGraphDestroy(n->graph);
        n->graph = nullptr;
    }
    if (int rc = ensure_lanes(n, st)) return rc;  // probing launches kernels: before the capture begins
    VGH_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    int rc = vgh_net_forward(n, image_dev, image_fmt, B, stream);
    hipError_t e = hipStreamEndCapture(st, &n->graph);
    if (rc) return rc;
    VGH_HIP(e);
    VGH_HIP(hipGraphInstantiate(&n->graph_exec, n->graph, nullptr, nullptr, 0));
    return VGH_OK;
}

int vgh_net_forward_graph(vgh_net* n, void* stream) {
    VGH_REQUIRE(n && n->graph_exec, "net_forward_graph: call vgh_net_capture first");
    VGH_REQUIRE(!n->pred_guard, "net_forward_graph: a prediction guard event is set (detector overlap mode); use vgh_net_forward");
    VGH_HIP(hipGraphLaunch(n->graph_exec, (hipStream_t)stream));
    return VGH_OK;
}

void* vgh_net_buffer(vgh_net* n, int buf_id) { return (n && buf_id >= 0 && buf_id < (int)n->buf_ptr.size()) ? n->buf_ptr[buf_id] : nullptr; }
int64_t vgh_net_buffer_bytes(vgh_net* n, int buf_id) { return (n && buf_id >= 0 && buf_id < (int)n->buf_bytes.size()) ? n->buf_bytes[buf_id] : -1; }

#ifdef VGH_EXPERIMENTS
int vgh_net_set_lane_lag(int ops) {
    VGH_REQUIRE(ops >= 0, "net_set_lane_lag: negative");
    g_lane_lag.store(ops, std::memory_order_relaxed);
    return VGH_OK;
}
#endif

int vgh_net_set_split(vgh_net* n, int nsplit) {
    VGH_REQUIRE(n, "net_set_split: null handle");
    VGH_REQUIRE(nsplit >= 1 && nsplit <= vgh_net::kLanes, "net_set_split: 1..%d lanes", vgh_net::kLanes);
    n->nsplit = nsplit;
    return VGH_OK;
}

#ifdef VGH_EXPERIMENTS
int vgh_net_set_fuse_stem(vgh_net* n, int enable) {
    VGH_REQUIRE(n, "net_set_fuse_stem: null handle");
    n->fuse_stem = enable ? 1 : 0;
    return VGH_OK;
}
#endif

int vgh_net_set_pred_guard(vgh_net* n, void* event) {
    VGH_REQUIRE(n, "net_set_pred_guard: null handle");
    n->pred_guard = (hipEvent_t)event;
    return VGH_OK;
}

// Back-to-back GEMM pairs (a stage's downsample + the CSP layer's conv1|conv2 behind it) as one launch each (default) or as two: the outputs are the same bits; unfused,
// every intermediate tensor of the program exists in the arena (what the per-op parity tests read)
int vgh_net_set_b2b(vgh_net* n, int enable) {
    VGH_REQUIRE(n, "net_set_b2b: null handle");
    n->fuse_b2b = (enable == 2 || enable == 3) ? enable : enable ? 1 : 0;  // 1: pairs fused, the stem conv inside the stage-1 pair's launch for u8 images; 3: without the stem; 2: every pair on the implicit-GEMM b2b tile
    return VGH_OK;
}
int vgh_net_stem_fused(vgh_net* n) {  // 1: forwards of u8 images run the stem conv inside the stage-1 pair's launch (the stem tensor is not written)
    return (n && n->fuse_b2b == 1 && n->stem_pair >= 0 && n->stem3_ok == 1 && n->ops[n->stem_pair + 1].b2b == 1) ? 1 : 0;
}
int vgh_net_b2b_pairs(vgh_net* n) {
    int k = 0;
    if (n)
        for (const NetOp& op : n->ops) k += op.b2b == 1;
    return k;
}

int vgh_net_max_batch(vgh_net* n) { return n ? n->max_batch : 0; }
int vgh_net_device(vgh_net* n) { return n ? n->device : 0; }
int vgh_net_image_size(vgh_net* n) { return n ? n->image_size : 0; }

int vgh_net_set_cfg(vgh_net* n, int op_index, int cfg) {
    VGH_REQUIRE(n && op_index >= 0 && op_index < (int)n->ops.size(), "net_set_cfg: bad op index");
    const int fmt0 = n->bufs[n->ops[op_index].d.kind == VGH_OP_CONV ? n->ops[op_index].d.in_buf : 0].is_f32;
    const bool split_net = vgh_fmt_planes(fmt0) > 1 || fmt0 == VGH_FMT_F16;  // the single-plane fp16 nets index the split modes' tile table too
    VGH_REQUIRE(cfg >= -1 && cfg < (split_net ? vgh_conv_split_num_cfgs() : vgh_conv_num_cfgs()), "net_set_cfg: bad cfg");
    NetOp& op = n->ops[op_index];
    if (cfg >= 0 && op.d.kind == VGH_OP_CONV && !split_net && (n->bufs[op.d.in_buf].is_f32 == VGH_FMT_BF16 || vgh_fmt_is_q8(n->bufs[op.d.in_buf].is_f32))) {
        // eligibility is decided ONCE, for the arena batch: a tile that fits a small chunk but not max_batch (the ping-pong tiles' 2 GiB rule depends on the
        // pixel count) would otherwise run some batch sizes and silently fall back on others -- two summation orders for one op, against the "same bits for
        // any chunk" invariant of NetOp::auto_cfg.  Such a tile is replaced by the automatic one for every batch, and the replacement is logged.
        ConvArgs ref;
        int rc = net_conv_args(n, op, n->max_batch, 0, &ref);
        if (!rc) rc = vgh_conv_prepare(ref);
        if (rc) return rc;
        if (!vgh_conv_cfg_ok_for(cfg, ref)) {
            fprintf(stderr, "[vgh] net_set_cfg: op %d: tile %s cannot run this op at max_batch=%d; its automatic tile %s runs it at every batch size\n", op_index, vgh_conv_cfg_name(cfg),
                    n->max_batch, vgh_conv_cfg_name(op.auto_cfg));
            cfg = -1;
        }
    }
    op.d.force_cfg = cfg;
    return VGH_OK;
}

static uint16_t* g_zeros[16] = {nullptr};

int vgh_conv2d(const vgh_conv_call* c, void* stream) {
    VGH_REQUIRE(c && c->in_dev && c->wpack_dev && c->bias_dev && c->out_dev, "conv2d: null argument");
    int dev = 0;
    VGH_HIP(hipGetDevice(&dev));
    VGH_REQUIRE(dev >= 0 && dev < 16, "conv2d: device index");
    if (!g_zeros[dev]) {
        VGH_HIP(hipMalloc((void**)&g_zeros[dev], 256));
        VGH_HIP(hipMemset(g_zeros[dev], 0, 256));
    }
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    const int planes = vgh_fmt_planes(c->fmt);
    VGH_REQUIRE(c->fmt == VGH_FMT_BF16 || c->fmt == VGH_FMT_BF16X2 || c->fmt == VGH_FMT_F16X2 || c->fmt == VGH_FMT_FP8 || c->fmt == VGH_FMT_F16 || c->fmt == VGH_FMT_I8, "conv2d: fmt %d", c->fmt);
    a.in_fp8 = vgh_fmt_q8_kind(c->fmt);
    a.out_fp8 = c->out_fp8;
    VGH_REQUIRE(c->out_fp8 >= 0 && c->out_fp8 <= 2 && (!a.in_fp8 || !a.out_fp8 || a.in_fp8 == a.out_fp8), "conv2d: out_fp8 is 0, 1 (e4m3) or 2 (int8), and an 8-bit input keeps its format");
    a.gscale = c->gscale_dev;
    a.dvec = c->diag_dev;
    VGH_REQUIRE(!c->diag_dev || (c->fmt == VGH_FMT_I8 && !c->out_fp8 && c->cin >= c->cout_pad && c->cout_pad <= 1024), "conv2d: diag_dev belongs to an int8 -> bf16 conv with cout_pad <= min(cin, 1024)");
    VGH_REQUIRE(!(a.in_fp8 || a.out_fp8) || (c->gscale_dev && (vgh_fmt_is_q8(c->fmt) || c->fmt == VGH_FMT_BF16) && !c->out_f32), "conv2d: an e4m3 / int8 conv needs gscale_dev, a bf16 or e4m3 input and no fp32 output");
    const bool h16 = c->fmt == VGH_FMT_F16;  // single-plane fp16: the fp16 split kernels with one K segment and plane strides 0
    a.split = h16 ? VGH_FMT_F16X2 : planes > 1 ? c->fmt : 0;
    a.nseg = h16 ? 1 : 3;
    a.in_plane = h16 ? 0 : (int)c->in_pitch;
    a.out_plane = h16 ? 0 : (int)c->out_pitch;
    a.res_plane = h16 ? 0 : (int)c->res_pitch;
    a.out_scale = c->out_scale;
    a.grp_cout = c->grp_cout;
    a.grp_in_stride = c->grp_in_stride;
    a.in = (const uint16_t*)c->in_dev;
    a.in_pitch = c->in_pitch * planes;
    a.in_coff = c->in_coff;
    a.cin = c->cin;
    a.B = c->B;
    a.H = c->H;
    a.W = c->W;
    a.ksize = c->ksize;
    a.stride = c->stride;
    a.pad = c->ksize / 2;
    a.Ho = (c->H + 2 * a.pad - c->ksize) / c->stride + 1;
    a.Wo = (c->W + 2 * a.pad - c->ksize) / c->stride + 1;
    a.wpack = (const uint16_t*)c->wpack_dev;
    a.bias = c->bias_dev;
    a.out = c->out_dev;
    a.out_pitch = c->out_f32 ? c->out_pitch : c->out_pitch * planes;
    a.out_coff = c->out_coff;
    a.out_coff2 = c->out_coff2;
    a.out_split = c->out_split;
    a.cout_pad = c->cout_pad;
    a.cout_store = c->cout_store;
    a.out_f32 = c->out_f32;
    a.res = (const uint16_t*)c->res_dev;
    a.res_pitch = c->res_pitch * planes;
    a.res_coff = c->res_coff;
    a.alpha = c->alpha;
    a.act = c->act;
    a.shuffle = c->shuffle;
    a.shuffle_c = c->shuffle ? c->cout_pad / 4 : 0;
    a.zeros = g_zeros[dev];
    a.P = c->B * a.Ho * a.Wo;
    a.cblocks = c->cin / (a.in_fp8 ? 64 : 32);
    a.nkb = c->ksize * c->ksize * a.cblocks;
    return vgh_launch_conv(a, c->force_cfg, (hipStream_t)stream);
}

int vgh_pack_conv_weights(const float* w_host, int cout_pad, int ksize, int cin, uint16_t* wpack_host) {
    VGH_REQUIRE(w_host && wpack_host, "pack: null argument");
    VGH_REQUIRE(cin % 32 == 0 && cout_pad % 32 == 0 && (ksize == 1 || ksize == 3), "pack: cin/cout_pad must be multiples of 32, ksize 1 or 3");
    vgh_pack_conv_weights_host(w_host, cout_pad, ksize, cin, wpack_host);
    return VGH_OK;
}

int vgh_pack_conv_weights_fp8(const float* w_host, int cout_pad, int ksize, int cin, uint8_t* wpack_host, float* wscale_host) {
    VGH_REQUIRE(w_host && wpack_host && wscale_host, "pack_fp8: null argument");
    VGH_REQUIRE(cin % 64 == 0 && cout_pad % 32 == 0 && ksize == 3, "pack_fp8: cin must be a multiple of 64, cout_pad of 32, ksize 3");
    vgh_pack_conv_weights_fp8_host(w_host, cout_pad, ksize, cin, wpack_host, wscale_host);
    return VGH_OK;
}
int vgh_net_set_i8_diag(int on) {
    g_i8_diag.store(on ? 1 : 0, std::memory_order_relaxed);
    return VGH_OK;
}
int vgh_net_op_has_diag(vgh_net* n, int op_index) { return (n && op_index >= 0 && op_index < (int)n->ops.size() && n->ops[op_index].dvec) ? 1 : 0; }
int vgh_pack_conv_weights_i8(const float* w_host, int cout_pad, int ksize, int cin, uint8_t* wpack_host, float* wscale_host) {
    VGH_REQUIRE(w_host && wpack_host && wscale_host, "pack_i8: null argument");
    VGH_REQUIRE(cin % 64 == 0 && cout_pad % 32 == 0 && ksize == 3, "pack_i8: cin must be a multiple of 64, cout_pad of 32, ksize 3");
    vgh_pack_conv_weights_i8_host(w_host, cout_pad, ksize, cin, wpack_host, wscale_host);
    return VGH_OK;
}

int vgh_stream_create(int device, void** stream_out) {
    VGH_REQUIRE(stream_out, "stream_create: null");
    VGH_HIP(hipSetDevice(device));
    hipStream_t s;
    VGH_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream_out = (void*)s;
    return VGH_OK;
}
int vgh_stream_destroy(void* stream) {
    VGH_HIP(hipStreamDestroy((hipStream_t)stream));
    return VGH_OK;
}
int vgh_stream_sync(void* stream) {
    VGH_HIP(hipStreamSynchronize((hipStream_t)stream));
    return VGH_OK;
}
int vgh_event_create(void** ev_out) {
    VGH_REQUIRE(ev_out, "event_create: null");
    hipEvent_t e;
    VGH_HIP(hipEventCreate(&e));
    *ev_out = (void*)e;
    return VGH_OK;
}
int vgh_event_destroy(void* ev) {
    VGH_HIP(hipEventDestroy((hipEvent_t)ev));
    return VGH_OK;
}
int vgh_event_record(void* ev, void* stream) {
    VGH_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return VGH_OK;
}
int vgh_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out) {
    VGH_REQUIRE(ms_out, "event_elapsed: null");
    VGH_HIP(hipEventSynchronize((hipEvent_t)ev_stop));
    VGH_HIP(hipEventElapsedTime(ms_out, (hipEvent_t)ev_start, (hipEvent_t)ev_stop));
    return VGH_OK;
}

}  // extern "C"
