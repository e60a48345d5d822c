// Fused detector: the whole of HeadDetector._process + _parse_predictions' device-side arithmetic
// (head_detector/detector.py:54-90) behind ONE host call with no host round trip:
//   net (rows a2-a5) -> box/score decode (a6) -> top-k (a7) -> gather + FLAME fix-up (a6/a6') -> NMS for every image (a8)
//   -> compaction into fixed-capacity slabs -> head list (device prefix sum of the per-image counts)
//   -> FLAME decode of every survivor with the un-pad / un-scale fused (a9-a15) + calculate_rpy (a16).
// The data-dependent head count never visits the host: the FLAME kernels are launched at capacity and read the
// live count from device memory, so a caller can queue step s+1 while step s still runs.
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>

#include "vgh_internal.h"

struct vgh_detector {
    vgh_net* net = nullptr;
    vgh_flame* flame = nullptr;
    vgh_detect_cfg cfg{};
    int device = 0, S = 0, arena_batch = 0, A = 0;
    float* boxes_all = nullptr;    // [max_batch, A, 4]
    float* scores_all = nullptr;   // [max_batch, A]
    int32_t* idx = nullptr;        // [max_batch, pre_k]
    float* cand_scores = nullptr;  // [max_batch, pre_k]
    float* cand_boxes = nullptr;   // [max_batch, pre_k, 4]
    float* cand_flame = nullptr;   // [max_batch, pre_k, 413]
    int32_t* keep_idx = nullptr;   // [max_batch, keep_k]
    int32_t* head_row = nullptr;   // [max_batch * keep_k]
    int32_t* head_image = nullptr; // [max_batch * keep_k]
    int32_t* ticket = nullptr;     // [1], zero between launches: vgh_nms_select's last-block ticket
    // lazy FLAME gather (r06, vgh_detector_set_lazy_flame): the candidate stage gathers boxes only; the 413-vectors of the SURVIVORS are built by the select from the
    // prediction buffers (106 MB of candidate vectors per 64 images for ~3 survivors per image otherwise).  flame_pending: a lazy candidate stage is waiting for its select
    bool lazy_flame = false, flame_pending = false;
    // overlap mode: the select half (NMS .. FLAME decode: small, latency-bound kernels) runs on a detector-owned side stream,
    // concurrently with the network of the NEXT batch on the caller's stream
    bool overlap = false, side_pending = false;
    hipStream_t side = nullptr, side_main = nullptr;  // side: picked by ensure_side for work entering on side_main
    bool side_low = true;   // priority class of `side` (vgh_detector_set_side_priority)
    bool side_own = false;  // `side` was created for this detector alone (vgh_detector_renew_side): destroyed, not parked, on release
    hipEvent_t ev_net = nullptr, ev_cand = nullptr, ev_side = nullptr;
};

namespace {

// counts[B] -> image-major head list: head i of image b lives in slab row b*keep_k + i.
__global__ __launch_bounds__(1024) void head_list_kernel(const int32_t* __restrict__ counts, int B, int keep_k, int capacity, int32_t* __restrict__ head_row,
                                                        int32_t* __restrict__ head_image, int32_t* __restrict__ n_heads) {
    __shared__ int s_off[1025];
    const int t = threadIdx.x;
    // B can exceed the block: each thread serially sums its contiguous run of images, then a block scan over the runs
    const int per = (B + 1023) / 1024;
    const int b0 = t * per, b1 = min(B, b0 + per);
    int mine = 0;
    for (int b = b0; b < b1; ++b) mine += min(max(counts[b], 0), keep_k);
    s_off[t] = mine;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = (t >= d) ? s_off[t - d] : 0;
        __syncthreads();
        s_off[t] += v;
        __syncthreads();
    }
    int at = s_off[t] - mine;
    if (t == 1023) *n_heads = min(s_off[1023], capacity);
    for (int b = b0; b < b1; ++b) {
        const int c = min(max(counts[b], 0), keep_k);
        for (int i = 0; i < c; ++i) {
            if (at + i < capacity) {
                head_row[at + i] = b * keep_k + i;
                head_image[at + i] = b;
            }
        }
        at += c;
    }
}

// The side stream has to run next to the network of the following batch: measured to overlap with the caller's stream and the
// net's lane streams (streams.hip), picked on first use and again when the caller's stream changes.
// the side stream runs at the lowest stream priority: its small latency-bound kernels then fill in around the network's instead of
// taking compute units from them (measured r02, L b64: 13.25 ms per step vs 13.42 without overlap and 13.6-13.7 with a normal-priority
// side stream; M b32 5.32 vs 5.42)
constexpr bool kSideLowPriority = true;

void release_side(vgh_detector* d) {
    if (!d->side) return;
    (void)hipStreamSynchronize(d->side);
    if (d->side_own)
        (void)hipStreamDestroy(d->side);
    else
        vgh_stream_release_internal(d->device, d->side, d->side_low);
    d->side = nullptr;
    d->side_own = false;
}

int ensure_side(vgh_detector* d, hipStream_t main) {
    if (d->side && d->side_main == main) return VGH_OK;
    hipStream_t avoid[4] = {main};
    if (int rc = vgh_net_lane_streams(d->net, main, avoid + 1)) return rc;
    release_side(d);
    if (int rc = vgh_stream_acquire_internal(d->device, avoid, 4, &d->side, d->side_low)) return rc;
    d->side_main = main;
    return VGH_OK;
}

}  // namespace

extern "C" {

int vgh_detector_create(vgh_net* net, vgh_flame* flame, const vgh_detect_cfg* cfg, vgh_detector** out) {
    VGH_REQUIRE(net && cfg && out, "detector_create: null argument");
    VGH_REQUIRE(cfg->n_levels >= 1 && cfg->n_levels <= VGH_MAX_LEVELS, "detector_create: n_levels must be 1..%d", VGH_MAX_LEVELS);
    VGH_REQUIRE(cfg->pre_k >= 1 && cfg->pre_k <= 1024, "detector_create: pre_k must be 1..1024 (vgh_nms limit)");
    VGH_REQUIRE(cfg->keep_k >= 1 && cfg->keep_k <= cfg->pre_k, "detector_create: keep_k must be 1..pre_k");
    VGH_REQUIRE(cfg->max_batch >= 1, "detector_create: max_batch must be positive");
    VGH_REQUIRE(cfg->shape_live >= 0 && cfg->shape_live <= 300 && cfg->expr_live >= 0 && cfg->expr_live <= 100, "detector_create: live counts out of range");
    vgh_detector* d = new (std::nothrow) vgh_detector();
    VGH_REQUIRE(d, "detector_create: out of host memory");
    d->net = net;
    d->flame = flame;
    d->cfg = *cfg;
    d->S = vgh_net_image_size(net);
    d->device = vgh_net_device(net);
    d->arena_batch = vgh_net_max_batch(net);
    int A = 0;
    for (int l = 0; l < cfg->n_levels; ++l) {
        if (!vgh_net_buffer(net, cfg->level_buf[l])) {
            delete d;
            vgh_set_error("detector_create: level %d names buffer %d which the net does not have", l, cfg->level_buf[l]);
            return VGH_ERR_INVALID;
        }
        A += cfg->level_h[l] * cfg->level_w[l];
    }
    d->A = A;
    if (cfg->pre_k > A) {
        delete d;
        vgh_set_error("detector_create: pre_k %d exceeds the %d anchors", cfg->pre_k, A);
        return VGH_ERR_INVALID;
    }
    const size_t MB = (size_t)cfg->max_batch, K = (size_t)cfg->pre_k, KK = (size_t)cfg->keep_k;
#define ALLOC(field, count)                                                           \
    if (hipMalloc((void**)&d->field, (count) * sizeof(*d->field)) != hipSuccess) {     \
        vgh_set_error("detector_create: hipMalloc(%s, %zu) failed", #field, (size_t)((count) * sizeof(*d->field))); \
        vgh_detector_destroy(d);                                                      \
        return VGH_ERR_NOMEM;                                                         \
    }
    ALLOC(boxes_all, MB * A * 4);
    ALLOC(scores_all, MB * A);
    ALLOC(idx, MB * K);
    ALLOC(cand_scores, MB * K);
    ALLOC(cand_boxes, MB * K * 4);
    ALLOC(cand_flame, MB * K * VGH_NUM_FLAME_PARAMS);
    ALLOC(keep_idx, MB * KK);
    ALLOC(head_row, MB * KK);
    ALLOC(head_image, MB * KK);
    ALLOC(ticket, 1);
#undef ALLOC
    if (hipMemset(d->ticket, 0, sizeof(int32_t)) != hipSuccess) {
        vgh_set_error("detector_create: hipMemset failed");
        vgh_detector_destroy(d);
        return VGH_ERR_HIP;
    }
    *out = d;
    return VGH_OK;
}

void vgh_detector_destroy(vgh_detector* d) {
    if (!d) return;
    hipFree(d->boxes_all);
    hipFree(d->scores_all);
    hipFree(d->idx);
    hipFree(d->cand_scores);
    hipFree(d->cand_boxes);
    hipFree(d->cand_flame);
    hipFree(d->keep_idx);
    hipFree(d->head_row);
    hipFree(d->head_image);
    hipFree(d->ticket);
    release_side(d);
    if (d->net) vgh_net_set_pred_guard(d->net, nullptr);
    if (d->ev_net) hipEventDestroy(d->ev_net);
    if (d->ev_cand) hipEventDestroy(d->ev_cand);
    if (d->ev_side) hipEventDestroy(d->ev_side);
    delete d;
}

int vgh_detector_candidates(vgh_detector* d, const void* images_dev, int image_fmt, int B, void* stream) {
    VGH_REQUIRE(d && images_dev, "detector_candidates: null argument");
    VGH_REQUIRE(B >= 1 && B <= d->cfg.max_batch, "detector_candidates: batch %d outside 1..%d", B, d->cfg.max_batch);
    VGH_REQUIRE(image_fmt == VGH_IMG_F32_NCHW || image_fmt == VGH_IMG_U8_NHWC, "detector_candidates: unknown image format %d", image_fmt);
    const size_t img_bytes = (size_t)d->S * d->S * 3 * (image_fmt == VGH_IMG_F32_NCHW ? 4 : 1);
    // arena-sized chunks (the conv kernels address < 2 GiB per tensor; the arena is planned for arena_batch images)
    const bool lazy = d->lazy_flame;
    if (B > d->arena_batch) d->lazy_flame = false;  // chunks: the next chunk's forward overwrites the prediction buffers a lazy select would read
    int rc = VGH_OK;
    for (int at = 0; at < B && !rc; at += d->arena_batch) {
        const int n = (B - at < d->arena_batch) ? B - at : d->arena_batch;
        rc = vgh_net_forward(d->net, (const char*)images_dev + (size_t)at * img_bytes, image_fmt, n, stream);
        if (!rc) rc = vgh_detector_decode_candidates(d, n, at, stream);
    }
    d->lazy_flame = lazy;
    return rc;
}

int vgh_detector_decode_candidates(vgh_detector* d, int n, int at, void* stream) {
    VGH_REQUIRE(d, "detector_decode_candidates: null handle");
    VGH_REQUIRE(n >= 1 && at >= 0 && at + n <= d->cfg.max_batch && n <= d->arena_batch, "detector_decode_candidates: rows [%d,%d) outside the buffers", at, at + n);
    const vgh_detect_cfg& c = d->cfg;
    vgh_head_level lv[VGH_MAX_LEVELS];
    for (int l = 0; l < c.n_levels; ++l) {
        lv[l].pred_dev = (const float*)vgh_net_buffer(d->net, c.level_buf[l]);
        lv[l].h = c.level_h[l];
        lv[l].w = c.level_w[l];
        lv[l].pitch = c.level_pitch[l];
        lv[l].stride = c.level_stride[l];
    }
    void* st = stream;
    if (d->overlap) {  // predictions ready on `stream` -> side stream (ordered after everything queued there, incl. the last select)
        if (int rc0 = ensure_side(d, (hipStream_t)stream)) return rc0;
        VGH_HIP(hipEventRecord(d->ev_net, (hipStream_t)stream));
        VGH_HIP(hipStreamWaitEvent(d->side, d->ev_net, 0));
        st = d->side;
    }
    float* ba = d->boxes_all + (size_t)at * d->A * 4;
    float* sa = d->scores_all + (size_t)at * d->A;
    int32_t* ix = d->idx + (size_t)at * c.pre_k;
    int rc;
    if ((rc = vgh_head_decode(lv, c.n_levels, n, ba, sa, st))) return rc;
    if ((rc = vgh_topk(sa, n, d->A, c.pre_k, ix, d->cand_scores + (size_t)at * c.pre_k, st))) return rc;
    const bool lazy = d->lazy_flame && at == 0;  // (rows at > 0 belong to a chunked batch)
    if ((rc = vgh_gather_candidates(lv, c.n_levels, n, d->A, c.shape_live, c.expr_live, ba, ix, c.pre_k, d->cand_boxes + (size_t)at * c.pre_k * 4,
                                    lazy ? nullptr : d->cand_flame + (size_t)at * c.pre_k * VGH_NUM_FLAME_PARAMS, st)))
        return rc;
    d->flame_pending = lazy;
    if (d->overlap) {  // the next forward may run its backbone / neck now, but must not overwrite the predictions before this point
        VGH_HIP(hipEventRecord(d->ev_cand, d->side));
        if ((rc = vgh_net_set_pred_guard(d->net, d->ev_cand))) return rc;
        d->side_pending = true;
    }
    return VGH_OK;
}

int vgh_detector_candidate_buffers(vgh_detector* d, float** boxes_dev, float** scores_dev, float** flame_dev) {
    VGH_REQUIRE(d, "detector_candidate_buffers: null handle");
    if (boxes_dev) *boxes_dev = d->cand_boxes;
    if (scores_dev) *scores_dev = d->cand_scores;
    if (flame_dev) *flame_dev = d->cand_flame;
    return VGH_OK;
}

void* vgh_detector_scratch(vgh_detector* d, int which) {
    if (!d) return nullptr;
    switch (which) {
        case VGH_SCRATCH_BOXES_ALL: return d->boxes_all;
        case VGH_SCRATCH_SCORES_ALL: return d->scores_all;
        case VGH_SCRATCH_TOPK_IDX: return d->idx;
        case VGH_SCRATCH_KEEP_IDX: return d->keep_idx;
        case VGH_SCRATCH_HEAD_ROW: return d->head_row;
        default: return nullptr;
    }
}

int vgh_detector_set_flame(vgh_detector* d, vgh_flame* flame) {
    VGH_REQUIRE(d, "detector_set_flame: null handle");
    d->flame = flame;
    return VGH_OK;
}

static int select_on(vgh_detector* d, int B, float conf_thr, float iou_thr, vgh_detect_out* o, void* stream) {
    const vgh_detect_cfg& c = d->cfg;
    int rc;
    const bool want_heads = o->proj_dev || o->verts_dev || o->rot_dev || o->rpy_dev || o->n_heads_dev || o->head_image_dev;
    VGH_REQUIRE(!want_heads || o->n_heads_dev, "detector_select: n_heads_dev is required when any per-head output is requested");
    const int cap_all = B * c.keep_k;
    const int capacity = (o->head_capacity > 0 && o->head_capacity < cap_all) ? o->head_capacity : cap_all;
    int32_t* himg = o->head_image_dev ? o->head_image_dev : d->head_image;
    const bool lazy = d->flame_pending;
    d->flame_pending = false;
    VGH_REQUIRE(!lazy || c.keep_k <= 1024, "detector_select: the lazy FLAME gather needs keep_k <= 1024");
    if (c.keep_k <= 1024) {  // r06: NMS + compaction + head list as one launch
        vgh_head_level lv[VGH_MAX_LEVELS];
        for (int l = 0; l < c.n_levels; ++l) {
            lv[l].pred_dev = (const float*)vgh_net_buffer(d->net, c.level_buf[l]);
            lv[l].h = c.level_h[l];
            lv[l].w = c.level_w[l];
            lv[l].pitch = c.level_pitch[l];
            lv[l].stride = c.level_stride[l];
        }
        if ((rc = vgh_nms_select(d->cand_boxes, d->cand_scores, d->cand_flame, B, c.pre_k, conf_thr, iou_thr, c.keep_k, d->keep_idx, o->counts_dev, o->boxes_dev,
                                 o->scores_dev, o->flame_dev, capacity, want_heads ? d->head_row : nullptr, himg, o->n_heads_dev, d->ticket, lazy ? lv : nullptr,
                                 c.n_levels, lazy ? d->idx : nullptr, c.shape_live, c.expr_live, stream)))
            return rc;
        if (lazy && d->overlap && stream == (void*)d->side) {
            // the select has just read the prediction buffers: the next forward's guard moves behind it (same event, recorded again; the net waits for the latest record)
            VGH_HIP(hipEventRecord(d->ev_cand, d->side));
        }
        if (!want_heads) return VGH_OK;
    } else {
        if ((rc = vgh_nms(d->cand_boxes, d->cand_scores, B, c.pre_k, conf_thr, iou_thr, c.keep_k, d->keep_idx, o->counts_dev, stream))) return rc;
        if ((rc = vgh_compact(d->cand_boxes, d->cand_scores, d->cand_flame, B, c.pre_k, d->keep_idx, c.keep_k, o->boxes_dev, o->scores_dev, o->flame_dev, stream)))
            return rc;
        if (!want_heads) return VGH_OK;
        hipLaunchKernelGGL(head_list_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const int32_t*)o->counts_dev, B, c.keep_k, capacity, d->head_row, himg,
                           o->n_heads_dev);
        VGH_HIP(hipGetLastError());
    }
    if (!(o->proj_dev || o->verts_dev || o->rot_dev || o->rpy_dev)) return VGH_OK;
    VGH_REQUIRE(d->flame, "detector_select: per-head FLAME outputs requested but the detector was created without a FLAME handle");
    return vgh_flame_decode_indirect(d->flame, o->flame_dev, d->head_row, himg, o->n_heads_dev, capacity, c.shape_live, c.expr_live, o->unpad_dev, o->verts_dev,
                                     o->rot_dev, o->rpy_dev, o->proj_dev, stream);
}

int vgh_detector_select(vgh_detector* d, int B, float conf_thr, float iou_thr, vgh_detect_out* o, void* stream) {
    VGH_REQUIRE(d && o, "detector_select: null argument");
    VGH_REQUIRE(B >= 1 && B <= d->cfg.max_batch, "detector_select: batch %d outside 1..%d", B, d->cfg.max_batch);
    VGH_REQUIRE(o->boxes_dev && o->scores_dev && o->flame_dev && o->counts_dev, "detector_select: boxes/scores/flame/counts outputs are mandatory");
    if (!d->overlap) return select_on(d, B, conf_thr, iou_thr, o, stream);
    // overlap mode: the candidates were produced on the side stream; the select simply follows them there
    if (int rc0 = ensure_side(d, (hipStream_t)stream)) return rc0;
    const int rc = select_on(d, B, conf_thr, iou_thr, o, d->side);
    d->side_pending = true;
    return rc;
}

#ifdef VGH_EXPERIMENTS
// The side stream's priority class (default: lowest) -- takes effect at the next overlapped call, which picks a new side stream.  Between batches: a pending select is joined
// into nothing (the side stream is drained) before the stream is given back.
int vgh_detector_set_side_priority(vgh_detector* d, int low) {
    VGH_REQUIRE(d, "detector_set_side_priority: null handle");
    if ((low != 0) == d->side_low) return VGH_OK;
    if (d->side) VGH_HIP(hipStreamSynchronize(d->side));
    release_side(d);
    if (d->net) vgh_net_set_pred_guard(d->net, nullptr);
    d->side_pending = false;
    d->side_low = low != 0;
    return VGH_OK;
}
#endif

int vgh_detector_set_overlap(vgh_detector* d, int enable) {
    VGH_REQUIRE(d, "detector_set_overlap: null handle");
    if (enable && !d->ev_net) {  // the side stream itself is picked on first use (ensure_side: it has to overlap with the caller's stream)
        VGH_HIP(hipEventCreateWithFlags(&d->ev_net, hipEventDisableTiming));
        VGH_HIP(hipEventCreateWithFlags(&d->ev_cand, hipEventDisableTiming));
        VGH_HIP(hipEventCreateWithFlags(&d->ev_side, hipEventDisableTiming));
    }
    if (!enable && d->side) {
        VGH_HIP(hipStreamSynchronize(d->side));
        vgh_net_set_pred_guard(d->net, nullptr);
        d->side_pending = false;
    }
    d->overlap = enable != 0;
    return VGH_OK;
}

int vgh_detector_set_lazy_flame(vgh_detector* d, int enable) {
    VGH_REQUIRE(d, "detector_set_lazy_flame: null handle");
    d->lazy_flame = enable != 0;
    return VGH_OK;
}

int vgh_detector_streams(vgh_detector* d, void* main_stream, void** out) {
    VGH_REQUIRE(d && out, "detector_streams: null argument");
    hipStream_t lanes[3];
    if (int rc = vgh_net_lane_streams(d->net, (hipStream_t)main_stream, lanes)) return rc;
    for (int i = 0; i < 3; ++i) out[i] = (void*)lanes[i];
    out[3] = nullptr;
    if (d->overlap) {
        if (int rc = ensure_side(d, (hipStream_t)main_stream)) return rc;
        out[3] = (void*)d->side;
    }
    return VGH_OK;
}

int vgh_detector_join(vgh_detector* d, void* stream) {
    VGH_REQUIRE(d, "detector_join: null handle");
    if (d->overlap && d->side && d->side_pending) {
        VGH_HIP(hipEventRecord(d->ev_side, d->side));
        VGH_HIP(hipStreamWaitEvent((hipStream_t)stream, d->ev_side, 0));
    }
    return VGH_OK;
}

int vgh_detector_record(vgh_detector* d, void* event, void* stream) {
    VGH_REQUIRE(d && event, "detector_record: null argument");
    // the caller's event behind everything queued so far for the post-network stages: on the side stream in overlap mode (nothing is made to WAIT for it on the
    // device -- a host that polls / synchronises on it can queue dependent work later without parking a hardware queue behind a low-priority stream), else on `stream`
    if (d->overlap && d->side && d->side_pending)
        VGH_HIP(hipEventRecord((hipEvent_t)event, d->side));
    else
        VGH_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return VGH_OK;
}

int vgh_detect(vgh_detector* d, const void* images_dev, int image_fmt, int B, float conf_thr, float iou_thr, vgh_detect_out* o, void* stream) {
    int rc = vgh_detector_candidates(d, images_dev, image_fmt, B, stream);
    if (rc) return rc;
    return vgh_detector_select(d, B, conf_thr, iou_thr, o, stream);
}

}  // extern "C"
