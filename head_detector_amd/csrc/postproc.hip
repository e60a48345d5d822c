// Post-network stages for gfx950: DFL/box/score decode, per-image top-k, candidate gather with FLAME
// parameter fix-up, greedy NMS, slab compaction.  All fp32; IoU arithmetic is IEEE (no contraction,
// true division) so keep/suppress decisions are bit-identical to the reference CPU path.
//
// Reference: yolo_head_training/yolo_head/yolo_head_ndfl_heads.py:143-172 (decode + fix-up),
// yolo_head_dfl_head.py:162-184 (activations, zero pad, channel order), yolo_heads.py:63-86 (top-k),
// head_detector/utils.py:159-194 + torchvision.ops.nms (conf filter, NMS, keep-100).
#include "vgh_internal.h"

namespace {

constexpr int MAX_LEVELS = 4;
struct Levels {
    const float* pred[MAX_LEVELS];
    int h[MAX_LEVELS], w[MAX_LEVELS], pitch[MAX_LEVELS], stride[MAX_LEVELS], start[MAX_LEVELS + 1];
    int n;
};

__device__ __forceinline__ int find_level(const Levels& L, int a) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < MAX_LEVELS; ++i)
        if (i < L.n && a >= L.start[i]) l = i;
    return l;
}

// ---- K6: boxes + scores for every anchor --------------------------------------------------------
__global__ __launch_bounds__(256) void head_decode_kernel(Levels L, int B, int A, float* __restrict__ boxes, float* __restrict__ scores) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)B * A) return;
    const int b = (int)(gid / A), a = (int)(gid - (int64_t)b * A);
    const int l = find_level(L, a);
    const int p = a - L.start[l];
    const int hw = L.h[l] * L.w[l];
    const float* pr = L.pred[l] + ((int64_t)b * hw + p) * L.pitch[l];
    const int ay = p / L.w[l], ax = p - ay * L.w[l];
    float d[4];
#pragma unroll
    for (int side = 0; side < 4; ++side) {
        float x[17];
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < 17; ++k) {
            x[k] = pr[side * 17 + k];
            m = fmaxf(m, x[k]);
        }
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 17; ++k) {
            x[k] = expf(x[k] - m);
            s += x[k];
        }
        float e = 0.0f;
#pragma unroll
        for (int k = 0; k < 17; ++k) e += (x[k] / s) * (float)k;  // softmax(dim=bins) * linspace(0..16), summed
        d[side] = e;
    }
    const float cx = (float)ax + 0.5f, cy = (float)ay + 0.5f, st = (float)L.stride[l];
    float4 o;
    o.x = (cx - d[0]) * st;  // batch_distance2bbox: x1y1 = p - lt, x2y2 = p + rb ; then * stride
    o.y = (cy - d[1]) * st;
    o.z = (cx + d[2]) * st;
    o.w = (cy + d[3]) * st;
    *(float4*)(boxes + gid * 4) = o;
    scores[gid] = 1.0f / (1.0f + expf(-pr[VGH_PRED_CLS_OFF]));
}

// ---- K7: per-image top-k via radix select on a 64-bit composite (score key, ~index) -------------
__device__ __forceinline__ uint32_t float_key(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // order-preserving
}
__device__ __forceinline__ uint64_t composite(float f, int idx) { return ((uint64_t)float_key(f) << 32) | (uint32_t)(~(uint32_t)idx); }

// r06 (profiles/r06_latency_trace_l1.txt: 47 us of a 2.0-ms single-image call): (i) RC composites per thread live in registers when the row fits (A <= RC * 1024: the
// 8 400 anchors of a 640 image; the 33 600 of a 1280 image re-read the scores as before), so the scores are read once instead of once per pass; (ii) the radix
// select STOPS at the first pass whose selected bin holds exactly the remaining count -- every element of that bin is then selected, the threshold is the bin's lower
// edge, and with distinct scores that is the 3rd or 4th of the 8 passes (the index bits only matter when the k-th score is tied).  Same selected set, same order.
template <bool CACHED>
__global__ __launch_bounds__(1024) void topk_kernel(const float* __restrict__ scores, int A, int k, int32_t* __restrict__ out_idx,
                                                    float* __restrict__ out_scores) {
    constexpr int RC = 9;
    __shared__ uint32_t hist[256];
    __shared__ uint64_t s_prefix;
    __shared__ int s_kk;
    __shared__ int s_count;
    __shared__ int s_done;
    __shared__ uint64_t sel[1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* sc = scores + (int64_t)b * A;
    if (tid == 0) {
        s_prefix = 0;
        s_kk = k;
        s_count = 0;
        s_done = 0;
    }
    sel[tid] = 0;
    uint64_t cache[RC];
    if constexpr (CACHED) {
#pragma unroll
        for (int r = 0; r < RC; ++r) {
            const int i = tid + r * 1024;
            cache[r] = i < A ? composite(sc[i], i) : 0ull;  // 0 is below every real composite (its low word would be the index 0xffffffff)
        }
    }
    __syncthreads();
    // up to 8 passes x 8 bits, MSB first: after the last pass s_prefix == the k-th largest composite (or the lower edge of a bin that is selected as a whole)
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 56 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const uint64_t prefix = s_prefix;
        if constexpr (CACHED) {
#pragma unroll
            for (int r = 0; r < RC; ++r) {
                const uint64_t c = cache[r];
                const bool match = c != 0ull && ((pass == 0) || ((c >> (shift + 8)) == (prefix >> (shift + 8))));
                if (match) atomicAdd(&hist[(c >> shift) & 255], 1u);
            }
        } else {
            for (int i = tid; i < A; i += 1024) {
                const uint64_t c = composite(sc[i], i);
                const bool match = (pass == 0) || ((c >> (shift + 8)) == (prefix >> (shift + 8)));
                if (match) atomicAdd(&hist[(c >> shift) & 255], 1u);
            }
        }
        __syncthreads();
        const int kk = s_kk;
        __syncthreads();
        if (tid < 256) {
            uint32_t above = 0;
            for (int j = tid + 1; j < 256; ++j) above += hist[j];
            const uint32_t mine = hist[tid];
            if ((int)above < kk && kk <= (int)(above + mine)) {
                s_prefix = prefix | ((uint64_t)tid << shift);
                s_kk = kk - (int)above;
                if (kk == (int)(above + mine)) s_done = 1;  // the whole bin is selected: its lower edge is a valid threshold
            }
        }
        __syncthreads();
        if (s_done) break;  // (block-uniform: read after the barrier)
    }
    const uint64_t T = s_prefix;
    if constexpr (CACHED) {
#pragma unroll
        for (int r = 0; r < RC; ++r) {
            const uint64_t c = cache[r];
            if (c != 0ull && c >= T) {
                const int slot = atomicAdd(&s_count, 1);
                if (slot < 1024) sel[slot] = c;
            }
        }
    } else {
        for (int i = tid; i < A; i += 1024) {
            const uint64_t c = composite(sc[i], i);
            if (c >= T) {
                const int slot = atomicAdd(&s_count, 1);
                if (slot < 1024) sel[slot] = c;
            }
        }
    }
    __syncthreads();
    // bitonic sort, descending, 1024 entries (zeros = padding sink to the end).  r06: every thread keeps ITS entry in registers; the 45 compare-exchange steps whose
    // partner sits in the same wave (stride < 64) are cross-lane shuffles with no barrier, only the 10 steps with stride >= 64 go through LDS (20 barriers instead
    // of 55: the sort was ~15 us of a 34-us launch whose single block per image has nothing else to hide behind).  Same network, same comparisons, same result.
    uint64_t v = sel[tid];
    for (int size = 2; size <= 1024; size <<= 1) {
        const bool desc = (tid & size) == 0;
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            uint64_t pv;
            if (stride >= 64) {
                __syncthreads();  // (everybody has read its partner of the previous LDS step)
                sel[tid] = v;
                __syncthreads();
                pv = sel[tid ^ stride];
            } else {
                pv = __shfl_xor(v, stride, 64);
            }
            const bool take_max = ((tid & stride) == 0) == desc;  // the lower index of a pair keeps the larger entry in a descending run
            v = take_max ? (v > pv ? v : pv) : (v < pv ? v : pv);
        }
    }
    if (tid < k) {
        const int idx = (int)(~(uint32_t)(v & 0xffffffffu));
        out_idx[(int64_t)b * k + tid] = idx;
        if (out_scores) out_scores[(int64_t)b * k + tid] = sc[idx];
    }
}

// ---- K6b: gather boxes + FLAME 413-vector for selected anchors ----------------------------------
// channel c of the reference's 413-vector for the anchor whose prediction row is `pr` (yolo_head_dfl_head.py:162-184, yolo_head_ndfl_heads.py:143-172): tanh * 3 on the
// live shape / expression channels (zeros behind them), the rotation / jaw permutation, translation += anchor centre * stride, exp(scale) / 0.05 * stride
__device__ __forceinline__ float flame_channel(const float* __restrict__ pr, int c, int S, int E, int ax, int ay, float st) {
    const int o_shape = VGH_PRED_FLAME_OFF, o_expr = o_shape + S, o_rot = o_expr + E, o_jaw = o_rot + 6, o_tr = o_jaw + 3, o_sc = o_tr + 3;
    float v = 0.0f;
    if (c < 300) {
        if (c < S) v = tanhf(pr[o_shape + c]) * 3.0f;
    } else if (c < 400) {
        if (c - 300 < E) v = tanhf(pr[o_expr + c - 300]) * 3.0f;
    } else if (c < 403) {
        v = pr[o_rot + 3 + (c - 400)];  // T[400:403] = O[403:406] = rot branch [3:6]
    } else if (c < 406) {
        v = pr[o_jaw + (c - 403)];  // T[403:406] = O[406:409] = jaw branch
    } else if (c < 409) {
        v = pr[o_rot + (c - 406)];  // T[406:409] = O[400:403] = rot branch [0:3]
    } else if (c == 409) {
        v = pr[o_tr + 0] + ((float)ax + 0.5f) * st;  // translation[:, 0:2] += anchor * stride
    } else if (c == 410) {
        v = pr[o_tr + 1] + ((float)ay + 0.5f) * st;
    } else if (c == 411) {
        v = pr[o_tr + 2];
    } else {
        v = (expf(pr[o_sc]) / 0.05f) * st;  // exp(x)/0.05, then scale *= stride
    }
    return v;
}

// (r06: four candidate rows per 256-thread block, one wave each -- 16 000 blocks per 64 images instead of 64 000 one-wave blocks)
__global__ __launch_bounds__(256) void gather_kernel(Levels L, int A, int S, int E, const float* __restrict__ boxes, const int32_t* __restrict__ idx,
                                                     int k, float* __restrict__ out_boxes, float* __restrict__ out_flame) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y, lane = threadIdx.x & 63;
    if (j >= k) return;
    const int a = idx[(int64_t)b * k + j];
    const int l = find_level(L, a);
    const int p = a - L.start[l];
    const int hw = L.h[l] * L.w[l];
    const float* pr = L.pred[l] + ((int64_t)b * hw + p) * L.pitch[l];
    const int ay = p / L.w[l], ax = p - ay * L.w[l];
    const float st = (float)L.stride[l];
    if (lane < 4) out_boxes[((int64_t)b * k + j) * 4 + lane] = boxes[((int64_t)b * A + a) * 4 + lane];
    if (!out_flame) return;  // lazy mode (vgh_detector_set_lazy_flame): the 413-vector is built for the SURVIVORS only, inside nms_select_kernel
    float* of = out_flame + ((int64_t)b * k + j) * VGH_NUM_FLAME_PARAMS;
    for (int c = lane; c < VGH_NUM_FLAME_PARAMS; c += 64) of[c] = flame_channel(pr, c, S, E, ax, ay, st);
}

// ---- K8: conf filter + greedy NMS + keep-k -------------------------------------------------------
__device__ __forceinline__ bool iou_gt(const float4 a, const float area_a, const float4 b, const float thr) {
#pragma clang fp contract(off)
    const float area_b = (b.z - b.x) * (b.w - b.y);
    const float xx1 = fmaxf(a.x, b.x), yy1 = fmaxf(a.y, b.y);
    const float xx2 = fminf(a.z, b.z), yy2 = fminf(a.w, b.w);
    const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
    const float inter = w * h;
    const float ovr = inter / (area_a + area_b - inter);
    return ovr > thr;
}

__global__ __launch_bounds__(1024) void nms_kernel(const float* __restrict__ boxes, const float* __restrict__ scores, int n_in, float conf, float thr,
                                                   int keep_k, int32_t* __restrict__ keep_idx, int32_t* __restrict__ counts) {
    __shared__ float4 sb[1024];
    __shared__ unsigned long long removed[16];
    __shared__ int s_n;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < 16) removed[tid] = 0ull;
    if (tid == 0) s_n = 0;
    __syncthreads();
    bool valid = false;
    if (tid < n_in) {
        sb[tid] = *(const float4*)(boxes + ((int64_t)b * n_in + tid) * 4);
        valid = scores[(int64_t)b * n_in + tid] >= conf;  // utils.py:174  (inputs sorted descending => a prefix)
    }
    const unsigned long long bal = __ballot(valid);
    if ((tid & 63) == 0) atomicAdd(&s_n, __popcll(bal));
    __syncthreads();
    const int n = s_n;
    for (int j = tid; j < keep_k; j += 1024) keep_idx[(int64_t)b * keep_k + j] = -1;
    int kept = 0;
    int i = 0;
    while (kept < keep_k) {
        // next candidate not yet suppressed (uniform: every thread scans the same LDS words)
        int nxt = -1;
        for (int wd = i >> 6; wd < 16 && nxt < 0; ++wd) {
            unsigned long long free_bits = ~removed[wd];
            if (wd == (i >> 6)) free_bits &= ~0ull << (i & 63);
            if (free_bits) nxt = wd * 64 + __ffsll((long long)free_bits) - 1;
        }
        if (nxt < 0 || nxt >= n) break;
        i = nxt;
        if (tid == 0) keep_idx[(int64_t)b * keep_k + kept] = i;
        ++kept;
        const float4 bi = sb[i];
        float area_i;
        {
#pragma clang fp contract(off)
            area_i = (bi.z - bi.x) * (bi.w - bi.y);
        }
        const bool sup = (tid > i) && (tid < n) && iou_gt(bi, area_i, sb[tid < n_in ? tid : 0], thr);
        const unsigned long long m = __ballot(sup);
        __syncthreads();  // everyone has finished scanning `removed` for this round
        if ((tid & 63) == 0 && m) removed[tid >> 6] |= m;
        __syncthreads();
        ++i;
    }
    if (tid == 0) counts[b] = kept;
}

// r06 (single-image latency: profiles/r06_latency_trace_l1.txt shows nms -> compact -> head_list as three 5.7-us launches, each waiting for the one before): the three
// as ONE launch.  Block b = image b: the same greedy loop as nms_kernel (kept positions also kept in LDS), then its 16 waves copy the image's keep_k survivor rows
// (compact_kernel's copy, zeros behind the count); the block that finishes LAST (a ticket in device memory, reset for the next launch) builds the image-major head list
// from counts[B] (head_list_kernel's scan).  Same decisions, same bytes: tests compare it with the staged entry points.
__global__ __launch_bounds__(1024) void nms_select_kernel(const float* __restrict__ boxes, const float* __restrict__ scores, const float* __restrict__ flame, int n_in,
                                                          float conf, float thr, int keep_k, int32_t* __restrict__ keep_idx, int32_t* __restrict__ counts,
                                                          float* __restrict__ ob, float* __restrict__ os, float* __restrict__ of, int B, int capacity,
                                                          int32_t* __restrict__ head_row, int32_t* __restrict__ head_image, int32_t* __restrict__ n_heads,
                                                          int32_t* __restrict__ ticket, Levels L, const int32_t* __restrict__ lazy_idx, int S, int E) {
    __shared__ float4 sb[1024];
    __shared__ unsigned long long removed[16];
    __shared__ int s_n;
    __shared__ int s_keep[1024];
    __shared__ int s_off[1025];
    __shared__ int s_last;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < 16) removed[tid] = 0ull;
    if (tid == 0) s_n = 0;
    __syncthreads();
    bool valid = false;
    if (tid < n_in) {
        sb[tid] = *(const float4*)(boxes + ((int64_t)b * n_in + tid) * 4);
        valid = scores[(int64_t)b * n_in + tid] >= conf;
    }
    const unsigned long long bal = __ballot(valid);
    if ((tid & 63) == 0) atomicAdd(&s_n, __popcll(bal));
    __syncthreads();
    const int n = s_n;
    for (int j = tid; j < keep_k; j += 1024) keep_idx[(int64_t)b * keep_k + j] = -1;
    int kept = 0;
    int i = 0;
    while (kept < keep_k) {
        int nxt = -1;
        for (int wd = i >> 6; wd < 16 && nxt < 0; ++wd) {
            unsigned long long free_bits = ~removed[wd];
            if (wd == (i >> 6)) free_bits &= ~0ull << (i & 63);
            if (free_bits) nxt = wd * 64 + __ffsll((long long)free_bits) - 1;
        }
        if (nxt < 0 || nxt >= n) break;
        i = nxt;
        if (tid == 0) {
            keep_idx[(int64_t)b * keep_k + kept] = i;
            s_keep[kept] = i;
        }
        ++kept;
        const float4 bi = sb[i];
        float area_i;
        {
#pragma clang fp contract(off)
            area_i = (bi.z - bi.x) * (bi.w - bi.y);
        }
        const bool sup = (tid > i) && (tid < n) && iou_gt(bi, area_i, sb[tid < n_in ? tid : 0], thr);
        const unsigned long long m = __ballot(sup);
        __syncthreads();
        if ((tid & 63) == 0 && m) removed[tid >> 6] |= m;
        __syncthreads();
        ++i;
    }
    if (tid == 0) counts[b] = kept;
    __syncthreads();  // s_keep is complete
    // ---- the image's survivor rows ----
    const int w = tid >> 6, lane = tid & 63;
    for (int j = w; j < keep_k; j += 16) {
        const int64_t o = (int64_t)b * keep_k + j;
        if (j >= kept) {
            if (lane < 4) ob[o * 4 + lane] = 0.0f;
            if (lane == 0) os[o] = 0.0f;
            if (of)
                for (int c = lane; c < VGH_NUM_FLAME_PARAMS; c += 64) of[o * VGH_NUM_FLAME_PARAMS + c] = 0.0f;
            continue;
        }
        const int64_t sr = (int64_t)b * n_in + s_keep[j];
        if (lane < 4) ob[o * 4 + lane] = boxes[sr * 4 + lane];
        if (lane == 0) os[o] = scores[sr];
        if (of) {
            if (lazy_idx) {  // lazy mode: the candidate tensor was never filled; the survivor's vector straight from its prediction row (gather_kernel's arithmetic)
                const int a = lazy_idx[sr];
                const int l = find_level(L, a);
                const int p = a - L.start[l];
                const float* pr = L.pred[l] + ((int64_t)b * (L.h[l] * L.w[l]) + p) * L.pitch[l];
                const int ay = p / L.w[l], ax = p - ay * L.w[l];
                const float st = (float)L.stride[l];
                for (int c = lane; c < VGH_NUM_FLAME_PARAMS; c += 64) of[o * VGH_NUM_FLAME_PARAMS + c] = flame_channel(pr, c, S, E, ax, ay, st);
            } else {
                for (int c = lane; c < VGH_NUM_FLAME_PARAMS; c += 64) of[o * VGH_NUM_FLAME_PARAMS + c] = flame[sr * VGH_NUM_FLAME_PARAMS + c];
            }
        }
    }
    if (!head_row) return;
    // ---- the head list, by the block that finishes last ----
    __threadfence();  // this block's counts[b] is visible before its ticket
    __syncthreads();
    if (tid == 0) s_last = (atomicAdd(ticket, 1) == B - 1) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    if (tid == 0) *ticket = 0;  // (every other block has taken its ticket: nobody touches it again in this launch)
    __threadfence();
    const int per = (B + 1023) / 1024;
    const int b0 = tid * per, b1 = min(B, b0 + per);
    int mine = 0;
    for (int q = b0; q < b1; ++q) mine += min(max(__hip_atomic_load(counts + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), 0), keep_k);
    s_off[tid] = mine;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = (tid >= d) ? s_off[tid - d] : 0;
        __syncthreads();
        s_off[tid] += v;
        __syncthreads();
    }
    int at = s_off[tid] - mine;
    if (tid == 1023) *n_heads = min(s_off[1023], capacity);
    for (int q = b0; q < b1; ++q) {
        const int c = min(max(__hip_atomic_load(counts + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), 0), keep_k);
        for (int e = 0; e < c; ++e) {
            if (at + e < capacity) {
                head_row[at + e] = q * keep_k + e;
                head_image[at + e] = q;
            }
        }
        at += c;
    }
}

__global__ __launch_bounds__(64) void compact_kernel(const float* __restrict__ boxes, const float* __restrict__ scores, const float* __restrict__ flame,
                                                     int n_in, const int32_t* __restrict__ keep_idx, int keep_k, float* __restrict__ ob,
                                                     float* __restrict__ os, float* __restrict__ of) {
    const int j = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int src = keep_idx[(int64_t)b * keep_k + j];
    const int64_t o = (int64_t)b * keep_k + j;
    if (src < 0) {
        if (lane < 4) ob[o * 4 + lane] = 0.0f;
        if (lane == 0) os[o] = 0.0f;
        if (of)
            for (int c = lane; c < VGH_NUM_FLAME_PARAMS; c += 64) of[o * VGH_NUM_FLAME_PARAMS + c] = 0.0f;
        return;
    }
    const int64_t s = (int64_t)b * n_in + src;
    if (lane < 4) ob[o * 4 + lane] = boxes[s * 4 + lane];
    if (lane == 0) os[o] = scores[s];
    if (of)
        for (int c = lane; c < VGH_NUM_FLAME_PARAMS; c += 64) of[o * VGH_NUM_FLAME_PARAMS + c] = flame[s * VGH_NUM_FLAME_PARAMS + c];
}

// rows of boxes [B,n,4] picked by idx [B,k] -> sorted boxes [B,k,4]
__global__ __launch_bounds__(256) void gather_boxes_kernel(const float* __restrict__ boxes, const int32_t* __restrict__ idx, int n, int k, float* __restrict__ out) {
    const int j = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (j >= k) return;
    const int src = idx[(int64_t)b * k + j];
    const f32x4_t v = *(const f32x4_t*)(boxes + ((int64_t)b * n + src) * 4);
    *(f32x4_t*)(out + ((int64_t)b * k + j) * 4) = v;
}

// survivors of the NMS (positions into the top-k order) -> rows of the ORIGINAL tensors: src = idx[b][keep[b][j]]
__global__ __launch_bounds__(64) void compact_indirect_kernel(const float* __restrict__ boxes, const float* __restrict__ flame, int flame_w, int n,
                                                              const int32_t* __restrict__ idx, const float* __restrict__ sorted_scores, int k,
                                                              const int32_t* __restrict__ keep_idx, int keep_k, float* __restrict__ ob, float* __restrict__ os,
                                                              float* __restrict__ of) {
    const int j = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int pos = keep_idx[(int64_t)b * keep_k + j];
    const int64_t o = (int64_t)b * keep_k + j;
    if (pos < 0) {
        if (lane < 4) ob[o * 4 + lane] = 0.0f;
        if (lane == 0) os[o] = 0.0f;
        if (of)
            for (int c = lane; c < flame_w; c += 64) of[o * flame_w + c] = 0.0f;
        return;
    }
    const int64_t s = (int64_t)b * n + idx[(int64_t)b * k + pos];
    if (lane < 4) ob[o * 4 + lane] = boxes[s * 4 + lane];
    if (lane == 0) os[o] = sorted_scores[(int64_t)b * k + pos];
    if (of)
        for (int c = lane; c < flame_w; c += 64) of[o * flame_w + c] = flame[s * flame_w + c];
}

int make_levels(const vgh_head_level* levels, int n_levels, Levels* L) {
    VGH_REQUIRE(n_levels >= 1 && n_levels <= MAX_LEVELS, "head: n_levels=%d out of range", n_levels);
    L->n = n_levels;
    L->start[0] = 0;
    for (int i = 0; i < MAX_LEVELS; ++i) {
        if (i < n_levels) {
            L->pred[i] = levels[i].pred_dev;
            L->h[i] = levels[i].h;
            L->w[i] = levels[i].w;
            L->pitch[i] = levels[i].pitch;
            L->stride[i] = levels[i].stride;
            L->start[i + 1] = L->start[i] + levels[i].h * levels[i].w;
        } else {
            L->pred[i] = nullptr;
            L->h[i] = L->w[i] = L->pitch[i] = L->stride[i] = 0;
            L->start[i + 1] = L->start[i];
        }
    }
    return VGH_OK;
}

}  // namespace

extern "C" {

int vgh_head_decode(const vgh_head_level* levels, int n_levels, int B, float* boxes_dev, float* scores_dev, void* stream) {
    Levels L;
    if (int rc = make_levels(levels, n_levels, &L)) return rc;
    const int A = L.start[n_levels];
    if (B * (int64_t)A == 0) return VGH_OK;
    for (int i = 0; i < n_levels; ++i) VGH_REQUIRE(levels[i].pitch >= 69, "head_decode: pitch too small");
    const int64_t total = (int64_t)B * A;
    hipLaunchKernelGGL(head_decode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, L, B, A, boxes_dev, scores_dev);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}

int vgh_topk(const float* scores_dev, int B, int A, int k, int32_t* idx_dev, float* topk_scores_dev, void* stream) {
    VGH_REQUIRE(k >= 1 && k <= 1024, "topk: k=%d must be in [1,1024]", k);
    VGH_REQUIRE(k <= A, "topk: k=%d exceeds the number of anchors %d (torch.topk raises too)", k, A);
    if (B == 0) return VGH_OK;
    if (A <= 9 * 1024)
        hipLaunchKernelGGL(topk_kernel<true>, dim3(B), dim3(1024), 0, (hipStream_t)stream, scores_dev, A, k, idx_dev, topk_scores_dev);
    else
        hipLaunchKernelGGL(topk_kernel<false>, dim3(B), dim3(1024), 0, (hipStream_t)stream, scores_dev, A, k, idx_dev, topk_scores_dev);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}

int vgh_gather_candidates(const vgh_head_level* levels, int n_levels, int B, int A, int shape_c, int expr_c, const float* boxes_dev,
                          const int32_t* idx_dev, int k, float* out_boxes_dev, float* out_flame_dev, void* stream) {
    Levels L;
    if (int rc = make_levels(levels, n_levels, &L)) return rc;
    VGH_REQUIRE(A == L.start[n_levels], "gather: A=%d does not match the levels (%d)", A, L.start[n_levels]);
    VGH_REQUIRE(shape_c >= 0 && shape_c <= 300 && expr_c >= 0 && expr_c <= 100, "gather: bad live channel counts");
    for (int i = 0; i < n_levels; ++i) VGH_REQUIRE(levels[i].pitch >= VGH_PRED_FLAME_OFF + shape_c + expr_c + 13, "gather: pitch too small");
    if (B == 0 || k == 0) return VGH_OK;
    hipLaunchKernelGGL(gather_kernel, dim3((k + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, L, A, shape_c, expr_c, boxes_dev, idx_dev, k, out_boxes_dev,
                       out_flame_dev);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}

int vgh_nms(const float* boxes_dev, const float* scores_dev, int B, int n_in, float conf_thr, float iou_thr, int keep_k, int32_t* keep_idx_dev,
            int32_t* counts_dev, void* stream) {
    VGH_REQUIRE(n_in >= 0 && n_in <= 1024, "nms: n_in=%d must be <= 1024", n_in);
    VGH_REQUIRE(keep_k >= 1, "nms: keep_k must be positive");
    if (B == 0) return VGH_OK;
    hipLaunchKernelGGL(nms_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, boxes_dev, scores_dev, n_in, conf_thr, iou_thr, keep_k, keep_idx_dev,
                       counts_dev);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}

}  // extern "C"
// (internal, C++ linkage: declared in vgh_internal.h)
int vgh_nms_select(const float* boxes_dev, const float* scores_dev, const float* flame_dev, int B, int n_in, float conf_thr, float iou_thr, int keep_k,
                   int32_t* keep_idx_dev, int32_t* counts_dev, float* out_boxes_dev, float* out_scores_dev, float* out_flame_dev, int capacity, int32_t* head_row_dev,
                   int32_t* head_image_dev, int32_t* n_heads_dev, int32_t* ticket_dev, const vgh_head_level* lazy_levels, int n_levels, const int32_t* lazy_idx_dev,
                   int shape_c, int expr_c, void* stream) {
    VGH_REQUIRE(n_in >= 0 && n_in <= 1024 && keep_k >= 1 && keep_k <= 1024, "nms_select: n_in=%d / keep_k=%d must be <= 1024", n_in, keep_k);
    VGH_REQUIRE(!head_row_dev || (head_image_dev && n_heads_dev && ticket_dev), "nms_select: the head list needs head_image, n_heads and the ticket");
    if (B == 0) return VGH_OK;
    Levels L;
    memset(&L, 0, sizeof(L));
    if (lazy_idx_dev) {
        VGH_REQUIRE(lazy_levels, "nms_select: the lazy FLAME gather needs the head levels");
        if (int rc = make_levels(lazy_levels, n_levels, &L)) return rc;
    }
    hipLaunchKernelGGL(nms_select_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, boxes_dev, scores_dev, flame_dev, n_in, conf_thr, iou_thr, keep_k, keep_idx_dev,
                       counts_dev, out_boxes_dev, out_scores_dev, out_flame_dev, B, capacity, head_row_dev, head_image_dev, n_heads_dev, ticket_dev, L, lazy_idx_dev, shape_c,
                       expr_c);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}

extern "C" {
int vgh_compact(const float* boxes_dev, const float* scores_dev, const float* flame_dev, int B, int n_in, const int32_t* keep_idx_dev, int keep_k,
                float* out_boxes_dev, float* out_scores_dev, float* out_flame_dev, void* stream) {
    if (B == 0) return VGH_OK;
    hipLaunchKernelGGL(compact_kernel, dim3(keep_k, B), dim3(64), 0, (hipStream_t)stream, boxes_dev, scores_dev, flame_dev, n_in, keep_idx_dev, keep_k,
                       out_boxes_dev, out_scores_dev, out_flame_dev);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}

int64_t vgh_topk_nms_workspace_bytes(int B, int n, int pre_k, int keep_k) {
    const int64_t k = pre_k < n ? pre_k : n;
    return (int64_t)B * (k * 4 /*idx*/ + k * 4 /*sorted scores*/ + k * 16 /*sorted boxes*/ + (int64_t)keep_k * 4 /*keep*/) + 256;
}

int vgh_topk_nms(const float* boxes_dev, const float* scores_dev, const float* flame_dev, int flame_width, int B, int n, float conf_thr, float iou_thr, int pre_k,
                 int keep_k, void* workspace_dev, float* out_boxes_dev, float* out_scores_dev, float* out_flame_dev, int32_t* counts_dev, void* stream) {
    VGH_REQUIRE(boxes_dev && scores_dev && workspace_dev && out_boxes_dev && out_scores_dev && counts_dev, "topk_nms: null argument");
    VGH_REQUIRE((flame_dev == nullptr) == (out_flame_dev == nullptr) && (!flame_dev || flame_width > 0), "topk_nms: flame input / output must both be given or both be NULL");
    VGH_REQUIRE(B >= 0 && n >= 1 && pre_k >= 1 && keep_k >= 1, "topk_nms: bad sizes");
    const int k = pre_k < n ? pre_k : n;
    VGH_REQUIRE(k <= 1024, "topk_nms: min(pre_k, n) = %d exceeds the 1024 candidates the NMS kernel holds", k);
    if (B == 0) return VGH_OK;
    VGH_REQUIRE(((uintptr_t)workspace_dev & 15) == 0, "topk_nms: workspace must be 16-byte aligned");
    auto al16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    char* ws = (char*)workspace_dev;
    const size_t o_ssc = al16((size_t)B * k * 4), o_sbx = al16(o_ssc + (size_t)B * k * 4), o_keep = o_sbx + (size_t)B * k * 16;
    int32_t* idx = (int32_t*)ws;
    float* ssc = (float*)(ws + o_ssc);
    float* sbx = (float*)(ws + o_sbx);
    int32_t* keep = (int32_t*)(ws + o_keep);
    int rc;
    // utils.py:174-185: score >= conf filter and top-k are one stable descending sort (ties by ascending index); the conf filter
    // is the prefix test inside vgh_nms
    if ((rc = vgh_topk(scores_dev, B, n, k, idx, ssc, stream))) return rc;
    hipLaunchKernelGGL(gather_boxes_kernel, dim3((k + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, boxes_dev, (const int32_t*)idx, n, k, sbx);
    VGH_HIP(hipGetLastError());
    if ((rc = vgh_nms(sbx, ssc, B, k, conf_thr, iou_thr, keep_k, keep, counts_dev, stream))) return rc;
    hipLaunchKernelGGL(compact_indirect_kernel, dim3(keep_k, B), dim3(64), 0, (hipStream_t)stream, boxes_dev, flame_dev, flame_width, n, (const int32_t*)idx,
                       (const float*)ssc, k, (const int32_t*)keep, keep_k, out_boxes_dev, out_scores_dev, out_flame_dev);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}

}  // extern "C"
