// Two-plane 16-bit storage of the split-precision parity modes (device helpers shared by the conv, stem and pool kernels).
#pragma once
#include <type_traits>

#include "vgh_internal.h"

// SP = VGH_FMT_BF16X2 / VGH_FMT_F16X2 (SP = 0: the plain bf16 throughput kernels)
// A value v lives in two 16-bit planes: hi = rn16(v), lo = rn16((v - hi) * lo_scale); v - hi is exact in fp32, so the pair carries
// 16 (bf16) / 22 (fp16) significand bits.  fp16 stores lo scaled by 2^11 so that it stays in fp16's normal range (the split
// of Ootomo & Yokota's error-corrected tensor-core GEMM); hi values below fp16's smallest normal are flushed to zero (their whole
// value then sits in lo), values beyond +-65504 saturate.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
template <int SP>
struct SplitT {
    using e = __bf16;
    using v8 = bf16x8_t;
    using v4 = bf16x4_t;
};
template <>
struct SplitT<VGH_FMT_F16X2> {
    using e = _Float16;
    using v8 = f16x8_t;
    using v4 = f16x4_t;
};
template <int SP>
__device__ __forceinline__ void split_elem(float v, float lo_scale, typename SplitT<SP>::e& h, typename SplitT<SP>::e& l) {
    using E = typename SplitT<SP>::e;
    if constexpr (SP == VGH_FMT_F16X2) {
        // saturate at fp16's largest finite value (|v| > 65504 and +-Inf are clipped: documented in DESIGN 3.8); a NaN is NOT a number to clamp --
        // fminf / fmaxf would turn it into -65504, i.e. hide a numerical fault the bf16 and fp32 paths propagate -- so it passes through into both planes
        const float c = fminf(fmaxf(v, -65504.0f), 65504.0f);
        v = (v != v) ? v : c;
        h = fabsf(v) < 6.103515625e-05f ? (E)0.0f : (E)v;
    } else {
        h = (E)v;
    }
    l = (E)((v - (float)h) * lo_scale);
}
// N = 8 / 4 consecutive channels: hi vector to p, lo vector `plane` elements behind it
template <int SP, int N>
__device__ __forceinline__ void split_store(const float (&v)[N], float lo_scale, uint16_t* p, int plane) {
    using V = std::conditional_t<N == 8, typename SplitT<SP>::v8, typename SplitT<SP>::v4>;
    V h, l;
#pragma unroll
    for (int e = 0; e < N; ++e) {
        typename SplitT<SP>::e he, le;
        split_elem<SP>(v[e], lo_scale, he, le);
        h[e] = he;
        l[e] = le;
    }
    *(V*)p = h;
    if (plane) *(V*)(p + plane) = l;  // plane == 0: the single-plane fp16 format (VGH_FMT_F16, r05) -- the value is its hi plane alone
}
template <int SP, int N>
__device__ __forceinline__ void join_load(const uint16_t* p, int plane, float lo_inv, float (&v)[N]) {
    using V = std::conditional_t<N == 8, typename SplitT<SP>::v8, typename SplitT<SP>::v4>;
    const V h = *(const V*)p;
    if (!plane) {  // single-plane fp16 (VGH_FMT_F16)
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = (float)h[e];
        return;
    }
    const V l = *(const V*)(p + plane);
#pragma unroll
    for (int e = 0; e < N; ++e) v[e] = (float)h[e] + (float)l[e] * lo_inv;
}
