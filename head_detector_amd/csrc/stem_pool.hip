// Stem conv (3 -> 48, 3x3 stride 2) and SPP max-pool kernels for gfx950.
//
// Stem replaces YoloNASStem = QARepVGGBlock(3, 48, stride=2) (arch yaml :8-10) folded to one conv, fused with
// the `.float() / 255.0` of HeadDetector._transform_image (head_detector/detector.py:51) when the input is u8.
// K = 27 is far too short for an MFMA pipeline and the layer is HBM/issue bound, so it
// is an exact-fp32 VALU kernel: image tile in LDS, weights through the scalar cache (wave-uniform).
//
// SPP pool replaces the three nn.MaxPool2d(k, 1, k//2), k = 5, 9, 13 of SG's SPP (arch yaml :41-45) using the
// exact cascade pool13 = pool5(pool5(pool5(x))), and writes the results next to x (concat-by-offset).
#include "vgh_internal.h"
#include "split_fmt.h"

namespace {

constexpr int ST = 16;            // 16x16 output pixels per block
constexpr int SIN = 2 * ST + 1;   // 33x33 input patch
constexpr int STEM_CO = 48, STEM_CP = 64;

// SP != 0 (split parity modes, STAGE = 0 only): the exact fp32 result leaves as a hi and a lo 16-bit plane (split_fmt.h)
// NCH (STAGE = 1): 16-byte chunks stored per pixel -- 8: the 48 channels + 16 zero channels of a 64-channel pitch (the K padding of the next conv);
// 6: the 48 channels alone at a 48-channel pitch (r04: 96-byte pixels, the next conv reads a 64-channel window whose last 16 channels are the
// neighbouring pixel's first 16 and meet all-zero weight rows -- net.hip checks that -- so the 210 MB of zeros per 64 images are neither written nor read)
// H16 (STAGE = 1, SP = 0): the staged path storing ONE fp16 plane (VGH_FMT_F16, r05) instead of bf16 -- same LDS transpose, same full-line stores
template <int FMT, int STAGE, int SP = 0, int NCH = 8, int H16 = 0>
__global__ __launch_bounds__(256, 4) void stem_kernel(const void* __restrict__ image, int H, int W, const float* __restrict__ wgt /*[27][48]*/,
                                                   const float* __restrict__ bias /*[48]*/, uint16_t* __restrict__ out, int64_t out_pitch,
                                                   int out_coff, int plane, float lo_scale) {
    static_assert(SP == 0 || STAGE == 0, "the split modes store directly");
    // the 32 KiB output staging [pixel][chunk ^ (pixel & 7)] aliases the input patch (dead once every lane holds its 27 taps)
    __shared__ __attribute__((aligned(16))) char smem_raw[STAGE ? 256 * 8 * 16 : 3 * SIN * (SIN + 1) * 4];
    float (*patch)[SIN][SIN + 1] = (float (*)[SIN][SIN + 1]) smem_raw;
    bf16x8_t* stage = (bf16x8_t*)smem_raw;
    __shared__ float lut[256];  // u8 -> float(v) / 255.0f, the true (correctly rounded) division of detector.py:51, once per block
    const int Ho = H / 2, Wo = W / 2;
    const int b = blockIdx.z, ty = blockIdx.y * ST, tx = blockIdx.x * ST;
    const int tid = threadIdx.x;
    if (FMT == VGH_IMG_U8_NHWC) lut[tid] = (float)tid / 255.0f;  // visible after the barrier that follows the load issue
    const int iy_base = ty * 2 - 1, ix_base = tx * 2 - 1;
    // fixed trip count + full unroll: all 13 loads of a lane are in flight together (as a rolled loop every iteration exposed
    // one full memory round trip: 13 x ~1 us per block)
    constexpr int NLD = (3 * SIN * SIN + 255) / 256;
    float pv[NLD];
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
        const int e = tid + it * 256;
        int ci, r, c;
        if (FMT == VGH_IMG_F32_NCHW) {
            ci = e / (SIN * SIN);
            const int rem = e - ci * SIN * SIN;
            r = rem / SIN;
            c = rem - r * SIN;
        } else {
            r = e / (SIN * 3);
            const int rem = e - r * SIN * 3;
            c = rem / 3;
            ci = rem - c * 3;
        }
        const int iy = iy_base + r, ix = ix_base + c;
        float v = 0.0f;
        int q = -1;
        if (e < 3 * SIN * SIN && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
            if (FMT == VGH_IMG_F32_NCHW)
                v = ((const float*)image)[(((int64_t)b * 3 + ci) * H + iy) * W + ix];
            else
                q = ((const uint8_t*)image)[(((int64_t)b * H + iy) * W + ix) * 3 + ci];
        }
        pv[it] = (FMT == VGH_IMG_F32_NCHW) ? v : __int_as_float(q);
    }
    if (FMT == VGH_IMG_U8_NHWC) __syncthreads();
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
        const int e = tid + it * 256;
        if (e >= 3 * SIN * SIN) break;
        int ci, r, c;
        if (FMT == VGH_IMG_F32_NCHW) {
            ci = e / (SIN * SIN);
            const int rem = e - ci * SIN * SIN;
            r = rem / SIN;
            c = rem - r * SIN;
        } else {
            r = e / (SIN * 3);
            const int rem = e - r * SIN * 3;
            c = rem / 3;
            ci = rem - c * 3;
        }
        float v = pv[it];
        if (FMT == VGH_IMG_U8_NHWC) {
            const int q = __float_as_int(pv[it]);
            v = q >= 0 ? lut[q] : 0.0f;  // (float)u8 / 255.0f, detector.py:51
        }
        patch[ci][r][c] = v;
    }
    __syncthreads();
    const int ly = tid / ST, lx = tid % ST;
    float x[27];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) x[(ky * 3 + kx) * 3 + ci] = patch[ci][2 * ly + ky][2 * lx + kx];
    if (STAGE) __syncthreads();  // the patch is dead: its memory becomes the staging buffer
    const int oy_l = ty + ly, ox_l = tx + lx;
    const bool ok = oy_l < Ho && ox_l < Wo;
    uint16_t* op = out + (((int64_t)b * Ho + oy_l) * Wo + ox_l) * out_pitch + out_coff;
    // Results leave through an LDS transpose: a lane owns one pixel = 128 B (48 channels + 16 zero channels), so direct stores
    // would touch 64 different cache lines with 16 B each per instruction (measured: 1.9 TB/s).  Staged as [pixel][8 chunks]
    // (chunk slot XOR-swizzled by the pixel so the strided writes spread over the banks), 8 consecutive lanes then write
    // one whole 128-B line.
    // two output channels per instruction: <2 x float> fma = v_pk_fma_f32 (each lane an IEEE fma, bit-identical to fmaf)
#pragma unroll
    for (int cg = 0; cg < STEM_CO / 16; ++cg) {
        f32x2_t acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = f32x2_t{0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            const f32x2_t xk = {x[k], x[k]};
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = __builtin_elementwise_fma(xk, *(const f32x2_t*)(wgt + k * STEM_CO + cg * 16 + 2 * c), acc[c]);
        }
        if constexpr (SP != 0) {
            float r0[8], r1[8];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                r0[2 * c] = fmaxf(acc[c][0] + bias[cg * 16 + 2 * c], 0.0f);
                r0[2 * c + 1] = fmaxf(acc[c][1] + bias[cg * 16 + 2 * c + 1], 0.0f);
                r1[2 * c] = fmaxf(acc[4 + c][0] + bias[cg * 16 + 8 + 2 * c], 0.0f);
                r1[2 * c + 1] = fmaxf(acc[4 + c][1] + bias[cg * 16 + 8 + 2 * c + 1], 0.0f);
            }
            if (ok) {
                split_store<SP, 8>(r0, lo_scale, op + cg * 16, plane);
                split_store<SP, 8>(r1, lo_scale, op + cg * 16 + 8, plane);
            }
            continue;
        }
        bf16x8_t o0, o1;
        // ReLU through fmaxf also turns a NaN into 0.  The 48-channel pitch (NCH = 6) additionally needs FINITE stores: the stage-1 downsample reads 16 channels of the
        // NEXT pixel / image over zero weight columns (net.hip), and Inf x 0 would carry one image's fault into its neighbour -- only a float image can overflow (a u8
        // image bounds |out| by sum|w| + |b|), so only that instantiation pays the clamp to bf16's largest finite value
        auto act = [](float v) {
            v = fmaxf(v, 0.0f);
            if constexpr (FMT == VGH_IMG_F32_NCHW && NCH == 6) v = fminf(v, 3.3895313892515355e38f);
            if constexpr (H16) {  // the 16 bits of the fp16 value travel in the bf16-typed staging vector; values beyond fp16's range saturate
                const _Float16 h = (_Float16)fminf(v, 65504.0f);
                return __builtin_bit_cast(__bf16, h);
            } else {
                return (__bf16)v;
            }
        };
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            o0[2 * c] = act(acc[c][0] + bias[cg * 16 + 2 * c]);
            o0[2 * c + 1] = act(acc[c][1] + bias[cg * 16 + 2 * c + 1]);
            o1[2 * c] = act(acc[4 + c][0] + bias[cg * 16 + 8 + 2 * c]);
            o1[2 * c + 1] = act(acc[4 + c][1] + bias[cg * 16 + 8 + 2 * c + 1]);
        }
        if (STAGE) {
            stage[tid * 8 + ((2 * cg) ^ (tid & 7))] = o0;
            stage[tid * 8 + ((2 * cg + 1) ^ (tid & 7))] = o1;
        } else if (ok) {
            *(bf16x8_t*)(op + cg * 16) = o0;
            *(bf16x8_t*)(op + cg * 16 + 8) = o1;
        }
    }
    bf16x8_t z;
#pragma unroll
    for (int c = 0; c < 8; ++c) z[c] = (__bf16)0.0f;
    if (!STAGE) {
        if (ok) {
            *(bf16x8_t*)(op + 48) = z;  // channels 48..63: exact zeros (K padding of the next conv)
            *(bf16x8_t*)(op + 56) = z;
            if (SP != 0) {  // zero bits are zero in bf16 and fp16 alike
                *(bf16x8_t*)(op + plane + 48) = z;
                *(bf16x8_t*)(op + plane + 56) = z;
            }
        }
        return;
    }
    if constexpr (NCH == 8) {
        stage[tid * 8 + (6 ^ (tid & 7))] = z;
        stage[tid * 8 + (7 ^ (tid & 7))] = z;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        const int e = it * 256 + tid;
        const int p = e / NCH, ch = e - p * NCH;  // consecutive lanes write consecutive 16-byte chunks: whole lines also across the 96-byte pixels of NCH = 6
        const int oy = ty + (p / ST), ox = tx + (p % ST);
        if (oy < Ho && ox < Wo)
            *(bf16x8_t*)(out + (((int64_t)b * Ho + oy) * Wo + ox) * out_pitch + out_coff + ch * 8) = stage[p * 8 + (ch ^ (p & 7))];
    }
}

#ifdef VGH_EXPERIMENTS  // measured slower than the VALU kernel (442 vs 306 us per 64 images): experiments build only since r06
// ---- stem on the matrix cores (r05, bf16 throughput mode, u8 images) ----------------------------------------------------------------------
// The VALU kernel above is exact fp32 and VALU-bound: 27 x 48 FMAs per output pixel = 17 GFLOP per 64 images at ~35 % of the vector rate is 0.31 ms, 2.6 x the
// time its 0.7 GB of traffic needs.  In the bf16 mode the stem's OUTPUT is rounded to bf16 anyway, so here the layer is a K = 27 (padded
// to 32) bf16 GEMM: a u8 pixel value is exact in bf16 (8 significant bits), the /255 of detector.py:51 is folded into the weights (w / 255 rounded to bf16 --
// the one new rounding, 2^-9 relative on a weight, below the bf16 rounding of the output), fp32 accumulate from the bias.  Per 32-pixel group and wave:
// 4 x v_mfma_f32_32x32x16_bf16 (two cout groups x two k slices) instead of 648 v_pk_fma_f32 per lane.  Same 16 x 16 tile, same LDS-transposed full-line stores.
// The parity modes (and float images) keep the exact kernel.
__device__ __forceinline__ unsigned stem_pack_bf16(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    const f32x2_t f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf2));
}
template <int NCH>
__global__ __launch_bounds__(256, 4) void stem_mfma_kernel(const uint8_t* __restrict__ image, int H, int W, const float* __restrict__ wgt /*[27][48]*/, const float* __restrict__ bias /*[48]*/,
                                                        uint16_t* __restrict__ out, int64_t out_pitch, int out_coff) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    constexpr int PR = 100;  // bf16 values per patch row: 33 pixels x 3 channels (+1)
    __shared__ __attribute__((aligned(16))) char smem_raw[256 * 8 * 16];  // the 32 KiB output staging aliases the 6.6 KB patch (dead once the fragments are in registers)
    uint16_t* patch = (uint16_t*)smem_raw;                               // [33][PR]: bf16(u8 value) of image pixel (2 ty - 1 + r, 2 tx - 1 + c), channel ci at [r][3 c + ci]
    bf16x8_t* stage = (bf16x8_t*)smem_raw;
    const int Ho = H / 2, Wo = W / 2;
    const int b = blockIdx.z, ty = blockIdx.y * ST, tx = blockIdx.x * ST;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n32 = lane & 31, hi = lane >> 5;
    const int iy_base = ty * 2 - 1, ix_base = tx * 2 - 1;
    // ---- weights: this lane's A fragments.  Lane (n32, hi) of cout group i, k slice s holds w[k = 16 s + 8 hi + e][cout = 32 i + n32] / 255, e = 0 .. 7 ----
    float wf[2][2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 16 * sl + 8 * hi + e, co = 32 * i + n32;
                wf[i][sl][e] = (k < 27 && co < STEM_CO) ? wgt[k * STEM_CO + co] : 0.0f;
            }
    // ---- image patch: 33 x 99 bytes, one byte per load, all of a lane's loads in flight together ----
    constexpr int NLD = (3 * SIN * SIN + 255) / 256;
    int pq[NLD];
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
        const int e = tid + it * 256;
        const int r = e / (SIN * 3), rem = e - r * SIN * 3, c = rem / 3;
        const int iy = iy_base + r, ix = ix_base + c;
        int q = 0;
        if (e < 3 * SIN * SIN && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) q = image[(((int64_t)b * H + iy) * W) * 3 + (int64_t)ix_base * 3 + rem];
        pq[it] = q;
    }
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
        const int e = tid + it * 256;
        if (e >= 3 * SIN * SIN) break;
        const int r = e / (SIN * 3), rem = e - r * SIN * 3;
        patch[r * PR + rem] = (uint16_t)(__float_as_uint((float)pq[it]) >> 16);  // 0 .. 255 is exact in bf16
    }
    bf16x8_t afr[2][2];
    const float inv255 = 1.0f / 255.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const u32x4 v = {stem_pack_bf16(wf[i][sl][0] * inv255, wf[i][sl][1] * inv255), stem_pack_bf16(wf[i][sl][2] * inv255, wf[i][sl][3] * inv255),
                             stem_pack_bf16(wf[i][sl][4] * inv255, wf[i][sl][5] * inv255), stem_pack_bf16(wf[i][sl][6] * inv255, wf[i][sl][7] * inv255)};
            afr[i][sl] = __builtin_bit_cast(bf16x8_t, v);
        }
    __syncthreads();
    // ---- B fragments: pixel p = 64 wv + 32 g + n32 of the tile, k = 16 s + 8 hi + e -> patch[(2 ly + k / 9) * PR + 6 lx + k % 9] ----
    bf16x8_t bfr[2][2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int pidx = 64 * wv + 32 * g + n32, ly = pidx / ST, lx = pidx % ST;
        const uint16_t* pb = patch + (2 * ly) * PR + 6 * lx;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            unsigned hv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k0 = 16 * sl + e, k1 = k0 + 8;  // this element's k for hi = 0 / hi = 1 (k1 < 27 unless sl = 1 and e >= 3)
                const int o0 = (k0 / 9) * PR + k0 % 9, o1 = k1 < 27 ? (k1 / 9) * PR + k1 % 9 : 0;
                const unsigned v = pb[hi ? o1 : o0];
                hv[e] = (hi && k1 >= 27) ? 0u : v;
            }
            const u32x4 v = {hv[0] | (hv[1] << 16), hv[2] | (hv[3] << 16), hv[4] | (hv[5] << 16), hv[6] | (hv[7] << 16)};
            bfr[g][sl] = __builtin_bit_cast(bf16x8_t, v);
        }
    }
    __syncthreads();  // the patch is dead: its memory becomes the staging buffer
    // ---- accumulate from the bias: register 4 q + e of lane (n32, hi) is cout 32 i + 8 q + 4 hi + e of pixel n32 ----
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = 32 * i + 8 * q + 4 * hi + e;
                const float bv = co < STEM_CO ? bias[co] : 0.0f;
                acc[0][i][4 * q + e] = bv;
                acc[1][i][4 * q + e] = bv;
            }
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) acc[g][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[i][sl], bfr[g][sl], acc[g][i], 0, 0, 0);
    // ---- ReLU, bf16, half-wave exchange: lane (n32, hi) ends with couts 32 i + 16 m + 8 hi .. + 7 of its pixel = 16-byte chunk 4 i + 2 m + hi ----
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int pidx = 64 * wv + 32 * g + n32;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                if (4 * i + 2 * m >= NCH) continue;
                unsigned pa0 = stem_pack_bf16(fmaxf(acc[g][i][8 * m + 0], 0.0f), fmaxf(acc[g][i][8 * m + 1], 0.0f)), pa1 = stem_pack_bf16(fmaxf(acc[g][i][8 * m + 2], 0.0f), fmaxf(acc[g][i][8 * m + 3], 0.0f));
                unsigned pb0 = stem_pack_bf16(fmaxf(acc[g][i][8 * m + 4], 0.0f), fmaxf(acc[g][i][8 * m + 5], 0.0f)), pb1 = stem_pack_bf16(fmaxf(acc[g][i][8 * m + 6], 0.0f), fmaxf(acc[g][i][8 * m + 7], 0.0f));
                auto r0 = __builtin_amdgcn_permlane32_swap(pa0, pb0, false, false);  // lanes 32-63 of pa trade places with lanes 0-31 of pb
                auto r1 = __builtin_amdgcn_permlane32_swap(pa1, pb1, false, false);
                const u32x4 v = {r0[0], r1[0], r0[1], r1[1]};
                const int ch = 4 * i + 2 * m + hi;
                stage[pidx * 8 + (ch ^ (pidx & 7))] = __builtin_bit_cast(bf16x8_t, v);
            }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        const int e = it * 256 + tid;
        const int p = e / NCH, ch = e - p * NCH;  // consecutive lanes write consecutive 16-byte chunks: whole lines also across the 96-byte pixels of NCH = 6
        const int oy = ty + (p / ST), ox = tx + (p % ST);
        if (oy < Ho && ox < Wo)
            *(bf16x8_t*)(out + (((int64_t)b * Ho + oy) * Wo + ox) * out_pitch + out_coff + ch * 8) = stage[p * 8 + (ch ^ (p & 7))];
    }
}

#endif  // VGH_EXPERIMENTS

// ---- SPP ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bf16x8_t max8(bf16x8_t a, bf16x8_t b) {
    bf16x8_t r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = ((float)a[e] >= (float)b[e]) ? a[e] : b[e];
    return r;
}

// one block = one image x CG channels; LDS [H*W][CG] bf16 x 2.  A 5x5 max is separable: 5 taps along x into T, 5 taps along y
// back into X (10 LDS reads per element instead of 25); pool9 / pool13 are the exact cascade pool5(pool5(.)) / pool5^3.
__global__ __launch_bounds__(256) void spp_pool_kernel(uint16_t* __restrict__ buf, int64_t pitch, int coff, int C, int H, int W, int CG) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int HW = H * W;
    const int cv = CG / 8;  // 16-byte vectors per pixel in this group
    bf16x8_t* X = (bf16x8_t*)smem;
    bf16x8_t* T = X + (size_t)HW * cv;
    const int b = blockIdx.y, cg0 = blockIdx.x * CG;
    uint16_t* base = buf + (int64_t)b * HW * pitch + coff + cg0;
    const int n = HW * cv;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int p = e / cv, v = e - p * cv;
        X[e] = *(const bf16x8_t*)(base + (int64_t)p * pitch + v * 8);
    }
    __syncthreads();
    for (int pass = 0; pass < 3; ++pass) {
        for (int e = threadIdx.x; e < n; e += blockDim.x) {
            const int p = e / cv, v = e - p * cv;
            const int y = p / W, x = p - y * W;
            bf16x8_t m = X[e];
#pragma unroll
            for (int dx = -2; dx <= 2; ++dx) {
                const int xx = x + dx;
                if (dx != 0 && (unsigned)xx < (unsigned)W) m = max8(m, X[(y * W + xx) * cv + v]);
            }
            T[e] = m;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < n; e += blockDim.x) {
            const int p = e / cv, v = e - p * cv;
            const int y = p / W, x = p - y * W;
            bf16x8_t m = T[e];
#pragma unroll
            for (int dy = -2; dy <= 2; ++dy) {
                const int yy = y + dy;
                if (dy != 0 && (unsigned)yy < (unsigned)H) m = max8(m, T[(yy * W + x) * cv + v]);
            }
            X[e] = m;
            *(bf16x8_t*)(base + (int64_t)p * pitch + (int64_t)(pass + 1) * C + v * 8) = m;
        }
        __syncthreads();
    }
}

// Split parity modes: the same separable cascade on the JOINED fp32 values (hi + lo / L); the maximum is one of the inputs, so
// re-splitting it reproduces that input's two planes bit for bit.  LDS [H*W][CG] fp32 x 2.
template <int SP>
__global__ __launch_bounds__(256) void spp_pool_split_kernel(uint16_t* __restrict__ buf, int64_t pitch, int plane, int coff, int C, int H, int W, int CG, float lo_scale,
                                                             float lo_inv) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int HW = H * W;
    float* X = (float*)smem;
    float* T = X + (size_t)HW * CG;
    const int b = blockIdx.y, cg0 = blockIdx.x * CG;
    uint16_t* base = buf + (int64_t)b * HW * pitch + coff + cg0;
    const int cv = CG / 4, n = HW * cv;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int p = e / cv, v = e - p * cv;
        float f[4];
        join_load<SP, 4>(base + (int64_t)p * pitch + v * 4, plane, lo_inv, f);
        *(f32x4_t*)(X + (size_t)p * CG + v * 4) = f32x4_t{f[0], f[1], f[2], f[3]};
    }
    __syncthreads();
    for (int pass = 0; pass < 3; ++pass) {
        for (int e = threadIdx.x; e < n; e += blockDim.x) {
            const int p = e / cv, v = e - p * cv;
            const int y = p / W, x = p - y * W;
            f32x4_t m = *(const f32x4_t*)(X + (size_t)p * CG + v * 4);
#pragma unroll
            for (int dx = -2; dx <= 2; ++dx) {
                const int xx = x + dx;
                if (dx != 0 && (unsigned)xx < (unsigned)W) {
                    const f32x4_t o = *(const f32x4_t*)(X + (size_t)(y * W + xx) * CG + v * 4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) m[k] = m[k] >= o[k] ? m[k] : o[k];
                }
            }
            *(f32x4_t*)(T + (size_t)p * CG + v * 4) = m;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < n; e += blockDim.x) {
            const int p = e / cv, v = e - p * cv;
            const int y = p / W, x = p - y * W;
            f32x4_t m = *(const f32x4_t*)(T + (size_t)p * CG + v * 4);
#pragma unroll
            for (int dy = -2; dy <= 2; ++dy) {
                const int yy = y + dy;
                if (dy != 0 && (unsigned)yy < (unsigned)H) {
                    const f32x4_t o = *(const f32x4_t*)(T + (size_t)(yy * W + x) * CG + v * 4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) m[k] = m[k] >= o[k] ? m[k] : o[k];
                }
            }
            *(f32x4_t*)(X + (size_t)p * CG + v * 4) = m;
            const float mv[4] = {m[0], m[1], m[2], m[3]};
            split_store<SP, 4>(mv, lo_scale, base + (int64_t)p * pitch + (int64_t)(pass + 1) * C + v * 4, plane);
        }
        __syncthreads();
    }
}

}  // namespace

#include <atomic>
// default OFF: measured r05 (tools/ab_knob.py vgh_stem_set_mfma, one box, alternating) the matrix-core stem is correct but SLOWER -- 442 vs 306 us per 64 images
// (M b32: 232 vs 168 us; two-lane forward L b64 12.47 vs 12.25 ms): with 25 600 blocks of 256 pixels the kernel is a chain of latencies per block (image bytes ->
// LDS -> 32 scattered 2-byte LDS gathers per lane -> 8 MFMAs -> LDS transpose -> stores), not the VALU-bound loop the FMA count suggested (EXPERIMENTS.md 8e)
#ifdef VGH_EXPERIMENTS
static std::atomic<int> g_stem_mfma{0};
extern "C" int vgh_stem_set_mfma(int on) {
    g_stem_mfma.store(on ? 1 : 0, std::memory_order_relaxed);
    return VGH_OK;
}
#endif

int vgh_launch_stem(const void* image, int image_fmt, int B, int H, int W, const float* w, const float* bias, uint16_t* out, int64_t out_pitch,
                    int out_coff, int store_ch, int fmt, int plane, hipStream_t stream) {
    VGH_REQUIRE(store_ch == 64 || (store_ch == 48 && fmt == VGH_FMT_BF16), "stem: stores 64 channels (48 + 16 zeros), or 48 in the bf16 mode; got %d", store_ch);
    VGH_REQUIRE(H % 2 == 0 && W % 2 == 0, "stem: image size must be even");
    VGH_REQUIRE(out_pitch % 8 == 0 && out_coff % 8 == 0 && plane % 8 == 0, "stem: output alignment");
    VGH_REQUIRE(image_fmt == VGH_IMG_F32_NCHW || image_fmt == VGH_IMG_U8_NHWC, "stem: unknown image format %d", image_fmt);
    if (B == 0) return VGH_OK;
    dim3 grid((W / 2 + ST - 1) / ST, (H / 2 + ST - 1) / ST, B);
    const bool u8 = image_fmt == VGH_IMG_U8_NHWC;
#define VGH_STEM_LAUNCH(FMT, STAGE, SP, LO) \
    hipLaunchKernelGGL((stem_kernel<FMT, STAGE, SP>), grid, dim3(256), 0, stream, image, H, W, w, bias, out, out_pitch, out_coff, plane, LO)
#define VGH_STEM_LAUNCH48(FMT) \
    hipLaunchKernelGGL((stem_kernel<FMT, 1, 0, 6>), grid, dim3(256), 0, stream, image, H, W, w, bias, out, out_pitch, out_coff, plane, 1.0f)
    if (fmt == VGH_FMT_F16X2 && plane == 0 && store_ch == 64) {  // single-plane fp16 (VGH_FMT_F16): the staged kernel with fp16 stores (zero bits are zero in both formats)
        if (u8)
            hipLaunchKernelGGL((stem_kernel<VGH_IMG_U8_NHWC, 1, 0, 8, 1>), grid, dim3(256), 0, stream, image, H, W, w, bias, out, out_pitch, out_coff, plane, 1.0f);
        else
            hipLaunchKernelGGL((stem_kernel<VGH_IMG_F32_NCHW, 1, 0, 8, 1>), grid, dim3(256), 0, stream, image, H, W, w, bias, out, out_pitch, out_coff, plane, 1.0f);
    } else if (fmt == VGH_FMT_F16X2) {
        if (u8) VGH_STEM_LAUNCH(VGH_IMG_U8_NHWC, 0, VGH_FMT_F16X2, 2048.0f);
        else VGH_STEM_LAUNCH(VGH_IMG_F32_NCHW, 0, VGH_FMT_F16X2, 2048.0f);
    } else if (fmt == VGH_FMT_BF16X2) {
        if (u8) VGH_STEM_LAUNCH(VGH_IMG_U8_NHWC, 0, VGH_FMT_BF16X2, 1.0f);
        else VGH_STEM_LAUNCH(VGH_IMG_F32_NCHW, 0, VGH_FMT_BF16X2, 1.0f);
    } else {
        VGH_REQUIRE(fmt == VGH_FMT_BF16, "stem: output format %d", fmt);
#ifdef VGH_EXPERIMENTS
        if (u8 && g_stem_mfma.load(std::memory_order_relaxed)) {  // bf16 mode, u8 image: the layer on the matrix cores (stem_mfma_kernel)
            if (store_ch == 48)
                hipLaunchKernelGGL((stem_mfma_kernel<6>), grid, dim3(256), 0, stream, (const uint8_t*)image, H, W, w, bias, out, out_pitch, out_coff);
            else
                hipLaunchKernelGGL((stem_mfma_kernel<8>), grid, dim3(256), 0, stream, (const uint8_t*)image, H, W, w, bias, out, out_pitch, out_coff);
            VGH_HIP(hipGetLastError());
            return VGH_OK;
        }
#endif
        if (store_ch == 48) {
            if (u8) VGH_STEM_LAUNCH48(VGH_IMG_U8_NHWC);
            else VGH_STEM_LAUNCH48(VGH_IMG_F32_NCHW);
            VGH_HIP(hipGetLastError());
            return VGH_OK;
        }
#ifdef VGH_EXPERIMENTS
        static const int stage = getenv("VGH_STEM_STAGE") ? atoi(getenv("VGH_STEM_STAGE")) : 1;  // A/B switch: 0 = direct per-lane stores
        if (!stage) {
            if (u8) VGH_STEM_LAUNCH(VGH_IMG_U8_NHWC, 0, 0, 1.0f);
            else VGH_STEM_LAUNCH(VGH_IMG_F32_NCHW, 0, 0, 1.0f);
        } else
#endif
        {
            if (u8) VGH_STEM_LAUNCH(VGH_IMG_U8_NHWC, 1, 0, 1.0f);
            else VGH_STEM_LAUNCH(VGH_IMG_F32_NCHW, 1, 0, 1.0f);
        }
    }
#undef VGH_STEM_LAUNCH
#undef VGH_STEM_LAUNCH48
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}

int vgh_launch_spp_pool(uint16_t* buf, int64_t pitch, int coff, int C, int B, int H, int W, int fmt, int plane, hipStream_t stream) {
    VGH_REQUIRE(C % 8 == 0 && pitch % 8 == 0 && coff % 8 == 0, "spp: channel alignment");
    if (B == 0) return VGH_OK;
    if (fmt == VGH_FMT_BF16X2 || fmt == VGH_FMT_F16X2) {
        int CG = 16;
        while (CG > 4 && ((size_t)2 * H * W * CG * 4 > 64 * 1024 || C % CG != 0)) CG /= 2;
        VGH_REQUIRE(C % CG == 0 && (size_t)2 * H * W * CG * 4 <= 160 * 1024, "spp: feature map %dx%d too large for the LDS tile", H, W);
        const size_t lds = (size_t)2 * H * W * CG * 4;
        if (fmt == VGH_FMT_F16X2) {
            if (lds > 64 * 1024) VGH_HIP(hipFuncSetAttribute((const void*)spp_pool_split_kernel<VGH_FMT_F16X2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(spp_pool_split_kernel<VGH_FMT_F16X2>, dim3(C / CG, B), dim3(256), lds, stream, buf, pitch, plane, coff, C, H, W, CG, 2048.0f, 1.0f / 2048.0f);
        } else {
            if (lds > 64 * 1024) VGH_HIP(hipFuncSetAttribute((const void*)spp_pool_split_kernel<VGH_FMT_BF16X2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(spp_pool_split_kernel<VGH_FMT_BF16X2>, dim3(C / CG, B), dim3(256), lds, stream, buf, pitch, plane, coff, C, H, W, CG, 1.0f, 1.0f);
        }
        VGH_HIP(hipGetLastError());
        return VGH_OK;
    }
    VGH_REQUIRE(fmt == VGH_FMT_BF16, "spp: format %d", fmt);
    int CG = 16;  // 16 channels per block: B * C/16 blocks (768 for the M net at B = 32) of 25 KiB LDS at 20x20
    while (CG > 8 && ((size_t)2 * H * W * CG * 2 > 64 * 1024 || C % CG != 0)) CG /= 2;
    VGH_REQUIRE(C % CG == 0 && (size_t)2 * H * W * CG * 2 <= 160 * 1024, "spp: feature map %dx%d too large for the LDS tile", H, W);
    const size_t lds = (size_t)2 * H * W * CG * 2;
    if (lds > 64 * 1024) VGH_HIP(hipFuncSetAttribute((const void*)spp_pool_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(spp_pool_kernel, dim3(C / CG, B), dim3(256), lds, stream, buf, pitch, coff, C, H, W, CG);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}
