// Fused stem + stage-1 downsample for gfx950: YoloNASStem (QARepVGG 3 -> 48, 3x3 stride 2, arch yaml :8-10) and the first
// backbone downsample (QARepVGG 48 -> 96, 3x3 stride 2, arch yaml :11-17) of the network behind head_detector/detector.py:58-59,
// in ONE kernel: the 320 x 320 x 48 stem activation never goes to HBM.
//
// Why (profiles/r03_per_layer_l64.json, L b64): the stem (0.32 ms) writes 839 MB -- 48 channels padded to 64 so that the next conv's K blocks are
// whole -- and the downsample (0.38 ms) reads them back: 0.3 ms of pure HBM traffic for a tensor with one consumer.  Here a block computes the
// 9 x 33 stem pixels a 4 x 16 output tile needs (halo included: 4.64 stem pixels per output pixel instead of 4) into LDS and runs the stride-2 conv from
// there.  The stem stays what it was -- an exact fp32 FMA chain in ascending k per output (image / 255 by the correctly rounded division), bias, ReLU, ONE
// rounding to bf16 -- and the conv accumulates its 27 k16 steps in the order of the implicit-GEMM kernel (tap major, channels ascending; the 16 zero
// channels of the padded tensor only ever added exact zeros), so the fused result is BIT-IDENTICAL to the two-kernel path (tests/test_gpu_parity.py::
// test_fused_stem_downsample_is_bit_identical).
//
// Block = 5 waves: 297 stem pixels on 320 lanes (fp32 VALU, weights through the scalar cache), then waves 0..2 own one 32-cout group each x both 32-pixel MFMA
// groups: 54 x v_mfma_f32_32x32x16_bf16, A fragments straight from L2 (the 83 KB weight image is re-read per tile: 36 GB/s per CU), B fragments from the LDS
// stem tile (pixel pitch 112 B: a stride-2 gather is at worst 2-way bank conflicted).  49 KB of LDS: three blocks per CU, so one block's VALU phase runs
// under the others' MFMA / store phases.  The kernel is VALU-bound by design: 17 GFLOP of stem FMAs per 64 images.
#include "vgh_internal.h"

namespace {

constexpr int FD_TH = 4, FD_TW = 16;                      // output tile (pixels of the downsample map)
constexpr int FD_SH = 2 * FD_TH + 1, FD_SW = 2 * FD_TW + 1;  // stem pixels it needs: 9 x 33
constexpr int FD_IH = 2 * FD_SH + 1, FD_IW = 2 * FD_SW + 1;  // image pixels they need: 19 x 67
constexpr int FD_NS = FD_SH * FD_SW;                      // 297
constexpr int FD_XP = 112;                                // LDS bytes per stem pixel: 48 bf16 + 16 B pad
constexpr int FD_THREADS = 320;
constexpr int FD_CO = 48, FD_CD = 96;                     // stem / downsample output channels

struct StemDsArgs {
    const void* image;
    const float* wstem;   // [27][48]
    const float* bstem;   // [48]
    const uint16_t* wds;  // bf16 [9 taps][3 k16 chunks][96 couts][16]
    const float* bds;     // [96]
    uint16_t* out;        // bf16 NHWC
    int64_t out_pitch;
    int out_coff;
    int H, W;             // image size
};

template <int FMT>
__global__ __launch_bounds__(FD_THREADS, 4) void stem_ds_kernel(const StemDsArgs a, const float* __restrict__ wstem /*[27][48]*/, const float* __restrict__ bstem /*[48]*/) {
    __shared__ __attribute__((aligned(16))) float img[3][FD_IH][FD_IW + 1];  // 15.5 KB; reused as the epilogue's transpose strips
    __shared__ __attribute__((aligned(16))) char X[FD_NS * FD_XP];           // 33.3 KB
    __shared__ float lut[256];
    const int Hs = a.H / 2, Ws = a.W / 2, Ho = a.H / 4, Wo = a.W / 4;
    const int b = blockIdx.z, oy0 = blockIdx.y * FD_TH, ox0 = blockIdx.x * FD_TW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (FMT == VGH_IMG_U8_NHWC && tid < 256) lut[tid] = (float)tid / 255.0f;  // the true division of detector.py:51, once per block
    // ---- image patch: rows 4*oy0 - 3 .. +18, columns 4*ox0 - 3 .. +66 (zeros outside the image = the stem conv's own padding) ----
    const int iy0 = 4 * oy0 - 3, ix0 = 4 * ox0 - 3;
    constexpr int NE = 3 * FD_IH * FD_IW, NLD = (NE + FD_THREADS - 1) / FD_THREADS;
    float pv[NLD];
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
        const int e = tid + it * FD_THREADS;
        int ci, r, c;
        if (FMT == VGH_IMG_F32_NCHW) {
            ci = e / (FD_IH * FD_IW);
            const int rem = e - ci * FD_IH * FD_IW;
            r = rem / FD_IW;
            c = rem - r * FD_IW;
        } else {
            r = e / (FD_IW * 3);
            const int rem = e - r * FD_IW * 3;
            c = rem / 3;
            ci = rem - c * 3;
        }
        const int iy = iy0 + r, ix = ix0 + c;
        float v = 0.0f;
        int q = -1;
        if (e < NE && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
            if (FMT == VGH_IMG_F32_NCHW)
                v = ((const float*)a.image)[(((int64_t)b * 3 + ci) * a.H + iy) * a.W + ix];
            else
                q = ((const uint8_t*)a.image)[(((int64_t)b * a.H + iy) * a.W + ix) * 3 + ci];
        }
        pv[it] = (FMT == VGH_IMG_F32_NCHW) ? v : __int_as_float(q);
    }
    if (FMT == VGH_IMG_U8_NHWC) __syncthreads();  // lut visible
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
        const int e = tid + it * FD_THREADS;
        if (e >= NE) break;
        int ci, r, c;
        if (FMT == VGH_IMG_F32_NCHW) {
            ci = e / (FD_IH * FD_IW);
            const int rem = e - ci * FD_IH * FD_IW;
            r = rem / FD_IW;
            c = rem - r * FD_IW;
        } else {
            r = e / (FD_IW * 3);
            const int rem = e - r * FD_IW * 3;
            c = rem / 3;
            ci = rem - c * 3;
        }
        float v = pv[it];
        if (FMT == VGH_IMG_U8_NHWC) {
            const int q = __float_as_int(pv[it]);
            v = q >= 0 ? lut[q] : 0.0f;
        }
        img[ci][r][c] = v;
    }
    __syncthreads();
    // ---- stem: one pixel per lane, 48 channels in three groups of 16 (wave-uniform weights: scalar loads), exactly the arithmetic of stem_kernel ----
    // ---- stem on the FP32 matrix cores: D[cout][pixel] += W[k][cout] * x[pixel][k], v_mfma_f32_32x32x2_f32 = an exact fmaf chain in ascending k (two
    //      k per instruction, in order; verified bit for bit in the FLAME kernels), so every stem value is the same fp32 number as stem_kernel's VALU
    //      chain -- at the full 64 FLOP/clk/SIMD of the matrix pipe instead of the ~50 % of it the scalar-operand v_pk_fma stream reaches (r03 A/B:
    //      the VALU version of this kernel ran 703 us against 307 + 369 us for the two launches).  K = 27 (+1 zero) = 14 steps, couts 48 (+16 zero rows)
    //      = two 32-row groups, the block's 297 (+23 idle) pixels = ten 32-column groups, two per wave.
    {
        const int r32 = lane & 31, h = lane >> 5;
        float A[14][2];  // this lane's A operands for every (k pair, cout group): W[2kp + h][32 r + r32], zero beyond k = 26 / cout 47
#pragma unroll
        for (int kp = 0; kp < 14; ++kp)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int k = 2 * kp + h, co = 32 * r + r32;
                A[kp][r] = (k < 27 && co < FD_CO) ? wstem[k * FD_CO + co] : 0.0f;
            }
        f32x4_t bq[6];  // bias of this lane's output channels: group 0 rows 8q + 4h .. +3 (q < 4), group 1 rows 32 + 8q + 4h (q < 2)
#pragma unroll
        for (int q = 0; q < 6; ++q) bq[q] = *(const f32x4_t*)(bstem + 8 * q + 4 * h);
        const float* const imgf = &img[0][0][0];
        constexpr int IP = FD_IW + 1, IC = FD_IH * IP;  // row pitch, channel pitch of the image patch
#pragma unroll 1
        for (int gi = 0; gi < 2; ++gi) {
            const int p = (w * 2 + gi) * 32 + r32;
            const int sp = p < FD_NS ? p : FD_NS - 1;
            const int sy = sp / FD_SW, sx = sp - sy * FD_SW;
            const int gy = 2 * oy0 - 1 + sy, gx = 2 * ox0 - 1 + sx;  // position in the stem map
            // pixels outside the stem map are the downsample conv's zero padding: a multiply by 0 / 1 (a select would let hipcc sink work into a branch)
            const float keep = ((unsigned)gy < (unsigned)Hs && (unsigned)gx < (unsigned)Ws) ? 1.0f : 0.0f;
            const int base = (2 * sy) * IP + 2 * sx;
            f32x16_t acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;
#pragma unroll
            for (int kp = 0; kp < 14; ++kp) {
                // k = (ky*3 + kx)*3 + ci: even k for lanes 0..31, odd k for lanes 32..63 (k = 27 is the zero pad)
                constexpr auto off = [](int k) { return (k % 3) * IC + (k / 9) * IP + (k / 3) % 3; };
                const float xe = imgf[base + off(2 * kp)];
                const float xo = (2 * kp + 1 < 27) ? imgf[base + off(2 * kp + 1 < 27 ? 2 * kp + 1 : 0)] : 0.0f;
                const float bx = h ? xo : xe;
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[kp][0], bx, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[kp][1], bx, acc1, 0, 0, 0);
            }
            if (p < FD_NS) {
                char* const xp = X + p * FD_XP;
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const int c0 = 8 * q + 4 * h;  // first of this lane's 4 consecutive channels
                    bf16x4_t o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (__bf16)(fmaxf((q < 4 ? acc0[q * 4 + e] : acc1[(q - 4) * 4 + e]) + bq[q][e], 0.0f) * keep);
                    *(bf16x4_t*)(xp + c0 * 2) = o;
                }
            }
        }
    }
    __syncthreads();
    if (w >= 3) return;  // waves 3, 4 only computed stem pixels
    // ---- 3x3 stride-2 conv 48 -> 96 from the LDS stem tile: wave w owns couts [32w, 32w + 32) x both pixel groups ----
    const int r32 = lane & 31, half = lane >> 5;
    f32x16_t acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    // lane (pixel p of group j): output (ty, tx) = (2j + p / 16, p % 16) -> stem tile pixel (2ty + ky, 2tx + kx)
    int xb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ty = 2 * j + (r32 >> 4), tx = r32 & 15;
        xb[j] = ((2 * ty) * FD_SW + 2 * tx) * FD_XP + half * 16;
    }
    const uint16_t* const wrow = a.wds + ((size_t)(w * 32 + r32)) * 16 + half * 8;  // + (tap * 3 + chunk) * 96 * 16
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        bf16x8_t af[9];  // the nine (kx, chunk) A fragments of this kernel row: all loads in flight together
#pragma unroll
        for (int s = 0; s < 9; ++s) af[s] = *(const bf16x8_t*)(wrow + (size_t)(ky * 9 + s) * FD_CD * 16);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bf16x8_t bf = *(const bf16x8_t*)(X + xb[j] + (ky * FD_SW + kx) * FD_XP + c * 32);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kx * 3 + c], bf, acc[j], 0, 0, 0);
                }
            }
    }
    // ---- epilogue: bias + ReLU, fp32 -> wave-private LDS strip [32 px][36 floats] (the image patch is dead) -> 16-byte stores ----
    float* const stg = &img[0][0][0] + w * (32 * 36);
    const int half4 = half * 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cl = q * 8 + half4;
            const f32x4_t bv = *(const f32x4_t*)(a.bds + w * 32 + cl);
            f32x4_t v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[j][q * 4 + e] + bv[e], 0.0f);
            *(f32x4_t*)(stg + r32 * 36 + cl) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int it = lane + 64 * t;  // 32 px x 4 chunks of 8 channels
            const int px = it >> 2, ch = it & 3;
            const int oy = oy0 + 2 * j + (px >> 4), ox = ox0 + (px & 15);
            const f32x4_t v0 = *(const f32x4_t*)(stg + px * 36 + ch * 8), v1 = *(const f32x4_t*)(stg + px * 36 + ch * 8 + 4);
            bf16x8_t ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ov[e] = (__bf16)v0[e];
                ov[4 + e] = (__bf16)v1[e];
            }
            if (oy < Ho && ox < Wo) *(bf16x8_t*)(a.out + (((int64_t)b * Ho + oy) * Wo + ox) * a.out_pitch + a.out_coff + w * 32 + ch * 8) = ov;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

// dense [96][3][3][64] fp32 (channels >= 48 are the zero padding of the unfused tensor) -> bf16 [9 taps][3 chunks of 16 channels][96][16]
void vgh_pack_stem_ds_weights_host(const float* w, uint16_t* dst) {
    for (int tap = 0; tap < 9; ++tap)
        for (int c = 0; c < 3; ++c)
            for (int co = 0; co < FD_CD; ++co)
                for (int e = 0; e < 16; ++e) dst[(((size_t)tap * 3 + c) * FD_CD + co) * 16 + e] = vgh_f32_to_bf16_host(w[((size_t)co * 9 + tap) * 64 + c * 16 + e]);
}

int vgh_launch_stem_ds(const void* image, int image_fmt, int B, int H, int W, const float* wstem, const float* bstem, const uint16_t* wds, const float* bds, uint16_t* out,
                       int64_t out_pitch, int out_coff, hipStream_t stream) {
    VGH_REQUIRE(H % 4 == 0 && W % 4 == 0, "stem_ds: image size must be a multiple of 4");
    VGH_REQUIRE(out_pitch % 8 == 0 && out_coff % 8 == 0, "stem_ds: output alignment");
    VGH_REQUIRE(image_fmt == VGH_IMG_F32_NCHW || image_fmt == VGH_IMG_U8_NHWC, "stem_ds: unknown image format %d", image_fmt);
    if (B == 0) return VGH_OK;
    StemDsArgs a{image, wstem, bstem, wds, bds, out, out_pitch, out_coff, H, W};
    const dim3 grid((W / 4 + FD_TW - 1) / FD_TW, (H / 4 + FD_TH - 1) / FD_TH, B);
    if (image_fmt == VGH_IMG_U8_NHWC)
        hipLaunchKernelGGL(stem_ds_kernel<VGH_IMG_U8_NHWC>, grid, dim3(FD_THREADS), 0, stream, a, wstem, bstem);
    else
        hipLaunchKernelGGL(stem_ds_kernel<VGH_IMG_F32_NCHW>, grid, dim3(FD_THREADS), 0, stream, a, wstem, bstem);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}
