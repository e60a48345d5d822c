// fp32 "parity mode" convolution: plain LDS-tiled FMA kernel, fp32 NHWC activations, fp32 weights, fp32 accumulate.
//
// Purpose: run the *same* lowered op program with no bf16 anywhere, so that the engine can be compared with the
// reference PyTorch CPU path (fp32) at north_star's tolerances (bbox IoU >= 0.999, FLAME params within 1e-4).  It is
// not a throughput path (fp32 has no fast matrix format on gfx950: mfma f32 runs at the VALU rate, MI355X_MICROARCH.md),
// so it is a straightforward 64 pixel x 64 cout register-tiled kernel with the same fused epilogue semantics as
// conv_igemm.hip: bias, ReLU/SiLU, + alpha*residual after the activation, two-segment channel store, ConvTranspose
// pixel-shuffle store.
#include "vgh_internal.h"

namespace {

constexpr int TP = 64, TC = 64, TK = 32;

__global__ __launch_bounds__(256) void conv_f32_kernel(const ConvArgs a, const float* __restrict__ wdense) {
    __shared__ float Xs[TK][TP + 4];
    __shared__ float Ws[TK][TC + 4];
    const int tid = threadIdx.x;
    const int p0 = blockIdx.x * TP, c0 = blockIdx.y * TC;
    const int HoWo = a.Ho * a.Wo;
    const float* in = (const float*)a.in;
    // loader roles: thread -> (row = tid / 4, 8-float chunk = tid % 4)
    const int lrow = tid >> 2, lch = (tid & 3) * 8;
    const int m_l = p0 + lrow;
    const bool prow_ok = m_l < a.P;
    int lb = 0, loy = 0, lox = 0;
    if (prow_ok) {
        lb = m_l / HoWo;
        const int rem = m_l - lb * HoWo;
        loy = rem / a.Wo;
        lox = rem - loy * a.Wo;
    }
    const int taps = a.ksize * a.ksize;
    const int K = taps * a.cin;
    const int wrow = c0 + lrow;  // cout row this thread stages
    // compute roles: 16 x 16 threads, 4 pixels x 4 couts each
    const int ty = tid >> 4, tx = tid & 15;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;

    for (int tap = 0; tap < taps; ++tap) {
        const int ky = tap / a.ksize, kx = tap - ky * a.ksize;
        const int iy = loy * a.stride - a.pad + ky, ix = lox * a.stride - a.pad + kx;
        const bool ok = prow_ok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const float* xsrc = in + (((int64_t)lb * a.H + iy) * a.W + ix) * a.in_pitch + a.in_coff + lch;
        for (int cb = 0; cb < a.cin; cb += TK) {
            f32x4_t x0 = {0, 0, 0, 0}, x1 = {0, 0, 0, 0}, w0 = {0, 0, 0, 0}, w1 = {0, 0, 0, 0};
            if (ok) {
                x0 = *(const f32x4_t*)(xsrc + cb);
                x1 = *(const f32x4_t*)(xsrc + cb + 4);
            }
            if (wrow < a.cout_pad) {
                const float* wsrc = wdense + (int64_t)wrow * K + (int64_t)tap * a.cin + cb + lch;
                w0 = *(const f32x4_t*)(wsrc);
                w1 = *(const f32x4_t*)(wsrc + 4);
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                Xs[lch + e][lrow] = x0[e];
                Xs[lch + 4 + e][lrow] = x1[e];
                Ws[lch + e][lrow] = w0[e];
                Ws[lch + 4 + e][lrow] = w1[e];
            }
            __syncthreads();
#pragma unroll 8
            for (int k = 0; k < TK; ++k) {
                const f32x4_t xv = *(const f32x4_t*)(&Xs[k][ty * 4]);
                const f32x4_t wv = *(const f32x4_t*)(&Ws[k][tx * 4]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(xv[i], wv[j], acc[i][j]);
            }
        }
    }
    float* out = (float*)a.out;
    const float* res = (const float*)a.res;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = p0 + ty * 4 + i;
        if (m >= a.P) continue;
        int sb = 0, sy = 0, sx = 0;
        if (a.shuffle) {
            sb = m / HoWo;
            const int rem = m - sb * HoWo;
            sy = rem / a.Wo;
            sx = rem - sy * a.Wo;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + tx * 4 + j;
            if (c >= a.cout_store) continue;
            float v = acc[i][j] + a.bias[c];
            if (a.act == VGH_ACT_RELU) v = fmaxf(v, 0.0f);
            else if (a.act == VGH_ACT_SILU) v = v / (1.0f + expf(-v));
            int oc = c;
            int64_t opix = m;
            if (a.shuffle) {
                const int d = c / a.shuffle_c;
                oc = c - d * a.shuffle_c;
                opix = ((int64_t)sb * (2 * a.Ho) + 2 * sy + (d >> 1)) * (2 * a.Wo) + 2 * sx + (d & 1);
            }
            if (res) v += a.alpha * res[(int64_t)m * a.res_pitch + a.res_coff + oc];
            const int ochan = (oc >= a.out_split) ? a.out_coff2 + (oc - a.out_split) : a.out_coff + oc;
            out[opix * a.out_pitch + ochan] = v;
        }
    }
}

template <int FMT>
__global__ __launch_bounds__(256) void stem_f32_kernel(const void* __restrict__ image, int H, int W, const float* __restrict__ wgt /*[27][48]*/,
                                                       const float* __restrict__ bias, float* __restrict__ out, int64_t out_pitch, int out_coff, int B) {
    const int Ho = H / 2, Wo = W / 2;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)B * Ho * Wo) return;
    const int b = (int)(gid / (Ho * Wo));
    const int rem = (int)(gid - (int64_t)b * Ho * Wo);
    const int oy = rem / Wo, ox = rem - oy * Wo;
    float x[27];
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx)
            for (int ci = 0; ci < 3; ++ci) {
                const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
                float v = 0.0f;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                    if (FMT == VGH_IMG_F32_NCHW)
                        v = ((const float*)image)[(((int64_t)b * 3 + ci) * H + iy) * W + ix];
                    else
                        v = (float)((const uint8_t*)image)[(((int64_t)b * H + iy) * W + ix) * 3 + ci] / 255.0f;
                }
                x[(ky * 3 + kx) * 3 + ci] = v;
            }
    float* op = out + gid * out_pitch + out_coff;
    for (int c = 0; c < 48; ++c) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 27; ++k) acc = fmaf(x[k], wgt[k * 48 + c], acc);
        op[c] = fmaxf(acc + bias[c], 0.0f);
    }
    for (int c = 48; c < 64; ++c) op[c] = 0.0f;
}

__global__ __launch_bounds__(256) void spp_pool_f32_kernel(float* __restrict__ buf, int64_t pitch, int coff, int C, int B, int H, int W) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)B * H * W * C) return;
    const int c = (int)(gid % C);
    const int64_t pix = gid / C;
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const int64_t b = pix / ((int64_t)W * H);
    const float* src = buf + b * H * W * pitch + coff + c;
    float m5 = -INFINITY, m9 = -INFINITY, m13 = -INFINITY;
    for (int dy = -6; dy <= 6; ++dy) {
        const int yy = y + dy;
        if ((unsigned)yy >= (unsigned)H) continue;
        for (int dx = -6; dx <= 6; ++dx) {
            const int xx = x + dx;
            if ((unsigned)xx >= (unsigned)W) continue;
            const float v = src[((int64_t)yy * W + xx) * pitch];
            m13 = fmaxf(m13, v);
            if (dy >= -4 && dy <= 4 && dx >= -4 && dx <= 4) m9 = fmaxf(m9, v);
            if (dy >= -2 && dy <= 2 && dx >= -2 && dx <= 2) m5 = fmaxf(m5, v);
        }
    }
    float* dst = buf + pix * pitch + coff + c;
    dst[C] = m5;
    dst[2 * C] = m9;
    dst[3 * C] = m13;
}

}  // namespace

int vgh_launch_conv_f32(const ConvArgs& a, const float* wdense, hipStream_t stream) {
    VGH_REQUIRE(a.cin % 32 == 0 && a.in_coff % 4 == 0 && a.in_pitch % 4 == 0, "conv_f32: cin %% 32 / 16-byte alignment of the input view");
    if (a.P == 0) return VGH_OK;
    dim3 grid((a.P + TP - 1) / TP, (a.cout_pad + TC - 1) / TC);
    hipLaunchKernelGGL(conv_f32_kernel, grid, dim3(256), 0, stream, a, wdense);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}

int vgh_launch_stem_f32(const void* image, int image_fmt, int B, int H, int W, const float* w, const float* bias, float* out, int64_t out_pitch, int out_coff,
                        hipStream_t stream) {
    if (B == 0) return VGH_OK;
    const int64_t n = (int64_t)B * (H / 2) * (W / 2);
    dim3 grid((unsigned)((n + 255) / 256));
    if (image_fmt == VGH_IMG_F32_NCHW)
        hipLaunchKernelGGL(stem_f32_kernel<VGH_IMG_F32_NCHW>, grid, dim3(256), 0, stream, image, H, W, w, bias, out, out_pitch, out_coff, B);
    else if (image_fmt == VGH_IMG_U8_NHWC)
        hipLaunchKernelGGL(stem_f32_kernel<VGH_IMG_U8_NHWC>, grid, dim3(256), 0, stream, image, H, W, w, bias, out, out_pitch, out_coff, B);
    else
        VGH_REQUIRE(false, "stem_f32: unknown image format %d", image_fmt);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}

int vgh_launch_spp_pool_f32(float* buf, int64_t pitch, int coff, int C, int B, int H, int W, hipStream_t stream) {
    if (B == 0) return VGH_OK;
    const int64_t n = (int64_t)B * H * W * C;
    hipLaunchKernelGGL(spp_pool_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, buf, pitch, coff, C, B, H, W);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}
