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

namespace {

constexpr int ST = 16;            // 16x16 output pixels per block
constexpr int SIN = 2 * ST + 1;   // 33x33 input patch
constexpr int STEM_CO = 48, STEM_CP = 64;

template <int FMT>
__global__ __launch_bounds__(256) void stem_kernel(const void* __restrict__ image, int H, int W, const float* __restrict__ wgt /*[27][48]*/,
                                                   const float* __restrict__ bias /*[48]*/, uint16_t* __restrict__ out, int64_t out_pitch,
                                                   int out_coff) {
    __shared__ float patch[3][SIN][SIN + 1];
    const int Ho = H / 2, Wo = W / 2;
    const int b = blockIdx.z, ty = blockIdx.y * ST, tx = blockIdx.x * ST;
    const int tid = threadIdx.x;
    const int iy_base = ty * 2 - 1, ix_base = tx * 2 - 1;
    for (int e = tid; e < 3 * SIN * SIN; e += 256) {
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
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
            if (FMT == VGH_IMG_F32_NCHW)
                v = ((const float*)image)[(((int64_t)b * 3 + ci) * H + iy) * W + ix];
            else
                v = (float)((const uint8_t*)image)[(((int64_t)b * H + iy) * W + ix) * 3 + ci] / 255.0f;  // detector.py:51
        }
        patch[ci][r][c] = v;
    }
    __syncthreads();
    const int ly = tid / ST, lx = tid % ST;
    const int oy = ty + ly, ox = tx + lx;
    if (oy >= Ho || ox >= Wo) return;
    float x[27];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) x[(ky * 3 + kx) * 3 + ci] = patch[ci][2 * ly + ky][2 * lx + kx];
    uint16_t* op = out + (((int64_t)b * Ho + oy) * Wo + ox) * out_pitch + out_coff;
#pragma unroll
    for (int cg = 0; cg < STEM_CO / 16; ++cg) {
        float acc[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = 0.0f;
#pragma unroll
        for (int k = 0; k < 27; ++k) {
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[c] = fmaf(x[k], wgt[k * STEM_CO + cg * 16 + c], acc[c]);
        }
        bf16x8_t o0, o1;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            o0[c] = (__bf16)fmaxf(acc[c] + bias[cg * 16 + c], 0.0f);
            o1[c] = (__bf16)fmaxf(acc[8 + c] + bias[cg * 16 + 8 + c], 0.0f);
        }
        *(bf16x8_t*)(op + cg * 16) = o0;
        *(bf16x8_t*)(op + cg * 16 + 8) = o1;
    }
    bf16x8_t z;
#pragma unroll
    for (int c = 0; c < 8; ++c) z[c] = (__bf16)0.0f;
    *(bf16x8_t*)(op + 48) = z;  // channels 48..63: exact zeros (K padding of the next conv)
    *(bf16x8_t*)(op + 56) = z;
}

// ---- SPP ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bf16x8_t max8(bf16x8_t a, bf16x8_t b) {
    bf16x8_t r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = ((float)a[e] >= (float)b[e]) ? a[e] : b[e];
    return r;
}

// one block = one image x CG channels; LDS ping-pong [H*W][CG] bf16
__global__ __launch_bounds__(256) void spp_pool_kernel(uint16_t* __restrict__ buf, int64_t pitch, int coff, int C, int H, int W, int CG) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int HW = H * W;
    const int cv = CG / 8;  // 16-byte vectors per pixel in this group
    bf16x8_t* A = (bf16x8_t*)smem;
    bf16x8_t* Bf = A + (size_t)HW * cv;
    const int b = blockIdx.y, cg0 = blockIdx.x * CG;
    uint16_t* base = buf + (int64_t)b * HW * pitch + coff + cg0;
    const int n = HW * cv;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int p = e / cv, v = e - p * cv;
        A[e] = *(const bf16x8_t*)(base + (int64_t)p * pitch + v * 8);
    }
    __syncthreads();
    bf16x8_t* src = A;
    bf16x8_t* dst = Bf;
    for (int pass = 0; pass < 3; ++pass) {
        for (int e = threadIdx.x; e < n; e += blockDim.x) {
            const int p = e / cv, v = e - p * cv;
            const int y = p / W, x = p - y * W;
            bf16x8_t m = src[e];
            for (int dy = -2; dy <= 2; ++dy) {
                const int yy = y + dy;
                if ((unsigned)yy >= (unsigned)H) continue;
                for (int dx = -2; dx <= 2; ++dx) {
                    const int xx = x + dx;
                    if ((unsigned)xx >= (unsigned)W) continue;
                    m = max8(m, src[(yy * W + xx) * cv + v]);
                }
            }
            dst[e] = m;
            *(bf16x8_t*)(base + (int64_t)p * pitch + (int64_t)(pass + 1) * C + v * 8) = m;
        }
        __syncthreads();
        bf16x8_t* t = src;
        src = dst;
        dst = t;
    }
}

}  // namespace

int vgh_launch_stem(const void* image, int image_fmt, int B, int H, int W, const float* w, const float* bias, uint16_t* out, int64_t out_pitch,
                    int out_coff, hipStream_t stream) {
    VGH_REQUIRE(H % 2 == 0 && W % 2 == 0, "stem: image size must be even");
    VGH_REQUIRE(out_pitch % 8 == 0 && out_coff % 8 == 0, "stem: output alignment");
    if (B == 0) return VGH_OK;
    dim3 grid((W / 2 + ST - 1) / ST, (H / 2 + ST - 1) / ST, B);
    if (image_fmt == VGH_IMG_F32_NCHW)
        hipLaunchKernelGGL(stem_kernel<VGH_IMG_F32_NCHW>, grid, dim3(256), 0, stream, image, H, W, w, bias, out, out_pitch, out_coff);
    else if (image_fmt == VGH_IMG_U8_NHWC)
        hipLaunchKernelGGL(stem_kernel<VGH_IMG_U8_NHWC>, grid, dim3(256), 0, stream, image, H, W, w, bias, out, out_pitch, out_coff);
    else
        VGH_REQUIRE(false, "stem: unknown image format %d", image_fmt);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}

int vgh_launch_spp_pool(uint16_t* buf, int64_t pitch, int coff, int C, int B, int H, int W, hipStream_t stream) {
    VGH_REQUIRE(C % 8 == 0 && pitch % 8 == 0 && coff % 8 == 0, "spp: channel alignment");
    if (B == 0) return VGH_OK;
    int CG = 32;
    while (CG > 8 && ((size_t)2 * H * W * CG * 2 > 64 * 1024 || C % CG != 0)) CG /= 2;
    VGH_REQUIRE(C % CG == 0 && (size_t)2 * H * W * CG * 2 <= 160 * 1024, "spp: feature map %dx%d too large for the LDS tile", H, W);
    const size_t lds = (size_t)2 * H * W * CG * 2;
    if (lds > 64 * 1024) VGH_HIP(hipFuncSetAttribute((const void*)spp_pool_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(spp_pool_kernel, dim3(C / CG, B), dim3(256), lds, stream, buf, pitch, coff, C, H, W, CG);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}
