// Result-side consumers of the decoded meshes, on the device:
//   * z-buffer rasteriser = Sim3DR `_rasterize` (head_detector/Sim3DR/lib/rasterize_kernel.cpp:219-293, barycentric weights
//     `get_point_weight` :53-79) behind `Sim3DR.rasterize` (Sim3DR.py:17-38) -- the only native code of the reference
//   * PNCC composition     = `PNCCProcessor.__call__` (head_detector/pncc_processor.py:66-73)
//   * refined head bbox    = `refined_head_bbox` (head_detector/utils.py:26-35)
//
// The reference rasteriser is a serial loop over triangles with a strict `>` depth test, i.e. per pixel the winner is the
// covering triangle of greatest interpolated depth, ties going to the EARLIEST triangle.  That is order-free once stated as
// a maximum over the 64-bit key (order-preserving depth bits << 32 | ~triangle index), so the device version is
//   pass A  one lane per triangle: walk its clipped bounding box, atomicMax the key of every covered pixel
//   pass B  one lane per pixel   : re-evaluate the winner's barycentric weights (same float ops, no contraction) -> colour,
//                                  and reset the key for the next mesh.
// All arithmetic is IEEE float32 in the reference's operation order (#pragma clang fp contract(off), true division), so
// images are bit-identical to the reference's C++ (tests: oracle/_ref/libsim3dr_ref.so built from the reference sources).
#include <hip/hip_runtime.h>

#include "vgh_internal.h"

#pragma clang fp contract(off)

namespace {

constexpr unsigned long long kLowEmpty = 0xFFFFFFFFull;

__device__ __forceinline__ unsigned ord_bits(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline unsigned long long empty_key() {
    // depth buffer initial value -1e8 (Sim3DR.py:30); a triangle must be STRICTLY deeper to paint
    const float init = -1e8f;
    unsigned u;
    memcpy(&u, &init, 4);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | kLowEmpty;
}

struct Tri2 {
    float p0x, p0y, p1x, p1y, p2x, p2y;
    float v0x, v0y, v1x, v1y, dot00, dot01, dot11, inv;
};

__device__ __forceinline__ void tri_setup(Tri2& t) {
    // get_point_weight, pixel-independent part (rasterize_kernel.cpp:55-72)
    t.v0x = t.p2x - t.p0x;
    t.v0y = t.p2y - t.p0y;
    t.v1x = t.p1x - t.p0x;
    t.v1y = t.p1y - t.p0y;
    t.dot00 = t.v0x * t.v0x + t.v0y * t.v0y;
    t.dot01 = t.v0x * t.v1x + t.v0y * t.v1y;
    t.dot11 = t.v1x * t.v1x + t.v1y * t.v1y;
    const float den = t.dot00 * t.dot11 - t.dot01 * t.dot01;
    t.inv = (den == 0.0f) ? 0.0f : 1.0f / den;
}
__device__ __forceinline__ void tri_weights(const Tri2& t, float px, float py, float& w0, float& w1, float& w2) {
    const float v2x = px - t.p0x, v2y = py - t.p0y;
    const float dot02 = t.v0x * v2x + t.v0y * v2y;
    const float dot12 = t.v1x * v2x + t.v1y * v2y;
    const float u = (t.dot11 * dot02 - t.dot01 * dot12) * t.inv;
    const float v = (t.dot00 * dot12 - t.dot01 * dot02) * t.inv;
    w0 = 1.0f - u - v;
    w1 = v;
    w2 = u;
}

__global__ __launch_bounds__(256) void raster_clear_kernel(unsigned long long* __restrict__ zbuf, int n, unsigned long long key) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) zbuf[i] = key;
}

__global__ __launch_bounds__(64) void raster_depth_kernel(const float* __restrict__ ver, const int32_t* __restrict__ tri, int ntri, int h, int w, float zsign,
                                                          unsigned long long* __restrict__ zbuf) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= ntri) return;
    const int i0 = tri[3 * i], i1 = tri[3 * i + 1], i2 = tri[3 * i + 2];
    Tri2 t;
    t.p0x = ver[3 * i0];
    t.p0y = ver[3 * i0 + 1];
    t.p1x = ver[3 * i1];
    t.p1y = ver[3 * i1 + 1];
    t.p2x = ver[3 * i2];
    t.p2y = ver[3 * i2 + 1];
    const float d0 = zsign * ver[3 * i0 + 2], d1 = zsign * ver[3 * i1 + 2], d2 = zsign * ver[3 * i2 + 2];
    const float fx0 = fminf(t.p0x, fminf(t.p1x, t.p2x)), fx1 = fmaxf(t.p0x, fmaxf(t.p1x, t.p2x));
    const float fy0 = fminf(t.p0y, fminf(t.p1y, t.p2y)), fy1 = fmaxf(t.p0y, fmaxf(t.p1y, t.p2y));
    if (!(fx0 == fx0 && fx1 == fx1 && fy0 == fy0 && fy1 == fy1) || isinf(fx0) || isinf(fx1) || isinf(fy0) || isinf(fy1)) return;
    // clamp in float first: (int)ceil(1e30f) is undefined in C; the clamped result is what any in-range input gives
    const int x_min = max((int)ceilf(fmaxf(fx0, -1.0f)), 0), x_max = min((int)floorf(fminf(fx1, (float)w)), w - 1);
    const int y_min = max((int)ceilf(fmaxf(fy0, -1.0f)), 0), y_max = min((int)floorf(fminf(fy1, (float)h)), h - 1);
    if (x_max < x_min || y_max < y_min) return;
    tri_setup(t);
    const unsigned long long low = 0xFFFFFFFEull - (unsigned)i;
    for (int y = y_min; y <= y_max; ++y) {
        for (int x = x_min; x <= x_max; ++x) {
            float w0, w1, w2;
            tri_weights(t, (float)x, (float)y, w0, w1, w2);
            if (w2 > 0 && w1 > 0 && w0 > 0) {
                float pd = w0 * d0 + w1 * d1 + w2 * d2;
                if (pd != pd) continue;  // NaN never passes `p_depth > depth_buffer`
                pd = pd + 0.0f;          // -0 -> +0: the two zeros compare equal in the reference's test
                atomicMax(&zbuf[(size_t)y * w + x], ((unsigned long long)ord_bits(pd) << 32) | low);
            }
        }
    }
}

// mode 0: Sim3DR.rasterize (alpha = 1): covered pixels take the interpolated colour.
// mode 1: one head of PNCCProcessor.__call__: covered pixels whose painted colour is not all-zero take it
//         (`pncc_image[current.sum(2) != 0] = current[...]`, pncc_processor.py:72).
__global__ __launch_bounds__(256) void raster_resolve_kernel(const float* __restrict__ ver, const int32_t* __restrict__ tri, const float* __restrict__ col, int C, int h,
                                                            int w, int reverse, int mode, unsigned long long* __restrict__ zbuf, unsigned long long empty,
                                                            uint8_t* __restrict__ image) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= h * w) return;
    const unsigned long long key = zbuf[p];
    if ((key & 0xFFFFFFFFull) == kLowEmpty) return;
    zbuf[p] = empty;
    const int i = (int)(0xFFFFFFFEull - (key & 0xFFFFFFFFull));
    const int y = p / w, x = p - y * w;
    const int i0 = tri[3 * i], i1 = tri[3 * i + 1], i2 = tri[3 * i + 2];
    Tri2 t;
    t.p0x = ver[3 * i0];
    t.p0y = ver[3 * i0 + 1];
    t.p1x = ver[3 * i1];
    t.p1y = ver[3 * i1 + 1];
    t.p2x = ver[3 * i2];
    t.p2y = ver[3 * i2 + 1];
    tri_setup(t);
    float w0, w1, w2;
    tri_weights(t, (float)x, (float)y, w0, w1, w2);
    uint8_t* px = image + ((size_t)(reverse ? (h - 1 - y) : y) * w + x) * C;
    uint8_t q[8];
    unsigned sum = 0;
    for (int k = 0; k < C && k < 8; ++k) {
        const float pc = w0 * col[C * i0 + k] + w1 * col[C * i1 + k] + w2 * col[C * i2 + k];
        // (unsigned char)((1 - alpha) * image + alpha * 255 * p_color), alpha = 1 (rasterize_kernel.cpp:277-283)
        const float val = 0.0f * (float)px[k] + 255.0f * pc;
        q[k] = (uint8_t)((int)val & 0xFF);  // C's float -> unsigned char: truncation, low 8 bits for in-range values
        sum += q[k];
    }
    if (mode == 1 && sum == 0) return;
    for (int k = 0; k < C && k < 8; ++k) px[k] = q[k];
}

// refined_head_bbox (utils.py:26-35): int() of min/max x,y over a vertex subset -> (x, y, w, h)
__global__ __launch_bounds__(256) void head_bbox_kernel(const float* __restrict__ verts, int V, const int32_t* __restrict__ idx, int n_idx, int32_t* __restrict__ out) {
    __shared__ float s[4][256];
    const float* v = verts + (size_t)blockIdx.x * V * 3;
    float x0 = INFINITY, y0 = INFINITY, x1 = -INFINITY, y1 = -INFINITY;
    for (int e = threadIdx.x; e < n_idx; e += 256) {
        const int k = idx[e];
        const float x = v[3 * k], y = v[3 * k + 1];
        x0 = fminf(x0, x);
        y0 = fminf(y0, y);
        x1 = fmaxf(x1, x);
        y1 = fmaxf(y1, y);
    }
    s[0][threadIdx.x] = x0;
    s[1][threadIdx.x] = y0;
    s[2][threadIdx.x] = x1;
    s[3][threadIdx.x] = y1;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d) {
            s[0][threadIdx.x] = fminf(s[0][threadIdx.x], s[0][threadIdx.x + d]);
            s[1][threadIdx.x] = fminf(s[1][threadIdx.x], s[1][threadIdx.x + d]);
            s[2][threadIdx.x] = fmaxf(s[2][threadIdx.x], s[2][threadIdx.x + d]);
            s[3][threadIdx.x] = fmaxf(s[3][threadIdx.x], s[3][threadIdx.x + d]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int x = (int)s[0][0], y = (int)s[1][0], xx = (int)s[2][0], yy = (int)s[3][0];
        int32_t* o = out + (size_t)blockIdx.x * 4;
        o[0] = x;
        o[1] = y;
        o[2] = xx - x;
        o[3] = yy - y;
    }
}

int raster_one(const float* ver, const int32_t* tri, int ntri, const float* col, int C, uint8_t* image, int h, int w, int reverse, int mode, float zsign,
               unsigned long long* zbuf, hipStream_t st) {
    if (ntri > 0) {
        hipLaunchKernelGGL(raster_depth_kernel, dim3((ntri + 63) / 64), dim3(64), 0, st, ver, tri, ntri, h, w, zsign, zbuf);
        VGH_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(raster_resolve_kernel, dim3((h * w + 255) / 256), dim3(256), 0, st, ver, tri, col, C, h, w, reverse, mode, zbuf, empty_key(), image);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}

}  // namespace

extern "C" {

int vgh_rasterize(const float* verts_dev, const int32_t* tri_dev, int ntri, const float* colors_dev, int channels, uint8_t* image_dev, int H, int W, int reverse,
                  uint64_t* zbuf_dev, void* stream) {
    VGH_REQUIRE(image_dev && zbuf_dev && ((verts_dev && tri_dev && colors_dev) || ntri == 0), "rasterize: null argument");
    VGH_REQUIRE(H > 0 && W > 0 && (int64_t)H * W < (1ll << 31) && channels >= 1 && channels <= 8 && ntri >= 0, "rasterize: bad geometry (H %d W %d C %d)", H, W, channels);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(raster_clear_kernel, dim3((H * W + 255) / 256), dim3(256), 0, st, (unsigned long long*)zbuf_dev, H * W, empty_key());
    VGH_HIP(hipGetLastError());
    return raster_one(verts_dev, tri_dev, ntri, colors_dev, channels, image_dev, H, W, reverse, 0, 1.0f, (unsigned long long*)zbuf_dev, st);
}

int vgh_pncc_render(const float* verts_dev, int n_heads, int V, const int32_t* tri_dev, int ntri, const float* colors_dev, uint8_t* image_dev, int H, int W,
                    uint64_t* zbuf_dev, void* stream) {
    VGH_REQUIRE(image_dev && zbuf_dev && ((tri_dev && colors_dev && verts_dev) || n_heads == 0 || ntri == 0), "pncc_render: null argument");
    VGH_REQUIRE(H > 0 && W > 0 && (int64_t)H * W < (1ll << 31) && V > 0 && ntri >= 0 && n_heads >= 0, "pncc_render: bad geometry");
    hipStream_t st = (hipStream_t)stream;
    VGH_HIP(hipMemsetAsync(image_dev, 0, (size_t)H * W * 3, st));  // pncc_image = np.zeros_like(image), pncc_processor.py:67
    hipLaunchKernelGGL(raster_clear_kernel, dim3((H * W + 255) / 256), dim3(256), 0, st, (unsigned long long*)zbuf_dev, H * W, empty_key());
    VGH_HIP(hipGetLastError());
    for (int i = 0; i < n_heads; ++i) {  // heads paint in order; each gets a fresh depth buffer (Sim3DR.py:30)
        int rc = raster_one(verts_dev + (size_t)i * V * 3, tri_dev, ntri, colors_dev, 3, image_dev, H, W, 0, 1, -1.0f /* vertices[:, 2] *= -1 */,
                            (unsigned long long*)zbuf_dev, st);
        if (rc) return rc;
    }
    return VGH_OK;
}

int vgh_refined_head_bbox(const float* verts_dev, int n_heads, int V, const int32_t* idx_dev, int n_idx, int32_t* out_dev, void* stream) {
    VGH_REQUIRE(out_dev && idx_dev && (verts_dev || n_heads == 0), "refined_head_bbox: null argument");
    VGH_REQUIRE(n_idx > 0 && V > 0 && n_heads >= 0, "refined_head_bbox: bad sizes");
    if (n_heads == 0) return VGH_OK;
    hipLaunchKernelGGL(head_bbox_kernel, dim3(n_heads), dim3(256), 0, (hipStream_t)stream, verts_dev, V, idx_dev, n_idx, out_dev);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}

}  // extern "C"
