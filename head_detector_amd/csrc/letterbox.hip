// GPU letterbox = HeadDetector._transform_image (head_detector/detector.py:40-52) without the host round trip:
//   cv2.resize(image, (new_w, new_h), INTER_LANCZOS4) -> cv2.copyMakeBorder(..., BORDER_CONSTANT, value=127) -> u8 NHWC canvas that the
//   stem kernel consumes directly (its /255 is fused there, detector.py:51).
// The arithmetic is OpenCV's 8-bit fixed-point path (imgproc/src/resize.cpp: HResizeLanczos4<uchar,int,short> then
// VResizeLanczos4<uchar,int,short, FixedPtCast<int,uchar,22>>): 8x8 taps, weights = saturate_cast<short>(w * 2048) supplied by
// the host (head_detector_amd/letterbox.py builds them exactly as resize.cpp does), int32 accumulation, (v + 2^21) >> 22,
// out-of-range taps replicate the edge pixel.  Integer arithmetic => the 2-D sum can be evaluated per output pixel in any order.
// PARITY UNPINNED against cv2 itself (absent from this image); bit-exact against oracle/letterbox_oracle.py.
#include <hip/hip_runtime.h>

#include "vgh_internal.h"

namespace {

struct LbArgs {
    const uint8_t* src;  // [src_h, src_w, 3+] u8, pixel stride src_cn, row stride src_pitch bytes
    int src_h, src_w, src_cn;
    int64_t src_pitch;
    const int32_t* xofs;   // [new_w]
    const int16_t* alpha;  // [new_w][8]
    const int32_t* yofs;   // [new_h]
    const int16_t* beta;   // [new_h][8]
    int new_w, new_h, pad_x, pad_y, S;
    uint8_t pad[3];
    uint8_t* dst;  // [S,S,3]
};

__global__ __launch_bounds__(256) void letterbox_kernel(LbArgs a) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= a.S || y >= a.S) return;
    uint8_t* o = a.dst + ((size_t)y * a.S + x) * 3;
    const int dx = x - a.pad_x, dy = y - a.pad_y;
    if ((unsigned)dx >= (unsigned)a.new_w || (unsigned)dy >= (unsigned)a.new_h) {
        o[0] = a.pad[0];
        o[1] = a.pad[1];
        o[2] = a.pad[2];
        return;
    }
    const int sx = a.xofs[dx] - 3, sy = a.yofs[dy] - 3;
    int al[8], cx[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        al[k] = a.alpha[dx * 8 + k];
        cx[k] = min(max(sx + k, 0), a.src_w - 1) * a.src_cn;
    }
    unsigned acc[3] = {0u, 0u, 0u};  // unsigned: C's int accumulation with defined wrap-around
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ry = min(max(sy + j, 0), a.src_h - 1);
        const uint8_t* row = a.src + (size_t)ry * a.src_pitch;
        unsigned h0 = 0u, h1 = 0u, h2 = 0u;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint8_t* p = row + cx[k];
            h0 += (unsigned)((int)p[0] * al[k]);
            h1 += (unsigned)((int)p[1] * al[k]);
            h2 += (unsigned)((int)p[2] * al[k]);
        }
        const int b = a.beta[dy * 8 + j];
        acc[0] += (unsigned)((int)h0 * b);
        acc[1] += (unsigned)((int)h1 * b);
        acc[2] += (unsigned)((int)h2 * b);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int v = ((int)acc[c] + (1 << 21)) >> 22;  // FixedPtCast<int, uchar, 22>
        o[c] = (uint8_t)min(max(v, 0), 255);
    }
}

}  // namespace

extern "C" int vgh_letterbox(const uint8_t* src_dev, int src_h, int src_w, int src_channels, int64_t src_pitch_bytes, const int32_t* xofs_dev,
                             const int16_t* alpha_dev, const int32_t* yofs_dev, const int16_t* beta_dev, int new_w, int new_h, int pad_x, int pad_y,
                             const uint8_t* pad_rgb, uint8_t* dst_dev, int S, void* stream) {
    VGH_REQUIRE(src_dev && xofs_dev && alpha_dev && yofs_dev && beta_dev && dst_dev && pad_rgb, "letterbox: null argument");
    VGH_REQUIRE(src_h > 0 && src_w > 0 && src_channels >= 3 && src_pitch_bytes >= (int64_t)src_w * src_channels, "letterbox: bad source geometry");
    VGH_REQUIRE(new_w > 0 && new_h > 0 && pad_x >= 0 && pad_y >= 0 && pad_x + new_w <= S && pad_y + new_h <= S, "letterbox: the resized image does not fit the %dx%d canvas", S, S);
    LbArgs a;
    a.src = src_dev;
    a.src_h = src_h;
    a.src_w = src_w;
    a.src_cn = src_channels;
    a.src_pitch = src_pitch_bytes;
    a.xofs = xofs_dev;
    a.alpha = alpha_dev;
    a.yofs = yofs_dev;
    a.beta = beta_dev;
    a.new_w = new_w;
    a.new_h = new_h;
    a.pad_x = pad_x;
    a.pad_y = pad_y;
    a.S = S;
    a.pad[0] = pad_rgb[0];
    a.pad[1] = pad_rgb[1];
    a.pad[2] = pad_rgb[2];
    a.dst = dst_dev;
    hipLaunchKernelGGL(letterbox_kernel, dim3((S + 31) / 32, (S + 7) / 8), dim3(256), 0, (hipStream_t)stream, a);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}
