// Calibration micro-benchmark: what the SHAPE of an LDS-DMA / store wave-instruction costs on the texture addresser when the bytes are the same.
// A 1x1 conv streams a contiguous [pixels][C] bf16 tensor; the kernels stage it "fragment-shaped" (one instruction = 16 pixels x one 64-B channel block: 16
// lines touched) because that is the MFMA operand layout.  The alternative is "line-shaped" (one instruction = 1 KiB contiguous: 8 whole lines) with the
// operand layout recovered by the LDS addressing.  Same for the output stores (64 B per pixel per instruction vs whole 128-B lines).
//   hipcc --offload-arch=gfx950 -O3 -o ta_shape ta_shape.hip && ./ta_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
#define AS3 __attribute__((address_space(3)))

__device__ __forceinline__ void bload_lds16(const void* base, unsigned voffset, char* lds_wave_base) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x80000000, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (AS3 void*)lds_wave_base, 16, voffset, 0, 0, 0);
}

// One persistent block per slot, tiles of BP = 128 pixels x PITCH bytes.  MODE bit0: loads line-shaped, bit1: stores line-shaped; DO_LD / DO_ST select the phases.
// OUTMUL: the output pixel pitch is OUTMUL x PITCH (a conv writing its channels into a wider concat buffer: partial lines per pixel);
// WKB: KiB of L2-resident "weights" every tile additionally stages through LDS-DMA (what a conv tile loads besides its pixels)
template <int PITCH, int MODE, int DO_LD, int DO_ST, int OUTMUL = 1, int WKB = 0>
__global__ __launch_bounds__(256, 2) void shape_kernel(const char* __restrict__ in, char* __restrict__ out, int ntiles, const char* __restrict__ wts) {
    constexpr int BP = 128, TILE = BP * PITCH, NI = TILE / 1024;  // 1 KiB wave-instructions per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const char* src = in + (size_t)t * TILE;
        char* dst = out + (size_t)t * TILE * OUTMUL;
        if (WKB) {
            for (int i = w; i < WKB; i += 4) bload_lds16(wts, i * 1024 + lane * 16, smem + TILE + i * 1024);
        }
        if (DO_LD) {
            for (int i = w; i < NI; i += 4) {
                unsigned off;
                if (MODE & 1) {
                    off = i * 1024 + lane * 16;  // 1 KiB contiguous
                } else {
                    // instruction i = (16-pixel unit u, channel block cb): lane -> pixel (lane >> 2), 16-byte chunk (lane & 3) of the 64-B block
                    constexpr int CB = PITCH / 64;
                    const int u = i / CB, cb = i - u * CB;
                    off = (u * 16 + (lane >> 2)) * PITCH + cb * 64 + (lane & 3) * 16;
                }
                bload_lds16(src, off, smem + i * 1024);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (DO_ST) {
            const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, 0x80000000, 0x00020000);
            for (int i = w; i < NI; i += 4) {
                const u32x4_t v = *(const u32x4_t*)(smem + i * 1024 + lane * 16);
                unsigned off;
                if ((MODE & 2) && OUTMUL == 1) {
                    off = i * 1024 + lane * 16;
                } else if (MODE & 2) {
                    const int byte = i * 1024 + lane * 16, px = byte / PITCH;  // whole pixels' worth of consecutive lanes, pixels OUTMUL x PITCH apart
                    off = px * (PITCH * OUTMUL) + (byte - px * PITCH);
                } else {
                    constexpr int CB = PITCH / 64;
                    const int u = i / CB, cb = i - u * CB;
                    off = (u * 16 + (lane >> 2)) * (PITCH * OUTMUL) + cb * 64 + (lane & 3) * 16;
                }
                __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, off, 0, 0);
            }
            __syncthreads();
        }
    }
}

template <typename F>
float time_it(F f, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    f();
    f();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

template <int PITCH>
void run(const char* in, char* out, size_t bytes) {
    constexpr int TILE = 128 * PITCH;
    const int ntiles = (int)(bytes / TILE);
    const int lds = TILE;
    const int blocks = 256 * 2;
    auto go = [&](auto kern, const char* what, double moved) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        const float ms = time_it([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, in, out, ntiles, (const char*)nullptr); }, 10);
        printf("  pitch %3d B  %-34s %7.3f ms = %5.2f TB/s\n", PITCH, what, ms, moved / ms / 1e9);
    };
    const double b = (double)ntiles * TILE;
    go(shape_kernel<PITCH, 0, 1, 0>, "load  fragment-shaped (16 x 64 B)", b);
    go(shape_kernel<PITCH, 1, 1, 0>, "load  line-shaped (1 KiB)", b);
    go(shape_kernel<PITCH, 0, 0, 1>, "store fragment-shaped (16 x 64 B)", b);
    go(shape_kernel<PITCH, 2, 0, 1>, "store line-shaped (1 KiB)", b);
    go(shape_kernel<PITCH, 0, 1, 1>, "copy  fragment / fragment", 2 * b);
    go(shape_kernel<PITCH, 1, 1, 1>, "copy  line / fragment", 2 * b);
    go(shape_kernel<PITCH, 2, 1, 1>, "copy  fragment / line", 2 * b);
    go(shape_kernel<PITCH, 3, 1, 1>, "copy  line / line", 2 * b);
}

template <int PITCH>
void run_conv_like(const char* in, char* out, size_t bytes, const char* wts) {
    constexpr int TILE = 128 * PITCH;
    const int ntiles = (int)(bytes / 4 / TILE);  // the output region is 4 x the input region
    const int blocks = 256 * 2;
    const double b = (double)ntiles * TILE;
    auto go = [&](auto kern, int lds, const char* what, double moved) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        const float ms = time_it([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, in, out, ntiles, wts); }, 10);
        printf("  pitch %3d B  %-58s %7.3f ms = %5.2f TB/s\n", PITCH, what, ms, moved / ms / 1e9);
    };
    go(shape_kernel<PITCH, 0, 1, 1, 1, 0>, TILE, "copy fragment / fragment, same pitch", 2 * b);
    go(shape_kernel<PITCH, 0, 1, 1, 4, 0>, TILE, "copy, output pixels 4 x pitch apart (concat buffer)", 2 * b);
    go(shape_kernel<PITCH, 2, 1, 1, 4, 0>, TILE, "same, whole-pixel (line-shaped) stores", 2 * b);
    go(shape_kernel<PITCH, 0, 1, 1, 1, 18>, TILE + 18 * 1024, "copy + 18 KiB of L2-resident weights per tile", 2 * b);
    go(shape_kernel<PITCH, 0, 1, 1, 4, 18>, TILE + 18 * 1024, "copy + weights, output 4 x pitch apart", 2 * b);
}

int main() {
    const size_t bytes = (size_t)640 << 20;  // one 96-channel tensor at 160^2 x 64 images is 315 MB; 640 MB defeats the 256 MB MALL
    char *in, *out;
    hipMalloc(&in, bytes);
    hipMalloc(&out, bytes);
    hipMemset(in, 1, bytes);
    hipMemset(out, 0, bytes);
    printf("LDS-DMA loads / 16-byte stores of a contiguous [pixels][pitch] tensor, 512 persistent blocks x 4 waves, tile = 128 pixels:\n");
    run<128>(in, out, bytes);
    run<192>(in, out, bytes);
    run<256>(in, out, bytes);
    run<384>(in, out, bytes);
    char* wts;
    hipMalloc(&wts, 64 << 10);
    hipMemset(wts, 2, 64 << 10);
    printf("conv-like variants (the input region is a quarter of the buffer, the output spreads over all of it):\n");
    run_conv_like<128>(in, out, bytes, wts);
    run_conv_like<192>(in, out, bytes, wts);
    run_conv_like<256>(in, out, bytes, wts);
    return 0;
}
