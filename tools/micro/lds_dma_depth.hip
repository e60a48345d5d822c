// Calibration micro-benchmark (r06): how many 1-KiB LDS-DMA wave-instructions can ONE wave keep in flight, and what does a CU ingest with W waves doing so?
// The implicit-GEMM K loop at small batch behaves as  T = a * steps + b * (LDS-DMA instructions per wave)  with b ~ 56 ns whatever the ring depth
// (profiles/r06_tune_l1.json: 64x64_k4 r2 / r3 / r4 = 30.5 / 26.8 / 27.6 us for K = 4608) -- a ring that does not hide latency means the loads themselves are
// rate-limited per wave.  Each wave issues DEPTH loads (distinct lines, marching through a buffer far larger than L2 + MALL), waits vmcnt(0), repeats ITER times.
//   MODE 0: buffer_load_dwordx4 ... lds (what the conv kernels use)      MODE 1: global_load_dwordx4 into VGPRs (64 lanes x 16 B, same bytes)
//   hipcc --offload-arch=gfx950 -O3 -o lds_dma_depth lds_dma_depth.hip && ./lds_dma_depth
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
#define AS3 __attribute__((address_space(3)))

__device__ __forceinline__ void bload_lds16(const void* base, unsigned voffset, unsigned soffset, char* lds_wave_base) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x80000000, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (AS3 void*)lds_wave_base, 16, voffset, soffset, 0, 0);
}

template <int DEPTH, int MODE>
__global__ __launch_bounds__(1024) void depth_kernel(const char* __restrict__ in, size_t bytes, int iters, unsigned long long* __restrict__ cyc, unsigned* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    // every wave of the launch walks its own region; consecutive loads of a wave are 1 KiB apart (whole lines, all channels of the HBM stacks over time)
    const size_t gw = (size_t)blockIdx.x * nw + w, nwaves = (size_t)gridDim.x * nw;
    const size_t region = (bytes / nwaves) & ~(size_t)1023;
    const char* base = in + gw * region;
    const size_t per_iter = (size_t)DEPTH * 1024;
    u32x4_t acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    size_t off = 0;
    for (int it = 0; it < iters; ++it) {
        if (off + per_iter > region) off = 0;
        if (MODE == 0) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) bload_lds16(base + off, (unsigned)(d * 1024 + lane * 16), 0, smem + (w * DEPTH + d) * 1024);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            u32x4_t v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = *(const u32x4_t*)(base + off + d * 1024 + lane * 16);
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc += v[d];
        }
        off += per_iter;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (MODE == 1 && acc.x == 0x12345678u) sink[0] = acc.y;
    if (lane == 0) cyc[gw] = t1 - t0;
}

template <int DEPTH, int MODE>
static void run(const char* d_in, size_t bytes, int waves, int blocks, unsigned long long* d_cyc, unsigned* d_sink) {
    const int iters = 400;
    const int lds = MODE == 0 ? waves * DEPTH * 1024 : 0;
    if (lds > 160 * 1024) return;
    hipFuncSetAttribute((const void*)depth_kernel<DEPTH, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((depth_kernel<DEPTH, MODE>), dim3(blocks), dim3(waves * 64), lds, 0, d_in, bytes, 50, d_cyc, d_sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((depth_kernel<DEPTH, MODE>), dim3(blocks), dim3(waves * 64), lds, 0, d_in, bytes, iters, d_cyc, d_sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const int n = blocks * waves;
    unsigned long long* h = (unsigned long long*)malloc(n * sizeof(*h));
    hipMemcpy(h, d_cyc, n * sizeof(*h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < n; ++i) s += (double)h[i];
    free(h);
    const double cyc_iter = s / n / iters;  // shader cycles (s_memtime) per round, averaged over the waves
    const double ns_iter = ms * 1e6 / iters;  // every wave runs the whole launch: wall time / rounds
    const double gb = (double)n * iters * DEPTH * 1024 / 1e9;
    printf("%-5s depth %2d  waves/CU %2d  blocks %4d : %8.0f ns = %6.0f cycles per round (%6.1f ns per 1-KiB instruction per wave)  %7.1f GB/s per CU  %6.2f TB/s chip\n", MODE ? "vgpr" : "lds", DEPTH, waves,
           blocks, ns_iter, cyc_iter, ns_iter / DEPTH, gb / (ms * 1e-3) / blocks, gb / (ms * 1e-3) / 1e3);
}

int main() {
    const size_t bytes = (size_t)8 << 30;
    char* d_in;
    if (hipMalloc(&d_in, bytes) != hipSuccess) return 1;
    hipMemset(d_in, 1, bytes);
    unsigned long long* d_cyc;
    hipMalloc(&d_cyc, 256 * 16 * 4 * sizeof(*d_cyc));
    unsigned* d_sink;
    hipMalloc(&d_sink, 16);
    printf("# one block per CU (256 blocks): what one CU pulls with W waves, each keeping DEPTH 1-KiB loads in flight (HBM-resident data, every load a fresh line)\n");
    for (int waves : {1, 4, 8, 16}) {
        run<1, 0>(d_in, bytes, waves, 256, d_cyc, d_sink);
        run<2, 0>(d_in, bytes, waves, 256, d_cyc, d_sink);
        run<4, 0>(d_in, bytes, waves, 256, d_cyc, d_sink);
        run<8, 0>(d_in, bytes, waves, 256, d_cyc, d_sink);
        run<16, 0>(d_in, bytes, waves, 256, d_cyc, d_sink);
        run<32, 0>(d_in, bytes, waves, 256, d_cyc, d_sink);
    }
    for (int waves : {1, 4, 8, 16}) {
        run<1, 1>(d_in, bytes, waves, 256, d_cyc, d_sink);
        run<4, 1>(d_in, bytes, waves, 256, d_cyc, d_sink);
        run<8, 1>(d_in, bytes, waves, 256, d_cyc, d_sink);
        run<16, 1>(d_in, bytes, waves, 256, d_cyc, d_sink);
        run<32, 1>(d_in, bytes, waves, 256, d_cyc, d_sink);
    }
    printf("# a few CUs only (32 blocks, 4 waves): the same with an idle memory system -- the unloaded round trip\n");
    run<1, 0>(d_in, bytes, 4, 32, d_cyc, d_sink);
    run<4, 0>(d_in, bytes, 4, 32, d_cyc, d_sink);
    run<8, 0>(d_in, bytes, 4, 32, d_cyc, d_sink);
    run<16, 0>(d_in, bytes, 4, 32, d_cyc, d_sink);
    run<32, 0>(d_in, bytes, 4, 32, d_cyc, d_sink);
    run<8, 1>(d_in, bytes, 4, 32, d_cyc, d_sink);
    run<32, 1>(d_in, bytes, 4, 32, d_cyc, d_sink);
    return 0;
}
