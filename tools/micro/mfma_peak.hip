// Calibration micro-benchmark: what a register-only v_mfma_f32_32x32x16_bf16 loop sustains on this MI355X, and the
// shader clock it runs at (s_memtime cycles vs the constant-rate wall clock).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, long long* clk, int iters, float seed) {
    f32x16_t acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8_t a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (__bf16)(seed * (threadIdx.x % 7 + e));
        b[e] = (__bf16)(seed * (threadIdx.x % 5 + e));
    }
    long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        clk[0] = c1 - c0;
        clk[1] = w1 - w0;
    }
}

int main(int argc, char** argv) {
    int blocks_per_cu = argc > 1 ? atoi(argv[1]) : 2;
    int iters = argc > 2 ? atoi(argv[2]) : 20000;
    float seedv = argc > 3 ? atof(argv[3]) : 0.01f;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    int blocks = cus * blocks_per_cu;
    float* out;
    long long* clk;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipMalloc(&clk, 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(256), 0, 0, out, clk, iters, seedv);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        long long h[2];
        hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        double flops = 2.0 * 32 * 32 * 16 * 4.0 * iters * (double)blocks * 4;
        printf("CUs %d blocks/CU %d (waves/SIMD %d) iters %d seed %.3g: %.3f ms  %.1f TFLOP/s  shader-clk cycles %lld wall ticks %lld -> %.0f MHz (wall clock 100 MHz)  MFMA/cycle/SIMD %.3f\n",
               cus, blocks_per_cu, blocks_per_cu, iters, seedv, ms, flops / ms / 1e9, h[0], h[1], (double)h[0] / h[1] * 100.0,
               4.0 * iters * blocks_per_cu / (double)h[0] * 32.0 / 32.0);
    }
    return 0;
}
