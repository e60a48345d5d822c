// Calibration micro-benchmark: fp32 VALU roof -- v_fma_f32 vs v_pk_fma_f32 vs v_mfma_f32_32x32x2_f32, register-only loops.
// hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((__vector_size__(2 * sizeof(float)))) float f32x2_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;

constexpr int NACC = 32;

__global__ __launch_bounds__(256) void fma_loop(float* out, int iters, float seed) {
    float acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = seed * i;
    float a = seed * (threadIdx.x % 7 + 1), b = seed * (threadIdx.x % 5 + 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void pk_fma_loop(float* out, int iters, float seed) {
    f32x2_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x2_t{seed * i, seed};
    f32x2_t a = {seed * (threadIdx.x % 7 + 1), seed}, b = {seed * (threadIdx.x % 5 + 1), seed * 3};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// op_sel broadcast of the low half of A (what a packed two-heads-per-register FLAME loop would issue)
__global__ __launch_bounds__(256) void pk_fma_bcast_loop(float* out, int iters, float seed) {
    f32x2_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x2_t{seed * i, seed};
    f32x2_t a = {seed * (threadIdx.x % 7 + 1), seed}, b = {seed * (threadIdx.x % 5 + 1), seed * 3};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void mfma_f32_loop(float* out, int iters, float seed) {
    f32x16_t acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = seed * (threadIdx.x % 7 + 1), b = seed * (threadIdx.x % 5 + 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static void run(const char* name, K kernel, int blocks, int iters, double flops_per_thread_iter, float* out, float seed) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, iters, seed);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) printf("%-22s %.3f ms  %.1f TFLOP/s\n", name, ms, flops_per_thread_iter * iters * (double)blocks * 256 / ms / 1e9);
    }
}

int main(int argc, char** argv) {
    int blocks_per_cu = argc > 1 ? atoi(argv[1]) : 2;
    int iters = argc > 2 ? atoi(argv[2]) : 20000;
    float seed = argc > 3 ? atof(argv[3]) : 0.01f;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int blocks = p.multiProcessorCount * blocks_per_cu;
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    printf("CUs %d, %d blocks/CU x 256 threads, %d iters\n", p.multiProcessorCount, blocks_per_cu, iters);
    run("v_fma_f32", fma_loop, blocks, iters, 2.0 * NACC, out, seed);
    run("v_pk_fma_f32", pk_fma_loop, blocks, iters, 4.0 * NACC, out, seed);
    run("v_pk_fma_f32 op_sel", pk_fma_bcast_loop, blocks, iters, 4.0 * NACC, out, seed);
    run("v_mfma_f32_32x32x2_f32", mfma_f32_loop, blocks, iters, 2.0 * 32 * 32 * 2 * 4 / 64.0, out, seed);
    return 0;
}
