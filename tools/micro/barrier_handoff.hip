// What does one s_barrier cost an 8-wave workgroup whose waves keep the matrix pipe busy?  The skeleton of conv_pp.hip's K loop in miniature (no loads, no LDS
// reads): one 512-thread workgroup per CU (150 KB of dynamic LDS keeps a second one out), two waves per SIMD, each "M run" = 16 x v_mfma_f32_32x32x16_bf16 on 8
// independent accumulators (512 matrix-pipe cycles).  Modes:
//   0  runs only, no barrier                     -> the pipe's own rate with two waves per SIMD feeding it (1024 cycles per iteration and SIMD)
//   1  every wave: run, s_barrier                -> 1024 + what a barrier costs when BOTH waves of a SIMD arrive with MFMAs in flight
//   2  s_barrier only                            -> the bare round trip of the barrier
//   3  ping-pong, two barriers per iteration     -> conv_pp.hip V = 1 ("g" tiles): group 0 = [run, bar, (L phase: nothing), bar], group 1 one barrier behind
//   4  ping-pong, barrier 4 MFMAs before the end of the run (PP_BAR_TAIL = 4)
//   5  ping-pong, one barrier per iteration      -> conv_pp.hip V = 2 ("h" tiles): group 0 = [run, (L), bar], group 1 = [(L), run, bar]
//   7 / 8  as 3 with the accumulator re-use pattern of the conv tiles (NR / 2 accumulators, two k-halves each): at distance NR / 2, or back to back
//   6  as 3 with s_sleep(1)-polled LDS flags instead of s_barrier (each group publishes a counter, the other waits for it)
// Prints shader cycles per iteration (s_memtime) and the wall-clock rate.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/barrier_handoff.hip -o tools/micro/barrier_handoff ; ./barrier_handoff [iterations]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __forceinline__ void bar() { asm volatile("s_barrier" ::: "memory"); }

template <int N>
__device__ __forceinline__ void run(f32x16_t (&acc)[8], const bf16x8_t (&a)[4], const bf16x8_t (&b)[4], int first) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int m = (first + k) & 15;
        acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(m >> 1) & 3], b[(m & 1) + 2 * (m >> 3)], acc[m & 7], 0, 0, 0);
    }
}
// a run over NA accumulators, each used TWICE (the two k-halves of a tap): DIST = NA -> conv_pp.hip's order (all accumulators with half 0, then all with half 1:
// an accumulator comes back after NA MFMAs); DIST = 1 -> both halves of an accumulator back to back
template <int NA, int DIST>
__device__ __forceinline__ void run2(f32x16_t (&acc)[8], const bf16x8_t (&a)[4], const bf16x8_t (&b)[4]) {
#pragma unroll
    for (int k = 0; k < 2 * NA; ++k) {
        const int m = DIST == 1 ? k >> 1 : k % NA, h = DIST == 1 ? k & 1 : k / NA;
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(m >> 1) & 3], b[(m & 1) + 2 * h], acc[m], 0, 0, 0);
    }
}

template <int MODE, int NR = 16>  // NR: MFMAs per run (16 = a 128-cout wave tile, 12 = 96, 8 = 64)
__global__ __launch_bounds__(512, 2) void handoff_kernel(float* out, unsigned long long* cycles, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), grp = w >> 2;
    volatile int* flags = (volatile int*)smem;  // [2] progress counters of the two groups (mode 6)
    if (tid < 2) flags[tid] = 0;
    __syncthreads();
    bf16x8_t a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[i][e] = (__bf16)(1.0f + 0.001f * (float)((lane * 7 + i * 3 + e) & 63));
            b[i][e] = (__bf16)(1.0f - 0.001f * (float)((lane * 5 + i + e) & 63));
        }
    f32x16_t acc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    if (MODE == 3 || MODE == 4 || MODE == 7 || MODE == 8) {
        if (grp) bar();  // group 1 runs one barrier behind group 0
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
            run<16>(acc, a, b, 0);
        } else if constexpr (MODE == 1) {
            run<16>(acc, a, b, 0);
            bar();
        } else if constexpr (MODE == 2) {
            bar();
        } else if constexpr (MODE == 3) {
            __builtin_amdgcn_s_setprio(1);
            run<NR>(acc, a, b, 0);
            __builtin_amdgcn_s_setprio(0);
            bar();
            bar();  // end of the (empty) L phase
        } else if constexpr (MODE == 7 || MODE == 8) {  // mode 3 with NR / 2 accumulators used twice per run: 7 = at distance NR / 2, 8 = back to back
            __builtin_amdgcn_s_setprio(1);
            run2<NR / 2, MODE == 7 ? NR / 2 : 1>(acc, a, b);
            __builtin_amdgcn_s_setprio(0);
            bar();
            bar();
        } else if constexpr (MODE == 4) {
            __builtin_amdgcn_s_setprio(1);
            run<12>(acc, a, b, 0);
            __builtin_amdgcn_sched_barrier(0);
            bar();
            __builtin_amdgcn_sched_barrier(0);
            run<4>(acc, a, b, 12);
            __builtin_amdgcn_s_setprio(0);
            bar();
        } else if constexpr (MODE == 5) {
            __builtin_amdgcn_s_setprio(1);
            run<16>(acc, a, b, 0);
            __builtin_amdgcn_s_setprio(0);
            bar();
        } else if constexpr (MODE == 6) {
            // group g: run, publish "run it done", wait until the other group has published its run of the same slot parity
            __builtin_amdgcn_s_setprio(1);
            run<16>(acc, a, b, 0);
            __builtin_amdgcn_s_setprio(0);
            if (lane == 0 && (w & 3) == 0) flags[grp] = it + 1;
            while (flags[grp ^ 1] < it + (grp ? 1 : 0)) __builtin_amdgcn_s_sleep(1);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (MODE == 3 || MODE == 4 || MODE == 7 || MODE == 8) {
        if (!grp) bar();
    }
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m) s += acc[m][lane & 15];
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE, int NR = 16>
void measure(const char* what, int iters, float* out, unsigned long long* cyc, int cus) {
    const int lds = 150 * 1024;
    (void)hipFuncSetAttribute((const void*)handoff_kernel<MODE, NR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((handoff_kernel<MODE, NR>), dim3(cus), dim3(512), lds, 0, out, cyc, iters / 10 + 1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((handoff_kernel<MODE, NR>), dim3(cus), dim3(512), lds, 0, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(cus);
    hipMemcpy(h.data(), cyc, cus * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= cus;
    const double mfma_per_simd = MODE == 2 ? 0.0 : 2.0 * NR;  // two waves x NR per iteration
    printf("mode %d  %-62s %9.1f counter ticks / iteration   %8.3f us / 1000 iterations   %s\n", MODE, what, mean / iters, ms * 1000.0 / iters * 1000.0 / 1000.0,
           mfma_per_simd > 0 ? "" : "(no MFMAs)");
    if (mfma_per_simd > 0) {
        const double tf = (double)cus * 4 * mfma_per_simd * 32768.0 * iters / (ms * 1e-3) / 1e12;
        printf("        -> %.0f TFLOP/s, i.e. %.1f %% of the 2.5 PFLOP/s dense bf16 peak (%.0f ns per iteration; its %d pipe cycles at 2.4 GHz = %.0f ns)\n", tf, tf / 25.0, ms * 1e6 / iters, 64 * NR, 64 * NR / 2.4);
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, (size_t)cus * 512 * 4);
    hipMalloc(&cyc, (size_t)cus * 8);
    printf("%s: %d CUs, %d iterations per kernel\n", p.name, cus, iters);
    measure<0>("runs only (two waves per SIMD)", iters, out, cyc, cus);
    measure<1>("every wave: run, s_barrier", iters, out, cyc, cus);
    measure<2>("s_barrier only", iters, out, cyc, cus);
    measure<3>("ping-pong, 2 barriers / iteration (g tiles)", iters, out, cyc, cus);
    measure<3, 12>("ping-pong, 2 barriers / iteration, 12-MFMA runs (96 couts)", iters, out, cyc, cus);
    measure<3, 8>("ping-pong, 2 barriers / iteration, 8-MFMA runs (64 couts)", iters, out, cyc, cus);
    measure<7, 16>("as 3, 8 accumulators x 2 k-halves, reuse distance 8 (g128 order)", iters, out, cyc, cus);
    measure<7, 12>("as 3, 6 accumulators x 2 k-halves, reuse distance 6 (g96 order)", iters, out, cyc, cus);
    measure<7, 8>("as 3, 4 accumulators x 2 k-halves, reuse distance 4 (g64 order)", iters, out, cyc, cus);
    measure<8, 16>("as 3, 8 accumulators, the two k-halves back to back", iters, out, cyc, cus);
    measure<8, 12>("as 3, 6 accumulators, the two k-halves back to back", iters, out, cyc, cus);
    measure<8, 8>("as 3, 4 accumulators, the two k-halves back to back", iters, out, cyc, cus);
    measure<4>("ping-pong, barrier 4 MFMAs before the end of the run", iters, out, cyc, cus);
    measure<5>("one barrier / iteration, both groups run (h tiles' skeleton)", iters, out, cyc, cus);
    measure<6>("ping-pong on s_sleep-polled LDS flags", iters, out, cyc, cus);
    return 0;
}
