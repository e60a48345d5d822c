// What would 8-bit operands buy the conv loop on this MI355X?  The inner loop of the patch kernel in miniature: per sub-step every wave reads TI + TJ
// fragments from LDS and issues TI * TJ MFMAs on a 64 x 64 accumulator tile, 2 waves per SIMD, sustained for seconds so that the board's power cap
// governs (profiles/r03_power.txt: the real kernels sit at 1400 W).  Variants:
//   bf16   v_mfma_f32_32x32x16_bf16            16-byte fragments (ds_read_b128): 1 KB of LDS reads per MFMA   <- what conv_kernels.inc does
//   fp8    v_mfma_f32_32x32x16_fp8_fp8          8-byte fragments (ds_read_b64):  0.5 KB per MFMA, same FLOPs per instruction
//   mxfp8  v_mfma_scale_f32_32x32x64_f8f6f4    32-byte fragments (2 x ds_read_b128) for 4 x the K: 0.5 KB per bf16-MFMA-equivalent, 2 x the FLOP rate
//   noLDS  the bf16 MFMAs on registers only (the 1.75 PFLOP/s ceiling of DESIGN.md 5)
// hipcc --offload-arch=gfx950 -O3 tools/micro/operand_energy.hip -o tools/micro/operand_energy ; ./operand_energy [seconds per variant]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;

constexpr int TI = 2, TJ = 2;

template <int MODE>  // 0 bf16, 1 fp8, 2 mxfp8, 3 bf16 without LDS reads
__global__ __launch_bounds__(256, 2) void loop_kernel(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 32 KB: [512 rows][64 B], filled with a benign byte pattern
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // pseudo-random operands (data-dependent power: constant operands ran the register-only loop at 2.35 PFLOP/s and 1.04 kW, random ones pull the clock down):
    // bf16: +-[1, 2) with random mantissas; fp8 e4m3: +-[1, 1.875]
    for (int i = tid; i < 32 * 1024 / 4; i += 256) {
        unsigned h = (unsigned)i * 2654435761u;
        h ^= h >> 15;
        h *= 2246822519u;
        h ^= h >> 13;
        ((unsigned*)smem)[i] = (MODE == 0 || MODE == 3) ? ((h & 0x807f807fu) | 0x3f803f80u) : ((h & 0x87878787u) | 0x38383838u);
    }
    __syncthreads();
    f32x16_t acc[TI][TJ];
    for (int i = 0; i < TI; ++i)
        for (int j = 0; j < TJ; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const char* base = smem + ((lane & 31) * 64 + (lane >> 5) * 16);  // a fragment row per lane, as in the conv kernels
    int off = w * 2048;
    bf16x8_t ra[TI], rb[TJ];
    for (int i = 0; i < TI; ++i) ra[i] = *(const bf16x8_t*)(base + i * 2048 + w * 4096);
    for (int j = 0; j < TJ; ++j) rb[j] = *(const bf16x8_t*)(base + 8192 + j * 2048 + w * 4096);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int sub = 0; sub < 6; ++sub) {  // 6 sub-steps between "barriers", like a (channel block, kernel row) step
            const char* p = base + ((off + sub * 4096) & 0x7fff);
            if constexpr (MODE == 0 || MODE == 3) {
                bf16x8_t a[TI], b[TJ];
                if constexpr (MODE == 0) {
                    for (int i = 0; i < TI; ++i) a[i] = *(const bf16x8_t*)(p + i * 2048);
                    for (int j = 0; j < TJ; ++j) b[j] = *(const bf16x8_t*)(p + 8192 + j * 2048);
                } else {
                    // operands from LDS once per iteration (negligible), re-used by the six sub-steps: the MFMAs without their fragment traffic
                    if (sub == 0 || true) {
                        for (int i = 0; i < TI; ++i) a[i] = ra[i];
                        for (int j = 0; j < TJ; ++j) b[j] = rb[j];
                    }
                }
                for (int i = 0; i < TI; ++i)
                    for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            } else if constexpr (MODE == 1) {
                long a[TI], b[TJ];
                for (int i = 0; i < TI; ++i) a[i] = *(const long*)(p + i * 2048);
                for (int j = 0; j < TJ; ++j) b[j] = *(const long*)(p + 8192 + j * 2048);
                for (int i = 0; i < TI; ++i)
                    for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a[i], b[j], acc[i][j], 0, 0, 0);
            } else {
                i32x8_t a[TI], b[TJ];
                for (int i = 0; i < TI; ++i) {
                    const i32x4_t lo = *(const i32x4_t*)(p + i * 2048), hi = *(const i32x4_t*)(p + i * 2048 + 1024);
                    a[i] = i32x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
                for (int j = 0; j < TJ; ++j) {
                    const i32x4_t lo = *(const i32x4_t*)(p + 8192 + j * 2048), hi = *(const i32x4_t*)(p + 8192 + j * 2048 + 1024);
                    b[j] = i32x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
                for (int i = 0; i < TI; ++i)
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i], b[j], acc[i][j], 0 /*A fp8 e4m3*/, 0 /*B fp8 e4m3*/, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            }
        }
        off += 64;
    }
    float s = 0.f;
    for (int i = 0; i < TI; ++i)
        for (int j = 0; j < TJ; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const char* name, double flop_per_mfma, double seconds, float* out, int blocks) {
    const int iters = 2000;
    const double flops_per_launch = flop_per_mfma * TI * TJ * 6.0 * iters * blocks * 4.0;
    hipFuncSetAttribute((const void*)loop_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 1024);
    hipLaunchKernelGGL(loop_kernel<MODE>, dim3(blocks), dim3(256), 32 * 1024, 0, out, iters);
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    int launches = 0;
    double last_ms = 0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        hipEventRecord(e0);
        for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(loop_kernel<MODE>, dim3(blocks), dim3(256), 32 * 1024, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        last_ms = ms / 20;
        launches += 20;
    }
    printf("%-6s steady state after %d launches: %.3f ms per launch = %.1f TFLOP/s\n", name, launches, last_ms, flops_per_launch / last_ms / 1e9);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 6.0;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 2;  // two 4-wave blocks per CU = 2 waves per SIMD
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    run<3>("noLDS", 2.0 * 32 * 32 * 16, seconds, out, blocks);
    run<0>("bf16", 2.0 * 32 * 32 * 16, seconds, out, blocks);
    run<1>("fp8", 2.0 * 32 * 32 * 16, seconds, out, blocks);
    run<2>("mxfp8", 2.0 * 32 * 32 * 64, seconds, out, blocks);
    return 0;
}
