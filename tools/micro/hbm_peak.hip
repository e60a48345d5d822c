// Calibration micro-benchmark: sustained HBM bandwidth of this MI355X for the access mixes the 1x1 convs produce
// (streaming 16-byte reads and writes, read:write ratio as given).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4_t;

// each thread: RD reads (16 B, different streams) -> 1 write
template <int RD>
__global__ __launch_bounds__(256) void stream_kernel(const f32x4_t* __restrict__ in, f32x4_t* __restrict__ out, size_t n_out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_out; i += (size_t)gridDim.x * 256) {
        f32x4_t acc = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < RD; ++r) {
            const f32x4_t v = in[(size_t)r * n_out + i];
            acc += v;
        }
        out[i] = acc;
    }
}
__global__ __launch_bounds__(256) void read_kernel(const f32x4_t* __restrict__ in, float* __restrict__ out, size_t n) {
    f32x4_t acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += in[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.0f;
}
__global__ __launch_bounds__(256) void write_kernel(f32x4_t* __restrict__ out, size_t n) {
    const f32x4_t v = {1, 2, 3, 4};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = v;
}

template <typename F>
float time_it(F f, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    f();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    const size_t MB = 1 << 20;
    const size_t out_bytes = 160 * MB;  // one 1x1-conv output tensor at B=32
    f32x4_t *in, *out;
    hipMalloc(&in, 4 * out_bytes);
    hipMalloc(&out, out_bytes);
    hipMemset(in, 0, 4 * out_bytes);
    const size_t n_out = out_bytes / 16;
    for (int blocks : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
        float r = time_it([&] { hipLaunchKernelGGL(read_kernel, dim3(blocks), dim3(256), 0, 0, in, (float*)out, 4 * n_out); }, 10);
        float w = time_it([&] { hipLaunchKernelGGL(write_kernel, dim3(blocks), dim3(256), 0, 0, out, n_out); }, 10);
        float c1 = time_it([&] { hipLaunchKernelGGL(stream_kernel<1>, dim3(blocks), dim3(256), 0, 0, in, out, n_out); }, 10);
        float c3 = time_it([&] { hipLaunchKernelGGL(stream_kernel<3>, dim3(blocks), dim3(256), 0, 0, in, out, n_out); }, 10);
        printf("blocks %5d: read 640MB %.3f ms = %.2f TB/s | write 160MB %.3f ms = %.2f TB/s | 1r:1w %.3f ms = %.2f TB/s | 3r:1w %.3f ms = %.2f TB/s\n", blocks, r,
               4 * out_bytes / r / 1e9, w, out_bytes / w / 1e9, c1, 2 * out_bytes / c1 / 1e9, c3, 4 * out_bytes / c3 / 1e9);
    }
    return 0;
}
