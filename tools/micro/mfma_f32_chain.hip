// Is a FP32 matrix instruction an ordered fmaf chain over its k?  (csrc/flame.hip rests on it for v_mfma_f32_32x32x2_f32: every FLAME vertex kernel must produce the
// bits of the VALU kernel's ascending-k fmaf chain.)  Checks v_mfma_f32_32x32x2_f32 (K = 2) and v_mfma_f32_16x16x4_f32 (K = 4) on random operands against the
// candidate orders (ascending chain, descending chain, pairwise tree, unfused products) bit for bit, and times a DEPENDENT chain of each (shader cycles per
// instruction: what bounds a FLAME decode of a few heads).   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off mfma_f32_chain.hip -o mfma_f32_chain
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4_t;

__global__ void k32(const float* A, const float* B, const float* C, float* D) {  // A[32][2], B[2][32], C / D [32][32]
    const int l = threadIdx.x, j = l & 31, h = l >> 5;
    f32x16_t c;
    for (int r = 0; r < 16; ++r) c[r] = C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + j];
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[j * 2 + h], B[h * 32 + j], c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + j] = c[r];
}
__global__ void k16(const float* A, const float* B, const float* C, float* D) {  // A[16][4], B[4][16], C / D [16][16]
    const int l = threadIdx.x, j = l & 15, q = l >> 4;
    f32x4_t c;
    for (int r = 0; r < 4; ++r) c[r] = C[(4 * q + r) * 16 + j];
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j * 4 + q], B[q * 16 + j], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * q + r) * 16 + j] = c[r];
}
template <int W>
__global__ void chain(float* out, long long* cyc, int n, float a0, float b0) {
    const float a = a0 * (threadIdx.x % 5), b = b0 * (threadIdx.x % 3);
    f32x16_t c32;
    f32x4_t c16;
    for (int r = 0; r < 16; ++r) c32[r] = 0.f;
    for (int r = 0; r < 4; ++r) c16[r] = 0.f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        if (W == 32) c32 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c32, 0, 0, 0);
        else c16 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c16, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c32[r];
    for (int r = 0; r < 4; ++r) s += c16[r];
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
static float rnd() {  // mixed magnitudes so that every rounding shows
    const float m = (float)rand() / RAND_MAX * 2.f - 1.f;
    return ldexpf(m, rand() % 9 - 4);
}
static bool same(float x, float y) { return memcmp(&x, &y, 4) == 0; }

int main() {
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 4096), hipMalloc(&dB, 4096), hipMalloc(&dC, 8192), hipMalloc(&dD, 8192);
    const char* names[4] = {"ascending fmaf chain", "descending fmaf chain", "pairwise tree of fused products", "unfused products, ascending adds"};
    for (int W : {32, 16}) {
        const int M = W, K = W == 32 ? 2 : 4, trials = 2000;
        long bad[4] = {0, 0, 0, 0};
        std::vector<float> A(M * K), B(K * M), C(M * M), D(M * M);
        srand(7 + W);
        for (int t = 0; t < trials; ++t) {
            for (auto& x : A) x = rnd();
            for (auto& x : B) x = rnd();
            for (auto& x : C) x = rnd();
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice), hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice), hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
            if (W == 32) hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
            else hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
            hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
            for (int i = 0; i < M; ++i)
                for (int j = 0; j < M; ++j) {
                    float up = C[i * M + j], dn = C[i * M + j], uf = C[i * M + j];
                    for (int k = 0; k < K; ++k) up = fmaf(A[i * K + k], B[k * M + j], up), uf = uf + A[i * K + k] * B[k * M + j];
                    for (int k = K - 1; k >= 0; --k) dn = fmaf(A[i * K + k], B[k * M + j], dn);
                    float tr;
                    if (K == 2) tr = fmaf(A[i * K], B[j], A[i * K + 1] * B[M + j]) + C[i * M + j];
                    else tr = (fmaf(A[i * K], B[j], A[i * K + 1] * B[M + j]) + fmaf(A[i * K + 2], B[2 * M + j], A[i * K + 3] * B[3 * M + j])) + C[i * M + j];
                    const float cand[4] = {up, dn, tr, uf};
                    for (int c = 0; c < 4; ++c) bad[c] += !same(cand[c], D[i * M + j]);
                }
        }
        printf("v_mfma_f32_%dx%dx%d_f32, %d random tiles (%ld outputs):\n", W, W, K, trials, (long)trials * M * M);
        for (int c = 0; c < 4; ++c) printf("    %-36s %ld outputs differ%s\n", names[c], bad[c], bad[c] == 0 ? "   <== bit-exact" : "");
    }
    float* dO;
    long long* dT;
    hipMalloc(&dO, 1024), hipMalloc(&dT, 64);
    for (int W : {32, 16}) {
        long long cyc = 0;
        const int n = 4096;
        for (int rep = 0; rep < 3; ++rep) {
            if (W == 32) hipLaunchKernelGGL(chain<32>, dim3(1), dim3(64), 0, 0, dO, dT, n, 1e-3f, 2e-3f);
            else hipLaunchKernelGGL(chain<16>, dim3(1), dim3(64), 0, 0, dO, dT, n, 1e-3f, 2e-3f);
            hipMemcpy(&cyc, dT, 8, hipMemcpyDeviceToHost);
        }
        printf("dependent chain of %d %s: %.1f shader cycles per instruction (k per instruction %d -> %.1f cycles per k)\n", n, W == 32 ? "v_mfma_f32_32x32x2_f32" : "v_mfma_f32_16x16x4_f32",
               (double)cyc / n, W == 32 ? 2 : 4, (double)cyc / n / (W == 32 ? 2 : 4));
    }
    return 0;
}
