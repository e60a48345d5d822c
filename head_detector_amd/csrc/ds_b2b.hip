// "t" tile (r06): the stage-1 downsample (QARepVGG 3x3 / stride 2, 48 -> 96 channels, yolo_heads_{m,l}_arch_params.yaml:11-17) and the CSP layer's merged conv1|conv2
// (1x1, 96 -> 128 / 192) behind it -- the first back-to-back pair of the TorchScript blob called at head_detector/detector.py:58-59 -- as ONE PERSISTENT launch whose
// weights never move: every wave keeps its 32-cout slice of BOTH convs in registers for the whole launch.
//
// Why (profiles/r06_igemm_trace.txt, r06_family_table_l64.md): on the implicit-GEMM b2b tile this pair is the slowest op of the forward after the 512-cout head conv
// (484 us per 64 images at 1.97 TB/s).  A K step of that tile is one loaded memory round trip (~1 500 cycles for 192 cycles of MFMAs), it has 18 of them per tile, and the
// 3 x 3 taps pull every input line through the LDS-DMA path up to nine times (3.2 GB of L2 -> LDS traffic for 0.63 GB of input).  Here
//   * a tile = 8 x 8 output pixels of one image; its 17 x 17 x 48-channel input patch is fetched ONCE (1.13 x the input bytes), all 36 LDS-DMA pieces of it in flight
//     together and a whole tile ahead (double-buffered), so a tile costs one round trip, hidden under the previous tile;
//   * the patch lands de-interleaved into the four parity planes [iy & 1][ix & 1] (the LDS-DMA source address is free per lane), so tap (ky, kx) of the stride-2 conv
//     reads plane (ky & 1, kx & 1) at a stride-1 offset: a plane row = one 1-KiB piece = 9 pixels x 6 sixteen-byte chunks, rotated by one chunk on odd rows -- the 16 lanes
//     of every ds_read_b128 lane group (MI355X_MICROARCH.md, LDS: {0-3, 12-15, 20-27}, ...) then cover two adjacent rows x eight pixels = 16 distinct 16-byte slots;
//   * wave w of the 3-wave workgroup owns couts 32 w .. + 31 of the 3x3 conv for both 32-pixel groups: 27 A fragments (9 taps x three 16-channel steps; the padded
//     channels 48 .. 63 of the implicit-GEMM launch carry zero weights and are skipped: they only ever added exact zeros) = 108 VGPRs, loaded once;
//   * the three waves trade their bf16 results through LDS in fragment order (what the b2b tile's one wave holds in registers) and run the 1x1 conv with cout groups
//     w and w + 3 from 24 more resident fragment registers each; only the second conv's output is stored.
// Same instructions (v_mfma_f32_32x32x16_bf16), same operand slots, same k order (tap major, channels ascending; then k-block, half) and same roundings as the two
// implicit-GEMM launches: BIT-IDENTICAL outputs (tests/test_gpu_parity.py::test_b2b_pairs_equal_their_two_launches).
// Two workgroups per CU (72 KB of LDS, <= 256 registers each): one computes while the other waits for its barrier.
#include <atomic>

#include "vgh_internal.h"

#define AS3 __attribute__((address_space(3)))
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

namespace {

constexpr int DT_ROW = 1024;               // bytes of one plane row (one LDS-DMA piece): 9 pixels x 96 B + the rotation chunk, padded to 64 chunks
constexpr int DT_PLANE = 9 * DT_ROW;       // planes (0, x) hold 9 rows, planes (1, x) 8 (their ninth row is filled with zeros and never read)
constexpr int DT_BUF = 4 * DT_PLANE;       // one patch
constexpr int DT_BIAS = 2 * DT_BUF;        // bias vectors of both convs (fp32), staged once per workgroup
constexpr int DT_LDS = DT_BIAS + 2048;
constexpr unsigned DT_OOB = 0xC0000000u;   // out of the descriptors' 2-GiB range, and not wrapped past 2^32 by the scalar offsets added to it
static_assert(2 * DT_LDS <= 160 * 1024, "two workgroups per CU");

// experiments build: s_memtime marks per (workgroup, wave, tile) behind the 8192 rows the other conv kernels use of the tools' trace buffer (tools/ds_trace.py)
#ifdef VGH_EXPERIMENTS
#define DT_MARK(k)                                                                                                                                         \
    do {                                                                                                                                                   \
        if (a.trace && lane == 0 && tno < 8) a.trace[((size_t)8192 * VGH_TRACE_TILES * VGH_TRACE_MARKS) + ((size_t)(blockIdx.x * 3 + w) * 8 + tno) * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define DT_MARK(k) \
    do {           \
    } while (0)
#endif

struct DtDiv {
    unsigned m_per, s_per, m_nsx, s_nsx;
};
__device__ __forceinline__ int dt_div(int n, unsigned m, unsigned s) { return (int)((__umulhi((unsigned)n, m) + (unsigned)n) >> s); }
template <int N>
__device__ __forceinline__ void dt_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void dt_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ unsigned dt_pk(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    const f32x2_t f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf2));
}
__device__ __forceinline__ void dt_sw32(unsigned& x, unsigned& y) {  // lanes 32-63 of x trade places with lanes 0-31 of y
    const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    x = r[0];
    y = r[1];
}

// 8 accumulator values (runs q = 2 m, 2 m + 1 of a 32-cout group) -> this lane's 16 bytes: + bias (packed fp32 adds), bf16, ReLU on the packed bf16 values (v_pk_max_i16
// against 0: a bf16 is negative as int16 exactly when the float is; bound INT16_MIN = no activation) -- the same bits as max(x + b, 0) -> bf16 --, half-wave exchange
__device__ __forceinline__ u32x4_t dt_epi8(const f32x16_t& acc, int m, const f32x4_t b0, const f32x4_t b1, unsigned bound) {
    typedef __attribute__((ext_vector_type(2))) short s16x2;
    const f32x2_t a0 = {acc[8 * m + 0], acc[8 * m + 1]}, a1 = {acc[8 * m + 2], acc[8 * m + 3]}, a2 = {acc[8 * m + 4], acc[8 * m + 5]}, a3 = {acc[8 * m + 6], acc[8 * m + 7]};
    const f32x2_t s0 = a0 + f32x2_t{b0[0], b0[1]}, s1 = a1 + f32x2_t{b0[2], b0[3]}, s2 = a2 + f32x2_t{b1[0], b1[1]}, s3 = a3 + f32x2_t{b1[2], b1[3]};
    auto relu = [&](unsigned d) { return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, d), __builtin_bit_cast(s16x2, bound))); };
    unsigned pa0 = relu(dt_pk(s0[0], s0[1])), pa1 = relu(dt_pk(s1[0], s1[1])), pb0 = relu(dt_pk(s2[0], s2[1])), pb1 = relu(dt_pk(s3[0], s3[1]));
    dt_sw32(pa0, pb0);
    dt_sw32(pa1, pb1);
    return u32x4_t{pa0, pa1, pb0, pb1};
}

// T2 = cout groups (of 32) of the second conv: 4 (M) or 6 (L)
template <int T2>
__global__ __launch_bounds__(192, 2) void ds_b2b_kernel(const ConvArgs a, const int nsx, const int per, const int total_tiles, const int chunk, const DtDiv dv) {
    constexpr int NT2 = (T2 + 2) / 3;  // cout groups of the second conv per wave (wave w: groups w, w + 3)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n32 = lane & 31, hi = lane >> 5, half4 = hi * 4;
    const int xcd = blockIdx.x & 7, gpx = gridDim.x >> 3;
    const int pixb = (int)a.in_pitch * 2;  // bytes per input pixel (96)
    const int rowb = a.W * pixb;

    // ---- once per workgroup: bias vectors -> LDS, weights -> registers ----
    {
        float* const bl = (float*)(smem + DT_BIAS);
        for (int i = tid; i < 96; i += 192) bl[i] = a.bias[i];
        for (int i = tid; i < 32 * T2; i += 192) bl[96 + i] = a.bias2[i];
    }
    bf16x8_t W1[27];  // A fragments of the 3x3 conv: step s = tap * 3 + c covers channels 16 c .. + 15 of tap (ky, kx); lane (n32, hi) = cout 32 w + n32, k = 8 hi .. + 7
    {
        const int co = w * 32 + n32, sw = (co >> 2) & 3;
#pragma unroll
        for (int s = 0; s < 27; ++s) {
            const int tap = s / 3, c = s % 3, kb = tap * 2 + (c >> 1), ch = 2 * (c & 1) + hi;
            W1[s] = *(const bf16x8_t*)(a.wpack + ((size_t)kb * 96 + co) * 32 + ((ch ^ sw) * 8));
        }
    }
    bf16x8_t W2[NT2][3][2];  // A fragments of the 1x1 conv: cout group w + 3 tt, k-block i (32 channels), half m
#pragma unroll
    for (int tt = 0; tt < NT2; ++tt) {
        const int t = w + 3 * tt;
        const int co = (t < T2 ? t : 0) * 32 + n32, sw = (co >> 2) & 3;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int m = 0; m < 2; ++m) W2[tt][i][m] = *(const bf16x8_t*)(a.w2pack + ((size_t)i * (32 * T2) + co) * 32 + (((2 * m + hi) ^ sw) * 8));
    }

    // ---- per-lane constants ----
    // LDS-DMA source offsets inside a plane row, by the row's parity (odd rows are rotated by one chunk): lane l holds chunk s = l - parity: pixel s / 6, chunk s % 6.
    // q = 0 / 1: rows whose parity equals / differs from this wave's (unit i of a wave fills plane row 3 (i % 3) + w)
    unsigned rel0[2], rel1[2], relL[2];  // planes (y, 0) / planes (y, 1): eight pixels / planes (y, 0) of a tile on the left image edge (pixel 0 = column -1: zeros)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int s = lane - ((q ^ w) & 1);
        const bool valid = (unsigned)s < 54u;
        const int col = s / 6, chn = s - col * 6;
        const unsigned r = VGH_ABLATE(a, 16) ? (unsigned)(s * 16) : (unsigned)(col * 2 * pixb + chn * 16);  // (16: timing experiment, contiguous source bytes -- wrong results)
        rel0[q] = valid ? r : DT_OOB;
        rel1[q] = (valid && col < 8) ? r : DT_OOB;
        relL[q] = (valid && col > 0) ? r : DT_OOB;
    }
    // this lane's output pixel inside a pixel group: rows / columns chosen per ds_read_b128 lane group (two adjacent rows x eight columns each)
    int ty4, tx;
    if (n32 < 4) ty4 = 0, tx = n32;
    else if (n32 < 12) ty4 = 2, tx = n32 - 4;
    else if (n32 < 16) ty4 = 0, tx = n32 - 8;
    else if (n32 < 20) ty4 = 3, tx = n32 - 16;
    else if (n32 < 28) ty4 = 1, tx = n32 - 20;
    else ty4 = 3, tx = n32 - 24;
    unsigned rb[2];  // fragment read offset of the lane for taps with ky >> 1 = 0 / 1 (the row parity moves the rotation)
#pragma unroll
    for (int kyh = 0; kyh < 2; ++kyh) rb[kyh] = (unsigned)(ty4 * DT_ROW + (tx * 6 + hi + ((ty4 + kyh) & 1)) * 16);
    const unsigned bound1 = a.act == VGH_ACT_RELU ? 0u : 0x80008000u, bound2 = a.act2 == VGH_ACT_RELU ? 0u : 0x80008000u;
    unsigned ovo[2];  // byte offset of the lane's output pixel (+ its 8-channel half) of pixel group j from the tile's first pixel
#pragma unroll
    for (int j = 0; j < 2; ++j) ovo[j] = (unsigned)(((4 * j + ty4) * a.Wo + tx) * (int)a.out2_pitch * 2 + hi * 16);
    uint16_t* const out2 = (uint16_t*)a.out2;
    const float* const bl = (const float*)(smem + DT_BIAS);

    // one tile's patch -> buffer `buf`: 12 pieces per wave (piece i of wave w = row 3 (i % 3) + w of plane i / 3)
    auto issue_patch = [&](int tile, int buf) {
        const int b = dt_div(tile, dv.m_per, dv.s_per);
        const int rem = tile - b * per;
        const int tyi = dt_div(rem, dv.m_nsx, dv.s_nsx), txi = rem - tyi * nsx;
        const int iy0 = 16 * tyi - 1, ix0 = 16 * txi - 1;  // input pixel of patch position (0, 0)
        const bool top = tyi == 0, left = txi == 0;
        const char* const base = (const char*)a.in + (int64_t)a.in_coff * 2 + ((int64_t)(b * a.H + iy0) * a.W + ix0) * pixb;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x80000000, 0x00020000);
        char* const dst = smem + buf * DT_BUF;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int plane = i / 3, py = plane >> 1, px = plane & 1, q = (i % 3) & 1;
            const int r = 3 * (i % 3) + w;
            const int lr = 2 * r + py;
            const bool row_ok = lr <= 16 && !(top && lr == 0);
            unsigned vo = px ? rel1[q] : (left ? relL[q] : rel0[q]);
            vo = row_ok ? vo : DT_OOB;
            const unsigned so = row_ok ? (unsigned)(lr * rowb + px * pixb) : 0u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (AS3 void*)(dst + (plane * 9 + r) * DT_ROW), 16, vo, so, 0, 0);
        }
    };

    int local = blockIdx.x >> 3;
    auto tile_of = [&](int l) { return (l < chunk && xcd * chunk + l < total_tiles) ? xcd * chunk + l : -1; };
    int tile = tile_of(local);
    if (tile < 0) return;
    issue_patch(tile, 0);
    int cur = 0;
    bool first = true;
    int tno = 0;
    (void)tno;
    while (true) {
        const int nxt_tile = tile_of(local + gpx);
        DT_MARK(0);
        // ---- this tile's patch has landed (counted: the previous tile's stores, issued behind it, may still be in flight) ----
        if (first) {
            dt_wait_vm<0>();
        } else if (NT2 == 2 && (T2 == 6 || w == 0)) {
            dt_wait_vm<8>();
        } else {
            dt_wait_vm<4>();
        }
        first = false;
        DT_MARK(1);
        dt_barrier();  // everybody's pieces landed; everybody is done with the other buffer (the previous tile's exchange area)
        DT_MARK(2);
        if (nxt_tile >= 0 && !VGH_ABLATE(a, 1)) issue_patch(nxt_tile, cur ^ 1);
        DT_MARK(3);
        const char* const xb = smem + cur * DT_BUF;

        // ---- 3x3 / stride-2 conv: 27 k steps x two pixel groups ----
        f32x16_t acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
        {
            // fragment reads run two k steps ahead of the MFMAs (three pairs of registers): one wave has to cover its own LDS latency -- the CU's other workgroup is
            // as likely at a barrier as in its K loop
            auto frag = [&](int s, int j) -> bf16x8_t {
                const int ky = s / 9, kx = (s / 3) % 3, c = s % 3;
                const int imm = ((ky & 1) * 2 + (kx & 1)) * DT_PLANE + (4 * j + (ky >> 1)) * DT_ROW + ((kx >> 1) * 6 + c * 2) * 16;
                return *(const bf16x8_t*)(xb + rb[ky >> 1] + imm);
            };
            bf16x8_t F[3][2];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int j = 0; j < 2; ++j) F[p][j] = frag(p, j);
            if (!VGH_ABLATE(a, 2))
#pragma unroll
            for (int s = 0; s < 27; ++s) {
                if (s + 2 < 27) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) F[(s + 2) % 3][j] = frag(s + 2, j);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W1[s], F[s % 3][j], acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- its epilogue in registers: bias, ReLU, bf16, half-wave exchange -> lane (pixel n, half hi) holds channels 32 w + 16 m + 8 hi .. + 7: the B operand of
        //      k step (w, m) of the second GEMM ----
        bf16x8_t B2[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const f32x4_t b0 = *(const f32x4_t*)(bl + w * 32 + (2 * m) * 8 + half4), b1 = *(const f32x4_t*)(bl + w * 32 + (2 * m + 1) * 8 + half4);
                B2[j][m] = __builtin_bit_cast(bf16x8_t, dt_epi8(acc[j], m, b0, b1, bound1));
            }
        DT_MARK(4);
        dt_barrier();  // every wave is done reading this patch: its first 12 KB become the exchange area [j][producer wave][m][lane]
        char* const ex = smem + cur * DT_BUF;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int m = 0; m < 2; ++m) *(bf16x8_t*)(ex + ((j * 3 + w) * 2 + m) * 1024 + lane * 16) = B2[j][m];
        DT_MARK(5);
        dt_barrier();
        DT_MARK(6);

        // ---- 1x1 conv: K = 96 in the k order of a plain launch (k-block i, half m), cout groups w and w + 3; stores ----
        {
            const int b = dt_div(tile, dv.m_per, dv.s_per);
            const int rem = tile - b * per;
            const int tyi = dt_div(rem, dv.m_nsx, dv.s_nsx), txi = rem - tyi * nsx;
            // the tile's output through a buffer descriptor based at its first pixel: per-lane 32-bit offsets (set once), the channel offset is scalar
            const __amdgpu_buffer_rsrc_t orsrc =
                __builtin_amdgcn_make_buffer_rsrc((void*)(out2 + ((size_t)(b * a.Ho + tyi * 8) * a.Wo + txi * 8) * a.out2_pitch), 0, 0x80000000, 0x00020000);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16_t acc2[NT2];
#pragma unroll
                for (int tt = 0; tt < NT2; ++tt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[tt][r] = 0.0f;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const bf16x8_t bf = *(const bf16x8_t*)(ex + ((j * 3 + i) * 2 + m) * 1024 + lane * 16);
#pragma unroll
                        for (int tt = 0; tt < NT2; ++tt) acc2[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2[tt][i][m], bf, acc2[tt], 0, 0, 0);
                    }
#pragma unroll
                for (int tt = 0; tt < NT2; ++tt) {
                    const int t = w + 3 * tt;
                    if (t < T2) {
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            const f32x4_t b0 = *(const f32x4_t*)(bl + 96 + t * 32 + (2 * m) * 8 + half4), b1 = *(const f32x4_t*)(bl + 96 + t * 32 + (2 * m + 1) * 8 + half4);
                            const u32x4_t v = dt_epi8(acc2[tt], m, b0, b1, bound2);
                            const int c = t * 32 + 16 * m;  // first cout of the 16 this instruction stores per pixel (wave-uniform; this lane: + 8 hi)
                            const int ochan = (c >= a.out2_split) ? a.out2_coff2 + (c - a.out2_split) : a.out2_coff + c;
                            if (!VGH_ABLATE(a, 8)) __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, ovo[j], (unsigned)ochan * 2u, 0);
                        }
                    }
                }
            }
        }
        DT_MARK(7);
        ++tno;
        if (nxt_tile < 0) break;
        tile = nxt_tile;
        local += gpx;
        cur ^= 1;
    }
}

constexpr int kMaxDev = 16;
template <int T2>
int launch_dt(const ConvArgs& a, hipStream_t st) {
    static std::atomic<int> done[kMaxDev];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) dev = 0;
    if (!done[dev].load(std::memory_order_acquire)) {
        VGH_HIP(hipFuncSetAttribute((const void*)ds_b2b_kernel<T2>, hipFuncAttributeMaxDynamicSharedMemorySize, DT_LDS));
        done[dev].store(1, std::memory_order_release);
    }
    const int nsx = a.Wo / 8, nsy = a.Ho / 8, per = nsx * nsy;
    const int64_t total = (int64_t)a.B * per;
    VGH_REQUIRE(total < (1ll << 30), "conv b2b: too many tiles");
    const int chunk = (int)((total + 7) / 8);
    int gpx = vgh_conv_persistent_blocks_per_xcd(a, chunk, VGH_ABLATE(a, 32) ? 1 : 2);  // two workgroups per CU (shared with the executor's other lane streams)
    if (gpx > chunk) gpx = chunk;
    DtDiv dv;
    vgh_fastdiv_magic((unsigned)per, &dv.m_per, &dv.s_per);
    vgh_fastdiv_magic((unsigned)nsx, &dv.m_nsx, &dv.s_nsx);
    hipLaunchKernelGGL((ds_b2b_kernel<T2>), dim3(gpx * 8), dim3(192), DT_LDS, st, a, nsx, per, (int)total, chunk, dv);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}

}  // namespace

// the pair a (3x3 / stride 2 / pad 1, 48 real input channels at a 48-channel pitch declared as cin = 64, 96 couts) -> 1x1 with 128 / 192 couts, on a map of whole 8 x 8 tiles
int vgh_conv_ds_b2b_ok(const ConvArgs& a) {
    return a.ksize == 3 && a.stride == 2 && a.pad == 1 && a.cin == 64 && a.in_pitch == 48 && a.in_coff % 8 == 0 && a.cout_pad == 96 && a.cout_store == 96 && (a.cout2_pad == 128 || a.cout2_pad == 192) && a.cout2_store == a.cout2_pad &&
           a.H == 2 * a.Ho && a.W == 2 * a.Wo && a.Ho % 8 == 0 && a.Wo % 8 == 0 && (int64_t)a.W * a.in_pitch * 2 * 20 < (1ll << 30) && !a.split && !a.res && !a.shuffle && !a.grp_cout &&
           !a.in_fp8 && !a.out_fp8 && !a.out_f32 && a.act != VGH_ACT_SILU && a.act2 != VGH_ACT_SILU;
}

// `a` prepared, with its b2b fields set and checked by vgh_launch_conv_b2b
int vgh_launch_conv_ds_b2b(const ConvArgs& a, hipStream_t stream) {
    VGH_REQUIRE(vgh_conv_ds_b2b_ok(a), "conv b2b: not a stage-1 downsample pair (the t tile)");
    return a.cout2_pad == 192 ? launch_dt<6>(a, stream) : launch_dt<4>(a, stream);
}
