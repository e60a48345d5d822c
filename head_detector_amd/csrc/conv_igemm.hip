// Implicit-GEMM convolution for gfx950 (MI355X): NHWC bf16 activations, MFMA 32x32x16 bf16 with
// fp32 accumulation, fused epilogue {bias, ReLU/SiLU, alpha*residual, concat-by-offset store,
// two-segment store, fp32 store, ConvTranspose pixel-shuffle store}.
//
// Replaces (reference: every conv inside the TorchScript blob called at head_detector/detector.py:58-59;
// definition yolo_head_training/configs/arch_params/yolo_heads_{m,l}_arch_params.yaml:4-137 +
// yolo_head_training/yolo_head/yolo_head_dfl_head.py:74-135): the eval-mode QARepVGG / Conv+BN+ReLU /
// ConvBNReLU / ConvTranspose2d blocks, after folding to one conv + bias.
//
// GEMM view:  D[cout][pixel] = sum_k Wt[cout][k] * X[pixel][k],  k = (ky, kx, cin) in 32-channel k-blocks.
//   A operand (MFMA rows i)  = weights  -> every lane ends up holding 4 *consecutive couts* of one pixel
//   B operand (MFMA cols j)  = pixels      per accumulator quad => 8-byte bf16x4 stores, NHWC-contiguous.
// Staging: both tiles go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction).
//   LDS tiles are [rows][32 bf16] = 64 B rows; the four 16-B chunks of a row are XOR-swizzled with
//   ((row>>2)&3) so that ds_read_b128 fragment reads (lane = row) are bank-conflict free.  Because an
//   LDS-DMA destination is lane-linear, the swizzle is applied on the SOURCE side: per-lane gather
//   address for activations (which also implements im2col + zero padding), host pre-swizzled image for
//   weights (vgh_pack_conv_weights_host).
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "vgh_internal.h"

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))
#define AS4 __attribute__((address_space(4)))

namespace {

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const AS1 void*)gsrc, (AS3 void*)lds_wave_base, 16, 0, 0);
}

// LDS-DMA through a buffer descriptor: 16 bytes per lane to lds_wave_base + lane*16; an out-of-range voffset reads zeros.
__device__ __forceinline__ void bload_lds16(const void* base, unsigned voffset, unsigned soffset, char* lds_wave_base) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x80000000, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (AS3 void*)lds_wave_base, 16, voffset, soffset, 0, 0);
}

__device__ __forceinline__ int fastdiv(int n, unsigned m, unsigned s) { return (int)((__umulhi((unsigned)n, m) + (unsigned)n) >> s); }

// 9-bit (ky*3+kx) "tap inside the image" mask of a 3x3 window centred (pad 1) on (cy, cx); 1 bit for a 1x1 conv
__device__ __forceinline__ unsigned tap_mask(int cy, int cx, int H, int W, int ksize) {
    if (ksize == 1) return 1u;
    const unsigned xm = (cx > 0 ? 1u : 0u) | 2u | (cx + 1 < W ? 4u : 0u);
    return (cy > 0 ? xm : 0u) | (xm << 3) | (cy + 1 < H ? xm << 6 : 0u);
}

__device__ __forceinline__ float act_fn(float v, int act) {
    if (act == VGH_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == VGH_ACT_SILU) return v / (1.0f + __expf(-v));
    return v;
}
// Branch-free activation for the epilogues: with a runtime `act` inside the unrolled element loops hipcc emits a scalar
// compare + branch PER ELEMENT.  ReLU / none become ONE v_max_f32 against a per-launch bound: 0 for ReLU, a quiet NaN for
// "none" (IEEE maxNum(v, NaN) = v, so the value -- including a NaN accumulator -- passes through unchanged).  SiLU (unused by
// the VGGHeads graphs) is applied afterwards under one wave-uniform branch per output vector.
__device__ __forceinline__ float act_bound(int act) { return act == VGH_ACT_RELU ? 0.0f : __builtin_nanf(""); }
__device__ __forceinline__ float silu_fn(float v) { return v / (1.0f + __expf(-v)); }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most `stages_in_flight` * LPS of this wave's LDS-DMA loads are outstanding (counted, never a full drain)
template <int LPS>
__device__ __forceinline__ void wait_stages(int stages_in_flight) {
    switch (stages_in_flight) {
        case 0: wait_vmcnt<0>(); break;
        case 1: wait_vmcnt<LPS>(); break;
        case 2: wait_vmcnt<2 * LPS>(); break;
        default: wait_vmcnt<3 * LPS>(); break;
    }
}

template <int BP, int BC, int WP, int WC, int KBS, int EPI, int NST>
__global__ __launch_bounds__((BP / WP) * (BC / WC) * 64, ((BP / WP) * (BC / WC) >= 8 ? ((BP / WP) * (BC / WC)) / 4 : ((WP / 32) * (WC / 32) <= 4 ? (NST > 2 ? 2 : 3) : 1))) void conv_igemm_kernel(const ConvArgs a, const int ntc, const int total_tiles,
                                                                                 const int chunk) {
    constexpr int NWP = BP / WP, NWC = BC / WC, NW = NWP * NWC;
    constexpr int XR = BP / 16 / NW;  // activation row-blocks (16 rows = 1 KiB) staged by each wave per k-block
    constexpr int WRB = BC / 16;      // weight row-blocks per k-block
    constexpr int WR = (WRB + NW - 1) / NW;
    constexpr int XBYTES = BP * 64, WBYTES = BC * 64;
    constexpr int STAGE = KBS * (XBYTES + WBYTES);
    constexpr int TI = WC / 32, TJ = WP / 32;
    static_assert(BP % (16 * NW) == 0, "BP must split into 16-row blocks across waves");
    static_assert(WP % 32 == 0 && WC % 32 == 0, "wave tile is made of 32x32 MFMA tiles");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    // XCD-aware tile order: physical block b runs on XCD b%8; give each XCD a contiguous run of
    // logical tiles (cout-tile fastest) so neighbouring pixel tiles (3x3 halos) share one L2.
    const int bid = blockIdx.x;
    const int tile = (bid & 7) * chunk + (bid >> 3);
    if (tile >= total_tiles) return;
    const int ptile = tile / ntc, ctile = tile - ptile * ntc;
    const int p0 = ptile * BP, c0 = ctile * BC;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = w % NWP, wc = w / NWP;

    // ---- loader ------------------------------------------------------------------------------------
    // LDS-DMA through buffer descriptors (buffer_load_dwordx4 ... lds): everything per-lane is computed ONCE --
    // a 32-bit offset of the row's centre tap and a 9-bit "tap in bounds" mask -- and the per-k-block work is scalar:
    // the tap / channel-block displacement moves the descriptor BASE (2 SALU), the weight tile is addressed through
    // soffset.  A lane whose tap falls into the zero padding uses an out-of-range voffset: the hardware range check
    // returns 0 for it, which is exactly the im2col zero.  (The counters showed ~17 VALU+SALU per MFMA with 64-bit
    // per-lane pointer arithmetic; the matrix pipe starved on instruction issue, not on bandwidth.)
    const int HoWo = a.Ho * a.Wo;
    unsigned xoff[XR], xmask[XR];
#pragma unroll
    for (int t = 0; t < XR; ++t) {
        const int r = (w + NW * t) * 16 + (lane >> 2);
        const int m = p0 + r;
        const bool valid = m < a.P;
        const int mm = valid ? m : 0;
        const int b = fastdiv(mm, a.div_howo_m, a.div_howo_s);
        const int rem = mm - b * HoWo;
        const int oy = fastdiv(rem, a.div_wo_m, a.div_wo_s);
        const int ox = rem - oy * a.Wo;
        const int cy = oy * a.stride, cx = ox * a.stride;  // centre tap (always inside the image)
        xoff[t] = 2u * (unsigned)(((b * a.H + cy) * a.W + cx) * (int)a.in_pitch + a.in_coff) + (((lane & 3) ^ ((lane >> 4) & 3))) * 16;
        const unsigned mask = valid ? tap_mask(cy, cx, a.H, a.W, a.ksize) : 0u;
        xmask[t] = mask;
    }
    constexpr unsigned OOB = 0xFFFFFFF0u;  // >= num_records: the buffer range check returns zeros
    const unsigned wvoff = lane * 16;
    const char* const wbase = (const char*)a.wpack + (int64_t)c0 * 64;
    // running k-block state (stage_load is called for consecutive steps 0,1,2,...)
    int l_kb = 0, l_cb = 0, l_kx = 0, l_tap = 0;
    int64_t l_xdelta = -2 * (int64_t)((a.pad * a.W + a.pad) * (int)a.in_pitch);  // tap (0,0) relative to the centre tap
    unsigned l_woff = 0;

    auto stage_load = [&](char* sbase) {
#pragma unroll
        for (int kbs = 0; kbs < KBS; ++kbs) {
            const bool kvalid = l_kb < a.nkb;
            const char* const xbase = (const char*)a.in + l_xdelta;
#pragma unroll
            for (int t = 0; t < XR; ++t) {
                const bool ok = kvalid && ((xmask[t] >> l_tap) & 1u);
                bload_lds16(xbase, ok ? xoff[t] : OOB, 0, sbase + kbs * XBYTES + (w + NW * t) * 1024);
            }
#pragma unroll
            for (int t = 0; t < WR; ++t) {
                const int rb = w + NW * t;  // wave-uniform
                if (rb < WRB) {
                    bload_lds16(wbase, kvalid ? wvoff + rb * 1024 : OOB, l_woff, sbase + KBS * XBYTES + kbs * WBYTES + rb * 1024);
                } else if (NST > 2) {
                    bload_lds16(wbase, OOB, 0, smem + NST * STAGE + w * 1024);  // keeps the vmcnt arithmetic wave-uniform
                }
            }
            // advance to the next k-block: channel block fastest, then kx, then ky
            ++l_kb;
            l_woff += (unsigned)a.cout_pad * 64u;
            l_xdelta += 64;
            if (++l_cb == a.cblocks) {
                l_cb = 0;
                ++l_tap;
                l_xdelta += 2 * (int64_t)a.in_pitch - 64 * (int64_t)a.cblocks;
                if (++l_kx == a.ksize) {
                    l_kx = 0;
                    l_xdelta += 2 * (int64_t)(a.W - a.ksize) * a.in_pitch;
                }
            }
        }
    };

    f32x16_t acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // fragment read offsets (bytes) inside a [rows][64 B] tile; slot = logical chunk ^ ((row>>2)&3)
    const int lrow = lane & 31;
    const int sw = (lane >> 2) & 3;
    const int foff0 = lrow * 64 + (((0 + (lane >> 5)) ^ sw) * 16);
    const int foff1 = lrow * 64 + (((2 + (lane >> 5)) ^ sw) * 16);

    // Fragment loads are software-pipelined by hand: the ds_reads of sub-step n+1 are issued before the MFMAs of
    // sub-step n (two named register sets), so the matrix pipe does not idle behind every LDS round trip.
    auto load_frags = [&](const char* sbase, int sub, bf16x8_t (&af)[TI], bf16x8_t (&bfr)[TJ]) {
        const int kbs = sub >> 1, h = sub & 1;
        const char* xt = sbase + kbs * XBYTES + (wp * WP) * 64;
        const char* wt = sbase + KBS * XBYTES + kbs * WBYTES + (wc * WC) * 64;
        const int fo = h ? foff1 : foff0;
#pragma unroll
        for (int i = 0; i < TI; ++i) af[i] = *(const bf16x8_t*)(wt + i * 2048 + fo);
#pragma unroll
        for (int j = 0; j < TJ; ++j) bfr[j] = *(const bf16x8_t*)(xt + j * 2048 + fo);
    };
    auto mma = [&](const bf16x8_t (&af)[TI], const bf16x8_t (&bfr)[TJ]) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    };
    auto stage_compute = [&](const char* sbase) {
        constexpr int NSUB = 2 * KBS;
        bf16x8_t a0[TI], b0[TJ], a1[TI], b1[TJ];
#ifdef VGH_SETPRIO_IGEMM
        __builtin_amdgcn_s_setprio(VGH_SETPRIO_IGEMM);
#endif
        load_frags(sbase, 0, a0, b0);
#pragma unroll
        for (int sub = 0; sub < NSUB; sub += 2) {
            load_frags(sbase, sub + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMAs (hipcc would sink it to its first use)
            mma(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (sub + 2 < NSUB) load_frags(sbase, sub + 2, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mma(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef VGH_SETPRIO_IGEMM
        __builtin_amdgcn_s_setprio(0);
#endif
    };

    // ---- main loop ------------------------------------------------------------------------------------
    const int nsteps = (a.nkb + KBS - 1) / KBS;
    if constexpr (NST == 2) {
        // 2-stage ring, __syncthreads(): loads of step s+1 fly under the MFMAs of step s, drained at every barrier
        stage_load(smem);
        __syncthreads();
        for (int s = 0; s < nsteps; ++s) {
            char* cur = smem + (s & 1) * STAGE;
            char* nxt = smem + ((s + 1) & 1) * STAGE;
            if (s + 1 < nsteps && !VGH_ABLATE(a, 1)) stage_load(nxt);
            if (!VGH_ABLATE(a, 2)) stage_compute(cur);
            __syncthreads();
        }
    } else {
        // NST-stage ring with COUNTED vmcnt and a raw s_barrier: NST-1 stages of LDS-DMA stay in flight across barriers
        // (a __syncthreads() would drain vmcnt(0) and serialise load latency with the MFMA phase).  One barrier per step:
        //   wait(own loads of step s) ; barrier (=> everybody's step-s tiles landed AND everybody finished reading step s-1)
        //   ; issue loads of step s+NST-1 into the buffer step s-1 used ; compute step s.
        constexpr int LPS = KBS * (XR + WR);  // LDS-DMA instructions per wave per stage (uniform across waves)
#pragma unroll
        for (int p = 0; p < NST - 1; ++p)
            if (p < nsteps) stage_load(smem + p * STAGE);
        int buf = 0;
        for (int s = 0; s < nsteps; ++s) {
            const int last_issued = (s + NST - 2 < nsteps - 1) ? s + NST - 2 : nsteps - 1;
            wait_stages<LPS>(last_issued - s);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            int nb = buf + NST - 1;
            if (nb >= NST) nb -= NST;
            if (s + NST - 1 < nsteps && !VGH_ABLATE(a, 1)) stage_load(smem + nb * STAGE);
            if (!VGH_ABLATE(a, 2)) stage_compute(smem + buf * STAGE);
            if (++buf == NST) buf = 0;
        }
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();  // all waves done reading the last stage before the epilogue reuses the LDS
        asm volatile("" ::: "memory");
    }

    // ---- epilogue ---------------------------------------------------------------------------------
    const float act_lo = act_bound(a.act);
    if (VGH_ABLATE(a, 8)) return;
    const int half4 = (lane >> 5) * 4;
    if constexpr (EPI == 1) {
        // Fast path (bf16 out, every channel offset a multiple of 8): each wave transposes its accumulators through a
        // private LDS strip [32 pixels][WC floats] so that results leave as 16-byte stores covering whole 128-B lines
        // per pixel (the MFMA layout would give 8-byte stores scattered over 64 lines per instruction), and the
        // residual arrives by 16-byte loads in the same pattern.  Single rounding: fp32 until the final convert.
        constexpr int EP = WC + 4;  // floats per staged pixel row (+16 B: conflict-free ds_write_b128)
        constexpr int CH = WC / 8;  // 16-byte output chunks per pixel
        constexpr int NIT = 32 * CH / 64;  // 16-byte items per lane per 32-pixel strip
        float* stg = (float*)smem + w * (32 * EP);
        // per-item geometry is the same for every strip j: item it -> (pixel px, chunk ch)
        int ipx[NIT], ich[NIT];
#pragma unroll
        for (int t = 0; t < NIT; ++t) {
            const int it = lane + 64 * t;
            ipx[t] = it / CH;
            ich[t] = it - ipx[t] * CH;
        }
        // residual tile of this wave: issue every 16-byte load up front so they are all in flight together
        bf16x8_t rres[TJ][NIT];
        if (a.res) {
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int t = 0; t < NIT; ++t) {
                    const int m = p0 + wp * WP + j * 32 + ipx[t];
                    const int c = c0 + wc * WC + ich[t] * 8;
                    const int oc = a.shuffle ? c % a.shuffle_c : c;
                    if (m < a.P && c < a.cout_store) rres[j][t] = *(const bf16x8_t*)(a.res + (int64_t)m * a.res_pitch + a.res_coff + oc);
                }
        }
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
#pragma unroll
            for (int i = 0; i < TI; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cl = i * 32 + q * 8 + half4;
                    const f32x4_t bv = *(const f32x4_t*)(a.bias + c0 + wc * WC + cl);
                    f32x4_t v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[i][j][q * 4 + e] + bv[e], act_lo);
                    *(f32x4_t*)(stg + lrow * EP + cl) = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < NIT; ++t) {
                const int px = ipx[t], ch = ich[t];
                const int m = p0 + wp * WP + j * 32 + px;
                const int c = c0 + wc * WC + ch * 8;
                if (m < a.P && c < a.cout_store) {
                    const f32x4_t v0 = *(const f32x4_t*)(stg + px * EP + ch * 8);
                    const f32x4_t v1 = *(const f32x4_t*)(stg + px * EP + ch * 8 + 4);
                    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    if (a.act == VGH_ACT_SILU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = silu_fn(v[e]);
                    }
                    int oc = c;
                    int64_t opix = m;
                    if (a.shuffle) {
                        const int sb = m / HoWo;
                        const int rem = m - sb * HoWo;
                        const int sy = rem / a.Wo, sx = rem - sy * a.Wo;
                        const int d = c / a.shuffle_c;
                        oc = c - d * a.shuffle_c;
                        opix = ((int64_t)sb * (2 * a.Ho) + 2 * sy + (d >> 1)) * (2 * a.Wo) + 2 * sx + (d & 1);
                    }
                    if (a.res) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += a.alpha * (float)rres[j][t][e];
                    }
                    const int ochan = (oc >= a.out_split) ? a.out_coff2 + (oc - a.out_split) : a.out_coff + oc;
                    if (a.out_f32) {
                        // fp32 prediction buffers with 16-byte aligned channel offsets: two float4 stores per item, scalar tail
                        // for the ragged last chunk (69 / 13 live channels)
                        float* op = (float*)a.out + opix * a.out_pitch + ochan;
                        if (c + 8 <= a.cout_store) {
                            *(f32x4_t*)op = f32x4_t{v[0], v[1], v[2], v[3]};
                            *(f32x4_t*)(op + 4) = f32x4_t{v[4], v[5], v[6], v[7]};
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                if (c + e < a.cout_store) op[e] = v[e];
                        }
                    } else {
                        bf16x8_t ov;
#pragma unroll
                        for (int e = 0; e < 8; ++e) ov[e] = (__bf16)v[e];
                        *(bf16x8_t*)((uint16_t*)a.out + opix * a.out_pitch + ochan) = ov;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    } else {
    // General path (fp32 prediction buffers, ragged channel counts, unaligned offsets): direct stores from the MFMA layout.
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int m = p0 + wp * WP + j * 32 + lrow;
        if (m >= a.P) continue;
        int64_t opix = m;
        int sb = 0, sy = 0, sx = 0;
        if (a.shuffle) {
            sb = m / HoWo;
            const int rem = m - sb * HoWo;
            sy = rem / a.Wo;
            sx = rem - sy * a.Wo;
        }
#pragma unroll
        for (int i = 0; i < TI; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = c0 + wc * WC + i * 32 + q * 8 + half4;  // first of 4 consecutive couts
                if (c >= a.cout_store) continue;
                const f32x4_t bv = *(const f32x4_t*)(a.bias + c);
                f32x4_t v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[i][j][q * 4 + e] + bv[e], act_lo);
                if (a.act == VGH_ACT_SILU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = silu_fn(v[e]);
                }
                int oc = c;
                if (a.shuffle) {
                    const int d = c / a.shuffle_c;
                    oc = c - d * a.shuffle_c;
                    opix = ((int64_t)sb * (2 * a.Ho) + 2 * sy + (d >> 1)) * (2 * a.Wo) + 2 * sx + (d & 1);
                }
                if (a.res) {
                    const bf16x4_t rv = *(const bf16x4_t*)(a.res + (int64_t)m * a.res_pitch + a.res_coff + oc);
                    const f32x4_t rf = __builtin_convertvector(rv, f32x4_t);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += a.alpha * rf[e];
                }
                const int ochan = (oc >= a.out_split) ? a.out_coff2 + (oc - a.out_split) : a.out_coff + oc;
                if (a.out_f32) {
                    float* op = (float*)a.out + opix * a.out_pitch + ochan;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c + e < a.cout_store) op[e] = v[e];
                } else {
                    bf16x4_t ov = __builtin_convertvector(v, bf16x4_t);
                    *(bf16x4_t*)((uint16_t*)a.out + opix * a.out_pitch + ochan) = ov;
                }
            }
        }
    }
    }
}

// =====================================================================================================
// Patch kernel for 3x3 / stride 1 / pad 1 convolutions: the input halo patch of one 32-channel block stays
// resident in LDS and is re-read by all 9 taps (the implicit-GEMM kernel above re-fetches every pixel 9x
// from L2); weights stream through LDS one kernel row (3 taps) at a time.  Per (channel-block, ky) step every
// wave issues 3 * TI * TJ * 2 MFMAs between barriers.
//   block tile : TH x TW output pixels of ONE image (raster order, 32-pixel MFMA groups) x BC couts
//   wave tile  : TJ = groups/NW pixel groups x ALL BC couts (TI = BC/32) -> full 2*BC-byte lines per pixel
//   LDS        : X[2][(TH+2)*(TW+2) halo pixels][64 B] + W[2][3 taps][BC][64 B], both 16-B-chunk swizzled
//   bytes/FLOP : (HP + 9*BC) * 64 B per 9*BP*BC*64 FLOP  ->  ~200 FLOP/B at 256 px x 128 couts (v1: 64-85)
// =====================================================================================================
// LDS layout of the halo patch, per tile width: row pitch (halo pixels) and the 16-byte-chunk swizzle
//     slot = chunk ^ (((hx >> SH) + ROW * hy) & 3)
// chosen (exhaustive search over pitches / shifts / row terms against the ds_read_b128 lane groups {0-3,12-15,20-27}, ...) so
// that the tap-shifted B-fragment reads of every 32-pixel MFMA group are bank-conflict free:
//     TW = 16: pitch 18, SH 1           TW = 32: pitch 34, SH 2           TW = 40: pitch 44 (2 padding pixels), SH 2, ROW 2
// (TW = 20 would need pitch 24 with ROW 1, whose kernel-row dependence is not an XOR: it keeps 2-way conflicts.)
template <int TW>
struct PatchLayout {
    static constexpr int HW = (TW == 40) ? 44 : TW + 2;
    static constexpr int SH = (TW == 16) ? 1 : 2;
    static constexpr int ROW = (TW == 40) ? 2 : 0;
};

template <int TW, int TH, int BC, int NWP, int NWC>
constexpr int patch_lds() {
    constexpr int HP = PatchLayout<TW>::HW * (TH + 2), HPU = (HP + 15) / 16;
    constexpr int loop = 2 * HPU * 1024 + 2 * 3 * BC * 64, epi = NWP * NWC * 32 * (BC / NWC + 4) * 4;
    return loop > epi ? loop : epi;
}

// waves per SIMD the register allocator must leave room for: the blocks that fit a CU by LDS (160 KiB) x waves per block over 4 SIMDs.
// Without it hipcc sizes registers for ONE block per CU (e.g. 184 VGPRs for the 5-wave tile: the second resident block is lost).
template <int TW, int TH, int BC, int NWP, int NWC>
constexpr int patch_wps() {
    constexpr int blocks = (160 * 1024) / patch_lds<TW, TH, BC, NWP, NWC>() < 1 ? 1 : (160 * 1024) / patch_lds<TW, TH, BC, NWP, NWC>();
    // a block's waves are dealt to the SIMDs cyclically from a varying start, so with a wave count that is not a multiple of 4
    // two resident blocks can stack their extra waves on the same SIMD: leave one more slot (5-wave tile at 144 VGPRs = 3 slots
    // per SIMD ran one block per CU and lost 40 %)
    constexpr int wps = (blocks * NWP * NWC + 3) / 4 + ((NWP * NWC) % 4 != 0 && blocks > 1 ? 1 : 0);
    // 128 accumulator registers (TI*TJ = 8) cannot share a SIMD four ways
    constexpr int acc = (BC / 32 / NWC) * (((TW * TH + 31) / 32) / NWP) * 16;
    constexpr int cap = acc >= 128 ? 2 : 4;
    return wps > cap ? cap : wps;
}

template <int TW, int TH, int BC, int NWP, int NWC>
__global__ __launch_bounds__(NWP * NWC * 64, (patch_wps<TW, TH, BC, NWP, NWC>())) void conv3x3_patch_kernel(const ConvArgs a, const int ntc, const int ntx, const int nty, const int total_tiles,
                                                                        const int chunk) {
    constexpr int NW = NWP * NWC;
    constexpr int NPX = TW * TH, NG = (NPX + 31) / 32, TJ = NG / NWP, TI = BC / 32 / NWC, WC = BC / NWC;
    constexpr int HW = PatchLayout<TW>::HW, HP = HW * (TH + 2), HPU = (HP + 15) / 16;
    constexpr int SWZ_SH = PatchLayout<TW>::SH, SWZ_ROW = PatchLayout<TW>::ROW;
    constexpr int XBYTES = HPU * 1024, WTAP = BC * 64, WSTEP = 3 * WTAP;
    constexpr int XUW = (HPU + NW - 1) / NW;  // halo units (16 pixels = 1 KiB) staged per wave per channel block
    constexpr int WU = 3 * BC / 16, WUW = (WU + NW - 1) / NW;
    static_assert(NG % NWP == 0 && (BC / 32) % NWC == 0, "tile must split evenly across the wave grid");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xbuf = smem;
    char* const wbuf = smem + 2 * XBYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = w % NWP, wc = w / NWP;
    const int lrow = lane & 31, hi = lane >> 5;
    constexpr unsigned OOB = 0xFFFFFFF0u;

    const unsigned wvoff = lane * 16;
    const unsigned wkstride = (unsigned)a.cout_pad * 64u;  // bytes between consecutive k-blocks of the packed weights
    int nx_mine = 0;  // halo-load instructions this wave issues per channel block (XUW or XUW-1: units are dealt round-robin)
#pragma unroll
    for (int t = 0; t < XUW; ++t) nx_mine += (HPU % NW == 0 || w + NW * t < HPU) ? 1 : 0;
    const float act_lo = act_bound(a.act);

    // ---- persistent block: XCD x owns the contiguous tile range [x*chunk, (x+1)*chunk) (3x3 halos of neighbouring tiles share
    //      one L2); its gridDim/8 resident blocks stride through it.  Between tiles nothing is relaunched, and the output stores
    //      of tile t drain from the memory pipeline while tile t+1 already loads and computes ----
    const int xcd = blockIdx.x & 7, gpx = gridDim.x >> 3;
    int trace_no = -1;
    for (int local = blockIdx.x >> 3; local < chunk; local += gpx) {
    int tile = xcd * chunk + local;
    if (tile >= total_tiles) break;
    ++trace_no;
    VGH_MARK(a, trace_no, 0);
    const int ctile = tile % ntc;
    tile /= ntc;
    const int txi = tile % ntx;
    tile /= ntx;
    const int tyi = tile % nty;
    const int b = tile / nty;
    const int y0 = tyi * TH, x0 = txi * TW, c0 = ctile * BC;
    // opaque per-tile copy of the lane id: everything per-lane below is derived from it, so hipcc re-materialises it per tile
    // instead of hoisting tile-invariant sub-expressions out of the tile loop and keeping ~40 extra VGPRs alive (measured:
    // 127 -> 184 VGPRs, i.e. the second resident block per CU was lost)
    int lane_t = lane;
    asm volatile("" : "+v"(lane_t));

    // ---- halo loader state: one 32-bit offset per staged 16-pixel unit, fixed over the whole K loop (the channel block only
    //      moves the descriptor base by 64 B); pixels outside the image / beyond the patch use an out-of-range offset = zeros
    unsigned xoff[XUW];
#pragma unroll
    for (int t = 0; t < XUW; ++t) {
        const int hp = (w + NW * t) * 16 + (lane_t >> 2);
        const int hy = hp / HW, hx = hp - hy * HW;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const bool ok = hp < HP && hx < TW + 2 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        // source-side swizzle: LDS slot (lane & 3) of halo pixel (hy, hx) holds channel chunk slot ^ swz(hy, hx)
        const int swz = ((hx >> SWZ_SH) + SWZ_ROW * hy) & 3;
        xoff[t] = ok ? 2u * (unsigned)(((b * a.H + iy) * a.W + ix) * (int)a.in_pitch + a.in_coff) + (((lane_t & 3) ^ swz)) * 16 : OOB;
    }
    const char* const wbase = (const char*)a.wpack + (int64_t)c0 * 64;

    auto load_x = [&](int cb, char* dst) {
        const char* const xbase = (const char*)a.in + cb * 64;
#pragma unroll
        for (int t = 0; t < XUW; ++t) {
            const int u = w + NW * t;
            if (HPU % NW == 0 || u < HPU) bload_lds16(xbase, xoff[t], 0, dst + u * 1024);
        }
    };
    auto load_w = [&](int cb, int ky, char* dst) {
#pragma unroll
        for (int t = 0; t < WUW; ++t) {
            const int v = w + NW * t;  // wave-uniform
            if (WU % NW == 0 || v < WU) {
                const int kx = v / (BC / 16), rb = v - kx * (BC / 16);
                const unsigned kb = (unsigned)((ky * 3 + kx) * a.cblocks + cb);
                bload_lds16(wbase, wvoff + rb * 1024, kb * wkstride, dst + kx * WTAP + rb * 1024);
            }
        }
    };

    // ---- fragment addressing: the swizzle depends on the halo COLUMN only, so the kernel row ky is a wave-uniform LDS offset and
    //      the per-lane byte offsets of all (kx, k16-half) combinations are computed once per tile.  They are derived from an
    //      opaque copy of the lane id so that hipcc re-materialises them per tile instead of keeping ~16 VGPRs alive across the
    //      epilogue (hoisted, they pushed the 16-wave tiles over the 128-VGPR budget into scratch) ----
    const int lrow_t = lane_t & 31, hi_t = lane_t >> 5;
    int boff[TJ][3][2];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int p = (wp * TJ + j) * 32 + lrow_t;
        const int ty = p / TW, tx = p - ty * TW;
        const int r00 = (p < NPX) ? ty * HW + tx : 0, hx0 = (p < NPX) ? tx : 0;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int h = 0; h < 2; ++h)  // swizzle of halo row ty (kernel row 0); rows ty+1 / ty+2 differ by ROW*ky, see load_frags
                boff[j][kx][h] = (r00 + kx) * 64 + (((2 * h + hi_t) ^ ((((hx0 + kx) >> SWZ_SH) + SWZ_ROW * (p < NPX ? ty : 0)) & 3)) * 16);
    }
    const int sw = (lane_t >> 2) & 3;
    const int aoff0 = (wc * WC + lrow_t) * 64 + (((0 + hi_t) ^ sw) * 16);
    const int aoff1 = (wc * WC + lrow_t) * 64 + (((2 + hi_t) ^ sw) * 16);

    f32x16_t acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    auto load_frags = [&](const char* X, const char* Wt, int ky, int sub, bf16x8_t (&af)[TI], bf16x8_t (&bfr)[TJ]) {
        const int kx = sub >> 1, h = sub & 1;
#pragma unroll
        for (int i = 0; i < TI; ++i) af[i] = *(const bf16x8_t*)(Wt + kx * WTAP + i * 2048 + (h ? aoff1 : aoff0));
        const char* Xk = X + ky * (HW * 64);
        // ROW = 2: the swizzle of halo row ty + ky is that of row ty plus 2*ky (mod 4) = bit 1 flipped for ky = 1 = byte offset ^ 32
        const int kyx = (SWZ_ROW == 2 && (ky & 1)) ? 32 : 0;
#pragma unroll
        for (int j = 0; j < TJ; ++j) bfr[j] = *(const bf16x8_t*)(Xk + (boff[j][kx][h] ^ kyx));
    };
    auto mma = [&](const bf16x8_t (&af)[TI], const bf16x8_t (&bfr)[TJ]) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    };
    // 6 sub-steps (3 taps x 2 k16 halves) per step, fragment loads one sub-step ahead of the MFMAs (two sub-steps ahead with a
    // third register set was measured: -15 % on the 5-wave tile, +6 % on p8x40x128 -> not kept)
    auto compute = [&](const char* X, const char* Wt, int ky) {
        bf16x8_t a0[TI], b0[TJ], a1[TI], b1[TJ];
#ifdef VGH_EXPERIMENTS
        if (VGH_ABLATE(a, 32)) {  // MFMAs on whatever the registers hold: the loop without its LDS fragment traffic
#pragma unroll
            for (int i = 0; i < TI; ++i) asm volatile("" : "=v"(a0[i]), "=v"(a1[i]));
#pragma unroll
            for (int j = 0; j < TJ; ++j) asm volatile("" : "=v"(b0[j]), "=v"(b1[j]));
#pragma unroll
            for (int sub = 0; sub < 6; sub += 2) {
                mma(a0, b0);
                mma(a1, b1);
            }
            return;
        }
#endif
#ifdef VGH_SETPRIO
        __builtin_amdgcn_s_setprio(VGH_SETPRIO);  // A/B knob (build.py -DVGH_SETPRIO=n): the wave inside its MFMA phase wins issue arbitration
#endif
        load_frags(X, Wt, ky, 0, a0, b0);
#pragma unroll
        for (int sub = 0; sub < 6; sub += 2) {
            load_frags(X, Wt, ky, sub + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMAs (hipcc would sink it to its first use)
            mma(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (sub + 2 < 6) load_frags(X, Wt, ky, sub + 2, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mma(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef VGH_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    };

    // ---- main loop over (channel block, kernel row) steps.  Weights of step s+1 (L2-resident, short latency) are issued at the
    //      start of step s; the halo patch of channel block cb+1 comes from HBM on first touch, so it is issued THREE steps ahead
    //      (at ky = 0 of block cb: its buffer was released by the barrier that ended block cb-1) instead of one ----
    const int nsteps = a.cblocks * 3;
    load_x(0, xbuf);
    load_w(0, 0, wbuf);
    __syncthreads();
    VGH_MARK(a, trace_no, 1);
    int cb = 0, ky = 0;
    for (int s = 0; s < nsteps; ++s) {
        int ncb = cb, nky = ky + 1;
        if (nky == 3) {
            nky = 0;
            ++ncb;
        }
        bool x_flying = false;
        if (!VGH_ABLATE(a, 1)) {
            if (s + 1 < nsteps) load_w(ncb, nky, wbuf + ((s + 1) & 1) * WSTEP);
            if (ky == 0 && cb + 1 < a.cblocks) {
                load_x(cb + 1, xbuf + ((cb + 1) & 1) * XBYTES);
                x_flying = true;
            }
        }
        if (!VGH_ABLATE(a, 2)) compute(xbuf + (cb & 1) * XBYTES, wbuf + (s & 1) * WSTEP, ky);
        // the weights of step s+1 must have landed; the (younger) halo loads may stay in flight across this barrier:
        // LDS-DMA loads retire in order, so "at most my own halo loads outstanding" == "my weight loads are done"
        if (x_flying) {
            if (nx_mine == XUW)
                wait_vmcnt<XUW>();
            else
                wait_vmcnt<(XUW > 0 ? XUW - 1 : 0)>();
        } else {
            wait_vmcnt<0>();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!VGH_ABLATE(a, 16)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        cb = ncb;
        ky = nky;
    }
    VGH_MARK(a, trace_no, 2);
    if (VGH_ABLATE(a, 8)) continue;

    // ---- epilogue: LDS transpose -> 16-byte stores (same scheme as the fast path of the implicit-GEMM kernel), one 32-pixel
    //      group at a time; the residual vectors of a group are loaded before its transpose and consumed after it ----
    constexpr int EP = WC + 4, CH = WC / 8, NIT = 32 * CH / 64;
    float* stg = (float*)smem + w * (32 * EP);
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int half4 = (lane_e >> 5) * 4, lrow_e = lane_e & 31;
    const int cw0 = c0 + wc * WC;
    const int pix00 = (b * a.H + y0) * a.W + x0;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        int opix[NIT];  // output pixel index (b*H + y)*W + x, or -1
        bf16x8_t rres[NIT];
#pragma unroll
        for (int t = 0; t < NIT; ++t) {
            const int it = lane_e + 64 * t;
            const int p = (wp * TJ + j) * 32 + it / CH;
            const int ty = p / TW, tx = p - ty * TW;
            const bool ok = p < NPX && y0 + ty < a.H && x0 + tx < a.W && (cw0 + (it % CH) * 8) < a.cout_store;
            opix[t] = ok ? pix00 + ty * a.W + tx : -1;
            if (a.res && ok) rres[t] = *(const bf16x8_t*)(a.res + (int64_t)opix[t] * a.res_pitch + a.res_coff + cw0 + (it % CH) * 8);
        }
#pragma unroll
        for (int i = 0; i < TI; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cl = i * 32 + q * 8 + half4;
                const f32x4_t bv = *(const f32x4_t*)(a.bias + cw0 + cl);
                f32x4_t v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[i][j][q * 4 + e] + bv[e], act_lo);
                *(f32x4_t*)(stg + lrow_e * EP + cl) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < NIT; ++t) {
            if (opix[t] >= 0) {
                const int it = lane_e + 64 * t;
                const int px = it / CH, ch = it - px * CH;
                const f32x4_t v0 = *(const f32x4_t*)(stg + px * EP + ch * 8);
                const f32x4_t v1 = *(const f32x4_t*)(stg + px * EP + ch * 8 + 4);
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                if (a.act == VGH_ACT_SILU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = silu_fn(v[e]);
                }
                if (a.res) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += a.alpha * (float)rres[t][e];
                }
                const int oc = cw0 + ch * 8;
                const int ochan = (oc >= a.out_split) ? a.out_coff2 + (oc - a.out_split) : a.out_coff + oc;
                bf16x8_t ov;
#pragma unroll
                for (int e = 0; e < 8; ++e) ov[e] = (__bf16)v[e];
                *(bf16x8_t*)((uint16_t*)a.out + (int64_t)opix[t] * a.out_pitch + ochan) = ov;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    // every wave is done with its staging strip before the next tile's LDS-DMA loads overwrite the region
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    VGH_MARK(a, trace_no, 3);
    }  // tile loop
}

// =====================================================================================================
// Patch kernel v3 ("q" tiles): the tile mathematics of conv3x3_patch_kernel, re-pipelined ACROSS the tiles of the
// persistent block.  Per-launch profiles of v2 (profiles/r01_pmc_patch.txt) showed three phases that add up
// instead of overlapping: ~35 us of MFMA work, ~10 us of tile prologues (address math + the first HBM round trip, every
// block of the chip at once) and ~14 us of epilogues (every block storing its tile at once).  v3:
//   * the first halo patch and weight stage of tile t+1 are issued during the LAST channel block of tile t, into the
//     ring slots tile t no longer needs, so a tile starts with its operands already in LDS;
//   * LDS is two regions [halo | weights]; tile t's last step leaves one region free for the epilogue staging while the
//     other region is being filled for tile t+1 (the region parity simply keeps running across tiles);
//   * output stores go through a buffer descriptor with an out-of-range offset for masked-off lanes, so every wave
//     issues EXACTLY NS stores per tile: the next tile waits `vmcnt(NS)` -- "my prefetched operands landed" -- and the
//     stores of tile t drain underneath the first steps of tile t+1 (vmcnt retires loads and stores in issue order);
//   * blocks in odd resident slots of a CU start `stagger` sleeps late, so that co-resident blocks (and the two halves
//     of the chip) are not in their store burst / operand-fetch phase at the same moment.
// =====================================================================================================
template <int TW, int TH, int BC, int NWP, int NWC>
struct Patch3 {
    static constexpr int NW = NWP * NWC;
    static constexpr int NPX = TW * TH, NG = (NPX + 31) / 32, TJ = NG / NWP, TI = BC / 32 / NWC, WC = BC / NWC;
    static constexpr int HW = PatchLayout<TW>::HW, HP = HW * (TH + 2), HPU = (HP + 15) / 16;
    static constexpr int XBYTES = HPU * 1024, WTAP = BC * 64, WSTEP = 3 * WTAP;
    static constexpr int R = XBYTES + WSTEP;          // one region: halo patch of a channel block + one kernel row of weights
    static constexpr int EP = WC + 4, CH = WC / 8;    // staged fp32 row (+16 B: conflict-free ds_write_b128), 16-byte output chunks per pixel
    static constexpr int STRIP = (NW * 32 * EP * 4 <= R) ? 32 : 16;  // pixels staged per pass: whole MFMA groups when they fit a region
    static constexpr int STG = NW * STRIP * EP * 4;
    static constexpr int RS = ((R > STG ? R : STG) + 1023) / 1024 * 1024;
    static constexpr int LDS = 2 * RS;
    static constexpr int NITS = STRIP * CH / 64;      // 16-byte items per lane per pass
    static constexpr int NS = TJ * (32 / STRIP) * NITS;  // output stores per wave per tile (exact: masked lanes store out of range)
    static_assert(NG % NWP == 0 && (BC / 32) % NWC == 0, "tile must split evenly across the wave grid");
    static_assert((STRIP * CH) % 64 == 0, "staging pass must be whole wave instructions");
    static constexpr int blocks_per_cu() { return (160 * 1024) / LDS < 1 ? 1 : (160 * 1024) / LDS; }
    static constexpr int wps() {
        constexpr int blocks = blocks_per_cu();
        constexpr int w0 = (blocks * NW + 3) / 4 + ((NW % 4 != 0 && blocks > 1) ? 1 : 0);
        constexpr int acc = TI * TJ * 16;
        constexpr int cap = acc >= 128 ? 2 : 4;
        return w0 > cap ? cap : w0;
    }
};

template <int TW, int TH, int BC, int NWP, int NWC>
__global__ __launch_bounds__(NWP * NWC * 64, (Patch3<TW, TH, BC, NWP, NWC>::wps())) void conv3x3_patch3_kernel(const ConvArgs a, const int ntc, const int ntx, const int nty,
                                                                                                               const int total_tiles, const int chunk) {
    using G = Patch3<TW, TH, BC, NWP, NWC>;
    constexpr int NW = G::NW, NPX = G::NPX, TJ = G::TJ, TI = G::TI, WC = G::WC;
    constexpr int HW = G::HW, HP = G::HP, HPU = G::HPU;
    constexpr int SWZ_SH = PatchLayout<TW>::SH, SWZ_ROW = PatchLayout<TW>::ROW;
    constexpr int XBYTES = G::XBYTES, WTAP = G::WTAP, RS = G::RS;
    constexpr int XUW = (HPU + NW - 1) / NW;
    constexpr int WU = 3 * BC / 16, WUW = (WU + NW - 1) / NW;
    constexpr int EP = G::EP, CH = G::CH, STRIP = G::STRIP, NITS = G::NITS, PASSES = 32 / G::STRIP;
    static_assert(XUW <= 8 && G::NS <= 16, "per-lane flag words hold 4 bits per halo unit / 2 bits per store");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = w % NWP, wc = w / NWP;
    constexpr unsigned OOB = 0xFFFFFFF0u;

    // ---- this block's CONTIGUOUS run of tiles inside its XCD's chunk (cout tile fastest, then x, y, image): consecutive tiles
    //      differ by scalar increments, so nothing per-lane is recomputed between tiles ----
    const int xcd = blockIdx.x & 7, gpx = gridDim.x >> 3, local = blockIdx.x >> 3;
    const int lo = (int)(((int64_t)local * chunk) / gpx), hi_ = (int)(((int64_t)(local + 1) * chunk) / gpx);
    int tile = xcd * chunk + lo;
    const int tile_end = (xcd * chunk + hi_ < total_tiles) ? xcd * chunk + hi_ : total_tiles;
    if (tile >= tile_end) return;
    // de-phase the co-resident blocks of a CU (block `local` and block `local + 32` of an XCD share a CU; speed only)
    if (a.stagger > 0 && ((local >> 5) & 1))
        for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(16);

    const unsigned wvoff = lane * 16;
    const unsigned wkstride = (unsigned)a.cout_pad * 64u;
    int nx_mine = 0;
#pragma unroll
    for (int t = 0; t < XUW; ++t) nx_mine += (HPU % NW == 0 || w + NW * t < HPU) ? 1 : 0;
    const float act_lo = act_bound(a.act);
    const int C = a.cblocks, nsteps = 3 * C;
    const int hr = a.H - (nty - 1) * TH, wr = a.W - (ntx - 1) * TW;  // rows / columns of the last (possibly ragged) tile row / column

    // ---- per-lane constants of the whole launch ----
    // halo loader: byte offset of each staged 16-pixel unit relative to the halo origin (y0-1, x0-1) of a tile (+ the source-side
    // swizzle), and 4 flag bits per unit: halo pixel lies in the {top row, rows past the last tile's ragged end, left column,
    // columns past the ragged end}; a tile at the matching image border masks those lanes (out-of-range offset = zeros)
    unsigned xrel[XUW], xflags = 0;
#pragma unroll
    for (int t = 0; t < XUW; ++t) {
        const int hp = (w + NW * t) * 16 + (lane >> 2);
        const int hy = hp / HW, hx = hp - hy * HW;
        const bool ok = hp < HP && hx < TW + 2;
        const int swz = ((hx >> SWZ_SH) + SWZ_ROW * hy) & 3;
        xrel[t] = ok ? 2u * (unsigned)((hy * a.W + hx) * (int)a.in_pitch) + (((lane & 3) ^ swz)) * 16 : OOB;
        xflags |= (unsigned)((hy == 0 ? 1 : 0) | (hy > hr ? 2 : 0) | (hx == 0 ? 4 : 0) | (hx > wr ? 8 : 0)) << (4 * t);
    }
    // fragment byte offsets (see conv3x3_patch_kernel): swizzle of the halo column only, kernel row = scalar LDS offset
    const int lrow = lane & 31, hi = lane >> 5;
    int boff[TJ][3][2];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int p = (wp * TJ + j) * 32 + lrow;
        const int ty = p / TW, tx = p - ty * TW;
        const int r00 = (p < NPX) ? ty * HW + tx : 0, hx0 = (p < NPX) ? tx : 0;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                boff[j][kx][h] = (r00 + kx) * 64 + (((2 * h + hi) ^ ((((hx0 + kx) >> SWZ_SH) + SWZ_ROW * (p < NPX ? ty : 0)) & 3)) * 16);
    }
    const int sw = (lane >> 2) & 3;
    const int aoff0 = (wc * WC + lrow) * 64 + (((0 + hi) ^ sw) * 16);
    const int aoff1 = (wc * WC + lrow) * 64 + (((2 + hi) ^ sw) * 16);
    // epilogue: byte offsets of this lane's 16-byte items relative to the tile origin (output and residual pitches), 2 flag bits per
    // item: pixel beyond the ragged last tile column / row
    unsigned orel[TJ][PASSES][NITS], rrel[TJ][PASSES][NITS], oflags = 0;
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int hp = 0; hp < PASSES; ++hp)
#pragma unroll
            for (int t = 0; t < NITS; ++t) {
                const int it = lane + 64 * t;
                const int pl = it / CH, ch = it - pl * CH;
                const int p = (wp * TJ + j) * 32 + hp * STRIP + pl;
                const int ty = p / TW, tx = p - ty * TW;
                orel[j][hp][t] = p < NPX ? 2u * (unsigned)((ty * a.W + tx) * (int)a.out_pitch + ch * 8) : OOB;
                rrel[j][hp][t] = p < NPX ? 2u * (unsigned)((ty * a.W + tx) * (int)a.res_pitch + ch * 8) : OOB;
                oflags |= (unsigned)((tx >= wr ? 1 : 0) | (ty >= hr ? 2 : 0)) << (2 * ((j * PASSES + hp) * NITS + t));
            }
    const int half4 = hi * 4;

    // ---- tile coordinates: decoded once, then advanced with scalar increments ----
    int ct, txi, tyi, b;
    {
        int q = tile;
        ct = q % ntc;
        q /= ntc;
        txi = q % ntx;
        q /= ntx;
        tyi = q % nty;
        b = q / nty;
    }
    auto advance = [&](int& ct_, int& tx_, int& ty_, int& b_) {
        if (++ct_ == ntc) {
            ct_ = 0;
            if (++tx_ == ntx) {
                tx_ = 0;
                if (++ty_ == nty) {
                    ty_ = 0;
                    ++b_;
                }
            }
        }
    };
    // halo offsets of a tile: the launch constants, masked where the tile touches an image border
    auto tile_xoff = [&](int tx_, int ty_, unsigned (&xo)[XUW]) {
        const unsigned m = (ty_ == 0 ? 1u : 0u) | (ty_ == nty - 1 ? 2u : 0u) | (tx_ == 0 ? 4u : 0u) | (tx_ == ntx - 1 ? 8u : 0u);
        const unsigned bad = xflags & (m * 0x11111111u);
#pragma unroll
        for (int t = 0; t < XUW; ++t) xo[t] = ((bad >> (4 * t)) & 15u) ? OOB : xrel[t];
    };
    auto halo_base = [&](int tx_, int ty_, int b_) {  // (y0 - 1, x0 - 1) of the tile; may lie before the tensor for masked lanes only
        return (const char*)a.in + 2 * (((int64_t)(b_ * a.H + ty_ * TH - 1) * a.W + (tx_ * TW - 1)) * a.in_pitch + a.in_coff);
    };
    auto load_x = [&](const char* hbase, const unsigned (&xo)[XUW], int cb, char* dst) {
        const char* const xbase = hbase + cb * 64;
#pragma unroll
        for (int t = 0; t < XUW; ++t) {
            const int u = w + NW * t;
            if (HPU % NW == 0 || u < HPU) bload_lds16(xbase, xo[t], 0, dst + u * 1024);
        }
    };
    auto load_w = [&](const char* wbase, int cb, int ky, char* dst) {
#pragma unroll
        for (int t = 0; t < WUW; ++t) {
            const int v = w + NW * t;  // wave-uniform
            if (WU % NW == 0 || v < WU) {
                const int kx = v / (BC / 16), rb = v - kx * (BC / 16);
                const unsigned kb = (unsigned)((ky * 3 + kx) * a.cblocks + cb);
                bload_lds16(wbase, wvoff + rb * 1024, kb * wkstride, dst + kx * WTAP + rb * 1024);
            }
        }
    };

    unsigned xoff[XUW];
    tile_xoff(txi, tyi, xoff);
    const char* hbase = halo_base(txi, tyi, b);
    const char* wbase = (const char*)a.wpack + (int64_t)(ct * BC) * 64;
    load_x(hbase, xoff, 0, smem);
    load_w(wbase, 0, 0, smem + XBYTES);
    int par = 0;  // region holding channel block 0 / step 0 of the current tile
    bool first = true;
    int trace_no = -1;

    while (true) {
        ++trace_no;
        VGH_MARK(a, trace_no, 0);
        const bool has_next = tile + 1 < tile_end;
        int nct = ct, ntxi = txi, ntyi = tyi, nb = b;
        advance(nct, ntxi, ntyi, nb);
        const char* const wbase_n = (const char*)a.wpack + (int64_t)(nct * BC) * 64;
        const char* const hbase_n = halo_base(ntxi, ntyi, nb);
        unsigned xoff_n[XUW];
        const int c0 = ct * BC;

        // the accumulators start at the bias (the C input of the first MFMA of each chain).  It arrives by SCALAR loads (the wave's
        // couts are wave-uniform; the two half-waves own alternate groups of 4): no VGPRs held across tiles and, above all, no vector
        // loads in the epilogue whose waits would drain the previous strip's stores (vmcnt is one in-order queue)
        f32x16_t acc[TI][TJ];
        {
            const AS4 f32x4_t* const bp = (const AS4 f32x4_t*)(a.bias + c0 + wc * WC);
            const bool upper = hi != 0;
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4_t lo4 = bp[i * 8 + q * 2], up4 = bp[i * 8 + q * 2 + 1];
                    asm volatile("" : "+s"(lo4), "+s"(up4));  // keep them scalar: hipcc would fold the select into a per-lane address
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float bv = upper ? up4[e] : lo4[e];
#pragma unroll
                        for (int j = 0; j < TJ; ++j) acc[i][j][q * 4 + e] = bv;
                    }
                }
        }

        auto load_frags = [&](const char* X, const char* Wt, int ky, int sub, bf16x8_t (&af)[TI], bf16x8_t (&bfr)[TJ]) {
            const int kx = sub >> 1, h = sub & 1;
#pragma unroll
            for (int i = 0; i < TI; ++i) af[i] = *(const bf16x8_t*)(Wt + kx * WTAP + i * 2048 + (h ? aoff1 : aoff0));
            const char* Xk = X + ky * (HW * 64);
            const int kyx = (SWZ_ROW == 2 && (ky & 1)) ? 32 : 0;
#pragma unroll
            for (int j = 0; j < TJ; ++j) bfr[j] = *(const bf16x8_t*)(Xk + (boff[j][kx][h] ^ kyx));
        };
        auto mma = [&](const bf16x8_t (&af)[TI], const bf16x8_t (&bfr)[TJ]) {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        };
        auto compute = [&](const char* X, const char* Wt, int ky) {
            bf16x8_t a0[TI], b0[TJ], a1[TI], b1[TJ];
#ifdef VGH_SETPRIO
            __builtin_amdgcn_s_setprio(VGH_SETPRIO);
#endif
            load_frags(X, Wt, ky, 0, a0, b0);
#pragma unroll
            for (int sub = 0; sub < 6; sub += 2) {
                load_frags(X, Wt, ky, sub + 1, a1, b1);
                __builtin_amdgcn_sched_barrier(0);
                mma(a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                if (sub + 2 < 6) load_frags(X, Wt, ky, sub + 2, a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                mma(a1, b1);
                __builtin_amdgcn_sched_barrier(0);
            }
#ifdef VGH_SETPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        };

        // ---- operands of step 0 landed (they are older than the NS stores of the previous tile's epilogue); every wave has
        //      finished that epilogue, so its staging region may be overwritten by this tile's loads ----
        if (first)
            wait_vmcnt<0>();
        else
            wait_vmcnt<G::NS>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        VGH_MARK(a, trace_no, 1);

        int cb = 0, ky = 0;
        for (int s = 0; s < nsteps; ++s) {
            int ncb = cb, nky = ky + 1;
            if (nky == 3) {
                nky = 0;
                ++ncb;
            }
            char* const wnext = smem + ((par + s + 1) & 1) * RS + XBYTES;
            bool x_flying = false;
            if (!VGH_ABLATE(a, 1)) {
                if (s + 1 < nsteps)
                    load_w(wbase, ncb, nky, wnext);
                else if (has_next)
                    load_w(wbase_n, 0, 0, wnext);  // (par + nsteps) & 1 == (par + C) & 1: the region tile t+1 starts in
                if (ky == 0) {
                    if (cb + 1 < C) {
                        load_x(hbase, xoff, cb + 1, smem + ((par + cb + 1) & 1) * RS);
                        x_flying = true;
                    } else if (has_next) {
                        tile_xoff(ntxi, ntyi, xoff_n);
                        load_x(hbase_n, xoff_n, 0, smem + ((par + C) & 1) * RS);
                        x_flying = true;
                    }
                }
            }
            if (!VGH_ABLATE(a, 2)) compute(smem + ((par + cb) & 1) * RS, smem + ((par + s) & 1) * RS + XBYTES, ky);
            if (s + 1 < nsteps) {  // the weights of step s+1 must have landed; younger halo loads may stay in flight
                if (x_flying) {
                    if (nx_mine == XUW)
                        wait_vmcnt<XUW>();
                    else
                        wait_vmcnt<(XUW > 0 ? XUW - 1 : 0)>();
                } else {
                    wait_vmcnt<0>();
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            cb = ncb;
            ky = nky;
        }

        VGH_MARK(a, trace_no, 2);
        // ---- epilogue: fp32 -> LDS strip -> 16-byte buffer stores; the strip lives in the region the last step just released.
        //      Straight-line code: offsets are launch constants, only the descriptor bases move with the tile; the residual tile
        //      arrives by buffer loads issued up front (out-of-range lanes read zeros), so hipcc's own vmcnt bookkeeping stays
        //      counted and no load wait ever drains the stores before it ----
        if (!VGH_ABLATE(a, 8)) {
            float* stg = (float*)(smem + ((par + C - 1) & 1) * RS) + w * (STRIP * EP);
            const int cw0 = c0 + wc * WC;
            const int ochan0 = (cw0 >= a.out_split) ? a.out_coff2 + (cw0 - a.out_split) : a.out_coff + cw0;
            const int64_t pix00 = (int64_t)(b * a.H + tyi * TH) * a.W + txi * TW;
            const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((char*)a.out + 2 * (pix00 * a.out_pitch + ochan0), 0, 0x80000000, 0x00020000);
            const unsigned em = (txi == ntx - 1 && wr < TW ? 1u : 0u) | (tyi == nty - 1 && hr < TH ? 2u : 0u);
            const unsigned ebad = oflags & (em * 0x55555555u);
            auto run = [&](auto res_tag) {
                constexpr bool HAS_RES = decltype(res_tag)::value;
                u32x4_t rres[HAS_RES ? TJ : 1][PASSES][NITS];
                if constexpr (HAS_RES) {
                    const __amdgpu_buffer_rsrc_t rrsrc =
                        __builtin_amdgcn_make_buffer_rsrc((char*)a.res + 2 * (pix00 * a.res_pitch + a.res_coff + cw0), 0, 0x80000000, 0x00020000);
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
#pragma unroll
                        for (int hp = 0; hp < PASSES; ++hp)
#pragma unroll
                            for (int t = 0; t < NITS; ++t) {
                                const bool bad = (ebad >> (2 * ((j * PASSES + hp) * NITS + t))) & 3u;
                                rres[j][hp][t] = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, bad ? OOB : rrel[j][hp][t], 0, 0);
                            }
                }
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
#pragma unroll
                    for (int hp = 0; hp < PASSES; ++hp) {
                        if (STRIP == 32 || (lrow >> 4) == hp) {
#pragma unroll
                            for (int i = 0; i < TI; ++i) {
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    f32x4_t v;  // the bias is already inside the accumulator (it was the MFMA chain's C input)
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[i][j][q * 4 + e], act_lo);
                                    *(f32x4_t*)(stg + (lrow & (STRIP - 1)) * EP + i * 32 + q * 8 + half4) = v;
                                }
                            }
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int t = 0; t < NITS; ++t) {
                            const int it = lane + 64 * t;
                            const int pl = it / CH, ch = it - pl * CH;
                            const f32x4_t v0 = *(const f32x4_t*)(stg + pl * EP + ch * 8);
                            const f32x4_t v1 = *(const f32x4_t*)(stg + pl * EP + ch * 8 + 4);
                            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                            if (a.act == VGH_ACT_SILU) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] = silu_fn(v[e]);
                            }
                            if constexpr (HAS_RES) {
                                const bf16x8_t rv = __builtin_bit_cast(bf16x8_t, rres[j][hp][t]);
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] += a.alpha * (float)rv[e];
                            }
                            bf16x8_t ov;
#pragma unroll
                            for (int e = 0; e < 8; ++e) ov[e] = (__bf16)v[e];
                            const bool bad = (ebad >> (2 * ((j * PASSES + hp) * NITS + t))) & 3u;
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ov), orsrc, bad ? OOB : orel[j][hp][t], 0, 0);
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            };
            if (a.res)
                run(std::true_type{});
            else
                run(std::false_type{});
        }
        VGH_MARK(a, trace_no, 3);
        if (!has_next) break;
        ++tile;
        ct = nct;
        txi = ntxi;
        tyi = ntyi;
        b = nb;
        wbase = wbase_n;
        hbase = hbase_n;
#pragma unroll
        for (int t = 0; t < XUW; ++t) xoff[t] = xoff_n[t];
        par = (par + C) & 1;
        first = false;
    }
}

struct CfgEntry {
    const char* name;
    int BP, BC, threads, lds;
    void (*launch)(const ConvArgs&, int, int, int, int, hipStream_t);
    int patch, TW, TH;  // patch != 0: conv3x3_patch_kernel (1) / conv3x3_patch3_kernel (2): 3x3, stride 1, fast epilogue only, tile TH x TW
    void (*launch_patch)(const ConvArgs&, int, int, int, int, int, int, hipStream_t);
};

// Per-device launch state: the >64 KiB dynamic-LDS opt-in and the occupancy query act on the CURRENT device, so they are cached
// per device id (a second engine on another GPU of the same process needs its own opt-in).  Racing threads compute the same value.
constexpr int kMaxDevices = 16;
static int current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) d = 0;
    return d;
}

template <typename K>
static int patch_blocks_per_cu(K kernel, int threads, int lds, std::atomic<int> (&cache)[kMaxDevices]) {
    const int dev = current_device();
    int n = cache[dev].load(std::memory_order_acquire);
    if (n == 0) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)kernel, threads, lds) != hipSuccess || n < 1) n = 1;
        cache[dev].store(n, std::memory_order_release);
    }
    return n;
}

static std::atomic<int> g_max_blocks_per_xcd{0};

static int persistent_blocks_per_xcd(const ConvArgs& a, int chunk, int per_cu) {
    // as many blocks per XCD as its 32 CUs keep resident, each looping over tiles; with the batch split over lane streams every
    // lane's kernel takes its share of the slots so that kernels of different lanes are co-resident (and out of phase)
    int slots = 32 * per_cu / (a.grid_share > 1 ? a.grid_share : 1);
    if (slots < 32) slots = 32;
    const int cap = g_max_blocks_per_xcd.load(std::memory_order_relaxed);
    if (cap > 0 && slots > cap) slots = cap;
#ifdef VGH_EXPERIMENTS
    static const int persist_env = getenv("VGH_PATCH_PERSIST") ? atoi(getenv("VGH_PATCH_PERSIST")) : 1;  // 0: one tile per block
    if (!persist_env) return chunk;
#endif
    return chunk < slots ? chunk : slots;
}

template <int TW, int TH, int BC, int NWP, int NWC>
void launch_patch_cfg(const ConvArgs& a, int ntc, int ntx, int nty, int total, int chunk, int lds, hipStream_t st) {
    static std::atomic<int> per_cu[kMaxDevices];
    const int n = patch_blocks_per_cu(conv3x3_patch_kernel<TW, TH, BC, NWP, NWC>, NWP * NWC * 64, lds, per_cu);
    const int gpx = persistent_blocks_per_xcd(a, chunk, n);
    hipLaunchKernelGGL((conv3x3_patch_kernel<TW, TH, BC, NWP, NWC>), dim3(gpx * 8), dim3(NWP * NWC * 64), lds, st, a, ntc, ntx, nty, total, chunk);
}

template <int TW, int TH, int BC, int NWP, int NWC>
void launch_patch3_cfg(const ConvArgs& a, int ntc, int ntx, int nty, int total, int chunk, int lds, hipStream_t st) {
    static std::atomic<int> per_cu[kMaxDevices];
    const int n = patch_blocks_per_cu(conv3x3_patch3_kernel<TW, TH, BC, NWP, NWC>, NWP * NWC * 64, lds, per_cu);
    const int gpx = persistent_blocks_per_xcd(a, chunk, n);
    hipLaunchKernelGGL((conv3x3_patch3_kernel<TW, TH, BC, NWP, NWC>), dim3(gpx * 8), dim3(NWP * NWC * 64), lds, st, a, ntc, ntx, nty, total, chunk);
}

template <int BP, int BC, int WP, int WC, int KBS, int NST>
void launch_cfg(const ConvArgs& a, int ntc, int total, int chunk, int lds, hipStream_t st) {
    static std::atomic<int> attr_done[kMaxDevices];
    if (lds > 64 * 1024) {
        const int dev = current_device();
        if (!attr_done[dev].load(std::memory_order_acquire)) {
            (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<BP, BC, WP, WC, KBS, 0, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<BP, BC, WP, WC, KBS, 1, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            attr_done[dev].store(1, std::memory_order_release);
        }
    }
    const dim3 grid(chunk * 8), block((BP / WP) * (BC / WC) * 64);
    if (NST == 2 && (a.nkb + KBS - 1) / KBS == 1) {  // the whole K fits one stage: no second buffer -> more blocks per CU
        const int one = KBS * (BP + BC) * 64, epi = (BP / WP) * (BC / WC) * 32 * (WC + 4) * 4;
        lds = one > epi ? one : epi;
    }
    if (a.fast_epi)
        hipLaunchKernelGGL((conv_igemm_kernel<BP, BC, WP, WC, KBS, 1, NST>), grid, block, lds, st, a, ntc, total, chunk);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<BP, BC, WP, WC, KBS, 0, NST>), grid, block, lds, st, a, ntc, total, chunk);
}

constexpr int lds_bytes(int BP, int BC, int WP, int WC, int KBS, int NST) {
    const int nw = (BP / WP) * (BC / WC);
    const int loop = NST * KBS * (BP + BC) * 64 + (NST > 2 ? nw * 1024 : 0), epi = nw * 32 * (WC + 4) * 4;
    return loop > epi ? loop : epi;
}
#define CFG(BP, BC, WP, WC, KBS) \
    { #BP "x" #BC "_w" #WP "x" #WC "_k" #KBS, BP, BC, (BP / WP) * (BC / WC) * 64, lds_bytes(BP, BC, WP, WC, KBS, 2), launch_cfg<BP, BC, WP, WC, KBS, 2>, 0, 0, 0, nullptr }
#define CFGR(BP, BC, WP, WC, KBS, NST) \
    { #BP "x" #BC "_w" #WP "x" #WC "_k" #KBS "_r" #NST, BP, BC, (BP / WP) * (BC / WC) * 64, lds_bytes(BP, BC, WP, WC, KBS, NST), launch_cfg<BP, BC, WP, WC, KBS, NST>, 0, 0, 0, nullptr }
#define PCFG(TW, TH, BC, NWP, NWC) \
    { "p" #TH "x" #TW "x" #BC "_n" #NWP "x" #NWC, (TW) * (TH), BC, (NWP) * (NWC) * 64, patch_lds<TW, TH, BC, NWP, NWC>(), nullptr, 1, TW, TH, launch_patch_cfg<TW, TH, BC, NWP, NWC> }

#define QCFG(TW, TH, BC, NWP, NWC) \
    { "q" #TH "x" #TW "x" #BC "_n" #NWP "x" #NWC, (TW) * (TH), BC, (NWP) * (NWC) * 64, Patch3<TW, TH, BC, NWP, NWC>::LDS, nullptr, 2, TW, TH, launch_patch3_cfg<TW, TH, BC, NWP, NWC> }

const CfgEntry g_cfgs[] = {
    CFG(128, 128, 64, 64, 1),  // 0
    CFG(256, 64, 64, 64, 1),   // 1
    CFG(256, 96, 64, 96, 1),   // 2
    CFG(128, 64, 32, 64, 1),   // 3
    CFG(256, 32, 64, 32, 1),   // 4
    CFG(64, 128, 32, 64, 1),   // 5
    CFG(64, 64, 32, 32, 1),    // 6
    CFG(128, 128, 64, 64, 2),  // 7
    CFG(256, 64, 64, 64, 2),   // 8
    CFG(256, 96, 64, 96, 2),   // 9
    CFG(128, 96, 32, 96, 1),   // 10
    CFG(128, 32, 32, 32, 1),   // 11
    CFG(64, 32, 32, 32, 1),    // 12 (2 waves)
    CFG(256, 128, 64, 128, 1), // 13
    CFG(128, 64, 32, 64, 2),   // 14
    PCFG(16, 16, 64, 4, 1),    // 15  256 px x 64: 4 waves x (64 px x 64)
    PCFG(16, 16, 128, 4, 1),   // 16  256 px x 128, 1 wave/SIMD
    PCFG(16, 16, 128, 4, 2),   // 17  256 px x 128: 8 waves x (64 px x 64)
    PCFG(16, 16, 96, 4, 1),    // 18
    PCFG(40, 8, 64, 5, 1),     // 19  320 px (full-width rows of a 40-wide map)
    PCFG(40, 8, 128, 5, 2),    // 20  10 waves x (64 px x 64)
    PCFG(40, 8, 96, 5, 1),     // 21
    PCFG(32, 8, 128, 4, 2),    // 22  8 rows x 32 cols: conflict-free fragment reads, 8 waves x (64 px x 64)
    PCFG(32, 8, 64, 4, 1),     // 23  4 waves x (64 px x 64)
    PCFG(16, 16, 32, 4, 1),    // 24
    CFGR(128, 128, 64, 64, 1, 3),  // 25  counted-vmcnt rings
    CFGR(128, 128, 64, 64, 1, 4),  // 26
    CFGR(128, 64, 32, 64, 1, 4),   // 27
    CFGR(256, 64, 64, 64, 1, 3),   // 28
    CFGR(128, 96, 32, 96, 1, 4),   // 29
    CFGR(64, 128, 32, 64, 1, 4),   // 30
    CFGR(64, 64, 32, 32, 1, 4),    // 31
    CFGR(128, 128, 64, 64, 2, 3),  // 32
    CFGR(128, 64, 32, 64, 2, 3),   // 33
    CFGR(128, 32, 32, 32, 1, 4),   // 34
    CFGR(256, 128, 64, 64, 1, 3),  // 35  8 waves
    CFGR(256, 128, 64, 64, 1, 2),  // 36  8 waves, 2-stage
    CFGR(256, 128, 64, 64, 2, 2),  // 37
    CFGR(256, 256, 64, 64, 1, 2),  // 38  16 waves
    CFGR(256, 256, 64, 64, 1, 3),  // 39
    CFGR(512, 128, 64, 64, 1, 2),  // 40  16 waves
    CFGR(256, 64, 32, 64, 1, 2),   // 41  8 waves, 32x64 wave tiles
    CFGR(256, 64, 32, 64, 1, 3),   // 42
    CFGR(256, 96, 32, 96, 1, 2),   // 43  8 waves
    CFGR(128, 128, 32, 64, 1, 2),  // 44  8 waves
    CFGR(128, 128, 32, 64, 1, 3),  // 45
    CFGR(512, 64, 64, 64, 1, 2),   // 46  8 waves
    CFGR(256, 128, 128, 64, 1, 2), // 47  4 waves, 128x64 wave tiles (8 MFMA tiles / wave)
    CFGR(256, 128, 128, 64, 1, 3), // 48
    CFGR(256, 128, 64, 128, 1, 2), // 49  4 waves, 64x128 wave tiles
    CFGR(256, 64, 128, 64, 1, 2),  // 50  2 waves
    CFGR(512, 128, 128, 64, 1, 2), // 51  8 waves, 128x64 wave tiles
    CFGR(256, 256, 128, 64, 1, 2), // 52  8 waves
    CFGR(256, 256, 64, 128, 1, 2), // 53  8 waves
    PCFG(32, 8, 256, 4, 4),    // 54  256 px x 256: 16 waves x (64 px x 64)
    PCFG(32, 16, 128, 8, 2),   // 55  512 px x 128: 16 waves
    PCFG(32, 8, 96, 4, 1),     // 56
    PCFG(32, 8, 64, 4, 2),     // 57  8 waves x (64 px x 32)
    PCFG(16, 16, 256, 4, 4),   // 58
    PCFG(20, 8, 128, 5, 2),    // 59  160 px (20-wide maps), 10 waves
    PCFG(32, 4, 128, 4, 2),    // 60  128 px x 128, 8 waves x (32 px x 64)
    CFG(128, 128, 64, 64, 3),  // 61  deep single-shot stages for short K (1x1 convs): K = 96 / 128 in ONE load round trip
    CFG(128, 128, 64, 64, 4),  // 62
    CFG(128, 64, 32, 64, 3),   // 63
    CFG(128, 64, 32, 64, 4),   // 64
    CFG(128, 96, 32, 96, 3),   // 65
    CFG(128, 96, 32, 96, 4),   // 66
    CFG(64, 64, 32, 32, 3),    // 67
    CFG(64, 64, 32, 32, 4),    // 68
    CFG(256, 64, 64, 64, 3),   // 69
    CFG(64, 128, 32, 64, 3),   // 70
    CFG(64, 128, 32, 64, 4),   // 71
    CFG(128, 32, 32, 32, 4),   // 72
    CFG(64, 96, 32, 96, 3),    // 73
    CFG(256, 128, 64, 64, 3),  // 74  8 waves
    PCFG(32, 16, 128, 4, 2),   // 75  512 px x 128: 8 waves x (128 px x 64): 6 fragment reads per 8 MFMAs
    PCFG(32, 16, 64, 8, 1),    // 76  512 px x 64: 8 waves x (64 px x 64)
    // cross-tile pipelined patch kernels (v3), same tile shapes as their "p" twins
    QCFG(16, 16, 64, 4, 1),    // 77
    QCFG(16, 16, 128, 4, 2),   // 78
    QCFG(16, 16, 96, 4, 1),    // 79
    QCFG(32, 8, 96, 4, 1),     // 80
    QCFG(32, 8, 64, 4, 1),     // 81
    QCFG(32, 8, 128, 4, 2),    // 82
    QCFG(40, 8, 64, 5, 1),     // 83
    QCFG(40, 8, 128, 5, 2),    // 84
    QCFG(16, 16, 256, 4, 4),   // 85
    QCFG(32, 8, 256, 4, 4),    // 86
    QCFG(16, 16, 32, 4, 1),    // 87
    QCFG(20, 8, 128, 5, 2),    // 88
    QCFG(32, 16, 128, 4, 2),   // 89  512 px x 128: 8 waves x (128 px x 64)
    QCFG(16, 16, 128, 4, 1),   // 90
    QCFG(32, 8, 64, 4, 2),     // 91
    QCFG(32, 4, 128, 4, 2),    // 92
    QCFG(40, 8, 96, 5, 1),     // 93
    PCFG(16, 8, 64, 4, 1),     // 94  128 px x 64: 48 KiB of LDS -> three blocks per CU (3 waves per SIMD), 1.5 KiB of fragments per MFMA
    QCFG(16, 8, 64, 4, 1),     // 95
    PCFG(16, 8, 128, 4, 2),    // 96  128 px x 128, 8 waves: 72 KiB -> two blocks = 4 waves per SIMD
    QCFG(16, 8, 128, 4, 2),    // 97
    PCFG(16, 8, 256, 4, 4),    // 98  128 px x 256, 16 waves
    // (measured and dropped: one-block 16-wave shapes p8x32x128_n8x2 / p16x16x128_n8x2 700 / 650 TFLOP/s where two 8-wave blocks reach 840-880;
    //  p8x16x96_n4x1 654 vs 781 for p8x32x96)
};
constexpr int kNumCfgs = sizeof(g_cfgs) / sizeof(g_cfgs[0]);

}  // namespace

#ifdef VGH_EXPERIMENTS
static unsigned long long* g_trace = nullptr;
extern "C" int vgh_conv_set_trace(void* dev_buffer) {
    g_trace = (unsigned long long*)dev_buffer;
    return VGH_OK;
}
#endif
int vgh_conv_num_cfgs() { return kNumCfgs; }
int vgh_conv_set_max_blocks_per_xcd(int blocks) {
    VGH_REQUIRE(blocks >= 0, "conv_set_max_blocks_per_xcd: negative");
    g_max_blocks_per_xcd.store(blocks, std::memory_order_relaxed);
    return VGH_OK;
}
int vgh_conv_cfg_ok(int cfg, int ksize, int stride, int cout_pad, int fast_epilogue, int shuffle) {
    if (cfg < 0 || cfg >= kNumCfgs) return 0;
    const CfgEntry& e = g_cfgs[cfg];
    if (cout_pad % e.BC) return 0;
    if (e.patch && !(ksize == 3 && stride == 1 && fast_epilogue && !shuffle)) return 0;
    return 1;
}
const char* vgh_conv_cfg_name(int cfg) { return (cfg >= 0 && cfg < kNumCfgs) ? g_cfgs[cfg].name : "?"; }

namespace {
// Tile choice for shapes without a measured entry in tuning/conv_cfg.json: a class table distilled from the per-op tuning
// reports (tools/gen_heuristic.py).  Class = (kernel size, stride, pixel-count bucket, largest of {128,96,64,32} dividing
// cout, cout > 128, K bucket, output width divisible by 40 / 32 / 16).
struct HeurRow {
    int ks, st, pb, nd, nb, kb, wo;
    const char* name;
};
const HeurRow g_heur[] = {
#include "conv_heur_table.inc"
};
constexpr int kNumHeur = sizeof(g_heur) / sizeof(g_heur[0]);

int cfg_by_name(const char* name) {
    for (int i = 0; i < kNumCfgs; ++i)
        if (strcmp(g_cfgs[i].name, name) == 0) return i;
    return -1;
}

int pick_size_only(const ConvArgs& a) {  // last resort: largest plain implicit-GEMM tile that still fills the chip
    const int cp = a.cout_pad;
    const int64_t P = a.P;
    auto tiles = [&](int cfg) { return ((P + g_cfgs[cfg].BP - 1) / g_cfgs[cfg].BP) * (int64_t)(cp / g_cfgs[cfg].BC); };
    int cand[4];
    int n = 0;
    if (cp % 128 == 0) cand[n++] = 0;
    if (cp % 96 == 0) cand[n++] = 2;
    if (cp % 64 == 0) cand[n++] = 1;
    cand[n++] = 4;
    for (int i = 0; i < n; ++i)
        if (tiles(cand[i]) >= 512) return cand[i];
    if (cp % 128 == 0 && tiles(5) >= 256) return 5;
    if (cp % 64 == 0) return 6;
    if (cp % 32 == 0) return (P >= 128 * 512) ? 4 : 11;
    return cand[0];
}
}  // namespace

int vgh_conv_pick_cfg(const ConvArgs& a) {
    // A measured per-layer table (tuning/*.json) overrides this through force_cfg.
    static int row_cfg[kNumHeur];
    static bool resolved = false;
    if (!resolved) {
        for (int i = 0; i < kNumHeur; ++i) row_cfg[i] = cfg_by_name(g_heur[i].name);
        resolved = true;
    }
    const int64_t P = a.P;
    const int N = a.cout_pad, K = a.nkb * 32;
    const int pb = P < 8192 ? 0 : P < 40000 ? 1 : P < 160000 ? 2 : P < 600000 ? 3 : 4;
    const int nd = N % 128 == 0 ? 128 : N % 96 == 0 ? 96 : N % 64 == 0 ? 64 : 32;
    const int nb = N > 128, kb = K <= 128 ? 0 : K <= 512 ? 1 : 2;
    const int wo = (a.ksize == 3 && a.stride == 1) ? ((a.Wo % 40 == 0 && a.Wo % 32 != 0) | ((a.Wo % 32 == 0) << 1) | ((a.Wo % 16 == 0) << 2)) : 0;
    int best = -1, best_d = 1 << 30;
    for (int i = 0; i < kNumHeur; ++i) {
        const HeurRow& r = g_heur[i];
        if (r.ks != a.ksize || r.st != a.stride || row_cfg[i] < 0) continue;
        const int d = 4 * abs(r.pb - pb) + 8 * (r.nd != nd) + 2 * (r.nb != nb) + abs(r.kb - kb) + 3 * (r.wo != wo);
        if (d < best_d && vgh_conv_cfg_ok(row_cfg[i], a.ksize, a.stride, a.cout_pad, a.fast_epi && !a.out_f32, a.shuffle)) {
            best_d = d;
            best = row_cfg[i];
        }
    }
    return best >= 0 ? best : pick_size_only(a);
}

int vgh_launch_conv(const ConvArgs& a, int force_cfg, hipStream_t stream) {
    VGH_REQUIRE(a.cin % 32 == 0 && a.cin > 0, "conv: cin=%d must be a positive multiple of 32", a.cin);
    VGH_REQUIRE(a.cout_pad % 32 == 0 && a.cout_pad > 0, "conv: cout_pad=%d must be a multiple of 32", a.cout_pad);
    VGH_REQUIRE(a.ksize == 1 || a.ksize == 3, "conv: ksize=%d unsupported", a.ksize);
    VGH_REQUIRE(a.in_coff % 8 == 0 && a.in_pitch % 8 == 0, "conv: input channel offset / pitch must keep 16-byte alignment");
    VGH_REQUIRE(a.out_f32 || (a.out_split % 4 == 0 && a.out_coff % 4 == 0 && a.out_coff2 % 4 == 0 && a.cout_store % 4 == 0 && a.out_pitch % 4 == 0),
                "conv: bf16 output needs 8-byte aligned channel offsets and cout_store %% 4 == 0");
    VGH_REQUIRE(!a.res || (a.res_coff % 4 == 0 && a.res_pitch % 4 == 0), "conv: residual alignment");
    VGH_REQUIRE((int64_t)a.B * a.H * a.W * a.in_pitch * 2 < (1ll << 31), "conv: input tensor must stay below 2 GiB (32-bit buffer offsets); run the batch in chunks");
    VGH_REQUIRE(!a.shuffle || (a.shuffle_c % 4 == 0 && a.cout_pad >= 4 * a.shuffle_c && a.ksize == 1 && a.stride == 1), "conv: bad shuffle");
    if (a.P == 0) return VGH_OK;
    VGH_REQUIRE((int64_t)a.B * a.Ho * a.Wo < (1ll << 30), "conv: too many output pixels for one launch");
#ifdef VGH_EXPERIMENTS
    static const int ablate = getenv("VGH_CONV_ABLATE") ? atoi(getenv("VGH_CONV_ABLATE")) : 0;
    static const int stagger_env = getenv("VGH_STAGGER") ? atoi(getenv("VGH_STAGGER")) : -1;
    static const int share_env = getenv("VGH_GRID_SHARE") ? atoi(getenv("VGH_GRID_SHARE")) : -1;
    const_cast<ConvArgs&>(a).trace = g_trace;
    if (stagger_env >= 0) const_cast<ConvArgs&>(a).stagger = stagger_env;
    if (share_env >= 1) const_cast<ConvArgs&>(a).grid_share = share_env;
#else
    constexpr int ablate = 0;
#endif
    const_cast<ConvArgs&>(a).ablate = ablate;
    vgh_fastdiv_magic((unsigned)(a.Ho * a.Wo), &const_cast<ConvArgs&>(a).div_howo_m, &const_cast<ConvArgs&>(a).div_howo_s);
    vgh_fastdiv_magic((unsigned)a.Wo, &const_cast<ConvArgs&>(a).div_wo_m, &const_cast<ConvArgs&>(a).div_wo_s);
    const bool al8 = a.out_coff % 8 == 0 && a.out_coff2 % 8 == 0 && a.out_split % 8 == 0 && a.cout_store % 8 == 0 && a.out_pitch % 8 == 0 &&
                     (!a.res || (a.res_coff % 8 == 0 && a.res_pitch % 8 == 0)) && (!a.shuffle || a.shuffle_c % 8 == 0);
    // fp32 outputs (prediction buffers) take the transposed epilogue too when every pixel row starts 16-byte aligned
    const bool al4f = a.out_f32 && a.out_coff % 4 == 0 && a.out_pitch % 4 == 0 && a.out_split >= a.cout_store && !a.res && !a.shuffle;
    const_cast<ConvArgs&>(a).fast_epi = (((!a.out_f32 && al8) || al4f) && !(ablate & 4)) ? 1 : 0;
    int cfg = force_cfg >= 0 ? force_cfg : vgh_conv_pick_cfg(a);
    VGH_REQUIRE(cfg < kNumCfgs, "conv: cfg %d out of range", cfg);
    if (!vgh_conv_cfg_ok(cfg, a.ksize, a.stride, a.cout_pad, a.fast_epi && !a.out_f32, a.shuffle)) {
        VGH_REQUIRE(force_cfg < 0, "conv: cfg %s cannot run this conv (cout_pad=%d k=%d s=%d)", g_cfgs[cfg].name, a.cout_pad, a.ksize, a.stride);
        cfg = 4;
    }
    if (g_cfgs[cfg].patch == 2 && !(a.cout_store == a.cout_pad && (a.out_split >= a.cout_pad || a.out_split % g_cfgs[cfg].BC == 0))) {
        // the pipelined kernel stores whole cout tiles into ONE output segment: ragged channel counts run on its "p" twin
        char twin[48];
        snprintf(twin, sizeof(twin), "p%s", g_cfgs[cfg].name + 1);
        const int t = cfg_by_name(twin);
        VGH_REQUIRE(t >= 0, "conv: cfg %s has no fallback tile", g_cfgs[cfg].name);
        cfg = t;
    }
    const CfgEntry& e = g_cfgs[cfg];
    if (e.patch) {
        const int ntc = a.cout_pad / e.BC, ntx = (a.W + e.TW - 1) / e.TW, nty = (a.H + e.TH - 1) / e.TH;
        const int64_t total = (int64_t)a.B * nty * ntx * ntc;
        VGH_REQUIRE(total < (1ll << 30), "conv: too many tiles");
        const int chunk = (int)((total + 7) / 8);
        e.launch_patch(a, ntc, ntx, nty, (int)total, chunk, e.lds, stream);
        VGH_HIP(hipGetLastError());
        return VGH_OK;
    }
    const int ntc = a.cout_pad / e.BC;
    const int64_t ntp = ((int64_t)a.P + e.BP - 1) / e.BP;
    const int64_t total = ntp * ntc;
    VGH_REQUIRE(total < (1ll << 30), "conv: too many tiles");
    const int chunk = (int)((total + 7) / 8);
    e.launch(a, ntc, (int)total, chunk, e.lds, stream);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}

void vgh_pack_conv_weights_host(const float* w, int cout_pad, int ksize, int cin, uint16_t* dst) {
    // dst[kb][cout][slot][8] with slot = chunk ^ ((cout>>2)&3), kb = (ky*ks + kx)*(cin/32) + cb
    const int cblocks = cin / 32, taps = ksize * ksize;
    for (int tap = 0; tap < taps; ++tap)
        for (int cb = 0; cb < cblocks; ++cb) {
            const int kb = tap * cblocks + cb;
            for (int co = 0; co < cout_pad; ++co) {
                const float* src = w + ((size_t)co * taps + tap) * cin + cb * 32;
                uint16_t* d = dst + ((size_t)kb * cout_pad + co) * 32;
                const int sw = (co >> 2) & 3;
                for (int chunk = 0; chunk < 4; ++chunk) {
                    const int slot = chunk ^ sw;
                    for (int e = 0; e < 8; ++e) d[slot * 8 + e] = vgh_f32_to_bf16_host(src[chunk * 8 + e]);
                }
            }
        }
}
