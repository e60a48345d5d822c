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

#include "vgh_internal.h"

#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

namespace {

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const AS1 void*)gsrc, (AS3 void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float act_fn(float v, int act) {
    if (act == VGH_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == VGH_ACT_SILU) return v / (1.0f + __expf(-v));
    return v;
}

template <int BP, int BC, int WP, int WC, int KBS, int EPI>
__global__ __launch_bounds__((BP / WP) * (BC / WC) * 64, ((WP / 32) * (WC / 32) <= 4 ? 3 : 1)) void conv_igemm_kernel(const ConvArgs a, const int ntc, const int total_tiles,
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

    // ---- per-lane gather state for the activation rows this lane stages -------------------------
    const int HoWo = a.Ho * a.Wo;
    const char* xbase[XR];
    int xiy[XR], xix[XR];
#pragma unroll
    for (int t = 0; t < XR; ++t) {
        const int r = (w + NW * t) * 16 + (lane >> 2);
        const int m = p0 + r;
        const bool valid = m < a.P;
        const int mm = valid ? m : 0;
        const int b = mm / HoWo;
        const int rem = mm - b * HoWo;
        const int oy = rem / a.Wo;
        const int ox = rem - oy * a.Wo;
        const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
        xbase[t] = (const char*)a.in + 2 * ((((int64_t)b * a.H + iy0) * a.W + ix0) * a.in_pitch + a.in_coff);
        xiy[t] = valid ? iy0 : -(1 << 20);
        xix[t] = ix0;
    }
    const int csel16 = (((lane & 3) ^ ((lane >> 4) & 3))) * 16;  // logical 16-B chunk this lane fetches (source swizzle)
    const char* zsrc = (const char*)a.zeros + csel16;
    const char* wlane = (const char*)a.wpack + (int64_t)c0 * 64 + lane * 16;  // pre-swizzled image: linear copy

    auto stage_load = [&](int step, char* sbase) {
#pragma unroll
        for (int kbs = 0; kbs < KBS; ++kbs) {
            const int kb = step * KBS + kbs;  // wave-uniform
            const bool kvalid = kb < a.nkb;
            const int tap = kb / a.cblocks;
            const int cb = kb - tap * a.cblocks;
            const int ky = tap / a.ksize;
            const int kx = tap - ky * a.ksize;
            const int delta = 2 * ((ky * a.W + kx) * (int)a.in_pitch + cb * 32) + csel16;
#pragma unroll
            for (int t = 0; t < XR; ++t) {
                const bool ok = kvalid && (unsigned)(xiy[t] + ky) < (unsigned)a.H && (unsigned)(xix[t] + kx) < (unsigned)a.W;
                const char* src = ok ? xbase[t] + delta : zsrc;
                glds16(src, sbase + kbs * XBYTES + (w + NW * t) * 1024);
            }
            const char* wsrc = wlane + (int64_t)kb * a.cout_pad * 64;
#pragma unroll
            for (int t = 0; t < WR; ++t) {
                const int rb = w + NW * t;  // wave-uniform
                if (rb < WRB) {
                    const char* src = kvalid ? wsrc + rb * 1024 : (const char*)a.zeros + (lane & 3) * 16;
                    glds16(src, sbase + KBS * XBYTES + kbs * WBYTES + rb * 1024);
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

    auto stage_compute = [&](const char* sbase) {
#pragma unroll
        for (int kbs = 0; kbs < KBS; ++kbs) {
            const char* xt = sbase + kbs * XBYTES + (wp * WP) * 64;
            const char* wt = sbase + KBS * XBYTES + kbs * WBYTES + (wc * WC) * 64;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int fo = h ? foff1 : foff0;
                bf16x8_t af[TI], bfr[TJ];
#pragma unroll
                for (int i = 0; i < TI; ++i) af[i] = *(const bf16x8_t*)(wt + i * 2048 + fo);
#pragma unroll
                for (int j = 0; j < TJ; ++j) bfr[j] = *(const bf16x8_t*)(xt + j * 2048 + fo);
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
    };

    // ---- main loop: 2-stage LDS ring, loads of step s+1 in flight under the MFMAs of step s -------
    const int nsteps = (a.nkb + KBS - 1) / KBS;
    stage_load(0, smem);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        char* cur = smem + (s & 1) * STAGE;
        char* nxt = smem + ((s + 1) & 1) * STAGE;
        if (s + 1 < nsteps && !(a.ablate & 1)) stage_load(s + 1, nxt);
        if (!(a.ablate & 2)) stage_compute(cur);
        __syncthreads();
    }

    // ---- epilogue ---------------------------------------------------------------------------------
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
                    for (int e = 0; e < 4; ++e) v[e] = act_fn(acc[i][j][q * 4 + e] + bv[e], a.act);
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
                    bf16x8_t ov;
#pragma unroll
                    for (int e = 0; e < 8; ++e) ov[e] = (__bf16)v[e];
                    *(bf16x8_t*)((uint16_t*)a.out + opix * a.out_pitch + ochan) = ov;
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
                for (int e = 0; e < 4; ++e) v[e] = act_fn(acc[i][j][q * 4 + e] + bv[e], a.act);
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

struct CfgEntry {
    const char* name;
    int BP, BC, threads, lds;
    void (*launch)(const ConvArgs&, int, int, int, int, hipStream_t);
};

template <int BP, int BC, int WP, int WC, int KBS>
void launch_cfg(const ConvArgs& a, int ntc, int total, int chunk, int lds, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done && lds > 64 * 1024) {
        (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<BP, BC, WP, WC, KBS, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<BP, BC, WP, WC, KBS, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    const dim3 grid(chunk * 8), block((BP / WP) * (BC / WC) * 64);
    if (a.fast_epi)
        hipLaunchKernelGGL((conv_igemm_kernel<BP, BC, WP, WC, KBS, 1>), grid, block, lds, st, a, ntc, total, chunk);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<BP, BC, WP, WC, KBS, 0>), grid, block, lds, st, a, ntc, total, chunk);
}

constexpr int lds_bytes(int BP, int BC, int WP, int WC, int KBS) {
    const int loop = 2 * KBS * (BP + BC) * 64, epi = (BP / WP) * (BC / WC) * 32 * (WC + 4) * 4;
    return loop > epi ? loop : epi;
}
#define CFG(BP, BC, WP, WC, KBS) \
    { #BP "x" #BC "_w" #WP "x" #WC "_k" #KBS, BP, BC, (BP / WP) * (BC / WC) * 64, lds_bytes(BP, BC, WP, WC, KBS), launch_cfg<BP, BC, WP, WC, KBS> }

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
};
constexpr int kNumCfgs = sizeof(g_cfgs) / sizeof(g_cfgs[0]);

}  // namespace

int vgh_conv_num_cfgs() { return kNumCfgs; }
const char* vgh_conv_cfg_name(int cfg) { return (cfg >= 0 && cfg < kNumCfgs) ? g_cfgs[cfg].name : "?"; }

int vgh_conv_pick_cfg(const ConvArgs& a) {
    // Heuristic fallback; a measured per-layer table (tuning/*.json) overrides it through force_cfg.
    const int cp = a.cout_pad;
    const int64_t P = a.P;
    auto tiles = [&](int cfg) { return ((P + g_cfgs[cfg].BP - 1) / g_cfgs[cfg].BP) * (int64_t)(cp / g_cfgs[cfg].BC); };
    int cand[4];
    int n = 0;
    if (cp % 128 == 0) cand[n++] = 0;
    if (cp % 96 == 0) cand[n++] = 2;
    if (cp % 64 == 0) cand[n++] = 1;
    cand[n++] = 4;
    int best = cand[0];
    // prefer the largest tile that still fills the chip (>= 2 blocks per CU), else fall to smaller tiles
    for (int i = 0; i < n; ++i)
        if (tiles(cand[i]) >= 512) return cand[i];
    if (cp % 128 == 0 && tiles(5) >= 256) return 5;
    if (cp % 64 == 0) return 6;
    if (cp % 32 == 0) return (P >= 128 * 512) ? 4 : 11;
    return best;
}

int vgh_launch_conv(const ConvArgs& a, int force_cfg, hipStream_t stream) {
    VGH_REQUIRE(a.cin % 32 == 0 && a.cin > 0, "conv: cin=%d must be a positive multiple of 32", a.cin);
    VGH_REQUIRE(a.cout_pad % 32 == 0 && a.cout_pad > 0, "conv: cout_pad=%d must be a multiple of 32", a.cout_pad);
    VGH_REQUIRE(a.ksize == 1 || a.ksize == 3, "conv: ksize=%d unsupported", a.ksize);
    VGH_REQUIRE(a.in_coff % 8 == 0 && a.in_pitch % 8 == 0, "conv: input channel offset / pitch must keep 16-byte alignment");
    VGH_REQUIRE(a.out_f32 || (a.out_split % 4 == 0 && a.out_coff % 4 == 0 && a.out_coff2 % 4 == 0 && a.cout_store % 4 == 0 && a.out_pitch % 4 == 0),
                "conv: bf16 output needs 8-byte aligned channel offsets and cout_store %% 4 == 0");
    VGH_REQUIRE(!a.res || (a.res_coff % 4 == 0 && a.res_pitch % 4 == 0), "conv: residual alignment");
    VGH_REQUIRE(!a.shuffle || (a.shuffle_c % 4 == 0 && a.cout_pad >= 4 * a.shuffle_c && a.ksize == 1 && a.stride == 1), "conv: bad shuffle");
    if (a.P == 0) return VGH_OK;
    static const int ablate = getenv("VGH_CONV_ABLATE") ? atoi(getenv("VGH_CONV_ABLATE")) : 0;  // perf experiments only
    const_cast<ConvArgs&>(a).ablate = ablate;
    const bool al8 = a.out_coff % 8 == 0 && a.out_coff2 % 8 == 0 && a.out_split % 8 == 0 && a.cout_store % 8 == 0 && a.out_pitch % 8 == 0 &&
                     (!a.res || (a.res_coff % 8 == 0 && a.res_pitch % 8 == 0)) && (!a.shuffle || a.shuffle_c % 8 == 0);
    const_cast<ConvArgs&>(a).fast_epi = (!a.out_f32 && al8 && !(ablate & 4)) ? 1 : 0;
    int cfg = force_cfg >= 0 ? force_cfg : vgh_conv_pick_cfg(a);
    VGH_REQUIRE(cfg < kNumCfgs, "conv: cfg %d out of range", cfg);
    if (a.cout_pad % g_cfgs[cfg].BC != 0) {
        VGH_REQUIRE(force_cfg < 0, "conv: cfg %s does not divide cout_pad=%d", g_cfgs[cfg].name, a.cout_pad);
        cfg = 4;
    }
    const CfgEntry& e = g_cfgs[cfg];
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
