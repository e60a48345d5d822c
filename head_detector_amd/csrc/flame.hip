// FLAME decode for gfx950: blendshapes + pose correctives + linear-blend skinning + rigid reprojection.
//
// Replaces FLAMELayer.forward (head_detector/flame.py:122-169) -> smplx.lbs.lbs (third-party, call site
// flame.py:152-161) -> reproject_spatial_vertices (flame.py:179-208) -> rot_mat_from_6dof
// (head_detector/utils.py:120-128) -> vertex un-pad/un-scale of HeadDetector._parse_predictions
// (head_detector/detector.py:67-69).
//
// MI355X-first restructure (not a translation of the einsum chain):
//  * the joint regressor is folded into the shape basis at create time (fp64 on the host):
//      J = J_regressor (v_template + S beta) = J0 + (J_regressor S) beta,   JS is [3*NJ x NB]
//    which removes the global "all vertices -> joints" dependency, so one tiny per-head prologue kernel
//    produces everything the vertex kernel needs (5 skinning transforms, 36 pose features, R, s, t);
//  * basis / template / skinning weights are stored as SoA planes [k][xyz][Vp] so one lane owns 4 consecutive
//    vertices and streams the 26 MB basis with 16-byte loads (it stays resident in the 256 MB Infinity Cache);
//  * a block owns (256 x 4 vertices) x HT heads: every basis value fetched feeds HT FMAs (coefficients
//    broadcast from LDS), the skinning / rigid / un-pad epilogue runs in registers and results leave with
//    16-byte stores.  Exact fp32 FMA chains in fixed k order.
//  * (r02 - r04) the blend is a [heads x K] . [K x 3V] contraction, and gfx950's fp32 matrix instructions are exact ascending fmaf chains over their k
//    (tools/micro/mfma_f32_chain.hip): the matrix-core kernels below -- register-fed, LDS-staged 128-head tiles, and from r04 the component-split "c3" tiles
//    that run 1 ... 2 048 heads (one coordinate plane per wave, coefficient tile in LDS, k-interleaved basis copy, prologue waves inside the block up to 8 heads)
//    -- produce the VALU kernel's bits, so which one runs is a pure speed choice (run_decode_on; vgh_flame_set_matrix_path for tests and A/B).
#include <math.h>
#include <stdlib.h>

#include <atomic>
#include <vector>

#define AS3 __attribute__((address_space(3)))

#include "vgh_internal.h"

typedef __attribute__((ext_vector_type(3))) float f32x3_t;

namespace {

constexpr int MAXJ = 8;
constexpr int HP_A = 0;                      // A[j][12] (3x4 row major) : MAXJ*12 floats
constexpr int HP_R = MAXJ * 12;              // R[9]
constexpr int HP_S = HP_R + 9;               // clamp(scale, 1e-8)
constexpr int HP_T = HP_S + 1;               // translation[3]
constexpr int HP_U = HP_T + 3;               // unpad (pad_x, pad_y, scale_factor)
constexpr int HP_SIZE = 128;                 // floats per head

}  // namespace

struct vgh_flame {
    int device, V, Vp, NB, NJ, NP, K, Kp;  // NP = 9*(NJ-1); K = NB + NP; Kp = K rounded up to 8
    int max_heads;
    int ncu;  // compute units of the device (rounds of blocks: the c3 block size choice)
    float* basis;    // [K][3][Vp]
    float* basis8;   // [Kp / 8][3][2][Vp][4]: basis8[((g * 3 + c) * 2 + (k & 1)) * Vp + v][(k >> 1) & 3] = basis[k = 8g + ..][c][v]; rows K .. Kp - 1 zero
    float* vt;       // [3][Vp]
    float* wts;      // [NJ][Vp]
    float* J0;       // [3*NJ]
    float* JS;       // [NB][MAXJ*3] (coefficient-major, zero columns beyond 3*NJ: the prologue reads a row as six 16-byte loads)
    int32_t* parents;  // [NJ]
    float* coef;     // scratch [Kp][npad] (TRANSPOSED: heads contiguous; npad = max_heads rounded up to 128; columns beyond the heads of the last
                     // decode hold zeros from creation or stale coefficients of an earlier, larger decode: harmless, MFMA rows are independent and
                     // the stores are masked by the head count)
    int npad;
    float* headpack; // scratch [npad][HP_SIZE]
    // the scratch is per handle: a decode on another stream than the previous one is ordered after it (event on the previous stream)
    hipStream_t last_stream = nullptr;
    hipEvent_t ev_scratch = nullptr;
    bool used = false;
};

// vgh_flame_set_matrix_path: 0 VALU kernels only, 1 automatic (default), 2 always the register-fed matrix-core kernel, 3 / 4 always the
// LDS-staged one (3: 128-head blocks, 4: 64-head blocks, 5: 32-head blocks).  All vertex kernels are bit-identical; tests and tools/flame_sweep.py switch between them.
static std::atomic<int> g_flame_mode{1};
#ifdef VGH_EXPERIMENTS
static unsigned long long* g_prep_trace = nullptr;  // vgh_flame_set_trace
#endif
constexpr int kLdsMinHeads = 1024;  // from here on: LDS-staged tiles with 128-head blocks
constexpr int kC3MaxHeads = 2048;   // ... component-split tiles up to here (r04; automatic mode, direct batches)
constexpr int kLdsMidHeads = 256;   // ... 64-head blocks (r02: the register-fed kernel ran n = 256 .. 1024 at 0.22 - 0.35 of the fp32 roof)

namespace {

struct PrepArgs {
    const float* params;  // [n,413] or null
    const float* betas;   // [n,NB] or null
    const float* pose;    // [n,3*NJ] or null
    const float* unpad;   // [n,3] or null
    const float* J0;
    const float* JS;
    const int32_t* parents;
    float* coef;        // [Kp][npad]
    float* headpack;
    int npad;
    float* rot_out;     // [n,9] or null
    float* joints_out;  // [n,NJ,3] or null
    int NB, NJ, Kp;
    // indirect (fused detector) mode: head h reads params row head_row[h] and unpad row head_image[h];
    // *n_dev is the live head count (blocks beyond it exit) -- no host round trip for the data-dependent n
    const int32_t* head_row;
    const int32_t* head_image;
    const int32_t* n_dev;
    float* rpy_out;  // [n,3] (roll, pitch, yaw) degrees or null: calculate_rpy, utils.py:146-151
    int n;           // heads of this launch (capacity when n_dev is set)
    int live0_end, live1_begin, live1_end;  // betas outside [0, live0_end) u [live1_begin, live1_end) are exact zeros (skipped: fma(w, 0, s) = s)
    unsigned long long* trace;  // -DVGH_EXPERIMENTS only: wall-clock (100 MHz) phase stamps of head 0's prologue
};

#ifdef VGH_EXPERIMENTS
#define PMARK(a, h, lane, k) do { if ((a).trace && (h) == 0 && (lane) == 0) (a).trace[k] = wall_clock64(); } while (0)
#else
#define PMARK(a, h, lane, k) do { } while (0)
#endif

// Per-wave LDS scratch of the per-head prologue
struct PrepScratch {
    float J[MAXJ * 3];
    float R[MAXJ * 9];
    float pose[MAXJ * 3];
    float tail[16];  // params[400..412]: jaw 3 | rot6 | trans 3 | scale 1
    int par[MAXJ];
    float unpad[4];
    float Rg[MAXJ * 9], tg[MAXJ * 3];  // global rotations / translations of the kinematic chain
};

// Everything the vertex kernel needs for head h, computed by ONE wave: blend coefficients [betas | pose features] -> coef[k * cstride],
// the 5 skinning transforms + 6D rotation + clamp(scale) + translation + un-pad triple -> hp[HP_SIZE] (LDS).  `emit`: also write
// the caller-visible per-head outputs (rotation matrix, joints, roll/pitch/yaw).  The arithmetic is the same wherever it runs
// (stand-alone prologue kernel for large batches, or the vertex kernel's own prologue for small ones).
// `betas_out` false: the raw betas are not copied to coef (the caller reads them in place); `pose_row0` >= 0: the pose features go to rows pose_row0 .. of coef
// instead of rows NB .. (a compact coefficient tile).
// WIDE (the one-wave-per-block prologue kernel, which has the registers): the joint regression's loads of 7 x 64 coefficients are issued as ONE batch.
template <bool WIDE = false>
__device__ __forceinline__ void prep_head(const PrepArgs& a, int h, int lane, PrepScratch& S, float* coef, int cstride, float* hp, bool emit, const bool betas_out = true,
                                          const int pose_row0 = -1) {
// every rounding is spelled out (explicit fmaf where a fused multiply-add is meant): the function is inlined into several kernels and
// must not be contracted differently from one to the next, or a head's vertices would depend on the batch it is decoded in
#pragma clang fp contract(off)
    const int NB = a.NB, NJ = a.NJ;
    PMARK(a, h, lane, 0);
    for (int e = lane; e < HP_SIZE; e += 64) hp[e] = 0.0f;
    const int64_t prow = a.head_row ? a.head_row[h] : h;
    const int64_t urow = a.head_image ? a.head_image[h] : h;
    const float* p = a.params ? a.params + prow * VGH_NUM_FLAME_PARAMS : nullptr;
    // betas = [shape(300) | expression(100)]  (flame.py:132-140; FLAME_CONSTS widths make the padding empty) -> blend coefficients, and
    // joints J = J0 + JS beta in the same pass: a lane owns the coefficients l = lane, lane + 64, ..., per output a lane-strided fmaf
    // chain in ascending l (the xor butterfly follows).  Coefficients outside the live ranges are exact zeros: fma(w, 0, s) = s, so
    // skipping them leaves every chain as it is.  Row l of JS is six independent 16-byte loads, no branch per output.
    // (history: a [3*NJ][NB] layout with a runtime bound per output made hipcc wait for each of ~105 loads in turn -- 16 of the
    //  prologue's 21 us; J0[o] loaded by lane 0 inside the butterfly was another 4.5 us of dependent round trips)
    float s[MAXJ * 3];
#pragma unroll
    for (int o = 0; o < MAXJ * 3; ++o) s[o] = 0.0f;
    if constexpr (WIDE) {
        // the same fmaf chains (a lane's l ascending; a coefficient outside the live ranges enters as 0: fma(w, 0, s) = s), but every load of a chunk of 7 x 64
        // coefficients -- the value and its six 16-byte row pieces, from clamped (always valid) addresses -- is in flight before the first FMA: one memory round
        // trip for FLAME's 400 betas where the loop below takes two and a wait per divergent `continue` (r04: 3.8 - 4.3 -> ~2 us of the prologue kernel)
        constexpr int CH = 7;
        const float* const bsrc = p ? p : a.betas + (int64_t)h * NB;
        for (int c0 = 0; c0 < NB; c0 += 64 * CH) {
            float v[CH];
            f32x4_t w[CH][MAXJ * 3 / 4];
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int lc = min(c0 + i * 64 + lane, NB - 1);
                v[i] = bsrc[lc];
                const f32x4_t* const row = (const f32x4_t*)(a.JS + (int64_t)lc * (MAXJ * 3));
#pragma unroll
                for (int q = 0; q < MAXJ * 3 / 4; ++q) w[i][q] = row[q];
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int l = c0 + i * 64 + lane;
                if (betas_out && l < NB) coef[(int64_t)l * cstride] = v[i];
                const float vv = (l < NB && (l < a.live0_end || (l >= a.live1_begin && l < a.live1_end))) ? v[i] : 0.0f;
#pragma unroll
                for (int o = 0; o < MAXJ * 3; ++o) s[o] = fmaf(w[i][o >> 2][o & 3], vv, s[o]);
            }
        }
    } else
#pragma unroll 4
    for (int l = lane; l < NB; l += 64) {
        const float v = p ? p[l] : a.betas[(int64_t)h * NB + l];
        if (betas_out) coef[(int64_t)l * cstride] = v;
        if (!(l < a.live0_end || (l >= a.live1_begin && l < a.live1_end))) continue;
        const f32x4_t* const row = (const f32x4_t*)(a.JS + (int64_t)l * (MAXJ * 3));
        f32x4_t w[MAXJ * 3 / 4];
#pragma unroll
        for (int q = 0; q < MAXJ * 3 / 4; ++q) w[q] = row[q];
#pragma unroll
        for (int o = 0; o < MAXJ * 3; ++o) s[o] = fmaf(w[o >> 2][o & 3], v, s[o]);
    }
    // the joint sums end up three per lane group of 8 (below): outputs jbase .. jbase + 2 in the lanes with lane % 8 == 0
    const int jbase = ((lane & 32) ? 12 : 0) + ((lane & 16) ? 6 : 0) + ((lane & 8) ? 3 : 0);
    float j0[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) j0[q] = ((lane & 7) == 0 && jbase + q < NJ * 3) ? a.J0[jbase + q] : 0.0f;
    if (p && lane < 13) S.tail[lane] = p[400 + lane];
    if (lane < NJ) S.par[lane] = a.parents[lane];
    if (lane < 3) S.unpad[lane] = a.unpad ? a.unpad[urow * 3 + lane] : (lane == 2 ? 1.0f : 0.0f);
    // full_pose = [global 0 | neck 0 | jaw | eyes 0]  (flame.py:141-148)
    if (lane < NJ * 3) {
        float v;
        if (p)
            v = (lane >= 6 && lane < 9) ? p[400 + lane - 6] : 0.0f;
        else
            v = a.pose[(int64_t)h * NJ * 3 + lane];
        S.pose[lane] = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    PMARK(a, h, lane, 1);
    {
        // 24 sums over the 64 lanes.  The addition tree of every output is the xor butterfly's (level by level the same two partial sums, own + partner's), but
        // on the first three levels the partners split the outputs between them -- a lane keeps half and sends the other half -- so that 12 + 6 + 3 + 3 x 3 = 30
        // cross-lane moves do what 24 x 6 = 144 did (r04: 1.85 -> 0.5 us of every prologue; same bits)
        static_assert(MAXJ * 3 == 24, "the halving below is written for 24 outputs");
        const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
        float t12[12], t6[6], t3[3];
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const float keep = b5 ? s[q + 12] : s[q], send = b5 ? s[q] : s[q + 12];
            t12[q] = keep + __shfl_xor(send, 32);
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const float keep = b4 ? t12[q + 6] : t12[q], send = b4 ? t12[q] : t12[q + 6];
            t6[q] = keep + __shfl_xor(send, 16);
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float keep = b3 ? t6[q + 3] : t6[q], send = b3 ? t6[q] : t6[q + 3];
            t3[q] = keep + __shfl_xor(send, 8);
        }
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int off = 4; off > 0; off >>= 1) t3[q] += __shfl_xor(t3[q], off);
        if ((lane & 7) == 0) {
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (jbase + q < NJ * 3) S.J[jbase + q] = j0[q] + t3[q];
        }
    }
    PMARK(a, h, lane, 2);
    // smplx batch_rodrigues per joint
    if (lane < NJ) {
        const float rx0 = S.pose[lane * 3 + 0], ry0 = S.pose[lane * 3 + 1], rz0 = S.pose[lane * 3 + 2];
        const float ex = rx0 + 1e-8f, ey = ry0 + 1e-8f, ez = rz0 + 1e-8f;
        const float angle = sqrtf(ex * ex + ey * ey + ez * ez);
        const float rx = rx0 / angle, ry = ry0 / angle, rz = rz0 / angle;
        const float sn = sinf(angle), cs = cosf(angle);
        const float K[9] = {0.0f, -rz, ry, rz, 0.0f, -rx, -ry, rx, 0.0f};
        float KK[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) KK[r * 3 + c] = K[r * 3 + 0] * K[0 * 3 + c] + K[r * 3 + 1] * K[1 * 3 + c] + K[r * 3 + 2] * K[2 * 3 + c];
#pragma unroll
        for (int e = 0; e < 9; ++e) S.R[lane * 9 + e] = ((e % 4 == 0) ? 1.0f : 0.0f) + sn * K[e] + (1.0f - cs) * KK[e];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    PMARK(a, h, lane, 3);
    // pose_feature = (rot_mats[:,1:] - I).view(-1)
    const int NP = 9 * (NJ - 1);
    const int prow0 = pose_row0 >= 0 ? pose_row0 : NB;
    if (lane < NP) coef[(int64_t)(prow0 + lane) * cstride] = S.R[9 + lane] - ((lane % 9) % 4 == 0 ? 1.0f : 0.0f);
    if (lane >= NP && NB + lane < a.Kp) coef[(int64_t)(prow0 + lane) * cstride] = 0.0f;
    // batch_rigid_transform: chain along parents, A_j = [Rg_j | tg_j - Rg_j J_j].  Twelve lanes own one entry each (lanes 0-8: the
    // 3x3 of Rg_j, lanes 9-11: tg_j), the chain state lives in LDS (a per-lane array indexed by the runtime `parents` would sit in
    // scratch memory: ~100 dependent global round trips, the bulk of the old single-lane prologue's ~45 us)
    for (int j = 0; j < NJ; ++j) {
        const int par = S.par[j];
        float rel[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) rel[c] = S.J[j * 3 + c] - (j > 0 ? S.J[par * 3 + c] : 0.0f);
        if (lane < 9) {
            const int r = lane / 3, c = lane - r * 3;
            S.Rg[j * 9 + lane] = (j == 0) ? S.R[lane]
                                          : S.Rg[par * 9 + r * 3 + 0] * S.R[j * 9 + 0 * 3 + c] + S.Rg[par * 9 + r * 3 + 1] * S.R[j * 9 + 1 * 3 + c] +
                                                S.Rg[par * 9 + r * 3 + 2] * S.R[j * 9 + 2 * 3 + c];
        } else if (lane < 12) {
            const int r = lane - 9;
            S.tg[j * 3 + r] = (j == 0) ? rel[r]
                                       : S.Rg[par * 9 + r * 3 + 0] * rel[0] + S.Rg[par * 9 + r * 3 + 1] * rel[1] + S.Rg[par * 9 + r * 3 + 2] * rel[2] + S.tg[par * 3 + r];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    PMARK(a, h, lane, 4);
    for (int e = lane; e < 12 * NJ; e += 64) {
        const int j = e / 12, q = e - j * 12, r = q >> 2, c = q & 3;
        float v;
        if (c < 3)
            v = S.Rg[j * 9 + r * 3 + c];
        else
            v = S.tg[j * 3 + r] - (S.Rg[j * 9 + r * 3 + 0] * S.J[j * 3 + 0] + S.Rg[j * 9 + r * 3 + 1] * S.J[j * 3 + 1] + S.Rg[j * 9 + r * 3 + 2] * S.J[j * 3 + 2]);
        hp[HP_A + e] = v;
        if (emit && a.joints_out && c == 3) a.joints_out[((int64_t)h * NJ + j) * 3 + r] = S.tg[j * 3 + r];
    }
    PMARK(a, h, lane, 5);
    if (lane == 1) {
        // rot_mat_from_6dof (utils.py:120-128): F.normalize eps = 1e-12, columns (b1, b2, b3) -- on a second lane, next to the chain
        float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        float sc = 1.0f, t[3] = {0, 0, 0};
        if (p) {
            const float* v = S.tail + 3;
            float b1[3], b3[3], b2[3];
            float n1 = fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-12f);
            for (int c = 0; c < 3; ++c) b1[c] = v[c] / n1;
            float cr[3] = {b1[1] * v[5] - b1[2] * v[4], b1[2] * v[3] - b1[0] * v[5], b1[0] * v[4] - b1[1] * v[3]};
            float n3 = fmaxf(sqrtf(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]), 1e-12f);
            for (int c = 0; c < 3; ++c) b3[c] = cr[c] / n3;
            b2[0] = -(b1[1] * b3[2] - b1[2] * b3[1]);
            b2[1] = -(b1[2] * b3[0] - b1[0] * b3[2]);
            b2[2] = -(b1[0] * b3[1] - b1[1] * b3[0]);
            for (int r = 0; r < 3; ++r) {
                R[r * 3 + 0] = b1[r];
                R[r * 3 + 1] = b2[r];
                R[r * 3 + 2] = b3[r];
            }
            sc = fmaxf(S.tail[12], 1e-8f);  // torch.clamp(scale, 1e-8)
            for (int c = 0; c < 3; ++c) t[c] = S.tail[9 + c];
        }
        for (int e = 0; e < 9; ++e) {
            hp[HP_R + e] = R[e];
            if (emit && a.rot_out) a.rot_out[(int64_t)h * 9 + e] = R[e];
        }
        hp[HP_S] = sc;
        for (int c = 0; c < 3; ++c) hp[HP_T + c] = t[c];
        hp[HP_U + 0] = S.unpad[0];
        hp[HP_U + 1] = S.unpad[1];
        hp[HP_U + 2] = S.unpad[2];
        if (emit && a.rpy_out) {
            // calculate_rpy (utils.py:146-151): Rotation.from_matrix(R^T).as_euler("xyz", degrees) in closed form.
            // M = R^T = Rz(c) Ry(b) Rx(a) (extrinsic xyz): b = -asin(M20), a = atan2(M21, M22), c = atan2(M10, M00);
            // at gimbal lock (|M20| = 1) scipy sets the third angle to 0 and folds it into the first.  fp32 throughout: R itself
            // is fp32, and atan2f stays within ~1e-5 degrees of the double-precision evaluation (which, being software fp64 atan2 on
            // a single lane, took ~40 us per launch)
            const float m00 = R[0], m10 = R[1], m20 = R[2], m21 = R[5], m22 = R[8], m01 = R[3], m11 = R[4];
            const float RAD = 57.29577951308232f;
            float ea, eb, ec;
            const float cb = sqrtf(m00 * m00 + m10 * m10);
            eb = atan2f(-m20, cb);
            if (cb > 1e-6f) {
                ea = atan2f(m21, m22);
                ec = atan2f(m10, m00);
            } else {
                ec = 0.0f;
                ea = (m20 < 0) ? atan2f(m01, m11) : atan2f(-m01, m11);
            }
            float ang[3] = {ec * RAD, ea * RAD - 180.0f, eb * RAD};  // roll = a[2], pitch = a[0] - 180, yaw = a[1]
            for (int c = 0; c < 3; ++c) {
                float g = ang[c];  // limit_angle (utils.py:131-143)
                if (g < -180.0f) {
                    const int q = (int)(g / 180.0f);                     // int() truncates, // floors (q <= -1 here)
                    const int fl = (q >= 0) ? q / 2 : -((-q + 1) / 2);
                    g += -2.0f * (float)fl * 180.0f;
                }
                if (g > 180.0f) g -= 2.0f * (float)((((int)(g / 180.0f)) + 1) / 2) * 180.0f;
                a.rpy_out[(int64_t)h * 3 + c] = g;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    PMARK(a, h, lane, 6);
}

// stand-alone prologue (large batches, and the detector's capacity-sized launches): one wave per head
__global__ __launch_bounds__(64) void flame_prep_kernel(PrepArgs a) {
    __shared__ PrepScratch S;
    __shared__ float s_hp[HP_SIZE];  // the head pack is assembled here and leaves as one coalesced store
    const int h = blockIdx.x, lane = threadIdx.x;
    if (a.n_dev && h >= *a.n_dev) return;
    prep_head<true>(a, h, lane, S, a.coef + h, a.npad, s_hp, true);
    float* const hpg = a.headpack + (int64_t)h * HP_SIZE;
    for (int e = lane; e < HP_SIZE; e += 64) hpg[e] = s_hp[e];
}

// HB heads per block, one wave each: the blend coefficients go to the TRANSPOSED scratch coef[k][head] (what the matrix-core kernels
// read 128-byte coalesced), so a lone wave's column is 436 scattered 4-byte stores -- at n = 8192 the single-head kernel spent 112 us
// mostly on that write amplification.  Here the HB columns meet in LDS and leave as HB*4-byte row segments.
template <int HB>
__global__ __launch_bounds__(HB * 64) void flame_prep_multi_kernel(PrepArgs a) {
    extern __shared__ __attribute__((aligned(16))) float psm[];
    PrepScratch* const S = (PrepScratch*)psm;                                   // [HB]
    float* const s_hp = psm + HB * (sizeof(PrepScratch) / sizeof(float));      // [HB][HP_SIZE]
    float* const s_coef = s_hp + HB * HP_SIZE;                                  // [Kp][HB]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int h0 = blockIdx.x * HB, h = h0 + wv;
    const int n = a.n_dev ? min(a.n, *a.n_dev) : a.n;
    if (h0 >= n) return;
    if (h < n) {
        prep_head(a, h, lane, S[wv], s_coef + wv, HB, s_hp + wv * HP_SIZE, true);
    } else {
        for (int k = lane; k < a.Kp; k += 64) s_coef[k * HB + wv] = 0.0f;
        for (int e = lane; e < HP_SIZE; e += 64) s_hp[wv * HP_SIZE + e] = 0.0f;
    }
    __syncthreads();
    for (int e = tid; e < a.Kp * HB; e += HB * 64) a.coef[(int64_t)(e / HB) * a.npad + h0 + (e % HB)] = s_coef[e];
    float* const hpg = a.headpack + (int64_t)h0 * HP_SIZE;
    for (int e = tid; e < HB * HP_SIZE; e += HB * 64) hpg[e] = s_hp[e];
}

struct VertArgs {
    const float* basis;  // [K][3][Vp]
    const float* basis8; // [Kp / 8][3][2][Vp][4]: the same values, k-interleaved (c3 tiles) or null
    const float* vt;     // [3][Vp]
    const float* wts;    // [NJ][Vp]
    const float* coef;   // [Kp][npad]
    const float* headpack;
    int npad;
    float* verts;  // [n][V][3] or null
    float* proj;   // [n][V][3] or null
    const int32_t* n_dev;  // live head count on the device (fused detector) or null
    int n, V, Vp, NJ, Kp;
    int r0_begin, r0_end, r1_begin, r1_end, r2_begin, r2_end;  // k ranges (shape live, expr live, pose)
    float z_offset;
    int do_unpad;
    int ablate;  // -DVGH_EXPERIMENTS only: bit0 no operand loads, bit1 no epilogue
};

// BT = threads per block, VPL = vertices per lane (4: 16-byte basis loads; 1: dword loads, 4x the blocks -- the basis stream of a
// handful of heads is pure latency, so it is spread over as many CUs as there are vertices / 64), HT = heads per block (every basis
// value fetched feeds HT FMAs).  FUSED: the block computes the prologue of its own HT heads (prep_head, one wave per head) instead
// of reading what flame_prep_kernel left in HBM: one launch, no dependent kernel boundary.  The per-vertex arithmetic (fmaf chain in
// ascending k, skinning, rigid, un-pad) is identical in every variant, so results do not depend on the batch a head is decoded in.
template <int HT, int BT, int VPL, bool FUSED>
__global__ __launch_bounds__(BT) void flame_vertex_kernel(VertArgs a, PrepArgs pa) {
#pragma clang fp contract(off)  // same reason as prep_head: identical roundings in every (HT, BT, VPL, FUSED) instantiation
    constexpr int UNR = VPL == 1 ? 16 : (HT >= 8 ? 4 : 8);
    constexpr int NWV = BT / 64;
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    float* s_coef = fsm;                   // [Kp][HT]
    float* s_hp = fsm + (size_t)a.Kp * HT; // [HT][HP_SIZE]
    const int tid = threadIdx.x;
    const int h0 = blockIdx.y * HT;
    if (a.n_dev) {
        a.n = min(a.n, *a.n_dev);
        if (h0 >= a.n) return;
    }
    if constexpr (FUSED) {
        PrepScratch* scr = (PrepScratch*)(s_hp + HT * HP_SIZE);  // one per wave
        const int wv = tid >> 6, lane = tid & 63;
        for (int hh = wv; hh < HT; hh += NWV) {
            if (h0 + hh < a.n) {
                prep_head(pa, h0 + hh, lane, scr[wv], s_coef + hh, HT, s_hp + hh * HP_SIZE, blockIdx.x == 0);
            } else {
                for (int k = lane; k < a.Kp; k += 64) s_coef[k * HT + hh] = 0.0f;
                for (int e = lane; e < HP_SIZE; e += 64) s_hp[hh * HP_SIZE + e] = 0.0f;
            }
        }
    } else {
        for (int e = tid; e < a.Kp * HT; e += BT) {
            const int k = e / HT, hh = e - k * HT;
            s_coef[e] = (h0 + hh < a.n) ? a.coef[(int64_t)k * a.npad + h0 + hh] : 0.0f;
        }
        for (int e = tid; e < HT * HP_SIZE; e += BT) {
            const int hh = e / HP_SIZE;
            s_hp[e] = (h0 + hh < a.n) ? a.headpack[(int64_t)(h0 + hh) * HP_SIZE + (e - hh * HP_SIZE)] : 0.0f;
        }
    }
    __syncthreads();
    const int v0 = (blockIdx.x * BT + tid) * VPL;
    if (v0 >= a.Vp) return;
    const int64_t plane = a.Vp;
    typedef float vecf __attribute__((ext_vector_type(VPL == 1 ? 1 : 4)));
    float acc[HT][3][VPL];
    {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const vecf tv = *(const vecf*)(a.vt + c * plane + v0);
#pragma unroll
            for (int hh = 0; hh < HT; ++hh)
#pragma unroll
                for (int e = 0; e < VPL; ++e) acc[hh][c][e] = ((const float*)&tv)[e];
        }
    }
    auto run = [&](int kb, int ke) {
        const float* bp = a.basis + (int64_t)kb * 3 * plane + v0;
        // the basis stream is pure latency at small n: keep UNR k-planes (3 loads each) in flight
#pragma unroll UNR
        for (int k = kb; k < ke; ++k, bp += 3 * plane) {
            const vecf bx = *(const vecf*)(bp);
            const vecf by = *(const vecf*)(bp + plane);
            const vecf bz = *(const vecf*)(bp + 2 * plane);
            const float* ck = s_coef + k * HT;
#pragma unroll
            for (int hh = 0; hh < HT; ++hh) {
                const float c = ck[hh];
#pragma unroll
                for (int e = 0; e < VPL; ++e) {
                    acc[hh][0][e] = fmaf(c, ((const float*)&bx)[e], acc[hh][0][e]);
                    acc[hh][1][e] = fmaf(c, ((const float*)&by)[e], acc[hh][1][e]);
                    acc[hh][2][e] = fmaf(c, ((const float*)&bz)[e], acc[hh][2][e]);
                }
            }
        }
    };
    run(a.r0_begin, a.r0_end);
    run(a.r1_begin, a.r1_end);
    run(a.r2_begin, a.r2_end);
    // skinning weights for this lane's vertices
    float wj[MAXJ][VPL];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j)
        if (j < a.NJ) {
            const vecf wv = *(const vecf*)(a.wts + (int64_t)j * plane + v0);
#pragma unroll
            for (int e = 0; e < VPL; ++e) wj[j][e] = ((const float*)&wv)[e];
        }
#pragma unroll
    for (int hh = 0; hh < HT; ++hh) {
        if (h0 + hh >= a.n) break;
        const float* hp = s_hp + hh * HP_SIZE;
        float outv[3 * VPL], outp[3 * VPL];
#pragma unroll
        for (int e = 0; e < VPL; ++e) {
            float T[12];
#pragma unroll
            for (int q = 0; q < 12; ++q) T[q] = 0.0f;
#pragma unroll
            for (int j = 0; j < MAXJ; ++j)
                if (j < a.NJ) {
#pragma unroll
                    for (int q = 0; q < 12; ++q) T[q] = fmaf(wj[j][e], hp[HP_A + j * 12 + q], T[q]);
                }
            const float px = acc[hh][0][e], py = acc[hh][1][e], pz = acc[hh][2][e];
            const float vx = fmaf(T[0], px, fmaf(T[1], py, fmaf(T[2], pz, T[3])));
            const float vy = fmaf(T[4], px, fmaf(T[5], py, fmaf(T[6], pz, T[7])));
            const float vz = fmaf(T[8], px, fmaf(T[9], py, fmaf(T[10], pz, T[11]))) + a.z_offset;
            outv[e * 3 + 0] = vx;
            outv[e * 3 + 1] = vy;
            outv[e * 3 + 2] = vz;
            const float* R = hp + HP_R;
            const float s = hp[HP_S];
            float qx = (R[0] * vx + R[1] * vy + R[2] * vz) * s + hp[HP_T + 0];
            float qy = (R[3] * vx + R[4] * vy + R[5] * vz) * s + hp[HP_T + 1];
            float qz = (R[6] * vx + R[7] * vy + R[8] * vz) * s + hp[HP_T + 2];
            if (a.do_unpad) {  // detector.py:67-69: x -= pad_x; y -= pad_y; all /= scale
                qx = (qx - hp[HP_U + 0]) / hp[HP_U + 2];
                qy = (qy - hp[HP_U + 1]) / hp[HP_U + 2];
                qz = qz / hp[HP_U + 2];
            }
            outp[e * 3 + 0] = qx;
            outp[e * 3 + 1] = qy;
            outp[e * 3 + 2] = qz;
        }
        const int64_t obase = ((int64_t)(h0 + hh) * a.V + v0) * 3;
        const int nv = min(VPL, a.V - v0);
        if (a.verts) {
            if (VPL == 4 && nv == 4 && (obase & 3) == 0) {
#pragma unroll
                for (int q = 0; q < 3; ++q) *(f32x4_t*)(a.verts + obase + q * 4) = f32x4_t{outv[(q * 4) % (3 * VPL)], outv[(q * 4 + 1) % (3 * VPL)], outv[(q * 4 + 2) % (3 * VPL)], outv[(q * 4 + 3) % (3 * VPL)]};
            } else {
                for (int q = 0; q < nv * 3; ++q) a.verts[obase + q] = outv[q];
            }
        }
        if (a.proj) {
            if (VPL == 4 && nv == 4 && (obase & 3) == 0) {
#pragma unroll
                for (int q = 0; q < 3; ++q) *(f32x4_t*)(a.proj + obase + q * 4) = f32x4_t{outp[(q * 4) % (3 * VPL)], outp[(q * 4 + 1) % (3 * VPL)], outp[(q * 4 + 2) % (3 * VPL)], outp[(q * 4 + 3) % (3 * VPL)]};
            } else {
                for (int q = 0; q < nv * 3; ++q) a.proj[obase + q] = outp[q];
            }
        }
    }
}

template <int HT, int BT, int VPL, bool FUSED>
int launch_vertex_cfg(const VertArgs& va, const PrepArgs& pa, hipStream_t st) {
    const size_t lds = ((size_t)va.Kp * HT + (size_t)HT * HP_SIZE) * sizeof(float) + (FUSED ? (BT / 64) * sizeof(PrepScratch) : 0);
    const int lanes = (va.Vp + VPL - 1) / VPL;
    const int groups = (va.n + HT - 1) / HT;
    hipLaunchKernelGGL((flame_vertex_kernel<HT, BT, VPL, FUSED>), dim3((lanes + BT - 1) / BT, groups), dim3(BT), lds, st, va, pa);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}

// ---- matrix-core vertex kernel ----------------------------------------------------------------------------------------------
// The blend is a dense [heads x K] . [K x 3V] contraction (K = 436, or 228 / 132 live): on v_mfma_f32_32x32x2_f32 it runs at the
// f32 vector rate with ONE VGPR per operand instead of a broadcast + FMA per element.  gfx950's f32 MFMA is an exact, k-ordered fmaf
// chain (D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)), one rounding per product), i.e. bit-for-bit what flame_vertex_kernel computes per
// vertex, so the two kernels are interchangeable and the choice between them is a pure speed choice.
//   tile roles : MFMA rows i = heads (A operand = coefficients, read from the transposed scratch coef[k][head]: 128-byte coalesced),
//                MFMA cols j = 32 consecutive vertices of ONE coordinate plane (B operand = basis[k][c][v], coalesced);
//                a wave owns 32 vertices x (x, y, z) x MT head tiles, so a lane ends up with x, y, z of its vertex for 16 heads per
//                tile and runs the skinning / rigid / un-pad epilogue on them in registers;
//   operands   : straight from L2 / Infinity Cache into VGPRs (the 26 MB basis is resident there), two bursts of UQ k-pairs in
//                flight; no LDS traffic in the loop (LDS only holds the head packs for the epilogue).
// Shared tail of the matrix-core kernels: skinning transform on the MFMA pipe, rigid transform + un-pad, 12-byte stores.
// acc[t][c]: blend result of heads h0 + hoff + t*32 + (MFMA row map) x vertex v, component c.  s_hpT: [HP_SIZE][NH] head packs of the
// block's NH heads (field-major).
template <int MT>
__device__ __forceinline__ void mfma_epilogue(const VertArgs& a, f32x16_t (&acc)[MT][3], const float* s_hpT, const int NH, const int hoff, const int h0, const int v,
                                              const int half, const int j, const int lane) {
#pragma clang fp contract(off)
    const int64_t plane = a.Vp;
    // ---- skinning on the matrix cores too: T[h][q](v) = sum_j w_j(v) A_j(h)[q] is a K = NJ contraction whose result lands in the
    //      SAME (head, vertex) -> (lane, register) map as the blend accumulators; chain order j ascending from 0, as the VALU kernel's
    //      fmaf chain (the pad joint contributes fma(0, 0, T) = T).  One output row (4 entries of T) at a time: 4 accumulators. ----
    if (VGH_ABLATE(a, 2)) {
        if (a.proj && lane == 0) a.proj[((int64_t)(h0 + hoff) * a.V + v) * 3] = acc[0][0][0] + acc[MT - 1][2][15];
        return;
    }
    // Branch-free over all MAXJ joint slots: the head packs hold exact zeros for joints >= NJ (prep_head clears them), so a pad joint
    // contributes fma(0, w, T) = T like the VALU chain's pad; the weight plane index is clamped (its value is multiplied by 0).
    // (per-lane conditional loads here made hipcc split the epilogue into ~200 exec-masked blocks and spill the accumulators)
    float wq[(MAXJ + 1) / 2];  // this lane's B operands: w_{2p + half}(v)
#pragma unroll
    for (int p = 0; p < (MAXJ + 1) / 2; ++p) {
        const int jn = 2 * p + half;
        const float w = a.wts[(int64_t)min(jn, a.NJ - 1) * plane + v];
        wq[p] = jn < a.NJ ? w : 0.0f;
    }
    const bool vok = v < a.V;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        float outv[3][16];
#pragma unroll
        for (int row = 0; row < 3; ++row) {
            // o = fmaf(T0, x, fmaf(T1, y, fmaf(T2, z, T3))) built innermost first, two transform entries at a time (register pressure:
            // four live 16-register accumulators next to the 96 blend accumulators spilled)
            float o[16];
#pragma unroll
            for (int qp = 1; qp >= 0; --qp) {
                f32x16_t T[2];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) T[q][r] = 0.0f;
#pragma unroll
                for (int p = 0; p < (MAXJ + 1) / 2; ++p) {
                    const int jn = 2 * p + half;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const float av = s_hpT[(HP_A + jn * 12 + row * 4 + qp * 2 + q) * NH + hoff + t * 32 + j];
                        T[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wq[p], T[q], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (qp == 1) o[r] = fmaf(T[0][r], acc[t][2][r], T[1][r]);                         // fmaf(T2, z, T3)
                    else o[r] = fmaf(T[0][r], acc[t][0][r], fmaf(T[1][r], acc[t][1][r], o[r]));  // fmaf(T0, x, fmaf(T1, y, .))
                }
                __builtin_amdgcn_sched_barrier(0);  // keep the transform entries sequential: the scheduler otherwise starts all of them at once
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) outv[row][r] = row == 2 ? o[r] + a.z_offset : o[r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int hh = hoff + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (!vok || h0 + hh >= a.n) continue;
            const float vx = outv[0][r], vy = outv[1][r], vz = outv[2][r];
            const int64_t obase = ((int64_t)(h0 + hh) * a.V + v) * 3;
            if (a.verts) *(f32x3_t*)(a.verts + obase) = f32x3_t{vx, vy, vz};  // 12 bytes per lane, contiguous across the wave
            if (a.proj) {
                const float* hp = s_hpT + hh;  // field f of this head: hp[f * NH] (all lanes of a half-wave read the same word)
                const float s = hp[HP_S * NH];
                float qx = (hp[(HP_R + 0) * NH] * vx + hp[(HP_R + 1) * NH] * vy + hp[(HP_R + 2) * NH] * vz) * s + hp[(HP_T + 0) * NH];
                float qy = (hp[(HP_R + 3) * NH] * vx + hp[(HP_R + 4) * NH] * vy + hp[(HP_R + 5) * NH] * vz) * s + hp[(HP_T + 1) * NH];
                float qz = (hp[(HP_R + 6) * NH] * vx + hp[(HP_R + 7) * NH] * vy + hp[(HP_R + 8) * NH] * vz) * s + hp[(HP_T + 2) * NH];
                if (a.do_unpad) {  // detector.py:67-69
                    const float us = hp[(HP_U + 2) * NH];
                    qx = (qx - hp[(HP_U + 0) * NH]) / us;
                    qy = (qy - hp[(HP_U + 1) * NH]) / us;
                    qz = qz / us;
                }
                *(f32x3_t*)(a.proj + obase) = f32x3_t{qx, qy, qz};
            }
            asm volatile("" ::: "memory");  // one head's 16 pack fields at a time: hoisting all 16 x 16 LDS reads costs 256 registers
        }
    }
}

template <int MT>
__global__ __launch_bounds__(256) void flame_mfma_kernel(VertArgs a) {
#pragma clang fp contract(off)
    constexpr int UQ = MT == 1 ? 16 : 8;  // k-pairs per burst (two bursts in flight: the basis comes from L2 / Infinity Cache; at MT = 1 the
                                           // accumulators leave room for twice the depth, and small batches are pure load latency)
    constexpr int NH = MT * 32;
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    float* s_hpT = fsm;  // [HP_SIZE][NH]: the head packs TRANSPOSED (field-major), so that 32 lanes reading one field of 32 heads hit 32 banks
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int h0 = blockIdx.y * NH;
    if (a.n_dev) a.n = min(a.n, *a.n_dev);
    if (h0 >= a.n) return;
    for (int e = tid; e < NH * HP_SIZE; e += 256) {
        const int hh = e / HP_SIZE, fld = e - hh * HP_SIZE;
        s_hpT[fld * NH + hh] = (h0 + hh < a.n) ? a.headpack[(int64_t)(h0 + hh) * HP_SIZE + fld] : 0.0f;
    }
    __syncthreads();
    const int vbase = (blockIdx.x * 4 + wv) * 32;
    if (vbase >= a.V) return;
    const int j = lane & 31, half = lane >> 5;
    const int v = vbase + j;  // < Vp (Vp is a multiple of 32)
    const int64_t plane = a.Vp;
    f32x16_t acc[MT][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float tv = a.vt[c * plane + v];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][c][r] = tv;
    }
    // ---- blend: ascending k over the three live ranges (even lengths), pairs (k, k+1) per MFMA ----
    const float* const bl = a.basis + v + (int64_t)half * 3 * plane;  // + k*3*plane + c*plane
    const float* const al = a.coef + h0 + j + (int64_t)half * a.npad; // + k*npad + t*32
    auto run = [&](int kb, int ke) {
        const int np = (ke - kb) >> 1;
        if (np <= 0) return;
        float B0[UQ][3], A0[UQ][MT], B1[UQ][3], A1[UQ][MT];
        auto fetch = [&](float (&B)[UQ][3], float (&A)[UQ][MT], int p0) {
#pragma unroll
            for (int u = 0; u < UQ; ++u) {
                const int p = (p0 + u < np) ? p0 + u : np - 1;  // past the end: a valid, unused pair
                const int64_t k = kb + 2 * p;
                if (VGH_ABLATE(a, 1)) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) B[u][c] = (float)(lane + c);
#pragma unroll
                    for (int t = 0; t < MT; ++t) A[u][t] = (float)(lane - t);
                    continue;
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) B[u][c] = bl[(k * 3 + c) * plane];
#pragma unroll
                for (int t = 0; t < MT; ++t) A[u][t] = al[k * a.npad + t * 32];
            }
        };
        auto consume = [&](const float (&B)[UQ][3], const float (&A)[UQ][MT], int p0) {
#pragma unroll
            for (int u = 0; u < UQ; ++u) {
                if (p0 + u < np) {  // wave-uniform
#pragma unroll
                    for (int t = 0; t < MT; ++t)
#pragma unroll
                        for (int c = 0; c < 3; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[u][t], B[u][c], acc[t][c], 0, 0, 0);
                }
            }
        };
        fetch(B0, A0, 0);
        for (int p = 0; p < np; p += 2 * UQ) {
            fetch(B1, A1, p + UQ);
            consume(B0, A0, p);
            fetch(B0, A0, p + 2 * UQ);
            consume(B1, A1, p + UQ);
        }
    };
    run(a.r0_begin, a.r0_end);
    run(a.r1_begin, a.r1_end);
    run(a.r2_begin, a.r2_end);
    mfma_epilogue<MT>(a, acc, s_hpT, NH, 0, h0, v, half, j, lane);
}

// Tail of the LDS-staged matrix-core kernel: the skinning / rigid / un-pad arithmetic of flame_vertex_kernel, statement for statement
// (same fmaf chains, same operation order => bit-identical), on the MFMA accumulator layout: register r of acc[t][c] is head
// hoff + t*32 + (r&3) + 8*(r>>2) + 4*half, vertex v.  s_hp: [NH][HP_SIZE] head packs (head-major: a half-wave reads ONE head, so every
// 16-byte LDS read is a broadcast).  One head at a time keeps this at ~40 live registers next to the 96 accumulators.
template <int MT>
__device__ __forceinline__ void valu_epilogue(const VertArgs& a, f32x16_t (&acc)[MT][3], const float* s_hp, const int hoff, const int h0, const int v, const int half) {
#pragma clang fp contract(off)
    const int64_t plane = a.Vp;
    float wj[MAXJ];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) wj[j] = j < a.NJ ? a.wts[(int64_t)j * plane + v] : 0.0f;
    const bool vok = v < a.V;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int hh = hoff + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float* hp = s_hp + hh * HP_SIZE;
            float T[12];
#pragma unroll
            for (int q = 0; q < 12; ++q) T[q] = 0.0f;
#pragma unroll
            for (int j = 0; j < MAXJ; ++j)
                if (j < a.NJ) {  // wave-uniform
                    const f32x4_t A0 = *(const f32x4_t*)(hp + HP_A + j * 12), A1 = *(const f32x4_t*)(hp + HP_A + j * 12 + 4), A2 = *(const f32x4_t*)(hp + HP_A + j * 12 + 8);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        T[q] = fmaf(wj[j], A0[q], T[q]);
                        T[4 + q] = fmaf(wj[j], A1[q], T[4 + q]);
                        T[8 + q] = fmaf(wj[j], A2[q], T[8 + q]);
                    }
                }
            const float px = acc[t][0][r], py = acc[t][1][r], pz = acc[t][2][r];
            const float vx = fmaf(T[0], px, fmaf(T[1], py, fmaf(T[2], pz, T[3])));
            const float vy = fmaf(T[4], px, fmaf(T[5], py, fmaf(T[6], pz, T[7])));
            const float vz = fmaf(T[8], px, fmaf(T[9], py, fmaf(T[10], pz, T[11]))) + a.z_offset;
            if (vok && h0 + hh < a.n) {
                const int64_t obase = ((int64_t)(h0 + hh) * a.V + v) * 3;
                if (a.verts) *(f32x3_t*)(a.verts + obase) = f32x3_t{vx, vy, vz};  // 12 bytes per lane, contiguous across a half-wave
                if (a.proj) {
                    const f32x4_t R0 = *(const f32x4_t*)(hp + HP_R), R1 = *(const f32x4_t*)(hp + HP_R + 4), R2 = *(const f32x4_t*)(hp + HP_R + 8), R3 = *(const f32x4_t*)(hp + HP_R + 12);
                    // R0 = R[0..3], R1 = R[4..7], R2 = {R[8], s, t0, t1}, R3 = {t2, u0, u1, u2}
                    const float s = R2[1];
                    float qx = (R0[0] * vx + R0[1] * vy + R0[2] * vz) * s + R2[2];
                    float qy = (R0[3] * vx + R1[0] * vy + R1[1] * vz) * s + R2[3];
                    float qz = (R1[2] * vx + R1[3] * vy + R2[0] * vz) * s + R3[0];
                    if (a.do_unpad) {  // detector.py:67-69
                        qx = (qx - R3[1]) / R3[3];
                        qy = (qy - R3[2]) / R3[3];
                        qz = qz / R3[3];
                    }
                    *(f32x3_t*)(a.proj + obase) = f32x3_t{qx, qy, qz};
                }
            }
            asm volatile("" ::: "memory");  // one head at a time
        }
    }
}

// Crowd-scale variant of the matrix-core kernel: block = 128 heads x 128 vertices (8 waves = 2 head groups x 4 vertex groups, each
// wave the same 64 heads x 32 vertices x 3 components as flame_mfma_kernel<2>), operands staged through LDS by LDS-DMA in slabs of
// 8 k-pairs, double buffered.  Why: the register-fed kernel loads 5 operand dwords per lane per 6 MFMAs straight from L2 -- 13 B/clk/CU,
// which is what pins it (and the VALU kernel) near 66 TFLOP/s at n = 8192; here a slab of 32 KB feeds 384 MFMAs (49 FLOP per byte
// from L2, < 3.2 TB/s at the full fp32 matrix rate) and the MFMA operands are conflict-free ds_read_b32.  Same k-ordered chain per
// output element, so results are bit-identical to the other FLAME kernels.
constexpr int LB_V = 128;                  // vertices per block
// k rows per slab (floats of a basis slab [k][c][128] = KS * 3 * LB_V): 16 for the 64- / 128-head blocks, 32 for the 32-head blocks (r04: the pieces of a slab
// -- KS * LBH / 256 coefficient pieces + KS * 384 / 256 basis pieces -- must deal evenly to the waves: 4 + 48 over 4 waves)
template <int LBH>
constexpr int lb_ks() { return LBH == 32 ? 32 : 16; }

// LBH = heads per block: 128 (8 waves = 2 head groups x 4 vertex groups; crowd scale) or 64 (4 waves: twice the blocks for the same batch --
// a few hundred heads then still give every CU work -- at the price of the basis slab feeding half as many MFMAs)
template <int WPS, int LBH>  // WPS: waves per SIMD the register allocation aims for
__global__ __launch_bounds__((LBH >= 64 ? LBH / 64 : 1) * 256, WPS) void flame_mfma_lds_kernel(VertArgs a) {
#pragma clang fp contract(off)
    constexpr int MT = LBH >= 64 ? 2 : 1;                      // 32-head MFMA tiles per wave (LBH = 32: one -- a few dozen heads still make 40 x ceil(n / 32) blocks)
    constexpr int LB_KS = lb_ks<LBH>(), LB_B = LB_KS * 3 * LB_V;
    constexpr int NT = (LBH >= 64 ? LBH / 64 : 1) * 256, NWV = NT / 64;  // threads, waves
    constexpr int LB_A = LB_KS * LBH, LB_STAGE = LB_A + LB_B;  // floats per slab: coefficients [k][LBH] + basis
    constexpr int NPA = LB_A / 256, NPT = NPA + LB_B / 256;    // 1 KiB LDS-DMA pieces per slab: coefficient pieces, all pieces
    constexpr int PPW = NPT / NWV;                             // pieces per wave (4 / 7)
    constexpr int RPP = 256 / LBH, LPR = 64 / RPP;             // coefficient rows per piece (2 / 4), lanes per row
    static_assert(NPT % NWV == 0 && LBH * HP_SIZE <= 2 * LB_STAGE, "slab pieces must split evenly; the head packs reuse the two slabs");
    extern __shared__ __attribute__((aligned(16))) float fsm[];  // 2 slabs during the blend loop, then the head packs [LBH][HP_SIZE]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int hw = wv >> 2, vw = wv & 3;
    const int h0 = blockIdx.x * LBH, v0 = blockIdx.y * LB_V;  // heads fastest: co-resident blocks share the (3x larger) basis slabs
    if (a.n_dev) a.n = min(a.n, *a.n_dev);
    if (h0 >= a.n) return;
    const int j = lane & 31, half = lane >> 5;
    const int v = v0 + vw * 32 + j;
    const int64_t plane = a.Vp;
    f32x16_t acc[MT][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float tv = (v < a.Vp) ? a.vt[c * plane + v] : 0.0f;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][c][r] = tv;
    }
    // virtual k-pair index q over the three live ranges -> first k of the pair
    const int np0 = (a.r0_end - a.r0_begin) >> 1, np1 = (a.r1_end - a.r1_begin) >> 1, np2 = (a.r2_end - a.r2_begin) >> 1;
    const int npt = np0 + np1 + np2;
    const int nstage = (npt + LB_KS / 2 - 1) / (LB_KS / 2);
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)a.coef, 0, (unsigned)((int64_t)a.Kp * a.npad * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)a.basis, 0, (unsigned)((int64_t)a.r2_end * 3 * plane * 4), 0x00020000);  // r2_end = K: the basis has K rows
    // first k of every virtual pair (-1 past the end), once per block: keeps the per-slab address math to one LDS read + one multiply
    int* const s_kq = (int*)(fsm + 2 * LB_STAGE);
    for (int q = tid; q < nstage * (LB_KS / 2); q += NT)
        s_kq[q] = q < np0 ? a.r0_begin + 2 * q : q < np0 + np1 ? a.r1_begin + 2 * (q - np0) : q < npt ? a.r2_begin + 2 * (q - np0 - np1) : -1;
    __syncthreads();
    // per-piece constants of this lane: k slot inside the slab and the k-independent part of the byte offset.  A coefficient piece is RPP rows of
    // LBH floats ([k][npad] in memory), a basis piece two 128-float rows rr -> (k slot, component) of [k][c][Vp]
    int p_ks[PPW];
    unsigned p_mul[PPW], p_add[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int piece = wv * PPW + i;
        if (piece < NPA) {
            p_ks[i] = piece * RPP + lane / LPR;
            p_mul[i] = (unsigned)a.npad * 4u;
            p_add[i] = (unsigned)(h0 + (lane % LPR) * 4) * 4u;
        } else {
            const int rr = (piece - NPA) * 2 + half;
            const int ks = rr / 3, c = rr - ks * 3;
            p_ks[i] = ks;
            p_mul[i] = (unsigned)plane * 12u;
            p_add[i] = (unsigned)(c * plane + v0 + (lane & 31) * 4) * 4u;
        }
    }
    auto issue = [&](int st, int buf) {
        float* const base = fsm + buf * LB_STAGE;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int piece = wv * PPW + i;  // wave-uniform
            const int k0 = s_kq[st * (LB_KS / 2) + (p_ks[i] >> 1)];
            const unsigned off = k0 >= 0 ? (unsigned)(k0 + (p_ks[i] & 1)) * p_mul[i] + p_add[i] : 0xFFFFFFF0u;  // past the end: out of range -> zeros
            if (piece < NPA) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (AS3 void*)(base + piece * 256), 16, off, 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (AS3 void*)(base + LB_A + (piece - NPA) * 256), 16, off, 0, 0, 0);
        }
    };
    // raw barrier: __syncthreads() carries a fence that drains vmcnt(0) -- the next slab's LDS-DMA loads would be waited for at every barrier and the
    // double buffer would overlap nothing.  The counted s_waitcnt above the first barrier of a step is the only load wait (r03: -4 ... -6 % on every
    // LDS-staged decode; bit-identical)
    auto lds_barrier = [] {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    issue(0, 0);
    for (int st = 0; st < nstage; ++st) {
        const int buf = st & 1;
        if (st + 1 < nstage) {
            issue(st + 1, buf ^ 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        lds_barrier();
        const float* const sa = fsm + buf * LB_STAGE + half * LBH + hw * (32 * MT) + j;       // + pair*2*LBH + t*32
        const float* const sb = fsm + buf * LB_STAGE + LB_A + half * 3 * LB_V + vw * 32 + j;  // + pair*6*LB_V + c*LB_V
        const int pairs = min(LB_KS / 2, npt - st * (LB_KS / 2));
        float A[2][MT], B[2][3];
        auto rd = [&](int u, float (&Ao)[MT], float (&Bo)[3]) {
#pragma unroll
            for (int t = 0; t < MT; ++t) Ao[t] = sa[u * 2 * LBH + t * 32];
#pragma unroll
            for (int c = 0; c < 3; ++c) Bo[c] = sb[u * 6 * LB_V + c * LB_V];
        };
        rd(0, A[0], B[0]);
#pragma unroll
        for (int u = 0; u < LB_KS / 2; ++u) {
            if (u + 1 < LB_KS / 2) rd(u + 1, A[(u + 1) & 1], B[(u + 1) & 1]);  // next pair's operands under this pair's MFMAs
            if (u < pairs) {  // wave-uniform
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[u & 1][t], B[u & 1][c], acc[t][c], 0, 0, 0);
            }
        }
        lds_barrier();  // slab `buf` is rewritten by the loads issued at the top of the next iteration
    }
    // head packs of the block's heads into the (now free) slab memory: a straight copy, [head][HP_SIZE]
    float* const s_hp = fsm;
    for (int e = tid * 4; e < LBH * HP_SIZE; e += NT * 4) {
        const int hh = e / HP_SIZE;
        *(f32x4_t*)(s_hp + e) = (h0 + hh < a.n) ? *(const f32x4_t*)(a.headpack + (int64_t)h0 * HP_SIZE + e) : f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
    }
    __syncthreads();
    if (v >= a.Vp) return;
    valu_epilogue<MT>(a, acc, s_hp, hw * (32 * MT), h0, v, half);
}

template <int WPS, int LBH>
int launch_mfma_lds(const VertArgs& va, hipStream_t st) {
    constexpr int LB_KS = lb_ks<LBH>(), LB_B = LB_KS * 3 * LB_V;
    const size_t lds = (size_t)2 * (LB_KS * LBH + LB_B) * sizeof(float) + 1024;  // two slabs + the pair table (<= 256 entries): two blocks per CU (one for LBH = 32)
    static std::atomic<int> attr_done[16];
    int dev = 0;
    VGH_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < 16 && !attr_done[dev].load(std::memory_order_acquire)) {
        VGH_HIP(hipFuncSetAttribute((const void*)flame_mfma_lds_kernel<WPS, LBH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done[dev].store(1, std::memory_order_release);
    }
    hipLaunchKernelGGL((flame_mfma_lds_kernel<WPS, LBH>), dim3((va.n + LBH - 1) / LBH, (va.V + LB_V - 1) / LB_V), dim3((LBH >= 64 ? LBH / 64 : 1) * 256), lds, st, va);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}

template <int MT>
int launch_mfma(const VertArgs& va, hipStream_t st) {
    const size_t lds = (size_t)MT * 32 * HP_SIZE * sizeof(float);
    const int vgroups = (va.V + 31) / 32, hgroups = (va.n + MT * 32 - 1) / (MT * 32);
    hipLaunchKernelGGL((flame_mfma_kernel<MT>), dim3((vgroups + 3) / 4, hgroups), dim3(256), lds, st, va);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}

// ---- component-split matrix-core kernel ("c3" tiles, r04) -------------------------------------------------------------------
// Why: between a handful and ~100 heads the register-fed kernel above is neither MFMA- nor bandwidth-bound, it is ONE WAVE'S CHAIN long: a wave owns
// 32 heads x 32 vertices x (x, y, z) = 3 matrix instructions of 64 cycles and 4 operand dwords per k-pair, 218 pairs, on a single SIMD, while (n = 96: 471 waves
// on 1 024 SIMDs) most of the chip has nothing to do; measured 72 ns per k.  Here a block is NV vertices x 32 heads and its three compute waves take ONE
// coordinate plane each: a third of the chain per wave and three times the waves; NV = 16 (MFMA columns 16 .. 31 mirror 0 .. 15, the matrix pipe is not what
// is short) doubles the blocks again so that 32 heads already reach every CU.  The coefficient tile of the block's heads sits in LDS ([k][head], row stride 33:
// k-major staging writes and head-major operand reads both spread over the banks), so the only vector-memory stream of the K loop is the basis: one dword per
// lane and k-pair, 48 pairs in flight per wave, ONE pair sequence over the three live ranges (no drain between them).  The shape / expression coefficients are
// the raw parameters, read in place (any head_row indirection included); the pose features and head packs come from the prologue.  The price of the split is an
// exchange: the blend results meet in LDS ([component][register][lane], conflict-free both ways; the buffer aliases the coefficient tile) and the epilogue --
// flame_vertex_kernel's skinning / rigid / un-pad statements, one (head, vertex) per lane, so the bits are the same -- is dealt slot by slot (a slot = one
// accumulator register = 2 heads x 32 vertices) to ALL waves of the block.
//   NPW = 0: pose features and head packs come from the prologue kernel (coef scratch rows NB .., headpack).
//   NPW > 0 ("fused", n <= NPW heads): NPW extra waves run prep_head for one head each WHILE the compute waves stream the shape / expression range; the
//            pose-feature pairs and the epilogue wait for them at one barrier.  One launch, no dependent kernel boundary, the prologue's ~7 us under the stream.
#ifdef VGH_EXPERIMENTS
#define C3MARK(i) do { if (pa.trace && bx == 1 && by == 0 && threadIdx.x == 0) { pa.trace[8 + (i)] = wall_clock64(); if ((i) == 1 || (i) == 2) pa.trace[13 + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)  /* [14], [15]: shader-clock ticks around the K loop */
#else
#define C3MARK(i) do { } while (0)
#endif

//   NPW > 0 ("fused", n <= NPW heads, 3 + NPW waves): NPW extra waves run prep_head for one head each WHILE the compute waves stream the shape / expression
//            groups (raw betas read in place); the pose groups and the epilogue wait for them at one barrier.  One launch, the prologue's ~8 us under the stream.
//   NPW = 0 (3 VG compute waves + NHL helpers that stage and take epilogue slots; VG = 1: 5 helpers for a single head tile, 1 from two tiles on so that two blocks
//            share a CU): coefficients and head packs come from the prologue KERNEL launched before.
//   VG = 4 / 5 (a few hundred heads and more): a block is 128 (160) vertices x 32 heads, its TWELVE (fifteen) compute waves (vertex groups x three planes, three - four chains per SIMD:
//            the matrix pipe of a CU stays busy from one block) share ONE coefficient tile -- 56 KB of LDS per 12 waves instead of per 3, operands still one
//            16-byte basis load per lane and 4 pairs; vertex blocks of one head tile run on one XCD (40 blocks per tile row, 40 % 8 = 0) and share its L2.
//   (r04, measured and removed: the prologue in the first blocks of the SAME launch, released to the vertex blocks by per-head flags -- agent-scope release /
//    acquire = buffer_wbl2 / buffer_inv of a whole L2 and hundreds of polling waves: the flag of a lone head became visible 14 us into the launch, EXPERIMENTS 8d)
//   Q = 1 ("quad" tiles, r06; VG = 1, MT = 1): a compute wave owns 16 heads x 16 vertices of its plane and runs v_mfma_f32_16x16x4_f32 -- FOUR k per instruction at a
//            dependent-chain latency of 44 cycles where v_mfma_f32_32x32x2_f32 takes 64 cycles for two (profiles/r04_mfma_f32_chain.txt: 11.0 vs 32.1 cycles per k;
//            both are the ascending fmaf chain, bit for bit).  Below ~32 heads a decode is ONE WAVE'S CHAIN long, so the chain is what there is to shorten: 109
//            instructions instead of 218, 4.8 k instead of 14 k cycles.  Same operands: a lane (vertex = lane & 15, kq = lane >> 4) loads the 16 bytes of basis8 row
//            kq & 1 -- its vertex at k = 8g + (kq & 1) + {0, 2, 4, 6} -- and uses element kq >> 1 for the quad 8g .. 8g + 3 and 2 + (kq >> 1) for 8g + 4 .. 8g + 7 (half of
//            the loaded bytes are unused: the price of keeping one basis layout); the coefficient tile is read at rows kq and 4 + kq of the group.  Twice the vertex
//            blocks (314 groups of 16), head tiles of 16.
template <int NPW, int NHL, int VG, int MT = 1, int Q = 0>
__global__ __launch_bounds__((3 * VG + NPW + NHL) * 64) void flame_c3_kernel(VertArgs a, PrepArgs pa) {
#pragma clang fp contract(off)
    static_assert(!(NPW > 0 && NHL > 0) && (VG == 1 || NPW == 0) && (MT == 1 || (NPW == 0 && NHL == 0 && VG > 1)), "prologue waves or helper waves; the fused variant has one vertex group; two head tiles per wave only in the large blocks");
    static_assert(!Q || (VG == 1 && MT == 1), "quad tiles: one 16-vertex group, one 16-head tile per block");
    constexpr int NCW = 3 * VG;             // compute waves: wave w = vertex group w / 3, coordinate plane w % 3
    constexpr int NW = NCW + NPW + NHL;
    constexpr int VW = Q ? 16 : 32;         // vertices of a vertex group
    constexpr int NH = NPW > 0 ? NPW : Q ? 16 : 32 * MT;  // head packs held by the block (MT = 2: each compute wave runs TWO head tiles' chains on one basis operand -- half the
                                                 // operand bytes per MFMA: a CU's L1 fill rate, ~16 B/clk, is what a block of twelve one-tile waves runs into)
    constexpr int AS = NPW > 0 ? 33 : 32;   // row stride of the coefficient tile: 32 = what an LDS-DMA instruction writes (8 rows x 128 bytes; the two half-waves of an
                                            // operand read then cover the 64 banks); 33 for the fused variant's k-major register staging
    constexpr int UQ = NW >= 14 ? 5 : NW >= 11 ? (MT == 2 ? 7 : 9) : 14;   // k-groups (4 pairs, one 16-byte load per lane) per burst, THREE bursts in flight = 168 of the longest chain's 220 pairs
                                            // (11 waves leave 168 registers per lane: 9 groups per burst):
                                            // a burst is asked for two consume times (2 x 56 MFMAs) ahead, more than a load takes under this traffic
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (a.n_dev) a.n = min(a.n, *a.n_dev);
    // block -> (head tile by, vertex block bx) so that the tiles of ONE vertex block run on one XCD and share its L2 copy of the basis rows: workgroups go to the
    // XCDs round-robin (XCD = blockIdx.x % 8); inside an XCD's sequence the vertex blocks bx = x, x + 8, ... of a tile row come first, then the next row
    const int vgroups = (a.V + VW - 1) / VW, vblocks = (vgroups + VG - 1) / VG, vb8 = (vblocks + 7) >> 3;
    const int bid = (int)blockIdx.x, bq = bid >> 3;
    const int by = bq / vb8, bx = (bq - by * vb8) * 8 + (bid & 7);
    if (bx >= vblocks) return;  // (the row is padded to a multiple of 8 blocks)
    const int cvg = wv < NCW ? wv / 3 : 0, cpl = wv < NCW ? wv - cvg * 3 : 0;  // this wave's vertex group inside the block and its coordinate plane
    const bool cw = wv < NCW && bx * VG + cvg < vgroups;                       // a compute wave with vertices (the last block of a tile row may have idle groups)
    // live k-groups (8 consecutive k = 4 MFMA pairs = one row block of the interleaved basis copy): the groups that hold a pair of the shape, expression or pose
    // range, in ascending k; gi -> g.  A pair is live when its k lies in one of the ranges (even bounds: a pair is in or out as a whole).
    const int g0e = (a.r0_end + 7) >> 3;  // r0 starts at 0
    const int g1b = a.r1_end > a.r1_begin ? max(a.r1_begin >> 3, g0e) : g0e, g1e = a.r1_end > a.r1_begin ? max((a.r1_end + 7) >> 3, g1b) : g0e;
    const int g2b = max(a.r2_begin >> 3, g1e), g2e = max((a.r2_end + 7) >> 3, g2b);
    const int c0 = g0e, c01 = c0 + (g1e - g1b), ng = c01 + (g2e - g2b);
    auto gof = [&](int gi) { return gi < c0 ? gi : gi < c01 ? g1b + (gi - c0) : g2b + (gi - c01); };
    auto live = [&](int k) { return k < a.r0_end || (k >= a.r1_begin && k < a.r1_end) || (k >= a.r2_begin && k < a.r2_end); };
    float* const s_A = fsm;  // [ng][MT][8][AS] coefficients of heads h0 + 32 t .. + 31: row (gi * MT + t) * 8 + (k & 7); after the blend s_x [VG][MT][3][16][64]
    // the exchange buffer aliases the tile (one barrier between them), except in the large one-tile blocks (VG > 1, MT = 1): there it has its own memory behind
    // the head packs, so that a vertex group's three waves hand over among themselves (an LDS counter) and start their epilogue while other groups' chains still run
    constexpr bool XOWN = VG > 1 && MT == 1;
    float* const s_hp = fsm + ((max(ng * MT * 8 * AS, XOWN ? 0 : VG * MT * 3 * 16 * 64) + 3) & ~3);  // [NH][HP_SIZE] head packs, head-major (16-byte broadcast reads)
    float* const s_x = XOWN ? s_hp + NH * HP_SIZE : fsm;
    int* const s_cnt = (int*)(s_x + VG * MT * 3 * 16 * 64);  // XOWN: one arrival counter per vertex group
    if (XOWN && threadIdx.x < VG) s_cnt[threadIdx.x] = 0;     // (ahead of the staging barrier)
    const int h0 = by * (Q ? 16 : 32 * MT);
    if (h0 >= a.n) return;
    // 32 x 32 x 2: lane = (head / vertex j = lane & 31, k parity half = lane >> 5); quad tiles: lane = (head / vertex j = lane & 15, k within the quad kq = lane >> 4),
    // whose basis8 row is kq & 1 ("half") and whose element inside the 16 loaded bytes is kq >> 1 (+ 2 for the group's second quad)
    const int j = Q ? lane & 15 : lane & 31, half = Q ? (lane >> 4) & 1 : lane >> 5;
    const int kq = lane >> 4;
    const int v = min(bx * VG + cvg, vgroups - 1) * VW + j;  // < Vp (a multiple of 32); helper / prologue waves: the (one) vertex group, idle compute waves: a valid one
    const int64_t plane = a.Vp;
    C3MARK(0);
    // ---- blend operands: the k-interleaved basis copy [k / 8][c][k & 1][Vp][4]: 16 bytes of a lane = its vertex at k = 8g + half + {0, 2, 4, 6} -- the B operands of
    //      the four pairs of group g.  One vector-memory instruction per 4 pairs, so that 28 instructions in flight are half of the longest chain (what bounds this
    //      kernel is how much of its chain a wave has in flight: with dword loads and 48 pairs in flight the K loop ran 111 ns per pair, a 64-cycle MFMA apart) ----
    const f32x4_t* const bl = (const f32x4_t*)a.basis8 + ((int64_t)cpl * 2 + half) * plane + v;  // + g * 6 * plane
    f32x4_t B0[UQ], B1[UQ], B2[UQ];
    f32x16_t acc[MT];
    f32x4_t accq;  // quad tiles: D[head 4 (lane >> 4) + r][vertex lane & 15]
    auto fetch = [&](f32x4_t (&B)[UQ], int g0) {
        if (g0 >= ng || !cw) return;
#pragma unroll
        for (int u = 0; u < UQ; ++u) {
            const int gi = (g0 + u < ng) ? g0 + u : ng - 1;  // past the end: a valid, unused group
            B[u] = bl[(int64_t)gof(gi) * 6 * plane];
        }
    };
    // LDS-DMA of the tile, 1 KiB per instruction = 8 rows of the transposed scratch coef[k][head] (128 bytes per row and tile) = one live k-group
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)a.coef, 0, (unsigned)((int64_t)a.Kp * a.npad * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc((void*)(a.headpack + (int64_t)h0 * HP_SIZE), 0, (unsigned)(NH * HP_SIZE * 4), 0x00020000);
    auto dma_group = [&](int gi) {
#pragma unroll
        for (int t = 0; t < MT; ++t)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (AS3 void*)(s_A + (gi * MT + t) * 256), 16, (unsigned)((gof(gi) * 8 + (lane >> 3)) * a.npad + h0 + t * 32 + (lane & 7) * 4) * 4u, 0, 0, 0);
    };
    auto dma_packs = [&](int first, int step) {  // head packs as they lie
        for (int g = first; g < NH * HP_SIZE / 256; g += step) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_h, (AS3 void*)(s_hp + g * 256), 16, (unsigned)(g * 256 + lane * 4) * 4u, 0, 0, 0);
    };
    // rows of a live group whose k is outside the live ranges (group 37 when the shape range ends before 300, the tail of a range that is not a multiple of 8,
    // the pad behind the pose features) are cleared: their pairs then add fma(0, b, acc) = acc, and the K loop runs without a branch per pair
    auto zero_dead_rows = [&](int gi) {
        if (gi < 0 || gi >= ng) return;
        const int k0 = gof(gi) * 8;
        if (live(k0) && live(k0 + 7)) return;
        for (int r = lane >> 5; r < 8; r += 2)
            if (!live(k0 + r)) {
#pragma unroll
                for (int t = 0; t < MT; ++t) s_A[((gi * MT + t) * 8 + r) * AS + (lane & 31)] = 0.0f;
            }
    };
    float wj[MAXJ];  // skinning weights of this lane's vertex (every wave takes epilogue slots)
#pragma unroll
    for (int q = 0; q < MAXJ; ++q) wj[q] = q < a.NJ ? a.wts[(int64_t)q * plane + v] : 0.0f;
    {
        const float tv = a.vt[cpl * plane + v];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = tv;
        accq = f32x4_t{tv, tv, tv, tv};
    }
    if constexpr (NPW > 0) {
        // fused: the raw betas of the block's heads, read in place.  Every compute wave stages the whole (small) tile itself -- identical values from every
        // writer, so a wave needs nothing but its own writes to have landed and no hand-over exists that the prologue waves (busy until the pose barrier) would
        // have to attend.  A lane owns head lane % NPW and the live coefficients e = lane / NPW + (64 / NPW) i of the two ranges taken as one sequence; a batch
        // of 32 loads per lane (everything up to 4 heads, two batches at 8) is issued, THEN the first two basis bursts, and only then the values are written:
        // one memory round trip from the kernel top to the first MFMA (the loads of a range, a wait, its writes, the next range, then the basis cost 5.2 us)
        constexpr int KPI = 64 / NPW, NB_ = NPW == 1 ? 7 : NPW == 2 ? 14 : NPW == 4 ? 28 : 32;  // loads per lane and batch: FLAME's 400 betas in one batch up to 4 heads
        const int hh = lane & (NPW - 1), ks = lane / NPW;
        const int n0 = a.r0_end - a.r0_begin, nlive = n0 + (a.r1_end - a.r1_begin);
        // every lane loads, from a valid address (a head slot past the batch reads the last head's row -- MFMA rows are independent and those rows are never
        // stored; an index past the sequence re-reads its last element): a load under a per-lane condition makes hipcc wait vmcnt(0) per load
        const float* src;
        {
            const int hs = min(hh, a.n - 1);
            const int64_t prow = pa.head_row ? pa.head_row[hs] : hs;
            src = pa.params ? pa.params + prow * VGH_NUM_FLAME_PARAMS : pa.betas + (int64_t)hs * pa.NB;
        }
        auto kof_e = [&](int e) { return e < n0 ? a.r0_begin + e : a.r1_begin + (e - n0); };
        auto stage_load = [&](float (&t)[NB_], int e0) {
#pragma unroll
            for (int i = 0; i < NB_; ++i) {
                const int e = e0 + i * KPI + ks;
                t[i] = src[kof_e(min(e, nlive - 1))];
            }
        };
        auto stage_write = [&](const float (&t)[NB_], int e0) {
#pragma unroll
            for (int i = 0; i < NB_; ++i) {
                const int e = e0 + i * KPI + ks;
                if (e < nlive) {
                    const int k = kof_e(e), g = k >> 3;
                    s_A[((g < g0e ? g : c0 + (g - g1b)) * 8 + (k & 7)) * AS + hh] = t[i];
                }
            }
        };
        if (cw && !VGH_ABLATE(a, 4)) {
            zero_dead_rows(c0 - 1);
            if (c01 > c0) {
                zero_dead_rows(c0);
                zero_dead_rows(c01 - 1);
            }
            float t[NB_];
            stage_load(t, 0);
            fetch(B0, 0);  // the basis stream starts behind the tile's loads (in-order return: whatever is issued first is waited for first)
            fetch(B1, UQ);
            stage_write(t, 0);
            for (int e0 = KPI * NB_; e0 < nlive; e0 += KPI * NB_) {
                stage_load(t, e0);
                stage_write(t, e0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
        if (!VGH_ABLATE(a, 4)) {
            for (int gi = wv; gi < ng; gi += NW) dma_group(gi);
            dma_packs(wv, NW);
        }
        asm volatile("" ::: "memory");  // the counted wait below counts on this order
        if (cw) {  // the basis stream starts behind the tile's LDS-DMA (in-order return) and stays in flight across the barrier
            fetch(B0, 0);
            fetch(B1, UQ);
            if (ng > UQ) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * UQ) : "memory");  // (a second burst exists)
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UQ) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        for (int gi = wv; gi < ng; gi += NW) zero_dead_rows(gi);  // behind this wave's own LDS-DMA of the group
        __syncthreads();
    }
    C3MARK(1);
    if (wv < NCW) {
        // blend: pairs (k, k + 1) per MFMA in ascending k over the live groups: the chain of flame_mfma_kernel for one component
        const float* const sa = s_A + half * AS + j;  // + ((gi * MT + t) * 8 + 2i) * AS
        auto read_a = [&](int gi, float (&A)[MT][4]) {
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const float* const p = sa + (gi * MT + t) * 8 * AS;
#pragma unroll
                for (int i = 0; i < 4; ++i) A[t][i] = (NPW == 0 || j < NH) ? p[2 * i * AS] : 0.0f;
            }
        };
        // the A operands of a group are read while the previous group's four MFMAs run (one wave-uniform branch per group; a read right in front of its MFMA
        // behind a branch per pair cost ~150 cycles per 64-cycle MFMA)
        // quad tiles: two instructions per group (k = 8g + kq, then 8g + 4 + kq), their A operands = rows kq and 4 + kq of the group's tile rows
        const float* const saq = s_A + kq * AS + j;
        auto consume_q = [&](const f32x4_t (&B)[UQ], int g0) {
            if (g0 >= ng || !cw) return;
            const bool hi2 = (kq >> 1) != 0;
            float a0 = (NPW == 0 || j < NH) ? saq[g0 * 8 * AS] : 0.0f, a1 = (NPW == 0 || j < NH) ? saq[(g0 * 8 + 4) * AS] : 0.0f;
#pragma unroll
            for (int u = 0; u < UQ; ++u) {
                float n0 = 0.0f, n1 = 0.0f;
                if (u + 1 < UQ) {  // the next group's operands under this group's two instructions (never past the burst: see consume)
                    const int gn = min(g0 + u + 1, ng - 1);
                    n0 = (NPW == 0 || j < NH) ? saq[gn * 8 * AS] : 0.0f;
                    n1 = (NPW == 0 || j < NH) ? saq[(gn * 8 + 4) * AS] : 0.0f;
                }
                if (g0 + u < ng) {  // wave-uniform
                    accq = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, hi2 ? B[u][1] : B[u][0], accq, 0, 0, 0);
                    accq = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, hi2 ? B[u][3] : B[u][2], accq, 0, 0, 0);
                }
                a0 = n0;
                a1 = n1;
            }
        };
        auto consume = [&](const f32x4_t (&B)[UQ], int g0) {
            if constexpr (Q) {
                consume_q(B, g0);
                return;
            }
            if (g0 >= ng || !cw) return;
            float Ac[MT][4], An[MT][4];
            read_a(g0, Ac);
#pragma unroll
            for (int u = 0; u < UQ; ++u) {
                if (u + 1 < UQ) read_a(min(g0 + u + 1, ng - 1), An);  // never past this burst: the next one may still wait for its rows (pose barrier)
                if (g0 + u < ng) {  // wave-uniform
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ac[t][i], B[u][i], acc[t], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) Ac[t][i] = An[t][i];
            }
        };
        bool synced = NPW == 0;  // fused: the first burst that holds a pose group waits for the prologue waves (pose rows of the tile, head packs)
        auto pose_sync = [&](int gend) {
            if (!synced && gend > c01) {
                __syncthreads();
                synced = true;
            }
        };
        for (int g = 0; g < ng; g += 3 * UQ) {
            fetch(B2, g + 2 * UQ);
            pose_sync(g + UQ);
            consume(B0, g);
            fetch(B0, g + 3 * UQ);
            pose_sync(g + 2 * UQ);
            consume(B1, g + UQ);
            fetch(B1, g + 4 * UQ);
            pose_sync(g + 3 * UQ);
            consume(B2, g + 2 * UQ);
        }
        if (!synced) __syncthreads();
        C3MARK(2);
    } else if constexpr (NPW > 0) {
        const int hh = wv - NCW;
        PrepScratch* const scr = (PrepScratch*)(s_hp + NH * HP_SIZE);
        if (hh < a.n) {
            prep_head(pa, hh, lane, scr[hh], s_A + hh, AS, s_hp + hh * HP_SIZE, bx == 0, false, c01 * 8);  // NB = 8 g2b: the pose rows start a group
        } else {
            for (int q = lane; q < (ng - c01) * 8; q += 64) s_A[(c01 * 8 + q) * AS + hh] = 0.0f;
            for (int e = lane; e < HP_SIZE; e += 64) s_hp[hh * HP_SIZE + e] = 0.0f;
        }
        __syncthreads();  // the pose barrier of the compute waves
    }
    if constexpr (XOWN) {
        if (cw) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_x[((cvg * 3 + cpl) * 16 + r) * 64 + lane] = acc[0][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(&s_cnt[cvg], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (__hip_atomic_load(&s_cnt[cvg], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 3) __builtin_amdgcn_s_sleep(1);  // the group's three planes are in
        }
    } else {
        __syncthreads();  // every wave is done with the coefficient tile: its memory becomes the exchange buffer
        if (wv < NCW) {
            if constexpr (Q) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s_x[(cpl * 4 + r) * 64 + lane] = accq[r];
            } else {
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s_x[(((cvg * MT + t) * 3 + cpl) * 16 + r) * 64 + lane] = acc[t][r];
            }
        }
        __syncthreads();
    }
    C3MARK(3);
    // ---- epilogue: flame_vertex_kernel's statements per (head, vertex), slots dealt round-robin to the waves ----
    // VG = 1: slots dealt round-robin to all waves (every wave's lanes hold the one vertex group's weights); VG > 1: a compute wave takes every third slot of
    // its own vertex group
    const bool vok = v < a.V && (VG == 1 || cw);
    if (VGH_ABLATE(a, 2)) {
        if (a.proj && tid == 0) a.proj[((int64_t)h0 * a.V + v) * 3] = s_x[lane];
        return;
    }
    for (int sl = (VG == 1 ? wv : cpl); sl < (Q ? 4 : VG == 1 || cw ? 16 * MT : 0); sl += (VG == 1 ? NW : 3)) {
        const int t = sl >> 4, r = sl & 15;
        const float* const s_xg = s_x + (cvg * MT + t) * (3 * 16 * 64);
        // 32 x 32 tiles: slot = one accumulator register = heads hlo (lower half-wave) and hlo + 4 (upper) x 32 vertices; quad tiles: register r = heads r, 4 + r,
        // 8 + r, 12 + r (one per 16 lanes) x 16 vertices
        const int hlo = Q ? r : t * 32 + (r & 3) + 8 * (r >> 2);
        if (h0 + hlo >= a.n) continue;           // wave-uniform: no lane has a live head
        const int hh = Q ? 4 * kq + r : hlo + 4 * half;
        const float* const hp = s_hp + min(hh, NH - 1) * HP_SIZE;
        const float px = Q ? s_x[(0 * 4 + r) * 64 + lane] : s_xg[(0 * 16 + r) * 64 + lane], py = Q ? s_x[(1 * 4 + r) * 64 + lane] : s_xg[(1 * 16 + r) * 64 + lane],
                    pz = Q ? s_x[(2 * 4 + r) * 64 + lane] : s_xg[(2 * 16 + r) * 64 + lane];
        float T[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) T[q] = 0.0f;
#pragma unroll
        for (int jn = 0; jn < MAXJ; ++jn)
            if (jn < a.NJ) {  // wave-uniform
                const f32x4_t A0 = *(const f32x4_t*)(hp + HP_A + jn * 12), A1 = *(const f32x4_t*)(hp + HP_A + jn * 12 + 4), A2 = *(const f32x4_t*)(hp + HP_A + jn * 12 + 8);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    T[q] = fmaf(wj[jn], A0[q], T[q]);
                    T[4 + q] = fmaf(wj[jn], A1[q], T[4 + q]);
                    T[8 + q] = fmaf(wj[jn], A2[q], T[8 + q]);
                }
            }
        const float vx = fmaf(T[0], px, fmaf(T[1], py, fmaf(T[2], pz, T[3])));
        const float vy = fmaf(T[4], px, fmaf(T[5], py, fmaf(T[6], pz, T[7])));
        const float vz = fmaf(T[8], px, fmaf(T[9], py, fmaf(T[10], pz, T[11]))) + a.z_offset;
        if (vok && h0 + hh < a.n) {
            const int64_t obase = ((int64_t)(h0 + hh) * a.V + v) * 3;
            if (a.verts) { if (VG > 1) __builtin_nontemporal_store(f32x3_t{vx, vy, vz}, (f32x3_t*)(a.verts + obase)); else *(f32x3_t*)(a.verts + obase) = f32x3_t{vx, vy, vz}; }  // 12 bytes per lane, contiguous across a half-wave
            if (a.proj) {
                const f32x4_t R0 = *(const f32x4_t*)(hp + HP_R), R1 = *(const f32x4_t*)(hp + HP_R + 4), R2 = *(const f32x4_t*)(hp + HP_R + 8), R3 = *(const f32x4_t*)(hp + HP_R + 12);
                // R0 = R[0..3], R1 = R[4..7], R2 = {R[8], s, t0, t1}, R3 = {t2, u0, u1, u2}
                const float sc = R2[1];
                float qx = (R0[0] * vx + R0[1] * vy + R0[2] * vz) * sc + R2[2];
                float qy = (R0[3] * vx + R1[0] * vy + R1[1] * vz) * sc + R2[3];
                float qz = (R1[2] * vx + R1[3] * vy + R2[0] * vz) * sc + R3[0];
                if (a.do_unpad) {  // detector.py:67-69
                    qx = (qx - R3[1]) / R3[3];
                    qy = (qy - R3[2]) / R3[3];
                    qz = qz / R3[3];
                }
                if (VG > 1) __builtin_nontemporal_store(f32x3_t{qx, qy, qz}, (f32x3_t*)(a.proj + obase)); else *(f32x3_t*)(a.proj + obase) = f32x3_t{qx, qy, qz};
            }
        }
    }
    C3MARK(4);
}

template <int NPW, int NHL, int VG = 1, int MT = 1, int Q = 0>
int launch_c3(const VertArgs& va, const PrepArgs& pa, hipStream_t st) {
    constexpr int NH = NPW > 0 ? NPW : Q ? 16 : 32 * MT, NW = 3 * VG + NPW + NHL, VW = Q ? 16 : 32;
    const int ngmax = (va.Kp + 7) / 8;  // live groups <= all groups
    const int g0e = (va.r0_end + 7) >> 3;
    const int g1b = va.r1_end > va.r1_begin ? std::max(va.r1_begin >> 3, g0e) : g0e, g1e = va.r1_end > va.r1_begin ? std::max((va.r1_end + 7) >> 3, g1b) : g0e;
    const int g2b = std::max(va.r2_begin >> 3, g1e), g2e = std::max((va.r2_end + 7) >> 3, g2b);
    const int nrows8 = std::min(ngmax, g0e + (g1e - g1b) + (g2e - g2b)) * 8;
    const int tile = (std::max(nrows8 * MT * (NPW > 0 ? 33 : 32), VG > 1 && MT == 1 ? 0 : VG * MT * 3 * 16 * 64) + 3) & ~3;
    constexpr bool XOWN = VG > 1 && MT == 1;  // the exchange buffer has its own memory (+ a counter per vertex group)
    const size_t lds = ((size_t)tile + (size_t)NH * HP_SIZE + (XOWN ? VG * 3 * 16 * 64 + 16 : 0)) * sizeof(float) + (NPW > 0 ? NPW * sizeof(PrepScratch) : 0);
    static std::atomic<int> attr_done[16];
    int dev = 0;
    VGH_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < 16 && !attr_done[dev].load(std::memory_order_acquire)) {
        VGH_HIP(hipFuncSetAttribute((const void*)flame_c3_kernel<NPW, NHL, VG, MT, Q>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done[dev].store(1, std::memory_order_release);
    }
    if (lds > 160 * 1024) {
        vgh_set_error("flame c3 tiles: %zu bytes of LDS for %d coefficient rows", lds, nrows8);
        return VGH_ERR_INVALID;
    }
    const int vblocks = ((va.V + VW - 1) / VW + VG - 1) / VG, hgroups = NPW > 0 ? 1 : Q ? (va.n + 15) / 16 : (va.n + 32 * MT - 1) / (32 * MT);
    hipLaunchKernelGGL((flame_c3_kernel<NPW, NHL, VG, MT, Q>), dim3((vblocks + 7) / 8 * 8 * hgroups), dim3(NW * 64), lds, st, va, pa);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}

int run_decode_on(vgh_flame* f, const PrepArgs& pa_in, int n, int shape_live, int expr_live, bool detector_mode, float* verts, float* proj, hipStream_t st);

int run_decode(vgh_flame* f, const PrepArgs& pa_in, int n, int shape_live, int expr_live, bool detector_mode, float* verts, float* proj, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (f->used && f->last_stream != st) VGH_HIP(hipStreamWaitEvent(st, f->ev_scratch, 0));  // coef / headpack still belong to the decode queued there
    const int rc = run_decode_on(f, pa_in, n, shape_live, expr_live, detector_mode, verts, proj, st);
    // recorded after every decode (never on a remembered stream handle: its owner may have destroyed it by the next call)
    if (!f->ev_scratch) VGH_HIP(hipEventCreateWithFlags(&f->ev_scratch, hipEventDisableTiming));
    VGH_HIP(hipEventRecord(f->ev_scratch, st));
    f->last_stream = st;
    f->used = true;
    return rc;
}

int run_decode_on(vgh_flame* f, const PrepArgs& pa_in, int n, int shape_live, int expr_live, bool detector_mode, float* verts, float* proj, hipStream_t st) {
    if (pa_in.n_dev && n > f->max_heads) {
        vgh_set_error("flame decode (indirect): capacity %d exceeds max_heads %d", n, f->max_heads);
        return VGH_ERR_INVALID;
    }
    for (int done = 0; done < n; done += f->max_heads) {
        const int m = (n - done < f->max_heads) ? n - done : f->max_heads;
        PrepArgs pa = pa_in;
        pa.n = m;
        pa.live0_end = detector_mode ? shape_live : f->NB;
        pa.live1_begin = detector_mode ? 300 : 0;
        pa.live1_end = detector_mode ? 300 + expr_live : 0;
#ifdef VGH_EXPERIMENTS
        pa.trace = g_prep_trace;
#endif
        if (pa.params) pa.params += (int64_t)done * VGH_NUM_FLAME_PARAMS;
        if (pa.betas) pa.betas += (int64_t)done * f->NB;
        if (pa.pose) pa.pose += (int64_t)done * f->NJ * 3;
        if (pa.unpad) pa.unpad += (int64_t)done * 3;
        if (pa.rot_out) pa.rot_out += (int64_t)done * 9;
        if (pa.joints_out) pa.joints_out += (int64_t)done * f->NJ * 3;
        // matrix-core path from a handful of heads on (or the live count is only known on the device) and the
        // live ranges have even length; tiny direct batches keep the fused VALU kernel (one launch, the whole chip on one head's basis)
        const bool even = ((shape_live | expr_live | f->NB | f->K) & 1) == 0;
        // (measured, profiles/r02_flame_sweep.json: the matrix-core kernel wins from a handful of heads up to a few thousand; at crowd scale the
        //  VALU kernel's 8-heads-per-basis-load reuse is ahead again; with a device-side count the launch is capacity-sized and mostly
        //  exits at once, so the tile count that matters is the live one)
        const int mode = g_flame_mode.load(std::memory_order_relaxed);
        const int amode = (mode == 8 || mode == 9) ? 1 : mode;  // 8 / 9: the automatic choice with the quad tiles forced on / off
        const int npairs = ((detector_mode ? shape_live + expr_live : f->NB) + f->NP + 1) / 2;
        // crowd scale: operands staged through LDS (its k-pair table holds 256 entries: FLAME has 218; a model with more coefficients keeps the other kernels)
        // r04: 32-head blocks of the LDS-staged kernel (mode 5 only).  Hypothesis: the register-fed kernel keeps at most 63 dword loads (8 KB) in flight per wave
        // and runs 77 ns per k at n = 96, so 1-KiB LDS-DMA pieces should free it.  Measured (profiles/r04_flame_sweep.json): n = 96 / 192 with all 400
        // coefficients 63.0 / 64.5 us against 59.4 / 60.2 for the register-fed kernel -- the mid range is bound by its ~30 us of K-independent work (prologue
        // kernel, two launches, head-pack staging, per-head skinning epilogue), not by loads in flight.  Kept selectable (bit-identical), not automatic.
        const bool lds32 = even && npairs <= 248 && mode == 5;
        const bool lds = lds32 || (even && npairs <= 248 && (mode == 3 || mode == 4 || (amode == 1 && !pa.n_dev && m >= kLdsMidHeads)));
        const bool mfma = lds || (even && (mode == 2 || (amode == 1 && (pa.n_dev ? m <= 16384 : (m >= 5 && m < 2048)))));
        // c3 tiles (component-split waves, coefficient tile in LDS; K - NB pose features in one k-group run, NB a multiple of 8 so that the pose rows start a
        // group): mode 6 always with the prologue kernel, mode 7 the same with the fused variant (prologue waves inside the block) up to 8 heads.  Automatic
        // (measured, profiles/r04_flame_sweep.json; us per call, all 400 / L-live / M-live coefficients; old = the kernels above):
        //   n = 1: 22.4 / 18.2 / 16.5 (old 33.4 / 19.2 / 16.2)    n = 8: 27.0 / 21.3 / 18.4 (52.7 / 36.6 / 30.9)    n = 32: 27.7 / 23.0 / 20.2 (58.0 / 42.0 / 36.4)
        //   n = 96: 39.1 / 30.4 / 25.4 (59.0 / 43.3 / 37.3)       n = 128: 49.9 / 35.9 / 28.3 (59.8 / 43.8 / 37.9)  n = 192: 51.3 / 37.2 / 30.0 (60.5 / 44.9 / 38.9)
        //   n = 256: 82.7 / 56.9 / 44.1 (100.4 / 71.9 / 58.9)     n = 512: 122.4 / 83.4 / 63.8 (146.4 / 101.5 / 80.2)  n = 1024: 193.3 / 127.8 / 95.0 (249.7 / 163.3 / 123.4)
        //   n = 2048: 364 / 237 / 173 (387 / 245 / 180)   n = 3072: 536 / 348 / 253 (469 / 298 / 218: the 128-head LDS-staged blocks take over)   n = 8192: 1 400 / 900 / 653 (1 192 / 766 / 569)
        // -> direct batches up to kC3MaxHeads; with a device-side count the launch is capacity-sized and the dead tile rows exit at once: any capacity the
        //    register-fed kernel took (the live count of a detector batch is a few hundred at most: 4-wave blocks up to a capacity of 128, 128-vertex blocks beyond).
        const bool c3_ok = even && f->K - f->NB <= 64 && (f->NB & 7) == 0 && f->basis8;
        const bool c3_auto = amode == 1 && (pa.n_dev ? m <= 16384 : m <= kC3MaxHeads) && !(m <= 2 && npairs < 100);  // (1 - 2 heads of the M set: the VALU kernel)
        const bool c3 = c3_ok && (mode == 6 || mode == 7 || c3_auto);
        const bool c3_fused = c3 && mode != 6 && !pa.n_dev && m <= 8 && (verts || proj);
        const bool fused = c3_fused || (!c3 && !mfma && !pa.n_dev && m <= 256);  // the vertex kernel computes its own heads' prologue
        if (!fused || (!verts && !proj)) {
            constexpr int HB = 16;
            if (m >= (c3 ? 4096 : 512)) {  // (the c3 tiles fetch the rows by LDS-DMA either way: the one-wave-per-block prologue is ~5 us shorter below a few thousand heads)
                            // coalesced coefficient rows (measured: 16 waves per block cost 17 us vs 9.5 us at n = 96, but 92 vs 112 us at n = 8192); the
                            // padded heads of the last block stay inside the scratch (npad is a multiple of 128)
                const size_t lds = (size_t)HB * sizeof(PrepScratch) + (size_t)HB * HP_SIZE * 4 + (size_t)f->Kp * HB * 4;
                static std::atomic<int> attr_done[16];
                int dev = 0;
                VGH_HIP(hipGetDevice(&dev));
                if (dev >= 0 && dev < 16 && !attr_done[dev].load(std::memory_order_acquire)) {
                    VGH_HIP(hipFuncSetAttribute((const void*)flame_prep_multi_kernel<HB>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
                    attr_done[dev].store(1, std::memory_order_release);
                }
                hipLaunchKernelGGL(flame_prep_multi_kernel<HB>, dim3((m + HB - 1) / HB), dim3(HB * 64), lds, st, pa);
            } else {
                hipLaunchKernelGGL(flame_prep_kernel, dim3(m), dim3(64), 0, st, pa);
            }
            VGH_HIP(hipGetLastError());
        }
        if (!verts && !proj) continue;
        VertArgs va;
        va.basis = f->basis;
        va.basis8 = f->basis8;
        va.vt = f->vt;
        va.wts = f->wts;
        va.coef = f->coef;
        va.npad = f->npad;
        va.headpack = f->headpack;
        va.verts = verts ? verts + (int64_t)done * f->V * 3 : nullptr;
        va.proj = proj ? proj + (int64_t)done * f->V * 3 : nullptr;
        va.n = m;
        va.n_dev = pa.n_dev;
        va.V = f->V;
        va.Vp = f->Vp;
        va.NJ = f->NJ;
        va.Kp = f->Kp;
        if (detector_mode) {
            // betas = [shape 300 | expr 100]: only the leading *_live entries of each part can be non-zero
            va.r0_begin = 0;
            va.r0_end = shape_live;
            va.r1_begin = 300;
            va.r1_end = 300 + expr_live;
        } else {
            va.r0_begin = 0;
            va.r0_end = f->NB;
            va.r1_begin = va.r1_end = 0;
        }
        va.r2_begin = f->NB;
        va.r2_end = f->K;
        va.z_offset = detector_mode ? 0.05f : 0.0f;  // MESH_OFFSET_Z, flame.py:34,164
        va.do_unpad = pa.unpad != nullptr;
        va.ablate = 0;
#ifdef VGH_EXPERIMENTS
        if (getenv("VGH_FLAME_ABLATE")) va.ablate = atoi(getenv("VGH_FLAME_ABLATE"));
#endif
        int rc;
        // quad tiles (16 x 16 x 4 MFMAs, r06): measured and NOT adopted -- bit-identical, but no faster at one head (14.8 vs 15.0 us, M set; 19.4 vs 19.7 with all 400
        // coefficients) and 30 - 60 % slower from two heads on (profiles/r06_flame_quad_tiles.txt): a small decode is not its dependent MFMA chain long, it is the
        // basis stream per wave.  Kept in the -DVGH_EXPERIMENTS build (mode 8 forces them up to 128 heads; mode 9 = mode 1 = without them).
#ifdef VGH_EXPERIMENTS
        const bool quad = c3 && mode == 8 && m <= 128;
#else
        const bool quad = false;
#endif
        if (false) {
#ifdef VGH_EXPERIMENTS
        } else if (c3_fused && quad) {
            rc = m <= 1 ? launch_c3<1, 0, 1, 1, 1>(va, pa, st) : m <= 2 ? launch_c3<2, 0, 1, 1, 1>(va, pa, st) : m <= 4 ? launch_c3<4, 0, 1, 1, 1>(va, pa, st) : launch_c3<8, 0, 1, 1, 1>(va, pa, st);
        } else if (c3 && quad && !c3_fused) {
            rc = m <= 32 ? launch_c3<0, 5, 1, 1, 1>(va, pa, st) : launch_c3<0, 1, 1, 1, 1>(va, pa, st);
#endif
        } else if (c3_fused) {
            rc = m <= 1 ? launch_c3<1, 0>(va, pa, st) : m <= 2 ? launch_c3<2, 0>(va, pa, st) : m <= 4 ? launch_c3<4, 0>(va, pa, st) : launch_c3<8, 0>(va, pa, st);
        } else if (c3) {
            // one head tile: 3 + 5 waves; up to 96 heads: 3 + 1 (two blocks per CU; n = 112: 51.1 vs 50.1 us for the 128-vertex blocks); beyond: 128-vertex blocks of 12 compute waves (64-vertex blocks of 6 measured
            // slower than both everywhere)
            if (pa.n_dev) rc = m <= 128 ? launch_c3<0, 1>(va, pa, st) : launch_c3<0, 0, 4>(va, pa, st);  // (capacity, not the live count)
            else if (m <= 96) rc = m <= 32 ? launch_c3<0, 5>(va, pa, st) : launch_c3<0, 1>(va, pa, st);
            else {
                // one block per CU: 128 vertices x 32 heads (12 compute waves), 160 x 32 (15 waves) or 128 x 64 (12 waves, two head tiles' chains per wave on one
                // basis operand) -- whichever needs the least time in ROUNDS of blocks over the CUs, a round costing 1.0 / 1.3 / 1.72 of the first's (measured,
                // all 400 coefficients: n = 256 59.7 vs 82.9 us for 160- vs 128-vertex blocks, 384 100.8 vs 86.0, 512 103.2 vs 118.3 (64-head blocks 134.2),
                // 1 024 189.4 vs 190.2 (196.8), 2 048 361.9 vs 364.3 (323.5))
                const int vgroups = (f->V + 31) / 32, hg = (m + 31) / 32, ncu = f->ncu > 0 ? f->ncu : 256;
                const int r4 = (hg * ((vgroups + 3) / 4) + ncu - 1) / ncu, r5 = (hg * ((vgroups + 4) / 5) + ncu - 1) / ncu, r2 = ((hg + 1) / 2 * ((vgroups + 3) / 4) + ncu - 1) / ncu;
                const int c4 = 100 * r4, c5 = 130 * r5, c2 = 172 * r2;
                rc = c2 < c4 && c2 < c5 ? launch_c3<0, 0, 4, 2>(va, pa, st) : c5 < c4 ? launch_c3<0, 0, 5>(va, pa, st) : launch_c3<0, 0, 4>(va, pa, st);
            }
        } else if (lds) {
            // 128-head blocks at crowd scale; 64-head blocks (twice the blocks) below it and in mode 4
            rc = lds32 ? launch_mfma_lds<1, 32>(va, st) : (mode == 4 || (mode == 1 && m < kLdsMinHeads)) ? launch_mfma_lds<2, 64>(va, st) : launch_mfma_lds<4, 128>(va, st);
        } else if (mfma) {
            if (pa.n_dev || m <= 512)
                rc = launch_mfma<1>(va, st);  // one 32-head tile per wave: twice the waves, two resident per SIMD
            else
                rc = launch_mfma<2>(va, st);
        } else if (fused) {
            if (m <= 4)
                rc = launch_vertex_cfg<1, 64, 1, true>(va, pa, st);   // 79 blocks per head: the whole chip pulls the basis of one head
            else if (m <= 32)
                rc = launch_vertex_cfg<4, 64, 1, true>(va, pa, st);   // 79 x ceil(m/4) blocks
            else if (m <= 96)
                rc = launch_vertex_cfg<4, 64, 4, true>(va, pa, st);   // 20 x ceil(m/4)
            else
                rc = launch_vertex_cfg<8, 256, 4, true>(va, pa, st);  // 5 x ceil(m/8)
        } else if (m <= 24) {
            rc = launch_vertex_cfg<4, 64, 4, false>(va, pa, st);
        } else {
            rc = launch_vertex_cfg<8, 256, 4, false>(va, pa, st);
        }
        if (rc) return rc;
    }
    return VGH_OK;
}

}  // namespace

extern "C" {

int vgh_flame_create(int device, int V, int NB, int NJ, const float* v_template, const float* shapedirs, const float* posedirs, const float* J_regressor,
                     const int32_t* parents, const float* lbs_weights, int max_heads, vgh_flame** out) {
    VGH_REQUIRE(out && v_template && shapedirs && posedirs && J_regressor && parents && lbs_weights, "flame_create: null argument");
    VGH_REQUIRE(NJ >= 1 && NJ <= MAXJ, "flame_create: NJ=%d unsupported (max %d)", NJ, MAXJ);
    VGH_REQUIRE(NB >= 1 && NB <= 1024 && V >= 1, "flame_create: bad NB/V");
    VGH_REQUIRE(9 * (NJ - 1) <= 64, "flame_create: too many pose features");
    VGH_REQUIRE(parents[0] == -1, "flame_create: parents[0] must be -1 (flame.py:91-93)");
    for (int j = 1; j < NJ; ++j) VGH_REQUIRE(parents[j] >= 0 && parents[j] < j, "flame_create: parents must be topologically ordered");
    VGH_HIP(hipSetDevice(device));
    vgh_flame* f = new vgh_flame();
    memset(f, 0, sizeof(*f));
    f->device = device;
    f->V = V;
    f->Vp = (V + 31) / 32 * 32;  // whole 32-vertex MFMA column groups (and 16-byte lanes for the VALU kernel)
    f->NB = NB;
    f->NJ = NJ;
    f->NP = 9 * (NJ - 1);
    f->K = NB + f->NP;
    f->Kp = (f->K + 7) / 8 * 8;
    f->max_heads = max_heads > 0 ? max_heads : 1024;
    VGH_HIP(hipDeviceGetAttribute(&f->ncu, hipDeviceAttributeMultiprocessorCount, device));
    const int Vp = f->Vp, K = f->K;
    std::vector<float> basis((size_t)K * 3 * Vp, 0.0f), vt((size_t)3 * Vp, 0.0f), wts((size_t)NJ * Vp, 0.0f), basis8((size_t)f->Kp * 3 * Vp, 0.0f);
    for (int v = 0; v < V; ++v)
        for (int c = 0; c < 3; ++c) {
            vt[(size_t)c * Vp + v] = v_template[(size_t)v * 3 + c];
            const float* sd = shapedirs + ((size_t)v * 3 + c) * NB;
            for (int l = 0; l < NB; ++l) basis[((size_t)l * 3 + c) * Vp + v] = sd[l];
            for (int pz = 0; pz < f->NP; ++pz) basis[((size_t)(NB + pz) * 3 + c) * Vp + v] = posedirs[(size_t)pz * 3 * V + (size_t)v * 3 + c];
        }
    for (int k = 0; k < K; ++k)
        for (int c = 0; c < 3; ++c)
            for (int v = 0; v < V; ++v) basis8[((((size_t)(k >> 3) * 3 + c) * 2 + (k & 1)) * Vp + v) * 4 + ((k >> 1) & 3)] = basis[((size_t)k * 3 + c) * Vp + v];
    for (int v = 0; v < V; ++v)
        for (int j = 0; j < NJ; ++j) wts[(size_t)j * Vp + v] = lbs_weights[(size_t)v * NJ + j];
    // fold the joint regressor into the shape basis (fp64)
    std::vector<float> J0((size_t)3 * NJ), JS((size_t)NB * MAXJ * 3, 0.0f);
    {
        std::vector<double> accS((size_t)NB);
        for (int j = 0; j < NJ; ++j)
            for (int c = 0; c < 3; ++c) {
                double a0 = 0.0;
                std::fill(accS.begin(), accS.end(), 0.0);
                for (int v = 0; v < V; ++v) {
                    const double w = J_regressor[(size_t)j * V + v];
                    if (w == 0.0) continue;
                    a0 += w * (double)v_template[(size_t)v * 3 + c];
                    const float* sd = shapedirs + ((size_t)v * 3 + c) * NB;
                    for (int l = 0; l < NB; ++l) accS[l] += w * (double)sd[l];
                }
                J0[(size_t)j * 3 + c] = (float)a0;
                for (int l = 0; l < NB; ++l) JS[(size_t)l * (MAXJ * 3) + j * 3 + c] = (float)accS[l];
            }
    }
#define UP(dst, vec)                                                                              \
    VGH_HIP(hipMalloc((void**)&f->dst, (vec).size() * sizeof((vec)[0])));                         \
    VGH_HIP(hipMemcpy(f->dst, (vec).data(), (vec).size() * sizeof((vec)[0]), hipMemcpyHostToDevice))
    UP(basis, basis);
    UP(basis8, basis8);
    UP(vt, vt);
    UP(wts, wts);
    UP(J0, J0);
    UP(JS, JS);
#undef UP
    VGH_HIP(hipMalloc((void**)&f->parents, NJ * sizeof(int32_t)));
    VGH_HIP(hipMemcpy(f->parents, parents, NJ * sizeof(int32_t), hipMemcpyHostToDevice));
    f->npad = (f->max_heads + 127) / 128 * 128;
    VGH_HIP(hipMalloc((void**)&f->coef, (size_t)f->npad * f->Kp * sizeof(float)));
    VGH_HIP(hipMemset(f->coef, 0, (size_t)f->npad * f->Kp * sizeof(float)));
    VGH_HIP(hipMalloc((void**)&f->headpack, (size_t)f->npad * HP_SIZE * sizeof(float)));  // npad rows: the multi-head prologue writes whole 16-head groups
    *out = f;
    return VGH_OK;
}

void vgh_flame_destroy(vgh_flame* f) {
    if (!f) return;
    hipFree(f->basis);
    hipFree(f->basis8);
    hipFree(f->vt);
    hipFree(f->wts);
    hipFree(f->J0);
    hipFree(f->JS);
    hipFree(f->parents);
    hipFree(f->coef);
    hipFree(f->headpack);
    if (f->ev_scratch) hipEventDestroy(f->ev_scratch);
    delete f;
}

int vgh_flame_decode(vgh_flame* f, const float* params_dev, int n, int shape_live, int expr_live, const float* unpad_dev, float* verts_dev,
                     float* rot_dev, float* proj_dev, void* stream) {
    VGH_REQUIRE(f, "flame_decode: null handle");
    VGH_REQUIRE(f->NB == 400 && f->NJ == 5, "flame_decode: the 413-parameter layout needs NB=400, NJ=5 (FLAME_CONSTS, head_info.py:12-21)");
    VGH_REQUIRE(shape_live >= 0 && shape_live <= 300 && expr_live >= 0 && expr_live <= 100, "flame_decode: live counts out of range");
    if (n == 0) return VGH_OK;  // flame.py:186-189 fast path: empty outputs
    VGH_REQUIRE(params_dev, "flame_decode: null params");
    PrepArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.params = params_dev;
    pa.unpad = unpad_dev;
    pa.J0 = f->J0;
    pa.JS = f->JS;
    pa.parents = f->parents;
    pa.coef = f->coef;
    pa.npad = f->npad;
    pa.headpack = f->headpack;
    pa.rot_out = rot_dev;
    pa.NB = f->NB;
    pa.NJ = f->NJ;
    pa.Kp = f->Kp;
    return run_decode(f, pa, n, shape_live, expr_live, true, verts_dev, proj_dev, stream);
}

int vgh_flame_decode_indirect(vgh_flame* f, const float* params_dev, const int32_t* head_row_dev, const int32_t* head_image_dev, const int32_t* n_heads_dev,
                              int capacity, int shape_live, int expr_live, const float* unpad_dev, float* verts_dev, float* rot_dev, float* rpy_dev,
                              float* proj_dev, void* stream) {
    VGH_REQUIRE(f, "flame_decode_indirect: null handle");
    VGH_REQUIRE(f->NB == 400 && f->NJ == 5, "flame_decode_indirect: the 413-parameter layout needs NB=400, NJ=5 (FLAME_CONSTS, head_info.py:12-21)");
    VGH_REQUIRE(shape_live >= 0 && shape_live <= 300 && expr_live >= 0 && expr_live <= 100, "flame_decode_indirect: live counts out of range");
    VGH_REQUIRE(params_dev && head_row_dev && n_heads_dev, "flame_decode_indirect: null argument");
    VGH_REQUIRE(!unpad_dev || head_image_dev, "flame_decode_indirect: unpad rows are indexed by head_image");
    if (capacity <= 0) return VGH_OK;
    PrepArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.params = params_dev;
    pa.unpad = unpad_dev;
    pa.J0 = f->J0;
    pa.JS = f->JS;
    pa.parents = f->parents;
    pa.coef = f->coef;
    pa.npad = f->npad;
    pa.headpack = f->headpack;
    pa.rot_out = rot_dev;
    pa.rpy_out = rpy_dev;
    pa.head_row = head_row_dev;
    pa.head_image = head_image_dev;
    pa.n_dev = n_heads_dev;
    pa.NB = f->NB;
    pa.NJ = f->NJ;
    pa.Kp = f->Kp;
    return run_decode(f, pa, capacity, shape_live, expr_live, true, verts_dev, proj_dev, stream);
}

#ifdef VGH_EXPERIMENTS
int vgh_flame_set_trace(void* dev_buffer) {
    g_prep_trace = (unsigned long long*)dev_buffer;
    return VGH_OK;
}
#endif

int vgh_flame_set_matrix_path(int mode) {
    VGH_REQUIRE(mode >= 0 && mode <= 9, "flame_set_matrix_path: mode %d outside 0..9", mode);
    g_flame_mode.store(mode, std::memory_order_relaxed);
    return VGH_OK;
}

int vgh_flame_lbs(vgh_flame* f, const float* betas_dev, const float* pose_dev, int n, float* verts_dev, float* joints_dev, void* stream) {
    VGH_REQUIRE(f, "flame_lbs: null handle");
    if (n == 0) return VGH_OK;
    VGH_REQUIRE(betas_dev && pose_dev, "flame_lbs: null input");
    PrepArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.betas = betas_dev;
    pa.pose = pose_dev;
    pa.J0 = f->J0;
    pa.JS = f->JS;
    pa.parents = f->parents;
    pa.coef = f->coef;
    pa.npad = f->npad;
    pa.headpack = f->headpack;
    pa.joints_out = joints_dev;
    pa.NB = f->NB;
    pa.NJ = f->NJ;
    pa.Kp = f->Kp;
    return run_decode(f, pa, n, 0, 0, false, verts_dev, nullptr, stream);
}

}  // extern "C"
