// The multiband Gaussian levels (stitcher.py:218, 226) on the matrix cores.
//
// A 1-D blur is a product with a banded Toeplitz matrix, so both passes of one
// level run as v_mfma_f32_32x32x16_f16 products, fused, with the intermediate
// image never leaving registers:
//
//   row pass     Mid[y][x]  = sum_k  In[y][k]   * Tx[k][x]     A = image, B = Toeplitz
//   column pass  Out[y][x]  = sum_k  Ty[y][k]   * Mid[k][x]    A = Toeplitz, B = Mid
//
// The 32 x 32 result of the row pass has its column on the lane and its rows in
// the 16 accumulator registers, which is the B-operand layout of a product that
// sums over its ROW index: the column pass takes it as it is (registers 8s..8s+7
// of a lane are k-step s, in a fixed permuted k order that the Toeplitz operand
// is built to match).
//
// float32 accuracy out of float16 operands: every operand is split into
// hi = f16(v), lo = f16(v - hi) (22-23 significant bits; inputs and taps are
// pre-scaled by powers of two so that lo stays a normal number) and a product is
// three MFMAs, hi*hi + hi*lo + lo*hi, accumulated in float32; the dropped lo*lo
// term is 2^-22 relative.  Measured against the float32 oracle the blurred
// planes agree to ~2e-7.
//
// Work decomposition: a workgroup (4 waves) owns 128 columns of one record,
// channel and level and slides down the rows 32 at a time.  Per step it stages
// one 32-row band of the input (converted to hi/lo float16 once, shared by the
// waves, whose 32-column outputs need overlapping inputs), each wave makes its
// 32 x 32 tile of Mid and adds its contribution to the 2*dmax+1 output tiles
// within reach (live accumulators, rotated), and the tile that just received
// its last contribution is stored.  Tiles are anchored at multiples of 32 in
// patch coordinates, so a pixel's sum does not depend on how the rectangle A was
// cut (column strips, windows): results are bit-identical across decompositions.
//
// Activity: with an interior map only the 32 x 32 tiles that hold a pixel the
// collapse will gather are produced (flags, one byte per tile).
#include <stdlib.h>

#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short short8 __attribute__((ext_vector_type(8)));
typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));   // dword-aligned 16-B load
// Pointers that come out of a pano_patch record are generic to the compiler (flat_load,
// which also counts on the LDS counter); these say "global memory".
#define GLOBAL_AS __attribute__((address_space(1)))
typedef const GLOBAL_AS float *gcf32;
typedef GLOBAL_AS float *gf32;
typedef const GLOBAL_AS int16_t *gci16;
typedef const GLOBAL_AS float4u *gcf32x4;

#define MB_XT 128                   // output columns per workgroup
#define MB_PITCH 264                // halfs per band row (128 + 2*64 + 8): 528 B, conflict-free b128
#define MB_CMAX 4                   // ceil(64 / 16): radius up to 64
#define MB_IN_SCALE 2048.0f         // inputs in [0, 1] -> hi/lo normal in f16
#define MB_TAP_SCALE 256.0f         // taps <= 0.1
#define MB_MID_SCALE (1.0f / 256.0f)            // Mid back to input scale before its split
#define MB_OUT_SCALE (1.0f / (2048.0f * 256.0f))

struct MbLevels {
    const float *w[PANO_MAX_LEVELS];    // first tap of each level
    int ntaps[PANO_MAX_LEVELS];
    int n;
};

__device__ __forceinline__ void split16(float v, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

__device__ __forceinline__ float tap_at(const float *w, int ntaps, int i) {   // w: LDS copy
    return w[(unsigned)i < (unsigned)ntaps ? i : ntaps] * MB_TAP_SCALE;           // w[ntaps] = 0
}

struct MbGeom {
    int gx0, ntx, O0, O1;           // tile grid of rectangle A (patch coordinates / 32)
};

__device__ __forceinline__ MbGeom mb_geom(const pano_patch &p) {
    MbGeom g;
    g.gx0 = (p.ax0 >> 5) << 5;
    g.ntx = ((p.ax0 + p.aw - 1) >> 5) - (p.ax0 >> 5) + 1;
    g.O0 = p.ay0 >> 5;
    g.O1 = (p.ay0 + p.ah - 1) >> 5;
    return g;
}

#define MB_NEED_PAD 8               // slack entries either side of a strip's need flags
#define MB_NEED_MAX 1024            // 32-row tiles of the tallest patch (rows < 32768)

// Column pass of one step.  U = t mod NB fixes which accumulator belongs to which output
// tile, so the roles are compile-time constants.
template <int C, int U>
__device__ __forceinline__ void mb_colpass(f32x16 (&acc)[2 * ((C + 1) / 2) + 1], const f32x16 &mid,
                                           const half8 *s_ty, const uint8_t *need_t,
                                           const int lane) {
    constexpr int DMAX = (C + 1) / 2, NB = 2 * DMAX + 1;
    // Mid as the B operand: registers 8s..8s+7 are k-step s
    half8 m_hi[2], m_lo[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            _Float16 a, b;
            split16(mid[8 * s + j] * MB_MID_SCALE, a, b);
            m_hi[s][j] = a;
            m_lo[s][j] = b;
        }
#pragma unroll
    for (int di = 0; di < NB; ++di) {
        if (!need_t[-(di - DMAX)]) continue;                        // tile t - d, wave-uniform
        constexpr int NB2 = 2 * NB;
        const int slot = (U - (di - DMAX) + NB2) % NB;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const half8 t_hi = s_ty[((di * 2 + s) * 2) * 64 + lane];
            const half8 t_lo = s_ty[((di * 2 + s) * 2 + 1) * 64 + lane];
            acc[slot] = __builtin_amdgcn_mfma_f32_32x32x16_f16(t_hi, m_hi[s], acc[slot], 0, 0, 0);
            acc[slot] = __builtin_amdgcn_mfma_f32_32x32x16_f16(t_lo, m_hi[s], acc[slot], 0, 0, 0);
            acc[slot] = __builtin_amdgcn_mfma_f32_32x32x16_f16(t_hi, m_lo[s], acc[slot], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);           // keep the operand reads next to their use
        }
    }
}

// After step t, output tile t - DMAX has all its contributions: store it, clear the
// accumulator for the tile that takes its place.
template <int C, int U>
__device__ __forceinline__ void mb_store(f32x16 (&acc)[2 * ((C + 1) / 2) + 1],
                                         const uint8_t *need_t, const int t, const int lane,
                                         const pano_patch &p, const gf32 dst, const int px0) {
    constexpr int DMAX = (C + 1) / 2, NB = 2 * DMAX + 1;
    constexpr int slot = (U + DMAX + 1) % NB;
    const int n = lane & 31, h = lane >> 5;
    if (need_t[-DMAX]) {
        const int o = t - DMAX, ax = px0 + n - p.ax0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int ay = 32 * o + (q & 3) + 8 * (q >> 2) + 4 * h - p.ay0;
            if ((unsigned)ay < (unsigned)p.ah && (unsigned)ax < (unsigned)p.aw)
                dst[(size_t)ay * p.apitch + ax] = acc[slot][q] * MB_OUT_SCALE;
        }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[slot][q] = 0.0f;
}

template <int C>
__device__ __forceinline__ void mb_body(const pano_patch &p, const int ch, const int level,
                                        const float *__restrict__ w_global, const int ntaps,
                                        const int16_t *__restrict__ owner_, const int W,
                                        const uint8_t *__restrict__ flags, _Float16 *s_hi,
                                        _Float16 *s_lo, half8 *s_tx, half8 *s_ty,
                                        uint8_t *s_need, uint8_t *s_any, short *s_col,
                                        float *s_w, const int dbg) {
    constexpr int KS = 2 + 2 * C, DMAX = (C + 1) / 2, NB = 2 * DMAX + 1, BW = MB_XT + 32 * C;
    constexpr int GPR = BW / 8, NGRP = 32 * GPR, NPF = (NGRP + 255) / 256;   // groups of 8 columns
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 31, h = lane >> 5;
    const int r = ntaps >> 1;
    const bool alpha = ch == 3;
    const MbGeom g = mb_geom(p);
    const int X0 = g.gx0 + MB_XT * (int)blockIdx.x;     // first output column of the workgroup
    const int px0 = X0 + 32 * wv;                       // ... of this wave
    const int nty = g.O1 - g.O0 + 1;

    // need flags of this wave's tile column, zero-padded so that o - O0 in [-PAD, nty + PAD)
    // needs no range check; any[] = OR over the waves (is band t staged at all)
    {
        const int txg = (px0 - g.gx0) >> 5;
        uint8_t *mine = s_need + wv * (MB_NEED_MAX + 2 * MB_NEED_PAD);
        for (int i = lane; i < nty + 2 * MB_NEED_PAD; i += 64) {
            const int o = i - MB_NEED_PAD;
            bool v = txg < g.ntx && o >= 0 && o < nty;
            if (v && flags) v = flags[p.tiles_off + o * g.ntx + txg] != 0;
            mine[i] = v ? 1 : 0;
        }
    }
    for (int i = tid; i <= ntaps; i += 256) s_w[i] = i < ntaps ? w_global[i] : 0.0f;
    const float *w = s_w;
    __syncthreads();
    for (int bc = tid; bc < 256 + 8; bc += 256) {        // band column -> column of V, or -1
        const int vc = reflect_101(X0 - 16 * C + bc, p.w) - p.vx0;
        s_col[bc] = bc < BW && (unsigned)vc < (unsigned)p.vw ? (short)vc : (short)-1;
    }
    // Toeplitz operand of the row pass, shared: B[k][n] = tap[16 (s - C) + k - n + r]
    for (int idx = tid; idx < KS * 64; idx += 256) {
        const int s = idx >> 6, l = idx & 63, nn = l & 31, hh = l >> 5;
        half8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            _Float16 a, b;
            split16(tap_at(w, ntaps, 16 * (s - C) + 8 * hh + j - nn + r), a, b);
            hi[j] = a;
            lo[j] = b;
        }
        s_tx[(s * 2) * 64 + l] = hi;
        s_tx[(s * 2 + 1) * 64 + l] = lo;
    }
    // Toeplitz operand of the column pass, shared: A[m][k] = tap[32 d + k - m + r] with k in
    // the order the row pass's accumulator registers hold Mid's rows
    for (int idx = tid; idx < NB * 2 * 64; idx += 256) {
        const int ds = idx >> 6, l = idx & 63, mm = l & 31, hh = l >> 5;
        const int di = ds >> 1, s = ds & 1;
        half8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3);
            _Float16 a, b;
            split16(tap_at(w, ntaps, 32 * (di - DMAX) + kk - mm + r), a, b);
            hi[j] = a;
            lo[j] = b;
        }
        s_ty[(ds * 2) * 64 + l] = hi;
        s_ty[(ds * 2 + 1) * 64 + l] = lo;
    }
    __syncthreads();
    const int t_first = g.O0 - DMAX, t_last = g.O1 + DMAX;
    // any[i]: some wave wants Mid tile t = t_first + i, i.e. needs an output tile within DMAX
    for (int i = tid; i <= t_last - t_first + 1; i += 256) {
        bool v = false;
        for (int wq = 0; wq < 4; ++wq)
            for (int d = -DMAX; d <= DMAX; ++d)
                v |= s_need[wq * (MB_NEED_MAX + 2 * MB_NEED_PAD) + (t_first + i - d - g.O0) +
                            MB_NEED_PAD] != 0;
        s_any[i] = v ? 1 : 0;
    }
    __syncthreads();

    f32x16 acc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][q] = 0.0f;

    const gcf32 src = alpha ? nullptr : (gcf32)(p.planes + (size_t)ch * p.vh * p.vpitch);
    const gf32 dst = (gf32)(p.blurred + (size_t)(level * 4 + ch) * p.ah * p.apitch);
    const gci16 owner = (gci16)owner_;
    const int bx0 = X0 - 16 * C;                        // patch column of band column 0
    const uint8_t *need_w = s_need + wv * (MB_NEED_MAX + 2 * MB_NEED_PAD) + MB_NEED_PAD - g.O0;

    // band t (rows 32 t .. 32 t + 31) into registers, 8 consecutive columns per group.
    // s_col[bc] = column of window V behind band column bc, or -1 (beyond V: only zero
    // taps reach it); the same for every step, so the reflection is worked out once.
    float pf[NPF][8];
    const bool vec_ok = p.vw >= 8;                       // uniform
    auto fetch = [&](const int t) {
        // every load is unconditional (clamped address, value selected afterwards): a
        // load under a divergent branch is waited for on the spot, one round trip each
#pragma unroll
        for (int it = 0; it < NPF; ++it) {
            const int grp = tid + 256 * it;
            const int rr = grp / GPR, bc0 = (grp - rr * GPR) * 8;
            const int ry = reflect_101(32 * t + rr, p.h);
            const int vr = ry - p.vy0;
            const bool row_ok = grp < NGRP && (unsigned)vr < (unsigned)p.vh;
            const short8 cm = *(const short8 *)(s_col + (grp < NGRP ? bc0 : 0));
            if (src) {
                const gcf32 q = src + (size_t)(row_ok ? vr : 0) * p.vpitch;
                const bool contig = cm[0] >= 0 && cm[7] - cm[0] == 7;
                if (vec_ok) {
                    const gcf32 qb = q + (contig ? cm[0] : 0);
                    const float4u lo4 = *(gcf32x4)qb;
                    const float4u hi4 = *(gcf32x4)(qb + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        pf[it][j] = row_ok && contig ? lo4[j] : 0.0f;
                        pf[it][4 + j] = row_ok && contig ? hi4[j] : 0.0f;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[it][j] = 0.0f;
                }
                const bool ragged = row_ok && !(contig && vec_ok) &&
                                    (cm[0] | cm[1] | cm[2] | cm[3] | cm[4] | cm[5] | cm[6] | cm[7]) >= 0
                                    ? true
                                    : (row_ok && !(contig && vec_ok) &&
                                       (cm[0] >= 0 || cm[1] >= 0 || cm[2] >= 0 || cm[3] >= 0 ||
                                        cm[4] >= 0 || cm[5] >= 0 || cm[6] >= 0 || cm[7] >= 0));
                if (__any(ragged)) {                     // a patch / window edge inside the group: rare
                    float e[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) e[j] = q[cm[j] >= 0 ? cm[j] : 0];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (ragged) pf[it][j] = cm[j] >= 0 ? e[j] : 0.0f;
                }
            } else {
                const gci16 q = owner + (size_t)(p.y0 + (row_ok ? ry : 0)) * W + p.x0 + p.vx0;
                int16_t e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j] = q[cm[j] >= 0 ? cm[j] : 0];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    pf[it][j] = row_ok && cm[j] >= 0 && e[j] == p.index ? 1.0f : 0.0f;
            }
        }
    };
    auto commit = [&]() {                                // registers -> hi / lo float16 in LDS
#pragma unroll
        for (int it = 0; it < NPF; ++it) {
            const int grp = tid + 256 * it;
            if (grp >= NGRP) break;
            const int rr = grp / GPR, bc0 = (grp - rr * GPR) * 8;
            half8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                _Float16 a, b;
                split16(pf[it][j] * MB_IN_SCALE, a, b);
                hi[j] = a;
                lo[j] = b;
            }
            *(half8 *)(s_hi + rr * MB_PITCH + bc0) = hi;
            *(half8 *)(s_lo + rr * MB_PITCH + bc0) = lo;
        }
    };

    int rem = t_first % NB;
    if (rem < 0) rem += NB;
    int u = rem;                                         // t mod NB
    int t = t_first;
    while (t <= t_last && !s_any[t - t_first]) {         // leading steps nobody wants
        ++t;
        u = u + 1 == NB ? 0 : u + 1;
    }
    if (t <= t_last) fetch(t);
    // steps [done_t, done_end) have had their column pass; the tiles they completed are
    // stored at the start of the NEXT step's arithmetic, not before its barriers: a
    // barrier waits for every outstanding store, which put the full write latency of
    // the output into each step
    int done_t = t, done_end = t, done_u = u;
    auto flush = [&]() {
        for (int tt = done_t; tt < done_end; ++tt) {
            switch (done_u) {                            // wave-uniform
                case 0: mb_store<C, 0>(acc, need_w + tt, tt, lane, p, dst, px0); break;
                case 1: mb_store<C, 1>(acc, need_w + tt, tt, lane, p, dst, px0); break;
                case 2: mb_store<C, 2>(acc, need_w + tt, tt, lane, p, dst, px0); break;
                case 3: mb_store<C, 3 % NB>(acc, need_w + tt, tt, lane, p, dst, px0); break;
                default: mb_store<C, 4 % NB>(acc, need_w + tt, tt, lane, p, dst, px0); break;
            }
            done_u = done_u + 1 == NB ? 0 : done_u + 1;
        }
        done_t = done_end;
    };
    while (t <= t_last) {
        // pf holds band t, which somebody wants
        __syncthreads();                                 // everybody finished reading the band
        if (!(dbg & 4)) commit();
        __syncthreads();
        if (!(dbg & 1)) flush(); else done_t = done_end;
        int tn = t + 1;                                  // next band anybody wants: its loads
        while (tn <= t_last && !s_any[tn - t_first]) ++tn;   // fly during this step's MFMAs
        if (tn <= t_last && !(dbg & 2)) fetch(tn);
        bool want = false;
#pragma unroll
        for (int d = -DMAX; d <= DMAX; ++d) want |= need_w[t - d] != 0;
        if (want && !(dbg & 8)) {
            f32x16 mid;
#pragma unroll
            for (int q = 0; q < 16; ++q) mid[q] = 0.0f;
            const _Float16 *arow = s_hi + n * MB_PITCH + 32 * wv + 8 * h;
            const _Float16 *brow = s_lo + n * MB_PITCH + 32 * wv + 8 * h;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const half8 a_hi = *(const half8 *)(arow + 16 * s);
                const half8 b_hi = s_tx[(s * 2) * 64 + lane];
                const half8 b_lo = s_tx[(s * 2 + 1) * 64 + lane];
                mid = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, mid, 0, 0, 0);
                mid = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, mid, 0, 0, 0);
                if (!alpha) {                            // the sharp mask is exact in float16
                    const half8 a_lo = *(const half8 *)(brow + 16 * s);
                    mid = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, mid, 0, 0, 0);
                }
                if (s & 1) __builtin_amdgcn_sched_barrier(0);
            }
            switch (u) {                                 // wave-uniform
                case 0: mb_colpass<C, 0>(acc, mid, s_ty, need_w + t, lane); break;
                case 1: mb_colpass<C, 1>(acc, mid, s_ty, need_w + t, lane); break;
                case 2: mb_colpass<C, 2>(acc, mid, s_ty, need_w + t, lane); break;
                case 3: mb_colpass<C, 3 % NB>(acc, mid, s_ty, need_w + t, lane); break;
                default: mb_colpass<C, 4 % NB>(acc, mid, s_ty, need_w + t, lane); break;
            }
        }
        done_end = tn;                                   // the gap steps only complete tiles
        u = (u + (tn - t)) % NB;
        t = tn;
    }
    flush();
}

__global__ __launch_bounds__(256, 2) void blur_mfma_kernel(
    const pano_patch *__restrict__ table, MbLevels L, const int16_t *__restrict__ owner, int W,
    const uint8_t *__restrict__ flags, int dbg) {
    __shared__ __attribute__((aligned(16))) _Float16 s_hi[32 * MB_PITCH];
    __shared__ __attribute__((aligned(16))) _Float16 s_lo[32 * MB_PITCH];
    __shared__ half8 s_tx[(2 + 2 * MB_CMAX) * 2 * 64];
    __shared__ half8 s_ty[5 * 2 * 2 * 64];
    __shared__ uint8_t s_need[4 * (MB_NEED_MAX + 2 * MB_NEED_PAD)];
    __shared__ uint8_t s_any[MB_NEED_MAX + 2 * MB_NEED_PAD];
    __shared__ __attribute__((aligned(16))) short s_col[256 + 8];
    __shared__ float s_w[PANO_MAX_TAPS + 1];
    // heavy levels first: z = ((n_levels - 1 - level) * n_records + record) * 4 + channel
    const int ch = blockIdx.z & 3, rest = blockIdx.z >> 2;
    const int nrec = gridDim.z / (4 * L.n);
    const int level = L.n - 1 - rest / nrec, pid = rest % nrec;
    const pano_patch p = table[pid];
    const MbGeom g = mb_geom(p);
    if (p.aw <= 0 || p.ah <= 0 || (int)blockIdx.x * 4 >= g.ntx) return;       // uniform
    const int ntaps = L.ntaps[level], c = ((ntaps >> 1) + 15) >> 4;
#define MB_CALL(CC) \
    mb_body<CC>(p, ch, level, L.w[level], ntaps, owner, W, flags, s_hi, s_lo, s_tx, s_ty, s_need, s_any, s_col, s_w, dbg)
#ifdef MB_ONLY
    MB_CALL(MB_ONLY);
#else
    switch (c) {
        case 0:
        case 1: MB_CALL(1); break;
        case 2: MB_CALL(2); break;
        case 3: MB_CALL(3); break;
        default: MB_CALL(4); break;
    }
#endif
#undef MB_CALL
}

// One thread per 32 x 32 tile of every record: active = some 8 x 8 block under the
// tile (cut to A) is not interior.
__global__ __launch_bounds__(256) void tile_flags32_kernel(const pano_patch *__restrict__ table,
                                                           const uint8_t *__restrict__ interior,
                                                           int W8, uint8_t *__restrict__ flags) {
    const pano_patch p = table[blockIdx.z];
    if (p.aw <= 0 || p.ah <= 0) return;
    const MbGeom g = mb_geom(p);
    const int tx = blockIdx.x * 32 + (threadIdx.x & 31), ty = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (tx >= g.ntx || ty > g.O1 - g.O0) return;
    int x0 = g.gx0 + 32 * tx, y0 = 32 * (g.O0 + ty), x1 = x0 + 32, y1 = y0 + 32;
    x0 = max(x0, p.ax0);
    y0 = max(y0, p.ay0);
    x1 = min(x1, p.ax0 + p.aw);
    y1 = min(y1, p.ay0 + p.ah);
    const int bx0 = (p.x0 + x0) >> 3, bx1 = (p.x0 + x1 - 1) >> 3;
    const int by0 = (p.y0 + y0) >> 3, by1 = (p.y0 + y1 - 1) >> 3;
    bool active = false;
    for (int by = by0; by <= by1; ++by)
        for (int bx = bx0; bx <= bx1; ++bx) active |= interior[(size_t)by * W8 + bx] == 0;
    flags[p.tiles_off + ty * g.ntx + tx] = active ? 1 : 0;
}

// Host side: called by pano_multiband_blur (blur.hip).  taps / ntaps: the caller's
// padded tables (include/pano360.h); `extra[k]` zeros precede level k's first tap
// after the PANO_TAP_LEAD ones.
int pano_launch_blur_mfma(const pano_patch *table, int n, int max_aw, int max_ah,
                          const int16_t *owner, int W, const float *taps, const int *ntaps,
                          int n_blur, const uint8_t *interior, uint8_t *tile_flags,
                          hipStream_t stream) {
    MbLevels L = {};
    L.n = n_blur;
    int rmax = 0;
    for (int k = 0; k < n_blur; ++k) rmax = ntaps[k] / 2 > rmax ? ntaps[k] / 2 : rmax;
    size_t off = 0;
    for (int k = 0; k < n_blur; ++k) {
        L.w[k] = taps + off + PANO_TAP_LEAD + ((rmax - ntaps[k] / 2) & 3);
        L.ntaps[k] = ntaps[k];
        off += (size_t)ntaps[k] + PANO_TAP_PAD;
    }
    const int ntx_max = (max_aw + 62) / 32, nty_max = (max_ah + 62) / 32;
    const uint8_t *flags = nullptr;
    if (interior) {
        dim3 grid(ceil_div(ntx_max, 32), ceil_div(nty_max, 8), n);
        PANO_TIMED(PK_TILE_FLAGS, stream,
                   hipLaunchKernelGGL(tile_flags32_kernel, grid, dim3(256), 0, stream, table,
                                      interior, ceil_div(W, 8), tile_flags));
        PANO_LAUNCH_CHECK("tile_flags32_kernel");
        flags = tile_flags;
    }
    static int dbg = -1;                  // PANO_MFMA_DBG: switch parts off (timing experiments)
    if (dbg < 0) dbg = getenv("PANO_MFMA_DBG") ? atoi(getenv("PANO_MFMA_DBG")) : 0;
    dim3 grid(ceil_div(ntx_max, 4), 1, n * 4 * n_blur);
    PANO_TIMED(PK_BLUR_MFMA, stream,
               hipLaunchKernelGGL(blur_mfma_kernel, grid, dim3(256), 0, stream, table, L, owner, W,
                                  flags, dbg));
    PANO_LAUNCH_CHECK("blur_mfma_kernel");
    return PANO_OK;
}
