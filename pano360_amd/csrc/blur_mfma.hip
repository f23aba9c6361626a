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
// planes agree to 6.6e-7 absolute (the tests' bound is 1e-6; the float32-product vector-ALU
// kernels of blur.hip reach 7.7e-7).
//
// Work decomposition: a workgroup = GROUP levels x 2 adjacent tile columns (MB_XT = 64 output
// columns: a work-list item) of one record and channel; GROUP = 4: eight waves, a wave pair per
// level, one workgroup per CU; GROUP = 2: four waves, two workgroups per CU.  It slides down
// the rows 32 at a time.  Per step it stages one 32-row band of the input (converted to hi/lo
// float16 once, shared by the group's levels and the two tile columns, whose outputs need
// overlapping inputs), each wave makes its 32 x 32 tile of Mid and adds its contribution to
// the 2*dmax+1 output tiles within reach (live accumulators, rotated), and the tile that
// just received its last contribution is stored.  Tiles are anchored at multiples of 32 in
// patch coordinates, so a pixel's sum does not depend on how the rectangle A was
// cut (column strips, windows): results are bit-identical across decompositions.
// blur_lean_kernel (further down) runs groups of up to four (five) levels with apertures of up to
// 97 taps through a leaner instruction stream with two band buffers; this kernel keeps the other
// level groups (two levels per workgroup, apertures above 97 taps, more than one group).
//
// Activity: with an interior map only the 32 x 32 tiles that hold a pixel the
// collapse will gather are produced (flags, one byte per tile).
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"

// The hand-written hazard handling below (s_nop counts behind v_mfma_f32_32x32x16_f16 results
// read by inline-asm vector instructions, vmcnt values that assume loads and stores retire in
// order, SDWA / v_fma_mix encodings) is gfx950's: another target must not compile this silently.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "blur_mfma.hip is written for gfx950 (MI355X): its inline-asm wait states are that chip's"
#endif

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef short shortx4 __attribute__((ext_vector_type(4)));
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

#ifndef MB_SCHED
#define MB_SCHED 1
#endif
#define MB_XT 64                    // output columns per workgroup: two 32-column tiles
// halfs per band row: 64 + 2 * 48 + 8.  336 B = 84 dwords, and 84 = 4 * 21 with 21 odd: the 16
// lanes of a ds_read_b128 group (one band row each) start on 16 different multiples of 4
// banks - conflict-free (as 400 B = 4 * 25 dwords was)
#define MB_PITCH 168
#define MB_PITCH4 200               // groups with a level of more than 97 taps (4 K-steps either side)
// levels per workgroup, one pair of waves (two tile columns) each.  2: four waves, and with the
// tables cut to their non-zero blocks two workgroups fit a CU's LDS - they run out of step, one
// storing and fetching while the other multiplies; 4: eight waves in lockstep, one workgroup per CU
// (template parameter GROUP of the kernel: 2 whenever no level needs more than 3 K-steps either
// side - every level count of the reference's defaults - else 4)
#define MB_THREADS_OF(G) (128 * (G))
// band chunks (4 columns) per thread: the widest band of the form is 64 + 32 CM columns
#define MB_ITS_OF(G) ((32 * ((MB_XT + 32 * ((G) == 2 ? 3 : 4)) / 4) + 128 * (G) - 1) / (128 * (G)))
#define MB_IN_SCALE 2048.0f         // inputs in [0, 1] -> hi/lo normal in f16
#define MB_TAP_SCALE 256.0f         // taps <= 0.1
#define MB_MID_SCALE (1.0f / 256.0f)            // Mid back to input scale before its split
#define MB_OUT_SCALE (1.0f / (2048.0f * 256.0f))
#define MB_NEED_PAD 8               // slack entries either side of a strip's need flags
#define MB_NEED_MAX 256             // 32-row tiles of the tallest patch (rows <= 8192)
#define MB_NEED_LEN (MB_NEED_MAX + 2 * MB_NEED_PAD)
#define MB_WLEN (PANO_MAX_TAPS + 3)

#ifdef MB_STAMP
// phase timers (timing experiments only): [wave 0 | wave 4][phase] summed cycles, and counts
__device__ unsigned long long g_mb_stamps[4][12];
__device__ unsigned long long g_mb_setup[8];   // blur_lean_kernel: cycles of the set-up phases, and workgroups
#define STAMP(k)                                                                         \
    do {                                                                                 \
        if (stamped) {                                                                   \
            const unsigned long long now_ = __builtin_readcyclecounter();               \
            if (lane == 0) atomicAdd(&g_mb_stamps[(wv >> 2) & 1][k], now_ - tlast);      \
            tlast = __builtin_readcyclecounter();                                        \
        }                                                                                \
    } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

struct MbLevels {                       // entries in WORK order: a group = GROUP consecutive entries
    const float *w[PANO_MAX_LEVELS];    // first tap of each level
    int ntaps[PANO_MAX_LEVELS];
    int tab_off[PANO_MAX_LEVELS];       // byte offset of each level's Toeplitz tables
    int out[PANO_MAX_LEVELS];           // the level's index in the blurred planes (stitcher.py:218's lvl)
    int n;
};

__host__ __device__ static inline int mb_c_of(int ntaps) {       // K-steps either side = ceil(r / 16)
    const int c = ((ntaps >> 1) + 15) >> 4;
    return c < 1 ? 1 : c;
}
// which of a group's levels the wave pair q works on: waves w and w + 4 share a SIMD, so
// pairs q and q + 2 do: heavy levels (large radius) sit next to light ones
// Five levels (blur_lean_kernel only): the heaviest three a pair each, the two lightest both on
// the fourth pair (mb_second_level_of_pair), which works on them one after the other.
__host__ __device__ static inline int mb_level_of_pair(int nl, int q) {
    const int order[5][4] = {{0, -1, -1, -1}, {1, 0, -1, -1}, {2, 1, 0, -1}, {3, 2, 0, 1}, {4, 3, 2, 1}};
    return q < 4 ? order[nl - 1][q] : -1;
}
__host__ __device__ static inline int mb_second_level_of_pair(int nl, int q) {
    return nl == 5 && q == 3 ? 0 : -1;
}
// Column-pass Toeplitz blocks (d, s), d = -DMAX .. DMAX, s = 0, 1, hold tap[32 d + k - m + r] for
// k in [16 s, 16 s + 16), m in [0, 32): block (-DMAX, 0) / (DMAX, 1) is all zeros when it lies
// beyond the radius, and is then neither stored nor read (z0 / z1)
__host__ __device__ static inline bool mb_block_zero(int d, int s, int r) {
    return 32 * d + 16 * s - 31 > r || 32 * d + 16 * s + 15 < -r;
}
// dynamic LDS: band (hi, lo), need flags of the two tile columns, any flags, column map,
// then each pair's two Toeplitz tables
__host__ __device__ static inline int mb_pitch_of(int cm) { return cm > 3 ? MB_PITCH4 : MB_PITCH; }
__host__ __device__ static inline int mb_fixed_bytes(int cm) {
    return 2 * 32 * mb_pitch_of(cm) * 2 + 2 * MB_NEED_LEN + MB_NEED_LEN + 4 * MB_NEED_LEN +
           (MB_PITCH4 + 8) * 2;
}
// Row-pass Toeplitz operand of a level, compact: the block of k-step s + 1 is the block of k-step
// s shifted by 16 columns (B_s[k][n] = tap[16 (s - C) + k - n + r]), so all KS blocks are windows
// of ONE table over m = n + 16 (KS - 1 - s), 0 <= m < 16 (KS + 1): [hi, lo][k half h][m] entries of
// eight float16 (k = 8 h .. 8 h + 7).  A lane's entry for k-step s lies 256 (KS - 1 - s) bytes
// past its entry for the last k-step - an immediate offset - and consecutive lanes read
// consecutive 16-byte entries as before.  (KS + 1) KB per level instead of 2 KS.
__host__ __device__ static inline int mb_tx_entries(int ks) { return 16 * (ks + 1); }
__host__ __device__ static inline int mb_tx_bytes(int ks) { return 4 * mb_tx_entries(ks) * 16; }
// index (in half8 entries from the level's table) of lane `lane`'s B operand for k-step s
__host__ __device__ static inline int mb_tx_index(int ks, int lane, int s, int lo) {
    const int nm = mb_tx_entries(ks);
    return (2 * lo + (lane >> 5)) * nm + (lane & 31) + 16 * (ks - 1 - s);
}
__host__ __device__ static inline int mb_table_bytes(int ntaps) {
    const int c = mb_c_of(ntaps), ks = 2 + 2 * c, dmax = (c + 1) / 2, nb = 2 * dmax + 1, r = ntaps >> 1;
    const int blocks = 2 * nb - (mb_block_zero(-dmax, 0, r) ? 1 : 0) - (mb_block_zero(dmax, 1, r) ? 1 : 0);
    return mb_tx_bytes(ks) + blocks * 2 * 1024;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every
// outstanding global load and store of the wave (release fence), which would put the
// latency of the prefetched bands and of the output stores into every step.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

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

// A work item whose bands are not rows of aligned four-float chunks that reflect at most once
// vertically: the pair reaches over the patch's left or right edge (reflected columns there), V
// does not begin and end on a chunk, or the patch is lower than four bands.
__host__ __device__ static inline bool mb_item_edge(const pano_patch &p, int gx0, int tx0, int cm) {
    const int X0 = gx0 + 32 * tx0;
    return X0 - 16 * cm < 0 || X0 + 64 + 16 * cm > p.w || (p.vx0 & 3) != 0 ||
           !(((p.vx0 + p.vw) & 3) == 0 || p.vx0 + p.vw == p.w) || p.h < 128;
}
// (The lean kernels take such items too: ms_body's EDGE form loads their bands element by
// element, at offsets of any number of reflections.  Rounds 3 - 4 left them to a general-path
// kernel beside the lean one: profiles/r06/probes/blur_dead_generations.patch.)

__device__ __forceinline__ MbGeom mb_geom(const pano_patch &p) {
    MbGeom g;
    g.gx0 = (p.ax0 >> 5) << 5;
    g.ntx = ((p.ax0 + p.aw - 1) >> 5) - (p.ax0 >> 5) + 1;
    g.O0 = p.ay0 >> 5;
    g.O1 = (p.ay0 + p.ah - 1) >> 5;
    return g;
}

// Column pass of one step.  Accumulator k belongs to the output tile o with o mod NB == k;
// at step t (u = t mod NB) that is tile t - d with d = (u - k) folded into [-DMAX, DMAX].
// The accumulators are addressed statically; what varies with u - the Toeplitz block and
// the need flag - is an LDS address.
template <int C>
__device__ __forceinline__ void mb_colpass(f32x16 (&acc)[2 * ((C + 1) / 2) + 1], const f32x16 &mid,
                                           const half8 *s_ty, const unsigned inf,
                                           const int lane, const int u, const int r) {
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
    const int z0 = mb_block_zero(-DMAX, 0, r) ? 1 : 0;              // first block not stored
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        int d = u - k;
        d = d > DMAX ? d - NB : (d < -DMAX ? d + NB : d);
        if (!((inf >> (d + 2)) & 1u)) continue;                     // tile t - d, wave-uniform
        const half8 *ty = s_ty + ((d + DMAX) * 2 - z0) * 2 * 64 + lane;   // [block][hi, lo][lane]
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (mb_block_zero(d, s, r)) continue;                   // uniform; such a block is not stored
            const half8 t_hi = ty[(s * 2) * 64];
            const half8 t_lo = ty[(s * 2 + 1) * 64];
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(t_hi, m_hi[s], acc[k], 0, 0, 0);
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(t_lo, m_hi[s], acc[k], 0, 0, 0);
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(t_hi, m_lo[s], acc[k], 0, 0, 0);
            if (MB_SCHED) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// After step t, output tile t - DMAX has all its contributions: store it, clear the
// accumulator for the tile that takes its place.
// Stores go through a buffer descriptor of the (level, channel) plane: the address of a lane
// is one 32-bit byte offset (a 64-bit address per lane and dword made the stores the longest
// phase of a step: the 8 waves of a workgroup push 128 of them through the CU's one path to
// the address unit); the row of each register goes into the instruction's scalar offset.
template <int C>
__device__ __forceinline__ void mb_store(f32x16 (&acc)[2 * ((C + 1) / 2) + 1],
                                         const bool wanted, const int t, const int lane,
                                         const pano_patch &p, const __amdgpu_buffer_rsrc_t plane,
                                         const int px0, const int u) {
    constexpr int DMAX = (C + 1) / 2, NB = 2 * DMAX + 1;
    const int slot = __builtin_amdgcn_readfirstlane((u + DMAX + 1) % NB);
    const int n = lane & 31, h = lane >> 5;
    const int o = t - DMAX, ax = px0 + n - p.ax0;
    const int rowstep = p.apitch * 4;                               // bytes, uniform
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        if (k != slot) continue;                                    // wave-uniform
        if (wanted) {
            // byte offset of register 0's pixel in a VGPR, the row of register q in the scalar
            // offset: no per-store address arithmetic on the vector side
            const int row0 = 32 * o + 4 * h - p.ay0;
            const unsigned base = (unsigned)(row0 * p.apitch + ax) * 4u;
            const bool inside = 32 * o >= p.ay0 && 32 * o + 32 <= p.ay0 + p.ah && px0 >= p.ax0 &&
                                px0 + 32 <= p.ax0 + p.aw;           // wave-uniform
            if (inside) {                                // the whole tile lies in A: no masks
                // Scaled in place, then sixteen stores that share one offset register and write
                // no vector register between them: a store holds its operand registers until
                // the address unit has read them, and a register rewritten for the next store
                // (one value / one address at a time) makes every store wait out that hand-over.
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[k][q] = acc[k][q] * MB_OUT_SCALE;
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[k][q]), plane, base,
                                                          ((q & 3) + 8 * (q >> 2)) * rowstep, 0);
            } else {
                // a tile on the edge of A: the same sixteen stores, a pixel outside A gets an
                // offset beyond the plane and the hardware drops it (no branches, no masks)
                const bool col_in = (unsigned)ax < (unsigned)p.aw;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int dy = (q & 3) + 8 * (q >> 2);
                    const bool in = col_in && (unsigned)(row0 + dy) < (unsigned)p.ah;
                    __builtin_amdgcn_raw_buffer_store_b32(
                        __float_as_uint(acc[k][q] * MB_OUT_SCALE), plane,
                        in ? base + (unsigned)(dy * rowstep) : 0x80000000u, 0, 0);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[k][q] = 0.0f;
    }
}

// What every wave of the workgroup shares.
struct MbShared {
    _Float16 *hi, *lo;              // band, [32][P]
    uint8_t *need;                  // [2][MB_NEED_LEN]: output tile o of tile column 0 / 1 is wanted
    uint8_t *any;                   // [MB_NEED_LEN]: band t is wanted by some wave
    // [2][MB_NEED_LEN], one word per band t and tile column: bit d + 2 = tile t - d is wanted
    // (d = -2 .. 2), bit 8 = band t is wanted by some wave, bit 9 = band t + 1 is
    uint16_t *info;
    short *col;                     // band column -> column of V, or -1
    int CM;                         // largest C of the group: the band reaches 16 CM columns out
    int P;                          // halfs per band row (mb_pitch_of(CM))
    int t_lo, t_hi;                 // bands any wave may want
};

template <int C, int GROUP>
__device__ __forceinline__ void mb_body(const pano_patch &p, const int ch, const int out_level,
                                        const bool live, const half8 *s_tx,
                                        const half8 *s_ty, const MbShared &sh,
                                        const int16_t *__restrict__ owner_, const int W,
                                        const int r, const int tx0) {
    constexpr int KS = 2 + 2 * C, DMAX = (C + 1) / 2, NB = 2 * DMAX + 1;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 31, h = lane >> 5;
    const int tile = wv & 1;
    const bool alpha = ch == 3;
    const MbGeom g = mb_geom(p);
    const int X0 = g.gx0 + 32 * tx0;                    // first output column of the workgroup
    const int px0 = X0 + 32 * tile;                     // ... of this wave
    const int BW = MB_XT + 32 * sh.CM;

    f32x16 acc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][q] = 0.0f;

    // this wave's output plane as a buffer: base and size are wave-uniform
    const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.blurred + (size_t)(out_level * 4 + ch) * p.ah * p.apitch), 0, p.ah * p.apitch * 4,
        0x00020000);
    const gci16 owner = (gci16)owner_;
    // need flags of this wave's tile column; a wave without a level wants nothing
    const uint16_t *info_w = sh.info + (live ? tile * MB_NEED_LEN : 0) + MB_NEED_PAD - g.O0;
    const unsigned keep = live ? 0xffffu : 0xff00u;      // a wave without a level wants no tile
    auto info_at = [&](const int t) -> unsigned {
        return (unsigned)__builtin_amdgcn_readfirstlane(info_w[t]) & keep;
    };
    const int my_lo = g.O0 - DMAX, my_hi = g.O1 + DMAX;             // bands this wave can use

    // Band t (rows 32 t .. 32 t + 31) into registers, four consecutive columns (one 16-byte
    // load) per chunk, chunks dealt out row-major so that a wave's load covers whole rows; two
    // bands in flight.  Colour planes are read through a buffer descriptor: the address is one
    // 32-bit offset, and a chunk that does not exist (row beyond V, column beyond the window)
    // gets an out-of-range offset, for which the hardware returns zeros - exactly the value
    // such samples must have (only zero taps reach them), so there are no masks and no selects.
    // Every load is unconditional: a load under a divergent branch is waited for on the spot,
    // and nothing is computed from a loaded value before the commit.
    constexpr unsigned OOB = 0x80000000u;                // beyond any plane (planes < 2 GiB)
    const int CPR = BW >> 2, NCH = 32 * CPR;             // chunks per band row / per band
    constexpr int ITS = MB_ITS_OF(GROUP), THREADS = MB_THREADS_OF(GROUP);
    float pf[2][ITS][4];
    unsigned pm[2][ITS];                                   // alpha only: bit j = a real sample
    // what does not change from band to band: a chunk's row, its place in the row and its
    // four columns of V (first column, validity bits, contiguous or not)
    int g_rr[ITS];
    unsigned g_ci[ITS];                                    // c0 | bits << 16 | contig << 20 | c4 << 24
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
        const int grp = tid + THREADS * it;
        const int rr = grp / CPR, c4 = grp - rr * CPR;
        const shortx4 cm = *(const shortx4 *)(sh.col + (grp < NCH ? 4 * c4 : 0));
        unsigned bits = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) bits |= (cm[j] >= 0 ? 1u : 0u) << j;
        const bool contig = cm[0] >= 0 && cm[3] - cm[0] == 3;
        g_rr[it] = grp < NCH ? rr : -1;
        g_ci[it] = (unsigned)(unsigned short)cm[0] | bits << 16 | (contig ? 1u << 20 : 0u) |
                   (unsigned)c4 << 24;
    }
    const __amdgpu_buffer_rsrc_t srcb = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(alpha ? p.planes : p.planes + (size_t)ch * p.vh * p.vpitch), 0,
        p.vh * p.vpitch * 4, 0x00020000);
    auto fetch = [&](auto slot_c, const int t) {
        constexpr int S = decltype(slot_c)::value;
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
            const int rr = g_rr[it];
            const unsigned ci = g_ci[it];
            const int ry = reflect_101(32 * t + (rr < 0 ? 0 : rr), p.h);
            const int vr = ry - p.vy0;
            const bool row_ok = rr >= 0 && (unsigned)vr < (unsigned)p.vh;
            const bool contig = (ci >> 20) & 1u;
            const unsigned bits = (ci >> 16) & 15u;
            if (!alpha) {
                const unsigned rowoff = (unsigned)vr * (unsigned)p.vpitch;
                const unsigned voff = row_ok && contig ? (rowoff + (ci & 0xffffu)) * 4u : OOB;
                const uint4 v = __builtin_bit_cast(
                    uint4, __builtin_amdgcn_raw_buffer_load_b128(srcb, voff, 0, 0));
                pf[S][it][0] = __uint_as_float(v.x);
                pf[S][it][1] = __uint_as_float(v.y);
                pf[S][it][2] = __uint_as_float(v.z);
                pf[S][it][3] = __uint_as_float(v.w);
                const bool ragged = row_ok && bits != 0 && !contig;
                if (__any(ragged)) {                     // a patch / window edge inside the chunk: rare
                    const shortx4 cm = *(const shortx4 *)(sh.col + 4 * (ci >> 24));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned o = ragged && cm[j] >= 0 ? (rowoff + (unsigned)cm[j]) * 4u : OOB;
                        const float e = __uint_as_float(
                            __builtin_amdgcn_raw_buffer_load_b32(srcb, o, 0, 0));
                        if (ragged) pf[S][it][j] = e;
                    }
                }
            } else {
                const shortx4 cm = *(const shortx4 *)(sh.col + 4 * (ci >> 24));
                const unsigned rowoff = (unsigned)(p.y0 + (row_ok ? ry : 0)) * (unsigned)W +
                                        (unsigned)(p.x0 + p.vx0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    pf[S][it][j] = __int_as_float(
                        (int)owner[rowoff + (unsigned)(cm[j] >= 0 ? cm[j] : 0)]);
                pm[S][it] = row_ok ? bits : 0u;
            }
        }
    };
    auto commit = [&](auto slot_c) {                     // registers -> hi / lo float16 in LDS
        constexpr int S = decltype(slot_c)::value;
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
            const int rr = g_rr[it];
            if (rr < 0) continue;
            half4 hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = pf[S][it][j];
                if (alpha) {                             // stitcher.py:208; beyond V: zero
                    v = __float_as_int(v) == p.index ? 1.0f : 0.0f;
                    v = (pm[S][it] >> j) & 1u ? v : 0.0f;
                }
                _Float16 a, b;
                split16(v * MB_IN_SCALE, a, b);
                hi[j] = a;
                lo[j] = b;
            }
            const int at = rr * sh.P + 4 * (int)(g_ci[it] >> 24);
            *(half4 *)(sh.hi + at) = hi;
            *(half4 *)(sh.lo + at) = lo;
        }
    };
    auto next_wanted = [&](int t) {                      // first band after t that any wave wants
        ++t;
        while (t <= sh.t_hi && !sh.any[t - sh.t_lo]) ++t;
        return t;
    };

    // Steps [done_t, done_end) have had their column pass; the tiles they completed are
    // stored at the start of the NEXT step's arithmetic, not before its barriers: a
    // barrier waits for every outstanding store.
    int done_t = my_lo, done_end = my_lo;
    auto flush = [&]() {
        int uu = done_t % NB;
        if (uu < 0) uu += NB;
        for (int tt = done_t; tt < done_end; ++tt) {
            mb_store<C>(acc, (info_at(tt) >> (DMAX + 2)) & 1u, tt, lane, p, dst, px0, uu);
            uu = uu + 1 == NB ? 0 : uu + 1;
        }
        done_t = done_end;
    };

    // the sequence of wanted bands t, t1, t2, ...: band t is committed from slot S while t1
    // is in flight in the other slot and t2 is fetched into S
    int t = next_wanted(sh.t_lo - 1), t1 = next_wanted(t);
    // Nothing before the first wanted band has touched an accumulator: the flush starts there.
    // (Starting at my_lo, the first flush - one step AFTER the first band - would clear the slots
    // of the never-computed tiles above, and slot (t - 1 - DMAX) mod NB is also tile t + DMAX's,
    // which band t has just given its first, outermost contribution.)
    if (t > done_t) done_t = done_end = t < my_hi + 1 ? t : my_hi + 1;
    if (t <= sh.t_hi) fetch(std::integral_constant<int, 0>{}, t);
    if (t1 <= sh.t_hi) fetch(std::integral_constant<int, 1>{}, t1);
    unsigned inf = t <= sh.t_hi ? info_at(t) : 0u;       // flags of band t
#ifdef MB_STAMP
    const bool stamped = (wv & 3) == 0 && ch == 0 && (blockIdx.x >> 2) % 7 == 3;
    unsigned long long tlast = __builtin_readcyclecounter();
#endif
    auto step = [&](auto slot_c) {
        STAMP(0);                                        // loop skeleton since the last stamp
        const unsigned inf1 = t1 <= sh.t_hi ? info_at(t1) : 0u;    // one flag word per step
        lds_barrier();                                   // everybody finished reading the band
        STAMP(1);
        commit(slot_c);
        STAMP(2);
        lds_barrier();
        STAMP(3);
        if (live) flush();
        STAMP(4);
        const int t2 = (inf1 >> 9) & 1u ? t1 + 1 : next_wanted(t1);
        if (t2 <= sh.t_hi) fetch(slot_c, t2);
        STAMP(5);
        if (t >= my_lo && t <= my_hi && (inf & (DMAX == 1 ? 0x0eu : 0x1fu))) {
            f32x16 mid;
#pragma unroll
            for (int q = 0; q < 16; ++q) mid[q] = 0.0f;
            const int o = n * sh.P + 16 * (sh.CM - C) + 32 * tile + 8 * h;
            const _Float16 *arow = sh.hi + o, *brow = sh.lo + o;
            // The operands of k-step s + 1 are read from LDS before the products of k-step s
            // are issued: with the reads right in front of each product (what the compiler
            // emits for the plain loop) every product waited out an LDS latency.
            half8 a_hi[2], a_lo[2], b_hi[2], b_lo[2];
            auto operands = [&](const int s, const int buf) {
                a_hi[buf] = *(const half8 *)(arow + 16 * s);
                b_hi[buf] = s_tx[mb_tx_index(KS, lane, s, 0)];
                b_lo[buf] = s_tx[mb_tx_index(KS, lane, s, 1)];
                if (!alpha) a_lo[buf] = *(const half8 *)(brow + 16 * s);
            };
            operands(0, 0);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int cur = s & 1;
                if (s + 1 < KS) operands(s + 1, cur ^ 1);
                mid = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[cur], b_hi[cur], mid, 0, 0, 0);
                mid = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[cur], b_lo[cur], mid, 0, 0, 0);
                if (!alpha)                              // the sharp mask is exact in float16
                    mid = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo[cur], b_hi[cur], mid, 0, 0, 0);
#if MB_SCHED
                // next k-step's reads (DS read, mask 0x100) first, then this one's products
                // (MFMA, mask 0x008)
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
#endif
            }
            STAMP(6);                                    // row pass
            int u = t % NB;
            if (u < 0) u += NB;
            mb_colpass<C>(acc, mid, s_ty, inf, lane, u, r);
            STAMP(7);                                    // column pass
        }
#ifdef MB_STAMP
        if (stamped && lane == 0) atomicAdd(&g_mb_stamps[(wv >> 2) & 1][11], 1ull);
#endif
        // bands up to the next wanted one only complete tiles
        const int upto = t1 < my_hi + 1 ? t1 : my_hi + 1;
        if (upto > done_end) done_end = upto;
        t = t1;
        t1 = t2;
        inf = inf1;
    };
    while (t <= sh.t_hi) {
        step(std::integral_constant<int, 0>{});
        if (t > sh.t_hi) break;
        step(std::integral_constant<int, 1>{});
    }
    if (live) {
        if (my_hi + 1 > done_end) done_end = my_hi + 1;
        flush();
    }
}

// The two Toeplitz operands of every level, in the lane order the MFMAs read them
// (built once per tap set; the workgroups copy their levels' tables into LDS):
//   row pass     B[k][n] = tap[16 (s - C) + k - n + r],      s = 0 .. KS-1
//   column pass  A[m][k] = tap[32 d + k - m + r], d = -DMAX .. DMAX, k in the order the
//                row pass's accumulator registers hold Mid's rows
// Layout per level: the compact row-pass table (mb_tx_index), then [d][s][hi, lo][lane] half8.
__global__ __launch_bounds__(128) void mb_tables_kernel(MbLevels L, unsigned char *out) {
    __shared__ float w[MB_WLEN];
    const int level = blockIdx.x, ntaps = L.ntaps[level], r = ntaps >> 1;
    const int c = mb_c_of(ntaps), KS = 2 + 2 * c, DMAX = (c + 1) / 2, NB = 2 * DMAX + 1;
    for (int i = threadIdx.x; i <= ntaps; i += 128) w[i] = i < ntaps ? L.w[level][i] : 0.0f;
    __syncthreads();
    half8 *tx = (half8 *)(out + L.tab_off[level]), *ty = (half8 *)(out + L.tab_off[level] + mb_tx_bytes(KS));
    const int NM = mb_tx_entries(KS);
    for (int idx = threadIdx.x; idx < 2 * NM; idx += 128) {
        const int hh = idx / NM, m = idx - hh * NM;         // m = n + 16 (KS - 1 - s)
        half8 hi, lo;
        for (int j = 0; j < 8; ++j) {
            _Float16 a, b;
            split16(tap_at(w, ntaps, 16 * (KS - 1 - c) + 8 * hh + j - m + r), a, b);
            hi[j] = a;
            lo[j] = b;
        }
        tx[hh * NM + m] = hi;
        tx[(2 + hh) * NM + m] = lo;
    }
    const int z0 = mb_block_zero(-DMAX, 0, r) ? 1 : 0;
    for (int idx = threadIdx.x; idx < NB * 2 * 64; idx += 128) {
        const int full = idx >> 6, l = idx & 63, mm = l & 31, hh = l >> 5;
        const int di = full >> 1, s = full & 1;
        if (mb_block_zero(di - DMAX, s, r)) continue;               // not stored
        const int ds = full - z0;
        half8 hi, lo;
        for (int j = 0; j < 8; ++j) {
            const int kk = 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3);
            _Float16 a, b;
            split16(tap_at(w, ntaps, 32 * (di - DMAX) + kk - mm + r), a, b);
            hi[j] = a;
            lo[j] = b;
        }
        ty[(ds * 2) * 64 + l] = hi;
        ty[(ds * 2 + 1) * 64 + l] = lo;
    }
}

template <int GROUP>
__device__ __forceinline__ void mb_general(
    const pano_patch *__restrict__ table, const MbLevels &L, const unsigned char *__restrict__ tables,
    const int16_t *__restrict__ owner, int W, const uint8_t *__restrict__ flags,
    const int2 *__restrict__ items, unsigned char *smem, const int work) {
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    // the items - (record, first tile column of a pair) - are sorted by decreasing length
    // (mb_sort_kernel), so the hardware's in-order dispatch starts the long strips first
    // Workgroup ids go round the 8 XCDs (id mod 8), each with its own L2.  The level groups of
    // one (item, channel) read the same band: they get ids 8 apart - the same XCD, dispatched
    // within a few dozen workgroups of each other - so that the second and third group's
    // fetches find part of the band in that L2 (config 5, three groups: 15.2 -> 11.9 GB of reads
    // per launch, measured; one pass over the windows is 4.8 GB; the time did not change).
    const int ngroups = (L.n + GROUP - 1) / GROUP;
    const int per = 8 * ngroups, blk = work / per, within = work - blk * per;
    const int grp = within >> 3, pair = blk * 8 + (within & 7);
    const int ch = pair & 3, slot = pair >> 2;
    const int2 item = items[slot];
    if (item.x < 0) return;                                                     // uniform
    const int pid = item.x & 0xffff, tx0 = item.x >> 16;
    const pano_patch p = table[pid];
    const MbGeom g = mb_geom(p);
    // a vertical segment of the item (mb_sort_kernel): tile rows [o_begin, o_end) of the record
    const int n_seg = item.y >> 16, seg = (item.y >> 12) & 15, nty_all = g.O1 - g.O0 + 1;
    const int o_begin = n_seg > 1 ? nty_all * seg / n_seg : 0;
    const int o_end = n_seg > 1 ? nty_all * (seg + 1) / n_seg : nty_all;
    const int l0 = GROUP * grp, nl = L.n - l0 < GROUP ? L.n - l0 : GROUP;
    const int q = __builtin_amdgcn_readfirstlane(wv >> 1);
    const int lv = mb_level_of_pair(nl, q);
    const bool live = lv >= 0;
    const int level = l0 + (live ? lv : 0);
    const int ntaps = L.ntaps[level], c = mb_c_of(ntaps);

    // the group's levels: its reach CM (largest C), and where each pair's tables start
    MbShared sh;
    sh.CM = 1;
    int dmax_of[GROUP], rel = 0, my_tx = 0, my_ty = 0;
    for (int k = 0; k < GROUP; ++k) {
        const int lk = mb_level_of_pair(nl, k);
        dmax_of[k] = -1;
        if (lk < 0) continue;
        const int ck = mb_c_of(L.ntaps[l0 + lk]);
        sh.CM = ck > sh.CM ? ck : sh.CM;
        dmax_of[k] = (ck + 1) / 2;
        if (k == q) {
            my_tx = rel;
            my_ty = rel + mb_tx_bytes(2 + 2 * ck);
        }
        rel += mb_table_bytes(L.ntaps[l0 + lk]);
    }
    sh.P = mb_pitch_of(sh.CM);
    sh.hi = (_Float16 *)smem;
    sh.lo = sh.hi + 32 * sh.P;
    sh.need = (uint8_t *)(sh.lo + 32 * sh.P);
    sh.any = sh.need + 2 * MB_NEED_LEN;
    sh.info = (uint16_t *)(sh.any + MB_NEED_LEN);
    sh.col = (short *)(sh.info + 2 * MB_NEED_LEN);
    my_tx += mb_fixed_bytes(sh.CM);
    my_ty += mb_fixed_bytes(sh.CM);
    const int dmaxm = (sh.CM + 1) / 2;
    sh.t_lo = g.O0 - dmaxm;
    sh.t_hi = g.O1 + dmaxm;

    // this pair's Toeplitz tables (row pass, then column pass), column map, need flags
    if (live) {
        const uint4 *from = (const uint4 *)(tables + L.tab_off[level]);
        uint4 *to = (uint4 *)(smem + my_tx);
        const int n16 = mb_table_bytes(ntaps) >> 4;
        for (int i = tid & 127; i < n16; i += 128) to[i] = from[i];
    }
    const int X0 = g.gx0 + 32 * tx0, BW = MB_XT + 32 * sh.CM;
    for (int bc = tid; bc < MB_PITCH4 + 8; bc += MB_THREADS_OF(GROUP)) {
        const int vc = reflect_101(X0 - 16 * sh.CM + bc, p.w) - p.vx0;
        sh.col[bc] = bc < BW && (unsigned)vc < (unsigned)p.vw ? (short)vc : (short)-1;
    }
    const int nty = g.O1 - g.O0 + 1;
    if (wv < 2) {
        const int txg = ((X0 - g.gx0) >> 5) + wv;
        for (int i = lane; i < nty + 2 * MB_NEED_PAD; i += 64) {
            const int o = i - MB_NEED_PAD;
            bool v = txg < g.ntx && o >= o_begin && o < o_end;
            if (v && flags) v = flags[p.tiles_off + o * g.ntx + txg] != 0;
            sh.need[wv * MB_NEED_LEN + i] = v ? 1 : 0;
        }
    }
    __syncthreads();
    // any[i]: band t_lo + i is within reach (its own DMAX) of a wanted tile of some pair
    for (int i = tid; i <= sh.t_hi - sh.t_lo; i += MB_THREADS_OF(GROUP)) {
        const int o = sh.t_lo + i - g.O0 + MB_NEED_PAD;         // index of tile t into need[]
        bool v = false;
        for (int k = 0; k < GROUP; ++k)
            for (int d = -dmax_of[k]; d <= dmax_of[k]; ++d)
                v |= (sh.need[o - d] | sh.need[MB_NEED_LEN + o - d]) != 0;
        sh.any[i] = v ? 1 : 0;
    }
    __syncthreads();
    for (int i = tid; i < 2 * (sh.t_hi - sh.t_lo + 1); i += MB_THREADS_OF(GROUP)) {
        const int col = i & 1, k = i >> 1, t = sh.t_lo + k;
        const int o = t - g.O0 + MB_NEED_PAD;                   // index of tile t into need[]
        unsigned v = 0;
        for (int d = -2; d <= 2; ++d) v |= (sh.need[col * MB_NEED_LEN + o - d] ? 1u : 0u) << (d + 2);
        v |= sh.any[k] ? 0x100u : 0u;
        v |= t < sh.t_hi && sh.any[k + 1] ? 0x200u : 0u;
        sh.info[col * MB_NEED_LEN + o] = (uint16_t)v;
    }
    __syncthreads();
    const half8 *s_tx = (const half8 *)(smem + my_tx), *s_ty = (const half8 *)(smem + my_ty);
    const int out_level = L.out[level];
    switch (c) {                                         // wave-uniform
        case 1: mb_body<1, GROUP>(p, ch, out_level, live, s_tx, s_ty, sh, owner, W, ntaps >> 1, tx0); break;
        case 2: mb_body<2, GROUP>(p, ch, out_level, live, s_tx, s_ty, sh, owner, W, ntaps >> 1, tx0); break;
        case 3: mb_body<3, GROUP>(p, ch, out_level, live, s_tx, s_ty, sh, owner, W, ntaps >> 1, tx0); break;
        default: mb_body<4, GROUP>(p, ch, out_level, live, s_tx, s_ty, sh, owner, W, ntaps >> 1, tx0); break;
    }
}

template <int GROUP>
__global__ __launch_bounds__(MB_THREADS_OF(GROUP), 4 / GROUP) void blur_mfma_kernel(
    const pano_patch *__restrict__ table, MbLevels L, const unsigned char *__restrict__ tables,
    const int16_t *__restrict__ owner, int W, const uint8_t *__restrict__ flags,
    const int2 *__restrict__ items) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    mb_general<GROUP>(table, L, tables, owner, W, flags, items, smem, (int)blockIdx.x);
}


// =====================================================================================
// The same blur with a lean instruction stream, for the work items it fits.
//
// blur_mfma_kernel above handles every case inside its step - bands whose rows reflect
// several times, chunks that straddle the window's or the patch's edge, tiles wanted or not
// one by one, a variable number of finished tiles per step - and pays for that generality on
// every step: ~790 instructions per wave and 32-row step for 34 matrix products (counters of
// rounds 2 and 3), 11 cycles each; the ablations of profiles/r03/notes.md show no unit and no
// wait to blame, only the length of that stream.  This kernel runs the same arithmetic, tile
// for tile and product for product (results are bit-identical, tests compare them), with all of
// that static or turned into data (ms_body; its EDGE form takes the items with reflected
// columns, unaligned windows and low patches, mb_item_edge):
//  * the bands an item steps through are listed once, compacted, in LDS (band index and the
//    two tile columns' wanted bits in one word): no search for the next wanted band, no
//    flag words read and re-read inside the step;
//  * a band's chunks are whole (window V ends on multiples of 4 columns, include/pano360.h)
//    and its rows reflect at most once: a chunk's address is its row's offset plus a constant,
//    which threads stage which chunk is wave-uniform;
//  * the column pass addresses the accumulators statically: inside a run of consecutive bands
//    a tile's accumulator is its position in the run mod (2 DMAX + 1) and the run loop is
//    unrolled that many times; a tile's first contribution takes a zero C operand instead of
//    a cleared accumulator (unwanted tiles are skipped by wave-uniform tests, as above);
//  * exactly one tile per step can finish, and it is stored at the start of the next step
//    (behind that step's barriers, in front of its fetch): no bookkeeping of finished ranges.
#ifndef MB_LEAN
#define MB_LEAN 1
#endif
// 1: the lean path keeps TWO band buffers in LDS and one barrier per step: while a wave
// multiplies band t it stages its share of band t + 1 into the other buffer and fetches band
// t + 2, in the shadow of its own (and its SIMD partner's) matrix products
#ifndef ML_OVERLAP
#define ML_OVERLAP 1
#endif

// Tile o of this wave's column (accumulator `a`) to its plane.
// ALWAYS sixteen store instructions: with `wanted` false every one of them gets an offset
// beyond the plane and the hardware drops it.  (A step then issues the same number of memory
// operations on every path, and the compiler can count: it waits for a band's loads with
// s_waitcnt vmcnt(16), leaving the stores in flight, instead of vmcnt(0).)
__device__ __forceinline__ void ml_store(const f32x16 &a, const int o, const int lane,
                                         const pano_patch &p, const __amdgpu_buffer_rsrc_t plane,
                                         const int px0, const bool wanted) {
    const int n = lane & 31, h = lane >> 5;
    const int ax = px0 + n - p.ax0, row0 = 32 * o + 4 * h - p.ay0;
    const int rowstep = p.apitch * 4;
    const unsigned base = wanted ? (unsigned)(row0 * p.apitch + ax) * 4u : 0x80000000u;
    const bool inside = !wanted || (32 * o >= p.ay0 && 32 * o + 32 <= p.ay0 + p.ah &&
                                    px0 >= p.ax0 && px0 + 32 <= p.ax0 + p.aw);     // wave-uniform
    if (inside) {
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = a[q] * MB_OUT_SCALE;
#pragma unroll
        for (int q = 0; q < 16; ++q)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[q]), plane, base,
                                                  ((q & 3) + 8 * (q >> 2)) * rowstep, 0);
    } else {
        const bool col_in = (unsigned)ax < (unsigned)p.aw;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int dy = (q & 3) + 8 * (q >> 2);
            const bool in = col_in && (unsigned)(row0 + dy) < (unsigned)p.ah;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(a[q] * MB_OUT_SCALE), plane,
                                                  in ? base + (unsigned)(dy * rowstep) : 0x80000000u,
                                                  0, 0);
        }
    }
}

// =====================================================================================
// Round 4: the lean step as ONE basic block.  (ml_body, round 3's step, is gone from this file:
// profiles/r06/probes/blur_dead_generations.patch.)
//
// ml_body's step is a chain of phases - fetch, store, row pass, split, column pass, commit -
// each behind wave-uniform branches (chunk exists? tile inside A? tile wanted? band wanted?),
// so every phase is a basic block of its own and the compiler cannot move one phase's
// vector instructions under another phase's matrix products: the two waves of a SIMD, in
// lockstep behind the one barrier per step, issue their vector phases together (matrix pipe
// idle) and their product chains together (vector issue idle).  A 32x32x16 product holds the
// SIMD's vector issue for 8 of its 32 cycles (MI355X_MICROARCH.md): 24 cycles of every
// product are free for other instructions of the SAME wave if they stand next to it in
// program order.  Here the irregular cases are turned into DATA, so that a step is a short
// scalar prologue and one straight-line block the scheduler can interleave:
//   * every thread stages exactly 2.5 chunks (two 16-byte loads and one 8-byte load; a slot a
//     narrower band does not fill repeats the thread's first chunk): no per-slot "chunk
//     exists" branches;
//   * a band whose 32 rows lie inside the patch and inside V (89 % of config 3's) gets its
//     load offsets by ONE add per slot (row-independent part + 32 t pitch); the others take a
//     branch in the prologue that computes the same three offsets the long way;
//   * the finished tile is always stored by sixteen stores whose per-lane base is beyond the
//     plane for lanes outside A's columns and for tiles nobody wants; only a tile cut by A's
//     top or bottom row takes the masked path, in the prologue;
//   * the column pass computes every tile within reach, wanted or not (85 % are; an unwanted
//     tile's accumulator is never stored); a wave that wants nothing of a band runs both
//     passes all the same (skipping them, -DMS_SKIP_IDLE, put a branch in front of the block
//     and measured slower; the switch is kept for A/B only);
//   * f32 -> (hi, lo) float16 is two instructions per value (v_fma_mixlo/hi_f16: hi =
//     f16(v s), lo = f16(fma(v, s, -hi)), the same bits as ml_body's (v s) - hi: the
//     difference is exact in float32), written into packed halves directly;
//   * the column pass runs k-half-major (all tiles' first half-blocks, then their second):
//     the second half of Mid is split under the first half's products.  Each accumulator
//     still receives its products in ml_body's order: results are bit-identical.
// (hi, lo) float16 pairs of floats scaled by the power of two s: hi = f16(v s), lo = f16(v s - hi),
// two instructions per value (the compiler's own selection takes 3.5: it forms hi a second time
// for the packed word).  v_fma_mix{lo,hi}_f16 writes one half of its destination and keeps the
// other; a half written by one instruction is read two instructions later at the earliest
// (forwarding of partial writes), and the trailing s_nop covers the reader that follows.
// AFTER_MFMA: the inputs are the result of a matrix product that may still be in flight - the
// compiler does not see into the statement, so the wait states it would insert are written out.
template <bool AFTER_MFMA>
__device__ __forceinline__ void ms_split4(const float a0, const float b0, const float a1,
                                          const float b1, const float s, unsigned &h0,
                                          unsigned &l0, unsigned &h1, unsigned &l1) {
    if (AFTER_MFMA)
        asm("s_nop 11\n\t"
            "v_fma_mixlo_f16 %0, %4, %8, 0\n\t"
            "v_fma_mixlo_f16 %2, %6, %8, 0\n\t"
            "v_fma_mixhi_f16 %0, %5, %8, 0\n\t"
            "v_fma_mixhi_f16 %2, %7, %8, 0\n\t"
            "v_fma_mixlo_f16 %1, %4, %8, -%0 op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixlo_f16 %3, %6, %8, -%2 op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %1, %5, %8, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %3, %7, %8, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
            "s_nop 0"
            : "=&v"(h0), "=&v"(l0), "=&v"(h1), "=&v"(l1)
            : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "s"(s));
    else
    asm("v_fma_mixlo_f16 %0, %4, %8, 0\n\t"
        "v_fma_mixlo_f16 %2, %6, %8, 0\n\t"
        "v_fma_mixhi_f16 %0, %5, %8, 0\n\t"
        "v_fma_mixhi_f16 %2, %7, %8, 0\n\t"
        "v_fma_mixlo_f16 %1, %4, %8, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %3, %6, %8, -%2 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %5, %8, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %3, %7, %8, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "s_nop 0"
        : "=&v"(h0), "=&v"(l0), "=&v"(h1), "=&v"(l1)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "s"(s));
}
__device__ __forceinline__ void ms_split2(const float a, const float b, const float s, unsigned &h,
                                          unsigned &l) {
    asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
        "s_nop 0\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "s_nop 0"
        : "=&v"(h), "=&v"(l)
        : "v"(a), "v"(b), "s"(s));
}

// CB != 0: the wave pair works on TWO levels, C and then CB, from the same staged band (their
// accumulators must rotate alike: equal DMAX) - five levels on the four wave pairs of a
// workgroup, so that a sixth pyramid level (stitcher.py:186, n_levels = 6) does not cost a second
// launch that stages every band again.
// EDGE: the item's bands hold reflected columns (the pair reaches over the patch's left or right
// edge, BORDER_REFLECT_101 there): a chunk is then not four consecutive floats of a row, and every
// ELEMENT of the thread's 2.5 chunks gets its own load (ten dword loads instead of two 16-byte
// and one 8-byte load; the mask channel loads ten shorts either way) at an offset of its own,
// computed once: the reflected column, or "beyond the plane" outside V.  Everything behind the
// loads is the same code; 2 % of the steps run this form.
template <int C, bool SHARP, int CB = 0, bool EDGE = false>
__device__ __forceinline__ void ms_body(const pano_patch &p, const int ch, const int out_level,
                                        const bool live, const half8 *s_tx, const half8 *s_ty,
                                        const MbShared &sh, const uint32_t *list, const int nlist,
                                        const int16_t *__restrict__ owner_, const int W,
                                        const int tx0, const int second, const int out_level_b = 0,
                                        const half8 *s_tx_b = nullptr,
                                        const half8 *s_ty_b = nullptr) {
    constexpr int KS = 2 + 2 * C, DMAX = (C + 1) / 2, NB = 2 * DMAX + 1, Z = C & 1;
    constexpr int KS_B = 2 + 2 * CB, Z_B = CB & 1;
    static_assert(CB == 0 || 2 * ((CB + 1) / 2) + 1 == NB, "two levels of a wave: equal DMAX");
    constexpr unsigned OOB = 0x80000000u;
    constexpr int NV = EDGE ? 10 : 3;                    // load offsets per thread and band
    // the slot of offset entry e, and the entry of element j of slot k
    auto slot_of = [](const int e) { return EDGE ? (e < 4 ? 0 : (e < 8 ? 1 : 2)) : e; };
    auto entry_of = [](const int k, const int j) { return EDGE ? 4 * k + j : k; };
    typedef unsigned uint4v __attribute__((ext_vector_type(4)));
    typedef unsigned uint2v __attribute__((ext_vector_type(2)));
    typedef int int4v __attribute__((ext_vector_type(4)));
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    const int bstride = __builtin_amdgcn_readfirstlane(second);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 31, h = lane >> 5;
    const int tile = __builtin_amdgcn_readfirstlane(wv & 1);
    const MbGeom g = mb_geom(p);
    const int X0 = g.gx0 + 32 * tx0, px0 = X0 + 32 * tile;
    const int P = sh.P, CM = sh.CM;
    const int BW = MB_XT + 32 * CM, CPR = BW >> 2, NCH = 32 * CPR;
    const int my_lo = g.O0 - DMAX, my_hi = g.O1 + DMAX;

    f32x16 acc[NB], acc_b[CB ? NB : 1];
    const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.blurred + (size_t)(out_level * 4 + ch) * p.ah * p.apitch), 0, p.ah * p.apitch * 4,
        0x00020000);
    const __amdgpu_buffer_rsrc_t dst_b = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.blurred + (size_t)((CB ? out_level_b : out_level) * 4 + ch) * p.ah * p.apitch), 0,
        p.ah * p.apitch * 4, 0x00020000);
    (void)KS_B;

    // ---- this thread's 2.5 chunks ------------------------------------------------------
    // slot 0, 1: chunk tid, 512 + tid (four columns, 16 bytes); slot 2: half (tid & 1) of chunk
    // 1024 + tid / 2 (two columns).  c_off: the row-independent part of the load offset (colour:
    // bytes into the plane; mask: bytes into the owner map), c_lds: where the hi halfs go.
    // A slot beyond the band (a group whose reach is less than 3 K-steps has fewer chunks)
    // repeats slot 0's chunk: the same bytes land on the same LDS halfs twice, no traffic
    // beyond the L1.
    int c_rr[3], c_lds[3];
    unsigned c_off[NV];
    bool c_ok[NV];
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        int c = it < 2 ? tid + 512 * it : 1024 + (tid >> 1);
        int half = it == 2 ? (tid & 1) : 0;
        if (c >= NCH) {
            c = tid;
            half = 0;
        }
        const int rr = c / CPR, c4 = c - rr * CPR;
        c_rr[it] = rr;
        c_lds[it] = (rr * P + 4 * c4 + 2 * half) * 2;
        if (!EDGE) {
            const int colv = X0 - 16 * CM + 4 * c4 - p.vx0;      // the chunk's first column, in V
            const int vc = colv + 2 * half;
            c_ok[it] = colv >= 0 && colv + 4 <= p.vw;
            if (SHARP)
                c_off[it] = c_ok[it] ? (unsigned)((p.y0 + rr) * W + p.x0 + p.vx0 + vc) * 2u : 0u;
            else
                c_off[it] = c_ok[it] ? (unsigned)((rr - p.vy0) * p.vpitch + vc) * 4u : OOB;
        } else {
#pragma unroll
            for (int j = 0; j < (it < 2 ? 4 : 2); ++j) {
                const int e = 4 * it + j;
                // the element's column of the patch, reflected (stitcher.py:226: BORDER_REFLECT_101), in V
                const int vc = reflect_101(X0 - 16 * CM + 4 * c4 + 2 * half + j, p.w) - p.vx0;
                c_ok[e] = (unsigned)vc < (unsigned)p.vw;
                if (SHARP)
                    c_off[e] = c_ok[e] ? (unsigned)((p.y0 + rr) * W + p.x0 + p.vx0 + vc) * 2u : 0u;
                else
                    c_off[e] = c_ok[e] ? (unsigned)((rr - p.vy0) * p.vpitch + vc) * 4u : OOB;
            }
        }
    }
    // bands [row_lo, row_hi): inside the patch and inside V, no reflection
    const int row_lo = p.vy0 > 0 ? p.vy0 : 0;
    const int row_hi = p.vy0 + p.vh < p.h ? p.vy0 + p.vh : p.h;
    const unsigned long long src_base =
        (unsigned long long)(SHARP ? (const void *)owner_
                                   : (const void *)(p.planes + (size_t)ch * p.vh * p.vpitch));
    int4v rs;                                            // colour: the plane as a buffer descriptor
    rs[0] = __builtin_amdgcn_readfirstlane((int)(src_base & 0xffffffffull));
    rs[1] = __builtin_amdgcn_readfirstlane((int)((src_base >> 32) & 0xffffull));
    rs[2] = __builtin_amdgcn_readfirstlane(p.vh * p.vpitch * 4);
    rs[3] = 0x00020000;
    uint2v sbase2;                                       // mask: the owner map's address (scalar pair)
    sbase2[0] = (unsigned)__builtin_amdgcn_readfirstlane((int)(src_base & 0xffffffffull));
    sbase2[1] = (unsigned)__builtin_amdgcn_readfirstlane((int)(src_base >> 32));

    struct Band {                   // a band in flight: the loads' destination registers
        uint4v v0, v1;              // colour
        uint2v v2;
        unsigned e[10];             // mask: owner-map entries (4 + 4 + 2)
    };
    // The loads are inline assembly (the compiler's wait insertion must not see them: with the
    // previous tile's stores pending on the same counter it would wait for vmcnt(0)), issued and
    // consumed inside one step (see ml_body).
    auto issue = [&](Band &pf, const unsigned (&voff_)[NV]) {
#ifdef MS_ABL_NOLOAD                                      // timing experiment: no band traffic
        unsigned none[NV];
#pragma unroll
        for (int e = 0; e < NV; ++e) none[e] = SHARP ? 0u : OOB;
        const unsigned (&voff)[NV] = none;
        (void)voff_;
#else
        const unsigned (&voff)[NV] = voff_;
#endif
        // (s_nop 4: a scalar operand may have come back from a spill lane by v_readlane just
        // before, and a vector-memory instruction may read a scalar register a vector instruction
        // wrote only five wait states later - the compiler does not look into the statement.
        // ONE statement: no reload of the operand can come between the wait states and the loads.)
        if constexpr (EDGE && !SHARP) {
            asm volatile("s_nop 4\n\t"
                         "buffer_load_dword %0, %10, %20, 0 offen\n\t"
                         "buffer_load_dword %1, %11, %20, 0 offen\n\t"
                         "buffer_load_dword %2, %12, %20, 0 offen\n\t"
                         "buffer_load_dword %3, %13, %20, 0 offen\n\t"
                         "buffer_load_dword %4, %14, %20, 0 offen\n\t"
                         "buffer_load_dword %5, %15, %20, 0 offen\n\t"
                         "buffer_load_dword %6, %16, %20, 0 offen\n\t"
                         "buffer_load_dword %7, %17, %20, 0 offen\n\t"
                         "buffer_load_dword %8, %18, %20, 0 offen\n\t"
                         "buffer_load_dword %9, %19, %20, 0 offen"
                         : "=&v"(pf.e[0]), "=&v"(pf.e[1]), "=&v"(pf.e[2]), "=&v"(pf.e[3]), "=&v"(pf.e[4]),
                           "=&v"(pf.e[5]), "=&v"(pf.e[6]), "=&v"(pf.e[7]), "=&v"(pf.e[8]), "=&v"(pf.e[9])
                         : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "v"(voff[4]),
                           "v"(voff[5]), "v"(voff[6]), "v"(voff[7]), "v"(voff[8]), "v"(voff[9]), "s"(rs)
                         : "memory");
        } else if constexpr (EDGE && SHARP) {
            asm volatile("s_nop 4\n\t"
                         "global_load_sshort %0, %10, %20\n\t"
                         "global_load_sshort %1, %11, %20\n\t"
                         "global_load_sshort %2, %12, %20\n\t"
                         "global_load_sshort %3, %13, %20\n\t"
                         "global_load_sshort %4, %14, %20\n\t"
                         "global_load_sshort %5, %15, %20\n\t"
                         "global_load_sshort %6, %16, %20\n\t"
                         "global_load_sshort %7, %17, %20\n\t"
                         "global_load_sshort %8, %18, %20\n\t"
                         "global_load_sshort %9, %19, %20"
                         : "=&v"(pf.e[0]), "=&v"(pf.e[1]), "=&v"(pf.e[2]), "=&v"(pf.e[3]), "=&v"(pf.e[4]),
                           "=&v"(pf.e[5]), "=&v"(pf.e[6]), "=&v"(pf.e[7]), "=&v"(pf.e[8]), "=&v"(pf.e[9])
                         : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "v"(voff[4]),
                           "v"(voff[5]), "v"(voff[6]), "v"(voff[7]), "v"(voff[8]), "v"(voff[9]), "s"(sbase2)
                         : "memory");
        } else if constexpr (!SHARP) {
            asm volatile("s_nop 4\n\t"
                         "buffer_load_dwordx4 %0, %3, %6, 0 offen\n\t"
                         "buffer_load_dwordx4 %1, %4, %6, 0 offen\n\t"
                         "buffer_load_dwordx2 %2, %5, %6, 0 offen"
                         : "=&v"(pf.v0), "=&v"(pf.v1), "=&v"(pf.v2)
                         : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "s"(rs)
                         : "memory");
        } else {
            asm volatile("s_nop 4\n\t"
                         "global_load_sshort %0, %10, %13\n\t"
                         "global_load_sshort %1, %10, %13 offset:2\n\t"
                         "global_load_sshort %2, %10, %13 offset:4\n\t"
                         "global_load_sshort %3, %10, %13 offset:6\n\t"
                         "global_load_sshort %4, %11, %13\n\t"
                         "global_load_sshort %5, %11, %13 offset:2\n\t"
                         "global_load_sshort %6, %11, %13 offset:4\n\t"
                         "global_load_sshort %7, %11, %13 offset:6\n\t"
                         "global_load_sshort %8, %12, %13\n\t"
                         "global_load_sshort %9, %12, %13 offset:2"
                         : "=&v"(pf.e[0]), "=&v"(pf.e[1]), "=&v"(pf.e[2]), "=&v"(pf.e[3]), "=&v"(pf.e[4]),
                           "=&v"(pf.e[5]), "=&v"(pf.e[6]), "=&v"(pf.e[7]), "=&v"(pf.e[8]), "=&v"(pf.e[9])
                         : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "s"(sbase2)
                         : "memory");
        }
    };
    // registers -> LDS, in pieces (so that a piece fits the shadow of one matrix product):
    //   commit_wait<N>: waits for the band's loads, not for the N stores issued behind them;
    //   commit_piece<K>, K = 0 .. 2: slot K's values converted and written.
    auto commit_wait = [&](Band &pf, auto n_c) {
        constexpr int N = decltype(n_c)::value;
        if (SHARP || EDGE)
            asm volatile("s_waitcnt vmcnt(%10)"
                         : "+v"(pf.e[0]), "+v"(pf.e[1]), "+v"(pf.e[2]), "+v"(pf.e[3]), "+v"(pf.e[4]),
                           "+v"(pf.e[5]), "+v"(pf.e[6]), "+v"(pf.e[7]), "+v"(pf.e[8]), "+v"(pf.e[9])
                         : "n"(N) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(%3)" : "+v"(pf.v0), "+v"(pf.v1), "+v"(pf.v2) : "n"(N) : "memory");
    };
    const int lo_bytes_c = 32 * P * 2;
    auto commit_piece = [&](Band &pf, const int (&okv)[NV], const int buf, const int k) {
        unsigned char *const base = (unsigned char *)sh.hi + 2 * buf;
        if (!SHARP) {
            const float in_scale = __builtin_bit_cast(
                float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, MB_IN_SCALE)));
            if (k < 2) {
                uint4v v = k ? pf.v1 : pf.v0;
                if (EDGE) v = uint4v{pf.e[4 * k], pf.e[4 * k + 1], pf.e[4 * k + 2], pf.e[4 * k + 3]};
                unsigned h0, l0, h1, l1;
                ms_split4<false>(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]),
                                 __uint_as_float(v[3]), in_scale, h0, l0, h1, l1);
                const uint2v hi = {h0, h1}, lo = {l0, l1};
                *(uint2v *)(base + c_lds[k]) = hi;
                *(uint2v *)(base + c_lds[k] + lo_bytes_c) = lo;
            } else {
                unsigned h2, l2;
                ms_split2(__uint_as_float(EDGE ? pf.e[8] : pf.v2[0]),
                          __uint_as_float(EDGE ? pf.e[9] : pf.v2[1]), in_scale, h2, l2);
                *(unsigned *)(base + c_lds[2]) = h2;
                *(unsigned *)(base + c_lds[2] + lo_bytes_c) = l2;
            }
        } else {                                         // stitcher.py:207-208: the 0 / 1 mask, no low part
            const _Float16 one = (_Float16)MB_IN_SCALE, zero = (_Float16)0.0f;
            if (k < 2) {
                half4 hi;
#pragma unroll
                for (int j = 0; j < 4; ++j) hi[j] = (int)pf.e[4 * k + j] == okv[entry_of(k, j)] ? one : zero;
                *(half4 *)(base + c_lds[k]) = hi;
            } else {
                half2v h2;
                h2[0] = (int)pf.e[8] == okv[entry_of(2, 0)] ? one : zero;
                h2[1] = (int)pf.e[9] == okv[entry_of(2, 1)] ? one : zero;
                *(half2v *)(base + c_lds[2]) = h2;
            }
        }
    };
    auto word_at = [&](const int i) -> unsigned {
        return (unsigned)__builtin_amdgcn_readfirstlane((int)list[i]);
    };
    auto band_of = [](const unsigned w) { return (int)(short)(w & 0xffffu); };
    const unsigned keep = live ? 0x1fu : 0u;
    auto bits_of = [&](const unsigned w) { return (w >> (16 + 8 * tile)) & keep; };
    if (nlist <= 0) return;                               // uniform

    // store geometry: per lane the column part of a tile's offset (beyond the plane outside A)
    const int ax = px0 + n - p.ax0;
    const unsigned s_lane = (unsigned)ax < (unsigned)p.aw ? (unsigned)(4 * h * p.apitch + ax) * 4u : OOB;
    const int rowstep = p.apitch * 4;
    // operand addresses that do not change: this lane's band row / Toeplitz columns
    const int a_lane = (n * P + 16 * (CM - C) + 32 * tile + 8 * h) * 2;        // bytes from sh.hi
    const int lo_bytes = 32 * P * 2;

    int i = 0;
    unsigned word = word_at(0);
    int prev_o = 0, prev_u = 0;
    bool prev_store = false;
    auto store_prev_generic = [&]() {                    // the run's last tile (outside the step)
        switch (prev_u) {
#define MS_STORE_CASE(UU)                                                                      \
    case UU:                                                                                   \
        if constexpr (UU < NB) {                                                               \
            ml_store(acc[(UU + DMAX + 1) % NB], prev_o, lane, p, dst, px0, prev_store);        \
            if constexpr (CB != 0)                                                             \
                ml_store(acc_b[(UU + DMAX + 1) % NB], prev_o, lane, p, dst_b, px0, prev_store); \
        }                                                                                      \
        break;
            MS_STORE_CASE(0) MS_STORE_CASE(1) MS_STORE_CASE(2) MS_STORE_CASE(3) MS_STORE_CASE(4)
#undef MS_STORE_CASE
        }
        prev_store = false;
    };
    // The offsets of band tb's chunks (and, for the mask, the owner value that counts as "ours").
    // A band whose 32 rows lie inside the patch and inside V: one add per slot; any other band
    // (rows that reflect, rows beyond V, no band at all): the long way, under a branch.
    auto band_inside = [&](const int tb, const bool exists) {
        return exists && 32 * tb >= row_lo && 32 * tb + 32 <= row_hi;                  // uniform
    };
    auto offsets_fast = [&](const int tb, unsigned (&voff)[NV], int (&okv)[NV]) {
        const unsigned s_off = (unsigned)(32 * tb) * (unsigned)(SHARP ? W * 2 : p.vpitch * 4);
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            voff[e] = c_off[e] + s_off;                  // (beyond the plane stays beyond it)
            okv[e] = c_ok[e] ? p.index : -2;
        }
    };
    auto offsets_general = [&](const int tb, const bool exists, unsigned (&voff)[NV], int (&okv)[NV]) {
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            const int rr = c_rr[slot_of(e)];
            // (EDGE takes the low patches too, whose band rows may reflect more than once)
            const int prow = 32 * tb + rr;
            const int ry = EDGE ? reflect_101(prow, p.h)
                                : (prow < 0 ? -prow : (prow >= p.h ? 2 * p.h - 2 - prow : prow));
            const int vr = ry - p.vy0;
            const bool ok = exists && c_ok[e] && (unsigned)vr < (unsigned)p.vh;
            if (SHARP)
                voff[e] = ok ? c_off[e] + (unsigned)(ry - rr) * (unsigned)(W * 2) : 0u;
            else
                voff[e] = ok ? c_off[e] + (unsigned)(ry - rr) * (unsigned)(p.vpitch * 4) : OOB;
            okv[e] = ok ? p.index : -2;
        }
    };
#ifdef MB_STAMP
    // phase timers (timing experiments only): waves 0, 2, 4, 6 of a sample of colour workgroups
#ifndef MB_STAMP_CH
#define MB_STAMP_CH 0
#endif
    const bool stamped = (wv & 1) == 0 && ch == MB_STAMP_CH && (blockIdx.x >> 2) % 7 == 3;
    unsigned long long tlast = __builtin_readcyclecounter();
#define MS_STAMP(k)                                                                      \
    do {                                                                                 \
        if (stamped) {                                                                   \
            const unsigned long long now_ = __builtin_readcyclecounter();               \
            if (lane == 0) atomicAdd(&g_mb_stamps[wv >> 1][k], now_ - tlast);            \
            tlast = __builtin_readcyclecounter();                                        \
        }                                                                                \
    } while (0)
#else
#define MS_STAMP(k) do { } while (0)
#endif
    int off_cur = __builtin_amdgcn_readfirstlane(0), off_nxt = bstride;
    // What a step finds prepared (by the step before it, in the shadow of its products; by the
    // lines below for the first step): the offsets of the chunks it will fetch - band i + 1's -
    // assuming that band lies inside (fetch_general: it does not, the step's head computes them
    // the long way), and the finished tile's store base (store_edge: the tile is cut by A's
    // first or last row, the head stores it under masks).
    unsigned voff[NV];
    int okv[NV];
    unsigned next = nlist > 1 ? word_at(1) : 0u;        // band i + 1's word
    unsigned s_at = OOB;
    bool store_edge = false, fetch_general;
    {
        Band pf;
        if (band_inside(band_of(word), true))
            offsets_fast(band_of(word), voff, okv);
        else
            offsets_general(band_of(word), true, voff, okv);
        issue(pf, voff);
        commit_wait(pf, std::integral_constant<int, 0>{});      // (no stores behind these loads)
#pragma unroll
        for (int k = 0; k < 3; ++k) commit_piece(pf, okv, off_cur, k);
        offsets_fast(band_of(next), voff, okv);
        fetch_general = !band_inside(band_of(next), nlist > 1);
    }
    auto step = [&](auto u_c) -> bool {
        constexpr int U = decltype(u_c)::value;
        constexpr int KP = (U + DMAX) % NB;              // the tile last step's band completed
        const int t = band_of(word);
        const unsigned inf = bits_of(word);
        const bool more = i + 1 < nlist;
        off_cur = __builtin_amdgcn_readfirstlane(off_cur);
        off_nxt = __builtin_amdgcn_readfirstlane(off_nxt);
        asm volatile("" : "+s"(off_cur), "+s"(off_nxt));
        MS_STAMP(0);                // the step's scalar head
#ifdef MS_ABL_HALF_BARRIERS         // timing experiment (results wrong): a barrier every second step
        if (!(i & 1))
#endif
        lds_barrier();              // band i is whole; nobody reads the other buffer any more
        MS_STAMP(1);                // waiting at the barrier
        // ---- the rare cases the previous step left to this one
        Band pf;
        unsigned nn = 0;                                 // the list word after next (prepare_next)
        if (fetch_general) offsets_general(band_of(next), more, voff, okv);
        if (store_edge) {                                // cut by A's first or last row
            ml_store(acc[KP], prev_o, lane, p, dst, px0, true);
            if constexpr (CB != 0) ml_store(acc_b[CB ? KP : 0], prev_o, lane, p, dst_b, px0, true);
        }
        // (its sixteen stores stand in front of the block's, whose base is beyond the plane then)
        const bool work = t >= my_lo && t <= my_hi && inf != 0;                        // uniform
        // ---- the block.  Its order is written out by hand and pinned: the scheduler may not
        // move anything across a sched_barrier(0), so every product is followed by the few
        // vector / memory instructions that are to issue in its shadow (a 32x32x16 product
        // occupies the matrix pipe for 32 cycles and the SIMD's vector issue for 8 of them).
#define MS_PIN() __builtin_amdgcn_sched_barrier(0)
// cache policy of the tile stores: 2 = nt (the blurred copies are written once and read by the
// collapse a kernel later, long after they have left the L2: blur 0.697 -> 0.680 ms)
#ifndef MS_STORE_AUX
#define MS_STORE_AUX 2
#endif
#ifdef MS_ABL_NOSTORE                                    // timing experiment: every store dropped
#define MS_STORE_AT OOB
#else
#define MS_STORE_AT s_at
#endif
        auto block = [&](auto work_c) {
            constexpr bool WORK = decltype(work_c)::value;
            constexpr int PR = SHARP ? 2 : 3;            // products per k-step of the row pass
            // the block's gaps (one behind every product), numbered through the passes of the
            // wave's level - or of its two levels, one after the other (CB)
            constexpr int BLK_A = NB - Z, GR_A = KS * PR, GC_A = 2 * BLK_A * 3;
            constexpr int BLK_B = CB ? NB - Z_B : 0, GR_B = CB ? KS_B * PR : 0, GC_B = 2 * BLK_B * 3;
            constexpr int G = GR_A + GC_A + GR_B + GC_B;
            constexpr int NSTORE = CB ? 32 : 16;         // values of the finished tile(s) per lane
            constexpr int PER_R = (NSTORE + GR_A - 1) / GR_A;   // scaled values per gap of the first row pass
            constexpr int G_COMMIT = GR_A + BLK_A * 3;   // first gap of the first column pass's second half
            MS_STAMP(2);            // prologue
            issue(pf, voff);
            // the list word after next (wanted at the block's end: its latency is covered)
            const bool more2 = i + 2 < nlist;
            const unsigned nn_raw = list[more2 ? i + 2 : 0];
            MS_PIN();
            // ---- what the NEXT step finds prepared (see above), computed in a product's shadow
            auto prepare_next = [&]() {
                const unsigned got = (unsigned)__builtin_amdgcn_readfirstlane((int)nn_raw);
                nn = more2 ? got : 0u;                   // (a select, not a branch)
                offsets_fast(band_of(nn), voff, okv);
                fetch_general = !band_inside(band_of(nn), more2);
                const bool wanted = (inf >> (DMAX + 2)) & 1u;        // tile t - DMAX: complete now
                const int o = t - DMAX;
                const bool rows_in = 32 * o >= p.ay0 && 32 * o + 32 <= p.ay0 + p.ah;   // uniform
                store_edge = wanted && !rows_in;
                // (a lane outside A's columns: beyond the plane plus less than a plane stays beyond it)
                s_at = wanted && rows_in ? s_lane + (unsigned)((32 * o - p.ay0) * p.apitch) * 4u : OOB;
            };
            // (opaque: the same product with the literal stands in ml_store, and the compiler
            // would compute the sixteen of them once, in front of the branch that leads here)
            float out_scale = MB_OUT_SCALE;
            asm("" : "+s"(out_scale));
            // The finished tile(s): scaled during the first gaps, stored at an even pace over ALL
            // the block's gaps.  (A CU passes one dword store instruction per ~11 cycles, 128 of
            // them per step for its eight waves: issued back to back behind the barrier they made
            // the row pass 2.5 - 3.3 k cycles long for 0.8 k cycles of products - phase timers,
            // profiles/r04/notes.md.)
            float sc[NSTORE];
            auto store_gap = [](const int q) { return ((2 * q + 1) * G) / (2 * NSTORE); };
            auto fill = [&](const int g) {               // what issues in gap g's shadow
                if (g < GR_A) {
#pragma unroll
                    for (int c = 0; c < PER_R; ++c) {
                        const int q = g * PER_R + c;
                        if (q < 16) sc[q] = acc[KP][q] * out_scale;
                        if (CB && q >= 16 && q < NSTORE) sc[q] = acc_b[CB ? KP : 0][q & 15] * out_scale;
                    }
                }
#pragma unroll
                for (int q = 0; q < NSTORE; ++q)
                    if (store_gap(q) == g)
                        __builtin_amdgcn_raw_buffer_store_b32(
                            __float_as_uint(sc[q]), q < 16 ? dst : dst_b, MS_STORE_AT,
                            (((q & 15) & 3) + 8 * ((q & 15) >> 2)) * rowstep, MS_STORE_AUX);
            };
            // how many of the tile's stores fill() has issued in the gaps in front of the commit:
            // commit_wait's vmcnt leaves exactly those in flight behind the band's loads, so the
            // count must follow store_gap (the same expression) and the stores must be the
            // raw_buffer_store_b32 of fill(), one VMEM instruction each
            constexpr int stores_before_commit = [] {
                int n = 0;
                for (int q = 0; q < NSTORE; ++q) n += ((2 * q + 1) * G) / (2 * NSTORE) < G_COMMIT ? 1 : 0;
                return n;
            }();
            static_assert(stores_before_commit >= 0 && stores_before_commit <= NSTORE,
                          "the commit's vmcnt counts the stores fill() issued before it");
            if constexpr (!WORK) {
#pragma unroll
                for (int g = 0; g < G; ++g) fill(g);
                MS_PIN();
                commit_wait(pf, std::integral_constant<int, NSTORE>{});
#pragma unroll
                for (int k = 0; k < 3; ++k) commit_piece(pf, okv, off_nxt, k);
                prepare_next();
            } else {
                // One level's two passes.  G0: the level's first gap; COMMIT: the next band is
                // converted and written to LDS under the second half of this level's column pass;
                // LAST: the next step's scalars are prepared in this level's last gap.
                auto level = [&](auto cc_c, auto g0_c, auto commit_c, auto last_c, f32x16 (&A)[NB],
                                 const half8 *tx, const half8 *ty) {
                    constexpr int CC = decltype(cc_c)::value, G0 = decltype(g0_c)::value;
                    constexpr bool COMMIT = decltype(commit_c)::value, LAST = decltype(last_c)::value;
                    constexpr int KSL = 2 + 2 * CC, DMAXL = (CC + 1) / 2, ZL = CC & 1, BLK = NB - ZL;
                    static_assert(2 * DMAXL + 1 == NB, "a wave's levels share the accumulator rotation");
                    // row pass: the operands of k-step s + 1 are read while k-step s multiplies
                    const unsigned char *const arow =
                        (const unsigned char *)sh.hi + (a_lane + 32 * (C - CC)) + 2 * off_cur;
                    f32x16 mid;
#pragma unroll
                    for (int q = 0; q < 16; ++q) mid[q] = 0.0f;
                    // (MS_AHEAD k-steps of operands in flight: 1 = the next k-step's reads in
                    // front of this one's products)
#ifndef MS_AHEAD
#define MS_AHEAD 1
#endif
                    constexpr int NBUF = MS_AHEAD + 1;
                    half8 a_hi[NBUF], a_lo[NBUF], b_hi[NBUF], b_lo[NBUF];
                    auto operands = [&](const int s, const int b) {
                        a_hi[b] = *(const half8 *)(arow + 32 * s);
                        b_hi[b] = tx[mb_tx_index(KSL, lane, s, 0)];
                        b_lo[b] = tx[mb_tx_index(KSL, lane, s, 1)];
                        if (!SHARP) a_lo[b] = *(const half8 *)(arow + lo_bytes + 32 * s);
                    };
#pragma unroll
                    for (int s = 0; s < MS_AHEAD && s < KSL; ++s) operands(s, s % NBUF);
#pragma unroll
                    for (int s = 0; s < KSL; ++s) {
                        const int cur = s % NBUF;
                        if (s + MS_AHEAD < KSL) operands(s + MS_AHEAD, (s + MS_AHEAD) % NBUF);
                        MS_PIN();
#pragma unroll
                        for (int m = 0; m < PR; ++m) {
                            mid = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                m == 2 ? a_lo[cur] : a_hi[cur], m == 1 ? b_lo[cur] : b_hi[cur], mid, 0, 0, 0);
                            MS_PIN();
                            fill(G0 + s * PR + m);
                            MS_PIN();
                        }
                    }
                    if (G0 == 0) MS_STAMP(3);            // (first) row pass
                    // column pass, k-half-major.  Block b = (half, d): two operand reads and three
                    // products; the operands of block b + 1 are read in front of block b's
                    // products; the second half of Mid is split under the first half's products.
                    unsigned m_hi[2][4], m_lo[2][4];      // eight float16 each: the B operands
                    auto operand_b = [](const unsigned (&w)[4]) {
                        const uint4v v = {w[0], w[1], w[2], w[3]};
                        return __builtin_bit_cast(half8, v);
                    };
                    const float mid_scale = __builtin_bit_cast(
                        float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, MB_MID_SCALE)));
                    half8 t_hi[2], t_lo[2];
                    auto ty_read = [&](const int bb, const int buf) {       // bb = s2 * BLK + b
                        const int s2 = bb / BLK, b = bb - s2 * BLK;
                        const int d = -DMAXL + b + (ZL && s2 == 0 ? 1 : 0);
                        const half8 *at = ty + (((d + DMAXL) * 2 + s2 - ZL) * 2) * 64 + lane;
                        t_hi[buf] = at[0];
                        t_lo[buf] = at[64];
                    };
                    ty_read(0, 0);
                    MS_PIN();
                    ms_split4<true>(mid[0], mid[1], mid[2], mid[3], mid_scale, m_hi[0][0], m_lo[0][0],
                                    m_hi[0][1], m_lo[0][1]);
                    ms_split4<false>(mid[4], mid[5], mid[6], mid[7], mid_scale, m_hi[0][2], m_lo[0][2],
                                     m_hi[0][3], m_lo[0][3]);
                    MS_PIN();
                    constexpr int GAPS_H = BLK * 3, PAIRS_PER = (4 + GAPS_H - 1) / GAPS_H;
#pragma unroll
                    for (int bb = 0; bb < 2 * BLK; ++bb) {
                        const int s2 = bb / BLK, b = bb - s2 * BLK, buf = bb & 1;
                        const int d = -DMAXL + b + (ZL && s2 == 0 ? 1 : 0);
                        const int k = ((U - d) % NB + NB) % NB;      // tile t - d lives in accumulator k
                        const bool first = d == -DMAXL && s2 == ZL;  // starts tile t + DMAX's sum
                        if (bb + 1 < 2 * BLK) ty_read(bb + 1, buf ^ 1);
                        MS_PIN();
#pragma unroll
                        for (int m = 0; m < 3; ++m) {
                            if (m == 0 && first) {
                                f32x16 zero;
#pragma unroll
                                for (int q = 0; q < 16; ++q) zero[q] = 0.0f;
                                A[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                    t_hi[buf], operand_b(m_hi[s2]), zero, 0, 0, 0);
                            } else {
                                A[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                    m == 1 ? t_lo[buf] : t_hi[buf],
                                    operand_b(m == 2 ? m_lo[s2] : m_hi[s2]), A[k], 0, 0, 0);
                            }
                            MS_PIN();
                            const int hg = b * 3 + m;                // gap within this half
                            if (s2 == 0) {
#pragma unroll
                                for (int c = 0; c < PAIRS_PER; ++c) {
                                    const int pr = hg * PAIRS_PER + c;
                                    if (pr < 4)
                                        ms_split2(mid[8 + 2 * pr], mid[9 + 2 * pr], mid_scale,
                                                  m_hi[1][pr], m_lo[1][pr]);
                                }
                            } else if (COMMIT) {
                                // (the loads were issued some 3 k cycles ago; the stores issued
                                // since stay in flight)
                                if (hg == 0)
                                    commit_wait(pf, std::integral_constant<int, stores_before_commit>{});
                                if (hg < 3) commit_piece(pf, okv, off_nxt, hg);
                            }
                            fill(G0 + KSL * PR + s2 * GAPS_H + hg);
                            // (after this gap's store: the store's base is the one prepared for THIS step)
                            if (LAST && s2 == 1 && hg == GAPS_H - 1) prepare_next();
                            MS_PIN();
                        }
                    }
                };
                level(std::integral_constant<int, C>{}, std::integral_constant<int, 0>{}, std::true_type{},
                      std::integral_constant<bool, CB == 0>{}, acc, s_tx, s_ty);
                if constexpr (CB != 0)
                    level(std::integral_constant<int, CB>{}, std::integral_constant<int, GR_A + GC_A>{},
                          std::false_type{}, std::true_type{}, acc_b, s_tx_b, s_ty_b);
                MS_STAMP(4);        // split + column pass (+ the next band's conversion)
            }
            MS_PIN();
            MS_STAMP(5);
#ifdef MB_STAMP
            if (stamped && lane == 0) atomicAdd(&g_mb_stamps[wv >> 1][11], 1ull);
#endif
        };
#ifdef MS_SKIP_IDLE
        if (work)
            block(std::true_type{});
        else
            block(std::false_type{});
#else
        (void)work;
        block(std::true_type{});
#endif
        prev_store = (inf >> (DMAX + 2)) & 1u;           // tile t - DMAX is wanted: complete now
        prev_o = t - DMAX;
        prev_u = U;
        const bool run_on = more && band_of(next) == t + 1;
        word = next;
        next = nn;
        {
            const int tmp = off_cur;
            off_cur = off_nxt;
            off_nxt = tmp;
        }
        ++i;
        return run_on;
    };
#ifdef MS_PRIO
    // (experiment: the second-dispatched half of the workgroup loses every arbitration against
    // its SIMD partners, MI355X_MICROARCH.md "Two waves per SIMD" item 4)
    if (wv >= 4) __builtin_amdgcn_s_setprio(MS_PRIO);
#endif
    while (i < nlist) {                                  // one run per trip
#pragma unroll
        for (int k = 0; k < NB; ++k)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                acc[k][q] = 0.0f;
                if (CB) acc_b[CB ? k : 0][q] = 0.0f;
            }
        for (;;) {
            if (!step(std::integral_constant<int, 0>{})) break;
            if (!step(std::integral_constant<int, 1>{})) break;
            if (!step(std::integral_constant<int, 2>{})) break;
            if constexpr (NB > 3) {
                if (!step(std::integral_constant<int, 3>{})) break;
                if (!step(std::integral_constant<int, 4>{})) break;
            }
        }
        if (prev_store) store_prev_generic();            // the run's last tile, before the reset
        s_at = OOB;                                      // (stored: nothing is left for the next run's first step)
        store_edge = false;
    }
}

// FIVE: the launch of five levels on four wave pairs.  It is a kernel of its own
// (blur_lean5_kernel) because its two-level wave pair needs all 256 vector registers a wave can
// have at two waves per SIMD, and a kernel's allocation is its largest path's: the four-level
// kernel stays at 217, which leaves a SIMD's register file room for a third, small wave of
// another kernel (the collapse of the other stitch in flight: 54 registers).
template <bool FIVE>
__device__ __forceinline__ void lean_kernel_body(
    const pano_patch *__restrict__ table, const MbLevels &L, const unsigned char *__restrict__ tables,
    const int16_t *__restrict__ owner, int W, const uint8_t *__restrict__ flags,
    const int2 *__restrict__ items, unsigned char *smem) {
    constexpr int GROUP = 4;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
#ifdef MB_STAMP
    unsigned long long ts_last = __builtin_readcyclecounter();
#define MS_SETUP_STAMP(k)                                                               \
    do {                                                                                \
        const unsigned long long now_ = __builtin_readcyclecounter();                   \
        if (tid == 0) atomicAdd(&g_mb_setup[k], now_ - ts_last);                        \
        ts_last = now_;                                                                 \
    } while (0)
#else
#define MS_SETUP_STAMP(k) do { } while (0)
#endif
    // (five levels: ONE group of four wave pairs, the two lightest levels on one pair)
    constexpr bool five = FIVE;
    const int ngroups = five ? 1 : (L.n + GROUP - 1) / GROUP;
    const int per = 8 * ngroups, blk = blockIdx.x / per, within = blockIdx.x - blk * per;
    const int grp = within >> 3, pair = blk * 8 + (within & 7);
    const int ch = pair & 3, slot = pair >> 2;
    const int2 item = items[slot];
    if (item.x < 0) return;                                                     // uniform
    const int pid = item.x & 0xffff, tx0 = item.x >> 16;
    const pano_patch p = table[pid];
    const MbGeom g = mb_geom(p);
    const int l0 = GROUP * grp, nl = five ? 5 : (L.n - l0 < GROUP ? L.n - l0 : GROUP);
    // the group's reach, and whether this item is one of ours
    MbShared sh;
    sh.CM = 1;
    for (int k = 0; k < nl; ++k) {
        const int ck = mb_c_of(L.ntaps[l0 + k]);
        sh.CM = ck > sh.CM ? ck : sh.CM;
    }
    const int n_seg = item.y >> 16, seg = (item.y >> 12) & 15, nty_all = g.O1 - g.O0 + 1;
    const int o_begin = n_seg > 1 ? nty_all * seg / n_seg : 0;
    const int o_end = n_seg > 1 ? nty_all * (seg + 1) / n_seg : nty_all;
    const int q = __builtin_amdgcn_readfirstlane(wv >> 1);
    const int lv = mb_level_of_pair(nl, q);
    const bool live = lv >= 0;
    const int level = l0 + (live ? lv : 0);
    const int ntaps = L.ntaps[level], c = mb_c_of(ntaps);
    const int lv_b = mb_second_level_of_pair(nl, q);    // the pair's second level, or -1
    const int level_b = l0 + (lv_b >= 0 ? lv_b : 0);
    int dmax_of[GROUP], rel = 0, my_tx = 0, my_ty = 0, my_tx_b = 0, my_ty_b = 0;
    for (int k = 0; k < GROUP; ++k) {
        const int lk = mb_level_of_pair(nl, k);
        dmax_of[k] = -1;
        if (lk < 0) continue;
        const int ck = mb_c_of(L.ntaps[l0 + lk]);
        dmax_of[k] = (ck + 1) / 2;
        if (k == q) {
            my_tx = rel;
            my_ty = rel + mb_tx_bytes(2 + 2 * ck);
        }
        rel += mb_table_bytes(L.ntaps[l0 + lk]);
        const int lk2 = mb_second_level_of_pair(nl, k);
        if (lk2 >= 0) {                                  // (equal DMAX: the host checked)
            const int ck2 = mb_c_of(L.ntaps[l0 + lk2]);
            if (k == q) {
                my_tx_b = rel;
                my_ty_b = rel + mb_tx_bytes(2 + 2 * ck2);
            }
            rel += mb_table_bytes(L.ntaps[l0 + lk2]);
        }
    }
    sh.P = mb_pitch_of(sh.CM);
    sh.hi = (_Float16 *)smem;
    sh.lo = sh.hi + 32 * sh.P;
    sh.need = (uint8_t *)(sh.lo + 32 * sh.P);
    sh.any = sh.need + 2 * MB_NEED_LEN;
    sh.info = (uint16_t *)(sh.any + MB_NEED_LEN);
    sh.col = (short *)(sh.info + 2 * MB_NEED_LEN);
    uint32_t *list = (uint32_t *)sh.info;                // [MB_NEED_LEN] words: same bytes
    my_tx += mb_fixed_bytes(sh.CM);
    my_ty += mb_fixed_bytes(sh.CM);
    my_tx_b += mb_fixed_bytes(sh.CM);
    my_ty_b += mb_fixed_bytes(sh.CM);
    const int dmaxm = (sh.CM + 1) / 2;
    sh.t_lo = g.O0 - dmaxm;
    sh.t_hi = g.O1 + dmaxm;
    MS_SETUP_STAMP(0);                                   // item, record, geometry
    for (int which = 0; which < (lv_b >= 0 ? 2 : 1); ++which)
    if (live) {
        const uint4 *from = (const uint4 *)(tables + L.tab_off[which ? level_b : level]);
        uint4 *to = (uint4 *)(smem + (which ? my_tx_b : my_tx));
        const int n16 = mb_table_bytes(L.ntaps[which ? level_b : level]) >> 4;
        // Every load of a round issued before its first LDS store, and every one of them
        // UNCONDITIONAL (an index past the table is clamped to its last entry, which is then
        // written twice with the same bytes): a load under a branch is waited for on the spot,
        // and thirteen such round trips were 21 k cycles of a workgroup's 35 k of set-up
        // (phase timers, profiles/r04/notes.md).
        constexpr int DEPTH = 8;
        for (int i0 = tid & 127; i0 < n16; i0 += 128 * DEPTH) {
            uint4 v[DEPTH];
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) v[j] = from[min(i0 + 128 * j, n16 - 1)];
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) to[min(i0 + 128 * j, n16 - 1)] = v[j];
        }
    }
    MS_SETUP_STAMP(1);                                   // table copy
    const int X0 = g.gx0 + 32 * tx0;
    const int nty = g.O1 - g.O0 + 1;
    if (wv < 2) {
        const int txg = ((X0 - g.gx0) >> 5) + wv;
        for (int i = lane; i < nty + 2 * MB_NEED_PAD; i += 64) {
            const int o = i - MB_NEED_PAD;
            bool v = txg < g.ntx && o >= o_begin && o < o_end;
            if (v && flags) v = flags[p.tiles_off + o * g.ntx + txg] != 0;
            sh.need[wv * MB_NEED_LEN + i] = v ? 1 : 0;
        }
    }
    int *const s_count = (int *)sh.any;                  // (the `any` flags are not used here)
    __syncthreads();
    MS_SETUP_STAMP(2);                                   // need flags + barrier (the table copy's LDS writes land)
    // the bands some wave wants, in order, compacted: one wave, 64 bands per round
    if (wv == 0) {
        int count = 0;
        for (int base = 0; base <= sh.t_hi - sh.t_lo; base += 64) {
            const int i = base + lane, t = sh.t_lo + i;
            const int o = t - g.O0 + MB_NEED_PAD;                    // index of tile t into need[]
            bool any = false;
            unsigned w = 0;
            if (i <= sh.t_hi - sh.t_lo) {
                // (ten byte reads: the wanted bits of the two columns; a band is listed when a
                // tile within the group's largest reach of it is wanted - the union of the pairs'
                // windows, which the loop over the pairs used to read one by one)
#pragma unroll
                for (int col = 0; col < 2; ++col)
#pragma unroll
                    for (int d = -2; d <= 2; ++d)
                        w |= (sh.need[col * MB_NEED_LEN + o - d] ? 1u : 0u) << (16 + 8 * col + d + 2);
                const unsigned reach_bits = dmaxm >= 2 ? 0x1fu : 0x0eu;      // d = -dmaxm .. dmaxm
                any = (((w >> 16) | (w >> 24)) & reach_bits) != 0;
                w |= (unsigned)(unsigned short)(short)t;
            }
            const unsigned long long bal = __ballot(any);
            if (any) list[count + __popcll(bal & ((1ull << lane) - 1ull))] = w;
            count += __popcll(bal);
        }
        if (lane == 0) *s_count = count;
    }
    __syncthreads();
    MS_SETUP_STAMP(3);                                   // list
#ifdef MB_STAMP
    if (tid == 0) atomicAdd(&g_mb_setup[7], 1ull);
#endif
    const int nlist = *s_count;
    const half8 *s_tx = (const half8 *)(smem + my_tx), *s_ty = (const half8 *)(smem + my_ty);
    const int out_level = L.out[level];
    // the second band buffer lies behind the group's tables (the host sized the LDS for it)
    const int second = (mb_fixed_bytes(sh.CM) + rel + 15) / 16 * 8;     // halfs from sh.hi
#define ML_BODY_FN ms_body
    const bool edge = mb_item_edge(p, g.gx0, tx0, sh.CM);                               // uniform
    if constexpr (FIVE)
    if (lv_b >= 0) {                                     // wave-uniform: levels of 2 and 1 K-steps' reach
        const half8 *s_tx_b = (const half8 *)(smem + my_tx_b), *s_ty_b = (const half8 *)(smem + my_ty_b);
#define MS_DUAL(SH, ED)                                                                        \
    ms_body<2, SH, 1, ED>(p, ch, out_level, live, s_tx, s_ty, sh, list, nlist, owner, W, tx0,  \
                          second, L.out[level_b], s_tx_b, s_ty_b)
        if (edge) {
            if (ch == 3) MS_DUAL(true, true); else MS_DUAL(false, true);
        } else {
            if (ch == 3) MS_DUAL(true, false); else MS_DUAL(false, false);
        }
#undef MS_DUAL
        return;
    }
    if (edge) {
        switch (c) {                                     // wave-uniform
#define MS_EDGE(CC)                                                                            \
    if (ch == 3)                                                                               \
        ms_body<CC, true, 0, true>(p, ch, out_level, live, s_tx, s_ty, sh, list, nlist, owner, W, tx0, second); \
    else                                                                                       \
        ms_body<CC, false, 0, true>(p, ch, out_level, live, s_tx, s_ty, sh, list, nlist, owner, W, tx0, second); \
    break;
            case 1: MS_EDGE(1)
            case 2: MS_EDGE(2)
            default: MS_EDGE(3)
#undef MS_EDGE
        }
        return;
    }
    switch (c) {                                         // wave-uniform
#define ML_BODY(CC)                                                                            \
    if (ch == 3)                                                                               \
        ML_BODY_FN<CC, true>(p, ch, out_level, live, s_tx, s_ty, sh, list, nlist, owner, W, tx0, second); \
    else                                                                                       \
        ML_BODY_FN<CC, false>(p, ch, out_level, live, s_tx, s_ty, sh, list, nlist, owner, W, tx0, second); \
    break;
        case 1: ML_BODY(1)
        case 2: ML_BODY(2)
        case 3: ML_BODY(3)
        default: ML_BODY(4)
#undef ML_BODY
    }
}

__global__ __launch_bounds__(MB_THREADS_OF(4), 1) void blur_lean_kernel(
    const pano_patch *__restrict__ table, MbLevels L, const unsigned char *__restrict__ tables,
    const int16_t *__restrict__ owner, int W, const uint8_t *__restrict__ flags,
    const int2 *__restrict__ items) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lean_kernel_body<false>(table, L, tables, owner, W, flags, items, smem);
}

__global__ __launch_bounds__(MB_THREADS_OF(4), 1) void blur_lean5_kernel(
    const pano_patch *__restrict__ table, MbLevels L, const unsigned char *__restrict__ tables,
    const int16_t *__restrict__ owner, int W, const uint8_t *__restrict__ flags,
    const int2 *__restrict__ items) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lean_kernel_body<true>(table, L, tables, owner, W, flags, items, smem);
}

// One thread per 32 x 32 tile of every record: active = some interior-map block under the
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
    const int bx0 = (p.x0 + x0) / PANO_INTERIOR_BLOCK, bx1 = (p.x0 + x1 - 1) / PANO_INTERIOR_BLOCK;
    const int by0 = (p.y0 + y0) / PANO_INTERIOR_BLOCK, by1 = (p.y0 + y1 - 1) / PANO_INTERIOR_BLOCK;
    bool active = false;
    static_assert(PANO_INTERIOR_BLOCK == 4, "a 32-pixel tile spans at most nine interior blocks");
    if (bx0 + 8 <= W8) {
        // a tile row's (at most nine) map bytes as one 8-byte load at any alignment, plus the
        // ninth: 81 byte loads per tile were the kernel (its bytes are 0 or 1)
        typedef uint64_t u64_any __attribute__((aligned(1)));
        const int nb = bx1 - bx0 + 1;                                    // 1 .. 9
        const uint64_t ones = nb >= 8 ? 0x0101010101010101ull : (0x0101010101010101ull >> (8 * (8 - nb)));
        for (int by = by0; by <= by1; ++by) {
            const uint8_t *row = interior + (size_t)by * W8 + bx0;
            active |= (*(const u64_any *)row & ones) != ones;
            if (nb == 9) active |= row[8] == 0;
        }
    } else {
        for (int by = by0; by <= by1; ++by)
            for (int bx = bx0; bx <= bx1; ++bx) active |= interior[(size_t)by * W8 + bx] == 0;
    }
    flags[p.tiles_off + ty * g.ntx + tx] = active ? 1 : 0;
}

// Which tiles' pixels of the warped planes anything will read: the blur fetches, for a pair
// of tile columns opened at an active column, the band columns [32 c - 16 CM, 32 c + 64 + 16 CM)
// of every band within DMAX tiles of a wanted tile, and the collapse reads the non-interior
// pixels (inside active tiles).  need = the active flags dilated by hx tile columns and vy
// tile rows (a superset); the warp skips the rest of window V (about half of it on cfg3).
__global__ __launch_bounds__(256) void warp_need_kernel(const pano_patch *__restrict__ table,
                                                        const uint8_t *__restrict__ flags, int hx,
                                                        int vy, uint8_t *__restrict__ need) {
    const pano_patch p = table[blockIdx.z];
    if (p.aw <= 0 || p.ah <= 0) return;
    const MbGeom g = mb_geom(p);
    const int nty = g.O1 - g.O0 + 1;
    const int tx = blockIdx.x * 32 + (threadIdx.x & 31), ty = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (tx >= g.ntx || ty >= nty) return;
    bool any = false;
    for (int y = max(ty - vy, 0); y <= min(ty + vy, nty - 1) && !any; ++y)
        for (int x = max(tx - hx, 0); x <= min(tx + hx, g.ntx - 1) && !any; ++x)
            any = flags[p.tiles_off + y * g.ntx + x] != 0;
    need[p.tiles_off + ty * g.ntx + tx] = any ? 1 : 0;
}

// The work list.  An item = (record, pair of adjacent tile columns); every item becomes one
// workgroup per channel and level group.  A seam's active tiles span a few tile columns that
// start at any parity and drift with the row, so (a) pairs open at every active column not
// yet covered, not at fixed even positions, and (b) items differ a lot in length (cfg3: 13 to
// 82 bands), which is why they are sorted: dispatched longest first, the strips pack the CUs
// (in blockIdx order of a (pairs, records) grid the kernel ran at 60 % occupancy).
//   item.x = record | first tile column << 16,  item.y = bands the workgroup will process
#define MB_SORT_BINS 2048
__global__ __launch_bounds__(256) void mb_items_kernel(const pano_patch *__restrict__ table,
                                                       const uint8_t *__restrict__ flags,
                                                       int2 *__restrict__ items,
                                                       int *__restrict__ counter, int cap) {
    __shared__ int16_t s_start[1024];
    __shared__ unsigned long long s_any[4];
    __shared__ int s_base;
    const pano_patch p = table[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (p.aw <= 0 || p.ah <= 0) return;
    const MbGeom g = mb_geom(p);
    const int nty = g.O1 - g.O0 + 1;
    const uint8_t *rec = flags ? flags + p.tiles_off : nullptr;
    // the record's flags once into LDS (when they fit): the scans below read every flag half a
    // dozen times in short dependent loads - 25 us alone, 130 us beside the warp
    __shared__ uint8_t s_flags[16384];
    if (rec && nty * g.ntx <= (int)sizeof(s_flags)) {
        for (int i = tid; i < nty * g.ntx; i += 256) s_flags[i] = rec[i];
        __syncthreads();
        rec = s_flags;
    }
    int count = 0;
    bool covered = false;                                // column 0 of this chunk is in a pair
    for (int c0 = 0; c0 < g.ntx; c0 += 64) {
        // which columns of the chunk hold an active tile: the four waves take the rows in turn
        const int tx = c0 + lane;
        bool any = false;
        if (tx < g.ntx) {
            any = rec == nullptr;
            // (no early exit: stopping at the first hit makes the loads a dependent chain)
            if (rec)
#pragma unroll 4
                for (int ty = wave; ty < nty; ty += 4) any |= rec[ty * g.ntx + tx] != 0;
        }
        const unsigned long long mine = __ballot(any);
        __syncthreads();                                 // s_any of the previous chunk is read
        if (lane == 0) s_any[wave] = mine;
        __syncthreads();
        unsigned long long m = s_any[0] | s_any[1] | s_any[2] | s_any[3];
        if (covered) m &= ~1ull;
        covered = false;
        while (m) {                                      // uniform across the block
            const int b = __ffsll((long long)m) - 1;
            if (tid == 0 && count < 1024) s_start[count] = (int16_t)(c0 + b);
            ++count;
            m &= ~(3ull << b);
            covered = b == 63;
        }
    }
    count = count < 1024 ? count : 1024;                 // 65536 columns: beyond any mosaic
    if (tid == 0) s_base = atomicAdd(counter, count);
    __syncthreads();
    const int base = s_base;
    // One wave per pair: bands within two tiles of a wanted tile of the pair (what the kernel
    // steps through).  Lane = band; its five rows' flags are independent loads.
    for (int k = wave; k < count; k += 4) {
        const int tx0 = s_start[k];
        const bool two = tx0 + 1 < g.ntx;
        int bands = 0;
        for (int r0 = 0; r0 < nty; r0 += 64) {
            const int ty = r0 + lane;
            bool reach = false;
            if (ty < nty) {
                reach = rec == nullptr;
                for (int d = -2; d <= 2 && rec; ++d) {
                    const int y = ty + d;
                    if (y < 0 || y >= nty) continue;
                    reach |= rec[y * g.ntx + tx0] != 0;
                    if (two) reach |= rec[y * g.ntx + tx0 + 1] != 0;
                }
            }
            bands += __popcll(__ballot(reach));
        }
        // the kernel also walks DMAX bands past either end of the strip
        if (lane == 0 && base + k < cap)
            items[base + k] = make_int2((int)blockIdx.x | (tx0 << 16), bands + 4);
    }
}

// Counting sort of the items by decreasing length (one workgroup); slots past the last item
// get record -1.  Resets the counter for the next launch.
// Few items - one GPU's column strip of a panorama, a small scene - leave most CUs idle while
// every workgroup marches its whole column: each item is then cut into S vertical segments.
// S minimises a two-term model of the launch, in bands: the longest workgroup,
// ceil(Lmax / S) + lead, against the work per workgroup slot, (sum of lengths + n S lead)
// wgs_per_item / slots, where `lead` = the 2 DMAX bands a segment steps through outside its
// rows plus the copy of the operand tables into LDS (MB_SEG_LEAD bands in all, an estimate).
// S > 1 only when the model gains 15 %; at most MB_SEG_MAX; the sorted list has
// `scap` >= cap + MB_SEG_SLOTS slots, which n S never exceeds.  A segment's entry is
// (item.x, length | segment << 12 | S << 16); the kernel keeps the need flags of its rows only.
#define MB_SEG_MAX 8
#define MB_SEG_LEAD 8
#define MB_SEG_SLOTS 192
__global__ __launch_bounds__(256) void mb_sort_kernel(const int2 *__restrict__ items,
                                                      int *__restrict__ counter, int cap, int scap,
                                                      int wgs_per_item, int slots,
                                                      int2 *__restrict__ sorted,
                                                      const pano_patch *__restrict__ table, int cm,
                                                      int force_t) {
    __shared__ int s_hist[MB_SORT_BINS];
    __shared__ int s_lmax, s_lsum, s_total;
    const int tid = threadIdx.x;
    const int n = min(*counter, cap);
    (void)table, (void)cm;
    // T: the segment length (bands) items are cut to, 0 = nothing is cut.  Items differ in length
    // by a factor of six (13 - 82 bands on config 3): one segment COUNT for all of them (rounds
    // 2 - 4) cut the short items into segments that were mostly lead while the long ones still set
    // the launch's length; one segment LENGTH gives a long item many segments and a short one none.
    // Which length: the workgroups are dispatched longest first onto `slots` one-per-CU slots, so
    // a launch lasts as long as that list schedule, and a lower bound (the longest workgroup, the
    // work per slot) ranks the candidates wrongly exactly where it matters - a few more workgroups
    // than slots make a second round (a world-8 strip of config 3, PANO_BLUR_SEG_T forced: 29 bands,
    // the bound's choice, 0.135 ms; 20 or 48 bands 0.100; profiles/r05/blur_seg_t_strip8.txt).
    // Every candidate therefore gets the histogram of its workgroups' lengths and an estimate of
    // the schedule from its order statistics l_1 >= l_2 >= ...: one round lasts l_1; two rounds
    // pair the k-th workgroup of the second with the slot that frees k-th (the shortest of the
    // first round first): max(l_1, l_slots + l_slots+1, l_2 slots + 1 - W + l_W); more rounds:
    // the work per slot plus half the shortest workgroup.
    int T = 0;
    constexpr int NCAND = 13, HL = 128;                          // candidate 0 = nothing cut
    constexpr int MINE = 4;                                      // items a thread keeps in registers
    __shared__ int s_segs[NCAND], s_est[NCAND];
    __shared__ __align__(16) unsigned short s_len[NCAND][HL];    // workgroup lengths (bands + lead), in segments
    // (with three rounds of workgroups and more nothing is gained by cutting: the launch is bound by
    // its work per slot, which segments only add to - and the estimate below would cost a large
    // launch's sort kernel tens of microseconds for nothing)
    const bool consider = slots > 0 && n > 0 && !force_t && (long long)n * wgs_per_item < 3ll * slots &&
                          n <= MINE * 256;
    if (slots > 0 && n > 0) {                                    // uniform
        if (tid == 0) s_lmax = s_lsum = 0;
        if (tid < NCAND) s_segs[tid] = 0, s_est[tid] = 0x7fffffff;
        for (int i = tid; i < NCAND * HL / 2; i += 256) ((unsigned int *)&s_len[0][0])[i] = 0u;
        __syncthreads();
        int lmax = 0, lsum = 0, mine[MINE];
        // (static indices only: a `mine[r]` with a running r put the array into scratch memory and
        // every one of the estimate's 52 reads of it became a trip to memory - 30 us of this kernel)
#pragma unroll
        for (int r = 0; r < MINE; ++r) {
            const int i = tid + 256 * r;
            mine[r] = 0;
            if (i < n) {
                const int2 it = items[i];
                lmax = max(lmax, it.y);
                lsum += it.y;
                mine[r] = it.y;
            }
        }
        for (int i = tid + 256 * MINE; i < n; i += 256) {
            const int2 it = items[i];
            lmax = max(lmax, it.y);
            lsum += it.y;
        }
        atomicMax(&s_lmax, lmax);
        atomicAdd(&s_lsum, lsum);
        __syncthreads();
        const int Lmax = s_lmax, Lsum = s_lsum;
        // candidate lengths: none, Lmax / 2 ... Lmax / 8 and a few absolute ones
        auto cand = [&](const int c) {
            if (c == 0) return 1 << 20;
            // ~ Lmax / 2 ... Lmax / 8 (multipliers 65536 / (c + 1) + 1: no division), then 16 ... 48
            const int mul = c == 1 ? 32769 : c == 2 ? 21846 : c == 3 ? 16385 : c == 4 ? 13108
                          : c == 5 ? 10923 : c == 6 ? 9363 : 8193;
            const int t = c < 8 ? (int)((unsigned)((Lmax + c) * mul) >> 16) : 8 * (c - 6);
            return max(t, 8);
        };
        // floor(a / b) for 0 <= a < 2^22, b >= 1 without the integer-division sequence (a dozen of
        // them per candidate, one behind the other, were 28 us of this kernel): a float estimate
        // that is off by at most one, put right by two compares
        auto div_small = [](const int a, const int b, const float rcp_b) {
            int q = (int)((float)a * rcp_b);
            q -= q * b > a ? 1 : 0;
            q += (q + 1) * b <= a ? 1 : 0;
            return q;
        };
        if (consider) {
            // Up to 64 MINE items (a strip, a small scene): every wave holds ALL the items, a lane
            // MINE of them, and takes every fourth candidate - three or four trips instead of
            // thirteen, one behind the other.  More items: a thread keeps its own, all candidates.
            const bool by_wave = n <= 64 * MINE;
            const int lane = tid & 63, wave = tid >> 6;
            if (by_wave) {
#pragma unroll
                for (int r = 0; r < MINE; ++r) {
                    const int i = lane + 64 * r;
                    mine[r] = i < n ? items[i].y : 0;
                }
            }
            const int rounds = by_wave ? (n + 63) >> 6 : (n + 255) >> 8;      // uniform: slots of `mine` in use
#pragma unroll 1
            for (int c = by_wave ? wave : 0; c < NCAND; c += by_wave ? 4 : 1) {
                const int t = cand(c);
                const float rcp_t = 1.0f / (float)t;
                int segs = 0;
#pragma unroll
                for (int r = 0; r < MINE; ++r) {
                    if (r >= rounds) break;
                    const int len = mine[r];
                    if (len > 0) {
                        const int ns = min(MB_SEG_MAX, div_small(len + t - 1, t, rcp_t));
                        const int wl = min(div_small(len + ns - 1, ns, 1.0f / (float)ns) + MB_SEG_LEAD, HL - 1);
                        segs += ns;
                        atomicAdd((unsigned int *)&s_len[c][wl & ~1], (unsigned int)ns << (16 * (wl & 1)));
                    }
                }
                // one add per wave
                for (int off = 32; off > 0; off >>= 1) segs += __shfl_xor(segs, off, 64);
                if (lane == 0 && segs) atomicAdd(&s_segs[c], segs);
            }
        }
        __syncthreads();
        // order statistics of a candidate's workgroup lengths, longest first: l(k), k = 1 .. W.  A
        // candidate is a group of 16 threads, a thread one chunk of eight bins (one 16-byte LDS read):
        // the chunks' totals are prefix-summed down from the longest across the group, and a thread
        // resolves the ranks that fall into its own chunk
        __shared__ int s_got[NCAND][4], s_lmin[NCAND];
        if (tid < NCAND) {
            s_got[tid][0] = s_got[tid][1] = s_got[tid][2] = s_got[tid][3] = 0;
            s_lmin[tid] = HL;
        }
        __syncthreads();
        static_assert(HL == 16 * 8 && NCAND * 16 <= 256, "16 chunks of eight bins per candidate");
        if (consider && tid < NCAND * 16) {
            const int c = tid >> 4, chunk = 15 - (tid & 15);             // lane 0 of a group: the longest bins
            const int W = s_segs[c] * wgs_per_item;
            const uint4 q = *(const uint4 *)&s_len[c][8 * chunk];
            const unsigned int word[4] = {q.x, q.y, q.z, q.w};
            int cnt[8], total = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                cnt[e] = (int)((word[e >> 1] >> (16 * (e & 1))) & 0xffffu) * wgs_per_item;
                total += cnt[e];
            }
            int seen = total;                                            // inclusive prefix over the group
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                const int v = __shfl_up(seen, off, 16);
                if ((tid & 15) >= off) seen += v;
            }
            seen -= total;                                               // workgroups longer than this chunk's
            if (total) {
                const int want[4] = {1, slots, slots + 1, 2 * slots + 1 - W};
                int lmin = HL;
#pragma unroll
                for (int e = 7; e >= 0; --e) {
                    if (!cnt[e]) continue;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (want[u] > seen && want[u] <= seen + cnt[e]) s_got[c][u] = 8 * chunk + e;
                    seen += cnt[e];
                    lmin = 8 * chunk + e;
                }
                atomicMin(&s_lmin[c], lmin);
            }
        }
        __syncthreads();
        if (consider && tid < NCAND) {
            const int c = tid, t = cand(c), segs = s_segs[c];
            int est = 0x7fffffff;
            if (c == 0 || (t < Lmax && segs > n && segs <= scap && segs <= n + MB_SEG_SLOTS)) {
                const int W = segs * wgs_per_item, lmin = s_lmin[c];
                const int *got = s_got[c];
                const int share = (int)(((long long)Lsum + (long long)segs * MB_SEG_LEAD) *
                                        wgs_per_item / slots);
                if (W <= slots)
                    est = got[0];
                else if (W <= 2 * slots)
                    est = max(max(got[0], got[1] + got[2]), max(got[3] + lmin, share));
                else
                    est = max(got[0], share + lmin / 2);
            }
            s_est[c] = est;
        }
        __syncthreads();
        const int base = s_est[0];
        int best = base;
        for (int c = 1; c < NCAND && consider; ++c) {
            const int m = s_est[c];
            if (m < best && (long long)m * 100 <= (long long)base * 85) {
                best = m;
                T = cand(c);
            }
        }
        if (force_t) {                                          // (PANO_OPT_BLUR_SEG_LEN)
            T = 0;
            if (force_t >= 4) {
                __syncthreads();
                if (tid == 0) s_segs[0] = 0;
                __syncthreads();
                int segs = 0;
                for (int i = tid; i < n; i += 256) segs += min(MB_SEG_MAX, (items[i].y + force_t - 1) / force_t);
                if (segs) atomicAdd(&s_segs[0], segs);
                __syncthreads();
                if (s_segs[0] <= scap && s_segs[0] <= n + MB_SEG_SLOTS) T = force_t;   // the list holds them
            }
        }
        __syncthreads();
    }
    // segments of item i: its length over T
    auto segments_of = [&](const int2 it) {
        return T ? min(MB_SEG_MAX, (it.y + T - 1) / T) : 1;
    };
    auto entry = [&](const int2 it, const int seg, const int ns) {
        const int len = ns == 1 ? it.y : (it.y + ns - 1) / ns + 4;
        return make_int2(it.x, min(len, MB_SORT_BINS - 1) | seg << 12 | (ns == 1 ? 0 : ns) << 16);
    };
    for (int i = tid; i < MB_SORT_BINS; i += 256) s_hist[i] = 0;
    if (tid == 0) s_total = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const int2 it = items[i];
        const int ns = segments_of(it);
        for (int seg = 0; seg < ns; ++seg)
            atomicAdd(&s_hist[MB_SORT_BINS - 1 - (entry(it, seg, ns).y & 0xfff)], 1);   // long first
        atomicAdd(&s_total, ns);
    }
    __syncthreads();
    const int total = s_total;
    // exclusive prefix over the bins: eight bins per thread, then a scan of the 256 sums
    __shared__ int s_part[256];
    constexpr int PER = MB_SORT_BINS / 256;
    int mine[PER], sum = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        mine[j] = s_hist[tid * PER + j];
        sum += mine[j];
    }
    s_part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {            // Hillis-Steele, inclusive
        const int v = tid >= off ? s_part[tid - off] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int run = s_part[tid] - sum;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        s_hist[tid * PER + j] = run;
        run += mine[j];
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const int2 it = items[i];
        const int ns = segments_of(it);
        for (int seg = 0; seg < ns; ++seg) {
            const int2 e = entry(it, seg, ns);
            sorted[atomicAdd(&s_hist[MB_SORT_BINS - 1 - (e.y & 0xfff)], 1)] = e;
        }
    }
    for (int i = total + tid; i < scap; i += 256) sorted[i] = make_int2(-1, 0);
    __syncthreads();
    if (tid == 0) *counter = 0;
}

// slots of the sorted list: the items, room for their segments, rounded up to an even count
// (4 channels x an even count = whole rounds of the 8 XCDs for the kernel's id mapping)
static inline int mb_sorted_slots(int cap) { return (cap + MB_SEG_SLOTS + 1) & ~1; }

// Host side: called by pano_multiband_blur (blur.hip).  taps / ntaps: the caller's
// padded tables (include/pano360.h); `extra[k]` zeros precede level k's first tap
// after the PANO_TAP_LEAD ones.
#ifdef MB_STAMP
extern "C" int pano_debug_stamps(unsigned long long *out, int reset) {
    if (out) {
        PANO_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mb_stamps), sizeof(unsigned long long) * 48));
        PANO_HIP(hipMemcpyFromSymbol(out + 48, HIP_SYMBOL(g_mb_setup), sizeof(unsigned long long) * 8));
    }
    if (reset) {
        unsigned long long zero[48] = {};
        PANO_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_mb_stamps), zero, sizeof(zero)));
        PANO_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_mb_setup), zero, sizeof(unsigned long long) * 8));
    }
    return PANO_OK;
}
#endif

// The work list of a table of records: tile flags (with an interior map), items, sort.
// Depends on the records' geometry and the interior map only, not on the warped planes, so
// the caller may queue it on another stream while the warp runs (pano_multiband_blur_prepare).
// The buffers and the note of whose list / flags they hold belong to the context.
static int launch_tile_flags(pano_ctx *ctx, const pano_patch *table, int n, int ntx_max,
                             int nty_max, int W, const uint8_t *interior, uint8_t *tile_flags) {
    const hipStream_t stream = ctx->stream;
    dim3 grid(ceil_div(ntx_max, 32), ceil_div(nty_max, 8), n);
    PANO_TIMED(PK_TILE_FLAGS, stream,
               hipLaunchKernelGGL(tile_flags32_kernel, grid, dim3(256), 0, stream, table, interior,
                                  ceil_div(W, PANO_INTERIOR_BLOCK), tile_flags));
    PANO_LAUNCH_CHECK("tile_flags32_kernel");
    return PANO_OK;
}

// Tile flags and, from them, the tiles of V the warp has to fill; the following
// pano_prepare_blur_mfma / pano_launch_blur_mfma on the same table reuse the flags.
int pano_tiles_blur_mfma(pano_ctx *ctx, const pano_patch *table, int n, int max_aw, int max_ah,
                         int W, int radius, const uint8_t *interior, uint8_t *tile_flags,
                         uint8_t *warp_need) {
    const hipStream_t stream = ctx->stream;
    const int ntx_max = (max_aw + 62) / 32, nty_max = (max_ah + 62) / 32;
    if (int rc = launch_tile_flags(ctx, table, n, ntx_max, nty_max, W, interior, tile_flags))
        return rc;
    ctx->flags_table = table;
    ctx->flags_n = n;
    ctx->blur_cm = (radius + 15) / 16 < 1 ? 1 : (radius + 15) / 16;     // (for the work list's segments)
    if (warp_need) {
        const int cm = (radius + 15) / 16 < 1 ? 1 : (radius + 15) / 16;
        const int hx = (16 * cm + 31) / 32 + 1, vy = (cm + 1) / 2;
        dim3 grid(ceil_div(ntx_max, 32), ceil_div(nty_max, 8), n);
        hipLaunchKernelGGL(warp_need_kernel, grid, dim3(256), 0, stream, table, tile_flags, hx, vy,
                           warp_need);
        PANO_LAUNCH_CHECK("warp_need_kernel");
    }
    return PANO_OK;
}

int pano_prepare_blur_mfma(pano_ctx *ctx, const pano_patch *table, int n, int max_aw, int max_ah,
                           int W, const uint8_t *interior, uint8_t *tile_flags) {
    const hipStream_t stream = ctx->stream;
    const int ntx_max = (max_aw + 62) / 32, nty_max = (max_ah + 62) / 32;
    PANO_REQUIRE(nty_max <= MB_NEED_MAX, "pano_multiband_blur: %d rows of tiles exceed %d", nty_max,
                 MB_NEED_MAX);
    const uint8_t *flags = nullptr;
    if (interior) {
        if (ctx->flags_table != table || ctx->flags_n != n)
            if (int rc = launch_tile_flags(ctx, table, n, ntx_max, nty_max, W, interior, tile_flags))
                return rc;
        ctx->flags_table = nullptr;
        flags = tile_flags;
    }
    // at most ceil(ntx / 2) pairs per record
    const int cap = n * ceil_div(ntx_max, 2);
    if (cap > ctx->item_cap) {
        if (ctx->item_buf) {
            PANO_HIP(hipStreamSynchronize(stream));          // a queued blur may still read it
            PANO_HIP(hipFree(ctx->item_buf));
            ctx->item_buf = nullptr;
        }
        ctx->item_cap = cap * 2;
        PANO_HIP(hipMalloc((void **)&ctx->item_buf,
                           ((size_t)ctx->item_cap * 2 + MB_SEG_SLOTS + 2) * sizeof(int2)));
    }
    if (!ctx->item_counter) {
        PANO_HIP(hipMalloc((void **)&ctx->item_counter, sizeof(int)));
        PANO_HIP(hipMemsetAsync(ctx->item_counter, 0, sizeof(int), stream));
    }
    hipLaunchKernelGGL(mb_items_kernel, dim3(n), dim3(256), 0, stream, table, flags, ctx->item_buf,
                       ctx->item_counter, cap);
    PANO_LAUNCH_CHECK("mb_items_kernel");
    // (option PANO_OPT_BLUR_SEG_LEN: a segment length in bands, -1 = no cut, 0 = the estimate's choice)
    const int force_t = ctx->opt[PANO_OPT_BLUR_SEG_LEN];
    // 4 channels (x level groups) workgroups per item
    hipLaunchKernelGGL(mb_sort_kernel, dim3(1), dim3(256), 0, stream, ctx->item_buf,
                       ctx->item_counter, cap, mb_sorted_slots(cap), 4,
                       ctx->opt[PANO_OPT_BLUR_SEGMENTS] ? 256 : 0,
                       ctx->item_buf + ctx->item_cap, table,
                       ctx->opt[PANO_OPT_BLUR_SEGMENTS] ? (ctx->blur_cm > 0 ? ctx->blur_cm : 3) : 0,
                       force_t);
    PANO_LAUNCH_CHECK("mb_sort_kernel");
    ctx->prepared_table = ctx->list_table = table;
    ctx->prepared_n = ctx->list_n = n;
    return PANO_OK;
}

// 116 KiB of Toeplitz tables per workgroup (once per device; pano_ctx_create)
int pano_blur_mfma_opt_in(void) {
    PANO_HIP(hipFuncSetAttribute((const void *)blur_mfma_kernel<2>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PANO_HIP(hipFuncSetAttribute((const void *)blur_mfma_kernel<4>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PANO_HIP(hipFuncSetAttribute((const void *)blur_lean_kernel,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PANO_HIP(hipFuncSetAttribute((const void *)blur_lean5_kernel,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return PANO_OK;
}

// One launch for the levels [lev0, lev0 + cnt) of the caller's tap tables, `group` of them per
// workgroup.  rmax_all: the largest radius of ALL the caller's levels (it fixes the zeros in
// front of each padded table, include/pano360.h).
static int launch_levels(pano_ctx *ctx, const pano_patch *table, int n, int max_aw,
                         const int16_t *owner, int W, const float *host_taps_all,
                         const int *ntaps_all, int lev0, int cnt, int rmax_all, int group,
                         const uint8_t *flags) {
    const hipStream_t stream = ctx->stream;
    const int *ntaps = ntaps_all + lev0;
    size_t skip = 0;
    for (int k = 0; k < lev0; ++k) skip += (size_t)ntaps_all[k] + PANO_TAP_PAD;
    const float *host_taps = host_taps_all + skip;
    // Work order of the levels: a workgroup takes `group` consecutive entries.  With two per
    // workgroup the heaviest level goes with the lightest, the second heaviest with the second
    // lightest, ...: the groups' tables then have about the same size, and two workgroups fit
    // the LDS of a CU.  (Apertures grow with the level: stitcher.py:218.)
    int ord[PANO_MAX_LEVELS];
    if (group == 2) {
        int by_size[PANO_MAX_LEVELS];
        for (int k = 0; k < cnt; ++k) by_size[k] = k;
        for (int a = 1; a < cnt; ++a)                         // insertion sort, largest aperture first
            for (int b = a; b > 0 && ntaps[by_size[b]] > ntaps[by_size[b - 1]]; --b) {
                const int tmp = by_size[b];
                by_size[b] = by_size[b - 1];
                by_size[b - 1] = tmp;
            }
        for (int i = 0, lo = 0, hi = cnt - 1; i < cnt; ++i)
            ord[i] = (i & 1) ? by_size[hi--] : by_size[lo++];
    } else {
        for (int k = 0; k < cnt; ++k) ord[k] = k;
    }
    MbLevels L = {};
    L.n = cnt;
    int total = 0;
    for (int i = 0; i < cnt; ++i) {
        L.tab_off[i] = total;
        total += mb_table_bytes(ntaps[ord[i]]);
    }
    // device copy of the taps and the Toeplitz operand tables of this tap set: kept by the
    // context, keyed on the tap values, built on first use in stream order
    PanoTapSet *set = nullptr;
    bool fresh = false;
    if (int rc = pano_ctx_tap_set(ctx, host_taps, ntaps, cnt, (size_t)total, &set, &fresh))
        return rc;
    size_t first[PANO_MAX_LEVELS], off = 0;
    for (int k = 0; k < cnt; ++k) {
        first[k] = off + PANO_TAP_LEAD + ((rmax_all - ntaps[k] / 2) & 3);
        off += (size_t)ntaps[k] + PANO_TAP_PAD;
    }
    for (int i = 0; i < cnt; ++i) {
        L.w[i] = set->taps + first[ord[i]];
        L.ntaps[i] = ntaps[ord[i]];
        L.out[i] = lev0 + ord[i];
    }
    unsigned char *tables = set->tables;
    if (fresh) {
        hipLaunchKernelGGL(mb_tables_kernel, dim3(cnt), dim3(128), 0, stream, L, tables);
        PANO_LAUNCH_CHECK("mb_tables_kernel");
        if (int rc = pano_ctx_tap_set_built(ctx, set)) return rc;
    }
    const int ntx_max = (max_aw + 62) / 32;
    const int cap = mb_sorted_slots(n * ceil_div(ntx_max, 2));    // slots of the sorted list
    const int2 *sorted = ctx->item_buf + ctx->item_cap;
    // dynamic LDS: the largest level group's band, flags and tables
    const int ngroups = ceil_div(cnt, group);
    int lds = 0;
    for (int gidx = 0; gidx < ngroups; ++gidx) {
        int bytes = 0, cm = 1;
        for (int i = group * gidx; i < cnt && i < group * (gidx + 1); ++i) {
            bytes += mb_table_bytes(L.ntaps[i]);
            cm = mb_c_of(L.ntaps[i]) > cm ? mb_c_of(L.ntaps[i]) : cm;
        }
        bytes += mb_fixed_bytes(cm);
        lds = bytes > lds ? bytes : lds;
    }
    PANO_REQUIRE(lds <= 160 * 1024, "pano_multiband_blur: %d bytes of LDS tables", lds);
    PANO_REQUIRE((long long)cap * 4 * ngroups < (1ll << 31), "pano_multiband_blur: %d work items",
                 cap);
    dim3 grid((unsigned)cap * 4 * ngroups, 1, 1);
    // Groups of four levels with real apertures: the kernel that runs the regular items
    // through the lean path (and the others through the general one).
    int lean = MB_LEAN && ctx->opt[PANO_OPT_BLUR_LEAN] && group == 4 ? 1 : 0;
    // five levels in one group (blur_lean_kernel: the two lightest on one wave pair), when their
    // reaches allow: 1 and 2 K-steps (equal DMAX) for the lightest two
    const bool five = lean && cnt == 5 && mb_c_of(L.ntaps[0]) == 1 &&
                      mb_c_of(L.ntaps[1]) == 2;
    if (cnt > 4 && !five) lean = lean && false;          // (more than one group: the general kernel)
    for (int i = 0; i < cnt; ++i)
        if (L.ntaps[i] < 3) lean = 0;
    int lds_lean = lds;
    for (int i = 0; i < cnt; ++i)    // (ms_body stages at most 1280 chunks per band: a reach of 3 K-steps)
        if (mb_c_of(L.ntaps[i]) > 3) lean = 0;
    if (ML_OVERLAP) {            // + the second band buffer, behind the largest group's tables
        lds_lean = 0;
        for (int gidx = 0; gidx < (five ? 1 : ngroups); ++gidx) {
            int bytes = 0, cm = 1;
            for (int i = group * gidx; i < cnt && (five || i < group * (gidx + 1)); ++i) {
                bytes += mb_table_bytes(L.ntaps[i]);
                cm = mb_c_of(L.ntaps[i]) > cm ? mb_c_of(L.ntaps[i]) : cm;
            }
            bytes = (bytes + mb_fixed_bytes(cm) + 15) / 16 * 16 + 2 * 32 * mb_pitch_of(cm) * 2;
            lds_lean = bytes > lds_lean ? bytes : lds_lean;
        }
        if (lds_lean > 160 * 1024) lean = 0;             // (apertures above 97 taps: the general kernel)
    }
    if (lean) {
        if (five) {
            grid = dim3((unsigned)cap * 4, 1, 1);        // one group
            PANO_TIMED(PK_BLUR_LEAN5, stream,
                       hipLaunchKernelGGL(blur_lean5_kernel, grid, dim3(MB_THREADS_OF(4)), lds_lean,
                                          stream, table, L, tables, owner, W, flags, sorted));
            PANO_LAUNCH_CHECK("blur_lean5_kernel");
        } else {
            PANO_TIMED(PK_BLUR_LEAN, stream,
                       hipLaunchKernelGGL(blur_lean_kernel, grid, dim3(MB_THREADS_OF(4)), lds_lean,
                                          stream, table, L, tables, owner, W, flags, sorted));
            PANO_LAUNCH_CHECK("blur_lean_kernel");
        }
        return PANO_OK;
    }
    if (group == 2)
        PANO_TIMED(PK_BLUR_MFMA, stream,
                   hipLaunchKernelGGL(blur_mfma_kernel<2>, grid, dim3(MB_THREADS_OF(2)), lds, stream,
                                      table, L, tables, owner, W, flags, sorted));
    else
        PANO_TIMED(PK_BLUR_MFMA, stream,
                   hipLaunchKernelGGL(blur_mfma_kernel<4>, grid, dim3(MB_THREADS_OF(4)), lds, stream,
                                      table, L, tables, owner, W, flags, sorted));
    PANO_LAUNCH_CHECK("blur_mfma_kernel");
    return PANO_OK;
}

int pano_launch_blur_mfma(pano_ctx *ctx, const pano_patch *table, int n, int max_aw, int max_ah,
                          const int16_t *owner, int W, const float *host_taps, const int *ntaps,
                          int n_blur, const uint8_t *interior, uint8_t *tile_flags) {
    // Levels per workgroup.  Four (eight waves in lockstep, one workgroup per CU) shares one
    // staged band between four levels; two (four waves, two workgroups per CU running out of
    // step) stages every band twice as often.  Measured (profiles/r02/notes.md): four levels -
    // the reference's default - 0.87 ms as one group of four against 1.00 ms as two groups of
    // two (config 3); five levels in ONE launch 10.9 ms as 4 + 1 (the lone level's workgroup
    // reserves the LDS of four) against 9.4 ms as 2 + 2 + 1 (config 5).  Two per workgroup only
    // while every level fits 3 K-steps either side (the narrow band pitch; apertures up to 97).
    bool narrow = true;
    int rmax = 0;
    for (int k = 0; k < n_blur; ++k) {
        if (mb_c_of(ntaps[k]) > 3) narrow = false;
        rmax = ntaps[k] / 2 > rmax ? ntaps[k] / 2 : rmax;
    }
    // the work list: prepared by the caller for this table, or made here
    if (ctx->prepared_table != table || ctx->prepared_n != n)
        if (int rc = pano_prepare_blur_mfma(ctx, table, n, max_aw, max_ah, W, interior, tile_flags))
            return rc;
    ctx->prepared_table = nullptr;
    const uint8_t *flags = interior ? tile_flags : nullptr;
#ifndef MB_SPLIT_LAUNCH
#define MB_SPLIT_LAUNCH 1
#endif
#ifndef MB_FIVE_IN_ONE
#define MB_FIVE_IN_ONE 1
#endif
    if (narrow && n_blur == 5 && ctx->opt[PANO_OPT_BLUR_LEAN] &&
        mb_c_of(ntaps[0]) == 1 && mb_c_of(ntaps[1]) == 2 && ntaps[0] >= 3 && MB_FIVE_IN_ONE)
        // six pyramid levels: all five Gaussian levels in ONE launch (every band staged once)
        return launch_levels(ctx, table, n, max_aw, owner, W, host_taps, ntaps, 0, 5, rmax, 4, flags);
    if (MB_SPLIT_LAUNCH && narrow && (n_blur == 5 || n_blur == 6)) {
        // four levels as one group of four, the rest in a launch of their own (small workgroups,
        // several per CU): the band is staged twice per item instead of three times.  Config 5
        // (five levels): 8.9-9.05 ms against 9.2 ms as 2 + 2 + 1 in one launch.
        // The two launches touch different planes and fill the chip differently - eight-wave
        // workgroups, one per CU, against four-wave ones of which a CU holds two - so the second
        // goes to the context's side stream: on one GPU it fills the first one's tail, on a column
        // strip of an 8-GPU run (a few dozen work items, far fewer workgroups than CUs) the two
        // run side by side.
        // (not while kernels are being timed: the two launches' events would span each other)
        const bool beside = ctx->opt[PANO_OPT_STITCH_STREAMS] != 0 && !ctx->timing_on;
        const hipStream_t main_stream = ctx->stream;
        if (beside) {
            if (int rc = pano_ctx_side_stream(ctx)) return rc;
            PANO_HIP(hipEventRecord(ctx->ev_fork, main_stream));
            PANO_HIP(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
        }
        if (int rc = launch_levels(ctx, table, n, max_aw, owner, W, host_taps, ntaps, 0, 4, rmax, 4,
                                   flags))
            return rc;
        if (beside) ctx->stream = ctx->side;
        const int rc2 = launch_levels(ctx, table, n, max_aw, owner, W, host_taps, ntaps, 4, n_blur - 4,
                                      rmax, 2, flags);
        ctx->stream = main_stream;
        if (rc2) return rc2;
        if (beside) {
            PANO_HIP(hipEventRecord(ctx->ev_join, ctx->side));
            PANO_HIP(hipStreamWaitEvent(main_stream, ctx->ev_join, 0));
        }
        return PANO_OK;
    }
    const int group = narrow && (n_blur <= 2 || n_blur == 5 || n_blur == 6) ? 2 : 4;
    return launch_levels(ctx, table, n, max_aw, owner, W, host_taps, ntaps, 0, n_blur, rmax, group,
                         flags);
}
