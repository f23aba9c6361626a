// Exhaustive two-nearest-neighbour search between two descriptor sets, the arithmetic
// behind `flann_matching` (features.py:222-232: FLANN's kd-trees asked for k = 2, then
// Lowe's ratio test).  An exact search replaces FLANN's approximate one.
//
// This is the one dense contraction of the whole path - |q - t|^2 = |q|^2 + |t|^2 - 2 q.t
// with the cross terms a [queries x train] matrix product - so it runs on the matrix cores:
// v_mfma_f32_32x32x16_f16 with every operand split into float16 hi + lo (hi.hi + hi.lo +
// lo.hi, float32 accumulate: ~2^-21 relative, i.e. float32-GEMM accuracy).  The product
// only RANKS the candidates: the four best per query are then re-evaluated exactly in
// float32 (sum of squared differences in a fixed order), and a bound on the product's error
// proves per query that no other row can beat the second of them; a query for which the
// proof fails (near ties) is rescanned exactly.  So the answer is the exact one.
//
// Layout: both sets are packed once into MFMA fragment order (tile of 32 rows, k-step of 16
// features, hi / lo, lane: 8 halves) so that every operand read is one aligned 16-byte
// access.  A wave keeps 32 queries' fragments in registers (the B operand: the result tile
// then has the query on the lane and 16 train rows in the lane's registers, so the running
// candidate list is private to a lane) and streams the train tiles through LDS, shared by the 4 waves
// of the workgroup and double-buffered.
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define KNN_MAX_KS 8                  // k-steps of 16 features: d <= 128
// error bound of a ranked value, relative to |q|^2 + max |t|^2: 384 float32 accumulations
// (<= 384 * 2^-24 of sum |q_i t_i| <= half of |q|^2 + |t|^2), twice for the factor -2, plus
// the split's 3 * 2^-22 and the norms' own rounding: 2.6e-5; taken as 4e-5
#define KNN_EPS 4.0e-5f
#define KNN_KEEP 4                    // candidates kept per query: the proof compares the exact
                                      // second with the ranked value of the last one kept

__device__ __forceinline__ void knn_split(float v, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

// rows [n][d] float -> fragments [tile][k-step][hi, lo][lane] half8 (values scaled), zero
// beyond n / d.  One thread per (tile, k-step, lane).
__global__ __launch_bounds__(256) void knn_pack_kernel(const float *__restrict__ src, int n, int d,
                                                       int ks, float scale,
                                                       half8 *__restrict__ packed) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = (int)(i & 63);
    const size_t ts = i >> 6;
    const int s = (int)(ts % ks);
    const size_t tile = ts / ks;
    const size_t row = tile * 32 + (lane & 31);
    if (tile * 32 >= (size_t)((n + 31) & ~31)) return;
    half8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 16 * s + 8 * (lane >> 5) + j;
        const float v = row < (size_t)n && k < d ? src[row * d + k] * scale : 0.0f;
        _Float16 a, b;
        knn_split(v, a, b);
        hi[j] = a;
        lo[j] = b;
    }
    packed[((tile * ks + s) * 2) * 64 + lane] = hi;
    packed[((tile * ks + s) * 2 + 1) * 64 + lane] = lo;
}

// |row * scale|^2 in float32, one wave per row; rows beyond n get +inf (never chosen);
// the largest norm goes to *maxnorm (float bits, non-negative: integer max).
__global__ __launch_bounds__(256) void knn_norm_kernel(const float *__restrict__ src, int n,
                                                       int n_pad, int d, float scale,
                                                       float *__restrict__ norms,
                                                       unsigned *__restrict__ maxnorm) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n_pad) return;
    float acc = 0.0f;
    if (row < n)
        for (int k = lane; k < d; k += 64) {
            const float v = src[(size_t)row * d + k] * scale;
            acc = __builtin_fmaf(v, v, acc);
        }
#pragma unroll
    for (int off = 32; off; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) {
        norms[row] = row < n ? acc : __builtin_inff();
        if (row < n && maxnorm) atomicMax(maxnorm, __float_as_uint(acc));
    }
}

struct KnnTop {
    float v[KNN_KEEP];                // ascending
    int i[KNN_KEEP];
};

__device__ __forceinline__ bool knn_before(float v, int j, float w, int k) {
    return v < w || (v == w && j < k);
}

__device__ __forceinline__ void knn_insert(KnnTop &t, float v, int j) {
    if (!knn_before(v, j, t.v[KNN_KEEP - 1], t.i[KNN_KEEP - 1])) return;
#pragma unroll
    for (int s = KNN_KEEP - 1; s >= 0; --s) {
        const bool up = s > 0 && knn_before(v, j, t.v[s - 1], t.i[s - 1]);
        t.v[s] = up ? t.v[s - 1] : v;
        t.i[s] = up ? t.i[s - 1] : j;
        if (!up) break;
    }
}

// cand_idx [nq][KNN_KEEP], cand_val [nq][KNN_KEEP] (scaled |t|^2 - 2 q.t, ascending)
template <int KS>
__global__ __launch_bounds__(256) void knn2_kernel(const half8 *__restrict__ pq,
                                                   const half8 *__restrict__ pt,
                                                   const float *__restrict__ norm_t, int nq, int nt,
                                                   int32_t *__restrict__ cand_idx,
                                                   float *__restrict__ cand_val) {
    __shared__ __attribute__((aligned(16))) half8 s_tile[2][KS * 2 * 64];
    __shared__ __attribute__((aligned(16))) float s_norm[2][32];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    const int tq = blockIdx.x * 4 + wv;                       // this wave's query tile
    const int ntq = (nq + 31) >> 5, ntt = (nt + 31) >> 5;
    const bool live = tq < ntq;
    half8 bq[KS][2];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int part = 0; part < 2; ++part)
            bq[s][part] = pq[(((size_t)(live ? tq : 0) * KS + s) * 2 + part) * 64 + lane];
    KnnTop top;
#pragma unroll
    for (int s = 0; s < KNN_KEEP; ++s) {
        top.v[s] = __builtin_inff();
        top.i[s] = 0x7fffffff;
    }
    constexpr int CHUNKS = KS * 2 * 64 / 256;                 // 16-byte pieces per thread and tile
    half8 next[CHUNKS];
    float next_norm = 0.0f;
    auto fetch = [&](int tt) {                                // global -> registers
        const half8 *src = pt + (size_t)tt * KS * 2 * 64;
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c) next[c] = src[tid + 256 * c];
        if (tid < 32) next_norm = norm_t[tt * 32 + tid];
    };
    auto commit = [&](int buf) {                              // registers -> LDS
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c) s_tile[buf][tid + 256 * c] = next[c];
        if (tid < 32) s_norm[buf][tid] = next_norm;
    };
    fetch(0);
    commit(0);
    __syncthreads();
    for (int tt = 0; tt < ntt; ++tt) {
        const int buf = tt & 1;
        if (tt + 1 < ntt) fetch(tt + 1);                      // in flight while this tile multiplies
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const half8 a_hi = s_tile[buf][(s * 2) * 64 + lane];
            const half8 a_lo = s_tile[buf][(s * 2 + 1) * 64 + lane];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, bq[s][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, bq[s][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, bq[s][0], acc, 0, 0, 0);
        }
        // register q of lane (n, h) = query n against train row (q & 3) + 8 (q >> 2) + 4 h
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 nr = *(const float4 *)(&s_norm[buf][8 * g + 4 * h]);
            const float nv[4] = {nr.x, nr.y, nr.z, nr.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = __builtin_fmaf(-2.0f, acc[4 * g + e], nv[e]);    // +inf beyond nt
                if (__any(v < top.v[KNN_KEEP - 1])) knn_insert(top, v, tt * 32 + 8 * g + 4 * h + e);
            }
        }
        if (tt + 1 < ntt) commit(buf ^ 1);                    // nobody reads that buffer now
        __syncthreads();
    }
    // the two halves of a wave hold different train rows for the same query: merge
    KnnTop other;
#pragma unroll
    for (int s = 0; s < KNN_KEEP; ++s) {
        other.v[s] = __shfl_xor(top.v[s], 32);
        other.i[s] = __shfl_xor(top.i[s], 32);
    }
#pragma unroll
    for (int s = 0; s < KNN_KEEP; ++s) knn_insert(top, other.v[s], other.i[s]);
    const int q = tq * 32 + n;
    if (live && h == 0 && q < nq) {
#pragma unroll
        for (int s = 0; s < KNN_KEEP; ++s) {
            cand_idx[KNN_KEEP * q + s] = top.i[s];
            cand_val[KNN_KEEP * q + s] = top.v[s];
        }
    }
}

// sum_k (q_k - t_k)^2 in float32, features dealt out over the lanes, a fixed reduction tree
__device__ __forceinline__ float knn_exact(const float *__restrict__ q, const float *__restrict__ t,
                                           int d, int lane) {
    float acc = 0.0f;
    for (int k = lane; k < d; k += 64) {
        const float df = q[k] - t[k];
        acc = __builtin_fmaf(df, df, acc);
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) acc += __shfl_xor(acc, off);
    return acc;
}

// One wave per query: exact distances of its three candidates, the proof that nothing else
// can be among the best two (or an exact rescan), the two nearest in order.
__global__ __launch_bounds__(256) void knn2_refine_kernel(
    const float *__restrict__ query, const float *__restrict__ train, int nq, int nt, int d,
    float scale, const float *__restrict__ norm_q, const unsigned *__restrict__ maxnorm,
    const int32_t *__restrict__ cand_idx, const float *__restrict__ cand_val,
    int32_t *__restrict__ idx, float *__restrict__ dist, int *__restrict__ n_rescan,
    int32_t *__restrict__ rescan_list, unsigned long long *__restrict__ rescan_keys) {
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (qi >= nq) return;
    const float *q = query + (size_t)qi * d;
    float best = __builtin_inff(), second = __builtin_inff();
    int bi = 0x7fffffff, si = 0x7fffffff;
    auto take = [&](float v, int j) {
        if (v < best || (v == best && j < bi)) {
            second = best;
            si = bi;
            best = v;
            bi = j;
        } else if (v < second || (v == second && j < si)) {
            second = v;
            si = j;
        }
    };
    for (int c = 0; c < KNN_KEEP; ++c) {
        const int j = cand_idx[KNN_KEEP * qi + c];
        if (j < nt) take(knn_exact(q, train + (size_t)j * d, d, lane), j);
    }
    // every row outside the list ranked at or above the last listed value, and a ranked value
    // is within eps of the true one: its squared distance is at least this
    const float s2 = scale * scale;
    const float eps = KNN_EPS * (norm_q[qi] + __uint_as_float(*maxnorm));
    const float floor_d2 = (cand_val[KNN_KEEP * qi + KNN_KEEP - 1] + norm_q[qi] - eps) / s2;
    if (nt > KNN_KEEP && !(second <= floor_d2)) {
        // not proven: the query goes on the list of exact rescans (knn2_rescan_kernel: one wave
        // scanning all rows of `train` alone took 90 ms for 100 000 rows and set the duration of
        // the whole search)
        if (lane == 0) {
            const int slot = atomicAdd(n_rescan, 1);
            rescan_list[slot] = qi;
            rescan_keys[2 * slot] = rescan_keys[2 * slot + 1] = ~0ull;
        }
        return;
    }
    if (lane == 0) {
        idx[2 * qi] = bi;
        idx[2 * qi + 1] = si;
        dist[2 * qi] = sqrtf(best);
        dist[2 * qi + 1] = sqrtf(second);
    }
}

// Exact rescans, all of them at once: block (x, y) scans rows x, x + gridDim.x, ... x 4 waves
// of `train` for the listed queries y, y + gridDim.y, ...; a wave keeps its two best as keys
// (distance bits << 32 | row: float order = integer order for non-negative floats, equal
// distances: lower row first) and merges them into the query's pair with two atomic minima -
// the smaller of (new, old first) stays first, the larger is offered to the second place, so
// every value but the final minimum is offered to the second place exactly once.
#define KNN_RESCAN_X 256
__global__ __launch_bounds__(256) void knn2_rescan_kernel(
    const float *__restrict__ query, const float *__restrict__ train, int nt, int d,
    const int *__restrict__ n_rescan, const int32_t *__restrict__ rescan_list,
    unsigned long long *__restrict__ rescan_keys) {
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int count = *n_rescan;
    for (int r = blockIdx.y; r < count; r += gridDim.y) {
        const float *q = query + (size_t)rescan_list[r] * d;
        unsigned long long b1 = ~0ull, b2 = ~0ull;
        for (int j = wave; j < nt; j += 4 * KNN_RESCAN_X) {
            const float v = knn_exact(q, train + (size_t)j * d, d, lane);
            const unsigned long long key = (unsigned long long)__float_as_uint(v) << 32 | (unsigned)j;
            if (key < b1) {
                b2 = b1;
                b1 = key;
            } else if (key < b2) {
                b2 = key;
            }
        }
        if (lane == 0) {
            unsigned long long *g = rescan_keys + 2 * r;
            for (const unsigned long long key : {b1, b2}) {
                if (key == ~0ull) continue;
                const unsigned long long prev = atomicMin(&g[0], key);
                atomicMin(&g[1], prev > key ? prev : key);
            }
        }
    }
}

__global__ __launch_bounds__(256) void knn2_finish_kernel(
    const int *__restrict__ n_rescan, const int32_t *__restrict__ rescan_list,
    const unsigned long long *__restrict__ rescan_keys, int32_t *__restrict__ idx,
    float *__restrict__ dist) {
    const int count = *n_rescan;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < count; r += gridDim.x * 256) {
        const int qi = rescan_list[r];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const unsigned long long key = rescan_keys[2 * r + k];
            idx[2 * qi + k] = (int32_t)(unsigned)key;
            dist[2 * qi + k] = sqrtf(__uint_as_float((unsigned)(key >> 32)));
        }
    }
}

static size_t knn_align(size_t v) { return (v + 255) & ~(size_t)255; }

struct KnnWork {
    size_t pq, pt, nq, ntn, cidx, cval, scal, rlist, rkeys, total;
};

static KnnWork knn_layout(int nq, int nt, int d) {
    const int ks = d <= 64 ? 4 : 8;
    const size_t tq = (size_t)((nq + 31) / 32), tt = (size_t)((nt + 31) / 32);
    KnnWork w;
    size_t off = 0;
    w.pq = off;
    off += knn_align(tq * ks * 2 * 64 * sizeof(half8));
    w.pt = off;
    off += knn_align(tt * ks * 2 * 64 * sizeof(half8));
    w.nq = off;
    off += knn_align(tq * 32 * sizeof(float));
    w.ntn = off;
    off += knn_align(tt * 32 * sizeof(float));
    w.cidx = off;
    off += knn_align((size_t)nq * KNN_KEEP * sizeof(int32_t));
    w.cval = off;
    off += knn_align((size_t)nq * KNN_KEEP * sizeof(float));
    w.scal = off;
    off += 256;
    w.rlist = off;
    off += knn_align((size_t)nq * sizeof(int32_t));
    w.rkeys = off;
    off += knn_align((size_t)nq * 2 * sizeof(unsigned long long));
    w.total = off;
    return w;
}

extern "C" size_t pano_knn2_work_bytes(int nq, int nt, int d) {
    if (nq < 0 || nt < 0 || d < 1) return 0;
    return knn_layout(nq, nt, d).total;
}

extern "C" int pano_knn2(pano_ctx *ctx, const float *query, int nq, const float *train, int nt, int d,
                         float scale, void *work, int32_t *idx, float *dist, int *rescans) {
    PANO_ENTER(ctx, "pano_knn2");
    PANO_REQUIRE(query && train && work && idx && dist, "pano_knn2: null pointer");
    PANO_REQUIRE(nq >= 0 && nt >= 2, "pano_knn2: %d queries against %d rows (at least 2)", nq, nt);
    PANO_REQUIRE(d >= 1 && d <= 16 * KNN_MAX_KS, "pano_knn2: %d features (at most %d)", d,
                 16 * KNN_MAX_KS);
    PANO_REQUIRE(scale > 0.0f, "pano_knn2: scale %g", (double)scale);
    if (nq == 0) return PANO_OK;
    const hipStream_t s = (hipStream_t)stream;
    const KnnWork w = knn_layout(nq, nt, d);
    const int ks = d <= 64 ? 4 : 8;
    unsigned char *base = (unsigned char *)work;
    half8 *pq = (half8 *)(base + w.pq), *pt = (half8 *)(base + w.pt);
    float *norm_q = (float *)(base + w.nq), *norm_t = (float *)(base + w.ntn);
    int32_t *cidx = (int32_t *)(base + w.cidx);
    float *cval = (float *)(base + w.cval);
    unsigned *maxnorm = (unsigned *)(base + w.scal);
    int *n_rescan = (int *)(base + w.scal) + 1;          // both inside the 256 zeroed bytes
    int32_t *rlist = (int32_t *)(base + w.rlist);
    unsigned long long *rkeys = (unsigned long long *)(base + w.rkeys);
    const int tq = (nq + 31) / 32, tt = (nt + 31) / 32;
    PANO_HIP(hipMemsetAsync(maxnorm, 0, 256, s));
    hipLaunchKernelGGL(knn_pack_kernel, dim3((unsigned)(((size_t)tq * ks * 64 + 255) / 256)), dim3(256),
                       0, s, query, nq, d, ks, scale, pq);
    hipLaunchKernelGGL(knn_pack_kernel, dim3((unsigned)(((size_t)tt * ks * 64 + 255) / 256)), dim3(256),
                       0, s, train, nt, d, ks, scale, pt);
    hipLaunchKernelGGL(knn_norm_kernel, dim3(ceil_div(tq * 32, 4)), dim3(256), 0, s, query, nq,
                       tq * 32, d, scale, norm_q, (unsigned *)nullptr);
    hipLaunchKernelGGL(knn_norm_kernel, dim3(ceil_div(tt * 32, 4)), dim3(256), 0, s, train, nt,
                       tt * 32, d, scale, norm_t, maxnorm);
    PANO_LAUNCH_CHECK("knn_pack_kernel");
    if (ks == 4)
        PANO_TIMED(PK_KNN2, s,
                   hipLaunchKernelGGL(knn2_kernel<4>, dim3(ceil_div(tq, 4)), dim3(256), 0, s, pq, pt,
                                      norm_t, nq, nt, cidx, cval));
    else
        PANO_TIMED(PK_KNN2, s,
                   hipLaunchKernelGGL(knn2_kernel<8>, dim3(ceil_div(tq, 4)), dim3(256), 0, s, pq, pt,
                                      norm_t, nq, nt, cidx, cval));
    PANO_LAUNCH_CHECK("knn2_kernel");
    hipLaunchKernelGGL(knn2_refine_kernel, dim3(ceil_div(nq, 4)), dim3(256), 0, s, query, train, nq, nt,
                       d, scale, norm_q, maxnorm, cidx, cval, idx, dist, n_rescan, rlist, rkeys);
    PANO_LAUNCH_CHECK("knn2_refine_kernel");
    // the unproven queries (their number is on the device: the grid walks the list)
    hipLaunchKernelGGL(knn2_rescan_kernel, dim3(KNN_RESCAN_X, nq < 64 ? nq : 64), dim3(256), 0, s,
                       query, train, nt, d, n_rescan, rlist, rkeys);
    PANO_LAUNCH_CHECK("knn2_rescan_kernel");
    hipLaunchKernelGGL(knn2_finish_kernel, dim3(ceil_div(nq < 65536 ? nq : 65536, 256)), dim3(256),
                       0, s, n_rescan, rlist, rkeys, idx, dist);
    PANO_LAUNCH_CHECK("knn2_finish_kernel");
    if (rescans)
        PANO_HIP(hipMemcpyAsync(rescans, n_rescan, sizeof(int), hipMemcpyDeviceToDevice, s));
    return PANO_OK;
}
