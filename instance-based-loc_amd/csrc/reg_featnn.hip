// reg_featnn.hip -- 33-d nearest-neighbour search of the registration features on the matrix cores.
//
// The search (utils/fpfh_register.py:110-119 -> Open3D's feature matching: for every point the nearest feature of the other
// cloud) is a dense distance matrix.  On the synthetic workload the detections keep ~800 noisy points after outlier
// removal, their nearest features are far (d2 ~ 1000) and the early-abandon VALU search of reg_register.hip runs nearly the
// whole 33-term chain for every candidate: 80 VALU instructions per (wave, candidate), 8.6 ms per step.  Here the matrix
// cores do the bulk and the exact arithmetic is kept for the few candidates that can matter:
//   d2(q, t) = |q|^2 + |t|^2 - 2 q.t,   q.t ~ qh.th + qh.tl + ql.th   (x = xh + xl + r, bf16 hi/lo split, |r| <= 2^-16 |x|)
// computed with v_mfma_f32_32x32x16_bf16 (three products x three 16-wide k steps over the 33 -> 48 padded terms).  Error of
// approx against the fp32 chain the VALU search / oracle evaluates, with N = |q|^2 + |t|^2 (so |q||t| <= N / 2, d2 <= 2 N):
//   split      bf16 keeps 8 significant bits: |x - xh| <= 2^-8 |x|, |x - xh - xl| <= 2^-16 |x|; the dropped terms ql.tl, rq.t,
//              q.rt are each <= 2^-16 |q||t|, times the factor 2 of the expansion:          6 * 2^-16 |q||t| <= 4.6e-5 N
//   accumulate 144 exact bf16 products summed in fp32 by the MFMAs, times 2:                 1.8e-5 |q||t|    <= 0.9e-5 N
//   norms      two 33-term fp32 fmaf chains:                                                                     0.2e-5 N
//   chain      the exact chain's own rounding (34 roundings of values <= d2):                                    0.4e-5 N
//   epilogue   the three fp32 operations that form the bound:                                                    0.1e-5 N
// total <= 6.2e-5 N; E = FM_C N with FM_C = 7e-5.
//   pass 1   up(q) = min_t (approx + E)                       -- an upper bound of the exact minimum
//   pass 2   every t with approx - E <= up(q) is a candidate  -- the exact minimiser (and every exact tie) is among them
//   exact    the fp32 fmaf chain of the VALU search for each candidate, folded with a 64-bit atomic min on
//            (distance bits, database index): the lexicographic minimum, i.e. exactly the result of the full scan.
#pragma clang fp contract(off)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "reg_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 fm_bf16x8;
typedef __attribute__((ext_vector_type(16))) float fm_f32x16;

#define FM_DT 32                 // database rows per chunk (one 32 x 32 MFMA tile per wave)
#define FM_ROWB 112              // LDS bytes per row: 48 bf16 + 16 pad (2-way instead of 4-way bank conflicts on ds_read_b128)
#define FM_C 7e-5f               // E = FM_C (|q|^2 + |t|^2)
#define FM_NQ 2                  // 32-query tiles per wave (they share every database fragment read)
#define FM_QUEUE 256             // per-wave candidate queue, flushed when fewer than 64 slots (one append step) are left

struct FmCand { int pair, qi, t, pad; };

struct FmTile {
    __attribute__((aligned(16))) unsigned char hi[FM_DT * FM_ROWB];
    __attribute__((aligned(16))) unsigned char lo[FM_DT * FM_ROWB];
    __attribute__((aligned(16))) float dn[FM_DT];        // (1 +- C) |t|^2 for pass 1 / 2   (+inf past the end of the database)
};

// grid (query tiles of 128 FM_NQ, pairs); PASS 1: up[] ; PASS 2: candidates.  Operands come pre-split from the instance features
// (fpfh_split: 48 hi | 48 lo bf16 per row, fpfh_norm).  Database chunks of 32 rows are copied to LDS (16-byte pieces, fetched
// into registers one chunk ahead); every wave holds its 32 queries as the B operand in registers.
template <int PASS, bool INDEXED>
__global__ __launch_bounds__(256) void ibl_feat_mfma_kernel(const FeatPair* __restrict__ pairs, FeatSources src, float* __restrict__ up,
                                                            FmCand* __restrict__ cand, int* __restrict__ n_cand, int cand_cap,
                                                            const int* __restrict__ need_pos, const int* __restrict__ need_list, int out0) {
    const FeatPair P = pairs[blockIdx.y];
    int n_q = P.qcnt, l0 = 0;
    if (INDEXED) { l0 = need_pos[P.out - out0]; n_q = need_pos[P.out - out0 + P.qcnt] - l0; }
    const int q0 = blockIdx.x * (128 * FM_NQ);
    if (q0 >= n_q) return;
    __shared__ FmTile tiles[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, kg = lane >> 5;
    const uint4* __restrict__ qs = reinterpret_cast<const uint4*>(src.split[P.qkind] + (int64_t)P.qsrc * 96);
    const uint4* __restrict__ ds = reinterpret_cast<const uint4*>(src.split[P.dkind] + (int64_t)P.dsrc * 96);     // 12 pieces per row
    const float* __restrict__ dnorm = src.norm[P.dkind] + P.dsrc;
    constexpr float SGN = PASS == 1 ? 1.0f + FM_C : 1.0f - FM_C;

    // this wave's FM_NQ x 32 queries as B operands (one set of database fragments serves them all): lane (n, kg) holds terms
    // 16 s + 8 kg + 0..7 of query n for k step s
    bool valid[FM_NQ];
    int qi[FM_NQ];                                     // local index inside the query instance
    fm_bf16x8 qh[FM_NQ][3], ql[FM_NQ][3];
    float qn_s[FM_NQ], mup[FM_NQ];
#pragma unroll
    for (int u = 0; u < FM_NQ; ++u) {
        const int qv = q0 + (wave * FM_NQ + u) * 32 + n;
        valid[u] = qv < n_q;
        const int qc = valid[u] ? qv : n_q - 1;
        qi[u] = INDEXED ? need_list[l0 + qc] - (P.out - out0) : qc;
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
            const uint4 h = qs[(int64_t)qi[u] * 12 + 2 * s3 + kg], l = qs[(int64_t)qi[u] * 12 + 6 + 2 * s3 + kg];
            __builtin_memcpy(&qh[u][s3], &h, 16);
            __builtin_memcpy(&ql[u][s3], &l, 16);
        }
        qn_s[u] = src.norm[P.qkind][P.qsrc + qi[u]] * SGN;
        mup[u] = PASS == 1 ? INFINITY : up[P.out + qi[u]];
    }

    // chunk staging: 32 rows x 12 pieces = 384 pieces of 16 bytes; thread t carries pieces t and t + 256 (t < 128)
    const int n_chunks = (P.dcnt + FM_DT - 1) / FM_DT;
    uint4 pre0 = make_uint4(0, 0, 0, 0), pre1 = make_uint4(0, 0, 0, 0);
    float pren = INFINITY;
    auto fetch = [&](int t0) {
        const int r0 = tid / 12, p0 = tid - r0 * 12;
        pre0 = t0 + r0 < P.dcnt ? ds[(int64_t)(t0 + r0) * 12 + p0] : make_uint4(0, 0, 0, 0);
        if (tid < 128) {
            const int e = tid + 256, r1 = e / 12, p1 = e - r1 * 12;
            pre1 = t0 + r1 < P.dcnt ? ds[(int64_t)(t0 + r1) * 12 + p1] : make_uint4(0, 0, 0, 0);
        }
        if (tid < FM_DT) pren = t0 + tid < P.dcnt ? dnorm[t0 + tid] * SGN : INFINITY;
    };
    auto stash = [&](FmTile& T) {
        const int r0 = tid / 12, p0 = tid - r0 * 12;
        *reinterpret_cast<uint4*>((p0 < 6 ? T.hi : T.lo) + r0 * FM_ROWB + 16 * (p0 < 6 ? p0 : p0 - 6)) = pre0;
        if (tid < 128) {
            const int e = tid + 256, r1 = e / 12, p1 = e - r1 * 12;
            *reinterpret_cast<uint4*>((p1 < 6 ? T.hi : T.lo) + r1 * FM_ROWB + 16 * (p1 < 6 ? p1 : p1 - 6)) = pre1;
        }
        if (tid < FM_DT) T.dn[tid] = pren;
    };
    __shared__ int2 queue[PASS == 2 ? 4 : 1][PASS == 2 ? FM_QUEUE : 1];
    int qcount = 0;                                  // wave-uniform
    auto flush = [&]() {
        if (qcount == 0) return;
        int base = 0;
        if (lane == 0) base = atomicAdd(n_cand, qcount);
        base = __shfl(base, 0, 64);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < qcount; i += 64)
            if (base + i < cand_cap) { const int2 e = queue[wave][i]; cand[base + i] = FmCand{(int)blockIdx.y, e.x, e.y, 0}; }
        __builtin_amdgcn_wave_barrier();
        qcount = 0;
    };
    fetch(0);
    stash(tiles[0]);
    __syncthreads();
    for (int c = 0; c < n_chunks; ++c) {
        FmTile& T = tiles[c & 1];
        const bool more = c + 1 < n_chunks;
        if (more) fetch((c + 1) * FM_DT);
        fm_f32x16 acc[FM_NQ];
#pragma unroll
        for (int u = 0; u < FM_NQ; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[u][i] = 0.0f;
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
            // A operand: lane (m = n, kg) holds terms 16 s + 8 kg + 0..7 of database row m
            const fm_bf16x8 ah = *reinterpret_cast<const fm_bf16x8*>(T.hi + n * FM_ROWB + 32 * s3 + 16 * kg);
            const fm_bf16x8 al = *reinterpret_cast<const fm_bf16x8*>(T.lo + n * FM_ROWB + 32 * s3 + 16 * kg);
#pragma unroll
            for (int u = 0; u < FM_NQ; ++u) {
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qh[u][s3], acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ql[u][s3], acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, qh[u][s3], acc[u], 0, 0, 0);
            }
        }
        // acc[u][i] = q . t for database row m = 8 (i / 4) + 4 kg + (i % 4) of the chunk and query n of tile u;
        // bound(i) = (1 +- C)(|q|^2 + |t|^2) - 2 q . t
        float dnv[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 du = *reinterpret_cast<const float4*>(&T.dn[8 * g + 4 * kg]);
            dnv[4 * g] = du.x; dnv[4 * g + 1] = du.y; dnv[4 * g + 2] = du.z; dnv[4 * g + 3] = du.w;
        }
#pragma unroll
        for (int u = 0; u < FM_NQ; ++u) {
            float lowest = INFINITY;
#pragma unroll
            for (int i = 0; i < 16; ++i) lowest = fminf(lowest, __builtin_fmaf(-2.0f, acc[u][i], dnv[i] + qn_s[u]));
            if (PASS == 1) {
                mup[u] = fminf(mup[u], lowest);
            } else if (__ballot(lowest <= mup[u] && valid[u]) != 0ull) {
                // some query of this tile has a candidate in this chunk: append to the wave's LDS queue (ballot compaction, no
                // atomics); the queue goes to the global list in batches -- one atomic per ~200 candidates instead of one each
                // (two million same-address atomics per step took longer than the whole search)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const bool hit = valid[u] && __builtin_fmaf(-2.0f, acc[u][i], dnv[i] + qn_s[u]) <= mup[u];
                    const unsigned long long m = __ballot(hit);
                    if (m) {
                        if (hit) queue[wave][qcount + __popcll(m & ((1ull << lane) - 1ull))] = make_int2(qi[u], c * FM_DT + 8 * (i >> 2) + 4 * kg + (i & 3));
                        qcount += __popcll(m);
                        if (qcount > FM_QUEUE - 64) flush();
                    }
                }
            }
        }
        if (more) stash(tiles[(c + 1) & 1]);        // the other buffer: its readers passed the barrier that ended chunk c - 1
        __syncthreads();
    }
    if (PASS == 1) {
#pragma unroll
        for (int u = 0; u < FM_NQ; ++u) {
            const float m2 = fminf(mup[u], __shfl_xor(mup[u], 32, 64));
            if (valid[u] && kg == 0) up[P.out + qi[u]] = m2;
        }
    } else {
        flush();
    }
}

// thread per candidate: the exact fp32 chain (the summation order of the VALU search / oracle: rows are stored in matching
// order, terms 0..32), folded into the lexicographic minimum of (distance, database index)
__global__ __launch_bounds__(256) void ibl_feat_exact_kernel(const FeatPair* __restrict__ pairs, FeatSources src, const FmCand* __restrict__ cand,
                                                             const int* __restrict__ n_cand, int cand_cap,
                                                             unsigned long long* __restrict__ best) {
    const int total = min(*n_cand, cand_cap);
    for (int c = blockIdx.x * 256 + threadIdx.x; c < total; c += gridDim.x * 256) {
        const FmCand K = cand[c];
        const FeatPair P = pairs[K.pair];
        const float* __restrict__ q = src.fpfh[P.qkind] + ((int64_t)P.qsrc + K.qi) * 33;
        const float* __restrict__ t = src.fpfh[P.dkind] + ((int64_t)P.dsrc + K.t) * 33;
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 33; ++k) { const float d = q[k] - t[k]; acc = __builtin_fmaf(d, d, acc); }
        const unsigned long long key = ((unsigned long long)__float_as_uint(acc) << 32) | (unsigned)K.t;
        atomicMin(&best[P.out + K.qi], key);
    }
}

__global__ __launch_bounds__(256) void ibl_feat_finish_kernel(const unsigned long long* __restrict__ best, int64_t i0, int64_t n,
                                                              int* __restrict__ pair_idx, float* __restrict__ pair_d2) {
    const int64_t i = i0 + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= i0 + n) return;
    const unsigned long long k = best[i];
    if (k == 0xFFFFFFFFFFFFFFFFull) { pair_idx[i] = 0; pair_d2[i] = INFINITY; }       // not searched (not needed)
    else { pair_idx[i] = (int)(unsigned)(k & 0xFFFFFFFFull); pair_d2[i] = __uint_as_float((unsigned)(k >> 32)); }
}

int ibl_feat_search_mfma(ibl_reg_ctx* ctx, const FeatPair* d_pairs, int n_pairs, int max_q, const FeatSources& src, int* pair_idx,
                         float* pair_d2, const int* need_pos, const int* need_list, int out0, int64_t out_count, bool* overflow,
                         hipStream_t s) {
    *overflow = false;
    if (n_pairs <= 0 || out_count <= 0) return IBL_OK;
    if (n_pairs > 32768) { *overflow = true; return IBL_OK; }      // candidates carry the pair id as blockIdx.y: one launch only
    ArenaMark mark(ctx);
    int cand_cap = (int)std::min<int64_t>(out_count * 8 + 65536, (int64_t)1 << 27);
    if (const char* e = getenv("IBL_FEAT_CAND_CAP")) cand_cap = std::max(1, atoi(e));      // tests: force the overflow fallback
    float* up; FmCand* cand; int* n_cand; unsigned long long* best;
    IBL_ARENA(up, float, out_count + 64);
    IBL_ARENA(cand, FmCand, cand_cap);
    IBL_ARENA(n_cand, int, 64);
    IBL_ARENA(best, unsigned long long, out_count + 64);
    float* up0 = up - out0;                      // kernels index the output space of all pairs; this region starts at out0
    unsigned long long* best0 = best - out0;
    IBL_HIP_CHECK(hipMemsetAsync(n_cand, 0, sizeof(int), s));
    IBL_HIP_CHECK(hipMemsetAsync(best, 0xFF, sizeof(unsigned long long) * (size_t)out_count, s));
    const bool indexed = need_pos != nullptr;
    for (int p0 = 0; p0 < n_pairs; p0 += 32768) {
        const unsigned np = (unsigned)std::min(32768, n_pairs - p0);
        const dim3 grid((max_q + 128 * FM_NQ - 1) / (128 * FM_NQ), np);
        if (indexed) {
            hipLaunchKernelGGL((ibl_feat_mfma_kernel<1, true>), grid, dim3(256), 0, s, d_pairs + p0, src, up0, cand, n_cand, cand_cap, need_pos, need_list, out0);
            IBL_LAUNCH_CHECK();
            hipLaunchKernelGGL((ibl_feat_mfma_kernel<2, true>), grid, dim3(256), 0, s, d_pairs + p0, src, up0, cand, n_cand, cand_cap, need_pos, need_list, out0);
        } else {
            hipLaunchKernelGGL((ibl_feat_mfma_kernel<1, false>), grid, dim3(256), 0, s, d_pairs + p0, src, up0, cand, n_cand, cand_cap, need_pos, need_list, out0);
            IBL_LAUNCH_CHECK();
            hipLaunchKernelGGL((ibl_feat_mfma_kernel<2, false>), grid, dim3(256), 0, s, d_pairs + p0, src, up0, cand, n_cand, cand_cap, need_pos, need_list, out0);
        }
        IBL_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(ibl_feat_exact_kernel, dim3(2048), dim3(256), 0, s, d_pairs, src, cand, n_cand, cand_cap, best0);
    IBL_LAUNCH_CHECK();
    hipLaunchKernelGGL(ibl_feat_finish_kernel, dim3((unsigned)((out_count + 255) / 256)), dim3(256), 0, s, best0, (int64_t)out0, out_count,
                       pair_idx, pair_d2);
    IBL_LAUNCH_CHECK();
    int h_cand = 0;
    IBL_HIP_CHECK(hipMemcpyAsync(&h_cand, n_cand, sizeof(int), hipMemcpyDeviceToHost, s));
    IBL_HIP_CHECK(hipStreamSynchronize(s));
    if (h_cand > cand_cap) *overflow = true;
    return IBL_OK;
}
