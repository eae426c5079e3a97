// reg_featnn.hip -- 33-d nearest-neighbour search of the registration features on the matrix cores.
//
// The search (utils/fpfh_register.py:110-119 -> Open3D's feature matching: for every point the nearest feature of the other
// cloud) is a dense distance matrix: 5 000 x 5 000 rows per instance pair, ~600 pairs and both directions per bench step.  The
// matrix cores do the bulk and the exact arithmetic is kept for the few candidates that can matter.  Every feature row is stored
// once more as 48 fp16 "search operands" (ibl_fpfh_half_kernel) of the CENTRED row x = row - FM_MU (a constant vector: distances are
// unchanged, the norms -- and with them the error band below -- shrink to ~0.35 of the raw rows'; reg_api.hip):
//   x'  = [ x_0 .. x_32 | 8  8 | nh  nl | cu | 0 ... ]      nh + nl = |x|^2 / 8 (fp16 hi + lo),  cu = C |x|^2 + A rounded up
// and a query row is turned in registers into
//   q'' = [ -2 q_0 .. -2 q_32 | sh  sl | 8  8 | +-1 | 0 ... ]   sh + sl = (1 +- C) |q|^2 / 8
// so that ONE MFMA chain (three v_mfma_f32_32x32x16_f16 over the 48 terms) yields the whole bound
//   q'' . t' = (1 +- C)|q|^2 + |t|^2 +- C|t|^2 - 2 qh.th  =  d2_approx +- E,      E = C (|q|^2 + |t|^2)
// and the epilogue is a minimum (pass 1) or a compare (pass 2) per distance -- no norm adds, no hi/lo split products (the bf16
// hi | lo operands of round 1 needed three products per distance: 3x the matrix work, 2x the operand bytes).
// Error of d2_approx against the fp32 chain the VALU search / oracle evaluates, with N = |q|^2 + |t|^2 (|q||t| <= N / 2):
//   operands   fp16 keeps 11 significant bits, |x - xh| <= 2^-11 |x| (feature values are 0 .. 200: no overflow, no subnormals that
//              matter): |q.t - qh.th| <= 2^-11 (2 + 2^-11) |q||t|, times the factor 2 of the expansion:       <= 9.78e-4 N
//   accumulate 38 exact fp16 products summed in fp32 by the MFMAs (largest partial sum ~N):                    <= 0.5e-5 N
//   norms      fp32 fmaf chains, stored as fp16 hi + lo pairs (2^-22 relative):                                <= 0.3e-5 N
//   chain      the exact chain's own rounding (34 roundings of values <= d2 <= 2 N):                           <= 0.4e-5 N
//   centring   x = fl(row - mu) carries 2^-24 |x| per component into d2_approx against the chain on the raw rows:  <= 0.02e-5 N
//   absolute   a centred component (or norm term) in fp16's subnormal range is rounded to 2^-25 absolute instead of 2^-11
//              relative: <= 2 * 33 * 2 * 200 * 2^-25 = 8e-4 for the products, 8 * 2 * 2^-25 for the norm terms:    <= A = 4e-3
// total <= 9.9e-4 N + A; E = FM_C N + A with FM_C = 1.0e-3 (the C|t|^2 + A slot is rounded UP to fp16, the query's factor is applied
// in fp32).
//   pass 1   up(q) = min_t (d2_approx + E)                    -- an upper bound of the exact minimum
//   pass 2   every t with d2_approx - E <= up(q) is a candidate -- the exact minimiser (and every exact tie) is among them
//   exact    the fp32 fmaf chain of the VALU search for each candidate, folded with a 64-bit atomic min on
//            (distance bits, database index): the lexicographic minimum, i.e. exactly the result of the full scan.
// The band 2 E is ~14x wider than with the split operands, so pass 2 passes ~4 instead of ~2 candidates per query to the exact
// kernel -- 0.5 ms more there against 6 ms less here.
#pragma clang fp contract(off)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "reg_common.h"

typedef __attribute__((ext_vector_type(8))) _Float16 fm_h16x8;
typedef __attribute__((ext_vector_type(16))) float fm_f32x16;

#ifndef FM_SUB
#define FM_SUB 1                 // 32-row MFMA tiles per chunk (one workgroup barrier per chunk); 2 / 4 measured 4 % / 14 % slower per registration call
#endif
#define FM_DT (32 * FM_SUB)      // database rows per chunk
#define FM_ROWB 112              // LDS bytes per row: 48 fp16 + 16 pad (2-way instead of 4-way bank conflicts on ds_read_b128)
#define FM_C 1.0e-3f             // E = FM_C (|q|^2 + |t|^2)
#define FM_NS 8.0f               // norms are stored divided by 8 (|x|^2 reaches 1.2e5, fp16 ends at 65 504) against a constant 8
#ifndef FM_NQ
#define FM_NQ 2                  // 32-query tiles per wave (they share every database fragment read)
#endif
#define FM_P1_STRIDE 2           // pass 1 visits one database chunk in FM_P1_STRIDE (ibl_feat_search_mfma)
#define FM_QUEUE 256             // per-wave candidate queue, flushed when fewer than 64 slots (one append step) are left

struct FmCand { int pair, qi, t, pad; };

struct FmTile {
    __attribute__((aligned(16))) unsigned char rows[FM_DT * FM_ROWB];      // search operands of 32 database rows
};

// fp32 value -> fp16 hi + lo of value / FM_NS
__device__ __forceinline__ void fm_split_norm(float v, _Float16* h, _Float16* l) {
    const float w = v * (1.0f / FM_NS);
    *h = (_Float16)w;
    *l = (_Float16)(w - (float)*h);
}

// grid (query tiles of 128 FM_NQ, pairs); PASS 1: up[] ; PASS 2: candidates.  Operands come from the instance features
// (fpfh_split: 48 fp16 per row, laid out as in the file header).  Database chunks of 32 rows are copied to LDS (16-byte pieces,
// fetched into registers one chunk ahead); every wave holds its 32 queries as the B operand in registers.
// CONV: some instance-feature set of the call has no resident operand rows (src.split[kind] == null: compact features, 168 instead of 264
// bytes per point): its pieces are built from the fp32 rows (fm_operand_piece: the same bits ibl_fpfh_half_kernel would have stored) --
// the queries' once per wave, the database's as a chunk is stashed (the raw floats wait in registers where the stored piece would).
template <int PASS, bool INDEXED, bool CONV>
// pass 2 waits on its hit path (PMC: 63 % of its wave time): five instead of four waves per SIMD (a 96-register bound; three dwords
// spill outside the chunk loop) 2.05 -> 1.91 ms; six (80 registers) spills into the loop, 2x slower
#ifndef FM_OCC2
#define FM_OCC2 5
#endif
__global__ __launch_bounds__(256, PASS == 2 ? FM_OCC2 : 2) void ibl_feat_mfma_kernel(const FeatPair* __restrict__ pairs, FeatSources src, float* __restrict__ up,
                                                            FmCand* __restrict__ cand, unsigned long long* __restrict__ n_cand, int cand_cap,
                                                            const int* __restrict__ need_pos, const int* __restrict__ need_list, int out0, int cstride) {
    const FeatPair P = pairs[blockIdx.y];
    int n_q = P.qcnt, l0 = 0;
    if (INDEXED) { l0 = need_pos[P.out - out0]; n_q = need_pos[P.out - out0 + P.qcnt] - l0; }
    const int q0 = blockIdx.x * (128 * FM_NQ);
    if (q0 >= n_q) return;
    __shared__ FmTile tiles[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, kg = lane >> 5;
    const uint4* __restrict__ qs = reinterpret_cast<const uint4*>(src.split[P.qkind] + (int64_t)P.qsrc * 48);      // 6 pieces per row
    const uint4* __restrict__ ds = reinterpret_cast<const uint4*>(src.split[P.dkind] + (int64_t)P.dsrc * 48);
    const bool q_has_split = !CONV || src.split[P.qkind] != nullptr, d_has_split = !CONV || src.split[P.dkind] != nullptr;
    const float* __restrict__ dfp = src.fpfh[P.dkind] + (int64_t)P.dsrc * 33;
    const float* __restrict__ dnorm = src.norm[P.dkind] + P.dsrc;
    constexpr float SGN = PASS == 1 ? 1.0f + FM_C : 1.0f - FM_C;

    // this wave's FM_NQ x 32 queries as B operands (one set of database fragments serves them all): lane (n, kg) holds terms
    // 16 s + 8 kg + 0..7 of query n for k step s, turned from the stored x' into q'' (file header)
    bool valid[FM_NQ];
    int qi[FM_NQ];                                     // local index inside the query instance
    fm_h16x8 qh[FM_NQ][3];
    float mup[FM_NQ];
#pragma unroll
    for (int u = 0; u < FM_NQ; ++u) {
        const int qv = q0 + (wave * FM_NQ + u) * 32 + n;
        valid[u] = qv < n_q;
        const int qc = valid[u] ? qv : n_q - 1;
        qi[u] = INDEXED ? need_list[l0 + qc] - (P.out - out0) : qc;
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
            if (CONV && !q_has_split) {
                const int pc = 2 * s3 + kg;
                const float* __restrict__ row = src.fpfh[P.qkind] + ((int64_t)P.qsrc + qi[u]) * 33;
                float x8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x8[e] = 8 * pc + e < 33 ? row[8 * pc + e] : 0.0f;
                const fm_piece_t r = fm_operand_piece(x8, src.norm[P.qkind][P.qsrc + qi[u]], pc);
                __builtin_memcpy(&qh[u][s3], &r, 16);
            } else {
                const uint4 h = qs[(int64_t)qi[u] * 6 + 2 * s3 + kg];
                __builtin_memcpy(&qh[u][s3], &h, 16);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) qh[u][s3][e] = (_Float16)-2.0f * qh[u][s3][e];      // exact (a power of two)
        }
        if (kg == 0) {                                  // terms 32 .. 39 live on the kg = 0 lanes of k step 2
            _Float16 sh, sl;
            fm_split_norm(src.norm[P.qkind][P.qsrc + qi[u]] * SGN, &sh, &sl);
            qh[u][2][1] = sh; qh[u][2][2] = sl;
            qh[u][2][3] = (_Float16)FM_NS; qh[u][2][4] = (_Float16)FM_NS;
            qh[u][2][5] = (_Float16)(PASS == 1 ? 1.0f : -1.0f);
            qh[u][2][6] = (_Float16)0.0f; qh[u][2][7] = (_Float16)0.0f;
        }
        mup[u] = PASS == 1 ? INFINITY : up[P.out + qi[u]];
    }

    // chunk staging: 32 rows x 6 pieces = 192 pieces of 16 bytes, one per thread (threads 192 .. 255 idle here).  Rows past the
    // end of the database get a squared norm beyond any real bound instead (terms 35 / 36; d2 <= 2 N <= 4.8e5 for FPFH rows, whose
    // histograms sum to 200), so that they are never the minimum of pass 1; pass 2 checks the row index
    const int n_chunks = (P.dcnt + FM_DT - 1) / FM_DT;
    constexpr int NPIECE = FM_DT * 6, PPT = (NPIECE + 255) / 256;          // 16-byte pieces of a chunk, per thread
    uint4 pre[PPT];
    [[maybe_unused]] uint4 pre2[CONV ? PPT : 1];     // CONV: a piece's eight raw floats (pre, pre2) until the chunk is stashed
    [[maybe_unused]] float prn[CONV ? PPT : 1];      //       and, for piece 4, the row's centred norm
    [[maybe_unused]] bool raw[CONV ? PPT : 1];       //       (false: a ready-made piece -- a stored one, or the filler of a row past the end)
    int pr[PPT], pp[PPT];
#pragma unroll
    for (int v = 0; v < PPT; ++v) { pre[v] = make_uint4(0, 0, 0, 0); pr[v] = (tid + 256 * v) / 6; pp[v] = (tid + 256 * v) - pr[v] * 6; }
    auto fetch = [&](int t0) {
#pragma unroll
        for (int v = 0; v < PPT; ++v)
            if (tid + 256 * v < NPIECE) {
                if (CONV) raw[v] = false;
                if (t0 + pr[v] < P.dcnt) {
                    if (CONV && !d_has_split) {
                        raw[v] = true;
                        const float* row = dfp + (int64_t)(t0 + pr[v]) * 33 + 8 * pp[v];
                        if (pp[v] < 4) {                                   // (a 132-byte row is 4-byte aligned)
                            const ibl_u4_a4 lo = *reinterpret_cast<const ibl_u4_a4*>(row), hi = *reinterpret_cast<const ibl_u4_a4*>(row + 4);
                            pre[v] = make_uint4(lo.x, lo.y, lo.z, lo.w);
                            pre2[v] = make_uint4(hi.x, hi.y, hi.z, hi.w);
                        }
                        else if (pp[v] == 4) { pre[v].x = __float_as_uint(row[0]); prn[v] = dnorm[t0 + pr[v]]; }
                    } else {
                        pre[v] = ds[(int64_t)(t0 + pr[v]) * 6 + pp[v]];
                    }
                } else {
                    // terms 35, 36 = 65 504: a bound of 1.05e6 > 2 N
                    pre[v] = pp[v] == 4 ? make_uint4(0, 0x7BFFu << 16, 0x7BFFu, 0) : make_uint4(0, 0, 0, 0);
                }
            }
    };
    auto stash = [&](FmTile& T) {
#pragma unroll
        for (int v = 0; v < PPT; ++v)
            if (tid + 256 * v < NPIECE) {
                uint4 out = pre[v];
                if (CONV && raw[v]) {
                    float x8[8] = {__uint_as_float(pre[v].x), __uint_as_float(pre[v].y), __uint_as_float(pre[v].z), __uint_as_float(pre[v].w),
                                   __uint_as_float(pre2[v].x), __uint_as_float(pre2[v].y), __uint_as_float(pre2[v].z), __uint_as_float(pre2[v].w)};
                    const fm_piece_t r = fm_operand_piece(x8, prn[v], pp[v]);
                    __builtin_memcpy(&out, &r, 16);
                }
                *reinterpret_cast<uint4*>(T.rows + pr[v] * FM_ROWB + 16 * pp[v]) = out;
            }
    };
    __shared__ int2 queue[PASS == 2 ? 4 : 1][PASS == 2 ? FM_QUEUE : 1];
    int qcount = 0;                                  // wave-uniform
    auto flush = [&]() {
        if (qcount == 0) return;
        // 64-bit count: a database with thousands of near-identical rows (the interior of a large planar face) passes that many rows
        // per query, 10^10 per instance pair at 100 000 points -- a 32-bit counter wrapped and the slots below went out of bounds
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(n_cand, (unsigned long long)qcount);
        base = __shfl(base, 0, 64);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < qcount; i += 64)
            if (base + (unsigned long long)i < (unsigned long long)cand_cap) { const int2 e = queue[wave][i]; cand[base + i] = FmCand{(int)blockIdx.y, e.x, e.y, 0}; }
        __builtin_amdgcn_wave_barrier();
        qcount = 0;
    };
    // pass 1 visits every cstride-th chunk only (file header: any non-empty subset of the database gives a valid upper bound)
    fetch(0);
    stash(tiles[0]);
    __syncthreads();
    for (int c = 0, it = 0; c < n_chunks; c += cstride, ++it) {
        FmTile& T = tiles[it & 1];
        const bool more = c + cstride < n_chunks;
        if (more) fetch((c + cstride) * FM_DT);
        // all fragment reads of the chunk first: the MFMAs of its first 32 rows run while the later reads return
        fm_h16x8 ah[FM_SUB][3];
#pragma unroll
        for (int h = 0; h < FM_SUB; ++h)
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3)
                ah[h][s3] = *reinterpret_cast<const fm_h16x8*>(T.rows + (32 * h + n) * FM_ROWB + 32 * s3 + 16 * kg);
#pragma unroll
        for (int h = 0; h < FM_SUB; ++h) {
        fm_f32x16 acc[FM_NQ];
#pragma unroll
        for (int u = 0; u < FM_NQ; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[u][i] = 0.0f;
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
            // A operand: lane (m = n, kg) holds terms 16 s + 8 kg + 0..7 of database row 32 h + m of the chunk
#pragma unroll
            for (int u = 0; u < FM_NQ; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[h][s3], qh[u][s3], acc[u], 0, 0, 0);
        }
        // acc[u][i] = the bound (d2_approx + E in pass 1, - E in pass 2) of database row m = 8 (i / 4) + 4 kg + (i % 4) of the chunk
        // and query n of tile u
#pragma unroll
        for (int u = 0; u < FM_NQ; ++u) {
            // minima of the four groups of four accumulators (rows 8 g + 4 kg + 0..3), then of the tile
            float gm[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) gm[g] = fminf(fminf(acc[u][4 * g], acc[u][4 * g + 1]), fminf(acc[u][4 * g + 2], acc[u][4 * g + 3]));
            const float lowest = fminf(fminf(gm[0], gm[1]), fminf(gm[2], gm[3]));
            if (PASS == 1) {
                mup[u] = fminf(mup[u], lowest);
            } else if (__builtin_amdgcn_ballot_w64(lowest <= mup[u] && valid[u]) != 0ull) {
                // some query of this tile has a candidate in this chunk: append to the wave's LDS queue (ballot compaction, no
                // atomics); the queue goes to the global list in batches -- one atomic per ~200 candidates instead of one each
                // (two million same-address atomics per step took longer than the whole search).  The groups without a hit are
                // skipped as a whole: a hit is rare (0.2 - 0.4 per tile and chunk), and sixteen ballots per tile that has one made
                // pass 2 take twice the time of pass 1.
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (__builtin_amdgcn_ballot_w64(gm[g] <= mup[u] && valid[u]) == 0ull) continue;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = 4 * g + e;
                        const int row = c * FM_DT + 32 * h + 8 * g + 4 * kg + e;
                        const bool hit = valid[u] && acc[u][i] <= mup[u] && row < P.dcnt;
                        const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
                        if (m) {
                            if (hit) queue[wave][qcount + __popcll(m & ((1ull << lane) - 1ull))] = make_int2(qi[u], row);
                            qcount += __popcll(m);
                            if (qcount > FM_QUEUE - 64) flush();
                        }
                    }
                }
            }
        }
        }
        if (more) stash(tiles[(it + 1) & 1]);       // the other buffer: its readers passed the barrier that ended the chunk before this one
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
                                                             const unsigned long long* __restrict__ n_cand, int cand_cap,
                                                             unsigned long long* __restrict__ best) {
    const int total = (int)min(*n_cand, (unsigned long long)cand_cap);
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
                                                              int* __restrict__ pair_idx, float* __restrict__ pair_d2,
                                                              const unsigned long long* __restrict__ n_cand, int cand_cap, int* __restrict__ status) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && *n_cand > (unsigned long long)cand_cap) atomicOr(status, IBL_ST_FEAT_OVERFLOW);
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
    float* up; FmCand* cand; unsigned long long* n_cand; unsigned long long* best;
    IBL_ARENA(up, float, out_count + 64);
    IBL_ARENA(cand, FmCand, cand_cap);
    IBL_ARENA(n_cand, unsigned long long, 32);
    IBL_ARENA(best, unsigned long long, out_count + 64);
    float* up0 = up - out0;                      // kernels index the output space of all pairs; this region starts at out0
    unsigned long long* best0 = best - out0;
    IBL_HIP_CHECK(hipMemsetAsync(n_cand, 0, sizeof(unsigned long long), s));
    IBL_HIP_CHECK(hipMemsetAsync(best, 0xFF, sizeof(unsigned long long) * (size_t)out_count, s));
    const bool indexed = need_pos != nullptr;
    // pass 1 on every FM_P1_STRIDE-th chunk of the database: 1 / stride of a pass for a bound that is the stride-th smallest distance or so
    static int p1s = -1;
    if (p1s < 0) { const char* e = getenv("IBL_FEAT_P1_STRIDE"); p1s = e ? std::max(1, atoi(e)) : FM_P1_STRIDE; }
    bool conv = false;                  // a feature set without resident operand rows takes part
    for (int k = 0; k < 3; ++k) conv = conv || (src.fpfh[k] && !src.split[k]);
    for (int p0 = 0; p0 < n_pairs; p0 += 32768) {
        const unsigned np = (unsigned)std::min(32768, n_pairs - p0);
        const dim3 grid((max_q + 128 * FM_NQ - 1) / (128 * FM_NQ), np);
        auto passes = [&](auto cv) {
            constexpr bool CV = decltype(cv)::value;
            if (indexed) {
                hipLaunchKernelGGL((ibl_feat_mfma_kernel<1, true, CV>), grid, dim3(256), 0, s, d_pairs + p0, src, up0, cand, n_cand, cand_cap, need_pos, need_list, out0, p1s);
                hipLaunchKernelGGL((ibl_feat_mfma_kernel<2, true, CV>), grid, dim3(256), 0, s, d_pairs + p0, src, up0, cand, n_cand, cand_cap, need_pos, need_list, out0, 1);
            } else {
                hipLaunchKernelGGL((ibl_feat_mfma_kernel<1, false, CV>), grid, dim3(256), 0, s, d_pairs + p0, src, up0, cand, n_cand, cand_cap, need_pos, need_list, out0, p1s);
                hipLaunchKernelGGL((ibl_feat_mfma_kernel<2, false, CV>), grid, dim3(256), 0, s, d_pairs + p0, src, up0, cand, n_cand, cand_cap, need_pos, need_list, out0, 1);
            }
        };
        if (conv) passes(std::true_type{});
        else passes(std::false_type{});
        IBL_LAUNCH_CHECK();
    }
    if (getenv("IBL_TIMING") && atoi(getenv("IBL_TIMING")) >= 2) {
        unsigned long long h = 0;
        const hipError_t e = hipMemcpy(&h, n_cand, sizeof(h), hipMemcpyDeviceToHost);
        fprintf(stderr, "[reg-dbg] mfma passes done: %s; candidates %llu of cap %d, out_count %lld, pairs %d, max_q %d\n", hipGetErrorString(e), h, cand_cap,
                (long long)out_count, n_pairs, max_q);
    }
    hipLaunchKernelGGL(ibl_feat_exact_kernel, dim3(2048), dim3(256), 0, s, d_pairs, src, cand, n_cand, cand_cap, best0);
    IBL_LAUNCH_CHECK();
    // a candidate list that overflowed sets IBL_ST_FEAT_OVERFLOW in the context's status word; the driver reads it with its results
    // and redoes the call with the VALU search (no read-back here)
    hipLaunchKernelGGL(ibl_feat_finish_kernel, dim3((unsigned)((out_count + 255) / 256)), dim3(256), 0, s, best0, (int64_t)out0, out_count,
                       pair_idx, pair_d2, n_cand, cand_cap, ctx->d_status);
    IBL_LAUNCH_CHECK();
    return IBL_OK;
}
