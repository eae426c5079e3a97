// reg_register.hip -- batched registration of (detected, memory) cloud pairs and scoring against the
// whole memory: job assembly (concatenate + centre), 33-d feature matching with mutual filter,
// RANSAC on correspondences, coloured / point-to-point ICP, hash-grid evaluation.
//
// Replaces, for a whole batch of (frame, assignment) jobs at once,
//   object_memory/object_memory.py:1023-1034   concatenate the chosen clouds, subtract each side's mean
//   utils/fpfh_register.py:100-143            register_point_clouds (Open3D RANSAC + coloured ICP, p2p fallback)
//   utils/fpfh_register.py:145-150            evaluate_transform against the concatenation of all memory clouds
// RANSAC follows the index-ordered, Philox-driven semantics of oracle/oracle_reg.c (Open3D's own loop is
// unseeded and OpenMP-racy): hypotheses are generated and scored in parallel rounds, then folded by a
// sequential scan that reproduces the "better result / confidence-based early exit" bookkeeping exactly.
#pragma clang fp contract(off)
#include <hip/hip_runtime.h>

#include <memory>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <array>
#include <cmath>
#include <map>
#include <vector>

#include "ibloc.h"
#include "reg_common.h"

// normals (radius 2 voxel, 30 nn) and FPFH (5 voxel, 100 nn) of every cloud of a batch, colour gradients (grad_radius, 30 nn) of
// the points [gq0, gq1) -- reg_api.hip
int ibl_features_on_batch(ibl_reg_ctx* ctx, const float4* P, const int* seg_off_dev, const int* seg_off_host, int n_seg, const float* bbox_host, double voxel_size,
                          double grad_radius, int gq0, int gq1, float4* normals, float* fpfh, unsigned short* fpfh_split, float* fpfh_norm,
                          float4* grad, hipStream_t s);

#ifndef ICP_CELL_DIV
#define ICP_CELL_DIV 2.0     // cells of the ICP neighbour grid per correspondence distance (finer cells: fewer candidates per walk)
#endif
// ICP iteration from which a source point is searched by 8 lanes (ibl_icp_nn_group_kernel).  Measured per-launch times of the T
// workload (us): thread per point 446 376 351 347 342 324 229 136 115 100 ... 85 (floor); eight lanes 411 186 106 80 65 ... 41 (floor)
#ifndef ICP_GROUP_FROM
#define ICP_GROUP_FROM 8
#endif
// (32 lanes per point from iteration 11 on measured 100 us per launch: the grid of 32x the blocks, nearly all of finished jobs, costs
// more to schedule than the shorter walk saves)
#ifndef ICP_LPQ
#define ICP_LPQ 8            // lanes per source point of ibl_icp_nn_group_kernel
#endif
#define ICP_ACT_Y 32      // block rows of the ICP kernels once they walk the active-job list (iterations >= ICP_GROUP_FROM >= 1)
static_assert(ICP_GROUP_FROM >= 1, "the first active-job list is written by the update of iteration ICP_GROUP_FROM - 1");
#define ICP_BPJ 8        // blocks per job in the ICP / evaluation reductions (each ends in a 29-value fp64 block reduction)
#define ICP_NACC 29      // 21 (JTJ upper) + 6 (JTr) + count + err2  |  p2p: 3 + 3 + 9 + count + err2

// ------------------------------------------------------------------------------------------------
// job assembly
// ------------------------------------------------------------------------------------------------
struct JobDesc {             // host-built, copied to the device
    int src_seg[3];          // segments of the detected pool (-1 = unused)
    int tgt_seg[3];          // segments of the memory pool
};

// grid (J, 2): mean of the concatenated clouds in double
__global__ __launch_bounds__(256) void ibl_job_mean_kernel(const JobDesc* __restrict__ jobs, const float4* __restrict__ det,
                                                           const int* __restrict__ det_off, const float4* __restrict__ mem,
                                                           const int* __restrict__ mem_off, int center, double* __restrict__ means /* [J][2][3] */) {
    const int j = blockIdx.x, side = blockIdx.y;
    const float4* pool = side == 0 ? det : mem;
    const int* off = side == 0 ? det_off : mem_off;
    double sx = 0, sy = 0, sz = 0;
    long long cnt = 0;
    for (int t = 0; t < 3; ++t) {
        const int sg = side == 0 ? jobs[j].src_seg[t] : jobs[j].tgt_seg[t];
        if (sg < 0) continue;
        const int b = off[sg], e = off[sg + 1];
        cnt += e - b;
        for (int i = b + threadIdx.x; i < e; i += 256) { const float4 p = pool[i]; sx += p.x; sy += p.y; sz += p.z; }
    }
    __shared__ double sh[3][256];
    sh[0][threadIdx.x] = sx; sh[1][threadIdx.x] = sy; sh[2][threadIdx.x] = sz;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (threadIdx.x < st) for (int a = 0; a < 3; ++a) sh[a][threadIdx.x] += sh[a][threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        for (int a = 0; a < 3; ++a) means[(j * 2 + side) * 3 + a] = (center && cnt > 0) ? sh[a][0] / (double)cnt : 0.0;
}

// writes the centred job clouds: segments [0, J) = sources, [J, 2J) = targets
__global__ __launch_bounds__(256) void ibl_job_gather_kernel(const JobDesc* __restrict__ jobs, int J, const float4* __restrict__ det,
                                                             const int* __restrict__ det_off, const float4* __restrict__ mem,
                                                             const int* __restrict__ mem_off, const int* __restrict__ job_off /* [2J+1] */,
                                                             const double* __restrict__ means, float4* __restrict__ out) {
    const int n = job_off[2 * J];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int sgi = seg_of(job_off, 2 * J, i);
    const int side = sgi >= J ? 1 : 0, j = side ? sgi - J : sgi;
    int local = i - job_off[sgi];
    const float4* pool = side == 0 ? det : mem;
    const int* off = side == 0 ? det_off : mem_off;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < 3; ++t) {
        const int sg = side == 0 ? jobs[j].src_seg[t] : jobs[j].tgt_seg[t];
        if (sg < 0) continue;
        const int len = off[sg + 1] - off[sg];
        if (local < len) { p = pool[off[sg] + local]; break; }
        local -= len;
    }
    const double* m = means + (j * 2 + side) * 3;
    out[i] = make_float4((float)((double)p.x - m[0]), (float)((double)p.y - m[1]), (float)((double)p.z - m[2]), p.w);
}

// ------------------------------------------------------------------------------------------------
// feature stage: recomputed groups (raw concatenations of the instances that influence each other) and the
// assembly of the per-job feature arrays from the instance caches / the recomputed groups
// ------------------------------------------------------------------------------------------------
struct GroupDesc { int pool; int seg[3]; };          // pool 0 = detected, 1 = memory; -1 = unused slot
struct FeatCopy { int dst, src, count, kind; };      // kind & 3: 0 detected cache, 1 memory cache, 2 recomputed groups
#define FEATCOPY_GRAD 4                              // also copy the colour gradients (target sides)

__global__ __launch_bounds__(256) void ibl_group_gather_kernel(const GroupDesc* __restrict__ groups, int G, const float4* __restrict__ det,
                                                               const int* __restrict__ det_off, const float4* __restrict__ mem,
                                                               const int* __restrict__ mem_off, const int* __restrict__ grp_off,
                                                               float4* __restrict__ out) {
    const int n = grp_off[G];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int g = seg_of(grp_off, G, i);
    int local = i - grp_off[g];
    const float4* pool = groups[g].pool ? mem : det;
    const int* off = groups[g].pool ? mem_off : det_off;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < 3; ++t) {
        const int sg = groups[g].seg[t];
        if (sg < 0) continue;
        const int len = off[sg + 1] - off[sg];
        if (local < len) { p = pool[off[sg] + local]; break; }
        local -= len;
    }
    out[i] = p;
}

// Exact form of "instance B is within the influence radius of instance A": is any point of A closer than R to a point of B?
// Bounding boxes alone call most neighbouring instances close (their boxes overlap in empty corners), which forces their
// features to be recomputed in every job that contains both.  grid (NEAR_SPLIT, pairs): every block keeps the points of B
// that lie within R of A's box in LDS and sweeps its share of A's points (those within R of B's box) over them.
#define NEAR_SPLIT 8
#define NEAR_CAP 5120
struct NearPair { int pool, a, b, pad; float boxa[6], boxb[6]; };

__device__ __forceinline__ float box_dist2(const float* bx, float x, float y, float z) {
    const float dx = fmaxf(fmaxf(bx[0] - x, x - bx[3]), 0.0f), dy = fmaxf(fmaxf(bx[1] - y, y - bx[4]), 0.0f),
                dz = fmaxf(fmaxf(bx[2] - z, z - bx[5]), 0.0f);
    return dx * dx + dy * dy + dz * dz;
}

__global__ __launch_bounds__(256) void ibl_near_pair_kernel(const NearPair* __restrict__ pairs, const float4* __restrict__ det,
                                                            const int* __restrict__ det_off, const float4* __restrict__ mem,
                                                            const int* __restrict__ mem_off, float R2, int* __restrict__ flags) {
    const NearPair P = pairs[blockIdx.y];
    const float4* pool = P.pool ? mem : det;
    const int* off = P.pool ? mem_off : det_off;
    const int ab = off[P.a], ae = off[P.a + 1], bb = off[P.b], be = off[P.b + 1];
    __shared__ float sx[NEAR_CAP], sy[NEAR_CAP], sz[NEAR_CAP];
    __shared__ int nb, found;
    if (threadIdx.x == 0) { nb = 0; found = 0; }
    __syncthreads();
    for (int i = bb + threadIdx.x; i < be; i += 256) {
        const float4 p = pool[i];
        if (box_dist2(P.boxa, p.x, p.y, p.z) < R2) {
            const int pos = atomicAdd(&nb, 1);
            if (pos < NEAR_CAP) { sx[pos] = p.x; sy[pos] = p.y; sz[pos] = p.z; }
        }
    }
    __syncthreads();
    const int n = nb;
    if (n > NEAR_CAP) { if (threadIdx.x == 0) atomicOr(&flags[blockIdx.y], 1); return; }     // too many to hold: call it close
    if (n == 0) return;
    for (int i0 = ab + blockIdx.x * 256; i0 < ae; i0 += NEAR_SPLIT * 256) {
        const int i = i0 + threadIdx.x;
        bool hit = false;
        if (i < ae) {
            const float4 p = pool[i];
            if (box_dist2(P.boxb, p.x, p.y, p.z) < R2)
                for (int j = 0; j < n; ++j)
                    if (dist2f(p.x, p.y, p.z, sx[j], sy[j], sz[j]) < R2) { hit = true; break; }
        }
        if (hit) found = 1;
        __syncthreads();
        if (found) break;
    }
    if (threadIdx.x == 0 && found) atomicOr(&flags[blockIdx.y], 1);
}

// grid (tiles, copies): contiguous block copies (an instance's features are contiguous at both ends)
__global__ __launch_bounds__(256) void ibl_feat_assemble_kernel(const FeatCopy* __restrict__ copies, FeatSources src, float4* __restrict__ normals,
                                                                float* __restrict__ fpfh, float4* __restrict__ grad) {
    const FeatCopy c = copies[blockIdx.y];
    const int k = c.kind & 3;
    if (fpfh) {
        const float* sf = src.fpfh[k] + (int64_t)c.src * 33;
        float* df = fpfh + (int64_t)c.dst * 33;
        const int nf = c.count * 33;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < nf; i += gridDim.x * 256) df[i] = sf[i];
    }
    const float4* sn = src.normals[k] + c.src;
    const float4* sg = src.grad[k] + c.src;
    const bool want_grad = (c.kind & FEATCOPY_GRAD) != 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < c.count; i += gridDim.x * 256) {
        normals[c.dst + i] = sn[i];
        if (want_grad) grad[c.dst + i] = sg[i];
    }
}

// ------------------------------------------------------------------------------------------------
// feature matching: 1-NN in 33-d (fp32 fmaf chain over the bins in matching order, first minimum wins), both directions
// grid (tiles, 2J): y < J: queries = source j, database = target j;  y >= J: the reverse
// ------------------------------------------------------------------------------------------------
// A job side is a concatenation of up to three instances, and the same (query instance, database instance) pair recurs in
// many jobs of a frame (every assignment that contains both), so the search runs once per distinct PAIR and a job's
// nearest neighbours are folded from its pairs in concatenation order with a strict '<' -- exactly the first minimum the
// scan over the concatenated database finds.  Features are read in place (instance caches / recomputed groups).
#define FT_TILE 32
struct SidePairs { int qcnt[3]; int dcnt[3]; int pair[3][3]; };              // per job side: slot sizes, pair ids (-1 = none)

// grid (query tiles, pairs)
// INDEXED: only the queries listed for the pair are searched (the target points that are some source point's nearest
// neighbour -- the mutual filter reads no other target's result): need_pos = exclusive scan of the need flags over the
// region of these pairs' outputs (which starts at out0), need_list = the flagged positions in order.
template <bool INDEXED>
__global__ __launch_bounds__(256) void ibl_feat_pair_nn_kernel(const FeatPair* __restrict__ pairs, FeatSources src, int* __restrict__ out_idx,
                                                               float* __restrict__ out_d2, const int* __restrict__ need_pos,
                                                               const int* __restrict__ need_list, int out0) {
    const FeatPair P = pairs[blockIdx.y];
    const int q0 = blockIdx.x * 256;
    int n_q = P.qcnt, l0 = 0;
    if (INDEXED) { l0 = need_pos[P.out - out0]; n_q = need_pos[P.out - out0 + P.qcnt] - l0; }
    if (q0 >= n_q) return;
    const float* __restrict__ qf = src.fpfh[P.qkind] + (int64_t)P.qsrc * 33;
    const float* __restrict__ df = src.fpfh[P.dkind] + (int64_t)P.dsrc * 33;
    const bool valid = q0 + (int)threadIdx.x < n_q;
    const int qv = valid ? q0 + (int)threadIdx.x : n_q - 1;
    const int qi = INDEXED ? need_list[l0 + qv] - (P.out - out0) : qv;          // local index of the query inside its instance
    float f[33];
    {
        const float* s = qf + (int64_t)qi * 33;
#pragma unroll
        for (int k = 0; k < 33; ++k) f[k] = s[k];
    }
    // Database rows go through LDS in tiles of FT_TILE rows (padded to 36 floats so that a row is read with broadcast
    // ds_read_b128), shared by the four waves of the block and double-buffered: the next tile's global loads are issued
    // before the current tile is searched and land in registers meanwhile.  (Reading the rows per wave through the scalar
    // cache instead re-fetched every row from L2 once per wave: 5.7 TB/s of L2 traffic, which bound the kernel.)
    __shared__ __attribute__((aligned(16))) float tiles[2][FT_TILE * 36];
    constexpr int PRE = (FT_TILE * 33 + 255) / 256;
    float pre[PRE];
    auto fetch = [&](int t0) {
        const int nt = min(FT_TILE, P.dcnt - t0) * 33;
        const float* __restrict__ g = df + (int64_t)t0 * 33;
#pragma unroll
        for (int i = 0; i < PRE; ++i) { const int e = threadIdx.x + 256 * i; pre[i] = e < nt ? g[e] : 0.0f; }
    };
    auto stash = [&](float* __restrict__ tile) {
#pragma unroll
        for (int i = 0; i < PRE; ++i) {
            const int e = threadIdx.x + 256 * i;
            if (e < FT_TILE * 33) { const int r = e / 33; tile[r * 36 + (e - r * 33)] = pre[i]; }
        }
    };
    float best = INFINITY;
    int bj = 0;
    fetch(0);
    stash(tiles[0]);
    __syncthreads();
    int cur = 0;
    for (int t0 = 0; t0 < P.dcnt; t0 += FT_TILE, cur ^= 1) {
        const bool more = t0 + FT_TILE < P.dcnt;
        if (more) fetch(t0 + FT_TILE);
        const float* __restrict__ tile = tiles[cur];
        const int nt = min(FT_TILE, P.dcnt - t0);
        for (int t = 0; t < nt; ++t) {
            const float* __restrict__ row = tile + t * 36;
            // Rows are stored in matching order (bins from the histogram centres outwards, FEAT_POS in reg_knn.hip), so the
            // chain is k = 0..32 over contiguous memory.  Its partial sums are non-decreasing: a target is abandoned as soon
            // as no lane of the wave can still beat its running minimum (checked after 4, 8, 12, 16 and 24 terms); the
            // surviving distances are the complete chains, bit-identical to the unpruned form.
            float acc = 0.0f;
#define FT_STAGE(k0, k1)                                                                                       \
            _Pragma("unroll") for (int k = k0; k < k1; ++k) { const float d = f[k] - row[k]; acc = __builtin_fmaf(d, d, acc); }
            FT_STAGE(0, 4)
            if (__ballot(acc < best) == 0ull) continue;
            FT_STAGE(4, 8)
            if (__ballot(acc < best) == 0ull) continue;
            FT_STAGE(8, 12)
            if (__ballot(acc < best) == 0ull) continue;
            FT_STAGE(12, 16)
            if (__ballot(acc < best) == 0ull) continue;
            FT_STAGE(16, 24)
            if (__ballot(acc < best) == 0ull) continue;
            FT_STAGE(24, 33)
#undef FT_STAGE
            if (acc < best) { best = acc; bj = t0 + t; }
        }
        if (more) stash(tiles[cur ^ 1]);
        __syncthreads();
    }
    if (valid) { out_idx[P.out + qi] = bj; out_d2[P.out + qi] = best; }
}

// thread per point of every job side: fold the pair results of its instance over the database instances in order
__global__ __launch_bounds__(256) void ibl_feat_fold_kernel(const SidePairs* __restrict__ sides, const FeatPair* __restrict__ pairs,
                                                            const int* __restrict__ pair_idx, const float* __restrict__ pair_d2,
                                                            const int* __restrict__ job_off, int J, int i0, int i1, int* __restrict__ nn) {
    const int i = i0 + blockIdx.x * 256 + threadIdx.x;
    if (i >= i1) return;
    const int sgi = seg_of(job_off, 2 * J, i);
    const SidePairs S = sides[sgi];
    int local = i - job_off[sgi], a = 0;
    while (a < 2 && local >= S.qcnt[a]) { local -= S.qcnt[a]; ++a; }
    float best = INFINITY;
    int bj = 0, dbase = 0;
    for (int b = 0; b < 3; ++b) {
        const int p = S.pair[a][b];
        if (p >= 0) {
            const int o = pairs[p].out + local;
            const float d = pair_d2[o];
            if (d < best) { best = d; bj = dbase + pair_idx[o]; }
        }
        dbase += S.dcnt[b];
    }
    nn[i] = bj;
}

// thread per source point: flag the target point it matched as needed in every (target instance -> source instance) pair of
// its job (the target's own nearest neighbour is folded over all source instances of the job)
__global__ __launch_bounds__(256) void ibl_feat_need_kernel(const SidePairs* __restrict__ sides, const FeatPair* __restrict__ pairs,
                                                            const int* __restrict__ job_off, int J, const int* __restrict__ nn, int out0,
                                                            int* __restrict__ need) {
    const int ns = job_off[J];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ns) return;
    const int j = seg_of(job_off, J, i);
    const SidePairs S = sides[j];                 // source side: dcnt = sizes of the target instances
    int local = nn[i], b = 0;
    if (local >= S.dcnt[0] + S.dcnt[1] + S.dcnt[2]) return;        // empty target side
    while (b < 2 && local >= S.dcnt[b]) { local -= S.dcnt[b]; ++b; }
    const SidePairs T = sides[J + j];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int p = T.pair[b][a];
        if (p >= 0) need[pairs[p].out - out0 + local] = 1;
    }
}

__global__ __launch_bounds__(256) void ibl_feat_need_list_kernel(const int* __restrict__ need, const int* __restrict__ pos, int n,
                                                                 int* __restrict__ list) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < n && need[e]) list[pos[e]] = e;
}

// one block per job: mutual filter + ordered compaction; falls back to all source->target matches when fewer than
// 3 * ransac_n survive (Open3D RegistrationRANSACBasedOnFeatureMatching)
__global__ __launch_bounds__(256) void ibl_mutual_kernel(const int* __restrict__ nn, const int* __restrict__ job_off, int J, int mutual,
                                                         int min_mutual, int2* __restrict__ corr /* capacity: source offsets */,
                                                         int* __restrict__ n_corr) {
    const int j = blockIdx.x;
    const int sb = job_off[j], se = job_off[j + 1], tb = job_off[J + j], te = job_off[J + j + 1];
    const int ns = se - sb, nt = te - tb;
    __shared__ int wave_cnt[4];
    __shared__ int base;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    if (ns == 0 || nt == 0) { if (threadIdx.x == 0) n_corr[j] = 0; return; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (mutual) {
        for (int i0 = 0; i0 < ns; i0 += 256) {
            const int i = i0 + threadIdx.x;
            bool keep = false;
            int tj = 0;
            if (i < ns) { tj = nn[sb + i]; keep = nn[tb + tj] == i; }
            const unsigned long long m = __ballot(keep);
            if (lane == 0) wave_cnt[wave] = __popcll(m);
            __syncthreads();
            int pre = base;
            for (int w = 0; w < wave; ++w) pre += wave_cnt[w];
            if (keep) corr[sb + pre + __popcll(m & ((1ull << lane) - 1ull))] = make_int2(i, tj);
            __syncthreads();
            if (threadIdx.x == 0) base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
            __syncthreads();
        }
        if (base >= min_mutual) { if (threadIdx.x == 0) n_corr[j] = base; return; }
    }
    for (int i = threadIdx.x; i < ns; i += 256) corr[sb + i] = make_int2(i, nn[sb + i]);
    if (threadIdx.x == 0) n_corr[j] = ns;
}

// ------------------------------------------------------------------------------------------------
// RANSAC
//
// Rounds of up to RANSAC_MAX_ROUND hypotheses per job.  Per round:
//   flag    thread per hypothesis: Philox draw, edge-length check, 3-point Kabsch, distance check -> 1 byte,
//           plus the count of survivors of every 256-hypothesis block
//   scan    exclusive sum of the block counts (hipcub) -> ordered offsets
//   scatter ordered list of surviving (job, hypothesis) ids
//   score   thread per survivor: its transform (once); then one wavefront per survivor: validate it on the correspondence set
//   fold    one wavefront per job: walk the survivors in hypothesis order and reproduce the sequential
//           "better result -> tighten est_k" bookkeeping of the reference loop exactly
// Correspondences are packed as (source xyz, target xyz) pairs so that a draw costs two 16-byte loads.
// ------------------------------------------------------------------------------------------------
#define RANSAC_MAX_ROUND 262144
#define RANSAC_FIRST_ROUND 4096
#ifndef RANSAC_WIDE_MIN_BLOCKS
#define RANSAC_WIDE_MIN_BLOCKS 1024       // fewer 16 k-hypothesis blocks than this in a round: 4 k blocks instead (run_round)
#endif
#define RANSAC_TAIL_JOBS 8                 // with at most this many jobs left ...
#define RANSAC_TAIL_ROUND (1 << 20)        // ... a round walks this many hypotheses per job

struct RansacState {
    double best_T[16];
    double best_fit, best_rmse;
    long long est_k, next_i, walked, validated, last_update;
    int best_inl;
    int done;
    unsigned job_id;          // the Philox counter word of this job: job_id_base + slot, or the caller's own id (ibl_register_batch_ids)
    int pad_;
};

__global__ void ibl_ransac_init_kernel(RansacState* __restrict__ st, const int* __restrict__ n_corr, int J, long long max_iter,
                                       double max_dist, int* __restrict__ active, unsigned job_id_base,
                                       const unsigned* __restrict__ job_ids) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= J) return;
    RansacState s;
    for (int i = 0; i < 16; ++i) s.best_T[i] = (i % 5) == 0 ? 1.0 : 0.0;
    s.best_fit = 0; s.best_rmse = 0; s.est_k = max_iter; s.next_i = 0; s.walked = 0; s.validated = 0; s.best_inl = 0; s.last_update = -1;
    s.done = (n_corr[j] < 3 || max_dist <= 0) ? 1 : 0;
    s.job_id = job_ids ? job_ids[j] : job_id_base + (unsigned)j;
    s.pad_ = 0;
    st[j] = s;
    active[j] = j;           // the first rounds run every job slot (a finished job's blocks leave at once); the host compacts later
}

// done flags for the host's read-back between groups of rounds
__global__ void ibl_ransac_done_kernel(const RansacState* __restrict__ st, int J, int* __restrict__ done) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < J) done[j] = st[j].done;
}

// packed correspondences: cp[2c] = source point, cp[2c + 1] = target point
__global__ __launch_bounds__(256) void ibl_pack_corr_kernel(const float4* __restrict__ pts, const int* __restrict__ job_off, int J,
                                                            const int2* __restrict__ corr, const int* __restrict__ n_corr,
                                                            float4* __restrict__ cp) {
    const int j = blockIdx.y;
    const int sb = job_off[j], tb = job_off[J + j], nc = n_corr[j];
    for (int c = blockIdx.x * 256 + threadIdx.x; c < nc; c += gridDim.x * 256) {
        const int2 cc = corr[sb + c];
        cp[2 * (int64_t)(sb + c)] = pts[sb + cc.x];
        cp[2 * (int64_t)(sb + c) + 1] = pts[tb + cc.y];
    }
}

// hypothesis i: Philox draw of three packed correspondences
__device__ __forceinline__ void ransac_draw(long long i, unsigned job_id, unsigned seed_lo, unsigned seed_hi, const float4* __restrict__ cp,
                                            int nc, double* s, double* d) {
    unsigned r[4];
    philox4x32((unsigned)i, job_id, (unsigned)((unsigned long long)i >> 32), 0u, seed_lo, seed_hi, r);
    for (int t = 0; t < 3; ++t) {
        const int pick = (int)(((unsigned long long)r[t] * (unsigned long long)nc) >> 32);
        const float4 ps = cp[2 * pick], pd = cp[2 * pick + 1];
        s[3 * t] = ps.x; s[3 * t + 1] = ps.y; s[3 * t + 2] = ps.z;
        d[3 * t] = pd.x; d[3 * t + 1] = pd.y; d[3 * t + 2] = pd.z;
    }
}

// CorrespondenceCheckerBasedOnEdgeLength
__device__ __forceinline__ bool ransac_edge_ok(const double* s, const double* d, double edge_sim) {
    for (int a = 0; a < 3; ++a)
        for (int b = a + 1; b < 3; ++b) {
            const double ds = sqrt((s[3 * a] - s[3 * b]) * (s[3 * a] - s[3 * b]) + (s[3 * a + 1] - s[3 * b + 1]) * (s[3 * a + 1] - s[3 * b + 1]) +
                                   (s[3 * a + 2] - s[3 * b + 2]) * (s[3 * a + 2] - s[3 * b + 2]));
            const double dt = sqrt((d[3 * a] - d[3 * b]) * (d[3 * a] - d[3 * b]) + (d[3 * a + 1] - d[3 * b + 1]) * (d[3 * a + 1] - d[3 * b + 1]) +
                                   (d[3 * a + 2] - d[3 * b + 2]) * (d[3 * a + 2] - d[3 * b + 2]));
            if (ds < dt * edge_sim || dt < ds * edge_sim) return false;
        }
    return true;
}

// The same check for the flag kernel's first pass (every hypothesis, ~99 % rejected): squared edge lengths in fp32 against
// edge_sim^2 with a 3e-6 guard band (the fp32 ratio is good to ~5e-7), no square roots.  Only a hypothesis with an edge
// ratio inside the band -- or a zero-length edge -- falls back to the exact double-precision form above, so the verdict
// is always the exact one.
__device__ __forceinline__ bool ransac_edge_ok_draw(long long i, unsigned job_id, unsigned seed_lo, unsigned seed_hi,
                                                    const float4* __restrict__ cp, int nc, double edge_sim, float e2_lo, float e2_hi) {
    unsigned r[4];
    philox4x32((unsigned)i, job_id, (unsigned)((unsigned long long)i >> 32), 0u, seed_lo, seed_hi, r);
    float4 ps[3], pd[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int pick = (int)(((unsigned long long)r[t] * (unsigned long long)nc) >> 32);
        ps[t] = cp[2 * pick]; pd[t] = cp[2 * pick + 1];
    }
    bool borderline = false;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = a + 1; b < 3; ++b) {
            const float sx = ps[a].x - ps[b].x, sy = ps[a].y - ps[b].y, sz = ps[a].z - ps[b].z;
            const float tx = pd[a].x - pd[b].x, ty = pd[a].y - pd[b].y, tz = pd[a].z - pd[b].z;
            const float ds2 = sx * sx + sy * sy + sz * sz, dt2 = tx * tx + ty * ty + tz * tz;
            if (ds2 < dt2 * e2_lo || dt2 < ds2 * e2_lo) return false;
            if (!(ds2 > dt2 * e2_hi && dt2 > ds2 * e2_hi)) borderline = true;
        }
    if (!borderline) return true;
    double s[9], d[9];
    for (int t = 0; t < 3; ++t) {
        s[3 * t] = ps[t].x; s[3 * t + 1] = ps[t].y; s[3 * t + 2] = ps[t].z;
        d[3 * t] = pd[t].x; d[3 * t + 1] = pd[t].y; d[3 * t + 2] = pd[t].z;
    }
    return ransac_edge_ok(s, d, edge_sim);
}

// 3-point Kabsch + CorrespondenceCheckerBasedOnDistance
__device__ inline bool ransac_fit_ok(const double* s, const double* d, double max_dist, double* T) {
    double sm[3] = {0, 0, 0}, dm[3] = {0, 0, 0};
    for (int t = 0; t < 3; ++t) for (int a = 0; a < 3; ++a) { sm[a] += s[3 * t + a]; dm[a] += d[3 * t + a]; }
    for (int a = 0; a < 3; ++a) { sm[a] /= 3; dm[a] /= 3; }
    double H[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int t = 0; t < 3; ++t)
        for (int rr = 0; rr < 3; ++rr) for (int cc = 0; cc < 3; ++cc) H[rr][cc] += (d[3 * t + rr] - dm[rr]) * (s[3 * t + cc] - sm[cc]);
    kabsch_from_moments(sm, dm, H, T);
    for (int t = 0; t < 3; ++t) {
        double p[3];
        xform_d(T, s[3 * t], s[3 * t + 1], s[3 * t + 2], p);
        const double dx = p[0] - d[3 * t], dy = p[1] - d[3 * t + 1], dz = p[2] - d[3 * t + 2];
        if (sqrt(dx * dx + dy * dy + dz * dz) > max_dist) return false;
    }
    return true;
}

__device__ inline bool ransac_hypothesis(long long i, unsigned job_id, unsigned seed_lo, unsigned seed_hi, const float4* __restrict__ cp,
                                         int nc, double max_dist, double edge_sim, double* T) {
    double s[9], d[9];
    ransac_draw(i, job_id, seed_lo, seed_hi, cp, nc, s, d);
    if (!ransac_edge_ok(s, d, edge_sim)) return false;
    return ransac_fit_ok(s, d, max_dist, T);
}

// grid (round / (1024 RANSAC_SUBS), active jobs): each block walks 1024 RANSAC_SUBS consecutive hypotheses.  The cheap part (draw + edge-length
// check, ~99 % rejected) runs on every lane; the survivors of the whole chunk are compacted through LDS so that the
// expensive part (fp64 Kabsch + distance check) runs once, on densely packed lanes.  `flags` is zeroed by the host before the launch.
constexpr int RANSAC_LDS_CORR = 1024;    // correspondences staged in LDS (32 KiB)
// Blocks of the well-filled rounds (run_round): 8 k hypotheses.  16 k amortise the dense Kabsch pass over more survivors, but their 32 KB
// survivor table + the 32 KB of staged correspondences leave two workgroups per CU; 8 k blocks (48 KB, registers bound to 168) run three:
// 653 -> 554 us for the 262 144-hypothesis round of a bench step.
#ifndef RANSAC_BIG_SUBS
#define RANSAC_BIG_SUBS 8
#endif
#ifndef RANSAC_BIG_OCC
#define RANSAC_BIG_OCC 3                 // its workgroups per CU the registers must allow
#endif
template <int RANSAC_SUBS>               // 1024-hypothesis passes per block (one Kabsch pass over all their survivors)
__global__ __launch_bounds__(256, RANSAC_SUBS == RANSAC_BIG_SUBS ? RANSAC_BIG_OCC : 1) void ibl_ransac_flag_kernel(const RansacState* __restrict__ st, const float4* __restrict__ cp,
                                                              const int* __restrict__ job_off, const int* __restrict__ n_corr,
                                                              long long max_iter, double max_dist, double edge_sim, unsigned seed_lo,
                                                              unsigned seed_hi, unsigned job_id_base, int round_size,
                                                              unsigned char* __restrict__ flags /* [J][round] */,
                                                              int* __restrict__ blk_cnt /* [J][round/256] */,
                                                              const int* __restrict__ active /* job ids still running */) {
    constexpr int RANSAC_CHUNK = 1024 * RANSAC_SUBS;
    const int a = blockIdx.y, j = active[a];        // per-round tables are indexed by the job's slot in the active list
    const int nblk = round_size / 256;
    const RansacState& S = st[j];
    const long long next_i = S.next_i, est_k = S.est_k;
    const bool job_on = !S.done;
    const float4* c = cp + 2 * (int64_t)job_off[j];
    const int nc = n_corr[j];
    const unsigned job_id = S.job_id;
    __shared__ unsigned short surv[RANSAC_CHUNK];     // slot within the chunk
    __shared__ int nsurv;
    __shared__ int cnt16[4 * RANSAC_SUBS];
    // The packed correspondences of the job (32 B each) are drawn 3 at a time by every hypothesis: random 16-byte gathers that sat on
    // L2 latency with two waves per SIMD to hide it.  Up to RANSAC_LDS_CORR of them are staged in LDS once per block (a block draws
    // from them 6 x 4096 ... 16384 times); larger jobs keep reading global memory.  Same values either way.
    __shared__ float4 sc[2 * RANSAC_LDS_CORR];
    if (!job_on) {            // uniform per block: nothing survives, the counts of this block's 256-hypothesis groups are zero
        if (threadIdx.x < 4 * RANSAC_SUBS) blk_cnt[a * nblk + blockIdx.x * (4 * RANSAC_SUBS) + threadIdx.x] = 0;
        return;
    }
    const bool in_lds = nc <= RANSAC_LDS_CORR;
    if (in_lds && job_on)
        for (int t = threadIdx.x; t < 2 * nc; t += 256) sc[t] = c[t];
    if (threadIdx.x < 4 * RANSAC_SUBS) cnt16[threadIdx.x] = 0;
    if (threadIdx.x == 0) nsurv = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const float e2 = (float)(edge_sim * edge_sim), e2_lo = e2 * (1.0f - 3e-6f), e2_hi = e2 * (1.0f + 3e-6f);
    for (int r = 0; r < 4 * RANSAC_SUBS; ++r) {
        const int slot = blockIdx.x * RANSAC_CHUNK + r * 256 + threadIdx.x;
        const long long i = next_i + slot;
        bool ok = false;
        if (job_on && i < est_k && i < max_iter)
            ok = in_lds ? ransac_edge_ok_draw(i, job_id, seed_lo, seed_hi, sc, nc, edge_sim, e2_lo, e2_hi)
                        : ransac_edge_ok_draw(i, job_id, seed_lo, seed_hi, c, nc, edge_sim, e2_lo, e2_hi);
        const unsigned long long m = __ballot(ok);
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(&nsurv, __popcll(m));
        base = __shfl(base, 0, 64);
        if (ok) surv[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)(r * 256 + threadIdx.x);
    }
    __syncthreads();
    // the survivors of the whole chunk (~1 %) go through the fp64 Kabsch together: a single, densely packed pass
    const int ns = nsurv;
    for (int t = threadIdx.x; t < ns; t += 256) {
        const int slot = blockIdx.x * RANSAC_CHUNK + (int)surv[t];
        double sp[9], dp[9], T[16];
        if (in_lds) ransac_draw(next_i + slot, job_id, seed_lo, seed_hi, sc, nc, sp, dp);
        else ransac_draw(next_i + slot, job_id, seed_lo, seed_hi, c, nc, sp, dp);
        if (ransac_fit_ok(sp, dp, max_dist, T)) {
            flags[(int64_t)a * round_size + slot] = 1;
            atomicAdd(&cnt16[(slot - blockIdx.x * RANSAC_CHUNK) >> 8], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < 4 * RANSAC_SUBS) blk_cnt[a * nblk + blockIdx.x * (4 * RANSAC_SUBS) + threadIdx.x] = cnt16[threadIdx.x];
}

__global__ __launch_bounds__(256) void ibl_ransac_scatter_kernel(const unsigned char* __restrict__ flags, const int* __restrict__ blk_off,
                                                                 int round_size, int* __restrict__ list /* slot ids, ordered */, int list_cap) {
    const int a = blockIdx.y;
    const int nblk = round_size / 256;
    const int slot = blockIdx.x * 256 + threadIdx.x;
    const bool ok = flags[(int64_t)a * round_size + slot] != 0;
    __shared__ int wc[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(ok);
    if (lane == 0) wc[wave] = __popcll(m);
    __syncthreads();
    if (ok) {
        int pre = blk_off[a * nblk + blockIdx.x];
        for (int w = 0; w < wave; ++w) pre += wc[w];
        const int pos = pre + __popcll(m & ((1ull << lane) - 1ull));
        if (pos < list_cap) list[pos] = slot;              // an overflowing round is reported by the transform kernel
    }
}

// thread per survivor e in [0, total): its transform, once (job = the active slot whose offset range contains e)
__global__ __launch_bounds__(256) void ibl_ransac_transform_kernel(const RansacState* __restrict__ st, const float4* __restrict__ cp,
                                                                   const int* __restrict__ job_off, const int* __restrict__ n_corr,
                                                                   const int* __restrict__ active, int n_active, double max_dist, double edge_sim,
                                                                   unsigned seed_lo, unsigned seed_hi, unsigned job_id_base, int round_size,
                                                                   const int* __restrict__ blk_off, const int* __restrict__ list,
                                                                   const int* __restrict__ total_ptr, int list_cap, int* __restrict__ status,
                                                                   int* __restrict__ e_job, double* __restrict__ e_T) {
    int total = *total_ptr;                    // the round's survivors: the last entry of the block-count scan
    if (total > list_cap) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(status, IBL_ST_RANSAC_OVERFLOW);
        total = list_cap;
    }
    const int nblk = round_size / 256;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    int lo = 0, hi = n_active;                // largest active slot a with blk_off[a * nblk] <= e
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (blk_off[mid * nblk] <= e) lo = mid; else hi = mid;
    }
    const int j = active[lo];
    const long long i = st[j].next_i + list[e];
    double T[16];
    ransac_hypothesis(i, st[j].job_id, seed_lo, seed_hi, cp + 2 * (int64_t)job_off[j], n_corr[j], max_dist, edge_sim, T);
    e_job[e] = j;
#pragma unroll
    for (int t = 0; t < 12; ++t) e_T[(int64_t)e * 12 + t] = T[t];
    }
}

// wave per survivor: inliers and squared error of its transform over the job's correspondences
__global__ __launch_bounds__(256) void ibl_ransac_score_kernel(const float4* __restrict__ cp, const int* __restrict__ job_off,
                                                               const int* __restrict__ n_corr, double max_dist,
                                                               const int* __restrict__ total_ptr, int list_cap,
                                                               const int* __restrict__ e_job, const double* __restrict__ e_T,
                                                               int* __restrict__ e_inl, double* __restrict__ e_err2) {
    const int total = min(*total_ptr, list_cap);
    const int lane = threadIdx.x & 63;
    for (int e = blockIdx.x * 4 + (threadIdx.x >> 6); e < total; e += gridDim.x * 4) {
    const int j = e_job[e];
    const float4* c = cp + 2 * (int64_t)job_off[j];
    const int nc = n_corr[j];
    double T[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) T[t] = e_T[(int64_t)e * 12 + t];
    int inl = 0;
    double err2 = 0;
    const double md2_hi = max_dist * max_dist * (1.0 + 1e-12);      // d2 >= this => sqrt(d2) >= max_dist for certain
    for (int k = lane; k < nc; k += 64) {
        const float4 ps = c[2 * k], q = c[2 * k + 1];
        double p[3];
        xform_d(T, ps.x, ps.y, ps.z, p);
        const double dx = p[0] - q.x, dy = p[1] - q.y, dz = p[2] - q.z;
        const double d2 = dx * dx + dy * dy + dz * dz;
        if (d2 < md2_hi) {          // most correspondences of most hypotheses are far outliers: no square root for them
            const double dd = sqrt(d2);
            if (dd < max_dist) { ++inl; err2 += dd * dd; }
        }
    }
    inl = wave_sum_i(inl);
    err2 = wave_sum_d(err2);
    if (lane == 0) { e_inl[e] = inl; e_err2[e] = err2; }
    }
}

// wave per job: fold the round's survivors in hypothesis order
__global__ __launch_bounds__(64) void ibl_ransac_fold_kernel(RansacState* __restrict__ st, int J, const int* __restrict__ n_corr,
                                                             long long max_iter, double confidence, int round_size,
                                                             const int* __restrict__ blk_off, const int* __restrict__ list,
                                                             const int* __restrict__ e_inl, const double* __restrict__ e_err2,
                                                             const double* __restrict__ e_T, const int* __restrict__ active) {
    const int a = blockIdx.x, j = active[a];
    const int lane = threadIdx.x;
    RansacState S = st[j];
    if (S.done) return;
    const int nblk = round_size / 256;
    const int b = blk_off[a * nblk], e = blk_off[(a + 1) * nblk];      // the scan has one entry past the last slot (= total)
    const int nc = n_corr[j];
    bool stop = false;
    long long n_before_stop = 0;          // survivors with index < the stopping index in this round
    for (int c0 = b; c0 < e && !stop; c0 += 64) {
        const int k = c0 + lane;
        const bool v = k < e;
        const int inl = v ? e_inl[k] : -1;
        const long long idx = v ? S.next_i + list[k] : 0x7FFFFFFFFFFFFFFFll;
        unsigned long long cand = __ballot(v && inl > 0 && inl >= S.best_inl);
        while (cand) {
            const int t = __ffsll((long long)cand) - 1;
            cand &= cand - 1;
            const long long ci = __shfl(idx, t, 64);
            if (ci >= S.est_k) { stop = true; break; }
            const int cinl = __shfl(inl, t, 64);
            if (cinl < S.best_inl) continue;                    // best_inl may have grown inside this chunk
            const double cerr2 = e_err2[c0 + t];
            const double fit = (double)cinl / (double)nc, rmse = sqrt(cerr2 / cinl);
            if (fit > S.best_fit || (fit == S.best_fit && rmse < S.best_rmse)) {
                S.best_fit = fit; S.best_rmse = rmse; S.best_inl = cinl; S.last_update = ci;
                if (lane < 12) S.best_T[lane] = e_T[(int64_t)(c0 + t) * 12 + lane];
                if (confidence > 0.0) {          // (<= 0: fixed budget, IBL_REG_FIXED_BUDGET)
                    const double ek = log(1.0 - confidence) / log(1.0 - pow(fit, 3.0));
                    if (ek < (double)S.est_k) S.est_k = (long long)ceil(ek);
                }
            }
        }
        // survivors of this chunk that the reference loop walks: it stands at max(last update + 1, est_k)
        long long lim = S.last_update + 1 > S.est_k ? S.last_update + 1 : S.est_k;
        if (lim > max_iter) lim = max_iter;
        n_before_stop += __popcll(__ballot(v && idx < lim));
        if (!stop) { const unsigned long long beyond = __ballot(v && idx >= lim); if (beyond) stop = true; }
    }
    const long long end = S.next_i + round_size;
    const long long lim = S.est_k < max_iter ? S.est_k : max_iter;
    S.validated += n_before_stop;
    S.next_i = end;
    if (stop || end >= lim) {
        S.done = 1;
        long long w = S.last_update + 1 > lim ? S.last_update + 1 : lim;   // the reference loop stands at max(i0 + 1, est_k)
        if (w > max_iter) w = max_iter;
        S.walked = w;
    } else {
        S.walked = end;
    }
    // best_T lanes 0..11 were written by the owning lanes: gather them to lane 0's copy
    double bt = lane < 12 ? S.best_T[lane] : 0.0;
    for (int t = 0; t < 12; ++t) { const double v = __shfl(bt, t, 64); if (lane == 0) S.best_T[t] = v; }
    if (lane == 0) st[j] = S;
}

// ------------------------------------------------------------------------------------------------
// ICP (coloured / point-to-point)
// ------------------------------------------------------------------------------------------------
struct IcpState {
    double T[16];
    double fitness, rmse;
    int iter, done, started;
};

__global__ void ibl_icp_init_kernel(IcpState* __restrict__ st, int J, const RansacState* __restrict__ rs, int from_identity) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= J) return;
    IcpState s;
    for (int i = 0; i < 16; ++i) s.T[i] = (from_identity || !rs) ? ((i % 5) == 0 ? 1.0 : 0.0) : rs[j].best_T[i];
    s.T[12] = s.T[13] = s.T[14] = 0.0; s.T[15] = 1.0;
    s.fitness = 0; s.rmse = 0; s.iter = 0; s.done = 0; s.started = 0;
    st[j] = s;
}

// nearest target point with d2 < r2 (fp32 distance on the float-rounded query); (d2, index) lexicographic minimum.
// The row of cells through the query's own cell is scanned first: it nearly always holds the neighbour (or one almost as
// close), after which the other rows are skipped unless their distance lower bound can still reach the best -- a row is
// only skipped when the bound is strictly larger, so equal distances are always compared by index and the result does
// not depend on the scan order.
// `pos` / `*d2out` carry the minimum so far in and out (-1 / r2 to start): a target side is searched piece by piece.  The minimum is
// carried as a POSITION in the cell-sorted arrays (the caller reads g.order[pos] once, at the end).
// Inner loop (round 3): eight candidates per step, their loads issued together (one thread's walk is a chain of dependent latencies);
// the comparisons are branch-free selects -- the branchy form (`if (d2 < bd) ... else if (d2 == bd) ...` per candidate) compiled to ~10
// exec-mask / branch instructions per candidate, more than the arithmetic.  An exact tie with the running best (equal fp32 distances of
// two different points: duplicate points, symmetric configurations) is only DETECTED there; the step is then redone from its saved
// state with the sequential rule (lowest original index wins), so the result is the sequential scan's, bit for bit.  Candidates past
// the end of a row are clamped to its last point: a repeat of a candidate changes neither the minimum nor a tie.
// (Round 3 also measured: pruning the row cell by cell -- own cell first, the others by their x gap -- 30 % slower: more dependent
// cell-table loads than points saved.)
__device__ __forceinline__ int nn_within(const BatchGrid& g, const SegGrid& sg, float qx, float qy, float qz, float radius, int pos,
                                         float* d2out) {
    int reach = (int)ceilf(radius * sg.inv);
    if (reach < 1) reach = 1;
    const int cx = (int)floorf((qx - sg.minx) * sg.inv), cy = (int)floorf((qy - sg.miny) * sg.inv), cz = (int)floorf((qz - sg.minz) * sg.inv);
    float bd = *d2out;
    const int x0 = max(cx - reach, 0), x1 = min(cx + reach, sg.nx - 1);
    const float csz = 1.0f / sg.inv, slack = 1e-4f * csz + 1e-6f;
    auto scan_row = [&](int z, int y) {
        const int row = sg.cell_base + (z * sg.ny + y) * sg.nx;
        const int b = g.cell_start[row + x0], e = g.cell_start[row + x1 + 1];
        for (int jj = b; jj < e; jj += 8) {
            int c[8];
            float4 p[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { c[u] = min(jj + u, e - 1); p[u] = g.sorted_pts[c[u]]; }
            const float bd0 = bd;
            const int pos0 = pos;
            unsigned long long tie = 0ull;          // (wave masks OR-ed on the scalar unit: a per-lane flag got packed bit by bit)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float d2 = dist2f(qx, qy, qz, p[u].x, p[u].y, p[u].z);
                const bool lt = d2 < bd;
                tie |= __builtin_amdgcn_ballot_w64((d2 == bd) & (c[u] != pos));          // (pos < 0: a d2 equal to r2 -- the redo rejects it)
                bd = lt ? d2 : bd;
                pos = lt ? c[u] : pos;
            }
            if (tie != 0ull) {          // rare: the sequential rule from the saved state
                bd = bd0;
                pos = pos0;
#pragma unroll
                for (int u = 0; u < 8; ++u) {          // (unrolled: a loop would index p[] dynamically and put it in scratch)
                    const float d2 = dist2f(qx, qy, qz, p[u].x, p[u].y, p[u].z);
                    if (d2 < bd) { bd = d2; pos = c[u]; }
                    else if (d2 == bd && pos >= 0 && c[u] != pos) { if (g.order[c[u]] < g.order[pos]) pos = c[u]; }
                }
            }
        }
    };
    if (x1 >= x0) {
        const bool centre = cz >= 0 && cz < sg.nz && cy >= 0 && cy < sg.ny;
        if (centre) scan_row(cz, cy);
        for (int z = max(cz - reach, 0); z <= min(cz + reach, sg.nz - 1); ++z) {
            // lower bound of the distance to any point of the row: skip what cannot reach the current best
            const float zlo = sg.minz + (float)z * csz;
            const float gz = fmaxf((qz < zlo ? zlo - qz : (qz > zlo + csz ? qz - zlo - csz : 0.0f)) - slack, 0.0f);
            for (int y = max(cy - reach, 0); y <= min(cy + reach, sg.ny - 1); ++y) {
                if (centre && z == cz && y == cy) continue;
                const float ylo = sg.miny + (float)y * csz;
                const float gy = fmaxf((qy < ylo ? ylo - qy : (qy > ylo + csz ? qy - ylo - csz : 0.0f)) - slack, 0.0f);
                if (gz * gz + gy * gy > bd) continue;
                scan_row(z, y);
            }
        }
    }
    *d2out = bd;
    return pos;
}

// The same search by a GROUP of LPQ lanes per query (tail iterations of the ICP, below): a lane takes four consecutive candidates of
// every 4 * LPQ, so a row of n candidates costs n / (4 LPQ) dependent load rounds instead of n / 8, and the lanes of a group share
// their best distance after every row for the pruning.  The group's result is the (d2, index) lexicographic minimum over its lanes:
// the same neighbour as nn_within, whatever the order.  A lane's minimum is a position here too.
template <int LPQ>
__device__ __forceinline__ int nn_within_group(const BatchGrid& g, const SegGrid& sg, float qx, float qy, float qz, float radius, int sub, int pos,
                                               float* d2out) {
    int reach = (int)ceilf(radius * sg.inv);
    if (reach < 1) reach = 1;
    const int cx = (int)floorf((qx - sg.minx) * sg.inv), cy = (int)floorf((qy - sg.miny) * sg.inv), cz = (int)floorf((qz - sg.minz) * sg.inv);
    float bd = *d2out;            // this lane's best
    float gbd = bd;               // the group's best distance (pruning bound)
#pragma unroll
    for (int off = 1; off < LPQ; off <<= 1) gbd = fminf(gbd, __shfl_xor(gbd, off, 64));
    const int x0 = max(cx - reach, 0), x1 = min(cx + reach, sg.nx - 1);
    const float csz = 1.0f / sg.inv, slack = 1e-4f * csz + 1e-6f;
    auto scan_row = [&](int z, int y) {
        const int row = sg.cell_base + (z * sg.ny + y) * sg.nx;
        const int b = g.cell_start[row + x0], e = g.cell_start[row + x1 + 1];
        for (int jj = b + 4 * sub; jj < e; jj += 4 * LPQ) {
            int c[4];
            float4 p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { c[u] = min(jj + u, e - 1); p[u] = g.sorted_pts[c[u]]; }
            const float bd0 = bd;
            const int pos0 = pos;
            unsigned long long tie = 0ull;          // (wave masks OR-ed on the scalar unit: a per-lane flag got packed bit by bit)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float d2 = dist2f(qx, qy, qz, p[u].x, p[u].y, p[u].z);
                const bool lt = d2 < bd;
                tie |= __builtin_amdgcn_ballot_w64((d2 == bd) & (c[u] != pos));          // (pos < 0: a d2 equal to r2 -- the redo rejects it)
                bd = lt ? d2 : bd;
                pos = lt ? c[u] : pos;
            }
            if (tie != 0ull) {
                bd = bd0;
                pos = pos0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {          // (unrolled: a loop would index p[] dynamically and put it in scratch)
                    const float d2 = dist2f(qx, qy, qz, p[u].x, p[u].y, p[u].z);
                    if (d2 < bd) { bd = d2; pos = c[u]; }
                    else if (d2 == bd && pos >= 0 && c[u] != pos) { if (g.order[c[u]] < g.order[pos]) pos = c[u]; }
                }
            }
        }
        gbd = fminf(gbd, bd);
#pragma unroll
        for (int off = 1; off < LPQ; off <<= 1) gbd = fminf(gbd, __shfl_xor(gbd, off, 64));
    };
    if (x1 >= x0) {
        const bool centre = cz >= 0 && cz < sg.nz && cy >= 0 && cy < sg.ny;
        if (centre) scan_row(cz, cy);
        for (int z = max(cz - reach, 0); z <= min(cz + reach, sg.nz - 1); ++z) {
            const float zlo = sg.minz + (float)z * csz;
            const float gz = fmaxf((qz < zlo ? zlo - qz : (qz > zlo + csz ? qz - zlo - csz : 0.0f)) - slack, 0.0f);
            for (int y = max(cy - reach, 0); y <= min(cy + reach, sg.ny - 1); ++y) {
                if (centre && z == cz && y == cy) continue;
                const float ylo = sg.miny + (float)y * csz;
                const float gy = fmaxf((qy < ylo ? ylo - qy : (qy > ylo + csz ? qy - ylo - csz : 0.0f)) - slack, 0.0f);
                if (gz * gz + gy * gy > gbd) continue;          // (strictly larger than the group's best: ties are still compared)
                scan_row(z, y);
            }
        }
    }
    *d2out = bd;
    return pos;
}

// Late ICP iterations: the jobs still running are the ones that do not converge (wrong assignments: sources with no target inside the
// correspondence distance scan their whole 5 x 5 x 5 neighbourhood, ~1 000 candidates in ~125 dependent rounds), and with few jobs
// left a launch is as long as one thread's walk (~90 us, 22 launches per batch).  Here LPQ lanes share a query.
template <int LPQ>
__global__ __launch_bounds__(256) void ibl_icp_nn_group_kernel(BatchGrid g, const float4* __restrict__ pts, const int* __restrict__ job_off, int J,
                                                               const int* __restrict__ piece_off, const IcpState* __restrict__ st, float radius,
                                                               float r2, int* __restrict__ nn_idx, float* __restrict__ nn_d2,
                                                               const int* __restrict__ act_list, const int* __restrict__ act_cnt) {
    // grid (chunks of the largest source side, ICP_ACT_Y): block row y walks the ACTIVE jobs y, y + ICP_ACT_Y, ... of the list the previous
    // iteration's update kernel wrote -- a (chunks, J) grid spent 40 us per launch on dispatching the ~90 000 blocks of finished jobs
    const int n_act = *act_cnt;
    const int sub = threadIdx.x % LPQ;
    for (int a = blockIdx.y; a < n_act; a += gridDim.y) {
        const int j = act_list[a];
        const IcpState& S = st[j];
        const int p = job_off[j] + blockIdx.x * (256 / LPQ) + threadIdx.x / LPQ;
        if (p >= job_off[j + 1]) continue;               // (whole groups leave together)
        const int i = g.order[p];
        const float4 s4 = pts[i];
        double T[12], vs[3];
        for (int t = 0; t < 12; ++t) T[t] = S.T[t];
        xform_d(T, s4.x, s4.y, s4.z, vs);
        float d2 = r2;
        int pos = -1;
#pragma unroll 1
        for (int t = 0; t < 3; ++t) {
            const int k = J + 3 * j + t;
            if (piece_off[k + 1] > piece_off[k])      // (a lane keeps its own best from piece to piece; the group's best prunes)
                pos = nn_within_group<LPQ>(g, g.seg[k], (float)vs[0], (float)vs[1], (float)vs[2], radius, sub, pos, &d2);
        }
        int best = pos >= 0 ? g.order[pos] : -1;
        // (d2, index) minimum over the group; best = -1 (with d2 = r2) marks a lane that found nothing
#pragma unroll
        for (int off = 1; off < LPQ; off <<= 1) {
            const float od = __shfl_xor(d2, off, 64);
            const int ob = __shfl_xor(best, off, 64);
            if (od < d2 || (od == d2 && ob >= 0 && (best < 0 || ob < best))) { d2 = od; best = ob; }
        }
        if (sub == 0) {
            nn_idx[i] = best;
            nn_d2[i] = d2;
        }
    }
}

// Thread per source point of every job: nearest target point under the job's current T.  Split from the accumulation so
// that this latency-bound neighbour walk runs with few registers (many waves per SIMD hide the dependent cell / point
// loads) while the fp64 normal equations run in their own kernel on coalesced inputs.
// The grid has one segment per source side (0 .. J) and one per target INSTANCE (J + 3 j + t, `piece_off`): a side made of instances far
// apart would otherwise get one coarse grid over their union (>= extent / 128 per cell, hundreds of points per cell: one such job
// quadrupled the ICP time of its batch).
__global__ __launch_bounds__(256) void ibl_icp_nn_kernel(BatchGrid g, const float4* __restrict__ pts, const int* __restrict__ job_off, int J,
                                                         const int* __restrict__ piece_off, const IcpState* __restrict__ st, float radius,
                                                         float r2, int* __restrict__ nn_idx, float* __restrict__ nn_d2) {
    // grid (chunks of the largest source side, J): the job is the block's y index -- a finished job's blocks leave on their first load,
    // and a thread does not find its job by a binary search over the offsets (eight dependent loads before the walk could start)
    const int j = blockIdx.y;
    const IcpState& S = st[j];
    if (S.done) return;
    const int p = job_off[j] + blockIdx.x * 256 + threadIdx.x;
    if (p >= job_off[j + 1]) return;
    // walk the sources in the cell order of their own grid (segment j of the batch grid = the job's source side, so its sorted positions
    // are [job_off[j], job_off[j + 1]) too): neighbouring lanes then query neighbouring cells of the target grid (a rigid transform
    // keeps them together) and share cache lines
    const int i = g.order[p];
    const float4 s4 = pts[i];
    double T[12], vs[3];
    for (int t = 0; t < 12; ++t) T[t] = S.T[t];
    xform_d(T, s4.x, s4.y, s4.z, vs);
    float d2 = r2;
    int pos = -1;
#pragma unroll 1
    for (int t = 0; t < 3; ++t) {
        const int k = J + 3 * j + t;
        if (piece_off[k + 1] > piece_off[k]) pos = nn_within(g, g.seg[k], (float)vs[0], (float)vs[1], (float)vs[2], radius, pos, &d2);
    }
    nn_idx[i] = pos >= 0 ? g.order[pos] : -1;
    nn_d2[i] = d2;
}

// grid (ICP_BPJ, J), or (ICP_BPJ, ICP_ACT_Y) over the active-job list (act_list != null): the normal-equation / Kabsch moments of the
// correspondences found by ibl_icp_nn_kernel
__global__ __launch_bounds__(256) void ibl_icp_step_kernel(const float4* __restrict__ pts, const float4* __restrict__ normals,
                                                           const float4* __restrict__ grad, const int* __restrict__ job_off, int J,
                                                           const IcpState* __restrict__ st, const int* __restrict__ nn_idx,
                                                           const float* __restrict__ nn_d2, int colored,
                                                           double sl_g, double sl_p, double* __restrict__ partial /* [J][BPJ][NACC] */,
                                                           const int* __restrict__ act_list, const int* __restrict__ act_cnt) {
    const int n_act = act_list ? *act_cnt : J;
    for (int a = blockIdx.y; a < n_act; a += gridDim.y) {
    const int j = act_list ? act_list[a] : a;
    const IcpState& S = st[j];
    if (S.done) continue;
    const int sb = job_off[j], se = job_off[j + 1];
    double T[12];
    for (int t = 0; t < 12; ++t) T[t] = S.T[t];
    double acc[ICP_NACC];
    for (int t = 0; t < ICP_NACC; ++t) acc[t] = 0.0;
    for (int i = sb + blockIdx.x * 256 + threadIdx.x; i < se; i += ICP_BPJ * 256) {
        const int tj = nn_idx[i];
        if (tj < 0) continue;
        const float d2 = nn_d2[i];
        const float4 s4 = pts[i];
        double vs[3];
        xform_d(T, s4.x, s4.y, s4.z, vs);
        acc[27] += 1.0;
        acc[28] += (double)d2;
        const float4 t4 = pts[tj];
        const double vt[3] = {t4.x, t4.y, t4.z};
        if (colored) {
            const float4 n4 = normals[tj], g4 = grad[tj];
            const double nt[3] = {n4.x, n4.y, n4.z};
            double Jr[6], r, c[3];
            const double dv[3] = {vs[0] - vt[0], vs[1] - vt[1], vs[2] - vt[2]};
            cross3d(vs, nt, c);
            Jr[0] = sl_g * c[0]; Jr[1] = sl_g * c[1]; Jr[2] = sl_g * c[2]; Jr[3] = sl_g * nt[0]; Jr[4] = sl_g * nt[1]; Jr[5] = sl_g * nt[2];
            r = sl_g * dot3d(dv, nt);
            int q = 0;
            for (int a = 0; a < 6; ++a) { for (int b = a; b < 6; ++b) acc[q++] += Jr[a] * Jr[b]; }
            for (int a = 0; a < 6; ++a) acc[21 + a] += Jr[a] * r;
            const double pr = dot3d(dv, nt);
            const double vp[3] = {vs[0] - pr * nt[0], vs[1] - pr * nt[1], vs[2] - pr * nt[2]};
            const double is = s4.w, itg = t4.w;
            const double dit[3] = {g4.x, g4.y, g4.z};
            const double dp[3] = {vp[0] - vt[0], vp[1] - vt[1], vp[2] - vt[2]};
            const double is0 = dot3d(dit, dp) + itg;
            const double dn = dot3d(dit, nt);
            const double ditM[3] = {-(dit[0] - dn * nt[0]), -(dit[1] - dn * nt[1]), -(dit[2] - dn * nt[2])};
            cross3d(vs, ditM, c);
            Jr[0] = sl_p * c[0]; Jr[1] = sl_p * c[1]; Jr[2] = sl_p * c[2]; Jr[3] = sl_p * ditM[0]; Jr[4] = sl_p * ditM[1]; Jr[5] = sl_p * ditM[2];
            r = sl_p * (is - is0);
            q = 0;
            for (int a = 0; a < 6; ++a) { for (int b = a; b < 6; ++b) acc[q++] += Jr[a] * Jr[b]; }
            for (int a = 0; a < 6; ++a) acc[21 + a] += Jr[a] * r;
        } else {
            for (int a = 0; a < 3; ++a) { acc[a] += vs[a]; acc[3 + a] += vt[a]; }
            for (int rr = 0; rr < 3; ++rr) for (int cc = 0; cc < 3; ++cc) acc[6 + 3 * rr + cc] += vt[rr] * vs[cc];
        }
    }
    __shared__ double sh[4][ICP_NACC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = 0; t < ICP_NACC; ++t) {
        const double v = wave_sum_d(acc[t]);
        if (lane == 0) sh[wave][t] = v;
    }
    __syncthreads();
    if (threadIdx.x < ICP_NACC)
        partial[((int64_t)j * ICP_BPJ + blockIdx.x) * ICP_NACC + threadIdx.x] =
            ((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + sh[2][threadIdx.x]) + sh[3][threadIdx.x];
    __syncthreads();          // (sh is reused by the next job of this block row)
    }
}

__device__ inline bool solve6_d(double A[6][6], double* b, double* x) {
    double M[6][7];
    for (int i = 0; i < 6; ++i) { for (int j = 0; j < 6; ++j) M[i][j] = A[i][j]; M[i][6] = b[i]; }
    for (int c = 0; c < 6; ++c) {
        int piv = c;
        double mx = fabs(M[c][c]);
        for (int r = c + 1; r < 6; ++r) if (fabs(M[r][c]) > mx) { mx = fabs(M[r][c]); piv = r; }
        if (mx == 0.0 || !isfinite(mx)) return false;
        if (piv != c) for (int jj = 0; jj <= 6; ++jj) { const double t = M[c][jj]; M[c][jj] = M[piv][jj]; M[piv][jj] = t; }
        for (int r = c + 1; r < 6; ++r) {
            const double f = M[r][c] / M[c][c];
            for (int jj = c; jj <= 6; ++jj) M[r][jj] -= f * M[c][jj];
        }
    }
    for (int i = 5; i >= 0; --i) {
        double s = M[i][6];
        for (int jj = i + 1; jj < 6; ++jj) s -= M[i][jj] * x[jj];
        x[i] = s / M[i][i];
    }
    return true;
}

// wave per job: finish the reduction, convergence test, Gauss-Newton / Kabsch update
// act_list / act_cnt: the active jobs of this iteration (null: all J, one block each); next_list / next_cnt (may be null): the jobs
// still running after this update are appended for the next iteration (in any order: jobs are independent)
__device__ __forceinline__ void icp_update_job(IcpState* __restrict__ st, int j, const int* __restrict__ job_off, const double* __restrict__ partial,
                                               int colored, int max_iter, double rel_fitness, double rel_rmse, int* __restrict__ next_list,
                                               int* __restrict__ next_cnt);
__global__ __launch_bounds__(64) void ibl_icp_update_kernel(IcpState* __restrict__ st, int J, const int* __restrict__ job_off, const double* __restrict__ partial,
                                      int colored, int max_iter, double rel_fitness, double rel_rmse, const int* __restrict__ act_list,
                                      const int* __restrict__ act_cnt, int* __restrict__ next_list, int* __restrict__ next_cnt) {
    const int n_act = act_list ? *act_cnt : J;
    for (int a = blockIdx.x; a < n_act; a += gridDim.x)
        icp_update_job(st, act_list ? act_list[a] : a, job_off, partial, colored, max_iter, rel_fitness, rel_rmse, next_list, next_cnt);
}
__device__ __forceinline__ void icp_update_job(IcpState* __restrict__ st, int j, const int* __restrict__ job_off, const double* __restrict__ partial,
                                               int colored, int max_iter, double rel_fitness, double rel_rmse, int* __restrict__ next_list,
                                               int* __restrict__ next_cnt) {
    // one wavefront per job: lane t folds moment t over the blocks (in block order), lane 0 solves
    if (st[j].done) return;
    double mine = 0.0;
    if (threadIdx.x < ICP_NACC)
        for (int b = 0; b < ICP_BPJ; ++b) mine += partial[((int64_t)j * ICP_BPJ + b) * ICP_NACC + threadIdx.x];
    double a[ICP_NACC];
    for (int t = 0; t < ICP_NACC; ++t) a[t] = __shfl(mine, t, 64);
    if (threadIdx.x != 0) return;
    IcpState S = st[j];
    const int ns = job_off[j + 1] - job_off[j];
    const double cnt = a[27], err2 = a[28];
    const double nf = ns > 0 ? cnt / (double)ns : 0.0, nr = cnt > 0 ? sqrt(err2 / cnt) : 0.0;
    if (S.started && fabs(S.fitness - nf) < rel_fitness && fabs(S.rmse - nr) < rel_rmse) {
        S.fitness = nf; S.rmse = nr; S.done = 1;
        st[j] = S;
        return;
    }
    S.fitness = nf; S.rmse = nr; S.started = 1;
    if (S.iter >= max_iter) { S.done = 1; st[j] = S; return; }
    double U[16];
    for (int i = 0; i < 16; ++i) U[i] = (i % 5) == 0 ? 1.0 : 0.0;
    if (cnt > 0) {
        if (colored) {
            double A[6][6], nb[6], x[6];
            int q = 0;
            for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) { A[r][c] = a[q]; A[c][r] = a[q]; ++q; }
            for (int r = 0; r < 6; ++r) nb[r] = -a[21 + r];
            if (solve6_d(A, nb, x)) {
                const double ca = cos(x[0]), sa = sin(x[0]), cb = cos(x[1]), sb = sin(x[1]), cg = cos(x[2]), sg = sin(x[2]);
                U[0] = cg * cb; U[1] = cg * sb * sa - sg * ca; U[2] = cg * sb * ca + sg * sa; U[3] = x[3];
                U[4] = sg * cb; U[5] = sg * sb * sa + cg * ca; U[6] = sg * sb * ca - cg * sa; U[7] = x[4];
                U[8] = -sb;     U[9] = cb * sa;                U[10] = cb * ca;               U[11] = x[5];
            }
        } else {
            double sm[3], dm[3], H[3][3];
            for (int t = 0; t < 3; ++t) { sm[t] = a[t] / cnt; dm[t] = a[3 + t] / cnt; }
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) H[r][c] = a[6 + 3 * r + c] - cnt * dm[r] * sm[c];
            kabsch_from_moments(sm, dm, H, U);
        }
    }
    double R[16];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) {
        double s = 0;
        for (int k = 0; k < 4; ++k) s += U[4 * r + k] * S.T[4 * k + c];
        R[4 * r + c] = s;
    }
    for (int i = 0; i < 16; ++i) S.T[i] = R[i];
    S.iter++;
    st[j] = S;
    if (next_list) next_list[atomicAdd(next_cnt, 1)] = j;        // still running
}

// ------------------------------------------------------------------------------------------------
// the fused driver
// ------------------------------------------------------------------------------------------------
static thread_local bool tl_force_valu = false;     // set while a call is redone after the matrix-core search overflowed its list
static thread_local bool tl_ransac_full_list = false;   // set while a call is redone after a RANSAC round overflowed the survivor list
static thread_local const uint32_t* tl_job_ids = nullptr;   // ibl_register_batch_ids: the caller's job ids for the duration of its call

__global__ void ibl_status_clear_kernel(int* __restrict__ status, int mask) { atomicAnd(status, ~mask); }
__global__ void ibl_status_set_kernel(int* __restrict__ status, int mask) { atomicOr(status, mask); }
extern "C" int ibl_register_batch(ibl_reg_ctx* ctx, const float* det_pts4, const int32_t* det_off_dev, const int32_t* det_off_host,
                                  int n_det_seg, const float* mem_pts4, const int32_t* mem_off_dev, const int32_t* mem_off_host,
                                  int n_mem_seg, const int32_t* job_src_seg, const int32_t* job_tgt_seg, int n_jobs, double voxel_size,
                                  double global_dist_factor, double local_dist_factor, uint64_t seed, uint32_t job_id_base,
                                  int64_t ransac_max_iter, int flags, double* T_out, double* rmse_out, double* fitness_out,
                                  double* means_out, double* T_ransac_out, int64_t* ransac_stats_out, void* stream) {
    return ibl_register_batch_cached(ctx, det_pts4, det_off_dev, det_off_host, n_det_seg, mem_pts4, mem_off_dev, mem_off_host, n_mem_seg,
                                     job_src_seg, job_tgt_seg, n_jobs, voxel_size, global_dist_factor, local_dist_factor, seed, job_id_base,
                                     ransac_max_iter, flags, nullptr, nullptr, T_out, rmse_out, fitness_out, means_out, T_ransac_out,
                                     ransac_stats_out, nullptr, stream);
}

// ibl_register_batch_cached with explicit RANSAC job ids (host array, one per job): a job keeps its id -- and therefore its result, bit
// for bit -- whichever rank and batch it is executed in (the sharded-cloud routing of routing.py).
extern "C" int ibl_register_batch_ids(ibl_reg_ctx* ctx, const float* det_pts4, const int32_t* det_off_dev, const int32_t* det_off_host,
                                      int n_det_seg, const float* mem_pts4, const int32_t* mem_off_dev, const int32_t* mem_off_host,
                                      int n_mem_seg, const int32_t* job_src_seg, const int32_t* job_tgt_seg, const uint32_t* job_ids, int n_jobs,
                                      double voxel_size, double global_dist_factor, double local_dist_factor, uint64_t seed,
                                      int64_t ransac_max_iter, int flags, const ibl_instance_features* det_features,
                                      const ibl_instance_features* mem_features, double* T_out, double* rmse_out, double* fitness_out,
                                      double* means_out, double* T_ransac_out, int64_t* ransac_stats_out, int64_t* reuse_stats_out, void* stream) {
    if (!job_ids) return ibl_set_error(IBL_ERR_ARG, "ibl_register_batch_ids: null job ids");
    tl_job_ids = job_ids;
    const int st = ibl_register_batch_cached(ctx, det_pts4, det_off_dev, det_off_host, n_det_seg, mem_pts4, mem_off_dev, mem_off_host, n_mem_seg,
                                             job_src_seg, job_tgt_seg, n_jobs, voxel_size, global_dist_factor, local_dist_factor, seed, 0,
                                             ransac_max_iter, flags, det_features, mem_features, T_out, rmse_out, fitness_out, means_out,
                                             T_ransac_out, ransac_stats_out, reuse_stats_out, stream);
    tl_job_ids = nullptr;
    return st;
}

// One pass of the registration call.  *redo (bits REDO_*) tells the caller that the results of this pass are unusable and which
// of the two overflowing lists to avoid in the next one; the pass has released its arena allocations by then (the redo used
// to recurse from inside the pass, with the first pass's allocations still stacked under the second's).
enum { REDO_RANSAC_FULL_LIST = 1, REDO_FEAT_VALU = 2 };
static int register_batch_cached_pass(ibl_reg_ctx* ctx, const float* det_pts4, const int32_t* det_off_dev, const int32_t* det_off_host,
                                      int n_det_seg, const float* mem_pts4, const int32_t* mem_off_dev, const int32_t* mem_off_host,
                                      int n_mem_seg, const int32_t* job_src_seg, const int32_t* job_tgt_seg, int n_jobs,
                                      double voxel_size, double global_dist_factor, double local_dist_factor, uint64_t seed,
                                      uint32_t job_id_base, int64_t ransac_max_iter, int flags,
                                      const ibl_instance_features* det_features, const ibl_instance_features* mem_features,
                                      double* T_out, double* rmse_out, double* fitness_out, double* means_out, double* T_ransac_out,
                                      int64_t* ransac_stats_out, int64_t* reuse_stats_out, void* stream, int* redo) {
    *redo = 0;
    if (!ctx || !det_pts4 || !mem_pts4 || !det_off_dev || !mem_off_dev || !det_off_host || !mem_off_host || !job_src_seg || !job_tgt_seg ||
        !T_out || !rmse_out || !fitness_out)
        return ibl_set_error(IBL_ERR_ARG, "ibl_register_batch: null pointer");
    if (n_jobs <= 0 || voxel_size <= 0) return ibl_set_error(IBL_ERR_ARG, "ibl_register_batch: bad sizes");
    const int J = n_jobs;
    const bool colored = (flags & IBL_REG_HAVE_COLORS) != 0;
    const bool center = (flags & IBL_REG_CENTER) != 0;
    const int flags_in = flags;        // (`flags` names a RANSAC scratch array further down)
    hipStream_t s = (hipStream_t)stream;
    ArenaMark mark(ctx);
    void *tok_match = nullptr, *tok_ransac = nullptr, *tok_icp = nullptr;       // stage brackets of the in-process timer (bench.py)
    hipLaunchKernelGGL(ibl_status_clear_kernel, dim3(1), dim3(1), 0, s, ctx->d_status, IBL_ST_FEAT_OVERFLOW | IBL_ST_RANSAC_OVERFLOW);
    IBL_LAUNCH_CHECK();
    const bool timing = getenv("IBL_TIMING") != nullptr;
    auto now = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    double t_prev = now();
    auto dbg = [&](const char* what) {          // IBL_TIMING=2: synchronise after every launch group of the search phase
        static const bool on = getenv("IBL_TIMING") && atoi(getenv("IBL_TIMING")) >= 2;
        if (!on) return;
        const hipError_t e = hipStreamSynchronize(s);
        fprintf(stderr, "[reg-dbg] %-36s %s\n", what, hipGetErrorString(e));
    };
    auto phase = [&](const char* what) {
        if (!timing) return;
        (void)hipStreamSynchronize(s);
        const double t = now();
        fprintf(stderr, "[reg] %-28s %7.3f ms\n", what, t - t_prev);
        t_prev = t;
    };

    // ---- host: job table + job cloud offsets ------------------------------------------------------
    std::vector<JobDesc> jobs(J);
    std::vector<int> job_off(2 * J + 1, 0);
    for (int j = 0; j < J; ++j) {
        int ns = 0, nt = 0;
        for (int t = 0; t < 3; ++t) {
            const int a = job_src_seg[3 * j + t], b = job_tgt_seg[3 * j + t];
            if (a >= n_det_seg || b >= n_mem_seg) return ibl_set_error(IBL_ERR_ARG, "ibl_register_batch: segment index out of range");
            jobs[j].src_seg[t] = a; jobs[j].tgt_seg[t] = b;
            if (a >= 0) ns += det_off_host[a + 1] - det_off_host[a];
            if (b >= 0) nt += mem_off_host[b + 1] - mem_off_host[b];
        }
        job_off[j + 1] = ns;          // sizes first, prefix below
        job_off[J + j + 1] = nt;
    }
    for (int i = 0; i < 2 * J; ++i) job_off[i + 1] += job_off[i];
    const int N = job_off[2 * J], Ns = job_off[J];

    JobDesc* d_jobs; int* d_job_off; double* d_means; float4* P; float4* normals;
    IBL_ARENA(d_jobs, JobDesc, J);
    IBL_ARENA(d_job_off, int, 2 * J + 1);
    IBL_ARENA(d_means, double, (int64_t)J * 6);
    IBL_ARENA(P, float4, N + 1);
    IBL_ARENA(normals, float4, N + 1);
    int st = ibl_stage_upload(ctx, d_jobs, jobs.data(), sizeof(JobDesc) * (int64_t)J, s);
    if (st) return st;
    st = ibl_stage_upload(ctx, d_job_off, job_off.data(), sizeof(int) * (int64_t)(2 * J + 1), s);
    if (st) return st;
    const float4* det = reinterpret_cast<const float4*>(det_pts4);
    const float4* mem = reinterpret_cast<const float4*>(mem_pts4);
    hipLaunchKernelGGL(ibl_job_mean_kernel, dim3(J, 2), dim3(256), 0, s, d_jobs, det, det_off_dev, mem, mem_off_dev, center ? 1 : 0, d_means);
    IBL_LAUNCH_CHECK();
    if (N > 0) {
        hipLaunchKernelGGL(ibl_job_gather_kernel, dim3((N + 255) / 256), dim3(256), 0, s, d_jobs, J, det, det_off_dev, mem, mem_off_dev,
                           d_job_off, d_means, P);
        IBL_LAUNCH_CHECK();
    }
    RansacState* rs = nullptr;
    IcpState* is;
    IBL_ARENA(is, IcpState, J);
    const double max_dist_icp = voxel_size * local_dist_factor;

    // grid C (cell = ICP correspondence distance) lives until the end: ICP neighbours (reach 1) and colour gradients
    // (radius 2 * max_dist, reach 2); grid A (cell = normal radius) only serves the normals
    phase("job assembly");
    BatchGrid gC;
    int* d_piece_off = nullptr;
    {
        // With the instance boxes on the host (instance features carry them) the table is sized from an upper bound of every job
        // side's extent -- centring moves a side, it does not stretch it beyond rounding -- and the build needs no read-back.
        // A batch with many spread-out job sides (assignments to instances far apart) would not fit the cell budget at the nominal cell
        // size: the cell grows until it does.  The neighbour walk derives its reach from the cell size, so only the work changes.
        const bool have = det_features && mem_features && det_features->bbox && mem_features->bbox;
        const int64_t budget = (int64_t)128 << 20;
        float cellC = (float)(max_dist_icp / ICP_CELL_DIV);
        auto bound_for = [&](float cell0) -> int64_t {
            int64_t bound = 0;
            for (int sgi = 0; sgi < 4 * J; ++sgi) {
                const int pl = sgi >= J ? 1 : 0, j = pl ? (sgi - J) / 3 : sgi, only = pl ? (sgi - J) % 3 : -1;
                const int* segs = pl ? jobs[j].tgt_seg : jobs[j].src_seg;
                const int* off = pl ? mem_off_host : det_off_host;
                const float* boxes = pl ? mem_features->bbox : det_features->bbox;
                float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
                bool any = false;
                for (int t = 0; t < 3; ++t) {
                    if (only >= 0 && t != only) continue;
                    if (segs[t] < 0 || off[segs[t] + 1] == off[segs[t]]) continue;
                    const float* b = boxes + 6 * (size_t)segs[t];
                    for (int c = 0; c < 3; ++c) { lo[c] = any ? std::min(lo[c], b[c]) : b[c]; hi[c] = any ? std::max(hi[c], b[3 + c]) : b[3 + c]; }
                    any = true;
                }
                double cells = 1;
                float cell = cell0;
                const float emax = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
                if (emax * 1.0001f / 128.0f > cell) cell = emax / 128.0f;
                for (int c = 0; c < 3; ++c) cells *= std::floor((double)(hi[c] - lo[c]) * 1.0001 / cell) + 2.0;
                bound += (int64_t)cells;
            }
            return bound;
        };
        // segments of this grid: the J source sides, then every target side instance by instance
        std::vector<int> piece_off(4 * (size_t)J + 1, 0);
        for (int j = 0; j <= J; ++j) piece_off[j] = job_off[j];
        for (int j = 0; j < J; ++j)
            for (int t = 0; t < 3; ++t) {
                const int b = jobs[j].tgt_seg[t];
                piece_off[J + 3 * j + t + 1] = piece_off[J + 3 * j + t] + (b >= 0 ? mem_off_host[b + 1] - mem_off_host[b] : 0);
            }
        IBL_ARENA(d_piece_off, int, 4 * (int64_t)J + 1);
        st = ibl_stage_upload(ctx, d_piece_off, piece_off.data(), sizeof(int) * (4 * (int64_t)J + 1), s);
        if (st) return st;
        int64_t bound = have ? bound_for(cellC) : 0;
        for (int tries = 0; have && bound >= budget && tries < 12; ++tries) { cellC *= 1.5f; bound = bound_for(cellC); }
        if (have && bound < budget) {
            st = ibl_build_batch_grid_bounded(ctx, P, d_piece_off, piece_off.data(), 4 * J, cellC, bound, &gC, s);
        } else {
            for (int tries = 0; tries < 8; ++tries) {
                st = ibl_build_batch_grid(ctx, P, d_piece_off, piece_off.data(), 4 * J, cellC, budget, &gC, s);
                if (st != IBL_ERR_OVERFLOW) break;
                cellC *= 2.0f;
            }
        }
    }
    if (st) return st;
    phase("grid C");
    float4* grad = nullptr;
    // host-side plans of the feature stage; they must outlive their H2D copies (synchronised in the RANSAC prologue)
    std::vector<GroupDesc> groups;
    std::vector<int> grp_off;
    std::vector<FeatCopy> copies;
    std::vector<FeatPair> pairs;
    std::vector<SidePairs> sides;
    std::vector<NearPair> near;
    std::vector<int> near_flag;
    if (colored) {
        IBL_ARENA(grad, float4, N + 1);
        IBL_ARENA(rs, RansacState, J);
        int2* corr; int* n_corr;
        IBL_ARENA(corr, int2, Ns + 1);
        IBL_ARENA(n_corr, int, J + 1);
        {
            // ---- normals + FPFH + colour gradients (instance cache / recomputed groups), then matching ----------
            ArenaMark m2(ctx);
            int* nn;
            IBL_ARENA(nn, int, N + 64);
            int* pair_idx = nullptr; float* pair_d2 = nullptr; FeatPair* d_pairs = nullptr; SidePairs* d_sides = nullptr;
            const double grad_radius = max_dist_icp * 2.0;
            const ibl_instance_features* feat[2] = {det_features, mem_features};
            for (int pl = 0; pl < 2; ++pl) {
                if (!feat[pl]) continue;
                if (!feat[pl]->normals4 || !feat[pl]->fpfh || !feat[pl]->fpfh_norm || !feat[pl]->bbox)       // (fpfh_split may be null: compact features)
                    return ibl_set_error(IBL_ERR_ARG, "ibl_register_batch_cached: instance features with null arrays");
                if (fabs(feat[pl]->voxel_size - voxel_size) > 1e-12 * voxel_size)
                    return ibl_set_error(IBL_ERR_ARG, "ibl_register_batch_cached: instance features were built for voxel_size %g, not %g",
                                         feat[pl]->voxel_size, voxel_size);
                if (pl == 1 && (!feat[pl]->grad4 || fabs(feat[pl]->grad_radius - grad_radius) > 1e-12 * grad_radius))
                    return ibl_set_error(IBL_ERR_ARG, "ibl_register_batch_cached: memory features need colour gradients of radius %g "
                                         "(2 * voxel_size * local_dist_factor)", grad_radius);
            }
            // influence radius of a foreign point on the features of an instance (see ibloc.h) + rounding margin
            const double rn = voxel_size * 2, rf = voxel_size * 5;
            const double R = std::max(2 * rf + rn, grad_radius + rn) * 1.001 + 1e-4;
            // ---- instances of one job side whose boxes are within R: decide exactly (point sets) on the device --------
            auto box_gap2 = [&](int pl, int sa, int sb) {
                const float* ba = feat[pl]->bbox + 6 * (size_t)sa;
                const float* bb = feat[pl]->bbox + 6 * (size_t)sb;
                double g2 = 0;
                for (int c = 0; c < 3; ++c) {
                    const double gap = std::max(0.0, std::max((double)ba[c] - (double)bb[3 + c], (double)bb[c] - (double)ba[3 + c]));
                    g2 += gap * gap;
                }
                return g2;
            };
            std::map<std::array<int, 3>, int> near_id;
            for (int sgi = 0; sgi < 2 * J; ++sgi) {
                const int pl = sgi >= J ? 1 : 0, j = pl ? sgi - J : sgi;
                if (!feat[pl]) continue;
                const int* segs = pl ? jobs[j].tgt_seg : jobs[j].src_seg;
                for (int a = 0; a < 3; ++a)
                    for (int b = a + 1; b < 3; ++b) {
                        if (segs[a] < 0 || segs[b] < 0) continue;
                        const std::array<int, 3> key = {pl, std::min(segs[a], segs[b]), std::max(segs[a], segs[b])};
                        if (near_id.count(key) || box_gap2(pl, key[1], key[2]) >= R * R) continue;
                        NearPair np;
                        np.pool = pl; np.a = key[1]; np.b = key[2]; np.pad = 0;
                        for (int c = 0; c < 6; ++c) { np.boxa[c] = feat[pl]->bbox[6 * (size_t)key[1] + c]; np.boxb[c] = feat[pl]->bbox[6 * (size_t)key[2] + c]; }
                        near_id[key] = (int)near.size();
                        near.push_back(np);
                    }
            }
            near_flag.assign(near.size(), 0);
            if (!near.empty()) {
                ArenaMark mn(ctx);
                NearPair* d_near; int* d_flags;
                IBL_ARENA(d_near, NearPair, (int64_t)near.size());
                IBL_ARENA(d_flags, int, (int64_t)near.size());
                st = ibl_stage_upload(ctx, d_near, near.data(), sizeof(NearPair) * (int64_t)near.size(), s);
                if (st) return st;
                IBL_HIP_CHECK(hipMemsetAsync(d_flags, 0, sizeof(int) * near.size(), s));
                const float Rf = nextafterf((float)R, INFINITY);
                for (size_t p0 = 0; p0 < near.size(); p0 += 32768) {
                    const unsigned np = (unsigned)std::min<size_t>(32768, near.size() - p0);
                    hipLaunchKernelGGL(ibl_near_pair_kernel, dim3(NEAR_SPLIT, np), dim3(256), 0, s, d_near + p0, det, det_off_dev, mem, mem_off_dev,
                                       Rf * Rf * 1.000001f, d_flags + p0);
                    IBL_LAUNCH_CHECK();
                }
                IBL_HIP_CHECK(hipMemcpyAsync(near_flag.data(), d_flags, sizeof(int) * near.size(), hipMemcpyDeviceToHost, s));
                IBL_HIP_CHECK(hipStreamSynchronize(s));
            }
            phase("near-pair test");
            // ---- plan: which instances of every job side keep their stand-alone features -------------------------
            std::map<std::array<int, 4>, int> gid[2];
            std::vector<std::array<int, 4>> gkeys[2];
            struct SlotPlan { int dst, count, pool, seg, grp, pos, side, kind, src; };
            std::vector<SlotPlan> slots;
            slots.reserve((size_t)6 * J);
            int64_t pts_cached = 0;
            for (int sgi = 0; sgi < 2 * J; ++sgi) {
                const int pl = sgi >= J ? 1 : 0, j = pl ? sgi - J : sgi;
                const int* segs = pl ? jobs[j].tgt_seg : jobs[j].src_seg;
                const int* off = pl ? mem_off_host : det_off_host;
                bool dirty[3] = {false, false, false};
                for (int a = 0; a < 3; ++a) {
                    if (segs[a] < 0) continue;
                    if (!feat[pl]) { dirty[a] = true; continue; }
                    for (int b = 0; b < 3; ++b) {
                        if (b == a || segs[b] < 0) continue;
                        const auto it = near_id.find({pl, std::min(segs[a], segs[b]), std::max(segs[a], segs[b])});
                        if (it != near_id.end() && near_flag[it->second]) dirty[a] = true;
                    }
                }
                std::array<int, 4> key = {pl, -1, -1, -1};
                int nd = 0, pos[3] = {0, 0, 0}, acc = 0;
                for (int a = 0; a < 3; ++a)
                    if (segs[a] >= 0 && dirty[a]) { key[1 + nd++] = segs[a]; pos[a] = acc; acc += off[segs[a] + 1] - off[segs[a]]; }
                int g = -1;
                if (nd > 0) {
                    auto it = gid[pl].find(key);
                    if (it == gid[pl].end()) { g = (int)gkeys[pl].size(); gid[pl][key] = g; gkeys[pl].push_back(key); }
                    else g = it->second;
                }
                int dst = job_off[sgi];
                for (int a = 0; a < 3; ++a) {
                    if (segs[a] < 0) continue;
                    const int cnt = off[segs[a] + 1] - off[segs[a]];
                    slots.push_back({dst, cnt, pl, segs[a], dirty[a] ? g : -1, pos[a], sgi, 0, 0});
                    if (!dirty[a]) pts_cached += cnt;
                    dst += cnt;
                }
            }
            // recomputed groups: detected-pool groups first, memory-pool groups last (their points get colour gradients)
            const int G0 = (int)gkeys[0].size(), G = G0 + (int)gkeys[1].size();
            groups.resize(G);
            grp_off.assign(G + 1, 0);
            for (int g = 0; g < G; ++g) {
                const std::array<int, 4>& k = g < G0 ? gkeys[0][g] : gkeys[1][g - G0];
                const int* off = k[0] ? mem_off_host : det_off_host;
                groups[g].pool = k[0];
                int cnt = 0;
                for (int t = 0; t < 3; ++t) { groups[g].seg[t] = k[1 + t]; if (k[1 + t] >= 0) cnt += off[k[1 + t] + 1] - off[k[1 + t]]; }
                grp_off[g + 1] = grp_off[g] + cnt;
            }
            const int Nd = grp_off[G];
            if (reuse_stats_out) { reuse_stats_out[0] = pts_cached; reuse_stats_out[1] = Nd; reuse_stats_out[2] = G; reuse_stats_out[3] = 2 * J; }
            copies.reserve(slots.size());
            for (SlotPlan& sp : slots) {
                if (sp.grp < 0) { sp.kind = sp.pool; sp.src = (sp.pool ? mem_off_host : det_off_host)[sp.seg]; }
                else { sp.kind = 2; sp.src = grp_off[(sp.pool ? G0 : 0) + sp.grp] + sp.pos; }
                if (sp.count <= 0) continue;
                FeatCopy c;
                c.dst = sp.dst; c.count = sp.count; c.kind = sp.kind; c.src = sp.src;
                if (sp.pool == 1) c.kind |= FEATCOPY_GRAD;
                copies.push_back(c);
            }
            // ---- feature matching plan: distinct (query instance, database instance) pairs -----------------------
            std::vector<int> side_first(2 * J + 1, 0);      // slots are stored side by side, in slot order
            for (const SlotPlan& sp : slots) ++side_first[sp.side + 1];
            for (int i = 0; i < 2 * J; ++i) side_first[i + 1] += side_first[i];
            std::map<std::array<int, 4>, int> pid;
            sides.assign(2 * J, SidePairs{});
            int64_t pair_pts = 0, pts0 = 0;     // outputs of the source-query pairs come first: [0, pts0)
            int n_pairs0 = 0;
            int max_q = 1;
            for (int sgi = 0; sgi < 2 * J; ++sgi) {
                if (sgi == J) { pts0 = pair_pts; n_pairs0 = (int)pairs.size(); }
                const int other = sgi < J ? sgi + J : sgi - J;
                SidePairs& S = sides[sgi];
                for (int a = 0; a < 3; ++a) { S.qcnt[a] = S.dcnt[a] = 0; for (int b = 0; b < 3; ++b) S.pair[a][b] = -1; }
                const int nq = side_first[sgi + 1] - side_first[sgi], nd = side_first[other + 1] - side_first[other];
                for (int a = 0; a < nq; ++a) S.qcnt[a] = slots[side_first[sgi] + a].count;
                for (int b = 0; b < nd; ++b) S.dcnt[b] = slots[side_first[other] + b].count;
                for (int a = 0; a < nq; ++a)
                    for (int b = 0; b < nd; ++b) {
                        const SlotPlan& q = slots[side_first[sgi] + a];
                        const SlotPlan& d = slots[side_first[other] + b];
                        if (q.count <= 0 || d.count <= 0) continue;
                        const std::array<int, 4> key = {q.kind, q.src, d.kind, d.src};
                        auto it = pid.find(key);
                        int id;
                        if (it == pid.end()) {
                            if (pair_pts + q.count > 0x7fffffff) return ibl_set_error(IBL_ERR_OVERFLOW, "feature matching: pair table exceeds 2^31 entries");
                            id = (int)pairs.size();
                            pid[key] = id;
                            pairs.push_back({q.kind, q.src, q.count, d.kind, d.src, d.count, (int)pair_pts, 0});
                            pair_pts += q.count;
                            max_q = std::max(max_q, q.count);
                        } else id = it->second;
                        S.pair[a][b] = id;
                    }
            }
            if (reuse_stats_out) {
                int64_t uses = 0;
                for (const SidePairs& S : sides)
                    for (int a = 0; a < 3; ++a)
                        for (int b = 0; b < 3; ++b) uses += S.pair[a][b] >= 0 ? 1 : 0;
                reuse_stats_out[4] = (int64_t)pairs.size();
                reuse_stats_out[5] = uses;
            }
            phase("host plan");
            IBL_ARENA(d_sides, SidePairs, 2 * J);
            IBL_ARENA(d_pairs, FeatPair, (int64_t)pairs.size() + 1);
            IBL_ARENA(pair_idx, int, pair_pts + 64);
            IBL_ARENA(pair_d2, float, pair_pts + 64);
            {
                ArenaMark md(ctx);
                FeatSources src{};
                for (int pl = 0; pl < 2; ++pl)
                    if (feat[pl]) {
                        src.normals[pl] = reinterpret_cast<const float4*>(feat[pl]->normals4);
                        src.fpfh[pl] = feat[pl]->fpfh;
                        src.split[pl] = feat[pl]->fpfh_split;
                        src.norm[pl] = feat[pl]->fpfh_norm;
                        src.grad[pl] = reinterpret_cast<const float4*>(feat[pl]->grad4);
                    }
                if (Nd > 0) {
                    GroupDesc* d_groups; int* d_grp_off; float4 *Pd, *normals_d, *grad_d; float *fpfh_d, *norm_d; unsigned short* split_d;
                    IBL_ARENA(d_groups, GroupDesc, G);
                    IBL_ARENA(d_grp_off, int, G + 1);
                    IBL_ARENA(Pd, float4, Nd + 1);
                    IBL_ARENA(normals_d, float4, Nd + 1);
                    IBL_ARENA(grad_d, float4, Nd + 1);
                    IBL_ARENA(fpfh_d, float, (int64_t)Nd * 33 + 64);
                    IBL_ARENA(split_d, unsigned short, (int64_t)Nd * 48 + 64);
                    IBL_ARENA(norm_d, float, (int64_t)Nd + 64);
                    st = ibl_stage_upload(ctx, d_groups, groups.data(), sizeof(GroupDesc) * (int64_t)G, s);
                    if (st) return st;
                    st = ibl_stage_upload(ctx, d_grp_off, grp_off.data(), sizeof(int) * (int64_t)(G + 1), s);
                    if (st) return st;
                    hipLaunchKernelGGL(ibl_group_gather_kernel, dim3((Nd + 255) / 256), dim3(256), 0, s, d_groups, G, det, det_off_dev, mem,
                                       mem_off_dev, d_grp_off, Pd);
                    IBL_LAUNCH_CHECK();
                    // a group's bounding box is the union of its instances' boxes (the gathered points are theirs, untouched), which
                    // the host holds with the instance features: the group grids are dimensioned without a read-back
                    std::vector<float> grp_bbox;
                    bool have_boxes = true;
                    for (int g = 0; g < G && have_boxes; ++g) have_boxes = feat[groups[g].pool] != nullptr;
                    if (have_boxes) {
                        grp_bbox.resize((size_t)G * 6);
                        for (int g = 0; g < G; ++g) {
                            float* o = &grp_bbox[6 * (size_t)g];
                            bool any = false;
                            for (int t = 0; t < 3; ++t) {
                                const int sg = groups[g].seg[t];
                                if (sg < 0) continue;
                                const int* off = groups[g].pool ? mem_off_host : det_off_host;
                                if (off[sg + 1] == off[sg]) continue;                    // empty instance: its stored box is zeros
                                const float* b = feat[groups[g].pool]->bbox + 6 * (size_t)sg;
                                for (int c = 0; c < 3; ++c) {
                                    o[c] = any ? std::min(o[c], b[c]) : b[c];
                                    o[3 + c] = any ? std::max(o[3 + c], b[3 + c]) : b[3 + c];
                                }
                                any = true;
                            }
                            if (!any) for (int c = 0; c < 6; ++c) o[c] = 0.0f;
                        }
                    }
                    st = ibl_features_on_batch(ctx, Pd, d_grp_off, grp_off.data(), G, have_boxes ? grp_bbox.data() : nullptr, voxel_size, grad_radius,
                                               grp_off[G0], Nd, normals_d, fpfh_d, split_d, norm_d, grad_d, s);
                    if (st) return st;
                    src.normals[2] = normals_d; src.fpfh[2] = fpfh_d; src.grad[2] = grad_d; src.split[2] = split_d; src.norm[2] = norm_d;
                }
                if (!copies.empty()) {
                    FeatCopy* d_copies;
                    IBL_ARENA(d_copies, FeatCopy, (int64_t)copies.size());
                    st = ibl_stage_upload(ctx, d_copies, copies.data(), sizeof(FeatCopy) * (int64_t)copies.size(), s);
                    if (st) return st;
                    for (size_t c0 = 0; c0 < copies.size(); c0 += 32768) {
                        const unsigned nc = (unsigned)std::min<size_t>(32768, copies.size() - c0);
                        hipLaunchKernelGGL(ibl_feat_assemble_kernel, dim3(8, nc), dim3(256), 0, s, d_copies + c0, src, normals, (float*)nullptr, grad);
                        IBL_LAUNCH_CHECK();
                    }
                }
                phase("recomputed groups + assemble");
                {
                    double fl = 0;
                    for (int j = 0; j < J; ++j) fl += 4.0 * 33.0 * (double)(job_off[j + 1] - job_off[j]) * (double)(job_off[J + j + 1] - job_off[J + j]);
                    ibl_prof_begin(IBL_PROF_ST_FEATMATCH, fl, s, &tok_match);
                }
                // matching reads the features in place (caches / recomputed groups), once per distinct pair
                st = ibl_stage_upload(ctx, d_sides, sides.data(), sizeof(SidePairs) * (int64_t)sides.size(), s);
                if (st) return st;
                if (!pairs.empty() && N > 0) {
                    st = ibl_stage_upload(ctx, d_pairs, pairs.data(), sizeof(FeatPair) * (int64_t)pairs.size(), s);
                    if (st) return st;
                    // (1) every source point's nearest target: source-query pairs, folded per job
                    // matrix-core filter + exact recheck (reg_featnn.hip); the VALU search only if its candidate list overflowed
                    const bool use_mfma = getenv("IBL_FEAT_VALU") == nullptr && !tl_force_valu;      // read per call: the tests compare both searches
                    bool over = !use_mfma;
                    dbg("plan uploads");
                    if (use_mfma) {
                        st = ibl_feat_search_mfma(ctx, d_pairs, n_pairs0, max_q, src, pair_idx, pair_d2, nullptr, nullptr, 0, pts0, &over, s);
                        if (st) return st;
                    }
                    dbg("forward search");
                    for (int p0 = 0; over && p0 < n_pairs0; p0 += 32768) {
                        const unsigned np = (unsigned)std::min(32768, n_pairs0 - p0);
                        hipLaunchKernelGGL(ibl_feat_pair_nn_kernel<false>, dim3((max_q + 255) / 256, np), dim3(256), 0, s, d_pairs + p0, src, pair_idx,
                                           pair_d2, (const int*)nullptr, (const int*)nullptr, 0);
                        IBL_LAUNCH_CHECK();
                    }
                    if (Ns > 0) {
                        hipLaunchKernelGGL(ibl_feat_fold_kernel, dim3((Ns + 255) / 256), dim3(256), 0, s, d_sides, d_pairs, pair_idx, pair_d2, d_job_off,
                                           J, 0, Ns, nn);
                        IBL_LAUNCH_CHECK();
                    }
                    dbg("forward fold");
                    // (2) the reverse search only for the target points that were matched (a third to a half of them): flag,
                    //     scan, list, search the listed queries; the other targets keep d2 = +inf and are never read
                    const int n1 = (int)(pair_pts - pts0);
                    const int n_pairs1 = (int)pairs.size() - n_pairs0;
                    if (n1 > 0 && n_pairs1 > 0 && Ns > 0) {
                        int *need, *need_pos, *need_list;
                        IBL_ARENA(need, int, (int64_t)n1 + 1);
                        IBL_ARENA(need_pos, int, (int64_t)n1 + 1);
                        IBL_ARENA(need_list, int, (int64_t)n1 + 1);
                        IBL_HIP_CHECK(hipMemsetAsync(need, 0, sizeof(int) * ((size_t)n1 + 1), s));
                        IBL_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)(pair_d2 + pts0), 0x7f800000, (size_t)n1, s));
                        IBL_HIP_CHECK(hipMemsetAsync(pair_idx + pts0, 0, sizeof(int) * (size_t)n1, s));
                        hipLaunchKernelGGL(ibl_feat_need_kernel, dim3((Ns + 255) / 256), dim3(256), 0, s, d_sides, d_pairs, d_job_off, J, nn, (int)pts0, need);
                        IBL_LAUNCH_CHECK();
                        size_t tmp_bytes = 0;
                        IBL_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, need, need_pos, n1 + 1, s));
                        unsigned char* tmp;
                        IBL_ARENA(tmp, unsigned char, (int64_t)tmp_bytes + 256);
                        IBL_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, need, need_pos, n1 + 1, s));
                        hipLaunchKernelGGL(ibl_feat_need_list_kernel, dim3((n1 + 255) / 256), dim3(256), 0, s, need, need_pos, n1, need_list);
                        IBL_LAUNCH_CHECK();
                        dbg("need list");
                        over = !use_mfma;
                        if (use_mfma) {
                            st = ibl_feat_search_mfma(ctx, d_pairs + n_pairs0, n_pairs1, max_q, src, pair_idx, pair_d2, need_pos, need_list, (int)pts0, n1,
                                                      &over, s);
                            if (st) return st;
                        }
                        for (int p0 = 0; over && p0 < n_pairs1; p0 += 32768) {
                            const unsigned np = (unsigned)std::min(32768, n_pairs1 - p0);
                            hipLaunchKernelGGL(ibl_feat_pair_nn_kernel<true>, dim3((max_q + 255) / 256, np), dim3(256), 0, s, d_pairs + n_pairs0 + p0, src,
                                               pair_idx, pair_d2, need_pos, need_list, (int)pts0);
                            IBL_LAUNCH_CHECK();
                        }
                    }
                    dbg("reverse search");
                    if (N > Ns) {
                        hipLaunchKernelGGL(ibl_feat_fold_kernel, dim3((N - Ns + 255) / 256), dim3(256), 0, s, d_sides, d_pairs, pair_idx, pair_d2, d_job_off,
                                           J, Ns, N, nn);
                        IBL_LAUNCH_CHECK();
                    }
                    dbg("reverse fold");
                } else if (N > 0) {
                    IBL_HIP_CHECK(hipMemsetAsync(nn, 0, sizeof(int) * (size_t)N, s));
                }
                // (the group scratch released here is reused by later kernels of the same stream only: no synchronisation)
            }
            ibl_prof_end(tok_match, s);
            phase("feature search");
            hipLaunchKernelGGL(ibl_mutual_kernel, dim3(J), dim3(256), 0, s, nn, d_job_off, J, 1, 9, corr, n_corr);
            IBL_LAUNCH_CHECK();
        }
        // ---- RANSAC ----------------------------------------------------------------------------------
        {
            ArenaMark m3(ctx);
            ibl_prof_begin(IBL_PROF_ST_RANSAC, 0.0, s, &tok_ransac);
            const double max_dist = voxel_size * global_dist_factor;
            const int max_round = RANSAC_MAX_ROUND;
            // a round's tables hold (active jobs) x (round size) hypotheses; when only a few jobs are left (wrong assignments
            // that never reach the confidence exit walk all 4 M), rounds grow to RANSAC_TAIL_ROUND so that they still fill the GPU
            const int64_t cap_slots = std::max<int64_t>((int64_t)J * max_round, (int64_t)RANSAC_TAIL_JOBS * RANSAC_TAIL_ROUND);
            const int64_t cap_blk = cap_slots / 256;
            float4* cp; unsigned char* flags; int *blk_cnt, *blk_off, *list;
            IBL_ARENA(cp, float4, 2 * (int64_t)Ns + 2);
            IBL_ARENA(flags, unsigned char, cap_slots);
            IBL_ARENA(blk_cnt, int, cap_blk + 1);
            IBL_ARENA(blk_off, int, cap_blk + 1);
            // survivors of the edge-length test of one round: ~1 % of the hypotheses on real clouds; a batch whose round exceeds the list is
            // redone once with a list that holds every hypothesis (tl_ransac_full_list, below)
            const int list_cap = (int)std::min<int64_t>(tl_ransac_full_list ? cap_slots + 65536 : cap_slots / 16 + 65536, (int64_t)1 << 27);
            IBL_ARENA(list, int, list_cap);
            int *e_inl, *e_job; double *e_err2, *e_T;
            IBL_ARENA(e_inl, int, list_cap);
            IBL_ARENA(e_job, int, list_cap);
            IBL_ARENA(e_err2, double, list_cap);
            IBL_ARENA(e_T, double, (int64_t)list_cap * 12);
            size_t tmp_bytes = 0;
            IBL_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, blk_cnt, blk_off, (int)(cap_blk + 1), s));
            unsigned char* tmp;
            IBL_ARENA(tmp, unsigned char, (int64_t)tmp_bytes + 256);
            hipLaunchKernelGGL(ibl_pack_corr_kernel, dim3(16, J), dim3(256), 0, s, P, d_job_off, J, corr, n_corr, cp);
            IBL_LAUNCH_CHECK();
            int *active, *done_flags;
            IBL_ARENA(active, int, J + 1);
            IBL_ARENA(done_flags, int, J + 1);
            unsigned* d_job_ids = nullptr;
            if (tl_job_ids) {
                IBL_ARENA(d_job_ids, unsigned, J + 1);
                st = ibl_stage_upload(ctx, d_job_ids, tl_job_ids, sizeof(unsigned) * (int64_t)J, s);
                if (st) return st;
            }
            hipLaunchKernelGGL(ibl_ransac_init_kernel, dim3((J + 63) / 64), dim3(64), 0, s, rs, n_corr, J, (long long)ransac_max_iter, max_dist, active,
                               job_id_base, d_job_ids);
            IBL_LAUNCH_CHECK();
            // Rounds are enqueued without asking the device anything: every per-round kernel finds the round's survivor count in
            // device memory (the last entry of the block-count scan), and the blocks of a job that has met its confidence bound
            // leave at once.  The host only looks between GROUPS of rounds -- after the first three (4 k + 32 k + 256 k hypotheses,
            // where most jobs stop) and then after every two -- to compact the list of running jobs and to stop.
            auto run_round = [&](int n_act, int round_size) -> int {
                const int nblk = round_size / 256;
                const int n_tab = n_act * nblk;           // tables are indexed by (slot in the active list, block)
                IBL_HIP_CHECK(hipMemsetAsync(flags, 0, (size_t)n_act * round_size, s));
                // large blocks amortise the dense Kabsch pass best, but a round of few jobs (or the 32 k round of all of them) is a few
                // hundred of them -- under two per CU, each ~120 us long: those rounds run as 4 k blocks (same flags: a hypothesis does not
                // know its block)
                constexpr int BIG = 1024 * RANSAC_BIG_SUBS;
                if (round_size >= BIG && (int64_t)(round_size / 16384) * n_act >= RANSAC_WIDE_MIN_BLOCKS)
                    hipLaunchKernelGGL(ibl_ransac_flag_kernel<RANSAC_BIG_SUBS>, dim3(round_size / BIG, n_act), dim3(256), 0, s, rs, cp, d_job_off, n_corr,
                                       (long long)ransac_max_iter, max_dist, 0.9, (unsigned)seed, (unsigned)(seed >> 32), job_id_base, round_size, flags,
                                       blk_cnt, active);
                else
                    hipLaunchKernelGGL(ibl_ransac_flag_kernel<4>, dim3(round_size / 4096, n_act), dim3(256), 0, s, rs, cp, d_job_off, n_corr,
                                       (long long)ransac_max_iter, max_dist, 0.9, (unsigned)seed, (unsigned)(seed >> 32), job_id_base, round_size, flags,
                                       blk_cnt, active);
                IBL_LAUNCH_CHECK();
                IBL_HIP_CHECK(hipMemsetAsync(blk_cnt + n_tab, 0, sizeof(int), s));
                IBL_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, blk_cnt, blk_off, n_tab + 1, s));
                const int* total_ptr = blk_off + n_tab;
                hipLaunchKernelGGL(ibl_ransac_scatter_kernel, dim3(nblk, n_act), dim3(256), 0, s, flags, blk_off, round_size, list, list_cap);
                IBL_LAUNCH_CHECK();
                const int sweep = (int)std::min<int64_t>(2048, ((int64_t)list_cap + 255) / 256);
                hipLaunchKernelGGL(ibl_ransac_transform_kernel, dim3(sweep), dim3(256), 0, s, rs, cp, d_job_off, n_corr, active, n_act, max_dist, 0.9,
                                   (unsigned)seed, (unsigned)(seed >> 32), job_id_base, round_size, blk_off, list, total_ptr, list_cap, ctx->d_status,
                                   e_job, e_T);
                IBL_LAUNCH_CHECK();
                hipLaunchKernelGGL(ibl_ransac_score_kernel, dim3(4096), dim3(256), 0, s, cp, d_job_off, n_corr, max_dist, total_ptr, list_cap, e_job,
                                   e_T, e_inl, e_err2);
                IBL_LAUNCH_CHECK();
                hipLaunchKernelGGL(ibl_ransac_fold_kernel, dim3(n_act), dim3(64), 0, s, rs, J, n_corr, (long long)ransac_max_iter,
                                   (flags_in & IBL_REG_FIXED_BUDGET) ? -1.0 : 0.99, round_size,
                                   blk_off, list, e_inl, e_err2, e_T, active);
                IBL_LAUNCH_CHECK();
                return IBL_OK;
            };
            std::vector<int> h_done(J, 0), h_list;
            long long walked = 0;
            int round_size = RANSAC_FIRST_ROUND;
            int n_act = J;
            int group = 3;
            while (walked < ransac_max_iter && n_act > 0) {
                for (int r = 0; r < group && walked < ransac_max_iter; ++r) {
                    if (n_act <= RANSAC_TAIL_JOBS && round_size == max_round) round_size = RANSAC_TAIL_ROUND;
                    st = run_round(n_act, round_size);
                    if (st) return st;
                    walked += round_size;
                    if (round_size < max_round) round_size = std::min(max_round, round_size * 8);
                }
                if (walked >= ransac_max_iter) break;
                hipLaunchKernelGGL(ibl_ransac_done_kernel, dim3((J + 63) / 64), dim3(64), 0, s, rs, J, done_flags);
                IBL_LAUNCH_CHECK();
                IBL_HIP_CHECK(hipMemcpyAsync(h_done.data(), done_flags, sizeof(int) * J, hipMemcpyDeviceToHost, s));
                IBL_HIP_CHECK(hipStreamSynchronize(s));
                h_list.clear();
                for (int j = 0; j < J; ++j) if (!h_done[j]) h_list.push_back(j);
                n_act = (int)h_list.size();
                if (n_act > 0) {
                    st = ibl_stage_upload(ctx, active, h_list.data(), sizeof(int) * (int64_t)n_act, s);
                    if (st) return st;
                }
                group = n_act <= RANSAC_TAIL_JOBS ? 1 : 2;
            }
        }
    }
    ibl_prof_end(tok_ransac, s);
    phase("ransac");
    // ---- ICP ------------------------------------------------------------------------------------------
    ibl_prof_begin(IBL_PROF_ST_ICP, 0.0, s, &tok_icp);
    {
        double* partial; int* icp_nn; float* icp_d2;
        IBL_ARENA(partial, double, (int64_t)J * ICP_BPJ * ICP_NACC);
        IBL_ARENA(icp_nn, int, Ns + 64);
        IBL_ARENA(icp_d2, float, Ns + 64);
        hipLaunchKernelGGL(ibl_icp_init_kernel, dim3((J + 63) / 64), dim3(64), 0, s, is, J, rs, colored ? 0 : 1);
        IBL_LAUNCH_CHECK();
        const double lambda_geometric = 0.968;
        const int max_iter = 30;
        int max_side = 0;
        for (int j = 0; j < J; ++j) max_side = std::max(max_side, job_off[j + 1] - job_off[j]);
        const unsigned chunks = (unsigned)std::max(1, (max_side + 255) / 256);
        // from ICP_GROUP_FROM on the kernels walk the list of jobs still running (written by the previous update: list it & 1, count
        // act_cnt[it]) on a grid of ICP_ACT_Y block rows instead of one row per job
        int *act_list, *act_cnt;
        IBL_ARENA(act_list, int, 2 * (int64_t)J + 64);
        IBL_ARENA(act_cnt, int, max_iter + 8);
        IBL_HIP_CHECK(hipMemsetAsync(act_cnt, 0, sizeof(int) * (max_iter + 8), s));
        const unsigned act_y = (unsigned)std::min(J, ICP_ACT_Y);
        for (int it = 0; it <= max_iter; ++it) {
            const bool listed = it >= ICP_GROUP_FROM;
            const int* cur_list = listed ? act_list + (size_t)(it & 1) * J : nullptr;
            const int* cur_cnt = listed ? act_cnt + it : nullptr;
            int* nxt_list = it + 1 >= ICP_GROUP_FROM ? act_list + (size_t)((it + 1) & 1) * J : nullptr;
            int* nxt_cnt = it + 1 >= ICP_GROUP_FROM ? act_cnt + it + 1 : nullptr;
            if (!listed)
                hipLaunchKernelGGL(ibl_icp_nn_kernel, dim3(chunks, J), dim3(256), 0, s, gC, P, d_job_off, J, d_piece_off, is, (float)max_dist_icp,
                                   (float)(max_dist_icp * max_dist_icp), icp_nn, icp_d2);
            else
                hipLaunchKernelGGL(ibl_icp_nn_group_kernel<ICP_LPQ>, dim3(chunks * ICP_LPQ, act_y), dim3(256), 0, s, gC, P, d_job_off, J, d_piece_off, is,
                                   (float)max_dist_icp, (float)(max_dist_icp * max_dist_icp), icp_nn, icp_d2, cur_list, cur_cnt);
            IBL_LAUNCH_CHECK();
            hipLaunchKernelGGL(ibl_icp_step_kernel, dim3(ICP_BPJ, listed ? act_y : (unsigned)J), dim3(256), 0, s, P, normals, grad, d_job_off, J, is, icp_nn,
                               icp_d2, colored ? 1 : 0, sqrt(lambda_geometric), sqrt(1.0 - lambda_geometric), partial, cur_list, cur_cnt);
            IBL_LAUNCH_CHECK();
            hipLaunchKernelGGL(ibl_icp_update_kernel, dim3(listed ? act_y : (unsigned)J), dim3(64), 0, s, is, J, d_job_off, partial, colored ? 1 : 0, max_iter,
                               1e-6, 1e-6, cur_list, cur_cnt, nxt_list, nxt_cnt);
            IBL_LAUNCH_CHECK();
        }
    }
    ibl_prof_end(tok_icp, s);
    phase("icp");
    // ---- results ----------------------------------------------------------------------------------------
    std::vector<IcpState> h_is(J);
    IBL_HIP_CHECK(hipMemcpyAsync(h_is.data(), is, sizeof(IcpState) * J, hipMemcpyDeviceToHost, s));
    std::vector<RansacState> h_rs;
    if (rs && (T_ransac_out || ransac_stats_out)) {
        h_rs.resize(J);
        IBL_HIP_CHECK(hipMemcpyAsync(h_rs.data(), rs, sizeof(RansacState) * J, hipMemcpyDeviceToHost, s));
    }
    std::vector<double> h_means((size_t)J * 6);
    IBL_HIP_CHECK(hipMemcpyAsync(h_means.data(), d_means, sizeof(double) * J * 6, hipMemcpyDeviceToHost, s));
    int h_status = 0;
    IBL_HIP_CHECK(hipMemcpyAsync(&h_status, ctx->d_status, sizeof(int), hipMemcpyDeviceToHost, s));
    IBL_HIP_CHECK(hipStreamSynchronize(s));
    ibl_stage_reset(ctx);
    if (tok_icp || tok_ransac) {                 // stage units known now: iterations run per job, hypotheses walked
        double icp_b = 0, hyp = 0;
        for (int j = 0; j < J; ++j) {
            icp_b += 56.0 * (double)(job_off[j + 1] - job_off[j]) * (double)std::max(1, h_is[j].iter);
            if (!h_rs.empty()) hyp += (double)h_rs[j].walked;
        }
        ibl_prof_set_units(tok_icp, icp_b);
        ibl_prof_set_units(tok_ransac, hyp);
    }
    // Overflowed lists make the results of this pass unusable.  Both are decided here, in one place: a pass that overflowed the RANSAC
    // survivor list (near-identical clouds, a loose edge criterion) is redone with a list that holds every hypothesis of a round (same
    // hypotheses, same fold order: the result of a pass whose list never overflowed); one whose matrix-core feature search overflowed
    // its candidate list is redone with the VALU search, which has none; a pass that hit both asks for both at once.
    if ((h_status & IBL_ST_RANSAC_OVERFLOW) && tl_ransac_full_list)
        return ibl_set_error(IBL_ERR_OVERFLOW, "ransac: the surviving hypotheses of one round exceed the list capacity");
    if (h_status & IBL_ST_RANSAC_OVERFLOW) *redo |= REDO_RANSAC_FULL_LIST;
    if ((h_status & IBL_ST_FEAT_OVERFLOW) && !tl_force_valu) *redo |= REDO_FEAT_VALU;
    if (*redo) return IBL_OK;
    for (int j = 0; j < J; ++j) {
        for (int i = 0; i < 16; ++i) T_out[16 * j + i] = h_is[j].T[i];
        rmse_out[j] = h_is[j].rmse;
        fitness_out[j] = h_is[j].fitness;
        if (means_out) for (int i = 0; i < 6; ++i) means_out[6 * j + i] = h_means[6 * j + i];
        if (T_ransac_out) for (int i = 0; i < 16; ++i) T_ransac_out[16 * j + i] = rs ? h_rs[j].best_T[i] : ((i % 5) == 0 ? 1.0 : 0.0);
        if (ransac_stats_out) {
            ransac_stats_out[3 * j] = rs ? h_rs[j].walked : 0;
            ransac_stats_out[3 * j + 1] = rs ? h_rs[j].validated : 0;
            ransac_stats_out[3 * j + 2] = rs ? h_rs[j].best_inl : 0;
        }
    }
    return IBL_OK;
}

extern "C" int ibl_register_batch_cached(ibl_reg_ctx* ctx, const float* det_pts4, const int32_t* det_off_dev, const int32_t* det_off_host,
                                         int n_det_seg, const float* mem_pts4, const int32_t* mem_off_dev, const int32_t* mem_off_host,
                                         int n_mem_seg, const int32_t* job_src_seg, const int32_t* job_tgt_seg, int n_jobs,
                                         double voxel_size, double global_dist_factor, double local_dist_factor, uint64_t seed,
                                         uint32_t job_id_base, int64_t ransac_max_iter, int flags,
                                         const ibl_instance_features* det_features, const ibl_instance_features* mem_features,
                                         double* T_out, double* rmse_out, double* fitness_out, double* means_out, double* T_ransac_out,
                                         int64_t* ransac_stats_out, int64_t* reuse_stats_out, void* stream) {
    // at most three passes: the first, one with the lists it asked to avoid, and one more if that pass overflowed the OTHER list
    int st = IBL_OK;
    for (int pass = 0; pass < 3; ++pass) {
        int redo = 0;
        st = register_batch_cached_pass(ctx, det_pts4, det_off_dev, det_off_host, n_det_seg, mem_pts4, mem_off_dev, mem_off_host, n_mem_seg,
                                        job_src_seg, job_tgt_seg, n_jobs, voxel_size, global_dist_factor, local_dist_factor, seed, job_id_base,
                                        ransac_max_iter, flags, det_features, mem_features, T_out, rmse_out, fitness_out, means_out,
                                        T_ransac_out, ransac_stats_out, reuse_stats_out, stream, &redo);
        if (st != IBL_OK || !redo) break;
        if (redo & REDO_RANSAC_FULL_LIST) tl_ransac_full_list = true;
        if (redo & REDO_FEAT_VALU) tl_force_valu = true;
        hipLaunchKernelGGL(ibl_status_set_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, ctx->d_status,
                           ((redo & REDO_RANSAC_FULL_LIST) ? IBL_ST_RANSAC_REDONE : 0) | ((redo & REDO_FEAT_VALU) ? IBL_ST_FEAT_REDONE : 0));
    }
    tl_ransac_full_list = false;
    tl_force_valu = false;
    return st;
}

// ------------------------------------------------------------------------------------------------
// whole-memory hash grid + evaluate_registration
// ------------------------------------------------------------------------------------------------
struct ibl_memgrid {
    float cell, inv;
    int64_t n;
    int n_cells;
    unsigned long long hmask;
    float4* sorted;                 // points in cell order
    unsigned long long* ukeys;      // unique cell keys
    int* ustart;                    // [n_cells + 1]
    unsigned long long* tkeys;      // hash table keys (EMPTY = ~0)
    int* tvals;                     // cell index
};

#define MG_EMPTY 0xFFFFFFFFFFFFFFFFull
__device__ __forceinline__ unsigned long long mg_key(int ix, int iy, int iz) {
    return ((unsigned long long)(unsigned)(ix + (1 << 20)) << 42) | ((unsigned long long)(unsigned)(iy + (1 << 20)) << 21) |
           (unsigned long long)(unsigned)(iz + (1 << 20));
}
__device__ __forceinline__ unsigned long long mg_hash(unsigned long long k) {
    k ^= k >> 33; k *= 0xFF51AFD7ED558CCDull; k ^= k >> 33; k *= 0xC4CEB9FE1A85EC53ull; k ^= k >> 33;
    return k;
}

__global__ __launch_bounds__(256) void ibl_mg_key_kernel(const float4* __restrict__ pts, int64_t n, float inv, unsigned long long* __restrict__ keys,
                                                         int* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    keys[i] = mg_key((int)floorf(p.x * inv), (int)floorf(p.y * inv), (int)floorf(p.z * inv));
    vals[i] = (int)i;
}

__global__ __launch_bounds__(256) void ibl_mg_heads_kernel(const unsigned long long* __restrict__ skeys, int64_t n, int* __restrict__ head) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    head[i] = (i == 0 || skeys[i] != skeys[i - 1]) ? 1 : 0;
}

__global__ __launch_bounds__(256) void ibl_mg_cells_kernel(const unsigned long long* __restrict__ skeys, const int* __restrict__ head,
                                                           const int* __restrict__ head_scan, const int* __restrict__ order,
                                                           const float4* __restrict__ pts, int64_t n, unsigned long long* __restrict__ ukeys,
                                                           int* __restrict__ ustart, float4* __restrict__ sorted) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    sorted[i] = pts[order[i]];
    if (head[i]) { const int c = head_scan[i]; ukeys[c] = skeys[i]; ustart[c] = (int)i; }
}

__global__ __launch_bounds__(256) void ibl_mg_insert_kernel(const unsigned long long* __restrict__ ukeys, int n_cells, unsigned long long hmask,
                                                            unsigned long long* __restrict__ tkeys, int* __restrict__ tvals) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n_cells) return;
    const unsigned long long k = ukeys[c];
    unsigned long long h = mg_hash(k) & hmask;
    while (true) {
        const unsigned long long prev = atomicCAS(&tkeys[h], MG_EMPTY, k);
        if (prev == MG_EMPTY) { tvals[h] = c; return; }
        h = (h + 1) & hmask;
    }
}

extern "C" int ibl_memgrid_build(ibl_reg_ctx* ctx, const float* mem_pts4, int64_t n, double cell, ibl_memgrid** out, void* stream) {
    if (!ctx || !mem_pts4 || !out || n <= 0 || n > 0x7FFFFFF0ll || cell <= 0) return ibl_set_error(IBL_ERR_ARG, "ibl_memgrid_build: bad argument");
    hipStream_t s = (hipStream_t)stream;
    std::unique_ptr<ibl_memgrid> owner(new ibl_memgrid());      // freed on every error return below
    ibl_memgrid* g = owner.get();
    g->cell = (float)cell; g->inv = 1.0f / (float)cell; g->n = n;
    const float4* P = reinterpret_cast<const float4*>(mem_pts4);
    // persistent part (stays allocated in the arena until the context is destroyed)
    IBL_ARENA(g->sorted, float4, n + 1);
    IBL_ARENA(g->ukeys, unsigned long long, n + 1);
    IBL_ARENA(g->ustart, int, n + 2);
    {
        ArenaMark scratch(ctx);
        unsigned long long *keys, *skeys; int *vals, *order, *head, *hscan; unsigned char* tmp;
        IBL_ARENA(keys, unsigned long long, n);
        IBL_ARENA(skeys, unsigned long long, n);
        IBL_ARENA(vals, int, n);
        IBL_ARENA(order, int, n);
        IBL_ARENA(head, int, n + 1);
        IBL_ARENA(hscan, int, n + 1);
        const unsigned nb = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(ibl_mg_key_kernel, dim3(nb), dim3(256), 0, s, P, n, g->inv, keys, vals);
        IBL_LAUNCH_CHECK();
        size_t t1 = 0, t2 = 0;
        IBL_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, t1, keys, skeys, vals, order, (int)n, 0, 63, s));
        IBL_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, t2, head, hscan, (int)n, s));
        IBL_ARENA(tmp, unsigned char, (int64_t)std::max(t1, t2) + 256);
        IBL_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp, t1, keys, skeys, vals, order, (int)n, 0, 63, s));
        hipLaunchKernelGGL(ibl_mg_heads_kernel, dim3(nb), dim3(256), 0, s, skeys, n, head);
        IBL_LAUNCH_CHECK();
        IBL_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp, t2, head, hscan, (int)n, s));
        hipLaunchKernelGGL(ibl_mg_cells_kernel, dim3(nb), dim3(256), 0, s, skeys, head, hscan, order, P, n, g->ukeys, g->ustart, g->sorted);
        IBL_LAUNCH_CHECK();
        int last_scan = 0, last_head = 0;
        IBL_HIP_CHECK(hipMemcpyAsync(&last_scan, hscan + (n - 1), sizeof(int), hipMemcpyDeviceToHost, s));
        IBL_HIP_CHECK(hipMemcpyAsync(&last_head, head + (n - 1), sizeof(int), hipMemcpyDeviceToHost, s));
        IBL_HIP_CHECK(hipStreamSynchronize(s));
        g->n_cells = last_scan + last_head;
        const int nn = (int)n;
        IBL_HIP_CHECK(hipMemcpyAsync(g->ustart + g->n_cells, &nn, sizeof(int), hipMemcpyHostToDevice, s));
        IBL_HIP_CHECK(hipStreamSynchronize(s));
    }
    // The table is dimensioned from the OCCUPIED CELLS, now that they are counted (round 4): it used to hold 2 n slots -- 128 M slots = 1.5 GB
    // for a 10 000-instance memory whose 50 M surface points occupy a few million 4 cm cells -- so that every probe of the evaluation was
    // a first touch of HBM (1.9 GB moved per launch for 158 MB of points).  At <= 1 / 3 load the table of the same memory is ~100 MB: it stays
    // in the Infinity Cache, and a miss walks 1.5 slots on average.  Same cells, same points, same minima.
    unsigned long long H = 1024;
    while (H < (unsigned long long)g->n_cells * 3) H <<= 1;
    g->hmask = H - 1;
    IBL_ARENA(g->tkeys, unsigned long long, (int64_t)H);
    IBL_ARENA(g->tvals, int, (int64_t)H);
    IBL_HIP_CHECK(hipMemsetAsync(g->tkeys, 0xFF, sizeof(unsigned long long) * H, s));
    hipLaunchKernelGGL(ibl_mg_insert_kernel, dim3((g->n_cells + 255) / 256), dim3(256), 0, s, g->ukeys, g->n_cells, g->hmask, g->tkeys, g->tvals);
    IBL_LAUNCH_CHECK();
    IBL_HIP_CHECK(hipStreamSynchronize(s));
    *out = owner.release();
    return IBL_OK;
}

extern "C" int ibl_memgrid_destroy(ibl_memgrid* g) {
    delete g;       // device memory belongs to the context arena
    return IBL_OK;
}

struct EvalJob {
    double T[12];
    int begin, end;      // detected point range (all cleaned detected clouds of the job's frame)
    long long out;       // offset of this job's per-point distances (ibl_evaluate_points)
};

// grid (ICP_BPJ, J): fitness / rmse partials of evaluate_registration against the whole memory
__global__ __launch_bounds__(256) void ibl_evaluate_kernel(ibl_memgrid g, const float4* __restrict__ det, const EvalJob* __restrict__ jobs,
                                                           float thr, float thr2, double* __restrict__ partial /* [J][BPJ][2] */,
                                                           float* __restrict__ d2_out /* per (job, point) or null */, int prune) {
    const int j = blockIdx.y;
    const EvalJob job = jobs[j];
    double cnt = 0, err2 = 0;
    for (int i = job.begin + blockIdx.x * 256 + threadIdx.x; i < job.end; i += ICP_BPJ * 256) {
        const float4 s4 = det[i];
        double p[3];
        xform_d(job.T, s4.x, s4.y, s4.z, p);
        const float qx = (float)p[0], qy = (float)p[1], qz = (float)p[2];
        float best = thr2;
        bool found = false;
        const int x0 = (int)floorf((qx - thr) * g.inv), x1 = (int)floorf((qx + thr) * g.inv);
        const int y0 = (int)floorf((qy - thr) * g.inv), y1 = (int)floorf((qy + thr) * g.inv);
        const int z0 = (int)floorf((qz - thr) * g.inv), z1 = (int)floorf((qz + thr) * g.inv);
        // The query's own cell first, then the (up to seven) others of its +-thr box only while they can still hold a closer point:
        // a neighbouring cell lies behind the face it shares with the own cell, so the distance to that face (per axis that differs) bounds
        // every point of it from below.  Round 4: an inlier's nearest point is millimetres away and the faces are centimetres away, so ~1.5
        // instead of 8 cells are looked up and read (1.6 of the 1.9 GB a launch moved were the points of those cells).  The bound is
        // conservative -- the slack covers the rounding of floorf(x * inv) against the geometric face, which grows with the coordinate -- and
        // a cell is skipped only when its bound already reaches the best: the minimum is that of the full scan.
        const int hx = (int)floorf(qx * g.inv), hy = (int)floorf(qy * g.inv), hz = (int)floorf(qz * g.inv);
        auto scan_cell = [&](int ix, int iy, int iz) {
            const unsigned long long k = mg_key(ix, iy, iz);
            unsigned long long h = mg_hash(k) & g.hmask;
            int c = -1;
            while (true) {
                const unsigned long long tk = g.tkeys[h];
                if (tk == k) { c = g.tvals[h]; break; }
                if (tk == MG_EMPTY) break;
                h = (h + 1) & g.hmask;
            }
            if (c < 0) return;
            const int b = g.ustart[c], e = g.ustart[c + 1];
            for (int t = b; t < e; t += 4) {           // four points in flight (past the end: the last point again -- a repeat changes no minimum)
                float4 m[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) m[u] = g.sorted[min(t + u, e - 1)];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float d2 = dist2f(qx, qy, qz, m[u].x, m[u].y, m[u].z);
                    if (d2 < best) { best = d2; found = true; }
                }
            }
        };
        auto face_gap = [&](int i, int hcell, float q) {   // distance from q to the face between its own cell and cell i on this axis (0: same cell)
            if (i == hcell) return 0.0f;
            const float face = (float)(i < hcell ? hcell : hcell + 1) * g.cell;
            return fmaxf(fabsf(q - face) - (1e-3f * g.cell + 5e-7f * fabsf(face)), 0.0f);
        };
        scan_cell(hx, hy, hz);
        for (int ix = x0; ix <= x1; ++ix) {
            const float gx = face_gap(ix, hx, qx);
            for (int iy = y0; iy <= y1; ++iy) {
                const float gy = face_gap(iy, hy, qy);
                for (int iz = z0; iz <= z1; ++iz) {
                    if (ix == hx && iy == hy && iz == hz) continue;
                    const float gz = face_gap(iz, hz, qz);
                    if (prune && gx * gx + gy * gy + gz * gz >= best) continue;
                    scan_cell(ix, iy, iz);
                }
            }
        }
        if (found) { cnt += 1.0; err2 += (double)best; }
        if (d2_out) d2_out[job.out + (i - job.begin)] = found ? best : INFINITY;
    }
    __shared__ double sh[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    cnt = wave_sum_d(cnt); err2 = wave_sum_d(err2);
    if (lane == 0) { sh[0][wave] = cnt; sh[1][wave] = err2; }
    __syncthreads();
    if (threadIdx.x < 2)
        partial[((int64_t)j * ICP_BPJ + blockIdx.x) * 2 + threadIdx.x] =
            ((sh[threadIdx.x][0] + sh[threadIdx.x][1]) + sh[threadIdx.x][2]) + sh[threadIdx.x][3];
}

static int evaluate_impl(ibl_reg_ctx* ctx, const ibl_memgrid* grid, const float* det_pts4, const int32_t* job_begin, const int32_t* job_end,
                         const double* T_global, int n_jobs, double threshold, double* rmse_out, double* fitness_out, float* d2_out, void* stream);

extern "C" int ibl_evaluate_batch(ibl_reg_ctx* ctx, const ibl_memgrid* grid, const float* det_pts4, const int32_t* job_begin,
                                  const int32_t* job_end, const double* T_global, int n_jobs, double threshold, double* rmse_out,
                                  double* fitness_out, void* stream) {
    return evaluate_impl(ctx, grid, det_pts4, job_begin, job_end, T_global, n_jobs, threshold, rmse_out, fitness_out, nullptr, stream);
}

extern "C" int ibl_evaluate_points(ibl_reg_ctx* ctx, const ibl_memgrid* grid, const float* det_pts4, const int32_t* job_begin,
                                   const int32_t* job_end, const double* T_global, int n_jobs, double threshold, float* d2_out,
                                   double* rmse_out, double* fitness_out, void* stream) {
    if (!d2_out) return ibl_set_error(IBL_ERR_ARG, "ibl_evaluate_points: d2_out is null");
    return evaluate_impl(ctx, grid, det_pts4, job_begin, job_end, T_global, n_jobs, threshold, rmse_out, fitness_out, d2_out, stream);
}

static int evaluate_impl(ibl_reg_ctx* ctx, const ibl_memgrid* grid, const float* det_pts4, const int32_t* job_begin, const int32_t* job_end,
                         const double* T_global, int n_jobs, double threshold, double* rmse_out, double* fitness_out, float* d2_out, void* stream) {
    if (!ctx || !grid || !det_pts4 || !job_begin || !job_end || !T_global || !rmse_out || !fitness_out || n_jobs <= 0 || threshold <= 0)
        return ibl_set_error(IBL_ERR_ARG, "ibl_evaluate_batch: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ArenaMark mark(ctx);
    const int J = n_jobs;
    std::vector<EvalJob> jobs(J);
    for (int j = 0; j < J; ++j) {
        for (int t = 0; t < 12; ++t) jobs[j].T[t] = T_global[16 * j + t];
        jobs[j].begin = job_begin[j]; jobs[j].end = job_end[j];
        if (job_end[j] < job_begin[j]) return ibl_set_error(IBL_ERR_ARG, "ibl_evaluate_batch: bad point range");
        jobs[j].out = j == 0 ? 0 : jobs[j - 1].out + (jobs[j - 1].end - jobs[j - 1].begin);
    }
    EvalJob* d_jobs; double* partial;
    IBL_ARENA(d_jobs, EvalJob, J);
    IBL_ARENA(partial, double, (int64_t)J * ICP_BPJ * 2);
    IBL_HIP_CHECK(hipMemcpyAsync(d_jobs, jobs.data(), sizeof(EvalJob) * J, hipMemcpyHostToDevice, s));
    const char* efs = getenv("IBL_EVAL_FULLSCAN");          // diagnostics: 1 = every cell of the query's box (the tests compare both)
    const int prune = !(efs && atoi(efs));
    void* tok;
    ibl_prof_begin(IBL_PROF_ST_EVAL, 24.0 * (double)(jobs[J - 1].out + (jobs[J - 1].end - jobs[J - 1].begin)), s, &tok);
    hipLaunchKernelGGL(ibl_evaluate_kernel, dim3(ICP_BPJ, J), dim3(256), 0, s, *grid, reinterpret_cast<const float4*>(det_pts4), d_jobs,
                       (float)threshold, (float)(threshold * threshold), partial, d2_out, prune);
    ibl_prof_end(tok, s);
    IBL_LAUNCH_CHECK();
    std::vector<double> h((size_t)J * ICP_BPJ * 2);
    IBL_HIP_CHECK(hipMemcpyAsync(h.data(), partial, sizeof(double) * h.size(), hipMemcpyDeviceToHost, s));
    IBL_HIP_CHECK(hipStreamSynchronize(s));
    for (int j = 0; j < J; ++j) {
        double cnt = 0, err2 = 0;
        for (int b = 0; b < ICP_BPJ; ++b) { cnt += h[((size_t)j * ICP_BPJ + b) * 2]; err2 += h[((size_t)j * ICP_BPJ + b) * 2 + 1]; }
        const int ns = job_end[j] - job_begin[j];
        fitness_out[j] = ns > 0 ? cnt / (double)ns : 0.0;
        rmse_out[j] = cnt > 0 ? std::sqrt(err2 / cnt) : 0.0;
    }
    return IBL_OK;
}
