// reg_common.h -- shared internals of the registration kernels (grids, context arena, small
// double-precision linear algebra used on the device).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ibl_common.h"

// One cloud ("segment") of a batch: uniform grid over its bounding box.
struct SegGrid {
    float minx, miny, minz, inv;   // inv = 1 / cell size
    int nx, ny, nz;
    int cell_base;                 // offset of this segment's cells in the batch-wide cell arrays
};

// Grid over a batch of clouds: points are sorted by (segment, cell) -- stable, so the candidate
// enumeration order (and with it every reduction order) is deterministic.
struct BatchGrid {
    const SegGrid* seg;        // [S]
    const int* cell_start;     // [total_cells + 1]
    const float4* sorted_pts;  // [N] (x, y, z, intensity) in cell order
    const int* order;          // [N] sorted position -> original point index (batch-global)
    int n_seg;
    // tiles (ibl_build_tile_grid only): cubes of ts^3 cells, one workgroup each in the LDS-staged k-NN kernels (reg_knn.hip)
    const int* tile_base;      // [S + 1] first tile of every segment, or null
    const int* tile_seg;       // [n_tiles] segment of every tile (a lookup instead of a binary search over tile_base), or null
    int n_tiles;
    int ts;
};

// ------------------------------------------------------------------------------------------------
// context: explicit device arena (bump allocator), created / destroyed through the C-ABI
// ------------------------------------------------------------------------------------------------
struct ibl_reg_ctx {
    unsigned char* base = nullptr;
    int64_t size = 0;
    int64_t used = 0;
    int64_t high_water = 0;
    int device = 0;
    int* d_status = nullptr;      // device status word(s): bit flags set by kernels (overflow etc.)
    // pinned host staging (bump allocator, reset by the entry point that used it after its final synchronisation): plan tables are
    // copied here before their hipMemcpyAsync, so the copy is asynchronous (pageable sources make the runtime wait for the stream)
    // and the caller's vectors may die at once
    unsigned char* pin = nullptr;
    int64_t pin_size = 0;
    int64_t pin_used = 0;
};

// copies `bytes` of host data to the device through the context's pinned staging buffer, asynchronously on `s`; falls back to a
// synchronous copy when the staging buffer is exhausted
int ibl_stage_upload(ibl_reg_ctx* ctx, void* dst_dev, const void* src_host, int64_t bytes, hipStream_t s);
static inline void ibl_stage_reset(ibl_reg_ctx* ctx) { ctx->pin_used = 0; }

struct ArenaMark {
    ibl_reg_ctx* ctx;
    int64_t mark;
    explicit ArenaMark(ibl_reg_ctx* c) : ctx(c), mark(c->used) {}
    ~ArenaMark() { ctx->used = mark; }
};

template <typename T>
static inline T* arena_alloc(ibl_reg_ctx* ctx, int64_t count, bool* ok) {
    int64_t bytes = ibl_align_up(count * (int64_t)sizeof(T), 256);
    if (bytes < 256) bytes = 256;
    if (ctx->used + bytes > ctx->size) {
        *ok = false;
        return nullptr;
    }
    T* p = reinterpret_cast<T*>(ctx->base + ctx->used);
    ctx->used += bytes;
    if (ctx->used > ctx->high_water) ctx->high_water = ctx->used;
    return p;
}

#define IBL_ARENA(ptr, type, count)                                                                  \
    do {                                                                                             \
        bool _ok = true;                                                                             \
        ptr = arena_alloc<type>(ctx, (count), &_ok);                                                 \
        if (!_ok)                                                                                    \
            return ibl_set_error(IBL_ERR_ARENA, "device arena exhausted (%lld of %lld bytes used, need %lld more) at %s:%d", \
                                 (long long)ctx->used, (long long)ctx->size,                         \
                                 (long long)((count) * (int64_t)sizeof(type)), __FILE__, __LINE__);  \
    } while (0)

#define IBL_ERR_ARENA (-5)
#define IBL_ERR_OVERFLOW (-6)

// status bits written by kernels
#define IBL_ST_GRID_OVERFLOW 1
#define IBL_ST_KNN_SLOWPATH 2
#define IBL_ST_FEAT_OVERFLOW 4       // the matrix-core feature search overflowed its candidate list: the call is redone with the VALU search
#define IBL_ST_RANSAC_OVERFLOW 8     // more surviving hypotheses in a round than the list holds: the call is redone with a full-size list
#define IBL_ST_FEAT_REDONE 16        // (sticky, informational) a registration call was redone with the VALU feature search
#define IBL_ST_RANSAC_REDONE 32      // (sticky, informational) a registration call was redone with a full-size survivor list

// grid construction (reg_grid.hip)
int ibl_build_batch_grid(ibl_reg_ctx* ctx, const float4* pts, const int* seg_off_dev, const int* seg_off_host, int n_seg,
                         float cell, int64_t max_cells, BatchGrid* out, hipStream_t s);
// The same grid without the read-back of the table size: `cells_bound` (>= the number of cells the dims kernel will find, e.g. from
// bounding boxes the host already holds) sizes the tables; a segment that does not fit collapses to one cell (status bit).
int ibl_build_batch_grid_bounded(ibl_reg_ctx* ctx, const float4* pts, const int* seg_off_dev, const int* seg_off_host, int n_seg,
                                 float cell, int64_t cells_bound, BatchGrid* out, hipStream_t s);
// Grid for the hybrid k-NN kernels, sized on the HOST from the segments' bounding boxes (bbox_host [S][6], the values
// ibl_launch_bbox produced): no read-back, no dims kernel.  The cell of a segment follows its point density (about max_nn points
// inside a ball of two cells, clamped to [radius / 12, radius]), tiles are cubes of ts^3 cells.
// Tile-grid tuning: the staging cube of a tile reaches ibl_knn_rho() cells past it, and the cell of a segment is sized so that the
// ball of that many cells holds ibl_knn_safety() x max_nn points at the segment's mean surface density (clouds are uneven: edges,
// corners and seams see half-empty balls).  Environment overrides IBL_KNN_RHO / IBL_KNN_SAFETY are for measurements.
double ibl_knn_safety();
int ibl_knn_rho();
int ibl_build_tile_grid(ibl_reg_ctx* ctx, const float4* pts, const int* seg_off_dev, const int* seg_off_host, int n_seg,
                        const float* bbox_host, double radius, int max_nn, int ts, int64_t max_cells, BatchGrid* out, hipStream_t s);

// per-segment axis-aligned bounding boxes [S][6] = (min xyz, max xyz); empty segments read as zeros
int ibl_launch_bbox(const float4* pts, const int* seg_off_dev, int n_seg, float* bbox_dev, hipStream_t s);

// feature arrays the registration driver reads in place: [0] detected-pool instance features, [1] memory-pool instance
// features, [2] groups recomputed in the context of their job
// (split[k] may be null -- an instance-feature set kept without its operand rows: the search then builds them from fpfh[k] / norm[k])
struct FeatSources { const float4* normals[3]; const float* fpfh[3]; const float4* grad[3]; const unsigned short* split[3]; const float* norm[3]; };
// one (query instance, database instance) nearest-neighbour search; kind = FeatSources index, src = point offset there,
// out = offset of the query instance's results in the pair output arrays
struct FeatPair { int qkind, qsrc, qcnt, dkind, dsrc, dcnt, out, pad; };

// reg_featnn.hip: the searches of `pairs` on the matrix cores (bf16 hi/lo split products as a rigorous filter, exact fp32
// chains for the few candidates that pass it).  need_pos / need_list (or NULL): only the listed queries (see
// ibl_feat_need_kernel).  *overflow is set when the candidate list did not fit (the caller falls back to the VALU search).
int ibl_feat_search_mfma(ibl_reg_ctx* ctx, const FeatPair* d_pairs, int n_pairs, int max_q, const FeatSources& src, int* pair_idx,
                         float* pair_d2, const int* need_pos, const int* need_list, int out0, int64_t out_count, bool* overflow,
                         hipStream_t s);

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
#ifdef __HIPCC__
// ---- the fp16 search operands of the feature search (reg_featnn.hip; layout and error budget in its header) ----
// The centring constant: distances do not change when the same vector is subtracted from every row, the filter's error bound
// C (|q|^2 + |t|^2) does -- FPFH rows share a strong common shape (each of the three histograms sums to 200 and peaks at its centre bin on
// smooth surfaces), and with it removed the squared norms drop to 0.33 - 0.44 of the raw ones on 5 000-point objects (the synthetic ones
// and the reference's own saved objects alike; 0.7 on sparser clouds), the band by as much.  A fixed table (integers, in matching
// order): any constant is correct, this one is a rounded mean over those objects.
__device__ __constant__ const float FM_MU[33] = {87.f, 46.f, 101.f, 26.f, 28.f, 17.f, 26.f, 26.f, 17.f, 14.f, 21.f, 7.f, 14.f, 19.f, 7.f, 8.f, 14.f,
                                                 6.f,  8.f,  13.f,  6.f,  5.f,  11.f, 6.f,  5.f,  10.f, 6.f,  3.f,  7.f,  14.f, 3.f, 6.f,  14.f};
typedef __attribute__((ext_vector_type(8))) _Float16 fm_piece_t;
// sixteen bytes at a 4-byte aligned address (rows of 33 floats, 36-byte histogram rows): one global_load_dwordx4
struct __attribute__((aligned(4))) ibl_u4_a4 { unsigned x, y, z, w; };
// squared norm of the centred row (fp32 fmaf chain in matching order)
__device__ __forceinline__ float fm_centred_norm(const float* __restrict__ x) {
    float a = 0.0f;
    for (int k = 0; k < 33; ++k) { const float v = x[k] - FM_MU[k]; a = __builtin_fmaf(v, v, a); }
    return a;
}
// 16-byte piece `pc` (0..5) of the operand row [x_0 .. x_32 | 8 8 | nh nl | cu | 0 ...] of a feature row x with centred norm a:
// nh + nl = a / 8 as fp16 hi + lo (against the constant 8 of the other side), cu = 1e-3 a + 4e-3 rounded UP (the bound must not
// shrink; the absolute term covers centred components and norm terms in fp16's subnormal range, rounded to 2^-25 absolute).
// x8: the eight floats x[8 pc .. 8 pc + 7] (only x8[0] is read for pc = 4, none for pc = 5).
__device__ __forceinline__ fm_piece_t fm_operand_piece(const float* x8, float a, int pc) {
    fm_piece_t r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (_Float16)0.0f;
    if (pc < 4) {
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (_Float16)(x8[e] - FM_MU[8 * pc + e]);
    } else if (pc == 4) {
        const float w = a * 0.125f;
        const _Float16 nh = (_Float16)w, nl = (_Float16)(w - (float)nh);
        const float cw = 1.0e-3f * a + 4.0e-3f;
        _Float16 cu = (_Float16)cw;
        if ((float)cu < cw) cu = __builtin_bit_cast(_Float16, (unsigned short)(__builtin_bit_cast(unsigned short, cu) + 1));
        r[0] = (_Float16)(x8[0] - FM_MU[32]);
        r[1] = (_Float16)8.0f; r[2] = (_Float16)8.0f; r[3] = nh; r[4] = nl; r[5] = cu;
    }
    return r;
}
__device__ __forceinline__ int seg_of(const int* __restrict__ seg_off, int n_seg, int i) {
    int lo = 0, hi = n_seg;   // find s with seg_off[s] <= i < seg_off[s+1]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (seg_off[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ int cell_clamp(float v, float mn, float inv, int n) {
    int c = (int)floorf((v - mn) * inv);
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

__device__ __forceinline__ float dist2f(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ void cross3d(const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ double dot3d(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

__device__ __forceinline__ void xform_d(const double* T, double x, double y, double z, double* o) {
    o[0] = T[0] * x + T[1] * y + T[2] * z + T[3];
    o[1] = T[4] * x + T[5] * y + T[6] * z + T[7];
    o[2] = T[8] * x + T[9] * y + T[10] * z + T[11];
}

// Philox4x32-10 (counter-based RNG shared with oracle/oracle_reg.c)
__device__ __forceinline__ void philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                           unsigned* out) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// ---- Kabsch without scaling (Eigen::umeyama semantics): rotation from H = sum (d - dm)(s - sm)^T ----
__device__ inline void jacobi_eig3_d(double A[3][3], double V[3][3], double* w) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (fabs(A[p][q]) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    w[0] = A[0][0]; w[1] = A[1][1]; w[2] = A[2][2];
}

__device__ inline void rotation_from_H_d(double H[3][3], double R[3][3]) {
    double HtH[3][3], V[3][3], w[3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += H[k][i] * H[k][j];
        HtH[i][j] = s;
    }
    jacobi_eig3_d(HtH, V, w);
    int ord[3] = {0, 1, 2};
    for (int a = 0; a < 2; ++a) for (int b = a + 1; b < 3; ++b) if (w[ord[b]] > w[ord[a]]) { int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
    double Vs[3][3], U[3][3], sig[3];
    for (int c = 0; c < 3; ++c) { sig[c] = sqrt(w[ord[c]] > 0 ? w[ord[c]] : 0.0); for (int r = 0; r < 3; ++r) Vs[r][c] = V[r][ord[c]]; }
    {
        double c0[3] = {Vs[0][0], Vs[1][0], Vs[2][0]}, c1[3] = {Vs[0][1], Vs[1][1], Vs[2][1]}, c2[3];
        cross3d(c0, c1, c2);
        Vs[0][2] = c2[0]; Vs[1][2] = c2[1]; Vs[2][2] = c2[2];
    }
    const double tol = 1e-12 * (sig[0] > 0 ? sig[0] : 1.0);
    int rank = 0;
    for (int c = 0; c < 3; ++c) {
        if (sig[c] > tol) {
            for (int r = 0; r < 3; ++r) U[r][c] = (H[r][0] * Vs[0][c] + H[r][1] * Vs[1][c] + H[r][2] * Vs[2][c]) / sig[c];
            ++rank;
        } else break;
    }
    if (rank == 0) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i][j] = i == j ? 1.0 : 0.0; return; }
    double u0[3] = {U[0][0], U[1][0], U[2][0]}, u1[3], u2[3];
    if (rank == 1) {
        double a[3] = {1, 0, 0};
        if (fabs(u0[0]) > 0.9) { a[0] = 0; a[1] = 1; }
        cross3d(u0, a, u1);
        const double l = sqrt(dot3d(u1, u1)); u1[0] /= l; u1[1] /= l; u1[2] /= l;
        cross3d(u0, u1, u2);
    } else {
        u1[0] = U[0][1]; u1[1] = U[1][1]; u1[2] = U[2][1];
        const double l0 = sqrt(dot3d(u0, u0)); u0[0] /= l0; u0[1] /= l0; u0[2] /= l0;
        const double pr = dot3d(u0, u1); u1[0] -= pr * u0[0]; u1[1] -= pr * u0[1]; u1[2] -= pr * u0[2];
        const double l1 = sqrt(dot3d(u1, u1)); u1[0] /= l1; u1[1] /= l1; u1[2] /= l1;
        cross3d(u0, u1, u2);
    }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i][j] = u0[i] * Vs[j][0] + u1[i] * Vs[j][1] + u2[i] * Vs[j][2];
}

// T (row-major 3x4 + [0 0 0 1]) from centroids and H
__device__ inline void kabsch_from_moments(const double* sm, const double* dm, double H[3][3], double* T) {
    double R[3][3];
    rotation_from_H_d(H, R);
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) T[4 * r + c] = R[r][c];
        T[4 * r + 3] = dm[r] - (R[r][0] * sm[0] + R[r][1] * sm[1] + R[r][2] * sm[2]);
    }
    T[12] = T[13] = T[14] = 0.0; T[15] = 1.0;
}
// Workgroup id -> position in a launch whose consecutive positions share data (the points of one cloud, the cells of one region).
// The dispatcher deals workgroups round-robin over the 8 XCDs, each with its own 4 MB L2: taken as is, the ~2 000 workgroups in flight put
// every cloud of that window into every L2 (8 MB of points against 4 MB: the gathers of ibl_normals_from_mask_kernel missed the L2 on
// nearly every access, 2.5 KB of memory-side traffic per point).  Here XCD x walks the x-th contiguous eighth of the launch instead.
__device__ __forceinline__ int ibl_xcd_block(int bid, int nb) {
    const int q = nb >> 3, r = nb & 7, x = bid & 7;
    return x * q + (x < r ? x : r) + (bid >> 3);
}
#ifdef IBL_NO_XCD_REMAP          // lab: the dispatcher's order
#define IBL_XCD_BLOCK(b, nb) ((int)(b))
#else
#define IBL_XCD_BLOCK(b, nb) ibl_xcd_block((int)(b), (int)(nb))
#endif
#endif  // __HIPCC__
