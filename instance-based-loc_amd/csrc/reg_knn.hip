// reg_knn.hip -- hybrid (radius + k-nearest) neighbourhood kernels on the batch grids and their
// fused consumers: normals, SPFH/FPFH, colour gradients, radius-outlier counts.
//
// Replaces Open3D's KDTreeSearchParamHybrid searches inside
//   estimate_normals / compute_fpfh_feature          (utils/fpfh_register.py:90-97)
//   InitializePointCloudForColoredICP                (registration_colored_icp, :132-135)
//   remove_radius_outlier                            (object_memory/object_memory.py:994-995)
//
// One wavefront per query point.  The <= max_nn nearest candidates with d2 < r2 are selected without
// sorting and without storing the candidate list: pass 1 histograms the in-radius candidates over 256
// linear d2 bins (LDS, per wave), the bin holding the k-th neighbour is located with a wave scan,
// pass 2 hands every candidate of a lower bin straight to the consumer and parks only the boundary
// bin (<= 256 entries) in LDS, where the exact (d2, index) order decides the rest.  Candidate rows are
// contiguous runs of the cell-sorted point array, read as coalesced float4.  Memory-bound HIP: no
// MFMA here (normals+FPFH row, HBM roofline).
#pragma clang fp contract(off)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "reg_common.h"

#define KNN_BINS 256
#ifndef KNN_ROWS
#define KNN_ROWS 8        // candidate rows whose first 64 points are in flight together
#endif
#define KNN_CAPB 256
#ifndef KNN_TILE_CAP3
#define KNN_TILE_CAP3 1600     // candidates of a packed tile: 3 x (1 600 x 16 B + tables + 4 x 6.5 KB of wave scratch) fit a CU's 160 KB
#endif
#ifndef KNN_GUESS_SHIFT
#define KNN_GUESS_SHIFT 2      // margin of the threshold-bin guess: + 1 / 4 (12.5 % and 50 % measured 1 % slower)
#endif

// inclusive prefix sum over the 64 lanes on the DPP network (row_shr 1 / 2 / 4 / 8 inside the rows of 16, then row_bcast 15 and 31):
// six VALU instructions, no LDS -- the ds_bpermute form (__shfl_up) cost six LDS round trips, and a query runs three or four scans
__device__ __forceinline__ int wave_incl_scan_i(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

struct WaveLds {
    int hist[KNN_BINS];
    unsigned b_bits[KNN_CAPB];
    int b_idx[KNN_CAPB];
    int b_j[KNN_CAPB];
    int scratch[64];         // [0, 33) SPFH histogram, [62, 64) key range of the index sort
    int rank_pre[64];        // index sort: packed per-word bitmap prefixes
    int sel_j[256];          // consumers that do heavy per-neighbour work first compact the selected set here
    float sel_d2[256];
};

#ifdef KNN_LAB_CLK           // lab builds only: clocks of the phases of a tile's workgroup (thread 0), summed over the tiles
__device__ unsigned long long knn_lab_clk[16];
#define KNN_CLK(k)                                                                            \
    do {                                                                                      \
        if (threadIdx.x == 0) {                                                               \
            const long long _t = (long long)__builtin_amdgcn_s_memtime();                     \
            atomicAdd(&knn_lab_clk[k], (unsigned long long)(_t - _clk_prev));                 \
            _clk_prev = _t;                                                                   \
        }                                                                                     \
    } while (0)
#else
#define KNN_CLK(k)
#endif

// A selected candidate is handed to the consumers as a HANDLE; the accessor turns it into the point and its original index.
//   GlobalAcc: handle = position in the cell-sorted point array (the per-query walk over the grid, hybrid_select)
//   TileAcc:   handle = slot of the workgroup's LDS-staged neighbourhood (tile_select)
struct GlobalAcc {
    const float4* sorted;
    const int* order;
    __device__ __forceinline__ float4 pt(int h) const { return sorted[h]; }
    __device__ __forceinline__ int ord(int h) const { return order[h]; }
};
struct TileAcc {
    const float4* pts;       // LDS
    const int* ordl;         // LDS, or null: the original index rides in the w component of the staged point (packed tiles)
    int key_base, key_span;  // the original indices of the tile's segment lie in [key_base, key_base + key_span)
    __device__ __forceinline__ float4 pt(int h) const { return pts[h]; }
    __device__ __forceinline__ int ord(int h) const { return ordl ? ordl[h] : __float_as_int(pts[h].w); }
};
// key range of the handles a consumer sorts by original index: known up front for a tile (its segment), found by a reduction otherwise
__device__ __forceinline__ bool key_range_hint(const GlobalAcc&, int*, unsigned*) { return false; }
__device__ __forceinline__ bool key_range_hint(const TileAcc& a, int* kmin, unsigned* span) { *kmin = a.key_base; *span = (unsigned)a.key_span; return true; }

// LDS-staged neighbourhood of one tile (a cube of ts^3 cells): every point of the cube of `rho` cells around the tile, copied once
// per workgroup and searched by all the tile's queries
#define KT_ROWS 144          // candidate rows (z, y) of the staging cube: (ts + 2 rho)^2 <= 144
// PACK: the consumer never reads a staged point's intensity (normals, SPFH: geometry only), so the point's original index is staged in
// its w component and the separate index array (4 B per candidate: 10 KB of the tile) goes -- with it and six waves per workgroup the
// 100-neighbour search keeps three waves per SIMD resident instead of two.
template <int KT_CAP, bool PACK = false>        // staged candidates
struct TileLds {
    static constexpr int CAP = KT_CAP;
    static constexpr bool PACKED = PACK;
    float4 pts[KT_CAP];
    int ord[PACK ? 1 : KT_CAP];
    int row_b[KT_ROWS];
    int row_off[KT_ROWS + 1];
    int q_b[16];
    int q_off[17];
    int next_q;              // the next query of the tile nobody has taken yet (tile_knn_block: waves take queries as they finish)
    __device__ __forceinline__ int ord_of(int t) const { return PACK ? __float_as_int(pts[t].w) : ord[t]; }
};

// ------------------------------------------------------------------------------------------------
// generic driver
// ------------------------------------------------------------------------------------------------
template <class Consumer>
__device__ void hybrid_select(const BatchGrid& g, const SegGrid sg, const float4 q, int qi, float radius, float r2, int max_nn,
                              WaveLds* L, Consumer& cons, int* status) {
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    int reach_max = (int)ceilf(radius * sg.inv);
    if (reach_max < 1) reach_max = 1;
    const int cx = cell_clamp(q.x, sg.minx, sg.inv, sg.nx), cy = cell_clamp(q.y, sg.miny, sg.inv, sg.ny),
              cz = cell_clamp(q.z, sg.minz, sg.inv, sg.nz);
    const float bscale = (float)KNN_BINS / r2;

    // The grids are finer than the search radius (cell = radius / 3 .. radius / 5, reg_api.hip): with max_nn = 100 neighbours inside
    // r = 0.25 m a dense cloud (5 000 points on a 0.5 m object) has ALL its points in radius, and scanning them twice per query
    // was 90 % of the feature time.  The max_nn nearest are searched in the cube of `reach` cells around the query's cell first:
    // that cube holds every point closer than reach * cell, so once max_nn candidates lie inside that ball the cube's k nearest
    // are the cloud's k nearest (anything outside the cube is farther than all of them), and the result is the one of the full
    // radius, bit for bit.  reach starts at the first cube whose population (read from the cell table, no distances) promises
    // enough candidates and doubles when the ball count falls short.
    int reach = reach_max;
    for (int rho = 1; rho < reach_max; ++rho) {
        const int xa = max(cx - rho, 0), xb = min(cx + rho, sg.nx - 1);
        const int ya = max(cy - rho, 0), yb = min(cy + rho, sg.ny - 1);
        const int za = max(cz - rho, 0), zb = min(cz + rho, sg.nz - 1);
        const int nyr = yb - ya + 1, nr = nyr * (zb - za + 1);
        int pop = 0;
        for (int r = lane; r < nr; r += 64) {
            const int row = sg.cell_base + ((za + r / nyr) * sg.ny + ya + r % nyr) * sg.nx;
            pop += g.cell_start[row + xb + 1] - g.cell_start[row + xa];
        }
        pop = wave_sum_i(pop);
        // a ball of radius rho * cell covers ~1/3 (rho = 1) .. ~1/2 of the surface patch its cube cuts out
        if (pop >= (rho == 1 ? 3 : 2) * max_nn + 8) { reach = rho; break; }
    }
    int x0, x1, y0, y1, z0, z1, ny, nrows, my_b, my_e;
    auto set_reach = [&](int rho) {
        x0 = max(cx - rho, 0); x1 = min(cx + rho, sg.nx - 1);
        y0 = max(cy - rho, 0); y1 = min(cy + rho, sg.ny - 1);
        z0 = max(cz - rho, 0); z1 = min(cz + rho, sg.nz - 1);
        ny = y1 - y0 + 1; nrows = ny * (z1 - z0 + 1);
        my_b = my_e = 0;
        if (lane < nrows) {             // bounds of the first 64 candidate rows (runs of the cell-sorted point array, one per (z, y))
            const int row = sg.cell_base + ((z0 + lane / ny) * sg.ny + y0 + lane % ny) * sg.nx;
            my_b = g.cell_start[row + x0];
            my_e = g.cell_start[row + x1 + 1];
        }
    };
    // A wave's walk is a chain of dependent latencies (row bounds -> points -> next row), and these kernels are bound by it, so the
    // bounds of 64 rows are fetched in one step (lane r holds row r), empty rows are dropped by ballot, and the first 64 points of
    // KNN_ROWS rows at a time are in flight together.
    auto scan = [&](auto&& f) {
        for (int rc = 0; rc < nrows; rc += 64) {
            int cb = my_b, ce = my_e;
            if (rc > 0) {
                cb = ce = 0;
                const int r = rc + lane;
                if (r < nrows) {
                    const int row = sg.cell_base + ((z0 + r / ny) * sg.ny + y0 + r % ny) * sg.nx;
                    cb = g.cell_start[row + x0];
                    ce = g.cell_start[row + x1 + 1];
                }
            }
            unsigned long long live = __ballot(ce > cb);
            while (live) {
                int b[KNN_ROWS], e[KNN_ROWS];
                float4 p[KNN_ROWS];
#pragma unroll
                for (int u = 0; u < KNN_ROWS; ++u) {
                    const bool have = live != 0ull;
                    const int r = have ? __ffsll((long long)live) - 1 : 0;
                    if (have) live &= live - 1ull;
                    b[u] = __shfl(cb, r, 64);
                    e[u] = have ? __shfl(ce, r, 64) : b[u];
                    p[u] = b[u] + lane < e[u] ? g.sorted_pts[b[u] + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < KNN_ROWS; ++u) {
                    for (int jb = b[u]; jb < e[u]; jb += 64) {
                        const int j = jb + lane;
                        float4 q4 = p[u];
                        if (jb != b[u]) q4 = j < e[u] ? g.sorted_pts[j] : make_float4(0.f, 0.f, 0.f, 0.f);
                        const float d2 = j < e[u] ? dist2f(q.x, q.y, q.z, q4.x, q4.y, q4.z) : INFINITY;
                        f(d2 < r2, j, q4, d2);
                    }
                }
            }
        }
    };
    auto bin_of = [&](float d2) { int b = (int)(d2 * bscale); return b > KNN_BINS - 1 ? KNN_BINS - 1 : b; };

    // ---- pass 1: histogram (repeated on a larger cube while the ball inside the cube holds fewer than max_nn candidates) ------
    for (;;) {
        set_reach(reach);
#pragma unroll
        for (int t = 0; t < KNN_BINS / 64; ++t) L->hist[lane * (KNN_BINS / 64) + t] = 0;
        wave_lds_sync();
        // every point closer than (reach - 0.001) cells is inside the cube (the margin covers the rounding of the cell index)
        const float ball = ((float)reach - 1e-3f) / sg.inv;
        const float ball2 = reach < reach_max ? ball * ball : INFINITY;
        int in_ball = 0;
        scan([&](bool in, int, const float4&, float d2) {
            if (in) atomicAdd(&L->hist[bin_of(d2)], 1);
            in_ball += (in && d2 < ball2) ? 1 : 0;
        });
        wave_lds_sync();
        if (reach >= reach_max) break;
        if (wave_sum_i(in_ball) >= max_nn) break;
        reach = min(reach_max, 2 * reach);
        wave_lds_sync();
    }
    int hb[KNN_BINS / 64];
    int s = 0;
#pragma unroll
    for (int t = 0; t < KNN_BINS / 64; ++t) { hb[t] = L->hist[lane * (KNN_BINS / 64) + t]; s += hb[t]; }
    int incl = wave_incl_scan_i(s);
    const int cnt = __shfl(incl, 63, 64);
    const int excl = incl - s;
    const int k = cnt < max_nn ? cnt : max_nn;
    bool select_all = cnt <= max_nn;
    int bstar = KNN_BINS, n_below = 0, pop = 0;
    if (!select_all) {
        const unsigned long long m = __ballot(incl >= max_nn);
        const int Lc = __ffsll((long long)m) - 1;
        int bin_here = 0, my_below = 0, my_pop = 0;
        {
            int run = excl;
#pragma unroll
            for (int t = 0; t < KNN_BINS / 64; ++t) {
                if (my_pop == 0 && run + hb[t] >= max_nn) { bin_here = lane * (KNN_BINS / 64) + t; my_below = run; my_pop = hb[t]; }
                run += hb[t];
            }
        }
        bstar = __shfl(bin_here, Lc, 64);
        n_below = __shfl(my_below, Lc, 64);
        pop = __shfl(my_pop, Lc, 64);
    }
    cons.begin(k);

    // ---- pass 2: emit lower bins, park the boundary bin ---------------------------------------
    const bool fast = pop <= KNN_CAPB;
    int bcount = 0;
    scan([&](bool in, int j, const float4& p, float d2) {
        const int b = in ? bin_of(d2) : KNN_BINS;
        cons.accept(in && (select_all || b < bstar), j, p, d2);
        if (!select_all && fast) {
            const bool park = in && b == bstar;
            const unsigned long long m = __ballot(park);
            if (park) {
                const int pos = bcount + __popcll(m & lt_mask);
                L->b_bits[pos] = __float_as_uint(d2);
                L->b_idx[pos] = g.order[j];
                L->b_j[pos] = j;
            }
            bcount += __popcll(m);
        }
    });
    if (!select_all) {
        const int need = max_nn - n_below;
        if (fast) {
            wave_lds_sync();
            unsigned eb[KNN_CAPB / 64];
            int ei[KNN_CAPB / 64];
            bool ev[KNN_CAPB / 64];
#pragma unroll
            for (int t = 0; t < KNN_CAPB / 64; ++t) {
                const int e = lane + 64 * t;
                ev[t] = e < pop;
                eb[t] = ev[t] ? L->b_bits[e] : 0xFFFFFFFFu;
                ei[t] = ev[t] ? L->b_idx[e] : 0x7FFFFFFF;
            }
            unsigned lo = 0, hi = 0x7F800000u;
            while (lo < hi) {
                const unsigned mid = lo + ((hi - lo) >> 1);
                int c = 0;
#pragma unroll
                for (int t = 0; t < KNN_CAPB / 64; ++t) c += __popcll(__ballot(ev[t] && eb[t] <= mid));
                if (c >= need) hi = mid; else lo = mid + 1;
            }
            const unsigned T = lo;
            int cl = 0, ct = 0;
#pragma unroll
            for (int t = 0; t < KNN_CAPB / 64; ++t) { cl += __popcll(__ballot(ev[t] && eb[t] < T)); ct += __popcll(__ballot(ev[t] && eb[t] == T)); }
            const int need2 = need - cl;
            int I = 0x7FFFFFFF;
            if (ct > need2) {
                int ilo = 0, ihi = 0x7FFFFFFF;
                while (ilo < ihi) {
                    const int mid = ilo + ((ihi - ilo) >> 1);
                    int c = 0;
#pragma unroll
                    for (int t = 0; t < KNN_CAPB / 64; ++t) c += __popcll(__ballot(ev[t] && eb[t] == T && ei[t] <= mid));
                    if (c >= need2) ihi = mid; else ilo = mid + 1;
                }
                I = ilo;
            }
#pragma unroll
            for (int t = 0; t < KNN_CAPB / 64; ++t) {
                const bool sel = ev[t] && (eb[t] < T || (eb[t] == T && ei[t] <= I));
                int j = 0;
                float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
                if (sel) { j = L->b_j[lane + 64 * t]; p = g.sorted_pts[j]; }
                cons.accept(sel, j, p, __uint_as_float(eb[t]));
            }
        } else {
            // boundary bin larger than the LDS list (many near-equidistant candidates): exact thresholds by
            // re-scanning the candidates -- slow, correct, flagged in the status word
            if (lane == 0) atomicOr(status, IBL_ST_KNN_SLOWPATH);
            unsigned lo = 0, hi = 0x7F800000u;
            while (lo < hi) {
                const unsigned mid = lo + ((hi - lo) >> 1);
                int c = 0;
                scan([&](bool in, int, const float4&, float d2) {
                    c += (in && bin_of(d2) == bstar && __float_as_uint(d2) <= mid) ? 1 : 0;
                });
                c = wave_sum_i(c);
                if (c >= need) hi = mid; else lo = mid + 1;
            }
            const unsigned T = lo;
            int cl = 0, ct = 0;
            scan([&](bool in, int, const float4&, float d2) {
                const bool bb = in && bin_of(d2) == bstar;
                cl += (bb && __float_as_uint(d2) < T) ? 1 : 0;
                ct += (bb && __float_as_uint(d2) == T) ? 1 : 0;
            });
            cl = wave_sum_i(cl); ct = wave_sum_i(ct);
            const int need2 = need - cl;
            int I = 0x7FFFFFFF;
            if (ct > need2) {
                int ilo = 0, ihi = 0x7FFFFFFF;
                while (ilo < ihi) {
                    const int mid = ilo + ((ihi - ilo) >> 1);
                    int c = 0;
                    scan([&](bool in, int j, const float4&, float d2) {
                        if (in && bin_of(d2) == bstar && __float_as_uint(d2) == T) c += (g.order[j] <= mid) ? 1 : 0;
                    });
                    c = wave_sum_i(c);
                    if (c >= need2) ihi = mid; else ilo = mid + 1;
                }
                I = ilo;
            }
            scan([&](bool in, int j, const float4& p, float d2) {
                bool sel = false;
                if (in && bin_of(d2) == bstar) {
                    const unsigned bits = __float_as_uint(d2);
                    sel = bits < T || (bits == T && g.order[j] <= I);
                }
                cons.accept(sel, j, p, d2);
            });
        }
    }
    cons.finish(k);
}

// ------------------------------------------------------------------------------------------------
// robust symmetric 3x3 eigen solver (Open3D FastEigen3x3 / Eberly): smallest-eigenvalue eigenvector
// ------------------------------------------------------------------------------------------------
__device__ inline void eigenvector0_d(const double* A, double ev, double* out) {
    double r0[3] = {A[0] - ev, A[1], A[2]}, r1[3] = {A[1], A[3] - ev, A[4]}, r2[3] = {A[2], A[4], A[5] - ev};
    double c01[3], c02[3], c12[3];
    cross3d(r0, r1, c01); cross3d(r0, r2, c02); cross3d(r1, r2, c12);
    const double d0 = dot3d(c01, c01), d1 = dot3d(c02, c02), d2 = dot3d(c12, c12);
    double dmax = d0; int imax = 0;
    if (d1 > dmax) { dmax = d1; imax = 1; }
    if (d2 > dmax) { imax = 2; }
    const double* c = imax == 0 ? c01 : (imax == 1 ? c02 : c12);
    const double d = imax == 0 ? d0 : (imax == 1 ? d1 : d2);
    const double s = sqrt(d);
    out[0] = c[0] / s; out[1] = c[1] / s; out[2] = c[2] / s;
}

__device__ inline void eigenvector1_d(const double* A, const double* e0, double ev1, double* out) {
    double U[3], V[3];
    if (fabs(e0[0]) > fabs(e0[1])) {
        const double inv = 1.0 / sqrt(e0[0] * e0[0] + e0[2] * e0[2]);
        U[0] = -e0[2] * inv; U[1] = 0; U[2] = e0[0] * inv;
    } else {
        const double inv = 1.0 / sqrt(e0[1] * e0[1] + e0[2] * e0[2]);
        U[0] = 0; U[1] = e0[2] * inv; U[2] = -e0[1] * inv;
    }
    cross3d(e0, U, V);
    double AU[3] = {A[0] * U[0] + A[1] * U[1] + A[2] * U[2], A[1] * U[0] + A[3] * U[1] + A[4] * U[2],
                    A[2] * U[0] + A[4] * U[1] + A[5] * U[2]};
    double AV[3] = {A[0] * V[0] + A[1] * V[1] + A[2] * V[2], A[1] * V[0] + A[3] * V[1] + A[4] * V[2],
                    A[2] * V[0] + A[4] * V[1] + A[5] * V[2]};
    double m00 = dot3d(U, AU) - ev1, m01 = dot3d(U, AV), m11 = dot3d(V, AV) - ev1;
    const double a00 = fabs(m00), a01 = fabs(m01), a11 = fabs(m11);
    if (a00 >= a11) {
        const double mx = a00 > a01 ? a00 : a01;
        if (mx > 0) {
            if (a00 >= a01) { m01 /= m00; m00 = 1 / sqrt(1 + m01 * m01); m01 *= m00; }
            else { m00 /= m01; m01 = 1 / sqrt(1 + m00 * m00); m00 *= m01; }
            for (int i = 0; i < 3; ++i) out[i] = m01 * U[i] - m00 * V[i];
        } else { out[0] = U[0]; out[1] = U[1]; out[2] = U[2]; }
    } else {
        const double mx = a11 > a01 ? a11 : a01;
        if (mx > 0) {
            if (a11 >= a01) { m01 /= m11; m11 = 1 / sqrt(1 + m01 * m01); m01 *= m11; }
            else { m11 /= m01; m01 = 1 / sqrt(1 + m11 * m11); m11 *= m01; }
            for (int i = 0; i < 3; ++i) out[i] = m11 * U[i] - m01 * V[i];
        } else { out[0] = U[0]; out[1] = U[1]; out[2] = U[2]; }
    }
}

__device__ inline void fast_eigen_normal_d(const double* cov, double* n) {
    double mc = cov[0];
    for (int i = 1; i < 6; ++i) if (cov[i] > mc) mc = cov[i];
    if (mc == 0) { n[0] = n[1] = n[2] = 0; return; }
    double A[6];
    for (int i = 0; i < 6; ++i) A[i] = cov[i] / mc;
    const double norm = A[1] * A[1] + A[2] * A[2] + A[4] * A[4];
    if (norm > 0) {
        const double q = (A[0] + A[3] + A[5]) / 3;
        const double b00 = A[0] - q, b11 = A[3] - q, b22 = A[5] - q;
        const double p = sqrt((b00 * b00 + b11 * b11 + b22 * b22 + norm * 2) / 6);
        const double c00 = b11 * b22 - A[4] * A[4];
        const double c01 = A[1] * b22 - A[4] * A[2];
        const double c02 = A[1] * A[4] - b11 * A[2];
        const double det = (b00 * c00 - A[1] * c01 + A[2] * c02) / (p * p * p);
        double half_det = det * 0.5;
        if (half_det < -1.0) half_det = -1.0;
        if (half_det > 1.0) half_det = 1.0;
        const double angle = acos(half_det) / 3.0;
        const double two_thirds_pi = 2.09439510239319549;
        const double beta2 = cos(angle) * 2;
        const double beta0 = cos(angle + two_thirds_pi) * 2;
        const double beta1 = -(beta0 + beta2);
        const double e0 = q + p * beta0, e1 = q + p * beta1, e2 = q + p * beta2;
        double v0[3], v1[3], v2[3];
        if (half_det >= 0) {
            eigenvector0_d(A, e2, v2);
            if (e2 < e0 && e2 < e1) { n[0] = v2[0]; n[1] = v2[1]; n[2] = v2[2]; return; }
            eigenvector1_d(A, v2, e1, v1);
            if (e1 < e0 && e1 < e2) { n[0] = v1[0]; n[1] = v1[1]; n[2] = v1[2]; return; }
            cross3d(v1, v2, v0);
            n[0] = v0[0]; n[1] = v0[1]; n[2] = v0[2];
        } else {
            eigenvector0_d(A, e0, v0);
            if (e0 < e1 && e0 < e2) { n[0] = v0[0]; n[1] = v0[1]; n[2] = v0[2]; return; }
            eigenvector1_d(A, v0, e1, v1);
            if (e1 < e0 && e1 < e2) { n[0] = v1[0]; n[1] = v1[1]; n[2] = v1[2]; return; }
            cross3d(v0, v1, v2);
            n[0] = v2[0]; n[1] = v2[1]; n[2] = v2[2];
        }
    } else {
        if (cov[0] < cov[3] && cov[0] < cov[5]) { n[0] = 1; n[1] = 0; n[2] = 0; }
        else if (cov[3] < cov[0] && cov[3] < cov[5]) { n[0] = 0; n[1] = 1; n[2] = 0; }
        else { n[0] = 0; n[1] = 0; n[2] = 1; }
    }
}

// ------------------------------------------------------------------------------------------------
// consumers
// ------------------------------------------------------------------------------------------------
// The selected neighbours arrive in the enumeration order of the grid.  Consumers reduce them in ascending ORIGINAL
// point index instead (rank sort of the <= 256 list entries in LDS), so that every floating-point sum is independent
// of the grid a cloud was enumerated from: features of an instance computed on its own (ibl_instance_features_batch)
// are bit-identical to those computed inside a concatenation whose other instances are out of reach.
// in: L->sel_j[0..k) (+ sel_d2); out: L->b_j[0..k) (+ b_bits = d2 bits) in ascending order[j].
template <bool WITH_D2, class Acc>
__device__ __forceinline__ void sort_selected_by_index(WaveLds* L, const Acc& acc, int k) {
    const int lane = threadIdx.x & 63;
    int kmin;
    unsigned span;
    if (key_range_hint(acc, &kmin, &span) && span <= 32u * KNN_BINS) {
        // a tile: every key lies in its segment's index range -- no reduction (64 lanes' atomicMin / atomicMax on one LDS word serialised)
        for (int t = lane; t < k; t += 64) L->b_idx[t] = acc.ord(L->sel_j[t]);
        if (span > 0) --span;
        wave_lds_sync();
    } else {
        int mn = 0x7FFFFFFF, mx = -1;
        for (int t = lane; t < k; t += 64) {
            const int key = acc.ord(L->sel_j[t]);
            L->b_idx[t] = key;
            mn = min(mn, key);
            mx = max(mx, key);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { mn = min(mn, __shfl_xor(mn, off, 64)); mx = max(mx, __shfl_xor(mx, off, 64)); }
        kmin = mn;
        span = (unsigned)(mx - mn);
        wave_lds_sync();
    }
    if (span < 32u * KNN_BINS) {
        // the keys are distinct original indices: rank = number of set bits below the key's bit in a bitmap of the span
        // (L->hist is free once the selection is over); per-word exclusive prefixes fit a byte (rank < k <= 256)
        unsigned* bits = reinterpret_cast<unsigned*>(L->hist);
        *reinterpret_cast<uint4*>(&bits[4 * lane]) = make_uint4(0u, 0u, 0u, 0u);
        wave_lds_sync();
        for (int t = lane; t < k; t += 64) {
            const unsigned o = (unsigned)(L->b_idx[t] - kmin);
            atomicOr(&bits[o >> 5], 1u << (o & 31u));
        }
        wave_lds_sync();
        const uint4 w = *reinterpret_cast<const uint4*>(&bits[4 * lane]);
        const int c0 = __popc(w.x), c1 = __popc(w.y), c2 = __popc(w.z), c3 = __popc(w.w);
        const int s = c0 + c1 + c2 + c3;
        int incl = wave_incl_scan_i(s);
        const unsigned e0 = (unsigned)(incl - s), e1 = e0 + c0, e2 = e1 + c1, e3 = e2 + c2;
        reinterpret_cast<unsigned*>(L->rank_pre)[lane] = (e0 & 255u) | ((e1 & 255u) << 8) | ((e2 & 255u) << 16) | ((e3 & 255u) << 24);
        wave_lds_sync();
        const unsigned char* pre = reinterpret_cast<const unsigned char*>(L->rank_pre);
        for (int t = lane; t < k; t += 64) {
            const unsigned o = (unsigned)(L->b_idx[t] - kmin);
            const int r = (int)pre[o >> 5] + __popc(bits[o >> 5] & ((1u << (o & 31u)) - 1u));
            L->b_j[r] = L->sel_j[t];
            if (WITH_D2) L->b_bits[r] = __float_as_uint(L->sel_d2[t]);
        }
    } else {
        for (int t = lane; t < k; t += 64) {
            const int key = L->b_idx[t];
            int r = 0;
            for (int u = 0; u < k; ++u) r += L->b_idx[u] < key ? 1 : 0;
            L->b_j[r] = L->sel_j[t];
            if (WITH_D2) L->b_bits[r] = __float_as_uint(L->sel_d2[t]);
        }
    }
    wave_lds_sync();
}
// Normals are solved in batches: a wave parks the sorted neighbour list of each query (<= NP_MAXK handles) in NormalPending and, once
// NP_SLOTS queries are parked (or its tile is finished), every lane takes one query: covariance sums in ascending original index, then
// the 3x3 eigen solve (~500 fp64 instructions that used to run on lane 0 alone, once per query, plus nine 64-lane fp64 reductions).
// The sums run in the same order on every path (tile, list, query kernel), so a point's normal does not depend on which one served it.
#define NP_SLOTS 16
#define NP_MAXK 32
template <int SLOTS_>
struct NormalPendingT {
    static constexpr int SLOTS = SLOTS_;
    unsigned short h[SLOTS_][NP_MAXK];
    int qi[SLOTS_];
    int k[SLOTS_];
    int n;
};
typedef NormalPendingT<NP_SLOTS> NormalPending;
struct NoPending { int n; };

// the nine moment sums of a normal's neighbours (ascending original index) -> covariance -> smallest eigenvector
__device__ __forceinline__ void normal_from_moments(double* c, int k, int qi, float4* __restrict__ normals);

template <class Acc, class H>
__device__ __forceinline__ void normal_solve(const Acc& acc, const H* __restrict__ hnd, int k, int qi, float4* __restrict__ normals) {
    double c[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) c[t] = 0.0;
    for (int t = 0; t < k; ++t) {
        const float4 p = acc.pt((int)hnd[t]);
        const double x = p.x, y = p.y, z = p.z;
        c[0] += x; c[1] += y; c[2] += z;
        c[3] += x * x; c[4] += x * y; c[5] += x * z; c[6] += y * y; c[7] += y * z; c[8] += z * z;
    }
    normal_from_moments(c, k, qi, normals);
}

__device__ __forceinline__ void normal_from_moments(double* c, int k, int qi, float4* __restrict__ normals) {
    double n[3];
    if (k >= 3) {
#pragma unroll
        for (int t = 0; t < 9; ++t) c[t] /= (double)k;
        double cov[6] = {c[3] - c[0] * c[0], c[4] - c[0] * c[1], c[5] - c[0] * c[2],
                         c[6] - c[1] * c[1], c[7] - c[1] * c[2], c[8] - c[2] * c[2]};
        fast_eigen_normal_d(cov, n);
    } else {
        double cov[6] = {1, 0, 0, 1, 0, 1};
        fast_eigen_normal_d(cov, n);
    }
    if (sqrt(dot3d(n, n)) == 0.0) { n[0] = 0; n[1] = 0; n[2] = 1; }
    normals[qi] = make_float4((float)n[0], (float)n[1], (float)n[2], 0.0f);
}

template <class Acc, class PT>
__device__ __forceinline__ void normal_flush(PT* P, const Acc& acc, float4* __restrict__ normals) {
    const int lane = threadIdx.x & 63;
    wave_lds_sync();
    const int n = P->n;
    if (lane < n) normal_solve(acc, P->h[lane], P->k[lane], P->qi[lane], normals);
    wave_lds_sync();
    if (lane == 0) P->n = 0;
    wave_lds_sync();
}

// End of a tile: the queries still parked by the four waves (two or three each -- a tile holds ~10 queries) are solved together by
// wave 0, one per lane, instead of once per wave with two or three lanes active.
template <class Acc, class PT>
__device__ __forceinline__ void normal_flush_block(PT* pend, const Acc& acc, float4* __restrict__ normals) {
    __syncthreads();
    if ((threadIdx.x >> 6) == 0) {
        const int lane = threadIdx.x & 63;
        const int n0 = pend[0].n, n1 = pend[1].n, n2 = pend[2].n, n3 = pend[3].n;
        if (lane < n0 + n1 + n2 + n3) {
            const int w = lane < n0 ? 0 : (lane < n0 + n1 ? 1 : (lane < n0 + n1 + n2 ? 2 : 3));
            const int sl = lane - (w > 0 ? n0 : 0) - (w > 1 ? n1 : 0) - (w > 2 ? n2 : 0);
            normal_solve(acc, pend[w].h[sl], pend[w].k[sl], pend[w].qi[sl], normals);
        }
    }
}

template <class Acc>
struct NormalConsumer {
    static constexpr bool WANTS_INNER = false;
    float4* normals;
    Acc acc;
    WaveLds* L;
    NormalPending* P;        // null: solve at once (list / query kernels, whose handles are 32-bit)
    int qi;
    int ncount;
    __device__ void begin(int) { ncount = 0; }
    __device__ void accept(bool sel, int j, const float4&, float) {
        const int lane = threadIdx.x & 63;
        const unsigned long long m = __ballot(sel);
        if (sel) L->sel_j[ncount + __popcll(m & ((1ull << lane) - 1ull))] = j;
        ncount += __popcll(m);
    }
    __device__ void finish(int k) {
        const int lane = threadIdx.x & 63;
        sort_selected_by_index<false>(L, acc, k);
        if (P != nullptr && k <= NP_MAXK) {
            const int slot = P->n;
            if (lane < k) P->h[slot][lane] = (unsigned short)L->b_j[lane];
            if (lane == 0) { P->qi[slot] = qi; P->k[slot] = k; P->n = slot + 1; }
            if (slot + 1 == NP_SLOTS) normal_flush(P, acc, normals);
            else wave_lds_sync();
        } else if (lane == 0) {
            normal_solve(acc, L->b_j, k, qi, normals);
        }
    }
};

__device__ inline void pair_features_d(const float4& p1, const float4& n1f, const float4& p2, const float4& n2f, double* f) {
    double d[3] = {(double)p2.x - (double)p1.x, (double)p2.y - (double)p1.y, (double)p2.z - (double)p1.z};
    double n1[3] = {n1f.x, n1f.y, n1f.z}, n2[3] = {n2f.x, n2f.y, n2f.z};
    const double r = sqrt(dot3d(d, d));
    f[0] = 5.0; f[1] = f[2] = 0;          // (all-zero features: theta = 0 lies in bin 5; f[0] is a bin index, see below)
    if (r == 0.0) return;
    const double a1 = dot3d(n1, d) / r, a2 = dot3d(n2, d) / r;
    double na[3], nb[3];
    // acos(|a1|) > acos(|a2|)  <=>  |a1| < |a2|  (false when either exceeds 1: acos -> NaN)
    if (fabs(a1) < fabs(a2) && fabs(a1) <= 1.0 && fabs(a2) <= 1.0) {
        for (int t = 0; t < 3; ++t) { na[t] = n2[t]; nb[t] = n1[t]; d[t] = -d[t]; }
        f[2] = -a2;
    } else {
        for (int t = 0; t < 3; ++t) { na[t] = n1[t]; nb[t] = n2[t]; }
        f[2] = a1;
    }
    double v[3], w[3];
    cross3d(d, na, v);
    const double vn = sqrt(dot3d(v, v));
    if (vn == 0.0) { f[0] = 5.0; f[1] = f[2] = 0; return; }
    v[0] /= vn; v[1] /= vn; v[2] /= vn;
    cross3d(na, v, w);
    f[1] = dot3d(v, nb);
    // f[0] carries the BIN of theta = atan2(w.nb, na.nb), floor(11 (theta + pi) / (2 pi)), found without the arctangent (~120 fp64
    // instructions): theta' = theta + pi is the angle of p = (-x, -y), and within a half plane "theta' >= 2 pi k / 11" is the sign of the
    // cross product with the boundary direction -- five tests.  The half plane follows atan2's sign-of-zero rule (y = +0, x < 0 is
    // +pi -> last bin; y = -0 is -pi -> bin 0: antiparallel normals on a plane land there).  It can differ from the arctangent only
    // for an angle within rounding of a boundary.
    {
        const double x = dot3d(na, nb), y = dot3d(w, nb);
        constexpr double CK[10] = {0.84125353283118121, 0.41541501300188644, -0.142314838273285, -0.65486073394528499, -0.95949297361449737, -0.95949297361449748, -0.65486073394528521, -0.14231483827328523, 0.41541501300188605, 0.84125353283118121};
        constexpr double SK[10] = {0.54064081745559756, 0.90963199535451833, 0.9898214418809328, 0.75574957435425827, 0.28173255684142967, -0.28173255684142939, -0.75574957435425816, -0.98982144188093268, -0.90963199535451855, -0.54064081745559744};
        const double px = -x, py = -y;
        int bin;
        if (!(__double_as_longlong(py) < 0)) {            // py = +0 or positive: theta' in [0, pi]
            bin = 0;
#pragma unroll
            for (int k = 0; k < 5; ++k) bin += (CK[k] * py - SK[k] * px >= 0.0) ? 1 : 0;
        } else {                                           // theta' in (pi, 2 pi]
            bin = 5;
#pragma unroll
            for (int k = 5; k < 10; ++k) bin += (CK[k] * py - SK[k] * px >= 0.0) ? 1 : 0;
        }
        if (x == 0.0 && y == 0.0) bin = (__double_as_longlong(x) < 0) ? ((__double_as_longlong(y) < 0) ? 0 : 10) : 5;   // atan2(+-0, +-0)
        f[0] = (double)bin;
    }
}

__device__ __forceinline__ int clamp_bin11(int h) { return h < 0 ? 0 : (h >= 11 ? 10 : h); }

// The three SPFH bins of a pair in fp32, or "undecided" (round 4).  pair_features_d spends ~150 fp64 operations (two square roots, five
// divisions) on a pair whose only output is three bin INDICES: the fp64 values matter where they sit next to a bin boundary, nowhere
// else.  This evaluates the same expressions in fp32 and calls a pair decided only when every decision on the way -- which normal carries
// the frame (|a1| < |a2|), the half plane and the five boundary tests of theta, the bins of v . nb and of a -- clears a guard band at least
// ten times the fp32 error of the value tested (u = 2^-24; inputs are fp32, normals unit to 1 ulp):
//   d = p2 - p1: one rounding (relative u).  a = (n . d) rsq(|d|^2): absolute error <= 8e-7 -> band 1e-5 on |a1| - |a2| and on 1 - |a|,
//     5e-5 on the bin fraction of a.
//   v = d x na: component error <= 6 u |d|; pairs with |v|^2 < 1e-2 |d|^2 (d within 5.7 degrees of the frame's normal) are undecided, for
//     the rest the normalised v is within 1.3e-5 of the fp64 one, and so are v . nb, w = na x v, y = w . nb and the boundary forms
//     CK y - SK x -> band SPFH_G = 2e-4 on y and the five forms, 11 / 2 SPFH_G on the bin fraction of v . nb.
// Undecided pairs (measured: see DESIGN.md) are evaluated by pair_features_d afterwards, so the histograms are those of the fp64 evaluation bit for bit
// (IBL_SPFH_F64=1 runs every pair in fp64: tests/test_gpu_features.py compares the two).
#define SPFH_G 2.0e-4f
__device__ __forceinline__ bool pair_bins_f32(const float4& p1, const float4& n1, const float4& p2, const float4& n2, int* b0, int* b1, int* b2) {
    float dx = p2.x - p1.x, dy = p2.y - p1.y, dz = p2.z - p1.z;
    *b0 = 5; *b1 = 5; *b2 = 5;
    if (dx == 0.0f && dy == 0.0f && dz == 0.0f) return true;                  // coincident points: the zero features' bins (d is exact there)
    const float r2 = dx * dx + dy * dy + dz * dz;
    if (r2 < 1.0e-20f) return false;
    const float rinv = __builtin_amdgcn_rsqf(r2);
    const float a1 = (n1.x * dx + n1.y * dy + n1.z * dz) * rinv, a2 = (n2.x * dx + n2.y * dy + n2.z * dz) * rinv;
    const float f1a = fabsf(a1), f2a = fabsf(a2);
    const bool same_n = n1.x == n2.x && n1.y == n2.y && n1.z == n2.z;          // (planes: a1 == a2 in fp64 as well -> no swap)
    bool sure = (same_n || fabsf(f1a - f2a) > 1.0e-5f) && fmaxf(f1a, f2a) < 1.0f - 1.0e-5f;
    const bool swap = f1a < f2a;
    float nax, nay, naz, nbx, nby, nbz, f3;
    if (swap) { nax = n2.x; nay = n2.y; naz = n2.z; nbx = n1.x; nby = n1.y; nbz = n1.z; dx = -dx; dy = -dy; dz = -dz; f3 = -a2; }
    else { nax = n1.x; nay = n1.y; naz = n1.z; nbx = n2.x; nby = n2.y; nbz = n2.z; f3 = a1; }
    float vx = dy * naz - dz * nay, vy = dz * nax - dx * naz, vz = dx * nay - dy * nax;
    const float vn2 = vx * vx + vy * vy + vz * vz;
    sure = sure && vn2 > 1.0e-2f * r2;                                         // sin^2 of the angle between d and the frame's normal
    const float vinv = __builtin_amdgcn_rsqf(fmaxf(vn2, 1e-30f));
    vx *= vinv; vy *= vinv; vz *= vinv;
    const float f2 = vx * nbx + vy * nby + vz * nbz;
    const float wx = nay * vz - naz * vy, wy = naz * vx - nax * vz, wz = nax * vy - nay * vx;
    const float x = nax * nbx + nay * nby + naz * nbz, y = wx * nbx + wy * nby + wz * nbz;
    const float px = -x, py = -y;
    sure = sure && fabsf(py) > SPFH_G;
    constexpr float CK[10] = {0.84125353283118121f, 0.41541501300188644f, -0.142314838273285f, -0.65486073394528499f, -0.95949297361449737f,
                              -0.95949297361449748f, -0.65486073394528521f, -0.14231483827328523f, 0.41541501300188605f, 0.84125353283118121f};
    constexpr float SK[10] = {0.54064081745559756f, 0.90963199535451833f, 0.9898214418809328f, 0.75574957435425827f, 0.28173255684142967f,
                              -0.28173255684142939f, -0.75574957435425816f, -0.98982144188093268f, -0.90963199535451855f, -0.54064081745559744f};
    int bin = py >= 0.0f ? 0 : 5;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float t = py >= 0.0f ? CK[k] * py - SK[k] * px : CK[k + 5] * py - SK[k + 5] * px;
        sure = sure && fabsf(t) > SPFH_G;
        bin += t >= 0.0f ? 1 : 0;
    }
    const float g1 = 11.0f * (f2 + 1.0f) * 0.5f, g2 = 11.0f * (f3 + 1.0f) * 0.5f;
    const float fl1 = floorf(g1), fl2 = floorf(g2);
    constexpr float GB1 = 5.5f * SPFH_G, GB2 = 5.0e-5f;
    sure = sure && (g1 - fl1) > GB1 && (g1 - fl1) < 1.0f - GB1 && (g2 - fl2) > GB2 && (g2 - fl2) < 1.0f - GB2;
    *b0 = bin; *b1 = clamp_bin11((int)fl1); *b2 = clamp_bin11((int)fl2);
    return sure;
}

template <class Acc>
struct SpfhConsumer {
    static constexpr bool WANTS_INNER = false;
    const float4* normals;   // original order
    Acc acc;
    unsigned char* spfh_cnt; // [N][36] integer SPFH histograms
    int* nbr_idx;            // [N][K]
    float* nbr_d2;           // [N][K]
    int* nbr_cnt;            // [N]
    int K;
    int qi;
    float4 q, qn;
    WaveLds* L;
    int ncount;
    __device__ void begin(int) {
        const int lane = threadIdx.x & 63;
        if (lane < 33) L->scratch[lane] = 0;
        ncount = 0;
        wave_lds_sync();
    }
    // selected candidates are only appended to the per-wave list here; the pair features (fp64, ~300 instructions)
    // are computed afterwards on densely packed lanes
    __device__ void accept(bool sel, int j, const float4&, float d2) {
        const int lane = threadIdx.x & 63;
        const unsigned long long m = __ballot(sel);
        if (sel) {
            const int pos = ncount + __popcll(m & ((1ull << lane) - 1ull));
            L->sel_j[pos] = j;
            L->sel_d2[pos] = d2;
        }
        ncount += __popcll(m);
    }
    __device__ void finish(int k) {
        const int lane = threadIdx.x & 63;
        int* hist = L->scratch;
        sort_selected_by_index<true>(L, acc, k);
        for (int t = lane; t < k; t += 64) {
            const int j = L->b_j[t];
            const int jo = acc.ord(j);
            nbr_idx[(int64_t)qi * K + t] = jo;
            nbr_d2[(int64_t)qi * K + t] = __uint_as_float(L->b_bits[t]);
            if (jo != qi) {
                double f[3];
                pair_features_d(q, qn, acc.pt(j), normals[jo], f);
                atomicAdd(&hist[clamp_bin11((int)f[0])], 1);
                atomicAdd(&hist[11 + clamp_bin11((int)floor(11 * (f[1] + 1.0) * 0.5))], 1);
                atomicAdd(&hist[22 + clamp_bin11((int)floor(11 * (f[2] + 1.0) * 0.5))], 1);
            }
        }
        wave_lds_sync();
        // SPFH(i)[b] = hist[b] * 100 / (k - 1): only the integer histogram (<= 255 per bin) is stored, 36 bytes per point,
        // so the FPFH pass gathers 36 B instead of 132 B per neighbour; the fp32 value is rebuilt on the fly
        if (lane < 36) spfh_cnt[(int64_t)qi * 36 + lane] = lane < 33 ? (unsigned char)hist[lane] : (unsigned char)0;
        if (lane == 0) nbr_cnt[qi] = k;
    }
};

// The 100-neighbour search and the normals in one pass (instance features: normal radius <= feature radius, <= 32 normal neighbours):
// the <= kn nearest of a point within the normal radius are among its k nearest within the feature radius, so the selected list
// serves both -- the neighbour lists are written for the SPFH / FPFH kernels that follow, and the normal's own neighbours are taken
// from the list (all entries inside the normal radius, or the kn smallest by (d2 bits, index) through a 64-bin histogram of the
// list) and parked for the batched solve, in the same ascending-index order as the stand-alone normals kernel: identical normals.
// This removes that kernel's whole search (2.9 of the 10.8 ms of the two searches per step).
#define NP_SLOTS_FUSED 8        // 80.9 KiB per workgroup with the 2 560-point tile: two workgroups per CU still fit
template <class Acc>
struct ListNormalConsumer {
    // tile_select_guess marks the normal's own neighbours (the <= kn nearest inside rn2) while it selects the list: they fall out of the
    // same histogram, so finish() need not find them again (its 64-bin histogram and rank loops were 4 300 of a query's 24 000 clocks)
    static constexpr bool WANTS_INNER = true;
    bool flagged = false;    // accept_in() was used: bit 31 of the stored d2 marks an inner neighbour
    // Round 3: the consumer no longer solves the normal.  It records WHICH entries of the query's neighbour list are the normal's
    // neighbours as a 128-bit mask over the list slots (the list is in ascending original index: the order of the covariance sums);
    // ibl_normals_from_mask_kernel then solves every point of the batch with one lane per point.  In this kernel the solve ran on 8 of
    // 64 lanes, every eighth query, behind ~250 VGPRs of fp64 code (the tile kernel spilled), and kept 2.3 KB of LDS per workgroup.
    unsigned* nrm_mask;      // out [N][4]
    Acc acc;
    int* nbr_idx;            // [N][K]
    float* nbr_d2;           // [N][K]
    int* nbr_cnt;            // [N]
    int K;
    float rn2;               // squared normal radius
    int kn;                  // normal neighbours (<= NP_MAXK)
    int qi;
    WaveLds* L;
    int ncount;
    __device__ void begin(int) { ncount = 0; }
    __device__ void accept(bool sel, int j, const float4&, float d2) {
        const int lane = threadIdx.x & 63;
        const unsigned long long m = __ballot(sel);
        if (sel) {
            const int pos = ncount + __popcll(m & ((1ull << lane) - 1ull));
            L->sel_j[pos] = j;
            L->sel_d2[pos] = d2;
        }
        ncount += __popcll(m);
    }
    __device__ void accept_in(bool sel, int j, float d2, bool inner) {
        const int lane = threadIdx.x & 63;
        const unsigned long long m = __ballot(sel);
        if (sel) {
            const int pos = ncount + __popcll(m & ((1ull << lane) - 1ull));
            L->sel_j[pos] = j;
            L->sel_d2[pos] = __uint_as_float(__float_as_uint(d2) | (inner ? 0x80000000u : 0u));
        }
        ncount += __popcll(m);
        flagged = true;
    }
    __device__ void finish_flagged(int k) {
        const int lane = threadIdx.x & 63;
        sort_selected_by_index<true>(L, acc, k);                  // b_j: handles, b_bits: d2 bits | inner flag, ascending original index
        unsigned long long mm[2] = {0ull, 0ull};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int t = 64 * c + lane;
            const bool v = t < k;
            const unsigned bits = v ? L->b_bits[t] : 0u;
            if (v) {
                nbr_idx[(int64_t)qi * K + t] = acc.ord(L->b_j[t]);
                nbr_d2[(int64_t)qi * K + t] = __uint_as_float(bits & 0x7FFFFFFFu);
            }
            mm[c] = __ballot(v && (bits >> 31) != 0u);
        }
        if (lane == 0) {
            nbr_cnt[qi] = k;
            *reinterpret_cast<uint4*>(nrm_mask + 4 * (int64_t)qi) =
                make_uint4((unsigned)mm[0], (unsigned)(mm[0] >> 32), (unsigned)mm[1], (unsigned)(mm[1] >> 32));
        }
    }
    __device__ void finish(int k) {
        if (flagged) { finish_flagged(k); return; }
        const int lane = threadIdx.x & 63;
        const unsigned long long lt_mask = (1ull << lane) - 1ull;
#ifdef KNN_LAB_CLK
        long long _clk_prev = (long long)__builtin_amdgcn_s_memtime();
#endif
        sort_selected_by_index<true>(L, acc, k);                  // b_j: handles, b_bits: d2 bits, ascending original index
        KNN_CLK(13);
        int n_in = 0;
        for (int t0 = 0; t0 < k; t0 += 64) {
            const int t = t0 + lane;
            const bool v = t < k;
            const unsigned bits = v ? L->b_bits[t] : 0x7F800000u;
            if (v) {
                nbr_idx[(int64_t)qi * K + t] = acc.ord(L->b_j[t]);
                nbr_d2[(int64_t)qi * K + t] = __uint_as_float(bits);
            }
            n_in += __popcll(__ballot(v && __uint_as_float(bits) < rn2));
        }
        if (lane == 0) nbr_cnt[qi] = k;
        KNN_CLK(14);
        // ---- the normal's neighbours
        int b30 = 64, need = 0, popb = 0;
        const float nscale = 64.0f / rn2;
        auto nbin = [&](float d2) { const int b = (int)(d2 * nscale); return b > 63 ? 63 : b; };
        if (n_in > kn) {
            int* nh = L->scratch;                                   // 64 bins over [0, rn2); free once the index sort is done
            nh[lane] = 0;
            wave_lds_sync();
            for (int t = lane; t < k; t += 64) {
                const float d2 = __uint_as_float(L->b_bits[t]);
                if (d2 < rn2) atomicAdd(&nh[nbin(d2)], 1);
            }
            wave_lds_sync();
            const int c = nh[lane];
            int incl = wave_incl_scan_i(c);
            const unsigned long long m = __ballot(incl >= kn);
            b30 = __ffsll((long long)m) - 1;
            need = kn - __shfl(incl - c, b30, 64);
            // the entries of the boundary bin, for the rank by (d2 bits, original index)
            for (int t0 = 0; t0 < k; t0 += 64) {
                const int t = t0 + lane;
                const float d2 = t < k ? __uint_as_float(L->b_bits[t]) : INFINITY;
                const bool bd = d2 < rn2 && nbin(d2) == b30;
                const unsigned long long mb = __ballot(bd);
                if (bd) {
                    const int pos = popb + __popcll(mb & lt_mask);
                    L->sel_d2[pos] = d2;
                    L->sel_j[pos] = t;                               // position in the index-sorted list = index order
                }
                popb += __popcll(mb);
            }
            wave_lds_sync();
        }
        unsigned long long mmk[2] = {0ull, 0ull};
        for (int t0 = 0; t0 < k; t0 += 64) {
            const int t = t0 + lane;
            const float d2 = t < k ? __uint_as_float(L->b_bits[t]) : INFINITY;
            bool member = d2 < rn2;
            if (member && n_in > kn) {
                const int b = nbin(d2);
                if (b > b30) member = false;
                else if (b == b30) {
                    const unsigned mb = __float_as_uint(d2);
                    int rank = 0;
                    for (int u = 0; u < popb; ++u) {
                        const unsigned ub = __float_as_uint(L->sel_d2[u]);
                        rank += (ub < mb || (ub == mb && L->sel_j[u] < t)) ? 1 : 0;
                    }
                    member = rank < need;
                }
            }
            mmk[t0 >> 6] = __ballot(member);
        }
        KNN_CLK(15);
        if (lane == 0)
            *reinterpret_cast<uint4*>(nrm_mask + 4 * (int64_t)qi) =
                make_uint4((unsigned)mmk[0], (unsigned)(mmk[0] >> 32), (unsigned)mmk[1], (unsigned)(mmk[1] >> 32));
    }
};

template <class Acc>
struct GradConsumer {
    static constexpr bool WANTS_INNER = false;
    const float4* normals;
    Acc acc;
    float4* grad;
    int qi;
    float4 q, qn;
    WaveLds* L;
    int ncount;
    __device__ void begin(int) { ncount = 0; }
    __device__ void accept(bool sel, int j, const float4&, float) {
        const int lane = threadIdx.x & 63;
        const unsigned long long m = __ballot(sel);
        if (sel) L->sel_j[ncount + __popcll(m & ((1ull << lane) - 1ull))] = j;
        ncount += __popcll(m);
    }
    __device__ void finish(int k) {
        const int lane = threadIdx.x & 63;
        sort_selected_by_index<false>(L, acc, k);
        double a[9];             // AtA (6 unique: 00 01 02 11 12 22) + Atb (3)
        for (int t = 0; t < 9; ++t) a[t] = 0.0;
        for (int t = lane; t < k; t += 64) {
            const int j = L->b_j[t];
            if (acc.ord(j) == qi) continue;
            const float4 p = acc.pt(j);
            const double vt[3] = {q.x, q.y, q.z}, nt[3] = {qn.x, qn.y, qn.z};
            const double dd[3] = {(double)p.x - vt[0], (double)p.y - vt[1], (double)p.z - vt[2]};
            const double pr = dot3d(dd, nt);
            const double r[3] = {(double)p.x - pr * nt[0] - vt[0], (double)p.y - pr * nt[1] - vt[1], (double)p.z - pr * nt[2] - vt[2]};
            const double b = (double)p.w - (double)q.w;
            a[0] += r[0] * r[0]; a[1] += r[0] * r[1]; a[2] += r[0] * r[2]; a[3] += r[1] * r[1]; a[4] += r[1] * r[2]; a[5] += r[2] * r[2];
            a[6] += r[0] * b; a[7] += r[1] * b; a[8] += r[2] * b;
        }
        for (int t = 0; t < 9; ++t) a[t] = wave_sum_d(a[t]);
        if (lane == 0) {
            double gx[3] = {0, 0, 0};
            if (k >= 4) {
                const double nt[3] = {qn.x, qn.y, qn.z};
                const double wgt = (double)(k - 1);
                double M[3][3] = {{a[0], a[1], a[2]}, {a[1], a[3], a[4]}, {a[2], a[4], a[5]}};
                for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M[r][c] += wgt * nt[r] * wgt * nt[c];
                const double B[3] = {a[6], a[7], a[8]};
                const double det = M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
                                   M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
                if (det != 0.0 && isfinite(det)) {
                    gx[0] = (B[0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (B[1] * M[2][2] - M[1][2] * B[2]) +
                             M[0][2] * (B[1] * M[2][1] - M[1][1] * B[2])) / det;
                    gx[1] = (M[0][0] * (B[1] * M[2][2] - M[1][2] * B[2]) - B[0] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
                             M[0][2] * (M[1][0] * B[2] - B[1] * M[2][0])) / det;
                    gx[2] = (M[0][0] * (M[1][1] * B[2] - B[1] * M[2][1]) - M[0][1] * (M[1][0] * B[2] - B[1] * M[2][0]) +
                             B[0] * (M[1][0] * M[2][1] - M[1][1] * M[2][0])) / det;
                }
            }
            grad[qi] = make_float4((float)gx[0], (float)gx[1], (float)gx[2], 0.0f);
        }
    }
};

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
// ---- the search on an LDS-staged neighbourhood ---------------------------------------------------------------------------------
// One wavefront, one query, `total` candidates in T (every point of the staging cube).  cover2 = squared distance from the query to
// the nearest face of the staging cube (INFINITY when the cube reaches past the search radius or the cloud's bounds on every side):
// all points closer than that are staged, so once max_nn of them are, the staged k nearest are the cloud's k nearest.  Returns
// false -- before any consumer call -- when that cannot be shown or the boundary bin overflows its list; the caller then runs
// hybrid_select for the query.  Selection rule and tie order (d2 bits, then original index) are those of hybrid_select.
#ifdef KNN_LAB_STATS
__device__ int knn_lab_stats[8];
#endif

// Round 3 fast path of tile_select: with a GUESS of the squared distance of the query's max_nn-th neighbour (the previous query of this
// wave -- a neighbour in the same tile -- plus a margin) ONE pass over the staged candidates collects those closer than the guess (two
// ballots and two LDS stores per 64 candidates: no histogram, no bins, no atomics in the pass that touches every candidate; it was 8 200
// of a query's 24 000 clocks).  The list then holds every staged candidate with d2 < guess; when it has at least max_nn entries (and at
// most the 256 the list holds) the max_nn nearest are among them -- anything outside is strictly farther -- and are selected from the
// list alone through a 256-bin histogram of the LIST over [0, guess): the same candidates, the same (d2 bits, original index) order as
// the two-pass selection below.  Returns 1 = done, 0 = not provable from the staged cube (the query joins the grid walk), -1 = the
// guess was too small or too large (the caller runs the two-pass selection, which also renews the guess).
template <class TL, class Consumer>
__device__ int tile_select_guess(const TL& T, int total, const float4 q, float cover2, float r2, int max_nn, WaveLds* L, Consumer& cons, float& gthr) {
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const float thr = fminf(gthr, r2);
#ifdef KNN_LAB_CLK
    long long _clk_prev = (long long)__builtin_amdgcn_s_memtime();
#endif
    int ccount = 0, in_ball = 0;
    // 128 candidates per step, both LDS reads of the NEXT step in flight while this one is evaluated (the pass is a chain of LDS round
    // trips otherwise: two waves per SIMD do not hide them)
    const int last = total - 1;
    float4 pa = T.pts[min(lane, last)], pb = T.pts[min(lane + 64, last)];
#pragma unroll 1
    for (int t0 = 0; t0 < total; t0 += 128) {
        const float4 p0 = pa, p1 = pb;
        pa = T.pts[min(t0 + 128 + lane, last)];
        pb = T.pts[min(t0 + 192 + lane, last)];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int t = t0 + 64 * h + lane;
            const float4 p = h ? p1 : p0;
            const float d2 = dist2f(q.x, q.y, q.z, p.x, p.y, p.z);
            const bool valid = t < total;
            in_ball += __popcll(__ballot(valid && d2 < cover2));
            const bool col = valid && d2 < thr;
            const unsigned long long mc = __ballot(col);
            if (col) {
                const int pos = ccount + __popcll(mc & lt_mask);
                if (pos < KNN_CAPB) { L->b_j[pos] = t; L->b_bits[pos] = __float_as_uint(d2); }
            }
            ccount += __popcll(mc);
        }
    }
    KNN_CLK(8);
    if (cover2 != INFINITY && in_ball < max_nn) return 0;
    if (ccount > KNN_CAPB || (ccount < max_nn && thr < r2)) return -1;
    wave_lds_sync();
    const int k = ccount < max_nn ? ccount : max_nn;
    const float4 none = make_float4(0.f, 0.f, 0.f, 0.f);
    // the list, four entries per lane
    unsigned eb[KNN_CAPB / 64];
    int et[KNN_CAPB / 64];
#pragma unroll
    for (int u = 0; u < KNN_CAPB / 64; ++u) {
        const int e = lane + 64 * u;
        eb[u] = e < ccount ? L->b_bits[e] : 0xFFFFFFFFu;
        et[u] = e < ccount ? L->b_j[e] : 0;
    }
    if (ccount <= max_nn) {                         // (thr == r2 here: the list is every staged candidate in radius)
        cons.begin(k);                              // (a consumer with an inner set finds it itself on this rare path)
#pragma unroll
        for (int u = 0; u < KNN_CAPB / 64; ++u) cons.accept(lane + 64 * u < ccount, et[u], none, __uint_as_float(eb[u]));
        cons.finish(k);
        return 1;
    }
    const float lscale = (float)KNN_BINS / thr;
    int ebin[KNN_CAPB / 64];
#pragma unroll
    for (int t = 0; t < KNN_BINS / 64; ++t) L->hist[lane * (KNN_BINS / 64) + t] = 0;
    wave_lds_sync();
#pragma unroll
    for (int u = 0; u < KNN_CAPB / 64; ++u) {
        const bool v = lane + 64 * u < ccount;
        ebin[u] = v ? (int)fminf(__uint_as_float(eb[u]) * lscale, (float)(KNN_BINS - 1)) : KNN_BINS;
        if (v) atomicAdd(&L->hist[ebin[u]], 1);
    }
    wave_lds_sync();
    int hb[KNN_BINS / 64];
    int s = 0;
#pragma unroll
    for (int t = 0; t < KNN_BINS / 64; ++t) { hb[t] = L->hist[lane * (KNN_BINS / 64) + t]; s += hb[t]; }
    int incl = wave_incl_scan_i(s);
    int bstar, n_below, pop;
    {
        const unsigned long long m = __ballot(incl >= max_nn);
        const int Lc = __ffsll((long long)m) - 1;
        int bin_here = 0, my_below = 0, my_pop = 0;
        int run = incl - s;
#pragma unroll
        for (int t = 0; t < KNN_BINS / 64; ++t) {
            if (my_pop == 0 && run + hb[t] >= max_nn) { bin_here = lane * (KNN_BINS / 64) + t; my_below = run; my_pop = hb[t]; }
            run += hb[t];
        }
        bstar = __shfl(bin_here, Lc, 64);
        n_below = __shfl(my_below, Lc, 64);
        pop = __shfl(my_pop, Lc, 64);
    }
    // The inner set of a consumer that wants one (the normal's <= kn nearest inside rn2): {d2 < rn2} is a prefix of the (d2, index) order
    // and so is the selection, hence n_in = min(k, list entries inside rn2); when n_in <= kn every selected entry inside rn2 belongs to
    // it, otherwise the kn first of the order do: the bins below the one that holds the kn-th entry + that bin's entries by rank.
    bool inner[KNN_CAPB / 64];
    bool use_flags = false;
    if constexpr (Consumer::WANTS_INNER) {
        const float rn2 = cons.rn2;
        const int kn = cons.kn;
        int n_in_list = 0;
#pragma unroll
        for (int u = 0; u < KNN_CAPB / 64; ++u) {
            inner[u] = lane + 64 * u < ccount && __uint_as_float(eb[u]) < rn2;
            n_in_list += __popcll(__ballot(inner[u]));
        }
        use_flags = true;
        if ((n_in_list < k ? n_in_list : k) > kn) {
            const unsigned long long m = __ballot(incl >= kn);
            const int Lc = __ffsll((long long)m) - 1;
            int bin_here = 0, my_below = 0, my_pop = 0;
            int run = incl - s;
#pragma unroll
            for (int t = 0; t < KNN_BINS / 64; ++t) {
                if (my_pop == 0 && run + hb[t] >= kn) { bin_here = lane * (KNN_BINS / 64) + t; my_below = run; my_pop = hb[t]; }
                run += hb[t];
            }
            const int b30 = __shfl(bin_here, Lc, 64);
            const int need30 = kn - __shfl(my_below, Lc, 64);
            const int pop30 = __shfl(my_pop, Lc, 64);
            if (pop30 > 64) use_flags = false;          // (the consumer finds its inner set itself)
            else {
                // the kn-th entry's bin: its entries by (d2 bits, original index), parked in the idle sort scratch
                int c30 = 0;
#pragma unroll
                for (int u = 0; u < KNN_CAPB / 64; ++u) {
                    const bool pk = ebin[u] == b30;
                    const unsigned long long mp = __ballot(pk);
                    if (pk) {
                        const int pos = c30 + __popcll(mp & lt_mask);
                        L->rank_pre[pos] = (int)eb[u];
                        L->scratch[pos] = T.ord_of(et[u]);
                    }
                    c30 += __popcll(mp);
                }
                wave_lds_sync();
#pragma unroll
                for (int u = 0; u < KNN_CAPB / 64; ++u) {
                    if (ebin[u] < b30) inner[u] = true;
                    else if (ebin[u] > b30) inner[u] = false;
                    else {
                        const int mi = T.ord_of(et[u]);
                        int rank = 0;
                        for (int w = 0; w < pop30; ++w) {
                            const unsigned ub = (unsigned)L->rank_pre[w];
                            rank += (ub < eb[u] || (ub == eb[u] && L->scratch[w] < mi)) ? 1 : 0;
                        }
                        inner[u] = rank < need30;
                    }
                }
                wave_lds_sync();
            }
        }
    }
    KNN_CLK(9);
    cons.begin(k);
    // entries below the threshold bin go to the consumer; the threshold bin's entries are parked at the head of the list arrays (every
    // lane holds its entries in registers by now)
    int bcount = 0;
#pragma unroll
    for (int u = 0; u < KNN_CAPB / 64; ++u) {
        if constexpr (Consumer::WANTS_INNER) {
            if (use_flags) cons.accept_in(ebin[u] < bstar, et[u], __uint_as_float(eb[u]), inner[u]);
            else cons.accept(ebin[u] < bstar, et[u], none, __uint_as_float(eb[u]));
        } else
        cons.accept(ebin[u] < bstar, et[u], none, __uint_as_float(eb[u]));
        const bool park = ebin[u] == bstar;
        const unsigned long long m = __ballot(park);
        if (park) {
            const int pos = bcount + __popcll(m & lt_mask);
            L->b_bits[pos] = eb[u];
            L->b_idx[pos] = T.ord_of(et[u]);
            bool fl = false;
            if constexpr (Consumer::WANTS_INNER) fl = use_flags && inner[u];
            L->b_j[pos] = et[u] | (fl ? 0x40000000 : 0);        // (bit 30: inner flag of a parked entry)
        }
        bcount += __popcll(m);
    }
    KNN_CLK(10);
    const int need = max_nn - n_below;
    wave_lds_sync();
    for (int e0 = 0; e0 < pop; e0 += 64) {
        const int e = e0 + lane;
        const bool v = e < pop;
        const unsigned mb = v ? L->b_bits[e] : 0xFFFFFFFFu;
        const int mi = v ? L->b_idx[e] : 0x7FFFFFFF;
        int rank = 0;
        for (int u = 0; u < pop; ++u) {
            const unsigned ub = L->b_bits[u];
            const int ui = L->b_idx[u];
            rank += (ub < mb || (ub == mb && ui < mi)) ? 1 : 0;
        }
        const int hj = v ? L->b_j[e] : 0;
        if constexpr (Consumer::WANTS_INNER) {
            if (use_flags) cons.accept_in(v && rank < need, hj & 0x3FFFFFFF, __uint_as_float(mb), (hj & 0x40000000) != 0);
            else cons.accept(v && rank < need, hj & 0x3FFFFFFF, none, __uint_as_float(mb));
        } else
        cons.accept(v && rank < need, hj & 0x3FFFFFFF, none, __uint_as_float(mb));
    }
    KNN_CLK(11);
    // next guess: the upper edge of the threshold bin + a margin (a 30-neighbour search has 8x head-room in the 256-entry list)
    gthr = fminf(r2, (float)(bstar + 1) / lscale * (max_nn * 4 <= KNN_CAPB ? 2.0f : 1.3f));
    cons.finish(k);
    KNN_CLK(12);
    return 1;
}

template <class TL, class Consumer>
__device__ bool tile_select(const TL& T, int total, const float4 q, float cover2, float r2, int max_nn, WaveLds* L, Consumer& cons, int& gbin) {
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const float bscale = (float)KNN_BINS / r2;
    auto bin_of = [&](float d2) { int b = (int)(d2 * bscale); return b > KNN_BINS - 1 ? KNN_BINS - 1 : b; };
#ifdef KNN_LAB_CLK
    long long _clk_prev = (long long)__builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int t = 0; t < KNN_BINS / 64; ++t) L->hist[lane * (KNN_BINS / 64) + t] = 0;
    wave_lds_sync();
    int in_ball = 0;
    // While the histogram is built, the candidates up to a GUESS of the threshold bin (the previous query of this wave, a neighbour in
    // the same tile, + 25 %) are parked in the boundary arrays, which are idle until the second pass.  When the guess covers the true
    // threshold bin and the list did not overflow, the selection below runs over that list (<= 256 entries) instead of all the staged
    // candidates a second time (~1 200): the same candidates in the same (d2 bits, index) order either way.
    const int guess = gbin;
    int ccount = 0;
    float4 pnext = T.pts[lane < total ? lane : 0];        // one step ahead: two waves per SIMD do not hide the LDS round trip
#pragma unroll 1
    for (int t0 = 0; t0 < total; t0 += 64) {
        const int t = t0 + lane;
        const float4 p = pnext;
        if (t0 + 64 < total) pnext = T.pts[t + 64 < total ? t + 64 : 0];
        const float d2 = dist2f(q.x, q.y, q.z, p.x, p.y, p.z);
        const bool in = t < total && d2 < r2;
        const int b = in ? bin_of(d2) : KNN_BINS;
        if (in) atomicAdd(&L->hist[b], 1);
        in_ball += __popcll(__ballot(in && d2 < cover2));
        const bool col = b <= guess;
        const unsigned long long mc = __ballot(col);
        if (col) {
            const int pos = ccount + __popcll(mc & lt_mask);
            if (pos < KNN_CAPB) { L->b_j[pos] = t; L->b_bits[pos] = __float_as_uint(d2); }
        }
        ccount += __popcll(mc);
    }
    wave_lds_sync();
    KNN_CLK(8);
    int hb[KNN_BINS / 64];
    int s = 0;
#pragma unroll
    for (int t = 0; t < KNN_BINS / 64; ++t) { hb[t] = L->hist[lane * (KNN_BINS / 64) + t]; s += hb[t]; }
    int incl = wave_incl_scan_i(s);
    const int cnt = __shfl(incl, 63, 64);
    if (cover2 != INFINITY && in_ball < max_nn) return false;
    const int excl = incl - s;
    const int k = cnt < max_nn ? cnt : max_nn;
    const bool select_all = cnt <= max_nn;
    int bstar = KNN_BINS, n_below = 0, pop = 0;
    if (!select_all) {
        const unsigned long long m = __ballot(incl >= max_nn);
        const int Lc = __ffsll((long long)m) - 1;
        int bin_here = 0, my_below = 0, my_pop = 0;
        int run = excl;
#pragma unroll
        for (int t = 0; t < KNN_BINS / 64; ++t) {
            if (my_pop == 0 && run + hb[t] >= max_nn) { bin_here = lane * (KNN_BINS / 64) + t; my_below = run; my_pop = hb[t]; }
            run += hb[t];
        }
        bstar = __shfl(bin_here, Lc, 64);
        n_below = __shfl(my_below, Lc, 64);
        pop = __shfl(my_pop, Lc, 64);
        if (pop > KNN_CAPB) return false;
        // (a 30-neighbour search has 8x head-room in the 256-entry list: its guess doubles the bin instead of adding a quarter)
        gbin = min(KNN_BINS - 1, bstar + (max_nn * 4 <= KNN_CAPB ? bstar : (bstar >> KNN_GUESS_SHIFT)) + 2);
    }
    cons.begin(k);
    KNN_CLK(9);
    int bcount = 0;
    const float4 none = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool from_list = !select_all && bstar <= guess && ccount <= KNN_CAPB;
#ifdef KNN_LAB_STATS
    if (lane == 0) { atomicAdd(&knn_lab_stats[0], 1); if (from_list) atomicAdd(&knn_lab_stats[1], 1); if (select_all) atomicAdd(&knn_lab_stats[2], 1); if (!select_all && bstar > guess) atomicAdd(&knn_lab_stats[3], 1); if (ccount > KNN_CAPB) atomicAdd(&knn_lab_stats[4], 1); }
#endif
    if (from_list) {
#pragma unroll 1
        for (int e0 = 0; e0 < ccount; e0 += 64) {
            const int e = e0 + lane;
            const bool v = e < ccount;
            const int t = v ? L->b_j[e] : 0;
            const unsigned bits = v ? L->b_bits[e] : 0u;
            const float d2 = __uint_as_float(bits);
            const int b = v ? bin_of(d2) : KNN_BINS;
            cons.accept(b < bstar, t, none, d2);
            // the boundary bin is compacted in place: a parked entry lands at or before the slot it was read from, and every lane of
            // this step has read its slot before the first store is issued
            const bool park = b == bstar;
            const unsigned long long m = __ballot(park);
            if (park) {
                const int pos = bcount + __popcll(m & lt_mask);
                L->b_bits[pos] = bits;
                L->b_idx[pos] = T.ord_of(t);
                L->b_j[pos] = t;
            }
            bcount += __popcll(m);
        }
    } else
#pragma unroll 1
    for (int t0 = 0; t0 < total; t0 += 64) {
        const int t = t0 + lane;
        const float4 p = T.pts[t < total ? t : 0];
        const float d2 = dist2f(q.x, q.y, q.z, p.x, p.y, p.z);        // the same bits as in pass 1
        const bool in = t < total && d2 < r2;
        const int b = in ? bin_of(d2) : KNN_BINS;
        cons.accept(in && (select_all || b < bstar), t, none, d2);
        if (!select_all) {
            const bool park = in && b == bstar;
            const unsigned long long m = __ballot(park);
            if (park) {
                const int pos = bcount + __popcll(m & lt_mask);
                L->b_bits[pos] = __float_as_uint(d2);
                L->b_idx[pos] = T.ord_of(t);
                L->b_j[pos] = t;
            }
            bcount += __popcll(m);
        }
    }
    KNN_CLK(10);
    if (!select_all) {
        // the boundary bin: an entry is selected when fewer than `need` entries precede it in (d2 bits, original index) order
        const int need = max_nn - n_below;
        wave_lds_sync();
        for (int e0 = 0; e0 < pop; e0 += 64) {
            const int e = e0 + lane;
            const bool v = e < pop;
            const unsigned mb = v ? L->b_bits[e] : 0xFFFFFFFFu;
            const int mi = v ? L->b_idx[e] : 0x7FFFFFFF;
            int rank = 0;
            for (int u = 0; u < pop; ++u) {
                const unsigned ub = L->b_bits[u];
                const int ui = L->b_idx[u];
                rank += (ub < mb || (ub == mb && ui < mi)) ? 1 : 0;
            }
            const bool sel = v && rank < need;
            cons.accept(sel, v ? L->b_j[e] : 0, none, __uint_as_float(mb));
        }
    }
    KNN_CLK(11);
    cons.finish(k);
    KNN_CLK(12);
    return true;
}

// Workgroup per tile: stage the neighbourhood, then every wavefront takes queries of the tile in turn.  need_pop[rho] = candidates
// the staging cube of reach rho should hold for the ball inside it to contain max_nn of them (host: 1.15 max_nn (ts + 2 rho)^2 /
// (pi rho^2)); the reach grows from 2 until it does, the cube fills the LDS budget or covers the whole radius.
struct NeedPop { float v[8]; int rho_start; int guess; };       // guess: tile_select_guess on (0: IBL_KNN_NOGUESS=1, the two-pass selection only)

template <int TS, int NW, class TL, class Factory>
__device__ void tile_knn_block(const BatchGrid& g, float radius, float r2, int max_nn, const NeedPop& need_pop, int* __restrict__ fb_list,
                               int* __restrict__ fb_count, const Factory& fac, int q_lo, int q_hi, TL& T, WaveLds* wl,
                               typename Factory::Pending* pend) {
    constexpr int KT_CAP = TL::CAP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    if (tile >= g.n_tiles) return;
#ifdef KNN_LAB_CLK
    long long _clk_prev = (long long)__builtin_amdgcn_s_memtime();
#endif
    int lo = 0, hi = g.n_seg;                      // the segment whose tile range holds `tile`
    if (g.tile_seg) lo = g.tile_seg[tile];
    else
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (g.tile_base[mid] <= tile) lo = mid; else hi = mid;
        }
    const SegGrid sg = g.seg[lo];
    const int ntx = (sg.nx + TS - 1) / TS, nty = (sg.ny + TS - 1) / TS;
    int tt = tile - g.tile_base[lo];
    const int tx = tt % ntx; tt /= ntx;
    const int ty = tt % nty, tz = tt / nty;
    const int cx0 = tx * TS, cx1 = min(cx0 + TS, sg.nx) - 1;
    const int cy0 = ty * TS, cy1 = min(cy0 + TS, sg.ny) - 1;
    const int cz0 = tz * TS, cz1 = min(cz0 + TS, sg.nz) - 1;
    // the tile's queries: runs of the cell-sorted array, one per (z, y) row of the tile
    const int qny = cy1 - cy0 + 1, qrows = qny * (cz1 - cz0 + 1);
    if (tid < qrows) {
        const int row = sg.cell_base + ((cz0 + tid / qny) * sg.ny + cy0 + tid % qny) * sg.nx;
        const int b = g.cell_start[row + cx0];
        T.q_b[tid] = b;
        T.q_off[tid + 1] = g.cell_start[row + cx1 + 1] - b;
    }
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        T.q_off[0] = 0;
        for (int r = 0; r < qrows; ++r) { acc += T.q_off[r + 1]; T.q_off[r + 1] = acc; }
        T.next_q = NW;
    }
    __syncthreads();
    const int nq = T.q_off[qrows];
    KNN_CLK(0);
    if (nq == 0) return;

    int reach_max = (int)ceilf(radius * sg.inv);
    if (reach_max < 1) reach_max = 1;
    const int rho_cap = min(reach_max, TS <= 2 ? 5 : 4);            // (TS + 2 rho)^2 <= KT_ROWS
    int rho = min(need_pop.rho_start, rho_cap);
    int x0, x1, y0, y1, z0, z1, total = 0;
    bool staged = false, final_try = false;
    for (int it = 0; it < 8; ++it) {
        x0 = max(cx0 - rho, 0); x1 = min(cx1 + rho, sg.nx - 1);
        y0 = max(cy0 - rho, 0); y1 = min(cy1 + rho, sg.ny - 1);
        z0 = max(cz0 - rho, 0); z1 = min(cz1 + rho, sg.nz - 1);
        const int ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
        __syncthreads();                              // row tables of the previous attempt are no longer read
        if (tid < nrows) {
            const int row = sg.cell_base + ((z0 + tid / ny) * sg.ny + y0 + tid % ny) * sg.nx;
            const int b = g.cell_start[row + x0];
            T.row_b[tid] = b;
            T.row_off[tid + 1] = g.cell_start[row + x1 + 1] - b;
        }
        __syncthreads();
        if (wave == 0) {                              // exclusive prefix of <= 144 row lengths: three per lane + a wave scan
            int v[3], sum = 0;
#pragma unroll
            for (int u = 0; u < 3; ++u) { const int r = 3 * lane + u; v[u] = r < nrows ? T.row_off[r + 1] : 0; sum += v[u]; }
            int incl = wave_incl_scan_i(sum);
            int run = incl - sum;
            if (lane == 0) T.row_off[0] = 0;
#pragma unroll
            for (int u = 0; u < 3; ++u) { const int r = 3 * lane + u; run += v[u]; if (r < nrows) T.row_off[r + 1] = run; }
        }
        __syncthreads();
        total = T.row_off[nrows];
        if (total > KT_CAP) {
            if (rho > 1 && !final_try) { --rho; final_try = true; continue; }
            break;                                     // does not fit at any reach: every query takes the global walk
        }
        if (final_try || rho >= rho_cap || (float)total >= need_pop.v[rho]) { staged = true; break; }
        ++rho;
    }
    const float cellw = 1.0f / sg.inv;
    KNN_CLK(1);
    if (staged) {
        const int ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
        for (int t = tid; t < total; t += NW * 64) {
            int a = 0, b = nrows;                      // the row r with row_off[r] <= t < row_off[r + 1]
            while (b - a > 1) {
                const int mid = (a + b) >> 1;
                if (T.row_off[mid] <= t) a = mid; else b = mid;
            }
            const int j = T.row_b[a] + (t - T.row_off[a]);
            float4 sp = g.sorted_pts[j];
            if (TL::PACKED) sp.w = __int_as_float(g.order[j]);
            else T.ord[t] = g.order[j];
            T.pts[t] = sp;
        }
    }
    __syncthreads();
    KNN_CLK(2);
    // faces of the staging cube (none where it reaches the cloud's bounds: nothing lies beyond), pulled in by the rounding margin of
    // the cell index
    const float mgn = 1e-3f * cellw;
    const float fx0 = x0 > 0 ? sg.minx + (float)x0 * cellw + mgn : -INFINITY, fx1 = x1 < sg.nx - 1 ? sg.minx + (float)(x1 + 1) * cellw - mgn : INFINITY;
    const float fy0 = y0 > 0 ? sg.miny + (float)y0 * cellw + mgn : -INFINITY, fy1 = y1 < sg.ny - 1 ? sg.miny + (float)(y1 + 1) * cellw - mgn : INFINITY;
    const float fz0 = z0 > 0 ? sg.minz + (float)z0 * cellw + mgn : -INFINITY, fz1 = z1 < sg.nz - 1 ? sg.minz + (float)(z1 + 1) * cellw - mgn : INFINITY;
    WaveLds* L = &wl[wave];
    typename Factory::Pending* P = &pend[wave];
    if (lane == 0) P->n = 0;
    wave_lds_sync();
    // original-index range of the tile's segment (points are sorted by (segment, cell): a segment's sorted positions are its indices)
    const int key_base = g.cell_start[sg.cell_base];
    const TileAcc tacc{T.pts, TL::PACKED ? nullptr : T.ord, key_base, g.cell_start[sg.cell_base + sg.nx * sg.ny * sg.nz] - key_base};
    // threshold-bin guess carried from query to query of this wave.  To start (a tile holds ~9 queries, so the four waves' first
    // queries are 4 of 9): a pilot -- the whole workgroup histograms the staged candidates around the tile's middle query (five steps
    // of 256 threads) and the bin that holds its max_nn-th neighbour, + the margin, seeds every wave.  A guess that is too small or
    // too large only costs that query the full second pass.
    int gbin = -1;
    float gthr = -1.0f;                            // round 3: the guess as a squared distance (tile_select_guess)
    if (staged && nq > 4) {
        int* ph = wl[0].hist;                          // idle until the first query
        __shared__ int pilot_bin;
        for (int t = tid; t < KNN_BINS; t += NW * 64) ph[t] = 0;
        __syncthreads();
        int pr = 0;
        const int pk = nq >> 1;
        while (pk >= T.q_off[pr + 1]) ++pr;
        const float4 pq = g.sorted_pts[T.q_b[pr] + (pk - T.q_off[pr])];
        const float bscale = (float)KNN_BINS / r2;
        for (int t = tid; t < total; t += NW * 64) {
            const float4 p = T.pts[t];
            const float d2 = dist2f(pq.x, pq.y, pq.z, p.x, p.y, p.z);
            if (d2 < r2) atomicAdd(&ph[min((int)(d2 * bscale), KNN_BINS - 1)], 1);
        }
        __syncthreads();
        if (wave == 0) {
            int hb4[KNN_BINS / 64], sum = 0;
#pragma unroll
            for (int u = 0; u < KNN_BINS / 64; ++u) { hb4[u] = ph[lane * (KNN_BINS / 64) + u]; sum += hb4[u]; }
            int incl = wave_incl_scan_i(sum);
            const unsigned long long m = __ballot(incl >= max_nn);
            if (lane == 0) pilot_bin = -1;
            if (m != 0ull && lane == __ffsll((long long)m) - 1) {
                int run = incl - sum, bb = lane * (KNN_BINS / 64);
#pragma unroll
                for (int u = 0; u < KNN_BINS / 64; ++u) {
                    if (run + hb4[u] >= max_nn) { bb = lane * (KNN_BINS / 64) + u; break; }
                    run += hb4[u];
                }
                pilot_bin = bb;
            }
        }
        __syncthreads();
        const int pb = pilot_bin;
        if (pb >= 0) {
            gbin = min(KNN_BINS - 1, pb + (max_nn * 4 <= KNN_CAPB ? pb : (pb >> KNN_GUESS_SHIFT)) + 2);
            gthr = fminf(r2, (float)(gbin + 1) / bscale);
        }
        __syncthreads();                               // wl[0].hist is wave 0's again
    }
    const bool use_guess = need_pop.guess != 0;
    KNN_CLK(3);
#ifdef KNN_LAB_CLK
    if (tid == 0) { atomicAdd(&knn_lab_clk[6], 1ull); atomicAdd(&knn_lab_clk[7], (unsigned long long)nq); }
#endif
    int run_r = 0;
    const int sny = y1 - y0 + 1;
    // A wave's first query is its own number; after that it takes the next one nobody has taken (queries differ in cost -- list or
    // two-pass selection, candidates in reach -- and a tile ends with its slowest wave: taking them in turn left the others idle)
    auto next_query = [&]() {
        int v = 0;
        if (lane == 0) v = atomicAdd(&T.next_q, 1);
        return __builtin_amdgcn_readfirstlane(v);
    };
    for (int qk = wave; qk < nq; qk = next_query()) {
        while (qk >= T.q_off[run_r + 1]) ++run_r;
        const int jq = T.q_b[run_r] + (qk - T.q_off[run_r]);
        // a query is a point of the tile, and the tile lies inside its staged cube: the point and its original index come from LDS (the two
        // dependent global loads per query were ~15 % of a tile's time with two waves per SIMD to hide them)
        int qi;
        float4 q;
        if (staged) {
            const int sr = (cz0 + run_r / qny - z0) * sny + (cy0 + run_r % qny - y0);
            const int ts_ = T.row_off[sr] + (jq - T.row_b[sr]);
            qi = T.ord_of(ts_);
            q = T.pts[ts_];
        } else {
            qi = g.order[jq];
            q = g.sorted_pts[jq];
        }
        if (qi < q_lo || qi >= q_hi) continue;
        bool done = false;
        if (staged) {
            float cover = fminf(fminf(q.x - fx0, fx1 - q.x), fminf(fminf(q.y - fy0, fy1 - q.y), fminf(q.z - fz0, fz1 - q.z)));
            if (cover < 0.f) cover = 0.f;
            const float cover2 = cover >= radius ? INFINITY : cover * cover;
            auto cons = fac.template make<TileAcc>(qi, q, L, tacc, P);
            int fast = -1;
            if (use_guess && gthr > 0.0f) fast = tile_select_guess(T, total, q, cover2, r2, max_nn, L, cons, gthr);
            if (fast < 0) {
                done = tile_select(T, total, q, cover2, r2, max_nn, L, cons, gbin);
                if (done && gbin >= 0) gthr = fminf(r2, (float)(gbin + 1) * r2 / (float)KNN_BINS);
            } else done = fast == 1;
        }
        // not provable from the staged cube (sparse spot, LDS budget, boundary-bin overflow): the query joins the list of the
        // per-query grid walk that runs after this kernel (ibl_knn_list_kernel)
        if (!done && lane == 0) fb_list[atomicAdd(fb_count, 1)] = jq;
    }
    KNN_CLK(4);
    fac.flush_block(pend, tacc);            // the queries still parked (normals)
    KNN_CLK(5);
}

struct NormalFactory {
    static constexpr bool WIDE = false;      // (its pending normals are solved by a 4-wave block flush)
    typedef NormalPending Pending;
    float4* normals;
    template <class Acc> __device__ NormalConsumer<Acc> make(int qi, const float4&, WaveLds* L, const Acc& acc, Pending* P = nullptr) const {
        NormalConsumer<Acc> c;
        c.normals = normals; c.acc = acc; c.L = L; c.P = P; c.qi = qi; c.ncount = 0;
        return c;
    }
    template <class Acc> __device__ void flush(Pending* P, const Acc& acc) const { normal_flush(P, acc, normals); }
    template <class Acc> __device__ void flush_block(Pending* pend, const Acc& acc) const { normal_flush_block(pend, acc, normals); }
};
struct SpfhFactory {
    static constexpr bool WIDE = true;       // geometry only: the packed tile (original index in the staged point's w)
    typedef NoPending Pending;
    const float4* normals; unsigned char* spfh_cnt; int* nbr_idx; float* nbr_d2; int* nbr_cnt; int K;
    template <class Acc> __device__ void flush(Pending*, const Acc&) const {}
    template <class Acc> __device__ void flush_block(Pending*, const Acc&) const {}
    template <class Acc> __device__ SpfhConsumer<Acc> make(int qi, const float4& q, WaveLds* L, const Acc& acc, Pending* = nullptr) const {
        SpfhConsumer<Acc> c;
        c.normals = normals; c.acc = acc; c.spfh_cnt = spfh_cnt; c.nbr_idx = nbr_idx; c.nbr_d2 = nbr_d2; c.nbr_cnt = nbr_cnt; c.K = K;
        c.qi = qi; c.q = q; c.qn = normals[qi]; c.L = L; c.ncount = 0;
        return c;
    }
};
struct ListNormalFactory {
    static constexpr bool WIDE = true;
    typedef NoPending Pending;
    unsigned* nrm_mask; int* nbr_idx; float* nbr_d2; int* nbr_cnt; int K; float rn2; int kn;
    template <class Acc> __device__ void flush(Pending*, const Acc&) const {}
    template <class Acc> __device__ void flush_block(Pending*, const Acc&) const {}
    template <class Acc> __device__ ListNormalConsumer<Acc> make(int qi, const float4&, WaveLds* L, const Acc& acc, Pending* = nullptr) const {
        ListNormalConsumer<Acc> c;
        c.nrm_mask = nrm_mask; c.acc = acc; c.nbr_idx = nbr_idx; c.nbr_d2 = nbr_d2; c.nbr_cnt = nbr_cnt; c.K = K; c.rn2 = rn2; c.kn = kn;
        c.qi = qi; c.L = L; c.ncount = 0;
        return c;
    }
};
struct GradFactory {
    static constexpr bool WIDE = false;      // (reads the staged intensity: no room for the index in w)
    typedef NoPending Pending;
    const float4* normals; float4* grad;
    template <class Acc> __device__ void flush(Pending*, const Acc&) const {}
    template <class Acc> __device__ void flush_block(Pending*, const Acc&) const {}
    template <class Acc> __device__ GradConsumer<Acc> make(int qi, const float4& q, WaveLds* L, const Acc& acc, Pending* = nullptr) const {
        GradConsumer<Acc> c;
        c.normals = normals; c.acc = acc; c.grad = grad; c.qi = qi; c.q = q; c.qn = normals[qi]; c.L = L; c.ncount = 0;
        return c;
    }
};

// tiles: one workgroup each.  NW waves per workgroup; PACK: original indices in the staged points' w (see TileLds)
template <int TS, int CAP, class Factory, int NW = 4, bool PACK = false, int OCC = 1>
__global__ __launch_bounds__(NW * 64, OCC) void ibl_knn_tile_kernel(BatchGrid g, float radius, float r2, int max_nn, NeedPop need_pop, Factory fac,
                                                              int q_lo, int q_hi, int* __restrict__ fb_list, int* __restrict__ fb_count) {
    __shared__ TileLds<CAP, PACK> T;
    __shared__ WaveLds wl[NW];
    __shared__ typename Factory::Pending pend[NW];
#ifdef KNN_LAB_PAD_LDS       // lab: pad the workgroup's LDS so that only one fits a CU (is the kernel bound by resident waves?)
    __shared__ int lab_pad[KNN_LAB_PAD_LDS / 4];
    if (threadIdx.x == 0 && q_lo == -12345) lab_pad[max_nn] = 1;
#endif
    tile_knn_block<TS, NW>(g, radius, r2, max_nn, need_pop, fb_list, fb_count, fac, q_lo, q_hi, T, wl, pend);
}

// the queries the tile kernel could not answer from its staged cubes: one wavefront each, walking the grid (sorted positions in list)
template <class Factory>
__global__ __launch_bounds__(256) void ibl_knn_list_kernel(BatchGrid g, const int* __restrict__ seg_off, float radius, float r2, int max_nn,
                                                           Factory fac, const int* __restrict__ list, const int* __restrict__ count, int* status) {
    __shared__ WaveLds lds[4];
    const int n = *count;
    for (int e = blockIdx.x * 4 + (threadIdx.x >> 6); e < n; e += gridDim.x * 4) {
        const int jq = list[e];
        const int qi = g.order[jq];
        const float4 q = g.sorted_pts[jq];
        const int s = seg_of(seg_off, g.n_seg, qi);
        auto cons = fac.template make<GlobalAcc>(qi, q, &lds[threadIdx.x >> 6], GlobalAcc{g.sorted_pts, g.order});
        hybrid_select(g, g.seg[s], q, qi, radius, r2, max_nn, &lds[threadIdx.x >> 6], cons, status);
    }
}

// no tiles (grids of ibl_build_batch_grid): one wavefront per query walks the grid
template <class Factory>
__global__ __launch_bounds__(256) void ibl_knn_query_kernel(BatchGrid g, const float4* __restrict__ pts, const int* __restrict__ seg_off,
                                                            float radius, float r2, int max_nn, Factory fac, int q0, int q1, int* status) {
    __shared__ WaveLds lds[4];
    const int qi = q0 + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= q1) return;
    const int s = seg_of(seg_off, g.n_seg, qi);
    const float4 q = pts[qi];
    auto cons = fac.template make<GlobalAcc>(qi, q, &lds[threadIdx.x >> 6], GlobalAcc{g.sorted_pts, g.order});
    hybrid_select(g, g.seg[s], q, qi, radius, r2, max_nn, &lds[threadIdx.x >> 6], cons, status);
}

// Position of histogram bin b in the "matching order" of the feature search (reg_register.hip, oracle_reg.c FEAT_ORDER: the
// three histograms from their centre bins outwards, interleaved).  Instance features are stored in that order so that the
// search reads the terms of its early-abandon chain contiguously.
__constant__ int FEAT_POS[33] = {29, 23, 17, 11, 5, 2, 8, 14, 20, 26, 32, 27, 21, 15, 9, 3, 0, 6, 12, 18, 24, 30, 28, 22, 16, 10, 4, 1, 7, 13, 19, 25, 31};

// FPFH(i) = 100 * sum_k SPFH(k)/d2_k / blocksum + SPFH(i)   (one wave per point, lanes = bins)
__device__ __forceinline__ float spfh_value(const unsigned char* __restrict__ spfh_cnt, const int* __restrict__ nbr_cnt, int j, int b) {
    const int kj = nbr_cnt[j];
    if (kj <= 1) return 0.0f;
    return (float)((double)spfh_cnt[(int64_t)j * 36 + b] * (100.0 / (double)(kj - 1)));
}

// Neighbour rows are gathered 64 at a time (one per lane, all loads in flight together) into a per-wave LDS tile;
// the lanes then switch roles to "one histogram bin each" and walk the tile.  The sum over neighbours keeps the list order.
struct FpfhTile {
    unsigned char cnt[64][36];
    double w[64];         // (100 / (k_j - 1)) / d2: the weight of the neighbour's integer histogram; 0 for a skipped entry (the point
                          // itself, zero distance, a neighbour without SPFH)
};

// A workgroup serves FPFH_Q = 7 points: thread p < 231 owns (point p / 33, bin p % 33), so 231 of its 256 lanes work (one wave per
// point left 31 of 64 idle in the loop below, which is the whole kernel).  Neighbour rows are staged 64 per point at a time; each
// (point, bin) sum runs over the neighbour list in its stored order, exactly as before.
#define FPFH_Q 7
__global__ __launch_bounds__(256) void ibl_fpfh_kernel(const unsigned char* __restrict__ spfh_cnt, const int* __restrict__ nbr_idx,
                                                       const float* __restrict__ nbr_d2, const int* __restrict__ nbr_cnt, int K, int n,
                                                       int matching_order, float* __restrict__ fpfh) {
    __shared__ FpfhTile tiles[FPFH_Q];
    __shared__ double accs[FPFH_Q][33];
    __shared__ int kq[FPFH_Q];
    const int tid = threadIdx.x;
    const int q0 = IBL_XCD_BLOCK(blockIdx.x, gridDim.x) * FPFH_Q;
    if (tid < FPFH_Q) kq[tid] = q0 + tid < n ? nbr_cnt[q0 + tid] : 0;
    __syncthreads();
    int kmax = 0;
#pragma unroll
    for (int u = 0; u < FPFH_Q; ++u) kmax = max(kmax, kq[u]);
    const int qq = tid / 33, b = tid - qq * 33;            // tid >= 231: no work (qq == 7)
    const bool mine = qq < FPFH_Q && q0 + qq < n;
    const int qi = q0 + qq;
    const int k = mine ? kq[qq] : 0;
    double acc = 0.0;
    for (int t0 = 0; t0 < kmax; t0 += 64) {
        for (int e = tid; e < FPFH_Q * 64; e += 256) {     // stage row t0 + r of point sq
            const int sq = e >> 6, r = e & 63, t = t0 + r, si = q0 + sq;
            if (si < n && t < kq[sq]) {
                FpfhTile& S = tiles[sq];
                const int j = nbr_idx[(int64_t)si * K + t];
                const double dist = (double)nbr_d2[(int64_t)si * K + t];
                const int kj = nbr_cnt[j];
                // a 36-byte row as 16 + 16 + 4 bytes (rows are 4-byte aligned: the 4-byte aligned struct makes these two
                // global_load_dwordx4 + one dword instead of nine dword gathers per row)
                const unsigned char* src = spfh_cnt + (int64_t)j * 36;
                const ibl_u4_a4 r0 = *reinterpret_cast<const ibl_u4_a4*>(src), r1 = *reinterpret_cast<const ibl_u4_a4*>(src + 16);
                const unsigned r2w = *reinterpret_cast<const unsigned*>(src + 32);
                unsigned int* dst = reinterpret_cast<unsigned int*>(S.cnt[r]);
                dst[0] = r0.x; dst[1] = r0.y; dst[2] = r0.z; dst[3] = r0.w; dst[4] = r1.x; dst[5] = r1.y; dst[6] = r1.z; dst[7] = r1.w; dst[8] = r2w;
                // SPFH(j)[b] / d2 = count x (increment / d2) in double (Open3D keeps its histograms in double; oracle_fpfh likewise): one
                // division per neighbour, one fma per (neighbour, bin); a zero weight or an empty bin adds an exact zero
                S.w[r] = (j == si || dist == 0.0 || kj <= 1) ? 0.0 : (100.0 / (double)(kj - 1)) / dist;
            }
        }
        __syncthreads();
        if (mine && k > 1) {
            const FpfhTile& T = tiles[qq];
            const int m = min(64, k - t0);
            // (count as a double without v_cvt_f64_u32, a quarter-rate instruction that was the hottest of this loop: 2^52 + c has c in its
            // low mantissa bits, and subtracting 2^52 leaves c exactly)
            for (int r = 0; r < m; ++r)
                acc = fma(__hiloint2double(0x43300000, (int)T.cnt[r][b]) - 4503599627370496.0, T.w[r], acc);
        }
        __syncthreads();
    }
    if (mine) accs[qq][b] = acc;
    __syncthreads();
    if (mine) {
        // block sums over bins 0-10, 11-21, 22-32
        double sum = 0.0;
        const int blk = b / 11;
        for (int t = 0; t < 11; ++t) sum += accs[qq][blk * 11 + t];
        float out = 0.0f;
        if (k > 1) {
            const double sc = sum != 0.0 ? 100.0 / sum : 0.0;
            out = (float)(acc * sc + (double)spfh_value(spfh_cnt, nbr_cnt, qi, b));
        }
        fpfh[(int64_t)qi * 33 + (matching_order ? FEAT_POS[b] : b)] = out;
    }
}

// radius outlier: keep[i] = (#points with d2 < r2, self included) > nb_points.  Thread per point, early exit.
__global__ __launch_bounds__(256) void ibl_radius_count_kernel(BatchGrid g, const float4* __restrict__ pts, const int* __restrict__ seg_off,
                                                               float radius, float r2, int nb_points, unsigned char* __restrict__ keep) {
    const int n = seg_off[g.n_seg];
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= n) return;
    const SegGrid sg = g.seg[seg_of(seg_off, g.n_seg, qi)];
    const float4 q = pts[qi];
    int reach = (int)ceilf(radius * sg.inv);
    if (reach < 1) reach = 1;
    const int cx = cell_clamp(q.x, sg.minx, sg.inv, sg.nx), cy = cell_clamp(q.y, sg.miny, sg.inv, sg.ny),
              cz = cell_clamp(q.z, sg.minz, sg.inv, sg.nz);
    int cnt = 0;
    for (int z = max(cz - reach, 0); z <= min(cz + reach, sg.nz - 1) && cnt <= nb_points; ++z)
        for (int y = max(cy - reach, 0); y <= min(cy + reach, sg.ny - 1) && cnt <= nb_points; ++y) {
            const int row = sg.cell_base + (z * sg.ny + y) * sg.nx;
            const int b = g.cell_start[row + max(cx - reach, 0)], e = g.cell_start[row + min(cx + reach, sg.nx - 1) + 1];
            for (int j = b; j < e && cnt <= nb_points; j += 4) {         // four loads in flight (the walk is a chain of load latencies)
                float4 p[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) p[v] = g.sorted_pts[min(j + v, e - 1)];
#pragma unroll
                for (int v = 0; v < 4; ++v) cnt += (j + v < e && dist2f(q.x, q.y, q.z, p[v].x, p[v].y, p[v].z) < r2) ? 1 : 0;
            }
        }
    keep[qi] = cnt > nb_points ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// host launchers (used by reg_api.hip)
// ------------------------------------------------------------------------------------------------
static NeedPop need_pop_table(int max_nn, int ts) {
    NeedPop np;
    np.v[0] = 0.f;
    for (int rho = 1; rho < 8; ++rho)
        np.v[rho] = (float)(ibl_knn_safety() * max_nn * (ts + 2.0 * rho) * (ts + 2.0 * rho) / (3.14159265358979 * rho * rho));
    np.rho_start = ibl_knn_rho();
    const char* e = getenv("IBL_KNN_NOGUESS");          // diagnostics, read per call: the tests compare both selections bit for bit
    np.guess = (e && atoi(e)) ? 0 : 1;
    return np;
}

template <class Factory>
static int launch_knn(ibl_reg_ctx* ctx, const BatchGrid& g, const float4* pts, const int* seg_off, int n, int q0, int q1, double radius, int max_nn,
                      const Factory& fac, int* status, hipStream_t s) {
    if (q1 <= q0) return IBL_OK;
    const float r = (float)radius, r2 = (float)(radius * radius);
    if (g.tile_base && g.n_tiles > 0) {
        ArenaMark m(ctx);                    // (released on return: later allocations are used by later kernels of the same stream)
        int *fb_list, *fb_count;
        IBL_ARENA(fb_list, int, (int64_t)n + 64);
        IBL_ARENA(fb_count, int, 64);
        IBL_HIP_CHECK(hipMemsetAsync(fb_count, 0, sizeof(int), s));
        const NeedPop np = need_pop_table(max_nn, g.ts);
        // LDS budget of the staged cube.  The kernel is bound by its resident workgroups: padded to ONE workgroup per CU it ran the
        // 100-neighbour search in 10.6 ms against 5.4 ms with two (lab: -DKNN_LAB_PAD_LDS).  Round 3, final form for the consumers that read
        // geometry only (Factory::WIDE): a PACKED tile (the original index rides in the staged point's w: 16 instead of 20 B per candidate)
        // of 1 600 candidates + the four waves' scratch = 53.5 KB and a 168-register launch bound -> THREE four-wave workgroups per CU:
        // feature call 8.49 -> 7.0 ms although 0.4 % (not 0.1 %) of the queries now overflow a cube and join the grid walk.  (Six waves per
        // workgroup on a 2 528-candidate packed tile -- also twelve waves per CU -- was slower, 7.13 vs 5.27 ms for the search: the
        // per-tile set-up and barriers are then shared by fewer queries per wave.)  The colour-gradient
        // search reads the staged intensity and keeps the 20-byte tile (1 024 candidates, three workgroups per CU as well).
        bool packed = false;
        if constexpr (Factory::WIDE) {
            if (g.ts == 2) {
                packed = true;
#ifdef KNN_LAB_NW          // lab: KNN_LAB_NW waves per workgroup, KNN_LAB_OCC workgroups per CU
                hipLaunchKernelGGL((ibl_knn_tile_kernel<2, KNN_TILE_CAP3, Factory, KNN_LAB_NW, true, KNN_LAB_OCC>), dim3(g.n_tiles), dim3(64 * KNN_LAB_NW), 0, s, g, r, r2, max_nn, np, fac, q0, q1, fb_list, fb_count);
#else
                hipLaunchKernelGGL((ibl_knn_tile_kernel<2, KNN_TILE_CAP3, Factory, 4, true, 3>), dim3(g.n_tiles), dim3(256), 0, s, g, r, r2, max_nn, np, fac, q0, q1, fb_list, fb_count);
#endif
            }
        }
        if (packed) {}
        else
        if (g.ts == 2) hipLaunchKernelGGL((ibl_knn_tile_kernel<2, 2560, Factory>), dim3(g.n_tiles), dim3(256), 0, s, g, r, r2, max_nn, np, fac, q0, q1, fb_list, fb_count);
        else if (g.ts == 4) hipLaunchKernelGGL((ibl_knn_tile_kernel<4, 1024, Factory>), dim3(g.n_tiles), dim3(256), 0, s, g, r, r2, max_nn, np, fac, q0, q1, fb_list, fb_count);
        else return ibl_set_error(IBL_ERR_INTERNAL, "k-NN tiles of %d^3 cells are not built", g.ts);
        IBL_LAUNCH_CHECK();
#ifdef KNN_LAB_CLK
        { unsigned long long h[16]; (void)hipStreamSynchronize(s); (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(knn_lab_clk), sizeof(h));
          const double nt = (double)std::max<unsigned long long>(h[6], 1);
          fprintf(stderr, "[knn-clk] k=%d ts=%d tiles with queries %llu (%.1f queries each): clocks (100 MHz ticks) per tile: query rows %.0f, rho loop %.0f, staging %.0f, "
                  "pilot %.0f, queries(wave 0) %.0f, flush %.0f\n", max_nn, g.ts, h[6], (double)h[7] / nt, h[0] / nt, h[1] / nt, h[2] / nt, h[3] / nt, h[4] / nt, h[5] / nt);
          const double nq0 = (double)h[7] / 4.0;         // queries of wave 0
          fprintf(stderr, "[knn-clk]   per query of wave 0: pass 1 %.0f, threshold %.0f, pass 2 %.0f, boundary rank %.0f, consumer finish %.0f\n", h[8] / nq0, h[9] / nq0,
                  h[10] / nq0, h[11] / nq0, h[12] / nq0);
          fprintf(stderr, "[knn-clk]   consumer finish: index sort %.0f, list write %.0f, inner selection %.0f\n", h[13] / nq0, h[14] / nq0, h[15] / nq0);
          unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(knn_lab_clk), z, sizeof(z)); }
#endif
#ifdef KNN_LAB_STATS
        { int h[8]; (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(knn_lab_stats), sizeof(h)); fprintf(stderr, "[knn-lab] k=%d queries %d from_list %d select_all %d guess_low %d overflow %d\n", max_nn, h[0], h[1], h[2], h[3], h[4]); int z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(knn_lab_stats), z, sizeof(z)); }
#endif
        const int blocks = std::max(1, std::min(2048, (q1 - q0 + 3) / 4));
        hipLaunchKernelGGL((ibl_knn_list_kernel<Factory>), dim3(blocks), dim3(256), 0, s, g, seg_off, r, r2, max_nn, fac, fb_list, fb_count, status);
        if (getenv("IBL_KNN_DEBUG")) {                      // diagnostics: how many queries the staged cubes could not answer
            int h = 0;
            (void)hipMemcpyAsync(&h, fb_count, sizeof(int), hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            fprintf(stderr, "[knn] r=%.3f k=%d ts=%d tiles=%d queries=%d fallback=%d (%.1f %%)\n", radius, max_nn, g.ts, g.n_tiles, q1 - q0, h, 100.0 * h / (q1 - q0));
        }
    } else {
        hipLaunchKernelGGL((ibl_knn_query_kernel<Factory>), dim3((q1 - q0 + 3) / 4), dim3(256), 0, s, g, pts, seg_off, r, r2, max_nn, fac, q0, q1, status);
    }
    IBL_LAUNCH_CHECK();
    return IBL_OK;
}

int ibl_launch_normals(ibl_reg_ctx* ctx, const BatchGrid& g, const float4* pts, const int* seg_off, int n, double radius, int max_nn, float4* normals,
                       int* status, hipStream_t s) {
    if (n <= 0) return IBL_OK;
    if (max_nn > 256) return ibl_set_error(IBL_ERR_UNSUPPORTED, "normals: max_nn %d > 256", max_nn);
    void* tok;
    ibl_prof_begin(IBL_PROF_NORMALS, 24.0 * (double)n, s, &tok);
    const int st = launch_knn(ctx, g, pts, seg_off, n, 0, n, radius, max_nn, NormalFactory{normals}, status, s);
    ibl_prof_end(tok, s);
    return st;
}

int ibl_launch_fpfh(ibl_reg_ctx* ctx, const BatchGrid& g, const float4* pts, const float4* normals, const int* seg_off, int n, double radius, int max_nn,
                    unsigned char* spfh, int* nbr_idx, float* nbr_d2, int* nbr_cnt, float* fpfh, int matching_order, int* status,
                    hipStream_t s) {
    if (n <= 0) return IBL_OK;
    if (max_nn > 256) return ibl_set_error(IBL_ERR_UNSUPPORTED, "fpfh: max_nn %d > 256 (SPFH histograms are stored as bytes)", max_nn);
    void* tok;
    ibl_prof_begin(IBL_PROF_SPFH, 156.0 * (double)n, s, &tok);
    const int st = launch_knn(ctx, g, pts, seg_off, n, 0, n, radius, max_nn, SpfhFactory{normals, spfh, nbr_idx, nbr_d2, nbr_cnt, max_nn}, status, s);
    ibl_prof_end(tok, s);
    if (st) return st;
    hipLaunchKernelGGL(ibl_fpfh_kernel, dim3((n + FPFH_Q - 1) / FPFH_Q), dim3(256), 0, s, spfh, nbr_idx, nbr_d2, nbr_cnt, max_nn, n, matching_order, fpfh);
    IBL_LAUNCH_CHECK();
    return IBL_OK;
}

// Normals of every point from its neighbour list and the 128-bit mask of the list slots that are the normal's own neighbours (written by
// ListNormalConsumer): one lane per point, the moment sums in list order = ascending original index, exactly the arithmetic of
// normal_solve -- a point's normal does not depend on which kernel found its neighbours.
__global__ __launch_bounds__(256) void ibl_normals_from_mask_kernel(const float4* __restrict__ pts, const int* __restrict__ nbr_idx, int K,
                                                                    const unsigned* __restrict__ nrm_mask, int n, float4* __restrict__ normals) {
    const int qi = IBL_XCD_BLOCK(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (qi >= n) return;
    const uint4 m4 = *reinterpret_cast<const uint4*>(nrm_mask + 4 * (int64_t)qi);
    const unsigned w[4] = {m4.x, m4.y, m4.z, m4.w};
    double c[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) c[t] = 0.0;
    int k = 0;
    // eight neighbours per step: their list entries are read together, then their points (a lane's walk was two dependent loads per
    // neighbour, one neighbour after the other -- the kernel waited 76 % of its wave time); the sums keep the list order
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        unsigned bits = w[u];
        while (bits) {
            int t[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                t[v] = -1;
                if (bits) { t[v] = 32 * u + __ffs((int)bits) - 1; bits &= bits - 1u; }
            }
            int j[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) j[v] = nbr_idx[(int64_t)qi * K + (t[v] >= 0 ? t[v] : 0)];
            float4 p[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) p[v] = pts[j[v]];
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                if (t[v] < 0) continue;
                const double x = p[v].x, y = p[v].y, z = p[v].z;
                c[0] += x; c[1] += y; c[2] += z;
                c[3] += x * x; c[4] += x * y; c[5] += x * z; c[6] += y * y; c[7] += y * z; c[8] += z * z;
                ++k;
            }
        }
    }
    normal_from_moments(c, k, qi, normals);
}

// SPFH histograms from stored neighbour lists (after the fused search above has written lists and normals): wave per point.
// FAST: the bins of a pair come from pair_bins_f32; the pairs it cannot decide are collected per point and appended, with ONE atomic per
// point that has any, to one of SPFH_NQ queues (consecutive workgroups use different queues: a single counter serialises ~10^6 returning
// atomics per batch at the L2 -- measured, it doubled the stage), then evaluated in fp64 by ibl_spfh_queue_kernel, which adds their three
// counts to the stored byte histograms.
#define SPFH_NQ 256
#define SPFH_QSTRIDE 64                 // ints between two queue counters (256 B: different L2 channels)
template <bool FAST>
__global__ __launch_bounds__(256) void ibl_spfh_lists_kernel(const float4* __restrict__ pts, const float4* __restrict__ normals,
                                                             const int* __restrict__ nbr_idx, const int* __restrict__ nbr_cnt, int K, int n,
                                                             unsigned char* __restrict__ spfh_cnt, int2* __restrict__ queue, int* __restrict__ q_count,
                                                             int q_cap) {
    __shared__ int hists[4][36];
    __shared__ int stash[4][FAST ? 128 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the fp64 form doubles as the fast form's overflow path: launched behind it with the queues' overflow flag (the word after the last
    // counter), it returns at once unless a queue overflowed, and then rewrites every histogram (grid-stride over the points)
    if (!FAST && q_count != nullptr && q_count[SPFH_NQ * SPFH_QSTRIDE] == 0) return;
    int* hist = hists[wave];
  for (int qi = IBL_XCD_BLOCK(blockIdx.x, gridDim.x) * 4 + wave; qi < n; qi += gridDim.x * 4) {
    if (lane < 36) hist[lane] = 0;
    wave_lds_sync();
    const int k = nbr_cnt[qi];
    const float4 q = pts[qi], qn = normals[qi];
    int n_unsure = 0;
    for (int t0 = 0; t0 < k; t0 += 64) {
        const int t = t0 + lane;
        const int jo = t < k ? nbr_idx[(int64_t)qi * K + t] : qi;
        bool unsure = false;
        if (jo != qi) {
            if (FAST) {
                int b0, b1, b2;
                if (pair_bins_f32(q, qn, pts[jo], normals[jo], &b0, &b1, &b2)) {
                    atomicAdd(&hist[b0], 1);
                    atomicAdd(&hist[11 + b1], 1);
                    atomicAdd(&hist[22 + b2], 1);
                } else {
                    unsure = true;
                }
            } else {
                double f[3];
                pair_features_d(q, qn, pts[jo], normals[jo], f);
                atomicAdd(&hist[clamp_bin11((int)f[0])], 1);
                atomicAdd(&hist[11 + clamp_bin11((int)floor(11 * (f[1] + 1.0) * 0.5))], 1);
                atomicAdd(&hist[22 + clamp_bin11((int)floor(11 * (f[2] + 1.0) * 0.5))], 1);
            }
        }
        if (FAST) {
            const unsigned long long m = __ballot(unsure);
            if (unsure) stash[wave][(n_unsure + __popcll(m & ((1ull << lane) - 1ull))) & 127] = jo;      // (K <= 128: ibl_normals_fpfh_fusable)
            n_unsure += __popcll(m);
        }
    }
    wave_lds_sync();
    if (lane < 36) spfh_cnt[(int64_t)qi * 36 + lane] = lane < 33 ? (unsigned char)hist[lane] : (unsigned char)0;
    if (FAST && n_unsure > 0) {
        const int qsel = blockIdx.x & (SPFH_NQ - 1);
        int base = 0;
        if (lane == 0) base = atomicAdd(&q_count[qsel * SPFH_QSTRIDE], n_unsure);
        base = __shfl(base, 0, 64);
        if (base + n_unsure <= q_cap) {
            for (int e = lane; e < n_unsure; e += 64) queue[(int64_t)qsel * q_cap + base + e] = make_int2(qi, stash[wave][e]);
        } else if (lane == 0) {
            q_count[SPFH_NQ * SPFH_QSTRIDE] = 1;           // this queue is full: the gated fp64 launch redoes the batch
        }
    }
    wave_lds_sync();
  }
}

// the undecided pairs of the fast kernel, in fp64: three byte counters of the point's stored histogram go up by one each (a 32-bit atomic on
// the word that holds the byte: a counter never exceeds the 100 neighbours of a point, so no carry crosses into the next byte).
// Grid: SPFH_NQ x 4 workgroups, four per queue.
__global__ __launch_bounds__(256) void ibl_spfh_queue_kernel(const float4* __restrict__ pts, const float4* __restrict__ normals,
                                                             const int2* __restrict__ queue, const int* __restrict__ q_count, int q_cap,
                                                             unsigned char* __restrict__ spfh_cnt) {
    if (q_count[SPFH_NQ * SPFH_QSTRIDE] != 0) return;            // (overflow: everything is redone)
    const int qsel = blockIdx.x & (SPFH_NQ - 1), part = blockIdx.x / SPFH_NQ, parts = gridDim.x / SPFH_NQ;
    const int total = min(q_count[qsel * SPFH_QSTRIDE], q_cap);
    for (int e = part * 256 + threadIdx.x; e < total; e += parts * 256) {
        const int2 pr = queue[(int64_t)qsel * q_cap + e];
        double f[3];
        pair_features_d(pts[pr.x], normals[pr.x], pts[pr.y], normals[pr.y], f);
        const int b[3] = {clamp_bin11((int)f[0]), 11 + clamp_bin11((int)floor(11 * (f[1] + 1.0) * 0.5)), 22 + clamp_bin11((int)floor(11 * (f[2] + 1.0) * 0.5))};
        unsigned* words = reinterpret_cast<unsigned*>(spfh_cnt + (int64_t)pr.x * 36);
#pragma unroll
        for (int c = 0; c < 3; ++c) atomicAdd(&words[b[c] >> 2], 1u << (8 * (b[c] & 3)));
    }
}

// normals + FPFH from ONE neighbour search (requires radius_normal <= radius_feature, max_nn_normal <= max_nn_feature and <= NP_MAXK)
bool ibl_normals_fpfh_fusable(double radius_normal, int max_nn_normal, double radius_feature, int max_nn_feature) {
    const char* e = getenv("IBL_FEAT_UNFUSED");          // diagnostics: 1 = the two separate searches (read per call: the tests compare both)
    const bool off = e && atoi(e);
    return !off && radius_normal <= radius_feature && max_nn_normal <= max_nn_feature && max_nn_normal <= NP_MAXK && max_nn_feature <= 128;   // (128-slot mask)
}
int ibl_launch_normals_fpfh(ibl_reg_ctx* ctx, const BatchGrid& g, const float4* pts, const int* seg_off, int n, double radius_normal,
                            int max_nn_normal, double radius_feature, int max_nn_feature, float4* normals, unsigned char* spfh, int* nbr_idx,
                            float* nbr_d2, int* nbr_cnt, float* fpfh, int matching_order, int* status, hipStream_t s) {
    if (n <= 0) return IBL_OK;
    ArenaMark mk(ctx);                  // (the mask is read by the kernel launched below, on the same stream)
    unsigned* nrm_mask;
    IBL_ARENA(nrm_mask, unsigned, 4 * (int64_t)n + 64);
    void* tok;
    ibl_prof_begin(IBL_PROF_SPFH, 156.0 * (double)n, s, &tok);
    const int st = launch_knn(ctx, g, pts, seg_off, n, 0, n, radius_feature, max_nn_feature,
                              ListNormalFactory{nrm_mask, nbr_idx, nbr_d2, nbr_cnt, max_nn_feature, (float)(radius_normal * radius_normal), max_nn_normal},
                              status, s);
    ibl_prof_end(tok, s);
    if (st) return st;
    hipLaunchKernelGGL(ibl_normals_from_mask_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pts, nbr_idx, max_nn_feature, nrm_mask, n, normals);
    IBL_LAUNCH_CHECK();
    {
        // fp32 bins + fp64 for the pairs next to a bin boundary (pair_bins_f32); the SPFH_NQ queues together hold 1 / 16 of all pairs
        // (measured: 0.8 % are undecided) -- a batch that overflows one is redone in fp64 by the gated third launch; IBL_SPFH_F64=1 runs
        // every pair in fp64 (the tests compare both), IBL_SPFH_QCAP=<entries per queue> shrinks the queues (the overflow test)
        const char* e64 = getenv("IBL_SPFH_F64");
        bool fast = !(e64 && atoi(e64));
        if (fast) {
            int64_t cap64 = std::min<int64_t>((int64_t)n * max_nn_feature / (16 * SPFH_NQ) + 256, (int64_t)1 << 20);
            if (const char* ec = getenv("IBL_SPFH_QCAP")) cap64 = std::max<int64_t>(1, atoll(ec));
            int2* queue; int* q_count;
            IBL_ARENA(queue, int2, cap64 * SPFH_NQ);
            IBL_ARENA(q_count, int, SPFH_NQ * SPFH_QSTRIDE + 64);
            IBL_HIP_CHECK(hipMemsetAsync(q_count, 0, sizeof(int) * (SPFH_NQ * SPFH_QSTRIDE + 1), s));
            hipLaunchKernelGGL(ibl_spfh_lists_kernel<true>, dim3((n + 3) / 4), dim3(256), 0, s, pts, normals, nbr_idx, nbr_cnt, max_nn_feature, n, spfh,
                               queue, q_count, (int)cap64);
            IBL_LAUNCH_CHECK();
            hipLaunchKernelGGL(ibl_spfh_queue_kernel, dim3(SPFH_NQ * 4), dim3(256), 0, s, pts, normals, queue, q_count, (int)cap64, spfh);
            IBL_LAUNCH_CHECK();
            hipLaunchKernelGGL(ibl_spfh_lists_kernel<false>, dim3(2048), dim3(256), 0, s, pts, normals, nbr_idx, nbr_cnt, max_nn_feature, n, spfh,
                               (int2*)nullptr, q_count, (int)cap64);
            IBL_LAUNCH_CHECK();
            if (const char* es = getenv("IBL_SPFH_STATS"); es && atoi(es)) {              // diagnostics: undecided pairs of this batch (synchronises)
                std::vector<int> cnt(SPFH_NQ * SPFH_QSTRIDE + 1);
                IBL_HIP_CHECK(hipMemcpyAsync(cnt.data(), q_count, sizeof(int) * cnt.size(), hipMemcpyDeviceToHost, s));
                IBL_HIP_CHECK(hipStreamSynchronize(s));
                long long tot = 0; int mx = 0;
                for (int i = 0; i < SPFH_NQ; ++i) { tot += cnt[i * SPFH_QSTRIDE]; mx = std::max(mx, cnt[i * SPFH_QSTRIDE]); }
                fprintf(stderr, "[ibloc] spfh: %lld undecided pairs of at most %lld; fullest queue %d of %lld; overflow %d\n", tot,
                        (long long)n * max_nn_feature, mx, (long long)cap64, cnt.back());
            }
        } else {
            hipLaunchKernelGGL(ibl_spfh_lists_kernel<false>, dim3((n + 3) / 4), dim3(256), 0, s, pts, normals, nbr_idx, nbr_cnt, max_nn_feature, n, spfh,
                               (int2*)nullptr, (int*)nullptr, 0);
            IBL_LAUNCH_CHECK();
        }
    }
    if (fpfh) {
        hipLaunchKernelGGL(ibl_fpfh_kernel, dim3((n + FPFH_Q - 1) / FPFH_Q), dim3(256), 0, s, spfh, nbr_idx, nbr_d2, nbr_cnt, max_nn_feature, n, matching_order, fpfh);
        IBL_LAUNCH_CHECK();
    }
    return IBL_OK;
}

int ibl_launch_color_grad(ibl_reg_ctx* ctx, const BatchGrid& g, const float4* pts, const float4* normals, const int* seg_off, int q0, int q1, double radius,
                          int max_nn, float4* grad, int* status, hipStream_t s) {
    if (q1 <= q0) return IBL_OK;
    if (max_nn > 256) return ibl_set_error(IBL_ERR_UNSUPPORTED, "colour gradient: max_nn %d > 256", max_nn);
    return launch_knn(ctx, g, pts, seg_off, q1, q0, q1, radius, max_nn, GradFactory{normals, grad}, status, s);
}

int ibl_launch_radius_count(const BatchGrid& g, const float4* pts, const int* seg_off, int n, double radius, int nb_points,
                            unsigned char* keep, hipStream_t s) {
    if (n <= 0) return IBL_OK;
    hipLaunchKernelGGL(ibl_radius_count_kernel, dim3((n + 255) / 256), dim3(256), 0, s, g, pts, seg_off, (float)radius,
                       (float)(radius * radius), nb_points, keep);
    IBL_LAUNCH_CHECK();
    return IBL_OK;
}
