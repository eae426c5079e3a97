// reg_grid.hip -- uniform grids over a batch of clouds (one dense grid per cloud), built on the device:
// bounding boxes -> grid dimensions -> cell ids + histogram -> exclusive scan -> stable radix sort.
// Replaces the KD-trees Open3D builds inside every estimate_normals / compute_fpfh_feature /
// registration call of utils/fpfh_register.py:86-150.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdlib>
#include <cstring>
#include <vector>

#include "reg_common.h"

// one block per segment: bounding box
__global__ __launch_bounds__(256) void ibl_bbox_kernel(const float4* __restrict__ pts, const int* __restrict__ seg_off,
                                                       float* __restrict__ bbox /* [S][6] */) {
    const int s = blockIdx.x;
    const int b = seg_off[s], e = seg_off[s + 1];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = b + threadIdx.x; i < e; i += 256) {
        const float4 p = pts[i];
        mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
        mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
    }
    __shared__ float smn[3][256], smx[3][256];
    for (int a = 0; a < 3; ++a) { smn[a][threadIdx.x] = mn[a]; smx[a][threadIdx.x] = mx[a]; }
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (threadIdx.x < st)
            for (int a = 0; a < 3; ++a) {
                smn[a][threadIdx.x] = fminf(smn[a][threadIdx.x], smn[a][threadIdx.x + st]);
                smx[a][threadIdx.x] = fmaxf(smx[a][threadIdx.x], smx[a][threadIdx.x + st]);
            }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        for (int a = 0; a < 3; ++a) {
            bbox[s * 6 + a] = e > b ? smn[a][0] : 0.0f;
            bbox[s * 6 + 3 + a] = e > b ? smx[a][0] : 0.0f;
        }
    }
}

int ibl_launch_bbox(const float4* pts, const int* seg_off_dev, int n_seg, float* bbox_dev, hipStream_t s) {
    if (n_seg <= 0) return IBL_OK;
    hipLaunchKernelGGL(ibl_bbox_kernel, dim3(n_seg), dim3(256), 0, s, pts, seg_off_dev, bbox_dev);
    IBL_LAUNCH_CHECK();
    return IBL_OK;
}

// single block of 256 threads: per-segment grid dimensions and the cell-base prefix, 256 segments per round (a serial loop over
// a few thousand job sides cost ~60 us per grid, several grids per batch)
__device__ __forceinline__ SegGrid seg_dims(const float* __restrict__ b, float cell) {
    const float ex = b[3] - b[0], ey = b[4] - b[1], ez = b[5] - b[2];
    float c = cell;
    const float emax = fmaxf(ex, fmaxf(ey, ez));
    if (emax / 128.0f > c) c = emax / 128.0f;     // bound the table: at most ~129 cells per axis
    SegGrid g;
    g.minx = b[0]; g.miny = b[1]; g.minz = b[2];
    g.inv = 1.0f / c;
    g.nx = (int)floorf(ex * g.inv) + 1;
    g.ny = (int)floorf(ey * g.inv) + 1;
    g.nz = (int)floorf(ez * g.inv) + 1;
    g.cell_base = 0;
    return g;
}

__global__ __launch_bounds__(256) void ibl_grid_dims_kernel(const float* __restrict__ bbox, int n_seg, float cell, long long max_cells,
                                                            SegGrid* __restrict__ seg, int* __restrict__ total_cells, int* __restrict__ status) {
    __shared__ long long wave_sum[4];
    __shared__ long long carry;
    __shared__ int overflow_at;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { carry = 0; overflow_at = -1; }
    __syncthreads();
    for (int base = 0; base < n_seg; base += 256) {
        const int s = base + tid;
        SegGrid g;
        long long cells = 0;
        if (s < n_seg) {
            g = seg_dims(bbox + s * 6, cell);
            cells = (long long)g.nx * g.ny * g.nz;
        }
        long long incl = cells;                      // inclusive scan inside the wave, then across the four waves
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const long long o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        long long before = carry;
        for (int w = 0; w < wave; ++w) before += wave_sum[w];
        const long long excl = before + incl - cells;
        if (s < n_seg) {
            if (excl + cells > max_cells) atomicMin(&overflow_at, s);     // handled serially below (error path)
            g.cell_base = (int)excl;
            seg[s] = g;
        }
        __syncthreads();
        if (tid == 0) carry += wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
        __syncthreads();
        if (overflow_at >= 0) break;
    }
    if (overflow_at >= 0) {
        // budget exceeded: from the first offending segment on, the serial rule (that segment and every later one that does not
        // fit collapse to one cell; the status bit tells the host)
        if (tid == 0) {
            long long base = seg[overflow_at].cell_base;
            for (int s = overflow_at; s < n_seg; ++s) {
                SegGrid g = seg_dims(bbox + s * 6, cell);
                g.cell_base = (int)base;
                base += (long long)g.nx * g.ny * g.nz;
                if (base > max_cells) {
                    atomicOr(status, IBL_ST_GRID_OVERFLOW);
                    g.nx = g.ny = g.nz = 1;                    // keep later kernels in bounds
                    base = g.cell_base + 1;
                }
                seg[s] = g;
            }
            *total_cells = (int)base;
        }
        return;
    }
    if (tid == 0) *total_cells = (int)carry;
}

__global__ __launch_bounds__(256) void ibl_cell_id_kernel(const float4* __restrict__ pts, const int* __restrict__ seg_off, int n_seg,
                                                          const SegGrid* __restrict__ seg, unsigned* __restrict__ keys,
                                                          int* __restrict__ vals, int* __restrict__ cell_count) {
    const int n = seg_off[n_seg];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int s = seg_of(seg_off, n_seg, i);
    const SegGrid g = seg[s];
    const float4 p = pts[i];
    const int cx = cell_clamp(p.x, g.minx, g.inv, g.nx), cy = cell_clamp(p.y, g.miny, g.inv, g.ny),
              cz = cell_clamp(p.z, g.minz, g.inv, g.nz);
    const int c = g.cell_base + (cz * g.ny + cy) * g.nx + cx;
    keys[i] = (unsigned)c;
    vals[i] = i;
    atomicAdd(&cell_count[c], 1);
}

__global__ __launch_bounds__(256) void ibl_gather_sorted_kernel(const float4* __restrict__ pts, const int* __restrict__ order, int n,
                                                                float4* __restrict__ sorted) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) sorted[i] = pts[order[i]];
}

// cell ids -> histogram -> exclusive scan -> stable radix sort -> gathered points, for a segment table already on the device
static int grid_fill(ibl_reg_ctx* ctx, const float4* pts, const int* seg_off_dev, int n_seg, int n, const SegGrid* seg, int h_total,
                     BatchGrid* out, hipStream_t s) {
    int* cell_count; int* cell_start; unsigned *keys, *keys_out; int *vals, *order;
    float4* sorted;
    IBL_ARENA(cell_start, int, (int64_t)h_total + 2);
    IBL_ARENA(order, int, n + 1);
    IBL_ARENA(sorted, float4, n + 1);
    out->cell_start = cell_start; out->sorted_pts = sorted; out->order = order;
    ArenaMark scratch(ctx);      // everything below is released on return
    IBL_ARENA(cell_count, int, (int64_t)h_total + 2);
    IBL_HIP_CHECK(hipMemsetAsync(cell_count, 0, sizeof(int) * (size_t)(h_total + 1), s));
    if (n == 0) {
        IBL_HIP_CHECK(hipMemsetAsync(cell_start, 0, sizeof(int) * (size_t)(h_total + 1), s));
        return IBL_OK;
    }
    IBL_ARENA(keys, unsigned, n);
    IBL_ARENA(keys_out, unsigned, n);
    IBL_ARENA(vals, int, n);
    hipLaunchKernelGGL(ibl_cell_id_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pts, seg_off_dev, n_seg, seg, keys, vals,
                       cell_count);
    IBL_LAUNCH_CHECK();
    size_t tmp_scan = 0, tmp_sort = 0;
    IBL_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_scan, cell_count, cell_start, h_total + 1, s));
    int end_bit = 1;
    while ((1ll << end_bit) < (long long)h_total + 1 && end_bit < 32) ++end_bit;
    IBL_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_sort, keys, keys_out, vals, order, n, 0, end_bit, s));
    unsigned char* tmp;
    IBL_ARENA(tmp, unsigned char, (int64_t)(tmp_scan > tmp_sort ? tmp_scan : tmp_sort) + 256);
    IBL_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_scan, cell_count, cell_start, h_total + 1, s));
    IBL_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_sort, keys, keys_out, vals, order, n, 0, end_bit, s));
    hipLaunchKernelGGL(ibl_gather_sorted_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pts, order, n, sorted);
    IBL_LAUNCH_CHECK();
    return IBL_OK;
}

int ibl_build_batch_grid(ibl_reg_ctx* ctx, const float4* pts, const int* seg_off_dev, const int* seg_off_host, int n_seg,
                         float cell, int64_t max_cells, BatchGrid* out, hipStream_t s) {
    const int n = seg_off_host[n_seg];
    float* bbox; SegGrid* seg; int* total;
    IBL_ARENA(bbox, float, (int64_t)n_seg * 6 + 6);
    IBL_ARENA(seg, SegGrid, n_seg + 1);
    IBL_ARENA(total, int, 4);
    out->seg = seg; out->cell_start = nullptr; out->sorted_pts = nullptr; out->order = nullptr; out->n_seg = n_seg;
    out->tile_base = nullptr; out->tile_seg = nullptr; out->n_tiles = 0; out->ts = 0;
    if (n_seg == 0) return IBL_OK;
    hipLaunchKernelGGL(ibl_bbox_kernel, dim3(n_seg), dim3(256), 0, s, pts, seg_off_dev, bbox);
    IBL_LAUNCH_CHECK();
    hipLaunchKernelGGL(ibl_grid_dims_kernel, dim3(1), dim3(256), 0, s, bbox, n_seg, cell, (long long)max_cells, seg, total,
                       ctx->d_status);
    IBL_LAUNCH_CHECK();
    int h_total = 0;
    IBL_HIP_CHECK(hipMemcpyAsync(&h_total, total, sizeof(int), hipMemcpyDeviceToHost, s));
    IBL_HIP_CHECK(hipStreamSynchronize(s));     // the table size decides the scan length (one small read-back per grid)
    if (h_total <= 0 || h_total > max_cells) return ibl_set_error(IBL_ERR_OVERFLOW, "grid: %d cells exceed the budget %lld", h_total, (long long)max_cells);
    return grid_fill(ctx, pts, seg_off_dev, n_seg, n, seg, h_total, out, s);
}

int ibl_build_batch_grid_bounded(ibl_reg_ctx* ctx, const float4* pts, const int* seg_off_dev, const int* seg_off_host, int n_seg,
                                 float cell, int64_t cells_bound, BatchGrid* out, hipStream_t s) {
    const int n = seg_off_host[n_seg];
    float* bbox; SegGrid* seg; int* total;
    IBL_ARENA(bbox, float, (int64_t)n_seg * 6 + 6);
    IBL_ARENA(seg, SegGrid, n_seg + 1);
    IBL_ARENA(total, int, 4);
    out->seg = seg; out->cell_start = nullptr; out->sorted_pts = nullptr; out->order = nullptr; out->n_seg = n_seg;
    out->tile_base = nullptr; out->tile_seg = nullptr; out->n_tiles = 0; out->ts = 0;
    if (n_seg == 0) return IBL_OK;
    if (cells_bound <= 0 || cells_bound > 0x7fff0000ll) return ibl_set_error(IBL_ERR_OVERFLOW, "grid: cell bound %lld out of range", (long long)cells_bound);
    hipLaunchKernelGGL(ibl_bbox_kernel, dim3(n_seg), dim3(256), 0, s, pts, seg_off_dev, bbox);
    IBL_LAUNCH_CHECK();
    hipLaunchKernelGGL(ibl_grid_dims_kernel, dim3(1), dim3(256), 0, s, bbox, n_seg, cell, (long long)cells_bound, seg, total, ctx->d_status);
    IBL_LAUNCH_CHECK();
    return grid_fill(ctx, pts, seg_off_dev, n_seg, n, seg, (int)cells_bound, out, s);
}

// Defaults from sweeps on the detections of a bench step (tools/lab_knn_grid.sh, 1.04 M points in 210 clouds, ms per feature call;
// the results do not depend on the knobs -- a tile proves its queries or hands them to the grid walk).  With the 100-neighbour search
// on 1 600-candidate packed tiles, three workgroups per CU (reg_knn.hip):
//   rho 2: safety 0.6 7.64 | 0.7 7.41 | 0.8 7.44 | 1.0 7.80 (9.7 % of the queries overflow their cube)
//   rho 3: safety 0.7 7.89 | 0.8 7.33 | 0.9 7.16 | 1.0 7.14 | 1.1 7.02 | 1.25 7.06 | 1.5 7.30      rho 4: 1.25 7.43
// (on the 2 560-candidate tiles, two workgroups per CU, the best was rho 2 / 0.8: 8.53 against 9.05 at 3 / 1.0.)
double ibl_knn_safety() {
    static const double v = [] { const char* e = getenv("IBL_KNN_SAFETY"); return e ? atof(e) : 1.1; }();
    return v;
}
int ibl_knn_rho() {
    static const int v = [] { const char* e = getenv("IBL_KNN_RHO"); const int r = e ? atoi(e) : 3; return r < 1 ? 1 : (r > 4 ? 4 : r); }();
    return v;
}

int ibl_stage_upload(ibl_reg_ctx* ctx, void* dst_dev, const void* src_host, int64_t bytes, hipStream_t s) {
    if (bytes <= 0) return IBL_OK;
    const int64_t need = ibl_align_up(bytes, 64);
    if (ctx->pin && ctx->pin_used + need <= ctx->pin_size) {
        unsigned char* p = ctx->pin + ctx->pin_used;
        ctx->pin_used += need;
        memcpy(p, src_host, (size_t)bytes);
        IBL_HIP_CHECK(hipMemcpyAsync(dst_dev, p, (size_t)bytes, hipMemcpyHostToDevice, s));
        return IBL_OK;
    }
    IBL_HIP_CHECK(hipMemcpyAsync(dst_dev, src_host, (size_t)bytes, hipMemcpyHostToDevice, s));
    IBL_HIP_CHECK(hipStreamSynchronize(s));      // staging exhausted: the caller's buffer may die after this call
    return IBL_OK;
}

// tile -> segment table: a workgroup of the tile search found its segment by a binary search over tile_base, eight dependent loads
// before it could do anything else
__global__ __launch_bounds__(256) void ibl_tile_seg_kernel(const int* __restrict__ tile_base, int* __restrict__ tile_seg) {
    const int sgi = blockIdx.x;
    for (int t = tile_base[sgi] + threadIdx.x; t < tile_base[sgi + 1]; t += 256) tile_seg[t] = sgi;
}

int ibl_build_tile_grid(ibl_reg_ctx* ctx, const float4* pts, const int* seg_off_dev, const int* seg_off_host, int n_seg,
                        const float* bbox_host, double radius, int max_nn, int ts, int64_t max_cells, BatchGrid* out, hipStream_t s) {
    const int n = seg_off_host[n_seg];
    SegGrid* seg; int* tile_base;
    IBL_ARENA(seg, SegGrid, n_seg + 1);
    IBL_ARENA(tile_base, int, n_seg + 2);
    out->seg = seg; out->cell_start = nullptr; out->sorted_pts = nullptr; out->order = nullptr; out->n_seg = n_seg;
    out->tile_base = tile_base; out->tile_seg = nullptr; out->n_tiles = 0; out->ts = ts;
    if (n_seg == 0) return IBL_OK;
    std::vector<SegGrid> h(n_seg);
    std::vector<int> tb(n_seg + 1, 0);
    const double safety = ibl_knn_safety(), rho_t = (double)ibl_knn_rho();
    long long cells = 0, tiles = 0;
    for (int sgi = 0; sgi < n_seg; ++sgi) {
        const float* b = bbox_host + 6 * (size_t)sgi;
        const float ex = b[3] - b[0], ey = b[4] - b[1], ez = b[5] - b[2];
        const int cnt = seg_off_host[sgi + 1] - seg_off_host[sgi];
        // mean surface density from the faces of the box; a ball of `rho` cells should hold safety x max_nn points
        const double area = 2.0 * ((double)ex * ey + (double)ey * ez + (double)ez * ex);
        double c = radius;
        if (cnt > 0 && area > 0) {
            const double dens = (double)cnt / area;
            c = sqrt(safety * (double)max_nn / (3.141592653589793 * rho_t * rho_t * dens));
        }
        if (c > radius) c = radius;
        if (c < radius / 12.0) c = radius / 12.0;
        float cf = (float)c;
        const float emax = fmaxf(ex, fmaxf(ey, ez));
        if (emax / 128.0f > cf) cf = emax / 128.0f;     // bound the table: at most ~129 cells per axis
        SegGrid g;
        g.minx = b[0]; g.miny = b[1]; g.minz = b[2];
        g.inv = 1.0f / cf;
        g.nx = (int)floorf(ex * g.inv) + 1;
        g.ny = (int)floorf(ey * g.inv) + 1;
        g.nz = (int)floorf(ez * g.inv) + 1;
        if (cells + (long long)g.nx * g.ny * g.nz > max_cells) { g.nx = g.ny = g.nz = 1; }      // budget: one cell (slow, correct)
        g.cell_base = (int)cells;
        cells += (long long)g.nx * g.ny * g.nz;
        h[sgi] = g;
        tb[sgi] = (int)tiles;
        tiles += (long long)((g.nx + ts - 1) / ts) * ((g.ny + ts - 1) / ts) * ((g.nz + ts - 1) / ts);
        if (tiles > 0x7fff0000ll) return ibl_set_error(IBL_ERR_OVERFLOW, "tile grid: too many tiles");
    }
    tb[n_seg] = (int)tiles;
    out->n_tiles = (int)tiles;
    int st = ibl_stage_upload(ctx, seg, h.data(), sizeof(SegGrid) * (int64_t)n_seg, s);
    if (st) return st;
    st = ibl_stage_upload(ctx, tile_base, tb.data(), sizeof(int) * (int64_t)(n_seg + 1), s);
    if (st) return st;
    if (tiles > 0) {
        int* tile_seg;
        IBL_ARENA(tile_seg, int, tiles + 1);
        hipLaunchKernelGGL(ibl_tile_seg_kernel, dim3(n_seg), dim3(256), 0, s, tile_base, tile_seg);
        IBL_LAUNCH_CHECK();
        out->tile_seg = tile_seg;
    }
    return grid_fill(ctx, pts, seg_off_dev, n_seg, n, seg, (int)cells, out, s);
}
