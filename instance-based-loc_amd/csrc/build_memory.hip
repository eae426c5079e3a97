// build_memory.hip -- kernels of the memory build / consolidation step: exact voxel down-sampling of object
// clouds and DBSCAN labels of concatenated clouds.  The build side of the reference keeps Open3D's double-precision clouds, so
// both entry points read and write fp64 and reproduce the reference's arithmetic order:
//   ibl_voxel_downsample_batch  <- utils/depth_utils.py:211-265 (python dict keyed by floor(p / voxel); np.mean of each voxel's
//                                  rows = running sum in input order / count; voxels in order of first occurrence), called per
//                                  object by ObjectInfo.downsample (object_info.py:95-97) <- downsample_all_objects (object_memory.py:258)
//   ibl_dbscan_batch            <- open3d PointCloud.cluster_dbscan(eps, min_points) as called at object_memory.py:305 and :631:
//                                  neighbours = points with squared distance < eps^2 (self included), core = at least min_points
//                                  neighbours, clusters numbered in the order a sequential scan meets their first core point, a
//                                  border point takes the lowest-numbered cluster among its core neighbours, the rest is -1
#pragma clang fp contract(off)
#include <hip/hip_runtime.h>

#include <hipcub/hipcub.hpp>

#include <climits>
#include <vector>

#include "ibloc.h"
#include "reg_common.h"

namespace {

int check_offsets(const int32_t* off, int n, const char* who) {
    if (n < 0 || !off || off[0] != 0) return ibl_set_error(IBL_ERR_ARG, "%s: offsets must start at 0", who);
    for (int i = 0; i < n; ++i)
        if (off[i + 1] < off[i]) return ibl_set_error(IBL_ERR_ARG, "%s: offsets must be non-decreasing", who);
    return IBL_OK;
}

// ------------------------------------------------------------------------------------------------
// voxel down-sampling
// ------------------------------------------------------------------------------------------------
// one block per object: voxel indices of its points and their range
__global__ __launch_bounds__(256) void ibl_vox_index_kernel(const double* __restrict__ pts, const int* __restrict__ seg_off, double voxel,
                                                            long long* __restrict__ vidx, long long* __restrict__ seg_rng /* [S][6] */) {
    const int s = blockIdx.x, b = seg_off[s], e = seg_off[s + 1];
    long long mn[3] = {LLONG_MAX, LLONG_MAX, LLONG_MAX}, mx[3] = {LLONG_MIN, LLONG_MIN, LLONG_MIN};
    for (int i = b + threadIdx.x; i < e; i += 256)
        for (int a = 0; a < 3; ++a) {
            const long long v = (long long)floor(pts[3 * (int64_t)i + a] / voxel);      // np.floor(points / voxel_size).astype(np.int64)
            vidx[3 * (int64_t)i + a] = v;
            mn[a] = v < mn[a] ? v : mn[a];
            mx[a] = v > mx[a] ? v : mx[a];
        }
    __shared__ long long smn[3][256], smx[3][256];
    for (int a = 0; a < 3; ++a) { smn[a][threadIdx.x] = mn[a]; smx[a][threadIdx.x] = mx[a]; }
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (threadIdx.x < st)
            for (int a = 0; a < 3; ++a) {
                const long long o = smn[a][threadIdx.x + st], p = smx[a][threadIdx.x + st];
                if (o < smn[a][threadIdx.x]) smn[a][threadIdx.x] = o;
                if (p > smx[a][threadIdx.x]) smx[a][threadIdx.x] = p;
            }
        __syncthreads();
    }
    if (threadIdx.x == 0)
        for (int a = 0; a < 3; ++a) {
            seg_rng[s * 6 + a] = e > b ? smn[a][0] : 0;
            seg_rng[s * 6 + 3 + a] = e > b ? smx[a][0] : 0;
        }
}

// key = object (16 bits) | voxel index relative to the object's minimum (3 x 16 bits): equal keys <=> same voxel of the same object
__global__ __launch_bounds__(256) void ibl_vox_key_kernel(const long long* __restrict__ vidx, const long long* __restrict__ seg_rng,
                                                          const int* __restrict__ seg_off, int n_seg, int n,
                                                          unsigned long long* __restrict__ keys, unsigned* __restrict__ vals) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int s = seg_of(seg_off, n_seg, i);
    const unsigned long long x = (unsigned long long)(vidx[3 * (int64_t)i] - seg_rng[s * 6]), y = (unsigned long long)(vidx[3 * (int64_t)i + 1] - seg_rng[s * 6 + 1]),
                             z = (unsigned long long)(vidx[3 * (int64_t)i + 2] - seg_rng[s * 6 + 2]);
    keys[i] = ((unsigned long long)s << 48) | (x << 32) | (y << 16) | z;
    vals[i] = (unsigned)i;
}

__global__ __launch_bounds__(256) void ibl_vox_head_kernel(const unsigned long long* __restrict__ keys, int n, int* __restrict__ head) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

// start position of every voxel in the sorted point list and the original index of its first point (the sort is stable, so a
// voxel's points appear in input order)
__global__ __launch_bounds__(256) void ibl_vox_start_kernel(const int* __restrict__ head, const int* __restrict__ vox_of, const unsigned* __restrict__ idx_sorted,
                                                            int n, int n_vox, int* __restrict__ start, unsigned* __restrict__ first_idx, unsigned* __restrict__ vox_id) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && head[i]) {
        const int v = vox_of[i];          // exclusive scan of head = voxel number of a head position
        start[v] = i;
        first_idx[v] = idx_sorted[i];
        vox_id[v] = (unsigned)v;
    }
    if (i == 0) start[n_vox] = n;
}

// one thread per output voxel (rank r in first-occurrence order): running fp64 sums in input order, then one division
__global__ __launch_bounds__(256) void ibl_vox_mean_kernel(const double* __restrict__ pts, const double* __restrict__ cols, const unsigned* __restrict__ idx_sorted,
                                                           const int* __restrict__ start, const unsigned* __restrict__ vox_by_rank, int n_vox,
                                                           double* __restrict__ out_pts, double* __restrict__ out_cols, int* __restrict__ out_cnt) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_vox) return;
    const int v = (int)vox_by_rank[r];
    const int b = start[v], e = start[v + 1];
    double sp[3], sc[3] = {0.0, 0.0, 0.0};
    {
        const int64_t i = idx_sorted[b];          // the reduction starts from the first row (no 0.0 + x)
        for (int a = 0; a < 3; ++a) sp[a] = pts[3 * i + a];
        if (cols)
            for (int a = 0; a < 3; ++a) sc[a] = cols[3 * i + a];
    }
    for (int t = b + 1; t < e; ++t) {
        const int64_t i = idx_sorted[t];
        for (int a = 0; a < 3; ++a) sp[a] += pts[3 * i + a];
        if (cols)
            for (int a = 0; a < 3; ++a) sc[a] += cols[3 * i + a];
    }
    const double cnt = (double)(e - b);
    for (int a = 0; a < 3; ++a) out_pts[3 * (int64_t)r + a] = sp[a] / cnt;
    if (cols)
        for (int a = 0; a < 3; ++a) out_cols[3 * (int64_t)r + a] = sc[a] / cnt;
    if (out_cnt) out_cnt[r] = e - b;
}

// out_off[s] = number of voxels whose first point lies before object s (first indices are sorted ascending)
__global__ void ibl_vox_offsets_kernel(const unsigned* __restrict__ first_sorted, int n_vox, const int* __restrict__ seg_off, int n_seg, int* __restrict__ out_off) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n_seg) return;
    const unsigned target = (unsigned)seg_off[s];
    int lo = 0, hi = n_vox;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (first_sorted[mid] < target) lo = mid + 1; else hi = mid;
    }
    out_off[s] = lo;
}

// ------------------------------------------------------------------------------------------------
// DBSCAN
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ibl_db_to_float_kernel(const double* __restrict__ pts, int n, float4* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = make_float4((float)pts[3 * (int64_t)i], (float)pts[3 * (int64_t)i + 1], (float)pts[3 * (int64_t)i + 2], 0.f);
}

__global__ __launch_bounds__(256) void ibl_db_gather_kernel(const double* __restrict__ pts, const int* __restrict__ order, int n, double* __restrict__ sorted) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const int64_t i = order[t];
    sorted[3 * (int64_t)t] = pts[3 * i]; sorted[3 * (int64_t)t + 1] = pts[3 * i + 1]; sorted[3 * (int64_t)t + 2] = pts[3 * i + 2];
}

// Visits every point of group `grp` whose squared distance to q is < eps2 (fp64, ((dx^2 + dy^2) + dz^2) like the KD-tree's metric).
// The grid was binned on the fp32-rounded coordinates with a monotone cell function, and rounding to fp32 is monotone too, so
// a point within eps of q along an axis lies in a cell between those of fl(q - eps) and fl(q + eps).  f(t) returns false to stop.
template <typename F>
__device__ __forceinline__ void db_scan(const BatchGrid& g, const double* __restrict__ spts, int grp, double qx, double qy, double qz, double eps,
                                        double eps2, F f) {
    const SegGrid sg = g.seg[grp];
    const int x0 = cell_clamp((float)(qx - eps), sg.minx, sg.inv, sg.nx), x1 = cell_clamp((float)(qx + eps), sg.minx, sg.inv, sg.nx);
    const int y0 = cell_clamp((float)(qy - eps), sg.miny, sg.inv, sg.ny), y1 = cell_clamp((float)(qy + eps), sg.miny, sg.inv, sg.ny);
    const int z0 = cell_clamp((float)(qz - eps), sg.minz, sg.inv, sg.nz), z1 = cell_clamp((float)(qz + eps), sg.minz, sg.inv, sg.nz);
    for (int z = z0; z <= z1; ++z)
        for (int y = y0; y <= y1; ++y) {
            const int row = sg.cell_base + (z * sg.ny + y) * sg.nx;
            const int b = g.cell_start[row + x0], e = g.cell_start[row + x1 + 1];
            for (int t = b; t < e; ++t) {
                const double dx = qx - spts[3 * (int64_t)t], dy = qy - spts[3 * (int64_t)t + 1], dz = qz - spts[3 * (int64_t)t + 2];
                double d2 = dx * dx;
                d2 += dy * dy;
                d2 += dz * dz;
                if (d2 < eps2)
                    if (!f(t)) return;
            }
        }
}

// all kernels below work on sorted positions t (cell order); order[t] is the original index
__global__ __launch_bounds__(256) void ibl_db_core_kernel(BatchGrid g, const double* __restrict__ spts, const int* __restrict__ grp_off, int n_grp, int n,
                                                          double eps, double eps2, int min_points, unsigned char* __restrict__ core, int* __restrict__ parent) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const int grp = seg_of(grp_off, n_grp, t);
    int cnt = 0;
    db_scan(g, spts, grp, spts[3 * (int64_t)t], spts[3 * (int64_t)t + 1], spts[3 * (int64_t)t + 2], eps, eps2, [&](int) { return ++cnt < min_points; });
    core[t] = cnt >= min_points ? 1 : 0;
    parent[t] = t;
}

__device__ __forceinline__ int db_find(int* parent, int x) {
    while (true) {
        const int p = parent[x];
        if (p == x) return x;
        const int gp = parent[p];
        if (gp != p) atomicCAS(&parent[x], p, gp);      // path halving; losing the race is harmless
        x = p;
    }
}

__device__ __forceinline__ void db_unite(int* parent, int a, int b) {
    while (true) {
        a = db_find(parent, a);
        b = db_find(parent, b);
        if (a == b) return;
        if (a < b) { const int tmp = a; a = b; b = tmp; }      // the larger root hangs under the smaller one
        if (atomicCAS(&parent[a], a, b) == a) return;
    }
}

__global__ __launch_bounds__(256) void ibl_db_union_kernel(BatchGrid g, const double* __restrict__ spts, const int* __restrict__ grp_off, int n_grp, int n,
                                                           double eps, double eps2, const unsigned char* __restrict__ core, int* __restrict__ parent) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n || !core[t]) return;
    const int grp = seg_of(grp_off, n_grp, t);
    db_scan(g, spts, grp, spts[3 * (int64_t)t], spts[3 * (int64_t)t + 1], spts[3 * (int64_t)t + 2], eps, eps2, [&](int u) {
        if (u < t && core[u]) db_unite(parent, t, u);
        return true;
    });
}

// root of every core point and, per root, the smallest ORIGINAL index among its core points (= the point at which a sequential
// scan opens this cluster)
__global__ __launch_bounds__(256) void ibl_db_root_kernel(const unsigned char* __restrict__ core, int* __restrict__ parent, const int* __restrict__ order, int n,
                                                          int* __restrict__ root, unsigned* __restrict__ min_idx) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    if (!core[t]) { root[t] = -1; return; }
    const int r = db_find(parent, t);
    root[t] = r;
    atomicMin(&min_idx[r], (unsigned)order[t]);
}

__global__ __launch_bounds__(256) void ibl_db_is_root_kernel(const int* __restrict__ root, int n, int* __restrict__ flag) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < n) flag[t] = root[t] == t ? 1 : 0;
}

__global__ __launch_bounds__(256) void ibl_db_collect_roots_kernel(const int* __restrict__ flag, const int* __restrict__ pos, const unsigned* __restrict__ min_idx, int n,
                                                                   unsigned* __restrict__ keys, unsigned* __restrict__ vals) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < n && flag[t]) { keys[pos[t]] = min_idx[t]; vals[pos[t]] = (unsigned)t; }
}

// roots sorted by their first core point: cluster number within the group = rank - (roots of earlier groups)
__global__ __launch_bounds__(256) void ibl_db_number_kernel(const unsigned* __restrict__ keys_sorted, const unsigned* __restrict__ roots_sorted, int n_roots,
                                                            const int* __restrict__ grp_off, int n_grp, int* __restrict__ cid, int* __restrict__ grp_first /* [n_grp + 1] */) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k <= n_grp) {
        const unsigned target = (unsigned)grp_off[k];
        int lo = 0, hi = n_roots;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (keys_sorted[mid] < target) lo = mid + 1; else hi = mid;
        }
        grp_first[k] = lo;
    }
    if (k < n_roots) cid[roots_sorted[k]] = k;      // global rank; the label kernel subtracts the group's first rank
}

__global__ __launch_bounds__(256) void ibl_db_label_kernel(BatchGrid g, const double* __restrict__ spts, const int* __restrict__ grp_off, int n_grp, int n,
                                                           double eps, double eps2, const unsigned char* __restrict__ core, const int* __restrict__ root,
                                                           const int* __restrict__ cid, const int* __restrict__ grp_first, const int* __restrict__ order,
                                                           int* __restrict__ labels) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const int grp = seg_of(grp_off, n_grp, t);
    int best = INT_MAX;
    if (core[t]) {
        best = cid[root[t]];
    } else {
        db_scan(g, spts, grp, spts[3 * (int64_t)t], spts[3 * (int64_t)t + 1], spts[3 * (int64_t)t + 2], eps, eps2, [&](int u) {
            if (core[u]) {
                const int c = cid[root[u]];
                best = c < best ? c : best;
            }
            return true;
        });
    }
    labels[order[t]] = best == INT_MAX ? -1 : best - grp_first[grp];
}

}  // namespace

extern "C" int ibl_voxel_downsample_batch(ibl_reg_ctx* ctx, const double* points, const double* colors, const int32_t* seg_off_host, int32_t n_seg,
                                          double voxel_size, double* out_points, double* out_colors, int32_t* out_counts,
                                          int32_t* out_seg_off_host, void* stream) {
    if (!ctx || !seg_off_host || !out_seg_off_host || !(voxel_size > 0)) return ibl_set_error(IBL_ERR_ARG, "ibl_voxel_downsample_batch: bad argument");
    int st = check_offsets(seg_off_host, n_seg, "ibl_voxel_downsample_batch");
    if (st) return st;
    const int n = seg_off_host[n_seg];
    for (int s = 0; s <= n_seg; ++s) out_seg_off_host[s] = 0;
    if (n == 0) return IBL_OK;
    if (!points || !out_points || (colors && !out_colors)) return ibl_set_error(IBL_ERR_ARG, "ibl_voxel_downsample_batch: null buffer");
    if (n_seg > 65535) return ibl_set_error(IBL_ERR_ARG, "ibl_voxel_downsample_batch: at most 65535 objects per call (got %d)", n_seg);
    hipStream_t s = (hipStream_t)stream;
    ArenaMark mark(ctx);
    int* seg_off; long long *vidx, *rng; unsigned long long *keys, *keys_out; unsigned *vals, *idx_sorted; int *head, *vox_of, *start, *out_off;
    unsigned *first_idx, *first_sorted, *vox_id, *vox_by_rank;
    IBL_ARENA(seg_off, int, n_seg + 1);
    IBL_ARENA(vidx, long long, (int64_t)3 * n);
    IBL_ARENA(rng, long long, (int64_t)6 * n_seg);
    IBL_ARENA(keys, unsigned long long, n);
    IBL_ARENA(keys_out, unsigned long long, n);
    IBL_ARENA(vals, unsigned, n);
    IBL_ARENA(idx_sorted, unsigned, n);
    IBL_ARENA(head, int, n + 1);
    IBL_ARENA(vox_of, int, n + 1);
    IBL_ARENA(out_off, int, n_seg + 1);
    IBL_HIP_CHECK(hipMemcpyAsync(seg_off, seg_off_host, sizeof(int) * (size_t)(n_seg + 1), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(ibl_vox_index_kernel, dim3(n_seg), dim3(256), 0, s, points, seg_off, voxel_size, vidx, rng);
    IBL_LAUNCH_CHECK();
    std::vector<long long> h_rng((size_t)6 * n_seg);
    IBL_HIP_CHECK(hipMemcpyAsync(h_rng.data(), rng, sizeof(long long) * h_rng.size(), hipMemcpyDeviceToHost, s));
    IBL_HIP_CHECK(hipStreamSynchronize(s));
    for (int k = 0; k < n_seg; ++k)
        for (int a = 0; a < 3; ++a)
            if (h_rng[6 * k + 3 + a] - h_rng[6 * k + a] > 65535)
                return ibl_set_error(IBL_ERR_ARG, "ibl_voxel_downsample_batch: object %d spans more than 65536 voxels along axis %d", k, a);
    const int nb = (n + 255) / 256;
    hipLaunchKernelGGL(ibl_vox_key_kernel, dim3(nb), dim3(256), 0, s, vidx, rng, seg_off, n_seg, n, keys, vals);
    IBL_LAUNCH_CHECK();
    size_t tmp_a = 0, tmp_b = 0, tmp_c = 0;
    IBL_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_a, keys, keys_out, vals, idx_sorted, n, 0, 64, s));
    IBL_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_b, head, vox_of, n + 1, s));
    IBL_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_c, vals, vals, vals, vals, n, 0, 32, s));
    size_t tmp_bytes = tmp_a > tmp_b ? tmp_a : tmp_b;
    if (tmp_c > tmp_bytes) tmp_bytes = tmp_c;
    unsigned char* tmp;
    IBL_ARENA(tmp, unsigned char, (int64_t)tmp_bytes + 256);
    IBL_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_a, keys, keys_out, vals, idx_sorted, n, 0, 64, s));      // stable
    hipLaunchKernelGGL(ibl_vox_head_kernel, dim3(nb), dim3(256), 0, s, keys_out, n, head);
    IBL_LAUNCH_CHECK();
    IBL_HIP_CHECK(hipMemsetAsync(head + n, 0, sizeof(int), s));
    IBL_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_b, head, vox_of, n + 1, s));
    int n_vox = 0;
    IBL_HIP_CHECK(hipMemcpyAsync(&n_vox, vox_of + n, sizeof(int), hipMemcpyDeviceToHost, s));
    IBL_HIP_CHECK(hipStreamSynchronize(s));
    IBL_ARENA(start, int, n_vox + 1);
    IBL_ARENA(first_idx, unsigned, n_vox);
    IBL_ARENA(first_sorted, unsigned, n_vox);
    IBL_ARENA(vox_id, unsigned, n_vox);
    IBL_ARENA(vox_by_rank, unsigned, n_vox);
    hipLaunchKernelGGL(ibl_vox_start_kernel, dim3(nb), dim3(256), 0, s, head, vox_of, idx_sorted, n, n_vox, start, first_idx, vox_id);
    IBL_LAUNCH_CHECK();
    IBL_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_c, first_idx, first_sorted, vox_id, vox_by_rank, n_vox, 0, 32, s));
    hipLaunchKernelGGL(ibl_vox_mean_kernel, dim3((n_vox + 255) / 256), dim3(256), 0, s, points, colors, idx_sorted, start, vox_by_rank, n_vox, out_points,
                       out_colors, out_counts);
    IBL_LAUNCH_CHECK();
    hipLaunchKernelGGL(ibl_vox_offsets_kernel, dim3((n_seg + 256) / 256), dim3(256), 0, s, first_sorted, n_vox, seg_off, n_seg, out_off);
    IBL_LAUNCH_CHECK();
    IBL_HIP_CHECK(hipMemcpyAsync(out_seg_off_host, out_off, sizeof(int) * (size_t)(n_seg + 1), hipMemcpyDeviceToHost, s));
    IBL_HIP_CHECK(hipStreamSynchronize(s));
    return IBL_OK;
}

extern "C" int ibl_dbscan_batch(ibl_reg_ctx* ctx, const double* points, const int32_t* grp_off_host, int32_t n_grp, double eps, int32_t min_points,
                                int32_t* labels, int32_t* n_clusters_host, void* stream) {
    if (!ctx || !grp_off_host || !(eps > 0) || min_points < 1) return ibl_set_error(IBL_ERR_ARG, "ibl_dbscan_batch: bad argument");
    int st = check_offsets(grp_off_host, n_grp, "ibl_dbscan_batch");
    if (st) return st;
    const int n = grp_off_host[n_grp];
    if (n_clusters_host)
        for (int k = 0; k < n_grp; ++k) n_clusters_host[k] = 0;
    if (n == 0) return IBL_OK;
    if (!points || !labels) return ibl_set_error(IBL_ERR_ARG, "ibl_dbscan_batch: null buffer");
    hipStream_t s = (hipStream_t)stream;
    ArenaMark mark(ctx);
    int* grp_off; float4* pts4; double* spts; unsigned char* core; int *parent, *root, *flag, *pos, *cid, *grp_first; unsigned *min_idx, *rkeys, *rvals, *rkeys_s, *rvals_s;
    IBL_ARENA(grp_off, int, n_grp + 1);
    IBL_ARENA(pts4, float4, n);
    IBL_HIP_CHECK(hipMemcpyAsync(grp_off, grp_off_host, sizeof(int) * (size_t)(n_grp + 1), hipMemcpyHostToDevice, s));
    const int nb = (n + 255) / 256;
    hipLaunchKernelGGL(ibl_db_to_float_kernel, dim3(nb), dim3(256), 0, s, points, n, pts4);
    IBL_LAUNCH_CHECK();
    BatchGrid g;
    st = ibl_build_batch_grid(ctx, pts4, grp_off, grp_off_host, n_grp, (float)eps, (int64_t)64 << 20, &g, s);
    if (st) return st;
    IBL_ARENA(spts, double, (int64_t)3 * n);
    IBL_ARENA(core, unsigned char, n);
    IBL_ARENA(parent, int, n);
    IBL_ARENA(root, int, n);
    IBL_ARENA(flag, int, n + 1);
    IBL_ARENA(pos, int, n + 1);
    IBL_ARENA(cid, int, n);
    IBL_ARENA(min_idx, unsigned, n);
    IBL_ARENA(grp_first, int, n_grp + 1);
    const double eps2 = eps * eps;
    hipLaunchKernelGGL(ibl_db_gather_kernel, dim3(nb), dim3(256), 0, s, points, g.order, n, spts);
    IBL_LAUNCH_CHECK();
    hipLaunchKernelGGL(ibl_db_core_kernel, dim3(nb), dim3(256), 0, s, g, spts, grp_off, n_grp, n, eps, eps2, min_points, core, parent);
    IBL_LAUNCH_CHECK();
    hipLaunchKernelGGL(ibl_db_union_kernel, dim3(nb), dim3(256), 0, s, g, spts, grp_off, n_grp, n, eps, eps2, core, parent);
    IBL_LAUNCH_CHECK();
    IBL_HIP_CHECK(hipMemsetAsync(min_idx, 0xff, sizeof(unsigned) * (size_t)n, s));
    hipLaunchKernelGGL(ibl_db_root_kernel, dim3(nb), dim3(256), 0, s, core, parent, g.order, n, root, min_idx);
    IBL_LAUNCH_CHECK();
    hipLaunchKernelGGL(ibl_db_is_root_kernel, dim3(nb), dim3(256), 0, s, root, n, flag);
    IBL_LAUNCH_CHECK();
    IBL_HIP_CHECK(hipMemsetAsync(flag + n, 0, sizeof(int), s));
    size_t tmp_a = 0, tmp_b = 0;
    IBL_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_a, flag, pos, n + 1, s));
    IBL_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_b, min_idx, min_idx, min_idx, min_idx, n, 0, 32, s));
    unsigned char* tmp;
    IBL_ARENA(tmp, unsigned char, (int64_t)(tmp_a > tmp_b ? tmp_a : tmp_b) + 256);
    IBL_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_a, flag, pos, n + 1, s));
    int n_roots = 0;
    IBL_HIP_CHECK(hipMemcpyAsync(&n_roots, pos + n, sizeof(int), hipMemcpyDeviceToHost, s));
    IBL_HIP_CHECK(hipStreamSynchronize(s));
    IBL_ARENA(rkeys, unsigned, n_roots + 1);
    IBL_ARENA(rvals, unsigned, n_roots + 1);
    IBL_ARENA(rkeys_s, unsigned, n_roots + 1);
    IBL_ARENA(rvals_s, unsigned, n_roots + 1);
    if (n_roots > 0) {
        hipLaunchKernelGGL(ibl_db_collect_roots_kernel, dim3(nb), dim3(256), 0, s, flag, pos, min_idx, n, rkeys, rvals);
        IBL_LAUNCH_CHECK();
        IBL_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_b, rkeys, rkeys_s, rvals, rvals_s, n_roots, 0, 32, s));
    }
    const int nk = (n_roots > n_grp + 1 ? n_roots : n_grp + 1);
    hipLaunchKernelGGL(ibl_db_number_kernel, dim3((nk + 255) / 256), dim3(256), 0, s, rkeys_s, rvals_s, n_roots, grp_off, n_grp, cid, grp_first);
    IBL_LAUNCH_CHECK();
    hipLaunchKernelGGL(ibl_db_label_kernel, dim3(nb), dim3(256), 0, s, g, spts, grp_off, n_grp, n, eps, eps2, core, root, cid, grp_first, g.order, labels);
    IBL_LAUNCH_CHECK();
    if (n_clusters_host) {
        std::vector<int> h_first((size_t)n_grp + 1);
        IBL_HIP_CHECK(hipMemcpyAsync(h_first.data(), grp_first, sizeof(int) * h_first.size(), hipMemcpyDeviceToHost, s));
        IBL_HIP_CHECK(hipStreamSynchronize(s));
        for (int k = 0; k < n_grp; ++k) n_clusters_host[k] = h_first[k + 1] - h_first[k];
    }
    return IBL_OK;
}
