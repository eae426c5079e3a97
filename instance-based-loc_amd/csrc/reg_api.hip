// reg_api.hip -- C-ABI entry points of the registration path: context (device arena), batched
// radius-outlier removal and normals + FPFH.  The fused register / evaluate drivers live in
// reg_register.hip.
#pragma clang fp contract(off)
#include <hip/hip_runtime.h>

#include <hipcub/hipcub.hpp>

#include <vector>

#include "ibloc.h"
#include "reg_common.h"

// Cells per search radius of the hybrid k-NN grids: the max_nn nearest neighbours of a dense cloud sit in a small fraction of the
// radius, and hybrid_select (reg_knn.hip) only walks the cube of cells that provably holds them.  (Per-segment cell sizes never drop
// below extent / 128, reg_grid.hip.)
#ifndef KNN_CELLS_NORMAL
#define KNN_CELLS_NORMAL 3.0        // r = 2 voxel, 30 neighbours
#endif
#ifndef KNN_CELLS_FEATURE
#define KNN_CELLS_FEATURE 5.0       // r = 5 voxel, 100 neighbours
#endif

#ifndef KNN_TILE_NORMAL
#define KNN_TILE_NORMAL 4           // tile edge in cells of the 30-neighbour searches (normals, colour gradients)
#endif
#ifndef KNN_TILE_FEATURE
#define KNN_TILE_FEATURE 2          // ... of the 100-neighbour SPFH search
#endif

int ibl_launch_normals(ibl_reg_ctx* ctx, const BatchGrid& g, const float4* pts, const int* seg_off, int n, double radius, int max_nn, float4* normals,
                       int* status, hipStream_t s);
int ibl_launch_fpfh(ibl_reg_ctx* ctx, const BatchGrid& g, const float4* pts, const float4* normals, const int* seg_off, int n, double radius, int max_nn,
                    unsigned char* spfh, int* nbr_idx, float* nbr_d2, int* nbr_cnt, float* fpfh, int matching_order, int* status,
                    hipStream_t s);
int ibl_launch_radius_count(const BatchGrid& g, const float4* pts, const int* seg_off, int n, double radius, int nb_points,
                            unsigned char* keep, hipStream_t s);

bool ibl_normals_fpfh_fusable(double radius_normal, int max_nn_normal, double radius_feature, int max_nn_feature);
int ibl_launch_normals_fpfh(ibl_reg_ctx* ctx, const BatchGrid& g, const float4* pts, const int* seg_off, int n, double radius_normal,
                            int max_nn_normal, double radius_feature, int max_nn_feature, float4* normals, unsigned char* spfh, int* nbr_idx,
                            float* nbr_d2, int* nbr_cnt, float* fpfh, int matching_order, int* status, hipStream_t s);
int ibl_launch_color_grad(ibl_reg_ctx* ctx, const BatchGrid& g, const float4* pts, const float4* normals, const int* seg_off, int q0, int q1, double radius,
                          int max_nn, float4* grad, int* status, hipStream_t s);

// bounding boxes of the segments on the host (synchronises the stream)
static int ibl_bbox_to_host(ibl_reg_ctx* ctx, const float4* P, const int* seg_off_dev, int n_seg, float* bbox_host, hipStream_t s) {
    if (n_seg <= 0) return IBL_OK;
    ArenaMark m(ctx);
    float* bbox;
    IBL_ARENA(bbox, float, (int64_t)n_seg * 6 + 6);
    int st = ibl_launch_bbox(P, seg_off_dev, n_seg, bbox, s);
    if (st) return st;
    IBL_HIP_CHECK(hipMemcpyAsync(bbox_host, bbox, sizeof(float) * 6 * (size_t)n_seg, hipMemcpyDeviceToHost, s));
    IBL_HIP_CHECK(hipStreamSynchronize(s));
    return IBL_OK;
}

extern "C" int ibl_reg_ctx_create(ibl_reg_ctx** out, int64_t arena_bytes) {
    if (!out || arena_bytes < (1 << 20)) return ibl_set_error(IBL_ERR_ARG, "ibl_reg_ctx_create: bad argument");
    ibl_reg_ctx* c = new ibl_reg_ctx();
    IBL_HIP_CHECK(hipGetDevice(&c->device));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->base), (size_t)arena_bytes);
    if (e != hipSuccess) {
        delete c;
        return ibl_set_error(IBL_ERR_HIP, "ibl_reg_ctx_create: hipMalloc(%lld) failed: %s", (long long)arena_bytes, hipGetErrorString(e));
    }
    c->size = arena_bytes;
    // the first 256 bytes hold the device status words
    c->d_status = reinterpret_cast<int*>(c->base);
    c->used = 256;
    e = hipMemset(c->base, 0, 256);
    if (e != hipSuccess) { (void)hipFree(c->base); delete c; return ibl_set_error(IBL_ERR_HIP, "ibl_reg_ctx_create: memset failed"); }
    c->pin_size = 32 << 20;            // pinned staging of the plan tables (job / pair / grid descriptors of one call)
    if (hipHostMalloc(reinterpret_cast<void**>(&c->pin), (size_t)c->pin_size, hipHostMallocDefault) != hipSuccess) { c->pin = nullptr; c->pin_size = 0; }
    *out = c;
    return IBL_OK;
}

extern "C" int ibl_reg_ctx_destroy(ibl_reg_ctx* ctx) {
    if (!ctx) return IBL_OK;
    (void)hipFree(ctx->base);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    delete ctx;
    return IBL_OK;
}

extern "C" int ibl_reg_ctx_reset(ibl_reg_ctx* ctx) {
    if (!ctx) return ibl_set_error(IBL_ERR_ARG, "ibl_reg_ctx_reset: null context");
    ctx->used = 256;           // drops every persistent allocation (memory grids built from this arena become invalid)
    return IBL_OK;
}

extern "C" int64_t ibl_reg_ctx_high_water(const ibl_reg_ctx* ctx) { return ctx ? ctx->high_water : -1; }

extern "C" int ibl_reg_ctx_status(ibl_reg_ctx* ctx, int clear) {
    if (!ctx) return -1;
    int h = 0;
    if (hipMemcpy(&h, ctx->d_status, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (clear) (void)hipMemset(ctx->d_status, 0, sizeof(int));
    return h;
}

static int check_seg(const int32_t* seg_off_host, int n_seg, const char* who) {
    if (!seg_off_host || n_seg < 0) return ibl_set_error(IBL_ERR_ARG, "%s: bad segment table", who);
    if (seg_off_host[0] != 0) return ibl_set_error(IBL_ERR_ARG, "%s: seg_off[0] must be 0", who);
    for (int s = 0; s < n_seg; ++s)
        if (seg_off_host[s + 1] < seg_off_host[s]) return ibl_set_error(IBL_ERR_ARG, "%s: seg_off must be non-decreasing", who);
    return IBL_OK;
}

extern "C" int ibl_radius_outlier_batch(ibl_reg_ctx* ctx, const float* pts4, const int32_t* seg_off_dev, const int32_t* seg_off_host,
                                        int n_seg, double radius, int nb_points, uint8_t* keep, void* stream) {
    if (!ctx || !pts4 || !seg_off_dev || !keep || radius <= 0) return ibl_set_error(IBL_ERR_ARG, "ibl_radius_outlier_batch: bad argument");
    int st = check_seg(seg_off_host, n_seg, "ibl_radius_outlier_batch");
    if (st) return st;
    ArenaMark mark(ctx);
    hipStream_t s = (hipStream_t)stream;
    BatchGrid g;
    void* tok;
    ibl_prof_begin(IBL_PROF_ST_OUTLIER, 13.0 * (double)seg_off_host[n_seg], s, &tok);
    st = ibl_build_batch_grid(ctx, reinterpret_cast<const float4*>(pts4), seg_off_dev, seg_off_host, n_seg, (float)radius,
                              (int64_t)64 << 20, &g, s);
    if (st) return st;
    st = ibl_launch_radius_count(g, reinterpret_cast<const float4*>(pts4), seg_off_dev, seg_off_host[n_seg], radius, nb_points, keep, s);
    ibl_prof_end(tok, s);
    return st;
}

extern "C" int ibl_normals_fpfh_batch(ibl_reg_ctx* ctx, const float* pts4, const int32_t* seg_off_dev, const int32_t* seg_off_host,
                                      int n_seg, double radius_normal, int max_nn_normal, double radius_feature, int max_nn_feature,
                                      float* normals4, float* fpfh, void* stream) {
    if (!ctx || !pts4 || !seg_off_dev || !normals4 || radius_normal <= 0 || max_nn_normal <= 0)
        return ibl_set_error(IBL_ERR_ARG, "ibl_normals_fpfh_batch: bad argument");
    int st = check_seg(seg_off_host, n_seg, "ibl_normals_fpfh_batch");
    if (st) return st;
    const int n = seg_off_host[n_seg];
    ArenaMark mark(ctx);
    hipStream_t s = (hipStream_t)stream;
    const float4* P = reinterpret_cast<const float4*>(pts4);
    std::vector<float> bbox_host((size_t)n_seg * 6 + 6);
    st = ibl_bbox_to_host(ctx, P, seg_off_dev, n_seg, bbox_host.data(), s);
    if (st) return st;
    const bool fused = fpfh && radius_feature > 0 && max_nn_feature > 0 && ibl_normals_fpfh_fusable(radius_normal, max_nn_normal, radius_feature, max_nn_feature);
    if (!fused) {
        ArenaMark m2(ctx);
        BatchGrid g;
        st = ibl_build_tile_grid(ctx, P, seg_off_dev, seg_off_host, n_seg, bbox_host.data(), radius_normal, max_nn_normal, KNN_TILE_NORMAL,
                                 (int64_t)64 << 20, &g, s);
        if (st) return st;
        st = ibl_launch_normals(ctx, g, P, seg_off_dev, n, radius_normal, max_nn_normal, reinterpret_cast<float4*>(normals4), ctx->d_status, s);
        if (st) return st;
    }
    if (fpfh) {
        if (radius_feature <= 0 || max_nn_feature <= 0) return ibl_set_error(IBL_ERR_ARG, "ibl_normals_fpfh_batch: bad feature parameters");
        BatchGrid g;
        st = ibl_build_tile_grid(ctx, P, seg_off_dev, seg_off_host, n_seg, bbox_host.data(), radius_feature, max_nn_feature, KNN_TILE_FEATURE,
                                 (int64_t)64 << 20, &g, s);
        if (st) return st;
        unsigned char* spfh; int* nbr_idx; float* nbr_d2; int* nbr_cnt;
        IBL_ARENA(spfh, unsigned char, (int64_t)n * 36 + 64);
        IBL_ARENA(nbr_idx, int, (int64_t)n * max_nn_feature + 64);
        IBL_ARENA(nbr_d2, float, (int64_t)n * max_nn_feature + 64);
        IBL_ARENA(nbr_cnt, int, n + 64);
        if (fused)           // one search for both (the 30 nearest within the normal radius are among the 100 nearest within the feature radius)
            st = ibl_launch_normals_fpfh(ctx, g, P, seg_off_dev, n, radius_normal, max_nn_normal, radius_feature, max_nn_feature,
                                         reinterpret_cast<float4*>(normals4), spfh, nbr_idx, nbr_d2, nbr_cnt, fpfh, 0, ctx->d_status, s);
        else
            st = ibl_launch_fpfh(ctx, g, P, reinterpret_cast<const float4*>(normals4), seg_off_dev, n, radius_feature, max_nn_feature, spfh,
                                 nbr_idx, nbr_d2, nbr_cnt, fpfh, 0, ctx->d_status, s);
        if (st) return st;
    }
    IBL_HIP_CHECK(hipStreamSynchronize(s));        // the pinned staging of the grid tables is recycled by the next call
    ibl_stage_reset(ctx);
    return IBL_OK;
}

// ------------------------------------------------------------------------------------------------
// registration features of a batch of clouds (shared by the instance cache and by ibl_register_batch_cached)
// ------------------------------------------------------------------------------------------------
// every FPFH row once more as the 48 fp16 search operands of reg_featnn.hip (layout, centring and error budget in its header and at
// fm_operand_piece, reg_common.h), and the squared norm of the CENTRED row.  split may be null (instance features kept without their
// operand rows: 168 instead of 264 resident bytes per point; the search then builds the pieces from the fp32 rows as it stages them).
__global__ __launch_bounds__(256) void ibl_fpfh_half_kernel(const float* __restrict__ fpfh, int n, unsigned short* __restrict__ split,
                                                            float* __restrict__ norm) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* __restrict__ x = fpfh + (int64_t)i * 33;
    const float a = fm_centred_norm(x);
    norm[i] = a;
    if (!split) return;
    uint4* dst = reinterpret_cast<uint4*>(split + (int64_t)i * 48);
#pragma unroll
    for (int pc = 0; pc < 6; ++pc) {
        float x8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x8[e] = 8 * pc + e < 33 ? x[8 * pc + e] : 0.0f;
        const fm_piece_t r = fm_operand_piece(x8, a, pc);
        uint4 v;
        __builtin_memcpy(&v, &r, 16);
        dst[pc] = v;
    }
}

// bbox_host: [n_seg][6] bounding boxes of the segments as ibl_launch_bbox computes them (the callers hold them: instance features
// return them, a recomputed group's box is the union of its instances' boxes), or null: computed and read back here (one
// synchronisation).  With the boxes on the host the three grids are dimensioned there: no further read-back.
int ibl_features_on_batch(ibl_reg_ctx* ctx, const float4* P, const int* seg_off_dev, const int* seg_off_host, int n_seg, const float* bbox_host,
                          double voxel_size, double grad_radius, int gq0, int gq1, float4* normals, float* fpfh, unsigned short* fpfh_split,
                          float* fpfh_norm, float4* grad, hipStream_t s) {
    const int n = seg_off_host[n_seg];
    if (n <= 0) return IBL_OK;
    int st;
    std::vector<float> own_bbox;
    if (!bbox_host) {
        own_bbox.resize((size_t)n_seg * 6 + 6);
        st = ibl_bbox_to_host(ctx, P, seg_off_dev, n_seg, own_bbox.data(), s);
        if (st) return st;
        bbox_host = own_bbox.data();
    }
    const bool fused = fpfh && ibl_normals_fpfh_fusable(voxel_size * 2, 30, voxel_size * 5, 100);
    if (!fused) {
        ArenaMark mA(ctx);
        BatchGrid gA;
        st = ibl_build_tile_grid(ctx, P, seg_off_dev, seg_off_host, n_seg, bbox_host, voxel_size * 2, 30, KNN_TILE_NORMAL, (int64_t)128 << 20, &gA, s);
        if (st) return st;
        st = ibl_launch_normals(ctx, gA, P, seg_off_dev, n, voxel_size * 2, 30, normals, ctx->d_status, s);
        if (st) return st;
    }
    if (fpfh) {
        ArenaMark mB(ctx);
        BatchGrid gB;
        st = ibl_build_tile_grid(ctx, P, seg_off_dev, seg_off_host, n_seg, bbox_host, voxel_size * 5, 100, KNN_TILE_FEATURE, (int64_t)128 << 20, &gB, s);
        if (st) return st;
        unsigned char* spfh; float* nbr_d2; int *nbr_idx, *nbr_cnt;
        IBL_ARENA(spfh, unsigned char, (int64_t)n * 36 + 64);
        IBL_ARENA(nbr_idx, int, (int64_t)n * 100 + 64);
        IBL_ARENA(nbr_d2, float, (int64_t)n * 100 + 64);
        IBL_ARENA(nbr_cnt, int, n + 64);
        if (fused) st = ibl_launch_normals_fpfh(ctx, gB, P, seg_off_dev, n, voxel_size * 2, 30, voxel_size * 5, 100, normals, spfh, nbr_idx, nbr_d2, nbr_cnt,
                                                fpfh, 1, ctx->d_status, s);
        else st = ibl_launch_fpfh(ctx, gB, P, normals, seg_off_dev, n, voxel_size * 5, 100, spfh, nbr_idx, nbr_d2, nbr_cnt, fpfh, 1, ctx->d_status, s);
        if (st) return st;
        if (fpfh_norm) {
            hipLaunchKernelGGL(ibl_fpfh_half_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, fpfh, n, fpfh_split, fpfh_norm);
            IBL_LAUNCH_CHECK();
        }
    }
    if (grad && gq1 > gq0) {
        if (grad_radius <= 0) return ibl_set_error(IBL_ERR_ARG, "colour gradients need a positive radius");
        ArenaMark mG(ctx);
        BatchGrid gG;
        st = ibl_build_tile_grid(ctx, P, seg_off_dev, seg_off_host, n_seg, bbox_host, grad_radius, 30, KNN_TILE_NORMAL, (int64_t)128 << 20, &gG, s);
        if (st) return st;
        st = ibl_launch_color_grad(ctx, gG, P, normals, seg_off_dev, gq0, gq1, grad_radius, 30, grad, ctx->d_status, s);
        if (st) return st;
    }
    return IBL_OK;
}

extern "C" int ibl_instance_features_batch(ibl_reg_ctx* ctx, const float* pts4, const int32_t* seg_off_dev, const int32_t* seg_off_host,
                                           int n_seg, double voxel_size, double grad_radius, float* normals4, float* fpfh,
                                           uint16_t* fpfh_split, float* fpfh_norm, float* grad4, float* bbox_host, void* stream) {
    if (!ctx || !pts4 || !seg_off_dev || !normals4 || !fpfh || !fpfh_norm || voxel_size <= 0)
        return ibl_set_error(IBL_ERR_ARG, "ibl_instance_features_batch: bad argument");
    if (grad4 && grad_radius <= 0) return ibl_set_error(IBL_ERR_ARG, "ibl_instance_features_batch: colour gradients need grad_radius > 0");
    int st = check_seg(seg_off_host, n_seg, "ibl_instance_features_batch");
    if (st) return st;
    ArenaMark mark(ctx);
    hipStream_t s = (hipStream_t)stream;
    const float4* P = reinterpret_cast<const float4*>(pts4);
    std::vector<float> own_bbox;
    if (!bbox_host) { own_bbox.resize((size_t)n_seg * 6 + 6); bbox_host = own_bbox.data(); }
    if (n_seg > 0) {
        st = ibl_bbox_to_host(ctx, P, seg_off_dev, n_seg, bbox_host, s);        // the one synchronisation of this call before its end
        if (st) return st;
    }
    void* tok;
    ibl_prof_begin(IBL_PROF_ST_FEATURES, 444.0 * (double)(n_seg > 0 ? seg_off_host[n_seg] : 0), s, &tok);
    // chunks of whole clouds bound the scratch (800 B / point of neighbour lists): a 10k-instance memory is 50M points.  Round 4: at most
    // 1.5 M points per chunk, and chunks of EQUAL size -- a bench step's 224 detections are 1.105 M points, which the former fixed 2^20-point
    // chunks split into 1.046 M + 0.059 M: a second round of grid builds, sorts and an almost empty tile search (features are per cloud: the
    // chunking does not touch the results)
    const int64_t total_pts = n_seg > 0 ? (int64_t)seg_off_host[n_seg] : 0;
    const int64_t cap_pts = (int64_t)3 << 19;
    const int64_t n_chunks = std::max<int64_t>(1, (total_pts + cap_pts - 1) / cap_pts);
    const int64_t chunk_pts = std::min<int64_t>(cap_pts, (total_pts + n_chunks - 1) / n_chunks + 65536);
    std::vector<int> rebased;
    for (int s0 = 0; s0 < n_seg;) {
        int s1 = s0 + 1;
        while (s1 < n_seg && (int64_t)seg_off_host[s1 + 1] - seg_off_host[s0] <= chunk_pts) ++s1;
        const int o0 = seg_off_host[s0], ns = s1 - s0;
        ArenaMark mc(ctx);
        const int* off_dev = seg_off_dev;
        const int* off_host = seg_off_host;
        if (!(s0 == 0 && s1 == n_seg)) {
            rebased.resize(ns + 1);
            for (int i = 0; i <= ns; ++i) rebased[i] = seg_off_host[s0 + i] - o0;
            int* d;
            IBL_ARENA(d, int, ns + 1);
            st = ibl_stage_upload(ctx, d, rebased.data(), sizeof(int) * (int64_t)(ns + 1), s);     // staged: `rebased` is rewritten by the next chunk
            if (st) return st;
            off_dev = d;
            off_host = rebased.data();
        }
        const int cnt = off_host[ns];
        st = ibl_features_on_batch(ctx, P + o0, off_dev, off_host, ns, bbox_host + 6 * (size_t)s0, voxel_size, grad_radius, 0, grad4 ? cnt : 0,
                                   reinterpret_cast<float4*>(normals4) + o0, fpfh + (int64_t)o0 * 33, fpfh_split ? fpfh_split + (int64_t)o0 * 48 : nullptr,
                                   fpfh_norm + o0, grad4 ? reinterpret_cast<float4*>(grad4) + o0 : nullptr, s);
        if (st) return st;
        s0 = s1;
        if (ctx->pin_used > ctx->pin_size / 2) {           // many chunks (a whole memory): recycle the staging buffer
            IBL_HIP_CHECK(hipStreamSynchronize(s));
            ibl_stage_reset(ctx);
        }
    }
    ibl_prof_end(tok, s);
    IBL_HIP_CHECK(hipStreamSynchronize(s));
    ibl_stage_reset(ctx);
    return IBL_OK;
}

// ------------------------------------------------------------------------------------------------
// depth + masks -> coloured clouds (utils/depth_utils.py:46-90,176-206 before the outlier step)
// ------------------------------------------------------------------------------------------------
// depth_type: 0 float32 (numpy keeps float32 arithmetic: float32 array op python float), 1 uint16, 2 float64 (both float64)
__device__ __forceinline__ double depth_at(const void* depth, int type, int64_t p) {
    return type == 1 ? (double)reinterpret_cast<const unsigned short*>(depth)[p]
                     : (type == 2 ? reinterpret_cast<const double*>(depth)[p] : (double)reinterpret_cast<const float*>(depth)[p]);
}
__device__ __forceinline__ double depth_z(const void* depth, int type, int64_t p, double factor) {
    if (type == 0) return (double)(reinterpret_cast<const float*>(depth)[p] / (float)factor);
    return depth_at(depth, type, p) / factor;
}

// flags[m * HW + p] = pixel p belongs to cloud m  (mask set and z != 0; z = depth / factor is non-zero iff it is after rounding)
__global__ __launch_bounds__(256) void ibl_unproject_flag_kernel(const void* __restrict__ depth, int is_u16, const unsigned char* __restrict__ masks,
                                                                 int64_t HW, int64_t total, double depth_factor, int* __restrict__ flags) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int64_t p = t % HW;
    const double z = depth_z(depth, is_u16, p, depth_factor);
    flags[t] = (masks[t] != 0 && z != 0.0) ? 1 : 0;
}

__global__ __launch_bounds__(256) void ibl_unproject_write_kernel(const void* __restrict__ depth, int is_u16, const unsigned char* __restrict__ rgb,
                                                                  const int* __restrict__ flags, const int* __restrict__ pos, int64_t HW, int W,
                                                                  int64_t total, const float* __restrict__ hx, const float* __restrict__ vy,
                                                                  double fx, double fy, double depth_factor, int64_t capacity,
                                                                  float4* __restrict__ out, double* __restrict__ out_pts64, double* __restrict__ out_cols64) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total || !flags[t]) return;
    const int o = pos[t];
    if (o >= capacity) return;
    const int64_t p = t % HW;
    const int row = (int)(p / W), col = (int)(p - (int64_t)row * W);
    const double z = depth_z(depth, is_u16, p, depth_factor);                     // depth_image / depth_factor
    double x, y;
    if (is_u16 == 0) {                                                            // float32 depth: numpy stays in float32
        const float zf = (float)z;
        x = (double)((hx[col] * zf) / (float)fx);
        y = (double)((vy[row] * zf) / (float)fy);
    } else {                                                                      // float32 grid * float64 depth / focal length
        x = (double)hx[col] * z / fx;
        y = (double)vy[row] * z / fy;
    }
    const float r = (float)rgb[3 * p] / 255.0f, g = (float)rgb[3 * p + 1] / 255.0f, b = (float)rgb[3 * p + 2] / 255.0f;
    const double inten = ((double)r + (double)g + (double)b) / 3.0;
    out[o] = make_float4((float)x, (float)y, (float)z, (float)inten);
    if (out_pts64) {            // what Open3D's Vector3dVector holds: the numpy values (float32 or float64) widened to double
        out_pts64[3 * (int64_t)o] = x; out_pts64[3 * (int64_t)o + 1] = y; out_pts64[3 * (int64_t)o + 2] = is_u16 == 0 ? (double)(float)z : z;
    }
    if (out_cols64) { out_cols64[3 * (int64_t)o] = (double)r; out_cols64[3 * (int64_t)o + 1] = (double)g; out_cols64[3 * (int64_t)o + 2] = (double)b; }
}

// np.linspace(start, stop, num, dtype=float32): float64 `i * step + start`, last element = stop, rounded to float32
static void linspace_f32(double start, double stop, int num, std::vector<float>& out) {
    out.resize(num);
    if (num == 1) { out[0] = (float)start; return; }
    const double step = (stop - start) / (double)(num - 1);
    for (int i = 0; i < num; ++i) {
        volatile double prod = (double)i * step;       // keep the product and the sum separate roundings, as numpy does
        out[i] = (float)(prod + start);
    }
    out[num - 1] = (float)stop;
}

extern "C" int ibl_unproject_masks_f64(ibl_reg_ctx* ctx, const void* depth, int depth_type, const uint8_t* rgb, const uint8_t* masks,
                                       int n_masks, int H, int W, double fx, double fy, double depth_factor, float* pts4, double* pts3_f64,
                                       double* colors3_f64, int64_t capacity, int32_t* seg_off_dev, int32_t* seg_off_host, void* stream);

extern "C" int ibl_unproject_masks(ibl_reg_ctx* ctx, const void* depth, int depth_type, const uint8_t* rgb, const uint8_t* masks,
                                   int n_masks, int H, int W, double fx, double fy, double depth_factor, float* pts4, int64_t capacity,
                                   int32_t* seg_off_dev, int32_t* seg_off_host, void* stream) {
    return ibl_unproject_masks_f64(ctx, depth, depth_type, rgb, masks, n_masks, H, W, fx, fy, depth_factor, pts4, nullptr, nullptr, capacity,
                                   seg_off_dev, seg_off_host, stream);
}

extern "C" int ibl_unproject_masks_f64(ibl_reg_ctx* ctx, const void* depth, int depth_type, const uint8_t* rgb, const uint8_t* masks,
                                       int n_masks, int H, int W, double fx, double fy, double depth_factor, float* pts4, double* pts3_f64,
                                       double* colors3_f64, int64_t capacity, int32_t* seg_off_dev, int32_t* seg_off_host, void* stream) {
    if (!ctx || !seg_off_dev || !seg_off_host || n_masks < 0 || (n_masks > 0 && (!depth || !rgb || !masks || !pts4)) || H <= 0 || W <= 0 || fx == 0 || fy == 0 ||
        depth_factor == 0 || capacity < 0 || depth_type < 0 || depth_type > 2)
        return ibl_set_error(IBL_ERR_ARG, "ibl_unproject_masks: bad argument");
    const int64_t HW = (int64_t)H * W, total = HW * n_masks;
    if (total > 0x7fffffff) return ibl_set_error(IBL_ERR_ARG, "ibl_unproject_masks: n_masks * H * W exceeds 2^31");
    hipStream_t s = (hipStream_t)stream;
    seg_off_host[0] = 0;
    if (n_masks == 0) { IBL_HIP_CHECK(hipMemsetAsync(seg_off_dev, 0, sizeof(int), s)); IBL_HIP_CHECK(hipStreamSynchronize(s)); return IBL_OK; }
    ArenaMark mark(ctx);
    std::vector<float> hx, vy;
    linspace_f32(-(double)W / 2.0, (double)W / 2.0, W, hx);       // the reference's `horizontal_distance` (its h is the column count)
    linspace_f32((double)H / 2.0, -(double)H / 2.0, H, vy);       // `vertical_distance`
    float *d_hx, *d_vy; int *flags, *pos;
    IBL_ARENA(d_hx, float, W);
    IBL_ARENA(d_vy, float, H);
    IBL_ARENA(flags, int, total + 1);
    IBL_ARENA(pos, int, total + 1);
    IBL_HIP_CHECK(hipMemcpyAsync(d_hx, hx.data(), sizeof(float) * W, hipMemcpyHostToDevice, s));
    IBL_HIP_CHECK(hipMemcpyAsync(d_vy, vy.data(), sizeof(float) * H, hipMemcpyHostToDevice, s));
    const unsigned nb = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(ibl_unproject_flag_kernel, dim3(nb), dim3(256), 0, s, depth, depth_type, masks, HW, total, depth_factor, flags);
    IBL_LAUNCH_CHECK();
    IBL_HIP_CHECK(hipMemsetAsync(flags + total, 0, sizeof(int), s));
    size_t tmp_bytes = 0;
    IBL_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, flags, pos, (int)(total + 1), s));
    unsigned char* tmp;
    IBL_ARENA(tmp, unsigned char, (int64_t)tmp_bytes + 256);
    IBL_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, flags, pos, (int)(total + 1), s));
    std::vector<int> off(n_masks + 1);
    for (int m = 0; m <= n_masks; ++m)
        IBL_HIP_CHECK(hipMemcpyAsync(&off[m], pos + (int64_t)m * HW, sizeof(int), hipMemcpyDeviceToHost, s));
    IBL_HIP_CHECK(hipStreamSynchronize(s));
    if (off[n_masks] > capacity)
        return ibl_set_error(IBL_ERR_ARG, "ibl_unproject_masks: %d points do not fit the output capacity %lld", off[n_masks], (long long)capacity);
    for (int m = 0; m <= n_masks; ++m) seg_off_host[m] = off[m];
    IBL_HIP_CHECK(hipMemcpyAsync(seg_off_dev, seg_off_host, sizeof(int) * (n_masks + 1), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(ibl_unproject_write_kernel, dim3(nb), dim3(256), 0, s, depth, depth_type, rgb, flags, pos, HW, W, total, d_hx, d_vy, fx, fy,
                       depth_factor, capacity, reinterpret_cast<float4*>(pts4), pts3_f64, colors3_f64);
    IBL_LAUNCH_CHECK();
    IBL_HIP_CHECK(hipStreamSynchronize(s));
    return IBL_OK;
}
