// localise.hip -- stage B of localise() as ONE C-ABI call: clean the detected clouds, compute their registration features, register
// every candidate assignment, score it against the whole memory and pick the winner per frame.
//
// Replaces the loop body of ObjectMemory.localise() after the assignment search
//   object_memory/object_memory.py:992-998   (radius-outlier removal of every detected cloud)
//   object_memory/object_memory.py:1020-1106 (per assignment: concatenate, register_point_clouds, evaluate_transform on the whole memory)
//   object_memory/object_memory.py:1111-1114 (best assignment = highest whole-memory fitness, first on ties)
// for a whole batch of frames.  SURVEY 8b names this fused driver; until round 3 the same sequence was issued by Python
// (ibloc_amd/engine.py) with torch ops for the compaction of the cleaned points between the calls.  Nothing numerical lives here: the
// stages are the library's own entry points, composed on the host, with the job tables built from the assignment lists in C++.
#pragma clang fp contract(off)
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <vector>

#include "ibloc.h"
#include "reg_common.h"

namespace {

struct KeepToInt {          // the scan must accumulate in int (hipcub's accumulator follows the input value type)
    __host__ __device__ __forceinline__ int operator()(unsigned char v) const { return v ? 1 : 0; }
};

// clean[pos[i]] = pts[i] for the kept points (order preserved); new_off[s] = pos[seg_off[s]]
__global__ __launch_bounds__(256) void ibl_compact_kept_kernel(const float4* __restrict__ pts, const unsigned char* __restrict__ keep,
                                                               const int* __restrict__ pos, int n, float4* __restrict__ clean) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && keep[i]) clean[pos[i]] = pts[i];
}

__global__ __launch_bounds__(256) void ibl_compact_offsets_kernel(const int* __restrict__ seg_off, const int* __restrict__ pos, int n_seg,
                                                                  int* __restrict__ new_off) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s <= n_seg) new_off[s] = pos[seg_off[s]];
}

}  // namespace

extern "C" int ibl_register_evaluate_batch(ibl_reg_ctx* ctx, const float* det_pts4, const int32_t* det_off_dev, const int32_t* det_off_host,
                                           int n_det_seg, const int32_t* q_per_frame, int n_frames, const int32_t* assn,
                                           const int32_t* assn_len, const int32_t* assn_count, int max_assn, const float* mem_pts4,
                                           const int32_t* mem_off_dev, const int32_t* mem_off_host, int n_mem_seg,
                                           const ibl_instance_features* mem_features, const ibl_memgrid* grid, double voxel_size,
                                           double global_dist_factor, double local_dist_factor, double outlier_radius, int outlier_nb_points,
                                           double eval_threshold, uint64_t seed, uint32_t job_id_base, int64_t ransac_max_iter, int flags,
                                           int max_jobs, int32_t* clean_off_host, int32_t* n_jobs_out, double* T_out, double* rmse_out,
                                           double* fitness_out, double* means_out, double* T_ransac_out, int64_t* ransac_stats_out,
                                           int64_t* reuse_stats_out, double* T_global_out, double* full_rmse_out, double* full_fitness_out,
                                           int32_t* best_out, void* stream) {
    if (!ctx || !det_pts4 || !det_off_dev || !det_off_host || !q_per_frame || !assn || !assn_len || !assn_count || !mem_pts4 || !mem_off_dev ||
        !mem_off_host || !mem_features || !grid || !clean_off_host || !n_jobs_out || !T_out || !rmse_out || !fitness_out || !means_out ||
        !T_global_out || !full_rmse_out || !full_fitness_out || !best_out)
        return ibl_set_error(IBL_ERR_ARG, "ibl_register_evaluate_batch: null pointer");
    if (n_frames < 0 || n_det_seg < 0 || max_assn <= 0 || voxel_size <= 0 || outlier_radius <= 0 || eval_threshold <= 0)
        return ibl_set_error(IBL_ERR_ARG, "ibl_register_evaluate_batch: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    // ---- job tables from the assignment lists (host; object_memory.py:1020-1034) ---------------------------------------------------
    std::vector<int> row0(n_frames + 1, 0);
    for (int f = 0; f < n_frames; ++f) {
        if (q_per_frame[f] < 0 || q_per_frame[f] > 7) return ibl_set_error(IBL_ERR_ARG, "ibl_register_evaluate_batch: q_per_frame out of range");
        row0[f + 1] = row0[f] + q_per_frame[f];
    }
    if (row0[n_frames] != n_det_seg) return ibl_set_error(IBL_ERR_ARG, "ibl_register_evaluate_batch: detections per frame do not add up to the segments");
    std::vector<int32_t> job_src, job_tgt;
    std::vector<int> job_frame, first_job(n_frames + 1, 0);
    for (int f = 0; f < n_frames; ++f) {
        const int na = assn_count[f];
        if (na < 0 || na > max_assn) return ibl_set_error(IBL_ERR_ARG, "ibl_register_evaluate_batch: assignment count out of range");
        for (int a = 0; a < na; ++a) {
            const int len = assn_len[(size_t)f * max_assn + a];
            if (len < 1 || len > 3) return ibl_set_error(IBL_ERR_ARG, "ibl_register_evaluate_batch: an assignment has 1 to 3 pairs");
            const int32_t* pr = assn + ((size_t)f * max_assn + a) * 6;
            for (int t = 0; t < 3; ++t) {
                if (t < len) {
                    if (pr[2 * t] < 0 || pr[2 * t] >= q_per_frame[f] || pr[2 * t + 1] < 0 || pr[2 * t + 1] >= n_mem_seg)
                        return ibl_set_error(IBL_ERR_ARG, "ibl_register_evaluate_batch: assignment index out of range");
                    job_src.push_back(row0[f] + pr[2 * t]);
                    job_tgt.push_back(pr[2 * t + 1]);
                } else {
                    job_src.push_back(-1);
                    job_tgt.push_back(-1);
                }
            }
            job_frame.push_back(f);
        }
        first_job[f + 1] = (int)job_frame.size();
    }
    const int J = (int)job_frame.size();
    *n_jobs_out = J;
    for (int f = 0; f < n_frames; ++f) best_out[f] = -1;
    if (J > max_jobs) return ibl_set_error(IBL_ERR_ARG, "ibl_register_evaluate_batch: %d jobs, the output arrays hold %d", J, max_jobs);

    // ---- clean the detected clouds (:992-998): mask, ordered compaction, new segment offsets -------------------------------------
    ArenaMark mark(ctx);
    const int n = det_off_host[n_det_seg];
    unsigned char* keep; int *pos, *new_off; float4* clean;
    IBL_ARENA(keep, unsigned char, (int64_t)n + 64);
    IBL_ARENA(pos, int, (int64_t)n + 64);
    IBL_ARENA(new_off, int, n_det_seg + 2);
    IBL_ARENA(clean, float4, (int64_t)n + 1);
    for (int i = 0; i <= n_det_seg; ++i) clean_off_host[i] = 0;
    if (n > 0) {
        int st = ibl_radius_outlier_batch(ctx, det_pts4, det_off_dev, det_off_host, n_det_seg, outlier_radius, outlier_nb_points, keep, stream);
        if (st) return st;
        IBL_HIP_CHECK(hipMemsetAsync(keep + n, 0, 1, s));           // the scan runs over n + 1 flags: pos[n] = number of kept points
        size_t tmp_bytes = 0;
        hipcub::TransformInputIterator<int, KeepToInt, const unsigned char*> flags(keep, KeepToInt());
        IBL_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, flags, pos, n + 1, s));
        unsigned char* tmp;
        IBL_ARENA(tmp, unsigned char, (int64_t)tmp_bytes + 256);
        IBL_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, flags, pos, n + 1, s));
        hipLaunchKernelGGL(ibl_compact_kept_kernel, dim3((n + 255) / 256), dim3(256), 0, s, reinterpret_cast<const float4*>(det_pts4), keep, pos, n, clean);
        IBL_LAUNCH_CHECK();
        hipLaunchKernelGGL(ibl_compact_offsets_kernel, dim3((n_det_seg + 256) / 256), dim3(256), 0, s, det_off_dev, pos, n_det_seg, new_off);
        IBL_LAUNCH_CHECK();
        IBL_HIP_CHECK(hipMemcpyAsync(clean_off_host, new_off, sizeof(int) * (size_t)(n_det_seg + 1), hipMemcpyDeviceToHost, s));
        IBL_HIP_CHECK(hipStreamSynchronize(s));
    }
    if (J == 0) return IBL_OK;
    const int nc = clean_off_host[n_det_seg];

    // ---- the detections' instance features (normals / FPFH once per cloud; the memory's are resident) -------------------------------
    float4* nrm; float* fpfh; uint16_t* split; float* fnorm;
    IBL_ARENA(nrm, float4, (int64_t)nc + 1);
    IBL_ARENA(fpfh, float, (int64_t)nc * 33 + 64);
    IBL_ARENA(split, uint16_t, (int64_t)nc * 48 + 64);
    IBL_ARENA(fnorm, float, (int64_t)nc + 64);
    std::vector<float> bbox((size_t)n_det_seg * 6 + 6, 0.0f);
    int st = ibl_instance_features_batch(ctx, reinterpret_cast<const float*>(clean), new_off, clean_off_host, n_det_seg, voxel_size, 0.0,
                                         reinterpret_cast<float*>(nrm), fpfh, split, fnorm, nullptr, bbox.data(), stream);
    if (st) return st;
    ibl_instance_features det_feat{reinterpret_cast<const float*>(nrm), fpfh, split, fnorm, nullptr, bbox.data(), voxel_size, 0.0};

    // ---- register every candidate assignment (:1036-1095) ---------------------------------------------------------------------------
    st = ibl_register_batch_cached(ctx, reinterpret_cast<const float*>(clean), new_off, clean_off_host, n_det_seg, mem_pts4, mem_off_dev,
                                   mem_off_host, n_mem_seg, job_src.data(), job_tgt.data(), J, voxel_size, global_dist_factor, local_dist_factor,
                                   seed, job_id_base, ransac_max_iter, flags, &det_feat, mem_features, T_out, rmse_out, fitness_out, means_out,
                                   T_ransac_out, ransac_stats_out, reuse_stats_out, stream);
    if (st) return st;

    // ---- global-frame transforms (:1096-1101) and the whole-memory evaluation (:1104) -------------------------------------------------
    // T maps centred detections onto centred memory clouds: G = [R | t + mean_mem - R mean_det]  (the arithmetic order of engine.py)
    std::vector<int32_t> jb(J), je(J);
    for (int j = 0; j < J; ++j) {
        const double* T = T_out + 16 * (size_t)j;
        const double* dm = means_out + 6 * (size_t)j;
        const double* mm = dm + 3;
        double* G = T_global_out + 16 * (size_t)j;
        for (int i = 0; i < 16; ++i) G[i] = T[i];
        for (int r = 0; r < 3; ++r) G[4 * r + 3] = T[4 * r + 3] + mm[r] - ((T[4 * r] * dm[0] + T[4 * r + 1] * dm[1]) + T[4 * r + 2] * dm[2]);
        const int f = job_frame[j];
        jb[j] = clean_off_host[row0[f]];
        je[j] = clean_off_host[row0[f + 1]];
    }
    st = ibl_evaluate_batch(ctx, grid, reinterpret_cast<const float*>(clean), jb.data(), je.data(), T_global_out, J, eval_threshold, full_rmse_out,
                            full_fitness_out, stream);
    if (st) return st;

    // ---- the winner of every frame: highest whole-memory fitness, the first on ties (sorted(..., reverse=True)[0], stable; :1111) ------
    for (int f = 0; f < n_frames; ++f) {
        int best = -1;
        for (int j = first_job[f]; j < first_job[f + 1]; ++j)
            if (best < 0 || full_fitness_out[j] > full_fitness_out[best]) best = j;
        best_out[f] = best < 0 ? -1 : best - first_job[f];
    }
    return IBL_OK;
}
