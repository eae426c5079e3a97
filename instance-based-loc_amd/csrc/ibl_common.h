// Shared internals of libibloc_hip.so (status codes, error channel, launch checks).
#pragma once
#include <cstdint>
#include <cstdio>

#define IBL_OK 0
#define IBL_ERR_ARG (-1)
#define IBL_ERR_HIP (-2)
#define IBL_ERR_INTERNAL (-3)
#define IBL_ERR_UNSUPPORTED (-4)

// records a thread-local message retrievable through ibl_last_error(); returns `code`
int ibl_set_error(int code, const char* fmt, ...);

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define IBL_HIP_CHECK(expr)                                                                         \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess)                                                                       \
            return ibl_set_error(IBL_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                 __FILE__, __LINE__);                                               \
    } while (0)
#define IBL_LAUNCH_CHECK() IBL_HIP_CHECK(hipGetLastError())
#endif

// kernel-family ids of the in-process timer (ibl_prof_*), units = algorithmic FLOPs or bytes per launch
#define IBL_PROF_GEMM 1        // ibl_gemm_bf16_tn (all epilogues): units = 2*M*N*K FLOPs
#define IBL_PROF_ATTN 2        // ibl_attention_kernel: 4*T*T*64 FLOPs per (crop, head)
#define IBL_PROF_SPFH 3        // ibl_spfh_kernel: 156 B per point (24 in + 132 out)
#define IBL_PROF_NORMALS 4     // ibl_normals_kernel: 24 B per point
#define IBL_PROF_FPFH 5        // ibl_fpfh_kernel: 264 B per point
#define IBL_PROF_FEATNN 6      // ibl_feat_nn_kernel: 2*33*Ns*Nt FLOPs per direction
#define IBL_PROF_ICP 7         // ibl_icp_step_kernel: 56 B per source point
#define IBL_PROF_RANSAC 8      // ibl_ransac_flag_kernel: hypotheses
#define IBL_PROF_ROWSIM 9      // ibl_rowsim_kernel: 2*rows*queries*dim FLOPs
// whole stages of the registration side (bench.py's per-stage roofline): one event pair around the stage's launches
#define IBL_PROF_ST_FEATURES 10   // ibl_instance_features_batch: 444 B per point (normals 24 + SPFH 156 + FPFH 264)
#define IBL_PROF_ST_FEATMATCH 11  // feature search of a registration call: 4*33*Ns*Nt FLOPs per job (both directions)
#define IBL_PROF_ST_RANSAC 12     // RANSAC of a registration call: hypotheses walked (all jobs)
#define IBL_PROF_ST_ICP 13        // ICP of a registration call: 56 B per source point and iteration run
#define IBL_PROF_ST_EVAL 14       // ibl_evaluate_*: 24 B per detected point and candidate
#define IBL_PROF_ST_OUTLIER 15    // ibl_radius_outlier_batch: 12 B in + 1 B out per point
#define IBL_PROF_MAX 24
int ibl_prof_enabled();
void ibl_prof_begin(int id, double units, void* stream, void** token);
void ibl_prof_end(void* token, void* stream);
void ibl_prof_set_units(void* token, double units);      // units known only after the stage ran (iterations, hypotheses)

static inline int64_t ibl_align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
