// match.hip -- embedding L2 normalisation and the closest-similarity matrix on gfx950.
//
// Replaces object_memory/object_memory.py:922-936 of the reference (normalise every stored and
// detected embedding; S[i][j] = max_e <mem_j,e , det_i>).  The dot products run on the fp32-input
// MFMA (v_mfma_f32_32x32x2_f32): exact f32, a k-ordered fmaf chain, so the result is bit-for-bit
// reproducible by a scalar CPU loop that walks k in the same order (oracle/oracle_match.c).
//
// Data layout in HBM: embeddings row-major fp32 [rows][dim]; the row-similarity scratch
// R[mem_row][query] (query contiguous) lets the MFMA accumulators be stored as 128-byte rows and
// the per-instance max be taken with coalesced reads.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "ibl_common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

// ------------------------------------------------------------------------------------------------
// L2 normalisation: one wave per row.
// canonical order: lane l accumulates x[l], x[l+64], ... with fmaf; xor-butterfly 32,16,...,1.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ibl_normalize_rows_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                 int64_t n_rows, int dim) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const float* x = in + row * dim;
    float s = 0.0f;
    for (int i = lane; i < dim; i += 64) s = __builtin_fmaf(x[i], x[i], s);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s = s + __shfl_xor(s, off, 64);
    const float nrm = __builtin_sqrtf(s);   // correctly rounded (default -fhip-fp32-correctly-rounded-divide-sqrt)
    float* y = out + row * dim;
    for (int i = lane; i < dim; i += 64) y[i] = x[i] / nrm;
}

extern "C" int ibl_normalize_rows(const float* in, float* out, int64_t n_rows, int dim, void* stream) {
    if (!in || !out || n_rows < 0 || dim <= 0) return ibl_set_error(IBL_ERR_ARG, "ibl_normalize_rows: bad argument");
    if (n_rows == 0) return IBL_OK;
    dim3 grid((unsigned)((n_rows + 3) / 4));
    hipLaunchKernelGGL(ibl_normalize_rows_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, out, n_rows, dim);
    IBL_LAUNCH_CHECK();
    return IBL_OK;
}

// ------------------------------------------------------------------------------------------------
// R[mem_row][query] = <mem[mem_row], det[query]>, f32 MFMA 32x32x2.
// A operand: lane l holds A[i = l&31][k = l>>5];  B operand: B[k = l>>5][j = l&31].
// Lane (i, h) loads the 4 floats  x[8m + 4h .. 8m + 4h + 3]  of its row; MFMA c (c = 0..3) then
// consumes the k pair (8m + c, 8m + 4 + c).  Chain order inside each block of 8: 0,4,1,5,2,6,3,7.
// ------------------------------------------------------------------------------------------------
template <int NQ>
__global__ __launch_bounds__(256) void ibl_rowsim_kernel(const float* __restrict__ mem, int64_t n_rows,
                                                         const float* __restrict__ det, int64_t n_query, int dim,
                                                         float* __restrict__ R, int64_t r_stride) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    const int64_t row0 = tile * 32;
    if (row0 >= n_rows) return;
    const int64_t q0 = (int64_t)blockIdx.y * (32 * NQ);

    int64_t arow = row0 + i;
    if (arow >= n_rows) arow = n_rows - 1;            // clamped loads, stores are guarded
    const float* ap = mem + arow * dim + 4 * h;
    const float* bp[NQ];
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        int64_t q = q0 + 32 * t + i;
        if (q >= n_query) q = n_query - 1;
        bp[t] = det + q * dim + 4 * h;
    }
    f32x16 acc[NQ];
#pragma unroll
    for (int t = 0; t < NQ; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    for (int m = 0; m < dim; m += 8) {
        const float4 a4 = *reinterpret_cast<const float4*>(ap + m);
        float4 b4[NQ];
#pragma unroll
        for (int t = 0; t < NQ; ++t) b4[t] = *reinterpret_cast<const float4*>(bp[t] + m);
#pragma unroll
        for (int t = 0; t < NQ; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4[t].x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4[t].y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4[t].z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4[t].w, acc[t], 0, 0, 0);
        }
    }
    // C/D layout: col = lane & 31 (query), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        const int64_t q = q0 + 32 * t + i;
        if (q >= n_query) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < n_rows) R[row * r_stride + q] = acc[t][r];
        }
    }
}

// S[q][j] = max over the instance's rows; 32 x 32 (q x j) tile transposed through LDS
__global__ __launch_bounds__(256) void ibl_segmax_kernel(const float* __restrict__ R, int64_t r_stride,
                                                         const int32_t* __restrict__ emb_offsets, int64_t n_inst,
                                                         int64_t n_query, float* __restrict__ out_sims,
                                                         uint16_t* __restrict__ out_aug) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int64_t j0 = (int64_t)blockIdx.x * 32, q0 = (int64_t)blockIdx.y * 32;
    const int64_t q = q0 + tx;
    for (int jj = ty; jj < 32; jj += 8) {
        const int64_t j = j0 + jj;
        float v = -INFINITY;
        if (j < n_inst && q < n_query) {
            const int e0 = emb_offsets[j], e1 = emb_offsets[j + 1];
            for (int e = e0; e < e1; ++e) v = fmaxf(v, R[(int64_t)e * r_stride + q]);
        }
        tile[jj][tx] = v;
    }
    __syncthreads();
    const int64_t j = j0 + tx;
    for (int qq = ty; qq < 32; qq += 8) {
        const int64_t qo = q0 + qq;
        if (qo >= n_query) continue;
        if (j < n_inst) {
            const float v = tile[tx][qq];
            if (out_sims) out_sims[qo * n_inst + j] = v;
            if (out_aug) out_aug[qo * (n_inst + 1) + j] = __half_as_ushort(__float2half_rn(v));
        }
        if (out_aug && blockIdx.x == 0 && tx == 0) out_aug[qo * (n_inst + 1) + n_inst] = 0x3C00;  // 1.0
    }
}

static inline int64_t pad32(int64_t x) { return (x + 31) / 32 * 32; }

extern "C" int64_t ibl_closest_similarity_workspace_bytes(int64_t n_query, int64_t n_mem_rows) {
    if (n_query < 0 || n_mem_rows < 0) return -1;
    return n_mem_rows * pad32(n_query) * (int64_t)sizeof(float) + 256;
}

extern "C" int ibl_closest_similarity(const float* det, int64_t n_query, const float* mem, int64_t n_mem_rows,
                                      const int32_t* emb_offsets, int64_t n_inst, int dim, float* out_sims,
                                      uint16_t* out_aug, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!det || !mem || !emb_offsets || !workspace)
        return ibl_set_error(IBL_ERR_ARG, "ibl_closest_similarity: null pointer");
    if (n_query <= 0 || n_mem_rows <= 0 || n_inst <= 0 || dim <= 0 || (dim % 8) != 0)
        return ibl_set_error(IBL_ERR_ARG, "ibl_closest_similarity: sizes must be positive and dim %% 8 == 0");
    if (workspace_bytes < ibl_closest_similarity_workspace_bytes(n_query, n_mem_rows))
        return ibl_set_error(IBL_ERR_ARG, "ibl_closest_similarity: workspace too small");
    if ((reinterpret_cast<uintptr_t>(det) | reinterpret_cast<uintptr_t>(mem)) & 15)
        return ibl_set_error(IBL_ERR_ARG, "ibl_closest_similarity: embeddings must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    float* R = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const int64_t r_stride = pad32(n_query);
    const int64_t tiles = (n_mem_rows + 31) / 32;
    constexpr int NQ = 4;
    dim3 grid((unsigned)((tiles + 3) / 4), (unsigned)((n_query + 32 * NQ - 1) / (32 * NQ)));
    hipLaunchKernelGGL(ibl_rowsim_kernel<NQ>, grid, dim3(256), 0, s, mem, n_mem_rows, det, n_query, dim, R, r_stride);
    IBL_LAUNCH_CHECK();
    dim3 grid2((unsigned)((n_inst + 31) / 32), (unsigned)((n_query + 31) / 32));
    hipLaunchKernelGGL(ibl_segmax_kernel, grid2, dim3(256), 0, s, R, r_stride, emb_offsets, n_inst, n_query, out_sims,
                       out_aug);
    IBL_LAUNCH_CHECK();
    return IBL_OK;
}
