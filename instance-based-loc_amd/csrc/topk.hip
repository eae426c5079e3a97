// topk.hip -- per-row two-ended candidate selection on the fp16 similarity rows (the device half of the assignment search).
//
// The reference hands the whole (Q, M) closest-similarity matrix to SimVolume (object_memory/object_memory.py:974-982), which only
// ever returns cells whose coordinates sit at one of the two ends of their row's value order (utils/similarity_volume.py:102-164:
// the score is a chained fp16 product, monotone in every coordinate).  So the device keeps, per query row, the K_hi largest and the
// K_lo smallest entries of `aug` under the TOTAL order (value, then lower memory index first) -- the order np.argmax's tie rule
// induces -- and only those (value, global index) pairs leave the GPU (8 bytes each; 1.8 KB per row instead of 2 (M + 1) bytes) or
// cross xGMI when the memory is sharded by instance range.  The host search (assign.cpp, ibl_assign_candidates) runs on
// the candidates and PROVES per frame that no dropped entry could have reached the k-th best cell; a frame whose proof fails is
// redone on the full rows, so the result is always that of the full search.
//
// Kernel: one workgroup per row, three passes over the row's fp16 values (L2-resident: M = 10 000 is 20 KB): (1) histogram of the
// high byte of an order-preserving 16-bit key, (2) histogram of the low byte inside the two boundary bins -> the exact threshold
// keys and how many entries of the threshold's tie class belong to the selection, (3) ordered compaction (tie class in index
// order) into LDS, bitonic sort by (key, index), store.  HBM-bound in principle (2 B per entry and pass); the rows are small.
#include <hip/hip_runtime.h>

#include "ibl_common.h"

namespace {

constexpr int SEL_THREADS = 256;
constexpr int SEL_MAXK = 256;         // K_hi, K_lo <= 256 each

// IEEE binary16 bits -> 16-bit key whose unsigned order is the float order; -0 is folded onto +0 (np.argmax compares values)
__device__ __forceinline__ unsigned key_of(unsigned h) {
    if (h == 0x8000u) h = 0;
    return (h & 0x8000u) ? (~h & 0xFFFFu) : (h | 0x8000u);
}
__device__ __forceinline__ unsigned half_of(unsigned key) { return (key & 0x8000u) ? (key & 0x7FFFu) : (~key & 0xFFFFu); }

// block-wide exclusive prefix of a per-thread flag, in thread order; returns the block total through `total`
__device__ __forceinline__ int block_rank(bool flag, int* wave_cnt, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();                                  // wave_cnt is reused by consecutive calls
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int base = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < SEL_THREADS / 64; ++w) {
        if (w < wave) base += wave_cnt[w];
        total += wave_cnt[w];
    }
    return base + before;
}

// sorts n <= SEL_MAXK 64-bit keys ascending in LDS (padded to a power of two with ~0)
__device__ void bitonic_sort(unsigned long long* a, int n) {
    int p = 1;
    while (p < n) p <<= 1;
    for (int i = n + threadIdx.x; i < p; i += SEL_THREADS) a[i] = ~0ull;
    __syncthreads();
    for (int k = 2; k <= p; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < p; i += SEL_THREADS) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long x = a[i], y = a[l];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { a[i] = y; a[l] = x; }
                }
            }
            __syncthreads();
        }
}

__global__ __launch_bounds__(SEL_THREADS) void ibl_topk_select_kernel(const uint16_t* __restrict__ aug, int64_t ld, int n_cols, int k_hi,
                                                                      int k_lo, int index_base, uint16_t* __restrict__ out_val,
                                                                      int32_t* __restrict__ out_idx, int32_t* __restrict__ out_cnt) {
    __shared__ int hist[2][256];
    __shared__ int wave_cnt[SEL_THREADS / 64];
    __shared__ int s_bin[2], s_above[2], s_thr[2], s_need[2], s_fill[2];
    __shared__ unsigned long long sel[2][SEL_MAXK];
    const int tid = threadIdx.x;
    const int64_t row = blockIdx.x;
    const uint16_t* v = aug + row * ld;
    const int S = k_hi + k_lo;
    uint16_t* oval = out_val + row * S;
    int32_t* oidx = out_idx + row * S;

    if (n_cols <= S) {
        // the whole row is the candidate set: descending (value, index) order, no low list
        __shared__ unsigned long long all[2 * SEL_MAXK];
        for (int i = tid; i < n_cols; i += SEL_THREADS) all[i] = ((unsigned long long)(0xFFFFu - key_of(v[i])) << 32) | (unsigned)i;
        __syncthreads();
        // 2 * SEL_MAXK entries at most: two-buffer bitonic on the first power of two
        int p = 1;
        while (p < n_cols) p <<= 1;
        for (int i = n_cols + tid; i < p; i += SEL_THREADS) all[i] = ~0ull;
        __syncthreads();
        for (int k = 2; k <= p; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < p; i += SEL_THREADS) {
                    const int l = i ^ j;
                    if (l > i) {
                        const unsigned long long x = all[i], y = all[l];
                        const bool up = (i & k) == 0;
                        if ((x > y) == up) { all[i] = y; all[l] = x; }
                    }
                }
                __syncthreads();
            }
        for (int i = tid; i < n_cols; i += SEL_THREADS) {
            const unsigned key = 0xFFFFu - (unsigned)(all[i] >> 32);
            oval[i] = (uint16_t)half_of(key);
            oidx[i] = index_base + (int)(all[i] & 0xFFFFFFFFu);
        }
        if (tid == 0) { out_cnt[2 * row] = n_cols; out_cnt[2 * row + 1] = 0; }
        return;
    }

    // ---- pass 1: high-byte histogram ----------------------------------------------------------------------------------------
    hist[0][tid] = 0;
    __syncthreads();
    for (int i = tid; i < n_cols; i += SEL_THREADS) atomicAdd(&hist[0][key_of(v[i]) >> 8], 1);
    __syncthreads();
    if (tid == 0) {
        int acc = 0, b = 255;                                   // from the top: the bin in which the k_hi-th largest falls
        for (; b > 0 && acc + hist[0][b] < k_hi; --b) acc += hist[0][b];
        s_bin[0] = b; s_above[0] = acc;
        acc = 0; b = 0;                                         // from the bottom: the k_lo-th smallest
        for (; b < 255 && acc + hist[0][b] < k_lo; ++b) acc += hist[0][b];
        s_bin[1] = b; s_above[1] = acc;
    }
    __syncthreads();
    const int bin_hi = s_bin[0], bin_lo = s_bin[1];
    // ---- pass 2: low-byte histograms inside the two boundary bins --------------------------------------------------------------
    hist[0][tid] = 0;
    hist[1][tid] = 0;
    __syncthreads();
    for (int i = tid; i < n_cols; i += SEL_THREADS) {
        const unsigned k = key_of(v[i]);
        if ((int)(k >> 8) == bin_hi) atomicAdd(&hist[0][k & 255], 1);
        if ((int)(k >> 8) == bin_lo) atomicAdd(&hist[1][k & 255], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int acc = s_above[0], b = 255;
        for (; b > 0 && acc + hist[0][b] < k_hi; --b) acc += hist[0][b];
        s_thr[0] = (bin_hi << 8) | b;                           // threshold key: entries above it are all taken (acc of them)
        s_need[0] = k_hi - acc;                                 // ... plus this many of the tie class, lowest indices first
        acc = s_above[1]; b = 0;
        for (; b < 255 && acc + hist[1][b] < k_lo; ++b) acc += hist[1][b];
        s_thr[1] = (bin_lo << 8) | b;
        s_need[1] = k_lo - acc;
        s_fill[0] = s_fill[1] = 0;
    }
    __syncthreads();
    const unsigned thr_hi = (unsigned)s_thr[0], thr_lo = (unsigned)s_thr[1];
    const int need_hi = s_need[0], need_lo = s_need[1];
    // ---- pass 3: compaction; the tie classes in index order ---------------------------------------------------------------------
    int tie_hi_seen = 0, tie_lo_seen = 0;                       // entries of the tie classes in earlier chunks (block-uniform)
    for (int c0 = 0; c0 < n_cols; c0 += SEL_THREADS) {
        const int i = c0 + tid;
        const bool in = i < n_cols;
        const unsigned k = in ? key_of(v[i]) : 0u;
        const bool gt = in && k > thr_hi, eqh = in && k == thr_hi;
        const bool lt = in && k < thr_lo, eql = in && k == thr_lo;
        int tot_h, tot_l;
        const int rh = block_rank(eqh, wave_cnt, tot_h);
        const int rl = block_rank(eql, wave_cnt, tot_l);
        if (gt || (eqh && tie_hi_seen + rh < need_hi)) {
            const int slot = atomicAdd(&s_fill[0], 1);
            sel[0][slot] = ((unsigned long long)(0xFFFFu - k) << 32) | (unsigned)i;        // ascending = value desc, index asc
        }
        if (lt || (eql && tie_lo_seen + rl < need_lo)) {
            const int slot = atomicAdd(&s_fill[1], 1);
            sel[1][slot] = ((unsigned long long)k << 32) | (unsigned)i;                    // ascending = value asc, index asc
        }
        tie_hi_seen += tot_h;
        tie_lo_seen += tot_l;
    }
    __syncthreads();
    bitonic_sort(sel[0], k_hi);
    bitonic_sort(sel[1], k_lo);
    for (int i = tid; i < k_hi; i += SEL_THREADS) {
        oval[i] = (uint16_t)half_of(0xFFFFu - (unsigned)(sel[0][i] >> 32));
        oidx[i] = index_base + (int)(sel[0][i] & 0xFFFFFFFFu);
    }
    for (int i = tid; i < k_lo; i += SEL_THREADS) {
        oval[k_hi + i] = (uint16_t)half_of((unsigned)(sel[1][i] >> 32));
        oidx[k_hi + i] = index_base + (int)(sel[1][i] & 0xFFFFFFFFu);
    }
    if (tid == 0) { out_cnt[2 * row] = k_hi; out_cnt[2 * row + 1] = k_lo; }
}

}  // namespace

extern "C" int ibl_topk_select(const uint16_t* aug, int64_t n_rows, int64_t ld, int n_cols, int k_hi, int k_lo, int index_base,
                               uint16_t* out_val, int32_t* out_idx, int32_t* out_cnt, void* stream) {
    if (!aug || !out_val || !out_idx || !out_cnt) return ibl_set_error(IBL_ERR_ARG, "ibl_topk_select: null pointer");
    if (n_rows < 0 || n_cols <= 0 || ld < n_cols || k_hi <= 0 || k_lo < 0 || k_hi > SEL_MAXK || k_lo > SEL_MAXK)
        return ibl_set_error(IBL_ERR_ARG, "ibl_topk_select: bad sizes (1 <= k_hi <= 256, 0 <= k_lo <= 256)");
    if (n_rows == 0) return IBL_OK;
    hipLaunchKernelGGL(ibl_topk_select_kernel, dim3((unsigned)n_rows), dim3(SEL_THREADS), 0, (hipStream_t)stream, aug, ld, n_cols, k_hi, k_lo,
                       index_base, out_val, out_idx, out_cnt);
    IBL_LAUNCH_CHECK();
    return IBL_OK;
}

extern "C" int ibl_closest_similarity(const float* det, int64_t n_query, const float* mem, int64_t n_mem_rows, const int32_t* emb_offsets,
                                      int64_t n_inst, int dim, float* out_sims, uint16_t* out_aug, void* workspace, int64_t workspace_bytes,
                                      void* stream);
extern "C" int64_t ibl_closest_similarity_workspace_bytes(int64_t n_query, int64_t n_mem_rows);

extern "C" int64_t ibl_match_topk_workspace_bytes(int64_t n_query, int64_t n_mem_rows, int64_t n_inst) {
    if (n_query < 0 || n_mem_rows < 0 || n_inst < 0) return -1;
    return ibl_closest_similarity_workspace_bytes(n_query, n_mem_rows) + ibl_align_up(n_query * (n_inst + 1) * 2, 256) + 256;
}

extern "C" int ibl_match_topk(const float* det, int64_t n_query, const float* mem, int64_t n_mem_rows, const int32_t* emb_offsets,
                              int64_t n_inst, int dim, int k_hi, int k_lo, int index_base, uint16_t* out_val, int32_t* out_idx,
                              int32_t* out_cnt, uint16_t* out_aug, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!workspace || workspace_bytes < ibl_match_topk_workspace_bytes(n_query, n_mem_rows, n_inst))
        return ibl_set_error(IBL_ERR_ARG, "ibl_match_topk: workspace too small");
    unsigned char* p = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    uint16_t* aug = out_aug;
    if (!aug) {
        aug = reinterpret_cast<uint16_t*>(p);
        p += ibl_align_up(n_query * (n_inst + 1) * 2, 256);
    }
    const int64_t rest = workspace_bytes - (p - reinterpret_cast<unsigned char*>(workspace));
    int st = ibl_closest_similarity(det, n_query, mem, n_mem_rows, emb_offsets, n_inst, dim, nullptr, aug, p, rest, stream);
    if (st) return st;
    return ibl_topk_select(aug, n_query, n_inst + 1, (int)n_inst, k_hi, k_lo, index_base, out_val, out_idx, out_cnt, stream);
}
