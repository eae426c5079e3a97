/*
 * ibloc.h -- C-ABI of libibloc_hip.so, the MI355X (gfx950) native implementation of the
 * embed -> match -> assign -> register hot path of instance-based-loc
 * (ObjectMemory.localise(), /root/reference/object_memory/object_memory.py:852-1169).
 *
 * The reference is pure Python and has no FFI of its own; every entry point below names the
 * reference interface (file:line under /root/reference) whose arithmetic it replaces.  The Python
 * host layer (instance-based-loc_amd/) binds these with ctypes and mirrors the reference's module
 * surface (utils.embeddings, utils.similarity_volume, utils.fpfh_register, object_memory).
 *
 * Conventions
 *   - every function returns an int32 status: 0 = OK, < 0 = error (ibl_last_error() has the text,
 *     thread-local).  No exception crosses the boundary.
 *   - the CALLER owns all memory.  Pointers marked [dev] are device (HBM) pointers, e.g.
 *     torch.Tensor.data_ptr(); pointers marked [host] are host pointers.  Workspace sizes come
 *     from the *_workspace_bytes() queries.  The library never allocates persistent device memory.
 *   - all launches are asynchronous on the hipStream_t passed as `void* stream`
 *     (torch.cuda.current_stream().cuda_stream); there is no hidden synchronisation unless a
 *     function's comment says it returns host results.
 *   - plain C types only; no torch types in any signature.
 */
#ifndef IBLOC_H
#define IBLOC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------ */
/* library                                                                                     */
/* ------------------------------------------------------------------------------------------ */
int ibl_version(void);
const char* ibl_last_error(void);

/* ------------------------------------------------------------------------------------------ */
/* match: L2 normalisation + closest-similarity matrix                 */
/* ------------------------------------------------------------------------------------------ */

/* Row-wise L2 normalisation, fp32.  Replaces `e/np.linalg.norm(e)` and
 * `detected_embs /= np.linalg.norm(detected_embs, axis=-1, keepdims=True)`
 * (object_memory/object_memory.py:922,924).  Summation order is fixed (64 strided partial sums
 * with fmaf, xor-butterfly 32..1, IEEE sqrt and divide) so that oracle/ reproduces it bit-exactly.
 * in/out: [dev] n_rows x dim (row stride = dim), may alias. */
int ibl_normalize_rows(const float* in, float* out, int64_t n_rows, int dim, void* stream);

/* Closest-similarity matrix: S[q][j] = max_e <mem[e], det[q]> over the stored embeddings
 * e in [emb_offsets[j], emb_offsets[j+1]) of instance j.  Replaces the Python double loop
 * object_memory/object_memory.py:933-936.  Inputs must already be L2-normalised.
 *   det         [dev] n_query x dim fp32
 *   mem         [dev] n_mem_rows x dim fp32       (all stored embeddings, instance-major)
 *   emb_offsets [dev] (n_inst + 1) int32, emb_offsets[0] = 0, emb_offsets[n_inst] = n_mem_rows
 *   out_sims    [dev] n_query x n_inst fp32 or NULL
 *   out_aug     [dev] n_query x (n_inst + 1) IEEE binary16 bits or NULL: the reference's
 *               `aug = [sims | 1]` cast to float16 (utils/similarity_volume.py:13-18)
 *   workspace   [dev] ibl_closest_similarity_workspace_bytes() bytes
 * The dot product is computed on the fp32-input MFMA (v_mfma_f32_32x32x2_f32), which is a
 * k-ordered fmaf chain; the k order is the fixed permutation documented in DESIGN.md and
 * restated by oracle/, so the result is bit-exact against the oracle. dim % 8 == 0 required. */
int64_t ibl_closest_similarity_workspace_bytes(int64_t n_query, int64_t n_mem_rows);
int ibl_closest_similarity(const float* det, int64_t n_query, const float* mem, int64_t n_mem_rows,
                           const int32_t* emb_offsets, int64_t n_inst, int dim, float* out_sims,
                           uint16_t* out_aug, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* assign: exact similarity-volume search                                   */
/* ------------------------------------------------------------------------------------------ */

/* Replaces SimVolume(sims).fast_construct_volume(min(Q,3)) +
 * .get_top_indices_from_subvolumes(num_per_length)  (utils/similarity_volume.py:13-18,102-164,
 * 213-270; called at object_memory/object_memory.py:974-982).  Bit-exact assignment lists
 * without materialising the (M+1)^3 volumes.  Host function (integer / fp16 index logic).
 *   aug_half     [host] n_frames x q_stride x (M+1) IEEE binary16 bits ([sims | 1])
 *   q_per_frame  [host] n_frames, number of valid detection rows of each frame (<= q_stride)
 *   out_assn     [host] n_frames x max_assn x 3 x 2 int32: (detection idx, memory idx), -1 padded
 *   out_len      [host] n_frames x max_assn: pairs in each assignment (1..3)
 *   out_count    [host] n_frames: assignments of each frame (<= 6)
 *   max_assn     >= 6;  n_threads: host threads to spread frames over. */
int ibl_assign_batch(const uint16_t* aug_half, const int32_t* q_per_frame, int n_frames, int q_stride,
                     int M, int num_per_length, int32_t* out_assn, int32_t* out_len, int32_t* out_count,
                     int max_assn, int n_threads);

#ifdef __cplusplus
}
#endif
#endif /* IBLOC_H */
