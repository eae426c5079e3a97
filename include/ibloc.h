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
 *     (torch.cuda.current_stream().cuda_stream).  Functions whose results are DEVICE arrays (embed, match, candidate selection)
 *     never synchronise.  Functions that return HOST results synchronise the stream -- how often is stated per function:
 *     ibl_radius_outlier_batch 1 (grid table size), ibl_instance_features_batch 2 (bounding boxes; end),
 *     ibl_register_batch_cached 2-4 (one per group of RANSAC rounds -- most calls need one -- plus the results; one more when
 *     instances of a job lie within the influence radius of each other), ibl_evaluate_batch 1, ibl_memgrid_build 2 (once per
 *     memory).  Plan tables are staged through pinned host memory of the registration context, so uploads never wait.
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

/* In-process kernel timer for the roofline line of bench.py: when enabled, selected kernel families are
 * bracketed by HIP events ON THEIR LAUNCH STREAM.  ibl_prof_read synchronises those events and returns the
 * accumulated device time, the accumulated algorithmic units (FLOPs or bytes, see csrc/ibl_common.h) and the
 * number of launches of family `id` (1 = fp16 GEMM, 3 = SPFH k-NN, ...). */
int ibl_prof_enable(int on);
int ibl_prof_read(int id, double* ms, double* units, int64_t* launches);

/* ------------------------------------------------------------------------------------------ */
/* embed: crop preprocessing + ViT encoder forward                       */
/* ------------------------------------------------------------------------------------------ */

#define IBL_VIT_MAX_LAYERS 32
#define IBL_VIT_LAYERSCALE 1      /* DINOv2 LayerScale (per-layer ls1/ls2 non-NULL)                */
#define IBL_VIT_PRE_LN 2          /* CLIP ln_pre on the embedded tokens                            */
#define IBL_VIT_FINAL_LN 4        /* final LayerNorm (DINOv2 / ViT layernorm, CLIP ln_post)        */
#define IBL_VIT_QUICK_GELU 8      /* x*sigmoid(1.702x) (OpenAI CLIP); laion2b ViT-B-32 uses GELU   */
#define IBL_VIT_PROJ 16           /* CLS -> out_dim projection (CLIP visual.proj)                  */
#define IBL_VIT_OUT_ALL_TOKENS 32 /* return every token (DATOR/TransReID local_feature=True)       */
#define IBL_VIT_ACT_TERMS2 64     /* some block has o_terms / fc2_terms = 2: attention output / hidden layer as rows of 2 terms  */
#define IBL_VIT_ACT_TERMS3 128    /* ... = 3: rows of 3 terms (sizes the workspace)                                             */
#define IBL_VIT_SPLIT_SCALE 64.0f /* S of the two-term operands below                              */

typedef struct {
    int32_t dim, depth, heads, mlp_dim;
    int32_t patch, img_h, img_w;   /* model input size after preprocessing                          */
    int32_t n_tokens;              /* 1 + (img_h/patch)*(img_w/patch)                               */
    int32_t patch_k_pad;           /* 3*patch*patch rounded up to a multiple of 64 (zero padded)    */
    int32_t flags;                 /* IBL_VIT_*                                                     */
    int32_t n_blocks_run;          /* blocks actually executed (DATOR runs depth-1)                 */
    int32_t out_dim;               /* dim, or the projection width                                  */
    float ln_eps;
} ibl_vit_desc;

typedef struct {                   /* all [dev]; weights fp16 [N][K] row-major (nn.Linear layout)   */
    const float* ln1_g; const float* ln1_b;
    const void* w_qkv;  const float* b_qkv;     /* [3*dim][dim] = [Wq; Wk; Wv], [3*dim]              */
    const void* w_o;    const float* b_o;       /* [dim][dim]                                         */
    const float* ls1;                            /* [dim] or NULL.  NULL (no LayerScale in the model, or ls1 multiplied into the rows of
                                                  * w_o and into b_o by the host, as ibloc_amd.vit does) selects the faster residual GEMM:
                                                  * the residual tile is preloaded into the accumulators and the epilogue only stores   */
    const float* ln2_g; const float* ln2_b;
    const void* w_fc1;  const float* b_fc1;     /* [mlp_dim][dim]                                     */
    const void* w_fc2;  const float* b_fc2;     /* [dim][mlp_dim]                                     */
    const float* ls2;
    /* Optional two-term fp16 operands (round 3; all NULL / 0 = plain fp16 operands, what rounds 1-2 ran).  With random-init
     * weights the first blocks carry most of the encoder's fp16 rounding error (the residual stream is still small there), so
     * the host may hand those blocks their weights as W = W_hi + W_lo (both fp16, W_lo stored times IBL_VIT_SPLIT_SCALE so it
     * stays a normal number) and ask for the LayerNorm output as a_hi + a_lo as well.  K-extended operand rows:
     *   terms 2: A' = [a_hi | a_hi / S],             W' = [W_hi | W_lo * S]              (weights exact to ~2^-22)
     *   terms 3: A' = [a_hi | a_lo * S | a_hi / S],  W' = [W_hi | W_hi / S | W_lo * S]   (+ the activation's second term)
     * one accumulation, same kernel, K' = terms * dim.  The two residual GEMMs (linear epilogue) take the second weight term
     * as a second launch that accumulates scale_lo[n] * (a W_lo'^T) into the residual (scale_lo = LayerScale / S, or 1 / S). */
    const void* w_qkv_x;                         /* fp16 [3*dim][qkv_terms*dim] or NULL                */
    const void* w_o_lo;  const float* ls1_lo;   /* fp16 [dim][dim] = W_lo * S; fp32 [dim] = ls1 / S, or NULL: the term is added with the
                                                  * factor 1 / S (no LayerScale, or LayerScale folded into W_o by the host)            */
    const void* w_fc1_x;                         /* fp16 [mlp_dim][fc1_terms*dim] or NULL              */
    const void* w_fc2_lo; const float* ls2_lo;  /* fp16 [dim][mlp_dim]; fp32 [dim]                    */
    int32_t qkv_terms, fc1_terms;                /* 1 (or 0), 2 or 3                                   */
    /* round 4: the two residual GEMMs with K-extended operands as well -- ONE launch of K' = terms * K instead of one read-modify-write pass
     * over the residual per term: the attention kernel / the fc1 epilogue write the rows [a_hi | a_hi / S] (terms 2) or
     * [a_hi | a_lo * S | a_hi / S] (terms 3: + the second fp16 term of the attention output / of the GELU hidden layer); w_o_x / w_fc2_x are laid
     * out like w_qkv_x.  Needs IBL_VIT_ACT_TERMS2 / 3 in the descriptor's flags.  With w_o_x / w_fc2_x NULL the older w_o_lo / w_fc2_lo launches run. */
    int32_t o_terms, fc2_terms;
    const void* w_o_x;                           /* fp16 [dim][o_terms*dim] or NULL                     */
    const void* w_fc2_x;                         /* fp16 [dim][fc2_terms*mlp_dim] or NULL               */
} ibl_vit_layer;

typedef struct {
    const void* w_patch;           /* fp16 [dim][patch_k_pad]: conv weight flattened (c, kh, kw)     */
    const float* b_patch;          /* [dim] or NULL                                                  */
    const float* cls_pos;          /* [dim]: cls_token + position_embedding[0]                       */
    const float* pos_patch;        /* [n_tokens-1][dim]: position embeddings of the patch tokens,
                                      already interpolated to the (img_h/patch, img_w/patch) grid    */
    const float* ln_pre_g; const float* ln_pre_b;
    const float* ln_f_g;   const float* ln_f_b;
    const void* w_proj;            /* fp16 [out_dim][dim] or NULL                                    */
    const void* w_patch_lo;        /* optional fp16 [dim][patch_k_pad] = (W - fp16(W)) * S: second term of the patch weights  */
    const void* w_proj_x;          /* optional fp16 [out_dim][3*dim] = [W_hi | W_hi / S | W_lo * S]: the projection with three-term
                                      operands (round 4: the final LayerNorm row is written as [a_hi | a_lo * S | a_hi / S]); the
                                      projection is the last arithmetic before the output, so its fp16 roundings are not attenuated
                                      by anything downstream (4e-4 of the embedding with plain operands) and it costs nothing     */
    ibl_vit_layer layers[IBL_VIT_MAX_LAYERS];
} ibl_vit_weights;

/* One crop of a preprocessing batch.  The separable resample coefficient tables are the
 * fixed-point (22-bit) tables of Pillow's 8-bit resampler, computed on the host in float64
 * (instance-based-loc_amd/preprocess.py) so that the device arithmetic is pure integer and
 * bit-identical to PIL.Image.resize -- the resize every reference embedding function applies
 * through its HF / open_clip processor (utils/embeddings.py:41-42, 64-65, 86-89).
 * Table layout per pass: out_size records of (2 + ksize) int32 = [first_tap, n_taps, k0 .. ]. */
typedef struct {
    int64_t src_offset;            /* byte offset of the crop (HWC u8, 3 channels) in `src`          */
    int32_t in_h, in_w;
    int32_t h_table, h_ksize;      /* int32 index into `tables` of the horizontal pass, taps/record  */
    int32_t v_table, v_ksize;      /* vertical pass (NULL pass = identity: ksize 0)                  */
    int64_t tmp_offset;            /* byte offset of this crop's [in_h][out_w][3] scratch in `tmp`   */
} ibl_crop_desc;

/* One pass of Pillow's 8-bit resampler as a fixed-point table (host function; libImaging/Resample.c precompute_coeffs +
 * normalize_coeffs_8bpc restated): output samples [win0, win0 + win_n) of a resize in_size -> out_size.  rec: [HOST] win_n records of
 * (2 + ksize) int32 = [first tap, tap count, 22-bit weights ...]; ksize = ibl_resample_ksize(...).  Returns ksize (> 0) or a negative
 * status.  What the reference's HF / open_clip processors compute per crop on the CPU (utils/embeddings.py:41-42, 64-65, 86-89). */
#define IBL_FILTER_BILINEAR 0
#define IBL_FILTER_BICUBIC 1
int ibl_resample_ksize(int in_size, int out_size, int filter);
int ibl_resample_table(int in_size, int out_size, int filter, int win0, int win_n, int32_t* rec);

/* u8 crops -> (optional R<->B swap, utils/embeddings.py:41,64,86) -> PIL-exact resize ->
 * window (centre crop) -> ((u8/255) - mean) / std -> fp16 im2col patch matrix
 * [n_crops * (out_h/patch)*(out_w/patch)][patch_k_pad], k = c*patch*patch + kh*patch + kw.
 *   max_in_h: tallest crop of the batch (sizes the launch);
 *   src, tables, descs, tmp: [dev];  mean/stdv: 3 floats each (host), indexed by MODEL channel
 *   out_u8: optional [dev] n_crops x out_h x out_w x 3 resized+cropped u8 image (for parity tests) */
int ibl_preprocess_crops(const uint8_t* src, const ibl_crop_desc* descs, int n_crops, int max_in_h,
                         const int32_t* tables,
                         uint8_t* tmp, int out_h, int out_w, int patch, int patch_k_pad, int swap_rb,
                         const float* mean, const float* stdv, void* patches, uint8_t* out_u8, void* stream);

/* ViT forward over a batch of crops.  Replaces the batch-1 torch forward of
 * utils/embeddings.py:46,69,93 and dator/model/backbones/vit_pytorch.py:422-443.
 *   patches [dev] fp16 [batch*(n_tokens-1)][patch_k_pad] (from ibl_preprocess_crops)
 *   out     [dev] fp32 [batch][out_dim]  (CLS embedding; un-normalised, like the reference), or
 *           fp32 [batch][n_tokens][dim] with IBL_VIT_OUT_ALL_TOKENS */
int64_t ibl_vit_workspace_bytes(const ibl_vit_desc* desc, int batch);
int ibl_vit_forward(const ibl_vit_desc* desc, const ibl_vit_weights* weights, const void* patches, int batch,
                    float* out, void* workspace, int64_t workspace_bytes, void* stream);

/* One linear layer of the encoder on its own: out = epilogue(x W^T + bias).  The nn.Linear of the reference's
 * encoders (transformers' ViTSelfAttention / ViTIntermediate / ViTOutput called from utils/embeddings.py:46,69,93).
 *   x [dev] fp16 [rows][ldx], W [dev] fp16 [n_out][ldw] (nn.Linear layout), bias [dev] fp32 [n_out] or NULL
 *   epilogue: IBL_LINEAR_F16 -> out fp16; IBL_LINEAR_GELU_F16 -> out = gelu(.) fp16 (erf form);
 *             IBL_LINEAR_RESID_F32 -> out fp32 += scale[n] * (.) (scale NULL = 1); IBL_LINEAR_F32 -> out fp32
 *   n_out % 128 == 0, n_in % 64 == 0, ldx / ldw / ldo in elements with 16-byte aligned rows */
enum { IBL_LINEAR_F16 = 0, IBL_LINEAR_GELU_F16 = 1, IBL_LINEAR_RESID_F32 = 2, IBL_LINEAR_F32 = 4 };
int ibl_linear_f16(const void* x, int64_t ldx, const void* W, int64_t ldw, const float* bias, const float* scale,
                    int64_t rows, int n_out, int n_in, int epilogue, void* out, int64_t ldo, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* DATOR RGB-D encoder                                                       */
/* ------------------------------------------------------------------------------------------ */

/* Fusion-head weights of build_FourDNet (dator/model/make_model.py:484-591), all [dev] fp32.  Linear weights are
 * [out][in] row-major; hyper-network convolution weights are repacked to [9 taps][in][out]. */
typedef struct {
    const float *proj_local_rgb_w, *proj_local_rgb_b, *proj_global_rgb_w, *proj_global_rgb_b, *merge_rgb_w, *merge_rgb_b;
    const float *proj_local_depth_w, *proj_local_depth_b, *proj_global_depth_w, *proj_global_depth_b, *merge_depth_w, *merge_depth_b;
    const float *Q_r_w, *Q_r_b, *V_r_w, *V_r_b, *Q_d_w, *Q_d_b, *V_d_w, *V_d_b;
    const float *r2r_sel_w, *r2r_sel_b, *r2r_aw_w, *r2r_aw_b, *r2r_ffn_w, *r2r_ffn_b, *r2r_norm_g, *r2r_norm_b;
    const float *d2d_sel_w, *d2d_sel_b, *d2d_aw_w, *d2d_aw_b, *d2d_ffn_w, *d2d_ffn_b, *d2d_norm_g, *d2d_norm_b;
    const float *d2r_sel_w, *d2r_sel_b, *d2r_aw_w, *d2r_aw_b, *d2r_ffn_w, *d2r_ffn_b, *d2r_norm_g, *d2r_norm_b;
    const float *r2d_sel_w, *r2d_sel_b, *r2d_aw_w, *r2d_aw_b, *r2d_ffn_w, *r2d_ffn_b, *r2d_norm_g, *r2d_norm_b;
    const float *hyper0_w, *hyper0_b, *hyper1_w, *hyper1_b, *hyper2_w, *hyper2_b, *hyper3_w, *hyper3_b;
} ibl_dator_head_weights;

/* build_FourDNet.forward after the two backbones (dator/model/make_model.py:676-843, eval mode):
 * rgb_tokens / depth_tokens [dev] fp32 [batch][129][768] (ibl_vit_forward with IBL_VIT_OUT_ALL_TOKENS on the TransReID
 * streams, 11 of 12 blocks, no final norm) -> out [dev] fp32 [batch][128].  Replaces get_dator_embeddings
 * (utils/embeddings.py:105-121) together with ibl_preprocess_crops / ibl_preprocess_depth. */
int64_t ibl_dator_head_workspace_bytes(int batch);
int ibl_dator_head_forward(const ibl_dator_head_weights* w, const float* rgb_tokens, const float* depth_tokens, int batch,
                           float* out, void* workspace, int64_t workspace_bytes, void* stream);

/* Depth crops (float, [dev], crop i = sizes[2i] x sizes[2i+1] floats at src + offsets[i]) -> fp16 patch matrix of the
 * depth stream: bilinear resize to out_h x out_w (cv2.INTER_LINEAR convention), clip [dmin, dmax], (d - dmin)/(dmax - dmin),
 * (x - 0.5)/0.5, three identical channels (dator/get_embeds.py:129-136; the reference's dator_wrapper is missing). */
int ibl_preprocess_depth(const float* src, const int64_t* offsets, const int32_t* sizes, int n_crops, int out_h, int out_w,
                         int patch, int patch_k_pad, float dmin, float dmax, void* patches, void* stream);

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

/* Per-row two-ended candidate selection on the fp16 rows `aug` ([sims | 1], ld >= n_cols + 1 halves per row): the k_hi largest and
 * the k_lo smallest of the first n_cols entries of every row under the total order (value, then lower column first) -- the order
 * np.argmax's tie rule induces in utils/similarity_volume.py:219-225 -- sorted, as
 *   out_val [dev] n_rows x (k_hi + k_lo) IEEE binary16 bits, out_idx [dev] n_rows x (k_hi + k_lo) int32 = index_base + column,
 *   out_cnt [dev] n_rows x 2 int32 = (entries of the high list, entries of the low list).
 * A row with n_cols <= k_hi + k_lo is returned whole in the high list (count n_cols, 0).  k_hi, k_lo <= 256.  Only these lists leave
 * the GPU (8 bytes per entry instead of the 2 (M + 1)-byte row) or cross xGMI when the memory is sharded by instance range
 * (index_base = first instance of the shard).  ibl_assign_candidates runs the assignment search on them. */
int ibl_topk_select(const uint16_t* aug, int64_t n_rows, int64_t ld, int n_cols, int k_hi, int k_lo, int index_base,
                    uint16_t* out_val, int32_t* out_idx, int32_t* out_cnt, void* stream);

/* ibl_closest_similarity + ibl_topk_select in one call (SURVEY §8b `ibl_match_topk`): query rows against this rank's memory rows,
 * the candidates of every row with GLOBAL instance indices.  out_aug [dev] n_query x (n_inst + 1) or NULL (kept in the workspace).
 * Replaces object_memory/object_memory.py:933-936 and the cast of utils/similarity_volume.py:13-18 for a memory shard. */
int64_t ibl_match_topk_workspace_bytes(int64_t n_query, int64_t n_mem_rows, int64_t n_inst);
int ibl_match_topk(const float* det, int64_t n_query, const float* mem, int64_t n_mem_rows, const int32_t* emb_offsets, int64_t n_inst,
                   int dim, int k_hi, int k_lo, int index_base, uint16_t* out_val, int32_t* out_idx, int32_t* out_cnt, uint16_t* out_aug,
                   void* workspace, int64_t workspace_bytes, void* stream);

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

/* The same search on per-row candidate lists (ibl_topk_select / ibl_match_topk, possibly the union of the lists of several memory
 * shards after the all-gather): row r has cand_cnt[r] entries (value bits, global column) at cand_val / cand_idx + r * cand_stride,
 * in any order; frame f owns the rows row_first[f] .. row_first[f] + q_per_frame[f] - 1 (q <= 7).  M_total = instances of the whole
 * memory; k_hi / k_lo = the list sizes every producer used (they define what a producer may have dropped: entries between its
 * k_lo-th smallest and k_hi-th largest).  The search runs on the union of the rows' candidate columns and PROVES per frame that no
 * dropped entry could have reached the k-th best cell of any sub-volume (the chained fp16 product is monotone in every coordinate,
 * so its maximum over the dropped range is attained at a corner: csrc/assign.cpp).  out_exact[f] = 1: the list equals
 * ibl_assign_batch on the full rows; 0: the proof failed (ties at the candidate threshold) and the caller redoes the frame on
 * full rows.  All pointers [host]. */
int ibl_assign_candidates(const uint16_t* cand_val, const int32_t* cand_idx, const int32_t* cand_cnt, int64_t cand_stride,
                          const int32_t* row_first, const int32_t* q_per_frame, int n_frames, int M_total, int k_hi, int k_lo,
                          int num_per_length, int32_t* out_assn, int32_t* out_len, int32_t* out_count, uint8_t* out_exact,
                          int max_assn, int n_threads);

/* ------------------------------------------------------------------------------------------ */
/* exchange: the collectives of the sharded path on RCCL                      */
/* ------------------------------------------------------------------------------------------ */

/* The reference issues no collective on this path (one process); with the memory sharded by instance range over the GPUs of a node
 * two are needed: the all-gather of the per-shard candidate lists before the assignment search (north star) and the all-reduce(MIN)
 * of per-point nearest distances for evaluate_transform against sharded clouds.  One communicator per process / GPU:
 * ibl_comm_unique_id on rank 0 (returns the id size, 128 bytes), the id reaches the other ranks through the host's own channel,
 * ibl_comm_init on every rank (current HIP device), then the collectives on the caller's stream.  The Python layer can use
 * torch.distributed (the same RCCL) instead: ibloc_amd.parallel. */
typedef struct ibl_comm ibl_comm;
int ibl_comm_unique_id(void* out, int out_bytes);
int ibl_comm_init(ibl_comm** out, int rank, int world, const void* unique_id, int id_bytes);
int ibl_comm_destroy(ibl_comm* comm);
/* recv [dev] world x bytes_per_rank: the contribution of rank r at offset r * bytes_per_rank */
int ibl_allgather_topk(ibl_comm* comm, const void* send, void* recv, int64_t bytes_per_rank, void* stream);
/* Equal-block all-to-all (ncclSend / ncclRecv pairs in one group): block r of `send` ([world][bytes_per_pair]) goes to rank r, block r
 * of `recv` comes from rank r.  The exchange of the per-shard candidate lists since round 3: the owner of a query row receives the
 * `world` lists of that row and nothing else (the all-gather above delivered every rank's lists to every rank). */
int ibl_alltoall(ibl_comm* comm, const void* send, void* recv, int64_t bytes_per_pair, void* stream);
/* in place, element-wise minimum over the ranks of n floats (ibl_evaluate_points distances) */
int ibl_allreduce_min(ibl_comm* comm, float* buf, int64_t n, void* stream);
/* in place, element-wise maximum of n int32 (the "some rank needs the full rows" flag of a step) */
int ibl_allreduce_max_i32(ibl_comm* comm, int32_t* buf, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* register: grids, normals + FPFH, feature matching, RANSAC, coloured ICP (rows a9-a13)        */
/* ------------------------------------------------------------------------------------------ */

/* Registration context: one device arena (bump allocator) that the batched calls carve their
 * grids, neighbour lists and per-job state from.  The only persistent device allocation of the
 * library; created and destroyed explicitly.  Not thread-safe: one context per stream/thread. */
typedef struct ibl_reg_ctx ibl_reg_ctx;
int ibl_reg_ctx_create(ibl_reg_ctx** out, int64_t arena_bytes);
int ibl_reg_ctx_destroy(ibl_reg_ctx* ctx);
/* drop every allocation of the arena, including memory grids built from it (they become invalid) */
int ibl_reg_ctx_reset(ibl_reg_ctx* ctx);
int64_t ibl_reg_ctx_high_water(const ibl_reg_ctx* ctx);
/* device status word (bit 0: grid table overflow, bit 1: a k-NN query took the re-scan slow path; bits 2-3 are internal to
 * ibl_register_batch_cached; bits 4 / 5, informational: a registration call since the last clear was redone with the VALU feature
 * search / with a full-size RANSAC survivor list after the fast path's list overflowed -- same results, more time); synchronises the
 * device */
int ibl_reg_ctx_status(ibl_reg_ctx* ctx, int clear);

/* Clouds are passed as batches of segments: pts4 [dev] N x float4 (x, y, z, intensity =
 * (r+g+b)/3), seg_off [dev] and [host] copies of the (n_seg + 1) int32 segment boundaries. */

/* Depth image + instance masks -> one coloured cloud per mask.  Replaces
 * get_mask_coloured_pointclouds_from_depth / get_coloured_pointcloud_from_depth (utils/depth_utils.py:176-206, 46-90)
 * before their outlier step (-> ibl_radius_outlier_batch): the reference's centred pixel grid
 *     x = linspace(-W/2, W/2, W)[col] * z / fx,   y = linspace(H/2, -H/2, H)[row] * z / fy,   z = depth / depth_factor
 * (float32 linspace values, utils/depth_utils.py:57-67, whose `w, h` names are swapped), pixels with z == 0 or outside the
 * mask dropped, row-major pixel order kept, intensity = mean of the float32 colours / 255.  The arithmetic type follows
 * numpy's promotion of the reference's expressions: a float32 depth image stays in float32 throughout, an integer or
 * float64 one promotes the products to float64 (then rounded to the float32 of the HBM layout).
 *   depth [dev] H x W; depth_type IBL_DEPTH_F32 / IBL_DEPTH_U16 / IBL_DEPTH_F64; rgb [dev] H x W x 3 u8; masks [dev] n_masks x H x W u8
 *   pts4 [dev] capacity x float4 (n_masks * H * W always suffices), seg_off_dev [dev] / seg_off_host [HOST] n_masks + 1
 * The call synchronises (the cloud sizes are returned to the host). */
enum { IBL_DEPTH_F32 = 0, IBL_DEPTH_U16 = 1, IBL_DEPTH_F64 = 2 };
int ibl_unproject_masks(ibl_reg_ctx* ctx, const void* depth, int depth_type, const uint8_t* rgb, const uint8_t* masks, int n_masks,
                        int H, int W, double fx, double fy, double depth_factor, float* pts4, int64_t capacity,
                        int32_t* seg_off_dev, int32_t* seg_off_host, void* stream);
/* Same, and additionally the clouds as the reference's Open3D containers hold them for the memory build (process_image,
 * object_memory/object_memory.py:163-256): pts3_f64 / colors3_f64 [dev] capacity x 3 doubles or NULL = the numpy values (float32
 * or float64 by the promotion rule above; colours = float32 rgb / 255) widened to double, same order as pts4. */
int ibl_unproject_masks_f64(ibl_reg_ctx* ctx, const void* depth, int depth_type, const uint8_t* rgb, const uint8_t* masks, int n_masks,
                            int H, int W, double fx, double fy, double depth_factor, float* pts4, double* pts3_f64, double* colors3_f64,
                            int64_t capacity, int32_t* seg_off_dev, int32_t* seg_off_host, void* stream);

/* keep[i] = 1 iff the point has more than nb_points points (itself included) within `radius` of its
 * own cloud.  Replaces PointCloud.remove_radius_outlier (object_memory/object_memory.py:994-995,
 * utils/depth_utils.py:87-88).  keep: [dev] N bytes. */
int ibl_radius_outlier_batch(ibl_reg_ctx* ctx, const float* pts4, const int32_t* seg_off_dev, const int32_t* seg_off_host,
                             int n_seg, double radius, int nb_points, uint8_t* keep, void* stream);

/* Normals (hybrid radius_normal / max_nn_normal, Open3D fast 3x3 eigen solver, no orientation) and
 * FPFH (hybrid radius_feature / max_nn_feature) of every cloud of the batch.  Replaces
 * downsample_and_compute_fpfh (utils/fpfh_register.py:86-98).  normals4: [dev] N x float4,
 * fpfh: [dev] N x 33 fp32 (point-major) or NULL to skip the features.  When radius_normal <= radius_feature and max_nn_normal <=
 * min(max_nn_feature, 32) -- the reference's 2 / 5 voxel, 30 / 100 neighbours -- ONE neighbour search serves both (the normal's
 * neighbours are among the feature neighbours); the results are those of the two searches bit for bit (IBL_FEAT_UNFUSED=1 runs them). */
int ibl_normals_fpfh_batch(ibl_reg_ctx* ctx, const float* pts4, const int32_t* seg_off_dev, const int32_t* seg_off_host,
                           int n_seg, double radius_normal, int max_nn_normal, double radius_feature, int max_nn_feature,
                           float* normals4, float* fpfh, void* stream);

/* Batched register_point_clouds (utils/fpfh_register.py:100-143) over n_jobs (frame, assignment) jobs.
 * Job j registers the concatenation of up to three segments of the detected pool (job_src_seg[3j..],
 * -1 padded) onto the concatenation of up to three segments of the memory pool (job_tgt_seg), exactly
 * like object_memory/object_memory.py:1023-1034,1087-1089:
 *   IBL_REG_CENTER      subtract each side's mean first (localise does; the stand-alone function does not)
 *   IBL_REG_HAVE_COLORS normals + FPFH + RANSAC + coloured ICP; without it the reference's exception path:
 *                       point-to-point ICP from the identity (fpfh_register.py:137-141)
 * Normals, FPFH and the targets' colour gradients are evaluated on the concatenations BEFORE centring (they are
 * translation invariant; see ibl_instance_features); RANSAC and ICP run between the centred clouds.
 * RANSAC hypothesis i of job j is drawn from Philox4x32-10(counter = (i, job_id_base + j, 0, 0), key = seed).
 * Outputs are HOST arrays (the call synchronises): T_out [n_jobs][16] row-major double (between the centred
 * clouds), rmse_out / fitness_out [n_jobs] (result_icp.inlier_rmse / .fitness), means_out [n_jobs][2][3]
 * (detected mean, memory mean; zeros without IBL_REG_CENTER) or NULL, T_ransac_out [n_jobs][16] or NULL,
 * ransac_stats_out [n_jobs][3] = (hypotheses walked, validated, best inlier count) or NULL. */
#define IBL_REG_HAVE_COLORS 1
#define IBL_REG_CENTER 2
#define IBL_REG_FIXED_BUDGET 4   /* RANSAC walks exactly ransac_max_iter hypotheses per job (confidence exit off): the fixed-budget
                                    hypotheses/s figure of the benchmark, never the product setting */
int ibl_register_batch(ibl_reg_ctx* ctx, const float* det_pts4, const int32_t* det_off_dev, const int32_t* det_off_host,
                       int n_det_seg, const float* mem_pts4, const int32_t* mem_off_dev, const int32_t* mem_off_host,
                       int n_mem_seg, const int32_t* job_src_seg, const int32_t* job_tgt_seg, int n_jobs, double voxel_size,
                       double global_dist_factor, double local_dist_factor, uint64_t seed, uint32_t job_id_base,
                       int64_t ransac_max_iter, int flags, double* T_out, double* rmse_out, double* fitness_out,
                       double* means_out, double* T_ransac_out, int64_t* ransac_stats_out, void* stream);

/* Per-instance registration features, computed ONCE per cloud and kept resident in HBM instead of once per
 * (frame, assignment) job: the reference re-runs estimate_normals / compute_fpfh_feature
 * (utils/fpfh_register.py:86-98) and the colour gradients inside registration_colored_icp
 * (utils/fpfh_register.py:125-133) on the concatenated clouds of every assignment (object_memory.py:1023-1034,
 * 1087-1089), although a memory instance never changes and a detected instance is shared by all assignments of its
 * frame.  Normals, FPFH and colour gradients only depend on point differences, so they are evaluated in the frame
 * the clouds are stored in (before the per-job centring) and the value at a point of instance A inside a
 * concatenation A+B+C equals its stand-alone value unless a point of B or C lies within the influence radius
 *     R = max(2 * 5 * voxel + 2 * voxel, grad_radius + 2 * voxel)
 * of A.  ibl_register_batch_cached uses the stored values for every instance none of whose points is within R (plus a
 * rounding margin) of a point of another instance of its job side -- bounding boxes first, then an exact point-set test
 * on the device for the box pairs that are close -- and recomputes the rest in the context of the job's concatenation,
 * which makes its results bit-identical to ibl_register_batch.
 *   normals4 [dev] N x float4, fpfh [dev] N x 33 with every row in MATCHING ORDER (bin 11 b + c at position
 *   3 * rank(c) + {b=1: 0, b=2: 1, b=0: 2}, rank over c = 5,4,6,3,7,2,8,1,9,0,10: the three histograms from their centre
 *   bins outwards, interleaved -- the order in which the feature search sums its squared differences, so that its
 *   early-abandon chain reads contiguously), fpfh_split / fpfh_norm [dev]: every row once more, CENTRED (x = row - a constant table, FM_MU of
 *   csrc/reg_common.h), as 48 fp16 search operands
 *   [x_0 .. x_32 | 8 8 | |x|^2 / 8 as hi + lo | 1e-3 |x|^2 + 4e-3 rounded up | 0 ..] and its squared norm -- the operands of the
 *   matrix-core filter of the feature search, whose one MFMA chain yields the whole distance bound (csrc/reg_featnn.hip), grad4 [dev] N x float4 or NULL (grad_radius <= 0: targets'
 *   gradients are recomputed per job), bbox [HOST] n_seg x 6 = (min xyz, max xyz) -- all written by
 *   ibl_instance_features_batch (the call synchronises); voxel_size / grad_radius: the parameters they hold for
 *   (grad_radius = 2 * voxel_size * local_dist_factor in register_point_clouds). */
typedef struct {
    const float* normals4;
    const float* fpfh;
    const uint16_t* fpfh_split;   /* [dev] N x 48 fp16 search operands (layout above), or NULL: COMPACT features (168 instead of 264
                                   * bytes per point) -- the search builds the same operands from fpfh / fpfh_norm as it stages them */
    const float* fpfh_norm;       /* [dev] N: |centred row|^2 */
    const float* grad4;
    const float* bbox;
    double voxel_size;
    double grad_radius;
} ibl_instance_features;
int ibl_instance_features_batch(ibl_reg_ctx* ctx, const float* pts4, const int32_t* seg_off_dev, const int32_t* seg_off_host,
                                int n_seg, double voxel_size, double grad_radius, float* normals4, float* fpfh, uint16_t* fpfh_split,
                                float* fpfh_norm, float* grad4, float* bbox_host, void* stream);
/* ibl_register_batch with the instance features of the detected pool and / or the memory pool (either may be NULL).
 * reuse_stats_out [HOST][6] or NULL: points served by the instance features, points recomputed, recomputed groups,
 * job sides, distinct (query instance, database instance) feature-matching pairs searched, pair uses by the jobs. */
int ibl_register_batch_cached(ibl_reg_ctx* ctx, const float* det_pts4, const int32_t* det_off_dev, const int32_t* det_off_host,
                              int n_det_seg, const float* mem_pts4, const int32_t* mem_off_dev, const int32_t* mem_off_host,
                              int n_mem_seg, const int32_t* job_src_seg, const int32_t* job_tgt_seg, int n_jobs, double voxel_size,
                              double global_dist_factor, double local_dist_factor, uint64_t seed, uint32_t job_id_base,
                              int64_t ransac_max_iter, int flags, const ibl_instance_features* det_features,
                              const ibl_instance_features* mem_features, double* T_out, double* rmse_out, double* fitness_out,
                              double* means_out, double* T_ransac_out, int64_t* ransac_stats_out, int64_t* reuse_stats_out,
                              void* stream);
/* ibl_register_batch_cached with an explicit RANSAC job id per job (job_ids [HOST][n_jobs]) in place of job_id_base + j: the result of
 * a job is a function of its clouds, the seed and its id only, so a job routed to another rank (memory clouds sharded by instance
 * range: the owner of its target instances runs it, SURVEY 8e / routing.py) returns the bits it would have returned at home.  The
 * loop being distributed: object_memory/object_memory.py:1020-1106.  Synchronisation: as ibl_register_batch_cached. */
int ibl_register_batch_ids(ibl_reg_ctx* ctx, const float* det_pts4, const int32_t* det_off_dev, const int32_t* det_off_host,
                           int n_det_seg, const float* mem_pts4, const int32_t* mem_off_dev, const int32_t* mem_off_host,
                           int n_mem_seg, const int32_t* job_src_seg, const int32_t* job_tgt_seg, const uint32_t* job_ids, int n_jobs,
                           double voxel_size, double global_dist_factor, double local_dist_factor, uint64_t seed,
                           int64_t ransac_max_iter, int flags, const ibl_instance_features* det_features,
                           const ibl_instance_features* mem_features, double* T_out, double* rmse_out, double* fitness_out,
                           double* means_out, double* T_ransac_out, int64_t* ransac_stats_out, int64_t* reuse_stats_out, void* stream);

/* ---- memory build / consolidation; fp64 clouds, as the build side of the reference keeps them ---- */

/* Voxel down-sampling with colours of every object of a batch.  Replaces voxel_down_sample_with_colors
 * (utils/depth_utils.py:211-265) as applied per object by ObjectInfo.downsample (object_memory/object_info.py:95-97) <-
 * ObjectMemory.downsample_all_objects (object_memory/object_memory.py:258-263): voxel = floor(p / voxel_size) per axis, output
 * voxels in order of first occurrence, point / colour = running fp64 sum in input order / count (bit-identical to np.mean).
 *   points, colors [dev] N x 3 doubles (colors may be NULL); seg_off_host [HOST] n_seg + 1 object offsets (n_seg <= 65535, an
 *   object may span at most 65536 voxels per axis); out_points / out_colors [dev] N x 3 doubles (the first
 *   out_seg_off_host[n_seg] rows are written); out_counts [dev] N ints or NULL (points per voxel); out_seg_off_host [HOST]
 *   n_seg + 1.  The call synchronises. */
int ibl_voxel_downsample_batch(ibl_reg_ctx* ctx, const double* points, const double* colors, const int32_t* seg_off_host, int32_t n_seg,
                               double voxel_size, double* out_points, double* out_colors, int32_t* out_counts,
                               int32_t* out_seg_off_host, void* stream);

/* DBSCAN labels, one independent clustering per group of a batch of concatenated clouds.  Replaces
 * open3d.geometry.PointCloud.cluster_dbscan(eps, min_points) as called by recluster_objects_with_dbscan
 * (object_memory/object_memory.py:300-305) and per agglomerative cluster by recluster_via_clustering_and_IoU (:627-631):
 * neighbours are the points at squared distance < eps^2 (the point itself included), a core point has >= min_points of them,
 * clusters are numbered from 0 per group in the order a sequential scan meets their first core point, a border point takes the
 * lowest-numbered cluster among its core neighbours, every other point is -1.
 *   points [dev] N x 3 doubles; grp_off_host [HOST] n_grp + 1; labels [dev] N ints; n_clusters_host [HOST] n_grp or NULL. */
int ibl_dbscan_batch(ibl_reg_ctx* ctx, const double* points, const int32_t* grp_off_host, int32_t n_grp, double eps, int32_t min_points,
                     int32_t* labels, int32_t* n_clusters_host, void* stream);

/* Persistent spatial hash over ALL memory points (world frame), built once per memory upload from the
 * context arena.  Replaces the KD-tree Open3D rebuilds over `all_memory_pcd` on every evaluate_registration
 * call (utils/fpfh_register.py:146-148 <- object_memory/object_memory.py:1104).  cell >= 2 * threshold keeps a
 * query to <= 8 cells. */
typedef struct ibl_memgrid ibl_memgrid;
int ibl_memgrid_build(ibl_reg_ctx* ctx, const float* mem_pts4, int64_t n, double cell, ibl_memgrid** out, void* stream);
int ibl_memgrid_destroy(ibl_memgrid* grid);

/* evaluate_transform(all_detected_pcd, all_memory_pcd, T) for n_jobs candidates: job j transforms the detected
 * points [job_begin[j], job_end[j]) of det_pts4 [dev] by T_global[j] (host, 16 doubles row-major) and looks for
 * the nearest memory point within `threshold`.  rmse_out / fitness_out: HOST arrays (the call synchronises). */
int ibl_evaluate_batch(ibl_reg_ctx* ctx, const ibl_memgrid* grid, const float* det_pts4, const int32_t* job_begin,
                       const int32_t* job_end, const double* T_global, int n_jobs, double threshold, double* rmse_out,
                       double* fitness_out, void* stream);

/* Stage B of localise() for a batch of frames in ONE call (SURVEY 8b's fused driver; csrc/localise.hip): radius-outlier removal of
 * every detected cloud and ordered compaction (object_memory/object_memory.py:992-998), the detections' instance features, one
 * registration job per candidate assignment (:1020-1095), the global-frame transform of every job (:1096-1101), its whole-memory
 * evaluation (:1104) and the winner of every frame (highest whole-memory fitness, the first on ties, :1111-1114).  Everything is
 * the library's own stage entry points composed on the host -- the results are those of calling ibl_radius_outlier_batch,
 * ibl_instance_features_batch, ibl_register_batch_cached and ibl_evaluate_batch in turn, bit for bit.
 *   det_pts4 / det_off_dev / det_off_host: the RAW detected clouds, segments in frame order, q_per_frame [HOST][n_frames] of them per
 *   frame (<= 7 each); assn [HOST][n_frames][max_assn][6] = up to three (detection within its frame, GLOBAL memory instance) pairs
 *   per assignment, assn_len [HOST][n_frames][max_assn] pairs used, assn_count [HOST][n_frames] -- the output layout of
 *   ibl_assign_batch / ibl_assign_candidates; mem_*: the memory pool, its resident instance features and its spatial hash.
 * Outputs (HOST; the call synchronises): clean_off_host [n_det_seg + 1] offsets of the cleaned clouds, *n_jobs_out = J (jobs in
 * frame order, a frame's jobs in assignment order; J <= max_jobs = the capacity of the per-job arrays), T_out .. reuse_stats_out as
 * ibl_register_batch_cached, T_global_out [J][16], full_rmse_out / full_fitness_out [J], best_out [n_frames] = winning assignment of
 * the frame (index into its list) or -1 for a frame without assignments. */
int ibl_register_evaluate_batch(ibl_reg_ctx* ctx, const float* det_pts4, const int32_t* det_off_dev, const int32_t* det_off_host,
                                int n_det_seg, const int32_t* q_per_frame, int n_frames, const int32_t* assn, const int32_t* assn_len,
                                const int32_t* assn_count, int max_assn, const float* mem_pts4, const int32_t* mem_off_dev,
                                const int32_t* mem_off_host, int n_mem_seg, const ibl_instance_features* mem_features,
                                const ibl_memgrid* grid, double voxel_size, double global_dist_factor, double local_dist_factor,
                                double outlier_radius, int outlier_nb_points, double eval_threshold, uint64_t seed, uint32_t job_id_base,
                                int64_t ransac_max_iter, int flags, int max_jobs, int32_t* clean_off_host, int32_t* n_jobs_out,
                                double* T_out, double* rmse_out, double* fitness_out, double* means_out, double* T_ransac_out,
                                int64_t* ransac_stats_out, int64_t* reuse_stats_out, double* T_global_out, double* full_rmse_out,
                                double* full_fitness_out, int32_t* best_out, void* stream);

/* Same, and additionally the squared distance of every transformed detected point to its nearest memory point within `threshold`
 * (+inf when there is none): d2_out [dev] floats, job j's points at offset sum_{i<j} (job_end[i] - job_begin[i]).  This is the
 * per-shard half of the sharded whole-memory evaluation of SURVEY 8(e): every rank evaluates against the memory points it owns, the
 * distances are combined with an all-reduce(MIN) (RCCL) and fitness / rmse follow from the combined array
 * (ibloc_amd.parallel.evaluate_sharded). */
int ibl_evaluate_points(ibl_reg_ctx* ctx, const ibl_memgrid* grid, const float* det_pts4, const int32_t* job_begin,
                        const int32_t* job_end, const double* T_global, int n_jobs, double threshold, float* d2_out, double* rmse_out,
                        double* fitness_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IBLOC_H */
