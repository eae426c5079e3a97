/* ORACLE (test infrastructure; never linked or called by the product path).
 *
 * CPU restatement of the reference's embedding normalisation and closest-similarity matrix:
 *   /root/reference/object_memory/object_memory.py:922-925  e / ||e||, detected /= ||detected||
 *   /root/reference/object_memory/object_memory.py:933-936  closest[i][j] = max_e dot(mem_j[e], det_i)
 *   /root/reference/utils/similarity_volume.py:13-18        aug = [sims | 1] as float16
 * The reference leaves the fp32 summation order to numpy/BLAS.  This restatement fixes it to the
 * order documented in include/ibloc.h (the one the gfx950 kernels use), so the HIP path can be
 * checked bit-exactly; tests/test_oracle_match.py pins it against plain numpy (the reference's
 * arithmetic) to 1e-6.
 */
#include <math.h>
#include <stdint.h>

/* 64 strided fmaf partial sums, xor-butterfly 32..1, IEEE sqrt and divide */
void oracle_normalize_rows(const float* in, float* out, int64_t n_rows, int dim) {
    for (int64_t r = 0; r < n_rows; ++r) {
        const float* x = in + r * dim;
        float s[64];
        for (int l = 0; l < 64; ++l) {
            float acc = 0.0f;
            for (int i = l; i < dim; i += 64) acc = fmaf(x[i], x[i], acc);
            s[l] = acc;
        }
        for (int off = 32; off >= 1; off >>= 1) {
            float t[64];
            for (int l = 0; l < 64; ++l) t[l] = s[l] + s[l ^ off];
            for (int l = 0; l < 64; ++l) s[l] = t[l];
        }
        const float nrm = sqrtf(s[0]);
        float* y = out + r * dim;
        for (int i = 0; i < dim; ++i) y[i] = x[i] / nrm;
    }
}

/* dot product as an fmaf chain; inside every block of 8 the order is 0,4,1,5,2,6,3,7 */
static float dot_chain(const float* a, const float* b, int dim) {
    static const int ord[8] = {0, 4, 1, 5, 2, 6, 3, 7};
    float acc = 0.0f;
    for (int m = 0; m < dim; m += 8)
        for (int c = 0; c < 8; ++c) acc = fmaf(a[m + ord[c]], b[m + ord[c]], acc);
    return acc;
}

void oracle_closest_similarity(const float* det, int64_t n_query, const float* mem, const int32_t* emb_offsets,
                               int64_t n_inst, int dim, float* out_sims) {
    for (int64_t q = 0; q < n_query; ++q)
        for (int64_t j = 0; j < n_inst; ++j) {
            float best = -INFINITY;
            for (int e = emb_offsets[j]; e < emb_offsets[j + 1]; ++e) {
                float v = dot_chain(mem + (int64_t)e * dim, det + q * dim, dim);
                if (v > best) best = v;
            }
            out_sims[q * n_inst + j] = best;
        }
}
