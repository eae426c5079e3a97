// dator.hip -- DATOR fusion head (build_FourDNet.forward, /root/reference/dator/model/make_model.py:629-843) and the
// depth-crop preprocessing of its second stream (dator/get_embeds.py:129-136) on gfx950.
//
// The two TransReID streams (11 ViT-B/16 blocks at 256x128) run on the shared ViT kernels (vit.hip) and hand over
// fp32 token tensors [B][129][768].  The head is ~0.2 GFLOP per crop against 42.5 GFLOP of backbone, all of it on
// 128-token x 128-channel maps; it is kept in fp32 (the reference's precision) as a short sequence of small kernels:
// a generic LDS-tiled fp32 linear, a token-major 3x3 convolution for the hyper-network, the deformable bilinear
// sampler (F.grid_sample semantics: align_corners=True, zero padding), residual + gate + LayerNorm, gated mean.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "ibl_common.h"
#include "ibloc.h"

#define DT 128          // tokens per crop (16 x 8)
#define DC 128          // reduced channel dimension

// ------------------------------------------------------------------------------------------------
// out[r][n] = act(b[n] + sum_k x[src(r)][k] * W[n][k]);   src(r) = (r / rpg) * gstride + roff + (r % rpg)
// 64 x 64 output tile, 16-deep K chunks through LDS, 4 x 4 outputs per thread
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ibl_linear_f32_kernel(const float* __restrict__ x, int64_t ldx, int rpg, int gstride, int roff,
                                                             const float* __restrict__ W, int64_t ldw, const float* __restrict__ bias,
                                                             float* __restrict__ out, int64_t ldo, int R, int K, int N, int act) {
    __shared__ float sx[16][65], sw[16][65];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int r0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    float acc[4][4] = {{0}};
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int e = threadIdx.x; e < 64 * 16; e += 256) {
            const int row = e >> 4, kk = e & 15;
            const int r = r0 + row, n = n0 + row, k = k0 + kk;
            float vx = 0.f, vw = 0.f;
            if (r < R && k < K) {
                const int64_t src = (int64_t)(r / rpg) * gstride + roff + (r % rpg);
                vx = x[src * ldx + k];
            }
            if (n < N && k < K) vw = W[(int64_t)n * ldw + k];
            sx[kk][row] = vx;
            sw[kk][row] = vw;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = sx[kk][ty * 4 + i]; b[i] = sw[kk][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty * 4 + i;
        if (r >= R) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= N) continue;
            float v = acc[i][j] + (bias ? bias[n] : 0.f);
            if (act == 1) v = fmaxf(v, 0.f);
            else if (act == 2) v = 1.0f / (1.0f + expf(-v));
            out[(int64_t)r * ldo + n] = v;
        }
    }
}

static int launch_linear(const float* x, int64_t ldx, int rpg, int gstride, int roff, const float* W, int64_t ldw, const float* bias,
                         float* out, int64_t ldo, int R, int K, int N, int act, hipStream_t s) {
    if (R <= 0) return IBL_OK;
    dim3 grid((N + 63) / 64, (R + 63) / 64);
    hipLaunchKernelGGL(ibl_linear_f32_kernel, grid, dim3(256), 0, s, x, ldx, rpg, gstride, roff, W, ldw, bias, out, ldo, R, K, N, act);
    IBL_LAUNCH_CHECK();
    return IBL_OK;
}

// cat[r] = [global[b] | local[r]]   (torch.cat((global.unsqueeze(1).repeat(1, N, 1), local), -1), make_model.py:692,708)
__global__ void ibl_dator_cat_kernel(const float* __restrict__ g, const float* __restrict__ l, float* __restrict__ cat, int R) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)R * 2 * DC) return;
    const int r = (int)(i / (2 * DC)), c = (int)(i % (2 * DC));
    cat[i] = c < DC ? g[(int64_t)(r / DT) * DC + c] : l[(int64_t)r * DC + (c - DC)];
}

// hyper-network input: channels [depth 128 | rgb 128] per token (make_model.py:715-717)
__global__ void ibl_dator_hyperin_kernel(const float* __restrict__ fd, const float* __restrict__ fr, float* __restrict__ o, int R) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)R * 2 * DC) return;
    const int r = (int)(i / (2 * DC)), c = (int)(i % (2 * DC));
    o[i] = c < DC ? fd[(int64_t)r * DC + c] : fr[(int64_t)r * DC + (c - DC)];
}

// 3x3 convolution, padding 1, on the 16 x 8 token grid; token-major activations [B*128][Cin], weights [9][Cin][Cout].
// One block per token: the 9 x Cin input window is staged in LDS, threads own output channels.
__global__ __launch_bounds__(128) void ibl_conv3x3_tok_kernel(const float* __restrict__ in, int cin, const float* __restrict__ W,
                                                              const float* __restrict__ bias, float* __restrict__ out, int cout, int relu) {
    extern __shared__ float win[];            // [9][cin]
    const int r = blockIdx.x, b = r / DT, n = r % DT;
    const int y = n / 8, x = n % 8;
    for (int e = threadIdx.x; e < 9 * cin; e += 128) {
        const int tap = e / cin, ci = e - tap * cin;
        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        win[e] = (yy >= 0 && yy < 16 && xx >= 0 && xx < 8) ? in[((int64_t)b * DT + yy * 8 + xx) * cin + ci] : 0.f;
    }
    __syncthreads();
    for (int co = threadIdx.x; co < cout; co += 128) {
        float acc = bias[co];
        for (int e = 0; e < 9 * cin; ++e) acc = fmaf(win[e], W[(int64_t)e * cout + co], acc);
        out[(int64_t)r * cout + co] = relu ? fmaxf(acc, 0.f) : acc;
    }
}

// softmax over the 2 hyper-network channels -> (rgb_filter, depth_filter) per token (make_model.py:719-723)
__global__ void ibl_dator_gate_kernel(const float* __restrict__ h, float* __restrict__ rgb_f, float* __restrict__ depth_f, int R) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float a = h[2 * r], b = h[2 * r + 1];
    const float m = fmaxf(a, b);
    const float ea = expf(a - m), eb = expf(b - m);
    rgb_f[r] = ea / (ea + eb);
    depth_f[r] = eb / (ea + eb);
}

// deformable sampling + weighted sum: one block per token, threads = channels.
// sel [R][48] already passed the sigmoid; awl [R][24] are the attention logits (softmax here); v token-major [B*128][128].
__global__ __launch_bounds__(128) void ibl_dator_sample_kernel(const float* __restrict__ sel, const float* __restrict__ awl,
                                                               const float* __restrict__ v, float* __restrict__ out) {
    const int r = blockIdx.x, b = r / DT, c = threadIdx.x;
    __shared__ float s_aw[24], s_sel[48];
    if (c < 48) s_sel[c] = sel[(int64_t)r * 48 + c];
    if (c < 24) s_aw[c] = awl[(int64_t)r * 24 + c];
    __syncthreads();
    float m = -INFINITY;
    for (int t = 0; t < 24; ++t) m = fmaxf(m, s_aw[t]);
    float den = 0.f;
    for (int t = 0; t < 24; ++t) den += expf(s_aw[t] - m);
    const float* vb = v + (int64_t)b * DT * DC;
    float acc = 0.f;
    for (int t = 0; t < 24; ++t) {
        // F.grid_sample(align_corners=True): pixel = ((g + 1) / 2) * (size - 1), g = sel * 2 - 1; x -> width 8, y -> height 16
        const float gx = s_sel[t] * 2.f - 1.f, gy = s_sel[24 + t] * 2.f - 1.f;
        const float ix = ((gx + 1.f) / 2.f) * 7.f, iy = ((gy + 1.f) / 2.f) * 15.f;
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
        const float wx1 = ix - fx0, wx0 = 1.f - wx1, wy1 = iy - fy0, wy0 = 1.f - wy1;
        float val = 0.f;
        if (y0 >= 0 && y0 < 16) {
            if (x0 >= 0 && x0 < 8) val += vb[(y0 * 8 + x0) * DC + c] * (wy0 * wx0);
            if (x1 >= 0 && x1 < 8) val += vb[(y0 * 8 + x1) * DC + c] * (wy0 * wx1);
        }
        if (y1 >= 0 && y1 < 16) {
            if (x0 >= 0 && x0 < 8) val += vb[(y1 * 8 + x0) * DC + c] * (wy1 * wx0);
            if (x1 >= 0 && x1 < 8) val += vb[(y1 * 8 + x1) * DC + c] * (wy1 * wx1);
        }
        acc += val * (expf(s_aw[t] - m) / den);
    }
    out[(int64_t)r * DC + c] = acc;
}

// f[r] = LayerNorm(f[r] + feat[r] * gate[r]) over 128 channels (eps 1e-5); gate may be null.  One wave per token.
__global__ __launch_bounds__(256) void ibl_dator_add_ln_kernel(float* __restrict__ f, const float* __restrict__ feat,
                                                               const float* __restrict__ gate, const float* __restrict__ g,
                                                               const float* __restrict__ b, int R) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float gt = gate ? gate[r] : 1.f;
    float v0 = f[(int64_t)r * DC + lane] + feat[(int64_t)r * DC + lane] * gt;
    float v1 = f[(int64_t)r * DC + 64 + lane] + feat[(int64_t)r * DC + 64 + lane] * gt;
    float s = v0 + v1;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s / (float)DC;
    const float d0 = v0 - mean, d1 = v1 - mean;
    float q = d0 * d0 + d1 * d1;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off, 64);
    const float rstd = 1.0f / sqrtf(q / (float)DC + 1e-5f);
    f[(int64_t)r * DC + lane] = d0 * rstd * g[lane] + b[lane];
    f[(int64_t)r * DC + 64 + lane] = d1 * rstd * g[64 + lane] + b[64 + lane];
}

// out[b][c] = mean_n (fd[b][n][c] * depth_f[b][n] + fr[b][n][c] * rgb_f[b][n])   (make_model.py:826-834)
__global__ __launch_bounds__(128) void ibl_dator_pool_kernel(const float* __restrict__ fd, const float* __restrict__ fr,
                                                             const float* __restrict__ depth_f, const float* __restrict__ rgb_f,
                                                             float* __restrict__ out) {
    const int b = blockIdx.x, c = threadIdx.x;
    float acc = 0.f;
    for (int n = 0; n < DT; ++n) {
        const int64_t r = (int64_t)b * DT + n;
        acc += fd[r * DC + c] * depth_f[r] + fr[r * DC + c] * rgb_f[r];
    }
    out[(int64_t)b * DC + c] = acc / (float)DT;
}

extern "C" int64_t ibl_dator_head_workspace_bytes(int batch) {
    if (batch <= 0) return -1;
    const int64_t R = (int64_t)batch * DT;
    // floats: fr fd (2 x 128) + lp (128) + cat/hyper-in (256) + h1 (128) h2 (32) h3 (8) h4 (2) + gates (2) + q/v x4 (512)
    //         + sel (48) + aw (24) + samp (128) + feat (128) + global (batch x 128 x 2)
    const int64_t per_row = 2 * 128 + 128 + 256 + 128 + 32 + 8 + 2 + 2 + 512 + 48 + 24 + 128 + 128;
    return (R * per_row + (int64_t)batch * 256) * 4 + 32 * 256;
}

extern "C" int ibl_dator_head_forward(const ibl_dator_head_weights* w, const float* rgb_tokens, const float* depth_tokens, int batch,
                                      float* out, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!w || !rgb_tokens || !depth_tokens || !out || !workspace || batch <= 0)
        return ibl_set_error(IBL_ERR_ARG, "ibl_dator_head_forward: bad argument");
    if (workspace_bytes < ibl_dator_head_workspace_bytes(batch)) return ibl_set_error(IBL_ERR_ARG, "ibl_dator_head_forward: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int B = batch, R = batch * DT, T = DT + 1, D = 768;
    unsigned char* p = reinterpret_cast<unsigned char*>(workspace);
    auto carve = [&](int64_t floats) {
        p = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(p) + 255) & ~(uintptr_t)255);
        float* r = reinterpret_cast<float*>(p);
        p += floats * 4;
        return r;
    };
    float* fr = carve((int64_t)R * DC); float* fd = carve((int64_t)R * DC);
    float* gl = carve((int64_t)B * DC); float* lp = carve((int64_t)R * DC); float* cat = carve((int64_t)R * 2 * DC);
    float* h1 = carve((int64_t)R * 128); float* h2 = carve((int64_t)R * 32); float* h3 = carve((int64_t)R * 8); float* h4 = carve((int64_t)R * 2);
    float* rgb_f = carve(R); float* depth_f = carve(R);
    float* q_r = carve((int64_t)R * DC); float* v_r = carve((int64_t)R * DC); float* q_d = carve((int64_t)R * DC); float* v_d = carve((int64_t)R * DC);
    float* sel = carve((int64_t)R * 48); float* aw = carve((int64_t)R * 24); float* samp = carve((int64_t)R * DC); float* feat = carve((int64_t)R * DC);
    int st;
    const int nb_cat = (int)(((int64_t)R * 2 * DC + 255) / 256);
    // ---- global / local projections and merge, per stream (make_model.py:680-712) ----------------------------
    struct Side { const float* tok; const float *gw, *gb, *lw, *lb, *mw, *mb; float* f; };
    Side sides[2] = {{rgb_tokens, w->proj_global_rgb_w, w->proj_global_rgb_b, w->proj_local_rgb_w, w->proj_local_rgb_b, w->merge_rgb_w, w->merge_rgb_b, fr},
                     {depth_tokens, w->proj_global_depth_w, w->proj_global_depth_b, w->proj_local_depth_w, w->proj_local_depth_b, w->merge_depth_w, w->merge_depth_b, fd}};
    for (int sd = 0; sd < 2; ++sd) {
        const Side& S = sides[sd];
        st = launch_linear(S.tok, D, 1, T, 0, S.gw, D, S.gb, gl, DC, B, D, DC, 0, s);                 // CLS rows
        if (st) return st;
        st = launch_linear(S.tok, D, DT, T, 1, S.lw, D, S.lb, lp, DC, R, D, DC, 0, s);                // patch rows
        if (st) return st;
        hipLaunchKernelGGL(ibl_dator_cat_kernel, dim3(nb_cat), dim3(256), 0, s, gl, lp, cat, R);
        IBL_LAUNCH_CHECK();
        st = launch_linear(cat, 2 * DC, R, 0, 0, S.mw, 2 * DC, S.mb, S.f, DC, R, 2 * DC, DC, 0, s);
        if (st) return st;
    }
    // ---- hyper-network gates (:714-727) ---------------------------------------------------------------------------
    hipLaunchKernelGGL(ibl_dator_hyperin_kernel, dim3(nb_cat), dim3(256), 0, s, fd, fr, cat, R);
    IBL_LAUNCH_CHECK();
    hipLaunchKernelGGL(ibl_conv3x3_tok_kernel, dim3(R), dim3(128), 9 * 256 * 4, s, cat, 256, w->hyper0_w, w->hyper0_b, h1, 128, 1);
    IBL_LAUNCH_CHECK();
    hipLaunchKernelGGL(ibl_conv3x3_tok_kernel, dim3(R), dim3(128), 9 * 128 * 4, s, h1, 128, w->hyper1_w, w->hyper1_b, h2, 32, 1);
    IBL_LAUNCH_CHECK();
    hipLaunchKernelGGL(ibl_conv3x3_tok_kernel, dim3(R), dim3(128), 9 * 32 * 4, s, h2, 32, w->hyper2_w, w->hyper2_b, h3, 8, 1);
    IBL_LAUNCH_CHECK();
    hipLaunchKernelGGL(ibl_conv3x3_tok_kernel, dim3(R), dim3(128), 9 * 8 * 4, s, h3, 8, w->hyper3_w, w->hyper3_b, h4, 2, 0);
    IBL_LAUNCH_CHECK();
    hipLaunchKernelGGL(ibl_dator_gate_kernel, dim3((R + 255) / 256), dim3(256), 0, s, h4, rgb_f, depth_f, R);
    IBL_LAUNCH_CHECK();
    // ---- queries / values from the un-updated features (:730-733) -----------------------------------------------------
    st = launch_linear(fr, DC, R, 0, 0, w->Q_r_w, DC, w->Q_r_b, q_r, DC, R, DC, DC, 0, s); if (st) return st;
    st = launch_linear(fr, DC, R, 0, 0, w->V_r_w, DC, w->V_r_b, v_r, DC, R, DC, DC, 0, s); if (st) return st;
    st = launch_linear(fd, DC, R, 0, 0, w->Q_d_w, DC, w->Q_d_b, q_d, DC, R, DC, DC, 0, s); if (st) return st;
    st = launch_linear(fd, DC, R, 0, 0, w->V_d_w, DC, w->V_d_b, v_d, DC, R, DC, DC, 0, s); if (st) return st;
    // ---- four deformable attentions (:736-821): R2R, D2D, D2R (depth queries sample the RGB values, gated into the RGB
    //      path), R2D (the mirror image) ------------------------------------------------------------------------------------
    struct Op { const float* q; const float* v; float* f; const float* gate; const float *sw, *sb, *aww, *awb, *fw, *fb, *ng, *nb; };
    Op ops[4] = {{q_r, v_r, fr, nullptr, w->r2r_sel_w, w->r2r_sel_b, w->r2r_aw_w, w->r2r_aw_b, w->r2r_ffn_w, w->r2r_ffn_b, w->r2r_norm_g, w->r2r_norm_b},
                 {q_d, v_d, fd, nullptr, w->d2d_sel_w, w->d2d_sel_b, w->d2d_aw_w, w->d2d_aw_b, w->d2d_ffn_w, w->d2d_ffn_b, w->d2d_norm_g, w->d2d_norm_b},
                 {q_d, v_r, fr, rgb_f, w->d2r_sel_w, w->d2r_sel_b, w->d2r_aw_w, w->d2r_aw_b, w->d2r_ffn_w, w->d2r_ffn_b, w->d2r_norm_g, w->d2r_norm_b},
                 {q_r, v_d, fd, depth_f, w->r2d_sel_w, w->r2d_sel_b, w->r2d_aw_w, w->r2d_aw_b, w->r2d_ffn_w, w->r2d_ffn_b, w->r2d_norm_g, w->r2d_norm_b}};
    for (int o = 0; o < 4; ++o) {
        const Op& O = ops[o];
        st = launch_linear(O.q, DC, R, 0, 0, O.sw, DC, O.sb, sel, 48, R, DC, 48, 2, s); if (st) return st;
        st = launch_linear(O.q, DC, R, 0, 0, O.aww, DC, O.awb, aw, 24, R, DC, 24, 0, s); if (st) return st;
        hipLaunchKernelGGL(ibl_dator_sample_kernel, dim3(R), dim3(128), 0, s, sel, aw, O.v, samp);
        IBL_LAUNCH_CHECK();
        st = launch_linear(samp, DC, R, 0, 0, O.fw, DC, O.fb, feat, DC, R, DC, DC, 0, s); if (st) return st;
        hipLaunchKernelGGL(ibl_dator_add_ln_kernel, dim3((R + 3) / 4), dim3(256), 0, s, O.f, feat, O.gate, O.ng, O.nb, R);
        IBL_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(ibl_dator_pool_kernel, dim3(B), dim3(128), 0, s, fd, fr, depth_f, rgb_f, out);
    IBL_LAUNCH_CHECK();
    return IBL_OK;
}

// ------------------------------------------------------------------------------------------------
// depth preprocessing: bilinear resize (cv2.INTER_LINEAR convention) -> clip -> scale -> normalise -> 3 identical
// channels -> fp16 im2col patch matrix of the depth stream
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ibl_depth_prep_kernel(const float* __restrict__ src, const int64_t* __restrict__ offs,
                                                             const int* __restrict__ sizes, int out_h, int out_w, int patch, int kpad,
                                                             float dmin, float dmax, unsigned short* __restrict__ patches) {
    const int b = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= out_h * out_w) return;
    const int oy = idx / out_w, ox = idx - oy * out_w;
    const int h = sizes[2 * b], w = sizes[2 * b + 1];
    const float* d = src + offs[b];
    auto coord = [](int o, int osz, int n, int& i0, int& i1, float& fr) {
        const float f = ((float)o + 0.5f) * ((float)n / (float)osz) - 0.5f;
        int i = (int)floorf(f);
        fr = f - (float)i;
        if (i < 0) { i = 0; fr = 0.f; }
        if (i >= n - 1) { i = n - 1; fr = 0.f; }
        i0 = i;
        i1 = i + 1 < n ? i + 1 : n - 1;
        if (i >= n - 1) i1 = i0;
    };
    int y0, y1, x0, x1;
    float fy, fx;
    coord(oy, out_h, h, y0, y1, fy);
    coord(ox, out_w, w, x0, x1, fx);
    const float top = d[(int64_t)y0 * w + x0] * (1.f - fx) + d[(int64_t)y0 * w + x1] * fx;
    const float bot = d[(int64_t)y1 * w + x0] * (1.f - fx) + d[(int64_t)y1 * w + x1] * fx;
    float r = top * (1.f - fy) + bot * fy;
    r = fminf(fmaxf(r, dmin), dmax);
    r = (r - dmin) / (dmax - dmin);
    r = (r - 0.5f) / 0.5f;
    const unsigned short v = __builtin_bit_cast(unsigned short, (_Float16)r);
    const int gw = out_w / patch;
    const int py = oy / patch, ky = oy - py * patch, px = ox / patch, kx = ox - px * patch;
    const int64_t prow = (int64_t)b * (out_h / patch) * gw + py * gw + px;
    for (int c = 0; c < 3; ++c) patches[prow * kpad + c * patch * patch + ky * patch + kx] = v;
    if (ky == 0 && kx == 0)
        for (int k = 3 * patch * patch; k < kpad; ++k) patches[prow * kpad + k] = 0;
}

extern "C" int ibl_preprocess_depth(const float* src, const int64_t* offsets, const int32_t* sizes, int n_crops, int out_h, int out_w,
                                    int patch, int patch_k_pad, float dmin, float dmax, void* patches, void* stream) {
    if (!src || !offsets || !sizes || !patches || n_crops <= 0 || out_h % patch || out_w % patch || dmax <= dmin)
        return ibl_set_error(IBL_ERR_ARG, "ibl_preprocess_depth: bad argument");
    dim3 grid((out_h * out_w + 255) / 256, n_crops);
    hipLaunchKernelGGL(ibl_depth_prep_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, offsets, sizes, out_h, out_w, patch, patch_k_pad,
                       dmin, dmax, reinterpret_cast<unsigned short*>(patches));
    IBL_LAUNCH_CHECK();
    return IBL_OK;
}
