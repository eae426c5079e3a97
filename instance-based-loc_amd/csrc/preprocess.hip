// preprocess.hip -- object-crop preprocessing on gfx950: PIL-exact separable u8 resample (bicubic /
// bilinear with antialiasing), centre-crop window, channel swap, normalisation, im2col to the fp16
// patch matrix the patch-embedding GEMM consumes.
//
// Replaces the CPU PIL / HF-processor step of every reference embedding function
// (utils/embeddings.py:41-42 CLIP, :64-65 DINOv2, :86-89 ViT; dator/get_embeds.py:80-87 for DATOR).
// The resample arithmetic is Pillow's 8-bit fixed-point scheme (22 fractional bits, two passes with
// a u8 intermediate), so the u8 result is bit-identical to PIL.Image.resize; the coefficient tables
// are computed on the host in float64 exactly as Pillow's precompute_coeffs does.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "ibl_common.h"
#include "ibloc.h"

#define PRECISION_BITS 22

__device__ __forceinline__ unsigned char clip8(int v) {
    v >>= PRECISION_BITS;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: tmp[row][ox][c] for all source rows, window columns only
__global__ __launch_bounds__(256) void ibl_resample_h_kernel(const unsigned char* __restrict__ src,
                                                             const ibl_crop_desc* __restrict__ descs,
                                                             const int* __restrict__ tables,
                                                             unsigned char* __restrict__ tmp, int out_w) {
    const ibl_crop_desc d = descs[blockIdx.y];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= d.in_h * out_w) return;
    const int row = idx / out_w, ox = idx - row * out_w;
    const unsigned char* srow = src + d.src_offset + (int64_t)row * d.in_w * 3;
    unsigned char* o = tmp + d.tmp_offset + ((int64_t)row * out_w + ox) * 3;
    if (d.h_ksize == 0) {   // identity pass: h_table is the first source column of the window
        const unsigned char* p = srow + (int64_t)(d.h_table + ox) * 3;
        o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
        return;
    }
    const int* rec = tables + d.h_table + ox * (2 + d.h_ksize);
    const int x0 = rec[0], n = rec[1];
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int t = 0; t < n; ++t) {
        const int k = rec[2 + t];
        const unsigned char* p = srow + (int64_t)(x0 + t) * 3;
        s0 += p[0] * k; s1 += p[1] * k; s2 += p[2] * k;
    }
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

// vertical pass + window + normalise + im2col
__global__ __launch_bounds__(256) void ibl_resample_v_kernel(const unsigned char* __restrict__ tmp,
                                                             const ibl_crop_desc* __restrict__ descs,
                                                             const int* __restrict__ tables, int out_h, int out_w,
                                                             int patch, int patch_k_pad, int swap_rb, float3 mean,
                                                             float3 stdv, unsigned short* __restrict__ patches,
                                                             unsigned char* __restrict__ out_u8) {
    const int b = blockIdx.y;
    const ibl_crop_desc d = descs[b];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= out_h * out_w) return;
    const int oy = idx / out_w, ox = idx - oy * out_w;
    const unsigned char* base = tmp + d.tmp_offset;
    unsigned char px[3];
    if (d.v_ksize == 0) {
        const unsigned char* p = base + ((int64_t)(d.v_table + oy) * out_w + ox) * 3;
        px[0] = p[0]; px[1] = p[1]; px[2] = p[2];
    } else {
        const int* rec = tables + d.v_table + oy * (2 + d.v_ksize);
        const int y0 = rec[0], n = rec[1];
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int t = 0; t < n; ++t) {
            const int k = rec[2 + t];
            const unsigned char* p = base + ((int64_t)(y0 + t) * out_w + ox) * 3;
            s0 += p[0] * k; s1 += p[1] * k; s2 += p[2] * k;
        }
        px[0] = clip8(s0); px[1] = clip8(s1); px[2] = clip8(s2);
    }
    if (out_u8) {   // in MODEL channel order (after the swap), HWC
        unsigned char* o = out_u8 + (((int64_t)b * out_h + oy) * out_w + ox) * 3;
        o[0] = px[swap_rb ? 2 : 0]; o[1] = px[1]; o[2] = px[swap_rb ? 0 : 2];
    }
    const int gw = out_w / patch;
    const int py = oy / patch, ky = oy - py * patch, pxi = ox / patch, kx = ox - pxi * patch;
    const int64_t prow = (int64_t)b * (out_h / patch) * gw + py * gw + pxi;
    const float mm[3] = {mean.x, mean.y, mean.z}, ss[3] = {stdv.x, stdv.y, stdv.z};
#pragma unroll
    for (int c = 0; c < 3; ++c) {           // c = model channel
        const unsigned char u = px[swap_rb ? 2 - c : c];
        const float r = (float)((double)u * (1.0 / 255.0));
        const float v = (r - mm[c]) / ss[c];
        const _Float16 h = (_Float16)v;        // fp16 operand of the patch-embedding GEMM (round to nearest even)
        patches[prow * patch_k_pad + c * patch * patch + ky * patch + kx] = __builtin_bit_cast(unsigned short, h);
    }
}

// zero the K padding columns of the patch matrix
__global__ void ibl_zero_pad_kernel(unsigned short* __restrict__ patches, int64_t rows, int k_real, int k_pad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int npad = k_pad - k_real;
    if (i >= rows * npad) return;
    const int64_t r = i / npad;
    const int c = (int)(i - r * npad);
    patches[r * k_pad + k_real + c] = 0;
}

extern "C" int ibl_preprocess_crops(const uint8_t* src, const ibl_crop_desc* descs, int n_crops, int max_in_h,
                                    const int32_t* tables,
                                    uint8_t* tmp, int out_h, int out_w, int patch, int patch_k_pad, int swap_rb,
                                    const float* mean, const float* stdv, void* patches, uint8_t* out_u8, void* stream) {
    if (!src || !descs || !tmp || !mean || !stdv || !patches)
        return ibl_set_error(IBL_ERR_ARG, "ibl_preprocess_crops: null pointer");
    if (max_in_h <= 0 || n_crops <= 0 || out_h <= 0 || out_w <= 0 || patch <= 0 || out_h % patch || out_w % patch ||
        patch_k_pad < 3 * patch * patch)
        return ibl_set_error(IBL_ERR_ARG, "ibl_preprocess_crops: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    // grid.x is sized for the tallest source crop of the batch; threads beyond in_h*out_w exit
    dim3 gh((unsigned)(((int64_t)max_in_h * out_w + 255) / 256), (unsigned)n_crops);
    hipLaunchKernelGGL(ibl_resample_h_kernel, gh, dim3(256), 0, s, src, descs, tables, tmp, out_w);
    IBL_LAUNCH_CHECK();
    dim3 gv((unsigned)((out_h * out_w + 255) / 256), (unsigned)n_crops);
    hipLaunchKernelGGL(ibl_resample_v_kernel, gv, dim3(256), 0, s, tmp, descs, tables, out_h, out_w, patch,
                       patch_k_pad, swap_rb, make_float3(mean[0], mean[1], mean[2]),
                       make_float3(stdv[0], stdv[1], stdv[2]), reinterpret_cast<unsigned short*>(patches), out_u8);
    IBL_LAUNCH_CHECK();
    const int k_real = 3 * patch * patch;
    if (patch_k_pad > k_real) {
        const int64_t rows = (int64_t)n_crops * (out_h / patch) * (out_w / patch);
        const int64_t n = rows * (patch_k_pad - k_real);
        hipLaunchKernelGGL(ibl_zero_pad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                           reinterpret_cast<unsigned short*>(patches), rows, k_real, patch_k_pad);
        IBL_LAUNCH_CHECK();
    }
    return IBL_OK;
}
