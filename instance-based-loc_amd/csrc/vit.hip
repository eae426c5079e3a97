// vit.hip -- ViT encoder forward on gfx950 (fp16 MFMA 16x16x32, fp32 accumulate, fp32 residual).
//
// Replaces the torch/cuDNN forward the reference reaches through
//   utils/embeddings.py:46 (CLIP ViT-B/32), :69 (DINOv2), :93 (ViT-B/16), :119 (DATOR streams,
//   dator/model/backbones/vit_pytorch.py:422-443)
// one crop at a time; here a whole batch of crops runs per call.
//
// HBM layout (all row-major, token-major): residual stream x fp32 [B*T][D]; LayerNorm output,
// QKV, attention output and MLP hidden as fp16 (11-bit significand: rel-L2 1.2e-3 on the 12-layer ViT-B/14 where bf16 operands gave 1.1e-2, same MFMA rate); weights fp16 [N][K] (K contiguous, i.e. the
// nn.Linear layout) so that both MFMA operands are 16-byte K-contiguous fragments.
//
// Kernels: gemm_f16_tn (256x256x64 / 128x128x64 LDS tiles filled by direct-to-LDS buffer loads, software-pipelined K loop, fused epilogues:
// bias / bias+GELU / bias*layerscale+residual / patch-embed scatter+pos-embed), layernorm (one
// wave per row), attention (one workgroup per (crop, head), K and V^T of the head staged in LDS,
// S^T = K Q^T so that the softmaxed probabilities are already the A operand of P V), CLS/final
// LayerNorm.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <atomic>

#include "ibl_common.h"
#include "ibloc.h"

typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short u16;

// fp32 -> fp16 operand, round-to-nearest-even (v_cvt_f16_f32), clamped to the finite range: an activation beyond 65504 would
// otherwise become an infinity and poison the row (not reached by ViT-B activations; the residual stream itself stays fp32)
__device__ __forceinline__ u16 f2h(float f) {
    const _Float16 h = (_Float16)__builtin_amdgcn_fmed3f(f, -65504.0f, 65504.0f);
    return __builtin_bit_cast(u16, h);
}

__device__ __forceinline__ float h2f(u16 h) { return (float)__builtin_bit_cast(_Float16, h); }

// two fp32 -> two fp16 operands in one dword: v_cvt_pk_f16_f32 (gfx950: packed round-to-nearest-even) + packed clamp to the finite range
// (an infinity becomes 65504: the same value f2h gives) -- three instructions per pair where med3 + cvt per element + pack took five
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned f2h_pk(float a, float b) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    h16x2 r = __builtin_convertvector(f32x2_{a, b}, h16x2);
    const h16x2 hi = {(_Float16)65504.0f, (_Float16)65504.0f};
    r = __builtin_elementwise_min(r, hi);
    r = __builtin_elementwise_max(r, -hi);
    return __builtin_bit_cast(unsigned, r);
}

// ------------------------------------------------------------------------------------------------
// GEMM  C[M][N] = A[M][K] * W[N][K]^T  (+ epilogue)
// ------------------------------------------------------------------------------------------------
enum { EPI_BIAS_H16 = 0, EPI_BIAS_GELU_H16 = 1, EPI_RESID_F32 = 2, EPI_PATCH_F32 = 3, EPI_BIAS_F32 = 4,
       // x += alpha * (a W^T + bias), alpha a power of two, with the residual tile PRELOADED into the accumulators: the MFMA chain starts
       // from (x + bias) / alpha and the epilogue is 32 plain stores per lane -- no read-modify-write after the K loop (round 4)
       EPI_RESID_PRE_F32 = 5,
       // EPI_BIAS_GELU_H16 writing K-extended operand rows for the next GEMM (round 4; ldo = terms * N):
       //   KX2: [h | h / S]  (fc2 weights in two terms)     KX3: [h | (value - h) * S | h / S]  (+ the hidden layer's second term)
       EPI_BIAS_GELU_H16KX2 = 6, EPI_BIAS_GELU_H16KX3 = 7 };

struct GemmEpi {
    const float* bias;      // [N] or null
    const float* scale;     // [N] LayerScale or null (EPI_RESID)
    const float* pos;       // [P][N] position embedding rows for patches (EPI_PATCH)
    void* out;              // fp16 or fp32
    int64_t ldo;            // output row stride (elements)
    int tokens_per_crop;    // T (EPI_PATCH)
    int patches_per_crop;   // P (EPI_PATCH)
    int accumulate;         // EPI_PATCH: x += alpha * acc instead of x = acc + bias + pos (second weight term of a two-term operand)
    float alpha;
    int algo_k;             // K the in-process timer counts as algorithmic (0: K itself; -1: none -- extra terms of a two-term operand are overhead, not model FLOPs)
    int stagger;            // persistent grid: workgroup b sleeps (b % 4) * stagger * 64 clocks before its first tile (launch_gemm_cfg)
};

constexpr int GBK = 64;                 // K granule every caller guarantees (K % 64 == 0)

// LDS images: a tile row holds 8 chunks of 16 bytes; chunk c of row r lives at chunk c ^ key(r).
//   activation tile: key = r & 7            (fragment rows are consecutive -> conflict-free ds_read_b128)
//   weight tile:     key = ((r >> 4) & 3) * 2 + ((r >> 1) & 1)   (fragment rows are 16a + 4j + b, see below)
__device__ __forceinline__ int key_act(int r) { return r & 7; }
__device__ __forceinline__ int key_w(int r) { return (((r >> 4) & 3) << 1) | ((r >> 1) & 1); }
__device__ __forceinline__ int key_pair(int r) { return (((r >> 3) & 3) << 1) | ((r >> 1) & 1); }   // rows 8 a + 4 b + c, a = 0..3, c = 0..3
// 64-byte tile rows (BK = 32): 4 chunks per row, four rows per 256-byte bank row; the same three fragment-row maps are
// conflict-free in every 16-lane ds_read_b128 group with a one-bit key (found by enumeration, tools/lds_keys.py)
template <int BK> __device__ __forceinline__ int keyx_act(int r) { return BK == 64 ? key_act(r) : ((r >> 2) & 1) << 1; }
template <int BK> __device__ __forceinline__ int keyx_w(int r) { return BK == 64 ? key_w(r) : ((r >> 4) & 1) << 1; }
template <int BK> __device__ __forceinline__ int keyx_pair(int r) { return BK == 64 ? key_pair(r) : ((r >> 3) & 1) << 1; }

// C[M][N] = A[M][K] * W[N][K]^T.  Block tile (WM * MI * 16) x (WN * 64) x BK, one wave per (MI * 16) x 64 sub-tile, two LDS stages:
//   <MI = 4, WM = 2, WN = 2, BK = 64>: 128 x 128, 256 threads, 64 KiB LDS (2 blocks / CU)  -- any N % 128 == 0
//   <MI = 8, WM = 2, WN = 4, BK = 64>: 256 x 256, 512 threads, 128 KiB LDS (1 block / CU) -- N % 256 == 0; halves the operand
//                             traffic per FLOP (the 128 x 128 form saturates the L2 path at ~7 TB/s)
//   <MI = 8, WM = 2, WN = 2, BK = 32>: 256 x 128, 256 threads, 48 KiB LDS (2 blocks / CU); measured slower than 256 x 256 on
//                             every ViT-B shape (launch_gemm, IBL_GEMM_CFG=2)
// The MFMA computes the TRANSPOSED tile (W is the A operand, the activations the B operand) and MFMA row 4*fg + r of
// n-tile j is mapped to weight row 16*fg + 4*j + r, so that every lane ends up with 16 CONSECUTIVE output columns of one
// output row: the epilogue is 16-byte vector stores.  Operand tiles are staged with direct-to-LDS loads
// (buffer_load_dwordx4 ... offen lds): one wave instruction writes 1 KiB = 8 tile rows linearly, so the XOR swizzle is applied
// to the per-lane SOURCE offset.
// Exact-form GELU, 0.5 v (1 + erf(v / sqrt 2)), with erfc from Abramowitz & Stegun 7.1.26 (|error| < 1.5e-7 in erf, two orders
// below the fp16 rounding of the stored activation).  libm's erff costs ~35 VALU instructions per element, which made the
// fc1 epilogue as long as its K loop; this form is 14.
__device__ __forceinline__ float gelu_erf(float v) {
    const float z = fabsf(v) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float c = p * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);      // erfc(z)
    return 0.5f * v * (v < 0.f ? c : 2.0f - c);
}

// Two elements at a time on the packed fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two lanes of arithmetic per
// instruction).  Same erfc polynomial; the sign select is folded away with |v| = sqrt(2) z:
//   0.5 v (1 + erf(v / sqrt 2)) = 0.5 v + 0.5 |v| (1 - erfc(z)) = 0.5 v + (z / sqrt 2) (1 - erfc(z))
// (for v << 0 the two terms cancel to an absolute error of ~|v| 6e-8, three orders below the fp16 rounding of the row's other
// activations).  13 packed + 3 scalar operations per pair against 19 scalar per element: the fc1 epilogue was VALU-bound.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 v) {
    f32x2 z;
    z.x = fabsf(v.x) * 0.70710678118654752f;
    z.y = fabsf(v.y) * 0.70710678118654752f;
    const f32x2 d = z * 0.3275911f + 1.0f;
    f32x2 t;
    t.x = __builtin_amdgcn_rcpf(d.x);
    t.y = __builtin_amdgcn_rcpf(d.y);
    f32x2 p = t * 1.061405429f + -1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t + -0.284496736f;
    p = p * t + 0.254829592f;
    const f32x2 a = (z * -1.4426950408889634f) * z;
    f32x2 e;
    e.x = __builtin_amdgcn_exp2f(a.x);
    e.y = __builtin_amdgcn_exp2f(a.y);
    const f32x2 u = 1.0f - (p * t) * e;                       // erf(z)
    return (z * u) * 0.70710678118654752f + v * 0.5f;
}

#ifdef IBL_GEMM_STAMPS     // lab builds only (tools/perf_gemm.py --stamps): phase clocks of every tile a (persistent) block walks
// [block < 512][tile iteration < 16][4]: 0 = tile loop top, 1 = first stage landed (after the barrier), 2 = K loop done, 3 = epilogue issued
__device__ long long ibl_gemm_stamps[512 * 16 * 4];
__device__ int ibl_gemm_stamp_iter[512];
#define GEMM_STAMP(k)                                                                                  \
    do {                                                                                               \
        if (threadIdx.x == 0 && blockIdx.x < 512 && stamp_it < 16)                                     \
            ibl_gemm_stamps[(blockIdx.x * 16 + stamp_it) * 4 + (k)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
extern "C" int ibl_gemm_stamps_read(long long* dst, int n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(ibl_gemm_stamps), sizeof(long long) * n) == hipSuccess ? 0 : -1;
}
extern "C" int ibl_gemm_stamps_clear() {
    static long long zeros[512 * 16 * 4];
    return hipMemcpyToSymbol(HIP_SYMBOL(ibl_gemm_stamps), zeros, sizeof(zeros)) == hipSuccess ? 0 : -1;
}
#else
#define GEMM_STAMP(k)
#endif

#ifdef IBL_GEMM_NOMFMA             // lab: the K loop without its MFMAs -- how fast does a CU stream the operand stages into LDS?
#define IBL_MFMA(a, b, c, x, y, z) (c)
#else
#define IBL_MFMA(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z)
#endif
#ifndef IBL_GEMM_L2AHEAD
#define IBL_GEMM_L2AHEAD 0          // lab (-DIBL_GEMM_L2AHEAD=D): every workgroup requests its share of the activation panel's lines of K step
                                    // kt + D into L2.  Measured per ViT-B layer: 1 034 us off, 1 073 / 1 089 / 1 088 us at D = 2 / 3 / 4 --
                                    // the operand stream is not slow because it misses L2
#endif
#ifndef IBL_GEMM_RMW_PIPE
#define IBL_GEMM_RMW_PIPE 1         // residual epilogue: the loads of a row group ahead of the previous group's stores (0: round 3's serial rounds)
#endif
#ifndef IBL_GEMM_TOUCH
#define IBL_GEMM_TOUCH 0            // lab (-DIBL_GEMM_TOUCH=1): L2 touches of the next tile's first stages, one tile early.  Measured: the wait
                                    // at the tile top 6.2 -> 4.9 k clocks, but the K step that carries the touches waits for them (K loop
                                    // 37.2 -> 37.9 k, fc1 35.6 -> 38.5 k): encoder forward 15.65 -> 16.34 ms.  The first stages are not
                                    // late because they miss L2: 112 KB at the ~21 B / clock a CU streams into LDS take 5 k clocks
#endif
template <int EPI, int MI, int WM, int WN, int BK, int OCC, int NS, bool PIPE = false>
__global__ __launch_bounds__(WM * WN * 64, OCC) void ibl_gemm_f16_tn(const u16* __restrict__ A, int64_t lda, const u16* __restrict__ W,
                                                                  int64_t ldw, int M, int N, int K, GemmEpi epi) {
#if defined(__HIP_DEVICE_COMPILE__)          // the buffer-resource type and builtins exist in the device pass only
    constexpr int BM = WM * MI * 16, BN = WN * 64, NW = WM * WN;
    constexpr int LDS_ROW = BK * 2;          // bytes per tile row, XOR-swizzled 16-byte chunks (no padding)
    constexpr int CPR = BK / 8, RPI = 64 / CPR;             // chunks per row; rows per 1 KiB wave instruction
    constexpr int A_BYTES = BM * LDS_ROW, W_BYTES = BN * LDS_ROW, STAGE = A_BYTES + W_BYTES;
    constexpr int GA = BM / RPI / NW, GW = BN / RPI / NW;   // 1 KiB wave instructions per wave and stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // fp32 read-modify-write epilogue: MFMA row 4 fg + r of n-tile j is weight row 16 j + 4 fg + r (natural order), so that one
    // wave instruction covers 64 contiguous bytes of each of its 16 output rows; with the 16-consecutive-columns-per-lane map
    // of the fp16 epilogues every float4 instruction touched 64 separate cache lines and the address coalescer (not HBM) bound
    // the epilogue: 52 k clocks per tile, more than the projection's K loop.
    constexpr bool NAT = EPI == EPI_RESID_F32 || EPI == EPI_RESID_PRE_F32;
    constexpr bool PRE = EPI == EPI_RESID_PRE_F32;
    // fp16 epilogues: MFMA row 4 fg + r of n-tile j is weight row 32 (j / 2) + 8 fg + 4 (j % 2) + r: a lane owns 8 consecutive
    // columns (one 16-byte store) of n-tile pair j / 2 and the four lane groups cover 64 contiguous bytes of the row
    constexpr int GT = EPI == EPI_BIAS_GELU_H16KX3 ? 3 : (EPI == EPI_BIAS_GELU_H16KX2 ? 2 : 1);     // column blocks the fp16 epilogue writes
    constexpr bool GELU = EPI == EPI_BIAS_GELU_H16 || GT > 1;
    constexpr bool PAIR = EPI == EPI_BIAS_H16 || GELU;
    constexpr int PAIR_STORES = GT * 2 * MI;          // stores per lane of a full tile's fp16 epilogue
#define KEYW(r) (NAT ? keyx_act<BK>(r) : (PAIR ? keyx_pair<BK>(r) : keyx_w<BK>(r)))
    const int nbn = N / BN;
    const int nbm = (M + BM - 1) / BM;
    const int nwg = nbn * nbm;
    // source of piece i: a buffer resource on the tile's first row (SGPRs) + a per-lane 32-bit byte offset + the K offset (SGPR):
    // buffer_load_dwordx4 ... offen lds needs no 64-bit address pair per piece (16 VGPRs with global_load_lds)
    __amdgpu_buffer_rsrc_t a_rsrc, w_rsrc;
    unsigned a_off[GA], w_off[GW];
    int row0, col0;
    // tile t of the launch -> (row0, col0) and this wave's source addresses.  XCD-aware remap: consecutive tiles along N (sharing
    // the A panel) stay on one XCD's L2 (t % 8 labels the XCD for the one-shot grid and for the persistent one, whose stride is a
    // multiple of 8)
#define GEMM_SET_TILE(t)                                                                           \
    do {                                                                                           \
        int _bid = (t);                                                                            \
        {                                                                                          \
            const int _q = nwg / 8, _r = nwg % 8, _xcd = _bid % 8;                                 \
            _bid = (_xcd < _r ? _xcd * (_q + 1) : _r * (_q + 1) + (_xcd - _r) * _q) + _bid / 8;    \
        }                                                                                          \
        row0 = (_bid / nbn) * BM;                                                                  \
        col0 = (_bid % nbn) * BN;                                                                  \
        a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (int64_t)row0 * lda), 0, -1, 0x00020000); \
        w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (int64_t)col0 * ldw), 0, -1, 0x00020000); \
        _Pragma("unroll") for (int _i = 0; _i < GA; ++_i) {                                        \
            const int _rr = (wave + NW * _i) * RPI + lane / CPR, _pch = lane % CPR;                \
            int _ar = _rr;                                                                         \
            if (row0 + _ar >= M) _ar = M - 1 - row0;                                               \
            a_off[_i] = (unsigned)(((int64_t)_ar * lda + ((_pch ^ keyx_act<BK>(_rr)) << 3)) * 2);  \
        }                                                                                          \
        _Pragma("unroll") for (int _i = 0; _i < GW; ++_i) {                                        \
            const int _rr = (wave + NW * _i) * RPI + lane / CPR, _pch = lane % CPR;                \
            w_off[_i] = (unsigned)(((int64_t)_rr * ldw + ((_pch ^ KEYW(_rr)) << 3)) * 2);          \
        }                                                                                          \
    } while (0)
    int tile = blockIdx.x;
    if constexpr (PIPE) {
        // Staggered start (residual epilogue only, set by the launcher): the workgroups of a persistent grid run in lockstep -- all K
        // loops together (HBM nearly idle), then all fp32 read-modify-write epilogues together (HBM saturated at 5.8 TB/s, MFMA idle).
        // Four phase groups, a quarter of a tile apart, spread the epilogue traffic over the whole launch.
        if (epi.stagger > 0) {
            const int ph = ((blockIdx.x >> 3) & 3) * epi.stagger;      // (blockIdx % 8 labels the XCD: every XCD gets all four phases)
            for (int i = 0; i < ph; ++i) __builtin_amdgcn_s_sleep(64);
        }
    }
    GEMM_SET_TILE(tile);
#define GEMM_GLDS(buf, kt)                                                                                              \
    do {                                                                                                                \
        const int64_t _ko = (int64_t)(kt) * BK;                                                                         \
        _Pragma("unroll") for (int _i = 0; _i < GA; ++_i)                                                               \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(smem + (buf) * STAGE + (wave + NW * _i) * 1024), \
                                                     16, a_off[_i], (unsigned)(_ko * 2), 0, 0);                          \
        _Pragma("unroll") for (int _i = 0; _i < GW; ++_i)                                                               \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(smem + (buf) * STAGE + A_BYTES + (wave + NW * _i) * 1024), \
                                                     16, w_off[_i], (unsigned)(_ko * 2), 0, 0);                          \
    } while (0)

    f32x4 acc[MI][4];      // [m-tile i][n-tile j]
    const int nk = K / BK;
    constexpr int NP = GA + GW;
    const int fr = lane & 15, fg = lane >> 4;
    int arow[MI], wrow[4];
#pragma unroll
    for (int i = 0; i < MI; ++i) arow[i] = wm * (MI * 16) + i * 16 + fr;              // activation row (B operand column)
#pragma unroll
    for (int j = 0; j < 4; ++j)                   // weight row of MFMA row fr in n-tile j
        wrow[j] = NAT ? wn * 64 + 16 * j + fr
                      : (PAIR ? wn * 64 + 32 * (j >> 1) + 8 * (fr >> 2) + 4 * (j & 1) + (fr & 3) : wn * 64 + 16 * (fr >> 2) + 4 * j + (fr & 3));
#define GEMM_PIECE(stage, pc)                                                                                                       \
    do {                                                                                                                            \
        const int64_t _ko = (int64_t)(stage) * BK;                                                                                  \
        unsigned char* _dst = smem + ((stage) & 1) * STAGE;                                                                         \
        if ((pc) < GA)                                                                                                              \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(_dst + (wave + NW * (pc)) * 1024), 16,   \
                                                     a_off[(pc) < GA ? (pc) : 0], (unsigned)(_ko * 2), 0, 0);                               \
        else                                                                                                                        \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(_dst + A_BYTES + (wave + NW * ((pc) - GA)) * 1024), \
                                                     16, w_off[(pc) >= GA ? (pc) - GA : 0], (unsigned)(_ko * 2), 0, 0);                     \
    } while (0)
#define FRAG_W(buf, ks, j) (*reinterpret_cast<const h16x8*>(smem + (buf) * STAGE + A_BYTES + wrow[j] * LDS_ROW + (((4 * (ks) + fg) ^ KEYW(wrow[j])) << 4)))
#define FRAG_A(buf, ks, i) (*reinterpret_cast<const h16x8*>(smem + (buf) * STAGE + arow[i] * LDS_ROW + (((4 * (ks) + fg) ^ keyx_act<BK>(arow[i])) << 4)))
    // The pipelined form is launched as a persistent grid (one workgroup per CU walks the tiles t, t + grid, ...): the next tile's
    // first stage is requested before the epilogue of the current one, so its latency hides under the stores.
    static_assert(!PIPE || (BK == 64 && NS == 2 && MI % 2 == 0), "pipelined loop: BK 64, two stages");
#ifndef IBL_GEMM_HA_DIV
#define IBL_GEMM_HA_DIV 4
#endif
    constexpr int HA = MI / IBL_GEMM_HA_DIV, N2B = MI - HA;       // groups before / after the barrier in the second K half
    constexpr int NL = MI + 4;                                    // fragment reads per K half
    constexpr int LB = (NL + MI - 2) / (MI - 1);                  // set-B reads per phase-1 group (none after the last group)
    constexpr int LA = (NL + N2B - 2) / (N2B - 1);                // set-A reads per phase-2b group (none before the last group)
    constexpr int NSLOT = N2B + MI;
    static_assert(!PIPE || NP <= 2 * NSLOT, "pieces per stage exceed the slots of a step");
    bool first_tile = true;
    bool stores_pending = false;      // the previous tile's epilogue issued exactly 2 * MI stores per lane after this tile's first pieces
#ifdef IBL_GEMM_STAMPS
    int stamp_it = 0;
#endif
    for (;;) {
    GEMM_STAMP(0);
    if constexpr (PRE) {
        // Residual tile -> accumulators.  Per-tile stamps (round 4): the fp32 read-modify-write AFTER the K loop cost 38 k clocks of a proj
        // tile's 85 k (eight dependent rounds of 4 loads / 4 stores per lane, each waiting for the previous round's stores: vmcnt counts
        // stores on gfx9).  Here the 32 float4 of the tile are requested at the tile top, where the wait for the first operand stage
        // already sits, and the epilogue only stores.  Lane (fr, fg): rows row0 + wm * 128 + 16 i + fr, columns col0 + wn * 64 + 16 j +
        // 4 fg .. + 3 (rows beyond M read row M - 1; their lanes store nothing).
        const int nbp = col0 + wn * 64 + 4 * fg;
        const float ia = 1.0f / epi.alpha;                    // alpha is a power of two: exact
        if (row0 + BM <= M) {
            const float* const xb = reinterpret_cast<const float*>(epi.out) + (int64_t)(row0 + wm * (MI * 16) + fr) * epi.ldo + nbp;
            const int64_t gstride = 16 * epi.ldo;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 x4 = *reinterpret_cast<const float4*>(xb + i * gstride + 16 * j);
                    acc[i][j] = f32x4{x4.x, x4.y, x4.z, x4.w};
                }
        } else {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                int row = row0 + wm * (MI * 16) + i * 16 + fr;
                row = row < M ? row : M - 1;
                const float* xr = reinterpret_cast<const float*>(epi.out) + (int64_t)row * epi.ldo + nbp;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 x4 = *reinterpret_cast<const float4*>(xr + 16 * j);
                    acc[i][j] = f32x4{x4.x, x4.y, x4.z, x4.w};
                }
            }
        }
        if (epi.bias) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 b4 = *reinterpret_cast<const float4*>(epi.bias + nbp + 16 * j);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    acc[i][j][0] += b4.x; acc[i][j][1] += b4.y; acc[i][j][2] += b4.z; acc[i][j][3] += b4.w;
                }
            }
        }
        if (epi.alpha != 1.0f) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] *= ia;
        }
        // (the compiler's own wait for these loads lands here, before the K loop: without the empty asm it would sit in front of the
        // first MFMA of every accumulator, i.e. between the direct-to-LDS pieces of the first K step, as vmcnt(0))
#pragma unroll
        for (int i = 0; i < MI; ++i)
            asm volatile("" : "+v"(acc[i][0]), "+v"(acc[i][1]), "+v"(acc[i][2]), "+v"(acc[i][3]));
    } else {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (PIPE) {
        // Software-pipelined K loop (BK = 64, two LDS stages, two fragment register sets).  One K step of a wave:
        //   phase 1   MI groups of 4 MFMAs on the step's first K half (set A), the reads of its second half (set B) between them
        //   phase 2a  MI / 2 groups on set B
        //   wait for this wave's pieces of the NEXT stage, workgroup barrier
        //   phase 2b  MI / 2 groups on set B, the reads of the next stage's first half (set A) between them
        // so that no MFMA waits on an LDS read issued just before it (the compiler's own schedule of the plain loop read each
        // fragment pair right before its use: five exposed LDS round trips per step, in all eight waves at once).  The direct-to-LDS
        // pieces of stage s + 2 overwrite the buffer of stage s: the first half of them is issued in phase 2b of step s (every wave has
        // passed the barrier, so every wave's reads of that buffer have returned), the rest in phase 1 of step s + 1; they have a
        // whole step to land before the wait that precedes the barrier of step s + 1.
        // Piece slots: one after each MFMA group of phase 2b (stage kt + 2) and of phase 1 (stage kt + 1, i.e. the same stage one step
        // later); slot u carries piece u, the pieces beyond the slot count ride on the first slots.  (Giving the two waves of a SIMD
        // alternate slots, so that one issues MFMAs while the other waits on its piece, measured 4 % slower.)
#define GEMM_SLOT(u, stage)                                                                        \
    do {                                                                                           \
        if ((u) < NP) GEMM_PIECE(stage, (u) < NP ? (u) : 0);                                       \
        if (NSLOT + (u) < NP) GEMM_PIECE(stage, NSLOT + (u) < NP ? NSLOT + (u) : 0);               \
    } while (0)
#define GEMM_PROLOGUE()                                                                            \
    do {                                                                                           \
        _Pragma("unroll") for (int _pc = 0; _pc < NP; ++_pc) GEMM_PIECE(0, _pc);                   \
        if (nk > 1) {                                   /* the phase-2b slots of a step before the first */ \
            _Pragma("unroll") for (int _g = 0; _g < N2B; ++_g) GEMM_SLOT(_g, 1);                   \
        }                                                                                          \
    } while (0)
        if (first_tile) GEMM_PROLOGUE();
        // fp16 epilogues: the 2 * MI stores of the previous (full) tile were issued AFTER this tile's first pieces, so "at most 2 * MI
        // operations outstanding" already means the pieces have landed (vmcnt retires in issue order): the stores drain under the first K
        // step instead of in front of it (a raw barrier: __syncthreads() would add its own vmcnt(0) while LDS-DMA is pending)
        if (PAIR && stores_pending) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PAIR_STORES) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        GEMM_STAMP(1);
        if (IBL_GEMM_TOUCH && tile + (int)gridDim.x < nwg) {     // issued here, behind the tile-top wait and before the fragments exist:
                                                                 // the K loop has no register to spare (two more live VGPRs spill its LDS addresses)
            int _bid = tile + (int)gridDim.x;
            const int _q = nwg / 8, _r = nwg % 8, _xcd = _bid % 8;
            _bid = (_xcd < _r ? _xcd * (_q + 1) : _r * (_q + 1) + (_xcd - _r) * _q) + _bid / 8;
            const int prow0 = (_bid / nbn) * BM, pcol0 = (_bid % nbn) * BN;
            // (32-bit arithmetic on a freshly formed lane id: the K loop has no register to spare -- two more live VGPRs spill its
            // LDS fragment addresses)
            const int wv = __builtin_amdgcn_readfirstlane(wave);
            const bool act = wv < NW / 2;
            const unsigned ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            unsigned r = (unsigned)((act ? prow0 : pcol0) + (wv % (NW / 2)) * 64) + ln;
            if (act) r = r < (unsigned)M ? r : (unsigned)(M - 1);
            const unsigned touch_off = r * (unsigned)((act ? lda : ldw) * 2);
            const __amdgpu_buffer_rsrc_t t_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(act ? A : W), 0, -1, 0x00020000);
            unsigned char* const tdst = smem + NS * STAGE + wave * 512;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(t_rsrc, (__attribute__((address_space(3))) void*)tdst, 4, touch_off, 0, 0, 0);
            if (nk > 1)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(t_rsrc, (__attribute__((address_space(3))) void*)(tdst + 256), 4, touch_off, BK * 2, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#if IBL_GEMM_L2AHEAD > 0
        // (wave-uniform: SGPRs) which quarter(s) of the row panel this workgroup requests ahead; not on the last row of tiles (rows beyond M)
        const bool l2ahead = row0 + BM <= M && nk > IBL_GEMM_L2AHEAD;
        const int tcol = col0 / BN;
        const bool l2two = nbn < 4;
        const unsigned touch_row0 = (unsigned)((tcol & 3) * 64 * lda * 2);
        const unsigned touch_row1 = (unsigned)(((tcol + nbn) & 3) * 64 * lda * 2);
#endif
        h16x8 afA[MI], wfA[4], afB[MI], wfB[4];
        // L2 touches for the NEXT tile of this block (round 4).  Per-tile stamps: from the request of a tile's first two operand stages to the
        // start of its K loop pass ~14 k clocks whatever the epilogue in between does (shortening the fp16 epilogue by 1.5 k lengthened the
        // wait at the tile top by 1.5 k; 32 instead of 256 active CUs change nothing): the 112 KB are first touches of a new row panel,
        // L2 misses, and a CU sustains ~8.5 B / clock of those.  So the lines of those two stages are requested one tile EARLY, as 4-byte
        // direct-to-LDS loads into a scratch area (no VGPR, no use of the data): one per lane and stage, waves 0 .. NW / 2 - 1 the rows of
        // the activation tile, the others the rows of the weight tile; the pieces issued at the end of this tile then hit L2.
        // (the offsets are formed where the touches are issued: nothing of this lives across the K loop)
#pragma unroll
        for (int j = 0; j < 4; ++j) wfA[j] = FRAG_W(0, 0, j);
#pragma unroll
        for (int i = 0; i < MI; ++i) afA[i] = FRAG_A(0, 0, i);
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
            // ---- phase 1: read t of a K half is weight fragment t / 2 (t even, t < 8) or the next activation fragment
#pragma unroll
            for (int g = 0; g < MI; ++g) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[g][j] = IBL_MFMA(wfA[j], afA[g], acc[g][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);         // reads after the group: the wait before it then covers set A only
#pragma unroll
                for (int t = g * LB; t < (g + 1) * LB && t < NL; ++t) {
                    if (t < 8 && (t & 1) == 0) wfB[t / 2] = FRAG_W(buf, 1, t / 2);
                    else afB[t < 8 ? t / 2 : t - 4] = FRAG_A(buf, 1, t < 8 ? t / 2 : t - 4);
                }
                if (more1) GEMM_SLOT(N2B + g, kt + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- phase 2a
#pragma unroll
            for (int g = 0; g < HA; ++g) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[g][j] = IBL_MFMA(wfB[j], afB[g], acc[g][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#if IBL_GEMM_L2AHEAD > 0
            // Cooperative L2 prefetch of the activation panel (round 4).  The K loop takes the same ~3 000 clocks per step WITHOUT its MFMAs
            // (lab build -DIBL_GEMM_NOMFMA): it is bound by how fast a CU streams its 64 KB of operands into LDS, and the tiles that find
            // their panel in L2 run at 1 300 clocks per step.  The nbn column tiles of a row panel run side by side on one XCD and all miss
            // on the same activation lines at the same time.  So every workgroup requests a QUARTER of its row panel's lines of K step
            // kt + IBL_GEMM_L2AHEAD now (4-byte direct-to-LDS loads into a scratch area: the offsets of piece 0 cover 64 rows, the quarter is
            // chosen by the tile's column), as the LAST memory operation of the step: the wait below then lets exactly that one stay in
            // flight (vmcnt retires in order), and it is the oldest operation when the next step waits.
            if (l2ahead && kt + IBL_GEMM_L2AHEAD < nk) {
                const unsigned ko2 = (unsigned)((kt + IBL_GEMM_L2AHEAD) * BK * 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(smem + NS * STAGE + wave * 512), 4, a_off[0],
                                                         touch_row0 + ko2, 0, 0);
                if (l2two) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(smem + NS * STAGE + wave * 512 + 256), 4,
                                                             a_off[0], touch_row1 + ko2, 0, 0);
                    asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
                }
            } else {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
#else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // this wave's pieces have landed, its set-B reads have returned
            __syncthreads();
#endif
            // ---- phase 2b
            const int nb = buf ^ 1;
#pragma unroll
            for (int g = 0; g < N2B; ++g) {
                if (more1) {
#pragma unroll
                    for (int t = g * LA; t < (g + 1) * LA && t < NL; ++t) {
                        if (t < 8 && (t & 1) == 0) wfA[t / 2] = FRAG_W(nb, 0, t / 2);
                        else afA[t < 8 ? t / 2 : t - 4] = FRAG_A(nb, 0, t < 8 ? t / 2 : t - 4);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[HA + g][j] = IBL_MFMA(wfB[j], afB[HA + g], acc[HA + g][j], 0, 0, 0);
                if (more2) GEMM_SLOT(g, kt + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
    // NS LDS stages: the loads of K steps 0 .. NS - 2 are in flight before the loop, step kt issues those of step kt + NS - 1, and the
    // wait that closes a step lets the (NS - 2) youngest stages stay in flight (vmcnt counts LDS-DMA pieces in issue order)
#pragma unroll
    for (int st = 0; st < NS - 1; ++st)
        if (st < nk) GEMM_GLDS(st, st);
    if (NS == 2 || nk < NS - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"i"((NS - 2) * NP) : "memory");
    __syncthreads();
    GEMM_STAMP(1);
    // One K step = 2 * MI groups of 4 MFMAs.  The GA + GW direct-to-LDS pieces of the NEXT tile are issued one at a time
    // between those groups: a piece blocks its wave's issue port for ~100 cycles, and eight of them back to back at the top
    // of the step (right after the barrier, in every wave at once) left the MFMA pipe idle for a third of the step.
    constexpr int KS = BK / 32, NG = KS * MI;
#ifndef IBL_GEMM_NGI_DIV
#define IBL_GEMM_NGI_DIV 2
#endif
    constexpr int NGI = NG / IBL_GEMM_NGI_DIV;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt % NS;
        const bool more = kt + NS - 1 < nk;
        const int64_t ko = (int64_t)(kt + NS - 1) * BK;
        unsigned char* nxt = smem + ((kt + NS - 1) % NS) * STAGE;
        const unsigned char* pa = smem + buf * STAGE;
        const unsigned char* pw = pa + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int ch = 4 * ks + fg;
            h16x8 af[MI], wf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const h16x8*>(pw + wrow[j] * LDS_ROW + ((ch ^ KEYW(wrow[j])) << 4));
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const h16x8*>(pa + arow[i] * LDS_ROW + ((ch ^ keyx_act<BK>(arow[i])) << 4));
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = IBL_MFMA(wf[j], af[i], acc[i][j], 0, 0, 0);
                const int g = ks * MI + i;
#pragma unroll
                // pieces are issued over the first half of the step's groups: one issued in the last groups has not landed at the
                // vmcnt(0) that closes the step (744 -> 760 TFLOP/s per layer; the first quarter only: 753)
                for (int pc = (g < NGI ? g * NP / NGI : NP); pc < (g < NGI ? (g + 1) * NP / NGI : NP); ++pc) {
                    if (more) {
                        if (pc < GA)
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(nxt + (wave + NW * pc) * 1024), 16,
                                                                     a_off[pc], (unsigned)(ko * 2), 0, 0);
                        else
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(nxt + A_BYTES + (wave + NW * (pc - GA)) * 1024), 16,
                                                                     w_off[pc - GA], (unsigned)(ko * 2), 0, 0);
                    }
                }
            }
        }
        // stage kt + 1 has landed; the pieces of later steps (issued during this step and the ones before) may still be in flight
        if (NS == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            const int later = nk - 2 - kt;               // K steps after kt + 1 ...
            if (later >= NS - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((NS - 2) * NP) : "memory");
            else if (NS >= 4 && later == NS - 3) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((NS >= 4 ? NS - 3 : 0) * NP) : "memory");
            else if (NS >= 5 && later == NS - 4) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((NS >= 5 ? NS - 4 : 0) * NP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
    }
    GEMM_STAMP(2);
    // the tile this block computes next: its first stage is requested now (both LDS buffers are free: every wave's last LDS read
    // returned before the barrier of the final K step)
    const int erow0 = row0, ecol0 = col0;
    // fp16 epilogues: this lane's 16 bias values are fetched -- and waited for -- BEFORE the next tile's direct-to-LDS pieces are issued.
    // Round 4 (per-tile stamps, tools/perf_gemm.py --stamps): "prologue of the next tile + epilogue" took 12.9 k of a qkv tile's 51 k
    // clocks although it stores 128 KB.  The bias loads used to follow the pieces, and with LDS-DMA pieces in flight the compiler waits
    // vmcnt(0) at every use of an ordinary load's result -- in each of the eight per-row-group blocks (`if (row >= M) continue`), i.e.
    // also for the previous group's STORES to complete: eight exposed store latencies per tile.  Now: bias first (the empty asm consumes
    // the registers, so the compiler's wait sits here, with nothing else outstanding), then the pieces, then branch-free stores.
    float bb[2][8];
    if constexpr (PAIR) {
        const int nbp = ecol0 + wn * 64 + 8 * fg;
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 b4 = epi.bias ? *reinterpret_cast<const float4*>(epi.bias + nbp + 32 * j2 + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
                bb[j2][4 * h] = b4.x; bb[j2][4 * h + 1] = b4.y; bb[j2][4 * h + 2] = b4.z; bb[j2][4 * h + 3] = b4.w;
            }
        asm volatile("" ::"v"(bb[0][0]), "v"(bb[0][1]), "v"(bb[0][2]), "v"(bb[0][3]), "v"(bb[0][4]), "v"(bb[0][5]), "v"(bb[0][6]), "v"(bb[0][7]),
                     "v"(bb[1][0]), "v"(bb[1][1]), "v"(bb[1][2]), "v"(bb[1][3]), "v"(bb[1][4]), "v"(bb[1][5]), "v"(bb[1][6]), "v"(bb[1][7]));
    }
    float4 b4[4], s4[4];
    if constexpr (NAT && !PRE) {               // read-modify-write epilogue: bias and scale of this lane's 16 columns, likewise ahead of the pieces
        const int nbq = ecol0 + wn * 64 + 4 * fg;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            b4[j] = epi.bias ? *reinterpret_cast<const float4*>(epi.bias + nbq + 16 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
            s4[j] = epi.scale ? *reinterpret_cast<const float4*>(epi.scale + nbq + 16 * j) : make_float4(1.f, 1.f, 1.f, 1.f);
        }
#ifndef IBL_GEMM_EPI_BATCHED
#pragma unroll
        for (int j = 0; j < 4; ++j)
            asm volatile("" ::"v"(b4[j].x), "v"(b4[j].y), "v"(b4[j].z), "v"(b4[j].w), "v"(s4[j].x), "v"(s4[j].y), "v"(s4[j].z), "v"(s4[j].w));
#endif
    }
    tile += gridDim.x;
    const bool has_next = PIPE && tile < nwg;
    if constexpr (PIPE) {
        if (has_next) {
            GEMM_SET_TILE(tile);
            GEMM_PROLOGUE();
        }
    }


    // epilogue.  D layout: col = lane & 15 -> output row m; MFMA row 4*fg + r of n-tile j -> output column 16*fg + 4*j + r
    if (PRE) {
        // the accumulators hold (x + bias) / alpha + a W^T: 32 plain stores
        const int nb = ecol0 + wn * 64 + 4 * fg;
        const float al = epi.alpha;
        if (erow0 + BM <= M) {
            float* const obase = reinterpret_cast<float*>(epi.out) + (int64_t)(erow0 + wm * (MI * 16) + fr) * epi.ldo + nb;
            const int64_t gstride = 16 * epi.ldo;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<float4*>(obase + i * gstride + 16 * j) =
                        make_float4(acc[i][j][0] * al, acc[i][j][1] * al, acc[i][j][2] * al, acc[i][j][3] * al);
        } else {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = erow0 + wm * (MI * 16) + i * 16 + fr;
                if (row >= M) continue;
                float* orow = reinterpret_cast<float*>(epi.out) + (int64_t)row * epi.ldo + nb;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<float4*>(orow + 16 * j) = make_float4(acc[i][j][0] * al, acc[i][j][1] * al, acc[i][j][2] * al, acc[i][j][3] * al);
            }
        }
    } else if (NAT) {
        // x[row][n] += scale[n] * (acc + bias[n]); lane (fr, fg) owns columns 16 j + 4 fg + 0..3 of n-tile j
        const int nb = ecol0 + wn * 64 + 4 * fg;
#ifdef IBL_GEMM_EPI_BATCHED
        // Round 4 (lab switch, measured SLOWER: proj 131 -> 147 us, fc2 298 -> 313 us -- the compiler folds the additions into the
        // accumulators and keeps three loads in flight whatever the source order; 16 + 16 spills).  The read-modify-write used to run as eight dependent rounds (4 loads, wait, 4 stores per 16-row group -- and vmcnt counts
        // stores too on gfx9, so every round also waited for the previous round's stores): ~16 exposed memory latencies, 34 k clocks per tile
        // at K = 768 where the whole K loop is 35 k.  Now: the increments are formed in place in the accumulators (bias / scale registers
        // die), then the 32 float4 of the residual tile are requested in three batches (12 + 12 + 8) with two batches always in flight,
        // and a batch's stores are issued after the NEXT batch's loads: three exposed latencies.  (16 + 16 spilled: the K loop's
        // 254 registers leave 96 beside the accumulators.)  The last row of tiles keeps the masked serial form.
        {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 b4 = epi.bias ? *reinterpret_cast<const float4*>(epi.bias + nb + 16 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 s4 = epi.scale ? *reinterpret_cast<const float4*>(epi.scale + nb + 16 * j) : make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    acc[i][j][0] = (acc[i][j][0] + b4.x) * s4.x; acc[i][j][1] = (acc[i][j][1] + b4.y) * s4.y;
                    acc[i][j][2] = (acc[i][j][2] + b4.z) * s4.z; acc[i][j][3] = (acc[i][j][3] + b4.w) * s4.w;
                }
            }
        }
        if (erow0 + BM <= M) {                                     // full tile (all but the last row of tiles): no row is clamped or masked
            constexpr int B0 = MI == 8 ? 3 : 2, B1 = MI == 8 ? 3 : 2;  // 16-row groups of batch A, batch B; the rest is batch C (in A's registers)
            float* const obase = reinterpret_cast<float*>(epi.out) + (int64_t)(erow0 + wm * (MI * 16) + fr) * epi.ldo + nb;
            const int64_t gstride = 16 * epi.ldo;
            float4 xa[B0][4], xb[B1][4];
#pragma unroll
            for (int i = 0; i < B0; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) xa[i][j] = *reinterpret_cast<const float4*>(obase + i * gstride + 16 * j);
#pragma unroll
            for (int i = 0; i < B1; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) xb[i][j] = *reinterpret_cast<const float4*>(obase + (B0 + i) * gstride + 16 * j);
            __builtin_amdgcn_sched_barrier(0);
#define RESID_STORE(X, I)                                                                                                           \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                           \
                *reinterpret_cast<float4*>(obase + (I) * gstride + 16 * j) = make_float4(X[j].x + acc[I][j][0], X[j].y + acc[I][j][1], \
                                                                                         X[j].z + acc[I][j][2], X[j].w + acc[I][j][3]);
#pragma unroll
            for (int i = 0; i < B0; ++i) { RESID_STORE(xa[i], i) }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MI - B0 - B1; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) xa[i][j] = *reinterpret_cast<const float4*>(obase + (B0 + B1 + i) * gstride + 16 * j);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < B1; ++i) { RESID_STORE(xb[i], B0 + i) }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MI - B0 - B1; ++i) { RESID_STORE(xa[i], B0 + B1 + i) }
        } else {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = erow0 + wm * (MI * 16) + i * 16 + fr;
                if (row >= M) continue;
                float* orow = reinterpret_cast<float*>(epi.out) + (int64_t)row * epi.ldo + nb;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float4* o = reinterpret_cast<float4*>(orow + 16 * j);
                    float4 x = *o;
                    x.x += acc[i][j][0]; x.y += acc[i][j][1]; x.z += acc[i][j][2]; x.w += acc[i][j][3];
                    *o = x;
                }
            }
        }
#undef RESID_STORE
#else
        // (b4 / s4 were fetched before the next tile's pieces went out, above)
#define RESID_RMW(X, I, PTR)                                                                                                        \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                             \
            float4 x = X[j];                                                                                                        \
            x.x += (acc[I][j][0] + b4[j].x) * s4[j].x; x.y += (acc[I][j][1] + b4[j].y) * s4[j].y;                                   \
            x.z += (acc[I][j][2] + b4[j].z) * s4[j].z; x.w += (acc[I][j][3] + b4[j].w) * s4[j].w;                                   \
            *reinterpret_cast<float4*>((PTR) + 16 * j) = x;                                                                         \
        }
        if (IBL_GEMM_RMW_PIPE && erow0 + BM <= M) {
            // Full tile: the read-modify-write of a 16-row group used to be 4 loads, wait, 4 stores -- and since vmcnt retires in order
            // and counts stores, the wait of group i + 1 also waited for the stores of group i: eight rounds of (store latency + load
            // latency), 38 k clocks of a proj tile's 85 k.  Now the loads of group i + 1 are issued BEFORE the stores of group i (two
            // groups of the residual tile in registers instead of one; same arithmetic, same bits), and a round waits for ONE load:
            //     L0 L1 | wait(4) S0 L2 | wait(8) S1 L3 | ... | wait(8) S6 | wait(4) S7
            // The memory operations and their waits are inline assembly: with loads AND stores pending the compiler treats vmcnt as
            // unordered and waits vmcnt(0) at every use (LLVM SIInsertWaitcnts: mixed pending events), which is the serial form again.
            // No compiler-issued memory operation may sit between them (the kernel has no spills; bias / scale were consumed above).
            float* const obase = reinterpret_cast<float*>(epi.out) + (int64_t)(erow0 + wm * (MI * 16) + fr) * epi.ldo + nb;
            const int64_t gstride = 16 * epi.ldo;
            f32x4 xa[4], xb[4];
#define RMW_LOAD(X, PTR)                                                                                                            \
            do {                                                                                                                    \
                const float* _p = (PTR);                                                                                            \
                asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(X[0]) : "v"(_p) : "memory");                                 \
                asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=&v"(X[1]) : "v"(_p) : "memory");                       \
                asm volatile("global_load_dwordx4 %0, %1, off offset:128" : "=&v"(X[2]) : "v"(_p) : "memory");                      \
                asm volatile("global_load_dwordx4 %0, %1, off offset:192" : "=&v"(X[3]) : "v"(_p) : "memory");                      \
            } while (0)
#define RMW_WAIT(N, X) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3])::"memory")
#define RMW_STORE(X, I, PTR)                                                                                                        \
            do {                                                                                                                    \
                float* _p = (PTR);                                                                                                  \
                f32x4 _v[4];                                                                                                        \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                     \
                    _v[j][0] = X[j][0] + (acc[I][j][0] + b4[j].x) * s4[j].x; _v[j][1] = X[j][1] + (acc[I][j][1] + b4[j].y) * s4[j].y; \
                    _v[j][2] = X[j][2] + (acc[I][j][2] + b4[j].z) * s4[j].z; _v[j][3] = X[j][3] + (acc[I][j][3] + b4[j].w) * s4[j].w; \
                }                                                                                                                   \
                asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(_p), "v"(_v[0]) : "memory");                                  \
                asm volatile("global_store_dwordx4 %0, %1, off offset:64" ::"v"(_p), "v"(_v[1]) : "memory");                        \
                asm volatile("global_store_dwordx4 %0, %1, off offset:128" ::"v"(_p), "v"(_v[2]) : "memory");                       \
                asm volatile("global_store_dwordx4 %0, %1, off offset:192" ::"v"(_p), "v"(_v[3]) : "memory");                       \
            } while (0)
            static_assert(MI % 2 == 0, "read-modify-write epilogue: row groups in pairs");
            RMW_LOAD(xa, obase);
#pragma unroll
            for (int i = 0; i < MI; i += 2) {
                RMW_LOAD(xb, obase + (i + 1) * gstride);
                if (i == 0) RMW_WAIT(4, xa); else RMW_WAIT(8, xa);
                RMW_STORE(xa, i, obase + i * gstride);
                if (i + 2 < MI) {
                    RMW_LOAD(xa, obase + (i + 2) * gstride);
                    RMW_WAIT(8, xb);
                } else {
                    RMW_WAIT(4, xb);
                }
                RMW_STORE(xb, i + 1, obase + (i + 1) * gstride);
            }
#undef RMW_LOAD
#undef RMW_WAIT
#undef RMW_STORE
        } else {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = erow0 + wm * (MI * 16) + i * 16 + fr;
                if (row >= M) continue;
                float* orow = reinterpret_cast<float*>(epi.out) + (int64_t)row * epi.ldo + nb;
                float4 xr[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) xr[j] = *reinterpret_cast<const float4*>(orow + 16 * j);
                RESID_RMW(xr, i, orow)
            }
        }
#undef RESID_RMW
#endif
    } else if (PAIR) {
        const int nb = ecol0 + wn * 64 + 8 * fg;
#define PAIR_STORE(I, OROW)                                                                                                         \
        _Pragma("unroll") for (int j2 = 0; j2 < 2; ++j2) {                                                                          \
            float v[8];                                                                                                             \
            _Pragma("unroll") for (int t = 0; t < 8; t += 2) {                                                                      \
                f32x2 x = {acc[I][2 * j2 + (t >> 2)][t & 3], acc[I][2 * j2 + (t >> 2)][(t & 3) + 1]};                               \
                x += f32x2{bb[j2][t], bb[j2][t + 1]};                                                                               \
                if (GELU) x = gelu_erf2(x);                                                                                         \
                v[t] = x.x; v[t + 1] = x.y;                                                                                         \
            }                                                                                                                       \
            const uint4 hi4 = make_uint4(f2h_pk(v[0], v[1]), f2h_pk(v[2], v[3]), f2h_pk(v[4], v[5]), f2h_pk(v[6], v[7]));           \
            *reinterpret_cast<uint4*>((OROW) + 32 * j2) = hi4;                                                                      \
            if (GT > 1) {                                                                                                           \
                const unsigned hw_[4] = {hi4.x, hi4.y, hi4.z, hi4.w};                                                               \
                unsigned lw_[4], sw_[4];                                                                                            \
                _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                                     \
                    const float h0 = h2f((u16)(hw_[t] & 0xFFFFu)), h1 = h2f((u16)(hw_[t] >> 16));                                   \
                    sw_[t] = f2h_pk(h0 * (1.0f / IBL_VIT_SPLIT_SCALE), h1 * (1.0f / IBL_VIT_SPLIT_SCALE));                          \
                    lw_[t] = f2h_pk((v[2 * t] - h0) * IBL_VIT_SPLIT_SCALE, (v[2 * t + 1] - h1) * IBL_VIT_SPLIT_SCALE);              \
                }                                                                                                                   \
                if (GT == 3) *reinterpret_cast<uint4*>((OROW) + N + 32 * j2) = make_uint4(lw_[0], lw_[1], lw_[2], lw_[3]);          \
                *reinterpret_cast<uint4*>((OROW) + (GT - 1) * (int64_t)N + 32 * j2) = make_uint4(sw_[0], sw_[1], sw_[2], sw_[3]);   \
            }                                                                                                                       \
        }
        if (erow0 + BM <= M) {                 // full tile (all but the last row of tiles): no per-row branch between the stores
            u16* const obase = reinterpret_cast<u16*>(epi.out) + (int64_t)(erow0 + wm * (MI * 16) + fr) * epi.ldo + nb;
            const int64_t gstride = 16 * epi.ldo;
#pragma unroll
            for (int i = 0; i < MI; ++i) { PAIR_STORE(i, obase + i * gstride) }
            stores_pending = true;
        } else {
            stores_pending = false;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = erow0 + wm * (MI * 16) + i * 16 + fr;
                if (row >= M) continue;
                u16* orow = reinterpret_cast<u16*>(epi.out) + (int64_t)row * epi.ldo + nb;
                PAIR_STORE(i, orow)
            }
        }
#undef PAIR_STORE
    } else {
    const int n0 = ecol0 + wn * 64 + 16 * fg;          // first of this lane's 16 consecutive columns
    float bias[16], scale[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), s4 = make_float4(1.f, 1.f, 1.f, 1.f);
        if (epi.bias) b4 = *reinterpret_cast<const float4*>(epi.bias + n0 + 4 * q);
        if (EPI == EPI_RESID_F32 && epi.scale) s4 = *reinterpret_cast<const float4*>(epi.scale + n0 + 4 * q);
        bias[4 * q] = b4.x; bias[4 * q + 1] = b4.y; bias[4 * q + 2] = b4.z; bias[4 * q + 3] = b4.w;
        scale[4 * q] = s4.x; scale[4 * q + 1] = s4.y; scale[4 * q + 2] = s4.z; scale[4 * q + 3] = s4.w;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int row = erow0 + wm * (MI * 16) + i * 16 + fr;
        if (row >= M) continue;
        float v[16];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * j + r] = acc[i][j][r] + bias[4 * j + r];
        if (PAIR) {
            if (GELU) {
#pragma unroll
                for (int t = 0; t < 16; ++t) v[t] = gelu_erf(v[t]);
            }
            unsigned int pk[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) pk[t] = (unsigned int)f2h(v[2 * t]) | ((unsigned int)f2h(v[2 * t + 1]) << 16);
            uint4* o = reinterpret_cast<uint4*>(reinterpret_cast<u16*>(epi.out) + (int64_t)row * epi.ldo + n0);
            o[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            o[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
            if (GT > 1) {
                unsigned int lk[8], sk[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const float h0 = h2f((u16)(pk[t] & 0xFFFFu)), h1 = h2f((u16)(pk[t] >> 16));
                    sk[t] = (unsigned int)f2h(h0 * (1.0f / IBL_VIT_SPLIT_SCALE)) | ((unsigned int)f2h(h1 * (1.0f / IBL_VIT_SPLIT_SCALE)) << 16);
                    lk[t] = (unsigned int)f2h((v[2 * t] - h0) * IBL_VIT_SPLIT_SCALE) | ((unsigned int)f2h((v[2 * t + 1] - h1) * IBL_VIT_SPLIT_SCALE) << 16);
                }
                if (GT == 3) {
                    o[N / 8] = make_uint4(lk[0], lk[1], lk[2], lk[3]);
                    o[N / 8 + 1] = make_uint4(lk[4], lk[5], lk[6], lk[7]);
                }
                o[(GT - 1) * (N / 8)] = make_uint4(sk[0], sk[1], sk[2], sk[3]);
                o[(GT - 1) * (N / 8) + 1] = make_uint4(sk[4], sk[5], sk[6], sk[7]);
            }
        } else if (EPI == EPI_RESID_F32) {
            float4* o = reinterpret_cast<float4*>(reinterpret_cast<float*>(epi.out) + (int64_t)row * epi.ldo + n0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 x = o[q];
                x.x += v[4 * q] * scale[4 * q]; x.y += v[4 * q + 1] * scale[4 * q + 1];
                x.z += v[4 * q + 2] * scale[4 * q + 2]; x.w += v[4 * q + 3] * scale[4 * q + 3];
                o[q] = x;
            }
        } else if (EPI == EPI_PATCH_F32) {
            const int b = row / epi.patches_per_crop, p = row - b * epi.patches_per_crop;
            const int64_t orow = (int64_t)b * epi.tokens_per_crop + 1 + p;
            float4* o = reinterpret_cast<float4*>(reinterpret_cast<float*>(epi.out) + orow * epi.ldo + n0);
            if (epi.accumulate) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 xx = o[q];
                    xx.x = fmaf(v[4 * q], epi.alpha, xx.x); xx.y = fmaf(v[4 * q + 1], epi.alpha, xx.y);
                    xx.z = fmaf(v[4 * q + 2], epi.alpha, xx.z); xx.w = fmaf(v[4 * q + 3], epi.alpha, xx.w);
                    o[q] = xx;
                }
            } else {
                const float4* ps = reinterpret_cast<const float4*>(epi.pos + (int64_t)p * N + n0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 pp = ps[q];
                    o[q] = make_float4(v[4 * q] + pp.x, v[4 * q + 1] + pp.y, v[4 * q + 2] + pp.z, v[4 * q + 3] + pp.w);
                }
            }
        } else {
            float4* o = reinterpret_cast<float4*>(reinterpret_cast<float*>(epi.out) + (int64_t)row * epi.ldo + n0);
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        }
    }
    }
    GEMM_STAMP(3);                 // epilogue issued (its stores drain under the next tile's top wait)
#ifdef IBL_GEMM_STAMPS
    ++stamp_it;
#endif
    if (!has_next) break;
    first_tile = false;
    }
#undef GEMM_PROLOGUE
#undef GEMM_SLOT
#undef GEMM_PIECE
#undef FRAG_W
#undef FRAG_A
#undef GEMM_GLDS
#undef GEMM_SET_TILE
#endif
}

template <int EPI, int MI, int WM, int WN, int BK, int OCC, int NS, bool PIPE = false>
static int launch_gemm_cfg(const u16* A, int64_t lda, const u16* W, int64_t ldw, int M, int N, int K, const GemmEpi& epi, hipStream_t s) {
    constexpr int BM = WM * MI * 16, BN = WN * 64;
    const int nwg = (N / BN) * ((M + BM - 1) / BM);
    const size_t lds = NS * (size_t)(BM + BN) * BK * 2 + (PIPE ? (size_t)WM * WN * 512 : 0);     // (+ the landing area of the L2 touches, 512 B per wave)
    // the dynamic-LDS limit is a per-device property of the function: set once per device of this process.  The bit mask is only
    // a cache -- two threads racing here both set the same value (localise_concurrent lanes launch from several host threads)
    static std::atomic<unsigned long long> attr_set{0};
    int dev = 0;
    IBL_HIP_CHECK(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_set.load(std::memory_order_acquire) & bit)) {
        IBL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ibl_gemm_f16_tn<EPI, MI, WM, WN, BK, OCC, NS, PIPE>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set.fetch_or(bit, std::memory_order_release);
    }
    int grid = nwg;
#ifndef IBL_GEMM_PERSIST
#define IBL_GEMM_PERSIST 1
#endif
    if (PIPE && IBL_GEMM_PERSIST) {          // persistent: one workgroup per CU (a multiple of 8, see GEMM_SET_TILE)
        static std::atomic<int> n_cu{0};
        int cus = n_cu.load(std::memory_order_relaxed);
        if (cus == 0) {
            IBL_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
            cus = cus >= 8 ? cus / 8 * 8 : 8;
            n_cu.store(cus, std::memory_order_relaxed);
        }
        if (grid > cus * OCC) grid = cus * OCC;
        static int cap = -2;                 // lab: IBL_GEMM_MAXGRID caps the persistent grid (is a per-tile phase bound by the chip or by the CU?)
        if (cap == -2) { const char* e = getenv("IBL_GEMM_MAXGRID"); cap = e ? atoi(e) : 0; }
        if (cap > 0 && grid > cap) grid = cap;
    }
    GemmEpi e2 = epi;
    if (PIPE && grid < nwg) {
        // lab only (IBL_GEMM_STAGGER = s_sleep(64) units per phase group): measured and rejected in round 3 -- proj 125 / 142 / 155 / 171 us
        // and fc2 304 / 311 / 322 / 341 us at 0 / 3 / 6 / 10: the epilogue is not slowed by the other workgroups' epilogues, the delay is
        // pure tail
        static int stag = -2;
        if (stag == -2) { const char* e = getenv("IBL_GEMM_STAGGER"); stag = e ? atoi(e) : 0; }
        e2.stagger = stag;
    }
    void* tok;
    ibl_prof_begin(IBL_PROF_GEMM, 2.0 * (double)M * (double)N * (double)(epi.algo_k < 0 ? 0 : (epi.algo_k ? epi.algo_k : K)), s, &tok);
    hipLaunchKernelGGL((ibl_gemm_f16_tn<EPI, MI, WM, WN, BK, OCC, NS, PIPE>), dim3(grid), dim3(WM * WN * 64), lds, s, A, lda, W, ldw, M, N, K, e2);
    ibl_prof_end(tok, s);
    IBL_LAUNCH_CHECK();
    return IBL_OK;
}

// Tile configurations (IBL_GEMM_CFG overrides the choice; the lab build -DIBL_GEMM_LAB adds the variants measured and rejected):
//   8  256 x 256, BK 64, 8 waves, 128 KiB LDS, software-pipelined K loop, persistent grid      (default for N % 256 == 0, M >= 4096)
//   0  256 x 256, BK 64, 8 waves, 128 KiB LDS, plain K loop, one block per tile
//   1  128 x 128, BK 64, 4 waves,  64 KiB LDS, 2 blocks / CU  (any N % 128 == 0, small M)
//   2  256 x 128, BK 32, 4 waves,  48 KiB LDS, 2 blocks / CU  (the epilogue of one block overlaps the K loop of the other)
//   3  256 x 256, BK 32, 8 waves, 4 LDS stages = 128 KiB      (loads two K steps ahead, no full drain at the step boundary)
// Measured on the four ViT-B/14 layer shapes (57 568 rows), TFLOP/s per layer, one process: 8: 783, 8 without the persistent
// grid: 775, 0: 745, 3: 649, 2: 632, 1: 629; lab: 256 x 128 BK 32 three stages 643, 128 x 256 BK 32 639, 128 x 128 pipelined 620.
static int gemm_cfg_override() {
    static int v = -2;
    if (v == -2) {
        const char* e = getenv("IBL_GEMM_CFG");
        v = e ? atoi(e) : -1;
    }
    return v;
}

// EPI_RESID_PRE_F32 (residual tile preloaded into the accumulators, store-only epilogue) is a lab switch: measured per tile with
// tools/perf_gemm.py --stamps, the preload of the 256 KB tile at the tile top costs what the read-modify-write behind the K loop costs
// (30 k vs 38 - 12 k clocks: a CU sustains ~8.5 B / clock of L2-missing loads whatever their arrangement, and the count does not change
// with 32 instead of 256 active CUs), end to end proj 131 -> 138 us, fc2 298 -> 309 us.  IBL_GEMM_RESID_PRE=1 selects it where no scale
// vector is given.
static bool gemm_resid_pre() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("IBL_GEMM_RESID_PRE"); v = (e && atoi(e)) ? 1 : 0; }
    return v == 1;
}

template <int EPI>
static int launch_gemm(const u16* A, int64_t lda, const u16* W, int64_t ldw, int M, int N, int K, const GemmEpi& epi,
                       hipStream_t s) {
    if (M <= 0) return IBL_OK;
    if (N % 128 != 0 || K % GBK != 0)
        return ibl_set_error(IBL_ERR_ARG, "gemm: N (%d) must be a multiple of 128 and K (%d) of 64", N, K);
    int cfg = (N % 256 == 0 && M >= 4096) ? 8 : 1;
    const int ov = gemm_cfg_override();
    if (ov == 1 || ov == 2 || ((ov == 0 || ov == 3 || ov == 8) && N % 256 == 0)) cfg = ov;
#ifdef IBL_GEMM_LAB
    if (ov == 4 || ov == 6 || ov == 7 || ov == 9 || (ov == 5 && N % 256 == 0)) cfg = ov;
    if (cfg == 9) return launch_gemm_cfg<EPI, 4, 2, 2, 64, 2, 2, true>(A, lda, W, ldw, M, N, K, epi, s);
    if (cfg == 4) return launch_gemm_cfg<EPI, 8, 2, 2, 32, 2, 3>(A, lda, W, ldw, M, N, K, epi, s);
    if (cfg == 5) return launch_gemm_cfg<EPI, 4, 2, 4, 32, 4, 3>(A, lda, W, ldw, M, N, K, epi, s);
    if (cfg == 6) return launch_gemm_cfg<EPI, 4, 2, 2, 32, 2, 4>(A, lda, W, ldw, M, N, K, epi, s);
    if (cfg == 7) return launch_gemm_cfg<EPI, 4, 2, 2, 32, 3, 3>(A, lda, W, ldw, M, N, K, epi, s);
#endif
#ifdef IBL_GEMM_FORCE128
    cfg = 1;
#endif
    if (cfg == 8) return launch_gemm_cfg<EPI, 8, 2, 4, 64, 1, 2, true>(A, lda, W, ldw, M, N, K, epi, s);
    if (cfg == 0) return launch_gemm_cfg<EPI, 8, 2, 4, 64, 1, 2>(A, lda, W, ldw, M, N, K, epi, s);
    if (cfg == 2) return launch_gemm_cfg<EPI, 8, 2, 2, 32, 2, 2>(A, lda, W, ldw, M, N, K, epi, s);
    if (cfg == 3) return launch_gemm_cfg<EPI, 8, 2, 4, 32, 1, 4>(A, lda, W, ldw, M, N, K, epi, s);
    return launch_gemm_cfg<EPI, 4, 2, 2, 64, 2, 2>(A, lda, W, ldw, M, N, K, epi, s);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, fp32 in -> fp16 out (or fp32 out for the final CLS rows)
// ------------------------------------------------------------------------------------------------
// TERMS (fp16 output only): 1 = the row; 2 = [a_hi | a_hi / S]; 3 = [a_hi | a_lo * S | a_hi / S] (two-term operand rows, ibloc.h)
template <bool OUT_F32, int TERMS = 1>
__global__ __launch_bounds__(256) void ibl_layernorm_kernel(const float* x, int64_t in_row_stride,
                                                            int64_t n_rows, int dim, const float* __restrict__ g,
                                                            const float* __restrict__ b, float eps, void* out,
                                                            int64_t out_row_stride) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const float* xr = x + row * in_row_stride;
    constexpr int MAXV = 4;                 // dim <= 1024
    float4 v[MAXV];
    const int nv = dim / 256;               // float4 per lane (dim % 256 == 0 handled; remainder below)
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < nv) {
            v[i] = *reinterpret_cast<const float4*>(xr + (i * 64 + lane) * 4);
            s += v[i].x + v[i].y + v[i].z + v[i].w;
        }
    const int rem0 = nv * 256;
    float tail[4] = {0.f, 0.f, 0.f, 0.f};     // up to 255 remaining elements, 4 per lane
    int ntail = 0;
    for (int i = rem0 + lane; i < dim; i += 64) { tail[ntail] = xr[i]; s += tail[ntail]; ++ntail; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < nv) {
            const float a = v[i].x - mean, c = v[i].y - mean, d = v[i].z - mean, e = v[i].w - mean;
            q += a * a + c * c + d * d + e * e;
        }
    for (int t = 0; t < ntail; ++t) { const float a = tail[t] - mean; q += a * a; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off, 64);
    const float rstd = rsqrtf(q / (float)dim + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < nv) {
            const int c0 = (i * 64 + lane) * 4;
            const float4 gg = *reinterpret_cast<const float4*>(g + c0);
            const float4 bb = *reinterpret_cast<const float4*>(b + c0);
            const float y0 = (v[i].x - mean) * rstd * gg.x + bb.x;
            const float y1 = (v[i].y - mean) * rstd * gg.y + bb.y;
            const float y2 = (v[i].z - mean) * rstd * gg.z + bb.z;
            const float y3 = (v[i].w - mean) * rstd * gg.w + bb.w;
            if (OUT_F32) {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + row * out_row_stride + c0) =
                    make_float4(y0, y1, y2, y3);
            } else {
                ushort4 o;
                o.x = f2h(y0); o.y = f2h(y1); o.z = f2h(y2); o.w = f2h(y3);
                u16* orow = reinterpret_cast<u16*>(out) + row * out_row_stride;
                *reinterpret_cast<ushort4*>(orow + c0) = o;
                if (TERMS > 1) {
                    const float h0 = (float)__builtin_bit_cast(_Float16, o.x), h1 = (float)__builtin_bit_cast(_Float16, o.y);
                    const float h2 = (float)__builtin_bit_cast(_Float16, o.z), h3 = (float)__builtin_bit_cast(_Float16, o.w);
                    constexpr float S = IBL_VIT_SPLIT_SCALE, IS = 1.0f / IBL_VIT_SPLIT_SCALE;
                    ushort4 d;
                    d.x = f2h(h0 * IS); d.y = f2h(h1 * IS); d.z = f2h(h2 * IS); d.w = f2h(h3 * IS);
                    *reinterpret_cast<ushort4*>(orow + (TERMS - 1) * dim + c0) = d;
                    if (TERMS == 3) {
                        ushort4 l;
                        l.x = f2h((y0 - h0) * S); l.y = f2h((y1 - h1) * S); l.z = f2h((y2 - h2) * S); l.w = f2h((y3 - h3) * S);
                        *reinterpret_cast<ushort4*>(orow + dim + c0) = l;
                    }
                }
            }
        }
    int t = 0;
    for (int i = rem0 + lane; i < dim; i += 64, ++t) {
        const float y = (tail[t] - mean) * rstd * g[i] + b[i];
        if (OUT_F32) reinterpret_cast<float*>(out)[row * out_row_stride + i] = y;
        else {
            u16* orow = reinterpret_cast<u16*>(out) + row * out_row_stride;
            const u16 hb = f2h(y);
            orow[i] = hb;
            if (TERMS > 1) {
                const float hf = (float)__builtin_bit_cast(_Float16, hb);
                orow[(TERMS - 1) * dim + i] = f2h(hf * (1.0f / IBL_VIT_SPLIT_SCALE));
                if (TERMS == 3) orow[dim + i] = f2h((y - hf) * IBL_VIT_SPLIT_SCALE);
            }
        }
    }
}

// x[b*T + 0][:] = cls_pos[:]   (cls token + its position embedding)
__global__ void ibl_set_cls_kernel(float* __restrict__ x, const float* __restrict__ cls_pos, int B, int T, int dim) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * dim) return;
    const int b = (int)(i / dim), c = (int)(i % dim);
    x[(int64_t)b * T * dim + c] = cls_pos[c];
}

// ------------------------------------------------------------------------------------------------
// Attention: one workgroup (4 waves) per (crop, head); head_dim = 64; T <= 16 * NT.
// qkv fp16 [B*T][3*D] = [q | k | v], each [heads][64].   out fp16 [B*T][D].
// ------------------------------------------------------------------------------------------------
#ifndef ATT_THREADS
#define ATT_THREADS 512
#endif
// TERMS: the output as K-extended operand rows for the projection (row stride TERMS * D; round 4): 2 = [a | a / S] (projection weights
// in two terms), 3 = [a | (value - a) * S | a / S] (+ the attention output's own second term)
template <int NT, int TERMS = 1>
__global__ __launch_bounds__(ATT_THREADS, 2) void ibl_attention_kernel(const u16* __restrict__ qkv, u16* __restrict__ out, int T,
                                                            int D, int heads, float scale, int cls_only) {
    constexpr int KEYS = NT * 16;
    constexpr int KROW = 144;               // bytes per K row (64 fp16 + 16 B pad)
    constexpr int VROW = KEYS * 2 + 16;     // bytes per V^T row
    __shared__ __attribute__((aligned(16))) unsigned char sK[KEYS * KROW];
    __shared__ __attribute__((aligned(16))) unsigned char sV[64 * VROW];
#ifdef ATT_LAB_PAD_LDS       // lab: pad the LDS so that one workgroup fits a CU (is the kernel bound by its resident workgroups?)
    __shared__ int lab_pad[ATT_LAB_PAD_LDS / 4];
    if (threadIdx.x == 0 && T == -12345) lab_pad[D] = 1;
#endif

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int64_t tok0 = (int64_t)b * T;
    const int64_t ld = 3 * (int64_t)D;
    const u16* qbase = qkv + tok0 * ld + h * 64;
    const u16* kbase = qbase + D;
    const u16* vbase = qbase + 2 * D;

    // stage K rows (zero beyond T)
    for (int c = tid; c < KEYS * 8; c += ATT_THREADS) {
        const int key = c >> 3, c16 = c & 7;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (key < T) val = *reinterpret_cast<const uint4*>(kbase + (int64_t)key * ld + c16 * 8);
        *reinterpret_cast<uint4*>(sK + key * KROW + c16 * 16) = val;
    }
    // stage V transposed: task = (key pair, 8-wide d chunk); writes packed dwords V^T[d][key..key+1]
    for (int c = tid; c < (KEYS / 2) * 8; c += ATT_THREADS) {
        const int kp = c % (KEYS / 2), dch = c / (KEYS / 2);
        const int key = kp * 2;
        uint4 v0 = make_uint4(0, 0, 0, 0), v1 = make_uint4(0, 0, 0, 0);
        if (key < T) v0 = *reinterpret_cast<const uint4*>(vbase + (int64_t)key * ld + dch * 8);
        if (key + 1 < T) v1 = *reinterpret_cast<const uint4*>(vbase + (int64_t)(key + 1) * ld + dch * 8);
        const unsigned int a[4] = {v0.x, v0.y, v0.z, v0.w}, d[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned int lo = (a[i] & 0xFFFFu) | (d[i] << 16);            // element 2i of both keys
            const unsigned int hi = (a[i] >> 16) | (d[i] & 0xFFFF0000u);        // element 2i+1
            *reinterpret_cast<unsigned int*>(sV + (dch * 8 + 2 * i) * VROW + key * 2) = lo;
            *reinterpret_cast<unsigned int*>(sV + (dch * 8 + 2 * i + 1) * VROW + key * 2) = hi;
        }
    }
    __syncthreads();

    const int fr = lane & 15, fg = lane >> 4;
    const int nqt = cls_only ? 1 : (T + 15) / 16;        // cls_only: only token 0 of every crop is a query (last block of the ViT)
    for (int qt = wave; qt < nqt; qt += ATT_THREADS / 64) {
        // B operand = Q^T: lane (q = fr, g): Q[q0 + fr][32 ks + 8 g .. +7]
        int qrow = qt * 16 + fr;
        if (qrow >= T) qrow = T - 1;
        const u16* qp = qbase + (int64_t)qrow * ld + fg * 8;
        const h16x8 qf0 = *reinterpret_cast<const h16x8*>(qp);
        const h16x8 qf1 = *reinterpret_cast<const h16x8*>(qp + 32);

        // S^T tiles: acc[t][r] = S[q = fr][key = 16 t + 4 g + r]
        f32x4 sc[NT];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const unsigned char* kp = sK + (t * 16 + fr) * KROW + fg * 16;
            const h16x8 k0 = *reinterpret_cast<const h16x8*>(kp);
            const h16x8 k1 = *reinterpret_cast<const h16x8*>(kp + 64);
            f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
            a = IBL_MFMA(k0, qf0, a, 0, 0, 0);
            a = IBL_MFMA(k1, qf1, a, 0, 0, 0);
            // raw scores: the softmax scale is folded into the exponent below (scale > 0: the maximum commutes); only a tile that
            // reaches past T needs the key mask (the softmax arithmetic, not the MFMAs, bounds this kernel: ~10 VALU per score before)
            if (t * 16 + 16 > T) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (t * 16 + fg * 4 + r >= T) a[r] = -INFINITY;
            }
            mx = fmaxf(mx, fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])));
            sc[t] = a;
            // keep the K fragments of later tiles from being hoisted up here: unrolled, the 34 ds_read_b128 of a query tile were
            // all issued first (398 VGPRs -> one block per CU); with two blocks per CU the other block hides this latency
            asm volatile("" ::: "memory");
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
        const float c2 = scale * 1.4426950408889634f, mc = -mx * c2;       // exp(scale (s - mx)) = exp2(s c2 - mx c2)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(sc[t][r], c2, mc));
                sc[t][r] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);

        // O = P V : A operand (P) element j of lane (q = fr, g) for k-step s is key
        //   32 s + 4 g + j (j < 4, tile 2s)   or   32 s + 16 + 4 g + (j - 4) (tile 2s + 1);
        // B operand (V) uses the same key permutation, read from V^T rows.
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int NS = (NT + 1) / 2;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            h16x8 pf;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pf[j] = (_Float16)sc[2 * s][j];
                pf[4 + j] = (2 * s + 1 < NT) ? (_Float16)sc[(2 * s + 1 < NT) ? 2 * s + 1 : 0][j] : (_Float16)0.0f;
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const unsigned char* vp = sV + (dt * 16 + fr) * VROW + (32 * s + 4 * fg) * 2;
                const uint2 lo = *reinterpret_cast<const uint2*>(vp);
                uint2 hi = make_uint2(0, 0);
                if (2 * s + 1 < NT) hi = *reinterpret_cast<const uint2*>(vp + 32);
                uint4 packed = make_uint4(lo.x, lo.y, hi.x, hi.y);
                const h16x8 vf = *reinterpret_cast<const h16x8*>(&packed);
                o[dt] = IBL_MFMA(vf, pf, o[dt], 0, 0, 0);     // O^T tile: rows = d, columns = q
            }
            asm volatile("" ::: "memory");
        }
        // O^T layout: row = d = 16 dt + 4 g + r, col = q = fr -> a lane owns four consecutive head dimensions of ONE query row
        // (8-byte stores; the untransposed product left it with single fp16 elements of four rows) and that row's softmax
        // sum is already on this lane (it was reduced over the lane groups above).
        const int qg = qt * 16 + fr;
        if (qg < (cls_only ? 1 : T)) {
            const float inv = 1.0f / sum;
            u16* orow = out + (tok0 + qg) * (int64_t)(TERMS * D) + h * 64 + 4 * fg;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                float v[4];
                unsigned short h4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = o[dt][r] * inv; h4[r] = f2h(v[r]); }
                uint2 pk;
                pk.x = (unsigned)h4[0] | ((unsigned)h4[1] << 16);
                pk.y = (unsigned)h4[2] | ((unsigned)h4[3] << 16);
                *reinterpret_cast<uint2*>(orow + dt * 16) = pk;
                if (TERMS > 1) {
                    unsigned short l4[4], s4[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float hf = h2f(h4[r]);
                        s4[r] = f2h(hf * (1.0f / IBL_VIT_SPLIT_SCALE));
                        l4[r] = f2h((v[r] - hf) * IBL_VIT_SPLIT_SCALE);
                    }
                    if (TERMS == 3)
                        *reinterpret_cast<uint2*>(orow + D + dt * 16) = make_uint2((unsigned)l4[0] | ((unsigned)l4[1] << 16), (unsigned)l4[2] | ((unsigned)l4[3] << 16));
                    *reinterpret_cast<uint2*>(orow + (TERMS - 1) * D + dt * 16) =
                        make_uint2((unsigned)s4[0] | ((unsigned)s4[1] << 16), (unsigned)s4[2] | ((unsigned)s4[3] << 16));
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
static inline int64_t rows_pad(int64_t r) { return ibl_align_up(r, 128); }

extern "C" int ibl_linear_f16(const void* x, int64_t ldx, const void* W, int64_t ldw, const float* bias, const float* scale,
                               int64_t rows, int n_out, int n_in, int epilogue, void* out, int64_t ldo, void* stream) {
    if (rows == 0) return IBL_OK;
    if (!x || !W || !out) return ibl_set_error(IBL_ERR_ARG, "ibl_linear_f16: null operand");
    if (rows < 0 || rows > 0x7fffffff) return ibl_set_error(IBL_ERR_ARG, "ibl_linear_f16: rows out of range");
    if ((ldx & 7) || (ldw & 7) || ldx < n_in || ldw < n_in || ldo < n_out || (ldo & 7))
        return ibl_set_error(IBL_ERR_ARG, "ibl_linear_f16: row strides must be >= the row length and multiples of 8 elements");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    GemmEpi e{};
    e.bias = bias;
    e.scale = scale;
    e.out = out;
    e.ldo = ldo;
    const u16* a = reinterpret_cast<const u16*>(x);
    const u16* w = reinterpret_cast<const u16*>(W);
    switch (epilogue) {
        case EPI_BIAS_H16: return launch_gemm<EPI_BIAS_H16>(a, ldx, w, ldw, (int)rows, n_out, n_in, e, s);
        case EPI_BIAS_GELU_H16: return launch_gemm<EPI_BIAS_GELU_H16>(a, ldx, w, ldw, (int)rows, n_out, n_in, e, s);
        case EPI_RESID_F32:
            if (!e.scale && gemm_resid_pre()) { e.alpha = 1.0f; return launch_gemm<EPI_RESID_PRE_F32>(a, ldx, w, ldw, (int)rows, n_out, n_in, e, s); }
            return launch_gemm<EPI_RESID_F32>(a, ldx, w, ldw, (int)rows, n_out, n_in, e, s);
        case EPI_BIAS_F32: return launch_gemm<EPI_BIAS_F32>(a, ldx, w, ldw, (int)rows, n_out, n_in, e, s);
        default: return ibl_set_error(IBL_ERR_ARG, "ibl_linear_f16: unknown epilogue %d", epilogue);
    }
}

extern "C" int64_t ibl_vit_workspace_bytes(const ibl_vit_desc* d, int batch) {
    if (!d || batch <= 0) return -1;
    const int64_t R = rows_pad((int64_t)batch * d->n_tokens);
    int64_t bytes = 0;
    bytes += R * d->dim * 4;          // x
    bytes += R * d->dim * 2 * 3;      // xn (up to three terms per row, two-term operands of the early blocks)
    bytes += R * 3 * d->dim * 2;      // qkv
    const int at = (d->flags & IBL_VIT_ACT_TERMS3) ? 3 : ((d->flags & IBL_VIT_ACT_TERMS2) ? 2 : 1);   // widest K-extended input of a residual GEMM
    bytes += R * d->dim * 2 * at;     // attn out
    bytes += R * d->mlp_dim * 2 * at; // mlp hidden
    bytes += rows_pad(batch) * d->dim * 2 + rows_pad(batch) * d->dim * 4;  // final rows (fp16 + f32)
    return bytes + 9 * 256;
}

static int run_attention(const u16* qkv, u16* out, int B, int T, int D, int heads, int cls_only, hipStream_t s, int terms = 1) {
    const float scale = 0.125f;   // 1/sqrt(64)
    const int nt = (T + 15) / 16;
    dim3 grid(B * heads), block(ATT_THREADS);
#define IBL_ATT(NTV)                                                                                      \
    do {                                                                                                  \
        if (terms == 3) hipLaunchKernelGGL((ibl_attention_kernel<NTV, 3>), grid, block, 0, s, qkv, out, T, D, heads, scale, cls_only);        \
        else if (terms == 2) hipLaunchKernelGGL((ibl_attention_kernel<NTV, 2>), grid, block, 0, s, qkv, out, T, D, heads, scale, cls_only);   \
        else hipLaunchKernelGGL((ibl_attention_kernel<NTV, 1>), grid, block, 0, s, qkv, out, T, D, heads, scale, cls_only);                    \
    } while (0)
    if (getenv("IBL_DEBUG_OCC")) {
        int nb = -1;
        hipFuncAttributes fa{};
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ibl_attention_kernel<17, 1>, ATT_THREADS, 0);
        (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&ibl_attention_kernel<17, 1>));
        fprintf(stderr, "[occ] attention<17>: %d blocks/CU, %d regs, %zu B static LDS\n", nb, fa.numRegs, fa.sharedSizeBytes);
    }
    if (nt <= 4) IBL_ATT(4);
    else if (nt <= 9) IBL_ATT(9);
    else if (nt <= 13) IBL_ATT(13);
    else if (nt <= 17) IBL_ATT(17);
    else return ibl_set_error(IBL_ERR_UNSUPPORTED, "attention: n_tokens %d > 272 not supported", T);
#undef IBL_ATT
    IBL_LAUNCH_CHECK();
    return IBL_OK;
}

extern "C" int ibl_vit_forward(const ibl_vit_desc* d, const ibl_vit_weights* w, const void* patches, int batch,
                               float* out, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!d || !w || !patches || !out || !workspace) return ibl_set_error(IBL_ERR_ARG, "ibl_vit_forward: null pointer");
    if (batch <= 0) return ibl_set_error(IBL_ERR_ARG, "ibl_vit_forward: batch must be positive");
    const int D = d->dim, T = d->n_tokens, P = T - 1, H = d->heads;
    if (D != H * 64) return ibl_set_error(IBL_ERR_UNSUPPORTED, "ibl_vit_forward: head_dim must be 64");
    if (D % 128 || d->mlp_dim % 128 || d->patch_k_pad % 64 || D > 1024)
        return ibl_set_error(IBL_ERR_UNSUPPORTED, "ibl_vit_forward: dim/mlp_dim %% 128, patch_k_pad %% 64, dim <= 1024");
    if (d->n_blocks_run < 0 || d->n_blocks_run > d->depth || d->depth > IBL_VIT_MAX_LAYERS)
        return ibl_set_error(IBL_ERR_ARG, "ibl_vit_forward: bad depth");
    if (workspace_bytes < ibl_vit_workspace_bytes(d, batch))
        return ibl_set_error(IBL_ERR_ARG, "ibl_vit_forward: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int64_t Rn = (int64_t)batch * T, R = rows_pad(Rn);

    auto carve = [](unsigned char*& p, int64_t bytes) {
        p = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(p) + 255) & ~(uintptr_t)255);
        unsigned char* r = p;
        p += bytes;
        return r;
    };
    unsigned char* p = reinterpret_cast<unsigned char*>(workspace);
    float* x = reinterpret_cast<float*>(carve(p, R * D * 4));
    u16* xn = reinterpret_cast<u16*>(carve(p, R * D * 2 * 3));
    u16* qkv = reinterpret_cast<u16*>(carve(p, R * 3 * D * 2));
    const int at = (d->flags & IBL_VIT_ACT_TERMS3) ? 3 : ((d->flags & IBL_VIT_ACT_TERMS2) ? 2 : 1);
    u16* att = reinterpret_cast<u16*>(carve(p, R * D * 2 * at));
    u16* hid = reinterpret_cast<u16*>(carve(p, R * d->mlp_dim * 2 * at));
    u16* fin_bf = reinterpret_cast<u16*>(carve(p, rows_pad(batch) * D * 2));

    int st;
    // patch embedding (conv as GEMM over im2col'ed patches) + position embedding, scattered into x
    {
        GemmEpi e{};
        e.bias = w->b_patch; e.pos = w->pos_patch; e.out = x; e.ldo = D; e.tokens_per_crop = T; e.patches_per_crop = P;
        st = launch_gemm<EPI_PATCH_F32>(reinterpret_cast<const u16*>(patches), d->patch_k_pad,
                                        reinterpret_cast<const u16*>(w->w_patch), d->patch_k_pad, batch * P, D,
                                        d->patch_k_pad, e, s);
        if (st) return st;
        if (w->w_patch_lo) {             // second term of the patch weights: x += (patches W_lo'^T) / S
            GemmEpi e2 = e;
            e2.bias = nullptr; e2.accumulate = 1; e2.alpha = 1.0f / IBL_VIT_SPLIT_SCALE; e2.algo_k = -1;
            st = launch_gemm<EPI_PATCH_F32>(reinterpret_cast<const u16*>(patches), d->patch_k_pad,
                                            reinterpret_cast<const u16*>(w->w_patch_lo), d->patch_k_pad, batch * P, D,
                                            d->patch_k_pad, e2, s);
            if (st) return st;
        }
        const int64_t n = (int64_t)batch * D;
        hipLaunchKernelGGL(ibl_set_cls_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, w->cls_pos, batch,
                           T, D);
        IBL_LAUNCH_CHECK();
    }
    const dim3 ln_grid((unsigned)((Rn + 3) / 4));
    if (d->flags & IBL_VIT_PRE_LN) {
        // CLIP ln_pre: normalise the residual stream in place (through the fp16-free f32 path)
        hipLaunchKernelGGL(ibl_layernorm_kernel<true>, ln_grid, dim3(256), 0, s, x, (int64_t)D, Rn, D, w->ln_pre_g,
                           w->ln_pre_b, d->ln_eps, (void*)x, (int64_t)D);
        IBL_LAUNCH_CHECK();
    }
    for (int l = 0; l < d->n_blocks_run; ++l) {
        const ibl_vit_layer* L = &w->layers[l];
        // Only the CLS row leaves the encoder (unless all tokens are asked for), so in the LAST block the other rows are dead
        // after they have served as keys / values: its query projection, attention, output projection and MLP run on the CLS
        // rows alone (M = batch instead of batch * T; every row of a GEMM is computed independently, so the CLS rows come out
        // bit-identical).  The kernels take row strides, the CLS rows are addressed in place (stride T * D).
        const bool cls_only = (l == d->n_blocks_run - 1) && !(d->flags & IBL_VIT_OUT_ALL_TOKENS);
        const int64_t TD = (int64_t)T * D;
        // two-term operands (ibloc.h): only in blocks that run on every row (the CLS-only last block stays plain)
        const int qt = (!cls_only && L->w_qkv_x && L->qkv_terms > 1) ? L->qkv_terms : 1;
        const int ft = (!cls_only && L->w_fc1_x && L->fc1_terms > 1) ? L->fc1_terms : 1;
        if (qt > 3 || ft > 3) return ibl_set_error(IBL_ERR_ARG, "ibl_vit_forward: layer %d: at most three operand terms", l);
#define IBL_LN_TERMS(terms, g_, b_)                                                                                                 \
        do {                                                                                                                            \
            if ((terms) == 1) hipLaunchKernelGGL((ibl_layernorm_kernel<false, 1>), ln_grid, dim3(256), 0, s, x, (int64_t)D, Rn, D, g_, b_, d->ln_eps, (void*)xn, (int64_t)D);            \
            else if ((terms) == 2) hipLaunchKernelGGL((ibl_layernorm_kernel<false, 2>), ln_grid, dim3(256), 0, s, x, (int64_t)D, Rn, D, g_, b_, d->ln_eps, (void*)xn, (int64_t)2 * D);   \
            else hipLaunchKernelGGL((ibl_layernorm_kernel<false, 3>), ln_grid, dim3(256), 0, s, x, (int64_t)D, Rn, D, g_, b_, d->ln_eps, (void*)xn, (int64_t)3 * D);                      \
        } while (0)
        IBL_LN_TERMS(qt, L->ln1_g, L->ln1_b);
        IBL_LAUNCH_CHECK();
        if (!cls_only) {
            GemmEpi e{};
            e.bias = L->b_qkv; e.out = qkv; e.ldo = 3 * D; e.algo_k = D;
            st = launch_gemm<EPI_BIAS_H16>(xn, (int64_t)qt * D, reinterpret_cast<const u16*>(qt > 1 ? L->w_qkv_x : L->w_qkv), (int64_t)qt * D, (int)Rn,
                                           3 * D, qt * D, e, s);
            if (st) return st;
        } else {
            GemmEpi e{};                                  // keys and values of every token: weight rows D .. 3D
            e.bias = L->b_qkv + D; e.out = qkv + D; e.ldo = 3 * D;
            st = launch_gemm<EPI_BIAS_H16>(xn, D, reinterpret_cast<const u16*>(L->w_qkv) + (int64_t)D * D, D, (int)Rn, 2 * D, D, e, s);
            if (st) return st;
            GemmEpi q{};                                  // queries of the CLS rows only
            q.bias = L->b_qkv; q.out = qkv; q.ldo = 3 * TD;
            st = launch_gemm<EPI_BIAS_H16>(xn, TD, reinterpret_cast<const u16*>(L->w_qkv), D, batch, D, D, q, s);
            if (st) return st;
        }
        // K-extended inputs of the two residual GEMMs (w_o_x / w_fc2_x; round 4): ONE launch of K' = terms * K instead of one
        // read-modify-write pass over the residual per term
        const int ot = (!cls_only && L->w_o_x && L->o_terms > 1) ? L->o_terms : 1;
        const int f2t = (!cls_only && L->w_fc2_x && L->fc2_terms > 1) ? L->fc2_terms : 1;
        if (ot > at || f2t > at || ot > 3 || f2t > 3)
            return ibl_set_error(IBL_ERR_ARG, "ibl_vit_forward: layer %d: o_terms / fc2_terms %d / %d need IBL_VIT_ACT_TERMS%d in the descriptor", l, ot, f2t, ot > f2t ? ot : f2t);
        st = run_attention(qkv, att, batch, T, D, H, cls_only ? 1 : 0, s, ot);
        if (st) return st;
        {
            GemmEpi e{};
            e.bias = L->b_o; e.scale = L->ls1; e.out = x; e.ldo = cls_only ? TD : D; e.alpha = 1.0f; e.algo_k = D;
            const u16* wo = reinterpret_cast<const u16*>(ot > 1 ? L->w_o_x : L->w_o);
            const int64_t lda = cls_only ? TD : (int64_t)ot * D;
            // no LayerScale vector (none in the model, or folded into W_o / b_o by the host): the residual tile is preloaded into the
            // accumulators (EPI_RESID_PRE_F32, lab); with one, the read-modify-write epilogue applies it
            st = (L->ls1 || !gemm_resid_pre()) ? launch_gemm<EPI_RESID_F32>(att, lda, wo, (int64_t)ot * D, cls_only ? batch : (int)Rn, D, ot * D, e, s)
                        : launch_gemm<EPI_RESID_PRE_F32>(att, lda, wo, (int64_t)ot * D, cls_only ? batch : (int)Rn, D, ot * D, e, s);
            if (st) return st;
            if (!cls_only && ot == 1 && L->w_o_lo) {         // (older form) second weight term as its own launch: x += (ls1 / S) * (att W_lo'^T)
                e.bias = nullptr; e.scale = L->ls1_lo; e.algo_k = -1; e.alpha = 1.0f / IBL_VIT_SPLIT_SCALE;
                st = L->ls1_lo ? launch_gemm<EPI_RESID_F32>(att, D, reinterpret_cast<const u16*>(L->w_o_lo), D, (int)Rn, D, D, e, s)
                               : launch_gemm<EPI_RESID_PRE_F32>(att, D, reinterpret_cast<const u16*>(L->w_o_lo), D, (int)Rn, D, D, e, s);
                if (st) return st;
            }
        }
        if (!cls_only) {
            IBL_LN_TERMS(ft, L->ln2_g, L->ln2_b);
        } else {
            hipLaunchKernelGGL(ibl_layernorm_kernel<false>, dim3((unsigned)((batch + 3) / 4)), dim3(256), 0, s, x, TD, (int64_t)batch, D,
                               L->ln2_g, L->ln2_b, d->ln_eps, (void*)fin_bf, (int64_t)D);
        }
        IBL_LAUNCH_CHECK();
        const u16* mlp_in = cls_only ? fin_bf : xn;
        const int mlp_rows = cls_only ? batch : (int)Rn;
        {
            GemmEpi e{};
            e.bias = L->b_fc1; e.out = hid; e.ldo = (int64_t)f2t * d->mlp_dim; e.algo_k = D;
            if (d->flags & IBL_VIT_QUICK_GELU)
                return ibl_set_error(IBL_ERR_UNSUPPORTED, "ibl_vit_forward: QuickGELU not built");
            const u16* w1 = reinterpret_cast<const u16*>(ft > 1 ? L->w_fc1_x : L->w_fc1);
            if (f2t == 3) st = launch_gemm<EPI_BIAS_GELU_H16KX3>(mlp_in, (int64_t)ft * D, w1, (int64_t)ft * D, mlp_rows, d->mlp_dim, ft * D, e, s);
            else if (f2t == 2) st = launch_gemm<EPI_BIAS_GELU_H16KX2>(mlp_in, (int64_t)ft * D, w1, (int64_t)ft * D, mlp_rows, d->mlp_dim, ft * D, e, s);
            else st = launch_gemm<EPI_BIAS_GELU_H16>(mlp_in, (int64_t)ft * D, w1, (int64_t)ft * D, mlp_rows, d->mlp_dim, ft * D, e, s);
            if (st) return st;
        }
        {
            GemmEpi e{};
            e.bias = L->b_fc2; e.scale = L->ls2; e.out = x; e.ldo = cls_only ? TD : D; e.alpha = 1.0f; e.algo_k = d->mlp_dim;
            const u16* w2 = reinterpret_cast<const u16*>(f2t > 1 ? L->w_fc2_x : L->w_fc2);
            const int64_t k2 = (int64_t)f2t * d->mlp_dim;
            st = (L->ls2 || !gemm_resid_pre()) ? launch_gemm<EPI_RESID_F32>(hid, k2, w2, k2, mlp_rows, D, (int)k2, e, s)
                        : launch_gemm<EPI_RESID_PRE_F32>(hid, k2, w2, k2, mlp_rows, D, (int)k2, e, s);
            if (st) return st;
            if (!cls_only && f2t == 1 && L->w_fc2_lo) {
                e.bias = nullptr; e.scale = L->ls2_lo; e.algo_k = -1; e.alpha = 1.0f / IBL_VIT_SPLIT_SCALE;
                st = L->ls2_lo ? launch_gemm<EPI_RESID_F32>(hid, d->mlp_dim, reinterpret_cast<const u16*>(L->w_fc2_lo), d->mlp_dim, mlp_rows, D, d->mlp_dim, e, s)
                               : launch_gemm<EPI_RESID_PRE_F32>(hid, d->mlp_dim, reinterpret_cast<const u16*>(L->w_fc2_lo), d->mlp_dim, mlp_rows, D, d->mlp_dim, e, s);
                if (st) return st;
            }
        }
    }
#undef IBL_LN_TERMS
    if (d->flags & IBL_VIT_OUT_ALL_TOKENS) {
        // DATOR streams: all tokens, optionally through the final LayerNorm
        if (d->flags & IBL_VIT_FINAL_LN) {
            hipLaunchKernelGGL(ibl_layernorm_kernel<true>, ln_grid, dim3(256), 0, s, x, (int64_t)D, Rn, D, w->ln_f_g,
                               w->ln_f_b, d->ln_eps, (void*)out, (int64_t)D);
            IBL_LAUNCH_CHECK();
        } else {
            IBL_HIP_CHECK(hipMemcpyAsync(out, x, Rn * D * 4, hipMemcpyDeviceToDevice, s));
        }
        return IBL_OK;
    }
    // CLS rows -> (final LN) -> (projection) -> out fp32 [batch][out_dim]
    const dim3 cls_grid((unsigned)((batch + 3) / 4));
    if (d->flags & IBL_VIT_PROJ) {
        if (!(d->flags & IBL_VIT_FINAL_LN)) return ibl_set_error(IBL_ERR_UNSUPPORTED, "projection needs final LN");
        GemmEpi e{};
        e.bias = nullptr; e.out = out; e.ldo = d->out_dim;
        if (w->w_proj_x) {           // three-term operands: K' = 3 D, one accumulation (xn is free by now: batch <= R rows of 3 D)
            hipLaunchKernelGGL((ibl_layernorm_kernel<false, 3>), cls_grid, dim3(256), 0, s, x, (int64_t)T * D, (int64_t)batch, D,
                               w->ln_f_g, w->ln_f_b, d->ln_eps, (void*)xn, (int64_t)3 * D);
            IBL_LAUNCH_CHECK();
            e.algo_k = D;
            return launch_gemm<EPI_BIAS_F32>(xn, (int64_t)3 * D, reinterpret_cast<const u16*>(w->w_proj_x), (int64_t)3 * D, batch, d->out_dim,
                                             3 * D, e, s);
        }
        hipLaunchKernelGGL(ibl_layernorm_kernel<false>, cls_grid, dim3(256), 0, s, x, (int64_t)T * D, (int64_t)batch, D,
                           w->ln_f_g, w->ln_f_b, d->ln_eps, (void*)fin_bf, (int64_t)D);
        IBL_LAUNCH_CHECK();
        return launch_gemm<EPI_BIAS_F32>(fin_bf, D, reinterpret_cast<const u16*>(w->w_proj), D, batch, d->out_dim, D, e, s);
    }
    if (d->flags & IBL_VIT_FINAL_LN) {
        hipLaunchKernelGGL(ibl_layernorm_kernel<true>, cls_grid, dim3(256), 0, s, x, (int64_t)T * D, (int64_t)batch, D,
                           w->ln_f_g, w->ln_f_b, d->ln_eps, (void*)out, (int64_t)D);
        IBL_LAUNCH_CHECK();
    } else {
        IBL_HIP_CHECK(hipMemcpy2DAsync(out, (size_t)D * 4, x, (size_t)T * D * 4, (size_t)D * 4, batch,
                                       hipMemcpyDeviceToDevice, s));
    }
    return IBL_OK;
}
