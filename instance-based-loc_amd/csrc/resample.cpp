// Host side of crop preprocessing: Pillow's 8-bit resample coefficient tables (libImaging/Resample.c: precompute_coeffs +
// normalize_coeffs_8bpc, restated) for one pass of the separable resize that `ibl_preprocess_crops` executes on the device.
// The reference reaches this through the HF / open_clip image processors of its embedding functions (utils/embeddings.py:41-42, 64-65,
// 86-89), once per crop on the CPU; a batch of differently sized crops needs two tables per crop, and building them in numpy cost
// 0.25 ms per crop (a config-C1 step of 128 crops: 31 ms of host time in front of 5 ms of device work).
#include <cmath>
#include <cstdint>
#include <vector>

#include "ibl_common.h"
#include "ibloc.h"

#pragma clang fp contract(off)

namespace {
constexpr int PRECISION_BITS = 32 - 8 - 2;

inline double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}
inline double bilinear_filter(double x) {
    if (x < 0.0) x = -x;
    return x < 1.0 ? 1.0 - x : 0.0;
}
}  // namespace

extern "C" int ibl_resample_ksize(int in_size, int out_size, int filter) {
    if (in_size <= 0 || out_size <= 0 || (filter != IBL_FILTER_BILINEAR && filter != IBL_FILTER_BICUBIC)) return -1;
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale >= 1.0 ? scale : 1.0;
    const double support = (filter == IBL_FILTER_BICUBIC ? 2.0 : 1.0) * filterscale;
    return (int)std::ceil(support) * 2 + 1;
}

extern "C" int ibl_resample_table(int in_size, int out_size, int filter, int win0, int win_n, int32_t* rec) {
    const int ksize = ibl_resample_ksize(in_size, out_size, filter);
    if (ksize < 0 || win0 < 0 || win_n <= 0 || !rec) return ibl_set_error(IBL_ERR_ARG, "ibl_resample_table: bad argument");
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale >= 1.0 ? scale : 1.0;
    const double support = (filter == IBL_FILTER_BICUBIC ? 2.0 : 1.0) * filterscale;
    const double ss = 1.0 / filterscale;
    std::vector<double> w((size_t)ksize);
    for (int i = 0; i < win_n; ++i) {
        const double center = ((double)(win0 + i) + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            const double arg = ((double)x + (double)xmin - center + 0.5) * ss;
            w[x] = filter == IBL_FILTER_BICUBIC ? bicubic_filter(arg) : bilinear_filter(arg);
            ww += w[x];
        }
        int32_t* r = rec + (size_t)i * (2 + ksize);
        r[0] = xmin;
        r[1] = xmax;
        for (int x = 0; x < ksize; ++x) {
            if (x >= xmax) { r[2 + x] = 0; continue; }
            double v = w[x];
            if (ww != 0.0) v = v / ww;
            // C's double -> int conversion truncates toward zero, like Pillow's (int) casts
            r[2 + x] = v < 0 ? (int32_t)(-0.5 + v * (double)(1 << PRECISION_BITS)) : (int32_t)(0.5 + v * (double)(1 << PRECISION_BITS));
        }
    }
    return ksize;
}
