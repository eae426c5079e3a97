/* ORACLE (test infrastructure; never linked or called by the product path).
 *
 * CPU restatement of the point-cloud registration the reference delegates to Open3D 0.17.0
 * (third-party wheel, environment.yml:229; source NOT under /root/reference and not importable
 * here -> PARITY UNPINNED at the Open3D boundary).  Control flow and parameters
 * follow the reference's own call sites; the Open3D semantics follow SURVEY.md Appendix A and the
 * public 0.17.0 sources:
 *   utils/fpfh_register.py:86-98    normals (hybrid r = 2*voxel, 30 nn) + FPFH (r = 5*voxel, 100 nn)
 *   utils/fpfh_register.py:100-143  RANSAC on feature matches (mutual filter, n = 3, edge-length 0.9 and
 *                                   distance checkers, 4e6 iterations / 0.99) -> coloured ICP
 *                                   (lambda_geometric 0.968, 30 iterations, 1e-6 / 1e-6); fallback p2p ICP
 *   utils/fpfh_register.py:145-150  evaluate_registration
 *   object_memory/object_memory.py:992-998  remove_radius_outlier(nb_points, radius)
 * Open3D's RANSAC draws from an unseeded global RNG inside an OpenMP loop (non-deterministic).  This
 * restatement walks the hypotheses in index order with a counter-based Philox4x32-10 generator
 * (hypothesis i of job j draws from counter (i, j, 0, 0), key = seed), which is what the HIP path
 * reproduces.  Point coordinates, normals and features are fp32 (the HBM layout of the product);
 * neighbour selection uses the fp32 distance d2 = fmaf(dz,dz,fmaf(dy,dy,dx*dx)) so that neighbour
 * SETS are bit-identical to the device; everything else is double.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
void oracle_set_threads(int n) { omp_set_num_threads(n); }
#else
void oracle_set_threads(int n) { (void)n; }
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------------ */
/* uniform grid over one cloud (cell = search radius), stable counting sort                    */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    float minx, miny, minz, inv;
    int nx, ny, nz;
    int* start;   /* ncell + 1 */
    int* order;   /* n: point indices sorted by cell (stable) */
} grid_t;

static inline int cell_of(const grid_t* g, float v, float mn, int n) {
    int c = (int)floorf((v - mn) * g->inv);
    if (c < 0) c = 0;
    if (c >= n) c = n - 1;
    return c;
}

static void grid_build(grid_t* g, const float* p, int n, float cell) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) {
            if (p[3 * i + a] < mn[a]) mn[a] = p[3 * i + a];
            if (p[3 * i + a] > mx[a]) mx[a] = p[3 * i + a];
        }
    if (n == 0) { mn[0] = mn[1] = mn[2] = 0; mx[0] = mx[1] = mx[2] = 0; }
    g->minx = mn[0]; g->miny = mn[1]; g->minz = mn[2];
    g->inv = 1.0f / cell;
    g->nx = (int)floorf((mx[0] - mn[0]) * g->inv) + 1;
    g->ny = (int)floorf((mx[1] - mn[1]) * g->inv) + 1;
    g->nz = (int)floorf((mx[2] - mn[2]) * g->inv) + 1;
    /* keep the table bounded for degenerate inputs */
    while ((int64_t)g->nx * g->ny * g->nz > (int64_t)1 << 26) {
        cell *= 2; g->inv = 1.0f / cell;
        g->nx = (int)floorf((mx[0] - mn[0]) * g->inv) + 1;
        g->ny = (int)floorf((mx[1] - mn[1]) * g->inv) + 1;
        g->nz = (int)floorf((mx[2] - mn[2]) * g->inv) + 1;
    }
    int ncell = g->nx * g->ny * g->nz;
    g->start = (int*)calloc((size_t)ncell + 1, sizeof(int));
    g->order = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    int* cid = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) {
        int cx = cell_of(g, p[3 * i], g->minx, g->nx), cy = cell_of(g, p[3 * i + 1], g->miny, g->ny),
            cz = cell_of(g, p[3 * i + 2], g->minz, g->nz);
        cid[i] = (cz * g->ny + cy) * g->nx + cx;
        g->start[cid[i] + 1]++;
    }
    for (int c = 0; c < ncell; ++c) g->start[c + 1] += g->start[c];
    int* fill = (int*)malloc(sizeof(int) * (size_t)(ncell > 0 ? ncell : 1));
    memcpy(fill, g->start, sizeof(int) * (size_t)ncell);
    for (int i = 0; i < n; ++i) g->order[fill[cid[i]]++] = i;
    free(fill);
    free(cid);
}

static void grid_free(grid_t* g) { free(g->start); free(g->order); }

static inline float dist2f(const float* a, const float* b) {
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

typedef struct { float d2; int idx; } nb_t;
static int nb_cmp(const void* a, const void* b) {
    const nb_t* x = (const nb_t*)a; const nb_t* y = (const nb_t*)b;
    if (x->d2 < y->d2) return -1;
    if (x->d2 > y->d2) return 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

/* hybrid search: the <= max_nn nearest points with d2 < r2 (strict), sorted by (d2, idx).
 * `reach` = number of cells to scan on each side (radius may exceed the cell size). */
static int hybrid_search(const grid_t* g, const float* pts, const float* q, float r2, int max_nn, float radius,
                         nb_t* buf, int cap) {
    int reach = (int)ceilf(radius * g->inv);
    if (reach < 1) reach = 1;
    int cx = cell_of(g, q[0], g->minx, g->nx), cy = cell_of(g, q[1], g->miny, g->ny), cz = cell_of(g, q[2], g->minz, g->nz);
    int n = 0;
    for (int z = cz - reach; z <= cz + reach; ++z) {
        if (z < 0 || z >= g->nz) continue;
        for (int y = cy - reach; y <= cy + reach; ++y) {
            if (y < 0 || y >= g->ny) continue;
            int x0 = cx - reach < 0 ? 0 : cx - reach, x1 = cx + reach >= g->nx ? g->nx - 1 : cx + reach;
            int c0 = (z * g->ny + y) * g->nx + x0, c1 = (z * g->ny + y) * g->nx + x1;
            for (int s = g->start[c0]; s < g->start[c1 + 1]; ++s) {
                int j = g->order[s];
                float d2 = dist2f(q, pts + 3 * j);
                if (d2 < r2) {
                    if (n < cap) { buf[n].d2 = d2; buf[n].idx = j; }
                    ++n;
                }
            }
        }
    }
    if (n > cap) n = cap;   /* cap is sized to the cloud, cannot trigger */
    qsort(buf, (size_t)n, sizeof(nb_t), nb_cmp);
    return n < max_nn ? n : max_nn;
}

/* ------------------------------------------------------------------------------------------ */
/* remove_radius_outlier                                                                       */
/* ------------------------------------------------------------------------------------------ */
void oracle_radius_outlier(const float* pts, int n, double radius, int nb_points, uint8_t* keep) {
    grid_t g;
    grid_build(&g, pts, n, (float)radius);
    const float r2 = (float)(radius * radius);
    nb_t* buf = (nb_t*)malloc(sizeof(nb_t) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) {
        int cnt = hybrid_search(&g, pts, pts + 3 * i, r2, n, (float)radius, buf, n);
        keep[i] = cnt > nb_points;
    }
    free(buf);
    grid_free(&g);
}

/* ------------------------------------------------------------------------------------------ */
/* normals: covariance of the hybrid neighbourhood + Open3D's FastEigen3x3 (Eberly's robust
 * symmetric 3x3 solver), smallest-eigenvalue eigenvector, no orientation step                  */
/* ------------------------------------------------------------------------------------------ */
static void cross3(const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

static void eigenvector0(const double A[6], double ev, double* out) {
    /* A = [a00 a01 a02 a11 a12 a22] */
    double r0[3] = {A[0] - ev, A[1], A[2]}, r1[3] = {A[1], A[3] - ev, A[4]}, r2[3] = {A[2], A[4], A[5] - ev};
    double c01[3], c02[3], c12[3];
    cross3(r0, r1, c01); cross3(r0, r2, c02); cross3(r1, r2, c12);
    double d0 = dot3(c01, c01), d1 = dot3(c02, c02), d2 = dot3(c12, c12);
    double dmax = d0; int imax = 0;
    if (d1 > dmax) { dmax = d1; imax = 1; }
    if (d2 > dmax) { imax = 2; }
    const double* c = imax == 0 ? c01 : (imax == 1 ? c02 : c12);
    double d = imax == 0 ? d0 : (imax == 1 ? d1 : d2);
    double s = sqrt(d);
    out[0] = c[0] / s; out[1] = c[1] / s; out[2] = c[2] / s;
}

static void eigenvector1(const double A[6], const double* e0, double ev1, double* out) {
    double U[3], V[3];
    if (fabs(e0[0]) > fabs(e0[1])) {
        double inv = 1.0 / sqrt(e0[0] * e0[0] + e0[2] * e0[2]);
        U[0] = -e0[2] * inv; U[1] = 0; U[2] = e0[0] * inv;
    } else {
        double inv = 1.0 / sqrt(e0[1] * e0[1] + e0[2] * e0[2]);
        U[0] = 0; U[1] = e0[2] * inv; U[2] = -e0[1] * inv;
    }
    cross3(e0, U, V);
    double AU[3] = {A[0] * U[0] + A[1] * U[1] + A[2] * U[2], A[1] * U[0] + A[3] * U[1] + A[4] * U[2],
                    A[2] * U[0] + A[4] * U[1] + A[5] * U[2]};
    double AV[3] = {A[0] * V[0] + A[1] * V[1] + A[2] * V[2], A[1] * V[0] + A[3] * V[1] + A[4] * V[2],
                    A[2] * V[0] + A[4] * V[1] + A[5] * V[2]};
    double m00 = dot3(U, AU) - ev1, m01 = dot3(U, AV), m11 = dot3(V, AV) - ev1;
    double a00 = fabs(m00), a01 = fabs(m01), a11 = fabs(m11);
    if (a00 >= a11) {
        double mx = a00 > a01 ? a00 : a01;
        if (mx > 0) {
            if (a00 >= a01) { m01 /= m00; m00 = 1 / sqrt(1 + m01 * m01); m01 *= m00; }
            else { m00 /= m01; m01 = 1 / sqrt(1 + m00 * m00); m00 *= m01; }
            for (int i = 0; i < 3; ++i) out[i] = m01 * U[i] - m00 * V[i];
        } else { out[0] = U[0]; out[1] = U[1]; out[2] = U[2]; }
    } else {
        double mx = a11 > a01 ? a11 : a01;
        if (mx > 0) {
            if (a11 >= a01) { m01 /= m11; m11 = 1 / sqrt(1 + m01 * m01); m01 *= m11; }
            else { m11 /= m01; m01 = 1 / sqrt(1 + m11 * m11); m11 *= m01; }
            for (int i = 0; i < 3; ++i) out[i] = m11 * U[i] - m01 * V[i];
        } else { out[0] = U[0]; out[1] = U[1]; out[2] = U[2]; }
    }
}

/* cov = [c00 c01 c02 c11 c12 c22]; returns the eigenvector of the smallest eigenvalue */
static void fast_eigen_normal(const double cov[6], double* n) {
    double mc = cov[0];
    for (int i = 1; i < 6; ++i) if (cov[i] > mc) mc = cov[i];
    if (mc == 0) { n[0] = n[1] = n[2] = 0; return; }
    double A[6];
    for (int i = 0; i < 6; ++i) A[i] = cov[i] / mc;
    double norm = A[1] * A[1] + A[2] * A[2] + A[4] * A[4];
    if (norm > 0) {
        double q = (A[0] + A[3] + A[5]) / 3;
        double b00 = A[0] - q, b11 = A[3] - q, b22 = A[5] - q;
        double p = sqrt((b00 * b00 + b11 * b11 + b22 * b22 + norm * 2) / 6);
        double c00 = b11 * b22 - A[4] * A[4];
        double c01 = A[1] * b22 - A[4] * A[2];
        double c02 = A[1] * A[4] - b11 * A[2];
        double det = (b00 * c00 - A[1] * c01 + A[2] * c02) / (p * p * p);
        double half_det = det * 0.5;
        if (half_det < -1.0) half_det = -1.0;
        if (half_det > 1.0) half_det = 1.0;
        double angle = acos(half_det) / 3.0;
        const double two_thirds_pi = 2.09439510239319549;
        double beta2 = cos(angle) * 2;
        double beta0 = cos(angle + two_thirds_pi) * 2;
        double beta1 = -(beta0 + beta2);
        double e0 = q + p * beta0, e1 = q + p * beta1, e2 = q + p * beta2;
        double v0[3], v1[3], v2[3];
        if (half_det >= 0) {
            eigenvector0(A, e2, v2);
            if (e2 < e0 && e2 < e1) { memcpy(n, v2, sizeof(v2)); return; }
            eigenvector1(A, v2, e1, v1);
            if (e1 < e0 && e1 < e2) { memcpy(n, v1, sizeof(v1)); return; }
            cross3(v1, v2, v0);
            memcpy(n, v0, sizeof(v0));
        } else {
            eigenvector0(A, e0, v0);
            if (e0 < e1 && e0 < e2) { memcpy(n, v0, sizeof(v0)); return; }
            eigenvector1(A, v0, e1, v1);
            if (e1 < e0 && e1 < e2) { memcpy(n, v1, sizeof(v1)); return; }
            cross3(v0, v1, v2);
            memcpy(n, v2, sizeof(v2));
        }
    } else {
        if (cov[0] < cov[3] && cov[0] < cov[5]) { n[0] = 1; n[1] = 0; n[2] = 0; }
        else if (cov[3] < cov[0] && cov[3] < cov[5]) { n[0] = 0; n[1] = 1; n[2] = 0; }
        else { n[0] = 0; n[1] = 0; n[2] = 1; }
    }
}

void oracle_normals(const float* pts, int n, double radius, int max_nn, float* normals) {
    grid_t g;
    grid_build(&g, pts, n, (float)radius);
    const float r2 = (float)(radius * radius);
#pragma omp parallel
    {
        nb_t* buf = (nb_t*)malloc(sizeof(nb_t) * (size_t)(n > 0 ? n : 1));
#pragma omp for schedule(dynamic, 64)
        for (int i = 0; i < n; ++i) {
            int k = hybrid_search(&g, pts, pts + 3 * i, r2, max_nn, (float)radius, buf, n);
            double nrm[3];
            if (k >= 3) {
                double c[9] = {0};
                for (int t = 0; t < k; ++t) {
                    const float* p = pts + 3 * buf[t].idx;
                    double x = p[0], y = p[1], z = p[2];
                    c[0] += x; c[1] += y; c[2] += z;
                    c[3] += x * x; c[4] += x * y; c[5] += x * z; c[6] += y * y; c[7] += y * z; c[8] += z * z;
                }
                for (int t = 0; t < 9; ++t) c[t] /= (double)k;
                double cov[6] = {c[3] - c[0] * c[0], c[4] - c[0] * c[1], c[5] - c[0] * c[2],
                                 c[6] - c[1] * c[1], c[7] - c[1] * c[2], c[8] - c[2] * c[2]};
                fast_eigen_normal(cov, nrm);
            } else {
                double cov[6] = {1, 0, 0, 1, 0, 1};      /* identity covariance */
                fast_eigen_normal(cov, nrm);
            }
            if (sqrt(dot3(nrm, nrm)) == 0.0) { nrm[0] = 0; nrm[1] = 0; nrm[2] = 1; }
            normals[3 * i] = (float)nrm[0]; normals[3 * i + 1] = (float)nrm[1]; normals[3 * i + 2] = (float)nrm[2];
        }
        free(buf);
    }
    grid_free(&g);
}

/* ------------------------------------------------------------------------------------------ */
/* FPFH                                                                                         */
/* ------------------------------------------------------------------------------------------ */
static void pair_features(const float* p1f, const float* n1f, const float* p2f, const float* n2f, double* f) {
    double d[3] = {(double)p2f[0] - p1f[0], (double)p2f[1] - p1f[1], (double)p2f[2] - p1f[2]};
    double n1[3] = {n1f[0], n1f[1], n1f[2]}, n2[3] = {n2f[0], n2f[1], n2f[2]};
    double r = sqrt(dot3(d, d));
    f[0] = f[1] = f[2] = 0;
    if (r == 0.0) return;
    double a1 = dot3(n1, d) / r, a2 = dot3(n2, d) / r;
    double na[3], nb[3];
    if (acos(fabs(a1)) > acos(fabs(a2))) {
        memcpy(na, n2, sizeof(na)); memcpy(nb, n1, sizeof(nb));
        d[0] = -d[0]; d[1] = -d[1]; d[2] = -d[2];
        f[2] = -a2;
    } else {
        memcpy(na, n1, sizeof(na)); memcpy(nb, n2, sizeof(nb));
        f[2] = a1;
    }
    double v[3], w[3];
    cross3(d, na, v);
    double vn = sqrt(dot3(v, v));
    if (vn == 0.0) { f[0] = f[1] = f[2] = 0; return; }
    v[0] /= vn; v[1] /= vn; v[2] /= vn;
    cross3(na, v, w);
    f[1] = dot3(v, nb);
    f[0] = atan2(dot3(w, nb), dot3(na, nb));
}

static inline int clamp_bin(int h) { return h < 0 ? 0 : (h >= 11 ? 10 : h); }

void oracle_fpfh(const float* pts, const float* normals, int n, double radius, int max_nn, float* fpfh /* n x 33 */) {
    grid_t g;
    grid_build(&g, pts, n, (float)radius);
    const float r2 = (float)(radius * radius);
    float* spfh = (float*)calloc((size_t)n * 33 + 1, sizeof(float));
    unsigned char* cnt = (unsigned char*)calloc((size_t)n * 33 + 1, 1);      /* the integer SPFH histograms (<= max_nn - 1 per bin) */
    double* incs = (double*)calloc((size_t)n + 1, sizeof(double));           /* 100 / (k - 1) */
    int* nbr = (int*)malloc(sizeof(int) * (size_t)n * (size_t)max_nn + 4);
    float* nd2 = (float*)malloc(sizeof(float) * (size_t)n * (size_t)max_nn + 4);
    int* ncnt = (int*)malloc(sizeof(int) * (size_t)n + 4);
#pragma omp parallel
    {
        nb_t* buf = (nb_t*)malloc(sizeof(nb_t) * (size_t)(n > 0 ? n : 1));
#pragma omp for schedule(dynamic, 64)
        for (int i = 0; i < n; ++i) {
            int k = hybrid_search(&g, pts, pts + 3 * i, r2, max_nn, (float)radius, buf, n);
            ncnt[i] = k;
            for (int t = 0; t < k; ++t) { nbr[(size_t)i * max_nn + t] = buf[t].idx; nd2[(size_t)i * max_nn + t] = buf[t].d2; }
            if (k > 1) {
                int hist[33] = {0};
                for (int t = 0; t < k; ++t) {
                    int j = buf[t].idx;
                    if (j == i) continue;                      /* skip the point itself */
                    double f[3];
                    pair_features(pts + 3 * i, normals + 3 * i, pts + 3 * j, normals + 3 * j, f);
                    hist[clamp_bin((int)floor(11 * (f[0] + M_PI) / (2.0 * M_PI)))]++;
                    hist[11 + clamp_bin((int)floor(11 * (f[1] + 1.0) * 0.5))]++;
                    hist[22 + clamp_bin((int)floor(11 * (f[2] + 1.0) * 0.5))]++;
                }
                double inc = 100.0 / (double)(k - 1);
                incs[i] = inc;
                for (int b = 0; b < 33; ++b) { spfh[(size_t)i * 33 + b] = (float)(hist[b] * inc); cnt[(size_t)i * 33 + b] = (unsigned char)hist[b]; }
            }
        }
        free(buf);
    }
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; ++i) {
        int k = ncnt[i];
        double acc[33] = {0}, sum[3] = {0, 0, 0};
        if (k > 1) {
            for (int t = 0; t < k; ++t) {
                int j = nbr[(size_t)i * max_nn + t];
                if (j == i) continue;
                double dist = nd2[(size_t)i * max_nn + t];
                if (dist == 0.0) continue;
                /* SPFH(j)[b] / d2 with the neighbour's histogram in double, as Open3D keeps it (Feature::data_ is a MatrixXd): count x
                 * (increment / d2) -- one division per neighbour; it differs from (count x increment) / d2 by an ulp of double */
                const double w = incs[j] / dist;
                for (int b = 0; b < 33; ++b) {
                    double val = (double)cnt[(size_t)j * 33 + b] * w;
                    sum[b / 11] += val;
                    acc[b] += val;
                }
            }
            for (int s = 0; s < 3; ++s) if (sum[s] != 0.0) sum[s] = 100.0 / sum[s];
            for (int b = 0; b < 33; ++b) fpfh[(size_t)i * 33 + b] = (float)(acc[b] * sum[b / 11] + spfh[(size_t)i * 33 + b]);
        } else {
            for (int b = 0; b < 33; ++b) fpfh[(size_t)i * 33 + b] = 0.0f;
        }
    }
    free(spfh); free(cnt); free(incs); free(nbr); free(nd2); free(ncnt);
    grid_free(&g);
}

/* ------------------------------------------------------------------------------------------ */
/* feature matching: 33-d 1-NN both ways (fp32 chain), mutual filter                           */
/* ------------------------------------------------------------------------------------------ */
/* Summation order of the 33 squared differences ("matching order"): the bins of the three 11-bin histograms from their
 * centres outwards, interleaved (angle histograms vary most around their middle bins), so that a partial sum grows as fast
 * as possible and the product's early-abandon search drops a candidate after a few terms.  Any fixed order is a valid
 * restatement of the reference's L2 distance (Open3D's KD-tree fixes none); oracle and product share this one so that
 * every fp32 distance, and with it every nearest-neighbour decision, is bit-identical. */
static const int FEAT_ORDER[33] = {16, 27, 5, 15, 26, 4, 17, 28, 6, 14, 25, 3, 18, 29, 7, 13, 24, 2, 19, 30, 8, 12, 23, 1, 20, 31, 9, 11, 22, 0, 21, 32, 10};
static inline float feat_d2(const float* a, const float* b) {
    float acc = 0.0f;
    for (int t = 0; t < 33; ++t) { const int k = FEAT_ORDER[t]; float d = a[k] - b[k]; acc = fmaf(d, d, acc); }
    return acc;
}

static void nn_features(const float* fa, int na, const float* fb, int nb, int* out) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < na; ++i) {
        float best = INFINITY; int bj = 0;
        for (int j = 0; j < nb; ++j) {
            float d = feat_d2(fa + (size_t)i * 33, fb + (size_t)j * 33);
            if (d < best) { best = d; bj = j; }       /* first minimum wins */
        }
        out[i] = bj;
    }
}

/* returns the number of correspondences written to corr (pairs of int32: src idx, tgt idx) */
int oracle_feature_match(const float* fs, int ns, const float* ft, int nt, int mutual_filter, int ransac_n, int32_t* corr) {
    if (ns <= 0 || nt <= 0) return 0;
    int* ij = (int*)malloc(sizeof(int) * (size_t)ns);
    nn_features(fs, ns, ft, nt, ij);
    int nc = 0;
    if (mutual_filter) {
        int* ji = (int*)malloc(sizeof(int) * (size_t)nt);
        nn_features(ft, nt, fs, ns, ji);
        for (int i = 0; i < ns; ++i)
            if (ji[ij[i]] == i) { corr[2 * nc] = i; corr[2 * nc + 1] = ij[i]; ++nc; }
        free(ji);
        if (nc >= ransac_n * 3) { free(ij); return nc; }
    }
    for (int i = 0; i < ns; ++i) { corr[2 * i] = i; corr[2 * i + 1] = ij[i]; }
    free(ij);
    return ns;
}

/* ------------------------------------------------------------------------------------------ */
/* Philox4x32-10                                                                                */
/* ------------------------------------------------------------------------------------------ */
static void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* ------------------------------------------------------------------------------------------ */
/* Kabsch (no scaling) on k point pairs: R, t minimising sum |R s + t - d|^2                   */
/* symmetric 3x3 Jacobi on H^T H -> V, sigma; U = H V / sigma; det correction                   */
/* ------------------------------------------------------------------------------------------ */
static void jacobi_eig3(double A[3][3], double V[3][3], double* w) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = i == j;
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (fabs(A[p][q]) < 1e-300) continue;
                double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    w[0] = A[0][0]; w[1] = A[1][1]; w[2] = A[2][2];
}

/* H = sum (d - dm)(s - sm)^T  (3x3, "sigma" of umeyama); R = U S V^T */
static void rotation_from_H(double H[3][3], double R[3][3]) {
    double HtH[3][3], V[3][3], w[3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        HtH[i][j] = 0;
        for (int k = 0; k < 3; ++k) HtH[i][j] += H[k][i] * H[k][j];
    }
    jacobi_eig3(HtH, V, w);
    /* sort eigenpairs descending */
    int ord[3] = {0, 1, 2};
    for (int a = 0; a < 2; ++a) for (int b = a + 1; b < 3; ++b) if (w[ord[b]] > w[ord[a]]) { int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
    double Vs[3][3], U[3][3], sig[3];
    for (int c = 0; c < 3; ++c) { sig[c] = sqrt(w[ord[c]] > 0 ? w[ord[c]] : 0); for (int r = 0; r < 3; ++r) Vs[r][c] = V[r][ord[c]]; }
    /* make V right-handed */
    {
        double c0[3] = {Vs[0][0], Vs[1][0], Vs[2][0]}, c1[3] = {Vs[0][1], Vs[1][1], Vs[2][1]}, c2[3];
        cross3(c0, c1, c2);
        Vs[0][2] = c2[0]; Vs[1][2] = c2[1]; Vs[2][2] = c2[2];
    }
    const double tol = 1e-12 * (sig[0] > 0 ? sig[0] : 1.0);
    int rank = 0;
    for (int c = 0; c < 3; ++c) {
        if (sig[c] > tol) {
            for (int r = 0; r < 3; ++r) U[r][c] = (H[r][0] * Vs[0][c] + H[r][1] * Vs[1][c] + H[r][2] * Vs[2][c]) / sig[c];
            ++rank;
        } else break;
    }
    if (rank == 0) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i][j] = i == j; return; }
    if (rank == 1) {
        /* complete U with any orthonormal pair */
        double u0[3] = {U[0][0], U[1][0], U[2][0]}, a[3] = {1, 0, 0}, u1[3], u2[3];
        if (fabs(u0[0]) > 0.9) { a[0] = 0; a[1] = 1; }
        cross3(u0, a, u1);
        double l = sqrt(dot3(u1, u1)); u1[0] /= l; u1[1] /= l; u1[2] /= l;
        cross3(u0, u1, u2);
        for (int r = 0; r < 3; ++r) { U[r][1] = u1[r]; U[r][2] = u2[r]; }
    } else {
        /* re-orthonormalise column 1 against column 0, third column = u0 x u1 (proper rotation: the
           det(U) det(V) < 0 flip of umeyama Eq. (39) is absorbed because both frames are right-handed) */
        double u0[3] = {U[0][0], U[1][0], U[2][0]}, u1[3] = {U[0][1], U[1][1], U[2][1]}, u2[3];
        double l0 = sqrt(dot3(u0, u0)); u0[0] /= l0; u0[1] /= l0; u0[2] /= l0;
        double pr = dot3(u0, u1); u1[0] -= pr * u0[0]; u1[1] -= pr * u0[1]; u1[2] -= pr * u0[2];
        double l1 = sqrt(dot3(u1, u1)); u1[0] /= l1; u1[1] /= l1; u1[2] /= l1;
        cross3(u0, u1, u2);
        if (rank == 3) {
            /* full rank: a reflection is the LS optimum of the unconstrained problem when det < 0; umeyama then
               flips the axis of the smallest singular value.  u2 = u0 x u1 is exactly that flipped axis. */
        }
        for (int r = 0; r < 3; ++r) { U[r][0] = u0[r]; U[r][1] = u1[r]; U[r][2] = u2[r]; }
    }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i][j] = U[i][0] * Vs[j][0] + U[i][1] * Vs[j][1] + U[i][2] * Vs[j][2];
}

static void kabsch(const double* s, const double* d, int k, double T[16]) {
    double sm[3] = {0, 0, 0}, dm[3] = {0, 0, 0};
    for (int i = 0; i < k; ++i) for (int a = 0; a < 3; ++a) { sm[a] += s[3 * i + a]; dm[a] += d[3 * i + a]; }
    for (int a = 0; a < 3; ++a) { sm[a] /= k; dm[a] /= k; }
    double H[3][3] = {{0}};
    for (int i = 0; i < k; ++i)
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) H[r][c] += (d[3 * i + r] - dm[r]) * (s[3 * i + c] - sm[c]);
    double R[3][3];
    rotation_from_H(H, R);
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) T[4 * r + c] = R[r][c];
        T[4 * r + 3] = dm[r] - (R[r][0] * sm[0] + R[r][1] * sm[1] + R[r][2] * sm[2]);
    }
    T[12] = T[13] = T[14] = 0; T[15] = 1;
}

static inline void xform(const double T[16], const double* p, double* o) {
    o[0] = T[0] * p[0] + T[1] * p[1] + T[2] * p[2] + T[3];
    o[1] = T[4] * p[0] + T[5] * p[1] + T[6] * p[2] + T[7];
    o[2] = T[8] * p[0] + T[9] * p[1] + T[10] * p[2] + T[11];
}

/* ------------------------------------------------------------------------------------------ */
/* RANSAC on correspondences (sequential-order semantics of Open3D's loop, Philox draws)       */
/* stats[0] = hypotheses walked, stats[1] = hypotheses validated, stats[2] = best inliers        */
/* ------------------------------------------------------------------------------------------ */
void oracle_ransac(const float* src, const float* tgt, const int32_t* corr, int nc, double max_dist, uint64_t seed,
                   uint32_t job_id, int max_iter, double confidence, double edge_sim, double T_out[16], int64_t* stats) {
    for (int i = 0; i < 16; ++i) T_out[i] = (i % 5) == 0;
    stats[0] = stats[1] = stats[2] = 0;
    if (nc < 3 || max_dist <= 0) return;
    double best_fit = 0.0, best_rmse = 0.0;
    int64_t est_k = max_iter;
    int64_t itr = 0;
    for (; itr < max_iter && itr < est_k; ++itr) {
        uint32_t r[4];
        philox4x32((uint32_t)itr, job_id, (uint32_t)((uint64_t)itr >> 32), 0, (uint32_t)seed, (uint32_t)(seed >> 32), r);
        int pick[3];
        double s[9], d[9];
        for (int j = 0; j < 3; ++j) {
            pick[j] = (int)(((uint64_t)r[j] * (uint64_t)nc) >> 32);
            for (int a = 0; a < 3; ++a) { s[3 * j + a] = src[3 * corr[2 * pick[j]] + a]; d[3 * j + a] = tgt[3 * corr[2 * pick[j] + 1] + a]; }
        }
        /* edge-length checker */
        int ok = 1;
        for (int a = 0; a < 3 && ok; ++a)
            for (int b = a + 1; b < 3; ++b) {
                double ds = sqrt((s[3 * a] - s[3 * b]) * (s[3 * a] - s[3 * b]) + (s[3 * a + 1] - s[3 * b + 1]) * (s[3 * a + 1] - s[3 * b + 1]) +
                                 (s[3 * a + 2] - s[3 * b + 2]) * (s[3 * a + 2] - s[3 * b + 2]));
                double dt = sqrt((d[3 * a] - d[3 * b]) * (d[3 * a] - d[3 * b]) + (d[3 * a + 1] - d[3 * b + 1]) * (d[3 * a + 1] - d[3 * b + 1]) +
                                 (d[3 * a + 2] - d[3 * b + 2]) * (d[3 * a + 2] - d[3 * b + 2]));
                if (ds < dt * edge_sim || dt < ds * edge_sim) { ok = 0; break; }
            }
        if (!ok) continue;
        double T[16];
        kabsch(s, d, 3, T);
        /* distance checker */
        for (int j = 0; j < 3 && ok; ++j) {
            double p[3];
            xform(T, s + 3 * j, p);
            double dx = p[0] - d[3 * j], dy = p[1] - d[3 * j + 1], dz = p[2] - d[3 * j + 2];
            if (sqrt(dx * dx + dy * dy + dz * dz) > max_dist) ok = 0;
        }
        if (!ok) continue;
        /* validation on the correspondence set */
        int inl = 0; double err2 = 0;
        for (int c = 0; c < nc; ++c) {
            double sp[3] = {src[3 * corr[2 * c]], src[3 * corr[2 * c] + 1], src[3 * corr[2 * c] + 2]}, p[3];
            xform(T, sp, p);
            const float* q = tgt + 3 * corr[2 * c + 1];
            double dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
            double dd = sqrt(dx * dx + dy * dy + dz * dz);
            if (dd < max_dist) { ++inl; err2 += dd * dd; }
        }
        stats[1]++;
        double fit = (double)inl / (double)nc, rmse = inl ? sqrt(err2 / inl) : 0.0;
        if (inl > 0 && (fit > best_fit || (fit == best_fit && rmse < best_rmse))) {
            best_fit = fit; best_rmse = rmse;
            memcpy(T_out, T, sizeof(double) * 16);
            stats[2] = inl;
            double ek = log(1.0 - confidence) / log(1.0 - pow(fit, 3.0));
            if (ek < (double)est_k) est_k = (int64_t)ceil(ek);
        }
    }
    stats[0] = itr;
}

/* ------------------------------------------------------------------------------------------ */
/* 1-NN within max_dist (fp32 distance on the float-rounded query)                              */
/* ------------------------------------------------------------------------------------------ */
static int nn_within(const grid_t* g, const float* pts, const double* q, float r2, float radius, float* d2out) {
    float qf[3] = {(float)q[0], (float)q[1], (float)q[2]};
    int reach = (int)ceilf(radius * g->inv);
    if (reach < 1) reach = 1;
    int cx = (int)floorf((qf[0] - g->minx) * g->inv), cy = (int)floorf((qf[1] - g->miny) * g->inv), cz = (int)floorf((qf[2] - g->minz) * g->inv);
    int best = -1; float bd = r2;
    for (int z = cz - reach; z <= cz + reach; ++z) {
        if (z < 0 || z >= g->nz) continue;
        for (int y = cy - reach; y <= cy + reach; ++y) {
            if (y < 0 || y >= g->ny) continue;
            int x0 = cx - reach, x1 = cx + reach;
            if (x1 < 0 || x0 >= g->nx) continue;
            if (x0 < 0) x0 = 0;
            if (x1 >= g->nx) x1 = g->nx - 1;
            int c0 = (z * g->ny + y) * g->nx + x0, c1 = (z * g->ny + y) * g->nx + x1;
            for (int s = g->start[c0]; s < g->start[c1 + 1]; ++s) {
                int j = g->order[s];
                float d2 = dist2f(qf, pts + 3 * j);
                if (d2 < bd || (d2 == bd && best >= 0 && j < best)) { bd = d2; best = j; }
            }
        }
    }
    *d2out = bd;
    return best;
}

/* evaluate_registration: fitness = inliers / n_src, rmse over inliers */
void oracle_evaluate(const float* src, int ns, const float* tgt, int nt, const double T[16], double max_dist, double* fitness,
                     double* rmse) {
    grid_t g;
    grid_build(&g, tgt, nt, (float)max_dist);
    const float r2 = (float)(max_dist * max_dist);
    int64_t inl = 0; double err2 = 0;
#pragma omp parallel for reduction(+ : inl, err2) schedule(static)
    for (int i = 0; i < ns; ++i) {
        double sp[3] = {src[3 * i], src[3 * i + 1], src[3 * i + 2]}, p[3];
        xform(T, sp, p);
        float d2;
        int j = nn_within(&g, tgt, p, r2, (float)max_dist, &d2);
        if (j >= 0) { ++inl; err2 += (double)d2; }
    }
    *fitness = ns > 0 ? (double)inl / ns : 0.0;
    *rmse = inl > 0 ? sqrt(err2 / (double)inl) : 0.0;
    grid_free(&g);
}

/* Diagnostics for tests/test_open3d_fp64.py (the discrete decisions of the fp32 rule, to be counted against an independent fp64
 * restatement): the hybrid neighbour SETS (idx [n][max_nn], -1 padded, in (d2, idx) order; cnt [n] = all points inside the radius when
 * max_nn >= n) and the accepted correspondence of every transformed source point (corr [ns] = target index or -1). */
void oracle_hybrid_sets(const float* pts, int n, double radius, int max_nn, int32_t* idx, int32_t* cnt) {
    grid_t g;
    grid_build(&g, pts, n, (float)radius);
    const float r2 = (float)(radius * radius);
#pragma omp parallel
    {
        nb_t* buf = (nb_t*)malloc(sizeof(nb_t) * (size_t)(n > 0 ? n : 1));
#pragma omp for schedule(dynamic, 64)
        for (int i = 0; i < n; ++i) {
            int k = hybrid_search(&g, pts, pts + 3 * i, r2, max_nn, (float)radius, buf, n);
            cnt[i] = k;
            for (int t = 0; t < max_nn; ++t) idx[(size_t)i * max_nn + t] = t < k ? buf[t].idx : -1;
        }
        free(buf);
    }
    grid_free(&g);
}

void oracle_correspondences(const float* src, int ns, const float* tgt, int nt, const double T[16], double max_dist, int32_t* corr) {
    grid_t g;
    grid_build(&g, tgt, nt, (float)max_dist);
    const float r2 = (float)(max_dist * max_dist);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < ns; ++i) {
        double sp[3] = {src[3 * i], src[3 * i + 1], src[3 * i + 2]}, p[3];
        xform(T, sp, p);
        float d2;
        corr[i] = nn_within(&g, tgt, p, r2, (float)max_dist, &d2);
    }
    grid_free(&g);
}

/* ------------------------------------------------------------------------------------------ */
/* coloured ICP                                                                                 */
/* ------------------------------------------------------------------------------------------ */
/* 6x6 solve with partial pivoting; returns 0 on singular */
static int solve6(double A[6][6], double b[6], double x[6]) {
    int n = 6;
    double M[6][7];
    for (int i = 0; i < n; ++i) { for (int j = 0; j < n; ++j) M[i][j] = A[i][j]; M[i][6] = b[i]; }
    for (int c = 0; c < n; ++c) {
        int piv = c; double mx = fabs(M[c][c]);
        for (int r = c + 1; r < n; ++r) if (fabs(M[r][c]) > mx) { mx = fabs(M[r][c]); piv = r; }
        if (mx == 0.0 || !isfinite(mx)) return 0;
        if (piv != c) for (int j = 0; j <= n; ++j) { double t = M[c][j]; M[c][j] = M[piv][j]; M[piv][j] = t; }
        for (int r = c + 1; r < n; ++r) {
            double f = M[r][c] / M[c][c];
            for (int j = c; j <= n; ++j) M[r][j] -= f * M[c][j];
        }
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = M[i][6];
        for (int j = i + 1; j < n; ++j) s -= M[i][j] * x[j];
        x[i] = s / M[i][i];
    }
    return 1;
}

/* TransformVector6dToMatrix4d: R = Rz(x2) Ry(x1) Rx(x0), t = x[3..5] */
static void vec6_to_T(const double x[6], double T[16]) {
    double ca = cos(x[0]), sa = sin(x[0]), cb = cos(x[1]), sb = sin(x[1]), cg = cos(x[2]), sg = sin(x[2]);
    T[0] = cg * cb; T[1] = cg * sb * sa - sg * ca; T[2] = cg * sb * ca + sg * sa; T[3] = x[3];
    T[4] = sg * cb; T[5] = sg * sb * sa + cg * ca; T[6] = sg * sb * ca - cg * sa; T[7] = x[4];
    T[8] = -sb;     T[9] = cb * sa;                T[10] = cb * ca;               T[11] = x[5];
    T[12] = T[13] = T[14] = 0; T[15] = 1;
}

static void matmul4(const double A[16], const double B[16], double C[16]) {
    double t[16];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        double s = 0;
        for (int k = 0; k < 4; ++k) s += A[4 * i + k] * B[4 * k + j];
        t[4 * i + j] = s;
    }
    memcpy(C, t, sizeof(t));
}

/* InitializePointCloudForColoredICP: per target point colour gradient, hybrid (2 * max_dist, 30 nn) */
void oracle_color_gradient(const float* pts, const float* normals, const float* intensity, int n, double radius, int max_nn,
                           float* grad /* n x 3 */) {
    grid_t g;
    grid_build(&g, pts, n, (float)radius);
    const float r2 = (float)(radius * radius);
#pragma omp parallel
    {
        nb_t* buf = (nb_t*)malloc(sizeof(nb_t) * (size_t)(n > 0 ? n : 1));
#pragma omp for schedule(dynamic, 64)
        for (int i = 0; i < n; ++i) {
            int k = hybrid_search(&g, pts, pts + 3 * i, r2, max_nn, (float)radius, buf, n);
            double gx[3] = {0, 0, 0};
            if (k >= 4) {
                double vt[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]}, nt[3] = {normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]};
                double it = intensity[i];
                double AtA[3][3] = {{0}}, Atb[3] = {0, 0, 0};
                int rows = 0;
                for (int t = 0; t < k; ++t) {
                    int j = buf[t].idx;
                    if (j == i) continue;
                    double va[3] = {pts[3 * j], pts[3 * j + 1], pts[3 * j + 2]};
                    double dd[3] = {va[0] - vt[0], va[1] - vt[1], va[2] - vt[2]};
                    double pr = dot3(dd, nt);
                    double a[3] = {va[0] - pr * nt[0] - vt[0], va[1] - pr * nt[1] - vt[1], va[2] - pr * nt[2] - vt[2]};
                    double b = (double)intensity[j] - it;
                    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) AtA[r][c] += a[r] * a[c]; Atb[r] += a[r] * b; }
                    ++rows;
                }
                /* orthogonality constraint row: (k - 1) * nt, rhs 0 */
                double wgt = (double)(k - 1);
                for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) AtA[r][c] += wgt * nt[r] * wgt * nt[c];
                (void)rows;
                /* solve the 3x3 SPD system (Cramer) */
                double det = AtA[0][0] * (AtA[1][1] * AtA[2][2] - AtA[1][2] * AtA[2][1]) - AtA[0][1] * (AtA[1][0] * AtA[2][2] - AtA[1][2] * AtA[2][0]) +
                             AtA[0][2] * (AtA[1][0] * AtA[2][1] - AtA[1][1] * AtA[2][0]);
                if (det != 0.0 && isfinite(det)) {
                    gx[0] = (Atb[0] * (AtA[1][1] * AtA[2][2] - AtA[1][2] * AtA[2][1]) - AtA[0][1] * (Atb[1] * AtA[2][2] - AtA[1][2] * Atb[2]) +
                             AtA[0][2] * (Atb[1] * AtA[2][1] - AtA[1][1] * Atb[2])) / det;
                    gx[1] = (AtA[0][0] * (Atb[1] * AtA[2][2] - AtA[1][2] * Atb[2]) - Atb[0] * (AtA[1][0] * AtA[2][2] - AtA[1][2] * AtA[2][0]) +
                             AtA[0][2] * (AtA[1][0] * Atb[2] - Atb[1] * AtA[2][0])) / det;
                    gx[2] = (AtA[0][0] * (AtA[1][1] * Atb[2] - Atb[1] * AtA[2][1]) - AtA[0][1] * (AtA[1][0] * Atb[2] - Atb[1] * AtA[2][0]) +
                             Atb[0] * (AtA[1][0] * AtA[2][1] - AtA[1][1] * AtA[2][0])) / det;
                }
            }
            grad[3 * i] = (float)gx[0]; grad[3 * i + 1] = (float)gx[1]; grad[3 * i + 2] = (float)gx[2];
        }
        free(buf);
    }
    grid_free(&g);
}

/* RegistrationICP loop shared by the coloured and the point-to-point estimators.
 * colored != 0: TransformationEstimationForColoredICP (needs normals, intensities, gradients)
 * colored == 0: TransformationEstimationPointToPoint (Kabsch on the correspondences)           */
void oracle_icp(const float* src, const float* src_int, int ns, const float* tgt, const float* tgt_nrm, const float* tgt_int,
                const float* tgt_grad, int nt, double max_dist, const double T_init[16], int colored, double lambda_geometric,
                int max_iter, double rel_fitness, double rel_rmse, double T_out[16], double* fitness_out, double* rmse_out,
                int* iters_out) {
    grid_t g;
    grid_build(&g, tgt, nt, (float)max_dist);
    const float r2 = (float)(max_dist * max_dist);
    double T[16];
    memcpy(T, T_init, sizeof(T));
    int* cj = (int*)malloc(sizeof(int) * (size_t)(ns > 0 ? ns : 1));
    double* P = (double*)malloc(sizeof(double) * 3 * (size_t)(ns > 0 ? ns : 1));
    double fitness = 0, rmse = 0;
    const double sl_g = sqrt(lambda_geometric), sl_p = sqrt(1.0 - lambda_geometric);
    int it = 0;
    for (;; ++it) {
        /* correspondences with the current transformation */
        int64_t cnt = 0; double err2 = 0;
        for (int i = 0; i < ns; ++i) {
            double sp[3] = {src[3 * i], src[3 * i + 1], src[3 * i + 2]};
            xform(T, sp, P + 3 * i);
            float d2;
            cj[i] = nn_within(&g, tgt, P + 3 * i, r2, (float)max_dist, &d2);
            if (cj[i] >= 0) { ++cnt; err2 += (double)d2; }
        }
        double nf = ns > 0 ? (double)cnt / ns : 0.0, nr = cnt > 0 ? sqrt(err2 / (double)cnt) : 0.0;
        if (it > 0 && fabs(fitness - nf) < rel_fitness && fabs(rmse - nr) < rel_rmse) { fitness = nf; rmse = nr; break; }
        fitness = nf; rmse = nr;
        if (it >= max_iter) break;
        /* update */
        double U[16];
        for (int i = 0; i < 16; ++i) U[i] = (i % 5) == 0;
        if (cnt > 0) {
            if (colored) {
                double JTJ[6][6] = {{0}}, JTr[6] = {0};
                for (int i = 0; i < ns; ++i) {
                    int j = cj[i];
                    if (j < 0) continue;
                    const double* vs = P + 3 * i;
                    double vt[3] = {tgt[3 * j], tgt[3 * j + 1], tgt[3 * j + 2]}, nt_[3] = {tgt_nrm[3 * j], tgt_nrm[3 * j + 1], tgt_nrm[3 * j + 2]};
                    double J[6], r;
                    double c[3];
                    cross3(vs, nt_, c);
                    double dv[3] = {vs[0] - vt[0], vs[1] - vt[1], vs[2] - vt[2]};
                    J[0] = sl_g * c[0]; J[1] = sl_g * c[1]; J[2] = sl_g * c[2]; J[3] = sl_g * nt_[0]; J[4] = sl_g * nt_[1]; J[5] = sl_g * nt_[2];
                    r = sl_g * dot3(dv, nt_);
                    for (int a = 0; a < 6; ++a) { for (int b = 0; b < 6; ++b) JTJ[a][b] += J[a] * J[b]; JTr[a] += J[a] * r; }
                    /* photometric term */
                    double pr = dot3(dv, nt_);
                    double vp[3] = {vs[0] - pr * nt_[0], vs[1] - pr * nt_[1], vs[2] - pr * nt_[2]};
                    double is = src_int[i], itg = tgt_int[j];
                    double dit[3] = {tgt_grad[3 * j], tgt_grad[3 * j + 1], tgt_grad[3 * j + 2]};
                    double dp[3] = {vp[0] - vt[0], vp[1] - vt[1], vp[2] - vt[2]};
                    double is0 = dot3(dit, dp) + itg;
                    /* ditM = -dit^T M,  M = I - nt nt^T */
                    double dn = dot3(dit, nt_);
                    double ditM[3] = {-(dit[0] - dn * nt_[0]), -(dit[1] - dn * nt_[1]), -(dit[2] - dn * nt_[2])};
                    cross3(vs, ditM, c);
                    J[0] = sl_p * c[0]; J[1] = sl_p * c[1]; J[2] = sl_p * c[2]; J[3] = sl_p * ditM[0]; J[4] = sl_p * ditM[1]; J[5] = sl_p * ditM[2];
                    r = sl_p * (is - is0);
                    for (int a = 0; a < 6; ++a) { for (int b = 0; b < 6; ++b) JTJ[a][b] += J[a] * J[b]; JTr[a] += J[a] * r; }
                }
                double nb[6], x[6];
                for (int a = 0; a < 6; ++a) nb[a] = -JTr[a];
                if (solve6(JTJ, nb, x)) vec6_to_T(x, U);
            } else {
                double* S = (double*)malloc(sizeof(double) * 3 * (size_t)cnt);
                double* D = (double*)malloc(sizeof(double) * 3 * (size_t)cnt);
                int64_t m = 0;
                for (int i = 0; i < ns; ++i) {
                    int j = cj[i];
                    if (j < 0) continue;
                    for (int a = 0; a < 3; ++a) { S[3 * m + a] = P[3 * i + a]; D[3 * m + a] = tgt[3 * j + a]; }
                    ++m;
                }
                kabsch(S, D, (int)m, U);
                free(S); free(D);
            }
        }
        matmul4(U, T, T);
    }
    memcpy(T_out, T, sizeof(T));
    *fitness_out = fitness; *rmse_out = rmse; *iters_out = it;
    free(cj); free(P);
    grid_free(&g);
}

/* ------------------------------------------------------------------------------------------ */
/* register_point_clouds (utils/fpfh_register.py:100-143)                                       */
/* have_colors == 0 models the exception path: point-to-point ICP from identity                 */
/* ------------------------------------------------------------------------------------------ */
/* src_raw / tgt_raw (NULL = src / tgt): the same clouds before the caller centred them.  Normals, FPFH and   */
/* colour gradients are translation invariant; the product evaluates them in the frame the clouds are stored  */
/* in (so that per-instance results can be kept), and this restatement follows it: features from the raw      */
/* coordinates, RANSAC and ICP between the centred clouds (the reference computes all of it on the centred    */
/* clouds in double precision, where the two orders agree to ~1e-16).                                          */
void oracle_register_raw(const float* src, const float* src_raw, const float* src_int, int ns, const float* tgt, const float* tgt_raw,
                         const float* tgt_int, int nt, double voxel, double global_factor, double local_factor, int have_colors,
                         uint64_t seed, uint32_t job_id, int ransac_max_iter, double T_out[16], double* rmse_out, double* fitness_out,
                         double* T_ransac_out, int64_t* ransac_stats) {
    if (!src_raw) src_raw = src;
    if (!tgt_raw) tgt_raw = tgt;
    double I[16];
    for (int i = 0; i < 16; ++i) I[i] = (i % 5) == 0;
    int iters;
    if (!have_colors || ns <= 0 || nt <= 0) {
        oracle_icp(src, NULL, ns, tgt, NULL, NULL, NULL, nt, voxel * local_factor, I, 0, 0.968, 30, 1e-6, 1e-6, T_out, fitness_out,
                   rmse_out, &iters);
        if (T_ransac_out) memcpy(T_ransac_out, I, sizeof(I));
        return;
    }
    float* ns_n = (float*)malloc(sizeof(float) * 3 * (size_t)ns);
    float* nt_n = (float*)malloc(sizeof(float) * 3 * (size_t)nt);
    float* fs = (float*)malloc(sizeof(float) * 33 * (size_t)ns);
    float* ft = (float*)malloc(sizeof(float) * 33 * (size_t)nt);
    oracle_normals(src_raw, ns, voxel * 2, 30, ns_n);
    oracle_normals(tgt_raw, nt, voxel * 2, 30, nt_n);
    oracle_fpfh(src_raw, ns_n, ns, voxel * 5, 100, fs);
    oracle_fpfh(tgt_raw, nt_n, nt, voxel * 5, 100, ft);
    int32_t* corr = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)ns);
    int nc = oracle_feature_match(fs, ns, ft, nt, 1, 3, corr);
    double Tr[16];
    oracle_ransac(src, tgt, corr, nc, voxel * global_factor, seed, job_id, ransac_max_iter, 0.99, 0.9, Tr, ransac_stats);
    if (T_ransac_out) memcpy(T_ransac_out, Tr, sizeof(Tr));
    float* grad = (float*)malloc(sizeof(float) * 3 * (size_t)nt);
    double md = voxel * local_factor;
    oracle_color_gradient(tgt_raw, nt_n, tgt_int, nt, md * 2.0, 30, grad);
    oracle_icp(src, src_int, ns, tgt, nt_n, tgt_int, grad, nt, md, Tr, 1, 0.968, 30, 1e-6, 1e-6, T_out, fitness_out, rmse_out, &iters);
    free(ns_n); free(nt_n); free(fs); free(ft); free(corr); free(grad);
}

void oracle_register(const float* src, const float* src_int, int ns, const float* tgt, const float* tgt_int, int nt,
                     double voxel, double global_factor, double local_factor, int have_colors, uint64_t seed, uint32_t job_id,
                     int ransac_max_iter, double T_out[16], double* rmse_out, double* fitness_out, double* T_ransac_out,
                     int64_t* ransac_stats) {
    oracle_register_raw(src, NULL, src_int, ns, tgt, NULL, tgt_int, nt, voxel, global_factor, local_factor, have_colors, seed, job_id,
                        ransac_max_iter, T_out, rmse_out, fitness_out, T_ransac_out, ransac_stats);
}
