// Exact similarity-volume assignment search without materialising the volume.
//
// Replaces (same outputs, bit-exact indices) the reference's
//   SimVolume(sims).fast_construct_volume(min(Q,3)); .get_top_indices_from_subvolumes(npl)
//   -- /root/reference/utils/similarity_volume.py:13-18, 102-164, 213-270, called from
//   object_memory/object_memory.py:974-982.
//
// The reference builds C(Q,dim) float16 volumes of (M+1)^dim cells and extracts the top
// k = npl*Q*4 cells of each by repeated argmax.  That is O(M^3) memory/time and cannot run for
// M >~ 100.  Here the same ordered list of k cells per sub-volume is produced by
//   phase 1  value search on a pruned candidate set (provably contains every cell whose value is
//            strictly above the k-th largest value T, and determines T exactly), using the
//            coordinate-wise monotonicity of the chained fp16 product;
//   phase 2  a flat-index-order scan for the cells tied at T (np.argmax breaks ties by lowest
//            flat index), bounded by exact per-slab / per-row admissible maxima.
// Small volumes are enumerated outright.  Post-processing restates similarity_volume.py:227-270.
//
// This is host code (integer / fp16 index logic on Q<=7 rows); SURVEY §8(d) "Assign" row.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "ibl_common.h"

namespace {

// ---- IEEE binary16 helpers (values are kept as float holding half-representable numbers) ----
inline float half_bits_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t f;
    if (exp == 0) {
        if (man == 0) {
            f = sign;
        } else {  // subnormal
            int e = -1;
            do { man <<= 1; ++e; } while (!(man & 0x400u));
            man &= 0x3FFu;
            f = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        f = sign | 0x7F800000u | (man << 13);
    } else {
        f = sign | ((exp + 112u) << 23) | (man << 13);
    }
    float out;
    std::memcpy(&out, &f, 4);
    return out;
}

// round-to-nearest-even float -> half, returned as float (the numpy float32->float16 cast)
inline float round_half(float x) {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    uint32_t sign = u & 0x80000000u;
    uint32_t a = u & 0x7FFFFFFFu;
    if (a >= 0x7F800000u) return x;                       // inf / nan
    if (a >= 0x477FF000u) {                               // >= 65520 -> inf
        uint32_t r = sign | 0x7F800000u;
        float o; std::memcpy(&o, &r, 4); return o;
    }
    if (a < 0x38800000u) {                                // |x| < 2^-14: subnormal half (step 2^-24)
        float ax;
        std::memcpy(&ax, &a, 4);
        // exact: scale, round-to-nearest-even via the 2^23 trick, unscale
        float t = ax * 16777216.0f;                       // * 2^24  (exact)
        t = (t + 8388608.0f) - 8388608.0f;                // RNE to integer (t < 2^10)
        t = t * (1.0f / 16777216.0f);
        uint32_t r;
        std::memcpy(&r, &t, 4);
        r |= sign;
        float o; std::memcpy(&o, &r, 4); return o;
    }
    // normal half: keep 10 mantissa bits
    uint32_t lsb = (a >> 13) & 1u;
    a += 0xFFFu + lsb;
    a &= ~0x1FFFu;
    uint32_t r = sign | a;
    float o; std::memcpy(&o, &r, 4); return o;
}

inline float hmul(float a, float b) { return round_half(a * b); }  // a*b exact in float for halves

struct Cell {
    float v;
    int64_t flat;
};
inline bool cell_before(const Cell& x, const Cell& y) {     // value desc, flat asc
    if (x.v != y.v) return x.v > y.v;
    return x.flat < y.flat;
}

struct RowView {
    const float* v;   // assigned entries [0, n)
    int n;            // number of assigned entries usable for this coordinate (M, or 0 for a dummy row)
    bool has_unassigned;  // index M with value 1.0 allowed (coordinates 1 and 2)
};

struct SortedRow {   // per detection row: indices of assigned entries, best-first, both directions
    std::vector<int> desc, asc;   // truncated to `keep`
    // assigned entries by |value| descending (ties by index).  Only a prefix is sorted up front: the tie scan asks for "all entries
    // with |value| >= thr", which is a short prefix for every threshold that occurs with real similarities; the rare deeper request
    // sorts the rest on demand (a full sort of every row was the one O(M log M) term of the search: 6 ms per frame at M = 10 000).
    mutable std::vector<int> absord;
    mutable bool abs_full = false;
    float maxabs = 0.0f;          // max |value| over the assigned entries
    int M = 0;

    void sort_abs_all(const float* v) const {
        absord.resize(M);
        for (int j = 0; j < M; ++j) absord[j] = j;
        std::sort(absord.begin(), absord.end(), [v](int p, int q) {
            float ap = std::fabs(v[p]), aq = std::fabs(v[q]);
            if (ap != aq) return ap > aq;
            return p < q;
        });
        abs_full = true;
    }
    // number of leading entries with |v| >= thr (sorting deeper first when the sorted prefix does not reach below thr)
    int abs_prefix(const float* v, float thr) const {
        if (!abs_full && !absord.empty() && std::fabs(v[absord.back()]) >= thr) sort_abs_all(v);
        int lo = 0, hi = (int)absord.size();          // first position with |v| < thr
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (std::fabs(v[absord[mid]]) >= thr) lo = mid + 1; else hi = mid;
        }
        return lo;
    }
};

constexpr float NEG_INF = -std::numeric_limits<float>::infinity();

// candidate list in one direction over assigned entries (+ the unassigned slot), at most `lim`
static void dir_list(const float* v, const std::vector<int>& sorted, bool desc, bool with_unassigned, int M,
                     int lim, std::vector<std::pair<float, int>>& out) {
    out.clear();
    bool placed = !with_unassigned;
    for (size_t i = 0; i < sorted.size() && (int)out.size() < lim; ++i) {
        float x = v[sorted[i]];
        if (!placed && (desc ? (1.0f > x) : (1.0f < x))) {
            out.emplace_back(1.0f, M);
            placed = true;
            if ((int)out.size() >= lim) break;
        }
        out.emplace_back(x, sorted[i]);
    }
    if (!placed && (int)out.size() < lim) out.emplace_back(1.0f, M);
}

struct SubvolumeSearch {
    int M;
    int k;
    const float* r0; const float* r1; const float* r2;   // r2 == nullptr for dim 2
    const SortedRow* s0; const SortedRow* s1; const SortedRow* s2;
    int dim;

    // lists
    std::vector<std::pair<float, int>> d1, a1, d2, a2;        // coordinate 1/2 candidates incl. unassigned
    std::vector<std::pair<float, int>> e1, e2;                // extremes (first 3 of each end)
    std::vector<int> seen_stamp;                              // level-2 visited marks (stamped)
    int stamp = 0;

    inline int64_t flat(int a, int b, int c) const {
        return dim == 3 ? ((int64_t)a * (M + 1) + b) * (M + 1) + c : (int64_t)a * (M + 1) + b;
    }
    inline float val1(int b) const { return b == M ? 1.0f : r1[b]; }
    inline float val2(int c) const { return (c == M || dim == 2) ? 1.0f : r2[c]; }
    inline bool ok_b(int a, int b) const { return b == M || b != a; }
    inline bool ok_c(int a, int b, int c) const { return c == M || (c != a && c != b); }

    // exact admissible maximum of a slab
    float head_a(int a) const {
        float x = r0[a];
        float best = NEG_INF;
        for (const auto& pb : e1) {
            if (!ok_b(a, pb.second)) continue;
            float P = hmul(x, pb.first);
            for (const auto& pc : e2) {
                if (!ok_c(a, pb.second, pc.second)) continue;
                float f = hmul(P, pc.first);
                if (f > best) best = f;
            }
        }
        return best;
    }
    float head_ab(int a, int b, float P) const {
        float best = NEG_INF;
        for (const auto& pc : e2) {
            if (!ok_c(a, b, pc.second)) continue;
            float f = hmul(P, pc.first);
            if (f > best) best = f;
        }
        return best;
    }

    void brute(std::vector<Cell>& out) const {
        std::vector<Cell> cells;
        int nb = M + 1, nc = dim == 3 ? M + 1 : 1;
        if (M >= dim) {
            for (int a = 0; a < M; ++a) {
                for (int b = 0; b < nb; ++b) {
                    if (!ok_b(a, b)) continue;
                    float P = hmul(r0[a], val1(b));
                    for (int ci = 0; ci < nc; ++ci) {
                        int c = dim == 3 ? ci : M;
                        if (dim == 3 && !ok_c(a, b, c)) continue;
                        float f = dim == 3 ? hmul(P, val2(c)) : P;
                        if (std::isnan(f) || f == NEG_INF) continue;   // NaN -> -inf in the reference
                        cells.push_back({f, flat(a, b, c)});
                    }
                }
            }
        }
        size_t take = std::min<size_t>(k, cells.size());
        std::partial_sort(cells.begin(), cells.begin() + take, cells.end(), cell_before);
        out.assign(cells.begin(), cells.begin() + take);
        while ((int)out.size() < k) out.push_back({NEG_INF, 0});   // argmax of all -inf -> flat 0
    }

    void run(std::vector<Cell>& out) {
        int64_t ncell = (int64_t)(M + 1) * (M + 1) * (dim == 3 ? (M + 1) : 1);
        if (ncell <= 32768 || M < 8) { brute(out); return; }

        const int K = k + 2;
        dir_list(r1, s1->desc, true, true, M, K + 1, d1);
        dir_list(r1, s1->asc, false, true, M, K + 1, a1);
        if (dim == 3) {
            dir_list(r2, s2->desc, true, true, M, K + 1, d2);
            dir_list(r2, s2->asc, false, true, M, K + 1, a2);
        } else {
            d2.assign(1, {1.0f, M});
            a2 = d2;
        }
        e1.clear(); e2.clear();
        for (int i = 0; i < 3 && i < (int)d1.size(); ++i) e1.push_back(d1[i]);
        for (int i = 0; i < 3 && i < (int)a1.size(); ++i) e1.push_back(a1[i]);
        for (int i = 0; i < 3 && i < (int)d2.size(); ++i) e2.push_back(d2[i]);
        for (int i = 0; i < 3 && i < (int)a2.size(); ++i) e2.push_back(a2[i]);

        // ---------------- phase 1: T and the strict set ----------------
        // level 1: slabs
        std::vector<int> ca;
        {
            int lim = std::min<int>(K, (int)s0->desc.size());
            ca.assign(s0->desc.begin(), s0->desc.begin() + lim);
            int lim2 = std::min<int>(K, (int)s0->asc.size());
            for (int i = 0; i < lim2; ++i) ca.push_back(s0->asc[i]);
            std::sort(ca.begin(), ca.end());
            ca.erase(std::unique(ca.begin(), ca.end()), ca.end());
        }
        std::vector<std::pair<float, int>> heads;
        heads.reserve(ca.size());
        for (int a : ca) heads.emplace_back(head_a(a), a);
        if ((int)heads.size() > k) {
            std::nth_element(heads.begin(), heads.begin() + k, heads.end(),
                             [](const auto& x, const auto& y) { return x.first > y.first; });
            heads.resize(k);
        }
        // level 2: (a, b) pairs.  For a fixed slab the upper bound ub(P) = max_c h(P * z_c) (collisions
        // ignored) is V-shaped along the value-sorted b lists, so each list is walked from its best end
        // and abandoned at the first ub below tau, the current k-th best exact pair head.
        struct Pair { float head; float P; int a; int b; };
        std::vector<Pair> pairs;                       // min-heap on head, size <= k
        auto heap_cmp = [](const Pair& x, const Pair& y) { return x.head > y.head; };
        float tau = NEG_INF;
        std::sort(heads.begin(), heads.end(), [](const auto& x, const auto& y) { return x.first > y.first; });
        const float ztop = d2.empty() ? 1.0f : d2[0].first;
        const float zbot = a2.empty() ? 1.0f : a2[0].first;
        if ((int)seen_stamp.size() != M + 1) { seen_stamp.assign(M + 1, -1); stamp = 0; }
        for (const auto& h : heads) {
            if ((int)pairs.size() >= k && h.first < tau) break;    // no pair of this or later slabs can enter
            const int a = h.second;
            const float x = r0[a];
            ++stamp;
            for (int pass = 0; pass < 2; ++pass) {
                const auto& lst = pass == 0 ? d1 : a1;
                for (const auto& pb : lst) {
                    const int b = pb.second;
                    if (seen_stamp[b] == stamp) continue;
                    const float P = hmul(x, pb.first);
                    const float ub = std::max(hmul(P, ztop), hmul(P, zbot));
                    if ((int)pairs.size() >= k && ub < tau) break;
                    seen_stamp[b] = stamp;
                    if (!ok_b(a, b)) continue;
                    const float hd = head_ab(a, b, P);
                    if ((int)pairs.size() < k) {
                        pairs.push_back({hd, P, a, b});
                        std::push_heap(pairs.begin(), pairs.end(), heap_cmp);
                        if ((int)pairs.size() == k) tau = pairs.front().head;
                    } else if (hd > tau) {
                        std::pop_heap(pairs.begin(), pairs.end(), heap_cmp);
                        pairs.back() = {hd, P, a, b};
                        std::push_heap(pairs.begin(), pairs.end(), heap_cmp);
                        tau = pairs.front().head;
                    }
                }
            }
        }
        // level 3: cells.  Pairs in descending head order; along a pair's direction list the values are
        // non-increasing, so the walk stops at the first value below tau3 (current k-th best cell).
        std::sort(pairs.begin(), pairs.end(), [](const Pair& x, const Pair& y) { return x.head > y.head; });
        std::vector<Cell> cells;                          // min-heap on value, size <= k
        auto cell_heap_cmp = [](const Cell& x, const Cell& y) { return x.v > y.v; };
        float tau3 = NEG_INF;
        for (const auto& pr : pairs) {
            if ((int)cells.size() >= k && pr.head < tau3) break;
            const auto& lst = (pr.P < 0.0f) ? a2 : d2;
            for (const auto& pc : lst) {
                const float f = hmul(pr.P, pc.first);
                if ((int)cells.size() >= k && f < tau3) break;
                if (!ok_c(pr.a, pr.b, pc.second)) continue;
                if ((int)cells.size() < k) {
                    cells.push_back({f, flat(pr.a, pr.b, pc.second)});
                    std::push_heap(cells.begin(), cells.end(), cell_heap_cmp);
                    if ((int)cells.size() == k) tau3 = cells.front().v;
                } else if (f > tau3) {
                    std::pop_heap(cells.begin(), cells.end(), cell_heap_cmp);
                    cells.back() = {f, flat(pr.a, pr.b, pc.second)};
                    std::push_heap(cells.begin(), cells.end(), cell_heap_cmp);
                    tau3 = cells.front().v;
                }
            }
        }
        if ((int)cells.size() < k) { brute(out); return; }   // cannot happen for M >= 8; stay exact anyway
        std::sort(cells.begin(), cells.end(), cell_before);
        const float T = cells[k - 1].v;
        out.clear();
        for (int i = 0; i < k && cells[i].v > T; ++i) out.push_back(cells[i]);
        int n_tie = k - (int)out.size();

        // ---------------- phase 2: ties at T in flat-index order ----------------
        const float B1 = std::max(1.0f, s1->maxabs), B2 = dim == 3 ? std::max(1.0f, s2->maxabs) : 1.0f;
        if (T > 0.0f) {
            // |round_half(p)| <= p * 1.001 + 3e-8 for finite p, so a cell can only reach T when every
            // partial magnitude clears the inverted bound.  Candidates come from the |value|-sorted rows
            // (prefix by binary search), are re-sorted by index (flat order) and then checked exactly.
            auto inv = [](float t) { return (t - 3.0e-8f) / 1.001f; };
            const float tB = inv(inv(T) / B2);             // needed |x * y|
            std::vector<int> as, bs, cs;
            {
                const float thr = tB / B1 * 0.999f;
                int n = s0->abs_prefix(r0, thr);
                as.assign(s0->absord.begin(), s0->absord.begin() + n);
                std::sort(as.begin(), as.end());
            }
            for (size_t ia = 0; ia < as.size() && n_tie > 0; ++ia) {
                const int a = as[ia];
                const float x = r0[a];
                if (x == 0.0f || head_a(a) < T) continue;
                {
                    const float thr = tB / std::fabs(x) * 0.999f;
                    int n = s1->abs_prefix(r1, thr);
                    bs.assign(s1->absord.begin(), s1->absord.begin() + n);
                    std::sort(bs.begin(), bs.end());
                    if (1.0f >= thr) bs.push_back(M);
                }
                for (size_t ib = 0; ib < bs.size() && n_tie > 0; ++ib) {
                    const int b = bs[ib];
                    if (!ok_b(a, b)) continue;
                    const float P = hmul(x, val1(b));
                    if (P == 0.0f || head_ab(a, b, P) < T) continue;
                    if (dim == 2) {
                        if (P == T) { out.push_back({T, flat(a, b, M)}); --n_tie; }
                        continue;
                    }
                    const float thr = inv(T) / std::fabs(P) * 0.999f;
                    int n = s2->abs_prefix(r2, thr);
                    cs.assign(s2->absord.begin(), s2->absord.begin() + n);
                    std::sort(cs.begin(), cs.end());
                    if (1.0f >= thr) cs.push_back(M);
                    for (size_t ic = 0; ic < cs.size() && n_tie > 0; ++ic) {
                        const int c = cs[ic];
                        if (!ok_c(a, b, c)) continue;
                        if (hmul(P, val2(c)) == T) { out.push_back({T, flat(a, b, c)}); --n_tie; }
                    }
                }
            }
        } else {
            for (int a = 0; a < M && n_tie > 0; ++a) {
                float x = r0[a];
                if (head_a(a) < T) continue;
                for (int b = 0; b <= M && n_tie > 0; ++b) {
                    if (!ok_b(a, b)) continue;
                    float P = hmul(x, val1(b));
                    if (head_ab(a, b, P) < T) continue;
                    if (dim == 2) {
                        if (P == T) { out.push_back({T, flat(a, b, M)}); --n_tie; }
                        continue;
                    }
                    for (int c = 0; c <= M && n_tie > 0; ++c) {
                        if (!ok_c(a, b, c)) continue;
                        if (hmul(P, val2(c)) == T) { out.push_back({T, flat(a, b, c)}); --n_tie; }
                    }
                }
            }
        }
        // n_tie == 0 here because at least k cells have value >= T.
    }
};

struct Assn {
    int len;
    int32_t pair[3][2];
    float cost;
};
inline bool same_assn(const Assn& x, const Assn& y) {
    if (x.len != y.len) return false;
    for (int i = 0; i < x.len; ++i)
        if (x.pair[i][0] != y.pair[i][0] || x.pair[i][1] != y.pair[i][1]) return false;
    return true;
}

// Proof obligations of a search that ran on per-row candidate columns only (ibl_assign_candidates): every dropped entry of row i has a
// value in [lo[i], hi[i]]; rmin / rmax are the row's true extremes.  A sub-volume's result is that of the full search when its k-th
// best value T exceeds (a) every value a cell with a dropped coordinate could take -- the chained fp16 product is monotone in each
// coordinate, so its maximum over the box [lo, hi] x (row ranges of the other coordinates) is attained at a corner -- and (b) zero,
// the value of the cells that touch a column another row contributed (those entries are filled with 0, see ibl_assign_candidates).
struct Verify {
    const float* lo; const float* hi; const float* rmin; const float* rmax;
    const uint8_t* complete;      // row has every column: nothing was dropped
    bool any_holes;               // some row lacks a column another row contributed
    bool exact;
};

static bool subvolume_proved(const Verify& vf, const int* chosen, int dim, float T) {
    bool all_complete = true;
    for (int d = 0; d < dim; ++d) all_complete = all_complete && vf.complete[chosen[d]];
    if (all_complete && !vf.any_holes) return true;
    if (!(T > NEG_INF)) return false;
    if (vf.any_holes && !(T > 0.0f)) return false;
    float lo[3], hi[3];
    for (int d = 0; d < dim; ++d) {
        lo[d] = vf.rmin[chosen[d]];
        hi[d] = vf.rmax[chosen[d]];
        if (d > 0) { lo[d] = std::min(lo[d], 1.0f); hi[d] = std::max(hi[d], 1.0f); }     // coordinates 1, 2 may be unassigned (1.0)
    }
    for (int t = 0; t < dim; ++t) {
        if (vf.complete[chosen[t]]) continue;
        float a[3][2];
        for (int d = 0; d < dim; ++d) { a[d][0] = lo[d]; a[d][1] = hi[d]; }
        a[t][0] = vf.lo[chosen[t]];
        a[t][1] = vf.hi[chosen[t]];
        for (int c = 0; c < (1 << dim); ++c) {
            float f = hmul(a[0][c & 1], a[1][(c >> 1) & 1]);
            if (dim == 3) f = hmul(f, a[2][(c >> 2) & 1]);
            if (!(f < T)) return false;            // also catches NaN
        }
    }
    return true;
}

// one frame; rows is [Q][M+1] floats holding half-representable values; colmap (or null) renames the M assigned columns in the output
static int assign_rows(const std::vector<float>& rows, int Q, int M, int npl, int32_t* out_assn, int32_t* out_len, int max_assn,
                       Verify* vf, const int32_t* colmap) {
    if (Q <= 0) return 0;
    const int k = npl * Q * 4;

    static thread_local std::vector<Assn> uniq;     // first occurrences, in order
    static thread_local std::vector<int> table;     // open-addressing index over `uniq` (<= 35*k entries)
    uniq.clear();
    table.assign(16384, -1);
    auto push = [&](const Assn& a) {
        uint64_t h = 0x9E3779B97F4A7C15ull * (uint64_t)(a.len + 1);
        for (int i = 0; i < a.len; ++i) {
            h ^= ((uint64_t)(uint32_t)a.pair[i][0] << 32) | (uint32_t)a.pair[i][1];
            h *= 0xFF51AFD7ED558CCDull;
            h ^= h >> 29;
        }
        if (uniq.size() * 2 >= table.size()) {           // grow + rehash (rare)
            table.assign(table.size() * 2, -1);
            for (size_t u = 0; u < uniq.size(); ++u) {
                uint64_t g = 0x9E3779B97F4A7C15ull * (uint64_t)(uniq[u].len + 1);
                for (int i = 0; i < uniq[u].len; ++i) {
                    g ^= ((uint64_t)(uint32_t)uniq[u].pair[i][0] << 32) | (uint32_t)uniq[u].pair[i][1];
                    g *= 0xFF51AFD7ED558CCDull;
                    g ^= g >> 29;
                }
                size_t s = g & (table.size() - 1);
                while (table[s] >= 0) s = (s + 1) & (table.size() - 1);
                table[s] = (int)u;
            }
        }
        size_t slot = h & (table.size() - 1);
        while (table[slot] >= 0) {
            if (same_assn(uniq[table[slot]], a)) return;
            slot = (slot + 1) & (table.size() - 1);
        }
        table[slot] = (int)uniq.size();
        uniq.push_back(a);
    };

    if (Q == 1) {
        // similarity_volume.py:105-110: 1-D volume, unassigned slot = -inf
        std::vector<Cell> cells;
        for (int j = 0; j < M; ++j) {
            float f = rows[j];
            if (std::isnan(f) || f == NEG_INF) continue;
            cells.push_back({f, j});
        }
        size_t take = std::min<size_t>(k, cells.size());
        std::partial_sort(cells.begin(), cells.begin() + take, cells.end(), cell_before);
        cells.resize(take);
        while ((int)cells.size() < k) cells.push_back({NEG_INF, 0});
        if (vf && !vf->complete[0] && !(cells[k - 1].v > vf->hi[0])) vf->exact = false;   // 1-D: the score is the entry itself
        for (const auto& c : cells) {
            if ((int)c.flat == M) continue;    // unassigned index
            Assn a; a.len = 1; a.pair[0][0] = 0; a.pair[0][1] = (int32_t)c.flat; a.cost = c.v;
            push(a);
        }
    } else {
        const int dim = std::min(Q, 3);
        const int keep = k + 4;
        static thread_local std::vector<SortedRow> sorted;
        static thread_local std::vector<int> idx;
        if ((int)sorted.size() < Q) sorted.resize(Q);
        for (int i = 0; i < Q; ++i) {
            const float* v = &rows[(size_t)i * (M + 1)];
            idx.resize(M);
            for (int j = 0; j < M; ++j) idx[j] = j;
            int lim = std::min(keep, M);
            std::partial_sort(idx.begin(), idx.begin() + lim, idx.end(), [v](int p, int q) {
                if (v[p] != v[q]) return v[p] > v[q];
                return p < q;
            });
            sorted[i].desc.assign(idx.begin(), idx.begin() + lim);
            std::partial_sort(idx.begin(), idx.begin() + lim, idx.end(), [v](int p, int q) {
                if (v[p] != v[q]) return v[p] < v[q];
                return p < q;
            });
            sorted[i].asc.assign(idx.begin(), idx.begin() + lim);
            const int lim_abs = std::min(M, 256);
            std::partial_sort(idx.begin(), idx.begin() + lim_abs, idx.end(), [v](int p, int q) {
                float ap = std::fabs(v[p]), aq = std::fabs(v[q]);
                if (ap != aq) return ap > aq;
                return p < q;
            });
            sorted[i].absord.assign(idx.begin(), idx.begin() + lim_abs);
            sorted[i].abs_full = lim_abs == M;
            sorted[i].M = M;
            sorted[i].maxabs = M > 0 ? std::fabs(v[sorted[i].absord[0]]) : 0.0f;
        }
        static thread_local std::vector<Cell> cells;
        static thread_local SubvolumeSearch s;
        int chosen[3];
        // combinations in index order (itertools.combinations)
        for (chosen[0] = 0; chosen[0] < Q; ++chosen[0])
            for (chosen[1] = chosen[0] + 1; chosen[1] < Q; ++chosen[1])
                for (chosen[2] = (dim == 3 ? chosen[1] + 1 : 0); chosen[2] < (dim == 3 ? Q : 1); ++chosen[2]) {
                    s.M = M; s.k = k; s.dim = dim;
                    s.r0 = &rows[(size_t)chosen[0] * (M + 1)];
                    s.r1 = &rows[(size_t)chosen[1] * (M + 1)];
                    s.r2 = dim == 3 ? &rows[(size_t)chosen[2] * (M + 1)] : nullptr;
                    s.s0 = &sorted[chosen[0]]; s.s1 = &sorted[chosen[1]];
                    s.s2 = dim == 3 ? &sorted[chosen[2]] : nullptr;
                    s.run(cells);
                    if (vf && vf->exact && !subvolume_proved(*vf, chosen, dim, cells.empty() ? NEG_INF : cells.back().v)) vf->exact = false;
                    for (const auto& c : cells) {
                        int ind[3];
                        int64_t fl = c.flat;
                        if (dim == 3) { ind[2] = (int)(fl % (M + 1)); fl /= (M + 1); }
                        ind[1] = (int)(fl % (M + 1)); fl /= (M + 1);
                        ind[0] = (int)fl;
                        Assn a; a.len = 0; a.cost = c.v;
                        for (int d = 0; d < dim; ++d) {
                            if (ind[d] == M) continue;
                            a.pair[a.len][0] = chosen[d];
                            a.pair[a.len][1] = ind[d];
                            ++a.len;
                        }
                        if (a.len == 0) continue;
                        push(a);
                    }
                }
    }

    // similarity_volume.py:247-255: per length, stable descending sort, keep `length` best
    int n_out = 0;
    for (int length = 1; length <= Q && length <= 3; ++length) {
        std::vector<const Assn*> sel;
        for (const auto& u : uniq) if (u.len == length) sel.push_back(&u);
        std::stable_sort(sel.begin(), sel.end(), [](const Assn* x, const Assn* y) { return x->cost > y->cost; });
        int take = std::min<int>(std::max(1, length), (int)sel.size());
        for (int i = 0; i < take; ++i) {
            if (n_out >= max_assn) return -1;
            out_len[n_out] = sel[i]->len;
            for (int p = 0; p < 3; ++p) {
                out_assn[(n_out * 3 + p) * 2 + 0] = p < sel[i]->len ? sel[i]->pair[p][0] : -1;
                out_assn[(n_out * 3 + p) * 2 + 1] = p < sel[i]->len ? (colmap ? colmap[sel[i]->pair[p][1]] : sel[i]->pair[p][1]) : -1;
            }
            ++n_out;
        }
    }
    return n_out;
}

static int assign_frame(const uint16_t* aug, int Q, int M, int npl, int32_t* out_assn, int32_t* out_len, int max_assn) {
    // per-thread scratch, reused across frames (fresh large allocations per frame serialise threads in mmap)
    static thread_local std::vector<float> rows;
    rows.resize((size_t)std::max(Q, 0) * (M + 1));
    for (int i = 0; i < Q; ++i)
        for (int j = 0; j <= M; ++j) rows[(size_t)i * (M + 1) + j] = half_bits_to_float(aug[(size_t)i * (M + 1) + j]);
    return assign_rows(rows, Q, M, npl, out_assn, out_len, max_assn, nullptr, nullptr);
}

// One frame of ibl_assign_candidates: per row n entries (value bits, global column), the union over the memory shards of each
// shard's k_hi largest and k_lo smallest entries under (value, lower column first).
static int assign_frame_candidates(const uint16_t* val, const int32_t* idx, const int32_t* cnt, int64_t stride, int Q, int M_total, int k_hi,
                                   int k_lo, int npl, int32_t* out_assn, int32_t* out_len, int max_assn, uint8_t* out_exact) {
    struct Ent { float v; int32_t c; };
    static thread_local std::vector<Ent> ents, keep[7];
    static thread_local std::vector<int32_t> U;
    static thread_local std::vector<float> rows;
    float lo[7], hi[7], rmin[7], rmax[7];
    uint8_t complete[7];
    U.clear();
    for (int i = 0; i < Q; ++i) {
        int n = cnt[i];
        ents.resize(n);
        for (int e = 0; e < n; ++e) ents[e] = {half_bits_to_float(val[i * stride + e]), idx[i * stride + e]};
        // A producer's two ends overlap when its thresholds fall into one tie class (a row of n_cols - S + 2 or more equal values: the
        // zero-padded rows every shard matches, degenerate embeddings): both ends break ties by lower column first, so the same column
        // can arrive twice.  Columns are counted once -- `complete` and the hole test below count columns, not list entries.
        std::sort(ents.begin(), ents.end(), [](const Ent& x, const Ent& y) { return x.c < y.c; });
        ents.erase(std::unique(ents.begin(), ents.end(), [](const Ent& x, const Ent& y) { return x.c == y.c; }), ents.end());
        n = (int)ents.size();
        auto desc = [](const Ent& x, const Ent& y) { return x.v != y.v ? x.v > y.v : x.c < y.c; };
        auto asc = [](const Ent& x, const Ent& y) { return x.v != y.v ? x.v < y.v : x.c < y.c; };
        keep[i].clear();
        complete[i] = n >= M_total || n == 0;        // (an empty list can only belong to an empty memory)
        if (complete[i] || n <= k_hi + k_lo) {
            // every column of the row (or a single shard's whole list): nothing to merge
            keep[i] = ents;
            lo[i] = hi[i] = 0.0f;
            if (!complete[i]) {                                   // a single producer's list: its own k_hi-th / k_lo-th entries bound the rest
                std::sort(ents.begin(), ents.end(), desc);
                hi[i] = ents[std::min(n, k_hi) - 1].v;
                lo[i] = k_lo > 0 ? ents[n - std::min(n, k_lo)].v : -std::numeric_limits<float>::infinity();
            }
        } else {
            std::partial_sort(ents.begin(), ents.begin() + k_hi, ents.end(), desc);
            keep[i].assign(ents.begin(), ents.begin() + k_hi);
            hi[i] = ents[k_hi - 1].v;
            // the low end among the remaining entries (an entry is never taken twice)
            std::partial_sort(ents.begin() + k_hi, ents.begin() + k_hi + k_lo, ents.end(), asc);
            keep[i].insert(keep[i].end(), ents.begin() + k_hi, ents.begin() + k_hi + k_lo);
            lo[i] = k_lo > 0 ? ents[k_hi + k_lo - 1].v : -std::numeric_limits<float>::infinity();
        }
        rmin[i] = std::numeric_limits<float>::infinity();
        rmax[i] = -rmin[i];
        for (const auto& e : keep[i]) { rmin[i] = std::min(rmin[i], e.v); rmax[i] = std::max(rmax[i], e.v); U.push_back(e.c); }
        if (keep[i].empty()) { rmin[i] = rmax[i] = 0.0f; }
    }
    std::sort(U.begin(), U.end());
    U.erase(std::unique(U.begin(), U.end()), U.end());
    const int Mp = (int)U.size();
    rows.assign((size_t)Q * (Mp + 1), 0.0f);
    bool holes = false;
    for (int i = 0; i < Q; ++i) {
        float* r = &rows[(size_t)i * (Mp + 1)];
        r[Mp] = 1.0f;
        for (const auto& e : keep[i]) r[std::lower_bound(U.begin(), U.end(), e.c) - U.begin()] = e.v;
        holes = holes || (int)keep[i].size() < Mp;
    }
    Verify vf{lo, hi, rmin, rmax, complete, holes, true};
    const int n = assign_rows(rows, Q, Mp, npl, out_assn, out_len, max_assn, &vf, U.data());
    *out_exact = vf.exact ? 1 : 0;
    return n;
}

}  // namespace

extern "C" int ibl_assign_candidates(const uint16_t* cand_val, const int32_t* cand_idx, const int32_t* cand_cnt, int64_t cand_stride,
                                     const int32_t* row_first, const int32_t* q_per_frame, int n_frames, int M_total, int k_hi, int k_lo,
                                     int num_per_length, int32_t* out_assn, int32_t* out_len, int32_t* out_count, uint8_t* out_exact,
                                     int max_assn, int n_threads) {
    if (!cand_val || !cand_idx || !cand_cnt || !row_first || !q_per_frame || !out_assn || !out_len || !out_count || !out_exact)
        return ibl_set_error(IBL_ERR_ARG, "ibl_assign_candidates: null pointer");
    if (n_frames < 0 || M_total < 0 || num_per_length <= 0 || max_assn < 6 || k_hi <= 0 || k_lo < 0 || cand_stride <= 0)
        return ibl_set_error(IBL_ERR_ARG, "ibl_assign_candidates: bad sizes (max_assn must be >= 6)");
    for (int f = 0; f < n_frames; ++f) {
        if (q_per_frame[f] < 0 || q_per_frame[f] > 7) return ibl_set_error(IBL_ERR_ARG, "ibl_assign_candidates: q_per_frame out of range");
        for (int i = 0; i < q_per_frame[f]; ++i) {
            const int c = cand_cnt[row_first[f] + i];
            if (c < 0 || c > cand_stride) return ibl_set_error(IBL_ERR_ARG, "ibl_assign_candidates: candidate count out of range");
        }
    }
    std::vector<int> status(n_frames, 0);
    auto work = [&](int f0, int f1) {
        for (int f = f0; f < f1; ++f) {
            const int64_t r0 = row_first[f];
            int n = assign_frame_candidates(cand_val + r0 * cand_stride, cand_idx + r0 * cand_stride, cand_cnt + r0, cand_stride,
                                            q_per_frame[f], M_total, k_hi, k_lo, num_per_length, out_assn + (size_t)f * max_assn * 6,
                                            out_len + (size_t)f * max_assn, max_assn, out_exact + f);
            status[f] = n;
            out_count[f] = n < 0 ? 0 : n;
        }
    };
    int nt = std::max(1, std::min(n_threads, n_frames));
    if (nt == 1) {
        work(0, n_frames);
    } else {
        std::vector<std::thread> th;
        int per = (n_frames + nt - 1) / nt;
        for (int t = 0; t < nt; ++t) {
            int f0 = t * per, f1 = std::min(n_frames, f0 + per);
            if (f0 < f1) th.emplace_back(work, f0, f1);
        }
        for (auto& t : th) t.join();
    }
    for (int f = 0; f < n_frames; ++f)
        if (status[f] < 0) return ibl_set_error(IBL_ERR_INTERNAL, "ibl_assign_candidates: output overflow");
    return IBL_OK;
}

extern "C" int ibl_assign_batch(const uint16_t* aug_half, const int32_t* q_per_frame, int n_frames, int q_stride,
                                int M, int num_per_length, int32_t* out_assn, int32_t* out_len,
                                int32_t* out_count, int max_assn, int n_threads) {
    if (!aug_half || !q_per_frame || !out_assn || !out_len || !out_count)
        return ibl_set_error(IBL_ERR_ARG, "ibl_assign_batch: null pointer");
    if (n_frames < 0 || M < 0 || num_per_length <= 0 || max_assn < 6 || q_stride <= 0)
        return ibl_set_error(IBL_ERR_ARG, "ibl_assign_batch: bad sizes (max_assn must be >= 6)");
    for (int f = 0; f < n_frames; ++f)
        if (q_per_frame[f] < 0 || q_per_frame[f] > q_stride)
            return ibl_set_error(IBL_ERR_ARG, "ibl_assign_batch: q_per_frame out of range");
    // finite-product precondition of the pruned search: |aug| <= 40 keeps every chained product finite
    std::vector<int> status(n_frames, 0);
    auto work = [&](int f0, int f1) {
        for (int f = f0; f < f1; ++f) {
            const uint16_t* aug = aug_half + (size_t)f * q_stride * (M + 1);
            int n = assign_frame(aug, q_per_frame[f], M, num_per_length, out_assn + (size_t)f * max_assn * 6,
                                 out_len + (size_t)f * max_assn, max_assn);
            status[f] = n;
            out_count[f] = n < 0 ? 0 : n;
        }
    };
    int nt = std::max(1, std::min(n_threads, n_frames));
    if (nt == 1) {
        work(0, n_frames);
    } else {
        std::vector<std::thread> th;
        int per = (n_frames + nt - 1) / nt;
        for (int t = 0; t < nt; ++t) {
            int f0 = t * per, f1 = std::min(n_frames, f0 + per);
            if (f0 < f1) th.emplace_back(work, f0, f1);
        }
        for (auto& t : th) t.join();
    }
    for (int f = 0; f < n_frames; ++f)
        if (status[f] < 0) return ibl_set_error(IBL_ERR_INTERNAL, "ibl_assign_batch: output overflow");
    return IBL_OK;
}
