"""ibl_assign_candidates (the assignment search on per-row two-ended candidate lists, csrc/assign.cpp) against ibl_assign_batch on
the full rows: wherever the library reports a frame as proved (`exact`), the assignment list must be identical; a frame it cannot
prove is redone on the full rows by the engine, so `exact == False` is allowed, never a wrong list.  The candidate lists are
produced here by a numpy restatement of the device selection (csrc/topk.hip: the k_hi largest and k_lo smallest fp16 entries under
(value, lower index first)), per memory shard, as the all-gather delivers them."""
import numpy as np
import pytest

from ibloc_amd.assign import assign_batch, assign_candidates
from oracle import simvolume_oracle as so


def select_np(aug_row: np.ndarray, base: int, k_hi: int, k_lo: int):
    """one shard's list for one row: (values fp16, global indices); whole row when it has at most k_hi + k_lo columns"""
    v = aug_row.astype(np.float32)
    n = len(v)
    order = np.lexsort((np.arange(n), -v))                   # value desc, index asc
    if n <= k_hi + k_lo:
        return aug_row[order], base + order
    hi = order[:k_hi]
    lo = np.lexsort((np.arange(n), v))[:k_lo]                # value asc, index asc
    sel = np.concatenate([hi, lo])
    return aug_row[sel], base + sel


def candidates(aug: np.ndarray, k_hi: int, k_lo: int, shards: int):
    """aug (Q, M + 1) fp16 -> padded (Q, S * shards) lists + counts, shard by contiguous instance range"""
    Q, M = aug.shape[0], aug.shape[1] - 1
    bounds = [M * s // shards for s in range(shards + 1)]
    S = (k_hi + k_lo) * shards
    val = np.zeros((Q, S), dtype=np.float16)
    idx = np.full((Q, S), -1, dtype=np.int32)
    cnt = np.zeros(Q, dtype=np.int32)
    for i in range(Q):
        vs, js = [], []
        for s in range(shards):
            v, j = select_np(aug[i, bounds[s]:bounds[s + 1]], bounds[s], k_hi, k_lo)
            vs.append(v)
            js.append(j)
        v, j = np.concatenate(vs), np.concatenate(js)
        val[i, :len(v)], idx[i, :len(v)], cnt[i] = v, j, len(v)
    return val, idx, cnt


def _sims(rng, Q, M, kind):
    if kind == "reid":            # what the bench produces: a dense cloud of look-alikes and one true match per row
        s = rng.normal(0.84, 0.04, size=(Q, M))
        for i in range(Q):
            s[i, rng.integers(0, M)] = rng.uniform(0.96, 0.99)
        return np.clip(s, -1, 1)
    if kind == "peaked":
        s = rng.normal(0, 0.05, size=(Q, M))
        for i in range(Q):
            s[i, rng.integers(0, M)] = rng.uniform(0.5, 0.99)
        return s
    if kind == "uniform":
        return rng.uniform(-1, 1, size=(Q, M))
    if kind == "neg":
        return -np.abs(rng.uniform(0.01, 1, size=(Q, M)))
    if kind == "ties":
        return rng.choice([0.25, 0.5, -0.5, 0.125, 1.0, 0.0], size=(Q, M))
    if kind == "coarse":
        return np.round(rng.uniform(-1, 1, size=(Q, M)) * 64) / 64
    if kind == "const":
        return np.full((Q, M), 0.5)
    raise ValueError(kind)


@pytest.mark.parametrize("shards", [1, 2, 5])
@pytest.mark.parametrize("kind", ["reid", "peaked", "uniform", "neg", "ties", "coarse", "const"])
@pytest.mark.parametrize("Q,M", [(1, 900), (2, 1200), (3, 700), (5, 800), (7, 2000), (7, 300), (3, 150)])
def test_candidate_search_equals_full_search_where_proved(kind, Q, M, shards):
    import zlib
    rng = np.random.default_rng(zlib.crc32(f"{kind}-{Q}-{M}-{shards}".encode()))
    k_hi, k_lo = 192, 32
    frames = 4
    augs = np.stack([so._augment(_sims(rng, Q, M, kind).astype(np.float32)) for _ in range(frames)])
    full = assign_batch(augs, [Q] * frames, 4)
    vals, idxs, cnts = zip(*[candidates(a, k_hi, k_lo, shards) for a in augs])
    got, exact = assign_candidates(np.concatenate(vals), np.concatenate(idxs), np.concatenate(cnts), np.arange(frames) * Q, [Q] * frames,
                                   M, k_hi, k_lo, 4)
    for f in range(frames):
        if exact[f]:
            assert got[f] == full[f], (kind, Q, M, shards, f)
    complete = -(-M // shards) <= k_hi + k_lo              # every shard returns its whole range
    if kind in ("reid", "peaked") or complete:
        assert exact.all(), "realistic similarity rows (and complete rows) must be proved without the fall-back"
    if kind == "const" and not complete:
        assert not exact.any(), "all-equal rows tie at the threshold: the proof must refuse them"


def test_candidate_search_against_the_reference_golden_cases():
    """the reference's own SimVolume outputs (tests/golden/simvolume_golden.json): small M, every row complete"""
    import json
    import os
    cases = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "simvolume_golden.json")))["cases"]
    for c in cases:
        aug = so._augment(np.asarray(c["sims"], dtype=np.float32))
        Q, M = aug.shape[0], aug.shape[1] - 1
        if Q > M:                      # the engine truncates the matched detections to M first (object_memory.py:918-920)
            continue
        v, j, n = candidates(aug, 192, 32, 1)
        got, exact = assign_candidates(v, j, n, [0], [Q], M, 192, 32, c["num_per_length"])
        assert exact[0] and got[0] == c["expected"], c["name"]


def test_small_candidate_lists_fall_back_instead_of_guessing():
    """k_hi far below the k = 16 Q cells a sub-volume returns: nothing can be proved, nothing may be wrong"""
    rng = np.random.default_rng(5)
    Q, M = 7, 1500
    aug = so._augment(_sims(rng, Q, M, "reid").astype(np.float32))
    full = assign_batch(aug[None], [Q], 4)[0]
    for k_hi, k_lo in [(8, 2), (40, 4), (120, 8)]:
        v, j, n = candidates(aug, k_hi, k_lo, 1)
        got, exact = assign_candidates(v, j, n, [0], [Q], M, k_hi, k_lo, 4)
        assert (not exact[0]) or got[0] == full


@pytest.mark.parametrize("M", [225, 226, 230, 300, 1000])
def test_duplicate_columns_of_overlapping_ends_are_counted_once(M):
    """a row whose hi and lo thresholds fall into ONE tie class (constant rows; zero-padded query rows of the sharded match): both ends of
    the device selection break ties by lower index, so the two lists share columns (`keep[i].size()` counted them twice, which
    could hide a hole).  The host must count columns once: never a wrong list, and constant rows just above the list size are not
    mistaken for complete rows."""
    rng = np.random.default_rng(M)
    k_hi, k_lo, Q = 192, 32, 3
    sims = np.full((Q, M), 0.5, dtype=np.float32)
    sims[1] = rng.choice([0.5, 0.25], size=M)                   # two levels: both thresholds inside the 0.5 / 0.25 classes
    sims[2, :5] = [0.9, 0.8, 0.7, -0.3, -0.6]                   # a few distinct entries + a constant remainder
    aug = so._augment(sims)
    v, j, n = candidates(aug, k_hi, k_lo, 1)
    assert any(len(set(j[i, :n[i]].tolist())) < n[i] for i in range(Q)) or M <= k_hi + k_lo      # the lists really share columns
    full = assign_batch(aug[None], [Q], 4)[0]
    got, exact = assign_candidates(v, j, n, [0], [Q], M, k_hi, k_lo, 4)
    assert (not exact[0]) or got[0] == full
