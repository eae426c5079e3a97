"""ORACLE (test infrastructure, never imported by the product path).

CPU restatement of the reference's similarity-volume assignment search
(`utils/similarity_volume.py`), used as the checker for the HIP/C-ABI
`ibl_assign_batch` entry point.  Pinned against golden vectors produced by the reference's own
module in the build container (tests/golden/simvolume_golden.json, generator
tools/gen_golden_simvolume.py) -- see tests/test_oracle_simvolume.py.

Semantics restated (file:line under /root/reference):
  * utils/similarity_volume.py:13-18   aug = [sims | 1] cast to float16.
  * :102-110  Q == 1: the single 1-D "volume" is aug[0] with the unassigned slot set to -inf.
  * :114-124  one sub-volume per combination (index order) of `subvolume_size` detections; values
              are left-to-right chained float16 outer products (each product rounded to fp16).
  * :126-154  mask: a cell is admissible iff its assigned coordinates (those != M) are pairwise
              distinct, a full permutation exists (M >= dim), and -- because `mask[[-1]*(d+1)]`
              is a fancy index on axis 0 -- the FIRST coordinate is assigned.
  * :213-225  k = num_per_length * Q * 4 repeated argmax per sub-volume (first flat index wins
              ties; once only -inf is left argmax keeps returning flat index 0).
  * :227-270  drop unassigned pairs, de-duplicate keeping the first occurrence and its cost, then
              for every length l = 1..Q keep the l best by a stable descending sort.
"""
import itertools

import numpy as np


def _augment(sims):
    sims = np.asarray(sims)
    aug = np.ones((sims.shape[0], sims.shape[1] + 1), dtype=np.float16)
    aug[:, :-1] = sims
    return aug


def _admissible_mask(M, dim):
    """Boolean (M+1)^dim array, True where the reference's mask is 0 (similarity_volume.py:126-154)."""
    shape = (M + 1,) * dim
    if M < dim:
        return np.zeros(shape, dtype=bool)
    idx = np.indices(shape)
    ok = np.ones(shape, dtype=bool)
    for p in range(dim):
        for q in range(p + 1, dim):
            both_assigned = (idx[p] != M) & (idx[q] != M)
            ok &= ~(both_assigned & (idx[p] == idx[q]))
    ok &= idx[0] != M          # `mask[[-1, ...]] = -inf` hits the whole last slab of axis 0
    return ok


def build_subvolumes(sims, subvolume_size=None):
    """Returns (aug, chosen_objects, subvolumes) like SimVolume.fast_construct_volume."""
    aug = _augment(sims)
    Q = aug.shape[0]
    M = aug.shape[1] - 1
    if Q == 1:
        vol = aug[0].copy()
        vol[-1] = -np.inf
        return aug, [[0]], [vol]
    dim = min(Q, 3) if subvolume_size is None else subvolume_size
    assert Q >= dim
    chosen_objects = list(itertools.combinations(range(Q), dim))
    ok = _admissible_mask(M, dim)
    subvolumes = []
    for chosen in chosen_objects:
        sub = aug[list(chosen)]
        volume = np.einsum('i,j', sub[0], sub[1])
        for row in sub[2:]:
            volume = np.einsum('...i,j', volume, row)
        assert volume.dtype == np.float16
        rep = np.where(ok, volume, np.float16(-np.inf)).astype(np.float16)
        rep[np.isnan(rep)] = -np.inf
        subvolumes.append(rep)
    return aug, chosen_objects, subvolumes


def _topk_literal(subvol, k):
    """similarity_volume.py:218-225 verbatim semantics: k times argmax, overwrite with -inf."""
    vol = subvol.copy()
    out = []
    for _ in range(k):
        ind = np.unravel_index(np.argmax(vol, axis=None), vol.shape)
        out.append((tuple(int(x) for x in ind), vol[ind]))
        vol[ind] = -np.inf
    return out


def _topk_fast(subvol, k):
    """Same list as _topk_literal via one stable sort (value desc, flat index asc)."""
    flat = subvol.ravel()
    finite = np.flatnonzero(flat > -np.inf)
    vals = flat[finite].astype(np.float32)
    order = np.argsort(-vals, kind='stable')[:k]
    out = [(tuple(int(x) for x in np.unravel_index(finite[o], subvol.shape)), flat[finite[o]]) for o in order]
    zero = tuple(0 for _ in subvol.shape)
    while len(out) < k:
        out.append((zero, np.float16(-np.inf)))
    return out


def topk_cells(sims, num_per_length=4, subvolume_size=None, literal=False):
    """Per sub-volume the ordered list of (cell, cost) the reference extracts."""
    aug, chosen_objects, subvolumes = build_subvolumes(sims, subvolume_size)
    k = num_per_length * aug.shape[0] * 4
    f = _topk_literal if literal else _topk_fast
    return chosen_objects, [f(sv, k) for sv in subvolumes], subvolumes[0].shape[0] - 1


def postprocess(chosen_objects, per_subvolume_topk, unassigned_ind, Q):
    """similarity_volume.py:227-270."""
    assns = []
    all_filtered = []
    for chosen, cells in zip(chosen_objects, per_subvolume_topk):
        for ind, cost in cells:
            filtered = [[int(i), int(c)] for i, c in zip(chosen, ind) if c != unassigned_ind]
            if len(filtered) == 0:
                continue
            if filtered not in assns:
                assns.append(filtered)
                all_filtered.append((filtered, cost))
    picked = []
    for length in range(1, Q + 1):
        correct = [f for f in all_filtered if len(f[0]) == length]
        correct = sorted(correct, key=lambda x: x[-1], reverse=True)[:max(1, length)]
        picked += correct
    out = []
    for a, _ in picked:
        if a not in out:
            out.append(a)
    return out


def simvolume_assignments(sims, num_per_length=4, subvolume_size=None, literal=False):
    """== SimVolume(sims); fast_construct_volume(min(Q,3)); get_top_indices_from_subvolumes(npl)."""
    sims = np.asarray(sims)
    chosen_objects, cells, unassigned = topk_cells(sims, num_per_length, subvolume_size, literal)
    return postprocess(chosen_objects, cells, unassigned, sims.shape[0])
