"""Drop-in for the reference's `utils/similarity_volume.py` (SimVolume) on the MI355X build.

Same constructor / method names and the same returned assignment lists
(/root/reference/utils/similarity_volume.py:12-18, 102-164, 213-270), but the (M+1)^3 float16
volumes are never materialised: the ranked assignments come from the exact search behind the
C-ABI entry point `ibl_assign_batch` (csrc/assign.cpp)."""
import itertools

import numpy as np

from ibloc_amd.assign import assign_batch

MAX_LITERAL_CELLS = 1 << 25      # the literal (M + 1)^dim volumes are only built up to this many cells (32 M fp16 = 64 MB)


def _chained_product(rows):
    """left-to-right outer products of float16 rows, every product rounded to float16 (np.einsum on float16 operands, :122-124)"""
    vol = np.einsum('i,j', rows[0], rows[1])
    for r in rows[2:]:
        vol = np.einsum('...i,j', vol, r)
    return vol


def _admissible(M, dim):
    """True where the reference's mask is 0 (:126-154): assigned coordinates (!= M) pairwise distinct, a full permutation of the M memory
    objects over the dim axes exists (M >= dim), and -- `mask[[-1] * (dim + 1)] = -inf` is a fancy index on AXIS 0 -- the first
    coordinate assigned"""
    shape = (M + 1,) * dim
    if M < dim:
        return np.zeros(shape, dtype=bool)
    idx = np.indices(shape, sparse=True)
    ok = np.broadcast_to(idx[0] != M, shape).copy()
    for a in range(dim):
        for b in range(a + 1, dim):
            ok &= ~((idx[a] == idx[b]) & (idx[a] != M))
    return ok


def _masked_volume(rows):
    rows = np.asarray(rows, dtype=np.float16)
    dim, M = rows.shape[0], rows.shape[1] - 1
    if float(M + 1) ** dim > MAX_LITERAL_CELLS:
        raise MemoryError(f"a literal ({M + 1})^{dim} similarity volume is not built (limit {MAX_LITERAL_CELLS} cells); "
                          "fast_construct_volume(min(n_detected, 3)) + get_top_indices_from_subvolumes() never builds one")
    vol = _chained_product(rows)
    rep = np.where(_admissible(M, dim), vol, np.float16(-np.inf)).astype(np.float16)
    rep[np.isnan(rep)] = -np.inf
    return vol, rep


class SimVolume():
    def __init__(self, cosine_similarities) -> None:
        cosine_similarities = np.asarray(cosine_similarities)
        # e x (m + 1): the reference's augmentation with the "unassigned" column, cast to float16
        aug = np.ones((cosine_similarities.shape[0], cosine_similarities.shape[1] + 1), dtype=np.float16)
        aug[:, :-1] = cosine_similarities
        self.aug = aug
        self.subvolume_size = None
        self.chosen_objects = None
        # kept for attribute compatibility; the volumes themselves are never built
        self.subvolumes = []

    def fast_construct_volume(self, subvolume_size):
        """Records which detection combinations the search ranges over (reference :102-164)."""
        if self.aug.shape[0] == 1:
            self.chosen_objects = [[0]]
            self.subvolume_size = 1
            return
        assert self.aug.shape[0] >= subvolume_size
        self.subvolume_size = subvolume_size
        self.chosen_objects = list(itertools.combinations(range(self.aug.shape[0]), subvolume_size))
        if subvolume_size != min(self.aug.shape[0], 3):
            # not the value localise() uses (object_memory.py:980): the literal masked volumes, as the reference builds them (small
            # memories only; the exact search of ibl_assign_batch covers the size localise() asks for at any M)
            self.subvolumes = [_masked_volume(self.aug[list(ch)])[1] for ch in self.chosen_objects]

    def construct_volume(self):
        """The whole e-dimensional volume and its masked copy (reference :30-100; unused by localise): (volume, rep_volume), or `aug`
        itself for fewer than two detections.  Literal arrays -- small memories only (MAX_LITERAL_CELLS)."""
        if self.aug.shape[0] < 2:
            print("Too few detected embs")
            return self.aug
        return _masked_volume(self.aug)

    def construct_volume_choose_e(self, chosen_e):
        """Unmasked volume over the chosen detection rows (reference :169-180)"""
        assert len(chosen_e) <= self.aug.shape[0]
        rows = self.aug[list(chosen_e)]
        if float(rows.shape[1]) ** len(chosen_e) > MAX_LITERAL_CELLS:
            raise MemoryError("literal similarity volume too large")
        return _chained_product(rows)

    def get_top_indices(self, vol, k):
        """k times (argmax cell, value), each found cell overwritten with -inf IN `vol` (reference :182-194: first flat index wins ties)"""
        top_k = []
        for _ in range(k):
            ind = np.unravel_index(np.argmax(vol, axis=None), vol.shape)
            top_k.append([ind, vol[ind]])
            vol[ind] = -np.inf
        return top_k

    def conv_coords_to_pairs(self, vol, coords):
        """[(cell, cost)] -> [[[detection, memory object], ...], cost] without the unassigned coordinates; empty assignments dropped
        (reference :196-209)"""
        unassigned = vol.shape[0] - 1
        out = []
        for cell, cost in coords:
            pairs = [[i, c] for i, c in enumerate(cell) if c != unassigned]
            if pairs:
                out.append([pairs, cost])
        return out

    def get_top_indices_from_subvolumes(self, num_per_length=3):
        if self.chosen_objects is None:
            raise RuntimeError("call fast_construct_volume() first")
        Q = self.aug.shape[0]
        if self.subvolume_size in (1, min(Q, 3)):
            return assign_batch(self.aug[None], np.array([Q], dtype=np.int32), num_per_length)[0]
        # literal sub-volumes of another size (reference :213-270): k cells per sub-volume by repeated argmax, unassigned coordinates
        # dropped, first occurrence of an assignment keeps its cost, per length l the l best by a stable descending sort
        k = num_per_length * Q * 4
        unassigned = self.subvolumes[0].shape[0] - 1
        seen, ranked = [], []
        for chosen, sub in zip(self.chosen_objects, self.subvolumes):
            for cell, cost in self.get_top_indices(sub, k):
                pairs = [[int(i), int(c)] for i, c in zip(chosen, cell) if c != unassigned]
                if pairs and pairs not in seen:
                    seen.append(pairs)
                    ranked.append([pairs, cost])
        out = []
        for length in range(1, Q + 1):
            same = [r for r in ranked if len(r[0]) == length]
            out += [r[0] for r in sorted(same, key=lambda r: r[-1], reverse=True)[:max(1, length)]]
        dedup = []
        for a in out:
            if a not in dedup:
                dedup.append(a)
        return dedup
