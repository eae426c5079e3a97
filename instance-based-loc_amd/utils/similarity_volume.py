"""Drop-in for the reference's `utils/similarity_volume.py` (SimVolume) on the MI355X build.

Same constructor / method names and the same returned assignment lists
(/root/reference/utils/similarity_volume.py:12-18, 102-164, 213-270), but the (M+1)^3 float16
volumes are never materialised: the ranked assignments come from the exact search behind the
C-ABI entry point `ibl_assign_batch` (csrc/assign.cpp)."""
import itertools

import numpy as np

from ibloc_amd.assign import assign_batch


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
        if subvolume_size != min(self.aug.shape[0], 3):
            raise NotImplementedError(
                "only subvolume_size == min(n_detected, 3) (the value localise() uses, "
                "object_memory.py:980) is supported")
        self.subvolume_size = subvolume_size
        self.chosen_objects = list(itertools.combinations(range(self.aug.shape[0]), subvolume_size))

    def get_top_indices_from_subvolumes(self, num_per_length=3):
        if self.chosen_objects is None:
            raise RuntimeError("call fast_construct_volume() first")
        Q = self.aug.shape[0]
        return assign_batch(self.aug[None], np.array([Q], dtype=np.int32), num_per_length)[0]
