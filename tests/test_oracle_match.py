"""Pins the fixed-order C oracle of normalise + closest similarity against a literal numpy transcript
of the reference's lines (object_memory.py:922-936); tolerance 2e-6 absolute (fp32 summation order)."""
import numpy as np

from oracle import match_oracle as mo


def test_oracle_close_to_numpy_transcript():
    rng = np.random.default_rng(3)
    D, M = 768, 37
    counts = rng.integers(1, 6, size=M)
    mem_inst = [rng.normal(size=(c, D)).astype(np.float32) for c in counts]
    det = rng.normal(size=(5, D)).astype(np.float32)
    ref = mo.closest_similarity_numpy(det.copy(), mem_inst)
    mem = mo.normalize_rows(np.concatenate(mem_inst))
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    got = mo.closest_similarity(mo.normalize_rows(det), mem, off)
    assert np.max(np.abs(got - ref)) < 2e-6
    n = np.linalg.norm(mem.astype(np.float64), axis=1)
    assert np.max(np.abs(n - 1)) < 1e-6
