"""-m gpu: the device half of the assignment search -- per-row two-ended candidate selection (csrc/topk.hip) -- bit for bit against
a numpy restatement, and the engine with candidates against the engine on full rows (the reference's own data flow,
object_memory/object_memory.py:933-936 -> :974-982)."""
import numpy as np
import pytest
import torch

from tests.test_assign_candidates import select_np

pytestmark = pytest.mark.gpu


def _select_dev(aug: np.ndarray, n_cols: int, k_hi: int, k_lo: int, base: int):
    from ibloc_amd import _lib
    a = torch.from_numpy(aug.view(np.uint16).astype(np.int32)).to(torch.int16).cuda().view(torch.float16).contiguous()
    R, S = a.shape[0], k_hi + k_lo
    val = torch.zeros((R, S), dtype=torch.float16, device="cuda")
    idx = torch.full((R, S), -1, dtype=torch.int32, device="cuda")
    cnt = torch.zeros((R, 2), dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib.ibl_topk_select(a.data_ptr(), R, a.shape[1], n_cols, k_hi, k_lo, base, val.data_ptr(), idx.data_ptr(), cnt.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream), "ibl_topk_select")
    torch.cuda.synchronize()
    return val.cpu().numpy(), idx.cpu().numpy(), cnt.cpu().numpy()


def _rows(rng, R, M, kind):
    if kind == "normal":
        return rng.normal(0.8, 0.05, size=(R, M))
    if kind == "signed":
        return rng.uniform(-1, 1, size=(R, M))
    if kind == "ties":
        return rng.choice([0.25, 0.5, -0.5, 0.125, 1.0, 0.0, -0.0], size=(R, M))
    if kind == "const":
        return np.full((R, M), 0.75)
    if kind == "coarse":
        return np.round(rng.uniform(-1, 1, size=(R, M)) * 32) / 32
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["normal", "signed", "ties", "const", "coarse"])
@pytest.mark.parametrize("M,k_hi,k_lo", [(10000, 192, 32), (50000, 192, 32), (300, 192, 32), (224, 192, 32), (225, 192, 32), (5, 192, 32),
                                         (1000, 16, 0), (4097, 256, 256), (777, 1, 1)])
def test_select_bit_exact_vs_numpy(kind, M, k_hi, k_lo):
    import zlib
    rng = np.random.default_rng(zlib.crc32(f"{kind}-{M}-{k_hi}".encode()))
    R = 9
    aug = np.ones((R, M + 1), dtype=np.float16)
    aug[:, :M] = _rows(rng, R, M, kind).astype(np.float16)
    val, idx, cnt = _select_dev(aug, M, k_hi, k_lo, base=1000)
    for r in range(R):
        ev, ei = select_np(aug[r, :M], 1000, k_hi, k_lo)
        n = cnt[r].sum()
        assert n == len(ev), (r, cnt[r], len(ev))
        assert np.array_equal(idx[r, :n], ei), (kind, M, r)
        assert np.array_equal(val[r, :n].astype(np.float32), ev.astype(np.float32))       # value equality (-0 is returned as +0)
        if M > k_hi + k_lo:
            assert tuple(cnt[r]) == (k_hi, k_lo)
        else:
            assert tuple(cnt[r]) == (M, 0)


def test_match_topk_lists_come_from_the_aug_rows():
    from ibloc_amd import match
    g = torch.Generator().manual_seed(3)
    mem = match.normalize_rows(torch.randn(4000 * 3, 256, generator=g).cuda())
    det = match.normalize_rows(torch.randn(37, 256, generator=g).cuda())
    off = (torch.arange(4001, dtype=torch.int32) * 3).cuda()
    _, aug_ref = match.closest_similarity(det, mem, off, want_sims=False, want_aug=True)
    val, idx, cnt, aug = match.match_topk(det, mem, off, 192, 32, index_base=50)
    assert torch.equal(aug, aug_ref)
    a = aug.cpu().numpy()
    v, j, c = val.cpu().numpy(), idx.cpu().numpy(), cnt.cpu().numpy()
    for r in range(37):
        ev, ei = select_np(a[r, :4000], 50, 192, 32)
        assert np.array_equal(j[r], ei) and np.array_equal(v[r], ev)
        assert tuple(c[r]) == (192, 32)


@pytest.mark.parametrize("M,dup", [(300, False), (3000, False), (3000, True)])
def test_engine_on_candidates_equals_engine_on_full_rows(M, dup):
    """embedding-only memory (register=False): the assignment lists of the candidate path (device selection + host proof) must be
    those of the full-row search; with duplicated stored embeddings every row ties massively and the proof must refuse, not guess"""
    from ibloc_amd.engine import LocaliseEngine, MemoryShard
    from ibloc_amd.registration import RegContext
    rng = np.random.default_rng(M + dup)
    D, E, F = 64, 2, 12
    base = rng.normal(size=(M, D))
    if dup:
        base = np.repeat(base[:M // 500], 500, axis=0)[:M]
    emb = (base[:, None, :] + (0 if dup else 1) * rng.normal(0, 0.2, size=(M, E, D))).astype(np.float32)
    ctx = RegContext(64 << 20)
    eng = LocaliseEngine(MemoryShard(ctx, list(emb)))
    q = rng.integers(1, 8, size=F)
    ids = rng.integers(0, M, size=int(q.sum()))
    det = (base[ids] + rng.normal(0, 0.15, size=(len(ids), D))).astype(np.float32)
    a = eng.localise_batch(None, q, det_emb=det, register=False)
    eng.use_candidates = False
    b = eng.localise_batch(None, q, det_emb=det, register=False)
    assert [r.assignments for r in a] == [r.assignments for r in b]
    if dup:
        assert eng.stats["fallback_frames"] > 0
    elif M > 224:
        assert eng.stats["fallback_frames"] == 0, eng.stats
    ctx.close()


def test_sharded_exchange_through_the_library_rccl_communicator_world_1():
    """the collectives of the sharded stage A (include/ibloc.h ibl_comm_* / ibl_allgather_topk / ibl_allreduce_*) on a one-rank RCCL
    communicator: the engine takes the exchange path (query gather, candidate gather, flag all-reduce) and must return the lists of
    the un-sharded engine; the all-reduce(MIN) of the sharded evaluation is the identity"""
    from ibloc_amd.engine import LocaliseEngine, MemoryShard
    from ibloc_amd.parallel import RcclComm, evaluate_sharded
    from ibloc_amd.registration import RegContext
    comm = RcclComm.single()
    rng = np.random.default_rng(3)
    M, E, D, F = 2000, 2, 64, 10
    base = rng.normal(size=(M, D))
    emb = (base[:, None, :] + rng.normal(0, 0.2, size=(M, E, D))).astype(np.float32)
    q = rng.integers(1, 8, size=F)
    ids = rng.integers(0, M, size=int(q.sum()))
    det = (base[ids] + rng.normal(0, 0.15, size=(len(ids), D))).astype(np.float32)
    ctx = RegContext(64 << 20)
    plain = LocaliseEngine(MemoryShard(ctx, list(emb))).localise_batch(None, q, det_emb=det, register=False)
    eng = LocaliseEngine(MemoryShard(ctx, list(emb), shard=(0, 1)), comm=comm, rows_cap=80)
    assert eng.exchange is not None
    got = eng.localise_batch(None, q, det_emb=det, register=False)
    assert [r.assignments for r in got] == [r.assignments for r in plain]
    piped = list(eng.localise_stream([dict(det=None, q_per_frame=q, det_emb=det)] * 3, register=False))
    assert all([r.assignments for r in p] == [r.assignments for r in plain] for p in piped)
    d2 = torch.tensor([0.1, float("inf"), 0.3, 0.2], device="cuda")
    fit, rmse = evaluate_sharded(d2, [2, 2], comm=comm)
    assert np.allclose(fit, [0.5, 1.0]) and np.allclose(rmse, [np.sqrt(0.1), np.sqrt(0.25)])
    comm.close()
    ctx.close()


@pytest.mark.parametrize("M", [226, 300, 10000])
def test_constant_rows_through_the_device_selection_and_the_host_proof(M):
    """constant / two-level rows: the device's two ends share columns (both break ties by lower index); the host counts columns once and
    either proves the frame or reports it for the full-row search -- never a wrong list"""
    from ibloc_amd.assign import assign_batch, assign_candidates
    rng = np.random.default_rng(M)
    Q, k_hi, k_lo = 3, 192, 32
    aug = np.ones((Q, M + 1), dtype=np.float16)
    aug[0, :M] = 0.5
    aug[1, :M] = rng.choice([0.5, 0.25], size=M)
    aug[2, :M] = 0.5
    aug[2, :5] = [0.9, 0.8, 0.7, -0.3, -0.6]
    val, idx, cnt = _select_dev(aug, M, k_hi, k_lo, base=0)
    n = cnt.sum(axis=1).astype(np.int32)
    assert len(set(idx[0, :n[0]].tolist())) < n[0]                 # the ends of a constant row overlap
    full = assign_batch(aug[None], [Q], 4)[0]
    got, exact = assign_candidates(val, idx, n, [0], [Q], M, k_hi, k_lo, 4)
    assert (not exact[0]) or got[0] == full
