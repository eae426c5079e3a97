"""-m gpu: what the fp16 operands of the HIP encoder cost DOWNSTREAM.  The same crops go through the HIP encoder and through the
fp32 oracle (torch fp32, evaluated on the GPU so that a 10 000-instance memory -- 40 000 crops -- is embedded in a minute); both
embedding sets are matched and assigned by the same exact code; reported: the embedding error, the share of fp16 `aug` entries that
differ, and the share of frames whose assignment LIST differs -- the one quantity of the path the north star calls bit-exact "given
identical embeddings".  The numbers are recorded in DESIGN.md (c); the assertions are loose upper bounds of what was measured.

Second part: the literal numpy transcript of object_memory.py:922-936 (np.dot in fp32, then fp16) against the device's similarity
rows on IDENTICAL embeddings -- the fp32 dot products differ in summation order (<= 2e-6), which moves an fp16 rounding in a small
share of entries; the test counts them and the assignment lists they change."""
import numpy as np
import pytest
import torch

from oracle import vit_oracle as vo

pytestmark = pytest.mark.gpu


def _world(M, E, F, seed):
    import bench
    crops = bench.Crops("dinov2_vitb14", seed)
    rng = np.random.default_rng(seed)
    side = int(np.ceil(np.sqrt(M)))
    frames = []
    for _ in range(F):
        a = int(rng.integers(0, M))
        gx, gy = a % side, a // side
        ids = [k for dy in (-1, 0, 1) for dx in (-1, 0, 1) for k in [(gy + dy) * side + gx + dx]
               if 0 <= gx + dx < side and 0 <= gy + dy < side and k < M][:7]
        frames.append(ids)
    return crops, rng, frames


def _embed_both(enc, w, cfg, crops_u8):
    """crops (N, 224, 224, 3) u8 host array -> (HIP embeddings, fp32 oracle embeddings), both (N, D) numpy"""
    from ibloc_amd import preprocess as pp
    hip, ora = [], []
    wt = {k: torch.from_numpy(np.asarray(v, dtype=np.float32)).cuda() for k, v in w.items()}
    for i in range(0, len(crops_u8), 256):
        c = crops_u8[i:i + 256]
        hip.append(enc.embed(torch.from_numpy(c).cuda()).cpu().numpy())
        ora.append(vo.embed_crops(wt, cfg, pp.RECIPES[cfg.recipe], list(c), device="cuda"))
    return np.concatenate(hip), np.concatenate(ora)


@pytest.mark.parametrize("M,F", [(1000, 64), (10000, 64)], ids=["C2", "T"])
def test_assignment_flip_rate_fp16_encoder_vs_fp32_oracle(M, F):
    import os
    if M > 1000 and not os.environ.get("IBL_FULL_PARITY"):
        pytest.skip("the 10 000-instance case embeds 2 x 40 000 crops (3 minutes): IBL_FULL_PARITY=1; its output is committed "
                    "under profiles/r02/parity_flip_rate.txt")
    from ibloc_amd import vit as V
    from ibloc_amd.engine import LocaliseEngine, MemoryShard
    from ibloc_amd.registration import RegContext
    cfg = V.CONFIGS["dinov2_vitb14"]
    w = V.random_weights(cfg, 20)
    enc = V.VitEncoder(cfg, w)
    E = 4
    crops, rng, frames = _world(M, E, F, 21)
    mem_h, mem_o = [], []
    ids_all = np.repeat(np.arange(M), E)
    for i in range(0, len(ids_all), 1024):
        u8 = crops.variants(ids_all[i:i + 1024], rng, "cpu").numpy()
        h, o = _embed_both(enc, w, cfg, u8)
        mem_h.append(h)
        mem_o.append(o)
        crops._base.clear()
    mem_h, mem_o = np.concatenate(mem_h).reshape(M, E, -1), np.concatenate(mem_o).reshape(M, E, -1)
    q_ids = [k for f in frames for k in f]
    det_h, det_o = _embed_both(enc, w, cfg, crops.variants(q_ids, rng, "cpu").numpy())
    rel = np.linalg.norm(mem_h - mem_o, axis=-1) / np.linalg.norm(mem_o, axis=-1)
    q = [len(f) for f in frames]
    out = {}
    ctx = RegContext(64 << 20)
    for name, mem, det in (("hip", mem_h, det_h), ("oracle", mem_o, det_o)):
        eng = LocaliseEngine(MemoryShard(ctx, list(mem)))
        res = eng.localise_batch(None, q, det_emb=det, register=False)
        from ibloc_amd import match
        _, aug = match.closest_similarity(match.normalize_rows(torch.from_numpy(det).cuda()), eng.memory.mem_emb, eng.memory.emb_offsets,
                                          want_sims=False, want_aug=True)
        out[name] = ([r.assignments for r in res], aug.cpu().numpy())
    ctx.close()
    (la, aa), (lb, ab) = out["hip"], out["oracle"]
    aug_diff = float(np.mean(aa != ab))
    lists_equal = np.mean([x == y for x, y in zip(la, lb)])
    top1_equal = np.mean([[a for a in x if len(a) == 1][:1] == [a for a in y if len(a) == 1][:1] for x, y in zip(la, lb)])
    all_correct = []
    for lst in (la, lb):
        all_correct.append(np.mean([all(frames[f][d] == m for a in lst[f] for d, m in a) for f in range(F)]))
    print(f"[flip M={M}] embedding rel-L2: mean {rel.mean():.2e} max {rel.max():.2e}; fp16 aug entries that differ: {100 * aug_diff:.2f} %; "
          f"frames with identical assignment lists: {100 * lists_equal:.1f} %; identical best single match: {100 * top1_equal:.1f} %; "
          f"frames whose every listed pair is a true match: HIP {100 * all_correct[0]:.1f} % / fp32 {100 * all_correct[1]:.1f} %")
    assert rel.max() < 3e-3
    assert top1_equal == 1.0                                   # the decisive match never moves
    assert abs(all_correct[0] - all_correct[1]) <= 0.05        # ... and the lists are equally right on both sides


def test_similarity_rows_vs_the_literal_numpy_transcript():
    """identical embeddings on both sides: np.dot (fp32) -> max over the instance's views -> fp16 (object_memory.py:922-936,
    similarity_volume.py:13-18) against ibl_closest_similarity"""
    from ibloc_amd import match
    from ibloc_amd.assign import assign_batch
    rng = np.random.default_rng(5)
    M, E, D, F = 1000, 4, 768, 64
    base = rng.normal(size=(M, D))
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    common = rng.normal(size=D)
    common /= np.linalg.norm(common)
    mem = (0.9 * common + 0.45 * base[:, None, :] + rng.normal(0, 0.05 / np.sqrt(D), size=(M, E, D))).astype(np.float32)      # look-alikes: sims ~0.8
    ids = rng.integers(0, M, size=7 * F)
    det = (0.9 * common + 0.45 * base[ids] + rng.normal(0, 0.05 / np.sqrt(D), size=(7 * F, D))).astype(np.float32)
    # the transcript
    mem_n = (mem / np.linalg.norm(mem, axis=-1, keepdims=True)).astype(np.float32)
    det_n = (det / np.linalg.norm(det, axis=-1, keepdims=True)).astype(np.float32)
    # np.dot per stored view as in the reference's double loop (:933-936); (q, d) @ (d,) products evaluated by the same BLAS in fp32
    sims = np.stack([(mem_n.reshape(M * E, D) @ det_n[i]).reshape(M, E).max(-1) for i in range(7 * F)])
    aug_np = np.ones((7 * F, M + 1), dtype=np.float16)
    aug_np[:, :-1] = sims
    off = (torch.arange(M + 1, dtype=torch.int32) * E).cuda()
    _, aug = match.closest_similarity(match.normalize_rows(torch.from_numpy(det).cuda()),
                                      match.normalize_rows(torch.from_numpy(mem.reshape(M * E, D)).cuda()), off, want_sims=False, want_aug=True)
    aug_dev = aug.cpu().numpy()
    diff = aug_np != aug_dev
    ulp = np.abs(aug_np.view(np.int16).astype(np.int32) - aug_dev.view(np.int16).astype(np.int32))
    pad = lambda a: np.ascontiguousarray(a.reshape(F, 7, M + 1))
    la = assign_batch(pad(aug_np), [7] * F, 4)
    lb = assign_batch(pad(aug_dev), [7] * F, 4)
    flips = np.mean([x != y for x, y in zip(la, lb)])
    print(f"[transcript] fp16 aug entries that differ: {100 * diff.mean():.3f} % (never by more than {ulp.max()} fp16 step); "
          f"frames whose assignment list differs: {100 * flips:.1f} %")
    assert ulp.max() <= 1 and diff.mean() < 0.02
