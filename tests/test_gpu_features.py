"""-m gpu: grids, hybrid neighbourhoods, normals, FPFH and radius-outlier masks vs the C oracle."""
import numpy as np
import pytest
import torch

from ibloc_amd.synth import SynthWorld
from oracle import reg_oracle as ro

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from ibloc_amd.registration import RegContext
    c = RegContext(2 << 30)
    yield c
    c.close()


def clouds(sizes, seed):
    w = SynthWorld(len(sizes), pts_per_object=max(sizes), E=1, D=8, seed=seed)
    out = []
    for i, n in enumerate(sizes):
        p = w.points[i][:n]
        out.append((p - p.mean(0)).astype(np.float32))
    return out


def test_radius_outlier_bit_exact(ctx):
    from ibloc_amd.registration import CloudBatch, radius_outlier_batch
    cs = clouds([3000, 1, 0, 2500, 700], 3)
    rng = np.random.default_rng(0)
    cs[0] = np.concatenate([cs[0], rng.uniform(-3, 3, size=(40, 3)).astype(np.float32)])     # sprinkle outliers
    b = CloudBatch.from_numpy(cs)
    keep = radius_outlier_batch(ctx, b, 0.05, 8).cpu().numpy().astype(bool)
    off = b.seg_off_host
    for i, c in enumerate(cs):
        exp = ro.radius_outlier(c, 0.05, 8) if len(c) else np.zeros(0, bool)
        assert np.array_equal(keep[off[i]:off[i + 1]], exp), f"cloud {i}"
    assert keep.sum() > 0 and (~keep).sum() >= 30


def test_normals_and_fpfh_vs_oracle(ctx):
    from ibloc_amd.registration import CloudBatch, normals_fpfh_batch
    cs = clouds([4000, 2500, 3, 0, 1500], 5)
    # a concatenation of two objects, like a length-2 assignment (neighbourhoods may span both)
    cs.append(np.concatenate([cs[0][:1500] + np.float32([0.4, 0, 0]), cs[1][:1500]]))
    b = CloudBatch.from_numpy(cs)
    nrm, fpfh = normals_fpfh_batch(ctx, b, 0.1, 30, 0.25, 100)
    torch.cuda.synchronize()
    assert ctx.status() == 0
    nrm, fpfh = nrm.cpu().numpy(), fpfh.cpu().numpy()
    off = b.seg_off_host
    for i, c in enumerate(cs):
        if len(c) == 0:
            continue
        en = ro.normals(c, 0.1, 30)
        gn = nrm[off[i]:off[i + 1], :3]
        # same neighbour sets + same solver in double: identical up to fp32 rounding, sign included
        err = np.abs(gn - en).max(1)
        assert np.mean(err < 1e-5) > 0.999, f"cloud {i}: {np.mean(err < 1e-5)}"
        ef = ro.fpfh(c, en, 0.25, 100)
        gf = fpfh[off[i]:off[i + 1]]
        row = np.abs(gf - ef).max(1)
        ok = row < 2e-3
        assert np.mean(ok) > 0.995, f"cloud {i}: fpfh rows within tol {np.mean(ok)}"


def test_knn_slow_path_matches_fast_path(ctx):
    """many equidistant candidates (points on a sphere around the query) overflow the 256-entry boundary list"""
    from ibloc_amd.registration import CloudBatch, normals_fpfh_batch
    rng = np.random.default_rng(2)
    v = rng.normal(size=(3000, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    pts = np.concatenate([np.zeros((1, 3)), v * 0.05, rng.normal(size=(500, 3)) * 0.01]).astype(np.float32)
    b = CloudBatch.from_numpy([pts])
    nrm, fpfh = normals_fpfh_batch(ctx, b, 0.1, 30, 0.25, 100)
    torch.cuda.synchronize()
    en = ro.normals(pts, 0.1, 30)
    ef = ro.fpfh(pts, en, 0.25, 100)
    assert np.mean(np.abs(nrm.cpu().numpy()[:, :3] - en).max(1) < 1e-5) > 0.99
    assert np.mean(np.abs(fpfh.cpu().numpy() - ef).max(1) < 2e-3) > 0.99


def test_fused_normals_and_feature_search_equals_the_two_searches(ctx, monkeypatch):
    """instance features take the normals' <= 30 neighbours from the 100-neighbour list of the feature search (one search instead of two);
    IBL_FEAT_UNFUSED=1 runs the two stand-alone searches: normals and FPFH must agree bit for bit, dense and sparse clouds alike"""
    from ibloc_amd.registration import CloudBatch, instance_features_batch, normals_fpfh_batch
    rng = np.random.default_rng(77)
    cs = clouds([5000, 1200, 40, 3, 0], 9)
    cs.append((rng.uniform(-0.05, 0.05, size=(3000, 3))).astype(np.float32))          # > 100 points inside every normal radius
    b = CloudBatch.from_numpy(cs)
    got = instance_features_batch(ctx, b, 0.05)
    n1, f1 = normals_fpfh_batch(ctx, b, 0.1, 30, 0.25, 100)
    monkeypatch.setenv("IBL_FEAT_UNFUSED", "1")
    ref = instance_features_batch(ctx, b, 0.05)
    n2, f2 = normals_fpfh_batch(ctx, b, 0.1, 30, 0.25, 100)
    torch.cuda.synchronize()
    assert ctx.status() == 0
    assert torch.equal(got.normals[:b.n], ref.normals[:b.n]) and torch.equal(got.fpfh[:b.n], ref.fpfh[:b.n])
    assert torch.equal(n1, n2) and torch.equal(f1, f2)


def test_spfh_fp32_bins_with_fp64_for_undecided_pairs_equal_the_fp64_bins(ctx, monkeypatch):
    """round 4: the SPFH bins of a pair come from fp32 arithmetic when every decision clears a guard band (pair_bins_f32, csrc/reg_knn.hip)
    and from the fp64 pair features otherwise (queue + second kernel).  IBL_SPFH_F64=1 evaluates every pair in fp64; IBL_SPFH_QCAP=8
    overflows the queue, which the gated fp64 launch repairs: all three must give the same bytes -- noisy surfaces, exact planes (equal
    normals, theta on a bin boundary for every pair), a cloud 200 m from the origin, tiny and empty clouds"""
    from ibloc_amd.registration import CloudBatch, instance_features_batch
    rng = np.random.default_rng(79)
    cs = clouds([6000, 3000, 1200, 40, 3, 0], 13)
    g = np.stack(np.meshgrid(np.arange(60), np.arange(60), indexing="ij"), -1).reshape(-1, 2) * 0.011
    cs.append(np.concatenate([g, np.zeros((len(g), 1))], 1).astype(np.float32))                           # an exact plane on a lattice
    cs.append((np.concatenate([g, 0.002 * rng.normal(size=(len(g), 1))], 1) + [211.5, -187.25, 3.0]).astype(np.float32))
    cs.append((rng.uniform(-0.05, 0.05, size=(3000, 3))).astype(np.float32))
    box = rng.uniform(-0.2, 0.2, size=(6000, 3))
    box[np.arange(6000), rng.integers(0, 3, 6000)] = rng.choice([-0.2, 0.2], 6000)                        # the six faces of a cube
    cs.append(box.astype(np.float32))
    b = CloudBatch.from_numpy(cs)
    got = instance_features_batch(ctx, b, 0.05)
    monkeypatch.setenv("IBL_SPFH_F64", "1")
    ref = instance_features_batch(ctx, b, 0.05)
    monkeypatch.delenv("IBL_SPFH_F64")
    monkeypatch.setenv("IBL_SPFH_QCAP", "8")
    over = instance_features_batch(ctx, b, 0.05)
    torch.cuda.synchronize()
    assert ctx.status() == 0
    assert float(ref.fpfh[:b.n].abs().sum()) > 0
    assert torch.equal(got.fpfh[:b.n], ref.fpfh[:b.n]) and torch.equal(got.normals[:b.n], ref.normals[:b.n])
    assert torch.equal(over.fpfh[:b.n], ref.fpfh[:b.n])


def test_guess_threshold_selection_equals_the_two_pass_selection(ctx, monkeypatch):
    """round 3: the tile search collects the candidates below a GUESS of the k-th neighbour's distance in one pass and selects from that
    list (tile_select_guess); IBL_KNN_NOGUESS=1 runs the two-pass histogram selection for every query: neighbour sets, normals, FPFH
    and colour gradients must agree bit for bit -- dense, sparse, tiny and two-object clouds alike"""
    from ibloc_amd.registration import CloudBatch, instance_features_batch, normals_fpfh_batch
    rng = np.random.default_rng(78)
    cs = clouds([5000, 3000, 1200, 40, 3, 0], 11)
    cs.append((rng.uniform(-0.05, 0.05, size=(3000, 3))).astype(np.float32))          # > 100 points inside every normal radius
    cs.append(np.concatenate([cs[0][:2000] + np.float32([0.3, 0, 0]), cs[1][:2000]]))
    ints = [rng.uniform(0, 1, size=len(c)).astype(np.float32) for c in cs]
    b = CloudBatch.from_numpy(cs, ints)
    got = instance_features_batch(ctx, b, 0.05, grad_radius=0.15)
    n1, f1 = normals_fpfh_batch(ctx, b, 0.1, 30, 0.25, 100)
    monkeypatch.setenv("IBL_KNN_NOGUESS", "1")
    ref = instance_features_batch(ctx, b, 0.05, grad_radius=0.15)
    n2, f2 = normals_fpfh_batch(ctx, b, 0.1, 30, 0.25, 100)
    torch.cuda.synchronize()
    assert ctx.status() == 0
    assert torch.equal(got.normals[:b.n], ref.normals[:b.n]) and torch.equal(got.fpfh[:b.n], ref.fpfh[:b.n])
    assert torch.equal(got.grad[:b.n], ref.grad[:b.n]) and torch.equal(got.fpfh_split[:b.n], ref.fpfh_split[:b.n])
    assert torch.equal(n1, n2) and torch.equal(f1, f2)
