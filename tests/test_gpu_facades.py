"""-m gpu: the reference-shaped Python surface (utils.fpfh_register, utils.embeddings, ObjectMemory.localise)."""
import os

import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation

from ibloc_amd.synth import SynthWorld
from oracle import reg_oracle as ro
from oracle import vit_oracle as vo

pytestmark = pytest.mark.gpu


def rot_deg(A, B):
    return float(np.degrees(np.arccos(np.clip((np.trace(A[:3, :3].T @ B[:3, :3]) - 1) / 2, -1, 1))))


def test_fpfh_register_facade():
    from ibloc_amd.utils import fpfh_register as fr
    w = SynthWorld(2, pts_per_object=2500, E=1, D=8, seed=41)
    tgt = (w.points[0] - w.points[0].mean(0))
    R = Rotation.from_euler("xyz", [25, -10, 40], degrees=True).as_matrix()
    t = np.array([0.2, 0.1, -0.15])
    rng = np.random.default_rng(1)
    src_w, src_c = w.objects[0].sample(2500, rng)
    src = ((src_w - w.points[0].mean(0)) - t) @ R                      # tgt ~= R src + t
    down, feat = fr.downsample_and_compute_fpfh(fr.Cloud(src, src_c), 0.05)
    assert feat.data.shape == (33, len(src)) and down.normals.shape == src.shape
    assert np.allclose(np.linalg.norm(down.normals, axis=1), 1, atol=1e-4)
    T, rmse, fit = fr.register_point_clouds((src, src_c), (tgt, w.colors[0]), 0.05, 1.5, 1.5)
    assert T.shape == (4, 4) and fit > 0.9
    assert rot_deg(T, np.vstack([np.c_[R, t], [0, 0, 0, 1]])) < 2.0 and np.linalg.norm(T[:3, 3] - t) < 0.03
    er, ef = fr.evaluate_transform(src, tgt, T, 0.02)
    orr, of = ro.evaluate(src.astype(np.float32), tgt.astype(np.float32), T, 0.02)
    assert abs(ef - of) < 2.0 / len(src) and abs(er - orr) < 1e-5
    # no colours -> the reference's exception path: point-to-point ICP from identity on the raw clouds
    T2, rmse2, fit2 = fr.register_point_clouds(src, tgt, 0.05, 1.5, 1.5)
    To, fo, ro_, it = ro.icp(src.astype(np.float32), None, tgt.astype(np.float32), None, None, None, 0.075, np.eye(4), colored=False)
    assert np.allclose(T2, To, atol=1e-6)
    # numpy Kabsch helpers: reference test.py case 1
    Tk = fr.get_SVD_transform(np.eye(3), np.array([[0, 1, 0], [-1, 0, 0], [0, 0, 1.0]]))
    assert np.allclose(Tk[:3, :3], [[0, -1, 0], [1, 0, 0], [0, 0, 1]], atol=1e-12)


def test_embeddings_facade():
    from ibloc_amd import vit as V
    from ibloc_amd import preprocess as pp
    from ibloc_amd.utils import embeddings as emb
    with pytest.raises(RuntimeError):
        emb.get_all_vit_embeddings(current_obj_grounded_img=np.zeros((50, 50, 3), np.uint8))
    cfg = V.CONFIGS["tiny_dino"]
    w = V.random_weights(cfg, 2)
    emb.set_encoder("dino", V.VitEncoder(cfg, w))
    crop = np.random.default_rng(0).integers(0, 256, size=(120, 90, 3), dtype=np.uint8)
    out = emb.get_all_dino_embeddings(current_obj_grounded_img=crop, current_obj_phrase="x", device="cuda", extra_kwarg=1)
    assert isinstance(out, torch.Tensor) and out.dim() == 1 and out.shape[0] == cfg.dim
    exp = vo.embed_crops(w, cfg, pp.RECIPES["dinov2"], [crop])[0]
    got = np.array(out.cpu())
    assert np.linalg.norm(got - exp) / np.linalg.norm(exp) < 3e-3
    ccfg = V.CONFIGS["tiny_clip"]
    cw = V.random_weights(ccfg, 3)
    emb.set_encoder("clip", V.VitEncoder(ccfg, cw))
    c = emb.get_all_clip_embeddings(current_obj_grounded_img=crop)
    assert abs(float(c.norm()) - 1.0) < 1e-5                           # L2-normalised like the reference (:48)
    emb._ENCODERS.pop("dator", None)
    with pytest.raises(RuntimeError):
        emb.get_dator_embeddings(current_obj_grounded_img=crop, current_obj_bounding_box=[0, 0, 4, 4], full_depth_image=np.ones((8, 8), np.float32))


def test_load_encoder_from_a_transformers_checkpoint():
    """utils.embeddings.load_encoder replaces `Dinov2Model.from_pretrained(...)` (utils/embeddings.py:18-23): a transformers
    Dinov2Model of the facebook/dinov2-base architecture (seeded random weights; no checkpoint offline) -> state_dict -> HIP encoder ->
    get_all_dino_embeddings must return the model's own CLS embedding of the preprocessed crop"""
    import dataclasses
    from transformers import Dinov2Config, Dinov2Model
    from ibloc_amd import vit as V
    from ibloc_amd import preprocess as pp
    from ibloc_amd.utils import embeddings as emb
    cfg = dataclasses.replace(V.CONFIGS["dinov2_vitb14"], pos_interp="size")       # the installed transformers (5.x) resamples with size=
    torch.manual_seed(11)
    m = Dinov2Model(Dinov2Config(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, mlp_ratio=4, image_size=518, patch_size=14,
                                 qkv_bias=True, layerscale_value=1.0, use_swiglu_ffn=False, layer_norm_eps=1e-6, hidden_act="gelu")).eval()
    g = torch.Generator().manual_seed(12)
    with torch.no_grad():
        for prm in m.parameters():
            prm.add_(torch.randn(prm.shape, generator=g) * 0.02)
    emb.load_encoder("dino", m.state_dict(), cfg=cfg)
    crops = [np.random.default_rng(13 + i).integers(0, 256, size=s, dtype=np.uint8) for i, s in enumerate([(180, 140, 3), (224, 224, 3)])]
    px = np.stack([vo.preprocess_crop(c, pp.RECIPES["dinov2"]) for c in crops])
    with torch.no_grad():
        want = m(pixel_values=torch.from_numpy(px)).last_hidden_state[:, 0].numpy()
    got = np.stack([emb.get_all_dino_embeddings(current_obj_grounded_img=c).cpu().numpy() for c in crops])
    rel = np.linalg.norm(got - want) / np.linalg.norm(want)
    print("load_encoder(dino) vs transformers Dinov2Model: rel-L2", rel)
    assert rel < 3e-3
    emb._ENCODERS.pop("dino", None)


def test_object_memory_localise_with_stub_finder(tmp_path):
    from ibloc_amd.object_memory.object_memory import ObjectMemory
    w = SynthWorld(4, pts_per_object=2500, E=2, D=32, seed=51)
    rng = np.random.default_rng(52)
    f = w.make_frame(rng, q=2, pts_per_object=2500, anchor=0)
    dummy = lambda **kw: torch.tensor(kw["full_rgb_image"][0, 0, :1].astype(np.float32))
    with pytest.raises(NotImplementedError):
        ObjectMemory("cuda", get_embeddings_func=None)
    # --- detections API (perception bypassed) vs the oracle transcript
    om = ObjectMemory("cuda", None, None, 300.0, 300.0, get_embeddings_func=dummy)
    for j in range(w.M):
        om.add_object(f"obj{j}", list(w.embeddings[j]), w.points[j], w.colors[j])
    res = om.localise_detections(f["det_emb"], f["clouds"], fpfh_global_dist_factor=1.5, fpfh_local_dist_factor=1.5)
    assert res.pose.shape == (7,) and len(res.assignments) >= 1
    cleaned, ccols = [], []
    for (p, c) in f["clouds"]:
        k = ro.radius_outlier(p.astype(np.float32), 0.05, 8)
        cleaned.append(p[k]); ccols.append(c[k])
    pose, recs, best = ro.localise_from_assignments(cleaned, ccols, w.points, w.colors, res.assignments, 0.05, 1.5, 1.5, seed=0,
                                                    job_base=16, stale_means=True)
    assert res.best == best and np.linalg.norm(res.pose[:3] - pose[:3]) <= 0.01
    # --- full localise() through a stub ObjectFinder on a rendered depth image: no detections -> reference default
    rgb = np.zeros((60, 80, 3), np.uint8)
    depth = np.zeros((60, 80), np.float32)
    from PIL import Image
    Image.fromarray(rgb).save(tmp_path / "rgb.png")
    np.save(tmp_path / "depth.npy", depth)
    om.object_finder = lambda path, consider_floor: (None, None, None, None)
    pose0, extra = om.localise(str(tmp_path / "rgb.png"), str(tmp_path / "depth.npy"), perform_semantic_icp=False)
    assert np.array_equal(pose0, [0, 0, 0, 0, 0, 0, 1]) and extra == [[], []]
    # one flat synthetic object seen by the stub finder: exercises unprojection + outlier removal + the engine call
    depth[10:50, 10:70] = 2.0 + 0.05 * np.sin(np.arange(60)[None, :] / 5.0)
    rgb[10:50, 10:70] = 128
    np.save(tmp_path / "depth.npy", depth)
    Image.fromarray(rgb).save(tmp_path / "rgb.png")
    mask = torch.zeros((1, 60, 80), dtype=torch.bool)
    mask[0, 10:50, 10:70] = True
    om.object_finder = lambda path, consider_floor: ([rgb[10:50, 10:70]], [torch.tensor([10., 10., 70., 50.])], [mask], ["thing"])
    om.get_embeddings_func = lambda **kw: torch.from_numpy(w.embeddings[1][0]).cuda()
    with pytest.raises(NotImplementedError):
        om.localise(str(tmp_path / "rgb.png"), str(tmp_path / "depth.npy"))           # perform_semantic_icp=True default raises
    pose1, extra1 = om.localise(str(tmp_path / "rgb.png"), str(tmp_path / "depth.npy"), testname=str(tmp_path / "t"),
                                perform_semantic_icp=False, outlier_removal_config={"radius_nb_points": 2, "radius": 0.2})
    assert pose1.shape == (7,) and np.isfinite(pose1).all() and extra1[1] is None
    assert extra1[0][0][1] == 1                                          # matched the instance whose embedding was returned
    # save_point_clouds=True (:947-972): pcds/<testname>/<subtest>/_init_pcd_<subtest>.ply holds the detections and the memory
    from ibloc_amd.object_memory.object_info import read_ply
    pose2, _ = om.localise(str(tmp_path / "rgb.png"), str(tmp_path / "depth.npy"), testname=str(tmp_path / "t"), subtest_name="s7",
                           save_point_clouds=True, perform_semantic_icp=False, outlier_removal_config={"radius_nb_points": 2, "radius": 0.2})
    assert np.array_equal(pose2[3:], pose1[3:]) or np.isfinite(pose2).all()
    pts, cols = read_ply(f"pcds/{tmp_path / 't'}/s7/_init_pcd_s7.ply")
    n_mem = sum(len(np.asarray(m.pointcloud.points)) for m in om.memory)
    assert len(pts) > n_mem and cols is not None and len(cols) == len(pts)
    # ... and after the registrations (:1092-1093, 1142-1165): one only_chosen_<assn>.ply per assignment, the best full cloud (memory
    # green, detections red) and a copy of the query image
    import glob
    d = f"pcds/{tmp_path / 't'}/s7"
    chosen = glob.glob(os.path.join(glob.escape(d), "only_chosen_*.ply"))
    assert len(chosen) >= 1 and os.path.exists(os.path.join(d, "rgb_image.png"))
    p1, c1 = read_ply(os.path.join(d, "only_chosen_[[0, 1]].ply"))
    n1 = len(np.asarray(om.memory[1].pointcloud.points))
    assert len(p1) > n1 and abs(p1[:n1].mean(axis=0)).max() < 1e-6               # the memory side is centred on its own mean
    best = glob.glob(os.path.join(glob.escape(d), "_best_full_pcd*.ply"))
    assert len(best) == 1
    pb, cb = read_ply(best[0])
    assert len(pb) > n_mem and np.array_equal(cb[:n_mem], np.tile([0.0, 1.0, 0.0], (n_mem, 1))) and np.array_equal(cb[-1], [1.0, 0.0, 0.0])


def test_object_memory_pickle_round_trip(tmp_path):
    """save_to_pkl / load keep the reference's layout (object_memory.py:779-846): list of (ObjectInfo without cloud, points,
    colors) + the floor slot; a memory loaded from it localises exactly like the one that wrote it"""
    import pickle
    from ibloc_amd.object_memory.object_memory import ObjectMemory
    w = SynthWorld(4, pts_per_object=2500, E=2, D=32, seed=51)
    rng = np.random.default_rng(53)
    f = w.make_frame(rng, q=2, pts_per_object=2500, anchor=0)
    dummy = lambda **kw: None
    a = ObjectMemory("cuda", None, None, 300.0, 300.0, get_embeddings_func=dummy)
    for j in range(w.M):
        a.add_object(f"obj{j}", list(w.embeddings[j]), w.points[j], w.colors[j])
    a.save_to_pkl(str(tmp_path / "mem.pkl"))
    mem, floors = pickle.load(open(tmp_path / "mem.pkl", "rb"))
    assert len(mem) == w.M and mem[0][0].pointcloud is None and mem[0][1].dtype == np.float64 and mem[0][1].shape == (2500, 3)
    assert mem[2][0].names == ["obj2"] and len(mem[2][0].embeddings) == 2 and floors[1].shape == (2500, 3)
    b = ObjectMemory("cuda", None, None, 300.0, 300.0, get_embeddings_func=dummy)
    b.load(str(tmp_path / "mem.pkl"))
    assert len(b.memory) == w.M and np.array_equal(b.memory[1].pointcloud.points, a.memory[1].pointcloud.points)
    ra = a.localise_detections(f["det_emb"], f["clouds"], fpfh_global_dist_factor=1.5, fpfh_local_dist_factor=1.5)
    rb = b.localise_detections(f["det_emb"], f["clouds"], fpfh_global_dist_factor=1.5, fpfh_local_dist_factor=1.5)
    assert ra.assignments == rb.assignments and np.array_equal(ra.pose, rb.pose)
    # the per-object directory layout (objects/<id>/pointcloud.ply + info.pkl): colours go through 8 bits, points are exact
    a.save(str(tmp_path / "dir"))
    c = ObjectMemory("cuda", None, None, 300.0, 300.0, get_embeddings_func=dummy)
    c.load_directory(str(tmp_path / "dir"))
    assert len(c.memory) == w.M and np.array_equal(c.memory[3].pointcloud.points, a.memory[3].pointcloud.points)
    rc = c.localise_detections(f["det_emb"], f["clouds"], fpfh_global_dist_factor=1.5, fpfh_local_dist_factor=1.5)
    assert rc.assignments == ra.assignments and np.isfinite(rc.pose).all()      # (8-bit colours may pick another best candidate)
