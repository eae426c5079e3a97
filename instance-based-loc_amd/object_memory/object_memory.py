"""Drop-in for the query side of the reference's `object_memory/object_memory.py` on the MI355X build.

`ObjectMemory.localise()` keeps the reference's keyword list and return shape
(/root/reference/object_memory/object_memory.py:852-863, 896, 1169); everything from the embedding of the object
crops to the returned pose runs through the batched HIP engine (ibloc_amd.engine).  The perception front end
(RAM + GroundingDINO + SAM, object_memory/object_finder.py) is out of scope for this build: pass any callable with
the reference's `ObjectFinder.find` contract as `object_finder=` (find(rgb_path, consider_floor) ->
(grounded_imgs, bounding_boxes, masks, phrases) or (None, None, None, None)).  Memory construction mirrors
`process_image`, `downsample_all_objects`, `remove_points_below_floor`, `recluster_objects_with_dbscan`, `_recluster_IoU` and
`recluster_via_clustering_and_IoU`: voxel down-sampling and DBSCAN run on the device (ibloc_amd.build), agglomerative clustering
is scikit-learn's as in the reference, and the object-aligned IoU (Open3D OBB + Objectron, third-party) is passed in as
`iou_func`.  `add_object()` / `load()` fill the memory with (name, embeddings, cloud) records in the reference's ObjectInfo layout.
"""
import os

import numpy as np
import torch

from ibloc_amd.engine import LocaliseEngine, MemoryShard, intensity_from_colors
from ibloc_amd.registration import CloudBatch, RegContext, radius_outlier_batch, unproject_masks
from ibloc_amd.utils.fpfh_register import Cloud

from .object_info import ObjectInfo

DEFAULT_OUTLIER_REMOVAL_CONFIG = {"radius_nb_points": 12, "radius": 0.05}      # utils/depth_utils.py:5-10


def default_load_rgb(path: str) -> np.ndarray:
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))


def default_load_depth(path: str) -> np.ndarray:
    if path.split('.')[-1] == 'npy':
        return np.load(path)
    from PIL import Image
    return np.asarray(Image.open(path))


def _compact(batch: CloudBatch, keep: torch.Tensor) -> CloudBatch:
    """drop the points with keep == 0 (order preserved)"""
    kb = keep.bool()
    csum = torch.cat([torch.zeros(1, dtype=torch.int64, device=keep.device), torch.cumsum(kb.to(torch.int64), 0)])
    off = csum[torch.from_numpy(batch.seg_off_host.astype(np.int64)).to(keep.device)].cpu().numpy().astype(np.int32)
    return CloudBatch(batch.pts4[kb].contiguous(), off)


def _select(batch: CloudBatch, order) -> CloudBatch:
    """the clouds `order` of a batch, in that order"""
    off = batch.seg_off_host
    parts = [batch.pts4[off[i]:off[i + 1]] for i in order]
    sizes = [int(off[i + 1] - off[i]) for i in order]
    pts = torch.cat(parts).contiguous() if parts else batch.pts4[:0]
    return CloudBatch(pts, np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32))


class ObjectMemory():
    def __init__(self, device, ram_pretrained_path=None, sam_checkpoint_path=None, camera_focal_lenth_x=None,
                 camera_focal_lenth_y=None, get_embeddings_func=None, log_enabled=True,
                 mem_formation_bounding_box_threshold=0.3, mem_formation_occlusion_overlap_threshold=0.9,
                 object_info_max_embeddings_num=1000000, load_rgb_image_func=default_load_rgb,
                 load_depth_image_func=default_load_depth, dataset_floor_thickness=0.1, lora_path=None,
                 object_finder=None, arena_bytes=8 << 30):
        if get_embeddings_func is None:
            raise NotImplementedError("Need to pass in get_embeddings_func")       # reference :112
        self.device = device
        self.camera_focal_lenth_x = camera_focal_lenth_x
        self.camera_focal_lenth_y = camera_focal_lenth_y
        self.get_embeddings_func = get_embeddings_func
        self.log_enabled = log_enabled
        self.object_info_max_embeddings_num = object_info_max_embeddings_num
        self.load_rgb_image_func = load_rgb_image_func
        self.load_depth_image_func = load_depth_image_func
        self.object_finder = object_finder
        self.dataset_floor_thickness = dataset_floor_thickness
        self.iou_func = None            # (points_i, points_j) -> IoU for _recluster_IoU; the reference's Objectron IoU is third-party
        self.memory = []
        self.floors = None
        self._ctx = RegContext(arena_bytes)
        self._engine = None
        self._shard = None
        self.ransac_seed = 0
        self._n_queries = 0

    def __repr__(self):
        return "".join(f"\t{o}\n" for o in self.memory) or "\tNo objects in memory yet."

    # ---- memory content ---------------------------------------------------------------------------------
    def add_object(self, name, embeddings, points, colors):
        embs = [np.asarray(e) for e in embeddings]
        info = ObjectInfo(len(self.memory), name, embs[0], Cloud(points, colors), self.object_info_max_embeddings_num)
        info.embeddings = embs
        info._compute_means()
        self.memory.append(info)
        self._engine = None

    # ---- memory build (object_memory.py:163-256) ----------------------------------------------------------
    def _log(self, *a):
        if self.log_enabled:
            print(*a)

    def process_detections(self, obj_phrases, embs, obj_clouds, pose, add_noise=False, pose_noise={'trans': 0.0005, 'rot': 0.0005},
                           depth_noise=0.003, min_points=500):
        """process_image after perception (:187-256): obj_clouds = [(points (N, 3), colors (N, 3)), ...] in the camera frame, pose =
        x y z qx qy qz qw (its quaternion slice is normalised in place, as the reference's transform_pointcloud does)."""
        from ibloc_amd.build import transform_points
        from .object_finder_phrases import check_if_floor
        if add_noise:
            pose[:3] = pose[:3] + np.random.normal(0, pose_noise['trans'], pose[:3].shape)
            q = pose[3:] + np.random.normal(0, pose_noise['rot'], pose[3:].shape)
            nq = np.linalg.norm(q)
            pose[3:] = q if nq == 0 else q / nq
            obj_clouds = [(np.asarray(p) + np.random.normal(0, depth_noise, np.asarray(p).shape), c) for p, c in obj_clouds]
        for phrase, emb, (p, c) in zip(obj_phrases, embs, [(transform_points(p, pose), c) for p, c in obj_clouds]):
            if len(p) < min_points:
                self._log(f"\t\tSkipping as number of points {len(p)} < min_points = {min_points}.")
                continue
            info = ObjectInfo(len(self.memory), phrase, emb, Cloud(p, c), self.object_info_max_embeddings_num)
            if check_if_floor(info.names):
                self.floors = info if self.floors is None else self.floors + info
            else:
                self.memory.append(info)
        self._engine = None

    def process_image(self, rgb_image_path, depth_image_path, pose, consider_floor, outlier_removal_config=DEFAULT_OUTLIER_REMOVAL_CONFIG,
                      add_noise=False, pose_noise={'trans': 0.0005, 'rot': 0.0005}, depth_noise=0.003, min_points=500,
                      will_cluster_later=True, depth_factor=1.):
        if not will_cluster_later:
            raise NotImplementedError("Only final clustering available currently")                     # :233-234
        obj_phrases, embs, obj_clouds = self._get_object_info(rgb_image_path, depth_image_path, consider_floor, outlier_removal_config,
                                                              depth_factor=depth_factor, for_build=True)
        if obj_phrases is None:
            self._log("ObjectMemory.process_image did NOT find any objects")
            return
        self.process_detections(obj_phrases, embs, obj_clouds, pose, add_noise, pose_noise, depth_noise, min_points)

    def downsample_all_objects(self, voxel_size):
        """:258-263 -- every object (and the floor) in ONE device call (ibl_voxel_downsample_batch), bit-identical to the python loop"""
        from ibloc_amd.build import voxel_downsample_batch
        objs = list(self.memory) + ([self.floors] if self.floors is not None else [])
        if not objs:
            return
        have_cols = all(o.pointcloud.colors is not None for o in objs)
        pts, cols = voxel_downsample_batch(self._ctx, [o.pointcloud.points for o in objs],
                                           [o.pointcloud.colors for o in objs] if have_cols else None, voxel_size, device=self._device())
        for i, o in enumerate(objs):
            o.pointcloud = Cloud(pts[i], cols[i] if cols is not None else None)
            o._process_pointcloud()
        self._engine = None

    def remove_points_below_floor(self):
        """:265-293, including its list-mutation-while-iterating behaviour"""
        floor_height = float('inf')
        for info in self.memory:
            floor_height = min(np.min(info.pcd[1, :]), floor_height)
        for info in self.memory:
            info.update_pointcloud_with_mask(info.pcd[1, :] > floor_height + self.dataset_floor_thickness)
            if len(info.pointcloud.points) == 0:
                self.memory.remove(info)
        self._engine = None

    def _device(self):
        return self.device if str(self.device) != "cuda" else "cuda:0"

    def _dbscan_merge(self, groups, eps, min_points):
        """DBSCAN of each group's concatenated points (one device call for all groups), objects -> clusters by their first point,
        merge per cluster (:305-362 and :627-670)"""
        from ibloc_amd.build import clusters_of_first_points, dbscan_batch, merge_objects_by_cluster
        clouds = [np.concatenate([o.pcd for o in g], axis=-1).T for g in groups]
        labels, ncl = dbscan_batch(self._ctx, clouds, eps, min_points, device=self._device())
        out = []
        for g, pts, lab, k in zip(groups, clouds, labels, ncl):
            assn, multi = clusters_of_first_points(pts, lab, np.stack([o.pcd[:, 0] for o in g]))
            if multi:
                self._log("\t\tMULTIPLE LABELS IN COMBINED RECLUSTERING")
            out.append(merge_objects_by_cluster(g, assn, k))
        return out

    def recluster_objects_with_dbscan(self, eps=0.2, min_points_per_cluster=300, visualize=False):
        self._log("Clustering using DBSCAN")
        self.memory = self._dbscan_merge([list(self.memory)], eps, min_points_per_cluster)[0] if self.memory else []
        for i, o in enumerate(self.memory):
            o.id = i
        self._engine = None

    def _recluster_IoU(self, IoU_threshold=0.6, iou_func=None):
        """:710-747: average-linkage agglomerative clustering on 1 - IoU.  iou_func(points_i, points_j): the reference uses the
        object-aligned Objectron IoU (third-party, not in this build, see utils/IoU_ops.py); ObjectMemory.iou_func is the default."""
        from sklearn.cluster import AgglomerativeClustering
        iou_func = iou_func or self.iou_func
        if iou_func is None:
            from ibloc_amd.utils.IoU_ops import calculate_obj_aligned_3d_IoU as iou_func
        n = len(self.memory)
        IoUs = np.zeros((n, n))
        for i in range(n):
            for j in range(i, n):
                IoUs[i][j] = 1 if i == j else 1 - iou_func(np.asarray(self.memory[i].pointcloud.points), np.asarray(self.memory[j].pointcloud.points))
                IoUs[j][i] = IoUs[i][j]
        self._log("Clustering agglomeratively")
        labels = AgglomerativeClustering(n_clusters=None, distance_threshold=1 - IoU_threshold, metric='precomputed', linkage='average').fit(IoUs).labels_
        new_memory = [None for _ in set(labels)]
        for lab, obj in zip(labels, self.memory):
            new_memory[lab] = obj if new_memory[lab] is None else new_memory[lab] + obj
        self.memory = new_memory
        for i, o in enumerate(self.memory):
            o.id = i
            o._compute_means()
        self._engine = None

    def _embedding_labels(self, similarity, distance_func, embedding_distance_threshold):
        """average-linkage clustering of the mean embeddings on the reference's rescaled matrix: the pairwise cosine similarity
        (:382-386) or cosine distance (:518-522 / :565-569), shifted and scaled to [0, 1] and flipped (`1 - x`), exactly as written"""
        from sklearn.cluster import AgglomerativeClustering

        def df(all_obj_embs, all_obj_centroids):
            normalized = all_obj_embs / np.linalg.norm(all_obj_embs, axis=1, keepdims=True)
            sims = np.dot(normalized, normalized.T)
            return sims if similarity else 1 - sims

        D = (distance_func or df)(np.array([o.mean_emb for o in self.memory]), np.array([o.centroid for o in self.memory]))
        D -= np.min(D)
        D /= np.max(D)
        D = 1 - D
        self._log("Clustering agglomeratively")
        return AgglomerativeClustering(n_clusters=None, distance_threshold=embedding_distance_threshold, metric='precomputed',
                                       linkage='average').fit(D).labels_

    def recluster_via_agglomerative_clustering(self, distance_func=None, embedding_distance_threshold=0.4, distance_threshold=0.1):
        """:379-437: merge the objects of every embedding cluster (no geometry involved)"""
        labels = self._embedding_labels(True, distance_func, embedding_distance_threshold)
        self._log(f"{len(set(labels))} objects clustered")
        new_memory = [None for _ in set(labels)]
        for lab, obj in zip(labels, self.memory):
            new_memory[lab] = obj if new_memory[lab] is None else new_memory[lab] + obj
        self.memory = new_memory
        for i, o in enumerate(self.memory):
            o.id = i
        self._engine = None

    def recluster_via_combined(self, distance_func=None, embedding_distance_threshold=0.4, eps=0.4, min_points_per_cluster=150):
        """:443-553: embedding clusters, then DBSCAN inside every cluster (all clusters in one device call) and a merge per DBSCAN label"""
        labels = self._embedding_labels(False, distance_func, embedding_distance_threshold)
        self._log(f"{len(set(labels))} clusters initially")
        groups = [[o for i, o in enumerate(self.memory) if labels[i] == u] for u in set(labels)]
        self.memory = [o for merged in self._dbscan_merge(groups, eps, min_points_per_cluster) for o in merged]
        for i, o in enumerate(self.memory):
            o.id = i
        self._engine = None

    def recluster_via_clustering_and_IoU(self, distance_func=None, embedding_distance_threshold=0.4, eps=0.4, min_points_per_cluster=150,
                                         IoU_threshold=0.25, iou_func=None):
        """:562-708: `_recluster_IoU`, then the steps of `recluster_via_combined`"""
        self._recluster_IoU(IoU_threshold, iou_func)
        self.recluster_via_combined(distance_func, embedding_distance_threshold, eps, min_points_per_cluster)

    # ---- persistence: the reference's pickle layout (object_memory.py:779-846) -----------------------------
    def save_to_pkl(self, save_directory: str):
        """(list of (ObjectInfo without its cloud, points (N, 3) f64, colors (N, 3) f64), same tuple for the floor slot) --
        the layout `load` of either implementation reads."""
        import copy
        import pickle

        def strip(info):
            blank = copy.copy(info)
            blank.pointcloud = None
            blank.pcd = None
            blank.pcd_colors = None
            cols = info.pointcloud.colors
            return (blank, np.asarray(info.pointcloud.points, dtype=np.float64),
                    np.zeros((0, 3)) if cols is None else np.asarray(cols, dtype=np.float64))
        if not self.memory:
            raise RuntimeError("object memory is empty")
        mem = [strip(o) for o in self.memory]
        floors = strip(self.floors) if self.floors is not None else mem[-1]        # the reference stores the last object there
        with open(save_directory, "wb") as f:
            pickle.dump((mem, floors), f)

    def save(self, save_directory: str):
        """object_memory.py:753-777: `objects/<id>/{pointcloud.ply, info.pkl}`, `floors/all_floors/`, `memory.txt` (the combined
        PLY dumps of the reference are visualisation aids and are not written)."""
        os.makedirs(os.path.join(save_directory, "objects"), exist_ok=True)
        with open(os.path.join(save_directory, "memory.txt"), "w") as f:
            f.write(self.__repr__())
        for obj in self.memory:
            obj.save(os.path.join(save_directory, "objects", f"{obj.id}"))
        if self.floors is not None:
            self.floors.save(os.path.join(save_directory, "floors", "all_floors"))

    def load_directory(self, load_directory: str):
        """the directory layout written by `save` of either implementation"""
        root = os.path.join(load_directory, "objects")
        ids = sorted((d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d))), key=lambda d: (not d.isdigit(), int(d) if d.isdigit() else 0, d))
        self.memory = [ObjectInfo.load(os.path.join(root, d), id=i) for i, d in enumerate(ids)]
        floors = os.path.join(load_directory, "floors", "all_floors")
        self.floors = ObjectInfo.load(floors) if os.path.isdir(floors) else None
        self._engine = None

    def load(self, load_directory: str):
        """object_memory.py:831-846.  A pickle written by the reference resolves `object_memory.object_info.ObjectInfo` to this
        package's class when it is first on sys.path; its attributes are taken as they are."""
        import pickle
        with open(load_directory, "rb") as f:
            pklable_memory, pklable_floors = pickle.load(f)

        def conv(info_tuple):
            blank, pts, cols = info_tuple
            blank.pointcloud = Cloud(pts, cols if len(cols) == len(pts) else None)
            blank.embeddings = [np.asarray(e) for e in blank.embeddings]
            if hasattr(blank, "_process_pointcloud"):
                blank._process_pointcloud()
            return blank
        self.memory = [conv(t) for t in pklable_memory]
        self.floors = conv(pklable_floors)
        self._engine = None

    def _get_engine(self):
        if self._engine is None:
            if not self.memory:
                raise RuntimeError("object memory is empty")
            if self._shard is not None:          # a rebuilt memory: release the previous spatial hash's host record, then its arena
                self._shard.close()
            self._ctx.reset()
            shard = self._shard = MemoryShard(self._ctx, [np.stack(m.embeddings).astype(np.float32) for m in self.memory],
                                [np.asarray(m.pointcloud.points) for m in self.memory],
                                colors=[np.asarray(m.pointcloud.colors) for m in self.memory], device=self.device)
            self._engine = LocaliseEngine(shard)
        return self._engine

    # ---- query ------------------------------------------------------------------------------------------
    def _get_object_info(self, rgb_image_path, depth_image_path, consider_floor, outlier_removal_config, depth_factor=1., for_build=False):
        """object_memory.py:125-161: one get_embeddings_func call per detected object, depth -> coloured clouds per mask."""
        if self.object_finder is None:
            raise RuntimeError("no object_finder: the RAM/GroundingDINO/SAM front end is outside this build; pass object_finder=")
        imgs, boxes, masks, phrases = self.object_finder(rgb_image_path, consider_floor)
        if imgs is None:
            return None, None, None
        rgb = self.load_rgb_image_func(rgb_image_path)
        depth = self.load_depth_image_func(depth_image_path)
        embs = np.stack([np.array(self.get_embeddings_func(
            current_obj_grounded_img=imgs[i], current_obj_bounding_box=boxes[i], current_obj_mask=masks[i],
            current_obj_phrase=phrases[i], full_rgb_image=rgb, full_depth_image=depth, consider_floor=consider_floor,
            device=self.device).cpu()) for i in range(len(imgs))])
        # depth + masks -> one coloured cloud per object, then radius outlier removal: both on the device
        # (utils/depth_utils.py:176-206 + :87-88); the clouds stay in HBM as a CloudBatch
        dev = torch.device(self.device if str(self.device) != "cuda" else "cuda:0")
        m_t = torch.stack([torch.as_tensor(np.asarray(m.cpu() if hasattr(m, "cpu") else m)).reshape(depth.shape[:2]) for m in masks]) \
            if len(masks) else torch.zeros((0,) + depth.shape[:2], dtype=torch.uint8)
        d_np = np.ascontiguousarray(depth)
        if d_np.dtype == np.uint16:
            d_t = torch.from_numpy(d_np.view(np.int16)).to(dev).view(torch.uint16)
        elif d_np.dtype == np.float32:
            d_t = torch.from_numpy(d_np).to(dev)
        else:
            d_t = torch.from_numpy(d_np.astype(np.float64)).to(dev)
        clouds = unproject_masks(self._ctx, d_t, torch.from_numpy(np.array(rgb, dtype=np.uint8)).to(dev), (m_t != 0).to(dev),
                                 self.camera_focal_lenth_x, self.camera_focal_lenth_y, depth_factor, want_f64=for_build)
        if for_build:
            clouds, p64, c64 = clouds
        keep = None
        if outlier_removal_config is not None and clouds.n > 0:
            keep = radius_outlier_batch(self._ctx, clouds, outlier_removal_config["radius"], outlier_removal_config["radius_nb_points"])
        if for_build:
            # memory build: (points, colours) per object on the host with the values of the reference's Open3D clouds (float32 or
            # float64 products by numpy's promotion, colours = float32 rgb / 255; utils/depth_utils.py:63-80)
            kh = keep.bool().cpu().numpy() if keep is not None else np.ones(clouds.n, dtype=bool)
            pts, cols = p64.cpu().numpy(), c64.cpu().numpy()
            off = clouds.seg_off_host
            out = []
            for i in range(len(masks)):
                k = kh[off[i]:off[i + 1]]
                out.append((pts[off[i]:off[i + 1]][k], cols[off[i]:off[i + 1]][k]))
            return phrases, embs, out
        if keep is not None:
            clouds = _compact(clouds, keep)
        return phrases, embs, clouds

    def localise_detections(self, detected_embs, detected_clouds, outlier_removal_config=None, fpfh_global_dist_factor=2,
                            fpfh_local_dist_factor=0.4, fpfh_voxel_size=0.05, max_detected_object_num=7):
        """The body of localise() after perception (object_memory.py:899-1131) for one frame: embeddings (Q, D) and
        clouds [(points, colors), ...] -> FrameResult."""
        if outlier_removal_config is None:
            outlier_removal_config = {"radius_nb_points": 8, "radius": 0.05}
        if isinstance(detected_clouds, CloudBatch):
            sizes = np.diff(detected_clouds.seg_off_host)
            order = list(range(len(sizes)))
            if len(order) > max_detected_object_num:                             # :900-908 keep the largest, reorder
                order = sorted(order, key=lambda i: int(sizes[i]), reverse=True)[:max_detected_object_num]
            det = detected_clouds if order == list(range(len(sizes))) else _select(detected_clouds, order)
            n_det = len(order)
        else:
            order = list(range(len(detected_clouds)))
            if len(detected_clouds) > max_detected_object_num:                   # :900-908 keep the largest, reorder
                order = sorted(order, key=lambda i: len(detected_clouds[i][0]), reverse=True)[:max_detected_object_num]
            clouds = [detected_clouds[i] for i in order]
            det = CloudBatch.from_numpy([c[0] for c in clouds], [intensity_from_colors(c[1]) for c in clouds], device=self.device)
            n_det = len(clouds)
        embs = np.asarray(detected_embs, dtype=np.float32)[order]
        self._n_queries += 1
        eng = self._get_engine()
        return eng.localise_batch(det, [n_det], det_emb=embs, fpfh_voxel_size=fpfh_voxel_size,
                                  fpfh_global_dist_factor=fpfh_global_dist_factor, fpfh_local_dist_factor=fpfh_local_dist_factor,
                                  outlier_radius=outlier_removal_config["radius"], outlier_nb_points=outlier_removal_config["radius_nb_points"],
                                  seed=self.ransac_seed, job_id_base=16 * self._n_queries)[0]

    def localise(self, image_path, depth_image_path, testname="", subtest_name="", save_point_clouds=False,
                 outlier_removal_config=None, fpfh_global_dist_factor=2, fpfh_local_dist_factor=0.4, fpfh_voxel_size=0.05,
                 topK=5, useLora=True, save_localised_pcd_path=None, consider_floor=False, perform_semantic_icp=True,
                 depth_factor=1., max_detected_object_num=7):
        if outlier_removal_config is None:
            outlier_removal_config = {"radius_nb_points": 8, "radius": 0.05}
        consider_floor = False                                                        # :886
        phrases, embs, clouds = self._get_object_info(image_path, depth_image_path, consider_floor, outlier_removal_config,
                                                      depth_factor)
        if embs is None:
            return np.array([0., 0., 0., 0., 0., 0., 1.]), [[], []]                   # :895-896
        if perform_semantic_icp:
            raise NotImplementedError                                                  # :1039-1040 (every driver passes False)
        save_root = f"pcds/{testname}/"                                               # side effect of the reference (:947-950)
        if not os.path.exists(save_root):
            os.makedirs(save_root)
        if save_point_clouds:
            # :952-966 writes the detected clouds and the memory into one "_init_pcd_<subtest>.ply".  (The reference adds the
            # ObjectInfo records themselves to an Open3D cloud there -- `init_pcd += m`, :961-962 -- which Open3D rejects; what
            # is written here is what that block is after: every detected and every memory point, colours kept.)
            from .object_info import write_ply
            subsave_root = os.path.join(save_root, str(subtest_name))
            os.makedirs(subsave_root, exist_ok=True)
            if isinstance(clouds, CloudBatch):
                p4 = clouds.pts4.cpu().numpy()
                det_pts = [p4[:, :3].astype(np.float64)]
                det_cols = [np.repeat(p4[:, 3:4].astype(np.float64), 3, axis=1)]          # the batch keeps the intensity only
            else:
                det_pts, det_cols = [np.asarray(c[0], dtype=np.float64) for c in clouds], [np.asarray(c[1], dtype=np.float64) for c in clouds]
            mem_pts = [np.asarray(m.pointcloud.points, dtype=np.float64) for m in self.memory]
            mem_cols = [np.asarray(m.pointcloud.colors, dtype=np.float64) if len(np.asarray(m.pointcloud.colors)) == len(np.asarray(m.pointcloud.points))
                        else np.zeros((len(np.asarray(m.pointcloud.points)), 3)) for m in self.memory]
            write_ply(os.path.join(subsave_root, "_init_pcd_" + str(subtest_name) + ".ply"), np.concatenate(det_pts + mem_pts),
                      np.concatenate(det_cols + mem_cols))
        res = self.localise_detections(embs, clouds, outlier_removal_config, fpfh_global_dist_factor, fpfh_local_dist_factor,
                                       fpfh_voxel_size, max_detected_object_num)
        last = res.assignments[-1] if res.assignments else []
        if save_point_clouds and res.records:
            self._save_registration_dumps(res, clouds, det_pts, det_cols, mem_pts, mem_cols, outlier_removal_config, max_detected_object_num,
                                          subsave_root, image_path)
        return res.pose, [last, None]                                                 # :1169 (assn of the last loop iteration)

    def _save_registration_dumps(self, res, clouds, det_pts, det_cols, mem_pts, mem_cols, outlier_cfg, max_det, subsave_root, image_path):
        """The debugging artefacts of localise(save_point_clouds=True) after the registrations (object_memory.py:1092-1093, 1142-1165):
        per assignment `only_chosen_<assn>.ply` = the chosen memory clouds + the chosen (cleaned) detected clouds moved by the
        assignment's transform, both centred as the registration saw them; `_best_full_pcd<assn>.ply` = the whole memory (green) + all
        detections (red, outlier-filtered once more) moved by the best transform, centred with the means of the LAST assignment (the
        stale means of :1127, App. B); and a copy of the query image.  Host-side file writing, outside the hot path."""
        import shutil
        from .object_info import write_ply
        # the detections in the order and cleaned state the registration used (:900-908, :992-998)
        if isinstance(clouds, CloudBatch):
            off = clouds.seg_off_host
            dets = [(det_pts[0][off[i]:off[i + 1]], det_cols[0][off[i]:off[i + 1]]) for i in range(len(off) - 1)]
        else:
            dets = list(zip(det_pts, det_cols))
        if len(dets) > max_det:
            dets = [dets[i] for i in sorted(range(len(dets)), key=lambda i: len(dets[i][0]), reverse=True)[:max_det]]
        keep = radius_outlier_batch(self._ctx, CloudBatch.from_numpy([d[0] for d in dets], device=self.device), outlier_cfg["radius"],
                                    outlier_cfg["radius_nb_points"]).bool().cpu().numpy()
        o = np.concatenate([[0], np.cumsum([len(d[0]) for d in dets])])
        cleaned = [(d[0][keep[o[i]:o[i + 1]]], d[1][keep[o[i]:o[i + 1]]]) for i, d in enumerate(dets)]

        def moved(points, T):
            return points @ T[:3, :3].T + T[:3, 3]

        for rec in res.records:
            assn = rec["assn"]
            dp = np.concatenate([cleaned[d][0] for d, m in assn]) - rec["detected_mean"]
            dc = np.concatenate([cleaned[d][1] for d, m in assn])
            mp = np.concatenate([mem_pts[m] for d, m in assn]) - rec["memory_mean"]
            mc = np.concatenate([mem_cols[m] for d, m in assn])
            write_ply(os.path.join(subsave_root, "only_chosen_" + str(assn) + ".ply"), np.concatenate([mp, moved(dp, rec["T"])]),
                      np.concatenate([mc, dc]))
        best, lastrec = res.records[res.best], res.records[-1]
        all_det = np.concatenate([c[0] for c in cleaned]) - lastrec["detected_mean"]
        all_mem = np.concatenate(mem_pts) - lastrec["memory_mean"]
        k2 = radius_outlier_batch(self._ctx, CloudBatch.from_numpy([all_det], device=self.device), outlier_cfg["radius"],
                                  outlier_cfg["radius_nb_points"]).bool().cpu().numpy()
        green, red = np.tile([0.0, 1.0, 0.0], (len(all_mem), 1)), np.tile([1.0, 0.0, 0.0], (int(k2.sum()), 1))
        write_ply(os.path.join(subsave_root, "_best_full_pcd" + str(best["assn"]) + ".ply"),
                  np.concatenate([all_mem, moved(all_det[k2], best["T"])]), np.concatenate([green, red]))
        if image_path is not None and os.path.exists(str(image_path)):
            shutil.copy(str(image_path), os.path.join(subsave_root, "rgb_image." + str(image_path).split(".")[-1]))
