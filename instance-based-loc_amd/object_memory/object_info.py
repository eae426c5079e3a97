"""Per-instance memory record, mirroring /root/reference/object_memory/object_info.py:7-118 (data layout
only: names, list of embeddings, point cloud, mean embedding, centroid; `__add__` merges)."""
import numpy as np

from ibloc_amd.utils.fpfh_register import Cloud


class ObjectInfo:
    def __init__(self, id: int, name: str, emb: np.ndarray, pointcloud, max_embeddings_num: int = 1000000):
        self.id = id
        self.names = [name]
        self.embeddings = [np.asarray(emb)]
        self.pointcloud = pointcloud if isinstance(pointcloud, Cloud) else Cloud(np.asarray(pointcloud.points), np.asarray(pointcloud.colors))
        self.max_embeddings_num = max_embeddings_num
        self._process_pointcloud()
        self.mean_emb = None
        self.centroid = None
        self._compute_means()

    def __repr__(self):
        return f"ObjectInfo == ID: {self.id}, Names: {self.names}, Mean_Emb: {self.mean_emb.shape}, Num. Points: {self.pcd.shape}"

    def _process_pointcloud(self):
        self.pcd = np.asarray(self.pointcloud.points).T
        self.pcd_colors = np.asarray(self.pointcloud.colors).T if self.pointcloud.colors is not None else None

    def _compute_means(self):
        self.mean_emb = np.mean(np.array(self.embeddings), axis=0).squeeze()
        self.centroid = np.mean(self.pcd, axis=-1)

    def _add_name(self, new_name):
        if new_name not in self.names:
            self.names.append(new_name)

    def __add__(self, other):
        for n in other.names:
            self._add_name(n)
        self.embeddings += other.embeddings
        pts = np.vstack((np.asarray(self.pointcloud.points), np.asarray(other.pointcloud.points)))
        cols = None
        if self.pointcloud.colors is not None and other.pointcloud.colors is not None:
            cols = np.vstack((np.asarray(self.pointcloud.colors), np.asarray(other.pointcloud.colors)))
        self.pointcloud = Cloud(pts, cols)
        self._process_pointcloud()
        return self
