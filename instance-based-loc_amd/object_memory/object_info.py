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

    def _add_names(self, new_names):
        for n in new_names:
            self._add_name(n)

    def _add_embedding(self, new_emb):
        """object_info.py:40-59: below the cap append; at the cap the second-nearest stored embedding is replaced when its own
        nearest neighbour is closer than its distance to the new one (sklearn NearestNeighbors, as the reference)."""
        if len(self.embeddings) < self.max_embeddings_num:
            self.embeddings.append(new_emb)
            return
        from sklearn.neighbors import NearestNeighbors
        arr = np.array(self.embeddings)
        knn = NearestNeighbors(n_neighbors=2, metric="euclidean")
        knn.fit(arr)
        distances, indices = knn.kneighbors(np.asarray(new_emb).reshape(1, -1), n_neighbors=2)
        j = indices[0][1]
        if knn.kneighbors(arr[j].reshape(1, -1), n_neighbors=1)[0][0][0] < distances[0][1]:
            self.embeddings[j] = new_emb

    def _add_embeddings(self, new_embs):
        self.embeddings += new_embs

    def _add_pointcloud(self, new_pointcloud):
        pts = np.vstack((np.asarray(self.pointcloud.points), np.asarray(new_pointcloud.points)))
        cols = None
        if self.pointcloud.colors is not None and new_pointcloud.colors is not None:
            cols = np.vstack((np.asarray(self.pointcloud.colors), np.asarray(new_pointcloud.colors)))
        self.pointcloud = Cloud(pts, cols)
        self._process_pointcloud()

    def __add__(self, other):
        self._add_names(other.names)
        self._add_embeddings(other.embeddings)
        self._add_pointcloud(other.pointcloud)
        return self                      # means are NOT refreshed here (object_info.py:89-93); callers do it where the reference does

    def downsample(self, voxel_size, ctx=None):
        """object_info.py:95-97 -> utils/depth_utils.py:211-265, on the device (bit-identical; ObjectMemory.downsample_all_objects
        does all objects in one call)."""
        from ibloc_amd.build import default_ctx, voxel_downsample_batch
        pts, cols = voxel_downsample_batch(ctx or default_ctx(), [self.pointcloud.points],
                                           [self.pointcloud.colors] if self.pointcloud.colors is not None else None, voxel_size)
        self.pointcloud = Cloud(pts[0], cols[0] if cols is not None else None)
        self._process_pointcloud()

    def add_info(self, new_name, new_emb, new_pointcloud, align=False, max_iteration=30, max_correspondence_distance=0.01):
        if align:
            raise NotImplementedError("Aligning is a To-Do")
        self._add_name(new_name)
        self._add_embedding(new_emb)
        self._add_pointcloud(new_pointcloud if isinstance(new_pointcloud, Cloud) else Cloud(np.asarray(new_pointcloud.points), np.asarray(new_pointcloud.colors)))
        self._compute_means()

    def update_pointcloud_with_mask(self, mask):
        mask = np.asarray(mask)
        cols = self.pointcloud.colors[mask, :] if self.pointcloud.colors is not None else None
        self.pointcloud = Cloud(self.pointcloud.points[mask, :], cols)
        self._process_pointcloud()

    # ---- persistence: the reference's per-object directory layout (object_info.py:109-118) --------------------------
    def save(self, save_directory: str):
        """`pointcloud.ply` (binary little-endian, double x/y/z + uchar red/green/blue -- what Open3D's write_point_cloud
        emits for a coloured cloud) and `info.pkl` with names / embeddings / max_embeddings_num."""
        import os
        import pickle
        os.makedirs(save_directory, exist_ok=True)
        write_ply(os.path.join(save_directory, "pointcloud.ply"), self.pointcloud.points, self.pointcloud.colors)
        with open(os.path.join(save_directory, "info.pkl"), "wb") as f:
            pickle.dump({"names": self.names, "embeddings": self.embeddings, "max_embeddings_num": self.max_embeddings_num}, f)

    @classmethod
    def load(cls, load_directory: str, id: int = 0):
        import os
        import pickle
        pts, cols = read_ply(os.path.join(load_directory, "pointcloud.ply"))
        with open(os.path.join(load_directory, "info.pkl"), "rb") as f:
            info = pickle.load(f)
        obj = cls(id, info["names"][0], info["embeddings"][0], Cloud(pts, cols), info.get("max_embeddings_num", 1000000))
        obj.names = list(info["names"])
        obj.embeddings = [np.asarray(e) for e in info["embeddings"]]
        obj._compute_means()
        return obj


_PLY_TYPES = {"char": "i1", "uchar": "u1", "int8": "i1", "uint8": "u1", "short": "i2", "ushort": "u2", "int16": "i2", "uint16": "u2",
              "int": "i4", "uint": "u4", "int32": "i4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def write_ply(path, points, colors=None):
    """Vertex-only PLY, binary little-endian: double x, y, z (+ uchar red, green, blue from colours in [0, 1])."""
    pts = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    fields = [("x", "<f8"), ("y", "<f8"), ("z", "<f8")]
    if colors is not None and len(colors) == len(pts):
        fields += [("red", "u1"), ("green", "u1"), ("blue", "u1")]
    rec = np.zeros(len(pts), dtype=fields)
    rec["x"], rec["y"], rec["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
    if len(fields) == 6:
        c8 = np.rint(np.clip(np.asarray(colors, dtype=np.float64), 0.0, 1.0) * 255.0).astype(np.uint8)
        rec["red"], rec["green"], rec["blue"] = c8[:, 0], c8[:, 1], c8[:, 2]
    names = {"<f8": "double", "u1": "uchar"}
    header = ["ply", "format binary_little_endian 1.0", "comment written by ibloc_amd", f"element vertex {len(pts)}"]
    header += [f"property {names[t]} {n}" for n, t in fields] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(rec.tobytes())


def read_ply(path):
    """-> (points (N, 3) float64, colors (N, 3) float64 in [0, 1] or None).  Vertex element only; binary little-endian or ascii."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, n, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties on vertices are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt == "binary_little_endian":
            rec = np.frombuffer(f.read(n * np.dtype([(p, "<" + t) for p, t in props]).itemsize), dtype=[(p, "<" + t) for p, t in props])
        elif fmt == "ascii":
            rows = np.loadtxt(f, max_rows=n, ndmin=2) if n else np.zeros((0, len(props)))
            rec = {p: rows[:, i] for i, (p, _) in enumerate(props)}
        else:
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
    pts = np.stack([np.asarray(rec[k], dtype=np.float64) for k in ("x", "y", "z")], axis=1) if n else np.zeros((0, 3))
    cols = None
    if all(k in dict(props) for k in ("red", "green", "blue")):
        scale = 255.0 if dict(props)["red"] == "u1" else 1.0
        cols = np.stack([np.asarray(rec[k], dtype=np.float64) for k in ("red", "green", "blue")], axis=1) / scale if n else np.zeros((0, 3))
    return pts, cols
