"""ObjectInfo directory layout (reference object_info.py:109-118): PLY (binary little-endian double xyz + uchar rgb) + info.pkl."""
import os
import pickle
import struct

import numpy as np

from ibloc_amd.object_memory.object_info import ObjectInfo, read_ply, write_ply
from ibloc_amd.utils.fpfh_register import Cloud


def test_ply_round_trip_and_layout(tmp_path):
    rng = np.random.default_rng(0)
    pts = rng.normal(size=(37, 3))
    cols = rng.uniform(size=(37, 3))
    p = tmp_path / "c.ply"
    write_ply(p, pts, cols)
    raw = open(p, "rb").read()
    head, body = raw.split(b"end_header\n")
    assert b"format binary_little_endian 1.0" in head and b"property double x" in head and b"property uchar blue" in head
    assert len(body) == 37 * (3 * 8 + 3)
    assert struct.unpack_from("<ddd", body, 0) == tuple(pts[0])                    # exact doubles, little endian
    q, c = read_ply(p)
    assert np.array_equal(q, pts) and np.abs(c - cols).max() <= 0.5 / 255 + 1e-12
    write_ply(p, pts[:0], None)
    q, c = read_ply(p)
    assert q.shape == (0, 3) and c is None
    # ascii PLY with float properties and an extra column, as other tools write them
    with open(p, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 2\nproperty float x\nproperty float y\nproperty float z\nproperty float nx\n"
                "end_header\n1 2 3 0\n4 5 6 1\n")
    q, c = read_ply(p)
    assert np.array_equal(q, [[1, 2, 3], [4, 5, 6]]) and c is None


def test_object_info_directory_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    a = ObjectInfo(3, "chair", rng.normal(size=8), Cloud(rng.normal(size=(20, 3)), rng.uniform(size=(20, 3))))
    b = ObjectInfo(4, "seat", rng.normal(size=8), Cloud(rng.normal(size=(5, 3)), rng.uniform(size=(5, 3))))
    a = a + b
    a._compute_means()
    a.save(str(tmp_path / "obj"))
    assert sorted(os.listdir(tmp_path / "obj")) == ["info.pkl", "pointcloud.ply"]
    info = pickle.load(open(tmp_path / "obj" / "info.pkl", "rb"))
    assert sorted(info) == ["embeddings", "max_embeddings_num", "names"] and info["names"] == ["chair", "seat"]
    c = ObjectInfo.load(str(tmp_path / "obj"), id=9)
    assert c.id == 9 and c.names == ["chair", "seat"] and len(c.embeddings) == 2
    assert np.array_equal(c.pointcloud.points, a.pointcloud.points) and c.pcd.shape == (3, 25)
    assert np.allclose(c.mean_emb, a.mean_emb) and np.allclose(c.centroid, a.centroid)
