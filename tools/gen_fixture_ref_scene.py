#!/usr/bin/env python3
"""Copies DATA (no code) the reference ships into tests/golden/ref_scene/ -- the only reference-held inputs for the Open3D-backed
stages:
  * out/360_trial_with_floor/objects/{0,1,2}/pointcloud.ply -- a memory the reference saved (ObjectInfo.save, object_info.py:109-118):
    binary little-endian PLY, double x y z + uchar r g b; 11 209 / 53 968 / 9 842 points ("armchair", "armchair", "table",
    memory.txt).  The sibling info.pkl files are NOT read (untrusted pickles holding dummy [1, 2, 3] embeddings).
  * data/our-synthetic/360_basic_test: all eight of its 600 x 600 RGB-D views (float32 depth in metres, focal length 300,
    additional_information.txt) and all eight poses of poses.json (position + Euler angles, as the file stores them).
Run in the build container (the reference tree does not exist on the GPU box):  python tools/gen_fixture_ref_scene.py"""
import json
import os

import numpy as np
from PIL import Image

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_scene")


def read_ply(path):
    b = open(path, "rb").read()
    end = b.index(b"end_header\n") + len(b"end_header\n")
    hdr = b[:end].decode().splitlines()
    assert "format binary_little_endian 1.0" in hdr
    props = [l.split()[1:] for l in hdr if l.startswith("property")]
    assert props == [["double", "x"], ["double", "y"], ["double", "z"], ["uchar", "red"], ["uchar", "green"], ["uchar", "blue"]], props
    n = int([l for l in hdr if l.startswith("element vertex")][0].split()[-1])
    a = np.frombuffer(b, dtype=np.dtype([("p", "<f8", 3), ("c", "u1", 3)]), count=n, offset=end)
    return np.ascontiguousarray(a["p"]), np.ascontiguousarray(a["c"])


def main():
    os.makedirs(OUT, exist_ok=True)
    arrays = {}
    for i in range(3):
        p, c = read_ply(f"{REF}/out/360_trial_with_floor/objects/{i}/pointcloud.ply")
        arrays[f"obj{i}_xyz"], arrays[f"obj{i}_rgb"] = p, c
    arrays["names"] = np.array(["armchair", "armchair", "table"])
    np.savez_compressed(os.path.join(OUT, "memory_objects.npz"), **arrays)
    views = json.load(open(f"{REF}/data/our-synthetic/360_basic_test/poses.json"))["views"]
    pos = np.array([[v["position"][k] for k in "xyz"] for v in views], dtype=np.float64)
    eul = np.array([[v["rotation"][k] for k in "xyz"] for v in views], dtype=np.float64)
    keep = list(range(1, 9))        # round 3: all eight views (ground-truth poses of every one are asserted in the tests)
    va = {"position": pos, "euler_xyz_deg": eul, "view_ids": np.array(keep), "focal_length": np.float64(300.0)}
    for k in keep:
        va[f"depth{k}"] = np.load(f"{REF}/data/our-synthetic/360_basic_test/depth/view{k}.npy")
        va[f"rgb{k}"] = np.asarray(Image.open(f"{REF}/data/our-synthetic/360_basic_test/rgb/view{k}.png"))[:, :, :3].copy()
    np.savez_compressed(os.path.join(OUT, "views.npz"), **va)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
