"""Synthetic workload generator: object memories, query frames and ground-truth poses.

No dataset is reachable offline, so parity tests and bench.py draw from this seeded generator:
  * an object = union of 3-6 random boxes / ellipsoid shells (extent U[0.15, 0.6] m: desk-scale objects, so that
    5 000 surface points are dense enough -- > 8 neighbours within 5 cm -- to survive the radius-outlier removal
    localise() applies to every detection; SURVEY §8d's U[0.2, 1.5] m spread 5 000 points so thin that 85 % of every
    detection was deleted and the registration stage ran on ~700-point sources); points are sampled
    on the surfaces with N(0, (2 mm)^2) noise; colour = 0.5 + 0.5 sin(7 xyz + phi) (smooth, so the
    photometric ICP term is informative);
  * the memory places objects on a jittered grid in the world frame;
  * a query frame picks Q neighbouring objects, keeps the 60-90 % of each that faces the camera
    (half-space cut), re-samples and re-noises them, and expresses them in the camera frame of a
    random SE(3) pose -- that pose is the ground truth localise() must return.
"""
from dataclasses import dataclass

import numpy as np
from scipy.spatial.transform import Rotation


@dataclass
class Primitive:
    kind: int            # 0 box, 1 ellipsoid
    center: np.ndarray
    half: np.ndarray
    rot: np.ndarray      # 3x3


class SynthObject:
    def __init__(self, rng, world_center, extent=(0.15, 0.6), kinds=(0, 1)):
        n_prim = int(rng.integers(3, 7))
        self.prims = []
        spread = 0.4 * extent[1] / 1.5          # primitive centres stay inside the object's largest extent
        for _ in range(n_prim):
            ext = rng.uniform(extent[0], extent[1], size=3)
            self.prims.append(Primitive(int(kinds[int(rng.integers(0, len(kinds)))]), rng.uniform(-spread, spread, size=3), ext / 2,
                                        Rotation.random(random_state=rng).as_matrix()))
        self.world_center = np.asarray(world_center, dtype=np.float64)
        self.phi = rng.uniform(0, 2 * np.pi, size=3)
        areas = []
        for p in self.prims:
            a, b, c = p.half
            areas.append(8 * (a * b + b * c + a * c) if p.kind == 0 else 4 * np.pi * ((a * b) ** 1.6 + (a * c) ** 1.6 + (b * c) ** 1.6) ** (1 / 1.6) / 3 ** (1 / 1.6))
        self.area_w = np.asarray(areas) / np.sum(areas)

    def _sample_prim(self, p: Primitive, n, rng):
        if p.kind == 0:
            a, b, c = p.half
            fa = np.array([b * c, b * c, a * c, a * c, a * b, a * b])
            face = rng.choice(6, size=n, p=fa / fa.sum())
            u = rng.uniform(-1, 1, size=(n, 3)) * p.half
            ax = face // 2
            sign = np.where(face % 2 == 0, 1.0, -1.0)
            u[np.arange(n), ax] = sign * p.half[ax]
            local = u
        else:
            v = rng.normal(size=(n, 3))
            v /= np.linalg.norm(v, axis=1, keepdims=True)
            local = v * p.half
        return local @ p.rot.T + p.center

    def sample(self, n, rng, noise=0.002):
        """n surface points in the WORLD frame + colours in [0, 1]."""
        counts = rng.multinomial(n, self.area_w)
        pts = np.concatenate([self._sample_prim(p, c, rng) for p, c in zip(self.prims, counts) if c > 0])
        pts = pts + rng.normal(0, noise, size=pts.shape) + self.world_center
        col = 0.5 + 0.5 * np.sin(7 * pts + self.phi)
        return pts, col


class SynthWorld:
    """M objects on a jittered grid (spacing 2.5 m), E embeddings per instance of dimension D."""

    def __init__(self, M, pts_per_object=5000, E=4, D=768, seed=0, spacing=2.5, extent=(0.15, 0.6), sample_points=True, kinds=(0, 1)):
        rng = np.random.default_rng(seed)
        self.rng = rng
        self.M, self.E, self.D = M, E, D
        side = int(np.ceil(np.sqrt(M)))
        self.side, self.spacing = side, spacing
        self.objects = []
        for j in range(M):
            gx, gy = j % side, j // side
            c = np.array([gx * spacing, gy * spacing, 0.0]) + np.append(rng.uniform(-0.3, 0.3, size=2), rng.uniform(0.0, 1.0))
            self.objects.append(SynthObject(rng, c, extent, kinds))
        self.points = []
        self.colors = []
        for o in (self.objects if sample_points else []):       # embedding-only memories (BASELINE configs[3]) hold no clouds
            p, c = o.sample(pts_per_object, rng)
            self.points.append(p)
            self.colors.append(c)
        # instance embeddings: unit "identity" direction + per-view noise (E stored views each)
        base = rng.normal(size=(M, D))
        base /= np.linalg.norm(base, axis=1, keepdims=True)
        self.base_emb = base
        embs = base[:, None, :] + rng.normal(0, 0.35 / np.sqrt(D), size=(M, E, D))
        self.embeddings = embs.astype(np.float32)          # un-normalised, like the reference stores them

    def neighbours(self, j, q):
        """q object ids around object j (grid neighbourhood, includes j)."""
        gx, gy = j % self.side, j // self.side
        cand = []
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                x, y = gx + dx, gy + dy
                k = y * self.side + x
                if 0 <= x < self.side and 0 <= y < self.side and k < self.M:
                    cand.append(k)
        cand.sort(key=lambda k: (k != j, k))
        return cand[:q]

    def make_frame(self, rng, q=7, pts_per_object=5000, emb_noise=0.1, anchor=None, with_clouds=True):
        """Returns dict(ids, clouds (camera frame) list of (pts, cols), det_emb (q, D) fp32, pose T_wc 4x4)."""
        anchor = int(rng.integers(0, self.M)) if anchor is None else anchor
        ids = self.neighbours(anchor, q)
        centre = np.mean([self.objects[k].world_center for k in ids], axis=0)
        # camera somewhere around the group, looking roughly at it
        direction = rng.normal(size=3)
        direction[2] = abs(direction[2]) * 0.3
        direction /= np.linalg.norm(direction)
        R = Rotation.random(random_state=rng).as_matrix()
        t_wc = centre + direction * rng.uniform(2.0, 4.0) + rng.uniform(-1, 1, size=3) * 0.5
        T_wc = np.eye(4)
        T_wc[:3, :3], T_wc[:3, 3] = R, t_wc
        clouds = []
        for k in (ids if with_clouds else []):
            o = self.objects[k]
            frac = rng.uniform(0.6, 0.9)
            pts, col = o.sample(int(pts_per_object / frac * 1.05) + 16, rng)
            # visible part: the `frac` of points closest to the camera along the viewing direction
            depth = (pts - t_wc) @ (o.world_center - t_wc) / np.linalg.norm(o.world_center - t_wc)
            keep = np.argsort(depth, kind="stable")[:int(len(pts) * frac)]
            keep = np.sort(keep)[:pts_per_object]
            pw = pts[keep]
            pc = (pw - t_wc) @ R          # R^T (p - t)
            clouds.append((pc, col[keep]))
        det = self.base_emb[ids] + rng.normal(0, emb_noise / np.sqrt(self.D), size=(len(ids), self.D))
        return {"ids": ids, "clouds": clouds, "det_emb": det.astype(np.float32), "pose": T_wc}
