"""Host side of crop preprocessing: builds the fixed-point resample tables and crop descriptors
that `ibl_preprocess_crops` (csrc/preprocess.hip) consumes.

The per-model recipes mirror what the reference's embedding functions do on the CPU:
  * DINOv2  (utils/embeddings.py:64-65): cv2 BGR2RGB channel swap, HF BitImageProcessor of
    facebook/dinov2-base: shortest edge -> 256 (bicubic), centre crop 224, x/255, ImageNet mean/std.
  * ViT     (utils/embeddings.py:86-89): swap, ViTFeatureExtractor of google/vit-base-patch16-224-in21k:
    resize to 224x224 (bilinear), x/255, mean = std = 0.5.
  * CLIP    (utils/embeddings.py:41-42): swap, open_clip preprocess: shortest edge -> 224 (bicubic),
    centre crop 224, CLIP mean/std.
  * DATOR   (dator/get_embeds.py:80-87): resize to 256x128 (H x W, bilinear), mean = std = 0.5.

The coefficient tables are Pillow's (`precompute_coeffs` + `normalize_coeffs_8bpc` of
libImaging/Resample.c, restated here in float64 numpy): for every output sample a first tap, a tap
count and 22-bit fixed-point weights.  The device then does pure integer arithmetic, which makes the
u8 result bit-identical to PIL.Image.resize (checked against PIL in tests/).
"""
import ctypes as C
import math
from dataclasses import dataclass

import numpy as np

PRECISION_BITS = 32 - 8 - 2

BICUBIC = "bicubic"
BILINEAR = "bilinear"


def _bicubic_filter(x):
    a = -0.5
    x = np.abs(x)
    # (both polynomials on every element, then a select: same float64 values as the masked assignment, several times faster)
    p1 = ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    p2 = (((x - 5) * x + 8) * x - 4) * a
    return np.where(x < 1.0, p1, np.where(x < 2.0, p2, 0.0))


def _bilinear_filter(x):
    x = np.abs(x)
    return np.where(x < 1.0, 1.0 - x, 0.0)


_FILTERS = {BICUBIC: (_bicubic_filter, 2.0), BILINEAR: (_bilinear_filter, 1.0)}


def resample_table(in_size: int, out_size: int, filt: str, win0: int, win_n: int):
    """Fixed-point table for output samples [win0, win0 + win_n) of a resize in_size -> out_size.

    Returns (records int32 [win_n, 2 + ksize], ksize)."""
    fn, fsupport = _FILTERS[filt]
    scale = float(in_size) / float(out_size)
    filterscale = scale if scale >= 1.0 else 1.0
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    # all output samples at once (round 4: the per-sample Python loop cost 6 ms per crop -- a C1 step of 128 differently sized crops
    # spent 0.7 s on the host building tables); the arithmetic is the same float64 sequence per sample, including the left-to-right
    # sum of the weights (np.add.accumulate adds in order; the zero padding behind a row's taps does not change it)
    rec = np.zeros((win_n, 2 + ksize), dtype=np.int32)
    center = (win0 + np.arange(win_n, dtype=np.float64) + 0.5) * scale
    xmin = (center - support + 0.5).astype(np.int64)            # C's (int) cast: truncation toward zero
    xmin[xmin < 0] = 0
    xmax = (center + support + 0.5).astype(np.int64)
    xmax[xmax > in_size] = in_size
    xmax -= xmin
    xs = np.arange(ksize, dtype=np.float64)[None, :]
    taps = xs < xmax[:, None]
    w = fn((xs + xmin[:, None] - center[:, None] + 0.5) * ss) * taps
    ww = np.add.accumulate(w, axis=1)[:, -1]
    nz = ww != 0.0
    w[nz] = w[nz] / ww[nz, None]
    k = np.where(w < 0, (-0.5 + w * (1 << PRECISION_BITS)).astype(np.int64), (0.5 + w * (1 << PRECISION_BITS)).astype(np.int64))
    rec[:, 0] = xmin
    rec[:, 1] = xmax
    rec[:, 2:] = np.where(taps, k, 0)
    return rec, ksize


@dataclass(frozen=True)
class PreprocessRecipe:
    name: str
    out_h: int
    out_w: int
    resize_mode: str          # "shortest" (keep aspect, then centre crop) or "exact" (resize to out_h x out_w)
    shortest: int             # target of the shortest edge for "shortest"
    filt: str
    swap_rb: bool
    mean: tuple
    std: tuple
    crop_rounding: str = "floor"   # HF center_crop: (orig - crop) // 2 ; torchvision: int(round(.../2))


IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)

RECIPES = {
    "dinov2": PreprocessRecipe("dinov2", 224, 224, "shortest", 256, BICUBIC, True, IMAGENET_MEAN, IMAGENET_STD),
    "vit": PreprocessRecipe("vit", 224, 224, "exact", 0, BILINEAR, True, (0.5, 0.5, 0.5), (0.5, 0.5, 0.5)),
    "clip": PreprocessRecipe("clip", 224, 224, "shortest", 224, BICUBIC, True, CLIP_MEAN, CLIP_STD, "round"),
    "dator_rgb": PreprocessRecipe("dator_rgb", 256, 128, "exact", 0, BILINEAR, False, (0.5, 0.5, 0.5), (0.5, 0.5, 0.5)),
}


def resized_size(recipe: PreprocessRecipe, in_h: int, in_w: int):
    """(resized_h, resized_w, top, left) of the resize + centre-crop window for one crop."""
    if recipe.resize_mode == "exact":
        return recipe.out_h, recipe.out_w, 0, 0
    short, long = (in_h, in_w) if in_h <= in_w else (in_w, in_h)
    new_short = recipe.shortest
    new_long = int(recipe.shortest * long / short)
    rh, rw = (new_short, new_long) if in_h <= in_w else (new_long, new_short)
    if recipe.crop_rounding == "floor":
        top, left = (rh - recipe.out_h) // 2, (rw - recipe.out_w) // 2
    else:
        top, left = int(round((rh - recipe.out_h) / 2.0)), int(round((rw - recipe.out_w) / 2.0))
    if top < 0 or left < 0:
        raise ValueError("crop window larger than the resized image is not supported")
    return rh, rw, top, left


class CropDesc(C.Structure):
    _fields_ = [("src_offset", C.c_int64), ("in_h", C.c_int32), ("in_w", C.c_int32), ("h_table", C.c_int32),
                ("h_ksize", C.c_int32), ("v_table", C.c_int32), ("v_ksize", C.c_int32), ("tmp_offset", C.c_int64)]


_TABLES = {}            # (in_size, out_size, filter, window) -> (records, ksize): crop sizes recur from batch to batch


def resample_tables_native(keys):
    """`resample_table` through the library's host function `ibl_resample_table` (csrc/resample.cpp: the same float64 arithmetic in C++,
    microseconds per table); the numpy builders below stay as the checker the CPU tests compare it with."""
    import ctypes
    from . import _lib
    out = {}
    for key in keys:
        in_size, out_size, filt, win0, win_n = key
        f = 1 if filt == BICUBIC else 0
        ksize = _lib.lib.ibl_resample_ksize(in_size, out_size, f)
        rec = np.empty((win_n, 2 + ksize), dtype=np.int32)
        st = _lib.lib.ibl_resample_table(in_size, out_size, f, win0, win_n, rec.ctypes.data_as(ctypes.c_void_p))
        if st != ksize:
            _lib.check(st if st < 0 else -1, "ibl_resample_table")
        out[key] = (rec, ksize)
    return out


def resample_tables(keys):
    """`resample_table` for many (in_size, out_size, filt, win0, win_n) keys in one vectorised pass per (filter, window length): the
    tables of a batch of differently sized crops (two per crop) cost one numpy expression instead of one per table.  Same float64
    arithmetic per sample (rows are padded to the widest support of the group; padding taps are masked out before the sum)."""
    out = {}
    groups = {}
    for key in keys:
        groups.setdefault((key[2], key[4]), []).append(key)
    for (filt, win_n), ks in groups.items():
        fn, fsupport = _FILTERS[filt]
        in_size = np.array([k[0] for k in ks], dtype=np.float64)[:, None]
        out_size = np.array([k[1] for k in ks], dtype=np.float64)[:, None]
        win0 = np.array([k[3] for k in ks], dtype=np.float64)[:, None]
        scale = in_size / out_size
        filterscale = np.where(scale >= 1.0, scale, 1.0)
        support = fsupport * filterscale
        ksize = np.ceil(support).astype(np.int64)[:, 0] * 2 + 1
        kmax = int(ksize.max())
        ss = 1.0 / filterscale
        center = (win0 + np.arange(win_n, dtype=np.float64)[None, :] + 0.5) * scale            # [T][win_n]
        xmin = (center - support + 0.5).astype(np.int64)
        xmin[xmin < 0] = 0
        xmax = (center + support + 0.5).astype(np.int64)
        xmax = np.minimum(xmax, in_size.astype(np.int64))
        xmax -= xmin
        xs = np.arange(kmax, dtype=np.float64)[None, None, :]
        taps = xs < xmax[:, :, None]
        w = fn((xs + xmin[:, :, None] - center[:, :, None] + 0.5) * ss[:, :, None]) * taps
        ww = np.add.accumulate(w, axis=2)[:, :, -1]
        w = np.where((ww != 0.0)[:, :, None], w / np.where(ww != 0.0, ww, 1.0)[:, :, None], w)
        k = np.where(w < 0, (-0.5 + w * (1 << PRECISION_BITS)).astype(np.int64), (0.5 + w * (1 << PRECISION_BITS)).astype(np.int64))
        k = np.where(taps, k, 0)
        for t, key in enumerate(ks):
            rec = np.zeros((win_n, 2 + int(ksize[t])), dtype=np.int32)
            rec[:, 0] = xmin[t]
            rec[:, 1] = xmax[t]
            rec[:, 2:] = k[t, :, :int(ksize[t])]
            out[key] = (rec, int(ksize[t]))
    return out


def _cached_table(in_size, out_size, filt, win0, win_n):
    key = (in_size, out_size, filt, win0, win_n)
    t = _TABLES.get(key)
    if t is None:
        t = resample_table(in_size, out_size, filt, win0, win_n)
        if len(_TABLES) < 8192:
            _TABLES[key] = t
    return t


class TableCache:
    """Packs resample tables of a batch into one int32 array, sharing identical tables."""

    def __init__(self):
        self.chunks = []
        self.size = 0
        self.index = {}

    def get(self, in_size, out_size, filt, win0, win_n):
        key = (in_size, out_size, filt, win0, win_n)
        if key in self.index:
            return self.index[key]
        if in_size == out_size:
            val = (win0, 0)               # identity pass: table field carries the window start
        else:
            rec, ksize = _cached_table(in_size, out_size, filt, win0, win_n)
            val = (self.size, ksize)
            self.chunks.append(rec.reshape(-1))
            self.size += rec.size
        self.index[key] = val
        return val

    def packed(self):
        if not self.chunks:
            return np.zeros(1, dtype=np.int32)
        return np.concatenate(self.chunks).astype(np.int32)


def plan_batch(recipe: PreprocessRecipe, shapes):
    """shapes: list of (H, W).  Returns (descs ndarray of CropDesc, tables int32, src_bytes, tmp_bytes, max_h)."""
    cache = TableCache()
    # every table the batch needs and the process has not built yet, in one vectorised pass
    want = set()
    for (h, w) in shapes:
        rh, rw, top, left = resized_size(recipe, h, w)
        for key in ((w, rw, recipe.filt, left, recipe.out_w), (h, rh, recipe.filt, top, recipe.out_h)):
            if key[0] != key[1] and key not in _TABLES:
                want.add(key)
    if want:
        built = resample_tables_native(sorted(want))
        if len(_TABLES) + len(built) > 8192:
            _TABLES.clear()
        _TABLES.update(built)
    descs = (CropDesc * len(shapes))()
    src_off = 0
    tmp_off = 0
    max_h = 0
    for i, (h, w) in enumerate(shapes):
        rh, rw, top, left = resized_size(recipe, h, w)
        ht, hk = cache.get(w, rw, recipe.filt, left, recipe.out_w)
        vt, vk = cache.get(h, rh, recipe.filt, top, recipe.out_h)
        d = descs[i]
        d.src_offset = src_off
        d.in_h, d.in_w = h, w
        d.h_table, d.h_ksize = ht, hk
        d.v_table, d.v_ksize = vt, vk
        d.tmp_offset = tmp_off
        src_off += h * w * 3
        tmp_off += h * recipe.out_w * 3
        max_h = max(max_h, h)
    return descs, cache.packed(), src_off, tmp_off, max_h
