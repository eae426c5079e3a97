"""Drop-in for the reference's `utils/depth_utils.py` (/root/reference/utils/depth_utils.py) on the MI355X build: the same function
names and arguments, `Cloud` records (`.points`, `.colors` as (N, 3) float64, like the Open3D clouds the reference returns) instead of
`open3d.geometry.PointCloud`.  Unprojection, radius outlier removal and voxel down-sampling run in libibloc_hip.so
(`ibl_unproject_masks_f64`, `ibl_radius_outlier_batch`, `ibl_voxel_downsample_batch`); pose arithmetic stays numpy / scipy like the
reference's.  Masks are read as booleans (the reference multiplies the depth image by the mask, :200-201, which is the same thing
for the 0/1 masks SAM produces)."""
import numpy as np
import torch
from scipy.spatial.transform import Rotation

from ibloc_amd.build import default_ctx, transform_points, voxel_downsample_batch
from ibloc_amd.registration import radius_outlier_batch, unproject_masks
from ibloc_amd.utils.fpfh_register import Cloud, _as_cloud

DEFAULT_OUTLIER_REMOVAL_CONFIG = {"radius_nb_points": 12, "radius": 0.05}      # :5-9


def _depth_tensor(depth_image, dev):
    d = np.ascontiguousarray(depth_image)
    if d.dtype == np.uint16:
        return torch.from_numpy(d.view(np.int16)).to(dev).view(torch.uint16)
    if d.dtype == np.float32:
        return torch.from_numpy(d).to(dev)
    return torch.from_numpy(d.astype(np.float64)).to(dev)


def _clouds(depth_image, rgb_image, masks, fx, fy, outlier_removal_config, coloured, device="cuda:0"):
    ctx = default_ctx()
    H, W = np.asarray(depth_image).shape[:2]
    rgb = np.zeros((H, W, 3), np.uint8) if rgb_image is None else np.asarray(rgb_image, dtype=np.uint8)
    assert rgb.shape[:2] == (H, W), "Depth and RGB image dimensions do not match"
    m = torch.stack([torch.as_tensor(np.asarray(x.cpu() if hasattr(x, "cpu") else x)).reshape(H, W) for x in masks]) != 0
    batch, p64, c64 = unproject_masks(ctx, _depth_tensor(depth_image, device), torch.from_numpy(rgb).to(device), m.to(device), fx, fy, 1.0,
                                      want_f64=True)
    keep = np.ones(batch.n, dtype=bool)
    if outlier_removal_config is not None and batch.n > 0:
        keep = radius_outlier_batch(ctx, batch, outlier_removal_config["radius"], outlier_removal_config["radius_nb_points"]).bool().cpu().numpy()
    pts, cols, off = p64.cpu().numpy(), c64.cpu().numpy(), batch.seg_off_host
    return [Cloud(pts[off[i]:off[i + 1]][keep[off[i]:off[i + 1]]], cols[off[i]:off[i + 1]][keep[off[i]:off[i + 1]]] if coloured else None)
            for i in range(len(masks))]


def get_pointcloud_from_depth(depth_image, focal_lenth_x, focal_lenth_y, outlier_removal_config=DEFAULT_OUTLIER_REMOVAL_CONFIG):       # :11-44
    full = np.ones(np.asarray(depth_image).shape[:2], dtype=bool)
    return _clouds(depth_image, None, [full], focal_lenth_x, focal_lenth_y, outlier_removal_config, False)[0]


def get_coloured_pointcloud_from_depth(depth_image, rgb_image, focal_lenth_x, focal_lenth_y,
                                       outlier_removal_config=DEFAULT_OUTLIER_REMOVAL_CONFIG):                                          # :46-90
    full = np.ones(np.asarray(depth_image).shape[:2], dtype=bool)
    return _clouds(depth_image, rgb_image, [full], focal_lenth_x, focal_lenth_y, outlier_removal_config, True)[0]


def get_mask_pointclouds_from_depth(depth_image, masks, focal_length_x, focal_length_y, outlier_removal_config=DEFAULT_OUTLIER_REMOVAL_CONFIG):
    return _clouds(depth_image, None, list(masks), focal_length_x, focal_length_y, outlier_removal_config, False)                        # :146-174


def get_mask_coloured_pointclouds_from_depth(depth_image, rgb_image, masks, focal_length_x, focal_length_y,
                                             outlier_removal_config=DEFAULT_OUTLIER_REMOVAL_CONFIG):                                    # :176-206
    return _clouds(depth_image, rgb_image, list(masks), focal_length_x, focal_length_y, outlier_removal_config, True)


def transform_pointcloud(pointcloud, pose):                                                                                             # :92-116
    c = _as_cloud(pointcloud)
    return Cloud(transform_points(c.points, pose), c.colors)


def transform_pointcloud_kinect(pointcloud, pose):                                                                                      # :118-144
    c = _as_cloud(pointcloud)
    t, q = pose[:3], pose[3:]
    q /= np.linalg.norm(q)
    R = Rotation.from_quat(q).as_matrix()
    R2 = Rotation.from_euler('xyz', [0, np.pi, 0]).as_matrix()
    return Cloud((R @ R2 @ c.points.T).T - t, c.colors)


def voxel_down_sample_with_colors(pcd, voxel_size):                                                                                     # :211-265
    c = _as_cloud(pcd)
    if c.normals is not None and len(c.normals):
        raise NotImplementedError("voxel_down_sample_with_colors: clouds with normals are not supported by the device path")
    pts, cols = voxel_downsample_batch(default_ctx(), [c.points], [c.colors] if c.has_colors() else None, voxel_size)
    return Cloud(pts[0], cols[0] if cols is not None else None)


def combine_point_clouds(pcds):                                                                                                         # :268-272
    cs = [_as_cloud(p) for p in pcds]
    if not cs:
        return Cloud(np.zeros((0, 3)))
    cols = np.vstack([c.colors for c in cs]) if all(c.colors is not None for c in cs) else None
    return Cloud(np.vstack([c.points for c in cs]), cols)


def compute_center(point_cloud):                                                                                                        # :274-277
    return np.mean(np.asarray(_as_cloud(point_cloud).points), axis=0)


def decompose_pose_matrix(pose_matrix):                                                                                                 # :279-288
    return np.concatenate((pose_matrix[:3, 3], Rotation.from_matrix(pose_matrix[:3, :3]).as_quat()))
