"""ctypes binding of libibloc_hip.so (the C-ABI declared in include/ibloc.h).

The HIP library is the product path: if it is missing or fails to load this module raises --
there is no CPU fallback."""
import ctypes as C
import os

# torch first: its wheel bundles the HIP runtime (libamdhip64 / libhsa-runtime64) that owns the process's device state.
# Loading libibloc_hip.so before torch would pull a second runtime from /opt/rocm into the process and every launch
# from this library would then fail with hipErrorNoDevice.
import torch  # noqa: F401  (device memory and streams are torch's; see DESIGN.md "boundary")

_HERE = os.path.dirname(os.path.abspath(__file__))
# IBLOC_LIB selects another build of the same library (instrumented lab builds of tools/); never a different backend
LIB_PATH = os.environ.get("IBLOC_LIB") or os.path.join(_HERE, "libibloc_hip.so")


class IblError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise IblError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C instance-based-loc_amd/csrc`). There is no CPU fallback.")
    return C.CDLL(LIB_PATH)


lib = _load()

c_i32p = C.POINTER(C.c_int32)
c_u16p = C.POINTER(C.c_uint16)
vp = C.c_void_p

_SIGS = {
    "ibl_version": (C.c_int, []),
    "ibl_last_error": (C.c_char_p, []),
    "ibl_prof_enable": (C.c_int, [C.c_int]),
    "ibl_prof_read": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "ibl_normalize_rows": (C.c_int, [vp, vp, C.c_int64, C.c_int, vp]),
    "ibl_closest_similarity_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int64]),
    "ibl_closest_similarity": (C.c_int, [vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, C.c_int, vp, vp, vp,
                                         C.c_int64, vp]),
    "ibl_preprocess_crops": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       vp, vp, vp, vp, vp]),
    "ibl_vit_workspace_bytes": (C.c_int64, [vp, C.c_int]),
    "ibl_linear_f16": (C.c_int, [vp, C.c_int64, vp, C.c_int64, vp, vp, C.c_int64, C.c_int, C.c_int, C.c_int, vp, C.c_int64, vp]),
    "ibl_vit_forward": (C.c_int, [vp, vp, vp, C.c_int, vp, vp, C.c_int64, vp]),
    "ibl_reg_ctx_create": (C.c_int, [C.POINTER(vp), C.c_int64]),
    "ibl_reg_ctx_destroy": (C.c_int, [vp]),
    "ibl_reg_ctx_reset": (C.c_int, [vp]),
    "ibl_reg_ctx_high_water": (C.c_int64, [vp]),
    "ibl_reg_ctx_status": (C.c_int, [vp, C.c_int]),
    "ibl_unproject_masks": (C.c_int, [vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, vp, C.c_int64,
                                      vp, vp, vp]),
    "ibl_voxel_downsample_batch": (C.c_int, [vp, vp, vp, vp, C.c_int32, C.c_double, vp, vp, vp, vp, vp]),
    "ibl_dbscan_batch": (C.c_int, [vp, vp, vp, C.c_int32, C.c_double, C.c_int32, vp, vp, vp]),
    "ibl_unproject_masks_f64": (C.c_int, [vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, vp, vp, vp,
                                          C.c_int64, vp, vp, vp]),
    "ibl_radius_outlier_batch": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_double, C.c_int, vp, vp]),
    "ibl_normals_fpfh_batch": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, vp, vp, vp]),
    "ibl_register_batch": (C.c_int, [vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_double, C.c_double,
                                     C.c_double, C.c_uint64, C.c_uint32, C.c_int64, C.c_int, vp, vp, vp, vp, vp, vp, vp]),
    "ibl_instance_features_batch": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_double, C.c_double, vp, vp, vp, vp, vp, vp, vp]),
    "ibl_register_batch_cached": (C.c_int, [vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_double, C.c_double,
                                            C.c_double, C.c_uint64, C.c_uint32, C.c_int64, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp,
                                            vp, vp]),
    "ibl_register_batch_ids": (C.c_int, [vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_double, C.c_double,
                                         C.c_double, C.c_uint64, C.c_int64, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "ibl_memgrid_build": (C.c_int, [vp, vp, C.c_int64, C.c_double, C.POINTER(vp), vp]),
    "ibl_memgrid_destroy": (C.c_int, [vp]),
    "ibl_evaluate_batch": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_double, vp, vp, vp]),
    "ibl_register_evaluate_batch": (C.c_int, [vp, vp, vp, vp, C.c_int, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp,
                                              C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_double, C.c_uint64, C.c_uint32,
                                              C.c_int64, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "ibl_evaluate_points": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_double, vp, vp, vp, vp]),
    "ibl_dator_head_workspace_bytes": (C.c_int64, [C.c_int]),
    "ibl_dator_head_forward": (C.c_int, [vp, vp, vp, C.c_int, vp, vp, C.c_int64, vp]),
    "ibl_preprocess_depth": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, vp, vp]),
    "ibl_assign_batch": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, C.c_int]),
    "ibl_assign_candidates": (C.c_int, [vp, vp, vp, C.c_int64, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp,
                                        C.c_int, C.c_int]),
    "ibl_resample_ksize": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "ibl_resample_table": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "ibl_comm_unique_id": (C.c_int, [vp, C.c_int]),
    "ibl_comm_init": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int]),
    "ibl_comm_destroy": (C.c_int, [vp]),
    "ibl_allgather_topk": (C.c_int, [vp, vp, vp, C.c_int64, vp]),
    "ibl_alltoall": (C.c_int, [vp, vp, vp, C.c_int64, vp]),
    "ibl_allreduce_min": (C.c_int, [vp, vp, C.c_int64, vp]),
    "ibl_allreduce_max_i32": (C.c_int, [vp, vp, C.c_int64, vp]),
    "ibl_topk_select": (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]),
    "ibl_match_topk_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int64, C.c_int64]),
    "ibl_match_topk": (C.c_int, [vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp,
                                 C.c_int64, vp]),
}


def _bind():
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)       # AttributeError if the symbol is missing -> loud failure
        fn.restype = res
        fn.argtypes = args


_bind()


def check(status, what=""):
    if status != 0:
        msg = lib.ibl_last_error()
        raise IblError(f"{what} failed ({status}): {msg.decode() if msg else ''}")


def declared_symbols():
    return list(_SIGS.keys())
