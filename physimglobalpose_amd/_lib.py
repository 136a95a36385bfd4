"""ctypes binding of libpgp.so (include/pgp.h).  Fails loudly when the HIP library is missing:
there is no Python, torch or CPU fallback for any entry point."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# PGP_LIB: a diagnostic / A-B build of the same library (tools/ab/*.so), never a different implementation
LIB_PATH = os.environ.get("PGP_LIB") or os.path.join(HERE, "libpgp.so")

PGP_MODE_PLAIN = 0
PGP_MODE_WEIGHTED = 1

_f = C.POINTER(C.c_float)
_i = C.POINTER(C.c_int)


class IndexInfo(C.Structure):
    _fields_ = [("n_scene", C.c_int), ("n_model", C.c_int),
                ("grid_nx", C.c_int), ("grid_ny", C.c_int), ("grid_nz", C.c_int),
                ("cell_size", C.c_float), ("delta", C.c_float),
                ("n_cells", C.c_longlong), ("n_candidates", C.c_longlong),
                ("n_occupied", C.c_longlong), ("bytes_index", C.c_longlong), ("build_ms", C.c_float),
                ("sparse", C.c_int), ("n_blocks", C.c_longlong)]


class IcpParams(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("trim_fraction", C.c_float),
                ("max_corr_dist", C.c_float), ("energy_ratio", C.c_float)]


class IcpOptions(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("trim_fraction", C.c_float), ("max_corr_dist", C.c_float),
                ("energy_ratio", C.c_float), ("error_metric", C.c_int), ("transformation_epsilon", C.c_float),
                ("relative_mse", C.c_float), ("absolute_mse", C.c_float), ("min_diff_rot", C.c_float),
                ("min_diff_trans", C.c_float), ("smooth_length", C.c_int), ("nn_search", C.c_int)]


class IcpJob(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("d_src4", C.c_void_p), ("n_src", C.c_int), ("d_tgt4", C.c_void_p), ("n_tgt", C.c_int),
                ("d_T", C.c_void_p), ("n", C.c_int), ("d_energy", C.c_void_p), ("d_iters", C.c_void_p)]


class MultiIcpJob(C.Structure):
    _fields_ = [("src_xyz", C.POINTER(C.c_float)), ("n_src", C.c_int), ("tgt_xyz", C.POINTER(C.c_float)), ("n_tgt", C.c_int),
                ("T", C.POINTER(C.c_float)), ("n", C.c_int), ("energy", C.POINTER(C.c_float)), ("iters", C.POINTER(C.c_int))]


class MultiInfo(C.Structure):
    _fields_ = [("n_local", C.c_int), ("world", C.c_int), ("rank0", C.c_int), ("rccl_ranks", C.c_int), ("emulated", C.c_int),
                ("devices", C.c_int * 16), ("exchanges", C.c_longlong)]


class Camera(C.Structure):
    _fields_ = [("rows", C.c_int), ("cols", C.c_int), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("z_near", C.c_float), ("z_max", C.c_float)]


class ClusterParams(C.Structure):
    _fields_ = [("accept_fraction", C.c_float), ("rot_thresh_deg", C.c_float), ("trans_thresh", C.c_float)]


# every symbol include/pgp.h declares: (restype, argtypes)
SIGNATURES = {
    "pgp_version": (C.c_int, []),
    "pgp_last_error": (C.c_char_p, []),
    "pgp_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "pgp_destroy": (C.c_int, [C.c_void_p]),
    "pgp_center": (C.c_int, [_f, C.c_int, _f, C.c_int, _f, C.c_int, _f, _f]),
    "pgp_weights_from_image": (C.c_int, [_f, C.c_int, _f, _f, C.POINTER(C.c_ushort), C.c_int, C.c_int, _f]),
    "pgp_image_rows_needed": (C.c_int, [_f, C.c_int, _f, _f, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pgp_set_scene": (C.c_int, [C.c_void_p, _f, _f, _f, C.c_int, C.c_float]),
    "pgp_set_model": (C.c_int, [C.c_void_p, _f, _f, C.c_int]),
    "pgp_score_lcp": (C.c_int, [C.c_void_p, _f, C.c_int, C.c_int, C.c_float, _f, _i, _i, _f]),
    "pgp_reserve": (C.c_int, [C.c_void_p, C.c_int]),
    "pgp_score_lcp_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pgp_settle_best_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float,
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
    "pgp_registered": (C.c_int, [C.c_void_p, _f, C.c_int, C.c_float, _i, _i]),
    "pgp_registered_model": (C.c_int, [C.c_void_p, _f, _f, _f, C.c_int, C.c_float, _i, _i]),
    "pgp_find_congruent_4pcs": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, _i, C.c_int, _i, C.c_int,
                                          _i, C.c_int, _i]),
    "pgp_running_best": (C.c_int, [_f, C.c_int, _i, _i]),
    "pgp_set_search_model": (C.c_int, [C.c_void_p, _f, C.c_int]),
    "pgp_set_ppf_map": (C.c_int, [C.c_void_p, _i, _i, _i, C.c_int]),
    "pgp_select_bases": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int, _i, _f, _i]),
    "pgp_select_bases_rows": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int, _i, _f, _i, _i]),
    "pgp_select_bases_rows_begin": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int]),
    "pgp_select_bases_rows_end": (C.c_int, [C.c_void_p, _i, _f, _i, _i]),
    "pgp_ppf_features": (C.c_int, [C.c_void_p, _i, C.c_int, _i, _i]),
    "pgp_stocs_stage_weights": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _f, _f, _i]),
    "pgp_base_invariants": (C.c_int, [C.c_void_p, _i, C.c_int, _f, _i]),
    "pgp_rigid_from_congruent": (C.c_int, [C.c_void_p, _i, _i, C.c_int, _f, _f, _f,
                                           C.POINTER(C.c_double), _i, _f]),
    "pgp_rigid_from_congruent_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, _f, _f,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p]),
    "pgp_extract_pairs": (C.c_int, [C.c_void_p, C.c_float, C.c_float, _i, C.c_int, _i]),
    "pgp_find_congruent": (C.c_int, [C.c_void_p, _f, C.c_float, C.c_float, C.c_float, _i, C.c_int,
                                     _i, C.c_int, _i, C.c_int, _i]),
    "pgp_find_congruent_batch": (C.c_int, [C.c_void_p, _i, _f, _f, C.c_int, C.c_float, _i]),
    "pgp_find_congruent_batch_rows": (C.c_int, [C.c_void_p, _i, _f, _f, _i, C.c_int, C.c_float, _i]),
    "pgp_congruent_batch_quads": (C.c_int, [C.c_void_p, _i, C.c_int, _i]),
    "pgp_congruent_batch_fit": (C.c_int, [C.c_void_p, _i, _i, C.c_int, _f, _f, _f, C.POINTER(C.c_double), _i, _f]),
    "pgp_congruent_batch_fit_score": (C.c_int, [C.c_void_p, _i, _i, C.c_int, _f, _f, C.c_int, C.c_float, _f, _i, _i, _f]),
    "pgp_congruent_batch_fetch": (C.c_int, [C.c_void_p, _i, C.c_int, _f, C.POINTER(C.c_double)]),
    "pgp_congruent_batch_fit_score_list": (C.c_int, [C.c_void_p, _i, _i, C.c_int, _f, _f, C.c_int, C.c_float, C.c_int, _i, _i, _f, _f,
                                                    C.POINTER(C.c_double), _i, _i, _f, _f, C.POINTER(C.c_double), _i, _i]),
    "pgp_congruent_batch_sample_fit_score_list": (C.c_int, [C.c_void_p, C.c_ulonglong, C.c_int, _i, _f, _f, C.c_int, C.c_float, C.c_int,
                                                           _i, _i, _f, _f, C.POINTER(C.c_double), _i, _i, _f, _f,
                                                           C.POINTER(C.c_double), _i, _i, _i, _i]),
    "pgp_sample_quads": (C.c_int, [C.c_ulonglong, _i, C.c_int, C.c_int, _i, _i]),
    "pgp_icp_refine": (C.c_int, [C.c_void_p, _f, C.c_int, _f, C.c_int, _f, C.c_int,
                                 C.POINTER(IcpParams), _f, _i]),
    "pgp_icp_refine_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                        C.c_int, C.POINTER(IcpParams), C.c_void_p, C.c_void_p, C.c_void_p]),
    "pgp_icp_default_options": (C.c_int, [C.POINTER(IcpOptions)]),
    "pgp_icp_target_token": (C.c_int, [C.c_void_p, C.c_ulonglong]),
    "pgp_icp_refine_multi_device": (C.c_int, [C.POINTER(IcpJob), C.c_int, C.POINTER(IcpParams), C.c_void_p]),
    "pgp_select_top_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p]),
    "pgp_set_scene_weights": (C.c_int, [C.c_void_p, _f, C.c_int]),
    "pgp_unexplained_segment": (C.c_int, [C.c_void_p, _f, C.c_int, _f, _i, _f, C.c_int, C.c_float, C.c_char_p, _i]),
    "pgp_multi_set_scene_weights": (C.c_int, [C.c_void_p, _f, C.c_int]),
    "pgp_icp_refine_ex": (C.c_int, [C.c_void_p, _f, C.c_int, _f, _f, C.c_int, _f, C.c_int,
                                    C.POINTER(IcpOptions), _f, _i]),
    "pgp_icp_refine_ex_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                           C.c_void_p, C.c_int, C.POINTER(IcpOptions), C.c_void_p, C.c_void_p,
                                           C.c_void_p]),
    "pgp_radius_outlier_filter": (C.c_int, [C.c_void_p, _f, _f, C.c_int, C.c_float, C.c_int,
                                            C.POINTER(C.c_ubyte), _f, _i]),
    "pgp_backproject_depth": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_ubyte), C.c_int, C.c_int, _f,
                                        C.c_double, C.c_double, _f, C.c_int, _i]),
    "pgp_set_scene_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p]),
    "pgp_voxel_grid": (C.c_int, [C.c_void_p, _f, C.c_int, C.c_float, _f, C.c_int, _i]),
    "pgp_mls_normals": (C.c_int, [C.c_void_p, _f, C.c_int, C.c_float, _f, _f, _f, _i, C.c_int, _i]),
    "pgp_mls_normals_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_int, _i, C.c_void_p]),
    "pgp_set_exact_records": (C.c_int, [C.c_void_p, C.c_int]),
    "pgp_set_exact_ties": (C.c_int, [C.c_void_p, C.c_int]),
    "pgp_set_verify_early_out": (C.c_int, [C.c_void_p, C.c_int]),
    "pgp_verify_early_out_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pgp_settle_records_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "pgp_voxel_grid_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int, _i, C.c_void_p]),
    "pgp_pose_hausdorff": (C.c_int, [C.c_void_p, _f, C.c_int, _f, C.c_int, _i, C.c_int, _f, _f]),
    "pgp_backproject_depth_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, _f,
                                               C.c_double, C.c_double, C.c_void_p, C.c_int, _i, C.c_void_p]),
    "pgp_cluster_poses_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, _f,
                                           C.POINTER(ClusterParams), C.c_void_p, C.c_void_p, _i, C.c_void_p]),
    "pgp_depth_cost": (C.c_int, [C.c_void_p, _f, _f, C.c_int, C.c_int, C.c_int, C.c_float, _f, _i]),
    "pgp_depth_cost_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                        C.c_void_p, C.c_void_p, C.c_void_p]),
    "pgp_render_depth_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                          C.c_int, C.POINTER(Camera), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "pgp_render_depth": (C.c_int, [C.c_void_p, _f, C.c_int, _i, C.c_int, _f, C.c_int, C.POINTER(Camera), _f, _f]),
    "pgp_cluster_poses": (C.c_int, [C.c_void_p, _f, _f, C.c_int, C.c_float, _f, C.POINTER(ClusterParams), _i,
                                    C.c_int, _i, _i]),
    "pgp_pose_error": (C.c_int, [C.c_void_p, _f, _f, C.c_int, _f, _f, _f]),
    "pgp_multi_create": (C.c_int, [C.POINTER(C.c_void_p), _i, C.c_int]),
    "pgp_multi_destroy": (C.c_int, [C.c_void_p]),
    "pgp_multi_size": (C.c_int, [C.c_void_p]),
    "pgp_multi_context": (C.c_void_p, [C.c_void_p, C.c_int]),
    "pgp_multi_slice": (C.c_int, [C.c_int, C.c_int, C.c_int, _i, _i]),
    "pgp_multi_set_scene": (C.c_int, [C.c_void_p, _f, _f, _f, C.c_int, C.c_float]),
    "pgp_multi_set_model": (C.c_int, [C.c_void_p, _f, _f, C.c_int]),
    "pgp_multi_score_lcp": (C.c_int, [C.c_void_p, _f, C.c_int, C.c_int, C.c_float, _f, _i, _i, _f]),
    "pgp_multi_upload": (C.c_int, [C.c_void_p, _f, C.c_int]),
    "pgp_multi_score_uploaded": (C.c_int, [C.c_void_p, C.c_int, C.c_float, _f, _i, _i, _f]),
    "pgp_multi_last_timing": (C.c_int, [C.c_void_p, _f, _f, _f]),
    "pgp_multi_unique_id": (C.c_int, [C.c_void_p]),
    "pgp_multi_create_ranked": (C.c_int, [C.POINTER(C.c_void_p), _i, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pgp_multi_get_info": (C.c_int, [C.c_void_p, C.POINTER(MultiInfo)]),
    "pgp_multi_upload_slot": (C.c_int, [C.c_void_p, C.c_int, _f, C.c_int]),
    "pgp_multi_enqueue_slot": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float]),
    "pgp_multi_collect": (C.c_int, [C.c_void_p, _f, _i, _i, _f]),
    "pgp_multi_add_object": (C.c_int, [C.c_void_p]),
    "pgp_multi_objects": (C.c_int, [C.c_void_p]),
    "pgp_multi_object_context": (C.c_void_p, [C.c_void_p, C.c_int, C.c_int]),
    "pgp_multi_set_object_scene": (C.c_int, [C.c_void_p, C.c_int, _f, _f, _f, C.c_int, C.c_float]),
    "pgp_multi_set_object_scene_weights": (C.c_int, [C.c_void_p, C.c_int, _f, C.c_int]),
    "pgp_multi_set_object_model": (C.c_int, [C.c_void_p, C.c_int, _f, _f, C.c_int]),
    "pgp_multi_set_object_search_model": (C.c_int, [C.c_void_p, C.c_int, _f, C.c_int]),
    "pgp_multi_set_object_ppf_map": (C.c_int, [C.c_void_p, C.c_int, _i, _i, _i, C.c_int]),
    "pgp_multi_flat_slices": (C.c_int, [_i, C.c_int, C.c_int, C.c_int, _i, _i, _i, _i]),
    "pgp_multi_score_objects": (C.c_int, [C.c_void_p, C.POINTER(_f), _i, C.c_int, C.c_int, C.c_float, _f, _i, _i, _f]),
    "pgp_multi_upload_objects": (C.c_int, [C.c_void_p, C.POINTER(_f), _i, C.c_int]),
    "pgp_multi_score_objects_uploaded": (C.c_int, [C.c_void_p, C.c_int, C.c_float, _f, _i, _i, _f]),
    "pgp_multi_icp_refine": (C.c_int, [C.c_void_p, C.POINTER(MultiIcpJob), C.c_int, C.POINTER(IcpParams)]),
    "pgp_multi_find_congruent_batch": (C.c_int, [C.c_void_p, C.c_int, _i, _f, _f, C.c_int, C.c_float, _i]),
    "pgp_multi_congruent_batch_quads": (C.c_int, [C.c_void_p, C.c_int, _i, C.c_int, _i]),
    "pgp_multi_congruent_batch_fit": (C.c_int, [C.c_void_p, C.c_int, _i, _i, C.c_int, _f, _f, _f, C.POINTER(C.c_double), _i, _f]),
    "pgp_set_kernel_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "pgp_get_kernel_timing": (C.c_int, [C.c_void_p, _i, _f, C.c_int]),
    "pgp_get_index_info": (C.c_int, [C.c_void_p, C.POINTER(IndexInfo)]),
}

_lib = None


def load():
    """Load libpgp.so and bind every declared symbol; raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `make -C physimglobalpose_amd/csrc` "
            "(or __graft_entry__.build()); there is no fallback path")
    # PyTorch-ROCm wheels bundle their own libamdhip64; if libpgp.so pulls in the system HIP
    # runtime first, a later `import torch` finds two runtimes in the process and reports
    # "No HIP GPUs are available".  Loading torch first makes both share one runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class PgpError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        msg = load().pgp_last_error()
        raise PgpError(f"libpgp error {rc}: {msg.decode() if msg else '?'}")
