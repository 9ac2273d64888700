"""ctypes binding of liborbhip.so (the C ABI in include/orbhip.h).

The library is the product: if it is missing or cannot be loaded this module raises -- there is
no Python/CPU fallback for any compute step.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.normpath(os.path.join(_HERE, "..", "csrc"))
# liborbhip.so is the product: it reads four environment switches (equivalent, tested paths).  Every other ORBHIP_* knob
# (tuning, forced-overflow capacities, timing-ablation stops; csrc/orbhip_internal.h, ORB_TUNE) exists only in
# liborbhip_ablation.so, the same sources built with -DORBHIP_ABLATION.  This binding -- test and bench infrastructure, not
# what the C++ drop-in classes link -- loads the ablation build when one of those knobs is set in the environment (or with
# ORBHIP_ABLATION=1), so that tests/ and tools/ reach the fallback paths; otherwise the product library.
TUNE_KNOBS = (
    "BLUR_PLACE", "NO_SPLIT", "COPY_OUT", "BOW_THREADS", "BOW_GLOBAL_DESC", "BOW_PHASES", "CHAIN_DEPTH", "DESCRIBE_KPW",
    "DESCRIBE_MAP", "DESCRIBE_PHASES", "DESCRIBE_PADLDS", "DESCRIBE_AX4", "PROJ_SEQ", "PROJ_ROUNDS", "PROJ_K", "INIT_K",
    "QT_THREADS", "QT_PHASES", "QT_LDSPTS", "QT_THREADS_SMALL", "SETS_COPY", "STEREO_ENT_PER_KP", "FAST_TILE_CELLS", "FAST_LISTCAP",
    "FAST_PHASES", "FAST_PITCH", "FAST_DEFER", "FAST_LDS_PAD", "XCD_MAP", "XCD_CHUNK", "FAST_XCD", "FRAME_SPLIT", "INIT_STOP",
    "VOCAB_QUAD", "FUSE_BLUR", "RESIZE_FIT_P", "RESIZE_FIT", "RESIZE_LDS_PAD", "DESCRIBE_FUSED", "DESCRIBE_FUSED_SCHED", "DESCRIBE_FUSED_SPLIT", "BOW_LANE", "DEBUG_DESTROY",
)


def _wants_ablation():
    if os.environ.get("ORBHIP_ABLATION", "0") not in ("", "0"):
        return True
    return any(("ORBHIP_" + k) in os.environ for k in TUNE_KNOBS)


LIB_PATH = os.path.join(CSRC, "liborbhip_ablation.so" if _wants_ablation() else "liborbhip.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
CAND_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("score", "<i4")])
MAX_LEVELS = 16

# every symbol include/orbhip.h declares
SYMBOLS = [
    "orbhip_device_count", "orbhip_create", "orbhip_destroy", "orbhip_last_error", "orbhip_sync",
    "orbhip_stream", "orbhip_get_tables", "orbhip_tables", "orbhip_max_keypoints", "orbhip_level_size",
    "orbhip_extract", "orbhip_extract_batch", "orbhip_extract_batch_device", "orbhip_host_alloc", "orbhip_host_free",
    "orbhip_pipe_create", "orbhip_pipe_destroy", "orbhip_pipe_submit", "orbhip_pipe_wait", "orbhip_pipe_enable_bow", "orbhip_pipe_matches",
    "orbhip_get_pyramid_level", "orbhip_set_host_pyramid", "orbhip_host_pyramid_level", "orbhip_debug_get_blurred_level", "orbhip_debug_get_candidates",
    "orbhip_debug_get_level_keypoints", "orbhip_hamming_knn2", "orbhip_hamming_knn2_device",
    "orbhip_hamming_knn2_seq_device", "orbhip_get_stage_times", "orbhip_set_stage_timing", "orbhip_set_blur_placement", "orbhip_vocab_load", "orbhip_vocab_load_device",
    "orbhip_vocab_info", "orbhip_vocab_text_to_binary", "orbhip_vocab_transform", "orbhip_vocab_transform_device",
    "orbhip_search_by_bow_seq_device", "orbhip_stereo_match", "orbhip_stereo_match_device",
    "orbhip_hamming_knn2_lists", "orbhip_search_by_bow", "orbhip_comm_unique_id",
    "orbhip_comm_init", "orbhip_comm_destroy", "orbhip_comm_info", "orbhip_bcast_blob_device", "orbhip_knn2_allgather_merge_device",
    "orbhip_knn2_merge_device", "orbhip_grid_build_device", "orbhip_grid_build",
    "orbhip_features_in_area", "orbhip_search_by_projection", "orbhip_search_by_projection_device",
    "orbhip_search_for_initialization", "orbhip_search_for_initialization_device",
    "orbhip_search_for_triangulation", "orbhip_window_best", "orbhip_window_best_device",
    "orbhip_distinctive_descriptors", "orbhip_distinctive_descriptors_device",
    "orbhip_undistort_keypoints", "orbhip_undistort_keypoints_device", "orbhip_init_undistort_rectify_map",
    "orbhip_remap_set_maps", "orbhip_remap", "orbhip_remap_device",
    "orbhip_set_put", "orbhip_set_has", "orbhip_set_drop", "orbhip_search_by_bow_sets", "orbhip_window_best_set",
    "orbhip_set_info", "orbhip_set_fingerprint", "orbhip_set_fingerprint_rows", "orbhip_vocab_share", "orbhip_vocab_generation", "orbhip_set_limit", "orbhip_debug_roundtrip", "orbhip_debug_path_mask", "orbhip_frame_build", "orbhip_frame_fingerprint", "orbhip_set_put_from_frame",
]


class FrameParams(C.Structure):   # orbhip_frame_params
    _fields_ = [("K", C.c_float * 9), ("dist", C.c_float * 8), ("ndist", C.c_int), ("min_x", C.c_float), ("min_y", C.c_float),
                ("inv_w", C.c_float), ("inv_h", C.c_float), ("levelsup", C.c_int)]


QUERY_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("radius", "<f4"), ("proj_xr", "<f4"), ("min_level", "<i4"),
                        ("max_level", "<i4"), ("angle", "<f4"), ("flags", "<i4")])   # orbhip_proj_query
Q_ACTIVE, Q_OBSERVED = 1, 2
GRID_COLS, GRID_ROWS = 64, 48


class OrbHipError(RuntimeError):
    pass


_lib = None


def load():
    """Load liborbhip.so and declare the prototypes.  Raises OrbHipError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OrbHipError("liborbhip.so is not built (%s); run __graft_entry__.build() or "
                          "`make -C vi-orb-slam-icra2018_amd/csrc`" % LIB_PATH)
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:
        raise OrbHipError("cannot load %s: %s" % (LIB_PATH, e))
    vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
    ip = C.POINTER(C.c_int)
    L.orbhip_device_count.restype = i32
    L.orbhip_create.argtypes = [i32, i32, f32, i32, i32, i32, i32, i32, i32]
    L.orbhip_create.restype = vp
    L.orbhip_destroy.argtypes = [vp]
    L.orbhip_destroy.restype = None
    L.orbhip_last_error.argtypes = [vp]
    L.orbhip_last_error.restype = C.c_char_p
    L.orbhip_sync.argtypes = [vp]
    L.orbhip_stream.argtypes = [vp]
    L.orbhip_stream.restype = vp
    L.orbhip_get_tables.argtypes = [vp, ip, C.POINTER(C.c_double), vp, vp, vp, vp, vp, vp]
    L.orbhip_tables.argtypes = [i32, f32, i32, i32, i32, vp, vp, vp, vp, vp, vp]
    L.orbhip_max_keypoints.argtypes = [vp]
    L.orbhip_level_size.argtypes = [vp, i32, i32, i32, ip, ip]
    L.orbhip_extract.argtypes = [vp, vp, i32, i32, i32, vp, vp, i32, ip, vp]
    L.orbhip_extract_batch.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, i32, vp]
    L.orbhip_extract_batch_device.argtypes = [vp, vp, i32, i32, i32, i32, C.c_size_t, vp, vp, i32, vp]
    L.orbhip_host_alloc.argtypes = [C.c_size_t]
    L.orbhip_host_alloc.restype = vp
    L.orbhip_host_free.argtypes = [vp]
    L.orbhip_host_free.restype = None
    L.orbhip_pipe_create.argtypes = [vp, i32, i32, i32, i32]
    L.orbhip_pipe_destroy.argtypes = [vp]
    L.orbhip_pipe_submit.argtypes = [vp, vp, i32, i32, C.c_size_t]
    L.orbhip_pipe_wait.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), ip, ip]
    L.orbhip_pipe_enable_bow.argtypes = [vp, i32, f32, i32]
    L.orbhip_pipe_matches.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.orbhip_get_pyramid_level.argtypes = [vp, i32, i32, vp, i32, ip, ip]
    L.orbhip_set_host_pyramid.argtypes = [vp, i32]
    L.orbhip_host_pyramid_level.argtypes = [vp, i32, i32, C.POINTER(vp), ip, ip, ip]
    L.orbhip_debug_get_blurred_level.argtypes = [vp, i32, i32, vp, i32, ip, ip]
    L.orbhip_debug_get_candidates.argtypes = [vp, i32, i32, vp, i32, ip]
    L.orbhip_debug_get_level_keypoints.argtypes = [vp, i32, i32, vp, i32, ip]
    L.orbhip_hamming_knn2.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp]
    L.orbhip_hamming_knn2_device.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp]
    L.orbhip_hamming_knn2_seq_device.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp]
    L.orbhip_get_stage_times.argtypes = [vp, vp]
    L.orbhip_set_stage_timing.argtypes = [vp, i32]
    L.orbhip_set_blur_placement.argtypes = [vp, i32]
    L.orbhip_stereo_match.argtypes = [vp, vp, vp, vp, i32, vp, vp, i32, f32, f32, vp, vp, ip]
    L.orbhip_stereo_match_device.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, vp, vp, vp]
    L.orbhip_vocab_load.argtypes = [vp, vp, C.c_size_t]
    L.orbhip_vocab_load_device.argtypes = [vp, vp, C.c_size_t]
    L.orbhip_vocab_info.argtypes = [vp, ip, ip, ip, ip, ip, ip]
    L.orbhip_vocab_share.argtypes = [vp, vp]
    L.orbhip_vocab_generation.argtypes = [vp]
    L.orbhip_vocab_generation.restype = C.c_ulonglong
    L.orbhip_vocab_text_to_binary.argtypes = [C.c_char_p, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t), vp, C.c_size_t]
    L.orbhip_vocab_transform.argtypes = [vp, vp, i32, i32, vp, vp, vp]
    L.orbhip_vocab_transform_device.argtypes = [vp, vp, i32, i32, vp, vp, vp]
    L.orbhip_search_by_bow_seq_device.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp, vp, vp]
    L.orbhip_hamming_knn2_lists.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp, vp, vp]
    L.orbhip_search_by_bow.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, i32,
                                       vp, i32, vp, vp, vp, vp, vp, i32,
                                       i32, i32, f32, i32, vp, vp, ip]
    L.orbhip_grid_build_device.argtypes = [vp, vp, vp, i32, i32, f32, f32, f32, f32, vp, vp]
    L.orbhip_grid_build.argtypes = [vp, vp, i32, f32, f32, f32, f32, vp, vp]
    L.orbhip_features_in_area.argtypes = [vp, vp, i32, f32, f32, f32, f32, vp, i32, vp, vp, i32]
    L.orbhip_search_by_projection.argtypes = [vp, vp, vp, i32, vp, vp, f32, f32, f32, f32, vp, vp, i32, i32, f32, i32, i32,
                                              vp, ip]
    L.orbhip_search_by_projection_device.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, f32, f32, f32, f32, vp, vp, vp, vp, vp,
                                                     i32, i32, f32, i32, i32, vp, vp]
    L.orbhip_distinctive_descriptors.argtypes = [vp, vp, vp, i32, vp, vp]
    L.orbhip_distinctive_descriptors_device.argtypes = [vp, vp, vp, i32, vp, vp]
    L.orbhip_window_best.argtypes = [vp, vp, vp, i32, vp, vp, i32, f32, f32, f32, f32, vp, vp, i32, vp, vp]
    L.orbhip_window_best_device.argtypes = [vp, vp, vp, i32, i32, vp, vp, i32, f32, f32, f32, f32, vp, vp, vp, vp, vp, i32,
                                            vp, vp]
    L.orbhip_search_for_initialization.argtypes = [vp, vp, vp, i32, vp, vp, i32, f32, f32, f32, f32, vp, i32, f32, i32, vp, ip]
    L.orbhip_search_for_initialization_device.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, f32, f32, f32, f32, vp, vp,
                                                          vp, i32, f32, i32, vp, vp]
    L.orbhip_search_for_triangulation.argtypes = [vp] + [vp, vp, i32, vp, vp, vp, vp, vp, i32] * 2 + \
        [vp, f32, f32, vp, vp, i32, i32, i32, vp, ip]
    L.orbhip_undistort_keypoints.argtypes = [vp, vp, i32, vp, vp, i32, vp, vp]
    L.orbhip_undistort_keypoints_device.argtypes = [vp, vp, vp, i32, i32, vp, vp, i32, vp, vp]
    L.orbhip_init_undistort_rectify_map.argtypes = [vp, vp, i32, vp, vp, i32, i32, vp, vp]
    L.orbhip_remap_set_maps.argtypes = [vp, vp, vp, i32, i32]
    L.orbhip_remap.argtypes = [vp, vp, i32, i32, i32, vp, i32]
    L.orbhip_remap_device.argtypes = [vp, vp, i32, i32, i32, i32, C.c_size_t, vp, i32, C.c_size_t]
    L.orbhip_set_put.argtypes = [vp, C.c_uint64, vp, vp, i32, vp, vp, vp, i32, f32, f32, f32, f32]
    L.orbhip_set_has.argtypes = [vp, C.c_uint64, i32]
    L.orbhip_set_drop.argtypes = [vp, C.c_uint64]
    L.orbhip_set_limit.argtypes = [vp, i32]
    L.orbhip_set_limit.restype = i32
    L.orbhip_debug_path_mask.argtypes = [i32]
    L.orbhip_debug_path_mask.restype = C.c_uint32
    L.orbhip_search_by_bow_sets.argtypes = [vp, C.c_uint64, vp, C.c_uint64, vp, i32, i32, f32, i32, vp, vp, ip]
    L.orbhip_window_best_set.argtypes = [vp, C.c_uint64, vp, vp, i32, vp, vp, i32, vp, vp]
    L.orbhip_set_info.argtypes = [vp, C.c_uint64, ip, ip, C.POINTER(C.c_uint64)]
    L.orbhip_set_fingerprint.argtypes = [vp, vp, i32]
    L.orbhip_set_fingerprint.restype = C.c_uint64
    L.orbhip_frame_build.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, i32, ip, vp, vp, vp, vp, vp]
    L.orbhip_debug_roundtrip.argtypes = [vp, i32, i32, C.POINTER(C.c_double)]
    L.orbhip_frame_fingerprint.argtypes = [vp]
    L.orbhip_frame_fingerprint.restype = C.c_uint64
    L.orbhip_set_put_from_frame.argtypes = [vp, C.c_uint64, vp, vp, vp, vp, i32]
    L.orbhip_comm_unique_id.argtypes = [vp]
    L.orbhip_comm_init.argtypes = [vp, i32, i32, vp]
    L.orbhip_bcast_blob_device.argtypes = [vp, vp, C.c_size_t, i32]
    L.orbhip_comm_destroy.argtypes = [vp]
    L.orbhip_comm_info.argtypes = [vp, ip, ip]
    L.orbhip_knn2_allgather_merge_device.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, vp]
    L.orbhip_knn2_merge_device.argtypes = [vp, vp, i32, i32, vp, vp, vp]
    _lib = L
    return L


def exported_symbols():
    """Names from SYMBOLS that the built library actually exports (no GPU needed)."""
    L = C.CDLL(LIB_PATH)
    return [s for s in SYMBOLS if hasattr(L, s)]


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def last_error(ctx=None):
    s = load().orbhip_last_error(ctx)
    return s.decode() if s else ""


def check(rc, ctx=None, what=""):
    if rc != 0:
        raise OrbHipError("%s failed (%d): %s" % (what, rc, last_error(ctx)))
