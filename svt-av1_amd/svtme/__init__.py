"""Python binding (ctypes) of the MI355X open-loop motion-estimation engine's C ABI (include/svtme.h).

The product is the shared library `libsvtme.so` (HIP kernels + host engine, built by svt-av1_amd/build.sh);
this module only mirrors the C structs and wraps the entry points for the tests and bench.py. There is no CPU
fallback: loading fails loudly when the library is missing, and every call fails when no GPU is present.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVTME_LIB") or os.path.join(_HERE, "libsvtme.so")  # SVTME_LIB: a diagnostic build of the same library

MAX_LISTS, MAX_REFS, PU_COUNT = 2, 4, 85
OK, ERR_BAD_ARG, ERR_NO_RESOURCES, ERR_NOT_FOUND, ERR_DEVICE, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5


class SvtmeError(RuntimeError):
    def __init__(self, code, what):
        super().__init__("%s failed with svtme status %d" % (what, code))
        self.code = code


class Sa(C.Structure):
    _fields_ = [("width", C.c_uint16), ("height", C.c_uint16)]


class SaMinMax(C.Structure):
    _fields_ = [("sa_min", Sa), ("sa_max", Sa)]


class Controls(C.Structure):
    _fields_ = [
        ("me_sa", SaMinMax), ("hme_l0_sa", SaMinMax), ("hme_l1_sa", Sa), ("hme_l2_sa", Sa),
        ("num_hme_sa_w", C.c_uint16), ("num_hme_sa_h", C.c_uint16),
        ("enable_hme_flag", C.c_uint8), ("enable_hme_level0_flag", C.c_uint8), ("enable_hme_level1_flag", C.c_uint8),
        ("enable_hme_level2_flag", C.c_uint8), ("hme_search_method", C.c_uint8), ("me_search_method", C.c_uint8),
        ("prehme_enable", C.c_uint8), ("prehme_skip_search_line", C.c_uint8), ("prehme_l1_early_exit", C.c_uint8),
        ("reserved0", C.c_uint8), ("prehme_sa_cfg", SaMinMax * 2),
        ("enable_me_hme_ref_pruning", C.c_uint8), ("protect_closest_refs", C.c_uint8),
        ("prune_ref_if_hme_sad_dev_bigger_than_th", C.c_uint16), ("prune_ref_if_me_sad_dev_bigger_than_th", C.c_uint16),
        ("enable_me_sr_adjustment", C.c_uint8), ("distance_based_hme_resizing", C.c_uint8),
        ("reduce_me_sr_based_on_mv_length_th", C.c_uint16), ("stationary_hme_sad_abs_th", C.c_uint16),
        ("stationary_me_sr_divisor", C.c_uint16), ("reduce_me_sr_based_on_hme_sad_abs_th", C.c_uint16),
        ("me_sr_divisor_for_low_hme_sad", C.c_uint16),
        ("reduce_hme_l0_sr_th_min", C.c_uint8), ("reduce_hme_l0_sr_th_max", C.c_uint8), ("stat_factor", C.c_uint8),
        ("use_best_unipred_cand_only", C.c_uint8), ("prune_me_candidates_th", C.c_int32), ("me_early_exit_th", C.c_uint32),
    ]


assert C.sizeof(Controls) == 84


class PresetKey(C.Structure):
    _fields_ = [("enc_mode", C.c_int8), ("sc_class1", C.c_uint8), ("input_resolution", C.c_uint8),
                ("reserved", C.c_uint8), ("frame_rate", C.c_uint32)]


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("max_width", C.c_uint32), ("max_height", C.c_uint32),
                ("store_slots", C.c_uint32), ("max_in_flight", C.c_uint32), ("search_streams", C.c_uint32), ("kernel_timing", C.c_uint32),
                ("serial_hme", C.c_uint32)]


class Tail(C.Structure):
    _fields_ = [("enable", C.c_uint8), ("enable_me_8x8", C.c_uint8), ("enable_me_16x16", C.c_uint8), ("max_refs", C.c_uint8),
                ("max_cand", C.c_uint8), ("max_l0", C.c_uint8), ("input_resolution", C.c_uint8), ("gm_enabled", C.c_uint8),
                ("gm_use_distance_based_active_th", C.c_uint8), ("reserved", C.c_uint8 * 7)]


TAIL_MV_STRIDE, TAIL_CAND_STRIDE = 7, 23


class Picture(C.Structure):
    _fields_ = [
        ("cur_key", C.c_uint64), ("picture_number", C.c_uint64), ("width", C.c_uint32), ("height", C.c_uint32),
        ("temporal_layer_index", C.c_uint8), ("is_used_as_reference_flag", C.c_uint8),
        ("num_of_list_to_search", C.c_uint8), ("num_of_ref_pic_to_search", C.c_uint8 * 2), ("me_type", C.c_uint8),
        ("bypass_zz_check", C.c_uint8), ("restricted_motion_vector", C.c_uint8),
        ("ref_key", (C.c_uint64 * MAX_REFS) * MAX_LISTS), ("ref_picture_number", (C.c_uint64 * MAX_REFS) * MAX_LISTS),
        ("ctrl", Controls), ("tail", Tail), ("tf_me_exit_th", C.c_uint32), ("b64_step", C.c_uint8), ("reserved", C.c_uint8 * 3),
    ]


ME_OPEN_LOOP, ME_MCTF, ME_FIRST_PASS = 3, 1, 4  # EbMeType; 0 also reads as open loop


class SearchResult(C.Structure):
    """Field for field the reference's SearchResults (EbMotionEstimationContext.h:330-337), 24 bytes; tf_me_exit sits in
    that struct's tail padding."""
    _fields_ = [("list_i", C.c_uint8), ("ref_i", C.c_uint8), ("hme_sc_x", C.c_int16), ("hme_sc_y", C.c_int16),
                ("hme_sad", C.c_uint64), ("do_ref", C.c_uint8), ("tf_me_exit", C.c_uint8), ("reserved", C.c_uint8 * 6)]


SEARCH_RESULT_DTYPE = np.dtype({"names": ["list_i", "ref_i", "hme_sc_x", "hme_sc_y", "hme_sad", "do_ref", "tf_me_exit"],
                                "formats": ["u1", "u1", "<i2", "<i2", "<u8", "u1", "u1"],
                                "offsets": [0, 1, 2, 4, 8, 16, 17], "itemsize": 24})
SEARCH_RESULT_FIELDS = np.dtype([(f, SEARCH_RESULT_DTYPE.fields[f][0]) for f in SEARCH_RESULT_DTYPE.names])
assert SEARCH_RESULT_DTYPE.itemsize == C.sizeof(SearchResult) == 24
assert [getattr(SearchResult, f).offset for f in SEARCH_RESULT_DTYPE.names] == [0, 1, 2, 4, 8, 16, 17]


class Results(C.Structure):
    _fields_ = [("n_b64", C.c_uint32), ("p_sb_best_sad", C.POINTER(C.c_uint32)), ("p_sb_best_mv", C.POINTER(C.c_uint32)),
                ("search_results", C.POINTER(SearchResult)), ("zz_sad", C.POINTER(C.c_uint32)),
                ("reduce_me_sr_divisor", C.POINTER(C.c_uint32)),
                ("total_me_candidate_index", C.POINTER(C.c_uint8)), ("me_mv_array", C.POINTER(C.c_uint32)),
                ("me_candidate_array", C.POINTER(C.c_uint8)), ("me_distortion", C.POINTER(C.c_uint32)),
                ("b64_stats", C.POINTER(C.c_uint32)), ("final_do_ref", C.POINTER(C.c_uint8)),
                ("final_hme_sad", C.POINTER(C.c_uint64)), ("n_ref", C.c_uint32 * MAX_LISTS)]


EXPORTS = ("svtme_version", "svtme_derive_controls", "svtme_derive_tf_controls", "svtme_derive_first_pass_controls",
           "svtme_input_resolution", "svtme_create", "svtme_destroy",
           "svtme_store_put", "svtme_store_release", "svtme_store_read", "svtme_picture_submit", "svtme_picture_wait",
           "svtme_picture_done", "svtme_picture_timing", "svtme_kernel_timing", "svtme_picture_ops", "svtme_unpack_b64",
           "svtme_picture_search_static", "svtme_store_put_pinned", "svtme_store_wait", "svtme_host_register",
           "svtme_host_unregister", "svtme_host_alloc", "svtme_host_free", "svtme_pictures_submit",
           "svtme_group_create", "svtme_group_destroy", "svtme_group_size", "svtme_group_store_put", "svtme_group_store_release",
           "svtme_group_picture_submit", "svtme_group_picture_wait", "svtme_group_picture_done", "svtme_group_stats",
           "svtme_group_engine_for", "svtme_group_gm_set_tables",
           "svtme_gm_set_tables", "svtme_gm_corners", "svtme_gm_correspondences", "svtme_gm_warp_error", "svtme_gm_frame_error",
           "svtme_gm_refine",
           "svtme_tpl_search", "svtme_tpl_recon", "svtme_tpl_recon_keep", "svtme_set_tile_rects", "svtme_group_set_tile_rects",
           "svtme_store_put_many", "svtme_store_copy_peer", "svtme_group_pictures_submit", "svtme_group_ticket_engine",
           "svtme_group_set_peer_copies", "svtme_group_peer_stats")
MAX_BATCH = 32
MAX_PUT_BATCH = 16
GM_MAX_CORNERS = 4096


TPL_BLOCKS = 21
TPL_BLOCK = np.dtype([("best_intra_cost", "<i8"), ("best_inter_cost", "<i8"), ("mv_x", "<i2"), ("mv_y", "<i2"),
                      ("best_rf_idx", "i1"), ("best_is_inter", "u1"), ("best_intra_mode", "u1"), ("valid", "u1"),
                      ("src_recon_error", "<i8"), ("src_rate", "<i8")])


TPL_STATS = np.dtype([("srcrf_dist", "<i8"), ("recrf_dist", "<i8"), ("srcrf_rate", "<i8"), ("recrf_rate", "<i8")])  # svtme_tpl_stats


class TplPicture(C.Structure):
    """svtme_tpl_picture"""
    _fields_ = [("cur_key", C.c_uint64), ("ref_key", (C.c_uint64 * MAX_REFS) * MAX_LISTS), ("ref_skip", (C.c_uint8 * MAX_REFS) * MAX_LISTS),
                ("width", C.c_uint32), ("height", C.c_uint32),
                ("i_slice", C.c_uint8), ("dispenser_search_level", C.c_uint8), ("disable_intra_pred", C.c_uint8),
                ("tpl_opt_flag", C.c_uint8), ("use_pred_sad_in_intra_search", C.c_uint8),
                ("use_pred_sad_in_inter_search", C.c_uint8), ("enable_me_16x16", C.c_uint8), ("n_pu", C.c_uint8),
                ("max_cand", C.c_uint8), ("max_refs", C.c_uint8), ("max_l0", C.c_uint8),
                ("n_ref", C.c_uint8 * MAX_LISTS), ("subsample_tx", C.c_uint8), ("pf_shape", C.c_uint8), ("src_dist", C.c_uint8),
                ("y_round_fp", C.c_int16 * 2), ("y_quant_fp", C.c_int16 * 2), ("y_dequant", C.c_int16 * 2), ("intra_mode_end", C.c_uint8),
                ("reserved", C.c_uint8 * 3),
                ("total_me_candidate_index", C.POINTER(C.c_uint8)), ("me_candidate_array", C.POINTER(C.c_uint8)),
                ("me_mv_array", C.POINTER(C.c_uint32))]


class GmWarp(C.Structure):
    _fields_ = [("wmmat", C.c_int32 * 6), ("alpha", C.c_int16), ("beta", C.c_int16), ("gamma", C.c_int16), ("delta", C.c_int16)]
STATIC_MAX_B64 = 2040


class GmModel(C.Structure):
    _fields_ = [("wmmat", C.c_int32 * 6), ("alpha", C.c_int16), ("beta", C.c_int16), ("gamma", C.c_int16), ("delta", C.c_int16),
                ("wmtype", C.c_int32)]


def static_result_bytes(n_b64, n_refs):
    """SVTME_STATIC_RESULT_BYTES of include/svtme.h."""
    return n_b64 * 256 + 4096 + n_refs * n_b64 * PU_COUNT * 8


_lib = None


def load_library():
    """Loads libsvtme.so (no GPU needed to load it or to resolve its symbols)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run svt-av1_amd/build.sh (or __graft_entry__.build())" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.svtme_version.restype = C.c_char_p
    lib.svtme_derive_controls.argtypes = [C.POINTER(PresetKey), C.POINTER(Controls)]
    lib.svtme_derive_tf_controls.argtypes = [C.c_uint8, C.POINTER(Controls)]
    lib.svtme_derive_first_pass_controls.argtypes = [C.c_uint8, C.c_uint32, C.c_uint8, C.POINTER(Controls)]
    lib.svtme_input_resolution.argtypes = [C.c_uint32, C.c_uint32]
    lib.svtme_input_resolution.restype = C.c_uint8
    lib.svtme_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
    lib.svtme_destroy.argtypes = [C.c_void_p]
    lib.svtme_destroy.restype = None
    lib.svtme_store_put.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.svtme_store_release.argtypes = [C.c_void_p, C.c_uint64]
    lib.svtme_store_put_pinned.argtypes = lib.svtme_store_put.argtypes
    lib.svtme_store_wait.argtypes = [C.c_void_p, C.c_uint64]
    lib.svtme_store_put_many.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p),
                                         C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
    lib.svtme_store_copy_peer.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    lib.svtme_group_pictures_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]
    lib.svtme_group_ticket_engine.argtypes = [C.c_uint64]
    lib.svtme_group_ticket_engine.restype = C.c_uint32
    lib.svtme_group_set_peer_copies.argtypes = [C.c_void_p, C.c_int]
    lib.svtme_group_peer_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.svtme_host_register.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.svtme_host_unregister.argtypes = [C.c_void_p, C.c_void_p]
    lib.svtme_host_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    lib.svtme_host_free.argtypes = [C.c_void_p, C.c_void_p]
    lib.svtme_store_read.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_size_t] + [C.POINTER(C.c_uint32)] * 4
    lib.svtme_picture_submit.argtypes = [C.c_void_p, C.POINTER(Picture), C.POINTER(C.c_uint64)]
    lib.svtme_pictures_submit.argtypes = [C.c_void_p, C.POINTER(Picture), C.c_uint32, C.POINTER(C.c_uint64)]
    lib.svtme_picture_wait.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(Results)]
    lib.svtme_picture_done.argtypes = [C.c_void_p, C.c_uint64]
    lib.svtme_picture_timing.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_float * 8)]
    lib.svtme_kernel_timing.argtypes = [C.c_void_p, C.c_int]
    lib.svtme_picture_ops.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64 * 6)]
    lib.svtme_unpack_b64.argtypes = [C.POINTER(Results), C.c_uint32] + [C.c_void_p] * 5
    lib.svtme_picture_search_static.argtypes = [C.c_void_p, C.POINTER(Picture), C.c_void_p, C.c_size_t, C.POINTER(Results)]
    lib.svtme_group_create.argtypes = [C.POINTER(C.c_int32), C.c_uint32, C.POINTER(Config), C.POINTER(C.c_void_p)]
    lib.svtme_group_destroy.argtypes = [C.c_void_p]
    lib.svtme_group_destroy.restype = None
    lib.svtme_group_size.argtypes = [C.c_void_p]
    lib.svtme_group_size.restype = C.c_uint32
    lib.svtme_group_store_put.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
    lib.svtme_group_store_release.argtypes = [C.c_void_p, C.c_uint64]
    lib.svtme_group_picture_submit.argtypes = [C.c_void_p, C.POINTER(Picture), C.POINTER(C.c_uint64)]
    lib.svtme_group_picture_wait.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(Results)]
    lib.svtme_group_picture_done.argtypes = [C.c_void_p, C.c_uint64]
    lib.svtme_group_stats.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint64)] * 3
    lib.svtme_group_engine_for.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_void_p)]
    lib.svtme_group_gm_set_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.svtme_gm_set_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.svtme_gm_corners.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    lib.svtme_gm_correspondences.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.POINTER(C.c_int32)]
    lib.svtme_gm_warp_error.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(GmWarp), C.c_int64, C.POINTER(C.c_int64)]
    lib.svtme_gm_frame_error.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_int64)]
    lib.svtme_gm_refine.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_int64),
                                    C.POINTER(C.c_uint32)]
    lib.svtme_tpl_search.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.svtme_tpl_recon.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    lib.svtme_tpl_recon_keep.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
    lib.svtme_set_tile_rects.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
    lib.svtme_group_set_tile_rects.argtypes = lib.svtme_set_tile_rects.argtypes
    _lib = lib
    return lib


def input_resolution(width, height):
    return int(load_library().svtme_input_resolution(width, height))


def uniform_tile_rects(width, height, tile_w_b64=0, tile_h_b64=0):
    """svtme_tile_rect records [n_b64, 4] (start_x, start_y, end_x, end_y) of a grid of tiles of tile_w_b64 x tile_h_b64
    64x64 blocks (0 = one tile in that direction), tile ends clipped to the picture: what
    EbResourceCoordinationProcess.c:1304-1349 leaves in scs->sb_params_array for uniform tiles."""
    bw, bh = (width + 63) // 64, (height + 63) // 64
    r = np.zeros((bh, bw, 4), np.uint32)
    r[..., 2], r[..., 3] = width, height
    if tile_w_b64:
        sx = (np.arange(bw) // tile_w_b64) * tile_w_b64 * 64
        r[..., 0] = sx[None, :]
        r[..., 2] = np.minimum(sx + tile_w_b64 * 64, width)[None, :]
    if tile_h_b64:
        sy = (np.arange(bh) // tile_h_b64) * tile_h_b64 * 64
        r[..., 1] = sy[:, None]
        r[..., 3] = np.minimum(sy + tile_h_b64 * 64, height)[:, None]
    return r.reshape(bw * bh, 4)


def _set_tile_rects(fn, name, handle, width, height, rects):
    if rects is None:
        rc = fn(handle, width, height, None, 0)
    else:
        arr = np.ascontiguousarray(rects, np.uint32).reshape(-1, 4)
        rc = fn(handle, width, height, arr.ctypes.data, arr.shape[0])
    if rc:
        raise SvtmeError(rc, name)


def derive_controls(enc_mode, width, height, sc_class1=0, frame_rate=30 << 16):
    key = PresetKey(enc_mode, sc_class1, input_resolution(width, height), 0, frame_rate)
    c = Controls()
    rc = load_library().svtme_derive_controls(C.byref(key), C.byref(c))
    if rc:
        raise SvtmeError(rc, "svtme_derive_controls")
    return c


def derive_tf_controls(hme_me_level):
    """Controls of the temporal filter's motion search for tf_ctrls.hme_me_level 0..2."""
    c = Controls()
    rc = load_library().svtme_derive_tf_controls(hme_me_level, C.byref(c))
    if rc:
        raise SvtmeError(rc, "svtme_derive_tf_controls")
    return c


def derive_first_pass_controls(width, height, frame_rate=30 << 16, reduce_me_search=0):
    """Controls of the first-pass ME of a two-pass encode."""
    c = Controls()
    rc = load_library().svtme_derive_first_pass_controls(input_resolution(width, height), frame_rate, reduce_me_search, C.byref(c))
    if rc:
        raise SvtmeError(rc, "svtme_derive_first_pass_controls")
    return c


def results_to_arrays(r, copy=True):
    """A Results struct (filled by a wait call) as a dict of numpy arrays shaped [n_b64, 2, 4, ...]."""
    n = r.n_b64
    out = {}
    if r.p_sb_best_sad and copy:  # NULL for pictures submitted with tail.enable == 2
        # one [n_b64][85] plane per reference of the picture (list 0 first) -> MeContext's [list][ref][pu] order
        n_ref = [int(r.n_ref[0]), int(r.n_ref[1])]
        for name, ptr in (("best_sad", r.p_sb_best_sad), ("best_mv", r.p_sb_best_mv)):
            planes = np.ctypeslib.as_array(ptr, (n_ref[0] + n_ref[1], n, 85))
            full = np.zeros((n, 2, 4, 85), np.uint32)
            for l in range(2):
                for i in range(n_ref[l]):
                    full[:, l, i] = planes[(n_ref[0] if l else 0) + i]
            out[name] = full
    out.update({
        "zz_sad": np.ctypeslib.as_array(r.zz_sad, (n, 2, 4)),
        "sr_divisor": np.ctypeslib.as_array(r.reduce_me_sr_divisor, (n, 2, 4)),
    })
    raw = np.frombuffer(C.string_at(r.search_results, n * 8 * C.sizeof(SearchResult)), SEARCH_RESULT_DTYPE).reshape(n, 2, 4)
    sr = np.zeros(raw.shape, SEARCH_RESULT_FIELDS)  # the named fields only, without the C struct's padding bytes
    for f in SEARCH_RESULT_FIELDS.names:
        sr[f] = raw[f]
    out["search_results"] = sr
    if r.me_mv_array:  # the picture ran the host tail on the device
        out["total_me_candidate_index"] = np.ctypeslib.as_array(r.total_me_candidate_index, (n, 85))
        out["me_mv_array"] = np.ctypeslib.as_array(r.me_mv_array, (n, 85 * TAIL_MV_STRIDE))
        out["me_candidate_array"] = np.ctypeslib.as_array(r.me_candidate_array, (n, 85 * TAIL_CAND_STRIDE))
        out["me_distortion"] = np.ctypeslib.as_array(r.me_distortion, (n, 85))
        out["b64_stats"] = np.ctypeslib.as_array(r.b64_stats, (n, 8))
        out["final_do_ref"] = np.ctypeslib.as_array(r.final_do_ref, (n, 2, 4))
        out["final_hme_sad"] = np.ctypeslib.as_array(r.final_hme_sad, (n, 2, 4))
    if copy:
        out = {k: np.array(v) for k, v in out.items()}
    return out


class Group:
    """svtme_group: one engine per entry of `devices`; pictures go round robin, sources travel where they are read."""

    def __init__(self, devices, max_width, max_height, store_slots=24, max_in_flight=4, search_streams=0, kernel_timing=0):
        self.lib = load_library()
        self.h = C.c_void_p()
        cfg = Config(0, max_width, max_height, store_slots, max_in_flight, search_streams, kernel_timing, 0)
        dev = (C.c_int32 * len(devices))(*devices)
        rc = self.lib.svtme_group_create(dev, len(devices), C.byref(cfg), C.byref(self.h))
        if rc:
            raise SvtmeError(rc, "svtme_group_create")
        self._keep = {}  # key -> the announced luma array (the group reads it when a GPU first needs it)

    def close(self):
        if self.h:
            self.lib.svtme_group_destroy(self.h)
            self.h = C.c_void_p()
        self._keep.clear()

    def store_put(self, key, picture_number, luma, pinned=0):
        assert luma.dtype == np.uint8 and luma.strides[1] == 1
        h, w = luma.shape
        self._keep[key] = luma
        rc = self.lib.svtme_group_store_put(self.h, key, picture_number, luma.ctypes.data, luma.strides[0], w, h, pinned)
        if rc:
            raise SvtmeError(rc, "svtme_group_store_put")

    def store_release(self, key):
        rc = self.lib.svtme_group_store_release(self.h, key)
        self._keep.pop(key, None)
        if rc:
            raise SvtmeError(rc, "svtme_group_store_release")

    make_picture = None  # set below: the same description builder as Engine's

    def submit(self, pic):
        t = C.c_uint64()
        rc = self.lib.svtme_group_picture_submit(self.h, C.byref(pic), C.byref(t))
        if rc:
            raise SvtmeError(rc, "svtme_group_picture_submit")
        return t.value

    def submit_batch(self, pics):
        """svtme_group_pictures_submit: the pictures are split over the engines by accumulated work, one batch per engine."""
        n = len(pics)
        arr = (Picture * n)(*pics)
        t = (C.c_uint64 * n)()
        rc = self.lib.svtme_group_pictures_submit(self.h, arr, n, t)
        if rc:
            raise SvtmeError(rc, "svtme_group_pictures_submit")
        return list(t)

    def engine_of(self, ticket):
        return int(self.lib.svtme_group_ticket_engine(ticket))

    def set_peer_copies(self, enable):
        rc = self.lib.svtme_group_set_peer_copies(self.h, int(bool(enable)))
        if rc:
            raise SvtmeError(rc, "svtme_group_set_peer_copies")

    def wait(self, ticket, copy=True):
        r = Results()
        rc = self.lib.svtme_group_picture_wait(self.h, ticket, C.byref(r))
        if rc:
            raise SvtmeError(rc, "svtme_group_picture_wait")
        return results_to_arrays(r, copy)

    def wait_raw(self, ticket):
        r = Results()
        rc = self.lib.svtme_group_picture_wait(self.h, ticket, C.byref(r))
        if rc:
            raise SvtmeError(rc, "svtme_group_picture_wait")
        return r

    def done(self, ticket):
        rc = self.lib.svtme_group_picture_done(self.h, ticket)
        if rc:
            raise SvtmeError(rc, "svtme_group_picture_done")

    def engine_for(self, keys):
        """The group's engine on which all `keys` are resident (uploading what is missing), as a borrowed Engine: for
        tpl_search / gm_* calls. Do not close it."""
        arr = (C.c_uint64 * len(keys))(*keys)
        h = C.c_void_p()
        rc = self.lib.svtme_group_engine_for(self.h, arr, len(keys), C.byref(h))
        if rc:
            raise SvtmeError(rc, "svtme_group_engine_for")
        e = Engine.__new__(Engine)
        e.lib, e.h, e.borrowed = self.lib, h, True
        return e

    def gm_set_tables(self, warped_filter, error_lut):
        wf = np.ascontiguousarray(warped_filter, np.int16)
        lut = np.ascontiguousarray(error_lut, np.int32)
        rc = self.lib.svtme_group_gm_set_tables(self.h, wf.ctypes.data, lut.ctypes.data)
        if rc:
            raise SvtmeError(rc, "svtme_group_gm_set_tables")

    def set_tile_rects(self, width, height, rects):
        _set_tile_rects(self.lib.svtme_group_set_tile_rects, "svtme_group_set_tile_rects", self.h, width, height, rects)

    def stats(self):
        v = [C.c_uint64() for _ in range(3)]
        rc = self.lib.svtme_group_stats(self.h, *[C.byref(x) for x in v])
        if rc:
            raise SvtmeError(rc, "svtme_group_stats")
        p = [C.c_uint64() for _ in range(2)]
        rc = self.lib.svtme_group_peer_stats(self.h, *[C.byref(x) for x in p])
        if rc:
            raise SvtmeError(rc, "svtme_group_peer_stats")
        return dict(pictures=v[0].value, sources=v[1].value, uploads=v[2].value, peer_copies=p[0].value, peer_links=p[1].value)


class Engine:
    """One engine = one GPU. Pictures are put into the device store once and referenced by key afterwards."""

    def __init__(self, max_width, max_height, store_slots=24, max_in_flight=4, device=0, search_streams=0, kernel_timing=1,
                 serial_hme=0):
        self.lib = load_library()
        self.h = C.c_void_p()
        cfg = Config(device, max_width, max_height, store_slots, max_in_flight, search_streams, kernel_timing, serial_hme)
        rc = self.lib.svtme_create(C.byref(cfg), C.byref(self.h))
        if rc:
            raise SvtmeError(rc, "svtme_create")

    def close(self):
        if self.h and not getattr(self, "borrowed", False):  # a group's engine belongs to the group
            self.lib.svtme_destroy(self.h)
        self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def store_put(self, key, picture_number, luma):
        luma = np.ascontiguousarray(luma, dtype=np.uint8)
        h, w = luma.shape
        rc = self.lib.svtme_store_put(self.h, key, picture_number, luma.ctypes.data, w, w, h)
        if rc:
            raise SvtmeError(rc, "svtme_store_put")

    def host_alloc(self, shape):
        """A uint8 numpy array over page-locked host memory (svtme_host_alloc); give it back with host_free(array)."""
        n = int(np.prod(shape))
        p = C.c_void_p()
        rc = self.lib.svtme_host_alloc(self.h, n, C.byref(p))
        if rc:
            raise SvtmeError(rc, "svtme_host_alloc")
        return np.ctypeslib.as_array((C.c_uint8 * n).from_address(p.value)).reshape(shape)

    def host_free(self, array):
        rc = self.lib.svtme_host_free(self.h, array.ctypes.data)
        if rc:
            raise SvtmeError(rc, "svtme_host_free")

    def host_register(self, array):
        """Page-locks a numpy array's memory for store_put_pinned; keep the array alive until host_unregister."""
        rc = self.lib.svtme_host_register(self.h, array.ctypes.data, array.nbytes)
        if rc:
            raise SvtmeError(rc, "svtme_host_register")

    def host_unregister(self, array):
        rc = self.lib.svtme_host_unregister(self.h, array.ctypes.data)
        if rc:
            raise SvtmeError(rc, "svtme_host_unregister")

    def store_put_pinned(self, key, picture_number, luma):
        """luma: a 2-D uint8 view whose rows are contiguous (any row stride); copied by DMA as it lies, after the call
        returns: leave it unchanged until store_wait(key) or a wait of a picture that reads it."""
        h, w = luma.shape
        assert luma.dtype == np.uint8 and luma.strides[1] == 1
        rc = self.lib.svtme_store_put_pinned(self.h, key, picture_number, luma.ctypes.data, luma.strides[0], w, h)
        if rc:
            raise SvtmeError(rc, "svtme_store_put_pinned")

    def store_put_many(self, keys, picture_numbers, lumas, pinned=False):
        """svtme_store_put_many: 2-D uint8 arrays of one size and one row stride (contiguous rows), at most MAX_PUT_BATCH.
        pinned: the arrays lie in page-locked memory (host_alloc) and stay unchanged until their store_wait."""
        n = len(keys)
        if not pinned:
            lumas = [np.ascontiguousarray(a, dtype=np.uint8) for a in lumas]
        h, w = lumas[0].shape
        stride = lumas[0].strides[0]
        assert all(a.shape == (h, w) and a.strides == (stride, 1) and a.dtype == np.uint8 for a in lumas)
        k = (C.c_uint64 * n)(*keys)
        pn = (C.c_uint64 * n)(*picture_numbers)
        ptr = (C.c_void_p * n)(*[a.ctypes.data for a in lumas])
        rc = self.lib.svtme_store_put_many(self.h, n, k, pn, ptr, stride, w, h, 1 if pinned else 0)
        if rc:
            raise SvtmeError(rc, "svtme_store_put_many")

    def store_copy_from(self, other, key):
        """svtme_store_copy_peer: the built planes of `key` from another engine's store, device to device."""
        rc = self.lib.svtme_store_copy_peer(self.h, other.h, key)
        if rc:
            raise SvtmeError(rc, "svtme_store_copy_peer")

    def store_wait(self, key):
        rc = self.lib.svtme_store_wait(self.h, key)
        if rc:
            raise SvtmeError(rc, "svtme_store_wait")

    def store_release(self, key):
        rc = self.lib.svtme_store_release(self.h, key)
        if rc:
            raise SvtmeError(rc, "svtme_store_release")

    def store_read(self, key, level, shape_hint):
        h, w = shape_hint
        buf = np.zeros((h + 144) * (w + 144), np.uint8)
        st, pw, ph, pad = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        rc = self.lib.svtme_store_read(self.h, key, level, buf.ctypes.data, buf.size, C.byref(st), C.byref(pw),
                                       C.byref(ph), C.byref(pad))
        if rc:
            raise SvtmeError(rc, "svtme_store_read")
        rows = ph.value + 2 * pad.value
        return buf[:rows * st.value].reshape(rows, st.value).copy()

    def make_picture(self, width, height, poc, temporal_layer, is_ref, refs, ctrl, key_of=lambda poc: poc, tail=None,
                     tf_me_exit_th=None, first_pass_bypass_zz=None, restricted_motion_vector=0, number_of=lambda poc: poc, b64_step=0):
        """refs: [[pocs of list 0], [pocs of list 1]]; number_of: poc -> picture_number (the search areas grow with the
        picture-number distance); tail: a Tail to run the host tail on the device too;
        tf_me_exit_th: not None makes it a temporal-filter search (ME_MCTF) of refs[0][0] with that exit threshold;
        first_pass_bypass_zz: not None makes it a first-pass search (ME_FIRST_PASS) with that bypass_zz_check;
        b64_step: n > 1 searches only the 64x64 blocks whose column and row index are multiples of n."""
        p = Picture()
        p.b64_step = b64_step
        p.cur_key = key_of(poc)
        p.picture_number = number_of(poc)
        p.width, p.height = width, height
        p.temporal_layer_index = temporal_layer
        p.is_used_as_reference_flag = is_ref
        p.num_of_list_to_search = 2 if refs[1] else 1
        for l in range(2):
            p.num_of_ref_pic_to_search[l] = len(refs[l])
            for i, r in enumerate(refs[l]):
                p.ref_key[l][i] = key_of(r)
                p.ref_picture_number[l][i] = number_of(r)
        p.ctrl = ctrl
        if tail is not None:
            p.tail = tail
        if tf_me_exit_th is not None:
            p.me_type = ME_MCTF
            p.tf_me_exit_th = tf_me_exit_th
        p.restricted_motion_vector = restricted_motion_vector
        if first_pass_bypass_zz is not None:
            p.me_type = ME_FIRST_PASS
            p.bypass_zz_check = first_pass_bypass_zz
        return p

    def submit(self, pic):
        t = C.c_uint64()
        rc = self.lib.svtme_picture_submit(self.h, C.byref(pic), C.byref(t))
        if rc:
            raise SvtmeError(rc, "svtme_picture_submit")
        return t.value

    def submit_batch(self, pics):
        """svtme_pictures_submit: the pictures share their kernel launches; returns their tickets."""
        n = len(pics)
        arr = (Picture * n)(*pics)
        t = (C.c_uint64 * n)()
        rc = self.lib.svtme_pictures_submit(self.h, arr, n, t)
        if rc:
            raise SvtmeError(rc, "svtme_pictures_submit")
        return list(t)

    def wait(self, ticket, copy=True):
        """Returns a dict of numpy arrays shaped [n_b64, 2, 4, ...]."""
        r = Results()
        rc = self.lib.svtme_picture_wait(self.h, ticket, C.byref(r))
        if rc:
            raise SvtmeError(rc, "svtme_picture_wait")
        return results_to_arrays(r, copy)

    def wait_raw(self, ticket):
        """svtme_picture_wait without the conversion to numpy: the Results struct, whose pointers lie in the engine's pinned
        host buffers (valid until done()). What a C caller gets; for loops that only need the results to have landed."""
        r = Results()
        rc = self.lib.svtme_picture_wait(self.h, ticket, C.byref(r))
        if rc:
            raise SvtmeError(rc, "svtme_picture_wait")
        return r

    def search_static(self, pic, result_block, result_bytes):
        """svtme_picture_search_static: the picture's search without any allocation, results into caller-owned memory the
        device can write (`result_block`: an address). Returns the Results struct, whose pointers lie inside the block."""
        r = Results()
        rc = self.lib.svtme_picture_search_static(self.h, C.byref(pic), result_block, result_bytes, C.byref(r))
        if rc:
            raise SvtmeError(rc, "svtme_picture_search_static")
        return r

    def set_tile_rects(self, width, height, rects):
        """svtme_set_tile_rects: the sequence's tiles per 64x64 block ([n_b64, 4] as uniform_tile_rects gives them), which
        bound the windows of restricted temporal-filter and first-pass searches; None clears them."""
        _set_tile_rects(self.lib.svtme_set_tile_rects, "svtme_set_tile_rects", self.h, width, height, rects)

    # ---- global-motion estimation (svtme_gm_*): downsample level 0 / 1 / 2 = full / quarter / sixteenth plane
    def gm_set_tables(self, warped_filter, error_lut):
        wf = np.ascontiguousarray(warped_filter, np.int16)
        lut = np.ascontiguousarray(error_lut, np.int32)
        if wf.size != 193 * 8 or lut.size != 512:
            raise ValueError("warp filter [193, 8] int16 and error table [512] int32 expected")
        rc = self.lib.svtme_gm_set_tables(self.h, wf.ctypes.data, lut.ctypes.data)
        if rc:
            raise SvtmeError(rc, "svtme_gm_set_tables")

    def gm_corners(self, key, level=0, max_points=GM_MAX_CORNERS):
        pts = np.zeros((max(max_points, 1), 2), np.int32)
        n = C.c_int32(0)
        rc = self.lib.svtme_gm_corners(self.h, key, level, pts.ctypes.data, max_points, C.byref(n))
        if rc:
            raise SvtmeError(rc, "svtme_gm_corners")
        return pts[:n.value].copy()

    def gm_correspondences(self, cur_key, ref_key, level=0):
        out = np.zeros((GM_MAX_CORNERS, 4), np.int32)
        n = C.c_int32(0)
        rc = self.lib.svtme_gm_correspondences(self.h, cur_key, ref_key, level, out.ctypes.data, C.byref(n))
        if rc:
            raise SvtmeError(rc, "svtme_gm_correspondences")
        return out[:n.value].copy()

    def gm_warp_error(self, cur_key, ref_key, level, mat, shear, best_error=(1 << 63) - 1):
        w = GmWarp((C.c_int32 * 6)(*[int(v) for v in mat]), *[int(v) for v in shear])
        err = C.c_int64(0)
        rc = self.lib.svtme_gm_warp_error(self.h, cur_key, ref_key, level, C.byref(w), int(best_error), C.byref(err))
        if rc:
            raise SvtmeError(rc, "svtme_gm_warp_error")
        return err.value

    def gm_frame_error(self, cur_key, ref_key, level=0):
        err = C.c_int64(0)
        rc = self.lib.svtme_gm_frame_error(self.h, cur_key, ref_key, level, C.byref(err))
        if rc:
            raise SvtmeError(rc, "svtme_gm_frame_error")
        return err.value

    def gm_refine(self, cur_key, ref_key, level, mat, wmtype, steps, best_frame_error=(1 << 63) - 1, shear=(0, 0, 0, 0), model_wmtype=None):
        """svt_av1_refine_integerized_param as one call -> (mat [6], shear [4], wmtype, error, evaluations)."""
        m = GmModel((C.c_int32 * 6)(*[int(v) for v in mat]), *[int(v) for v in shear], int(wmtype if model_wmtype is None else model_wmtype))
        err, n = C.c_int64(0), C.c_uint32(0)
        rc = self.lib.svtme_gm_refine(self.h, cur_key, ref_key, level, C.byref(m), int(wmtype), int(steps), int(best_frame_error), C.byref(err),
                                      C.byref(n))
        if rc:
            raise SvtmeError(rc, "svtme_gm_refine")
        return list(m.wmmat), [m.alpha, m.beta, m.gamma, m.delta], int(m.wmtype), err.value, n.value

    def tpl_search(self, pic):
        """pic: a TplPicture (or any ctypes structure of that layout). Returns TPL_BLOCK records [n_b64, 21]."""
        n_b64 = ((pic.width + 63) // 64) * ((pic.height + 63) // 64)
        out = np.zeros((n_b64, TPL_BLOCKS), TPL_BLOCK)
        rc = self.lib.svtme_tpl_search(self.h, C.addressof(pic), out.ctypes.data)
        if rc:
            raise SvtmeError(rc, "svtme_tpl_search")
        return out

    def tpl_recon(self, pic, blocks, want_picture=True):
        """svtme_tpl_recon: pic as for tpl_search, blocks what tpl_search returned (TPL_BLOCK records [n_b64, 21]). Returns
        (TPL_STATS records [n_b64, 21], the reconstructed picture uint8 [height, width] or None)."""
        blocks = np.ascontiguousarray(blocks, TPL_BLOCK)
        stats = np.zeros(blocks.shape, TPL_STATS)
        recon = np.zeros((pic.height, pic.width), np.uint8) if want_picture else None
        rc = self.lib.svtme_tpl_recon(self.h, C.addressof(pic), blocks.ctypes.data, stats.ctypes.data,
                                      recon.ctypes.data if want_picture else None, pic.width)
        if rc:
            raise SvtmeError(rc, "svtme_tpl_recon")
        return stats, recon

    def tpl_recon_keep(self, key, picture_number=0):
        """svtme_tpl_recon_keep: the last reconstructed picture becomes a picture of the store under `key`, device to device."""
        rc = self.lib.svtme_tpl_recon_keep(self.h, key, picture_number)
        if rc:
            raise SvtmeError(rc, "svtme_tpl_recon_keep")

    def kernel_timing(self, on):
        """Per-kernel events on the submissions that follow: on / off (svtme_kernel_timing)."""
        rc = self.lib.svtme_kernel_timing(self.h, 1 if on else 0)
        if rc:
            raise SvtmeError(rc, "svtme_kernel_timing")

    def timing(self, ticket):
        ms = (C.c_float * 8)()
        rc = self.lib.svtme_picture_timing(self.h, ticket, C.byref(ms))
        if rc:
            raise SvtmeError(rc, "svtme_picture_timing")
        return list(ms)

    def ops(self, ticket):
        ops = (C.c_uint64 * 6)()
        rc = self.lib.svtme_picture_ops(self.h, ticket, C.byref(ops))
        if rc:
            raise SvtmeError(rc, "svtme_picture_ops")
        return [int(v) for v in ops]

    def done(self, ticket):
        rc = self.lib.svtme_picture_done(self.h, ticket)
        if rc:
            raise SvtmeError(rc, "svtme_picture_done")

    def run_picture(self, pic):
        t = self.submit(pic)
        out = self.wait(t)
        out["timing_ms"] = self.timing(t)
        out["ops"] = self.ops(t)
        self.done(t)
        return out


Group.make_picture = Engine.make_picture
