"""TEST INFRASTRUCTURE: ctypes access to the CPU restatement of the TPL dispenser's source-based mode search
(oracle/tpl_oracle.c) and the structures it shares with the engine's C ABI (include/svtme.h). Not part of the product."""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_LIB = os.path.join(HERE, "build", "libme_oracle.so")
TPL_BLOCKS = 21
INT64_MAX = (1 << 63) - 1

# tpl_blk_idx_tab (EbSourceBasedOperationsProcess.c:514-516): position and size of block i inside its 64x64 block
BLOCK_GEOMETRY = [(0, 0, 64)] + [((i & 1) * 32, (i >> 1) * 32, 32) for i in range(4)] + \
    [((q & 1) * 32 + (s & 1) * 16, (q >> 1) * 32 + (s >> 1) * 16, 16) for q in range(4) for s in range(4)]

TPL_BLOCK = np.dtype([("best_intra_cost", "<i8"), ("best_inter_cost", "<i8"), ("mv_x", "<i2"), ("mv_y", "<i2"),
                      ("best_rf_idx", "i1"), ("best_is_inter", "u1"), ("best_intra_mode", "u1"), ("valid", "u1"),
                      ("src_recon_error", "<i8"), ("src_rate", "<i8")])
assert TPL_BLOCK.itemsize == 40


CONTROL_FIELDS = ("dispenser_search_level", "disable_intra_pred", "tpl_opt_flag", "use_pred_sad_in_intra_search",
                  "use_pred_sad_in_inter_search", "subsample_tx", "pf_shape")


class TplPicture(ctypes.Structure):
    _fields_ = [("cur_key", ctypes.c_uint64), ("ref_key", (ctypes.c_uint64 * 4) * 2), ("ref_skip", (ctypes.c_uint8 * 4) * 2),
                ("width", ctypes.c_uint32), ("height", ctypes.c_uint32),
                ("i_slice", ctypes.c_uint8), ("dispenser_search_level", ctypes.c_uint8), ("disable_intra_pred", ctypes.c_uint8),
                ("tpl_opt_flag", ctypes.c_uint8), ("use_pred_sad_in_intra_search", ctypes.c_uint8),
                ("use_pred_sad_in_inter_search", ctypes.c_uint8), ("enable_me_16x16", ctypes.c_uint8), ("n_pu", ctypes.c_uint8),
                ("max_cand", ctypes.c_uint8), ("max_refs", ctypes.c_uint8), ("max_l0", ctypes.c_uint8),
                ("n_ref", ctypes.c_uint8 * 2), ("subsample_tx", ctypes.c_uint8), ("pf_shape", ctypes.c_uint8),
                ("src_dist", ctypes.c_uint8), ("y_round_fp", ctypes.c_int16 * 2), ("y_quant_fp", ctypes.c_int16 * 2),
                ("y_dequant", ctypes.c_int16 * 2), ("intra_mode_end", ctypes.c_uint8), ("reserved", ctypes.c_uint8 * 3),
                ("total_me_candidate_index", ctypes.POINTER(ctypes.c_uint8)),
                ("me_candidate_array", ctypes.POINTER(ctypes.c_uint8)),
                ("me_mv_array", ctypes.POINTER(ctypes.c_uint32))]


def make_picture(width, height, ctrl, me, cur_key=0, ref_keys=((), ()), ref_skip=None, i_slice=0, n_ref=None, quant=None):
    """ctrl: dict with dispenser_search_level, disable_intra_pred, tpl_opt_flag, use_pred_sad_in_intra_search,
    use_pred_sad_in_inter_search, subsample_tx, pf_shape; quant: None, or (y_round_fp, y_quant_fp, y_dequant) pairs [dc, ac],
    which also asks for the source distortion (src_dist); me: dict with total_me_candidate_index [n_b64, n_pu] u8, me_candidate_array
    [n_b64, n_pu * max_cand] u8, me_mv_array [n_b64, n_pu * max_refs] u32, enable_me_16x16, max_cand, max_refs, max_l0.
    Returns (TplPicture, keep-alive tuple)."""
    cnt = np.ascontiguousarray(me["total_me_candidate_index"], np.uint8)
    cand = np.ascontiguousarray(me["me_candidate_array"], np.uint8)
    mv = np.ascontiguousarray(me["me_mv_array"], np.uint32)
    n_pu = cnt.shape[1]
    assert cand.shape[1] == n_pu * me["max_cand"] and mv.shape[1] == n_pu * me["max_refs"]
    p = TplPicture()
    p.cur_key = cur_key
    for l in range(2):
        for r in range(4):
            p.ref_key[l][r] = ref_keys[l][r] if r < len(ref_keys[l]) else 0
            p.ref_skip[l][r] = int(ref_skip[l][r]) if ref_skip is not None else 0
    p.width, p.height, p.i_slice = width, height, i_slice
    p.n_ref[0], p.n_ref[1] = n_ref if n_ref is not None else (len(ref_keys[0]), len(ref_keys[1]))
    for k in CONTROL_FIELDS:
        setattr(p, k, int(ctrl[k]))
    p.intra_mode_end = int(ctrl.get("intra_mode_end", 12))  # PAETH_PRED: scs->enable_paeth / enable_smooth at their defaults
    if quant is not None:
        p.src_dist = 1
        for k in range(2):
            p.y_round_fp[k], p.y_quant_fp[k], p.y_dequant[k] = int(quant[0][k]), int(quant[1][k]), int(quant[2][k])
    p.enable_me_16x16, p.n_pu = int(me["enable_me_16x16"]), n_pu
    p.max_cand, p.max_refs, p.max_l0 = int(me["max_cand"]), int(me["max_refs"]), int(me["max_l0"])
    p.total_me_candidate_index = cnt.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
    p.me_candidate_array = cand.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
    p.me_mv_array = mv.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
    return p, (cnt, cand, mv)


_lib = None


def oracle_search(cur, refs, pic):
    """cur: uint8 [h, w]; refs: {(list, ref): uint8 [h, w]}; pic: TplPicture. Returns TPL_BLOCK records [n_b64, 21]."""
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(ORACLE_LIB)
    h, w = cur.shape
    cur = np.ascontiguousarray(cur, np.uint8)
    keep = {k: np.ascontiguousarray(v, np.uint8) for k, v in refs.items()}
    ptrs = (ctypes.POINTER(ctypes.c_uint8) * 8)()
    for (l, r), a in keep.items():
        assert a.shape == cur.shape
        ptrs[l * 4 + r] = a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
    n_b64 = ((w + 63) // 64) * ((h + 63) // 64)
    out = np.zeros((n_b64, TPL_BLOCKS), TPL_BLOCK)
    _lib.orc_tpl_search.restype = ctypes.c_int
    rc = _lib.orc_tpl_search(cur.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), w, ptrs, ctypes.byref(pic),
                             out.ctypes.data_as(ctypes.c_void_p))
    if rc != 0:
        raise RuntimeError("orc_tpl_search: %d" % rc)
    return out


TPL_STATS = np.dtype([("srcrf_dist", "<i8"), ("recrf_dist", "<i8"), ("srcrf_rate", "<i8"), ("recrf_rate", "<i8")])


def recon_available(ctrl):
    """The reconstruction path is restated for 16x16 blocks (tpl levels 1 to 6) and for 32x32 blocks with the 32x8 transform
    (tpl level 7)."""
    return ctrl["dispenser_search_level"] == 0 or (ctrl["dispenser_search_level"] == 1 and ctrl["subsample_tx"] == 2 and ctrl["tpl_opt_flag"] == 1)


def oracle_recon(cur, refs, pic, blocks):
    """The dispenser's reconstruction path on what oracle_search found (pic with src_dist): returns (reconstructed picture
    uint8 [h, w], TPL_STATS records [n_b64, 21])."""
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(ORACLE_LIB)
    h, w = cur.shape
    cur = np.ascontiguousarray(cur, np.uint8)
    keep = {k: np.ascontiguousarray(v, np.uint8) for k, v in refs.items()}
    ptrs = (ctypes.POINTER(ctypes.c_uint8) * 8)()
    for (l, r), a in keep.items():
        ptrs[l * 4 + r] = a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
    blocks = np.ascontiguousarray(blocks, TPL_BLOCK)
    recon = np.zeros((h, w), np.uint8)
    stats = np.zeros(blocks.shape, TPL_STATS)
    _lib.orc_tpl_recon.restype = ctypes.c_int
    rc = _lib.orc_tpl_recon(cur.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), w, ptrs, ctypes.byref(pic),
                            blocks.ctypes.data_as(ctypes.c_void_p), recon.ctypes.data_as(ctypes.c_void_p),
                            stats.ctypes.data_as(ctypes.c_void_p))
    if rc != 0:
        raise RuntimeError("orc_tpl_recon: %d" % rc)
    return recon, stats


def recon_hash(recon):
    """The 64-bit FNV-1a hash the reference driver records of the reconstructed picture (oracle/ref_driver.c, tpl_dump)."""
    h = 0xcbf29ce484222325
    for v in np.ascontiguousarray(recon, np.uint8).tobytes():
        h = ((h ^ v) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


def me_inputs_of(pd, bd):
    """The ME results a reference dump holds (refio.PIC_DUMP record, refio.B64_DUMP records) in the shape make_picture takes."""
    return dict(total_me_candidate_index=bd["total_me_candidate_index"].copy(),
                me_candidate_array=bd["me_candidate_array"][:, :85 * int(pd["max_cand"])].copy(),
                me_mv_array=bd["me_mv_array"][:, :85 * int(pd["max_refs"])].copy(),
                enable_me_16x16=int(pd["enable_me_16x16"]), max_cand=int(pd["max_cand"]), max_refs=int(pd["max_refs"]),
                max_l0=int(pd["max_l0"]))


def controls_of(tp):
    """refio.TPL_PIC_DUMP record -> the control dict make_picture takes."""
    return {k: int(tp[k]) for k in CONTROL_FIELDS}


def src_dist_available(ctrl):
    """The source distortion exists for the transforms the tpl levels use: 16x16, 16x8, 16x4 on 16x16 blocks, 32x8 on 32x32
    blocks (whose incomplete 64x64 blocks fall back to 16x16 blocks with 16x4)."""
    return ctrl["dispenser_search_level"] == 0 or (ctrl["dispenser_search_level"] == 1 and ctrl["subsample_tx"] == 2)


def quant_of(tp):
    """refio.TPL_PIC_DUMP record -> the quantiser triple make_picture takes."""
    return (tp["y_round_fp"].tolist(), tp["y_quant_fp"].tolist(), tp["y_dequant"].tolist())


def expected_from_reference(tb, ctrl, src_dist=False):
    """What the reference's dump (refio.TPL_BLK_DUMP records [n_b64, 21]) says the search found, as TPL_BLOCK records: the
    costs come from the recorded cost calls (the intra cost first when intra prediction is on, then one per candidate);
    src_dist: src_recon_error from the recorded srcrf_dist = (recon_error << TPL_DEP_COST_SCALE_LOG2) << subsample_tx of the
    blocks an inter candidate won."""
    out = np.zeros(tb.shape, TPL_BLOCK)
    out["valid"] = tb["valid"]
    v = tb["valid"] == 1
    # without tpl_opt_flag every intra mode from DC_PRED to intra_mode_end is costed, in that order
    n_intra = 0 if ctrl["disable_intra_pred"] else (1 if ctrl["tpl_opt_flag"] else ctrl.get("intra_mode_end", 12) + 1)
    costs = tb["costs"].astype(np.int64)
    k = np.arange(costs.shape[-1])
    inter_mask = (k >= n_intra) & (k < tb["n_costs"][..., None])
    inter = np.where(inter_mask, costs, INT64_MAX).min(axis=-1)
    out["best_inter_cost"] = np.where(v, inter, 0)
    out["best_intra_cost"] = np.where(v, costs[..., :n_intra].min(axis=-1) if n_intra else INT64_MAX, 0)
    out["best_rf_idx"] = np.where(v, tb["best_rf_idx"], 0)
    out["mv_x"] = np.where(v, tb["mv_col"], 0)
    out["mv_y"] = np.where(v, tb["mv_row"], 0)
    out["best_is_inter"] = np.where(v, tb["best_mode"] != 0, 0)
    out["best_intra_mode"] = np.where(v, tb["best_intra_mode"], 0)
    if src_dist:
        out["src_recon_error"] = np.where(v & (tb["best_mode"] != 0), tb["src_srcrf_dist"] >> (4 + ctrl["subsample_tx"]), 0)
        out["src_rate"] = np.where(v & (tb["best_mode"] != 0), tb["src_srcrf_rate"] >> (4 + ctrl["subsample_tx"]), 0)
    return out


def oracle_intra_costs(cur, x, y):
    """The restatement's transform-domain costs of the thirteen intra modes of the 16x16 block at (x, y) of cur (uint8 [h, w])."""
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(ORACLE_LIB)
    cur = np.ascontiguousarray(cur, np.uint8)
    h, w = cur.shape
    out = np.zeros(13, np.int64)
    _lib.orc_tpl_intra_costs.restype = None
    _lib.orc_tpl_intra_costs(cur.ctypes.data_as(ctypes.c_void_p), w, w, h, int(x), int(y), out.ctypes.data_as(ctypes.c_void_p))
    return out


def oracle_fwd_dct(residual, shape=0):
    """residual: int16 [h, w] -> int32 [h, w] coefficients of the restatement's forward DCT."""
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(ORACLE_LIB)
    r = np.ascontiguousarray(residual, np.int16)
    h, w = r.shape
    out = np.zeros((h, w), np.int32)
    rc = _lib.orc_tpl_fwd_dct(r.ctypes.data_as(ctypes.c_void_p), w, h, out.ctypes.data_as(ctypes.c_void_p), int(shape))
    assert rc == 0, (w, h)
    return out
