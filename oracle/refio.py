"""TEST INFRASTRUCTURE: writer for the job files and reader for the dump files of oracle/_ref/ref_me (the real
reference ME behind a driver) and oracle/build/oracle_me (the restatement behind the same driver interface).
Formats: oracle/ref_dump.h. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use this."""
import os
import struct
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_BIN = os.path.join(HERE, "_ref", "ref_me")
ORACLE_BIN = os.path.join(HERE, "build", "oracle_me")

MODE_DUMP, MODE_BENCH, MODE_PLANES, MODE_DUMP_NOTAIL, MODE_TAIL, MODE_SHIM, MODE_TPL = 0, 1, 2, 3, 4, 5, 6

SA = np.dtype([("width", "<u2"), ("height", "<u2")])
SAMM = np.dtype([("sa_min", SA), ("sa_max", SA)])
CONTROLS = np.dtype([
    ("me_sa", SAMM), ("hme_l0_sa", SAMM), ("hme_l1_sa", SA), ("hme_l2_sa", SA),
    ("num_hme_sa_w", "<u2"), ("num_hme_sa_h", "<u2"),
    ("enable_hme_flag", "u1"), ("enable_hme_level0_flag", "u1"), ("enable_hme_level1_flag", "u1"),
    ("enable_hme_level2_flag", "u1"), ("hme_search_method", "u1"), ("me_search_method", "u1"),
    ("prehme_enable", "u1"), ("prehme_skip_search_line", "u1"), ("prehme_l1_early_exit", "u1"), ("reserved0", "u1"),
    ("prehme_sa_cfg", SAMM, (2,)),
    ("enable_me_hme_ref_pruning", "u1"), ("protect_closest_refs", "u1"),
    ("prune_ref_if_hme_sad_dev_bigger_than_th", "<u2"), ("prune_ref_if_me_sad_dev_bigger_than_th", "<u2"),
    ("enable_me_sr_adjustment", "u1"), ("distance_based_hme_resizing", "u1"),
    ("reduce_me_sr_based_on_mv_length_th", "<u2"), ("stationary_hme_sad_abs_th", "<u2"),
    ("stationary_me_sr_divisor", "<u2"), ("reduce_me_sr_based_on_hme_sad_abs_th", "<u2"),
    ("me_sr_divisor_for_low_hme_sad", "<u2"),
    ("reduce_hme_l0_sr_th_min", "u1"), ("reduce_hme_l0_sr_th_max", "u1"), ("stat_factor", "u1"),
    ("use_best_unipred_cand_only", "u1"), ("prune_me_candidates_th", "<i4"), ("me_early_exit_th", "<u4"),
])
assert CONTROLS.itemsize == 84

PIC_DUMP = np.dtype([
    ("ctrl", CONTROLS), ("enable_me_8x8", "u1"), ("enable_me_16x16", "u1"), ("max_cand", "u1"),
    ("max_refs", "u1"), ("max_l0", "u1"), ("input_resolution", "u1"), ("pad", "u1", (2,)), ("n_b64", "<u4"),
])

B64_DUMP = np.dtype([
    ("best_sad", "<u4", (2, 4, 85)), ("best_mv", "<u4", (2, 4, 85)),
    ("hme_sc_x", "<i2", (2, 4)), ("hme_sc_y", "<i2", (2, 4)), ("hme_sad", "<u8", (2, 4)),
    ("do_ref", "u1", (2, 4)), ("zz_sad", "<u4", (2, 4)), ("sr_divisor", "<u4", (2, 4)),
    ("l0_x", "<i2", (2, 4, 2, 2)), ("l0_y", "<i2", (2, 4, 2, 2)), ("l0_sad", "<u8", (2, 4, 2, 2)),
    ("l1_x", "<i2", (2, 4, 2, 2)), ("l1_y", "<i2", (2, 4, 2, 2)), ("l1_sad", "<u8", (2, 4, 2, 2)),
    ("l2_x", "<i2", (2, 4, 2, 2)), ("l2_y", "<i2", (2, 4, 2, 2)), ("l2_sad", "<u8", (2, 4, 2, 2)),
    ("pre_x", "<i2", (2, 4, 2)), ("pre_y", "<i2", (2, 4, 2)), ("pre_sad", "<u8", (2, 4, 2)),
    ("me_distortion", "<u4", (85,)), ("total_me_candidate_index", "u1", (85,)),
    ("me_mv_array", "<u4", (85 * 7,)), ("me_candidate_array", "u1", (85 * 23,)),
    ("rc_me_distortion", "<u4"), ("me_64x64_distortion", "<u4"), ("me_32x32_distortion", "<u4"),
    ("me_16x16_distortion", "<u4"), ("me_8x8_distortion", "<u4"), ("me_8x8_cost_variance", "<u4"),
    ("stationary_block_present_sb", "u1"), ("rc_me_allow_gm", "u1"), ("tf_me_exit", "u1"),
])

TPL_MAX_COSTS = 24
TPL_PIC_DUMP = np.dtype([
    ("tpl_opt_flag", "u1"), ("use_pred_sad_in_intra_search", "u1"), ("use_pred_sad_in_inter_search", "u1"),
    ("dispenser_search_level", "u1"), ("subsample_tx", "u1"), ("pf_shape", "u1"), ("disable_intra_pred", "u1"),
    ("synth_blk_size", "u1"), ("qindex", "<u4"), ("recon_hash", "<u8"),
    ("y_round_fp", "<i2", (2,)), ("y_quant_fp", "<i2", (2,)), ("y_dequant", "<i2", (2,)), ("pad", "u1", (4,)),
])
TPL_BLK_DUMP = np.dtype([
    ("valid", "u1"), ("best_mode", "u1"), ("best_intra_mode", "u1"), ("n_costs", "u1"), ("best_rf_idx", "<i4"),
    ("mv_row", "<i2"), ("mv_col", "<i2"), ("ref_frame_poc", "<u8"), ("src_srcrf_dist", "<i8"), ("src_srcrf_rate", "<i8"),
    ("costs", "<i4", (TPL_MAX_COSTS,)),
    ("srcrf_dist", "<i8"), ("recrf_dist", "<i8"), ("srcrf_rate", "<i8"), ("recrf_rate", "<i8"),
    ("stats_mv_row", "<i2"), ("stats_mv_col", "<i2"), ("stats_ref_frame_poc", "<u8"),
])

_HDR = "<8sIIIIIIIIIII"
_PIC = "<IIQiBBBBBBBB8I8QBBHBBBB"
_OUT = "<8sIIIIIIdQ"


def write_job(path, frames, pics, enc_mode, *, mode=MODE_DUMP, threads=1, use_avx2=0, reps=1,
              frame_rate=30 << 16, sc_class1=0, gm_enabled=0, gm_dist_th=0, ref_counts=(4, 3), tf=None, first_pass=None,
              restricted_motion_vector=0, number_scale=1, tpl=(0, 0), tiles=(0, 0), b64_step=0):
    """frames: uint8 [n, h, w]; pics: list of svtme.gop.Picture (refs are POCs = frame indices).
    tf = (hme_me_level, me_exit_th) runs every picture as a temporal-filter search (ME_MCTF) of its first list-0
    reference: the controls then come from tf_signal_derivation_me_kernel_oq and enc_mode is not used.
    first_pass = (bypass_zz_check, reduce_me_search) runs every picture as first-pass ME (ME_FIRST_PASS): controls from
    first_pass_signal_derivation_me_kernel; enc_mode only selects the HME level flags.
    number_scale: picture numbers are poc * number_scale (frames stay indexed by poc): stretches the picture distances.
    tpl = (tpl_level, qp): MODE_TPL runs the TPL dispenser with these after every picture's ME.
    tiles = (w, h): uniform tiles of w x h 64x64 blocks (0 = one tile in that direction), at most 15 each; they bound the
    windows of temporal-filter and first-pass searches when restricted_motion_vector is set.
    b64_step: n > 1 searches only the 64x64 blocks whose column and row index are multiples of n (the first pass's
    bypass_blk_step); the records of the others stay zero."""
    variant = (1, tf[0], tf[1]) if tf else ((4, (first_pass[0] & 1) | ((first_pass[1] & 1) << 1), 0) if first_pass else (0, 0, 0))
    n, h, w = frames.shape
    with open(path, "wb") as f:
        f.write(struct.pack(_HDR, b"SVTMEJ01", w, h, n, len(pics), mode, threads, use_avx2, reps,
                            ref_counts[0], ref_counts[1], b64_step))
        f.write(np.ascontiguousarray(frames, dtype=np.uint8).tobytes())
        for p in pics:
            rf = [[0] * 4, [0] * 4]
            for l in range(2):
                for i, r in enumerate(p.refs[l][:4]):
                    rf[l][i] = r
            n_list = 2 if p.refs[1] else 1
            f.write(struct.pack(_PIC, p.poc, frame_rate, p.poc * number_scale, enc_mode, p.temporal_layer, p.is_ref, n_list,
                                len(p.refs[0]), len(p.refs[1]), sc_class1, gm_enabled, gm_dist_th,
                                *rf[0], *rf[1], *[r * number_scale for r in rf[0] + rf[1]], *variant, restricted_motion_vector,
                                tpl[0], tpl[1], (tiles[0] & 15) | ((tiles[1] & 15) << 4)))


def read_dump(path):
    """Returns (header dict, list of (pic_dump record, b64 record array)); MODE_TPL: list of (pic_dump, b64 records,
    tpl pic record, tpl block records [n_b64, 21])."""
    with open(path, "rb") as f:
        data = f.read()
    hs = struct.calcsize(_OUT)
    magic, n_pics, n_b64, pds, bds, avx2, mode, secs, npic = struct.unpack_from(_OUT, data, 0)
    assert magic == b"SVTMEO01", magic
    hdr = dict(n_pics=n_pics, n_b64=n_b64, used_avx2=avx2, mode=mode, bench_seconds=secs, bench_pictures=npic)
    out = []
    if mode in (MODE_DUMP, MODE_DUMP_NOTAIL, MODE_TAIL, MODE_SHIM, MODE_TPL):
        assert pds == PIC_DUMP.itemsize and bds == B64_DUMP.itemsize, (pds, PIC_DUMP.itemsize, bds, B64_DUMP.itemsize)
        off = hs
        for _ in range(n_pics):
            pd = np.frombuffer(data, PIC_DUMP, 1, off)[0]
            off += pds
            bd = np.frombuffer(data, B64_DUMP, n_b64, off)
            off += bds * n_b64
            if mode == MODE_TPL:
                tp = np.frombuffer(data, TPL_PIC_DUMP, 1, off)[0]
                off += TPL_PIC_DUMP.itemsize
                tb = np.frombuffer(data, TPL_BLK_DUMP, n_b64 * 21, off).reshape(n_b64, 21)
                off += TPL_BLK_DUMP.itemsize * n_b64 * 21
                out.append((pd, bd, tp, tb))
                continue
            out.append((pd, bd))
    elif mode == MODE_PLANES:
        off = hs
        for _ in range(n_pics):
            planes = []
            for _k in range(3):
                stride, w, h, pad = struct.unpack_from("<4I", data, off)
                off += 16
                size = stride * (h + 2 * pad)
                planes.append(np.frombuffer(data, np.uint8, size, off).reshape(h + 2 * pad, stride))
                off += size
            out.append(planes)
    return hdr, out


def run_binary(binary, frames, pics, enc_mode, extra_args=(), **kw):
    if not os.path.exists(binary):
        raise FileNotFoundError(binary)
    with tempfile.TemporaryDirectory(prefix="svtme_") as td:
        job, out = os.path.join(td, "job.bin"), os.path.join(td, "out.bin")
        write_job(job, frames, pics, enc_mode, **kw)
        r = subprocess.run([binary, job, out, *extra_args], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("%s failed (%d): %s" % (binary, r.returncode, r.stderr[-2000:]))
        hdr, recs = read_dump(out)
        hdr["stdout"] = r.stdout
        return hdr, recs


def run_oracle_tail(frames, pics, enc_mode, handover, **kw):
    """The host tail (me_prune_ref .. perform_gm_detection) of the restatement on handed-over search results.
    handover: per picture, a B64_DUMP record array of which best_sad, best_mv, hme_sc_x/y, hme_sad, do_ref, zz_sad and
    sr_divisor are read. Returns what run_oracle returns."""
    with tempfile.TemporaryDirectory(prefix="svtme_") as td:
        job, out, hand = os.path.join(td, "job.bin"), os.path.join(td, "out.bin"), os.path.join(td, "hand.bin")
        write_job(job, frames, pics, enc_mode, mode=MODE_TAIL, **kw)
        with open(hand, "wb") as f:
            for rec in handover:
                f.write(np.ascontiguousarray(rec, dtype=B64_DUMP).tobytes())
        r = subprocess.run([ORACLE_BIN, job, out, hand], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("oracle tail failed (%d): %s" % (r.returncode, r.stderr[-2000:]))
        return read_dump(out)


def run_reference(frames, pics, enc_mode, **kw):
    return run_binary(REF_BIN, frames, pics, enc_mode, **kw)


def run_reference_shim(frames, pics, enc_mode, library, **kw):
    """The reference-side binding end to end (ref_me's REF_MODE_SHIM): the searches run on the GPU engine in `library`
    (libsvtme.so) through its C ABI, each block's results are unpacked into a real MeContext and the reference's own
    me_prune_ref .. perform_gm_detection finish the block. Returns what run_reference returns."""
    return run_binary(REF_BIN, frames, pics, enc_mode, extra_args=(library,), mode=MODE_SHIM, **kw)


def run_reference_tpl(frames, pics, enc_mode, tpl_level, qp, library=None, **kw):
    """ref_me's MODE_TPL: every picture's ME, then the TPL dispenser on its results. With `library` (libsvtme.so) the
    dispenser's source-based search comes from the GPU engine and only its reconstruction path runs on the host."""
    return run_binary(REF_BIN, frames, pics, enc_mode, extra_args=(library,) if library else (), mode=MODE_TPL,
                      tpl=(tpl_level, qp), **kw)


def reference_dct(blocks):
    """The reference's forward transform of the TPL dispenser (svt_av1_wht_fwd_txfm) on int16 residual blocks [n, h, w] with
    (w, h) one of (16, 16), (16, 8), (16, 4), (32, 8): int32 [3, n, h, w] for DEFAULT_SHAPE, N2_SHAPE, N4_SHAPE."""
    blocks = np.ascontiguousarray(blocks, np.int16)
    n, h, w = blocks.shape
    with tempfile.TemporaryDirectory(prefix="svtme_") as td:
        a, b = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        blocks.tofile(a)
        subprocess.run([REF_BIN, "dct", str(w), str(h), a, b], check=True, timeout=120)
        return np.fromfile(b, np.int32).reshape(3, n, h, w)


def run_oracle(frames, pics, enc_mode, **kw):
    return run_binary(ORACLE_BIN, frames, pics, enc_mode, **kw)


def have_reference():
    return os.path.exists(REF_BIN)
