"""TEST INFRASTRUCTURE: jobs for oracle/_ref/ref_gm (the real reference's global-motion estimation, oracle/gm_driver.c), its
call records, and ctypes access to the CPU restatement (oracle/gm_oracle.c). Not part of the product."""
import ctypes
import os
import struct
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_GM = os.path.join(HERE, "_ref", "ref_gm")
ORACLE_LIB = os.path.join(HERE, "build", "libme_oracle.so")
MAX_CORNERS = 4096

REC_CORNERS, REC_CORRESPONDENCES, REC_WARP_ERROR, REC_FRAME_ERROR, REC_REFINE, REC_FINAL = 1, 2, 3, 4, 5, 9
WARP_FIELDS = ("m0", "m1", "m2", "m3", "m4", "m5", "alpha", "beta", "gamma", "delta", "wmtype", "best_error", "error")
# one svt_av1_refine_integerized_param call: the model that goes in (with the struct's shear and type), the type to force, the
# steps, the error to beat; the model that comes out and the function's result
REFINE_FIELDS = ("i0", "i1", "i2", "i3", "i4", "i5", "i_alpha", "i_beta", "i_gamma", "i_delta", "i_wmtype", "force_wmtype", "steps",
                 "best_frame_error", "o0", "o1", "o2", "o3", "o4", "o5", "o_alpha", "o_beta", "o_gamma", "o_delta", "o_wmtype", "error")
FINAL_FIELDS = ("wmtype", "m0", "m1", "m2", "m3", "m4", "m5", "m6", "m7", "alpha", "beta", "gamma", "delta", "invalid")


def have_reference():
    return os.path.exists(REF_GM)


def write_job(path, frames, jobs, mode=0):
    """frames: uint8 [n, h, w]; jobs: dicts with cur, ref, level, rotzoom_only, steps, allow_hp, and optionally direct."""
    n, h, w = frames.shape
    with open(path, "wb") as f:
        f.write(struct.pack("<8sIIIII", b"SVTGMJ02", w, h, n, len(jobs), mode))
        f.write(np.ascontiguousarray(frames, np.uint8).tobytes())
        for j in jobs:
            f.write(struct.pack("<6I", j["cur"], j["ref"], j.get("level", 0), j.get("rotzoom_only", 1), j.get("steps", 5), j.get("allow_hp", 0)))
            d = j.get("direct")  # (mat [6], wmtype, best_frame_error): one refinement call instead of the estimation
            f.write(struct.pack("<I6iiq", 1, *d[0], d[1], d[2]) if d else struct.pack("<I6iiq", 0, 0, 0, 0, 0, 0, 0, 0, 0))


def read_records(path, n_jobs):
    """-> one dict per job: corners_frm, corners_ref (int [n, 2]), correspondences (int [n, 4], the first call's),
    calls (list of ('corners' | 'corr' | 'warp' | 'frame' | 'refine', payload) in call order; a 'refine' entry follows the
    'warp' entries of the errors its loop took), final (dict)."""
    with open(path, "rb") as f:
        raw = f.read()
    assert raw[:8] == b"SVTGMO01", "bad record magic"
    words = np.frombuffer(raw[8:], np.int64)
    out, cur, i = [], None, 0
    while i < len(words):
        kind, n = int(words[i]), int(words[i + 1])
        p = words[i + 2:i + 2 + n]
        i += 2 + n
        if cur is None:
            cur = {"calls": []}
        if kind == REC_CORNERS:
            pts = p[2:2 + 2 * int(p[1])].reshape(-1, 2).copy()
            cur.setdefault("corners_ref" if p[0] else "corners_frm", pts)
            cur["calls"].append(("corners", (int(p[0]), pts)))
        elif kind == REC_CORRESPONDENCES:
            c = p[1:1 + 4 * int(p[0])].reshape(-1, 4).copy()
            cur.setdefault("correspondences", c)
            cur["calls"].append(("corr", c))
        elif kind == REC_WARP_ERROR:
            cur["calls"].append(("warp", dict(zip(WARP_FIELDS, (int(v) for v in p)))))
        elif kind == REC_FRAME_ERROR:
            cur["calls"].append(("frame", int(p[0])))
        elif kind == REC_REFINE:
            cur["calls"].append(("refine", dict(zip(REFINE_FIELDS, (int(v) for v in p)))))
        elif kind == REC_FINAL:
            cur["final"] = dict(zip(FINAL_FIELDS, (int(v) for v in p)))
            out.append(cur)
            cur = None
        else:
            raise ValueError("unknown record kind %d" % kind)
    assert cur is None and len(out) == n_jobs, "truncated record file"
    return out


def run_reference(frames, jobs, library=None, timeout=1200, use_avx2=False, engine_refine=False):
    """Runs compute_global_motion of the real reference per job; with `library` (path of libsvtme.so) its four data-parallel
    functions go to the GPU engine instead (the reference-side binding end to end), with engine_refine its refinement loop
    as a whole too (svtme_gm_refine). use_avx2: the reference's AVX2 kernels instead of its C ones for the warp filter and
    the frame error."""
    with tempfile.TemporaryDirectory() as d:
        jp, op = os.path.join(d, "job.bin"), os.path.join(d, "out.bin")
        write_job(jp, frames, jobs, mode=(1 if library else 0) | (2 if use_avx2 else 0) | (4 if library and engine_refine else 0))
        cmd = [REF_GM, jp, op] + ([library] if library else [])
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
        if r.returncode != 0:
            raise RuntimeError("ref_gm failed (%d): %s" % (r.returncode, r.stderr[-2000:]))
        return read_records(op, len(jobs))


def reference_tables():
    """(warped_filter int16 [193, 8], error_lut int32 [512]) of the reference build, taken from its binary at run time; they
    belong to the AV1 specification and are not kept in this repository."""
    with tempfile.TemporaryDirectory() as d:
        op = os.path.join(d, "tables.bin")
        subprocess.run([REF_GM, "tables", op], check=True, timeout=60)
        raw = open(op, "rb").read()
    wf = np.frombuffer(raw[:193 * 8 * 2], np.int16).reshape(193, 8).copy()
    lut = np.frombuffer(raw[193 * 8 * 2:], np.int32).copy()
    assert lut.shape == (512,)
    return wf, lut


# ------------------------------------------------------------------------------------------------ the CPU restatement
class _Tables(ctypes.Structure):
    _fields_ = [("warped_filter", ctypes.POINTER(ctypes.c_int16)), ("error_lut", ctypes.POINTER(ctypes.c_int32))]


class _Model(ctypes.Structure):
    _fields_ = [("wmmat", ctypes.c_int32 * 6), ("alpha", ctypes.c_int16), ("beta", ctypes.c_int16), ("gamma", ctypes.c_int16),
                ("delta", ctypes.c_int16), ("wmtype", ctypes.c_int32)]


class _Warp(ctypes.Structure):
    _fields_ = [("wmmat", ctypes.c_int32 * 6), ("alpha", ctypes.c_int16), ("beta", ctypes.c_int16), ("gamma", ctypes.c_int16),
                ("delta", ctypes.c_int16)]


_lib = None


def _oracle():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(ORACLE_LIB)
        _lib.orc_gm_warp_error.restype = ctypes.c_int64
        _lib.orc_gm_frame_error.restype = ctypes.c_int64
        _lib.orc_gm_refine.restype = ctypes.c_int64
    return _lib


def _u8(a):
    a = np.ascontiguousarray(a, np.uint8)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def oracle_corners(plane, max_points=MAX_CORNERS):
    a, pa = _u8(plane)
    pts = np.zeros((max_points, 2), np.int32)
    n = _oracle().orc_gm_corners(pa, a.shape[1], a.shape[0], a.shape[1], pts.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), max_points)
    assert n >= 0
    return pts[:n].copy()


def oracle_correspondences(frm, frm_corners, ref, ref_corners):
    a, pa = _u8(frm)
    b, pb = _u8(ref)
    fc = np.ascontiguousarray(frm_corners, np.int32)
    rc = np.ascontiguousarray(ref_corners, np.int32)
    out = np.zeros((max(len(fc), 1), 4), np.int32)
    i32 = ctypes.POINTER(ctypes.c_int32)
    n = _oracle().orc_gm_correspondences(pa, a.shape[1], fc.ctypes.data_as(i32), len(fc), pb, b.shape[1], rc.ctypes.data_as(i32), len(rc),
                                         a.shape[1], a.shape[0], out.ctypes.data_as(i32))
    return out[:n].copy()


def _tables(tables):
    wf = np.ascontiguousarray(tables[0], np.int16)
    lut = np.ascontiguousarray(tables[1], np.int32)
    t = _Tables(wf.ctypes.data_as(ctypes.POINTER(ctypes.c_int16)), lut.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    return t, (wf, lut)


def oracle_warp_error(tables, mat, shear, ref, frm, best_error):
    t, keep = _tables(tables)
    a, pa = _u8(ref)
    b, pb = _u8(frm)
    w = _Warp((ctypes.c_int32 * 6)(*[int(v) for v in mat]), *[int(v) for v in shear])
    return int(_oracle().orc_gm_warp_error(ctypes.byref(t), ctypes.byref(w), pa, a.shape[1], a.shape[0], a.shape[1], pb, b.shape[1],
                                           b.shape[0], b.shape[1], ctypes.c_int64(int(best_error))))


def oracle_frame_error(tables, ref, frm):
    t, keep = _tables(tables)
    a, pa = _u8(ref)
    b, pb = _u8(frm)
    return int(_oracle().orc_gm_frame_error(ctypes.byref(t), pa, a.shape[1], pb, b.shape[1], b.shape[0], b.shape[1]))


def oracle_refine(tables, mat, shear, model_wmtype, wmtype, ref, frm, steps, best_frame_error):
    """-> (mat [6], shear [4], wmtype, error, evaluations) of the restated svt_av1_refine_integerized_param."""
    t, keep = _tables(tables)
    a, pa = _u8(ref)
    b, pb = _u8(frm)
    m = _Model((ctypes.c_int32 * 6)(*[int(v) for v in mat]), *[int(v) for v in shear], int(model_wmtype))
    n = ctypes.c_int(0)
    err = int(_oracle().orc_gm_refine(ctypes.byref(t), ctypes.byref(m), int(wmtype), pa, a.shape[1], pb, b.shape[1], b.shape[0], b.shape[1],
                                      int(steps), ctypes.c_int64(int(best_frame_error)), ctypes.byref(n)))
    return list(m.wmmat), [m.alpha, m.beta, m.gamma, m.delta], int(m.wmtype), err, n.value


def plane_of(frame, level):
    """The plane global motion reads at a downsample level: the full picture, or the 2x2-box decimations the store builds
    (downsample_2d_c, Source/Lib/Encoder/Codec/EbPictureAnalysisProcess.c:187; applied twice for the sixteenth plane)."""
    p = np.asarray(frame, np.uint8)
    for _ in range(level):
        q = p.astype(np.uint32)
        p = ((q[0::2, 0::2] + q[0::2, 1::2] + q[1::2, 0::2] + q[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    return p
