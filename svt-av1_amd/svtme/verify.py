"""One digest per job over what the ME stage hands on, computed the same way from the engine's outputs and from a dump
of the reference encoder's ME: lets a benchmark check, outside its timed region, that the pictures it timed came out
bit-exact with the reference without shipping the reference's arrays (a committed digest per workload is enough).

Open-loop pictures: zero-MV SADs, search-range divisors, search centres and the best SAD / MV arrays of the references
whose integer search ran (the reference keeps stale scratch for pruned references, so those are masked). Temporal-filter
pairs: minima, vectors, centre, HME SAD and the exit flag of list 0 / reference 0. First-pass pictures: centres and the
arrays of the list-0 references.
"""
import hashlib

import numpy as np

MAX_SAD_VALUE = 128 * 128 * 255


def _update(h, arrays):
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode() + b"|" + repr(a.shape).encode() + b"|")
        h.update(a.tobytes())


def _have(n_ref):
    have = np.zeros((1, 2, 4), bool)
    for l in range(2):
        have[0, l, :n_ref[l]] = True
    return have


def engine_fields(out, pic, kind):
    """The digested arrays of one picture from an engine output dict (svtme.results_to_arrays)."""
    sr = out["search_results"]
    n_ref = [len(pic.refs[0]), len(pic.refs[1])]
    if kind == "tf":
        return [out["best_sad"][:, 0, 0], out["best_mv"][:, 0, 0], sr["hme_sc_x"][:, 0, 0].astype(np.int16),
                sr["hme_sc_y"][:, 0, 0].astype(np.int16), sr["hme_sad"][:, 0, 0].astype(np.uint64),
                sr["tf_me_exit"][:, 0, 0].astype(np.uint8)]
    if kind == "fp":
        nr = n_ref[0]
        return [sr["hme_sc_x"].astype(np.int16), sr["hme_sc_y"].astype(np.int16), out["best_sad"][:, 0, :nr].astype(np.uint32),
                out["best_mv"][:, 0, :nr].astype(np.uint32)]
    m = (_have(n_ref) & (sr["do_ref"] == 1))[:, :, :, None]
    return [out["zz_sad"].astype(np.uint32), out["sr_divisor"].astype(np.uint32), sr["hme_sc_x"].astype(np.int16),
            sr["hme_sc_y"].astype(np.int16), np.where(m, out["best_sad"], 0).astype(np.uint32),
            np.where(m, out["best_mv"], 0).astype(np.uint32)]


def reference_fields(pd, bd, pic, kind):
    """The same arrays from a final dump of the reference (or of the oracle): picture record pd, block records bd."""
    n_ref = [len(pic.refs[0]), len(pic.refs[1])]
    if kind == "tf":
        return [bd["best_sad"][:, 0, 0], bd["best_mv"][:, 0, 0], bd["hme_sc_x"][:, 0, 0].astype(np.int16),
                bd["hme_sc_y"][:, 0, 0].astype(np.int16), bd["hme_sad"][:, 0, 0].astype(np.uint64), bd["tf_me_exit"].astype(np.uint8)]
    if kind == "fp":
        nr = n_ref[0]
        return [bd["hme_sc_x"].astype(np.int16), bd["hme_sc_y"].astype(np.int16), bd["best_sad"][:, 0, :nr].astype(np.uint32),
                bd["best_mv"][:, 0, :nr].astype(np.uint32)]
    # me_prune_ref marks the references HME had pruned with hme_sad = MAX_SAD_VALUE * 64 (EbMotionEstimation.c:1583-1588)
    m = np.broadcast_to(_have(n_ref), bd["hme_sad"].shape).copy()
    if bool(pd["ctrl"]["enable_me_hme_ref_pruning"]):
        m &= bd["hme_sad"] != MAX_SAD_VALUE * 64
    m = m[:, :, :, None]
    return [bd["zz_sad"].astype(np.uint32), bd["sr_divisor"].astype(np.uint32), bd["hme_sc_x"].astype(np.int16),
            bd["hme_sc_y"].astype(np.int16), np.where(m, bd["best_sad"], 0).astype(np.uint32),
            np.where(m, bd["best_mv"], 0).astype(np.uint32)]


def kind_of(workload):
    return "tf" if workload.get("tf") else ("fp" if workload.get("fp") else "me")


def _lattice(n_b64, b64_w, step):
    """Indices of the 64x64 blocks a search with svtme_picture.b64_step = step covers (the others' results are undefined)."""
    idx = np.arange(n_b64)
    if step and step > 1:
        idx = idx[((idx % b64_w) % step == 0) & ((idx // b64_w) % step == 0)]
    return idx


def engine_digest(outs, pics, kind, b64_w=0, b64_step=0):
    h = hashlib.sha256()
    for out, pic in zip(outs, pics):
        f = engine_fields(out, pic, kind)
        sel = _lattice(len(f[0]), b64_w, b64_step)
        _update(h, [a[sel] for a in f])
    return h.hexdigest()


def reference_digest(recs, pics, kind, b64_w=0, b64_step=0):
    h = hashlib.sha256()
    for (pd, bd), pic in zip(recs, pics):
        f = reference_fields(pd, bd, pic, kind)
        sel = _lattice(len(f[0]), b64_w, b64_step)
        _update(h, [a[sel] for a in f])
    return h.hexdigest()
