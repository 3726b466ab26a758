"""Shared helpers of the test-suite: run the same synthetic job through the oracle (CPU restatement) and through
the HIP engine's C ABI, and compare every field the engine hands to the encoder's host tail."""
import numpy as np

import refio
from svtme import gop, synth


def make_job(width, height, n_frames, enc_mode, seed=1, structure="ra", kind="synth", motion=(3, -2)):
    if kind == "synth":
        frames = synth.synth_sequence(width, height, n_frames, seed=seed, motion=motion)
    else:
        frames = synth.extreme_sequence(width, height, n_frames, kind, seed=seed)
    if structure == "ra":
        pics = gop.random_access(n_frames, enc_mode)
    elif structure == "ldp":
        pics = gop.low_delay(n_frames, enc_mode, n_list=1)
    else:
        pics = gop.low_delay(n_frames, enc_mode, n_list=2)
    b0, b1, n0, n1 = gop.mrp_counts(enc_mode)
    return frames, pics, (max(b0, n0), max(b1, n1))


def oracle_handover(frames, pics, enc_mode, ref_counts, threads=8, **kw):
    """Oracle state right after integer_search_b64 for every picture: list of b64 record arrays."""
    _, recs = refio.run_oracle(frames, pics, enc_mode, mode=refio.MODE_DUMP_NOTAIL, threads=threads,
                               ref_counts=ref_counts, **kw)
    return [bd for _, bd in recs]


def compare_picture(out, bd, pic, where=""):
    """Bit-exact comparison of one picture's engine output with the oracle records. Returns a list of messages."""
    msgs = []
    n_ref = [len(pic.refs[0]), len(pic.refs[1])]

    def chk(name, a, b, mask=None):
        a, b = np.asarray(a), np.asarray(b)
        ne = a != b
        if mask is not None:
            ne &= mask
        if ne.any():
            idx = tuple(np.argwhere(ne)[0])
            msgs.append("%s %s: %d mismatches, first at %s: engine %s oracle %s" %
                        (where, name, int(ne.sum()), idx, a[idx], b[idx]))

    sr = out["search_results"]
    chk("zz_sad", out["zz_sad"], bd["zz_sad"])
    chk("sr_divisor", out["sr_divisor"], bd["sr_divisor"])
    chk("hme_sc_x", sr["hme_sc_x"], bd["hme_sc_x"])
    chk("hme_sc_y", sr["hme_sc_y"], bd["hme_sc_y"])
    chk("hme_sad", sr["hme_sad"], bd["hme_sad"])
    chk("do_ref", sr["do_ref"], bd["do_ref"])
    have = np.zeros((1, 2, 4, 1), bool)
    for l in range(2):
        have[0, l, :n_ref[l], 0] = True
    mask = have & (bd["do_ref"][:, :, :, None] == 1)
    mask = np.broadcast_to(mask, out["best_sad"].shape)
    chk("best_sad", out["best_sad"], bd["best_sad"], mask)
    chk("best_mv", out["best_mv"], bd["best_mv"], mask)
    return msgs


def run_engine_job(frames, pics, enc_mode, sc_class1=0, frame_rate=30 << 16, engine=None):
    """Runs every picture of the job through the engine's C ABI; returns the list of output dicts."""
    import svtme
    n, h, w = frames.shape
    own = engine is None
    eng = engine or svtme.Engine(w, h, store_slots=n + 1, max_in_flight=2)
    try:
        for i in range(n):
            eng.store_put(i, i, frames[i])
        ctrl = svtme.derive_controls(enc_mode, w, h, sc_class1, frame_rate)
        outs = []
        for p in pics:
            pic = eng.make_picture(w, h, p.poc, p.temporal_layer, p.is_ref, p.refs, ctrl)
            outs.append(eng.run_picture(pic))
        return outs
    finally:
        if own:
            eng.close()


def engine_vs_oracle(width, height, n_frames, enc_mode, seed=1, structure="ra", kind="synth", sc_class1=0,
                     frame_rate=30 << 16, max_msgs=12):
    frames, pics, rc = make_job(width, height, n_frames, enc_mode, seed, structure, kind)
    recs = oracle_handover(frames, pics, enc_mode, rc, sc_class1=sc_class1, frame_rate=frame_rate)
    outs = run_engine_job(frames, pics, enc_mode, sc_class1, frame_rate)
    msgs = []
    for p, out, bd in zip(pics, outs, recs):
        msgs += compare_picture(out, bd, p, "poc %d (layer %d, refs %s)" % (p.poc, p.temporal_layer, p.refs))
        if len(msgs) >= max_msgs:
            break
    return msgs
