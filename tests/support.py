"""Shared helpers of the test-suite: run the same synthetic job through the oracle (CPU restatement) and through
the HIP engine's C ABI, and compare every field the engine hands to the encoder's host tail."""
import numpy as np

import refio
from svtme import gop, synth

# fields of a b64 record that do not depend on which ME thread processed the previous b64 (the reference keeps
# per-thread scratch between b64s; see compare_records)
HANDOVER_FIELDS = ["zz_sad", "sr_divisor", "hme_sc_x", "hme_sc_y"]
TAIL_FIELDS = ["do_ref", "hme_sad", "me_distortion", "total_me_candidate_index", "me_mv_array", "me_candidate_array",
               "rc_me_distortion", "me_64x64_distortion", "me_32x32_distortion", "me_16x16_distortion",
               "me_8x8_distortion", "me_8x8_cost_variance", "stationary_block_present_sb", "rc_me_allow_gm"]


def make_job(width, height, n_frames, enc_mode, seed=1, structure="ra", kind="synth", motion=(3, -2), bit_depth=8):
    if kind in ("synth", "half_static"):
        frames = synth.synth_sequence(width, height, n_frames, seed=seed, motion=motion, bit_depth=bit_depth)
        if kind == "half_static":  # the left half never changes: perfect zero-vector matches there
            frames[:, :, :width // 2] = frames[0, :, :width // 2]
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


def make_tf_job(width, height, n_frames, seed=1, kind="synth", motion=(3, -2)):
    """A temporal-filter window: the centre frame searched in every other frame of the window, one (centre, neighbour)
    pair per picture as the temporal filter's block loop does (EbTemporalFiltering.c:4844-4873). kind "half_static"
    freezes the left half of every frame so that those blocks take the tf_me_exit_th exit."""
    if kind in ("synth", "half_static"):
        frames = synth.synth_sequence(width, height, n_frames, seed=seed, motion=motion)
        if kind == "half_static":
            frames[:, :, :width // 2] = frames[0, :, :width // 2]
    else:
        frames = synth.extreme_sequence(width, height, n_frames, kind, seed=seed)
    c = n_frames // 2
    pics = [gop.Picture(poc=c, temporal_layer=0, is_ref=1, refs=[[j], []]) for j in range(n_frames) if j != c]
    return frames, pics


def make_first_pass_job(width, height, n_frames, seed=1, kind="synth", motion=(3, -2)):
    """First-pass ME of a two-pass encode: every frame after the first searches the previous frame and, from the third
    frame on, the first ("golden") frame as well: one list, base layer, used as reference (firstpass.c:1166-1186)."""
    if kind in ("synth", "half_static"):
        frames = synth.synth_sequence(width, height, n_frames, seed=seed, motion=motion)
        if kind == "half_static":
            frames[:, :, :width // 2] = frames[0, :, :width // 2]
    else:
        frames = synth.extreme_sequence(width, height, n_frames, kind, seed=seed)
    pics = [gop.Picture(poc=i, temporal_layer=0, is_ref=1, refs=[[i - 1] + ([0] if i >= 2 else []), []])
            for i in range(1, n_frames)]
    return frames, pics


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


def run_engine_job(frames, pics, enc_mode, sc_class1=0, frame_rate=30 << 16, engine=None, tail=None, serial_hme=0,
                   restricted=0, number_scale=1, tiles=None, batch=1, edit_controls=None, repeat=1):
    """Runs every picture of the job through the engine's C ABI; returns the list of output dicts. tail: an
    svtme.Tail to run the host tail on the device as well. batch > 1: the pictures go through svtme_pictures_submit,
    `batch` at a time (one set of launches for all of them, one search stream), as the benchmark submits them.
    edit_controls: called with the derived controls before the pictures are described. repeat: the job's pictures are
    submitted this many times over, one copy after the other (batches larger than the job); the outputs come in that order."""
    import svtme
    n, h, w = frames.shape
    own = engine is None
    eng = engine or svtme.Engine(w, h, store_slots=n + 1, max_in_flight=max(2, batch), serial_hme=serial_hme)
    try:
        for i in range(n):
            eng.store_put(i, i, frames[i])
        set_uniform_tiles(eng, w, h, tiles)
        ctrl = svtme.derive_controls(enc_mode, w, h, sc_class1, frame_rate)
        if edit_controls:
            edit_controls(ctrl)  # settings no preset has
        outs = []
        descs = [eng.make_picture(w, h, p.poc, p.temporal_layer, p.is_ref, p.refs, ctrl, tail=tail,
                                  restricted_motion_vector=restricted, number_of=lambda poc: poc * number_scale) for p in pics]
        descs = descs * repeat
        if batch > 1:
            for i in range(0, len(descs), batch):
                tickets = eng.submit_batch(descs[i:i + batch]) if len(descs[i:i + batch]) > 1 else [eng.submit(descs[i])]
                for t in tickets:
                    outs.append(eng.wait(t))
                    eng.done(t)
            return outs
        for pic in descs:
            outs.append(eng.run_picture(pic))
        return outs
    finally:
        if own:
            eng.close()


def set_uniform_tiles(eng, w, h, tiles):
    """tiles = (width, height) of the sequence's tiles in 64x64 blocks (0 = one tile that way), None / (0, 0): none."""
    import svtme
    if tiles and (tiles[0] or tiles[1]):
        eng.set_tile_rects(w, h, svtme.uniform_tile_rects(w, h, *tiles))


def run_engine_tf_job(frames, pics, level, exit_th, engine=None, serial_hme=0, restricted=0, tiles=None):
    """Temporal-filter searches (ME_MCTF) of a make_tf_job window through the engine's C ABI."""
    import svtme
    n, h, w = frames.shape
    own = engine is None
    eng = engine or svtme.Engine(w, h, store_slots=n + 1, max_in_flight=2, serial_hme=serial_hme)
    try:
        for i in range(n):
            eng.store_put(i, i, frames[i])
        set_uniform_tiles(eng, w, h, tiles)
        ctrl = svtme.derive_tf_controls(level)
        outs = []
        for p in pics:
            pic = eng.make_picture(w, h, p.poc, p.temporal_layer, p.is_ref, p.refs, ctrl, tf_me_exit_th=exit_th,
                                   restricted_motion_vector=restricted)
            outs.append(eng.run_picture(pic))
        return outs
    finally:
        if own:
            eng.close()


def run_engine_first_pass_job(frames, pics, bypass_zz, reduce_me, frame_rate=30 << 16, tail=None, restricted=0, tiles=None,
                              b64_step=0):
    """First-pass searches (ME_FIRST_PASS) of a make_first_pass_job sequence through the engine's C ABI."""
    import svtme
    n, h, w = frames.shape
    eng = svtme.Engine(w, h, store_slots=n + 1, max_in_flight=2)
    try:
        for i in range(n):
            eng.store_put(i, i, frames[i])
        set_uniform_tiles(eng, w, h, tiles)
        ctrl = svtme.derive_first_pass_controls(w, h, frame_rate, reduce_me)
        return [eng.run_picture(eng.make_picture(w, h, p.poc, p.temporal_layer, p.is_ref, p.refs, ctrl, tail=tail,
                                                 first_pass_bypass_zz=bypass_zz, restricted_motion_vector=restricted,
                                                 b64_step=b64_step))
                for p in pics]
    finally:
        eng.close()


def tf_fields(out):
    """What a temporal-filter search hands back, in the shape of make_golden.tf_canonical."""
    sr = out["search_results"]
    return {"best_sad": out["best_sad"][:, 0, 0], "best_mv": out["best_mv"][:, 0, 0], "hme_sc_x": sr["hme_sc_x"][:, 0, 0],
            "hme_sc_y": sr["hme_sc_y"][:, 0, 0], "hme_sad": sr["hme_sad"][:, 0, 0], "tf_me_exit": sr["tf_me_exit"][:, 0, 0]}


def engine_vs_oracle(width, height, n_frames, enc_mode, seed=1, structure="ra", kind="synth", sc_class1=0,
                     frame_rate=30 << 16, max_msgs=12, serial_hme=0, motion=(3, -2), restricted=0, number_scale=1):
    frames, pics, rc = make_job(width, height, n_frames, enc_mode, seed, structure, kind, motion)
    recs = oracle_handover(frames, pics, enc_mode, rc, sc_class1=sc_class1, frame_rate=frame_rate,
                           restricted_motion_vector=restricted, number_scale=number_scale)
    outs = run_engine_job(frames, pics, enc_mode, sc_class1, frame_rate, serial_hme=serial_hme, restricted=restricted,
                          number_scale=number_scale)
    msgs = []
    for p, out, bd in zip(pics, outs, recs):
        msgs += compare_picture(out, bd, p, "poc %d (layer %d, refs %s)" % (p.poc, p.temporal_layer, p.refs))
        if len(msgs) >= max_msgs:
            break
    return msgs


def random_case(seed):
    """A seeded random open-loop job description (keyword arguments of engine_vs_oracle)."""
    rng = np.random.default_rng(seed)
    kind = str(rng.choice(["synth", "synth", "synth", "synth", "half_static", "noise", "flat", "max"]))
    return dict(width=int(rng.integers(8, 57)) * 8, height=int(rng.integers(8, 41)) * 8, n_frames=int(rng.integers(3, 10)),
                enc_mode=int(rng.integers(-2, 14)), seed=int(seed), structure=str(rng.choice(["ra", "ra", "ldb", "ldp"])),
                kind=kind, sc_class1=int(rng.random() < 0.25), frame_rate=int(rng.choice([24, 30, 50, 60])) << 16,
                motion=(int(rng.integers(-14, 15)), int(rng.integers(-9, 10))), restricted=int(rng.random() < 0.2))


def _random_tiles(seed):
    """restricted_motion_vector and the sequence's tiles for half of the seeds (a generator of their own, so that a seed's
    other fields do not depend on them)."""
    rng = np.random.default_rng(seed + 991)
    if rng.random() < 0.5:
        return dict(restricted=0, tiles=(0, 0))
    return dict(restricted=1, tiles=(int(rng.integers(0, 4)), int(rng.integers(0, 3))))


def random_tf_case(seed):
    rng = np.random.default_rng(seed)
    return dict(width=int(rng.integers(8, 57)) * 8, height=int(rng.integers(8, 41)) * 8, n_frames=int(rng.integers(2, 6)),
                seed=int(seed), kind=str(rng.choice(["synth", "synth", "synth", "half_static", "noise", "flat"])),
                motion=(int(rng.integers(-14, 15)), int(rng.integers(-9, 10))), level=int(rng.integers(0, 3)),
                exit_th=int(rng.choice([0, 0, 256, 4096])), **_random_tiles(seed))


def random_first_pass_case(seed):
    rng = np.random.default_rng(seed)
    return dict(width=int(rng.integers(8, 57)) * 8, height=int(rng.integers(8, 41)) * 8, n_frames=int(rng.integers(2, 6)),
                seed=int(seed), kind=str(rng.choice(["synth", "synth", "synth", "half_static", "noise", "flat"])),
                motion=(int(rng.integers(-14, 15)), int(rng.integers(-9, 10))), bypass_zz=int(rng.integers(0, 2)),
                reduce_me=int(rng.integers(0, 2)), frame_rate=int(rng.choice([24, 30, 60])) << 16, **_random_tiles(seed))


STAT_FIELDS = ["rc_me_distortion", "me_64x64_distortion", "me_32x32_distortion", "me_16x16_distortion",
               "me_8x8_distortion", "me_8x8_cost_variance", "stationary_block_present_sb", "rc_me_allow_gm"]


def tail_of(pd0, gm_enabled=1):
    """The svtme.Tail matching a dump's picture record."""
    import svtme
    return svtme.Tail(enable=1, enable_me_8x8=int(pd0["enable_me_8x8"]), enable_me_16x16=int(pd0["enable_me_16x16"]),
                      max_refs=int(pd0["max_refs"]), max_cand=int(pd0["max_cand"]), max_l0=int(pd0["max_l0"]),
                      input_resolution=int(pd0["input_resolution"]), gm_enabled=gm_enabled,
                      gm_use_distance_based_active_th=gm_enabled)


def device_tail_mismatches(out, bd, where):
    """Device-side host tail of one picture against a final dump's records."""
    msgs = []
    for f in ("total_me_candidate_index", "me_mv_array", "me_candidate_array", "me_distortion"):
        if not np.array_equal(out[f], bd[f]):
            msgs.append("%s %s" % (where, f))
    for j, f in enumerate(STAT_FIELDS):
        if not np.array_equal(out["b64_stats"][:, j], bd[f].astype(np.uint32)):
            msgs.append("%s %s" % (where, f))
    if not np.array_equal(out["final_do_ref"], bd["do_ref"]):
        msgs.append("%s do_ref" % where)
    if not np.array_equal(out["final_hme_sad"], bd["hme_sad"]):
        msgs.append("%s hme_sad" % where)
    return msgs


MAX_SAD_VALUE = 128 * 128 * 255


def searched_mask(bd, pic, pruning_enabled):
    """[n_b64, 2, 4] bool: references whose integer search ran, reconstructed from a FINAL reference dump:
    me_prune_ref (EbMotionEstimation.c:1583-1588) marks the references HME had pruned with hme_sad = MAX_SAD_VALUE*64."""
    n_ref = [len(pic.refs[0]), len(pic.refs[1])]
    have = np.zeros((1, 2, 4), bool)
    for l in range(2):
        have[0, l, :n_ref[l]] = True
    m = np.broadcast_to(have, bd["hme_sad"].shape).copy()
    if pruning_enabled:
        m &= bd["hme_sad"] != MAX_SAD_VALUE * 64
    return m


def canonical(bd, pic, pruning_enabled):
    """The thread-independent content of a final dump: best SAD / MV only where the search ran."""
    m = searched_mask(bd, pic, pruning_enabled)[:, :, :, None]
    out = {f: np.ascontiguousarray(bd[f]) for f in HANDOVER_FIELDS + TAIL_FIELDS}
    out["best_sad"] = np.where(m, bd["best_sad"], 0).astype(np.uint32)
    out["best_mv"] = np.where(m, bd["best_mv"], 0).astype(np.uint32)
    return out


class OracleEngine:
    """Stand-in with svtme.Engine's interface whose searches run on the CPU oracle: lets the host-side scheduling
    logic (sharding, store residency, in-flight queue) run in tests without a GPU. Raises if a picture is submitted
    while one of its sources is not resident, exactly what the device store would report as NOT_FOUND."""

    def __init__(self, enc_mode, ref_counts, store_slots=64):
        self.enc_mode, self.ref_counts, self.store_slots = enc_mode, ref_counts, store_slots
        self.store, self.pending, self.next_ticket = {}, {}, 1
        self.puts = self.releases = self.max_resident = 0

    def store_put(self, key, picture_number, luma):
        if key not in self.store and len(self.store) >= self.store_slots:
            raise RuntimeError("store full")
        self.store[key] = np.array(luma, np.uint8)
        self.puts += 1
        self.max_resident = max(self.max_resident, len(self.store))

    def store_release(self, key):
        del self.store[key]
        self.releases += 1

    def make_picture(self, width, height, poc, temporal_layer, is_ref, refs, ctrl):
        return gop.Picture(poc=poc, temporal_layer=temporal_layer, is_ref=is_ref, refs=[list(refs[0]), list(refs[1])])

    def submit(self, pic):
        need = [pic.poc] + [r for lst in pic.refs for r in lst]
        missing = [k for k in need if k not in self.store]
        if missing:
            raise KeyError("sources %s not resident" % missing)
        h, w = self.store[pic.poc].shape
        frames = np.zeros((max(need) + 1, h, w), np.uint8)
        for k in need:
            frames[k] = self.store[k]
        t = self.next_ticket
        self.next_ticket += 1
        self.pending[t] = oracle_handover(frames, [pic], self.enc_mode, self.ref_counts, threads=2)[0]
        return t

    def wait(self, ticket):
        bd = self.pending[ticket]
        return {f: np.array(bd[f]) for f in ("best_sad", "best_mv", "zz_sad", "sr_divisor", "hme_sc_x", "hme_sc_y", "do_ref")}

    def done(self, ticket):
        del self.pending[ticket]


def handover_records(results, n_b64):
    """refio.B64_DUMP records carrying what a device hands to the encoder's ME threads (best SAD / MV, search centres,
    HME SADs, do_ref, zero-MV SADs, search-range divisors), from an engine output dict or an oracle record array."""
    rec = np.zeros(n_b64, refio.B64_DUMP)
    if isinstance(results, dict):
        sr = results["search_results"]
        rec["best_sad"], rec["best_mv"] = results["best_sad"], results["best_mv"]
        rec["zz_sad"], rec["sr_divisor"] = results["zz_sad"], results["sr_divisor"]
        for f in ("hme_sc_x", "hme_sc_y", "hme_sad", "do_ref"):
            rec[f] = sr[f]
    else:
        for f in ("best_sad", "best_mv", "zz_sad", "sr_divisor", "hme_sc_x", "hme_sc_y", "hme_sad", "do_ref"):
            rec[f] = results[f]
    return rec


def tail_mismatches(tail_recs, golden_b64, pics, pruning):
    """Fields of the host tail's output (candidate lists, MV arrays, distortions, GM flags, final do_ref / hme_sad)
    that differ from a reference dump; [] when the hand-over + host tail reproduce the reference's final state."""
    msgs = []
    for i, ((_, bd), p) in enumerate(zip(tail_recs, pics)):
        want = canonical(golden_b64[i], p, pruning[i])
        got = canonical(bd, p, pruning[i])
        for f in TAIL_FIELDS + ["best_sad", "best_mv"]:
            if not np.array_equal(got[f], want[f]):
                msgs.append("picture %d (poc %d): %s" % (i, p.poc, f))
    return msgs
