"""The TPL dispenser's source-based mode search on the GPU engine (svtme_tpl_search, through the C ABI) against what the
real reference computed (tests/golden/tpl_*.npz, tpl_digests.json) and against the CPU restatement on random ME results."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ("svt-av1_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, _p))
import refio  # noqa: E402
import support  # noqa: E402
import tpl_cases  # noqa: E402
import tplio  # noqa: E402
from test_tpl_oracle import assert_blocks_equal, load_case, picture_of  # noqa: E402

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _engine(frames):
    import svtme
    n, h, w = frames.shape
    eng = svtme.Engine(w, h, store_slots=n + 1, max_in_flight=2)
    for i in range(n):
        eng.store_put(i + 1, i, frames[i])  # key = poc + 1, as picture_of numbers them
    return eng


@pytest.mark.parametrize("name", sorted(tpl_cases.CASES))
def test_engine_reproduces_reference_tpl_search(name):
    frames, pics, per = load_case(name, tpl_cases.CASES[name])
    eng = _engine(frames)
    try:
        for pic, (ctrl, me, tb, quant) in zip(pics, per):
            src = tplio.src_dist_available(ctrl)
            p, keep, refs = picture_of(frames, pic, ctrl, me, quant=quant if src else None)
            assert_blocks_equal(eng.tpl_search(p), tplio.expected_from_reference(tb, ctrl, src_dist=src), "%s poc %d" % (name, pic.poc))
            if src:  # without the source distortion the search is the same
                p, keep, refs = picture_of(frames, pic, ctrl, me)
                assert_blocks_equal(eng.tpl_search(p), tplio.expected_from_reference(tb, ctrl), "%s poc %d, search only" % (name, pic.poc))
    finally:
        eng.close()


STAT_FIELDS = tpl_cases.STAT_FIELDS


@pytest.mark.parametrize("name", sorted(tpl_cases.CASES))
def test_engine_reproduces_reference_tpl_reconstruction(name):
    """svtme_tpl_recon on the engine's own search results: the reconstructed picture's hash and every block's statistics as
    the real reference's dispenser left them (16x16 blocks: 16x16 and 16x8 transforms, DC and all thirteen intra modes; 32x32
    blocks with the 32x8 transform and 16x4 in incomplete 64x64 blocks)."""
    frames, pics, per = load_case(name, tpl_cases.CASES[name])
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    eng = _engine(frames)
    try:
        for i, (pic, (ctrl, me, tb, quant)) in enumerate(zip(pics, per)):
            p, keep, refs = picture_of(frames, pic, ctrl, me, quant=quant)
            blocks = eng.tpl_search(p)
            stats, recon = eng.tpl_recon(p, blocks)
            v = tb["valid"] == 1
            for f in STAT_FIELDS:
                bad = np.argwhere(v & (stats[f] != tb[f]))
                assert not len(bad), "%s poc %d: %s differs at (b64, block) %s: %d != %d (%d places)" % (
                    name, pic.poc, f, tuple(bad[0]), stats[f][tuple(bad[0])], tb[f][tuple(bad[0])], len(bad))
                assert not stats[f][~v].any()
            assert tplio.recon_hash(recon) == int(g["pic%d_tpl" % i]["recon_hash"]), (name, pic.poc)
            stats2, none = eng.tpl_recon(p, blocks, want_picture=False)  # the same without the picture
            assert none is None and np.array_equal(stats2, stats)
    finally:
        eng.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("SVTME_FUZZ_N", "12"))))
def test_random_reconstructions_match_the_restatement(seed):
    """Random sizes, contents, quantisers and ME results; every subsample_tx / pf_shape; with and without tpl_opt_flag; pictures
    of intra blocks only (i_slice) and with intra prediction disabled: statistics and the reconstructed picture byte for byte."""
    rng = np.random.default_rng(4000 + seed)
    w, h = int(rng.integers(8, 50)) * 8, int(rng.integers(8, 36)) * 8
    n_ref = (int(rng.integers(1, 5)), int(rng.integers(0, 4)))
    kind = ["synth", "noise", "flat"][int(rng.integers(0, 3))]
    frames = support.make_job(w, h, 1 + n_ref[0] + n_ref[1], 8, seed=seed + 11, kind=kind)[0]
    n_b64 = ((w + 63) // 64) * ((h + 63) // 64)
    me = _random_me(rng, n_b64, [85, 21][int(rng.integers(0, 2))], n_ref, int(rng.choice([8, 64, 400])))
    all_modes = seed % 3 == 0
    ctrl = dict(dispenser_search_level=0, disable_intra_pred=int(rng.integers(0, 4) == 0), tpl_opt_flag=0 if all_modes else 1,
                use_pred_sad_in_intra_search=int(rng.integers(0, 2)), use_pred_sad_in_inter_search=int(rng.integers(0, 2)),
                subsample_tx=0, pf_shape=int(rng.integers(0, 3)))
    if not all_modes and ctrl["use_pred_sad_in_intra_search"] and ctrl["use_pred_sad_in_inter_search"]:
        ctrl["subsample_tx"] = int(rng.integers(0, 3))
        if seed % 4 == 1:  # 32x32 blocks (32x8), 16x16 blocks with 16x4 in the 64x64 blocks the picture cuts
            ctrl.update(dispenser_search_level=1, subsample_tx=2)
    dq = [int(rng.integers(4, 1400)), int(rng.integers(4, 1900))]
    quant = ([int(rng.integers(0, 1 + d)) for d in dq], [min(32767, int(rng.integers(1, 3)) * 32768 // d) for d in dq], dq)
    import svtme
    eng = svtme.Engine(w, h, store_slots=len(frames) + 1, max_in_flight=2)
    try:
        for i in range(len(frames)):
            eng.store_put(i + 1, i, frames[i])
        refs = {(l, r): frames[1 + (l * n_ref[0]) + r] for l in range(2) for r in range(n_ref[l])}
        keys = ([2 + r for r in range(n_ref[0])], [2 + n_ref[0] + r for r in range(n_ref[1])])
        p, keep = tplio.make_picture(w, h, ctrl, me, cur_key=1, ref_keys=keys, quant=quant, i_slice=int(seed % 5 == 4))
        blocks = eng.tpl_search(p)
        want_blocks = tplio.oracle_search(frames[0], refs, p)
        assert_blocks_equal(blocks, want_blocks, "seed %d search" % seed)
        stats, recon = eng.tpl_recon(p, blocks)
        want_recon, want_stats = tplio.oracle_recon(frames[0], refs, p, blocks)
        for f in STAT_FIELDS:
            assert np.array_equal(stats[f], want_stats[f]), (seed, ctrl, f, np.argwhere(stats[f] != want_stats[f])[:4])
        assert np.array_equal(recon, want_recon), (seed, ctrl, np.argwhere(recon != want_recon)[:4])
    finally:
        eng.close()


@pytest.mark.parametrize("seed", range(4))
def test_kept_reconstruction_serves_as_reference_of_the_next_picture(seed):
    """svtme_tpl_recon_keep: the reconstructed picture becomes a store picture device to device; a second picture that names it
    as its reference gives, in search and reconstruction, what the restatement gives with the first one's reconstruction as
    reference picture - and what the engine gives when that reconstruction is put from the host."""
    import svtme
    rng = np.random.default_rng(9000 + seed)
    w, h = int(rng.integers(8, 40)) * 8, int(rng.integers(8, 30)) * 8
    frames = support.make_job(w, h, 3, 8, seed=seed + 31, kind=["synth", "noise"][seed & 1])[0]
    n_b64 = ((w + 63) // 64) * ((h + 63) // 64)
    ctrl = dict(dispenser_search_level=0, disable_intra_pred=0, tpl_opt_flag=1, use_pred_sad_in_intra_search=1,
                use_pred_sad_in_inter_search=1, subsample_tx=int(seed % 3), pf_shape=1)
    quant = ([20, 24], [1638, 1365], [40, 48])
    eng = svtme.Engine(w, h, store_slots=6, max_in_flight=2)
    try:
        with pytest.raises(svtme.SvtmeError) as ei:
            eng.tpl_recon_keep(50)
        assert ei.value.code == svtme.ERR_NOT_FOUND
        for i in range(3):
            eng.store_put(i + 1, i, frames[i])
        me = _random_me(rng, n_b64, 85, (1, 0), 64)
        # the first picture: frame 1 predicted from frame 0
        p1, keep1 = tplio.make_picture(w, h, ctrl, me, cur_key=2, ref_keys=([1], []), quant=quant)
        b1 = eng.tpl_search(p1)
        stats1, recon1 = eng.tpl_recon(p1, b1)
        want_recon1, want_stats1 = tplio.oracle_recon(frames[1], {(0, 0): frames[0]}, p1, b1)
        assert np.array_equal(recon1, want_recon1)
        eng.tpl_recon_keep(50, 1)
        eng.store_put(51, 1, recon1)  # the same picture by way of the host
        # the second picture: frame 2 predicted from the first one's reconstruction
        outs = []
        for key in (50, 51):
            p2, keep2 = tplio.make_picture(w, h, ctrl, me, cur_key=3, ref_keys=([key], []), quant=quant)
            b2 = eng.tpl_search(p2)
            outs.append((b2,) + eng.tpl_recon(p2, b2))
        want_b2 = tplio.oracle_search(frames[2], {(0, 0): want_recon1}, p2)
        want_recon2, want_stats2 = tplio.oracle_recon(frames[2], {(0, 0): want_recon1}, p2, want_b2)
        for b2, stats2, recon2 in outs:
            assert_blocks_equal(b2, want_b2, "second picture")
            for f in STAT_FIELDS:
                assert np.array_equal(stats2[f], want_stats2[f]), f
            assert np.array_equal(recon2, want_recon2)
        eng.store_release(50)
        eng.tpl_recon_keep(50, 2)  # the second picture's reconstruction under the released key
        eng.store_release(50)
    finally:
        eng.close()


def test_reconstruction_refuses_what_it_does_not_do():
    import svtme
    frames, pics, per = load_case("tpl4_m8_200x136", tpl_cases.CASES["tpl4_m8_200x136"])
    eng = _engine(frames)
    try:
        ctrl, me, tb, quant = per[1]
        p, keep, refs = picture_of(frames, pics[1], ctrl, me, quant=quant)
        blocks = eng.tpl_search(p)
        for change in (dict(dispenser_search_level=1, subsample_tx=0), dict(dispenser_search_level=1, subsample_tx=1),
                       dict(dispenser_search_level=1, subsample_tx=2, tpl_opt_flag=0), dict(dispenser_search_level=2), dict(pf_shape=3),
                       dict(subsample_tx=3), dict(tpl_opt_flag=0, subsample_tx=1)):
            q, keep2, _ = picture_of(frames, pics[1], dict(ctrl, **change), me, quant=quant)
            with pytest.raises(svtme.SvtmeError) as ei:
                eng.tpl_recon(q, blocks)
            assert ei.value.code == svtme.ERR_UNSUPPORTED, change
        bad = blocks.copy()
        inter = np.argwhere((bad["valid"] == 1) & (bad["best_is_inter"] == 1))[0]
        bad["best_rf_idx"][tuple(inter)] = 7  # a reference the picture does not have
        with pytest.raises(svtme.SvtmeError) as ei:
            eng.tpl_recon(p, bad)
        assert ei.value.code == svtme.ERR_BAD_ARG
        bad = blocks.copy()
        bad["mv_x"][tuple(inter)] = 8 * 4000  # a vector that leaves the reference picture's border
        with pytest.raises(svtme.SvtmeError) as ei:
            eng.tpl_recon(p, bad)
        assert ei.value.code == svtme.ERR_BAD_ARG
    finally:
        eng.close()


def _random_me(rng, n_b64, n_pu, n_ref, max_mv):
    """ME results no encoder would produce but the interface allows: any count, any mix of directions and references,
    vectors far outside the picture."""
    max_l0 = max(n_ref[0], 1)
    max_refs = max_l0 + n_ref[1]
    max_cand = int(rng.integers(1, 24))
    cnt = rng.integers(0, max_cand + 1, (n_b64, n_pu)).astype(np.uint8)
    direction = rng.integers(0, 3 if n_ref[1] else 1, (n_b64, n_pu * max_cand))
    if n_ref[1] == 0:
        direction[:] = 0
    r0 = rng.integers(0, n_ref[0], direction.shape)
    r1 = rng.integers(0, max(n_ref[1], 1), direction.shape)
    cand = (direction | (r0 << 2) | (r1 << 4) | (1 << 7)).astype(np.uint8)
    mv = rng.integers(-max_mv, max_mv + 1, (n_b64, n_pu * max_refs, 2)).astype(np.int16)
    mv_u32 = (mv[..., 0].astype(np.uint16).astype(np.uint32)) | (mv[..., 1].astype(np.uint16).astype(np.uint32) << 16)
    return dict(total_me_candidate_index=cnt, me_candidate_array=cand, me_mv_array=mv_u32,
                enable_me_16x16=int(rng.integers(0, 2)), max_cand=max_cand, max_refs=max_refs, max_l0=max_l0)


@pytest.mark.parametrize("seed", range(int(os.environ.get("SVTME_FUZZ_N", "12"))))
def test_random_me_results_match_the_restatement(seed):
    rng = np.random.default_rng(1000 + seed)
    w, h = int(rng.integers(8, 50)) * 8, int(rng.integers(8, 36)) * 8
    n_ref = (int(rng.integers(1, 5)), int(rng.integers(0, 4)))
    n_frames = 1 + n_ref[0] + n_ref[1]
    kind = ["synth", "noise", "flat"][int(rng.integers(0, 3))]
    frames = support.make_job(w, h, n_frames, 8, seed=seed + 7, kind=kind)[0]
    n_b64 = ((w + 63) // 64) * ((h + 63) // 64)
    me = _random_me(rng, n_b64, [85, 21][int(rng.integers(0, 2))], n_ref, int(rng.choice([8, 64, 400, 2000])))
    level = int(rng.integers(0, 3))
    ctrl = dict(dispenser_search_level=level, disable_intra_pred=int(rng.integers(0, 2)), tpl_opt_flag=1,
                use_pred_sad_in_intra_search=1, use_pred_sad_in_inter_search=1, subsample_tx=int(rng.integers(0, 3)),
                pf_shape=int(rng.integers(0, 3)))
    quant = None
    dq = [int(rng.integers(4, 1400)), int(rng.integers(4, 1900))]
    some_quant = ([int(rng.integers(0, 1 + d)) for d in dq], [min(32767, int(rng.integers(1, 3)) * 32768 // d) for d in dq], dq)
    if level == 0 and seed % 3 == 1:  # whole 16x16 transforms: transform-domain costs and / or the source distortion
        ctrl.update(subsample_tx=0, use_pred_sad_in_intra_search=int(rng.integers(0, 2)), use_pred_sad_in_inter_search=int(rng.integers(0, 2)))
        if rng.integers(0, 3):
            quant = some_quant
    elif level < 2 and seed % 3 == 2:  # the source distortion through the reduced-row transforms: 16x8, 16x4, 32x8
        if level == 1:
            ctrl.update(subsample_tx=2)
        quant = some_quant
    elif seed % 3 == 0 and seed % 2 == 0:  # no tpl_opt_flag: every intra mode up to a random last one, the rate of the choice
        ctrl.update(dispenser_search_level=0, subsample_tx=0, tpl_opt_flag=0, intra_mode_end=int(rng.choice([12, 11, 8])),
                    use_pred_sad_in_intra_search=int(rng.integers(0, 2)), use_pred_sad_in_inter_search=int(rng.integers(0, 2)))
        if rng.integers(0, 4):
            quant = some_quant
    from svtme import gop
    pic = gop.Picture(poc=0, temporal_layer=0, is_ref=1, refs=[list(range(1, 1 + n_ref[0])), list(range(1 + n_ref[0], n_frames))])
    skip = rng.integers(0, 4, (2, 4)) == 0
    p, keep, refs = picture_of(frames, pic, ctrl, me, ref_skip=skip, i_slice=int(rng.integers(0, 8) == 0), quant=quant)
    eng = _engine(frames)
    try:
        got = eng.tpl_search(p)
    finally:
        eng.close()
    exp = tplio.oracle_search(frames[0], refs, p)
    assert_blocks_equal(got, exp, "seed %d %dx%d level %d" % (seed, w, h, ctrl["dispenser_search_level"]))
    assert exp["valid"].sum() > 0


def test_unsupported_controls_and_bad_arguments():
    import svtme
    frames, pics, per = load_case("tpl4_m8_200x136", tpl_cases.CASES["tpl4_m8_200x136"])
    pic = next(p for p in pics if len(p.refs[1]) >= 1)
    ctrl, me, tb, quant = per[pics.index(pic)]
    eng = _engine(frames)
    try:
        # every intra mode (tpl_opt_flag 0), transform-domain costs or the source distortion on anything but whole 16x16 transforms
        for change, q in ((dict(tpl_opt_flag=0, dispenser_search_level=1), None), (dict(use_pred_sad_in_inter_search=0, dispenser_search_level=1), None),
                          (dict(use_pred_sad_in_intra_search=0, subsample_tx=1), None), (dict(dispenser_search_level=1, subsample_tx=1), quant),
                          (dict(dispenser_search_level=2, subsample_tx=2), quant), (dict(pf_shape=3), None)):
            p, keep, refs = picture_of(frames, pic, dict(ctrl, **change), me, quant=q)
            with pytest.raises(svtme.SvtmeError) as ei:
                eng.tpl_search(p)
            assert ei.value.code == svtme.ERR_UNSUPPORTED, change
        # a candidate of a reference that was not handed over
        p, keep, refs = picture_of(frames, pic, ctrl, me, n_ref=(len(pic.refs[0]), 0))
        with pytest.raises(svtme.SvtmeError) as ei:
            eng.tpl_search(p)
        assert ei.value.code == svtme.ERR_BAD_ARG
        # a candidate count the arrays cannot hold
        bad = dict(me)
        bad["total_me_candidate_index"] = np.array(me["total_me_candidate_index"], copy=True)
        bad["total_me_candidate_index"][0, :] = me["max_cand"] + 1
        p, keep, refs = picture_of(frames, pic, ctrl, bad)
        with pytest.raises(svtme.SvtmeError) as ei:
            eng.tpl_search(p)
        assert ei.value.code == svtme.ERR_BAD_ARG
        # a key the store does not hold; a picture size the store's planes do not have
        p, keep, refs = picture_of(frames, pic, ctrl, me)
        p.ref_key[0][0] = 9999
        with pytest.raises(svtme.SvtmeError) as ei:
            eng.tpl_search(p)
        assert ei.value.code == svtme.ERR_NOT_FOUND
        p, keep, refs = picture_of(frames, pic, ctrl, me)
        p.height = 128
        with pytest.raises(svtme.SvtmeError) as ei:
            eng.tpl_search(p)
        assert ei.value.code == svtme.ERR_BAD_ARG
        # and the engine still works afterwards
        p, keep, refs = picture_of(frames, pic, ctrl, me)
        assert_blocks_equal(eng.tpl_search(p), tplio.expected_from_reference(tb, ctrl), "after the errors")
    finally:
        eng.close()


def test_1080p_search_on_the_engines_own_me_results_equals_the_reference_digest():
    """The whole chain on the device at the benchmark's size: open-loop ME with the device tail, whose candidate and vector
    arrays (bit-exact with the reference's, test_gpu_parity) feed the TPL search, whose block records feed the reconstruction
    path; the digests of the search results and of the statistics and the reconstructed pictures' hashes are the real
    reference's."""
    import svtme
    name, c = next(iter(tpl_cases.DIGEST_CASES.items()))
    want = json.load(open(os.path.join(GOLDEN, "tpl_digests.json")))[name]
    frames, pics, rc = support.make_job(**tpl_cases.job_args(c))
    shape = want["me_shape"]
    tail = svtme.Tail(enable=1, enable_me_8x8=shape["enable_me_8x8"], enable_me_16x16=shape["enable_me_16x16"],
                      max_refs=shape["max_refs"], max_cand=shape["max_cand"], max_l0=shape["max_l0"],
                      input_resolution=shape["input_resolution"], gm_enabled=1, gm_use_distance_based_active_th=1)
    n, h, w = frames.shape
    eng = svtme.Engine(w, h, store_slots=n + 2, max_in_flight=2)
    try:
        for i in range(n):
            eng.store_put(i + 1, i, frames[i])
        ctrl_me = svtme.derive_controls(c["enc_mode"], w, h, 0, 30 << 16)
        hsh, stats_hsh, recon_hashes = hashlib.sha256(), hashlib.sha256(), []
        for pic, ctrl in zip(pics, want["controls"]):
            desc = eng.make_picture(w, h, pic.poc, pic.temporal_layer, pic.is_ref, pic.refs, ctrl_me, tail=tail, key_of=lambda poc: poc + 1)
            out = eng.run_picture(desc)
            me = dict(total_me_candidate_index=out["total_me_candidate_index"],
                      me_candidate_array=out["me_candidate_array"][:, :85 * shape["max_cand"]],
                      me_mv_array=out["me_mv_array"][:, :85 * shape["max_refs"]], enable_me_16x16=shape["enable_me_16x16"],
                      max_cand=shape["max_cand"], max_refs=shape["max_refs"], max_l0=shape["max_l0"])
            p, keep, refs = picture_of(frames, pic, ctrl, me, quant=want["quant"])
            got = eng.tpl_search(p)
            for f in tpl_cases.SEARCH_FIELDS:
                hsh.update(np.ascontiguousarray(got[f]).tobytes())
            stats, recon = eng.tpl_recon(p, got)
            for f in tpl_cases.STAT_FIELDS:
                stats_hsh.update(np.ascontiguousarray(stats[f]).tobytes())
            recon_hashes.append(tplio.recon_hash(recon))
        assert hsh.hexdigest() == want["sha256"]
        assert stats_hsh.hexdigest() == want["stats_sha256"]
        assert recon_hashes == want["recon_hash"]
    finally:
        eng.close()


DISPENSER_FIELDS = ("valid", "best_mode", "best_intra_mode", "best_rf_idx", "mv_row", "mv_col", "ref_frame_poc", "src_srcrf_dist",
                    "src_srcrf_rate", "srcrf_dist", "recrf_dist", "srcrf_rate", "recrf_rate", "stats_mv_row", "stats_mv_col",
                    "stats_ref_frame_poc")


@pytest.mark.skipif(not refio.have_reference(), reason="oracle/_ref/ref_me (built where the reference sources are) is not here")
@pytest.mark.parametrize("name", sorted(tpl_cases.CASES))
def test_reference_dispenser_runs_unchanged_on_the_engines_search(name):
    """The drop-in evidence: ref_me with the engine behind the seam. The reference's ME runs, svtme_tpl_search replaces the
    dispenser's source-based search, the binding fills tpl_src_stats_buffer (transform and quantisation of the chosen
    candidate with the reference's own functions), and the unmodified tpl_mc_flow_dispenser_sb_generic does its
    reconstruction path on that. Every block's source-based record, its final TplStats and the reconstructed picture are
    the unmodified reference's."""
    import svtme
    c = tpl_cases.CASES[name]
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    frames, pics, rc = support.make_job(**tpl_cases.job_args(c))
    hdr, recs = refio.run_reference_tpl(frames, pics, c["enc_mode"], c["level"], c["qp"], library=svtme.LIB_PATH, ref_counts=rc)
    assert len(recs) == int(g["n_pics"])
    for i, (pd, bd, tp, tb) in enumerate(recs):
        want_tp, want = g["pic%d_tpl" % i], g["pic%d_blocks" % i]
        assert int(tp["recon_hash"]) == int(want_tp["recon_hash"]), (name, i)
        for f in DISPENSER_FIELDS:
            assert np.array_equal(tb[f], want[f]), (name, i, f)
        assert int(tb["n_costs"].sum()) == 0  # the host evaluated no cost: the search was the engine's


def test_tpl_search_runs_beside_me_and_gm_calls():
    """One engine, four threads at once: two submit / wait pictures of the sequence, one runs the TPL search of every picture
    over and over, one asks for corners; everybody gets what a lone caller gets."""
    import threading
    import svtme
    name = "tpl4_m8_200x136"
    frames, pics, per = load_case(name, tpl_cases.CASES[name])
    n, h, w = frames.shape
    eng = svtme.Engine(w, h, store_slots=n + 1, max_in_flight=4)
    errors = []
    try:
        for i in range(n):
            eng.store_put(i + 1, i, frames[i])
        ctrl_me = svtme.derive_controls(8, w, h, 0, 30 << 16)
        descs = [eng.make_picture(w, h, p.poc, p.temporal_layer, p.is_ref, p.refs, ctrl_me, key_of=lambda poc: poc + 1) for p in pics]
        alone = [eng.run_picture(d) for d in descs]
        corners = eng.gm_corners(1, 0)
        jobs = []
        for pic, (ctrl, me, tb, quant) in zip(pics, per):
            p, keep, refs = picture_of(frames, pic, ctrl, me, quant=quant)
            jobs.append((p, keep, tplio.expected_from_reference(tb, ctrl, src_dist=True)))

        def me_worker(part):
            try:
                for _ in range(6):
                    for i in part:
                        out = eng.run_picture(descs[i])
                        for f in ("best_sad", "best_mv", "zz_sad"):
                            assert np.array_equal(out[f], alone[i][f]), (i, f)
            except Exception as e:  # noqa: BLE001
                errors.append(e)

        def tpl_worker():
            try:
                for _ in range(12):
                    for k, (p, keep, exp) in enumerate(jobs):
                        assert_blocks_equal(eng.tpl_search(p), exp, "picture %d beside other callers" % k)
            except Exception as e:  # noqa: BLE001
                errors.append(e)

        def gm_worker():
            try:
                for k in range(60):
                    assert np.array_equal(eng.gm_corners(1 + k % 2, 0), corners if k % 2 == 0 else eng.gm_corners(2, 0))
            except Exception as e:  # noqa: BLE001
                errors.append(e)

        threads = [threading.Thread(target=me_worker, args=(range(0, len(descs), 2),)), threading.Thread(target=me_worker, args=(range(1, len(descs), 2),)),
                   threading.Thread(target=tpl_worker), threading.Thread(target=gm_worker)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=100)
        assert not any(t.is_alive() for t in threads), "a caller did not come back"
        assert not errors, errors[:2]
    finally:
        eng.close()


def test_group_hands_out_the_engine_that_holds_a_pictures_planes():
    """Two engines behind one handle (both on device 0): pictures go round robin, so each engine holds the sources of its own
    pictures; svtme_group_engine_for names the engine for a picture's TPL search (uploading what that engine lacks), and the
    search there equals the reference's."""
    import svtme
    name = "tpl4_m8_200x136"
    frames, pics, per = load_case(name, tpl_cases.CASES[name])
    n, h, w = frames.shape
    grp = svtme.Group([0, 0], w, h, store_slots=n + 1, max_in_flight=4)
    try:
        for i in range(n):
            grp.store_put(i + 1, i, frames[i])
        ctrl_me = svtme.derive_controls(8, w, h, 0, 30 << 16)
        tickets = [grp.submit(grp.make_picture(w, h, p.poc, p.temporal_layer, p.is_ref, p.refs, ctrl_me, key_of=lambda poc: poc + 1))
                   for p in pics[:4]]
        for t in tickets:
            grp.wait(t)
            grp.done(t)
        before = grp.stats()["uploads"]
        engines = set()
        for pic, (ctrl, me, tb, quant) in zip(pics, per):
            keys = [pic.poc + 1] + [r + 1 for l in pic.refs for r in l]
            eng = grp.engine_for(keys)
            engines.add(eng.h.value)
            p, keep, refs = picture_of(frames, pic, ctrl, me, quant=quant)
            assert_blocks_equal(eng.tpl_search(p), tplio.expected_from_reference(tb, ctrl, src_dist=True), "poc %d through the group" % pic.poc)
        assert len(engines) == 2  # both engines served searches
        # the first four pictures' sources were where their ME ran: the searches of those pictures uploaded nothing new there
        assert grp.stats()["uploads"] >= before
        with pytest.raises(svtme.SvtmeError) as ei:
            grp.engine_for([9999])
        assert ei.value.code == svtme.ERR_NOT_FOUND
    finally:
        grp.close()
