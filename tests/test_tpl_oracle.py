"""CPU tests of the TPL dispenser's source-based mode search: the restatement (oracle/tpl_oracle.c) against what the real
reference computed (tests/golden/tpl_*.npz, written by tests/golden/make_tpl_golden.py from oracle/_ref/ref_me), and,
where the reference build exists, the fixtures against a fresh reference run."""
import ctypes
import json
import os

import numpy as np
import pytest

import refio
import support
import tpl_cases
import tplio
from tpl_cases import CASES, SEARCH_FIELDS

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name, c):
    """-> frames, pics, per picture (controls, me inputs, reference block records, quantiser)"""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    frames, pics, rc = support.make_job(**tpl_cases.job_args(c))
    assert int(g["n_pics"]) == len(pics)
    per = []
    for i in range(len(pics)):
        shape = g["pic%d_me_shape" % i]
        me = dict(total_me_candidate_index=g["pic%d_me_cnt" % i], me_candidate_array=g["pic%d_me_cand" % i],
                  me_mv_array=g["pic%d_me_mv" % i], enable_me_16x16=int(shape[0]), max_cand=int(shape[1]),
                  max_refs=int(shape[2]), max_l0=int(shape[3]))
        per.append((tplio.controls_of(g["pic%d_tpl" % i]), me, g["pic%d_blocks" % i], tplio.quant_of(g["pic%d_tpl" % i])))
    return frames, pics, per


def test_oracle_refuses_what_it_does_not_restate():
    frames, pics, per = load_case("tpl4_m8_200x136", CASES["tpl4_m8_200x136"])
    ctrl, me, tb, quant = per[0]
    for change, q in ((dict(tpl_opt_flag=0, dispenser_search_level=1), None), (dict(use_pred_sad_in_inter_search=0, dispenser_search_level=1), None),
                      (dict(use_pred_sad_in_intra_search=0, subsample_tx=1), None), (dict(dispenser_search_level=1, subsample_tx=1), quant),
                      (dict(dispenser_search_level=2, subsample_tx=2), quant), (dict(pf_shape=3), None)):
        p, keep, refs = picture_of(frames, pics[0], dict(ctrl, **change), me, quant=q)
        with pytest.raises(RuntimeError, match="-5"):
            tplio.oracle_search(frames[pics[0].poc], refs, p)


def picture_of(frames, pic, ctrl, me, **kw):
    h, w = frames.shape[1:]
    p, keep = tplio.make_picture(w, h, ctrl, me, cur_key=pic.poc + 1, ref_keys=([r + 1 for r in pic.refs[0]], [r + 1 for r in pic.refs[1]]), **kw)
    refs = {(l, r): frames[poc] for l in range(2) for r, poc in enumerate(pic.refs[l])}
    return p, keep, refs


def assert_blocks_equal(got, exp, what):
    for f in SEARCH_FIELDS:
        if not np.array_equal(got[f], exp[f]):
            i = tuple(np.argwhere(got[f] != exp[f])[0])
            raise AssertionError("%s: %s differs at (b64, block) %s: %s != %s (%d places)" %
                                 (what, f, i, got[f][i], exp[f][i], int((got[f] != exp[f]).sum())))


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_reference_tpl_search(name):
    frames, pics, per = load_case(name, CASES[name])
    n_blocks = n_intra = n_src = 0
    for pic, (ctrl, me, tb, quant) in zip(pics, per):
        src = tplio.src_dist_available(ctrl)
        p, keep, refs = picture_of(frames, pic, ctrl, me, quant=quant if src else None)
        got = tplio.oracle_search(frames[pic.poc], refs, p)
        exp = tplio.expected_from_reference(tb, ctrl, src_dist=src)
        assert_blocks_equal(got, exp, "%s poc %d" % (name, pic.poc))
        n_src += int((exp["src_recon_error"] > 0).sum())
        n_blocks += int(exp["valid"].sum())
        n_intra += int((exp["valid"] & (exp["best_is_inter"] == 0)).sum())
    # both outcomes occur in every case but the flat one without tpl_opt_flag, where every cost is 0 and intra stays
    assert n_blocks > 0 and 0 < n_intra and (n_intra < n_blocks or name == "tpl1_m2_flat_128x64")
    assert (n_src > 0) == (tplio.src_dist_available(per[0][0]) and n_intra < n_blocks)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_reference_tpl_reconstruction(name):
    """The dispenser's reconstruction path (prediction from the reconstructed / the reference picture, transform,
    quantisation, inverse transform) on the search's results: the reconstructed picture's hash and the four statistics of
    every block as the real reference left them."""
    frames, pics, per = load_case(name, CASES[name])
    if not tplio.recon_available(per[0][0]):
        pytest.skip("not restated")
    for pic, (ctrl, me, tb, quant) in zip(pics, per):
        p, keep, refs = picture_of(frames, pic, ctrl, me, quant=quant)
        blocks = tplio.oracle_search(frames[pic.poc], refs, p)
        recon, stats = tplio.oracle_recon(frames[pic.poc], refs, p, blocks)
        g = np.load(os.path.join(GOLDEN, name + ".npz"))
        v = tb["valid"] == 1
        for f in ("srcrf_dist", "recrf_dist", "srcrf_rate", "recrf_rate"):
            bad = np.argwhere(v & (stats[f] != tb[f]))
            assert not len(bad), "%s poc %d: %s differs at (b64, block) %s: %d != %d (%d places)" % (
                name, pic.poc, f, tuple(bad[0]), stats[f][tuple(bad[0])], tb[f][tuple(bad[0])], len(bad))
        assert tplio.recon_hash(recon) == int(g["pic%d_tpl" % pics.index(pic)]["recon_hash"]), (name, pic.poc)


@pytest.mark.parametrize("name", sorted(n for n in CASES if n.startswith("tpl1_")))
def test_oracle_reproduces_every_intra_mode_cost_the_reference_recorded(name):
    """tpl_opt_flag = 0: the reference's record holds the cost of each of the thirteen intra modes of every block, in mode
    order, in front of the inter costs; the restatement's predictors (neighbour arrays with their paddings, filtered edges,
    the three directional zones, smooth and Paeth) must give every one of them."""
    frames, pics, per = load_case(name, CASES[name])
    h, w = frames.shape[1:]
    n = 0
    for pic, (ctrl, me, tb, quant) in zip(pics, per):
        assert not ctrl["tpl_opt_flag"] and not ctrl["disable_intra_pred"]
        for b64 in range(tb.shape[0]):
            sx, sy = (b64 % ((w + 63) // 64)) * 64, (b64 // ((w + 63) // 64)) * 64
            for bi in range(5, 21):
                if not tb[b64, bi]["valid"]:
                    continue
                bx, by, _ = tplio.BLOCK_GEOMETRY[bi]
                got = tplio.oracle_intra_costs(frames[pic.poc], sx + bx, sy + by)
                assert got.tolist() == tb[b64, bi]["costs"][:13].tolist(), (name, pic.poc, b64, bi)
                n += 1
    assert n > 0


def test_block_geometry_follows_the_dispensers_tables():
    """21 blocks per 64x64 block: one 64x64, four 32x32 in raster order, sixteen 16x16 in z-order; valid blocks of a
    ragged picture are the ones at least half inside."""
    assert tplio.BLOCK_GEOMETRY[0] == (0, 0, 64) and tplio.BLOCK_GEOMETRY[2] == (32, 0, 32) and tplio.BLOCK_GEOMETRY[3] == (0, 32, 32)
    assert tplio.BLOCK_GEOMETRY[5:9] == [(0, 0, 16), (16, 0, 16), (0, 16, 16), (16, 16, 16)]
    assert tplio.BLOCK_GEOMETRY[9] == (32, 0, 16) and tplio.BLOCK_GEOMETRY[20] == (48, 48, 16)
    frames, pics, per = load_case("tpl7_m11_328x200", CASES["tpl7_m11_328x200"])
    ctrl, me, tb, quant = per[0]
    w, h = 328, 200
    for b64 in range(tb.shape[0]):
        sx, sy = (b64 % 6) * 64, (b64 // 6) * 64
        complete = sx + 64 <= w and sy + 64 <= h
        for bi, (bx, by, size) in enumerate(tplio.BLOCK_GEOMETRY):
            in_grid = size == (32 if complete else 16) and sx + bx + size // 2 <= w and sy + by + size // 2 <= h
            assert bool(tb[b64, bi]["valid"]) == in_grid, (b64, bi)


def test_i_slice_and_skipped_references():
    """No candidates on an I slice (best_rf_idx -1, DC wins); a skipped reference's candidates do not count."""
    frames, pics, per = load_case("tpl4_m8_200x136", CASES["tpl4_m8_200x136"])
    pic = next(p for p in pics if p.is_ref and len(p.refs[0]) + len(p.refs[1]) >= 2)
    ctrl, me, tb, quant = per[pics.index(pic)]
    ctrl = dict(ctrl, disable_intra_pred=0)
    p, keep, refs = picture_of(frames, pic, ctrl, me, i_slice=1)
    got = tplio.oracle_search(frames[pic.poc], refs, p)
    v = got["valid"] == 1
    assert (got["best_rf_idx"][v] == -1).all() and (got["best_is_inter"][v] == 0).all()
    assert (got["best_inter_cost"][v] == tplio.INT64_MAX).all() and (got["mv_x"][v] == 0).all()
    base = tplio.oracle_search(frames[pic.poc], refs, picture_of(frames, pic, ctrl, me)[0])
    skip = [[0] * 4, [0] * 4]
    skip[0][0] = 1
    p, keep, refs = picture_of(frames, pic, ctrl, me, ref_skip=skip)
    got = tplio.oracle_search(frames[pic.poc], refs, p)
    assert (got["best_rf_idx"][v] != 0).all() and (base["best_rf_idx"][v] == 0).any()
    assert (got["best_inter_cost"][v] >= base["best_inter_cost"][v]).all()


def test_candidate_of_a_missing_reference_is_an_error():
    frames, pics, per = load_case("tpl4_m8_200x136", CASES["tpl4_m8_200x136"])
    pic = next(p for p in pics if len(p.refs[1]) >= 1)
    ctrl, me, tb, quant = per[pics.index(pic)]
    p, keep, refs = picture_of(frames, pic, ctrl, me, n_ref=(len(pic.refs[0]), 0))
    with pytest.raises(RuntimeError, match="-1"):
        tplio.oracle_search(frames[pic.poc], refs, p)


@pytest.mark.skipif(not refio.have_reference(), reason="needs the reference build (oracle/_ref)")
@pytest.mark.parametrize("name", ["tpl7_m11_328x200", "tpl4_m8_noise_192x128", "tpl3_m6_192x128"])
def test_fixture_equals_a_fresh_reference_run(name):
    c = CASES[name]
    frames, pics, per = load_case(name, c)
    frames2, pics2, rc = support.make_job(**tpl_cases.job_args(c))
    hdr, recs = refio.run_reference_tpl(frames2, pics2, c["enc_mode"], c["level"], c["qp"], ref_counts=rc)
    for (ctrl, me, tb, quant), (pd, bd, tp, tb2) in zip(per, recs):
        assert tb.tobytes() == tb2.tobytes() and ctrl == tplio.controls_of(tp)


@pytest.mark.skipif(not refio.have_reference(), reason="needs the reference build (oracle/_ref)")
def test_reference_digest_of_the_1080p_case():
    import hashlib
    name, c = next(iter(tpl_cases.DIGEST_CASES.items()))
    frames, pics, rc = support.make_job(**tpl_cases.job_args(c))
    hdr, recs = refio.run_reference_tpl(frames, pics, c["enc_mode"], c["level"], c["qp"], ref_counts=rc)
    h = hashlib.sha256()
    for pic, (pd, bd, tp, tb) in zip(pics, recs):
        ctrl = tplio.controls_of(tp)
        exp = tplio.expected_from_reference(tb, ctrl, src_dist=True)
        for f in SEARCH_FIELDS:
            h.update(np.ascontiguousarray(exp[f]).tobytes())
    want = json.load(open(os.path.join(GOLDEN, "tpl_digests.json")))[name]
    assert h.hexdigest() == want["sha256"]
    assert want["controls"] == [tplio.controls_of(r[2]) for r in recs]
    # the restatement on the last picture of the same run
    assert ctrl["pf_shape"] == 2 and want["quant"] == [list(q) for q in tplio.quant_of(tp)]
    p, keep, refs = picture_of(frames, pic, ctrl, tplio.me_inputs_of(pd, bd), quant=tplio.quant_of(tp))
    assert_blocks_equal(tplio.oracle_search(frames[pic.poc], refs, p), exp, name)


@pytest.mark.skipif(not refio.have_reference(), reason="needs the reference build (oracle/_ref)")
@pytest.mark.parametrize("w,h", [(16, 16), (16, 8), (16, 4), (32, 8)])
def test_forward_dct_equals_the_references_on_random_and_extreme_residuals(w, h):
    """The restatement's forward DCT, size by size and shape by shape, against svt_av1_wht_fwd_txfm of the reference build:
    random residuals of every magnitude, the extremes of 8-bit residuals (+-255 everywhere, checkerboards, single rows /
    columns)."""
    rng = np.random.default_rng(7 + w + h)
    blocks = [rng.integers(-m, m + 1, (40, h, w)) for m in (1, 7, 60, 255)]
    ext = np.zeros((8, h, w), np.int64)
    ext[0] = 255
    ext[1] = -255
    ext[2] = 255 * (1 - 2 * ((np.arange(h)[:, None] + np.arange(w)[None, :]) & 1))
    ext[3, 0, :] = 255
    ext[4, :, 0] = -255
    ext[5] = 255 * (1 - 2 * (np.arange(w)[None, :] & 1))
    ext[6] = -255 * (1 - 2 * (np.arange(h)[:, None] & 1))
    ext[7, h - 1, w - 1] = 255
    blocks = np.concatenate(blocks + [ext]).astype(np.int16)
    want = refio.reference_dct(blocks)
    for shape in range(3):
        for i, b in enumerate(blocks):
            got = tplio.oracle_fwd_dct(b, shape)
            assert np.array_equal(got, want[shape, i]), (shape, i)
    for shape in (1, 2):  # the reduced shapes are the whole transform with everything outside the corner zeroed
        m = np.zeros((h, w), bool)
        m[:h >> shape, :w >> shape] = True
        assert np.array_equal(want[shape], np.where(m, want[0], 0))
