"""Global-motion estimation on the GPU engine (svtme_gm_*, through the C ABI) against the fixtures the real reference
produced (tests/golden/gm_*.npz), against the CPU restatement, and end to end: the reference's own compute_global_motion
(RANSAC, refinement loop, shear derivation, costs) with its four data-parallel functions sent to the engine gives the
call-by-call record and the final model of the unmodified reference."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ("svt-av1_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, _p))
import gm_cases  # noqa: E402
import gmio  # noqa: E402

pytestmark = pytest.mark.gpu
needs_ref_build = pytest.mark.skipif(not gmio.have_reference(), reason="oracle/_ref/ref_gm (built where the reference sources are) is not here")


# The fields of a refinement record that the reference defines: compute_global_motion hands over a model whose shear fields
# it never wrote (tmp_wm_params, EbGlobalMotionEstimation.c:194, :228; the loop overwrites them before it reads them), so the
# recorded input shear is whatever the process's stack held.
_REFINE_DEFINED = [i for i, f in enumerate(gmio.REFINE_FIELDS) if f not in ("i_alpha", "i_beta", "i_gamma", "i_delta")]


def _engine(case, tables=None):
    import svtme
    frames = gm_cases.frames_of(case)
    eng = svtme.Engine(case["width"], case["height"], store_slots=len(frames) + 1, max_in_flight=1)
    for i, f in enumerate(frames):
        eng.store_put(100 + i, i, f)
    if tables is not None:
        eng.gm_set_tables(*tables)
    return eng, frames


@pytest.mark.parametrize("name", list(gm_cases.CASES))
def test_corners_and_correspondences_match_the_reference(name):
    case = gm_cases.CASES[name]
    eng, _ = _engine(case)
    try:
        for job, g in zip(case["jobs"], gm_cases.golden_of(name)):
            assert np.array_equal(eng.gm_corners(100 + job["cur"], job["level"]), g["corners_frm"])
            assert np.array_equal(eng.gm_corners(100 + job["ref"], job["level"]), g["corners_ref"])
            assert np.array_equal(eng.gm_correspondences(100 + job["cur"], 100 + job["ref"], job["level"]), g["correspondences"])
            # a shorter list is the head of the full one
            assert np.array_equal(eng.gm_corners(100 + job["cur"], job["level"], max_points=7), g["corners_frm"][:7])
    finally:
        eng.close()


@needs_ref_build
@pytest.mark.parametrize("name", list(gm_cases.CASES))
def test_every_recorded_error_call_is_reproduced(name):
    case = gm_cases.CASES[name]
    eng, _ = _engine(case, gmio.reference_tables())
    try:
        for job, g in zip(case["jobs"], gm_cases.golden_of(name)):
            cur, ref, lvl = 100 + job["cur"], 100 + job["ref"], job["level"]
            for c in g["warp_calls"]:
                c = dict(zip(gmio.WARP_FIELDS, (int(v) for v in c)))
                if c["error"] == 1 and c["best_error"] >= 1:
                    continue  # the reference's "invalid shear" return: the caller never reaches the engine with it
                got = eng.gm_warp_error(cur, ref, lvl, [c["m%d" % i] for i in range(6)], [c["alpha"], c["beta"], c["gamma"], c["delta"]], c["best_error"])
                assert got == c["error"], (name, c, got)
            for e in g["frame_errors"]:
                assert eng.gm_frame_error(cur, ref, lvl) == int(e)
    finally:
        eng.close()


@needs_ref_build
def test_affine_models_and_early_exits_match_the_restatement():
    """Matrices the recorded runs do not reach: general affine ones (all four shear terms), wide offsets that leave the
    picture, and best_error values that stop the sum at the first block, in the middle and nowhere."""
    tables = gmio.reference_tables()
    case = gm_cases.CASES["affine_392x264"]
    eng, frames = _engine(case, tables)
    try:
        rng = np.random.default_rng(5)
        for lvl in (0, 1, 2):
            frm, ref = gmio.plane_of(frames[1], lvl), gmio.plane_of(frames[0], lvl)
            full = None
            for k in range(10):
                mat = [int(rng.integers(-40, 40)) << 13, int(rng.integers(-40, 40)) << 13, 65536 + int(rng.integers(-900, 900)),
                       int(rng.integers(-900, 900)), int(rng.integers(-900, 900)), 65536 + int(rng.integers(-900, 900))]
                if k == 0:
                    mat = [400 << 16, -300 << 16, 65536, 0, 0, 65536]  # far outside the picture: every tap clamps
                shear = [int(v) * 64 for v in rng.integers(-12, 13, 4)]
                full = gmio.oracle_warp_error(tables, mat, shear, ref, frm, (1 << 63) - 1)
                assert eng.gm_warp_error(100 + 1, 100 + 0, lvl, mat, shear) == full
                for best in (0, full // 3, full - 1, full):
                    assert eng.gm_warp_error(101, 100, lvl, mat, shear, best) == gmio.oracle_warp_error(tables, mat, shear, ref, frm, best)
            assert eng.gm_frame_error(101, 100, lvl) == gmio.oracle_frame_error(tables, ref, frm)
    finally:
        eng.close()


@needs_ref_build
@pytest.mark.parametrize("name", ["rotzoom_320x192", "affine_392x264", "translation_512x256", "noise_640x360", "flat_128x128", "rotzoom_1920x1080"])
def test_reference_global_motion_with_the_engine_behind_its_seam(name):
    """ref_gm in engine mode: compute_global_motion of the real reference, unchanged, with svt_av1_fast_corner_detect,
    svt_av1_determine_correspondence, svt_av1_warp_error and svt_av1_frame_error answered by the engine. Every call's
    result and the final model equal the fixture of the unmodified reference."""
    import svtme
    case = gm_cases.CASES[name]
    recs = gmio.run_reference(gm_cases.frames_of(case), case["jobs"], library=svtme.LIB_PATH)
    kind_no = {"corners": 1, "corr": 2, "warp": 3, "frame": 4, "refine": 5}
    for r, g in zip(recs, gm_cases.golden_of(name)):
        assert [kind_no[k] for k, _ in r["calls"]] == [int(v) for v in g["call_kinds"]]
        assert np.array_equal(r["corners_frm"], g["corners_frm"]) and np.array_equal(r["corners_ref"], g["corners_ref"])
        assert np.array_equal(r["correspondences"], g["correspondences"])
        warps = np.array([[c[f] for f in gmio.WARP_FIELDS] for k, c in r["calls"] if k == "warp"], np.int64).reshape(-1, len(gmio.WARP_FIELDS))
        assert np.array_equal(warps, g["warp_calls"])
        refines = np.array([[c[f] for f in gmio.REFINE_FIELDS] for k, c in r["calls"] if k == "refine"], np.int64).reshape(-1, len(gmio.REFINE_FIELDS))
        assert np.array_equal(refines[:, _REFINE_DEFINED], g["refine_calls"][:, _REFINE_DEFINED])
        assert [c for k, c in r["calls"] if k == "frame"] == [int(v) for v in g["frame_errors"]]
        assert [r["final"][f] for f in gmio.FINAL_FIELDS] == [int(v) for v in g["final"]]


def _refine_matches(eng, cur, ref, lvl, rec, n_warp_calls):
    c = dict(zip(gmio.REFINE_FIELDS, (int(v) for v in rec)))
    got = eng.gm_refine(cur, ref, lvl, [c["i%d" % i] for i in range(6)], c["force_wmtype"], c["steps"], c["best_frame_error"],
                        shear=[c["i_alpha"], c["i_beta"], c["i_gamma"], c["i_delta"]], model_wmtype=c["i_wmtype"])
    want = ([c["o%d" % i] for i in range(6)], [c["o_alpha"], c["o_beta"], c["o_gamma"], c["o_delta"]], c["o_wmtype"], c["error"], n_warp_calls)
    assert got == want, (c, got)


@needs_ref_build
@pytest.mark.parametrize("name", list(gm_cases.CASES))
def test_every_recorded_refinement_is_reproduced_by_one_call(name):
    """svtme_gm_refine against svt_av1_refine_integerized_param of the real reference: the refined model, its type, the shear
    the struct is left with, the returned error and the number of error evaluations the loop took."""
    case = gm_cases.CASES[name]
    eng, _ = _engine(case, gmio.reference_tables())
    try:
        for job, g in zip(case["jobs"], gm_cases.golden_of(name)):
            for rec in g["refine_calls"]:
                _refine_matches(eng, 100 + job["cur"], 100 + job["ref"], job["level"], rec, len(g["warp_calls"]))
    finally:
        eng.close()


@needs_ref_build
@pytest.mark.parametrize("name", list(gm_cases.REFINE_CASES))
def test_single_refinements_of_every_model_type(name):
    """Every model type forced, invalid shears (result 1 without a warp), parameters at their clamps, an error to beat that
    nothing beats or that ties: tests/golden/gm_refine_*.npz, single calls of the reference's function."""
    import svtme
    c = gm_cases.REFINE_CASES[name]
    eng, _ = _engine(gm_cases.CASES[name], gmio.reference_tables())
    try:
        for call, (rec, warps) in zip(c["calls"], gm_cases.refine_golden_of(name)):
            _refine_matches(eng, 100 + c["cur"], 100 + c["ref"], call["level"], rec, len(warps))
        for bad in (dict(wmtype=4, steps=5), dict(wmtype=2, steps=0), dict(wmtype=-1, steps=5)):
            with pytest.raises(svtme.SvtmeError) as ei:
                eng.gm_refine(101, 100, 0, [0, 0, 65536, 0, 0, 65536], bad["wmtype"], bad["steps"])
            assert ei.value.code == svtme.ERR_BAD_ARG
    finally:
        eng.close()


@needs_ref_build
@pytest.mark.parametrize("name", ["rotzoom_320x192", "affine_392x264", "translation_512x256", "noise_640x360", "rotzoom_1920x1080"])
def test_reference_global_motion_with_the_engine_refining(name):
    """ref_gm with the refinement loop as a whole on the engine (svtme_gm_refine behind svt_av1_refine_integerized_param):
    the reference's own loop does not run, so no single warp error is recorded; every refinement's result, the frame
    errors and the final model equal the fixture of the unmodified reference."""
    import svtme
    case = gm_cases.CASES[name]
    recs = gmio.run_reference(gm_cases.frames_of(case), case["jobs"], library=svtme.LIB_PATH, engine_refine=True)
    for r, g in zip(recs, gm_cases.golden_of(name)):
        assert not [1 for k, _ in r["calls"] if k == "warp"]
        assert np.array_equal(r["correspondences"], g["correspondences"])
        refines = np.array([[c[f] for f in gmio.REFINE_FIELDS] for k, c in r["calls"] if k == "refine"], np.int64).reshape(-1, len(gmio.REFINE_FIELDS))
        assert np.array_equal(refines[:, _REFINE_DEFINED], g["refine_calls"][:, _REFINE_DEFINED])
        assert [c for k, c in r["calls"] if k == "frame"] == [int(v) for v in g["frame_errors"]]
        assert [r["final"][f] for f in gmio.FINAL_FIELDS] == [int(v) for v in g["final"]]


def test_gm_error_codes_and_cache_after_a_new_put():
    import svtme
    case = gm_cases.CASES["rotzoom_320x192"]
    eng, frames = _engine(case)
    try:
        with pytest.raises(svtme.SvtmeError):
            eng.gm_corners(999, 0)  # unknown key
        with pytest.raises(svtme.SvtmeError):
            eng.gm_corners(100, 3)  # no such level
        with pytest.raises(svtme.SvtmeError):
            eng.gm_frame_error(101, 100, 0)  # tables not handed over yet
        a = eng.gm_corners(100, 0)
        assert np.array_equal(a, gmio.oracle_corners(frames[0]))
        # the key gets other planes: the corner list kept for it must not be served again
        eng.store_put(100, 7, frames[2])
        assert np.array_equal(eng.gm_corners(100, 0), gmio.oracle_corners(frames[2]))
        eng.store_put(100, 0, frames[0])
        assert np.array_equal(eng.gm_corners(100, 0), a)
        # a picture matched against itself: every corner is its own best match and stays where it is
        c = eng.gm_correspondences(100, 100, 0)
        assert len(c) and np.array_equal(c[:, :2], c[:, 2:])
    finally:
        eng.close()


def test_a_put_of_a_key_waits_for_the_global_motion_call_that_reads_it():
    """A source picture rewritten (same key) while svtme_gm_correspondences reads its planes: the put waits for the call,
    so every result is one of the two clean outcomes, never a mixture of the old and the new planes."""
    import threading
    import svtme
    name = next(iter(gm_cases.CASES))
    case = gm_cases.CASES[name]
    eng, frames = _engine(case)
    try:
        job = case["jobs"][0]
        cur, ref, lvl = 100 + job["cur"], 100 + job["ref"], job["level"]
        old = np.ascontiguousarray(frames[job["ref"]])
        new = np.ascontiguousarray(old[::-1, ::-1])
        clean = [eng.gm_correspondences(cur, ref, lvl)]
        eng.store_put(ref, job["ref"], new)
        clean.append(eng.gm_correspondences(cur, ref, lvl))
        assert not np.array_equal(clean[0], clean[1])
        eng.store_put(ref, job["ref"], old)

        def putter():
            for i in range(30):
                eng.store_put(ref, job["ref"], new if i % 2 == 0 else old)

        t = threading.Thread(target=putter)
        t.start()
        results = []
        while t.is_alive() or len(results) < 4:
            results.append(eng.gm_correspondences(cur, ref, lvl))
        t.join()
        for r in results:
            assert any(np.array_equal(r, c) for c in clean)
        with pytest.raises(svtme.SvtmeError) as ei:  # more corners than the engine's lists hold
            eng.gm_corners(cur, lvl, max_points=svtme.GM_MAX_CORNERS + 1)
        assert ei.value.code == svtme.ERR_BAD_ARG
    finally:
        eng.close()
