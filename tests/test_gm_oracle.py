"""The CPU restatement of the global-motion stages (oracle/gm_oracle.c) against the fixtures the real reference produced
(tests/golden/gm_*.npz): FAST-9 corners, correspondences, and - with the warp filter and error tables taken from the
reference build at run time, they are not kept in this repository - every warp / frame error call of the recorded runs and
every refinement (svt_av1_refine_integerized_param as a whole: the recorded runs' and tests/golden/gm_refine_*.npz)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ("svt-av1_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, _p))
import gm_cases  # noqa: E402
import gmio  # noqa: E402

LIGHT = [n for n in gm_cases.CASES if n not in gm_cases.HEAVY]


@pytest.mark.parametrize("name", LIGHT)
def test_corners_and_correspondences_match_the_reference(name):
    case = gm_cases.CASES[name]
    frames = gm_cases.frames_of(case)
    for job, g in zip(case["jobs"], gm_cases.golden_of(name)):
        frm = gmio.plane_of(frames[job["cur"]], job["level"])
        ref = gmio.plane_of(frames[job["ref"]], job["level"])
        cf, cr = gmio.oracle_corners(frm), gmio.oracle_corners(ref)
        assert np.array_equal(cf, g["corners_frm"]) and np.array_equal(cr, g["corners_ref"])
        assert np.array_equal(gmio.oracle_correspondences(frm, cf, ref, cr), g["correspondences"])


def test_corner_list_is_cut_in_raster_order():
    frames = gm_cases.frames_of(gm_cases.CASES["noise_640x360"])
    full = gmio.oracle_corners(frames[0], max_points=1 << 20)
    assert len(full) > gmio.MAX_CORNERS
    assert np.array_equal(gmio.oracle_corners(frames[0]), full[:gmio.MAX_CORNERS])
    order = full[:, 1].astype(np.int64) * 4096 + full[:, 0]
    assert np.all(np.diff(order) > 0)


@pytest.mark.skipif(not gmio.have_reference(), reason="the tables come from oracle/_ref/ref_gm, built where the reference sources are")
@pytest.mark.parametrize("name", LIGHT)
def test_every_recorded_error_call_is_reproduced(name):
    tables = gmio.reference_tables()
    case = gm_cases.CASES[name]
    frames = gm_cases.frames_of(case)
    for job, g in zip(case["jobs"], gm_cases.golden_of(name)):
        frm = gmio.plane_of(frames[job["cur"]], job["level"])
        ref = gmio.plane_of(frames[job["ref"]], job["level"])
        for c in g["warp_calls"]:
            c = dict(zip(gmio.WARP_FIELDS, (int(v) for v in c)))
            got = gmio.oracle_warp_error(tables, [c["m%d" % i] for i in range(6)], [c["alpha"], c["beta"], c["gamma"], c["delta"]], ref, frm,
                                         c["best_error"])
            assert got == c["error"], (name, c, got)
        for e in g["frame_errors"]:
            assert gmio.oracle_frame_error(tables, ref, frm) == int(e)


def _check_refine(tables, rec, n_warp_calls, ref, frm):
    c = dict(zip(gmio.REFINE_FIELDS, (int(v) for v in rec)))
    mat, shear, wmtype, err, evals = gmio.oracle_refine(tables, [c["i%d" % i] for i in range(6)], [c["i_alpha"], c["i_beta"], c["i_gamma"], c["i_delta"]],
                                                        c["i_wmtype"], c["force_wmtype"], ref, frm, c["steps"], c["best_frame_error"])
    assert (mat, shear, wmtype, err) == ([c["o%d" % i] for i in range(6)], [c["o_alpha"], c["o_beta"], c["o_gamma"], c["o_delta"]], c["o_wmtype"],
                                         c["error"]), c
    assert evals == n_warp_calls


@pytest.mark.skipif(not gmio.have_reference(), reason="the tables come from oracle/_ref/ref_gm, built where the reference sources are")
@pytest.mark.parametrize("name", LIGHT)
def test_every_recorded_refinement_is_reproduced(name):
    tables = gmio.reference_tables()
    case = gm_cases.CASES[name]
    frames = gm_cases.frames_of(case)
    for job, g in zip(case["jobs"], gm_cases.golden_of(name)):
        frm = gmio.plane_of(frames[job["cur"]], job["level"])
        ref = gmio.plane_of(frames[job["ref"]], job["level"])
        assert len(g["refine_calls"]) <= 1  # (one RANSAC motion per model; the warp calls of a job are its refinement's)
        for rec in g["refine_calls"]:
            _check_refine(tables, rec, len(g["warp_calls"]), ref, frm)


@pytest.mark.skipif(not gmio.have_reference(), reason="the tables come from oracle/_ref/ref_gm, built where the reference sources are")
@pytest.mark.parametrize("name", list(gm_cases.REFINE_CASES))
def test_single_refinements_of_every_model_type_are_reproduced(name):
    tables = gmio.reference_tables()
    c = gm_cases.REFINE_CASES[name]
    frames = gm_cases.frames_of(gm_cases.CASES[name])
    for call, (rec, warps) in zip(c["calls"], gm_cases.refine_golden_of(name)):
        frm = gmio.plane_of(frames[c["cur"]], call["level"])
        ref = gmio.plane_of(frames[c["ref"]], call["level"])
        _check_refine(tables, rec, len(warps), ref, frm)


@pytest.mark.skipif(not gmio.have_reference(), reason="needs oracle/_ref/ref_gm")
def test_fixtures_are_what_the_reference_gives_today():
    name = "rotzoom_320x192"
    case = gm_cases.CASES[name]
    recs = gmio.run_reference(gm_cases.frames_of(case), case["jobs"])
    for r, g in zip(recs, gm_cases.golden_of(name)):
        assert np.array_equal(r["corners_frm"], g["corners_frm"]) and np.array_equal(r["correspondences"], g["correspondences"])
        assert [r["final"][f] for f in gmio.FINAL_FIELDS] == [int(v) for v in g["final"]]


@pytest.mark.skipif(not gmio.have_reference(), reason="needs oracle/_ref/ref_gm")
def test_reference_avx2_kernels_give_the_same_record():
    """The fixtures come from the reference's C kernels; its AVX2 warp filter and frame error (what an AVX2
    machine runs for them) leave the same corners, correspondences, error values and final models."""
    for name in ("rotzoom_320x192", "affine_392x264"):
        case = gm_cases.CASES[name]
        recs = gmio.run_reference(gm_cases.frames_of(case), case["jobs"], use_avx2=True)
        for r, g in zip(recs, gm_cases.golden_of(name)):
            assert np.array_equal(r["correspondences"], g["correspondences"])
            warps = np.array([[c[f] for f in gmio.WARP_FIELDS] for k, c in r["calls"] if k == "warp"], np.int64).reshape(-1, len(gmio.WARP_FIELDS))
            assert np.array_equal(warps, g["warp_calls"])
            assert [r["final"][f] for f in gmio.FINAL_FIELDS] == [int(v) for v in g["final"]]
