"""The drop-in seam with the reference's own code on both sides of it: oracle/_ref/ref_me (the real reference, compiled
from /root/reference by oracle/Makefile where that exists; the binary travels) runs in REF_MODE_SHIM, i.e. it derives the
controls with the reference's signal_derivation_me_kernel_oq, hands the picture to the HIP engine through the C ABI of
include/svtme.h, unpacks every 64x64 block with svtme_unpack_b64 straight into a real MeContext (search_results by a
byte copy: the two structs are statically asserted to have one layout) and runs the reference's own me_prune_ref,
construct_me_candidate_array[_mrp_off], compute_distortion and perform_gm_detection on it. What that leaves in
me_results / MeContext / the picture statistics must equal the dumps of the unmodified reference (tests/golden/*.npz)."""
import os

import numpy as np
import pytest

import refio
import support
import svtme

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")


def _golden_mod():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


MOD = _golden_mod()


def _need_reference_binary():
    assert refio.have_reference(), "oracle/_ref/ref_me is missing: build it with `make -C oracle ref` where /root/reference exists"


@pytest.mark.parametrize("name", sorted(MOD.CASES))
def test_engine_plus_the_reference_host_tail_equals_the_reference(name):
    _need_reference_binary()
    c = MOD.CASES[name]
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    frames, pics, rc = support.make_job(c["width"], c["height"], c["n_frames"], c["enc_mode"], c["seed"],
                                        c.get("structure", "ra"), c.get("kind", "synth"), c.get("motion", (3, -2)))
    kw = dict(ref_counts=rc, sc_class1=c.get("sc_class1", 0), frame_rate=c.get("frame_rate", 30 << 16), gm_enabled=1, gm_dist_th=1,
              restricted_motion_vector=c.get("restricted", 0), tiles=c.get("tiles", (0, 0)))
    hdr, recs = refio.run_reference_shim(frames, pics, c["enc_mode"], svtme.LIB_PATH, **kw)
    assert hdr["mode"] == refio.MODE_SHIM and len(recs) == len(pics)
    # the controls the shim copied out of the reference's MeContext are the ones the golden run used
    for (pd, _), gpd in zip(recs, g["pic"]):
        assert pd.tobytes() == gpd.tobytes(), name
    pruning = [bool(pd["ctrl"]["enable_me_hme_ref_pruning"]) for pd in g["pic"]]
    msgs = support.tail_mismatches(recs, g["b64"], pics, pruning)
    # plus the hand-over fields that do not depend on a thread's leftovers
    for i, ((_, bd), p) in enumerate(zip(recs, pics)):
        for f in support.HANDOVER_FIELDS:
            if not np.array_equal(bd[f], g["b64"][i][f]):
                msgs.append("picture %d (poc %d): %s" % (i, p.poc, f))
    assert not msgs, "\n".join(msgs[:12])


@pytest.mark.parametrize("name", sorted(MOD.FP_CASES))
def test_first_pass_engine_plus_the_reference_host_tail_equals_the_reference(name):
    _need_reference_binary()
    c = MOD.FP_CASES[name]
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    frames, pics, kw = MOD.fp_job(c)
    _, recs = refio.run_reference_shim(frames, pics, 8, svtme.LIB_PATH, **kw)
    for i, ((pd, bd), p) in enumerate(zip(recs, pics)):
        want = g["b64"][i]
        assert pd.tobytes() == g["pic"][i].tobytes()
        nr = len(p.refs[0])
        for f in ("best_sad", "best_mv"):
            assert np.array_equal(bd[f][:, 0, :nr], want[f][:, 0, :nr]), (name, p.poc, f)
        for f in support.HANDOVER_FIELDS + support.TAIL_FIELDS:
            assert np.array_equal(bd[f], want[f]), (name, p.poc, f)


@pytest.mark.parametrize("name", sorted(MOD.TF_CASES))
def test_temporal_filter_engine_results_in_a_real_mecontext_equal_the_reference(name):
    _need_reference_binary()
    c = MOD.TF_CASES[name]
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    frames, pics, kw = MOD.tf_job(c)
    _, recs = refio.run_reference_shim(frames, pics, 0, svtme.LIB_PATH, **kw)
    for i, ((_, bd), p) in enumerate(zip(recs, pics)):
        want, got = MOD.tf_canonical(g["b64"][i]), MOD.tf_canonical(bd)
        for f in want:
            assert np.array_equal(got[f], want[f]), (name, "neighbour %d" % p.refs[0][0], f)
