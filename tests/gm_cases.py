"""The global-motion test jobs: seeded picture pairs and the compute_global_motion runs the fixtures tests/golden/gm_*.npz
hold (tests/golden/make_gm_golden.py wrote them from the real reference). A job: cur / ref frame index, downsample level
(gm_ctrls.downsample_level: 0 full, 1 quarter, 2 sixteenth plane), rotzoom_only, refinement steps."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ("svt-av1_amd", "oracle"):
    if os.path.join(ROOT, _p) not in sys.path:
        sys.path.insert(0, os.path.join(ROOT, _p))
from svtme import synth  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")

CASES = {
    # rotation + zoom + shift that grow with the frame index
    "rotzoom_320x192": dict(kind="warped", width=320, height=192, n_frames=3, seed=2,
                            jobs=[dict(cur=1, ref=0, level=0, rotzoom_only=1, steps=5), dict(cur=2, ref=0, level=0, rotzoom_only=0, steps=5),
                                  dict(cur=1, ref=0, level=1, rotzoom_only=1, steps=5), dict(cur=0, ref=2, level=0, rotzoom_only=1, steps=1)]),
    # shear + anisotropic zoom: not a rotation-zoom, so the affine model gets its turn; width and height not multiples of 32
    "affine_392x264": dict(kind="warped", width=392, height=264, n_frames=2, seed=5, rot=(60, 170), zoom=(90, 260), shift=(3, 2),
                           jobs=[dict(cur=1, ref=0, level=0, rotzoom_only=0, steps=5), dict(cur=1, ref=0, level=1, rotzoom_only=0, steps=3)]),
    # translation and independently moving rectangles (the motion-estimation test content); the sixteenth plane too
    "translation_512x256": dict(kind="synth", width=512, height=256, n_frames=2, seed=7, motion=(5, -3),
                                jobs=[dict(cur=1, ref=0, level=0, rotzoom_only=1, steps=5), dict(cur=1, ref=0, level=2, rotzoom_only=1, steps=5)]),
    # i.i.d. noise: more corners than the list holds (cut at 4096 in raster order), no match passes the threshold
    "noise_640x360": dict(kind="noise", width=640, height=360, n_frames=2, seed=9, jobs=[dict(cur=1, ref=0, level=0, rotzoom_only=1, steps=5)]),
    # no corners at all
    "flat_128x128": dict(kind="flat", width=128, height=128, n_frames=2, seed=1, jobs=[dict(cur=1, ref=0, level=0, rotzoom_only=1, steps=5)]),
    # the benchmark's picture size, full and quarter plane
    "rotzoom_1920x1080": dict(kind="warped", width=1920, height=1080, n_frames=2, seed=3, rot=(25, 0), zoom=(60, 0), shift=(4, -2),
                              jobs=[dict(cur=1, ref=0, level=0, rotzoom_only=1, steps=5), dict(cur=1, ref=0, level=1, rotzoom_only=1, steps=5)]),
}
# fixtures too large for the CPU restatement to replay call by call in the CPU suite (the GPU tests replay them in full)
HEAVY = ("rotzoom_1920x1080",)


def frames_of(case):
    w, h, n, seed = case["width"], case["height"], case["n_frames"], case["seed"]
    if case["kind"] == "warped":
        return synth.warped_sequence(w, h, n, seed=seed, rot=case.get("rot", (40, 0)), zoom=case.get("zoom", (200, 0)),
                                     shift=case.get("shift", (2, -1)))
    if case["kind"] == "synth":
        return synth.synth_sequence(w, h, n, seed=seed, motion=case.get("motion", (3, -2)))
    return synth.extreme_sequence(w, h, n, case["kind"], seed=seed)


def golden_of(name):
    """-> list of per-job dicts (corners_frm, corners_ref, correspondences, warp_calls, frame_errors, final, call_kinds)."""
    z = np.load(os.path.join(GOLDEN, "gm_%s.npz" % name))
    out = []
    for i in range(len(CASES[name]["jobs"])):
        out.append({k[len("j%d_" % i):]: z[k] for k in z.files if k.startswith("j%d_" % i)})
    return out
