"""Generates the global-motion fixtures tests/golden/gm_*.npz by running the REAL reference's compute_global_motion
(oracle/_ref/ref_gm, built from /root/reference by oracle/Makefile) on seeded synthetic picture pairs. Only runs where the
reference sources exist. The fixtures are data: inputs are regenerated from the seeds (tests/gm_cases.py); outputs are the
corners, the correspondences, every warp / frame error call and every refinement with its arguments and result, and the
final model. The warp
filter and error tables are NOT stored: tests that need them take them from the reference build at run time.

    python tests/golden/make_gm_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in ("svt-av1_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import gmio  # noqa: E402
import gm_cases  # noqa: E402


def main():
    for name, case in gm_cases.CASES.items():
        frames = gm_cases.frames_of(case)
        recs = gmio.run_reference(frames, case["jobs"])
        out = {}
        for i, r in enumerate(recs):
            out["j%d_corners_frm" % i] = r["corners_frm"].astype(np.int32)
            out["j%d_corners_ref" % i] = r["corners_ref"].astype(np.int32)
            out["j%d_correspondences" % i] = r["correspondences"].astype(np.int32)
            warps = [[c[f] for f in gmio.WARP_FIELDS] for k, c in r["calls"] if k == "warp"]
            out["j%d_warp_calls" % i] = np.array(warps, np.int64).reshape(-1, len(gmio.WARP_FIELDS))
            out["j%d_frame_errors" % i] = np.array([c for k, c in r["calls"] if k == "frame"], np.int64)
            refines = [[c[f] for f in gmio.REFINE_FIELDS] for k, c in r["calls"] if k == "refine"]
            out["j%d_refine_calls" % i] = np.array(refines, np.int64).reshape(-1, len(gmio.REFINE_FIELDS))
            out["j%d_final" % i] = np.array([r["final"][f] for f in gmio.FINAL_FIELDS], np.int64)
            out["j%d_call_kinds" % i] = np.array([{"corners": 1, "corr": 2, "warp": 3, "frame": 4, "refine": 5}[k] for k, _ in r["calls"]], np.uint8)
            print(name, i, "corners", len(r["corners_frm"]), len(r["corners_ref"]), "corr", len(r["correspondences"]), "warp calls", len(warps),
                  "final wmtype", r["final"]["wmtype"])
        np.savez_compressed(os.path.join(HERE, "gm_%s.npz" % name), **out)
    for name in gm_cases.REFINE_CASES:
        recs = gmio.run_reference(gm_cases.frames_of(gm_cases.CASES[name]), gm_cases.refine_jobs(name))
        out = {}
        for i, r in enumerate(recs):
            refines = [[c[f] for f in gmio.REFINE_FIELDS] for k, c in r["calls"] if k == "refine"]
            warps = [[c[f] for f in gmio.WARP_FIELDS] for k, c in r["calls"] if k == "warp"]
            assert len(refines) == 1
            out["c%d_refine" % i] = np.array(refines[0], np.int64)
            out["c%d_warp_calls" % i] = np.array(warps, np.int64).reshape(-1, len(gmio.WARP_FIELDS))
            print("refine", name, i, "evaluations", len(warps), dict(zip(gmio.REFINE_FIELDS, refines[0])))
        np.savez_compressed(os.path.join(HERE, "gm_refine_%s.npz" % name), **out)


if __name__ == "__main__":
    main()
