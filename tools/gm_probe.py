#!/usr/bin/env python3
"""Durations of the global-motion calls (svtme_gm_*) on one picture pair, and of the whole compute_global_motion of the
reference with and without the engine behind its seam (oracle/_ref/ref_gm). Diagnostic.
    python tools/gm_probe.py [case of tests/gm_cases.py] [job index] [--calls-only --tables=FILE]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ("svt-av1_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, _p))
import gm_cases
import gmio
import svtme

import numpy as np

args = [a for a in sys.argv[1:] if a != "--calls-only" and not a.startswith("--tables=")]
# for runs under a profiler, which must not start child processes: --calls-only --tables=<file written by `ref_gm tables <file>`>
calls_only = "--calls-only" in sys.argv
tables_file = ([a[9:] for a in sys.argv[1:] if a.startswith("--tables=")] or [None])[0]
name = args[0] if len(args) > 0 else "rotzoom_1920x1080"
ji = int(args[1]) if len(args) > 1 else 0
case = gm_cases.CASES[name]
job = case["jobs"][ji]
frames = gm_cases.frames_of(case)
g = gm_cases.golden_of(name)[ji]
lvl = job["level"]
eng = svtme.Engine(case["width"], case["height"], store_slots=4, max_in_flight=1)
eng.store_put(1, 1, frames[job["cur"]])
eng.store_put(0, 0, frames[job["ref"]])
have_ref = gmio.have_reference() and (tables_file or not calls_only)
if tables_file:
    raw = open(tables_file, "rb").read()
    eng.gm_set_tables(np.frombuffer(raw[:193 * 8 * 2], np.int16), np.frombuffer(raw[193 * 8 * 2:], np.int32))
elif have_ref:
    eng.gm_set_tables(*gmio.reference_tables())


def timed(label, f, reps=20):
    f()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    dt = (time.perf_counter() - t0) / reps
    print("%-44s %9.1f us per call" % (label, dt * 1e6), flush=True)
    return dt


def corners_uncached():
    eng.store_put(1, 1, frames[job["cur"]])  # a new put drops the cached list
    eng.gm_corners(1, lvl)


print("%s job %d: level %d, plane %dx%d, %d / %d corners, %d correspondences, %d warp calls" %
      (name, ji, lvl, case["width"] >> lvl, case["height"] >> lvl, len(g["corners_frm"]), len(g["corners_ref"]), len(g["correspondences"]),
       len(g["warp_calls"])))
timed("store_put + corners (score, nonmax, compact)", corners_uncached)
timed("store_put alone", lambda: (eng.store_put(1, 1, frames[job["cur"]]), eng.store_wait(1)))
timed("corners (cached list, copy only)", lambda: eng.gm_corners(1, lvl))
timed("correspondences (both lists cached)", lambda: eng.gm_correspondences(1, 0, lvl))
if have_ref and len(g["warp_calls"]):
    c = dict(zip(gmio.WARP_FIELDS, (int(v) for v in g["warp_calls"][0])))
    mat, shear = [c["m%d" % i] for i in range(6)], [c["alpha"], c["beta"], c["gamma"], c["delta"]]
    timed("warp error, whole picture", lambda: eng.gm_warp_error(1, 0, lvl, mat, shear))
    timed("warp error, stops at the first block", lambda: eng.gm_warp_error(1, 0, lvl, mat, shear, 0))
    timed("frame error", lambda: eng.gm_frame_error(1, 0, lvl))
    for rec in g["refine_calls"]:
        r = dict(zip(gmio.REFINE_FIELDS, (int(v) for v in rec)))
        dt = timed("refinement loop, one call", lambda: eng.gm_refine(1, 0, lvl, [r["i%d" % i] for i in range(6)], r["force_wmtype"], r["steps"]), reps=10)
        n = eng.gm_refine(1, 0, lvl, [r["i%d" % i] for i in range(6)], r["force_wmtype"], r["steps"])[4]
        print("    %d evaluations, %.1f us each" % (n, dt * 1e6 / n))
eng.close()
if have_ref and not calls_only:
    for lib, avx2, label in ((None, False, "reference alone (C kernels)"), (None, True, "reference alone (AVX2 warp and frame error)"),
                             (svtme.LIB_PATH, False, "reference with the engine behind its seam"),
                             (svtme.LIB_PATH, "refine", "the same, refinement loop on the engine")):
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            gmio.run_reference(frames, [job] * 4, library=lib, use_avx2=avx2 is True, engine_refine=avx2 == "refine")
            best = min(best, time.perf_counter() - t0)
        t0 = time.perf_counter()
        gmio.run_reference(frames, [], library=lib, use_avx2=avx2 is True, engine_refine=avx2 == "refine")  # start-up, picture store, engine creation: not global motion
        base = time.perf_counter() - t0
        print("%-44s %9.1f ms per compute_global_motion (4 runs %.0f ms, set-up %.0f ms)" % (label, (best - base) / 4 * 1e3, best * 1e3, base * 1e3),
              flush=True)
