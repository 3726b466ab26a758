"""Timing of the TPL dispenser's source-based search at 1080p preset 8 (tpl level 4): the reference's dispenser alone (one
host core) against the same dispenser with the search on the engine, and the engine call by itself.
    python tools/tpl_probe.py [--calls-only]   (--calls-only: no child processes, for use under a profiler)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("svt-av1_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import refio  # noqa: E402
import support  # noqa: E402
import svtme  # noqa: E402
import tpl_cases  # noqa: E402
import tplio  # noqa: E402
from test_tpl_oracle import picture_of  # noqa: E402


def main():
    calls_only = "--calls-only" in sys.argv
    name, c = next(iter(tpl_cases.DIGEST_CASES.items()))
    frames, pics, rc = support.make_job(**tpl_cases.job_args(c))
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "tpl_digests.json")))[name]
    shape = want["me_shape"]
    if not calls_only:
        # tpl level 4 is the preset's own; level 1 (every intra mode, transform-domain costs, rate) on the same pictures
        for level in (c["level"], 1):
            for lib in (None, svtme.LIB_PATH):
                hdr, recs = refio.run_reference_tpl(frames, pics, c["enc_mode"], level, c["qp"], library=lib, ref_counts=rc)
                t = json.loads(hdr["stdout"].strip().splitlines()[-1])
                print("reference dispenser at tpl level %d, search %s: %.2f ms per picture (+ %.2f ms binding: engine call, filling the records)"
                      % (level, "on the engine" if lib else "on the host  ", 1e3 * t["dispenser_seconds"] / t["pictures"],
                         1e3 * t["binding_seconds"] / t["pictures"]))
    n, h, w = frames.shape
    eng = svtme.Engine(w, h, store_slots=n + 2, max_in_flight=2)
    try:
        for i in range(n):
            eng.store_put(i + 1, i, frames[i])
        tail = svtme.Tail(enable=1, enable_me_8x8=shape["enable_me_8x8"], enable_me_16x16=shape["enable_me_16x16"],
                          max_refs=shape["max_refs"], max_cand=shape["max_cand"], max_l0=shape["max_l0"],
                          input_resolution=shape["input_resolution"], gm_enabled=1, gm_use_distance_based_active_th=1)
        ctrl_me = svtme.derive_controls(c["enc_mode"], w, h, 0, 30 << 16)
        jobs, jobs_satd, jobs_all = [], [], []
        for pic, ctrl in zip(pics, want["controls"]):
            out = eng.run_picture(eng.make_picture(w, h, pic.poc, pic.temporal_layer, pic.is_ref, pic.refs, ctrl_me, tail=tail,
                                                   key_of=lambda poc: poc + 1))
            me = dict(total_me_candidate_index=out["total_me_candidate_index"],
                      me_candidate_array=out["me_candidate_array"][:, :85 * shape["max_cand"]],
                      me_mv_array=out["me_mv_array"][:, :85 * shape["max_refs"]], enable_me_16x16=shape["enable_me_16x16"],
                      max_cand=shape["max_cand"], max_refs=shape["max_refs"], max_l0=shape["max_l0"])
            jobs.append(picture_of(frames, pic, ctrl, me, quant=want["quant"]))  # with the chosen candidate's quantisation error
            satd = dict(ctrl, use_pred_sad_in_intra_search=0, use_pred_sad_in_inter_search=0, pf_shape=0)
            jobs_satd.append(picture_of(frames, pic, satd, me, quant=want["quant"]))  # tpl level 3's costs on the same pictures
            every = dict(satd, tpl_opt_flag=0, disable_intra_pred=0)
            jobs_all.append(picture_of(frames, pic, every, me, quant=want["quant"]))  # tpl level 1: thirteen intra modes, rate
        for what, js in (("SAD costs (tpl level 4) + quantisation error of the choice", jobs),
                         ("transform-domain costs (tpl level 3) + quantisation error", jobs_satd),
                         ("thirteen intra modes, transform-domain costs, quantisation error and rate (tpl level 1)", jobs_all)):
            for p, keep, refs in js:
                eng.tpl_search(p)
            reps = 50
            t0 = time.perf_counter()
            for _ in range(reps):
                for p, keep, refs in js:
                    eng.tpl_search(p)
            dt = (time.perf_counter() - t0) / (reps * len(js))
            print("svtme_tpl_search, 1080p, %d blocks, %s: %.3f ms per call (ME results in: %.2f MB, block records out: %.2f MB)"
                  % (8160, what, 1e3 * dt, 510 * 85 * (1 + shape["max_cand"] + 4 * shape["max_refs"]) / 1e6, 510 * 21 * 40 / 1e6))
        # the reconstruction path on the engine's own search results (the reference's own, one host core: the dispenser's time
        # with the search on the engine, above)
        for what, js in (("tpl level 4 (DC from the reconstructed neighbours)", jobs), ("tpl level 1 (the chosen one of thirteen intra modes, rate)", jobs_all)):
            found = [eng.tpl_search(p) for p, keep, refs in js]
            n_intra = sum(int(((b["valid"] == 1) & (b["best_is_inter"] == 0)).sum()) for b in found)
            for (p, keep, refs), b in zip(js, found):
                eng.tpl_recon(p, b)
            for want_picture in (True, False):
                reps = 10
                t0 = time.perf_counter()
                for _ in range(reps):
                    for (p, keep, refs), b in zip(js, found):
                        eng.tpl_recon(p, b, want_picture=want_picture)
                dt = (time.perf_counter() - t0) / (reps * len(js))
                print("svtme_tpl_recon, 1080p, 8160 blocks (%.0f intra per picture), %s, %s: %.3f ms per call"
                      % (n_intra / len(js), what, "statistics and the reconstructed picture back" if want_picture else "statistics back", 1e3 * dt))
    finally:
        eng.close()


if __name__ == "__main__":
    main()
