#!/usr/bin/env python3
"""Throughput of the submit / wait loop for 1..3 search streams, with or without torch's HIP runtime loaded first
(PROBE_TORCH=1): diagnostic for the host-side limits of light workloads."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ("svt-av1_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, _p))
sys.path.insert(0, ROOT)
if os.environ.get("PROBE_TORCH") == "1":
    import torch
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
    print("torch initialised")
import bench, svtme
WL = sys.argv[1] if len(sys.argv) > 1 else "1080p_ra_m8"
wl = dict(bench.WORKLOADS[WL], name=WL)
frames, pics = bench.make_pictures(wl, 1000)
n, h, w = frames.shape
ctrl = svtme.derive_controls(wl["enc_mode"], w, h)
REPS = 20 if wl["enc_mode"] >= 8 else 6
COMBOS = [(3, 6), (3, 3), (2, 6), (1, 6), (3, 6)] if len(sys.argv) < 3 else [tuple(int(v) for v in c.split(",")) for c in sys.argv[2:]]
for streams, infl in COMBOS:
    if True:
        eng = svtme.Engine(w, h, store_slots=n + 1, max_in_flight=infl, device=0, search_streams=streams, kernel_timing=0)
        for i in range(n):
            eng.store_put(i, i, frames[i])
        descs = [eng.make_picture(w, h, p.poc, p.temporal_layer, p.is_ref, p.refs, ctrl) for p in pics]
        res = []
        for rep in range(4):
            tickets = []
            t0 = time.perf_counter()
            for _ in range(REPS):
                for d in descs:
                    if len(tickets) == infl:
                        t = tickets.pop(0)
                        eng.wait(t, copy=False)
                        eng.done(t)
                    tickets.append(eng.submit(d))
            while tickets:
                t = tickets.pop(0)
                eng.wait(t, copy=False)
                eng.done(t)
            res.append(REPS * len(descs) / (time.perf_counter() - t0))
        print("streams", streams, "in flight", infl, ["%.0f" % r for r in res], flush=True)
        eng.close()
