#!/usr/bin/env python3
"""Host-side cost of svtme_picture_submit / wait with several pictures in flight (diagnostic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ("svt-av1_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, _p))
sys.path.insert(0, ROOT)
import bench, svtme
if os.environ.get("PROBE_TORCH") == "1":
    import torch
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
    print("torch initialised")
WL = sys.argv[1] if len(sys.argv) > 1 else "1080p_ra_m3"
wl = dict(bench.WORKLOADS[WL], name=WL)
frames, pics = bench.make_pictures(wl, 1000)
n, h, w = frames.shape
eng = svtme.Engine(w, h, store_slots=n + 1, max_in_flight=6, device=0, kernel_timing=0)
for i in range(n):
    eng.store_put(i, i, frames[i])
ctrl = svtme.derive_controls(wl["enc_mode"], w, h)
descs = [eng.make_picture(w, h, p.poc, p.temporal_layer, p.is_ref, p.refs, ctrl) for p in pics]
for rep in range(3):
    ts = []
    tickets = []
    t0 = time.perf_counter()
    for d in descs[:6]:
        a = time.perf_counter()
        tickets.append(eng.submit(d))
        ts.append((time.perf_counter() - a) * 1e6)
    t1 = time.perf_counter()
    for t in tickets:
        eng.wait(t, copy=False)
        eng.done(t)
    t2 = time.perf_counter()
    print("rep", rep, "submit us", ["%.0f" % x for x in ts], "all submits %.0f us, drain %.0f us" % ((t1 - t0) * 1e6, (t2 - t1) * 1e6))
eng.close()

eng = svtme.Engine(w, h, store_slots=n + 1, max_in_flight=6, device=0, kernel_timing=0)
for i in range(n):
    eng.store_put(i, i, frames[i])
descs = [eng.make_picture(w, h, p.poc, p.temporal_layer, p.is_ref, p.refs, ctrl) for p in pics]
if os.environ.get("PROBE_TORCH") == "late":
    import torch
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
    print("torch initialised after the engine")
for mode in ("plain", "timing", "timing+ops", "plain"):
    for rep in range(2):
        tickets = []
        t0 = time.perf_counter()
        for _ in range(4):
            for d in descs:
                if len(tickets) == 6:
                    t = tickets.pop(0)
                    eng.wait(t, copy=False)
                    if mode != "plain":
                        eng.timing(t)
                    if mode == "timing+ops":
                        eng.ops(t)
                    eng.done(t)
                tickets.append(eng.submit(d))
        while tickets:
            t = tickets.pop(0)
            eng.wait(t, copy=False)
            eng.done(t)
        dt = time.perf_counter() - t0
        print(mode, "%.0f pictures/s" % (4 * len(descs) / dt))
eng.close()
