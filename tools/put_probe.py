#!/usr/bin/env python3
"""Host-side duration of svtme_store_put calls and the sustained put rate (diagnostic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ("svt-av1_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np
import svtme
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
n = 12
frames = np.random.default_rng(1).integers(0, 255, (n, h, w), dtype=np.uint8)
eng = svtme.Engine(w, h, store_slots=n + 1, max_in_flight=2)
for rep in range(3):
    ts = []
    t0 = time.perf_counter()
    for i in range(n):
        a = time.perf_counter()
        eng.store_put(i, i, frames[i])
        ts.append((time.perf_counter() - a) * 1e6)
    t1 = time.perf_counter()
    eng.store_read(n - 1, 0, (h, w))  # waits for the last put
    t2 = time.perf_counter()
    print("rep", rep, "put calls us", ["%.0f" % t for t in ts[:6]], "all calls %.0f us, drained after %.0f us -> %.0f puts/s" %
          ((t1 - t0) * 1e6, (t2 - t0) * 1e6, n / (t2 - t0)))
eng.close()
