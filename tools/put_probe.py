#!/usr/bin/env python3
"""Host-side duration of store puts and the sustained put rate, packing puts (svtme_store_put) against zero-copy puts
from page-locked buffers with the encoder's stride (svtme_store_put_pinned). Diagnostic.
    python tools/put_probe.py [width height]"""
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
pool = eng.host_alloc((n, h + 144, w + 144))
pool[:, 72:72 + h, 72:72 + w] = frames
for mode in ("copy", "pinned"):
    for rep in range(3):
        ts = []
        t0 = time.perf_counter()
        for i in range(n):
            a = time.perf_counter()
            if mode == "copy":
                eng.store_put(i, i, frames[i])
            else:
                eng.store_put_pinned(i, i, pool[i, 72:72 + h, 72:72 + w])
            ts.append((time.perf_counter() - a) * 1e6)
        t1 = time.perf_counter()
        eng.store_wait(n - 1)
        eng.store_wait(n - 2)
        t2 = time.perf_counter()
        print(mode, "rep", rep, "put calls us", ["%.0f" % t for t in ts[:6]], "all calls %.0f us, drained after %.0f us -> %.0f puts/s, %.1f GB/s" %
              ((t1 - t0) * 1e6, (t2 - t0) * 1e6, n / (t2 - t0), n * w * h / (t2 - t0) / 1e9), flush=True)
# a mini-GOP's sources in one call (svtme_store_put_many)
for rep in range(3):
    t0 = time.perf_counter()
    eng.store_put_many(list(range(n)), list(range(n)), [pool[i, 72:72 + h, 72:72 + w] for i in range(n)], pinned=True)
    t1 = time.perf_counter()
    eng.store_wait(n - 1)
    t2 = time.perf_counter()
    print("put_many pinned rep", rep, "call %.0f us, drained after %.0f us -> %.0f puts/s, %.1f GB/s" %
          ((t1 - t0) * 1e6, (t2 - t0) * 1e6, n / (t2 - t0), n * w * h / (t2 - t0) / 1e9), flush=True)
eng.host_free(pool)
eng.close()
