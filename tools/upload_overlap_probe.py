"""Where the host thread's time goes when every source is put inside the search loop (bench.py --with-upload): seconds in
the put calls, in the submit calls and in the waits, per picture, and the same loop with the puts made by a second
thread one step ahead (as the encoder's picture-analysis threads would).
    python tools/upload_overlap_probe.py [workload]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("svt-av1_amd", "oracle", "tests", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import bench  # noqa: E402
import svtme  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "1080p_ra_m8"
    wl = dict(bench.WORKLOADS[name], name=name)
    frames, pics = bench.make_pictures(wl, seed=1000)
    n, h, w = frames.shape
    ctrl = svtme.derive_controls(wl["enc_mode"], w, h)
    batch, in_flight, steps = 4, 12, 40
    eng = svtme.Engine(w, h, store_slots=2 * n + 1, max_in_flight=in_flight)
    pool = eng.host_alloc((n, h + 144, w + 144))
    pool[:, 72:72 + h, 72:72 + w] = frames
    descs = [[eng.make_picture(w, h, p.poc, p.temporal_layer, p.is_ref, p.refs, ctrl, key_of=lambda poc, ks=ks: 1000 * ks + poc)
              for p in pics] for ks in range(2)]
    sources = [list(dict.fromkeys([p.poc] + [r for lst in p.refs for r in lst])) for p in pics]
    t = {"put": 0.0, "submit": 0.0, "wait": 0.0}

    def put(k, ks):
        eng.store_put_pinned(1000 * ks + k, k, pool[k, 72:72 + h, 72:72 + w])

    def run(k_steps, putter):
        """putter: None = puts inline, right before the first picture that needs the source; else a semaphore pair with
        the thread that puts a whole step's sources ahead of its searches."""
        tickets, group = [], []

        def flush():
            while len(tickets) + len(group) > in_flight:
                t0 = time.perf_counter()
                tk = tickets.pop(0)
                eng.wait_raw(tk)
                eng.done(tk)
                t["wait"] += time.perf_counter() - t0
            t0 = time.perf_counter()
            tickets.extend(eng.submit_batch(group))
            t["submit"] += time.perf_counter() - t0
            group.clear()

        for step in range(k_steps):
            ks = step % 2
            if putter:
                putter[0].acquire()  # this step's sources have been put
            seen = set()
            for i, d in enumerate(descs[ks]):
                if not putter:
                    t0 = time.perf_counter()
                    for k in sources[i]:
                        if k not in seen:
                            seen.add(k)
                            put(k, ks)
                    t["put"] += time.perf_counter() - t0
                group.append(d)
                if len(group) == batch:
                    flush()
            if putter:
                # the step before this one used the other key set: its pictures must be done before that set is put again
                while tickets and len(tickets) > len(descs[ks]):
                    tk = tickets.pop(0)
                    eng.wait_raw(tk)
                    eng.done(tk)
                putter[1].release()
        if group:
            flush()
        while tickets:
            tk = tickets.pop(0)
            eng.wait_raw(tk)
            eng.done(tk)

    try:
        run(5, None)
        for k in t:
            t[k] = 0.0
        t0 = time.perf_counter()
        run(steps, None)
        dt = time.perf_counter() - t0
        npic = steps * len(pics)
        print("%s, puts inline: %.0f pictures/s; host time per picture: put %.1f us, submit %.1f us, wait + done %.1f us, total %.1f us"
              % (name, npic / dt, 1e6 * t["put"] / npic, 1e6 * t["submit"] / npic, 1e6 * t["wait"] / npic, 1e6 * dt / npic))

        ready, free = threading.Semaphore(0), threading.Semaphore(1)

        def putter_main(k_steps):
            for step in range(k_steps):
                free.acquire()
                for k in range(n):
                    put(k, step % 2)
                ready.release()

        th = threading.Thread(target=putter_main, args=(steps,))
        t0 = time.perf_counter()
        th.start()
        run(steps, (ready, free))
        th.join()
        dt = time.perf_counter() - t0
        print("%s, puts by a second thread one step ahead: %.0f pictures/s" % (name, npic / dt))
    finally:
        eng.close()


if __name__ == "__main__":
    main()
