"""Host time of the driving loop on the launch-bound workloads (inputs resident, a mini-GOP per submit): microseconds per
picture in svtme_pictures_submit and in wait + done against the wall time. Diagnostic.
    python tools/host_split_probe.py"""
import sys, time
sys.path.insert(0, "svt-av1_amd"); sys.path.insert(0, ".")
import bench, svtme
for name in ("720p_ld_m12", "1080p_first_pass", "1080p_ra_m8"):
    wl = dict(bench.WORKLOADS[name], name=name)
    frames, pics = bench.make_pictures(wl, seed=1000)
    n, h, w = frames.shape
    fp = wl.get("fp")
    ctrl = svtme.derive_first_pass_controls(w, h, reduce_me_search=fp[1]) if fp else svtme.derive_controls(wl["enc_mode"], w, h)
    eng = svtme.Engine(w, h, store_slots=n + 1, max_in_flight=48)
    for i in range(n): eng.store_put(i, i, frames[i])
    descs = [eng.make_picture(w, h, p.poc, p.temporal_layer, p.is_ref, p.refs, ctrl, first_pass_bypass_zz=fp[0] if fp else None) for p in pics]
    t_sub = t_wait = 0.0
    tickets = []
    def step():
        global t_sub, t_wait
        while len(tickets) + len(descs) > 48:
            t0 = time.perf_counter(); tk = tickets.pop(0); eng.wait_raw(tk); eng.done(tk); t_wait += time.perf_counter() - t0
        t0 = time.perf_counter(); tickets.extend(eng.submit_batch(descs)); t_sub += time.perf_counter() - t0
    for _ in range(10): step()
    t_sub = t_wait = 0.0
    t0 = time.perf_counter()
    for _ in range(100): step()
    while tickets:
        t1 = time.perf_counter(); tk = tickets.pop(0); eng.wait_raw(tk); eng.done(tk); t_wait += time.perf_counter() - t1
    dt = time.perf_counter() - t0
    npic = 100 * len(descs)
    print("%s: %.0f pictures/s; per picture: %.1f us in submit_batch (%d pictures per call), %.1f us in wait + done, %.1f us wall"
          % (name, npic / dt, 1e6 * t_sub / npic, len(descs), 1e6 * t_wait / npic, 1e6 * dt / npic))
    eng.close()
