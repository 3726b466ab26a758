"""Ad-hoc GPU check: engine vs oracle on a few small jobs. Usage: python tools/gpu_quick.py [enc_mode ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("svt-av1_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import support

modes = [int(a) for a in sys.argv[1:]] or [8]
bad = 0
for m in modes:
    t = time.time()
    if m == modes[0]:
        frames, pics, rc = support.make_job(320, 192, 17, m, 3 + m)
        recs = support.oracle_handover(frames, pics, m, rc)
        outs = support.run_engine_job(frames, pics, m)
        o, b = outs[1], recs[1]
        print("sample poc", pics[1].poc, "engine sad", o["best_sad"][3, 0, 0, :4], "oracle", b["best_sad"][3, 0, 0, :4],
              "mv", [hex(v) for v in o["best_mv"][3, 0, 0, :3]], [hex(v) for v in b["best_mv"][3, 0, 0, :3]],
              "sc", o["search_results"]["hme_sc_x"][3], b["hme_sc_x"][3], "timing", o["timing_ms"], flush=True)
    msgs = support.engine_vs_oracle(320, 192, 17, m, seed=3 + m)
    print("M%d 320x192: %d mismatch messages (%.1fs)" % (m, len(msgs), time.time() - t), flush=True)
    for s in msgs[:12]:
        print("   ", s)
    bad += len(msgs)
print("TOTAL", bad)
sys.exit(1 if bad else 0)
