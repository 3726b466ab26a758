"""Debug helper: temporal-filter golden case through the engine, prints where the minima differ."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("svt-av1_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import support, svtme
import importlib.util
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
MOD = importlib.util.module_from_spec(spec); spec.loader.exec_module(MOD)
name = sys.argv[1] if len(sys.argv) > 1 else "tf0_320x192"
c = MOD.TF_CASES[name]
g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
frames, pics, _ = MOD.tf_job(c)
outs = support.run_engine_tf_job(frames, pics, c["level"], c["exit_th"])
for i, (p, out) in enumerate(zip(pics, outs)):
    want = MOD.tf_canonical(g["b64"][i])
    got = support.tf_fields(out)
    for f in got:
        ne = np.argwhere(np.asarray(got[f]) != want[f])
        if len(ne):
            print("pair", i, "neighbour", p.refs[0][0], f, len(ne), "mismatches; first", ne[:12].tolist())
            for idx in ne[:6]:
                idx = tuple(idx)
                print("   ", idx, "engine", got[f][idx], hex(int(got[f][idx])), "want", want[f][idx], hex(int(want[f][idx])))
    sr = out["search_results"]
    print("pair", i, "centres", list(zip(sr["hme_sc_x"][:, 0, 0].tolist(), sr["hme_sc_y"][:, 0, 0].tolist())))
