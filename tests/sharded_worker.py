"""One rank of the multi-process scheduler test: runs its share of a small job on the oracle-backed stand-in engine
and gathers all results on every rank over gloo. Started by test_scheduler.py, one process per rank."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "svt-av1_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import hashlib  # noqa: E402

import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402

import support  # noqa: E402
from svtme import scheduler  # noqa: E402


def digest(out):
    h = hashlib.sha256()
    searched = np.asarray(out["do_ref"])[..., None] == 1  # pruned references keep whatever the ME thread held before
    for f in ("best_sad", "best_mv", "zz_sad", "sr_divisor", "hme_sc_x", "hme_sc_y", "do_ref"):
        a = np.asarray(out[f])
        if f in ("best_sad", "best_mv"):
            a = np.where(searched, a, 0)
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out_path, enc_mode = sys.argv[1], int(sys.argv[2])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames, pics, rc = support.make_job(128, 64, 9, enc_mode, seed=5)
    eng = support.OracleEngine(enc_mode, rc, store_slots=8)
    mine = scheduler.run_sharded(eng, {i: frames[i] for i in range(len(frames))}, pics, None, rank, world, in_flight=2)
    everything = scheduler.gather({i: digest(o) for i, o in mine.items()}, world)
    stats = [None] * world
    dist.all_gather_object(stats, {"rank": rank, "ran": sorted(mine), "puts": eng.puts, "releases": eng.releases,
                                   "left_resident": len(eng.store), "max_resident": eng.max_resident})
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump({"digests": {str(k): v for k, v in everything.items()}, "stats": stats}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
