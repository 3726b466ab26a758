"""Host-side picture sharding (svtme.scheduler): ownership, store residency, and a 2-rank gloo run on CPU whose
gathered results equal the single-process oracle's."""
import json
import os
import socket
import subprocess
import sys

import support
from svtme import gop, scheduler

HERE = os.path.dirname(os.path.abspath(__file__))


def test_every_picture_has_exactly_one_owner():
    pics = gop.random_access(33, 3)
    for world in (1, 2, 3, 4, 8):
        seen = []
        for rank in range(world):
            seen += [i for i, _ in scheduler.owned(pics, rank, world)]
        assert sorted(seen) == list(range(len(pics)))
        sizes = [len(scheduler.owned(pics, r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1


def test_balanced_assignment_evens_out_unequal_pictures():
    # a mini-GOP whose base-layer picture costs eight times a top-layer one, repeated
    costs = ([8.0, 4.0, 3.0, 2.0] + [1.0] * 12) * 6
    for world in (2, 4, 8):
        owners = scheduler.assign_balanced(costs, world)
        assert len(owners) == len(costs) and set(owners) == set(range(world))
        load = [sum(c for c, o in zip(costs, owners) if o == r) for r in range(world)]
        assert max(load) - min(load) <= max(costs)
        rr = [sum(c for i, c in enumerate(costs) if i % world == r) for r in range(world)]
        assert max(load) <= max(rr)  # never worse than round robin, which always hands rank 0 the expensive picture
        assert owners == scheduler.assign_balanced(costs, world)  # every rank computes the same list
    pics = gop.random_access(17, 3)
    owners = scheduler.assign_balanced([1.0] * len(pics), 2)
    seen = [i for r in range(2) for i, _ in scheduler.owned(pics, r, 2, owners)]
    assert sorted(seen) == list(range(len(pics)))


def test_needed_sources_cover_the_references_and_nothing_else():
    pics = gop.random_access(17, 3)
    for world in (1, 2, 4):
        for rank in range(world):
            need = set(scheduler.needed_sources(pics, rank, world))
            want = set()
            for _, p in scheduler.owned(pics, rank, world):
                want |= {p.poc, *p.refs[0], *p.refs[1]}
            assert need == want
            last = scheduler.last_use(pics, rank, world)
            assert set(last) == want
    assert scheduler.needed_sources(pics, 0, 1) == list(range(17))


def test_single_rank_matches_the_unsharded_run():
    frames, pics, rc = support.make_job(128, 64, 5, 8, seed=2)
    eng = support.OracleEngine(8, rc, store_slots=5)
    got = scheduler.run_sharded(eng, {i: frames[i] for i in range(len(frames))}, pics, None, 0, 1, in_flight=2)
    ref = support.oracle_handover(frames, pics, 8, rc)
    assert sorted(got) == list(range(len(pics)))
    for i, bd in enumerate(ref):
        for f in ("best_sad", "best_mv", "zz_sad", "sr_divisor", "hme_sc_x", "hme_sc_y", "do_ref"):
            assert (got[i][f] == bd[f]).all(), (i, f)
    assert not eng.store and eng.releases == eng.puts  # every source released after its last reader


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_over_gloo_reproduce_the_single_process_results(tmp_path):
    sys.path.insert(0, HERE)
    import sharded_worker
    enc_mode, world = 8, 2
    out = tmp_path / "gathered.json"
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "sharded_worker.py"), str(out), str(enc_mode)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        log, _ = p.communicate(timeout=600)
        assert p.returncode == 0, log.decode()[-2000:]
    got = json.loads(out.read_text())
    frames, pics, rc = support.make_job(128, 64, 9, enc_mode, seed=5)
    ref = support.oracle_handover(frames, pics, enc_mode, rc)
    want = {str(i): sharded_worker.digest({f: bd[f] for f in ("best_sad", "best_mv", "zz_sad", "sr_divisor", "hme_sc_x",
                                                                "hme_sc_y", "do_ref")}) for i, bd in enumerate(ref)}
    assert got["digests"] == want
    ran = [s["ran"] for s in got["stats"]]
    assert sorted(ran[0] + ran[1]) == list(range(len(pics))) and not set(ran[0]) & set(ran[1])
    for s in got["stats"]:
        assert s["left_resident"] == 0 and s["releases"] == s["puts"]
