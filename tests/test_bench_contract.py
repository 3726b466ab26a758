"""The one-line JSON contract of bench.py: checked on the committed runs under profiles/ (CPU) and on a short live run (GPU)."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TOP = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float,
       "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict, "roofline": dict}
ROOFLINE = {"bound": str, "achieved": float, "peak": float, "unit": str, "frac": float}
CPU_BASELINE = {"value": float, "unit": str, "cores": int, "kind": str, "sample": str}


def check_line(d, expect_cpu_baseline):
    for k, t in TOP.items():
        assert k in d and isinstance(d[k], t), k
    assert "vs_baseline" in d and d["vs_baseline"] is None  # no published number exists for this metric
    assert d["metric"] == "open-loop ME pictures/s" and d["unit"] == "pictures/s"
    # N ranks split one sequence ("strong"); replicas per rank would be "weak"
    assert d["higher_is_better"] is True and d["scaling"] in ("weak", "strong") and d["dtype"] == "u8" and d["data"] == "synthetic"
    assert isinstance(d["config"].get("workload"), str) and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    r = d["roofline"]
    for k, t in ROOFLINE.items():
        assert k in r and isinstance(r[k], t), k
    assert "traffic" in r and (r["traffic"] is None or r["traffic"] > 0)
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    if expect_cpu_baseline:
        c = d["cpu_baseline"]
        for k, t in CPU_BASELINE.items():
            assert k in c and isinstance(c[k], t), k
        assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_*.json"))))
def test_committed_bench_lines_keep_the_contract(path):
    # (the upload-inclusive, group and other-stream-count lines are a second view of a workload and carry no CPU baseline)
    second_view = any(t in path for t in ("with_upload", "_group", "_two_streams", "_one_stream"))
    with open(path) as f:
        check_line(json.loads(f.read().strip().splitlines()[-1]), expect_cpu_baseline=not second_view)


def test_full_size_parity_test_submits_what_the_benchmark_submits():
    """The batches of test_engine_matches_reference_digests_at_full_size are bench.py's (pictures per submit): the parity of
    the timed configuration cannot drift from the configuration that is timed."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    import svtme
    import test_gpu_parity as t
    same = {"m3_ra_1920x1080": "1080p_ra_m3", "m8_ra_1920x1080": "1080p_ra_m8", "m12_ldp_1280x720": "720p_ld_m12",
            "m8_ra_3840x2160": "4k_ra_m8", "m2_ra_7680x4320": "8k_ra_m2"}
    for case, workload in same.items():
        wl = bench.WORKLOADS[workload]
        assert t.BENCH_BATCH[case] == wl["batch"] <= svtme.MAX_BATCH, (case, workload)
        c = t.MOD.DIGEST_CASES.get(case) or t.MOD.BIG_DIGEST_CASES[case]
        assert (c["width"], c["height"], c["enc_mode"]) == (wl["width"], wl["height"], wl["enc_mode"]), (case, workload)


def test_bench_help_needs_no_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "--gpus" in r.stdout and "--steps" in r.stdout and "--warmup" in r.stdout


@pytest.mark.gpu
def test_live_bench_line_keeps_the_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--kernel-steps", "2",
                        "--workload", "1080p_ra_m8"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    check_line(d, expect_cpu_baseline=True)
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["config"]["workload"] == "1080p_ra_m8"


@pytest.mark.gpu
def test_two_ranks_split_one_sequence_on_this_box():
    """The N > 1 path of bench.py, rehearsed with two ranks on the one GPU of this box (process group over gloo, both
    ranks on device 0): one sequence, every picture run by exactly one rank, sources put where they are read."""
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, SVTME_BENCH_REHEARSAL="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--workload", "1080p_ra_m8", "--kernel-steps", "0", "--no-cpu-baseline", "--with-upload"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    sh = d["sharding"]
    assert sum(sh["pictures_per_rank"]) == 4 * d["config"]["pictures_per_step"] and min(sh["pictures_per_rank"]) > 0
    # every source goes to at least one and at most two GPUs
    per_source = sh["sources_put_per_picture"] * d["config"]["pictures_per_step"] / sh["sources_per_step"]
    assert 1.0 <= per_source <= 2.0
