#!/usr/bin/env python3
"""Open-loop ME throughput of the MI355X engine (pictures/s), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = the ME of one mini-GOP (all its pictures x all their references) over synthetic pictures that are already
resident in the device picture store when the timed region starts; results (best SAD / MV per partition, search
centres) are copied back to pinned host memory inside the timed region. The K timed steps run back to back with at
most --in-flight pictures outstanding (mini-GOPs follow each other without a pause, as in the encoder); all of them
have been waited for when the clock stops. Rank 0 prints one JSON line.

N > 1 shards ONE sequence: every rank generates the same K mini-GOPs (same seed) and runs its share of their pictures.
Pictures are independent (they read source pictures only), so there is no data-path collective; a picture goes to the
rank with the least work so far (svtme.scheduler.assign_balanced over the pictures' operation counts, which every rank
computes identically), `value` = all pictures of the sequence / the slowest rank's time ("strong" scaling: the
sequence is fixed, the ranks split it). With --with-upload every rank also puts, inside the timed region, the sources
its share of each mini-GOP reads; `sources_put_per_picture` then says how many GPUs a source travelled to.

`value` comes from the timed region. Batches (the default) go down one HIP stream, one kernel at a time on the GPU, and
the roofline figures are the per-kernel HIP events of that same region. With pictures submitted singly (--batch 1) the
engine rotates them over three streams (one picture's kernel tails overlap the other pictures' kernels); kernel durations
measured there include that sharing, so the roofline figures then come from a second pass over the same pictures with one
search stream, timed with the same HIP events on the kernels' stream; the overlapped durations are reported beside them.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in ("svt-av1_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, _p))

import numpy as np  # noqa: E402

# BASELINE.json configs -> shapes. "1080p_ra_m3" is configs[2]: 1080p random access, mini-GOP 16, HME L0/L1/L2 + full-pel
# search, up to 4+3 references (preset 3 = MRP level 3), the single-GPU 1080p configuration the metric is quoted on.
WORKLOADS = {
    "1080p_ra_m3": dict(width=1920, height=1080, enc_mode=3, structure="ra", n_frames=17, bit_depth=8, batch=16),
    # "batch": pictures per svtme_pictures_submit call (the pictures of a mini-GOP are posted back to back,
    # EbPictureDecisionProcess.c:6842-6874). The fast presets are bound by per-launch costs, which a mini-GOP submitted at
    # once pays once; the slow ones gain the shorter kernel tails (measured: profiles/r02_batch_sweep.txt). 1080p preset 8
    # and 720p preset 12 hand over two mini-GOPs per call (a look-ahead of 32 pictures): 7 us of kernels per 720p picture are
    # still short against what a set of launches costs (profiles/r03_batch_sweep.txt).
    "1080p_ra_m8": dict(width=1920, height=1080, enc_mode=8, structure="ra", n_frames=17, bit_depth=8, batch=32),
    # "streams": the batches alternate over two search streams, so that the kernels of one fill the tails of the other's -
    # every workload gains 5-25 % that way (profiles/r03_bench_*_two_streams.json), but kernels that share the GPU cannot be
    # timed one by one inside the timed region, so only this launch-bound workload (0.06 of the roof) does it by default
    "720p_ld_m12": dict(width=1280, height=720, enc_mode=12, structure="ldp", n_frames=17, bit_depth=8, batch=32, streams=2),
    "4k_ra_m8": dict(width=3840, height=2160, enc_mode=8, structure="ra", n_frames=17, bit_depth=10, batch=16),
    "8k_ra_m2": dict(width=7680, height=4320, enc_mode=2, structure="ra", n_frames=17, bit_depth=8, batch=2),
    # temporal-filter searches (ME_MCTF): a 7-frame window, the centre frame searched in each of its 6 neighbours; one
    # "picture" = one (centre, neighbour) pair. tf = (tf_ctrls.hme_me_level, tf_ctrls.me_exit_th): level 0 is what presets
    # up to M5 use, level 2 with the 16x16 exit threshold what the fast presets use (EbEncHandle.c:2556-2876). A submission
    # is the six pairs of one filtered picture, at level 2 those of two filtered pictures.
    "1080p_tf_l0": dict(width=1920, height=1080, enc_mode=0, structure="tf", n_frames=7, bit_depth=8, tf=(0, 0), batch=6),
    "1080p_tf_l2": dict(width=1920, height=1080, enc_mode=8, structure="tf", n_frames=7, bit_depth=8, tf=(2, 256), batch=12),
    # first-pass ME of a two-pass encode (ME_FIRST_PASS): every frame searches the previous and the first frame;
    # fp = (ipp_pass_ctrls.bypass_zz_check, ipp_pass_ctrls.reduce_me_search)
    "1080p_first_pass": dict(width=1920, height=1080, enc_mode=8, structure="fp", n_frames=17, bit_depth=8, fp=(0, 0), batch=16),
    # the fastest first pass of a 1080p two-pass encode (set_ipp_pass_ctrls level 2, EbEncHandle.c:3061-3069): pictures
    # down-sampled to 960x544, no zero-vector check, reduced search, every second 64x64 block of every second block row
    # (MeContext::bypass_blk_step, firstpass.c:1282-1285, :1313-1317)
    "1080p_first_pass_fast": dict(width=960, height=544, enc_mode=8, structure="fp", n_frames=17, bit_depth=8, fp=(1, 1), b64_step=2,
                                  batch=16),
}
STAGES = ["zz_centre", "prehme", "hme_l0", "hme_l1", "hme_l2", "integer_search"]
# Packed-SAD peak of gfx950: 256 CUs x 4 SIMDs x 16 lanes/clk (v_sad_u8 / v_qsad_pk_u16_u8 issue at a quarter of the
# lane rate, measured in profiles/r01_probe_sad_rates.txt: 146 T/s sustained) x 4 byte differences x 2.4 GHz.
PEAK_SAD_TOPS = 256 * 4 * 16 * 4 * 2.4e9 / 1e12
MEASURED_SAD_TOPS = 146.6  # sustained v_qsad_pk_u16_u8 rate of a probe kernel that does nothing else (profiles/r01_probe_sad_rates.txt)
PEAK_HBM_GBS = 8000.0
SEED = 1000  # of the synthetic sequence; tests/golden/bench_digests.json holds the reference's digests for it


def make_pictures(wl, seed):
    from svtme import gop, synth
    frames = synth.synth_sequence(wl["width"], wl["height"], wl["n_frames"], seed=seed, bit_depth=wl["bit_depth"])
    if wl["structure"] == "tf":
        c = wl["n_frames"] // 2
        pics = [gop.Picture(poc=c, temporal_layer=0, is_ref=1, refs=[[j], []]) for j in range(wl["n_frames"]) if j != c]
    elif wl["structure"] == "fp":  # firstpass.c:1166-1186: one list, base layer, used as reference
        pics = [gop.Picture(poc=i, temporal_layer=0, is_ref=1, refs=[[i - 1] + ([0] if i >= 2 else []), []])
                for i in range(1, wl["n_frames"])]
    elif wl["structure"] == "ra":
        pics = gop.random_access(wl["n_frames"], wl["enc_mode"])
    else:
        pics = gop.low_delay(wl["n_frames"], wl["enc_mode"], n_list=1 if wl["structure"] == "ldp" else 2)
    return frames, pics


def cpu_baseline(wl, frames, pics, budget_s=20.0):
    """Times the CPU path on this machine's host cores over a bounded sample of the same workload: the real reference
    ME with its AVX2 kernels (oracle/_ref/ref_me) when that binary is present, else the scalar oracle port."""
    import refio
    from svtme import gop
    threads = max(1, min(16, os.cpu_count() or 1))
    b0, b1, n0, n1 = gop.mrp_counts(wl["enc_mode"])
    rc = (2, 0) if wl.get("fp") else (max(b0, n0), max(b1, n1))
    kind, binary, avx2 = ("reference", refio.REF_BIN, 1) if os.path.exists(refio.REF_BIN) else ("port", refio.ORACLE_BIN, 0)
    sample = pics[:4] if kind == "port" else pics
    with tempfile.TemporaryDirectory(prefix="svtme_bench_") as td:
        job, out = os.path.join(td, "job.bin"), os.path.join(td, "out.bin")

        def run(reps, sample=sample, threads=threads):
            refio.write_job(job, frames, sample, wl["enc_mode"], mode=refio.MODE_BENCH, threads=threads, use_avx2=avx2,
                            reps=reps, ref_counts=rc, tf=wl.get("tf"), first_pass=wl.get("fp"), b64_step=wl.get("b64_step", 0))
            r = subprocess.run([binary, job, out], capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                raise RuntimeError(r.stderr[-500:])
            hdr, _ = refio.read_dump(out)
            return hdr
        hdr = run(1)
        per_pass = max(hdr["bench_seconds"], 1e-3)
        reps = int(min(20, max(1, (budget_s / threads) / per_pass)))
        if reps > 1:
            hdr = run(reps)
        # the same path on one host thread, over as many of the sample's pictures as a quarter of the budget allows
        n1 = int(min(len(sample), max(1, (budget_s / 4) / max(per_pass * threads / len(sample), 1e-3))))
        hdr1 = run(1, sample[:n1], 1)
    used_avx2 = bool(hdr.get("used_avx2", 0))
    return {
        "value": hdr["bench_pictures"] / hdr["bench_seconds"], "unit": "pictures/s", "cores": threads, "kind": kind,
        "single_thread": {"value": hdr1["bench_pictures"] / max(hdr1["bench_seconds"], 1e-9), "cores": 1,
                          "sample": "the first %d pictures of the same sample, one pass" % n1},
        "sample": "%d pictures x %d passes of the same %s workload, %s, %d threads over b64 rows" %
                  (len(sample), hdr["bench_pictures"] // max(1, len(sample)), wl["name"],
                   "reference C control code + its AVX2 SAD kernels" if (kind == "reference" and used_avx2)
                   else ("reference C-only kernels" if kind == "reference" else "scalar C restatement"), threads),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="1080p_ra_m3", choices=sorted(WORKLOADS))
    ap.add_argument("--in-flight", type=int, default=0, help="pictures outstanding at most (default: 6, or 3 batches)")
    ap.add_argument("--batch", type=int, default=0,
                    help="pictures per svtme_pictures_submit call (they share their kernel launches); 0 = the workload's default")
    ap.add_argument("--search-streams", type=int, default=0, choices=[0, 1, 2, 3],
                    help="0 = the engine's default: one stream for batches, three for pictures submitted singly")
    ap.add_argument("--kernel-steps", type=int, default=24, help="steps of the one-stream pass that times the kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the check (outside the timed region) of one more step's results against the reference's digest")
    ap.add_argument("--with-upload", action="store_true", default=None,
                    help="every step also puts each of its source pictures into the device store once, inside the timed "
                         "region (host -> device included). The default with more than one GPU, where how far the sources "
                         "travel is part of what is measured; with one GPU the default line keeps the inputs resident")
    ap.add_argument("--no-upload", dest="with_upload", action="store_false",
                    help="more than one GPU: every rank holds every source before the clock starts (the ranks are then "
                         "independent streams of searches)")
    ap.add_argument("--group", action="store_true",
                    help="one process drives --gpus engines through svtme_group (the encoder's shape: one host process, a "
                         "mini-GOP at a time split over the GPUs, sources from engine to engine where the GPUs are peers); "
                         "start it as a single process")
    ap.add_argument("--upload-mode", default="pinned", choices=["pinned", "copy"],
                    help="--with-upload: zero-copy puts from page-locked buffers with the encoder's stride, or svtme_store_put")
    ap.add_argument("--device-tail-only", action="store_true", help="--device-tail, and do not copy the search arrays back")
    ap.add_argument("--device-tail", action="store_true",
                    help="also run me_prune_ref / candidate construction / distortions / GM detection on the device")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the engine has no CPU fallback")
    # rehearsal on a one-GPU machine: SVTME_BENCH_REHEARSAL=1 puts every rank on GPU 0 and forms the group over gloo
    # (RCCL refuses two ranks on one device); the numbers of such a run mean nothing, it exercises the N > 1 code path
    rehearsal = os.environ.get("SVTME_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # under torch.distributed.run (RANK set) the group is always formed, also for one rank, so that the path is the same
    use_dist = world > 1 or "RANK" in os.environ

    def init_dist():
        """Forms the process group - AFTER the engines exist. HIP hands its few hardware queues (4 per process here) to
        streams in the order of their creation; RCCL creates streams of its own, and formed first they shift the engine's
        streams onto other queues, where the search stream shares one with the copy stream: the same single-GPU run is 16 %
        slower (4.2 k against 5.0 k pictures/s) although no collective runs inside the timed region."""
        if use_dist:
            if rehearsal:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import svtme
    wl = dict(WORKLOADS[args.workload], name=args.workload)
    sharded = world > 1 and not args.group  # one sequence, split over the ranks
    if args.group and world > 1 and rank == 0:
        print("bench.py --group drives all GPUs from one process: the other ranks only keep the barriers company", file=sys.stderr)
    if args.with_upload is None:
        args.with_upload = sharded or args.group
    if args.batch <= 0:
        args.batch = wl.get("batch", 1)
        if args.with_upload:  # a batch waits for the puts of all its pictures: shorter batches overlap better with the uploads
            args.batch = min(args.batch, 8)
    args.batch = min(args.batch, svtme.MAX_BATCH)
    if args.in_flight <= 0:
        args.in_flight = max(6, 3 * args.batch)
    args.in_flight = max(args.in_flight, args.batch)
    frames, pics = make_pictures(wl, seed=SEED)  # every rank sees the same sequence
    n, h, w = frames.shape
    tf, fp = wl.get("tf"), wl.get("fp")
    ctrl = (svtme.derive_tf_controls(tf[0]) if tf else
            svtme.derive_first_pass_controls(w, h, reduce_me_search=fp[1]) if fp else svtme.derive_controls(wl["enc_mode"], w, h))
    n_refs = sum(len(p.refs[0]) + len(p.refs[1]) for p in pics)

    tail = None
    if tf and (args.device_tail or args.device_tail_only):
        sys.exit("temporal-filter searches have no host tail")
    if fp and (args.device_tail or args.device_tail_only):
        tail = svtme.Tail(enable=2 if args.device_tail_only else 1, enable_me_8x8=1, enable_me_16x16=1, max_refs=2, max_cand=3,
                          max_l0=1, input_resolution=svtme.input_resolution(w, h), gm_enabled=0)
    elif args.device_tail or args.device_tail_only:
        from svtme import gop
        b0, b1, n0, n1 = gop.mrp_counts(wl["enc_mode"])
        l0, l1 = max(b0, n0), max(b1, n1)
        res = svtme.input_resolution(w, h)
        m = wl["enc_mode"]
        # pcs->enable_me_8x8 / enable_me_16x16 and the me_results sizes (EbPictureControlSet.c:83-117, :1533)
        no8 = 0 if m <= 8 else ((0 if res <= 4 else 1) if m <= 9 else ((0 if res <= 2 else 1) if m <= 11 else 1))
        tail = svtme.Tail(enable=2 if args.device_tail_only else 1, enable_me_8x8=0 if no8 else 1, enable_me_16x16=1 if m <= 12 else 0, max_refs=l0 + l1,
                          max_cand=l0 + l1 + l0 * l1 + (l0 - 1) + (1 if l1 == 3 else 0), max_l0=l0, input_resolution=res,
                          gm_enabled=1, gm_use_distance_based_active_th=1)

    KEY_SETS = 2  # --with-upload: consecutive steps use different store slots, as new pictures do in the encoder's pool

    def make_engine(search_streams, kernel_timing):
        e = svtme.Engine(w, h, store_slots=KEY_SETS * n + 1, max_in_flight=args.in_flight, device=local_rank,
                         search_streams=search_streams, kernel_timing=kernel_timing)
        for i in range(n):
            e.store_put(i, i, frames[i])  # inputs resident in HBM before the timed region
        return e, [e.make_picture(w, h, p.poc, p.temporal_layer, p.is_ref, p.refs, ctrl, tail=tail,
                                  tf_me_exit_th=tf[1] if tf else None,
                                  first_pass_bypass_zz=fp[0] if fp else None, b64_step=wl.get("b64_step", 0)) for p in pics]

    # per-kernel events only where they are needed: with one search stream they are the roofline timings
    if not args.search_streams and args.batch > 1 and not args.with_upload and not args.group:
        args.search_streams = wl.get("streams", 0)
    streams = args.search_streams or (1 if args.batch > 1 else 3)  # what the engine makes of 0
    eng, descs = make_engine(args.search_streams, 1 if streams == 1 else 0)
    engine_times_kernels, timing_state = [streams == 1], [True]  # (the engine starts with the events on)
    new_stats = lambda: {"ms": np.zeros(8, np.float64), "launches": 0, "ops": np.zeros(6, np.float64)}  # noqa: E731
    stats = new_stats()

    # --with-upload: the sources as the encoder holds them, padded input buffers (origin 72, 72; stride = width + 144)
    # registered once; each step puts every source of its mini-GOP once, right before the first picture that needs it
    pool = None
    if args.with_upload:
        pool = eng.host_alloc((n, h + 144, w + 144)) if args.upload_mode == "pinned" else np.zeros((n, h + 144, w + 144), np.uint8)
        pool[:, 72:72 + h, 72:72 + w] = frames
        # the same pictures under a second set of store keys for every other step
        descs_alt = [[eng.make_picture(w, h, p.poc, p.temporal_layer, p.is_ref, p.refs, ctrl, tail=tail,
                                       tf_me_exit_th=tf[1] if tf else None, first_pass_bypass_zz=fp[0] if fp else None,
                                       b64_step=wl.get("b64_step", 0),
                                       key_of=lambda poc, ks=ks: 1000 * ks + poc) for p in pics] for ks in range(KEY_SETS)]

    puts_made = [0]

    def put_sources(e, ks_list, ks):
        """The sources a submission needs and the store does not hold yet, as one batched put (svtme_store_put_many)."""
        for a in range(0, len(ks_list), svtme.MAX_PUT_BATCH):
            part = ks_list[a:a + svtme.MAX_PUT_BATCH]
            e.store_put_many([1000 * ks + k_ for k_ in part], part, [pool[k_, 72:72 + h, 72:72 + w] for k_ in part],
                             pinned=args.upload_mode == "pinned")
        puts_made[0] += len(ks_list)

    def run_steps(k, collect, e=None, ds=None, upload=False, owners=None, sample=None, keep=None):
        """k steps back to back: at most --in-flight pictures outstanding, also across step boundaries (mini-GOPs follow
        each other without a pause, as they do in the encoder); everything has been waited for on return. owners: the
        rank of every picture of the k steps (this rank runs its own ones only). sample = (stats, stride): the kernel
        durations of every submission of every stride-th step are added to stats (whole steps, because the submissions of
        a step differ - the pictures of a mini-GOP have different reference counts and distances - while the operation
        counts the durations are set against are averages over whole steps; one svtme_picture_timing call for a
        submission: its pictures share their launches and report equal shares). keep: a list that receives every picture's results."""
        e, ds = e or eng, ds or descs
        timing_switch = e is eng and engine_times_kernels[0]
        tickets, group, new_sources = [], [], []
        sampled, cur_step = set(), [0]

        def finish(e, t, collect):
            if keep is not None:
                keep.append(e.wait(t))
            else:
                e.wait_raw(t)  # the picture's results have landed in pinned host memory (no conversion to numpy in the timed loop)
            if t in sampled:
                sampled.discard(t)
                sample[0]["ms"] += np.array(e.timing(t), np.float64)
                sample[0]["launches"] += 1
            if collect is not None:
                collect["ms"] += np.array(e.timing(t), np.float64)
                collect["launches"] += 1
                collect["ops"] += np.array(e.ops(t), np.float64)
            e.done(t)

        def flush():
            while len(tickets) + len(group) > args.in_flight:
                finish(e, tickets.pop(0), collect)
            if new_sources:
                put_sources(e, new_sources, flush.ks)
                new_sources.clear()
            if timing_switch:  # per-kernel events only on the submissions whose durations will be read
                want = collect is not None or (sample is not None and cur_step[0] % sample[1] == 0)
                if want != timing_state[0]:
                    e.kernel_timing(want)
                    timing_state[0] = want
            new = e.submit_batch(group) if len(group) > 1 else [e.submit(group[0])]
            if sample is not None and cur_step[0] % sample[1] == 0:
                sampled.add(new[0])
            tickets.extend(new)
            group.clear()

        for step in range(k):
            ks = step % KEY_SETS
            cur_step[0] = step
            seen = set()  # sources this rank has put for this step
            for i, d in enumerate(descs_alt[ks] if upload else ds):
                if owners is not None and owners[step * len(ds) + i] != rank:
                    continue
                if upload:
                    for k_ in dict.fromkeys([pics[i].poc] + [r for lst in pics[i].refs for r in lst]):
                        if k_ not in seen:
                            seen.add(k_)
                            new_sources.append(k_)
                flush.ks = ks
                group.append(d)
                if len(group) == args.batch:
                    flush()
            if group and upload:  # (a step's key set ends with the step)
                flush()
        if group:
            flush()
        while tickets:
            finish(e, tickets.pop(0), collect)

    # --group: one process, one engine per GPU behind svtme_group. Every step announces its sources afresh (new pictures:
    # every copy on the GPUs is stale) and hands the mini-GOP over in submissions of --batch pictures; the group decides
    # where they run and brings the sources there, from the host or from the engine that already has them.
    grp, gdescs, per_engine = None, None, [0] * max(1, args.gpus)
    if args.group and rank == 0:
        devices = [0] * args.gpus if rehearsal else list(range(args.gpus))
        cap = args.in_flight * args.gpus
        grp = svtme.Group(devices, w, h, store_slots=KEY_SETS * n + 1, max_in_flight=cap, search_streams=args.search_streams)
        gdescs = [[grp.make_picture(w, h, p.poc, p.temporal_layer, p.is_ref, p.refs, ctrl, tail=tail,
                                    tf_me_exit_th=tf[1] if tf else None, first_pass_bypass_zz=fp[0] if fp else None,
                                    b64_step=wl.get("b64_step", 0), key_of=lambda poc, ks=ks: 1000 * ks + poc) for p in pics]
                  for ks in range(KEY_SETS)]

    init_dist()

    def run_group_steps(k, keep=None, count=False):
        tickets = []
        cap = args.in_flight * args.gpus

        def finish(t):
            if keep is not None:
                keep.append(grp.wait(t))
            else:
                grp.wait_raw(t)
            grp.done(t)

        for step in range(k):
            ks = step % KEY_SETS
            for k_ in range(n):
                grp.store_put(1000 * ks + k_, k_, pool[k_, 72:72 + h, 72:72 + w], pinned=1 if args.upload_mode == "pinned" else 0)
            for a in range(0, len(pics), args.batch):
                part = gdescs[ks][a:a + args.batch]
                while len(tickets) + len(part) > cap:
                    finish(tickets.pop(0))
                new = grp.submit_batch(part)
                if count:
                    for t in new:
                        per_engine[grp.engine_of(t)] += 1
                tickets.extend(new)
        while tickets:
            finish(tickets.pop(0))

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    owners_warm = owners_timed = None
    per_rank = [len(descs) * args.steps]
    if sharded:
        # what each picture of a mini-GOP costs: its operation count (exact, the same on every rank) plus a fixed share
        # for the per-picture overheads; every rank runs the mini-GOP once to read the counters
        cal = new_stats()
        per_pic = []
        for d in descs:
            t = eng.submit(d)
            eng.wait(t, copy=False)
            per_pic.append(float(sum(eng.ops(t))))
            eng.done(t)
        fixed = 0.25 * sum(per_pic) / len(per_pic)
        costs = [c + fixed for c in per_pic]
        from svtme import scheduler
        owners_warm = scheduler.assign_balanced(costs * max(1, args.warmup), world)
        owners_timed = scheduler.assign_balanced(costs * args.steps, world)
        per_rank = [owners_timed.count(r) for r in range(world)]
    if grp is not None:
        run_group_steps(args.warmup)
    elif not args.group:
        run_steps(args.warmup, None, upload=args.with_upload, owners=owners_warm)
    barrier()
    puts_made[0] = 0
    g0 = grp.stats() if grp is not None else None
    t0 = time.perf_counter()
    # nothing but (put,) submit / wait / done per picture inside the timed region - and, with one search stream, one
    # svtme_picture_timing call per submission of every few steps (16-odd steps over the region; the engine records the
    # per-kernel events for those submissions only, because they cost device time: svtme_kernel_timing): the kernels' HIP events
    timed = new_stats()
    if grp is not None:
        run_group_steps(args.steps, count=True)
    elif not args.group:
        run_steps(args.steps, None, upload=args.with_upload, owners=owners_timed,
                  sample=(timed, max(1, args.steps // 16)) if streams == 1 else None)
    if grp is not None and not rehearsal:
        for d in range(args.gpus):
            torch.cuda.synchronize(d)
    else:
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    puts_timed = puts_made[0]
    group_stats = None
    if grp is not None:
        g1 = grp.stats()
        group_stats = {k_: g1[k_] - g0[k_] for k_ in ("pictures", "sources", "uploads", "peer_copies")}
        group_stats["peer_links"] = g1["peer_links"]
        puts_timed = group_stats["uploads"] + group_stats["peer_copies"]
        per_rank = list(per_engine)
    if use_dist:  # the slowest rank's time (before rank 0 goes on to its reporting passes, so that no rank waits in a collective)
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        tp = torch.tensor([float(puts_timed)], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(tp, op=dist.ReduceOp.SUM)
        puts_timed = int(tp.item())
    run_steps(min(args.steps, 4), stats)  # kernel spans while the streams share the GPU (reported beside the solo timings)
    # host -> device: what putting one new source picture into the store costs (pinned staging copy, H2D, the pad /
    # decimate kernels); every picture is put once and then read by its own search and by up to 7 later ones
    upload_pps = None
    if rank == 0:
        t1 = time.perf_counter()
        for rep in range(4):
            for i in range(n):
                eng.store_put(i, i, frames[i])
        fh, fw = frames[0].shape
        eng.store_read(n - 1, 0, (fh // 4, fw // 4))  # waits for the store stream (reads back the smallest plane of the last put)
        upload_pps = 4 * n / max(1e-9, time.perf_counter() - t1)
    # the last word on parity: one more step, its results against the digest of the real reference's output for the same
    # seeded pictures (tests/golden/bench_digests.json, written by tests/golden/make_golden.py --bench)
    verified, verify_note = None, "skipped"
    if rank == 0 and not args.no_verify:
        from svtme import verify
        try:
            with open(os.path.join(ROOT, "tests", "golden", "bench_digests.json")) as f:
                want = json.load(f)["digests"].get(args.workload)
        except (OSError, ValueError, KeyError):
            want = None
        import hashlib
        if args.device_tail_only:
            verify_note = "--device-tail-only leaves the search arrays on the device"
        elif want is None:
            verify_note = "tests/golden/bench_digests.json holds no digest of this workload"
        elif hashlib.sha256(frames.tobytes()).hexdigest() != want["frames_sha256"]:
            verified, verify_note = False, "the synthetic pictures differ from the ones the reference digested"
        else:
            outs = []
            if grp is not None:
                run_group_steps(1, keep=outs)  # (through the group: what was timed)
            else:
                run_steps(1, None, keep=outs)
            got = verify.engine_digest(outs, pics, verify.kind_of(wl), (w + 63) // 64, wl.get("b64_step", 0))
            verified = got == want["sha256"]
            verify_note = ("%d pictures of one more step (submitted %d at a time, as in the timed region): sha256 over their search "
                           "centres, zero-MV SADs, divisors and best SAD / MV arrays %s the reference's (oracle/_ref/ref_me)"
                           % (len(outs), args.batch, "equals" if verified else "DIFFERS FROM"))
    # kernels alone on the GPU: same pictures, one search stream (rank 0 reports them)
    solo = stats
    if streams == 1 and timed["launches"]:
        # the timed region's own events; the operation counts (the same for every step: same pictures) from the pass after it
        solo = {"ms": timed["ms"], "launches": timed["launches"], "ops": stats["ops"] * (timed["launches"] / max(1, stats["launches"]))}
    if rank == 0 and streams != 1 and args.kernel_steps > 0:
        eng.close()
        eng, descs = make_engine(1, 1)
        engine_times_kernels[0], timing_state[0] = True, True
        solo = new_stats()
        run_steps(max(1, args.kernel_steps // 3), None)  # (the new engine's buffers took long enough for the clocks to drop)
        run_steps(args.kernel_steps, solo)
        torch.cuda.synchronize()
    if rank == 0:
        pictures = len(descs) * args.steps  # one sequence, whatever the number of ranks
        L = max(1, solo["launches"])
        ops = solo["ops"] / L  # per picture = per launch of each kernel
        ms = solo["ms"] / L  # per-kernel launch durations, HIP events on the kernels' stream, one search stream
        ms_shared = stats["ms"] / max(1, stats["launches"])  # the same in the timed region (streams share the GPU)
        # abs-diff-accumulate operations each kernel performs per launch (svtme_picture_ops counters)
        kernels = {
            "hme_l0_kernel": (float(ops[1] + ops[2]), float(ms[0])),  # pre-HME + HME level 0 (+ zero-MV SADs)
            "hme_l12_kernel": (float(ops[3] + ops[4]), float(ms[1])),
            "me_plan_kernel": (float(ops[0]), float(ms[2])),  # search-centre checks
            "me_search_kernel": (float(ops[5]), float(ms[3])),
            "me_finish_kernel": (0.0, float(ms[4])),
        }
        name = max(kernels, key=lambda k_: kernels[k_][1])
        dom_ops, dom_ms = kernels[name]
        achieved = dom_ops / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        # memory-side bytes per launch of that kernel: from the committed summary of the separate rocprofv3 --pmc
        # passes over this same command (FETCH_SIZE doubled: gfx950 tallies 128-byte read requests as 64 bytes)
        traffic, traffic_note = None, "no rocprofv3 --pmc summary of this workload under profiles/"
        try:
            import glob
            import hashlib
            src_hash = hashlib.sha256()
            for nm in sorted(os.listdir(os.path.join(ROOT, "svt-av1_amd", "csrc"))):
                with open(os.path.join(ROOT, "svt-av1_amd", "csrc", nm), "rb") as f:
                    src_hash.update(nm.encode() + b"\0" + f.read())
            for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic_%s.json" % args.workload)), reverse=True):
                with open(path) as f:
                    doc = json.load(f)
                if doc.get("source_sha256") != src_hash.hexdigest():
                    traffic_note = "%s was measured on other kernel sources: not reported" % os.path.basename(path)
                    continue
                t = doc["kernels"]["svtme::" + name]
                # (the counters are per launch of a whole batch; the figures here are per picture)
                traffic = (2.0 * t["fetch_kib"] + t["write_kib"]) * 1024.0 / max(1, int(doc.get("pictures_per_launch", 1)))
                traffic_note = "%s (2 x FETCH_SIZE + WRITE_SIZE per picture of a launch, measured on these kernel sources)" % os.path.basename(path)
                break
        except (OSError, KeyError, ValueError):
            pass
        # algorithmic HBM bytes of one picture: each needed plane once + results
        refs_per_pic = n_refs / len(pics)
        plane_bytes = w * h * (1 + 0.25 + 0.0625)
        alg_bytes = (1 + refs_per_pic) * plane_bytes + (w // 64) * ((h + 63) // 64) * refs_per_pic * 85 * 8
        kernel_ms = float(ms[5])
        out = {
            "metric": "open-loop ME pictures/s", "value": pictures / elapsed, "unit": "pictures/s",
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "n_gpus": args.gpus if args.group else world,
            "config": {"workload": args.workload, "width": w, "height": h,
                       "preset": ("tf hme_me_level %d, me_exit_th %d" % tf if tf else
                                  "first pass, bypass_zz_check %d, reduce_me_search %d" % fp if fp else wl["enc_mode"]),
                       "pred_structure": wl["structure"], "pictures_per_step": len(descs),
                       "refs_per_picture": round(refs_per_pic, 2), "bit_depth": wl["bit_depth"],
                       "parallelism": "the pictures of one sequence sharded over %d GPU(s) by least accumulated work, "
                                      "no data-path collective" % (args.gpus if args.group else world)},
            "sharding": {"mode": ("one process, svtme_group over %d engines" % args.gpus) if args.group else "one process per GPU",
                         "pictures_per_rank": per_rank,
                         # how many GPUs a source travelled to, per picture searched; of those, device to device
                         "sources_put_per_picture": (puts_timed / float(pictures)) if args.with_upload else None,
                         "peer_copies": group_stats["peer_copies"] if group_stats else 0,
                         "uploads": group_stats["uploads"] if group_stats else (puts_timed if args.with_upload else 0),
                         "peer_links": group_stats["peer_links"] if group_stats else 0,
                         "sources_per_step": n},
            "roofline": {
                "bound": "valu", "kernel": name, "achieved": achieved, "peak": PEAK_SAD_TOPS,
                "unit": "T abs-diff-acc/s", "frac": achieved / PEAK_SAD_TOPS, "traffic": traffic, "traffic_source": traffic_note,
                # a launch covers the `pictures_per_submit` pictures of a batch: operations, duration and traffic are the
                # launch's divided by that number, i.e. per picture
                "ops_per_launch": dom_ops, "avg_launch_ms": dom_ms,
                "frac_vs_measured_rate": achieved / MEASURED_SAD_TOPS,
                "measured_rate": MEASURED_SAD_TOPS,
                "measured": ("HIP events on the kernels' stream inside the timed region, %d of its submissions sampled" % timed["launches"]
                             if streams == 1 and timed["launches"] else
                             "HIP events on the kernels' stream, %d-step pass with one search stream after the timed region" % args.kernel_steps),
                "all_kernels": {"ops_per_picture": float(ops.sum()), "ms_per_picture": kernel_ms,
                                "achieved": float(ops.sum()) / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0,
                                "frac": (float(ops.sum()) / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0) / PEAK_SAD_TOPS},
                "hbm_view": {"algorithmic_bytes_per_picture": alg_bytes,
                             "achieved_GBps": alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0,
                             "peak_GBps": PEAK_HBM_GBS},
            },
            "kernels": {k_: {"avg_ms": v[1], "ops_per_launch": v[0], "tops": v[0] / (v[1] * 1e-3) / 1e12 if v[1] > 0 else 0.0}
                        for k_, v in kernels.items()},
            "picture_kernels_ms_in_timed_region": float(ms_shared[5]),  # first kernel start to last kernel end
            "search_streams": streams, "pictures_per_submit": args.batch, "in_flight": args.in_flight, "device_tail": (2 if args.device_tail_only else 1) if tail is not None else 0,
            "store_put_pictures_per_s": upload_pps,
            "with_upload": ({"mode": args.upload_mode, "sources_put_per_step": n,
                             "note": "every source picture of a step is put into the device store once inside the timed region"}
                            if args.with_upload else None),
            "verified": verified, "verify": verify_note,
            "ops_by_stage": dict(zip(STAGES, [float(v) for v in ops])),
            "picture_ms_incl_d2h": float(ms[6]),
        }
        if not args.no_cpu_baseline and world == 1:  # the host-core baseline is measured at N = 1 only
            try:
                out["cpu_baseline"] = cpu_baseline(wl, frames, pics)
            except Exception as ex:  # the baseline is a reported number, never a reason to lose the GPU line
                out["cpu_baseline"] = {"value": None, "unit": "pictures/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (ex,)}
        print(json.dumps(out), flush=True)
    if grp is not None:
        grp.close()
    eng.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
