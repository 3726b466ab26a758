"""Picture-sharding scheduler: which GPU (rank) runs the open-loop ME of which picture.

Open-loop ME reads only source pictures, so once the sources exist every picture of a mini-GOP is independent
(the reference posts them back to back, Source/Lib/Encoder/Codec/EbPictureDecisionProcess.c:6842-6874, and its ME
threads pick them up in any order). One process drives one GPU; picture i of the dispatch order goes to rank
i mod world. A rank's device store must hold the sources of its pictures and of their references, so every source is
put on every rank that needs it; results go from each GPU straight to its own pinned host buffers. There is no
exchange step inside a picture and no data-path collective. `gather` exists for callers (tests, tools) that want all
results in one process; it moves host-side results through torch.distributed objects, not through the GPUs.
"""


def owner(dispatch_index, world):
    """Rank that runs the picture with this position in dispatch order."""
    return dispatch_index % world


def owned(pics, rank, world, owners=None):
    """[(dispatch index, picture)] this rank runs, in dispatch order. owners: a per-picture rank list from
    assign_balanced instead of the round-robin rule."""
    if owners is not None:
        return [(i, p) for i, p in enumerate(pics) if owners[i] == rank]
    return [(i, p) for i, p in enumerate(pics) if owner(i, world) == rank]


def assign_balanced(costs, world):
    """Owners for a dispatch order whose pictures cost differently (a base-layer picture searches up to 7 references far
    away, a top-layer one 2 next to it): each picture goes to the rank with the least work so far, ties to the lowest
    rank. Deterministic, so every rank computes the same list from the same costs without talking to the others."""
    load = [0.0] * world
    owners = []
    for c in costs:
        r = min(range(world), key=lambda k: (load[k], k))
        owners.append(r)
        load[r] += float(c)
    return owners


def needed_sources(pics, rank, world, owners=None):
    """Sorted picture numbers whose source planes this rank's store must hold: its pictures and their references."""
    need = set()
    for _, p in owned(pics, rank, world, owners):
        need.add(p.poc)
        for lst in p.refs:
            need.update(lst)
    return sorted(need)


def last_use(pics, rank, world, owners=None):
    """{picture number: dispatch index of the last picture of this rank that reads it}: the store slot of a source
    can be released after that picture was waited for (what release_pa_reference_objects does on the host,
    Source/Lib/Encoder/Codec/EbReferenceObject.c:298)."""
    last = {}
    for i, p in owned(pics, rank, world, owners):
        last[p.poc] = i
        for lst in p.refs:
            for r in lst:
                last[r] = i
    return last


def run_sharded(engine, frames_by_poc, pics, ctrl, rank, world, in_flight=4, on_result=None, owners=None):
    """Runs this rank's share of `pics` on `engine` (an svtme.Engine or anything with its interface).

    frames_by_poc: {picture number: 2-D uint8 luma}. Sources are put into the store when first needed and released
    after their last reader; at most `in_flight` pictures are submitted ahead of the one being waited for.
    owners: per-picture ranks (assign_balanced) instead of the round-robin rule.
    Returns {dispatch index: results dict}, or passes each (index, results) to on_result and returns {}."""
    mine = owned(pics, rank, world, owners)
    last = last_use(pics, rank, world, owners)
    resident, results, queue = set(), {}, []

    def put(poc):
        if poc not in resident:
            engine.store_put(poc, poc, frames_by_poc[poc])
            resident.add(poc)

    def finish():
        i, p, t = queue.pop(0)
        out = engine.wait(t)
        engine.done(t)
        if on_result:
            on_result(i, out)
        else:
            results[i] = out
        for poc in [p.poc] + [r for lst in p.refs for r in lst]:
            if last.get(poc) == i and poc in resident:
                engine.store_release(poc)
                resident.discard(poc)

    for i, p in mine:
        while len(queue) >= in_flight:
            finish()
        put(p.poc)
        for lst in p.refs:
            for r in lst:
                put(r)
        h, w = frames_by_poc[p.poc].shape
        desc = engine.make_picture(w, h, p.poc, p.temporal_layer, p.is_ref, p.refs, ctrl)
        queue.append((i, p, engine.submit(desc)))
    while queue:
        finish()
    return results


def gather(results, world, group=None):
    """All ranks' {dispatch index: results} merged, on every rank (host objects over torch.distributed)."""
    if world == 1:
        return dict(results)
    import torch.distributed as dist
    parts = [None] * world
    dist.all_gather_object(parts, results, group=group)
    merged = {}
    for part in parts:
        merged.update(part)
    return merged
