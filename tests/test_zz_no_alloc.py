"""svtme_picture_search_static, the entry point that allocates nothing (its scratch is a file-scope device array of the
library, its results go into caller-owned memory), against svtme_picture_submit / svtme_picture_wait on the same
pictures: bit-exact, on two calls in a row of which the second has the smaller input, with the result block once in
pinned host memory and once in device memory. Collected last on purpose (the file name): a failure here must not hide
the tests before it."""
import ctypes as C

import numpy as np
import pytest

import support
import svtme

pytestmark = pytest.mark.gpu


class _Hip:
    """The few HIP runtime calls the test needs for the caller-owned result blocks (test infrastructure)."""

    def __init__(self):
        self.rt = C.CDLL("/opt/rocm/lib/libamdhip64.so")
        self.rt.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.rt.hipHostFree.argtypes = [C.c_void_p]
        self.rt.hipFree.argtypes = [C.c_void_p]

    def pinned(self, n):
        p = C.c_void_p()
        assert self.rt.hipHostMalloc(C.byref(p), n, 0) == 0
        C.memset(p, 0xA5, n)
        return p

    def device(self, n):
        p = C.c_void_p()
        assert self.rt.hipMalloc(C.byref(p), n) == 0
        assert self.rt.hipMemset(p, 0xA5, n) == 0
        return p

    def to_host(self, dev, n):
        buf = (C.c_uint8 * n)()
        assert self.rt.hipMemcpy(buf, dev, n, 2) == 0  # hipMemcpyDeviceToHost
        return buf


def _arrays(r, base, host_base):
    """The five result arrays of a Results struct whose pointers lie in the block at `base`, read from its host copy."""
    n, n_refs = r.n_b64, int(r.n_ref[0]) + int(r.n_ref[1])

    def view(ptr, dtype, count):
        off = C.cast(ptr, C.c_void_p).value - base
        return np.frombuffer((C.c_uint8 * (count * np.dtype(dtype).itemsize)).from_address(host_base + off), dtype).copy()

    sr = view(r.search_results, svtme.SEARCH_RESULT_DTYPE, n * 8).reshape(n, 2, 4)
    out = {"zz_sad": view(r.zz_sad, "<u4", n * 8).reshape(n, 2, 4),
           "sr_divisor": view(r.reduce_me_sr_divisor, "<u4", n * 8).reshape(n, 2, 4),
           "planes_sad": view(r.p_sb_best_sad, "<u4", n_refs * n * 85).reshape(n_refs, n, 85),
           "planes_mv": view(r.p_sb_best_mv, "<u4", n_refs * n * 85).reshape(n_refs, n, 85)}
    for f in svtme.SEARCH_RESULT_FIELDS.names:
        out[f] = sr[f]
    return out


def _same(eng, hip, pic, n_b64, n_refs, where, device_block):
    want = eng.run_picture(pic)
    nbytes = svtme.static_result_bytes(n_b64, n_refs)
    block = hip.device(nbytes) if device_block else hip.pinned(nbytes)
    try:
        r = eng.search_static(pic, block, nbytes)
        assert r.n_b64 == n_b64 and int(r.n_ref[0]) + int(r.n_ref[1]) == n_refs
        if device_block:
            host = hip.to_host(block, nbytes)
            got = _arrays(r, block.value, C.addressof(host))
        else:
            got = _arrays(r, block.value, block.value)
    finally:
        (hip.rt.hipFree if device_block else hip.rt.hipHostFree)(block)
    n_ref = [int(r.n_ref[0]), int(r.n_ref[1])]
    for l in range(2):
        for i in range(n_ref[l]):
            q = (n_ref[0] if l else 0) + i
            assert np.array_equal(got["planes_sad"][q], want["best_sad"][:, l, i]), (where, "best_sad", l, i)
            assert np.array_equal(got["planes_mv"][q], want["best_mv"][:, l, i]), (where, "best_mv", l, i)
    for f in ("zz_sad", "sr_divisor"):
        assert np.array_equal(got[f], want[f]), (where, f)
    for f in svtme.SEARCH_RESULT_FIELDS.names:
        assert np.array_equal(got[f], want["search_results"][f]), (where, f)


def test_static_scratch_entry_point_matches_submit_and_wait_on_two_calls_in_a_row():
    hip = _Hip()
    big, _, _ = support.make_job(448, 320, 9, 3, seed=91, motion=(5, -3))
    small, _, _ = support.make_job(192, 128, 9, 3, seed=92, motion=(-4, 2))
    eng = svtme.Engine(448, 320, store_slots=20, max_in_flight=2)
    try:
        for i in range(9):
            eng.store_put(100 + i, i, big[i])
            eng.store_put(200 + i, i, small[i])
        # first call: 35 blocks, 4 + 3 references, HME levels 0-2, result block in pinned host memory
        ctrl = svtme.derive_controls(3, 448, 320)
        pic = eng.make_picture(448, 320, 4, 2, 1, [[3, 2, 1, 0], [5, 6, 8]], ctrl, key_of=lambda poc: 100 + poc)
        _same(eng, hip, pic, 35, 7, "448x320 preset 3", device_block=False)
        # second call, smaller input: 6 blocks, 2 + 1 references of a fast preset, result block in device memory; then
        # the first picture again: whatever the smaller call left in the array must not show
        ctrl8 = svtme.derive_controls(8, 192, 128)
        pic2 = eng.make_picture(192, 128, 4, 0, 1, [[0, 2], [8]], ctrl8, key_of=lambda poc: 200 + poc)
        _same(eng, hip, pic2, 6, 3, "192x128 preset 8", device_block=True)
        _same(eng, hip, pic, 35, 7, "448x320 preset 3 again", device_block=True)
        # a temporal-filter pair and a first-pass picture through the same entry point
        tf = eng.make_picture(192, 128, 4, 0, 1, [[3], []], svtme.derive_tf_controls(0), key_of=lambda poc: 200 + poc, tf_me_exit_th=0)
        _same(eng, hip, tf, 6, 1, "192x128 temporal filter", device_block=False)
        fp = eng.make_picture(192, 128, 4, 0, 1, [[3, 0], []], svtme.derive_first_pass_controls(192, 128), key_of=lambda poc: 200 + poc,
                              first_pass_bypass_zz=0)
        _same(eng, hip, fp, 6, 2, "192x128 first pass", device_block=False)
    finally:
        eng.close()


def test_static_scratch_entry_point_rejects_what_it_cannot_hold():
    eng = svtme.Engine(3904, 2176, store_slots=2, max_in_flight=1)
    hip = _Hip()
    block = hip.pinned(1 << 16)
    try:
        frame = np.zeros((2176, 3904), np.uint8)
        eng.store_put(1, 0, frame)
        eng.store_put(2, 1, frame)
        ctrl = svtme.derive_controls(8, 3904, 2176)
        pic = eng.make_picture(3904, 2176, 1, 0, 1, [[0], []], ctrl, key_of=lambda poc: poc + 1)
        with pytest.raises(svtme.SvtmeError) as ei:  # 61 x 34 = 2074 blocks > SVTME_STATIC_MAX_B64
            eng.search_static(pic, block, 1 << 30)
        assert ei.value.code == svtme.ERR_BAD_ARG
    finally:
        hip.rt.hipHostFree(block)
        eng.close()
    eng = svtme.Engine(128, 128, store_slots=2, max_in_flight=1)
    block = hip.pinned(1 << 16)
    try:
        frame = np.zeros((128, 128), np.uint8)
        eng.store_put(1, 0, frame)
        eng.store_put(2, 1, frame)
        ctrl = svtme.derive_controls(8, 128, 128)
        pic = eng.make_picture(128, 128, 1, 0, 1, [[0], []], ctrl, key_of=lambda poc: poc + 1)
        with pytest.raises(svtme.SvtmeError) as ei:  # result block too small
            eng.search_static(pic, block, svtme.static_result_bytes(4, 1) - 1)
        assert ei.value.code == svtme.ERR_BAD_ARG
        # pageable host memory (numpy's) is neither device memory nor page-locked: refused, not a GPU memory fault
        pageable = np.zeros(svtme.static_result_bytes(4, 1) + 64, np.uint8)
        addr = (pageable.ctypes.data + 63) & ~63
        with pytest.raises(svtme.SvtmeError) as ei:
            eng.search_static(pic, C.c_void_p(addr), svtme.static_result_bytes(4, 1))
        assert ei.value.code == svtme.ERR_BAD_ARG
        r = eng.search_static(pic, block, 1 << 16)  # and the same picture into the page-locked block goes through
        assert r.n_b64 == 4
        with pytest.raises(svtme.SvtmeError) as ei:  # the device tail is not part of this form
            pic.tail = svtme.Tail(enable=1, enable_me_8x8=1, enable_me_16x16=1, max_refs=1, max_cand=1, max_l0=1)
            eng.search_static(pic, block, 1 << 16)
        assert ei.value.code == svtme.ERR_UNSUPPORTED
    finally:
        hip.rt.hipHostFree(block)
        eng.close()
