"""CPU-only checks of the product's host side: the C-ABI library loads and exports what include/svtme.h declares,
rejects bad arguments without a GPU, and its preset -> controls derivation equals the reference's
(golden controls dumped from signal_derivation_me_kernel_oq) and the oracle's for every preset."""
import ctypes as C
import glob
import os
import re

import numpy as np
import pytest

import refio
import svtme

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_library_exports_every_declared_symbol():
    lib = svtme.load_library()
    header = open(os.path.join(ROOT, "include", "svtme.h")).read()
    declared = set(re.findall(r"\b(svtme_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(svtme.EXPORTS), declared ^ set(svtme.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), "libsvtme.so does not export " + name
    assert lib.svtme_version().startswith(b"svtme")


def test_struct_sizes_match_the_header():
    assert C.sizeof(svtme.Controls) == 84 == refio.CONTROLS.itemsize
    assert C.sizeof(svtme.SearchResult) == 24
    assert C.sizeof(svtme.Tail) == 16
    # keys/poc/geometry, ref keys, ref pocs, controls, device-tail parameters, tf_me_exit_th
    assert C.sizeof(svtme.Picture) == 32 + 64 + 64 + 84 + 16 + 4 + 1 + 3 + 4  # .. tf_me_exit_th, b64_step, reserved, padding to 8


def test_bad_arguments_are_rejected_without_touching_a_device():
    lib = svtme.load_library()
    h = C.c_void_p()
    cfg = svtme.Config(0, 100, 100, 4, 2)  # not a multiple of 8
    assert lib.svtme_create(C.byref(cfg), C.byref(h)) == svtme.ERR_BAD_ARG
    assert lib.svtme_create(None, C.byref(h)) == svtme.ERR_BAD_ARG
    assert lib.svtme_derive_controls(None, None) == svtme.ERR_BAD_ARG
    key = svtme.PresetKey(14, 0, 4, 0, 30 << 16)
    c = svtme.Controls()
    assert lib.svtme_derive_controls(C.byref(key), C.byref(c)) == svtme.ERR_BAD_ARG
    assert lib.svtme_store_release(None, 1) == svtme.ERR_BAD_ARG
    assert lib.svtme_unpack_b64(None, 0, None, None, None, None, None) == svtme.ERR_BAD_ARG
    assert lib.svtme_tpl_search(None, None, None) == svtme.ERR_BAD_ARG
    assert lib.svtme_tpl_recon(None, None, None, None, None, 0) == svtme.ERR_BAD_ARG
    assert lib.svtme_tpl_recon_keep(None, 0, 0) == svtme.ERR_BAD_ARG
    assert lib.svtme_kernel_timing(None, 1) == svtme.ERR_BAD_ARG


def test_engine_creation_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(svtme.SvtmeError) as ei:
        svtme.Engine(640, 384)
    assert ei.value.code == svtme.ERR_DEVICE


def test_input_resolution_classes():
    # derive_input_resolution thresholds, EbDefinitions.h:2025-2030
    for (w, h), cls in (((320, 192), 0), ((640, 360), 1), ((704, 512), 2), ((1280, 720), 3), ((1920, 1080), 4),
                        ((3840, 2160), 5), ((7680, 4320), 6)):
        assert svtme.input_resolution(w, h) == cls


def _as_np(ctrl):
    return np.frombuffer(bytes(ctrl), refio.CONTROLS)[0]


def test_derive_controls_equals_reference_goldens():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n = 0
    for name, c in mod.CASES.items():
        g = np.load(os.path.join(HERE, "golden", name + ".npz"))
        mine = _as_np(svtme.derive_controls(c["enc_mode"], c["width"], c["height"], c.get("sc_class1", 0),
                                            c.get("frame_rate", 30 << 16)))
        for pd in g["pic"]:
            assert mine.tobytes() == pd["ctrl"].tobytes(), (name, mine, pd["ctrl"])
            n += 1
    assert n > 50


def test_derive_controls_equals_oracle_for_every_preset():
    orc = C.CDLL(os.path.join(ROOT, "oracle", "build", "libme_oracle.so"))
    orc.orc_derive_controls.argtypes = [C.POINTER(svtme.PresetKey), C.POINTER(svtme.Controls)]
    lib = svtme.load_library()
    n = 0
    for m in range(-2, 14):
        for sc in (0, 1):
            for res in range(7):
                for fps in (24, 30, 50, 60):
                    key = svtme.PresetKey(m, sc, res, 0, fps << 16)
                    a, b = svtme.Controls(), svtme.Controls()
                    assert lib.svtme_derive_controls(C.byref(key), C.byref(a)) == 0
                    assert orc.orc_derive_controls(C.byref(key), C.byref(b)) == 0
                    assert bytes(a) == bytes(b), (m, sc, res, fps, _as_np(a), _as_np(b))
                    n += 1
    assert n == 16 * 2 * 7 * 4


def test_derive_tf_controls_equals_reference_goldens_and_oracle():
    """Temporal-filter controls: golden controls dumped after tf_signal_derivation_me_kernel_oq + the per-block
    FULL_SAD_SEARCH override, for every hme_me_level, and the oracle's restatement."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    levels = set()
    for name, c in mod.TF_CASES.items():
        g = np.load(os.path.join(HERE, "golden", name + ".npz"))
        mine = _as_np(svtme.derive_tf_controls(c["level"]))
        for pd in g["pic"]:
            assert mine.tobytes() == pd["ctrl"].tobytes(), (name, mine, pd["ctrl"])
        levels.add(c["level"])
    assert levels == {0, 1, 2}
    orc = C.CDLL(os.path.join(ROOT, "oracle", "build", "libme_oracle.so"))
    orc.orc_derive_tf_controls.argtypes = [C.c_uint8, C.POINTER(svtme.Controls)]
    for lvl in range(3):
        b = svtme.Controls()
        assert orc.orc_derive_tf_controls(lvl, C.byref(b)) == 0
        assert bytes(svtme.derive_tf_controls(lvl)) == bytes(b)
    with pytest.raises(svtme.SvtmeError):
        svtme.derive_tf_controls(3)


def test_derive_first_pass_controls_equals_reference_goldens_and_oracle():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for name, c in mod.FP_CASES.items():
        g = np.load(os.path.join(HERE, "golden", name + ".npz"))
        mine = _as_np(svtme.derive_first_pass_controls(c["width"], c["height"], c.get("frame_rate", 30 << 16), c["reduce_me"]))
        for pd in g["pic"]:
            assert mine.tobytes() == pd["ctrl"].tobytes(), (name, mine, pd["ctrl"])
    orc = C.CDLL(os.path.join(ROOT, "oracle", "build", "libme_oracle.so"))
    orc.orc_derive_first_pass_controls.argtypes = [C.c_uint8, C.c_uint32, C.c_uint8, C.POINTER(svtme.Controls)]
    lib = svtme.load_library()
    for res in range(7):
        for fps in (24, 60):
            for red in (0, 1):
                a, b = svtme.Controls(), svtme.Controls()
                assert lib.svtme_derive_first_pass_controls(res, fps << 16, red, C.byref(a)) == 0
                assert orc.orc_derive_first_pass_controls(res, fps << 16, red, C.byref(b)) == 0
                assert bytes(a) == bytes(b), (res, fps, red)


def test_header_is_plain_c_and_matches_the_ctypes_mirror(tmp_path):
    """include/svtme.h compiles as C99 (the reference is C) and its struct sizes / field offsets are the ones the
    ctypes mirror uses."""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('''
#include <stdio.h>
#include <stddef.h>
#include "svtme.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(svtme_controls), sizeof(svtme_picture), sizeof(svtme_tail),
           sizeof(svtme_search_result), sizeof(svtme_results), sizeof(svtme_config), sizeof(svtme_preset_key),
           sizeof(svtme_sa_minmax));
    printf("%zu %zu %zu %zu %zu %zu\\n", offsetof(svtme_picture, ref_key), offsetof(svtme_picture, ctrl),
           offsetof(svtme_picture, tail), offsetof(svtme_picture, tf_me_exit_th), offsetof(svtme_results, n_ref),
           offsetof(svtme_search_result, hme_sad));
    printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(svtme_tpl_picture), sizeof(svtme_tpl_block), offsetof(svtme_tpl_picture, width),
           offsetof(svtme_tpl_picture, n_ref), offsetof(svtme_tpl_picture, me_mv_array), offsetof(svtme_tpl_block, best_rf_idx));
    return 0;
}
''')
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src),
                    "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    got = [int(v) for v in out]
    want = [C.sizeof(svtme.Controls), C.sizeof(svtme.Picture), C.sizeof(svtme.Tail), C.sizeof(svtme.SearchResult),
            C.sizeof(svtme.Results), C.sizeof(svtme.Config), C.sizeof(svtme.PresetKey), C.sizeof(svtme.SaMinMax),
            svtme.Picture.ref_key.offset, svtme.Picture.ctrl.offset, svtme.Picture.tail.offset,
            svtme.Picture.tf_me_exit_th.offset, svtme.Results.n_ref.offset, svtme.SearchResult.hme_sad.offset,
            C.sizeof(svtme.TplPicture), svtme.TPL_BLOCK.itemsize, svtme.TplPicture.width.offset, svtme.TplPicture.n_ref.offset,
            svtme.TplPicture.me_mv_array.offset, svtme.TPL_BLOCK.fields["best_rf_idx"][1]]
    assert got == want, (got, want)
