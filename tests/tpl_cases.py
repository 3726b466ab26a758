"""The TPL dispenser jobs behind tests/golden/tpl_*.npz: seeded synthetic sequences, the preset whose ME feeds the
dispenser, the TPL level (set_tpl_extended_controls, EbResourceCoordinationProcess.c:161-341; presets map to levels in
get_tpl_level, EbEncHandle.c:3098-3125) and the quantiser."""
CASES = {
    # 16x16 blocks, SAD costs (level 4: M7-M9), ragged height
    "tpl4_m8_200x136": dict(width=200, height=136, n_frames=9, enc_mode=8, seed=81, structure="ra", level=4, qp=32),
    # level 5 (M10): as level 4 for the search, half of the transform rows afterwards
    "tpl5_m10_264x136": dict(width=264, height=136, n_frames=9, enc_mode=10, seed=82, structure="ra", level=5, qp=40),
    # level 7 (M11 and faster): 32x32 blocks in complete 64x64 blocks, 16x16 in the ragged right column and bottom row
    "tpl7_m11_328x200": dict(width=328, height=200, n_frames=9, enc_mode=11, seed=83, structure="ra", level=7, qp=35),
    "tpl7_m13_256x128": dict(width=256, height=128, n_frames=9, enc_mode=13, seed=84, structure="ra", level=7, qp=50),
    # noise: nothing predicts, the costs are large and close; flat: every cost is 0 and the first candidate / DC wins
    "tpl4_m8_noise_192x128": dict(width=192, height=128, n_frames=5, enc_mode=8, seed=85, structure="ra", level=4, qp=32, kind="noise"),
    "tpl4_m9_flat_128x128": dict(width=128, height=128, n_frames=5, enc_mode=9, seed=86, structure="ra", level=4, qp=32, kind="flat"),
    # low delay, one list; large motion so that vectors reach the clamp at the picture border
    "tpl4_m7_ldp_256x192": dict(width=256, height=192, n_frames=6, enc_mode=7, seed=87, structure="ldp", level=4, qp=28, motion=(14, -11)),
    # transform-domain costs (level 3: M6; level 2 is level 3 with intra prediction in every picture), whole 16x16 DCT
    "tpl3_m6_192x128": dict(width=192, height=128, n_frames=5, enc_mode=6, seed=88, structure="ra", level=3, qp=32),
    "tpl2_m6_200x136": dict(width=200, height=136, n_frames=9, enc_mode=6, seed=91, structure="ra", level=2, qp=20),
    "tpl3_m6_noise_128x128": dict(width=128, height=128, n_frames=5, enc_mode=6, seed=92, structure="ra", level=3, qp=55, kind="noise"),
    # level 1 (M5 and slower, tpl_opt_flag 0): thirteen intra modes with filtered edges, transform-domain costs throughout and
    # a rate estimate of the chosen candidate; ragged sizes put blocks on the right and bottom edges, where the neighbour
    # arrays are padded; noise makes the directional and smooth modes win often, flat makes every cost tie
    "tpl1_m3_128x128": dict(width=128, height=128, n_frames=5, enc_mode=3, seed=89, structure="ra", level=1, qp=32),
    "tpl1_m5_200x136": dict(width=200, height=136, n_frames=9, enc_mode=5, seed=93, structure="ra", level=1, qp=24, motion=(6, -4)),
    "tpl1_m4_noise_136x200": dict(width=136, height=200, n_frames=5, enc_mode=4, seed=94, structure="ra", level=1, qp=45, kind="noise"),
    "tpl1_m2_flat_128x64": dict(width=128, height=64, n_frames=3, enc_mode=2, seed=95, structure="ra", level=1, qp=32, kind="flat"),
}
# (levels the restatement and the engine refuse would be listed here with the reference's record of them)
UNSUPPORTED_CASES = {}
# the benchmark's 1080p preset-8 shape: a digest of the reference's results only
DIGEST_CASES = {
    "tpl4_m8_1920x1080": dict(width=1920, height=1080, n_frames=5, enc_mode=8, seed=90, structure="ra", level=4, qp=32, motion=(5, -3)),
}
SEARCH_FIELDS = ("valid", "best_intra_cost", "best_inter_cost", "mv_x", "mv_y", "best_rf_idx", "best_is_inter", "best_intra_mode",
                 "src_recon_error", "src_rate")


STAT_FIELDS = ("srcrf_dist", "recrf_dist", "srcrf_rate", "recrf_rate")


def job_args(c):
    return {k: c[k] for k in ("width", "height", "n_frames", "enc_mode", "seed", "structure", "kind", "motion") if k in c}
