"""Random-access / low-delay picture lists for the ME engine's tests and bench.

The prediction structure itself is outside the ME path (picture decision chooses it,
Source/Lib/Encoder/Codec/EbPictureDecisionProcess.c:6691,6726); the engine only sees, per picture, its
temporal layer, reference flag and the two reference lists. This module produces lists shaped like the
reference's 5-layer hierarchy: the deltas below are the POC distances of its mini-GOP-16 configuration
(five_level_hierarchical_pred_struct, Source/Lib/Encoder/Codec/EbPredictionStructure.c:257-360) and the per
layer reference counts follow set_mrp_ctrl (Source/Lib/Encoder/Globals/EbEncHandle.c:2953-3033).
"""
from dataclasses import dataclass, field
from typing import List

# gop index -> (temporal layer, list0 deltas, list1 deltas); ref POC = POC - delta
_MG16 = {
    0: (0, (16, 48, 0, 0), (16, 32, 0, 0)),
    1: (4, (1, 9, 8, 17), (-1, -3, -7, 0)),
    2: (3, (2, 4, 10, 18), (-2, -6, -14, 0)),
    3: (4, (1, 3, 2, 11), (-1, -5, -13, 0)),
    4: (2, (4, 8, 12, 20), (-4, -12, 0, 0)),
    5: (4, (1, 5, 4, 13), (-1, -3, -11, 0)),
    6: (3, (2, 4, 6, 14), (-2, -10, 0, 0)),
    7: (4, (1, 3, 6, 7), (-1, -9, 0, 0)),
    8: (1, (8, 16, 24, 0), (-8, 12, 10, 0)),
    9: (4, (1, 9, 8, 17), (-1, -3, -7, 0)),
    10: (3, (2, 4, 10, 18), (-2, -6, 0, 0)),
    11: (4, (1, 3, 2, 11), (-1, -5, 0, 0)),
    12: (2, (4, 8, 12, 6), (-4, 11, 0, 0)),
    13: (4, (1, 5, 4, 13), (-1, -3, 0, 0)),
    14: (3, (2, 4, 6, 14), (-2, 5, 0, 0)),
    15: (4, (1, 3, 6, 7), (-1, 15, 0, 0)),
}
_DECODE_ORDER = (16, 8, 4, 2, 1, 3, 6, 5, 7, 12, 10, 9, 11, 14, 13, 15)

# mrp level -> (base l0, base l1, non-base l0, non-base l1), non screen content
_MRP = {0: (1, 1, 1, 1), 1: (4, 3, 4, 3), 2: (4, 3, 4, 3), 3: (4, 3, 4, 3), 4: (2, 2, 2, 2), 5: (2, 2, 1, 1)}


def mrp_level(enc_mode):
    """MRP level per preset, Source/Lib/Encoder/Globals/EbEncHandle.c:3531-3556."""
    if enc_mode <= -2:
        return 1
    if enc_mode <= 1:
        return 2
    if enc_mode <= 3:
        return 3
    if enc_mode <= 6:
        return 4
    if enc_mode <= 11:
        return 5
    return 0


def mrp_counts(enc_mode):
    return _MRP[mrp_level(enc_mode)]


@dataclass
class Picture:
    poc: int
    temporal_layer: int
    is_ref: int
    refs: List[List[int]] = field(default_factory=lambda: [[], []])  # POCs per list

    @property
    def n_list(self):
        return 2 if self.refs[1] else 1


def random_access(n_frames, enc_mode):
    """Pictures of a mini-GOP-16 random-access stream in decode order (the intra picture 0 has no ME)."""
    b0, b1, n0, n1 = mrp_counts(enc_mode)
    pics = []
    base = 0
    while base + 1 < n_frames:
        for g in _DECODE_ORDER:
            poc = base + g
            if poc >= n_frames:
                continue
            layer, d0, d1 = _MG16[g % 16]
            c0, c1 = (b0, b1) if layer == 0 else (n0, n1)
            lists = []
            for deltas, cnt in ((d0, c0), (d1, c1)):
                l = []
                for d in deltas:
                    if d == 0:
                        continue
                    r = poc - d
                    if r < 0 or r >= n_frames or r == poc or r in l:
                        continue
                    # a reference must precede the picture in decode order
                    if r > poc and not _decoded_before(r, poc, base):
                        continue
                    l.append(r)
                lists.append(l[:cnt])
            if not lists[0]:
                lists[0] = [max(0, base)]
            if not lists[1]:
                # base layer: list 1 repeats list 0 (the reference mirrors list 0 there,
                # EbMotionEstimation.c:1811-1819); elsewhere fall back to the closest past picture
                lists[1] = list(lists[0][:max(1, c1)])
            pics.append(Picture(poc, layer, 1 if layer < 4 else 0, lists))
        base += 16
    return pics


def _decoded_before(r, poc, base):
    order = {base + g: i for i, g in enumerate(_DECODE_ORDER)}
    return order.get(r, -1) < order.get(poc, 1 << 30) and r in order


def low_delay(n_frames, enc_mode, n_list=1):
    """Low-delay P (pred-struct 0) / B (pred-struct 1) stream: every picture references the previous one(s)."""
    b0, b1, n0, n1 = mrp_counts(enc_mode)
    pics = []
    for poc in range(1, n_frames):
        layer = 0 if poc % 4 == 0 else (1 if poc % 2 == 0 else 2)
        c0 = b0 if layer == 0 else n0
        l0 = [p for p in (poc - 1, poc - 2, poc - 3, poc - 4) if p >= 0][:c0]
        l1 = list(l0[:max(1, b1 if layer == 0 else n1)]) if n_list == 2 else []
        pics.append(Picture(poc, layer, 1, [l0, l1]))
    return pics
