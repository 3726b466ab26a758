"""Seeded synthetic 8-bit luma sequences for the ME engine, its tests and bench.py.

Integer arithmetic on a raw PCG64 bit stream only, so the same seed gives the same bytes on every
machine and numpy version. A sequence is a smooth multi-octave texture canvas translated by a global
motion per frame, a few rectangles that move on their own (so different 64x64 blocks want different
vectors and neither the zero-MV early exits nor the HME pruning are taken everywhere), plus small
i.i.d. noise per frame. 10-bit input is reduced to the 8-bit MSB plane the ME reads (pixel >> 2), as
svt_unpack_and_2bcompress does (Source/Lib/Common/C_DEFAULT/EbPackUnPack_C.c:93-94).
"""
import numpy as np


def _raw_bytes(seed, n):
    bg = np.random.PCG64(int(seed))
    words = bg.random_raw((n + 7) // 8)
    return words.view(np.uint8)[:n].copy()


def _box(a, k):
    """Separable k x k box mean, integer arithmetic, edge-replicated."""
    if k <= 1:
        return a.astype(np.uint32)
    p = k // 2
    x = np.pad(a.astype(np.uint32), ((p, k - 1 - p), (p, k - 1 - p)), mode="edge")
    c = np.cumsum(x, axis=0, dtype=np.uint64)
    c = np.vstack([np.zeros((1, c.shape[1]), np.uint64), c])
    x = (c[k:] - c[:-k])
    c = np.cumsum(x, axis=1, dtype=np.uint64)
    c = np.hstack([np.zeros((c.shape[0], 1), np.uint64), c])
    x = (c[:, k:] - c[:, :-k])
    return (x // (k * k)).astype(np.uint32)


def _octave(seed, hh, ww, scale):
    gh, gw = hh // scale + 2, ww // scale + 2
    g = _raw_bytes(seed, gh * gw).reshape(gh, gw)
    up = np.repeat(np.repeat(g, scale, axis=0), scale, axis=1)[:hh, :ww]
    return _box(up, scale)


def synth_sequence(width, height, n_frames, seed=1, motion=(3, -2), noise=2, bit_depth=8,
                   n_objects=6):
    """Returns uint8 array [n_frames, height, width]: the 8-bit luma planes the ME sees."""
    dx, dy = int(motion[0]), int(motion[1])
    m = 8 + max(abs(dx), abs(dy)) * max(n_frames, 1)
    hh, ww = height + 2 * m, width + 2 * m
    canvas = (_octave(seed * 7 + 1, hh, ww, 64) * 2 + _octave(seed * 7 + 2, hh, ww, 16) * 3 +
              _octave(seed * 7 + 3, hh, ww, 4) * 2 + _raw_bytes(seed * 7 + 4, hh * ww).reshape(hh, ww)) // 8
    canvas = canvas.astype(np.int32)
    # contrast stretch around the mean so that SADs are not tiny
    canvas = np.clip((canvas - 128) * 3 + 128, 0, 255)
    par = _raw_bytes(seed * 7 + 5, 16 * max(n_objects, 1)).astype(np.int32).reshape(-1, 16)
    frames = np.empty((n_frames, height, width), np.uint8)
    scale10 = 4 if bit_depth == 10 else 1
    for k in range(n_frames):
        oy, ox = m + k * dy, m + k * dx
        f = canvas[oy:oy + height, ox:ox + width].copy()
        for o in range(n_objects):
            p = par[o]
            ow = min(24 + (p[0] % 96), max(8, width // 2))
            oh = min(24 + (p[1] % 96), max(8, height // 2))
            vx = (p[2] % 15) - 7
            vy = (p[3] % 9) - 4
            x0 = (p[4] * 256 + p[5]) % max(1, width - ow) + vx * k
            y0 = (p[6] * 256 + p[7]) % max(1, height - oh) + vy * k
            x1, y1 = min(width, x0 + ow), min(height, y0 + oh)
            x0c, y0c = max(0, x0), max(0, y0)
            if x1 <= x0c or y1 <= y0c:
                continue
            # the object carries its own texture: a shifted view of the canvas
            sy = (m + (p[8] % 64) + (y0c - y0)) % max(1, hh - oh)
            sx = (m + (p[9] % 64) + (x0c - x0)) % max(1, ww - ow)
            patch = canvas[sy:sy + (y1 - y0c), sx:sx + (x1 - x0c)]
            f[y0c:y1, x0c:x1] = 255 - patch
        if noise:
            nz = _raw_bytes(seed * 1000003 + k + 11, height * width).reshape(height, width).astype(np.int32)
            f = f + (nz % (2 * noise + 1)) - noise
        f = np.clip(f, 0, 255)
        if bit_depth == 10:
            # a 10-bit picture with detail of its own below the 8-bit step (+-6 of 1023), then the MSB plane the ME reads
            fine = _raw_bytes(seed * 1000003 + k + 977, height * width).reshape(height, width).astype(np.int32)
            f10 = np.clip(f * scale10 + (fine % 13) - 6, 0, 1023)
            f = f10 >> 2
        frames[k] = f.astype(np.uint8)
    return frames


def extreme_sequence(width, height, n_frames, kind, seed=1):
    """Stress inputs in the spirit of the reference's SAD test patterns (test/SadTest.cc:92,205-251):
    'noise' = i.i.d. bytes, 'max' = alternating all-0 / all-255 frames, 'flat' = constant 128."""
    if kind == "noise":
        return _raw_bytes(seed + 99, n_frames * height * width).reshape(n_frames, height, width)
    if kind == "max":
        f = np.zeros((n_frames, height, width), np.uint8)
        f[1::2] = 255
        return f
    if kind == "flat":
        return np.full((n_frames, height, width), 128, np.uint8)
    raise ValueError(kind)
