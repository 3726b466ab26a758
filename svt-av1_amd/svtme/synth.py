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


def _canvas(seed, hh, ww):
    canvas = (_octave(seed * 7 + 1, hh, ww, 64) * 2 + _octave(seed * 7 + 2, hh, ww, 16) * 3 +
              _octave(seed * 7 + 3, hh, ww, 4) * 2 + _raw_bytes(seed * 7 + 4, hh * ww).reshape(hh, ww)) // 8
    canvas = canvas.astype(np.int32)
    # contrast stretch around the mean so that SADs are not tiny
    return np.clip((canvas - 128) * 3 + 128, 0, 255)


def synth_sequence(width, height, n_frames, seed=1, motion=(3, -2), noise=2, bit_depth=8,
                   n_objects=6):
    """Returns uint8 array [n_frames, height, width]: the 8-bit luma planes the ME sees."""
    dx, dy = int(motion[0]), int(motion[1])
    m = 8 + max(abs(dx), abs(dy)) * max(n_frames, 1)
    hh, ww = height + 2 * m, width + 2 * m
    canvas = _canvas(seed, hh, ww)
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


def warped_sequence(width, height, n_frames, seed=1, rot=(40, 0), zoom=(200, 0), shift=(2, -1), noise=1, detail=4):
    """Frames of one textured scene under a global motion that grows with the frame index, for the global-motion
    estimation: frame k samples the canvas at A_k * (x - centre) + centre + k * shift with
        A_k = [[1 + k * zoom[0] / 65536, k * (rot[1] - rot[0]) / 65536], [k * rot[0] / 65536, 1 + k * (zoom[0] + zoom[1]) / 65536]]
    i.e. per frame a turn by rot[0] / 65536 radians (small-angle form), a zoom by zoom[0] / 65536 and a move by `shift` pixels;
    rot[1] and zoom[1] add shear and anisotropy, which make the motion affine rather than a rotation-zoom. Bilinear
    resampling in 16.16 / 8-bit fixed point, integer arithmetic only (the same bytes on every machine). `detail` blurs the
    canvas first (box of that size), so that corners survive the resampling. Returns uint8 [n_frames, height, width]."""
    m = 32 + int((abs(rot[0]) + abs(rot[1]) + abs(zoom[0]) + abs(zoom[1])) * n_frames * max(width, height) / 65536 / 2) + \
        (abs(shift[0]) + abs(shift[1])) * n_frames
    hh, ww = height + 2 * m, width + 2 * m
    canvas = _canvas(seed, hh, ww).astype(np.int64)
    if detail > 1:
        canvas = _box(canvas, detail).astype(np.int64)
        canvas = np.clip((canvas - 128) * 2 + 128, 0, 255)
    ys, xs = np.mgrid[0:height, 0:width].astype(np.int64)
    cx, cy = xs - width // 2, ys - height // 2
    frames = np.empty((n_frames, height, width), np.uint8)
    for k in range(n_frames):
        a = 65536 + k * zoom[0]
        b = -k * rot[0] + k * rot[1]
        c = k * rot[0]
        d = 65536 + k * zoom[0] + k * zoom[1]
        u = a * cx + b * cy + ((width // 2 + m + k * shift[0]) << 16)
        v = c * cx + d * cy + ((height // 2 + m + k * shift[1]) << 16)
        ui, vi = u >> 16, v >> 16
        fu, fv = (u >> 8) & 255, (v >> 8) & 255
        ui = np.clip(ui, 0, ww - 2)
        vi = np.clip(vi, 0, hh - 2)
        p00, p01 = canvas[vi, ui], canvas[vi, ui + 1]
        p10, p11 = canvas[vi + 1, ui], canvas[vi + 1, ui + 1]
        top = p00 * (256 - fu) + p01 * fu
        bot = p10 * (256 - fu) + p11 * fu
        f = (top * (256 - fv) + bot * fv + 32768) >> 16
        if noise:
            nz = _raw_bytes(seed * 1000003 + k + 4711, height * width).reshape(height, width).astype(np.int64)
            f = f + (nz % (2 * noise + 1)) - noise
        frames[k] = np.clip(f, 0, 255).astype(np.uint8)
    return frames
