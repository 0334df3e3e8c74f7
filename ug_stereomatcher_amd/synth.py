"""Seeded synthetic stereo pairs.

The reference ships no sample images (`.MISSING_LARGE_BLOBS` lists left.tif/right.tif),
so every test, golden fixture and bench run draws its input from this generator.

Left image: multi-octave value noise + sparse random dots, three channels with
different seeds, quantised to uint8 in [1, 255] (no all-zero 5x5 patch, so the
squared-NCC denominator is never 0).  Ground truth: the match of left pixel (x, y)
is right (x + dx, y + dy) -- the reference's convention, getPointCloud.cpp:910-913 --
with
    dx(x, y) = a*W*(0.5 + 0.5*sin(2*pi*x/W)*cos(2*pi*y/H)) + b*x,   a = 0.01, b = 0.002
    dy(x, y) = 0.75*sin(2*pi*y/H)
Right image = left resampled bilinearly at (x - dx, y - dy), re-quantised.

numpy only; deterministic for a given (W, H, seed).
"""
from __future__ import annotations

import numpy as np

BASE_SEED = 20250310


def _value_noise(rng: np.random.Generator, W: int, H: int, cell: int) -> np.ndarray:
    """Bilinearly (smoothstep) interpolated random lattice with `cell`-pixel spacing, float32 in [0,1).
    Separable: interpolate the lattice rows along x first (small), then along y."""
    gw = W // cell + 2
    gh = H // cell + 2
    lat = rng.random((gh, gw), dtype=np.float32)
    xs = (np.arange(W, dtype=np.float32) + 0.5) / np.float32(cell)
    ys = (np.arange(H, dtype=np.float32) + 0.5) / np.float32(cell)
    x0 = np.floor(xs).astype(np.int64)
    y0 = np.floor(ys).astype(np.int64)
    fx = (xs - x0).astype(np.float32)
    fy = (ys - y0).astype(np.float32)
    # smoothstep keeps lattice lines from showing up as gradient discontinuities
    fx = fx * fx * (3 - 2 * fx)
    fy = fy * fy * (3 - 2 * fy)
    rows = lat[:, x0] * (1 - fx) + lat[:, x0 + 1] * fx  # (gh, W)
    out = np.take(rows, y0, axis=0)
    out *= (1 - fy)[:, None]
    out += np.take(rows, y0 + 1, axis=0) * fy[:, None]
    return out


def _texture(rng: np.random.Generator, W: int, H: int) -> np.ndarray:
    acc = np.zeros((H, W), dtype=np.float32)
    cell = 2
    total = 0.0
    while cell <= max(8, min(W, H) // 2) and cell <= 1024:
        amp = float(cell) ** 0.35
        acc += np.float32(amp) * _value_noise(rng, W, H, cell)
        total += amp
        cell *= 2
    acc /= np.float32(total)
    # sparse dots: 1.5 % of pixels pushed bright or dark
    dots = rng.random((H, W), dtype=np.float32)
    acc = np.where(dots < 0.0075, acc * 0.25, acc)
    acc = np.where(dots > 0.9925, 0.75 + acc * 0.25, acc)
    lo, hi = float(acc.min()), float(acc.max())
    return (acc - np.float32(lo)) / np.float32(max(hi - lo, 1e-6))


def truth_field(W: int, H: int, a: float = 0.01, b: float = 0.002, dy_amp: float = 0.75):
    x = np.arange(W, dtype=np.float64)[None, :]
    y = np.arange(H, dtype=np.float64)[:, None]
    sx = np.sin(2 * np.pi * x / W)  # separable: evaluate the trig on the axes only
    cy = np.cos(2 * np.pi * y / H)
    dx = a * W * (0.5 + 0.5 * (sx * cy)) + b * x
    dy = np.broadcast_to(dy_amp * np.sin(2 * np.pi * y / H), (H, W))
    return dx.astype(np.float32), np.ascontiguousarray(dy, dtype=np.float32)


def _bilinear(img: np.ndarray, sx: np.ndarray, sy: np.ndarray, block: int = 64) -> np.ndarray:
    """Bilinear resample of img at (sx, sy); processed in row blocks so the gathers stay in cache."""
    out = np.empty(np.broadcast_shapes(sx.shape, sy.shape), np.float32)
    for y in range(0, out.shape[0], block):
        ys = sy[y:y + block] if sy.shape[0] > 1 else sy
        out[y:y + block] = _bilinear_rows(img, sx[y:y + block] if sx.shape[0] > 1 else sx, ys)
    return out


def _bilinear_rows(img: np.ndarray, sx: np.ndarray, sy: np.ndarray) -> np.ndarray:
    H, W = img.shape
    sx = np.clip(sx, 0, W - 1)
    sy = np.clip(sy, 0, H - 1)
    x0 = np.floor(sx).astype(np.int32)
    y0 = np.floor(sy).astype(np.int32)
    fx = (sx - x0).astype(np.float32)
    fy = (sy - y0).astype(np.float32)
    x1 = np.minimum(x0 + 1, W - 1)
    y1 = np.minimum(y0 + 1, H - 1)
    flat = img.ravel()
    r0 = y0.astype(np.int64) * W
    r1 = y1.astype(np.int64) * W
    top = np.take(flat, r0 + x0)
    top += (np.take(flat, r0 + x1) - top) * fx
    bot = np.take(flat, r1 + x0)
    bot += (np.take(flat, r1 + x1) - bot) * fx
    top += (bot - top) * fy
    return top


def make_pair(W: int, H: int, seed: int = BASE_SEED, a: float = 0.01, b: float = 0.002,
              dy_amp: float = 0.75):
    """Returns (left, right, dx_true, dy_true): uint8 HxWx3 (rgb8, C-contiguous) x2, float32 HxW x2."""
    dx, dy = truth_field(W, H, a, b, dy_amp)
    xs = np.arange(W, dtype=np.float32)[None, :] - dx
    ys = np.arange(H, dtype=np.float32)[:, None] - dy
    left = np.empty((H, W, 3), dtype=np.uint8)
    right = np.empty((H, W, 3), dtype=np.uint8)

    def channel(ch: int):
        rng = np.random.Generator(np.random.PCG64(seed * 3 + ch))
        tex = _texture(rng, W, H)
        lq = np.clip(np.rint(1.0 + 254.0 * tex), 1, 255)
        left[:, :, ch] = lq.astype(np.uint8)
        # resample the quantised left so that a perfect matcher would see identical greys
        rr = _bilinear(lq.astype(np.float32), xs, ys)
        right[:, :, ch] = np.clip(np.rint(rr), 1, 255).astype(np.uint8)

    if W * H >= 1 << 20:  # channels are independent streams: run them on three threads
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(3) as ex:
            list(ex.map(channel, range(3)))
    else:
        for ch in range(3):
            channel(ch)
    return left, right, dx, dy
