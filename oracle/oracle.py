"""ctypes binding of oracle/libugsm_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this
module (as the checker).  The product package ug_stereomatcher_amd never does.
"parity unpinned" by the reference except for the blur (see ugsm_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libugsm_oracle.so")
_GOLD = os.path.join(_HERE, "_ref", "libgold.so")

_f32p = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int)


def build(force: bool = False) -> None:
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    src = os.path.join(_HERE, "ugsm_oracle.c")
    stale = (not os.path.exists(_LIB)) or os.path.getmtime(_LIB) < os.path.getmtime(src)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if os.path.isdir("/root/reference") and (force or not os.path.exists(_GOLD)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def gold_lib():
    """The reference's own convolutionRowCPU/ColumnCPU (oracle/_ref), or None."""
    if not os.path.exists(_GOLD):
        return None
    return C.CDLL(_GOLD)


def _fp(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_f32p)


def _up(a):
    assert a.dtype == np.uint8
    return a.ctypes.data_as(_u8p)


def set_num_threads(n: int) -> None:
    lib().orc_set_num_threads(int(n))


def num_threads() -> int:
    return int(lib().orc_num_threads())


def gauss_taps() -> np.ndarray:
    g = np.zeros(5, np.float32)
    lib().orc_gauss_taps(_fp(g))
    return g


def box_taps() -> np.ndarray:
    g = np.zeros(5, np.float32)
    lib().orc_box_taps(_fp(g))
    return g


def level_dims(W: int, H: int, levels: int = 14):
    w = (C.c_int * levels)()
    h = (C.c_int * levels)()
    rc = lib().orc_level_dims(W, H, levels, w, h)
    if rc:
        raise ValueError("pyramid level smaller than 1 px")
    return list(w), list(h)


def iterations_for_level(i: int) -> int:
    return int(lib().orc_iterations_for_level(i))


def smooth_passes_for_level(i: int) -> int:
    return int(lib().orc_smooth_passes_for_level(i))


def threshold_schedule(mi: int) -> np.ndarray:
    out = np.zeros(mi, np.float32)
    lib().orc_threshold_schedule(mi, _fp(out))
    return out


def rgb_to_planes(rgb: np.ndarray) -> np.ndarray:
    H, W, _ = rgb.shape
    out = np.empty((3, H, W), np.float32)
    lib().orc_rgb_to_planes(_up(rgb), W, H, rgb.strides[0], _fp(out))
    return out


def conv(src: np.ndarray, taps: np.ndarray, mode: str) -> np.ndarray:
    """Separable 5-tap conv, rows then columns; mode 'zero' or 'clamp'."""
    H, W = src.shape
    t = np.ascontiguousarray(taps, np.float32)
    tmp = np.empty_like(src)
    dst = np.empty_like(src)
    L = lib()
    if mode == "zero":
        L.orc_conv_rows_zero(_fp(tmp), _fp(src), W, H, _fp(t))
        L.orc_conv_cols_zero(_fp(dst), _fp(tmp), W, H, _fp(t))
    else:
        L.orc_conv_rows_clamp(_fp(tmp), _fp(src), W, H, _fp(t))
        L.orc_conv_cols_clamp(_fp(dst), _fp(tmp), W, H, _fp(t))
    return dst


def conv_rows_zero(src, taps):
    H, W = src.shape
    dst = np.empty_like(src)
    lib().orc_conv_rows_zero(_fp(dst), _fp(src), W, H, _fp(np.ascontiguousarray(taps, np.float32)))
    return dst


def conv_cols_zero(src, taps):
    H, W = src.shape
    dst = np.empty_like(src)
    lib().orc_conv_cols_zero(_fp(dst), _fp(src), W, H, _fp(np.ascontiguousarray(taps, np.float32)))
    return dst


def pyramid(planes0: np.ndarray, levels: int = 14):
    _, H, W = planes0.shape
    w, h = level_dims(W, H, levels)
    outs = [np.empty((3, h[i], w[i]), np.float32) for i in range(levels)]
    arr = (_f32p * levels)(*[_fp(o) for o in outs])
    rc = lib().orc_pyramid(_fp(np.ascontiguousarray(planes0)), W, H, levels, arr)
    if rc:
        raise RuntimeError(f"orc_pyramid rc={rc}")
    return outs


def poly(c: float, l: float, r: float, thr: float):
    d = C.c_float()
    k = C.c_float()
    lib().orc_poly(C.c_float(c), C.c_float(l), C.c_float(r), C.c_float(thr), C.byref(d), C.byref(k))
    return np.float32(d.value), np.float32(k.value)


def seed(src3: np.ndarray, W2: int, H2: int) -> np.ndarray:
    _, H, W = src3.shape
    dst = np.empty((3, H2, W2), np.float32)
    lib().orc_seed(_fp(dst), W2, H2, _fp(np.ascontiguousarray(src3)), W, H)
    return dst


def seed_fovea(src3: np.ndarray, Wup: int, Hup: int, l: int, u: int) -> np.ndarray:
    _, fh, fw = src3.shape
    dst = np.empty((3, fh, fw), np.float32)
    lib().orc_seed_fovea(_fp(dst), fw, fh, _fp(np.ascontiguousarray(src3)), Wup, Hup, l, u)
    return dst


def iterate_level(L3, R3, d3, mi, S, is_top, m_from=1, m_to=None, want_dbg=False):
    """Runs iterations m_from..m_to; returns (d3_out, dbg or None)."""
    _, H, W = L3.shape
    if m_to is None:
        m_to = mi
    d = np.ascontiguousarray(d3, np.float32).copy()
    dbg = np.empty((8, H, W), np.float32) if want_dbg else None
    lib().orc_iterate_level(_fp(np.ascontiguousarray(L3)), _fp(np.ascontiguousarray(R3)), _fp(d), W, H,
                            int(mi), int(S), int(bool(is_top)), int(m_from), int(m_to),
                            _fp(dbg) if want_dbg else None)
    return d, dbg


def smooth_pass(src3):
    _, H, W = src3.shape
    dst = np.empty_like(src3)
    lib().orc_smooth_pass(_fp(dst), _fp(np.ascontiguousarray(src3)), W, H)
    return dst


def box3(d3):
    _, H, W = d3.shape
    d = np.ascontiguousarray(d3).copy()
    lib().orc_box3(_fp(d), W, H)
    return d


def match_full(rgbL: np.ndarray, rgbR: np.ndarray, levels: int = 14) -> np.ndarray:
    H, W, _ = rgbL.shape
    assert rgbL.strides[0] == rgbR.strides[0]
    out = np.empty((3, H, W), np.float32)
    rc = lib().orc_match_full(_up(rgbL), _up(rgbR), W, H, rgbL.strides[0], levels, _fp(out))
    if rc:
        raise RuntimeError(f"orc_match_full rc={rc}")
    return out


def fovea_geometry(W, H, levels=14, F=7, off_x=0, off_y=0):
    fw, fh = C.c_int(), C.c_int()
    n = max(F - 1, 1)
    ox, oy, cx, cy = ((C.c_int * n)() for _ in range(4))
    lib().orc_fovea_geometry(W, H, levels, F, off_x, off_y, C.byref(fw), C.byref(fh), ox, oy, cx, cy)
    return fw.value, fh.value, list(ox), list(oy), list(cx), list(cy)


def match_foveated(rgbL, rgbR, levels=14, F=7, off_x=0, off_y=0, want_pyr=False):
    H, W, _ = rgbL.shape
    fw, fh, *_ = fovea_geometry(W, H, levels, F, off_x, off_y)
    stack = np.empty((3, F, fh, fw), np.float32)
    pl = np.empty((F, 3, fh, fw), np.float32) if want_pyr else None
    pr = np.empty((F, 3, fh, fw), np.float32) if want_pyr else None
    ofw, ofh = C.c_int(), C.c_int()
    rc = lib().orc_match_foveated(_up(rgbL), _up(rgbR), W, H, rgbL.strides[0], levels, F, off_x, off_y,
                                  _fp(stack), _fp(pl) if want_pyr else None, _fp(pr) if want_pyr else None,
                                  C.byref(ofw), C.byref(ofh))
    if rc:
        raise RuntimeError(f"orc_match_foveated rc={rc}")
    return stack, pl, pr


def triangulate(dispx: np.ndarray, dispy: np.ndarray, P1: np.ndarray, P2: np.ndarray) -> np.ndarray:
    """getPointCloud.cpp:886-949 (get3DPoint, non-foveated) for every pixel -> (3, H, W) X, Y, Z."""
    H, W = dispx.shape
    out = np.empty((3, H, W), np.float32)
    p1 = np.ascontiguousarray(P1, np.float64).reshape(12)
    p2 = np.ascontiguousarray(P2, np.float64).reshape(12)
    dp = C.POINTER(C.c_double)
    lib().orc_triangulate(_fp(np.ascontiguousarray(dispx)), _fp(np.ascontiguousarray(dispy)), W, H,
                          p1.ctypes.data_as(dp), p2.ctypes.data_as(dp), _fp(out))
    return out


def fovea_mapping(W: int, H: int, src_level: int, dest_level: int = 0):
    """getPointCloud.cpp:387-484 -> (left_margin, upper_margin, scale)."""
    l, u, sc = C.c_int(), C.c_int(), C.c_float()
    rc = lib().orc_fovea_mapping(W, H, src_level, dest_level, C.byref(l), C.byref(u), C.byref(sc))
    if rc:
        raise ValueError("level out of range")
    return l.value, u.value, np.float32(sc.value)


def triangulate_fovea(stackx: np.ndarray, stacky: np.ndarray, src_level: int, left: int, upper: int, scale, P1, P2) -> np.ndarray:
    """get3DPoint, foveated branch, for level src_level of (F, fovH, fovW) stacks -> (3, fovH, fovW)."""
    F, fh, fw = stackx.shape
    out = np.empty((3, fh, fw), np.float32)
    p1 = np.ascontiguousarray(P1, np.float64).reshape(12)
    p2 = np.ascontiguousarray(P2, np.float64).reshape(12)
    dp = C.POINTER(C.c_double)
    lib().orc_triangulate_fovea(_fp(np.ascontiguousarray(stackx)), _fp(np.ascontiguousarray(stacky)), fw, fh, int(src_level),
                                int(left), int(upper), C.c_float(float(scale)), p1.ctypes.data_as(dp), p2.ctypes.data_as(dp), _fp(out))
    return out


def reconstruct_full(stack3: np.ndarray, W: int, H: int, levels: int = 14, off_x: int = 0, off_y: int = 0) -> np.ndarray:
    """hierarchicalDisparity (MatchGPULib.cpp:2589-2701): (3, F, fovH, fovW) stack -> (3, H, W)."""
    _, F, fh, fw = stack3.shape
    out = np.empty((3, H, W), np.float32)
    rc = lib().orc_reconstruct_full(_fp(np.ascontiguousarray(stack3)), W, H, levels, F, off_x, off_y, _fp(out))
    if rc:
        raise RuntimeError(f"orc_reconstruct_full rc={rc}")
    return out


def weighted_difference(new3: np.ndarray, old3: np.ndarray):
    """Row f-4 (MatchGPULib.cpp:1323-1437): (difH, difV) = sum(|new - old| * conf_new) / sum(conf_new)."""
    _, H, W = new3.shape
    out = np.zeros(2, np.float32)
    lib().orc_weighted_difference(_fp(np.ascontiguousarray(new3, np.float32)), _fp(np.ascontiguousarray(old3, np.float32)), W, H, _fp(out))
    return float(out[0]), float(out[1])



def lr_check(left3: np.ndarray, right3: np.ndarray, tau: float):
    """LR-consistency check (no reference counterpart): returns (left3 with the confidence of inconsistent pixels
    zeroed, number of pixels marked)."""
    _, H, W = left3.shape
    out = np.ascontiguousarray(left3, np.float32).copy()
    f = lib().orc_lr_check
    f.restype = C.c_long
    marked = f(_fp(out), _fp(np.ascontiguousarray(right3, np.float32)), W, H, C.c_float(tau))
    return out, int(marked)
