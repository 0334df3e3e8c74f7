"""Host-side mirror of the reference's `MatchGPULib` class over the C-ABI.

Same method names, argument meaning and results as
/root/reference/src/gpu_matcher/MatchGPULib.h:6-47, so call sites written against the
reference (UG_GPU_matcher.cpp:160-181,423,530-535,645) read the same.  Images are numpy
`uint8` arrays of shape (rows, cols, 3) in rgb8 order -- what `cv_bridge::toCvCopy(msg,
RGB8)->image` holds.  Differences, all deliberate:
  * results are returned as numpy arrays owned by the caller (the reference mallocs and
    expects the caller to free, UG_GPU_matcher.cpp:487-489);
  * one persistent context per object, no cudaDeviceReset per call (MatchGPULib.cpp:400);
  * a failed call raises UgsmError instead of exit(EXIT_FAILURE).
The C++ twin of this class for the ROS node is ros/MatchGPULib_ugsm.hpp.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import Context, UgsmError


def _parse_argv(argv):
    """findCudaDevice's -device=N (MatchGPULib.cpp:254) and argv[2] = fovea levels (:259-264)."""
    device, fovea_levels = 0, 7
    argv = list(argv or [])
    for a in argv:
        if isinstance(a, str) and a.startswith("-device="):
            device = int(a.split("=", 1)[1])
    if len(argv) > 2:
        try:
            fovea_levels = int(argv[2])
        except ValueError:
            fovea_levels = 0  # atoi() of a non-number
    return device, fovea_levels


def _as_rgb8(img) -> np.ndarray:
    a = np.asarray(img)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise UgsmError(_lib.UGSM_ERR_BAD_ARG, "image must be uint8 (rows, cols, 3) rgb8")
    if a.strides[2] != 1 or a.strides[1] != 3:
        a = np.ascontiguousarray(a)
    return a


class MatchGPULib:
    MAX_LEVEL = 14  # MatchLib_common.h:13

    def __init__(self, argc: int = 0, argv=None, *, levels: int = MAX_LEVEL, kernel_path: int = 0, frames_in_flight: int = 1):
        """frames_in_flight > 1 (not in the reference): the context gets min(frames_in_flight, 4) slots and the pipelined calls below
        (enqueueMatch / enqueueStack / nextDone) keep that many pairs outstanding in the library's queue; the blocking calls are unchanged."""
        device, fl = _parse_argv(argv if argc else None)
        self.foveatedmatching = 0
        self.foveatelevel = fl
        self.fovH = 0
        self.fovW = 0
        self._levels = levels
        self.frames_in_flight = max(1, int(frames_in_flight))
        slots = min(self.frames_in_flight, 4)
        batch = min(8, -(-self.frames_in_flight // slots))
        self._ctx = Context(device=device, levels=levels, fovea_levels=fl, slots=slots, kernel_path=kernel_path, batch=batch)
        self._tags = {}     # tag -> (kind, rows, cols, want_pyr) of the pairs outstanding in the queue

    # -- getters / setters, MatchGPULib.cpp:268-301 --
    def getFoveaWidth(self) -> int:
        return self.fovW

    def getFoveaHeight(self) -> int:
        return self.fovH

    def getFoveateLevel(self) -> int:
        return self.foveatelevel

    def setFoveaWidth(self, rows: int):
        self.fovW = rows

    def setFoveaHeight(self, cols: int):
        self.fovH = cols

    def setFoveated(self, fov: int):
        self.foveatedmatching = fov

    # -- initStack, MatchGPULib.cpp:406-426 --
    def initStack(self, cv_ptrL, cv_ptrR=None) -> int:
        rows, cols = np.asarray(cv_ptrL).shape[:2]
        fw, fh = _lib.fovea_dims(cols, rows, self._levels, self.foveatelevel)
        self.setFoveaHeight(fh)
        self.setFoveaWidth(fw)
        return 0

    # -- match, MatchGPULib.cpp:303-403 --
    def match(self, cv_ptrL, cv_ptrR, fov: int = 0) -> np.ndarray:
        """Returns finDisp: float32 (3, rows, cols) = horizontal, vertical disparity and confidence."""
        L, R = _as_rgb8(cv_ptrL), _as_rgb8(cv_ptrR)
        if L.shape != R.shape or L.strides[0] != R.strides[0]:
            raise UgsmError(_lib.UGSM_ERR_SIZE_MISMATCH, "left/right images differ in size")
        self.foveatedmatching = fov
        rows, cols = L.shape[:2]
        out = np.empty((3, rows, cols), np.float32)
        c = self._ctx
        if fov == 1:
            # foveated matching + hierarchicalDisparity (MatchGPULib.cpp:354-360); the node never asks for it
            # (UG_GPU_matcher.cpp:421-423,644-645), SURVEY 8f row f-3
            self.fovW, self.fovH = _lib.fovea_dims(cols, rows, self._levels, self.foveatelevel)
            c.check(c.lib.ugsm_match_foveated_full(c.handle, L.ctypes.data, R.ctypes.data, cols, rows, L.strides[0], 0, 0,
                                                   out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data))
            return out
        c.check(c.lib.ugsm_match_full(c.handle, L.ctypes.data, R.ctypes.data, cols, rows, L.strides[0],
                                      out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data))
        return out

    # -- hierarchicalDisparity, MatchGPULib.cpp:2589-2701 --
    def hierarchicalDisparity(self, foveated: np.ndarray, widthInit: int, heightInit: int) -> np.ndarray:
        """foveated: (foveatelevel, 3, fovH, fovW) as matchStack returns it -> (3, heightInit, widthInit)."""
        st = np.ascontiguousarray(np.transpose(np.asarray(foveated, np.float32), (1, 0, 2, 3)))
        c = self._ctx
        ps = [c.to_device(st[k]) for k in range(3)]
        pout = c.alloc(3 * widthInit * heightInit * 4)
        try:
            c.reconstruct_full(ps[0], ps[1], ps[2], widthInit, heightInit, pout)
            return c.to_host(pout, (3, heightInit, widthInit))
        finally:
            for p in ps + [pout]:
                c.free(p)

    def _stack(self, cv_ptrL, cv_ptrR, want_pyr: bool, off_x: int = 0, off_y: int = 0):
        L, R = _as_rgb8(cv_ptrL), _as_rgb8(cv_ptrR)
        if L.shape != R.shape or L.strides[0] != R.strides[0]:
            raise UgsmError(_lib.UGSM_ERR_SIZE_MISMATCH, "left/right images differ in size")
        rows, cols = L.shape[:2]
        F = self.foveatelevel
        fw, fh = _lib.fovea_dims(cols, rows, self._levels, F)
        self.fovW, self.fovH = fw, fh  # matching() sets them too, MatchGPULib.cpp:1233-1234
        stack = np.empty((3, F, fh, fw), np.float32)
        pl = np.empty((F, 3, fh, fw), np.float32) if want_pyr else None
        pr = np.empty((F, 3, fh, fw), np.float32) if want_pyr else None
        c = self._ctx
        c.check(c.lib.ugsm_match_foveated(c.handle, L.ctypes.data, R.ctypes.data, cols, rows, L.strides[0], off_x, off_y,
                                          stack[0].ctypes.data, stack[1].ctypes.data, stack[2].ctypes.data,
                                          pl.ctypes.data if want_pyr else None, pr.ctypes.data if want_pyr else None))
        # reference indexing is disparity[level][plane][row*fovW + col]
        return np.ascontiguousarray(stack.transpose(1, 0, 2, 3)), pl, pr

    # -- matchStack, MatchGPULib.cpp:429-531 --
    def matchStack(self, cv_ptrL, cv_ptrR, off_x: int = 0, off_y: int = 0) -> np.ndarray:
        """Returns float32 (foveatelevel, 3, fovH, fovW): [level][dx|dy|conf]."""
        return self._stack(cv_ptrL, cv_ptrR, False, off_x, off_y)[0]

    # -- matchStackPyramid, MatchGPULib.cpp:534-700 --
    def matchStackPyramid(self, cv_ptrL, cv_ptrR, off_x: int = 0, off_y: int = 0):
        """Returns (disparity stack, leftFov, rightFov); the latter two are (foveatelevel, 3, fovH, fovW)."""
        return self._stack(cv_ptrL, cv_ptrR, True, off_x, off_y)

    # -- the pipelined twin of match / matchStack / matchStackPyramid (not in the reference; include/ugsm.h "the queue") --
    def enqueueMatch(self, cv_ptrL, cv_ptrR, tag: int):
        """match(L, R, 0) without the wait: the images are copied before the call returns; the result comes out of nextDone."""
        L, R = _as_rgb8(cv_ptrL), _as_rgb8(cv_ptrR)
        if L.shape != R.shape or L.strides[0] != R.strides[0]:
            raise UgsmError(_lib.UGSM_ERR_SIZE_MISMATCH, "left/right images differ in size")
        self._ctx.enqueue_full_managed(L, R, tag)
        self._tags[tag] = ("full", L.shape[0], L.shape[1], False)
        self._ctx.flush()   # a frame that arrives starts at once if a slot is free; under load the backlog batches itself

    def enqueueStack(self, cv_ptrL, cv_ptrR, tag: int, want_pyr: bool = False, off_x: int = 0, off_y: int = 0):
        """matchStack / matchStackPyramid without the wait."""
        L, R = _as_rgb8(cv_ptrL), _as_rgb8(cv_ptrR)
        if L.shape != R.shape or L.strides[0] != R.strides[0]:
            raise UgsmError(_lib.UGSM_ERR_SIZE_MISMATCH, "left/right images differ in size")
        rows, cols = L.shape[:2]
        self.fovW, self.fovH = _lib.fovea_dims(cols, rows, self._levels, self.foveatelevel)
        self._ctx.enqueue_foveated_managed(L, R, (off_x, off_y), want_pyr, tag)
        self._tags[tag] = ("stack", rows, cols, bool(want_pyr))
        self._ctx.flush()

    def outstanding(self) -> int:
        return len(self._tags)

    def nextDone(self, block: bool = True):
        """The oldest outstanding pair: (tag, result) with result as match / matchStack / matchStackPyramid return it (copies owned by the
        caller), or None if there is none (block=False: or it has not finished).  A pair whose call failed raises UgsmError (`.tag` names it);
        it no longer counts as outstanding."""
        try:
            c = self._ctx.next_done(block)
        except UgsmError as e:
            if e.tag is not None:
                self._tags.pop(e.tag, None)
            raise
        if c is None:
            return None
        kind, rows, cols, want_pyr = self._tags.pop(c.tag)
        if kind == "full":
            h, v, cf = self._ctx.managed_planes(c, [(rows, cols)] * 3)
            return c.tag, np.stack([h, v, cf])
        F = self.foveatelevel
        fw, fh = _lib.fovea_dims(cols, rows, self._levels, F)
        shapes = [(F, fh, fw)] * 3 + ([(F, 3, fh, fw)] * 2 if want_pyr else [])
        pl = self._ctx.managed_planes(c, shapes)
        stack = np.ascontiguousarray(np.stack(pl[:3]).transpose(1, 0, 2, 3))   # [level][plane][row][col], as matchStack
        if want_pyr:
            return c.tag, (stack, pl[3].copy(), pl[4].copy())
        return c.tag, stack

    def close(self):
        self._ctx.close()
