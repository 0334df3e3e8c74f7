"""ROS-free mirror of the `UG_matcher_gpu` node and `GetDisparitiesGPU.srv`.

Plain-data twins of the ROS messages on the hot path and a `GPUMatcher` class with the
reference node's two callbacks (`disparitySrv`, `mainRoutine`), same topic / service /
parameter names (UG_GPU_matcher.cpp:48-61).  It exists so that the boundary can be
exercised and tested without ROS; the catkin node in ros/ does the same against real
message types.

Message layouts follow /root/reference: srv/GetDisparitiesGPU.srv:1-9,
msg/foveatedstack.msg:1-21; `Image` and `DisparityImage` carry the fields of
sensor_msgs/Image and stereo_msgs/DisparityImage that the node touches.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .match_gpu_lib import MatchGPULib
from ._lib import UgsmError

# names, UG_GPU_matcher.cpp:48-61 and :742
NODE_NAME = "RH_GPU_matcher"
CAM_SUB_LEFT = "input_left_image"
CAM_SUB_RIGHT = "input_right_image"
CAM_PUB_HOR = "output_disparityH"
CAM_PUB_VER = "output_disparityV"
CAM_PUB_CONF = "output_disparityC"
CAM_PUB_STACK_HOR = "output_stackH"
CAM_PUB_STACK_VER = "output_stackV"
CAM_PUB_STACK_CONF = "output_stackC"
CAM_PUB_STACK_LEFTP = "output_stackL_pyramid"
CAM_PUB_STACK_LEFTR = "output_stackR_pyramid"
DISPARITIES_SRV = "get_disparities_srv"
FOVEATEDQ = "foveated"


@dataclass
class Header:
    seq: int = 0
    stamp: float = 0.0
    frame_id: str = ""


@dataclass
class Image:  # sensor_msgs/Image
    header: Header = field(default_factory=Header)
    height: int = 0
    width: int = 0
    encoding: str = ""
    is_bigendian: int = 0
    step: int = 0
    data: bytes = b""

    @staticmethod
    def from_array(a: np.ndarray, encoding: str, header: Header | None = None) -> "Image":
        a = np.ascontiguousarray(a)
        return Image(header or Header(), a.shape[0], a.shape[1], encoding, 0, a.strides[0], a.tobytes())

    def to_array(self) -> np.ndarray:
        if self.encoding == "32FC1":
            return np.frombuffer(self.data, np.float32).reshape(self.height, self.step // 4)[:, : self.width]
        if self.encoding in ("rgb8", "bgr8"):
            return np.frombuffer(self.data, np.uint8).reshape(self.height, self.step)[:, : 3 * self.width].reshape(
                self.height, self.width, 3)
        if self.encoding == "mono8":
            return np.frombuffer(self.data, np.uint8).reshape(self.height, self.step)[:, : self.width]
        raise ValueError(f"unsupported encoding {self.encoding!r}")


@dataclass
class DisparityImage:  # stereo_msgs/DisparityImage; the node fills header + image only (:469-475)
    header: Header = field(default_factory=Header)
    image: Image = field(default_factory=Image)
    f: float = 0.0
    T: float = 0.0
    min_disparity: float = 0.0
    max_disparity: float = 0.0
    delta_d: float = 0.0


@dataclass
class FoveatedStack:  # ug_stereomatcher/foveatedstack, msg/foveatedstack.msg
    header: Header = field(default_factory=Header)
    image_stack: Image = field(default_factory=Image)
    im_width: int = 0
    im_height: int = 0
    roi_width: int = 0
    roi_height: int = 0
    num_levels: int = 0


@dataclass
class GetDisparitiesGPURequest:
    imL: Image = field(default_factory=Image)
    imR: Image = field(default_factory=Image)


@dataclass
class GetDisparitiesGPUResponse:
    dispH: DisparityImage = field(default_factory=DisparityImage)
    dispV: DisparityImage = field(default_factory=DisparityImage)
    dispC: DisparityImage = field(default_factory=DisparityImage)
    fdispH: FoveatedStack = field(default_factory=FoveatedStack)
    fdispV: FoveatedStack = field(default_factory=FoveatedStack)
    fdispC: FoveatedStack = field(default_factory=FoveatedStack)


def to_cv_copy_rgb8(msg: Image) -> np.ndarray:
    """cv_bridge::toCvCopy(msg, RGB8), UG_GPU_matcher.cpp:143-144,513-514.  Raises ValueError
    (the node's cv_bridge::Exception) for encodings it cannot convert."""
    a = msg.to_array()
    if msg.encoding == "rgb8":
        return np.ascontiguousarray(a)
    if msg.encoding == "bgr8":
        return np.ascontiguousarray(a[:, :, ::-1])
    if msg.encoding == "mono8":
        return np.ascontiguousarray(np.repeat(a[:, :, None], 3, axis=2))
    raise ValueError(f"Could not convert from '{msg.encoding}' to 'rgb8'.")


class GPUMatcher:
    """class GPU_matcher, UG_GPU_matcher.cpp:66-738, minus the ROS transport."""

    def __init__(self, argc: int = 0, argv=None, params: dict | None = None, publish=None, matcher=None, frames_in_flight: int = 1, **lib_kwargs):
        """`matcher`: an object with MatchGPULib's interface to use instead of constructing one
        (the plumbing tests inject a CPU-oracle-backed double; the product never does).
        `frames_in_flight` (opt-in, default 1 = the reference's behaviour: one blocking match() per callback, UG_GPU_matcher.cpp:423):
        with n > 1 the topic path enqueues the synchronised pair into the library's queue and publishes results AS THEY COMPLETE, in
        arrival order, a frame or n - 1 later; a callback blocks only while n pairs are outstanding.  spinOnce() publishes what has
        finished meanwhile, drain() everything (node shutdown).  The service call stays blocking: a response belongs to its request."""
        self.frames_in_flight = max(1, int(frames_in_flight))
        self._pending = {}   # tag -> (imL header, imR header, image rows, image cols) of the frames in flight
        self._next_tag = 0
        self.failed_frames = 0   # frames whose library call failed (pipelined topic path): dropped, the node lives on
        self.cmd_argc, self.cmd_argv = argc, argv
        self.params = params if params is not None else {}
        self.published = {}
        self._publish = publish or (lambda topic, msg: self.published.__setitem__(topic, msg))
        self._lib_kwargs = lib_kwargs
        self._mgpu = matcher
        self.foveated = self._read_foveated()

    def _read_foveated(self) -> int:  # :96-102,152-158,522-528 -- re-read on every callback
        return int(self.params.get(FOVEATEDQ, 0))

    def _matcher(self) -> MatchGPULib:
        # the reference constructs a MatchGPULib per callback (:160,530); the context here persists
        if self._mgpu is None:
            kw = dict(self._lib_kwargs)
            if self.frames_in_flight > 1:
                kw["frames_in_flight"] = self.frames_in_flight
            self._mgpu = MatchGPULib(self.cmd_argc, self.cmd_argv, **kw)
        return self._mgpu

    @staticmethod
    def _stack_msg(planes: np.ndarray, header: Header, L: np.ndarray, fw: int, fh: int, F: int, fill_dims: bool):
        """(F, fovH, fovW) -> (F*fovH) x fovW 32FC1, finest level first (:293-320)."""
        st = FoveatedStack(header=header, image_stack=Image.from_array(planes.reshape(F * fh, fw), "32FC1", header))
        if fill_dims:  # :333-349 (topic path only; the service path leaves them 0, :590-608)
            st.im_width, st.im_height = L.shape[1], L.shape[0]
            st.roi_width, st.roi_height, st.num_levels = fw, fh, F
        return st

    def disparitySrv(self, req: GetDisparitiesGPURequest, rsp: GetDisparitiesGPUResponse) -> bool:
        """GPU_matcher::disparitySrv, :497-694."""
        try:
            L = to_cv_copy_rgb8(req.imL)
            R = to_cv_copy_rgb8(req.imR)
        except ValueError:
            return False  # :516-520
        self.foveated = self._read_foveated()
        mgpu = self._matcher()
        mgpu.setFoveated(self.foveated)
        self.drain()  # (pipelined topic path: the slots belong to the queue while frames are in flight; publish them first)
        try:
            if self.foveated == 1:
                stack = mgpu.matchStack(L, R)  # :535
                F, fw, fh = mgpu.getFoveateLevel(), mgpu.getFoveaWidth(), mgpu.getFoveaHeight()
                # U6: the reference indexes with the wrong stride and copies fovH rows only
                # (:559-563,572-578); the full stack is filled here, exactly like the topic path.
                rsp.fdispH = self._stack_msg(stack[:, 0], req.imL.header, L, fw, fh, F, False)
                rsp.fdispV = self._stack_msg(stack[:, 1], req.imR.header, L, fw, fh, F, False)
                rsp.fdispC = self._stack_msg(stack[:, 2], req.imL.header, L, fw, fh, F, False)
            else:
                fin = mgpu.match(L, R, self.foveated)  # :645
                rsp.dispH = DisparityImage(req.imL.header, Image.from_array(fin[0], "32FC1", req.imL.header))
                rsp.dispV = DisparityImage(req.imR.header, Image.from_array(fin[1], "32FC1", req.imR.header))
                rsp.dispC = DisparityImage(req.imL.header, Image.from_array(fin[2], "32FC1", req.imL.header))
        except UgsmError:
            return False  # reference: exit(); here the service call fails and the node lives on
        return True

    def mainRoutine(self, imL: Image, imR: Image) -> None:
        """GPU_matcher::mainRoutine, :126-494 (synchronised input_left_image/input_right_image)."""
        try:
            L = to_cv_copy_rgb8(imL)
            R = to_cv_copy_rgb8(imR)
        except ValueError:
            return  # :146-150
        self.foveated = self._read_foveated()
        mgpu = self._matcher()
        mgpu.setFoveated(self.foveated)
        if self.frames_in_flight > 1:
            # pipelined topic path: enqueue, publish what has finished, block only while frames_in_flight pairs are outstanding
            tag = self._next_tag
            self._next_tag += 1
            self._pending[tag] = (imL.header, imR.header, L.shape[0], L.shape[1], self.foveated)
            if self.foveated == 1:
                mgpu.initStack(L, R)
                mgpu.enqueueStack(L, R, tag, want_pyr=True)
            else:
                mgpu.enqueueMatch(L, R, tag)
            self.spinOnce()
            while mgpu.outstanding() >= self.frames_in_flight:
                if not self._publish_next(True):
                    break   # (nothing more will come out: never spin on a queue that answers "empty")
            return
        if self.foveated == 1:
            mgpu.initStack(L, R)  # :166
            stack, lf, rf = mgpu.matchStackPyramid(L, R)  # :181
            F, fw, fh = mgpu.getFoveateLevel(), mgpu.getFoveaWidth(), mgpu.getFoveaHeight()
            # :203-266 pyramid stacks: (F*3*fovH) x fovW, rows [level][channel][row]
            for topic, pyr, hdr in ((CAM_PUB_STACK_LEFTP, lf, imL.header), (CAM_PUB_STACK_LEFTR, rf, imR.header)):
                st = FoveatedStack(header=hdr, image_stack=Image.from_array(pyr.reshape(F * 3 * fh, fw), "32FC1", hdr))
                st.im_width, st.im_height = L.shape[1], L.shape[0]
                st.roi_width, st.roi_height, st.num_levels = fw, fh, F
                self._publish(topic, st)
            self._publish(CAM_PUB_STACK_HOR, self._stack_msg(stack[:, 0], imL.header, L, fw, fh, F, True))
            self._publish(CAM_PUB_STACK_VER, self._stack_msg(stack[:, 1], imR.header, L, fw, fh, F, True))
            self._publish(CAM_PUB_STACK_CONF, self._stack_msg(stack[:, 2], imL.header, L, fw, fh, F, True))
        else:
            fin = mgpu.match(L, R, self.foveated)  # :423
            self._publish(CAM_PUB_HOR, DisparityImage(imL.header, Image.from_array(fin[0], "32FC1", imL.header)))
            self._publish(CAM_PUB_VER, DisparityImage(imR.header, Image.from_array(fin[1], "32FC1", imR.header)))
            self._publish(CAM_PUB_CONF, DisparityImage(imL.header, Image.from_array(fin[2], "32FC1", imL.header)))

    # ---- the pipelined topic path (frames_in_flight > 1) ----------------------------------------------------------------------
    def _publish_next(self, block: bool) -> bool:
        mgpu = self._matcher()
        try:
            got = mgpu.nextDone(block)
        except UgsmError as e:
            # the frame's call failed (reference: exit(); here the frame is dropped and the node lives on): it has been reported, forget it
            if e.tag is None:
                raise
            self._pending.pop(e.tag, None)
            self.failed_frames += 1
            return True
        if got is None:
            return False
        tag, res = got
        hl, hr, rows, cols, fov = self._pending.pop(tag)
        if fov == 1:
            stack, lf, rf = res
            F, fw, fh = mgpu.getFoveateLevel(), stack.shape[3], stack.shape[2]
            Lshape = np.empty((rows, cols, 0), np.uint8)   # (only the image size is read)
            for topic, pyr, hdr in ((CAM_PUB_STACK_LEFTP, lf, hl), (CAM_PUB_STACK_LEFTR, rf, hr)):
                st = FoveatedStack(header=hdr, image_stack=Image.from_array(pyr.reshape(F * 3 * fh, fw), "32FC1", hdr))
                st.im_width, st.im_height = cols, rows
                st.roi_width, st.roi_height, st.num_levels = fw, fh, F
                self._publish(topic, st)
            self._publish(CAM_PUB_STACK_HOR, self._stack_msg(stack[:, 0], hl, Lshape, fw, fh, F, True))
            self._publish(CAM_PUB_STACK_VER, self._stack_msg(stack[:, 1], hr, Lshape, fw, fh, F, True))
            self._publish(CAM_PUB_STACK_CONF, self._stack_msg(stack[:, 2], hl, Lshape, fw, fh, F, True))
        else:
            self._publish(CAM_PUB_HOR, DisparityImage(hl, Image.from_array(res[0], "32FC1", hl)))
            self._publish(CAM_PUB_VER, DisparityImage(hr, Image.from_array(res[1], "32FC1", hr)))
            self._publish(CAM_PUB_CONF, DisparityImage(hl, Image.from_array(res[2], "32FC1", hl)))
        return True

    def spinOnce(self) -> int:
        """Publishes every frame that has finished (never blocks); returns how many."""
        n = 0
        while self._mgpu is not None and self._pending and self._publish_next(False):
            n += 1
        return n

    def drain(self) -> int:
        """Publishes every frame still in flight (blocks); returns how many."""
        n = 0
        while self._mgpu is not None and self._pending and self._publish_next(True):
            n += 1
        return n
