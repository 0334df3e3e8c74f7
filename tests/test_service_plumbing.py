"""BASELINE.json configs[0]: 640x480 synthetic pair, 3-level pyramid, through the
GetDisparitiesGPU.srv boundary with NO GPU: the node/service plumbing is driven with a test
double that answers MatchGPULib's interface from the CPU oracle.  (The double lives here, in
tests/; the product's GPUMatcher never constructs it.)"""
import numpy as np
import pytest

from ug_stereomatcher_amd import service as svc
from ug_stereomatcher_amd import synth


class OracleMatchGPULib:
    """MatchGPULib's interface (MatchGPULib.h:6-47) answered by the CPU oracle."""

    def __init__(self, orc, levels, fovea_levels=7):
        self.orc, self.levels, self.foveatelevel = orc, levels, fovea_levels
        self.fovW = self.fovH = 0
        self.foveatedmatching = 0

    def setFoveated(self, f):
        self.foveatedmatching = f

    def getFoveaWidth(self):
        return self.fovW

    def getFoveaHeight(self):
        return self.fovH

    def getFoveateLevel(self):
        return self.foveatelevel

    def initStack(self, L, R=None):
        self.fovW, self.fovH, *_ = self.orc.fovea_geometry(L.shape[1], L.shape[0], self.levels, self.foveatelevel)
        return 0

    def match(self, L, R, fov=0):
        return self.orc.match_full(L, R, self.levels)

    def matchStack(self, L, R):
        return self.matchStackPyramid(L, R)[0]

    def matchStackPyramid(self, L, R):
        st, pl, pr = self.orc.match_foveated(L, R, self.levels, self.foveatelevel, 0, 0, want_pyr=True)
        self.fovH, self.fovW = st.shape[2], st.shape[3]
        return np.ascontiguousarray(st.transpose(1, 0, 2, 3)), pl, pr


@pytest.fixture(scope="module")
def pair():
    return synth.make_pair(640, 480, synth.BASE_SEED + 0)


def test_service_full_mode_640x480_3_levels(orc, pair):
    L, R, dx, dy = pair
    hdrL, hdrR = svc.Header(7, 1.5, "left"), svc.Header(7, 1.5, "right")
    req = svc.GetDisparitiesGPURequest(svc.Image.from_array(L.reshape(480, -1), "rgb8", hdrL),
                                      svc.Image.from_array(R.reshape(480, -1), "rgb8", hdrR))
    req.imL.width = req.imR.width = 640
    node = svc.GPUMatcher(params={}, matcher=OracleMatchGPULib(orc, 3))
    rsp = svc.GetDisparitiesGPUResponse()
    assert node.disparitySrv(req, rsp) is True
    for img, hdr in ((rsp.dispH, hdrL), (rsp.dispV, hdrR), (rsp.dispC, hdrL)):
        assert img.image.encoding == "32FC1" and (img.image.height, img.image.width) == (480, 640)
        assert img.header is hdr and img.image.step == 640 * 4  # UG_GPU_matcher.cpp:671-683
    exp = orc.match_full(L, R, 3)
    assert np.array_equal(rsp.dispH.image.to_array(), exp[0])
    assert np.array_equal(rsp.dispV.image.to_array(), exp[1])
    assert np.array_equal(rsp.dispC.image.to_array(), exp[2])
    assert rsp.fdispH.image_stack.data == b""  # untouched in full mode
    # with only 3 levels the coarse search range is tiny; conf is still a valid map
    c = rsp.dispC.image.to_array()
    assert np.isfinite(c).all() and c.min() > 0 and c.max() <= 1


def test_service_rejects_unconvertible_encoding(orc, pair):
    L, R, *_ = pair
    bad = svc.Image.from_array(np.zeros((4, 4), np.float32), "32FC1")
    node = svc.GPUMatcher(matcher=OracleMatchGPULib(orc, 3))
    assert node.disparitySrv(svc.GetDisparitiesGPURequest(bad, bad), svc.GetDisparitiesGPUResponse()) is False


def test_service_foveated_mode_and_param_reread(orc):
    L, R, _, _ = synth.make_pair(320, 240, synth.BASE_SEED + 103)
    params = {svc.FOVEATEDQ: 0}
    node = svc.GPUMatcher(params=params, matcher=OracleMatchGPULib(orc, 9, 4))
    req = svc.GetDisparitiesGPURequest(svc.Image.from_array(L.reshape(240, -1), "rgb8"), svc.Image.from_array(R.reshape(240, -1), "rgb8"))
    req.imL.width = req.imR.width = 320
    params[svc.FOVEATEDQ] = 1  # the node re-reads the parameter on every call (:522-528)
    rsp = svc.GetDisparitiesGPUResponse()
    assert node.disparitySrv(req, rsp)
    st, _, _ = orc.match_foveated(L, R, 9, 4)
    F, fh, fw = st.shape[1:]
    img = rsp.fdispH.image_stack
    assert (img.height, img.width) == (F * fh, fw)  # levels stacked vertically, finest first (:293-320)
    assert np.array_equal(img.to_array(), st[0].reshape(F * fh, fw))
    assert np.array_equal(rsp.fdispC.image_stack.to_array(), st[2].reshape(F * fh, fw))
    assert rsp.fdispH.num_levels == 0  # service path leaves the size fields unset (:590-608)


def test_topic_callback_publishes_reference_topics(orc):
    L, R, _, _ = synth.make_pair(320, 240, synth.BASE_SEED + 103)
    imL = svc.Image.from_array(L.reshape(240, -1), "rgb8", svc.Header(1, 0.0, "l"))
    imR = svc.Image.from_array(R.reshape(240, -1), "rgb8", svc.Header(1, 0.0, "r"))
    imL.width = imR.width = 320
    node = svc.GPUMatcher(params={svc.FOVEATEDQ: 1}, matcher=OracleMatchGPULib(orc, 9, 4))
    node.mainRoutine(imL, imR)
    assert set(node.published) == {"output_stackH", "output_stackV", "output_stackC", "output_stackL_pyramid", "output_stackR_pyramid"}
    sh = node.published["output_stackH"]
    assert (sh.im_width, sh.im_height, sh.num_levels) == (320, 240, 4) and sh.roi_width == sh.image_stack.width
    pl = node.published["output_stackL_pyramid"].image_stack
    assert pl.height == 4 * 3 * sh.roi_height  # (F*3*fovH) x fovW, :217
    node2 = svc.GPUMatcher(params={}, matcher=OracleMatchGPULib(orc, 9, 4))
    node2.mainRoutine(imL, imR)
    assert set(node2.published) == {"output_disparityH", "output_disparityV", "output_disparityC"}
    assert node2.published["output_disparityV"].header.frame_id == "r"  # V takes the right header (:474)
