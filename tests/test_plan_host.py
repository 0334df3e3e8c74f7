"""Host-only checks of the per-level kernel choice (ugsm_plan_level: the same policy functions the launch path calls; no GPU needed).
Which kernel runs a level never changes a result (the GPU suite proves that); these tests pin the policy itself.  One question besides
the level's size decides (round 6): does the call have the chip to itself (`alone`), or does it share it with other calls?  The library
answers it per call from what is in flight (tests/test_gpu_queue.py::test_kernel_choices_follow_what_is_in_flight)."""
import pytest

from ug_stereomatcher_amd import _lib

TILED, MARCH, SMALL, STAGED, MARCH4 = 0, 1, 2, 3, 4


FRAME = (4928, 3264)


def levels_16mp():
    w, h = _lib.level_dims(*FRAME, 14)
    return list(zip(w, h))


def test_plan_of_a_16mp_pair_that_shares_the_chip():
    plans = [_lib.plan_level(w, h, alone=False, slots=4) for (w, h) in levels_16mp()]
    assert all(p["alone"] == 0 for p in plans)
    # levels 0-2 (>= 3 Mpx) march, levels 3-8 (50 k .. 3 Mpx) run the channel-parallel marching form, every marching level carries its own
    # seeding; levels 9-13 run the latency kernels -- on 18 x 18 smoothing tiles, the ones that redo the least
    assert [p["cost_kernel"] for p in plans] == [MARCH] * 3 + [MARCH4] * 6 + [SMALL] * 5
    assert [p["smooth_kernel"] for p in plans] == [TILED] * 9 + [SMALL] * 5
    assert [p["seed_fused"] for p in plans] == [1] * 9 + [0] * 5
    assert all(p["smooth_rh"] == 32 for p in plans[9:]) and all(p["smooth_rh"] == 0 for p in plans[:9])
    assert [p["smooth_tile_rows"] for p in plans] == [36] * 5 + [0] * 9       # the 112-column tile from 0.5 Mpx, always 36 rows high
    rows = [p["strip_rows"] for p in plans[:9]]
    assert rows[0] == 91 and all(6 <= r <= 100 for r in rows) and all(p["strip_rows"] == 0 for p in plans[9:])


def test_plan_of_a_16mp_pair_alone_on_the_chip():
    plans = [_lib.plan_level(w, h, alone=True) for (w, h) in levels_16mp()]
    assert all(p["alone"] == 1 for p in plans)
    # levels 3-6 (0.25 - 2 Mpx): a launch lasts as long as one strip, so the strip's channels go side by side (k_cost_march4); the latency
    # kernels take everything up to 0.15 Mpx
    assert [p["cost_kernel"] for p in plans] == [MARCH] * 3 + [MARCH4] * 4 + [SMALL] * 7
    assert [p["seed_fused"] for p in plans] == [1] * 7 + [0] * 7
    assert all(6 <= p["strip_rows"] <= 40 for p in plans[3:7])
    assert [p["smooth_rh"] for p in plans[7:]] == [32, 24, 18, 18, 18, 18, 18]           # the smallest tile that still fills the chip


def test_the_number_of_slots_decides_nothing():
    """Rounds 3-5 chose by ugsm_config.slots and the frame size; a lone 16 MP call on a four-slot context -- the node's service call -- lost
    11.6 % to that.  The plan of a level depends on (size, alone, batch) and on nothing else."""
    for (w, h) in levels_16mp() + [(1920, 1080), (615, 407), (300, 200)]:
        for alone in (True, False):
            ref = _lib.plan_level(w, h, alone=alone, slots=1)
            assert all(_lib.plan_level(w, h, alone=alone, slots=s) == ref for s in (2, 4, 8))
    # what `alone` changes, and only that: the reach and tile of the latency kernels, and the K-smooth tile height
    a, sh = _lib.plan_level(434, 287, alone=True), _lib.plan_level(434, 287, alone=False)        # 125 k pixels
    assert (a["cost_kernel"], sh["cost_kernel"]) == (SMALL, MARCH4)
    a, sh = _lib.plan_level(216, 142, alone=True), _lib.plan_level(216, 142, alone=False)        # 31 k pixels: small either way
    assert (a["smooth_rh"], sh["smooth_rh"]) == (18, 32) and a["cost_kernel"] == sh["cost_kernel"] == SMALL
    a, sh = _lib.plan_level(1741, 1153, alone=True), _lib.plan_level(1741, 1153, alone=False)    # 2 Mpx
    assert (a["smooth_tile_rows"], sh["smooth_tile_rows"]) == (19, 36) and a["cost_kernel"] == sh["cost_kernel"] == MARCH4


def test_switches():
    assert _lib.plan_level(4928, 3264, kernel_path=1)["cost_kernel"] == STAGED
    # round 1's LDS-tiled K-cost: libugsm_dev.so only (plan_level loads it for this configuration)
    assert _lib.plan_level(4928, 3264, march_min_pixels=-1) == dict(cost_kernel=TILED, smooth_kernel=TILED, smooth_rh=0, strip_rows=0, seed_fused=0,
                                                                     smooth_tile_rows=36, alone=1, pairs_per_launch=1)
    assert _lib.plan_level(200, 150, alone=False, small_max_pixels=-1)["cost_kernel"] == MARCH4   # no latency kernels: the marching forms take over
    assert _lib.plan_level(300, 200, small_max_pixels=-1)["cost_kernel"] == MARCH4
    assert _lib.plan_level(300, 200, march_min_pixels=1)["cost_kernel"] == MARCH
    assert _lib.plan_level(300, 200, march_min_pixels=1, march_rows=17)["strip_rows"] == 17
    assert _lib.plan_level(4928, 3264, early_exit_threshold=0.1)["seed_fused"] == 0     # the field before the first iteration is needed
    assert _lib.plan_level(4928, 3264, march_np=2) == _lib.plan_level(4928, 3264)           # ignored since ABI 3 (development form, tools/kbench)
    assert _lib.plan_level(4928, 3264, march_smooth=1) == _lib.plan_level(4928, 3264)       # ignored since ABI 6 (no longer built)
    with pytest.raises(_lib.UgsmError):
        _lib.plan_level(0, 10)


def test_strip_rows_fill_the_chip_or_one_round():
    """The strips of a marching level never need more than three waves per SIMD (3 072 strips of 58 columns), and small levels get
    short strips (a launch lasts as long as one strip)."""
    for (w, h) in levels_16mp()[:3]:
        rows = _lib.plan_level(w, h, alone=False)["strip_rows"]
        strips = -(-w // 58) * -(-h // rows)
        assert rows >= 6 and strips <= 3072 + (-(-w // 58)), (w, h, rows, strips)
    # below k_cost_march4's range the marching kernel is reached through march_min_pixels (tests) -- short strips there too
    assert _lib.plan_level(300, 200, march_min_pixels=1)["strip_rows"] <= 12


def test_smooth_tile_rows_fill_whole_rounds_for_a_call_alone_and_are_36_otherwise():
    """k_smooth_fused's 112-column tile (launches of >= 0.5 Mpx) may be 16..36 rows high; 512 workgroups are resident at a time."""
    lv = levels_16mp()
    assert [_lib.plan_level(w, h, alone=False)["smooth_tile_rows"] for (w, h) in lv] == [36] * 5 + [0] * 9
    one = [_lib.plan_level(w, h, alone=True)["smooth_tile_rows"] for (w, h) in lv]
    assert one == [36, 36, 36, 19, 18] + [0] * 9
    for (w, h), rows in list(zip(lv, one))[2:5]:            # the few-round levels: no partial round
        tiles = -(-w // 112) * -(-h // rows)
        assert tiles <= 512 * -(-tiles // 512) and tiles > 512 * (-(-tiles // 512)) - 64, (w, h, rows, tiles)
    for (w, h) in [(1920, 1080), (1000, 600), (3000, 200), (112, 5000), (5000, 113)]:
        for alone in (True, False):
            assert 16 <= _lib.plan_level(w, h, alone=alone)["smooth_tile_rows"] <= 36


def test_plan_follows_the_development_overrides_of_the_process(monkeypatch):
    """ugsm_plan_level applies the overrides ugsm_create applies: none without UGSM_DEV=1, all of them with it."""
    base = _lib.plan_level(300, 200)
    monkeypatch.setenv("UGSM_MARCH_MIN_PIXELS", "1")
    monkeypatch.delenv("UGSM_DEV", raising=False)
    assert _lib.plan_level(300, 200) == base                                    # a stray variable changes nothing in production
    monkeypatch.setenv("UGSM_DEV", "1")
    assert _lib.plan_level(300, 200)["cost_kernel"] == MARCH
    monkeypatch.delenv("UGSM_MARCH_MIN_PIXELS")
    monkeypatch.setenv("UGSM_SMALL_MASK", "1")
    p = _lib.plan_level(300, 200)
    assert p["cost_kernel"] == SMALL and p["smooth_kernel"] == TILED and p["smooth_rh"] == 0
    monkeypatch.setenv("UGSM_SMALL_MASK", "3")
    monkeypatch.setenv("UGSM_SMALL_RH", "24")
    assert _lib.plan_level(300, 200)["smooth_rh"] == 24
    monkeypatch.delenv("UGSM_SMALL_RH")
    monkeypatch.setenv("UGSM_FUSE_SEED", "0")
    assert _lib.plan_level(4928, 3264)["seed_fused"] == 0
    monkeypatch.setenv("UGSM_SMOOTH_ROWS", "24")
    assert _lib.plan_level(4928, 3264)["smooth_tile_rows"] == 24
    monkeypatch.delenv("UGSM_SMOOTH_ROWS")
    monkeypatch.setenv("UGSM_MARCH4", "1,100000")
    assert _lib.plan_level(300, 200)["cost_kernel"] == MARCH4 and _lib.plan_level(400, 300)["cost_kernel"] != MARCH4
    monkeypatch.setenv("UGSM_ALONE", "0")                                       # every call is taken to share the chip ...
    assert _lib.plan_level(216, 142, alone=True)["alone"] == 0 and _lib.plan_level(216, 142, alone=True)["smooth_rh"] == 32
    monkeypatch.setenv("UGSM_ALONE", "1")                                       # ... or to be alone, whatever is in flight
    assert _lib.plan_level(216, 142, alone=False)["alone"] == 1 and _lib.plan_level(216, 142, alone=False)["smooth_rh"] == 18
    monkeypatch.delenv("UGSM_ALONE")
    monkeypatch.setenv("UGSM_MARCH4", "0,0")
    assert all(_lib.plan_level(w, h)["cost_kernel"] != MARCH4 for (w, h) in levels_16mp())


def test_plan_of_a_batched_call():
    """ugsm_config.batch (round 4): the plan is that of a call of `batch` pairs -- levels of at most 9 Mpx are one launch for all of
    them, and every threshold is compared with what the launch holds."""
    W, H = 4928, 3264
    ws, hs = _lib.level_dims(W, H, 14)
    one = [_lib.plan_level(w, h, alone=False) for w, h in zip(ws, hs)]
    b4 = [_lib.plan_level(w, h, alone=False, batch=4) for w, h in zip(ws, hs)]
    assert [p["pairs_per_launch"] for p in one] == [1] * 14
    assert [p["pairs_per_launch"] for p in b4] == [1] + [4] * 13            # level 0 (16.1 Mpx) pair by pair, levels 1-13 batched
    assert b4[0] == dict(one[0], pairs_per_launch=1)                         # level 0: the launch of a single call
    assert one[9]["cost_kernel"] == SMALL and b4[9]["cost_kernel"] == MARCH4  # 216 x 142 = 30 672 pixels: four of them are past the 50 k threshold
    assert one[4]["cost_kernel"] == MARCH4 and b4[4]["cost_kernel"] == MARCH  # 1 Mpx: four of them are past k_cost_march4's 3 Mpx
    assert b4[13]["cost_kernel"] == SMALL and b4[13]["smooth_kernel"] == SMALL
    # the strips of a batched marching launch are sized for all its pairs' strips sharing the chip
    assert b4[1]["strip_rows"] > 3 * one[1]["strip_rows"]
    # the foveated stack, eight pairs per call: the 615 x 407 window is 2 Mpx per launch -- the channel-parallel marching K-cost and
    # the 112-column K-smooth tile, where a single window runs 64 x 32 tiles
    f1 = _lib.plan_level(615, 407, alone=False)
    f8 = _lib.plan_level(615, 407, alone=False, batch=8)
    assert f8["pairs_per_launch"] == 8 and f8["cost_kernel"] == MARCH4
    assert f1["smooth_tile_rows"] == 0 and f8["smooth_tile_rows"] == 36
    # a development override of the threshold
    import os
    os.environ["UGSM_BATCH_MAX_PIXELS"] = "-1"
    try:
        assert _lib.plan_level(615, 407, alone=False, batch=8)["pairs_per_launch"] == 1
    finally:
        del os.environ["UGSM_BATCH_MAX_PIXELS"]
