"""B pairs per call (round 4): ugsm_submit_full_batch / ugsm_submit_foveated_batch.

The pairs of a batch march through the levels in lockstep; every level of at most ~2 Mpx is one launch for all of them (a pair index in
every kernel's grid).  Same arithmetic, so EVERY pair of EVERY batch must equal the CPU oracle's answer for that pair bit for bit --
B = 1, 2, 4, 8; full and foveated mode; different images and different fovea offsets inside one batch; every K-cost / K-smooth form
forced onto the batched levels in turn; contexts whose options make a batch run pair by pair.
"""
import numpy as np
import pytest

from conftest import assert_bit_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build_library()
    from ug_stereomatcher_amd import _lib
    return _lib


def _pairs(W, H, n, seed0):
    from ug_stereomatcher_amd import synth
    out = []
    for j in range(n):
        L, R, _, _ = synth.make_pair(W, H, synth.BASE_SEED + seed0 + 7 * j)
        if j % 3 == 1:           # a zero patch in some pairs of the batch: 0/0 -> NaN correlations on their levels only
            L = L.copy()
            L[H // 5:H // 5 + 24, W // 4:W // 4 + 40] = 0
        out.append((L, R))
    return out


def _full_batch(c, pairs, W, H, slot=0):
    n = len(pairs)
    dL = [c.to_device(L) for L, _ in pairs]
    dR = [c.to_device(R) for _, R in pairs]
    dO = [c.alloc(3 * W * H * 4) for _ in range(n)]
    try:
        c.submit_full_batch(slot, dL, dR, W, H, 3 * W, dO)
        c.check(c.lib.ugsm_wait(c.handle, slot))
        return [c.to_host(p, (3, H, W)) for p in dO]
    finally:
        for p in dL + dR + dO:
            c.free(p)


def _fovea_batch(c, lib, pairs, W, H, levels, F, offsets, want_pyr=False, slot=0):
    n = len(pairs)
    fw, fh = lib.fovea_dims(W, H, levels, F)
    dL = [c.to_device(L) for L, _ in pairs]
    dR = [c.to_device(R) for _, R in pairs]
    dS = [c.alloc(3 * F * fh * fw * 4) for _ in range(n)]
    dPL = [c.alloc(3 * F * fh * fw * 4) for _ in range(n)] if want_pyr else None
    dPR = [c.alloc(3 * F * fh * fw * 4) for _ in range(n)] if want_pyr else None
    try:
        c.submit_foveated_batch(slot, dL, dR, W, H, 3 * W, offsets, dS, dPL, dPR)
        c.check(c.lib.ugsm_wait(c.handle, slot))
        st = [c.to_host(p, (3, F, fh, fw)) for p in dS]
        pl = [c.to_host(p, (F, 3, fh, fw)) for p in dPL] if want_pyr else None
        pr = [c.to_host(p, (F, 3, fh, fw)) for p in dPR] if want_pyr else None
        return st, pl, pr
    finally:
        for p in dL + dR + dS + (dPL or []) + (dPR or []):
            c.free(p)


@pytest.mark.parametrize("B", [1, 2, 4, 8, 16])
def test_full_batch_vs_oracle(lib, orc, B):
    """Every pair of a batch against the oracle: a size whose levels are all batched, an odd one whose coarse levels are smaller than a
    tile, on a one-slot and on a several-slot context (the latency and the several-slot kernel choices)."""
    for (W, H, lv, slots) in [(640, 480, 12, 1), (333, 251, 10, 2), (584, 190, 2, 2)]:   # (two levels: no k_pyr_base)
        pairs = _pairs(W, H, B, 500 + W)
        exp = [orc.match_full(L, R, lv) for L, R in pairs]
        with lib.Context(levels=lv, slots=slots, batch=B) as c:
            got = _full_batch(c, pairs, W, H, slot=slots - 1)
            for b in range(B):
                assert_bit_equal(got[b], exp[b], f"{W}x{H}, batch of {B}, slots={slots}, pair {b}")
            # the slot is reused by a single call and by a smaller batch afterwards
            out = c.alloc(3 * W * H * 4)
            dL, dR = c.to_device(pairs[-1][0]), c.to_device(pairs[-1][1])
            c.check(c.lib.ugsm_submit_full(c.handle, slots - 1, dL, dR, W, H, 3 * W, out))
            c.check(c.lib.ugsm_wait(c.handle, slots - 1))
            assert_bit_equal(c.to_host(out, (3, H, W)), exp[-1], "single call on the slot after a batch")
            for p in (out, dL, dR):
                c.free(p)


def test_full_batch_with_levels_that_are_not_batched(lib, orc):
    """2600 x 1700: level 0 (4.4 Mpx) is above the batch threshold and runs pair by pair, the levels below it as one launch for all
    pairs; and the same batch with the threshold moved so that nothing / everything is batched."""
    import os
    W, H, lv = 2600, 1700, 14
    pairs = _pairs(W, H, 3, 900)
    exp = [orc.match_full(L, R, lv) for L, R in pairs]
    for thr in (None, "-1", "100000", "100000000"):
        if thr is not None:
            os.environ["UGSM_BATCH_MAX_PIXELS"] = thr
        try:
            with lib.Context(levels=lv, slots=2, batch=3) as c:
                got = _full_batch(c, pairs, W, H)
        finally:
            os.environ.pop("UGSM_BATCH_MAX_PIXELS", None)
        for b in range(3):
            assert_bit_equal(got[b], exp[b], f"2600x1700 batch of 3, UGSM_BATCH_MAX_PIXELS={thr}, pair {b}")


@pytest.mark.parametrize("force", ["march", "march4", "shared", "alone", "tiled", "no_fused_seed"])
def test_full_batch_every_kernel_form_on_the_batched_levels(lib, orc, monkeypatch, force):
    """The batched launches of every K-cost / K-smooth form: the marching kernel and the channel-parallel marching kernel on every
    level (their seeded first launches included), the choices of a call that shares the chip and of one that has it to itself, the
    LDS-tiled K-smooth / k_cost_split (libugsm_dev.so; it has no batch index: pair by pair inside a batch), k_seed instead of the fused seeding."""
    env = {"march": {"UGSM_MARCH_MIN_PIXELS": "1"}, "march4": {"UGSM_MARCH4": "1,2000000000"}, "shared": {"UGSM_ALONE": "0"}, "alone": {"UGSM_ALONE": "1"},
           "tiled": {"UGSM_MARCH_MIN_PIXELS": "-1", "UGSM_SMALL_MAX_PIXELS": "-1", "UGSM_MARCH4": "0,0"}, "no_fused_seed": {"UGSM_FUSE_SEED": "0"}}[force]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    W, H, lv, B = 800, 600, 12, 3
    pairs = _pairs(W, H, B, 1300)
    exp = [orc.match_full(L, R, lv) for L, R in pairs]
    with lib.Context(levels=lv, slots=2, batch=B, dev=(force == "tiled")) as c:
        got = _full_batch(c, pairs, W, H)
    for b in range(B):
        assert_bit_equal(got[b], exp[b], f"800x600 batch of {B}, {force}, pair {b}")


@pytest.mark.parametrize("B", [1, 2, 4, 8, 16])
def test_foveated_batch_with_different_offsets_vs_oracle(lib, orc, B):
    """The foveated stack of every pair of a batch, every pair with its own window offset (centred, off-centre, clamped at the frame),
    pyramid stacks included, against the oracle's answer for that pair and that offset."""
    W, H, lv, F = 1280, 960, 12, 5
    offs = ([(0, 0), (-170, 90), (5000, -5000), (33, 17), (-64, -48), (250, 0), (0, -200), (-5000, 5000)] + [(37 * j - 200, 150 - 29 * j) for j in range(8)])[:B]
    pairs = _pairs(W, H, B, 2100)
    with lib.Context(levels=lv, fovea_levels=F, slots=2, batch=B) as c:
        st, pl, pr = _fovea_batch(c, lib, pairs, W, H, lv, F, offs, want_pyr=True, slot=1)
    for b in range(B):
        est, epl, epr = orc.match_foveated(pairs[b][0], pairs[b][1], lv, F, offs[b][0], offs[b][1], want_pyr=True)
        assert_bit_equal(st[b], est, f"foveated batch of {B}, pair {b}, offset {offs[b]}: disparity stack")
        assert_bit_equal(pl[b], epl, f"foveated batch of {B}, pair {b}: left pyramid stack")
        assert_bit_equal(pr[b], epr, f"foveated batch of {B}, pair {b}: right pyramid stack")


@pytest.mark.parametrize("force", ["march", "march4", "shared", "no_fused_seed"])
def test_foveated_batch_every_kernel_form(lib, orc, monkeypatch, force):
    env = {"march": {"UGSM_MARCH_MIN_PIXELS": "1"}, "march4": {"UGSM_MARCH4": "1,2000000000"}, "shared": {"UGSM_ALONE": "0"},
           "no_fused_seed": {"UGSM_FUSE_SEED": "0"}}[force]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    W, H, lv, F, B = 1000, 700, 11, 4, 3
    offs = [(0, 0), (120, -80), (-300, 200)]
    pairs = _pairs(W, H, B, 2500)
    with lib.Context(levels=lv, fovea_levels=F, slots=1, batch=B) as c:
        st, _, _ = _fovea_batch(c, lib, pairs, W, H, lv, F, offs)
    for b in range(B):
        est, _, _ = orc.match_foveated(pairs[b][0], pairs[b][1], lv, F, offs[b][0], offs[b][1])
        assert_bit_equal(st[b], est, f"foveated batch, {force}, pair {b}, offset {offs[b]}")


def test_batches_on_several_slots_in_flight(lib, orc):
    """Two slots, each holding a batch of three pairs, in flight together, twice over (the slots' buffers are reused)."""
    W, H, lv, B = 420, 300, 10, 3
    pairs = _pairs(W, H, 2 * B, 3100)
    exp = [orc.match_full(L, R, lv) for L, R in pairs]
    with lib.Context(levels=lv, slots=2, batch=B) as c:
        dL = [c.to_device(L) for L, _ in pairs]
        dR = [c.to_device(R) for _, R in pairs]
        dO = [c.alloc(3 * W * H * 4) for _ in range(2 * B)]
        for rep in range(2):
            for s in range(2):
                sel = slice(s * B, (s + 1) * B) if rep == 0 else slice((1 - s) * B, (2 - s) * B)
                c.submit_full_batch(s, dL[sel], dR[sel], W, H, 3 * W, dO[sel])
            c.check(c.lib.ugsm_wait_all(c.handle))
            for b in range(2 * B):
                assert_bit_equal(c.to_host(dO[b], (3, H, W)), exp[b], f"two batches in flight, round {rep}, pair {b}")
        for p in dL + dR + dO:
            c.free(p)


def test_batch_on_contexts_that_run_it_pair_by_pair(lib, orc):
    """Early exit, the LR check and kernel_path 1 need a host round trip per iteration / a second match / kernels without a batch
    index: such contexts take a batch pair by pair -- same entry point, the results of the equivalent single calls."""
    W, H, lv, B = 320, 240, 8, 3
    pairs = _pairs(W, H, B, 3700)
    exp = [orc.match_full(L, R, lv) for L, R in pairs]
    with lib.Context(levels=lv, kernel_path=1, batch=B) as c:      # (libugsm_dev.so)
        got = _full_batch(c, pairs, W, H)
    for b in range(B):
        assert_bit_equal(got[b], exp[b], f"kernel_path 1, batch of {B}, pair {b}")
    with lib.Context(levels=lv, lr_check_threshold=1.0, batch=B) as c:
        got = _full_batch(c, pairs, W, H)
        single = []
        for (L, R) in pairs:
            out = np.empty((3, H, W), np.float32)
            c.check(c.lib.ugsm_match_full(c.handle, L.ctypes.data, R.ctypes.data, W, H, 3 * W, out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data))
            single.append(out)
    for b in range(B):
        assert_bit_equal(got[b], single[b], f"LR check, batch of {B}, pair {b} vs the single call")
        assert_bit_equal(got[b][:2], exp[b][:2], "the LR check leaves dx, dy alone")
    with lib.Context(levels=lv, early_exit_threshold=0.02, batch=B) as c:
        got = _full_batch(c, pairs, W, H)
        out = np.empty((3, H, W), np.float32)
        L, R = pairs[1]
        c.check(c.lib.ugsm_match_full(c.handle, L.ctypes.data, R.ctypes.data, W, H, 3 * W, out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data))
    assert_bit_equal(got[1], out, "early exit, batch vs the single call")


def test_batch_bad_arguments(lib):
    import ctypes as C
    with lib.Context(levels=5, batch=2) as c:
        p = c.alloc(64 * 48 * 3 * 4)
        ptrs = (C.c_void_p * (lib.UGSM_MAX_BATCH + 1))(*([p] * (lib.UGSM_MAX_BATCH + 1)))
        assert c.lib.ugsm_submit_full_batch(c.handle, 0, 0, ptrs, ptrs, 64, 48, 192, ptrs) == lib.UGSM_ERR_BAD_ARG
        assert c.lib.ugsm_submit_full_batch(c.handle, 0, lib.UGSM_MAX_BATCH + 1, ptrs, ptrs, 64, 48, 192, ptrs) == lib.UGSM_ERR_BAD_ARG
        assert c.lib.ugsm_submit_full_batch(c.handle, 0, 2, None, ptrs, 64, 48, 192, ptrs) == lib.UGSM_ERR_BAD_ARG
        holes = (C.c_void_p * 2)(p, None)
        assert c.lib.ugsm_submit_full_batch(c.handle, 0, 2, ptrs, holes, 64, 48, 192, ptrs) == lib.UGSM_ERR_BAD_ARG
        assert c.lib.ugsm_submit_full_batch(c.handle, 0, 2, ptrs, ptrs, 64, 48, 100, ptrs) == lib.UGSM_ERR_SIZE_MISMATCH
        assert c.lib.ugsm_submit_foveated_batch(c.handle, 0, 2, ptrs, ptrs, 64, 48, 192, None, None, None, None, None) == lib.UGSM_ERR_BAD_ARG
        c.free(p)
    cfg = lib.Config()
    lib.load().ugsm_default_config(C.byref(cfg))
    cfg.batch = lib.UGSM_MAX_BATCH + 1
    h = C.c_void_p()
    assert lib.load().ugsm_create(C.byref(cfg), C.byref(h)) == lib.UGSM_ERR_BAD_ARG


def test_16mp_batches_vs_single_calls(lib):
    """BASELINE configs[2] / configs[4] at full size: a batch of two 16 MP pairs in full mode (level 0 pair by pair, levels 1-13 -- every
    level of at most 9 Mpx -- as one launch for both) and a batch of eight foveated 16 MP pairs with eight different windows, against the
    single calls on the same context (which tests/test_gpu_parity.py pins to the oracle at this size; full-mode calls of 4 / 5 / 7 / 8 / 16
    pairs meet the oracle directly in tests/test_gpu_queue.py and test_the_largest_documented_context_at_16mp)."""
    from ug_stereomatcher_amd import synth
    W, H, F = 4928, 3264, 7
    fw, fh = lib.fovea_dims(W, H, 14, F)
    imgs = [synth.make_pair(W, H, synth.BASE_SEED + 2 + 16 * j)[:2] for j in range(2)]
    with lib.Context(levels=14, fovea_levels=F, slots=2, batch=8) as c:
        dL = [c.to_device(L) for L, _ in imgs]
        dR = [c.to_device(R) for _, R in imgs]
        dO = [c.alloc(3 * W * H * 4) for _ in range(4)]
        for j in range(2):
            c.check(c.lib.ugsm_submit_full(c.handle, j, dL[j], dR[j], W, H, 3 * W, dO[j]))
        c.check(c.lib.ugsm_wait_all(c.handle))
        c.submit_full_batch(0, dL, dR, W, H, 3 * W, dO[2:])
        c.check(c.lib.ugsm_wait(c.handle, 0))
        for j in range(2):
            assert_bit_equal(c.to_host(dO[2 + j], (3, H, W)), c.to_host(dO[j], (3, H, W)), f"16 MP full, batch of 2, pair {j} vs the single call")
        for p in dO:
            c.free(p)
        offs = [(0, 0), (900, -600), (-1500, 400), (5000, 5000), (-5000, -5000), (123, 456), (-700, -300), (2000, 0)]
        sel = [j % 2 for j in range(8)]
        dS = [c.alloc(3 * F * fh * fw * 4) for _ in range(16)]
        for j in range(8):
            c.check(c.lib.ugsm_submit_foveated(c.handle, 1, dL[sel[j]], dR[sel[j]], W, H, 3 * W, offs[j][0], offs[j][1], dS[j], None, None))
            c.check(c.lib.ugsm_wait(c.handle, 1))
        c.submit_foveated_batch(0, [dL[k] for k in sel], [dR[k] for k in sel], W, H, 3 * W, offs, dS[8:])
        c.check(c.lib.ugsm_wait(c.handle, 0))
        for j in range(8):
            assert_bit_equal(c.to_host(dS[8 + j], (3, F, fh, fw)), c.to_host(dS[j], (3, F, fh, fw)), f"16 MP foveated, batch of 8, pair {j} at {offs[j]} vs the single call")
        for p in dL + dR + dS:
            c.free(p)


def test_two_contexts_in_one_process_with_stream_priority_pools(lib):
    """ugsm_config.stream_priority: HIP deals streams onto four hardware queues PER PRIORITY LEVEL, and two
    streams on one queue run strictly one after the other -- so a second context whose slots sit in the same pool as the first one's
    shares its queues.  A host that runs two contexts gives them different pools (default: slots 0-3 at the greatest priority; 3: all at
    the least) and each then runs at the rate a context alone in the process reaches; results do not depend on any of it.  The rates are
    measured by tests/two_contexts_child.py in a FRESH process: which queue a stream lands on depends on every stream the process has
    created before (this pytest process has created hundreds)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "two_contexts_child.py")], capture_output=True, text=True, timeout=300)
    print(r.stdout[-1500:])
    assert r.returncode == 0 and "TWO_CONTEXTS_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    for bad in (4, -1):
        with pytest.raises(lib.UgsmError):
            lib.Context(levels=14, stream_priority=bad)


def test_window_only_pyramids_do_not_serve_a_later_fine_phase(lib):
    """Round 4: the one-shot foveated calls store level 0 inside their windows only, so their pyramids must not be taken for whole ones --
    a fine phase at another offset needs ugsm_submit_pyramids first (UGSM_ERR_STATE otherwise), and gets the right answer then."""
    from ug_stereomatcher_amd import synth
    W, H, lv, F = 1280, 960, 12, 5
    L, R, _, _ = synth.make_pair(W, H, synth.BASE_SEED + 91)
    fw, fh = lib.fovea_dims(W, H, lv, F)
    with lib.Context(levels=lv, fovea_levels=F) as c:
        dL, dR = c.to_device(L), c.to_device(R)
        st = [c.alloc(3 * F * fh * fw * 4) for _ in range(3)]
        state = c.alloc(3 * fh * fw * 4)
        c.check(c.lib.ugsm_submit_foveated(c.handle, 0, dL, dR, W, H, 3 * W, 0, 0, st[0], None, None))
        assert c.lib.ugsm_submit_fovea_fine(c.handle, 0, state, 200, -100, st[1]) == lib.UGSM_ERR_STATE
        assert c.lib.ugsm_submit_fovea_coarse(c.handle, 0, state) == lib.UGSM_ERR_STATE
        c.check(c.lib.ugsm_submit_pyramids(c.handle, 0, dL, dR, W, H, 3 * W))
        c.check(c.lib.ugsm_submit_fovea_coarse(c.handle, 0, state))
        c.check(c.lib.ugsm_submit_fovea_fine(c.handle, 0, state, 200, -100, st[1]))
        c.check(c.lib.ugsm_submit_foveated(c.handle, 0, dL, dR, W, H, 3 * W, 200, -100, st[2], None, None))
        c.check(c.lib.ugsm_wait(c.handle, 0))
        assert_bit_equal(c.to_host(st[1], (3, F, fh, fw)), c.to_host(st[2], (3, F, fh, fw)), "split phases on whole pyramids vs the one-shot call")
        for p in [dL, dR, state] + st:
            c.free(p)


@pytest.mark.parametrize("knobs", [{"UGSM_PYR_BASE_STREAM": "2"}, {"UGSM_PYR_BASE_STREAM": "0", "UGSM_PYR_STREAM": "0"}, {}], ids=["streaming", "tiled", "default"])
def test_both_forms_of_the_pyramid_kernels(lib, orc, monkeypatch, knobs):
    """Round 4's streaming pyramid kernels (k_pyr_base_march: foveated calls by default; k_blur_decimate2: every factor-2 level) and the
    LDS-tiled ones they stand beside, each forced onto full AND foveated calls, single and batched, at sizes that are not multiples of
    anything: pyramid stacks (the levels themselves) and disparities against the oracle."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    for (W, H, lv, F) in [(333, 251, 9, 4), (1283, 731, 12, 6)]:
        B = 3
        offs = [(0, 0), (W // 7, -H // 9), (-W, H)]
        pairs = _pairs(W, H, B, 4100 + W)
        with lib.Context(levels=lv, fovea_levels=F, slots=2, batch=B) as c:
            full = _full_batch(c, pairs, W, H)
            st, pl, pr = _fovea_batch(c, lib, pairs, W, H, lv, F, offs, want_pyr=True)
            lvl = c.alloc(3 * W * H * 4)
            dL = c.to_device(pairs[0][0])
            got_levels = []
            ws, hs = lib.level_dims(W, H, lv)
            for i in range(min(lv, 5)):
                c.check(c.lib.ugsm_stage_pyramid(c.handle, dL, W, H, 3 * W, i, lvl))
                got_levels.append(c.to_host(lvl, (3, hs[i], ws[i])))
            c.free(lvl)
            c.free(dL)
        pyr = orc.pyramid(orc.rgb_to_planes(pairs[0][0]), lv)
        for i, g in enumerate(got_levels):
            assert_bit_equal(g, pyr[i], f"{W}x{H} {knobs}: pyramid level {i}")
        for b in range(B):
            assert_bit_equal(full[b], orc.match_full(pairs[b][0], pairs[b][1], lv), f"{W}x{H} {knobs}: full, pair {b}")
            est, epl, epr = orc.match_foveated(pairs[b][0], pairs[b][1], lv, F, offs[b][0], offs[b][1], want_pyr=True)
            assert_bit_equal(st[b], est, f"{W}x{H} {knobs}: foveated stack, pair {b}")
            assert_bit_equal(pl[b], epl, f"{W}x{H} {knobs}: left pyramid stack, pair {b}")
            assert_bit_equal(pr[b], epr, f"{W}x{H} {knobs}: right pyramid stack, pair {b}")


def test_batches_from_page_locked_host_memory(lib, orc):
    """ugsm_submit_full_batch_host / ugsm_submit_foveated_batch_host: uploads, one batched match and downloads enqueued on the slot's stream;
    page-locked images and results (pageable ones are refused); every pair against the oracle; two slots in flight."""
    import ctypes as C
    W, H, lv, F, B = 640, 480, 12, 5, 3
    pairs = _pairs(W, H, 2 * B, 5200)
    offs = [(0, 0), (90, -60), (-2000, 2000)]
    fw, fh = lib.fovea_dims(W, H, lv, F)
    with lib.Context(levels=lv, fovea_levels=F, slots=2, batch=B) as c:
        hL = [c.host_array((H, W, 3), np.uint8) for _ in pairs]
        hR = [c.host_array((H, W, 3), np.uint8) for _ in pairs]
        for (L, R), a, b in zip(pairs, hL, hR):
            a[...] = L
            b[...] = R
        outs = [c.host_array((3, H, W)) for _ in pairs]
        for s in range(2):
            c.submit_full_batch_host(s, hL[s * B:(s + 1) * B], hR[s * B:(s + 1) * B], W, H, 3 * W, outs[s * B:(s + 1) * B])
        c.check(c.lib.ugsm_wait_all(c.handle))
        for b, (L, R) in enumerate(pairs):
            assert_bit_equal(outs[b], orc.match_full(L, R, lv), f"full batch from host memory, pair {b}")
        stacks = [c.host_array((3, F, fh, fw)) for _ in range(B)]
        c.submit_foveated_batch_host(1, hL[:B], hR[:B], W, H, 3 * W, offs, stacks)
        c.check(c.lib.ugsm_wait(c.handle, 1))
        for b in range(B):
            est, _, _ = orc.match_foveated(pairs[b][0], pairs[b][1], lv, F, offs[b][0], offs[b][1])
            assert_bit_equal(stacks[b], est, f"foveated batch from host memory, pair {b}, offset {offs[b]}")
        # pageable buffers are refused, not copied synchronously
        pageable = np.empty((3, H, W), np.float32)
        bad = (C.c_void_p * B)(*[pageable[0].ctypes.data] * B)
        good = lambda arrs: (C.c_void_p * B)(*[a.ctypes.data for a in arrs])
        assert c.lib.ugsm_submit_full_batch_host(c.handle, 0, B, good(hL[:B]), good(hR[:B]), W, H, 3 * W, bad, bad, bad) == lib.UGSM_ERR_BAD_ARG


def test_out_of_device_memory_is_a_status_code_and_the_context_lives_on(lib, orc, monkeypatch):
    """UGSM_ERR_NOMEM, cleanly: the development limit UGSM_MEM_LIMIT_MB makes the slots' allocator refuse to grow past it --
    the same branch a failed hipMalloc takes, without exhausting a GPU.  A call whose buffers do not fit answers UGSM_ERR_NOMEM with a
    message, holds no half-grown buffer afterwards, and the context goes on serving calls that fit, bit for bit."""
    from ug_stereomatcher_amd import synth
    W, H, lv = 640, 480, 12
    L, R, _, _ = synth.make_pair(W, H, synth.BASE_SEED + 8100)
    exp = orc.match_full(L, R, lv)
    monkeypatch.setenv("UGSM_MEM_LIMIT_MB", "60")        # one 640 x 480 pair needs ~26 MB of slot buffers, a batch of four ~100 MB
    with lib.Context(levels=lv, slots=1, batch=1) as c:
        dL, dR = c.to_device(L), c.to_device(R)
        outs = [c.alloc(3 * W * H * 4) for _ in range(4)]
        assert c.device_bytes() == 0
        c.check(c.lib.ugsm_submit_full(c.handle, 0, dL, dR, W, H, 3 * W, outs[0]))
        c.check(c.lib.ugsm_wait(c.handle, 0))
        assert_bit_equal(c.to_host(outs[0], (3, H, W)), exp, "a call that fits")
        held = c.device_bytes()
        assert 20e6 < held < 60e6, held
        import ctypes as C
        ptrs = lambda ps: (C.c_void_p * len(ps))(*ps)   # noqa: E731
        st = c.lib.ugsm_submit_full_batch(c.handle, 0, 4, ptrs([dL] * 4), ptrs([dR] * 4), W, H, 3 * W, ptrs(outs))
        assert st == lib.UGSM_ERR_NOMEM, st
        assert b"hipMalloc" in c.lib.ugsm_last_error(c.handle)
        assert c.device_bytes() <= held          # nothing half-grown is kept
        c.check(c.lib.ugsm_wait(c.handle, 0))
        c.check(c.lib.ugsm_submit_full(c.handle, 0, dL, dR, W, H, 3 * W, outs[1]))   # ... and the context still works
        c.check(c.lib.ugsm_wait(c.handle, 0))
        assert_bit_equal(c.to_host(outs[1], (3, H, W)), exp, "a call that fits, after the refused one")
        # the queue reports the failure of a call through its pairs' completions
        comp = lib.Completion()
        for k in range(2):                        # (one slot, batch 1: two pairs may be outstanding)
            assert c.lib.ugsm_enqueue_full(c.handle, dL, dR, W, H, 3 * W, outs[k], k) == lib.UGSM_OK
        c.lib.ugsm_flush(c.handle)
        sts = []
        while c.lib.ugsm_next_done(c.handle, C.byref(comp), 1) == lib.UGSM_OK:
            sts.append(comp.status)
        assert sts == [0, 0]                      # single calls, which fit
        for p in [dL, dR] + outs:
            c.free(p)
    # ... and a queue-formed call that does NOT fit: its pairs come back with UGSM_ERR_NOMEM, in order, and the queue goes on
    with lib.Context(levels=lv, slots=1, batch=4) as c:
        dL, dR = c.to_device(L), c.to_device(R)
        outs = [c.alloc(3 * W * H * 4) for _ in range(5)]
        rets = [c.lib.ugsm_enqueue_full(c.handle, dL, dR, W, H, 3 * W, outs[k], 10 + k) for k in range(4)]   # one slot: the first call takes four
        assert rets == [0, 0, 0, 0], rets   # every pair was ACCEPTED (ABI 6: a return value speaks of the pair just enqueued, never of a call it happened to send)
        got = []
        while c.lib.ugsm_next_done(c.handle, C.byref(comp), 1) == lib.UGSM_OK:
            got.append((comp.tag, comp.status, comp.call_pairs))
        assert got == [(10 + k, lib.UGSM_ERR_NOMEM, 4) for k in range(4)], got
        assert c.lib.ugsm_enqueue_full(c.handle, dL, dR, W, H, 3 * W, outs[4], 99) == lib.UGSM_OK
        d = c.next_done(True)
        assert d.tag == 99 and d.status == 0 and d.call_pairs == 1
        assert_bit_equal(c.to_host(outs[4], (3, H, W)), exp, "a one-pair call after the refused call of four")
        # the Python binding: a failed pair raises WITH its tag (it has been reported; whoever keeps per-tag state drops it first), and the
        # reference-shaped mirror forgets it -- outstanding() goes back to zero, nothing hangs
        for k in range(4):
            c.enqueue_full(dL, dR, W, H, 3 * W, outs[k], 200 + k)
        tags = []
        for k in range(4):
            with pytest.raises(lib.UgsmError) as ei:
                c.next_done(True)
            assert ei.value.status == lib.UGSM_ERR_NOMEM
            tags.append(ei.value.tag)
        assert tags == [200, 201, 202, 203] and c.next_done(True) is None
        for p in [dL, dR] + outs:
            c.free(p)


def test_the_largest_documented_context_at_16mp(lib, oracle_16mp, oracle_16mp_b):
    """INTEGRATION.md section 5's upper end: four slots, batch 16, 16 MP -- 64 pairs in flight, ~86 GB of slot buffers.  One
    call of sixteen on every slot, all four in flight; a sample of the results against the live oracle, all of them against each other."""
    W, H = oracle_16mp["W"], oracle_16mp["H"]
    imgs = [oracle_16mp, oracle_16mp_b]
    slots, B = 4, 16
    with lib.Context(levels=14, slots=slots, batch=B) as c:
        dL = [c.to_device(g["L"]) for g in imgs]
        dR = [c.to_device(g["R"]) for g in imgs]
        outs = [[c.alloc(3 * W * H * 4) for _ in range(B)] for _ in range(slots)]
        sel = [(b + s) % 2 for s in range(slots) for b in range(B)]
        for s in range(slots):
            idx = sel[s * B:(s + 1) * B]
            c.submit_full_batch(s, [dL[j] for j in idx], [dR[j] for j in idx], W, H, 3 * W, outs[s])
        c.check(c.lib.ugsm_wait_all(c.handle))
        held = c.device_bytes()
        assert 70e9 < held < 110e9, held          # slots x batch x ~1.35 GB
        ref = [None, None]
        for s in range(slots):
            for b in range(B):
                j = sel[s * B + b]
                a = c.to_host(outs[s][b], (3, H, W))
                if ref[j] is None:
                    assert_bit_equal(a, imgs[j]["full"], f"16 MP, slot {s}, pair {b} of a call of sixteen vs oracle")
                    ref[j] = a
                else:
                    assert (a.view(np.uint32) == ref[j].view(np.uint32)).all(), f"16 MP, slot {s}, pair {b} of a call of sixteen differs from its image's result"
        for p in dL + dR + [q for o in outs for q in o]:
            c.free(p)
