"""The fovea shard with its exchange inside the library, on ONE GPU: ugsm_shard_init over a one-rank RCCL communicator,
ugsm_submit_fovea_shard (ncclBroadcast on the slot's own stream), ugsm_shard_count_ranks, ugsm_shard_gather, ugsm_shard_finalize.  The work
is done by tests/rccl_shard_child.py in a fresh process: the torch.distributed group that hands the id round has to be created before
anything else initialises the GPU in that process (this pytest process has long done so), and a hang inside RCCL must not take the
session down.  No two-rank run is possible on the one-GPU pool; the protocol itself is rehearsed over gloo in tests/test_dist_gloo.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("W,H,levels,F,off,steps", [(1280, 960, 12, 5, (-170, 90), 5), (4928, 3264, 14, 7, (900, -600), 4)],
                         ids=["1280x960", "16mp"])
def test_fovea_shard_over_a_one_rank_rccl_group(W, H, levels, F, off, steps):
    import __graft_entry__ as ge
    ge.build_library()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               UGSM_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("UGSM_DIST_BACKEND", None)
    cmd = [sys.executable, os.path.join(ROOT, "tests", "rccl_shard_child.py"), str(W), str(H), str(levels), str(F), str(off[0]), str(off[1]), str(steps)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    tail = (r.stdout[-3000:] + "\n" + r.stderr[-3000:])
    assert r.returncode == 0, tail
    assert "RCCL_SHARD_OK" in r.stdout, tail


def test_fovea_shard_two_ranks_over_a_fake_transport(tmp_path):
    """The multi-rank code of csrc/ugsm_shard.cpp with TWO ranks on the one GPU of this pool ("the receiving side has never
    run"): two processes, each with its own context; tests/fake_rccl.c -- a shared-memory stand-in with RCCL's signatures, test infrastructure
    -- is what UGSM_RCCL_PATH points the library's dlopen at, because RCCL refuses two ranks on one device.  Rank 0 at the centre window, rank
    1 off centre; rank 0 and then rank 1 as the source of the coarse state; every stack equals ugsm_submit_foveated at the rank's own window
    bit for bit; ugsm_shard_count_ranks = 2; ugsm_shard_gather delivers both stacks to rank 0.  (RCCL itself -- transport, asynchrony, xGMI --
    is NOT exercised by this test; the one-rank test above runs the real library.)"""
    import __graft_entry__ as ge
    ge.build_library()
    fake = _fake_rccl(tmp_path)
    env = dict(os.environ, UGSM_DEV="1", UGSM_RCCL_PATH=fake, HSA_ENABLE_IPC_MODE_LEGACY="0")
    idfile = str(tmp_path / "shard.id")
    args = ["1280", "960", "12", "5", "5"]
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_two_ranks_child.py"), str(r), "2", idfile] + args, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=400)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o[-2000:], e[-2000:]))
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0 and f"SHARD2_OK rank={r} world=2" in o, (r, rc, o, e)


def _fake_rccl(tmp_path):
    import shutil
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    fake = str(tmp_path / "libfake_rccl.so")
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "fake_rccl.c"), "-L/opt/rocm/lib", "-lamdhip64", "-lrt",
                           "-Wl,-rpath,/opt/rocm/lib", "-o", fake])
    return fake


def test_a_failing_rank_of_the_shard_does_not_strand_its_peers(tmp_path):
    """`ugsm_submit_fovea_shard` used to return before its ncclBroadcast when the pyramids or the coarse phase were refused, and
    the other ranks -- their broadcast enqueued -- waited for ever.  Now every rank reaches the exchange, the state carries the source's status
    and the peers' ugsm_wait answers UGSM_ERR_PEER (tests/shard_failure_child.py: rank 0 under UGSM_MEM_LIMIT_MB fails as the source and as
    a receiver; good steps in between; the communicator stays usable).  Two ranks on one GPU over tests/fake_rccl.c."""
    import __graft_entry__ as ge
    ge.build_library()
    fake = _fake_rccl(tmp_path)
    idfile = str(tmp_path / "shard.id")
    procs = []
    for r in range(2):
        env = dict(os.environ, UGSM_DEV="1", UGSM_RCCL_PATH=fake, HSA_ENABLE_IPC_MODE_LEGACY="0")
        if r == 0:
            env["UGSM_MEM_LIMIT_MB"] = "700"       # a 16 MP pair needs 1.35 GB of slot buffers; a 1280 x 960 pair 100 MB
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_failure_child.py"), str(r), "2", idfile], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=400)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o[-2000:], e[-2000:]))
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0 and f"SHARD_FAIL_OK rank={r}" in o, (r, rc, o, e)


def test_rccl_path_override_is_a_development_switch(tmp_path):
    """UGSM_RCCL_PATH chooses the library the PRODUCT dlopens: like every other UGSM_* variable it is honoured under UGSM_DEV=1 only."""
    fake = _fake_rccl(tmp_path)
    code = ("import sys; sys.path.insert(0, %r); from ug_stereomatcher_amd import _lib; import ctypes as C; "
            "ident = _lib.shard_unique_id(); print('ID', bytes(ident)[:16])" % ROOT)
    env = dict(os.environ, UGSM_RCCL_PATH=fake, UGSM_NO_TORCH_RUNTIME="1")
    env.pop("UGSM_DEV", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ugsm_fake_rccl" not in r.stdout, (r.stdout, r.stderr[-1500:])      # the real RCCL made the id
    r = subprocess.run([sys.executable, "-c", code], env=dict(env, UGSM_DEV="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ugsm_fake_rccl" in r.stdout, (r.stdout, r.stderr[-1500:])
