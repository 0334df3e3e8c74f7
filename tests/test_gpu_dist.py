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
