"""RCCL code paths on the one-GPU box (-m gpu): world size 1 over the ``nccl`` backend (= RCCL on ROCm; two ranks
cannot share a card under RCCL).  Runs tools/nccl_selftest.py in a CHILD process -- ``init_process_group("nccl",
device_id=...)``, ``all_gather_into_tensor``, ``all_to_all_single``, fp64 ``all_reduce``, ``all_reduce(MAX)``,
``barrier``, the eager sharded loss step, ``GraphedShardedStep`` and ``GraphedKSplitStep`` (hipGraph replay around
the collectives) and the sharded kernel smoothing -- and checks that each equals the single-GPU result.  The
multi-rank LOGIC is covered by the gloo tests (tests/test_dist_gloo.py); this test is what makes the driver execute
the RCCL calls themselves."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_rccl_collectives_and_graphed_sharded_steps_at_world_size_1():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nccl_selftest.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "nccl selftest ok: backend=nccl" in p.stdout, p.stdout[-2000:]
