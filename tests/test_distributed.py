"""Multi-rank tests: world_size-2 (and 3) gloo runs of tests/dist_worker.py."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _launch(nproc, *args, timeout=300, backend="gloo"):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", NALU_TEST_BACKEND=backend)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(29500 + os.getpid() % 500),
           os.path.join(HERE, "dist_worker.py"), *args]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0 and "DIST-OK" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]


@pytest.mark.parametrize("nproc", [2, 3])
def test_partition_and_halo_plan_gloo_cpu(nproc):
    _launch(nproc, "plan")


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,method", [(2, "gmres"), (2, "cg"), (3, "gmres")])
def test_multirank_assemble_and_solve_on_one_gpu(nproc, method):
    _launch(nproc, "gpu", method)


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["cg", "gmres"])
def test_comm_callbacks_over_rccl_world_1(method):
    """The bench's N > 1 path uses backend "nccl" (RCCL) on device buffers instead of the
    host-staged gloo path of the tests above.  RCCL wants one rank per GPU, so on the 1-GPU
    test box only world size 1 fits: it still drives every reduction of the solver through
    dist.all_reduce on the library's wrapped device buffer (no host staging)."""
    _launch(1, "gpu", method, backend="nccl")
