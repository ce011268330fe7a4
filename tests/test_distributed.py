"""Multi-rank tests: world_size-2 (and 3) gloo runs of tests/dist_worker.py."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _launch(nproc, *args, timeout=300, backend="gloo", side_stream=False, native_rccl=False):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", NALU_TEST_BACKEND=backend,
               NALU_TEST_SIDE_STREAM="1" if side_stream else "0",
               NALU_TEST_NATIVE_RCCL="1" if native_rccl else "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(29500 + os.getpid() % 500),
           os.path.join(HERE, "dist_worker.py"), *args]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0 and "DIST-OK" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]


@pytest.mark.parametrize("nproc", [2, 3])
def test_partition_and_halo_plan_gloo_cpu(nproc):
    _launch(nproc, "plan")


def test_block_partition_2x2x2_plan_gloo_cpu():
    """BoxBlock (2 x 2 x 2 bricks, up to 7 neighbours, indexed receive list) on 8 gloo ranks."""
    _launch(8, "plan_block")


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,method", [(2, "gmres"), (2, "cg"), (3, "gmres"), (2, "tfqmr"), (3, "biCgStab")])
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


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,backend", [(1, "nccl"), (2, "gloo")])
def test_comm_on_a_non_default_stream(nproc, backend):
    """GpuLinearSystem(stream=side stream) with torch's default stream current during the solve:
    the TorchComm callbacks switch to the library's stream, so halos and dot products are ordered
    after the kernels that produce them (same results as on the default stream)."""
    _launch(nproc, "gpu", "cg", backend=backend, side_stream=True)


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["gmres", "cg"])
def test_block_partition_2x2x2_on_one_gpu(method):
    """2 x 2 x 2 bricks (7 neighbours = the 7 xGMI links of a GPU): eight gloo ranks
    share the one GPU of the test box; indexed halo plan (the shared-not-owned nodes of a rank have
    up to 7 owners); owned rows, solution and residual history against the np = 1 oracle."""
    _launch(8, "block", method, timeout=400)


@pytest.mark.gpu
@pytest.mark.parametrize("nproc", [2, 3])
def test_export_mode_matches_owner_computes_and_np2_gold(nproc):
    """Reference-style assembly (each edge on one rank, shared-not-owned rows exported with ADD by
    loadComplete) gives the owner-computes rows to 1e-12, the np = 1 oracle rows, the same GMRES
    history, and the np = 2 gold matrix of unit_tests/UnitTestTpetra.C through the host sumInto."""
    _launch(nproc, "export")


@pytest.mark.gpu
@pytest.mark.parametrize("method,side_stream", [("cg", False), ("gmres", True)])
def test_native_rccl_callbacks_world_1(method, side_stream):
    """libnalu_gpu_rccl.so (include/nalu_gpu_rccl.h): unique id -> ncclCommInitRank -> callbacks
    attached to the system; every reduction of the solves goes through ncclAllReduce called from
    the library itself (no Python per iteration).  One rank per GPU, so world size 1 on the test
    box; the grouped send/recv half needs the 8-GPU node (bench.py --gpus N uses this path)."""
    _launch(1, "gpu", method, backend="nccl", side_stream=side_stream, native_rccl=True)


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,kind", [(2, "owner"), (3, "owner"), (2, "export"), (3, "export"), (8, "export_block")])
def test_adjacent_algorithms_and_parallel_sum_on_several_ranks(nproc, kind):
    """Hex8 geometry, projected nodal gradient, mass flow rate and velocity projection on 2 / 3 slabs and
    on 2 x 2 x 2 bricks against the np = 1 oracle on owned, shared and ghost entries: owner-computes
    (complete owned sums + halo exchange) and the reference's way (own elements / edges + the
    device-side stk::mesh::parallel_sum, nalu_gpu_parallel_sum), then the assembly that consumes them."""
    _launch(nproc, "adjacent", kind, timeout=400)


@pytest.mark.gpu
@pytest.mark.parametrize("nproc", [2, 3])
def test_composed_picard_loops_on_several_ranks(nproc):
    """three iterations of the pressure-projection loop and of the scalar transport loop with update_and_clip
    on 2 / 3 ranks, device resident, against the oracle's np = 1 composition (fields <= 1e-10)"""
    _launch(nproc, "picard", timeout=400)


@pytest.mark.gpu
def test_native_rccl_send_recv_loopback_on_a_periodic_box():
    """ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on the side stream, the
    event hand-over, the interior / boundary split of the SpMV and exchange_rows all execute on RCCL
    with ONE rank: a z-periodic box whose rank is its own neighbour.  Oracle parity (periodic operator
    on the CPU), and bit-identity with torch copies in place of RCCL."""
    _launch(1, "loopback", backend="nccl", timeout=240)


@pytest.mark.gpu
@pytest.mark.parametrize("nproc", [2, 3])
def test_an_error_on_one_rank_ends_the_solve_on_every_rank(nproc):
    """zero diagonal / zero row / NaN on one rank -> NALU_GPU_EINVAL on all ranks,
    no hang, and the ranks are still in step afterwards."""
    _launch(nproc, "baddiag", timeout=240)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["strong256", "weak512", "scalar256"])
def test_baseline_configs_3_4_5_two_ranks_on_one_gpu(kind):
    """BASELINE config 3 (256^3 Poisson split over the GPUs), the per-GPU geometry of config 5
    (512 x 512 x 64 slabs, 512^2 halo faces) and config 4 (256^3 scalar transport + GMRES(50)) at
    full size with two gloo ranks on the one GPU:
    iteration count, residual history (<= 1e-10), residual norms and matrix / rhs / solution sums
    equal the single-rank GPU run of the same box."""
    _launch(2, "config", kind, timeout=900)


@pytest.mark.gpu
def test_baseline_config_5_full_size_eight_ranks_on_one_gpu():
    """BASELINE config 5 at its own size: the 512^3 box (134 M nodes, 402 M edges) on
    EIGHT ranks -- 16.8 M nodes each, 512^2 halo faces, through bench.RankSystem -- sharing the one GPU over
    gloo, against the single-rank run of the whole 512^3 box on the same GPU: iteration count, residual
    history (<= 1e-10), residual norms, matrix / rhs / solution sums.  (Its own time guard: the host builds
    1.1 G mesh entities for the two runs.)"""
    _launch(8, "config", "full512", timeout=1100)


@pytest.mark.gpu
def test_bench_multirank_path_rehearsal():
    """bench.py --gpus 2 as the driver launches it, but with two ranks on the one GPU of the
    test box over gloo (NALU_BENCH_BACKEND): partition, halo plan, callbacks, barrier / MAX
    timing and the JSON line of the N > 1 path."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", NALU_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(29500 + (os.getpid() + 7) % 500),
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--box", "48"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert d["config"]["nodes"] == 48 * 48 * 96 and d["config"]["solver_iterations"] == 50
    assert d["value"] > 0 and d["roofline"]["frac"] > 0


@pytest.mark.gpu
def test_bench_block_partition_rehearsal():
    """bench.py --gpus 4 --partition block: the strong sub-record on 1 x 2 x 2 bricks (indexed halo plan, three
    neighbours per rank) next to the weak-scaling slabs, four gloo ranks on the one GPU."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", NALU_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4",
           "--master-addr", "127.0.0.1", "--master-port", str(29500 + (os.getpid() + 11) % 500),
           os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1", "--box", "32",
           "--partition", "block"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["config"]["parallelism"] == "z-slabs x4"
    st = d["strong"]
    assert st["partition"] == "bricks 1x2x2" and st["neighbours"] == 3 and st["nodes"] == 32 ** 3
    assert st["solver_iterations"] == 50 and st["value"] > 0
