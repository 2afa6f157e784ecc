"""N>1 path on CPU: world_size-2/3 gloo processes drive the LIBRARY's halo-exchange and plane all-gather code
(pointer arithmetic, neighbour selection, depth handling) over host buffers, plus the host-side slab planner."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_ranks(n, *args, timeout=300, extra_env=None):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + (os.getpid() % 400)), os.path.join(ROOT, "tests", "slab_worker.py"), *args]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize("n", [2, 3])
def test_halo_exchange_and_allgather_over_gloo(n):
    out = run_ranks(n, "cpu_halo")
    for r in range(n):
        assert f"rank {r}: cpu_halo ok" in out


@pytest.mark.parametrize("n", [2, 3])
def test_periodic_wrap_exchange_over_gloo(n):
    """z-periodic slabs: every rank has both neighbours, addressed modulo the size; with two ranks both are the same process"""
    out = run_ranks(n, "cpu_halo_periodic")
    for r in range(n):
        assert f"rank {r}: cpu_halo_periodic ok" in out


def test_slab_planner():
    import waterlily_jl_amd  # noqa: F401
    from waterlily_jl_amd import slab
    from waterlily_jl_amd._lib import WlError
    P = 8
    gs = [slab.slab_grid((514, 514, 514), r, P) for r in range(P)]
    assert all(g.k1 - g.k0 == 64 and g.nz == 68 and g.k0 == 2 for g in gs)
    assert [g.gk + g.k0 for g in gs] == [1 + 64 * r for r in range(P)]          # owned planes tile the interior exactly
    assert gs[-1].gk + gs[-1].k1 == 513
    g1 = slab.slab_grid((34, 34, 34), 0, 1)
    assert (g1.nz, g1.k0, g1.k1, g1.gk) == (34, 1, 33, 0)                       # one rank == the single-domain descriptor
    with pytest.raises(WlError):
        slab.slab_grid((34, 34, 35), 0, 2)                                        # 33 interior planes do not split in two
