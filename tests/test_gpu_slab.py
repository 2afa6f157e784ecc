"""z-slab path on the real kernels: P ranks (gloo, all on the box's one GPU) against the single-domain run."""
import pytest

from test_slab_cpu import run_ranks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,dims", [(2, "32x32x128"), (4, "32x32x256"), (2, "32x32x256"), (2, "64x64x64"), (4, "64x64x256"), (3, "128x32x96")])
def test_slab_ranks_match_single_domain(n, dims):
    out = run_ranks(n, "gpu_sim", dims, "3", timeout=600)
    for r in range(n):
        assert f"rank {r}: gpu_sim ok" in out


@pytest.mark.parametrize("n,dims", [(2, "64x32x32"), (4, "48x32x64")])
def test_slab_exit_bc_with_body_matches_single_domain(n, dims):
    out = run_ranks(n, "gpu_exit", dims, "3", timeout=600)
    for r in range(n):
        assert f"rank {r}: gpu_exit ok" in out


def test_rccl_transport_single_rank():
    out = run_ranks(1, "gpu_rccl1", timeout=600)
    assert "rank 0: gpu_rccl1 ok" in out
