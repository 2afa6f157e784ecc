"""z-slab path on the real kernels: P ranks (gloo, all on the box's one GPU) against the single-domain run."""
import pytest

from test_slab_cpu import run_ranks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,dims", [(2, "32x32x128"), (4, "32x32x256"), (2, "32x32x256"), (2, "64x64x64"), (4, "64x64x256"), (3, "128x32x96")])
def test_slab_ranks_match_single_domain(n, dims):
    out = run_ranks(n, "gpu_sim", dims, "3", timeout=600)
    for r in range(n):
        assert f"rank {r}: gpu_sim ok" in out


@pytest.mark.parametrize("n,dims", [(4, "128x64x128"), (4, "128x128x256")])
def test_small_distributed_slabs_as_on_eight_ranks(n, dims):
    """512³ on 8 GPUs has distributed levels with 64, 32 and 16 local planes; a one-GPU box allows 4 slab ranks.  With the replication
    threshold lowered (WL_REPLICATE_PLANES=16) the coarse levels stay distributed down to 16 / 8 local planes — three distributed levels,
    the pair smoother on slabs of 32·16(·8) planes — and must still match the single-domain run."""
    out = run_ranks(n, "gpu_sim", dims, "3", timeout=600, extra_env={"WL_REPLICATE_PLANES": "16"})
    for r in range(n):
        assert f"rank {r}: gpu_sim ok" in out
    kinds = [ln for ln in out.splitlines() if "smoother kinds" in ln]
    assert kinds and kinds[0].count("2") >= 2, kinds           # pair kernels on at least two distributed levels


@pytest.mark.parametrize("n,dims", [(2, "64x32x32"), (4, "48x32x64")])
def test_slab_exit_bc_with_body_matches_single_domain(n, dims):
    out = run_ranks(n, "gpu_exit", dims, "3", timeout=600)
    for r in range(n):
        assert f"rank {r}: gpu_exit ok" in out


def test_rccl_transport_single_rank():
    out = run_ranks(1, "gpu_rccl1", timeout=600)
    assert "rank 0: gpu_rccl1 ok" in out
