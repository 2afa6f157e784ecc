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


@pytest.mark.parametrize("n,dims", [(2, "128x64x64"), (4, "128x64x128"), (3, "192x48x96")])
def test_fused_projection_head_on_slabs_matches_single_domain(n, dims):
    """the fused projection head (div + x·dt + residual! + first Jacobi!, wl_resjac.hip) on z-slabs: the residual of the neighbour's boundary
    plane is recomputed from two ghost planes of x and u, Σr / L₁ / L∞ are combined over the ranks before the host decides on residual!'s mean
    shift, and the Jacobi r exchange of the two-kernel path disappears.  Size gate lowered (the boxes are small); same u, p, pois.n, Δt as the
    single domain (which runs its own fused head)."""
    out = run_ranks(n, "gpu_sim", dims, "3", timeout=600, extra_env={"WL_SLAB_OPTS": "resjac_min=0"})
    for r in range(n):
        assert f"rank {r}: gpu_sim ok" in out
        assert f"rank {r}: fused projection heads on the slab: 6" in out          # two solves per step, three steps


@pytest.mark.parametrize("n,dims,repl", [(4, "64x32x144", "8"), (2, "64x64x100", "8"), (3, "64x32x120", "16")])
def test_slab_sizes_that_are_not_P_times_a_power_of_two(n, dims, repl):
    """nz = 144 on 4 ranks: 36 -> 18 -> 9 planes per rank; nz = 100 on 2: 50 -> 25; nz = 120 on 3: 40 -> 20 -> 10 -> 5.  The level that would get an odd number
    of planes per rank while still being coarsened in z is replicated instead of distributed (with the replication threshold lowered so that
    this rule, not the size threshold, decides).  Same u, p, pois.n, Δt as the single domain."""
    out = run_ranks(n, "gpu_sim", dims, "3", timeout=600, extra_env={"WL_REPLICATE_PLANES": repl})
    for r in range(n):
        assert f"rank {r}: gpu_sim ok" in out


@pytest.mark.parametrize("n,dims", [(2, "64x32x32"), (4, "48x32x64")])
def test_slab_exit_bc_with_body_matches_single_domain(n, dims):
    out = run_ranks(n, "gpu_exit", dims, "3", timeout=600)
    for r in range(n):
        assert f"rank {r}: gpu_exit ok" in out


@pytest.mark.parametrize("n,dims,per", [(2, "64x32x64", "12"), (2, "64x64x64", "123"), (2, "64x32x64", "3"), (4, "64x64x128", "123"), (3, "48x32x96", "13"),
                                        (4, "64x32x256", "3")])
def test_periodic_directions_on_slabs_match_single_domain(n, dims, per):
    """periodic TGV on z-slabs (periodic z wraps rank P-1 <-> 0): x/y-periodic copies stay local, the z-periodic boundary is a halo
    exchange that wraps around — but only where the reference calls BC!/perBC!: between two colour sweeps the single domain reads STALE
    ghost cells at the periodic boundary, so the sweep-to-sweep exchanges do not wrap.  u, p, pois.n and Δt equal the single-domain run."""
    extra = {"WL_REPLICATE_PLANES": "16"} if dims.endswith("256") else None       # (two distributed levels in the last case)
    out = run_ranks(n, "gpu_per", dims, "3", per, timeout=600, extra_env=extra)
    for r in range(n):
        assert f"rank {r}: gpu_per ok" in out


@pytest.mark.parametrize("n,dims", [(2, "64x32x32"), (4, "48x32x64")])
def test_body_in_a_z_periodic_stream_on_slabs_matches_single_domain(n, dims):
    out = run_ranks(n, "gpu_per_body", dims, "3", timeout=600)
    for r in range(n):
        assert f"rank {r}: gpu_per_body ok" in out


def test_rccl_transport_single_rank():
    out = run_ranks(1, "gpu_rccl1", timeout=600)
    assert "rank 0: gpu_rccl1 ok" in out
