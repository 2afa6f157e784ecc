// conv_diff! (src/Flow.jl:38-62) [+ BDIM! for NoBody] as a z-MARCHING gather kernel.
// The plane kernel (k_conv_diff, wl_flow.hip) is limited by the L1/TA path, not by HBM (PMC at 512³: VALU 70 % busy, TCP stalled
// on outstanding misses 61 % of the cycles, 47 global loads per cell).  Here a thread walks a contiguous chunk of planes of ONE
// cell column and keeps the five z-neighbours u_c[k−2..k+2] of all three components in registers, so the z-star (12 of the
// 47 loads, and the ones that always miss L1) never touches memory again; x/y neighbours come from L1 as before.  No LDS, no
// barrier.  Waves whose 64 columns are all ≥2 cells away from the x/y walls, on planes ≥2 away from the z walls, run the
// branch-free INNER variant of cd_cell from the window; everything else runs the generic cd_cell from memory — the arithmetic
// per cell is the same statements either way ⇒ bit-identical to k_conv_diff.
#include "wl_conv_cell.hpp"

namespace {
__device__ __forceinline__ bool cm_cell_ij(const GridX& g, long m, int& i, int& j) {
  if (m >= g.sz) return false;
  j = (int)(m / g.nx); i = (int)(m - (long)j * g.nx);
  return true;
}
int g_convm_on = 0;   // measured 6 % SLOWER than the plane kernel at 512³ (120 VGPRs → 4 waves/SIMD): opt-in (wl_sim_set_option("convm",1))

template <int SCH, int PER, int FUSE>
__global__ void __launch_bounds__(WL_BLOCK) k_conv_march(GridX g, float* __restrict__ r, const float* __restrict__ u, float nu, unsigned per, int kfirst, int klast, int zchunk, BdimArgs bd) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  if (!cm_cell_ij(g, m, i, j)) return;
  const int ks = kfirst + pz * zchunk, ke = (ks + zchunk < klast) ? ks + zchunk : klast;
  if (ks >= ke) return;
  const int N[3] = {g.nx, g.ny, g.gnz};
  const int st[3] = {1, (int)g.sy, (int)g.sz};
  const bool deepxy = i + 1 >= 3 && i + 1 <= g.nx - 2 && j + 1 >= 3 && j + 1 <= g.ny - 2;
  const bool wave_deep = __all(deepxy) != 0;
  const bool inij = i >= 1 && i <= g.nx - 2 && j >= 1 && j <= g.ny - 2;
  // register window: W[c][q] = u_c at plane k-2+q (planes outside the array are clamped: never used by a deep plane)
  float W[3][5];
  auto ldp = [&](int c, int k) -> float { const int kk = k < 0 ? 0 : (k > g.nz - 1 ? g.nz - 1 : k); return u[(long)c * g.cs + m + (long)kk * g.sz]; };
#pragma unroll
  for (int c = 0; c < 3; c++) {
#pragma unroll
    for (int q = 0; q < 5; q++) W[c][q] = ldp(c, ks - 2 + q);
  }
  for (int k = ks; k < ke; k++) {
    const int o = (int)(m + (long)k * g.sz);
    const int I[3] = {i + 1, j + 1, g.gk + k + 1};
    float nw[3];
#pragma unroll
    for (int c = 0; c < 3; c++) nw[c] = ldp(c, k + 3);                // next plane of the window, in flight during this plane's arithmetic
    float out[3];
    const bool deepz = I[2] >= 3 && I[2] <= g.gnz - 2 && k >= 2 && k <= g.nz - 3;
    if (wave_deep && deepz) cd_cell<3, SCH, PER, int, 1, 1>(g, u, o, I, N, st, nu, per, W, out);
    else cd_cell<3, SCH, PER, int, 0, 0>(g, u, o, I, N, st, nu, per, nullptr, out);
    cd_store<3, int>(g, r, u, o, I, N, inij && k >= g.k0 && k < g.k1, out, FUSE, bd);
#pragma unroll
    for (int c = 0; c < 3; c++) { W[c][0] = W[c][1]; W[c][1] = W[c][2]; W[c][2] = W[c][3]; W[c][3] = W[c][4]; W[c][4] = nw[c]; }
  }
}
}  // namespace

namespace wl {
void conv_march_enable(int on) { g_convm_on = on; }
bool conv_march_ok(const GridX& g) { return g_convm_on && g.D == 3 && g.cs < (1L << 30) && g.nx >= 8 && g.ny >= 8; }
// conv_diff!(r,u) [bd: + BDIM! NoBody] over planes [kfirst,klast); Φ's stale ghost values (quirk Q1) are the caller's business
int conv_march(float* r, const float* u, const GridX& g, float nu, unsigned per, int scheme, int kfirst, int klast, const void* bdp, hipStream_t s) {
  const BdimArgs b0{nullptr, nullptr, nullptr, 0.f, 0.f, 1.f, 0, 0, {0.f, 0.f, 0.f}};
  const BdimArgs bd = bdp ? *(const BdimArgs*)bdp : b0;
  const int zc = wl_march_chunk(g, klast - kfirst);
  const dim3 grid = wl_plane_grid(g, wl_march_slots(klast - kfirst, zc));
#define WL_CM(SCHV, PERF, FUSEF) hipLaunchKernelGGL((k_conv_march<SCHV, PERF, FUSEF>), grid, dim3(WL_BLOCK), 0, s, g, r, u, nu, per, kfirst, klast, zc, bd)
#define WL_CM2(SCHV) do { if (bdp) { if (per) WL_CM(SCHV, 1, 1); else WL_CM(SCHV, 0, 1); } else { if (per) WL_CM(SCHV, 1, 0); else WL_CM(SCHV, 0, 0); } } while (0)
  switch (scheme) {
    case WL_QUICK: WL_CM2(WL_QUICK); break;
    case WL_VANLEER: WL_CM2(WL_VANLEER); break;
    case WL_CDS: WL_CM2(WL_CDS); break;
    default: wl_set_error("unknown scheme"); return WL_EINVAL;
  }
#undef WL_CM2
#undef WL_CM
  WL_LAUNCH_CHECK(); return 0;
}
}  // namespace wl
