// conv_diff! (+ BDIM! for NoBody) as a z-marching tile kernel in which every face flux is computed ONCE.
// The gather kernel of wl_flow.hip evaluates both faces of a cell in all three directions — 18 QUICK fluxes per cell,
// every one of them twice on the grid (≈1165 instructions per cell, VALU-bound at 512³).  Here a thread computes only
// the fluxes through the LOWER faces of its cell (9 per cell); the upper-face fluxes are its +x / +y neighbours' lower
// fluxes (exchanged through LDS) and, along z, the lower flux of the next plane of its own column (the kernel marches
// in z, one plane per step, and finishes plane K-1 when Φz(K) is known).  The reference's accumulation order
//     r[I,a] = ((((0 + Φx(I)) − Φx(I+δx)) + Φy(I)) − Φy(I+δy)) + Φz(I)) − Φz(I+δz)          src/Flow.jl:41-52
// is kept, so results stay bit-identical to the oracle.  3-D, non-periodic only (other cases use the gather kernel).
// Tiles: 64×TY threads (TY=4 measured best: more workgroups per CU overlap each other's load and compute phases across the
// per-plane barrier); the last lane / last row only feed their lower neighbours (tile stride 63×(TY−1)).
#include <cstdlib>

#include "wl_common.hpp"

#define CZ_X 64

namespace {
__device__ __forceinline__ float cz_med3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }
template <int SCH> __device__ __forceinline__ float cz_lam(float u, float c, float d) {
  if (SCH == WL_QUICK) return cz_med3((5 * c + 2 * d - u) / 6, c, cz_med3(10 * c - 9 * u, c, d));
  if (SCH == WL_VANLEER) return (c <= fminf(u, d) || c >= fmaxf(u, d)) ? c : c + (d - c) * (c - u) / (d - u);
  return (c + d) / 2;
}
// flux through the lower b-face of a cell: U = (u_b[P]+u_b[P−δa])/2, values of component a at P−2δb, P−δb, P, P+δb.
// lowb: the face lies on the lower wall (ϕuL), topb: on the upper wall (ϕuR)      src/Flow.jl:8,10,11
template <int SCH>
__device__ __forceinline__ float cz_flux(float U, float fm2, float fm1, float f0, float fp1, bool lowb, bool topb, float nu) {
  const bool pos = U > 0;
  float X = cz_lam<SCH>(pos ? fm2 : fp1, pos ? fm1 : f0, pos ? f0 : fm1);
  const bool use_avg = (lowb && pos) || (topb && (U < 0));
  X = use_avg ? (f0 + fm1) / 2 : X;
  return U * X - nu * (f0 - fm1);
}

struct CzBdim { const float* u0; const float* mu0; float* uout; float dt, pre, post; int scale_after; };

template <int SCH, int FUSE, int CZ_Y>
__global__ void __launch_bounds__(CZ_X * CZ_Y) k_conv_z(GridX g, float* __restrict__ r, const float* __restrict__ u, float nu, int kfirst, int klast, int zchunk, CzBdim bd) {
  constexpr int CZ_N = CZ_X * CZ_Y;
  __shared__ float sX[2][3][CZ_N + CZ_X];   // lower x-face fluxes of the newest plane (per component), +1 guard row
  __shared__ float sY[2][3][CZ_N + CZ_X];
  // ---- tile
  const int ntx = (g.nx + CZ_X - 2) / (CZ_X - 1), nty = (g.ny + CZ_Y - 2) / (CZ_Y - 1);
  const int ntiles = ntx * nty;
  const unsigned h = blockIdx.x, q = h & 7u, sq = h >> 3;
  const unsigned per = (unsigned)((ntiles + 7) >> 3);
  const int c = (int)(sq / per);
  const int tl = (int)(q * per + (sq - (unsigned)c * per));
  if (tl >= ntiles) return;
  const int tx = tl % ntx, ty = tl / ntx;
  const int lx = threadIdx.x & (CZ_X - 1), ly = threadIdx.x >> 6;
  const int i = tx * (CZ_X - 1) + lx, j = ty * (CZ_Y - 1) + ly;
  const bool indom = i < g.nx && j < g.ny;
  const bool core = indom && lx < CZ_X - 1 && ly < CZ_Y - 1;
  const int li = threadIdx.x;
  const int ks = kfirst + c * zchunk;
  const int ke = ks + zchunk < klast ? ks + zchunk : klast;   // output planes [ks,ke)
  if (ks >= ke) return;
  const int N0 = g.nx, N1 = g.ny, N2 = g.gnz;
  const int I0 = i + 1, I1 = j + 1;                              // Julia indices
  const bool okxy = indom && I0 >= 2 && I1 >= 2;
  const long oc = indom ? (long)i + (long)j * g.sy : 0;
  const float* __restrict__ U0 = u; const float* __restrict__ U1 = u + g.cs; const float* __restrict__ U2 = u + 2 * g.cs;
  // clamped in-plane offsets (a masked or boundary lane reads its own cell instead; the value is never used)
  const long xm1 = (okxy) ? -1 : 0, xm2 = (okxy && I0 >= 3) ? -2 : xm1, xp1 = (okxy && I0 <= N0 - 1) ? 1 : 0;
  const long ym1 = (okxy) ? -g.sy : 0, ym2 = (okxy && I1 >= 3) ? -2 * g.sy : ym1, yp1 = (okxy && I1 <= N1 - 1) ? g.sy : 0;
  const bool lowx = I0 == 2, topx = I0 == N0, lowy = I1 == 2, topy = I1 == N1;
  const bool conx = okxy && I0 <= N0 - 1, cony = okxy && I1 <= N1 - 1;   // (the z condition is applied per plane)
  // ---- z pipeline registers: component a at planes K-2, K-1, K, K+1 of this column
  float zm2[3], zm1[3], z0[3], zp1[3];
  auto ldz = [&](const float* __restrict__ p, int K) -> float { const int kk = K < 0 ? 0 : (K > g.nz - 1 ? g.nz - 1 : K); return indom ? p[oc + (long)kk * g.sz] : 0.f; };
  zm1[0] = ldz(U0, ks - 2); zm1[1] = ldz(U1, ks - 2); zm1[2] = ldz(U2, ks - 2);
  z0[0] = ldz(U0, ks - 1); z0[1] = ldz(U1, ks - 1); z0[2] = ldz(U2, ks - 1);
  zp1[0] = ldz(U0, ks); zp1[1] = ldz(U1, ks); zp1[2] = ldz(U2, ks);
  float accp[3] = {0.f, 0.f, 0.f};   // r of plane K-1 accumulated up to and including +Φz(K-1)
  bool conz_prev = false;            // did direction z contribute to plane K-1
  for (int K = ks; K <= ke; K++) {
    // shift the z pipeline; fetch plane K+1
#pragma unroll
    for (int a = 0; a < 3; a++) { zm2[a] = zm1[a]; zm1[a] = z0[a]; z0[a] = zp1[a]; }
    zp1[0] = ldz(U0, K + 1); zp1[1] = ldz(U1, K + 1); zp1[2] = ldz(U2, K + 1);
    const int I2 = g.gk + K + 1;
    const bool plane = K <= g.nz - 1;
    const bool ok = okxy && plane && I2 >= 2;
    const long o = oc + (long)(plane ? K : g.nz - 1) * g.sz;
    const bool lowz = I2 == 2, topz = I2 == N2;
    const bool conz = ok && I2 <= N2 - 1;
    // ---- in-plane stars of the three components (x: -2,-1,+1 ; y: -2,-1,+1)
    float sxm2[3], sxm1[3], sxp1[3], sym2[3], sym1[3], syp1[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const float* __restrict__ f = u + (long)a * g.cs;
      sxm2[a] = f[o + xm2]; sxm1[a] = f[o + xm1]; sxp1[a] = f[o + xp1];
      sym2[a] = f[o + ym2]; sym1[a] = f[o + ym1]; syp1[a] = f[o + yp1];
    }
    // component b at P−δa  (for U): a = x -> sxm1[b], a = y -> sym1[b], a = z -> zm1[b]
    float Fx[3], Fy[3], Fz[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const float nbx = (a == 0) ? sxm1[0] : (a == 1 ? sym1[0] : zm1[0]);   // u_x at P−δa
      const float nby = (a == 0) ? sxm1[1] : (a == 1 ? sym1[1] : zm1[1]);   // u_y at P−δa
      const float nbz = (a == 0) ? sxm1[2] : (a == 1 ? sym1[2] : zm1[2]);   // u_z at P−δa
      Fx[a] = cz_flux<SCH>((z0[0] + nbx) / 2, sxm2[a], sxm1[a], z0[a], sxp1[a], lowx, topx, nu);
      Fy[a] = cz_flux<SCH>((z0[1] + nby) / 2, sym2[a], sym1[a], z0[a], syp1[a], lowy, topy, nu);
      Fz[a] = cz_flux<SCH>((z0[2] + nbz) / 2, zm2[a], zm1[a], z0[a], zp1[a], lowz, topz, nu);
    }
    const int cb = K & 1;
#pragma unroll
    for (int a = 0; a < 3; a++) { sX[cb][a][li] = Fx[a]; sY[cb][a][li] = Fy[a]; }
    // ---- finish plane K-1 with −Φz(K), then BDIM!/store
    if (K > ks && core) {
      const long op = oc + (long)(K - 1) * g.sz;
      bool in = i >= 1 && i <= g.nx - 2 && j >= 1 && j <= g.ny - 2 && (K - 1) >= g.k0 && (K - 1) < g.k1;
#pragma unroll
      for (int a = 0; a < 3; a++) {
        float acc = accp[a];
        acc = conz_prev ? acc - Fz[a] : acc;
        const long oa = (long)a * g.cs + op;
        if (FUSE) {
          const float fn = bd.u0[oa] + bd.dt * acc - 0.f;                       // BDIM! :178 (V ≡ 0)
          if (r) r[oa] = fn;
          if (in) {
            const float xx = (0.f / 2 + 0.f) + bd.mu0[oa] * fn;                 // :179 (μ₁ ≡ 0, V ≡ 0)
            float un = (bd.pre == 0.f) ? xx : (u[oa] * bd.pre + xx);
            if (bd.scale_after) un = un * bd.post;
            bd.uout[oa] = un;
          }
        } else r[oa] = acc;
      }
    }
    __syncthreads();
    // ---- start plane K: x and y contributions (+ own lower flux, − neighbour's lower flux) and +Φz(K)
#pragma unroll
    for (int a = 0; a < 3; a++) {
      float acc = 0.f;
      const float fxu = sX[cb][a][li + 1], fyu = sY[cb][a][li + CZ_X];
      const bool cx = conx && ok, cy = cony && ok;
      acc = cx ? acc + Fx[a] : acc;
      acc = cx ? acc - fxu : acc;
      acc = cy ? acc + Fy[a] : acc;
      acc = cy ? acc - fyu : acc;
      acc = conz ? acc + Fz[a] : acc;
      accp[a] = acc;
    }
    conz_prev = conz;
  }
}
}  // namespace

namespace wl {
bool conv_z_ok(const GridX& g, unsigned per) { return g.D == 3 && per == 0 && g.nx >= 34 && g.ny >= 18 && g.cs < (1L << 30); }
// conv_diff!(f,u,…) [+ BDIM! NoBody when u0/mu0/u_out are given]; the Q1 ghost-plane write of Φ is done by the caller
int conv_diff_z(float* f, const float* u_adv, const float* u0, const float* mu0, float* u_out, const GridX& g, float nu, int scheme, float dt, float pre, float post, hipStream_t s) {
  static int ty_sel = -1;
  if (ty_sel < 0) { ty_sel = wl_exp_int("WL_CONVZ_TY", 4); if (ty_sel != 4 && ty_sel != 8 && ty_sel != 16) ty_sel = 4; }
  const int TY = ty_sel;
  int kfirst = (g.gk + g.k0 == 1) ? g.k0 - 1 : g.k0, klast = (g.gk + g.k1 == g.gnz - 1) ? g.k1 + 1 : g.k1;
  const int ntx = (g.nx + CZ_X - 2) / (CZ_X - 1), nty = (g.ny + TY - 2) / (TY - 1), nt = ntx * nty, per8 = (nt + 7) >> 3;
  const int np = klast - kfirst;
  int chunks = (2048 * 16 / TY / 8 + nt - 1) / nt; if (chunks < 1) chunks = 1;
  int zc = (np + chunks - 1) / chunks; if (zc < 16) zc = 16; if (zc > np) zc = np;
  const int nch = (np + zc - 1) / zc;
  const dim3 grid((unsigned)(8 * per8 * nch));
  const bool fuse = u_out != nullptr;
  if (fuse && u_out == u_adv) { wl_set_error("conv_diff_z: output aliases the advecting field"); return WL_EINVAL; }
  CzBdim bd{u0, mu0, u_out, dt, pre, post, (post != 1.f) ? 1 : 0};
#define WL_CZ2(SCHV, TYV)                                                                                                                      \
  do { if (fuse) hipLaunchKernelGGL((k_conv_z<SCHV, 1, TYV>), grid, dim3(CZ_X * TYV), 0, s, g, f, u_adv, nu, kfirst, klast, zc, bd);             \
       else hipLaunchKernelGGL((k_conv_z<SCHV, 0, TYV>), grid, dim3(CZ_X * TYV), 0, s, g, f, u_adv, nu, kfirst, klast, zc, bd); } while (0)
#define WL_CZ(SCHV) do { if (TY == 16) WL_CZ2(SCHV, 16); else if (TY == 8) WL_CZ2(SCHV, 8); else WL_CZ2(SCHV, 4); } while (0)
  switch (scheme) {
    case WL_QUICK: WL_CZ(WL_QUICK); break;
    case WL_VANLEER: WL_CZ(WL_VANLEER); break;
    case WL_CDS: WL_CZ(WL_CDS); break;
    default: wl_set_error("unknown scheme"); return WL_EINVAL;
  }
#undef WL_CZ
#undef WL_CZ2
  WL_LAUNCH_CHECK(); return 0;
}
}  // namespace wl
