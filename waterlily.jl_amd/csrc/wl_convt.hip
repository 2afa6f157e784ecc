// conv_diff! (src/Flow.jl:38-62) + BDIM! for NoBody (src/Flow.jl:176-180) as a z-MARCHING, LDS-TILED kernel with two cells
// per thread — the fast path of the fused predictor/corrector at sizes that fill the chip (wl::conv_tile_ok).
//
// Why (DESIGN.md §4, round-1 PMC): the plane kernel k_conv_diff issues 69 global loads and ≈800 VALU instructions per cell,
// every flux is evaluated twice (once per adjacent cell) and its address arithmetic alone is ≈140 instructions; it sustains
// only ≈2.2 TB/s because each wave has few HBM misses in flight.  Here
//   * a 512-thread workgroup owns a 64×16-cell tile of the x-y plane and marches a chunk of z-planes; the three velocity
//     components of planes k−1, k, k+1 (+ the plane being filled) live in an LDS ring with a 2-cell halo, so every stencil
//     operand is an LDS read at an immediate offset from one per-thread base (no address arithmetic, no L1/TA traffic);
//   * the next plane's loads (3 float2 + one halo pair per thread) are issued a whole plane ahead of their first use and
//     stay in flight across the plane's arithmetic and its one barrier — the HBM stream never waits for the VALU;
//   * a thread owns two x-adjacent cells: the x-face between them is evaluated once, the z-face flux of plane k+1 is
//     carried in registers to the next plane (flux-once in z), global accesses are 8-byte;
// 13.5 flux evaluations per cell instead of 18, ≈16 LDS reads instead of 69 global loads.
// Arithmetic per face and the accumulation order per cell are those of k_conv_diff / the reference (a outer, b inner,
// r += Φ(I) then r −= Φ(I+δ)) ⇒ bit-identical results.  Scope: D = 3, no periodic direction, BDIM! fused (NoBody), f not
// stored; everything else stays on k_conv_diff.
#include <cstdlib>
#include <type_traits>

#include "wl_bcfold.hpp"
#include "wl_conv_cell.hpp"

namespace {
#define CT_TX 32                 // threads along x (two cells each)
#define CT_TY 16                 // threads (= rows) along y
#define CT_N (CT_TX * CT_TY)     // 512 threads
#define CT_CX (2 * CT_TX)        // 64 core cells along x
#define CT_CY CT_TY              // 16 core rows
#define CT_W (CT_CX + 4)         // LDS row: core + 2 halo cells per side
#define CT_H (CT_CY + 4)
#define CT_P (CT_W * CT_H)       // floats per component-plane (1360)
#define CT_SLOT (3 * CT_P)       // floats per plane slot
#define CT_NSLOT 4               // ring: planes k-1, k, k+1 are read while k+2 is written
#define CT_HPC (CT_P / 2 - CT_CX / 2 * CT_CY)   // halo float2-pairs per component-plane (168)

struct __attribute__((aligned(4))) F2u { float x, y; };   // 8-byte global access that is only 4-byte aligned (tiles start at an odd cell)
__device__ __forceinline__ float2 ldg2(const float* __restrict__ p, unsigned o) { const F2u t = *reinterpret_cast<const F2u*>(p + o); return make_float2(t.x, t.y); }
__device__ __forceinline__ void stg2(float* __restrict__ p, unsigned o, float2 v) { F2u t; t.x = v.x; t.y = v.y; *reinterpret_cast<F2u*>(p + o) = t; }
__device__ __forceinline__ float2 lds2(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ float sel(const float2& v, int e) { return e ? v.y : v.x; }

// A pair (X, X+1) of row Y whose elements may lie outside the array: the load is moved to the nearest pair inside the array
// (so that no branch and no out-of-bounds access is needed) and `mode` says how to recover the elements that do exist:
//   0: pair loaded in place   1: loaded one element to the right (only element 1 exists: it is t.x)
//   2: loaded one element to the left (only element 0 exists: it is t.y)      3: nothing exists (any value)
// Elements that do not exist are never USED by the stencil (they are f[I±2δ] beyond a wall, overridden by the wall forms).
struct PairAddr { unsigned off; int mode; };
__device__ __forceinline__ PairAddr pair_addr(int X, int Y, int nx, int ny, unsigned sy) {
  PairAddr r;
  const int Yc = Y < 0 ? 0 : (Y > ny - 1 ? ny - 1 : Y);
  int Xc = X, mode = 0;
  if (X < 0) { Xc = X + 1; mode = 1; }
  else if (X + 1 > nx - 1) { Xc = X - 1; mode = 2; }
  if (Xc < 0 || Xc + 1 > nx - 1 || Y != Yc) { mode = 3; Xc = Xc < 0 ? 0 : (Xc + 1 > nx - 1 ? nx - 2 : Xc); }
  r.off = (unsigned)Xc + (unsigned)Yc * sy; r.mode = mode;
  return r;
}
__device__ __forceinline__ float2 pair_fix(float2 t, int mode) { return make_float2(mode == 2 ? t.y : t.x, mode == 1 ? t.x : t.y); }

// Φ at the face whose plus-side cell is F:  a = f[F−2δ], b = f[F−δ], c = f[F], d = f[F+δ];  U = advecting velocity at the face.
// wl: F is the first interior cell (Julia index 2: the reference's ϕuL form), wu: F is the upper ghost (index N: ϕuR).
// Same statements as face_flux()/cd_cell (wl_conv_cell.hpp) for the lower (a,b,c,d = f[I−2δ],f[I−δ],f[I],f[I+δ]) and the upper
// (f[I−δ],f[I],f[I+δ],f[I+2δ]) face of a cell.   src/Flow.jl:8-11,47-57
// WALLS = 0: the caller knows that neither flag can be set (tile away from the x/y walls, plane away from the z walls).
template <int SCH, int WALLS>
__device__ __forceinline__ float ct_flux(float U, float a, float b, float c, float d, bool wl, bool wu, float nu) {
#ifdef WL_CT_CHEAPFLUX   // timing experiment only (wrong results): what the kernel costs without the limiter arithmetic
  return U * (a + d) - nu * (c - b);
#endif
  const bool pos = U > 0;
  float X = lam<SCH>(pos ? a : d, pos ? b : c, pos ? c : b);
  if (WALLS) {
    const bool use_avg = (wl && pos) || (wu && (U < 0));
    X = use_avg ? (c + b) / 2 : X;
  }
  return U * X - nu * (c - b);
}

int g_convt_on = 1;
int g_convt_chunk = 0;   // 0 = automatic
long g_convt_min = 2048;  // tile-planes below which the gather kernel is at least as fast (128³ = 2048: −1 %, 192³: −5 %, tools/conv_gate.sh; tests set 0 to drive the kernel on small boxes)

// FULL: every tile lies inside the array with all its cells interior ((nx−2) % 64 == 0, (ny−2) % 16 == 0): centre loads/stores unmasked.
// U0ADV: u⁰ is the advecting field itself (predictor): its value is the plane's centre, no extra load.
template <int SCH, int FULL, int U0ADV>
__global__ void __launch_bounds__(CT_N, 4) k_conv_tile(GridX g, const float* __restrict__ u, float nu, int ka, int kb, int zchunk, BdimArgs bd) {
  __shared__ float lds[CT_NSLOT * CT_SLOT];
  const int ntx = (g.nx - 2 + CT_CX - 1) / CT_CX, nty = (g.ny - 2 + CT_CY - 1) / CT_CY;
  const int ntiles = ntx * nty;
  // XCD-aware map: hardware block h is dealt to XCD h%8; XCD q walks a contiguous range of tiles (a band of rows), chunk after chunk
  const unsigned h = blockIdx.x, q = h & 7u, s = h >> 3;
  const unsigned per = (unsigned)((ntiles + 7) >> 3);
  const int c = (int)(s / per);
  const int tl = (int)(q * per + (s - (unsigned)c * per));
  if (tl >= ntiles) return;                                  // block-uniform
  const int ks = ka + c * zchunk, ke = (ks + zchunk < kb) ? ks + zchunk : kb;
  if (ks >= ke) return;
  const int tx = tl % ntx, ty = tl / ntx;
  const int x0 = 1 + tx * CT_CX, y0 = 1 + ty * CT_CY;       // first core cell (0-based, ghosts included)
  const int tid = threadIdx.x, lx = tid & (CT_TX - 1), ly = tid >> 5;
  const int x = x0 + 2 * lx, y = y0 + ly;                    // the pair (x, x+1) of row y
  const int my = (ly + 2) * CT_W + 2 + 2 * lx;               // LDS index of cell 0 inside a component-plane (even)
  const bool in0 = FULL || (y <= g.ny - 2 && x <= g.nx - 2), in1 = FULL || (y <= g.ny - 2 && x + 1 <= g.nx - 2);   // interior cell (stored)
  const unsigned cs = (unsigned)g.cs, sz = (unsigned)g.sz, sy = (unsigned)g.sy;
  PairAddr pc;                                               // centre pair
  if (FULL) { pc.off = (unsigned)x + (unsigned)y * sy; pc.mode = 0; } else pc = pair_addr(x, y, g.nx, g.ny, sy);
  // one halo pair per thread: 168 pairs per component-plane × 3 components = 504; threads 504..511 repeat pair 503 (same value, benign)
  const int hid = tid < 3 * CT_HPC ? tid : 3 * CT_HPC - 1;
  const int hcmp = hid / CT_HPC, hh = hid - hcmp * CT_HPC;
  int R, col;
  if (hh < 4 * (CT_W / 2)) { const int rr = hh / (CT_W / 2); const int m = hh - rr * (CT_W / 2); R = rr < 2 ? rr : rr + CT_CY; col = 2 * m; }
  else { const int h2 = hh - 4 * (CT_W / 2); R = 2 + (h2 >> 1); col = (h2 & 1) ? CT_W - 2 : 0; }
  const PairAddr ph = pair_addr(x0 - 2 + col, y0 - 2 + R, g.nx, g.ny, sy);
  const unsigned hbase = (unsigned)hcmp * cs + ph.off;
  const int hl = hcmp * CT_P + R * CT_W + col;
  for (int i = tid; i < CT_NSLOT * CT_SLOT; i += CT_N) lds[i] = 0.f;
  __syncthreads();

  struct Stage { float2 c[3]; float2 h; };
  // planes outside the local array (k = −1 below the first plane, nz above the last) are never used either: clamp
  auto load_plane = [&](int kk) -> Stage {
    const unsigned ko = (unsigned)(kk < 0 ? 0 : (kk > g.nz - 1 ? g.nz - 1 : kk)) * sz;
    Stage st;
#pragma unroll
    for (int cc = 0; cc < 3; cc++) st.c[cc] = ldg2(u, (unsigned)cc * cs + ko + pc.off);
    st.h = ldg2(u, hbase + ko);
    return st;
  };
  auto write_plane = [&](int kk, const Stage& st) {
    float* sl = lds + (kk & (CT_NSLOT - 1)) * CT_SLOT;
#pragma unroll
    for (int cc = 0; cc < 3; cc++) *reinterpret_cast<float2*>(sl + cc * CT_P + my) = FULL ? st.c[cc] : pair_fix(st.c[cc], pc.mode);
    *reinterpret_cast<float2*>(sl + hl) = pair_fix(st.h, ph.mode);
  };
  Stage S;
  {
    const Stage s0 = load_plane(ks - 2), s1 = load_plane(ks - 1), s2 = load_plane(ks);
    S = load_plane(ks + 1);
    write_plane(ks - 2, s0); write_plane(ks - 1, s1); write_plane(ks, s2);
  }
  __syncthreads();

  // wall flags of the x and y faces (0-based cell index 1 = first interior cell, n−1 = upper ghost)
  const bool wlx = (x == 1);
  const bool wux0 = (x == g.nx - 1), wux1 = (x + 1 == g.nx - 1), wux2 = (x + 2 == g.nx - 1);
  const bool wly = (y == 1), wuy0 = (y == g.ny - 1), wuy1 = (y + 1 == g.ny - 1);
  float zf[3][2];     // Φ at the lower z-face of the pair, per component (carried from the previous plane)

  // Φ at the face k+1 (upper z-face of plane k = lower z-face of plane k+1) for the three components of the pair;  U = (u_z[F] + u_z[F−δa])/2 on plane k+1
  auto zfaces = [&](auto wtag, int k, const float2* Zm1, const float2* C1, const float2* Zp1, const float2* Zp2, const float* Pp, float (*Pz)[2]) {
    const int Kg = g.gk + k;
    const bool wlz = (Kg + 1 == 1), wuz = (Kg + 1 == g.gnz - 1);
    const float2 Exz = lds2(Pp + 2 * CT_P - 2);      // u_z(x−2.., y, k+1): .y = u_z(x−1)
    const float2 Eyz = lds2(Pp + 2 * CT_P - CT_W);   // u_z(x.., y−1, k+1)
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
      for (int e = 0; e < 2; e++) {
        float Uz;
        if (a == 0) Uz = (sel(Zp1[2], e) + (e ? Zp1[2].x : Exz.y)) / 2;
        else if (a == 1) Uz = (sel(Zp1[2], e) + sel(Eyz, e)) / 2;
        else Uz = (sel(Zp1[2], e) + sel(C1[2], e)) / 2;
        Pz[a][e] = ct_flux<SCH, decltype(wtag)::value>(Uz, sel(Zm1[a], e), sel(C1[a], e), sel(Zp1[a], e), sel(Zp2[a], e), wlz, wuz, nu);
      }
    }
  };
  // ---- priming iteration (plane ks−1): only the z-face fluxes of the first plane's lower faces
  {
    const int k = ks - 1;
    float2 Zp2[3], C1[3], Zm1[3], Zp1[3];
#pragma unroll
    for (int cc = 0; cc < 3; cc++) Zp2[cc] = FULL ? S.c[cc] : pair_fix(S.c[cc], pc.mode);
    write_plane(k + 2, S);
    S = load_plane(k + 3);
    const float* Pm = lds + ((k - 1) & (CT_NSLOT - 1)) * CT_SLOT + my;
    const float* P0 = lds + (k & (CT_NSLOT - 1)) * CT_SLOT + my;
    const float* Pp = lds + ((k + 1) & (CT_NSLOT - 1)) * CT_SLOT + my;
#pragma unroll
    for (int cc = 0; cc < 3; cc++) { C1[cc] = lds2(P0 + cc * CT_P); Zm1[cc] = lds2(Pm + cc * CT_P); Zp1[cc] = lds2(Pp + cc * CT_P); }
    zfaces(std::integral_constant<int, 1>{}, k, Zm1, C1, Zp1, Zp2, Pp, zf);
    __syncthreads();
  }
  const int N[3] = {g.nx, g.ny, g.gnz};
#ifdef WL_CT_NOWALLSPLIT
  const bool tile_walls = true;    // experiment: every tile runs the loop with the wall forms
#else
  const bool tile_walls = tx == 0 || tx == ntx - 1 || ty == 0 || ty == nty - 1;
#endif
  // The results of plane k are stored at the top of iteration k+1, AFTER that iteration's loads have been issued: every wait on
  // the vector-memory counter (in order, loads and stores alike) then only ever covers operations issued a whole plane earlier.
  float2 un[3];
  auto store_plane = [&](auto wtag, int kq) {
    const unsigned kqo = (unsigned)kq * sz;
    if (decltype(wtag)::value && SCH != WL_VANLEER && bd.bc_on && (x <= 1 || x + 1 >= g.nx - 2 || y <= 1 || y >= g.ny - 2)) {   // (vanLeer: no registers to spare — BC! stays a launch)
      // threads on an x/y wall: BC!(u_out, U) in x and y folded into the stores (wl_bcfold.hpp) — cells on a Dirichlet face take U, cells
      // next to a ghost column/row also fill it; the z ghost planes are completed by k_bc_zplanes after the launch
      if (in0) { const float v[3] = {un[0].x, un[1].x, un[2].x}; wl_bc_fold_store_xy(bd.uout, g, x, y, kq, v, bd.bcU); }
      if (in1) { const float v[3] = {un[0].y, un[1].y, un[2].y}; wl_bc_fold_store_xy(bd.uout, g, x + 1, y, kq, v, bd.bcU); }
      return;
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const unsigned oa = (unsigned)a * cs + kqo + pc.off;
      if (FULL) stg2(bd.uout, oa, un[a]);
      else if (pc.mode == 0 && in0 && in1) stg2(bd.uout, oa, un[a]);
      else { const unsigned o0 = (unsigned)a * cs + kqo + (unsigned)x + (unsigned)y * sy; if (in0) bd.uout[o0] = un[a].x; if (in1) bd.uout[o0 + 1] = un[a].y; }
    }
  };
  // The main loop exists twice: WALLS = 1 for tiles that touch an x or y wall (the reference's ϕuL/ϕuR forms are live on some of
  // their faces), WALLS = 0 for the others (≈70 % of the tiles at 512²) without the wall forms.  Two separate loops, not a branch
  // inside one loop: each is straight-line code for the register allocator.  The z faces keep their run-time wall flags.
  auto mainloop = [&](auto wtag) {
    constexpr int WALLS = decltype(wtag)::value;
    for (int k = ks; k < ke; k++) {
      // ---- stage: plane k+2 (loaded during the previous iteration) → LDS; its centres are this plane's f[I+2δz]; issue plane k+3
      float2 Zp2[3];
#pragma unroll
      for (int cc = 0; cc < 3; cc++) Zp2[cc] = FULL ? S.c[cc] : pair_fix(S.c[cc], pc.mode);
      write_plane(k + 2, S);
      S = load_plane(k + 3);
      const unsigned ko = (unsigned)k * sz;
      float2 u0v[3];
      if (!U0ADV) {
#pragma unroll
        for (int a = 0; a < 3; a++) { u0v[a] = ldg2(bd.u0, (unsigned)a * cs + ko + pc.off); if (!FULL) u0v[a] = pair_fix(u0v[a], pc.mode); }
      }
      if (k > ks) store_plane(wtag, k - 1);
      const float* Pm = lds + ((k - 1) & (CT_NSLOT - 1)) * CT_SLOT + my;
      const float* P0 = lds + (k & (CT_NSLOT - 1)) * CT_SLOT + my;
      const float* Pp = lds + ((k + 1) & (CT_NSLOT - 1)) * CT_SLOT + my;
      float2 C1[3], Zm1[3], Zp1[3];
#pragma unroll
      for (int cc = 0; cc < 3; cc++) { C1[cc] = lds2(P0 + cc * CT_P); Zm1[cc] = lds2(Pm + cc * CT_P); Zp1[cc] = lds2(Pp + cc * CT_P); }
      float acc[3][2];
      {
        float2 CA[3], CC[3], Ym2[3], Ym1[3], Yp1[3], Yp2[3];
#pragma unroll
        for (int cc = 0; cc < 3; cc++) {
          CA[cc] = lds2(P0 + cc * CT_P - 2); CC[cc] = lds2(P0 + cc * CT_P + 2);
          Ym2[cc] = lds2(P0 + cc * CT_P - 2 * CT_W); Ym1[cc] = lds2(P0 + cc * CT_P - CT_W);
          Yp1[cc] = lds2(P0 + cc * CT_P + CT_W); Yp2[cc] = lds2(P0 + cc * CT_P + 2 * CT_W);
        }
        const float Exy = lds2(P0 + 1 * CT_P + CT_W - 2).y;      // u_y(x−1, y+1, k)
        const float Eyx = lds2(P0 + 0 * CT_P - CT_W + 2).x;      // u_x(x+2, y−1, k)
        const float Ezx = lds2(Pm + 0 * CT_P + 2).x;             // u_x(x+2, y, k−1)
        const float2 Ezy = lds2(Pm + 1 * CT_P + CT_W);           // u_y(x.., y+1, k−1)
        // rows of u_x and u_y along x: index j ↔ cell x−2+j
        const float rx[6] = {CA[0].x, CA[0].y, C1[0].x, C1[0].y, CC[0].x, CC[0].y};
        const float ry[6] = {CA[1].x, CA[1].y, C1[1].x, C1[1].y, CC[1].x, CC[1].y};
#pragma unroll
        for (int a = 0; a < 3; a++) {
          const float r[6] = {CA[a].x, CA[a].y, C1[a].x, C1[a].y, CC[a].x, CC[a].y};
          // ---- b = x: faces x, x+1, x+2;  U = (u_x[F] + u_x[F−δa])/2                                    src/Flow.jl:3,47
          float Ux[3];
          if (a == 0) { Ux[0] = (rx[2] + rx[1]) / 2; Ux[1] = (rx[3] + rx[2]) / 2; Ux[2] = (rx[4] + rx[3]) / 2; }
          else if (a == 1) { Ux[0] = (rx[2] + Ym1[0].x) / 2; Ux[1] = (rx[3] + Ym1[0].y) / 2; Ux[2] = (rx[4] + Eyx) / 2; }
          else { Ux[0] = (rx[2] + Zm1[0].x) / 2; Ux[1] = (rx[3] + Zm1[0].y) / 2; Ux[2] = (rx[4] + Ezx) / 2; }
          const float Px0 = ct_flux<SCH, WALLS>(Ux[0], r[0], r[1], r[2], r[3], wlx, wux0, nu);
          const float Px1 = ct_flux<SCH, WALLS>(Ux[1], r[1], r[2], r[3], r[4], false, wux1, nu);
          const float Px2 = ct_flux<SCH, WALLS>(Ux[2], r[2], r[3], r[4], r[5], false, wux2, nu);
          float a0 = 0.f, a1 = 0.f;
          a0 = a0 + Px0; a0 = a0 - Px1;
          a1 = a1 + Px1; a1 = a1 - Px2;
          // ---- b = y: lower face (row y) and upper face (row y+1) of each cell;  U = (u_y[F] + u_y[F−δa])/2
#pragma unroll
          for (int e = 0; e < 2; e++) {
            const float vm2 = sel(Ym2[a], e), vm1 = sel(Ym1[a], e), v0 = sel(C1[a], e), vp1 = sel(Yp1[a], e), vp2 = sel(Yp2[a], e);
            float Ul, Uu;
            if (a == 0) { Ul = (ry[2 + e] + ry[1 + e]) / 2; Uu = (sel(Yp1[1], e) + (e ? Yp1[1].x : Exy)) / 2; }
            else if (a == 1) { Ul = (v0 + vm1) / 2; Uu = (vp1 + v0) / 2; }
            else { Ul = (sel(C1[1], e) + sel(Zm1[1], e)) / 2; Uu = (sel(Yp1[1], e) + sel(Ezy, e)) / 2; }
            const float Pl = ct_flux<SCH, WALLS>(Ul, vm2, vm1, v0, vp1, wly, wuy0, nu);
            const float Pu = ct_flux<SCH, WALLS>(Uu, vm1, v0, vp1, vp2, false, wuy1, nu);
            if (e == 0) { a0 = a0 + Pl; a0 = a0 - Pu; } else { a1 = a1 + Pl; a1 = a1 - Pu; }
          }
          acc[a][0] = a0; acc[a][1] = a1;
        }
      }
      // ---- b = z: lower face carried from the previous plane, upper face k+1 evaluated now
      float Pz[3][2];
      zfaces(std::integral_constant<int, 1>{}, k, Zm1, C1, Zp1, Zp2, Pp, Pz);
#pragma unroll
      for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int e = 0; e < 2; e++) { float t = acc[a][e]; t = t + zf[a][e]; t = t - Pz[a][e]; acc[a][e] = t; zf[a][e] = Pz[a][e]; }
      }
      // ---- BDIM! (NoBody: μ₁ ≡ 0, V ≡ 0) with scale_u! folded: f = u⁰ + Δt·r ; u_out = (u·pre + μ₀·f)·post       src/Flow.jl:176-180
      const int Kg = g.gk + k;
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const float2 u0a = U0ADV ? C1[a] : u0v[a];
        const int I0[3] = {x + 1, y + 1, Kg + 1}, I1[3] = {x + 2, y + 1, Kg + 1};      // Julia indices of the two cells
        const float m00 = wl::wl_cl_coef(I0[a], N[a], bd.cl_c[a]), m01 = wl::wl_cl_coef(I1[a], N[a], bd.cl_c[a]);   // μ₀ of a verified NoBody field
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const float fn = sel(u0a, e) + bd.dt * acc[a][e] - 0.f;
          const float xx = (0.f / 2 + 0.f) + (e ? m01 : m00) * fn;
          float v = (bd.pre == 0.f) ? xx : (sel(C1[a], e) * bd.pre + xx);
          if (bd.scale_after) v = v * bd.post;
          if (e) un[a].y = v; else un[a].x = v;
        }
      }
      __syncthreads();     // plane k+2 is visible to the next iteration; nobody still reads the slot the next iteration overwrites
    }
    store_plane(wtag, ke - 1);
  };
  if (tile_walls) mainloop(std::integral_constant<int, 1>{}); else mainloop(std::integral_constant<int, 0>{});
}
// completes BC!(u,U) after a producer that folded the x/y part into its stores: plane 0 ← plane 1 and plane nz−1 ← plane nz−2 for the
// tangential components (whole planes, ghost rows and columns included: they were filled by the wall tiles), U for the normal component
// on planes 0, 1 and nz−1 (Julia indices 1, 2, N)       src/core.jl:200-219
__global__ void k_bc_zplanes(GridX g, float* __restrict__ u, float U2) {
  const long sz = g.sz, cs = g.cs;
  for (long m = (long)blockIdx.x * WL_BLOCK + threadIdx.x; m < sz; m += (long)gridDim.x * WL_BLOCK) {
    const long top = (long)(g.nz - 1) * sz;
    u[m] = u[sz + m]; u[cs + m] = u[cs + sz + m];                                  // plane 0: x and y components of plane 1
    u[top + m] = u[top - sz + m]; u[cs + top + m] = u[cs + top - sz + m];          // plane nz−1: those of plane nz−2
    u[2 * cs + m] = U2; u[2 * cs + sz + m] = U2; u[2 * cs + top + m] = U2;         // normal component: U on planes 0, 1, nz−1
  }
}
}  // namespace

namespace wl {
int bc_zplanes(float* u, const GridX& g, float U2, hipStream_t s) {
  hipLaunchKernelGGL(k_bc_zplanes, dim3(256), dim3(WL_BLOCK), 0, s, g, u, U2);
  WL_LAUNCH_CHECK(); return 0;
}
void conv_tile_enable(int on, int chunk) { g_convt_on = on; g_convt_chunk = chunk; }
void conv_tile_min(long tile_planes) { g_convt_min = tile_planes; }
// geometry the tiled kernel pays for: 3-D, 32-bit offsets over the three components, enough tile-planes to fill the chip
bool conv_tile_ok(const GridX& g, unsigned per, int nplanes) {
  if (!g_convt_on || g.D != 3 || per != 0) return false;
  if (3L * g.cs >= (1L << 31) || g.nx < 34 || g.ny < 18) return false;
  const long ntiles = (long)((g.nx - 2 + CT_CX - 1) / CT_CX) * ((g.ny - 2 + CT_CY - 1) / CT_CY);
  return nplanes >= (g_convt_min > 0 ? 8 : 1) && ntiles * nplanes >= g_convt_min;
}
// conv_diff!(·,u_adv) + BDIM!(NoBody, μ₀ evaluated: bd.cl_on) → u_out on the owned interior planes [ka,kb) (f is not materialised)
int conv_tile(const float* u_adv, const GridX& g, float nu, int scheme, int ka, int kb, const void* bdp, hipStream_t s) {
  const BdimArgs bd = *(const BdimArgs*)bdp;
  if (ka < g.k0) ka = g.k0;
  if (kb > g.k1) kb = g.k1;
  if (ka >= kb) return 0;
  const int ntiles = ((g.nx - 2 + CT_CX - 1) / CT_CX) * ((g.ny - 2 + CT_CY - 1) / CT_CY);
  const int per = (ntiles + 7) >> 3;
  const int np = kb - ka;
  static const int envc = wl_exp_int("WL_CT_CHUNK", 0);
  int zc = g_convt_chunk > 0 ? g_convt_chunk : envc;
  if (zc <= 0) { long t = (long)np * ntiles / 1024; zc = (int)(t < 5 ? 5 : (t > 64 ? 64 : t)); }   // ≈1024 workgroups (512 resident); small boxes: short chunks (128³: 0.62 -> 0.58 ms/step with 5 planes instead of 16)
  if (g_convt_min == 0 && !g_convt_chunk && !envc) zc = 5;   // tests: several chunks on a small box
  if (zc > np) zc = np;
  if (conv_flux_on()) return conv_flux(u_adv, g, nu, scheme, ka, kb, zc, bdp, s);   // every flux once (wl_convf.hip); this kernel stays as the A/B reference (option convf=0)
  const int nch = (np + zc - 1) / zc;
  const dim3 grid((unsigned)(8 * per * nch), 1, 1);
  const bool full = (g.nx - 2) % CT_CX == 0 && (g.ny - 2) % CT_CY == 0;
  const bool u0adv = bd.u0 == u_adv;
#define WL_CT(SCHV, FULLV, ADV) hipLaunchKernelGGL((k_conv_tile<SCHV, FULLV, ADV>), grid, dim3(CT_N), 0, s, g, u_adv, nu, ka, kb, zc, bd)
#define WL_CT2(SCHV) do { if (full) { if (u0adv) WL_CT(SCHV, 1, 1); else WL_CT(SCHV, 1, 0); } else { if (u0adv) WL_CT(SCHV, 0, 1); else WL_CT(SCHV, 0, 0); } } while (0)
  switch (scheme) {
    case WL_QUICK: WL_CT2(WL_QUICK); break;
    case WL_VANLEER: WL_CT2(WL_VANLEER); break;
    case WL_CDS: WL_CT2(WL_CDS); break;
    default: wl_set_error("unknown scheme"); return WL_EINVAL;
  }
#undef WL_CT2
#undef WL_CT
  WL_LAUNCH_CHECK(); return 0;
}
}  // namespace wl
