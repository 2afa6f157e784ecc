// Temporally blocked red–black Gauss–Seidel for gfx950: GaussSeidelRB!(it=4) (src/Poisson.jl:141-148) in TWO
// z-marching kernels instead of six passes over HBM.
//   kernel A:  ϵ = r·iD ; colour sweep 1 ; colour sweep 2                 R r,L (16 B/cell)               W ϵ_mid (4)
//   kernel B:  colour sweep 3 ; colour sweep 4 ; increment!(ω)            R ϵ_mid,r,L,x (24)              W r',x[,ϵ] (8-12)
// (D and iD are recomputed from the face coefficients L that the stencil needs anyway.)
// A workgroup owns an x-y tile of 64×16 threads (one cell column per thread) and marches along z.  Each stage lags
// the previous one by one plane, so a cell's z-neighbours are the thread's own registers; x-y neighbours come from LDS
// (two or three double-buffered 64×16 planes).  Tiles overlap by the dependency depth (2 resp. 3 cells per side): halo
// threads recompute their neighbours' values instead of waiting for them, and results are taken from the tile core
// only.  Because halo threads read inputs that the owning tile also updates, outputs never alias inputs: ϵ_mid, ϵ and
// r' are separate arrays (the handle swaps r↔r' afterwards); x is read and written by its owner only.
// Arithmetic (operation order, colour rule, quirk Q4) is exactly that of k_gs_sweep/k_increment ⇒ bit-identical results.
// Preconditions (checked by the caller): D==3, non-periodic, not a z-slab level, ghost cells of r, iD, ϵ are zero
// (true for the arrays a wl_mg handle owns), so `r·iD` of a ghost cell reproduces the stored ghost ϵ (=0).
#include <cstdint>

#include <cstdlib>

#include "wl_common.hpp"

#ifndef ZT_X
#define ZT_X 64
#endif
#ifndef ZT_Y
#define ZT_Y 16
#endif
#define ZT_N (ZT_X * ZT_Y)
#define ZT_LDS ((ZT_Y + 2) * ZT_X)   // one guard row above and below: neighbour indices never leave the array

namespace {

struct ZTile {
  int i, j;        // global cell column of this thread
  int li;          // LDS index of this thread
  bool indom;      // column exists (0 <= i < nx, 0 <= j < ny)
  bool inter;      // interior column (may be updated)
  bool core;       // this thread's results are stored (inside the tile core and interior)
  long oc;         // i + j*sy
  int ks, ke;      // output planes [ks,ke) of this workgroup
  bool alive;      // tile exists
};

template <int H>
__device__ __forceinline__ ZTile ztile(const GridX& g, int zchunk) {
  ZTile t;
  const int CX = ZT_X - 2 * H, CY = ZT_Y - 2 * H;
  const int ntx = (g.nx - 2 + CX - 1) / CX, nty = (g.ny - 2 + CY - 1) / CY;
  const int ntiles = ntx * nty;
  const unsigned h = blockIdx.x, q = h & 7u, s = h >> 3;
  const unsigned per = (unsigned)((ntiles + 7) >> 3);      // XCD q walks a contiguous range of tiles
  const int c = (int)(s / per);
  const int tl = (int)(q * per + (s - (unsigned)c * per));
  t.alive = tl < ntiles;
  const int tx = tl % ntx, ty = tl / ntx;
  const int lx = threadIdx.x % ZT_X, ly = threadIdx.x / ZT_X;
  t.i = 1 + tx * CX - H + lx;
  t.j = 1 + ty * CY - H + ly;
  t.li = ZT_X + lx + ly * ZT_X;
  t.indom = t.i >= 0 && t.i < g.nx && t.j >= 0 && t.j < g.ny;
  t.inter = t.i >= 1 && t.i <= g.nx - 2 && t.j >= 1 && t.j <= g.ny - 2;
  t.core = t.inter && lx >= H && lx < ZT_X - H && ly >= H && ly < ZT_Y - H;
  t.oc = (long)t.i + (long)t.j * g.sy;
  t.ks = g.k0 + c * zchunk;
  t.ke = t.ks + zchunk < g.k1 ? t.ks + zchunk : g.k1;
  return t;
}
__host__ __device__ inline int ztile_count(int nx, int ny, int H) {
  const int CX = ZT_X - 2 * H, CY = ZT_Y - 2 * H;
  return ((nx - 2 + CX - 1) / CX) * ((ny - 2 + CY - 1) / CY);
}
// plane K of the array is an interior plane of the grid (k0/k1 only delimit the planes a launch OUTPUTS: smooth! may hand a
// sub-range of the planes to these kernels and the rest to the pair kernels of wl_fused2.hip)
__device__ __forceinline__ bool pint1(const GridX& g, int K) { const int Kg = g.gk + K; return K >= 0 && K < g.nz && Kg >= 1 && Kg <= g.gnz - 2; }
// may the cell (i,j,K) be updated by colour sweep k0?  colour rule + quirk Q4 exactly as k_gs_sweep (D==3)
__device__ __forceinline__ bool gs_upd(const GridX& g, int i, int j, int K, int k0) {
  return (((i + j + K + 3 + k0) & 1) != 0) && !(K + 1 > 2 * (g.gnz / 2) - 1);
}
// gauss(I,r,L,iD,ϵ)  src/Poisson.jl:116-122, with the neighbours handed in
__device__ __forceinline__ float gs_val(float r, float iD, float exm, float exp_, float eym, float eyp, float ezm, float ezp, float lx, float lxp, float ly, float lyp, float lz, float lzp) {
  float s = r;
  s -= (exm * lx + exp_ * lxp);
  s -= (eym * ly + eyp * lyp);
  s -= (ezm * lz + ezp * lzp);
  return s * iD;
}

// D and iD are functions of the six face coefficients around a cell (set_diag!, src/Poisson.jl:43-55).  The kernels load those
// coefficients anyway, so both are recomputed in registers — same operation order, hence the same bits as the stored arrays —
// instead of streaming 8 B/cell from HBM.
__device__ __forceinline__ float diag6(float lx, float lxp, float ly, float lyp, float lz, float lzp) {
  float s = 0.f;
  s -= (lx + lxp);
  s -= (ly + lyp);
  s -= (lz + lzp);
  return s;
}
__device__ __forceinline__ float inv_diag(float d) { return (d == 0.f) ? d : 1.0f / d; }
// Constant-coefficient levels (wl::ConstL, verified on device by wl::check_const_L at update!): L[I,a] depends only on the
// Julia index of I along a — zero on the wall faces (index 1, 2 and N, BC!(μ₀,0) of src/Flow.jl:145 and
// src/MultiLevelPoisson.jl:47), the constant c_a elsewhere — so the kernels evaluate it instead of loading 12 B/cell.
__device__ __forceinline__ float cl_coef(int Ia, int Na, float c) { return (Ia <= 2 || Ia >= Na) ? 0.f : c; }

// ------------------------------------------------------------------------------------------------------------------
// kernel A.  PRO = 1 prepends the Vcycle!'s `prolongate!(ϵ,x_c); increment!(ω)` (src/MultiLevelPoisson.jl:99-100) as one
// more pipeline stage on the newest plane: r' = r − ω·A(x_c[down]) , x += ω·x_c[down]; sweeps then start from r'.
// Every thread (halo included) derives r' of its own column from global data only, so the tile halo stays 2.
// ------------------------------------------------------------------------------------------------------------------
struct ProArgs { const float* xc; float* x; float* rnew; GridX gc; int cx, cy, cz; float w; };
__device__ __forceinline__ int dwn(int i, int c) { return c ? (i + 1) / 2 : i; }   // down(I,c), 0-based   :7

template <int PRO, int CL>
__global__ void __launch_bounds__(ZT_N, 8) k_gsrb_A(GridX g, float* __restrict__ emid, const float* __restrict__ r, const float* __restrict__ L, int zchunk, ProArgs pa, wl::ConstL cl) {
  __shared__ float sA[2][ZT_LDS];   // ϵ⁰ of the newest plane           (x-y neighbours of sweep 1 one step later)
  __shared__ float sB[2][ZT_LDS];   // ϵ after sweep 1 of plane K-1     (x-y neighbours of sweep 2 one step later)
  const ZTile t = ztile<2>(g, zchunk);
  if (!t.alive) return;
  for (int q = threadIdx.x; q < ZT_LDS; q += ZT_N) { sA[0][q] = 0.f; sA[1][q] = 0.f; sB[0][q] = 0.f; sB[1][q] = 0.f; }
  const float* __restrict__ Lx = L; const float* __restrict__ Ly = L + g.cs; const float* __restrict__ Lz = L + 2 * g.cs;
  // rolling registers; index 0 = plane K, 1 = K-1, 2 = K-2, 3 = K-3
  float e0 = 0, e1 = 0, e2 = 0, e3 = 0;
  float r0 = 0, r1 = 0, r2 = 0, d0 = 0, d1 = 0, d2 = 0, lz0 = 0, lz1 = 0, lz2 = 0;
  float lx0 = 0, lxp0 = 0, ly0 = 0, lyp0 = 0, lx1 = 0, lxp1 = 0, ly1 = 0, lyp1 = 0, lx2 = 0, lxp2 = 0, ly2 = 0, lyp2 = 0;
  const int Kbeg = t.ks - 2, Kend = t.ke + 1;     // planes whose ϵ⁰ is needed
  // coarse columns under this cell and its x/y neighbours (PRO)
  long c00 = 0, cxm = 0, cxp = 0, cym = 0, cyp = 0;
  if (PRO && t.inter) {
    const long cj = (long)dwn(t.j, pa.cy) * pa.gc.sy;
    c00 = dwn(t.i, pa.cx) + cj; cxm = dwn(t.i - 1, pa.cx) + cj; cxp = dwn(t.i + 1, pa.cx) + cj;
    cym = dwn(t.i, pa.cx) + (long)dwn(t.j - 1, pa.cy) * pa.gc.sy; cyp = dwn(t.i, pa.cx) + (long)dwn(t.j + 1, pa.cy) * pa.gc.sy;
  }
  // in-plane coefficients of this column on a constant-coefficient level (same for every plane)
  const float kx = CL ? cl_coef(t.i + 1, g.nx, cl.c[0]) : 0.f, kxp = CL ? cl_coef(t.i + 2, g.nx, cl.c[0]) : 0.f;
  const float ky = CL ? cl_coef(t.j + 1, g.ny, cl.c[1]) : 0.f, kyp = CL ? cl_coef(t.j + 2, g.ny, cl.c[1]) : 0.f;
  // operands of the NEXT step are fetched one plane ahead so that their latency overlaps this step's barrier and arithmetic
  float n_r0, n_lz0 = 0, n_lzp = 0, n_lx = 0, n_lxp = 0, n_ly = 0, n_lyp = 0;
  auto fetch = [&](int K) {
    const bool pl0 = t.indom && K >= 0 && K <= g.nz - 1;
    const long o0 = t.oc + (long)K * g.sz;
    n_r0 = pl0 ? r[o0] : 0.f;
    const bool pll = t.inter && pint1(g, K);           // interior cell of plane K: its six face coefficients
    if (CL) {
      n_lz0 = pl0 ? cl_coef(g.gk + K + 1, g.gnz, cl.c[2]) : 0.f;
      n_lx = pll ? kx : 0.f; n_lxp = pll ? kxp : 0.f; n_ly = pll ? ky : 0.f; n_lyp = pll ? kyp : 0.f;
      n_lzp = pll ? cl_coef(g.gk + K + 2, g.gnz, cl.c[2]) : 0.f;
    } else {
      n_lz0 = pl0 ? Lz[o0] : 0.f;
      n_lx = pll ? Lx[o0] : 0.f; n_lxp = pll ? Lx[o0 + 1] : 0.f; n_ly = pll ? Ly[o0] : 0.f; n_lyp = pll ? Ly[o0 + g.sy] : 0.f;
      n_lzp = pll ? Lz[o0 + g.sz] : 0.f;
    }
  };
  fetch(Kbeg);
  for (int K = Kbeg; K <= Kend; K++) {
    // ---- shift the pipeline
    e3 = e2; e2 = e1; e1 = e0; r2 = r1; r1 = r0; d2 = d1; d1 = d0; lz2 = lz1; lz1 = lz0;
    lx2 = lx1; lxp2 = lxp1; ly2 = ly1; lyp2 = lyp1; lx1 = lx0; lxp1 = lxp0; ly1 = ly0; lyp1 = lyp0;
    r0 = n_r0; lz0 = n_lz0; lx0 = n_lx; lxp0 = n_lxp; ly0 = n_ly; lyp0 = n_lyp;
    const float lzp0 = n_lzp;
    const long o0 = t.oc + (long)K * g.sz;
    const bool pl = t.inter && pint1(g, K);
    const float dg0 = pl ? diag6(lx0, lxp0, ly0, lyp0, lz0, lzp0) : 0.f;     // D[I]   (ghost cells: stored D is 0)
    d0 = inv_diag(dg0);                                                      // iD[I]
    if (PRO && pl) {   // increment!(fine;ω) with ϵ = x_c[down(I)]          src/Poisson.jl:100-104, mult :70-76
      const int Kg = g.gk + K;
      const long ck0 = (long)(dwn(Kg, pa.cz) - pa.gc.gk) * pa.gc.sz, ckm = (long)(dwn(Kg - 1, pa.cz) - pa.gc.gk) * pa.gc.sz, ckp = (long)(dwn(Kg + 1, pa.cz) - pa.gc.gk) * pa.gc.sz;
      const float ep = pa.xc[c00 + ck0];
      float s = ep * dg0;
      s += (pa.xc[cxm + ck0] * lx0 + pa.xc[cxp + ck0] * lxp0);
      s += (pa.xc[cym + ck0] * ly0 + pa.xc[cyp + ck0] * lyp0);
      s += (pa.xc[c00 + ckm] * lz0 + pa.xc[c00 + ckp] * lzp0);
      r0 = r0 - pa.w * s;
      if (t.core && K >= t.ks && K < t.ke) { pa.rnew[o0] = r0; pa.x[o0] = pa.x[o0] + pa.w * ep; }
    }
    if (K < Kend) fetch(K + 1);
    const bool pl1 = t.inter && pint1(g, K - 1);
    e0 = r0 * d0;                                                   // ϵ = r·iD   :142 (ghost cells: 0·0)
    __syncthreads();                                                // LDS of the previous step is complete
    const int pb = (K - 1) & 1, cb = K & 1;
    // ---- sweep 1 on plane K-1 (its x-y neighbours: ϵ⁰ written one step ago)
    if (pl1 && (K - 1) >= t.ks - 1 && gs_upd(g, t.i, t.j, g.gk + K - 1, 1))
      e1 = gs_val(r1, d1, sA[pb][t.li - 1], sA[pb][t.li + 1], sA[pb][t.li - ZT_X], sA[pb][t.li + ZT_X], e2, e0, lx1, lxp1, ly1, lyp1, lz1, lz0);
    // ---- sweep 2 on plane K-2 (x-y neighbours: plane K-2 after sweep 1, written one step ago)
    const bool pl2 = t.inter && pint1(g, K - 2);
    if (pl2 && (K - 2) >= t.ks && gs_upd(g, t.i, t.j, g.gk + K - 2, 2))
      e2 = gs_val(r2, d2, sB[pb][t.li - 1], sB[pb][t.li + 1], sB[pb][t.li - ZT_X], sB[pb][t.li + ZT_X], e3, e1, lx2, lxp2, ly2, lyp2, lz2, lz1);
    sA[cb][t.li] = e0;
    sB[cb][t.li] = e1;
    if (t.core && (K - 2) >= t.ks && (K - 2) < t.ke) emid[o0 - 2 * g.sz] = e2;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// kernel B.  NORMS = 1 also reduces L₁ = Σ|r'| and L∞ = max|r'| of the new residual (src/Poisson.jl:190-191) per workgroup;
// EPS = 1 stores the final ϵ (p.ϵ of the reference; nothing on the path reads it again).
// ------------------------------------------------------------------------------------------------------------------
template <int NORMS, int EPS, int CL>
__global__ void __launch_bounds__(ZT_N, 8) k_gsrb_B(GridX g, float* __restrict__ eout, float* __restrict__ rout, float* __restrict__ x, const float* __restrict__ emid,
                                                    const float* __restrict__ r, const float* __restrict__ L, float w, int zchunk,
                                                    double* __restrict__ part, float* __restrict__ pmax, wl::ConstL cl) {
  __shared__ float sA[2][ZT_LDS];   // ϵ_mid of the newest plane                 (neighbours of sweep 3 one step later)
  __shared__ float sB[2][ZT_LDS];   // plane K-1 after sweep 3                   (neighbours of sweep 4 one step later)
  __shared__ float sC[2][ZT_LDS];   // plane K-2 after sweep 4 = final ϵ         (neighbours of increment! one step later)
  const ZTile t = ztile<3>(g, zchunk);
  if (!t.alive) { if (NORMS && threadIdx.x == 0) { part[blockIdx.x] = 0.0; pmax[blockIdx.x] = 0.f; } return; }
  for (int q = threadIdx.x; q < ZT_LDS; q += ZT_N) { sA[0][q] = 0.f; sA[1][q] = 0.f; sB[0][q] = 0.f; sB[1][q] = 0.f; sC[0][q] = 0.f; sC[1][q] = 0.f; }
  const float* __restrict__ Lx = L; const float* __restrict__ Ly = L + g.cs; const float* __restrict__ Lz = L + 2 * g.cs;
  float e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0;                     // planes K .. K-4
  float r1 = 0, r2 = 0, r3 = 0, d1 = 0, d2 = 0, dg1 = 0, dg2 = 0, dg3 = 0, lz0 = 0, lz1 = 0, lz2 = 0, lz3 = 0;
  float lx1 = 0, lxp1 = 0, ly1 = 0, lyp1 = 0, lx2 = 0, lxp2 = 0, ly2 = 0, lyp2 = 0, lx3 = 0, lxp3 = 0, ly3 = 0, lyp3 = 0;
  double nsum = 0.0; float nmax = 0.f;
  const int Kbeg = t.ks - 3, Kend = t.ke + 2;
  const float kx = CL ? cl_coef(t.i + 1, g.nx, cl.c[0]) : 0.f, kxp = CL ? cl_coef(t.i + 2, g.nx, cl.c[0]) : 0.f;
  const float ky = CL ? cl_coef(t.j + 1, g.ny, cl.c[1]) : 0.f, kyp = CL ? cl_coef(t.j + 2, g.ny, cl.c[1]) : 0.f;
  float n_e0, n_lz0 = 0, n_r1, n_lx1 = 0, n_lxp1 = 0, n_ly1 = 0, n_lyp1 = 0, n_x3;
  auto fetch = [&](int K) {
    const bool pl0 = t.indom && K >= 0 && K <= g.nz - 1;
    const long o0 = t.oc + (long)K * g.sz;
    n_e0 = pl0 ? emid[o0] : 0.f;
    const bool pl1 = t.inter && pint1(g, K - 1);
    const long o1 = o0 - g.sz;
    n_r1 = pl1 ? r[o1] : 0.f;
    if (CL) {
      n_lz0 = pl0 ? cl_coef(g.gk + K + 1, g.gnz, cl.c[2]) : 0.f;
      n_lx1 = pl1 ? kx : 0.f; n_lxp1 = pl1 ? kxp : 0.f; n_ly1 = pl1 ? ky : 0.f; n_lyp1 = pl1 ? kyp : 0.f;
    } else {
      n_lz0 = pl0 ? Lz[o0] : 0.f;
      n_lx1 = pl1 ? Lx[o1] : 0.f; n_lxp1 = pl1 ? Lx[o1 + 1] : 0.f; n_ly1 = pl1 ? Ly[o1] : 0.f; n_lyp1 = pl1 ? Ly[o1 + g.sy] : 0.f;
    }
    const bool pl3 = t.core && (K - 3) >= t.ks && (K - 3) < t.ke;
    n_x3 = pl3 ? x[o0 - 3 * g.sz] : 0.f;
  };
  fetch(Kbeg);
  for (int K = Kbeg; K <= Kend; K++) {
    e4 = e3; e3 = e2; e2 = e1; e1 = e0; r3 = r2; r2 = r1; d2 = d1; dg3 = dg2; dg2 = dg1; lz3 = lz2; lz2 = lz1; lz1 = lz0;
    lx3 = lx2; lxp3 = lxp2; ly3 = ly2; lyp3 = lyp2; lx2 = lx1; lxp2 = lxp1; ly2 = ly1; lyp2 = lyp1;
    e0 = n_e0; lz0 = n_lz0; r1 = n_r1; lx1 = n_lx1; lxp1 = n_lxp1; ly1 = n_ly1; lyp1 = n_lyp1;
    const float x3 = n_x3;
    if (K < Kend) fetch(K + 1);
    const bool pl1 = t.inter && pint1(g, K - 1);
    dg1 = pl1 ? diag6(lx1, lxp1, ly1, lyp1, lz1, lz0) : 0.f;        // D of plane K-1 (its Lz[I+δz] is plane K's Lz)
    d1 = inv_diag(dg1);
    const bool pl3 = t.core && (K - 3) >= t.ks && (K - 3) < t.ke;
    const long o3 = t.oc + (long)(K - 3) * g.sz;
    __syncthreads();
    const int pb = (K - 1) & 1, cb = K & 1;
    // ---- sweep 3 on plane K-1
    if (pl1 && (K - 1) >= t.ks - 2 && gs_upd(g, t.i, t.j, g.gk + K - 1, 3))
      e1 = gs_val(r1, d1, sA[pb][t.li - 1], sA[pb][t.li + 1], sA[pb][t.li - ZT_X], sA[pb][t.li + ZT_X], e2, e0, lx1, lxp1, ly1, lyp1, lz1, lz0);
    // ---- sweep 4 on plane K-2
    const bool pl2 = t.inter && pint1(g, K - 2);
    if (pl2 && (K - 2) >= t.ks - 1 && gs_upd(g, t.i, t.j, g.gk + K - 2, 4))
      e2 = gs_val(r2, d2, sB[pb][t.li - 1], sB[pb][t.li + 1], sB[pb][t.li - ZT_X], sB[pb][t.li + ZT_X], e3, e1, lx2, lxp2, ly2, lyp2, lz2, lz1);
    // ---- increment! on plane K-3: r' = r − ω·Aϵ ; x += ω·ϵ          src/Poisson.jl:100-104, mult :70-76
    if (pl3) {
      float s = e3 * dg3;
      s += (sC[pb][t.li - 1] * lx3 + sC[pb][t.li + 1] * lxp3);
      s += (sC[pb][t.li - ZT_X] * ly3 + sC[pb][t.li + ZT_X] * lyp3);
      s += (e4 * lz3 + e2 * lz2);
      const float rn = r3 - w * s;
      rout[o3] = rn;
      x[o3] = x3 + w * e3;
      if (EPS) eout[o3] = e3;
      if (NORMS) { const float av = fabsf(rn); nsum += (double)av; nmax = fmaxf(nmax, av); }
    }
    sA[cb][t.li] = e0;
    sB[cb][t.li] = e1;
    sC[cb][t.li] = e2;
  }
  if (NORMS) {   // 16 waves -> one partial per workgroup
    __shared__ double shs[ZT_N / 64]; __shared__ float shm[ZT_N / 64];
    nsum = wave_sum(nsum); nmax = wave_max(nmax);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { shs[threadIdx.x >> 6] = nsum; shm[threadIdx.x >> 6] = nmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
      double a = 0.0; float mx = 0.f;
      for (int q = 0; q < ZT_N / 64; q++) { a += shs[q]; mx = fmaxf(mx, shm[q]); }
      part[blockIdx.x] = a; pmax[blockIdx.x] = mx;
    }
  }
}
}  // namespace

namespace wl {
// Eligibility of a level for the fused smoother
bool gsrb_fused_ok(const GridX& g, unsigned per, bool dist) {
  return g.D == 3 && per == 0 && !dist && g.nz == g.gnz && g.nx >= 34 && g.ny >= 18 && (g.k1 - g.k0) >= 8;
}
static int zchunk_for(const GridX& g, int H) {
  const int nt = ztile_count(g.nx, g.ny, H);
  const int np = g.k1 - g.k0;
  static const int zmin_env = wl_exp_int("WL_ZC_MIN", 0);
  if ((long)nt * ((np + 31) / 32) >= 2048) {   // many rounds of workgroups: long marches (the pipeline warm-up costs 2H-1 planes per chunk)
    const int chunks = (1536 + nt - 1) / nt;
    int zc = (np + chunks - 1) / chunks; if (zc < 16) zc = 16; if (zc > np) zc = np;
    return zc;
  }
  // few rounds (same reasoning as zchunk2 of wl_fused2.hip): minimise rounds of 512 workgroups × (planes per chunk + warm-up)
  const int warm = (H == 3 ? 5 : 3) + 3;
  long best = -1; int best_zc = np;
  for (int chunks = 1; chunks <= np; chunks++) {
    const int zc = (np + chunks - 1) / chunks;
    if (zc < (zmin_env ? zmin_env : 4)) break;
    const int nch = (np + zc - 1) / zc;
    const long W = (long)nt * nch;
    long cost = ((W + 511) / 512) * (zc + warm);
    if (W < 512) cost = (long)((zc + warm) * 1.25);
    if (best < 0 || cost < best) { best = cost; best_zc = zc; }
  }
  return best_zc;
}
// GaussSeidelRB!(it=4,ω): emid and rout are scratch arrays of the level (ghosts zero); on return eps holds the final ϵ,
// rout the new residual (caller swaps r<->rout) and x is updated in place.
static inline bool al8(const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr, const void* e = nullptr) {
  return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d | (uintptr_t)e) & 7u) == 0;
}
int gsrb_fused_A(float* emid, const float* r, const float* L, const GridX& g, const ConstL& cl, hipStream_t s) {
  if (gsrb_pair_ok(g, cl) && al8(emid, r)) return gsrb_pair_A(emid, r, g, cl, s);
  if (g.nz != g.gnz) { wl_set_error("blocked smoother on a z-slab level needs the pair kernels"); return WL_EINVAL; }
  const int zc = zchunk_for(g, 2);
  const int nt = ztile_count(g.nx, g.ny, 2), per = (nt + 7) >> 3, nch = (g.k1 - g.k0 + zc - 1) / zc;
  ProArgs pa{};
  const dim3 grid((unsigned)(8 * per * nch));
  if (cl.on) hipLaunchKernelGGL((k_gsrb_A<0, 1>), grid, dim3(ZT_N), 0, s, g, emid, r, L, zc, pa, cl);
  else hipLaunchKernelGGL((k_gsrb_A<0, 0>), grid, dim3(ZT_N), 0, s, g, emid, r, L, zc, pa, cl);
  WL_LAUNCH_CHECK(); return 0;
}
// prolongate!+increment!(ω) of the V-cycle folded into kernel A: r' -> rnew (≠ r), x updated in place, ϵ_mid from r'
int gsrb_fused_A_pro(float* emid, float* rnew, float* x, const float* r, const float* xc, const float* L, const GridX& g, const GridX& gc, float w, const ConstL& cl, hipStream_t s,
                     int xk0, int xk1, bool* defer_x, bool range) {
  if ((range ? gsrb_pair_ok_range(g, cl) : gsrb_pair_ok(g, cl)) && gc.cs < (1L << 30) && al8(emid, rnew, x, r)) {
    if (defer_x && *defer_x) { xk0 = 0; xk1 = 0; }    // x is left to kernel B (wl::XDefer)
    return gsrb_pair_A_pro(emid, rnew, x, r, xc, g, gc, w, cl, s, xk0, xk1);
  }
  if (defer_x) *defer_x = false;
  if (g.nz != g.gnz) { wl_set_error("blocked smoother on a z-slab level needs the pair kernels"); return WL_EINVAL; }
  const int zc = zchunk_for(g, 2);
  const int nt = ztile_count(g.nx, g.ny, 2), per = (nt + 7) >> 3, nch = (g.k1 - g.k0 + zc - 1) / zc;
  ProArgs pa{xc, x, rnew, gc, gc.nx < g.nx, gc.ny < g.ny, gc.gnz < g.gnz, w};
  const dim3 grid((unsigned)(8 * per * nch));
  if (cl.on) hipLaunchKernelGGL((k_gsrb_A<1, 1>), grid, dim3(ZT_N), 0, s, g, emid, (const float*)r, L, zc, pa, cl);
  else hipLaunchKernelGGL((k_gsrb_A<1, 0>), grid, dim3(ZT_N), 0, s, g, emid, (const float*)r, L, zc, pa, cl);
  WL_LAUNCH_CHECK(); return 0;
}
// ws != NULL: also leaves L₁/L∞ of the new residual in ws->res_d[slot_d] / ws->res_f[slot_f] (device); eps == NULL: final ϵ not stored
bool gsrb_pair_B_ok(const float* eps, const float* rout, const float* x, const float* emid, const float* r, const GridX& g, const ConstL& cl) {
  return gsrb_pair_ok(g, cl) && al8(eps, rout, x, emid, r);
}
int gsrb_fused_B(float* eps, float* rout, float* x, const float* emid, const float* r, const float* L, const GridX& g, float w,
                 const RedWs* ws, int slot_d, int slot_f, const ConstL& cl, hipStream_t s, const XDefer* xd) {
  if (gsrb_pair_B_ok(eps, rout, x, emid, r, g, cl)) return gsrb_pair_B(eps, rout, x, emid, r, g, w, ws, slot_d, slot_f, cl, s, xd);
  if (xd) { wl_set_error("gsrb_fused_B: a deferred x increment needs the pair kernel"); return WL_EINVAL; }
  if (g.nz != g.gnz) { wl_set_error("blocked smoother on a z-slab level needs the pair kernels"); return WL_EINVAL; }
  const int zc = zchunk_for(g, 3);
  const int nt = ztile_count(g.nx, g.ny, 3), per = (nt + 7) >> 3, nch = (g.k1 - g.k0 + zc - 1) / zc;
  const unsigned nb = (unsigned)(8 * per * nch);
  const bool norms = ws && nb <= WL_MAXPART;
  double* pa = norms ? ws->pa : nullptr; float* pm = norms ? ws->pm : nullptr;
#define WL_GB(NF, EF, CF) hipLaunchKernelGGL((k_gsrb_B<NF, EF, CF>), dim3(nb), dim3(ZT_N), 0, s, g, eps, rout, x, emid, r, L, w, zc, pa, pm, cl)
#define WL_GB2(NF, EF) do { if (cl.on) WL_GB(NF, EF, 1); else WL_GB(NF, EF, 0); } while (0)
  if (norms) { if (eps) WL_GB2(1, 1); else WL_GB2(1, 0); } else { if (eps) WL_GB2(0, 1); else WL_GB2(0, 0); }
#undef WL_GB2
#undef WL_GB
  if (norms) WL_TRY(finalize_sum_max(*ws, (int)nb, slot_d, slot_f, s));
  else if (ws) WL_TRY(norms_dev(rout, g, *ws, slot_d, slot_f, s));
  WL_LAUNCH_CHECK(); return 0;
}
}  // namespace wl
