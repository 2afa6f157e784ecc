// Temporally blocked GaussSeidelRB!(it=4) (src/Poisson.jl:141-148) for CONSTANT-COEFFICIENT levels (wl::ConstL, the
// NoBody hierarchy): "pair" variant of the z-marching kernels A/B of wl_fused.hip.
//
// Why a second variant: kernels A/B of wl_fused.hip are issue-bound, not bandwidth-bound (≈230 instructions per cell and
// plane; in every colour sweep half of the lanes idle because red and black cells alternate along x).  Here
//   * a thread owns TWO x-adjacent cells (one red, one black): every lane does exactly one Gauss–Seidel update per sweep
//     — which of its two cells is a per-row select, not divergence; one x-neighbour is the thread's own other cell;
//   * L, D, iD are never loaded nor pipelined: in-plane coefficients are per-thread constants, the z ones per-plane
//     scalars, iD comes from a 2-entry per-thread table (interior plane / plane next to a z-wall) — no division in the loop;
//   * global accesses are 8-byte (float2) with 32-bit offsets; LDS planes are split into an even-x and an odd-x array
//     so that all LDS reads are unit-stride (no bank conflicts);
//   * workgroup tile 64×32 cells (32×32 threads): 71 % of the thread-work is tile core (64×16 threads: 57 %).
// Arithmetic per cell (operation order, colour rule, quirk Q4) is that of k_gs_sweep/k_increment ⇒ bit-identical results
// (IEEE addition is commutative, which is all the operand re-pairing relies on).
// Preconditions (checked by gsrb_pair_ok): everything gsrb_fused_ok needs, plus cl.on, nx even (float2 alignment),
// and < 2^30 elements per field (32-bit offsets).
#include <cstdlib>

#include "wl_common.hpp"

#define PT_X 32                      // threads along x (two cells each)
#define PT_Y 32
#define PT_N (PT_X * PT_Y)
#define PL_W (PT_X + 2)              // LDS row: one guard entry on each side
#define PL_H ((PT_Y + 2) * PL_W)     // one parity array (guard rows above and below)
#define PL_SZ (2 * PL_H)             // a plane = even-x array followed by odd-x array

namespace {

struct PTile {
  int i0, j;         // cell columns (i0, i0+1) × row j, 0-based with ghosts; i0 is even
  int lq;            // LDS index inside a parity array
  bool indom;        // the pair exists in the array
  bool in0, in1;     // cell 0 / cell 1 is an interior cell (may be updated)
  bool st0, st1;     // cell is stored by this workgroup (tile core ∧ interior)
  unsigned oc;       // i0 + j*sy
  int ks, ke;        // output planes [ks,ke)
  bool alive;
};

// HX (even) / HY: halo in cells.  Core of tile (tx,ty): cells [tx*CX, (tx+1)*CX) × rows [1+ty*CY, 1+(ty+1)*CY)
template <int HX, int HY>
__device__ __forceinline__ PTile ptile(const GridX& g, int zchunk) {
  PTile t;
  const int CX = 2 * PT_X - 2 * HX, CY = PT_Y - 2 * HY;
  const int ntx = (g.nx - 1 + CX - 1) / CX, nty = (g.ny - 2 + CY - 1) / CY;
  const int ntiles = ntx * nty;
  const unsigned h = blockIdx.x, q = h & 7u, s = h >> 3;
  const unsigned per = (unsigned)((ntiles + 7) >> 3);      // XCD q walks a contiguous range of tiles
  const int c = (int)(s / per);
  const int tl = (int)(q * per + (s - (unsigned)c * per));
  t.alive = tl < ntiles;
  const int tx = tl % ntx, ty = tl / ntx;
  const int lx = threadIdx.x % PT_X, ly = threadIdx.x / PT_X;
  t.i0 = tx * CX - HX + 2 * lx;
  t.j = 1 + ty * CY - HY + ly;
  t.lq = (ly + 1) * PL_W + lx + 1;
  t.indom = t.i0 >= 0 && t.i0 <= g.nx - 2 && t.j >= 0 && t.j < g.ny;
  const bool jin = t.j >= 1 && t.j <= g.ny - 2;
  t.in0 = t.indom && jin && t.i0 >= 2;
  t.in1 = t.indom && jin && t.i0 + 1 <= g.nx - 2;
  const bool corep = 2 * lx >= HX && 2 * lx < 2 * PT_X - HX && ly >= HY && ly < PT_Y - HY;
  t.st0 = corep && t.in0; t.st1 = corep && t.in1;
  t.oc = t.indom ? (unsigned)t.i0 + (unsigned)t.j * (unsigned)g.sy : 0u;
  t.ks = g.k0 + c * zchunk;
  t.ke = t.ks + zchunk < g.k1 ? t.ks + zchunk : g.k1;
  return t;
}
__host__ __device__ inline int ptile_count(int nx, int ny, int HX, int HY) {
  const int CX = 2 * PT_X - 2 * HX, CY = PT_Y - 2 * HY;
  return ((nx - 1 + CX - 1) / CX) * ((ny - 2 + CY - 1) / CY);
}
__device__ __forceinline__ float cf(int Ia, int Na, float c) { return (Ia <= 2 || Ia >= Na) ? 0.f : c; }
__device__ __forceinline__ float invd(float d) { return (d == 0.f) ? d : 1.0f / d; }
__device__ __forceinline__ float2 ld2(const float* __restrict__ p, unsigned o) { return *reinterpret_cast<const float2*>(p + o); }
__device__ __forceinline__ void st2(float* __restrict__ p, unsigned o, float2 v, bool s0, bool s1) {
  if (s0 && s1) *reinterpret_cast<float2*>(p + o) = v;
  else { if (s0) p[o] = v.x; if (s1) p[o + 1] = v.y; }
}
// plane K of the local array is an interior plane of the GLOBAL grid (single domain: k0 <= K < k1; on a z-slab also the
// neighbour's planes held as ghost planes, which the tile pipeline recomputes instead of waiting for them)
__device__ __forceinline__ bool pint(const GridX& g, int K) { const int Kg = g.gk + K; return K >= 0 && K < g.nz && Kg >= 1 && Kg <= g.gnz - 2; }
// quirk Q4 (half_rangek, src/Poisson.jl:123-132): an odd last dimension leaves its last plane unswept
__device__ __forceinline__ bool q4ok(const GridX& g, int Kg) { return !(Kg + 1 > 2 * (g.gnz / 2) - 1); }

// Per-thread constants of a constant-coefficient level
struct PCoef {
  float cxa, cxb, cxc;     // x-face coefficients: left of cell 0, between the cells, right of cell 1
  float ky, kyp;           // y-face coefficients below / above the row
  float dxy0, dxy1;        // (0 − (lx+lxp)) − (ly+lyp) of the two cells: set_diag!'s partial sum (src/Poisson.jl:49-55)
  float ide0, ide1;        // iD next to a z-wall (one z-face coefficient is 0), 0 for a ghost cell
  float idm0, idm1;        // iD with both z-faces open
};
__device__ __forceinline__ PCoef pcoef(const GridX& g, const PTile& t, const wl::ConstL& cl) {
  PCoef k;
  k.cxa = cf(t.i0 + 1, g.nx, cl.c[0]); k.cxb = cf(t.i0 + 2, g.nx, cl.c[0]); k.cxc = cf(t.i0 + 3, g.nx, cl.c[0]);
  k.ky = cf(t.j + 1, g.ny, cl.c[1]); k.kyp = cf(t.j + 2, g.ny, cl.c[1]);
  float s = 0.f; s -= (k.cxa + k.cxb); s -= (k.ky + k.kyp); k.dxy0 = s;
  s = 0.f; s -= (k.cxb + k.cxc); s -= (k.ky + k.kyp); k.dxy1 = s;
  const float ze = cl.c[2] + 0.f, zm = cl.c[2] + cl.c[2];     // lz+lzp: 0+c == c+0
  k.ide0 = t.in0 ? invd(k.dxy0 - ze) : 0.f; k.ide1 = t.in1 ? invd(k.dxy1 - ze) : 0.f;
  k.idm0 = t.in0 ? invd(k.dxy0 - zm) : 0.f; k.idm1 = t.in1 ? invd(k.dxy1 - zm) : 0.f;
  return k;
}

// One colour sweep on a pair: the active cell (act1 ? cell 1 : cell 0) gets gauss(I,…) (src/Poisson.jl:116-122).
//   S: LDS plane holding the x-y neighbours; ep: the plane's pair (updated); em/en: the planes below/above; rp: r; d: iD.
__device__ __forceinline__ void pair_sweep(const float* __restrict__ S, int off_oth, int off_y, bool act1, bool gate, float lxo, const PCoef& k,
                                           float lzl, float lzu, const float2& rp, const float2& d, const float2& em, const float2& en, float2& ep) {
  const float oth = S[off_oth];                          // the x-neighbour owned by the adjacent thread
  const float ym = S[off_y - PL_W], yp = S[off_y + PL_W];
  const float partv = act1 ? ep.x : ep.y;                // the x-neighbour inside the pair
  float s = act1 ? rp.y : rp.x;
  s -= (oth * lxo + partv * k.cxb);                      // ϵ[I−δx]·L[I,1] + ϵ[I+δx]·L[I+δx,1] (either order, same bits)
  s -= (ym * k.ky + yp * k.kyp);
  s -= ((act1 ? em.y : em.x) * lzl + (act1 ? en.y : en.x) * lzu);
  const float v = s * (act1 ? d.y : d.x);
  if (gate) { if (act1) ep.y = v; else ep.x = v; }
}

struct ProArgs2 { const float* xc; float* x; float* rnew; GridX gc; int cx, cy, cz; float w; };
__device__ __forceinline__ int dwn(int i, int c) { return c ? (i + 1) / 2 : i; }   // down(I,c), 0-based  src/MultiLevelPoisson.jl:7

// ------------------------------------------------------------------------------------------------------------------
// kernel A:  [PRO: r' = r − ω·A(x_c↓), x += ω·x_c↓  (Vcycle!'s prolongate!+increment!, src/MultiLevelPoisson.jl:99-100)]
//            ϵ = r·iD ; colour sweep 1 ; colour sweep 2  →  ϵ_mid
// ------------------------------------------------------------------------------------------------------------------
template <int PRO>
__global__ void __launch_bounds__(PT_N, 8) k_gsrb2_A(GridX g, float* __restrict__ emid, const float* __restrict__ r, int zchunk, ProArgs2 pa, wl::ConstL cl) {
  __shared__ float sA[2][PL_SZ];   // ϵ⁰ of the newest plane
  __shared__ float sB[2][PL_SZ];   // plane K-1 after sweep 1
  const PTile t = ptile<2, 2>(g, zchunk);
  if (!t.alive) return;
  for (int q = threadIdx.x; q < PL_SZ; q += PT_N) { sA[0][q] = 0.f; sA[1][q] = 0.f; sB[0][q] = 0.f; sB[1][q] = 0.f; }
  const PCoef k = pcoef(g, t, cl);
  const float c2 = cl.c[2];
  const int jpar = (t.j + g.gk) & 1;
  const int oth0 = PL_H + t.lq - 1, oth1 = t.lq + 1;       // cell 0 active: odd cell of the left thread; cell 1: even cell of the right thread
  float2 e0 = {0.f, 0.f}, e1 = e0, e2 = e0, e3 = e0, r0 = e0, r1 = e0, r2 = e0, n_r0;
  const int Kbeg = t.ks - 2, Kend = t.ke + 1;
  // coarse columns under the pair and its x-neighbours (PRO)
  unsigned q0 = 0, q1 = 0, q2 = 0, q3 = 0, cj0 = 0, cjm = 0, cjp = 0;
  if (PRO == 1 && t.indom && t.j >= 1 && t.j <= g.ny - 2) {
    q0 = (unsigned)dwn(t.i0 > 0 ? t.i0 - 1 : 0, pa.cx); q1 = (unsigned)dwn(t.i0, pa.cx); q2 = (unsigned)dwn(t.i0 + 1, pa.cx);
    q3 = (unsigned)dwn(t.i0 + 2 <= g.nx - 1 ? t.i0 + 2 : g.nx - 1, pa.cx);
    cj0 = (unsigned)dwn(t.j, pa.cy) * (unsigned)pa.gc.sy; cjm = (unsigned)dwn(t.j - 1, pa.cy) * (unsigned)pa.gc.sy; cjp = (unsigned)dwn(t.j + 1, pa.cy) * (unsigned)pa.gc.sy;
  }
  // PRO == 2: every direction is coarsened (the usual case).  The pair (i0,i0+1) and its x-neighbours lie over the two coarse
  // columns qa, qa+1; of the rows j±1 (planes K±1) one lies over the cell's own coarse row (plane), the other over the
  // next one — by the parity of j (K).  Six coarse values per step, kept in registers: Cc (own plane, own row), Co (own
  // plane, other row), Cz (other plane, own row); when K advances either Cz or — after Cc and Cz trade places — Co is
  // replaced by two values fetched one step ahead.  2 loads per step instead of 12.
  const bool jodd = (t.j & 1) != 0;
  unsigned rc = 0, ro = 0;                 // qa + coarse row offset (own / other row)
  const int nzc = pa.gc.nz;
  float2 Cc = {0.f, 0.f}, Co = Cc, Cz = Cc, n_c = Cc, n_x0 = Cc;
  auto cpl = [&](int p) { return (unsigned)(p < 0 ? 0 : (p > nzc - 1 ? nzc - 1 : p)) * (unsigned)pa.gc.sz; };
  if (PRO == 2) {
    if (t.indom) {
      const unsigned qa = (unsigned)(t.i0 >> 1), cr = (unsigned)((t.j + 1) >> 1);
      rc = qa + cr * (unsigned)pa.gc.sy;
      ro = qa + (jodd ? cr - 1u : cr + 1u) * (unsigned)pa.gc.sy;
    }
    const int m = (Kbeg + 1) >> 1, zo = (Kbeg & 1) ? m - 1 : m + 1;
    Cc = make_float2(pa.xc[rc + cpl(m)], pa.xc[rc + cpl(m) + 1]);
    Co = make_float2(pa.xc[ro + cpl(m)], pa.xc[ro + cpl(m) + 1]);
    Cz = make_float2(pa.xc[rc + cpl(zo)], pa.xc[rc + cpl(zo) + 1]);
  }
  auto fetch = [&](int K) {
    const bool pl0 = t.indom && pint(g, K);                     // r of ghost planes/cells is 0
    const unsigned o = t.oc + (unsigned)K * (unsigned)g.sz;
    n_r0 = pl0 ? ld2(r, o) : make_float2(0.f, 0.f);
    if (PRO) n_x0 = ((t.st0 || t.st1) && K >= t.ks && K < t.ke) ? ld2(pa.x, o) : make_float2(0.f, 0.f);
    if (PRO == 2) {   // coarse values that become current at step K: plane m(K-1)+1; own row if K is even, other row if odd
      const unsigned a = ((K & 1) ? ro : rc) + cpl((K >> 1) + 1);
      n_c = make_float2(pa.xc[a], pa.xc[a + 1]);
    }
  };
  fetch(Kbeg);
  n_c = Cz;    // (the state for Kbeg was loaded directly; make the first transition below a no-op)
  for (int K = Kbeg; K <= Kend; K++) {
    e3 = e2; e2 = e1; e1 = e0; r2 = r1; r1 = r0; r0 = n_r0;
    const float2 x0 = n_x0;
    if (PRO == 2 && K > Kbeg) {
      if (K & 1) { const float2 tmp = Cc; Cc = Cz; Cz = tmp; Co = n_c; }
      else Cz = n_c;
    }
    const unsigned o0 = t.oc + (unsigned)K * (unsigned)g.sz;
    const bool plK = pint(g, K);
    const float lz0 = cf(g.gk + K + 1, g.gnz, c2), lzp0 = cf(g.gk + K + 2, g.gnz, c2);          // z-faces below / above plane K
    const float lz1 = cf(g.gk + K, g.gnz, c2), lz2 = cf(g.gk + K - 1, g.gnz, c2);
    if (PRO && plK) {   // increment!(fine;ω) with ϵ = x_c[down(I)]          src/Poisson.jl:100-104, mult :70-76
      float va, v0, v1, vb, ym0, yp0, ym1, yp1, zm0, zp0, zm1, zp1;
      if (PRO == 2) {
        const bool kodd = (K & 1) != 0;
        va = Cc.x; v0 = Cc.x; v1 = Cc.y; vb = Cc.y;
        ym0 = jodd ? Co.x : Cc.x; yp0 = jodd ? Cc.x : Co.x; ym1 = jodd ? Co.y : Cc.y; yp1 = jodd ? Cc.y : Co.y;
        zm0 = kodd ? Cz.x : Cc.x; zp0 = kodd ? Cc.x : Cz.x; zm1 = kodd ? Cz.y : Cc.y; zp1 = kodd ? Cc.y : Cz.y;
      } else {
        const int Kg = g.gk + K;
        const unsigned ck0 = (unsigned)(dwn(Kg, pa.cz) - pa.gc.gk) * (unsigned)pa.gc.sz, ckm = (unsigned)(dwn(Kg - 1, pa.cz) - pa.gc.gk) * (unsigned)pa.gc.sz,
                       ckp = (unsigned)(dwn(Kg + 1, pa.cz) - pa.gc.gk) * (unsigned)pa.gc.sz;
        const float* __restrict__ xc = pa.xc;
        va = xc[q0 + cj0 + ck0]; v0 = xc[q1 + cj0 + ck0]; v1 = xc[q2 + cj0 + ck0]; vb = xc[q3 + cj0 + ck0];
        ym0 = xc[q1 + cjm + ck0]; yp0 = xc[q1 + cjp + ck0]; ym1 = xc[q2 + cjm + ck0]; yp1 = xc[q2 + cjp + ck0];
        zm0 = xc[q1 + cj0 + ckm]; zp0 = xc[q1 + cj0 + ckp]; zm1 = xc[q2 + cj0 + ckm]; zp1 = xc[q2 + cj0 + ckp];
      }
      const float zs = lz0 + lzp0;
      float s = v0 * (t.in0 ? k.dxy0 - zs : 0.f);
      s += (va * k.cxa + v1 * k.cxb);
      s += (ym0 * k.ky + yp0 * k.kyp);
      s += (zm0 * lz0 + zp0 * lzp0);
      if (t.in0) r0.x = r0.x - pa.w * s;
      s = v1 * (t.in1 ? k.dxy1 - zs : 0.f);
      s += (v0 * k.cxb + vb * k.cxc);
      s += (ym1 * k.ky + yp1 * k.kyp);
      s += (zm1 * lz0 + zp1 * lzp0);
      if (t.in1) r0.y = r0.y - pa.w * s;
      if ((t.st0 || t.st1) && K >= t.ks && K < t.ke) {
        st2(pa.rnew, o0, r0, t.st0, t.st1);
        st2(pa.x, o0, make_float2(x0.x + pa.w * v0, x0.y + pa.w * v1), t.st0, t.st1);
      }
    }
    if (K < Kend) fetch(K + 1);
    {   // ϵ = r·iD   :142   (ghost cells and planes: iD = 0)
      const bool edge = (lz0 == 0.f) || (lzp0 == 0.f);
      const float d0x = plK ? (edge ? k.ide0 : k.idm0) : 0.f, d0y = plK ? (edge ? k.ide1 : k.idm1) : 0.f;
      e0.x = r0.x * d0x; e0.y = r0.y * d0y;
    }
    __syncthreads();                                                // LDS of the previous step is complete
    const int pb = (K - 1) & 1, cb = K & 1;
    const bool act1 = ((jpar ^ K) & 1) != 0;
    const bool gate = act1 ? t.in1 : t.in0;
    const int off_oth = act1 ? oth1 : oth0, off_y = act1 ? t.lq + PL_H : t.lq;
    const float lxo = act1 ? k.cxc : k.cxa;
    // ---- sweep 1 on plane K-1
    if (pint(g, K - 1) && (K - 1) >= t.ks - 1 && q4ok(g, g.gk + K - 1)) {
      const bool edge = (lz1 == 0.f) || (lz0 == 0.f);
      const float2 d = make_float2(edge ? k.ide0 : k.idm0, edge ? k.ide1 : k.idm1);
      pair_sweep(sA[pb], off_oth, off_y, act1, gate, lxo, k, lz1, lz0, r1, d, e2, e0, e1);
    }
    // ---- sweep 2 on plane K-2
    if (pint(g, K - 2) && (K - 2) >= t.ks && q4ok(g, g.gk + K - 2)) {
      const bool edge = (lz2 == 0.f) || (lz1 == 0.f);
      const float2 d = make_float2(edge ? k.ide0 : k.idm0, edge ? k.ide1 : k.idm1);
      pair_sweep(sB[pb], off_oth, off_y, act1, gate, lxo, k, lz2, lz1, r2, d, e3, e1, e2);
    }
    sA[cb][t.lq] = e0.x; sA[cb][t.lq + PL_H] = e0.y;
    sB[cb][t.lq] = e1.x; sB[cb][t.lq + PL_H] = e1.y;
    if ((t.st0 || t.st1) && (K - 2) >= t.ks && (K - 2) < t.ke) st2(emid, o0 - 2u * (unsigned)g.sz, e2, t.st0, t.st1);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// kernel B:  colour sweep 3 ; colour sweep 4 ; increment!(ω)  [NORMS: L₁, L∞ of r' per workgroup; EPS: store the final ϵ]
// ------------------------------------------------------------------------------------------------------------------
template <int NORMS, int EPS>
__global__ void __launch_bounds__(PT_N, 8) k_gsrb2_B(GridX g, float* __restrict__ eout, float* __restrict__ rout, float* __restrict__ x, const float* __restrict__ emid,
                                                     const float* __restrict__ r, float w, int zchunk, double* __restrict__ part, float* __restrict__ pmax, wl::ConstL cl) {
  __shared__ float sA[2][PL_SZ];   // ϵ_mid of the newest plane
  __shared__ float sB[2][PL_SZ];   // plane K-1 after sweep 3
  __shared__ float sC[2][PL_SZ];   // plane K-2 after sweep 4 = final ϵ
  const PTile t = ptile<4, 3>(g, zchunk);
  if (!t.alive) { if (NORMS && threadIdx.x == 0) { part[blockIdx.x] = 0.0; pmax[blockIdx.x] = 0.f; } return; }
  for (int q = threadIdx.x; q < PL_SZ; q += PT_N) { sA[0][q] = 0.f; sA[1][q] = 0.f; sB[0][q] = 0.f; sB[1][q] = 0.f; sC[0][q] = 0.f; sC[1][q] = 0.f; }
  const PCoef k = pcoef(g, t, cl);
  const float c2 = cl.c[2];
  const int jpar = (t.j + g.gk) & 1;
  const int oth0 = PL_H + t.lq - 1, oth1 = t.lq + 1;
  float2 e0 = {0.f, 0.f}, e1 = e0, e2 = e0, e3 = e0, e4 = e0, r1 = e0, r2 = e0, r3 = e0, n_e0, n_r1, n_x3;
  double nsum = 0.0; float nmax = 0.f;
  const int Kbeg = t.ks - 3, Kend = t.ke + 2;
  const bool stp = t.st0 || t.st1;
  auto fetch = [&](int K) {
    const unsigned o0 = t.oc + (unsigned)K * (unsigned)g.sz;
    n_e0 = (t.indom && K >= 0 && K <= g.nz - 1) ? ld2(emid, o0) : make_float2(0.f, 0.f);
    n_r1 = (t.indom && pint(g, K - 1)) ? ld2(r, o0 - (unsigned)g.sz) : make_float2(0.f, 0.f);
    n_x3 = (stp && (K - 3) >= t.ks && (K - 3) < t.ke) ? ld2(x, o0 - 3u * (unsigned)g.sz) : make_float2(0.f, 0.f);
  };
  fetch(Kbeg);
  for (int K = Kbeg; K <= Kend; K++) {
    e4 = e3; e3 = e2; e2 = e1; e1 = e0; r3 = r2; r2 = r1;
    e0 = n_e0; r1 = n_r1;
    const float2 x3 = n_x3;
    if (K < Kend) fetch(K + 1);
    const float lz0 = cf(g.gk + K + 1, g.gnz, c2), lz1 = cf(g.gk + K, g.gnz, c2), lz2 = cf(g.gk + K - 1, g.gnz, c2), lz3 = cf(g.gk + K - 2, g.gnz, c2);
    __syncthreads();
    const int pb = (K - 1) & 1, cb = K & 1;
    const bool act1 = ((jpar ^ K) & 1) != 0;
    const bool gate = act1 ? t.in1 : t.in0;
    const int off_oth = act1 ? oth1 : oth0, off_y = act1 ? t.lq + PL_H : t.lq;
    const float lxo = act1 ? k.cxc : k.cxa;
    // ---- sweep 3 on plane K-1
    if (pint(g, K - 1) && (K - 1) >= t.ks - 2 && q4ok(g, g.gk + K - 1)) {
      const bool edge = (lz1 == 0.f) || (lz0 == 0.f);
      const float2 d = make_float2(edge ? k.ide0 : k.idm0, edge ? k.ide1 : k.idm1);
      pair_sweep(sA[pb], off_oth, off_y, act1, gate, lxo, k, lz1, lz0, r1, d, e2, e0, e1);
    }
    // ---- sweep 4 on plane K-2
    if (pint(g, K - 2) && (K - 2) >= t.ks - 1 && q4ok(g, g.gk + K - 2)) {
      const bool edge = (lz2 == 0.f) || (lz1 == 0.f);
      const float2 d = make_float2(edge ? k.ide0 : k.idm0, edge ? k.ide1 : k.idm1);
      pair_sweep(sB[pb], off_oth, off_y, act1, gate, lxo, k, lz2, lz1, r2, d, e3, e1, e2);
    }
    // ---- increment! on plane K-3: r' = r − ω·Aϵ ; x += ω·ϵ          src/Poisson.jl:100-104, mult :70-76
    if (stp && (K - 3) >= t.ks && (K - 3) < t.ke) {
      const float* __restrict__ S = sC[pb];
      const float zs = lz3 + lz2;
      float s = e3.x * (k.dxy0 - zs);
      s += (S[oth0] * k.cxa + e3.y * k.cxb);
      s += (S[t.lq - PL_W] * k.ky + S[t.lq + PL_W] * k.kyp);
      s += (e4.x * lz3 + e2.x * lz2);
      const float rn0 = r3.x - w * s;
      s = e3.y * (k.dxy1 - zs);
      s += (e3.x * k.cxb + S[oth1] * k.cxc);
      s += (S[PL_H + t.lq - PL_W] * k.ky + S[PL_H + t.lq + PL_W] * k.kyp);
      s += (e4.y * lz3 + e2.y * lz2);
      const float rn1 = r3.y - w * s;
      const unsigned o3 = t.oc + (unsigned)(K - 3) * (unsigned)g.sz;
      st2(rout, o3, make_float2(rn0, rn1), t.st0, t.st1);
      st2(x, o3, make_float2(x3.x + w * e3.x, x3.y + w * e3.y), t.st0, t.st1);
      if (EPS) st2(eout, o3, e3, t.st0, t.st1);
      if (NORMS) {
        const float a0 = t.st0 ? fabsf(rn0) : 0.f, a1 = t.st1 ? fabsf(rn1) : 0.f;
        nsum += (double)a0; nsum += (double)a1; nmax = fmaxf(nmax, fmaxf(a0, a1));
      }
    }
    sA[cb][t.lq] = e0.x; sA[cb][t.lq + PL_H] = e0.y;
    sB[cb][t.lq] = e1.x; sB[cb][t.lq + PL_H] = e1.y;
    sC[cb][t.lq] = e2.x; sC[cb][t.lq + PL_H] = e2.y;
  }
  if (NORMS) {   // 16 waves -> one partial per workgroup
    __shared__ double shs[PT_N / 64]; __shared__ float shm[PT_N / 64];
    nsum = wave_sum(nsum); nmax = wave_max(nmax);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { shs[threadIdx.x >> 6] = nsum; shm[threadIdx.x >> 6] = nmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
      double a = 0.0; float mx = 0.f;
      for (int q = 0; q < PT_N / 64; q++) { a += shs[q]; mx = fmaxf(mx, shm[q]); }
      part[blockIdx.x] = a; pmax[blockIdx.x] = mx;
    }
  }
}

int g_pair_on = 1;
int g_pro_fast = 1;   // bit 1 of gsrb_pair_enable: register-window prolongation when every direction is coarsened
int zchunk2(const GridX& g, int HX, int HY) {
  const int nt = ptile_count(g.nx, g.ny, HX, HY);
  const int np = g.k1 - g.k0;
  static const int target = getenv("WL_PAIR_WGS") ? atoi(getenv("WL_PAIR_WGS")) : 0;
  static const int zmin_env = getenv("WL_ZC_MIN") ? atoi(getenv("WL_ZC_MIN")) : 0;
  if (target) {   // experiments: fixed workgroup target
    int chunks = target >= 0 ? (target + nt - 1) / nt : (-target) / nt; if (chunks < 1) chunks = 1;
    int zc = (np + chunks - 1) / chunks; const int zmin = zmin_env ? zmin_env : 4; if (zc < zmin) zc = zmin; if (zc > np) zc = np;
    return zc;
  }
  // Few rounds: equal workgroups run in rounds of 512 (256 CUs × 2 resident workgroups of 1024 threads): choose the chunk length that minimises
  //   rounds × (planes per chunk + pipeline warm-up + start-up),
  // i.e. trade the warm-up recomputation of short chunks against the idle tail of a partly filled last round.  Grids that cannot fill
  // one round are bound by the per-plane latency of the march: shortest chunks (>= 4 planes).
  if ((long)nt * ((np + 31) / 32) >= 2048) {   // many rounds: bandwidth-bound and self-balancing — ≈3072 workgroups measured best at 512³
    const int chunks = (3072 + nt - 1) / nt;
    int zc = (np + chunks - 1) / chunks; if (zc < 16) zc = 16; if (zc > np) zc = np;
    return zc;
  }
  const int warm = (HY == 3 ? 5 : 3) + 3;
  long best = -1; int best_zc = np;
  for (int chunks = 1; chunks <= np; chunks++) {
    const int zc = (np + chunks - 1) / chunks;
    if (zc < (zmin_env ? zmin_env : 4)) break;
    const int nch = (np + zc - 1) / zc;
    const long W = (long)nt * nch;
    const long rounds = (W + 511) / 512;
    long cost = rounds * (zc + warm);
    if (W < 512) cost = (long)((zc + warm) * 1.25);          // a partly filled chip still pays the full per-plane latency (and unbalanced CUs)
    if (best < 0 || cost < best) { best = cost; best_zc = zc; }
  }
  return best_zc;
}
}  // namespace

namespace wl {
void gsrb_pair_enable(int on) { g_pair_on = on & 1; g_pro_fast = (on & 2) == 0; }
// geometry: even nx (float2), tiles not mostly empty, 32-bit offsets; a z-slab needs 3 ghost planes per side (kernel B's halo)
bool gsrb_pair_geom_ok(const GridX& g) {
  return g_pair_on && g.D == 3 && (g.nx & 1) == 0 && g.nx >= 66 && g.ny >= 34 && g.gnz >= 10 && (g.k1 - g.k0) >= 8 && (g.nz == g.gnz || g.k0 >= 3) && g.cs < (1L << 30);
}
bool gsrb_pair_ok(const GridX& g, const ConstL& cl) { return cl.on && gsrb_pair_geom_ok(g); }
int gsrb_pair_A(float* emid, const float* r, const GridX& g, const ConstL& cl, hipStream_t s) {
  const int zc = zchunk2(g, 2, 2);
  const int nt = ptile_count(g.nx, g.ny, 2, 2), per = (nt + 7) >> 3, nch = (g.k1 - g.k0 + zc - 1) / zc;
  ProArgs2 pa{};
  hipLaunchKernelGGL((k_gsrb2_A<0>), dim3((unsigned)(8 * per * nch)), dim3(PT_N), 0, s, g, emid, r, zc, pa, cl);
  WL_LAUNCH_CHECK(); return 0;
}
int gsrb_pair_A_pro(float* emid, float* rnew, float* x, const float* r, const float* xc, const GridX& g, const GridX& gc, float w, const ConstL& cl, hipStream_t s) {
  const int zc = zchunk2(g, 2, 2);
  const int nt = ptile_count(g.nx, g.ny, 2, 2), per = (nt + 7) >> 3, nch = (g.k1 - g.k0 + zc - 1) / zc;
  ProArgs2 pa{xc, x, rnew, gc, gc.nx < g.nx, gc.ny < g.ny, gc.gnz < g.gnz, w};
  const bool fullc = pa.cx && pa.cy && pa.cz && gc.gk == 0 && gc.nz == gc.gnz && g.gk == 0 && g.nz == g.gnz && 2 * (gc.nx - 2) == g.nx - 2 && 2 * (gc.ny - 2) == g.ny - 2 && 2 * (gc.nz - 2) == g.nz - 2;
  if (fullc && g_pro_fast) hipLaunchKernelGGL((k_gsrb2_A<2>), dim3((unsigned)(8 * per * nch)), dim3(PT_N), 0, s, g, emid, r, zc, pa, cl);
  else hipLaunchKernelGGL((k_gsrb2_A<1>), dim3((unsigned)(8 * per * nch)), dim3(PT_N), 0, s, g, emid, r, zc, pa, cl);
  WL_LAUNCH_CHECK(); return 0;
}
int gsrb_pair_B(float* eps, float* rout, float* x, const float* emid, const float* r, const GridX& g, float w,
                const RedWs* ws, int slot_d, int slot_f, const ConstL& cl, hipStream_t s) {
  const int zc = zchunk2(g, 4, 3);
  const int nt = ptile_count(g.nx, g.ny, 4, 3), per = (nt + 7) >> 3, nch = (g.k1 - g.k0 + zc - 1) / zc;
  const unsigned nb = (unsigned)(8 * per * nch);
  const bool norms = ws && nb <= WL_MAXPART;
  double* pa = norms ? ws->pa : nullptr; float* pm = norms ? ws->pm : nullptr;
#define WL_GB(NF, EF) hipLaunchKernelGGL((k_gsrb2_B<NF, EF>), dim3(nb), dim3(PT_N), 0, s, g, eps, rout, x, emid, r, w, zc, pa, pm, cl)
  if (norms) { if (eps) WL_GB(1, 1); else WL_GB(1, 0); } else { if (eps) WL_GB(0, 1); else WL_GB(0, 0); }
#undef WL_GB
  if (norms) WL_TRY(finalize_sum_max(*ws, (int)nb, slot_d, slot_f, s));
  else if (ws) WL_TRY(norms_dev(rout, g, *ws, slot_d, slot_f, s));
  WL_LAUNCH_CHECK(); return 0;
}
}  // namespace wl
