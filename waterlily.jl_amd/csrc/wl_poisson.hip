// Poisson / multigrid leaf kernels for gfx950.  Reference semantics: /root/reference/src/Poisson.jl,
// src/MultiLevelPoisson.jl (file:line per kernel).  All kernels: one thread per cell, linear over an
// x-y plane (coalesced, ghosts masked), pz walks z planes.  Arithmetic order follows the
// reference statement by statement (compiled with -ffp-contract=off) so that element-wise results
// are bit-identical to the CPU restatement; only reductions differ in association order.
#include <vector>

#include "wl_common.hpp"
#include "wl_bcfold.hpp"

namespace {

__device__ __forceinline__ bool cell_ij(const GridX& g, long m, int& i, int& j) {
  if (m >= g.sz) return false;
  j = (int)(m / g.nx);
  i = (int)(m - (long)j * g.nx);
  return true;
}
__device__ __forceinline__ bool interior_ij(const GridX& g, int i, int j) { return i >= 1 && i <= g.nx - 2 && j >= 1 && j <= g.ny - 2; }

// mult(I,L,D,x)   src/Poisson.jl:70-76
template <int D>
__device__ __forceinline__ float Ax(const GridX& g, long o, const float* __restrict__ L, const float* __restrict__ Dg, const float* __restrict__ x) {
  float s = x[o] * Dg[o];
  s += (x[o - 1] * L[o] + x[o + 1] * L[o + 1]);
  s += (x[o - g.sy] * L[g.cs + o] + x[o + g.sy] * L[g.cs + o + g.sy]);
  if (D == 3) s += (x[o - g.sz] * L[2 * g.cs + o] + x[o + g.sz] * L[2 * g.cs + o + g.sz]);
  return s;
}

// set_diag!   src/Poisson.jl:43-55
template <int D>
__global__ void k_set_diag(GridX g, float* __restrict__ Dg, float* __restrict__ iD, const float* __restrict__ L) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  if (!cell_ij(g, m, i, j) || !interior_ij(g, i, j)) return;
  const long o = m + (long)(g.k0 + pz) * g.sz;
  float s = 0.f;
  s -= (L[o] + L[o + 1]);
  s -= (L[g.cs + o] + L[g.cs + o + g.sy]);
  if (D == 3) s -= (L[2 * g.cs + o] + L[2 * g.cs + o + g.sz]);
  Dg[o] = s;
  iD[o] = (s == 0.f) ? s : 1.0f / s;
}

// mult!   src/Poisson.jl:63-69 (interior only; caller zero-fills z first)
template <int D>
__global__ void k_mult(GridX g, float* __restrict__ z, const float* __restrict__ L, const float* __restrict__ Dg, const float* __restrict__ x) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  if (!cell_ij(g, m, i, j) || !interior_ij(g, i, j)) return;
  const long o = m + (long)(g.k0 + pz) * g.sz;
  z[o] = Ax<D>(g, o, L, Dg, x);
}

// residual!  r = iD==0 ? 0 : z - A x, with block partial sums of r   src/Poisson.jl:92-95
template <int D>
__global__ void k_residual(GridX g, float* __restrict__ r, const float* __restrict__ x, const float* __restrict__ z, const float* __restrict__ L,
                           const float* __restrict__ Dg, const float* __restrict__ iD, double* __restrict__ part) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  double acc = 0.0;
  const int nsl = wl_nslots(g);
  if (cell_ij(g, m, i, j) && interior_ij(g, i, j)) {
    for (int k = g.k0 + pz; k < g.k1; k += nsl) {
      const long o = m + (long)k * g.sz;
      const float v = (iD[o] == 0.f) ? 0.f : z[o] - Ax<D>(g, o, L, Dg, x);
      r[o] = v;
      acc += (double)v;
    }
  }
  acc = block_sum(acc);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
// mom_project! head (src/Flow.jl:225) + residual! (src/Poisson.jl:92-95) in ONE pass:
//   z = div(u) ; x_out = x·dt (ALL cells) ; r = iD==0 ? 0 : z − A·(x·dt), with block partial sums of r.
// x_out ≠ x (neighbours still read the unscaled x; x·dt of a neighbour is recomputed — the same product bit for bit).
template <int D, int CL>
__global__ void k_div_residual(GridX g, float* __restrict__ z, float* __restrict__ xout, float* __restrict__ r, const float* __restrict__ x, const float* __restrict__ u,
                               const float* __restrict__ L, const float* __restrict__ Dg, const float* __restrict__ iD, float dt, double* __restrict__ part, wl::ConstL cl, int zchunk,
                               int p0, int p1) {      // the launch covers the local planes [p0,p1) (all of them: 0, g.nz)
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  double acc = 0.0;
  if (cell_ij(g, m, i, j)) {
    const bool inij = interior_ij(g, i, j);
    // this slot marches over planes [ks,ke): x[k-1], x[k], x[k+1] and u_z[k], u_z[k+1] live in registers
    const int ks = p0 + pz * zchunk, ke = (ks + zchunk < p1) ? ks + zchunk : p1;
    long o = m + (long)ks * g.sz;
    float xkm = (D == 3 && ks > 0) ? x[o - g.sz] : 0.f, xk = (ks < ke) ? x[o] : 0.f;
    float uzk = (D == 3 && inij && ks < ke) ? u[2 * g.cs + o] : 0.f;
    float lx = 0.f, lxp = 0.f, ly = 0.f, lyp = 0.f;
    if (CL) {
      lx = wl::wl_cl_coef(i + 1, g.nx, cl.c[0]); lxp = wl::wl_cl_coef(i + 2, g.nx, cl.c[0]);
      ly = wl::wl_cl_coef(j + 1, g.ny, cl.c[1]); lyp = wl::wl_cl_coef(j + 2, g.ny, cl.c[1]);
    }
    for (int k = ks; k < ke; k++, o += g.sz) {
      const bool up = D == 3 && k + 1 < g.nz;
      const float xkp = up ? x[o + g.sz] : 0.f;
      const float uzkp = (up && inij) ? u[2 * g.cs + o + g.sz] : 0.f;
      const float xs = xk * dt;
      xout[o] = xs;
      bool in = inij;
      if (D == 3) in = in && k >= g.k0 && k < g.k1;
      if (in) {
        float dv = 0.f;
        dv += u[o + 1] - u[o];
        dv += u[g.cs + o + g.sy] - u[g.cs + o];
        if (D == 3) dv += uzkp - uzk;
        if (z) z[o] = dv;            // b.z itself is optional: solver! reads it only through this residual
        // D (and iD==0 ⇔ D==0) recomputed from the face coefficients the stencil needs anyway: same operation order as
        // set_diag! (src/Poisson.jl:43-55), same bits as the stored arrays, 8 B/cell less traffic
        float lz = 0.f, lzp = 0.f;
        if (CL) {
          if (D == 3) { lz = wl::wl_cl_coef(g.gk + k + 1, g.gnz, cl.c[2]); lzp = wl::wl_cl_coef(g.gk + k + 2, g.gnz, cl.c[2]); }
        } else {
          lx = L[o]; lxp = L[o + 1]; ly = L[g.cs + o]; lyp = L[g.cs + o + g.sy];
          if (D == 3) { lz = L[2 * g.cs + o]; lzp = L[2 * g.cs + o + g.sz]; }
        }
        float dgv = 0.f;
        dgv -= (lx + lxp);
        dgv -= (ly + lyp);
        if (D == 3) dgv -= (lz + lzp);
        float s = xs * dgv;
        s += ((x[o - 1] * dt) * lx + (x[o + 1] * dt) * lxp);
        s += ((x[o - g.sy] * dt) * ly + (x[o + g.sy] * dt) * lyp);
        if (D == 3) s += ((xkm * dt) * lz + (xkp * dt) * lzp);
        const float v = (dgv == 0.f) ? 0.f : dv - s;
        r[o] = v;
        acc += (double)v;
      }
      xkm = xk; xk = xkp; uzk = uzkp;
    }
  }
  acc = block_sum(acc);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
// float -> int that orders like the float (−0 < +0 aside): maxima of many blocks through integer atomicMax into a few slots
#define WL_ENC_SLOTS 1024
__device__ __forceinline__ int wl_enc_f(float v) { const int b = __float_as_int(v); return b >= 0 ? b : (b ^ 0x7FFFFFFF); }
__device__ __forceinline__ float wl_dec_f(int k) { return __int_as_float(k >= 0 ? k : (k ^ 0x7FFFFFFF)); }
// mom_project! tail (src/Flow.jl:227-230): u[I,i] -= L[I,i]·∂ᵢx ; p_out = x/dt (ALL cells), p_out ≠ x
template <int D, int CL>
__global__ void k_project_unscale(GridX g, float* __restrict__ u, const float* __restrict__ L, const float* __restrict__ x, float* __restrict__ pout, float dt, wl::ConstL cl, int zchunk,
                                  int p0, int p1, BcFold bc, int lin) {   // local planes [p0,p1) of this launch; bc.on: BC!(u,U) folded into the stores (wl_bcfold.hpp); lin: linear block order (wl_tile_lin)
  if (bc.go && *bc.go == 0.f) return;      // queued before the host knew whether the solve had converged: it had not
  int i, j; long m; int pz;
  if (lin) wl_tile_lin(g, m, pz); else wl_tile(g, m, pz);
  if (!cell_ij(g, m, i, j)) return;
  const bool inij = interior_ij(g, i, j);
  const int ks = p0 + pz * zchunk, ke = (ks + zchunk < p1) ? ks + zchunk : p1;
  long o = m + (long)ks * g.sz;
  float xkm = (D == 3 && ks > 0 && ks < ke) ? x[o - g.sz] : 0.f;      // x[k-1] stays in a register while marching
  const float lxc = CL ? wl::wl_cl_coef(i + 1, g.nx, cl.c[0]) : 0.f, lyc = CL ? wl::wl_cl_coef(j + 1, g.ny, cl.c[1]) : 0.f;
  for (int k = ks; k < ke; k++, o += g.sz) {
    const float xc = x[o];
    pout[o] = xc / dt;
    bool in = inij;
    if (D == 3) in = in && k >= g.k0 && k < g.k1;
    if (in) {
      const float lx = CL ? lxc : L[o], ly = CL ? lyc : L[g.cs + o];
      if (D == 3 && bc.on) {
        const float lz = CL ? wl::wl_cl_coef(g.gk + k + 1, g.gnz, cl.c[2]) : L[2 * g.cs + o];
        const float v[3] = {u[o] - lx * (xc - x[o - 1]), u[g.cs + o] - ly * (xc - x[o - g.sy]), u[2 * g.cs + o] - lz * (xc - xkm)};
        if (wl_bc_fold_plain(g, i, j, k)) { u[o] = v[0]; u[g.cs + o] = v[1]; u[2 * g.cs + o] = v[2]; }
        else wl_bc_fold_store(u, g, i, j, k, v, bc.U);
      } else {
      u[o] -= lx * (xc - x[o - 1]);
      u[g.cs + o] -= ly * (xc - x[o - g.sy]);
      if (D == 3) { const float lz = CL ? wl::wl_cl_coef(g.gk + k + 1, g.gnz, cl.c[2]) : L[2 * g.cs + o]; u[2 * g.cs + o] -= lz * (xc - xkm); }
      }
    }
    xkm = xc;
  }
}
// The corrector's mom_project! tail with CFL's flux_out folded in (src/Flow.jl:227-230 + :234-244): out-of-place u (u_out ≠ u_in, so
// the +δ neighbours' projected values can be recomputed from u_in — the same statements their owners execute, hence the same
// bits), σ = flux_out of the projected field, per-workgroup max over the planes [kfirst,klast) of σ (ghost cells: stale Φ, Q1).
// Valid when BC! does not change what flux_out reads: wall-normal boundary faces already hold U and L is 0 there
// (non-periodic, no exitBC).  Cells outside the interior of u_out are left for BC! to write.
template <int D, int CL>
__global__ void k_project_cfl(GridX g, float* __restrict__ uout, const float* __restrict__ uin, const float* __restrict__ L, const float* __restrict__ x, float* __restrict__ pout,
                              float* __restrict__ sigma, float dt, wl::ConstL cl, int zchunk, int kfirst, int klast, float* __restrict__ pmax, int p0, int p1, int store_sigma, BcFold bc, int lin) {
  int i, j; long m; int pz;
  if (lin) wl_tile_lin(g, m, pz); else wl_tile(g, m, pz);      // lin: linear block order, one plane per block; the maxima go to WL_ENC_SLOTS order-encoded integers
  float mx = -INFINITY;
  if (cell_ij(g, m, i, j)) {
    const bool inij = interior_ij(g, i, j);
    const int ks = p0 + pz * zchunk, ke = (ks + zchunk < p1) ? ks + zchunk : p1;
    long o = m + (long)ks * g.sz;
    float xkm = (D == 3 && ks > 0 && ks < ke) ? x[o - g.sz] : 0.f, xc = (ks < ke) ? x[o] : 0.f;
    float lx = 0.f, lxp = 0.f, ly = 0.f, lyp = 0.f;
    if (CL) {
      lx = wl::wl_cl_coef(i + 1, g.nx, cl.c[0]); lxp = wl::wl_cl_coef(i + 2, g.nx, cl.c[0]);
      ly = wl::wl_cl_coef(j + 1, g.ny, cl.c[1]); lyp = wl::wl_cl_coef(j + 2, g.ny, cl.c[1]);
    }
    for (int k = ks; k < ke; k++, o += g.sz) {
      const float xkp = (D == 3 && k + 1 < g.nz) ? x[o + g.sz] : 0.f;
      pout[o] = xc / dt;
      bool in = inij;
      if (D == 3) in = in && k >= g.k0 && k < g.k1;
      float sg;
      if (in) {
        float lz = 0.f, lzp = 0.f;
        if (CL) { if (D == 3) { lz = wl::wl_cl_coef(g.gk + k + 1, g.gnz, cl.c[2]); lzp = wl::wl_cl_coef(g.gk + k + 2, g.gnz, cl.c[2]); } }
        else {
          lx = L[o]; lxp = L[o + 1]; ly = L[g.cs + o]; lyp = L[g.cs + o + g.sy];
          if (D == 3) { lz = L[2 * g.cs + o]; lzp = L[2 * g.cs + o + g.sz]; }
        }
        const float uxn = uin[o] - lx * (xc - x[o - 1]), uxp = uin[o + 1] - lxp * (x[o + 1] - xc);
        const float uyn = uin[g.cs + o] - ly * (xc - x[o - g.sy]), uyp = uin[g.cs + o + g.sy] - lyp * (x[o + g.sy] - xc);
        const bool foldc = D == 3 && bc.on && !wl_bc_fold_plain(g, i, j, k);      // BC!(u_out,U) folded into the stores of the cells near the boundary
        if (!foldc) { uout[o] = uxn; uout[g.cs + o] = uyn; }
        sg = 0.f;
        sg += (fmaxf(0.f, uxp) + fmaxf(0.f, -uxn));
        sg += (fmaxf(0.f, uyp) + fmaxf(0.f, -uyn));
        if (D == 3) {
          const float uzn = uin[2 * g.cs + o] - lz * (xc - xkm), uzp = uin[2 * g.cs + o + g.sz] - lzp * (xkp - xc);
          if (!foldc) uout[2 * g.cs + o] = uzn;
          else { const float v[3] = {uxn, uyn, uzn}; wl_bc_fold_store(uout, g, i, j, k, v, bc.U); }
          sg += (fmaxf(0.f, uzp) + fmaxf(0.f, -uzn));
        }
        if (store_sigma) sigma[o] = sg;       // σ = flux_out is only read by the maximum taken here: materialised on request
      } else sg = sigma[o];
      if (k >= kfirst && k < klast) mx = fmaxf(mx, sg);
      xkm = xc; xc = xkp;
    }
  }
  mx = block_max(mx);
#ifdef WL_CFL_NOATOM   // timing experiment (wrong Δt): what the atomics cost
  if (threadIdx.x == 0) { if (lin) reinterpret_cast<int*>(pmax)[blockIdx.x & (WL_ENC_SLOTS - 1)] = wl_enc_f(mx); else pmax[blockIdx.x] = mx; }
#else
  if (threadIdx.x == 0) { if (lin) atomicMax(reinterpret_cast<int*>(pmax) + (blockIdx.x & (WL_ENC_SLOTS - 1)), wl_enc_f(mx)); else pmax[blockIdx.x] = mx; }
#endif
}
__global__ void k_enc_init(int* __restrict__ p) { p[threadIdx.x] = (int)0x80000000; }
// ---- the projection tails with TWO x-adjacent cells per thread, one plane per block, linear block order (3-D, constant coefficients, even nx) -----------
// Same statements per cell as k_project_unscale / k_project_cfl.  Why a second form: in linear order (the order HBM serves best, wl_tile_lin) the one-cell
// kernels are bound by L1 requests — 7 (unscale) / 13 (cfl) dword loads per cell; as float2 pairs the same cells need about half of them.
// Block h: chunk (h mod nb8) of 512 consecutive floats of plane (h div nb8); a pair never straddles a row (nx even, pairs start at even columns).
__device__ __forceinline__ float2 pl_ld2(const float* __restrict__ p, long o) { return *reinterpret_cast<const float2*>(p + o); }
__device__ __forceinline__ void pl_st2(float* __restrict__ p, long o, float2 v) { *reinterpret_cast<float2*>(p + o) = v; }
__device__ __forceinline__ bool pl_pair(const GridX& g, long& m, int& k, int p0) {
  const long np2 = (g.sz / 2 + WL_BLOCK - 1) / WL_BLOCK;          // 256-pair chunks per plane
  const unsigned nb8 = (unsigned)(((np2 + 7) >> 3) << 3);
  const unsigned h = blockIdx.x;
  const unsigned p = h / nb8;
  k = p0 + (int)p;
  m = 2 * ((long)(h - p * nb8) * WL_BLOCK + threadIdx.x);
  return m < g.sz;
}
static inline unsigned pl_grid(const GridX& g, int nplanes) { const long np2 = (g.sz / 2 + WL_BLOCK - 1) / WL_BLOCK; return (unsigned)((((np2 + 7) >> 3) << 3) * nplanes); }
__global__ void __launch_bounds__(WL_BLOCK) k_project_unscale2(GridX g, float* __restrict__ u, const float* __restrict__ x, float* __restrict__ pout, float dt, wl::ConstL cl,
                                                                  int p0, int p1, BcFold bc) {
  if (bc.go && *bc.go == 0.f) return;
  long m; int k;
  if (!pl_pair(g, m, k, p0) || k >= p1) return;
  const int j = (int)(m / g.nx), i0 = (int)(m - (long)j * g.nx);
  const long o = m + (long)k * g.sz;
  const float2 xc = pl_ld2(x, o);
  pl_st2(pout, o, make_float2(xc.x / dt, xc.y / dt));
  if (!(j >= 1 && j <= g.ny - 2 && k >= g.k0 && k < g.k1)) return;
  const bool in0 = i0 >= 1, in1 = i0 + 1 <= g.nx - 2;                 // (i0 is even: i0 <= nx-2 always; cell 0 is a ghost only at i0 == 0)
  const float xl = in0 ? x[o - 1] : 0.f;
  const float2 xy = pl_ld2(x, o - g.sy), xz = pl_ld2(x, o - g.sz);
  const float2 u0 = pl_ld2(u, o), u1 = pl_ld2(u, g.cs + o), u2 = pl_ld2(u, 2 * g.cs + o);
  const float lx0 = wl::wl_cl_coef(i0 + 1, g.nx, cl.c[0]), lx1 = wl::wl_cl_coef(i0 + 2, g.nx, cl.c[0]);
  const float ly = wl::wl_cl_coef(j + 1, g.ny, cl.c[1]), lz = wl::wl_cl_coef(g.gk + k + 1, g.gnz, cl.c[2]);
  const float va[3] = {u0.x - lx0 * (xc.x - xl), u1.x - ly * (xc.x - xy.x), u2.x - lz * (xc.x - xz.x)};
  const float vb[3] = {u0.y - lx1 * (xc.y - xc.x), u1.y - ly * (xc.y - xy.y), u2.y - lz * (xc.y - xz.y)};
  if (bc.on && !(wl_bc_fold_plain(g, i0, j, k) && wl_bc_fold_plain(g, i0 + 1, j, k))) {
    if (in0) { if (wl_bc_fold_plain(g, i0, j, k)) { u[o] = va[0]; u[g.cs + o] = va[1]; u[2 * g.cs + o] = va[2]; } else wl_bc_fold_store(u, g, i0, j, k, va, bc.U); }
    if (in1) { if (wl_bc_fold_plain(g, i0 + 1, j, k)) { u[o + 1] = vb[0]; u[g.cs + o + 1] = vb[1]; u[2 * g.cs + o + 1] = vb[2]; } else wl_bc_fold_store(u, g, i0 + 1, j, k, vb, bc.U); }
    return;
  }
  if (in0 && in1) { pl_st2(u, o, make_float2(va[0], vb[0])); pl_st2(u, g.cs + o, make_float2(va[1], vb[1])); pl_st2(u, 2 * g.cs + o, make_float2(va[2], vb[2])); }
  else if (in0) { u[o] = va[0]; u[g.cs + o] = va[1]; u[2 * g.cs + o] = va[2]; }
  else if (in1) { u[o + 1] = vb[0]; u[g.cs + o + 1] = vb[1]; u[2 * g.cs + o + 1] = vb[2]; }
}
__global__ void __launch_bounds__(WL_BLOCK) k_project_cfl2(GridX g, float* __restrict__ uout, const float* __restrict__ uin, const float* __restrict__ x, float* __restrict__ pout,
                                                              float* __restrict__ sigma, float dt, wl::ConstL cl, int kfirst, int klast, float* __restrict__ pmax, int p0, int p1,
                                                              int store_sigma, BcFold bc) {
  if (bc.go && *bc.go == 0.f) return;      // (block-uniform; the maximum's slots keep k_enc_init's −∞)
  long m; int k;
  float mx = -INFINITY;
  if (pl_pair(g, m, k, p0) && k < p1) {
    const int j = (int)(m / g.nx), i0 = (int)(m - (long)j * g.nx);
    const long o = m + (long)k * g.sz;
    const float2 xc = pl_ld2(x, o);
    pl_st2(pout, o, make_float2(xc.x / dt, xc.y / dt));
    const bool row = j >= 1 && j <= g.ny - 2 && k >= g.k0 && k < g.k1;
    const bool in0 = row && i0 >= 1, in1 = row && i0 + 1 <= g.nx - 2;
    const bool want = k >= kfirst && k < klast;
    float sg0 = 0.f, sg1 = 0.f;
    if (in0 || in1) {
      const float lx0 = wl::wl_cl_coef(i0 + 1, g.nx, cl.c[0]), lx1 = wl::wl_cl_coef(i0 + 2, g.nx, cl.c[0]), lx2 = wl::wl_cl_coef(i0 + 3, g.nx, cl.c[0]);
      const float ly = wl::wl_cl_coef(j + 1, g.ny, cl.c[1]), lyp = wl::wl_cl_coef(j + 2, g.ny, cl.c[1]);
      const float lz = wl::wl_cl_coef(g.gk + k + 1, g.gnz, cl.c[2]), lzp = wl::wl_cl_coef(g.gk + k + 2, g.gnz, cl.c[2]);
      const float xl = in0 ? x[o - 1] : 0.f, xr = in1 ? x[o + 2] : 0.f;
      const float2 xym = pl_ld2(x, o - g.sy), xyp = pl_ld2(x, o + g.sy), xzm = pl_ld2(x, o - g.sz), xzp = pl_ld2(x, o + g.sz);
      float2 a0 = pl_ld2(uin, o), a1 = pl_ld2(uin, g.cs + o), a2 = pl_ld2(uin, 2 * g.cs + o);
      const float axr = in1 ? uin[o + 2] : 0.f;
      float2 a1p = pl_ld2(uin, g.cs + o + g.sy), a2p = pl_ld2(uin, 2 * g.cs + o + g.sz);
      if (bc.usub) {   // BC!(u_in,U) deferred: the values BC! would have put on the wall-normal faces this stencil reads (index 1 and N−1 along the component's direction)
        if (i0 == 0 || i0 == g.nx - 2) a0.y = bc.U[0];
        if (j == 1) a1 = make_float2(bc.U[1], bc.U[1]);
        if (j + 1 == g.ny - 1) a1p = make_float2(bc.U[1], bc.U[1]);
        if (g.gk + k == 1) a2 = make_float2(bc.U[2], bc.U[2]);
        if (g.gk + k + 1 == g.gnz - 1) a2p = make_float2(bc.U[2], bc.U[2]);
      }
      // cell 0 (i0): same statements as k_project_cfl
      const float uxn0 = a0.x - lx0 * (xc.x - xl), uxp0 = a0.y - lx1 * (xc.y - xc.x);
      const float uyn0 = a1.x - ly * (xc.x - xym.x), uyp0 = a1p.x - lyp * (xyp.x - xc.x);
      const float uzn0 = a2.x - lz * (xc.x - xzm.x), uzp0 = a2p.x - lzp * (xzp.x - xc.x);
      // cell 1 (i0+1)
      const float uxn1 = a0.y - lx1 * (xc.y - xc.x), uxp1 = axr - lx2 * (xr - xc.y);
      const float uyn1 = a1.y - ly * (xc.y - xym.y), uyp1 = a1p.y - lyp * (xyp.y - xc.y);
      const float uzn1 = a2.y - lz * (xc.y - xzm.y), uzp1 = a2p.y - lzp * (xzp.y - xc.y);
      if (in0) { sg0 = 0.f; sg0 += (fmaxf(0.f, uxp0) + fmaxf(0.f, -uxn0)); sg0 += (fmaxf(0.f, uyp0) + fmaxf(0.f, -uyn0)); sg0 += (fmaxf(0.f, uzp0) + fmaxf(0.f, -uzn0)); }
      if (in1) { sg1 = 0.f; sg1 += (fmaxf(0.f, uxp1) + fmaxf(0.f, -uxn1)); sg1 += (fmaxf(0.f, uyp1) + fmaxf(0.f, -uyn1)); sg1 += (fmaxf(0.f, uzp1) + fmaxf(0.f, -uzn1)); }
      const float va[3] = {uxn0, uyn0, uzn0}, vb[3] = {uxn1, uyn1, uzn1};
      const bool plain0 = !bc.on || wl_bc_fold_plain(g, i0, j, k), plain1 = !bc.on || wl_bc_fold_plain(g, i0 + 1, j, k);
      if (in0 && in1 && plain0 && plain1) { pl_st2(uout, o, make_float2(va[0], vb[0])); pl_st2(uout, g.cs + o, make_float2(va[1], vb[1])); pl_st2(uout, 2 * g.cs + o, make_float2(va[2], vb[2])); }
      else {
        if (in0) { if (plain0) { uout[o] = va[0]; uout[g.cs + o] = va[1]; uout[2 * g.cs + o] = va[2]; } else wl_bc_fold_store(uout, g, i0, j, k, va, bc.U); }
        if (in1) { if (plain1) { uout[o + 1] = vb[0]; uout[g.cs + o + 1] = vb[1]; uout[2 * g.cs + o + 1] = vb[2]; } else wl_bc_fold_store(uout, g, i0 + 1, j, k, vb, bc.U); }
      }
      if (store_sigma) { if (in0) sigma[o] = sg0; if (in1) sigma[o + 1] = sg1; }
    }
    if (want) {      // cells outside the interior: the stale Φ the reference leaves in σ's ghost cells takes part in maximum(σ)  (quirk Q1)
      if (!in0) sg0 = sigma[o];
      if (!in1) sg1 = sigma[o + 1];
      mx = fmaxf(sg0, sg1);
    }
  }
  mx = block_max(mx);
  if (threadIdx.x == 0) atomicMax(reinterpret_cast<int*>(pmax) + (blockIdx.x & (WL_ENC_SLOTS - 1)), wl_enc_f(mx));
}

__global__ void k_fin_max_enc(const int* __restrict__ p, float* __restrict__ om) {
  float mx = -INFINITY;
  for (int q = threadIdx.x; q < WL_ENC_SLOTS; q += WL_BLOCK) { const int k = p[q]; if (k != (int)0x80000000) mx = fmaxf(mx, wl_dec_f(k)); }
  mx = block_max(mx); if (threadIdx.x == 0) *om = mx;
}
__global__ void k_fin_max2(const float* __restrict__ pmax, int n, float* __restrict__ om) {
  float mx = -INFINITY, m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY;
  int q = threadIdx.x;
  for (; q + 3 * WL_BLOCK < n; q += 4 * WL_BLOCK) { mx = fmaxf(mx, pmax[q]); m1 = fmaxf(m1, pmax[q + WL_BLOCK]); m2 = fmaxf(m2, pmax[q + 2 * WL_BLOCK]); m3 = fmaxf(m3, pmax[q + 3 * WL_BLOCK]); }
  for (; q < n; q += WL_BLOCK) mx = fmaxf(mx, pmax[q]);
  mx = fmaxf(fmaxf(mx, m1), fmaxf(m2, m3));
  mx = block_max(mx); if (threadIdx.x == 0) *om = mx;
}
// exact test of the constant-coefficient pattern over EVERY cell of L (ghosts included): L[I,a] == (I_a ∈ {1,2,N_a} ? 0 : c_a)
template <int D>
__global__ void k_check_const_L(GridX g, const float* __restrict__ L, float c0, float c1, float c2, int* __restrict__ flag) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  if (!cell_ij(g, m, i, j)) return;
  const int k = pz;
  // slab ranks: planes outside the global array (beyond the physical ghost plane) hold nothing meaningful
  const int Kj = (D == 3) ? g.gk + k + 1 : 1;
  if (D == 3 && (Kj < 1 || Kj > g.gnz)) return;
  if (D == 3 && g.nz != g.gnz && (k < g.k0 - 1 || k > g.k1)) return;        // deeper ghost planes of a slab are never exchanged for L
  const long o = m + (long)k * g.sz;
  const float c[3] = {c0, c1, c2};
  const int I[3] = {i + 1, j + 1, Kj};
  const int N[3] = {g.nx, g.ny, (D == 3) ? g.gnz : 1};
  bool bad = false;
  for (int a = 0; a < D; a++) {
    const float want = (I[a] <= 2 || I[a] >= N[a]) ? 0.f : c[a];
    bad = bad || (L[(long)a * g.cs + o] != want);
  }
  if (bad) atomicOr(flag, 1);
}
// ---- pcg!(p;it)   src/Poisson.jl:166-186 — Jacobi-preconditioned conjugate gradients of the single-level Poisson ----
// stage 0: z = ϵ = r·iD                      (+ Σ r·z)          :168-169
// stage 1: z = mult(I,L,D,ϵ)                 (+ Σ z·ϵ)          :173-174
// stage 2: x += α·ϵ ; r -= α·z  [; z = r·iD  (+ Σ r·z)]         :176-180
// stage 3: ϵ = β·ϵ + z                                          :183
template <int D, int STAGE>
__global__ void k_pcg(GridX g, float* __restrict__ eps, float* __restrict__ r, float* __restrict__ x, float* __restrict__ z, const float* __restrict__ L,
                      const float* __restrict__ Dg, const float* __restrict__ iD, float a, int more, double* __restrict__ part) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  double acc = 0.0;
  const int nsl = wl_nslots(g);
  if (cell_ij(g, m, i, j) && interior_ij(g, i, j)) {
    for (int k = g.k0 + pz; k < g.k1; k += nsl) {
      const long o = m + (long)k * g.sz;
      if (STAGE == 0) { const float v = r[o] * iD[o]; z[o] = v; eps[o] = v; acc += (double)r[o] * (double)v; }
      if (STAGE == 1) { const float v = Ax<D>(g, o, L, Dg, eps); z[o] = v; acc += (double)v * (double)eps[o]; }
      if (STAGE == 2) {
        x[o] += a * eps[o];
        const float rn = r[o] - a * z[o];
        r[o] = rn;
        if (more) { const float v = rn * iD[o]; z[o] = v; acc += (double)rn * (double)v; }
      }
      if (STAGE == 3) eps[o] = a * eps[o] + z[o];
    }
  }
  if (STAGE != 3) { acc = block_sum(acc); if (threadIdx.x == 0) part[blockIdx.x] = acc; }
}
// per-plane version of the test above: bad[k] = 1 if plane k holds a cell whose coefficients deviate from the constant pattern
template <int D>
__global__ void k_plane_bad(GridX g, const float* __restrict__ L, float c0, float c1, float c2, unsigned char* __restrict__ bad) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  if (!cell_ij(g, m, i, j)) return;
  const int k = pz;
  const long o = m + (long)k * g.sz;
  const float c[3] = {c0, c1, c2};
  const int I[3] = {i + 1, j + 1, (D == 3) ? g.gk + k + 1 : 1};
  const int N[3] = {g.nx, g.ny, (D == 3) ? g.gnz : 1};
  for (int a = 0; a < D; a++) if (L[(long)a * g.cs + o] != ((I[a] <= 2 || I[a] >= N[a]) ? 0.f : c[a])) bad[k] = 1;
}
// deterministic second stage: res_d[slot] = Σ partials
// (single-workgroup final stages: four independent accumulators per thread keep four loads in flight — with up to ≈18 000 partials
//  of the marching kernels the dependent one-load-at-a-time loop took 18–27 µs)
__global__ void k_final_sum(const double* __restrict__ part, int n, double* __restrict__ out) {
  double a = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int q = threadIdx.x;
  for (; q + 3 * WL_BLOCK < n; q += 4 * WL_BLOCK) { a += part[q]; a1 += part[q + WL_BLOCK]; a2 += part[q + 2 * WL_BLOCK]; a3 += part[q + 3 * WL_BLOCK]; }
  for (; q < n; q += WL_BLOCK) a += part[q];
  a = (a + a1) + (a2 + a3);
  a = block_sum(a);
  if (threadIdx.x == 0) *out = a;
}
__global__ void k_final_sum_max(const double* __restrict__ part, const float* __restrict__ pmax, int n, double* __restrict__ out_s, float* __restrict__ out_m) {
  double a = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0; float mx = -INFINITY, m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY;
  int q = threadIdx.x;
  for (; q + 3 * WL_BLOCK < n; q += 4 * WL_BLOCK) {
    a += part[q]; a1 += part[q + WL_BLOCK]; a2 += part[q + 2 * WL_BLOCK]; a3 += part[q + 3 * WL_BLOCK];
    mx = fmaxf(mx, pmax[q]); m1 = fmaxf(m1, pmax[q + WL_BLOCK]); m2 = fmaxf(m2, pmax[q + 2 * WL_BLOCK]); m3 = fmaxf(m3, pmax[q + 3 * WL_BLOCK]);
  }
  for (; q < n; q += WL_BLOCK) { a += part[q]; mx = fmaxf(mx, pmax[q]); }
  a = (a + a1) + (a2 + a3); mx = fmaxf(fmaxf(mx, m1), fmaxf(m2, m3));
  a = block_sum(a);
  mx = block_max(mx);
  if (threadIdx.x == 0) { *out_s = a; *out_m = mx; }
}
__global__ void k_final_max(const float* __restrict__ pmax, int n, float* __restrict__ out_m) {
  float mx = -INFINITY;
  for (int q = threadIdx.x; q < n; q += WL_BLOCK) mx = fmaxf(mx, pmax[q]);
  mx = block_max(mx);
  if (threadIdx.x == 0) *out_m = mx;
}
// s = Σr/N ; if |s| > 2eps(Float32): r -= s    src/Poisson.jl:95-97   (predicate evaluated on device: no host sync)
__global__ void k_mean_shift(GridX g, float* __restrict__ r, const double* __restrict__ sum, double n_inside) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  // Julia: sum(p.r) is Float32 (pairwise); s = that / length(inside).  We round the double sum to Float32 first.
  const float s = (float)(*sum) / (float)n_inside;
  if (fabsf(s) <= 2.f * 1.1920929e-7f) return;
  if (!cell_ij(g, m, i, j) || !interior_ij(g, i, j)) return;
  const long o = m + (long)(g.k0 + pz) * g.sz;
  r[o] = r[o] - s;
}
// L₁ = Σ|r|, L∞ = max|r| over the interior (ghosts of r are identically zero)   src/Poisson.jl:190-191
__global__ void k_norms(GridX g, const float* __restrict__ r, double* __restrict__ part, float* __restrict__ pmax) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  double acc = 0.0; float mx = 0.f;
  const int nsl = wl_nslots(g);
  if (cell_ij(g, m, i, j) && interior_ij(g, i, j)) {
    for (int k = g.k0 + pz; k < g.k1; k += nsl) {
      const float v = fabsf(r[m + (long)k * g.sz]);
      acc += (double)v; mx = fmaxf(mx, v);
    }
  }
  acc = block_sum(acc);
  mx = block_max(mx);
  if (threadIdx.x == 0) { part[blockIdx.x] = acc; pmax[blockIdx.x] = mx; }
}

// increment!  r -= ω A ϵ ; x += ω ϵ      src/Poisson.jl:100-104
template <int D>
__global__ void k_increment(GridX g, float* __restrict__ r, float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ L,
                            const float* __restrict__ Dg, float w) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  if (!cell_ij(g, m, i, j) || !interior_ij(g, i, j)) return;
  const long o = m + (long)(g.k0 + pz) * g.sz;
  r[o] = r[o] - w * Ax<D>(g, o, L, Dg, eps);
  x[o] = x[o] + w * eps[o];
}

// ϵ = r·iD    src/Poisson.jl:112,142
__global__ void k_gs_init(GridX g, float* __restrict__ eps, const float* __restrict__ r, const float* __restrict__ iD) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  if (!cell_ij(g, m, i, j) || !interior_ij(g, i, j)) return;
  const long o = m + (long)(g.k0 + pz) * g.sz;
  eps[o] = r[o] * iD[o];
}

// one colour of red–black Gauss–Seidel: gauss_rb / half_rangek   src/Poisson.jl:116-132,145
// colour rule: sweep k₀ updates the cells whose Julia (1-based, GLOBAL) index sum has the
// parity of k₀+1, i.e. (i+j+K + D + k₀) ODD with 0-based global indices.  Pairs are formed along the LAST
// dimension, so when that extent (with ghosts) is odd the last interior layer is never visited (quirk Q4).
template <int D>
__global__ void k_gs_sweep(GridX g, float* __restrict__ eps, const float* __restrict__ r, const float* __restrict__ L, const float* __restrict__ iD, int kk0) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  if (!cell_ij(g, m, i, j) || !interior_ij(g, i, j)) return;
  const int k = g.k0 + pz;
  const int K = (D == 3) ? g.gk + k : 0;
  if (((i + j + K + D + kk0) & 1) == 0) return;
  // Q4: last dimension index (1-based) must be <= 2*(Ng÷2)-1
  if (D == 3) { if (K + 1 > 2 * (g.gnz / 2) - 1) return; } else { if (j + 1 > 2 * (g.ny / 2) - 1) return; }
  const long o = m + (long)k * g.sz;
  float s = r[o];
  s -= (eps[o - 1] * L[o] + eps[o + 1] * L[o + 1]);
  s -= (eps[o - g.sy] * L[g.cs + o] + eps[o + g.sy] * L[g.cs + o + g.sy]);
  if (D == 3) s -= (eps[o - g.sz] * L[2 * g.cs + o] + eps[o + g.sz] * L[2 * g.cs + o + g.sz]);
  eps[o] = s * iD[o];
}

// ---- fused variants used by the V-cycle on non-periodic, non-distributed levels -------------------------------
// ϵ at a neighbour before any sweep: r·iD for an interior cell (bit-identical to what gs_init stores), the stored
// ghost value otherwise (ghosts of ϵ are never written by the smoother).
template <int D>
__device__ __forceinline__ bool is_inside(const GridX& g, int i, int j, int k) {
  bool in = i >= 1 && i <= g.nx - 2 && j >= 1 && j <= g.ny - 2;
  if (D == 3) in = in && k >= g.k0 && k < g.k1;
  return in;
}
// GaussSeidelRB!: `ϵ = r·iD` and the first colour sweep in ONE pass (24 instead of 12+28 B/cell)   src/Poisson.jl:142-145
template <int D>
__global__ void k_gs_init_sweep1(GridX g, float* __restrict__ eps, const float* __restrict__ r, const float* __restrict__ L, const float* __restrict__ iD) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  if (!cell_ij(g, m, i, j) || !interior_ij(g, i, j)) return;
  const int k = g.k0 + pz;
  const int K = (D == 3) ? g.gk + k : 0;
  const long o = m + (long)k * g.sz;
  const float e0 = r[o] * iD[o];
  bool upd = ((i + j + K + D + 1) & 1) != 0;                                    // colour of sweep k₀=1
  if (D == 3) upd = upd && !(K + 1 > 2 * (g.gnz / 2) - 1); else upd = upd && !(j + 1 > 2 * (g.ny / 2) - 1);   // quirk Q4
  if (!upd) { eps[o] = e0; return; }
  auto E = [&](long oo) -> float { return r[oo] * iD[oo]; };      // ghosts: 0·0 = the stored ghost ϵ (handle-owned arrays)
  float s = r[o];
  s -= (E(o - 1) * L[o] + E(o + 1) * L[o + 1]);
  s -= (E(o - g.sy) * L[g.cs + o] + E(o + g.sy) * L[g.cs + o + g.sy]);
  if (D == 3) s -= (E(o - g.sz) * L[2 * g.cs + o] + E(o + g.sz) * L[2 * g.cs + o + g.sz]);
  eps[o] = s * iD[o];
}
// Jacobi!(it=1,ω): ϵ=r·iD ; r -= ωAϵ ; x += ωϵ in ONE pass.  The new residual goes to `rout` (≠ r: neighbours still read
// the old r); the caller then swaps its r/ϵ buffers.  36 instead of 12+36 B/cell.                 src/Poisson.jl:111-114
// constant-coefficient Jacobi!: L, D, iD evaluated from the cell position (wl::ConstL) — 16 instead of 32 B/cell
template <int D>
__global__ void k_jacobi_pp_cl(GridX g, float* __restrict__ rout, const float* __restrict__ r, float* __restrict__ x, float w, wl::ConstL cl, int xzero) {
  __shared__ float sDt[27], siDt[27];
  if (threadIdx.x < 27) { sDt[threadIdx.x] = cl.Dt[threadIdx.x]; siDt[threadIdx.x] = cl.iDt[threadIdx.x]; }
  __syncthreads();
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  if (!cell_ij(g, m, i, j) || !interior_ij(g, i, j)) return;
  const int k = g.k0 + pz;
  const long o = m + (long)k * g.sz;
  const int I0 = i + 1, I1 = j + 1, I2 = (D == 3) ? g.gk + k + 1 : 3;
  const int N2 = (D == 3) ? g.gnz : 8;
  // non-wall face counts of this cell and of its ±1 neighbours along each direction (ghost neighbours: r = 0 there anyway)
  const int cx0 = wl::wl_cl_cnt(I0, g.nx), cxm = wl::wl_cl_cnt(I0 - 1, g.nx), cxp = wl::wl_cl_cnt(I0 + 1, g.nx);
  const int cy0 = wl::wl_cl_cnt(I1, g.ny), cym = wl::wl_cl_cnt(I1 - 1, g.ny), cyp = wl::wl_cl_cnt(I1 + 1, g.ny);
  const int cz0 = (D == 3) ? wl::wl_cl_cnt(I2, N2) : 0, czm = (D == 3) ? wl::wl_cl_cnt(I2 - 1, N2) : 0, czp = (D == 3) ? wl::wl_cl_cnt(I2 + 1, N2) : 0;
  const float lx = wl::wl_cl_coef(I0, g.nx, cl.c[0]), lxp = wl::wl_cl_coef(I0 + 1, g.nx, cl.c[0]);
  const float ly = wl::wl_cl_coef(I1, g.ny, cl.c[1]), lyp = wl::wl_cl_coef(I1 + 1, g.ny, cl.c[1]);
  const float lz = (D == 3) ? wl::wl_cl_coef(I2, N2, cl.c[2]) : 0.f, lzp = (D == 3) ? wl::wl_cl_coef(I2 + 1, N2, cl.c[2]) : 0.f;
  auto ID = [&](int a, int b, int c) -> float { return siDt[a + 3 * b + 9 * c]; };
  const float e0 = r[o] * ID(cx0, cy0, cz0);
  float s = e0 * sDt[cx0 + 3 * cy0 + 9 * cz0];
  s += ((r[o - 1] * ID(cxm, cy0, cz0)) * lx + (r[o + 1] * ID(cxp, cy0, cz0)) * lxp);
  s += ((r[o - g.sy] * ID(cx0, cym, cz0)) * ly + (r[o + g.sy] * ID(cx0, cyp, cz0)) * lyp);
  if (D == 3) s += ((r[o - g.sz] * ID(cx0, cy0, czm)) * lz + (r[o + g.sz] * ID(cx0, cy0, czp)) * lzp);
  rout[o] = r[o] - w * s;
  x[o] = (xzero ? 0.f : x[o]) + w * e0;
}
// The same Jacobi! as a z-march (3-D): a thread walks a chunk of planes of one interior column.  Everything that depends on (i,j)
// only — the in-plane coefficients and, for each of the three z-classes of a plane (0/1/2 open z-faces), D and iD of the cell
// and iD of its four in-plane neighbours — is looked up once; r[k±1] come from a register window.  ≈4× fewer instructions per
// cell than the plane kernel, which was instruction-bound (≈200 VALU instructions for 16 B/cell).
// SHIFT: residual!'s mean shift (src/Poisson.jl:95-97) is still pending on r — it is applied to the seven values as they are loaded
// (ghost cells: (0−s)·iD = ∓0 since their iD is 0) and L₁/L∞ of the shifted residual (solver!'s first norms) are accumulated on the way.
template <int SHIFT>
__global__ void k_jacobi_march_cl(GridX g, float* __restrict__ rout, const float* __restrict__ r, float* __restrict__ x, float w, wl::ConstL cl, int zchunk,
                                  const double* __restrict__ sum, double n_inside, double* __restrict__ part, float* __restrict__ pmax, int xzero) {
  __shared__ float sDt[27], siDt[27];
  if (threadIdx.x < 27) { sDt[threadIdx.x] = cl.Dt[threadIdx.x]; siDt[threadIdx.x] = cl.iDt[threadIdx.x]; }
  __syncthreads();
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  float c = 0.f;
  if (SHIFT) { const float sm = (float)(*sum) / (float)n_inside; c = (fabsf(sm) <= 2.f * 1.1920929e-7f) ? 0.f : sm; }
  float acc = 0.f, mx = 0.f;      // a thread's |r| over its ≤ zchunk planes in Float32, across threads in Float64
  const int ks = g.k0 + pz * zchunk, ke = (ks + zchunk < g.k1) ? ks + zchunk : g.k1;
  const bool live = cell_ij(g, m, i, j) && interior_ij(g, i, j) && ks < ke;
  if (!SHIFT && !live) return;
  if (live) {
  const int I0 = i + 1, I1 = j + 1;
  const int cx0 = wl::wl_cl_cnt(I0, g.nx), cxm = wl::wl_cl_cnt(I0 - 1, g.nx), cxp = wl::wl_cl_cnt(I0 + 1, g.nx);
  const int cy0 = wl::wl_cl_cnt(I1, g.ny), cym = wl::wl_cl_cnt(I1 - 1, g.ny), cyp = wl::wl_cl_cnt(I1 + 1, g.ny);
  const float lx = wl::wl_cl_coef(I0, g.nx, cl.c[0]), lxp = wl::wl_cl_coef(I0 + 1, g.nx, cl.c[0]);
  const float ly = wl::wl_cl_coef(I1, g.ny, cl.c[1]), lyp = wl::wl_cl_coef(I1 + 1, g.ny, cl.c[1]);
  float d0[3], id0[3], idxm[3], idxp[3], idym[3], idyp[3];     // by z-class of the plane
#pragma unroll
  for (int c = 0; c < 3; c++) {
    d0[c] = sDt[cx0 + 3 * cy0 + 9 * c]; id0[c] = siDt[cx0 + 3 * cy0 + 9 * c];
    idxm[c] = siDt[cxm + 3 * cy0 + 9 * c]; idxp[c] = siDt[cxp + 3 * cy0 + 9 * c];
    idym[c] = siDt[cx0 + 3 * cym + 9 * c]; idyp[c] = siDt[cx0 + 3 * cyp + 9 * c];
  }
  auto pick = [](const float* t, int c) -> float { return c == 2 ? t[2] : (c == 1 ? t[1] : t[0]); };
  long o = m + (long)ks * g.sz;
  float rm = r[o - g.sz] - c, r0 = r[o] - c;
  for (int k = ks; k < ke; k++, o += g.sz) {
    const float rp = r[o + g.sz] - c;
    if (SHIFT) { const float av = fabsf(r0); acc += av; mx = fmaxf(mx, av); }
    const int I2 = g.gk + k + 1;
    const int cz0 = wl::wl_cl_cnt(I2, g.gnz), czm = wl::wl_cl_cnt(I2 - 1, g.gnz), czp = wl::wl_cl_cnt(I2 + 1, g.gnz);   // uniform
    const float lz = wl::wl_cl_coef(I2, g.gnz, cl.c[2]), lzp = wl::wl_cl_coef(I2 + 1, g.gnz, cl.c[2]);
    const float e0 = r0 * pick(id0, cz0);
    float s = e0 * pick(d0, cz0);
    s += (((r[o - 1] - c) * pick(idxm, cz0)) * lx + ((r[o + 1] - c) * pick(idxp, cz0)) * lxp);
    s += (((r[o - g.sy] - c) * pick(idym, cz0)) * ly + ((r[o + g.sy] - c) * pick(idyp, cz0)) * lyp);
    s += ((rm * pick(id0, czm)) * lz + (rp * pick(id0, czp)) * lzp);
    rout[o] = r0 - w * s;
    x[o] = (xzero ? 0.f : x[o]) + w * e0;          // xzero: x ≡ 0 on entry (fill!(coarse.x,0) of the V-cycle folded in)
    rm = r0; r0 = rp;
  }
  }
  if (SHIFT) {
    const double accd = block_sum((double)acc); mx = block_max(mx);
    if (threadIdx.x == 0) { part[blockIdx.x] = accd; pmax[blockIdx.x] = mx; }
  }
}
template <int D>
__global__ void k_jacobi_pp(GridX g, float* __restrict__ rout, const float* __restrict__ r, float* __restrict__ x, const float* __restrict__ L,
                            const float* __restrict__ Dg, const float* __restrict__ iD, float w, int xzero) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  if (!cell_ij(g, m, i, j) || !interior_ij(g, i, j)) return;
  const int k = g.k0 + pz;
  const long o = m + (long)k * g.sz;
  // ghost cells of r, iD and ϵ are zero in the arrays a wl_mg handle owns, so r·iD of a ghost reproduces its stored ϵ (=0)
  auto E = [&](long oo) -> float { return r[oo] * iD[oo]; };
  const float e0 = r[o] * iD[o];
  const float lx = L[o], lxp = L[o + 1], ly = L[g.cs + o], lyp = L[g.cs + o + g.sy];
  const float lz = (D == 3) ? L[2 * g.cs + o] : 0.f, lzp = (D == 3) ? L[2 * g.cs + o + g.sz] : 0.f;
  float dgv = 0.f;                       // D[I] recomputed from L (set_diag! order) instead of loaded
  dgv -= (lx + lxp);
  dgv -= (ly + lyp);
  if (D == 3) dgv -= (lz + lzp);
  float s = e0 * dgv;
  s += (E(o - 1) * lx + E(o + 1) * lxp);
  s += (E(o - g.sy) * ly + E(o + g.sy) * lyp);
  if (D == 3) s += (E(o - g.sz) * lz + E(o + g.sz) * lzp);
  rout[o] = r[o] - w * s;
  x[o] = (xzero ? 0.f : x[o]) + w * e0;
}
// residual!'s mean shift and L₁/L∞ of the shifted residual in ONE pass   src/Poisson.jl:95-97,190-191
__global__ void k_shift_norms(GridX g, float* __restrict__ r, const double* __restrict__ sum, double n_inside, double* __restrict__ part, float* __restrict__ pmax) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  const float s = (float)(*sum) / (float)n_inside;
  const bool shift = !(fabsf(s) <= 2.f * 1.1920929e-7f);
  double acc = 0.0; float mx = 0.f;
  const int nsl = wl_nslots(g);
  if (cell_ij(g, m, i, j) && interior_ij(g, i, j)) {
    for (int k = g.k0 + pz; k < g.k1; k += nsl) {
      const long o = m + (long)k * g.sz;
      float v = r[o];
      if (shift) { v = v - s; r[o] = v; }
      const float av = fabsf(v);
      acc += (double)av; mx = fmaxf(mx, av);
    }
  }
  acc = block_sum(acc);
  mx = block_max(mx);
  if (threadIdx.x == 0) { part[blockIdx.x] = acc; pmax[blockIdx.x] = mx; }
}

// restrict!  a[I] = Σ_{J∈up(I,c)} b[J]   src/MultiLevelPoisson.jl:6,13-19,49  (children summed x fastest, like CartesianIndices)
template <int D>
__global__ void k_restrict(GridX gc, GridX gf, float* __restrict__ a, const float* __restrict__ b, int cx, int cy, int cz) {
  int i, j; long m; int pz;
  wl_tile(gc, m, pz);
  if (!cell_ij(gc, m, i, j) || !interior_ij(gc, i, j)) return;
  const int k = gc.k0 + pz;
  // 0-based: coarse i (>=1) has fine children 2i-1, 2i  (Julia: 2I-2 : 2I-1)
  const int fi = cx ? 2 * i - 1 : i, fj = cy ? 2 * j - 1 : j;
  int fk = 0;
  if (D == 3) { const int K = gc.gk + k; const int FK = cz ? 2 * K - 1 : K; fk = FK - gf.gk; }
  float s = 0.f;
  if (D == 3 && cx && cy && cz) {   // full coarsening: the eight loads are issued together, then summed in the reference's order (x fastest)
    const float* __restrict__ q = b + (long)fi + (long)fj * gf.sy + (long)fk * gf.sz;
    const float v0 = q[0], v1 = q[1], v2 = q[gf.sy], v3 = q[gf.sy + 1];
    const float v4 = q[gf.sz], v5 = q[gf.sz + 1], v6 = q[gf.sz + gf.sy], v7 = q[gf.sz + gf.sy + 1];
    s += v0; s += v1; s += v2; s += v3; s += v4; s += v5; s += v6; s += v7;
    a[m + (long)k * gc.sz] = s;
    return;
  }
  for (int c = 0; c <= (D == 3 ? cz : 0); c++)
    for (int bb = 0; bb <= cy; bb++)
      for (int aa = 0; aa <= cx; aa++) s += b[(long)(fi + aa) + (long)(fj + bb) * gf.sy + (long)(fk + c) * gf.sz];
  a[m + (long)k * gc.sz] = s;
}
// prolongate!  a[I] = b[down(I,c)]   src/MultiLevelPoisson.jl:7,50
template <int D>
__device__ __forceinline__ long down_off(const GridX& gf, const GridX& gc, int i, int j, int k, int cx, int cy, int cz) {
  // 0-based: down(i) = (i+1)/2 when coarsened  (Julia (I+2)÷2 with I=i+1 -> 0-based (i+3)/2-1 = (i+1)/2)
  const int ci = cx ? (i + 1) / 2 : i, cj = cy ? (j + 1) / 2 : j;
  long o = (long)ci + (long)cj * gc.sy;
  if (D == 3) { const int K = gf.gk + k; const int CK = cz ? (K + 1) / 2 : K; o += (long)(CK - gc.gk) * gc.sz; }
  return o;
}
template <int D>
__global__ void k_prolongate(GridX gf, GridX gc, float* __restrict__ a, const float* __restrict__ b, int cx, int cy, int cz) {
  int i, j; long m; int pz;
  wl_tile(gf, m, pz);
  if (!cell_ij(gf, m, i, j) || !interior_ij(gf, i, j)) return;
  const int k = gf.k0 + pz;
  a[m + (long)k * gf.sz] = b[down_off<D>(gf, gc, i, j, k, cx, cy, cz)];
}
// prolongate! + increment! fused (Vcycle! :99-100): ϵ_f = x_c[down(I)] on the interior, ghost ϵ_f read from memory
template <int D>
__global__ void k_prolong_increment(GridX gf, GridX gc, float* __restrict__ r, float* __restrict__ x, float* __restrict__ eps, const float* __restrict__ xc,
                                    const float* __restrict__ L, const float* __restrict__ Dg, int cx, int cy, int cz, float w, int write_eps) {
  int i, j; long m; int pz;
  wl_tile(gf, m, pz);
  if (!cell_ij(gf, m, i, j) || !interior_ij(gf, i, j)) return;
  const int k = gf.k0 + pz;
  const long o = m + (long)k * gf.sz;
  // a fine ghost cell maps onto a coarse ghost cell; both hold zero in handle-owned arrays, so no predicate is needed
  auto E = [&](int ii, int jj, int kk, long) -> float { return xc[down_off<D>(gf, gc, ii, jj, kk, cx, cy, cz)]; };
  const float e0 = xc[down_off<D>(gf, gc, i, j, k, cx, cy, cz)];
  float s = e0 * Dg[o];
  s += (E(i - 1, j, k, o - 1) * L[o] + E(i + 1, j, k, o + 1) * L[o + 1]);
  s += (E(i, j - 1, k, o - gf.sy) * L[gf.cs + o] + E(i, j + 1, k, o + gf.sy) * L[gf.cs + o + gf.sy]);
  if (D == 3) s += (E(i, j, k - 1, o - gf.sz) * L[2 * gf.cs + o] + E(i, j, k + 1, o + gf.sz) * L[2 * gf.cs + o + gf.sz]);
  r[o] = r[o] - w * s;
  x[o] = x[o] + w * e0;
  if (write_eps) eps[o] = e0;
}
// ---- the coarse tail of Vcycle! in ONE launch ------------------------------------------------------------------------
// Levels of at most WL_TAIL_CELLS cells cost nothing but launch latency (≈9 launches of ≈5 µs per level and V-cycle).  One
// 1024-thread workgroup walks all of them: Jacobi!, restrict!, x_c=0 going down; smooth! at the bottom; prolongate!+increment!,
// smooth! coming up — phases separated by workgroup barriers, the arrays stay in L2.  Every phase executes the statements of the
// kernel it replaces (k_gs_init, k_increment, k_restrict, k_prolong_increment, k_gs_sweep) ⇒ bit-identical.  3-D, non-periodic,
// non-distributed levels only.
struct TailLevel { GridX g; const float* L; const float* D; const float* iD; float* x; float* eps; float* r; int cx, cy, cz; };
struct TailArgs { int n; float w; TailLevel lv[WL_TAIL_MAXLV]; };

template <class F>
__device__ __forceinline__ void tail_inside(const GridX& g, F fn) {
  const int nxi = g.nx - 2, nyi = g.ny - 2, nzi = g.nz - 2;
  const int n = nxi * nyi * nzi;
  for (int c = threadIdx.x; c < n; c += blockDim.x) {
    const int i = 1 + c % nxi, t = c / nxi, j = 1 + t % nyi, k = 1 + t / nyi;
    fn(i, j, k, (long)i + (long)j * g.sy + (long)k * g.sz);
  }
}
__device__ __forceinline__ void tail_smooth(const TailLevel& v, float w) {     // GaussSeidelRB!(it=4,ω)   src/Poisson.jl:141-148
  const GridX& g = v.g;
  tail_inside(g, [&](int, int, int, long o) { v.eps[o] = v.r[o] * v.iD[o]; });
  __syncthreads();
  for (int kk0 = 1; kk0 <= 4; kk0++) {
    tail_inside(g, [&](int i, int j, int k, long o) {
      if (((i + j + k + 3 + kk0) & 1) == 0) return;
      if (k + 1 > 2 * (g.gnz / 2) - 1) return;                                  // quirk Q4
      float s = v.r[o];
      s -= (v.eps[o - 1] * v.L[o] + v.eps[o + 1] * v.L[o + 1]);
      s -= (v.eps[o - g.sy] * v.L[g.cs + o] + v.eps[o + g.sy] * v.L[g.cs + o + g.sy]);
      s -= (v.eps[o - g.sz] * v.L[2 * g.cs + o] + v.eps[o + g.sz] * v.L[2 * g.cs + o + g.sz]);
      v.eps[o] = s * v.iD[o];
    });
    __syncthreads();
  }
  tail_inside(g, [&](int, int, int, long o) {
    v.r[o] = v.r[o] - w * Ax<3>(g, o, v.L, v.D, v.eps);
    v.x[o] = v.x[o] + w * v.eps[o];
  });
  __syncthreads();
}
__global__ void __launch_bounds__(1024) k_vcycle_tail(TailArgs a) {
  // ---- down: Jacobi!(fine); restrict!(coarse.r, fine.r); coarse.x = 0            src/MultiLevelPoisson.jl:92-95
  for (int l = 0; l + 1 < a.n; l++) {
    const TailLevel& f = a.lv[l]; const TailLevel& c = a.lv[l + 1];
    tail_inside(f.g, [&](int, int, int, long o) { f.eps[o] = f.r[o] * f.iD[o]; });
    __syncthreads();
    tail_inside(f.g, [&](int, int, int, long o) {
      f.r[o] = f.r[o] - 1.f * Ax<3>(f.g, o, f.L, f.D, f.eps);
      f.x[o] = f.x[o] + 1.f * f.eps[o];
    });
    __syncthreads();
    for (long q = threadIdx.x; q < c.g.cs; q += blockDim.x) c.x[q] = 0.f;
    tail_inside(c.g, [&](int i, int j, int k, long o) {
      const int fi = f.cx ? 2 * i - 1 : i, fj = f.cy ? 2 * j - 1 : j, fk = f.cz ? 2 * k - 1 : k;
      float s = 0.f;
      for (int cc = 0; cc <= f.cz; cc++)
        for (int bb = 0; bb <= f.cy; bb++)
          for (int aa = 0; aa <= f.cx; aa++) s += f.r[(long)(fi + aa) + (long)(fj + bb) * f.g.sy + (long)(fk + cc) * f.g.sz];
      c.r[o] = s;
    });
    __syncthreads();
  }
  // ---- bottom and up: smooth!(coarse) ; prolongate!+increment!(fine;ω) ; ... ; smooth!(first level)            :96-100
  for (int l = a.n - 1; l >= 0; l--) {
    const TailLevel& v = a.lv[l];
    if (l + 1 < a.n) {
      const TailLevel& c = a.lv[l + 1];
      const GridX& gf = v.g; const GridX& gc = c.g;
      tail_inside(gf, [&](int i, int j, int k, long o) {
        auto E = [&](int ii, int jj, int kk) -> float { return c.x[down_off<3>(gf, gc, ii, jj, kk, v.cx, v.cy, v.cz)]; };
        const float e0 = E(i, j, k);
        float s = e0 * v.D[o];
        s += (E(i - 1, j, k) * v.L[o] + E(i + 1, j, k) * v.L[o + 1]);
        s += (E(i, j - 1, k) * v.L[gf.cs + o] + E(i, j + 1, k) * v.L[gf.cs + o + gf.sy]);
        s += (E(i, j, k - 1) * v.L[2 * gf.cs + o] + E(i, j, k + 1) * v.L[2 * gf.cs + o + gf.sz]);
        v.r[o] = v.r[o] - a.w * s;
        v.x[o] = v.x[o] + a.w * e0;
      });
      __syncthreads();
    }
    tail_smooth(v, a.w);
  }
}
// ---- the same tail with every level's r, x, ϵ resident in LDS (round 2) ---------------------------------------------------------
// The tail above is pure latency: ≈36 phases, each a global load → compute → global store → barrier round trip through L2
// (≈1.6 µs per phase, 57 µs per launch, twice per time step: 14 % of a 128³ step).  The arrays of all tail levels fit into one CU's LDS
// (3 fields × ≤ 8192 cells × 4 B for the first level, ≈1/8 of that for each further one): they are loaded once, every phase runs
// LDS → LDS, and r, x, ϵ of all levels are written back at the end (same final state of every array as the global-memory tail).
// The statements per cell are those of the kernel above ⇒ bit-identical.  CL: every tail level has verified constant
// coefficients (wl::ConstL) — L, D, iD are evaluated from the cell's indices instead of being loaded (identical bits, no global
// access inside the phases at all); otherwise they are read from global memory (read-only: no store → load dependency).
struct TailLds { int lo; float inx, iny; float c[3]; };          // LDS offset of the level's r (x and ϵ follow); 1/(nx-2), 1/(ny-2); ConstL::c
struct TailArgsL { TailArgs a; TailLds q[WL_TAIL_MAXLV]; float tab[54 * WL_TAIL_MAXLV]; };   // tab: Dt[27], iDt[27] per level
struct TailCoef { float lx, lxp, ly, lyp, lz, lzp, d, id; };
template <bool CL>
__device__ __forceinline__ TailCoef tail_coef(const TailLevel& v, const TailLds& q, const float* __restrict__ tb, int i, int j, int k, int o) {
  TailCoef t;
  if (CL) {
    const GridX& g = v.g;
    const int I0 = i + 1, I1 = j + 1, I2 = k + 1;
    t.lx = wl::wl_cl_coef(I0, g.nx, q.c[0]); t.lxp = wl::wl_cl_coef(I0 + 1, g.nx, q.c[0]);
    t.ly = wl::wl_cl_coef(I1, g.ny, q.c[1]); t.lyp = wl::wl_cl_coef(I1 + 1, g.ny, q.c[1]);
    t.lz = wl::wl_cl_coef(I2, g.gnz, q.c[2]); t.lzp = wl::wl_cl_coef(I2 + 1, g.gnz, q.c[2]);
    const int n = wl::wl_cl_cnt(I0, g.nx) + 3 * wl::wl_cl_cnt(I1, g.ny) + 9 * wl::wl_cl_cnt(I2, g.gnz);
    t.d = tb[n]; t.id = tb[27 + n];
  } else {
    const GridX& g = v.g;
    t.lx = v.L[o]; t.lxp = v.L[o + 1];
    t.ly = v.L[g.cs + o]; t.lyp = v.L[g.cs + o + g.sy];
    t.lz = v.L[2 * g.cs + o]; t.lzp = v.L[2 * g.cs + o + g.sz];
    t.d = v.D[o]; t.id = v.iD[o];
  }
  return t;
}
template <class F>
__device__ __forceinline__ void tail_inside_l(const GridX& g, const TailLds& q, F fn) {
  const int nxi = g.nx - 2, nyi = g.ny - 2, nzi = g.nz - 2;
  const int n = nxi * nyi * nzi, sy = (int)g.sy, sz = (int)g.sz;
  for (int c = threadIdx.x; c < n; c += 1024) {
    // c < 8192, extents < 8192: (c+½)·(1/n) lies ≥ 1/(2n) away from an integer, the Float32 error is ≤ 1e-3 ⇒ the truncation is exact
    const int t = (int)(((float)c + 0.5f) * q.inx), i = 1 + c - t * nxi;
    const int kk = (int)(((float)t + 0.5f) * q.iny), j = 1 + t - kk * nyi, k = 1 + kk;
    fn(i, j, k, i + j * sy + k * sz);
  }
}
// The cells a thread owns on a level (cell c = threadIdx.x + u·1024, u < Q) with everything that does not change between the phases of one
// visit of the level — offsets, colour, quirk Q4, and the eight coefficients — evaluated once and kept in registers: the phases are then
// 7 LDS reads and ≈12 flops per cell (the first version re-derived indices and coefficients in every phase and was bound by that: 40 µs).
template <int Q> struct TailCells { int o[Q]; int flag[Q]; TailCoef t[Q]; };   // flag: bit 0 = (i+j+k+3)&1, bit 1 = Q4 leaves the plane unswept; o < 0: no cell
template <bool CL, int Q>
__device__ __forceinline__ void tail_cells(const TailLevel& v, const TailLds& q, const float* __restrict__ tb, TailCells<Q>& c) {
  const GridX& g = v.g;
  const int nxi = g.nx - 2, nyi = g.ny - 2, nzi = g.nz - 2;
  const int n = nxi * nyi * nzi, sy = (int)g.sy, sz = (int)g.sz;
#pragma unroll
  for (int u = 0; u < Q; u++) {
    const int cc = (int)threadIdx.x + u * 1024;
    c.o[u] = -1; c.flag[u] = 0;
    c.t[u] = TailCoef{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (cc < n) {
      // cc < 8192, extents < 8192: (cc+½)·(1/n) lies ≥ 1/(2n) away from an integer, the Float32 error is ≤ 1e-3 ⇒ the truncation is exact
      const int t = (int)(((float)cc + 0.5f) * q.inx), i = 1 + cc - t * nxi;
      const int kk = (int)(((float)t + 0.5f) * q.iny), j = 1 + t - kk * nyi, k = 1 + kk;
      const int o = i + j * sy + k * sz;
      c.o[u] = o;
      c.flag[u] = ((i + j + k + 3) & 1) | ((k + 1 > 2 * (g.gnz / 2) - 1) ? 2 : 0);
      c.t[u] = tail_coef<CL>(v, q, tb, i, j, k, o);
    }
  }
}
// GaussSeidelRB!(it=4,ω) on the level   src/Poisson.jl:141-148
template <int Q>
__device__ __forceinline__ void tail_smooth_q(const TailLevel& v, const TailLds& q, float* __restrict__ sm, const TailCells<Q>& c, float w) {
  const GridX& g = v.g;
  const int cs = (int)g.cs, sy = (int)g.sy, sz = (int)g.sz;
  float* R = sm + q.lo; float* X = R + cs; float* E = X + cs;
#pragma unroll
  for (int u = 0; u < Q; u++) if (c.o[u] >= 0) E[c.o[u]] = R[c.o[u]] * c.t[u].id;
  __syncthreads();
  for (int kk0 = 1; kk0 <= 4; kk0++) {
#pragma unroll
    for (int u = 0; u < Q; u++) {
      const int o = c.o[u];
      if (o < 0 || (((c.flag[u] & 1) + kk0) & 1) == 0 || (c.flag[u] & 2)) continue;       // colour of the sweep; quirk Q4
      const TailCoef& t = c.t[u];
      float s = R[o];
      s -= (E[o - 1] * t.lx + E[o + 1] * t.lxp);
      s -= (E[o - sy] * t.ly + E[o + sy] * t.lyp);
      s -= (E[o - sz] * t.lz + E[o + sz] * t.lzp);
      E[o] = s * t.id;
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < Q; u++) {
    const int o = c.o[u];
    if (o < 0) continue;
    const TailCoef& t = c.t[u];
    float s = E[o] * t.d;
    s += (E[o - 1] * t.lx + E[o + 1] * t.lxp);
    s += (E[o - sy] * t.ly + E[o + sy] * t.lyp);
    s += (E[o - sz] * t.lz + E[o + sz] * t.lzp);
    R[o] = R[o] - w * s;
    X[o] = X[o] + w * E[o];
  }
  __syncthreads();
}
// down: Jacobi!(fine); restrict!(coarse.r, fine.r); coarse.x = 0            src/MultiLevelPoisson.jl:92-95
template <bool CL, int Q>
__device__ __forceinline__ void tail_down(const TailLevel& f, const TailLevel& c, const TailLds& qf, const TailLds& qc, float* __restrict__ sm, const float* __restrict__ tl) {
  TailCells<Q> cl;
  tail_cells<CL, Q>(f, qf, tl, cl);
  const int fcs = (int)f.g.cs, fsy = (int)f.g.sy, fsz = (int)f.g.sz, ccs = (int)c.g.cs;
  float* R = sm + qf.lo; float* X = R + fcs; float* E = X + fcs;
  float* Rc = sm + qc.lo; float* Xc = Rc + ccs;
#pragma unroll
  for (int u = 0; u < Q; u++) if (cl.o[u] >= 0) E[cl.o[u]] = R[cl.o[u]] * cl.t[u].id;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < Q; u++) {
    const int o = cl.o[u];
    if (o < 0) continue;
    const TailCoef& t = cl.t[u];
    float s = E[o] * t.d;
    s += (E[o - 1] * t.lx + E[o + 1] * t.lxp);
    s += (E[o - fsy] * t.ly + E[o + fsy] * t.lyp);
    s += (E[o - fsz] * t.lz + E[o + fsz] * t.lzp);
    R[o] = R[o] - 1.f * s;
    X[o] = X[o] + 1.f * E[o];
  }
  __syncthreads();
  for (int o = threadIdx.x; o < ccs; o += 1024) Xc[o] = 0.f;
  tail_inside_l(c.g, qc, [&](int i, int j, int k, int o) {
    const int fi = f.cx ? 2 * i - 1 : i, fj = f.cy ? 2 * j - 1 : j, fk = f.cz ? 2 * k - 1 : k;
    float s = 0.f;
    for (int cc = 0; cc <= f.cz; cc++)
      for (int bb = 0; bb <= f.cy; bb++)
        for (int aa = 0; aa <= f.cx; aa++) s += R[(fi + aa) + (fj + bb) * fsy + (fk + cc) * fsz];
    Rc[o] = s;
  });
  __syncthreads();
}
// up: [prolongate!+increment!(v;ω) from the level below;] smooth!(v)            :96-100
template <bool CL, int Q>
__device__ __forceinline__ void tail_up(const TailLevel& v, const TailLevel& c, bool has_c, const TailLds& qv, const TailLds& qc, float* __restrict__ sm, const float* __restrict__ tl, float w) {
  TailCells<Q> cl;
  tail_cells<CL, Q>(v, qv, tl, cl);
  if (has_c) {
    const int cs = (int)v.g.cs, csy = (int)c.g.sy, csz = (int)c.g.sz;
    float* R = sm + qv.lo; float* X = R + cs;
    const float* Xc = sm + qc.lo + (int)c.g.cs;
#pragma unroll
    for (int u = 0; u < Q; u++) {
      const int o = cl.o[u];
      if (o < 0) continue;
      const TailCoef& t = cl.t[u];
      const int cc = (int)threadIdx.x + u * 1024, nxi = v.g.nx - 2, nyi = v.g.ny - 2;           // (i,j,k) of the cell, as in tail_cells
      const int tq = (int)(((float)cc + 0.5f) * qv.inx), i = 1 + cc - tq * nxi;
      const int kq = (int)(((float)tq + 0.5f) * qv.iny), j = 1 + tq - kq * nyi, k = 1 + kq;
      // down(I) of wl down_off (0-based (i+1)/2 in a coarsened direction); gk = 0 on tail levels
      auto E = [&](int ii, int jj, int kk) -> float {
        const int ci = v.cx ? (ii + 1) / 2 : ii, cj = v.cy ? (jj + 1) / 2 : jj, ck = v.cz ? (kk + 1) / 2 : kk;
        return Xc[ci + cj * csy + ck * csz];
      };
      const float e0 = E(i, j, k);
      float s = e0 * t.d;
      s += (E(i - 1, j, k) * t.lx + E(i + 1, j, k) * t.lxp);
      s += (E(i, j - 1, k) * t.ly + E(i, j + 1, k) * t.lyp);
      s += (E(i, j, k - 1) * t.lz + E(i, j, k + 1) * t.lzp);
      R[o] = R[o] - w * s;
      X[o] = X[o] + w * e0;
    }
    __syncthreads();
  }
  tail_smooth_q<Q>(v, qv, sm, cl, w);
}
__device__ __forceinline__ int tail_qsel(const GridX& g) {     // cells per thread of a level, rounded up to 1, 2 or 4 (8 never occurs: see vcycle_tail)
  const int n = (g.nx - 2) * (g.ny - 2) * (g.nz - 2), q = (n + 1023) >> 10;
  return q <= 1 ? 1 : (q <= 2 ? 2 : (q <= 4 ? 4 : 8));
}
template <bool CL>
__global__ void __launch_bounds__(1024) k_vcycle_tail_lds(TailArgsL b) {
  extern __shared__ float sm[];
  const TailArgs& a = b.a;
  float* tb = sm;                                                               // [level][54]: Dt, iDt
  if (CL) {   // the tables are indexed per lane: read them through the kernel-argument segment's address rather than as scalars
    const float* kt = (const float*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(TailArgsL, tab));
    for (int q = threadIdx.x; q < 54 * a.n; q += 1024) tb[q] = kt[q];
  }
  // ---- load r, x, ϵ of every level (ghost cells included: the stencils read ϵ and x_c there)
  for (int l = 0; l < a.n; l++) {
    const TailLevel& v = a.lv[l];
    const int cs = (int)v.g.cs;
    float* R = sm + b.q[l].lo;
    for (int o = threadIdx.x; o < cs; o += 1024) { R[o] = v.r[o]; R[cs + o] = v.x[o]; R[2 * cs + o] = v.eps[o]; }
  }
  __syncthreads();
  for (int l = 0; l + 1 < a.n; l++) {
    const TailLevel& f = a.lv[l]; const TailLevel& c = a.lv[l + 1];
    const float* tl = tb + 54 * l;
    switch (tail_qsel(f.g)) {
      case 1: tail_down<CL, 1>(f, c, b.q[l], b.q[l + 1], sm, tl); break;
      case 2: tail_down<CL, 2>(f, c, b.q[l], b.q[l + 1], sm, tl); break;
      default: tail_down<CL, 4>(f, c, b.q[l], b.q[l + 1], sm, tl); break;   // (the launcher admits at most 4096 interior cells per level)
    }
  }
  for (int l = a.n - 1; l >= 0; l--) {
    const TailLevel& v = a.lv[l];
    const bool has_c = l + 1 < a.n;
    const int lc = has_c ? l + 1 : l;                                           // (kernel arguments are indexed, never pointed to: no private copy)
    const float* tl = tb + 54 * l;
    switch (tail_qsel(v.g)) {
      case 1: tail_up<CL, 1>(v, a.lv[lc], has_c, b.q[l], b.q[lc], sm, tl, a.w); break;
      case 2: tail_up<CL, 2>(v, a.lv[lc], has_c, b.q[l], b.q[lc], sm, tl, a.w); break;
      default: tail_up<CL, 4>(v, a.lv[lc], has_c, b.q[l], b.q[lc], sm, tl, a.w); break;
    }
  }
  // ---- write r, x, ϵ of every level back
  for (int l = 0; l < a.n; l++) {
    const TailLevel& v = a.lv[l];
    const int cs = (int)v.g.cs;
    const float* R = sm + b.q[l].lo;
    for (int o = threadIdx.x; o < cs; o += 1024) { v.r[o] = R[o]; v.x[o] = R[cs + o]; v.eps[o] = R[2 * cs + o]; }
  }
}
// restrictL!  a[I,i] = restrictL(I,i,b,c)   src/MultiLevelPoisson.jl:9-11,20-26,45  (BC!(a,0) applied afterwards by bc_vec)
template <int D>
__global__ void k_restrictL(GridX gc, GridX gf, float* __restrict__ a, const float* __restrict__ b, int cx, int cy, int cz) {
  int i, j; long m; int pz;
  wl_tile(gc, m, pz);
  if (!cell_ij(gc, m, i, j) || !interior_ij(gc, i, j)) return;
  const int k = gc.k0 + pz;
  const int c[3] = {cx, cy, (D == 3) ? cz : 0};
  int f0[3] = {cx ? 2 * i - 1 : i, cy ? 2 * j - 1 : j, 0};
  if (D == 3) { const int K = gc.gk + k; f0[2] = (cz ? 2 * K - 1 : K) - gf.gk; }
  for (int n = 0; n < D; n++) {
    // faces with normal n: only the first child index along n (Julia 2I-2), both children in the other coarsened dirs
    int hi[3] = {c[0], c[1], c[2]};
    hi[n] = 0;
    float s = 0.f;
    for (int cc = 0; cc <= hi[2]; cc++)
      for (int bb = 0; bb <= hi[1]; bb++)
        for (int aa = 0; aa <= hi[0]; aa++) s += b[(long)n * gf.cs + (long)(f0[0] + aa) + (long)(f0[1] + bb) * gf.sy + (long)(f0[2] + cc) * gf.sz];
    a[(long)n * gc.cs + m + (long)k * gc.sz] = c[n] ? s / 2 : s;
  }
}

inline void mask_of(const GridX& fine, const GridX& coarse, int& cx, int& cy, int& cz) {
  cx = coarse.nx < fine.nx; cy = coarse.ny < fine.ny; cz = (fine.D == 3) ? (coarse.gnz < fine.gnz) : 0;
}
}  // namespace

#define DSEL(D, KERN, ...)                                                           \
  do { if ((D) == 3) hipLaunchKernelGGL(KERN<3>, __VA_ARGS__); else hipLaunchKernelGGL(KERN<2>, __VA_ARGS__); } while (0)

namespace wl {
int set_diag(float* Dg, float* iD, const float* L, const GridX& g, hipStream_t s) {
  DSEL(g.D, k_set_diag, wl_plane_grid(g, g.k1 - g.k0), dim3(WL_BLOCK), 0, s, g, Dg, iD, L);
  WL_LAUNCH_CHECK(); return 0;
}
int mult(float* z, const float* L, const float* Dg, const float* x, const GridX& g, hipStream_t s) {
  WL_HIP(hipMemsetAsync(z, 0, sizeof(float) * (size_t)g.cs, s));
  DSEL(g.D, k_mult, wl_plane_grid(g, g.k1 - g.k0), dim3(WL_BLOCK), 0, s, g, z, L, Dg, x);
  WL_LAUNCH_CHECK(); return 0;
}
// residual!: r and the LOCAL Σr (-> ws.res_d[0]); the global mean shift follows once Σr is combined over ranks
int residual_part(float* r, const float* x, const float* z, const float* L, const float* Dg, const float* iD, const GridX& g, const RedWs& ws, hipStream_t s) {
  const int np = g.k1 - g.k0;
  dim3 grid = wl_plane_grid(g, wl_red_slots(g, np));
  DSEL(g.D, k_residual, grid, dim3(WL_BLOCK), 0, s, g, r, x, z, L, Dg, iD, ws.pa);
  hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(WL_BLOCK), 0, s, ws.pa, (int)grid.x, ws.res_d + 0);
  WL_LAUNCH_CHECK(); return 0;
}
#define DSEL2(D, CLF, KERN, ...)                                                                                     \
  do { if ((D) == 3) { if (CLF) hipLaunchKernelGGL((KERN<3, 1>), __VA_ARGS__); else hipLaunchKernelGGL((KERN<3, 0>), __VA_ARGS__); } \
       else { if (CLF) hipLaunchKernelGGL((KERN<2, 1>), __VA_ARGS__); else hipLaunchKernelGGL((KERN<2, 0>), __VA_ARGS__); } } while (0)
int div_residual(float* z, float* xout, float* r, const float* x, const float* u, const float* L, const float* Dg, const float* iD, const GridX& g, float dt, const RedWs& ws, const ConstL& cl, hipStream_t s) {
  const int zc = wl_march_chunk(g, g.nz);
  dim3 grid = wl_plane_grid(g, wl_march_slots(g.nz, zc));
  DSEL2(g.D, cl.on, k_div_residual, grid, dim3(WL_BLOCK), 0, s, g, z, xout, r, x, u, L, Dg, iD, dt, ws.pa, cl, zc, 0, g.nz);
  hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(WL_BLOCK), 0, s, ws.pa, (int)grid.x, ws.res_d + 0);
  WL_LAUNCH_CHECK(); return 0;
}
// The same on a level with a body: the local planes [0,na) and [nb,nz) follow the constant-coefficient pattern `far` (coefficients from
// the position), the planes [na,nb) around the body read L.  Three launches, their partial sums side by side, one final sum.
int div_residual_split(float* z, float* xout, float* r, const float* x, const float* u, const float* L, const float* Dg, const float* iD, const GridX& g, float dt, const RedWs& ws,
                       const ConstL& near, const ConstL& far, int na, int nb, hipStream_t s) {
  const int lo[3] = {0, na, nb}, hi[3] = {na, nb, g.nz};
  int off = 0;
  for (int q = 0; q < 3; q++) {
    const int np = hi[q] - lo[q];
    if (np <= 0) continue;
    const ConstL& cl = q == 1 ? near : far;
    const int zc = wl_march_chunk(g, np);
    dim3 grid = wl_plane_grid(g, wl_march_slots(np, zc));
    if (off + (int)grid.x > WL_MAXPART) { wl_set_error("div_residual_split: too many partial sums"); return WL_EINVAL; }
    DSEL2(g.D, cl.on, k_div_residual, grid, dim3(WL_BLOCK), 0, s, g, z, xout, r, x, u, L, Dg, iD, dt, ws.pa + off, cl, zc, lo[q], hi[q]);
    off += (int)grid.x;
  }
  hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(WL_BLOCK), 0, s, ws.pa, off, ws.res_d + 0);
  WL_LAUNCH_CHECK(); return 0;
}
// Block order of the projection tails (experiments: WL_TAIL_LIN / WL_TAIL_PAIR in a -DWL_EXPERIMENTS build; profiles/r03_experiments.md §4, §8):
//   first tail (k_project_unscale): linear block order, one plane per block, one cell per thread — 1.04 -> 0.875 ms at 512³ (two cells per thread: 0.915);
//   second tail (+ flux_out + max σ): in linear order the one-cell kernel is L1-bound (13 dword loads per cell: 1.33 -> 1.55 ms), with two cells per
//   thread (k_project_cfl2, ≈6.5 loads per cell) linear order wins: 1.34 -> 1.18 ms.
static int tail_pair_bits() { static const int v = wl_exp_int("WL_TAIL_PAIR", 2); return v; }     // bit 0: first tail, bit 1: second tail
static int tail_pair() { return tail_pair_bits() & 1; }
static int tail_lin(int bit) { static const int v = wl_exp_int("WL_TAIL_LIN", 1); return (v >> bit) & 1; }
int project_unscale(float* u, const float* L, const float* x, float* pout, const GridX& g, float dt, const ConstL& cl, hipStream_t s, const BcFold* fold) {
  const int lin = tail_lin(0) && g.D == 3;
  const int zc = lin ? 1 : wl_march_chunk(g, g.nz);
  BcFold bc{0, {0.f, 0.f, 0.f}};
  if (fold && fold->on && g.D == 3 && g.nz == g.gnz && g.nx >= 6 && g.ny >= 6 && g.nz >= 6) bc = *fold;
  bc.go = fold ? fold->go : nullptr;
  if (tail_pair() && lin && cl.on && (g.nx & 1) == 0 && g.k0 >= 1) {   // two cells per thread (k_project_unscale2): x[o−sz] of plane 0 is never read (k0 >= 1)
    hipLaunchKernelGGL(k_project_unscale2, dim3(pl_grid(g, g.nz)), dim3(WL_BLOCK), 0, s, g, u, x, pout, dt, cl, 0, g.nz, bc);
    WL_LAUNCH_CHECK(); return 0;
  }
  DSEL2(g.D, cl.on, k_project_unscale, wl_plane_grid(g, wl_march_slots(g.nz, zc)), dim3(WL_BLOCK), 0, s, g, u, L, x, pout, dt, cl, zc, 0, g.nz, bc, lin);
  WL_LAUNCH_CHECK(); return 0;
}
// level with a body: planes [0,na) and [nb,nz) with the constant-coefficient pattern `far`, [na,nb) reading L (see div_residual_split)
int project_unscale_split(float* u, const float* L, const float* x, float* pout, const GridX& g, float dt, const ConstL& near, const ConstL& far, int na, int nb, hipStream_t s) {
  const int lo[3] = {0, na, nb}, hi[3] = {na, nb, g.nz};
  for (int q = 0; q < 3; q++) {
    const int np = hi[q] - lo[q];
    if (np <= 0) continue;
    const ConstL& cl = q == 1 ? near : far;
    const int zc = wl_march_chunk(g, np);
    DSEL2(g.D, cl.on, k_project_unscale, wl_plane_grid(g, wl_march_slots(np, zc)), dim3(WL_BLOCK), 0, s, g, u, L, x, pout, dt, cl, zc, lo[q], hi[q], BcFold{0, {0.f, 0.f, 0.f}}, 0);
  }
  WL_LAUNCH_CHECK(); return 0;
}
// solver!'s break test on the device (src/MultiLevelPoisson.jl:122 with l1n_tol, src/Poisson.jl:194): res_f[out_slot] = 1 if L₁ < r1tol ∧ L∞ < rinftol — the statements
// wl_mg::solve evaluates on the host from the same two numbers — and, with check_head, the fused head's mean-shift test |Σr/N| ≤ 2eps (src/Poisson.jl:96) as well.
// The host takes its decision from this flag (read back with the norms), so a tail kernel gated by it and the solver loop can never disagree.
__global__ void k_decide(const double* __restrict__ res_d, float* __restrict__ res_f, double r1tol, double rinftol, double ninside, int check_head, int slot_d, int slot_f, int out_slot) {
  const float rnew = (float)res_d[slot_d], rinf = res_f[slot_f];
  bool ok = (double)rnew < r1tol && (double)rinf < rinftol;
  if (check_head) { const float sm = (float)res_d[0] / (float)ninside; ok = ok && fabsf(sm) <= 2.f * 1.1920929e-7f; }
  res_f[out_slot] = ok ? 1.f : 0.f;
}
int decide_converged(const RedWs& ws, double r1tol, double rinftol, double ninside, int check_head, int slot_d, int slot_f, int out_slot, hipStream_t s) {
  hipLaunchKernelGGL(k_decide, dim3(1), dim3(1), 0, s, (const double*)ws.res_d, ws.res_f, r1tol, rinftol, ninside, check_head, slot_d, slot_f, out_slot);
  WL_LAUNCH_CHECK(); return 0;
}
bool project_cfl_pair_path(const GridX& g, const ConstL& cl) { return (tail_pair_bits() & 2) && g.D == 3 && cl.on && (g.nx & 1) == 0 && g.k0 >= 1 && g.k1 <= g.nz - 1; }
// projection tail + CFL's σ and max(σ) -> ws.res_f[slot_f]; u_out must not alias u_in
int project_cfl(float* uout, const float* uin, const float* L, const float* x, float* pout, float* sigma, const GridX& g, float dt, const ConstL& cl, const RedWs& ws, int slot_f, hipStream_t s, int store_sigma, const BcFold* fold) {
  if (uout == uin) { wl_set_error("project_cfl: output aliases input"); return WL_EINVAL; }
  int kfirst = 0, klast = 1;
  if (g.D == 3) { kfirst = (g.gk + g.k0 == 1) ? g.k0 - 1 : g.k0; klast = (g.gk + g.k1 == g.gnz - 1) ? g.k1 + 1 : g.k1; }
  BcFold bc{0, {0.f, 0.f, 0.f}};
  if (fold && fold->on && g.D == 3 && g.nz == g.gnz && g.nx >= 6 && g.ny >= 6 && g.nz >= 6) bc = *fold;
  bc.go = fold ? fold->go : nullptr;
  if (bc.go && !project_cfl_pair_path(g, cl)) { wl_set_error("project_cfl: a tail queued ahead of the convergence read needs the two-cells-per-thread form"); return WL_EINVAL; }
  if (fold && fold->usub && !(bc.on && project_cfl_pair_path(g, cl))) { wl_set_error("project_cfl: deferred BC! needs the folded two-cells-per-thread tail"); return WL_EINVAL; }
  if (project_cfl_pair_path(g, cl)) {   // two cells per thread, linear order (k_project_cfl2)
    hipLaunchKernelGGL(k_enc_init, dim3(1), dim3(WL_ENC_SLOTS), 0, s, reinterpret_cast<int*>(ws.pm));
    hipLaunchKernelGGL(k_project_cfl2, dim3(pl_grid(g, g.nz)), dim3(WL_BLOCK), 0, s, g, uout, uin, x, pout, sigma, dt, cl, kfirst, klast, ws.pm, 0, g.nz, store_sigma, bc);
    hipLaunchKernelGGL(k_fin_max_enc, dim3(1), dim3(WL_BLOCK), 0, s, reinterpret_cast<const int*>(ws.pm), ws.res_f + slot_f);
    WL_LAUNCH_CHECK(); return 0;
  }
  const int lin = tail_lin(1) && g.D == 3;
  const int zc = lin ? 1 : wl_march_chunk(g, g.nz);
  const dim3 grid = wl_plane_grid(g, wl_march_slots(g.nz, zc));
  if (lin) hipLaunchKernelGGL(k_enc_init, dim3(1), dim3(WL_ENC_SLOTS), 0, s, reinterpret_cast<int*>(ws.pm));
  DSEL2(g.D, cl.on, k_project_cfl, grid, dim3(WL_BLOCK), 0, s, g, uout, uin, L, x, pout, sigma, dt, cl, zc, kfirst, klast, ws.pm, 0, g.nz, store_sigma, bc, lin);
  if (lin) hipLaunchKernelGGL(k_fin_max_enc, dim3(1), dim3(WL_BLOCK), 0, s, reinterpret_cast<const int*>(ws.pm), ws.res_f + slot_f);
  else hipLaunchKernelGGL(k_fin_max2, dim3(1), dim3(WL_BLOCK), 0, s, ws.pm, (int)grid.x, ws.res_f + slot_f);
  WL_LAUNCH_CHECK(); return 0;
}
int project_cfl_split(float* uout, const float* uin, const float* L, const float* x, float* pout, float* sigma, const GridX& g, float dt, const ConstL& near, const ConstL& far,
                      int na, int nb, const RedWs& ws, int slot_f, hipStream_t s, int store_sigma) {
  if (uout == uin) { wl_set_error("project_cfl: output aliases input"); return WL_EINVAL; }
  int kfirst = 0, klast = 1;
  if (g.D == 3) { kfirst = (g.gk + g.k0 == 1) ? g.k0 - 1 : g.k0; klast = (g.gk + g.k1 == g.gnz - 1) ? g.k1 + 1 : g.k1; }
  const int lo[3] = {0, na, nb}, hi[3] = {na, nb, g.nz};
  int off = 0;
  for (int q = 0; q < 3; q++) {
    const int np = hi[q] - lo[q];
    if (np <= 0) continue;
    const ConstL& cl = q == 1 ? near : far;
    const int zc = wl_march_chunk(g, np);
    const dim3 grid = wl_plane_grid(g, wl_march_slots(np, zc));
    if (off + (int)grid.x > WL_MAXPART) { wl_set_error("project_cfl_split: too many partial maxima"); return WL_EINVAL; }
    DSEL2(g.D, cl.on, k_project_cfl, grid, dim3(WL_BLOCK), 0, s, g, uout, uin, L, x, pout, sigma, dt, cl, zc, kfirst, klast, ws.pm + off, lo[q], hi[q], store_sigma, BcFold{0, {0.f, 0.f, 0.f}}, 0);
    off += (int)grid.x;
  }
  hipLaunchKernelGGL(k_fin_max2, dim3(1), dim3(WL_BLOCK), 0, s, ws.pm, off, ws.res_f + slot_f);
  WL_LAUNCH_CHECK(); return 0;
}
// host-synchronising (update! time only): reads one interior face value per component, then verifies the whole array on device
int check_const_L(const float* L, const GridX& g, ConstL* out, int* dev_flag, hipStream_t s) {
  out->on = 0; out->c[0] = out->c[1] = out->c[2] = 0.f;
  if (g.nx < 4 || g.ny < 4 || (g.D == 3 && (g.k1 - g.k0) < 1)) return 0;   // (4: one face per direction is not a wall face)
  // a cell whose lower faces are not wall faces in any direction: Julia index 3 in x,y and (globally) >= 3 in z
  int kk = (g.D == 3) ? g.k0 + ((g.gk + g.k0 + 1 >= 3) ? 0 : 1) : 0;
  if (g.D == 3 && (kk >= g.k1 || g.gk + kk + 1 >= g.gnz)) return 0;
  const long o = 2 + 2 * g.sy + (long)kk * g.sz;
  float c[3] = {0.f, 0.f, 0.f};
  for (int a = 0; a < g.D; a++) WL_HIP(hipMemcpyAsync(&c[a], L + (long)a * g.cs + o, sizeof(float), hipMemcpyDeviceToHost, s));
  WL_HIP(hipMemsetAsync(dev_flag, 0, sizeof(int), s));
  WL_HIP(hipStreamSynchronize(s));
  DSEL(g.D, k_check_const_L, wl_plane_grid(g, g.nz), dim3(WL_BLOCK), 0, s, g, L, c[0], c[1], c[2], dev_flag);
  int bad = 1;
  WL_HIP(hipMemcpyAsync(&bad, dev_flag, sizeof(int), hipMemcpyDeviceToHost, s));
  WL_HIP(hipStreamSynchronize(s));
  for (int a = 0; a < 3; a++) out->c[a] = c[a];       // (the sampled constants and their tables are returned even when the test fails:
  {                                                   //  smooth! may still use them on the planes that const_plane_range finds clean)
    out->on = bad ? 0 : 1;
    for (int nz = 0; nz < 3; nz++) for (int ny = 0; ny < 3; ny++) for (int nx = 0; nx < 3; nx++) {
      // pair sums lower+upper face: 2 non-wall faces c+c, one c+0 (or 0+c: same value), none 0+0
      const float px = nx == 2 ? c[0] + c[0] : (nx == 1 ? c[0] + 0.f : 0.f), py = ny == 2 ? c[1] + c[1] : (ny == 1 ? c[1] + 0.f : 0.f);
      const float pzv = nz == 2 ? c[2] + c[2] : (nz == 1 ? c[2] + 0.f : 0.f);
      float d = 0.f; d -= px; d -= py; if (g.D == 3) d -= pzv;                    // set_diag! order  src/Poisson.jl:49-55
      out->Dt[nx + 3 * ny + 9 * nz] = d; out->iDt[nx + 3 * ny + 9 * nz] = (d == 0.f) ? d : 1.0f / d;
    }
  }
  return 0;
}
// planes [za,zb] (inclusive, local indices) outside which every coefficient of L follows the constant pattern with constants c;
// za > zb: none deviates.  Host-synchronising (update! time only).  Single domain, 3-D.
int const_plane_range(const float* L, const GridX& g, const float* c, int* za, int* zb, hipStream_t s) {
  unsigned char* d = nullptr;
  WL_HIP(hipMalloc((void**)&d, (size_t)g.nz));
  WL_HIP(hipMemsetAsync(d, 0, (size_t)g.nz, s));
  DSEL(g.D, k_plane_bad, wl_plane_grid(g, g.nz), dim3(WL_BLOCK), 0, s, g, L, c[0], c[1], c[2], d);
  std::vector<unsigned char> h((size_t)g.nz);
  WL_HIP(hipMemcpyAsync(h.data(), d, h.size(), hipMemcpyDeviceToHost, s));
  WL_HIP(hipStreamSynchronize(s));
  (void)hipFree(d);
  *za = g.nz; *zb = -1;
  for (int k = 0; k < g.nz; k++) if (h[(size_t)k]) { if (k < *za) *za = k; if (k > *zb) *zb = k; }
  return 0;
}
int mean_shift(float* r, const GridX& g, const RedWs& ws, hipStream_t s) {
  hipLaunchKernelGGL(k_mean_shift, wl_plane_grid(g, g.k1 - g.k0), dim3(WL_BLOCK), 0, s, g, r, ws.res_d + 0, (double)wl_ninside_global(wl_grid{g.D, g.nx, g.ny, g.nz, g.k0, g.k1, g.gk, g.gnz}));
  WL_LAUNCH_CHECK(); return 0;
}
int residual(float* r, const float* x, const float* z, const float* L, const float* Dg, const float* iD, const GridX& g, const RedWs& ws, hipStream_t s) {
  WL_TRY(residual_part(r, x, z, L, Dg, iD, g, ws, s));
  return mean_shift(r, g, ws, s);
}
int norms_dev(const float* r, const GridX& g, const RedWs& ws, int slot_d, int slot_f, hipStream_t s) {
  dim3 grid = wl_plane_grid(g, wl_red_slots(g, g.k1 - g.k0));
  hipLaunchKernelGGL(k_norms, grid, dim3(WL_BLOCK), 0, s, g, r, ws.pa, ws.pm);
  hipLaunchKernelGGL(k_final_sum_max, dim3(1), dim3(WL_BLOCK), 0, s, ws.pa, ws.pm, (int)grid.x, ws.res_d + slot_d, ws.res_f + slot_f);
  WL_LAUNCH_CHECK(); return 0;
}
int finalize_sum_max(const RedWs& ws, int nparts, int slot_d, int slot_f, hipStream_t s) {
  hipLaunchKernelGGL(k_final_sum_max, dim3(1), dim3(WL_BLOCK), 0, s, ws.pa, ws.pm, nparts, ws.res_d + slot_d, ws.res_f + slot_f);
  WL_LAUNCH_CHECK(); return 0;
}
int increment(float* r, float* x, const float* eps, const float* L, const float* Dg, const GridX& g, float w, hipStream_t s) {
  DSEL(g.D, k_increment, wl_plane_grid(g, g.k1 - g.k0), dim3(WL_BLOCK), 0, s, g, r, x, eps, L, Dg, w);
  WL_LAUNCH_CHECK(); return 0;
}
// Jacobi!(it=1,ω)   src/Poisson.jl:111-114
int jacobi(float* eps, float* r, float* x, const float* L, const float* Dg, const float* iD, const GridX& g, float w, bool write_eps, hipStream_t s) {
  // two-pass form (ϵ staged in memory): exact reference order, in place, race free
  (void)write_eps;
  hipLaunchKernelGGL(k_gs_init, wl_plane_grid(g, g.k1 - g.k0), dim3(WL_BLOCK), 0, s, g, eps, r, iD);
  WL_LAUNCH_CHECK();
  return increment(r, x, eps, L, Dg, g, w, s);
}
int gs_init(float* eps, const float* r, const float* iD, const GridX& g, hipStream_t s) {
  hipLaunchKernelGGL(k_gs_init, wl_plane_grid(g, g.k1 - g.k0), dim3(WL_BLOCK), 0, s, g, eps, r, iD);
  WL_LAUNCH_CHECK(); return 0;
}
int gs_init_sweep1(float* eps, const float* r, const float* L, const float* iD, const GridX& g, hipStream_t s) {
  DSEL(g.D, k_gs_init_sweep1, wl_plane_grid(g, g.k1 - g.k0), dim3(WL_BLOCK), 0, s, g, eps, r, L, iD);
  WL_LAUNCH_CHECK(); return 0;
}
static int g_jacobi_march = 1;
void jacobi_march_enable(int on) { g_jacobi_march = on; }
bool jacobi_takes_shift(const GridX& g, const ConstL& cl) { return cl.on && g.D == 3 && g_jacobi_march && g.nz == g.gnz; }
// Jacobi! on a residual whose mean shift is still pending (Σr in ws.res_d[0]): shift on load, L₁ -> res_d[slot_d], L∞ -> res_f[slot_f]
int jacobi_pp_shift(float* rout, const float* r, float* x, const GridX& g, float w, const ConstL& cl, const RedWs& ws, int slot_d, int slot_f, hipStream_t s) {
  if (!jacobi_takes_shift(g, cl)) { wl_set_error("jacobi_pp_shift: level not eligible"); return WL_EINVAL; }
  const int zc = wl_march_chunk(g, g.k1 - g.k0);
  dim3 grid = wl_plane_grid(g, wl_march_slots(g.k1 - g.k0, zc));
  const double ni = (double)wl_ninside_global(wl_grid{g.D, g.nx, g.ny, g.nz, g.k0, g.k1, g.gk, g.gnz});
  hipLaunchKernelGGL(k_jacobi_march_cl<1>, grid, dim3(WL_BLOCK), 0, s, g, rout, r, x, w, cl, zc, (const double*)(ws.res_d + 0), ni, ws.pa, ws.pm, 0);
  hipLaunchKernelGGL(k_final_sum_max, dim3(1), dim3(WL_BLOCK), 0, s, ws.pa, ws.pm, (int)grid.x, ws.res_d + slot_d, ws.res_f + slot_f);
  WL_LAUNCH_CHECK(); return 0;
}
int jacobi_pp(float* rout, const float* r, float* x, const float* L, const float* Dg, const float* iD, const GridX& g, float w, const ConstL& cl, hipStream_t s, int xzero) {
  if (cl.on && g.D == 3 && g_jacobi_march) {
    const int zc = wl_march_chunk(g, g.k1 - g.k0);
    hipLaunchKernelGGL(k_jacobi_march_cl<0>, wl_plane_grid(g, wl_march_slots(g.k1 - g.k0, zc)), dim3(WL_BLOCK), 0, s, g, rout, r, x, w, cl, zc, (const double*)nullptr, 0.0, (double*)nullptr, (float*)nullptr, xzero);
  } else if (cl.on) DSEL(g.D, k_jacobi_pp_cl, wl_plane_grid(g, g.k1 - g.k0), dim3(WL_BLOCK), 0, s, g, rout, r, x, w, cl, xzero);
  else DSEL(g.D, k_jacobi_pp, wl_plane_grid(g, g.k1 - g.k0), dim3(WL_BLOCK), 0, s, g, rout, r, x, L, Dg, iD, w, xzero);
  WL_LAUNCH_CHECK(); return 0;
}
// if (levels below `first` exist) Vcycle!(first); smooth!(first)  — for the levels handed over in `lv` (coarsening flags in lv[l].c*)
int g_tail_lds = 1;
void tail_lds_enable(int on) { g_tail_lds = on; }
int vcycle_tail(const TailLevelHost* lv, int n, float w, hipStream_t s) {
  if (n < 1 || n > WL_TAIL_MAXLV) { wl_set_error("vcycle_tail: bad level count"); return WL_EINVAL; }
  TailArgsL b; TailArgs& a = b.a; a.n = n; a.w = w;
  for (int l = 0; l < n; l++) a.lv[l] = TailLevel{lv[l].g, lv[l].L, lv[l].D, lv[l].iD, lv[l].x, lv[l].eps, lv[l].r, lv[l].cx, lv[l].cy, lv[l].cz};
  // LDS-resident variant: r, x, ϵ of all levels (+ the coefficient tables) must fit into one CU's LDS
  bool cl = true, fits = g_tail_lds != 0;
  long lo = 54 * WL_TAIL_MAXLV;
  for (int l = 0; l < n; l++) {
    const GridX& g = lv[l].g;
    if (g.D != 3 || g.gk != 0 || g.nz != g.gnz || g.nx < 3 || g.ny < 3 || g.nz < 3 || g.cs > WL_TAIL_CELLS) fits = false;
    if ((long)(g.nx - 2) * (g.ny - 2) * (g.nz - 2) > 4096) fits = false;      // a thread caches the coefficients of at most 4 cells per level (k_vcycle_tail_lds)
    cl = cl && lv[l].cl && lv[l].cl->on;
    b.q[l].lo = (int)lo; b.q[l].inx = 1.f / (float)(g.nx - 2); b.q[l].iny = 1.f / (float)(g.ny - 2);
    for (int c = 0; c < 3; c++) b.q[l].c[c] = (lv[l].cl && lv[l].cl->on) ? lv[l].cl->c[c] : 0.f;
    for (int t = 0; t < 27; t++) { b.tab[54 * l + t] = (lv[l].cl && lv[l].cl->on) ? lv[l].cl->Dt[t] : 0.f; b.tab[54 * l + 27 + t] = (lv[l].cl && lv[l].cl->on) ? lv[l].cl->iDt[t] : 0.f; }
    lo += 3 * g.cs;
  }
  const size_t bytes = (size_t)lo * sizeof(float);
  if (fits && bytes <= 156 * 1024) {
    static bool attr[2] = {false, false};
    if (!attr[cl ? 1 : 0]) {   // dynamic LDS beyond 64 KiB has to be requested once per kernel
      WL_HIP(cl ? hipFuncSetAttribute((const void*)k_vcycle_tail_lds<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024)
                : hipFuncSetAttribute((const void*)k_vcycle_tail_lds<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
      attr[cl ? 1 : 0] = true;
    }
    if (cl) hipLaunchKernelGGL(k_vcycle_tail_lds<true>, dim3(1), dim3(1024), bytes, s, b);
    else hipLaunchKernelGGL(k_vcycle_tail_lds<false>, dim3(1), dim3(1024), bytes, s, b);
  } else {
    hipLaunchKernelGGL(k_vcycle_tail, dim3(1), dim3(1024), 0, s, a);
  }
  WL_LAUNCH_CHECK(); return 0;
}
// one stage of pcg! ; stages 0-2 leave their dot product in ws.res_d[0]
int pcg_stage(int stage, float* eps, float* r, float* x, float* z, const float* L, const float* Dg, const float* iD, const GridX& g, float a, int more, const RedWs& ws, hipStream_t s) {
  dim3 grid = wl_plane_grid(g, wl_red_slots(g, g.k1 - g.k0));
#define WL_PCG(ST) do { if (g.D == 3) hipLaunchKernelGGL((k_pcg<3, ST>), grid, dim3(WL_BLOCK), 0, s, g, eps, r, x, z, L, Dg, iD, a, more, ws.pa); \
                        else hipLaunchKernelGGL((k_pcg<2, ST>), grid, dim3(WL_BLOCK), 0, s, g, eps, r, x, z, L, Dg, iD, a, more, ws.pa); } while (0)
  switch (stage) { case 0: WL_PCG(0); break; case 1: WL_PCG(1); break; case 2: WL_PCG(2); break; default: WL_PCG(3); }
#undef WL_PCG
  if (stage != 3 && !(stage == 2 && !more)) hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(WL_BLOCK), 0, s, ws.pa, (int)grid.x, ws.res_d + 0);
  WL_LAUNCH_CHECK(); return 0;
}
int shift_norms_dev(float* r, const GridX& g, const RedWs& ws, int slot_d, int slot_f, hipStream_t s) {
  dim3 grid = wl_plane_grid(g, wl_red_slots(g, g.k1 - g.k0));
  const double ni = (double)wl_ninside_global(wl_grid{g.D, g.nx, g.ny, g.nz, g.k0, g.k1, g.gk, g.gnz});
  hipLaunchKernelGGL(k_shift_norms, grid, dim3(WL_BLOCK), 0, s, g, r, ws.res_d + 0, ni, ws.pa, ws.pm);
  hipLaunchKernelGGL(k_final_sum_max, dim3(1), dim3(WL_BLOCK), 0, s, ws.pa, ws.pm, (int)grid.x, ws.res_d + slot_d, ws.res_f + slot_f);
  WL_LAUNCH_CHECK(); return 0;
}
int gs_sweep(float* eps, const float* r, const float* L, const float* iD, const GridX& g, int k0, hipStream_t s) {
  DSEL(g.D, k_gs_sweep, wl_plane_grid(g, g.k1 - g.k0), dim3(WL_BLOCK), 0, s, g, eps, r, L, iD, k0);
  WL_LAUNCH_CHECK(); return 0;
}
int restrict_(float* a, const GridX& gc, const float* b, const GridX& gf, hipStream_t s) {
  int cx, cy, cz; mask_of(gf, gc, cx, cy, cz);
  DSEL(gc.D, k_restrict, wl_plane_grid(gc, gc.k1 - gc.k0), dim3(WL_BLOCK), 0, s, gc, gf, a, b, cx, cy, cz);
  WL_LAUNCH_CHECK(); return 0;
}
int prolongate(float* a, const GridX& gf, const float* b, const GridX& gc, hipStream_t s) {
  int cx, cy, cz; mask_of(gf, gc, cx, cy, cz);
  DSEL(gf.D, k_prolongate, wl_plane_grid(gf, gf.k1 - gf.k0), dim3(WL_BLOCK), 0, s, gf, gc, a, b, cx, cy, cz);
  WL_LAUNCH_CHECK(); return 0;
}
int prolong_increment(float* r, float* x, float* eps, const float* xc, const float* L, const float* Dg, const GridX& gf, const GridX& gc, float w, bool write_eps, hipStream_t s) {
  int cx, cy, cz; mask_of(gf, gc, cx, cy, cz);
  DSEL(gf.D, k_prolong_increment, wl_plane_grid(gf, gf.k1 - gf.k0), dim3(WL_BLOCK), 0, s, gf, gc, r, x, eps, xc, L, Dg, cx, cy, cz, w, write_eps ? 1 : 0);
  WL_LAUNCH_CHECK(); return 0;
}
int restrictL(float* a, const GridX& gc, const float* b, const GridX& gf, unsigned per, hipStream_t s) {
  int cx, cy, cz; mask_of(gf, gc, cx, cy, cz);
  DSEL(gc.D, k_restrictL, wl_plane_grid(gc, gc.k1 - gc.k0), dim3(WL_BLOCK), 0, s, gc, gf, a, b, cx, cy, cz);
  WL_LAUNCH_CHECK();
  const float zero[3] = {0.f, 0.f, 0.f};
  return bc_vec(a, gc, zero, 0, per, s);   // BC!(a,zero,false,perdir)  :47
}
}  // namespace wl
