// BC!(u, U) for a tuple U (src/core.jl:200-219; no periodic direction, no saveexit) folded into the STORES of the kernel that
// produces u, so that the separate k_bc_vec launch (four per mom_step!, sector-granular x-faces) disappears.
//
// For a tuple U the reference's sequence of face updates reduces to a closed form per ghost/boundary location (k_bc_vec uses the same
// resolution): the component NORMAL to a boundary is U at the indices 1, 2, N of its direction (Julia, ghosts included); every other
// (tangential) coordinate that lies on a ghost layer is clamped into the interior, i.e. the value is that of the nearest interior cell.
// Seen from the producer: an interior cell (i,j,k) owns itself and every ghost location whose clamped coordinates are (i,j,k) — up to
// 8 locations for a corner cell, 1 for a cell away from the boundary — and each location gets, per component, either U or the cell's value.
#pragma once
#include "wl_common.hpp"

#ifdef __HIPCC__
// uo: the (3-component) output array; (i,j,k): LOCAL 0-based coordinates of an INTERIOR cell (single domain: global = local);
// v: the three components the producer computed for it.  Returns after storing the cell and the ghost locations it owns.
__device__ __forceinline__ void wl_bc_fold_store(float* __restrict__ uo, const GridX& g, int i, int j, int k, const float v[3], const float U[3]) {
  const long cs = g.cs;
  const int pi = (i == 1) ? 0 : ((i == g.nx - 2) ? g.nx - 1 : -1);      // the ghost column this cell also fills (−1: none)
  const int pj = (j == 1) ? 0 : ((j == g.ny - 2) ? g.ny - 1 : -1);
  const int pk = (k == 1) ? 0 : ((k == g.nz - 2) ? g.nz - 1 : -1);
  const int ti[2] = {i, pi}, tj[2] = {j, pj}, tk[2] = {k, pk};
#pragma unroll
  for (int c = 0; c < 2; c++) {
    if (tk[c] < 0) continue;
#pragma unroll
    for (int b = 0; b < 2; b++) {
      if (tj[b] < 0) continue;
#pragma unroll
      for (int a = 0; a < 2; a++) {
        if (ti[a] < 0) continue;
        const int I = ti[a], J = tj[b], K = tk[c];
        const long o = (long)I + (long)J * g.sy + (long)K * g.sz;
        uo[o] = (I <= 1 || I == g.nx - 1) ? U[0] : v[0];
        uo[cs + o] = (J <= 1 || J == g.ny - 1) ? U[1] : v[1];
        uo[2 * cs + o] = (K <= 1 || K == g.nz - 1) ? U[2] : v[2];
      }
    }
  }
}
// the same restricted to x and y (the plane K is fixed, the z component is stored as computed): for producers that complete the z ghost
// planes separately (k_bc_zplanes: plane 0 ← plane 1, plane nz−1 ← plane nz−2, normal component U on planes 0, 1, nz−1)
__device__ __forceinline__ void wl_bc_fold_store_xy(float* __restrict__ uo, const GridX& g, int i, int j, int K, const float v[3], const float U[3]) {
  const unsigned cs = (unsigned)g.cs;
  const int pi = (i == 1) ? 0 : ((i == g.nx - 2) ? g.nx - 1 : -1);
  const int pj = (j == 1) ? 0 : ((j == g.ny - 2) ? g.ny - 1 : -1);
  const int ti[2] = {i, pi}, tj[2] = {j, pj};
  const unsigned ko = (unsigned)K * (unsigned)g.sz;
#pragma unroll
  for (int b = 0; b < 2; b++) {
    if (tj[b] < 0) continue;
#pragma unroll
    for (int a = 0; a < 2; a++) {
      if (ti[a] < 0) continue;
      const int I = ti[a], J = tj[b];
      const unsigned o = (unsigned)I + (unsigned)J * (unsigned)g.sy + ko;
      uo[o] = (I <= 1 || I == g.nx - 1) ? U[0] : v[0];
      uo[cs + o] = (J <= 1 || J == g.ny - 1) ? U[1] : v[1];
      uo[2u * cs + o] = v[2];
    }
  }
}
// true when (i,j,k) is an interior cell that neither lies on a Dirichlet face nor owns a ghost location: a plain store is the whole BC
__device__ __forceinline__ bool wl_bc_fold_plain(const GridX& g, int i, int j, int k) {
  return i > 1 && i < g.nx - 2 && j > 1 && j < g.ny - 2 && k > 1 && k < g.nz - 2;
}
#endif
