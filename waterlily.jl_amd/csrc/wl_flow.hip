// Flow-side kernels for gfx950: conv_diff! (gather form), BDIM!, BC!, div, projection, CFL and the
// generic array ops.  Reference semantics: /root/reference/src/Flow.jl, src/core.jl (file:line per kernel).
// Arithmetic order follows the reference statement by statement (-ffp-contract=off).
#include <functional>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "wl_common.hpp"
#include "wl_bcfold.hpp"
#include "wl_conv_cell.hpp"

namespace {

__device__ __forceinline__ bool cell_ij(const GridX& g, long m, int& i, int& j) {
  if (m >= g.sz) return false;
  j = (int)(m / g.nx);
  i = (int)(m - (long)j * g.nx);
  return true;
}
__device__ __forceinline__ bool interior_ij(const GridX& g, int i, int j) { return i >= 1 && i <= g.nx - 2 && j >= 1 && j <= g.ny - 2; }

// ---- generic ------------------------------------------------------------------------------------
__global__ void k_fill(float* __restrict__ a, float v, size_t n) {
  for (size_t q = (size_t)blockIdx.x * WL_BLOCK + threadIdx.x; q < n; q += (size_t)gridDim.x * WL_BLOCK) a[q] = v;
}
__global__ void k_scale(float* __restrict__ a, float s, size_t n) {
  for (size_t q = (size_t)blockIdx.x * WL_BLOCK + threadIdx.x; q < n; q += (size_t)gridDim.x * WL_BLOCK) a[q] = a[q] * s;
}
__global__ void k_divs(float* __restrict__ a, float s, size_t n) {
  for (size_t q = (size_t)blockIdx.x * WL_BLOCK + threadIdx.x; q < n; q += (size_t)gridDim.x * WL_BLOCK) a[q] = a[q] / s;
}
// out = in / s (out ≠ in): the p = x/Δt half of mom_project!'s tail when the velocity half is evaluated by the corrector's loader (wl_convf.hip, PROJ)
__global__ void k_divs_to4(float4* __restrict__ out, const float4* __restrict__ in, float s, size_t n4) {
  for (size_t q = (size_t)blockIdx.x * WL_BLOCK + threadIdx.x; q < n4; q += (size_t)gridDim.x * WL_BLOCK) { const float4 v = in[q]; out[q] = make_float4(v.x / s, v.y / s, v.z / s, v.w / s); }
}
__global__ void k_divs_to(float* __restrict__ out, const float* __restrict__ in, float s, size_t n) {
  for (size_t q = (size_t)blockIdx.x * WL_BLOCK + threadIdx.x; q < n; q += (size_t)gridDim.x * WL_BLOCK) out[q] = in[q] / s;
}
__global__ void k_red_sum(const float* __restrict__ a, size_t n, double* __restrict__ part) {
  double acc = 0.0;
  for (size_t q = (size_t)blockIdx.x * WL_BLOCK + threadIdx.x; q < n; q += (size_t)gridDim.x * WL_BLOCK) acc += (double)a[q];
  acc = block_sum(acc);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
__global__ void k_red_l1_linf(const float* __restrict__ a, size_t n, double* __restrict__ part, float* __restrict__ pmax) {
  double acc = 0.0; float mx = 0.f;
  for (size_t q = (size_t)blockIdx.x * WL_BLOCK + threadIdx.x; q < n; q += (size_t)gridDim.x * WL_BLOCK) { const float v = fabsf(a[q]); acc += (double)v; mx = fmaxf(mx, v); }
  acc = block_sum(acc); mx = block_max(mx);
  if (threadIdx.x == 0) { part[blockIdx.x] = acc; pmax[blockIdx.x] = mx; }
}
__global__ void k_red_max(const float* __restrict__ a, size_t n, float* __restrict__ pmax) {
  float mx = -INFINITY;
  for (size_t q = (size_t)blockIdx.x * WL_BLOCK + threadIdx.x; q < n; q += (size_t)gridDim.x * WL_BLOCK) mx = fmaxf(mx, a[q]);
  mx = block_max(mx);
  if (threadIdx.x == 0) pmax[blockIdx.x] = mx;
}
__global__ void k_red_dot(const float* __restrict__ a, const float* __restrict__ b, size_t n, double* __restrict__ part) {
  double acc = 0.0;
  for (size_t q = (size_t)blockIdx.x * WL_BLOCK + threadIdx.x; q < n; q += (size_t)gridDim.x * WL_BLOCK) acc += (double)a[q] * (double)b[q];
  acc = block_sum(acc);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
__global__ void k_fin_sum(const double* __restrict__ part, int n, double* __restrict__ out) {
  double a = 0.0; for (int q = threadIdx.x; q < n; q += WL_BLOCK) a += part[q];
  a = block_sum(a); if (threadIdx.x == 0) *out = a;
}
__global__ void k_fin_sum_max(const double* __restrict__ part, const float* __restrict__ pmax, int n, double* __restrict__ os, float* __restrict__ om) {
  double a = 0.0; float mx = -INFINITY;
  for (int q = threadIdx.x; q < n; q += WL_BLOCK) { a += part[q]; mx = fmaxf(mx, pmax[q]); }
  a = block_sum(a); mx = block_max(mx);
  if (threadIdx.x == 0) { *os = a; *om = mx; }
}
__global__ void k_fin_max(const float* __restrict__ pmax, int n, float* __restrict__ om) {
  float mx = -INFINITY; for (int q = threadIdx.x; q < n; q += WL_BLOCK) mx = fmaxf(mx, pmax[q]);
  mx = block_max(mx); if (threadIdx.x == 0) *om = mx;
}

template <int D, int SCH, int PER, typename IDX, int FUSE>
#ifndef WL_CD_WAVES
#define WL_CD_WAVES 1
#endif
__global__ void __launch_bounds__(WL_BLOCK, WL_CD_WAVES) k_conv_diff(GridX g, float* __restrict__ r, const float* __restrict__ u, float nu, unsigned per, int kfirst, BdimArgs bd) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  if (!cell_ij(g, m, i, j)) return;
  const bool store = true;
  const int k = (D == 3) ? kfirst + pz : 0;
  const IDX o = (IDX)(m + (long)k * g.sz);
  // z-periodic domain on z-slabs: the ghost planes hold the wrapped neighbours' data (halo exchange, 2 deep), so along z every owned plane is an
  // inner plane — neither wall forms nor wrapped addresses (the values the periodic forms read are exactly the ones in the ghost planes)
  const bool zfree = PER && D == 3 && ((per >> 2) & 1u) && g.nz != g.gnz;
  if (zfree) per &= ~4u;
  const int I[3] = {i + 1, j + 1, (D == 3) ? (zfree ? 3 : g.gk + k + 1) : 2};    // Julia (global) indices
  const int N[3] = {g.nx, g.ny, (D == 3) ? (zfree ? (1 << 28) : g.gnz) : 4};
  const IDX st[3] = {1, (IDX)g.sy, (IDX)g.sz};
  float out[3];
#ifdef WL_CD_INNER
  // Experiment (off): waves whose 64 cells are all ≥ 2 cells away from every boundary take the variant without clamped addresses,
  // boundary flux forms and masked accumulation — a wave-uniform branch, ≈520 instead of 806 VALU instructions for ≈73 % of the
  // waves at 512³.  Measured: no change in kernel time (the 69 loads per cell are the same) — see DESIGN.md §4.
  bool inner = true;
#pragma unroll
  for (int c = 0; c < D; c++) inner = inner && I[c] >= 3 && I[c] <= N[c] - 2;
  if (PER == 0 && __all(inner)) cd_cell<D, SCH, PER, IDX, 1, 0>(g, u, o, I, N, st, nu, per, nullptr, out);
  else
#endif
  cd_cell<D, SCH, PER, IDX, 0, 0>(g, u, o, I, N, st, nu, per, nullptr, out);
  if (FUSE == 2) {
    // Flow with a body.  Far from it (μ₁ ≡ 0, V ≡ 0 in this workgroup's cells of this plane) BDIM! degenerates to the NoBody form and is
    // applied here; near it the raw r goes to f for the two-pass BDIM! kernels, which run on the near workgroups only.  Far
    // workgroups keep f = u⁰+Δt·r only where a near cell's μddn reads it (needf).
    bool in = interior_ij(g, i, j);
    if (D == 3) in = in && k >= g.k0 && k < g.k1;
    const long mi = (long)k * bd.nbm + (m / WL_BLOCK);
    if (bd.near[mi]) {
#pragma unroll
      for (int a = 0; a < D; a++) r[(long)a * g.cs + o] = out[a];
      return;
    }
    const bool keepf = bd.store_all || bd.needf[mi];
    const bool m0load = bd.m0var[mi] != 0;
#pragma unroll
    for (int a = 0; a < D; a++) {
      const long oa = (long)a * g.cs + o;
      const float fn = bd.u0[oa] + bd.dt * out[a] - 0.f;
      if (keepf) r[oa] = fn;
      if (in) {
        const float m0 = m0load ? bd.mu0[oa] : wl::wl_cl_coef(I[a], N[a], 1.f);       // verified per workgroup by k_body_mask2
        const float xx = (0.f / 2 + 0.f) + m0 * fn;
        float un = (bd.pre == 0.f) ? xx : (u[oa] * bd.pre + xx);
        if (bd.scale_after) un = un * bd.post;
        bd.uout[oa] = un;
      }
    }
    return;
  }
  if (FUSE) {
    bool in = interior_ij(g, i, j);
    if (D == 3) in = in && k >= g.k0 && k < g.k1;
#pragma unroll
    for (int a = 0; a < D; a++) {
      if (!store) break;
      const long oa = (long)a * g.cs + o;
      const float fn = bd.u0[oa] + bd.dt * out[a] - 0.f;
      if (r) r[oa] = fn;
      if (in) {
        const float m0 = bd.cl_on ? wl::wl_cl_coef(I[a], N[a], bd.cl_c[a]) : bd.mu0[oa];     // μ₀ on a verified NoBody field
        const float xx = (0.f / 2 + 0.f) + m0 * fn;
        float un = (bd.pre == 0.f) ? xx : (u[oa] * bd.pre + xx);
        if (bd.scale_after) un = un * bd.post;
        bd.uout[oa] = un;
      }
    }
    return;
  }
  if (!store) return;
#pragma unroll
  for (int a = 0; a < D; a++) r[(long)a * g.cs + o] = out[a];
}
// Quirk Q1: Φ≡σ keeps the fluxes of the LAST pass that covered a cell; interior values are
// overwritten by div/flux_out later, so only upper-ghost cells matter for CFL's maximum(σ).  Reproduced by this
// small kernel over the three upper ghost planes (blockIdx.y = direction whose index is Ng).
template <int D, int SCH>
__global__ void k_conv_q1(GridX g, float* __restrict__ Phi, const float* __restrict__ u, float nu, unsigned per) {
  const int d = blockIdx.y;
  const int N[3] = {g.nx, g.ny, (D == 3) ? g.gnz : 1};
  const long st[3] = {1, g.sy, g.sz};
  const int d1 = (d == 0) ? 1 : 0, d2 = (d == 2) ? 1 : 2;
  const int n1 = (d1 == 0) ? g.nx : g.ny;
  const long cnt = (long)n1 * ((D == 3) ? ((d2 == 1) ? g.ny : g.nz) : 1);
  const long q = (long)blockIdx.x * WL_BLOCK + threadIdx.x;
  if (q >= cnt) return;
  int loc[3] = {0, 0, 0};
  loc[d1] = (int)(q % n1);
  if (D == 3) loc[d2] = (int)(q / n1);
  if (d == 2) { const int kl = N[2] - 1 - g.gk; if (kl < 0 || kl >= g.nz) return; loc[2] = kl; } else loc[d] = N[d] - 1;
  const int I[3] = {loc[0] + 1, loc[1] + 1, (D == 3) ? g.gk + loc[2] + 1 : 2};
  if (D == 3 && d != 2) { const bool owned = (loc[2] >= g.k0 && loc[2] < g.k1) || (I[2] == N[2] && !(((per >> 2) & 1u) && g.nz != g.gnz)); if (!owned) return; }
  for (int c = 0; c < D; c++) if (I[c] < 2) return;
  const long o = (long)loc[0] + (long)loc[1] * g.sy + (long)loc[2] * g.sz;
  const int a = D - 1;
  const float* __restrict__ f = u + (long)a * g.cs;
  const bool zfree = D == 3 && ((per >> 2) & 1u) && g.nz != g.gnz;    // z-periodic slabs: along z every owned plane is an inner plane (wrapped halos)
  if (zfree && d == 2) return;                                        // (no physical ghost plane in z on any rank)
  for (int b = D - 1; b >= 0; b--) {
    const bool pb = (per >> b) & 1u;
    const bool zin = zfree && b == 2;
    const bool covered = zin || (I[b] >= 3 && I[b] <= N[b] - 1) || (pb && I[b] == 2);
    if (!covered) continue;
    const float* __restrict__ ub = u + (long)b * g.cs;
    Phi[o] = (I[b] == 2 && !zin) ? flux_lowerP<SCH>(f, ub, o, st[b], st[a], nu, o + (long)(N[b] - 4) * st[b]) : flux_inner<SCH>(f, ub, o, st[b], st[a], nu);
    break;
  }
}

// accelerate!(r,t,g,U) for accelerations that are uniform in space: r[I,i] += a_i on ALL cells   src/Flow.jl:69-73
// (the host evaluates g(i,t)+dU(i,t)/dt; position-dependent g/uBC are not a device path)
__global__ void k_accelerate(float* __restrict__ r, long cs, long n, float a0, float a1, float a2) {
  for (long q = (long)blockIdx.x * WL_BLOCK + threadIdx.x; q < n; q += (long)gridDim.x * WL_BLOCK) {
    const int c = (int)(q / cs);
    r[q] += (c == 0) ? a0 : (c == 1 ? a1 : a2);
  }
}
// BDIM!  src/Flow.jl:176-180 (+ scale_u! :211-214 folded in through pre/post)
// pass A: f = u⁰ + dt f − V on ALL cells
__global__ void k_bdim_f(GridX g, float* __restrict__ f, const float* __restrict__ u0, const float* __restrict__ V, float dt, long n) {
  for (long q = (long)blockIdx.x * WL_BLOCK + threadIdx.x; q < n; q += (long)gridDim.x * WL_BLOCK) {
    const float v = V ? V[q] : 0.f;
    f[q] = u0[q] + dt * f[q] - v;
  }
}
// pass B: u[I,i] = (u*pre + (μddn(I,μ₁,f) + V + μ₀ f)) * post   on the interior
// far[b] = 1: every cell of workgroup b has μ₁ ≡ 0 and V ≡ 0 (far from the body) — computed once per measure!/update!.  Those
// workgroups skip the 48 B/cell of μ₁ and V and the six neighbour values of f; the expression degenerates to the same bits
// ((±0)/2 + (+0)) + μ₀·f = μ₀·f.
template <int D>
__global__ void k_body_mask(GridX g, const float* __restrict__ V, const float* __restrict__ mu1, unsigned char* __restrict__ far) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  int bad = 0;
  if (cell_ij(g, m, i, j) && interior_ij(g, i, j)) {
    const long o = m + (long)(g.k0 + pz) * g.sz;
    for (int a = 0; a < D; a++) { if (V[(long)a * g.cs + o] != 0.f) bad = 1; for (int b = 0; b < D; b++) if (mu1[(long)(a + b * D) * g.cs + o] != 0.f) bad = 1; }
  }
  bad = __syncthreads_or(bad);
  if (threadIdx.x == 0) far[blockIdx.x] = bad ? 0 : 1;
}
template <int D>
__global__ void k_bdim_u(GridX g, float* __restrict__ u, const float* __restrict__ f, const float* __restrict__ V, const float* __restrict__ mu0, const float* __restrict__ mu1,
                         float pre, float post, int scale_after, const unsigned char* __restrict__ far) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  if (!cell_ij(g, m, i, j) || !interior_ij(g, i, j)) return;
  const long o = m + (long)(g.k0 + pz) * g.sz;
  const long st[3] = {1, g.sy, g.sz};
  if (far && far[blockIdx.x]) {
    for (int a = 0; a < D; a++) {
      const long oa = (long)a * g.cs + o;
      const float x = (0.f / 2 + 0.f) + mu0[oa] * f[oa];
      float un = (pre == 0.f) ? x : (u[oa] * pre + x);
      if (scale_after) un = un * post;
      u[oa] = un;
    }
    return;
  }
  for (int a = 0; a < D; a++) {
    const long oa = (long)a * g.cs + o;
    float s = 0.f;
    if (mu1) {
      for (int b = 0; b < D; b++) s += mu1[(long)(a + b * D) * g.cs + o] * (f[oa + st[b]] - f[oa - st[b]]);     // μddn :20-26
    }
    const float x = (s / 2 + (V ? V[oa] : 0.f)) + mu0[oa] * f[oa];
    float un = (pre == 0.f) ? x : (u[oa] * pre + x);
    if (scale_after) un = un * post;
    u[oa] = un;
  }
}
// ---- body-aware split of conv_diff!+BDIM! (k_conv_diff<…,FUSE=2>): masks indexed by (plane k, in-plane workgroup m/256) -------------
template <int D>
__global__ void k_body_mask2(GridX g, const float* __restrict__ V, const float* __restrict__ mu1, const float* __restrict__ mu0, unsigned char* __restrict__ near,
                             unsigned char* __restrict__ needf, unsigned char* __restrict__ m0var, int nbm) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  if (!cell_ij(g, m, i, j)) return;
  const int k = pz;
  const long o = m + (long)k * g.sz;
  {
    const int I[3] = {i + 1, j + 1, (D == 3) ? g.gk + k + 1 : 2}, N[3] = {g.nx, g.ny, (D == 3) ? g.gnz : 4};
    bool var = false;
    for (int a = 0; a < D; a++) if (mu0[(long)a * g.cs + o] != wl::wl_cl_coef(I[a], N[a], 1.f)) var = true;
    if (var) m0var[(long)k * nbm + m / WL_BLOCK] = 1;
  }
  bool nz1 = false, nzv = false;
  for (int a = 0; a < D; a++) { if (V[(long)a * g.cs + o] != 0.f) nzv = true; for (int b = 0; b < D; b++) if (mu1[(long)(a + b * D) * g.cs + o] != 0.f) nz1 = true; }
  if (nz1 || nzv) near[(long)k * nbm + m / WL_BLOCK] = 1;
  if (nz1) {   // μddn of this cell reads f at its six neighbours
    const long st[3] = {1, g.sy, g.sz};
    for (int b = 0; b < D; b++) for (int sg = -1; sg <= 1; sg += 2) {
      long mm = m; int kk = k;
      if (b == 2) kk += sg; else mm += sg * st[b];
      if (kk < 0 || kk >= g.nz || mm < 0 || mm >= g.sz) continue;
      needf[(long)kk * nbm + mm / WL_BLOCK] = 1;
    }
  }
}
// pass A on the near workgroups: f = u⁰ + Δt·f − V (every cell of the workgroup)
// (both near-workgroup kernels are launched over the bounding box of the near workgroups only: blockIdx.x ↔ in-plane workgroup b0+x,
//  blockIdx.y ↔ plane kb+y — dispatching the whole grid for a body that fills 2 % of it would cost more than the work)
template <int D>
__global__ void k_bdim_f_m(GridX g, float* __restrict__ f, const float* __restrict__ u0, const float* __restrict__ V, float dt, const unsigned char* __restrict__ near, int nbm, int b0, int kb) {
  int i, j;
  const long m = (long)(b0 + (int)blockIdx.x) * WL_BLOCK + threadIdx.x;
  const int pz = kb + (int)blockIdx.y;
  if (!cell_ij(g, m, i, j)) return;
  if (!near[(long)pz * nbm + m / WL_BLOCK]) return;
  const long o = m + (long)pz * g.sz;
  for (int a = 0; a < D; a++) { const long oa = (long)a * g.cs + o; f[oa] = u0[oa] + dt * f[oa] - V[oa]; }
}
// pass B on the near workgroups: u_out[I,i] = (u_in·pre + (μddn(I,μ₁,f) + V + μ₀ f))·post   on the interior
template <int D>
__global__ void k_bdim_u_m(GridX g, float* __restrict__ uout, const float* __restrict__ uin, const float* __restrict__ f, const float* __restrict__ V, const float* __restrict__ mu0,
                           const float* __restrict__ mu1, float pre, float post, int scale_after, const unsigned char* __restrict__ near, int nbm, int b0, int kb) {
  int i, j;
  const long m = (long)(b0 + (int)blockIdx.x) * WL_BLOCK + threadIdx.x;
  const int k = kb + (int)blockIdx.y;
  if (!cell_ij(g, m, i, j) || !interior_ij(g, i, j)) return;
  if (k < g.k0 || k >= g.k1) return;
  if (!near[(long)k * nbm + m / WL_BLOCK]) return;
  const long o = m + (long)k * g.sz;
  const long st[3] = {1, g.sy, g.sz};
  for (int a = 0; a < D; a++) {
    const long oa = (long)a * g.cs + o;
    float s = 0.f;
    for (int b = 0; b < D; b++) s += mu1[(long)(a + b * D) * g.cs + o] * (f[oa + st[b]] - f[oa - st[b]]);     // μddn :20-26
    const float x = (s / 2 + V[oa]) + mu0[oa] * f[oa];
    float un = (pre == 0.f) ? x : (uin[oa] * pre + x);
    if (scale_after) un = un * post;
    uout[oa] = un;
  }
}
// NoBody fast path (μ₁≡0, V≡0): both passes in one kernel over all cells
template <int D>
__global__ void k_bdim_nobody(GridX g, float* __restrict__ u, const float* __restrict__ u0, float* __restrict__ f, const float* __restrict__ mu0, float dt, float pre, float post,
                              int scale_after) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  if (!cell_ij(g, m, i, j)) return;
  const int k = (int)pz;
  const long o = m + (long)k * g.sz;
  bool in = interior_ij(g, i, j);
  if (D == 3) in = in && k >= g.k0 && k < g.k1;
  for (int a = 0; a < D; a++) {
    const long oa = (long)a * g.cs + o;
    const float fn = u0[oa] + dt * f[oa] - 0.f;
    f[oa] = fn;
    if (in) {
      const float x = (0.f / 2 + 0.f) + mu0[oa] * fn;
      float un = (pre == 0.f) ? x : (u[oa] * pre + x);
      if (scale_after) un = un * post;
      u[oa] = un;
    }
  }
}
template <int D>
__global__ void k_scale_u(GridX g, float* __restrict__ u, float sc) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  if (!cell_ij(g, m, i, j) || !interior_ij(g, i, j)) return;
  const long o = m + (long)(g.k0 + pz) * g.sz;
  for (int a = 0; a < D; a++) u[(long)a * g.cs + o] *= sc;
}

// z = div(I,u)  src/Flow.jl:13-19,225 ; optional fused x *= dt over ALL cells (x may be NULL)
template <int D>
__global__ void k_div(GridX g, float* __restrict__ z, float* __restrict__ x, const float* __restrict__ u, float dt) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  if (!cell_ij(g, m, i, j)) return;
  const int k = (int)pz;
  const long o = m + (long)k * g.sz;
  if (x) x[o] = x[o] * dt;
  bool in = interior_ij(g, i, j);
  if (D == 3) in = in && k >= g.k0 && k < g.k1;
  if (!in) return;
  float s = 0.f;
  s += u[o + 1] - u[o];
  s += u[g.cs + o + g.sy] - u[g.cs + o];
  if (D == 3) s += u[2 * g.cs + o + g.sz] - u[2 * g.cs + o];
  z[o] = s;
}
// u[I,i] -= L[I,i]*(x[I]-x[I-δᵢ])   src/Flow.jl:227-229
template <int D>
__global__ void k_project(GridX g, float* __restrict__ u, const float* __restrict__ L, const float* __restrict__ x) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  if (!cell_ij(g, m, i, j) || !interior_ij(g, i, j)) return;
  const long o = m + (long)(g.k0 + pz) * g.sz;
  const float xc = x[o];
  u[o] -= L[o] * (xc - x[o - 1]);
  u[g.cs + o] -= L[g.cs + o] * (xc - x[o - g.sy]);
  if (D == 3) u[2 * g.cs + o] -= L[2 * g.cs + o] * (xc - x[o - g.sz]);
}
// CFL: σ = flux_out on the interior; block max over ALL cells of σ (ghost planes keep stale Φ — quirk Q1)  src/Flow.jl:234-244
template <int D>
__global__ void k_cfl(GridX g, const float* __restrict__ u, float* __restrict__ sigma, float* __restrict__ pmax, int kfirst, int klast, int zchunk) {
  int i, j; long m; int pz;
  wl_tile(g, m, pz);
  float mx = -INFINITY;
  if (cell_ij(g, m, i, j)) {
    const bool inij = interior_ij(g, i, j);
    const int ks = kfirst + pz * zchunk, ke = (ks + zchunk < klast) ? ks + zchunk : klast;   // contiguous planes: u_z[k+1] becomes u_z[k]
    long o = m + (long)ks * g.sz;
    float uzk = (D == 3 && inij && ks < ke) ? u[2 * g.cs + o] : 0.f;
    for (int k = ks; k < ke; k++, o += g.sz) {
      const float uzkp = (D == 3 && inij && k + 1 < g.nz) ? u[2 * g.cs + o + g.sz] : 0.f;
      bool in = inij;
      if (D == 3) in = in && k >= g.k0 && k < g.k1;
      float s;
      if (in) {
        s = 0.f;
        s += (fmaxf(0.f, u[o + 1]) + fmaxf(0.f, -u[o]));
        s += (fmaxf(0.f, u[g.cs + o + g.sy]) + fmaxf(0.f, -u[g.cs + o]));
        if (D == 3) s += (fmaxf(0.f, uzkp) + fmaxf(0.f, -uzk));
        sigma[o] = s;
      } else s = sigma[o];
      mx = fmaxf(mx, s);
      uzk = uzkp;
    }
  }
  mx = block_max(mx);
  if (threadIdx.x == 0) pmax[blockIdx.x] = mx;
}

// BC!(a,U,saveexit,perdir) for tuple U — all faces and components in ONE launch.   src/core.jl:200-219
// The reference applies (i outer, j inner) face updates sequentially; the value a cell ends with is set by the
// LAST direction j that touches it, fed from a source already processed by the lower directions.  resolve()
// walks j = D..1 accordingly; the final source cell lies on no BC plane, so nothing it reads is written here.
template <int D>
__device__ __forceinline__ bool bc_touched(const int* I, const int* N, int a, int saveexit, unsigned per) {
  for (int b = 0; b < D; b++) {
    const bool pb = (per >> b) & 1u;
    if (pb) { if (I[b] == 1 || I[b] == N[b]) return true; }
    else if (a == b) { if (I[b] == 1 || I[b] == 2 || (I[b] == N[b] && !(saveexit && a == 0))) return true; }
    else { if (I[b] == 1 || I[b] == N[b]) return true; }
  }
  return false;
}
template <int D>
__global__ void k_bc_vec(GridX g, float* __restrict__ a_, float U0, float U1, float U2, int saveexit, unsigned per, int zwalls) {
  // pz = plane id: dir d = id/3, which = id%3 -> Julia index {1,2,N_d}
  const int pid = blockIdx.y, d = pid / 3, which = pid % 3;
  const int N[3] = {g.nx, g.ny, (D == 3) ? g.gnz : 1};
  const float U[3] = {U0, U1, U2};
  // enumerate the plane: the two other dims
  const int d1 = (d == 0) ? 1 : 0, d2 = (d == 2) ? 1 : 2;
  const int n1 = (d1 == 0) ? g.nx : g.ny;
  const long cnt = (long)n1 * ((D == 3) ? ((d2 == 1) ? g.ny : g.nz) : 1);
  const long q = (long)blockIdx.x * WL_BLOCK + threadIdx.x;
  if (q >= cnt) return;
  int loc[3] = {0, 0, 0};   // LOCAL 0-based coords
  loc[d1] = (int)(q % n1);
  if (D == 3) loc[d2] = (int)(q / n1);
  int Id = (which == 0) ? 1 : (which == 1 ? 2 : N[d]);
  int I[3];                 // Julia GLOBAL indices
  if (d == 2) { // z plane: only on ranks that hold that physical plane
    const int kl = Id - 1 - g.gk; if (kl < 0 || kl >= g.nz) return; loc[2] = kl;
    if (!zwalls) return;
  } else loc[d] = Id - 1;
  I[0] = loc[0] + 1; I[1] = loc[1] + 1; I[2] = (D == 3) ? g.gk + loc[2] + 1 : 1;
  if (D == 3 && d != 2) {   // x/y planes: skip halo planes owned by a neighbour rank (filled by the halo exchange)
    const int K = I[2];
    const bool owned = (loc[2] >= g.k0 && loc[2] < g.k1) || (zwalls && (K == 1 || K == N[2]));   // (z-periodic slabs: planes 1 and N are halo planes too)
    if (!owned) return;
  }
  const long o = (long)loc[0] + (long)loc[1] * g.sy + (long)loc[2] * g.sz;
  for (int a = 0; a < D; a++) {
    if (!bc_touched<D>(I, N, a, saveexit, per)) continue;
    int J[3] = {I[0], I[1], I[2]};
    bool dirichlet = false;
    for (int b = D - 1; b >= 0; b--) {
      const bool pb = (per >> b) & 1u;
      if (pb) { if (J[b] == 1) J[b] = N[b] - 1; else if (J[b] == N[b]) J[b] = 2; }
      else if (a == b) { if (J[b] == 1 || J[b] == 2 || (J[b] == N[b] && !(saveexit && a == 0))) { dirichlet = true; break; } }
      else { if (J[b] == 1) J[b] = 2; else if (J[b] == N[b]) J[b] = N[b] - 1; }
    }
    float v;
    if (dirichlet) v = U[a];
    else {
      const long os = (long)(J[0] - 1) + (long)(J[1] - 1) * g.sy + ((D == 3) ? (long)(J[2] - 1 - g.gk) * g.sz : 0);
      v = a_[(long)a * g.cs + os];
    }
    a_[(long)a * g.cs + o] = v;
  }
}
// BC!(a,uBC::Function,saveexit,perdir,t) with the boundary values tabulated by the host: Ub has the shape of `a` and holds
// uBC(i,loc(i,I),t) on the two outermost layers of every non-periodic direction (nothing else is read).  The reference applies the
// (i,j) face updates sequentially; a Neumann update is  a[I] = (uBC(I) + a[S]) - uBC(S)  with S the inward neighbour, and at edges
// and corners a[S] may itself be the product of an earlier direction.  The chain is collected from the last direction inwards
// (as in k_bc_vec) and then evaluated from its innermost cell outwards — the order in which the reference produces the values.
template <int D>
__global__ void k_bc_vec_fn(GridX g, float* __restrict__ a_, const float* __restrict__ Ub, int saveexit, unsigned per, int zwalls) {
  const int pid = blockIdx.y, d = pid / 3, which = pid % 3;
  const int N[3] = {g.nx, g.ny, (D == 3) ? g.gnz : 1};
  const int d1 = (d == 0) ? 1 : 0, d2 = (d == 2) ? 1 : 2;
  const int n1 = (d1 == 0) ? g.nx : g.ny;
  const long cnt = (long)n1 * ((D == 3) ? ((d2 == 1) ? g.ny : g.nz) : 1);
  const long q = (long)blockIdx.x * WL_BLOCK + threadIdx.x;
  if (q >= cnt) return;
  int loc[3] = {0, 0, 0};
  loc[d1] = (int)(q % n1);
  if (D == 3) loc[d2] = (int)(q / n1);
  const int Id = (which == 0) ? 1 : (which == 1 ? 2 : N[d]);
  int I[3];
  if (d == 2) { const int kl = Id - 1 - g.gk; if (kl < 0 || kl >= g.nz) return; loc[2] = kl; if (!zwalls) return; }
  else loc[d] = Id - 1;
  I[0] = loc[0] + 1; I[1] = loc[1] + 1; I[2] = (D == 3) ? g.gk + loc[2] + 1 : 1;
  if (D == 3 && d != 2) {
    const int K = I[2];
    const bool owned = (loc[2] >= g.k0 && loc[2] < g.k1) || (zwalls && (K == 1 || K == N[2]));   // (z-periodic slabs: planes 1 and N are halo planes too)
    if (!owned) return;
  }
  auto off = [&](const int* J) -> long { return (long)(J[0] - 1) + (long)(J[1] - 1) * g.sy + ((D == 3) ? (long)(J[2] - 1 - g.gk) * g.sz : 0); };
  for (int a = 0; a < D; a++) {
    if (!bc_touched<D>(I, N, a, saveexit, per)) continue;
    long chain[4]; bool neu[4]; int nc = 0;        // cells from I inwards; neu[q]: step q→q+1 is a Neumann (function) step, else a periodic copy
    int J[3] = {I[0], I[1], I[2]};
    chain[0] = off(J);
    bool dirichlet = false;
    for (int b = D - 1; b >= 0; b--) {
      const bool pb = (per >> b) & 1u;
      int moved = 0;
      if (pb) { if (J[b] == 1) { J[b] = N[b] - 1; moved = 1; } else if (J[b] == N[b]) { J[b] = 2; moved = 1; } }
      else if (a == b) { if (J[b] == 1 || J[b] == 2 || (J[b] == N[b] && !(saveexit && a == 0))) { dirichlet = true; break; } }
      else { if (J[b] == 1) { J[b] = 2; moved = 2; } else if (J[b] == N[b]) { J[b] = N[b] - 1; moved = 2; } }
      if (moved) { neu[nc] = moved == 2; nc++; chain[nc] = off(J); }
    }
    const float* __restrict__ Ua = Ub + (long)a * g.cs;
    float v = dirichlet ? Ua[chain[nc]] : a_[(long)a * g.cs + chain[nc]];
    for (int qk = nc - 1; qk >= 0; qk--) if (neu[qk]) v = (Ua[chain[qk]] + v) - Ua[chain[qk + 1]];
    a_[(long)a * g.cs + chain[0]] = v;
  }
}
// MeanFlow update!  src/Metrics.jl:236-248:  P = ε·p + (1−ε)·P ;  U = ε·u + (1−ε)·U ;  UU[I,i,j] = ε·(u[I,i]·u[I,j]) + (1−ε)·UU[I,i,j]  (all cells)
__global__ void k_meanflow(float* __restrict__ P, float* __restrict__ U, float* __restrict__ UU, const float* __restrict__ p, const float* __restrict__ u, long cs, int D, float e) {
  const float one_m = 1 - e;
  for (long q = (long)blockIdx.x * WL_BLOCK + threadIdx.x; q < cs; q += (long)gridDim.x * WL_BLOCK) {
    P[q] = e * p[q] + one_m * P[q];
    float uv[3];
    for (int i = 0; i < D; i++) { uv[i] = u[(long)i * cs + q]; U[(long)i * cs + q] = e * uv[i] + one_m * U[(long)i * cs + q]; }
    if (UU) for (int j = 0; j < D; j++) for (int i = 0; i < D; i++) { const long o = (long)(i + j * D) * cs + q; UU[o] = e * (uv[i] * uv[j]) + one_m * UU[o]; }
  }
}
// uu!(τ,a)  :250-252:  τ[I,i,j] = UU[I,i,j] − U[I,i]·U[I,j]
__global__ void k_meanflow_uu(float* __restrict__ tau, const float* __restrict__ UU, const float* __restrict__ U, long cs, int D) {
  for (long q = (long)blockIdx.x * WL_BLOCK + threadIdx.x; q < cs; q += (long)gridDim.x * WL_BLOCK)
    for (int j = 0; j < D; j++) for (int i = 0; i < D; i++) { const long o = (long)(i + j * D) * cs + q; tau[o] = UU[o] - U[(long)i * cs + q] * U[(long)j * cs + q]; }
}
__global__ void k_add_field(float* __restrict__ r, const float* __restrict__ gfield, long n) {
  for (long q = (long)blockIdx.x * WL_BLOCK + threadIdx.x; q < n; q += (long)gridDim.x * WL_BLOCK) r[q] += gfield[q];
}
// perBC!(a,perdir) for a scalar   src/core.jl:239-243  (same last-direction-wins resolution)
template <int D>
__global__ void k_bc_per_scalar(GridX g, float* __restrict__ a_, unsigned per) {
  const int pid = blockIdx.y, d = pid / 2, which = pid % 2;
  if (!((per >> d) & 1u)) return;
  const int N[3] = {g.nx, g.ny, (D == 3) ? g.gnz : 1};
  const int d1 = (d == 0) ? 1 : 0, d2 = (d == 2) ? 1 : 2;
  const int n1 = (d1 == 0) ? g.nx : g.ny;
  const long cnt = (long)n1 * ((D == 3) ? ((d2 == 1) ? g.ny : g.nz) : 1);
  const long q = (long)blockIdx.x * WL_BLOCK + threadIdx.x;
  if (q >= cnt) return;
  int loc[3] = {0, 0, 0};
  loc[d1] = (int)(q % n1);
  if (D == 3) loc[d2] = (int)(q / n1);
  const int Id = which == 0 ? 1 : N[d];
  if (d == 2) { const int kl = Id - 1 - g.gk; if (kl < 0 || kl >= g.nz) return; loc[2] = kl; } else loc[d] = Id - 1;
  int J[3] = {loc[0] + 1, loc[1] + 1, (D == 3) ? g.gk + loc[2] + 1 : 1};
  const long o = (long)loc[0] + (long)loc[1] * g.sy + (long)loc[2] * g.sz;
  for (int b = D - 1; b >= 0; b--) {
    if (!((per >> b) & 1u)) continue;
    if (J[b] == 1) J[b] = N[b] - 1; else if (J[b] == N[b]) J[b] = 2;
  }
  const long os = (long)(J[0] - 1) + (long)(J[1] - 1) * g.sy + ((D == 3) ? (long)(J[2] - 1 - g.gk) * g.sz : 0);
  a_[o] = a_[os];
}

inline unsigned grid1d(size_t n) { size_t b = (n + WL_BLOCK - 1) / WL_BLOCK; if (b > 4096) b = 4096; if (b < 1) b = 1; return (unsigned)b; }
}  // namespace

#define DSEL(D, KERN, ...)                                                           \
  do { if ((D) == 3) hipLaunchKernelGGL(KERN<3>, __VA_ARGS__); else hipLaunchKernelGGL(KERN<2>, __VA_ARGS__); } while (0)

namespace wl {
int fill(float* a, float v, size_t n, hipStream_t s) {
  if (v == 0.f) { WL_HIP(hipMemsetAsync(a, 0, n * sizeof(float), s)); return 0; }
  hipLaunchKernelGGL(k_fill, dim3(grid1d(n)), dim3(WL_BLOCK), 0, s, a, v, n); WL_LAUNCH_CHECK(); return 0;
}
int scale(float* a, float sc, size_t n, hipStream_t s) { hipLaunchKernelGGL(k_scale, dim3(grid1d(n)), dim3(WL_BLOCK), 0, s, a, sc, n); WL_LAUNCH_CHECK(); return 0; }
int div_scalar_to(float* out, const float* in, float sc, size_t n, hipStream_t s) {
  if (n % 4 == 0 && (((size_t)out | (size_t)in) & 15) == 0) hipLaunchKernelGGL(k_divs_to4, dim3(grid1d(n / 4)), dim3(WL_BLOCK), 0, s, (float4*)out, (const float4*)in, sc, n / 4);
  else hipLaunchKernelGGL(k_divs_to, dim3(grid1d(n)), dim3(WL_BLOCK), 0, s, out, in, sc, n);
  WL_LAUNCH_CHECK(); return 0;
}
int div_scalar(float* a, float sc, size_t n, hipStream_t s) { hipLaunchKernelGGL(k_divs, dim3(grid1d(n)), dim3(WL_BLOCK), 0, s, a, sc, n); WL_LAUNCH_CHECK(); return 0; }
int sum_dev(const float* a, size_t n, const RedWs& ws, int slot, hipStream_t s) {
  const unsigned nb = grid1d(n);
  hipLaunchKernelGGL(k_red_sum, dim3(nb), dim3(WL_BLOCK), 0, s, a, n, ws.pa);
  hipLaunchKernelGGL(k_fin_sum, dim3(1), dim3(WL_BLOCK), 0, s, ws.pa, (int)nb, ws.res_d + slot);
  WL_LAUNCH_CHECK(); return 0;
}
int l1_linf_dev(const float* a, size_t n, const RedWs& ws, int slot_d, int slot_f, hipStream_t s) {
  const unsigned nb = grid1d(n);
  hipLaunchKernelGGL(k_red_l1_linf, dim3(nb), dim3(WL_BLOCK), 0, s, a, n, ws.pa, ws.pm);
  hipLaunchKernelGGL(k_fin_sum_max, dim3(1), dim3(WL_BLOCK), 0, s, ws.pa, ws.pm, (int)nb, ws.res_d + slot_d, ws.res_f + slot_f);
  WL_LAUNCH_CHECK(); return 0;
}
int max_dev(const float* a, size_t n, const RedWs& ws, int slot_f, hipStream_t s) {
  const unsigned nb = grid1d(n);
  hipLaunchKernelGGL(k_red_max, dim3(nb), dim3(WL_BLOCK), 0, s, a, n, ws.pm);
  hipLaunchKernelGGL(k_fin_max, dim3(1), dim3(WL_BLOCK), 0, s, ws.pm, (int)nb, ws.res_f + slot_f);
  WL_LAUNCH_CHECK(); return 0;
}
int dot_dev(const float* a, const float* b, size_t n, const RedWs& ws, int slot, hipStream_t s) {
  const unsigned nb = grid1d(n);
  hipLaunchKernelGGL(k_red_dot, dim3(nb), dim3(WL_BLOCK), 0, s, a, b, n, ws.pa);
  hipLaunchKernelGGL(k_fin_sum, dim3(1), dim3(WL_BLOCK), 0, s, ws.pa, (int)nb, ws.res_d + slot);
  WL_LAUNCH_CHECK(); return 0;
}
// The pinned staging scalars are process-wide: one reader at a time (handles driven from different host threads / streams serialise here)
std::mutex& wl_read_mutex() { static std::mutex m; return m; }
int read_results(const RedWs& ws, double* hd, int nd, float* hf, int nf, hipStream_t s) {
  std::lock_guard<std::mutex> lock(wl_read_mutex());
  WlCtx& c = wl_ctx();
  if (nd > 0 && nf > 0 && (const char*)ws.res_f == (const char*)ws.res_d + 64 && nd <= 8 && nf <= 16) {
    WL_HIP(hipMemcpyAsync(c.h_d, ws.res_d, 64 + sizeof(float) * (size_t)nf, hipMemcpyDeviceToHost, s));      // one blit instead of two (≈5 µs each)
  } else {
    if (nd > 0) WL_HIP(hipMemcpyAsync(c.h_d, ws.res_d, sizeof(double) * (size_t)nd, hipMemcpyDeviceToHost, s));
    if (nf > 0) WL_HIP(hipMemcpyAsync(c.h_f, ws.res_f, sizeof(float) * (size_t)nf, hipMemcpyDeviceToHost, s));
  }
  WL_HIP(hipStreamSynchronize(s));
  for (int q = 0; q < nd; q++) hd[q] = c.h_d[q];
  for (int q = 0; q < nf; q++) hf[q] = c.h_f[q];
  return 0;
}

// the same read-back, with more work queued on the stream BEHIND the copy before the host waits — for the copy only (an event), not for that work
int read_results_overlapped(const RedWs& ws, double* hd, int nd, float* hf, int nf, hipStream_t s, hipEvent_t copied, const std::function<int()>& queue_behind) {
  std::lock_guard<std::mutex> lock(wl_read_mutex());
  WlCtx& c = wl_ctx();
  if (!(nd > 0 && nf > 0 && (const char*)ws.res_f == (const char*)ws.res_d + 64 && nd <= 8 && nf <= 16)) { wl_set_error("read_results_overlapped: layout"); return WL_EINVAL; }
  WL_HIP(hipMemcpyAsync(c.h_d, ws.res_d, 64 + sizeof(float) * (size_t)nf, hipMemcpyDeviceToHost, s));
  WL_HIP(hipEventRecord(copied, s));
  const int rc = queue_behind();
  WL_HIP(hipEventSynchronize(copied));
  for (int q = 0; q < nd; q++) hd[q] = c.h_d[q];
  for (int q = 0; q < nf; q++) hf[q] = c.h_f[q];
  return rc;
}

int bc_vec(float* a, const GridX& g, const float* U, int saveexit, unsigned per, hipStream_t s) {
  // largest plane cross-section decides grid.x
  long cmax = (long)g.ny * (g.D == 3 ? g.nz : 1);
  cmax = cmax > (long)g.nx * (g.D == 3 ? g.nz : 1) ? cmax : (long)g.nx * (g.D == 3 ? g.nz : 1);
  cmax = cmax > g.sz ? cmax : g.sz;
  const bool dist = (g.D == 3) && (g.nz != g.gnz);
  const int zwalls = (g.D == 3) ? ((dist && ((per >> 2) & 1u)) ? 0 : 1) : 0;
  dim3 grid((unsigned)((cmax + WL_BLOCK - 1) / WL_BLOCK), (unsigned)(3 * g.D), 1);
  DSEL(g.D, k_bc_vec, grid, dim3(WL_BLOCK), 0, s, g, a, U[0], U[1], g.D == 3 ? U[2] : 0.f, saveexit, per, zwalls);
  WL_LAUNCH_CHECK(); return 0;
}
int bc_vec_fn(float* a, const float* Ub, const GridX& g, int saveexit, unsigned per, hipStream_t s) {
  long cmax = (long)g.ny * (g.D == 3 ? g.nz : 1);
  cmax = cmax > (long)g.nx * (g.D == 3 ? g.nz : 1) ? cmax : (long)g.nx * (g.D == 3 ? g.nz : 1);
  cmax = cmax > g.sz ? cmax : g.sz;
  const bool dist = (g.D == 3) && (g.nz != g.gnz);
  const int zwalls = (g.D == 3) ? ((dist && ((per >> 2) & 1u)) ? 0 : 1) : 0;
  dim3 grid((unsigned)((cmax + WL_BLOCK - 1) / WL_BLOCK), (unsigned)(3 * g.D), 1);
  DSEL(g.D, k_bc_vec_fn, grid, dim3(WL_BLOCK), 0, s, g, a, Ub, saveexit, per, zwalls);
  WL_LAUNCH_CHECK(); return 0;
}
int meanflow_update(float* P, float* U, float* UU, const float* p, const float* u, const GridX& g, float e, hipStream_t s) {
  hipLaunchKernelGGL(k_meanflow, dim3(grid1d((size_t)g.cs)), dim3(WL_BLOCK), 0, s, P, U, UU, p, u, g.cs, g.D, e);
  WL_LAUNCH_CHECK(); return 0;
}
int meanflow_uu(float* tau, const float* UU, const float* U, const GridX& g, hipStream_t s) {
  hipLaunchKernelGGL(k_meanflow_uu, dim3(grid1d((size_t)g.cs)), dim3(WL_BLOCK), 0, s, tau, UU, U, g.cs, g.D);
  WL_LAUNCH_CHECK(); return 0;
}
int add_field(float* r, const float* gf, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_add_field, dim3(grid1d(n)), dim3(WL_BLOCK), 0, s, r, gf, (long)n);
  WL_LAUNCH_CHECK(); return 0;
}
int bc_per_scalar(float* a, const GridX& g, unsigned per, hipStream_t s) {
  if (!per) return 0;
  long cmax = (long)g.ny * (g.D == 3 ? g.nz : 1);
  cmax = cmax > (long)g.nx * (g.D == 3 ? g.nz : 1) ? cmax : (long)g.nx * (g.D == 3 ? g.nz : 1);
  cmax = cmax > g.sz ? cmax : g.sz;
  unsigned p = per;
  if (g.D == 3 && g.nz != g.gnz) p &= ~4u;   // distributed periodic z is a halo exchange, not a local copy
  dim3 grid((unsigned)((cmax + WL_BLOCK - 1) / WL_BLOCK), (unsigned)(2 * g.D), 1);
  DSEL(g.D, k_bc_per_scalar, grid, dim3(WL_BLOCK), 0, s, g, a, p);
  WL_LAUNCH_CHECK(); return 0;
}

template <int D, int SCH>
static int conv_diff_launch2(float* r, const float* u, float* Phi, const GridX& g, float nu, unsigned per, hipStream_t s, const BdimArgs* bd, int ka, int kb, bool q1, BcFold* fold) {
  // planes: owned planes plus the physical ghost planes held by this rank (single domain: all planes), cut to [ka,kb)
  int kfirst = 0, klast = 1;
  if (D == 3) { kfirst = (g.gk + g.k0 == 1) ? g.k0 - 1 : g.k0; klast = (g.gk + g.k1 == g.gnz - 1) ? g.k1 + 1 : g.k1; }
  const bool zfree = D == 3 && ((per >> 2) & 1u) && g.nz != g.gnz;          // z-periodic slabs: no physical ghost planes, the halos are exchanged
  if (zfree) { kfirst = g.k0; klast = g.k1; }
  if (D == 3) { if (ka > kfirst) kfirst = ka; if (kb < klast) klast = kb; }
  if (!q1) Phi = nullptr;
  if (kfirst >= klast) {
    if (Phi) {
      long cmax = (long)g.ny * (D == 3 ? g.nz : 1);
      cmax = cmax > (long)g.nx * (D == 3 ? g.nz : 1) ? cmax : (long)g.nx * (D == 3 ? g.nz : 1);
      cmax = cmax > g.sz ? cmax : g.sz;
      hipLaunchKernelGGL((k_conv_q1<D, SCH>), dim3((unsigned)((cmax + WL_BLOCK - 1) / WL_BLOCK), (unsigned)D, 1), dim3(WL_BLOCK), 0, s, g, Phi, u, nu, per);
      WL_LAUNCH_CHECK();
    }
    return 0;
  }
  const dim3 grid = wl_plane_grid(g, klast - kfirst);
  const bool small = g.cs < (1L << 30);   // 32-bit element offsets inside one component
  BdimArgs b0{nullptr, nullptr, nullptr, 0.f, 0.f, 1.f, 0, 0, {0.f, 0.f, 0.f}};
  const BdimArgs ba = bd ? *bd : b0;
  // fused NoBody conv_diff!+BDIM! without the f store: the LDS-tiled z-marching kernel (wl_convt.hip) on the owned interior planes
  const int own_a = kfirst > g.k0 ? kfirst : g.k0, own_b = klast < g.k1 ? klast : g.k1;
  const bool tiled = D == 3 && bd && !bd->near && !r && bd->cl_on && wl::conv_tile_ok(g, per, own_b - own_a);
  if (tiled) {
    // BC!(u_out,U) folded into the producer (wl_bcfold.hpp) when this launch covers the whole single domain: x/y in the wall tiles'
    // stores, the z ghost planes by one small launch — the caller then skips k_bc_vec (fold->on reports what happened)
    BdimArgs bt = *bd;
    const bool foldok = fold && fold->on && SCH != WL_VANLEER && g.nz == g.gnz && own_a == g.k0 && own_b == g.k1 && g.nx >= 6 && g.ny >= 6 && g.nz >= 6;
    bt.bc_on = foldok ? 1 : 0;
    if (foldok) for (int c = 0; c < 3; c++) bt.bcU[c] = fold->U[c];
    if (fold && fold->proj_x) {   // mom_project!'s deferred tail: the kernel reads u through u −= L∇x and BC!(u,U) (wl_convf.hip, PROJ; the caller checked conv_proj_ok)
      bt.px = fold->proj_x; for (int c = 0; c < 3; c++) bt.bcU[c] = fold->U[c];
      fold->proj_done = 1;
    }
    if (fold && fold->dt_dev) { if (!wl::conv_flux_on()) { wl_set_error("conv_diff_bdim: Δt on the device needs the flux-once kernel"); return WL_EINVAL; } bt.dt_dev = fold->dt_dev; }
    WL_TRY(wl::conv_tile(u, g, nu, SCH, own_a, own_b, &bt, s));
    if (foldok) WL_TRY(wl::bc_zplanes(bt.uout, g, fold->U[2], s));
    if (fold) fold->on = foldok ? 1 : 0;
  } else if (fold) { if (fold->dt_dev) { wl_set_error("conv_diff_bdim: Δt on the device needs the tiled kernel"); return WL_EINVAL; } fold->on = 0; }
  const bool march = !tiled && D == 3 && wl::conv_march_ok(g);   // z-marching variant (wl_convm.hip): same arithmetic, the z-star in registers
  if (march && !(bd && bd->near)) WL_TRY(wl::conv_march(r, u, g, nu, per, SCH, kfirst, klast, bd, s));
#define WL_CD(PERF, IDXT, FUSEF)                                                                                                                  \
  hipLaunchKernelGGL((k_conv_diff<D, SCH, PERF, IDXT, FUSEF>), grid, dim3(WL_BLOCK), 0, s, g, r, u, nu, per, kfirst, ba)
  if (tiled || (march && !(bd && bd->near))) {}
  else if (bd && bd->near) { if (per) { if (small) WL_CD(1, int, 2); else WL_CD(1, long, 2); } else { if (small) WL_CD(0, int, 2); else WL_CD(0, long, 2); } }
  else if (bd) { if (per) { if (small) WL_CD(1, int, 1); else WL_CD(1, long, 1); } else { if (small) WL_CD(0, int, 1); else WL_CD(0, long, 1); } }
  else    { if (per) { if (small) WL_CD(1, int, 0); else WL_CD(1, long, 0); } else { if (small) WL_CD(0, int, 0); else WL_CD(0, long, 0); } }
#undef WL_CD
  if (Phi) {
    long cmax = (long)g.ny * (D == 3 ? g.nz : 1);
    cmax = cmax > (long)g.nx * (D == 3 ? g.nz : 1) ? cmax : (long)g.nx * (D == 3 ? g.nz : 1);
    cmax = cmax > g.sz ? cmax : g.sz;
    hipLaunchKernelGGL((k_conv_q1<D, SCH>), dim3((unsigned)((cmax + WL_BLOCK - 1) / WL_BLOCK), (unsigned)D, 1), dim3(WL_BLOCK), 0, s, g, Phi, u, nu, per);
  }
  WL_LAUNCH_CHECK(); return 0;
}
template <int D>
static int conv_diff_launch(float* r, const float* u, float* Phi, const GridX& g, float nu, unsigned per, int scheme, hipStream_t s, const BdimArgs* bd,
                            int ka = -(1 << 30), int kb = 1 << 30, bool q1 = true, BcFold* fold = nullptr) {
  if (fold && g.D != 3) fold->on = 0;
  switch (scheme) {
    case WL_QUICK: return conv_diff_launch2<D, WL_QUICK>(r, u, Phi, g, nu, per, s, bd, ka, kb, q1, fold);
    case WL_VANLEER: return conv_diff_launch2<D, WL_VANLEER>(r, u, Phi, g, nu, per, s, bd, ka, kb, q1, fold);
    case WL_CDS: return conv_diff_launch2<D, WL_CDS>(r, u, Phi, g, nu, per, s, bd, ka, kb, q1, fold);
  }
  wl_set_error("unknown scheme"); return WL_EINVAL;
}
int conv_diff(float* r, const float* u, float* Phi, const GridX& g, float nu, unsigned per, int scheme, hipStream_t s) {
  return g.D == 3 ? conv_diff_launch<3>(r, u, Phi, g, nu, per, scheme, s, nullptr) : conv_diff_launch<2>(r, u, Phi, g, nu, per, scheme, s, nullptr);
}
// quirk Q1 alone: the stale ghost-plane fluxes conv_diff! leaves in Φ≡σ (used with the z-marching kernel)
int conv_q1(float* Phi, const float* u, const GridX& g, float nu, unsigned per, int scheme, hipStream_t s) {
  if (!Phi) return 0;
  long cmax = (long)g.ny * (g.D == 3 ? g.nz : 1);
  cmax = cmax > (long)g.nx * (g.D == 3 ? g.nz : 1) ? cmax : (long)g.nx * (g.D == 3 ? g.nz : 1);
  cmax = cmax > g.sz ? cmax : g.sz;
  const dim3 grid((unsigned)((cmax + WL_BLOCK - 1) / WL_BLOCK), (unsigned)g.D, 1);
#define WL_Q1(DD, SCHV) hipLaunchKernelGGL((k_conv_q1<DD, SCHV>), grid, dim3(WL_BLOCK), 0, s, g, Phi, u, nu, per)
  if (g.D == 3) { if (scheme == WL_QUICK) WL_Q1(3, WL_QUICK); else if (scheme == WL_VANLEER) WL_Q1(3, WL_VANLEER); else WL_Q1(3, WL_CDS); }
  else { if (scheme == WL_QUICK) WL_Q1(2, WL_QUICK); else if (scheme == WL_VANLEER) WL_Q1(2, WL_VANLEER); else WL_Q1(2, WL_CDS); }
#undef WL_Q1
  WL_LAUNCH_CHECK(); return 0;
}
// conv_diff!(f,u_adv,σ) + BDIM! (NoBody) in one launch: u_out (and f unless f == NULL) written, u_out must not alias u_adv
int conv_diff_bdim(float* f, const float* u_adv, float* Phi, const float* u0, const float* mu0, float* u_out, const GridX& g, float nu, unsigned per, int scheme,
                   float dt, float pre, float post, const ConstL& cl, hipStream_t s, int ka, int kb, bool q1, BcFold* fold) {
  if (u_out == u_adv) { wl_set_error("conv_diff_bdim: output aliases the advecting field"); return WL_EINVAL; }
  BdimArgs bd{u0, mu0, u_out, dt, pre, post, (post != 1.f) ? 1 : 0, cl.on, {cl.c[0], cl.c[1], cl.c[2]}};
  if (g.D != 3) { if (fold) fold->on = 0; return conv_diff_launch<2>(f, u_adv, Phi, g, nu, per, scheme, s, &bd); }
  return conv_diff_launch<3>(f, u_adv, Phi, g, nu, per, scheme, s, &bd, ka, kb, q1, fold);
}
// conv_diff!+BDIM! for a flow with a body (see k_conv_diff<…,FUSE=2>): u_out ≠ u_adv; f is an output array (raw r near the body)
// dz0..dz1 (inclusive; dz1 < dz0: unknown): the planes on which any workgroup is near the body, keeps f or has a μ₀ off the NoBody pattern
// (body_masks_planes).  Every other plane is a NoBody plane bit for bit (μ₁ ≡ 0, V ≡ 0, μ₀ = the wall pattern, f not read by anyone):
// those planes run the LDS-tiled kernel (wl_convt.hip) in its NoBody form, the planes dz0..dz1 this file's gather kernel (FUSE=2).
int g_body_tile = 1;
void conv_body_tile_enable(int on) { g_body_tile = on; }
int conv_diff_bdim_body(float* f, const float* u_adv, float* Phi, const float* u0, const float* mu0, float* u_out, const GridX& g, float nu, unsigned per, int scheme,
                        float dt, float pre, float post, const unsigned char* near, const unsigned char* needf, const unsigned char* m0var, int nbm, int store_all, hipStream_t s,
                        int dz0, int dz1) {
  if (u_out == u_adv || !near || !needf || !m0var || !f) { wl_set_error("conv_diff_bdim_body: bad arguments"); return WL_EINVAL; }
  BdimArgs bd{u0, mu0, u_out, dt, pre, post, (post != 1.f) ? 1 : 0, 0, {0.f, 0.f, 0.f}, near, needf, nbm, store_all, m0var};
  if (g.D == 3 && g_body_tile && !store_all && per == 0 && dz1 >= dz0 && g.nz == g.gnz) {
    int na = dz0 > g.k0 ? dz0 : g.k0, nb = dz1 + 1 < g.k1 ? dz1 + 1 : g.k1;        // gather range [na,nb)
    if (na - g.k0 < 8) na = g.k0;                                                     // far ranges too short for a march join the gather range
    if (g.k1 - nb < 8) nb = g.k1;
    const int nfar = (na - g.k0) + (g.k1 - nb);
    if (nfar > 0 && nb > na && wl::conv_tile_ok(g, per, 2 * nfar)) {                  // (half the NoBody size gate: the launch replaces a much slower kernel)
      BdimArgs bt{u0, mu0, u_out, dt, pre, post, (post != 1.f) ? 1 : 0, 1, {1.f, 1.f, 1.f}};
      if (na > g.k0) WL_TRY(wl::conv_tile(u_adv, g, nu, scheme, g.k0, na, &bt, s));
      if (g.k1 > nb) WL_TRY(wl::conv_tile(u_adv, g, nu, scheme, nb, g.k1, &bt, s));
      return conv_diff_launch<3>(f, u_adv, Phi, g, nu, per, scheme, s, &bd, na, nb);
    }
  }
  return g.D == 3 ? conv_diff_launch<3>(f, u_adv, Phi, g, nu, per, scheme, s, &bd) : conv_diff_launch<2>(f, u_adv, Phi, g, nu, per, scheme, s, &bd);
}
// first / last plane (inclusive) on which any workgroup is near the body, keeps f or loads μ₀; host-synchronising (measure!/update! time only)
int body_masks_planes(const unsigned char* near, const unsigned char* needf, const unsigned char* m0var, const GridX& g, int* dz, hipStream_t s) {
  const int nbm = body_masks_nbm(g);
  const size_t n = (size_t)nbm * g.nz;
  std::vector<unsigned char> h(3 * n);
  WL_HIP(hipMemcpyAsync(h.data(), near, n, hipMemcpyDeviceToHost, s));
  WL_HIP(hipMemcpyAsync(h.data() + n, needf, n, hipMemcpyDeviceToHost, s));
  WL_HIP(hipMemcpyAsync(h.data() + 2 * n, m0var, n, hipMemcpyDeviceToHost, s));
  WL_HIP(hipStreamSynchronize(s));
  dz[0] = g.nz; dz[1] = -1;
  for (int k = 0; k < g.nz; k++) {
    bool any = false;
    for (int b = 0; b < nbm && !any; b++) any = h[(size_t)k * nbm + b] || h[n + (size_t)k * nbm + b] || h[2 * n + (size_t)k * nbm + b];
    if (any) { if (k < dz[0]) dz[0] = k; if (k > dz[1]) dz[1] = k; }
  }
  return 0;
}
int body_masks_nbm(const GridX& g) { return 8 * wl_strip_blocks(g); }
int body_masks(unsigned char* near, unsigned char* needf, unsigned char* m0var, const float* V, const float* mu1, const float* mu0, const GridX& g, hipStream_t s) {
  const int nbm = body_masks_nbm(g);
  WL_HIP(hipMemsetAsync(near, 0, (size_t)nbm * g.nz, s)); WL_HIP(hipMemsetAsync(needf, 0, (size_t)nbm * g.nz, s)); WL_HIP(hipMemsetAsync(m0var, 0, (size_t)nbm * g.nz, s));
  DSEL(g.D, k_body_mask2, wl_plane_grid(g, g.nz), dim3(WL_BLOCK), 0, s, g, V, mu1, mu0, near, needf, m0var, nbm);
  WL_LAUNCH_CHECK(); return 0;
}
int bdim_near(float* uout, const float* uin, const float* u0, float* f, const float* V, const float* mu0, const float* mu1, const GridX& g, float dt, float pre, float post,
              const unsigned char* near, int nbm, const int* box, hipStream_t s) {
  if (box[1] < box[0] || box[3] < box[2]) return 0;            // no near workgroup at all
  const dim3 grid((unsigned)(box[1] - box[0] + 1), (unsigned)(box[3] - box[2] + 1), 1);
  DSEL(g.D, k_bdim_f_m, grid, dim3(WL_BLOCK), 0, s, g, f, u0, V, dt, near, nbm, box[0], box[2]);
  DSEL(g.D, k_bdim_u_m, grid, dim3(WL_BLOCK), 0, s, g, uout, uin, (const float*)f, V, mu0, mu1, pre, post, (post != 1.f) ? 1 : 0, near, nbm, box[0], box[2]);
  WL_LAUNCH_CHECK(); return 0;
}
// bounding box {b0,b1,k0,k1} (inclusive) of the near workgroups; host-synchronising (measure!/update! time only)
int body_masks_box(const unsigned char* near, const GridX& g, int* box, hipStream_t s) {
  const int nbm = body_masks_nbm(g);
  std::vector<unsigned char> h((size_t)nbm * g.nz);
  WL_HIP(hipMemcpyAsync(h.data(), near, h.size(), hipMemcpyDeviceToHost, s));
  WL_HIP(hipStreamSynchronize(s));
  box[0] = nbm; box[1] = -1; box[2] = g.nz; box[3] = -1;
  for (int k = 0; k < g.nz; k++) for (int b = 0; b < nbm; b++) if (h[(size_t)k * nbm + b]) {
    if (b < box[0]) box[0] = b; if (b > box[1]) box[1] = b; if (k < box[2]) box[2] = k; if (k > box[3]) box[3] = k;
  }
  return 0;
}
int bdim(float* u, const float* u0, float* f, const float* V, const float* mu0, const float* mu1, const GridX& g, float dt, float pre, float post, hipStream_t s) {
  const int scale_after = (post != 1.f) ? 1 : 0;
  if (!V && !mu1) {
    DSEL(g.D, k_bdim_nobody, wl_plane_grid(g, g.nz), dim3(WL_BLOCK), 0, s, g, u, u0, f, mu0, dt, pre, post, scale_after);
  } else {
    const long n = g.cs * g.D;
    hipLaunchKernelGGL(k_bdim_f, dim3(grid1d((size_t)n)), dim3(WL_BLOCK), 0, s, g, f, u0, V, dt, n);
    DSEL(g.D, k_bdim_u, wl_plane_grid(g, g.k1 - g.k0), dim3(WL_BLOCK), 0, s, g, u, f, V, mu0, mu1, pre, post, scale_after, (const unsigned char*)nullptr);
  }
  WL_LAUNCH_CHECK(); return 0;
}
size_t body_mask_bytes(const GridX& g) { return (size_t)wl_plane_grid(g, g.k1 - g.k0).x; }
int body_mask(unsigned char* far, const float* V, const float* mu1, const GridX& g, hipStream_t s) {
  DSEL(g.D, k_body_mask, wl_plane_grid(g, g.k1 - g.k0), dim3(WL_BLOCK), 0, s, g, V, mu1, far);
  WL_LAUNCH_CHECK(); return 0;
}
int accelerate(float* r, const GridX& g, const float* a, hipStream_t s) {
  const long n = g.cs * g.D;
  hipLaunchKernelGGL(k_accelerate, dim3(grid1d((size_t)n)), dim3(WL_BLOCK), 0, s, r, g.cs, n, a[0], a[1], g.D == 3 ? a[2] : 0.f);
  WL_LAUNCH_CHECK(); return 0;
}
int bdim_f(float* f, const float* u0, const float* V, const GridX& g, float dt, hipStream_t s) {
  const long n = g.cs * g.D;
  hipLaunchKernelGGL(k_bdim_f, dim3(grid1d((size_t)n)), dim3(WL_BLOCK), 0, s, g, f, u0, V, dt, n);
  WL_LAUNCH_CHECK(); return 0;
}
int bdim_u(float* u, const float* f, const float* V, const float* mu0, const float* mu1, const GridX& g, float pre, float post, hipStream_t s, const unsigned char* far) {
  DSEL(g.D, k_bdim_u, wl_plane_grid(g, g.k1 - g.k0), dim3(WL_BLOCK), 0, s, g, u, f, V, mu0, mu1, pre, post, (post != 1.f) ? 1 : 0, far);
  WL_LAUNCH_CHECK(); return 0;
}
int scale_u(float* u, const GridX& g, float sc, hipStream_t s) { DSEL(g.D, k_scale_u, wl_plane_grid(g, g.k1 - g.k0), dim3(WL_BLOCK), 0, s, g, u, sc); WL_LAUNCH_CHECK(); return 0; }
int div(float* z, const float* u, const GridX& g, hipStream_t s) { DSEL(g.D, k_div, wl_plane_grid(g, g.nz), dim3(WL_BLOCK), 0, s, g, z, (float*)nullptr, u, 1.f); WL_LAUNCH_CHECK(); return 0; }
int div_scale(float* z, float* x, const float* u, const GridX& g, float dt, hipStream_t s) { DSEL(g.D, k_div, wl_plane_grid(g, g.nz), dim3(WL_BLOCK), 0, s, g, z, x, u, dt); WL_LAUNCH_CHECK(); return 0; }
int project(float* u, const float* L, const float* x, const GridX& g, hipStream_t s) { DSEL(g.D, k_project, wl_plane_grid(g, g.k1 - g.k0), dim3(WL_BLOCK), 0, s, g, u, L, x); WL_LAUNCH_CHECK(); return 0; }
int cfl_dev(const float* u, float* sigma, const GridX& g, const RedWs& ws, int slot_f, hipStream_t s) {
  int kfirst = 0, klast = 1;
  if (g.D == 3) { kfirst = (g.gk + g.k0 == 1) ? g.k0 - 1 : g.k0; klast = (g.gk + g.k1 == g.gnz - 1) ? g.k1 + 1 : g.k1; }
  const int zc = wl_march_chunk(g, klast - kfirst);
  dim3 grid = wl_plane_grid(g, wl_march_slots(klast - kfirst, zc));
  DSEL(g.D, k_cfl, grid, dim3(WL_BLOCK), 0, s, g, u, sigma, ws.pm, kfirst, klast, zc);
  hipLaunchKernelGGL(k_fin_max, dim3(1), dim3(WL_BLOCK), 0, s, ws.pm, (int)grid.x, ws.res_f + slot_f);
  WL_LAUNCH_CHECK(); return 0;
}
}  // namespace wl
