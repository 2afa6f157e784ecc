// conv_diff! per-cell arithmetic shared by the plane kernel (wl_flow.hip) and the z-marching kernel (wl_convm.hip).
#pragma once
#include "wl_common.hpp"

namespace {
// ---- convective schemes   src/Flow.jl:4-6,27-36 --------------------------------------------------
// median(a,b,c) src/Flow.jl:27-36 — one v_med3_f32; identical value to the reference's branchy form for non-NaN inputs
// (the branchy form compiled to ~700 exec-mask instructions per cell in conv_diff!).
__device__ __forceinline__ float median3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }
// x/6 correctly rounded without the ~10-instruction IEEE division sequence: float(double(x)·(1/6)) equals x/6 for EVERY
// float x (verified exhaustively over all 2^24 significands × normal/subnormal exponents, tools/check_div6.py): the
// quotient of a float by 6 is never closer than ~2^-27 (relative) to a rounding midpoint, far above the 2^-53 product error.
#ifdef WL_NO_DIV6
__device__ __forceinline__ float div6(float x) { return x / 6; }
#elif defined(WL_DIV6_FMA)
// EXPERIMENT (not the default): q = x·RN(1/6), r = x − 6q (exact, one FMA), q + r·RN(1/6) (one FMA) — three full-rate
// instructions.  Verified exhaustively against IEEE x/6.0f over all 2^32 bit patterns (tools/check_div6_fma.c): correctly
// rounded for every |x| ≥ 2^-125; below that the quotient is subnormal, exact ties exist and 2.8 M inputs round the other
// way (and −0 gives +0).  A guard branch costs more than it saves (it splits the flux code into many basic blocks: 150+
// spilled VGPRs), so the default stays the double-precision product below, exact for every float.
__device__ __forceinline__ float div6(float x) { const float c = 0x1.555556p-3f; const float q = x * c; const float r = __builtin_fmaf(-6.f, q, x); return __builtin_fmaf(r, c, q); }
#else
__device__ __forceinline__ float div6(float x) { return (float)((double)x * (1.0 / 6.0)); }
#endif
template <int SCH> __device__ __forceinline__ float lam(float u, float c, float d) {
  if (SCH == WL_QUICK) return median3(div6(5 * c + 2 * d - u), c, median3(10 * c - 9 * u, c, d));
  if (SCH == WL_VANLEER) return (c <= fminf(u, d) || c >= fmaxf(u, d)) ? c : c + (d - c) * (c - u) / (d - u);
  return (c + d) / 2;
}

// flux Φ_ab at the lower b-face of the cell at offset o (component a advected, direction b).
//  pb   : Julia index of that cell along b (2..Ng_b),  nb = Ng_b,  sb = stride along b, sa = stride along a
//  variant by position: pb==2 lower boundary (ϕuL / periodic ϕuP), 3..nb-1 inner (ϕu), pb==nb upper (ϕuR / periodic reuse of index 2)
//  returns the value V such that the reference does  r[I] += V  for the cell on the UPPER side of the face
//  (lower/inner: V=Φ) — for the upper-boundary face the caller applies r[I-δ] += (-ϕuR + ν∂) itself.
template <int SCH>
__device__ __forceinline__ float flux_inner(const float* __restrict__ f, const float* __restrict__ ub, long o, long sb, long sa, float nu) {
  const float U = (ub[o] + ub[o - sa]) / 2;                                              // ϕ(i,CI(I,j),u)   src/Flow.jl:3,47
  const float conv = U > 0 ? U * lam<SCH>(f[o - 2 * sb], f[o - sb], f[o]) : U * lam<SCH>(f[o + sb], f[o], f[o - sb]);   // ϕu :8
  return conv - nu * (f[o] - f[o - sb]);
}
template <int SCH>
__device__ __forceinline__ float flux_lowerL(const float* __restrict__ f, const float* __restrict__ ub, long o, long sb, long sa, float nu) {
  const float U = (ub[o] + ub[o - sa]) / 2;
  const float conv = U > 0 ? U * ((f[o] + f[o - sb]) / 2) : U * lam<SCH>(f[o + sb], f[o], f[o - sb]);                   // ϕuL :10
  return conv - nu * (f[o] - f[o - sb]);
}
template <int SCH>
__device__ __forceinline__ float flux_lowerP(const float* __restrict__ f, const float* __restrict__ ub, long o, long sb, long sa, float nu, long op) {
  const float U = (ub[o] + ub[o - sa]) / 2;
  const float conv = U > 0 ? U * lam<SCH>(f[op], f[o - sb], f[o]) : U * lam<SCH>(f[o + sb], f[o], f[o - sb]);           // ϕuP :9
  return conv - nu * (f[o] - f[o - sb]);
}
template <int SCH>
__device__ __forceinline__ float flux_upperR(const float* __restrict__ f, const float* __restrict__ ub, long o, long sb, long sa, float nu) {
  const float U = (ub[o] + ub[o - sa]) / 2;
  const float conv = U < 0 ? U * ((f[o] + f[o - sb]) / 2) : U * lam<SCH>(f[o - 2 * sb], f[o - sb], f[o]);               // ϕuR :11
  return -conv + nu * (f[o] - f[o - sb]);                                                 // upperBoundary! :57
}

// conv_diff!(r,u,Φ,λ;ν,perdir) in gather form   src/Flow.jl:38-62, ranges src/core.jl:55-57,188-190
// One thread per cell of the WHOLE array (r .= 0 included).  For cell I (Julia indices) and component a:
//   r[I,a] = Σ_b [ +Φ_ab(I) − Φ_ab(I+δ_b) ]  taken in the reference's (j inner) order, direction b
//   contributing iff I_b ∈ 2..Ng_b−1 and every other I_c ∈ 2..Ng_c (upper ghost INCLUDED, as inside_u does).
// Straight-line, fully unrolled and predicated: every load of the (a,b) pair is independent of the others (the
// first version kept the a/b loops rolled — 32 VGPRs, one long chain of dependent loads, 6.4 ms at 512³).
// One flux formula serves all face variants:
//   Φ = U·X − ν(f[P]−f[P−δ]),  X = λ(upwind triple by sign of U), overridden by the plain average ϕ where the
//   reference uses ϕuL (lower wall, U>0) / ϕuR (upper wall, U<0); periodic lower faces only change the address
//   of the far-upwind point (ϕuP).  r[I−δ] += −ϕuR+ν∂ equals r[I−δ] −= (ϕuR−ν∂) bit for bit, so the upper wall
//   needs no separate accumulation form.  Addresses that a masked lane would take out of range are clamped to
//   its own cell (values unused).
template <int SCH>
__device__ __forceinline__ float face_flux(float U, float t0, float t1, float t2, float avg, bool use_avg, float fc, float fm, float nu) {
  float X = lam<SCH>(t0, t1, t2);
  X = use_avg ? avg : X;
  return U * X - nu * (fc - fm);
}
// PER = 0: no periodic direction (no wrapped addresses, no branches at all); IDX = int when every component offset fits 31 bits
// FUSE = 1 appends BDIM! for the NoBody case (μ₁≡0, V≡0; src/Flow.jl:176-180 + the folded scale_u!):
//   f = u⁰ + Δt·r (all cells) ; u_out = (u·pre + μ₀·f)·post (interior).  u_out must not alias the advecting field u.
// cd_cell: r[I,a] for the three components of one cell, in the reference's (a outer, b inner) order.
//  INNER = 1: the cell is at least 2 cells away from every boundary (3 ≤ I_c ≤ N_c−2): no clamped addresses, no boundary
//             variants, no masked accumulation.
//  WIN   = 1: the five z-neighbours u_c[k−2..k+2] of the cell's own column come from the register window W[c][0..4]
//             (z-marching kernel) instead of memory; requires INNER.
template <int D, int SCH, int PER, typename IDX, int INNER, int WIN>
__device__ __forceinline__ void cd_cell(const GridX& g, const float* __restrict__ u, IDX o, const int* I, const int* N, const IDX* st, float nu, unsigned per,
                                        const float (*W)[5], float* out) {
  bool ok = true;
  if (!INNER) {
#pragma unroll
    for (int c = 0; c < D; c++) ok = ok && (I[c] >= 2);
  }
#pragma unroll
  for (int a = 0; a < D; a++) {
    const float* __restrict__ f = u + (long)a * g.cs;
    const IDX sa = (INNER || ok) ? st[a] : 0;
    float acc = 0.f;
    const float f0 = WIN ? W[a][2] : f[o];
#pragma unroll
    for (int b = 0; b < D; b++) {
      const float* __restrict__ ub = u + (long)b * g.cs;
      const bool pb = !INNER && PER && ((per >> b) & 1u);
      const bool con = INNER || (ok && (I[b] <= N[b] - 1));
      const bool lowb = !INNER && (I[b] == 2), upb = !INNER && (I[b] + 1 == N[b]);
      const IDX sb = con ? st[b] : 0;
      // star of f along b, clamped where the variant never reads it
      const IDX om2 = lowb ? (pb ? (IDX)(N[b] - 4) * sb : -sb) : -2 * sb;       // far upwind of my lower face (ϕuP wraps)
      const IDX op2 = (INNER || I[b] + 2 <= N[b]) ? 2 * sb : sb;                 // far downwind of my upper face
      const bool wz = WIN && b == 2;                                             // z-star from the register window
      const float fm2 = wz ? W[a][0] : f[o + om2], fm1 = wz ? W[a][1] : f[o - sb], fp1 = wz ? W[a][3] : f[o + sb], fp2 = wz ? W[a][4] : f[o + op2];
      // advecting velocity at the two faces: U = ϕ(a, CI(I,b), u) = (u_b[I] + u_b[I−δ_a])/2        src/Flow.jl:3,47
      const float ub0 = WIN ? W[b][2] : ub[o];
      const float ubm = (WIN && a == 2) ? W[b][1] : ub[o - sa];                  // u_b[I−δ_a]
      const float ubp = (WIN && b == 2) ? W[b][3] : ub[o + sb];                  // u_b[I+δ_b]
      const float Ul = (ub0 + ubm) / 2;
      const bool posl = Ul > 0;
      const float Pl = face_flux<SCH>(Ul, posl ? fm2 : fp1, posl ? fm1 : f0, posl ? f0 : fm1, (f0 + fm1) / 2, lowb && !pb && posl, f0, fm1, nu);
      // upper face of I = lower face of I+δ_b
      float Pu;
      if (!INNER && PER && upb && pb) {   // periodic: Φ[CIj(j,I,2)] — the wrapped lower-face flux at index 2   src/Flow.jl:62 (rare plane)
        const long o2 = (long)o + (long)(2 - I[b]) * sb;
        Pu = flux_lowerP<SCH>(f, ub, o2, sb, sa, nu, o2 + (long)(N[b] - 4) * sb);
      } else {
        const float ubd = (WIN && a == 2 && b == 2) ? W[b][2] : ub[o + sb - sa];   // u_b[I+δ_b−δ_a]
        const float Uu = (ubp + ubd) / 2;
        const bool posu = Uu > 0;
        Pu = face_flux<SCH>(Uu, posu ? fm1 : fp2, posu ? f0 : fp1, posu ? fp1 : f0, (fp1 + f0) / 2, upb && !pb && (Uu < 0), fp1, f0, nu);
      }
      acc = con ? acc + Pl : acc;
      acc = con ? acc - Pu : acc;
    }
    out[a] = acc;
  }
}
struct BdimArgs { const float* u0; const float* mu0; float* uout; float dt, pre, post; int scale_after; int cl_on; float cl_c[3];
                  // FUSE == 2 (flows with a body): per workgroup and plane, 1 = the body is near (μ₁ or V nonzero) / 1 = a near cell reads f here
                  const unsigned char* near; const unsigned char* needf; int nbm; int store_all;
                  const unsigned char* m0var;      // m0var: 1 = μ₀ deviates from "1 inside, 0 on wall faces" somewhere in the workgroup's cells
                  int bc_on; float bcU[3];         // BC!(u_out, U) folded into the stores (wl_bcfold.hpp): only the tiled kernel honours it
                  const float* px;
                  const float* dt_dev; };          // flux-once kernel only: Δt read from the device (BcFold::dt_dev); `dt` is ignored then              // flux-once kernel only (wl_convf.hip, PROJ): the advecting field is u* − L∇x with BC!(·,bcU), x = px; bcU is then set even when bc_on = 0
// FUSE epilogue shared by both kernels: f = u⁰ + Δt·r (all cells) ; u_out = (u·pre + μ₀·f)·post (interior)      BDIM! NoBody, src/Flow.jl:176-180
template <int D, typename IDX>
__device__ __forceinline__ void cd_store(const GridX& g, float* __restrict__ r, const float* __restrict__ u, IDX o, const int* I, const int* N, bool in, const float* out, int fuse, const BdimArgs& bd) {
#pragma unroll
  for (int a = 0; a < D; a++) {
    const long oa = (long)a * g.cs + o;
    if (!fuse) { r[oa] = out[a]; continue; }
    const float fn = bd.u0[oa] + bd.dt * out[a] - 0.f;
    if (r) r[oa] = fn;                       // f itself is optional: nothing on the time-step path reads it again
    if (in) {
      const float m0 = bd.cl_on ? wl::wl_cl_coef(I[a], N[a], bd.cl_c[a]) : bd.mu0[oa];     // μ₀ on a verified NoBody field
      const float xx = (0.f / 2 + 0.f) + m0 * fn;
      float un = (bd.pre == 0.f) ? xx : (u[oa] * bd.pre + xx);
      if (bd.scale_after) un = un * bd.post;
      bd.uout[oa] = un;
    }
  }
}
}  // namespace
