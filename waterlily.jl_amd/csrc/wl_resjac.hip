// mom_project!'s head and the V-cycle's first smoother in ONE z-marching kernel (finest level, constant coefficients):
//   z = ∇·u ; x·= dt ; residual!(x)  (src/Flow.jl:225, src/Poisson.jl:92-95)   →   Jacobi!(it=1, ω=1)  (src/Poisson.jl:111-114,
//   src/MultiLevelPoisson.jl:92)   [+ Σr, and L₁/L∞ of r: solver!'s first norms, src/MultiLevelPoisson.jl:110]
//
// Why: as two kernels (k_div_residual, k_jacobi_march_cl) the pair moves 28 + 16 B/cell — r and the scaled x are written by the
// first and read back by the second.  Fused, r and ϵ = r·iD never leave the chip: a thread owns two x-adjacent cells of a 64×32-cell
// tile and marches along z; x' = x·dt of the plane being differenced and ϵ of the plane being relaxed sit in LDS (even/odd-x arrays,
// as in wl_fused2) for the in-plane neighbours, the z-neighbours in registers.  The residual is evaluated on the tile's core plus one
// ring (the relaxation's stencil), the scaled pressure on core plus two.  Reads p, u; writes x, r': ≈24 B/cell plus tile halos.
//
// residual!'s mean shift (r .-= Σr/N if |Σr/N| > 2eps, src/Poisson.jl:95-97) sits between the two operations and needs the GLOBAL
// sum.  The kernel assumes "no shift" (the outcome for every wall-bounded flow: Σ∇·u = 0 and A has zero row sums, so Σr is rounding
// noise), the host checks Σr afterwards and, if the shift is due after all, discards the outputs and runs the two-kernel path —
// the inputs are untouched.  Same statements per cell as k_div_residual / k_jacobi_march_cl ⇒ bit-identical fields.
#include <cstdlib>

#include "wl_common.hpp"

namespace {

__device__ __forceinline__ float rj_cf(int Ia, int Na, float c) { return (Ia <= 2 || Ia >= Na) ? 0.f : c; }
__device__ __forceinline__ float rj_inv(float d) { return (d == 0.f) ? d : 1.0f / d; }
__device__ __forceinline__ float2 rj_ld2(const float* __restrict__ p, unsigned o) { return *reinterpret_cast<const float2*>(p + o); }
__device__ __forceinline__ void rj_st2(float* __restrict__ p, unsigned o, float2 v, bool s0, bool s1) {
  if (s0 && s1) *reinterpret_cast<float2*>(p + o) = v;
  else { if (s0) p[o] = v.x; if (s1) p[o + 1] = v.y; }
}
struct RjBc { int on; float U[3]; };     // u is read through BC!(u,U): see wl_resjac_body.inc
int g_resjac_on = 1;
long g_resjac_min = 6L << 20;   // cells: below this the extra host read of Σr costs more than the fusion saves (tools/rj_gate.sh: 192³ −3.5 %, 160³ even, 128³ +4 %)

// The kernel exists for two tile heights (as the pair smoother): 64×32 cells (1024 threads, 82 % of a tile is core) where the launch fills the chip for
// many rounds, 64×16 cells (512 threads, 70 % core, four workgroups per CU) where it cannot — there the kernel is bound by the latency of a plane-step
// (its loads are consumed in the step that issues them; the other resident workgroups are what hides them).
#define RJ_Y 32
#define RJ_NS rj32
#include "wl_resjac_body.inc"
#undef RJ_Y
#undef RJ_NS
#define RJ_Y 16
#define RJ_NS rj16
#include "wl_resjac_body.inc"
#undef RJ_Y
#undef RJ_NS

// x_out = x·dt on the cells the march does not own: the ghost shell (mom_project!'s `b.x .*= dt` scales ALL cells, src/Flow.jl:225)
__global__ void k_scale_shell(GridX g, float* __restrict__ xout, const float* __restrict__ p, float dt) {
  const int k = blockIdx.y;
  const long base = (long)k * g.sz;
  if (k == 0 || k == g.nz - 1) {
    for (long m = (long)blockIdx.x * WL_BLOCK + threadIdx.x; m < g.sz; m += (long)gridDim.x * WL_BLOCK) xout[base + m] = p[base + m] * dt;
    return;
  }
  const long ring = 2L * g.nx + 2L * (g.ny - 2);
  for (long q = (long)blockIdx.x * WL_BLOCK + threadIdx.x; q < ring; q += (long)gridDim.x * WL_BLOCK) {
    long m;
    if (q < g.nx) m = q;                                             // row 0
    else if (q < 2L * g.nx) m = (long)(g.ny - 1) * g.sy + (q - g.nx);   // row ny−1
    else { const long t = q - 2L * g.nx; const long jj = 1 + (t >> 1); m = jj * g.sy + ((t & 1) ? g.nx - 1 : 0); }   // columns 0 and nx−1
    xout[base + m] = p[base + m] * dt;
  }
}
// does the ghost shell of p hold anything but +0?  (then x_out's shell — zero since allocation, rewritten as 0/dt by the projection tails — already IS p·dt
// and k_scale_shell can be skipped: 0.05 ms per solve at 512³, the strided columns make it slow for its size)
__global__ void k_shell_nonzero(GridX g, const float* __restrict__ p, int* __restrict__ flag) {
  const int k = blockIdx.y;
  const long base = (long)k * g.sz;
  bool bad = false;
  if (k == 0 || k == g.nz - 1) {
    for (long m = (long)blockIdx.x * WL_BLOCK + threadIdx.x; m < g.sz; m += (long)gridDim.x * WL_BLOCK) bad = bad || (__float_as_uint(p[base + m]) != 0u);
  } else {
    const long ring = 2L * g.nx + 2L * (g.ny - 2);
    for (long q = (long)blockIdx.x * WL_BLOCK + threadIdx.x; q < ring; q += (long)gridDim.x * WL_BLOCK) {
      long m;
      if (q < g.nx) m = q;
      else if (q < 2L * g.nx) m = (long)(g.ny - 1) * g.sy + (q - g.nx);
      else { const long t = q - 2L * g.nx; const long jj = 1 + (t >> 1); m = jj * g.sy + ((t & 1) ? g.nx - 1 : 0); }
      bad = bad || (__float_as_uint(p[base + m]) != 0u);
    }
  }
  if (bad) atomicOr(flag, 1);
}
// Σ of the per-workgroup partials → res_d[0] (Σr), res_d[slot_d] (L₁), res_f[slot_f] (L∞)
__global__ void k_resjac_fin(const double* __restrict__ psum, const double* __restrict__ pl1, const float* __restrict__ pmax, int n, double* __restrict__ res_d, float* __restrict__ res_f,
                             int slot_d, int slot_f) {
  double a = 0.0, l = 0.0; float mx = 0.f;
  for (int q = threadIdx.x; q < n; q += WL_BLOCK) { a += psum[q]; l += pl1[q]; mx = fmaxf(mx, pmax[q]); }
  __shared__ double sa[WL_BLOCK / 64], sl[WL_BLOCK / 64]; __shared__ float sm[WL_BLOCK / 64];
  a = wave_sum(a); l = wave_sum(l); mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = a; sl[threadIdx.x >> 6] = l; sm[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a = 0.0; l = 0.0; mx = 0.f;
    for (int q = 0; q < WL_BLOCK / 64; q++) { a += sa[q]; l += sl[q]; mx = fmaxf(mx, sm[q]); }
    res_d[0] = a; res_d[slot_d] = l; res_f[slot_f] = mx;
  }
}
}  // namespace

namespace wl {
void resjac_enable(int on, long min_cells) { g_resjac_on = on; if (min_cells >= 0) g_resjac_min = min_cells; }
// the fused head is worth it (and implemented) for: 3-D constant-coefficient finest levels (single domain or z-slab) of at least resjac_min cells
bool resjac_ok(const GridX& g, const ConstL& cl) {
  // (z-slab: two ghost planes of x and u per side — the residual of the neighbour's boundary plane is recomputed; the size gate is on the GLOBAL level)
  return g_resjac_on && cl.on && g.D == 3 && (g.nz == g.gnz ? g.gk == 0 : (g.k0 >= 2 && g.nz - g.k1 >= 2 && g.k1 - g.k0 >= 8)) && (g.nx & 1) == 0 && g.nx >= 66 && g.ny >= 34 &&
         g.gnz >= 10 && g.cs < (1L << 30) && (long)(g.nx - 2) * (g.ny - 2) * (g.gnz - 2) >= g_resjac_min;
}
// z=∇·u; x_out = x·dt (+ω·ϵ on interior cells); r_out = residual after Jacobi!(ω=w); Σr -> res_d[0], L₁(r) -> res_d[slot_d], L∞(r) -> res_f[slot_f]
// (the norms of the residual BEFORE Jacobi!, as solver! logs them).  x_out ≠ x, r_out's ghost cells are left untouched (zero).
// 1: some ghost cell of a (3-D single-domain array) is not +0, 0: all are; host-synchronising (called when the array may have been written from outside)
int shell_nonzero(const float* a, const GridX& g, int* dev_flag, hipStream_t s) {
  WL_HIP(hipMemsetAsync(dev_flag, 0, sizeof(int), s));
  hipLaunchKernelGGL(k_shell_nonzero, dim3(8, (unsigned)g.nz), dim3(WL_BLOCK), 0, s, g, a, dev_flag);
  int h = 1;
  WL_HIP(hipMemcpyAsync(&h, dev_flag, sizeof(int), hipMemcpyDeviceToHost, s));
  WL_HIP(hipStreamSynchronize(s));
  return h ? 1 : 0;
}
int resjac(float* xout, float* rout, const float* x, const float* u, const GridX& g, float dt, float w, const ConstL& cl, const RedWs& ws, int slot_d, int slot_f, hipStream_t s, bool shell, const float* bcU) {
  if (xout == x) { wl_set_error("resjac: output aliases input"); return WL_EINVAL; }
  const int np = g.k1 - g.k0;
  // 16-row tiles (512 threads, four resident workgroups per CU): measured faster at every size (tools/rj_rows.sh: head 0.30 -> 0.275 ms/step at 256³,
  // 1.63 -> 1.55 at 512³ — the latency hiding of four workgroups outweighs the 12 % more halo traffic); WL_RJ_ROWS=32 brings the 1024-thread tiles back
  static const int rows_env = wl_exp_int("WL_RJ_ROWS", 0);
  const bool r16 = rows_env != 32;
  const int ntiles = r16 ? rj16::rj_tiles(g) : rj32::rj_tiles(g), per = (ntiles + 7) >> 3;
  const long SX = r16 ? rj16::rj_slots_per_xcd() : rj32::rj_slots_per_xcd();
  static const int envc = wl_exp_int("WL_RJ_CHUNK", 0);
  int zc = envc;
  if (zc <= 0) {
    // equal workgroups run in rounds of 32 CUs × the resident workgroups per CU on every XCD (tile ranges are dealt XCD by XCD): minimise
    // rounds × (planes per chunk + the 2 warm-up planes), with a balance penalty for few rounds — as wl_fused2's zchunk2.  The former fixed target of
    // 3072 workgroups gave 6.2 rounds at 512³, i.e. a seventh, almost empty one.
    static const int chforce = wl_exp_int("WL_RJ_CHUNKS", 0);     // experiments: number of chunks
    double best = -1.0; zc = np;
    for (int chunks = 1; chunks <= np; chunks++) {
      const int z = (np + chunks - 1) / chunks;
      if (z < 8) break;
      const long W = (long)per * ((np + z - 1) / z), rounds = (W + SX - 1) / SX;
      double cost = (double)rounds * (z + 2) * (1.0 + 0.3 / (double)rounds);
      if (W < SX) cost = (z + 2) * 1.3;
      if (chforce > 0) cost = (chunks == chforce) ? 0.0 : 1e30;
      if (best < 0 || cost < best) { best = cost; zc = z; }
    }
  }
  if (zc > np) zc = np;
  const int nch = (np + zc - 1) / zc;
  const unsigned nb = (unsigned)(8 * per * nch);
  if (nb > WL_MAXPART) { wl_set_error("resjac: too many workgroups for the reduction workspace"); return WL_EINVAL; }
  if (shell) hipLaunchKernelGGL(k_scale_shell, dim3(8, (unsigned)g.nz), dim3(WL_BLOCK), 0, s, g, xout, x, dt);
  RjBc bc{0, {0.f, 0.f, 0.f}};
  if (bcU) { if (g.nz != g.gnz) { wl_set_error("resjac: BC! on load is for the single domain"); return WL_EINVAL; } bc.on = 1; for (int a = 0; a < 3; a++) bc.U[a] = bcU[a]; }
  if (r16) rj16::rj_launch(nb, s, g, xout, rout, x, u, dt, w, cl, zc, ws.pa, ws.pb, ws.pm, bc);
  else rj32::rj_launch(nb, s, g, xout, rout, x, u, dt, w, cl, zc, ws.pa, ws.pb, ws.pm, bc);
  hipLaunchKernelGGL(k_resjac_fin, dim3(1), dim3(WL_BLOCK), 0, s, (const double*)ws.pa, (const double*)ws.pb, (const float*)ws.pm, (int)nb, ws.res_d, ws.res_f, slot_d, slot_f);
  WL_LAUNCH_CHECK(); return 0;
}
}  // namespace wl
