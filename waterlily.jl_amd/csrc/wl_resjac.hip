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
#define RJ_X 32
#define RJ_Y 32
#define RJ_N (RJ_X * RJ_Y)
#define RJ_W (RJ_X + 2)
#define RJ_H ((RJ_Y + 2) * RJ_W)
#define RJ_SZ (2 * RJ_H)
#define RJ_HX 2
#define RJ_HY 2
#define RJ_CX (2 * RJ_X - 2 * RJ_HX)   // 60 core cells
#define RJ_CY (RJ_Y - 2 * RJ_HY)       // 28 core rows

__device__ __forceinline__ float rj_cf(int Ia, int Na, float c) { return (Ia <= 2 || Ia >= Na) ? 0.f : c; }
__device__ __forceinline__ float rj_inv(float d) { return (d == 0.f) ? d : 1.0f / d; }
__device__ __forceinline__ float2 rj_ld2(const float* __restrict__ p, unsigned o) { return *reinterpret_cast<const float2*>(p + o); }
__device__ __forceinline__ void rj_st2(float* __restrict__ p, unsigned o, float2 v, bool s0, bool s1) {
  if (s0 && s1) *reinterpret_cast<float2*>(p + o) = v;
  else { if (s0) p[o] = v.x; if (s1) p[o + 1] = v.y; }
}
int g_resjac_on = 1;
long g_resjac_min = 8L << 20;   // cells: below this the extra host read of Σr costs more than the fusion saves

__global__ void __launch_bounds__(RJ_N, 8) k_resjac(GridX g, float* __restrict__ xout, float* __restrict__ rout, const float* __restrict__ p, const float* __restrict__ u,
                                                    float dt, float w, wl::ConstL cl, int zchunk, double* __restrict__ psum, double* __restrict__ pl1, float* __restrict__ pmax) {
  __shared__ float sX[2][RJ_SZ];   // x' = x·dt of the plane whose residual is evaluated (even-x array, then odd-x array)
  __shared__ float sE[2][RJ_SZ];   // ϵ = r·iD of the plane that is relaxed
  const int ntx = (g.nx - 1 + RJ_CX - 1) / RJ_CX, nty = (g.ny - 2 + RJ_CY - 1) / RJ_CY;
  const int ntiles = ntx * nty;
  const unsigned h = blockIdx.x, q = h & 7u, sblk = h >> 3;
  const unsigned per = (unsigned)((ntiles + 7) >> 3);      // XCD q walks a contiguous range of tiles
  const int c = (int)(sblk / per);
  const int tl = (int)(q * per + (sblk - (unsigned)c * per));
  double nsum = 0.0, nl1 = 0.0; float nmax = 0.f;
  const int ks = g.k0 + c * zchunk, ke = (ks + zchunk < g.k1) ? ks + zchunk : g.k1;
  if (tl < ntiles && ks < ke) {
    const int tx = tl % ntx, ty = tl / ntx;
    const int lx = threadIdx.x % RJ_X, ly = threadIdx.x / RJ_X;
    const int i0 = tx * RJ_CX - RJ_HX + 2 * lx, j = 1 + ty * RJ_CY - RJ_HY + ly;       // cells (i0, i0+1) of row j, 0-based with ghosts; i0 even
    const int lq = (ly + 1) * RJ_W + lx + 1;
    const bool indom = i0 >= 0 && i0 <= g.nx - 2 && j >= 0 && j < g.ny;
    const bool jin = j >= 1 && j <= g.ny - 2;
    const bool in0 = indom && jin && i0 >= 2, in1 = indom && jin && i0 + 1 <= g.nx - 2;  // interior cells
    const bool corep = 2 * lx >= RJ_HX && 2 * lx < 2 * RJ_X - RJ_HX && ly >= RJ_HY && ly < RJ_Y - RJ_HY;
    const bool st0 = corep && in0, st1 = corep && in1;
    const unsigned oc = indom ? (unsigned)i0 + (unsigned)j * (unsigned)g.sy : 0u;
    const unsigned sz = (unsigned)g.sz, cs = (unsigned)g.cs;
    for (int qq = threadIdx.x; qq < RJ_SZ; qq += RJ_N) { sX[0][qq] = 0.f; sX[1][qq] = 0.f; sE[0][qq] = 0.f; sE[1][qq] = 0.f; }
    // in-plane face coefficients of the pair (wall faces: 0) and the partial diagonals  — set_diag!'s order, src/Poisson.jl:49-55
    const float cxa = rj_cf(i0 + 1, g.nx, cl.c[0]), cxb = rj_cf(i0 + 2, g.nx, cl.c[0]), cxc = rj_cf(i0 + 3, g.nx, cl.c[0]);
    const float ky = rj_cf(j + 1, g.ny, cl.c[1]), kyp = rj_cf(j + 2, g.ny, cl.c[1]);
    float dxy0 = 0.f; dxy0 -= (cxa + cxb); dxy0 -= (ky + kyp);
    float dxy1 = 0.f; dxy1 -= (cxb + cxc); dxy1 -= (ky + kyp);
    const float c2 = cl.c[2];
    const int oth0 = RJ_H + lq - 1, oth1 = lq + 1;     // x-neighbour outside the pair: odd cell of the left thread / even cell of the right thread
    auto plane_ok = [&](int K) { return K >= 0 && K <= g.nz - 1; };
    auto ldp = [&](int K) -> float2 {                  // x' = x·dt of plane K (all cells of the array; 0 outside it)
      if (!(indom && plane_ok(K))) return make_float2(0.f, 0.f);
      const float2 v = rj_ld2(p, oc + (unsigned)K * sz);
      return make_float2(v.x * dt, v.y * dt);
    };
    auto lduz = [&](int K) -> float2 { return (indom && plane_ok(K) && (in0 || in1)) ? rj_ld2(u, 2u * cs + oc + (unsigned)K * sz) : make_float2(0.f, 0.f); };
    const int K0 = ks - 1;
    float2 xm = ldp(K0 - 1), x0 = ldp(K0), xp;
    float2 uz0 = lduz(K0), uzp;
    float2 em1 = {0.f, 0.f}, e0 = em1, e1 = em1;       // ϵ of planes K−2, K−1, K
    float2 r1 = em1, r0 = em1;                         // r of planes K−1, K
    __syncthreads();                                                // (the zero fill above is by other threads)
    sX[K0 & 1][lq] = x0.x; sX[K0 & 1][lq + RJ_H] = x0.y;
    for (int K = K0; K <= ke; K++) {
      // ---- loads of this step: x', u_z of plane K+1 (the z-neighbours), u_x, u_y of plane K
      xp = ldp(K + 1); uzp = lduz(K + 1);
      const bool planeK = K >= g.k0 && K < g.k1;
      float2 ux = {0.f, 0.f}, uy = ux, uyp = ux; float uxr = 0.f;
      if (planeK && (in0 || in1)) {
        const unsigned o = oc + (unsigned)K * sz;
        ux = rj_ld2(u, o); uxr = u[o + 2];
        uy = rj_ld2(u, cs + o); uyp = rj_ld2(u, cs + o + (unsigned)g.sy);
      }
      __syncthreads();                                              // x'(K) and ϵ(K−1) of the previous step are complete; its readers are done
      sX[(K + 1) & 1][lq] = xp.x; sX[(K + 1) & 1][lq + RJ_H] = xp.y;
      // ---- residual! on plane K          r = iD==0 ? 0 : z − A·x'   (k_div_residual's statements)
      const float lz = rj_cf(g.gk + K + 1, g.gnz, c2), lzp = rj_cf(g.gk + K + 2, g.gnz, c2);    // z-faces below / above plane K
      const float zs = lz + lzp;
      const float d0 = dxy0 - zs, d1 = dxy1 - zs;                   // D of the two cells
      const float id0 = rj_inv(d0), id1 = rj_inv(d1);
      r1 = r0; em1 = e0; e0 = e1;
      r0 = make_float2(0.f, 0.f); e1 = r0;
      if (planeK) {
        const float* __restrict__ SX = sX[K & 1];
        if (in0) {
          float dv = 0.f;
          dv += ux.y - ux.x;
          dv += uyp.x - uy.x;
          dv += uzp.x - uz0.x;
          float s = x0.x * d0;
          s += (SX[oth0] * cxa + x0.y * cxb);
          s += (SX[lq - RJ_W] * ky + SX[lq + RJ_W] * kyp);
          s += (xm.x * lz + xp.x * lzp);
          r0.x = (d0 == 0.f) ? 0.f : dv - s;
          e1.x = r0.x * id0;
        }
        if (in1) {
          float dv = 0.f;
          dv += uxr - ux.y;
          dv += uyp.y - uy.y;
          dv += uzp.y - uz0.y;
          float s = x0.y * d1;
          s += (x0.x * cxb + SX[oth1] * cxc);
          s += (SX[RJ_H + lq - RJ_W] * ky + SX[RJ_H + lq + RJ_W] * kyp);
          s += (xm.y * lz + xp.y * lzp);
          r0.y = (d1 == 0.f) ? 0.f : dv - s;
          e1.y = r0.y * id1;
        }
        if (K >= ks && K < ke) {                                    // the planes this workgroup owns: Σr, L₁, L∞ of its core cells
          const float a0 = st0 ? r0.x : 0.f, a1 = st1 ? r0.y : 0.f;
          nsum += (double)a0; nsum += (double)a1;
          nl1 += (double)fabsf(a0); nl1 += (double)fabsf(a1); nmax = fmaxf(nmax, fmaxf(fabsf(a0), fabsf(a1)));
        }
      }
      sE[K & 1][lq] = e1.x; sE[K & 1][lq + RJ_H] = e1.y;
      // ---- Jacobi! on plane P = K−1          ϵ = r·iD ; r −= ω·Aϵ ; x += ω·ϵ   (k_jacobi_march_cl's statements, no pending shift)
      const int P = K - 1;
      if (P >= ks && P < ke && (st0 || st1)) {
        const float* __restrict__ SE = sE[P & 1];
        const float lzP = rj_cf(g.gk + P + 1, g.gnz, c2), lzpP = lz;          // z-faces below / above plane P (the upper one is plane K's lower)
        const float zsP = lzP + lzpP;
        float2 rn, xn;
        {
          float s = e0.x * (dxy0 - zsP);
          s += (SE[oth0] * cxa + e0.y * cxb);
          s += (SE[lq - RJ_W] * ky + SE[lq + RJ_W] * kyp);
          s += (em1.x * lzP + e1.x * lzpP);
          rn.x = r1.x - w * s; xn.x = xm.x + w * e0.x;
        }
        {
          float s = e0.y * (dxy1 - zsP);
          s += (e0.x * cxb + SE[oth1] * cxc);
          s += (SE[RJ_H + lq - RJ_W] * ky + SE[RJ_H + lq + RJ_W] * kyp);
          s += (em1.y * lzP + e1.y * lzpP);
          rn.y = r1.y - w * s; xn.y = xm.y + w * e0.y;
        }
        const unsigned oP = oc + (unsigned)P * sz;
        rj_st2(rout, oP, rn, st0, st1);
        rj_st2(xout, oP, xn, st0, st1);
      }
      xm = x0; x0 = xp; uz0 = uzp;
    }
  }
  // one partial (Σr, Σ|r|, max|r|) per workgroup
  __shared__ double shs[RJ_N / 64], shl[RJ_N / 64]; __shared__ float shm[RJ_N / 64];
  nsum = wave_sum(nsum); nl1 = wave_sum(nl1); nmax = wave_max(nmax);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { shs[threadIdx.x >> 6] = nsum; shl[threadIdx.x >> 6] = nl1; shm[threadIdx.x >> 6] = nmax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, l = 0.0; float mx = 0.f;
    for (int qq = 0; qq < RJ_N / 64; qq++) { a += shs[qq]; l += shl[qq]; mx = fmaxf(mx, shm[qq]); }
    psum[blockIdx.x] = a; pl1[blockIdx.x] = l; pmax[blockIdx.x] = mx;
  }
}
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
// the fused head is worth it (and implemented) for: 3-D single-domain constant-coefficient finest levels of at least resjac_min cells
bool resjac_ok(const GridX& g, const ConstL& cl) {
  return g_resjac_on && cl.on && g.D == 3 && g.nz == g.gnz && g.gk == 0 && (g.nx & 1) == 0 && g.nx >= 66 && g.ny >= 34 && g.nz >= 10 && g.cs < (1L << 30) &&
         (long)(g.nx - 2) * (g.ny - 2) * (g.nz - 2) >= g_resjac_min;
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
int resjac(float* xout, float* rout, const float* x, const float* u, const GridX& g, float dt, float w, const ConstL& cl, const RedWs& ws, int slot_d, int slot_f, hipStream_t s, bool shell) {
  if (xout == x) { wl_set_error("resjac: output aliases input"); return WL_EINVAL; }
  const int ntiles = ((g.nx - 1 + RJ_CX - 1) / RJ_CX) * ((g.ny - 2 + RJ_CY - 1) / RJ_CY), per = (ntiles + 7) >> 3;
  const int np = g.k1 - g.k0;
  static const int envc = getenv("WL_RJ_CHUNK") ? atoi(getenv("WL_RJ_CHUNK")) : 0;
  int zc = envc;
  if (zc <= 0) {
    // equal workgroups run in rounds of 32 CUs × 2 resident workgroups per XCD (tile ranges are dealt XCD by XCD): minimise
    // rounds × (planes per chunk + the 2 warm-up planes), with a balance penalty for few rounds — as wl_fused2's zchunk2.  The former fixed target of
    // 3072 workgroups gave 6.2 rounds at 512³, i.e. a seventh, almost empty one.
    static const int chforce = getenv("WL_RJ_CHUNKS") ? atoi(getenv("WL_RJ_CHUNKS")) : 0;     // experiments: number of chunks
    double best = -1.0; zc = np;
    for (int chunks = 1; chunks <= np; chunks++) {
      const int z = (np + chunks - 1) / chunks;
      if (z < 8) break;
      const long W = (long)per * ((np + z - 1) / z), rounds = (W + 63) / 64;
      double cost = (double)rounds * (z + 2) * (1.0 + 0.3 / (double)rounds);
      if (W < 64) cost = (z + 2) * 1.3;
      if (chforce > 0) cost = (chunks == chforce) ? 0.0 : 1e30;
      if (best < 0 || cost < best) { best = cost; zc = z; }
    }
  }
  if (zc > np) zc = np;
  const int nch = (np + zc - 1) / zc;
  const unsigned nb = (unsigned)(8 * per * nch);
  if (nb > WL_MAXPART) { wl_set_error("resjac: too many workgroups for the reduction workspace"); return WL_EINVAL; }
  if (shell) hipLaunchKernelGGL(k_scale_shell, dim3(8, (unsigned)g.nz), dim3(WL_BLOCK), 0, s, g, xout, x, dt);
  hipLaunchKernelGGL(k_resjac, dim3(nb), dim3(RJ_N), 0, s, g, xout, rout, x, u, dt, w, cl, zc, ws.pa, ws.pb, ws.pm);
  hipLaunchKernelGGL(k_resjac_fin, dim3(1), dim3(WL_BLOCK), 0, s, (const double*)ws.pa, (const double*)ws.pb, (const float*)ws.pm, (int)nb, ws.res_d, ws.res_f, slot_d, slot_f);
  WL_LAUNCH_CHECK(); return 0;
}
}  // namespace wl
