// conv_diff! (src/Flow.jl:38-62) + BDIM! for NoBody (src/Flow.jl:176-180): the z-marching, LDS-tiled kernel of wl_convt.hip with every
// face flux evaluated ONCE.
//
// Why (round 3, tools/probe/valu_probe.hip + tools/isa_mix.py): k_conv_tile is bound by vector-instruction issue, not by memory — its
// hot loop is ≈800 VALU instructions per thread and plane, 40 % of them in the slow issue class of gfx950 (v_cndmask, v_med3, v_cmp,
// the f64 round trip of the exact /6: ≈1.8 ns per wave-instruction and SIMD against ≈1.05 for v_add/v_mul/v_fma), which adds up to
// ≈1.1 ms of the 1.2–1.3 ms the kernel takes at 512³.  A thread of k_conv_tile evaluates 27 fluxes per plane for its two cells
// (13.5 per cell) where 9 per cell exist: the upper x-face of the pair and both upper y-faces are evaluated again by the neighbours.
// Here a thread evaluates only the LOWER faces of its two cells (x: 2, y: 2 per component) and the upper z-face (carried to the next
// plane as before) = 18 per plane; it gets
//   * the upper x-face of the pair from the next lane (one DPP move: wave_shl:1 — a wave holds two rows of 32 pairs),
//   * the upper y-faces from the thread one row up through a double-buffered LDS array (one 8-byte write, one 8-byte read per component),
//   * the 240 faces on the tile's upper x and y edges (whose plus-side cells belong to other tiles) from one extra GENERIC flux that
//     threads 0..239 (waves 0..3 — one wave per SIMD) evaluate per plane from LDS operands at per-thread offsets.
// Planes k, k+1 (+ the plane being filled) live in a 3-slot LDS ring with a 2-cell halo; plane k−1 is needed only at the thread's own
// cells (f[I−δz], u_x[F−δz], u_y[F−δz]) and stays in registers.  One barrier per plane, as before: the ring slot written during iteration k
// (plane k+2) was last read before the previous barrier, the flux arrays alternate.
// Operands, statement order per face and accumulation order per cell are those of k_conv_tile / k_conv_diff / the reference
// (r += Φ(I) then r −= Φ(I+δ), direction b inner) — a flux taken from a neighbour is the very value this thread would have computed —
// so the results are bit-identical.  Scope as k_conv_tile: D = 3, no periodic direction, BDIM! fused (NoBody), f not stored.
#include <cstdlib>
#include <type_traits>

#include "wl_bcfold.hpp"
#include "wl_conv_cell.hpp"

namespace {
#define CF_TX 32                 // threads along x (two cells each)
#define CF_TY 16                 // threads (= rows) along y
#define CF_N (CF_TX * CF_TY)     // 512 threads
#define CF_CX (2 * CF_TX)        // 64 core cells along x
#define CF_CY CF_TY              // 16 core rows
#define CF_W (CF_CX + 4)         // LDS row: core + 2 halo cells per side
#define CF_H (CF_CY + 4)
#define CF_P (CF_W * CF_H)       // floats per component-plane (1360)
#define CF_SLOT (3 * CF_P)       // floats per plane slot
#define CF_NSLOT 3               // ring: planes k, k+1 are read while k+2 is written
#define CF_HPC (CF_P / 2 - CF_CX / 2 * CF_CY)   // halo float2-pairs per component-plane (168)
#define CF_FYC ((CF_CY + 1) * CF_CX)            // lower y-face fluxes of rows 0..16 of one component (row 16: generic)
#define CF_FY (3 * CF_FYC)
#define CF_FX (3 * CF_CY)                       // fluxes of the faces x0+64 (generic), [component][row]
#define CF_FBUF (CF_FY + CF_FX)                 // one flux buffer (3312 floats)
#define CF_NGEN (3 * CF_CX + 3 * CF_CY)         // 240 generic faces per plane
#define CF_LDS (CF_NSLOT * CF_SLOT + 2 * CF_FBUF)   // 18864 floats = 75,456 B: two workgroups per CU

struct __attribute__((aligned(4))) F2v { float x, y; };   // 8-byte global access that is only 4-byte aligned (tiles start at an odd cell)
__device__ __forceinline__ float2 cf_ldg2(const float* __restrict__ p, unsigned o) { const F2v t = *reinterpret_cast<const F2v*>(p + o); return make_float2(t.x, t.y); }
__device__ __forceinline__ void cf_stg2(float* __restrict__ p, unsigned o, float2 v) { F2v t; t.x = v.x; t.y = v.y; *reinterpret_cast<F2v*>(p + o) = t; }
__device__ __forceinline__ float2 cf_lds2(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ float cf_sel(const float2& v, int e) { return e ? v.y : v.x; }
// value of the next lane (lane 63: unchanged)
__device__ __forceinline__ float cf_next_lane(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}

// see wl_convt.hip: a pair (X, X+1) of row Y whose elements may lie outside the array
struct CfPair { unsigned off; int mode; };
__device__ __forceinline__ CfPair cf_pair_addr(int X, int Y, int nx, int ny, unsigned sy) {
  CfPair r;
  const int Yc = Y < 0 ? 0 : (Y > ny - 1 ? ny - 1 : Y);
  int Xc = X, mode = 0;
  if (X < 0) { Xc = X + 1; mode = 1; }
  else if (X + 1 > nx - 1) { Xc = X - 1; mode = 2; }
  if (Xc < 0 || Xc + 1 > nx - 1 || Y != Yc) { mode = 3; Xc = Xc < 0 ? 0 : (Xc + 1 > nx - 1 ? nx - 2 : Xc); }
  r.off = (unsigned)Xc + (unsigned)Yc * sy; r.mode = mode;
  return r;
}
__device__ __forceinline__ float2 cf_pair_fix(float2 t, int mode) { return make_float2(mode == 2 ? t.y : t.x, mode == 1 ? t.x : t.y); }

// Φ at the face whose plus-side cell is F:  a = f[F−2δ], b = f[F−δ], c = f[F], d = f[F+δ];  U = advecting velocity at the face.
// wl: F is the first interior cell (the reference's ϕuL form), wu: F is the upper ghost (ϕuR).      src/Flow.jl:8-11,47-57
template <int SCH, int WALLS>
__device__ __forceinline__ float cf_flux(float U, float a, float b, float c, float d, bool wl, bool wu, float nu) {
  const bool pos = U > 0;
  float X = lam<SCH>(pos ? a : d, pos ? b : c, pos ? c : b);
  if (WALLS) {
    const bool use_avg = (wl && pos) || (wu && (U < 0));
    X = use_avg ? (c + b) / 2 : X;
  }
  return U * X - nu * (c - b);
}

int g_convf_on = 1;

// MODE: how BDIM!'s u_out = (u·pre + μ₀·f)·post is evaluated — 1: pre = 0 and post = 1 (predictor: u_out = 0 + μ₀·f), 2: pre ≠ 0 and post ≠ 1
// (corrector), 0: decided at run time per value (two selects per value: 5 % of the kernel's vector instructions).  Same arithmetic in all three.
// PROJ: the advecting field is read through mom_project!'s tail and BC! (src/Flow.jl:227-230, src/core.jl:200-219): u holds the UNPROJECTED
// predictor velocity u*, bd.px the solver's x (pressure·Δt); every value that enters the LDS planes is
//     U_a                                   where component a is normal to a boundary face (index 0, 1, N−1 along a),
//     u*_a[c] − c_a·(x[c] − x[c−δa])        elsewhere, c = the cell with the two other coordinates clamped into the interior
// — the statements of k_project_unscale (constant coefficients: L = c_a away from the wall faces) followed by BC!'s closed form for a tuple U
// (wl_bcfold.hpp), so the values are bit for bit those the separate tail + BC! launches would have left in memory; the projected predictor
// velocity is never written (−24 B/cell of the step's traffic and one launch).  FULL tiles, single domain, no periodic direction / exit / body.
template <int SCH, int FULL, int U0ADV, int MODE, int PROJ>
__global__ void __launch_bounds__(CF_N, 4) k_conv_flux(GridX g, const float* __restrict__ u, float nu, int ka, int kb, int zchunk, BdimArgs bd) {
  __shared__ float lds[CF_LDS];
  const int ntx = (g.nx - 2 + CF_CX - 1) / CF_CX, nty = (g.ny - 2 + CF_CY - 1) / CF_CY;
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
  const int x0 = 1 + tx * CF_CX, y0 = 1 + ty * CF_CY;       // first core cell (0-based, ghosts included)
  const int tid = threadIdx.x, lx = tid & (CF_TX - 1), ly = tid >> 5;
  const int x = x0 + 2 * lx, y = y0 + ly;                    // the pair (x, x+1) of row y
  const int my = (ly + 2) * CF_W + 2 + 2 * lx;               // LDS index of cell 0 inside a component-plane (even)
  const bool in0 = FULL || (y <= g.ny - 2 && x <= g.nx - 2), in1 = FULL || (y <= g.ny - 2 && x + 1 <= g.nx - 2);   // interior cell (stored)
  const unsigned cs = (unsigned)g.cs, sz = (unsigned)g.sz, sy = (unsigned)g.sy;
  CfPair pc;                                                 // centre pair
  if (FULL) { pc.off = (unsigned)x + (unsigned)y * sy; pc.mode = 0; } else pc = cf_pair_addr(x, y, g.nx, g.ny, sy);
  // one halo pair per thread: 168 pairs per component-plane × 3 components = 504; threads 504..511 repeat pair 503 (same value, benign)
  const int hid = tid < 3 * CF_HPC ? tid : 3 * CF_HPC - 1;
  const int hcmp = hid / CF_HPC, hh = hid - hcmp * CF_HPC;
  int R, col;
  if (hh < 4 * (CF_W / 2)) { const int rr = hh / (CF_W / 2); const int m = hh - rr * (CF_W / 2); R = rr < 2 ? rr : rr + CF_CY; col = 2 * m; }
  else { const int h2 = hh - 4 * (CF_W / 2); R = 2 + (h2 >> 1); col = (h2 & 1) ? CF_W - 2 : 0; }
  const CfPair ph = cf_pair_addr(x0 - 2 + col, y0 - 2 + R, g.nx, g.ny, sy);
  const unsigned hbase = (unsigned)hcmp * cs + ph.off;
  const int hl = hcmp * CF_P + R * CF_W + col;
  // ---- the generic face of this thread (threads 0..239): component ga, direction gb (0: the face x0+64 of row gr; 1: the face y0+16 of column gj)
  const bool gen = tid < CF_NGEN;
  int ga, gF, gsb, gb, gout; bool gwu;
  if (tid < 3 * CF_CX) { ga = tid >> 6; const int gj = tid & 63; gF = (2 + CF_CY) * CF_W + 2 + gj; gsb = CF_W; gb = 1; gwu = (y0 + CF_CY == g.ny - 1); gout = ga * CF_FYC + CF_CY * CF_CX + gj; }
  else { const int t = (tid - 3 * CF_CX) & 63; ga = t >> 4; const int gr = t & 15; gF = (2 + gr) * CF_W + 2 + CF_CX; gsb = 1; gb = 0; gwu = (x0 + CF_CX == g.nx - 1); gout = CF_FY + (ga > 2 ? 2 : ga) * CF_CY + gr; }
  if (ga > 2) ga = 2;                                        // (threads ≥ 240 never use these)
  const int gf = ga * CF_P + gF;                             // f[F] of the advected component
  const int gU0 = gb * CF_P + gF;                            // u_b[F]
  const int gUm = gU0 - (ga == 0 ? 1 : CF_W);                // u_b[F−δa] for a = x, y (a = z: the previous plane's u_b[F], carried in gprev)
  for (int i = tid; i < CF_LDS; i += CF_N) lds[i] = 0.f;
  __syncthreads();

  struct Stage { float2 c[3]; float2 h; };
  // PROJ: the two elements of the halo pair are loaded one by one from the cells their BC!-clamped coordinates name
  unsigned ho[2] = {0u, 0u}, hnb = 0u; bool hd[2] = {false, false}; float hU = 0.f, hc = 0.f;
  const float* __restrict__ px = bd.px;
  if (PROJ) {
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const int Xe = x0 - 2 + col + e, Ye = y0 - 2 + R;
      const int Xc = Xe < 1 ? 1 : (Xe > g.nx - 2 ? g.nx - 2 : Xe), Yc = Ye < 1 ? 1 : (Ye > g.ny - 2 ? g.ny - 2 : Ye);
      ho[e] = (unsigned)Xc + (unsigned)Yc * sy;
      hd[e] = hcmp == 0 ? (Xe <= 1 || Xe >= g.nx - 1) : (hcmp == 1 ? (Ye <= 1 || Ye >= g.ny - 1) : false);
    }
    hnb = hcmp == 0 ? 1u : (hcmp == 1 ? sy : sz);
    hU = hcmp == 0 ? bd.bcU[0] : (hcmp == 1 ? bd.bcU[1] : bd.bcU[2]);
    hc = hcmp == 0 ? bd.cl_c[0] : (hcmp == 1 ? bd.cl_c[1] : bd.cl_c[2]);
  }
  // planes outside the local array (k = −1 below the first plane, nz above the last) are never used either: clamp.  PROJ: the ghost planes 0 and nz−1
  // take the tangential components of planes 1 and nz−2 (BC!), so every load goes to an interior plane
  auto load_plane = [&](int kk) -> Stage {
    const int kc = PROJ ? (kk < 1 ? 1 : (kk > g.nz - 2 ? g.nz - 2 : kk)) : (kk < 0 ? 0 : (kk > g.nz - 1 ? g.nz - 1 : kk));
    const unsigned ko = (unsigned)kc * sz;
    Stage st;
#pragma unroll
    for (int cc = 0; cc < 3; cc++) st.c[cc] = cf_ldg2(u, (unsigned)cc * cs + ko + pc.off);
    if (PROJ) { st.h.x = u[(unsigned)hcmp * cs + ko + ho[0]]; st.h.y = u[(unsigned)hcmp * cs + ko + ho[1]]; }
    else st.h = cf_ldg2(u, hbase + ko);
    return st;
  };
  // x at the cells the projection of one plane reads: the pair, its lower x / y / z neighbours, and the same for the two halo elements
  struct XStage { float2 xc, xy, xk; float xm; float xh[2], xn[2]; };
  auto load_x = [&](int kk) -> XStage {
    const int kc = kk < 1 ? 1 : (kk > g.nz - 2 ? g.nz - 2 : kk);
    const unsigned ko = (unsigned)kc * sz;
    XStage X;
    X.xc = cf_ldg2(px, ko + pc.off); X.xm = px[ko + pc.off - 1u]; X.xy = cf_ldg2(px, ko + pc.off - sy); X.xk = cf_ldg2(px, ko - sz + pc.off);
#pragma unroll
    for (int e = 0; e < 2; e++) { X.xh[e] = px[ko + ho[e]]; X.xn[e] = px[ko + ho[e] - hnb]; }
    return X;
  };
  const bool dxc0 = (x == 1), dyc = (y == 1);      // the pair's first cell lies on the lower x wall face / the row on the lower y wall face
  auto project = [&](Stage& st, const XStage& X, int kk) {
    const bool zd = kk <= 1 || kk >= g.nz - 1;     // the z component is U on these planes
    const float c0 = bd.cl_c[0], c1 = bd.cl_c[1], c2 = bd.cl_c[2];
    const float p0x = st.c[0].x - c0 * (X.xc.x - X.xm), p0y = st.c[0].y - c0 * (X.xc.y - X.xc.x);
    const float p1x = st.c[1].x - c1 * (X.xc.x - X.xy.x), p1y = st.c[1].y - c1 * (X.xc.y - X.xy.y);
    const float p2x = st.c[2].x - c2 * (X.xc.x - X.xk.x), p2y = st.c[2].y - c2 * (X.xc.y - X.xk.y);
    st.c[0] = make_float2(dxc0 ? bd.bcU[0] : p0x, p0y);
    st.c[1] = make_float2(dyc ? bd.bcU[1] : p1x, dyc ? bd.bcU[1] : p1y);
    st.c[2] = make_float2(zd ? bd.bcU[2] : p2x, zd ? bd.bcU[2] : p2y);
    const float h0 = st.h.x - hc * (X.xh[0] - X.xn[0]), h1 = st.h.y - hc * (X.xh[1] - X.xn[1]);
    const bool hz = hcmp == 2 && zd;
    st.h = make_float2((hd[0] || hz) ? hU : h0, (hd[1] || hz) ? hU : h1);
  };
  auto slot_of = [&](int kk) -> float* { return lds + ((unsigned)(kk + 3) % 3u) * CF_SLOT; };
  auto write_plane = [&](int kk, const Stage& st) {
    float* sl = slot_of(kk);
#pragma unroll
    for (int cc = 0; cc < 3; cc++) *reinterpret_cast<float2*>(sl + cc * CF_P + my) = FULL ? st.c[cc] : cf_pair_fix(st.c[cc], pc.mode);
    *reinterpret_cast<float2*>(sl + hl) = PROJ ? st.h : cf_pair_fix(st.h, ph.mode);
  };
  // wall flags of the x and y faces (0-based cell index 1 = first interior cell, n−1 = upper ghost)
  const bool wlx = (x == 1);
  const bool wux0 = (x == g.nx - 1), wux1 = (x + 1 == g.nx - 1);
  const bool wly = (y == 1), wuy0 = (y == g.ny - 1);
  float zf[3][2];     // Φ at the lower z-face of the pair, per component (carried from the previous plane)
  float2 Zm1[3];      // plane k−1 at the thread's own cells
  float gprev = 0.f;  // generic face: u_b[F] of plane k−1

  // Φ at the face k+1 (upper z-face of plane k = lower z-face of plane k+1) for the three components of the pair;  U = (u_z[F] + u_z[F−δa])/2 on plane k+1
  auto zfaces = [&](auto ztag, int k, const float2* Zm, const float2* C1, const float2* Zp1, const float2* Zp2, const float* Pp, float (*Pz)[2]) {
    const int Kg = g.gk + k;
    const bool wlz = (Kg + 1 == 1), wuz = (Kg + 1 == g.gnz - 1);
    const float2 Exz = cf_lds2(Pp + 2 * CF_P - 2);      // u_z(x−2.., y, k+1): .y = u_z(x−1)
    const float2 Eyz = cf_lds2(Pp + 2 * CF_P - CF_W);   // u_z(x.., y−1, k+1)
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
      for (int e = 0; e < 2; e++) {
        float Uz;
        if (a == 0) Uz = (cf_sel(Zp1[2], e) + (e ? Zp1[2].x : Exz.y)) / 2;
        else if (a == 1) Uz = (cf_sel(Zp1[2], e) + cf_sel(Eyz, e)) / 2;
        else Uz = (cf_sel(Zp1[2], e) + cf_sel(C1[2], e)) / 2;
        Pz[a][e] = cf_flux<SCH, decltype(ztag)::value>(Uz, cf_sel(Zm[a], e), cf_sel(C1[a], e), cf_sel(Zp1[a], e), cf_sel(Zp2[a], e), wlz, wuz, nu);
      }
    }
  };
  Stage S; XStage XS;
  // ---- prologue: planes ks−1, ks, ks+1 → LDS; plane ks−2 only at the own cells; plane ks+2 in flight.  Priming = the z-face fluxes of the first plane's lower faces
  {
    float2 m2[3];
    if (PROJ) { Stage sm = load_plane(ks - 2); const XStage xm = load_x(ks - 2); project(sm, xm, ks - 2); for (int cc = 0; cc < 3; cc++) m2[cc] = sm.c[cc]; }
    else {
      const unsigned ko = (unsigned)(ks - 2 < 0 ? 0 : ks - 2) * sz;
#pragma unroll
      for (int cc = 0; cc < 3; cc++) { m2[cc] = cf_ldg2(u, (unsigned)cc * cs + ko + pc.off); if (!FULL) m2[cc] = cf_pair_fix(m2[cc], pc.mode); }
    }
    Stage s0 = load_plane(ks - 1), s1 = load_plane(ks), s2 = load_plane(ks + 1);
    if (PROJ) {
      const XStage x0s = load_x(ks - 1), x1s = load_x(ks), x2s = load_x(ks + 1);
      project(s0, x0s, ks - 1); project(s1, x1s, ks); project(s2, x2s, ks + 1);
    }
    S = load_plane(ks + 2);
    if (PROJ) XS = load_x(ks + 2);
    write_plane(ks - 1, s0); write_plane(ks, s1); write_plane(ks + 1, s2);
    float2 C1[3], Zp1[3], Zp2[3];
#pragma unroll
    for (int cc = 0; cc < 3; cc++) {
      C1[cc] = FULL ? s0.c[cc] : cf_pair_fix(s0.c[cc], pc.mode); Zp1[cc] = FULL ? s1.c[cc] : cf_pair_fix(s1.c[cc], pc.mode); Zp2[cc] = FULL ? s2.c[cc] : cf_pair_fix(s2.c[cc], pc.mode);
    }
    __syncthreads();
    zfaces(std::integral_constant<int, 1>{}, ks - 1, m2, C1, Zp1, Zp2, slot_of(ks) + my, zf);
    if (gen) gprev = slot_of(ks - 1)[gU0];
#pragma unroll
    for (int cc = 0; cc < 3; cc++) Zm1[cc] = C1[cc];
    __syncthreads();       // nobody still reads plane ks−1 when the first iteration overwrites its slot
  }
  const float dtv = bd.dt_dev ? *bd.dt_dev : bd.dt;      // (block-uniform scalar load)
  const int N[3] = {g.nx, g.ny, g.gnz};
  // μ₀ along x and y for the two cells (loop invariant; used by the wall tiles only)      Julia indices of the cells: (x+1, y+1), (x+2, y+1)
  const float mw[2][2] = {{wl::wl_cl_coef(x + 1, N[0], bd.cl_c[0]), wl::wl_cl_coef(x + 2, N[0], bd.cl_c[0])}, {wl::wl_cl_coef(y + 1, N[1], bd.cl_c[1]), wl::wl_cl_coef(y + 1, N[1], bd.cl_c[1])}};
  const bool tile_walls = tx == 0 || tx == ntx - 1 || ty == 0 || ty == nty - 1;
  // The results of plane k are stored at the top of iteration k+1, AFTER that iteration's loads have been issued: every wait on
  // the vector-memory counter (in order, loads and stores alike) then only ever covers operations issued a whole plane earlier.
  float2 un[3];
  auto store_plane = [&](auto wtag, int kq) {
    const unsigned kqo = (unsigned)kq * sz;
    if (decltype(wtag)::value && SCH != WL_VANLEER && bd.bc_on && (x <= 1 || x + 1 >= g.nx - 2 || y <= 1 || y >= g.ny - 2)) {   // (vanLeer: BC! stays a launch, as in k_conv_tile)
      // threads on an x/y wall: BC!(u_out, U) in x and y folded into the stores (wl_bcfold.hpp); the z ghost planes are completed by k_bc_zplanes after the launch
      if (in0) { const float v[3] = {un[0].x, un[1].x, un[2].x}; wl_bc_fold_store_xy(bd.uout, g, x, y, kq, v, bd.bcU); }
      if (in1) { const float v[3] = {un[0].y, un[1].y, un[2].y}; wl_bc_fold_store_xy(bd.uout, g, x + 1, y, kq, v, bd.bcU); }
      return;
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const unsigned oa = (unsigned)a * cs + kqo + pc.off;
      if (FULL) cf_stg2(bd.uout, oa, un[a]);
      else if (pc.mode == 0 && in0 && in1) cf_stg2(bd.uout, oa, un[a]);
      else { const unsigned o0 = (unsigned)a * cs + kqo + (unsigned)x + (unsigned)y * sy; if (in0) bd.uout[o0] = un[a].x; if (in1) bd.uout[o0 + 1] = un[a].y; }
    }
  };
  // Two copies of the main loop, as in k_conv_tile: WALLS = 1 for tiles that touch an x or y wall, WALLS = 0 for the others (no wall forms on x/y faces)
  auto mainloop = [&](auto wtag) {
    constexpr int WALLS = decltype(wtag)::value;
    for (int k = ks; k < ke; k++) {
      // ---- stage: plane k+2 (loaded during the previous iteration) → LDS; its centres are this plane's f[I+2δz]; issue plane k+3
      if (PROJ) project(S, XS, k + 2);
      float2 Zp2[3];
#pragma unroll
      for (int cc = 0; cc < 3; cc++) Zp2[cc] = FULL ? S.c[cc] : cf_pair_fix(S.c[cc], pc.mode);
      write_plane(k + 2, S);
      S = load_plane(k + 3);
      const unsigned ko = (unsigned)k * sz;
      float2 u0v[3];
      if (k > ks) store_plane(wtag, k - 1);
      const float* S0 = slot_of(k);
      const float* P0 = S0 + my;
      const float* Pp = slot_of(k + 1) + my;
      float* FB = lds + CF_NSLOT * CF_SLOT + (k & 1) * CF_FBUF;
      float2 C1[3], Zp1[3];
#pragma unroll
      for (int cc = 0; cc < 3; cc++) { C1[cc] = cf_lds2(P0 + cc * CF_P); Zp1[cc] = cf_lds2(Pp + cc * CF_P); }
      float a0[3], a1[3], Pl1[3], Px0[3];
      {
        float2 CA[3], Ym2[3], Ym1[3], Yp1[3]; float CCx[3];
#pragma unroll
        for (int cc = 0; cc < 3; cc++) {
          CA[cc] = cf_lds2(P0 + cc * CF_P - 2); CCx[cc] = P0[cc * CF_P + 2];
          Ym2[cc] = cf_lds2(P0 + cc * CF_P - 2 * CF_W); Ym1[cc] = cf_lds2(P0 + cc * CF_P - CF_W); Yp1[cc] = cf_lds2(P0 + cc * CF_P + CF_W);
        }
        // rows of u_x and u_y along x: index j ↔ cell x−2+j
        const float rx[4] = {CA[0].x, CA[0].y, C1[0].x, C1[0].y};
        const float ry[4] = {CA[1].x, CA[1].y, C1[1].x, C1[1].y};
#pragma unroll
        for (int a = 0; a < 3; a++) {
          const float r[5] = {CA[a].x, CA[a].y, C1[a].x, C1[a].y, CCx[a]};
          // ---- b = x: faces x and x+1;  U = (u_x[F] + u_x[F−δa])/2                                    src/Flow.jl:3,47
          float Ux[2];
          if (a == 0) { Ux[0] = (rx[2] + rx[1]) / 2; Ux[1] = (rx[3] + rx[2]) / 2; }
          else if (a == 1) { Ux[0] = (rx[2] + Ym1[0].x) / 2; Ux[1] = (rx[3] + Ym1[0].y) / 2; }
          else { Ux[0] = (rx[2] + Zm1[0].x) / 2; Ux[1] = (rx[3] + Zm1[0].y) / 2; }
          const float PxA = cf_flux<SCH, WALLS>(Ux[0], r[0], r[1], r[2], r[3], wlx, wux0, nu);
          const float PxB = cf_flux<SCH, WALLS>(Ux[1], r[1], r[2], r[3], r[4], false, wux1, nu);
          float t0 = 0.f, t1 = 0.f;
          t0 = t0 + PxA; t0 = t0 - PxB;
          t1 = t1 + PxB;
          Px0[a] = PxA;
          // ---- b = y: lower face (row y) of each cell;  U = (u_y[F] + u_y[F−δa])/2
          float Pl[2];
#pragma unroll
          for (int e = 0; e < 2; e++) {
            const float vm2 = cf_sel(Ym2[a], e), vm1 = cf_sel(Ym1[a], e), v0 = cf_sel(C1[a], e), vp1 = cf_sel(Yp1[a], e);
            float Ul;
            if (a == 0) Ul = (ry[2 + e] + ry[1 + e]) / 2;
            else if (a == 1) Ul = (v0 + vm1) / 2;
            else Ul = (cf_sel(C1[1], e) + cf_sel(Zm1[1], e)) / 2;
            Pl[e] = cf_flux<SCH, WALLS>(Ul, vm2, vm1, v0, vp1, wly, wuy0, nu);
          }
          t0 = t0 + Pl[0];
          a0[a] = t0; a1[a] = t1; Pl1[a] = Pl[1];
          *reinterpret_cast<float2*>(FB + a * CF_FYC + ly * CF_CX + 2 * lx) = make_float2(Pl[0], Pl[1]);
        }
      }
      // ---- the generic face (tile's upper x / y edge): waves 0..3
      if (gen) {
        const float fa = S0[gf - 2 * gsb], fb = S0[gf - gsb], fc = S0[gf], fd = S0[gf + gsb];
        const float ub0 = S0[gU0], ubm = S0[gUm];
        const float U = (ub0 + (ga == 2 ? gprev : ubm)) / 2;
        gprev = ub0;
        FB[gout] = cf_flux<SCH, WALLS>(U, fa, fb, fc, fd, false, gwu, nu);
      }
      // ---- b = z: upper face k+1 (it lies on a z wall on two planes of the whole domain: block-uniform branch, the common side carries no wall forms).
      // (Evaluating it after the barrier instead, behind the reads of the neighbours' fluxes, costs registers: +8 % on the kernel, profiles/r03_experiments.md §12.)
      float Pz[3][2];
      const int Kg = g.gk + k;
      const bool zwall = (Kg + 1 == 1 || Kg + 1 == g.gnz - 1);
      if (zwall) zfaces(std::integral_constant<int, 1>{}, k, Zm1, C1, Zp1, Zp2, Pp, Pz);
      else zfaces(std::integral_constant<int, 0>{}, k, Zm1, C1, Zp1, Zp2, Pp, Pz);
      // u⁰ (used after the barrier) and, with PROJ, the pressure values of plane k+3 are requested here, behind the flux arithmetic: their registers are
      // not live across it (8 -> 2 spilled registers in the corrector, conv_diff! 2.21 -> 2.12 ms per step) and the barrier wait covers most of their latency
      if (!U0ADV) {
#pragma unroll
        for (int a = 0; a < 3; a++) { u0v[a] = cf_ldg2(bd.u0, (unsigned)a * cs + ko + pc.off); if (!FULL) u0v[a] = cf_pair_fix(u0v[a], pc.mode); }
      }
      if (PROJ) XS = load_x(k + 3);
      __syncthreads();     // fluxes of this plane and plane k+2 are visible; nobody still reads the ring slot / flux buffer the next iteration overwrites
      float2 PuA[3]; float PxE[3];
#pragma unroll
      for (int a = 0; a < 3; a++) {
        PuA[a] = cf_lds2(FB + a * CF_FYC + (ly + 1) * CF_CX + 2 * lx);
        PxE[a] = 0.f;
        if (lx == CF_TX - 1) PxE[a] = FB[CF_FY + a * CF_CY + ly];
      }
      // ---- upper faces from the neighbours, accumulation in the reference's order, BDIM! (NoBody: μ₁ ≡ 0, V ≡ 0) with scale_u! folded   src/Flow.jl:176-180
      const float mz = wl::wl_cl_coef(Kg + 1, N[2], bd.cl_c[2]);      // block-uniform
#pragma unroll
      for (int a = 0; a < 3; a++) {
        float Px2 = cf_next_lane(Px0[a]);
        if (lx == CF_TX - 1) Px2 = PxE[a];
        const float2 Pu = PuA[a];
        float t0 = a0[a], t1 = a1[a];
        t0 = t0 - Pu.x;
        t1 = t1 - Px2; t1 = t1 + Pl1[a]; t1 = t1 - Pu.y;
        t0 = t0 + zf[a][0]; t0 = t0 - Pz[a][0]; zf[a][0] = Pz[a][0];
        t1 = t1 + zf[a][1]; t1 = t1 - Pz[a][1]; zf[a][1] = Pz[a][1];
        const float acc[2] = {t0, t1};
        const float2 u0a = U0ADV ? C1[a] : u0v[a];
        // μ₀ of a verified NoBody field: 0 on the wall faces of component a, c elsewhere — tiles away from the x/y walls have no wall faces in x and y
        const float m00 = a == 2 ? mz : (WALLS ? mw[a][0] : bd.cl_c[a]), m01 = a == 2 ? mz : (WALLS ? mw[a][1] : bd.cl_c[a]);
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const float fn = cf_sel(u0a, e) + dtv * acc[e] - 0.f;
          const float xx = (0.f / 2 + 0.f) + (e ? m01 : m00) * fn;
          float v;
          if (MODE == 1) v = xx;
          else if (MODE == 2) v = (cf_sel(C1[a], e) * bd.pre + xx) * bd.post;
          else { v = (bd.pre == 0.f) ? xx : (cf_sel(C1[a], e) * bd.pre + xx); if (bd.scale_after) v = v * bd.post; }
          if (e) un[a].y = v; else un[a].x = v;
        }
        Zm1[a] = C1[a];
      }
    }
    store_plane(wtag, ke - 1);
  };
  if (tile_walls) mainloop(std::integral_constant<int, 1>{}); else mainloop(std::integral_constant<int, 0>{});
}
}  // namespace

namespace wl {
void conv_flux_enable(int on) { g_convf_on = on; }
bool conv_flux_on() { return g_convf_on != 0; }
// conv_diff!(·,u_adv) + BDIM!(NoBody, μ₀ evaluated: bd.cl_on) → u_out on the owned interior planes [ka,kb) (f is not materialised); geometry checked by conv_tile_ok
// the geometry the fused projection (PROJ) needs on top of conv_tile_ok: whole tiles, the whole single domain
bool conv_proj_ok(const GridX& g, unsigned per) {
  return g_convf_on && g.D == 3 && g.nz == g.gnz && g.k0 == 1 && g.k1 == g.nz - 1 && g.nz >= 6 && (g.nx - 2) % CF_CX == 0 && (g.ny - 2) % CF_CY == 0 && conv_tile_ok(g, per, g.k1 - g.k0);
}
int conv_flux(const float* u_adv, const GridX& g, float nu, int scheme, int ka, int kb, int zc, const void* bdp, hipStream_t s) {
  const BdimArgs bd = *(const BdimArgs*)bdp;
  const int ntiles = ((g.nx - 2 + CF_CX - 1) / CF_CX) * ((g.ny - 2 + CF_CY - 1) / CF_CY);
  const int per = (ntiles + 7) >> 3;
  const int np = kb - ka;
  const int nch = (np + zc - 1) / zc;
  const dim3 grid((unsigned)(8 * per * nch), 1, 1);
  const bool full = (g.nx - 2) % CF_CX == 0 && (g.ny - 2) % CF_CY == 0;
  const bool u0adv = bd.u0 == u_adv;
  // predictor: u⁰ is the advecting field, pre = 0, post = 1; corrector: pre = 1, post = 1/2; anything else takes the run-time form
  const int mode = (u0adv && bd.pre == 0.f && !bd.scale_after) ? 1 : ((!u0adv && bd.pre != 0.f && bd.scale_after) ? 2 : 0);
  if (bd.px && !(full && mode == 2 && g.nz == g.gnz && ka == 1 && kb == g.nz - 1)) { wl_set_error("conv_flux: the fused projection needs whole tiles, the corrector's BDIM! form and the whole single domain"); return WL_EINVAL; }
#define WL_CF(SCHV, FULLV, ADV, MD, PJ) hipLaunchKernelGGL((k_conv_flux<SCHV, FULLV, ADV, MD, PJ>), grid, dim3(CF_N), 0, s, g, u_adv, nu, ka, kb, zc, bd)
#define WL_CF1(SCHV, FULLV) do { if (mode == 1) WL_CF(SCHV, FULLV, 1, 1, 0); else if (mode == 2) WL_CF(SCHV, FULLV, 0, 2, 0); else if (u0adv) WL_CF(SCHV, FULLV, 1, 0, 0); else WL_CF(SCHV, FULLV, 0, 0, 0); } while (0)
#define WL_CF2(SCHV) do { if (bd.px) WL_CF(SCHV, 1, 0, 2, 1); else if (full) WL_CF1(SCHV, 1); else WL_CF1(SCHV, 0); } while (0)
  switch (scheme) {
    case WL_QUICK: WL_CF2(WL_QUICK); break;
    case WL_VANLEER: WL_CF2(WL_VANLEER); break;
    case WL_CDS: WL_CF2(WL_CDS); break;
    default: wl_set_error("unknown scheme"); return WL_EINVAL;
  }
#undef WL_CF2
#undef WL_CF1
#undef WL_CF
  WL_LAUNCH_CHECK(); return 0;
}
}  // namespace wl
