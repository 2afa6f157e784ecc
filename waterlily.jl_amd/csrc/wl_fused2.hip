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

int g_pair_on = 1;
int g_pro_fast = 1;   // bit 1 of gsrb_pair_enable: register-window prolongation when every direction is coarsened

// The kernels exist for two tile heights: 64×32 cells (32×32 threads; 71 % of the thread-work is tile core) for grids that fill the
// chip many times over, and 64×16 cells (32×16 threads; 57 % core) for the smaller levels, where twice the number of tiles
// allows z-chunks twice as long (less pipeline warm-up) and fuller rounds of workgroups.
#define WL_PT_Y 32
#define WL_PNS pair32
#include "wl_fused2_body.inc"
#undef WL_PT_Y
#undef WL_PNS
#define WL_PT_Y 16
#define WL_PNS pair16
#include "wl_fused2_body.inc"
#undef WL_PT_Y
#undef WL_PNS

static int pair_min_nx() { static const int v = wl_exp_int("WL_PAIR_MIN_NX", 34); return v; }   // smallest level width of the pair kernels (34: 256³ 1.81 -> 1.78 ms/step against the one-cell kernels on the 34-wide level; tools/minnx_gate.sh)
namespace wl {
void gsrb_pair_enable(int on) { g_pair_on = on & 1; g_pro_fast = (on & 2) == 0; }
// geometry: even nx (float2), tiles not mostly empty, 32-bit offsets; a z-slab needs 3 ghost planes per side (kernel B's halo)
bool gsrb_pair_geom_ok(const GridX& g) {
  return g_pair_on && g.D == 3 && (g.nx & 1) == 0 && g.nx >= pair_min_nx() && g.ny >= 34 && g.gnz >= 10 && (g.k1 - g.k0) >= 8 && (g.nz == g.gnz || g.k0 >= 2) && g.cs < (1L << 30);   // (z-slab: kernel A reads 2 planes below its first output plane, B 3 — wl_mg::pair_slab checks the level's ghost depth)
}
bool gsrb_pair_ok(const GridX& g, const ConstL& cl) { return cl.on && gsrb_pair_geom_ok(g); }
// a plane sub-range [g.k0,g.k1) of a level that qualifies as a whole: any number of planes (boundary slices of a slab, ranges of the z-split)
bool gsrb_pair_ok_range(const GridX& g, const ConstL& cl) { GridX h = g; if (h.k1 - h.k0 < 8) h.k1 = h.k0 + 8; return g.k1 > g.k0 && gsrb_pair_ok(h, cl); }
// 16-row tiles where the 32-row tiling cannot fill the chip for many rounds (WL_PAIR_ROWS=16|32 forces one: experiments)
static bool rows16(const GridX& g, int kernel = 0) {   // kernel: 1 = A, 2 = B (experiments: WL_PAIR_ROWS_A / WL_PAIR_ROWS_B force one kernel's tile height)
  static const int force_all = wl_exp_int("WL_PAIR_ROWS", 0);
  static const int force_a = wl_exp_int("WL_PAIR_ROWS_A", 0);
  static const int force_b = wl_exp_int("WL_PAIR_ROWS_B", 0);
  const int force = (kernel == 1 && force_a) ? force_a : ((kernel == 2 && force_b) ? force_b : force_all);
  if (force == 16) return true;
  if (force == 32) return false;
  // measured with the 128-VGPR kernels (tools/rows_gate.sh, tools/rows512.sh): 16-row tiles win on every level below the 512²-plane class
  // (128³: 0.582 -> 0.555 ms/step, 256³: 1.845 -> 1.811), 32-row tiles on 512² planes (A 0.96 vs 1.01, B 1.41 vs 1.53 ms/step)
  const long tiles32 = (long)((g.nx + 55) / 56) * ((g.ny + 25) / 26);
  // (the in-plane size decides: a 512² plane has 200 tiles of 64×32 cells — enough parallelism per plane layer, also on a z-slab of few planes)
  return tiles32 < 128;
}
int gsrb_pair_A(float* emid, const float* r, const GridX& g, const ConstL& cl, hipStream_t s) {
  return rows16(g, 1) ? pair16::gsrb_pair_A(emid, r, g, cl, s) : pair32::gsrb_pair_A(emid, r, g, cl, s);
}
int gsrb_pair_A_pro(float* emid, float* rnew, float* x, const float* r, const float* xc, const GridX& g, const GridX& gc, float w, const ConstL& cl, hipStream_t s, int xk0, int xk1) {
  return rows16(g, 1) ? pair16::gsrb_pair_A_pro(emid, rnew, x, r, xc, g, gc, w, cl, s, xk0, xk1) : pair32::gsrb_pair_A_pro(emid, rnew, x, r, xc, g, gc, w, cl, s, xk0, xk1);
}
int gsrb_pair_B(float* eps, float* rout, float* x, const float* emid, const float* r, const GridX& g, float w,
                const RedWs* ws, int slot_d, int slot_f, const ConstL& cl, hipStream_t s, const XDefer* xd) {
  if (xd && xd->gc.cs >= (1L << 30)) { wl_set_error("gsrb_pair_B: coarse level too large for 32-bit offsets"); return WL_EINVAL; }
  return rows16(g, 2) ? pair16::gsrb_pair_B(eps, rout, x, emid, r, g, w, ws, slot_d, slot_f, cl, s, xd) : pair32::gsrb_pair_B(eps, rout, x, emid, r, g, w, ws, slot_d, slot_f, cl, s, xd);
}
}  // namespace wl

#ifdef WL_STAMP
// diagnostic build: per-wave cycle sums of kernel A's fast steps (pair32 instance) — [pre-barrier, barrier wait, sweeps, stores, steps]
extern "C" int wl_debug_stamps(unsigned long long* out5, int reset) {
  unsigned long long z[8] = {0};
  if (hipMemcpyFromSymbol(out5, HIP_SYMBOL(pair32::g_wl_stamp), 5 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(pair32::g_wl_stamp), z, sizeof(z)) != hipSuccess) return -1;
  return 0;
}
#endif
