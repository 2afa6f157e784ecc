/* wlhip_bench.h — MEASUREMENT interface of libwlhip.so (bench.py, tools/): HIP-event pairs on the launch stream, launch and
 * path counters.  NOT part of the drop-in boundary: nothing in the reference binds these (the reference-facing surface is wlhip.h);
 * they exist because the harness contract asks for kernel durations measured live inside the timed region, on the stream the
 * kernels are launched on. */
#ifndef WLHIP_BENCH_H
#define WLHIP_BENCH_H
#include "wlhip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- measurement hooks (bench.py): HIP-event pairs recorded on the launch stream around named launches ----
 * slots: 0 fine-level GS colour sweep (one launch), 1 fine-level smooth! (GaussSeidelRB! as a whole),
 *        2 fine-level Jacobi!, 3 conv_diff!, 4 fine-level residual!+norms, 5 BDIM!, 6 fine-level prolongate+increment,
 *        7 coarse levels (everything below level 1 of a V-cycle), 8 mom_step! as a whole,
 *        9 / 10 fine-level kernels A / B of the temporally blocked smoother (wl_fused.hip)                        */
enum { WL_PROF_GS_SWEEP = 0, WL_PROF_SMOOTH = 1, WL_PROF_JACOBI = 2, WL_PROF_CONVDIFF = 3, WL_PROF_RESIDUAL = 4,
       WL_PROF_BDIM = 5, WL_PROF_PROLONG = 6, WL_PROF_COARSE = 7, WL_PROF_STEP = 8, WL_PROF_GS_A = 9, WL_PROF_GS_B = 10, WL_PROF_NSLOTS = 11 };
int wl_prof_enable(int on);                                     /* 0 off, 1 all slots, 2 only slots 9 and 10 (each event pair costs a few µs of
                                                                    stream time); also resets all slots */
int wl_prof_read(int slot, int* host_count, double* host_total_ms);   /* synchronises the device */

/* kernel launches issued by the library in this process so far (every hipLaunchKernelGGL of libwlhip; copies and memsets are not counted) */
long wl_launch_count(void);
/* With WL_PLACEMENT_TRIALS=2…8 (default 1 = off) wl_sim_create on a large 3-D grid whose arrays the handle owns tries that many placements of its
 * allocations and keeps the fastest (wl_sim.hip "Placement trials" — measured: not a reliable remedy): the candidates' scores, ms of a timed mom_project! pair */
int wl_placement_scores(double* out, int cap);
/* per-handle path counters: "resjac" = solves whose fused projection head (div + x·dt + residual! + first Jacobi!) stood,
 * "resjac_redo" = solves where residual!'s mean shift was due after all and the head was redone on the two-kernel path,
 * "resjac_backoff" = 1 once three consecutive redos switched the fused head off for this handle (re-armed by wl_sim_update),
 * "bcdefer" = BC!(u,U) applications after conv_diff!+BDIM! that were left to the following projection (two per step when the option is live),
 * "tailspec" = projection tails that ran from inside the solver loop, ahead of the convergence read,
 * "tailfuse" = projections whose velocity update (u −= L∇x, BC!) was evaluated by the corrector's conv_diff! loader instead of a tail launch,
 * "xdefer" = what the finest level's last smooth! decided: 1 the V-cycle's x += ω·x_c↓ was applied by smoother kernel B, 0 by kernel A, −1 none yet
 * (decides which bytes bench.py books to kernels A and B) */
int wl_sim_counter(wl_sim* s, const char* name, long* out);

#ifdef __cplusplus
}
#endif
#endif /* WLHIP_BENCH_H */
