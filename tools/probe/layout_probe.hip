// The library's own z-marching kernels on two memory layouts of the same 512^3 problem (round 3, profiles/r03_experiments.md §15):
//   dense  — the reference's layout: rows of 514 floats, first interior cell at byte 4 of its row (what every caller-owned array has);
//   padded — rows of 544 floats (a multiple of 32), plane stride a multiple of 32, base shifted so that cell x = 1 starts a 128-byte line.
// The kernels index through GridX's strides, so the same launchers run on both (whole-array helpers that assume sy == nx are not used here).
// Timed per launch (HIP events, 10 repetitions after 3 warm-ups), same buffers for both layouts (one placement state), random data:
//   conv_diff!+BDIM! predictor and corrector (wl_convf.hip), the fused projection head (wl_resjac.hip), smoother kernels A and B (wl_fused2.hip).
// build (after make in waterlily.jl_amd/csrc):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I waterlily.jl_amd/csrc -I include tools/probe/layout_probe.hip -L waterlily.jl_amd -lwlhip -Wl,-rpath,'$ORIGIN/../../waterlily.jl_amd' -o tools/probe/layout_probe
#include <cstdio>
#include <vector>
#include "wl_common.hpp"
#include "wl_conv_cell.hpp"
extern "C" int wl_init(int);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define WT(x) do { int r_ = (x); if (r_) { printf("library error %d at line %d: %s\n", r_, __LINE__, wl_last_error()); return 1; } } while (0)
static const char* wl_last_error() { return "(see wl_last_error_string)"; }
int main(int argc, char** argv) {
  const bool only_head = argc > 1 && argv[1][0] == 'h';     // 'h': the fused head only (64-cell-core experiment: link against libwlhip_rj34.so, WL_RJ_X34=1)
  if (wl_init(0)) { printf("wl_init failed\n"); return 1; }
  const int N = 512, NG = N + 2, PITCH = 544;
  const size_t maxcs = (size_t)PITCH * NG * NG + 4096;
  float *u, *u0, *uo, *p, *x2, *r, *r2, *em, *eps; void* wsb;
  CK(hipMalloc(&u, 3 * maxcs * 4)); CK(hipMalloc(&u0, 3 * maxcs * 4)); CK(hipMalloc(&uo, 3 * maxcs * 4));
  for (float** q : {&p, &x2, &r, &r2, &em, &eps}) CK(hipMalloc(q, maxcs * 4));
  CK(hipMalloc(&wsb, wl_red_bytes()));
  std::vector<float> h(3 * maxcs);
  unsigned s = 777u; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.2f; }
  for (float* q : {u, u0, uo}) CK(hipMemcpy(q, h.data(), 3 * maxcs * 4, hipMemcpyHostToDevice));
  for (float* q : {p, x2, r, r2, em, eps}) CK(hipMemcpy(q, h.data(), maxcs * 4, hipMemcpyHostToDevice));
  const RedWs ws = wl_red_ws(wsb);
  wl::ConstL cl; cl.on = 1; cl.c[0] = cl.c[1] = cl.c[2] = 1.f;
  for (int nz = 0; nz < 3; nz++) for (int ny = 0; ny < 3; ny++) for (int nx = 0; nx < 3; nx++) {
    const float c = 1.f; const float px = nx == 2 ? c + c : (nx == 1 ? c + 0.f : 0.f), py = ny == 2 ? c + c : (ny == 1 ? c + 0.f : 0.f), pz = nz == 2 ? c + c : (nz == 1 ? c + 0.f : 0.f);
    float d = 0.f; d -= px; d -= py; d -= pz; cl.Dt[nx + 3 * ny + 9 * nz] = d; cl.iDt[nx + 3 * ny + 9 * nz] = (d == 0.f) ? d : 1.0f / d;
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto&& launch) -> int {
    for (int w = 0; w < 3; w++) if (launch()) { printf("%s: launch failed\n", name); return 1; }
    hipEventRecord(e0, 0);
    const int R = 10;
    for (int w = 0; w < R; w++) launch();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("   %-46s %7.3f ms per launch\n", name, ms / R); fflush(stdout);
    return 0;
  };
  for (int rep = 0; rep < 2; rep++) for (int lay = 0; lay < 3; lay++) {
    GridX g; g.D = 3; g.nx = NG; g.ny = NG; g.nz = NG; g.k0 = 1; g.k1 = NG - 1; g.gk = 0; g.gnz = NG;
    g.sy = lay ? PITCH : NG; g.sz = g.sy * NG; g.cs = g.sz * NG;
    const size_t off = lay == 1 ? 31 : 0;      // padded (1): element 0 of a row sits 4 bytes before a 128-byte boundary, cell x = 1 on it (hipMalloc returns 256-byte aligned memory)
    // padded (2): rows start ON a line boundary (cell x = 0): the pair kernels' even-x pairs stay 8-byte aligned
    printf("== %s layout: row pitch %ld floats, plane %ld floats, first interior cell at byte %zu (mod 128) of a line\n", lay == 0 ? "dense " : (lay == 1 ? "padded, x=1 on a line" : "padded, x=0 on a line"), g.sy, g.sz, (size_t)((off + 1) * 4 % 128));
    float *U = u + off, *U0 = u0 + off, *UO = uo + off, *P = p + off, *X2 = x2 + off, *R = r + off, *R2 = r2 + off, *EM = em + off, *EPS = eps + off;
    BdimArgs bp{U, nullptr, UO, 0.1f, 0.f, 1.f, 0, 1, {1.f, 1.f, 1.f}};                 // predictor: u⁰ is the advecting field, pre = 0, post = 1
    BdimArgs bc{U0, nullptr, UO, 0.1f, 1.f, 0.5f, 1, 1, {1.f, 1.f, 1.f}};               // corrector
    if (!only_head) {
    if (timeit("conv_diff!+BDIM! predictor (k_conv_flux)", [&] { return wl::conv_tile(U, g, 0.01f, WL_QUICK, g.k0, g.k1, &bp, 0); })) return 1;
    if (timeit("conv_diff!+BDIM! corrector (k_conv_flux)", [&] { return wl::conv_tile(U, g, 0.01f, WL_QUICK, g.k0, g.k1, &bc, 0); })) return 1;
    }
    if (timeit("fused projection head (k_resjac)", [&] { return wl::resjac(X2, R, P, U, g, 0.3f, 1.f, cl, ws, 0, 0, 0, false); })) return 1;
    if (only_head) continue;
    if (timeit("smoother kernel A (k_gsrb2_A)", [&] { return wl::gsrb_fused_A(EM, R, nullptr, g, cl, 0); })) return 1;
    if (timeit("smoother kernel B (k_gsrb2_B)", [&] { return wl::gsrb_fused_B(EPS, R2, X2, EM, R, nullptr, g, 1.f, nullptr, 0, 0, cl, 0); })) return 1;
  }
  CK(hipDeviceSynchronize());
  printf("done\n");
  return 0;
}
