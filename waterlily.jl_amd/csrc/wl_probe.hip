// Bandwidth probe for the roofline report: the memory mix of smoother kernel B (three fields read, two written: 20 B per cell) in
// kernel B's access shape (64×32-cell tiles of 256-byte row segments, marched in z) with no arithmetic to speak of.  What this
// trivial kernel reaches is the practical ceiling of that read/write mix on the device at hand (≈4.9 TB/s on MI355X, against the
// 8 TB/s specification the roofline is quoted on) — bench.py prints it beside the smoother's achieved rate.
#include "wl_common.hpp"

namespace {
__global__ void __launch_bounds__(1024) k_probe_tile(const float* __restrict__ e, const float* __restrict__ r, float* __restrict__ x, float* __restrict__ ro, int nx, int ny, int nz, int zc, float w) {
  const int ntx = nx / 64, nty = ny / 32;
  const int tile = blockIdx.x % (ntx * nty), ch = blockIdx.x / (ntx * nty);
  const int tx = tile % ntx, ty = tile / ntx;
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
  size_t o = (size_t)(tx * 64 + 2 * lx) + (size_t)(ty * 32 + ly) * nx + (size_t)ch * zc * nx * ny;
  for (int k = 0; k < zc && ch * zc + k < nz; k++, o += (size_t)nx * ny) {
    float2 a = *(const float2*)(e + o), b = *(const float2*)(r + o), c = *(const float2*)(x + o);
    b.x -= w * a.x; b.y -= w * a.y; c.x += w * a.x; c.y += w * a.y;
    *(float2*)(ro + o) = b; *(float2*)(x + o) = c;
  }
}
}  // namespace

// n: cells per side (multiple of 64); reps timed launches after one warm-up; *gbs = 20 B/cell · n³ ÷ average launch time
extern "C" int wl_probe_mix(int n, int reps, double* gbs, void* stream) {
  if (n < 64 || (n % 64) != 0 || reps < 1 || !gbs) { wl_set_error("wl_probe_mix: n must be a multiple of 64, reps >= 1"); return WL_EINVAL; }
  hipStream_t s = wl_stream(stream);
  const size_t cells = (size_t)n * n * n;
  float* buf = nullptr;
  WL_HIP(hipMalloc(&buf, 4 * cells * sizeof(float)));
  hipError_t e = hipMemsetAsync(buf, 0, 4 * cells * sizeof(float), s);
  hipEvent_t t0 = nullptr, t1 = nullptr;
  if (e == hipSuccess) e = hipEventCreate(&t0);
  if (e == hipSuccess) e = hipEventCreate(&t1);
  float ms = 0.f;
  if (e == hipSuccess) {
    const int zc = 32, nb = (n / 64) * (n / 32) * ((n + zc - 1) / zc);
    float *a = buf, *b = buf + cells, *c = buf + 2 * cells, *d = buf + 3 * cells;
    hipLaunchKernelGGL(k_probe_tile, dim3(nb), dim3(1024), 0, s, a, b, c, d, n, n, n, zc, 0.5f);
    (void)hipEventRecord(t0, s);
    for (int q = 0; q < reps; q++) hipLaunchKernelGGL(k_probe_tile, dim3(nb), dim3(1024), 0, s, a, b, c, d, n, n, n, zc, 0.5f);
    (void)hipEventRecord(t1, s);
    e = hipEventSynchronize(t1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, t0, t1);
    if (e == hipSuccess) e = hipGetLastError();
  }
  if (t0) (void)hipEventDestroy(t0);
  if (t1) (void)hipEventDestroy(t1);
  (void)hipFree(buf);
  WL_HIP(e);
  *gbs = 20.0 * (double)cells / ((double)ms / reps * 1e-3) / 1e9;
  return 0;
}
