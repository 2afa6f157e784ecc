// Bandwidth probe: the memory mix of smoother kernel B (3 streams read, 2 written, 20 B per cell) as a plain elementwise kernel,
// with 4-, 8- and 16-byte accesses per lane and with B's tile-shaped access (256-B row segments of a 64x32 tile, z-marched).
// build: hipcc -O3 --offload-arch=gfx950 -o bw_probe bw_probe.hip ; run: ./bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); return 1; } } while (0)

template <typename V>
__global__ void __launch_bounds__(256) k_elem(const V* __restrict__ e, const V* __restrict__ r, V* __restrict__ x, V* __restrict__ ro, size_t n, float w) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    V a = e[i], b = r[i], c = x[i];
    float* pa = (float*)&a; float* pb = (float*)&b; float* pc = (float*)&c;
    for (unsigned q = 0; q < sizeof(V) / 4; q++) { pb[q] = pb[q] - w * pa[q]; pc[q] = pc[q] + w * pa[q]; }
    ro[i] = b; x[i] = c;
  }
}
// tile march: workgroup = 32x32 threads, 2 cells per thread (64x32 cells), marches nzc planes; all cells stored (no halo) — the pure access shape
__global__ void __launch_bounds__(1024) k_tile(const float* __restrict__ e, const float* __restrict__ r, float* __restrict__ x, float* __restrict__ ro, int nx, int ny, int nz, int zc, float w) {
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
int main() {
  const int nx = 512, ny = 512, nz = 512;
  const size_t n = (size_t)nx * ny * nz;
  float *e, *r, *x, *ro;
  CK(hipMalloc(&e, n * 4)); CK(hipMalloc(&r, n * 4)); CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&ro, n * 4));
  CK(hipMemset(e, 0, n * 4)); CK(hipMemset(r, 0, n * 4)); CK(hipMemset(x, 0, n * 4)); CK(hipMemset(ro, 0, n * 4));
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  auto report = [&](const char* name, float ms) { printf("%-28s %7.3f ms  %6.2f TB/s (20 B/cell)\n", name, ms, 20.0 * n / (ms * 1e-3) / 1e12); };
  for (int rep = 0; rep < 2; rep++) {
    float ms;
    for (int grid : {2048, 8192, 65536}) {
      char nm[64];
      CK(hipEventRecord(t0)); for (int q = 0; q < 5; q++) k_elem<float><<<grid, 256>>>(e, r, x, ro, n, 0.5f); CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1)); CK(hipEventElapsedTime(&ms, t0, t1));
      snprintf(nm, 64, "elem 4B/lane grid %d", grid); report(nm, ms / 5);
      CK(hipEventRecord(t0)); for (int q = 0; q < 5; q++) k_elem<float2><<<grid, 256>>>((float2*)e, (float2*)r, (float2*)x, (float2*)ro, n / 2, 0.5f); CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1)); CK(hipEventElapsedTime(&ms, t0, t1));
      snprintf(nm, 64, "elem 8B/lane grid %d", grid); report(nm, ms / 5);
      CK(hipEventRecord(t0)); for (int q = 0; q < 5; q++) k_elem<float4><<<grid, 256>>>((float4*)e, (float4*)r, (float4*)x, (float4*)ro, n / 4, 0.5f); CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1)); CK(hipEventElapsedTime(&ms, t0, t1));
      snprintf(nm, 64, "elem 16B/lane grid %d", grid); report(nm, ms / 5);
    }
    for (int zc : {16, 32, 64, 128}) {
      const int nb = (nx / 64) * (ny / 32) * ((nz + zc - 1) / zc);
      char nm[64];
      CK(hipEventRecord(t0)); for (int q = 0; q < 5; q++) k_tile<<<nb, 1024>>>(e, r, x, ro, nx, ny, nz, zc, 0.5f); CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1)); CK(hipEventElapsedTime(&ms, t0, t1));
      snprintf(nm, 64, "tile 64x32 march zc %d", zc); report(nm, ms / 5);
    }
  }
  CK(hipDeviceSynchronize());
  printf("done\n");
  return 0;
}
