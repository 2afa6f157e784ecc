// Microbenchmark: how much does access width matter on MI355X for our access patterns?
//   stream:  y = a*x + z                 (dword / float2 / float4 per lane)
//   stencil: 7-point on a 514^3 padded array, 1 / 2 / 4 x-cells per thread
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench.hip -o tools/microbench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int W> struct Vec;
template <> struct Vec<1> { typedef float T; };
template <> struct Vec<2> { typedef float2 T; };
template <> struct Vec<4> { typedef float4 T; };

template <int W>
__global__ void k_triad(float* __restrict__ y, const float* __restrict__ x, const float* __restrict__ z, float a, long n) {
  typedef typename Vec<W>::T V;
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x);
  if (i * W >= n) return;
  V xv = ((const V*)x)[i], zv = ((const V*)z)[i], o;
  float* xo = (float*)&xv; float* zo = (float*)&zv; float* oo = (float*)&o;
#pragma unroll
  for (int q = 0; q < W; q++) oo[q] = a * xo[q] + zo[q];
  ((V*)y)[i] = o;
}

// XCD-aware tile map as in the library (strip per XCD)
__device__ __forceinline__ void tile(long sz_threads, long& m, int& p) {
  const unsigned h = blockIdx.x, q = h & 7u, s = h >> 3;
  const long nbx = (sz_threads + 255) / 256; const unsigned per = (unsigned)((nbx + 7) >> 3);
  p = (int)(s / per); const long bx = (long)q * per + (s - (unsigned)p * per);
  m = bx * 256 + threadIdx.x;
}
// r = z - A x with variable coefficients (like residual!): W cells per thread along x
template <int W>
__global__ void k_stencil(float* __restrict__ r, const float* __restrict__ x, const float* __restrict__ z, const float* __restrict__ L, const float* __restrict__ Dg, int nx, int ny, int nz) {
  typedef typename Vec<W>::T V;
  const long sy = nx, sz = (long)nx * ny, cs = sz * nz;
  long m; int p; tile(sz / W, m, p);
  if (m * W >= sz) return;
  const int k = 1 + p;
  const long o = m * W + (long)k * sz;
  const int i0 = (int)((m * W) % nx), j = (int)((m * W) / nx);
  if (j < 1 || j > ny - 2) return;
  float xc[W + 2];
  V c = *(const V*)(x + o); for (int q = 0; q < W; q++) xc[q + 1] = ((float*)&c)[q];
  xc[0] = x[o - 1]; xc[W + 1] = x[o + W];
  float ym[W], yp[W], zm[W], zp[W], lx[W + 1], ly[W], lyp[W], lz[W], lzp[W], dg[W], zz[W];
  if (W == 4) {  // rows are only 8-byte aligned: split the y-neighbour loads in two float2
    float2 a0 = *(const float2*)(x + o - sy), a1 = *(const float2*)(x + o - sy + 2); ym[0] = a0.x; ym[1] = a0.y; ym[2 % W] = a1.x; ym[3 % W] = a1.y;
    float2 b0 = *(const float2*)(x + o + sy), b1 = *(const float2*)(x + o + sy + 2); yp[0] = b0.x; yp[1] = b0.y; yp[2 % W] = b1.x; yp[3 % W] = b1.y;
    float2 c0 = *(const float2*)(L + cs + o + sy), c1 = *(const float2*)(L + cs + o + sy + 2); lyp[0] = c0.x; lyp[1] = c0.y; lyp[2 % W] = c1.x; lyp[3 % W] = c1.y;
  } else {
    V a = *(const V*)(x + o - sy), b = *(const V*)(x + o + sy), cc = *(const V*)(L + cs + o + sy);
    for (int q = 0; q < W; q++) { ym[q] = ((float*)&a)[q]; yp[q] = ((float*)&b)[q]; lyp[q] = ((float*)&cc)[q]; }
  }
  V v;
  v = *(const V*)(x + o - sz); for (int q = 0; q < W; q++) zm[q] = ((float*)&v)[q];
  v = *(const V*)(x + o + sz); for (int q = 0; q < W; q++) zp[q] = ((float*)&v)[q];
  v = *(const V*)(L + o); for (int q = 0; q < W; q++) lx[q] = ((float*)&v)[q];
  lx[W] = L[o + W];
  v = *(const V*)(L + cs + o); for (int q = 0; q < W; q++) ly[q] = ((float*)&v)[q];
  v = *(const V*)(L + 2 * cs + o); for (int q = 0; q < W; q++) lz[q] = ((float*)&v)[q];
  v = *(const V*)(L + 2 * cs + o + sz); for (int q = 0; q < W; q++) lzp[q] = ((float*)&v)[q];
  v = *(const V*)(Dg + o); for (int q = 0; q < W; q++) dg[q] = ((float*)&v)[q];
  v = *(const V*)(z + o); for (int q = 0; q < W; q++) zz[q] = ((float*)&v)[q];
  V out;
#pragma unroll
  for (int q = 0; q < W; q++) {
    float s = xc[q + 1] * dg[q];
    s += xc[q] * lx[q] + xc[q + 2] * lx[q + 1];
    s += ym[q] * ly[q] + yp[q] * lyp[q];
    s += zm[q] * lz[q] + zp[q] * lzp[q];
    const int i = i0 + q;
    ((float*)&out)[q] = (i >= 1 && i <= nx - 2) ? zz[q] - s : 0.f;
  }
  *(V*)(r + o) = out;
}

int main() {
  const int nx = 514, ny = 514, nz = 514;
  const long sz = (long)nx * ny, n = sz * nz;
  float *x, *z, *r, *L, *D;
  CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&z, n * 4)); CK(hipMalloc(&r, n * 4)); CK(hipMalloc(&L, 3 * n * 4 + 4096)); CK(hipMalloc(&D, n * 4));
  CK(hipMemset(x, 0, n * 4)); CK(hipMemset(z, 0, n * 4)); CK(hipMemset(L, 0, 3 * n * 4)); CK(hipMemset(D, 0, n * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](auto&& launch, const char* name, double bytes) {
    for (int w = 0; w < 3; w++) launch();
    hipEventRecord(e0, 0);
    const int R = 20;
    for (int w = 0; w < R; w++) launch();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= R;
    printf("%-34s %8.3f ms  %8.1f GB/s (compulsory)\n", name, ms, bytes / ms / 1e6);
    return 0;
  };
  timeit([&] { hipLaunchKernelGGL(k_triad<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, r, x, z, 2.f, n); }, "triad dword/lane", 12.0 * n);
  timeit([&] { hipLaunchKernelGGL(k_triad<2>, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, 0, r, x, z, 2.f, n); }, "triad float2/lane", 12.0 * n);
  timeit([&] { hipLaunchKernelGGL(k_triad<4>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, 0, r, x, z, 2.f, n); }, "triad float4/lane", 12.0 * n);
  const double sb = 32.0 * (double)(nx - 2) * (ny - 2) * (nz - 2);   // x,z,L*3,D read + r written
  auto grid = [&](int W) { long nbx = (sz / W + 255) / 256; long per = (nbx + 7) >> 3; return dim3((unsigned)(8 * per * (nz - 2))); };
  timeit([&] { hipLaunchKernelGGL(k_stencil<1>, grid(1), dim3(256), 0, 0, r, x, z, L, D, nx, ny, nz); }, "residual-like, 1 cell/thread", sb);
  timeit([&] { hipLaunchKernelGGL(k_stencil<2>, grid(2), dim3(256), 0, 0, r, x, z, L, D, nx, ny, nz); }, "residual-like, 2 cells/thread", sb);
  timeit([&] { hipLaunchKernelGGL(k_stencil<4>, grid(4), dim3(256), 0, 0, r, x, z, L, D, nx, ny, nz); }, "residual-like, 4 cells/thread", sb);
  CK(hipDeviceSynchronize());
  return 0;
}
