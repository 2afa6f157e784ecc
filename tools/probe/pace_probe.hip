// Can workgroups of one kernel pace each other through a global counter on a multi-XCD gfx950?  512 co-resident workgroups of 512 threads each run
// STEPS rounds: announce (atomic add), then poll (a) with an atomic RMW of 0, (b) with an agent-scope atomic load, until all have announced or a spin limit.
// Reports per mode: rounds that timed out, mean / max spins, kernel time.   build: hipcc -O3 --offload-arch=gfx950 tools/probe/pace_probe.hip -o tools/probe/pace_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define STEPS 64
template <int MODE> __global__ void __launch_bounds__(512) k(int* cnt, int nwg, int* spins_out, int* timeouts) {
  __shared__ float lds[18000];     // 72 KB: two workgroups per CU, like the conv kernel
  lds[threadIdx.x] = threadIdx.x;
  int tot = 0, to = 0;
  for (int r = 0; r < STEPS; r++) {
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(cnt + r, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int s = 0;
      while (s < 2000) {
        const int v = MODE == 0 ? __hip_atomic_fetch_add(cnt + r, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : __hip_atomic_load(cnt + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v >= nwg) break;
        __builtin_amdgcn_s_sleep(8); s++;
      }
      tot += s; if (s >= 2000) to++;
    }
    __syncthreads();
    // some work: ~2 us
    float a = lds[threadIdx.x];
    for (int i = 0; i < 400; i++) a = a * 1.0001f + 0.5f;
    lds[threadIdx.x] = a;
    __syncthreads();
  }
  if (threadIdx.x == 0) { spins_out[blockIdx.x] = tot; atomicAdd(timeouts, to); }
}
int main() {
  int *cnt, *sp, *to; hipMalloc(&cnt, STEPS * 4); hipMalloc(&sp, 4096 * 4); hipMalloc(&to, 4);
  for (int nwg : {256, 512}) for (int mode = 0; mode < 2; mode++) {
    hipMemset(cnt, 0, STEPS * 4); hipMemset(to, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    if (mode == 0) k<0><<<nwg, 512>>>(cnt, nwg, sp, to); else k<1><<<nwg, 512>>>(cnt, nwg, sp, to);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static int h[4096]; int hto; hipMemcpy(h, sp, nwg * 4, hipMemcpyDeviceToHost); hipMemcpy(&hto, to, 4, hipMemcpyDeviceToHost);
    long s = 0; int mx = 0; for (int i = 0; i < nwg; i++) { s += h[i]; if (h[i] > mx) mx = h[i]; }
    printf("workgroups %d  poll=%s  timed-out rounds %d of %d  spins per workgroup: mean %.1f max %d (over %d rounds)  kernel %.3f ms\n", nwg, mode == 0 ? "atomic add 0" : "atomic load ", hto, nwg * STEPS, (double)s / nwg, mx, STEPS, ms);
  }
  return 0;
}
