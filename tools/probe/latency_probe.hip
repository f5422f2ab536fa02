// Single-wave latency probes on gfx950: what does one dependent instruction cost a lone wave?
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probe/latency_probe.hip -o gpurun_out/latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
__global__ void probe(double* out, unsigned long long* cyc, double a, double b, int reps) {
  __shared__ double lds[256];
  const int lane = threadIdx.x;
  lds[lane] = a + lane; lds[lane + 64] = b;
  __syncthreads();
  double x = a + lane * 1e-9, y = b;
  unsigned long long t0, t1;
  // 1. dependent v_add_f64 chain
  t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
  for (int i = 0; i < N; ++i) x = x + y;
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[0] = t1 - t0;
  // 2. dependent v_mul_f64 / v_add_f64 alternating
  t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
  for (int i = 0; i < N; ++i) { x = x * y; x = x + y; }
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[1] = t1 - t0;
  // 3. two independent chains interleaved (ILP 2)
  double z = a * 0.5 + lane;
  t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
  for (int i = 0; i < N; ++i) { x = x + y; z = z + y; }
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[2] = t1 - t0;
  // 4. dependent LDS round trip (store, load, dependent address)
  int idx = lane;
  t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 4
  for (int i = 0; i < N / 8; ++i) { lds[idx & 63] = x; x = lds[(idx + 1) & 63] + y; idx = (int)x & 63; }
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[3] = t1 - t0;
  // 5. taken uniform branches (loop back edge every 2 instructions, not unrolled)
  t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < N; ++i) { x = x + y; asm volatile("" ::: "memory"); }
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[4] = t1 - t0;
  // 6. dependent fp32 chain for comparison
  float f = (float)a + lane, g = (float)b;
  t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
  for (int i = 0; i < N; ++i) f = f + g;
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[5] = t1 - t0;
  // 7. divergent skip: a block executed by no lane (s_cbranch_execz taken) between dependent adds
  t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < N; ++i) {
    x = x + y;
    if (x == -12345.0) { x = x * 3.0 + lds[(i + lane) & 127]; x = x * x - y; x = x / (y + 3.0); x = sqrt(x * x + 1.0); x = x * 1.5 + lds[(i * 3 + lane) & 127]; x = x / (y + 5.0); }
  }
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[6] = t1 - t0;
  out[lane] = x + z + f;
  (void)reps;
}
int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 64 * sizeof(double)); hipMalloc(&cyc, 8 * sizeof(unsigned long long));
  for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, cyc, 1.0, 1e-7, 1); hipDeviceSynchronize(); }
  unsigned long long h[8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  // s_memtime counts at a constant 100 MHz on this part; convert with the measured shader clock of pass 6 if needed
  printf("s_memtime ticks for %d iterations (one lone wave):\n", N);
  printf("  dependent v_add_f64              : %llu  (%.3f ticks/op)\n", h[0], (double)h[0] / N);
  printf("  dependent v_mul_f64 + v_add_f64  : %llu  (%.3f ticks/op)\n", h[1], (double)h[1] / (2.0 * N));
  printf("  two independent v_add_f64 chains : %llu  (%.3f ticks/op)\n", h[2], (double)h[2] / (2.0 * N));
  printf("  dependent LDS store+load         : %llu  (%.3f ticks/round trip)\n", h[3], (double)h[3] / (N / 8));
  printf("  v_add_f64 + taken loop branch    : %llu  (%.3f ticks/iteration)\n", h[4], (double)h[4] / N);
  printf("  dependent v_add_f32              : %llu  (%.3f ticks/op)\n", h[5], (double)h[5] / N);
  printf("  v_add_f64 + skipped block + loop : %llu  (%.3f ticks/iteration)\n", h[6], (double)h[6] / N);
  return 0;
}
