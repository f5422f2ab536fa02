// How long does a wave64 VALU instruction occupy a lone gfx950 wave as a function of WHICH lanes are active?
// (Does the SIMD skip the 16-lane passes whose EXEC bits are all zero?)  And what do the cross-lane primitives cost?
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probe/exec_probe.hip -o gpurun_out/exec_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
template <int KIND>
__device__ unsigned long long run(double& x, double y, float& f, int& k, unsigned long long mask, int lane) {
  unsigned long long t0 = 0, t1 = 0;
  double a = x, b = x * 0.5, c = x * 0.25, d = x * 0.125;
  float fa = f, fb = f * 0.5f, fc = f * 0.25f, fd = f * 0.125f;
  int ka = k, kb = k + 1, kc = k + 2, kd = k + 3;
  if ((mask >> lane) & 1ull) {
    t0 = __builtin_amdgcn_s_memtime();
    if (KIND == 0) {  // dependent f64 add
#pragma unroll 16
      for (int i = 0; i < N; ++i) a = a + y;
    } else if (KIND == 1) {  // 4 independent f64 chains
#pragma unroll 4
      for (int i = 0; i < N / 4; ++i) { a = a + y; b = b + y; c = c + y; d = d + y; }
    } else if (KIND == 2) {  // 4 independent f32 chains
#pragma unroll 4
      for (int i = 0; i < N / 4; ++i) { fa = fa + (float)y; fb = fb + (float)y; fc = fc + (float)y; fd = fd + (float)y; }
    } else if (KIND == 3) {  // 4 independent int chains
#pragma unroll 4
      for (int i = 0; i < N / 4; ++i) { ka = ka * 3 + 1; kb = kb * 5 + 1; kc = kc * 7 + 1; kd = kd * 9 + 1; }
    } else if (KIND == 4) {  // 4 independent f64 mul chains
#pragma unroll 4
      for (int i = 0; i < N / 4; ++i) { a = a * y; b = b * y; c = c * y; d = d * y; }
    }
    t1 = __builtin_amdgcn_s_memtime();
  }
  x = a + b + c + d; f = fa + fb + fc + fd; k = ka + kb + kc + kd;
  return t1 - t0;
}
__global__ void probe(double* out, unsigned long long* cyc, double y0) {
  const int lane = threadIdx.x;
  const unsigned long long masks[8] = {~0ull, 0xFFFFFFFFull, 0xFFFFull, 0x7FFull, 0x7FF000007FFull, 0x03FF03FF03FF03FFull, 0x1ull, 0xFFFF0000ull};
  double x = 1.0 + lane * 1e-9, y = y0; float f = 1.0f + lane; int k = lane;
  for (int m = 0; m < 8; ++m) {
    unsigned long long c0 = run<0>(x, y, f, k, masks[m], lane), c1 = run<1>(x, y, f, k, masks[m], lane), c2 = run<2>(x, y, f, k, masks[m], lane),
                       c3 = run<3>(x, y, f, k, masks[m], lane), c4 = run<4>(x, y, f, k, masks[m], lane);
    const int first = __builtin_ctzll(masks[m]);
    if (lane == first) { cyc[m * 8 + 0] = c0; cyc[m * 8 + 1] = c1; cyc[m * 8 + 2] = c2; cyc[m * 8 + 3] = c3; cyc[m * 8 + 4] = c4; }
  }
  // cross-lane primitives, all lanes active, dependent chains
  unsigned long long t0, t1;
  int v = lane;
  t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 8
  for (int i = 0; i < N / 8; ++i) v = __shfl(v, (v + 1) & 63, 64);  // ds_bpermute_b32, dependent
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[64] = t1 - t0;
  t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 8
  for (int i = 0; i < N / 8; ++i) v = __builtin_amdgcn_readlane(v, i & 63) + lane;  // v_readlane + dependent VALU
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[65] = t1 - t0;
  t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 8
  for (int i = 0; i < N / 8; ++i) v = __builtin_amdgcn_update_dpp(0, v, 0x111 /* row_shr:1 */, 0xF, 0xF, false) + 1;
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[66] = t1 - t0;
  // 8 independent bpermutes in flight then one wait
  int w0 = lane, w1 = lane + 1, w2 = lane + 2, w3 = lane + 3;
  t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 2
  for (int i = 0; i < N / 32; ++i) {
    int a0 = __shfl(w0, (lane + 1) & 63, 64), a1 = __shfl(w1, (lane + 2) & 63, 64), a2 = __shfl(w2, (lane + 3) & 63, 64), a3 = __shfl(w3, (lane + 4) & 63, 64);
    w0 = a1 + 1; w1 = a2 + 1; w2 = a3 + 1; w3 = a0 + 1;
  }
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[67] = t1 - t0;
  out[lane] = x + f + k + v + w0 + w1 + w2 + w3;
}
int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 64 * sizeof(double)); hipMalloc(&cyc, 80 * sizeof(unsigned long long));
  hipMemset(cyc, 0, 80 * sizeof(unsigned long long));
  for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, cyc, 1e-7); hipDeviceSynchronize(); }
  unsigned long long h[80];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[8] = {"all 64 lanes", "lanes 0-31", "lanes 0-15", "lanes 0-10", "lanes 0-10 + 32-42", "10 lanes in every quarter", "lane 0", "lanes 16-31"};
  printf("shader cycles per instruction of one lone wave (N = %d instructions per measurement)\n", N);
  printf("%-28s %10s %10s %10s %10s %10s\n", "active lanes", "f64 dep", "f64 x4", "f32 x4", "int x4", "f64mul x4");
  for (int m = 0; m < 8; ++m)
    printf("%-28s %10.2f %10.2f %10.2f %10.2f %10.2f\n", names[m], (double)h[m * 8] / N, (double)h[m * 8 + 1] / N, (double)h[m * 8 + 2] / N,
           (double)h[m * 8 + 3] / (2.0 * N), (double)h[m * 8 + 4] / N);
  printf("ds_bpermute_b32 dependent    : %.1f cycles each\n", (double)h[64] / (N / 8));
  printf("v_readlane + dependent add   : %.1f cycles each\n", (double)h[65] / (N / 8));
  printf("DPP row_shr mov + add        : %.1f cycles each\n", (double)h[66] / (N / 8));
  printf("4 independent ds_bpermute    : %.1f cycles per group of 4\n", (double)h[67] / (N / 32));
  return 0;
}
