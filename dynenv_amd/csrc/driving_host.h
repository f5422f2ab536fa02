// driving_host.h - what the host code (dynenv_capi.hip) needs from the Driving translation unit (driving_tu.hip).
// The Driving kernels are a translation unit of their own since round 5: they are 1.4-2 % faster compiled with -Os (-O2: 1.3-1.5 %) than with -O3
// (whole-episode mean at 4096 envs, three interleaved A/Bs: profiles/HISTORY.md "Round 5"), the RoboCup kernels 1.2-2.2 % slower -
// and a code object of their own also keeps edits to the Driving code away from the RoboCup code's instruction-cache phase.
#pragma once
#include <hip/hip_runtime.h>
#include "driving_dev.h"

#ifndef DRV_WAVES_PER_SIMD
#define DRV_WAVES_PER_SIMD 4 /* 4096 envs / (256 CUs * 4 SIMDs) */
#endif

// dense layout of one agent's Partial observation row (oracle/driving_partial.c has the same)
#define PV_CAP_CARS 24
#define PV_CAP_OBST 32
#define PV_CAP_PEDS 40
#define PV_CAP_LANES 16
// the LIMIT on a list's rows is its capacity in the layout - except in the test build that shows an overflow being reported
// (dynenv_amd/libdynenv_hip_testcaps.so, tests/test_gpu_boundary.py): the reference's lists have no cap (DrivingEnvironment.py:816-890),
// SURVEY Appendix E's worst case (39 cars) is above these capacities and astronomically unlikely (15 of ~30 objects misclassified / false
// positives at 0.4 % each), so rows beyond a limit are dropped and error bit 3 tells the host
#ifndef PV_LIM_CARS
#define PV_LIM_CARS PV_CAP_CARS
#define PV_LIM_OBST PV_CAP_OBST
#define PV_LIM_PEDS PV_CAP_PEDS
#define PV_LIM_LANES PV_CAP_LANES
#endif
static_assert(PV_LIM_CARS <= PV_CAP_CARS && PV_LIM_OBST <= PV_CAP_OBST && PV_LIM_PEDS <= PV_CAP_PEDS && PV_LIM_LANES <= PV_CAP_LANES, "limits inside the layout");
#define PV_DIM (9 + PV_CAP_CARS * 7 + PV_CAP_OBST * 6 + PV_CAP_PEDS * 2 + PV_CAP_LANES * 4 + 4)
#define PV_OFF_CARS 9
#define PV_OFF_OBST (9 + PV_CAP_CARS * 7)
#define PV_OFF_PEDS (PV_OFF_OBST + PV_CAP_OBST * 6)
#define PV_OFF_LANES (PV_OFF_PEDS + PV_CAP_PEDS * 2)

extern "C" __global__ void drv_tick_advance_kernel(DrvState S, int flipPv);
extern "C" __global__ void __launch_bounds__(64, DRV_WAVES_PER_SIMD)
drv_step_kernel(DrvState S, const int* __restrict__ actions, float* __restrict__ obs, double* __restrict__ rewards, uint8_t* __restrict__ dones);
extern "C" __global__ void __launch_bounds__(64, DRV_WAVES_PER_SIMD)
drv_step_partial_kernel(DrvState S, const int* __restrict__ actions, double* __restrict__ rewards, uint8_t* __restrict__ dones,
                        float* __restrict__ pobs, int pvNoise, double pvMagn);
extern "C" __global__ void __launch_bounds__(64) drv_obs_kernel(DrvState S, float* __restrict__ obs);
extern "C" __global__ void __launch_bounds__(64) drv_reset_kernel(DrvState S);
extern "C" __global__ void drv_stats_kernel(DrvState S, double* ep_r, double* ep_pos_r, double* ep_obs_r, int* goals);
extern "C" __global__ void drv_counts_kernel(DrvState S, int* counts);
extern "C" __global__ void math_selftest_kernel(const double* x, const double* y, int n, double* out);
extern "C" __global__ void __launch_bounds__(64, 4) drv_partial_obs_kernel(DrvState S, int noiseType, double magn, float* __restrict__ obs);
extern "C" __global__ void __launch_bounds__(64, 4) drv_partial_obs_deferred_kernel(DrvState S, int noiseType, double magn, float* __restrict__ obs);

// host helpers defined in driving_tu.hip (they touch that translation unit's device symbols / literal tables)
hipError_t drv_upload_consts(const DrvConst& c);   // -> __constant__ DrvConst C
bool drv_literals_ok(const DrvConst& c);           // the RoadK / CarK literals of the device code equal the computed constants, bit for bit
hipError_t drv_prof_read(int which, void* dst, size_t bytes);  // -DDRV_PROFILE builds: g_dbgr / g_dbgw / g_dbgp / g_dbgs / g_dbgl / g_pvprof = 0..5
