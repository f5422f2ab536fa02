// driving_tu.hip - the Driving translation unit: the kernels of driving_kernels.hip / driving_partial.hip and the host helpers that
// need this unit's device symbols.  Compiled with -Os (dynenv_amd/build.py; driving_host.h says why), linked with dynenv_capi.hip.
#include <hip/hip_runtime.h>

#include "driving_kernels.hip"
#include "driving_partial.hip"

hipError_t drv_upload_consts(const DrvConst& c) { return hipMemcpyToSymbol(HIP_SYMBOL(C), &c, sizeof(c)); }

// the device code spells the road constants as literals (RoadK<R>): they must equal the computed ones bit for bit
template <int R>
static bool road_literals_ok(const DrvRoad& r) {
  return r.p0.x == RoadK<R>::p0x && r.p0.y == RoadK<R>::p0y && r.dir.x == RoadK<R>::dirx && r.dir.y == RoadK<R>::diry &&
         (double)r.nLanes * r.width + 5.0 == RoadK<R>::lat && r.length == RoadK<R>::length &&
         r.dirAngle == RoadK<R>::dirAngle && r.cosDir0 == RoadK<R>::cosDir0 && r.normal.x == RoadK<R>::nx &&
         r.normal.y == RoadK<R>::ny;
}

// ... and so are the car-type constants (CarK)
static bool car_literals_ok(const DrvConst& c) {
  const double m[4] = {CarK::carMass0, CarK::carMass1, CarK::carMass2, CarK::carMass3}, hx[4] = {CarK::carHx0, CarK::carHx1, CarK::carHx2, CarK::carHx3};
  const double hy[4] = {CarK::carHy0, CarK::carHy1, CarK::carHy2, CarK::carHy3}, pw[4] = {CarK::carPower0, CarK::carPower1, CarK::carPower2, CarK::carPower3};
  const double in[4] = {CarK::carInertia0, CarK::carInertia1, CarK::carInertia2, CarK::carInertia3};
  for (int t = 0; t < 4; ++t)
    if (c.carMass[t] != m[t] || c.carHx[t] != hx[t] || c.carHy[t] != hy[t] || c.carPower[t] != pw[t] || c.carInertia[t] != in[t]) return false;
  return c.pedMass == CarK::pedMass && c.pedInertia == CarK::pedInertia;
}


bool drv_literals_ok(const DrvConst& c) { return road_literals_ok<0>(c.roads[0]) && road_literals_ok<1>(c.roads[1]) && car_literals_ok(c); }

hipError_t drv_prof_read(int which, void* dst, size_t bytes) {
#ifdef DRV_PROFILE
  switch (which) {
    case 0: return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_dbgr), bytes);
    case 1: return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_dbgw), bytes);
    case 2: return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_dbgp), bytes);
    case 3: return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_dbgs), bytes);
    case 4: return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_dbgl), bytes);
    case 5: return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_pvprof), bytes);
  }
#endif
  (void)which; (void)dst; (void)bytes;
  return hipErrorInvalidValue;
}
