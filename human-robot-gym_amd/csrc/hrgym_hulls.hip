// hrgym_hulls.hip -- the ReachHuman kernels compiled once more with the arm links' CONVEX HULLS as collision geometry (hrg_model_desc.robot_hulls = 1; robot.xml:29-55:
// mesh geoms, which MuJoCo convexifies at compile time): inside the collide rounds, a pair of an arm link with a human capsule or the table / floor plane that passed
// the bounding-capsule test runs the hull narrowphase before it is reported (hrgym_hull.h: support mapping over the vertices with lanes = vertices, GJK with its
// simplex in the collide part of the LDS union).  Its own translation unit, so the capsule-geometry kernels carry none of it: no code, no LDS, no registers.
#define HRG_HULLS 1
#include "hrgym_hip.hip"

// ---- test tap (include/hrgym.h: hrg_test_hull_queries): the wave routines on their own, one query per wavefront -- tests/test_hulls.py compares them with the
// oracle's hull_support / gjk_hull_segment / hull_lowest over random poses.  Pose and segment go through LDS as in the step kernel (kR / kp, hcap).
struct HullQuery { double R[9], p[3], s1[3], s2[3]; int32_t hull, pad; };
__global__ __launch_bounds__(64) void hrg_test_hull_kernel(const double* __restrict__ verts, const int32_t* __restrict__ off, const HullQuery* __restrict__ q, int n, double* __restrict__ out) {
  const int k = (int)blockIdx.x, lane = hrg_lane();
  if (k >= n) return;
  Lds& L = g_L;
  if (lane < 9) L.kR[0][lane] = q[k].R[lane];
  if (lane < 3) { L.kp[0][lane] = q[k].p[lane]; L.hcap[0][lane] = q[k].s1[lane]; L.hcap[0][3 + lane] = q[k].s2[lane]; }
  wave_sync();
  const int h = __builtin_amdgcn_readfirstlane(q[k].hull), o0 = __builtin_amdgcn_readfirstlane(off[h]), o1 = __builtin_amdgcn_readfirstlane(off[h + 1]);
  const HullRef H = {verts + 3 * o0, o1 - o0, L.kR[0], L.kp[0]};
  double wa[3], wb[3], low[3];
  const double d = gjk_hull_segment_wave(H, &L.hcap[0][0], &L.hcap[0][3], wa, wb);
  wave_sync();
  hull_lowest_wave(H, low);
  if (lane == 0) {
    double* o = out + 10 * (size_t)k;
    o[0] = d;
    for (int a = 0; a < 3; a++) { o[1 + a] = wa[a]; o[4 + a] = wb[a]; o[7 + a] = low[a]; }
  }
}
extern "C" int hrg_test_hull_queries(const double* verts_host, const int32_t* off_host, const void* queries_host, int32_t n, double* out_host) {
  if (!verts_host || !off_host || !queries_host || !out_host || n <= 0 || off_host[0] != 0) return -1;
  for (int h = 0; h < HRG_NHULL; h++) if (!(off_host[h + 1] > off_host[h] + 3)) return -1;
  const HullQuery* qh = (const HullQuery*)queries_host;
  for (int k = 0; k < n; k++) if (qh[k].hull < 0 || qh[k].hull >= HRG_NHULL) return -1;
  double *dv = nullptr, *dout = nullptr;
  int32_t* doff = nullptr;
  HullQuery* dq = nullptr;
  const size_t vb = sizeof(double) * 3 * (size_t)off_host[HRG_NHULL];
  int rc = -1;
  if (hipMalloc(&dv, vb) == hipSuccess && hipMalloc(&doff, sizeof(int32_t) * (HRG_NHULL + 1)) == hipSuccess && hipMalloc(&dq, sizeof(HullQuery) * (size_t)n) == hipSuccess &&
      hipMalloc(&dout, sizeof(double) * 10 * (size_t)n) == hipSuccess && hipMemcpy(dv, verts_host, vb, hipMemcpyHostToDevice) == hipSuccess &&
      hipMemcpy(doff, off_host, sizeof(int32_t) * (HRG_NHULL + 1), hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dq, qh, sizeof(HullQuery) * (size_t)n, hipMemcpyHostToDevice) == hipSuccess) {
    hipLaunchKernelGGL(hrg_test_hull_kernel, dim3((unsigned)n), dim3(64), 0, 0, dv, doff, dq, (int)n, dout);
    if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(out_host, dout, sizeof(double) * 10 * (size_t)n, hipMemcpyDeviceToHost) == hipSuccess) rc = 0;
  }
  hipFree(dv); hipFree(doff); hipFree(dq); hipFree(dout);
  return rc;
}
