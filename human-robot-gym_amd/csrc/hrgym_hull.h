// hrgym_hull.h — convex hulls of the arm links as collision geometry (hrg_model_desc.robot_hulls; robot.xml:29-55: mesh geoms, which MuJoCo convexifies).
// The whole wave works on ONE (hull, other geom) pair at a time: the support mapping -- the hull vertex farthest along a direction -- runs over the vertices
// with lanes = vertices (a hull has 122 .. 2211 of them) and a wave arg-max; the GJK iteration around it (Gilbert / Johnson / Keerthi; closest-point
// sub-problems after Ericson, Real-Time Collision Detection 5.1) is wave-uniform: every lane computes the same numbers.  Its simplex -- up to four points of the
// Minkowski difference with their witnesses -- lives in the collide part of the env's LDS image (Lds::gjk), written by lane 0 and indexed dynamically there; in
// registers it would be 36 doubles per lane behind dynamic indices (scratch), and as a real function its call made the register allocator spill 290 VGPRs
// across the whole step kernel (727 scratch loads; 4 x the step time).  Inlined with the simplex in LDS the step kernel keeps its allocation.
// Restated side by side in oracle/hrg_oracle.c (hull_support, gjk_hull_segment, hull_lowest).
#pragma once
#include "hrgym_device.h"

struct HullRef { const double* v; int n; const double* R; const double* p; };   // body-frame vertices (device memory); world pose of the body (LDS / constants)

// vertex farthest along the world direction d (the lowest index on a tie): index; world point in out.  Wave-uniform result.
DI int hull_support_wave(const HullRef& H, const double* d, double* out) {
  const int lane = hrg_lane();
  const double dl0 = H.R[0] * d[0] + H.R[3] * d[1] + H.R[6] * d[2], dl1 = H.R[1] * d[0] + H.R[4] * d[1] + H.R[7] * d[2], dl2 = H.R[2] * d[0] + H.R[5] * d[1] + H.R[8] * d[2];
  double bv = -1e300;
  int bi = 0x7fffffff;
  // four vertices per lane and trip: their twelve loads are in flight together (the table sits in L2; one wave per SIMD-slot pays the full latency per trip otherwise --
  // link 4 has 2211 vertices = 35 trips per support call, ~ 8 calls per GJK query)
  int i = lane;
#pragma unroll 1
  for (; i + 192 < H.n; i += 256) {
    double t[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const double* v = H.v + 3 * (i + 64 * u); t[u] = v[0] * dl0 + v[1] * dl1 + v[2] * dl2; }
#pragma unroll
    for (int u = 0; u < 4; u++) if (t[u] > bv) { bv = t[u]; bi = i + 64 * u; }
  }
#pragma unroll 1
  for (; i < H.n; i += 64) {
    const double t = H.v[3 * i] * dl0 + H.v[3 * i + 1] * dl1 + H.v[3 * i + 2] * dl2;
    if (t > bv) { bv = t; bi = i; }
  }
  const double mx = wave_max(bv);
  int cand = bv == mx ? bi : 0x7fffffff;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(cand, o, 64); cand = t < cand ? t : cand; }
  const int best = __builtin_amdgcn_readfirstlane(cand);
  const double vx = H.v[3 * best], vy = H.v[3 * best + 1], vz = H.v[3 * best + 2];
  out[0] = H.p[0] + H.R[0] * vx + H.R[1] * vy + H.R[2] * vz;
  out[1] = H.p[1] + H.R[3] * vx + H.R[4] * vy + H.R[5] * vz;
  out[2] = H.p[2] + H.R[6] * vx + H.R[7] * vy + H.R[8] * vz;
  return best;
}

// barycentric coordinates of the point of triangle (a, b, c) closest to the origin (Ericson 5.1.5)
DI void closest_triangle(const double* a, const double* b, const double* c, double& l0, double& l1, double& l2) {
  double ab[3], ac[3];
  v3sub(ab, b, a); v3sub(ac, c, a);
  const double d1 = -v3dot(ab, a), d2 = -v3dot(ac, a);
  const double d3 = -v3dot(ab, b), d4 = -v3dot(ac, b);
  const double d5 = -v3dot(ab, c), d6 = -v3dot(ac, c);
  const double vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
  if (d1 <= 0 && d2 <= 0) { l0 = 1; l1 = 0; l2 = 0; }
  else if (d3 >= 0 && d4 <= d3) { l0 = 0; l1 = 1; l2 = 0; }
  else if (vc <= 0 && d1 >= 0 && d3 <= 0) { const double v = d1 / (d1 - d3); l0 = 1 - v; l1 = v; l2 = 0; }
  else if (d6 >= 0 && d5 <= d6) { l0 = 0; l1 = 0; l2 = 1; }
  else if (vb <= 0 && d2 >= 0 && d6 <= 0) { const double w = d2 / (d2 - d6); l0 = 1 - w; l1 = 0; l2 = w; }
  else if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) { const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6)); l0 = 0; l1 = 1 - w; l2 = w; }
  else { const double den = 1.0 / (va + vb + vc), v = vb * den, w = vc * den; l0 = 1 - v - w; l1 = v; l2 = w; }
}

// point of the simplex (S = g_L.gjk, n points) closest to the origin: v; the simplex reduced to the vertices that carry it (n updated), their weights in
// S.lam; true when the origin is inside.  Every lane computes the same values from LDS; lane 0 writes.
DI bool simplex_closest(int& n, double* v) {
  GjkLds& S = g_L.gjk;
  const bool w0 = hrg_lane() == 0;
  if (n == 1) { if (w0) S.l[0] = 1; }
  else if (n == 2) {
    double ab[3];
    v3sub(ab, S.y[1], S.y[0]);
    const double den = v3dot(ab, ab);
    double t = den > 0 ? -v3dot(S.y[0], ab) / den : 0.0;
    t = t < 0 ? 0 : (t > 1 ? 1 : t);
    if (w0) { S.l[0] = 1 - t; S.l[1] = t; }
  } else if (n == 3) {
    double l0, l1, l2;
    closest_triangle(S.y[0], S.y[1], S.y[2], l0, l1, l2);
    if (w0) { S.l[0] = l0; S.l[1] = l1; S.l[2] = l2; }
  } else {   // tetrahedron (Ericson 5.1.6): the closest of the faces the origin lies outside of; inside all four -> the origin is in the simplex
    double best = 1e300, b0 = 0, b1 = 0, b2 = 0;
    int fb = -1;
#pragma unroll 1
    for (int f = 0; f < 4; f++) {
      // faces (0 1 2 | 3), (0 2 3 | 1), (0 3 1 | 2), (1 3 2 | 0)
      const int i0 = f == 3 ? 1 : 0, i1 = f == 0 ? 1 : (f == 1 ? 2 : 3), i2 = f == 0 ? 2 : (f == 1 ? 3 : (f == 2 ? 1 : 2)), i3 = f == 0 ? 3 : (f == 1 ? 1 : (f == 2 ? 2 : 0));
      const double *a = S.y[i0], *b = S.y[i1], *c = S.y[i2], *d = S.y[i3];
      double ab[3], ac[3], nrm[3], ad[3];
      v3sub(ab, b, a); v3sub(ac, c, a); v3cross(nrm, ab, ac); v3sub(ad, d, a);
      const double sp = -v3dot(a, nrm), sd = v3dot(ad, nrm);
      if (!(sp * sd < 0) && sd != 0) continue;
      double l0, l1, l2, q[3];
      closest_triangle(a, b, c, l0, l1, l2);
      for (int k = 0; k < 3; k++) q[k] = l0 * a[k] + l1 * b[k] + l2 * c[k];
      const double dq = v3dot(q, q);
      if (dq < best) { best = dq; fb = f; b0 = l0; b1 = l1; b2 = l2; }
    }
    if (fb < 0) { v3set(v, 0, 0, 0); return true; }
    if (w0) {
      const int i0 = fb == 3 ? 1 : 0, i1 = fb == 0 ? 1 : (fb == 1 ? 2 : 3), i2 = fb == 0 ? 2 : (fb == 1 ? 3 : (fb == 2 ? 1 : 2));
      S.l[0] = S.l[1] = S.l[2] = S.l[3] = 0;
      S.l[i0] = b0; S.l[i1] = b1; S.l[i2] = b2;
    }
  }
  wave_sync();
  int nk = 0;
  v3set(v, 0, 0, 0);
#pragma unroll 1
  for (int i = 0; i < n; i++) {
    const double li = S.l[i];
    if (!(li > 0)) continue;
    double y[3];
    v3cpy(y, S.y[i]);
    for (int k = 0; k < 3; k++) v[k] += li * y[k];
    if (w0) {
      S.lam[nk] = li;
      if (nk != i) { v3cpy(S.y[nk], y); v3cpy(S.a[nk], S.a[i]); v3cpy(S.b[nk], S.b[i]); S.ia[nk] = S.ia[i]; S.ib[nk] = S.ib[i]; }
    }
    nk++;
  }
  n = nk;
  wave_sync();
  return false;
}

// distance between the hull and the segment [s1, s2] (a capsule's axis, in LDS); witness points wa (on the hull) and wb (on the segment).  0 when they
// intersect.  Wave-uniform result.  `cutoff`: the caller only asks whether the two come closer than this -- every support point w bounds the distance from below by
// v.w / |v| (the supporting plane of the Minkowski difference), so the iteration stops at the first plane farther out than the cutoff and returns 1e300: the
// narrowphase of a pair whose bounding capsules touch but whose hull is clear ends after one or two support calls instead of converging on a distance nobody uses.
DI double gjk_hull_segment_wave(const HullRef& H, const double* s1, const double* s2, double* wa, double* wb, double cutoff = 1e300) {
  GjkLds& S = g_L.gjk;
  const bool w0 = hrg_lane() == 0;
  double v[3];
  int n = 1;
  {
    double a0[3];
    const double vx = H.v[0], vy = H.v[1], vz = H.v[2];
    a0[0] = H.p[0] + H.R[0] * vx + H.R[1] * vy + H.R[2] * vz; a0[1] = H.p[1] + H.R[3] * vx + H.R[4] * vy + H.R[5] * vz; a0[2] = H.p[2] + H.R[6] * vx + H.R[7] * vy + H.R[8] * vz;
    v3sub(v, a0, s1);
    if (w0) { v3cpy(S.y[0], v); v3cpy(S.a[0], a0); v3cpy(S.b[0], s1); S.ia[0] = 0; S.ib[0] = 0; S.lam[0] = 1; }
  }
  wave_sync();
  bool inside = false;
#pragma unroll 1
  for (int it = 0; it < 64; it++) {
    const double vv = v3dot(v, v);
    if (vv <= 1e-24) { inside = true; break; }
    const double dir[3] = {-v[0], -v[1], -v[2]};
    double a[3], w[3], b[3];
    const int ia = hull_support_wave(H, dir, a);
    const int ib = v3dot(v, s2) > v3dot(v, s1) ? 1 : 0;
    v3cpy(b, ib ? s2 : s1);
    v3sub(w, a, b);
    const double vw = v3dot(v, w);
    if (vw > 0 && vw * vw > cutoff * cutoff * vv) { v3set(wa, 0, 0, 0); v3set(wb, 0, 0, 0); return 1e300; }
    if (vv - vw <= 1e-12 * vv) break;
    bool seen = false;
    for (int q = 0; q < n; q++) if (S.ia[q] == ia && S.ib[q] == ib) seen = true;
    if (seen) break;
    if (w0) { v3cpy(S.y[n], w); v3cpy(S.a[n], a); v3cpy(S.b[n], b); S.ia[n] = ia; S.ib[n] = ib; }
    n++;
    wave_sync();
    if (simplex_closest(n, v)) { inside = true; break; }
  }
  v3set(wa, 0, 0, 0); v3set(wb, 0, 0, 0);
  if (inside) return 0.0;
#pragma unroll 1
  for (int q = 0; q < n; q++) { const double lq = S.lam[q]; for (int k = 0; k < 3; k++) { wa[k] += lq * S.a[q][k]; wb[k] += lq * S.b[q][k]; } }
  return fsqrt(v3dot(v, v));
}

// the hull against a horizontal plane -> out[3]: z = height of the lowest vertex, (x, y) = mean of the vertices within 1e-6 m of it (a link lying flat on a face
// or an edge touches in the middle of it, whichever vertex rounding makes the lowest).  Wave-uniform result.
DI void hull_lowest_wave(const HullRef& H, double* out) {
  const int lane = hrg_lane();
  const double r6 = H.R[6], r7 = H.R[7], r8 = H.R[8], pz = H.p[2];
  double zm = 1e300;
  int i = lane;
#pragma unroll 1
  for (; i + 192 < H.n; i += 256) {
    double z[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const double* v = H.v + 3 * (i + 64 * u); z[u] = pz + r6 * v[0] + r7 * v[1] + r8 * v[2]; }
#pragma unroll
    for (int u = 0; u < 4; u++) zm = z[u] < zm ? z[u] : zm;
  }
#pragma unroll 1
  for (; i < H.n; i += 64) {
    const double z = pz + r6 * H.v[3 * i] + r7 * H.v[3 * i + 1] + r8 * H.v[3 * i + 2];
    zm = z < zm ? z : zm;
  }
  const double zmin = -wave_max(-zm);
  double sx = 0, sy = 0, cnt = 0;
#pragma unroll 2
  for (int i = lane; i < H.n; i += 64) {
    const double vx = H.v[3 * i], vy = H.v[3 * i + 1], vz = H.v[3 * i + 2];
    const double z = pz + r6 * vx + r7 * vy + r8 * vz;
    if (z <= zmin + 1e-6) { sx += H.p[0] + H.R[0] * vx + H.R[1] * vy + H.R[2] * vz; sy += H.p[1] + H.R[3] * vx + H.R[4] * vy + H.R[5] * vz; cnt += 1; }
  }
  const double tc = wave_sum(cnt);
  out[0] = wave_sum(sx) / tc; out[1] = wave_sum(sy) / tc; out[2] = zmin;
}
