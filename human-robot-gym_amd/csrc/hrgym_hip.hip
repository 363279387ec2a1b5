// hrgym_hip.hip — step/reset kernels and the extern "C" ABI of libhrgym_hip.so (include/hrgym.h).
// One 64-lane wavefront per environment; see hrgym_device.h for the lane-role map.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "hrgym_kernels.h"

// ================================================================================================ solver
// mj_step for the robot tree (environments/manipulation/human_env.py:523): smooth acceleration, soft
// constraints (friction loss, joint limits, pyramidal contacts) by primal Newton + exact line search,
// semi-implicit Euler with implicit joint damping.  Lanes = constraint rows in fixed slots:
//   0..7 friction loss of dof r | 8..23 joint limit (dof, lo/hi) | 24..63 contact c, pyramid edge d.
template <class MD>
DI void impedance(const MD& m, double x0, double* imp) {  // the stiffness / damping of solref are model constants: DevModel.sol_K, sol_Bd
  const double d0 = m.solimp[0], dmax = m.solimp[1], width = m.solimp[2], mid = m.solimp[3], power = m.solimp[4];
  const double x = fabs(x0) / width;
  double y;
  if (x >= 1) y = 1;
  else if (x <= 0) y = 0;
  else if (x <= mid) { const double u = x / mid; y = u * u * mid; }   // solimp power 2 (MuJoCo's default; hrg_batch_create refuses another): the square -- a pow() in
  else { const double u = (1 - x) / (1 - mid); y = 1 - u * u * (1 - mid); }   // this spot put ~300 instructions and the spills around them into every step kernel
  (void)power;
  *imp = d0 + y * (dmax - d0);
}

// lim = floss / D of a friction row, computed once per row and substep (an FP64 division is a ~30-instruction sequence)
DI void row_cost(int type, double D, double floss, double lim, double x, double* c, double* g, double* h) {
  if (type == 2) { *c = 0.5 * D * x * x; *g = D * x; *h = D; }  // equality (weld) row: quadratic on both sides
  else if (type == 1) {
    if (x < 0) { *c = 0.5 * D * x * x; *g = D * x; *h = D; } else { *c = 0; *g = 0; *h = 0; }
  } else {
    if (x <= -lim) { *c = floss * (-x - 0.5 * lim); *g = -floss; *h = 0; }
    else if (x >= lim) { *c = floss * (x - 0.5 * lim); *g = floss; *h = 0; }
    else { *c = 0.5 * D * x * x; *g = D * x; *h = D; }
  }
}

DI int row_zone(int type, double lim, double x) {  // which quadratic / linear piece of its cost a row is in
  if (type == 2) return 0;
  if (type == 1) return x < 0;
  return x <= -lim ? -1 : (x >= lim ? 1 : 0);
}

// mj_Euler's implicit joint damping for the robot tree: qvel += h (M + h D)^-1 M qacc -- without a second 8 x 8 inverse.  D is diagonal: the two finger slides carry
// real damping (h d ~ 0.4 against an inertia of ~ 1), the six arm hinges 1e-4 (h d / M ~ 1e-5).  With A = M + h D_fingers,
//   A^-1 v = M^-1 v - M^-1[:, f] S^-1 (M^-1 v)[f],   S = (h D_f)^-1 + M^-1[f, f]                     (Woodbury, rank 2, exact)
//   (A + h D_arm)^-1 t = A^-1 t - A^-1 (h D_arm A^-1 t) + O((h d_arm / M)^2 ~ 1e-10)                  (one Neumann term; hrg_batch_create checks the ratio)
// lane (i, j) holds entry (i, j) of M and of M^-1; vectors go through L.d.  Returns nothing: qvel / qpos of the robot tree are integrated in place.
DI void euler_robot_tree(ModelPtr dm, int lane, double Mij, double Minv) {
  Lds& L = g_L;
  const auto& m = dm->m;
  hrg_env_state& s = L.st;
  const int mi = lane >> 3, mj = lane & 7;
  const double h = m.timestep;
  const double hd6 = h * m.jnt_damping[NARM], hd7 = h * m.jnt_damping[NARM + 1];
  const bool fingers = hd6 > 0 && hd7 > 0;
  const double S00 = (fingers ? 1.0 / hd6 : 0.0) + lane_value<6 * 9>(Minv), S01 = lane_value<6 * 8 + 7>(Minv), S11 = (fingers ? 1.0 / hd7 : 0.0) + lane_value<7 * 9>(Minv);
  const double idet = 1.0 / (S00 * S11 - S01 * S01);
  auto Ainv = [&](const double* v) -> double {   // (A^-1 v)_mi in every lane of row mi
    const double y = row8_sum(Minv * v[mj]);
    if (!fingers) return y;
    const double y6 = lane_value<6 * 8>(y), y7 = lane_value<7 * 8>(y);
    const double z0 = (S11 * y6 - S01 * y7) * idet, z1 = (S00 * y7 - S01 * y6) * idet;
    return y - row8_sum(mj == 6 ? Minv * z0 : (mj == 7 ? Minv * z1 : 0.0));
  };
  const double t = matvec_lanes(Mij, L.qacc, lane);
  if (lane < NV) s.qacc_warmstart[lane] = L.qacc[lane];
  wave_sync();
  if (mj == 0) L.d[mi] = t;
  wave_sync();
  const double x0 = Ainv(L.d);
  wave_sync();
  if (mj == 0) L.d[mi] = mi < NARM ? h * m.jnt_damping[mi] * x0 : 0.0;
  wave_sync();
  const double x = x0 - Ainv(L.d);
  if (mj == 0) {
    const double v = s.qvel[mi] + h * x;
    s.qvel[mi] = v;
    s.qpos[mi] = s.qpos[mi] + h * v;
  }
}

// returns 1 when the simulation diverged (MujocoException path, human_env.py:527-546)
#if HRG_BOX
// row a (0..5) of the box's inertia block times the 6 box entries v: m v_a for the translation, the world-frame rotational inertia
// row for the rotation (its off-diagonal entries are exact zeros for a cube)
DI double box_Mrow(double mass, int a, const double* v) {
  const Lds& L = g_L;
  if (a < 3) return mass * v[a];
  const int r = 3 * (a - 3);
  return L.bMr[r + (a - 3)] * v[a] + L.bMr[r + (a - 2) % 3] * v[3 + (a - 2) % 3] + L.bMr[r + (a - 1) % 3] * v[3 + (a - 1) % 3];
}
#endif
#if HRG_STACK
// ================================================================================================ solver (stacking)
// mj_step for the robot tree + the four free cubes of CollaborativeStackingCart: the same primal Newton method with exact line search as dynamics_step
// below, on 32 DoF and up to 128 constraint rows -- TWO rows per lane (r = lane, lane + 64):
//   0..7 friction loss | 8..23 joint limits | 24..35 the two cube <-> hand welds (hand, component) | 36..127 contact c, pyramid edge d.
// Contact rows of J are stored compact ([robot 8 | cube of geom 1: 6 | cube of geom 2: 6]).  M is block diagonal (8x8 robot, m 1 / I 1 per cube: cubes
// only, checked at create).  Without a robot-cube contact the Newton system splits: the robot block is factored in registers across the wave as in the
// ReachHuman kernel, the 24x24 cube block in LDS (packed); a robot-cube contact takes the packed 32x32 factorisation.
static_assert(NVS == 32, "entry (i, j) of the packed Hessian is decoded as (e >> 5, e & 31)");
struct SRow { bool active; int type, kind, dof; double sgn, D, floss, flim, aref, y, p; };   // kind: 0 joint row, 1 weld row, 2 contact row
DI double cube_mdiag(ModelPtr dm, int i) { return ((i - NV) % 6) < 3 ? dm->m.box_mass : dm->m.box_inertia[0]; }

PH_DYNSTEP int dynamics_step(const DevModel* __restrict__ dm_, int lane, int ncon) {
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  hrg_env_state& s = L.st;
  hrg_stack_state& sk = L.sk;
  const double h = m.timestep;
  const int mi = lane >> 3, mj = lane & 7;
  const int nc = ncon < NCON_DYN ? ncon : NCON_DYN;
  bool ok;
  const double Mij = L.M[lane];
  const double Minv = spd_inverse1(Mij, lane, &ok), M0inv = Minv;   // entry (mi, mj) of M^-1 of the robot tree
  if (!ok) return 1;
  if (lane < NV) {
    double act = L.ctrl[lane];
    if (lane >= NARM) act = clampd(m.finger_kp * (act - s.qpos[lane]), m.finger_forcerange[0], m.finger_forcerange[1]);
    L.Ma0[lane] = act - m.jnt_damping[lane] * s.qvel[lane] - L.bias[lane];
    L.qacc[lane] = s.qacc_warmstart[lane];
  } else if (lane < NVS) {   // free cubes: gravity; a cube's rotational inertia is isotropic, so there is no gyroscopic torque
    const int a = lane - NV, c = a / 6, k = a - 6 * c;
    const double a0v = k < 3 ? m.gravity[k] : 0.0;
    L.a0[lane] = a0v;
    L.Ma0[lane] = k < 3 ? m.box_mass * a0v : 0.0;
    L.qacc[lane] = sk.acc_warmstart[c][k];
  }
  wave_sync();
  {
    const double x = matvec_lanes(Minv, L.Ma0, lane);
    if (mj == 0) L.a0[mi] = x;
  }
  STAMP(20);
  // ---- per-contact bookkeeping: which cubes / whether the robot take part ----
  if (lane < NCON_DYN) {
    int ca = -1, cb = -1, rob = 0;
    if (lane < nc) {
      const Contact& cc = L.con[lane];
      ca = cc.b1 >= BODY_BOX ? cc.b1 - BODY_BOX : -1;
      cb = cc.b2 >= BODY_BOX ? cc.b2 - BODY_BOX : -1;
      rob = (cc.b1 >= 0 && cc.b1 < NV) || (cc.b2 >= 0 && cc.b2 < NV);
    }
    L.con_ca[lane] = ca; L.con_cb[lane] = cb; L.con_rob[lane] = rob;
  }
  wave_sync();
  // ---- this lane's two constraint rows ----
  SRow R[2];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int r = lane + 64 * k;
    SRow& w = R[k];
    w.active = false; w.type = 1; w.kind = 0; w.dof = 0; w.sgn = 0; w.D = 0; w.floss = 0; w.flim = 0; w.aref = 0; w.y = 0; w.p = 0;
    bool cand = false;
    double pos = 0, margin = 0, diag = 0, vel = 0;
    if (r < NV) {
      if (m.jnt_frictionloss[r] > 0) { cand = true; w.type = 0; w.floss = m.jnt_frictionloss[r]; diag = m.dof_invweight0[r]; w.dof = r; w.sgn = 1.0; vel = s.qvel[r]; }
    } else if (r < SROW_WELD0) {
      const int q = r - NV, dof = q >> 1, side = q & 1;
      const double dist = side ? m.jnt_range[dof][1] - s.qpos[dof] : s.qpos[dof] - m.jnt_range[dof][0];
      if (dist < 0) { cand = true; pos = dist; diag = m.dof_invweight0[dof]; w.dof = dof; w.sgn = side ? -1.0 : 1.0; vel = w.sgn * s.qvel[dof]; }
    } else if (r < SROW_CON0) {   // lh_weld_eq / rh_weld_eq (collaborative_stacking_cartesian_env.py:1255-1284): [p_cube - p_target; rotation vector of q_cube q_mocap^-1]
      const int q = r - SROW_WELD0, hd = q / 6, a = q - 6 * hd, cb = HRG_CUBE_L + hd;
      if (sk.weld_active[hd]) {
        const double qm[4] = {sk.mocap_quat[hd][0], sk.mocap_quat[hd][1], sk.mocap_quat[hd][2], sk.mocap_quat[hd][3]};
        if (a < 3) {   // where the weld wants the cube: mocap = cube o relpose  =>  p_cube = p_mocap - R_mocap relpos
          double Rm[9], tp[3];
          quat2mat(Rm, qm);
          m3mulv(tp, Rm, m.stack_weld_relpos);
          pos = sk.pos[cb][a] - (sk.mocap_pos[hd][a] - tp[a]);
        } else {
          const double qo[4] = {sk.quat[cb][0], sk.quat[cb][1], sk.quat[cb][2], sk.quat[cb][3]}, qc[4] = {qm[0], -qm[1], -qm[2], -qm[3]};
          double qe[4];
          quatmul(qe, qo, qc);
          if (qe[0] < 0) for (int z = 0; z < 4; z++) qe[z] = -qe[z];
          const double sn = fsqrt(qe[1] * qe[1] + qe[2] * qe[2] + qe[3] * qe[3]), ang = 2.0 * atan2(sn, qe[0]);
          pos = sn > 1e-12 ? qe[1 + (a - 3)] / sn * ang : 0.0;
        }
        cand = true; w.type = 2; w.kind = 1;
        diag = a < 3 ? 1.0 / m.box_mass : m.box_invweight_rot;
        w.dof = NV + 6 * cb + a; w.sgn = 1.0; vel = sk.vel[cb][a];
      }
    } else {
      const int q = r - SROW_CON0, c = q >> 2, d = q & 3;
      w.kind = 2; w.dof = q;
      if (c < nc) {
        const Contact& cc = L.con[c];
        const double n[3] = {cc.n[0], cc.n[1], cc.n[2]}, cp[3] = {cc.pos[0], cc.pos[1], cc.pos[2]};
        double t1[3], t2[3], dir[3];
        const double e1[3] = {1, 0, 0}, e2[3] = {0, 1, 0};
        v3cross(t1, n, fabs(n[0]) < 0.5 ? e1 : e2);
        v3scl(t1, t1, 1.0 / v3norm(t1));
        v3cross(t2, n, t1);
        const double sg = (d & 1) ? -1.0 : 1.0;
        for (int a = 0; a < 3; a++) dir[a] = n[a] + sg * m.friction_static * (d < 2 ? t1[a] : t2[a]);
        pos = cc.dist;
        margin = ((cc.g2 >= GEOM_HUMAN0 && cc.g2 < GEOM_TABLE) || (cc.g1 >= GEOM_HUMAN0 && cc.g1 < GEOM_TABLE)) ? m.contact_margin_human : 0.0;   // a human geom on either side
        const bool rb1 = cc.b1 >= 0 && cc.b1 < NV, rb2 = cc.b2 >= 0 && cc.b2 < NV;
        diag = (rb1 ? m.body_invweight0[cc.b1] : 0.0) + (rb2 ? m.body_invweight0[cc.b2] : 0.0);
        const int am1 = rb1 ? dm->anc_mask[cc.b1] : 0, am2 = rb2 ? dm->anc_mask[cc.b2] : 0;
        double nz = 0;
        double* Jr = L.Jc[q];
        if (!(rb1 || rb2)) { for (int i = 0; i < NV; i++) Jr[i] = 0.0; }
        else {
#pragma unroll 1
          for (int i = 0; i < NV; i++) {
            double t[3], v[3];
            v3cross(t, L.Sw[i], cp);
            v3add(v, L.Sv[i], t);
            const double jv = v3dot(dir, v);
            double acc = 0;
            if ((am1 >> i) & 1) acc += -1.0 * jv;
            if ((am2 >> i) & 1) acc += jv;
            Jr[i] = acc;
            vel += acc * s.qvel[i];
            nz += fabs(acc);
          }
        }
#pragma unroll
        for (int side = 0; side < 2; side++) {   // free bodies: J = -+dir . (v + w x r); body_invweight0 of a free body = 1/m
          const int body = side ? cc.b2 : cc.b1;
          double* Jb = Jr + 8 + 6 * side;
          if (body >= BODY_BOX) {
            const int cb = body - BODY_BOX;
            const double sgn = side ? 1.0 : -1.0;
            double rr[3], rxd[3];
            for (int a = 0; a < 3; a++) rr[a] = cp[a] - sk.pos[cb][a];
            v3cross(rxd, rr, dir);
            for (int a = 0; a < 3; a++) { Jb[a] = sgn * dir[a]; Jb[3 + a] = sgn * rxd[a]; vel += sgn * (dir[a] * sk.vel[cb][a] + rxd[a] * sk.vel[cb][3 + a]); nz += fabs(dir[a]) + fabs(rxd[a]); }
            diag += 1.0 / m.box_mass;
          } else for (int a = 0; a < 6; a++) Jb[a] = 0.0;
        }
        diag *= 1.0 + m.friction_static * m.friction_static;
        cand = nz > 0;
      }
    }
    w.active = cand && diag > 0;
    if (w.active) {
      double imp;
      const double K = dm->sol_K, Bd = dm->sol_Bd;
      impedance(m, pos - margin, &imp);
      w.aref = -Bd * vel - K * imp * (pos - margin);
      w.D = 1.0 / ((1 - imp) / imp * diag);
      if (w.type == 0) w.flim = w.floss / w.D;
    }
  }
  STAMP(21);
  COUNT(19, __popcll(__ballot(R[0].active)) + __popcll(__ballot(R[1].active)));
  const bool any_row = __any(R[0].active || R[1].active);
  // a robot-cube contact couples the robot block of the Newton system to the cube block
  bool cpl = false, ccpl = false;
  if (lane < nc) { cpl = L.con_rob[lane] && (L.con_ca[lane] >= 0 || L.con_cb[lane] >= 0); ccpl = L.con_ca[lane] >= 0 && L.con_cb[lane] >= 0; }
  const bool coupled = __any(cpl);
  // ... and a cube-cube contact couples two cubes; without either, the Newton system is block diagonal: the robot block and four 6x6 cube blocks, each inverted
  // in registers across the wave (the common case: cubes resting on the table, carried by the human's hands, falling)
  const bool blocks = !coupled && !__any(ccpl);
  const bool busy_rows = !blocks || __any(lane < nc && L.con_rob[lane]);
  wave_sync();
  auto rowdot = [&](const SRow& w, const double* x) -> double {
    if (!w.active) return 0.0;
    if (w.kind != 2) return w.sgn * x[w.dof];
    const int q = w.dof, c = q >> 2;
    const double* Jr = L.Jc[q];
    const int ca = L.con_ca[c], cb = L.con_cb[c];
    double t = 0;
    if (L.con_rob[c]) {
#pragma unroll
      for (int i = 0; i < NV; i++) t += Jr[i] * x[i];
    }
    if (ca >= 0) {
#pragma unroll
      for (int a = 0; a < 6; a++) t += Jr[8 + a] * x[NV + 6 * ca + a];
    }
    if (cb >= 0) {
#pragma unroll
      for (int a = 0; a < 6; a++) t += Jr[14 + a] * x[NV + 6 * cb + a];
    }
    return t;
  };
  // M x for the DoF of this lane (robot rows dense, cube DoF diagonal)
  auto Mrow = [&](const double* x) -> double {
    double t = 0;
    if (lane < NV) {
#pragma unroll
      for (int j = 0; j < NV; j++) t += L.M[lane * NV + j] * x[j];
    } else if (lane < NVS) t = cube_mdiag(dm, lane) * x[lane];
    return t;
  };
  if (!any_row) {
    if (lane < NVS) L.qacc[lane] = L.a0[lane];
    wave_sync();
  } else {
    { // warm start vs unconstrained acceleration: keep the cheaper point
      double c0 = 0, c1 = 0, g_, h_;
#pragma unroll
      for (int k = 0; k < 2; k++)
        if (R[k].active) {
          double t0, t1;
          row_cost(R[k].type, R[k].D, R[k].floss, R[k].flim, rowdot(R[k], L.qacc) - R[k].aref, &t0, &g_, &h_);
          row_cost(R[k].type, R[k].D, R[k].floss, R[k].flim, rowdot(R[k], L.a0) - R[k].aref, &t1, &g_, &h_);
          c0 += t0; c1 += t1;
        }
      const double ei = L.qacc[mi] - L.a0[mi], ej = L.qacc[mj] - L.a0[mj];
      double quad = 0.5 * Mij * ei * ej;
      if (lane >= NV && lane < NVS) { const double eb = L.qacc[lane] - L.a0[lane]; quad += 0.5 * cube_mdiag(dm, lane) * eb * eb; }
      const double cost_ws = wave_sum(quad + c0), cost_a0 = wave_sum(c1);
      wave_sync();
      if (!(cost_ws < cost_a0)) { if (lane < NVS) L.qacc[lane] = L.a0[lane]; }
      wave_sync();
    }
    STAMP(22);
    bool h_is_m = true;
    double Hrob = Minv;   // inverse of the robot block of the Newton Hessian
#pragma unroll 1
    for (int it = 0; it < m.solver_iters; it++) {
      COUNT(16, 1);
      double gg[2], hh[2];
#pragma unroll
      for (int k = 0; k < 2; k++) {
        double cc_ = 0;
        gg[k] = 0; hh[k] = 0;
        R[k].y = rowdot(R[k], L.qacc) - R[k].aref;
        if (R[k].active) row_cost(R[k].type, R[k].D, R[k].floss, R[k].flim, R[k].y, &cc_, &gg[k], &hh[k]);
        L.rg[lane + 64 * k] = gg[k]; L.rh[lane + 64 * k] = hh[k];
      }
      wave_sync();
      double gm = 0;
      if (lane < NVS) {   // gradient of the DoF of this lane
        double t = Mrow(L.qacc) - L.Ma0[lane];
        gm = t;
        if (lane < NV) {
          t += L.rg[lane];
          t += L.rg[NV + 2 * lane];
          t -= L.rg[NV + 2 * lane + 1];
          // (the robot part of a contact row without a robot body is stored as zeros: no test per contact, the loads of successive rows overlap)
#pragma unroll 4
          for (int q = 0; q < 4 * nc; q++) t += L.Jc[q][lane] * L.rg[SROW_CON0 + q];
        } else {
          const int a = lane - NV, cu = a / 6, k = a - 6 * cu;
          if (cu >= HRG_CUBE_L) t += L.rg[SROW_WELD0 + 6 * (cu - HRG_CUBE_L) + k];
#pragma unroll 2
          for (int c = 0; c < nc; c++) {   // branch-free: both cube parts of the four rows are loaded, the one that is this lane's cube is kept
            const double wa = L.con_ca[c] == cu ? 1.0 : 0.0, wb = L.con_cb[c] == cu ? 1.0 : 0.0;
            double ta = 0, tb = 0;
#pragma unroll
            for (int d = 0; d < 4; d++) { const double rgq = L.rg[SROW_CON0 + 4 * c + d]; ta += L.Jc[4 * c + d][8 + k] * rgq; tb += L.Jc[4 * c + d][14 + k] * rgq; }
            t += wa * ta + wb * tb;
          }
        }
        L.g[lane] = t;
      }
      wave_sync();
      {
        const double gl = lane < NVS ? L.g[lane] : 0.0, ml = lane < NVS ? L.Ma0[lane] : 0.0;
        const double gn = wave_sum(gl * gl), sc = wave_sum(ml * ml);
        if (fsqrt(gn) <= m.solver_tol * (1.0 + fsqrt(sc))) break;
      }
      STAMP(27);
      // ---- Newton Hessian: M + sum_r h_r J_r' J_r ----
      if (!coupled) {   // robot block in registers (lanes = (mi, mj)), as in the ReachHuman solver
        // ... whose Hessian is M until a row on the robot tree has curvature (a joint row in its quadratic zone, a contact of a robot body): M^-1 is already held
        bool rc = false;
#pragma unroll
        for (int k = 0; k < 2; k++) rc |= hh[k] != 0 && (R[k].kind == 0 || (R[k].kind == 2 && L.con_rob[(lane + 64 * k - SROW_CON0) >> 2]));
        if (__any(rc) || !h_is_m) {
          double hval = Mij;
          if (mi == mj) { hval += L.rh[mi]; hval += L.rh[NV + 2 * mi]; hval += L.rh[NV + 2 * mi + 1]; }
#pragma unroll 4
          for (int q = 0; q < 4 * nc; q++) hval += L.rh[SROW_CON0 + q] * L.Jc[q][mi] * L.Jc[q][mj];   // (zeros where the row has no robot part or no curvature)
          Hrob = spd_inverse1(hval, lane, &ok);
          if (!ok) break;
          h_is_m = false;
        }
        const double x = -matvec_lanes(Hrob, L.g, lane);
        if (mj == 0) L.d[mi] = x;
      }
      if (blocks) {   // four independent cube blocks: 6x6 padded to the 8x8 lane layout with a unit diagonal, inverted together
        bool good = true;
        double sv[NCUBE];
#pragma unroll
        for (int cu = 0; cu < NCUBE; cu++) {
          sv[cu] = mi == mj ? (mi < 6 ? cube_mdiag(dm, NV + mi) : 1.0) : 0.0;
          if (mi == mj && mi < 6 && cu >= HRG_CUBE_L) sv[cu] += L.rh[SROW_WELD0 + 6 * (cu - HRG_CUBE_L) + mi];
        }
        if (mi < 6 && mj < 6) {
#pragma unroll 2
          for (int c = 0; c < nc; c++) {
            const int cu = L.con_cb[c];   // (without cube-cube contacts a cube is always geom 2: table, floor; -1: a robot-only contact, whose cube part is zero)
            double t = 0;
#pragma unroll
            for (int d = 0; d < 4; d++) t += L.rh[SROW_CON0 + 4 * c + d] * L.Jc[4 * c + d][14 + mi] * L.Jc[4 * c + d][14 + mj];
#pragma unroll
            for (int q = 0; q < NCUBE; q++) sv[q] += cu == q ? t : 0.0;
          }
        }
        spd_inverse4(sv, lane, &ok);
        good = ok;
#pragma unroll
        for (int cu = 0; cu < NCUBE; cu++) {
          const double x = -row8_sum(sv[cu] * (mj < 6 ? L.g[NV + 6 * cu + mj] : 0.0));
          if (mj == 0 && mi < 6) L.d[NV + 6 * cu + mi] = x;
        }
        STAMP(23);
        COUNT(18, 1);
        if (!good) break;
        STAMP(24);
      } else {   // a cube-cube or robot-cube contact couples the blocks: the whole system as tiles in registers
        Tiles<5> T;
#pragma unroll
        for (int a = 0; a < 5; a++)
#pragma unroll
          for (int b_ = 0; b_ < 5; b_++) T.t[a][b_] = 0.0;
        if (coupled) {
          double hval = Mij;
          if (mi == mj) { hval += L.rh[mi]; hval += L.rh[NV + 2 * mi]; hval += L.rh[NV + 2 * mi + 1]; }
          T.t[0][0] = hval;
        }
#pragma unroll
        for (int cu = 0; cu < NCUBE; cu++) {
          double v = mi == mj ? (mi < 6 ? cube_mdiag(dm, NV + mi) : 1.0) : 0.0;
          if (mi == mj && mi < 6 && cu >= HRG_CUBE_L) v += L.rh[SROW_WELD0 + 6 * (cu - HRG_CUBE_L) + mi];
          T.t[1 + cu][1 + cu] = v;
        }
#pragma unroll 1
        for (int c = 0; c < nc; c++) {   // contact c: h J' J on its columns [robot 8 | cube of geom 1 | cube of geom 2] (absent parts are stored as zeros)
          const int ca = __builtin_amdgcn_readfirstlane(L.con_ca[c]), cb = __builtin_amdgcn_readfirstlane(L.con_cb[c]);
          double Srr = 0, Sra = 0, Sar = 0, Srb = 0, Sbr = 0, Saa = 0, Sab = 0, Sba = 0, Sbb = 0;
#pragma unroll
          for (int d = 0; d < 4; d++) {
            const double* Jr = L.Jc[4 * c + d];
            const double hq = L.rh[SROW_CON0 + 4 * c + d];
            const double ri = Jr[mi], rj = Jr[mj];
            const double ai = mi < 6 ? Jr[8 + mi] : 0.0, aj = mj < 6 ? Jr[8 + mj] : 0.0, bi = mi < 6 ? Jr[14 + mi] : 0.0, bj = mj < 6 ? Jr[14 + mj] : 0.0;
            Srr += hq * ri * rj; Sra += hq * ri * aj; Sar += hq * ai * rj; Srb += hq * ri * bj; Sbr += hq * bi * rj;
            Saa += hq * ai * aj; Sab += hq * ai * bj; Sba += hq * bi * aj; Sbb += hq * bi * bj;
          }
          if (coupled) T.t[0][0] += Srr;
#pragma unroll
          for (int a = 0; a < NCUBE; a++) {   // ca, cb are wave-uniform: scalar branches, the tile indices stay compile-time
            if (ca == a) {
              T.t[1 + a][1 + a] += Saa;
              if (coupled) { T.t[0][1 + a] += Sra; T.t[1 + a][0] += Sar; }
#pragma unroll
              for (int b_ = 0; b_ < NCUBE; b_++) if (cb == b_) { T.t[1 + a][1 + b_] += Sab; T.t[1 + b_][1 + a] += Sba; }
            }
            if (cb == a) {
              T.t[1 + a][1 + a] += Sbb;
              if (coupled) { T.t[0][1 + a] += Srb; T.t[1 + a][0] += Sbr; }
            }
          }
        }
        STAMP(23);
        COUNT(18, 1);
        bool good = true;
        if (coupled) {
          tiles_pivots<5, 0, 0>(T, mi, mj, 8, good);
          tiles_pivots<5, 1, 0>(T, mi, mj, 6, good); tiles_pivots<5, 2, 0>(T, mi, mj, 6, good); tiles_pivots<5, 3, 0>(T, mi, mj, 6, good); tiles_pivots<5, 4, 0>(T, mi, mj, 6, good);
        } else {
          tiles_pivots<5, 1, 1>(T, mi, mj, 6, good); tiles_pivots<5, 2, 1>(T, mi, mj, 6, good); tiles_pivots<5, 3, 1>(T, mi, mj, 6, good); tiles_pivots<5, 4, 1>(T, mi, mj, 6, good);
        }
        if (!good) break;
        STAMP(24);
        {   // direction d = -H^-1 g, one tile row at a time
          double gb[5];
          gb[0] = L.g[mj];
#pragma unroll
          for (int cu = 0; cu < NCUBE; cu++) gb[1 + cu] = mj < 6 ? L.g[NV + 6 * cu + mj] : 0.0;
#pragma unroll
          for (int a = 0; a < 5; a++) {
            if (a == 0 && !coupled) continue;   // the robot block's direction was written above
            double acc = 0;
#pragma unroll
            for (int b_ = 0; b_ < 5; b_++) if (b_ > 0 || coupled) acc += T.t[a][b_] * gb[b_];
            const double x = -row8_sum(acc);
            if (mj == 0 && (a == 0 || mi < 6)) L.d[a == 0 ? mi : NV + 6 * (a - 1) + mi] = x;
          }
        }
      }
      wave_sync();
#pragma unroll
      for (int k = 0; k < 2; k++) R[k].p = rowdot(R[k], L.d);
      double dd = 0, Mdi = 0;
      if (lane < NVS) { dd = L.d[lane]; Mdi = Mrow(L.d); }
      STAMP(25);
      const double dMd = wave_sum(dd * Mdi), gd0 = wave_sum(dd * gm);
      double al = 1.0, lo = 0, hi = -1;
      const double d1_0 = gd0 + wave_sum(gg[0] * R[0].p + gg[1] * R[1].p);
      const double noise = fabs(gd0) + dMd + wave_sum(fabs(gg[0] * R[0].p) + fabs(gg[1] * R[1].p));   // magnitude of the terms phi' is summed from
#pragma unroll 1
      for (int ls = 0; ls < 40; ls++) {
        COUNT(17, 1);
        double sg_ = 0, sh_ = 0;
#pragma unroll
        for (int k = 0; k < 2; k++)
          if (R[k].active) {
            double c2, g2, h2;
            row_cost(R[k].type, R[k].D, R[k].floss, R[k].flim, R[k].y + al * R[k].p, &c2, &g2, &h2);
            sg_ += g2 * R[k].p; sh_ += h2 * R[k].p * R[k].p;
          }
        const double d1 = gd0 + al * dMd + wave_sum(sg_);
        const double d2 = dMd + wave_sum(sh_);
        if (fabs(d1) <= 1e-10 * fabs(d1_0) || fabs(d1) <= 1e-14 * noise) break;   // converged, or down at the rounding noise of the sum's own terms
        if (d1 < 0) lo = al; else hi = al;
        double nx = al - d1 / d2;
        if (hi < 0) { if (!(nx > lo)) nx = 2 * al; }
        else if (!(nx > lo && nx < hi)) nx = 0.5 * (lo + hi);
        al = nx;
      }
      STAMP(26);
      if (lane < NVS) L.qacc[lane] += al * dd;
      wave_sync();
      bool moved = false;
#pragma unroll
      for (int k = 0; k < 2; k++)
        if (R[k].active) moved = moved || row_zone(R[k].type, R[k].flim, R[k].y) != row_zone(R[k].type, R[k].flim, R[k].y + R[k].p);
      if (al == 1.0 && !__any(moved)) break;
    }
  }
  const bool badacc = lane < NVS && !(fabs(L.qacc[lane]) < 1e10);
  if (__any(badacc)) return 1;
  // mj_Euler with implicit joint damping for the robot tree
  euler_robot_tree(dm, lane, Mij, M0inv);
  { // the cubes' free joints: lanes 8..31 = (cube, component); quaternions by lanes 0..3 of each cube's group
    double vnew = 0;
    const int a = lane - NV, cu = lane >= NV && lane < NVS ? a / 6 : 0, k = lane >= NV && lane < NVS ? a - 6 * cu : 0;
    if (lane >= NV && lane < NVS) {
      const double acc = L.qacc[lane];
      sk.acc_warmstart[cu][k] = acc;
      vnew = sk.vel[cu][k] + h * acc;
      sk.vel[cu][k] = vnew;
      if (k < 3) { const double p0 = sk.pos[cu][k]; sk.obs_pos[cu][k] = p0; sk.pos[cu][k] = p0 + h * vnew; }
    }
    wave_sync();
    if (lane < NCUBE) {
      const double w0 = sk.vel[lane][3], w1 = sk.vel[lane][4], w2 = sk.vel[lane][5];
      const double wn = fsqrt(w0 * w0 + w1 * w1 + w2 * w2), ang = h * wn;
      if (wn > 1e-12) {
        const double sh = sin(0.5 * ang) / wn, dq[4] = {cos(0.5 * ang), w0 * sh, w1 * sh, w2 * sh};
        const double qo[4] = {sk.quat[lane][0], sk.quat[lane][1], sk.quat[lane][2], sk.quat[lane][3]};
        double qn[4];
        quatmul(qn, dq, qo);
        const double nn = fsqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
        for (int z = 0; z < 4; z++) sk.quat[lane][z] = qn[z] / nn;
      }
    }
  }
  wave_sync();
  return busy_rows ? 2 : 0;   // bit 0: diverged; bit 1: the robot tree carries contact rows or cubes are stacked (the caller's measure of how busy this env is)
}
#elif HRG_HAMMER
// ================================================================================================ solver (hammering)
// mj_step for the robot tree + the board with the nail on its slide joint + the hammer of CollaborativeHammeringCart (oracle: env_step_hammer): the same primal
// Newton method with exact line search on 24 DoF in three blocks of eight and up to 128 constraint rows -- TWO rows per lane (r = lane, lane + 64; slots in
// hrgym_device.h).  Contact and equality rows of J are stored dense (24 wide); the Newton system lives in registers as 3 x 3 tiles of 8 x 8 (lane (i, j) owns
// entry (i, j) of every tile) and is inverted in place by Gauss-Jordan sweeps over its 21 real pivots.  The two equalities at the hands always carry curvature,
// so every iteration inverts the full system.
struct HRow { bool active; int type, kind, dof; double sgn, D, floss, flim, aref, y, p; };   // kind 0: a row on one DoF (sgn e_dof), 1: dense row `dof` of L.Jc

PH_DYNSTEP int dynamics_step(const DevModel* __restrict__ dm_, int lane, int ncon) {
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  hrg_env_state& s = L.st;
  hrg_hammer_state& hm = L.hm;
  const double h = m.timestep;
  const int mi = lane >> 3, mj = lane & 7;
  const int nc = ncon < NCON_DYN ? ncon : NCON_DYN;
  bool ok;
  const double Mij = L.M[lane];
  const double Minv = spd_inverse1(Mij, lane, &ok), M0inv = Minv;   // entry (mi, mj) of M^-1 of the robot tree
  if (!ok) return 1;
  // ---- board + nail subtree: the board as a free body with world-frame angular velocity, the nail head a point mass at r that slides along the axis a ----
  double Mb;
  {
    const double mb = m.hm_board_mass, mn = m.hm_nail_mass;
    double r[3], ax[3], rxa[3];
    for (int a = 0; a < 3; a++) { r[a] = L.gc[HRG_HG_NAIL][a] - hm.pos[0][a]; ax[a] = L.nail_axis[a]; }
    v3cross(rxa, r, ax);
    const double rr = v3dot(r, r);
    const double rx[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0};
    double v = 0;
    if (mi < 3 && mj < 3) v = mi == mj ? mb + mn : 0.0;
    else if (mi < 3 && mj < 6) v = -mn * rx[3 * mi + (mj - 3)];
    else if (mi < 6 && mj < 3) v = mn * rx[3 * (mi - 3) + mj];
    else if (mi < 6 && mj < 6) {
      const int a = mi - 3, b = mj - 3;
      double iw = 0;
      for (int k = 0; k < 3; k++) iw += L.gR[0][3 * a + k] * m.hm_board_inertia[k] * L.gR[0][3 * b + k];
      v = iw + mn * ((a == b ? rr : 0.0) - r[a] * r[b]);
    } else if (mi == 6 && mj == 6) v = mn;
    else if (mi == 7 && mj == 7) v = 1.0;
    else if (mi == 6 && mj < 3) v = mn * ax[mj];
    else if (mj == 6 && mi < 3) v = mn * ax[mi];
    else if (mi == 6 && mj < 6) v = mn * rxa[mj - 3];
    else if (mj == 6 && mi < 6) v = mn * rxa[mi - 3];
    Mb = v;
    L.Mb[lane] = v;
    if (lane < 8) {   // applied force: gravity, the board's gyroscopic torque, the nail's velocity-product acceleration c = w x (w x r) + 2 qd w x a
      const double w[3] = {hm.vel[0][3], hm.vel[0][4], hm.vel[0][5]};
      double Lw[3], gy[3], c[3], t1[3], t2[3], rxg[3], rxc[3];
      for (int a = 0; a < 3; a++) {
        double t = 0;
        for (int b = 0; b < 3; b++) { double iw = 0; for (int k = 0; k < 3; k++) iw += L.gR[0][3 * a + k] * m.hm_board_inertia[k] * L.gR[0][3 * b + k]; t += iw * w[b]; }
        Lw[a] = t;
      }
      v3cross(gy, Lw, w);
      v3cross(t1, w, r); v3cross(t1, w, t1);
      v3cross(t2, w, ax);
      for (int a = 0; a < 3; a++) c[a] = t1[a] + 2.0 * hm.nail_v * t2[a];
      v3cross(rxg, r, m.gravity); v3cross(rxc, r, c);
      double f;
      if (lane < 3) f = (mb + mn) * m.gravity[lane] - mn * c[lane];
      else if (lane < 6) f = gy[lane - 3] + mn * rxg[lane - 3] - mn * rxc[lane - 3];
      else if (lane == 6) f = mn * v3dot(ax, m.gravity) - mn * v3dot(ax, c);
      else f = 0.0;
      L.fb[lane] = f;
    }
    if (lane < 9) {   // the hammer's world-frame rotational inertia R diag(I) R'
      const int a = lane / 3, b = lane - 3 * a;
      double iw = 0;
      for (int k = 0; k < 3; k++) iw += L.gR[1][3 * a + k] * m.hm_hammer_inertia[k] * L.gR[1][3 * b + k];
      L.Ihw[lane] = iw;
    }
  }
  wave_sync();
  const double Mbinv = spd_inverse1(Mb, lane, &ok);
  if (!ok) return 1;
  if (lane < NV) {
    double act = L.ctrl[lane];
    if (lane >= NARM) act = clampd(m.finger_kp * (act - s.qpos[lane]), m.finger_forcerange[0], m.finger_forcerange[1]);
    L.Ma0[lane] = act - m.jnt_damping[lane] * s.qvel[lane] - L.bias[lane];
    L.qacc[lane] = s.qacc_warmstart[lane];
  } else if (lane < HM_OH) {
    const int k = lane - HM_OB;
    L.Ma0[lane] = L.fb[k];
    L.qacc[lane] = k < 6 ? hm.acc_warmstart[0][k] : (k == 6 ? hm.nail_acc_warmstart : 0.0);
  } else if (lane < NVS) {   // the hammer: gravity, gyroscopic torque -(w x I w); a0 = M^-1 of that in closed form
    const int k = lane - HM_OH;
    const double w[3] = {hm.vel[1][3], hm.vel[1][4], hm.vel[1][5]};
    double Lw[3], tau[3], tl[3];
    for (int a = 0; a < 3; a++) Lw[a] = L.Ihw[3 * a] * w[0] + L.Ihw[3 * a + 1] * w[1] + L.Ihw[3 * a + 2] * w[2];
    v3cross(tau, Lw, w);
    for (int kk = 0; kk < 3; kk++) tl[kk] = (L.gR[1][kk] * tau[0] + L.gR[1][3 + kk] * tau[1] + L.gR[1][6 + kk] * tau[2]) / m.hm_hammer_inertia[kk];
    double a0v = 0, f = 0;
    if (k < 3) { a0v = m.gravity[k]; f = m.hm_hammer_mass * a0v; }
    else if (k < 6) { const int a = k - 3; a0v = L.gR[1][3 * a] * tl[0] + L.gR[1][3 * a + 1] * tl[1] + L.gR[1][3 * a + 2] * tl[2]; f = tau[a]; }
    L.a0[lane] = a0v;
    L.Ma0[lane] = f;
    L.qacc[lane] = k < 6 ? hm.acc_warmstart[1][k] : 0.0;
  }
  wave_sync();
  {
    const double x = matvec_lanes(Minv, L.Ma0, lane), xb = row8_sum(Mbinv * L.fb[mj]);
    if (mj == 0) { L.a0[mi] = x; L.a0[HM_OB + mi] = xb; }
  }
  STAMP(20);
  // ---- this lane's two constraint rows ----
  HRow R[2];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int r = lane + 64 * k;
    HRow& w = R[k];
    w.active = false; w.type = 1; w.kind = 0; w.dof = 0; w.sgn = 0; w.D = 0; w.floss = 0; w.flim = 0; w.aref = 0; w.y = 0; w.p = 0;
    bool cand = false;
    double pos = 0, margin = 0, diag = 0, vel = 0, Bd_own = -1.0;
    if (r < NV) {
      if (m.jnt_frictionloss[r] > 0) { cand = true; w.type = 0; w.floss = m.jnt_frictionloss[r]; diag = m.dof_invweight0[r]; w.dof = r; w.sgn = 1.0; vel = s.qvel[r]; }
    } else if (r == HROW_NFRIC) {   // nail_head_joint0: frictionloss with its own solreffriction (nail.xml:7)
      if (m.hm_nail_frictionloss > 0) { cand = true; w.type = 0; w.floss = m.hm_nail_frictionloss; diag = m.hm_nail_invweight; w.dof = HM_ON; w.sgn = 1.0; vel = hm.nail_v; Bd_own = m.hm_nail_fric_damping / m.solimp[1]; }
    } else if (r < HROW_NLIM) {
      const int q = r - HROW_LIM0, dof = q >> 1, side = q & 1;
      const double dist = side ? m.jnt_range[dof][1] - s.qpos[dof] : s.qpos[dof] - m.jnt_range[dof][0];
      if (dist < 0) { cand = true; pos = dist; diag = m.dof_invweight0[dof]; w.dof = dof; w.sgn = side ? -1.0 : 1.0; vel = w.sgn * s.qvel[dof]; }
    } else if (r < HROW_EQ0) {   // the slide joint's range [0, hm_nail_range]
      const int side = r - HROW_NLIM;
      const double dist = side ? m.hm_nail_range - hm.nail_q : hm.nail_q;
      if (dist < 0) { cand = true; pos = dist; diag = m.hm_nail_invweight; w.dof = HM_ON; w.sgn = side ? -1.0 : 1.0; vel = w.sgn * hm.nail_v; }
    } else if (r < HROW_CON0) {   // lh_eq: connect(lh_grip, lh_mocap) rows 0..2; rh_eq: weld(rh_grip, rh_mocap) position rows 3..5, rotation rows 6..8 (1100-1145)
      const int e = r - HROW_EQ0;
      double* Jr = L.Jc[4 * NCON_DYN + e];
      for (int i = 0; i < NVS; i++) Jr[i] = 0.0;
      w.type = 2; w.kind = 1; w.dof = 4 * NCON_DYN + e;
      cand = true;
      if (e < 6) {
        const int hd = e < 3 ? 0 : 1, a = e - 3 * hd;
        double rr[3], rxe[3];
        m3mulv(rr, L.gR[0], m.hm_anchor[hd]);
        const double ea[3] = {a == 0 ? 1.0 : 0.0, a == 1 ? 1.0 : 0.0, a == 2 ? 1.0 : 0.0};
        v3cross(rxe, rr, ea);
        pos = hm.pos[0][a] + rr[a] - hm.mocap_pos[hd][a];
        Jr[HM_OB + a] = 1.0;
        vel = hm.vel[0][a];
        for (int b = 0; b < 3; b++) { Jr[HM_OB + 3 + b] = rxe[b]; vel += rxe[b] * hm.vel[0][3 + b]; }
        diag = 1.0 / m.hm_board_mass;
      } else {
        const int a = e - 6;
        const double qm[4] = {hm.mocap_quat[1][0], hm.mocap_quat[1][1], hm.mocap_quat[1][2], hm.mocap_quat[1][3]};
        const double qi[4] = {m.hm_weld_relquat[0], -m.hm_weld_relquat[1], -m.hm_weld_relquat[2], -m.hm_weld_relquat[3]};
        const double qo[4] = {hm.quat[0][0], hm.quat[0][1], hm.quat[0][2], hm.quat[0][3]};
        double qt[4], qe[4];
        quatmul(qt, qm, qi);
        const double qc[4] = {qt[0], -qt[1], -qt[2], -qt[3]};
        quatmul(qe, qo, qc);
        if (qe[0] < 0) for (int z = 0; z < 4; z++) qe[z] = -qe[z];
        const double sn = fsqrt(qe[1] * qe[1] + qe[2] * qe[2] + qe[3] * qe[3]), ang = 2.0 * atan2(sn, qe[0]);
        pos = sn > 1e-12 ? qe[1 + a] / sn * ang : 0.0;
        Jr[HM_OB + 3 + a] = 1.0;
        vel = hm.vel[0][3 + a];
        diag = m.hm_board_invweight_rot;
      }
    } else if (r < HROW_CON0 + 4 * NCON_DYN) {
      const int q = r - HROW_CON0, c = q >> 2, d = q & 3;
      w.kind = 1; w.dof = q;
      if (c < nc) {
        const Contact& cc = L.con[c];
        const double n[3] = {cc.n[0], cc.n[1], cc.n[2]}, cp[3] = {cc.pos[0], cc.pos[1], cc.pos[2]};
        double t1[3], t2[3], dir[3];
        const double e1[3] = {1, 0, 0}, e2[3] = {0, 1, 0};
        v3cross(t1, n, fabs(n[0]) < 0.5 ? e1 : e2);
        v3scl(t1, t1, 1.0 / v3norm(t1));
        v3cross(t2, n, t1);
        const double sg = (d & 1) ? -1.0 : 1.0;
        for (int a = 0; a < 3; a++) dir[a] = n[a] + sg * m.friction_static * (d < 2 ? t1[a] : t2[a]);
        pos = cc.dist;
        margin = ((cc.g2 >= GEOM_HUMAN0 && cc.g2 < GEOM_TABLE) || (cc.g1 >= GEOM_HUMAN0 && cc.g1 < GEOM_TABLE)) ? m.contact_margin_human : 0.0;   // a human geom on either side
        const bool rb1 = cc.b1 >= 0 && cc.b1 < NV, rb2 = cc.b2 >= 0 && cc.b2 < NV;
        diag = (rb1 ? m.body_invweight0[cc.b1] : 0.0) + (rb2 ? m.body_invweight0[cc.b2] : 0.0);
        const int am1 = rb1 ? dm->anc_mask[cc.b1] : 0, am2 = rb2 ? dm->anc_mask[cc.b2] : 0;
        double nz = 0;
        double* Jr = L.Jc[q];
        for (int i = NV; i < NVS; i++) Jr[i] = 0.0;
        if (!(rb1 || rb2)) { for (int i = 0; i < NV; i++) Jr[i] = 0.0; }
        else {
#pragma unroll 1
          for (int i = 0; i < NV; i++) {
            double t[3], v[3];
            v3cross(t, L.Sw[i], cp);
            v3add(v, L.Sv[i], t);
            const double jv = v3dot(dir, v);
            double acc = 0;
            if ((am1 >> i) & 1) acc += -1.0 * jv;
            if ((am2 >> i) & 1) acc += jv;
            Jr[i] = acc;
            vel += acc * s.qvel[i];
            nz += fabs(acc);
          }
        }
#pragma unroll
        for (int side = 0; side < 2; side++) {   // free bodies: J = -+dir . (v + w x r); the nail's point moves with the board and along the slide axis
          const int body = side ? cc.b2 : cc.b1;
          if (body >= BODY_BOX) {
            const int fb = body - BODY_BOX, bi = fb == HRG_HM_HAMMER ? 1 : 0, o = fb == HRG_HM_HAMMER ? HM_OH : HM_OB;
            const double sgn = side ? 1.0 : -1.0;
            double rr[3], rxd[3];
            for (int a = 0; a < 3; a++) rr[a] = cp[a] - hm.pos[bi][a];
            v3cross(rxd, rr, dir);
            for (int a = 0; a < 3; a++) { Jr[o + a] = sgn * dir[a]; Jr[o + 3 + a] = sgn * rxd[a]; vel += sgn * (dir[a] * hm.vel[bi][a] + rxd[a] * hm.vel[bi][3 + a]); nz += fabs(dir[a]) + fabs(rxd[a]); }
            if (fb == HRG_HM_NAIL) {
              const double ja = sgn * (dir[0] * L.nail_axis[0] + dir[1] * L.nail_axis[1] + dir[2] * L.nail_axis[2]);
              Jr[HM_ON] = ja; vel += ja * hm.nail_v; nz += fabs(ja);
              diag += m.hm_nail_invweight;
            } else diag += 1.0 / (fb == HRG_HM_HAMMER ? m.hm_hammer_mass : m.hm_board_mass);
          }
        }
        diag *= 1.0 + m.friction_static * m.friction_static;
        cand = nz > 0;
      }
    }
    w.active = cand && diag > 0;
    if (w.active) {
      double imp;
      const double K = dm->sol_K, Bd = Bd_own >= 0 ? Bd_own : dm->sol_Bd;
      impedance(m, pos - margin, &imp);
      w.aref = -Bd * vel - K * imp * (pos - margin);
      w.D = 1.0 / ((1 - imp) / imp * diag);
      if (w.type == 0) w.flim = w.floss / w.D;
    }
  }
  STAMP(21);
  COUNT(19, __popcll(__ballot(R[0].active)) + __popcll(__ballot(R[1].active)));
  // does a contact join the board / nail block to another block (hammer on the board or the nail, robot on the board)?  Otherwise the Newton system splits.
  bool bc_ = false;
  if (lane < nc) { const Contact& cc = L.con[lane]; const bool b1 = cc.b1 == BODY_BOX + HRG_HM_BOARD || cc.b1 == BODY_BOX + HRG_HM_NAIL, b2 = cc.b2 == BODY_BOX + HRG_HM_BOARD || cc.b2 == BODY_BOX + HRG_HM_NAIL;
    bc_ = (b1 && cc.b2 >= 0) || (b2 && cc.b1 >= 0); }
  const bool board_coupled = __any(bc_);
  wave_sync();
  auto rowdot = [&](const HRow& w, const double* x) -> double {
    if (!w.active) return 0.0;
    if (w.kind == 0) return w.sgn * x[w.dof];
    const double* Jr = L.Jc[w.dof];
    double t = 0;
#pragma unroll
    for (int i = 0; i < NVS; i++) t += Jr[i] * x[i];
    return t;
  };
  // M x for the DoF of this lane: robot rows dense, the board + nail block dense, the hammer m 1 / R diag(I) R', pad DoF 1
  auto Mrow = [&](const double* x) -> double {
    double t = 0;
    if (lane < NV) {
#pragma unroll
      for (int j = 0; j < NV; j++) t += L.M[lane * NV + j] * x[j];
    } else if (lane < HM_OH) {
#pragma unroll
      for (int j = 0; j < 8; j++) t += L.Mb[(lane - HM_OB) * 8 + j] * x[HM_OB + j];
    } else if (lane < NVS) {
      const int k = lane - HM_OH;
      if (k < 3) t = m.hm_hammer_mass * x[lane];
      else if (k < 6) t = L.Ihw[3 * (k - 3)] * x[HM_OH + 3] + L.Ihw[3 * (k - 3) + 1] * x[HM_OH + 4] + L.Ihw[3 * (k - 3) + 2] * x[HM_OH + 5];
      else t = x[lane];
    }
    return t;
  };
  { // warm start vs unconstrained acceleration: keep the cheaper point
    double c0 = 0, c1 = 0, g_, h_;
#pragma unroll
    for (int k = 0; k < 2; k++)
      if (R[k].active) {
        double t0, t1;
        row_cost(R[k].type, R[k].D, R[k].floss, R[k].flim, rowdot(R[k], L.qacc) - R[k].aref, &t0, &g_, &h_);
        row_cost(R[k].type, R[k].D, R[k].floss, R[k].flim, rowdot(R[k], L.a0) - R[k].aref, &t1, &g_, &h_);
        c0 += t0; c1 += t1;
      }
    if (lane < NVS) L.d[lane] = L.qacc[lane] - L.a0[lane];
    wave_sync();
    double quad = 0;
    if (lane < NVS) quad = 0.5 * L.d[lane] * Mrow(L.d);
    const double cost_ws = wave_sum(quad + c0), cost_a0 = wave_sum(c1);
    wave_sync();
    if (!(cost_ws < cost_a0)) { if (lane < NVS) L.qacc[lane] = L.a0[lane]; }
    wave_sync();
  }
  STAMP(22);
#pragma unroll 1
  for (int it = 0; it < m.solver_iters; it++) {
    COUNT(16, 1);
    double gg[2], hh[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
      double cc_ = 0;
      gg[k] = 0; hh[k] = 0;
      R[k].y = rowdot(R[k], L.qacc) - R[k].aref;
      if (R[k].active) row_cost(R[k].type, R[k].D, R[k].floss, R[k].flim, R[k].y, &cc_, &gg[k], &hh[k]);
      L.rg[lane + 64 * k] = gg[k]; L.rh[lane + 64 * k] = hh[k];
    }
    wave_sync();
    double gm = 0;
    if (lane < NVS) {   // gradient of the DoF of this lane
      double t = Mrow(L.qacc) - L.Ma0[lane];
      gm = t;
      if (lane < NV) { t += L.rg[lane]; t += L.rg[HROW_LIM0 + 2 * lane]; t -= L.rg[HROW_LIM0 + 2 * lane + 1]; }
      else if (lane == HM_ON) { t += L.rg[HROW_NFRIC]; t += L.rg[HROW_NLIM]; t -= L.rg[HROW_NLIM + 1]; }
#pragma unroll 4
      for (int q = 0; q < 4 * nc; q++) t += L.Jc[q][lane] * L.rg[HROW_CON0 + q];
#pragma unroll
      for (int e = 0; e < HROW_NEQ; e++) t += L.Jc[4 * NCON_DYN + e][lane] * L.rg[HROW_EQ0 + e];
      L.g[lane] = t;
    }
    wave_sync();
    {
      const double gl = lane < NVS ? L.g[lane] : 0.0, ml = lane < NVS ? L.Ma0[lane] : 0.0;
      const double gn = wave_sum(gl * gl), sc = wave_sum(ml * ml);
      if (fsqrt(gn) <= m.solver_tol * (1.0 + fsqrt(sc))) break;
    }
    STAMP(27);
    // ---- Newton Hessian M + sum_r h_r J_r' J_r as 3 x 3 tiles in registers ----
    Tiles<3> T;
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b_ = 0; b_ < 3; b_++) T.t[a][b_] = 0.0;
    {
      double t00 = Mij, t11 = Mb, t22 = 0.0;
      if (mi == mj) { t00 += L.rh[mi]; t00 += L.rh[HROW_LIM0 + 2 * mi]; t00 += L.rh[HROW_LIM0 + 2 * mi + 1]; }
      if (mi == 6 && mj == 6) { t11 += L.rh[HROW_NFRIC]; t11 += L.rh[HROW_NLIM]; t11 += L.rh[HROW_NLIM + 1]; }
      if (mi < 3 && mj < 3) t22 = mi == mj ? m.hm_hammer_mass : 0.0;
      else if (mi >= 3 && mi < 6 && mj >= 3 && mj < 6) t22 = L.Ihw[3 * (mi - 3) + (mj - 3)];
      else if (mi == mj && mi >= 6) t22 = 1.0;
      T.t[0][0] = t00; T.t[1][1] = t11; T.t[2][2] = t22;
    }
#pragma unroll 1
    for (int q = 0; q < 4 * nc + HROW_NEQ; q++) {
      const int row = q < 4 * nc ? q : 4 * NCON_DYN + (q - 4 * nc);
      const double hq = L.rh[q < 4 * nc ? HROW_CON0 + q : HROW_EQ0 + (q - 4 * nc)];
      if (__builtin_amdgcn_readfirstlane(hq != 0.0 ? 1 : 0)) {
        const double* Jr = L.Jc[row];
        const double ji[3] = {Jr[mi], Jr[8 + mi], Jr[16 + mi]}, jj[3] = {Jr[mj], Jr[8 + mj], Jr[16 + mj]};
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
          for (int b_ = 0; b_ < 3; b_++) T.t[a][b_] += hq * ji[a] * jj[b_];
      }
    }
    STAMP(23);
    COUNT(18, 1);
    {
      bool good = true;
      if (board_coupled) {
        tiles_pivots<3, 0, 0>(T, mi, mj, 8, good);
        tiles_pivots<3, 1, 0>(T, mi, mj, 7, good);
        tiles_pivots<3, 2, 0>(T, mi, mj, 6, good);
      } else {   // no contact row joins the board / nail to the robot or the hammer: the board block on its own, robot + hammer as 2 x 2 tiles
        Tiles<2> T2;
        Tiles<1> T1;
        T2.t[0][0] = T.t[0][0]; T2.t[0][1] = T.t[0][2]; T2.t[1][0] = T.t[2][0]; T2.t[1][1] = T.t[2][2];
        T1.t[0][0] = T.t[1][1];
        tiles_pivots<2, 0, 0>(T2, mi, mj, 8, good);
        tiles_pivots<2, 1, 0>(T2, mi, mj, 6, good);
        tiles_pivots<1, 0, 0>(T1, mi, mj, 7, good);
        T.t[0][0] = T2.t[0][0]; T.t[0][2] = T2.t[0][1]; T.t[2][0] = T2.t[1][0]; T.t[2][2] = T2.t[1][1];
        T.t[1][1] = T1.t[0][0];
        T.t[0][1] = 0.0; T.t[1][0] = 0.0; T.t[1][2] = 0.0; T.t[2][1] = 0.0;
      }
      if (!good) break;
    }
    STAMP(24);
    {   // direction d = -H^-1 g, one tile row at a time
      const double gb[3] = {L.g[mj], L.g[8 + mj], L.g[16 + mj]};
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const double x = -row8_sum(T.t[a][0] * gb[0] + T.t[a][1] * gb[1] + T.t[a][2] * gb[2]);
        if (mj == 0) L.d[8 * a + mi] = x;
      }
    }
    wave_sync();
#pragma unroll
    for (int k = 0; k < 2; k++) R[k].p = rowdot(R[k], L.d);
    double dd = 0, Mdi = 0;
    if (lane < NVS) { dd = L.d[lane]; Mdi = Mrow(L.d); }
    STAMP(25);
    const double dMd = wave_sum(dd * Mdi), gd0 = wave_sum(dd * gm);
    double al = 1.0, lo = 0, hi = -1;
    const double d1_0 = gd0 + wave_sum(gg[0] * R[0].p + gg[1] * R[1].p);
    const double noise = fabs(gd0) + dMd + wave_sum(fabs(gg[0] * R[0].p) + fabs(gg[1] * R[1].p));   // magnitude of the terms phi' is summed from
#pragma unroll 1
    for (int ls = 0; ls < 40; ls++) {
      COUNT(17, 1);
      double sg_ = 0, sh_ = 0;
#pragma unroll
      for (int k = 0; k < 2; k++)
        if (R[k].active) {
          double c2, g2, h2;
          row_cost(R[k].type, R[k].D, R[k].floss, R[k].flim, R[k].y + al * R[k].p, &c2, &g2, &h2);
          sg_ += g2 * R[k].p; sh_ += h2 * R[k].p * R[k].p;
        }
      const double d1 = gd0 + al * dMd + wave_sum(sg_);
      const double d2 = dMd + wave_sum(sh_);
      if (fabs(d1) <= 1e-10 * fabs(d1_0) || fabs(d1) <= 1e-14 * noise) break;   // converged, or down at the rounding noise of the sum's own terms
      if (d1 < 0) lo = al; else hi = al;
      double nx = al - d1 / d2;
      if (hi < 0) { if (!(nx > lo)) nx = 2 * al; }
      else if (!(nx > lo && nx < hi)) nx = 0.5 * (lo + hi);
      al = nx;
    }
    STAMP(26);
    if (lane < NVS) L.qacc[lane] += al * dd;
    wave_sync();
    bool moved = false;
#pragma unroll
    for (int k = 0; k < 2; k++)
      if (R[k].active) moved = moved || row_zone(R[k].type, R[k].flim, R[k].y) != row_zone(R[k].type, R[k].flim, R[k].y + R[k].p);
    if (al == 1.0 && !__any(moved)) break;
  }
  STAMP(28);
  // ---- MuJoCo's noslip post-pass (mj_solNoSlip [UPSTREAM]; collaborative_hammering_cartesian_env.py:1161: noslip_iterations = 20), restated as in oracle/hrg_oracle.c
  // noslip(): projected Gauss-Seidel sweeps over the friction dimensions only -- the nine friction-loss rows, then each contact's two pairs of opposing pyramid edges
  // (a pair keeps the sum of its two forces and moves along (1, -1)) -- without the constraint regularisation, every other force held.  The sweeps are sequential by
  // definition and there are up to 20 of them over up to 49 items, so an item must be cheap: the pass runs in the space of its own unknowns z (item j = lane j: a
  // friction force or a pair's half difference y).  Once per substep the items' Gram matrix G = U M^-1 U' (U = unit rows / J0 - J1 of a pair) goes to LDS (packed
  // lower triangle) and every lane takes its item's residual res_j = U_j a - aref_j; an item's update is then  z_i <- clamp(z_i - res_i / G_ii),  res_j += G_ij dz
  // for all lanes -- broadcasts of lane i's registers and one LDS read, no reduction, no barrier.  The changed forces go back into a = L.qacc at the end.
  if (m.noslip_iterations > 0) {
    constexpr int NI0 = HROW_NFRIC + 1;                    // items 0..8: friction loss of the robot tree's joints and of the nail's slide joint
    const int npair = 2 * nc, nitem = NI0 + npair;         // items 9 + p: pair p = rows 2p, 2p + 1 of the contact block
    auto tri = [](int i, int j) -> int { return i * (i + 1) / 2 + j; };   // i >= j
    double Mhinv = 0.0;   // entry (mi, mj) of the hammer block's inverse: 1/m | R diag(1/I) R' | unit pads
    if (mi < 3 && mj < 3) Mhinv = mi == mj ? 1.0 / m.hm_hammer_mass : 0.0;
    else if (mi >= 3 && mi < 6 && mj >= 3 && mj < 6) { for (int k = 0; k < 3; k++) Mhinv += L.gR[1][3 * (mi - 3) + k] * L.gR[1][3 * (mj - 3) + k] / m.hm_hammer_inertia[k]; }
    else if (mi == mj && mi >= 6) Mhinv = 1.0;
    double fsum = 0, f_own = 0;
#pragma unroll
    for (int k = 0; k < 2; k++) {   // forces of the primal solution: f = -s'(J a - aref)
      const int r = lane + 64 * k;
      double c_, g_ = 0, h_;
      if (R[k].active) row_cost(R[k].type, R[k].D, R[k].floss, R[k].flim, rowdot(R[k], L.qacc) - R[k].aref, &c_, &g_, &h_);
      L.rg[r] = -g_;
      if (k == 0) f_own = -g_;
      if (R[k].active) fsum += 0.5 * g_ * g_ / R[k].D;
    }
    const uint64_t act0 = __ballot(R[0].active), act1 = __ballot(R[1].active);
    auto row_active = [&](int r) -> bool { return r < 64 ? ((act0 >> r) & 1ull) != 0 : ((act1 >> (r - 64)) & 1ull) != 0; };
#pragma unroll
    for (int k = 0; k < 2; k++) { const int r = lane + 64 * k; if (r >= HROW_CON0 && r < HROW_CON0 + 4 * NCON_DYN) L.Jc[r - HROW_CON0][NVS] = R[k].aref; }
    const double imp0 = wave_sum(fsum);   // the first sweep also counts the removed regularisation cost, sum 1/2 R f^2
    wave_sync();
    // row 2p + 1 of a pair <- J0 - J1 (and aref0 - aref1 in the pad column): the rows themselves are not needed again
    if (lane <= NVS)
      for (int p = 0; p < npair; p++) L.Jc[2 * p + 1][lane] = L.Jc[2 * p][lane] - L.Jc[2 * p + 1][lane];
    wave_sync();
    // ---- this lane's item ----
    bool item = false;
    double z = 0, bnd = 0, res = 0;
    if (lane < NI0) {
      item = R[0].active;
      const int dof = lane == HROW_NFRIC ? HM_ON : lane;
      z = f_own; bnd = R[0].floss; res = L.qacc[dof] - R[0].aref;
    } else if (lane < nitem) {
      const int p = lane - NI0;
      item = row_active(HROW_CON0 + 2 * p) && row_active(HROW_CON0 + 2 * p + 1);
      const double f0 = L.rg[HROW_CON0 + 2 * p], f1 = L.rg[HROW_CON0 + 2 * p + 1];
      const double* dJ = L.Jc[2 * p + 1];
      z = 0.5 * (f0 - f1); bnd = 0.5 * (f0 + f1);
      if (bnd < 0) bnd = 0;
      double t = -dJ[NVS];
      for (int k = 0; k < NVS; k++) t += dJ[k] * L.qacc[k];
      res = t;
    }
    const double z0 = z;
    const uint64_t imask = __ballot(item);
    // ---- Gram matrix: friction x friction from the block inverses in registers ... ----
    if (mj <= mi) L.nsG[tri(mi, mj)] = Minv;                                   // items 0..7 = robot DoF 0..7
    if (lane < NI0) L.nsG[tri(HROW_NFRIC, lane)] = lane == HROW_NFRIC ? lane_value<6 * 9>(Mbinv) : 0.0;   // the nail's row: another block of M
    if (lane == HROW_NFRIC) L.nsG[tri(HROW_NFRIC, HROW_NFRIC)] = lane_value<6 * 9>(Mbinv);
    // ... and one row per pair: w = M^-1 (J0 - J1)' through L.d, then lanes = items take their entry
#pragma unroll 1
    for (int p = 0; p < npair; p++) {
      if (!((imask >> (NI0 + p)) & 1ull)) continue;
      const double* dJ = L.Jc[2 * p + 1];
      const double t0 = row8_sum(Minv * dJ[mj]), t1 = row8_sum(Mbinv * dJ[8 + mj]), t2 = row8_sum(Mhinv * dJ[16 + mj]);
      wave_sync();
      if (mj == 0) { L.d[mi] = t0; L.d[8 + mi] = t1; L.d[16 + mi] = t2; }
      wave_sync();
      if (lane <= NI0 + p) {
        double g_;
        if (lane < NI0) g_ = L.d[lane == HROW_NFRIC ? HM_ON : lane];
        else {
          const double* dQ = L.Jc[2 * (lane - NI0) + 1];
          g_ = 0;
          for (int k = 0; k < NVS; k++) g_ += dQ[k] * L.d[k];
        }
        L.nsG[tri(NI0 + p, lane)] = g_;
      }
    }
    wave_sync();
    // per item: curvature, its reciprocal and the bound stay in the item's own lane.  A pair without curvature (both edges then go to the mean) is a zero bound
    // with a zero step.  An update of item i:  every lane works out the step ITS item would take now, z' = clamp(z - res / G_jj), from its own registers; the step
    // of lane i is broadcast (two v_readlane) and every lane takes res += G_ij dl; lane i keeps z' and its share of the cost decrease.  The sweeps repeat over the
    // same items, so every lane first takes its entries of the Gram columns of the first NS_REG active items into registers (one wave per SIMD: up to 512 VGPRs).
    // An update is then 12 instructions without a memory access; broadcasting the item's five values and reading the column from LDS it was 34 and ~ 450 cycles of
    // a strictly sequential chain, ~ 260 times per substep (40 % of the kernel).  Items beyond NS_REG (more than seven contacts: 0.5 % of the substeps) read LDS.
    const double Gjj = lane < nitem ? L.nsG[tri(lane, lane)] : 0.0;
    const bool flat = lane >= NI0 && !(Gjj >= 1e-15);
    const double inv_own = flat ? 0.0 : 1.0 / (Gjj > 1e-15 ? Gjj : 1e-15), bnd_own = flat ? 0.0 : bnd, nbnd_own = -bnd_own, hG_own = 0.5 * Gjj;
    const int tri_lane = lane * (lane + 1) / 2;
    constexpr int NS_REG = 24;
    double gcol[NS_REG];
    uint64_t rest = imask;   // active items beyond the register columns
    {
      uint64_t mm = imask;
#pragma unroll
      for (int k = 0; k < NS_REG; k++) {
        gcol[k] = 0.0;
        if (mm) {
          const int i = __ffsll((long long)mm) - 1;
          mm &= mm - 1;
          if (item) gcol[k] = L.nsG[lane <= i ? i * (i + 1) / 2 + lane : tri_lane + i];
        }
      }
      rest = mm;
    }
    STAMP(31);
    double imp_own = 0.0;   // this lane's item: its cost decrease in the running sweep
    auto update = [&](int i, double gij) {
      const double zn = fmin(fmax(z - res * inv_own, nbnd_own), bnd_own);
      const double dl_own = zn - z;
      const double dl = lane_value_dyn(dl_own, i);
      if (lane == i) { imp_own -= dl_own * (res + hG_own * dl_own); z = zn; }
      res += gij * dl;
    };
#pragma unroll 1
    for (int it = 0; it < m.noslip_iterations; it++) {
      COUNT(30, 1);   // noslip sweeps
      imp_own = 0.0;
      uint64_t mm = imask;
#pragma unroll
      for (int k = 0; k < NS_REG; k++)
        if (mm) {
          const int i = __ffsll((long long)mm) - 1;
          mm &= mm - 1;
          update(i, gcol[k]);
        }
#pragma unroll 1
      for (uint64_t mr = rest; mr;) {
        const int i = __ffsll((long long)mr) - 1;
        mr &= mr - 1;
        update(i, item ? L.nsG[lane <= i ? i * (i + 1) / 2 + lane : tri_lane + i] : 0.0);
      }
      const double imp = (it == 0 ? imp0 : 0.0) + wave_sum(imp_own);
      if (imp * m.noslip_scale < m.noslip_tolerance) break;
    }
    // ---- the changed forces back into the acceleration: a += M^-1 U' dz ----
    const double dz = item ? z - z0 : 0.0;
    double u = 0;
    if (lane < NV) u = dz;                                            // item r = friction row of robot DoF r = lane r
    const double dz_nail = lane_value<HROW_NFRIC>(dz);
    if (lane == HM_ON) u = dz_nail;
    for (uint64_t mm = imask >> NI0; mm;) {
      const int p = __ffsll((long long)mm) - 1;
      mm &= mm - 1;
      const double dzp = lane_value_dyn(dz, NI0 + p);
      if (lane < NVS) u += L.Jc[2 * p + 1][lane] * dzp;
    }
    wave_sync();
    if (lane < NVS) L.d[lane] = u;
    wave_sync();
    {
      const double t0 = row8_sum(Minv * L.d[mj]), t1 = row8_sum(Mbinv * L.d[8 + mj]), t2 = row8_sum(Mhinv * L.d[16 + mj]);
      if (mj == 0) { L.qacc[mi] += t0; L.qacc[8 + mi] += t1; L.qacc[16 + mi] += t2; }
    }
    wave_sync();
  }
  STAMP(29);
  const bool badacc = lane < NVS && !(fabs(L.qacc[lane]) < 1e10);
  if (__any(badacc)) return 1;
  // mj_Euler with implicit joint damping for the robot tree
  euler_robot_tree(dm, lane, Mij, M0inv);
  hammer_obs_pos(dm_, lane);   // body_xpos of the forward pass inside mj_step (pre-integration)
  wave_sync();
  { // the free joints: lanes 8..13 board, 14 the nail's slide joint, 16..21 hammer; quaternions by lanes 0, 1
    if ((lane >= HM_OB && lane < HM_ON) || (lane >= HM_OH && lane < HM_OH + 6)) {
      const int fb = lane >= HM_OH ? 1 : 0, k = lane - (fb ? HM_OH : HM_OB);
      const double acc = L.qacc[lane];
      hm.acc_warmstart[fb][k] = acc;
      const double vnew = hm.vel[fb][k] + h * acc;
      hm.vel[fb][k] = vnew;
      if (k < 3) hm.pos[fb][k] = hm.pos[fb][k] + h * vnew;
    } else if (lane == HM_ON) {
      const double acc = L.qacc[lane];
      hm.nail_acc_warmstart = acc;
      const double vnew = hm.nail_v + h * acc;
      hm.nail_v = vnew;
      hm.nail_q = hm.nail_q + h * vnew;
    }
    wave_sync();
    if (lane < 2) {
      const double w0 = hm.vel[lane][3], w1 = hm.vel[lane][4], w2 = hm.vel[lane][5];
      const double wn = fsqrt(w0 * w0 + w1 * w1 + w2 * w2), ang = h * wn;
      if (wn > 1e-12) {
        const double sh = sin(0.5 * ang) / wn, dq[4] = {cos(0.5 * ang), w0 * sh, w1 * sh, w2 * sh};
        const double qo[4] = {hm.quat[lane][0], hm.quat[lane][1], hm.quat[lane][2], hm.quat[lane][3]};
        double qn[4];
        quatmul(qn, dq, qo);
        const double nn = fsqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
        for (int z = 0; z < 4; z++) hm.quat[lane][z] = qn[z] / nn;
      }
    }
  }
  wave_sync();
  return nc > 0 ? 2 : 0;   // bit 0: diverged; bit 1: contacts enter the solve (the caller's measure of how busy this env is)
}
#else
PH_DYNSTEP int dynamics_step(const DevModel* __restrict__ dm_, int lane, int ncon) {
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  hrg_env_state& s = L.st;
  const double h = m.timestep;
  const int mi = lane >> 3, mj = lane & 7;
  // ---- factor M across the wave; unconstrained acceleration a0 = M^-1 (actuation + passive - bias) ----
  bool ok;
  const double Mij = L.M[lane];
  double Minv = spd_inverse1(Mij, lane, &ok);   // entry (mi, mj) of M^-1; later the inverse of the Newton Hessian's robot block
  const double M0inv = Minv;                    // M^-1 itself: mj_Euler's implicit damping at the end of the substep
  if (!ok) return 1;
  if (lane < NV) {
    double act = L.ctrl[lane];
    if (lane >= NARM) act = clampd(m.finger_kp * (act - s.qpos[lane]), m.finger_forcerange[0], m.finger_forcerange[1]);
    L.Ma0[lane] = act - m.jnt_damping[lane] * s.qvel[lane] - L.bias[lane];  // M a0 (exactly the solve's right-hand side)
    L.qacc[lane] = s.qacc_warmstart[lane];
  }
#if HRG_BOX
  // free box, world-frame angular velocity: M = blockdiag(m 1, R diag(I) R'), bias torque w x (R diag(I) R' w), gravity.  The rotational
  // inertia is split into mean 1 + R diag(I - mean) R': a cube keeps an exactly diagonal M and no gyroscopic term
  hrg_box_state& bx = L.bx;
  const bool aniso = !(m.box_inertia[0] == m.box_inertia[1] && m.box_inertia[1] == m.box_inertia[2]);   // model constant: wave-uniform
  if (lane < 9) {
    const int i = lane / 3, j = lane - 3 * i;
    double t = 0;
    if (aniso) for (int k = 0; k < 3; k++) t += L.bR[3 * i + k] * (m.box_inertia[k] - m.box_inertia_mean) * L.bR[3 * j + k];
    L.bMr[lane] = (i == j ? m.box_inertia_mean : 0.0) + t;
  }
  wave_sync();
  if (lane >= NV && lane < NVT) {
    const int a = lane - NV;
    double a0v = a < 3 ? m.gravity[a] : 0.0, Ma0v = a < 3 ? m.box_mass * a0v : 0.0;
    if (aniso && a >= 3) {   // each rotational lane runs the short chain itself: a0 = R diag(1/I) R' ((R diag(I - mean) R' w) x w)
      const double w[3] = {bx.vel[3], bx.vel[4], bx.vel[5]};
      double Ld[3], Lw[3], tau[3], tl[3], ar[3];
      for (int k = 0; k < 3; k++) Ld[k] = (m.box_inertia[k] - m.box_inertia_mean) * (L.bR[k] * w[0] + L.bR[3 + k] * w[1] + L.bR[6 + k] * w[2]);
      for (int k = 0; k < 3; k++) Lw[k] = L.bR[3 * k] * Ld[0] + L.bR[3 * k + 1] * Ld[1] + L.bR[3 * k + 2] * Ld[2];
      v3cross(tau, Lw, w);   // -(w x L)
      for (int k = 0; k < 3; k++) tl[k] = (L.bR[k] * tau[0] + L.bR[3 + k] * tau[1] + L.bR[6 + k] * tau[2]) / m.box_inertia[k];
      for (int k = 0; k < 3; k++) ar[k] = L.bR[3 * k] * tl[0] + L.bR[3 * k + 1] * tl[1] + L.bR[3 * k + 2] * tl[2];
      a0v = a == 3 ? ar[0] : (a == 4 ? ar[1] : ar[2]);   // (selects: an index would put ar[] into scratch memory)
      Ma0v = L.bMr[3 * (a - 3)] * ar[0] + L.bMr[3 * (a - 3) + 1] * ar[1] + L.bMr[3 * (a - 3) + 2] * ar[2];
    }
    L.a0[lane] = a0v;
    L.Ma0[lane] = Ma0v;
    L.qacc[lane] = bx.acc_warmstart[a];
  }
#endif
  wave_sync();
  {
    const double x = matvec_lanes(Minv, L.Ma0, lane);   // a0 = M^-1 (M a0): lanes (i, 0) publish row i
    if (mj == 0) L.a0[mi] = x;
  }
  STAMP(20);
  // ---- this lane's constraint row (fixed slot) ----
  const int r = lane;
  int type = 1, rdof = 0;
  bool cand = false;
  [[maybe_unused]] bool rpart = true, bpart = false;  // row acts on the robot tree / on the cube
  double rsgn = 0, pos = 0, margin = 0, floss = 0, diag = 0, vel = 0;
  if (r < NV) {
    if (m.jnt_frictionloss[r] > 0) {
      cand = true; type = 0; floss = m.jnt_frictionloss[r]; diag = m.dof_invweight0[r];
      rdof = r; rsgn = 1.0; vel = s.qvel[r];
    }
  } else if (r < ROW_CON0) {
    const int k = r - NV, dof = k >> 1, side = k & 1;
    const double dist = side ? m.jnt_range[dof][1] - s.qpos[dof] : s.qpos[dof] - m.jnt_range[dof][0];
    if (dist < 0) {
      cand = true; pos = dist; diag = m.dof_invweight0[dof];
      rdof = dof; rsgn = side ? -1.0 : 1.0; vel = rsgn * s.qvel[dof];
    }
  } else if (r < ROW_WELD0) {
    const int c = (r - ROW_CON0) >> 2, d = (r - ROW_CON0) & 3;
    if (c < ncon && c < NCON_DYN) {
      const Contact& cc = L.con[c];
      const double n[3] = {cc.n[0], cc.n[1], cc.n[2]}, cp[3] = {cc.pos[0], cc.pos[1], cc.pos[2]};
      double t1[3], t2[3], dir[3];
      const double e1[3] = {1, 0, 0}, e2[3] = {0, 1, 0};
      v3cross(t1, n, fabs(n[0]) < 0.5 ? e1 : e2);
      v3scl(t1, t1, 1.0 / v3norm(t1));
      v3cross(t2, n, t1);
      const double sg = (d & 1) ? -1.0 : 1.0;
      for (int a = 0; a < 3; a++) dir[a] = n[a] + sg * m.friction_static * (d < 2 ? t1[a] : t2[a]);
      pos = cc.dist;
      margin = ((cc.g2 >= GEOM_HUMAN0 && cc.g2 < GEOM_TABLE) || (cc.g1 >= GEOM_HUMAN0 && cc.g1 < GEOM_TABLE)) ? m.contact_margin_human : 0.0;   // a human geom on either side
      const bool rb1 = cc.b1 >= 0 && cc.b1 < NV, rb2 = cc.b2 >= 0 && cc.b2 < NV;
      diag = (rb1 ? m.body_invweight0[cc.b1] : 0.0) + (rb2 ? m.body_invweight0[cc.b2] : 0.0);
      const int am1 = rb1 ? dm->anc_mask[cc.b1] : 0, am2 = rb2 ? dm->anc_mask[cc.b2] : 0;
      double nz = 0;
#if HRG_BOX
      if (!(rb1 || rb2)) { // table / floor - cube: no robot part
        for (int i = 0; i < NV; i++) L.Jc[r - ROW_CON0][i] = 0.0;
      } else
#endif
#pragma unroll 1
      for (int i = 0; i < NV; i++) {
        double t[3], v[3];
        v3cross(t, L.Sw[i], cp);
        v3add(v, L.Sv[i], t);
        const double jv = v3dot(dir, v);
        double acc = 0;
        if ((am1 >> i) & 1) acc += -1.0 * jv;
        if ((am2 >> i) & 1) acc += jv;
        L.Jc[r - ROW_CON0][i] = acc;
        vel += acc * s.qvel[i];
        nz += fabs(acc);
      }
      rpart = nz > 0;
#if HRG_BOX
      bpart = cc.b2 == BODY_BOX;
      if (cc.b2 == BODY_BOX) { // the cube is always geom 2: J = dir . (v + w x r); body_invweight0 of a free body = 1/m
        double rr[3], rxd[3];
        for (int a = 0; a < 3; a++) rr[a] = cp[a] - bx.pos[a];
        v3cross(rxd, rr, dir);
        for (int a = 0; a < 3; a++) { L.Jc[r - ROW_CON0][NV + a] = dir[a]; vel += dir[a] * bx.vel[a]; nz += fabs(dir[a]); }
        for (int a = 0; a < 3; a++) { L.Jc[r - ROW_CON0][NV + 3 + a] = rxd[a]; vel += rxd[a] * bx.vel[3 + a]; nz += fabs(rxd[a]); }
        diag += 1.0 / m.box_mass;
      } else {
        for (int a = 0; a < HRG_NBOXV; a++) L.Jc[r - ROW_CON0][NV + a] = 0.0;
      }
#endif
      diag *= 1.0 + m.friction_static * m.friction_static;
      cand = nz > 0;
    }
  }
#if HRG_LIFT
  else if (r < NROW && bx.weld_active) { // two connect equalities (collaborative_lifting_cartesian_env.py:924-958): the board's grip points follow the hand mocap
                                         // bodies; residual = p_board + R anchor - p_mocap, velocity of the point = v + w x r.  General rows: stored like contact rows
    const int hd = (r - ROW_WELD0) / 3, a = (r - ROW_WELD0) - 3 * hd;
    double rr[3], rxe[3];
    const double anc[3] = {m.lift_anchor[hd][0], m.lift_anchor[hd][1], m.lift_anchor[hd][2]};
    m3mulv(rr, L.bR, anc);
    const double ea[3] = {a == 0 ? 1.0 : 0.0, a == 1 ? 1.0 : 0.0, a == 2 ? 1.0 : 0.0};
    v3cross(rxe, rr, ea);   // (w x r) . e_a = w . (r x e_a)
    pos = bx.pos[a] + rr[a] - (hd ? bx.weld_off[a] : bx.mocap_pos[a]);
    for (int i = 0; i < NV; i++) L.Jc[r - ROW_CON0][i] = 0.0;
    for (int b = 0; b < 3; b++) { L.Jc[r - ROW_CON0][NV + b] = ea[b]; L.Jc[r - ROW_CON0][NV + 3 + b] = rxe[b]; }
    vel = bx.vel[a] + rxe[0] * bx.vel[3] + rxe[1] * bx.vel[4] + rxe[2] * bx.vel[5];
    cand = true; type = 2; rpart = false; bpart = true;
    diag = 1.0 / m.box_mass;
  }
#endif
#if HRG_HANDOVER
  else if (r < NROW && bx.weld_active) { // weld of the object frame onto the hand mocap frame (human_robot_handover_cartesian_env.py:870-903): residual =
                                         // [p_obj - p_mocap; rotation vector of q_obj q_mocap^-1]; unit rows on the cube's own DoF, the mocap body has no velocity
    const int a = r - ROW_WELD0;
    const double qm[4] = {bx.mocap_quat[0], bx.mocap_quat[1], bx.mocap_quat[2], bx.mocap_quat[3]};
    if (a < 3) { // where the weld wants the object: mocap frame o relative pose (identity for the human's own pickup)
      double Rm[9], tp[3];
      const double wo[3] = {bx.weld_off[0], bx.weld_off[1], bx.weld_off[2]};
      quat2mat(Rm, qm);
      m3mulv(tp, Rm, wo);
      pos = bx.pos[a] - (tp[a] + bx.mocap_pos[a]);
    } else {
      const double wr[4] = {bx.weld_rel[0], bx.weld_rel[1], bx.weld_rel[2], bx.weld_rel[3]}, qo[4] = {bx.quat[0], bx.quat[1], bx.quat[2], bx.quat[3]};
      double qt[4], qe[4];
      quatmul(qt, qm, wr);
      const double qc[4] = {qt[0], -qt[1], -qt[2], -qt[3]};
      quatmul(qe, qo, qc);
      if (qe[0] < 0) for (int k = 0; k < 4; k++) qe[k] = -qe[k];
      const double sn = fsqrt(qe[1] * qe[1] + qe[2] * qe[2] + qe[3] * qe[3]), ang = 2.0 * atan2(sn, qe[0]);
      pos = sn > 1e-12 ? qe[1 + (a - 3)] / sn * ang : 0.0;
    }
    cand = true; type = 2; rpart = false;
    diag = a < 3 ? 1.0 / m.box_mass : m.box_invweight_rot;
    rdof = NV + a; rsgn = 1.0; vel = bx.vel[a];
  }
#endif
  const bool active = cand && diag > 0;
#if HRG_LIFT
  const bool is_con = r >= ROW_CON0 && r < NROW;   // the connect rows are general rows on the board's DoF: handled like contact rows
#else
  const bool is_con = r >= ROW_CON0 && r < ROW_WELD0;
#endif
  double aref = 0, D = 0, flim = 0;
  if (active) {
    double imp;
    const double K = dm->sol_K, Bd = dm->sol_Bd;
    impedance(m, pos - margin, &imp);
    aref = -Bd * vel - K * imp * (pos - margin);
    D = 1.0 / ((1 - imp) / imp * diag);
    if (type == 0) flim = floss / D;
  }
  const uint64_t mask = __ballot(active);
#if HRG_HANDOVER
  const uint64_t cmask = (mask & ((1ull << ROW_WELD0) - 1)) >> ROW_CON0;  // active contact rows (the weld's unit rows are added on the diagonal, not through Jc)
#else
  const uint64_t cmask = mask >> ROW_CON0;  // active contact rows (lifting: and the connect rows behind them)
#endif
#if HRG_BOX
  // contact rows by the block of the Hessian they touch.  Without a robot-cube contact the 14-DoF system is block diagonal
  // (8x8 robot tree, 6x6 cube) and both blocks are factored in registers; a coupled system takes the 14x14 LDS path.
  const uint64_t rmask = __ballot(active && is_con && rpart) >> ROW_CON0, bmask = __ballot(active && is_con && bpart) >> ROW_CON0;
  const bool coupled = (rmask & bmask) != 0;
#else
  const uint64_t rmask = cmask;
#endif
  STAMP(21);
  COUNT(19, __popcll(mask));  // active rows
  wave_sync();
  // the contact rows in the sums below: ALL rows of the substep's contacts, in slot order, unrolled by contact -- an inactive row carries g = h = 0 over a finite
  // Jacobian row, so it adds an exact zero and the sums are those over the active rows (the oracle's), but the LDS reads of a contact's four rows are
  // independent of each other instead of hanging on a find-first-set chain through the row mask
  const int nq = 4 * (ncon < NCON_DYN ? ncon : NCON_DYN);
#if HRG_LIFT
  const uint64_t wmask = bmask >> (ROW_WELD0 - ROW_CON0);   // the connect rows behind the contact rows (their Jacobian rows exist only while the grip holds)
#endif
  // J_r . x for a wave-shared vector x (LDS)
  auto rowdot = [&](const double* x) -> double {
    if (!active) return 0.0;
    if (!is_con) return rsgn * x[rdof];
    const double* Jr = L.Jc[r - ROW_CON0];
    double t = 0;
#pragma unroll
    for (int i = 0; i < NVS; i++) t += Jr[i] * x[i];
    return t;
  };
  if (mask == 0) {
    if (lane < NVS) L.qacc[lane] = L.a0[lane];
    wave_sync();
  } else {
    { // warm start vs unconstrained acceleration: keep the cheaper point
      const double ei = L.qacc[mi] - L.a0[mi], ej = L.qacc[mj] - L.a0[mj];
      double c0 = 0, c1 = 0, g_, h_;
      if (active) { row_cost(type, D, floss, flim, rowdot(L.qacc) - aref, &c0, &g_, &h_); row_cost(type, D, floss, flim, rowdot(L.a0) - aref, &c1, &g_, &h_); }
      double quad = 0.5 * Mij * ei * ej;
#if HRG_BOX
      if (lane < HRG_NBOXV) {
        const double eb = L.qacc[NV + lane] - L.a0[NV + lane];
        double Me = m.box_mass * eb;
        if (lane >= 3) { Me = 0; for (int b = 0; b < 3; b++) Me += L.bMr[3 * (lane - 3) + b] * (L.qacc[NV + 3 + b] - L.a0[NV + 3 + b]); }
        quad += 0.5 * eb * Me;
      }
#endif
      const double cost_ws = wave_sum(quad + c0), cost_a0 = wave_sum(c1);
      wave_sync();
      if (!(cost_ws < cost_a0)) { if (lane < NVS) L.qacc[lane] = L.a0[lane]; }
      wave_sync();
    }
    STAMP(22);
    bool h_is_m = true;
#pragma unroll 1
    for (int it = 0; it < m.solver_iters; it++) {
      COUNT(16, 1);  // Newton iterations
      const double y = rowdot(L.qacc) - aref;
      double cc = 0, gg = 0, hh = 0;
      if (active) row_cost(type, D, floss, flim, y, &cc, &gg, &hh);
      if (r < NROW) { L.rg[r] = gg; L.rh[r] = hh; }
      wave_sync();
      double gm = 0;
      if (lane < NV) { // gradient, rows accumulated in slot order (the oracle's row order)
        double t = -L.Ma0[lane];
#pragma unroll
        for (int j = 0; j < NV; j++) t += L.M[lane * NV + j] * L.qacc[j];
        gm = t;
        t += L.rg[lane];
        t += L.rg[NV + 2 * lane];
        t -= L.rg[NV + 2 * lane + 1];
#pragma unroll 4
        for (int q = 0; q < nq; q++) t += L.Jc[q][lane] * L.rg[ROW_CON0 + q];
        L.g[lane] = t;
      }
#if HRG_BOX
      else if (lane < NVT) {
        double t = -L.Ma0[lane] + box_Mrow(m.box_mass, lane - NV, &L.qacc[NV]);
        gm = t;
#pragma unroll 4
        for (int q = 0; q < nq; q++) t += L.Jc[q][lane] * L.rg[ROW_CON0 + q];
#if HRG_LIFT
        for (uint64_t mm = wmask; mm;) { const int q = __ffsll((long long)mm) - 1 + ROW_WELD0 - ROW_CON0; mm &= mm - 1; t += L.Jc[q][lane] * L.rg[ROW_CON0 + q]; }
#endif
#if HRG_HANDOVER
        t += L.rg[ROW_WELD0 + lane - NV];   // weld row of this DoF (0 when the weld is off)
#endif
        L.g[lane] = t;
      }
      wave_sync();
      double gn = 0, sc = 0;
#pragma unroll
      for (int i = 0; i < NVT; i++) { gn += L.g[i] * L.g[i]; sc += L.Ma0[i] * L.Ma0[i]; }
      if (fsqrt(gn) <= m.solver_tol * (1.0 + fsqrt(sc))) break;
      if (!coupled) {
        // robot block (as in the ReachHuman solver) and cube block (6x6 padded to 8x8 with a unit diagonal), lanes = (mi, mj)
        double hval = Mij;
        if (mi == mj) { hval += L.rh[mi]; hval += L.rh[NV + 2 * mi]; hval += L.rh[NV + 2 * mi + 1]; }
#pragma unroll 4
        for (int q = 0; q < nq; q++) hval += L.rh[ROW_CON0 + q] * L.Jc[q][mi] * L.Jc[q][mj];
        double sval = mi == mj ? (mi < 3 ? m.box_mass : 1.0) : 0.0;
        if (mi >= 3 && mi < HRG_NBOXV && mj >= 3 && mj < HRG_NBOXV) sval = L.bMr[3 * (mi - 3) + (mj - 3)];
#if HRG_HANDOVER
        if (mi == mj && mi < HRG_NBOXV) sval += L.rh[ROW_WELD0 + mi];
#endif
        if (mi < HRG_NBOXV && mj < HRG_NBOXV) {
#pragma unroll 4
          for (int q = 0; q < nq; q++) sval += L.rh[ROW_CON0 + q] * L.Jc[q][NV + mi] * L.Jc[q][NV + mj];
#if HRG_LIFT
          for (uint64_t mm = wmask; mm;) {
            const int q = __ffsll((long long)mm) - 1 + ROW_WELD0 - ROW_CON0; mm &= mm - 1;
            const double hq = L.rh[ROW_CON0 + q];
            if (hq != 0) sval += hq * L.Jc[q][NV + mi] * L.Jc[q][NV + mj];
          }
#endif
        }
        STAMP(23);
        if (__any(hh != 0 && rpart) || !h_is_m) {  // a robot row has curvature: invert the robot block of the Hessian (until then it is M, whose inverse is already held)
          Minv = spd_inverse1(hval, lane, &ok);
          if (!ok) break;
          h_is_m = false;
        }
        // cube block: 6x6 padded to the 8x8 lane layout with a unit diagonal
        const double Sinv = spd_inverse1_6(sval, lane, &ok);
        if (!ok) break;
        STAMP(24);
        wave_sync();
        const double x1 = -matvec_lanes(Minv, L.g, lane);
        const double x2 = -row8_sum(Sinv * (mj < HRG_NBOXV ? L.g[NV + mj] : 0.0));
        if (mj == 0) { L.d[mi] = x1; if (mi < HRG_NBOXV) L.d[NV + mi] = x2; }
      } else {
      // a robot-cube contact (lifting: the grip) couples the blocks: the 14-DoF Newton system as 2 x 2 register tiles (robot tree | cube, 6 x 6 padded with a unit
      // diagonal), lane (mi, mj) owns entry (mi, mj) of each; inverted in place by Gauss-Jordan sweeps over the 14 real pivots
      Tiles<2> T;
      {
        double t00 = Mij;
        if (mi == mj) { t00 += L.rh[mi]; t00 += L.rh[NV + 2 * mi]; t00 += L.rh[NV + 2 * mi + 1]; }
        double t11 = mi == mj ? (mi < 3 ? m.box_mass : 1.0) : 0.0;
        if (mi >= 3 && mi < HRG_NBOXV && mj >= 3 && mj < HRG_NBOXV) t11 = L.bMr[3 * (mi - 3) + (mj - 3)];
#if HRG_HANDOVER
        if (mi == mj && mi < HRG_NBOXV) t11 += L.rh[ROW_WELD0 + mi];
#endif
        double t01 = 0, t10 = 0;
        const bool ci = mi < HRG_NBOXV, cj = mj < HRG_NBOXV;
        const int bi_ = NV + (ci ? mi : 0), bj_ = NV + (cj ? mj : 0);
#pragma unroll 4
        for (int q = 0; q < nq; q++) {
          const double hq = L.rh[ROW_CON0 + q], ri = L.Jc[q][mi], rj = L.Jc[q][mj], bi = ci ? L.Jc[q][bi_] : 0.0, bj = cj ? L.Jc[q][bj_] : 0.0;
          t00 += hq * ri * rj; t01 += hq * ri * bj; t10 += hq * bi * rj; t11 += hq * bi * bj;
        }
#if HRG_LIFT
        for (uint64_t mm = wmask; mm;) {   // the connect rows (cube part only)
          const int q = __ffsll((long long)mm) - 1 + ROW_WELD0 - ROW_CON0; mm &= mm - 1;
          const double hq = L.rh[ROW_CON0 + q], bi = ci ? L.Jc[q][bi_] : 0.0, bj = cj ? L.Jc[q][bj_] : 0.0;
          t11 += hq * bi * bj;
        }
#endif
        T.t[0][0] = t00; T.t[0][1] = t01; T.t[1][0] = t10; T.t[1][1] = t11;
      }
      STAMP(23);
      {
        bool good = true;
        tiles_pivots<2, 0, 0>(T, mi, mj, NV, good);
        tiles_pivots<2, 1, 0>(T, mi, mj, HRG_NBOXV, good);
        if (!good) break;
      }
      STAMP(24);
      {
        const double g0 = L.g[mj], g1 = mj < HRG_NBOXV ? L.g[NV + mj] : 0.0;
        const double x0 = -row8_sum(T.t[0][0] * g0 + T.t[0][1] * g1), x1 = -row8_sum(T.t[1][0] * g0 + T.t[1][1] * g1);
        if (mj == 0) { L.d[mi] = x0; if (mi < HRG_NBOXV) L.d[NV + mi] = x1; }
      }
      }
      wave_sync();
      const double p = rowdot(L.d);
      double dd = 0, Mdi = 0;
      if (lane < NV) {
        dd = L.d[lane];
#pragma unroll
        for (int j = 0; j < NV; j++) Mdi += L.M[lane * NV + j] * L.d[j];
      } else if (lane < NVT) { dd = L.d[lane]; Mdi = box_Mrow(m.box_mass, lane - NV, &L.d[NV]); }
#else
      double hval = Mij;
      if (mi == mj) { hval += L.rh[mi]; hval += L.rh[NV + 2 * mi]; hval += L.rh[NV + 2 * mi + 1]; }
#pragma unroll 4
      for (int q = 0; q < nq; q++) hval += L.rh[ROW_CON0 + q] * L.Jc[q][mi] * L.Jc[q][mj];
      wave_sync();
      double gn = 0, sc = 0;
#pragma unroll
      for (int i = 0; i < NV; i++) { gn += L.g[i] * L.g[i]; sc += L.Ma0[i] * L.Ma0[i]; }
      if (fsqrt(gn) <= m.solver_tol * (1.0 + fsqrt(sc))) break;
      COUNT(18, (__any(hh != 0) || !h_is_m) ? 1 : 0);  // Hessian factorizations
      if (__any(hh != 0) || !h_is_m) {  // a row has curvature: invert the Hessian (until then H == M, whose inverse is already held)
        Minv = spd_inverse1(hval, lane, &ok);
        if (!ok) break;
        h_is_m = false;
      }
      {
        const double x = -matvec_lanes(Minv, L.g, lane);
        if (mj == 0) L.d[mi] = x;
      }
      wave_sync();
      const double p = rowdot(L.d);
      double dd = 0, Mdi = 0;
      if (lane < NV) {
        dd = L.d[lane];
#pragma unroll
        for (int j = 0; j < NV; j++) Mdi += L.M[lane * NV + j] * L.d[j];
      }
#endif
      STAMP(25);
      const double dMd = wave_sum(dd * Mdi), gd0 = wave_sum(dd * gm);
      double al = 1.0, lo = 0, hi = -1;
      const double d1_0 = gd0 + wave_sum(gg * p);
      const double noise = fabs(gd0) + dMd + wave_sum(fabs(gg * p));   // magnitude of the terms phi' is summed from
#pragma unroll 1
      for (int ls = 0; ls < 40; ls++) {
        COUNT(17, 1);  // line-search evaluations
        double c2, g2 = 0, h2 = 0;
        if (active) row_cost(type, D, floss, flim, y + al * p, &c2, &g2, &h2);
        const double d1 = gd0 + al * dMd + wave_sum(g2 * p);
        const double d2 = dMd + wave_sum(h2 * p * p);
        if (fabs(d1) <= 1e-10 * fabs(d1_0) || fabs(d1) <= 1e-14 * noise) break;   // converged, or down at the rounding noise of the sum's own terms
        if (d1 < 0) lo = al; else hi = al;
        double nx = al - d1 / d2;
        if (hi < 0) { if (!(nx > lo)) nx = 2 * al; }
        else if (!(nx > lo && nx < hi)) nx = 0.5 * (lo + hi);
        al = nx;
      }
      STAMP(26);
      if (lane < NVS) L.qacc[lane] += al * dd;
      wave_sync();
      // a full Newton step that stayed inside one quadratic piece of every row solved the problem exactly
      bool moved = false;
      if (active) moved = row_zone(type, flim, y) != row_zone(type, flim, y + p);
      if (al == 1.0 && !__any(moved)) break;
    }
  }
  // mj_checkAcc
  const bool badacc = lane < NVS && !(fabs(L.qacc[lane]) < 1e10);
  if (__any(badacc)) return 1;
  // mj_Euler with implicit joint damping: (M + h D) qacc' = M qacc
  euler_robot_tree(dm, lane, Mij, M0inv);
#if HRG_BOX
  { // free joint: no damping; the quaternion is integrated with the world-frame angular velocity
    double vnew = 0;
    if (lane < HRG_NBOXV) {
      const double acc = L.qacc[NV + lane];
      bx.acc_warmstart[lane] = acc;
      vnew = bx.vel[lane] + h * acc;
      bx.vel[lane] = vnew;
    }
    if (lane < 3) { const double p0 = bx.pos[lane]; bx.obs_pos[lane] = p0; bx.pos[lane] = p0 + h * vnew; }  // obs_pos: body_xpos of the forward pass inside mj_step
    const double w0 = __shfl(vnew, 3, 64), w1 = __shfl(vnew, 4, 64), w2 = __shfl(vnew, 5, 64);
    const double wn = fsqrt(w0 * w0 + w1 * w1 + w2 * w2), ang = h * wn;
    if (wn > 1e-12) {
      const double sh = sin(0.5 * ang) / wn, dq[4] = {cos(0.5 * ang), w0 * sh, w1 * sh, w2 * sh};
      double qo[4] = {bx.quat[0], bx.quat[1], bx.quat[2], bx.quat[3]}, qn[4];
      quatmul(qn, dq, qo);
      const double nn = fsqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
      wave_sync();
      if (lane < 4) bx.quat[lane] = qn[lane] / nn;
    }
  }
#endif
  wave_sync();
  return rmask != 0 ? 2 : 0;   // bit 0: diverged; bit 1: the robot tree carries contact rows (the caller's measure of how busy this env is)
}
#endif   // HRG_STACK

// ================================================================================================ env
// HumanEnv._check_action_safety (human_env.py:931-946) for the configuration in L.cq: static collision objects (table
// volume, mount pedestal: human_env.py:1301-1348) and self collision (pinocchio_manipulator_model.py:168-236) on the
// capsule model.  Lane 0 runs the chain kinematics, then lanes = capsule end points / capsule pairs, verdict by __any.
#define NCAP_CHECK 8
DI bool config_collides(const DevModel* __restrict__ dm_, int lane) {
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  double* cc = &L.scap[0][0][0];  // 8 capsules x (p1, p2), shield scratch is free here
  if (lane == 0) {
    double R[9], p[3], t[3];
    for (int k = 0; k < 9; k++) R[k] = dm->Rbase[k];
    v3cpy(p, m.base_pos);
    m3mulv(t, R, m.rcap_p1[0]); v3add(cc, p, t);
    m3mulv(t, R, m.rcap_p2[0]); v3add(cc + 3, p, t);
#pragma unroll 1
    for (int i = 0; i < NARM; i++) {
      double Rl[9], Rj[9];
      m3mul(Rl, R, dm->Rq[i]);
      m3mulv(t, R, m.body_pos[i]);
      v3add(p, p, t);
      axisangle2mat(Rj, m.jnt_axis[i], L.cq[i]);
      m3mul(R, Rl, Rj);
      m3mulv(t, R, m.rcap_p1[i + 1]); v3add(cc + 6 * (i + 1), p, t);
      m3mulv(t, R, m.rcap_p2[i + 1]); v3add(cc + 6 * (i + 1) + 3, p, t);
    }
    // capsule 7 = the gripper cylinder of the reference's URDF collision model = the shield's gripper capsule (on link 6)
    m3mulv(t, R, m.scap_p1[HRG_NSHIELD_RCAP - 1]); v3add(cc + 6 * (NARM + 1), p, t);
    m3mulv(t, R, m.scap_p2[HRG_NSHIELD_RCAP - 1]); v3add(cc + 6 * (NARM + 1) + 3, p, t);
  }
  wave_sync();
  bool hit = false;
  if (lane < 2 * (NCAP_CHECK - 1)) {
    const int c = 1 + (lane >> 1);
    const double* p = cc + 6 * c + 3 * (lane & 1);
    const double r = c == NCAP_CHECK - 1 ? m.scap_r[HRG_NSHIELD_RCAP - 1] : m.rcap_r[c], mg = m.obstacle_margin;
    if (p[2] - r < m.table_top_z + mg && fabs(p[0] - m.table_center[0]) <= m.table_half[0] + 0.5 * mg + r && fabs(p[1] - m.table_center[1]) <= m.table_half[1] + 0.5 * mg + r) hit = true;
    const double dx = p[0] - m.base_pos[0], dy = p[1] - m.base_pos[1];
    if (p[2] - r < m.base_cyl_z && fsqrt(dx * dx + dy * dy) < m.base_cyl_r + mg + r) hit = true;
  } else if (lane >= 16 && lane - 16 < dm->n_chk) {
    const int i = dm->chk_i[lane - 16], j = dm->chk_j[lane - 16];
    double x1[3], x2[3];
    const double rj = j == NCAP_CHECK - 1 ? m.scap_r[HRG_NSHIELD_RCAP - 1] : m.rcap_r[j];
    if (fsqrt(seg_seg(cc + 6 * i, cc + 6 * i + 3, cc + 6 * j, cc + 6 * j + 3, x1, x2)) - m.rcap_r[i] - rj < m.self_collision_safety) hit = true;
  }
  const bool any = __any(hit);
  wave_sync();
  return any;
}

// ReachHuman._sample_valid_pos (reach_human_env.py:525-548) -> L.st.cur_goal
DI void goal_sample(const DevModel* __restrict__ dm_, int lane, int64_t gid, int idx) {
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  hrg_env_state& s = L.st;
#pragma unroll 1
  for (int t = 0; t < 20; t++) {
    if (lane < NARM) {
      const double u = rng_u01(m.seed, (uint64_t)gid, (uint64_t)s.episode, STREAM_GOAL, (uint64_t)((idx * 20 + t) * NARM + lane));
      L.cq[lane] = m.qpos_limits[0][lane] + (m.qpos_limits[1][lane] - m.qpos_limits[0][lane]) * u;
    }
    wave_sync();
    const bool bad = m.goal_check && config_collides(dm_, lane);
    if (!bad) {
      if (lane < NARM) s.cur_goal[lane] = L.cq[lane];
      wave_sync();
      return;
    }
  }
  if (lane < NARM) s.cur_goal[lane] = 0.0;
  wave_sync();
}

// goal configuration of an action -> L.cq (HumanEnv.check_collision_action, human_env.py:588-627)
DI void action_goal(ModelPtr dm, int lane, const double* act) {
  Lds& L = g_L;
  const auto& m = dm->m;
  if (lane < NARM) {
    const double scale = fabs(m.act_out_max - m.act_out_min) / fabs(m.act_in_max - m.act_in_min);
    const double otr = 0.5 * (m.act_out_max + m.act_out_min), itr = 0.5 * (m.act_in_max + m.act_in_min);
    const double a = clampd(act[lane], m.act_in_min, m.act_in_max);
    L.cq[lane] = clampd(L.st.qpos[lane] + ((a - itr) * scale + otr), m.qpos_limits[0][lane], m.qpos_limits[1][lane]);
  }
  wave_sync();
}

// IKPositionDeltaWrapper.step (wrappers/ik_position_delta_wrapper.py:93-142): L.act = [dx, dy, dz, gripper] -> joint action.
// Damped least squares on [position; orientation] of the end-effector link (orientation held at the initial one).  Lane 0 runs
// the chain kinematics, lanes = joints build the Jacobian columns, lanes = (i, j) build and factor J J' + lambda^2 I (6x6 padded
// into the 8x8 lane Cholesky), lanes = joints apply the step.  Scratch: the shield's part of the LDS union (idle here).
__device__ __noinline__ void ik_action(const DevModel* __restrict__ dm_, int lane) {   // a real function: its atan2 and its 6 x 6 factorisation stay out of the step kernel's register allocation (they cost it 19 spill stores per step, used or not)
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  const hrg_env_state& s = L.st;
  double* sc = &L.scap[0][0][0];
  double *ax = sc, *org = sc + 18, *R6 = sc + 36, *pee = sc + 45, *ee = sc + 48;  // 54 of 84 doubles
  double* J = &L.rc[0][0];                                                          // [6][6] of 49 doubles
  const double grip = clampd(L.act[3], -1.0, 1.0);
  const double ws = lane < 3 ? clampd(L.act[lane], -m.ik_action_limit, m.ik_action_limit) * m.ik_x_output_max : 0.0;
  wave_sync();
  if (lane < NARM) L.cq[lane] = s.qpos[lane];
  wave_sync();
  double target = 0;  // lane a < 3 holds target[a]
#pragma unroll 1
  for (int it = 0;; it++) {
    if (lane == 0) {
      double R[9], p[3], t[3];
      for (int k = 0; k < 9; k++) R[k] = dm->Rbase[k];
      v3cpy(p, m.base_pos);
#pragma unroll 1
      for (int i = 0; i < NARM; i++) {
        double Rl[9], Rj[9];
        m3mul(Rl, R, dm->Rq[i]);
        m3mulv(t, R, m.body_pos[i]);
        v3add(p, p, t);
        axisangle2mat(Rj, m.jnt_axis[i], L.cq[i]);
        m3mul(R, Rl, Rj);
        m3mulv(t, R, m.jnt_axis[i]);
        v3cpy(ax + 3 * i, t);
        v3cpy(org + 3 * i, p);
      }
      for (int k = 0; k < 9; k++) R6[k] = R[k];
      m3mulv(t, R, m.ik_ee_offset);
      v3add(t, p, t);
      v3cpy(pee, t);
    }
    wave_sync();
    if (it == 0 && lane < 3) {
      target = pee[lane] + ws;
      if (m.ik_use_pos_limits) target = clampd(target, m.ik_pos_limits[0][lane], m.ik_pos_limits[1][lane]);
    }
    const double dl = lane < 3 ? target - pee[lane] : 0.0;
    const double d0 = __shfl(dl, 0, 64), d1 = __shfl(dl, 1, 64), d2 = __shfl(dl, 2, 64);
    if (it > 0 && fsqrt(d0 * d0 + d1 * d1 + d2 * d2) <= m.ik_residual_threshold) break;
    if (it >= m.ik_max_iter) break;
    if (lane == 0) {  // error vector: position, rotation vector of R_target R6'
      double E[9], v[3];
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) E[3 * a + b] = m.ik_target_rot[3 * a] * R6[3 * b] + m.ik_target_rot[3 * a + 1] * R6[3 * b + 1] + m.ik_target_rot[3 * a + 2] * R6[3 * b + 2];
      v3set(v, 0.5 * (E[7] - E[5]), 0.5 * (E[2] - E[6]), 0.5 * (E[3] - E[1]));
      const double sn = v3norm(v), cs = 0.5 * (E[0] + E[4] + E[8] - 1.0), ang = atan2(sn, cs);
      ee[0] = d0; ee[1] = d1; ee[2] = d2;
      for (int a = 0; a < 3; a++) ee[3 + a] = sn > 1e-12 ? v[a] / sn * ang : 0.0;
    }
    if (lane < NARM) {
      double r[3], c[3];
      v3sub(r, pee, org + 3 * lane);
      v3cross(c, ax + 3 * lane, r);
      for (int a = 0; a < 3; a++) { J[a * 6 + lane] = c[a]; J[(3 + a) * 6 + lane] = ax[3 * lane + a]; }
    }
    wave_sync();
    const int mi = lane >> 3, mj = lane & 7;
    double aij = mi == mj ? 1.0 : 0.0;
    if (mi < 6 && mj < 6) {
      double t = 0;
      for (int j = 0; j < NARM; j++) t += J[mi * 6 + j] * J[mj * 6 + j];
      aij = t + (mi == mj ? m.ik_damping * m.ik_damping : 0.0);
    }
    bool ok;
    const double lij = chol_lanes(aij, lane, &ok);
    if (!ok) break;
    chol_store(lij, lane, L.M, L.bias);   // scratch: the mass matrix and the bias vector are rebuilt by the first cycle's dynamics terms before anything reads them
    wave_sync();
    const double y = chol_solve_lanes(L.M, L.bias, lane < 6 ? ee[lane] : 0.0, lane);
    double dq = 0;
    for (int a = 0; a < 6; a++) { const double ya = __shfl(y, a, 64); if (lane < NARM) dq += J[a * 6 + lane] * ya; }
    double mx = 0;
    for (int j = 0; j < NARM; j++) { const double v = fabs(__shfl(dq, j, 64)); if (v > mx) mx = v; }
    const double scl = mx > 0.25 * HRG_PI ? 0.25 * HRG_PI / mx : 1.0;
    wave_sync();
    if (lane < NARM) L.cq[lane] = L.cq[lane] + scl * dq;
    wave_sync();
  }
  wave_sync();
  const double out = lane < NARM ? L.cq[lane] - s.qpos[lane] : grip;
  wave_sync();
  if (lane <= NARM) L.act[lane] = out;
  wave_sync();
}

// CollisionPreventionWrapper.action (wrappers/collision_prevention_wrapper.py:46-103) on L.act
__device__ __noinline__ void screen_action(const DevModel* __restrict__ dm_, int lane, int64_t gid) {   // (a real function, like ik_action)
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  hrg_env_state& s = L.st;
  if (!m.cp_enabled) return;
  action_goal(dm, lane, L.act);
  if (!config_collides(dm_, lane)) return;
  s.action_resamples = s.action_resamples + 1;
  double* cand = L.cv;              // 7 values in cv[6] + ca[0] (contiguous)
  double* best = &L.rc[0][0];
  double bestd = 1e300;
  int found = 0;
  if (m.cp_replace_type != 0) {
#pragma unroll 1
    for (int t = 0; t < m.cp_n_resamples; t++) {
      double dd = 0;
      if (lane < HRG_ACT_DIM) {
        const double c = 2.0 * rng_u01(m.seed, (uint64_t)gid, (uint64_t)s.episode, STREAM_ACTION, (uint64_t)((s.timestep * 64 + t) * HRG_ACT_DIM + lane)) - 1.0;
        cand[lane] = c;
        dd = (L.act[lane] - c) * (L.act[lane] - c);
      }
      const double d = wave_sum(dd);
      wave_sync();
      action_goal(dm, lane, cand);
      if (config_collides(dm_, lane)) continue;
      if (m.cp_replace_type == 1) { if (lane < HRG_ACT_DIM) best[lane] = cand[lane]; found = 1; wave_sync(); break; }
      if (d < bestd) { bestd = d; if (lane < HRG_ACT_DIM) best[lane] = cand[lane]; found = 1; }
      wave_sync();
    }
  }
  if (lane < HRG_ACT_DIM) L.act[lane] = found ? best[lane] : 0.0;
  wave_sync();
}

#if HRG_LIFT || HRG_HANDOVER
// rotation matrix (row-major) -> unit quaternion (w, x, y, z), largest-component branch
DI void mat2quat(double* q, const double* R) {
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0) { const double s_ = fsqrt(tr + 1.0) * 2; q[0] = 0.25 * s_; q[1] = (R[7] - R[5]) / s_; q[2] = (R[2] - R[6]) / s_; q[3] = (R[3] - R[1]) / s_; }
  else if (R[0] > R[4] && R[0] > R[8]) { const double s_ = fsqrt(1.0 + R[0] - R[4] - R[8]) * 2; q[0] = (R[7] - R[5]) / s_; q[1] = 0.25 * s_; q[2] = (R[1] + R[3]) / s_; q[3] = (R[2] + R[6]) / s_; }
  else if (R[4] > R[8]) { const double s_ = fsqrt(1.0 + R[4] - R[0] - R[8]) * 2; q[0] = (R[2] - R[6]) / s_; q[1] = (R[1] + R[3]) / s_; q[2] = 0.25 * s_; q[3] = (R[5] + R[7]) / s_; }
  else { const double s_ = fsqrt(1.0 + R[8] - R[0] - R[4]) * 2; q[0] = (R[3] - R[1]) / s_; q[1] = (R[2] + R[6]) / s_; q[2] = (R[5] + R[7]) / s_; q[3] = 0.25 * s_; }
}
#endif
// observation: object-state (vec/dist eef -> L hand, R hand, head; human_env.py:1536-1590) + goal_difference
// (environments/manipulation/reach_human_env.py:649-651); lanes = observation entries
DI void write_obs(const DevModel* __restrict__ dm_, int lane, const double* goal, float* out) {
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  const hrg_env_state& s = L.st;
  if (lane < HRG_OBS_DIM) {
    double v;
    if (lane < 12) {
      const int k = lane >> 2, a = lane & 3;
      const int site = k == 0 ? m.site_lhand : (k == 1 ? m.site_rhand : m.site_head);
      double d[3];
      v3sub(d, s.human_site[site], s.eef_pos);
      v = a < 3 ? d[a] : v3norm(d);
    } else if (lane < 18) v = goal[lane - 12] - s.qpos[lane - 12];
    else if (lane < 24) v = s.qpos[lane - 18];      // robot0_joint_pos
    else if (lane < 30) v = s.qvel[lane - 24];      // robot0_joint_vel
    else if (lane < 33) v = s.eef_pos[lane - 30];   // robot0_eef_pos
    else if (lane < 39) v = goal[lane - 33];        // desired_goal
    else if (lane >= 53 && lane < 55) v = s.qpos[NARM + lane - 53];  // robot0_gripper_qpos
    else if (lane >= 55 && lane < 57) v = s.qvel[NARM + lane - 55];  // robot0_gripper_qvel
    else v = 0.0;                                   // PickPlaceHumanCart columns
#if HRG_STACK
    // CollaborativeStackingCart._setup_observables (collaborative_stacking_cartesian_env.py:1306-1524): vec_eef_to_all_objects (a, b, l, r) in the 12 joint-space
    // columns the cube tasks leave empty, object_gripped 39, vec_eef_to_object 40:43 (the cube the robot places next: b, then a), vec_eef_to_target 43:46,
    // gripper_aperture 46, that cube's position 47:50, next_target_pos 50:53 (sk.target: the eef position while there is no target)
    const hrg_stack_state& sk = L.sk;
    const int nxt = sk.n_stack >= 2 ? HRG_CUBE_A : HRG_CUBE_B;
    if (lane >= 12 && lane < 18) v = sk.obs_pos[(lane - 12) / 3][(lane - 12) % 3] - s.eef_pos[(lane - 12) % 3];
    else if (lane >= 33 && lane < 39) v = sk.obs_pos[2 + (lane - 33) / 3][(lane - 33) % 3] - s.eef_pos[(lane - 33) % 3];
    else if (lane == 39) v = (double)sk.gripped;
    else if (lane >= 40 && lane < 43) v = sk.obs_pos[nxt][lane - 40] - s.eef_pos[lane - 40];
    else if (lane >= 43 && lane < 46) v = sk.target[lane - 43] - s.eef_pos[lane - 43];
    else if (lane == 46) {
      double ap = 0;
      for (int f = 0; f < HRG_NFINGER; f++) ap += (s.qpos[NARM + f] - m.finger_qpos_range[0][f]) / (m.finger_qpos_range[1][f] - m.finger_qpos_range[0][f]);
      v = ap / HRG_NFINGER;
    } else if (lane >= 47 && lane < 50) v = sk.obs_pos[nxt][lane - 47];
    else if (lane >= 50 && lane < 53) v = sk.target[lane - 50];
#endif
#if HRG_HAMMER
    // CollaborativeHammeringCart._setup_observables (collaborative_hammering_cartesian_env.py:1151-1323; oracle: compute_obs_hammer): hammer_quat (w, x, y, z) 12:16,
    // board_pos 33:36, vec_eef_to_board 36:39, hammer_gripped 39, vec_eef_to_hammer 40:43, vec_eef_to_nail 43:46, gripper_aperture 46, hammer_pos 47:50, nail_pos 50:53,
    // board_quat (x, y, z, w) 57:61, nail_hammering_progress 61
    const hrg_hammer_state& hm = L.hm;
    if (lane >= 12 && lane < 16) v = hm.quat[1][lane - 12];
    else if ((lane >= 16 && lane < 18)) v = 0.0;
    else if (lane >= 33 && lane < 36) v = hm.obs_pos[0][lane - 33];
    else if (lane >= 36 && lane < 39) v = hm.obs_pos[0][lane - 36] - s.eef_pos[lane - 36];
    else if (lane == 39) v = (double)hm.gripped;
    else if (lane >= 40 && lane < 43) v = hm.obs_pos[1][lane - 40] - s.eef_pos[lane - 40];
    else if (lane >= 43 && lane < 46) v = hm.obs_pos[2][lane - 43] - s.eef_pos[lane - 43];
    else if (lane == 46) {
      double ap = 0;
      for (int f = 0; f < HRG_NFINGER; f++) ap += (s.qpos[NARM + f] - m.finger_qpos_range[0][f]) / (m.finger_qpos_range[1][f] - m.finger_qpos_range[0][f]);
      v = ap / HRG_NFINGER;
    } else if (lane >= 47 && lane < 50) v = hm.obs_pos[1][lane - 47];
    else if (lane >= 50 && lane < 53) v = hm.obs_pos[2][lane - 50];
    else if (lane >= 57 && lane < 60) v = hm.quat[0][1 + lane - 57];
    else if (lane == 60) v = hm.quat[0][0];
    else if (lane == 61) v = clampd(hm.nail_q / m.hm_nail_range, 0.0, 1.0);
#endif
#if HRG_BOX
    // PickPlaceHumanCart._setup_observables (pick_place_human_cartesian_env.py:726-841), gripper_aperture (human_env.py:1508-1524)
    const hrg_box_state& bx = L.bx;
    if (m.task == HRG_TASK_REACH_BOX) {}   // ReachHuman does not observe its smallBox: the ReachHuman layout above stands
    else
    if ((lane >= 12 && lane < 18) || (lane >= 33 && lane < 39)) v = 0.0;
    if (m.task == HRG_TASK_REACH_BOX) {}
    else
    if (lane >= 12 && lane < 16) v = bx.quat[lane == 15 ? 0 : lane - 11];   // object_quat, (x, y, z, w) like T.convert_quat(..., to="xyzw") (human_robot_handover_cartesian_env.py:849-858)
    else if (lane == 39) v = (double)bx.gripped;
    else if (lane >= 40 && lane < 43) v = bx.obs_pos[lane - 40] - s.eef_pos[lane - 40];
    else if (lane >= 43 && lane < 46) v = bx.target[lane - 43] - s.eef_pos[lane - 43];
    else if (lane == 46) {
      double ap = 0;
      for (int f = 0; f < HRG_NFINGER; f++) ap += (s.qpos[NARM + f] - m.finger_qpos_range[0][f]) / (m.finger_qpos_range[1][f] - m.finger_qpos_range[0][f]);
      v = ap / HRG_NFINGER;
    } else if (lane >= 47 && lane < 50) v = bx.obs_pos[lane - 47];
    else if (lane >= 50 && lane < 53) v = bx.target[lane - 50];
#if HRG_LIFT || HRG_HANDOVER
    if (lane >= 57 && lane < 61 && (HRG_IS_HANDOVER(m.task) || m.task == HRG_TASK_LIFTING)) {
      // quat_eef_to_object (human_robot_handover_cartesian_env.py:916-924, robot_human_handover_cartesian_env.py:998-1006) / quat_eef_to_board
      // (collaborative_lifting_cartesian_env.py:1057-1065) as the reference computes them (oracle: compute_obs_e): the (x, y, z, w) observables are read as
      // (w, x, y, z), so the product is that of the scrambled quaternions (scalar x, vector (y, z, w)); A conj(B) as (x, y, z, w)
      double qe[4], Re[9];
      for (int a = 0; a < 9; a++) Re[a] = L.kR[NARM][a];
      mat2quat(qe, Re);
      const double sA = bx.quat[1], vA[3] = {bx.quat[2], bx.quat[3], bx.quat[0]}, sB = qe[1], vB[3] = {qe[2], qe[3], qe[0]};
      double cr[3];
      v3cross(cr, vA, vB);
      v = lane == 60 ? sA * sB + v3dot(vA, vB) : -sA * vB[lane - 57] + sB * vA[lane - 57] - cr[lane - 57];
    }
#endif
#if HRG_LIFT
    if (m.task == HRG_TASK_LIFTING) { // collaborative_lifting_cartesian_env.py:982-1085: board_balance in the first target column, board_quat (x, y, z, w) in 43-45 and 51
      if (lane >= 43 && lane < 46) v = bx.quat[1 + lane - 43];
      else if (lane == 51) v = bx.quat[0];
      else if (lane == 52) v = 0.0;
      else if (lane == 50) { double Rx[9]; const double qb[4] = {bx.quat[0], bx.quat[1], bx.quat[2], bx.quat[3]}; quat2mat(Rx, qb); v = Rx[8]; }
    }
#endif
#endif
    out[lane] = (float)v;
  }
}

#if HRG_HANDOVER
// The human takes the object: pose of the holding hand at the current animation frame -> mocap body, object teleported into it, weld
// on (_control_human + _human_pickup_object, human_robot_handover_cartesian_env.py:598-633, 700-711).  The reference teleports to the stale
// mocap pose first and lets the weld drag the object over; here the fresh hand pose is used directly.
DI void handover_pickup(const DevModel* __restrict__ dm_, int lane, int64_t gid, bool at_reset) {
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  human_control(dm_, lane, gid);   // human pose + L.hand_q
  const int hl = dm->clips.clip_holding_hand[clip_of(dm, gid, L.st.episode, L.st.anim_index)];
  const int site = hl ? m.site_lhand : m.site_rhand;
  wave_sync();
  hrg_box_state& bx = L.bx;
  const bool r2h = m.task == HRG_TASK_HANDOVER_R2H;   // there the human starts empty-handed: only the mocap body (= the target) is posed
  if (lane < 3) {
    const double p = L.st.human_site[site][lane] + L.hand_off[lane];
    bx.mocap_pos[lane] = p;
    if (r2h) bx.target[lane] = p;
    else { bx.pos[lane] = p; if (at_reset) bx.obs_pos[lane] = p; bx.weld_off[lane] = 0.0; }
  }
  if (lane < 4) {
    const double q = L.hand_q[lane];
    bx.mocap_quat[lane] = q;
    if (!r2h) { bx.quat[lane] = q; bx.weld_rel[lane] = lane == 0 ? 1.0 : 0.0; }
  }
  bx.weld_active = r2h ? 0 : 1;
  wave_sync();
}
#endif
#if HRG_BOX
// i-th object placement / target of an episode (UniformRandomSampler over the bins, pick_place_human_cartesian_env.py:613-635,
// 843-875), counter-based; wave-uniform
DI void placement_of(ModelPtr dm, int64_t gid, int episode, int idx, int target, double* p) {
  const auto& m = dm->m;
  const uint64_t st = target ? STREAM_TARGET : STREAM_OBJECT;
  const double u0 = rng_u01(m.seed, (uint64_t)gid, (uint64_t)episode, st, (uint64_t)(2 * idx)), u1 = rng_u01(m.seed, (uint64_t)gid, (uint64_t)episode, st, (uint64_t)(2 * idx + 1));
  if (target) { p[0] = m.tgt_bin[0] + (m.tgt_bin[1] - m.tgt_bin[0]) * u0; p[1] = m.tgt_bin[2] + (m.tgt_bin[3] - m.tgt_bin[2]) * u1; p[2] = m.tgt_z; }
  else { p[0] = m.obj_bin[0] + (m.obj_bin[1] - m.obj_bin[0]) * u0; p[1] = m.obj_bin[2] + (m.obj_bin[3] - m.obj_bin[2]) * u1; p[2] = m.obj_z; }
}
#endif

#if HRG_STACK
// next_target_position (collaborative_stacking_cartesian_env.py:522-548) -> sk.target / sk.has_target; the eef position while it is not the robot's turn (1347-1355).
// Wave-uniform; called before an observation is written.
DI void stack_update_target(const DevModel* __restrict__ dm_, int lane, int64_t gid) {
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  hrg_stack_state& sk = L.sk;
  const int left = dm->clips.clip_holding_hand[clip_of(dm, gid, L.st.episode, L.st.anim_index)];   // first_placing_hand
  int c = -1;
  if (sk.task_phase == HRG_STK_WAIT_FOR_SECOND) c = left ? HRG_CUBE_L : HRG_CUBE_R;
  else if (sk.task_phase == HRG_STK_WAIT_FOR_FOURTH) c = left ? HRG_CUBE_R : HRG_CUBE_L;
  double tv = 0;
  if (lane < 3) tv = c >= 0 ? sk.obs_pos[c][lane] + (lane == 2 ? 2.0 * m.box_half[2] : 0.0) : L.st.eef_pos[lane];
  wave_sync();
  if (lane < 3) sk.target[lane] = tv;
  sk.has_target = c >= 0;
  wave_sync();
}
// _id_of_cube_at_target (590-610): a robot cube within goal_dist (maximum norm) of the target that is not part of the stack; -1 = none.  Needs sk.target.
DI int stack_cube_at_target(ModelPtr dm) {
  const hrg_stack_state& sk = g_L.sk;
  if (!sk.has_target) return -1;
  for (int c = HRG_CUBE_A; c <= HRG_CUBE_B; c++) {
    double dmax = 0;
    for (int a = 0; a < 3; a++) { const double d = fabs(sk.target[a] - sk.obs_pos[c][a]); if (d > dmax) dmax = d; }
    bool in_stack = false;
    for (int q = 0; q < NCUBE; q++) if (q < sk.n_stack && sk.stack_ids[q] == c) in_stack = true;
    if (dmax < dm->m.goal_dist && !in_stack) return c;
  }
  return -1;
}
// _human_pickup_objects (1053-1056) as restated in the oracle (stack_pickup): both cubes put into the hands at rest, both welds on.  Needs sk.mocap_*.
DI void stack_pickup(const DevModel* __restrict__ dm_, int lane) {
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  hrg_stack_state& sk = L.sk;
  wave_sync();
  if (lane < 2) {
    const int hd = lane, c = HRG_CUBE_L + hd;
    double Rm[9], t[3];
    const double qm[4] = {sk.mocap_quat[hd][0], sk.mocap_quat[hd][1], sk.mocap_quat[hd][2], sk.mocap_quat[hd][3]};
    quat2mat(Rm, qm);
    m3mulv(t, Rm, m.stack_weld_relpos);
    for (int a = 0; a < 3; a++) { const double pv = sk.mocap_pos[hd][a] - t[a]; sk.pos[c][a] = pv; sk.obs_pos[c][a] = pv; }
    for (int a = 0; a < 4; a++) sk.quat[c][a] = qm[a];
    for (int a = 0; a < 6; a++) { sk.vel[c][a] = 0.0; sk.acc_warmstart[c][a] = 0.0; }
    sk.weld_active[hd] = 1;
  }
  wave_sync();
}
// idx-th placement of robot cube c (oracle stack_placement); wave-uniform
DI void stack_placement(ModelPtr dm, int64_t gid, int episode, int idx, int c, double* p) {
  const auto& m = dm->m;
  const double u0 = rng_u01(m.seed, (uint64_t)gid, (uint64_t)episode, STREAM_OBJECT, (uint64_t)(4 * idx + 2 * c)), u1 = rng_u01(m.seed, (uint64_t)gid, (uint64_t)episode, STREAM_OBJECT, (uint64_t)(4 * idx + 2 * c + 1));
  p[0] = m.obj_bin[0] + (m.obj_bin[1] - m.obj_bin[0]) * u0;
  p[1] = m.obj_bin[2] + (m.obj_bin[3] - m.obj_bin[2]) * u1;
  p[2] = m.obj_z;
}
// _reset_animation (991-998) after the human has been posed
DI void stack_reset_animation(const DevModel* __restrict__ dm_, int lane) {
  hrg_stack_state& sk = g_L.sk;
  wave_sync();
  sk.task_phase = HRG_STK_APPROACH; sk.n_delayed[0] = 0; sk.n_delayed[1] = 0; sk.n_stack = 0; sk.max_stack_height = 0;
  if (lane < NCUBE) sk.stack_ids[lane] = -1;
  stack_pickup(dm_, lane);
}
#endif

#if HRG_HAMMER
// _reset_board + _human_take_board_from_table + _reset_nail as restated in the oracle (hammer_take_board): the board where the right-hand weld holds it, at rest, the
// nail pulled out at its nail_index-th placement.  Needs hm.mocap_*.
DI void hammer_take_board(const DevModel* __restrict__ dm_, int lane, int64_t gid) {
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  hrg_hammer_state& hm = L.hm;
  wave_sync();
  const double qm[4] = {hm.mocap_quat[1][0], hm.mocap_quat[1][1], hm.mocap_quat[1][2], hm.mocap_quat[1][3]};
  const double qi[4] = {m.hm_weld_relquat[0], -m.hm_weld_relquat[1], -m.hm_weld_relquat[2], -m.hm_weld_relquat[3]};
  double qt[4], Rt[9], t[3];
  quatmul(qt, qm, qi);
  quat2mat(Rt, qt);
  m3mulv(t, Rt, m.hm_anchor[1]);
  const double u0 = rng_u01(m.seed, (uint64_t)gid, (uint64_t)L.st.episode, STREAM_OBJECT, (uint64_t)(2 * hm.nail_index)),
               u1 = rng_u01(m.seed, (uint64_t)gid, (uint64_t)L.st.episode, STREAM_OBJECT, (uint64_t)(2 * hm.nail_index + 1));
  wave_sync();
  if (lane < 4) hm.quat[0][lane] = qt[lane];
  if (lane < 3) hm.pos[0][lane] = hm.mocap_pos[1][lane] - t[lane];
  if (lane < 6) { hm.vel[0][lane] = 0.0; hm.acc_warmstart[0][lane] = 0.0; }
  hm.nail_q = 0.0; hm.nail_v = 0.0; hm.nail_acc_warmstart = 0.0; hm.nail_touch = 0; hm.pad_ = 0;
  if (lane == 0) hm.nail_xy[0] = m.hm_nail_bin[0] + (m.hm_nail_bin[1] - m.hm_nail_bin[0]) * u0;
  if (lane == 1) hm.nail_xy[1] = m.hm_nail_bin[2] + (m.hm_nail_bin[3] - m.hm_nail_bin[2]) * u1;
  wave_sync();
}
#endif

DI void eef_update(const DevModel* __restrict__ dm_) {
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  double t[3], e[3];
  m3mulv(t, L.kR[NARM - 1], dm->m.eef_pos);
  v3add(e, L.kp[NARM - 1], t);
  v3cpy(L.st.eef_pos, e);
}

#if HRG_LIFT
// _update_mocap_body_transforms (collaborative_lifting_cartesian_env.py:590-616): the two mocap bodies sit at the hand sites
DI void lifting_mocap(const DevModel* __restrict__ dm_, int lane) {
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  wave_sync();
  if (lane < 3) { L.bx.mocap_pos[lane] = L.st.human_site[m.site_lhand][lane]; L.bx.weld_off[lane] = L.st.human_site[m.site_rhand][lane]; }
  wave_sync();
}
// CollaborativeLiftingCart._reset_animation (644-657) puts the board where the Schunk gripper's init_qpos straddles it; with the stand-in gripper
// the pose follows from the gripper frame (see oracle/hrg_oracle.c lifting_place_board).  Needs the chain kinematics and eef_pos of the posture.
DI void lifting_place_board(const DevModel* __restrict__ dm_, int lane) {
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  hrg_box_state& bx = L.bx;
  double Re[9], Rb[9], q[4], xb[3], yb[3], zb[3], ze[3], pos[3];
  for (int a = 0; a < 9; a++) Re[a] = L.kR[NARM][a];   // a finger body's frame = the hand frame (right_hand is turned -45 deg about link 6's axis): y = closing axis
  const double sg = Re[7] >= 0 ? 1.0 : -1.0;
  for (int a = 0; a < 3; a++) { ze[a] = Re[3 * a + 2]; xb[a] = -ze[a]; zb[a] = sg * Re[3 * a + 1]; }
  v3cross(yb, zb, xb);
  for (int a = 0; a < 3; a++) { Rb[3 * a] = xb[a]; Rb[3 * a + 1] = yb[a]; Rb[3 * a + 2] = zb[a]; }
  mat2quat(q, Rb);
  for (int a = 0; a < 3; a++) pos[a] = L.st.eef_pos[a] + ze[a] * (m.box_half[0] - m.lift_grip_depth);
  wave_sync();
  if (lane < 3) { bx.pos[lane] = pos[lane]; bx.obs_pos[lane] = pos[lane]; }
  if (lane < 4) bx.quat[lane] = q[lane];
  if (lane < HRG_NBOXV) { bx.vel[lane] = 0.0; bx.acc_warmstart[lane] = 0.0; }
  wave_sync();
}
#endif

// HumanEnv._reset_internal (human_env.py:1604-1673) + ReachHuman._reset_internal (reach_human_env.py:509-523)
// + FailsafeController.reset (failsafe_controller.py:204-250)
// (a real function: an episode ends once in ~100 steps; its draws -- log, cos -- and goal sampling stay out of the step kernel's register allocation)
__device__ __noinline__ void env_reset(const DevModel* __restrict__ dm_, int lane, int64_t gid, float* obs_out) {
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  hrg_env_state& s = L.st;
  const int episode = s.episode + 1;
  wave_sync();
  for (int k = lane; k < (int)(sizeof(hrg_env_state) / sizeof(double)); k += 64) ((double*)&s)[k] = 0.0;
  wave_sync();
  s.episode = episode;
  s.stream_id = (int32_t)gid;
  if (lane < NARM) s.qpos[lane] = m.init_qpos[lane] + m.init_noise * rng_gauss(m.seed, (uint64_t)gid, (uint64_t)episode, STREAM_NOISE, (uint64_t)lane);
  else if (lane < NV) s.qpos[lane] = m.finger_init_qpos[lane - NARM];
#if HRG_HAMMER
  // _put_hammer_into_gripper (790-812) closes the gripper on the handle: the fingers start where their pads touch it, commanded shut
  if (lane >= NARM && lane < NV) s.qpos[lane] = m.hm_finger_grip_qpos[lane - NARM];
  s.grip_action = -1.0;
#endif
  const double ux = rng_u01(m.seed, (uint64_t)gid, (uint64_t)episode, STREAM_HUMAN, 0), uy = rng_u01(m.seed, (uint64_t)gid, (uint64_t)episode, STREAM_HUMAN, 1),
               uz = rng_u01(m.seed, (uint64_t)gid, (uint64_t)episode, STREAM_HUMAN, 2);
  s.human_pos_offset[0] = m.base_human_pos_offset[0] + (2 * ux - 1) * m.human_rand[0];
  s.human_pos_offset[1] = m.base_human_pos_offset[1] + (2 * uy - 1) * m.human_rand[1];
  s.human_pos_offset[2] = m.base_human_pos_offset[2];
  const double yaw = (2 * uz - 1) * m.human_rand[2];
  s.human_rot_offset[0] = cos(0.5 * yaw); s.human_rot_offset[1] = 0; s.human_rot_offset[2] = 0; s.human_rot_offset[3] = sin(0.5 * yaw);
  s.animation_time = -1;
  wave_sync();
  const double p0[3] = {0, 0, 0}, q0[4] = {1, 0, 0, 0};
  human_fk_lanes(dm_, lane, p0, q0, nullptr);
  robot_chain_fk(dm_, lane, false);
  eef_update(dm_);
  shield_reset(dm_, lane);
  wave_sync();
#if HRG_BOX
  { // PickPlaceHumanCart._reset_internal: first object placement and target, object at rest
    hrg_box_state& bx = L.bx;
    for (int k = lane; k < (int)(sizeof(hrg_box_state) / sizeof(double)); k += 64) ((double*)&bx)[k] = 0.0;
    wave_sync();
    double po[3], pt[3];
    placement_of(dm, gid, episode, 0, 0, po);
    placement_of(dm, gid, episode, 0, 1, pt);
    if (m.task == HRG_TASK_INSPECTION) { // the target comes with the animation (info json), not from a bin
      const int clip = clip_of(dm, gid, episode, 0);
      for (int a = 0; a < 3; a++) pt[a] = dm->clips.clip_target_pos[clip][a] + s.human_pos_offset[a];
    }
    if (lane < 3) { bx.pos[lane] = po[lane]; bx.obs_pos[lane] = po[lane]; bx.target[lane] = pt[lane]; }
    if (lane == 0) bx.quat[0] = 1.0;
    wave_sync();
#if HRG_HANDOVER
    if (HRG_IS_HANDOVER(m.task)) handover_pickup(dm_, lane, gid, true);   // _reset_animation + _control_human (H2R 635-647, 686-711; R2H 650-660, 700-706)
#endif
#if HRG_LIFT
    if (m.task == HRG_TASK_LIFTING) { // _reset_internal (670-681): _control_human + _reset_animation; the human holds the board from the start
      human_control(dm_, lane, gid);
      lifting_mocap(dm_, lane);
      lifting_place_board(dm_, lane);
      bx.weld_active = 1;
      if (lane < 3) bx.target[lane] = bx.pos[lane];
      wave_sync();
    }
#endif
  }
  if (m.task == HRG_TASK_REACH_BOX) goal_sample(dm_, lane, gid, 0);   // ReachHuman with its smallBox: the reach goals of ReachHuman._reset_internal
#elif HRG_STACK
  { // CollaborativeStackingCart._reset_internal (938-959): the robot's cubes in their bin, the human's cubes in the hands, phase APPROACH
    hrg_stack_state& sk = L.sk;
    for (int k = lane; k < (int)(sizeof(hrg_stack_state) / sizeof(double)); k += 64) ((double*)&sk)[k] = 0.0;
    wave_sync();
    double pa[3], pb[3];
    stack_placement(dm, gid, episode, 0, HRG_CUBE_A, pa);
    stack_placement(dm, gid, episode, 0, HRG_CUBE_B, pb);
    if (lane < 3) { sk.pos[HRG_CUBE_A][lane] = pa[lane]; sk.obs_pos[HRG_CUBE_A][lane] = pa[lane]; sk.pos[HRG_CUBE_B][lane] = pb[lane]; sk.obs_pos[HRG_CUBE_B][lane] = pb[lane]; }
    if (lane < 2) sk.quat[lane][0] = 1.0;
    wave_sync();
    human_control(dm_, lane, gid);
    stack_reset_animation(dm_, lane);
    stack_update_target(dm_, lane, gid);
  }
#elif HRG_HAMMER
  { // CollaborativeHammeringCart._reset_internal (717-747): the human holds the board, the hammer sits in the closed gripper, phase APPROACH
    hrg_hammer_state& hm = L.hm;
    for (int k = lane; k < (int)(sizeof(hrg_hammer_state) / sizeof(double)); k += 64) ((double*)&hm)[k] = 0.0;
    wave_sync();
    hm.quat[0][0] = 1.0;
    wave_sync();
    human_control(dm_, lane, gid);
    hammer_take_board(dm_, lane, gid);
    double Rg[9], t[3];
    quat2mat(Rg, m.hm_hammer_grip_quat);
    m3mulv(t, Rg, m.hm_hammer_com);
    if (lane < 4) hm.quat[1][lane] = m.hm_hammer_grip_quat[lane];
    if (lane < 3) hm.pos[1][lane] = s.eef_pos[lane] + t[lane];
    wave_sync();
    hammer_geometry(dm_, lane);
    hammer_obs_pos(dm_, lane);
    wave_sync();
  }
#else
  goal_sample(dm_, lane, gid, 0);
#endif
  if (obs_out) write_obs(dm_, lane, s.cur_goal, obs_out);
  wave_sync();
}

#if HRG_HANDOVER
// end of the handover tasks' first physics pass: time, grip site, the hand mocap body re-posed (_update_mocap_body_transform, 609-633), then the robot
// terms of sim.forward() for the second pass.  A real function: its temporaries stay out of the cycle body's register allocation.
__device__ __noinline__ void handover_first_pass_tail(const DevModel* __restrict__ dm_, int lane, int64_t gid) {
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  hrg_env_state& s = L.st;
      s.time = s.time + m.timestep;
  eef_update(dm_);
  const int hl = dm->clips.clip_holding_hand[clip_of(dm, gid, s.episode, s.anim_index)];
  const int site = hl ? m.site_lhand : m.site_rhand;
  wave_sync();
  if (lane < 3) {
    const double mp = s.human_site[site][lane] + L.hand_off[lane];
    L.bx.mocap_pos[lane] = mp;
    if (m.task == HRG_TASK_HANDOVER_R2H) L.bx.target[lane] = mp;   // target_pos property: the hand the object has to reach (448-450)
  }
  if (lane < 4) L.bx.mocap_quat[lane] = L.hand_q[lane];
  wave_sync();
  robot_chain_fk(dm_, lane, false);
  robot_dynamics_terms(dm_, lane);
  // the human capsules share their LDS with the solver scratch of the step above: lay them out again for the second collision phase
  for (int k = lane; k < HRG_NHB * 6; k += 64) (&L.hcap[0][0])[k] = (&L.hcap_keep[0][0])[k];
  wave_sync();
}
#endif

// HRG_FAIR: even progress of the waves that share a SIMD.  A launch ends with its slowest wave; the four (three, two) single-wave workgroups of a SIMD saturate its
// issue slots, and the arbiter serves the oldest ready wave first, so the youngest wave of every SIMD is starved while its siblings run and then finishes alone,
// latency-bound (one instruction every ~16 cycles instead of every 4).  Every wave publishes its progress (the shield cycle it is in) in a table indexed by its
// hardware position (XCC, SE, SH, CU, SIMD | wave slot), reads its siblings' entries at the top of every cycle and raises its issue priority when it is behind,
// lowers it when it is ahead.  The order in which waves issue never changes what an env computes.
// Measured per kernel at 4096 envs (8192 for PickPlace), law 2 (last of its SIMD -> 3, behind -> 2, level -> 1, leading -> 0, busy envs at least 2): ReachHuman 1.389 -> 1.319 ms,
// PickPlace 4.21 -> 4.10 ms, lifting 3.10 -> 3.04 ms, inspection unchanged; the handover kernel (two waves per SIMD) and the stacking kernel are 1 - 2 % slower with it and keep
// the plain busy-first priority, as does the one-wave hammering kernel.  Checking three times per cycle instead of once is slower everywhere (1.41 ms).
#ifndef HRG_FAIR
#if HRG_HANDOVER || HRG_STACK || HRG_HAMMER
#define HRG_FAIR 0
#else
#define HRG_FAIR 2
#endif
#endif
#if HRG_FAIR
// the table: progress + 1 of the wave in slot w of a SIMD (0: no wave there); device memory owned by the host side, one per device (StepOrder.fair)
// The readers of an entry are the waves of the SAME compute unit (other workgroups, but behind the same vector L1 and the same XCD's L2): workgroup-scope atomics
// (plain cached loads / write-through stores) are coherent among them in practice; agent scope sends every access past the caches (16 MB more HBM traffic per
// 4096-env launch).  The memory model does not promise visibility across workgroups at workgroup scope: the table is a heuristic, a stale entry costs a priority
// level for one cycle and never a result.
#ifndef FAIR_SCOPE
#define FAIR_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
#endif
DI int fair_key() {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
  const unsigned wave = hw & 0xf, simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
  return (int)((((((xcc & 7) * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd) * 8 + (wave & 7));
}
DI void fair_publish(int* g_fair, int lane, int progress) {   // progress >= 0; -1 = this wave is done
  const int key = fair_key();
  if (lane == 0) __hip_atomic_store(&g_fair[key], progress + 1, __ATOMIC_RELAXED, FAIR_SCOPE);
}
// top of a cycle: publish this wave's progress and start reading the siblings' (the load's latency hides behind the shield phase) ...
DI int fair_begin(int* g_fair, int lane, int progress) {
  const int key = fair_key(), base = key & ~7;
  if (lane == 0) __hip_atomic_store(&g_fair[key], progress + 1, __ATOMIC_RELAXED, FAIR_SCOPE);
  int v = 0;
  if (lane < 8) v = __hip_atomic_load(&g_fair[base + lane], __ATOMIC_RELAXED, FAIR_SCOPE);
  return v;
}
// ... and after it: behind a sibling -> higher issue priority, ahead of them -> lower
DI void fair_apply(int lane, int progress, int v, bool busy) {
  const int me = fair_key() & 7;
  const bool other = lane < 8 && lane != me && v > 0;
  int vmin = other ? v : 0x7fffffff, vmax = other ? v : 0;
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) {
    const int a = __shfl_xor(vmin, o, 64), b = __shfl_xor(vmax, o, 64);
    vmin = a < vmin ? a : vmin; vmax = b > vmax ? b : vmax;
  }
  vmin = __builtin_amdgcn_readfirstlane(vmin); vmax = __builtin_amdgcn_readfirstlane(vmax);
  const int own = progress + 1;
  int pr;
  if (vmax == 0) pr = 0;                           // no sibling
  else {
    pr = own < vmax ? (own <= vmin ? 3 : 2)        // somebody is ahead of this wave: last -> 3, in between -> 2
                    : (own > vmin ? 0 : 1);        // this wave leads -> 0; all level -> 1
  }
  if (busy && pr < 2) pr = 2;                      // busy envs at least 2, also when no sibling has published yet
  pr = __builtin_amdgcn_readfirstlane(pr);
  if (pr == 3) __builtin_amdgcn_s_setprio(3);
  else if (pr == 2) __builtin_amdgcn_s_setprio(2);
  else if (pr == 1) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
}
#endif

// one shield cycle (human_env.py:503-526): controller goal on policy steps, shield, dynamics terms, PD+ torque,
// human playback, contacts, integration.  Returns 1 when the simulation diverged.
DI int cycle_body(const DevModel* __restrict__ dm_, int lane, int e, int64_t gid, int cyc, int busy, double* dbg_r, double* dbg_h, int32_t* dbg_nh, int32_t* fair) {
  (void)fair;
  // opaque per cycle: nothing derived from the lane id is hoisted out of the 25-cycle loop and kept live (spilled)
  asm volatile("" : "+v"(lane));
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  hrg_env_state& s = L.st;
  STAMP(0);
#define FAIR_P(k) cyc
#if HRG_FAIR
  int fair_v = fair_begin(fair, lane, FAIR_P(0));
  const bool fair_busy = __builtin_amdgcn_readfirstlane((busy || !s.is_safe) ? 1 : 0) != 0;
#elif !defined(HRG_NO_PRIO)
  // a launch ends with its slowest wave, and the waves of a SIMD share its issue slots: an env whose robot was in contact in the last substep (Newton
  // iterations, Hessian inversions) or that is under a fail-safe manoeuvre (replanning every cycle) has the longer instruction stream ahead of it, so its
  // wave issues first (measured with ReachHuman at 4096 envs: 2.06 -> 1.79 ms per step)
  // (the condition goes through readfirstlane: s_setprio is a scalar instruction that ignores the exec mask, so a lane-masked if/else would run both arms)
  if (__builtin_amdgcn_readfirstlane((busy || !s.is_safe) ? 1 : 0)) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
#endif
  if (cyc == 0) { // FailsafeController.set_goal, failsafe_controller.py:252-300
    if (lane < NARM) {
      const double scale = fabs(m.act_out_max - m.act_out_min) / fabs(m.act_in_max - m.act_in_min);
      const double otr = 0.5 * (m.act_out_max + m.act_out_min), itr = 0.5 * (m.act_in_max + m.act_in_min);
      const double a = clampd(L.act[lane], m.act_in_min, m.act_in_max);
      const double g = clampd(s.qpos[lane] + ((a - itr) * scale + otr), m.qpos_limits[0][lane], m.qpos_limits[1][lane]);
      s.goal_qpos[lane] = g;
      s.new_goal_q[lane] = g;
    }
    s.new_goal = 1;
    wave_sync();
  }
  // humanMeasurement + SafetyShield.step; also runs the chain kinematics of sim.forward() #1
#ifdef HRG_STAMPS
  const int pm = dm->phase_mask;   // diagnostic build only (tools/phase_timing.sh): phases can be switched off for timing, results are then invalid
#else
  constexpr int pm = 0xff;         // shipping build: every phase always runs; there is no switch
#endif
  if (pm & 1) shield_step(dm_, lane, e, dbg_r, dbg_h, dbg_nh); else robot_chain_fk(dm_, lane, false);
  STAMP(1);
#if HRG_FAIR
  fair_apply(lane, FAIR_P(0), fair_v, fair_busy);
#endif
  if (pm & 2) robot_dynamics_terms(dm_, lane);
  STAMP(2);
  if (cyc == 0) { // Controller.update(): mj_fullM -> stale 6x6 block
    if (lane < NARM * NARM) s.mass_matrix[lane] = L.M[(lane / NARM) * NV + (lane % NARM)];
    wave_sync();
  }
  if (lane < NARM) { // run_controller, failsafe_controller.py:356-369
    double t = 0;
    for (int j = 0; j < NARM; j++) t += s.mass_matrix[lane * NARM + j] * (m.kp * (s.des_q[j] - s.qpos[j]) + m.kd * (s.des_v[j] - s.qvel[j]) + s.des_a[j]);
    const double tq = clampd(t + L.bias[lane], m.arm_ctrlrange[lane][0], m.arm_ctrlrange[lane][1]);
    L.ctrl[lane] = tq;
  }
  { // gripper: RethinkGripper.format_action + ctrl-range mapping
    const double grip_a = L.act[NARM];
    const double sg = grip_a > 0 ? 1.0 : (grip_a < 0 ? -1.0 : 0.0);
    const double ga = clampd(s.grip_action - m.gripper_speed * sg, -1.0, 1.0);
    wave_sync();
    s.grip_action = ga;
    if (lane < HRG_NFINGER) {
      const double lo = m.finger_ctrlrange[lane][0], hi = m.finger_ctrlrange[lane][1];
      L.ctrl[NARM + lane] = 0.5 * (hi + lo) + 0.5 * (hi - lo) * (lane == 0 ? ga : -ga);
    }
  }
  if (!L.acc_failsafe && !s.is_safe) { L.acc_failsafe = 1; s.failsafe_interventions = s.failsafe_interventions + 1; }
  wave_sync();
  STAMP(3);
  if (pm & 4) human_control(dm_, lane, gid); // _control_human + kinematics of sim.forward() #2
#if HRG_LIFT
  if (m.task == HRG_TASK_LIFTING) lifting_mocap(dm_, lane);   // CollaborativeLiftingCart._control_human (583-588)
#endif
  STAMP(4);
  int ncon = 0;
  int crash = 0, busy_out = 0;
#if HRG_HANDOVER
  // HumanRobotHandoverCart._control_human (human_robot_handover_cartesian_env.py:598-633) runs one more sim.step() with the new human pose
  // (no bookkeeping), then re-poses the hand mocap body and sim.forward() runs again: pass 0 = that step, pass 1 = the cycle's regular step.
  // One loop body for both passes keeps a single inlined copy of the contact and solver code.
  if (HRG_IS_HANDOVER(m.task)) {
    for (int k = lane; k < HRG_NHB * 6; k += 64) (&L.hcap_keep[0][0])[k] = (&L.hcap[0][0])[k];
  }
#pragma unroll 1
  for (int pass = HRG_IS_HANDOVER(m.task) ? 0 : 1; pass < 2 && !crash; pass++) {
    collide(dm_, lane, &ncon);
    STAMP(5);
    if (pass == 1) {
      int hc = L.acc_has_collision, ct = L.acc_collision_type;
      classify(dm_, lane, ncon, &hc, &ct);
      L.acc_has_collision = hc; L.acc_collision_type = ct;
      STAMP(6);
    }
    { const int r = dynamics_step(dm_, lane, ncon); crash = r & 1; busy_out |= r & 2; }
    STAMP(7);
    if (pass == 0 && !crash) {
      handover_first_pass_tail(dm_, lane, gid);
      STAMP(27);
    }
  }
#else
  if (pm & 8) collide(dm_, lane, &ncon);
  STAMP(5);
  int hc = L.acc_has_collision, ct = L.acc_collision_type;
  classify(dm_, lane, ncon, &hc, &ct);
  L.acc_has_collision = hc; L.acc_collision_type = ct;
  STAMP(6);
  if (pm & 16) { const int r = dynamics_step(dm_, lane, ncon); crash = r & 1; busy_out |= r & 2; }
  STAMP(7);
#endif
  if (!crash) {
    s.time = s.time + m.timestep;
    eef_update(dm_);
    s.low_level_time = s.low_level_time + 1;
  }
  wave_sync();
  STAMP_FLUSH(lane);
  return crash | busy_out;
}

// HumanEnv.step (human_env.py:470-586) + ReachHuman.step tail (reach_human_env.py:399-407) + TimeLimit
// (wrappers/time_limit.py:31-44) + VecEnv auto-reset
DI int env_step(const DevModel* __restrict__ dm_, int lane, int e, int64_t own_gid, double* __restrict__ action, float* obs, float* term_obs,
                 float* reward, uint8_t* done, int32_t* info, double* dbg_r, double* dbg_h, int32_t* dbg_nh, int32_t* fair) {
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  hrg_env_state& s = L.st;
  (void)fair;
  const int64_t gid = s.stream_id;  // in-episode draws follow the state's streams (= own_gid unless the state was copied in)
  int has_collision = 0, collision_type = HRG_COL_NULL, crash = 0;
  STAMP_INIT(lane);
  if (lane < NV) L.act[lane] = lane < HRG_ACT_DIM ? action[lane] : 0.0;
  wave_sync();
  if (m.ik_enabled) ik_action(dm_, lane);  // IKPositionDeltaWrapper is the outermost action wrapper (utils/training_utils.py:358-373)
  if (m.cp_enabled) screen_action(dm_, lane, gid);  // CollisionPreventionWrapper.step wraps env.step: uses the pre-step state
#if HRG_LIFT
  if (m.task == HRG_TASK_LIFTING) { wave_sync(); if (lane == 0) L.act[NARM] = 1.0; wave_sync(); }   // CollaborativeLiftingCart.step (368-391): the gripper action is replaced by 'close'
#endif
#if HRG_HAMMER
  if (!m.gripper_controllable) { wave_sync(); if (lane == 0) L.act[NARM] = 1.0; wave_sync(); }   // CollaborativeHammeringCart.step (486-487): always close the gripper
#endif
  if ((m.cp_enabled || m.ik_enabled) && lane < HRG_ACT_DIM) action[lane] = L.act[lane];
  s.timestep = s.timestep + 1;
  L.acc_has_collision = 0; L.acc_collision_type = HRG_COL_NULL; L.acc_failsafe = 0;
  wave_sync();
  int busy = 0;   // of the previous substep; a step starts unbiased
#pragma unroll 1
  for (int cyc = 0; cyc < m.n_cycles && !crash; cyc++) { const int r = cycle_body(dm_, lane, e, gid, cyc, busy, dbg_r, dbg_h, dbg_nh, fair); crash = r & 1; busy = r >> 1; }
#if HRG_FAIR
  fair_publish(fair, lane, -1);   // this wave's slot is free again
  __builtin_amdgcn_s_setprio(0);
#endif
  has_collision = L.acc_has_collision; collision_type = L.acc_collision_type;
  // ---- observation / success / info / reward / done ----
  double goal[NARM];
  for (int j = 0; j < NARM; j++) goal[j] = s.cur_goal[j];
#if HRG_BOX
  if (m.task == HRG_TASK_POINTING) { // target_pos property: the elbow -> hand ray extended to the table (pick_place_pointing_human_cartesian_env.py:336-360)
    const int left = dm->clips.clip_pointing_hand[clip_of(dm, gid, s.episode, s.anim_index)];
    const int sh = left ? m.site_lhand : m.site_rhand, se = left ? m.site_lelbow : m.site_relbow;
    double dz = s.human_site[sh][2] - s.human_site[se][2];
    if (dz == 0) dz += 1e-6;
    const double scaling = (s.human_site[sh][2] - m.table_top_z) / dz;
    double tv = 0;
    if (lane < 3) { const double dl = lane == 2 ? dz : s.human_site[sh][lane] - s.human_site[se][lane]; tv = s.human_site[sh][lane] - scaling * dl; }
    wave_sync();
    if (lane < 3) L.bx.target[lane] = tv;
    wave_sync();
  }
#endif
#if HRG_STACK
  stack_update_target(dm_, lane, gid);
#endif
  write_obs(dm_, lane, goal, term_obs);
#if HRG_STACK
  // CollaborativeStackingCart: success = the animation ran to its end (_check_success, 746-756); _sparse_reward (700-744); _check_stack_toppled (656-671)
  hrg_stack_state& sk = L.sk;
  const int goal_reached = !crash && sk.task_phase == HRG_STK_COMPLETE;
  int toppled = 0;
  if (sk.n_stack >= 2) {
    const double min_h = sk.obs_pos[sk.stack_ids[0]][2] + m.box_half[2];
    for (int q = 1; q < NCUBE; q++) if (q < sk.n_stack && sk.obs_pos[sk.stack_ids[q]][2] < min_h) toppled = 1;
  }
  const int at_target = stack_cube_at_target(dm);
  // _check_first / _check_second_manipulation_object_in_target_zone (620-654)
  const int first_zone = sk.n_stack < 1 ? -1 : (sk.n_stack > 1 ? sk.stack_ids[1] : at_target);
  const int second_zone = sk.n_stack < 3 ? -1 : (sk.n_stack > 3 ? sk.stack_ids[3] : at_target);
  double r;
  if (goal_reached) r = m.task_reward;
  else if (toppled) r = m.stack_toppled_reward;
  else {
    r = second_zone >= 0 ? m.fourth_cube_at_target_reward : (first_zone >= 0 ? m.second_cube_at_target_reward : -1.0);
    if (sk.gripped) r += m.object_gripped_reward;
  }
  const double dense = 0.0;   // _dense_reward is a TODO returning 0 (681-698)
#elif HRG_HAMMER
  // CollaborativeHammeringCart: success = the animation ran to its end (_check_success, 576-587); _sparse_reward (522-556); _check_nail_hammered_in (505-520)
  hrg_hammer_state& hm = L.hm;
  const double progress = clampd(hm.nail_q / m.hm_nail_range, 0.0, 1.0);
  const int hammered_in = 1.0 - progress < m.hm_goal_tolerance;
  const int goal_reached = !crash && hm.task_phase == HRG_HM_COMPLETE;
  double r;
  if (goal_reached) r = m.task_reward;
  else {
    r = hammered_in ? m.nail_hammered_in_reward : -1.0;
    if (hm.gripped) r += m.hammer_gripped_reward_bonus;
  }
  const double dense = 0.0;   // _dense_reward is a TODO returning 0 (558-574)
#elif HRG_BOX
  // PickPlaceHumanCart: achieved goal = [eef_pos, object_pos, object_gripped], desired goal = target_pos (574-611);
  // _check_object_in_target_zone (550-572), _sparse_reward (471-500), _dense_reward (502-526)
  hrg_box_state& bx = L.bx;
  double e2o = 0, o2t = 0;
  for (int a = 0; a < 3; a++) { e2o += (bx.obs_pos[a] - s.eef_pos[a]) * (bx.obs_pos[a] - s.eef_pos[a]); o2t += (bx.target[a] - bx.obs_pos[a]) * (bx.target[a] - bx.obs_pos[a]); }
  const int in_zone = fsqrt(o2t) <= m.goal_dist;
  // HumanObjectInspectionCart: success = the inspection animation ran to its end (human_object_inspection_cartesian_env.py:553-600)
  const int inspection = m.task == HRG_TASK_INSPECTION || m.task == HRG_TASK_HANDOVER_H2R;   // success = the task's animation ran to its end
  int goal_reached = !crash && (m.task == HRG_TASK_HANDOVER_R2H ? bx.task_phase == HRG_R2H_COMPLETE : ((inspection || m.task == HRG_TASK_LIFTING) ? bx.task_phase == HRG_PHASE_COMPLETE : in_zone));
  double r = goal_reached ? m.task_reward : ((inspection && in_zone) ? m.object_at_target_reward : (bx.gripped ? m.object_gripped_reward : -1.0));
#if !HRG_HANDOVER && !HRG_LIFT
  double reach_dist = 0;
  if (m.task == HRG_TASK_REACH_BOX) {   // ReachHuman with its smallBox: the task logic of ReachHuman (reach_human_env.py:437-475; human_env.py:666-691)
    double dist2 = 0;
    for (int j = 0; j < NARM; j++) dist2 += (s.qpos[j] - goal[j]) * (s.qpos[j] - goal[j]);
    reach_dist = fsqrt(dist2);
    goal_reached = !crash && reach_dist <= m.goal_dist;
    r = goal_reached ? m.task_reward : -1.0;
  }
#endif
#if HRG_HANDOVER
  if (m.task == HRG_TASK_HANDOVER_R2H)   // robot_human_handover_cartesian_env.py:507-555
    r = goal_reached ? m.task_reward : (bx.task_phase == HRG_R2H_RETREAT ? m.object_in_human_hand_reward : (bx.gripped ? m.object_gripped_reward : -1.0));
#endif
  double dense = -(fsqrt(e2o) * 0.2 + fsqrt(o2t)) * 0.1;
#if !HRG_HANDOVER && !HRG_LIFT
  if (m.task == HRG_TASK_REACH_BOX) dense = -0.1 * reach_dist;
#endif
#if HRG_LIFT
  double balance = 1.0;
  if (m.task == HRG_TASK_LIFTING) { // collaborative_lifting_cartesian_env.py:429-507: success = the animation ran to its end; base reward +1; dense = normalised balance angle - 2
    double Rx[9];
    const double qb[4] = {bx.quat[0], bx.quat[1], bx.quat[2], bx.quat[3]};
    quat2mat(Rx, qb);
    balance = Rx[8];
    r = (!crash && bx.task_phase == HRG_PHASE_COMPLETE) ? m.task_reward : (balance < m.min_balance ? m.imbalance_failure_reward : (!bx.gripped ? m.board_released_reward : 1.0));
    const double ba = asin(clampd(balance, -1.0, 1.0)) * 2 / HRG_PI, mba = asin(m.min_balance) * 2 / HRG_PI;
    dense = (ba - mba) / (1 - mba) - 2.0;
  }
#endif
#else
  double dist2 = 0;
  for (int j = 0; j < NARM; j++) dist2 += (s.qpos[j] - goal[j]) * (s.qpos[j] - goal[j]);
  const double dist = fsqrt(dist2);
  const int goal_reached = !crash && dist <= m.goal_dist;
  double r = goal_reached ? m.task_reward : -1.0;
  const double dense = -0.1 * dist;
#endif
  if (goal_reached) s.n_goal_reached = s.n_goal_reached + 1;
  const int illegal = (collision_type & (HRG_COL_STATIC | HRG_COL_ROBOT | HRG_COL_HUMAN_CRIT)) != 0;
  if (m.reward_shaping) r += 1.0 + dense;
  if (illegal) r += m.collision_reward;
  r *= m.reward_scale;
  int d = 0;
  if (crash) { r += m.sim_crash_reward; d = 1; }
  else {
    if (m.done_at_collision && illegal) d = 1;
    if (m.done_at_success && goal_reached) d = 1;
#if HRG_STACK
    if (toppled) d = 1;   // _check_done (758-778)
#endif
#if HRG_LIFT
    if (m.task == HRG_TASK_LIFTING) { // _check_done (509-561): unbalanced, or the board out of the gripper for more than 5 steps in a row
      const int nd = L.bx.gripped ? 0 : L.bx.n_delayed + 1;
      wave_sync();
      L.bx.n_delayed = nd;
      wave_sync();
      if (balance < m.min_balance || nd > 5) d = 1;
    }
#endif
  }
  const int ncoll = s.n_collisions_static + s.n_collisions_robot + s.n_collisions_human + s.n_collisions_critical;
  int truncated = 0;
  if (s.timestep >= m.horizon) { truncated = !d; d = 1; }
  if (lane < HRG_INFO_DIM) {
    int v = 0;
    switch (lane) {
      case HRG_INFO_COLLISION: v = has_collision; break;
      case HRG_INFO_COLLISION_TYPE: v = collision_type; break;
      case HRG_INFO_N_COLLISIONS: v = ncoll; break;
      case HRG_INFO_N_COLLISIONS_STATIC: v = s.n_collisions_static; break;
      case HRG_INFO_N_COLLISIONS_ROBOT: v = s.n_collisions_robot; break;
      case HRG_INFO_N_COLLISIONS_HUMAN: v = s.n_collisions_human; break;
      case HRG_INFO_N_COLLISIONS_CRITICAL: v = s.n_collisions_critical; break;
      case HRG_INFO_TIMEOUT: v = s.timestep >= m.horizon; break;
      case HRG_INFO_FAILSAFE_INTERVENTIONS: v = s.failsafe_interventions; break;
      case HRG_INFO_N_GOAL_REACHED: v = s.n_goal_reached; break;
      case HRG_INFO_TRUNCATED: v = truncated; break;
      case HRG_INFO_SIM_CRASH: v = crash; break;
      case HRG_INFO_ACTION_RESAMPLES: v = s.action_resamples; break;
#if HRG_BOX
      case HRG_INFO_N_OBJECT_HANDED_OVER: v = L.bx.n_handed_over; break;
#endif
#if HRG_STACK
      case HRG_INFO_MAX_STACK_HEIGHT: v = L.sk.max_stack_height; break;   // _get_info (673-679): the value before this step's transitions
#endif
    }
    info[lane] = v;
  }
  if (lane == 0) { *reward = (float)r; *done = (uint8_t)d; }
  wave_sync();
#if HRG_BOX
  if (!d) write_obs(dm_, lane, goal, obs);  // the step's observation predates _on_goal_reached (pick_place_human_cartesian_env.py:414-438)
  wave_sync();
#if HRG_LIFT
  if (m.task == HRG_TASK_LIFTING) {
    if (goal_reached && !m.done_at_success && !d) { // _on_goal_reached (631-642): the robot back at its initial posture (deterministic), controller reset, next animation,
                                                     // _control_human, board back in the gripper
      const int ai = (s.anim_index + 1) % m.n_anim_ids, st = (int)((double)s.low_level_time / m.anim_step_length);
      wave_sync();
      if (lane < NV) { s.qpos[lane] = lane < NARM ? m.init_qpos[lane] : m.finger_init_qpos[lane - NARM]; s.qvel[lane] = 0.0; s.qacc_warmstart[lane] = 0.0; }
      s.grip_action = 0.0;
      { // FailsafeController.reset: the shield's trajectory / path / measurement memory back to the state of a fresh episode
        double* z0 = (double*)&s.ltt;
        const int nz = (int)(((const double*)&s.human_site[0][0]) - z0);
        for (int k = lane; k < nz; k += 64) z0[k] = 0.0;
      }
      s.new_goal = 0; s.n_meas = 0;
      s.anim_index = ai; s.animation_time = 0; s.anim_start_time = st;
      bx.task_phase = HRG_PHASE_APPROACH; bx.n_delayed = 0;
      wave_sync();
      robot_chain_fk(dm_, lane, false);
      eef_update(dm_);
      shield_reset(dm_, lane);
      wave_sync();
      human_control(dm_, lane, gid);
      lifting_mocap(dm_, lane);
      lifting_place_board(dm_, lane);
    }
  } else
#endif
#if HRG_HANDOVER
  if (m.task == HRG_TASK_HANDOVER_R2H && !d) {
    if (goal_reached && !m.done_at_success) { // _on_goal_reached (robot_human_handover_cartesian_env.py:662-676): next placement, next animation, the human lets go
      const int oi = (bx.obj_index + 1) % m.n_obj_placements;
      double po[3];
      placement_of(dm, gid, s.episode, oi, 0, po);
      const int ai = (s.anim_index + 1) % m.n_anim_ids, st = (int)((double)s.low_level_time / m.anim_step_length);
      wave_sync();
      bx.obj_index = oi;
      if (lane < 3) bx.pos[lane] = po[lane];
      if (lane < 4) bx.quat[lane] = lane == 0 ? 1.0 : 0.0;
      s.anim_index = ai; s.animation_time = 0; s.anim_start_time = st;
      bx.task_phase = HRG_R2H_APPROACH; bx.n_delayed = 0;
      wave_sync();
      handover_pickup(dm_, lane, gid, false);   // poses the mocap body / target, weld off
    }
    // RobotHumanHandoverCart.step (452-474): the human takes the object when it touches the palm of the extended hand; the weld keeps the
    // pose the object has relative to the hand mocap body at that moment (730-748)
    const int ph = bx.task_phase;
    wave_sync();
    if (ph == HRG_R2H_REACH_OUT && L.palm_hit) {
      double Rm[9], dv[3], off[3], wr[4];
      const double qm[4] = {bx.mocap_quat[0], bx.mocap_quat[1], bx.mocap_quat[2], bx.mocap_quat[3]}, qo[4] = {bx.quat[0], bx.quat[1], bx.quat[2], bx.quat[3]};
      const double qc[4] = {qm[0], -qm[1], -qm[2], -qm[3]};
      quat2mat(Rm, qm);
      for (int a = 0; a < 3; a++) dv[a] = bx.pos[a] - bx.mocap_pos[a];
      for (int a = 0; a < 3; a++) off[a] = Rm[a] * dv[0] + Rm[3 + a] * dv[1] + Rm[6 + a] * dv[2];
      quatmul(wr, qc, qo);
      wave_sync();
      if (lane < 3) bx.weld_off[lane] = off[lane];
      if (lane < 4) bx.weld_rel[lane] = wr[lane];
      bx.weld_active = 1; bx.task_phase = HRG_R2H_RETREAT; bx.n_handed_over = bx.n_handed_over + 1;
    }
    wave_sync();
  } else
  if (m.task == HRG_TASK_HANDOVER_H2R && !d) {
    if (goal_reached && !m.done_at_success) { // _on_goal_reached (human_robot_handover_cartesian_env.py:649-668): next target, next animation, the human picks the object up again
      const int ti = (bx.tgt_index + 1) % m.n_targets;
      double pt[3];
      placement_of(dm, gid, s.episode, ti, 1, pt);
      const int ai = (s.anim_index + 1) % m.n_anim_ids, st = (int)((double)s.low_level_time / m.anim_step_length);
      wave_sync();
      bx.tgt_index = ti;
      if (lane < 3) bx.target[lane] = pt[lane];
      s.anim_index = ai; s.animation_time = 0; s.anim_start_time = st;
      bx.task_phase = HRG_PHASE_APPROACH; bx.n_delayed = 0; bx.n_delayed2 = 0;
      wave_sync();
      handover_pickup(dm_, lane, gid, false);   // body_xpos (the observed object position) is refreshed by the next forward pass
    }
    // HumanRobotHandoverCart.step (465-483): the human lets go once the robot has gripped the object; the retreat starts when it is placed
    const int ph = bx.task_phase;
    wave_sync();
    if (ph == HRG_PHASE_PRESENT && bx.gripped) { bx.weld_active = 0; bx.task_phase = HRG_PHASE_WAIT; bx.n_handed_over = bx.n_handed_over + 1; }
    else if (ph == HRG_PHASE_WAIT && in_zone) bx.task_phase = HRG_PHASE_RETREAT;
    wave_sync();
  } else
#endif
  if (inspection && !d) {
    if (goal_reached && !m.done_at_success) { // _on_goal_reached (human_object_inspection_cartesian_env.py:492-505): next placement, next animation
      const int oi = (bx.obj_index + 1) % m.n_obj_placements;
      double po[3];
      placement_of(dm, gid, s.episode, oi, 0, po);
      const int ai = (s.anim_index + 1) % m.n_anim_ids, st = (int)((double)s.low_level_time / m.anim_step_length);
      wave_sync();
      bx.obj_index = oi;
      if (lane < 3) bx.pos[lane] = po[lane];
      if (lane < 4) bx.quat[lane] = lane == 0 ? 1.0 : 0.0;
      s.anim_index = ai; s.animation_time = 0; s.anim_start_time = st;
      bx.task_phase = HRG_PHASE_APPROACH; bx.n_delayed = 0;
      wave_sync();
    }
    // HumanObjectInspectionCart.step (461-490): READY -> INSPECTION when the object is in the target zone, back when it leaves it
    const int ph = bx.task_phase;
    int nph = ph;
    if (ph == HRG_PHASE_READY && in_zone) nph = HRG_PHASE_INSPECTION;
    else if (ph == HRG_PHASE_INSPECTION && !(fsqrt(o2t) - m.goal_exit_tolerance <= m.goal_dist)) nph = HRG_PHASE_READY;
    wave_sync();
    bx.task_phase = nph;
    wave_sync();
  } else
#if !HRG_HANDOVER && !HRG_LIFT
  if (m.task == HRG_TASK_REACH_BOX) {
    if (goal_reached && !d) { // reach_human_env.py:399-407
      s.goal_index = (s.goal_index + 1) % m.n_goals;
      goal_sample(dm_, lane, gid, s.goal_index);   // (the step's observation, written above, still shows the goal that was reached)
    }
  } else
#endif
  if (goal_reached && !d) { // _on_goal_reached (440-453): next target, object teleported to its next placement (velocity kept)
    const int ti = (bx.tgt_index + 1) % m.n_targets, oi = (bx.obj_index + 1) % m.n_obj_placements;
    double po[3], pt[3];
    placement_of(dm, gid, s.episode, oi, 0, po);
    placement_of(dm, gid, s.episode, ti, 1, pt);
    wave_sync();
    bx.tgt_index = ti; bx.obj_index = oi;
    if (lane < 3) { bx.pos[lane] = po[lane]; bx.target[lane] = pt[lane]; }
    if (lane < 4) bx.quat[lane] = lane == 0 ? 1.0 : 0.0;
    wave_sync();
  }
#elif HRG_STACK
  if (!d) write_obs(dm_, lane, goal, obs);   // the step's observation predates the transitions below (CollaborativeStackingCart.step, 550-588)
  wave_sync();
  if (goal_reached && !m.done_at_success && !d) { // _on_goal_reached (961-977): next placements of the robot's cubes (velocities kept), next animation, both cubes back in the hands
    const int oi = (sk.obj_index + 1) % m.n_obj_placements;
    double pa[3], pb[3];
    stack_placement(dm, gid, s.episode, oi, HRG_CUBE_A, pa);
    stack_placement(dm, gid, s.episode, oi, HRG_CUBE_B, pb);
    const int ai = (s.anim_index + 1) % m.n_anim_ids, st = (int)((double)s.low_level_time / m.anim_step_length);
    wave_sync();
    sk.obj_index = oi;
    if (lane < 3) { sk.pos[HRG_CUBE_A][lane] = pa[lane]; sk.pos[HRG_CUBE_B][lane] = pb[lane]; }
    if (lane < 4) { sk.quat[HRG_CUBE_A][lane] = lane == 0 ? 1.0 : 0.0; sk.quat[HRG_CUBE_B][lane] = lane == 0 ? 1.0 : 0.0; }
    s.anim_index = ai; s.animation_time = 0; s.anim_start_time = st;
    sk.task_phase = HRG_STK_APPROACH; sk.n_delayed[0] = 0; sk.n_delayed[1] = 0;
    wave_sync();
    human_control(dm_, lane, gid);
    stack_reset_animation(dm_, lane);
  }
  if (!d) { // the phase machine of CollaborativeStackingCart.step (563-586); wave-uniform
    const int clip = clip_of(dm, gid, s.episode, s.anim_index);
    const int left = dm->clips.clip_holding_hand[clip], k1 = dm->clips.clip_stack_keyframes[clip][1], k3 = dm->clips.clip_stack_keyframes[clip][3];
    const int ph = sk.task_phase, ns = sk.n_stack, at = s.animation_time, gr = sk.gripped;
    const int fz = ns < 1 ? -1 : (ns > 1 ? sk.stack_ids[1] : stack_cube_at_target(dm)), sz = ns < 3 ? -1 : (ns > 3 ? sk.stack_ids[3] : stack_cube_at_target(dm));
    double top[3] = {0, 0, 0};
    if (ns >= 2) for (int a = 0; a < 3; a++) top[a] = sk.obs_pos[sk.stack_ids[1]][a];
    wave_sync();
    int nns = ns;
    if (ph == HRG_STK_PLACE_FIRST && at > k1) { // _human_place_first_object (1004-1011)
      const int hd = left ? 0 : 1;
      sk.stack_ids[ns] = HRG_CUBE_L + hd; nns = ns + 1;
      sk.weld_active[hd] = 0;
      sk.task_phase = HRG_STK_WAIT_FOR_SECOND;
    } else if (ph == HRG_STK_WAIT_FOR_SECOND && fz >= 0 && !gr) {
      sk.task_phase = HRG_STK_PLACE_THIRD;
      sk.stack_ids[ns] = fz; nns = ns + 1;
    } else if (ph == HRG_STK_PLACE_THIRD && at > k3) { // _human_place_third_object (1013-1043): released directly above the second cube, at rest
      const int hd = left ? 1 : 0, c = HRG_CUBE_L + hd;
      sk.weld_active[hd] = 0;
      if (lane < 3) sk.pos[c][lane] = top[lane] + (lane == 2 ? 2.0 * m.box_half[2] : 0.0);
      if (lane < 4) sk.quat[c][lane] = lane == 0 ? 1.0 : 0.0;
      if (lane < 6) sk.vel[c][lane] = 0.0;
      sk.stack_ids[ns] = c; nns = ns + 1;
      sk.task_phase = HRG_STK_WAIT_FOR_FOURTH;
    } else if (ph == HRG_STK_WAIT_FOR_FOURTH && sz >= 0 && !gr) {
      sk.task_phase = HRG_STK_RETREAT;
      sk.stack_ids[ns] = sz; nns = ns + 1;
    }
    sk.n_stack = nns;
    if (nns > sk.max_stack_height) sk.max_stack_height = nns;
    wave_sync();
  }
#elif HRG_HAMMER
  if (!d) write_obs(dm_, lane, goal, obs);   // the step's observation predates the transitions below (CollaborativeHammeringCart.step, 490-503)
  wave_sync();
  if (goal_reached && !m.done_at_success && !d) { // _on_goal_reached (749-767): next nail placement, board and nail reset, next animation
    const int ni = (hm.nail_index + 1) % m.n_obj_placements;
    const int ai = (s.anim_index + 1) % m.n_anim_ids, st = (int)((double)s.low_level_time / m.anim_step_length);
    wave_sync();
    hm.nail_index = ni;
    s.anim_index = ai; s.animation_time = 0; s.anim_start_time = st;
    hm.task_phase = HRG_HM_APPROACH; hm.n_delayed = 0;
    wave_sync();
    human_control(dm_, lane, gid);
    wave_sync();
    hm.task_phase = HRG_HM_APPROACH; hm.n_delayed = 0;   // _reset_animation (764-767, 770-774)
    hammer_take_board(dm_, lane, gid);
  }
  {
    const int ph = hm.task_phase;
    wave_sync();
    if (!d && ph == HRG_HM_PRESENT && hammered_in) hm.task_phase = HRG_HM_RETREAT;   // 496-501
    wave_sync();
  }
#else
  if (goal_reached && !d) { // reach_human_env.py:399-407 (a finished episode resamples its goals at reset anyway)
    s.goal_index = (s.goal_index + 1) % m.n_goals;
    goal_sample(dm_, lane, gid, s.goal_index);
  }
#endif
  STAMP(8);
  if (d) env_reset(dm_, lane, own_gid, obs);
#if !HRG_BOX && !HRG_STACK && !HRG_HAMMER
  else write_obs(dm_, lane, goal, obs);
#endif
  STAMP(9);
  STAMP_FINAL(lane);
  return (busy || !s.is_safe) && !d;   // a freshly reset env starts like any other
}

// ================================================================================================ kernels
#if HRG_STACK
#define hrg_step_kernel hrg_step_kernel_stack
#define hrg_reset_kernel hrg_reset_kernel_stack
typedef hrg_stack_state ObjState;   // the per-env object block this variant streams next to hrg_env_state
#elif HRG_HAMMER
#define hrg_step_kernel hrg_step_kernel_hammer
#define hrg_reset_kernel hrg_reset_kernel_hammer
typedef hrg_hammer_state ObjState;
#elif HRG_HULLS
#define hrg_step_kernel hrg_step_kernel_hull
#define hrg_reset_kernel hrg_reset_kernel_hull
typedef hrg_box_state ObjState;     // (ReachHuman streams no object block: the pointer is null)
#else
typedef hrg_box_state ObjState;
#endif
#if HRG_BOX && HRG_HANDOVER
#define hrg_step_kernel hrg_step_kernel_ho
#define hrg_reset_kernel hrg_reset_kernel_ho
#define hrg_box_launch_step hrg_ho_launch_step
#define hrg_box_launch_reset hrg_ho_launch_reset
#elif HRG_BOX && HRG_LIFT
#define hrg_step_kernel hrg_step_kernel_lift
#define hrg_reset_kernel hrg_reset_kernel_lift
#define hrg_box_launch_step hrg_lift_launch_step
#define hrg_box_launch_reset hrg_lift_launch_reset
#elif HRG_BOX
#define hrg_step_kernel hrg_step_kernel_box
#define hrg_reset_kernel hrg_reset_kernel_box
#endif
#if HRG_BOX
#ifndef HRG_BOX_WAVES
#define HRG_BOX_WAVES 3
#endif
#define HRG_KERNEL_WAVES HRG_BOX_WAVES   // the variant with the cube needs more registers and LDS
// the cube's state block: streamed like the env block
DI void box_load(const hrg_box_state* __restrict__ boxes, int e, int lane) {
  constexpr int NB = (int)(sizeof(hrg_box_state) / sizeof(double));
  const double* src = (const double*)(boxes + e);
  double* dst = (double*)&g_L.bx;
  if (lane < NB) dst[lane] = src[lane];
}
DI void box_store(hrg_box_state* __restrict__ boxes, int e, int lane) {
  constexpr int NB = (int)(sizeof(hrg_box_state) / sizeof(double));
  double* out = (double*)(boxes + e);
  const double* src = (const double*)&g_L.bx;
  if (lane < NB) out[lane] = src[lane];
}
#elif HRG_STACK
#ifndef HRG_STACK_WAVES
#define HRG_STACK_WAVES 2   // measured: 8.05 -> 7.45 ms per 4096-env step (mixed six-task batch 5.68 -> 5.33 ms), 768 B/lane of scratch
#endif
#define HRG_KERNEL_WAVES HRG_STACK_WAVES   // 28 KB of LDS per env (92 compact contact rows): 1 = 4 workgroups per CU, one wave per SIMD, up to 512 VGPRs; 2 = 256 registers, five per CU
DI void box_load(const hrg_stack_state* __restrict__ stacks, int e, int lane) {
  constexpr int NB = (int)(sizeof(hrg_stack_state) / sizeof(double));
  const double* src = (const double*)(stacks + e);
  double* dst = (double*)&g_L.sk;
  for (int k = lane; k < NB; k += 64) dst[k] = src[k];
}
DI void box_store(hrg_stack_state* __restrict__ stacks, int e, int lane) {
  constexpr int NB = (int)(sizeof(hrg_stack_state) / sizeof(double));
  double* out = (double*)(stacks + e);
  const double* src = (const double*)&g_L.sk;
  for (int k = lane; k < NB; k += 64) out[k] = src[k];
}
#elif HRG_HAMMER
#ifndef HRG_HAMMER_WAVES
#define HRG_HAMMER_WAVES 1   // the allocator's budget (512 registers); it uses 209 without a spill, so the hardware runs a second wave on a SIMD whenever LDS allows
#endif
#define HRG_KERNEL_WAVES HRG_HAMMER_WAVES   // 29.5 KB of LDS per env (65 dense rows of J over 24 DoF, the noslip pass's Gram matrix): 5 workgroups per CU
DI void box_load(const hrg_hammer_state* __restrict__ hammers, int e, int lane) {
  constexpr int NB = (int)(sizeof(hrg_hammer_state) / sizeof(double));
  const double* src = (const double*)(hammers + e);
  double* dst = (double*)&g_L.hm;
  for (int k = lane; k < NB; k += 64) dst[k] = src[k];
}
DI void box_store(hrg_hammer_state* __restrict__ hammers, int e, int lane) {
  constexpr int NB = (int)(sizeof(hrg_hammer_state) / sizeof(double));
  double* out = (double*)(hammers + e);
  const double* src = (const double*)&g_L.hm;
  for (int k = lane; k < NB; k += 64) out[k] = src[k];
}
#else
#define HRG_KERNEL_WAVES HRG_MIN_WAVES
#endif
__global__ __launch_bounds__(64 * HRG_WG_WAVES, HRG_KERNEL_WAVES) void hrg_step_kernel(const DevModel* __restrict__ dm, hrg_env_state* __restrict__ states, double* __restrict__ actions,
                                                     float* __restrict__ obs, float* __restrict__ term_obs, float* __restrict__ reward, uint8_t* __restrict__ done,
                                                     int32_t* __restrict__ info, double* dbg_r, double* dbg_h, int32_t* dbg_nh, int64_t env_id0, float* __restrict__ scratch_obs,
                                                     ObjState* __restrict__ boxes, int n_envs, StepOrder ord) {
  const int slot = hrg_env(), lane = hrg_lane();
  if (slot >= n_envs) return;   // a partly filled last workgroup (HRG_WG_WAVES > 1); the waves of a workgroup never wait for each other
  const int e = __builtin_amdgcn_readfirstlane(ord.buf[(size_t)ord.parity * ord.n + slot]);
  if ((unsigned)e >= (unsigned)n_envs) return;   // cannot happen while the two orders are permutations; never index the batch with anything else
  Lds& L = g_L;
  (void)boxes;  // the cube's state array: only the HRG_BOX variant streams it
  const double* src = (const double*)(states + e);
  double* dst = (double*)&L.st;
  constexpr int NW = (int)(sizeof(hrg_env_state) / sizeof(double));
  for (int k = lane; k < NW; k += 64) dst[k] = src[k];
#if HRG_BOX || HRG_STACK || HRG_HAMMER
  box_load(boxes, e, lane);
#endif
  wave_sync();
  float* tobs = term_obs ? term_obs + (size_t)e * HRG_OBS_DIM : scratch_obs + (size_t)e * HRG_OBS_DIM;
  const int busy = env_step(dm, lane, e, env_id0 + e, actions + (size_t)e * HRG_ACT_DIM, obs + (size_t)e * HRG_OBS_DIM, tobs, reward + e, done + e,
                            info + (size_t)e * HRG_INFO_DIM, dbg_r, dbg_h, dbg_nh, ord.fair);
  wave_sync();
  if (lane == 0) {   // this env's place in the next launch
    int32_t* ctr = ord.buf + 2 * (size_t)ord.n + 2 * (1 - ord.parity);
    const int at = busy ? atomicAdd(ctr, 1) : ord.n - 1 - atomicAdd(ctr + 1, 1);
    ord.buf[(size_t)(1 - ord.parity) * ord.n + at] = e;
    if (slot == 0) { int32_t* mine = ord.buf + 2 * (size_t)ord.n + 2 * ord.parity; mine[0] = 0; mine[1] = 0; }
  }
  hrg_env_state* st_out = states;
  asm volatile("" : "+s"(st_out));   // the write-back address is computed here, from the kernel argument, instead of sitting in two VGPRs across the whole step
  double* out = (double*)(st_out + e);
  for (int k = lane; k < NW; k += 64) out[k] = dst[k];
#if HRG_BOX || HRG_STACK || HRG_HAMMER
  box_store(boxes, e, lane);
#endif
}

__global__ __launch_bounds__(64 * HRG_WG_WAVES, HRG_KERNEL_WAVES) void hrg_reset_kernel(const DevModel* __restrict__ dm, hrg_env_state* __restrict__ states, const uint8_t* __restrict__ mask,
                                                      float* __restrict__ obs, int64_t env_id0, ObjState* __restrict__ boxes, int n_envs) {
  const int e = hrg_env(), lane = hrg_lane();
  if (e >= n_envs) return;
  Lds& L = g_L;
  (void)boxes;
  if (mask && !mask[e]) return;
  const double* src = (const double*)(states + e);
  double* dst = (double*)&L.st;
  constexpr int NW = (int)(sizeof(hrg_env_state) / sizeof(double));
  for (int k = lane; k < NW; k += 64) dst[k] = src[k];
  wave_sync();
  env_reset(dm, lane, env_id0 + e, obs ? obs + (size_t)e * HRG_OBS_DIM : nullptr);
  wave_sync();
  hrg_env_state* st_out = states;
  asm volatile("" : "+s"(st_out));   // the write-back address is computed here, from the kernel argument, instead of sitting in two VGPRs across the whole step
  double* out = (double*)(st_out + e);
  for (int k = lane; k < NW; k += 64) out[k] = dst[k];
#if HRG_BOX || HRG_STACK || HRG_HAMMER
  box_store(boxes, e, lane);
#endif
}

#if HRG_BASE_TU
// HumanEnv.check_collision_action for every env: goal configuration of the action at the env's current joint angles -> pre-check capsule model.
// The check reads the robot part of the state only, so one kernel serves every task.
__global__ __launch_bounds__(64 * HRG_WG_WAVES, HRG_KERNEL_WAVES) void hrg_check_kernel(const DevModel* __restrict__ dm_, const hrg_env_state* __restrict__ states, const double* __restrict__ actions,
                                                                          uint8_t* __restrict__ collides, int n_envs) {
  const int e = hrg_env(), lane = hrg_lane();
  if (e >= n_envs) return;
  Lds& L = g_L;
  const double* src = (const double*)(states + e);
  double* dst = (double*)&L.st;
  constexpr int NW = (int)(sizeof(hrg_env_state) / sizeof(double));
  for (int k = lane; k < NW; k += 64) dst[k] = src[k];
  if (lane < NV) L.act[lane] = lane < HRG_ACT_DIM ? actions[(size_t)e * HRG_ACT_DIM + lane] : 0.0;
  wave_sync();
  const ModelPtr dm = uniform_model(dm_);
  action_goal(dm, lane, L.act);
  const bool hit = config_collides(dm_, lane);
  if (lane == 0) collides[e] = hit ? 1 : 0;
}
#endif

// launch shims of the cube variant: defined by hrgym_box.hip (this file compiled with HRG_BOX=1), called by the host side below
extern "C" __attribute__((visibility("hidden"))) void hrg_box_launch_step(int n_envs, hipStream_t st, const DevModel* dm, hrg_env_state* states, double* actions, float* obs, float* term_obs,
                                                                           float* reward, uint8_t* done, int32_t* info, double* dbg_r, double* dbg_h, int32_t* dbg_nh, int64_t env_id0,
                                                                           float* scratch_obs, hrg_box_state* boxes, StepOrder ord);
extern "C" __attribute__((visibility("hidden"))) void hrg_box_launch_reset(int n_envs, hipStream_t st, const DevModel* dm, hrg_env_state* states, const uint8_t* mask, float* obs,
                                                                            int64_t env_id0, hrg_box_state* boxes);
#if HRG_BASE_TU
// ... of the stacking variant (hrgym_stack.hip)
extern "C" __attribute__((visibility("hidden"))) void hrg_stack_launch_step(int n_envs, hipStream_t st, const DevModel* dm, hrg_env_state* states, double* actions, float* obs, float* term_obs,
                                                                             float* reward, uint8_t* done, int32_t* info, double* dbg_r, double* dbg_h, int32_t* dbg_nh, int64_t env_id0,
                                                                             float* scratch_obs, hrg_stack_state* stacks, StepOrder ord);
extern "C" __attribute__((visibility("hidden"))) void hrg_stack_launch_reset(int n_envs, hipStream_t st, const DevModel* dm, hrg_env_state* states, const uint8_t* mask, float* obs,
                                                                              int64_t env_id0, hrg_stack_state* stacks);
#endif
#if HRG_BASE_TU
// ... of the hammering variant (hrgym_hammer.hip)
extern "C" __attribute__((visibility("hidden"))) void hrg_hammer_launch_step(int n_envs, hipStream_t st, const DevModel* dm, hrg_env_state* states, double* actions, float* obs, float* term_obs,
                                                                              float* reward, uint8_t* done, int32_t* info, double* dbg_r, double* dbg_h, int32_t* dbg_nh, int64_t env_id0,
                                                                              float* scratch_obs, hrg_hammer_state* hammers, StepOrder ord);
extern "C" __attribute__((visibility("hidden"))) void hrg_hammer_launch_reset(int n_envs, hipStream_t st, const DevModel* dm, hrg_env_state* states, const uint8_t* mask, float* obs,
                                                                               int64_t env_id0, hrg_hammer_state* hammers);
// the same shims of the handover variant (hrgym_handover.hip)
extern "C" __attribute__((visibility("hidden"))) void hrg_ho_launch_step(int n_envs, hipStream_t st, const DevModel* dm, hrg_env_state* states, double* actions, float* obs, float* term_obs,
                                                                          float* reward, uint8_t* done, int32_t* info, double* dbg_r, double* dbg_h, int32_t* dbg_nh, int64_t env_id0,
                                                                          float* scratch_obs, hrg_box_state* boxes, StepOrder ord);
extern "C" __attribute__((visibility("hidden"))) void hrg_ho_launch_reset(int n_envs, hipStream_t st, const DevModel* dm, hrg_env_state* states, const uint8_t* mask, float* obs,
                                                                           int64_t env_id0, hrg_box_state* boxes);
// ... and of the lifting variant (hrgym_lift.hip)
extern "C" __attribute__((visibility("hidden"))) void hrg_lift_launch_step(int n_envs, hipStream_t st, const DevModel* dm, hrg_env_state* states, double* actions, float* obs, float* term_obs,
                                                                            float* reward, uint8_t* done, int32_t* info, double* dbg_r, double* dbg_h, int32_t* dbg_nh, int64_t env_id0,
                                                                            float* scratch_obs, hrg_box_state* boxes, StepOrder ord);
extern "C" __attribute__((visibility("hidden"))) void hrg_lift_launch_reset(int n_envs, hipStream_t st, const DevModel* dm, hrg_env_state* states, const uint8_t* mask, float* obs,
                                                                             int64_t env_id0, hrg_box_state* boxes);
#endif
#if HRG_BASE_TU
// ... of the hull variant of the ReachHuman kernels (hrgym_hulls.hip)
extern "C" __attribute__((visibility("hidden"))) void hrg_hull_launch_step(int n_envs, hipStream_t st, const DevModel* dm, hrg_env_state* states, double* actions, float* obs, float* term_obs,
                                                                            float* reward, uint8_t* done, int32_t* info, double* dbg_r, double* dbg_h, int32_t* dbg_nh, int64_t env_id0,
                                                                            float* scratch_obs, StepOrder ord);
extern "C" __attribute__((visibility("hidden"))) void hrg_hull_launch_reset(int n_envs, hipStream_t st, const DevModel* dm, hrg_env_state* states, const uint8_t* mask, float* obs,
                                                                             int64_t env_id0);
#endif
#if HRG_HULLS
extern "C" void hrg_hull_launch_step(int n_envs, hipStream_t st, const DevModel* dm, hrg_env_state* states, double* actions, float* obs, float* term_obs, float* reward, uint8_t* done,
                                     int32_t* info, double* dbg_r, double* dbg_h, int32_t* dbg_nh, int64_t env_id0, float* scratch_obs, StepOrder ord) {
  hipLaunchKernelGGL(hrg_step_kernel, HRG_LAUNCH_DIMS(n_envs), 0, st, dm, states, actions, obs, term_obs, reward, done, info, dbg_r, dbg_h, dbg_nh, env_id0, scratch_obs, (ObjState*)nullptr, n_envs, ord);
}
extern "C" void hrg_hull_launch_reset(int n_envs, hipStream_t st, const DevModel* dm, hrg_env_state* states, const uint8_t* mask, float* obs, int64_t env_id0) {
  hipLaunchKernelGGL(hrg_reset_kernel, HRG_LAUNCH_DIMS(n_envs), 0, st, dm, states, mask, obs, env_id0, (ObjState*)nullptr, n_envs);
}
#endif
#if HRG_HAMMER
extern "C" void hrg_hammer_launch_step(int n_envs, hipStream_t st, const DevModel* dm, hrg_env_state* states, double* actions, float* obs, float* term_obs, float* reward, uint8_t* done,
                                       int32_t* info, double* dbg_r, double* dbg_h, int32_t* dbg_nh, int64_t env_id0, float* scratch_obs, hrg_hammer_state* hammers, StepOrder ord) {
  hipLaunchKernelGGL(hrg_step_kernel, HRG_LAUNCH_DIMS(n_envs), 0, st, dm, states, actions, obs, term_obs, reward, done, info, dbg_r, dbg_h, dbg_nh, env_id0, scratch_obs, hammers, n_envs, ord);
}
extern "C" void hrg_hammer_launch_reset(int n_envs, hipStream_t st, const DevModel* dm, hrg_env_state* states, const uint8_t* mask, float* obs, int64_t env_id0, hrg_hammer_state* hammers) {
  hipLaunchKernelGGL(hrg_reset_kernel, HRG_LAUNCH_DIMS(n_envs), 0, st, dm, states, mask, obs, env_id0, hammers, n_envs);
}
#endif
#if HRG_STACK
extern "C" void hrg_stack_launch_step(int n_envs, hipStream_t st, const DevModel* dm, hrg_env_state* states, double* actions, float* obs, float* term_obs, float* reward, uint8_t* done,
                                      int32_t* info, double* dbg_r, double* dbg_h, int32_t* dbg_nh, int64_t env_id0, float* scratch_obs, hrg_stack_state* stacks, StepOrder ord) {
  hipLaunchKernelGGL(hrg_step_kernel, HRG_LAUNCH_DIMS(n_envs), 0, st, dm, states, actions, obs, term_obs, reward, done, info, dbg_r, dbg_h, dbg_nh, env_id0, scratch_obs, stacks, n_envs, ord);
}
extern "C" void hrg_stack_launch_reset(int n_envs, hipStream_t st, const DevModel* dm, hrg_env_state* states, const uint8_t* mask, float* obs, int64_t env_id0, hrg_stack_state* stacks) {
  hipLaunchKernelGGL(hrg_reset_kernel, HRG_LAUNCH_DIMS(n_envs), 0, st, dm, states, mask, obs, env_id0, stacks, n_envs);
}
#endif
#if HRG_BOX
extern "C" void hrg_box_launch_step(int n_envs, hipStream_t st, const DevModel* dm, hrg_env_state* states, double* actions, float* obs, float* term_obs, float* reward, uint8_t* done,
                                    int32_t* info, double* dbg_r, double* dbg_h, int32_t* dbg_nh, int64_t env_id0, float* scratch_obs, hrg_box_state* boxes, StepOrder ord) {
  hipLaunchKernelGGL(hrg_step_kernel, HRG_LAUNCH_DIMS(n_envs), 0, st, dm, states, actions, obs, term_obs, reward, done, info, dbg_r, dbg_h, dbg_nh, env_id0, scratch_obs, boxes, n_envs, ord);
}
extern "C" void hrg_box_launch_reset(int n_envs, hipStream_t st, const DevModel* dm, hrg_env_state* states, const uint8_t* mask, float* obs, int64_t env_id0, hrg_box_state* boxes) {
  hipLaunchKernelGGL(hrg_reset_kernel, HRG_LAUNCH_DIMS(n_envs), 0, st, dm, states, mask, obs, env_id0, boxes, n_envs);
}
#endif

#ifdef HRG_STAMPS
#if HRG_STACK
#define hrg_debug_stamps hrg_debug_stamps_stack
#elif HRG_HAMMER
#define hrg_debug_stamps hrg_debug_stamps_hammer
#elif HRG_HANDOVER
#define hrg_debug_stamps hrg_debug_stamps_ho
#elif HRG_LIFT
#define hrg_debug_stamps hrg_debug_stamps_lift
#elif HRG_BOX
#define hrg_debug_stamps hrg_debug_stamps_box
#elif HRG_HULLS
#define hrg_debug_stamps hrg_debug_stamps_hull
#endif
#if HRG_BASE_TU
// waves that live longer than `thresh` shader cycles are also summed into a second set of accumulators: out[0..31] phase sums, out[32] their number, out[33] lifetime sum
extern "C" int hrg_debug_stamps_slow(double* out, unsigned long long thresh, int reset) {
  unsigned long long h[34];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps_slow), sizeof h) != hipSuccess) return -1;
  for (int i = 0; i < 34; i++) out[i] = (double)h[i];
  if (reset) { memset(h, 0, sizeof h); hipMemcpyToSymbol(HIP_SYMBOL(g_stamps_slow), h, sizeof h); }
  hipMemcpyToSymbol(HIP_SYMBOL(g_slow_thresh), &thresh, sizeof thresh);
  return 0;
}
#endif
extern "C" int hrg_debug_stamps(double* out, int reset) {
  unsigned long long h[32];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps), sizeof h) != hipSuccess) return -1;
  for (int i = 0; i < 32; i++) out[i] = (double)h[i];
  if (reset) { memset(h, 0, sizeof h); hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), h, sizeof h); }
  return 0;
}
#if HRG_HAMMER
#define hrg_debug_envacc hrg_debug_envacc_hammer
#define hrg_debug_envcyc hrg_debug_envcyc_hammer
#elif HRG_STACK
#define hrg_debug_envacc hrg_debug_envacc_stack
#define hrg_debug_envcyc hrg_debug_envcyc_stack
#elif HRG_HANDOVER
#define hrg_debug_envacc hrg_debug_envacc_ho
#define hrg_debug_envcyc hrg_debug_envcyc_ho
#elif HRG_LIFT
#define hrg_debug_envacc hrg_debug_envacc_lift
#define hrg_debug_envcyc hrg_debug_envcyc_lift
#elif HRG_BOX
#define hrg_debug_envacc hrg_debug_envacc_box
#define hrg_debug_envcyc hrg_debug_envcyc_box
#elif HRG_HULLS
#define hrg_debug_envacc hrg_debug_envacc_hull
#define hrg_debug_envcyc hrg_debug_envcyc_hull
#endif
#if 1
extern "C" int hrg_debug_envacc(unsigned long long* out, int n) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_envacc), sizeof(unsigned long long) * 32 * n) == hipSuccess ? 0 : -1; }
extern "C" int hrg_debug_envcyc(unsigned long long* out, int n) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_envcyc), sizeof(unsigned long long) * 3 * n) == hipSuccess ? 0 : -1; }
#endif
#endif

#if HRG_BASE_TU

// ================================================================================================ host side
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(x)                                                                                         \
  do {                                                                                                    \
    hipError_t _e = (x);                                                                                  \
    if (_e != hipSuccess) return fail(HRG_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(_e));        \
  } while (0)

struct hrg_batch {
  int device = 0;
  int32_t n_envs = 0;
  int64_t env_id0 = 0;
  DevModel* d_model = nullptr;
  double* d_frames = nullptr;
  double* d_hull = nullptr;            // hull vertices of the arm links (robot_hulls)
  bool hulls = false;                  // the hull variant of the ReachHuman kernels steps this batch (hrgym_hulls.hip)
  hrg_env_state* d_states = nullptr;
  double* d_rcaps = nullptr;
  double* d_hcaps = nullptr;
  int32_t* d_nh = nullptr;
  float* d_scratch_obs = nullptr;
  hrg_box_state* d_boxes = nullptr;   // the manipulation object of each env (PickPlaceHumanCart)
  hrg_stack_state* d_stacks = nullptr; // the four cubes of each env (CollaborativeStackingCart)
  hrg_hammer_state* d_hammers = nullptr; // board, hammer, nail of each env (CollaborativeHammeringCart)
  int32_t* d_order = nullptr;          // launch order of the step kernel (StepOrder): two orders of n_envs + two pairs of counters
  int32_t parity = 0;                  // which of the two orders the next step launch reads
  int32_t task = HRG_TASK_REACH;
  bool timing = false;
  bool taps = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
};

// the progress table of the level-waves priority (StepOrder.fair): one per device, shared by the batches and kernel variants that run on it; never freed
static int32_t* fair_table(int device) {
  static int32_t* tab[64] = {nullptr};
  if (device < 0 || device >= 64) return nullptr;
  if (!tab[device]) {
    int32_t* p = nullptr;
    if (hipMalloc(&p, sizeof(int32_t) * HRG_FAIR_SLOTS) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, sizeof(int32_t) * HRG_FAIR_SLOTS) != hipSuccess) { hipFree(p); return nullptr; }
    tab[device] = p;
  }
  return tab[device];
}

static void mat_from_quat(double* M, const double* q) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  M[0] = 1 - 2 * (y * y + z * z); M[1] = 2 * (x * y - w * z); M[2] = 2 * (x * z + w * y);
  M[3] = 2 * (x * y + w * z); M[4] = 1 - 2 * (x * x + z * z); M[5] = 2 * (y * z - w * x);
  M[6] = 2 * (x * z - w * y); M[7] = 2 * (y * z + w * x); M[8] = 1 - 2 * (x * x + y * y);
}

extern "C" {

const char* hrg_last_error(void) { return g_err.c_str(); }
const char* hrg_version(void) { return "hrgym-hip 0.1.0 (gfx950)"; }
size_t hrg_state_bytes(void) { return sizeof(hrg_env_state); }
size_t hrg_box_bytes(void) { return sizeof(hrg_box_state); }
size_t hrg_stack_bytes(void) { return sizeof(hrg_stack_state); }
size_t hrg_hammer_bytes(void) { return sizeof(hrg_hammer_state); }

int hrg_batch_create(const hrg_model_desc* desc, const hrg_clip_table* clips, int32_t n_envs, int64_t env_id0, int32_t device, hrg_batch** out) {
  if (!desc || !clips || !out || n_envs <= 0) return fail(HRG_ERR_INVALID, "null argument or n_envs <= 0");
  if (desc->failsafe_sdot != 0.0) return fail(HRG_ERR_INVALID, "failsafe_sdot must be 0: SSM / OFF brake to a stop, PFL computes its speed from pfl_v_safe / pfl_reach");
  if (desc->shield_type == HRG_SHIELD_PFL) {
    if (!(desc->pfl_v_safe > 0)) return fail(HRG_ERR_INVALID, "PFL: pfl_v_safe must be positive");
    for (int j = 0; j < NARM; j++) if (!(desc->pfl_reach[j] > 0)) return fail(HRG_ERR_INVALID, "PFL: pfl_reach must be positive");
  }
  for (int i = 0; i < NARM; i++)   // mj_Euler's implicit damping treats the arm's joint damping as a perturbation (euler_robot_tree): one Neumann term, error (h d / M)^2
    if (!(desc->jnt_damping[i] >= 0 && desc->timestep * desc->jnt_damping[i] * desc->dof_invweight0[i] < 1e-3))
      return fail(HRG_ERR_UNSUPPORTED, "arm joint damping too large for the implicit-damping expansion (h d / M must stay below 1e-3; robot.xml: 1e-4)");
  if ((desc->jnt_damping[NARM] > 0) != (desc->jnt_damping[NARM + 1] > 0)) return fail(HRG_ERR_UNSUPPORTED, "the two finger slides must both carry damping, or neither");
  if (desc->solimp[4] != 2.0) return fail(HRG_ERR_UNSUPPORTED, "solimp power must be 2 (MuJoCo's default): the HIP stepper evaluates the impedance sigmoid as a square");
  if (clips->n_clips < 1 || clips->n_clips > HRG_MAX_CLIPS || clips->n_clips != desc->n_clips) return fail(HRG_ERR_INVALID, "clip table / desc.n_clips mismatch");
  for (int i = 0; i < NV; i++) {
    if ((i < NARM) != (desc->jnt_type[i] == 0)) return fail(HRG_ERR_INVALID, "expected 6 hinges followed by 2 slides");
    if (i < NARM && desc->body_parent[i] != i - 1) return fail(HRG_ERR_INVALID, "arm must be a serial chain");
    if (i < NARM && !(desc->jnt_axis[i][0] == 0 && desc->jnt_axis[i][1] == 0 && desc->jnt_axis[i][2] == 1)) return fail(HRG_ERR_INVALID, "arm hinges must turn about the local z axis");
    if (i >= NARM && desc->body_parent[i] != NARM - 1) return fail(HRG_ERR_INVALID, "fingers must hang off the last link");
  }
  for (int c = 0; c < HRG_NSHIELD_RCAP; c++)
    if (desc->scap_body[c] != (c < NARM ? c : NARM - 1)) return fail(HRG_ERR_INVALID, "shield capsule c must sit on link c (gripper on link 6)");
  if (desc->n_bodypart > HRG_NBODYPART_MAX || desc->n_extremity > HRG_NEXTREMITY_MAX) return fail(HRG_ERR_INVALID, "too many body parts");
  if (desc->task < HRG_TASK_REACH || desc->task > HRG_TASK_HAMMERING) return fail(HRG_ERR_UNSUPPORTED, "unknown task");
  if (desc->task == HRG_TASK_STACKING) {
    if (!(desc->box_inertia[0] == desc->box_inertia[1] && desc->box_inertia[1] == desc->box_inertia[2]))
      return fail(HRG_ERR_UNSUPPORTED, "CollaborativeStackingCart: the HIP stepper stacks cubes (object_full_size with three equal edges)");
    for (int c = 0; c < clips->n_clips; c++) {
      const int32_t* kf = clips->clip_stack_keyframes[c];
      if (!(kf[0] >= 0 && kf[0] <= kf[1] && kf[1] <= kf[2] && kf[2] <= kf[3] && kf[3] <= kf[4] && clips->clip_n_loop[c] >= 0 && clips->clip_n_loop[c] <= HRG_MAX_LOOP &&
            clips->clip_n_loop2[c] >= 0 && clips->clip_n_loop2[c] <= HRG_MAX_LOOP))
        return fail(HRG_ERR_INVALID, "CollaborativeStackingCart: every clip needs five ascending keyframes and at most 4 loop sines per waiting phase in its info");
    }
  }
  if (desc->task == HRG_TASK_LIFTING && !(desc->min_balance > -1 && desc->min_balance < 1)) return fail(HRG_ERR_INVALID, "CollaborativeLiftingCart: min_balance must lie in (-1, 1)");
  if (HRG_IS_HANDOVER(desc->task))
    for (int c = 0; c < clips->n_clips; c++)
      if (!(clips->clip_n_loop2[c] >= 0 && clips->clip_n_loop2[c] <= HRG_MAX_LOOP)) return fail(HRG_ERR_INVALID, "HumanRobotHandoverCart: at most 4 loop sines per stage");
  if (desc->task == HRG_TASK_INSPECTION || HRG_IS_HANDOVER(desc->task))
    for (int c = 0; c < clips->n_clips; c++)
      if (!(clips->clip_n_loop[c] >= 0 && clips->clip_n_loop[c] <= HRG_MAX_LOOP && clips->clip_keyframes[c][0] >= 0 && clips->clip_keyframes[c][0] <= clips->clip_keyframes[c][1]))
        return fail(HRG_ERR_INVALID, "HumanObjectInspectionCart: every clip needs keyframes (k0 <= k1) and at most 4 loop sines in its info");
  if (desc->ik_enabled && !(desc->ik_max_iter >= 1 && desc->ik_max_iter <= 1000 && desc->ik_damping > 0 && desc->ik_action_limit > 0 && desc->ik_residual_threshold >= 0))
    return fail(HRG_ERR_INVALID, "ik: need 1 <= max_iter <= 1000, damping > 0, action_limit > 0, residual_threshold >= 0");
  if (desc->task == HRG_TASK_HAMMERING) {
    if (!(desc->hm_board_mass > 0 && desc->hm_hammer_mass > 0 && desc->hm_nail_mass > 0 && desc->hm_nail_range > 0 && desc->hm_nail_invweight > 0 && desc->n_obj_placements > 0 &&
          desc->hm_board_inertia[0] > 0 && desc->hm_board_inertia[1] > 0 && desc->hm_board_inertia[2] > 0 && desc->hm_hammer_inertia[0] > 0 && desc->hm_hammer_inertia[1] > 0 &&
          desc->hm_hammer_inertia[2] > 0))
      return fail(HRG_ERR_INVALID, "CollaborativeHammeringCart needs positive board / hammer / nail masses and inertias, a nail range and n_obj_placements > 0");
    for (int c = 0; c < clips->n_clips; c++)
      if (!(clips->clip_keyframes[c][0] >= 0 && clips->clip_keyframes[c][0] <= clips->clip_keyframes[c][1] && clips->clip_n_loop[c] >= 0 && clips->clip_n_loop[c] <= HRG_MAX_LOOP))
        return fail(HRG_ERR_INVALID, "CollaborativeHammeringCart: every clip needs two ascending keyframes and at most 4 loop sines in its info");
  } else
  if (desc->task != HRG_TASK_REACH && !(desc->box_half[0] > 0 && desc->box_half[1] > 0 && desc->box_half[2] > 0 && desc->box_mass > 0 && desc->box_inertia[0] > 0 && desc->box_inertia[1] > 0 &&
                                       desc->box_inertia[2] > 0 && desc->box_inertia_mean > 0 && desc->box_invweight_rot > 0 && desc->n_targets > 0 && desc->n_obj_placements > 0))
    return fail(HRG_ERR_INVALID, "PickPlaceHumanCart needs box_half, box_mass, box_inertia, n_targets, n_obj_placements > 0");
  HIPCHK(hipSetDevice(device));
  hrg_batch* b = new hrg_batch();
  b->device = device; b->n_envs = n_envs; b->env_id0 = env_id0; b->task = desc->task;
  // ---- device model ----
  std::unique_ptr<DevModel> hm_own(new DevModel());
  DevModel* hm = hm_own.get();
  memset(hm, 0, sizeof *hm);
  // every early return below releases the batch and whatever device buffers it already owns
  auto bail = [&](int code, const std::string& msg) { hrg_batch_destroy(b); return fail(code, msg); };
#define HIPCHK_C(x)                                                                                       \
  do {                                                                                                    \
    hipError_t _e = (x);                                                                                  \
    if (_e != hipSuccess) return bail(HRG_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(_e));        \
  } while (0)
  hm->m = *desc;
  mat_from_quat(hm->Rbase, desc->base_quat);
  for (int i = 0; i < NV; i++) {
    mat_from_quat(hm->Rq[i], desc->body_quat[i]);
    int mask = 0;
    for (int k = i; k >= 0; k = desc->body_parent[k]) mask |= 1 << k;
    hm->anc_mask[i] = mask;
  }
  int maxd = 0;
  for (int i = 0; i < HRG_NHB; i++) maxd = desc->hb_depth[i] > maxd ? desc->hb_depth[i] : maxd;
  hm->hb_maxdepth = maxd;
  { // mj_makeImpedance constants of solref (the oracle's `impedance`)
    double tc = desc->solref[0];
    const double dr = desc->solref[1], dmax = desc->solimp[1];
    if (tc < 2 * desc->timestep) tc = 2 * desc->timestep;
    hm->sol_K = 1.0 / (dmax * dmax * tc * tc * dr * dr);
    hm->sol_Bd = 2.0 / (dmax * tc);
  }
  for (int c = 0; c < HRG_NRCAP; c++) {
    double d2 = 0;
    for (int a = 0; a < 3; a++) d2 += (desc->rcap_p2[c][a] - desc->rcap_p1[c][a]) * (desc->rcap_p2[c][a] - desc->rcap_p1[c][a]);
    hm->rcap_hl[c] = 0.5 * sqrt(d2);
  }
  for (int c = 0; c < HRG_NHB; c++) {
    double d2 = 0;
    for (int a = 0; a < 3; a++) d2 += (desc->hcap_p2[c][a] - desc->hcap_p1[c][a]) * (desc->hcap_p2[c][a] - desc->hcap_p1[c][a]);
    hm->hcap_hl[c] = 0.5 * sqrt(d2);
  }
  if (maxd + 1 > 16) { return bail(HRG_ERR_INVALID, "human tree deeper than 15"); }
  for (int i = 0; i < HRG_NHB; i++) hm->hb_jump[0][i] = desc->hb_parent[i];
  for (int s = 1; s < 4; s++)
    for (int i = 0; i < HRG_NHB; i++) { const int a = hm->hb_jump[s - 1][i]; hm->hb_jump[s][i] = a < 0 ? -1 : hm->hb_jump[s - 1][a]; }
  hm->hb_njump = 0;
  while ((1 << hm->hb_njump) < maxd + 1) hm->hb_njump++;
  int n = 0;
  for (int k = 0; k < desc->n_bodypart; k++, n++) { hm->hc_kind[n] = 0; hm->hc_j1[n] = desc->bp_joint[k][0]; hm->hc_j2[n] = desc->bp_joint[k][1]; hm->hc_th[n] = desc->bp_thickness[k]; hm->hc_a[n] = desc->bp_amax[k]; hm->hc_v[n] = desc->bp_vmax[k]; }
  for (int k = 0; k < desc->n_bodypart; k++, n++) { hm->hc_kind[n] = 1; hm->hc_j1[n] = desc->bp_joint[k][0]; hm->hc_j2[n] = desc->bp_joint[k][1]; hm->hc_th[n] = desc->bp_thickness[k]; hm->hc_v[n] = desc->bp_vmax[k]; }
  for (int k = 0; k < desc->n_extremity; k++, n++) { hm->hc_kind[n] = 2; hm->hc_j1[n] = desc->ext_joint[k]; hm->hc_j2[n] = desc->ext_joint[k]; hm->hc_th[n] = desc->ext_thickness[k]; hm->hc_v[n] = desc->ext_vmax[k]; hm->hc_len[n] = desc->ext_length[k]; }
  for (int k = 0; k < desc->n_bodypart; k++) {
    if (!desc->bp_in_pos[k]) continue;
    if (n >= HRG_NHCAP_MAX) { return bail(HRG_ERR_INVALID, "more than 64 human reach capsules"); }
    hm->hc_kind[n] = 3; hm->hc_j1[n] = desc->bp_joint[k][0]; hm->hc_j2[n] = desc->bp_joint[k][1]; hm->hc_th[n] = desc->bp_thickness[k]; hm->hc_v[n] = desc->bp_vmax[k];
    n++;
  }
  if (n > HRG_NHCAP_MAX) { return bail(HRG_ERR_INVALID, "more than 64 human reach capsules"); }
  hm->hc_n = n;
  int ns = 0;
  for (int i = 0; i < HRG_NRCAP; i++)
    for (int j = i + 1; j < HRG_NRCAP; j++)
      if ((desc->rcap_selfmask[i] >> j) & 1u) { hm->self_i[ns] = i; hm->self_j[ns] = j; ns++; }
  hm->n_self = ns;
  int nc = 0;
  for (int i = 0; i < 8; i++)
    for (int j = i + 1; j < 8; j++)
      if (((desc->chk_selfmask[i] >> j) & 1u) && nc < 32) { hm->chk_i[nc] = i; hm->chk_j[nc] = j; nc++; }
  hm->n_chk = nc;
  hm->phase_mask = 0xff;
  {
    double se, ve, ae;
    path_plan(&hm->brake_full, 0.0, 1.0, 0.0, desc->failsafe_sdot, desc->path_amax, desc->path_jmax);
    hm->brake_T = path_total(&hm->brake_full);
    path_eval(&hm->brake_full, hm->brake_T, desc->failsafe_sdot, &se, &ve, &ae);
    hm->brake_ds = se;
  }
#ifdef HRG_STAMPS
  if (const char* pm = getenv("HRG_PHASE_MASK")) hm->phase_mask = atoi(pm); // diagnostic build only: results are invalid
#endif
  // clips
  const size_t fbytes = sizeof(double) * HRG_FRAME_DIM * (size_t)clips->total_frames;
  int64_t tot = 0;
  for (int c = 0; c < clips->n_clips; c++) {
    if (clips->clip_len[c] < 1 || clips->clip_offset[c] != tot) { return bail(HRG_ERR_INVALID, "clip table must be densely packed"); }
    tot += clips->clip_len[c];
  }
  if (tot != clips->total_frames) { return bail(HRG_ERR_INVALID, "clip table total_frames mismatch"); }
  HIPCHK_C(hipMalloc(&b->d_frames, fbytes));
  HIPCHK_C(hipMemcpy(b->d_frames, clips->frames, fbytes, hipMemcpyHostToDevice));
  hm->clips = *clips;
  hm->clips.frames = b->d_frames;
  hm->hull_dev = nullptr;
  if (desc->robot_hulls) {   // convex hulls of the arm links: the vertex table goes to device memory like the clip frames
    if (desc->task != HRG_TASK_REACH) return bail(HRG_ERR_UNSUPPORTED, "robot_hulls: the hull variant of the step kernel exists for ReachHuman (the lean model) so far");
    b->hulls = true;
    if (!desc->hull_verts || desc->hull_off[0] != 0) return bail(HRG_ERR_INVALID, "robot_hulls: hull_verts / hull_off missing");
    for (int h = 0; h < HRG_NHULL; h++)
      if (!(desc->hull_off[h + 1] > desc->hull_off[h] + 3 && desc->hull_off[h + 1] <= 1000000)) return bail(HRG_ERR_INVALID, "robot_hulls: every hull needs at least four vertices");
    const size_t hbytes = sizeof(double) * 3 * (size_t)desc->hull_off[HRG_NHULL];
    HIPCHK_C(hipMalloc(&b->d_hull, hbytes));
    HIPCHK_C(hipMemcpy(b->d_hull, desc->hull_verts, hbytes, hipMemcpyHostToDevice));
    hm->hull_dev = b->d_hull;
  }
  HIPCHK_C(hipMalloc(&b->d_model, sizeof(DevModel)));
  HIPCHK_C(hipMemcpy(b->d_model, hm, sizeof(DevModel), hipMemcpyHostToDevice));
  // ---- state ----
  HIPCHK_C(hipMalloc(&b->d_states, sizeof(hrg_env_state) * (size_t)n_envs));
  std::vector<hrg_env_state> init((size_t)n_envs);
  memset(init.data(), 0, sizeof(hrg_env_state) * (size_t)n_envs);
  for (auto& s : init) s.episode = -1;
  HIPCHK_C(hipMemcpy(b->d_states, init.data(), sizeof(hrg_env_state) * (size_t)n_envs, hipMemcpyHostToDevice));
  HIPCHK_C(hipMalloc(&b->d_rcaps, sizeof(double) * 7 * HRG_NSHIELD_RCAP * (size_t)n_envs));
  HIPCHK_C(hipMalloc(&b->d_hcaps, sizeof(double) * 7 * HRG_NHCAP_MAX * (size_t)n_envs));
  HIPCHK_C(hipMalloc(&b->d_nh, sizeof(int32_t) * (size_t)n_envs));
  HIPCHK_C(hipMalloc(&b->d_scratch_obs, sizeof(float) * HRG_OBS_DIM * (size_t)n_envs));
  HIPCHK_C(hipMemset(b->d_rcaps, 0, sizeof(double) * 7 * HRG_NSHIELD_RCAP * (size_t)n_envs));
  HIPCHK_C(hipMemset(b->d_hcaps, 0, sizeof(double) * 7 * HRG_NHCAP_MAX * (size_t)n_envs));
  HIPCHK_C(hipMemset(b->d_nh, 0, sizeof(int32_t) * (size_t)n_envs));
  HIPCHK_C(hipMalloc(&b->d_boxes, sizeof(hrg_box_state) * (size_t)n_envs));
  HIPCHK_C(hipMemset(b->d_boxes, 0, sizeof(hrg_box_state) * (size_t)n_envs));
  if (desc->task == HRG_TASK_HAMMERING) {
    HIPCHK_C(hipMalloc(&b->d_hammers, sizeof(hrg_hammer_state) * (size_t)n_envs));
    HIPCHK_C(hipMemset(b->d_hammers, 0, sizeof(hrg_hammer_state) * (size_t)n_envs));
  }
  if (desc->task == HRG_TASK_STACKING) {
    HIPCHK_C(hipMalloc(&b->d_stacks, sizeof(hrg_stack_state) * (size_t)n_envs));
    HIPCHK_C(hipMemset(b->d_stacks, 0, sizeof(hrg_stack_state) * (size_t)n_envs));
  }
  {
    std::vector<int32_t> ord(2 * (size_t)n_envs + 4, 0);
    for (int32_t e = 0; e < n_envs; e++) ord[e] = ord[(size_t)n_envs + e] = e;
    HIPCHK_C(hipMalloc(&b->d_order, sizeof(int32_t) * ord.size()));
    HIPCHK_C(hipMemcpy(b->d_order, ord.data(), sizeof(int32_t) * ord.size(), hipMemcpyHostToDevice));
  }
#undef HIPCHK_C
  *out = b;
  return HRG_OK;
}

void hrg_batch_destroy(hrg_batch* b) {
  if (!b) return;
  hipSetDevice(b->device);
  hipDeviceSynchronize();
  for (auto& p : b->events) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
  for (auto& p : b->pool) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
  hipFree(b->d_model); hipFree(b->d_frames); hipFree(b->d_hull); hipFree(b->d_states); hipFree(b->d_rcaps); hipFree(b->d_hcaps); hipFree(b->d_nh); hipFree(b->d_scratch_obs); hipFree(b->d_boxes); hipFree(b->d_stacks); hipFree(b->d_hammers); hipFree(b->d_order);
  delete b;
}

int hrg_batch_reset(hrg_batch* b, const uint8_t* mask_dev, float* obs_dev, void* stream) {
  if (!b) return fail(HRG_ERR_INVALID, "null batch");
  HIPCHK(hipSetDevice(b->device));
  if (b->task == HRG_TASK_HAMMERING) hrg_hammer_launch_reset(b->n_envs, (hipStream_t)stream, b->d_model, b->d_states, mask_dev, obs_dev, b->env_id0, b->d_hammers);
  else if (b->task == HRG_TASK_STACKING) hrg_stack_launch_reset(b->n_envs, (hipStream_t)stream, b->d_model, b->d_states, mask_dev, obs_dev, b->env_id0, b->d_stacks);
  else if (b->task == HRG_TASK_LIFTING) hrg_lift_launch_reset(b->n_envs, (hipStream_t)stream, b->d_model, b->d_states, mask_dev, obs_dev, b->env_id0, b->d_boxes);
  else if (HRG_IS_HANDOVER(b->task)) hrg_ho_launch_reset(b->n_envs, (hipStream_t)stream, b->d_model, b->d_states, mask_dev, obs_dev, b->env_id0, b->d_boxes);
  else if (b->task != HRG_TASK_REACH) hrg_box_launch_reset(b->n_envs, (hipStream_t)stream, b->d_model, b->d_states, mask_dev, obs_dev, b->env_id0, b->d_boxes);
  else if (b->hulls) hrg_hull_launch_reset(b->n_envs, (hipStream_t)stream, b->d_model, b->d_states, mask_dev, obs_dev, b->env_id0);
  else hipLaunchKernelGGL(hrg_reset_kernel, HRG_LAUNCH_DIMS(b->n_envs), 0, (hipStream_t)stream, b->d_model, b->d_states, mask_dev, obs_dev, b->env_id0, b->d_boxes, b->n_envs);
  HIPCHK(hipGetLastError());
  return HRG_OK;
}

int hrg_batch_check_actions(hrg_batch* b, const double* actions_dev, uint8_t* collides_dev, void* stream) {
  if (!b || !actions_dev || !collides_dev) return fail(HRG_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(b->device));
  hipLaunchKernelGGL(hrg_check_kernel, HRG_LAUNCH_DIMS(b->n_envs), 0, (hipStream_t)stream, b->d_model, b->d_states, actions_dev, collides_dev, b->n_envs);
  HIPCHK(hipGetLastError());
  return HRG_OK;
}

int hrg_batch_step(hrg_batch* b, double* actions_dev, float* obs_dev, float* term_obs_dev, float* reward_dev, uint8_t* done_dev, int32_t* info_dev, void* stream) {
  if (!b || !actions_dev || !obs_dev || !reward_dev || !done_dev || !info_dev) return fail(HRG_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(b->device));
  hipStream_t st = (hipStream_t)stream;
  std::pair<hipEvent_t, hipEvent_t> ev;
  if (b->timing) {
    if (!b->pool.empty()) { ev = b->pool.back(); b->pool.pop_back(); }
    else { HIPCHK(hipEventCreate(&ev.first)); HIPCHK(hipEventCreate(&ev.second)); }
    HIPCHK(hipEventRecord(ev.first, st));
  }
  int32_t* fair = fair_table(b->device);
  if (!fair) return fail(HRG_ERR_HIP, "cannot allocate the wave-progress table");
  const StepOrder ord{b->d_order, b->n_envs, b->parity, fair};
  if (b->task == HRG_TASK_HAMMERING)
    hrg_hammer_launch_step(b->n_envs, st, b->d_model, b->d_states, actions_dev, obs_dev, term_obs_dev, reward_dev, done_dev, info_dev,
                           b->taps ? b->d_rcaps : nullptr, b->taps ? b->d_hcaps : nullptr, b->taps ? b->d_nh : nullptr, b->env_id0, b->d_scratch_obs, b->d_hammers, ord);
  else if (b->task == HRG_TASK_STACKING)
    hrg_stack_launch_step(b->n_envs, st, b->d_model, b->d_states, actions_dev, obs_dev, term_obs_dev, reward_dev, done_dev, info_dev,
                          b->taps ? b->d_rcaps : nullptr, b->taps ? b->d_hcaps : nullptr, b->taps ? b->d_nh : nullptr, b->env_id0, b->d_scratch_obs, b->d_stacks, ord);
  else if (b->task == HRG_TASK_LIFTING)
    hrg_lift_launch_step(b->n_envs, st, b->d_model, b->d_states, actions_dev, obs_dev, term_obs_dev, reward_dev, done_dev, info_dev,
                         b->taps ? b->d_rcaps : nullptr, b->taps ? b->d_hcaps : nullptr, b->taps ? b->d_nh : nullptr, b->env_id0, b->d_scratch_obs, b->d_boxes, ord);
  else if (HRG_IS_HANDOVER(b->task))
    hrg_ho_launch_step(b->n_envs, st, b->d_model, b->d_states, actions_dev, obs_dev, term_obs_dev, reward_dev, done_dev, info_dev,
                       b->taps ? b->d_rcaps : nullptr, b->taps ? b->d_hcaps : nullptr, b->taps ? b->d_nh : nullptr, b->env_id0, b->d_scratch_obs, b->d_boxes, ord);
  else if (b->task != HRG_TASK_REACH)
    hrg_box_launch_step(b->n_envs, st, b->d_model, b->d_states, actions_dev, obs_dev, term_obs_dev, reward_dev, done_dev, info_dev,
                        b->taps ? b->d_rcaps : nullptr, b->taps ? b->d_hcaps : nullptr, b->taps ? b->d_nh : nullptr, b->env_id0, b->d_scratch_obs, b->d_boxes, ord);
  else if (b->hulls)
    hrg_hull_launch_step(b->n_envs, st, b->d_model, b->d_states, actions_dev, obs_dev, term_obs_dev, reward_dev, done_dev, info_dev,
                         b->taps ? b->d_rcaps : nullptr, b->taps ? b->d_hcaps : nullptr, b->taps ? b->d_nh : nullptr, b->env_id0, b->d_scratch_obs, ord);
  else
    hipLaunchKernelGGL(hrg_step_kernel, HRG_LAUNCH_DIMS(b->n_envs), 0, st, b->d_model, b->d_states, actions_dev, obs_dev, term_obs_dev, reward_dev, done_dev, info_dev,
                       b->taps ? b->d_rcaps : nullptr, b->taps ? b->d_hcaps : nullptr, b->taps ? b->d_nh : nullptr, b->env_id0, b->d_scratch_obs, b->d_boxes, b->n_envs, ord);
  HIPCHK(hipGetLastError());
  b->parity ^= 1;
  if (b->timing) { HIPCHK(hipEventRecord(ev.second, st)); b->events.push_back(ev); }
  return HRG_OK;
}

int hrg_batch_launch_order(hrg_batch* b, int32_t* order_host, int32_t* n_busy_host) {
  if (!b || !order_host || !n_busy_host) return fail(HRG_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  const size_t n = (size_t)b->n_envs;
  HIPCHK(hipMemcpy(order_host, b->d_order + (size_t)b->parity * n, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(n_busy_host, b->d_order + 2 * n + 2 * (size_t)b->parity, sizeof(int32_t), hipMemcpyDeviceToHost));   // the front counter that filled this order
  return HRG_OK;
}

int hrg_batch_contacts(hrg_batch* b, int32_t* pairs_host, int32_t* ncon_host) {
  if (!b) return fail(HRG_ERR_INVALID, "null batch");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  std::vector<hrg_env_state> st((size_t)b->n_envs);
  HIPCHK(hipMemcpy(st.data(), b->d_states, sizeof(hrg_env_state) * (size_t)b->n_envs, hipMemcpyDeviceToHost));
  for (int e = 0; e < b->n_envs; e++) {
    ncon_host[e] = st[e].ncon;
    memcpy(pairs_host + (size_t)e * HRG_NCON_MAX * 2, st[e].con_pairs, sizeof(int32_t) * HRG_NCON_MAX * 2);
  }
  return HRG_OK;
}

int hrg_batch_capsules(hrg_batch* b, double* robot_host, double* human_host, int32_t* n_human_host) {
  if (!b) return fail(HRG_ERR_INVALID, "null batch");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(robot_host, b->d_rcaps, sizeof(double) * 7 * HRG_NSHIELD_RCAP * (size_t)b->n_envs, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(human_host, b->d_hcaps, sizeof(double) * 7 * HRG_NHCAP_MAX * (size_t)b->n_envs, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(n_human_host, b->d_nh, sizeof(int32_t) * (size_t)b->n_envs, hipMemcpyDeviceToHost));
  return HRG_OK;
}

int hrg_batch_get_state(hrg_batch* b, int32_t env, void* buf_host, size_t bytes) {
  if (!b || env < 0 || env >= b->n_envs || bytes != sizeof(hrg_env_state)) return fail(HRG_ERR_INVALID, "bad env index or buffer size");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(buf_host, b->d_states + env, bytes, hipMemcpyDeviceToHost));
  return HRG_OK;
}

int hrg_batch_set_state(hrg_batch* b, int32_t env, const void* buf_host, size_t bytes) {
  if (!b || env < 0 || env >= b->n_envs || bytes != sizeof(hrg_env_state)) return fail(HRG_ERR_INVALID, "bad env index or buffer size");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(b->d_states + env, buf_host, bytes, hipMemcpyHostToDevice));
  return HRG_OK;
}

int hrg_batch_get_box(hrg_batch* b, int32_t env, void* buf_host, size_t bytes) {
  if (!b || env < 0 || env >= b->n_envs || bytes != sizeof(hrg_box_state)) return fail(HRG_ERR_INVALID, "bad env index or buffer size");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(buf_host, b->d_boxes + env, bytes, hipMemcpyDeviceToHost));
  return HRG_OK;
}

int hrg_batch_set_box(hrg_batch* b, int32_t env, const void* buf_host, size_t bytes) {
  if (!b || env < 0 || env >= b->n_envs || bytes != sizeof(hrg_box_state)) return fail(HRG_ERR_INVALID, "bad env index or buffer size");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(b->d_boxes + env, buf_host, bytes, hipMemcpyHostToDevice));
  return HRG_OK;
}

int hrg_batch_get_stack(hrg_batch* b, int32_t env, void* buf_host, size_t bytes) {
  if (!b || env < 0 || env >= b->n_envs || bytes != sizeof(hrg_stack_state)) return fail(HRG_ERR_INVALID, "bad env index or buffer size");
  if (!b->d_stacks) return fail(HRG_ERR_INVALID, "not a CollaborativeStackingCart batch");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(buf_host, b->d_stacks + env, bytes, hipMemcpyDeviceToHost));
  return HRG_OK;
}

int hrg_batch_set_stack(hrg_batch* b, int32_t env, const void* buf_host, size_t bytes) {
  if (!b || env < 0 || env >= b->n_envs || bytes != sizeof(hrg_stack_state)) return fail(HRG_ERR_INVALID, "bad env index or buffer size");
  if (!b->d_stacks) return fail(HRG_ERR_INVALID, "not a CollaborativeStackingCart batch");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(b->d_stacks + env, buf_host, bytes, hipMemcpyHostToDevice));
  return HRG_OK;
}

int hrg_batch_get_hammer(hrg_batch* b, int32_t env, void* buf_host, size_t bytes) {
  if (!b || env < 0 || env >= b->n_envs || bytes != sizeof(hrg_hammer_state)) return fail(HRG_ERR_INVALID, "bad env index or buffer size");
  if (!b->d_hammers) return fail(HRG_ERR_INVALID, "not a CollaborativeHammeringCart batch");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(buf_host, b->d_hammers + env, bytes, hipMemcpyDeviceToHost));
  return HRG_OK;
}

int hrg_batch_set_hammer(hrg_batch* b, int32_t env, const void* buf_host, size_t bytes) {
  if (!b || env < 0 || env >= b->n_envs || bytes != sizeof(hrg_hammer_state)) return fail(HRG_ERR_INVALID, "bad env index or buffer size");
  if (!b->d_hammers) return fail(HRG_ERR_INVALID, "not a CollaborativeHammeringCart batch");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(b->d_hammers + env, buf_host, bytes, hipMemcpyHostToDevice));
  return HRG_OK;
}

int hrg_batch_get_states(hrg_batch* b, const int32_t* envs_host, int32_t n, void* states_host, void* boxes_host) {
  if (!b || !envs_host || !states_host || n < 0) return fail(HRG_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  for (int32_t k = 0; k < n; k++) {
    const int32_t e = envs_host[k];
    if (e < 0 || e >= b->n_envs) return fail(HRG_ERR_INVALID, "bad env index");
    HIPCHK(hipMemcpyAsync((hrg_env_state*)states_host + k, b->d_states + e, sizeof(hrg_env_state), hipMemcpyDeviceToHost, 0));
    if (boxes_host) HIPCHK(hipMemcpyAsync((hrg_box_state*)boxes_host + k, b->d_boxes + e, sizeof(hrg_box_state), hipMemcpyDeviceToHost, 0));
  }
  HIPCHK(hipDeviceSynchronize());
  return HRG_OK;
}

int hrg_batch_set_states(hrg_batch* b, const int32_t* envs_host, int32_t n, const void* states_host, const void* boxes_host) {
  if (!b || !envs_host || !states_host || n < 0) return fail(HRG_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  for (int32_t k = 0; k < n; k++) {
    const int32_t e = envs_host[k];
    if (e < 0 || e >= b->n_envs) return fail(HRG_ERR_INVALID, "bad env index");
    HIPCHK(hipMemcpyAsync(b->d_states + e, (const hrg_env_state*)states_host + k, sizeof(hrg_env_state), hipMemcpyHostToDevice, 0));
    if (boxes_host) HIPCHK(hipMemcpyAsync(b->d_boxes + e, (const hrg_box_state*)boxes_host + k, sizeof(hrg_box_state), hipMemcpyHostToDevice, 0));
  }
  HIPCHK(hipDeviceSynchronize());
  return HRG_OK;
}

int hrg_batch_enable_taps(hrg_batch* b, int32_t on) {
  if (!b) return fail(HRG_ERR_INVALID, "null batch");
  b->taps = on != 0;
  return HRG_OK;
}


int hrg_batch_kernel_time(hrg_batch* b, double* avg_ms, int64_t* n_launches) {
  if (!b) return fail(HRG_ERR_INVALID, "null batch");
  HIPCHK(hipSetDevice(b->device));
  double tot = 0;
  int64_t n = 0;
  for (auto& p : b->events) {
    HIPCHK(hipEventSynchronize(p.second));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, p.first, p.second));
    tot += ms; n++;
    b->pool.push_back(p);
  }
  b->events.clear();
  b->timing = true; // first call arms the timer
  if (avg_ms) *avg_ms = n ? tot / (double)n : 0.0;
  if (n_launches) *n_launches = n;
  return HRG_OK;
}

} // extern "C"
#endif // HRG_BASE_TU
