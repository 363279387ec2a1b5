// hrgym_kernels.h — the per-environment step: phases of one shield cycle and the env epilogue.
// Reference call sites are cited per phase (paths relative to the reference root, human_robot_gym/...).
#pragma once
#include "hrgym_device.h"
#if HRG_HULLS
#include "hrgym_hull.h"
#endif

// ================================================================================================ robot tree
// Chain kinematics for THREE configurations at once (lanes 0,1,2): 0 = shield's current commanded motion,
// 1 = configuration at the end of the fail-safe brake (both feed RobotReach), 2 = simulation state
// (mj_kinematics of sim.forward(), environments/manipulation/human_env.py:504).
DI void robot_chain_fk(const DevModel* __restrict__ dm_, int lane, bool shield_on) {
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  // the 18 joint angles (3 configurations x 6 hinges) take their sine and cosine on 18 lanes at once; the walk along the chain below reads them back (scratch:
  // the robot reach capsules L.rc, which the shield fills only after this call)
  double* sc = &L.rc[0][0];
  if (lane < 3 * NARM && (lane >= 2 * NARM || shield_on)) {
    const int cfg = lane / NARM, i = lane - NARM * cfg;
    const double q = cfg == 0 ? L.cq[i] : (cfg == 1 ? L.qe[i] : L.st.qpos[i]);
    double sn, cs;
    sincos_small(q, &sn, &cs);
    sc[2 * lane] = sn; sc[2 * lane + 1] = cs;
  }
  wave_sync();
  // lanes = (configuration, row): row r of a product R A is (row r of R) A and component r of R v is (row r of R) . v, so a
  // lane that carries one row of the running rotation and one component of the running position needs nothing from the
  // other rows; 9 lanes walk the chain with a third of the serial work and no hand-offs
  if (lane < 9 && (lane >= 6 || shield_on)) {
    const auto& m = dm->m;
    const int cfg = lane / 3, row = lane - 3 * cfg;
    double r0 = dm->Rbase[3 * row], r1 = dm->Rbase[3 * row + 1], r2 = dm->Rbase[3 * row + 2];
    double p = m.base_pos[row];
#pragma unroll 1
    for (int i = 0; i < NARM; i++) {
      const double sn = sc[2 * (NARM * cfg + i)], cs = sc[2 * (NARM * cfg + i) + 1];
      p += r0 * m.body_pos[i][0] + r1 * m.body_pos[i][1] + r2 * m.body_pos[i][2];
      // arm hinges turn about the local z axis (robot.xml:33-58, checked at create): Rq Rz(q) mixes the first two columns
      double n0 = 0, n1 = 0, n2 = 0;
      {
        const double ra[3] = {r0, r1, r2};
#pragma unroll
        for (int a = 0; a < 3; a++) {
          const double c0 = dm->Rq[i][3 * a], c1 = dm->Rq[i][3 * a + 1], c2 = dm->Rq[i][3 * a + 2];
          n0 += ra[a] * (cs * c0 + sn * c1);
          n1 += ra[a] * (cs * c1 - sn * c0);
          n2 += ra[a] * c2;
        }
      }
      r0 = n0; r1 = n1; r2 = n2;
      if (cfg == 2) {
        L.kR[i][3 * row] = r0; L.kR[i][3 * row + 1] = r1; L.kR[i][3 * row + 2] = r2;
        L.kp[i][row] = p;
      } else {
        L.scap[cfg][i][row] = p + (r0 * m.scap_p1[i][0] + r1 * m.scap_p1[i][1] + r2 * m.scap_p1[i][2]);
        L.scap[cfg][i][3 + row] = p + (r0 * m.scap_p2[i][0] + r1 * m.scap_p2[i][1] + r2 * m.scap_p2[i][2]);
        if (i == NARM - 1) {
          L.scap[cfg][NARM][row] = p + (r0 * m.scap_p1[NARM][0] + r1 * m.scap_p1[NARM][1] + r2 * m.scap_p1[NARM][2]);
          L.scap[cfg][NARM][3 + row] = p + (r0 * m.scap_p2[NARM][0] + r1 * m.scap_p2[NARM][1] + r2 * m.scap_p2[NARM][2]);
        }
      }
    }
    if (cfg == 2) {
#pragma unroll 1
      for (int f = 0; f < HRG_NFINGER; f++) {
        const int i = NARM + f;
        double l[3] = {0, 0, 0};
        {
          const double ra[3] = {r0, r1, r2};
#pragma unroll
          for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) l[b] += ra[a] * dm->Rq[i][3 * a + b];
        }
        double pf = p + (r0 * m.body_pos[i][0] + r1 * m.body_pos[i][1] + r2 * m.body_pos[i][2]);
        const double axw = l[0] * m.jnt_axis[i][0] + l[1] * m.jnt_axis[i][1] + l[2] * m.jnt_axis[i][2];
        pf += axw * L.st.qpos[i];
        L.kR[i][3 * row] = l[0]; L.kR[i][3 * row + 1] = l[1]; L.kR[i][3 * row + 2] = l[2];
        L.kp[i][row] = pf;
      }
    }
  }
  wave_sync();
}

// mj_comPos / mj_crb / mj_rne(flg_acc=0) for the robot tree (SURVEY.md Appendix B.1 position+velocity stages).
// All spatial quantities are world-frame about the world origin, so nothing has to be transformed between bodies and
// the tree recursions collapse into independent per-lane sums:
//   lane = body i : velocity/acceleration = walk over the (<= 7) ancestors in registers; force of body i
//   lane = body k : composite inertia and joint force = sums over the descendants of k
//   lane = (i,j)  : M_ij = S_i . (Ic_j S_j)
// -> three hand-offs through LDS, no serial LDS read-modify-write chains.
PH_DYNTERMS void robot_dynamics_terms(const DevModel* __restrict__ dm_, int lane) {
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  double bI[10];  // this body's spatial inertia about the world origin (m, h, I)
  double Sw[3] = {0, 0, 0}, Sv[3] = {0, 0, 0};
  if (lane < NV) {
    const int i = lane;
    const double* R = L.kR[i];
    double axw[3], t[3], c[3];
    m3mulv(axw, R, m.jnt_axis[i]);
    if (i < NARM) { v3cpy(Sw, axw); v3cross(Sv, L.kp[i], axw); }
    else v3cpy(Sv, axw);
    v3cpy(L.Sw[i], Sw); v3cpy(L.Sv[i], Sv);
    m3mulv(t, R, m.body_com[i]);
    v3add(c, L.kp[i], t);
    // world-frame rotational inertia R Ib R' (symmetric: six entries): T = R Ib, then W_ab = T_a . R_b for a <= b
    const auto* I = m.body_inertia[i];
    const double Ib[9] = {I[0], I[3], I[4], I[3], I[1], I[5], I[4], I[5], I[2]};
    double T[9];
    m3mul(T, R, Ib);
    const double Iw[6] = {v3dot(T, R), v3dot(T + 3, R + 3), v3dot(T + 6, R + 6), v3dot(T, R + 3), v3dot(T, R + 6), v3dot(T + 3, R + 6)};
    sinertia_body(bI, m.body_mass[i], c, Iw);
  } else {
#pragma unroll
    for (int a = 0; a < 10; a++) bI[a] = 0.0;
  }
  const int i = lane < NV ? lane : 0;
  // ---- composite inertia of the subtree rooted at i, applied to the joint axis: subtree sums as suffix scans along the chain (DPP row shifts), no hand-off through
  //      LDS.  (Stage by stage with scheduling barriers in between: interleaving the stages for instruction-level parallelism costs more registers than the
  //      128 this kernel has.) ----
  {
    double cc[10];
#pragma unroll
    for (int a = 0; a < 10; a++) cc[a] = subtree_sum(bI[a], lane);
    double n[3], f[3];
    sinertia_mul(n, f, cc, Sw, Sv);
    if (lane < NV) {
#pragma unroll
      for (int a = 0; a < 3; a++) { L.F[i][a] = n[a]; L.F[i][3 + a] = f[a]; }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    // ---- velocity / acceleration of body i: accumulate along the ancestor path (ancestors have smaller indices) ----
    // The arm is a serial chain and the fingers hang off its last link (checked at create), so the running sums along the
    // ancestor path are prefix sums over lanes 0..5 (+ the finger's own term): three DPP row-shift steps per component
    // instead of a loop over up to 7 ancestors.  vw_k, vv_k = body velocity, term_k = the Coriolis terms joint k adds.
    double vw[3], vv[3], aw[3], av[3];
    {
      double jw[3], jv[3];
      const double qd = lane < NV ? L.st.qvel[i] : 0.0;
      v3scl(jw, Sw, qd);
      v3scl(jv, Sv, qd);
      for (int a = 0; a < 3; a++) { vw[a] = chain_prefix(jw[a], lane); vv[a] = chain_prefix(jv[a], lane); }
      double t1[3], t2[3], t3[3];
      v3cross(t1, vw, jw);
      v3cross(t2, vw, jv);
      v3cross(t3, vv, jw);
      for (int a = 0; a < 3; a++) { aw[a] = chain_prefix(t1[a], lane); av[a] = chain_prefix(t2[a] + t3[a], lane) - m.gravity[a]; }
    }
    if (lane < NV) { v3cpy(L.vw[i], vw); v3cpy(L.vv[i], vv); }
    __builtin_amdgcn_sched_barrier(0);
    // ---- force of body i: f = I a + v x* (I v); the joint force of the bias term is S_k . (sum of the forces of the bodies in the subtree of k) ----
    double fn[3], ff[3];
    {
      double n1[3], f1[3];
      sinertia_mul(n1, f1, bI, aw, av);
      double n2[3], f2[3], t1[3], t2[3], t3[3];
      sinertia_mul(n2, f2, bI, vw, vv);
      v3cross(t1, vw, n2);
      v3cross(t2, vv, f2);
      v3cross(t3, vw, f2);
#pragma unroll
      for (int a = 0; a < 3; a++) { fn[a] = n1[a] + t1[a] + t2[a]; ff[a] = f1[a] + t3[a]; }
    }
    __builtin_amdgcn_sched_barrier(0);
    double gm = 0;
#pragma unroll
    for (int a = 0; a < 3; a++) gm += Sw[a] * subtree_sum(fn[a], lane) + Sv[a] * subtree_sum(ff[a], lane);
    if (lane < NV) L.bias[i] = gm;
  }
  wave_sync();
  { // mass matrix: lane (i,j)
    const int i = lane >> 3, j = lane & 7;
    double v = 0;
    if ((dm->anc_mask[j] >> i) & 1) v = v3dot(L.Sw[i], &L.F[j][0]) + v3dot(L.Sv[i], &L.F[j][3]);
    else if ((dm->anc_mask[i] >> j) & 1) v = v3dot(L.Sw[j], &L.F[i][0]) + v3dot(L.Sv[j], &L.F[i][3]);
    if (i == j) v += m.jnt_armature[i];
    L.M[lane] = v;
  }
  wave_sync();
}

DI void robot_point_vel(int b, const double* r, double* v) {
  Lds& L = g_L;
  if (b < 0) { v3set(v, 0, 0, 0); return; }
  double t[3];
  v3cross(t, L.vw[b], r);
  v3add(v, L.vv[b], t);
}

// ================================================================================================ human
// HumanEnv._control_human (human_env.py:1710-1767) + kinematics of the 24-body tree on lanes = bodies.
DI int clip_of(ModelPtr dm, int64_t gid, int episode, int anim_index) {
  // every argument is wave-uniform but arrives in VGPRs (loaded from the LDS image): moved to SGPRs, so that the four 64-bit mixing rounds of the hash run
  // once on the scalar unit instead of on 64 lanes (this runs every substep)
  episode = __builtin_amdgcn_readfirstlane(episode);
  anim_index = __builtin_amdgcn_readfirstlane(anim_index);
  gid = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)gid >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)gid));
  double u = rng_u01(dm->m.seed, (uint64_t)gid, (uint64_t)episode, STREAM_ANIM, (uint64_t)anim_index);
  int c = (int)(u * dm->m.n_clips);
  return c >= dm->m.n_clips ? dm->m.n_clips - 1 : c;
}

DI void human_fk_lanes(const DevModel* __restrict__ dm_, int lane, const double* mocap_pos, const double* mocap_quat, const double* qh /*global or null*/,
                       int hold_body = -1, int hold_left = 0) {
  (void)hold_body; (void)hold_left;
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  const int b = lane < HRG_NHB ? lane : 0;
  // Every body starts with its transform relative to its parent: a rotation Rz Ry Rx about the joint anchor,
  // (Rloc, anchor - Rloc anchor); the pelvis starts with the mocap pose.  Pointer jumping then composes each body with the
  // partial product of its 1st, 2nd, 4th, 8th ancestor: log2(depth) rounds of one 12-double shuffle and one rigid compose
  // instead of one round per tree level.
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0};
  if (lane == 0) { quat2mat(R, mocap_quat); v3cpy(p, mocap_pos); }
  else if (lane < HRG_NHB) {
    double qz = 0, qy = 0, qx = 0, anchor[3], t[3];
    v3cpy(anchor, m.hb_anchor[b]);
    if (qh) { qz = qh[3 * (b - 1)]; qy = qh[3 * (b - 1) + 1]; qx = qh[3 * (b - 1) + 2]; }
    // Rz(qz) Ry(qy) Rx(qx) in closed form (the oracle multiplies the three axis-angle matrices: the same rotation, a third of the products)
    double sz, cz, sy, cy, sx, cx;
    sincos_small(qz, &sz, &cz);
    sincos_small(qy, &sy, &cy);
    sincos_small(qx, &sx, &cx);
    const double czsy = cz * sy, szsy = sz * sy;
    R[0] = cz * cy; R[1] = czsy * sx - sz * cx; R[2] = czsy * cx + sz * sx;
    R[3] = sz * cy; R[4] = szsy * sx + cz * cx; R[5] = szsy * cx - cz * sx;
    R[6] = -sy;     R[7] = cy * sx;             R[8] = cy * cx;
    m3mulv(t, R, anchor);
    v3sub(p, anchor, t);
  }
#pragma unroll 1
  for (int s = 0; s < dm->hb_njump; s++) {
    const int a = lane < HRG_NHB ? dm->hb_jump[s][b] : -1;
    const int src = a < 0 ? 0 : a;
    double Ra[9], pa[3];
#pragma unroll
    for (int k = 0; k < 9; k++) Ra[k] = __shfl(R[k], src, 64);
#pragma unroll
    for (int k = 0; k < 3; k++) pa[k] = __shfl(p[k], src, 64);
    if (a >= 0) {
      double t[3];
      m3mulv(t, Ra, p);
      v3add(p, pa, t);
      m3mul(R, Ra, R);
    }
  }
  if (lane < HRG_NHB) {
    double t[3];
    m3mulv(t, R, m.hcap_p1[b]); v3add(&L.hcap[b][0], p, t);
    m3mulv(t, R, m.hcap_p2[b]); v3add(&L.hcap[b][3], p, t);
  }
#if HRG_HANDOVER
  if (hold_body >= 0 && lane == hold_body) { // _update_mocap_body_transform (human_robot_handover_cartesian_env.py:609-633): hand rotation turned -+90 deg about its y axis
    const int r2h = m.task == HRG_TASK_HANDOVER_R2H;  // robot_human_handover_cartesian_env.py:615-648: opposite turn, offset towards the thumb
    const double ang = (hold_left != r2h) ? 0.5 * HRG_PI : -0.5 * HRG_PI, c = cos(ang), sn = sin(ang);
    const double Ry[9] = {c, 0, sn, 0, 1, 0, -sn, 0, c};
    double Rm[9], q[4];
    m3mul(Rm, R, Ry);
    const double tr = Rm[0] + Rm[4] + Rm[8];
    if (tr > 0) { const double S = fsqrt(tr + 1.0) * 2; q[0] = 0.25 * S; q[1] = (Rm[7] - Rm[5]) / S; q[2] = (Rm[2] - Rm[6]) / S; q[3] = (Rm[3] - Rm[1]) / S; }
    else if (Rm[0] > Rm[4] && Rm[0] > Rm[8]) { const double S = fsqrt(1.0 + Rm[0] - Rm[4] - Rm[8]) * 2; q[0] = (Rm[7] - Rm[5]) / S; q[1] = 0.25 * S; q[2] = (Rm[1] + Rm[3]) / S; q[3] = (Rm[2] + Rm[6]) / S; }
    else if (Rm[4] > Rm[8]) { const double S = fsqrt(1.0 + Rm[4] - Rm[0] - Rm[8]) * 2; q[0] = (Rm[2] - Rm[6]) / S; q[1] = (Rm[1] + Rm[3]) / S; q[2] = 0.25 * S; q[3] = (Rm[5] + Rm[7]) / S; }
    else { const double S = fsqrt(1.0 + Rm[8] - Rm[0] - Rm[4]) * 2; q[0] = (Rm[3] - Rm[1]) / S; q[1] = (Rm[2] + Rm[6]) / S; q[2] = (Rm[5] + Rm[7]) / S; q[3] = 0.25 * S; }
    for (int a = 0; a < 4; a++) L.hand_q[a] = q[a];
    const double off[3] = {r2h ? (hold_left ? 0.02 : -0.02) : 0.0, r2h ? -0.03 : 0.0, r2h ? -0.03 : 0.0};
    double to[3];
    m3mulv(to, Rm, off);
    for (int a = 0; a < 3; a++) L.hand_off[a] = to[a];
  }
#endif
#if HRG_STACK
  { // _update_mocap_body_transforms (collaborative_stacking_cartesian_env.py:905-936): each hand's mocap body = the hand site moved 3 cm towards the thumb
    // (hand z axis), the hand rotation turned -90 deg (left) / +90 deg (right) about its y axis; computed by the lanes of the two hand bodies
    const int bl = m.meas_body[m.site_lhand], br = m.meas_body[m.site_rhand];
    if (lane == bl || lane == br) {
      const int hd = lane == bl ? 0 : 1;
      const double sn = hd == 0 ? -1.0 : 1.0;                  // cos(-+pi/2) = 0, sin = -+1: R Ry = [-sn c2 | c1 | sn c0] by columns
      double Rm[9], q[4], site[3], t[3];
      for (int a = 0; a < 3; a++) { Rm[3 * a] = -sn * R[3 * a + 2]; Rm[3 * a + 1] = R[3 * a + 1]; Rm[3 * a + 2] = sn * R[3 * a]; }
      const double tr = Rm[0] + Rm[4] + Rm[8];
      if (tr > 0) { const double S = fsqrt(tr + 1.0) * 2; q[0] = 0.25 * S; q[1] = (Rm[7] - Rm[5]) / S; q[2] = (Rm[2] - Rm[6]) / S; q[3] = (Rm[3] - Rm[1]) / S; }
      else if (Rm[0] > Rm[4] && Rm[0] > Rm[8]) { const double S = fsqrt(1.0 + Rm[0] - Rm[4] - Rm[8]) * 2; q[0] = (Rm[7] - Rm[5]) / S; q[1] = 0.25 * S; q[2] = (Rm[1] + Rm[3]) / S; q[3] = (Rm[2] + Rm[6]) / S; }
      else if (Rm[4] > Rm[8]) { const double S = fsqrt(1.0 + Rm[4] - Rm[0] - Rm[8]) * 2; q[0] = (Rm[2] - Rm[6]) / S; q[1] = (Rm[1] + Rm[3]) / S; q[2] = 0.25 * S; q[3] = (Rm[5] + Rm[7]) / S; }
      else { const double S = fsqrt(1.0 + Rm[8] - Rm[0] - Rm[4]) * 2; q[0] = (Rm[3] - Rm[1]) / S; q[1] = (Rm[2] + Rm[6]) / S; q[2] = (Rm[5] + Rm[7]) / S; q[3] = 0.25 * S; }
      m3mulv(t, R, m.hb_anchor[b]);
      v3add(site, p, t);
      for (int a = 0; a < 4; a++) L.sk.mocap_quat[hd][a] = q[a];
      for (int a = 0; a < 3; a++) L.sk.mocap_pos[hd][a] = site[a] + 0.03 * R[3 * a + 2];
    }
  }
#endif
#if HRG_HAMMER
  { // _update_mocap_body_transforms (collaborative_hammering_cartesian_env.py:688-715): each hand's mocap body = the hand site, the hand rotation turned
    // -90 deg (left) / +90 deg (right) about its y axis; computed by the lanes of the two hand bodies
    const int bl = m.meas_body[m.site_lhand], br = m.meas_body[m.site_rhand];
    if (lane == bl || lane == br) {
      const int hd = lane == bl ? 0 : 1;
      const double sn = hd == 0 ? -1.0 : 1.0;
      double Rm[9], q[4], site[3], t[3];
      for (int a = 0; a < 3; a++) { Rm[3 * a] = -sn * R[3 * a + 2]; Rm[3 * a + 1] = R[3 * a + 1]; Rm[3 * a + 2] = sn * R[3 * a]; }
      const double tr = Rm[0] + Rm[4] + Rm[8];
      if (tr > 0) { const double S = fsqrt(tr + 1.0) * 2; q[0] = 0.25 * S; q[1] = (Rm[7] - Rm[5]) / S; q[2] = (Rm[2] - Rm[6]) / S; q[3] = (Rm[3] - Rm[1]) / S; }
      else if (Rm[0] > Rm[4] && Rm[0] > Rm[8]) { const double S = fsqrt(1.0 + Rm[0] - Rm[4] - Rm[8]) * 2; q[0] = (Rm[7] - Rm[5]) / S; q[1] = 0.25 * S; q[2] = (Rm[1] + Rm[3]) / S; q[3] = (Rm[2] + Rm[6]) / S; }
      else if (Rm[4] > Rm[8]) { const double S = fsqrt(1.0 + Rm[4] - Rm[0] - Rm[8]) * 2; q[0] = (Rm[2] - Rm[6]) / S; q[1] = (Rm[1] + Rm[3]) / S; q[2] = 0.25 * S; q[3] = (Rm[5] + Rm[7]) / S; }
      else { const double S = fsqrt(1.0 + Rm[8] - Rm[0] - Rm[4]) * 2; q[0] = (Rm[3] - Rm[1]) / S; q[1] = (Rm[2] + Rm[6]) / S; q[2] = (Rm[5] + Rm[7]) / S; q[3] = 0.25 * S; }
      m3mulv(t, R, m.hb_anchor[b]);
      v3add(site, p, t);
      for (int a = 0; a < 4; a++) L.hm.mocap_quat[hd][a] = q[a];
      for (int a = 0; a < 3; a++) L.hm.mocap_pos[hd][a] = site[a];
    }
  }
#endif
  // sites of the measured joints: site k sits at the anchor of body meas_body[k]
  {
    const int k = lane < HRG_NHJ ? lane : 0;
    const int sb = m.meas_body[k];
    double Rs[9], ps[3];
#pragma unroll
    for (int a = 0; a < 9; a++) Rs[a] = __shfl(R[a], sb, 64);
#pragma unroll
    for (int a = 0; a < 3; a++) ps[a] = __shfl(p[a], sb, 64);
    if (lane < HRG_NHJ) {
      double t[3];
      m3mulv(t, Rs, m.hb_anchor[sb]);
      v3add(L.st.human_site[k], ps, t);
    }
  }
  wave_sync();
}

// pose of the human at frame `at` of clip `clip`: root pose chain (human_env.py:1736-1763) + tree kinematics -> L.hcap, human_site
DI void human_pose_fk(const DevModel* __restrict__ dm_, int lane, int clip, int at, int hold_body = -1, int hold_left = 0) {
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  hrg_env_state& s = L.st;
  const double* fr = dm->clips.frames + (dm->clips.clip_offset[clip] + at) * HRG_FRAME_DIM;
  double qi[4] = {dm->clips.clip_quat[clip][3], dm->clips.clip_quat[clip][0], dm->clips.clip_quat[clip][1], dm->clips.clip_quat[clip][2]};
  double qbi[4], Rbi[9], pa[3], pr[3], mp[3], mq[4], q1[4];
  quatmul(qbi, m.human_base_quat, qi);
  quat2mat(Rbi, qbi);
  for (int a = 0; a < 3; a++) pa[a] = fr[a] + dm->clips.clip_pos_offset[clip][a];
  m3mulv(pr, Rbi, pa);
  v3add(mp, pr, s.human_pos_offset);
  double qa[4] = {fr[6], fr[3], fr[4], fr[5]};
  quatmul(q1, s.human_rot_offset, qbi);
  quatmul(mq, q1, qa);
  human_fk_lanes(dm_, lane, mp, mq, fr + 7, hold_body, hold_left);
}

#if HRG_BOX || HRG_STACK || HRG_HAMMER
// amplitude (speed = 0) or speed modifier (1) of layered sine k of the idle loop of animation slot ai in this episode
// (sample_animation_loop_properties, utils/animation_utils.py:122-176), drawn counter-based on demand
DI double loop_prop(ModelPtr dm, int64_t gid, int episode, int ai, int clip, int k, int speed) {
  double base = speed ? dm->clips.clip_loop_speed[clip][k % HRG_MAX_LOOP] : dm->clips.clip_loop_amp[clip][k % HRG_MAX_LOOP];
  if (k >= HRG_MAX_LOOP) base = speed ? dm->clips.clip_loop2_speed[clip][k - HRG_MAX_LOOP] : dm->clips.clip_loop2_amp[clip][k - HRG_MAX_LOOP];  // second loop stage ("wait")
  const double sf = speed ? dm->clips.clip_loop_speed_std[clip] : dm->clips.clip_loop_amp_std[clip];
  const double z = clampd(rng_gauss(dm->m.seed, (uint64_t)gid, (uint64_t)episode, STREAM_LOOP, (uint64_t)((ai * 2 * HRG_MAX_LOOP + k) * 2 + speed)), -3.0, 3.0);
  return base * exp(z * log(sf));
}
// layered_sin_modulations (utils/animation_utils.py:91-119) over loop sines kfirst..kfirst+n-1 of the clip
// (wave-uniform call; the 2n draws run on lanes 0..2n-1 and are handed round by shuffles, the sum runs in the reference's order)
DI double layered_sines(ModelPtr dm, int64_t gid, int episode, int ai, int clip, int kfirst, int n, double t, double start) {
  static_assert(2 * HRG_MAX_LOOP <= 64, "one lane per loop property");
  const int lane = hrg_lane(), kk = lane % HRG_MAX_LOOP, sp = (lane / HRG_MAX_LOOP) & 1;
  const double mine = loop_prop(dm, gid, episode, ai, clip, kfirst + (kk < n ? kk : 0), sp);
  double sum = 0;
  for (int k = 0; k < n; k++) {
    const double A = __shfl(mine, k), S = __shfl(mine, HRG_MAX_LOOP + k);
    sum += A * sin((t - start) / (A / S)) + start;
  }
  return sum - start * (double)(n - 1);
}
#endif

PH_HUMAN void human_control(const DevModel* __restrict__ dm_, int lane, int64_t gid) {
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  hrg_env_state& s = L.st;
  // human_env.py:1719-1731 (wave-uniform)
  int control_time = (int)floor((double)s.low_level_time / m.anim_step_length);
  int at = control_time - s.anim_start_time;
  int anim_index = s.anim_index, anim_start = s.anim_start_time;
  int clip = clip_of(dm, gid, s.episode, anim_index);
#if HRG_BOX
  if (m.task == HRG_TASK_INSPECTION) { // HumanObjectInspectionCart._compute_animation_time (human_object_inspection_cartesian_env.py:602-652); wave-uniform
    hrg_box_state& bx = L.bx;
    const int classic = control_time - s.anim_start_time, k0 = dm->clips.clip_keyframes[clip][0], k1 = dm->clips.clip_keyframes[clip][1], len = dm->clips.clip_len[clip];
    int phase = bx.task_phase, nd = bx.n_delayed;
    at = classic;   // clamped to the clip below: the clip only advances through _on_goal_reached
    if (at > k0 && phase == HRG_PHASE_APPROACH) phase = HRG_PHASE_READY;
    if (phase == HRG_PHASE_READY) { // idle loop around the first keyframe: layered_sin_modulations (utils/animation_utils.py:62-119)
      const int nl = dm->clips.clip_n_loop[clip];
      double sum = 0;
      for (int k = 0; k < nl; k++) {
        const double A = loop_prop(dm, gid, s.episode, s.anim_index, clip, k, 0), S = loop_prop(dm, gid, s.episode, s.anim_index, clip, k, 1);
        sum += A * sin((double)(classic - k0) / (A / S)) + (double)k0;
      }
      at = (int)(sum - (double)k0 * (double)(nl - 1));
      nd = classic - at;
    } else at -= nd;
    if (at > k1) phase = HRG_PHASE_RETREAT;
    if (at >= len - 1) { phase = HRG_PHASE_COMPLETE; at = len - 1; }
    if (at < 0) at = 0;
    wave_sync();
    bx.task_phase = phase; bx.n_delayed = nd;
    if (lane < 3) bx.target[lane] = dm->clips.clip_target_pos[clip][lane] + s.human_pos_offset[lane];  // target_pos property (447-459)
  }
  int hold_body = -1, hold_left = 0;
#if HRG_HANDOVER
  if (m.task == HRG_TASK_HANDOVER_H2R) { // HumanRobotHandoverCart._compute_animation_time (human_robot_handover_cartesian_env.py:530-596); wave-uniform
    hrg_box_state& bx = L.bx;
    const int classic = at, k0 = dm->clips.clip_keyframes[clip][0], k1 = dm->clips.clip_keyframes[clip][1], len = dm->clips.clip_len[clip];
    int phase = bx.task_phase, nd = bx.n_delayed, nd2 = bx.n_delayed2;
    if (at > k0 && phase == HRG_PHASE_APPROACH) phase = HRG_PHASE_PRESENT;
    else if ((double)at > (double)k0 + (double)(k1 - k0) / 2.0 && phase == HRG_PHASE_PRESENT) {
      at = (int)layered_sines(dm, gid, s.episode, s.anim_index, clip, 0, dm->clips.clip_n_loop[clip], (double)classic, (double)(k0 + k1) / 2.0);
      nd = classic - at; nd2 = 0;
    } else if (phase == HRG_PHASE_WAIT) {
      at = classic - nd;
      if (at >= k1) at = (int)layered_sines(dm, gid, s.episode, s.anim_index, clip, HRG_MAX_LOOP, dm->clips.clip_n_loop2[clip], (double)at, (double)k1);
      nd2 = classic - at;
    } else if (phase == HRG_PHASE_RETREAT) at -= nd2;
    if (at >= len - 1) { phase = HRG_PHASE_COMPLETE; at = len - 1; }
    if (at < 0) at = 0;
    wave_sync();
    bx.task_phase = phase; bx.n_delayed = nd; bx.n_delayed2 = nd2;
    hold_left = dm->clips.clip_holding_hand[clip];
    hold_body = m.meas_body[hold_left ? m.site_lhand : m.site_rhand];
  }
  if (m.task == HRG_TASK_HANDOVER_R2H) { // RobotHumanHandoverCart._compute_animation_time (robot_human_handover_cartesian_env.py:556-606); wave-uniform
    hrg_box_state& bx = L.bx;
    const int classic = at, k0 = dm->clips.clip_keyframes[clip][0], k1 = dm->clips.clip_keyframes[clip][1], len = dm->clips.clip_len[clip];
    int phase = bx.task_phase, nd = bx.n_delayed;
    if (at > k0 && phase == HRG_R2H_APPROACH) phase = HRG_R2H_REACH_OUT;
    if ((double)at > (double)k0 + (double)(k1 - k0) / 2.0 && phase == HRG_R2H_REACH_OUT) {
      at = (int)layered_sines(dm, gid, s.episode, s.anim_index, clip, 0, dm->clips.clip_n_loop[clip], (double)classic, (double)(k0 + k1) / 2.0);
      nd = classic - at;
    }
    if (phase == HRG_R2H_RETREAT) at -= nd;
    if (at >= len - 1) { phase = HRG_R2H_COMPLETE; at = len - 1; }
    if (at < 0) at = 0;
    wave_sync();
    bx.task_phase = phase; bx.n_delayed = nd;
    hold_left = dm->clips.clip_holding_hand[clip];
    hold_body = m.meas_body[hold_left ? m.site_lhand : m.site_rhand];
  }
#endif
#endif
#if HRG_STACK
  { // CollaborativeStackingCart._compute_animation_time (collaborative_stacking_cartesian_env.py:825-897); wave-uniform
    hrg_stack_state& sk = L.sk;
    const int classic = at, len = dm->clips.clip_len[clip];
    const int k0 = dm->clips.clip_stack_keyframes[clip][0], k2 = dm->clips.clip_stack_keyframes[clip][2], k4 = dm->clips.clip_stack_keyframes[clip][4];
    int phase = sk.task_phase, nd0 = sk.n_delayed[0], nd1 = sk.n_delayed[1];
    if (phase == HRG_STK_APPROACH && at > k0) phase = HRG_STK_PLACE_FIRST;
    else if (phase == HRG_STK_WAIT_FOR_SECOND) {
      if (at >= k2) at = (int)layered_sines(dm, gid, s.episode, s.anim_index, clip, 0, dm->clips.clip_n_loop[clip], (double)classic, (double)k2);
      nd0 = classic - at; nd1 = 0;
    } else if (phase == HRG_STK_PLACE_THIRD) at = classic - nd0;
    else if (phase == HRG_STK_WAIT_FOR_FOURTH) {
      at = classic - nd0;
      if (at >= k4) at = (int)layered_sines(dm, gid, s.episode, s.anim_index, clip, HRG_MAX_LOOP, dm->clips.clip_n_loop2[clip], (double)at, (double)k4);
      nd1 = classic - at;
    } else if (phase == HRG_STK_RETREAT) at = classic - nd1;
    if (at >= len - 1) { phase = HRG_STK_COMPLETE; at = len - 1; }
    if (at < 0) at = 0;
    wave_sync();
    sk.task_phase = phase; sk.n_delayed[0] = nd0; sk.n_delayed[1] = nd1;
  }
#endif
#if HRG_HAMMER
  { // CollaborativeHammeringCart._compute_animation_time (collaborative_hammering_cartesian_env.py:636-680); wave-uniform
    hrg_hammer_state& hm = L.hm;
    const int classic = at, len = dm->clips.clip_len[clip];
    const double k0 = (double)dm->clips.clip_keyframes[clip][0], mid = 0.5 * (k0 + (double)dm->clips.clip_keyframes[clip][1]);
    int phase = hm.task_phase, nd = hm.n_delayed;
    if (phase == HRG_HM_APPROACH && (double)at > k0) phase = HRG_HM_PRESENT;
    else if (phase == HRG_HM_PRESENT && (double)at > mid) {   // idle loop until the nail is hammered in
      at = (int)layered_sines(dm, gid, s.episode, s.anim_index, clip, 0, dm->clips.clip_n_loop[clip], (double)classic, mid);
      nd = classic - at;
    } else if (phase == HRG_HM_RETREAT) at -= nd;
    if (at >= len - 1) { phase = HRG_HM_COMPLETE; at = len - 1; }
    if (at < 0) at = 0;
    wave_sync();
    hm.task_phase = phase; hm.n_delayed = nd;
  }
#endif
#if HRG_LIFT
  if (m.task == HRG_TASK_LIFTING) { // CollaborativeLiftingCart._compute_animation_time (collaborative_lifting_cartesian_env.py:563-581): frozen at the last frame
    const int len = dm->clips.clip_len[clip];
    const bool complete = at >= len - 1;
    if (complete) at = len - 1;
    if (at < 0) at = 0;
    wave_sync();
    if (complete) L.bx.task_phase = HRG_PHASE_COMPLETE;
  }
#endif
  if (at > dm->clips.clip_len[clip] - 1) {
    anim_index = (anim_index + 1) % m.n_anim_ids; // human_env.py:1704-1708
    at = 0;
    anim_start = control_time;
    clip = clip_of(dm, gid, s.episode, anim_index);
  }
  s.anim_index = anim_index; s.anim_start_time = anim_start; s.animation_time = at;
#if HRG_BOX
  human_pose_fk(dm_, lane, clip, at, hold_body, hold_left);
#else
  human_pose_fk(dm_, lane, clip, at);
#endif
}

// ================================================================================================ shield
// SafetyShield.humanMeasurement + step (controllers/failsafe_controller/failsafe_controller/failsafe_controller.py:310,329),
// restated as in oracle/hrg_oracle.c: candidate = one recovery step + fail-safe brake; robot reach capsules;
// human reach capsules (ACC/VEL/POS) on lanes; swept-capsule test lanes x 7 robot capsules; __ballot verdict.
PH_SHIELD void shield_step(const DevModel* __restrict__ dm_, int lane, int e, double* __restrict__ dbg_r, double* __restrict__ dbg_h, int32_t* __restrict__ dbg_nh) {
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  hrg_env_state& s = L.st;
  const double dt = m.timestep, t = s.time;
  const bool shield_on = m.shield_type != HRG_SHIELD_OFF;
  const bool have_vel = s.n_meas >= 1 && t > s.meas_prev_t;
  // current motion = the Motion returned last cycle (same trajectory, same path state -> bitwise the same evaluation)
  if (lane < NARM) { L.cq[lane] = s.des_q[lane]; L.cv[lane] = s.des_v[lane]; L.ca[lane] = s.des_a[lane]; }
  wave_sync();
  int use_cand = 0;
  if (s.new_goal) {
    // candidate trajectory from the NOMINAL state of the active one at the current path position (q, dq/ds, d2q/ds2): it
    // continues with the current path velocity, so it can be swapped in at any path speed without a jump
    double nq = 0, nv = 0, na = 0;
    if (lane < NARM) ltt_eval(&s.ltt, lane, s.path_s, &nq, &nv, &na);
    const bool bad = lane < NARM && fabs(na) > m.a_max_ltt[lane];
    if (!__any(bad)) {
      use_cand = 1;
      double tj = 0;
      LttTail tail;
      tail.n0 = 0; tail.v = tail.sg = tail.Dm = tail.vm = tail.w = 0;
      if (lane < NARM) {
        ltt_plan_joint(&L.cand, lane, nq, nv, na, s.new_goal_q[lane], m.v_max_ltt[lane], m.a_max_ltt[lane], m.j_max_ltt[lane], &tail);
        for (int i = 0; i < HRG_LTT_NSEG; i++) tj += L.cand.dur[lane][i];
      }
      const double Tall = wave_max(tj);
      L.cand.T = Tall;
      if (m.ltt_time_sync && lane < NARM) ltt_sync_joint(&L.cand, lane, Tall, m.a_max_ltt[lane], m.j_max_ltt[lane], tail);   // the joints arrive together
    }
    wave_sync();
  }
  STAMP(10);
  const hrg_ltt* Lp = use_cand ? &L.cand : &s.ltt;
  const double ps = use_cand ? 0.0 : s.path_s, pv = s.path_v, pa = s.path_a;
  hrg_path fs2;
  double s1, v1, a1, se, Tb;
  const bool pfl = m.shield_type == HRG_SHIELD_PFL, steady = pv == 1.0 && pa == 0.0;
  if (steady) { s1 = ps + dt; v1 = 1.0; a1 = 0.0; }   // steady state on the trajectory (the common case): the recovery step is s += dt
  else {
    hrg_path rec;
    path_plan(&rec, ps, pv, pa, 1.0, m.path_amax, m.path_jmax);
    path_eval(&rec, dt, 1.0, &s1, &v1, &a1);
  }
  // speed the fail-safe manoeuvre brakes to: a full stop (SSM), or under PFL the path speed at which no point of the arm exceeds pfl_v_safe on this trajectory:
  // |v_point| <= s' sum_j |dq_j/ds| r_j with dq/ds taken where the manoeuvre starts (lanes = joints, one wave reduction; re-evaluated every cycle)
  double ve_fs = m.failsafe_sdot;
  if (pfl) {
    double q_, d1_ = 0, d2_;
    if (lane < NARM) ltt_eval(Lp, lane, s1, &q_, &d1_, &d2_);
    const double vc = wave_sum(lane < NARM ? fabs(d1_) * m.pfl_reach[lane] : 0.0);
    ve_fs = vc > m.pfl_v_safe ? m.pfl_v_safe / vc : 1.0;
  }
  if (steady && !pfl) {
    // ... and the fail-safe profile is the model constant "brake from full path speed to a stop" shifted to s1
    fs2.s0 = s1; fs2.v0 = 1.0; fs2.a0 = 0.0; fs2.k = 0.0;
    for (int a = 0; a < 3; a++) { fs2.dur[a] = dm->brake_full.dur[a]; fs2.jerk[a] = dm->brake_full.jerk[a]; }
    Tb = dm->brake_T;
    se = s1 + dm->brake_ds;
  } else {
    double ve_, ae;
    path_plan(&fs2, s1, v1, a1, ve_fs, m.path_amax, m.path_jmax);
    Tb = path_total(&fs2);
    path_eval(&fs2, Tb, ve_fs, &se, &ve_, &ae);
  }
  // the path state of this cycle is wave-uniform and lives across the reach-capsule verification below: into scalar registers
  s1 = uniform_f64(s1); v1 = uniform_f64(v1); a1 = uniform_f64(a1); se = uniform_f64(se); Tb = uniform_f64(Tb);
  fs2.s0 = uniform_f64(fs2.s0); fs2.v0 = uniform_f64(fs2.v0); fs2.a0 = uniform_f64(fs2.a0); fs2.k = uniform_f64(fs2.k);
  for (int a = 0; a < 3; a++) { fs2.dur[a] = uniform_f64(fs2.dur[a]); fs2.jerk[a] = uniform_f64(fs2.jerk[a]); }
  STAMP(11);
  if (shield_on && lane < NARM) {
    // configuration at the end of the brake, and in the same walk the Motion of the next cycle for the (usual) case that the
    // step is verified safe: path state (s1, v1, a1) on this trajectory.  An unsafe verdict re-evaluates it below.
    double qn, q1, q2, qv;
    ltt_eval2(Lp, lane, s1, se, &qn, &q1, &q2, &qv);
    L.qe[lane] = qv;
    s.des_q[lane] = qn; s.des_v[lane] = q1 * v1; s.des_a[lane] = q1 * a1 + q2 * v1 * v1;
  }
  wave_sync();
  STAMP(12);
  robot_chain_fk(dm_, lane, shield_on);
  STAMP(13);
  int safe = 1;
  if (shield_on) {
    const double sdiff = se - ps;
    double rc_hl = 0;  // lane c < 7: half length of robot reach capsule c (read by the verification loop with a scalar readlane)
    if (lane < HRG_NSHIELD_RCAP) {
      const int c = lane;
      double d[3], l1, l2;
      v3sub(d, &L.scap[0][c][0], &L.scap[1][c][0]); l1 = v3norm(d);
      v3sub(d, &L.scap[0][c][3], &L.scap[1][c][3]); l2 = v3norm(d);
      for (int a = 0; a < 6; a++) L.rc[c][a] = 0.5 * (L.scap[0][c][a] + L.scap[1][c][a]);
      L.rc[c][6] = m.scap_r[c] + m.secure_radius + 0.5 * (l1 > l2 ? l1 : l2) + m.scap_alpha[c] * sdiff * sdiff / 8.0;
      double hv[3];
      for (int a = 0; a < 3; a++) hv[a] = 0.5 * (L.rc[c][3 + a] - L.rc[c][a]);
      rc_hl = fsqrt(v3dot(hv, hv));
      L.scap[0][c][0] = rc_hl;   // (this lane's own chain-kinematics entry, dead from here on: the verification loop reads the half length from LDS)
      if (dbg_r) for (int a = 0; a < 7; a++) dbg_r[((size_t)e * HRG_NSHIELD_RCAP + c) * 7 + a] = L.rc[c][a];
    }
    wave_sync();
    double rbl[3], rbh[3];   // box around the robot reach capsules (each inflated by its radius), the same in every lane
    {
      double bl[3], bh[3];
      if (lane < HRG_NSHIELD_RCAP) {
        const double rr = L.rc[lane][6];
        for (int a = 0; a < 3; a++) { const double p1 = L.rc[lane][a], p2 = L.rc[lane][3 + a]; bl[a] = (p1 < p2 ? p1 : p2) - rr; bh[a] = (p1 > p2 ? p1 : p2) + rr; }
      } else for (int a = 0; a < 3; a++) { bl[a] = 1e300; bh[a] = -1e300; }
      for (int a = 0; a < 3; a++) { rbh[a] = lane_value<0>(row16_max(bh[a])); rbl[a] = -lane_value<0>(row16_max(-bl[a])); }
    }
    // lanes = human reach capsules
    const int nh = dm->hc_n;
    int mdl = -1;
    double c1[3] = {0, 0, 0}, c2[3] = {0, 0, 0}, r = 0, hc[3] = {0, 0, 0}, hl = 0;
    bool near_robot = false;
    if (lane < nh) {
      const int kind = dm->hc_kind[lane], j1 = dm->hc_j1[lane], j2 = dm->hc_j2[lane];
      const double Td = dt + Tb + m.delay;
      if (kind == 0) {
        double va[3], vb[3];
        const double idt = have_vel ? 1.0 / (t - s.meas_prev_t) : 0.0;
        for (int a = 0; a < 3; a++) {
          va[a] = (s.human_site[j1][a] - s.meas_prev[j1][a]) * idt;   // finite-difference velocity: one reciprocal, six products
          vb[a] = (s.human_site[j2][a] - s.meas_prev[j2][a]) * idt;
        }
        const double base = 0.5 * dm->hc_a[lane] * Td * Td + m.meas_err_pos + m.meas_err_vel * Td;
        const double r1 = v3norm(va) * Td * 0.5 + base, r2 = v3norm(vb) * Td * 0.5 + base;
        v3madd(c1, s.human_site[j1], va, 0.5 * Td);
        v3madd(c2, s.human_site[j2], vb, 0.5 * Td);
        r = (r1 > r2 ? r1 : r2) + dm->hc_th[lane];
        mdl = 0;
      } else if (kind == 2) {
        v3cpy(c1, s.human_site[j1]);
        v3cpy(c2, s.human_site[j1]);
        r = dm->hc_len[lane] + dm->hc_th[lane] + m.meas_err_pos + dm->hc_v[lane] * Td;
        mdl = 2;
      } else {
        v3cpy(c1, s.human_site[j1]);
        v3cpy(c2, s.human_site[j2]);
        r = dm->hc_th[lane] + m.meas_err_pos + dm->hc_v[lane] * Td;
        mdl = kind == 1 ? 1 : 2;
      }
      double hh[3];
      for (int a = 0; a < 3; a++) { hc[a] = 0.5 * (c1[a] + c2[a]); hh[a] = 0.5 * (c2[a] - c1[a]); }
      hl = fsqrt(v3dot(hh, hh));
      // whole-robot cull: the box around the seven robot reach capsules against this capsule's bounding sphere (conservative: a culled lane cannot intersect
      // any of them); the pair tests below run only for the lanes that come near
      double d2b = 0;
      for (int a = 0; a < 3; a++) { const double ee = hc[a] < rbl[a] ? rbl[a] - hc[a] : (hc[a] > rbh[a] ? hc[a] - rbh[a] : 0.0); d2b += ee * ee; }
      near_robot = d2b <= (hl + r + 1e-9) * (hl + r + 1e-9);
      if (dbg_h) {
        double* o = dbg_h + ((size_t)e * HRG_NHCAP_MAX + lane) * 7;
        o[0] = c1[0]; o[1] = c1[1]; o[2] = c1[2]; o[3] = c2[0]; o[4] = c2[1]; o[5] = c2[2]; o[6] = r;
      }
    }
    if (dbg_nh && lane == 0) dbg_nh[e] = nh;
    // The step is safe as soon as ONE of the three human models (ACC, VEL, POS) has no capsule that meets a robot reach capsule: the models are tested one after
    // the other and the rest is skipped once one is clear -- usually the first (ACC: the tightest sets while the human moves slowly), so the large POS / VEL
    // capsules, which come near the arm far more often, rarely reach their segment-segment tests at all; inside a model the search ends with its first hit.
    // The verdict is the one of testing every capsule (oracle: shield_step).
    bool all_hit = true;
#pragma unroll 1
    for (int mm = 0; mm < 3 && all_hit; mm++) {
      if (mm == 0 && !have_vel) continue;   // without a velocity estimate the ACC model counts as violated
      const bool mine = mdl == mm && near_robot;
      bool hit = false;
      if (__any(mine)) {
#pragma unroll 1
        for (int c = 0; c < HRG_NSHIELD_RCAP; c++) {
          bool close = false;
          const double rr = L.rc[c][6] + r;
          if (mine) {
            double rcn[3], dc[3];
            for (int a = 0; a < 3; a++) rcn[a] = 0.5 * (L.rc[c][a] + L.rc[c][3 + a]);
            v3sub(dc, rcn, hc);
            const double reach = L.scap[0][c][0] + hl + rr + 1e-9;   // (half length of robot capsule c, stored above)
            close = !(v3dot(dc, dc) > reach * reach);               // bounding spheres apart: cannot intersect
          }
          if (!__any(close)) continue;
          if (close) { double x1[3], x2[3]; if (seg_seg(&L.rc[c][0], &L.rc[c][3], c1, c2, x1, x2) < rr * rr) hit = true; }
          if (__any(hit)) break;
        }
      }
      all_hit = __any(hit);
    }
    safe = !all_hit;
  }
  wave_sync();
  STAMP(14);
  // humanMeasurement bookkeeping: previous measurement <- current sites
  for (int k = lane; k < HRG_NHJ * 3; k += 64) (&s.meas_prev[0][0])[k] = (&s.human_site[0][0])[k];
  s.meas_prev_t = t;
  if (s.n_meas < 2) s.n_meas = s.n_meas + 1;
  if (safe) {
    if (use_cand) {
      double* dst = (double*)&s.ltt;
      const double* src = (const double*)&L.cand;
      for (int k = lane; k < (int)(sizeof(hrg_ltt) / sizeof(double)); k += 64) dst[k] = src[k];
      s.new_goal = 0;
    }
    s.path_s = s1; s.path_v = v1; s.path_a = a1;
    s.safe_path = fs2;
  } else {
    // follow the last verified fail-safe profile
    const double k = s.safe_path.k + 1.0;
    s.safe_path.k = k;
    double ns, nv, na;
    const hrg_path spc = s.safe_path;
    const double vend = pfl ? path_vend(&spc) : m.failsafe_sdot;   // the speed that profile was planned to brake to
    path_eval(&s.safe_path, k * dt, vend, &ns, &nv, &na);
    // already at (or below) the fail-safe speed — stopped under SSM, at the PFL safe speed under PFL: a new trajectory may
    // be swapped in although it is not verified safe (sara-shield swaps "if safe or stopped")
    if (use_cand && s.path_v <= vend + 1e-9 && fabs(s.path_a) <= 1e-9) {
      const double adv = ns - s.path_s;
      double* dst = (double*)&s.ltt;
      const double* src = (const double*)&L.cand;
      for (int q = lane; q < (int)(sizeof(hrg_ltt) / sizeof(double)); q += 64) dst[q] = src[q];
      s.new_goal = 0;
      ns = adv;  // the candidate's path axis starts at the current position
      hrg_path sp;
      path_plan(&sp, ns, nv, na, vend, m.path_amax, m.path_jmax);
      s.safe_path = sp;
    }
    s.path_s = ns; s.path_v = nv; s.path_a = na;
  }
  s.is_safe = safe;
  wave_sync();
  if (lane < NARM && !(shield_on && safe)) {
    double qq, q1, q2;
    ltt_eval(&s.ltt, lane, s.path_s, &qq, &q1, &q2);
    s.des_q[lane] = qq; s.des_v[lane] = q1 * s.path_v; s.des_a[lane] = q1 * s.path_a + q2 * s.path_v * s.path_v;
  }
  wave_sync();
  STAMP(15);
}

DI void shield_reset(const DevModel* __restrict__ /*dm_*/, int lane) {
  Lds& L = g_L;
  hrg_env_state& s = L.st;
  // ltt_const + zero paths: the state block was zeroed by the caller
  if (lane < NARM) {
    const double q = s.qpos[lane];
    s.ltt.q0[lane] = q; s.ltt.qT[lane] = q;
    s.des_q[lane] = q; s.new_goal_q[lane] = q; s.goal_qpos[lane] = q;
  }
  s.is_safe = 1;
}

// ================================================================================================ contacts
#if HRG_STACK || HRG_HAMMER || HRG_LIFT
// Contacts of two boxes with half extents ha / hb (centres pa / pb, rotations Ra / Rb row-major in LDS; the stacking task's cubes share one h): separating-axis test over the 15 axes, then
// the reference face's rectangle clipped against the incident face (candidates: incident vertices, rectangle corners under the incident face, edge
// crossings; at most four penetrating candidates that span the patch are kept) or one edge-edge contact.  Restated as in
// oracle/hrg_oracle.c box_box; one lane runs one pair.  Normal from box a to box b.  Returns the number of contacts (<= 4).
// near-ties between separating axes / candidate depths go to the earlier one unless the later wins by this margin (1 nm)
#define BB_TIE 1e-9
struct BBContact { double pos[3], n[3], dist; };
// T: 144 doubles of the caller's LDS scratch -- the candidate table (72) and the function's index-addressed arrays (axes, their products, the clipping frame: as
// private arrays they live in scratch memory, and a single lane's scratch stores cost a partial cache line each: 5.6 KB of HBM writes per call)
#define BB_WORK 152   // (138 used by box_box2; the rest is the caller's: the lifting task keeps the slab's and the board's extents there)
DI int box_box2(const double* pa, const double* Ra, const double* ha, const double* pb, const double* Rb, const double* hb, BBContact* out, double* T) {
  double (*A)[3] = (double (*)[3])(T + 72), (*B)[3] = (double (*)[3])(T + 81), (*C)[3] = (double (*)[3])(T + 90), (*AC)[3] = (double (*)[3])(T + 99);
  double t[3], ta[3], tb[3];
  v3sub(t, pb, pa);
  for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) { A[i][k] = Ra[3 * k + i]; B[i][k] = Rb[3 * k + i]; }
  for (int i = 0; i < 3; i++) { ta[i] = v3dot(t, A[i]); tb[i] = v3dot(t, B[i]); for (int j = 0; j < 3; j++) { C[i][j] = v3dot(A[i], B[j]); AC[i][j] = fabs(C[i][j]); } }
  double sf = -1e300, se = -1e300;
  int bf = 0, be = -1;
  for (int i = 0; i < 3; i++) {
    const double s_ = fabs(ta[i]) - (ha[i] + hb[0] * AC[i][0] + hb[1] * AC[i][1] + hb[2] * AC[i][2]);
    if (s_ > 0) return 0;
    if (s_ > sf + BB_TIE) { sf = s_; bf = i; }
  }
  for (int j = 0; j < 3; j++) {
    const double s_ = fabs(tb[j]) - (hb[j] + ha[0] * AC[0][j] + ha[1] * AC[1][j] + ha[2] * AC[2][j]);
    if (s_ > 0) return 0;
    if (s_ > sf + BB_TIE) { sf = s_; bf = 3 + j; }
  }
#pragma unroll 1
  for (int ij = 0; ij < 9; ij++) {
    const int i = ij / 3, j = ij - 3 * i, i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    const double l2 = 1.0 - C[i][j] * C[i][j];
    if (l2 < 1e-12) continue;
    const double l = fsqrt(l2);
    const double tl = ta[i2] * C[i1][j] - ta[i1] * C[i2][j];
    const double s_ = (fabs(tl) - (ha[i1] * AC[i2][j] + ha[i2] * AC[i1][j] + hb[j1] * AC[i][j2] + hb[j2] * AC[i][j1])) / l;
    if (s_ > 0) return 0;
    if (s_ > se + BB_TIE) { se = s_; be = ij; }
  }
  if (be >= 0 && se * 1.05 > sf) {
    const int i = be / 3, j = be - 3 * i;
    double n[3], pA[3], pB[3], d[3];
    v3cross(n, A[i], B[j]);
    v3scl(n, n, 1.0 / v3norm(n));
    if (v3dot(n, t) < 0) v3scl(n, n, -1.0);
    v3cpy(pA, pa); v3cpy(pB, pb);
    for (int k = 0; k < 3; k++) {
      if (k != i) v3madd(pA, pA, A[k], (v3dot(n, A[k]) > 0 ? 1.0 : -1.0) * ha[k]);
      if (k != j) v3madd(pB, pB, B[k], (v3dot(n, B[k]) > 0 ? -1.0 : 1.0) * hb[k]);
    }
    v3sub(d, pB, pA);
    const double uaub = C[i][j], q1 = v3dot(A[i], d), q2 = -v3dot(B[j], d), den = 1.0 - uaub * uaub;
    const double al = (q1 + uaub * q2) / den, be_ = (uaub * q1 + q2) / den;
    double xa[3], xb[3];
    v3madd(xa, pA, A[i], al);
    v3madd(xb, pB, B[j], be_);
    for (int k = 0; k < 3; k++) out[0].pos[k] = 0.5 * (xa[k] + xb[k]);
    v3cpy(out[0].n, n);
    out[0].dist = se;
    return 1;
  }
  const bool refA = bf < 3;
  const int r = refA ? bf : bf - 3, r1 = (r + 1) % 3, r2 = (r + 2) % 3;
  double (*Rf)[3] = (double (*)[3])(T + 108), (*In)[3] = (double (*)[3])(T + 117);
  double pr[3], pi_[3];
  for (int a = 0; a < 3; a++) for (int k = 0; k < 3; k++) { Rf[a][k] = refA ? A[a][k] : B[a][k]; In[a][k] = refA ? B[a][k] : A[a][k]; }
  for (int k = 0; k < 3; k++) { pr[k] = refA ? pa[k] : pb[k]; pi_[k] = refA ? pb[k] : pa[k]; }
  const double* hr = refA ? ha : hb;   // half extents of the reference / the incident box
  const double* hi = refA ? hb : ha;
  const double sg = refA ? (ta[r] >= 0 ? 1.0 : -1.0) : (tb[r] >= 0 ? -1.0 : 1.0);
  double nr[3], cr[3], ci[3];
  v3scl(nr, Rf[r], sg);
  v3madd(cr, pr, nr, hr[r]);
  int k = 0;
  double best = -1;
  for (int q = 0; q < 3; q++) { const double c_ = fabs(v3dot(In[q], nr)); if (c_ > best) { best = c_; k = q; } }
  const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
  const double si = v3dot(In[k], nr) > 0 ? -1.0 : 1.0;
  v3madd(ci, pi_, In[k], si * hi[k]);
  const double hu = hr[r1], hv = hr[r2];
  const double S1[4] = {1, -1, -1, 1}, S2[4] = {1, 1, -1, -1};
  double *vu = T + 126, *vv = T + 130, *vd = T + 134;
  for (int q = 0; q < 4; q++) {
    double x[3], d[3];
    v3madd(x, ci, In[k1], S1[q] * hi[k1]);
    v3madd(x, x, In[k2], S2[q] * hi[k2]);
    v3sub(d, x, cr);
    vu[q] = v3dot(d, Rf[r1]); vv[q] = v3dot(d, Rf[r2]); vd[q] = v3dot(d, nr);
  }
  // candidate table (u, v, depth; depth >= 0 marks "not a penetrating candidate") in the caller's LDS scratch: T[0..23] u, T[24..47] v, T[48..71] depth
  for (int q = 0; q < 24; q++) T[48 + q] = 1.0;
  auto offer = [&](int idx, double u, double v, double dpt) { if (dpt < 0) { T[idx] = u; T[24 + idx] = v; T[48 + idx] = dpt; } };
  for (int q = 0; q < 4; q++)
    if (fabs(vu[q]) <= hu && fabs(vv[q]) <= hv) offer(q, vu[q], vv[q], vd[q]);
  {
    double d0[3];
    v3sub(d0, ci, cr);
    const double c0u = v3dot(d0, Rf[r1]), c0v = v3dot(d0, Rf[r2]), c0d = v3dot(d0, nr);
    const double e1u = hi[k1] * v3dot(In[k1], Rf[r1]), e1v = hi[k1] * v3dot(In[k1], Rf[r2]), e1d = hi[k1] * v3dot(In[k1], nr);
    const double e2u = hi[k2] * v3dot(In[k2], Rf[r1]), e2v = hi[k2] * v3dot(In[k2], Rf[r2]), e2d = hi[k2] * v3dot(In[k2], nr);
    const double det = e1u * e2v - e1v * e2u;
    if (fabs(det) > 1e-12 * hu * hv)
      for (int q = 0; q < 4; q++) {
        const double pu = S1[q] * hu - c0u, pv = S2[q] * hv - c0v;
        const double al = (pu * e2v - pv * e2u) / det, be_ = (e1u * pv - e1v * pu) / det;
        if (fabs(al) <= 1 && fabs(be_) <= 1) offer(4 + q, S1[q] * hu, S2[q] * hv, c0d + al * e1d + be_ * e2d);
      }
  }
#pragma unroll 1
  for (int q = 0; q < 4; q++) {
    const int q1 = (q + 1) & 3;
    const double du = vu[q1] - vu[q], dv = vv[q1] - vv[q], dd = vd[q1] - vd[q];
#pragma unroll 1
    for (int e = 0; e < 4; e++) {
      const double lim = (e & 1) ? -1.0 : 1.0;
      if (e < 2) {
        if (fabs(du) < 1e-14) continue;
        const double tt = (lim * hu - vu[q]) / du, w = vv[q] + tt * dv;
        if (tt > 0 && tt < 1 && fabs(w) < hv) offer(8 + 4 * q + e, lim * hu, w, vd[q] + tt * dd);
      } else {
        if (fabs(dv) < 1e-14) continue;
        const double tt = (lim * hv - vv[q]) / dv, w = vu[q] + tt * du;
        if (tt > 0 && tt < 1 && fabs(w) < hu) offer(8 + 4 * q + e, w, lim * hv, vd[q] + tt * dd);
      }
    }
  }
  // keep at most four that span the patch: the deepest, the one farthest from it, the farthest from their line on either side (ties -> lowest index)
  int pick[4], np_ = 0;
  const double eps2 = 1e-12 * (hu * hu + hv * hv);
  {
    int arg = -1;
    double bd = 0;
#pragma unroll 1
    for (int q = 0; q < 24; q++) { const double dq = T[48 + q]; if (dq < 0 && (arg < 0 || dq < bd - BB_TIE)) { arg = q; bd = dq; } }
    if (arg >= 0) pick[np_++] = arg;
  }
  if (np_ == 1) {
    int arg = -1;
    double bestv = eps2;
    const double u0 = T[pick[0]], v0 = T[24 + pick[0]];
#pragma unroll 1
    for (int q = 0; q < 24; q++) {
      if (!(T[48 + q] < 0)) continue;
      const double du = T[q] - u0, dv = T[24 + q] - v0, val = du * du + dv * dv;
      if (val > bestv * (1 + 1e-9)) { arg = q; bestv = val; }
    }
    if (arg >= 0) pick[np_++] = arg;
  }
  if (np_ == 2) {
    const double u0 = T[pick[0]], v0 = T[24 + pick[0]], lu = T[pick[1]] - u0, lv = T[24 + pick[1]] - v0, epsc = fsqrt(eps2 * (lu * lu + lv * lv));
    int argp = -1, argn = -1;
    double bp = epsc, bn = epsc;
#pragma unroll 1
    for (int q = 0; q < 24; q++) {
      if (!(T[48 + q] < 0)) continue;
      const double cr_ = lu * (T[24 + q] - v0) - lv * (T[q] - u0);
      if (cr_ > bp * (1 + 1e-9)) { argp = q; bp = cr_; }
      if (-cr_ > bn * (1 + 1e-9)) { argn = q; bn = -cr_; }
    }
    if (argp >= 0) pick[np_++] = argp;
    if (argn >= 0) pick[np_++] = argn;
  }
  for (int z = 0; z < np_; z++) {
    const double u = T[pick[z]], v = T[24 + pick[z]], dpt = T[48 + pick[z]];
    for (int a = 0; a < 3; a++) {
      out[z].pos[a] = cr[a] + u * Rf[r1][a] + v * Rf[r2][a] + 0.5 * dpt * nr[a];
      out[z].n[a] = refA ? nr[a] : -nr[a];
    }
    out[z].dist = dpt;
  }
  return np_;
}
DI int box_box(const double* pa, const double* Ra, const double* pb, const double* Rb, const double* h, BBContact* out, double* T) { return box_box2(pa, Ra, h, pb, Rb, h, out, T); }
#endif
#if HRG_STACK

// The cubes' part of the stacking task's contact list (after the robot's own rounds): robot capsule x cube first points (cube-major), table corners,
// floor corners, cube pairs (a < b; one lane per pair), second points of capsules lying along a face; and the object_gripped sensor.
DI void collide_cubes(const DevModel* __restrict__ dm_, int lane, int* base_io) {
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  hrg_stack_state& sk = L.sk;
  int base = *base_io;
  const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  if (lane < NCUBE) { double Rm[9]; quat2mat(Rm, sk.quat[lane]); for (int a = 0; a < 9; a++) L.cR[lane][a] = Rm[a]; }
  wave_sync();
  const double hb[3] = {m.box_half[0], m.box_half[1], m.box_half[2]};
  const double circ2 = hb[0] * hb[0] + hb[1] * hb[1] + hb[2] * hb[2];
  bool f0a = false, f1a = false, f0b = false, f1b = false;
#pragma unroll 1
  for (int round = 0; round < 4; round++) {
    Contact c;
    c.g1 = c.g2 = c.b1 = c.b2 = 0; c.dist = 0; v3set(c.n, 0, 0, 1); v3set(c.pos, 0, 0, 0);
    bool hit = false;
    if (round == 0 || round == 3) {   // capsule i x cube cb: first point / second point (a capsule lying along a face)
      const bool second = round == 3;
      const int cb = lane / HRG_NRCAP, i = lane - cb * HRG_NRCAP;
      bool near = false;
      if (lane < NCUBE * HRG_NRCAP && m.rcap_body[i] >= 0) {
        double dc[3];
        for (int a = 0; a < 3; a++) dc[a] = 0.5 * (L.rcapw[i][a] + L.rcapw[i][3 + a]) - sk.pos[cb][a];
        const double reach = dm->rcap_hl[i] + m.rcap_r[i] + fsqrt(circ2) + 1e-9;
        near = v3dot(dc, dc) <= reach * reach;
      }
      if (__any(near) && near) {
        double cs[3], cbp[3];
        const double e2 = seg_box(&L.rcapw[i][0], &L.rcapw[i][3], sk.pos[cb], L.cR[cb], hb, cs, cbp);
        double dd = fsqrt(e2), dist = dd - m.rcap_r[i];
        if (dist < 0) {
          double s2[3], b2[3];
          const bool two = dd > 1e-9 && cap_box_two(&L.rcapw[i][0], &L.rcapw[i][3], sk.pos[cb], L.cR[cb], hb, m.rcap_r[i], cs, cbp, second ? 1 : 0, s2, b2);
          if (two) { v3cpy(cs, s2); v3cpy(cbp, b2); double dv[3]; v3sub(dv, cbp, cs); dd = v3norm(dv); dist = dd - m.rcap_r[i]; }
          hit = two || !second;
          if (dd > 1e-9) { v3sub(c.n, cbp, cs); v3scl(c.n, c.n, 1.0 / dd); }
          else {
            double loc[3], rel[3], best = 1e300;
            int ax = 0;
            v3sub(rel, cs, sk.pos[cb]);
            for (int a = 0; a < 3; a++) { loc[a] = L.cR[cb][a] * rel[0] + L.cR[cb][3 + a] * rel[1] + L.cR[cb][6 + a] * rel[2]; if (hb[a] - fabs(loc[a]) < best) { best = hb[a] - fabs(loc[a]); ax = a; } }
            const double sg = loc[ax] >= 0 ? -1.0 : 1.0;
            for (int a = 0; a < 3; a++) c.n[a] = sg * L.cR[cb][3 * a + ax];
            dist = -best - m.rcap_r[i];
          }
          v3madd(c.pos, cs, c.n, m.rcap_r[i] + 0.5 * dist);
          c.g1 = i; c.g2 = GEOM_BOX + cb; c.b1 = m.rcap_body[i]; c.b2 = BODY_BOX + cb; c.dist = dist;
        }
      }
      const uint64_t mask = __ballot(hit);
      const int slot = base + __popcll(mask & lt);
      if (hit) {
        if (slot < NCON_DYN) L.con[slot] = c;
        if (slot < HRG_NCON_MAX) { L.st.con_pairs[slot][0] = c.g1; L.st.con_pairs[slot][1] = c.g2; }
      }
      if (!second) {   // object_gripped (1467-1481): both fingers on cube A, or both on cube B (contacts beyond the reported list do not count)
        const bool rep = hit && slot < HRG_NCON_MAX;
        f0a = __any(rep && lane == HRG_CUBE_A * HRG_NRCAP + HRG_NRCAP - 2); f1a = __any(rep && lane == HRG_CUBE_A * HRG_NRCAP + HRG_NRCAP - 1);
        f0b = __any(rep && lane == HRG_CUBE_B * HRG_NRCAP + HRG_NRCAP - 2); f1b = __any(rep && lane == HRG_CUBE_B * HRG_NRCAP + HRG_NRCAP - 1);
      }
      base += __popcll(mask);
    } else if (round == 1) {   // lanes 0..31: table x corner cn of cube cb, lanes 32..63: floor
      const int pl = lane >> 5, cb = (lane >> 3) & 3, cn = lane & 7;
      const double loc[3] = {(cn & 1) ? hb[0] : -hb[0], (cn & 2) ? hb[1] : -hb[1], (cn & 4) ? hb[2] : -hb[2]};
      double p[3];
      m3mulv(p, L.cR[cb], loc);
      v3add(p, p, sk.pos[cb]);
      const double z0 = pl ? m.floor_z : m.table_top_z, dist = p[2] - z0;
      bool ok = true;
      if (pl == 0) ok = fabs(p[0] - m.table_center[0]) <= m.table_half[0] && fabs(p[1] - m.table_center[1]) <= m.table_half[1] && p[2] > z0 - 0.05;
      if (ok && dist < 0) {
        hit = true;
        v3set(c.n, 0, 0, 1);
        v3set(c.pos, p[0], p[1], z0 + 0.5 * dist);
        c.g1 = pl ? GEOM_FLOOR : GEOM_TABLE; c.g2 = GEOM_BOX + cb; c.b1 = -1; c.b2 = BODY_BOX + cb; c.dist = dist;
      }
      const uint64_t mask = __ballot(hit);
      if (hit) {
        const int slot = base + __popcll(mask & lt);
        if (slot < NCON_DYN) L.con[slot] = c;
        if (slot < HRG_NCON_MAX) { L.st.con_pairs[slot][0] = c.g1; L.st.con_pairs[slot][1] = c.g2; }
      }
      base += __popcll(mask);
    } else {   // cube pairs: lane p < 6 runs pair (a, b)
      int nc = 0;
      BBContact bc[4];
      int pa_ = 0, pb_ = 1;
      if (lane < 6) {
        const int PA[6] = {0, 0, 0, 1, 1, 2}, PB[6] = {1, 2, 3, 2, 3, 3};
        pa_ = PA[lane]; pb_ = PB[lane];
        double d[3];
        v3sub(d, sk.pos[pb_], sk.pos[pa_]);
        // candidate scratch: the tail of the (dead) solver rows; the collide arrays (hcap, rcapw, cur) sit in the first 1.7 KB of the same union
        static_assert(40 * 21 + 6 * BB_WORK <= 4 * NCON_DYN * 21, "box_box2 workspace of the six cube pairs");
        if (!(v3dot(d, d) > 4.0 * circ2)) nc = box_box(sk.pos[pa_], L.cR[pa_], sk.pos[pb_], L.cR[pb_], hb, bc, &L.Jc[40][0] + BB_WORK * lane);
      }
      // exclusive prefix of the contact counts over lanes 0..5
      int pre = 0, tot = 0;
      for (int q = 0; q < 6; q++) { const int nq = __shfl(nc, q, 64); if (q < lane) pre += nq; tot += nq; }
      for (int q = 0; q < nc; q++) {
        const int slot = base + pre + q;
        c.g1 = GEOM_BOX + pa_; c.g2 = GEOM_BOX + pb_; c.b1 = BODY_BOX + pa_; c.b2 = BODY_BOX + pb_; c.dist = bc[q].dist;
        v3cpy(c.n, bc[q].n); v3cpy(c.pos, bc[q].pos);
        if (slot < NCON_DYN) L.con[slot] = c;
        if (slot < HRG_NCON_MAX) { L.st.con_pairs[slot][0] = c.g1; L.st.con_pairs[slot][1] = c.g2; }
      }
      base += tot;
    }
  }
  sk.gripped = (f0a && f1a) || (f0b && f1b);
  *base_io = base;
}
#endif
#if HRG_HAMMER
// world poses of the hammering task's collision geoms, the nail_head body origin and the slide axis -> L.gR, L.gc, L.nail_org, L.nail_axis (oracle: hammer_geometry)
DI void hammer_geometry(const DevModel* __restrict__ dm_, int lane) {
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  hrg_hammer_state& hm = L.hm;
  wave_sync();
  if (lane < 2) { double Rm[9]; const double q[4] = {hm.quat[lane][0], hm.quat[lane][1], hm.quat[lane][2], hm.quat[lane][3]}; quat2mat(Rm, q); for (int a = 0; a < 9; a++) L.gR[lane][a] = Rm[a]; }
  wave_sync();
  if (lane < 3) {
    const int a = lane;
    L.gc[HRG_HG_BOARD][a] = hm.pos[0][a];
    double th = 0, te = 0, tn = 0, tg = 0;
    for (int k = 0; k < 3; k++) {
      th += L.gR[1][3 * a + k] * m.hm_geom_pos[HRG_HG_HANDLE][k];
      te += L.gR[1][3 * a + k] * m.hm_geom_pos[HRG_HG_HEAD][k];
      const double lk = k == 0 ? hm.nail_xy[0] : (k == 1 ? hm.nail_xy[1] : m.hm_nail_z0 - hm.nail_q);
      tn += L.gR[0][3 * a + k] * lk;
      tg += L.gR[0][3 * a + k] * m.hm_geom_pos[HRG_HG_NAIL][k];
    }
    L.gc[HRG_HG_HANDLE][a] = hm.pos[1][a] + th;
    L.gc[HRG_HG_HEAD][a] = hm.pos[1][a] + te;
    const double org = hm.pos[0][a] + tn;
    L.nail_org[a] = org;
    L.gc[HRG_HG_NAIL][a] = org + tg;
    L.nail_axis[a] = -L.gR[0][3 * a + 2];
  }
  wave_sync();
}
// body_xpos of board_main, the hammer's root body and nail_head (what the observables read) from the current geometry
DI void hammer_obs_pos(const DevModel* __restrict__ dm_, int lane) {
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  hrg_hammer_state& hm = L.hm;
  if (lane < 3) {
    double t = 0;
    for (int k = 0; k < 3; k++) t += L.gR[1][3 * lane + k] * m.hm_hammer_com[k];
    hm.obs_pos[0][lane] = hm.pos[0][lane];
    hm.obs_pos[1][lane] = hm.pos[1][lane] - t;
    hm.obs_pos[2][lane] = L.nail_org[lane];
  }
}
// The hammering task's part of the contact list (after the robot's own rounds; oracle: collide_hammer): robot capsule x {board, handle, head, nail head} first points
// (geom-major; the handle meets the two finger bars only), table corners, floor corners, the box pairs head - nail, handle - nail, head - board, handle - board
// (one lane per pair), second points of capsules lying along a face; and the hammer_gripped sensor.
DI void collide_hammer(const DevModel* __restrict__ dm_, int lane, int* base_io) {
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  hrg_hammer_state& hm = L.hm;
  int base = *base_io;
  const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  hammer_geometry(dm_, lane);
  bool f0 = false, f1 = false;
#pragma unroll 1
  for (int round = 0; round < 4; round++) {
    Contact c;
    c.g1 = c.g2 = c.b1 = c.b2 = 0; c.dist = 0; v3set(c.n, 0, 0, 1); v3set(c.pos, 0, 0, 0);
    bool hit = false;
    if (round == 0 || round == 3) {   // capsule i x geom g: first point / second point (a capsule lying along a face)
      const bool second = round == 3;
      const int g = lane < HRG_HM_NGEOM * HRG_NRCAP ? lane / HRG_NRCAP : 0, i = lane - g * HRG_NRCAP < HRG_NRCAP ? lane - g * HRG_NRCAP : 0;
      const int fb = (g == HRG_HG_HANDLE || g == HRG_HG_HEAD) ? 1 : 0;   // rotation: the nail head turns with the board
      const double hb[3] = {m.hm_geom_half[g][0], m.hm_geom_half[g][1], m.hm_geom_half[g][2]};
      bool near = false;
      if (lane < HRG_HM_NGEOM * HRG_NRCAP && m.rcap_body[i] >= 0 && !(g == HRG_HG_HANDLE && i < HRG_NRCAP - 2)) {
        // broadphase: the capsule's bounding sphere against the box itself (distance of its centre to the box in the box frame); conservative
        double dc[3], d2 = 0;
        for (int a = 0; a < 3; a++) dc[a] = 0.5 * (L.rcapw[i][a] + L.rcapw[i][3 + a]) - L.gc[g][a];
        for (int a = 0; a < 3; a++) {
          const double la = fabs(L.gR[fb][a] * dc[0] + L.gR[fb][3 + a] * dc[1] + L.gR[fb][6 + a] * dc[2]) - hb[a];
          if (la > 0) d2 += la * la;
        }
        const double reach = dm->rcap_hl[i] + m.rcap_r[i] + 1e-9;
        near = d2 <= reach * reach;
      }
      if (__any(near) && near) {
        double cs[3], cbp[3];
        const double e2 = seg_box(&L.rcapw[i][0], &L.rcapw[i][3], L.gc[g], L.gR[fb], hb, cs, cbp);
        double dd = fsqrt(e2), dist = dd - m.rcap_r[i];
        if (dist < 0) {
          double s2[3], b2[3];
          const bool two = dd > 1e-9 && cap_box_two(&L.rcapw[i][0], &L.rcapw[i][3], L.gc[g], L.gR[fb], hb, m.rcap_r[i], cs, cbp, second ? 1 : 0, s2, b2);
          if (two) { v3cpy(cs, s2); v3cpy(cbp, b2); double dv[3]; v3sub(dv, cbp, cs); dd = v3norm(dv); dist = dd - m.rcap_r[i]; }
          hit = two || !second;
          if (dd > 1e-9) { v3sub(c.n, cbp, cs); v3scl(c.n, c.n, 1.0 / dd); }
          else {
            double loc[3], rel[3], best = 1e300;
            int ax = 0;
            v3sub(rel, cs, L.gc[g]);
            for (int a = 0; a < 3; a++) { loc[a] = L.gR[fb][a] * rel[0] + L.gR[fb][3 + a] * rel[1] + L.gR[fb][6 + a] * rel[2]; if (hb[a] - fabs(loc[a]) < best) { best = hb[a] - fabs(loc[a]); ax = a; } }
            const double sg = loc[ax] >= 0 ? -1.0 : 1.0;
            for (int a = 0; a < 3; a++) c.n[a] = sg * L.gR[fb][3 * a + ax];
            dist = -best - m.rcap_r[i];
          }
          v3madd(c.pos, cs, c.n, m.rcap_r[i] + 0.5 * dist);
          c.g1 = i; c.g2 = GEOM_BOX + g; c.b1 = m.rcap_body[i]; c.b2 = BODY_BOX + (g == HRG_HG_NAIL ? HRG_HM_NAIL : fb); c.dist = dist;
        }
      }
      const uint64_t mask = __ballot(hit);
      const int slot = base + __popcll(mask & lt);
      if (hit) {
        if (slot < NCON_DYN) L.con[slot] = c;
        if (slot < HRG_NCON_MAX) { L.st.con_pairs[slot][0] = c.g1; L.st.con_pairs[slot][1] = c.g2; }
      }
      if (!second) {   // hammer_gripped (1283-1289): both fingers touch a geom of the hammer (contacts beyond the reported list do not count)
        const bool rep = hit && slot < HRG_NCON_MAX && fb == 1;
        f0 = __any(rep && i == HRG_NRCAP - 2); f1 = __any(rep && i == HRG_NRCAP - 1);
      }
      base += __popcll(mask);
    } else if (round == 1) {   // lanes 0..23: table x corner cn of geom g (board, handle, head), lanes 24..47: floor
      if (lane < 48) {
        const int pl = lane / 24, g = (lane - 24 * pl) >> 3, cn = lane & 7, fb = g == HRG_HG_BOARD ? 0 : 1;
        const double hb[3] = {m.hm_geom_half[g][0], m.hm_geom_half[g][1], m.hm_geom_half[g][2]};
        const double loc[3] = {(cn & 1) ? hb[0] : -hb[0], (cn & 2) ? hb[1] : -hb[1], (cn & 4) ? hb[2] : -hb[2]};
        double p[3];
        m3mulv(p, L.gR[fb], loc);
        v3add(p, p, L.gc[g]);
        const double z0 = pl ? m.floor_z : m.table_top_z, dist = p[2] - z0;
        bool ok = true;
        if (pl == 0) ok = fabs(p[0] - m.table_center[0]) <= m.table_half[0] && fabs(p[1] - m.table_center[1]) <= m.table_half[1] && p[2] > z0 - 0.05;
        if (ok && dist < 0) {
          hit = true;
          v3set(c.n, 0, 0, 1);
          v3set(c.pos, p[0], p[1], z0 + 0.5 * dist);
          c.g1 = pl ? GEOM_FLOOR : GEOM_TABLE; c.g2 = GEOM_BOX + g; c.b1 = -1; c.b2 = BODY_BOX + fb; c.dist = dist;
        }
      }
      const uint64_t mask = __ballot(hit);
      if (hit) {
        const int slot = base + __popcll(mask & lt);
        if (slot < NCON_DYN) L.con[slot] = c;
        if (slot < HRG_NCON_MAX) { L.st.con_pairs[slot][0] = c.g1; L.st.con_pairs[slot][1] = c.g2; }
      }
      base += __popcll(mask);
    } else {   // box pairs: lane p < 4 runs (head | handle) x (nail | board)
      int nc = 0;
      BBContact bc[4];
      int ga = HRG_HG_HEAD, gb = HRG_HG_NAIL;
      if (lane < 4) {
        ga = (lane & 1) ? HRG_HG_HANDLE : HRG_HG_HEAD; gb = lane < 2 ? HRG_HG_NAIL : HRG_HG_BOARD;
        const double ha[3] = {m.hm_geom_half[ga][0], m.hm_geom_half[ga][1], m.hm_geom_half[ga][2]}, hb[3] = {m.hm_geom_half[gb][0], m.hm_geom_half[gb][1], m.hm_geom_half[gb][2]};
        // broadphase: the circumsphere of the hammer's box against the other box itself (conservative: separated boxes give no contacts anyway)
        double d[3], d2 = 0;
        v3sub(d, L.gc[ga], L.gc[gb]);
        for (int a = 0; a < 3; a++) {
          const double la = fabs(L.gR[0][a] * d[0] + L.gR[0][3 + a] * d[1] + L.gR[0][6 + a] * d[2]) - hb[a];
          if (la > 0) d2 += la * la;
        }
        const double ra = fsqrt(ha[0] * ha[0] + ha[1] * ha[1] + ha[2] * ha[2]) + 1e-9;
        // candidate scratch: the tail of the (dead) solver rows; the collide arrays (hcap, rcapw, cur) sit in the first 1.7 KB of the same union
        static_assert(40 * (NVS + 1) + 4 * BB_WORK <= (4 * NCON_DYN + HROW_NEQ) * (NVS + 1), "box_box2 workspace of the four box pairs");
        if (!(d2 > ra * ra)) nc = box_box2(L.gc[ga], L.gR[1], ha, L.gc[gb], L.gR[0], hb, bc, &L.Jc[40][0] + BB_WORK * lane);
      }
      int pre = 0, tot = 0;
      for (int q = 0; q < 4; q++) { const int nq = __shfl(nc, q, 64); if (q < lane) pre += nq; tot += nq; }
      for (int q = 0; q < nc; q++) {
        const int slot = base + pre + q;
        c.g1 = GEOM_BOX + ga; c.g2 = GEOM_BOX + gb; c.b1 = BODY_BOX + HRG_HM_HAMMER; c.b2 = BODY_BOX + (gb == HRG_HG_NAIL ? HRG_HM_NAIL : HRG_HM_BOARD); c.dist = bc[q].dist;
        v3cpy(c.n, bc[q].n); v3cpy(c.pos, bc[q].pos);
        if (slot < NCON_DYN) L.con[slot] = c;
        if (slot < HRG_NCON_MAX) { L.st.con_pairs[slot][0] = c.g1; L.st.con_pairs[slot][1] = c.g2; }
      }
      base += tot;
    }
  }
  hm.gripped = f0 && f1;
  *base_io = base;
}
#endif
// Stand-in for mj_collision (bounding capsules, table top face, floor plane), pair order = contact order.
PH_COLLIDE void collide(const DevModel* __restrict__ dm_, int lane, int* ncon_out) {
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  if (lane < HRG_NRCAP) {
    const int c = lane, b = m.rcap_body[c];
    double R[9], p[3], t[3], mid[3];
    if (b < 0) { for (int a = 0; a < 9; a++) R[a] = dm->Rbase[a]; v3cpy(p, m.base_pos); }
    else { for (int a = 0; a < 9; a++) R[a] = L.kR[b][a]; v3cpy(p, L.kp[b]); }
    m3mulv(t, R, m.rcap_p1[c]); v3add(&L.rcapw[c][0], p, t);
    m3mulv(t, R, m.rcap_p2[c]); v3add(&L.rcapw[c][3], p, t);
    for (int a = 0; a < 3; a++) mid[a] = 0.5 * (m.rcap_p1[c][a] + m.rcap_p2[c][a]);
    m3mulv(t, R, mid);
    v3add(L.rcen[c], p, t);
  }
  wave_sync();
  // Cull of the 240 robot-human pairs: the box around all robot capsules, each inflated by its radius and the human contact margin, against the bounding sphere
  // of every human capsule.  A human capsule whose sphere misses the box cannot be in contact with any robot capsule (conservative); the capsules that come near
  // (none while the human stands clear of the arm, a forearm and a hand when it reaches in) are listed in LDS, and the pair rounds below run over
  // robot capsule x listed human capsule only: one round of 64 lanes for up to six of them instead of four rounds over all 240 pairs.  The pairs keep their
  // order (robot capsule major, human capsule minor), so the contact list is the one of the full enumeration.
  int n_hnear;
  {
    double bl[3], bh[3];
    if (lane < HRG_NRCAP) {
      const double rr = m.rcap_r[lane] + m.contact_margin_human + 1e-9;
      for (int a = 0; a < 3; a++) {
        const double p1 = L.rcapw[lane][a], p2 = L.rcapw[lane][3 + a];
        bl[a] = (p1 < p2 ? p1 : p2) - rr; bh[a] = (p1 > p2 ? p1 : p2) + rr;
      }
    } else for (int a = 0; a < 3; a++) { bl[a] = 1e300; bh[a] = -1e300; }
    double d2 = 0;
    const int hb = lane < HRG_NHB ? lane : 0;
    for (int a = 0; a < 3; a++) {
      const double hi = lane_value<0>(row16_max(bh[a])), lo = -lane_value<0>(row16_max(-bl[a]));
      const double c = 0.5 * (L.hcap[hb][a] + L.hcap[hb][3 + a]);
      const double e = c < lo ? lo - c : (c > hi ? c - hi : 0.0);
      d2 += e * e;
    }
    const double rad = dm->hcap_hl[hb] + m.hcap_r[hb] + 1e-9;
    const bool hn = lane < HRG_NHB && d2 <= rad * rad;
    const uint64_t hmask = __ballot(hn);
    n_hnear = __popcll(hmask);
    if (hn) L.hnear[__popcll(hmask & ((1ull << lane) - 1))] = lane;
    if (n_hnear) wave_sync();
  }
  int base = 0;
  const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const int n_hpairs = HRG_NRCAP * n_hnear, n_hrounds = (n_hpairs + 63) >> 6;
  const float inv_hnear = n_hnear ? 1.0f / (float)n_hnear : 0.0f;
  // rounds: 0 = robot-robot, 1..n_hrounds = robot-human (the listed capsules), last = planes
#pragma unroll 1
  for (int round = 0; round < 2 + n_hrounds; round++) {
    bool hit = false;
    Contact c;
    c.g1 = c.g2 = c.b1 = c.b2 = 0; c.dist = 0; v3set(c.n, 0, 0, 1); v3set(c.pos, 0, 0, 0);
    if (round <= n_hrounds) {
      int i = 0, g2 = 0;
      double r2 = 0, margin = 0, hl2 = 0;
      const double *a1, *a2;
      bool valid;
      if (round == 0) {
        valid = lane < dm->n_self;
        i = valid ? dm->self_i[lane] : 0;
        const int j = valid ? dm->self_j[lane] : 1;
        g2 = j; a1 = &L.rcapw[j][0]; a2 = &L.rcapw[j][3]; r2 = m.rcap_r[j]; hl2 = dm->rcap_hl[j];
        c.b2 = m.rcap_body[j];
      } else {
        const int q = (round - 1) * 64 + lane;
        valid = q < n_hpairs;
        i = valid ? (int)(((float)q + 0.5f) * inv_hnear) : 0;   // q / n_hnear (exact: q <= 240, the quotient's distance to an integer is at least 0.5 / 24)
        const int hb = valid ? L.hnear[q - i * n_hnear] : 0;
        g2 = GEOM_HUMAN0 + hb; a1 = &L.hcap[hb][0]; a2 = &L.hcap[hb][3]; r2 = m.hcap_r[hb]; hl2 = dm->hcap_hl[hb];
        margin = m.contact_margin_human;
        c.b2 = -2;
      }
      // broadphase: bounding spheres of the two capsules (conservative: a culled pair cannot be a contact); the whole
      // wave skips the narrowphase when no lane survives, which is the common case
      bool near = false;
      if (valid) {
        double ca[3], cb[3], dc[3];
        for (int a = 0; a < 3; a++) { ca[a] = 0.5 * (L.rcapw[i][a] + L.rcapw[i][3 + a]); cb[a] = 0.5 * (a1[a] + a2[a]); }
        v3sub(dc, ca, cb);
        // half lengths are model constants (rigid capsules); 1e-9 m of slack covers the rounding of the world-frame end points
        const double reach = dm->rcap_hl[i] + hl2 + m.rcap_r[i] + r2 + margin + 1e-9;
        near = v3dot(dc, dc) <= reach * reach;
      }
      if (__any(near) && valid && near) {
        double c1[3], c2[3], d[3];
        const double d2 = seg_seg(&L.rcapw[i][0], &L.rcapw[i][3], a1, a2, c1, c2), dd = fsqrt(d2), dist = dd - m.rcap_r[i] - r2;
        if (dist < margin) {
          hit = true;
          v3sub(d, c2, c1);
          if (dd > 1e-12) v3scl(c.n, d, 1.0 / dd);
          v3madd(c.pos, c1, c.n, m.rcap_r[i] + 0.5 * dist);
          c.g1 = i; c.g2 = g2; c.b1 = m.rcap_body[i]; c.dist = dist;
        }
      }
    } else if (lane < 4 * HRG_NRCAP) {
      const int pl = lane / (2 * HRG_NRCAP), i = (lane % (2 * HRG_NRCAP)) >> 1, en = lane & 1;
      if (m.rcap_body[i] >= 0) {
        const double* p = &L.rcapw[i][3 * en];
        const double z0 = pl ? m.floor_z : m.table_top_z, dist = p[2] - m.rcap_r[i] - z0;
        bool ok = true;
        if (pl == 0) ok = fabs(p[0] - m.table_center[0]) <= m.table_half[0] && fabs(p[1] - m.table_center[1]) <= m.table_half[1] && p[2] > z0 - 0.025;  // end point above the mid-plane of the 0.05 m slab
        if (ok && dist < 0) {
          hit = true;
          v3set(c.n, 0, 0, -1);
          v3set(c.pos, p[0], p[1], z0 + 0.5 * dist);
          c.g1 = i; c.g2 = pl ? GEOM_FLOOR : GEOM_TABLE; c.b1 = m.rcap_body[i]; c.b2 = -1; c.dist = dist;
        }
      }
    }
#if HRG_HULLS
    // hrg_model_desc.robot_hulls: the capsule tests above are the BROADPHASE of the arm links (geoms 0 .. 6).  A pair that passed -- link against a human capsule
    // or against the table / floor plane -- now runs the narrowphase of the link's CONVEX HULL before it is reported (oracle: collide, robot_hulls): distance of
    // the hull to the capsule's axis minus its radius, normal along the witness points, position half way between the two surfaces (an axis that pierces the hull
    // keeps the capsule contact); the hull's lowest point under a plane, one contact per link and plane (the capsule has one per end point).  A hull that does
    // not reach the other geom drops the pair.  One pair at a time in lane (= pair) order, the whole wave on the hull's support mappings; the lane that owns the
    // pair takes the result into its registers.
    if (m.robot_hulls) {
      uint64_t todo = __ballot(hit && c.g1 < HRG_NHULL && c.g2 >= GEOM_HUMAN0);
      int last_plane = -1;
      while (todo) {
        const int src = __ffsll((unsigned long long)todo) - 1;
        todo &= todo - 1;
        const int g1 = __builtin_amdgcn_readlane(c.g1, src), g2 = __builtin_amdgcn_readlane(c.g2, src);
        const int lb = m.rcap_body[g1];
        const HullRef H = {dm_->hull_dev + 3 * m.hull_off[g1], m.hull_off[g1 + 1] - m.hull_off[g1], lb < 0 ? dm_->Rbase : L.kR[lb], lb < 0 ? dm_->m.base_pos : L.kp[lb]};
        if (g2 < GEOM_TABLE) {
          const int hb = g2 - GEOM_HUMAN0;
          double wa[3], wb[3];
          const double dh = gjk_hull_segment_wave(H, &L.hcap[hb][0], &L.hcap[hb][3], wa, wb, (m.hcap_r[hb] + m.contact_margin_human) * (1.0 + 1e-9));   // (a pair it cuts off is dropped below either way)
          if (dh > 1e-9 && lane == src) {
            const double dist = dh - m.hcap_r[hb];
            if (!(dist < m.contact_margin_human)) hit = false;
            else {
              for (int a = 0; a < 3; a++) { c.n[a] = (wb[a] - wa[a]) / dh; c.pos[a] = wa[a] + c.n[a] * (0.5 * dist); }
              c.dist = dist;
            }
          }
        } else {
          const int key = 2 * g1 + (g2 == GEOM_FLOOR ? 1 : 0);
          if (key == last_plane) { if (lane == src) hit = false; continue; }   // the capsule's second end point: the hull has one contact with the plane
          last_plane = key;
          double low[3];
          hull_lowest_wave(H, low);
          const double z0 = g2 == GEOM_FLOOR ? m.floor_z : m.table_top_z, dist = low[2] - z0;
          bool ok = dist < 0;
          if (g2 == GEOM_TABLE) ok = ok && fabs(low[0] - m.table_center[0]) <= m.table_half[0] && fabs(low[1] - m.table_center[1]) <= m.table_half[1] && low[2] > z0 - 0.025;
          if (lane == src) {
            if (!ok) hit = false;
            else { c.pos[0] = low[0]; c.pos[1] = low[1]; c.pos[2] = z0 + 0.5 * dist; c.dist = dist; }
          }
        }
      }
    }
#endif
    const uint64_t mask = __ballot(hit);
    if (hit) {
      const int idx = base + __popcll(mask & lt);
      if (idx < NCON_DYN) L.con[idx] = c;  // full geometry only for the contacts that enter the solve
      if (idx < HRG_NCON_MAX) { L.st.con_pairs[idx][0] = c.g1; L.st.con_pairs[idx][1] = c.g2; }
    }
    base += __popcll(mask);
  }
#if HRG_BOX
  { // the cube, pass 0: lanes 0..9 robot capsule - cube, 16..23 table - cube corners, 24..31 floor - cube corners;
    // pass 1: lanes 0..23 human capsule - cube (human.xml:5: the human's geoms collide with the manipulation object like everything else -- contype / conaffinity
    // 7, margin 0.001; the animated human does not yield).  One copy of the capsule - box narrowphase serves both; the human's contacts close the list, the robot's
    // and the table's come first into the NCON_DYN the solve takes.  Pass 1 runs only when the human does not hold the object and a capsule's bounding sphere reaches the cube's.
    hrg_box_state& bx = L.bx;
    if (lane == 0) { double Rm[9]; quat2mat(Rm, bx.quat); for (int a = 0; a < 9; a++) L.bR[a] = Rm[a]; }
    wave_sync();
    const double hb[3] = {m.box_half[0], m.box_half[1], m.box_half[2]};
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++) {
    bool hit = false;
    Contact c;
    c.g1 = c.g2 = c.b1 = c.b2 = 0; c.dist = 0; v3set(c.n, 0, 0, 1); v3set(c.pos, 0, 0, 0);
    // capsule lanes: the first contact of capsule i on lane i, the second one of a capsule lying along a face on lane SEC0 + i (behind the corner contacts)
    const int ncap = pass ? HRG_NHB : HRG_NRCAP, sec0 = pass ? 32 : 40;
    const bool cap_lane = lane < ncap || (lane >= sec0 && lane < sec0 + ncap);
    const bool second = lane >= sec0;
    const int i = cap_lane ? (second ? lane - sec0 : lane) : 0;
    const double* a1 = pass ? &L.hcap[i][0] : &L.rcapw[i][0];
    const double rad = pass ? m.hcap_r[i] : m.rcap_r[i], margin = pass ? m.contact_margin_human : 0.0;
    bool live = cap_lane && (pass ? true : m.rcap_body[i] >= 0);
    if (pass) {   // broadphase of the human pass: bounding spheres.  No human contacts for an object the human holds (weld / connects active): it lies partly inside
                  // the human's BOUNDING capsules (oracle: collide)
      if (bx.weld_active || HRG_LIFT) break;   // (lifting: the board's pose between the hands stays inside them when let go)
      double d2 = 0;
      for (int a = 0; a < 3; a++) { const double d = 0.5 * (a1[a] + a1[3 + a]) - bx.pos[a]; d2 += d * d; }
      const double reach = dm->hcap_hl[i] + rad + margin + fsqrt(hb[0] * hb[0] + hb[1] * hb[1] + hb[2] * hb[2]) + 1e-9;
      live = live && d2 <= reach * reach;
      if (!__any(live)) break;
    }
    if (cap_lane) {
      if (live) {
        double cs[3], cb[3];
        const double e2 = seg_box(a1, a1 + 3, bx.pos, L.bR, hb, cs, cb);
        double dd = fsqrt(e2);
        double dist = dd - rad;
        if (dist < margin) {
          double s2[3], b2[3];
          const bool two = dist < 0 && dd > 1e-9 && cap_box_two(a1, a1 + 3, bx.pos, L.bR, hb, rad, cs, cb, second ? 1 : 0, s2, b2);
          if (two) { v3cpy(cs, s2); v3cpy(cb, b2); double dv[3]; v3sub(dv, cb, cs); dd = v3norm(dv); dist = dd - rad; }
          hit = two || !second;
          if (dd > 1e-9) { v3sub(c.n, cb, cs); v3scl(c.n, c.n, 1.0 / dd); }
          else { // capsule axis inside the cube: push out through the nearest face
            double loc[3], rel[3], best = 1e300;
            int ax = 0;
            v3sub(rel, cs, bx.pos);
            for (int a = 0; a < 3; a++) { loc[a] = L.bR[a] * rel[0] + L.bR[3 + a] * rel[1] + L.bR[6 + a] * rel[2]; if (hb[a] - fabs(loc[a]) < best) { best = hb[a] - fabs(loc[a]); ax = a; } }
            const double sg = loc[ax] >= 0 ? -1.0 : 1.0;
            for (int a = 0; a < 3; a++) c.n[a] = sg * L.bR[3 * a + ax];
            dist = -best - rad;
          }
          v3madd(c.pos, cs, c.n, rad + 0.5 * dist);
          c.g1 = pass ? GEOM_HUMAN0 + i : i; c.g2 = GEOM_BOX; c.b1 = pass ? -2 : m.rcap_body[i]; c.b2 = BODY_BOX; c.dist = dist;
        }
      }
    }
    else if (pass) {}
#if HRG_LIFT
    // CollaborativeLiftingCart: the board against the table slab (0.4 m wide, 5 cm thick, one metre in front of the robot: collaborative_lifting_cartesian_env.py:
    // 280-284, 742-746) by box-box contacts -- the two overlap in a cross, no corner of either lies over the other.  Lane 16 runs the pair (SAT + clipping, D13)
    // when the board's circumsphere comes near the slab, lanes 16..19 take one contact each; the corner-against-plane test below is left to the floor.
    else if (lane >= 16 && lane < 24) {
      double* T = &L.Jc[20][0];      // scratch behind the collide arrays (the solver rows are dead here): BB_WORK doubles for box_box2, then the contacts
      double* res = &L.Jc[34][0];
      static_assert(20 * (NVS + 1) + BB_WORK <= 32 * (NVS + 1) && 34 * (NVS + 1) + 28 <= (4 * NCON_DYN + 6) * (NVS + 1), "box_box2 workspace");
      int nb = 0;
      if (lane == 16) {
        const double pt[3] = {m.table_center[0], m.table_center[1], m.table_top_z - 0.025}, ht[3] = {m.table_half[0], m.table_half[1], 0.025};
        double d2 = 0;
        for (int a = 0; a < 3; a++) { const double la = fabs(bx.pos[a] - pt[a]) - ht[a]; if (la > 0) d2 += la * la; }
        // ... and the board's extent along z against the slab's (a separating axis of the pair: box_box2 would find no contact): the board is carried 9 cm above the
        // table, so the SAT + clipping below runs only for a board that has come down
        const double ez = fabs(L.bR[6]) * hb[0] + fabs(L.bR[7]) * hb[1] + fabs(L.bR[8]) * hb[2];
        if (!(d2 > hb[0] * hb[0] + hb[1] * hb[1] + hb[2] * hb[2] + 1e-9) && bx.pos[2] - ez < m.table_top_z + 1e-9 && bx.pos[2] + ez > m.table_top_z - 0.05 - 1e-9) {
          // box_box2 indexes centres and extents by axis numbers it finds: they go through LDS (private arrays would sit in scratch memory, as its own arrays did)
          double *wpt = T + 138, *wht = T + 141, *whb = T + 144, *wRt = &L.Jc[32][0];
          for (int a = 0; a < 3; a++) { wpt[a] = pt[a]; wht[a] = ht[a]; whb[a] = hb[a]; }
          for (int a = 0; a < 9; a++) wRt[a] = (a & 3) == 0 ? 1.0 : 0.0;
          BBContact bc[4];
          nb = box_box2(wpt, wRt, wht, bx.pos, L.bR, whb, bc, T);
          for (int q = 0; q < nb; q++) { for (int a = 0; a < 3; a++) { res[7 * q + a] = bc[q].pos[a]; res[7 * q + 3 + a] = bc[q].n[a]; } res[7 * q + 6] = bc[q].dist; }
        }
      }
      __builtin_amdgcn_wave_barrier();   // lane 16's LDS stores stay ahead of the other lanes' loads (a wave's LDS operations execute in issue order)
      nb = __shfl(nb, 16, 64);   // (a shuffle inside a divergent branch reads lane 16, which is active here)
      const int q = lane - 16;
      if (q < nb) {
        hit = true;
        for (int a = 0; a < 3; a++) { c.pos[a] = res[7 * q + a]; c.n[a] = res[7 * q + 3 + a]; }
        c.g1 = GEOM_TABLE; c.g2 = GEOM_BOX; c.b1 = -1; c.b2 = BODY_BOX; c.dist = res[7 * q + 6];
      }
    } else if (lane >= 24 && lane < 32) {
      const int pl = 1, cn = lane & 7;
#else
    else if (lane >= 16 && lane < 32) {
      const int pl = (lane - 16) >> 3, cn = lane & 7;
#endif
      const double loc[3] = {(cn & 1) ? hb[0] : -hb[0], (cn & 2) ? hb[1] : -hb[1], (cn & 4) ? hb[2] : -hb[2]};
      double p[3];
      m3mulv(p, L.bR, loc);
      v3add(p, p, bx.pos);
      const double z0 = pl ? m.floor_z : m.table_top_z, dist = p[2] - z0;
      bool ok = true;
      if (pl == 0) ok = fabs(p[0] - m.table_center[0]) <= m.table_half[0] && fabs(p[1] - m.table_center[1]) <= m.table_half[1] && p[2] > z0 - 0.05;
      if (ok && dist < 0) {
        hit = true;
        v3set(c.n, 0, 0, 1);
        v3set(c.pos, p[0], p[1], z0 + 0.5 * dist);
        c.g1 = pl ? GEOM_FLOOR : GEOM_TABLE; c.g2 = GEOM_BOX; c.b1 = -1; c.b2 = BODY_BOX; c.dist = dist;
      }
    }
#if HRG_HANDOVER
    if (pass == 0) { // RobotHumanHandoverCart._get_object_palm_contact_pos (476-505): does the cube touch the palm (= the collision capsule of the holding hand's body)?
      bool palm = false;
      if (m.task == HRG_TASK_HANDOVER_R2H && lane == 32) {
        const int hl = dm->clips.clip_holding_hand[clip_of(dm, (int64_t)L.st.stream_id, L.st.episode, L.st.anim_index)];
        const int body = m.meas_body[hl ? m.site_lhand : m.site_rhand];
        double cs[3], cb[3];
        palm = fsqrt(seg_box(&L.hcap[body][0], &L.hcap[body][3], bx.pos, L.bR, hb, cs, cb)) - m.hcap_r[body] < 0;
      }
      L.palm_hit = __any(palm);
    }
#endif
    const uint64_t mask = __ballot(hit);
    if (hit) {
      const int idx = base + __popcll(mask & lt);
      if (idx < NCON_DYN) L.con[idx] = c;
      if (idx < HRG_NCON_MAX) { L.st.con_pairs[idx][0] = c.g1; L.st.con_pairs[idx][1] = c.g2; }
    }
    if (pass == 0) {   // _check_grasp: both fingers touch the cube (contacts beyond HRG_NCON_MAX are not reported and do not count)
      const int slot = base + __popcll(mask & lt);
      const bool f0 = __any(hit && lane == HRG_NRCAP - 2 && slot < HRG_NCON_MAX), f1 = __any(hit && lane == HRG_NRCAP - 1 && slot < HRG_NCON_MAX);
      bx.gripped = f0 && f1;
    }
    base += __popcll(mask);
    }
  }
#endif
#if HRG_STACK
  collide_cubes(dm_, lane, &base);
#endif
#if HRG_HAMMER
  collide_hammer(dm_, lane, &base);
#endif
  const int ncon = base < HRG_NCON_MAX ? base : HRG_NCON_MAX;
  if (lane < HRG_NCON_MAX && lane >= ncon) { L.st.con_pairs[lane][0] = -1; L.st.con_pairs[lane][1] = -1; }
  L.st.ncon = ncon;
  *ncon_out = ncon;
  wave_sync();
#if HRG_HAMMER
  { // diagnostic: who touches the nail head (hrg_hammer_state.nail_touch)
    int t = 0;
    if (lane < ncon) {
      const int gn = GEOM_BOX + HRG_HG_NAIL, g1 = L.st.con_pairs[lane][0], g2 = L.st.con_pairs[lane][1], o = g1 == gn ? g2 : (g2 == gn ? g1 : -1);
      if (o >= 0 && o != GEOM_BOX + HRG_HG_BOARD) t = (o == GEOM_BOX + HRG_HG_HANDLE || o == GEOM_BOX + HRG_HG_HEAD) ? 1 : 2;
    }
    const int any = (__any(t & 1) ? 1 : 0) | (__any(t & 2) ? 2 : 0);
    if (any) L.hm.nail_touch = L.hm.nail_touch | any;
  }
#endif
}

// the manipulation object is whitelisted -> COLLISION_TYPE.ALLOWED (pick_place_human_cartesian_env.py:710-717)
// GEOM_BOX + c: cube c of the stacking task; ReachHuman's smallBox is NOT whitelisted (a robot contact with it is a static collision)
// the hammering task white-lists the hammer's two geoms only: "the board is not white-listed" (collaborative_hammering_cartesian_env.py:1325-1337)
DI int geom_class(int g, int task) {
  if (g < HRG_NRCAP) return HRG_GEOM_ROBOT;
  if (g < GEOM_TABLE) return HRG_GEOM_HUMAN;
  if (g < GEOM_BOX || task == HRG_TASK_REACH_BOX) return HRG_GEOM_STATIC;
  if (task == HRG_TASK_HAMMERING) return g == GEOM_BOX + HRG_HG_HANDLE || g == GEOM_BOX + HRG_HG_HEAD ? HRG_GEOM_ALLOWED : HRG_GEOM_STATIC;
  return HRG_GEOM_ALLOWED;
}
DI int cantor(int a, int b) { return (a + b) * (a + b + 1) / 2 + b; }

// HumanEnv._collision_detection, human_env.py:1082-1123 (+ 966-1080).  lanes = contacts of the substep; the reference's loop over the contact list is
// sequential in two places only: the list of remembered pairs is appended in contact order (prefix count of a ballot), and the debounce timer lets the
// FIRST new human contact through and mutes the ones behind it (lowest set bit of a ballot).  Everything else is a count.
PH_CLASSIFY void classify(const DevModel* __restrict__ dm_, int lane, int ncon, int* has_collision, int* collision_type) {
  const ModelPtr dm = uniform_model(dm_);
  Lds& L = g_L;
  const auto& m = dm->m;
  hrg_env_state& s = L.st;
  const double tm = s.debounce_timer - m.timestep;
  double deb = tm > 0 ? tm : 0;
  if (ncon == 0) {   // the usual case
    wave_sync();
    s.debounce_timer = deb;
    s.n_prev = 0;
    return;
  }
  const int n_prev = s.n_prev;
  bool rob = false, fresh = false;
  int h12 = 0, h21 = 0, rg = 0, ot = 0;
  if (lane < ncon) {
    const int g1 = s.con_pairs[lane][0], g2 = s.con_pairs[lane][1];
    const int t1 = geom_class(g1, m.task), t2 = geom_class(g2, m.task);
    rob = t1 == HRG_GEOM_ROBOT || t2 == HRG_GEOM_ROBOT;
    h12 = cantor(g1, g2); h21 = cantor(g2, g1);
    rg = t1 == HRG_GEOM_ROBOT ? g1 : g2;
    ot = t1 == HRG_GEOM_ROBOT ? t2 : t1;
    bool seen = false;
    for (int i = 0; i < n_prev; i++) seen |= s.prev_pairs[i] == h12;
    fresh = rob && !seen;
  }
  const uint64_t rmask = __ballot(rob), lt = (1ull << lane) - 1;
  const int idx = 2 * __popcll(rmask & lt);
  int ncur = 2 * __popcll(rmask);
  if (ncur > HRG_NPREV_MAX) ncur = HRG_NPREV_MAX & ~1;
  const uint64_t m_rr = __ballot(fresh && ot == HRG_GEOM_ROBOT), m_hu = __ballot(fresh && ot == HRG_GEOM_HUMAN);
  const uint64_t m_al = __ballot(fresh && ot == HRG_GEOM_ALLOWED), m_st = __ballot(fresh && ot == HRG_GEOM_STATIC);
  int ct = *collision_type;
  bool crit = false;
  const bool human_counts = m_hu != 0 && !(deb > 0);
  if (human_counts) {
    const int first = __ffsll((long long)m_hu) - 1;
    if (lane == first) {
      double v[3];
      robot_point_vel(m.rcap_body[rg], L.rcen[rg], v);
      crit = !(v3norm(v) <= m.safe_vel);
    }
    crit = __any(crit);
    deb = m.collision_debounce_delay;
  }
  wave_sync();   // every lane has read the remembered pairs and the counters
  if (rob && idx + 2 <= HRG_NPREV_MAX) { s.prev_pairs[idx] = h12; s.prev_pairs[idx + 1] = h21; }
  if ((m_rr | m_hu | m_al | m_st) != 0) *has_collision = 1;
  if (m_rr) { ct |= HRG_COL_ROBOT; s.n_collisions_robot = s.n_collisions_robot + __popcll(m_rr); }
  if (human_counts) {
    if (crit) { ct |= HRG_COL_HUMAN_CRIT; s.n_collisions_critical = s.n_collisions_critical + 1; }
    else { ct |= HRG_COL_HUMAN; s.n_collisions_human = s.n_collisions_human + 1; }
  }
  if (m_al) ct |= HRG_COL_ALLOWED;
  if (m_st) { ct |= HRG_COL_STATIC; s.n_collisions_static = s.n_collisions_static + __popcll(m_st); }
  *collision_type = ct;
  s.debounce_timer = deb;
  s.n_prev = ncur;
}
