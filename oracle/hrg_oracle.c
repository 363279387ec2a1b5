/* hrg_oracle.c — CPU restatement (plain scalar C, double precision) of human-robot-gym's ReachHuman
 * hot path.  TEST INFRASTRUCTURE ONLY: nothing in the product path (human-robot-gym_amd/, bench.py's GPU leg)
 * may link, import or execute this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and there only as the checker.
 *
 * PARITY UNPINNED.  The reference's arithmetic for this path lives in three packages whose sources are
 * absent from the reference checkout and not installed here: MuJoCo 2.1.0 (via mujoco_py==2.1.2.14,
 * requirements.txt:4), sara-shield (git submodule, branch main, unpinned commit, .gitmodules:1-4, directory
 * empty) and robosuite==1.3.2 (requirements.txt:8).  The reference holds no golden vectors, known-answer
 * tests or fixtures for mj_step, SafetyShield.step or env.step.  This file therefore
 * restates (i) the reference's own Python control flow line by line where it exists (cited below as
 * file:line relative to the reference root) and (ii) the published algorithms of the absent engines
 * (Featherstone CRBA/RNEA; MuJoCo's soft-constraint formulation; Thumm & Althoff's fail-safe shield,
 * arXiv 2205.06311; SaRA reachable sets).  What the tests pin is agreement between this restatement and
 * the HIP kernels, plus analytic known-answer tests of the pieces — NOT agreement with MuJoCo/sara-shield.
 * Deviations from the reference stack that are deliberate in this round are listed in DESIGN.md §4.
 */
#define _GNU_SOURCE /* pthread_setaffinity_np (CPU-baseline harness) */
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/hrgym.h"
#include "../include/hrgym_state.h"

#define NV HRG_NV
#define NARM HRG_NARM
#define NVT HRG_NVT    /* robot tree + free joint of the manipulation object */
#define NEFC_MAX 160   /* the stacking task: 8 + 16 + 12 + 4 x 23 rows */
#define NVMAX HRG_NV_STACK /* largest constrained system: robot tree + the four cubes of CollaborativeStackingCart */
#define NCUBE HRG_NCUBE
#define BODY_BOX 100   /* body code of the manipulation object in contact_t.b1/b2 */
#define PI 3.14159265358979323846
#define SIXTH (1.0 / 6.0) /* cubic term of the constant-jerk profiles */

#include <stdio.h>
static int g_debug = 0;
void hrgo_set_debug(int d) { g_debug = d; }
/* =============================================================================================== vec/quat */
static void v3set(double* r, double a, double b, double c) { r[0] = a; r[1] = b; r[2] = c; }
static void v3cpy(double* r, const double* a) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
static void v3add(double* r, const double* a, const double* b) { r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; }
static void v3sub(double* r, const double* a, const double* b) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
static void v3scl(double* r, const double* a, double s) { r[0] = a[0] * s; r[1] = a[1] * s; r[2] = a[2] * s; }
static void v3madd(double* r, const double* a, const double* b, double s) { r[0] = a[0] + b[0] * s; r[1] = a[1] + b[1] * s; r[2] = a[2] + b[2] * s; }
static double v3dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double v3norm(const double* a) { return sqrt(v3dot(a, a)); }
static void v3cross(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
/* row-major 3x3 */
static void m3mulv(double* r, const double* M, const double* v) {
  double x = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
  double y = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
  double z = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void m3mul(double* R, const double* A, const double* B) {
  double T[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(R, T, sizeof T);
}
/* quaternion (w,x,y,z) -> rotation matrix */
static void quat2mat(double* M, const double* q) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  M[0] = 1 - 2 * (y * y + z * z); M[1] = 2 * (x * y - w * z); M[2] = 2 * (x * z + w * y);
  M[3] = 2 * (x * y + w * z); M[4] = 1 - 2 * (x * x + z * z); M[5] = 2 * (y * z - w * x);
  M[6] = 2 * (x * z - w * y); M[7] = 2 * (y * z + w * x); M[8] = 1 - 2 * (x * x + y * y);
}
static void quatmul(double* r, const double* a, const double* b) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
/* rotation about a unit axis by angle, as a matrix */
static void axisangle2mat(double* M, const double* ax, double ang) {
  double c = cos(ang), s = sin(ang), t = 1 - c, x = ax[0], y = ax[1], z = ax[2];
  M[0] = t * x * x + c; M[1] = t * x * y - s * z; M[2] = t * x * z + s * y;
  M[3] = t * x * y + s * z; M[4] = t * y * y + c; M[5] = t * y * z - s * x;
  M[6] = t * x * z - s * y; M[7] = t * y * z + s * x; M[8] = t * z * z + c;
}
static double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* =============================================================================================== RNG
 * The reference draws from the process-global legacy numpy stream seeded with seed+rank
 * (human_env.py:336,1637; reach_human_env.py:536).  A batched env cannot share one sequential stream,
 * so draws are counter-based: u = hash(seed, global env id, episode, stream, index).  Same distributions,
 * different numbers; independent of how envs are sharded over GPUs. */
static uint64_t mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ULL;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}
static double rng_u01(uint64_t seed, uint64_t env, uint64_t episode, uint64_t stream, uint64_t idx) {
  uint64_t h = mix64(seed);
  h = mix64(h ^ (env * 0xD1B54A32D192ED03ULL));
  h = mix64(h ^ (episode * 0x8CB92BA72F3D8DD7ULL));
  h = mix64(h ^ (stream * 0xABC98388FB8FAC03ULL + idx));
  return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}
enum { STREAM_NOISE = 0, STREAM_HUMAN = 1, STREAM_ANIM = 2, STREAM_GOAL = 3, STREAM_ACTION = 4, STREAM_OBJECT = 5, STREAM_TARGET = 6, STREAM_LOOP = 7 };
static double rng_gauss(uint64_t seed, uint64_t env, uint64_t ep, uint64_t stream, uint64_t idx) {
  double u1 = rng_u01(seed, env, ep, stream, 2 * idx), u2 = rng_u01(seed, env, ep, stream, 2 * idx + 1);
  return sqrt(-2.0 * log(1.0 - u1)) * cos(2.0 * PI * u2);
}

/* =============================================================================================== batch */
typedef struct hrgo_batch {
  hrg_model_desc m;
  hrg_clip_table clips;
  double* frames; /* owned copy */
  double* hull_verts; /* owned copy of the arm links' hull vertices (robot_hulls) */
  int32_t n_envs;
  int64_t env_id0;
  hrg_env_state* st;
  hrg_box_state* box; /* manipulation object per env (unused by ReachHuman) */
  hrg_stack_state* stk; /* the four cubes + bookkeeping of CollaborativeStackingCart */
  hrg_hammer_state* hmr; /* board, hammer, nail + bookkeeping of CollaborativeHammeringCart */
  /* parity taps of the last shield cycle */
  double (*rcaps)[HRG_NSHIELD_RCAP][7];
  double (*hcaps)[HRG_NHCAP_MAX][7];
  int32_t* n_hcaps;
} hrgo_batch;

/* =============================================================================================== robot
 * Kinematics + CRBA + RNEA of the 8-DoF robot tree in world coordinates about the world origin
 * (Featherstone, RBDA ch. 5-6; the quantities mj_kinematics/mj_crb/mj_rne produce for this tree:
 * SURVEY.md Appendix B.1). */
typedef struct {
  double R[NV][9], p[NV][3]; /* body frames */
  double Sw[NV][3], Sv[NV][3]; /* joint motion subspace: angular, linear (about world origin) */
  double com[NV][3];           /* world com */
  double Iw[NV][6];            /* world rotational inertia about com: xx yy zz xy xz yz */
  double vw[NV][3], vv[NV][3]; /* body spatial velocity (filled by robot_bias) */
} robot_kin;

static void robot_fk(const hrg_model_desc* m, const double* q, robot_kin* k) {
  double Rb[9];
  quat2mat(Rb, m->base_quat);
  for (int i = 0; i < NV; i++) {
    const double *Rp, *pp;
    int par = m->body_parent[i];
    if (par < 0) { Rp = Rb; pp = m->base_pos; } else { Rp = k->R[par]; pp = k->p[par]; }
    double Rq[9], Rl[9], t[3], axw[3];
    quat2mat(Rq, m->body_quat[i]);
    m3mul(Rl, Rp, Rq);
    m3mulv(t, Rp, m->body_pos[i]);
    v3add(k->p[i], pp, t);
    if (m->jnt_type[i] == 0) {
      double Rj[9];
      axisangle2mat(Rj, m->jnt_axis[i], q[i]);
      m3mul(k->R[i], Rl, Rj);
      m3mulv(axw, k->R[i], m->jnt_axis[i]);
      v3cpy(k->Sw[i], axw);
      v3cross(k->Sv[i], k->p[i], axw);
    } else {
      memcpy(k->R[i], Rl, sizeof Rl);
      m3mulv(axw, k->R[i], m->jnt_axis[i]);
      v3madd(k->p[i], k->p[i], axw, q[i]);
      v3set(k->Sw[i], 0, 0, 0);
      v3cpy(k->Sv[i], axw);
    }
    m3mulv(t, k->R[i], m->body_com[i]);
    v3add(k->com[i], k->p[i], t);
    /* Iw = R I R^T */
    const double* I = m->body_inertia[i];
    double Ib[9] = {I[0], I[3], I[4], I[3], I[1], I[5], I[4], I[5], I[2]}, T[9], Rt[9], W[9];
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Rt[3 * a + b] = k->R[i][3 * b + a];
    m3mul(T, k->R[i], Ib);
    m3mul(W, T, Rt);
    k->Iw[i][0] = W[0]; k->Iw[i][1] = W[4]; k->Iw[i][2] = W[8]; k->Iw[i][3] = W[1]; k->Iw[i][4] = W[2]; k->Iw[i][5] = W[5];
  }
}

/* spatial inertia about the world origin in additive form: m, h = m*c, IO(6) */
typedef struct { double m, h[3], I[6]; } sinertia;
static void sinertia_body(sinertia* s, double m, const double* c, const double* Ic) {
  s->m = m;
  v3scl(s->h, c, m);
  double cc = v3dot(c, c);
  s->I[0] = Ic[0] + m * (cc - c[0] * c[0]);
  s->I[1] = Ic[1] + m * (cc - c[1] * c[1]);
  s->I[2] = Ic[2] + m * (cc - c[2] * c[2]);
  s->I[3] = Ic[3] - m * c[0] * c[1];
  s->I[4] = Ic[4] - m * c[0] * c[2];
  s->I[5] = Ic[5] - m * c[1] * c[2];
}
/* force = I * motion : n = IO w + h x v ; f = m v - h x w */
static void sinertia_mul(double* n, double* f, const sinertia* s, const double* w, const double* v) {
  double hv[3], hw[3];
  v3cross(hv, s->h, v);
  v3cross(hw, s->h, w);
  n[0] = s->I[0] * w[0] + s->I[3] * w[1] + s->I[4] * w[2] + hv[0];
  n[1] = s->I[3] * w[0] + s->I[1] * w[1] + s->I[5] * w[2] + hv[1];
  n[2] = s->I[4] * w[0] + s->I[5] * w[1] + s->I[2] * w[2] + hv[2];
  f[0] = s->m * v[0] - hw[0];
  f[1] = s->m * v[1] - hw[1];
  f[2] = s->m * v[2] - hw[2];
}

/* composite-rigid-body mass matrix (dense NV x NV, row major), armature on the diagonal */
static void robot_crba(const hrg_model_desc* m, const robot_kin* k, double* M) {
  sinertia c[NV];
  for (int i = 0; i < NV; i++) sinertia_body(&c[i], m->body_mass[i], k->com[i], k->Iw[i]);
  for (int i = NV - 1; i >= 0; i--) {
    int par = m->body_parent[i];
    if (par >= 0) {
      c[par].m += c[i].m;
      for (int a = 0; a < 3; a++) c[par].h[a] += c[i].h[a];
      for (int a = 0; a < 6; a++) c[par].I[a] += c[i].I[a];
    }
  }
  memset(M, 0, sizeof(double) * NV * NV);
  for (int j = 0; j < NV; j++) {
    double n[3], f[3];
    sinertia_mul(n, f, &c[j], k->Sw[j], k->Sv[j]);
    for (int i = j; i >= 0; i = m->body_parent[i]) {
      double v = v3dot(k->Sw[i], n) + v3dot(k->Sv[i], f);
      M[i * NV + j] = v;
      M[j * NV + i] = v;
    }
    M[j * NV + j] += m->jnt_armature[j];
  }
}

/* bias forces c(q, qd) = RNEA(q, qd, 0) incl. gravity; also leaves body velocities in k */
static void robot_bias(const hrg_model_desc* m, robot_kin* k, const double* qd, double* bias) {
  double aw[NV][3], av[NV][3], fn[NV][3], ff[NV][3];
  for (int i = 0; i < NV; i++) {
    int par = m->body_parent[i];
    double pw[3] = {0, 0, 0}, pv[3] = {0, 0, 0}, paw[3] = {0, 0, 0}, pav[3];
    v3scl(pav, m->gravity, -1.0);
    if (par >= 0) { v3cpy(pw, k->vw[par]); v3cpy(pv, k->vv[par]); v3cpy(paw, aw[par]); v3cpy(pav, av[par]); }
    double jw[3], jv[3];
    v3scl(jw, k->Sw[i], qd[i]);
    v3scl(jv, k->Sv[i], qd[i]);
    v3add(k->vw[i], pw, jw);
    v3add(k->vv[i], pv, jv);
    /* a = a_parent + crm(v_i) (S qd) : [w x jw ; w x jv + v x jw] */
    double t1[3], t2[3], t3[3];
    v3cross(t1, k->vw[i], jw);
    v3cross(t2, k->vw[i], jv);
    v3cross(t3, k->vv[i], jw);
    v3add(aw[i], paw, t1);
    v3add(av[i], pav, t2);
    v3add(av[i], av[i], t3);
    /* f = I a + crf(v)(I v) : [w x n + v x f ; w x f] */
    sinertia s;
    sinertia_body(&s, m->body_mass[i], k->com[i], k->Iw[i]);
    double n1[3], f1[3], n2[3], f2[3];
    sinertia_mul(n1, f1, &s, aw[i], av[i]);
    sinertia_mul(n2, f2, &s, k->vw[i], k->vv[i]);
    v3cross(t1, k->vw[i], n2);
    v3cross(t2, k->vv[i], f2);
    v3cross(t3, k->vw[i], f2);
    for (int a = 0; a < 3; a++) { fn[i][a] = n1[a] + t1[a] + t2[a]; ff[i][a] = f1[a] + t3[a]; }
  }
  for (int i = NV - 1; i >= 0; i--) {
    bias[i] = v3dot(k->Sw[i], fn[i]) + v3dot(k->Sv[i], ff[i]);
    int par = m->body_parent[i];
    if (par >= 0) for (int a = 0; a < 3; a++) { fn[par][a] += fn[i][a]; ff[par][a] += ff[i][a]; }
  }
}

/* velocity of a world point rigidly attached to body b (b<0: static) */
static void robot_point_vel(const robot_kin* k, int b, const double* r, double* v) {
  if (b < 0) { v3set(v, 0, 0, 0); return; }
  double t[3];
  v3cross(t, k->vw[b], r);
  v3add(v, k->vv[b], t);
}
/* Jacobian row: d/dqd of (dir . velocity of point r on body b) */
static void robot_point_jac(const hrg_model_desc* m, const robot_kin* k, int b, const double* r, const double* dir, double sign, double* J) {
  for (int i = b; i >= 0; i = m->body_parent[i]) {
    double t[3], v[3];
    v3cross(t, k->Sw[i], r);
    v3add(v, k->Sv[i], t);
    J[i] += sign * v3dot(dir, v);
  }
}

/* Cholesky of an n x n SPD matrix in place (lower), returns 0 if not PD */
static int chol(double* A, int n) {
  for (int j = 0; j < n; j++) {
    double d = A[j * n + j];
    for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
    if (!(d > 0)) return 0;
    d = sqrt(d);
    A[j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = A[i * n + j];
      for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = s / d;
    }
  }
  return 1;
}
static void chol_solve(const double* L, int n, double* x) {
  for (int i = 0; i < n; i++) {
    double s = x[i];
    for (int k = 0; k < i; k++) s -= L[i * n + k] * x[k];
    x[i] = s / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = x[i];
    for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k];
    x[i] = s / L[i * n + i];
  }
}

/* =============================================================================================== capsules */
/* closest points of two segments (Ericson, Real-Time Collision Detection 5.1.9; the clamped
 * closest-point computation SURVEY.md B.3 names).  Returns squared distance, c1/c2 closest points. */
static double seg_seg(const double* p1, const double* q1, const double* p2, const double* q2, double* c1, double* c2) {
  double d1[3], d2[3], r[3];
  v3sub(d1, q1, p1);
  v3sub(d2, q2, p2);
  v3sub(r, p1, p2);
  double a = v3dot(d1, d1), e = v3dot(d2, d2), f = v3dot(d2, r), s, t;
  const double EPS = 1e-12;
  if (a <= EPS && e <= EPS) { s = t = 0; }
  else if (a <= EPS) { s = 0; t = clampd(f / e, 0, 1); }
  else {
    double c = v3dot(d1, r);
    if (e <= EPS) { t = 0; s = clampd(-c / a, 0, 1); }
    else {
      double b = v3dot(d1, d2), den = a * e - b * b;
      s = den > EPS * a * e ? clampd((b * f - c * e) / den, 0, 1) : 0;
      t = (b * s + f) / e;
      if (t < 0) { t = 0; s = clampd(-c / a, 0, 1); }
      else if (t > 1) { t = 1; s = clampd((b - c) / a, 0, 1); }
    }
  }
  v3madd(c1, p1, d1, s);
  v3madd(c2, p2, d2, t);
  double d[3];
  v3sub(d, c1, c2);
  return v3dot(d, d);
}

/* =============================================================================================== human
 * Animation clock and root/joint pose: HumanEnv._control_human, human_env.py:1710-1767.
 * Kinematics of the 24-body tree: the mj_kinematics rule for bodies whose three hinges share an anchor
 * (R_b = R_parent Rz Ry Rx ; the anchor stays fixed), human.xml:50-59. */
typedef struct {
  double R[HRG_NHB][9], p[HRG_NHB][3];
  double cap1[HRG_NHB][3], cap2[HRG_NHB][3]; /* world collision capsules */
} human_kin;

static int clip_of(const hrgo_batch* b, int64_t gid, const hrg_env_state* s, int anim_index) {
  double u = rng_u01(b->m.seed, (uint64_t)gid, (uint64_t)s->episode, STREAM_ANIM, (uint64_t)anim_index);
  int c = (int)(u * b->m.n_clips);
  return c >= b->m.n_clips ? b->m.n_clips - 1 : c;
}

/* amplitude (speed = 0) or speed modifier (1) of layered sine k of the loop of animation slot ai in this episode:
 * sample_animation_loop_properties (utils/animation_utils.py:122-176): info value x exp(clip(N(0,1), -3, 3) log(std factor)); drawn
 * counter-based on demand instead of as lists filled at reset */
static double loop_prop(const hrgo_batch* b, int64_t gid, const hrg_env_state* s, int ai, int clip, int k, int speed) {
  const hrg_clip_table* c = &b->clips;
  double base = speed ? c->clip_loop_speed[clip][k % HRG_MAX_LOOP] : c->clip_loop_amp[clip][k % HRG_MAX_LOOP];
  if (k >= HRG_MAX_LOOP) base = speed ? c->clip_loop2_speed[clip][k - HRG_MAX_LOOP] : c->clip_loop2_amp[clip][k - HRG_MAX_LOOP]; /* second loop stage ("wait") */
  double sf = speed ? c->clip_loop_speed_std[clip] : c->clip_loop_amp_std[clip];
  double z = clampd(rng_gauss(b->m.seed, (uint64_t)gid, (uint64_t)s->episode, STREAM_LOOP, (uint64_t)((ai * 2 * HRG_MAX_LOOP + k) * 2 + speed)), -3.0, 3.0);
  return base * exp(z * log(sf));
}

/* layered_sin_modulations (utils/animation_utils.py:91-119) over loop sines k0..k0+n-1 of the clip */
static double layered_sines(const hrgo_batch* b, int64_t gid, const hrg_env_state* s, int clip, int kfirst, int n, double t, double start) {
  double sum = 0;
  for (int k = 0; k < n; k++) {
    const double A = loop_prop(b, gid, s, s->anim_index, clip, kfirst + k, 0), S = loop_prop(b, gid, s, s->anim_index, clip, kfirst + k, 1);
    sum += A * sin((t - start) / (A / S)) + start;
  }
  return sum - start * (double)(n - 1);
}

static void stack_animation_time(const hrgo_batch* b, int64_t gid, const hrg_env_state* s, hrg_stack_state* sk, int clip, int* at_io);
static void hammer_animation_time(const hrgo_batch* b, int64_t gid, const hrg_env_state* s, hrg_hammer_state* hm, int clip, int* at_io);
static void human_control_all(const hrgo_batch* b, int64_t gid, hrg_env_state* s, hrg_box_state* bx, hrg_stack_state* sk, hrg_hammer_state* hm, double* mocap_pos, double* mocap_quat, const double** qh) {
  const hrg_model_desc* m = &b->m;
  /* human_env.py:1719-1731 */
  int control_time = (int)floor((double)s->low_level_time / m->anim_step_length);
  int at = control_time - s->anim_start_time;
  int clip = clip_of(b, gid, s, s->anim_index);
  if (m->task == HRG_TASK_INSPECTION) { /* HumanObjectInspectionCart._compute_animation_time, human_object_inspection_cartesian_env.py:602-652 */
    const int classic = at, k0 = b->clips.clip_keyframes[clip][0], k1 = b->clips.clip_keyframes[clip][1], len = b->clips.clip_len[clip];
    if (at > k0 && bx->task_phase == HRG_PHASE_APPROACH) bx->task_phase = HRG_PHASE_READY;
    if (bx->task_phase == HRG_PHASE_READY) { /* idle loop around the first keyframe: layered_sin_modulations, utils/animation_utils.py:62-119 */
      const int nl = b->clips.clip_n_loop[clip];
      double sum = 0;
      for (int k = 0; k < nl; k++) {
        const double A = loop_prop(b, gid, s, s->anim_index, clip, k, 0), S = loop_prop(b, gid, s, s->anim_index, clip, k, 1);
        sum += A * sin((double)(classic - k0) / (A / S)) + (double)k0;
      }
      at = (int)(sum - (double)k0 * (double)(nl - 1));
      bx->n_delayed = classic - at;
    } else at -= bx->n_delayed;
    if (at > k1) bx->task_phase = HRG_PHASE_RETREAT;
    if (at >= len - 1) { bx->task_phase = HRG_PHASE_COMPLETE; at = len - 1; }
    if (at < 0) at = 0; /* a loop amplitude larger than the first keyframe must not index before the clip */
    for (int a = 0; a < 3; a++) bx->target[a] = b->clips.clip_target_pos[clip][a] + s->human_pos_offset[a]; /* target_pos property, 447-459 */
  }
  if (m->task == HRG_TASK_HANDOVER_H2R) { /* HumanRobotHandoverCart._compute_animation_time, human_robot_handover_cartesian_env.py:530-596 */
    const int classic = at, k0 = b->clips.clip_keyframes[clip][0], k1 = b->clips.clip_keyframes[clip][1], len = b->clips.clip_len[clip];
    if (at > k0 && bx->task_phase == HRG_PHASE_APPROACH) bx->task_phase = HRG_PHASE_PRESENT;
    else if ((double)at > (double)k0 + (double)(k1 - k0) / 2.0 && bx->task_phase == HRG_PHASE_PRESENT) { /* present the object: loop around the middle of the two keyframes */
      at = (int)layered_sines(b, gid, s, clip, 0, b->clips.clip_n_loop[clip], (double)classic, (double)(k0 + k1) / 2.0);
      bx->n_delayed = classic - at; bx->n_delayed2 = 0;
    } else if (bx->task_phase == HRG_PHASE_WAIT) { /* wait at the second keyframe until the object is placed */
      at = classic - bx->n_delayed;
      if (at >= k1) at = (int)layered_sines(b, gid, s, clip, HRG_MAX_LOOP, b->clips.clip_n_loop2[clip], (double)at, (double)k1);
      bx->n_delayed2 = classic - at;
    } else if (bx->task_phase == HRG_PHASE_RETREAT) at -= bx->n_delayed2;
    if (at >= len - 1) { bx->task_phase = HRG_PHASE_COMPLETE; at = len - 1; }
    if (at < 0) at = 0;
  }
  if (m->task == HRG_TASK_HANDOVER_R2H) { /* RobotHumanHandoverCart._compute_animation_time, robot_human_handover_cartesian_env.py:556-606 */
    const int classic = at, k0 = b->clips.clip_keyframes[clip][0], k1 = b->clips.clip_keyframes[clip][1], len = b->clips.clip_len[clip];
    if (at > k0 && bx->task_phase == HRG_R2H_APPROACH) bx->task_phase = HRG_R2H_REACH_OUT;
    if ((double)at > (double)k0 + (double)(k1 - k0) / 2.0 && bx->task_phase == HRG_R2H_REACH_OUT) { /* hold the hand out: loop around the middle of the keyframes */
      at = (int)layered_sines(b, gid, s, clip, 0, b->clips.clip_n_loop[clip], (double)classic, (double)(k0 + k1) / 2.0);
      bx->n_delayed = classic - at;
    }
    if (bx->task_phase == HRG_R2H_RETREAT) at -= bx->n_delayed;
    if (at >= len - 1) { bx->task_phase = HRG_R2H_COMPLETE; at = len - 1; }
    if (at < 0) at = 0;
  }
  if (m->task == HRG_TASK_LIFTING) { /* CollaborativeLiftingCart._compute_animation_time (collaborative_lifting_cartesian_env.py:563-581): frozen at the last frame */
    const int len = b->clips.clip_len[clip];
    if (at >= len - 1) { bx->task_phase = HRG_PHASE_COMPLETE; at = len - 1; }
    if (at < 0) at = 0;
  }
  if (m->task == HRG_TASK_STACKING) stack_animation_time(b, gid, s, sk, clip, &at);
  if (m->task == HRG_TASK_HAMMERING) hammer_animation_time(b, gid, s, hm, clip, &at);
  s->animation_time = at;
  if (at > b->clips.clip_len[clip] - 1) {
    s->anim_index = (s->anim_index + 1) % m->n_anim_ids; /* human_env.py:1704-1708 */
    s->animation_time = 0;
    s->anim_start_time = control_time;
    clip = clip_of(b, gid, s, s->anim_index);
  }
  const double* fr = b->frames + (b->clips.clip_offset[clip] + s->animation_time) * HRG_FRAME_DIM;
  /* human_env.py:1736-1763: pos = R(base*info) (p_anim + off_info) + offset_env ; rot = rot_env*base*info*pelvis */
  double qi[4] = {b->clips.clip_quat[clip][3], b->clips.clip_quat[clip][0], b->clips.clip_quat[clip][1], b->clips.clip_quat[clip][2]};
  double qbi[4], Rbi[9], pa[3], pr[3];
  quatmul(qbi, m->human_base_quat, qi);
  quat2mat(Rbi, qbi);
  for (int a = 0; a < 3; a++) pa[a] = fr[a] + b->clips.clip_pos_offset[clip][a];
  m3mulv(pr, Rbi, pa);
  v3add(mocap_pos, pr, s->human_pos_offset);
  double qa[4] = {fr[6], fr[3], fr[4], fr[5]}, q1[4];
  quatmul(q1, s->human_rot_offset, qbi);
  quatmul(mocap_quat, q1, qa);
  *qh = fr + 7;
}

static void human_control_sk(const hrgo_batch* b, int64_t gid, hrg_env_state* s, hrg_box_state* bx, hrg_stack_state* sk, double* mocap_pos, double* mocap_quat, const double** qh) {
  human_control_all(b, gid, s, bx, sk, NULL, mocap_pos, mocap_quat, qh);
}
static void human_control(const hrgo_batch* b, int64_t gid, hrg_env_state* s, hrg_box_state* bx, double* mocap_pos, double* mocap_quat, const double** qh) {
  human_control_all(b, gid, s, bx, NULL, NULL, mocap_pos, mocap_quat, qh);
}

static void human_fk(const hrg_model_desc* m, const double* mocap_pos, const double* mocap_quat, const double* qh, human_kin* h, double site[HRG_NHJ][3]) {
  static const double ez[3] = {0, 0, 1}, ey[3] = {0, 1, 0}, ex[3] = {1, 0, 0};
  quat2mat(h->R[0], mocap_quat);
  v3cpy(h->p[0], mocap_pos);
  for (int b = 1; b < HRG_NHB; b++) {
    int par = m->hb_parent[b];
    double Rz[9], Ry[9], Rx[9], T[9], anc[3], t[3];
    const double* q = qh + 3 * (b - 1); /* z, y, x (human.xml joint order) */
    axisangle2mat(Rz, ez, q[0]);
    axisangle2mat(Ry, ey, q[1]);
    axisangle2mat(Rx, ex, q[2]);
    m3mul(T, h->R[par], Rz);
    m3mul(T, T, Ry);
    m3mul(h->R[b], T, Rx);
    m3mulv(t, h->R[par], m->hb_anchor[b]);
    v3add(anc, h->p[par], t);
    m3mulv(t, h->R[b], m->hb_anchor[b]);
    v3sub(h->p[b], anc, t);
  }
  for (int b = 0; b < HRG_NHB; b++) {
    double t[3];
    m3mulv(t, h->R[b], m->hcap_p1[b]); v3add(h->cap1[b], h->p[b], t);
    m3mulv(t, h->R[b], m->hcap_p2[b]); v3add(h->cap2[b], h->p[b], t);
  }
  for (int j = 0; j < HRG_NHJ; j++) {
    int b = m->meas_body[j];
    double t[3];
    m3mulv(t, h->R[b], m->hb_anchor[b]);
    v3add(site[j], h->p[b], t);
  }
}

/* =============================================================================================== LTT
 * Long-term trajectory: per joint jerk-limited point-to-point profile from (q0,v0,a0) to (goal,0,0)
 * (role of sara-shield LongTermPlanner behind SafetyShield.newLongTermTrajectory,
 * failsafe_controller.py:300).  Own construction (the planner source is absent): ramp a->0, then
 * S-curve / cruise / S-curve with the cruise velocity found by a root search; the joints are then time-synchronised to the slowest one (ltt_sync_joint). */
static int scurve(double va, double vb, double amax, double jmax, double* dur, double* jerk) {
  double d = vb - va, ad = fabs(d), sg = d >= 0 ? 1.0 : -1.0;
  if (ad >= amax * amax / jmax) {
    double tj = amax / jmax;
    dur[0] = tj; dur[1] = ad / amax - tj; dur[2] = tj;
  } else {
    double tj = sqrt(ad / jmax);
    dur[0] = tj; dur[1] = 0; dur[2] = tj;
  }
  jerk[0] = sg * jmax; jerk[1] = 0; jerk[2] = -sg * jmax;
  return 3;
}
static double scurve_time(double dv, double amax, double jmax) {
  double ad = fabs(dv);
  return ad >= amax * amax / jmax ? ad / amax + amax / jmax : 2 * sqrt(ad / jmax);
}
/* distance of [S-curve va->vc][S-curve vc->0] without cruise */
static double dist_nocruise(double va, double vc, double amax, double jmax) {
  return 0.5 * (va + vc) * scurve_time(vc - va, amax, jmax) + 0.5 * vc * scurve_time(vc, amax, jmax);
}

/* what the last three parts of a joint's profile ([S-curve v -> sg w][cruise][S-curve sg w -> 0], segments n0 ..) were planned from: the time synchronisation re-plans them */
typedef struct { int n0; double v, sg, Dm, vm, w; } ltt_tail;
static void ltt_plan_joint(hrg_ltt* L, int j, double q0, double v0, double a0, double goal, double vmax, double amax, double jmax, ltt_tail* tail) {
  double* dur = L->dur[j];
  double* jerk = L->jerk[j];
  for (int i = 0; i < HRG_LTT_NSEG; i++) { dur[i] = 0; jerk[i] = 0; }
  L->q0[j] = q0; L->v0[j] = v0; L->a0[j] = a0; L->qT[j] = goal;
  int n = 0;
  double q = q0, v = v0;
  /* seg 0: acceleration to zero */
  if (fabs(a0) > 1e-9) { /* below that the ramp is a no-op (and the sign of rounding noise must not matter) */
    double t = fabs(a0) / jmax, jj = a0 > 0 ? -jmax : jmax;
    dur[n] = t; jerk[n] = jj; n++;
    q += v0 * t + 0.5 * a0 * t * t + jj * t * t * t * SIXTH;
    v += a0 * t + 0.5 * jj * t * t;
  } else n++;
  double D = goal - q;
  double dstop = 0.5 * v * scurve_time(v, amax, jmax);
  double sg = (D - dstop) >= 0 ? 1.0 : -1.0;
  if (sg * v < -1e-9) {
    /* moving away from where we must go: stop first, continue from rest */
    n += scurve(v, 0, amax, jmax, dur + n, jerk + n);
    D -= dstop;
    v = 0;
  } else n += 3;
  /* now sg*v >= 0 and sg*(D - dstop(v)) >= 0: find cruise velocity vc = sg*w, w in [|v|, wmax] */
  double w_lo = fabs(v), w_hi = vmax > w_lo ? vmax : w_lo, w, tc = 0;
  double Dm = sg * D, vm = sg * v; /* mirrored problem: everything positive */
  if (Dm >= dist_nocruise(vm, w_hi, amax, jmax)) {
    w = w_hi;
    tc = w > 0 ? (Dm - dist_nocruise(vm, w, amax, jmax)) / w : 0;
  } else {
    /* root of the increasing, convex, C1 function f(w) = dist_nocruise(vm, w) - Dm on [w_lo, w_hi]: Newton with the analytic derivative, safeguarded by
     * the bracket (bisection when a step leaves it).  Start: the cruise speed of an all-trapezoidal profile from rest over Dm plus the ramp 0 -> vm (the root
     * itself when vm = 0 and both ramps reach amax).  An iterate that hits the root (f == 0, or a step that no longer moves) ends the search: it must not
     * be mistaken for a step onto the bracket's edge, which would send the search bisecting around the root it already has. */
    double lo = w_lo, hi = w_hi;
    const double vtri = amax * amax / jmax;
    const double De = Dm + 0.5 * vm * scurve_time(vm, amax, jmax);
    w = 0.5 * (sqrt(vtri * vtri + 4.0 * amax * De) - vtri);
    if (!(w > lo)) w = lo;
    if (!(w < hi)) w = hi;
    for (int it = 0; it < 80; it++) {
      const double d1 = w - vm, T1 = scurve_time(d1, amax, jmax), T2 = scurve_time(w, amax, jmax);
      const double f = 0.5 * (vm + w) * T1 + 0.5 * w * T2 - Dm;
      if (f == 0) break;
      if (f < 0) lo = w; else hi = w;
      const double T1p = fabs(d1) >= vtri ? 1.0 / amax : (fabs(d1) > 0 ? 1.0 / sqrt(jmax * fabs(d1)) : 0.0);
      const double T2p = fabs(w) >= vtri ? 1.0 / amax : (fabs(w) > 0 ? 1.0 / sqrt(jmax * fabs(w)) : 0.0);
      const double fp = 0.5 * T1 + 0.5 * (vm + w) * T1p + 0.5 * T2 + 0.5 * w * T2p;
      double nw = fp > 0 ? w - f / fp : 0.5 * (lo + hi);
      if (!(nw >= lo && nw <= hi)) nw = 0.5 * (lo + hi);
      const double step = fabs(nw - w);
      w = nw;
      if (step <= 4e-16 * (1.0 + fabs(w)) || hi - lo <= 4e-16 * (1.0 + hi)) break;
    }
  }
  if (tail) { tail->n0 = n; tail->v = v; tail->sg = sg; tail->Dm = Dm; tail->vm = vm; tail->w = w; }
  n += scurve(v, sg * w, amax, jmax, dur + n, jerk + n);
  dur[n] = tc; jerk[n] = 0; n++;
  n += scurve(sg * w, 0, amax, jmax, dur + n, jerk + n);
}

/* Time synchronisation of a long-term trajectory (sara-shield's LongTermPlanner [UPSTREAM]: the joints are "time-synchronised to the slowest joint", SURVEY.md B.3;
 * arXiv 2205.06311 section IV): a joint whose own time-optimal profile ends before T keeps its first parts (the ramp of its acceleration to zero, a stop if it moves
 * away from its goal) and drives the rest at a LOWER cruise speed w, chosen so that it arrives exactly at T.  The duration of the rest,
 *   g(w) = T1(w - vm) + T2(w) + (Dm - dist_nocruise(vm, w)) / w,
 * falls monotonically from infinity (w -> 0) to the time-optimal duration (w = the speed planned): Newton on g(w) = Tt with the analytic derivative, safeguarded by the
 * bracket.  A joint that has to use its whole remaining distance to stop (Dm = its stopping distance) cannot be slowed down and keeps its profile. */
static void ltt_sync_joint(hrg_ltt* L, int j, double T, double amax, double jmax, const ltt_tail* tl) {
  double* dur = L->dur[j];
  double* jerk = L->jerk[j];
  double tpre = 0, tall = 0;
  for (int i = 0; i < HRG_LTT_NSEG; i++) { if (i < tl->n0) tpre += dur[i]; tall += dur[i]; }
  if (!(tall < T - 1e-12) || !(tl->w > 0)) return;
  const double Tt = T - tpre, vm = tl->vm, Dm = tl->Dm, vtri = amax * amax / jmax;
  double lo = 0, hi = tl->w;                 /* g(lo) > Tt >= g(hi) */
  double w = hi * (tall - tpre) / Tt;        /* the duration goes roughly like 1 / w */
  int found = 0;
  for (int it = 0; it < 80; it++) {
    const double d1 = w - vm, T1 = scurve_time(d1, amax, jmax), T2 = scurve_time(w, amax, jmax);
    const double dnc = 0.5 * (vm + w) * T1 + 0.5 * w * T2, rest = Dm - dnc;
    const double g = T1 + T2 + rest / w - Tt;
    if (g == 0) { found = 1; break; }
    if (g > 0) lo = w; else hi = w;
    const double sd = d1 >= 0 ? 1.0 : -1.0;
    const double T1p = sd * (fabs(d1) >= vtri ? 1.0 / amax : (fabs(d1) > 0 ? 1.0 / sqrt(jmax * fabs(d1)) : 0.0));
    const double T2p = w >= vtri ? 1.0 / amax : (w > 0 ? 1.0 / sqrt(jmax * w) : 0.0);
    const double dncp = 0.5 * T1 + 0.5 * (vm + w) * T1p + 0.5 * T2 + 0.5 * w * T2p;
    const double gp = T1p + T2p - dncp / w - rest / (w * w);
    double nw = gp < 0 ? w - g / gp : 0.5 * (lo + hi);
    if (!(nw > lo && nw < hi)) nw = 0.5 * (lo + hi);
    const double step = fabs(nw - w);
    w = nw;
    found = 1;
    if (step <= 4e-16 * (1.0 + w) || hi - lo <= 4e-16 * (1.0 + hi)) break;
  }
  if (!found || !(w > 0)) return;
  const double tc = (Dm - dist_nocruise(vm, w, amax, jmax)) / w;
  if (!(tc >= 0)) return;                    /* (the whole remaining distance is needed to stop: nothing to stretch) */
  int n = tl->n0;
  n += scurve(tl->v, tl->sg * w, amax, jmax, dur + n, jerk + n);
  dur[n] = tc; jerk[n] = 0; n++;
  n += scurve(tl->sg * w, 0, amax, jmax, dur + n, jerk + n);
}

static void ltt_plan(const hrg_model_desc* m, hrg_ltt* L, const double* q0, const double* v0, const double* a0, const double* goal) {
  double T = 0;
  ltt_tail tail[NARM];
  for (int j = 0; j < NARM; j++) {
    ltt_plan_joint(L, j, q0[j], v0[j], a0[j], goal[j], m->v_max_ltt[j], m->a_max_ltt[j], m->j_max_ltt[j], &tail[j]);
    double t = 0;
    for (int i = 0; i < HRG_LTT_NSEG; i++) t += L->dur[j][i];
    if (t > T) T = t;
  }
  L->T = T;
  if (m->ltt_time_sync)
    for (int j = 0; j < NARM; j++) ltt_sync_joint(L, j, T, m->a_max_ltt[j], m->j_max_ltt[j], &tail[j]);
}
static void ltt_const(hrg_ltt* L, const double* q) {
  memset(L, 0, sizeof *L);
  for (int j = 0; j < NARM; j++) { L->q0[j] = q[j]; L->qT[j] = q[j]; }
}
/* state of joint j at trajectory time s: q, q', q'' (derivatives w.r.t. s) */
static void ltt_eval(const hrg_ltt* L, int j, double s, double* q, double* v, double* a) {
  double qq = L->q0[j], vv = L->v0[j], aa = L->a0[j], t = s;
  if (t < 0) t = 0;
  for (int i = 0; i < HRG_LTT_NSEG; i++) {
    double d = L->dur[j][i], jj = L->jerk[j][i];
    if (t < d) {
      *q = qq + vv * t + 0.5 * aa * t * t + jj * t * t * t * SIXTH;
      *v = vv + aa * t + 0.5 * jj * t * t;
      *a = aa + jj * t;
      return;
    }
    qq += vv * d + 0.5 * aa * d * d + jj * d * d * d * SIXTH;
    vv += aa * d + 0.5 * jj * d * d;
    aa += jj * d;
    t -= d;
  }
  *q = L->qT[j]; *v = 0; *a = 0;
}

/* =============================================================================================== path
 * Three-phase jerk-limited profile of the path velocity from (v0,a0) to (ve,0): the fail-safe (ve=0) and
 * recovery (ve=1) manoeuvres of the shield (sara-shield planSafetyShield; Beckert/Pereira/Althoff 2017). */
static void path_plan(hrg_path* P, double s0, double v0, double a0, double ve, double amax, double jmax) {
  P->s0 = s0; P->v0 = v0; P->a0 = a0; P->k = 0;
  for (int i = 0; i < 3; i++) { P->dur[i] = 0; P->jerk[i] = 0; }
  if (fabs(v0 - ve) < 1e-12 && fabs(a0) < 1e-12) { P->v0 = ve; P->a0 = 0; return; }
  double v_at = v0 + a0 * fabs(a0) / (2 * jmax);
  double dir = ve >= v_at ? 1.0 : -1.0;
  double A = dir * a0, dv = dir * (ve - v0);
  double apk = amax > A ? amax : A;
  double t1 = (apk - A) / jmax, t3 = apk / jmax;
  double dv2 = dv - 0.5 * (A + apk) * t1 - 0.5 * apk * t3, t2;
  if (dv2 >= 0) t2 = dv2 / apk;
  else {
    double r = 0.5 * A * A + jmax * dv;
    apk = sqrt(r > 0 ? r : 0);
    if (apk < A) apk = A;
    t1 = (apk - A) / jmax; t2 = 0; t3 = apk / jmax;
  }
  P->dur[0] = t1; P->dur[1] = t2; P->dur[2] = t3;
  P->jerk[0] = dir * jmax; P->jerk[1] = 0; P->jerk[2] = -dir * jmax;
}
static double path_total(const hrg_path* P) { return P->dur[0] + P->dur[1] + P->dur[2]; }
/* speed a profile ends at (what it was planned to brake to): the state after its three phases, snapped to the exact values 0 and 1 it can be planned for */
static double path_vend(const hrg_path* P) {
  double vv = P->v0, aa = P->a0;
  for (int i = 0; i < 3; i++) { const double d = P->dur[i], jj = P->jerk[i]; vv += aa * d + 0.5 * jj * d * d; aa += jj * d; }
  if (fabs(vv) < 1e-12) return 0.0;
  if (fabs(vv - 1.0) < 1e-12) return 1.0;
  return vv;
}
static void path_eval(const hrg_path* P, double t, double ve, double* s, double* v, double* a) {
  double ss = P->s0, vv = P->v0, aa = P->a0;
  for (int i = 0; i < 3; i++) {
    double d = P->dur[i], jj = P->jerk[i];
    if (t < d) {
      *s = ss + vv * t + 0.5 * aa * t * t + jj * t * t * t * SIXTH;
      *v = vv + aa * t + 0.5 * jj * t * t;
      *a = aa + jj * t;
      return;
    }
    ss += vv * d + 0.5 * aa * d * d + jj * d * d * d * SIXTH;
    vv += aa * d + 0.5 * jj * d * d;
    aa += jj * d;
    t -= d;
  }
  /* profile finished: hold the end velocity exactly */
  *s = ss + ve * t; *v = ve; *a = 0;
}

/* =============================================================================================== shield
 * SafetyShield.step / humanMeasurement / newLongTermTrajectory / getSafety as called from
 * failsafe_controller.py:300,310,329-332,381 — restated after SURVEY.md B.3 and arXiv 2205.06311:
 * candidate = one recovery step then fail-safe braking; robot reach capsules over the candidate; human reach
 * capsules (ACC / VEL / POS models) over the braking time; safe iff some model is disjoint from the robot. */
static void shield_arm_fk(const hrg_model_desc* m, const double* q6, double cp1[HRG_NSHIELD_RCAP][3], double cp2[HRG_NSHIELD_RCAP][3]) {
  double q[NV];
  robot_kin k;
  for (int i = 0; i < NARM; i++) q[i] = q6[i];
  for (int i = NARM; i < NV; i++) q[i] = 0;
  robot_fk(m, q, &k);
  for (int c = 0; c < HRG_NSHIELD_RCAP; c++) {
    int b = m->scap_body[c];
    double t[3];
    m3mulv(t, k.R[b], m->scap_p1[c]); v3add(cp1[c], k.p[b], t);
    m3mulv(t, k.R[b], m->scap_p2[c]); v3add(cp2[c], k.p[b], t);
  }
}

static void motion_at(const hrg_ltt* L, double s, double sv, double sa, double* q, double* v, double* a) {
  for (int j = 0; j < NARM; j++) {
    double qq, q1, q2;
    ltt_eval(L, j, s, &qq, &q1, &q2);
    q[j] = qq; v[j] = q1 * sv; a[j] = q1 * sa + q2 * sv * sv;
  }
}

static void shield_reset(const hrg_model_desc* m, hrg_env_state* s, const double* q) {
  ltt_const(&s->ltt, q);
  path_plan(&s->safe_path, 0, 0, 0, 0, m->path_amax, m->path_jmax);
  s->path_s = 0; s->path_v = 0; s->path_a = 0;
  s->new_goal = 0; s->is_safe = 1; s->n_meas = 0;
  for (int j = 0; j < NARM; j++) { s->des_q[j] = q[j]; s->des_v[j] = 0; s->des_a[j] = 0; s->new_goal_q[j] = q[j]; }
  memset(s->meas_prev, 0, sizeof s->meas_prev);
  s->meas_prev_t = 0;
}

/* human reach capsules for the horizon T; returns count; model id per capsule in mdl[] (0 ACC,1 VEL,2 POS) */
static int human_reach(const hrg_model_desc* m, const double meas[HRG_NHJ][3], const double vel[HRG_NHJ][3], int have_vel, double T,
                       double caps[HRG_NHCAP_MAX][7], int* mdl) {
  int n = 0;
  double Td = T + m->delay;
  for (int b = 0; b < m->n_bodypart; b++) { /* ACC */
    int i1 = m->bp_joint[b][0], i2 = m->bp_joint[b][1];
    double r1 = v3norm(vel[i1]) * Td * 0.5 + 0.5 * m->bp_amax[b] * Td * Td + m->meas_err_pos + m->meas_err_vel * Td;
    double r2 = v3norm(vel[i2]) * Td * 0.5 + 0.5 * m->bp_amax[b] * Td * Td + m->meas_err_pos + m->meas_err_vel * Td;
    v3madd(caps[n], meas[i1], vel[i1], 0.5 * Td);
    v3madd(caps[n] + 3, meas[i2], vel[i2], 0.5 * Td);
    caps[n][6] = (r1 > r2 ? r1 : r2) + m->bp_thickness[b];
    mdl[n++] = 0;
  }
  for (int b = 0; b < m->n_bodypart; b++) { /* VEL */
    int i1 = m->bp_joint[b][0], i2 = m->bp_joint[b][1];
    v3cpy(caps[n], meas[i1]);
    v3cpy(caps[n] + 3, meas[i2]);
    caps[n][6] = m->bp_thickness[b] + m->meas_err_pos + m->bp_vmax[b] * Td;
    mdl[n++] = 1;
  }
  for (int e = 0; e < m->n_extremity; e++) { /* POS: ball at the proximal joint */
    int i = m->ext_joint[e];
    v3cpy(caps[n], meas[i]);
    v3cpy(caps[n] + 3, meas[i]);
    caps[n][6] = m->ext_length[e] + m->ext_thickness[e] + m->meas_err_pos + m->ext_vmax[e] * Td;
    mdl[n++] = 2;
  }
  for (int b = 0; b < m->n_bodypart; b++) { /* POS: non-extremity parts keep their VEL capsule */
    if (!m->bp_in_pos[b]) continue;
    int i1 = m->bp_joint[b][0], i2 = m->bp_joint[b][1];
    v3cpy(caps[n], meas[i1]);
    v3cpy(caps[n] + 3, meas[i2]);
    caps[n][6] = m->bp_thickness[b] + m->meas_err_pos + m->bp_vmax[b] * Td;
    mdl[n++] = 2;
  }
  (void)have_vel;
  return n;
}

static void shield_step(hrgo_batch* B, int e, double t) {
  const hrg_model_desc* m = &B->m;
  hrg_env_state* s = &B->st[e];
  const double dt = m->timestep;
  /* ---- humanMeasurement (failsafe_controller.py:302-310): finite-difference joint velocities ---- */
  double vel[HRG_NHJ][3];
  int have_vel = s->n_meas >= 1 && t > s->meas_prev_t;
  for (int j = 0; j < HRG_NHJ; j++)
    for (int a = 0; a < 3; a++) vel[j][a] = have_vel ? (s->human_site[j][a] - s->meas_prev[j][a]) / (t - s->meas_prev_t) : 0.0;
  memcpy(s->meas_prev, s->human_site, sizeof s->meas_prev);
  s->meas_prev_t = t;
  if (s->n_meas < 2) s->n_meas++;
  /* ---- current motion ---- */
  double cq[NARM], cv[NARM], ca[NARM];
  motion_at(&s->ltt, s->path_s, s->path_v, s->path_a, cq, cv, ca);
  /* ---- new goal: plan a candidate long-term trajectory from the NOMINAL state of the active trajectory at the current
   * path position (q, dq/ds, d2q/ds2).  The candidate then continues with the current path velocity/acceleration, so it
   * can be swapped in at any path speed without a jump in the commanded motion (at s' = 1 nominal = actual). ---- */
  hrg_ltt cand;
  int use_cand = 0;
  if (s->new_goal) {
    double nq[NARM], nv[NARM], na[NARM];
    int plannable = 1;
    for (int j = 0; j < NARM; j++) {
      ltt_eval(&s->ltt, j, s->path_s, &nq[j], &nv[j], &na[j]);
      if (fabs(na[j]) > m->a_max_ltt[j]) plannable = 0;
    }
    if (plannable) { ltt_plan(m, &cand, nq, nv, na, s->new_goal_q); use_cand = 1; }
  }
  const hrg_ltt* L = use_cand ? &cand : &s->ltt;
  double ps = use_cand ? 0.0 : s->path_s, pv = s->path_v, pa = s->path_a;
  /* ---- candidate: one recovery step towards s'=1, then fail-safe brake to s'=0 ---- */
  hrg_path rec, fs2;
  double s1, v1, a1, se, ve_, ae;
  path_plan(&rec, ps, pv, pa, 1.0, m->path_amax, m->path_jmax);
  path_eval(&rec, dt, 1.0, &s1, &v1, &a1);
  /* speed the fail-safe manoeuvre brakes to: a full stop (SSM), or under PFL the path speed at which no point of the arm exceeds pfl_v_safe on this
   * trajectory: |v_point| <= s' sum_j |dq_j/ds| r_j, dq/ds taken where the manoeuvre starts (re-evaluated every cycle) */
  double ve_fs = m->failsafe_sdot;
  if (m->shield_type == HRG_SHIELD_PFL) {
    double vc = 0;
    for (int j = 0; j < NARM; j++) { double q_, d1_, d2_; ltt_eval(L, j, s1, &q_, &d1_, &d2_); vc += fabs(d1_) * m->pfl_reach[j]; }
    ve_fs = vc > m->pfl_v_safe ? m->pfl_v_safe / vc : 1.0;
  }
  path_plan(&fs2, s1, v1, a1, ve_fs, m->path_amax, m->path_jmax);
  double Tb = path_total(&fs2);
  path_eval(&fs2, Tb, ve_fs, &se, &ve_, &ae);
  int safe = 1;
  if (m->shield_type != HRG_SHIELD_OFF) {
    /* robot reach over [current config, config at the end of the brake] */
    double qe[NARM], d1, d2;
    for (int j = 0; j < NARM; j++) ltt_eval(L, j, se, &qe[j], &d1, &d2);
    double a1p[HRG_NSHIELD_RCAP][3], a2p[HRG_NSHIELD_RCAP][3], b1p[HRG_NSHIELD_RCAP][3], b2p[HRG_NSHIELD_RCAP][3];
    shield_arm_fk(m, cq, a1p, a2p);
    shield_arm_fk(m, qe, b1p, b2p);
    double sdiff = se - ps;
    double (*rc)[7] = B->rcaps[e];
    for (int c = 0; c < HRG_NSHIELD_RCAP; c++) {
      double d[3], l1, l2;
      v3sub(d, a1p[c], b1p[c]); l1 = v3norm(d);
      v3sub(d, a2p[c], b2p[c]); l2 = v3norm(d);
      for (int a = 0; a < 3; a++) { rc[c][a] = 0.5 * (a1p[c][a] + b1p[c][a]); rc[c][3 + a] = 0.5 * (a2p[c][a] + b2p[c][a]); }
      rc[c][6] = m->scap_r[c] + m->secure_radius + 0.5 * (l1 > l2 ? l1 : l2) + m->scap_alpha[c] * sdiff * sdiff / 8.0;
    }
    /* human reach over the braking time */
    int mdl[HRG_NHCAP_MAX];
    double (*hc)[7] = B->hcaps[e];
    int nh = human_reach(m, (const double(*)[3])s->human_site, (const double(*)[3])vel, have_vel, dt + Tb, hc, mdl);
    B->n_hcaps[e] = nh;
    int hit[3] = {0, 0, 0};
    if (!have_vel) hit[0] = 1; /* no velocity estimate yet: the ACC model cannot certify */
    for (int k = 0; k < nh; k++)
      for (int c = 0; c < HRG_NSHIELD_RCAP; c++) {
        double c1[3], c2[3], rr = rc[c][6] + hc[k][6];
        if (seg_seg(rc[c], rc[c] + 3, hc[k], hc[k] + 3, c1, c2) < rr * rr) hit[mdl[k]] = 1;
      }
    safe = !(hit[0] && hit[1] && hit[2]);
  }
  if (safe) {
    if (use_cand) { s->ltt = cand; s->new_goal = 0; }
    s->path_s = s1; s->path_v = v1; s->path_a = a1;
    s->safe_path = fs2;
  } else {
    /* follow the last verified fail-safe profile */
    double ns, nv_, na_;
    s->safe_path.k += 1.0;
    const double vend = m->shield_type == HRG_SHIELD_PFL ? path_vend(&s->safe_path) : m->failsafe_sdot; /* the speed that profile was planned to brake to */
    path_eval(&s->safe_path, s->safe_path.k * dt, vend, &ns, &nv_, &na_);
    /* the robot already moves at (or below) the fail-safe speed — stopped under SSM, at the PFL safe speed under PFL: a new
     * trajectory may be swapped in although it is not verified safe (sara-shield swaps "if safe or stopped") */
    if (use_cand && s->path_v <= vend + 1e-9 && fabs(s->path_a) <= 1e-9) {
      double adv = ns - s->path_s;
      s->ltt = cand; s->new_goal = 0;
      ns = adv;  /* the candidate's path axis starts at the current position */
      path_plan(&s->safe_path, ns, nv_, na_, vend, m->path_amax, m->path_jmax);
    }
    s->path_s = ns; s->path_v = nv_; s->path_a = na_;
  }
  s->is_safe = safe;
  motion_at(&s->ltt, s->path_s, s->path_v, s->path_a, s->des_q, s->des_v, s->des_a);
}

/* =============================================================================================== contacts
 * Stand-in for mj_collision on this model: every collision geom is the bounding capsule of the reference
 * mesh (tools/compile_model.py), the table is its top face with the slab footprint, the floor a plane.
 * geom ids: robot capsules 0..9, human bodies 10..33, table 34, floor 35.  Contacts are emitted in pair
 * enumeration order (robot-robot, robot-human, robot-table, robot-floor), which is the order
 * HumanEnv._collision_detection walks (human_env.py:1094). */
#define GEOM_HUMAN0 HRG_NRCAP
#define GEOM_TABLE (HRG_NRCAP + HRG_NHB)
#define GEOM_FLOOR (GEOM_TABLE + 1)
typedef struct {
  int g1, g2, b1, b2; /* geom ids, robot body of each side (-1 static, -2 not robot) */
  double dist, n[3], pos[3];
} contact_t;

#define GEOM_BOX (GEOM_FLOOR + 1)

/* ---- manipulation object: a cube with a free joint (BoxObject, pick_place_human_cartesian_env.py:660-678) ---- */
/* Closest points of a segment p1-p2 and a cube (centre c, rotation R row-major, half edge hb): squared distance of
 * f(t) = |P(t) - clamp(P(t))|^2 in the cube frame, a convex piecewise quadratic in t, minimised on [0,1] by a
 * safeguarded Newton iteration (each step is exact inside one piece).  Stand-in for mjc_CapsuleBox. */
static double seg_box(const double* p1, const double* p2, const double* c, const double* R, const double* hb, double* tmin, double* on_seg, double* on_box) {
  double a[3], d[3], t0[3];
  v3sub(t0, p1, c);
  for (int k = 0; k < 3; k++) a[k] = R[k] * t0[0] + R[3 + k] * t0[1] + R[6 + k] * t0[2]; /* R' (p1 - c) */
  v3sub(t0, p2, p1);
  for (int k = 0; k < 3; k++) d[k] = R[k] * t0[0] + R[3 + k] * t0[1] + R[6 + k] * t0[2];
  double g0 = 0, g1 = 0, t;
  for (int k = 0; k < 3; k++) {
    double x0 = a[k], x1 = a[k] + d[k];
    g0 += (x0 > hb[k] ? x0 - hb[k] : (x0 < -hb[k] ? x0 + hb[k] : 0.0)) * d[k];
    g1 += (x1 > hb[k] ? x1 - hb[k] : (x1 < -hb[k] ? x1 + hb[k] : 0.0)) * d[k];
  }
  if (g0 >= 0) t = 0; /* convex: slope >= 0 at t = 0 */
  else if (g1 <= 0) t = 1;
  else {
    double lo = 0, hi = 1;
    t = -g0 / (g1 - g0);
    for (int it = 0; it < 10; it++) {
      double g = 0, H = 0;
      for (int k = 0; k < 3; k++) {
        double x = a[k] + t * d[k], e = x > hb[k] ? x - hb[k] : (x < -hb[k] ? x + hb[k] : 0.0);
        g += e * d[k];
        if (e != 0) H += d[k] * d[k];
      }
      if (fabs(g) <= 1e-13 * (g1 - g0)) break; /* the slope vanished up to rounding: t is the minimiser */
      if (g < 0) lo = t; else hi = t;
      double nt = H > 0 ? t - g / H : 0.5 * (lo + hi);
      if (!(nt > lo && nt < hi)) nt = 0.5 * (lo + hi);
      t = nt;
    }
  }
  double x[3], y[3], e2 = 0;
  for (int k = 0; k < 3; k++) { x[k] = a[k] + t * d[k]; y[k] = clampd(x[k], -hb[k], hb[k]); e2 += (x[k] - y[k]) * (x[k] - y[k]); }
  *tmin = t;
  for (int k = 0; k < 3; k++) {
    on_seg[k] = c[k] + R[3 * k] * x[0] + R[3 * k + 1] * x[1] + R[3 * k + 2] * x[2];
    on_box[k] = c[k] + R[3 * k] * y[0] + R[3 * k + 1] * y[1] + R[3 * k + 2] * y[2];
  }
  return e2;
}

/* A capsule lying along a box face touches it in a stretch, not a point (MuJoCo's capsule-box generates two contacts there): kn = the box axis the
 * closest pair (cs on the capsule axis, cb on the box) is separated along; the stretch = the part of the axis whose other two box coordinates stay
 * inside the box.  Returns 1 with the axis / box point pairs at both ends of the stretch when both are closer than r and at least 1 mm apart. */
static int cap_box_two(const double* p1, const double* p2, const double* c, const double* R, const double* hb, double r, const double* cs, const double* cb,
                       double s2[2][3], double b2[2][3]) {
  double a[3], d[3], v[3], t0[3];
  v3sub(t0, p1, c);
  for (int k = 0; k < 3; k++) a[k] = R[k] * t0[0] + R[3 + k] * t0[1] + R[6 + k] * t0[2];
  v3sub(t0, p2, p1);
  for (int k = 0; k < 3; k++) d[k] = R[k] * t0[0] + R[3 + k] * t0[1] + R[6 + k] * t0[2];
  v3sub(t0, cs, cb);
  for (int k = 0; k < 3; k++) v[k] = R[k] * t0[0] + R[3 + k] * t0[1] + R[6 + k] * t0[2];
  int kn = 0;
  for (int k = 1; k < 3; k++) if (fabs(v[k]) > fabs(v[kn])) kn = k;
  double lo = 0, hi = 1;
  for (int k = 0; k < 3; k++) {
    if (k == kn) continue;
    if (fabs(d[k]) < 1e-12) { if (fabs(a[k]) > hb[k]) return 0; }
    else {
      double ta = (-hb[k] - a[k]) / d[k], tb = (hb[k] - a[k]) / d[k];
      if (ta > tb) { double t = ta; ta = tb; tb = t; }
      if (ta > lo) lo = ta;
      if (tb < hi) hi = tb;
    }
  }
  if (!(hi > lo)) return 0;
  if ((hi - lo) * sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) < 1e-3) return 0;
  for (int e = 0; e < 2; e++) {
    const double t = e ? hi : lo;
    double x[3], y[3], e2 = 0;
    for (int k = 0; k < 3; k++) { x[k] = a[k] + t * d[k]; y[k] = clampd(x[k], -hb[k], hb[k]); e2 += (x[k] - y[k]) * (x[k] - y[k]); }
    const double dd = sqrt(e2);
    if (!(dd - r < 0) || !(dd > 1e-9)) return 0;
    for (int k = 0; k < 3; k++) {
      s2[e][k] = c[k] + R[3 * k] * x[0] + R[3 * k + 1] * x[1] + R[3 * k + 2] * x[2];
      b2[e][k] = c[k] + R[3 * k] * y[0] + R[3 * k + 1] * y[1] + R[3 * k + 2] * y[2];
    }
  }
  return 1;
}

/* ---- convex hulls of the arm links (robot_hulls = 1): MuJoCo convexifies a mesh geom at compile time and collides its hull (robot.xml:29-55).
 * Restated with the textbook tools: the support mapping of the hull (the vertex farthest along a direction) and the GJK distance iteration (Gilbert, Johnson,
 * Keerthi 1988; closest-point sub-problems as in Ericson, Real-Time Collision Detection 5.1.2 / 5.1.5 / 5.1.6) between the hull and the AXIS of a capsule -- the
 * capsule's radius is taken off the distance afterwards, which is exact while the axis stays outside the hull. ---- */
typedef struct { const double* v; int n; const double* R; const double* p; } hull_t; /* body-frame vertices, world pose of the body */
/* vertex of the hull farthest along the world direction d (the first one on a tie): index, world point in out */
static int hull_support(const hull_t* H, const double* d, double* out) {
  const double dl[3] = {H->R[0] * d[0] + H->R[3] * d[1] + H->R[6] * d[2], H->R[1] * d[0] + H->R[4] * d[1] + H->R[7] * d[2], H->R[2] * d[0] + H->R[5] * d[1] + H->R[8] * d[2]};
  int best = 0;
  double bv = -1e300;
  for (int i = 0; i < H->n; i++) {
    const double t = H->v[3 * i] * dl[0] + H->v[3 * i + 1] * dl[1] + H->v[3 * i + 2] * dl[2];
    if (t > bv) { bv = t; best = i; }
  }
  m3mulv(out, H->R, H->v + 3 * best);
  v3add(out, out, H->p);
  return best;
}
/* barycentric coordinates of the point of triangle (a, b, c) closest to the origin (Ericson 5.1.5) */
static void closest_triangle(const double* a, const double* b, const double* c, double* l) {
  double ab[3], ac[3];
  v3sub(ab, b, a); v3sub(ac, c, a);
  const double d1 = -v3dot(ab, a), d2 = -v3dot(ac, a);
  if (d1 <= 0 && d2 <= 0) { l[0] = 1; l[1] = 0; l[2] = 0; return; }
  const double d3 = -v3dot(ab, b), d4 = -v3dot(ac, b);
  if (d3 >= 0 && d4 <= d3) { l[0] = 0; l[1] = 1; l[2] = 0; return; }
  const double vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) { const double v = d1 / (d1 - d3); l[0] = 1 - v; l[1] = v; l[2] = 0; return; }
  const double d5 = -v3dot(ab, c), d6 = -v3dot(ac, c);
  if (d6 >= 0 && d5 <= d6) { l[0] = 0; l[1] = 0; l[2] = 1; return; }
  const double vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) { const double w = d2 / (d2 - d6); l[0] = 1 - w; l[1] = 0; l[2] = w; return; }
  const double va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) { const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6)); l[0] = 0; l[1] = 1 - w; l[2] = w; return; }
  const double den = 1.0 / (va + vb + vc), v = vb * den, w = vc * den;
  l[0] = 1 - v - w; l[1] = v; l[2] = w;
}
typedef struct { int n; double y[4][3], a[4][3], b[4][3]; int ia[4], ib[4]; } gjk_simplex;
/* point of the simplex closest to the origin: v, the simplex reduced to the vertices that carry it, their weights in lam; returns 1 when the origin is inside */
static int simplex_closest(gjk_simplex* S, double* v, double* lam) {
  double l[4] = {0, 0, 0, 0};
  int keep[4] = {0, 1, 2, 3}, nk = S->n;
  if (S->n == 1) l[0] = 1;
  else if (S->n == 2) {
    double ab[3];
    v3sub(ab, S->y[1], S->y[0]);
    const double den = v3dot(ab, ab);
    double t = den > 0 ? -v3dot(S->y[0], ab) / den : 0.0;
    t = t < 0 ? 0 : (t > 1 ? 1 : t);
    l[0] = 1 - t; l[1] = t;
  } else if (S->n == 3) closest_triangle(S->y[0], S->y[1], S->y[2], l);
  else { /* tetrahedron (Ericson 5.1.6): the closest of the faces the origin lies outside of; inside all four -> the origin is in the simplex */
    static const int F[4][4] = {{0, 1, 2, 3}, {0, 2, 3, 1}, {0, 3, 1, 2}, {1, 3, 2, 0}}; /* face vertices, then the opposite vertex */
    double best = 1e300;
    int any = 0;
    for (int f = 0; f < 4; f++) {
      const double *a = S->y[F[f][0]], *b = S->y[F[f][1]], *c = S->y[F[f][2]], *d = S->y[F[f][3]];
      double ab[3], ac[3], nrm[3], ad[3];
      v3sub(ab, b, a); v3sub(ac, c, a); v3cross(nrm, ab, ac); v3sub(ad, d, a);
      const double sp = -v3dot(a, nrm), sd = v3dot(ad, nrm);
      if (!(sp * sd < 0) && sd != 0) continue; /* the origin is on the inner side of this face (a flat tetrahedron counts as outside) */
      double lf[3], q[3];
      closest_triangle(a, b, c, lf);
      for (int k = 0; k < 3; k++) q[k] = lf[0] * a[k] + lf[1] * b[k] + lf[2] * c[k];
      const double dq = v3dot(q, q);
      if (dq < best) { best = dq; any = 1; l[0] = l[1] = l[2] = l[3] = 0; l[F[f][0]] = lf[0]; l[F[f][1]] = lf[1]; l[F[f][2]] = lf[2]; }
    }
    if (!any) { v3set(v, 0, 0, 0); return 1; }
  }
  nk = 0;
  for (int i = 0; i < S->n; i++) if (l[i] > 0) keep[nk++] = i;
  v3set(v, 0, 0, 0);
  for (int q = 0; q < nk; q++) {
    const int i = keep[q];
    lam[q] = l[i];
    for (int k = 0; k < 3; k++) v[k] += l[i] * S->y[i][k];
    if (q != i) { v3cpy(S->y[q], S->y[i]); v3cpy(S->a[q], S->a[i]); v3cpy(S->b[q], S->b[i]); S->ia[q] = S->ia[i]; S->ib[q] = S->ib[i]; }
  }
  S->n = nk;
  return 0;
}
/* distance between the hull and the segment [s1, s2]; witness points wa (on the hull) and wb (on the segment).  0 when they intersect. */
static double gjk_hull_segment(const hull_t* H, const double* s1, const double* s2, double* wa, double* wb) {
  gjk_simplex S;
  S.n = 0;
  double v[3], lam[4] = {1, 0, 0, 0};
  { /* start: the hull's first vertex against the segment's first end */
    double a0[3];
    m3mulv(a0, H->R, H->v); v3add(a0, a0, H->p);
    v3sub(v, a0, s1);
    v3cpy(S.y[0], v); v3cpy(S.a[0], a0); v3cpy(S.b[0], s1); S.ia[0] = 0; S.ib[0] = 0; S.n = 1;
  }
  for (int it = 0; it < 64; it++) {
    const double vv = v3dot(v, v);
    if (vv <= 1e-24) { v3cpy(wa, S.a[0]); v3cpy(wb, S.a[0]); return 0.0; }
    double dir[3] = {-v[0], -v[1], -v[2]}, a[3], w[3];
    const int ia = hull_support(H, dir, a);
    const int ib = v3dot(v, s2) > v3dot(v, s1) ? 1 : 0; /* support of the segment along +v */
    const double* b = ib ? s2 : s1;
    v3sub(w, a, b);
    if (vv - v3dot(v, w) <= 1e-12 * vv) break;      /* no vertex pair gets closer along -v: v is the closest point of A - B */
    int seen = 0;
    for (int q = 0; q < S.n; q++) if (S.ia[q] == ia && S.ib[q] == ib) seen = 1;
    if (seen) break;
    v3cpy(S.y[S.n], w); v3cpy(S.a[S.n], a); v3cpy(S.b[S.n], b); S.ia[S.n] = ia; S.ib[S.n] = ib; S.n++;
    if (simplex_closest(&S, v, lam)) { v3cpy(wa, a); v3cpy(wb, a); return 0.0; }
  }
  v3set(wa, 0, 0, 0); v3set(wb, 0, 0, 0);
  for (int q = 0; q < S.n; q++) for (int k = 0; k < 3; k++) { wa[k] += lam[q] * S.a[q][k]; wb[k] += lam[q] * S.b[q][k]; }
  return sqrt(v3dot(v, v));
}
/* the hull against a horizontal plane: out = (x, y, z) with z the height of the hull's lowest vertex and (x, y) the mean of the vertices within 1e-6 m of it -- a
 * link that lies flat on a face or an edge touches in the middle of that face or edge, whichever vertex rounding makes the lowest */
static void hull_lowest(const hull_t* H, double* out) {
  double zmin = 1e300;
  for (int i = 0; i < H->n; i++) {
    const double z = H->p[2] + H->R[6] * H->v[3 * i] + H->R[7] * H->v[3 * i + 1] + H->R[8] * H->v[3 * i + 2];
    if (z < zmin) zmin = z;
  }
  double sx = 0, sy = 0, cnt = 0;
  for (int i = 0; i < H->n; i++) {
    const double* v = H->v + 3 * i;
    const double z = H->p[2] + H->R[6] * v[0] + H->R[7] * v[1] + H->R[8] * v[2];
    if (z <= zmin + 1e-6) {
      sx += H->p[0] + H->R[0] * v[0] + H->R[1] * v[1] + H->R[2] * v[2];
      sy += H->p[1] + H->R[3] * v[0] + H->R[4] * v[1] + H->R[5] * v[2];
      cnt += 1;
    }
  }
  out[0] = sx / cnt; out[1] = sy / cnt; out[2] = zmin;
}

typedef struct { double pos[3], n[3], dist; } bb_contact;
static int box_box2(const double* pa, const double* Ra, const double* ha, const double* pb, const double* Rb, const double* hb, bb_contact out[4]);
static int collide(const hrg_model_desc* m, const robot_kin* k, const human_kin* h, const hrg_box_state* bx, contact_t* con) {
  double rp1[HRG_NRCAP][3], rp2[HRG_NRCAP][3];
  double Rb[9];
  quat2mat(Rb, m->base_quat);
  for (int c = 0; c < HRG_NRCAP; c++) {
    int b = m->rcap_body[c];
    const double* R = b < 0 ? Rb : k->R[b];
    const double* p = b < 0 ? m->base_pos : k->p[b];
    double t[3];
    m3mulv(t, R, m->rcap_p1[c]); v3add(rp1[c], p, t);
    m3mulv(t, R, m->rcap_p2[c]); v3add(rp2[c], p, t);
  }
  int n = 0;
#define EMIT(G1, G2, B1, B2, DIST, NRM, POS) \
  do { if (n < HRG_NCON_MAX) { con[n].g1 = G1; con[n].g2 = G2; con[n].b1 = B1; con[n].b2 = B2; con[n].dist = DIST; v3cpy(con[n].n, NRM); v3cpy(con[n].pos, POS); n++; } } while (0)
  for (int i = 0; i < HRG_NRCAP; i++)
    for (int j = i + 1; j < HRG_NRCAP; j++) {
      if (!((m->rcap_selfmask[i] >> j) & 1u)) continue;
      double c1[3], c2[3], d[3], nn[3] = {0, 0, 1}, pos[3];
      double d2 = seg_seg(rp1[i], rp2[i], rp1[j], rp2[j], c1, c2), dd = sqrt(d2), dist = dd - m->rcap_r[i] - m->rcap_r[j];
      if (dist < 0) {
        v3sub(d, c2, c1);
        if (dd > 1e-12) v3scl(nn, d, 1.0 / dd);
        v3madd(pos, c1, nn, m->rcap_r[i] + 0.5 * dist);
        EMIT(i, j, m->rcap_body[i], m->rcap_body[j], dist, nn, pos);
      }
    }
  for (int i = 0; i < HRG_NRCAP; i++)
    for (int b = 0; b < HRG_NHB; b++) {
      double c1[3], c2[3], d[3], nn[3] = {0, 0, 1}, pos[3];
      double d2 = seg_seg(rp1[i], rp2[i], h->cap1[b], h->cap2[b], c1, c2), dd = sqrt(d2), dist = dd - m->rcap_r[i] - m->hcap_r[b];
      if (dist < m->contact_margin_human) {
        v3sub(d, c2, c1);
        if (dd > 1e-12) v3scl(nn, d, 1.0 / dd);
        v3madd(pos, c1, nn, m->rcap_r[i] + 0.5 * dist);
        if (m->robot_hulls && i < HRG_NHULL) {
          /* robot_hulls: the capsule test was the BROADPHASE of an arm link (geoms 0 .. 6: robot.xml:29-55, mesh geoms that MuJoCo convexifies).  The pair now
           * runs the narrowphase of the link's CONVEX HULL: distance of the hull to the human capsule's axis minus its radius, normal along the witness points,
           * position half way between the two surfaces (an axis that pierces the hull -- a penetration deeper than the capsule's radius -- keeps the capsule
           * contact).  A hull that does not come within the margin drops the pair. */
          const int lb = m->rcap_body[i];
          const hull_t H = {m->hull_verts + 3 * m->hull_off[i], m->hull_off[i + 1] - m->hull_off[i], lb < 0 ? Rb : k->R[lb], lb < 0 ? m->base_pos : k->p[lb]};
          double wa[3], wb[3];
          const double dh = gjk_hull_segment(&H, h->cap1[b], h->cap2[b], wa, wb);
          if (dh > 1e-9) {
            dist = dh - m->hcap_r[b];
            if (!(dist < m->contact_margin_human)) continue;
            for (int a = 0; a < 3; a++) { nn[a] = (wb[a] - wa[a]) / dh; pos[a] = wa[a] + nn[a] * (0.5 * dist); }
          }
        }
        EMIT(i, GEOM_HUMAN0 + b, m->rcap_body[i], -2, dist, nn, pos);
      }
    }
  for (int pl = 0; pl < 2; pl++)
    for (int i = 0; i < HRG_NRCAP; i++) {
      if (m->rcap_body[i] < 0) continue; /* welded to the world: static-static pairs are filtered */
      int hull_done = 0;
      for (int e = 0; e < 2; e++) {
        const double* p = e ? rp2[i] : rp1[i];
        double z0 = pl ? m->floor_z : m->table_top_z, dist = p[2] - m->rcap_r[i] - z0;
        if (pl == 0 && !(fabs(p[0] - m->table_center[0]) <= m->table_half[0] && fabs(p[1] - m->table_center[1]) <= m->table_half[1] && p[2] > z0 - 0.025)) /* end point above the mid-plane of the 0.05 m slab */ continue;
        if (dist < 0) {
          double nn[3] = {0, 0, -1}, pos[3] = {p[0], p[1], z0 + 0.5 * dist};
          if (m->robot_hulls && i < HRG_NHULL) {
            /* robot_hulls: an end point of the bounding capsule under the plane = broadphase; the contact is the hull's lowest point, ONE per link and plane
             * (the capsule has one per end point), with the plane's own footprint test on that point. */
            if (hull_done) continue;
            hull_done = 1;
            const int lb = m->rcap_body[i];
            const hull_t H = {m->hull_verts + 3 * m->hull_off[i], m->hull_off[i + 1] - m->hull_off[i], k->R[lb], k->p[lb]};
            double low[3];
            hull_lowest(&H, low);
            dist = low[2] - z0;
            int ok = dist < 0;
            if (pl == 0) ok = ok && fabs(low[0] - m->table_center[0]) <= m->table_half[0] && fabs(low[1] - m->table_center[1]) <= m->table_half[1] && low[2] > z0 - 0.025;
            if (!ok) continue;
            pos[0] = low[0]; pos[1] = low[1]; pos[2] = z0 + 0.5 * dist;
          }
          EMIT(i, pl ? GEOM_FLOOR : GEOM_TABLE, m->rcap_body[i], -1, dist, nn, pos);
        }
      }
    }
  if (bx) { /* robot capsule - cube, table - cube corners, floor - cube corners */
    double Rx[9];
    quat2mat(Rx, bx->quat);
    const double* hb = m->box_half;
    int n_second = 0, second_i[HRG_NRCAP];
    double second_s[HRG_NRCAP][3], second_b[HRG_NRCAP][3];
    for (int i = 0; i < HRG_NRCAP; i++) {
      if (m->rcap_body[i] < 0) continue;
      double t, cs[3], cb[3], nn[3], pos[3], s2[2][3], b2[2][3];
      double e2 = seg_box(rp1[i], rp2[i], bx->pos, Rx, hb, &t, cs, cb), dd = sqrt(e2), dist = dd - m->rcap_r[i];
      if (!(dist < 0)) continue;
      if (dd > 1e-9 && cap_box_two(rp1[i], rp2[i], bx->pos, Rx, hb, m->rcap_r[i], cs, cb, s2, b2)) { /* along a face: both ends of the stretch; the second
                                                                                                       * one joins the contact list behind the corner contacts */
        v3cpy(cs, s2[0]); v3cpy(cb, b2[0]);
        v3sub(nn, cb, cs); dd = v3norm(nn); dist = dd - m->rcap_r[i];
        second_i[n_second] = i; v3cpy(second_s[n_second], s2[1]); v3cpy(second_b[n_second], b2[1]); n_second++;
      }
      if (dd > 1e-9) { v3sub(nn, cb, cs); v3scl(nn, nn, 1.0 / dd); }
      else { /* capsule axis inside the cube: push out through the nearest face */
        double loc[3], best = 1e300; int ax = 0;
        v3sub(pos, cs, bx->pos);
        for (int a = 0; a < 3; a++) { loc[a] = Rx[a] * pos[0] + Rx[3 + a] * pos[1] + Rx[6 + a] * pos[2]; if (hb[a] - fabs(loc[a]) < best) { best = hb[a] - fabs(loc[a]); ax = a; } }
        double sg = loc[ax] >= 0 ? -1.0 : 1.0; /* normal points from the capsule into the cube */
        for (int a = 0; a < 3; a++) nn[a] = sg * Rx[3 * a + ax];
        dist = -best - m->rcap_r[i];
      }
      v3madd(pos, cs, nn, m->rcap_r[i] + 0.5 * dist);
      EMIT(i, GEOM_BOX, m->rcap_body[i], BODY_BOX, dist, nn, pos);
    }
    if (m->task == HRG_TASK_LIFTING) {
      /* CollaborativeLiftingCart: the 1.0 x 0.4 m board over a table 0.4 m wide (collaborative_lifting_cartesian_env.py:280-284, 742-746) -- board and table top
       * overlap in a cross, no corner of either lies over the other: box-box contacts (MuJoCo: mjc_BoxBox; here the SAT + clipping of D13) of the board with the
       * table slab (5 cm thick, TableArena) instead of the corner-against-plane test of the cube tasks */
      const double pt[3] = {m->table_center[0], m->table_center[1], m->table_top_z - 0.025}, ht[3] = {m->table_half[0], m->table_half[1], 0.025};
      const double Rt[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      double d2 = 0;
      for (int a = 0; a < 3; a++) { const double la = fabs(bx->pos[a] - pt[a]) - ht[a]; if (la > 0) d2 += la * la; }
      const double ez = fabs(Rx[6]) * hb[0] + fabs(Rx[7]) * hb[1] + fabs(Rx[8]) * hb[2]; /* the board's half extent along z: a separating axis of the pair */
      if (!(d2 > hb[0] * hb[0] + hb[1] * hb[1] + hb[2] * hb[2] + 1e-9) && bx->pos[2] - ez < m->table_top_z + 1e-9 && bx->pos[2] + ez > m->table_top_z - 0.05 - 1e-9) { /* broadphase: the board's circumsphere against the slab itself, then its z extent */
        bb_contact bc[4];
        const int nb = box_box2(pt, Rt, ht, bx->pos, Rx, hb, bc);
        for (int q = 0; q < nb; q++) EMIT(GEOM_TABLE, GEOM_BOX, -1, BODY_BOX, bc[q].dist, bc[q].n, bc[q].pos);
      }
    }
    for (int pl = m->task == HRG_TASK_LIFTING ? 1 : 0; pl < 2; pl++)
      for (int cn = 0; cn < 8; cn++) {
        double loc[3] = {(cn & 1) ? hb[0] : -hb[0], (cn & 2) ? hb[1] : -hb[1], (cn & 4) ? hb[2] : -hb[2]}, p[3];
        m3mulv(p, Rx, loc);
        v3add(p, p, bx->pos);
        double z0 = pl ? m->floor_z : m->table_top_z, dist = p[2] - z0;
        if (pl == 0 && !(fabs(p[0] - m->table_center[0]) <= m->table_half[0] && fabs(p[1] - m->table_center[1]) <= m->table_half[1] && p[2] > z0 - 0.05)) continue;
        if (dist < 0) {
          double nn[3] = {0, 0, 1}, pos[3] = {p[0], p[1], z0 + 0.5 * dist};
          EMIT(pl ? GEOM_FLOOR : GEOM_TABLE, GEOM_BOX, -1, BODY_BOX, dist, nn, pos);
        }
      }
    for (int q = 0; q < n_second; q++) {
      const int i = second_i[q];
      double nn[3], pos[3];
      v3sub(nn, second_b[q], second_s[q]);
      const double dd = v3norm(nn), dist = dd - m->rcap_r[i];
      v3scl(nn, nn, 1.0 / dd);
      v3madd(pos, second_s[q], nn, m->rcap_r[i] + 0.5 * dist);
      EMIT(i, GEOM_BOX, m->rcap_body[i], BODY_BOX, dist, nn, pos);
    }
    /* human capsule - object (human.xml:5: contype / conaffinity 7, margin 0.001 -- the human's geoms collide with the manipulation object like everything else;
     * the human itself is animated and does not yield).  The same narrowphase as for the arm's capsules, with the human geoms' contact margin; these contacts
     * close the list: the robot's and the table's come first into the HRG_NCON_DYN_BOX the solve takes.
     * NOT while the human holds the object (weld / connects active): the human here is a set of BOUNDING capsules of its meshes (D1), and an object in its hands
     * lies partly inside them -- the held board of the lifting task reaches into the capsules of pelvis and thighs, a welded cube into the hand's -- where the
     * reference's mesh hulls leave it free; those contacts would be artefacts of the approximation fighting the weld. */
    n_second = 0;
    int second_h[HRG_NHB];
    double second_hs[HRG_NHB][3], second_hb[HRG_NHB][3];
    for (int b = 0; b < HRG_NHB && !bx->weld_active && m->task != HRG_TASK_LIFTING; b++) { /* (lifting: the board's pose between the hands stays inside them when let go) */
      double t, cs[3], cb[3], nn[3], pos[3], s2[2][3], b2[2][3];
      const double r = m->hcap_r[b];
      double e2 = seg_box(h->cap1[b], h->cap2[b], bx->pos, Rx, hb, &t, cs, cb), dd = sqrt(e2), dist = dd - r;
      if (!(dist < m->contact_margin_human)) continue;
      if (dist < 0 && dd > 1e-9 && cap_box_two(h->cap1[b], h->cap2[b], bx->pos, Rx, hb, r, cs, cb, s2, b2)) {
        v3cpy(cs, s2[0]); v3cpy(cb, b2[0]);
        v3sub(nn, cb, cs); dd = v3norm(nn); dist = dd - r;
        second_h[n_second] = b; v3cpy(second_hs[n_second], s2[1]); v3cpy(second_hb[n_second], b2[1]); n_second++;
      }
      if (dd > 1e-9) { v3sub(nn, cb, cs); v3scl(nn, nn, 1.0 / dd); }
      else {
        double loc[3], best = 1e300; int ax = 0;
        v3sub(pos, cs, bx->pos);
        for (int a = 0; a < 3; a++) { loc[a] = Rx[a] * pos[0] + Rx[3 + a] * pos[1] + Rx[6 + a] * pos[2]; if (hb[a] - fabs(loc[a]) < best) { best = hb[a] - fabs(loc[a]); ax = a; } }
        double sg = loc[ax] >= 0 ? -1.0 : 1.0;
        for (int a = 0; a < 3; a++) nn[a] = sg * Rx[3 * a + ax];
        dist = -best - r;
      }
      v3madd(pos, cs, nn, r + 0.5 * dist);
      EMIT(GEOM_HUMAN0 + b, GEOM_BOX, -2, BODY_BOX, dist, nn, pos);
    }
    for (int q = 0; q < n_second; q++) {
      const int b = second_h[q];
      double nn[3], pos[3];
      v3sub(nn, second_hb[q], second_hs[q]);
      const double dd = v3norm(nn), dist = dd - m->hcap_r[b];
      v3scl(nn, nn, 1.0 / dd);
      v3madd(pos, second_hs[q], nn, m->hcap_r[b] + 0.5 * dist);
      EMIT(GEOM_HUMAN0 + b, GEOM_BOX, -2, BODY_BOX, dist, nn, pos);
    }
  }
#undef EMIT
  return n;
}

/* the manipulation object is whitelisted: COLLISION_TYPE.ALLOWED (pick_place_human_cartesian_env.py:710-717) */
/* GEOM_BOX + c: cube c of the stacking task (1526-1549); the hammering task white-lists the hammer's geoms only, "the board is not white-listed"
 * (collaborative_hammering_cartesian_env.py:1325-1337) */
static int geom_class_t(int g, int task) {
  if (g < HRG_NRCAP) return HRG_GEOM_ROBOT;
  if (g < GEOM_TABLE) return HRG_GEOM_HUMAN;
  if (g < GEOM_BOX || task == HRG_TASK_REACH_BOX) return HRG_GEOM_STATIC;
  if (task == HRG_TASK_HAMMERING) return g == GEOM_BOX + HRG_HG_HANDLE || g == GEOM_BOX + HRG_HG_HEAD ? HRG_GEOM_ALLOWED : HRG_GEOM_STATIC;
  return HRG_GEOM_ALLOWED;
}
static int cantor(int a, int b) { return (a + b) * (a + b + 1) / 2 + b; } /* utils/pairing.py:4-16 */

/* HumanEnv._collision_detection + _on_*_detected, human_env.py:966-1123 */
static void classify(const hrg_model_desc* m, const robot_kin* k, hrg_env_state* s, const contact_t* con, int ncon,
                     const double rcap_center[HRG_NRCAP][3], int* has_collision, int* collision_type) {
  int cur[HRG_NPREV_MAX], ncur = 0;
  double tm = s->debounce_timer - m->timestep; /* human_env.py:1090 */
  s->debounce_timer = tm > 0 ? tm : 0;
  for (int c = 0; c < ncon; c++) {
    int t1 = geom_class_t(con[c].g1, m->task), t2 = geom_class_t(con[c].g2, m->task); /* ReachHuman's smallBox is not whitelisted (reach_human_env.py has no _setup_collision_info override) */
    if (t1 != HRG_GEOM_ROBOT && t2 != HRG_GEOM_ROBOT) continue;
    int h12 = cantor(con[c].g1, con[c].g2), h21 = cantor(con[c].g2, con[c].g1);
    if (ncur + 2 <= HRG_NPREV_MAX) { cur[ncur++] = h12; cur[ncur++] = h21; }
    int seen = 0;
    for (int i = 0; i < s->n_prev; i++) if (s->prev_pairs[i] == h12) seen = 1;
    if (seen) continue;
    int rg = t1 == HRG_GEOM_ROBOT ? con[c].g1 : con[c].g2;
    int ot = t1 == HRG_GEOM_ROBOT ? t2 : t1;
    *has_collision = 1; /* human_env.py:1055 */
    if (ot == HRG_GEOM_ROBOT) { *collision_type |= HRG_COL_ROBOT; s->n_collisions_robot++; }
    else if (ot == HRG_GEOM_HUMAN) {
      if (s->debounce_timer > 0) continue; /* human_env.py:981-983 */
      s->debounce_timer = m->collision_debounce_delay;
      double v[3];
      robot_point_vel(k, m->rcap_body[rg], rcap_center[rg], v); /* geom_xvelp, human_env.py:1008 */
      if (v3norm(v) <= m->safe_vel) { *collision_type |= HRG_COL_HUMAN; s->n_collisions_human++; }
      else { *collision_type |= HRG_COL_HUMAN_CRIT; s->n_collisions_critical++; }
    } else if (ot == HRG_GEOM_ALLOWED) { *collision_type |= HRG_COL_ALLOWED; }
    else { *collision_type |= HRG_COL_STATIC; s->n_collisions_static++; }
  }
  s->n_prev = ncur;
  for (int i = 0; i < ncur; i++) s->prev_pairs[i] = cur[i];
}

/* =============================================================================================== solver
 * mj_step for the robot tree: unconstrained acceleration, soft constraints (friction loss, joint limits,
 * pyramidal frictional contacts) solved by a primal Newton method with exact line search on MuJoCo's convex
 * objective  1/2 (a-a0)' M (a-a0) + sum_i s_i(J_i a - aref_i), then semi-implicit Euler with implicit joint
 * damping. */
enum { ROW_FRICTION = 0, ROW_UNILATERAL = 1, ROW_EQUALITY = 2 /* weld rows: quadratic on both sides */ };
typedef struct {
  int n;
  int type[NEFC_MAX];
  int nv; /* DoF of the system the rows act on: NV, or NVT with the manipulation object */
  int grp[NEFC_MAX]; /* contact rows: 4 * contact + pyramid edge (the noslip pass pairs opposing edges); -1: not a contact row */
  double J[NEFC_MAX][NVMAX], aref[NEFC_MAX], D[NEFC_MAX], floss[NEFC_MAX];
} efc_t;

static void impedance(const hrg_model_desc* m, double pos_minus_margin, double* imp, double* K, double* Bd) {
  double d0 = m->solimp[0], dmax = m->solimp[1], width = m->solimp[2], mid = m->solimp[3], power = m->solimp[4];
  double x = fabs(pos_minus_margin) / width, y;
  if (x >= 1) y = 1;
  else if (x <= 0) y = 0;
  else if (power == 2.0) { /* MuJoCo's default solimp power: the square, not a libm call */
    if (x <= mid) { const double u = x / mid; y = u * u * mid; }
    else { const double u = (1 - x) / (1 - mid); y = 1 - u * u * (1 - mid); }
  }
  else if (x <= mid) y = pow(x / mid, power) * mid;
  else y = 1 - pow((1 - x) / (1 - mid), power) * (1 - mid);
  *imp = d0 + y * (dmax - d0);
  double tc = m->solref[0], dr = m->solref[1];
  if (tc < 2 * m->timestep) tc = 2 * m->timestep;
  *K = 1.0 / (dmax * dmax * tc * tc * dr * dr);
  *Bd = 2.0 / (dmax * tc);
}

static void efc_add_b(const hrg_model_desc* m, efc_t* E, const double* J, const double* qd, int type, double pos, double margin, double floss, double diag, double Bd_own);
static void efc_add(const hrg_model_desc* m, efc_t* E, const double* J, const double* qd, int type, double pos, double margin, double floss, double diag) {
  efc_add_b(m, E, J, qd, type, pos, margin, floss, diag, -1.0);
}
/* Bd_own >= 0: the row's own damping of its reference acceleration (a joint's solreffriction in the direct format), else the model's solref */
static void efc_add_b(const hrg_model_desc* m, efc_t* E, const double* J, const double* qd, int type, double pos, double margin, double floss, double diag, double Bd_own) {
  /* R = (1-imp)/imp * diagApprox with MuJoCo's constant approximations (dof_invweight0 for joint rows, sum of
   * the two bodies' translational body_invweight0 for contact rows), not the exact J M^-1 J' */
  if (E->n >= NEFC_MAX) return;
  double vel = 0, nz = 0;
  for (int i = 0; i < E->nv; i++) { vel += J[i] * qd[i]; nz += fabs(J[i]); }
  if (!(nz > 0) || !(diag > 0)) return; /* row does not act on the robot tree */
  double imp, K, Bd;
  impedance(m, pos - margin, &imp, &K, &Bd);
  if (Bd_own >= 0) Bd = Bd_own;
  int r = E->n++;
  memcpy(E->J[r], J, sizeof(double) * E->nv);
  E->type[r] = type;
  E->grp[r] = -1;
  E->aref[r] = -Bd * vel - K * imp * (pos - margin);
  E->D[r] = 1.0 / ((1 - imp) / imp * diag);
  E->floss[r] = floss;
}

/* cost pieces of one row at x = J a - aref: value, first and second derivative */
static void row_cost(const efc_t* E, int r, double x, double* c, double* g, double* h) {
  double D = E->D[r];
  if (E->type[r] == ROW_EQUALITY) { *c = 0.5 * D * x * x; *g = D * x; *h = D; }
  else if (E->type[r] == ROW_UNILATERAL) {
    if (x < 0) { *c = 0.5 * D * x * x; *g = D * x; *h = D; } else { *c = 0; *g = 0; *h = 0; }
  } else {
    double f = E->floss[r], lim = f / D;
    if (x <= -lim) { *c = f * (-x - 0.5 * lim); *g = -f; *h = 0; }
    else if (x >= lim) { *c = f * (x - 0.5 * lim); *g = f; *h = 0; }
    else { *c = 0.5 * D * x * x; *g = D * x; *h = D; }
  }
}

/* which quadratic / linear piece of its cost a row is in at x */
static int row_zone(const efc_t* E, int r, double x) {
  if (E->type[r] == ROW_EQUALITY) return 0;
  if (E->type[r] == ROW_UNILATERAL) return x < 0;
  double lim = E->floss[r] / E->D[r];
  return x <= -lim ? -1 : (x >= lim ? 1 : 0);
}

static void solve(const hrg_model_desc* m, const double* M, const double* a0, const efc_t* E, double* a) {
  const int nv = E->nv; /* M is nv x nv, row-major */
  /* a: in = warm start, out = solution */
  double Ma0[NVMAX];
  for (int i = 0; i < nv; i++) { double t = 0; for (int j = 0; j < nv; j++) t += M[i * nv + j] * a0[j]; Ma0[i] = t; }
  if (E->n == 0) { memcpy(a, a0, sizeof(double) * nv); return; }
  /* pick the better of warm start and unconstrained acceleration */
  double cost_ws = 0, cost_a0 = 0;
  for (int pass = 0; pass < 2; pass++) {
    const double* x = pass ? a0 : a;
    double c = 0;
    for (int i = 0; i < nv; i++) { double t = 0; for (int j = 0; j < nv; j++) t += M[i * nv + j] * (x[j] - a0[j]); c += 0.5 * (x[i] - a0[i]) * t; }
    for (int r = 0; r < E->n; r++) {
      double y = -E->aref[r], cc, g, h;
      for (int i = 0; i < nv; i++) y += E->J[r][i] * x[i];
      row_cost(E, r, y, &cc, &g, &h);
      c += cc;
    }
    if (pass) cost_a0 = c; else cost_ws = c;
  }
  if (!(cost_ws < cost_a0)) memcpy(a, a0, sizeof(double) * nv);
  for (int it = 0; it < m->solver_iters; it++) {
    double x[NEFC_MAX], g[NVMAX], H[NVMAX * NVMAX];
    for (int i = 0; i < nv; i++) { double t = -Ma0[i]; for (int j = 0; j < nv; j++) t += M[i * nv + j] * a[j]; g[i] = t; }
    memcpy(H, M, sizeof(double) * nv * nv);
    for (int r = 0; r < E->n; r++) {
      double y = -E->aref[r], cc, gg, hh;
      for (int i = 0; i < nv; i++) y += E->J[r][i] * a[i];
      x[r] = y;
      row_cost(E, r, y, &cc, &gg, &hh);
      for (int i = 0; i < nv; i++) g[i] += E->J[r][i] * gg;
      if (hh != 0) for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) H[i * nv + j] += hh * E->J[r][i] * E->J[r][j];
    }
    double gn = 0, sc = 0;
    for (int i = 0; i < nv; i++) { gn += g[i] * g[i]; sc += Ma0[i] * Ma0[i]; }
    if (sqrt(gn) <= m->solver_tol * (1.0 + sqrt(sc))) break;
    double d[NVMAX], Md[NVMAX], p[NEFC_MAX];
    for (int i = 0; i < nv; i++) d[i] = -g[i];
    if (!chol(H, nv)) break;
    chol_solve(H, nv, d);
    for (int i = 0; i < nv; i++) { double t = 0; for (int j = 0; j < nv; j++) t += M[i * nv + j] * d[j]; Md[i] = t; }
    for (int r = 0; r < E->n; r++) { double t = 0; for (int i = 0; i < nv; i++) t += E->J[r][i] * d[i]; p[r] = t; }
    /* exact line search on phi(al) = cost(a + al d): safeguarded Newton on phi' */
    double dMd = 0, gd0 = 0;
    for (int i = 0; i < nv; i++) { dMd += d[i] * Md[i]; double t = -Ma0[i]; for (int j = 0; j < nv; j++) t += M[i * nv + j] * a[j]; gd0 += d[i] * t; }
    double al = 1.0, lo = 0, hi = -1, d1_0 = 0, noise = 0;
#ifdef HRG_LS_DEBUG
    double tr_al[40], tr_d1[40]; int tr_n = 0;
#endif
    for (int ls = 0; ls < 40; ls++) {
      double d1 = gd0 + al * dMd, d2 = dMd;
      for (int r = 0; r < E->n; r++) {
        double cc, gg, hh;
        row_cost(E, r, x[r] + al * p[r], &cc, &gg, &hh);
        d1 += gg * p[r];
        d2 += hh * p[r] * p[r];
      }
      if (ls == 0) { /* also need phi'(0) for the stopping scale */
        d1_0 = gd0;
        noise = fabs(gd0) + dMd; /* magnitude of the terms phi' is summed from: 1e-14 of it is rounding noise, not slope */
        for (int r = 0; r < E->n; r++) { double cc, gg, hh; row_cost(E, r, x[r], &cc, &gg, &hh); d1_0 += gg * p[r]; noise += fabs(gg * p[r]); }
      }
#ifdef HRG_LS_DEBUG
      tr_al[tr_n] = al; tr_d1[tr_n++] = d1;
#endif
      if (fabs(d1) <= 1e-10 * fabs(d1_0) || fabs(d1) <= 1e-14 * noise) break; /* converged, or down at the rounding noise of the sum's own terms */
      if (d1 < 0) lo = al; else hi = al;
      double nx = al - d1 / d2;
      if (hi < 0) { if (!(nx > lo)) nx = 2 * al; }
      else if (!(nx > lo && nx < hi)) nx = 0.5 * (lo + hi);
      al = nx;
    }
#ifdef HRG_LS_DEBUG
    if (tr_n >= 12) {
      fprintf(stderr, "LS it %d n %d rows %d d1_0 %.3e gd0 %.3e dMd %.3e:", it, tr_n, E->n, d1_0, gd0, dMd);
      for (int k = 0; k < tr_n; k++) fprintf(stderr, " (%.17g %.2e)", tr_al[k], tr_d1[k]);
      fprintf(stderr, "\n");
    }
#endif
    /* a full Newton step that stayed inside one quadratic piece of every row solves the problem exactly: the gradient at the
     * new point is zero up to rounding, so the next iteration would only confirm convergence */
    int exact = al == 1.0;
    for (int r = 0; r < E->n && exact; r++) if (row_zone(E, r, x[r]) != row_zone(E, r, x[r] + p[r])) exact = 0;
    for (int i = 0; i < nv; i++) a[i] += al * d[i];
    if (exact) break;
  }
}

/* MuJoCo's noslip post-pass (mj_solNoSlip [UPSTREAM]; switched on by collaborative_hammering_cartesian_env.py:1161, noslip_iterations = 20), restated from MuJoCo's
 * documentation of it: a projected Gauss-Seidel pass AFTER the main solver that updates only the friction dimensions -- dry friction (friction-loss rows) and contact
 * friction -- with the constraint regularisation removed (R = 0), every other constraint force held at the main solver's value:
 *   dry friction row i:  f_i <- clamp(f_i - res_i / A_ii, +-frictionloss),  res = J a - aref (the row's acceleration residual without R f), A = J M^-1 J';
 *   pyramidal contact:   each pair of opposing pyramid edges (n + mu t, n - mu t) keeps the sum of its two forces (the normal force) and moves along (1, -1) to the
 *                        exact minimiser of the unregularised dual cost, both forces kept >= 0:  y <- clamp(y - (res_0 - res_1) / (A_00 + A_11 - 2 A_01), +-mid);
 *   a sweep's cost improvement, scaled by 1 / (meaninertia nv), below noslip_tolerance ends the pass (the first sweep also counts the removed sum 1/2 R f^2).
 * Here in acceleration space: a = a_smooth + M^-1 J' f is updated with W_r = M^-1 J_r' whenever a force changes, so res_r = J_r a - aref_r.  The forces the pass
 * starts from are those of the primal solution, f_r = -s_r'(J_r a - aref_r).  With soft rows alone the nail of the hammering task creeps in under its own weight
 * (the friction row cancels 90 % of the free acceleration); with the pass its acceleration is the row's reference -b v exactly while |f| < frictionloss. */
static void noslip(const hrg_model_desc* m, const double* M, const efc_t* E, double* a) {
  const int nv = E->nv, n = E->n;
  if (m->noslip_iterations <= 0 || n == 0) return;
  double* L = (double*)malloc(sizeof(double) * nv * nv);
  double (*W)[NVMAX] = malloc(sizeof(double[NVMAX]) * (size_t)n);
  double f[NEFC_MAX], Ad[NEFC_MAX];
  int part[NEFC_MAX]; /* contact rows: index of the opposing edge's row (-1: none) */
  memcpy(L, M, sizeof(double) * nv * nv);
  if (!chol(L, nv)) { free(L); free(W); return; }
  for (int r = 0; r < n; r++) {
    double y = -E->aref[r], c, g, h;
    for (int i = 0; i < nv; i++) y += E->J[r][i] * a[i];
    row_cost(E, r, y, &c, &g, &h);
    f[r] = -g;
    part[r] = -1;
    if (E->grp[r] >= 0 && (E->grp[r] & 1) == 0 && r + 1 < n && E->grp[r + 1] == E->grp[r] + 1) part[r] = r + 1;
    const int fric = E->type[r] == ROW_FRICTION || part[r] >= 0 || (r > 0 && part[r - 1] == r);
    if (!fric) continue;
    memcpy(W[r], E->J[r], sizeof(double) * nv);
    chol_solve(L, nv, W[r]);
    double t = 0;
    for (int i = 0; i < nv; i++) t += E->J[r][i] * W[r][i];
    Ad[r] = t;
  }
  for (int it = 0; it < m->noslip_iterations; it++) {
    double imp = 0;
    if (it == 0) for (int r = 0; r < n; r++) imp += 0.5 * f[r] * f[r] / E->D[r];
    for (int r = 0; r < n; r++) { /* dry friction */
      if (E->type[r] != ROW_FRICTION) continue;
      double res = -E->aref[r];
      for (int i = 0; i < nv; i++) res += E->J[r][i] * a[i];
      const double A = Ad[r] > 1e-15 ? Ad[r] : 1e-15, old = f[r];
      f[r] = clampd(old - res / A, -E->floss[r], E->floss[r]);
      const double dl = f[r] - old;
      for (int i = 0; i < nv; i++) a[i] += W[r][i] * dl;
      imp -= dl * res + 0.5 * Ad[r] * dl * dl;
    }
    for (int r = 0; r < n; r++) { /* contact friction: pairs of opposing pyramid edges */
      if (part[r] < 0) continue;
      const int q = part[r];
      double res0 = -E->aref[r], res1 = -E->aref[q], A01 = 0;
      for (int i = 0; i < nv; i++) { res0 += E->J[r][i] * a[i]; res1 += E->J[q][i] * a[i]; A01 += E->J[q][i] * W[r][i]; }
      const double mid = 0.5 * (f[r] + f[q]), y0 = 0.5 * (f[r] - f[q]), K1 = Ad[r] + Ad[q] - 2.0 * A01;
      double y = 0;
      if (K1 >= 1e-15) y = clampd(y0 - (res0 - res1) / K1, -mid, mid);
      if (mid < 0) y = 0; /* (cannot happen: unilateral forces are >= 0) */
      const double dy = y - y0;
      f[r] = mid + y; f[q] = mid - y;
      for (int i = 0; i < nv; i++) a[i] += (W[r][i] - W[q][i]) * dy;
      imp -= dy * (res0 - res1) + 0.5 * K1 * dy * dy;
    }
    if (g_debug > 2) {
      fprintf(stderr, "[noslip] it %d imp %.3e scaled %.3e |", it, imp, imp * m->noslip_scale);
      for (int r = 0; r < n; r++) if (E->type[r] == ROW_FRICTION && E->floss[r] > 1) { double res = -E->aref[r]; for (int i = 0; i < nv; i++) res += E->J[r][i] * a[i]; fprintf(stderr, " nail f %.4g res %.3e aref %.3e", f[r], res, E->aref[r]); }
      fprintf(stderr, "\n");
    }
    if (imp * m->noslip_scale < m->noslip_tolerance) break;
  }
  free(L); free(W);
}

/* =============================================================================================== env */
static void mat2quat(double* q, const double* R);
/* Re: rotation matrix of the robot's right_hand body (= the finger bodies' frame) of the last forward pass, or NULL (tasks without a relative-quaternion observable) */
static void compute_obs_e(const hrg_model_desc* m, const hrg_env_state* s, const hrg_box_state* bx, const double* goal, const double* Re, float* obs);
static void compute_obs(const hrg_model_desc* m, const hrg_env_state* s, const hrg_box_state* bx, const double* goal, float* obs) { compute_obs_e(m, s, bx, goal, NULL, obs); }
static void compute_obs_e(const hrg_model_desc* m, const hrg_env_state* s, const hrg_box_state* bx, const double* goal, const double* Re, float* obs) {
  /* object-state: vec/dist eef -> {L hand, R hand, head} (human_env.py:1536-1590, reach_human_env.py:637-640),
   * then goal_difference (reach_human_env.py:649-651) */
  int sites[3] = {m->site_lhand, m->site_rhand, m->site_head};
  for (int i = 0; i < 3; i++) {
    double d[3];
    v3sub(d, s->human_site[sites[i]], s->eef_pos);
    obs[4 * i] = (float)d[0]; obs[4 * i + 1] = (float)d[1]; obs[4 * i + 2] = (float)d[2];
    obs[4 * i + 3] = (float)v3norm(d);
  }
  for (int j = 0; j < NARM; j++) obs[12 + j] = (float)(goal[j] - s->qpos[j]);
  /* robot0_proprio-state = joint_pos, joint_vel, eef_pos (reach_human_env.py:619-636); goal modality: desired_goal (646-647) */
  for (int j = 0; j < NARM; j++) { obs[18 + j] = (float)s->qpos[j]; obs[24 + j] = (float)s->qvel[j]; obs[33 + j] = (float)goal[j]; }
  for (int a = 0; a < 3; a++) obs[30 + a] = (float)s->eef_pos[a];
  for (int c = 39; c < HRG_OBS_DIM; c++) obs[c] = 0.0f;
  for (int f = 0; f < HRG_NFINGER; f++) { obs[53 + f] = (float)s->qpos[NARM + f]; obs[55 + f] = (float)s->qvel[NARM + f]; }
  if (bx && m->task != HRG_TASK_REACH_BOX) { /* PickPlaceHumanCart._setup_observables, pick_place_human_cartesian_env.py:726-841; gripper_aperture human_env.py:1508-1524 (ReachHuman does not observe its box) */
    for (int j = 0; j < NARM; j++) { obs[12 + j] = 0.0f; obs[33 + j] = 0.0f; }
    for (int a = 0; a < 3; a++) obs[12 + a] = (float)bx->quat[1 + a]; /* object_quat, (x, y, z, w) like T.convert_quat(..., to="xyzw") (human_robot_handover_cartesian_env.py:849-858) */
    obs[15] = (float)bx->quat[0];
    obs[39] = (float)bx->gripped;
    double ap = 0;
    for (int f = 0; f < HRG_NFINGER; f++) ap += (s->qpos[NARM + f] - m->finger_qpos_range[0][f]) / (m->finger_qpos_range[1][f] - m->finger_qpos_range[0][f]);
    obs[46] = (float)(ap / HRG_NFINGER);
    for (int a = 0; a < 3; a++) {
      obs[40 + a] = (float)(bx->obs_pos[a] - s->eef_pos[a]);
      obs[43 + a] = (float)(bx->target[a] - s->eef_pos[a]);
      obs[47 + a] = (float)bx->obs_pos[a];
      obs[50 + a] = (float)bx->target[a];
    }
    if (Re && (HRG_IS_HANDOVER(m->task) || m->task == HRG_TASK_LIFTING)) {
      /* quat_eef_to_object (human_robot_handover_cartesian_env.py:916-924, robot_human_handover_cartesian_env.py:998-1006) / quat_eef_to_board
       * (collaborative_lifting_cartesian_env.py:1057-1065), restated as the reference computes them: object_quat / board_quat and robot0_eef_quat are
       * (x, y, z, w) arrays, quat_to_rot reads them as (w, x, y, z) -- so the rotations that get multiplied are those of the scrambled quaternions
       * (scalar x, vector (y, z, w)).  Result = A conj(B) as (x, y, z, w); its overall sign is a convention (scipy keeps whatever the product gives). */
      double qe[4];
      mat2quat(qe, Re);
      const double sA = bx->quat[1], vA[3] = {bx->quat[2], bx->quat[3], bx->quat[0]}, sB = qe[1], vB[3] = {qe[2], qe[3], qe[0]};
      double cr[3];
      v3cross(cr, vA, vB);
      for (int a = 0; a < 3; a++) obs[57 + a] = (float)(-sA * vB[a] + sB * vA[a] - cr[a]);
      obs[60] = (float)(sA * sB + v3dot(vA, vB));
    }
    if (m->task == HRG_TASK_LIFTING) { /* collaborative_lifting_cartesian_env.py:982-1085: board_pos / vec_eef_to_board / board_gripped sit in the object columns;
                                        * board_balance takes the first target column, board_quat columns 43-45 and 51 (quat_eef_to_board: not served) */
      double Rx[9];
      quat2mat(Rx, bx->quat);
      for (int a = 0; a < 3; a++) { obs[43 + a] = (float)bx->quat[1 + a]; obs[50 + a] = 0.0f; } /* board_quat, (x, y, z, w) like T.convert_quat(..., to="xyzw") */
      obs[51] = (float)bx->quat[0];
      obs[50] = (float)Rx[8];
    }
  }
}

/* i-th object placement / target of an episode: UniformRandomSampler over the bins (pick_place_human_cartesian_env.py:
 * 613-635, 680-708, 843-875), drawn counter-based instead of as lists filled at reset */
static void placement_of(const hrgo_batch* B, int64_t gid, int episode, int idx, int target, double* p) {
  const hrg_model_desc* m = &B->m;
  const double* bin = target ? m->tgt_bin : m->obj_bin;
  uint64_t st = target ? STREAM_TARGET : STREAM_OBJECT;
  p[0] = bin[0] + (bin[1] - bin[0]) * rng_u01(m->seed, (uint64_t)gid, (uint64_t)episode, st, (uint64_t)(2 * idx));
  p[1] = bin[2] + (bin[3] - bin[2]) * rng_u01(m->seed, (uint64_t)gid, (uint64_t)episode, st, (uint64_t)(2 * idx + 1));
  p[2] = target ? m->tgt_z : m->obj_z;
}

/* HumanEnv._check_action_safety (human_env.py:931-946): does the arm at configuration q6 hit the static collision objects
 * (table volume, mount pedestal cylinder: _setup_collision_objects, human_env.py:1301-1348) or itself
 * (pinocchio_manipulator_model.py:168-236)?  Restated on the capsule model: links 0..6 and the gripper body (no fingers,
 * like the reference's URDF collision model). */
#define NCAP_CHECK 8
static int config_collides(const hrg_model_desc* m, const double* q6) {
  double q[NV], c1[NCAP_CHECK][3], c2[NCAP_CHECK][3], Rb[9];
  robot_kin k;
  for (int i = 0; i < NARM; i++) q[i] = q6[i];
  for (int i = NARM; i < NV; i++) q[i] = 0;
  robot_fk(m, q, &k);
  quat2mat(Rb, m->base_quat);
  double rad[NCAP_CHECK];
  for (int c = 0; c < NCAP_CHECK; c++) {
    int b = m->rcap_body[c];
    /* capsule 7 = the gripper cylinder of the reference's URDF collision model (r 0.07, l 0.11) = the shield's gripper capsule */
    const double* q1 = c == NCAP_CHECK - 1 ? m->scap_p1[HRG_NSHIELD_RCAP - 1] : m->rcap_p1[c];
    const double* q2 = c == NCAP_CHECK - 1 ? m->scap_p2[HRG_NSHIELD_RCAP - 1] : m->rcap_p2[c];
    rad[c] = c == NCAP_CHECK - 1 ? m->scap_r[HRG_NSHIELD_RCAP - 1] : m->rcap_r[c];
    double t[3];
    m3mulv(t, b < 0 ? Rb : k.R[b], q1); v3add(c1[c], b < 0 ? m->base_pos : k.p[b], t);
    m3mulv(t, b < 0 ? Rb : k.R[b], q2); v3add(c2[c], b < 0 ? m->base_pos : k.p[b], t);
  }
  const double mg = m->obstacle_margin;
  for (int c = 1; c < NCAP_CHECK; c++)
    for (int e = 0; e < 2; e++) {
      const double* p = e ? c2[c] : c1[c];
      const double r = rad[c];
      if (p[2] - r < m->table_top_z + mg && fabs(p[0] - m->table_center[0]) <= m->table_half[0] + 0.5 * mg + r && fabs(p[1] - m->table_center[1]) <= m->table_half[1] + 0.5 * mg + r) return 1 + 16 * c;
      const double dx = p[0] - m->base_pos[0], dy = p[1] - m->base_pos[1];
      if (p[2] - r < m->base_cyl_z && sqrt(dx * dx + dy * dy) < m->base_cyl_r + mg + r) return 2 + 16 * c;
    }
  for (int i = 0; i < NCAP_CHECK; i++)
    for (int j = i + 1; j < NCAP_CHECK; j++) {
      if (!((m->chk_selfmask[i] >> j) & 1u)) continue;
      double x1[3], x2[3];
      if (sqrt(seg_seg(c1[i], c2[i], c1[j], c2[j], x1, x2)) - rad[i] - rad[j] < m->self_collision_safety) return 3 + 16 * i + 256 * j;
    }
  return 0;
}

static void goal_of(const hrgo_batch* B, int64_t gid, const hrg_env_state* s, int idx, double* g) {
  /* ReachHuman._sample_valid_pos (reach_human_env.py:525-548): up to 20 uniform samples in the position limits, the first
   * collision-free one wins, else the zero configuration */
  const hrg_model_desc* m = &B->m;
  for (int t = 0; t < 20; t++) {
    for (int j = 0; j < NARM; j++) {
      double u = rng_u01(m->seed, (uint64_t)gid, (uint64_t)s->episode, STREAM_GOAL, (uint64_t)((idx * 20 + t) * NARM + j));
      g[j] = m->qpos_limits[0][j] + (m->qpos_limits[1][j] - m->qpos_limits[0][j]) * u;
    }
    if (!m->goal_check || !config_collides(m, g)) return;
    for (int j = 0; j < NARM; j++) g[j] = 0;
  }
}

/* HumanEnv.check_collision_action (human_env.py:588-627): goal = clip(q + scale(action)), then the pre-check */
static int action_collides(const hrg_model_desc* m, const hrg_env_state* s, const double* act) {
  double scale = fabs(m->act_out_max - m->act_out_min) / fabs(m->act_in_max - m->act_in_min);
  double otr = 0.5 * (m->act_out_max + m->act_out_min), itr = 0.5 * (m->act_in_max + m->act_in_min), g[NARM];
  for (int j = 0; j < NARM; j++) {
    double a = clampd(act[j], m->act_in_min, m->act_in_max);
    g[j] = clampd(s->qpos[j] + ((a - itr) * scale + otr), m->qpos_limits[0][j], m->qpos_limits[1][j]);
  }
  return config_collides(m, g);
}

/* IKPositionDeltaWrapper.step (wrappers/ik_position_delta_wrapper.py:93-142): position delta of the end-effector link with
 * its orientation held at the initial one -> joint delta.  pybullet.calculateInverseKinematics is [UPSTREAM]; restated as the
 * damped-least-squares iteration it documents (BussIK DLS): dq = J' (J J' + lambda^2 I)^-1 [e_pos; e_rot], at least one
 * step, then until the position residual is below the threshold or max_iter steps are done; steps are scaled down to
 * 45 deg per joint.  Kinematics = the stepper's own chain (same joint frames as robot_pybullet.urdf). */
static void ik_fk(const hrg_model_desc* m, const double* q6, double ax[NARM][3], double org[NARM][3], double* R6, double* pee) {
  double R[9], p[3], t[3];
  quat2mat(R, m->base_quat);
  v3cpy(p, m->base_pos);
  for (int i = 0; i < NARM; i++) {
    double Rq[9], Rl[9], Rj[9];
    quat2mat(Rq, m->body_quat[i]);
    m3mul(Rl, R, Rq);
    m3mulv(t, R, m->body_pos[i]);
    v3add(p, p, t);
    axisangle2mat(Rj, m->jnt_axis[i], q6[i]);
    m3mul(R, Rl, Rj);
    m3mulv(ax[i], R, m->jnt_axis[i]);
    v3cpy(org[i], p);
  }
  memcpy(R6, R, sizeof(double) * 9);
  m3mulv(t, R, m->ik_ee_offset);
  v3add(pee, p, t);
}
static void ik_action(const hrg_model_desc* m, const hrg_env_state* s, double* act) {
  double ws[3], q[NARM], ax[NARM][3], org[NARM][3], R6[9], pee[3], target[3];
  for (int a = 0; a < 3; a++) ws[a] = clampd(act[a], -m->ik_action_limit, m->ik_action_limit) * m->ik_x_output_max;
  const double grip = clampd(act[3], -1.0, 1.0);
  for (int j = 0; j < NARM; j++) q[j] = s->qpos[j];
  ik_fk(m, q, ax, org, R6, pee);
  for (int a = 0; a < 3; a++) {
    target[a] = pee[a] + ws[a];
    if (m->ik_use_pos_limits) target[a] = clampd(target[a], m->ik_pos_limits[0][a], m->ik_pos_limits[1][a]);
  }
  for (int it = 0;; it++) {
    if (it > 0) ik_fk(m, q, ax, org, R6, pee);
    double e[6], d[3];
    v3sub(d, target, pee);
    if (it > 0 && v3norm(d) <= m->ik_residual_threshold) break;
    if (it >= m->ik_max_iter) break;
    /* orientation error: rotation vector of R_target R6' */
    double E[9], v[3];
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) E[3 * a + b] = m->ik_target_rot[3 * a] * R6[3 * b] + m->ik_target_rot[3 * a + 1] * R6[3 * b + 1] + m->ik_target_rot[3 * a + 2] * R6[3 * b + 2];
    v3set(v, 0.5 * (E[7] - E[5]), 0.5 * (E[2] - E[6]), 0.5 * (E[3] - E[1]));
    double sn = v3norm(v), cs = 0.5 * (E[0] + E[4] + E[8] - 1.0), ang = atan2(sn, cs);
    for (int a = 0; a < 3; a++) { e[a] = d[a]; e[3 + a] = sn > 1e-12 ? v[a] / sn * ang : 0.0; }
    double J[6][NARM], A[36], y[6];
    for (int j = 0; j < NARM; j++) {
      double r[3], c[3];
      v3sub(r, pee, org[j]);
      v3cross(c, ax[j], r);
      for (int a = 0; a < 3; a++) { J[a][j] = c[a]; J[3 + a][j] = ax[j][a]; }
    }
    for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) {
      double t = 0;
      for (int j = 0; j < NARM; j++) t += J[a][j] * J[b][j];
      A[6 * a + b] = t + (a == b ? m->ik_damping * m->ik_damping : 0.0);
    }
    memcpy(y, e, sizeof y);
    if (!chol(A, 6)) break;
    chol_solve(A, 6, y);
    double dq[NARM], mx = 0;
    for (int j = 0; j < NARM; j++) { double t = 0; for (int a = 0; a < 6; a++) t += J[a][j] * y[a]; dq[j] = t; if (fabs(t) > mx) mx = fabs(t); }
    const double sc = mx > 0.25 * PI ? 0.25 * PI / mx : 1.0;
    for (int j = 0; j < NARM; j++) q[j] += sc * dq[j];
  }
  for (int j = 0; j < NARM; j++) act[j] = q[j] - s->qpos[j];
  act[NARM] = grip;
}

/* CollisionPreventionWrapper.action (wrappers/collision_prevention_wrapper.py:46-103) */
static void screen_action(const hrgo_batch* B, int64_t gid, hrg_env_state* s, double* act) {
  const hrg_model_desc* m = &B->m;
  if (!m->cp_enabled || !action_collides(m, s, act)) return;
  s->action_resamples++;
  double best[HRG_ACT_DIM], bestd = 1e300;
  int found = 0;
  if (m->cp_replace_type != 0)
    for (int t = 0; t < m->cp_n_resamples; t++) {
      double c[HRG_ACT_DIM], d = 0;
      for (int j = 0; j < HRG_ACT_DIM; j++) {
        c[j] = 2.0 * rng_u01(m->seed, (uint64_t)gid, (uint64_t)s->episode, STREAM_ACTION, (uint64_t)((s->timestep * 64 + t) * HRG_ACT_DIM + j)) - 1.0;
        d += (act[j] - c[j]) * (act[j] - c[j]);
      }
      if (action_collides(m, s, c)) continue;
      if (m->cp_replace_type == 1) { memcpy(best, c, sizeof best); found = 1; break; }
      if (d < bestd) { bestd = d; memcpy(best, c, sizeof best); found = 1; }
    }
  for (int j = 0; j < HRG_ACT_DIM; j++) act[j] = found ? best[j] : 0.0;
}

/* HumanRobotHandoverCart._update_mocap_body_transform (human_robot_handover_cartesian_env.py:609-633): the mocap body sits at the site of the
 * holding hand, rotated like the hand body turned by -+90 deg about its y axis */
static void handover_mocap(const hrgo_batch* B, int64_t gid, const hrg_env_state* s, hrg_box_state* bx, const human_kin* hk) {
  const hrg_model_desc* m = &B->m;
  const int left = B->clips.clip_holding_hand[clip_of(B, gid, s, s->anim_index)];
  const int site = left ? m->site_lhand : m->site_rhand, body = m->meas_body[site];
  const int r2h = m->task == HRG_TASK_HANDOVER_R2H; /* robot_human_handover_cartesian_env.py:615-648: opposite turn, offset towards the thumb */
  const double ang = (left != r2h) ? 0.5 * PI : -0.5 * PI, c = cos(ang), sn = sin(ang);
  const double Ry[9] = {c, 0, sn, 0, 1, 0, -sn, 0, c};
  double R[9];
  m3mul(R, hk->R[body], Ry);
  /* rotation matrix -> quaternion (w,x,y,z), w >= 0 branch-free enough for proper rotations: Shepperd's method */
  double q[4], tr = R[0] + R[4] + R[8];
  if (tr > 0) { double S = sqrt(tr + 1.0) * 2; q[0] = 0.25 * S; q[1] = (R[7] - R[5]) / S; q[2] = (R[2] - R[6]) / S; q[3] = (R[3] - R[1]) / S; }
  else if (R[0] > R[4] && R[0] > R[8]) { double S = sqrt(1.0 + R[0] - R[4] - R[8]) * 2; q[0] = (R[7] - R[5]) / S; q[1] = 0.25 * S; q[2] = (R[1] + R[3]) / S; q[3] = (R[2] + R[6]) / S; }
  else if (R[4] > R[8]) { double S = sqrt(1.0 + R[4] - R[0] - R[8]) * 2; q[0] = (R[2] - R[6]) / S; q[1] = (R[1] + R[3]) / S; q[2] = 0.25 * S; q[3] = (R[5] + R[7]) / S; }
  else { double S = sqrt(1.0 + R[8] - R[0] - R[4]) * 2; q[0] = (R[3] - R[1]) / S; q[1] = (R[2] + R[6]) / S; q[2] = (R[5] + R[7]) / S; q[3] = 0.25 * S; }
  for (int a = 0; a < 4; a++) bx->mocap_quat[a] = q[a];
  v3cpy(bx->mocap_pos, s->human_site[site]);
  if (r2h) {
    const double off[3] = {left ? 0.02 : -0.02, -0.03, -0.03};
    double t[3];
    m3mulv(t, R, off);
    v3add(bx->mocap_pos, bx->mocap_pos, t);
    v3cpy(bx->target, bx->mocap_pos); /* target_pos property: the hand the object has to reach (448-450) */
  }
}
/* _human_pickup_object (700-711): the object jumps into the hand and the weld is switched on */
static void handover_pickup(hrg_box_state* bx) {
  v3cpy(bx->pos, bx->mocap_pos);
  for (int a = 0; a < 4; a++) bx->quat[a] = bx->mocap_quat[a];
  v3set(bx->weld_off, 0, 0, 0);
  bx->weld_rel[0] = 1; bx->weld_rel[1] = bx->weld_rel[2] = bx->weld_rel[3] = 0;
  bx->weld_active = 1;
}

/* _human_pickup_object of RobotHumanHandoverCart (730-748): the weld takes the pose the object has relative to the hand mocap body at that
 * moment (the reference re-poses a grip sub-body at the palm contact; same rigid attachment, expressed at the object's origin) */
static void handover_attach(hrg_box_state* bx) {
  double Rm[9], d[3], qc[4] = {bx->mocap_quat[0], -bx->mocap_quat[1], -bx->mocap_quat[2], -bx->mocap_quat[3]};
  quat2mat(Rm, bx->mocap_quat);
  v3sub(d, bx->pos, bx->mocap_pos);
  for (int a = 0; a < 3; a++) bx->weld_off[a] = Rm[a] * d[0] + Rm[3 + a] * d[1] + Rm[6 + a] * d[2]; /* Rm' d */
  quatmul(bx->weld_rel, qc, bx->quat);
  bx->weld_active = 1;
}
/* contact between the cube and the palm (= the collision capsule of the holding hand's body): _get_object_palm_contact_pos (476-505) */
static int palm_contact(const hrgo_batch* B, int64_t gid, const hrg_env_state* s, const hrg_box_state* bx, const human_kin* hk) {
  const hrg_model_desc* m = &B->m;
  const int left = B->clips.clip_holding_hand[clip_of(B, gid, s, s->anim_index)];
  const int body = m->meas_body[left ? m->site_lhand : m->site_rhand];
  double Rx[9], t, cs[3], cb[3];
  quat2mat(Rx, bx->quat);
  return sqrt(seg_box(hk->cap1[body], hk->cap2[body], bx->pos, Rx, m->box_half, &t, cs, cb)) - m->hcap_r[body] < 0;
}

/* rotation matrix (row-major) -> unit quaternion (w, x, y, z), largest-component branch */
static void mat2quat(double* q, const double* R) {
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0) { double s = sqrt(tr + 1.0) * 2; q[0] = 0.25 * s; q[1] = (R[7] - R[5]) / s; q[2] = (R[2] - R[6]) / s; q[3] = (R[3] - R[1]) / s; }
  else if (R[0] > R[4] && R[0] > R[8]) { double s = sqrt(1.0 + R[0] - R[4] - R[8]) * 2; q[0] = (R[7] - R[5]) / s; q[1] = 0.25 * s; q[2] = (R[1] + R[3]) / s; q[3] = (R[2] + R[6]) / s; }
  else if (R[4] > R[8]) { double s = sqrt(1.0 + R[4] - R[0] - R[8]) * 2; q[0] = (R[2] - R[6]) / s; q[1] = (R[1] + R[3]) / s; q[2] = 0.25 * s; q[3] = (R[5] + R[7]) / s; }
  else { double s = sqrt(1.0 + R[8] - R[0] - R[4]) * 2; q[0] = (R[3] - R[1]) / s; q[1] = (R[2] + R[6]) / s; q[2] = (R[5] + R[7]) / s; q[3] = 0.25 * s; }
}
/* CollaborativeLiftingCart._reset_animation (collaborative_lifting_cartesian_env.py:644-657) puts the board at a fixed pose that the Schunk
 * gripper's init_qpos (673) straddles; with the stand-in gripper the pose follows from the gripper frame instead: board x axis = from the
 * board centre back to the robot (-gripper axis), board normal = the finger closing axis turned upwards, robot-side edge lift_grip_depth past
 * the grip site.  At rest, no warm start. */
static void lifting_place_board(const hrg_model_desc* m, const robot_kin* k, const double* eef, hrg_box_state* bx) {
  const double* Re = k->R[NARM]; /* a finger body's frame = the hand frame (right_hand is turned -45 deg about link 6's axis, robot.xml:61): y = closing axis, z = gripper axis */
  double ze[3] = {Re[2], Re[5], Re[8]}, ye[3] = {Re[1], Re[4], Re[7]}, xb[3], yb[3], zb[3], Rb[9];
  const double sg = ye[2] >= 0 ? 1.0 : -1.0;
  for (int a = 0; a < 3; a++) { xb[a] = -ze[a]; zb[a] = sg * ye[a]; }
  v3cross(yb, zb, xb);
  for (int a = 0; a < 3; a++) { Rb[3 * a] = xb[a]; Rb[3 * a + 1] = yb[a]; Rb[3 * a + 2] = zb[a]; }
  mat2quat(bx->quat, Rb);
  for (int a = 0; a < 3; a++) bx->pos[a] = eef[a] + ze[a] * (m->box_half[0] - m->lift_grip_depth);
  for (int a = 0; a < HRG_NBOXV; a++) { bx->vel[a] = 0; bx->acc_warmstart[a] = 0; }
  v3cpy(bx->obs_pos, bx->pos);
}
/* _update_mocap_body_transforms (590-616): the two mocap bodies sit at the hand sites (their orientation does not enter a connect equality) */
static void lifting_mocap(const hrg_model_desc* m, const hrg_env_state* s, hrg_box_state* bx) {
  v3cpy(bx->mocap_pos, s->human_site[m->site_lhand]);
  v3cpy(bx->weld_off, s->human_site[m->site_rhand]);
}

static void eef_of(const hrg_model_desc* m, const robot_kin* k, double* eef) {
  double t[3];
  m3mulv(t, k->R[NARM - 1], m->eef_pos);
  v3add(eef, k->p[NARM - 1], t);
}

static void env_reset_stack(hrgo_batch* B, int e, const robot_kin* k);
static void compute_obs_stack(const hrgo_batch* B, int64_t gid, const hrg_env_state* s, hrg_stack_state* sk, float* obs);
static void env_step_stack(hrgo_batch* B, int e, double* action, float* obs, float* term_obs, float* reward, uint8_t* done, int32_t* info);
static void env_reset_hammer(hrgo_batch* B, int e, const robot_kin* k);
static void compute_obs_hammer(const hrgo_batch* B, const hrg_env_state* s, const hrg_hammer_state* hm, float* obs);
static void env_step_hammer(hrgo_batch* B, int e, double* action, float* obs, float* term_obs, float* reward, uint8_t* done, int32_t* info);
static void env_reset(hrgo_batch* B, int e, float* obs) {
  const hrg_model_desc* m = &B->m;
  hrg_env_state* s = &B->st[e];
  int64_t gid = B->env_id0 + e;
  int episode = s->episode + 1;
  memset(s, 0, sizeof *s);
  s->episode = episode;
  s->stream_id = (int32_t)gid;
  /* robot.reset: init_qpos + N(0, 0.02^2) (robosuite "default" initialization_noise) */
  for (int j = 0; j < NARM; j++) s->qpos[j] = m->init_qpos[j] + m->init_noise * rng_gauss(m->seed, (uint64_t)gid, (uint64_t)episode, STREAM_NOISE, (uint64_t)j);
  for (int j = 0; j < HRG_NFINGER; j++) s->qpos[NARM + j] = m->finger_init_qpos[j];
  if (m->task == HRG_TASK_HAMMERING) { /* _put_hammer_into_gripper (790-812) closes the gripper on the handle: the fingers start where their pads touch it, commanded shut */
    for (int j = 0; j < HRG_NFINGER; j++) s->qpos[NARM + j] = m->hm_finger_grip_qpos[j];
    s->grip_action = -1.0;
  }
  /* human placement: x, y, yaw uniform in +-human_rand (human_env.py:1376-1387, 1650-1656) */
  double ux = rng_u01(m->seed, (uint64_t)gid, (uint64_t)episode, STREAM_HUMAN, 0), uy = rng_u01(m->seed, (uint64_t)gid, (uint64_t)episode, STREAM_HUMAN, 1),
         uz = rng_u01(m->seed, (uint64_t)gid, (uint64_t)episode, STREAM_HUMAN, 2);
  s->human_pos_offset[0] = m->base_human_pos_offset[0] + (2 * ux - 1) * m->human_rand[0];
  s->human_pos_offset[1] = m->base_human_pos_offset[1] + (2 * uy - 1) * m->human_rand[1];
  s->human_pos_offset[2] = m->base_human_pos_offset[2];
  double yaw = (2 * uz - 1) * m->human_rand[2];
  s->human_rot_offset[0] = cos(0.5 * yaw); s->human_rot_offset[1] = 0; s->human_rot_offset[2] = 0; s->human_rot_offset[3] = sin(0.5 * yaw);
  s->animation_time = -1; /* human_env.py:1644 */
  /* human pose before the first _control_human: qpos0 at the mocap body's default pose (origin, identity) */
  {
    human_kin h;
    double zero[HRG_NHQ], p0[3] = {0, 0, 0}, q0[4] = {1, 0, 0, 0};
    memset(zero, 0, sizeof zero);
    human_fk(m, p0, q0, zero, &h, s->human_site);
  }
  robot_kin k;
  robot_fk(m, s->qpos, &k);
  eef_of(m, &k, s->eef_pos);
  shield_reset(m, s, s->qpos); /* FailsafeController.reset, failsafe_controller.py:204-250 */
  for (int j = 0; j < NARM; j++) s->goal_qpos[j] = s->qpos[j];
  if (m->task == HRG_TASK_STACKING) {
    env_reset_stack(B, e, &k);
    if (obs) compute_obs_stack(B, gid, s, &B->stk[e], obs);
    return;
  }
  if (m->task == HRG_TASK_HAMMERING) {
    env_reset_hammer(B, e, &k);
    if (obs) compute_obs_hammer(B, s, &B->hmr[e], obs);
    return;
  }
  hrg_box_state* bx = m->task != HRG_TASK_REACH ? &B->box[e] : NULL;
  if (bx) { /* PickPlaceHumanCart._reset_internal: first object placement and target, object at rest */
    memset(bx, 0, sizeof *bx);
    placement_of(B, gid, episode, 0, 0, bx->pos);
    placement_of(B, gid, episode, 0, 1, bx->target);
    if (m->task == HRG_TASK_INSPECTION) { /* the target comes with the animation (info json), not from a bin */
      const int clip = clip_of(B, gid, s, 0);
      for (int a = 0; a < 3; a++) bx->target[a] = B->clips.clip_target_pos[clip][a] + s->human_pos_offset[a];
    }
    bx->quat[0] = 1;
    if (m->task == HRG_TASK_HANDOVER_H2R) { /* _reset_animation + _control_human (human_robot_handover_cartesian_env.py:635-647, 686-711): the human starts
                                             * with the object welded into the holding hand.  The reference teleports it to the stale mocap pose first
                                             * and lets the weld drag it over; here the hand pose of the first animation frame is used directly */
      human_kin hk;
      double mp[3], mq[4];
      const double* qh;
      human_control(B, gid, s, bx, mp, mq, &qh);
      human_fk(m, mp, mq, qh, &hk, s->human_site);
      handover_mocap(B, gid, s, bx, &hk);
      handover_pickup(bx);
    }
    if (m->task == HRG_TASK_LIFTING) { /* _reset_internal (670-681): _control_human + _reset_animation; the human holds the board from the start (reset's
                                        * deterministic branch, 696-700: the grasp-and-retry loop of 702-735 is not run) */
      human_kin hk;
      double mp[3], mq[4];
      const double* qh;
      human_control(B, gid, s, bx, mp, mq, &qh);
      human_fk(m, mp, mq, qh, &hk, s->human_site);
      lifting_mocap(m, s, bx);
      lifting_place_board(m, &k, s->eef_pos, bx);
      bx->weld_active = 1;
      v3cpy(bx->target, bx->pos);
    }
    if (m->task == HRG_TASK_HANDOVER_R2H) { /* _reset_animation (the human holds nothing) + _control_human (650-660, 700-706): object in its bin, hand = target */
      human_kin hk;
      double mp[3], mq[4];
      const double* qh;
      human_control(B, gid, s, bx, mp, mq, &qh);
      human_fk(m, mp, mq, qh, &hk, s->human_site);
      handover_mocap(B, gid, s, bx, &hk);
    }
    if (m->task != HRG_TASK_LIFTING) v3cpy(bx->obs_pos, bx->pos);
    if (m->task == HRG_TASK_REACH_BOX) goal_of(B, gid, s, 0, s->cur_goal);
  } else goal_of(B, gid, s, 0, s->cur_goal);
  if (obs) compute_obs_e(m, s, bx, s->cur_goal, k.R[NARM], obs);
}

static void env_step(hrgo_batch* B, int e, double* action, float* obs, float* term_obs, float* reward, uint8_t* done, int32_t* info) {
  const hrg_model_desc* m = &B->m;
  if (m->task == HRG_TASK_STACKING) { env_step_stack(B, e, action, obs, term_obs, reward, done, info); return; }
  if (m->task == HRG_TASK_HAMMERING) { env_step_hammer(B, e, action, obs, term_obs, reward, done, info); return; }
  hrg_env_state* s = &B->st[e];
  int64_t gid = s->stream_id; /* in-episode draws follow the state's streams (= the env's own id unless the state was copied in) */
  const double h = m->timestep;
  if (m->ik_enabled) ik_action(m, s, action); /* IKPositionDeltaWrapper is the outermost action wrapper (utils/training_utils.py:358-373) */
  screen_action(B, gid, s, action); /* CollisionPreventionWrapper.step wraps env.step: uses the pre-step state */
  if (m->task == HRG_TASK_LIFTING) action[NARM] = 1; /* CollaborativeLiftingCart.step (368-391): the gripper action is replaced by 'close' */
  s->timestep += 1; /* human_env.py:490 */
  int has_collision = 0, collision_type = HRG_COL_NULL, failsafe_intervention = 0, crash = 0;
  robot_kin k;
  human_kin hk;
  double M[NV * NV], bias[NV];
  hrg_box_state* bx = m->task != HRG_TASK_REACH ? &B->box[e] : NULL;
  const int nvt = bx ? NVT : NV, ncon_dyn = bx ? HRG_NCON_DYN_BOX : HRG_NCON_DYN;
  int palm_hit = 0;
  for (int cyc = 0; cyc < m->n_cycles && !crash; cyc++) {
    /* ---- sim.forward() #1 (human_env.py:504): positions, M, bias at the current state ---- */
    robot_fk(m, s->qpos, &k);
    robot_crba(m, &k, M);
    robot_bias(m, &k, s->qvel, bias);
    /* ---- controller (SingleArm.control -> set_goal / run_controller) ---- */
    if (cyc == 0) { /* policy step: failsafe_controller.py:252-300 */
      for (int i = 0; i < NARM; i++) for (int j = 0; j < NARM; j++) s->mass_matrix[i * NARM + j] = M[i * NV + j];
      double scale = fabs(m->act_out_max - m->act_out_min) / fabs(m->act_in_max - m->act_in_min);
      double otr = 0.5 * (m->act_out_max + m->act_out_min), itr = 0.5 * (m->act_in_max + m->act_in_min);
      for (int j = 0; j < NARM; j++) {
        double a = clampd(action[j], m->act_in_min, m->act_in_max);
        double g = s->qpos[j] + ((a - itr) * scale + otr);
        s->goal_qpos[j] = clampd(g, m->qpos_limits[0][j], m->qpos_limits[1][j]);
        s->new_goal_q[j] = s->goal_qpos[j];
      }
      s->new_goal = 1; /* newLongTermTrajectory, failsafe_controller.py:300 */
    }
    shield_step(B, e, s->time); /* humanMeasurement + SafetyShield.step, human_env.py:505, failsafe_controller.py:329 */
    double ctrl[NV];
    for (int i = 0; i < NARM; i++) { /* failsafe_controller.py:356-369 */
      double t = 0;
      for (int j = 0; j < NARM; j++) t += s->mass_matrix[i * NARM + j] * (m->kp * (s->des_q[j] - s->qpos[j]) + m->kd * (s->des_v[j] - s->qvel[j]) + s->des_a[j]);
      ctrl[i] = clampd(t + bias[i], m->arm_ctrlrange[i][0], m->arm_ctrlrange[i][1]);
    }
    { /* RethinkGripper.format_action + actuator ctrl range mapping; +1 closes, -1 opens (experts/pick_place_human_cart_expert.py:282-288), finger 0 opens towards positive qpos (rethink_valid_gripper.py:25-42) */
      double a = action[NARM], sg = a > 0 ? 1.0 : (a < 0 ? -1.0 : 0.0);
      s->grip_action = clampd(s->grip_action - m->gripper_speed * sg, -1.0, 1.0);
      for (int f = 0; f < HRG_NFINGER; f++) {
        double lo = m->finger_ctrlrange[f][0], hi = m->finger_ctrlrange[f][1];
        ctrl[NARM + f] = 0.5 * (hi + lo) + 0.5 * (hi - lo) * (f == 0 ? s->grip_action : -s->grip_action);
      }
    }
    if (!failsafe_intervention && !s->is_safe) { failsafe_intervention = 1; s->failsafe_interventions++; } /* human_env.py:509-513 */
    /* ---- _control_human + sim.forward() #2 (human_env.py:516-519) ---- */
    double mp[3], mq[4];
    const double* qh;
    human_control(B, gid, s, bx, mp, mq, &qh);
    human_fk(m, mp, mq, qh, &hk, s->human_site);
    if (m->task == HRG_TASK_LIFTING) lifting_mocap(m, s, bx); /* CollaborativeLiftingCart._control_human (583-588) */
    /* HumanRobotHandoverCart._control_human (human_robot_handover_cartesian_env.py:598-633) runs one more sim.step() with the new human
     * pose before it re-poses the hand mocap body: pass 0 = that step (no bookkeeping), pass 1 = the cycle's regular step */
    for (int pass = HRG_IS_HANDOVER(m->task) ? 0 : 1; pass < 2 && !crash; pass++) {
    /* ---- contacts + bookkeeping (human_env.py:522) ---- */
    contact_t con[HRG_NCON_MAX];
    int ncon = collide(m, &k, &hk, bx, con);
    if (bx) { /* _check_grasp [UPSTREAM robosuite]: both fingers touch the object (pick_place_human_cartesian_env.py:804-809) */
      int f0 = 0, f1 = 0;
      for (int c = 0; c < ncon; c++) if (con[c].g2 == GEOM_BOX) { f0 |= con[c].g1 == HRG_NRCAP - 2; f1 |= con[c].g1 == HRG_NRCAP - 1; }
      bx->gripped = f0 && f1;
    }
    double rc[HRG_NRCAP][3], Rb[9];
    quat2mat(Rb, m->base_quat);
    for (int c = 0; c < HRG_NRCAP; c++) {
      int b = m->rcap_body[c];
      double t[3], mid[3];
      for (int a = 0; a < 3; a++) mid[a] = 0.5 * (m->rcap_p1[c][a] + m->rcap_p2[c][a]);
      m3mulv(t, b < 0 ? Rb : k.R[b], mid);
      v3add(rc[c], b < 0 ? m->base_pos : k.p[b], t);
    }
    if (pass == 1) {
      if (m->task == HRG_TASK_HANDOVER_R2H) palm_hit = palm_contact(B, gid, s, bx, &hk); /* sim.data.contact of the cycle's collision phase */
      classify(m, &k, s, con, ncon, rc, &has_collision, &collision_type);
      s->ncon = ncon;
      for (int c = 0; c < HRG_NCON_MAX; c++) { s->con_pairs[c][0] = c < ncon ? con[c].g1 : -1; s->con_pairs[c][1] = c < ncon ? con[c].g2 : -1; }
    }
    /* ---- sim.step() (human_env.py:523): smooth acceleration, constraints, Euler ---- */
    double LM[NV * NV], a0[NVT], frc[NV], qd[NVT], Mt[NVT * NVT];
    memcpy(LM, M, sizeof LM);
    if (!chol(LM, NV)) { crash = 1; break; }
    for (int i = 0; i < NV; i++) {
      double act = ctrl[i];
      if (i >= NARM) act = clampd(m->finger_kp * (ctrl[i] - s->qpos[i]), m->finger_forcerange[0], m->finger_forcerange[1]);
      frc[i] = act - m->jnt_damping[i] * s->qvel[i] - bias[i];
      a0[i] = frc[i];
      qd[i] = s->qvel[i];
    }
    chol_solve(LM, NV, a0);
    if (bx) { /* free box, world-frame angular velocity: M = blockdiag(m 1, R diag(I) R'), bias torque w x (R diag(I) R' w), gravity.
               * The rotational inertia is split into mean * 1 + R diag(I - mean) R': a cube keeps an exactly diagonal M and no gyroscopic term */
      double Rx[9], Mdev[9], w[3] = {bx->vel[3], bx->vel[4], bx->vel[5]}, Ld[3], Lw[3], tau[3], tl[3];
      quat2mat(Rx, bx->quat);
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double t = 0;
        for (int kk = 0; kk < 3; kk++) t += Rx[3 * i + kk] * (m->box_inertia[kk] - m->box_inertia_mean) * Rx[3 * j + kk];
        Mdev[3 * i + j] = t;
      }
      for (int kk = 0; kk < 3; kk++) Ld[kk] = (m->box_inertia[kk] - m->box_inertia_mean) * (Rx[kk] * w[0] + Rx[3 + kk] * w[1] + Rx[6 + kk] * w[2]);
      m3mulv(Lw, Rx, Ld); /* R diag(I - mean) R' w */
      v3cross(tau, Lw, w); /* -(w x L) */
      for (int kk = 0; kk < 3; kk++) tl[kk] = (Rx[kk] * tau[0] + Rx[3 + kk] * tau[1] + Rx[6 + kk] * tau[2]) / m->box_inertia[kk]; /* diag(1/I) R' tau */
      memset(Mt, 0, sizeof Mt);
      for (int i = 0; i < NV; i++) for (int j = 0; j < NV; j++) Mt[i * NVT + j] = M[i * NV + j];
      for (int a = 0; a < 3; a++) {
        Mt[(NV + a) * NVT + NV + a] = m->box_mass;
        for (int b = 0; b < 3; b++) Mt[(NV + 3 + a) * NVT + NV + 3 + b] = (a == b ? m->box_inertia_mean : 0.0) + Mdev[3 * a + b];
        a0[NV + a] = m->gravity[a];
        a0[NV + 3 + a] = Rx[3 * a] * tl[0] + Rx[3 * a + 1] * tl[1] + Rx[3 * a + 2] * tl[2];
      }
      for (int a = 0; a < HRG_NBOXV; a++) qd[NV + a] = bx->vel[a];
    } else memcpy(Mt, M, sizeof M);
    efc_t E;
    E.n = 0;
    E.nv = nvt;
    for (int i = 0; i < NV; i++) /* friction loss rows */
      if (m->jnt_frictionloss[i] > 0) { double J[NVT] = {0}; J[i] = 1; efc_add(m, &E, J, qd, ROW_FRICTION, 0, 0, m->jnt_frictionloss[i], m->dof_invweight0[i]); }
    for (int i = 0; i < NV; i++) { /* joint limit rows */
      double dlo = s->qpos[i] - m->jnt_range[i][0], dhi = m->jnt_range[i][1] - s->qpos[i];
      if (dlo < 0) { double J[NVT] = {0}; J[i] = 1; efc_add(m, &E, J, qd, ROW_UNILATERAL, dlo, 0, 0, m->dof_invweight0[i]); }
      if (dhi < 0) { double J[NVT] = {0}; J[i] = -1; efc_add(m, &E, J, qd, ROW_UNILATERAL, dhi, 0, 0, m->dof_invweight0[i]); }
    }
    for (int c = 0; c < ncon && c < ncon_dyn; c++) { /* pyramidal frictional contact rows */
      const double* n = con[c].n;
      double t1[3], t2[3], e1[3] = {1, 0, 0}, e2[3] = {0, 1, 0};
      v3cross(t1, n, fabs(n[0]) < 0.5 ? e1 : e2);
      v3scl(t1, t1, 1.0 / v3norm(t1));
      v3cross(t2, n, t1);
      double margin = (con[c].g2 >= GEOM_HUMAN0 && con[c].g2 < GEOM_TABLE) || (con[c].g1 >= GEOM_HUMAN0 && con[c].g1 < GEOM_TABLE) ? m->contact_margin_human : 0.0; /* a human geom on either side */
      for (int d = 0; d < 4; d++) {
        double dir[3], J[NVT] = {0};
        const double* tt = d < 2 ? t1 : t2;
        double sg = (d & 1) ? -1.0 : 1.0;
        for (int a = 0; a < 3; a++) dir[a] = n[a] + sg * m->friction_static * tt[a];
        /* separation velocity along n (from geom1 to geom2): v2 - v1 */
        if (con[c].b1 >= 0 && con[c].b1 < NV) robot_point_jac(m, &k, con[c].b1, con[c].pos, dir, -1.0, J);
        if (con[c].b2 >= 0 && con[c].b2 < NV) robot_point_jac(m, &k, con[c].b2, con[c].pos, dir, +1.0, J);
        double diag = (con[c].b1 >= 0 && con[c].b1 < NV ? m->body_invweight0[con[c].b1] : 0.0) + (con[c].b2 >= 0 && con[c].b2 < NV ? m->body_invweight0[con[c].b2] : 0.0);
        if (con[c].b2 == BODY_BOX) { /* the cube is always geom 2: J = dir . (v + w x r), body_invweight0 of a free body = 1/m */
          double r[3], rxd[3];
          v3sub(r, con[c].pos, bx->pos);
          v3cross(rxd, r, dir);
          for (int a = 0; a < 3; a++) { J[NV + a] = dir[a]; J[NV + 3 + a] = rxd[a]; }
          diag += 1.0 / m->box_mass;
        }
        efc_add(m, &E, J, qd, ROW_UNILATERAL, con[c].dist, margin, 0, diag * (1.0 + m->friction_static * m->friction_static));
      }
    }
    if (bx && bx->weld_active && m->task == HRG_TASK_LIFTING) { /* two connect equalities (collaborative_lifting_cartesian_env.py:924-958): the board's grip points follow the
                                                                  * hand mocap bodies; residual = p_board + R anchor - p_mocap, velocity of the point = v + w x r */
      double Rx[9];
      quat2mat(Rx, bx->quat);
      for (int hd = 0; hd < 2; hd++) {
        double rr[3], pt[3];
        const double* mp_ = hd ? bx->weld_off : bx->mocap_pos;
        m3mulv(rr, Rx, m->lift_anchor[hd]);
        v3add(pt, bx->pos, rr);
        for (int a = 0; a < 3; a++) {
          double J[NVT] = {0}, ea[3] = {a == 0, a == 1, a == 2}, rxe[3];
          v3cross(rxe, rr, ea); /* (w x r) . e_a = w . (r x e_a) */
          J[NV + a] = 1;
          for (int b_ = 0; b_ < 3; b_++) J[NV + 3 + b_] = rxe[b_];
          efc_add(m, &E, J, qd, ROW_EQUALITY, pt[a] - mp_[a], 0, 0, 1.0 / m->box_mass);
        }
      }
    } else if (bx && bx->weld_active) { /* weld of the object frame onto the hand mocap frame (human_robot_handover_cartesian_env.py:870-903): residual =
                                  * [p_obj - p_mocap; rotation vector of q_obj q_mocap^-1]; the mocap body has no velocity; relpose = identity */
      double epos[3], erot[3], qt[4], qe[4], Rm[9], tp[3];
      quat2mat(Rm, bx->mocap_quat);
      m3mulv(tp, Rm, bx->weld_off);
      v3add(tp, tp, bx->mocap_pos);          /* where the weld wants the object: mocap frame o relative pose */
      v3sub(epos, bx->pos, tp);
      quatmul(qt, bx->mocap_quat, bx->weld_rel);
      double qc[4] = {qt[0], -qt[1], -qt[2], -qt[3]};
      quatmul(qe, bx->quat, qc);
      if (qe[0] < 0) for (int a = 0; a < 4; a++) qe[a] = -qe[a];
      const double sn = sqrt(qe[1] * qe[1] + qe[2] * qe[2] + qe[3] * qe[3]), ang = 2.0 * atan2(sn, qe[0]);
      for (int a = 0; a < 3; a++) erot[a] = sn > 1e-12 ? qe[1 + a] / sn * ang : 0.0;
      for (int a = 0; a < 3; a++) { double J[NVT] = {0}; J[NV + a] = 1; efc_add(m, &E, J, qd, ROW_EQUALITY, epos[a], 0, 0, 1.0 / m->box_mass); }
      for (int a = 0; a < 3; a++) { double J[NVT] = {0}; J[NV + 3 + a] = 1; efc_add(m, &E, J, qd, ROW_EQUALITY, erot[a], 0, 0, m->box_invweight_rot); }
    }
    double qacc[NVT];
    memcpy(qacc, s->qacc_warmstart, sizeof(double) * NV);
    if (bx) memcpy(qacc + NV, bx->acc_warmstart, sizeof(double) * HRG_NBOXV);
    solve(m, Mt, a0, &E, qacc);
    if (g_debug && (ncon > 0 || g_debug > 1)) {
      double mx = 0; for (int i = 0; i < nvt; i++) if (fabs(qacc[i]) > mx) mx = fabs(qacc[i]);
      fprintf(stderr, "[oracle] env %d cyc %d ncon %d nefc %d max|qacc| %.3e", e, cyc, ncon, E.n, mx);
      for (int c = 0; c < ncon; c++) fprintf(stderr, " (%d,%d d=%.4f)", con[c].g1, con[c].g2, con[c].dist);
      fprintf(stderr, "\n");
    }
    /* mj_checkAcc -> MujocoException handler (human_env.py:527-546) */
    for (int i = 0; i < nvt; i++) if (!(fabs(qacc[i]) < 1e10)) crash = 1;
    if (crash) break;
    memcpy(s->qacc_warmstart, qacc, sizeof(double) * NV);
    /* mj_Euler with implicit joint damping: (M + h D) qacc' = M qacc */
    double Mh[NV * NV], rhs[NV];
    memcpy(Mh, M, sizeof Mh);
    for (int i = 0; i < NV; i++) { Mh[i * NV + i] += h * m->jnt_damping[i]; double t = 0; for (int j = 0; j < NV; j++) t += M[i * NV + j] * qacc[j]; rhs[i] = t; }
    if (!chol(Mh, NV)) { crash = 1; break; }
    chol_solve(Mh, NV, rhs);
    for (int i = 0; i < NV; i++) { s->qvel[i] += h * rhs[i]; s->qpos[i] += h * s->qvel[i]; }
    if (bx) { /* free joint: no damping; quaternion integrated with the world-frame angular velocity */
      memcpy(bx->acc_warmstart, qacc + NV, sizeof(double) * HRG_NBOXV);
      v3cpy(bx->obs_pos, bx->pos); /* body_xpos of the forward pass inside mj_step (pre-integration) */
      for (int a = 0; a < HRG_NBOXV; a++) bx->vel[a] += h * qacc[NV + a];
      for (int a = 0; a < 3; a++) bx->pos[a] += h * bx->vel[a];
      double w[3] = {bx->vel[3], bx->vel[4], bx->vel[5]}, wn = v3norm(w), ang = h * wn;
      if (wn > 1e-12) {
        double sh = sin(0.5 * ang) / wn, dq[4] = {cos(0.5 * ang), w[0] * sh, w[1] * sh, w[2] * sh}, qn[4];
        quatmul(qn, dq, bx->quat);
        double nn = sqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
        for (int a = 0; a < 4; a++) bx->quat[a] = qn[a] / nn;
      }
    }
    s->time += h;
    eef_of(m, &k, s->eef_pos); /* site_xpos of the forward pass inside mj_step (pre-integration) */
    if (pass == 0) { /* _update_mocap_body_transform (609-633) + sim.forward() (human_env.py:519) */
      handover_mocap(B, gid, s, bx, &hk);
      robot_fk(m, s->qpos, &k);
      robot_crba(m, &k, M);
      robot_bias(m, &k, s->qvel, bias);
    }
    } /* pass */
    if (crash) break;
    s->low_level_time += 1; /* human_env.py:526 */
  }
  /* ---- observation, success, info, reward, done (human_env.py:561-581) ---- */
  double goal[NARM];
  memcpy(goal, s->cur_goal, sizeof goal);
  if (bx && m->task == HRG_TASK_POINTING) { /* target_pos property: the elbow -> hand ray extended to the table (pick_place_pointing_human_cartesian_env.py:336-360) */
    const int left = B->clips.clip_pointing_hand[clip_of(B, gid, s, s->anim_index)];
    const double* hand = s->human_site[left ? m->site_lhand : m->site_rhand];
    const double* elbow = s->human_site[left ? m->site_lelbow : m->site_relbow];
    double dir[3];
    v3sub(dir, hand, elbow);
    if (dir[2] == 0) dir[2] += 1e-6;
    const double scaling = (hand[2] - m->table_top_z) / dir[2];
    for (int a = 0; a < 3; a++) bx->target[a] = hand[a] - scaling * dir[a];
  }
  compute_obs_e(m, s, bx, goal, k.R[NARM], term_obs);
  double dist2 = 0, dense;
  int goal_reached;
  double r;
  if (bx && m->task != HRG_TASK_REACH_BOX) { /* PickPlaceHumanCart: achieved goal = [eef_pos, object_pos, object_gripped], desired goal = target_pos (574-611) */
    double e2o = 0, o2t = 0;
    for (int a = 0; a < 3; a++) { e2o += (bx->obs_pos[a] - s->eef_pos[a]) * (bx->obs_pos[a] - s->eef_pos[a]); o2t += (bx->target[a] - bx->obs_pos[a]) * (bx->target[a] - bx->obs_pos[a]); }
    const int in_zone = sqrt(o2t) <= m->goal_dist; /* _check_object_in_target_zone, 550-572 */
    if (m->task == HRG_TASK_INSPECTION || m->task == HRG_TASK_HANDOVER_H2R) { /* success = the animation ran to its end; human_object_inspection_cartesian_env.py:553-600, human_robot_handover_cartesian_env.py:485-528 */
      goal_reached = !crash && bx->task_phase == HRG_PHASE_COMPLETE;
      r = goal_reached ? m->task_reward : (in_zone ? m->object_at_target_reward : (bx->gripped ? m->object_gripped_reward : -1.0));
    } else if (m->task == HRG_TASK_LIFTING) { /* collaborative_lifting_cartesian_env.py:429-478: success = the animation ran to its end; base reward +1 */
      double Rx[9];
      quat2mat(Rx, bx->quat);
      const double balance = Rx[8]; /* board normal . world up (board_balance, 1033-1041) */
      goal_reached = !crash && bx->task_phase == HRG_PHASE_COMPLETE;
      r = goal_reached ? m->task_reward : (balance < m->min_balance ? m->imbalance_failure_reward : (!bx->gripped ? m->board_released_reward : 1.0));
    } else if (m->task == HRG_TASK_HANDOVER_R2H) { /* robot_human_handover_cartesian_env.py:507-555 */
      goal_reached = !crash && bx->task_phase == HRG_R2H_COMPLETE;
      r = goal_reached ? m->task_reward : (bx->task_phase == HRG_R2H_RETREAT ? m->object_in_human_hand_reward : (bx->gripped ? m->object_gripped_reward : -1.0));
    } else {
      goal_reached = !crash && in_zone;
      r = goal_reached ? m->task_reward : (bx->gripped ? m->object_gripped_reward : -1.0); /* _sparse_reward, 471-500 */
    }
    dense = -(sqrt(e2o) * 0.2 + sqrt(o2t)) * 0.1; /* _dense_reward, 502-526 */
    if (m->task == HRG_TASK_LIFTING) { /* _dense_reward (480-507): balance angle normalised between arcsin(min_balance) and 1, minus 2 */
      double Rx[9];
      quat2mat(Rx, bx->quat);
      const double ba = asin(clampd(Rx[8], -1.0, 1.0)) * 2 / PI, mba = asin(m->min_balance) * 2 / PI;
      dense = (ba - mba) / (1 - mba) - 2.0;
    }
  } else {
    for (int j = 0; j < NARM; j++) dist2 += (s->qpos[j] - goal[j]) * (s->qpos[j] - goal[j]);
    double dist = sqrt(dist2);
    goal_reached = !crash && dist <= m->goal_dist; /* reach_human_env.py:457-475 */
    r = goal_reached ? m->task_reward : -1.0; /* human_env.py:666-691 */
    dense = -0.1 * dist; /* reach_human_env.py:437-455 */
  }
  if (goal_reached) s->n_goal_reached++;
  int illegal = (collision_type & (HRG_COL_STATIC | HRG_COL_ROBOT | HRG_COL_HUMAN_CRIT)) != 0; /* human_env.py:860-878 */
  if (m->reward_shaping) r += 1.0 + dense; /* human_env.py:650-651 */
  if (illegal) r += m->collision_reward;
  r *= m->reward_scale;
  int d = 0;
  if (crash) { r += m->sim_crash_reward; d = 1; }
  else {
    if (m->done_at_collision && illegal) d = 1; /* human_env.py:835-858 */
    if (m->done_at_success && goal_reached) d = 1;
    if (m->task == HRG_TASK_LIFTING) { /* _check_done (509-561): unbalanced, or the board out of the gripper for more than 5 steps in a row */
      double Rx[9];
      quat2mat(Rx, bx->quat);
      bx->n_delayed = bx->gripped ? 0 : bx->n_delayed + 1;
      if (Rx[8] < m->min_balance || bx->n_delayed > 5) d = 1;
    }
  }
  int ncoll = s->n_collisions_static + s->n_collisions_robot + s->n_collisions_human + s->n_collisions_critical;
  info[HRG_INFO_COLLISION] = has_collision;
  info[HRG_INFO_COLLISION_TYPE] = collision_type;
  info[HRG_INFO_N_COLLISIONS] = ncoll;
  info[HRG_INFO_N_COLLISIONS_STATIC] = s->n_collisions_static;
  info[HRG_INFO_N_COLLISIONS_ROBOT] = s->n_collisions_robot;
  info[HRG_INFO_N_COLLISIONS_HUMAN] = s->n_collisions_human;
  info[HRG_INFO_N_COLLISIONS_CRITICAL] = s->n_collisions_critical;
  info[HRG_INFO_TIMEOUT] = s->timestep >= m->horizon;
  info[HRG_INFO_FAILSAFE_INTERVENTIONS] = s->failsafe_interventions;
  info[HRG_INFO_N_GOAL_REACHED] = s->n_goal_reached;
  info[HRG_INFO_SIM_CRASH] = crash;
  info[HRG_INFO_TRUNCATED] = 0;
  info[HRG_INFO_ACTION_RESAMPLES] = s->action_resamples;
  info[HRG_INFO_N_OBJECT_HANDED_OVER] = bx ? bx->n_handed_over : 0;
  if (bx && m->task == HRG_TASK_LIFTING) {
    if (goal_reached && !m->done_at_success) { /* _on_goal_reached (631-642): the robot back at its initial posture (deterministic), controller reset, next
                                                * animation, _control_human, board back in the gripper */
      for (int j = 0; j < NARM; j++) { s->qpos[j] = m->init_qpos[j]; s->qvel[j] = 0; s->qacc_warmstart[j] = 0; }
      for (int j = 0; j < HRG_NFINGER; j++) { s->qpos[NARM + j] = m->finger_init_qpos[j]; s->qvel[NARM + j] = 0; s->qacc_warmstart[NARM + j] = 0; }
      s->grip_action = 0;
      robot_fk(m, s->qpos, &k);
      eef_of(m, &k, s->eef_pos);
      shield_reset(m, s, s->qpos);
      for (int j = 0; j < NARM; j++) s->goal_qpos[j] = s->qpos[j];
      s->anim_index = (s->anim_index + 1) % m->n_anim_ids;
      s->animation_time = 0;
      s->anim_start_time = (int)((double)s->low_level_time / m->anim_step_length);
      bx->task_phase = HRG_PHASE_APPROACH; bx->n_delayed = 0;
      human_kin hk2;
      double mp[3], mq[4];
      const double* qh;
      human_control(B, gid, s, bx, mp, mq, &qh);
      human_fk(m, mp, mq, qh, &hk2, s->human_site);
      lifting_mocap(m, s, bx);
      lifting_place_board(m, &k, s->eef_pos, bx);
    }
  } else if (bx && m->task == HRG_TASK_HANDOVER_R2H) {
    if (goal_reached && !m->done_at_success) { /* _on_goal_reached (662-676): next placement, next animation, the human lets go */
      bx->obj_index = (bx->obj_index + 1) % m->n_obj_placements;
      placement_of(B, gid, s->episode, bx->obj_index, 0, bx->pos);
      bx->quat[0] = 1; bx->quat[1] = bx->quat[2] = bx->quat[3] = 0;
      s->anim_index = (s->anim_index + 1) % m->n_anim_ids;
      s->animation_time = 0;
      s->anim_start_time = (int)((double)s->low_level_time / m->anim_step_length);
      bx->task_phase = HRG_R2H_APPROACH; bx->n_delayed = 0; bx->weld_active = 0;
      human_kin hk2;
      double mp[3], mq[4];
      const double* qh;
      human_control(B, gid, s, bx, mp, mq, &qh);
      human_fk(m, mp, mq, qh, &hk2, s->human_site);
      handover_mocap(B, gid, s, bx, &hk2);
      hk = hk2;
    }
    /* RobotHumanHandoverCart.step (452-474): the human takes the object when it touches the palm of the extended hand */
    if (bx->task_phase == HRG_R2H_REACH_OUT && palm_hit) { handover_attach(bx); bx->task_phase = HRG_R2H_RETREAT; bx->n_handed_over++; }
  } else if (bx && m->task == HRG_TASK_HANDOVER_H2R) {
    double o2t = 0;
    for (int a = 0; a < 3; a++) o2t += (bx->target[a] - bx->obs_pos[a]) * (bx->target[a] - bx->obs_pos[a]);
    if (goal_reached && !m->done_at_success) { /* _on_goal_reached (649-668): next target, next animation, the human picks the object up again */
      bx->tgt_index = (bx->tgt_index + 1) % m->n_targets;
      placement_of(B, gid, s->episode, bx->tgt_index, 1, bx->target);
      s->anim_index = (s->anim_index + 1) % m->n_anim_ids;
      s->animation_time = 0;
      s->anim_start_time = (int)((double)s->low_level_time / m->anim_step_length);
      bx->task_phase = HRG_PHASE_APPROACH; bx->n_delayed = 0; bx->n_delayed2 = 0;
      human_kin hk2;
      double mp[3], mq[4];
      const double* qh;
      human_control(B, gid, s, bx, mp, mq, &qh);
      human_fk(m, mp, mq, qh, &hk2, s->human_site);
      handover_mocap(B, gid, s, bx, &hk2);
      handover_pickup(bx);
    }
    /* HumanRobotHandoverCart.step (465-483): the human lets go once the robot has gripped the object; the retreat starts when it is placed */
    if (bx->task_phase == HRG_PHASE_PRESENT && bx->gripped) { bx->weld_active = 0; bx->task_phase = HRG_PHASE_WAIT; bx->n_handed_over++; }
    else if (bx->task_phase == HRG_PHASE_WAIT && sqrt(o2t) <= m->goal_dist) bx->task_phase = HRG_PHASE_RETREAT;
  } else if (bx && m->task == HRG_TASK_INSPECTION) {
    double o2t = 0;
    for (int a = 0; a < 3; a++) o2t += (bx->target[a] - bx->obs_pos[a]) * (bx->target[a] - bx->obs_pos[a]);
    if (goal_reached && !m->done_at_success) { /* _on_goal_reached, human_object_inspection_cartesian_env.py:492-505: next placement, next animation */
      bx->obj_index = (bx->obj_index + 1) % m->n_obj_placements;
      placement_of(B, gid, s->episode, bx->obj_index, 0, bx->pos);
      bx->quat[0] = 1; bx->quat[1] = bx->quat[2] = bx->quat[3] = 0;
      s->anim_index = (s->anim_index + 1) % m->n_anim_ids; /* _progress_to_next_animation, human_env.py:1698-1708 + 663-670 */
      s->animation_time = 0;
      s->anim_start_time = (int)((double)s->low_level_time / m->anim_step_length);
      bx->task_phase = HRG_PHASE_APPROACH;
      bx->n_delayed = 0;
    }
    /* HumanObjectInspectionCart.step, 461-490: the inspection starts when the object enters the target zone and is interrupted
     * when it leaves it by more than goal_exit_tolerance */
    if (bx->task_phase == HRG_PHASE_READY && sqrt(o2t) <= m->goal_dist) bx->task_phase = HRG_PHASE_INSPECTION;
    else if (bx->task_phase == HRG_PHASE_INSPECTION && !(sqrt(o2t) - m->goal_exit_tolerance <= m->goal_dist)) bx->task_phase = HRG_PHASE_READY;
  } else if (goal_reached && bx && m->task != HRG_TASK_REACH_BOX) { /* _on_goal_reached, pick_place_human_cartesian_env.py:440-453: next target, object teleported to its next placement (velocity kept) */
    bx->tgt_index = (bx->tgt_index + 1) % m->n_targets;
    bx->obj_index = (bx->obj_index + 1) % m->n_obj_placements;
    placement_of(B, gid, s->episode, bx->tgt_index, 1, bx->target);
    placement_of(B, gid, s->episode, bx->obj_index, 0, bx->pos);
    bx->quat[0] = 1; bx->quat[1] = bx->quat[2] = bx->quat[3] = 0;
  } else if (goal_reached) { /* reach_human_env.py:399-407 */
    s->goal_index = (s->goal_index + 1) % m->n_goals;
    goal_of(B, gid, s, s->goal_index, s->cur_goal);
  }
  if (s->timestep >= m->horizon) { info[HRG_INFO_TRUNCATED] = !d; d = 1; } /* time_limit.py:40-43 */
  *reward = (float)r;
  *done = (uint8_t)d;
  if (d) env_reset(B, e, obs); /* VecEnv auto-reset; terminal observation stays in term_obs */
  else memcpy(obs, term_obs, sizeof(float) * HRG_OBS_DIM);
}

/* =============================================================================================== CollaborativeStackingCart
 * collaborative_stacking_cartesian_env.py: four cubes with free joints (manipulation_object_a / _b for the robot, human_l_cube / human_r_cube welded to
 * mocap bodies at the human's hands), box-box contacts between them, the seven-phase animation state machine, stack bookkeeping, rewards.
 * Geometry / contact stand-ins as for the single cube; cube-cube contacts by separating-axis test + reference-face clipping (D13). */
#define GEOM_CUBE(c) (GEOM_BOX + (c))
#define BODY_CUBE(c) (BODY_BOX + (c))

/* CollaborativeStackingCart._compute_animation_time (825-897), applied to the classic animation time *at_io; wave-uniform bookkeeping in sk */
static void stack_animation_time(const hrgo_batch* b, int64_t gid, const hrg_env_state* s, hrg_stack_state* sk, int clip, int* at_io) {
  const int classic = *at_io, len = b->clips.clip_len[clip];
  const int32_t* kf = b->clips.clip_stack_keyframes[clip];
  int at = classic;
  if (sk->task_phase == HRG_STK_APPROACH && at > kf[0]) sk->task_phase = HRG_STK_PLACE_FIRST;
  else if (sk->task_phase == HRG_STK_WAIT_FOR_SECOND) { /* loop until the robot has placed its cube */
    if (at >= kf[2]) at = (int)layered_sines(b, gid, s, clip, 0, b->clips.clip_n_loop[clip], (double)classic, (double)kf[2]);
    sk->n_delayed[0] = classic - at; sk->n_delayed[1] = 0;
  } else if (sk->task_phase == HRG_STK_PLACE_THIRD) at = classic - sk->n_delayed[0];
  else if (sk->task_phase == HRG_STK_WAIT_FOR_FOURTH) {
    at = classic - sk->n_delayed[0];
    if (at >= kf[4]) at = (int)layered_sines(b, gid, s, clip, HRG_MAX_LOOP, b->clips.clip_n_loop2[clip], (double)at, (double)kf[4]);
    sk->n_delayed[1] = classic - at;
  } else if (sk->task_phase == HRG_STK_RETREAT) at = classic - sk->n_delayed[1];
  if (at >= len - 1) { sk->task_phase = HRG_STK_COMPLETE; at = len - 1; }
  if (at < 0) at = 0;
  *at_io = at;
}

/* Contacts of two boxes with half extents ha / hb (centres pa / pb, rotations Ra / Rb row-major; the stacking task's cubes share one h): separating-axis test over the 15 axes; the axis of
 * least penetration decides.  A face axis: the face of the other box most anti-parallel to it is clipped against the reference face's rectangle --
 * candidates = incident vertices inside the rectangle (0..3), rectangle corners under the incident face (4..7), crossings of the incident edges with the
 * rectangle's sides (8..23); of those that penetrate, at most four are kept: the deepest, the one farthest from it, and the farthest from their line on
 * either side (a resting face keeps a quadrilateral that spans its support polygon).  An edge-edge axis (only when clearly less penetrating, factor 1.05): one contact between the closest points of the two
 * edges.  Normal from box a to box b.  Stand-in for mjc_BoxBox [UPSTREAM]. */
/* near-ties between separating axes / candidate depths are broken towards the earlier one unless the later wins by this margin (1 nm): two nearly
 * parallel cubes keep their reference face while rounding-level differences come and go */
#define BB_TIE 1e-9
static int box_box2(const double* pa, const double* Ra, const double* ha, const double* pb, const double* Rb, const double* hb, bb_contact out[4]) {
  double A[3][3], B[3][3], C[3][3], AC[3][3], t[3], ta[3], tb[3];
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
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    const double l2 = 1.0 - C[i][j] * C[i][j];
    if (l2 < 1e-12) continue; /* parallel edges: the face axes cover it */
    const double l = sqrt(l2);
    const double tl = ta[i2] * C[i1][j] - ta[i1] * C[i2][j]; /* t . (A_i x B_j) */
    const double s_ = (fabs(tl) - (ha[i1] * AC[i2][j] + ha[i2] * AC[i1][j] + hb[j1] * AC[i][j2] + hb[j2] * AC[i][j1])) / l;
    if (s_ > 0) return 0;
    if (s_ > se + BB_TIE) { se = s_; be = 3 * i + j; }
  }
  if (be >= 0 && se * 1.05 > sf) { /* edge - edge */
    const int i = be / 3, j = be % 3;
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
  /* face: reference box / incident box */
  const int refA = bf < 3, r = refA ? bf : bf - 3, r1 = (r + 1) % 3, r2 = (r + 2) % 3;
  const double (*Rf)[3] = refA ? A : B;
  const double (*In)[3] = refA ? B : A;
  const double* pr = refA ? pa : pb;
  const double* pi = refA ? pb : pa;
  const double* hr = refA ? ha : hb;   /* half extents of the reference / the incident box */
  const double* hi = refA ? hb : ha;
  const double sg = refA ? (ta[r] >= 0 ? 1.0 : -1.0) : (tb[r] >= 0 ? -1.0 : 1.0); /* reference normal points at the incident box */
  double nr[3], cr[3], ci[3];
  v3scl(nr, Rf[r], sg);
  v3madd(cr, pr, nr, hr[r]);
  int k = 0;
  double best = -1;
  for (int q = 0; q < 3; q++) { const double c_ = fabs(v3dot(In[q], nr)); if (c_ > best) { best = c_; k = q; } }
  const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
  const double si = v3dot(In[k], nr) > 0 ? -1.0 : 1.0; /* incident face normal opposes the reference normal */
  v3madd(ci, pi, In[k], si * hi[k]);
  const double hu = hr[r1], hv = hr[r2];
  static const double S1[4] = {1, -1, -1, 1}, S2[4] = {1, 1, -1, -1};
  double vu[4], vv[4], vd[4];
  for (int q = 0; q < 4; q++) {
    double x[3], d[3];
    v3madd(x, ci, In[k1], S1[q] * hi[k1]);
    v3madd(x, x, In[k2], S2[q] * hi[k2]);
    v3sub(d, x, cr);
    vu[q] = v3dot(d, Rf[r1]); vv[q] = v3dot(d, Rf[r2]); vd[q] = v3dot(d, nr);
  }
  double cu[24], cv[24], cd[24];
  int ok[24];
  for (int q = 0; q < 24; q++) ok[q] = 0;
  for (int q = 0; q < 4; q++) /* incident vertices over the reference rectangle */
    if (fabs(vu[q]) <= hu && fabs(vv[q]) <= hv) { ok[q] = 1; cu[q] = vu[q]; cv[q] = vv[q]; cd[q] = vd[q]; }
  { /* rectangle corners under the incident face: the face projects onto the reference plane as the parallelogram c0 + al e1 + be e2, |al|, |be| <= 1 */
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
        if (fabs(al) <= 1 && fabs(be_) <= 1) { ok[4 + q] = 1; cu[4 + q] = S1[q] * hu; cv[4 + q] = S2[q] * hv; cd[4 + q] = c0d + al * e1d + be_ * e2d; }
      }
  }
  for (int q = 0; q < 4; q++) { /* incident edge q -> q + 1 against the four sides of the rectangle */
    const int q1 = (q + 1) & 3;
    const double du = vu[q1] - vu[q], dv = vv[q1] - vv[q], dd = vd[q1] - vd[q];
    for (int e = 0; e < 4; e++) {
      const int c_ = 8 + 4 * q + e;
      const double lim = (e & 1) ? -1.0 : 1.0;
      if (e < 2) { /* u = +-hu */
        if (fabs(du) < 1e-14) continue;
        const double tt = (lim * hu - vu[q]) / du, w = vv[q] + tt * dv;
        if (tt > 0 && tt < 1 && fabs(w) < hv) { ok[c_] = 1; cu[c_] = lim * hu; cv[c_] = w; cd[c_] = vd[q] + tt * dd; }
      } else { /* v = +-hv */
        if (fabs(dv) < 1e-14) continue;
        const double tt = (lim * hv - vv[q]) / dv, w = vu[q] + tt * du;
        if (tt > 0 && tt < 1 && fabs(w) < hu) { ok[c_] = 1; cu[c_] = w; cv[c_] = lim * hv; cd[c_] = vd[q] + tt * dd; }
      }
    }
  }
  /* of the penetrating candidates keep at most four that span the contact patch: the deepest, the one farthest from it, then the one farthest from
   * their line on either side (ties -> the lowest candidate index; a candidate closer than 1e-6 of an edge to what is already kept adds nothing) */
  int pick[4], np_ = 0;
  const double eps2 = 1e-12 * (hu * hu + hv * hv);
  for (int c_ = 0; c_ < 24; c_++) if (ok[c_] && !(cd[c_] < 0)) ok[c_] = 0;
  {
    int arg = -1;
    for (int c_ = 0; c_ < 24; c_++) if (ok[c_] && (arg < 0 || cd[c_] < cd[arg] - BB_TIE)) arg = c_;
    if (arg >= 0) pick[np_++] = arg;
  }
  if (np_ == 1) {
    int arg = -1;
    double bestv = eps2;
    for (int c_ = 0; c_ < 24; c_++) {
      if (!ok[c_]) continue;
      const double du = cu[c_] - cu[pick[0]], dv = cv[c_] - cv[pick[0]], val = du * du + dv * dv;
      if (val > bestv * (1 + 1e-9)) { arg = c_; bestv = val; }
    }
    if (arg >= 0) pick[np_++] = arg;
  }
  if (np_ == 2) {
    const double lu = cu[pick[1]] - cu[pick[0]], lv = cv[pick[1]] - cv[pick[0]], epsc = sqrt(eps2 * (lu * lu + lv * lv));
    int argp = -1, argn = -1;
    double bp = epsc, bn = epsc;
    for (int c_ = 0; c_ < 24; c_++) {
      if (!ok[c_]) continue;
      const double cr_ = lu * (cv[c_] - cv[pick[0]]) - lv * (cu[c_] - cu[pick[0]]);
      if (cr_ > bp * (1 + 1e-9)) { argp = c_; bp = cr_; }
      if (-cr_ > bn * (1 + 1e-9)) { argn = c_; bn = -cr_; }
    }
    if (argp >= 0) pick[np_++] = argp;
    if (argn >= 0) pick[np_++] = argn;
  }
  for (int z = 0; z < np_; z++) {
    const int c_ = pick[z];
    for (int a = 0; a < 3; a++) {
      out[z].pos[a] = cr[a] + cu[c_] * Rf[r1][a] + cv[c_] * Rf[r2][a] + 0.5 * cd[c_] * nr[a];
      out[z].n[a] = refA ? nr[a] : -nr[a];
    }
    out[z].dist = cd[c_];
  }
  return np_;
}

static int box_box(const double* pa, const double* Ra, const double* pb, const double* Rb, const double* h, bb_contact out[4]) { return box_box2(pa, Ra, h, pb, Rb, h, out); }

/* contact list of the stacking task: the robot's own contacts (collide), then per cube c the robot capsules' first points, table corners, floor corners,
 * then the cube pairs (a < b), then the second points of capsules lying along a face.  object_gripped (1467-1481): a finger pair holds cube A or B --
 * the reference's `root_body not in _object_stack_body_ids` compares a name with ids and never excludes a stacked cube; restated as it behaves. */
static int collide_stack(const hrg_model_desc* m, const robot_kin* k, const human_kin* h, hrg_stack_state* sk, contact_t* con) {
  int n = collide(m, k, h, NULL, con);
  double rp1[HRG_NRCAP][3], rp2[HRG_NRCAP][3], Rb[9], Rx[NCUBE][9];
  quat2mat(Rb, m->base_quat);
  for (int c = 0; c < HRG_NRCAP; c++) {
    int b = m->rcap_body[c];
    const double* R = b < 0 ? Rb : k->R[b];
    const double* p = b < 0 ? m->base_pos : k->p[b];
    double t[3];
    m3mulv(t, R, m->rcap_p1[c]); v3add(rp1[c], p, t);
    m3mulv(t, R, m->rcap_p2[c]); v3add(rp2[c], p, t);
  }
  for (int c = 0; c < NCUBE; c++) quat2mat(Rx[c], sk->quat[c]);
  const double* hb = m->box_half;
#define EMIT(G1, G2, B1, B2, DIST, NRM, POS) \
  do { if (n < HRG_NCON_MAX) { con[n].g1 = G1; con[n].g2 = G2; con[n].b1 = B1; con[n].b2 = B2; con[n].dist = DIST; v3cpy(con[n].n, NRM); v3cpy(con[n].pos, POS); n++; } } while (0)
  int n_second = 0, second_i[NCUBE * HRG_NRCAP], second_c[NCUBE * HRG_NRCAP];
  double second_s[NCUBE * HRG_NRCAP][3], second_b[NCUBE * HRG_NRCAP][3];
  int f0[NCUBE] = {0}, f1[NCUBE] = {0};
  for (int c = 0; c < NCUBE; c++)
    for (int i = 0; i < HRG_NRCAP; i++) {
      if (m->rcap_body[i] < 0) continue;
      double t, cs[3], cb[3], nn[3], pos[3], s2[2][3], b2[2][3];
      double e2 = seg_box(rp1[i], rp2[i], sk->pos[c], Rx[c], hb, &t, cs, cb), dd = sqrt(e2), dist = dd - m->rcap_r[i];
      if (!(dist < 0)) continue;
      if (dd > 1e-9 && cap_box_two(rp1[i], rp2[i], sk->pos[c], Rx[c], hb, m->rcap_r[i], cs, cb, s2, b2)) {
        v3cpy(cs, s2[0]); v3cpy(cb, b2[0]);
        v3sub(nn, cb, cs); dd = v3norm(nn); dist = dd - m->rcap_r[i];
        second_i[n_second] = i; second_c[n_second] = c; v3cpy(second_s[n_second], s2[1]); v3cpy(second_b[n_second], b2[1]); n_second++;
      }
      if (dd > 1e-9) { v3sub(nn, cb, cs); v3scl(nn, nn, 1.0 / dd); }
      else { /* capsule axis inside the cube: push out through the nearest face */
        double loc[3], best = 1e300; int ax = 0;
        v3sub(pos, cs, sk->pos[c]);
        for (int a = 0; a < 3; a++) { loc[a] = Rx[c][a] * pos[0] + Rx[c][3 + a] * pos[1] + Rx[c][6 + a] * pos[2]; if (hb[a] - fabs(loc[a]) < best) { best = hb[a] - fabs(loc[a]); ax = a; } }
        double sg = loc[ax] >= 0 ? -1.0 : 1.0;
        for (int a = 0; a < 3; a++) nn[a] = sg * Rx[c][3 * a + ax];
        dist = -best - m->rcap_r[i];
      }
      v3madd(pos, cs, nn, m->rcap_r[i] + 0.5 * dist);
      if (n < HRG_NCON_MAX) { f0[c] |= i == HRG_NRCAP - 2; f1[c] |= i == HRG_NRCAP - 1; } /* contacts beyond the reported list do not count */
      EMIT(i, GEOM_CUBE(c), m->rcap_body[i], BODY_CUBE(c), dist, nn, pos);
    }
  for (int pl = 0; pl < 2; pl++)
    for (int c = 0; c < NCUBE; c++)
      for (int cn = 0; cn < 8; cn++) {
        double loc[3] = {(cn & 1) ? hb[0] : -hb[0], (cn & 2) ? hb[1] : -hb[1], (cn & 4) ? hb[2] : -hb[2]}, p[3];
        m3mulv(p, Rx[c], loc);
        v3add(p, p, sk->pos[c]);
        double z0 = pl ? m->floor_z : m->table_top_z, dist = p[2] - z0;
        if (pl == 0 && !(fabs(p[0] - m->table_center[0]) <= m->table_half[0] && fabs(p[1] - m->table_center[1]) <= m->table_half[1] && p[2] > z0 - 0.05)) continue;
        if (dist < 0) {
          double nn[3] = {0, 0, 1}, pos[3] = {p[0], p[1], z0 + 0.5 * dist};
          EMIT(pl ? GEOM_FLOOR : GEOM_TABLE, GEOM_CUBE(c), -1, BODY_CUBE(c), dist, nn, pos);
        }
      }
  for (int a = 0; a < NCUBE; a++)
    for (int b_ = a + 1; b_ < NCUBE; b_++) {
      double d[3];
      v3sub(d, sk->pos[b_], sk->pos[a]);
      const double reach2 = 4.0 * (hb[0] * hb[0] + hb[1] * hb[1] + hb[2] * hb[2]);
      if (v3dot(d, d) > reach2) continue; /* circumspheres apart */
      bb_contact bc[4];
      const int nc = box_box(sk->pos[a], Rx[a], sk->pos[b_], Rx[b_], hb, bc);
      for (int q = 0; q < nc; q++) EMIT(GEOM_CUBE(a), GEOM_CUBE(b_), BODY_CUBE(a), BODY_CUBE(b_), bc[q].dist, bc[q].n, bc[q].pos);
    }
  for (int q = 0; q < n_second; q++) {
    const int i = second_i[q], c = second_c[q];
    double nn[3], pos[3];
    v3sub(nn, second_b[q], second_s[q]);
    const double dd = v3norm(nn), dist = dd - m->rcap_r[i];
    v3scl(nn, nn, 1.0 / dd);
    v3madd(pos, second_s[q], nn, m->rcap_r[i] + 0.5 * dist);
    EMIT(i, GEOM_CUBE(c), m->rcap_body[i], BODY_CUBE(c), dist, nn, pos);
  }
#undef EMIT
  sk->gripped = (f0[HRG_CUBE_A] && f1[HRG_CUBE_A]) || (f0[HRG_CUBE_B] && f1[HRG_CUBE_B]);
  return n;
}

/* _update_mocap_body_transforms (905-936): each mocap body sits at its hand site, 3 cm towards the thumb (hand z axis), rotated like the hand body
 * turned by -90 deg (left) / +90 deg (right) about its y axis */
static void stack_mocap(const hrg_model_desc* m, const hrg_env_state* s, hrg_stack_state* sk, const human_kin* hk) {
  for (int hd = 0; hd < 2; hd++) {
    const int site = hd == 0 ? m->site_lhand : m->site_rhand, body = m->meas_body[site];
    const double ang = hd == 0 ? -0.5 * PI : 0.5 * PI, c = cos(ang), sn = sin(ang);
    const double Ry[9] = {c, 0, sn, 0, 1, 0, -sn, 0, c}, ez[3] = {0, 0, 1};
    double R[9], t[3];
    m3mulv(t, hk->R[body], ez);
    for (int a = 0; a < 3; a++) sk->mocap_pos[hd][a] = s->human_site[site][a] + 0.03 * t[a];
    m3mul(R, hk->R[body], Ry);
    mat2quat(sk->mocap_quat[hd], R);
  }
}
/* where the weld of hand hd wants its cube: mocap = cube o relpose  =>  cube = mocap o relpose^-1 (relpose = (stack_weld_relpos, identity)) */
static void stack_weld_target(const hrg_model_desc* m, const hrg_stack_state* sk, int hd, double* pos) {
  double Rm[9], t[3];
  quat2mat(Rm, sk->mocap_quat[hd]);
  m3mulv(t, Rm, m->stack_weld_relpos);
  v3sub(pos, sk->mocap_pos[hd], t);
}
/* _human_pickup_objects (1053-1056): both welds on.  The reference leaves the cubes where they are and lets the soft welds drag them into the hands;
 * here they are put there directly, at rest (as for the handover object) */
static void stack_pickup(const hrg_model_desc* m, hrg_stack_state* sk) {
  for (int hd = 0; hd < 2; hd++) {
    const int c = HRG_CUBE_L + hd;
    stack_weld_target(m, sk, hd, sk->pos[c]);
    for (int a = 0; a < 4; a++) sk->quat[c][a] = sk->mocap_quat[hd][a];
    for (int a = 0; a < 6; a++) { sk->vel[c][a] = 0; sk->acc_warmstart[c][a] = 0; }
    v3cpy(sk->obs_pos[c], sk->pos[c]);
    sk->weld_active[hd] = 1;
  }
}
/* idx-th placement of robot cube c (A or B): UniformRandomSampler over the bin, rotation (0, 0), no overlap test (1188-1194; human_env.py:1221-1270) */
static void stack_placement(const hrgo_batch* B, int64_t gid, int episode, int idx, int c, double* p) {
  const hrg_model_desc* m = &B->m;
  const double u0 = rng_u01(m->seed, (uint64_t)gid, (uint64_t)episode, STREAM_OBJECT, (uint64_t)(4 * idx + 2 * c)), u1 = rng_u01(m->seed, (uint64_t)gid, (uint64_t)episode, STREAM_OBJECT, (uint64_t)(4 * idx + 2 * c + 1));
  p[0] = m->obj_bin[0] + (m->obj_bin[1] - m->obj_bin[0]) * u0;
  p[1] = m->obj_bin[2] + (m->obj_bin[3] - m->obj_bin[2]) * u1;
  p[2] = m->obj_z;
}
/* _reset_animation (991-998) */
static void stack_reset_animation(const hrg_model_desc* m, hrg_stack_state* sk) {
  sk->task_phase = HRG_STK_APPROACH;
  sk->n_delayed[0] = sk->n_delayed[1] = 0;
  sk->n_stack = 0;
  for (int a = 0; a < NCUBE; a++) sk->stack_ids[a] = -1;
  sk->max_stack_height = 0;
  stack_pickup(m, sk);
}
/* next_target_position (522-548): one cube height above the cube the human placed last; returns 0 when it is not the robot's turn */
static int stack_next_target(const hrgo_batch* B, int64_t gid, const hrg_env_state* s, const hrg_stack_state* sk, double* tgt) {
  const hrg_model_desc* m = &B->m;
  const int left = B->clips.clip_holding_hand[clip_of(B, gid, s, s->anim_index)]; /* first_placing_hand */
  int c = -1;
  if (sk->task_phase == HRG_STK_WAIT_FOR_SECOND) c = left ? HRG_CUBE_L : HRG_CUBE_R;
  else if (sk->task_phase == HRG_STK_WAIT_FOR_FOURTH) c = left ? HRG_CUBE_R : HRG_CUBE_L;
  if (c < 0) return 0;
  v3cpy(tgt, sk->obs_pos[c]);
  tgt[2] += 2.0 * m->box_half[2];
  return 1;
}
/* _id_of_cube_at_target (590-610): a robot cube within goal_dist (maximum norm, 612-618) of the target that is not part of the stack; -1 = none */
static int stack_cube_at_target(const hrgo_batch* B, int64_t gid, const hrg_env_state* s, const hrg_stack_state* sk) {
  double tgt[3];
  if (!stack_next_target(B, gid, s, sk, tgt)) return -1;
  for (int c = HRG_CUBE_A; c <= HRG_CUBE_B; c++) {
    double dmax = 0;
    for (int a = 0; a < 3; a++) { const double d = fabs(tgt[a] - sk->obs_pos[c][a]); if (d > dmax) dmax = d; }
    int in_stack = 0;
    for (int q = 0; q < sk->n_stack; q++) if (sk->stack_ids[q] == c) in_stack = 1;
    if (dmax < B->m.goal_dist && !in_stack) return c;
  }
  return -1;
}
/* _check_first / _check_second_manipulation_object_in_target_zone (620-654) */
static int stack_first_in_zone(const hrgo_batch* B, int64_t gid, const hrg_env_state* s, const hrg_stack_state* sk) {
  if (sk->n_stack < 1) return -1;
  if (sk->n_stack > 1) return sk->stack_ids[1];
  return stack_cube_at_target(B, gid, s, sk);
}
static int stack_second_in_zone(const hrgo_batch* B, int64_t gid, const hrg_env_state* s, const hrg_stack_state* sk) {
  if (sk->n_stack < 3) return -1;
  if (sk->n_stack > 3) return sk->stack_ids[3];
  return stack_cube_at_target(B, gid, s, sk);
}
/* _check_stack_toppled (656-671): the centre of a stacked cube below the top of the bottom one */
static int stack_toppled(const hrg_model_desc* m, const hrg_stack_state* sk) {
  if (sk->n_stack < 2) return 0;
  const double min_h = sk->obs_pos[sk->stack_ids[0]][2] + m->box_half[2];
  for (int q = 1; q < sk->n_stack; q++) if (sk->obs_pos[sk->stack_ids[q]][2] < min_h) return 1;
  return 0;
}

/* _setup_observables (1306-1524) in the columns of the observation superset: vec_eef_to_all_objects (a, b, l, r) takes the 12 joint-space columns the cube
 * tasks leave empty ([12:18] + [33:39]); object_gripped 39, vec_eef_to_object 40:43 (the cube the robot has to place next: b, then a), vec_eef_to_target
 * 43:46, gripper_aperture 46, that cube's position 47:50, next_target_pos 50:53 (the eef position while there is no target) */
static void compute_obs_stack(const hrgo_batch* B, int64_t gid, const hrg_env_state* s, hrg_stack_state* sk, float* obs) {
  const hrg_model_desc* m = &B->m;
  double goal[NARM] = {0};
  compute_obs(m, s, NULL, goal, obs);
  double tgt[3];
  sk->has_target = stack_next_target(B, gid, s, sk, tgt);
  if (!sk->has_target) v3cpy(tgt, s->eef_pos);
  v3cpy(sk->target, tgt);
  for (int c = 0; c < NCUBE; c++)
    for (int a = 0; a < 3; a++) obs[(c < 2 ? 12 : 33) + 3 * (c & 1) + a] = (float)(sk->obs_pos[c][a] - s->eef_pos[a]);
  const int nxt = sk->n_stack >= 2 ? HRG_CUBE_A : HRG_CUBE_B; /* vec_eef_to_object (1445-1456) */
  obs[39] = (float)sk->gripped;
  double ap = 0;
  for (int f = 0; f < HRG_NFINGER; f++) ap += (s->qpos[NARM + f] - m->finger_qpos_range[0][f]) / (m->finger_qpos_range[1][f] - m->finger_qpos_range[0][f]);
  obs[46] = (float)(ap / HRG_NFINGER);
  for (int a = 0; a < 3; a++) {
    obs[40 + a] = (float)(sk->obs_pos[nxt][a] - s->eef_pos[a]);
    obs[43 + a] = (float)(tgt[a] - s->eef_pos[a]);
    obs[47 + a] = (float)sk->obs_pos[nxt][a];
    obs[50 + a] = (float)tgt[a];
  }
}

/* CollaborativeStackingCart._reset_internal (938-959) after the common part of env_reset */
static void env_reset_stack(hrgo_batch* B, int e, const robot_kin* k) {
  const hrg_model_desc* m = &B->m;
  hrg_env_state* s = &B->st[e];
  hrg_stack_state* sk = &B->stk[e];
  const int64_t gid = B->env_id0 + e;
  (void)k;
  memset(sk, 0, sizeof *sk);
  for (int c = HRG_CUBE_A; c <= HRG_CUBE_B; c++) {
    stack_placement(B, gid, s->episode, 0, c, sk->pos[c]);
    sk->quat[c][0] = 1;
    v3cpy(sk->obs_pos[c], sk->pos[c]);
  }
  human_kin hk;
  double mp[3], mq[4];
  const double* qh;
  human_control_sk(B, gid, s, NULL, sk, mp, mq, &qh); /* _reset_animation + _control_human: phase APPROACH, the human holds both cubes */
  human_fk(m, mp, mq, qh, &hk, s->human_site);
  stack_mocap(m, s, sk, &hk);
  stack_reset_animation(m, sk);
}

static void env_step_stack(hrgo_batch* B, int e, double* action, float* obs, float* term_obs, float* reward, uint8_t* done, int32_t* info) {
  const hrg_model_desc* m = &B->m;
  hrg_env_state* s = &B->st[e];
  hrg_stack_state* sk = &B->stk[e];
  int64_t gid = s->stream_id;
  const double h = m->timestep;
  if (m->ik_enabled) ik_action(m, s, action);
  screen_action(B, gid, s, action);
  s->timestep += 1;
  int has_collision = 0, collision_type = HRG_COL_NULL, failsafe_intervention = 0, crash = 0;
  robot_kin k;
  human_kin hk;
  double M[NV * NV], bias[NV];
  for (int cyc = 0; cyc < m->n_cycles && !crash; cyc++) {
    robot_fk(m, s->qpos, &k);
    robot_crba(m, &k, M);
    robot_bias(m, &k, s->qvel, bias);
    if (cyc == 0) { /* failsafe_controller.py:252-300 */
      for (int i = 0; i < NARM; i++) for (int j = 0; j < NARM; j++) s->mass_matrix[i * NARM + j] = M[i * NV + j];
      double scale = fabs(m->act_out_max - m->act_out_min) / fabs(m->act_in_max - m->act_in_min);
      double otr = 0.5 * (m->act_out_max + m->act_out_min), itr = 0.5 * (m->act_in_max + m->act_in_min);
      for (int j = 0; j < NARM; j++) {
        double a = clampd(action[j], m->act_in_min, m->act_in_max);
        double g = s->qpos[j] + ((a - itr) * scale + otr);
        s->goal_qpos[j] = clampd(g, m->qpos_limits[0][j], m->qpos_limits[1][j]);
        s->new_goal_q[j] = s->goal_qpos[j];
      }
      s->new_goal = 1;
    }
    shield_step(B, e, s->time);
    double ctrl[NV];
    for (int i = 0; i < NARM; i++) {
      double t = 0;
      for (int j = 0; j < NARM; j++) t += s->mass_matrix[i * NARM + j] * (m->kp * (s->des_q[j] - s->qpos[j]) + m->kd * (s->des_v[j] - s->qvel[j]) + s->des_a[j]);
      ctrl[i] = clampd(t + bias[i], m->arm_ctrlrange[i][0], m->arm_ctrlrange[i][1]);
    }
    {
      double a = action[NARM], sg = a > 0 ? 1.0 : (a < 0 ? -1.0 : 0.0);
      s->grip_action = clampd(s->grip_action - m->gripper_speed * sg, -1.0, 1.0);
      for (int f = 0; f < HRG_NFINGER; f++) {
        double lo = m->finger_ctrlrange[f][0], hi = m->finger_ctrlrange[f][1];
        ctrl[NARM + f] = 0.5 * (hi + lo) + 0.5 * (hi - lo) * (f == 0 ? s->grip_action : -s->grip_action);
      }
    }
    if (!failsafe_intervention && !s->is_safe) { failsafe_intervention = 1; s->failsafe_interventions++; }
    /* _control_human (899-903): super + sim.forward() + the two hand mocap bodies */
    double mp[3], mq[4];
    const double* qh;
    human_control_sk(B, gid, s, NULL, sk, mp, mq, &qh);
    human_fk(m, mp, mq, qh, &hk, s->human_site);
    stack_mocap(m, s, sk, &hk);
    contact_t con[HRG_NCON_MAX];
    int ncon = collide_stack(m, &k, &hk, sk, con);
    double rc[HRG_NRCAP][3], Rb[9];
    quat2mat(Rb, m->base_quat);
    for (int c = 0; c < HRG_NRCAP; c++) {
      int b = m->rcap_body[c];
      double t[3], mid[3];
      for (int a = 0; a < 3; a++) mid[a] = 0.5 * (m->rcap_p1[c][a] + m->rcap_p2[c][a]);
      m3mulv(t, b < 0 ? Rb : k.R[b], mid);
      v3add(rc[c], b < 0 ? m->base_pos : k.p[b], t);
    }
    classify(m, &k, s, con, ncon, rc, &has_collision, &collision_type);
    s->ncon = ncon;
    for (int c = 0; c < HRG_NCON_MAX; c++) { s->con_pairs[c][0] = c < ncon ? con[c].g1 : -1; s->con_pairs[c][1] = c < ncon ? con[c].g2 : -1; }
    /* ---- sim.step(): robot tree + four free cubes (block-diagonal M; cubes: m 1, R diag(I) R' with the mean / deviation split of the single cube) ---- */
    double LM[NV * NV], a0[NVMAX], qd[NVMAX], frc[NV];
    static const int NVK = NVMAX;
    double* Mt = (double*)calloc((size_t)NVK * NVK, sizeof(double));
    memcpy(LM, M, sizeof LM);
    if (!chol(LM, NV)) { crash = 1; free(Mt); break; }
    for (int i = 0; i < NV; i++) {
      double act = ctrl[i];
      if (i >= NARM) act = clampd(m->finger_kp * (ctrl[i] - s->qpos[i]), m->finger_forcerange[0], m->finger_forcerange[1]);
      frc[i] = act - m->jnt_damping[i] * s->qvel[i] - bias[i];
      a0[i] = frc[i];
      qd[i] = s->qvel[i];
    }
    chol_solve(LM, NV, a0);
    for (int i = 0; i < NV; i++) for (int j = 0; j < NV; j++) Mt[i * NVK + j] = M[i * NV + j];
    for (int c = 0; c < NCUBE; c++) {
      const int o = NV + 6 * c;
      double Rx[9], Mdev[9], w[3] = {sk->vel[c][3], sk->vel[c][4], sk->vel[c][5]}, Ld[3], Lw[3], tau[3], tl[3];
      quat2mat(Rx, sk->quat[c]);
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double t = 0;
        for (int kk = 0; kk < 3; kk++) t += Rx[3 * i + kk] * (m->box_inertia[kk] - m->box_inertia_mean) * Rx[3 * j + kk];
        Mdev[3 * i + j] = t;
      }
      for (int kk = 0; kk < 3; kk++) Ld[kk] = (m->box_inertia[kk] - m->box_inertia_mean) * (Rx[kk] * w[0] + Rx[3 + kk] * w[1] + Rx[6 + kk] * w[2]);
      m3mulv(Lw, Rx, Ld);
      v3cross(tau, Lw, w);
      for (int kk = 0; kk < 3; kk++) tl[kk] = (Rx[kk] * tau[0] + Rx[3 + kk] * tau[1] + Rx[6 + kk] * tau[2]) / m->box_inertia[kk];
      for (int a = 0; a < 3; a++) {
        Mt[(o + a) * NVK + o + a] = m->box_mass;
        for (int b = 0; b < 3; b++) Mt[(o + 3 + a) * NVK + o + 3 + b] = (a == b ? m->box_inertia_mean : 0.0) + Mdev[3 * a + b];
        a0[o + a] = m->gravity[a];
        a0[o + 3 + a] = Rx[3 * a] * tl[0] + Rx[3 * a + 1] * tl[1] + Rx[3 * a + 2] * tl[2];
      }
      for (int a = 0; a < 6; a++) qd[o + a] = sk->vel[c][a];
    }
    efc_t* E = (efc_t*)malloc(sizeof(efc_t));
    E->n = 0;
    E->nv = NVK;
    for (int i = 0; i < NV; i++)
      if (m->jnt_frictionloss[i] > 0) { double J[NVMAX] = {0}; J[i] = 1; efc_add(m, E, J, qd, ROW_FRICTION, 0, 0, m->jnt_frictionloss[i], m->dof_invweight0[i]); }
    for (int i = 0; i < NV; i++) {
      double dlo = s->qpos[i] - m->jnt_range[i][0], dhi = m->jnt_range[i][1] - s->qpos[i];
      if (dlo < 0) { double J[NVMAX] = {0}; J[i] = 1; efc_add(m, E, J, qd, ROW_UNILATERAL, dlo, 0, 0, m->dof_invweight0[i]); }
      if (dhi < 0) { double J[NVMAX] = {0}; J[i] = -1; efc_add(m, E, J, qd, ROW_UNILATERAL, dhi, 0, 0, m->dof_invweight0[i]); }
    }
    for (int c = 0; c < ncon && c < HRG_NCON_DYN_STACK; c++) {
      const double* n = con[c].n;
      double t1[3], t2[3], e1[3] = {1, 0, 0}, e2[3] = {0, 1, 0};
      v3cross(t1, n, fabs(n[0]) < 0.5 ? e1 : e2);
      v3scl(t1, t1, 1.0 / v3norm(t1));
      v3cross(t2, n, t1);
      double margin = (con[c].g2 >= GEOM_HUMAN0 && con[c].g2 < GEOM_TABLE) || (con[c].g1 >= GEOM_HUMAN0 && con[c].g1 < GEOM_TABLE) ? m->contact_margin_human : 0.0; /* a human geom on either side */
      for (int d = 0; d < 4; d++) {
        double dir[3], J[NVMAX] = {0};
        const double* tt = d < 2 ? t1 : t2;
        double sg = (d & 1) ? -1.0 : 1.0;
        for (int a = 0; a < 3; a++) dir[a] = n[a] + sg * m->friction_static * tt[a];
        if (con[c].b1 >= 0 && con[c].b1 < NV) robot_point_jac(m, &k, con[c].b1, con[c].pos, dir, -1.0, J);
        if (con[c].b2 >= 0 && con[c].b2 < NV) robot_point_jac(m, &k, con[c].b2, con[c].pos, dir, +1.0, J);
        double diag = (con[c].b1 >= 0 && con[c].b1 < NV ? m->body_invweight0[con[c].b1] : 0.0) + (con[c].b2 >= 0 && con[c].b2 < NV ? m->body_invweight0[con[c].b2] : 0.0);
        for (int side = 0; side < 2; side++) { /* free bodies: J = +-dir . (v + w x r), body_invweight0 = 1/m */
          const int body = side ? con[c].b2 : con[c].b1;
          if (body < BODY_BOX) continue;
          const int cb = body - BODY_BOX, o = NV + 6 * cb;
          const double sgn = side ? 1.0 : -1.0;
          double r[3], rxd[3];
          v3sub(r, con[c].pos, sk->pos[cb]);
          v3cross(rxd, r, dir);
          for (int a = 0; a < 3; a++) { J[o + a] = sgn * dir[a]; J[o + 3 + a] = sgn * rxd[a]; }
          diag += 1.0 / m->box_mass;
        }
        efc_add(m, E, J, qd, ROW_UNILATERAL, con[c].dist, margin, 0, diag * (1.0 + m->friction_static * m->friction_static));
      }
    }
    for (int hd = 0; hd < 2; hd++) { /* lh_weld_eq / rh_weld_eq (1255-1284): residual [p_cube - p_target; rotation vector of q_cube q_mocap^-1] on the cube's own DoF */
      if (!sk->weld_active[hd]) continue;
      const int cb = HRG_CUBE_L + hd, o = NV + 6 * cb;
      double tp[3], epos[3], erot[3], qe[4];
      stack_weld_target(m, sk, hd, tp);
      v3sub(epos, sk->pos[cb], tp);
      double qc[4] = {sk->mocap_quat[hd][0], -sk->mocap_quat[hd][1], -sk->mocap_quat[hd][2], -sk->mocap_quat[hd][3]};
      quatmul(qe, sk->quat[cb], qc);
      if (qe[0] < 0) for (int a = 0; a < 4; a++) qe[a] = -qe[a];
      const double sn = sqrt(qe[1] * qe[1] + qe[2] * qe[2] + qe[3] * qe[3]), ang = 2.0 * atan2(sn, qe[0]);
      for (int a = 0; a < 3; a++) erot[a] = sn > 1e-12 ? qe[1 + a] / sn * ang : 0.0;
      for (int a = 0; a < 3; a++) { double J[NVMAX] = {0}; J[o + a] = 1; efc_add(m, E, J, qd, ROW_EQUALITY, epos[a], 0, 0, 1.0 / m->box_mass); }
      for (int a = 0; a < 3; a++) { double J[NVMAX] = {0}; J[o + 3 + a] = 1; efc_add(m, E, J, qd, ROW_EQUALITY, erot[a], 0, 0, m->box_invweight_rot); }
    }
    double qacc[NVMAX];
    memcpy(qacc, s->qacc_warmstart, sizeof(double) * NV);
    for (int c = 0; c < NCUBE; c++) memcpy(qacc + NV + 6 * c, sk->acc_warmstart[c], sizeof(double) * 6);
    solve(m, Mt, a0, E, qacc);
    if (g_debug && (ncon > 0 || g_debug > 1)) {
      double mx = 0; for (int i = 0; i < NVK; i++) if (fabs(qacc[i]) > mx) mx = fabs(qacc[i]);
      fprintf(stderr, "[oracle stack] env %d cyc %d ncon %d nefc %d max|qacc| %.3e", e, cyc, ncon, E->n, mx);
      for (int c = 0; c < ncon; c++) fprintf(stderr, " (%d,%d d=%.5f)", con[c].g1, con[c].g2, con[c].dist);
      fprintf(stderr, "\n");
    }
    free(E); free(Mt);
    for (int i = 0; i < NVK; i++) if (!(fabs(qacc[i]) < 1e10)) crash = 1;
    if (crash) break;
    memcpy(s->qacc_warmstart, qacc, sizeof(double) * NV);
    double Mh[NV * NV], rhs[NV];
    memcpy(Mh, M, sizeof Mh);
    for (int i = 0; i < NV; i++) { Mh[i * NV + i] += h * m->jnt_damping[i]; double t = 0; for (int j = 0; j < NV; j++) t += M[i * NV + j] * qacc[j]; rhs[i] = t; }
    if (!chol(Mh, NV)) { crash = 1; break; }
    chol_solve(Mh, NV, rhs);
    for (int i = 0; i < NV; i++) { s->qvel[i] += h * rhs[i]; s->qpos[i] += h * s->qvel[i]; }
    for (int c = 0; c < NCUBE; c++) {
      memcpy(sk->acc_warmstart[c], qacc + NV + 6 * c, sizeof(double) * 6);
      v3cpy(sk->obs_pos[c], sk->pos[c]);
      for (int a = 0; a < 6; a++) sk->vel[c][a] += h * qacc[NV + 6 * c + a];
      for (int a = 0; a < 3; a++) sk->pos[c][a] += h * sk->vel[c][a];
      double w[3] = {sk->vel[c][3], sk->vel[c][4], sk->vel[c][5]}, wn = v3norm(w), ang = h * wn;
      if (wn > 1e-12) {
        double sh = sin(0.5 * ang) / wn, dq[4] = {cos(0.5 * ang), w[0] * sh, w[1] * sh, w[2] * sh}, qn[4];
        quatmul(qn, dq, sk->quat[c]);
        double nn = sqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
        for (int a = 0; a < 4; a++) sk->quat[c][a] = qn[a] / nn;
      }
    }
    s->time += h;
    eef_of(m, &k, s->eef_pos);
    s->low_level_time += 1;
  }
  /* ---- observation, success, info, reward, done ---- */
  compute_obs_stack(B, gid, s, sk, term_obs);
  const int goal_reached = !crash && sk->task_phase == HRG_STK_COMPLETE; /* _check_success (746-756) */
  const int toppled = stack_toppled(m, sk);
  double r;
  if (goal_reached) r = m->task_reward; /* _sparse_reward (700-744) */
  else if (toppled) r = m->stack_toppled_reward;
  else {
    if (stack_second_in_zone(B, gid, s, sk) >= 0) r = m->fourth_cube_at_target_reward;
    else if (stack_first_in_zone(B, gid, s, sk) >= 0) r = m->second_cube_at_target_reward;
    else r = -1.0;
    if (sk->gripped) r += m->object_gripped_reward;
  }
  if (goal_reached) s->n_goal_reached++;
  int illegal = (collision_type & (HRG_COL_STATIC | HRG_COL_ROBOT | HRG_COL_HUMAN_CRIT)) != 0;
  if (m->reward_shaping) r += 1.0 + 0.0; /* _dense_reward is a TODO returning 0 (681-698) */
  if (illegal) r += m->collision_reward;
  r *= m->reward_scale;
  int d = 0;
  if (crash) { r += m->sim_crash_reward; d = 1; }
  else {
    if (toppled) d = 1; /* _check_done (758-778) */
    if (m->done_at_collision && illegal) d = 1;
    if (m->done_at_success && goal_reached) d = 1;
  }
  int ncoll = s->n_collisions_static + s->n_collisions_robot + s->n_collisions_human + s->n_collisions_critical;
  info[HRG_INFO_COLLISION] = has_collision;
  info[HRG_INFO_COLLISION_TYPE] = collision_type;
  info[HRG_INFO_N_COLLISIONS] = ncoll;
  info[HRG_INFO_N_COLLISIONS_STATIC] = s->n_collisions_static;
  info[HRG_INFO_N_COLLISIONS_ROBOT] = s->n_collisions_robot;
  info[HRG_INFO_N_COLLISIONS_HUMAN] = s->n_collisions_human;
  info[HRG_INFO_N_COLLISIONS_CRITICAL] = s->n_collisions_critical;
  info[HRG_INFO_TIMEOUT] = s->timestep >= m->horizon;
  info[HRG_INFO_FAILSAFE_INTERVENTIONS] = s->failsafe_interventions;
  info[HRG_INFO_N_GOAL_REACHED] = s->n_goal_reached;
  info[HRG_INFO_SIM_CRASH] = crash;
  info[HRG_INFO_TRUNCATED] = 0;
  info[HRG_INFO_ACTION_RESAMPLES] = s->action_resamples;
  info[HRG_INFO_MAX_STACK_HEIGHT] = sk->max_stack_height; /* _get_info (673-679): the value before this step's transitions */
  /* ---- CollaborativeStackingCart.step tail (550-588) ---- */
  if (goal_reached && !m->done_at_success && !d) { /* _on_goal_reached (961-977): next placements of the robot's cubes (velocities kept), next animation */
    sk->obj_index = (sk->obj_index + 1) % m->n_obj_placements;
    for (int c = HRG_CUBE_A; c <= HRG_CUBE_B; c++) {
      stack_placement(B, gid, s->episode, sk->obj_index, c, sk->pos[c]);
      sk->quat[c][0] = 1; sk->quat[c][1] = sk->quat[c][2] = sk->quat[c][3] = 0;
    }
    s->anim_index = (s->anim_index + 1) % m->n_anim_ids;
    s->animation_time = 0;
    s->anim_start_time = (int)((double)s->low_level_time / m->anim_step_length);
    sk->task_phase = HRG_STK_APPROACH; sk->n_delayed[0] = sk->n_delayed[1] = 0;
    human_kin hk2;
    double mp[3], mq[4];
    const double* qh;
    human_control_sk(B, gid, s, NULL, sk, mp, mq, &qh);
    human_fk(m, mp, mq, qh, &hk2, s->human_site);
    stack_mocap(m, s, sk, &hk2);
    stack_reset_animation(m, sk);
  }
  if (!d) {
    const int left = B->clips.clip_holding_hand[clip_of(B, gid, s, s->anim_index)]; /* first_placing_hand */
    const int32_t* kf = B->clips.clip_stack_keyframes[clip_of(B, gid, s, s->anim_index)];
    int body;
    if (sk->task_phase == HRG_STK_PLACE_FIRST && s->animation_time > kf[1]) { /* _human_place_first_object (1004-1011) */
      const int hd = left ? 0 : 1;
      sk->stack_ids[sk->n_stack++] = HRG_CUBE_L + hd;
      sk->weld_active[hd] = 0;
      sk->task_phase = HRG_STK_WAIT_FOR_SECOND;
    } else if (sk->task_phase == HRG_STK_WAIT_FOR_SECOND && (body = stack_first_in_zone(B, gid, s, sk)) >= 0 && !sk->gripped) {
      sk->task_phase = HRG_STK_PLACE_THIRD;
      sk->stack_ids[sk->n_stack++] = body;
    } else if (sk->task_phase == HRG_STK_PLACE_THIRD && s->animation_time > kf[3]) { /* _human_place_third_object (1013-1043): released directly above the second cube */
      const int hd = left ? 1 : 0, c = HRG_CUBE_L + hd;
      sk->weld_active[hd] = 0;
      v3cpy(sk->pos[c], sk->obs_pos[sk->stack_ids[1]]);
      sk->pos[c][2] += 2.0 * m->box_half[2];
      sk->quat[c][0] = 1; sk->quat[c][1] = sk->quat[c][2] = sk->quat[c][3] = 0;
      for (int a = 0; a < 6; a++) sk->vel[c][a] = 0;
      sk->stack_ids[sk->n_stack++] = c;
      sk->task_phase = HRG_STK_WAIT_FOR_FOURTH;
    } else if (sk->task_phase == HRG_STK_WAIT_FOR_FOURTH && (body = stack_second_in_zone(B, gid, s, sk)) >= 0 && !sk->gripped) {
      sk->task_phase = HRG_STK_RETREAT;
      sk->stack_ids[sk->n_stack++] = body;
    }
    if (sk->n_stack > sk->max_stack_height) sk->max_stack_height = sk->n_stack;
  }
  if (s->timestep >= m->horizon) { info[HRG_INFO_TRUNCATED] = !d; d = 1; }
  *reward = (float)r;
  *done = (uint8_t)d;
  if (d) env_reset(B, e, obs);
  else memcpy(obs, term_obs, sizeof(float) * HRG_OBS_DIM);
}

/* =============================================================================================== CollaborativeHammeringCart
 * collaborative_hammering_cartesian_env.py: the human carries a board (free joint; held by a weld at the right and a connect at the left hand mocap body),
 * a nail rides on the board on a slide joint (models/assets/objects/nail.xml), the robot holds a hammer (free joint) in its closed gripper and has to drive
 * the nail in while the human presents the board; three-phase animation machine.  Constrained system: 24 DoF in three blocks of eight --
 * robot tree 0..7 | board 8..13 + the nail's slide joint 14 (+ pad) | hammer 16..21 (+ 2 pads); a pad DoF has unit mass, no force and no constraint row.
 * Stand-ins: the hammer is two boxes (robosuite's composite HammerObject is absent), the nail head a box, box-box contacts by box_box2 (D13),
 * capsule-box contacts as for the single cube (D8); MuJoCo's noslip post-pass (noslip_iterations = 20, 1161): noslip() above. */
#define NVH HRG_NV_HAMMER
#define HM_OB NV        /* board DoF */
#define HM_ON (NV + 6)  /* nail slide joint */
#define HM_OH (NV + 8)  /* hammer DoF */
#define GEOM_HM(g) (GEOM_BOX + (g))
#define BODY_HM(b) (BODY_BOX + (b))

/* CollaborativeHammeringCart._compute_animation_time (636-680), applied to the classic animation time *at_io */
static void hammer_animation_time(const hrgo_batch* b, int64_t gid, const hrg_env_state* s, hrg_hammer_state* hm, int clip, int* at_io) {
  const int classic = *at_io, len = b->clips.clip_len[clip];
  const double k0 = (double)b->clips.clip_keyframes[clip][0], mid = 0.5 * (k0 + (double)b->clips.clip_keyframes[clip][1]);
  int at = classic;
  if (hm->task_phase == HRG_HM_APPROACH && (double)at > k0) hm->task_phase = HRG_HM_PRESENT;
  else if (hm->task_phase == HRG_HM_PRESENT && (double)at > mid) { /* idle loop until the nail is hammered in */
    at = (int)layered_sines(b, gid, s, clip, 0, b->clips.clip_n_loop[clip], (double)classic, mid);
    hm->n_delayed = classic - at;
  } else if (hm->task_phase == HRG_HM_RETREAT) at -= hm->n_delayed;
  if (at >= len - 1) { hm->task_phase = HRG_HM_COMPLETE; at = len - 1; }
  if (at < 0) at = 0;
  *at_io = at;
}

/* world poses of the four collision geoms, the nail_head body origin and the slide axis */
typedef struct { double c[HRG_HM_NGEOM][3], R[2][9], nail_org[3], axis[3]; } hammer_geo;
static const int HM_GEOM_BODY[HRG_HM_NGEOM] = {HRG_HM_BOARD, HRG_HM_HAMMER, HRG_HM_HAMMER, HRG_HM_NAIL};
static void hammer_geometry(const hrg_model_desc* m, const hrg_hammer_state* hm, hammer_geo* G) {
  double t[3];
  quat2mat(G->R[0], hm->quat[0]);
  quat2mat(G->R[1], hm->quat[1]);
  v3cpy(G->c[HRG_HG_BOARD], hm->pos[0]);
  for (int g = HRG_HG_HANDLE; g <= HRG_HG_HEAD; g++) { m3mulv(t, G->R[1], m->hm_geom_pos[g]); v3add(G->c[g], hm->pos[1], t); }
  const double loc[3] = {hm->nail_xy[0], hm->nail_xy[1], m->hm_nail_z0 - hm->nail_q};
  m3mulv(t, G->R[0], loc); v3add(G->nail_org, hm->pos[0], t);
  m3mulv(t, G->R[0], m->hm_geom_pos[HRG_HG_NAIL]); v3add(G->c[HRG_HG_NAIL], G->nail_org, t);
  for (int a = 0; a < 3; a++) G->axis[a] = -G->R[0][3 * a + 2]; /* joint axis (0, 0, -1) of the board */
}

/* contact list of the hammering task: the robot's own contacts (collide), then per geom g (board, handle, head, nail head) the robot capsules' first points, the
 * corners on the table and on the floor, then the box pairs (head - nail, handle - nail, head - board, handle - board), then the second points of capsules
 * lying along a face.  The handle meets the two finger bars only (the stand-in hand capsule envelops the real gripper's palm, through which the handle
 * passes).  hammer_gripped (1283-1289): _check_grasp = both fingers touch a geom of the hammer. */
static int collide_hammer(const hrg_model_desc* m, const robot_kin* k, const human_kin* h, hrg_hammer_state* hm, const hammer_geo* G, contact_t* con) {
  int n = collide(m, k, h, NULL, con);
  double rp1[HRG_NRCAP][3], rp2[HRG_NRCAP][3], Rb[9];
  quat2mat(Rb, m->base_quat);
  for (int c = 0; c < HRG_NRCAP; c++) {
    int b = m->rcap_body[c];
    const double* R = b < 0 ? Rb : k->R[b];
    const double* p = b < 0 ? m->base_pos : k->p[b];
    double t[3];
    m3mulv(t, R, m->rcap_p1[c]); v3add(rp1[c], p, t);
    m3mulv(t, R, m->rcap_p2[c]); v3add(rp2[c], p, t);
  }
#define EMIT(G1, G2, B1, B2, DIST, NRM, POS) \
  do { if (n < HRG_NCON_MAX) { con[n].g1 = G1; con[n].g2 = G2; con[n].b1 = B1; con[n].b2 = B2; con[n].dist = DIST; v3cpy(con[n].n, NRM); v3cpy(con[n].pos, POS); n++; } } while (0)
  int n_second = 0, second_i[HRG_HM_NGEOM * HRG_NRCAP], second_g[HRG_HM_NGEOM * HRG_NRCAP];
  double second_s[HRG_HM_NGEOM * HRG_NRCAP][3], second_b[HRG_HM_NGEOM * HRG_NRCAP][3];
  int f0 = 0, f1 = 0;
  for (int g = HRG_HG_BOARD; g <= HRG_HG_NAIL; g++) {
    const double* Rx = G->R[HM_GEOM_BODY[g] == HRG_HM_HAMMER ? 1 : 0]; /* the nail head turns with the board */
    const double* hb = m->hm_geom_half[g];
    const int body = BODY_HM(HM_GEOM_BODY[g]);
    for (int i = 0; i < HRG_NRCAP; i++) {
      if (m->rcap_body[i] < 0) continue;
      if (g == HRG_HG_HANDLE && i < HRG_NRCAP - 2) continue;
      double t, cs[3], cb[3], nn[3], pos[3], s2[2][3], b2[2][3];
      double e2 = seg_box(rp1[i], rp2[i], G->c[g], Rx, hb, &t, cs, cb), dd = sqrt(e2), dist = dd - m->rcap_r[i];
      if (!(dist < 0)) continue;
      if (dd > 1e-9 && cap_box_two(rp1[i], rp2[i], G->c[g], Rx, hb, m->rcap_r[i], cs, cb, s2, b2)) {
        v3cpy(cs, s2[0]); v3cpy(cb, b2[0]);
        v3sub(nn, cb, cs); dd = v3norm(nn); dist = dd - m->rcap_r[i];
        second_i[n_second] = i; second_g[n_second] = g; v3cpy(second_s[n_second], s2[1]); v3cpy(second_b[n_second], b2[1]); n_second++;
      }
      if (dd > 1e-9) { v3sub(nn, cb, cs); v3scl(nn, nn, 1.0 / dd); }
      else { /* capsule axis inside the box: push out through the nearest face */
        double loc[3], best = 1e300; int ax = 0;
        v3sub(pos, cs, G->c[g]);
        for (int a = 0; a < 3; a++) { loc[a] = Rx[a] * pos[0] + Rx[3 + a] * pos[1] + Rx[6 + a] * pos[2]; if (hb[a] - fabs(loc[a]) < best) { best = hb[a] - fabs(loc[a]); ax = a; } }
        double sg = loc[ax] >= 0 ? -1.0 : 1.0;
        for (int a = 0; a < 3; a++) nn[a] = sg * Rx[3 * a + ax];
        dist = -best - m->rcap_r[i];
      }
      v3madd(pos, cs, nn, m->rcap_r[i] + 0.5 * dist);
      if (n < HRG_NCON_MAX && (g == HRG_HG_HANDLE || g == HRG_HG_HEAD)) { f0 |= i == HRG_NRCAP - 2; f1 |= i == HRG_NRCAP - 1; }
      EMIT(i, GEOM_HM(g), m->rcap_body[i], body, dist, nn, pos);
    }
  }
  for (int pl = 0; pl < 2; pl++)
    for (int g = HRG_HG_BOARD; g <= HRG_HG_HEAD; g++) {
      const double* Rx = G->R[HM_GEOM_BODY[g] == HRG_HM_HAMMER ? 1 : 0];
      const double* hb = m->hm_geom_half[g];
      for (int cn = 0; cn < 8; cn++) {
        double loc[3] = {(cn & 1) ? hb[0] : -hb[0], (cn & 2) ? hb[1] : -hb[1], (cn & 4) ? hb[2] : -hb[2]}, p[3];
        m3mulv(p, Rx, loc);
        v3add(p, p, G->c[g]);
        double z0 = pl ? m->floor_z : m->table_top_z, dist = p[2] - z0;
        if (pl == 0 && !(fabs(p[0] - m->table_center[0]) <= m->table_half[0] && fabs(p[1] - m->table_center[1]) <= m->table_half[1] && p[2] > z0 - 0.05)) continue;
        if (dist < 0) {
          double nn[3] = {0, 0, 1}, pos[3] = {p[0], p[1], z0 + 0.5 * dist};
          EMIT(pl ? GEOM_FLOOR : GEOM_TABLE, GEOM_HM(g), -1, BODY_HM(HM_GEOM_BODY[g]), dist, nn, pos);
        }
      }
    }
  static const int PA[4] = {HRG_HG_HEAD, HRG_HG_HANDLE, HRG_HG_HEAD, HRG_HG_HANDLE}, PB[4] = {HRG_HG_NAIL, HRG_HG_NAIL, HRG_HG_BOARD, HRG_HG_BOARD};
  for (int q = 0; q < 4; q++) {
    const int ga = PA[q], gb = PB[q];
    const double *ha = m->hm_geom_half[ga], *hb = m->hm_geom_half[gb];
    double d[3];
    v3sub(d, G->c[gb], G->c[ga]);
    const double ra = sqrt(ha[0] * ha[0] + ha[1] * ha[1] + ha[2] * ha[2]), rb = sqrt(hb[0] * hb[0] + hb[1] * hb[1] + hb[2] * hb[2]);
    if (v3dot(d, d) > (ra + rb) * (ra + rb)) continue; /* circumspheres apart */
    bb_contact bc[4];
    const int nc = box_box2(G->c[ga], G->R[1], ha, G->c[gb], G->R[0], hb, bc);
    for (int z = 0; z < nc; z++) EMIT(GEOM_HM(ga), GEOM_HM(gb), BODY_HM(HRG_HM_HAMMER), BODY_HM(HM_GEOM_BODY[gb]), bc[z].dist, bc[z].n, bc[z].pos);
  }
  for (int q = 0; q < n_second; q++) {
    const int i = second_i[q], g = second_g[q];
    double nn[3], pos[3];
    v3sub(nn, second_b[q], second_s[q]);
    const double dd = v3norm(nn), dist = dd - m->rcap_r[i];
    v3scl(nn, nn, 1.0 / dd);
    v3madd(pos, second_s[q], nn, m->rcap_r[i] + 0.5 * dist);
    EMIT(i, GEOM_HM(g), m->rcap_body[i], BODY_HM(HM_GEOM_BODY[g]), dist, nn, pos);
  }
#undef EMIT
  hm->gripped = f0 && f1;
  return n;
}

/* _update_mocap_body_transforms (688-715): each mocap body sits at its hand site, rotated like the hand body turned by -90 deg (left) / +90 deg (right)
 * about its y axis */
static void hammer_mocap(const hrg_model_desc* m, const hrg_env_state* s, hrg_hammer_state* hm, const human_kin* hk) {
  for (int hd = 0; hd < 2; hd++) {
    const int site = hd == 0 ? m->site_lhand : m->site_rhand, body = m->meas_body[site];
    const double ang = hd == 0 ? -0.5 * PI : 0.5 * PI, c = cos(ang), sn = sin(ang);
    const double Ry[9] = {c, 0, sn, 0, 1, 0, -sn, 0, c};
    double R[9];
    v3cpy(hm->mocap_pos[hd], s->human_site[site]);
    m3mul(R, hk->R[body], Ry);
    mat2quat(hm->mocap_quat[hd], R);
  }
}
/* orientation the weld rh_eq wants for the board: mocap = board o relpose  =>  q_board = q_mocap o relquat^-1 (1123-1131) */
static void hammer_weld_quat(const hrg_model_desc* m, const hrg_hammer_state* hm, double* qt) {
  const double qi[4] = {m->hm_weld_relquat[0], -m->hm_weld_relquat[1], -m->hm_weld_relquat[2], -m->hm_weld_relquat[3]};
  quatmul(qt, hm->mocap_quat[1], qi);
}
/* _reset_board (769-779) + _human_take_board_from_table (831-835).  The reference puts the board on the table and lets the soft weld / connect drag it into
 * the hands over the next substeps; here it is put where the weld holds it, at rest (DESIGN.md D14's choice).  _reset_nail (781-788): pulled out. */
static void hammer_take_board(const hrgo_batch* B, int64_t gid, const hrg_env_state* s, hrg_hammer_state* hm) {
  const hrg_model_desc* m = &B->m;
  double Rt[9], t[3];
  hammer_weld_quat(m, hm, hm->quat[0]);
  quat2mat(Rt, hm->quat[0]);
  m3mulv(t, Rt, m->hm_anchor[1]);
  v3sub(hm->pos[0], hm->mocap_pos[1], t);
  for (int a = 0; a < 6; a++) { hm->vel[0][a] = 0; hm->acc_warmstart[0][a] = 0; }
  hm->nail_q = 0; hm->nail_v = 0; hm->nail_acc_warmstart = 0; hm->nail_touch = 0; hm->pad_ = 0;
  /* nail_placements[index]: UniformRandomSampler over the board (889-960), drawn counter-based on demand (D6) */
  hm->nail_xy[0] = m->hm_nail_bin[0] + (m->hm_nail_bin[1] - m->hm_nail_bin[0]) * rng_u01(m->seed, (uint64_t)gid, (uint64_t)s->episode, STREAM_OBJECT, (uint64_t)(2 * hm->nail_index));
  hm->nail_xy[1] = m->hm_nail_bin[2] + (m->hm_nail_bin[3] - m->hm_nail_bin[2]) * rng_u01(m->seed, (uint64_t)gid, (uint64_t)s->episode, STREAM_OBJECT, (uint64_t)(2 * hm->nail_index + 1));
}
static void hammer_obs_pos(const hrg_model_desc* m, hrg_hammer_state* hm) { /* body_xpos of board_main, the hammer's root body, nail_head */
  hammer_geo G;
  hammer_geometry(m, hm, &G);
  double t[3];
  v3cpy(hm->obs_pos[0], hm->pos[0]);
  m3mulv(t, G.R[1], m->hm_hammer_com); v3sub(hm->obs_pos[1], hm->pos[1], t);
  v3cpy(hm->obs_pos[2], G.nail_org);
}

/* _setup_observables (1151-1323) in the columns of the observation superset: hammer_quat (w, x, y, z as body_xquat gives it) 12:16, board_pos 33:36,
 * vec_eef_to_board 36:39, hammer_gripped 39, vec_eef_to_hammer 40:43, vec_eef_to_nail 43:46, gripper_aperture 46, hammer_pos 47:50, nail_pos 50:53,
 * board_quat (x, y, z, w) 57:61, nail_hammering_progress 61.  quat_eef_to_hammer / quat_eef_to_board look "object_quat" up in the observation cache,
 * which this env never fills: constant zeros (1217-1225, 1259-1267), served by the host. */
static void compute_obs_hammer(const hrgo_batch* B, const hrg_env_state* s, const hrg_hammer_state* hm, float* obs) {
  const hrg_model_desc* m = &B->m;
  double goal[NARM] = {0};
  compute_obs(m, s, NULL, goal, obs);
  for (int j = 0; j < NARM; j++) { obs[12 + j] = 0.0f; obs[33 + j] = 0.0f; }
  for (int a = 0; a < 4; a++) obs[12 + a] = (float)hm->quat[1][a];
  obs[39] = (float)hm->gripped;
  double ap = 0;
  for (int f = 0; f < HRG_NFINGER; f++) ap += (s->qpos[NARM + f] - m->finger_qpos_range[0][f]) / (m->finger_qpos_range[1][f] - m->finger_qpos_range[0][f]);
  obs[46] = (float)(ap / HRG_NFINGER);
  for (int a = 0; a < 3; a++) {
    obs[33 + a] = (float)hm->obs_pos[0][a];
    obs[36 + a] = (float)(hm->obs_pos[0][a] - s->eef_pos[a]);
    obs[40 + a] = (float)(hm->obs_pos[1][a] - s->eef_pos[a]);
    obs[43 + a] = (float)(hm->obs_pos[2][a] - s->eef_pos[a]);
    obs[47 + a] = (float)hm->obs_pos[1][a];
    obs[50 + a] = (float)hm->obs_pos[2][a];
    obs[57 + a] = (float)hm->quat[0][1 + a];
  }
  obs[60] = (float)hm->quat[0][0];
  obs[61] = (float)clampd(hm->nail_q / m->hm_nail_range, 0.0, 1.0);
}

/* CollaborativeHammeringCart._reset_internal (717-747) after the common part of env_reset */
static void env_reset_hammer(hrgo_batch* B, int e, const robot_kin* k) {
  const hrg_model_desc* m = &B->m;
  hrg_env_state* s = &B->st[e];
  hrg_hammer_state* hm = &B->hmr[e];
  const int64_t gid = B->env_id0 + e;
  (void)k;
  memset(hm, 0, sizeof *hm);
  hm->quat[0][0] = 1;
  human_kin hk;
  double mp[3], mq[4], Rg[9], t[3];
  const double* qh;
  human_control_all(B, gid, s, NULL, NULL, hm, mp, mq, &qh); /* _control_human + _reset_animation: phase APPROACH (memset), the human holds the board */
  human_fk(m, mp, mq, qh, &hk, s->human_site);
  hammer_mocap(m, s, hm, &hk);
  hammer_take_board(B, gid, s, hm);
  /* _put_hammer_into_gripper (790-812): the hammer's root body at the grip site, turned 90 deg about y; at rest */
  for (int a = 0; a < 4; a++) hm->quat[1][a] = m->hm_hammer_grip_quat[a];
  quat2mat(Rg, hm->quat[1]);
  m3mulv(t, Rg, m->hm_hammer_com);
  v3add(hm->pos[1], s->eef_pos, t);
  hammer_obs_pos(m, hm);
}

/* mass matrix (row-major 8 x 8, the pad DoF with a unit diagonal) and applied force of the board + nail subtree: the board as a free body with world-frame
 * angular velocity, the nail head as a point mass at r = R rho(q) that slides along a = -R e_z:  v_nail = v + w x r + a qd */
static void hammer_board_block(const hrg_model_desc* m, const hrg_hammer_state* hm, const hammer_geo* G, double* M8, double* f8) {
  const double mb = m->hm_board_mass, mn = m->hm_nail_mass;
  const double* R = G->R[0];
  double r[3], Iw[9], w[3] = {hm->vel[0][3], hm->vel[0][4], hm->vel[0][5]}, Lw[3], gy[3], c[3], t1[3], t2[3], rxa[3], rxg[3], rxc[3];
  v3sub(r, G->c[HRG_HG_NAIL], hm->pos[0]);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double t = 0; for (int kk = 0; kk < 3; kk++) t += R[3 * i + kk] * m->hm_board_inertia[kk] * R[3 * j + kk]; Iw[3 * i + j] = t; }
  m3mulv(Lw, Iw, w);
  v3cross(gy, Lw, w); /* -(w x I w) */
  v3cross(t1, w, r); v3cross(t1, w, t1); /* w x (w x r) */
  v3cross(t2, w, G->axis); /* w x a */
  for (int a = 0; a < 3; a++) c[a] = t1[a] + 2.0 * hm->nail_v * t2[a];
  v3cross(rxa, r, G->axis); v3cross(rxg, r, m->gravity); v3cross(rxc, r, c);
  memset(M8, 0, sizeof(double) * 64);
  const double rr = v3dot(r, r);
  const double rx[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0}; /* [r]x */
  for (int i = 0; i < 3; i++) {
    M8[i * 8 + i] = mb + mn;
    for (int j = 0; j < 3; j++) {
      M8[i * 8 + 3 + j] = -mn * rx[3 * i + j];
      M8[(3 + i) * 8 + j] = mn * rx[3 * i + j];
      M8[(3 + i) * 8 + 3 + j] = Iw[3 * i + j] + mn * ((i == j ? rr : 0.0) - r[i] * r[j]);
    }
    M8[i * 8 + 6] = M8[6 * 8 + i] = mn * G->axis[i];
    M8[(3 + i) * 8 + 6] = M8[6 * 8 + 3 + i] = mn * rxa[i];
    f8[i] = (mb + mn) * m->gravity[i] - mn * c[i];
    f8[3 + i] = gy[i] + mn * rxg[i] - mn * rxc[i];
  }
  M8[6 * 8 + 6] = mn;
  M8[7 * 8 + 7] = 1.0;
  f8[6] = mn * v3dot(G->axis, m->gravity) - mn * v3dot(G->axis, c);
  f8[7] = 0.0;
}

static void env_step_hammer(hrgo_batch* B, int e, double* action, float* obs, float* term_obs, float* reward, uint8_t* done, int32_t* info) {
  const hrg_model_desc* m = &B->m;
  hrg_env_state* s = &B->st[e];
  hrg_hammer_state* hm = &B->hmr[e];
  int64_t gid = s->stream_id;
  const double h = m->timestep;
  if (m->ik_enabled) ik_action(m, s, action);
  screen_action(B, gid, s, action);
  if (!m->gripper_controllable) action[NARM] = 1; /* CollaborativeHammeringCart.step (486-487): always close the gripper */
  s->timestep += 1;
  int has_collision = 0, collision_type = HRG_COL_NULL, failsafe_intervention = 0, crash = 0;
  robot_kin k;
  human_kin hk;
  double M[NV * NV], bias[NV];
  for (int cyc = 0; cyc < m->n_cycles && !crash; cyc++) {
    robot_fk(m, s->qpos, &k);
    robot_crba(m, &k, M);
    robot_bias(m, &k, s->qvel, bias);
    if (cyc == 0) { /* failsafe_controller.py:252-300 */
      for (int i = 0; i < NARM; i++) for (int j = 0; j < NARM; j++) s->mass_matrix[i * NARM + j] = M[i * NV + j];
      double scale = fabs(m->act_out_max - m->act_out_min) / fabs(m->act_in_max - m->act_in_min);
      double otr = 0.5 * (m->act_out_max + m->act_out_min), itr = 0.5 * (m->act_in_max + m->act_in_min);
      for (int j = 0; j < NARM; j++) {
        double a = clampd(action[j], m->act_in_min, m->act_in_max);
        double g = s->qpos[j] + ((a - itr) * scale + otr);
        s->goal_qpos[j] = clampd(g, m->qpos_limits[0][j], m->qpos_limits[1][j]);
        s->new_goal_q[j] = s->goal_qpos[j];
      }
      s->new_goal = 1;
    }
    shield_step(B, e, s->time);
    double ctrl[NV];
    for (int i = 0; i < NARM; i++) {
      double t = 0;
      for (int j = 0; j < NARM; j++) t += s->mass_matrix[i * NARM + j] * (m->kp * (s->des_q[j] - s->qpos[j]) + m->kd * (s->des_v[j] - s->qvel[j]) + s->des_a[j]);
      ctrl[i] = clampd(t + bias[i], m->arm_ctrlrange[i][0], m->arm_ctrlrange[i][1]);
    }
    {
      double a = action[NARM], sg = a > 0 ? 1.0 : (a < 0 ? -1.0 : 0.0);
      s->grip_action = clampd(s->grip_action - m->gripper_speed * sg, -1.0, 1.0);
      for (int f = 0; f < HRG_NFINGER; f++) {
        double lo = m->finger_ctrlrange[f][0], hi = m->finger_ctrlrange[f][1];
        ctrl[NARM + f] = 0.5 * (hi + lo) + 0.5 * (hi - lo) * (f == 0 ? s->grip_action : -s->grip_action);
      }
    }
    if (!failsafe_intervention && !s->is_safe) { failsafe_intervention = 1; s->failsafe_interventions++; }
    /* _control_human (682-686): super + sim.forward() + the two hand mocap bodies */
    double mp[3], mq[4];
    const double* qh;
    human_control_all(B, gid, s, NULL, NULL, hm, mp, mq, &qh);
    human_fk(m, mp, mq, qh, &hk, s->human_site);
    hammer_mocap(m, s, hm, &hk);
    hammer_geo G;
    hammer_geometry(m, hm, &G);
    contact_t con[HRG_NCON_MAX];
    int ncon = collide_hammer(m, &k, &hk, hm, &G, con);
    for (int c = 0; c < ncon; c++) { /* diagnostic: who touches the nail head */
      const int gn = GEOM_BOX + HRG_HG_NAIL, o = con[c].g1 == gn ? con[c].g2 : (con[c].g2 == gn ? con[c].g1 : -1);
      if (o >= 0 && o != GEOM_BOX + HRG_HG_BOARD) hm->nail_touch |= (o == GEOM_BOX + HRG_HG_HANDLE || o == GEOM_BOX + HRG_HG_HEAD) ? 1 : 2;
    }
    double rc[HRG_NRCAP][3], Rb[9];
    quat2mat(Rb, m->base_quat);
    for (int c = 0; c < HRG_NRCAP; c++) {
      int b = m->rcap_body[c];
      double t[3], mid[3];
      for (int a = 0; a < 3; a++) mid[a] = 0.5 * (m->rcap_p1[c][a] + m->rcap_p2[c][a]);
      m3mulv(t, b < 0 ? Rb : k.R[b], mid);
      v3add(rc[c], b < 0 ? m->base_pos : k.p[b], t);
    }
    classify(m, &k, s, con, ncon, rc, &has_collision, &collision_type);
    s->ncon = ncon;
    for (int c = 0; c < HRG_NCON_MAX; c++) { s->con_pairs[c][0] = c < ncon ? con[c].g1 : -1; s->con_pairs[c][1] = c < ncon ? con[c].g2 : -1; }
    /* ---- sim.step(): robot tree | board + nail | hammer ---- */
    double LM[NV * NV], a0[NVH], qd[NVH], frc[NV], Mt[NVH * NVH], M8[64], f8[8];
    memcpy(LM, M, sizeof LM);
    if (!chol(LM, NV)) { crash = 1; break; }
    memset(Mt, 0, sizeof Mt);
    memset(a0, 0, sizeof a0);
    memset(qd, 0, sizeof qd);
    for (int i = 0; i < NV; i++) {
      double act = ctrl[i];
      if (i >= NARM) act = clampd(m->finger_kp * (ctrl[i] - s->qpos[i]), m->finger_forcerange[0], m->finger_forcerange[1]);
      frc[i] = act - m->jnt_damping[i] * s->qvel[i] - bias[i];
      a0[i] = frc[i];
      qd[i] = s->qvel[i];
    }
    chol_solve(LM, NV, a0);
    for (int i = 0; i < NV; i++) for (int j = 0; j < NV; j++) Mt[i * NVH + j] = M[i * NV + j];
    hammer_board_block(m, hm, &G, M8, f8);
    for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) Mt[(HM_OB + i) * NVH + HM_OB + j] = M8[8 * i + j];
    {
      double L8[64], x8[8];
      memcpy(L8, M8, sizeof L8); memcpy(x8, f8, sizeof x8);
      if (!chol(L8, 8)) { crash = 1; break; }
      chol_solve(L8, 8, x8);
      for (int i = 0; i < 8; i++) a0[HM_OB + i] = x8[i];
    }
    { /* the hammer: free body, M = blockdiag(m 1, R diag(I) R'), gyroscopic torque -(w x I w), gravity */
      const double* R = G.R[1];
      double Iw[9], w[3] = {hm->vel[1][3], hm->vel[1][4], hm->vel[1][5]}, Lw[3], tau[3], tl[3];
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double t = 0; for (int kk = 0; kk < 3; kk++) t += R[3 * i + kk] * m->hm_hammer_inertia[kk] * R[3 * j + kk]; Iw[3 * i + j] = t; }
      m3mulv(Lw, Iw, w);
      v3cross(tau, Lw, w);
      for (int kk = 0; kk < 3; kk++) tl[kk] = (R[kk] * tau[0] + R[3 + kk] * tau[1] + R[6 + kk] * tau[2]) / m->hm_hammer_inertia[kk];
      for (int a = 0; a < 3; a++) {
        Mt[(HM_OH + a) * NVH + HM_OH + a] = m->hm_hammer_mass;
        for (int b_ = 0; b_ < 3; b_++) Mt[(HM_OH + 3 + a) * NVH + HM_OH + 3 + b_] = Iw[3 * a + b_];
        a0[HM_OH + a] = m->gravity[a];
        a0[HM_OH + 3 + a] = R[3 * a] * tl[0] + R[3 * a + 1] * tl[1] + R[3 * a + 2] * tl[2];
      }
      Mt[(HM_OH + 6) * NVH + HM_OH + 6] = 1.0; Mt[(HM_OH + 7) * NVH + HM_OH + 7] = 1.0;
    }
    for (int a = 0; a < 6; a++) { qd[HM_OB + a] = hm->vel[0][a]; qd[HM_OH + a] = hm->vel[1][a]; }
    qd[HM_ON] = hm->nail_v;
    efc_t* E = (efc_t*)malloc(sizeof(efc_t));
    E->n = 0;
    E->nv = NVH;
    for (int i = 0; i < NV; i++)
      if (m->jnt_frictionloss[i] > 0) { double J[NVMAX] = {0}; J[i] = 1; efc_add(m, E, J, qd, ROW_FRICTION, 0, 0, m->jnt_frictionloss[i], m->dof_invweight0[i]); }
    if (m->hm_nail_frictionloss > 0) { /* nail_head_joint0: frictionloss with its own solreffriction (nail.xml:7) */
      double J[NVMAX] = {0}; J[HM_ON] = 1;
      efc_add_b(m, E, J, qd, ROW_FRICTION, 0, 0, m->hm_nail_frictionloss, m->hm_nail_invweight, m->hm_nail_fric_damping / m->solimp[1]);
    }
    for (int i = 0; i < NV; i++) {
      double dlo = s->qpos[i] - m->jnt_range[i][0], dhi = m->jnt_range[i][1] - s->qpos[i];
      if (dlo < 0) { double J[NVMAX] = {0}; J[i] = 1; efc_add(m, E, J, qd, ROW_UNILATERAL, dlo, 0, 0, m->dof_invweight0[i]); }
      if (dhi < 0) { double J[NVMAX] = {0}; J[i] = -1; efc_add(m, E, J, qd, ROW_UNILATERAL, dhi, 0, 0, m->dof_invweight0[i]); }
    }
    { /* the slide joint's range [0, hm_nail_range] */
      double dlo = hm->nail_q, dhi = m->hm_nail_range - hm->nail_q;
      if (dlo < 0) { double J[NVMAX] = {0}; J[HM_ON] = 1; efc_add(m, E, J, qd, ROW_UNILATERAL, dlo, 0, 0, m->hm_nail_invweight); }
      if (dhi < 0) { double J[NVMAX] = {0}; J[HM_ON] = -1; efc_add(m, E, J, qd, ROW_UNILATERAL, dhi, 0, 0, m->hm_nail_invweight); }
    }
    for (int c = 0; c < ncon && c < HRG_NCON_DYN_HAMMER; c++) {
      const double* n = con[c].n;
      double t1[3], t2[3], e1[3] = {1, 0, 0}, e2[3] = {0, 1, 0};
      v3cross(t1, n, fabs(n[0]) < 0.5 ? e1 : e2);
      v3scl(t1, t1, 1.0 / v3norm(t1));
      v3cross(t2, n, t1);
      double margin = (con[c].g2 >= GEOM_HUMAN0 && con[c].g2 < GEOM_TABLE) || (con[c].g1 >= GEOM_HUMAN0 && con[c].g1 < GEOM_TABLE) ? m->contact_margin_human : 0.0; /* a human geom on either side */
      for (int d = 0; d < 4; d++) {
        double dir[3], J[NVMAX] = {0};
        const double* tt = d < 2 ? t1 : t2;
        double sg = (d & 1) ? -1.0 : 1.0;
        for (int a = 0; a < 3; a++) dir[a] = n[a] + sg * m->friction_static * tt[a];
        if (con[c].b1 >= 0 && con[c].b1 < NV) robot_point_jac(m, &k, con[c].b1, con[c].pos, dir, -1.0, J);
        if (con[c].b2 >= 0 && con[c].b2 < NV) robot_point_jac(m, &k, con[c].b2, con[c].pos, dir, +1.0, J);
        double diag = (con[c].b1 >= 0 && con[c].b1 < NV ? m->body_invweight0[con[c].b1] : 0.0) + (con[c].b2 >= 0 && con[c].b2 < NV ? m->body_invweight0[con[c].b2] : 0.0);
        for (int side = 0; side < 2; side++) { /* free bodies: J = +-dir . (v + w x r); the nail's point moves with the board and along the slide axis */
          const int body = side ? con[c].b2 : con[c].b1;
          if (body < BODY_BOX) continue;
          const int fb = body - BODY_BOX, o = fb == HRG_HM_HAMMER ? HM_OH : HM_OB;
          const double sgn = side ? 1.0 : -1.0;
          double r[3], rxd[3];
          v3sub(r, con[c].pos, hm->pos[fb == HRG_HM_HAMMER ? 1 : 0]);
          v3cross(rxd, r, dir);
          for (int a = 0; a < 3; a++) { J[o + a] = sgn * dir[a]; J[o + 3 + a] = sgn * rxd[a]; }
          if (fb == HRG_HM_NAIL) { J[HM_ON] = sgn * v3dot(dir, G.axis); diag += m->hm_nail_invweight; }
          else diag += 1.0 / (fb == HRG_HM_HAMMER ? m->hm_hammer_mass : m->hm_board_mass);
        }
        const int n_before = E->n;
        efc_add(m, E, J, qd, ROW_UNILATERAL, con[c].dist, margin, 0, diag * (1.0 + m->friction_static * m->friction_static));
        if (E->n > n_before) E->grp[n_before] = 4 * c + d;
      }
    }
    { /* lh_eq: connect(lh_grip, lh_mocap); rh_eq: weld(rh_grip, rh_mocap) (1100-1145); both stay active (the switch to rh_backup_eq is commented out, 650) */
      const double* R = G.R[0];
      for (int hd = 0; hd < 2; hd++) {
        double rr[3], pt[3];
        m3mulv(rr, R, m->hm_anchor[hd]);
        v3add(pt, hm->pos[0], rr);
        for (int a = 0; a < 3; a++) {
          double J[NVMAX] = {0}, ea[3] = {a == 0, a == 1, a == 2}, rxe[3];
          v3cross(rxe, rr, ea);
          J[HM_OB + a] = 1;
          for (int b_ = 0; b_ < 3; b_++) J[HM_OB + 3 + b_] = rxe[b_];
          efc_add(m, E, J, qd, ROW_EQUALITY, pt[a] - hm->mocap_pos[hd][a], 0, 0, 1.0 / m->hm_board_mass);
        }
      }
      double qt[4], qe[4], erot[3];
      hammer_weld_quat(m, hm, qt);
      const double qc[4] = {qt[0], -qt[1], -qt[2], -qt[3]};
      quatmul(qe, hm->quat[0], qc);
      if (qe[0] < 0) for (int a = 0; a < 4; a++) qe[a] = -qe[a];
      const double sn = sqrt(qe[1] * qe[1] + qe[2] * qe[2] + qe[3] * qe[3]), ang = 2.0 * atan2(sn, qe[0]);
      for (int a = 0; a < 3; a++) erot[a] = sn > 1e-12 ? qe[1 + a] / sn * ang : 0.0;
      for (int a = 0; a < 3; a++) { double J[NVMAX] = {0}; J[HM_OB + 3 + a] = 1; efc_add(m, E, J, qd, ROW_EQUALITY, erot[a], 0, 0, m->hm_board_invweight_rot); }
    }
    double qacc[NVH];
    memset(qacc, 0, sizeof qacc);
    memcpy(qacc, s->qacc_warmstart, sizeof(double) * NV);
    for (int a = 0; a < 6; a++) { qacc[HM_OB + a] = hm->acc_warmstart[0][a]; qacc[HM_OH + a] = hm->acc_warmstart[1][a]; }
    qacc[HM_ON] = hm->nail_acc_warmstart;
    solve(m, Mt, a0, E, qacc);
    noslip(m, Mt, E, qacc); /* opt.noslip_iterations = 20 (1161) */
    if (g_debug && (ncon > 0 || g_debug > 1)) {
      double mx = 0; for (int i = 0; i < NVH; i++) if (fabs(qacc[i]) > mx) mx = fabs(qacc[i]);
      fprintf(stderr, "[oracle hammer] env %d cyc %d ncon %d nefc %d max|qacc| %.3e", e, cyc, ncon, E->n, mx);
      for (int c = 0; c < ncon; c++) fprintf(stderr, " (%d,%d d=%.5f)", con[c].g1, con[c].g2, con[c].dist);
      fprintf(stderr, "\n");
    }
    free(E);
    for (int i = 0; i < NVH; i++) if (!(fabs(qacc[i]) < 1e10)) crash = 1;
    if (crash) break;
    memcpy(s->qacc_warmstart, qacc, sizeof(double) * NV);
    double Mh[NV * NV], rhs[NV];
    memcpy(Mh, M, sizeof Mh);
    for (int i = 0; i < NV; i++) { Mh[i * NV + i] += h * m->jnt_damping[i]; double t = 0; for (int j = 0; j < NV; j++) t += M[i * NV + j] * qacc[j]; rhs[i] = t; }
    if (!chol(Mh, NV)) { crash = 1; break; }
    chol_solve(Mh, NV, rhs);
    for (int i = 0; i < NV; i++) { s->qvel[i] += h * rhs[i]; s->qpos[i] += h * s->qvel[i]; }
    hammer_obs_pos(m, hm); /* body_xpos of the forward pass inside mj_step (pre-integration) */
    for (int fb = 0; fb < 2; fb++) {
      const int o = fb ? HM_OH : HM_OB;
      for (int a = 0; a < 6; a++) { hm->acc_warmstart[fb][a] = qacc[o + a]; hm->vel[fb][a] += h * qacc[o + a]; }
      for (int a = 0; a < 3; a++) hm->pos[fb][a] += h * hm->vel[fb][a];
      double w[3] = {hm->vel[fb][3], hm->vel[fb][4], hm->vel[fb][5]}, wn = v3norm(w), ang = h * wn;
      if (wn > 1e-12) {
        double sh = sin(0.5 * ang) / wn, dq[4] = {cos(0.5 * ang), w[0] * sh, w[1] * sh, w[2] * sh}, qn[4];
        quatmul(qn, dq, hm->quat[fb]);
        double nn = sqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
        for (int a = 0; a < 4; a++) hm->quat[fb][a] = qn[a] / nn;
      }
    }
    if (g_debug > 2) fprintf(stderr, "[nail] cyc %d qacc %.4e v %.4e q %.4e ncon %d nefc %d\n", cyc, qacc[HM_ON], hm->nail_v, hm->nail_q, ncon, 0);
    hm->nail_acc_warmstart = qacc[HM_ON];
    hm->nail_v += h * qacc[HM_ON];
    hm->nail_q += h * hm->nail_v;
    s->time += h;
    eef_of(m, &k, s->eef_pos);
    s->low_level_time += 1;
  }
  /* ---- observation, success, info, reward, done ---- */
  compute_obs_hammer(B, s, hm, term_obs);
  const double progress = clampd(hm->nail_q / m->hm_nail_range, 0.0, 1.0); /* nail_hammering_progress (1291-1303) */
  const int hammered_in = 1.0 - progress < m->hm_goal_tolerance; /* _check_nail_hammered_in (505-520) */
  const int goal_reached = !crash && hm->task_phase == HRG_HM_COMPLETE; /* _check_success (576-587) */
  double r;
  if (goal_reached) r = m->task_reward; /* _sparse_reward (522-556) */
  else {
    r = hammered_in ? m->nail_hammered_in_reward : -1.0;
    if (hm->gripped) r += m->hammer_gripped_reward_bonus;
  }
  if (goal_reached) s->n_goal_reached++;
  int illegal = (collision_type & (HRG_COL_STATIC | HRG_COL_ROBOT | HRG_COL_HUMAN_CRIT)) != 0;
  if (m->reward_shaping) r += 1.0 + 0.0; /* _dense_reward is a TODO returning 0 (558-574) */
  if (illegal) r += m->collision_reward;
  r *= m->reward_scale;
  int d = 0;
  if (crash) { r += m->sim_crash_reward; d = 1; }
  else {
    if (m->done_at_collision && illegal) d = 1;
    if (m->done_at_success && goal_reached) d = 1;
  }
  int ncoll = s->n_collisions_static + s->n_collisions_robot + s->n_collisions_human + s->n_collisions_critical;
  info[HRG_INFO_COLLISION] = has_collision;
  info[HRG_INFO_COLLISION_TYPE] = collision_type;
  info[HRG_INFO_N_COLLISIONS] = ncoll;
  info[HRG_INFO_N_COLLISIONS_STATIC] = s->n_collisions_static;
  info[HRG_INFO_N_COLLISIONS_ROBOT] = s->n_collisions_robot;
  info[HRG_INFO_N_COLLISIONS_HUMAN] = s->n_collisions_human;
  info[HRG_INFO_N_COLLISIONS_CRITICAL] = s->n_collisions_critical;
  info[HRG_INFO_TIMEOUT] = s->timestep >= m->horizon;
  info[HRG_INFO_FAILSAFE_INTERVENTIONS] = s->failsafe_interventions;
  info[HRG_INFO_N_GOAL_REACHED] = s->n_goal_reached;
  info[HRG_INFO_SIM_CRASH] = crash;
  info[HRG_INFO_TRUNCATED] = 0;
  info[HRG_INFO_ACTION_RESAMPLES] = s->action_resamples;
  info[HRG_INFO_N_OBJECT_HANDED_OVER] = 0;
  /* ---- CollaborativeHammeringCart.step tail (490-503) ---- */
  if (goal_reached && !m->done_at_success && !d) { /* _on_goal_reached (749-767): next nail placement, board and nail reset, next animation */
    hm->nail_index = (hm->nail_index + 1) % m->n_obj_placements;
    s->anim_index = (s->anim_index + 1) % m->n_anim_ids;
    s->animation_time = 0;
    s->anim_start_time = (int)((double)s->low_level_time / m->anim_step_length);
    hm->task_phase = HRG_HM_APPROACH; hm->n_delayed = 0;
    human_kin hk2;
    double mp[3], mq[4];
    const double* qh;
    human_control_all(B, gid, s, NULL, NULL, hm, mp, mq, &qh);
    human_fk(m, mp, mq, qh, &hk2, s->human_site);
    hammer_mocap(m, s, hm, &hk2);
    hm->task_phase = HRG_HM_APPROACH; hm->n_delayed = 0; /* _reset_animation (764-767, 770-774) */
    hammer_take_board(B, gid, s, hm);
  }
  if (!d && hm->task_phase == HRG_HM_PRESENT && hammered_in) hm->task_phase = HRG_HM_RETREAT; /* 496-501 */
  if (s->timestep >= m->horizon) { info[HRG_INFO_TRUNCATED] = !d; d = 1; }
  *reward = (float)r;
  *done = (uint8_t)d;
  if (d) env_reset(B, e, obs);
  else memcpy(obs, term_obs, sizeof(float) * HRG_OBS_DIM);
}

/* =============================================================================================== API */
int hrgo_create(const hrg_model_desc* desc, const hrg_clip_table* clips, int32_t n_envs, int64_t env_id0, hrgo_batch** out) {
  hrgo_batch* B = (hrgo_batch*)calloc(1, sizeof *B);
  B->m = *desc;
  B->clips = *clips;
  B->frames = (double*)malloc(sizeof(double) * HRG_FRAME_DIM * (size_t)clips->total_frames);
  memcpy(B->frames, clips->frames, sizeof(double) * HRG_FRAME_DIM * (size_t)clips->total_frames);
  B->clips.frames = B->frames;
  if (desc->robot_hulls) { /* own copy of the hull vertices (the caller's table need not outlive the call) */
    const size_t nb = sizeof(double) * 3 * (size_t)desc->hull_off[HRG_NHULL];
    B->hull_verts = (double*)malloc(nb);
    memcpy(B->hull_verts, desc->hull_verts, nb);
    B->m.hull_verts = B->hull_verts;
  }
  B->n_envs = n_envs;
  B->env_id0 = env_id0;
  B->st = (hrg_env_state*)calloc((size_t)n_envs, sizeof(hrg_env_state));
  B->box = (hrg_box_state*)calloc((size_t)n_envs, sizeof(hrg_box_state));
  B->stk = (hrg_stack_state*)calloc((size_t)n_envs, sizeof(hrg_stack_state));
  B->hmr = (hrg_hammer_state*)calloc((size_t)n_envs, sizeof(hrg_hammer_state));
  B->rcaps = calloc((size_t)n_envs, sizeof *B->rcaps);
  B->hcaps = calloc((size_t)n_envs, sizeof *B->hcaps);
  B->n_hcaps = calloc((size_t)n_envs, sizeof(int32_t));
  for (int e = 0; e < n_envs; e++) B->st[e].episode = -1;
  *out = B;
  return 0;
}
void hrgo_destroy(hrgo_batch* B) {
  if (!B) return;
  free(B->frames); free(B->hull_verts); free(B->st); free(B->box); free(B->stk); free(B->hmr); free(B->rcaps); free(B->hcaps); free(B->n_hcaps); free(B);
}
int hrgo_reset(hrgo_batch* B, const uint8_t* mask, float* obs) {
  for (int e = 0; e < B->n_envs; e++) if (!mask || mask[e]) env_reset(B, e, obs + (size_t)e * HRG_OBS_DIM);
  return 0;
}
int hrgo_step(hrgo_batch* B, double* actions, float* obs, float* term_obs, float* reward, uint8_t* done, int32_t* info) {
  for (int e = 0; e < B->n_envs; e++) {
    float tmp[HRG_OBS_DIM];
    env_step(B, e, actions + (size_t)e * HRG_ACT_DIM, obs + (size_t)e * HRG_OBS_DIM, term_obs ? term_obs + (size_t)e * HRG_OBS_DIM : tmp,
             reward + e, done + e, info + (size_t)e * HRG_INFO_DIM);
  }
  return 0;
}
/* step a sub-range only (CPU baseline workers) */
int hrgo_step_range(hrgo_batch* B, int e0, int e1, double* actions, float* obs, float* reward, uint8_t* done, int32_t* info) {
  for (int e = e0; e < e1; e++) {
    float tmp[HRG_OBS_DIM];
    env_step(B, e, actions + (size_t)e * HRG_ACT_DIM, obs + (size_t)e * HRG_OBS_DIM, tmp, reward + e, done + e, info + (size_t)e * HRG_INFO_DIM);
  }
  return 0;
}
/* CPU-baseline harness (bench.py's cpu_baseline leg): the shape of the reference's SubprocVecEnv (utils/env_util_SB3.py:75-87) -- P workers stepping
 * the envs of a vec-step, synchronised once per vec-step.  n_steps vec-steps with actions cycled from a pool [n_pool][n_envs][HRG_ACT_DIM];
 * cpus (or NULL) pins worker w to logical cpu cpus[w] (one per physical core).
 * chunk = 0: worker w owns the fixed env range [n w / P, n (w + 1) / P) (one env process per worker, as SubprocVecEnv has it).
 * chunk > 0: the workers of a vec-step draw runs of `chunk` envs from a shared counter until the step is used up -- the per-env cost depends on the
 *            env's state (contacts, fail-safe manoeuvres: 2 - 3 x), and on a shared host a pinned worker may lose its core for a while; with fixed ranges
 *            every vec-step then waits for its unluckiest worker.  Results do not depend on who steps an env (envs never interact).
 * busy_s (or NULL): per worker, the seconds it spent inside env_step (the rest of the wall time it waited at the barrier). */
typedef struct {
  hrgo_batch* B; int w, nw, n_steps, n_pool, chunk; const double* pool; double* actions; float* obs; float* term_obs; float* reward; uint8_t* done; int32_t* info;
  pthread_barrier_t* bar; int cpu; int* next; double busy;
} hrgo_worker_t;
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
static void* hrgo_worker(void* arg) {
  hrgo_worker_t* W = (hrgo_worker_t*)arg;
  if (W->cpu >= 0) { cpu_set_t set; CPU_ZERO(&set); CPU_SET(W->cpu, &set); pthread_setaffinity_np(pthread_self(), sizeof set, &set); }
  const int n = W->B->n_envs;
  double act[HRG_ACT_DIM];
  for (int k = 0; k < W->n_steps; k++) {
    const double* a = W->pool ? W->pool + (size_t)(k % W->n_pool) * n * HRG_ACT_DIM : NULL;
    const double t0 = now_s();
    for (;;) {
      int e0, e1;
      if (W->chunk > 0) { e0 = __atomic_fetch_add(&W->next[k], W->chunk, __ATOMIC_RELAXED); e1 = e0 + W->chunk < n ? e0 + W->chunk : n; }
      else { e0 = (int)((long long)n * W->w / W->nw); e1 = (int)((long long)n * (W->w + 1) / W->nw); }
      for (int e = e0; e < e1; e++) {
        float tmp[HRG_OBS_DIM];
        double* ap = act;
        if (a) memcpy(act, a + (size_t)e * HRG_ACT_DIM, sizeof act); /* the wrappers rewrite the action row in place: work on a copy of the pool */
        else ap = W->actions + (size_t)e * HRG_ACT_DIM;              /* a caller's own action block (one vec-step): rewritten in place like hrgo_step does */
        env_step(W->B, e, ap, W->obs + (size_t)e * HRG_OBS_DIM, W->term_obs ? W->term_obs + (size_t)e * HRG_OBS_DIM : tmp, W->reward + e, W->done + e,
                 W->info + (size_t)e * HRG_INFO_DIM);
      }
      if (W->chunk <= 0 || e1 >= n) break;
    }
    W->busy += now_s() - t0;
    pthread_barrier_wait(W->bar); /* VecEnv.step_wait: every worker has finished the step */
  }
  return NULL;
}
static int run_workers(hrgo_batch* B, int n_workers, const int32_t* cpus, int n_steps, const double* pool, int n_pool, double* actions, int chunk, float* obs, float* term_obs,
                       float* reward, uint8_t* done, int32_t* info, double* busy_s) {
  if (n_workers < 1 || n_workers > 1024 || n_pool < 1 || n_steps < 0) return -1;
  pthread_t th[1024];
  hrgo_worker_t* W = (hrgo_worker_t*)calloc((size_t)n_workers, sizeof *W);
  int* next = (int*)calloc((size_t)n_steps + 1, sizeof(int));
  pthread_barrier_t bar;
  pthread_barrier_init(&bar, NULL, (unsigned)n_workers);
  int started = 0, rc = 0;
  for (int w = 0; w < n_workers; w++) {
    W[w] = (hrgo_worker_t){B, w, n_workers, n_steps, n_pool, chunk, pool, actions, obs, term_obs, reward, done, info, &bar, cpus ? cpus[w] : -1, next, 0.0};
    if (pthread_create(&th[w], NULL, hrgo_worker, &W[w]) != 0) { rc = -2; break; }
    started++;
  }
  if (rc != 0) { /* a team that is short of a member would wait at the barrier for ever: the members that did start are cancelled */
    for (int w = 0; w < started; w++) pthread_cancel(th[w]);
  }
  for (int w = 0; w < started; w++) pthread_join(th[w], NULL);
  if (busy_s) for (int w = 0; w < n_workers; w++) busy_s[w] = W[w].busy;
  pthread_barrier_destroy(&bar);
  free(next); free(W);
  return rc;
}
int hrgo_rollout_parallel(hrgo_batch* B, int n_workers, const int32_t* cpus, int n_steps, const double* pool, int n_pool, float* obs, float* reward, uint8_t* done, int32_t* info) {
  return run_workers(B, n_workers, cpus, n_steps, pool, n_pool, NULL, 0, obs, NULL, reward, done, info, NULL);
}
int hrgo_rollout_parallel2(hrgo_batch* B, int n_workers, const int32_t* cpus, int n_steps, const double* pool, int n_pool, int chunk, float* obs, float* reward, uint8_t* done,
                           int32_t* info, double* busy_s) {
  return run_workers(B, n_workers, cpus, n_steps, pool, n_pool, NULL, chunk, obs, NULL, reward, done, info, busy_s);
}
/* one vec-step of the whole batch on n_workers threads (what hrgo_step does on one): the parity tests at the benchmark's batch sizes */
int hrgo_step_parallel(hrgo_batch* B, int n_workers, double* actions, float* obs, float* term_obs, float* reward, uint8_t* done, int32_t* info) {
  return run_workers(B, n_workers, NULL, 1, NULL, 1, actions, 8, obs, term_obs, reward, done, info, NULL);
}
/* all state blocks at once (arrays of n_envs blocks; a NULL pointer skips that kind): copying a HIP batch's state into the checker and back */
int hrgo_get_states(hrgo_batch* B, hrg_env_state* st, hrg_box_state* box, hrg_stack_state* stk, hrg_hammer_state* hmr) {
  const size_t n = (size_t)B->n_envs;
  if (st) memcpy(st, B->st, n * sizeof *st);
  if (box) memcpy(box, B->box, n * sizeof *box);
  if (stk) memcpy(stk, B->stk, n * sizeof *stk);
  if (hmr) memcpy(hmr, B->hmr, n * sizeof *hmr);
  return 0;
}
int hrgo_set_states(hrgo_batch* B, const hrg_env_state* st, const hrg_box_state* box, const hrg_stack_state* stk, const hrg_hammer_state* hmr) {
  const size_t n = (size_t)B->n_envs;
  if (st) memcpy(B->st, st, n * sizeof *st);
  if (box) memcpy(B->box, box, n * sizeof *box);
  if (stk) memcpy(B->stk, stk, n * sizeof *stk);
  if (hmr) memcpy(B->hmr, hmr, n * sizeof *hmr);
  return 0;
}
int hrgo_get_state(hrgo_batch* B, int e, void* buf, size_t bytes) {
  if (bytes != sizeof(hrg_env_state)) return -1;
  memcpy(buf, &B->st[e], bytes);
  return 0;
}
int hrgo_set_state(hrgo_batch* B, int e, const void* buf, size_t bytes) {
  if (bytes != sizeof(hrg_env_state)) return -1;
  memcpy(&B->st[e], buf, bytes);
  return 0;
}
int hrgo_get_box(hrgo_batch* B, int e, void* buf, size_t bytes) {
  if (bytes != sizeof(hrg_box_state)) return -1;
  memcpy(buf, &B->box[e], bytes);
  return 0;
}
int hrgo_set_box(hrgo_batch* B, int e, const void* buf, size_t bytes) {
  if (bytes != sizeof(hrg_box_state)) return -1;
  memcpy(&B->box[e], buf, bytes);
  return 0;
}
int hrgo_get_stack(hrgo_batch* B, int e, void* buf, size_t bytes) {
  if (bytes != sizeof(hrg_stack_state)) return -1;
  memcpy(buf, &B->stk[e], bytes);
  return 0;
}
int hrgo_get_hammer(hrgo_batch* B, int e, void* buf, size_t bytes) {
  if (e < 0 || e >= B->n_envs || bytes != sizeof(hrg_hammer_state)) return -1;
  memcpy(buf, &B->hmr[e], bytes);
  return 0;
}
int hrgo_set_hammer(hrgo_batch* B, int e, const void* buf, size_t bytes) {
  if (e < 0 || e >= B->n_envs || bytes != sizeof(hrg_hammer_state)) return -1;
  memcpy(&B->hmr[e], buf, bytes);
  return 0;
}
int hrgo_set_stack(hrgo_batch* B, int e, const void* buf, size_t bytes) {
  if (bytes != sizeof(hrg_stack_state)) return -1;
  memcpy(&B->stk[e], buf, bytes);
  return 0;
}
size_t hrgo_stack_bytes(void) { return sizeof(hrg_stack_state); }
/* known-answer tap: contacts of two equal boxes (tests/test_stacking.py); out = n x [pos 3, normal 3, dist] */
int hrgo_test_boxbox(const double* pa, const double* qa, const double* pb, const double* qb, const double* half, double* out) {
  double Ra[9], Rb[9];
  bb_contact bc[4];
  quat2mat(Ra, qa); quat2mat(Rb, qb);
  const int n = box_box(pa, Ra, pb, Rb, half, bc);
  for (int q = 0; q < n; q++) { for (int a = 0; a < 3; a++) { out[7 * q + a] = bc[q].pos[a]; out[7 * q + 3 + a] = bc[q].n[a]; } out[7 * q + 6] = bc[q].dist; }
  return n;
}
size_t hrgo_state_bytes(void) { return sizeof(hrg_env_state); }
size_t hrgo_box_bytes(void) { return sizeof(hrg_box_state); }
size_t hrgo_desc_bytes(void) { return sizeof(hrg_model_desc); }
/* HumanEnv.check_collision_action for every env (human_env.py:588-627): goal configuration of the action at the current joint angles -> pre-check model */
/* two boxes with different half extents: out = n x [pos 3 | normal 3 | dist] */
int hrgo_test_boxbox2(const double* pa, const double* qa, const double* ha, const double* pb, const double* qb, const double* hb, double* out) {
  double Ra[9], Rb[9];
  bb_contact bc[4];
  quat2mat(Ra, qa); quat2mat(Rb, qb);
  const int n = box_box2(pa, Ra, ha, pb, Rb, hb, bc);
  for (int z = 0; z < n; z++) { for (int a = 0; a < 3; a++) { out[7 * z + a] = bc[z].pos[a]; out[7 * z + 3 + a] = bc[z].n[a]; } out[7 * z + 6] = bc[z].dist; }
  return n;
}
int hrgo_check_actions(hrgo_batch* B, const double* actions, uint8_t* collides) {
  for (int e = 0; e < B->n_envs; e++) collides[e] = (uint8_t)(action_collides(&B->m, &B->st[e], actions + (size_t)e * HRG_ACT_DIM) != 0);
  return 0;
}
int hrgo_contacts(hrgo_batch* B, int32_t* pairs, int32_t* ncon) {
  for (int e = 0; e < B->n_envs; e++) {
    ncon[e] = B->st[e].ncon;
    memcpy(pairs + (size_t)e * HRG_NCON_MAX * 2, B->st[e].con_pairs, sizeof(int32_t) * HRG_NCON_MAX * 2);
  }
  return 0;
}
int hrgo_capsules(hrgo_batch* B, double* robot, double* human, int32_t* n_human) {
  memcpy(robot, B->rcaps, sizeof(*B->rcaps) * (size_t)B->n_envs);
  memcpy(human, B->hcaps, sizeof(*B->hcaps) * (size_t)B->n_envs);
  memcpy(n_human, B->n_hcaps, sizeof(int32_t) * (size_t)B->n_envs);
  return 0;
}

/* ---- unit-level taps used by tests/ (known-answer tests of the pieces) ---- */
/* hull h of the desc at body pose (R row-major, p): vertex farthest along d -> out[3]; returns its index */
int hrgo_test_hull_support(const hrg_model_desc* m, int h, const double* R, const double* p, const double* d, double* out) {
  const hull_t H = {m->hull_verts + 3 * m->hull_off[h], m->hull_off[h + 1] - m->hull_off[h], R, p};
  return hull_support(&H, d, out);
}
/* ... its lowest point over a horizontal plane (hull_lowest) -> out[3] */
void hrgo_test_hull_lowest(const hrg_model_desc* m, int h, const double* R, const double* p, double* out) {
  const hull_t H = {m->hull_verts + 3 * m->hull_off[h], m->hull_off[h + 1] - m->hull_off[h], R, p};
  hull_lowest(&H, out);
}
/* ... and its GJK distance to the segment [s1, s2]: out = [distance, witness on the hull 3, witness on the segment 3] */
void hrgo_test_hull_segment(const hrg_model_desc* m, int h, const double* R, const double* p, const double* s1, const double* s2, double* out) {
  const hull_t H = {m->hull_verts + 3 * m->hull_off[h], m->hull_off[h + 1] - m->hull_off[h], R, p};
  out[0] = gjk_hull_segment(&H, s1, s2, out + 1, out + 4);
}
void hrgo_test_robot(const hrg_model_desc* m, const double* q, const double* qd, double* M, double* bias, double* eef) {
  robot_kin k;
  robot_fk(m, q, &k);
  robot_crba(m, &k, M);
  robot_bias(m, &k, qd, bias);
  eef_of(m, &k, eef);
}
double hrgo_test_segseg(const double* p1, const double* q1, const double* p2, const double* q2, double* c1, double* c2) { return seg_seg(p1, q1, p2, q2, c1, c2); }
void hrgo_test_ltt(const hrg_model_desc* m, const double* q0, const double* v0, const double* a0, const double* goal, hrg_ltt* L) { ltt_plan(m, L, q0, v0, a0, goal); }
void hrgo_test_ltt_eval(const hrg_ltt* L, int j, double s, double* out3) { ltt_eval(L, j, s, out3, out3 + 1, out3 + 2); }
void hrgo_test_path_eval(const hrg_path* P, double t, double ve, double* out3) { path_eval(P, t, ve, out3, out3 + 1, out3 + 2); }
void hrgo_test_path(double s0, double v0, double a0, double ve, double amax, double jmax, double t, double* out4) {
  hrg_path P;
  path_plan(&P, s0, v0, a0, ve, amax, jmax);
  path_eval(&P, t, ve, out4, out4 + 1, out4 + 2);
  out4[3] = path_total(&P);
}
void hrgo_test_human_fk(const hrg_model_desc* m, const double* mp, const double* mq, const double* qh, double* sites, double* caps) {
  human_kin h;
  human_fk(m, mp, mq, qh, &h, (double(*)[3])sites);
  for (int b = 0; b < HRG_NHB; b++) { memcpy(caps + 6 * b, h.cap1[b], 24); memcpy(caps + 6 * b + 3, h.cap2[b], 24); }
}
int hrgo_test_config_collides(const hrg_model_desc* m, const double* q6) { return config_collides(m, q6); }
double hrgo_test_u01(uint64_t seed, uint64_t env, uint64_t ep, uint64_t stream, uint64_t idx) { return rng_u01(seed, env, ep, stream, idx); }

/* =============================================================================================== human dynamics (study)
 * Articulated-body forward dynamics of the 23 x 3-hinge human tree hanging off the (static) mocap pelvis, world frame
 * about the world origin.  In the reference these 69 hinges are dynamic DoFs whose qpos is overwritten every substep
 * (human_env.py:1766-1767) while qvel integrates freely.  Used by hrgo_test_human_dyn to study how
 * that velocity drift behaves; not wired into env_step yet. */
static void m6_mulv(double* r, const double* A, const double* x) {
  for (int i = 0; i < 6; i++) { double t = 0; for (int j = 0; j < 6; j++) t += A[6 * i + j] * x[j]; r[i] = t; }
}
static void crm6(double* r, const double* v, const double* m_) { /* motion cross product */
  double a[3], b[3], c[3];
  v3cross(a, v, m_); v3cross(b, v, m_ + 3); v3cross(c, v + 3, m_);
  for (int i = 0; i < 3; i++) { r[i] = a[i]; r[3 + i] = b[i] + c[i]; }
}
static void crf6(double* r, const double* v, const double* f) { /* force cross product */
  double a[3], b[3], c[3];
  v3cross(a, v, f); v3cross(b, v + 3, f + 3); v3cross(c, v, f + 3);
  for (int i = 0; i < 3; i++) { r[i] = a[i] + b[i]; r[3 + i] = c[i]; }
}
static void human_aba(const hrg_model_desc* m, const human_kin* h, const double* qh, const double* qd, double armature, double* qacc) {
  static const double ez[3] = {0, 0, 1}, ey[3] = {0, 1, 0}, ex[3] = {1, 0, 0};
  double S[HRG_NHB][3][6], v[HRG_NHB][6], c[HRG_NHB][6], IA[HRG_NHB][36], pA[HRG_NHB][6], U[HRG_NHB][3][6], Di[HRG_NHB][9], u[HRG_NHB][3], a[HRG_NHB][6];
  memset(v[0], 0, sizeof v[0]);
  for (int b = 1; b < HRG_NHB; b++) {
    int par = m->hb_parent[b];
    double anc[3], t[3], R1[9], R2[9], Rz[9], Ry[9], w[3][3];
    m3mulv(t, h->R[par], m->hb_anchor[b]); v3add(anc, h->p[par], t);
    axisangle2mat(Rz, ez, qh[3 * (b - 1)]); axisangle2mat(Ry, ey, qh[3 * (b - 1) + 1]);
    m3mul(R1, h->R[par], Rz); m3mul(R2, R1, Ry);
    m3mulv(w[0], h->R[par], ez); m3mulv(w[1], R1, ey); m3mulv(w[2], R2, ex);
    memcpy(v[b], v[par], sizeof v[b]);
    memset(c[b], 0, sizeof c[b]);
    for (int k = 0; k < 3; k++) {
      v3cpy(S[b][k], w[k]); v3cross(S[b][k] + 3, anc, w[k]);
      double j[6], cr[6];
      for (int i = 0; i < 6; i++) j[i] = S[b][k][i] * qd[3 * (b - 1) + k];
      for (int i = 0; i < 6; i++) v[b][i] += j[i];
      crm6(cr, v[b], j);
      for (int i = 0; i < 6; i++) c[b][i] += cr[i];
    }
    /* spatial inertia about the origin: mass 1 at the body-frame origin, unit isotropic rotational inertia (human.xml:46) */
    const double* cm = h->p[b];
    double mass = 1.0, cc = v3dot(cm, cm), I3[9], hx[9] = {0, -cm[2], cm[1], cm[2], 0, -cm[0], -cm[1], cm[0], 0};
    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) I3[3 * i + k] = (i == k ? 1.0 + mass * cc : 0.0) - mass * cm[i] * cm[k];
    memset(IA[b], 0, sizeof IA[b]);
    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) {
      IA[b][6 * i + k] = I3[3 * i + k];
      IA[b][6 * i + 3 + k] = mass * hx[3 * i + k];
      IA[b][6 * (3 + i) + k] = -mass * hx[3 * i + k];
      IA[b][6 * (3 + i) + 3 + k] = i == k ? mass : 0.0;
    }
    double Iv[6];
    m6_mulv(Iv, IA[b], v[b]);
    crf6(pA[b], v[b], Iv);
  }
  for (int b = HRG_NHB - 1; b >= 1; b--) {
    double D[9];
    for (int k = 0; k < 3; k++) m6_mulv(U[b][k], IA[b], S[b][k]);
    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) { double t = 0; for (int q = 0; q < 6; q++) t += S[b][i][q] * U[b][k][q]; D[3 * i + k] = t + (i == k ? armature : 0.0); }
    double det = D[0] * (D[4] * D[8] - D[5] * D[7]) - D[1] * (D[3] * D[8] - D[5] * D[6]) + D[2] * (D[3] * D[7] - D[4] * D[6]);
    double* X = Di[b];
    X[0] = (D[4] * D[8] - D[5] * D[7]) / det; X[1] = (D[2] * D[7] - D[1] * D[8]) / det; X[2] = (D[1] * D[5] - D[2] * D[4]) / det;
    X[3] = (D[5] * D[6] - D[3] * D[8]) / det; X[4] = (D[0] * D[8] - D[2] * D[6]) / det; X[5] = (D[2] * D[3] - D[0] * D[5]) / det;
    X[6] = (D[3] * D[7] - D[4] * D[6]) / det; X[7] = (D[1] * D[6] - D[0] * D[7]) / det; X[8] = (D[0] * D[4] - D[1] * D[3]) / det;
    for (int k = 0; k < 3; k++) { double t = 0; for (int q = 0; q < 6; q++) t += S[b][k][q] * pA[b][q]; u[b][k] = -t; }
    int par = m->hb_parent[b];
    if (par >= 1) {
      double UD[3][6], Ia[36], pa[6], Iac[6], Du[3];
      for (int k = 0; k < 3; k++) for (int q = 0; q < 6; q++) UD[k][q] = U[b][0][q] * X[0 * 3 + k] + U[b][1][q] * X[1 * 3 + k] + U[b][2][q] * X[2 * 3 + k];
      for (int i = 0; i < 6; i++) for (int q = 0; q < 6; q++) Ia[6 * i + q] = IA[b][6 * i + q] - (UD[0][i] * U[b][0][q] + UD[1][i] * U[b][1][q] + UD[2][i] * U[b][2][q]);
      m6_mulv(Iac, Ia, c[b]);
      for (int k = 0; k < 3; k++) Du[k] = X[3 * k] * u[b][0] + X[3 * k + 1] * u[b][1] + X[3 * k + 2] * u[b][2];
      for (int q = 0; q < 6; q++) pa[q] = pA[b][q] + Iac[q] + U[b][0][q] * Du[0] + U[b][1][q] * Du[1] + U[b][2][q] * Du[2];
      for (int q = 0; q < 36; q++) IA[par][q] += Ia[q];
      for (int q = 0; q < 6; q++) pA[par][q] += pa[q];
    }
  }
  memset(a[0], 0, sizeof a[0]);
  for (int i = 0; i < 3; i++) a[0][3 + i] = -m->gravity[i];
  for (int b = 1; b < HRG_NHB; b++) {
    int par = m->hb_parent[b];
    double ap[6], r[3];
    for (int q = 0; q < 6; q++) ap[q] = a[par][q] + c[b][q];
    for (int k = 0; k < 3; k++) { double t = 0; for (int q = 0; q < 6; q++) t += U[b][k][q] * ap[q]; r[k] = u[b][k] - t; }
    for (int k = 0; k < 3; k++) qacc[3 * (b - 1) + k] = Di[b][3 * k] * r[0] + Di[b][3 * k + 1] * r[1] + Di[b][3 * k + 2] * r[2];
    for (int q = 0; q < 6; q++) a[b][q] = ap[q] + S[b][0][q] * qacc[3 * (b - 1)] + S[b][1][q] * qacc[3 * (b - 1) + 1] + S[b][2][q] * qacc[3 * (b - 1) + 2];
  }
}
/* free evolution of the human joint velocities along one clip: out[t] = max |qvel| after substep t */
void hrgo_test_human_dyn(const hrg_model_desc* m, const double* frames, int n_frames, int n_sub, double armature, double* out_maxvel, double* out_maxacc) {
  double qd[HRG_NHQ] = {0}, qa[HRG_NHQ];
  human_kin h;
  double site[HRG_NHJ][3];
  for (int t = 0; t < n_sub; t++) {
    int fr = (int)floor((double)t / m->anim_step_length) % n_frames;
    const double* f = frames + (size_t)fr * HRG_FRAME_DIM;
    double mp[3] = {f[0], f[1], f[2]}, mq[4] = {f[6], f[3], f[4], f[5]};
    human_fk(m, mp, mq, f + 7, &h, site);
    human_aba(m, &h, f + 7, qd, armature, qa);
    double mv = 0, ma = 0;
    for (int i = 0; i < HRG_NHQ; i++) { qd[i] += m->timestep * qa[i]; if (fabs(qd[i]) > mv) mv = fabs(qd[i]); if (fabs(qa[i]) > ma) ma = fabs(qa[i]); }
    out_maxvel[t] = mv; out_maxacc[t] = ma;
  }
}
/* out: squared distance, closest point on the segment (3), closest point on the cube (3) */
void hrgo_test_segbox(const double* p1, const double* p2, const double* c, const double* quat, const double* hb, double* out) {
  double R[9], t;
  quat2mat(R, quat);
  out[0] = seg_box(p1, p2, c, R, hb, &t, out + 1, out + 4);
}
void hrgo_test_ik(const hrg_model_desc* m, const double* q6, double* act7) {
  hrg_env_state s;
  memset(&s, 0, sizeof s);
  for (int j = 0; j < NARM; j++) s.qpos[j] = q6[j];
  ik_action(m, &s, act7);
}
void hrgo_test_ik_fk(const hrg_model_desc* m, const double* q6, double* pee3, double* R9) {
  double ax[NARM][3], org[NARM][3];
  ik_fk(m, q6, ax, org, R9, pee3);
}
