// hrgym_device.h — device-side building blocks of the batched ReachHuman stepper (gfx950 / CDNA4).
//
// Execution model: ONE WAVEFRONT (64 lanes) PER ENVIRONMENT, one 64-thread workgroup per wavefront.
// The env's resident state block is staged HBM -> LDS once per policy step, all 25 shield cycles run out
// of LDS/registers, and the block is written back once.  Inside a cycle, lanes take different roles:
//   lanes = human bodies (24)        : tree kinematics, parent frames fetched with wave shuffles
//   lanes = human reach capsules (<=64) x loop over 7 robot capsules : swept-capsule verification, __ballot
//   lanes = capsule pairs            : contact broadphase/narrowphase (segment-segment), ballot compaction
//   lanes = joints (6)               : long-term-trajectory planning (bisection), trajectory evaluation
//   lanes = constraint rows (<=64)   : Newton solver rows live in registers, wave reductions for the line search
//   lanes = (i,j) of the 8x8 mass matrix, lanes = kinematic configurations (3 chain FKs at once)
// Small serial pieces (RNEA recursion, 8x8 Cholesky, path profiles) run wave-uniform: every lane computes the
// same value and stores the same value, so no intra-wave hand-off is needed for them.
// No MFMA: the work is tree-structured 3x3/6x6/8x8 FP64, not a dense contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hrgym.h"
#include "../../include/hrgym_state.h"

#define NV HRG_NV
#define NARM HRG_NARM
// HRG_BOX=1 builds the variant with the manipulation object (PickPlaceHumanCart): this file set is compiled a second time
// by hrgym_box.hip into its own kernels, so the ReachHuman kernels carry none of the object's registers, LDS or code.
#ifndef HRG_BOX
#define HRG_BOX 0
#endif
// HRG_HANDOVER=1 (with HRG_BOX=1, hrgym_handover.hip): the cube variant plus what only the handover tasks need -- the object <-> hand
// weld rows, a second physics step per cycle, the hand mocap pose.  A third translation unit, so the pick-place kernels stay lean.
#ifndef HRG_LIFT
#define HRG_LIFT 0   // HRG_LIFT=1 (with HRG_BOX=1, hrgym_lift.hip): CollaborativeLiftingCart -- two connect equalities board <-> hand mocap bodies, the lifting task logic
#endif
#ifndef HRG_HANDOVER
#define HRG_HANDOVER 0
#endif
#ifndef HRG_STACK
#define HRG_STACK 0   // HRG_STACK=1 (hrgym_stack.hip, HRG_BOX=0): CollaborativeStackingCart -- four free cubes, box-box contacts, two welds; its own collision tail and solver
#endif
#ifndef HRG_HULLS
#define HRG_HULLS 0   // HRG_HULLS=1 (hrgym_hulls.hip, ReachHuman): the arm links collide as the convex hulls of their meshes (hrg_model_desc.robot_hulls; hrgym_hull.h)
#endif
#ifndef HRG_HAMMER
#define HRG_HAMMER 0  // HRG_HAMMER=1 (hrgym_hammer.hip, HRG_BOX=0): CollaborativeHammeringCart -- board + nail + hammer, a 24-DoF system in three 8-wide blocks; its own collision tail and solver
#endif
#define HRG_BASE_TU (!HRG_BOX && !HRG_STACK && !HRG_HAMMER && !HRG_HULLS)   // hrgym_hip.hip itself: the ReachHuman kernels, the pre-check kernel and the host side (C ABI)
#define NVT HRG_NVT
#if HRG_HAMMER
#define NVS HRG_NV_HAMMER       // robot tree | board + nail (+ pad) | hammer (+ 2 pads)
#define NCON_DYN HRG_NCON_DYN_HAMMER
#define HM_OB NV                // board DoF
#define HM_ON (NV + 6)          // nail slide joint
#define HM_OH (NV + 8)          // hammer DoF
#elif HRG_STACK
#define NVS HRG_NV_STACK        // robot tree + four free joints
#define NCON_DYN HRG_NCON_DYN_STACK
#define NCUBE HRG_NCUBE
#elif HRG_BOX
#define NVS NVT                 // DoF of the constrained system: robot tree + free joint of the cube
#define NCON_DYN HRG_NCON_DYN_BOX
#else
#define NVS NV
#define NCON_DYN HRG_NCON_DYN
#endif
#define BODY_BOX 100            // body code of the cube in Contact.b1/b2
#define DI __device__ __forceinline__
// Shipping configuration (measured on MI355X, profiles/README.md): every phase inlined into the kernel, 128-VGPR cap (4 waves/SIMD: all 4096 envs of a
// batch resident at once, 16 single-wave workgroups per CU) and the per-cycle "opaque lane id / opaque model pointer" barriers that stop loop-invariant
// hoisting out of the 25-cycle loop.  Phases as real functions ran at the same speed but saved callee-saved VGPRs to scratch on every call (2.3 GB of HBM
// writes per launch instead of 57 MB): the switches that built them are gone, the measurements are in DESIGN.md section 8.
#ifndef HRG_MIN_WAVES
#define HRG_MIN_WAVES 4
#endif
#ifndef HRG_PHASE   // (tools/regs.py builds the phases as real functions to read their register needs one by one)
#define HRG_PHASE __device__ __forceinline__
#endif
#define PH_DYNTERMS HRG_PHASE
#define PH_HUMAN HRG_PHASE
#define PH_SHIELD HRG_PHASE
#define PH_COLLIDE HRG_PHASE
#define PH_CLASSIFY HRG_PHASE
#define PH_DYNSTEP HRG_PHASE
#define HRG_PI 3.14159265358979323846
#define SIXTH (1.0 / 6.0)   // cubic term of the constant-jerk profiles: a product, not an FP64 division per segment

#define GEOM_HUMAN0 HRG_NRCAP
#define GEOM_TABLE (HRG_NRCAP + HRG_NHB)
#define GEOM_FLOOR (GEOM_TABLE + 1)
#define GEOM_BOX (GEOM_FLOOR + 1)   // stacking: GEOM_BOX + c = cube c, BODY_BOX + c its body code

// ---- diagnostic build only (-DHRG_STAMPS): where do the cycles go?  s_memtime deltas per phase, summed over waves.
#ifdef HRG_STAMPS
static __device__ unsigned long long g_stamps[32];
static __device__ unsigned long long g_stamps_slow[34];     // the same sums over the waves that lived longer than g_slow_thresh cycles ([32] = their number, [33] = their total lifetime)
static __device__ unsigned long long g_slow_thresh = ~0ull;
static __device__ unsigned long long g_envcyc[16384][3];   // per env: start, end timestamp of its last step
static __device__ unsigned long long g_envacc[16384][32];  // per env: the phase sums / counters of its last step
__shared__ unsigned long long g_tbeg;
__shared__ unsigned long long g_acc[32];   // per-wave accumulators (diagnostic build only: costs one workgroup of occupancy)
__shared__ unsigned long long g_t0;
#define STAMP_NOW(var) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP_DECL
#define STAMP_INIT(lane) do { if ((lane) < 32) g_acc[lane] = 0; unsigned long long _t; STAMP_NOW(_t); g_t0 = _t; g_tbeg = _t; __syncthreads(); } while (0)
#define STAMP(k) do { unsigned long long _t; STAMP_NOW(_t); if (threadIdx.x == 0) { g_acc[k] += _t - g_t0; g_t0 = _t; } } while (0)
#define STAMP_FLUSH(lane)
#define COUNT(k, n) do { if (threadIdx.x == 0) g_acc[k] += (n); } while (0)
#define STAMP_FINAL(lane) do { __syncthreads(); if ((lane) < 32) atomicAdd(&g_stamps[lane], g_acc[lane]); { unsigned long long _te; STAMP_NOW(_te); if (_te - g_tbeg > g_slow_thresh) { if ((lane) < 32) atomicAdd(&g_stamps_slow[lane], g_acc[lane]); if ((lane) == 32) atomicAdd(&g_stamps_slow[32], 1ull); if ((lane) == 33) atomicAdd(&g_stamps_slow[33], _te - g_tbeg); } } if ((lane) < 32 && blockIdx.x < 16384) g_envacc[blockIdx.x][lane] = g_acc[lane]; if ((lane) == 0 && blockIdx.x < 16384) { unsigned long long _t; STAMP_NOW(_t); g_envcyc[blockIdx.x][0] = g_tbeg; g_envcyc[blockIdx.x][1] = _t; unsigned _hw, _xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(_hw), "=s"(_xcc)); g_envcyc[blockIdx.x][2] = ((unsigned long long)_xcc << 32) | _hw; } } while (0)
#else
#define STAMP_DECL
#define STAMP_INIT(lane)
#define STAMP(k)
#define STAMP_FLUSH(lane)
#define COUNT(k, n)
#define STAMP_FINAL(lane)
#endif

// ------------------------------------------------------------------------------------------------ model
struct DevModel {
  hrg_model_desc m;
  double Rbase[9];
  double Rq[NV][9];          // body_quat as matrices
  int32_t anc_mask[NV];      // bit i set: body i is an ancestor-or-self of body j
  int32_t hb_maxdepth;
  double sol_K, sol_Bd;                  // constraint stiffness / damping of solref (constants of the model)
  double rcap_hl[HRG_NRCAP], hcap_hl[HRG_NHB];  // half lengths of the collision capsules (rigid: constants of the model)
  int32_t hb_njump;                      // rounds of pointer jumping that cover the deepest path
  int32_t hb_jump[4][HRG_NHB];           // ancestor 2^s levels up (-1: beyond the root)
  int32_t phase_mask;        // read by the -DHRG_STAMPS diagnostic build only (HRG_PHASE_MASK); the shipping kernels ignore it
  hrg_path brake_full;       // fail-safe profile from the steady state (s'=1, s''=0): constant per model
  double brake_T, brake_ds;
  // human reach capsule table (one entry per lane): kind 0 ACC, 1 VEL, 2 POS ball, 3 POS part
  int32_t hc_n;
  int32_t hc_kind[HRG_NHCAP_MAX], hc_j1[HRG_NHCAP_MAX], hc_j2[HRG_NHCAP_MAX];
  double hc_th[HRG_NHCAP_MAX], hc_v[HRG_NHCAP_MAX], hc_a[HRG_NHCAP_MAX], hc_len[HRG_NHCAP_MAX];
  // robot self-collision candidate pairs in enumeration order
  int32_t n_self;
  int32_t self_i[64], self_j[64];
  // pre-check model pairs (capsules 0..6 + gripper cylinder)
  int32_t n_chk;
  int32_t chk_i[32], chk_j[32];
  // animation clips (frames in device memory)
  hrg_clip_table clips;
  const double* hull_dev;    // hull vertices of the arm links in device memory (m.robot_hulls; m.hull_verts is the creator's host pointer)
};

struct Contact {
  int32_t g1, g2, b1, b2;
  double dist, n[3], pos[3];
};

// constraint-row slots (lanes): 0..7 friction loss | 8..23 joint limits (dof, lo/hi) | 24.. contacts x 4 pyramid edges
#define ROW_CON0 24
#define ROW_WELD0 (ROW_CON0 + 4 * NCON_DYN)   // 6 equality rows of the object <-> hand weld (cube variant: lanes 56..61)
#if HRG_HAMMER
// hammering: two rows per lane -- 0..7 friction loss of the robot tree | 8 of the nail's slide joint | 9..24 robot joint limits | 25, 26 the nail's range |
// 27..29 connect at the left hand, 30..35 weld at the right hand | 36..115 contacts x 4 pyramid edges
#define HROW_NFRIC 8
#define HROW_LIM0 9
#define HROW_NLIM 25
#define HROW_EQ0 27
#define HROW_CON0 36
#define HROW_NEQ 9
#define NROW 128
#elif HRG_STACK
// stacking: two rows per lane -- 0..7 friction loss | 8..23 joint limits | 24..35 the two welds (hand, component) | 36..127 contacts x 4 pyramid edges
#define SROW_WELD0 24
#define SROW_CON0 36
#define NROW HRG_NROW_STACK
#elif HRG_BOX && (HRG_HANDOVER || HRG_LIFT)
#define NROW (ROW_WELD0 + 6)   // lifting: the 6 rows of the two connect equalities take the weld's slots (general rows, stored after the contact rows in Jc)
#else
#define NROW ROW_WELD0
#endif

// per-workgroup (= per-env) LDS image.  Sized to <= 10 KB so that 16 envs (4 waves/SIMD) are resident per CU:
// 4096 envs on 256 CUs then run in one round.  Phase-local scratch shares one union.
struct GjkLds { double y[4][3], a[4][3], b[4][3], l[4], lam[4]; int ia[4], ib[4]; };   // GJK simplex: points of the Minkowski difference, their witnesses on the hull / the segment, weights
struct Lds {
  hrg_env_state st;
  // robot tree at the simulation state (live across the whole cycle)
  double kR[NV][9], kp[NV][3], Sw[NV][3], Sv[NV][3], vw[NV][3], vv[NV][3];
  double M[NV * NV];                     // (the Newton systems live in registers; the IK front-end borrows M / bias as scratch for its 6 x 6 factor)
  double bias[NV], a0[NVS], Ma0[NVS], ctrl[NV], qacc[NVS], g[NVS], d[NVS];
#if HRG_BOX
  hrg_box_state bx;                      // the cube (streamed from its own HBM array)
  double bR[9];                          // its rotation matrix at the current substep
  double bMr[9];                         // its world-frame rotational inertia  mean 1 + R diag(I - mean) R'  (exactly diagonal for a cube)
  double hand_q[4], hand_off[3];         // orientation of the hand mocap body / its offset from the hand site, computed with the human tree (handover tasks)
  int palm_hit, palm_pad;                // the cube touches the palm of the holding hand (RobotHumanHandoverCart)
#if HRG_HANDOVER
  double hcap_keep[HRG_NHB][6];          // the cycle's human capsules, kept across the first physics pass (hcap shares its LDS with the solver scratch)
#endif
#endif
  double act[NV];                        // this step's action (7 used)
  int acc_has_collision, acc_collision_type, acc_failsafe, acc_pad;  // per-policy-step accumulators
#if HRG_HAMMER
  hrg_hammer_state hm;                   // board, hammer, nail + task bookkeeping (streamed from its own HBM array)
  double gR[2][9];                       // rotation matrices of the board and the hammer at the current substep
  double gc[HRG_HM_NGEOM][3];            // world centres of the four collision geoms (board, handle, head, nail head)
  double nail_org[3], nail_axis[3];      // nail_head body origin and the slide axis, world
  double Mb[64], fb[8];                  // mass matrix (8 x 8, pad DoF with a unit diagonal) and applied force of the board + nail subtree
  double Ihw[9], tauh[3];                // the hammer's world-frame rotational inertia and gyroscopic torque
  double nsG[(9 + 2 * NCON_DYN) * (10 + 2 * NCON_DYN) / 2];   // noslip pass: Gram matrix of its items (9 friction-loss rows + 2 pairs per contact), packed lower triangle
  Contact con[NCON_DYN];
#elif HRG_STACK
  hrg_stack_state sk;                    // the four cubes + task bookkeeping (streamed from its own HBM array)
  Contact con[NCON_DYN];                 // (the coupled Newton system lives in registers: Tiles, hrgym_hip.hip)
#elif HRG_BOX
  Contact con[NCON_DYN];                 // collide -> constraint-row set-up (the coupled 14-DoF Newton system lives in registers: Tiles<2>)
#else
  Contact con[NCON_DYN];
#endif
  union {
    struct {  // shield_step
      double cq[NARM], cv[NARM], ca[NARM], qe[NARM];
      double scap[2][HRG_NSHIELD_RCAP][6];
      double rc[HRG_NSHIELD_RCAP][7];
      hrg_ltt cand;
    };
    struct {  // robot_dynamics_terms: composite inertia of the subtree of body j applied to its joint axis (body lanes -> mass-matrix lanes)
      double F[NV][6];
    };
    struct {  // human_control + collide + classify (the dynamics step's rows take this space afterwards)
      double hcap[HRG_NHB][6], rcapw[HRG_NRCAP][6];
      int hnear[HRG_NHB];                // human capsules whose bounding sphere comes near the robot (the pair rounds of collide run over these)
      double rcen[HRG_NRCAP][3];         // capsule centres: collide -> classify (the speed of a robot geom at a human contact)
#if HRG_HULLS
      GjkLds gjk;                        // the simplex of a hull query (hrgym_hull.h)
#endif
#if HRG_STACK
      double cR[NCUBE][9];               // cube rotation matrices of the substep's narrowphase
#endif
    };
#if HRG_HAMMER
    struct {  // dynamics_step (hammering): dense rows of J over the 24 DoF (+1 pad: odd stride) -- 4 per contact, then the 9 equality rows; per-row gradient / curvature
      double Jc[4 * NCON_DYN + HROW_NEQ][NVS + 1], rg[NROW], rh[NROW];
    };
#elif HRG_STACK
    struct {  // dynamics_step_stack: contact rows of J, compact: [robot 8 | first cube 6 | second cube 6] (+1 pad: odd stride), per-row gradient / curvature
      double Jc[4 * NCON_DYN][21], rg[NROW], rh[NROW];
      int con_ca[NCON_DYN], con_cb[NCON_DYN], con_rob[NCON_DYN + 1];   // cube index of geom 1 / geom 2 (-1: none), contact has a robot part
    };
#else
    struct {  // dynamics_step: contact rows of J (padded to 9: conflict-free ds_read_b64), per-row gradient / curvature
      double Jc[4 * NCON_DYN + (HRG_LIFT ? 6 : 0)][NVS + 1], rg[NROW], rh[NROW];
    };
#endif
  };
};

// The per-env LDS image is a file-scope __shared__ object: every device function addresses it directly (ds_* instructions),
// nothing is passed around as a generic pointer.
// HRG_WG_WAVES envs (one wave each) may share a workgroup: with W = 4 the dispatcher has to place the four waves of a workgroup at once, one per
// SIMD, so every SIMD of a full CU runs exactly LDS-capacity / W envs from the first cycle of a launch.  Single-wave workgroups are bound to a SIMD
// one by one, 3 to 5 per SIMD at a 4096-env launch (profiles/r02a_env_times.log): the fifth waits a whole env-step for a slot.  The waves of a
// workgroup share nothing; each addresses its own image, and wave_sync() is a wave-level fence, never an s_barrier.
#ifndef HRG_WG_WAVES
#define HRG_WG_WAVES 1
#endif
#if HRG_WG_WAVES > 1
__shared__ Lds g_Lw[HRG_WG_WAVES];
#define g_L (g_Lw[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))])
#else
__shared__ Lds g_L;
#endif
DI int hrg_lane() { return (int)(threadIdx.x & 63u); }
// Launch order of a step: workgroup i steps env cur[i].  Every wave leaves its env in the order of the NEXT launch -- envs that were busy (robot contacts, fail-safe
// manoeuvre) from the front, the others from the back -- so busy envs start first and, with the dispatcher's placement (consecutive workgroups go round the 256 CUs, and
// round the 4 SIMDs of a CU every 256), land on different SIMDs instead of piling their long instruction streams onto one.  buf = [order A (n) | order B (n) |
// front/back counters A (2) | counters B (2)]; the launch with parity p reads order p, fills order 1-p through counters 1-p and clears counters p for the launch after it.
// fair = the progress table of the level-waves priority (hrgym_hip.hip, HRG_FAIR): ONE table per device, shared by every kernel variant, so that waves of different
// kernels of a mixed batch that share a SIMD see each other.
#define HRG_FAIR_SLOTS (8 * 8 * 2 * 16 * 4 * 8)   // XCC x SE x SH x CU x SIMD x wave slot
struct StepOrder { int32_t* buf; int32_t n; int32_t parity; int32_t* fair; };
DI int hrg_env() { return (int)blockIdx.x * HRG_WG_WAVES + (int)(threadIdx.x >> 6); }
#define HRG_LAUNCH_DIMS(n) dim3(((n) + HRG_WG_WAVES - 1) / HRG_WG_WAVES), dim3(64 * HRG_WG_WAVES)

// Model constants are read through the CONSTANT address space from a wave-uniform base held in SGPRs, so that every
// uniform-index access becomes a scalar load (s_load_*, scalar cache, SGPR destination) instead of a per-lane flat load.
typedef const DevModel __attribute__((address_space(4)))* ModelPtr;
DI ModelPtr uniform_model(const DevModel* dm) {
  const unsigned long long a = (unsigned long long)dm;
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(a & 0xffffffffull));
  unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  // opaque to the optimiser: model loads are re-issued where they are used (cheap scalar-cache hits) instead of being
  // hoisted out of the 25-cycle loop and kept live in SGPRs, which would spill
  asm volatile("" : "+s"(lo), "+s"(hi));
  return (ModelPtr)(((unsigned long long)hi << 32) | lo);
}

// ------------------------------------------------------------------------------------------------ math
// sqrt without the denormal-range scaling and the IEEE fix-ups of the generic lowering (20 instructions): v_rsq_f64, one Goldschmidt step and two residual
// corrections -- the same iteration, for arguments that are lengths, norms and discriminants of this model (never denormal; zero is handled).  Within an ulp
// or two of the correctly rounded root; the oracle's sqrt is IEEE, the parity tolerance (1e-5 relative) is eleven orders above the difference.
DI double fsqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
  g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
  return x > 0 ? g : 0.0;
}
// read-only operands are templated on the pointer type so that LDS / constant / private operands keep their address space
DI void v3set(double* r, double a, double b, double c) { r[0] = a; r[1] = b; r[2] = c; }
template <class PA>
DI void v3cpy(double* r, PA a) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
template <class PA, class PB>
DI void v3add(double* r, PA a, PB b) { r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; }
template <class PA, class PB>
DI void v3sub(double* r, PA a, PB b) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
template <class PA>
DI void v3scl(double* r, PA a, double s) { r[0] = a[0] * s; r[1] = a[1] * s; r[2] = a[2] * s; }
template <class PA, class PB>
DI void v3madd(double* r, PA a, PB b, double s) { r[0] = a[0] + b[0] * s; r[1] = a[1] + b[1] * s; r[2] = a[2] + b[2] * s; }
template <class PA, class PB>
DI double v3dot(PA a, PB b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
template <class PA>
DI double v3norm(PA a) { return fsqrt(v3dot(a, a)); }
template <class PA, class PB>
DI void v3cross(double* r, PA a, PB b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
template <class PA, class PB>
DI void m3mulv(double* r, PA M, PB v) {
  double x = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
  double y = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
  double z = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
template <class PA, class PB>
DI void m3mul(double* R, PA A, PB B) {
  double T[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
#pragma unroll
  for (int i = 0; i < 9; i++) R[i] = T[i];
}
template <class PA>
DI void quat2mat(double* M, PA q) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  M[0] = 1 - 2 * (y * y + z * z); M[1] = 2 * (x * y - w * z); M[2] = 2 * (x * z + w * y);
  M[3] = 2 * (x * y + w * z); M[4] = 1 - 2 * (x * x + z * z); M[5] = 2 * (y * z - w * x);
  M[6] = 2 * (x * z - w * y); M[7] = 2 * (y * z + w * x); M[8] = 1 - 2 * (x * x + y * y);
}
template <class PA, class PB>
DI void quatmul(double* r, PA a, PB b) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
// sin/cos for bounded arguments (joint angles, |x| < ~100): Cody-Waite reduction by pi/2 in two pieces and the
// fdlibm kernel polynomials on [-pi/4, pi/4]; ~1 ulp, a fraction of the registers/instructions of the generic ocml path.
DI void sincos_small(double x, double* sn, double* cs) {
  const double k = rint(x * 6.36619772367581382433e-01);
  const double r = (x - k * 1.57079632673412561417e+00) - k * 6.07710050650619224932e-11;
  const double z = r * r;
  const double ps = r + r * z * (-1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
                    z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)))));
  const double pc = 1.0 - 0.5 * z + z * z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                    z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
  const int q = ((int)k) & 3;
  const double s0 = (q & 1) ? pc : ps, c0 = (q & 1) ? ps : pc;
  *sn = (q & 2) ? -s0 : s0;
  *cs = ((q + 1) & 2) ? -c0 : c0;
}
template <class PA>
DI void axisangle2mat(double* M, PA ax, double ang) {
  double s, c;
  sincos_small(ang, &s, &c);
  double t = 1 - c, x = ax[0], y = ax[1], z = ax[2];
  M[0] = t * x * x + c; M[1] = t * x * y - s * z; M[2] = t * x * z + s * y;
  M[3] = t * x * y + s * z; M[4] = t * y * y + c; M[5] = t * y * z - s * x;
  M[6] = t * x * z - s * y; M[7] = t * y * z + s * x; M[8] = t * z * z + c;
}
DI double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

// wave-wide helpers (wave = 64 lanes on gfx950)
// one DPP step of a 64-bit value: lanes outside row_mask receive 0.0
template <int CTRL, int ROW_MASK>
DI double dpp_f64(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// wave-wide sum on the DPP network (no LDS crossbar): quad swaps, row half-mirror / mirror, then the gfx9 row
// broadcasts 15 and 31; the total lands in lane 63 and is read back as a scalar.
DI double wave_sum(double v) {
  v += dpp_f64<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x141, 0xf>(v);  // row_half_mirror
  v += dpp_f64<0x140, 0xf>(v);  // row_mirror
  v += dpp_f64<0x142, 0xa>(v);  // row_bcast15 -> rows 1,3
  v += dpp_f64<0x143, 0xc>(v);  // row_bcast31 -> rows 2,3
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), 63);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
DI double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { double t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
  return v;
}
// maximum over each row of 16 lanes on the DPP network (every lane of the row receives it)
DI double row16_max(double v) {
  double t;
  t = dpp_f64<0xB1, 0xf>(v); v = t > v ? t : v;    // quad_perm [1,0,3,2]
  t = dpp_f64<0x4E, 0xf>(v); v = t > v ? t : v;    // quad_perm [2,3,0,1]
  t = dpp_f64<0x141, 0xf>(v); v = t > v ? t : v;   // row_half_mirror
  t = dpp_f64<0x140, 0xf>(v); v = t > v ? t : v;   // row_mirror
  return v;
}
// orders LDS traffic between the lane roles of ONE wave.  A wave's LDS instructions execute in issue order, so no hardware wait is needed between a
// write by one lane and a read by another; what has to be stopped is the compiler moving one across the other.  (With one wave per workgroup
// __syncthreads() lowers to the same thing plus a drain of every outstanding memory operation.)
#if HRG_WG_WAVES > 1
DI void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
#else
DI void wave_sync() { __syncthreads(); }
#endif

// ------------------------------------------------------------------------------------------------ RNG
DI uint64_t mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ULL;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}
DI double rng_u01(uint64_t seed, uint64_t env, uint64_t episode, uint64_t stream, uint64_t idx) {
  uint64_t h = mix64(seed);
  h = mix64(h ^ (env * 0xD1B54A32D192ED03ULL));
  h = mix64(h ^ (episode * 0x8CB92BA72F3D8DD7ULL));
  h = mix64(h ^ (stream * 0xABC98388FB8FAC03ULL + idx));
  return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}
enum { STREAM_NOISE = 0, STREAM_HUMAN = 1, STREAM_ANIM = 2, STREAM_GOAL = 3, STREAM_ACTION = 4, STREAM_OBJECT = 5, STREAM_TARGET = 6, STREAM_LOOP = 7 };
DI double rng_gauss(uint64_t seed, uint64_t env, uint64_t ep, uint64_t stream, uint64_t idx) {
  double u1 = rng_u01(seed, env, ep, stream, 2 * idx), u2 = rng_u01(seed, env, ep, stream, 2 * idx + 1);
  return fsqrt(-2.0 * log(1.0 - u1)) * cos(2.0 * HRG_PI * u2);
}

// ------------------------------------------------------------------------------------------------ segments
// closest points of two segments; returns squared distance (Ericson 5.1.9)
template <class PA, class PB, class PC, class PD>
DI double seg_seg(PA p1, PB q1, PC p2, PD q2, double* c1, double* c2) {
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

#if HRG_BOX || HRG_STACK || HRG_HAMMER
// closest points of a segment and a box (centre c, rotation R row-major, half extents hb[3]): the squared distance along the
// segment is a convex piecewise quadratic in t; safeguarded Newton on its derivative (exact inside one piece).
template <class PA, class PB, class PC, class PR>
DI double seg_box(PA p1, PB p2, PC c, PR R, const double* hb, double* on_seg, double* on_box) {
  double a[3], d[3], t0[3];
  v3sub(t0, p1, c);
  for (int k = 0; k < 3; k++) a[k] = R[k] * t0[0] + R[3 + k] * t0[1] + R[6 + k] * t0[2];
  v3sub(t0, p2, p1);
  for (int k = 0; k < 3; k++) d[k] = R[k] * t0[0] + R[3 + k] * t0[1] + R[6 + k] * t0[2];
  double g0 = 0, g1 = 0, t;
  for (int k = 0; k < 3; k++) {
    const double x0 = a[k], x1 = a[k] + d[k];
    g0 += (x0 > hb[k] ? x0 - hb[k] : (x0 < -hb[k] ? x0 + hb[k] : 0.0)) * d[k];
    g1 += (x1 > hb[k] ? x1 - hb[k] : (x1 < -hb[k] ? x1 + hb[k] : 0.0)) * d[k];
  }
  if (g0 >= 0) t = 0;
  else if (g1 <= 0) t = 1;
  else {
    double lo = 0, hi = 1;
    t = -g0 / (g1 - g0);
#pragma unroll 1
    for (int it = 0; it < 10; it++) {
      double g = 0, H = 0;
      for (int k = 0; k < 3; k++) {
        const double x = a[k] + t * d[k], e = x > hb[k] ? x - hb[k] : (x < -hb[k] ? x + hb[k] : 0.0);
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
  for (int k = 0; k < 3; k++) {
    on_seg[k] = c[k] + R[3 * k] * x[0] + R[3 * k + 1] * x[1] + R[3 * k + 2] * x[2];
    on_box[k] = c[k] + R[3 * k] * y[0] + R[3 * k + 1] * y[1] + R[3 * k + 2] * y[2];
  }
  return e2;
}

// A capsule lying along a box face touches it in a stretch, not a point (MuJoCo's capsule-box generates two contacts there): kn = the box axis the
// closest pair (cs on the capsule axis, cb on the box) is separated along; the stretch = the part of the axis whose other two box coordinates stay
// inside the box.  Returns true with the axis / box point pair at end `which` of the stretch when both ends are closer than r and at least 1 mm apart.
template <class PA, class PB, class PC, class PR>
DI bool cap_box_two(PA p1, PB p2, PC c, PR R, const double* hb, double r, const double* cs, const double* cb, int which, double* s_out, double* b_out) {
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
  bool ok = true;
  for (int k = 0; k < 3; k++) {
    if (k == kn) continue;
    if (fabs(d[k]) < 1e-12) { if (fabs(a[k]) > hb[k]) ok = false; }
    else {
      double ta = (-hb[k] - a[k]) / d[k], tb = (hb[k] - a[k]) / d[k];
      if (ta > tb) { const double t = ta; ta = tb; tb = t; }
      if (ta > lo) lo = ta;
      if (tb < hi) hi = tb;
    }
  }
  if (!ok || !(hi > lo)) return false;
  if ((hi - lo) * fsqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) < 1e-3) return false;
  for (int e = 0; e < 2; e++) {
    const double t = e ? hi : lo;
    double x[3], y[3], e2 = 0;
    for (int k = 0; k < 3; k++) { x[k] = a[k] + t * d[k]; y[k] = clampd(x[k], -hb[k], hb[k]); e2 += (x[k] - y[k]) * (x[k] - y[k]); }
    const double dd = fsqrt(e2);
    if (!(dd - r < 0) || !(dd > 1e-9)) return false;
    if (e == which)
      for (int k = 0; k < 3; k++) {
        s_out[k] = c[k] + R[3 * k] * x[0] + R[3 * k + 1] * x[1] + R[3 * k + 2] * x[2];
        b_out[k] = c[k] + R[3 * k] * y[0] + R[3 * k + 1] * y[1] + R[3 * k + 2] * y[2];
      }
  }
  return true;
}

#endif

// ------------------------------------------------------------------------------------------------ profiles
DI double scurve_time(double dv, double amax, double jmax) {
  double ad = fabs(dv);
  return ad >= amax * amax / jmax ? ad / amax + amax / jmax : 2 * fsqrt(ad / jmax);
}
DI double dist_nocruise(double va, double vc, double amax, double jmax) {
  return 0.5 * (va + vc) * scurve_time(vc - va, amax, jmax) + 0.5 * vc * scurve_time(vc, amax, jmax);
}
DI void scurve(double va, double vb, double amax, double jmax, double* dur, double* jerk) {
  double d = vb - va, ad = fabs(d), sg = d >= 0 ? 1.0 : -1.0;
  if (ad >= amax * amax / jmax) {
    double tj = amax / jmax;
    dur[0] = tj; dur[1] = ad / amax - tj; dur[2] = tj;
  } else {
    double tj = fsqrt(ad / jmax);
    dur[0] = tj; dur[1] = 0; dur[2] = tj;
  }
  jerk[0] = sg * jmax; jerk[1] = 0; jerk[2] = -sg * jmax;
}

// what the last three parts of a joint's profile ([S-curve v -> sg w][cruise][S-curve sg w -> 0], segments n0 ..) were planned from: the time synchronisation re-plans them
struct LttTail { int n0; double v, sg, Dm, vm, w; };
// one joint of a long-term trajectory (executed by lane j < 6)
DI void ltt_plan_joint(hrg_ltt* L, int j, double q0, double v0, double a0, double goal, double vmax, double amax, double jmax, LttTail* tail) {
  double* dur = L->dur[j];
  double* jerk = L->jerk[j];
  for (int i = 0; i < HRG_LTT_NSEG; i++) { dur[i] = 0; jerk[i] = 0; }
  L->q0[j] = q0; L->v0[j] = v0; L->a0[j] = a0; L->qT[j] = goal;
  int n = 0;
  double q = q0, v = v0;
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
    scurve(v, 0, amax, jmax, dur + n, jerk + n);
    D -= dstop;
    v = 0;
  }
  n += 3;
  double w_lo = fabs(v), w_hi = vmax > w_lo ? vmax : w_lo, w, tc = 0;
  double Dm = sg * D, vm = sg * v;
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
    w = 0.5 * (fsqrt(vtri * vtri + 4.0 * amax * De) - vtri);
    if (!(w > lo)) w = lo;
    if (!(w < hi)) w = hi;
    for (int it = 0; it < 80; it++) {
      const double d1 = w - vm, T1 = scurve_time(d1, amax, jmax), T2 = scurve_time(w, amax, jmax);
      const double f = 0.5 * (vm + w) * T1 + 0.5 * w * T2 - Dm;
      if (f == 0) break;
      if (f < 0) lo = w; else hi = w;
      const double T1p = fabs(d1) >= vtri ? 1.0 / amax : (fabs(d1) > 0 ? 1.0 / fsqrt(jmax * fabs(d1)) : 0.0);
      const double T2p = fabs(w) >= vtri ? 1.0 / amax : (fabs(w) > 0 ? 1.0 / fsqrt(jmax * fabs(w)) : 0.0);
      const double fp = 0.5 * T1 + 0.5 * (vm + w) * T1p + 0.5 * T2 + 0.5 * w * T2p;
      double nw = fp > 0 ? w - f / fp : 0.5 * (lo + hi);
      if (!(nw >= lo && nw <= hi)) nw = 0.5 * (lo + hi);
      const double step = fabs(nw - w);
      w = nw;
      if (step <= 4e-16 * (1.0 + fabs(w)) || hi - lo <= 4e-16 * (1.0 + hi)) break;
    }
  }
  tail->n0 = n; tail->v = v; tail->sg = sg; tail->Dm = Dm; tail->vm = vm; tail->w = w;
  scurve(v, sg * w, amax, jmax, dur + n, jerk + n);
  n += 3;
  dur[n] = tc; jerk[n] = 0; n++;
  scurve(sg * w, 0, amax, jmax, dur + n, jerk + n);
}
// Time synchronisation (sara-shield's LongTermPlanner [UPSTREAM]; oracle: ltt_sync_joint): a joint whose own time-optimal profile ends before T keeps its
// first parts and drives the rest at a lower cruise speed w with g(w) = T1(w - vm) + T2(w) + (Dm - dist_nocruise(vm, w)) / w = T - (time of the first parts):
// g falls monotonically from infinity to the time-optimal duration, Newton with the analytic derivative inside the bracket.  Lane j < 6.
DI void ltt_sync_joint(hrg_ltt* L, int j, double T, double amax, double jmax, const LttTail& tl) {
  double* dur = L->dur[j];
  double* jerk = L->jerk[j];
  double tpre = 0, tall = 0;
  for (int i = 0; i < HRG_LTT_NSEG; i++) { if (i < tl.n0) tpre += dur[i]; tall += dur[i]; }
  if (!(tall < T - 1e-12) || !(tl.w > 0)) return;
  const double Tt = T - tpre, vm = tl.vm, Dm = tl.Dm, vtri = amax * amax / jmax;
  double lo = 0, hi = tl.w;
  double w = hi * (tall - tpre) / Tt;
  bool found = false;
  for (int it = 0; it < 80; it++) {
    const double d1 = w - vm, T1 = scurve_time(d1, amax, jmax), T2 = scurve_time(w, amax, jmax);
    const double dnc = 0.5 * (vm + w) * T1 + 0.5 * w * T2, rest = Dm - dnc;
    const double g = T1 + T2 + rest / w - Tt;
    if (g == 0) { found = true; break; }
    if (g > 0) lo = w; else hi = w;
    const double sd = d1 >= 0 ? 1.0 : -1.0;
    const double T1p = sd * (fabs(d1) >= vtri ? 1.0 / amax : (fabs(d1) > 0 ? 1.0 / fsqrt(jmax * fabs(d1)) : 0.0));
    const double T2p = w >= vtri ? 1.0 / amax : (w > 0 ? 1.0 / fsqrt(jmax * w) : 0.0);
    const double dncp = 0.5 * T1 + 0.5 * (vm + w) * T1p + 0.5 * T2 + 0.5 * w * T2p;
    const double gp = T1p + T2p - dncp / w - rest / (w * w);
    double nw = gp < 0 ? w - g / gp : 0.5 * (lo + hi);
    if (!(nw > lo && nw < hi)) nw = 0.5 * (lo + hi);
    const double step = fabs(nw - w);
    w = nw;
    found = true;
    if (step <= 4e-16 * (1.0 + w) || hi - lo <= 4e-16 * (1.0 + hi)) break;
  }
  if (!found || !(w > 0)) return;
  const double tc = (Dm - dist_nocruise(vm, w, amax, jmax)) / w;
  if (!(tc >= 0)) return;
  int n = tl.n0;
  scurve(tl.v, tl.sg * w, amax, jmax, dur + n, jerk + n);
  n += 3;
  dur[n] = tc; jerk[n] = 0; n++;
  scurve(tl.sg * w, 0, amax, jmax, dur + n, jerk + n);
}

DI void ltt_eval(const hrg_ltt* L, int j, double s, double* q, double* v, double* a) {
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

// the same walk for two path positions sa <= sb at once: (q, q', q'') at sa and q at sb; per target the operations are those of
// ltt_eval, so the results are bitwise the ones of two separate calls
DI void ltt_eval2(const hrg_ltt* L, int j, double sa, double sb, double* qa, double* va, double* aa_, double* qb) {
  double qq = L->q0[j], vv = L->v0[j], aa = L->a0[j], ta = sa < 0 ? 0 : sa, tb = sb < 0 ? 0 : sb;
  bool done_a = false;
  for (int i = 0; i < HRG_LTT_NSEG; i++) {
    const double d = L->dur[j][i], jj = L->jerk[j][i];
    if (!done_a && ta < d) {
      *qa = qq + vv * ta + 0.5 * aa * ta * ta + jj * ta * ta * ta * SIXTH;
      *va = vv + aa * ta + 0.5 * jj * ta * ta;
      *aa_ = aa + jj * ta;
      done_a = true;
    }
    if (tb < d) {
      *qb = qq + vv * tb + 0.5 * aa * tb * tb + jj * tb * tb * tb * SIXTH;
      return;
    }
    qq += vv * d + 0.5 * aa * d * d + jj * d * d * d * SIXTH;
    vv += aa * d + 0.5 * jj * d * d;
    aa += jj * d;
    ta -= d;
    tb -= d;
  }
  if (!done_a) { *qa = L->qT[j]; *va = 0; *aa_ = 0; }
  *qb = L->qT[j];
}

__host__ DI void path_plan(hrg_path* P, double s0, double v0, double a0, double ve, double amax, double jmax) {
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
__host__ DI double path_total(const hrg_path* P) { return P->dur[0] + P->dur[1] + P->dur[2]; }
// speed a profile ends at (what it was planned to brake to): the state after its three phases, snapped to the exact values 0 and 1 it can be planned for
__host__ DI double path_vend(const hrg_path* P) {
  double vv = P->v0, aa = P->a0;
  for (int i = 0; i < 3; i++) { const double d = P->dur[i], jj = P->jerk[i]; vv += aa * d + 0.5 * jj * d * d; aa += jj * d; }
  if (fabs(vv) < 1e-12) return 0.0;
  if (fabs(vv - 1.0) < 1e-12) return 1.0;
  return vv;
}
__host__ DI void path_eval(const hrg_path* P, double t, double ve, double* s, double* v, double* a) {
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
  *s = ss + ve * t; *v = ve; *a = 0;
}

// ------------------------------------------------------------------------------------------------ spatial inertia
template <class PA, class PB>
DI void sinertia_body(double* s /*10*/, double m, PA c, PB Ic) {
  s[0] = m;
  s[1] = c[0] * m; s[2] = c[1] * m; s[3] = c[2] * m;
  double cc = v3dot(c, c);
  s[4] = Ic[0] + m * (cc - c[0] * c[0]);
  s[5] = Ic[1] + m * (cc - c[1] * c[1]);
  s[6] = Ic[2] + m * (cc - c[2] * c[2]);
  s[7] = Ic[3] - m * c[0] * c[1];
  s[8] = Ic[4] - m * c[0] * c[2];
  s[9] = Ic[5] - m * c[1] * c[2];
}
template <class PA, class PB, class PC>
DI void sinertia_mul(double* n, double* f, PA s, PB w, PC v) {
  double hv[3], hw[3];
  v3cross(hv, s + 1, v);
  v3cross(hw, s + 1, w);
  n[0] = s[4] * w[0] + s[7] * w[1] + s[8] * w[2] + hv[0];
  n[1] = s[7] * w[0] + s[5] * w[1] + s[9] * w[2] + hv[1];
  n[2] = s[8] * w[0] + s[9] * w[1] + s[6] * w[2] + hv[2];
  f[0] = s[0] * v[0] - hw[0];
  f[1] = s[0] * v[1] - hw[1];
  f[2] = s[0] * v[2] - hw[2];
}

// 8x8 Cholesky across the wave: lane (i,j) = (lane>>3, lane&7) owns A_ij in a register; right-looking
// elimination with three shuffles per pivot, no LDS round trips.  Returns L_ij in lanes i >= j.
// The subtraction order per entry (k ascending) is the one of the oracle's left-looking loop.
DI double chol_lanes(double a, int lane, bool* ok) {
  const int i = lane >> 3, j = lane & 7;
  bool good = true;
#pragma unroll
  for (int k = 0; k < NV; k++) {
    const double akk = __shfl(a, k * 9, 64);
    if (!(akk > 0)) good = false;
    const double inv = rsqrt(akk), d = akk * inv;  // one transcendental per pivot
    const double lik = __shfl(a, i * 8 + k, 64) * inv;
    const double ljk = __shfl(a, j * 8 + k, 64) * inv;
    if (j == k) { if (i == k) a = d; else if (i > k) a = lik; }
    else if (i > k && j > k) a -= lik * ljk;
  }
  *ok = good;
  return a;
}
// inclusive prefix sum along the robot's kinematic chain for lanes = bodies: lanes 0..5 the serial arm, lanes 6, 7 (fingers,
// children of link 6) = prefix of lane 5 + own value.  DPP row shifts inside row 0; shifted-in lanes contribute 0.
DI double chain_prefix(double x, int lane) {
  double y = lane < NARM ? x : 0.0;
  y += dpp_f64<0x111, 0xf>(y);  // row_shr:1
  y += dpp_f64<0x112, 0xf>(y);  // row_shr:2
  y += dpp_f64<0x114, 0xf>(y);  // row_shr:4
  return lane < NARM ? y : y + x;
}
// sum over the subtree of body i = lane i of the robot's kinematic tree: bodies i..7 for the arm links 0..5 (the fingers 6, 7 hang off link 5), the body itself for
// a finger.  DPP row shifts to the left inside row 0; lanes 8..15 contribute 0.
DI double subtree_sum(double x, int lane) {
  double y = lane < NV ? x : 0.0;
  y += dpp_f64<0x101, 0xf>(y);  // row_shl:1
  y += dpp_f64<0x102, 0xf>(y);  // row_shl:2
  y += dpp_f64<0x104, 0xf>(y);  // row_shl:4
  return lane < NARM ? y : x;
}
// a wave-uniform double that arrived in a vector register (loaded from the LDS image, computed from such loads): moved to a scalar register pair.  A value that
// lives across a register-hungry stretch then costs two SGPRs (spilled, if need be, to one lane of a VGPR) instead of two VGPRs of the 128.
DI double uniform_f64(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffLL));
  const int hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// a value of lane K (compile-time) in every lane: two v_readlane into SGPRs instead of an LDS-crossbar shuffle
template <int K>
DI double lane_value(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), K);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), K);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// Inverse of a symmetric positive definite 8x8 matrix across the wave: in-place Gauss-Jordan sweeps, lane (i,j) owns entry (i,j), three shuffles and one reciprocal
// per pivot -- the cost of a factorisation, but what comes out turns every later solve with the matrix (unconstrained acceleration, the Newton direction while no
// row has curvature, the implicit-damping step) into one product per lane and a three-step row reduction, instead of a 16-step dependent substitution chain on
// 8 lanes.  Pivots are the same Schur complements as the Cholesky pivots: positive definiteness is checked on them.
#define HRG_PIVOT(v, K) lane_value<(K) * 9>(v)   // the pivot is one lane's value: scalar broadcast, and the reciprocal starts without waiting for a shuffle (-0.7 %)
template <int K>
DI void inv_pivot(double& a, int i, int j, bool& good) {
  const double akk = HRG_PIVOT(a, K);
  if (!(akk > 0)) good = false;
  const double pa = 1.0 / akk;
  const double aik = __shfl(a, i * 8 + K, 64);
  const double akj = __shfl(a, K * 8 + j, 64);
  if (i == K) a = j == K ? pa : akj * pa;
  else if (j == K) a = -aik * pa;
  else a -= aik * akj * pa;
}
// (M of a substep; the Newton Hessian once a row has curvature)
DI double spd_inverse1(double a, int lane, bool* ok) {
  const int i = lane >> 3, j = lane & 7;
  bool good = true;
  inv_pivot<0>(a, i, j, good); inv_pivot<1>(a, i, j, good); inv_pivot<2>(a, i, j, good); inv_pivot<3>(a, i, j, good);
  inv_pivot<4>(a, i, j, good); inv_pivot<5>(a, i, j, good); inv_pivot<6>(a, i, j, good); inv_pivot<7>(a, i, j, good);
  *ok = good;
  return a;
}
// a 6 x 6 block padded to the 8 x 8 lane layout with a unit diagonal: the two padded pivots would change nothing (pivot 1, zero row and column), so six sweeps
DI double spd_inverse1_6(double a, int lane, bool* ok) {
  const int i = lane >> 3, j = lane & 7;
  bool good = true;
  inv_pivot<0>(a, i, j, good); inv_pivot<1>(a, i, j, good); inv_pivot<2>(a, i, j, good); inv_pivot<3>(a, i, j, good);
  inv_pivot<4>(a, i, j, good); inv_pivot<5>(a, i, j, good);
  *ok = good;
  return a;
}
// ... and for four matrices at once (the four cube blocks of the stacking task's Newton system): the four dependency chains interleave
template <int K>
DI void inv_pivot4(double* a, int i, int j, bool& good) {
#pragma unroll
  for (int q = 0; q < 4; q++) inv_pivot<K>(a[q], i, j, good);
}
DI void spd_inverse4(double* a, int lane, bool* ok) {
  const int i = lane >> 3, j = lane & 7;
  bool good = true;
  inv_pivot4<0>(a, i, j, good); inv_pivot4<1>(a, i, j, good); inv_pivot4<2>(a, i, j, good); inv_pivot4<3>(a, i, j, good);
  inv_pivot4<4>(a, i, j, good); inv_pivot4<5>(a, i, j, good);   // 6 x 6 blocks padded with a unit diagonal: the padded pivots are no-ops
  *ok = good;
}
// A coupled Newton system in registers: N x N tiles of 8 x 8 -- tile row / column 0 = the robot tree, the others = free bodies (6 x 6, padded to 8 x 8 with a unit
// diagonal) -- and lane (i, j) owns entry (i, j) of every tile.  Gauss-Jordan sweeps over the real pivots turn the tiles into the inverse in place (the lane-parallel
// sweep of spd_inverse1, tile by tile): per pivot one scalar broadcast, one shuffle per tile row and tile column, one multiply-add per tile.  A = the tile row the
// pivots are in, np = how many (8 robot DoF, 6 of a free body), A0 = first tile that takes part (1: the robot block is inverted on its own).
template <int N> struct Tiles { double t[N][N]; };
DI double lane_value_dyn(double v, int k) {   // k wave-uniform
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), k);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), k);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int N, int A, int A0>
DI void tiles_pivots(Tiles<N>& T, int i, int j, int np, bool& good) {
#pragma unroll 1
  for (int P = 0; P < np; P++) {
    const double akk = lane_value_dyn(T.t[A][A], P * 9);
    if (!(akk > 0)) good = false;
    const double pa = 1.0 / akk;
    double col[N], row[N];
#pragma unroll
    for (int a = A0; a < N; a++) { col[a] = __shfl(T.t[a][A], i * 8 + P, 64); row[a] = __shfl(T.t[A][a], P * 8 + j, 64); }
#pragma unroll
    for (int a = A0; a < N; a++)
#pragma unroll
      for (int b = A0; b < N; b++) {
        const bool prow = a == A && i == P, pcol = b == A && j == P;
        const double x = T.t[a][b];
        T.t[a][b] = prow ? (pcol ? pa : row[b] * pa) : (pcol ? -col[a] * pa : x - col[a] * row[b] * pa);
      }
  }
}
// sum over each group of 8 consecutive lanes (one matrix row in the (i,j) lane layout) on the DPP network; every lane of the group receives it
DI double row8_sum(double v) {
  v += dpp_f64<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
  v += dpp_f64<0x141, 0xf>(v);   // row_half_mirror: lane k <-> 7 - k within each half row
  return v;
}
// y = A x for lanes = (i, j): A_ij in a register, x in LDS; returns y_i in every lane of row i
DI double matvec_lanes(double aij, const double* x, int lane) { return row8_sum(aij * x[lane & 7]); }
// publish the factor for the wave-uniform solves: lower triangle + reciprocal diagonal
DI void chol_store(double l, int lane, double* Lm, double* invd) {
  Lm[lane] = l;
  if ((lane >> 3) == (lane & 7)) invd[lane & 7] = 1.0 / l;
}
// solve L L' x = b with x_i living in lane i (< 8): column-oriented substitution, the pivot value is broadcast
// with a scalar readlane, the column of the factor comes from LDS.  Register footprint: one double.
DI double chol_solve_lanes(const double* Lm, const double* invd, double b, int lane) {
  double x = b;
  const int i = lane & 7;
#pragma unroll
  for (int k = 0; k < NV; k++) {
    const double xk = __shfl(x, k, 64) * invd[k];
    if (lane == k) x = xk;
    else if (lane > k && lane < NV) x -= Lm[i * NV + k] * xk;
  }
#pragma unroll
  for (int k = NV - 1; k >= 0; k--) {
    const double xk = __shfl(x, k, 64) * invd[k];
    if (lane == k) x = xk;
    else if (lane < k) x -= Lm[k * NV + i] * xk;
  }
  return x;
}
