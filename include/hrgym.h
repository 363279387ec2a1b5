/* hrgym.h — C ABI of libhrgym_hip.so, the MI355X-native batched stepper for human-robot-gym's
 * ReachHuman hot path (HumanEnv.step = 25 x {mj_forward, SafetyShield.step, controller, human playback,
 * contact bookkeeping, mj_step} + observation/reward/done/info + auto-reset).
 *
 * The reference has NO C interface for this path: it reaches its native engines through two Python
 * extension modules (mujoco_py's MjSim, safety_shield_py's SafetyShield).  Each entry point below names
 * the reference call site(s) it replaces (paths relative to the reference checkout):
 *
 *   hrg_batch_create   <- robosuite.make("ReachHuman", **env_kwargs)      utils/env_util.py:21-26
 *                         + FailsafeController.__init__ / SafetyShield(...) controllers/failsafe_controller/
 *                                                                          failsafe_controller/failsafe_controller.py:113-191
 *                         + SubprocVecEnv([...]) worker spawn              utils/env_util_SB3.py:75-87
 *   hrg_batch_reset    <- VecEnv.reset -> ReachHuman.reset -> HumanEnv._reset_internal
 *                                                                          environments/manipulation/human_env.py:1604-1673
 *                                                                          environments/manipulation/reach_human_env.py:509-523
 *                         + FailsafeController.reset / SafetyShield.reset  failsafe_controller.py:204-250
 *   hrg_batch_step     <- VecEnv.step_async/step_wait -> HumanEnv.step     human_env.py:470-586
 *                         (sim.forward x2 + sim.step per cycle: 504,519,523; SafetyShield.step:
 *                          failsafe_controller.py:329; humanMeasurement: 310; newLongTermTrajectory: 300)
 *                         + ReachHuman.step goal cycling                   reach_human_env.py:383-410
 *                         + TimeLimit.step                                 wrappers/time_limit.py:31-44
 *                         + SubprocVecEnv auto-reset / terminal_observation [SB3 1.5.0]
 *   hrg_batch_contacts <- sim.data.contact[:ncon] as read by HumanEnv._collision_detection
 *                                                                          human_env.py:1082-1123
 *   hrg_batch_get_state / hrg_batch_set_state
 *                      <- HumanEnv.get/set_environment_state               human_env.py:1845-1900
 *   hrg_batch_capsules <- SafetyShield.getRobotReachCapsules / getHumanReachCapsules
 *                                                                          failsafe_controller.py:393,416
 *
 * Conventions: every function returns 0 on success or a negative hrg_status; hrg_last_error() gives the
 * message (thread local).  All `dev` pointers are device (HBM) pointers owned by the caller (e.g.
 * torch tensors' data_ptr()); `host` pointers are host memory.  A batch is bound to one HIP device, is not
 * thread-safe, and orders its work on the stream handed to each call (0 = the null stream).  A simulation
 * that diverges is not an error: the env reports done=1, reward += -10, info[HRG_INFO_SIM_CRASH]=1
 * (mirrors the MujocoException handler at human_env.py:527-546).
 */
#ifndef HRGYM_H
#define HRGYM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------ sizes */
#define HRG_NARM 6        /* Schunk LWA-4P arm hinges (robot.xml:33-58) */
#define HRG_NFINGER 2     /* RethinkValidGripper slide joints (rethink_valid_gripper.py:37-42) */
#define HRG_NV (HRG_NARM + HRG_NFINGER)
#define HRG_NHB 24        /* human bodies incl. pelvis (human.xml:45-330) */
#define HRG_NHJ 23        /* measured human joints (models/objects/human/human.py:57-81) */
#define HRG_NHQ 69        /* human hinge DoF = 23 x 3 */
#define HRG_NRCAP 10      /* robot collision capsules: link0..link6, gripper base, 2 fingers */
#define HRG_NHULL 7         /* arm links with a mesh collision geom: link0 .. link6 = robot capsules 0 .. 6 (robot.xml:29-55) */
#define HRG_NSHIELD_RCAP 7 /* robot capsules the shield tracks: 6 links + gripper */
#define HRG_NBODYPART_MAX 20 /* human body parts (capsule between two measured joints) */
#define HRG_NEXTREMITY_MAX 4 /* POS-model extremities (ball at proximal joint) */
#define HRG_NHCAP_MAX 64  /* human reach capsules over all three models (fits one wavefront) */
#define HRG_LTT_NSEG 12   /* constant-jerk segments per joint of a long-term trajectory */
#define HRG_OBS_DIM 64    /* superset of the flat observation (64 floats = one 256-byte row per env: a wavefront stores a row with one coalesced
                          *  instruction); the host selects columns by obs_keys:
                          *  [0:12] object-state  [12:18] goal_difference  [18:24] robot0_joint_pos  [24:30] robot0_joint_vel
                          *  [30:33] robot0_eef_pos  [33:39] desired_goal   (human_env.py:1483-1602, reach_human_env.py:608-666)
                          *  the cube tasks serve object_quat (x, y, z, w) in [12:16] (those columns are joint-space entries of ReachHuman only)
                          *  PickPlaceHumanCart (pick_place_human_cartesian_env.py:726-841; zero for ReachHuman):
                          *  [39] object_gripped  [40:43] vec_eef_to_object  [43:46] vec_eef_to_target  [46] gripper_aperture
                          *  [47:50] object_pos  [50:53] target_pos
                          *  both tasks: [53:55] robot0_gripper_qpos  [55:57] robot0_gripper_qvel (the scripted experts' inputs,
                          *  demonstrations/experts/pick_place_human_cart_expert.py:24-41)
                          *  [57:61] quat_eef_to_object of the handover tasks / quat_eef_to_board of CollaborativeLiftingCart (x, y, z, w;
                          *  human_robot_handover_cartesian_env.py:860-875, collaborative_lifting_cartesian_env.py:1010-1031), board_quat of
                          *  CollaborativeHammeringCart; [61] its nail_hammering_progress; [62:64] unused (zero) */
#define HRG_ACT_DIM 7     /* 6 joint deltas + 1 gripper (reach_human_expert.py:82-83) */
#define HRG_INFO_DIM 14
#define HRG_NCON_MAX 24   /* contacts reported per env per substep */
#define HRG_NCON_DYN 6    /* contacts that enter the constraint solve (4 pyramid rows each) */
#define HRG_NCON_DYN_BOX 8 /* ... for tasks with the manipulation object (rows 24 + 32 still fit one wavefront) */
#define HRG_NBOXV 6       /* free-joint DoF of the manipulation object */
#define HRG_NVT (HRG_NV + HRG_NBOXV)
#define HRG_NCUBE 4       /* CollaborativeStackingCart: manipulation_object_a, manipulation_object_b, human_l_cube, human_r_cube
                           * (collaborative_stacking_cartesian_env.py:1151-1178) */
#define HRG_NV_STACK (HRG_NV + HRG_NCUBE * HRG_NBOXV) /* DoF of the stacking task's constrained system: robot tree + four free joints */
#define HRG_NCON_DYN_STACK 23 /* contacts that enter its solve: 8 + 16 + 12 + 4 x 23 = 128 constraint rows = two per lane of a wavefront */
#define HRG_NROW_STACK 128
#define HRG_NV_HAMMER 24   /* CollaborativeHammeringCart: three 8-wide blocks -- robot tree 0..7 | board 8..13, nail slide joint 14, pad | hammer 16..21, pad, pad */
#define HRG_NCON_DYN_HAMMER 14 /* contacts that enter its solve: 9 + 18 + 9 + 4 x 14 = 92 constraint rows (two per lane of a wavefront).  14: a soak of 2.4 M env steps had
                                 * more contacts at the end of 6 of them (tools/soak_hammering.py prints the histogram); the rows of 20 cost the kernel two of its five workgroups per CU */
#define HRG_NPREV_MAX 24  /* remembered robot contact pairs (edge trigger, human_env.py:1109-1121) */
#define HRG_MAX_CLIPS 16
#define HRG_MAX_LOOP 4     /* layered sines of an animation loop (utils/animation_utils.py:91-119) */

/* info columns (human_env.py:752-763 + TimeLimit + sim crash) */
enum {
  HRG_INFO_COLLISION = 0,
  HRG_INFO_COLLISION_TYPE = 1,
  HRG_INFO_N_COLLISIONS = 2,
  HRG_INFO_N_COLLISIONS_STATIC = 3,
  HRG_INFO_N_COLLISIONS_ROBOT = 4,
  HRG_INFO_N_COLLISIONS_HUMAN = 5,
  HRG_INFO_N_COLLISIONS_CRITICAL = 6,
  HRG_INFO_TIMEOUT = 7,
  HRG_INFO_FAILSAFE_INTERVENTIONS = 8,
  HRG_INFO_N_GOAL_REACHED = 9,
  HRG_INFO_TRUNCATED = 10, /* TimeLimit.truncated (time_limit.py:42) */
  HRG_INFO_SIM_CRASH = 11,
  HRG_INFO_ACTION_RESAMPLES = 12, /* CollisionPreventionWrapper.action_resamples */
  HRG_INFO_N_OBJECT_HANDED_OVER = 13, /* human_robot_handover_cartesian_env.py:507-511 */
  HRG_INFO_MAX_STACK_HEIGHT = 13      /* CollaborativeStackingCart._get_info (673-679): the same (task-specific) column */
};

/* COLLISION_TYPE flag values, human_env.py:55-77 */
enum { HRG_COL_NULL = 0, HRG_COL_ALLOWED = 1, HRG_COL_HUMAN = 2, HRG_COL_ROBOT = 4, HRG_COL_STATIC = 8, HRG_COL_HUMAN_CRIT = 16 };

/* shield types, failsafe_controller.py:23,173 */
enum { HRG_SHIELD_OFF = 0, HRG_SHIELD_SSM = 1, HRG_SHIELD_PFL = 2 };

/* geom classes used by the contact classifier (human_env.py:948-964) */
/* tasks: ReachHuman (reach_human_env.py), PickPlaceHumanCart (pick_place_human_cartesian_env.py) */
enum { HRG_TASK_REACH = 0, HRG_TASK_PICK_PLACE = 1, HRG_TASK_INSPECTION = 2 /* HumanObjectInspectionCart */,
       HRG_TASK_POINTING = 3 /* PickPlacePointingHumanCart: the target is where the human points (pick_place_pointing_human_cartesian_env.py:336-360) */,
       HRG_TASK_HANDOVER_H2R = 4 /* HumanRobotHandoverCart (human_robot_handover_cartesian_env.py) */,
       HRG_TASK_HANDOVER_R2H = 5 /* RobotHumanHandoverCart (robot_human_handover_cartesian_env.py) */,
       HRG_TASK_LIFTING = 6 /* CollaborativeLiftingCart (collaborative_lifting_cartesian_env.py): robot and human carry a board together */,
       HRG_TASK_STACKING = 7 /* CollaborativeStackingCart (collaborative_stacking_cartesian_env.py): human and robot build a stack of four cubes in turns */,
       HRG_TASK_REACH_BOX = 8 /* ReachHuman with its free `smallBox` object (reach_human_env.py:573-579, 5 cm cube placed anywhere on the table; not whitelisted: a
                               * robot contact with it is a static collision); task logic of ReachHuman, stepped by the cube kernel */,
       HRG_TASK_HAMMERING = 9 /* CollaborativeHammeringCart (collaborative_hammering_cartesian_env.py): the robot holds a hammer and drives a nail into a board the
                               * human presents -- two free bodies (board, hammer), the nail on a slide joint of the board, a weld + a connect to the hands */ };
#define HRG_IS_HANDOVER(task) ((task) == HRG_TASK_HANDOVER_H2R || (task) == HRG_TASK_HANDOVER_R2H)
/* ObjectInspectionPhase, human_object_inspection_cartesian_env.py:43-49 */
enum { HRG_PHASE_APPROACH = 0, HRG_PHASE_READY = 1, HRG_PHASE_INSPECTION = 2, HRG_PHASE_RETREAT = 3, HRG_PHASE_COMPLETE = 4 };
/* HumanRobotHandoverPhase, human_robot_handover_cartesian_env.py:50-56 (same numbering) */
enum { HRG_PHASE_PRESENT = 1, HRG_PHASE_WAIT = 2 };
/* RobotHumanHandoverPhase, robot_human_handover_cartesian_env.py:49-55 */
enum { HRG_R2H_APPROACH = 0, HRG_R2H_REACH_OUT = 1, HRG_R2H_RETREAT = 2, HRG_R2H_COMPLETE = 3 };
/* CollaborativeStackingPhase, collaborative_stacking_cartesian_env.py:52-60 */
enum { HRG_STK_APPROACH = 0, HRG_STK_PLACE_FIRST = 1, HRG_STK_WAIT_FOR_SECOND = 2, HRG_STK_PLACE_THIRD = 3, HRG_STK_WAIT_FOR_FOURTH = 4, HRG_STK_RETREAT = 5,
       HRG_STK_COMPLETE = 6 };
/* CollaborativeHammeringPhase, collaborative_hammering_cartesian_env.py:48-53 */
enum { HRG_HM_APPROACH = 0, HRG_HM_PRESENT = 1, HRG_HM_RETREAT = 3, HRG_HM_COMPLETE = 4 };
/* free bodies / collision geoms of the hammering task: geom GEOM_BOX + g, body BODY_BOX + b (the nail's contacts act on the board's DoF and the slide joint) */
enum { HRG_HM_BOARD = 0, HRG_HM_HAMMER = 1, HRG_HM_NAIL = 2 };
enum { HRG_HG_BOARD = 0, HRG_HG_HANDLE = 1, HRG_HG_HEAD = 2, HRG_HG_NAIL = 3, HRG_HM_NGEOM = 4 };
/* cube indices of the stacking task (order of `self.objects`, 1176-1180): the robot's two cubes, then the cubes in the human's hands */
enum { HRG_CUBE_A = 0, HRG_CUBE_B = 1, HRG_CUBE_L = 2, HRG_CUBE_R = 3 };

enum { HRG_GEOM_ROBOT = 0, HRG_GEOM_HUMAN = 1, HRG_GEOM_ALLOWED = 2, HRG_GEOM_STATIC = 3 };

typedef enum {
  HRG_OK = 0,
  HRG_ERR_INVALID = -1,
  HRG_ERR_HIP = -2,
  HRG_ERR_NOMEM = -3,
  HRG_ERR_UNSUPPORTED = -4
} hrg_status;

/* ------------------------------------------------------------------------------------------ model (POD) */

/* One constant-model description: kinematic/inertial tables, collision capsules, controller and shield
 * parameters, env (task) parameters.  Plain doubles/ints so that ctypes can fill it. */
typedef struct hrg_model_desc {
  /* ---- robot tree: 8 moving bodies (link1..link6, finger_l, finger_r), each with one joint ---------- */
  double base_pos[3];           /* world pose of the robot root (link0 frame) */
  double base_quat[4];          /* (w,x,y,z) */
  double body_pos[HRG_NV][3];   /* frame offset in parent body frame */
  double body_quat[HRG_NV][4];
  int32_t body_parent[HRG_NV];  /* index of parent moving body, -1 = base */
  int32_t jnt_type[HRG_NV];     /* 0 hinge, 1 slide */
  double jnt_axis[HRG_NV][3];   /* in body frame */
  double jnt_range[HRG_NV][2];
  double jnt_damping[HRG_NV];
  double jnt_frictionloss[HRG_NV];
  double jnt_armature[HRG_NV];
  double body_mass[HRG_NV];     /* welded children already merged in */
  double body_com[HRG_NV][3];   /* in body frame */
  double body_inertia[HRG_NV][6]; /* about com, body frame: xx yy zz xy xz yz */
  double dof_invweight0[HRG_NV];  /* (M^-1)_ii at qpos0 (MuJoCo dof_invweight0) */
  double body_invweight0[HRG_NV]; /* trace(Jv M^-1 Jv')/3 of the body com at qpos0 (MuJoCo body_invweight0, translational) */
  double gravity[3];
  double eef_pos[3];            /* grip site in the link6 body frame */
  /* actuation: arm motors (robot.xml:4-9), finger position servos */
  double arm_ctrlrange[HRG_NARM][2];
  double finger_kp;
  double finger_ctrlrange[HRG_NFINGER][2];
  double finger_forcerange[2];
  double finger_init_qpos[HRG_NFINGER];
  double gripper_speed;         /* RethinkGripper.format_action step */
  /* ---- constraint solver (MuJoCo-style soft constraints) -------------------------------------------- */
  double timestep;              /* opt.timestep = control_sample_time (human_env.py:332) */
  double solref[2];             /* timeconst, dampratio */
  double solimp[5];             /* dmin dmax width midpoint power */
  double contact_margin_human;  /* human.xml:5 margin */
  double friction_static;       /* tangential friction of robot-static contacts (pyramidal) */
  int32_t solver_iters;
  double solver_tol;
  /* ---- collision capsules ---------------------------------------------------------------------------- */
  int32_t rcap_body[HRG_NRCAP]; /* moving-body index, -1 = base (link0) */
  double rcap_p1[HRG_NRCAP][3];
  double rcap_p2[HRG_NRCAP][3];
  double rcap_r[HRG_NRCAP];
  uint32_t rcap_selfmask[HRG_NRCAP]; /* bit j set: pair (i,j), j>i, is a candidate self-collision pair */
  double table_top_z, table_half[2], floor_z;
  double table_center[2];        /* x, y of the table's centre (TableArena table_offset: (0, 0) for most tasks, (1.0, 0) for CollaborativeLiftingCart, 284) */
  /* ---- human -------------------------------------------------------------------------------------------- */
  int32_t hb_parent[HRG_NHB];
  int32_t hb_depth[HRG_NHB];
  double hb_anchor[HRG_NHB][3]; /* joint anchor = site position, model frame (human.xml site pos) */
  double hcap_p1[HRG_NHB][3];   /* collision capsule, model frame */
  double hcap_p2[HRG_NHB][3];
  double hcap_r[HRG_NHB];
  int32_t meas_body[HRG_NHJ];   /* measured joint k -> human body index (human.py:57-81 order) */
  int32_t site_lhand, site_rhand, site_head; /* indices into the measured-joint list */
  int32_t site_lelbow, site_relbow;
  double human_base_quat[4];    /* (w,x,y,z) of Rotation.from_quat([.5,.5,.5,.5]) human_env.py:373 */
  double base_human_pos_offset[3];
  double human_rand[3];
  /* ---- controller (failsafe.json, schunk.json) ----------------------------------------------------- */
  double kp, kd;
  double act_in_min, act_in_max, act_out_min, act_out_max;
  double qpos_limits[2][HRG_NARM];
  double init_qpos[HRG_NARM];
  double init_noise;            /* robosuite "default" initialization noise magnitude */
  /* ---- shield (synthetic stand-ins for sara-shield's three YAML files) ----------------------------- */
  int32_t shield_type;
  int32_t ltt_time_sync;        /* 1: the joints of a long-term trajectory arrive together, each stretched to the slowest joint's duration (sara-shield's LongTermPlanner,
                                 * SURVEY.md B.3); 0: every joint runs its own time-optimal profile (rounds 1-2: A/B of what the synchronisation costs the shield) */
  double v_max_allowed[HRG_NARM], a_max_allowed[HRG_NARM], j_max_allowed[HRG_NARM];
  double v_max_ltt[HRG_NARM], a_max_ltt[HRG_NARM], j_max_ltt[HRG_NARM];
  double path_amax, path_jmax;  /* limits on s'' and s''' of fail-safe / recovery manoeuvres */
  double failsafe_sdot;         /* path speed the fail-safe manoeuvre brakes to under SSM / OFF: 0 (full stop).  PFL computes its own every cycle (pfl_* below) */
  int32_t scap_body[HRG_NSHIELD_RCAP];
  double scap_p1[HRG_NSHIELD_RCAP][3];
  double scap_p2[HRG_NSHIELD_RCAP][3];
  double scap_r[HRG_NSHIELD_RCAP];
  double scap_alpha[HRG_NSHIELD_RCAP]; /* max Cartesian acceleration bound of the capsule end points */
  double secure_radius;
  int32_t n_bodypart;
  int32_t bp_joint[HRG_NBODYPART_MAX][2]; /* measured-joint indices (proximal, distal) */
  double bp_thickness[HRG_NBODYPART_MAX];
  double bp_vmax[HRG_NBODYPART_MAX];
  double bp_amax[HRG_NBODYPART_MAX];
  int32_t bp_in_pos[HRG_NBODYPART_MAX];   /* part keeps its VEL capsule in the POS model (torso, head) */
  int32_t n_extremity;
  int32_t ext_joint[HRG_NEXTREMITY_MAX];  /* proximal measured joint */
  double ext_length[HRG_NEXTREMITY_MAX];
  double ext_thickness[HRG_NEXTREMITY_MAX];
  double ext_vmax[HRG_NEXTREMITY_MAX];
  double meas_err_pos, meas_err_vel, delay;
  /* ---- task / env (reach_human.yaml, human_env.yaml) ---------------------------------------------- */
  int32_t n_cycles;             /* int(control_timestep / control_sample_time) human_env.py:503 */
  int32_t horizon;
  int32_t n_goals;              /* reach_human_env.py:318-321 */
  int32_t n_anim_ids;           /* human_env.py:379-382 */
  int32_t n_clips;
  double anim_step_length;      /* int(1/timestep)/human_animation_freq, human_env.py:1462-1465 */
  double goal_dist, reward_scale, task_reward, collision_reward, sim_crash_reward;
  int32_t reward_shaping, done_at_collision, done_at_success;
  double safe_vel, collision_debounce_delay;
  /* ---- static / self collision pre-check of a goal configuration (HumanEnv.check_collision_action, human_env.py:588-627;
   *      collision objects of _setup_collision_objects, human_env.py:1301-1348; pinocchio_manipulator_model.py:168-236) ---- */
  int32_t cp_enabled;           /* CollisionPreventionWrapper in the stack (config/wrappers/safe.yaml) */
  int32_t cp_replace_type;      /* 0 zero action, 1 random safe action, 2 closest safe action (collision_prevention_wrapper.py:25-46) */
  int32_t cp_n_resamples;
  int32_t goal_check;           /* ReachHuman._sample_valid_pos rejects colliding goals (reach_human_env.py:525-548) */
  double self_collision_safety; /* reach_human.yaml:25 */
  double obstacle_margin;       /* safety_margin of the table / base obstacles (reach_human_env.py:589-593) */
  double base_cyl_r, base_cyl_z; /* mount pedestal cylinder (human_env.py:1333-1339) */
  uint32_t chk_selfmask[HRG_NRCAP]; /* self-collision candidates of the pre-check model: capsules 0..6 + gripper cylinder (7) */
  /* ---- manipulation object + task of PickPlaceHumanCart (pick_place_human_cartesian_env.py:257-404, 637-708) ---- */
  int32_t task;                 /* HRG_TASK_* */
  int32_t n_obj_placements, n_targets; /* max(int(horizon * n_*_sampled_per_100_steps / 100), 1): 338-349 */
  double box_half[3];           /* half extents of the box object (object_full_size / 2) */
  double box_mass, box_inertia[3]; /* BoxObject default density 1000; principal inertia m (b^2 + c^2) / 3 per axis (half extents b, c) */
  double box_inertia_mean;      /* mean of box_inertia: the rotational inertia is handled as mean * identity + R diag(inertia - mean) R' */
  double box_invweight_rot;     /* body_invweight0 (rotation) of the free body: mean of 1 / box_inertia */
  /* ---- CollaborativeLiftingCart (collaborative_lifting_cartesian_env.py) ---- */
  double lift_anchor[2][3];     /* board-frame anchors of the left / right hand grips: connect equalities to the hand mocap bodies (786-795, 924-958) */
  double lift_grip_depth;       /* how far the board's robot-side edge reaches past the grip site along the gripper axis at a reset */
  double min_balance;           /* episode ends when (board normal . world up) falls below (509-533) */
  double imbalance_failure_reward, board_released_reward; /* _sparse_reward (446-478) */
  double obj_bin[4], tgt_bin[4]; /* xmin xmax ymin ymax of the sampling bins (843-875) */
  double obj_z, tgt_z;          /* z of a sampled object centre / target (UniformRandomSampler reference_pos + z_offset) */
  double object_gripped_reward;
  double object_at_target_reward, goal_exit_tolerance; /* HumanObjectInspectionCart, human_object_inspection_cartesian_env.py:318-321 */
  double object_in_human_hand_reward; /* RobotHumanHandoverCart, robot_human_handover_cartesian_env.py:530-555 */
  double finger_qpos_range[2][HRG_NFINGER]; /* RethinkValidGripper.qpos_range, rethink_valid_gripper.py:29-42 */
  /* ---- Cartesian action front-end (IKPositionDeltaWrapper, wrappers/ik_position_delta_wrapper.py:26-142;
   *      config/wrappers/ik_position_delta/default_ik_position_delta.yaml).  When enabled an action row is
   *      [dx, dy, dz, gripper, -, -, -] and is rewritten in place to the joint action it was converted to. ---- */
  int32_t ik_enabled;
  int32_t ik_max_iter;          /* max_iter (50) */
  int32_t ik_use_pos_limits;    /* x_position_limits is not None */
  double ik_action_limit;       /* action_limit (0.15): clip of the position delta */
  double ik_x_output_max;       /* x_output_max (1) */
  double ik_residual_threshold; /* residual_threshold (1e-3) */
  double ik_damping;            /* lambda of the damped least squares step (pybullet [UPSTREAM]: stand-in 0.1) */
  double ik_pos_limits[2][3];
  double ik_ee_offset[3];       /* end-effector link origin in the link-6 frame: fixed_gripper_joint of robot_pybullet.urdf (0, 0, 0.17) */
  double ik_target_rot[9];      /* end-effector orientation at init_qpos, held fixed (ik_position_delta_wrapper.py:74-82) */
  /* ---- CollaborativeStackingCart (collaborative_stacking_cartesian_env.py:316-480): box_half / box_mass / box_inertia describe each of the four cubes ---- */
  double stack_toppled_reward, second_cube_at_target_reward, fourth_cube_at_target_reward; /* _sparse_reward (700-744) */
  /* ---- PFL (power and force limiting): the fail-safe manoeuvre brakes to the path speed at which no point of the arm moves faster than pfl_v_safe on the
   *      trajectory actually planned: s'_pfl = min(1, pfl_v_safe / sum_j |dq_j/ds| pfl_reach[j]) (demos/demo_gym_functionality_Schunk_pfl_criterion.py:1-8) ---- */
  double pfl_v_safe;            /* Cartesian speed [m/s] the arm may keep while the reachable sets intersect */
  double pfl_reach[HRG_NARM];   /* largest distance of a point of the arm downstream of joint j from that joint (lever arm of its velocity) */
  double stack_weld_relpos[3];  /* relpose of the cube <-> hand mocap welds: mocap body origin in the cube frame, "0 0.045 0" (1263-1281) */
  /* ---- CollaborativeHammeringCart (collaborative_hammering_cartesian_env.py:291-381, 889-960, 1100-1145; models/assets/objects/nail.xml) ---- */
  double hm_board_half[3], hm_board_mass, hm_board_inertia[3]; /* BoxObject "board" (934-938): board_full_size / 2, default density 1000 */
  double hm_board_invweight_rot;  /* body_invweight0 (rotation) of the board: mean of 1 / inertia */
  double hm_anchor[2][3];         /* l_anchor / r_anchor: board-frame positions of lh_grip (connect) and rh_grip (weld), 1001-1002 */
  double hm_weld_relquat[4];      /* relpose quaternion of rh_eq (w, x, y, z), 1123-1131 */
  double hm_hammer_mass, hm_hammer_inertia[3]; /* HammerObject [UPSTREAM robosuite]: stand-in of two boxes (handle, head); inertia about the COM, body axes */
  double hm_hammer_invweight_rot;
  double hm_hammer_com[3];        /* COM in the hammer's body frame (origin = middle of the handle) */
  double hm_geom_pos[HRG_HM_NGEOM][3];  /* geom centres in their body frame: board (0), handle and head relative to the hammer's COM, nail head relative to the nail_head body */
  double hm_geom_half[HRG_HM_NGEOM][3];
  double hm_hammer_grip_quat[4];  /* world orientation the hammer is put into the gripper with: Ry(90 deg), _put_hammer_into_gripper (790-812) */
  double hm_finger_grip_qpos[HRG_NFINGER]; /* finger positions at a reset: pads touching the handle (the reference closes the gripper over 100 sim steps, 795-809) */
  double hm_nail_mass;            /* nail_head_g0, cylinder r 0.02 h 0.004 at density 1000 (nail.xml:5) */
  double hm_nail_z0;              /* nail_head body origin above the board's centre at joint position 0: placement z + 0.06 (nail.xml:3, 946-960) */
  double hm_nail_range;           /* slide joint range [0, 0.06], axis (0, 0, -1) of the board (nail.xml:7) */
  double hm_nail_frictionloss, hm_nail_fric_damping; /* frictionloss 10000, solreffriction (-100, -100): reference acceleration -damping / dmax * velocity */
  double hm_nail_invweight;       /* dof_invweight0 of the slide joint */
  double hm_nail_bin[4];          /* xmin xmax ymin ymax of the nail placements on the board (838-853) */
  double hm_goal_tolerance;       /* nail counts as hammered in when 1 - progress < goal_tolerance (505-520) */
  double hammer_gripped_reward_bonus, nail_hammered_in_reward; /* _sparse_reward (522-556) */
  /* ---- MuJoCo's noslip post-pass (mj_solNoSlip [UPSTREAM]); collaborative_hammering_cartesian_env.py:1161 sets noslip_iterations = 20, no other task does ---- */
  double noslip_tolerance;        /* opt.noslip_tolerance (MuJoCo default 1e-6): the pass ends when the scaled cost improvement of a sweep falls below it */
  double noslip_scale;            /* 1 / (stat.meaninertia * nv): the scale of that improvement (mean diagonal of M at qpos0 over the stepper's real DoF) */
  int32_t gripper_controllable;   /* False: the gripper action is replaced by 'close' (486-487) */
  int32_t noslip_iterations;      /* opt.noslip_iterations: sweeps of the pass at most; 0 = off (every task but CollaborativeHammeringCart) */
  /* ---- collision geometry of the seven arm links (robot.xml:29-55: mesh geoms, which MuJoCo convexifies at compile time) ----
   * robot_hulls = 1: contacts of an arm link with the human's capsules and with the table / floor planes are those of the link's CONVEX HULL (support mapping
   * over its vertices: GJK distance to a capsule's axis, deepest vertex under a plane); the link's bounding capsule is then only the broadphase.  0: the bounding
   * capsule itself is the collision geom (rounds 1-2).  Hull vertices: body frame, hull h = vertices hull_off[h] .. hull_off[h + 1] - 1 of
   * hull_verts[.][3] (host memory, copied at create like the clip frames; compiled from the STL files by tools/compile_model.py). */
  int32_t robot_hulls;
  int32_t hull_off[HRG_NHULL + 1];
  const double* hull_verts;
  uint64_t seed;
} hrg_model_desc;

/* Human animation clips, shared by all envs of a batch.  Frame layout (doubles):
 *   [0:3] Pelvis_pos_{x,y,z}   [3:7] Pelvis_quat (x,y,z,w — scipy order, convert_bvh.py:84-101)
 *   [7:76] 69 joint angles in qpos order of human.xml (body DFS order, per body z,y,x)
 * plus per-clip info (position_offset[3], orientation_quat[4] (x,y,z,w)): animation_utils.py:50-54 */
#define HRG_FRAME_DIM 76
typedef struct hrg_clip_table {
  int32_t n_clips;
  int32_t clip_len[HRG_MAX_CLIPS];     /* frames */
  int64_t clip_offset[HRG_MAX_CLIPS];  /* first frame index into `frames` */
  double clip_pos_offset[HRG_MAX_CLIPS][3];
  double clip_quat[HRG_MAX_CLIPS][4];  /* (x,y,z,w) */
  const double* frames;                /* host pointer, [total_frames][HRG_FRAME_DIM] */
  int64_t total_frames;
  /* per-clip entries of the animation info files the collaboration tasks read (human_object_inspection_cartesian_env.py:447-459,
   * 602-652; utils/animation_utils.py:122-176); zero for clips without them */
  int32_t clip_keyframes[HRG_MAX_CLIPS][2];
  double clip_target_pos[HRG_MAX_CLIPS][3];
  int32_t clip_n_loop[HRG_MAX_CLIPS];
  double clip_loop_amp[HRG_MAX_CLIPS][HRG_MAX_LOOP];
  double clip_loop_speed[HRG_MAX_CLIPS][HRG_MAX_LOOP];
  double clip_loop_amp_std[HRG_MAX_CLIPS], clip_loop_speed_std[HRG_MAX_CLIPS];
  int32_t clip_pointing_hand[HRG_MAX_CLIPS]; /* 0 right, 1 left ("pointing_hand" of the info file) */
  /* handover clips: two loop stages ("present" uses clip_loop_*, "wait" the arrays below) and the hand that holds the object
   * (human_robot_handover_cartesian_env.py:440-463) */
  int32_t clip_holding_hand[HRG_MAX_CLIPS];  /* 0 right, 1 left ("object_holding_hand") */
  int32_t clip_n_loop2[HRG_MAX_CLIPS];
  double clip_loop2_amp[HRG_MAX_CLIPS][HRG_MAX_LOOP];
  double clip_loop2_speed[HRG_MAX_CLIPS][HRG_MAX_LOOP];
  /* stacking clips (collaborative_stacking_cartesian_env.py:512-520, 825-897): five keyframes -- pass the table / release the first cube / first waiting
   * loop starts / release the third cube / second waiting loop starts; "first_placing_hand" is stored in clip_holding_hand, the loops of
   * "wait_for_second" / "wait_for_fourth" in clip_loop_* / clip_loop2_* */
  int32_t clip_stack_keyframes[HRG_MAX_CLIPS][5];
} hrg_clip_table;

typedef struct hrg_batch hrg_batch; /* opaque */

/* ----------------------------------------------------------------------------------------------- entry points */
const char* hrg_last_error(void);
const char* hrg_version(void);

/* size in bytes of one env's resident state block (for get/set_state buffers) */
size_t hrg_state_bytes(void);

/* Create n_envs environments with global ids [env_id0, env_id0+n_envs) on HIP device `device`.
 * Per-env random draws are keyed by (desc->seed, global env id, episode index), so the results do not
 * depend on how the global batch is sharded over GPUs. */
int hrg_batch_create(const hrg_model_desc* desc, const hrg_clip_table* clips, int32_t n_envs,
                     int64_t env_id0, int32_t device, hrg_batch** out);
void hrg_batch_destroy(hrg_batch* b);

/* Reset envs whose mask byte is non-zero (mask == NULL: all).  mask is a DEVICE pointer [n_envs].
 * obs_dev: float[n_envs][HRG_OBS_DIM], rows of reset envs are overwritten (others untouched). */
int hrg_batch_reset(hrg_batch* b, const uint8_t* mask_dev, float* obs_dev, void* stream);

/* One policy step of every env: n_cycles shield cycles, observation, reward, done, info, auto-reset.
 *   actions_dev  double[n_envs][HRG_ACT_DIM]  (with collision prevention on, rows are overwritten by the executed
 *                action, the wrapper's info["action"])
 *   obs_dev      float[n_envs][HRG_OBS_DIM]   (observation AFTER auto-reset where done)
 *   term_obs_dev float[n_envs][HRG_OBS_DIM]   (observation BEFORE auto-reset; may be NULL)
 *   reward_dev   float[n_envs];  done_dev uint8_t[n_envs];  info_dev int32_t[n_envs][HRG_INFO_DIM]
 * Asynchronous with respect to the host. */
int hrg_batch_step(hrg_batch* b, double* actions_dev, float* obs_dev, float* term_obs_dev,
                   float* reward_dev, uint8_t* done_dev, int32_t* info_dev, void* stream);

/* Parity hooks (synchronous, host buffers). */
/* contacts of the LAST substep of the last step: pairs_host int32[n_envs][HRG_NCON_MAX][2], ncon_host int32[n_envs] */
int hrg_batch_contacts(hrg_batch* b, int32_t* pairs_host, int32_t* ncon_host);
/* reach capsules of the last shield cycle: robot double[n_envs][HRG_NSHIELD_RCAP][7],
 * human double[n_envs][HRG_NHCAP_MAX][7] (p1,p2,r), n_human int32[n_envs] */
int hrg_batch_capsules(hrg_batch* b, double* robot_host, double* human_host, int32_t* n_human_host);
/* launch order of the NEXT step (diagnostic; no reference counterpart): order_host int32[n_envs] = the env each workgroup will step, n_busy_host = how many of
 * them (from the front) were busy in the last step -- robot contacts or a fail-safe manoeuvre.  The order never changes what an env computes, only when its wave starts
 * (busy envs first: the step kernel ends with its slowest wave). */
int hrg_batch_launch_order(hrg_batch* b, int32_t* order_host, int32_t* n_busy_host);
/* the capsule taps cost ~4 KB of HBM writes per env per shield cycle, so they are off unless enabled here */
int hrg_batch_enable_taps(hrg_batch* b, int32_t on);
int hrg_batch_get_state(hrg_batch* b, int32_t env, void* buf_host, size_t bytes);
int hrg_batch_set_state(hrg_batch* b, int32_t env, const void* buf_host, size_t bytes);
/* the manipulation object's part of the environment state (hrg_box_state, include/hrgym_state.h):
 * PickPlaceHumanCart.get/set_environment_state, pick_place_human_cartesian_env.py:946-975; zeros for ReachHuman */
/* batched form for reference-state initialisation (wrappers/dataset_wrapper.py:88-160 resets envs to dataset states):
 * states_host = n x hrg_env_state, boxes_host = n x hrg_box_state or NULL, for the envs listed in envs_host */
int hrg_batch_get_states(hrg_batch* b, const int32_t* envs_host, int32_t n, void* states_host, void* boxes_host);
int hrg_batch_set_states(hrg_batch* b, const int32_t* envs_host, int32_t n, const void* states_host, const void* boxes_host);
size_t hrg_box_bytes(void);
int hrg_batch_get_box(hrg_batch* b, int32_t env, void* buf_host, size_t bytes);
int hrg_batch_set_box(hrg_batch* b, int32_t env, const void* buf_host, size_t bytes);
/* the four cubes + task bookkeeping of CollaborativeStackingCart (hrg_stack_state, include/hrgym_state.h):
 * CollaborativeStackingCart.get/set_environment_state, collaborative_stacking_cartesian_env.py:1551-1600 */
size_t hrg_stack_bytes(void);
int hrg_batch_get_stack(hrg_batch* b, int32_t env, void* buf_host, size_t bytes);
int hrg_batch_set_stack(hrg_batch* b, int32_t env, const void* buf_host, size_t bytes);
/* board, hammer, nail + task bookkeeping of CollaborativeHammeringCart (hrg_hammer_state, include/hrgym_state.h):
 * CollaborativeHammeringCart.get/set_environment_state, collaborative_hammering_cartesian_env.py:1339-1377 */
size_t hrg_hammer_bytes(void);
int hrg_batch_get_hammer(hrg_batch* b, int32_t env, void* buf_host, size_t bytes);
int hrg_batch_set_hammer(hrg_batch* b, int32_t env, const void* buf_host, size_t bytes);
/* test tap of the hull variant (robot_hulls; oracle counterparts: hrgo_test_hull_segment / hrgo_test_hull_lowest): n queries {R[9] row-major, p[3], s1[3], s2[3], hull,
 * pad} = 152 bytes each, against the vertex table (verts_host, off_host[HRG_NHULL + 1]) -> out_host[n][10] = GJK distance hull - segment, witness on the hull 3, witness
 * on the segment 3, lowest point over a horizontal plane 3.  One wavefront per query runs the step kernel's own wave routines (csrc/hrgym_hull.h).  0 / -1. */
int hrg_test_hull_queries(const double* verts_host, const int32_t* off_host, const void* queries_host, int32_t n, double* out_host);

/* HumanEnv.check_collision_action (human_env.py:588-627; called by CollisionPreventionWrapper, wrappers/collision_prevention_wrapper.py:38-51, and
 * utils/training_utils.py:362-366): would the joint-space action drive the robot into the static scene or itself?  The goal configuration the
 * controller would set for each env (current joint angles + scaled action, clipped to the joint limits) is tested with the pre-check capsule
 * model; nothing is stepped.  actions: device, [n_envs][HRG_ACT_DIM] f64 (not modified); collides: device, [n_envs] u8 (1 = collision). */
int hrg_batch_check_actions(hrg_batch* b, const double* actions_dev, uint8_t* collides_dev, void* hip_stream);

/* Kernel timing hook for bench.py: records HIP events on the launch stream around every step kernel
 * since the last call; returns average kernel milliseconds and the number of launches measured. */
int hrg_batch_kernel_time(hrg_batch* b, double* avg_ms, int64_t* n_launches);

#ifdef __cplusplus
}
#endif
#endif /* HRGYM_H */
