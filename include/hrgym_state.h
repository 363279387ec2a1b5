/* hrgym_state.h — layout of one environment's resident state block.
 *
 * This is the byte image exchanged by hrg_batch_get_state / hrg_batch_set_state (the batched equivalent of
 * HumanEnvState / ReachHumanEnvState, human_env.py:80-103, reach_human_env.py:33-54: flattened sim state
 * + animation / goal bookkeeping) and the block the step kernel streams HBM -> LDS -> HBM once per policy
 * step.  All doubles first, then int32s; sizeof is a multiple of 8.
 */
#ifndef HRGYM_STATE_H
#define HRGYM_STATE_H

#include "hrgym.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Long-term trajectory: per joint an initial state and HRG_LTT_NSEG constant-jerk segments
 * (replaces sara-shield's sampled LongTermTraj; evaluated in closed form at path parameter s). */
typedef struct hrg_ltt {
  double q0[HRG_NARM], v0[HRG_NARM], a0[HRG_NARM];
  double qT[HRG_NARM]; /* goal configuration (state after the last segment) */
  double dur[HRG_NARM][HRG_LTT_NSEG];
  double jerk[HRG_NARM][HRG_LTT_NSEG];
  double T; /* duration of the slowest joint */
} hrg_ltt;

/* Three-phase constant-jerk profile of the path parameter s (sara-shield `Path`). */
typedef struct hrg_path {
  double s0, v0, a0;
  double dur[3];
  double jerk[3];
  double k; /* whole sample steps elapsed since the profile start */
} hrg_path;

typedef struct hrg_env_state {
  /* ---- MjSimState slice that is dynamic in this model (robot tree) ---- */
  double qpos[HRG_NV], qvel[HRG_NV], qacc_warmstart[HRG_NV];
  double time;
  /* ---- controller (FailsafeController / SingleArm) ---- */
  double goal_qpos[HRG_NARM];
  double mass_matrix[HRG_NARM * HRG_NARM]; /* stale 6x6 block, refreshed on policy steps only */
  double grip_action;                      /* RethinkGripper.current_action */
  /* ---- shield ---- */
  hrg_ltt ltt;
  hrg_path safe_path;       /* last verified fail-safe profile */
  double path_s, path_v, path_a; /* current path state on `ltt` */
  double new_goal_q[HRG_NARM];
  double meas_prev[HRG_NHJ][3];
  double meas_prev_t;
  double des_q[HRG_NARM], des_v[HRG_NARM], des_a[HRG_NARM]; /* last Motion returned by the shield */
  /* ---- human ---- */
  double human_site[HRG_NHJ][3]; /* world positions of the 23 joint sites at the current pose */
  double human_pos_offset[3];
  double human_rot_offset[4];    /* (w,x,y,z) */
  double debounce_timer;
  double eef_pos[3];             /* grip-site position of the last forward pass (observable source) */
  double cur_goal[HRG_NARM];     /* _desired_goals[_desired_goals_index] */
  /* ---- integers ---- */
  int32_t timestep;          /* policy steps in this episode */
  int32_t low_level_time;    /* human_env.py:526 */
  int32_t anim_index;        /* _human_animation_ids_index */
  int32_t anim_start_time;
  int32_t animation_time;
  int32_t goal_index;        /* _desired_goals_index */
  int32_t episode;           /* episode counter = RNG key */
  int32_t new_goal;          /* shield: goal pending */
  int32_t is_safe;
  int32_t n_meas;            /* measurements received since reset (0,1,2+) */
  int32_t failsafe_interventions;
  int32_t n_collisions_static, n_collisions_robot, n_collisions_human, n_collisions_critical;
  int32_t n_goal_reached;
  int32_t action_resamples;
  int32_t n_prev;                    /* previous_robot_collisions (cantor hashes of both orders) */
  int32_t prev_pairs[HRG_NPREV_MAX];
  int32_t ncon;                      /* contacts of the last substep (parity hook) */
  int32_t con_pairs[HRG_NCON_MAX][2];
  int32_t stream_id;                 /* key of the episode's random streams (with `episode`): the env's global id at reset; travels
                                      * with a copied state so that a restored episode keeps its animation / goal / placement sequence */
} hrg_env_state;

/* The manipulation object of the tasks that have one (PickPlaceHumanCart).  Kept out of hrg_env_state so that ReachHuman
 * neither streams nor holds it; get/set_state exchange hrg_env_state followed by hrg_box_state. */
typedef struct hrg_box_state {
  double pos[3], quat[4];       /* free joint qpos (w,x,y,z) */
  double vel[6];                /* linear, angular velocity (world frame) */
  double acc_warmstart[6];
  double obs_pos[3];            /* body_xpos of the last forward pass (what the observables read) */
  double target[3];
  int32_t obj_index, tgt_index; /* _object_placements_list_index, _target_positions_index */
  int32_t gripped;              /* _check_grasp at the last substep */
  int32_t task_phase;           /* HRG_PHASE_* (HumanObjectInspectionCart) */
  int32_t n_delayed;            /* _n_delayed_timesteps: frames the loop phase held the animation back (lifting: _n_steps_without_gripped_board) */
  int32_t n_delayed2;           /* second entry of _n_delayed_timesteps (handover: delay accumulated in the WAIT loop) */
  int32_t weld_active;          /* eq_active of the object <-> hand mocap weld (handover tasks) */
  int32_t n_handed_over;        /* _n_object_handed_over */
  double mocap_pos[3], mocap_quat[4]; /* pose of the mocap body at the human's holding hand (set once per cycle); lifting: mocap_pos = left-hand mocap body,
                                       * weld_off = right-hand mocap body (positions only: connect equalities) */
  double weld_off[3], weld_rel[4];    /* relative pose of the weld: object origin in the mocap frame, q_mocap^-1 q_obj (identity when the human
                                       * picks the object up from its own hand; taken at the palm contact in RobotHumanHandoverCart, 730-748) */
} hrg_box_state;

/* CollaborativeStackingCart: four cubes with free joints, the two hand mocap bodies their welds pull towards, and the task's bookkeeping
 * (CollaborativeStackingEnvState, collaborative_stacking_cartesian_env.py:63-97).  Kept in its own HBM array, streamed only by that task's kernel;
 * get/set through hrg_batch_get_stack / hrg_batch_set_stack. */
typedef struct hrg_stack_state {
  double pos[HRG_NCUBE][3], quat[HRG_NCUBE][4]; /* free joint qpos, (w,x,y,z); cube order HRG_CUBE_A, _B, _L, _R */
  double vel[HRG_NCUBE][6];                     /* linear, angular velocity (world frame) */
  double acc_warmstart[HRG_NCUBE][6];
  double obs_pos[HRG_NCUBE][3];                 /* body_xpos of the last forward pass (what observables, targets and the toppled test read) */
  double mocap_pos[2][3], mocap_quat[2][4];     /* lh_mocap_object, rh_mocap_object (905-936), set once per cycle */
  double target[3];                             /* next_target_position of the last epilogue (522-548); the eef position while there is none (1347-1355) */
  int32_t obj_index;                            /* _object_placements_list_index */
  int32_t gripped;                              /* object_gripped sensor (1467-1481) at the last substep */
  int32_t task_phase;                           /* HRG_STK_* */
  int32_t n_delayed[2];                         /* _n_delayed_timesteps */
  int32_t weld_active[2];                       /* eq_active of lh_weld_eq, rh_weld_eq */
  int32_t n_stack, stack_ids[HRG_NCUBE];        /* _object_stack_body_ids, bottom to top (cube indices) */
  int32_t max_stack_height;                     /* _max_stack_height */
  int32_t has_target;                           /* next_target_position is not None */
  int32_t pad;
} hrg_stack_state;

/* CollaborativeHammeringCart: the board the human holds (free joint; the nail rides on it on a slide joint), the hammer in the robot's gripper (free
 * joint), the two hand mocap bodies, and the task's bookkeeping (CollaborativeHammeringEnvState, collaborative_hammering_cartesian_env.py:56-89).  Its own
 * HBM array, streamed only by that task's kernel; get/set through hrg_batch_get_hammer / hrg_batch_set_hammer. */
typedef struct hrg_hammer_state {
  double pos[2][3], quat[2][4];      /* HRG_HM_BOARD, HRG_HM_HAMMER: centre of mass, orientation (w,x,y,z) */
  double vel[2][6];                  /* linear, angular velocity (world frame) */
  double acc_warmstart[2][6];
  double nail_q, nail_v, nail_acc_warmstart; /* nail_head_joint0: how far the nail is driven in [m], 0 .. hm_nail_range */
  double nail_xy[2];                 /* current nail placement on the board (board frame): nail_placements[nail_placements_index] */
  double obs_pos[3][3];              /* body_xpos of the last forward pass: board_main, hammer root body (middle of the handle), nail_head */
  double mocap_pos[2][3], mocap_quat[2][4]; /* lh_mocap, rh_mocap (688-715), set once per cycle */
  int32_t nail_index;                /* _nail_placements_index */
  int32_t gripped;                   /* hammer_gripped sensor (1283-1289) at the last substep */
  int32_t task_phase;                /* HRG_HM_* */
  int32_t n_delayed;                 /* _n_delayed_timesteps */
  int32_t nail_touch;                /* diagnostic: who has touched the nail head at the end of ANY substep since the nail was last pulled out -- bit 0 a hammer geom
                                      * (handle / head), bit 1 anything else (an arm link, the gripper); tools/soak_hammering.py: is a driven-in nail the hammer's doing */
  int32_t pad_;
} hrg_hammer_state;

#ifdef __cplusplus
}
#endif
#endif
