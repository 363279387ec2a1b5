// hrgym_lift.hip — the box variant compiled once more for CollaborativeLiftingCart (environments/manipulation/
// collaborative_lifting_cartesian_env.py): two connect equalities between the board's grip points and the mocap bodies at the human's hands
// (6 general equality rows on the board's DoF), the board placed into the gripper at a reset, balance / grip termination rules.  A fourth
// translation unit, so the other kernels carry none of it.
#define HRG_BOX 1
#define HRG_LIFT 1
#include "hrgym_hip.hip"
