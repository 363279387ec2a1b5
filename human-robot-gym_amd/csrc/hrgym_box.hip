// hrgym_box.hip — the stepper compiled a second time with the manipulation object (PickPlaceHumanCart,
// environments/manipulation/pick_place_human_cartesian_env.py): 14-DoF constrained system (robot tree + free cube), cube
// contacts, task logic.  Only the kernels and their launch shims come out of this translation unit; the C ABI lives in
// hrgym_hip.hip, whose ReachHuman kernels carry none of this.
#define HRG_BOX 1
#include "hrgym_hip.hip"
