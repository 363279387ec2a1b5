// hrgym_handover.hip — the cube variant compiled once more with what only the handover tasks need (HumanRobotHandoverCart,
// environments/manipulation/human_robot_handover_cartesian_env.py): the weld equality between the object and the mocap body at the
// human's holding hand (6 equality rows), the second physics step per cycle of its _control_human, the hand mocap pose, the
// handover phase machine.  A third translation unit, so the pick-place / inspection kernels carry none of it.
#define HRG_BOX 1
#define HRG_HANDOVER 1
// two waves per SIMD (256 registers): at three the two-pass cycle body spills 648 B per lane inside the cycle loop, which moved 2.9 GB of scratch
// through HBM per 4096-env launch (98.6 MB algorithmic) and was 9 % slower (profiles/r01t_tasks_summary.md)
#ifndef HRG_BOX_WAVES
#define HRG_BOX_WAVES 2
#endif
#include "hrgym_hip.hip"
