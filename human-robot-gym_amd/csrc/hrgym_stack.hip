// hrgym_stack.hip — the kernel variant of CollaborativeStackingCart: the sources of hrgym_hip.hip compiled with HRG_STACK=1 (four free cubes, box-box
// contacts, two cube <-> hand welds, a 32-DoF Newton step with two constraint rows per lane).  Its own translation unit, so none of its registers, LDS or
// code reaches the other tasks' kernels.
#define HRG_STACK 1
#undef HRG_WG_WAVES
#define HRG_WG_WAVES 1   // 28 KB of LDS per env: one env per workgroup
#include "hrgym_hip.hip"
