// hrgym_hammer.hip — the kernel variant of CollaborativeHammeringCart: the sources of hrgym_hip.hip compiled with HRG_HAMMER=1 (the board with the nail on its
// slide joint, the two-box hammer, box-box contacts of boxes with different extents, a connect + a weld at the human's hands, a 24-DoF Newton step in three
// 8-wide blocks with two constraint rows per lane).  Its own translation unit, so none of its registers, LDS or code reaches the other tasks' kernels.
#define HRG_HAMMER 1
#undef HRG_WG_WAVES
#define HRG_WG_WAVES 1   // 33 KB of LDS per env: one env per workgroup
#include "hrgym_hip.hip"
