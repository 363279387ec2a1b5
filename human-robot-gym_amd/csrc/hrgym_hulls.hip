// hrgym_hulls.hip -- the ReachHuman kernels compiled once more with the arm links' CONVEX HULLS as collision geometry (hrg_model_desc.robot_hulls = 1; robot.xml:29-55:
// mesh geoms, which MuJoCo convexifies at compile time): every reported contact keeps its geometry in LDS (24 instead of 6), and a pass after the capsule narrowphase
// replaces the contacts of arm links with human capsules / the table and floor planes by those of the hulls (hrgym_hull.h: support mapping over the vertices with
// lanes = vertices, GJK).  Its own translation unit, so the capsule-geometry kernels carry none of it: no code, no LDS, no registers.
#define HRG_HULLS 1
#include "hrgym_hip.hip"
