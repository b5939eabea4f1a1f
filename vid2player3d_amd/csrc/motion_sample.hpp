// Reference-motion sampler, device side.  Restates MotionLib.get_motion_state
// (embodied_pose/utils/motion_lib.py:164-266) + _calc_frame_blend (:427-436) +
// _local_rotation_to_dof (:460-488) for one (query, body) pair per thread, so that the two
// frame rows of a query (24 bodies x {3,4,4} floats, contiguous in the reference's [F,24,*]
// tables) are read by 24 adjacent lanes.
#pragma once
#include "v2p_internal.hpp"
#include "v2p_math.hpp"

namespace v2p {
#include "motion_sample.inc"
}  // namespace v2p
