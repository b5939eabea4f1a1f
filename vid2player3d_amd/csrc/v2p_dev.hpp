// Device-side helpers of the task kernels in namespace v2p: fp32 vector / quaternion math (v2p_math.inc: one value per lane, no arrays that
// would force scratch; quaternions xyzw; the task-side helpers restate embodied_pose/utils/torch_utils.py with the same thresholds) and the
// reference-motion sampler (motion_sample.inc: MotionLib.get_motion_state, utils/motion_lib.py:164-266).  The bodies live in .inc files
// because physics_ll.hip includes them a second time into its `strict` namespace, compiled under precise floating-point pragmas.
#pragma once
#include <hip/hip_runtime.h>

#include "v2p_internal.hpp"

namespace v2p {
#include "v2p_math.inc"
#include "motion_sample.inc"
}  // namespace v2p
