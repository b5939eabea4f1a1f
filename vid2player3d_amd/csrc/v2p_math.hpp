// Device-side fp32 vector / quaternion helpers for gfx950 (one value per lane, no arrays
// that would force scratch).  Quaternions are xyzw.  The task-side helpers restate
// embodied_pose/utils/torch_utils.py with the same thresholds so results match the
// reference within float32 rounding.
#pragma once
#include <hip/hip_runtime.h>

namespace v2p {
#include "v2p_math.inc"
}  // namespace v2p
