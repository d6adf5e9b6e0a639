// placeholder -- replaced below by raycast / sample / marching cubes kernels
#include "tsdf_common.h"
extern "C" int tsdf_hip_raycast(tsdf_handle, const float *, const float *, int, float *) { return TSDF_HIP_E_UNSUPPORTED; }
extern "C" int tsdf_hip_sample(tsdf_handle, const float *, size_t, float *, float *, float *, uint8_t *) { return TSDF_HIP_E_UNSUPPORTED; }
extern "C" int tsdf_hip_march(tsdf_handle, float, int, uint64_t *) { return TSDF_HIP_E_UNSUPPORTED; }
extern "C" int tsdf_hip_march_fetch(tsdf_handle, float *, uint8_t *, uint64_t *) { return TSDF_HIP_E_UNSUPPORTED; }
