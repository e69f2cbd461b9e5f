// Exchanges between the lanes of a wave straight out of registers (DPP quad permutes / row shifts / rotate, ds_swizzle,
// ds_bpermute: no LDS storage) - the lane-bit moves of the window solve (wsolve.h) and of the in-row solve of a tile (tsolve.h).
#pragma once
#include <hip/hip_runtime.h>

namespace mmhn {

// value of the lane that differs in lane bit I, for the lanes that have the move (forward: bit set, the lane below;
// transposed: bit clear, the lane above); the other lanes get some finite value of the wave
template <int I, bool TR>
__device__ __forceinline__ int lane_nbr32(int v, int lane) {
  if constexpr (I == 0) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);         // quad_perm [1,0,3,2]
  else if constexpr (I == 1) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
  else if constexpr (I == 2) return __builtin_amdgcn_update_dpp(0, v, TR ? 0x104 : 0x114, 0xF, 0xF, true);   // row_shl:4 / row_shr:4
  else if constexpr (I == 3) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true);   // row_ror:8
  else if constexpr (I == 4) return __builtin_amdgcn_ds_swizzle(v, 0x401F);                     // swap the halves of 32 lanes
  else return __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, v);
}
template <int I, bool TR>
__device__ __forceinline__ double lane_nbr(double v, int lane) {
  return __hiloint2double(lane_nbr32<I, TR>(__double2hiint(v), lane), lane_nbr32<I, TR>(__double2loint(v), lane));
}
template <int I, bool TR>
__device__ __forceinline__ float lane_nbr(float v, int lane) {
  return __int_as_float(lane_nbr32<I, TR>(__float_as_int(v), lane));
}
}  // namespace mmhn
