// k_misc.hip - small stand-alone kernels: action-id check, one-hot -> ids.

#include "campx_common.hip.h"

namespace campx_impl {

__global__ void check_actions_kernel(const int8_t* __restrict__ actions, int64_t n,
                                     int32_t* bad_count) {
  int bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    bad += ((unsigned)actions[i] > 4u);
  if (bad) atomicAdd(bad_count, bad);
}

__global__ void onehot_to_ids_kernel(const float* __restrict__ onehot, int8_t* __restrict__ ids,
                                     int64_t n, int32_t* bad_count) {
  int bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    int id = 4, ones = 0, others = 0;
#pragma unroll
    for (int j = 0; j < CAMPX_N_ACTIONS; ++j) {
      const float v = onehot[i * CAMPX_N_ACTIONS + j];
      if (v == 1.0f) {
        id = j;
        ++ones;
      } else if (v != 0.0f) {
        ++others;
      }
    }
    const bool not_one_hot = ones != 1 || others != 0;
    bad += not_one_hot;
    // such a row becomes id 5: every kernel treats it as "stay" and reports it, so a
    // caller need not look at the count before stepping
    ids[i] = (int8_t)(not_one_hot ? CAMPX_N_ACTIONS : id);
  }
  if (bad) atomicAdd(bad_count, bad);
}

}  // namespace campx_impl

using namespace campx_impl;

extern "C" {

int32_t campx_check_actions_launch(const int8_t* actions, int64_t n, int32_t* bad_count,
                                   void* stream) {
  if (!actions || !bad_count || n < 0) return CAMPX_EINVAL;
  if (n == 0) return CAMPX_OK;
  const int64_t want = (n + 255) / 256;
  const unsigned grid = (unsigned)(want < 2048 ? want : 2048);
  hipLaunchKernelGGL(check_actions_kernel, dim3(grid), dim3(256), 0,
                     static_cast<hipStream_t>(stream), actions, n, bad_count);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? CAMPX_OK : hip_failed(e);
}

int32_t campx_onehot_to_ids_launch(const float* onehot, int8_t* ids, int64_t n,
                                   int32_t* bad_count, void* stream) {
  if (!onehot || !ids || !bad_count || n < 0) return CAMPX_EINVAL;
  if (n == 0) return CAMPX_OK;
  const int64_t want = (n + 255) / 256;
  const unsigned grid = (unsigned)(want < 2048 ? want : 2048);
  hipLaunchKernelGGL(onehot_to_ids_kernel, dim3(grid), dim3(256), 0,
                     static_cast<hipStream_t>(stream), onehot, ids, n, bad_count);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? CAMPX_OK : hip_failed(e);
}

}  // extern "C"
