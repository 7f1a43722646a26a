// k_misc.hip - small stand-alone kernels: action-id check, one-hot -> ids, the write-ceiling probe.

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

// The denominator SURVEY section 8(d) asks for beside the vendor peak: what THIS chip sustains for
// a pure stream of the render kernel's own stores (16 bytes per lane, `sc0 sc1 nt`, one-shot waves
// of one aligned 2 KiB window each, XCD-contiguous block order) with nothing to compute and nothing
// to read.  bench.py times it over exactly the bytes a rollout launch writes.
__global__ void __launch_bounds__(128) write_probe_kernel(u32x4* __restrict__ dst, int64_t n16, uint32_t value) {
  // (block b runs on XCD b % 8: give each XCD one contiguous eighth of the buffer, as render_kernel does)
  const int64_t nb = gridDim.x, per = nb / 8;
  const int64_t b = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  // a wave's window: 2 KiB = two rounds of 64 lanes x 16 bytes; a block: two waves, 4 KiB
  const int64_t i = b * 256 + (threadIdx.x >> 6) * 128 + (threadIdx.x & 63);
  const u32x4 v{value, value, value, value};
  if (i < n16) store16_streaming(dst + i, v);
  if (i + 64 < n16) store16_streaming(dst + i + 64, v);
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

int32_t campx_write_probe_launch(void* dst, int64_t n_bytes, uint32_t value, void* stream) {
  if (!dst || n_bytes < 0 || (reinterpret_cast<uintptr_t>(dst) & 15) || (n_bytes & 15)) return CAMPX_EINVAL;
  if (n_bytes == 0) return CAMPX_OK;
  const int64_t n16 = n_bytes / 16;
  const int64_t blocks = ((n16 + 255) / 256 + 7) / 8 * 8;     // (4 KiB a block; a multiple of 8: the XCD order)
  if (blocks > 0x7fffffffll) return CAMPX_EINVAL;
  hipLaunchKernelGGL(write_probe_kernel, dim3((unsigned)blocks), dim3(128), 0,
                     static_cast<hipStream_t>(stream), static_cast<u32x4*>(dst), n16, value);
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
