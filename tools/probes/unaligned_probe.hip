// Do 16-byte global stores / loads at addresses that are not 16- (or even 4-) byte aligned work
// on this stack (gfx950, ROCm KFD "unaligned" memory mode), and what do they cost?
//   hipcc --offload-arch=gfx950 -O3 unaligned_probe.hip -o unaligned_probe && ./unaligned_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void fill(uint8_t* base, size_t n_chunks, int misalign) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_chunks) return;
  const uint32_t v = (uint32_t)i * 2654435761u;
  const u32x4 w = {v, v + 1u, v + 2u, v + 3u};
  u32x4* p = reinterpret_cast<u32x4*>(base + misalign + 16 * i);
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
}

int main() {
  const size_t n = (size_t)64 << 20;   // 1 GiB of chunks
  uint8_t* d = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&d), 16 * n + 64) != hipSuccess) return 1;
  uint8_t* h = static_cast<uint8_t*>(malloc(4096 + 64));
  for (int mis : {0, 4, 8, 1, 3, 7, 13}) {
    hipMemset(d, 0xee, 4096 + 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    fill<<<(unsigned)(n / 256), 256>>>(d, n, mis);   // warm
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) fill<<<(unsigned)(n / 256), 256>>>(d, n, mis);
    hipEventRecord(e1);
    hipError_t err = hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, d, 4096 + 64, hipMemcpyDeviceToHost);
    int bad = 0;
    for (size_t i = 0; i < 256; ++i) {
      const uint32_t v = (uint32_t)i * 2654435761u;
      uint32_t w[4];
      memcpy(w, h + mis + 16 * i, 16);
      bad += !(w[0] == v && w[1] == v + 1u && w[2] == v + 2u && w[3] == v + 3u);
    }
    printf("misalign %2d: %s, %d bad chunks of 256, %.1f GB/s\n", mis, hipGetErrorString(err), bad,
           16.0 * n * 5 / (ms * 1e-3) / 1e9);
  }
  return 0;
}
