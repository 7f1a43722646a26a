// What does the shortest possible kernel cost on this stack?  Run under
//   rocprofv3 --kernel-trace --stats -- ./empty_kernel_probe
// and compare its duration with step_table_kernel's 3.2 us for one workgroup.
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void empty_kernel() {}
__global__ void one_store_kernel(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) *p = 1; }
__global__ void load_store_kernel(const int* q, int* p) { p[threadIdx.x] = q[threadIdx.x] + 1; }

int main() {
  int* d = nullptr;
  hipMalloc(reinterpret_cast<void**>(&d), 4096);
  hipMemset(d, 0, 4096);
  for (int i = 0; i < 200; ++i) {
    empty_kernel<<<1, 64>>>();
    one_store_kernel<<<1, 64>>>(d);
    load_store_kernel<<<1, 64>>>(d, d + 512);
  }
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < 1000; ++i) empty_kernel<<<1, 64>>>();
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("1000 back-to-back empty kernels: %.2f us each (launch-to-launch)\n", ms);
  return 0;
}
