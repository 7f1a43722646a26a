// Store-stream micro-benchmark: what write bandwidth does the rollout kernel's
// access pattern (W waves, each writing a private 11 200-byte tile per frame into a
// [T, B*175] trajectory buffer) reach when there is NO game logic at all?
// Build: hipcc --offload-arch=gfx950 -O3 stream_probe.hip -o stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// MODE 0: registers -> global. MODE 1: LDS tile -> global (ds_read_b128 each frame).
template <int MODE, bool NT, int WAVES_PER_BLOCK>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void probe(int8_t* dst, int64_t t_stride, int tile_bytes,
                                                             int64_t tile_pitch, int T) {
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t tile = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave;
  const int nvec = tile_bytes >> 4;
  u32x4* my = reinterpret_cast<u32x4*>(lds + (size_t)wave * ((tile_bytes + 15) & ~15));
  if (MODE == 1) {
    for (int i = lane; i < nvec; i += 64) my[i] = u32x4{(uint32_t)i, 1u, 2u, 3u};
    __syncthreads();
  }
  for (int t = 0; t < T; ++t) {
    u32x4* out = reinterpret_cast<u32x4*>(dst + t * t_stride + tile * tile_pitch);
    if (MODE == 1) {
      if (lane == 0) reinterpret_cast<int8_t*>(my)[t & 127] = (int8_t)t;  // a "patch"
      __syncthreads();
    }
#pragma unroll 4
    for (int i = lane; i < nvec; i += 64) {
      u32x4 v = (MODE == 1) ? my[i] : u32x4{(uint32_t)t, (uint32_t)i, 2u, 3u};
      if (NT) __builtin_nontemporal_store(v, &out[i]); else out[i] = v;
    }
    if (MODE == 1) __syncthreads();
  }
}

template <int MODE, bool NT, int WPB>
float run(int8_t* buf, int n_tiles, int tile_bytes, int tile_pitch, int T, int reps, bool tile_major) {
  int64_t t_stride = (int64_t)n_tiles * tile_pitch;
  if (tile_major) { t_stride = tile_pitch; tile_pitch = tile_pitch * T; }
  const size_t shmem = MODE == 1 ? (size_t)WPB * ((tile_bytes + 15) & ~15) : 0;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; ++i)
    hipLaunchKernelGGL((probe<MODE, NT, WPB>), dim3(n_tiles / WPB), dim3(64 * WPB), shmem, 0, buf, t_stride, tile_bytes, tile_pitch, T);
  hipEventRecord(a, 0);
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL((probe<MODE, NT, WPB>), dim3(n_tiles / WPB), dim3(64 * WPB), shmem, 0, buf, t_stride, tile_bytes, tile_pitch, T);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

__global__ void fill_like(u32x4* dst, size_t nvec) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = u32x4{1u, 2u, 3u, 4u};
}

template <int PER_THREAD, bool NT>
__global__ __launch_bounds__(256) void fill_oneshot(u32x4* dst, size_t nvec) {
  const size_t base = (size_t)blockIdx.x * (256 * PER_THREAD) + threadIdx.x;
#pragma unroll
  for (int j = 0; j < PER_THREAD; ++j) {
    const size_t i = base + (size_t)j * 256;
    if (i < nvec) { u32x4 v = u32x4{1u, 2u, 3u, (uint32_t)j}; if (NT) __builtin_nontemporal_store(v, &dst[i]); else dst[i] = v; }
  }
}

template <int PER_THREAD, bool NT>
float run_oneshot(u32x4* buf, size_t nvec) {
  const unsigned grid = (unsigned)((nvec + 256 * PER_THREAD - 1) / (256 * PER_THREAD));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((fill_oneshot<PER_THREAD, NT>), dim3(grid), dim3(256), 0, 0, buf, nvec);
  hipEventRecord(a, 0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((fill_oneshot<PER_THREAD, NT>), dim3(grid), dim3(256), 0, 0, buf, nvec);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 10;
}

int main() {
  const int T = 100;
  int8_t* buf; CHECK(hipMalloc((void**)&buf, (size_t)T * 4096 * 11264 / 4 + (1 << 20)));
  {
    const size_t nvec = (size_t)T * 1024 * 11200 / 16;
    for (int grid : {2048, 8192, 65536}) {
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipLaunchKernelGGL(fill_like, dim3(grid), dim3(256), 0, 0, (u32x4*)buf, nvec);
      hipEventRecord(a, 0);
      for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(fill_like, dim3(grid), dim3(256), 0, 0, (u32x4*)buf, nvec);
      hipEventRecord(b, 0); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
      printf("fill_like grid=%d: %.4f ms %.0f GB/s\n", grid, ms, nvec * 16.0 / ms / 1e6);
    }
  }
  {
    const size_t nvec = (size_t)T * 1024 * 11200 / 16;
    float ms;
    ms = run_oneshot<1, false>((u32x4*)buf, nvec); printf("oneshot 1/thread      : %.4f ms %.0f GB/s\n", ms, nvec * 16.0 / ms / 1e6);
    ms = run_oneshot<4, false>((u32x4*)buf, nvec); printf("oneshot 4/thread      : %.4f ms %.0f GB/s\n", ms, nvec * 16.0 / ms / 1e6);
    ms = run_oneshot<4, true>((u32x4*)buf, nvec);  printf("oneshot 4/thread nt   : %.4f ms %.0f GB/s\n", ms, nvec * 16.0 / ms / 1e6);
    ms = run_oneshot<16, false>((u32x4*)buf, nvec); printf("oneshot 16/thread     : %.4f ms %.0f GB/s\n", ms, nvec * 16.0 / ms / 1e6);
    ms = run_oneshot<16, true>((u32x4*)buf, nvec); printf("oneshot 16/thread nt  : %.4f ms %.0f GB/s\n", ms, nvec * 16.0 / ms / 1e6);
    ms = run_oneshot<64, false>((u32x4*)buf, nvec); printf("oneshot 64/thread     : %.4f ms %.0f GB/s\n", ms, nvec * 16.0 / ms / 1e6);
  }
  struct Cfg { const char* name; int n_tiles, tile_bytes, pitch; bool tile_major; };
  Cfg cfgs[] = {
    {"1024 tiles x 11200 B (boat race, 64 envs/wave)", 1024, 11200, 11200, false},
    {"1024 tiles x 11264 B (whole 128B lines)", 1024, 11264, 11264, false},
    {"1024 tiles x 11200 B TILE-MAJOR [tile][T]", 1024, 11200, 11200, true},
    {"1024 tiles x 11264 B TILE-MAJOR [tile][T]", 1024, 11264, 11264, true},
    {"4096 tiles x 2800 B", 4096, 2800, 2800, false},
    {"4096 tiles x 2816 B (whole lines)", 4096, 2816, 2816, false},
    {"4096 tiles x 2800 B TILE-MAJOR", 4096, 2800, 2800, true},
    {"512 tiles x 22400 B", 512, 22400, 22400, false},
    {"256 tiles x 44800 B", 256, 44800, 44800, false},
  };
  for (auto& c : cfgs) {
    const double bytes = (double)T * c.n_tiles * c.tile_bytes;
    printf("%s\n", c.name);
#define RUN(MODE, NT, WPB) { float ms = run<MODE, NT, WPB>(buf, c.n_tiles, c.tile_bytes, c.pitch, T, 10, c.tile_major); \
      printf("  mode=%s nt=%d waves/block=%d : %.4f ms  %.0f GB/s\n", MODE ? "lds " : "regs", NT, WPB, ms, bytes / ms / 1e6); }
    RUN(0, false, 1) RUN(0, true, 1) RUN(1, false, 1)
  }
  return 0;
}
