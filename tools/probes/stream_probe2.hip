// Follow-up store-stream experiments (see stream_probe.hip).
//  A. one-shot blocks, one 16-B store per thread, LINEAR block->address map      (reference, ~6.9 TB/s)
//  B. same, but block b writes where the persistent tile pattern would write:
//     isolates the ADDRESS ORDER from the persistence of the waves
//  C. persistent waves (1024 x 64 lanes, private 11 200-B tile per frame) with an
//     s_waitcnt vmcnt(N) after each frame's stores, N = 0 / 11 / 22 ...: isolates the
//     number of stores a wave keeps in flight
//  D. one-shot with 64-thread blocks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int THREADS>
__global__ __launch_bounds__(THREADS) void oneshot_linear(u32x4* dst, size_t nvec) {
  const size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x;
  if (i < nvec) dst[i] = u32x4{1u, 2u, 3u, 4u};
}

// block b = (frame t, tile w, piece j): tile-major dispatch order within a frame
__global__ __launch_bounds__(64) void oneshot_tiles(u32x4* dst, int n_tiles, int pieces, int tile_vec, int order) {
  const int b = blockIdx.x;
  int t, w, j;
  if (order == 0) {        // pieces of one tile are consecutive blocks (== linear)
    j = b % pieces; w = (b / pieces) % n_tiles; t = b / (pieces * n_tiles);
  } else {                 // piece j of ALL tiles, then piece j+1 ... (what lock-stepped persistent waves do)
    w = b % n_tiles; j = (b / n_tiles) % pieces; t = b / (pieces * n_tiles);
  }
  const int i = j * 64 + threadIdx.x;
  if (i < tile_vec) dst[((size_t)t * n_tiles + w) * tile_vec + i] = u32x4{1u, 2u, 3u, 4u};
}

// 256-thread one-shot blocks; wave v of block b writes piece j of tile w where the
// (t, j, w) order is piece-major: consecutive waves write addresses one TILE apart.
__global__ __launch_bounds__(256) void oneshot_tiles256(u32x4* dst, int n_tiles, int pieces, int tile_vec, int order) {
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6);   // global wave index
  int t, w, j;
  if (order == 0) { j = g % pieces; w = (g / pieces) % n_tiles; t = g / (pieces * n_tiles); }
  else            { w = g % n_tiles; j = (g / n_tiles) % pieces; t = g / (pieces * n_tiles); }
  const int i = j * 64 + (threadIdx.x & 63);
  if (i < tile_vec) dst[((size_t)t * n_tiles + w) * tile_vec + i] = u32x4{1u, 2u, 3u, 4u};
}

// One-shot block per (frame, tile): THREADS threads write the tile's `tile_vec`
// 16-byte chunks, thread i taking chunks i, i+THREADS, ...
template <int THREADS>
__global__ __launch_bounds__(THREADS) void oneshot_tile_block(u32x4* dst, int tile_vec) {
  u32x4* out = dst + (size_t)blockIdx.x * tile_vec;
  for (int i = threadIdx.x; i < tile_vec; i += THREADS) out[i] = u32x4{1u, 2u, 3u, (uint32_t)i};
}

template <int WAIT>
__global__ __launch_bounds__(64) void persistent(u32x4* dst, int n_tiles, int tile_vec, int T) {
  const int lane = threadIdx.x;
  for (int t = 0; t < T; ++t) {
    u32x4* out = dst + ((size_t)t * n_tiles + blockIdx.x) * tile_vec;
#pragma unroll 4
    for (int i = lane; i < tile_vec; i += 64) out[i] = u32x4{(uint32_t)t, 2u, 3u, 4u};
    if (WAIT == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (WAIT == 1) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    if (WAIT == 2) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
    if (WAIT == 3) asm volatile("s_waitcnt vmcnt(44)" ::: "memory");
  }
}

template <typename F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); f();
  hipEventRecord(a, 0);
  for (int i = 0; i < 10; ++i) f();
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 10;
}

int main() {
  const int T = 100, n_tiles = 1024, tile_vec = 700;
  const size_t nvec = (size_t)T * n_tiles * tile_vec;
  const double bytes = nvec * 16.0;
  u32x4* buf; if (hipMalloc((void**)&buf, nvec * 16 + 4096) != hipSuccess) return 1;
  float ms;
  ms = timeit([&] { hipLaunchKernelGGL(oneshot_linear<256>, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, buf, nvec); });
  printf("A  one-shot linear, 256-thread blocks      %.4f ms %.0f GB/s\n", ms, bytes / ms / 1e6);
  ms = timeit([&] { hipLaunchKernelGGL(oneshot_linear<64>, dim3((unsigned)((nvec + 63) / 64)), dim3(64), 0, 0, buf, nvec); });
  printf("D  one-shot linear, 64-thread blocks       %.4f ms %.0f GB/s\n", ms, bytes / ms / 1e6);
  ms = timeit([&] { hipLaunchKernelGGL(oneshot_linear<1024>, dim3((unsigned)((nvec + 1023) / 1024)), dim3(1024), 0, 0, buf, nvec); });
  printf("D' one-shot linear, 1024-thread blocks     %.4f ms %.0f GB/s\n", ms, bytes / ms / 1e6);
  for (int shift : {4, 8, 16, 32}) {   // in 16-byte units: 64, 128, 256, 512 bytes
    ms = timeit([&] { hipLaunchKernelGGL(oneshot_linear<256>, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, buf + shift, nvec - 64); });
    printf("A+ one-shot linear, base shifted by %4d B    %.4f ms %.0f GB/s\n", shift * 16, ms, bytes / ms / 1e6);
  }
  const int pieces = (tile_vec + 63) / 64;
  for (int order = 0; order < 2; ++order) {
    ms = timeit([&] { hipLaunchKernelGGL(oneshot_tiles, dim3((unsigned)(T * n_tiles * pieces)), dim3(64), 0, 0, buf, n_tiles, pieces, tile_vec, order); });
    printf("B%d one-shot 64-thr blocks, %s   %.4f ms %.0f GB/s\n", order, order ? "piece-major (tile stride)" : "tile-major (linear)      ", ms, bytes / ms / 1e6);
  }
  for (int order = 0; order < 2; ++order) {
    ms = timeit([&] { hipLaunchKernelGGL(oneshot_tiles256, dim3((unsigned)(T * n_tiles * pieces / 4)), dim3(256), 0, 0, buf, n_tiles, pieces, tile_vec, order); });
    printf("E%d one-shot 256-thr blocks, %s  %.4f ms %.0f GB/s\n", order, order ? "piece-major (tile stride)" : "tile-major (linear)      ", ms, bytes / ms / 1e6);
  }
  ms = timeit([&] { hipLaunchKernelGGL(oneshot_tile_block<256>, dim3((unsigned)(T * n_tiles)), dim3(256), 0, 0, buf, tile_vec); });
  printf("F  one-shot block per (frame,tile), 256 thr    %.4f ms %.0f GB/s\n", ms, bytes / ms / 1e6);
  ms = timeit([&] { hipLaunchKernelGGL(oneshot_tile_block<512>, dim3((unsigned)(T * n_tiles)), dim3(512), 0, 0, buf, tile_vec); });
  printf("F  one-shot block per (frame,tile), 512 thr    %.4f ms %.0f GB/s\n", ms, bytes / ms / 1e6);
  ms = timeit([&] { hipLaunchKernelGGL(oneshot_tile_block<704>, dim3((unsigned)(T * n_tiles)), dim3(704), 0, 0, buf, tile_vec); });
  printf("G  one-shot block per (frame,tile), 704 thr    %.4f ms %.0f GB/s\n", ms, bytes / ms / 1e6);
  ms = timeit([&] { hipLaunchKernelGGL(oneshot_tile_block<128>, dim3((unsigned)(T * n_tiles)), dim3(128), 0, 0, buf, tile_vec); });
  printf("F  one-shot block per (frame,tile), 128 thr    %.4f ms %.0f GB/s\n", ms, bytes / ms / 1e6);
  {
    const int tv = 1400;  // 128 environments x 175 B = 22 400 B = 175 whole 128-byte lines
    const unsigned nb = (unsigned)(T * n_tiles / 2);
    ms = timeit([&] { hipLaunchKernelGGL(oneshot_tile_block<256>, dim3(nb), dim3(256), 0, 0, buf, tv); });
    printf("H  one-shot block per (frame, 128-env tile), 256 thr   %.4f ms %.0f GB/s\n", ms, bytes / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(oneshot_tile_block<512>, dim3(nb), dim3(512), 0, 0, buf, tv); });
    printf("H  one-shot block per (frame, 128-env tile), 512 thr   %.4f ms %.0f GB/s\n", ms, bytes / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(oneshot_tile_block<1024>, dim3(nb), dim3(1024), 0, 0, buf, tv); });
    printf("H  one-shot block per (frame, 128-env tile), 1024 thr  %.4f ms %.0f GB/s\n", ms, bytes / ms / 1e6);
    const int tv2 = 2800;  // 256 environments
    ms = timeit([&] { hipLaunchKernelGGL(oneshot_tile_block<1024>, dim3(nb / 2), dim3(1024), 0, 0, buf, tv2); });
    printf("H  one-shot block per (frame, 256-env tile), 1024 thr  %.4f ms %.0f GB/s\n", ms, bytes / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(oneshot_tile_block<256>, dim3(nb / 2), dim3(256), 0, 0, buf, tv2); });
    printf("H  one-shot block per (frame, 256-env tile), 256 thr   %.4f ms %.0f GB/s\n", ms, bytes / ms / 1e6);
  }
  ms = timeit([&] { hipLaunchKernelGGL(persistent<-1>, dim3(n_tiles), dim3(64), 0, 0, buf, n_tiles, tile_vec, T); });
  printf("C  persistent, no wait                     %.4f ms %.0f GB/s\n", ms, bytes / ms / 1e6);
  ms = timeit([&] { hipLaunchKernelGGL(persistent<0>, dim3(n_tiles), dim3(64), 0, 0, buf, n_tiles, tile_vec, T); });
  printf("C0 persistent, vmcnt(0) per frame          %.4f ms %.0f GB/s\n", ms, bytes / ms / 1e6);
  ms = timeit([&] { hipLaunchKernelGGL(persistent<1>, dim3(n_tiles), dim3(64), 0, 0, buf, n_tiles, tile_vec, T); });
  printf("C1 persistent, vmcnt(11) per frame         %.4f ms %.0f GB/s\n", ms, bytes / ms / 1e6);
  ms = timeit([&] { hipLaunchKernelGGL(persistent<2>, dim3(n_tiles), dim3(64), 0, 0, buf, n_tiles, tile_vec, T); });
  printf("C2 persistent, vmcnt(22) per frame         %.4f ms %.0f GB/s\n", ms, bytes / ms / 1e6);
  ms = timeit([&] { hipLaunchKernelGGL(persistent<3>, dim3(n_tiles), dim3(64), 0, 0, buf, n_tiles, tile_vec, T); });
  printf("C3 persistent, vmcnt(44) per frame         %.4f ms %.0f GB/s\n", ms, bytes / ms / 1e6);
  return 0;
}
