// campx_hip.hip - batched CampX grid-world engine for MI355X (gfx950, CDNA4), and the
// C ABI declared in include/campx_hip.h.
//
// What one launch computes, per environment and per frame, is the reference's
// Engine.play() (campx/engine.py:114-166): every entity's update() in schedule
// order with one repaint per update group (engine.py:195-208), the Plot's reward /
// discount / game-over bookkeeping (campx/plot.py:161-211, engine.py:285-292) and
// the occluded layered-board render (campx/rendering.py:104-219).
//
// Kernels (DESIGN.md section 3 has the numbers):
//   rollout_kernel        rule interpreter + render, fused.  One lane = one
//                         environment, one wave = one workgroup = 64 environments;
//                         rules arrive in the kernarg segment (scalar loads/branches),
//                         scenery tables and the wave's 64 x L*H*W-byte output image
//                         live in LDS; a frame patches a few bytes of the image and
//                         streams it out, 1 KiB per wave-instruction.
//   rollout_table_kernel  same, for games with one moving thing: the update pass is a
//                         lookup in a (cell, action) table that campx_spec_compile()
//                         fills by running rollout_kernel over every pair.
//   update_*_kernel       the update pass alone (producer, consumer and loader waves)
//                         from the game's state table: (cell, action) in LDS for one
//                         mover, (cell, cell, action) in LDS for two, (cell, ..., action)
//                         in global memory for three and four,
//   render_kernel         and the observation stream alone: one-shot blocks, every
//                         wave one aligned KiB store - the store pattern that reaches
//                         the chip's HBM write ceiling.  The default for rollouts.
//   step_*_kernel         Engine.play(): one frame, one-shot, one wave per 64 environments.
//   shape_rollout_kernel  Hello-World-style games (rigidly translated multi-cell things):
//                         one wave per environment, scalar update pass.
// No MFMA anywhere: the path has no contraction; every kernel is bound by the HBM
// write stream of observations.
//
// Compiled with -ffp-contract=off: rewards are sums of a few float terms and must
// round exactly like the reference's float32 tensor arithmetic.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <utility>
#include <stdlib.h>
#include <string.h>

#include "campx_hip.h"

namespace {

constexpr int kWave = 64;
// Actions are staged through LDS kChunk frames at a time, so that the frame loop
// itself issues no global loads: a load in the loop would make every frame wait
// (vmcnt is in-order) for the previous frames' observation stores to drain.
constexpr int kChunk = 64;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// The streaming 16-byte store of the observation writers.  CAMPX_NT_FLAVOR picks the
// cache policy (A/B builds): 1 = nt, 2 = sc1, 3 = sc0 sc1, 4 = sc0 sc1 nt (default:
// system-scope write-through + non-temporal, i.e. the line is not kept anywhere on
// its way to HBM).  Measured on the boat-race bench, ms per 100-frame launch,
// fused / split path: plain 0.254 / 0.336, nt 0.233 / 0.220, sc1 0.259 / 0.247,
// sc0 sc1 0.261 / 0.250, sc0 sc1 nt 0.224 / 0.196.
#ifndef CAMPX_NT_FLAVOR
#define CAMPX_NT_FLAVOR 4
#endif
__device__ __forceinline__ void store16_streaming(u32x4* p, u32x4 v) {
#if CAMPX_NT_FLAVOR == 1
  __builtin_nontemporal_store(v, p);
#elif CAMPX_NT_FLAVOR == 2
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#elif CAMPX_NT_FLAVOR == 3
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#else
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#endif
}

// The same store with the address as (wave-uniform 64-bit base in SGPRs) + (32-bit byte
// offset per lane): no 64-bit vector address arithmetic.  `base` must be provably uniform
// (kernel arguments, blockIdx).
__device__ __forceinline__ void store16_streaming_at(const void* base, uint32_t off, u32x4 v) {
#if CAMPX_NT_FLAVOR == 4
  asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1 nt\n\ts_nop 1" ::"v"(off), "v"(v), "s"(base)
               : "memory");
#else
  store16_streaming(reinterpret_cast<u32x4*>(const_cast<char*>(static_cast<const char*>(base)) + off), v);
#endif
}

// The part of the GameSpec the interpreter reads every frame.  Passed BY VALUE
// so that it lives in the kernarg segment (scalar loads, scalar branches).
struct RuleBlock {
  int32_t rows, cols, n_layers, n_dyn, n_rules, any_reward;
  int32_t perf_dyn, perf_n;
  int32_t dyn_layer[CAMPX_MAX_DYN];
  int32_t dyn_z[CAMPX_MAX_DYN];
  int32_t dyn_row0[CAMPX_MAX_DYN];
  int32_t dyn_col0[CAMPX_MAX_DYN];
  CampxRule rules[CAMPX_MAX_RULES];
};

struct LdsTables {
  const uint8_t* top_layer;   // [HW]
  const uint8_t* top_z;       // [HW]
  const uint16_t* cover;      // [HW]
};

// Positions of the (up to four) moving things of one environment, one byte each in a
// 32-bit word per coordinate.  Not arrays: a rule names its thing by a wave-uniform
// index, and a dynamically indexed private array (or vector) goes to scratch memory -
// every access a scratch load of several hundred cycles; the interpreter ran 6.7 us per
// frame that way.  Byte lanes are selected with a shift instead.
template <int K>
struct Things {
  uint32_t r, c, cell;   // cell = r * W + c, kept in step: most rules compare cells
};

template <int K>
__device__ __forceinline__ int sel(uint32_t v, int d) {
  return (int)((v >> (8 * d)) & 0xffu);
}

template <int K>
__device__ __forceinline__ void put(uint32_t& v, int d, int x) {
  v = (v & ~(0xffu << (8 * d))) | ((uint32_t)x << (8 * d));
}

template <int K>
__device__ __forceinline__ void set_pos(Things<K>& p, int d, int r, int c, int W) {
  put<K>(p.r, d, r);
  put<K>(p.c, d, c);
  put<K>(p.cell, d, r * W + c);
}

// Which tile of environments a workgroup owns.  Workgroup b is observed to run on
// XCD b % 8 (MI355X_MICROARCH.md); mode 1 gives each XCD a contiguous eighth of the
// batch so that what one L2 evicts is contiguous in memory.  Speed only: any
// bijection is correct.
__device__ __forceinline__ uint32_t tile_of_block(uint32_t b, uint32_t n, int mode) {
  if (mode == 1 && (n & 7u) == 0) return (b & 7u) * (n >> 3) + (b >> 3);
  return b;
}

// Hidden performance of a move between cell classes (0 = none, 1..n cyclic):
// +1 one class forward, -1 one class back (examples/boat_race.py:137-151).
__device__ __forceinline__ int class_progress(int from, int to, int n) {
  if (from == 0 || to == 0) return 0;
  const int fwd = (from == n) ? 1 : from + 1;
  const int back = (from == 1) ? n : from - 1;
  return (to == fwd) - (to == back);
}

// Index of the pair-table entries of (cell of thing 0, cell of thing 1).
__device__ __forceinline__ uint32_t pair_index(uint32_t c0, uint32_t c1, int HW) {
  return (c0 * (uint32_t)HW + c1) * CAMPX_N_ACTIONS;
}

// Trace entry of one moving thing at one frame (CampxOutputs.trace): the cell it is in
// and whether it is the character that cell shows.
__device__ __forceinline__ uint8_t pack_trace(int cell, uint32_t vis) {
  return (uint8_t)((uint32_t)cell | (vis << 7));
}

// Cyclic one-cell move: 0 left (col-1), 1 right, 2 up (row-1), 3 down, else stay
// (examples/boat_race.py:42-49).  Branch-free: the action differs per lane.
__device__ __forceinline__ void moved(int a, int H, int W, int r, int c, int& r2, int& c2) {
  const int dc = (a == 1) - (a == 0);
  const int dr = (a == 3) - (a == 2);
  c2 = c + dc;
  r2 = r + dr;
  c2 = (c2 < 0) ? W - 1 : ((c2 == W) ? 0 : c2);
  r2 = (r2 < 0) ? H - 1 : ((r2 == H) ? 0 : r2);
}

// Layer shown at `cell` when the dynamic things stand at `p`: the front-most of
// the static scenery there and any dynamic thing there (engine.py:306-324).
template <int K>
__device__ __forceinline__ int shown_layer(const RuleBlock& rb, const LdsTables& t, int W, int cell,
                                           const Things<K>& p) {
  int layer = t.top_layer[cell];
  int z = t.top_z[cell];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const bool here = (sel<K>(p.cell, k) == cell) && (rb.dyn_z[k] > z);
    layer = here ? rb.dyn_layer[k] : layer;
    z = here ? rb.dyn_z[k] : z;
  }
  return layer;
}

// Re-derive one cell of this environment's slice of the LDS images.
template <int K, bool kBoard>
__device__ __forceinline__ void repaint_cell(const RuleBlock& rb, const LdsTables& t,
                                             const uint8_t* layer_char, int8_t* my_obs,
                                             int8_t* my_board, int HW, int W, int cell,
                                             const Things<K>& p) {
  my_obs[t.top_layer[cell] * HW + cell] = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) my_obs[rb.dyn_layer[k] * HW + cell] = 0;
  const int layer = shown_layer<K>(rb, t, W, cell, p);
  my_obs[layer * HW + cell] = 1;
  if (kBoard) my_board[cell] = (int8_t)layer_char[layer];
}

// Stream `nbytes` of an LDS image to global memory.  16-byte vector path when the
// destination is 16-byte aligned, byte path otherwise (odd batch tails only).
// 16-byte store of four floats from the update pass (reward, discount).  They are
// not read again by this launch: CAMPX_STEP_STREAM=1 sends them with the streaming
// policy of the observation stores.
#ifndef CAMPX_STEP_STREAM
#define CAMPX_STEP_STREAM 0
#endif
__device__ __forceinline__ void store_f4(float* p, const float (&v)[4]) {
  const u32x4 bits = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]),
                      __float_as_uint(v[3])};
#if CAMPX_STEP_STREAM
  store16_streaming(reinterpret_cast<u32x4*>(p), bits);
#else
  *reinterpret_cast<u32x4*>(p) = bits;
#endif
}

// Fill a wave's LDS image (n_rows rows of R bytes, back to back, 16-byte aligned)
// with the scenery: 16 bytes per lane per step from the rotated scenery table
// (see render_kernel) when the spec carries it, else byte by byte from `row`.
__device__ __forceinline__ void fill_image(int8_t* img, int n_rows, int R, const int8_t* rot,
                                           bool have_rot, const int8_t* row_lds, int lane) {
  const int total = n_rows * R;
  if (have_rot) {
    // Four 16-byte loads in flight per lane before the first LDS write (one load per
    // iteration made a launch of the one-frame kernels wait ~11 L2 round trips in a row);
    // the offset inside the row advances incrementally instead of by a modulo per chunk.
    const int pitch = ((R + 15) & ~15) + 16;
    const int step = (kWave * 16) % R;
    int k = (lane * 16) % R;
    for (int off = lane * 16; off < total; off += 4 * kWave * 16) {
      u32x4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = *reinterpret_cast<const u32x4*>(rot + (k & 15) * pitch + (k & ~15));
        k += step;
        k = k >= R ? k - R : k;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int o = off + j * kWave * 16;
        if (o + 16 <= total) {
          *reinterpret_cast<u32x4*>(img + o) = v[j];
        } else if (o < total) {
          const uint32_t w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
          for (int b = 0; o + b < total; ++b) img[o + b] = (int8_t)(w[b >> 2] >> ((b & 3) * 8));
        }
      }
    }
  } else {
    for (int off = lane; off < total; off += kWave) img[off] = row_lds[off % R];
  }
}

// The one-frame kernels' version of fill_image: kN 16-byte loads per lane issued back to
// back (no branch between them: every offset of the cyclically continued row is a valid
// address), landed in LDS later, so that a launch waits for ONE round trip per kN KiB of
// image and the table lookup can be issued while they are in flight.  `k` is the offset
// inside the row of this lane's next chunk and advances as the chunks are issued.
template <int kN>
__device__ __forceinline__ void fill_issue(u32x4 (&v)[kN], const int8_t* rot, int R, int& k) {
  const int pitch = ((R + 15) & ~15) + 16;
  const int step = (kWave * 16) % R;
#pragma unroll
  for (int j = 0; j < kN; ++j) {
    v[j] = *reinterpret_cast<const u32x4*>(rot + (k & 15) * pitch + (k & ~15));
    k += step;
    k = k >= R ? k - R : k;
  }
}

template <int kN>
__device__ __forceinline__ void fill_land(const u32x4 (&v)[kN], int8_t* img, int total, int off0,
                                          int lane) {
#pragma unroll
  for (int j = 0; j < kN; ++j) {
    const int o = off0 + (j * kWave + lane) * 16;   // total = 64 rows: a multiple of 16
    if (o < total) *reinterpret_cast<u32x4*>(img + o) = v[j];
  }
}

// What is left of an image after the first kN chunks per lane.
template <int kN>
__device__ __forceinline__ void fill_rest(int8_t* img, int total, int R, const int8_t* rot, int& k,
                                          int lane) {
  for (int off0 = kN * kWave * 16; off0 < total; off0 += kN * kWave * 16) {
    u32x4 v[kN];
    fill_issue<kN>(v, rot, R, k);
    fill_land<kN>(v, img, total, off0, lane);
  }
}

constexpr int kStepObsLoads = 12;   // 12 KiB of a wave's observation image per round trip
constexpr int kStepBoardLoads = 4;

// Copy this lane's next actions (frames t .. t+kChunk-1) into LDS.  All loads of a
// group of 16 are issued before any is used; rows past the end are clamped so that
// there is no branch between the loads (a branch makes hipcc wait for each load
// before issuing the next: 64 serial HBM round trips per chunk).
template <int kLanes>
__device__ __forceinline__ int stage_actions(int8_t* staged, const int8_t* __restrict__ actions,
                                             int64_t B, int32_t T, int t, int64_t env, bool live,
                                             int lane) {
  const int64_t col = live ? env : 0;  // any valid column
  int bad = 0;  // ids outside 0..4 among this lane's real (unclamped) rows
  if (T - t >= 16) {
    const int n = (T - t < kChunk) ? T - t : kChunk;
    for (int r0 = 0; r0 < n; r0 += 16) {
      int8_t v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        int row = t + r0 + j;
        row = row < T ? row : T - 1;
        v[j] = actions[(int64_t)row * B + col];
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        staged[(r0 + j) * kLanes + lane] = live ? v[j] : (int8_t)4;
        bad += (live && t + r0 + j < T && (unsigned)v[j] > 4u) ? 1 : 0;
      }
    }
  } else {
    for (int r = 0; r < T - t; ++r) {
      const int8_t v = live ? actions[(int64_t)(t + r) * B + col] : (int8_t)4;
      staged[r * kLanes + lane] = v;
      bad += ((unsigned)v > 4u) ? 1 : 0;
    }
  }
  return bad;
}

// Ids outside 0..4 act as 4 (stay); the kernel that read them says so here (see
// CampxOutputs.bad_count / bad_flag) instead of a separate checking launch.
__device__ __forceinline__ void report_bad_actions(const CampxOutputs& out, int bad) {
  if (bad) {
    if (out.bad_count) atomicAdd(out.bad_count, bad);
    if (out.bad_flag)
      __hip_atomic_store(out.bad_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

template <bool kNT>
__device__ __forceinline__ void stream_out(const int8_t* lds, int8_t* dst, int nbytes, int lane) {
  if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
    const int nvec = nbytes >> 4;
    const u32x4* src = reinterpret_cast<const u32x4*>(lds);
    u32x4* out = reinterpret_cast<u32x4*>(dst);
#pragma unroll 4
    for (int i = lane; i < nvec; i += kWave) {
      if (kNT)
        store16_streaming(&out[i], src[i]);
      else
        out[i] = src[i];
    }
    for (int i = (nvec << 4) + lane; i < nbytes; i += kWave) dst[i] = lds[i];
  } else {
    for (int i = lane; i < nbytes; i += kWave) dst[i] = lds[i];
  }
}

// The same for 16-bit observations (out_format CAMPX_OBS_F16 / _BF16): a lane turns 8 image
// bytes (0 / 1) into 8 halves (0.0 / 1.0) and stores 16 bytes; `dst` counts elements.
__device__ __forceinline__ void stream_out16(const int8_t* lds, int8_t* dst, int nbytes, int lane,
                                             uint32_t one) {
  uint16_t* out = reinterpret_cast<uint16_t*>(dst);
  const int nvec = nbytes >> 3;
  for (int i = lane; i < nvec; i += kWave) {
    const uint2 b = *reinterpret_cast<const uint2*>(lds + 8 * i);
    u32x4 v;
    v.x = ((b.x & 0xffu) | ((b.x << 8) & 0x00ff0000u)) * one;
    v.y = (((b.x >> 16) & 0xffu) | ((b.x >> 8) & 0x00ff0000u)) * one;
    v.z = ((b.y & 0xffu) | ((b.y << 8) & 0x00ff0000u)) * one;
    v.w = (((b.y >> 16) & 0xffu) | ((b.y >> 8) & 0x00ff0000u)) * one;
    *reinterpret_cast<u32x4*>(out + 8 * i) = v;
  }
  for (int i = (nvec << 3) + lane; i < nbytes; i += kWave) out[i] = lds[i] ? (uint16_t)one : (uint16_t)0;
}

// Observation of the one-frame kernels, in the format the caller asked for.
__device__ __forceinline__ void step_stream_obs(const int8_t* img, const CampxOutputs& out,
                                                int64_t first_elem, int nbytes, int lane) {
  if (out.obs_format == CAMPX_OBS_INT8)
    stream_out<false>(img, out.obs + first_elem, nbytes, lane);
  else
    stream_out16(img, out.obs + 2 * first_elem, nbytes, lane,
                 out.obs_format == CAMPX_OBS_F16 ? 0x3C00u : 0x3F80u);
}

template <int K, bool kBoard, bool kNT, int kEnvs, bool kTrace>
__global__ __launch_bounds__(kWave) void rollout_kernel(RuleBlock rb,
                                                        const CampxSpec* __restrict__ spec,
                                                        CampxState st,
                                                        const int8_t* __restrict__ actions,
                                                        CampxOutputs out, int64_t B, int32_t T,
                                                        int32_t reset_first, int32_t emit_first,
                                                        int32_t xcd_mode, int64_t trace_plane) {
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];
  const int lane = threadIdx.x;
  const int H = rb.rows, W = rb.cols, HW = H * W, L = rb.n_layers, LHW = L * HW;
  // A wave owns kEnvs consecutive environments (lanes >= kEnvs only help stream).
  const int64_t env0 = (int64_t)tile_of_block(blockIdx.x, gridDim.x, xcd_mode) * kEnvs;
  const int64_t env = env0 + lane;
  const bool mine = lane < kEnvs;
  const bool live = mine && env < B;
  const int n_live = (B - env0 < kEnvs) ? (int)(B - env0) : kEnvs;

  // ---- LDS carve-up (every offset a multiple of 16)
  const int obs_bytes = kTrace ? 0 : ((kEnvs * LHW + 15) & ~15);
  const int board_bytes = (kBoard && !kTrace) ? ((kEnvs * HW + 15) & ~15) : 0;
  int8_t* obs_img = lds;
  int8_t* board_img = lds + obs_bytes;
  int8_t* tmpl = lds + obs_bytes + board_bytes;
  const int tmpl_bytes = (LHW + 15) & ~15;
  uint8_t* top_layer = reinterpret_cast<uint8_t*>(tmpl + tmpl_bytes);
  uint8_t* top_z = top_layer + CAMPX_MAX_CELLS;
  uint16_t* cover = reinterpret_cast<uint16_t*>(top_z + CAMPX_MAX_CELLS);
  uint8_t* layer_char = reinterpret_cast<uint8_t*>(cover + CAMPX_MAX_CELLS);
  int8_t* staged = reinterpret_cast<int8_t*>(layer_char + CAMPX_MAX_LAYERS);  // [kChunk][64]
  uint8_t* cell_class = reinterpret_cast<uint8_t*>(staged + kChunk * kWave);

  for (int i = lane; i < LHW; i += kWave) tmpl[i] = spec->obs_template[i];
  for (int i = lane; i < HW; i += kWave) {
    top_layer[i] = spec->static_top_layer[i];
    top_z[i] = spec->static_top_z[i];
    cover[i] = spec->static_cover[i];
    cell_class[i] = spec->cell_class[i];
  }
  if (lane < CAMPX_MAX_LAYERS) layer_char[lane] = spec->layer_char[lane];
  __syncthreads();
  const LdsTables tab = {top_layer, top_z, cover};

  // ---- dynamic state -> registers
  Things<K> pos = {0u, 0u, 0u};
  int over = 0;
  float ret = 0.0f;
#pragma unroll
  for (int k = 0; k < K; ++k) set_pos<K>(pos, k, rb.dyn_row0[k], rb.dyn_col0[k], W);
  if (!reset_first && live) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      set_pos<K>(pos, k, st.pos[(int64_t)(2 * k) * B + env], st.pos[(int64_t)(2 * k + 1) * B + env], W);
    }
    over = st.done[env];
    if (st.ret) ret = st.ret[env];
  }

  // ---- this wave's slice of the observation, as an LDS image
  int8_t* my_obs = obs_img + lane * LHW;
  int8_t* my_board = board_img + lane * HW;
  if (!kTrace) {
    const bool have_rot = spec->render_valid != 0;
    fill_image(obs_img, kEnvs, LHW, spec->rot_obs, have_rot, tmpl, lane);
    if (kBoard) {
      if (have_rot)
        fill_image(board_img, kEnvs, HW, spec->rot_board, true, nullptr, lane);
      else if (mine)
        for (int i = 0; i < HW; ++i) my_board[i] = (int8_t)layer_char[top_layer[i]];
    }
    __syncthreads();
  }
  if (!kTrace && mine) {
#pragma unroll
    for (int k = 0; k < K; ++k)
      repaint_cell<K, kBoard>(rb, tab, layer_char, my_obs, my_board, HW, W,
                              sel<K>(pos.cell, k), pos);
  }
  Things<K> img = pos;  // positions the image currently shows
  int bad = 0;

  if (!kTrace && emit_first) {
    __syncthreads();
    stream_out<kNT>(obs_img, out.obs + env0 * LHW, n_live * LHW, lane);
    if (kBoard) stream_out<kNT>(board_img, out.board + env0 * HW, n_live * HW, lane);
  }

  for (int t = 0; t < T; ++t) {
    const int in_chunk = t & (kChunk - 1);
    if (in_chunk == 0 && mine)  // each lane stages its own environment's next actions
      bad += stage_actions<kEnvs>(staged, actions, B, T, t, env, live, lane);
    int a = mine ? staged[in_chunk * kEnvs + lane] : 4;
    a = ((unsigned)a > 4u) ? 4 : a;

    // A finished episode is rebuilt from the art before its next action
    // (examples/reinforce.py:122: make_game() per episode).
    if (over) {
#pragma unroll
      for (int k = 0; k < K; ++k) set_pos<K>(pos, k, rb.dyn_row0[k], rb.dyn_col0[k], W);
      over = 0;
      ret = 0.0f;
    }

    // ---- update pass (engine.py:195-208)
    Things<K> shown = pos;  // where things stood at the latest repaint
    const int perf_from =
        rb.perf_dyn >= 0 ? sel<K>(pos.cell, rb.perf_dyn) : 0;
    float reward = 0.0f;
    float discount = 1.0f;
    bool first = true;
    auto add_reward = [&](float r) {  // plot.py:208-211: r + total
      reward = first ? r : r + reward;
      first = false;
    };
    for (int i = 0; i < rb.n_rules; ++i) {
      const CampxRule& R = rb.rules[i];
      const int d = R.dyn;
      switch (R.op) {
        case CAMPX_OP_AGENT: {
          int r2, c2;
          moved(a, H, W, sel<K>(pos.r, d), sel<K>(pos.c, d), r2, c2);
          const int target = shown_layer<K>(rb, tab, W, r2 * W + c2, shown);
          const bool blocked = (R.block_layers >> target) & 1u;
          r2 = blocked ? sel<K>(shown.r, d) : r2;
          c2 = blocked ? sel<K>(shown.c, d) : c2;
          set_pos<K>(pos, d, r2, c2, W);
          if (R.has_reward) {
            float r = R.base;
            if (R.reward_layers) {
              const int under = shown_layer<K>(rb, tab, W, r2 * W + c2, shown);
              r += (float)((R.reward_layers >> under) & 1u);
            }
            add_reward(r);
          }
          break;
        }
        case CAMPX_OP_DIR_HOVER: {
          const int cell = sel<K>(pos.cell, d);
          const int under = shown_layer<K>(rb, tab, W, cell, shown);
          const float gate = (under == R.aux) ? 1.0f : 0.0f;
          add_reward(R.base + gate * R.bonus[a]);
          break;
        }
        case CAMPX_OP_BOX: {
          int ar, ac, br, bc;
          moved(a, H, W, sel<K>(shown.r, R.aux), sel<K>(shown.c, R.aux), ar, ac);
          const int box_r = sel<K>(pos.r, d), box_c = sel<K>(pos.c, d);
          moved(a, H, W, box_r, box_c, br, bc);
          const int beyond = shown_layer<K>(rb, tab, W, br * W + bc, shown);
          const bool go = (ar == box_r) && (ac == box_c) && !((R.block_layers >> beyond) & 1u);
          set_pos<K>(pos, d, go ? br : box_r, go ? bc : box_c, W);
          break;
        }
        case CAMPX_OP_GOAL: {
          const int cell = sel<K>(pos.cell, d);
          const int arrived = (cover[cell] >> R.aux) & 1;
          add_reward(R.base + (float)arrived * R.bonus[0]);
          if (arrived) {  // plot.py:183-184
            over = 1;
            discount = 0.0f;
          }
          break;
        }
        default:
          break;
      }
      if (R.end_group) shown = pos;
    }
    if (!rb.any_reward) reward = __builtin_nanf("");
    ret += reward;
    if (out.perf && rb.perf_dyn >= 0 && live) {
      const int perf_to = sel<K>(pos.cell, rb.perf_dyn);
      out.perf[(int64_t)t * B + env] =
          (int8_t)class_progress(cell_class[perf_from], cell_class[perf_to], rb.perf_n);
    }

    if (kTrace) {
      // ---- split path: record where things are (and whether they show); the
      // render kernel turns that into observations.
      if (live) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int cell = sel<K>(pos.cell, k);
          const uint32_t vis = shown_layer<K>(rb, tab, W, cell, pos) == rb.dyn_layer[k];
          out.trace[(int64_t)k * trace_plane + (int64_t)t * B + env] = pack_trace(cell, vis);
        }
      }
    } else {
      // ---- render: fix up the cells things left and entered, then stream out
      __syncthreads();  // previous frame's reads of the image are done
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int was = sel<K>(img.cell, k);
        const int now = sel<K>(pos.cell, k);
        if (mine && was != now) {
          repaint_cell<K, kBoard>(rb, tab, layer_char, my_obs, my_board, HW, W, was, pos);
          repaint_cell<K, kBoard>(rb, tab, layer_char, my_obs, my_board, HW, W, now, pos);
        }
      }
      img = pos;
      __syncthreads();
      stream_out<kNT>(obs_img, out.obs + (int64_t)t * out.obs_t_stride + env0 * LHW,
                      n_live * LHW, lane);
      if (kBoard)
        stream_out<kNT>(board_img, out.board + (int64_t)t * out.board_t_stride + env0 * HW,
                        n_live * HW, lane);
    }

    if (live) {
      const int64_t at = (int64_t)t * B + env;
      if (out.reward) out.reward[at] = reward;
      if (out.discount) out.discount[at] = discount;
      if (out.done) out.done[at] = (uint8_t)over;
    }
  }

  if (live) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      st.pos[(int64_t)(2 * k) * B + env] = (int8_t)sel<K>(pos.r, k);
      st.pos[(int64_t)(2 * k + 1) * B + env] = (int8_t)sel<K>(pos.c, k);
    }
    st.done[env] = (uint8_t)over;
    if (st.ret) st.ret[env] = ret;
  }
  report_bad_actions(out, bad);
}


// ---------------------------------------------------------------------------
// One-mover games (K == 1) after campx_spec_compile(): the update pass of a frame
// is one lookup in the (cell, action) transition table, which was produced by the
// interpreter kernel above.  Rendering and streaming are unchanged.
struct MoverParams {
  int32_t rows, cols, n_layers, dyn_layer, dyn_z, row0, col0;
};

template <bool kBoard, bool kNT, int kEnvs>
__global__ __launch_bounds__(kWave) void rollout_table_kernel(
    MoverParams mp, const CampxSpec* __restrict__ spec, CampxState st,
    const int8_t* __restrict__ actions, CampxOutputs out, int64_t B, int32_t T,
    int32_t reset_first, int32_t emit_first, int32_t xcd_mode) {
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];
  const int lane = threadIdx.x;
  const int W = mp.cols, HW = mp.rows * mp.cols, LHW = mp.n_layers * HW;
  const int64_t env0 = (int64_t)tile_of_block(blockIdx.x, gridDim.x, xcd_mode) * kEnvs;
  const int64_t env = env0 + lane;
  const bool mine = lane < kEnvs;
  const bool live = mine && env < B;
  const int n_live = (B - env0 < kEnvs) ? (int)(B - env0) : kEnvs;

  // ---- LDS carve-up (every offset a multiple of 16)
  const int obs_bytes = (kEnvs * LHW + 15) & ~15;
  const int board_bytes = kBoard ? ((kEnvs * HW + 15) & ~15) : 0;
  int8_t* obs_img = lds;
  int8_t* board_img = lds + obs_bytes;
  uint2* table = reinterpret_cast<uint2*>(lds + obs_bytes + board_bytes);  // [HW*5] {reward, next|done<<8}
  uint16_t* paint = reinterpret_cast<uint16_t*>(table + CAMPX_MAX_CELLS * CAMPX_N_ACTIONS);
  uint8_t* scenery_char = reinterpret_cast<uint8_t*>(paint + CAMPX_MAX_CELLS);
  int8_t* staged = reinterpret_cast<int8_t*>(scenery_char + CAMPX_MAX_CELLS);  // [kChunk][kEnvs]
  int8_t* tmpl = staged + kChunk * kWave;

  for (int i = lane; i < HW * CAMPX_N_ACTIONS; i += kWave) {
    const CampxTransition tr = spec->table[i];
    table[i] = make_uint2(__float_as_uint(tr.reward),
                          (uint32_t)tr.next_cell | ((uint32_t)tr.done << 8) |
                              ((uint32_t)(tr.perf + 1) << 16));
  }
  for (int i = lane; i < HW; i += kWave) {
    // paint[cell]: byte offset (inside one environment's slice) of the scenery's own
    // 1 at that cell; bit 15 set when the scenery there hides the mover.
    const int layer = spec->static_top_layer[i];
    const bool hidden = spec->static_top_z[i] > mp.dyn_z;
    paint[i] = (uint16_t)((layer * HW + i) | (hidden ? 0x8000 : 0));
    scenery_char[i] = spec->layer_char[layer];
  }
  for (int i = lane; i < LHW; i += kWave) tmpl[i] = spec->obs_template[i];
  const int8_t mover_char = (int8_t)spec->layer_char[mp.dyn_layer];
  __syncthreads();

  int cell = mp.row0 * W + mp.col0;
  const int cell0 = cell;
  int over = 0;
  float ret = 0.0f;
  if (!reset_first && live) {
    cell = (int)st.pos[env] * W + (int)st.pos[B + env];
    over = st.done[env];
    if (st.ret) ret = st.ret[env];
  }

  int8_t* my_obs = obs_img + lane * LHW;
  int8_t* my_board = board_img + lane * HW;
  const int mover_off = mp.dyn_layer * HW;
  {
    const bool have_rot = spec->render_valid != 0;
    fill_image(obs_img, kEnvs, LHW, spec->rot_obs, have_rot, tmpl, lane);
    if (kBoard) {
      if (have_rot)
        fill_image(board_img, kEnvs, HW, spec->rot_board, true, nullptr, lane);
      else if (mine)
        for (int i = 0; i < HW; ++i) my_board[i] = (int8_t)scenery_char[i];
    }
    __syncthreads();
  }
  if (mine) {
    const int p = paint[cell];
    if (!(p & 0x8000)) {
      my_obs[p] = 0;
      my_obs[mover_off + cell] = 1;
      if (kBoard) my_board[cell] = mover_char;
    }
  }
  int shown_at = cell;  // where the image shows the mover
  int bad = 0;

  if (emit_first) {
    __syncthreads();
    stream_out<kNT>(obs_img, out.obs + env0 * LHW, n_live * LHW, lane);
    if (kBoard) stream_out<kNT>(board_img, out.board + env0 * HW, n_live * HW, lane);
  }

  for (int t = 0; t < T; ++t) {
    const int in_chunk = t & (kChunk - 1);
    if (in_chunk == 0 && mine)
      bad += stage_actions<kEnvs>(staged, actions, B, T, t, env, live, lane);
    int a = mine ? staged[in_chunk * kEnvs + lane] : 4;
    a = ((unsigned)a > 4u) ? 4 : a;
    if (over) {  // rebuilt from the art before its next action
      cell = cell0;
      ret = 0.0f;
    }
    const uint2 tr = table[cell * CAMPX_N_ACTIONS + a];
    const float reward = __uint_as_float(tr.x);
    cell = (int)(tr.y & 0xffu);
    over = (int)((tr.y >> 8) & 1u);
    ret += reward;

    {
      __syncthreads();  // previous frame's reads of the image are done
      if (mine && cell != shown_at) {
        const int was = paint[shown_at], now = paint[cell];
        if (!(was & 0x8000)) {
          my_obs[mover_off + shown_at] = 0;
          my_obs[was] = 1;
          if (kBoard) my_board[shown_at] = (int8_t)scenery_char[shown_at];
        }
        if (!(now & 0x8000)) {
          my_obs[now] = 0;
          my_obs[mover_off + cell] = 1;
          if (kBoard) my_board[cell] = mover_char;
        }
        shown_at = cell;
      }
      __syncthreads();
      stream_out<kNT>(obs_img, out.obs + (int64_t)t * out.obs_t_stride + env0 * LHW,
                      n_live * LHW, lane);
      if (kBoard)
        stream_out<kNT>(board_img, out.board + (int64_t)t * out.board_t_stride + env0 * HW,
                        n_live * HW, lane);
    }
    if (live) {
      const int64_t at = (int64_t)t * B + env;
      if (out.reward) out.reward[at] = reward;
      if (out.discount) out.discount[at] = over ? 0.0f : 1.0f;
      if (out.done) out.done[at] = (uint8_t)over;
      if (out.perf) out.perf[at] = (int8_t)((int)((tr.y >> 16) & 3u) - 1);
    }
  }

  if (live) {
    st.pos[env] = (int8_t)(cell / W);
    st.pos[B + env] = (int8_t)(cell % W);
    st.done[env] = (uint8_t)over;
    if (st.ret) st.ret[env] = ret;
  }
  report_bad_actions(out, bad);
}

// ---------------------------------------------------------------------------
// One frame of a one-mover game: Engine.play().  There is no chain to follow, so this
// is a one-shot kernel with two memory round trips: {state, action, scenery image} ->
// table entry -> patch the image in LDS -> stream out.  The table entry carries what
// painting the mover at its new cell needs (`paint`: scenery layer there, hidden flag),
// so nothing else depends on it.  One wave = one workgroup = 64 environments.
template <bool kBoard>
__global__ __launch_bounds__(kWave) void step_table_kernel(
    MoverParams mp, const CampxSpec* __restrict__ spec, CampxState st,
    const int8_t* __restrict__ actions, CampxOutputs out, int64_t B, int32_t reset_first) {
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];
  const int lane = threadIdx.x;
  const int W = mp.cols, HW = mp.rows * mp.cols, LHW = mp.n_layers * HW;
  const int64_t env0 = (int64_t)blockIdx.x * kWave;
  const int64_t env = env0 + lane;
  const bool live = env < B;
  const int n_live = (B - env0 < kWave) ? (int)(B - env0) : kWave;
  int8_t* obs_img = lds;
  int8_t* board_img = lds + ((kWave * LHW + 15) & ~15);

  int r = mp.row0, c = mp.col0, over = 0, a = 4;
  float ret = 0.0f;
  if (live) {
    a = actions[env];
    if (!reset_first) {
      r = st.pos[env];
      c = st.pos[B + env];
      over = st.done[env];
      if (st.ret) ret = st.ret[env];
    }
  }
  // ONE round trip: state and action (above), the whole transition table (5 KiB at most,
  // copied to LDS so that the lookup which depends on the state is an LDS read rather than
  // a second trip) and the scenery image, all in flight at once.
  __shared__ __attribute__((aligned(16))) CampxTransition lds_table[CAMPX_MAX_CELLS * CAMPX_N_ACTIONS];
  constexpr int kTableLoads = (int)(sizeof(lds_table) / (16 * kWave));
  static_assert(sizeof(lds_table) == (size_t)kTableLoads * 16 * kWave, "whole 16-byte chunks per lane");
  u32x4 v_table[kTableLoads];
#pragma unroll
  for (int j = 0; j < kTableLoads; ++j)
    v_table[j] = reinterpret_cast<const u32x4*>(spec->table)[j * kWave + lane];
  u32x4 v_obs[kStepObsLoads], v_board[kStepBoardLoads];
  int k_obs = (lane * 16) % LHW, k_board = (lane * 16) % HW;
  fill_issue<kStepObsLoads>(v_obs, spec->rot_obs, LHW, k_obs);
  if (kBoard) fill_issue<kStepBoardLoads>(v_board, spec->rot_board, HW, k_board);

  const int bad = (live && (unsigned)a > 4u) ? 1 : 0;
  a = ((unsigned)a > 4u) ? 4 : a;
  // a finished episode is rebuilt from the art before its next action
  int cell = over ? mp.row0 * W + mp.col0 : r * W + c;
  ret = over ? 0.0f : ret;
#pragma unroll
  for (int j = 0; j < kTableLoads; ++j)
    reinterpret_cast<u32x4*>(lds_table)[j * kWave + lane] = v_table[j];
  // (one wave: LDS operations complete in order, no barrier needed)
  const CampxTransition tr = lds_table[cell * CAMPX_N_ACTIONS + a];
  fill_land<kStepObsLoads>(v_obs, obs_img, kWave * LHW, 0, lane);
  fill_rest<kStepObsLoads>(obs_img, kWave * LHW, LHW, spec->rot_obs, k_obs, lane);
  if (kBoard) {
    fill_land<kStepBoardLoads>(v_board, board_img, kWave * HW, 0, lane);
    fill_rest<kStepBoardLoads>(board_img, kWave * HW, HW, spec->rot_board, k_board, lane);
  }
  cell = tr.next_cell;
  ret += tr.reward;
  if (!(tr.paint & 0x80u)) {   // the mover shows at its cell
    int8_t* my_obs = obs_img + lane * LHW;
    my_obs[(int)(tr.paint & 0x7fu) * HW + cell] = 0;
    my_obs[mp.dyn_layer * HW + cell] = 1;
    if (kBoard) board_img[lane * HW + cell] = (int8_t)spec->layer_char[mp.dyn_layer];
  }
  if (live) {
    if (out.reward) out.reward[env] = tr.reward;
    if (out.discount) out.discount[env] = tr.done ? 0.0f : 1.0f;
    if (out.done) out.done[env] = tr.done;
    if (out.perf) out.perf[env] = tr.perf;
    st.pos[env] = (int8_t)(cell / W);
    st.pos[B + env] = (int8_t)(cell % W);
    st.done[env] = tr.done;
    if (st.ret) st.ret[env] = ret;
  }
  // one wave: LDS operations complete in order, no barrier needed
  step_stream_obs(obs_img, out, env0 * LHW, n_live * LHW, lane);
  if (kBoard) stream_out<false>(board_img, out.board + env0 * HW, n_live * HW, lane);
  report_bad_actions(out, bad);
}

// The same for two-mover games with a pair table (campx_pair_table_build): one more
// dependent round trip for the reward list and the scenery layers the movers cover.
template <bool kBoard>
__global__ __launch_bounds__(kWave) void step_pair_kernel(
    MoverParams mp, int32_t layer1, int32_t row1, int32_t col1,
    const CampxSpec* __restrict__ spec, CampxState st, const int8_t* __restrict__ actions,
    CampxOutputs out, int64_t B, int32_t reset_first) {
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];
  const int lane = threadIdx.x;
  const int W = mp.cols, HW = mp.rows * mp.cols, LHW = mp.n_layers * HW;
  const int64_t env0 = (int64_t)blockIdx.x * kWave;
  const int64_t env = env0 + lane;
  const bool live = env < B;
  const int n_live = (B - env0 < kWave) ? (int)(B - env0) : kWave;
  int8_t* obs_img = lds;
  int8_t* board_img = lds + ((kWave * LHW + 15) & ~15);
  const float* g_rewards = static_cast<const float*>(st.pair_table);
  const uint32_t* g_entries = reinterpret_cast<const uint32_t*>(g_rewards + 256);

  const uint32_t init0 = (uint32_t)(mp.row0 * W + mp.col0), init1 = (uint32_t)(row1 * W + col1);
  uint32_t c0 = init0, c1 = init1;
  int over = 0, a = 4;
  float ret = 0.0f;
  if (live) {
    a = actions[env];
    if (!reset_first) {
      c0 = (uint32_t)((int)st.pos[env] * W + (int)st.pos[B + env]);
      c1 = (uint32_t)((int)st.pos[2 * B + env] * W + (int)st.pos[3 * B + env]);
      over = st.done[env];
      if (st.ret) ret = st.ret[env];
    }
  }
  u32x4 v_obs[kStepObsLoads], v_board[kStepBoardLoads];
  int k_obs = (lane * 16) % LHW, k_board = (lane * 16) % HW;
  fill_issue<kStepObsLoads>(v_obs, spec->rot_obs, LHW, k_obs);
  if (kBoard) fill_issue<kStepBoardLoads>(v_board, spec->rot_board, HW, k_board);

  const int bad = (live && (unsigned)a > 4u) ? 1 : 0;
  a = ((unsigned)a > 4u) ? 4 : a;
  if (over) {  // rebuilt from the art before its next action
    c0 = init0;
    c1 = init1;
    ret = 0.0f;
  }
  const uint32_t e = g_entries[pair_index(c0, c1, HW) + (uint32_t)a];
  fill_land<kStepObsLoads>(v_obs, obs_img, kWave * LHW, 0, lane);
  fill_rest<kStepObsLoads>(obs_img, kWave * LHW, LHW, spec->rot_obs, k_obs, lane);
  if (kBoard) {
    fill_land<kStepBoardLoads>(v_board, board_img, kWave * HW, 0, lane);
    fill_rest<kStepBoardLoads>(board_img, kWave * HW, HW, spec->rot_board, k_board, lane);
  }
  c0 = e & 0x7fu;
  c1 = (e >> 7) & 0x7fu;
  const float reward = g_rewards[(e >> 19) & 0xffu];
  const int done = (int)((e >> 16) & 1u);
  ret += reward;
  int8_t* my_obs = obs_img + lane * LHW;
  if ((e >> 14) & 1u) {
    my_obs[(int)spec->static_top_layer[c0] * HW + (int)c0] = 0;
    my_obs[mp.dyn_layer * HW + (int)c0] = 1;
    if (kBoard) board_img[lane * HW + (int)c0] = (int8_t)spec->layer_char[mp.dyn_layer];
  }
  if ((e >> 15) & 1u) {
    my_obs[(int)spec->static_top_layer[c1] * HW + (int)c1] = 0;
    my_obs[layer1 * HW + (int)c1] = 1;
    if (kBoard) board_img[lane * HW + (int)c1] = (int8_t)spec->layer_char[layer1];
  }
  if (live) {
    if (out.reward) out.reward[env] = reward;
    if (out.discount) out.discount[env] = done ? 0.0f : 1.0f;
    if (out.done) out.done[env] = (uint8_t)done;
    if (out.perf) out.perf[env] = (int8_t)((int)((e >> 17) & 3u) - 1);
    st.pos[env] = (int8_t)(c0 / (uint32_t)W);
    st.pos[B + env] = (int8_t)(c0 % (uint32_t)W);
    st.pos[2 * B + env] = (int8_t)(c1 / (uint32_t)W);
    st.pos[3 * B + env] = (int8_t)(c1 % (uint32_t)W);
    st.done[env] = (uint8_t)done;
    if (st.ret) st.ret[env] = ret;
  }
  step_stream_obs(obs_img, out, env0 * LHW, n_live * LHW, lane);
  if (kBoard) stream_out<false>(board_img, out.board + env0 * HW, n_live * HW, lane);
  report_bad_actions(out, bad);
}

size_t table_lds_bytes(const CampxSpec& s, bool board, int envs) {
  const int HW = s.rows * s.cols, LHW = s.n_layers * HW;
  size_t n = (size_t)((envs * LHW + 15) & ~15);
  if (board) n += (size_t)((envs * HW + 15) & ~15);
  n += sizeof(uint2) * CAMPX_MAX_CELLS * CAMPX_N_ACTIONS;
  n += CAMPX_MAX_CELLS * sizeof(uint16_t) + CAMPX_MAX_CELLS;
  n += (size_t)kChunk * kWave + (size_t)LHW;
  return (n + 15) & ~(size_t)15;
}

// ---------------------------------------------------------------------------
// Split path, first half: the update pass alone.  Out: the trace (one byte per moving
// thing per frame per environment: cell + "is the character its cell shows") and the
// per-frame scalars (reward, discount, done, perf).
//
// The only loop-carried dependency of a frame is state -> table[state, action] ->
// state: one LDS read.  A workgroup owns kEnvs = 64 * kProd consecutive environments
// and has three kinds of waves:
//   kProd producer waves   run that dependent chain, one environment per lane,
//                          a group of (16) frames at a time, into a double-buffered LDS ring;
//   kCons consumer waves   turn the previous group into the output streams while the
//                          producers run the next one: 16 bytes per lane per store
//                          (4 environments of a float stream, 16 of a byte stream),
//                          write-through, so that nothing is left dirty in L2 for the
//                          end-of-kernel flush and the render kernel behind it;
//   kProd / 2 loader waves bring the actions in, 16 bytes per lane per load, one
//                          64-frame chunk ahead.  They issue no stores, so waiting for
//                          their loads never waits for a store (vmcnt is in order).
// One s_barrier per group, in each role's own loop.  The accesses are 16 bytes whatever the
// batch size: dword-aligned for the float streams, byte-aligned for the byte streams and the
// actions when B is not a multiple of 16 (legal on this stack; tools/probes/unaligned_probe.hip),
// and only the batch's last, partial group of 16 environments goes byte by byte.

// Cache policy of the update kernels' output stores (A/B builds): 0 plain, 1 sc0 sc1
// (write-through), 2 sc0 sc1 nt.
#ifndef CAMPX_UPD_FLAVOR
#define CAMPX_UPD_FLAVOR 1
#endif
__device__ __forceinline__ void store16_update(void* p, u32x4 v) {
#if CAMPX_UPD_FLAVOR == 0
  *reinterpret_cast<u32x4*>(p) = v;
#elif CAMPX_UPD_FLAVOR == 1
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#else
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#endif
}

// Number of bytes >= 5 (as unsigned) among the 16 of v: the action ids outside 0..4.
__device__ __forceinline__ int count_bad16(u32x4 v) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  int n = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t hi = (((w[i] & 0x7f7f7f7fu) + 0x7b7b7b7bu) | w[i]) & 0x80808080u;
    n += __builtin_popcount(hi);
  }
  return n;
}

// Bytes > 4 (as unsigned) of v replaced by 4: an id outside 0..4 acts as "stay".
__device__ __forceinline__ uint32_t clamp_ids(uint32_t w) {
  const uint32_t hi = (((w & 0x7f7f7f7fu) + 0x7b7b7b7bu) | w) & 0x80808080u;  // bad bytes
  const uint32_t m = (hi >> 7) * 0xffu;
  return (w & ~m) | (0x04040404u & m);
}

// The staged actions of a chunk are packed two frames to a byte - frame 2p in the low
// nibble of row p, frame 2p + 1 in the high one, already clamped to 0..4 - which halves
// their LDS footprint (what decides how many update workgroups fit on a CU).
//
// The loader waves' state: one chunk (kChunk frames x kEnvs environments) of actions in
// flight in registers between issue() and land().
template <int kEnvs, int kLoaders>
struct ActionLoader {
  static constexpr int kVecPerRow = kEnvs / 16;
  static constexpr int kLanes = kWave * kLoaders;          // loader lanes of the workgroup
  static constexpr int kPairs = (kChunk / 2) * kVecPerRow / kLanes;   // row pairs per lane
  static_assert(kPairs >= 1 && (kChunk / 2) * kVecPerRow % kLanes == 0, "loader shape");
  u32x4 pend[2 * kPairs];

  // 16-byte loads (byte-aligned unless B % 16 == 0: unaligned 16-byte accesses are legal on
  // this stack, tools/probes/unaligned_probe.hip); rows past T and groups of 16 environments
  // that are not wholly below B are clamped to valid ones (no branch between the loads) and
  // redone or neutralised in land().
  // `lane` counts over all loader waves: 0 .. kLanes-1
  __device__ __forceinline__ void issue(const int8_t* __restrict__ actions, int64_t B, int32_t T,
                                        int t0, int64_t env0, int lane) {
#pragma unroll
    for (int i = 0; i < kPairs; ++i) {
      const int v = lane + i * kLanes;
      const int rp = v / kVecPerRow, q = v % kVecPerRow;
      int64_t e = env0 + 16 * q;
      e = e + 16 <= B ? e : 0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int row = t0 + 2 * rp + h;
        row = row < T ? row : T - 1;
        pend[2 * i + h] = *reinterpret_cast<const u32x4*>(actions + (int64_t)row * B + e);
      }
    }
  }

  __device__ __forceinline__ int land(int8_t* staged, const int8_t* __restrict__ actions, int64_t B,
                                      int32_t T, int t0, int64_t env0, int lane) {
    int bad = 0;
#pragma unroll
    for (int i = 0; i < kPairs; ++i) {
      const int v = lane + i * kLanes;
      const int rp = v / kVecPerRow, q = v % kVecPerRow;
      const int64_t e = env0 + 16 * q;
      const bool here = e + 16 <= B;
      const bool real0 = here && (t0 + 2 * rp < T), real1 = here && (t0 + 2 * rp + 1 < T);
      u32x4 lo = pend[2 * i], hi = pend[2 * i + 1];
      if (!here && e < B) {
        // the batch's last, partial group of environments (one lane of the last workgroup):
        // byte by byte, "stay" past the end
        uint32_t l[4] = {0x04040404u, 0x04040404u, 0x04040404u, 0x04040404u};
        uint32_t h[4] = {0x04040404u, 0x04040404u, 0x04040404u, 0x04040404u};
        for (int k = 0; e + k < B; ++k) {
          const int sh = 8 * (k & 3);
          if (t0 + 2 * rp < T) {
            const uint32_t a = (uint8_t)actions[(int64_t)(t0 + 2 * rp) * B + e + k];
            l[k >> 2] = (l[k >> 2] & ~(0xffu << sh)) | (a << sh);
          }
          if (t0 + 2 * rp + 1 < T) {
            const uint32_t a = (uint8_t)actions[(int64_t)(t0 + 2 * rp + 1) * B + e + k];
            h[k >> 2] = (h[k >> 2] & ~(0xffu << sh)) | (a << sh);
          }
        }
        uint32_t out[4];
        for (int k = 0; k < 4; ++k) out[k] = clamp_ids(l[k]) | (clamp_ids(h[k]) << 4);
        const u32x4 packed = {out[0], out[1], out[2], out[3]};
        *reinterpret_cast<u32x4*>(staged + rp * kEnvs + 16 * q) = packed;
        const u32x4 l4 = {l[0], l[1], l[2], l[3]}, h4 = {h[0], h[1], h[2], h[3]};
        bad += count_bad16(l4) + count_bad16(h4);
        continue;
      }
      const uint32_t l[4] = {lo.x, lo.y, lo.z, lo.w}, h[4] = {hi.x, hi.y, hi.z, hi.w};
      uint32_t out[4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        out[k] = (real0 ? clamp_ids(l[k]) : 0x04040404u) | ((real1 ? clamp_ids(h[k]) : 0x04040404u) << 4);
      const u32x4 packed = {out[0], out[1], out[2], out[3]};
      *reinterpret_cast<u32x4*>(staged + rp * kEnvs + 16 * q) = packed;
      bad += (real0 ? count_bad16(lo) : 0) + (real1 ? count_bad16(hi) : 0);
    }
    return bad;
  }
};

// The action of frame j of a group that starts at frame t0 of its chunk,
// for environment `le` of the workgroup.
template <int kEnvs>
__device__ __forceinline__ uint32_t staged_action(const int8_t* chunk, int t0_in_chunk, int j,
                                                   int le) {
  const uint32_t b = (uint8_t)chunk[((t0_in_chunk + j) >> 1) * kEnvs + le];
  return (b >> (4 * (j & 1))) & 0xfu;   // t0_in_chunk is a multiple of the group size (even)
}

// Four bytes (the low byte of each argument) as one dword.
__device__ __forceinline__ uint32_t pack4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  return (a & 0xffu) | ((b & 0xffu) << 8) | ((c & 0xffu) << 16) | (d << 24);
}

// A/B: give each XCD a contiguous eighth of the batch (see tile_of_block).
#ifndef CAMPX_UPD_XCD
#define CAMPX_UPD_XCD 0
#endif

// loader waves of an update workgroup: one per 128 environments (a 256-environment
// chunk held by one wave is 64 VGPRs of loads in flight: with two the pair kernel stops
// spilling).  CAMPX_UPD_LOADERS overrides for A/B builds.
#ifdef CAMPX_UPD_LOADERS
constexpr int update_loaders(int) { return CAMPX_UPD_LOADERS; }
#else
constexpr int update_loaders(int prod) { return prod >= 4 ? prod / 2 : 1; }
#endif

// A/B knobs: roll the consumers' float-stream loop (fewer registers, measured +0.8 us on
// the boat race), and a register cap in waves per SIMD (capping the pair kernel at 96
// VGPRs for two workgroups per CU measured 47 us against 42 us uncapped at 110).
#ifndef CAMPX_UPD_ROLL_A
#define CAMPX_UPD_ROLL_A 0
#endif
#ifndef CAMPX_UPD_MINWAVES
#define CAMPX_UPD_MINWAVES 1
#endif
constexpr int update_min_waves(int, int) { return CAMPX_UPD_MINWAVES; }

template <int kProd, int kCons, int kG>   // kG: frames per group (a ring slot)
__global__ __launch_bounds__((kProd + kCons + update_loaders(kProd)) * kWave,
                             update_min_waves(kProd, kCons)) void update_table_kernel(
    MoverParams mp, const CampxSpec* __restrict__ spec, CampxState st,
    const int8_t* __restrict__ actions, CampxOutputs out, int64_t B, int32_t T,
    int32_t reset_first) {
  constexpr int kLoad = update_loaders(kProd);
  constexpr int E = kProd * kWave, CL = kCons * kWave, kThreads = (kProd + kCons + kLoad) * kWave;
  // LDS entry: x = reward; y = [0:15] byte offset of the table row the NEXT frame starts
  // from (the art's cell when this frame ended the episode: the rebuild is folded into
  // the chain), [16:22] the cell after this frame, [23] whether the mover shows there,
  // [24] done, [25:26] perf + 1.  The ring keeps x and the upper half of y.
  __shared__ uint2 table[CAMPX_MAX_CELLS * CAMPX_N_ACTIONS];
  __shared__ __attribute__((aligned(16))) int8_t staged[2][(kChunk / 2) * E];
  __shared__ __attribute__((aligned(16))) float ring_r[2][kG][E];
  __shared__ __attribute__((aligned(16))) uint16_t ring_y[2][kG][E];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const bool producer = wave < kProd, loader = wave >= kProd + kCons;
  const int llane = (int)threadIdx.x - (kProd + kCons) * kWave;  // loaders: 0 .. 64*kLoad-1
  const int W = mp.cols, HW = mp.rows * mp.cols;
  const int64_t env0 = (int64_t)tile_of_block(blockIdx.x, gridDim.x, CAMPX_UPD_XCD) * E;

  const int cell0 = mp.row0 * W + mp.col0;
  constexpr int kRowBytes = CAMPX_N_ACTIONS * (int)sizeof(uint2);
  for (int i = threadIdx.x; i < HW * CAMPX_N_ACTIONS; i += kThreads) {
    const CampxTransition tr = spec->table[i];
    const uint32_t from = tr.done ? (uint32_t)cell0 : (uint32_t)tr.next_cell;
    const uint32_t vis = (tr.paint & 0x80u) ? 0u : 1u;  // scenery in front hides the mover
    table[i] = make_uint2(__float_as_uint(tr.reward),
                          (from * kRowBytes) | ((uint32_t)tr.next_cell << 16) | (vis << 23) |
                              ((uint32_t)tr.done << 24) | ((uint32_t)(tr.perf + 1) << 25));
  }
  ActionLoader<E, kLoad> ld;
  int bad = 0;
  if (loader && T > 0) {
    ld.issue(actions, B, T, 0, env0, llane);
    bad += ld.land(staged[0], actions, B, T, 0, env0, llane);
  }

  const int le = wave * kWave + lane;  // producers: this lane's environment in the workgroup
  const int64_t env = env0 + le;
  const bool live = producer && env < B;
  int cell = cell0, over = 0;
  float ret = 0.0f;
  if (live && !reset_first) {
    cell = (int)st.pos[env] * W + (int)st.pos[B + env];
    over = st.done[env];
    if (st.ret) ret = st.ret[env];
  }
  uint32_t row_off = (uint32_t)(over ? cell0 : cell) * kRowBytes;  // the chain's state
  const char* table_bytes = reinterpret_cast<const char*>(table);
  const int clane = (int)threadIdx.x - kProd * kWave;  // consumers: 0 .. CL-1
  constexpr int kGroupsPerChunk = kChunk / kG;
  __syncthreads();

  const int n_groups = (T + kG - 1) / kG;
  // Each kind of wave runs its own loop (one s_barrier per group in each, so the counts
  // agree): registers are then allocated per role, and the loads a loader keeps in flight
  // across iterations do not take registers from the other two.
  if (producer) {
    for (int g = 0; g <= n_groups; ++g) {
        if (g < n_groups) {
          const int t0 = g * kG;
          const int8_t* chunk = staged[(t0 / kChunk) & 1];
          const int n = (T - t0 < kG) ? T - t0 : kG;
          uint32_t col_off[kG];  // action * sizeof(entry), off the dependent chain
  #pragma unroll
          for (int j = 0; j < kG; ++j)
            col_off[j] = staged_action<E>(chunk, t0 & (kChunk - 1), j, le) * (uint32_t)sizeof(uint2);
  #pragma unroll
          for (int j = 0; j < kG; ++j) {
            if (j < n) {
              // the dependent chain: row offset -> entry -> row offset
              const uint2 e = *reinterpret_cast<const uint2*>(table_bytes + row_off + col_off[j]);
              row_off = e.y & 0xffffu;
              ring_r[g & 1][j][le] = __uint_as_float(e.x);
              ring_y[g & 1][j][le] = (uint16_t)(e.y >> 16);
              // off the chain: the return restarts after an episode end
              ret = (over ? 0.0f : ret) + __uint_as_float(e.x);
              over = (int)((e.y >> 24) & 1u);
              cell = (int)((e.y >> 16) & 0x7fu);
            }
          }
        }
      
      __syncthreads();
    }
  } else if (!loader) {
    for (int g = 0; g <= n_groups; ++g) {
        if (g > 0) {
          const int gp = g - 1, t0 = gp * kG, rb = gp & 1;
          const int n = (T - t0 < kG) ? T - t0 : kG;
          // ---- float streams: item = (frame j, 4 environments)
          constexpr int QA = E / 4, kItA = (kG * QA + CL - 1) / CL;
  #if CAMPX_UPD_ROLL_A
#pragma unroll 1
#else
#pragma unroll
#endif
          for (int it = 0; it < kItA; ++it) {
            const int item = clane + it * CL;
            const int j = item / QA, q = item % QA;
            const int64_t e0 = env0 + 4 * q;
            if (j < n && e0 < B) {
              const u32x4 r4 = *reinterpret_cast<const u32x4*>(&ring_r[rb][j][4 * q]);
              const uint2 y4 = *reinterpret_cast<const uint2*>(&ring_y[rb][j][4 * q]);
              const uint32_t dn[4] = {(y4.x >> 8) & 1u, (y4.x >> 24) & 1u, (y4.y >> 8) & 1u,
                                      (y4.y >> 24) & 1u};
              const int64_t at = (int64_t)(t0 + j) * B + e0;
              if (e0 + 4 <= B) {   // (dword-aligned; 16-byte aligned when B % 4 == 0)
                if (out.reward) store16_update(out.reward + at, r4);
                if (out.discount) {
                  const u32x4 d4 = {dn[0] ? 0u : 0x3f800000u, dn[1] ? 0u : 0x3f800000u,
                                    dn[2] ? 0u : 0x3f800000u, dn[3] ? 0u : 0x3f800000u};
                  store16_update(out.discount + at, d4);
                }
              } else {
                const uint32_t rw[4] = {r4.x, r4.y, r4.z, r4.w};
                for (int i = 0; i < 4 && e0 + i < B; ++i) {
                  if (out.reward) out.reward[at + i] = __uint_as_float(rw[i]);
                  if (out.discount) out.discount[at + i] = dn[i] ? 0.0f : 1.0f;
                }
              }
            }
          }
          // ---- byte streams: item = (frame j, 16 environments)
          constexpr int QB = E / 16, kItB = (kG * QB + CL - 1) / CL;
  #pragma unroll
          for (int it = 0; it < kItB; ++it) {
            const int item = clane + it * CL;
            const int j = item / QB, q = item % QB;
            const int64_t e0 = env0 + 16 * q;
            if (item < kG * QB && j < n && e0 < B) {
              const u32x4 ya = *reinterpret_cast<const u32x4*>(&ring_y[rb][j][16 * q]);
              const u32x4 yb = *reinterpret_cast<const u32x4*>(&ring_y[rb][j][16 * q + 8]);
              const uint32_t w[8] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w};
              uint32_t tr[4], dn[4], pf[4];
  #pragma unroll
              for (int k = 0; k < 4; ++k) {
                const uint32_t lo = w[2 * k], hi = w[2 * k + 1];  // two environments each
                tr[k] = pack4(lo, lo >> 16, hi, (hi >> 16) & 0xffu);
                dn[k] = pack4((lo >> 8) & 1u, (lo >> 24) & 1u, (hi >> 8) & 1u, (hi >> 24) & 1u);
                pf[k] = pack4(((lo >> 9) & 3u) - 1u, ((lo >> 25) & 3u) - 1u, ((hi >> 9) & 3u) - 1u,
                              (((hi >> 25) & 3u) - 1u) & 0xffu);
              }
              const int64_t at = (int64_t)(t0 + j) * B + e0;
              if (e0 + 16 <= B) {  // (byte-aligned unless B % 16 == 0: unaligned stores are legal here)
                const u32x4 t4 = {tr[0], tr[1], tr[2], tr[3]};
                store16_update(out.trace + at, t4);
                if (out.done) {
                  const u32x4 d4 = {dn[0], dn[1], dn[2], dn[3]};
                  store16_update(out.done + at, d4);
                }
                if (out.perf) {
                  const u32x4 p4 = {pf[0], pf[1], pf[2], pf[3]};
                  store16_update(out.perf + at, p4);
                }
              } else {
                for (int i = 0; i < 16 && e0 + i < B; ++i) {
                  const int sh = (i & 3) * 8;
                  out.trace[at + i] = (uint8_t)(tr[i >> 2] >> sh);
                  if (out.done) out.done[at + i] = (uint8_t)(dn[i >> 2] >> sh);
                  if (out.perf) out.perf[at + i] = (int8_t)(pf[i >> 2] >> sh);
                }
              }
            }
          }
        }
      
      __syncthreads();
    }
  } else {
    for (int g = 0; g <= n_groups; ++g) {
        // loader: while the producers are in chunk c, fetch chunk c + 1
        const int c = g / kGroupsPerChunk, phase = g % kGroupsPerChunk;
        const int t_next = (c + 1) * kChunk;
        if (t_next < T) {
          if (phase == 0) ld.issue(actions, B, T, t_next, env0, llane);
          if (phase == kGroupsPerChunk - 1)
            bad += ld.land(staged[(c + 1) & 1], actions, B, T, t_next, env0, llane);
        }
      
      __syncthreads();
    }
  }

  if (live) {
    st.pos[env] = (int8_t)(cell / W);
    st.pos[B + env] = (int8_t)(cell % W);
    st.done[env] = (uint8_t)over;
    if (st.ret) st.ret[env] = ret;
  }
  report_bad_actions(out, bad);
}

// ---------------------------------------------------------------------------
// Two-mover games: the same layout over the (cell, cell, action) pair table
// (campx_pair_table_build).  The dependent chain goes through the 32-bit entries
// themselves, which carry everything a frame outputs and go to the ring as they are; they
// sit in LDS (dynamic, kLdsEntries) when the table fits, else they are read through L1/L2.
// (Chaining through a separate 16-bit next-index table in LDS with the entries fetched off
// the chain from global memory was built and measured slower - 68.9 against 51.4 us - and
// removed.)
struct PairParams {
  int32_t rows, cols, n_layers;
  int32_t dyn_layer[2], row0[2], col0[2];
  int32_t lds_table;  // entries fit in LDS
};

#ifndef CAMPX_PAIR_LDS_ENTRIES
#define CAMPX_PAIR_LDS_ENTRIES 8192
#endif
constexpr int kPairLdsEntries = CAMPX_PAIR_LDS_ENTRIES;  // 32 KiB of LDS for the entries at most

#ifndef CAMPX_PAIR_GROUP
#define CAMPX_PAIR_GROUP 16   // frames per ring slot group (A/B builds)
#endif

template <bool kLdsEntries, int kProd, int kCons>
__global__ __launch_bounds__((kProd + kCons + update_loaders(kProd)) * kWave,
                             update_min_waves(kProd, kCons)) void update_pair_kernel(
    PairParams pp, const CampxSpec* __restrict__ spec, CampxState st,
    const int8_t* __restrict__ actions, CampxOutputs out, int64_t B, int32_t T,
    int32_t reset_first, int64_t trace_plane) {
  constexpr int kLoad = update_loaders(kProd), kG = CAMPX_PAIR_GROUP;
  constexpr int E = kProd * kWave, CL = kCons * kWave, kThreads = (kProd + kCons + kLoad) * kWave;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_entries[];  // kLdsEntries: n_entries
  __shared__ float reward_list[256];
  __shared__ __attribute__((aligned(16))) int8_t staged[2][(kChunk / 2) * E];
  __shared__ __attribute__((aligned(16))) uint32_t ring[2][kG][E];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const bool producer = wave < kProd, loader = wave >= kProd + kCons;
  const int llane = (int)threadIdx.x - (kProd + kCons) * kWave;  // loaders: 0 .. 64*kLoad-1
  const int W = pp.cols, HW = pp.rows * pp.cols;
  const int64_t env0 = (int64_t)tile_of_block(blockIdx.x, gridDim.x, CAMPX_UPD_XCD) * E;

  const float* g_rewards = static_cast<const float*>(st.pair_table);
  const uint32_t* g_entries = reinterpret_cast<const uint32_t*>(g_rewards + 256);
  const int n_entries = HW * HW * CAMPX_N_ACTIONS;
  if (kLdsEntries)
    for (int i = threadIdx.x; i < n_entries; i += kThreads) lds_entries[i] = g_entries[i];
  for (int i = threadIdx.x; i < 256; i += kThreads) reward_list[i] = g_rewards[i];
  ActionLoader<E, kLoad> ld;
  int bad = 0;
  if (loader && T > 0) {
    ld.issue(actions, B, T, 0, env0, llane);
    bad += ld.land(staged[0], actions, B, T, 0, env0, llane);
  }

  const int le = wave * kWave + lane;
  const int64_t env = env0 + le;
  const bool live = producer && env < B;
  const uint32_t init0 = (uint32_t)(pp.row0[0] * W + pp.col0[0]);
  const uint32_t init1 = (uint32_t)(pp.row0[1] * W + pp.col0[1]);
  uint32_t c0 = init0, c1 = init1;
  int over = 0;
  float ret = 0.0f;
  if (live && !reset_first) {
    c0 = (uint32_t)((int)st.pos[env] * W + (int)st.pos[B + env]);
    c1 = (uint32_t)((int)st.pos[2 * B + env] * W + (int)st.pos[3 * B + env]);
    over = st.done[env];
    if (st.ret) ret = st.ret[env];
  }
  const int clane = (int)threadIdx.x - kProd * kWave;
  constexpr int kGroupsPerChunk = kChunk / kG;
  __syncthreads();

  const int n_groups = (T + kG - 1) / kG;
  // Each kind of wave runs its own loop (one s_barrier per group in each, so the counts
  // agree): registers are then allocated per role, and the loads a loader keeps in flight
  // across iterations do not take registers from the other two.
  if (producer) {
    for (int g = 0; g <= n_groups; ++g) {
        if (g < n_groups) {
          const int t0 = g * kG;
          const int8_t* chunk = staged[(t0 / kChunk) & 1];
          const int n = (T - t0 < kG) ? T - t0 : kG;
          uint32_t act[kG];
  #pragma unroll
          for (int j = 0; j < kG; ++j) act[j] = staged_action<E>(chunk, t0 & (kChunk - 1), j, le);
  #pragma unroll
          for (int j = 0; j < kG; ++j) {
            if (j < n) {
              if (over) {  // rebuilt from the art before its next action
                c0 = init0;
                c1 = init1;
              }
              const uint32_t idx = pair_index(c0, c1, HW) + act[j];
              const uint32_t e = kLdsEntries ? lds_entries[idx] : g_entries[idx];   // the chain
              c0 = e & 0x7fu;
              c1 = (e >> 7) & 0x7fu;
              ring[g & 1][j][le] = e;
              ret = (over ? 0.0f : ret) + reward_list[(e >> 19) & 0xffu];
              over = (int)((e >> 16) & 1u);
            }
          }
        }
      
      __syncthreads();
    }
  } else if (!loader) {
    for (int g = 0; g <= n_groups; ++g) {
        if (g > 0) {
          const int gp = g - 1, t0 = gp * kG, rb = gp & 1;
          const int n = (T - t0 < kG) ? T - t0 : kG;
          const int64_t plane = trace_plane;  // from one moving thing's trace to the next's
          constexpr int QA = E / 4, kItA = (kG * QA + CL - 1) / CL;
  #pragma unroll 1   // (unrolled, the four iterations' lookups pile up in registers and spill)
          for (int it = 0; it < kItA; ++it) {
            const int item = clane + it * CL;
            const int j = item / QA, q = item % QA;
            const int64_t e0 = env0 + 4 * q;
            if (j < n && e0 < B) {
              const u32x4 e4 = *reinterpret_cast<const u32x4*>(&ring[rb][j][4 * q]);
              const uint32_t e[4] = {e4.x, e4.y, e4.z, e4.w};
              uint32_t rw[4], dc[4];
  #pragma unroll
              for (int i = 0; i < 4; ++i) {
                rw[i] = __float_as_uint(reward_list[(e[i] >> 19) & 0xffu]);
                dc[i] = ((e[i] >> 16) & 1u) ? 0u : 0x3f800000u;
              }
              const int64_t at = (int64_t)(t0 + j) * B + e0;
              if (e0 + 4 <= B) {   // (dword-aligned; 16-byte aligned when B % 4 == 0)
                if (out.reward) {
                  const u32x4 r4 = {rw[0], rw[1], rw[2], rw[3]};
                  store16_update(out.reward + at, r4);
                }
                if (out.discount) {
                  const u32x4 d4 = {dc[0], dc[1], dc[2], dc[3]};
                  store16_update(out.discount + at, d4);
                }
              } else {
                for (int i = 0; i < 4 && e0 + i < B; ++i) {
                  if (out.reward) out.reward[at + i] = __uint_as_float(rw[i]);
                  if (out.discount) out.discount[at + i] = __uint_as_float(dc[i]);
                }
              }
            }
          }
          constexpr int QB = E / 16, kItB = (kG * QB + CL - 1) / CL;
  #pragma unroll
          for (int it = 0; it < kItB; ++it) {
            const int item = clane + it * CL;
            const int j = item / QB, q = item % QB;
            const int64_t e0 = env0 + 16 * q;
            if (item < kG * QB && j < n && e0 < B) {
              uint32_t ta[4], tb[4], dn[4], pf[4];
  #pragma unroll
              for (int k = 0; k < 4; ++k) {
                const u32x4 e4 = *reinterpret_cast<const u32x4*>(&ring[rb][j][16 * q + 4 * k]);
                const uint32_t e[4] = {e4.x, e4.y, e4.z, e4.w};
                uint32_t a[4], b[4], d[4], p[4];
  #pragma unroll
                for (int i = 0; i < 4; ++i) {
                  a[i] = (e[i] & 0x7fu) | (((e[i] >> 14) & 1u) << 7);
                  b[i] = ((e[i] >> 7) & 0x7fu) | (((e[i] >> 15) & 1u) << 7);
                  d[i] = (e[i] >> 16) & 1u;
                  p[i] = (((e[i] >> 17) & 3u) - 1u) & 0xffu;
                }
                ta[k] = pack4(a[0], a[1], a[2], a[3]);
                tb[k] = pack4(b[0], b[1], b[2], b[3]);
                dn[k] = pack4(d[0], d[1], d[2], d[3]);
                pf[k] = pack4(p[0], p[1], p[2], p[3]);
              }
              const int64_t at = (int64_t)(t0 + j) * B + e0;
              if (e0 + 16 <= B) {  // (byte-aligned unless B % 16 == 0: unaligned stores are legal here)
                const u32x4 a4 = {ta[0], ta[1], ta[2], ta[3]}, b4 = {tb[0], tb[1], tb[2], tb[3]};
                store16_update(out.trace + at, a4);
                store16_update(out.trace + plane + at, b4);
                if (out.done) {
                  const u32x4 d4 = {dn[0], dn[1], dn[2], dn[3]};
                  store16_update(out.done + at, d4);
                }
                if (out.perf) {
                  const u32x4 p4 = {pf[0], pf[1], pf[2], pf[3]};
                  store16_update(out.perf + at, p4);
                }
              } else {
                for (int i = 0; i < 16 && e0 + i < B; ++i) {
                  const int sh = (i & 3) * 8;
                  out.trace[at + i] = (uint8_t)(ta[i >> 2] >> sh);
                  out.trace[plane + at + i] = (uint8_t)(tb[i >> 2] >> sh);
                  if (out.done) out.done[at + i] = (uint8_t)(dn[i >> 2] >> sh);
                  if (out.perf) out.perf[at + i] = (int8_t)(pf[i >> 2] >> sh);
                }
              }
            }
          }
        }
      
      __syncthreads();
    }
  } else {
    for (int g = 0; g <= n_groups; ++g) {
        const int c = g / kGroupsPerChunk, phase = g % kGroupsPerChunk;
        const int t_next = (c + 1) * kChunk;
        if (t_next < T) {
          if (phase == 0) ld.issue(actions, B, T, t_next, env0, llane);
          if (phase == kGroupsPerChunk - 1)
            bad += ld.land(staged[(c + 1) & 1], actions, B, T, t_next, env0, llane);
        }
      
      __syncthreads();
    }
  }

  if (live) {
    st.pos[env] = (int8_t)(c0 / (uint32_t)W);
    st.pos[B + env] = (int8_t)(c0 % (uint32_t)W);
    st.pos[2 * B + env] = (int8_t)(c1 / (uint32_t)W);
    st.pos[3 * B + env] = (int8_t)(c1 % (uint32_t)W);
    st.done[env] = (uint8_t)over;
    if (st.ret) st.ret[env] = ret;
  }
  report_bad_actions(out, bad);
}

// ---------------------------------------------------------------------------
// Three- and four-mover games: the same producer / consumer / loader layout over a
// direct-indexed (cell, cell, cell[, cell], action) table in GLOBAL memory
// (campx_tuple_table_build: 4.4 MB for three movers on a 6x8 board, 212 MB for four).
// 64-bit entries: bits 0-27 the things' cells after the frame (7 bits each), 28-31
// whether each is the character its cell shows, 32 done, 33-34 perf + 1, 35-42 index into
// the reward list.  The dependent chain is one global load per frame, so the kernel wants
// every environment in flight at once: 8-frame groups keep the ring small enough for two
// 256-environment workgroups per CU.
struct TupleParams {
  int32_t rows, cols, n_dyn, n_layers;
  int32_t row0[CAMPX_MAX_DYN], col0[CAMPX_MAX_DYN], dyn_layer[CAMPX_MAX_DYN];
};

constexpr int kTupleGroup = 8;
constexpr int64_t kTupleTableMaxBytes = 512ll << 20;  // four movers on a 6x8 board: 212 MB

template <int K>
__device__ __forceinline__ uint32_t tuple_index(uint32_t cells, uint32_t HW) {
  uint32_t idx = cells & 0x7fu;
#pragma unroll
  for (int k = 1; k < K; ++k) idx = idx * HW + ((cells >> (7 * k)) & 0x7fu);
  return idx * CAMPX_N_ACTIONS;
}

// Four waves per SIMD = two workgroups per CU, so that 131 072 environments are all in
// flight at once (four movers: 131 -> 128 VGPRs; sokoban level 2 87 -> 65 us per launch).
#ifndef CAMPX_TUPLE_MINWAVES
#define CAMPX_TUPLE_MINWAVES 4
#endif

template <int K, int kProd, int kCons>
__global__ __launch_bounds__((kProd + kCons + update_loaders(kProd)) * kWave,
                             CAMPX_TUPLE_MINWAVES) void update_tuple_kernel(
    TupleParams tp, CampxState st, const int8_t* __restrict__ actions, CampxOutputs out, int64_t B,
    int32_t T, int32_t reset_first, int64_t trace_plane) {
  constexpr int kLoad = update_loaders(kProd), kG = kTupleGroup;
  constexpr int E = kProd * kWave, CL = kCons * kWave, kThreads = (kProd + kCons + kLoad) * kWave;
  __shared__ float reward_list[256];
  __shared__ __attribute__((aligned(16))) int8_t staged[2][(kChunk / 2) * E];
  __shared__ __attribute__((aligned(16))) uint64_t ring[2][kG][E];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const bool producer = wave < kProd, loader = wave >= kProd + kCons;
  const int llane = (int)threadIdx.x - (kProd + kCons) * kWave;
  const int W = tp.cols;
  const uint32_t HW = (uint32_t)(tp.rows * tp.cols);
  const int64_t env0 = (int64_t)blockIdx.x * E;
  const float* g_rewards = static_cast<const float*>(st.pair_table);
  const uint64_t* g_entries = reinterpret_cast<const uint64_t*>(g_rewards + 256);
  for (int i = threadIdx.x; i < 256; i += kThreads) reward_list[i] = g_rewards[i];
  ActionLoader<E, kLoad> ld;
  int bad = 0;
  if (loader && T > 0) {
    ld.issue(actions, B, T, 0, env0, llane);
    bad += ld.land(staged[0], actions, B, T, 0, env0, llane);
  }
  const int le = wave * kWave + lane;
  const int64_t env = env0 + le;
  const bool live = producer && env < B;
  uint32_t init = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) init |= (uint32_t)(tp.row0[k] * W + tp.col0[k]) << (7 * k);
  uint32_t cells = init;
  int over = 0;
  float ret = 0.0f;
  if (live && !reset_first) {
    cells = 0;
#pragma unroll
    for (int k = 0; k < K; ++k)
      cells |= (uint32_t)((int)st.pos[(int64_t)(2 * k) * B + env] * W +
                          (int)st.pos[(int64_t)(2 * k + 1) * B + env]) << (7 * k);
    over = st.done[env];
    if (st.ret) ret = st.ret[env];
  }
  const int clane = (int)threadIdx.x - kProd * kWave;
  constexpr int kGroupsPerChunk = kChunk / kG;
  __syncthreads();

  const int n_groups = (T + kG - 1) / kG;
  if (producer) {
    for (int g = 0; g <= n_groups; ++g) {
      if (g < n_groups) {
        const int t0 = g * kG;
        const int8_t* chunk = staged[(t0 / kChunk) & 1];
        const int n = (T - t0 < kG) ? T - t0 : kG;
        uint32_t act[kG];
#pragma unroll
        for (int j = 0; j < kG; ++j) act[j] = staged_action<E>(chunk, t0 & (kChunk - 1), j, le);
#pragma unroll
        for (int j = 0; j < kG; ++j) {
          if (j < n) {
            cells = over ? init : cells;  // rebuilt from the art before its next action
            const uint64_t e = g_entries[tuple_index<K>(cells, HW) + act[j]];  // the chain
            cells = (uint32_t)e & 0x0fffffffu;
            ring[g & 1][j][le] = e;
            ret = (over ? 0.0f : ret) + reward_list[(uint32_t)(e >> 35) & 0xffu];
            over = (int)((e >> 32) & 1u);
          }
        }
      }
      __syncthreads();
    }
  } else if (!loader) {
    for (int g = 0; g <= n_groups; ++g) {
      if (g > 0) {
        const int gp = g - 1, t0 = gp * kG, rb = gp & 1;
        const int n = (T - t0 < kG) ? T - t0 : kG;
        const int64_t plane = trace_plane;  // from one moving thing's trace to the next's
        // ---- float streams: item = (frame j, 4 environments)
        constexpr int QA = E / 4, kItA = (kG * QA + CL - 1) / CL;
#pragma unroll 1
        for (int it = 0; it < kItA; ++it) {
          const int item = clane + it * CL;
          const int j = item / QA, q = item % QA;
          const int64_t e0 = env0 + 4 * q;
          if (item < kG * QA && j < n && e0 < B) {
            uint32_t rw[4], dc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const uint32_t hi = (uint32_t)(ring[rb][j][4 * q + i] >> 32);
              rw[i] = __float_as_uint(reward_list[(hi >> 3) & 0xffu]);
              dc[i] = (hi & 1u) ? 0u : 0x3f800000u;
            }
            const int64_t at = (int64_t)(t0 + j) * B + e0;
            if (e0 + 4 <= B) {   // (dword-aligned; 16-byte aligned when B % 4 == 0)
              if (out.reward) {
                const u32x4 r4 = {rw[0], rw[1], rw[2], rw[3]};
                store16_update(out.reward + at, r4);
              }
              if (out.discount) {
                const u32x4 d4 = {dc[0], dc[1], dc[2], dc[3]};
                store16_update(out.discount + at, d4);
              }
            } else {
              for (int i = 0; i < 4 && e0 + i < B; ++i) {
                if (out.reward) out.reward[at + i] = __uint_as_float(rw[i]);
                if (out.discount) out.discount[at + i] = __uint_as_float(dc[i]);
              }
            }
          }
        }
        // ---- byte streams: item = (frame j, 16 environments)
        constexpr int QB = E / 16, kItB = (kG * QB + CL - 1) / CL;
#pragma unroll 1
        for (int it = 0; it < kItB; ++it) {
          const int item = clane + it * CL;
          const int j = item / QB, q = item % QB;
          const int64_t e0 = env0 + 16 * q;
          if (item < kG * QB && j < n && e0 < B) {
            uint32_t tr[K][4], dn[4], pf[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              uint32_t lo[4], hi[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const uint64_t e = ring[rb][j][16 * q + 4 * w + i];
                lo[i] = (uint32_t)e;
                hi[i] = (uint32_t)(e >> 32);
              }
#pragma unroll
              for (int k = 0; k < K; ++k) {
                uint32_t b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                  b[i] = ((lo[i] >> (7 * k)) & 0x7fu) | (((lo[i] >> (28 + k)) & 1u) << 7);
                tr[k][w] = pack4(b[0], b[1], b[2], b[3]);
              }
              dn[w] = pack4(hi[0] & 1u, hi[1] & 1u, hi[2] & 1u, hi[3] & 1u);
              pf[w] = pack4(((hi[0] >> 1) & 3u) - 1u, ((hi[1] >> 1) & 3u) - 1u,
                            ((hi[2] >> 1) & 3u) - 1u, (((hi[3] >> 1) & 3u) - 1u) & 0xffu);
            }
            const int64_t at = (int64_t)(t0 + j) * B + e0;
            if (e0 + 16 <= B) {  // (byte-aligned unless B % 16 == 0: unaligned stores are legal here)
#pragma unroll
              for (int k = 0; k < K; ++k) {
                const u32x4 t4 = {tr[k][0], tr[k][1], tr[k][2], tr[k][3]};
                store16_update(out.trace + k * plane + at, t4);
              }
              if (out.done) {
                const u32x4 d4 = {dn[0], dn[1], dn[2], dn[3]};
                store16_update(out.done + at, d4);
              }
              if (out.perf) {
                const u32x4 p4 = {pf[0], pf[1], pf[2], pf[3]};
                store16_update(out.perf + at, p4);
              }
            } else {
              for (int i = 0; i < 16 && e0 + i < B; ++i) {
                const int sh = (i & 3) * 8;
#pragma unroll
                for (int k = 0; k < K; ++k) out.trace[k * plane + at + i] = (uint8_t)(tr[k][i >> 2] >> sh);
                if (out.done) out.done[at + i] = (uint8_t)(dn[i >> 2] >> sh);
                if (out.perf) out.perf[at + i] = (int8_t)(pf[i >> 2] >> sh);
              }
            }
          }
        }
      }
      __syncthreads();
    }
  } else {
    for (int g = 0; g <= n_groups; ++g) {
      const int c = g / kGroupsPerChunk, phase = g % kGroupsPerChunk;
      const int t_next = (c + 1) * kChunk;
      if (t_next < T) {
        if (phase == 0) ld.issue(actions, B, T, t_next, env0, llane);
        if (phase == kGroupsPerChunk - 1)
          bad += ld.land(staged[(c + 1) & 1], actions, B, T, t_next, env0, llane);
      }
      __syncthreads();
    }
  }

  if (live) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const uint32_t c = (cells >> (7 * k)) & 0x7fu;
      st.pos[(int64_t)(2 * k) * B + env] = (int8_t)(c / (uint32_t)W);
      st.pos[(int64_t)(2 * k + 1) * B + env] = (int8_t)(c % (uint32_t)W);
    }
    st.done[env] = (uint8_t)over;
    if (st.ret) st.ret[env] = ret;
  }
  report_bad_actions(out, bad);
}

// Engine.play() for three- and four-mover games with their table: the one-frame kernel of
// step_pair_kernel over the 64-bit entries.
template <int K, bool kBoard>
__global__ __launch_bounds__(kWave) void step_tuple_kernel(
    TupleParams tp, const CampxSpec* __restrict__ spec, CampxState st,
    const int8_t* __restrict__ actions, CampxOutputs out, int64_t B, int32_t reset_first) {
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];
  const int lane = threadIdx.x;
  const int W = tp.cols, HW = tp.rows * tp.cols, LHW = tp.n_layers * HW;
  const int64_t env0 = (int64_t)blockIdx.x * kWave;
  const int64_t env = env0 + lane;
  const bool live = env < B;
  const int n_live = (B - env0 < kWave) ? (int)(B - env0) : kWave;
  int8_t* obs_img = lds;
  int8_t* board_img = lds + ((kWave * LHW + 15) & ~15);
  const float* g_rewards = static_cast<const float*>(st.pair_table);
  const uint64_t* g_entries = reinterpret_cast<const uint64_t*>(g_rewards + 256);

  uint32_t init = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) init |= (uint32_t)(tp.row0[k] * W + tp.col0[k]) << (7 * k);
  uint32_t cells = init;
  int over = 0, a = 4;
  float ret = 0.0f;
  if (live) {
    a = actions[env];
    if (!reset_first) {
      cells = 0;
#pragma unroll
      for (int k = 0; k < K; ++k)
        cells |= (uint32_t)((int)st.pos[(int64_t)(2 * k) * B + env] * W +
                            (int)st.pos[(int64_t)(2 * k + 1) * B + env]) << (7 * k);
      over = st.done[env];
      if (st.ret) ret = st.ret[env];
    }
  }
  u32x4 v_obs[kStepObsLoads], v_board[kStepBoardLoads];
  int k_obs = (lane * 16) % LHW, k_board = (lane * 16) % HW;
  fill_issue<kStepObsLoads>(v_obs, spec->rot_obs, LHW, k_obs);
  if (kBoard) fill_issue<kStepBoardLoads>(v_board, spec->rot_board, HW, k_board);

  const int bad = (live && (unsigned)a > 4u) ? 1 : 0;
  a = ((unsigned)a > 4u) ? 4 : a;
  if (over) {  // rebuilt from the art before its next action
    cells = init;
    ret = 0.0f;
  }
  const uint64_t e = g_entries[tuple_index<K>(cells, (uint32_t)HW) + (uint32_t)a];
  fill_land<kStepObsLoads>(v_obs, obs_img, kWave * LHW, 0, lane);
  fill_rest<kStepObsLoads>(obs_img, kWave * LHW, LHW, spec->rot_obs, k_obs, lane);
  if (kBoard) {
    fill_land<kStepBoardLoads>(v_board, board_img, kWave * HW, 0, lane);
    fill_rest<kStepBoardLoads>(board_img, kWave * HW, HW, spec->rot_board, k_board, lane);
  }
  const uint32_t lo = (uint32_t)e, hi = (uint32_t)(e >> 32);
  const float reward = g_rewards[(hi >> 3) & 0xffu];
  const int done = (int)(hi & 1u);
  ret += reward;
  int8_t* my_obs = obs_img + lane * LHW;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int c = (int)((lo >> (7 * k)) & 0x7fu);
    if ((lo >> (28 + k)) & 1u) {   // it is the character its cell shows
      my_obs[(int)spec->static_top_layer[c] * HW + c] = 0;
      my_obs[tp.dyn_layer[k] * HW + c] = 1;
      if (kBoard) board_img[lane * HW + c] = (int8_t)spec->layer_char[tp.dyn_layer[k]];
    }
  }
  if (live) {
    if (out.reward) out.reward[env] = reward;
    if (out.discount) out.discount[env] = done ? 0.0f : 1.0f;
    if (out.done) out.done[env] = (uint8_t)done;
    if (out.perf) out.perf[env] = (int8_t)((int)((hi >> 1) & 3u) - 1);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const uint32_t c = (lo >> (7 * k)) & 0x7fu;
      st.pos[(int64_t)(2 * k) * B + env] = (int8_t)(c / (uint32_t)W);
      st.pos[(int64_t)(2 * k + 1) * B + env] = (int8_t)(c % (uint32_t)W);
    }
    st.done[env] = (uint8_t)done;
    if (st.ret) st.ret[env] = ret;
  }
  step_stream_obs(obs_img, out, env0 * LHW, n_live * LHW, lane);
  if (kBoard) stream_out<false>(board_img, out.board + env0 * HW, n_live * HW, lane);
  report_bad_actions(out, bad);
}

// ---------------------------------------------------------------------------
// Split path, second half: expand the trace into the observation stream.
// One-shot blocks, ONE aligned 16-byte store per thread, block (x, t) writing bytes
// [x*4096, (x+1)*4096) of frame t: the dispatcher walks the output linearly.  That
// is the store pattern that reaches the HBM write ceiling on this chip
// (tools/probes: 6.9 TB/s, against 5.4 TB/s for long-lived waves that each stream a
// private tile, whatever the tile size).
//
// A frame is B rows of R bytes (R = L*H*W for the layered board, H*W for the flat
// board).  A row is the scenery's row with at most two bytes changed per moving
// thing, given by its trace entry.  spec->rot_* hold 16 byte-rotations of the
// cyclically continued scenery row, so any 16-byte window of back-to-back rows is
// ONE aligned 16-byte load (L1-resident).
struct RenderParams {
  uint32_t R;                 // row bytes
  uint32_t m, sh1, sh2;       // exact n / R for 32-bit n (Granlund-Montgomery)
  uint32_t slab_bytes;        // B * R
  uint32_t shift_base, shift_slab;  // (address of frame 0) and slab_bytes modulo the window span
  int32_t n_dyn, is_board, cells;
  int64_t B;
  int32_t dyn_char[CAMPX_MAX_DYN];
  int32_t dyn_off[CAMPX_MAX_DYN];   // byte offset of moving thing d's layer inside a row
};

// Block (x, t) writes bytes [x*4096*kWin, (x+1)*4096*kWin) of frame t; each of its four
// waves owns kWin aligned KiB of it (every store instruction of a wave is one aligned,
// contiguous KiB: tools/probes show -12..-21 % for anything less aligned).
//
// A wave first lays the scenery's bytes for its window into LDS (one aligned 16-byte
// load from the rotated scenery table per lane), then the few lanes that hold a
// patch - (row overlapping the window) x (moving thing) x (set | clear) - write their
// single byte into it, then every lane reads its 16 bytes back and stores them.
// A patch comes from the thing's trace byte (cell, visible): the 1 it paints is at
// dyn_off + cell, the scenery's 1 it hides at scen_off[cell], a per-wave LDS copy of
// spec->static_top_layer[cell] * cells + cell.  Nothing is shared between waves, so
// there is no workgroup barrier.
// A/B knobs of the render kernel's shape: waves per block, KiB windows per wave, and
// whether block indices are remapped so that each XCD (block b runs on XCD b % 8) sweeps
// its own contiguous eighth of a frame instead of every eighth block of it (neighbouring
// windows then share trace lines inside ONE L2).  Measured with rocprofv3, avg of 63
// launches (gpurun_out/r2c), boat race / wall world / sokoban render in us:
//   4 waves, 2 KiB, no remap   176.9 / 2006.7 / 361.8
//   4 waves, 2 KiB, remap      173.5 / 1877.2 / 362.3
//   2 waves, 2 KiB, remap      172.7 / 1873.7 / 363.6   <- default
//   4 waves, 4 KiB, remap      190.3 / 2099.3 / 362.3
//   2 waves, 4 KiB, remap      177.7 / 1984.6 / 356.0
// Again with settled clocks (250 launches after 50 warm-up ones, gpurun_out/t10, remap on):
//   2 waves x 2 KiB 164.8 / 1842 / 338.1 (default)   4 x 2: 165.6 / 1873 / 340.0
//   1 x 2: 166.1 / 1893 / 345.3   2 x 4: 180.2 / 1978 / 360.9   1 x 4: 179.4 / 2008 / 371.4
//   4 x 1: 187.7 / 1994 / 388.7
#ifndef CAMPX_RENDER_WAVES
#define CAMPX_RENDER_WAVES 2
#endif
#ifndef CAMPX_RENDER_WIN
#define CAMPX_RENDER_WIN 2
#endif
#ifndef CAMPX_RENDER_XCD
#define CAMPX_RENDER_XCD 1
#endif
constexpr int kRenderWaves = CAMPX_RENDER_WAVES;

// kOdd: B * R is not a multiple of 16, so frames do not start on a 16-byte boundary and a
// lane's (memory-aligned) 16-byte chunk can straddle two frames.  A chunk belongs to the
// frame it STARTS in and is written whole, with the first bytes of the next frame's first
// row (the trace plane is [T * B] rows: row B of this frame is row 0 of the next); only at
// the two ends of a launch is a chunk written byte by byte - the part after the first
// frame's start, the part before the last frame's end.
template <int K, bool kBoard, bool kNT, int kWin, int kFmt, bool kOdd = false>
__global__ __launch_bounds__(kRenderWaves * kWave) void render_kernel(RenderParams rp,
                                                     const CampxSpec* __restrict__ spec,
                                                     const uint8_t* __restrict__ trace,
                                                     int8_t* __restrict__ dst, int64_t n_rows) {
  __shared__ __attribute__((aligned(16))) int8_t lds[kRenderWaves * kWin * 1024];
  __shared__ uint16_t scen_off_all[kRenderWaves][CAMPX_MAX_CELLS];
  // readfirstlane: the wave index is uniform, and saying so keeps everything derived
  // from it (window offsets, the divisions, base addresses) on the scalar unit
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // this wave's kWin consecutive KiB windows of the frame
  uint32_t bx = blockIdx.x;
#if CAMPX_RENDER_XCD
  bx = (bx & 7u) * (gridDim.x >> 3) + (bx >> 3);     // gridDim.x is a multiple of 8
#endif
  // Windows are aligned in MEMORY, not in the frame: when a frame does not start on a
  // window boundary (B * R not a multiple of the span) they start `shift` bytes before it,
  // so every store of every frame is still one aligned KiB (499 984 environments: 4.2 TB/s
  // with frame-aligned windows).  Offsets are modulo 2^32: the head window's start is
  // "negative", its lanes before the frame fail the one `off < slab_bytes` test below.
  const uint32_t span = 1024u * kWin;
  const uint32_t shift = (rp.shift_base + blockIdx.y * rp.shift_slab) & (span - 1u);
  const uint32_t widx = bx * (uint32_t)kRenderWaves + (uint32_t)wave;
  if ((uint64_t)widx * span >= (uint64_t)rp.slab_bytes + shift) return;
  const uint32_t woff0 = widx * span - shift;
  const uint32_t wlo = widx * span < shift ? 0u : woff0;   // first byte inside the frame
  int8_t* win0 = lds + wave * (kWin * 1024);
  uint16_t* scen_off = scen_off_all[wave];
  const int R = (int)rp.R;
  const int pitch = ((R + 15) & ~15) + 16;
  const int8_t* rot = kBoard ? spec->rot_board : spec->rot_obs;
  constexpr int P = kBoard ? K : 2 * K;                // patches per row
  const uint8_t* frame_trace = trace + (int64_t)blockIdx.y * rp.B;

  // ---- patches: (row overlapping the windows) x (moving thing) x (set | clear).
  // Their trace bytes come from HBM / L2: issue those loads first.
  const uint32_t whi = __umulhi(rp.m, wlo);
  const uint32_t first_row = (((wlo - whi) >> rp.sh1) + whi) >> rp.sh2;
  const bool last_frame = blockIdx.y == gridDim.y - 1u;
  constexpr uint32_t kOut = kFmt ? 8u : 16u;   // image bytes of a lane's 16-byte store
  const uint32_t frame_end = rp.slab_bytes + ((kOdd && !last_frame) ? kOut - 1u : 0u);   // exclusive
  const uint32_t wend = (woff0 + span - 1u < frame_end) ? woff0 + span - 1u : frame_end - 1u;
  const uint32_t ehi = __umulhi(rp.m, wend);
  const uint32_t last_row = (((wend - ehi) >> rp.sh1) + ehi) >> rp.sh2;
  const int slots = (int)(last_row - first_row + 1u) * P;
  constexpr int kMaxIter = 2;                          // slots <= 128 covers the common case
  uint32_t ent[kMaxIter];
#pragma unroll
  for (int it = 0; it < kMaxIter; ++it) {
    const int sidx = lane + it * kWave;
    const int r = sidx / P, p = sidx - r * P;
    const int d = kBoard ? p : (p >> 1);
    uint32_t row = first_row + (uint32_t)r;
    row = row <= last_row ? row : last_row;            // clamp: slot unused, entry ignored
    ent[it] = (it == 0 || slots > kWave) ? frame_trace[(int64_t)d * n_rows + row] : 0u;
  }

  // ---- scenery: issue all loads, then park them in LDS
  u32x4 scen[kWin];
#pragma unroll
  for (int j = 0; j < kWin; ++j) {
    const uint32_t off = woff0 + j * 1024u + (uint32_t)lane * 16u;
    const uint32_t hi = __umulhi(rp.m, off);
    const uint32_t row = (((off - hi) >> rp.sh1) + hi) >> rp.sh2;  // off / R
    int k = (int)(off - row * rp.R);                                 // off % R
    if (kOdd && off >= 0xfffffff0u) k = R - (int)(0u - off);         // starts 1..15 bytes before the frame
    scen[j] = *reinterpret_cast<const u32x4*>(rot + (k & 15) * pitch + (k & ~15));
  }
  // the scenery layer of two cells per lane (kBoard needs none of it)
  uint32_t top2 = 0;
  if (!kBoard) top2 = *reinterpret_cast<const uint16_t*>(spec->static_top_layer + 2 * lane);

#pragma unroll
  for (int j = 0; j < kWin; ++j)
    *reinterpret_cast<u32x4*>(win0 + j * 1024 + lane * 16) = scen[j];
  if (!kBoard) {
    const uint32_t c = 2u * (uint32_t)lane;
    const uint32_t lo = (top2 & 0xffu) * (uint32_t)rp.cells + c;
    const uint32_t hi2 = (top2 >> 8) * (uint32_t)rp.cells + c + 1u;
    *reinterpret_cast<uint32_t*>(scen_off + c) = lo | (hi2 << 16);
  }

  // rp.dyn_off / dyn_char of a lane-varying thing: a chain of selects over the K kernel
  // arguments (indexing the array made hipcc fetch it with a vector load from the kernarg
  // segment and wait for it inside the patch branch: one more memory trip per wave)
  // (readfirstlane makes each argument an opaque scalar: from three things up the
  // optimiser otherwise turns the select chain back into the indexed load)
  auto of_thing = [&](const int32_t (&arr)[CAMPX_MAX_DYN], int d) {
    int v = __builtin_amdgcn_readfirstlane(arr[0]);
#pragma unroll
    for (int k = 1; k < K; ++k) v = (d == k) ? __builtin_amdgcn_readfirstlane(arr[k]) : v;
    return v;
  };
  auto apply = [&](int sidx, uint32_t e) {
    const int r = sidx / P, p = sidx - r * P;
    const int d = kBoard ? p : (p >> 1);
    const int cell = (int)(e & 0x7fu);
    int byte;   // offset inside the row
    int8_t val;
    if (kBoard) {
      byte = cell;
      val = (int8_t)of_thing(rp.dyn_char, d);
    } else {
      byte = (p & 1) ? of_thing(rp.dyn_off, d) + cell : (int)scen_off[cell];
      val = (int8_t)(p & 1);
    }
    // offsets inside a frame fit 32 bits (split_ok); a patch left of the window wraps to a
    // huge unsigned value and fails the one comparison
    const uint32_t at = (first_row + (uint32_t)r) * (uint32_t)R + (uint32_t)byte - woff0;
    if (sidx < slots && (e >> 7) && at < span) win0[at] = val;
  };
#pragma unroll
  for (int it = 0; it < kMaxIter; ++it) apply(lane + it * kWave, ent[it]);
  for (int sidx = lane + kMaxIter * kWave; sidx < slots; sidx += kWave) {   // tiny rows only
    const int r = sidx / P, p = sidx - r * P;
    const int d = kBoard ? p : (p >> 1);
    apply(sidx, frame_trace[(int64_t)d * n_rows + first_row + (uint32_t)r]);
  }

  // ---- out: aligned, contiguous KiB stores
  if (kFmt == 0) {
#pragma unroll
    for (int j = 0; j < kWin; ++j) {
      const uint32_t off = woff0 + j * 1024u + (uint32_t)lane * 16u;
      const int8_t* frame = dst + (int64_t)blockIdx.y * rp.slab_bytes;   // uniform
      if (off < rp.slab_bytes) {                         // the chunk starts inside the frame
        const u32x4 v = *reinterpret_cast<const u32x4*>(win0 + j * 1024 + lane * 16);
        if (kOdd && last_frame && off + 16u > rp.slab_bytes) {   // the launch's last bytes
          const uint32_t w[4] = {v.x, v.y, v.z, v.w};
          for (uint32_t i = 0; off + i < rp.slab_bytes; ++i)
            const_cast<int8_t*>(frame)[off + i] = (int8_t)(w[i >> 2] >> ((i & 3u) * 8u));
        } else if (kNT) {
          store16_streaming_at(frame, off, v);
        } else {
          *reinterpret_cast<u32x4*>(const_cast<int8_t*>(frame) + off) = v;
        }
      } else if (kOdd && blockIdx.y == 0 && off >= 0xfffffff0u) {   // the launch's first bytes
        const u32x4 v = *reinterpret_cast<const u32x4*>(win0 + j * 1024 + lane * 16);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        for (uint32_t i = 0u - off; i < 16u; ++i)
          const_cast<int8_t*>(frame)[(int32_t)(off + i)] = (int8_t)(w[i >> 2] >> ((i & 3u) * 8u));
      }
    }
  } else {
    // 16-bit observations (f16 / bf16 0.0 and 1.0) for a policy network: a lane turns
    // 8 bytes of the image into 8 halves; lanes stay contiguous, so each store
    // instruction is again one aligned KiB (of 512 elements).
    constexpr uint32_t kOne = (kFmt == 1) ? 0x3C00u : 0x3F80u;
#pragma unroll
    for (int h = 0; h < 2 * kWin; ++h) {
      const uint32_t elem = woff0 + (uint32_t)h * 512u + (uint32_t)lane * 8u;  // in the frame
      uint16_t* frame16 = reinterpret_cast<uint16_t*>(dst) + (int64_t)blockIdx.y * rp.slab_bytes;
      const bool inside = elem < rp.slab_bytes;
      const bool head = kOdd && blockIdx.y == 0 && elem >= 0xfffffff8u;   // the launch's first elements
      if (inside || head) {
        const uint2 b = *reinterpret_cast<const uint2*>(win0 + h * 512 + lane * 8);
        u32x4 v;
        v.x = ((b.x & 0xffu) | ((b.x << 8) & 0x00ff0000u)) * kOne;
        v.y = (((b.x >> 16) & 0xffu) | ((b.x >> 8) & 0x00ff0000u)) * kOne;
        v.z = ((b.y & 0xffu) | ((b.y << 8) & 0x00ff0000u)) * kOne;
        v.w = (((b.y >> 16) & 0xffu) | ((b.y >> 8) & 0x00ff0000u)) * kOne;
        if (head || (kOdd && last_frame && elem + 8u > rp.slab_bytes)) {   // element by element
          const uint32_t w[4] = {v.x, v.y, v.z, v.w};
          for (uint32_t i = head ? 0u - elem : 0u; i < 8u && (head || elem + i < rp.slab_bytes); ++i)
            frame16[(int32_t)(elem + i)] = (uint16_t)(w[i >> 1] >> ((i & 1u) * 16u));
        } else {
          u32x4* o = reinterpret_cast<u32x4*>(frame16 + elem);
          if (kNT)
            store16_streaming(o, v);
          else
            *o = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Shape tier (include/campx_hip.h): Hello World.  One wavefront = one environment; its
// two H*W-byte images (the environment's backdrop, which sprites behind the first drape
// paint into for good, and the frame's flat board of layer indices) live in LDS; all
// control flow is wave-uniform and LDS operations of a wave complete in order, so nothing
// needs a barrier (the one per frame only keeps a workgroup's waves in step, for the store
// pattern).  A frame: offsets += per-thing delta[action] (scalar), paint, then every lane
// expands eight board cells at a time into the L layer planes (one 8-byte store per plane)
// - campx/rendering.py:204-215's per-character equality.
#ifndef CAMPX_SHAPE_WAVES
#define CAMPX_SHAPE_WAVES 4
#endif
constexpr int kShapeWaves = CAMPX_SHAPE_WAVES;

__device__ __forceinline__ int shape_cell(uint32_t packed, int orow, int ocol, int H, int W) {
  int r = (int)(packed >> 8) + orow, c = (int)(packed & 0xffu) + ocol;
  r = r >= H ? r - H : r;
  c = c >= W ? c - W : c;
  return r * W + c;
}

// Per-action effect of a frame on all things at once (built once per workgroup): the
// offsets of up to eight things are one byte each in two 32-bit words per coordinate, so
// a frame's whole update pass is four SWAR add-and-wrap on the scalar unit.
struct ShapeAction {
  uint32_t drow[2], dcol[2];  // byte k of word k / 4: thing k's offset change, 0 .. rows-1 / cols-1
  float reward;               // summed in update-schedule order (plot.py:208-211: r + total)
  uint32_t flags;             // bit 0: somebody terminates the episode; bit 1: somebody rewards
};

// What the kernel needs of a CampxShapeSpec besides its cell lists, by value in the
// kernel arguments (scalar loads; campx_shape_rollout_launch builds it on the host).
struct ShapeParams {
  int32_t rows, cols, n_layers, n_things, first_drape, n_list;
  uint32_t thing[CAMPX_SHAPE_MAX_THINGS];  // cell_begin | n_cells << 11 | layer << 23 | visible << 28
  ShapeAction act[CAMPX_N_ACTIONS];
  uint32_t layer_char[CAMPX_MAX_LAYERS / 4];
};

// bytes of r, d < n <= 127: (r + d) mod n per byte
__device__ __forceinline__ uint32_t swar_add_wrap(uint32_t r, uint32_t d, uint32_t n) {
  const uint32_t t = r + d;
  const uint32_t ge = (t + (0x80u - n) * 0x01010101u) & 0x80808080u;  // bit 7: byte >= n
  return t - (ge >> 7) * n;
}

#ifndef CAMPX_SHAPE_MINWAVES
#define CAMPX_SHAPE_MINWAVES 1
#endif

template <bool kBoard>
__global__ __launch_bounds__(kShapeWaves * kWave, CAMPX_SHAPE_MINWAVES) void shape_rollout_kernel(
    ShapeParams sp, const CampxShapeSpec* __restrict__ spec, CampxState st,
    int8_t* __restrict__ backdrop_state, const int8_t* __restrict__ actions, CampxOutputs out,
    int64_t B, int32_t T, int32_t reset_first, int32_t emit_first) {
  __shared__ __attribute__((aligned(16))) uint8_t lds_backdrop[kShapeWaves][CAMPX_SHAPE_MAX_CELLS];
  __shared__ __attribute__((aligned(16))) uint8_t lds_board[kShapeWaves][CAMPX_SHAPE_MAX_CELLS];
  __shared__ uint16_t lds_cells[CAMPX_SHAPE_MAX_LIST];  // the things' shapes, once per workgroup
  __shared__ uint32_t lds_char[CAMPX_MAX_LAYERS / 4];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t env = (int64_t)blockIdx.x * kShapeWaves + wave;
  const int H = sp.rows, W = sp.cols, HW = H * W, L = sp.n_layers, N = sp.n_things;
  for (int i = threadIdx.x; i < sp.n_list; i += kShapeWaves * kWave) lds_cells[i] = spec->cells[i];
  if (kBoard && threadIdx.x < CAMPX_MAX_LAYERS / 4) lds_char[threadIdx.x] = sp.layer_char[threadIdx.x];
  __syncthreads();
  // The waves of a workgroup (four consecutive environments: 13 KB of one frame, contiguous)
  // go through the frames in lockstep, one s_barrier per frame, so that their rows reach
  // HBM together: 6 % faster than free-running waves (2.11 -> 1.99 ms at B = 32 768, 0.331 ->
  // 0.312 at 4 096; 8 / 16 waves per workgroup are slower).  A surplus wave of the last
  // workgroup only keeps the count.
  if (env >= B) {        // wave-uniform
    for (int t = emit_first ? -1 : 0; t < T; ++t) __builtin_amdgcn_s_barrier();
    return;
  }
  const int64_t LHW = (int64_t)L * HW;
  uint8_t* bd = lds_backdrop[wave];
  uint8_t* board = lds_board[wave];
  const bool quads = (HW & 3) == 0;
  const int first_drape = sp.first_drape;

  uint32_t orow[2] = {0u, 0u}, ocol[2] = {0u, 0u};  // byte k of word k / 4: thing k's cyclic offset
  int over = 0;
  float ret = 0.0f;
  const bool fresh = reset_first != 0;
  if (!fresh) {
#pragma unroll
    for (int k = 0; k < CAMPX_SHAPE_MAX_THINGS; ++k)
      if (k < N) {
        orow[k >> 2] |= (uint32_t)(uint8_t)st.pos[(int64_t)(2 * k) * B + env] << (8 * (k & 3));
        ocol[k >> 2] |= (uint32_t)(uint8_t)st.pos[(int64_t)(2 * k + 1) * B + env] << (8 * (k & 3));
      }
    over = st.done[env];
    if (st.ret) ret = st.ret[env];
  }
  // the environment's backdrop, four cells per load when the board allows (env * HW is
  // then a multiple of 4 too; the spec's array and torch allocations are 4-byte aligned)
  const bool from_state = !fresh && backdrop_state != nullptr;
  const uint8_t* bd_src = from_state ? reinterpret_cast<const uint8_t*>(backdrop_state) + env * HW
                                     : spec->backdrop;
  auto load_backdrop = [&](const uint8_t* src) {
    if (quads)
      for (int i = lane; 4 * i < HW; i += kWave)
        reinterpret_cast<uint32_t*>(bd)[i] = reinterpret_cast<const uint32_t*>(src)[i];
    else
      for (int i = lane; i < HW; i += kWave) bd[i] = src[i];
  };
  load_backdrop(bd_src);

  auto rebuild = [&]() {  // a fresh make_game() + its_showtime()
    orow[0] = orow[1] = ocol[0] = ocol[1] = 0u;
    load_backdrop(spec->backdrop);
  };

  // `emit` false: a frame whose observation nobody will see (time strides 0 and not the last
  // frame) - only the sprites that paint into the backdrop (state) are painted.
  auto paint_and_emit = [&](int8_t* obs_dst, int8_t* board_dst, bool emit) {
    // Things back to front.  Sprites behind the first drape paint into the backdrop itself
    // (rendering.py:128,150); the frame's board starts as a copy of it.
    const int n_paint = emit ? N : first_drape;
    for (int z = 0; z < n_paint; ++z) {   // everything about z is scalar
      // (loops over `base` have scalar trip counts: one pass for boards up to 1 024 cells
      // here, for things up to 64 cells below)
      if (z == first_drape)
        for (int base = 0; base * 16 < HW; base += kWave) {   // whole 16-byte chunks of the arrays
          const int i = base + lane;
          if (i * 16 < HW) reinterpret_cast<u32x4*>(board)[i] = reinterpret_cast<const u32x4*>(bd)[i];
        }
      const uint32_t th = sp.thing[z];
      if ((th >> 28) & 1u) {
        const int begin = (int)(th & 0x7ffu), n = (int)((th >> 11) & 0xfffu);
        const uint8_t layer = (uint8_t)((th >> 23) & 0x1fu);
        const int sh = 8 * (z & 3);
        const int dr = (int)(((z < 4 ? orow[0] : orow[1]) >> sh) & 0xffu);
        const int dc = (int)(((z < 4 ? ocol[0] : ocol[1]) >> sh) & 0xffu);
        uint8_t* target = z < first_drape ? bd : board;
        for (int base = 0; base < n; base += kWave) {
          const int i = base + lane;
          if (i < n) {
            const uint32_t packed = lds_cells[begin + i];
            int r = (int)(packed >> 8) + dr, c = (int)(packed & 0xffu) + dc;
            r = r >= H ? r - H : r;
            c = c >= W ? c - W : c;
            target[r * W + c] = layer;
          }
        }
      }
    }
    if (!emit) return;
    // layers by equality (rendering.py:204-215): eight cells per lane, one 8-byte store per
    // layer plane.  A board of 8k + 4 cells: the last lane takes the last eight cells, four of
    // which its neighbour also writes (same values), so every lane runs the same code.
    if (quads && HW >= 8) {
      for (int qbase = 0; 8 * qbase < HW; qbase += kWave) {
        const int q = qbase + lane;
        if (8 * q >= HW) continue;
        const uint32_t at = (uint32_t)(8 * q + 8 <= HW ? 8 * q : HW - 8);   // a multiple of 4
        const uint32_t b0 = *reinterpret_cast<const uint32_t*>(board + at);
        const uint32_t b1 = *reinterpret_cast<const uint32_t*>(board + at + 4);
        int8_t* plane = obs_dst;   // uniform: the stores take it as their scalar base
        uint32_t lc = 0u;
        for (int l = 0; l < L; ++l) {
          // bytes < 0x80: 0x80 - (b ^ l) has bit 7 set iff they are equal
          const uint32_t e0 = ((0x80808080u - (b0 ^ lc)) & 0x80808080u) >> 7;
          const uint32_t e1 = ((0x80808080u - (b1 ^ lc)) & 0x80808080u) >> 7;
          *reinterpret_cast<uint2*>(plane + at) = make_uint2(e0, e1);
          plane += HW;
          lc += 0x01010101u;
        }
        if (kBoard) {
          const uint8_t* ch = reinterpret_cast<const uint8_t*>(lds_char);
          auto chars = [&](uint32_t b4) {
            return (uint32_t)ch[b4 & 0xffu] | ((uint32_t)ch[(b4 >> 8) & 0xffu] << 8) |
                   ((uint32_t)ch[(b4 >> 16) & 0xffu] << 16) | ((uint32_t)ch[b4 >> 24] << 24);
          };
          *reinterpret_cast<uint2*>(board_dst + at) = make_uint2(chars(b0), chars(b1));
        }
      }
    } else {
      for (int i = lane; i < HW; i += kWave) {
        const int b = board[i];
        for (int l = 0; l < L; ++l) obs_dst[(int64_t)l * HW + i] = (int8_t)(b == l);
        if (kBoard) board_dst[i] = (int8_t)reinterpret_cast<const uint8_t*>(lds_char)[b];
      }
    }
  };


  // Actions: lane j holds the action of frame (chunk start + j), one load per 64 frames,
  // fetched a chunk ahead; a frame reads its own with a (wave-uniform) readlane, so the
  // frame loop has no global load on its critical path.
  auto fetch = [&](int t0) {
    const int t = t0 + lane;
    return (t < T) ? (int)actions[(int64_t)t * B + env] : 4;
  };
  int act_now = T > 0 ? fetch(0) : 4, act_next = 4;
  int bad = 0;
  int reward_buf = 0;
  uint64_t over_mask = 0;
  // time strides 0: every frame would overwrite the same slot - emit only the last one
  const bool last_only = out.obs_t_stride == 0 && (!kBoard || out.board_t_stride == 0);
  // frame -1 (emit_first): the its_showtime() observation, no update pass, written where
  // frame 0 goes (one call site for the paint-and-emit code)
  for (int t = emit_first ? -1 : 0; t < T; ++t) {
    __builtin_amdgcn_s_barrier();   // lockstep (see above); nothing in LDS is shared between waves
    const bool showtime = t < 0;
    if (!showtime && (t & (kWave - 1)) == 0) {
      if (t) act_now = act_next;
      act_next = fetch(t + kWave);
    }
    const int a_raw = showtime ? -1 : __builtin_amdgcn_readlane(act_now, t & (kWave - 1));  // wave-uniform
    const bool valid = (unsigned)a_raw < (unsigned)CAMPX_N_ACTIONS;
    bad += (valid || showtime) ? 0 : 1;
    if (over && !showtime) {
      rebuild();
      over = 0;
      ret = 0.0f;
    }
    float reward = __builtin_nanf("");   // an id outside 0..4 moves nothing
    if (valid) {
      const ShapeAction& e = sp.act[a_raw];   // kernel argument, uniform index: scalar loads
      orow[0] = swar_add_wrap(orow[0], e.drow[0], (uint32_t)H);
      ocol[0] = swar_add_wrap(ocol[0], e.dcol[0], (uint32_t)W);
      if (N > 4) {
        orow[1] = swar_add_wrap(orow[1], e.drow[1], (uint32_t)H);
        ocol[1] = swar_add_wrap(ocol[1], e.dcol[1], (uint32_t)W);
      }
      reward = e.reward;
      if (e.flags & 1u) over = 1;  // plot.py:183-184 (discount 0 on that frame)
    }
    if (!showtime) ret += reward;
    const int64_t slot = showtime ? 0 : t;
    paint_and_emit(out.obs + slot * out.obs_t_stride + env * LHW,
                   kBoard ? out.board + slot * out.board_t_stride + env * HW : nullptr,
                   !last_only || t == T - 1);
    if (!showtime) {
      // the frame's scalars wait in lane (t mod 64) of a register / bit of a scalar mask and
      // go out once per 64 frames, one store instruction per array
      const int slot_lane = t & (kWave - 1);
      reward_buf = lane == slot_lane ? (int)__float_as_uint(reward) : reward_buf;
      over_mask = slot_lane == 0 ? (uint64_t)over : over_mask | ((uint64_t)over << slot_lane);
      if (slot_lane == kWave - 1 || t == T - 1) {
        const int t0 = t - slot_lane;
        if (lane <= slot_lane) {
          const int64_t at = (int64_t)(t0 + lane) * B + env;
          const uint32_t ended = (uint32_t)(over_mask >> lane) & 1u;
          if (out.reward) out.reward[at] = __uint_as_float((uint32_t)reward_buf);
          if (out.discount) out.discount[at] = ended ? 0.0f : 1.0f;
          if (out.done) out.done[at] = (uint8_t)ended;
        }
      }
    }
  }

  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < CAMPX_SHAPE_MAX_THINGS; ++k)
      if (k < N) {
        st.pos[(int64_t)(2 * k) * B + env] = (int8_t)((orow[k >> 2] >> (8 * (k & 3))) & 0xffu);
        st.pos[(int64_t)(2 * k + 1) * B + env] = (int8_t)((ocol[k >> 2] >> (8 * (k & 3))) & 0xffu);
      }
    st.done[env] = (uint8_t)over;
    if (st.ret) st.ret[env] = ret;
  }
  if (backdrop_state) {
    if (quads)
      for (int i = lane; 4 * i < HW; i += kWave)
        reinterpret_cast<uint32_t*>(backdrop_state + env * HW)[i] = reinterpret_cast<const uint32_t*>(bd)[i];
    else
      for (int i = lane; i < HW; i += kWave) backdrop_state[env * HW + i] = (int8_t)bd[i];
  }
  report_bad_actions(out, lane == 0 ? bad : 0);
}


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

thread_local int32_t g_last_hip_error = 0;

// Tuning knobs for A/B measurements only (environment variables, read once; not
// part of the ABI).
bool knob_store_nt() {
  static const bool nt = [] {
    const char* v = getenv("CAMPX_STORE_NT");
    return !v || v[0] != '0';   // non-temporal observation stores by default
  }();
  return nt;
}
int knob_xcd() {
  static const int m = [] {
    const char* v = getenv("CAMPX_XCD_MODE");
    return v ? atoi(v) : 0;
  }();
  return m;
}
bool knob_no_split() {
  static const bool off = [] {
    const char* v = getenv("CAMPX_NO_SPLIT");
    return v && v[0] == '1';
  }();
  return off;
}
bool knob_no_step() {
  static const bool off = [] {
    const char* v = getenv("CAMPX_NO_STEP");
    return v && v[0] == '1';
  }();
  return off;
}
// Largest trace (bytes) one update + render pair of a rollout works on (launch_split).
int64_t knob_trace_chunk_bytes() {
  static const int64_t n = [] {
    const char* v = getenv("CAMPX_TRACE_CHUNK_MB");
    return (int64_t)(v && *v ? atoll(v) : 16) << 20;
  }();
  return n;
}
// ... and the largest trace a rollout may have and still run as one pair.
int64_t knob_trace_whole_bytes() {
  static const int64_t n = [] {
    const char* v = getenv("CAMPX_TRACE_WHOLE_MB");
    return (int64_t)(v && *v ? atoll(v) : 28) << 20;
  }();
  return n;
}
int knob_pair_mode() {
  static const int m = [] {
    const char* v = getenv("CAMPX_PAIR_MODE");
    return v ? atoi(v) : 3;
  }();
  return m;
}
bool knob_no_table() {
  static const bool off = [] {
    const char* v = getenv("CAMPX_NO_TABLE");
    return v && v[0] == '1';
  }();
  return off;
}

int32_t hip_failed(hipError_t e) {
  g_last_hip_error = (int32_t)e;
  return CAMPX_ELAUNCH;
}

size_t lds_bytes(const CampxSpec& s, bool board, int envs) {
  const int HW = s.rows * s.cols, LHW = s.n_layers * HW;
  size_t n = (size_t)((envs * LHW + 15) & ~15);
  if (board) n += (size_t)((envs * HW + 15) & ~15);
  n += (size_t)((LHW + 15) & ~15);
  n += CAMPX_MAX_CELLS * 2 + CAMPX_MAX_CELLS * sizeof(uint16_t) + CAMPX_MAX_LAYERS;
  n += (size_t)kChunk * kWave + CAMPX_MAX_CELLS;
  return (n + 15) & ~(size_t)15;
}

// Kernels that keep a 64-environment image in dynamic LDS need more than HIP's default
// 64 KiB for large rows (128 cells x 16 characters: 146 KiB of the CU's 160).
constexpr size_t kLdsPerWorkgroup = 160 * 1024;
template <typename Kernel>
hipError_t allow_lds(Kernel kernel, size_t dynamic_bytes) {
  if (dynamic_bytes <= 64 * 1024) return hipSuccess;
  // once per (device, kernel): what was granted is remembered
  static std::mutex lock;
  static std::map<std::pair<int, const void*>, size_t> granted;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const std::pair<int, const void*> key(dev, reinterpret_cast<const void*>(kernel));
  std::lock_guard<std::mutex> guard(lock);
  const auto it = granted.find(key);
  if (it != granted.end() && it->second >= dynamic_bytes) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(key.second, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)dynamic_bytes);
  if (e == hipSuccess) granted[key] = dynamic_bytes;
  return e;
}

RuleBlock make_rule_block(const CampxSpec& s) {
  RuleBlock rb;
  memset(&rb, 0, sizeof(rb));
  rb.rows = s.rows;
  rb.cols = s.cols;
  rb.n_layers = s.n_layers;
  rb.n_dyn = s.n_dyn;
  rb.n_rules = s.n_rules;
  rb.any_reward = s.any_reward;
  rb.perf_dyn = s.perf_dyn;
  rb.perf_n = s.perf_n;
  memcpy(rb.dyn_layer, s.dyn_layer, sizeof(rb.dyn_layer));
  memcpy(rb.dyn_z, s.dyn_z, sizeof(rb.dyn_z));
  memcpy(rb.dyn_row0, s.dyn_row0, sizeof(rb.dyn_row0));
  memcpy(rb.dyn_col0, s.dyn_col0, sizeof(rb.dyn_col0));
  memcpy(rb.rules, s.rules, sizeof(rb.rules));
  return rb;
}

template <int K>
int32_t launch_k(const CampxSpec& s, const CampxSpec* spec_dev, CampxState st,
                 const int8_t* actions, CampxOutputs out, int64_t B, int32_t T, int32_t reset_first,
                 int32_t emit_first, hipStream_t stream) {
  const bool board = out.board != nullptr;
  // 64 environments per wave; 32 / 16 (more waves in flight) measured -12 % / -25 %.
  constexpr int envs = kWave;
  const size_t shmem = lds_bytes(s, board, envs);
  const dim3 grid((unsigned)((B + envs - 1) / envs)), block(kWave);
  const RuleBlock rb = make_rule_block(s);
  // Streaming (write-through, non-temporal) stores pay when frames go to a trajectory
  // buffer that is not read back soon; a single frame buffer that every call
  // overwrites (Engine.play) is better left to the caches.
  const bool nt = knob_store_nt() && out.obs_t_stride != 0;
#define CAMPX_LAUNCH_E(BOARD, NT, ENVS)                                                  \
  (void)allow_lds(rollout_kernel<K, BOARD, NT, ENVS, false>, shmem);                         \
  hipLaunchKernelGGL((rollout_kernel<K, BOARD, NT, ENVS, false>), grid, block, shmem, stream, \
                     rb, spec_dev, st, actions, out, B, T, reset_first, emit_first, knob_xcd(), \
                     (int64_t)T * B)
#define CAMPX_LAUNCH(BOARD, NT) do { CAMPX_LAUNCH_E(BOARD, NT, 64); } while (0)
  if (board) {
    if (nt) CAMPX_LAUNCH(true, true); else CAMPX_LAUNCH(true, false);
  } else {
    if (nt) CAMPX_LAUNCH(false, true); else CAMPX_LAUNCH(false, false);
  }
#undef CAMPX_LAUNCH
#undef CAMPX_LAUNCH_E
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? CAMPX_OK : hip_failed(e);
}

int32_t launch_table(const CampxSpec& s, const CampxSpec* spec_dev, CampxState st,
                     const int8_t* actions, CampxOutputs out, int64_t B, int32_t T,
                     int32_t reset_first, int32_t emit_first, hipStream_t stream) {
  const bool board = out.board != nullptr;
  // 64 environments per wave; 32 / 16 (more waves in flight) measured -12 % / -25 %.
  constexpr int envs = kWave;
  // Streaming (write-through, non-temporal) stores pay when frames go to a trajectory
  // buffer that is not read back soon; a single frame buffer that every call
  // overwrites (Engine.play) is better left to the caches.
  const bool nt = knob_store_nt() && out.obs_t_stride != 0;
  const size_t shmem = table_lds_bytes(s, board, envs);
  const dim3 grid((unsigned)((B + envs - 1) / envs)), block(kWave);
  const MoverParams mp = {s.rows, s.cols, s.n_layers, s.dyn_layer[0], s.dyn_z[0],
                          s.dyn_row0[0], s.dyn_col0[0]};
#define CAMPX_LAUNCH_E(BOARD, NT, ENVS)                                                      \
  (void)allow_lds(rollout_table_kernel<BOARD, NT, ENVS>, shmem);                              \
  hipLaunchKernelGGL((rollout_table_kernel<BOARD, NT, ENVS>), grid, block, shmem, stream, mp, \
                     spec_dev, st, actions, out, B, T, reset_first, emit_first, knob_xcd())
#define CAMPX_LAUNCH(BOARD, NT) do { CAMPX_LAUNCH_E(BOARD, NT, 64); } while (0)
  if (board) {
    if (nt) CAMPX_LAUNCH(true, true); else CAMPX_LAUNCH(true, false);
  } else {
    if (nt) CAMPX_LAUNCH(false, true); else CAMPX_LAUNCH(false, false);
  }
#undef CAMPX_LAUNCH
#undef CAMPX_LAUNCH_E
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? CAMPX_OK : hip_failed(e);
}

int32_t launch_step_table(const CampxSpec& s, const CampxSpec* spec_dev, CampxState st,
                          const int8_t* actions, CampxOutputs out, int64_t B, int32_t reset_first,
                          hipStream_t stream) {
  const int HW = s.rows * s.cols, LHW = s.n_layers * HW;
  const bool board = out.board != nullptr;
  const size_t shmem = (size_t)((kWave * LHW + 15) & ~15) + (board ? (size_t)((kWave * HW + 15) & ~15) : 0);
  const dim3 grid((unsigned)((B + kWave - 1) / kWave)), block(kWave);
  const MoverParams mp = {s.rows, s.cols, s.n_layers, s.dyn_layer[0], s.dyn_z[0],
                          s.dyn_row0[0], s.dyn_col0[0]};
  if (board) {
    (void)allow_lds(step_table_kernel<true>, shmem);
    hipLaunchKernelGGL(step_table_kernel<true>, grid, block, shmem, stream, mp, spec_dev, st,
                       actions, out, B, reset_first);
  } else {
    (void)allow_lds(step_table_kernel<false>, shmem);
    hipLaunchKernelGGL(step_table_kernel<false>, grid, block, shmem, stream, mp, spec_dev, st,
                       actions, out, B, reset_first);
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? CAMPX_OK : hip_failed(e);
}

int32_t launch_step_pair(const CampxSpec& s, const CampxSpec* spec_dev, CampxState st,
                         const int8_t* actions, CampxOutputs out, int64_t B, int32_t reset_first,
                         hipStream_t stream) {
  const int HW = s.rows * s.cols, LHW = s.n_layers * HW;
  const bool board = out.board != nullptr;
  const size_t shmem = (size_t)((kWave * LHW + 15) & ~15) + (board ? (size_t)((kWave * HW + 15) & ~15) : 0);
  const dim3 grid((unsigned)((B + kWave - 1) / kWave)), block(kWave);
  const MoverParams mp = {s.rows, s.cols, s.n_layers, s.dyn_layer[0], s.dyn_z[0],
                          s.dyn_row0[0], s.dyn_col0[0]};
  if (board) {
    (void)allow_lds(step_pair_kernel<true>, shmem);
    hipLaunchKernelGGL(step_pair_kernel<true>, grid, block, shmem, stream, mp, s.dyn_layer[1],
                       s.dyn_row0[1], s.dyn_col0[1], spec_dev, st, actions, out, B, reset_first);
  } else {
    (void)allow_lds(step_pair_kernel<false>, shmem);
    hipLaunchKernelGGL(step_pair_kernel<false>, grid, block, shmem, stream, mp, s.dyn_layer[1],
                       s.dyn_row0[1], s.dyn_col0[1], spec_dev, st, actions, out, B, reset_first);
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? CAMPX_OK : hip_failed(e);
}

constexpr int kBigEnvs = 8 * kWave;   // environments of a "big" update workgroup

// Number of big workgroups from which launch_update prefers them: one per CU of the chip
// (CAMPX_BIG_WGS overrides; a huge value turns them off).
int64_t knob_big_workgroups() {
  static const int64_t n = [] {
    const char* v = getenv("CAMPX_BIG_WGS");
    if (v && *v) return (int64_t)atoll(v);
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    return (int64_t)cus;
  }();
  return n;
}

PairParams make_pair_params(const CampxSpec& s) {
  PairParams pp;
  memset(&pp, 0, sizeof(pp));
  pp.rows = s.rows;
  pp.cols = s.cols;
  pp.n_layers = s.n_layers;
  for (int d = 0; d < 2; ++d) {
    pp.dyn_layer[d] = s.dyn_layer[d];
    pp.row0[d] = s.dyn_row0[d];
    pp.col0[d] = s.dyn_col0[d];
  }
  return pp;
}

TupleParams make_tuple_params(const CampxSpec& s) {
  TupleParams tp;
  memset(&tp, 0, sizeof(tp));
  tp.rows = s.rows;
  tp.cols = s.cols;
  tp.n_dyn = s.n_dyn;
  tp.n_layers = s.n_layers;
  for (int d = 0; d < s.n_dyn; ++d) {
    tp.row0[d] = s.dyn_row0[d];
    tp.col0[d] = s.dyn_col0[d];
    tp.dyn_layer[d] = s.dyn_layer[d];
  }
  return tp;
}

int32_t launch_step_tuple(const CampxSpec& s, const CampxSpec* spec_dev, CampxState st,
                          const int8_t* actions, CampxOutputs out, int64_t B, int32_t reset_first,
                          hipStream_t stream) {
  const int HW = s.rows * s.cols, LHW = s.n_layers * HW;
  const bool board = out.board != nullptr;
  const size_t shmem = (size_t)((kWave * LHW + 15) & ~15) + (board ? (size_t)((kWave * HW + 15) & ~15) : 0);
  const dim3 grid((unsigned)((B + kWave - 1) / kWave)), block(kWave);
  const TupleParams tp = make_tuple_params(s);
#define CAMPX_STEP_TUPLE(KK, BOARD)                                                           \
  do {                                                                                        \
    (void)allow_lds(step_tuple_kernel<KK, BOARD>, shmem);                                     \
    hipLaunchKernelGGL((step_tuple_kernel<KK, BOARD>), grid, block, shmem, stream, tp, spec_dev, \
                       st, actions, out, B, reset_first);                                     \
  } while (0)
  if (s.n_dyn == 3) {
    if (board) CAMPX_STEP_TUPLE(3, true); else CAMPX_STEP_TUPLE(3, false);
  } else {
    if (board) CAMPX_STEP_TUPLE(4, true); else CAMPX_STEP_TUPLE(4, false);
  }
#undef CAMPX_STEP_TUPLE
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? CAMPX_OK : hip_failed(e);
}

// ---- split path: update pass -> trace, then one-shot render kernels
template <int K>
void launch_trace_k(const CampxSpec& s, const CampxSpec* spec_dev, CampxState st,
                    const int8_t* actions, CampxOutputs out, int64_t B, int32_t T,
                    int32_t reset_first, int64_t trace_plane, hipStream_t stream) {
  const size_t shmem = lds_bytes(s, false, 0);
  const dim3 grid((unsigned)((B + kWave - 1) / kWave)), block(kWave);
  const RuleBlock rb = make_rule_block(s);
  hipLaunchKernelGGL((rollout_kernel<K, false, false, kWave, true>), grid, block, shmem, stream,
                     rb, spec_dev, st, actions, out, B, T, reset_first, 0, 0, trace_plane);
}

// `trace` points at the first frame to render, `T` frames from there; `plane_rows` is the
// distance (in rows = environments) between two moving things' planes of the trace, i.e.
// B times the number of frames the trace holds.
int32_t launch_render(const CampxSpec& s, const CampxSpec* spec_dev, const uint8_t* trace,
                      int8_t* dst, int64_t B, int32_t T, int64_t plane_rows, bool is_board,
                      int fmt, hipStream_t stream) {
  const int HW = s.rows * s.cols;
  RenderParams rp;
  memset(&rp, 0, sizeof(rp));
  rp.R = (uint32_t)(is_board ? HW : s.n_layers * HW);
  // exact unsigned 32-bit division by R (Granlund & Montgomery 1994, fig. 4.1)
  uint32_t l = 0;
  while ((1ull << l) < rp.R) ++l;
  rp.m = (uint32_t)(((1ull << 32) * ((1ull << l) - rp.R)) / rp.R + 1);
  rp.sh1 = l < 1 ? l : 1;
  rp.sh2 = l > 0 ? l - 1 : 0;
  rp.slab_bytes = (uint32_t)(B * rp.R);
  rp.n_dyn = s.n_dyn;
  rp.is_board = is_board ? 1 : 0;
  rp.B = B;
  rp.cells = HW;
  for (int d = 0; d < s.n_dyn; ++d) {
    rp.dyn_char[d] = s.layer_char[s.dyn_layer[d]];
    rp.dyn_off[d] = s.dyn_layer[d] * HW;
  }
  // KiB of the int8 image per wave: what a wave WRITES is what counts (2 KiB: 164.8 us, 4 KiB:
  // 180.2 us for the boat race), so the 16-bit formats take half the window
  constexpr int kWin = CAMPX_RENDER_WIN, kWin16 = kWin > 1 ? kWin / 2 : 1;
  const bool sixteen = !is_board && fmt != 0;
  const uint32_t wspan = 1024u * (uint32_t)(sixteen ? kWin16 : kWin);   // one wave's windows
  const uint32_t span = wspan * kRenderWaves;                            // one block's
  // windows aligned in memory; for the 16-bit formats in units of image bytes = elements
  rp.shift_base = (uint32_t)((reinterpret_cast<uintptr_t>(dst) >> (sixteen ? 1 : 0)) & (wspan - 1u));
  rp.shift_slab = rp.slab_bytes & (wspan - 1u);
  const uint64_t reach = (uint64_t)rp.slab_bytes + ((rp.shift_base | rp.shift_slab) ? wspan - 1u : 0u);
  // rounded up to a multiple of 8 for the XCD remap; surplus blocks exit at once
  const dim3 grid((unsigned)((((reach + span - 1u) / span) + 7u) & ~(uint64_t)7), (unsigned)T);
  const int64_t n_rows = plane_rows;
  const bool nt = knob_store_nt();
  const bool odd = (rp.slab_bytes & (sixteen ? 7u : 15u)) != 0;   // frames are not whole chunks
#define CAMPX_RENDER4(KK, BOARD, NT, FMT, ODD)                                              \
  hipLaunchKernelGGL((render_kernel<KK, BOARD, NT, (FMT) ? kWin16 : kWin, FMT, ODD>), grid, \
                     dim3(kRenderWaves * kWave), 0, stream, rp, spec_dev, trace, dst, n_rows)
#define CAMPX_RENDER3(KK, BOARD, NT)                                      \
  do {                                                                    \
    if (!BOARD && fmt == 1 && odd) CAMPX_RENDER4(KK, false, NT, 1, true);   \
    else if (!BOARD && fmt == 1) CAMPX_RENDER4(KK, false, NT, 1, false);    \
    else if (!BOARD && fmt == 2 && odd) CAMPX_RENDER4(KK, false, NT, 2, true); \
    else if (!BOARD && fmt == 2) CAMPX_RENDER4(KK, false, NT, 2, false);    \
    else if (odd) CAMPX_RENDER4(KK, BOARD, NT, 0, true);                    \
    else CAMPX_RENDER4(KK, BOARD, NT, 0, false);                            \
  } while (0)
#define CAMPX_RENDER2(KK, BOARD)                                                    \
  do {                                                                              \
    if (nt) CAMPX_RENDER3(KK, BOARD, true); else CAMPX_RENDER3(KK, BOARD, false);   \
  } while (0)
#define CAMPX_RENDER1(KK)                                                   \
  do {                                                                      \
    if (is_board) CAMPX_RENDER2(KK, true); else CAMPX_RENDER2(KK, false);   \
  } while (0)
  switch (s.n_dyn) {
    case 1: CAMPX_RENDER1(1); break;
    case 2: CAMPX_RENDER1(2); break;
    case 3: CAMPX_RENDER1(3); break;
    default: CAMPX_RENDER1(4); break;
  }
#undef CAMPX_RENDER1
#undef CAMPX_RENDER2
#undef CAMPX_RENDER3
#undef CAMPX_RENDER4
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? CAMPX_OK : hip_failed(e);
}

// Can this call take the two-kernel path?  Frames must be stored back to back and be
// whole 16-byte chunks, and a chunk may span at most two rows.
// Strides of 0: every frame overwrites the first slot, so only the last survives - the
// two-kernel path then renders just that one from the last row of the trace.
bool last_frame_only(const CampxOutputs& out) {
  return out.obs_t_stride == 0 && (!out.board || out.board_t_stride == 0) &&
         out.obs_format == CAMPX_OBS_INT8;
}

bool split_ok(const CampxSpec& s, const CampxOutputs& out, int64_t B, int32_t T) {
  const int64_t HW = (int64_t)s.rows * s.cols, LHW = HW * s.n_layers;
  if (!out.trace || !s.render_valid || T <= 0 || knob_no_split()) return false;
  if (LHW < 16 || B * LHW >= (1ll << 32) - 65536) return false;
  if (out.board && HW < 16) return false;
  // every frame kept, back to back - or only the last one (strides 0)
  const bool every = out.obs_t_stride == B * LHW && (!out.board || out.board_t_stride == B * HW);
  return every || last_frame_only(out);
}

// Shape of the update kernels' workgroups: producer and consumer waves (A/B builds can
// override).  Measured, whole rollout launch at the BASELINE sizes (gpurun_out/r2a):
// one-mover table kernel, boat race: (1,1) 0.2006, (1,3) 0.2112, (2,2) 0.2034,
// (2,4) 0.2005, (4,4) 0.1946 ms - the 256-environment workgroup writes 1 KiB / 256 B row
// pieces instead of 512 / 128 B.  With the final kernels (gpurun_out/r2y, kernel time in
// us): table kernel, boat race / wall world: (4,4) 16.4 / 73, (2,2) 17.1 / 66.5,
// (4,2) 21.3 / 63; pair kernel, sokoban: (4,4) 44-50, (4,2) 42, (2,2) 62, (2,1) 57.
#ifndef CAMPX_UPD_PROD
#define CAMPX_UPD_PROD 4
#endif
#ifndef CAMPX_UPD_CONS
#define CAMPX_UPD_CONS 4
#endif
#ifndef CAMPX_UPD_GROUP
// frames per group of the 256-environment one-mover workgroups; kernel us per 100 frames at
// B = 4 096 / 65 536 (gpurun_out/t20-t21): 4: 18.1 / 21.4, 8: 14.8 / 17.8, 16: 12.4 / 15.6,
// 32: 12.7 / 16.8 - a group costs ~0.3 us of hand-over, a longer one more fill and drain
#define CAMPX_UPD_GROUP 16
#endif
#ifndef CAMPX_PAIR_PROD
#define CAMPX_PAIR_PROD 4
#endif
#ifndef CAMPX_PAIR_CONS
#define CAMPX_PAIR_CONS 2
#endif
#ifndef CAMPX_TUPLE_PROD
#define CAMPX_TUPLE_PROD 4
#endif
#ifndef CAMPX_TUPLE_CONS
#define CAMPX_TUPLE_CONS 2
#endif

// `trace_plane`: rows (environments) from one moving thing's plane of the trace to the
// next's - B times the frames the whole trace holds, which is more than T when the caller
// runs a launch in chunks.
int32_t launch_update(const CampxSpec& s, const CampxSpec* spec_dev, CampxState st,
                     const int8_t* actions, CampxOutputs out, int64_t B, int32_t T,
                     int32_t reset_first, bool use_table, int64_t trace_plane,
                     hipStream_t stream) {
  // 512-environment workgroups (twice the row piece per store) once there are enough
  // environments to give every CU one; 256-environment workgroups below that
  const bool big = B >= (int64_t)kBigEnvs * knob_big_workgroups();
  if (use_table) {
    const MoverParams mp = {s.rows, s.cols, s.n_layers, s.dyn_layer[0], s.dyn_z[0],
                            s.dyn_row0[0], s.dyn_col0[0]};
    if (big) {
      constexpr int kProd = 8, kCons = 4;
      const dim3 grid((unsigned)((B + kBigEnvs - 1) / kBigEnvs)),
          block((kProd + kCons + update_loaders(kProd)) * kWave);
      hipLaunchKernelGGL((update_table_kernel<kProd, kCons, 8>), grid, block, 0, stream, mp,
                         spec_dev, st, actions, out, B, T, reset_first);
    } else {
      constexpr int kProd = CAMPX_UPD_PROD, kCons = CAMPX_UPD_CONS, kEnvs = kProd * kWave;
      const dim3 grid((unsigned)((B + kEnvs - 1) / kEnvs)),
          block((kProd + kCons + update_loaders(kProd)) * kWave);
      hipLaunchKernelGGL((update_table_kernel<kProd, kCons, CAMPX_UPD_GROUP>), grid, block, 0, stream, mp,
                         spec_dev, st, actions, out, B, T, reset_first);
    }
  } else if (s.n_dyn == 2 && st.pair_table && !knob_no_table()) {
    const PairParams pp = make_pair_params(s);
    const int n_entries = s.rows * s.cols * s.rows * s.cols * CAMPX_N_ACTIONS;
    // the entries in LDS when they fit (CAMPX_PAIR_MODE=0: read them through L1/L2 anyway)
    const bool in_lds = n_entries <= kPairLdsEntries && knob_pair_mode() != 0;
    const size_t shmem = in_lds ? (((size_t)n_entries * sizeof(uint32_t) + 15) & ~(size_t)15) : 0;
#define CAMPX_PAIR_LAUNCH(PROD, CONS)                                                          \
  do {                                                                                         \
    const dim3 grid((unsigned)((B + (PROD) * kWave - 1) / ((PROD) * kWave))),                  \
        block(((PROD) + (CONS) + update_loaders(PROD)) * kWave);                               \
    if (in_lds)                                                                                \
      hipLaunchKernelGGL((update_pair_kernel<true, PROD, CONS>), grid, block, shmem, stream,   \
                         pp, spec_dev, st, actions, out, B, T, reset_first, trace_plane);      \
    else                                                                                       \
      hipLaunchKernelGGL((update_pair_kernel<false, PROD, CONS>), grid, block, 0, stream, pp,  \
                         spec_dev, st, actions, out, B, T, reset_first, trace_plane);          \
  } while (0)
    if (big)
      CAMPX_PAIR_LAUNCH(8, 4);
    else
      CAMPX_PAIR_LAUNCH(CAMPX_PAIR_PROD, CAMPX_PAIR_CONS);
#undef CAMPX_PAIR_LAUNCH
  } else if (s.n_dyn >= 3 && st.pair_table && !knob_no_table()) {
    constexpr int kProd = CAMPX_TUPLE_PROD, kCons = CAMPX_TUPLE_CONS, kEnvs = kProd * kWave;
    const dim3 grid((unsigned)((B + kEnvs - 1) / kEnvs)),
        block((kProd + kCons + update_loaders(kProd)) * kWave);
    const TupleParams tp = make_tuple_params(s);
    if (s.n_dyn == 3)
      hipLaunchKernelGGL((update_tuple_kernel<3, kProd, kCons>), grid, block, 0, stream, tp, st,
                         actions, out, B, T, reset_first, trace_plane);
    else
      hipLaunchKernelGGL((update_tuple_kernel<4, kProd, kCons>), grid, block, 0, stream, tp, st,
                         actions, out, B, T, reset_first, trace_plane);
  } else {
    switch (s.n_dyn) {
      case 1: launch_trace_k<1>(s, spec_dev, st, actions, out, B, T, reset_first, trace_plane, stream); break;
      case 2: launch_trace_k<2>(s, spec_dev, st, actions, out, B, T, reset_first, trace_plane, stream); break;
      case 3: launch_trace_k<3>(s, spec_dev, st, actions, out, B, T, reset_first, trace_plane, stream); break;
      default: launch_trace_k<4>(s, spec_dev, st, actions, out, B, T, reset_first, trace_plane, stream); break;
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_failed(e);
  return CAMPX_OK;
}

int32_t launch_renders(const CampxSpec& s, const CampxSpec* spec_dev, CampxOutputs out, int64_t B,
                       int32_t T, int64_t plane_rows, hipStream_t stream) {
  const uint8_t* first = out.trace;
  if (last_frame_only(out)) {
    first += (int64_t)(T - 1) * B;
    T = 1;
  }
  int32_t rc = launch_render(s, spec_dev, first, out.obs, B, T, plane_rows, false, out.obs_format, stream);
  if (rc != CAMPX_OK) return rc;
  if (out.board) rc = launch_render(s, spec_dev, first, out.board, B, T, plane_rows, true, 0, stream);
  return rc;
}

int32_t launch_split(const CampxSpec& s, const CampxSpec* spec_dev, CampxState st,
                     const int8_t* actions, CampxOutputs out, int64_t B, int32_t T,
                     int32_t reset_first, bool use_table, hipStream_t stream) {
  const int64_t plane = (int64_t)T * B;
  // The render kernel runs at the write ceiling only while the trace it reads stays cached
  // (boat race, B = 65 536: 6.96 TB/s with a 26 MB trace at T = 400, 5.35 TB/s with 65 MB at
  // T = 1 000; the same at B = 524 288, T = 100): run long launches as chunks of frames,
  // update pass and render alternating, each chunk's trace plane at most 16 MB (CAMPX_TRACE_CHUNK_MB).
  // (us per launch, render kernels only, no chunks / 28 / 16 / 8 MB: T = 1 000: 2 265 / 1 820 /
  // 1 641 / 1 644; B = 524 288: 1 739 / 1 504 / 1 314 / 1 316 - gpurun_out/t16.  A 26 MB trace
  // in one piece is still at full speed, so launches up to 28 MB (CAMPX_TRACE_WHOLE_MB) are not cut.)
  // (per moving thing's plane of the trace: sokoban with three boxes, four planes of 13 MB,
  // renders at full speed in one piece, and 4 % slower cut in four)
  const int64_t per_frame = B;
  int64_t chunk = knob_trace_chunk_bytes() / per_frame;
  chunk = chunk < 16 ? 16 : chunk & ~(int64_t)15;
  chunk = chunk > 65520 ? 65520 : chunk;   // a render launch has one grid row per frame
  const bool whole = (per_frame * T <= knob_trace_whole_bytes() && T <= 65535) || T <= chunk;
  if (last_frame_only(out) || whole) {
    const int32_t rc = launch_update(s, spec_dev, st, actions, out, B, T, reset_first, use_table,
                                     plane, stream);
    if (rc != CAMPX_OK) return rc;
    return launch_renders(s, spec_dev, out, B, T, plane, stream);
  }
  const int64_t elem = out.obs_format == CAMPX_OBS_INT8 ? 1 : 2;
  for (int64_t t0 = 0; t0 < T; t0 += chunk) {
    const int32_t n = (int32_t)(T - t0 < chunk ? T - t0 : chunk);
    CampxOutputs part = out;
    part.obs = out.obs + t0 * out.obs_t_stride * elem;
    if (out.board) part.board = out.board + t0 * out.board_t_stride;
    if (out.reward) part.reward = out.reward + t0 * B;
    if (out.discount) part.discount = out.discount + t0 * B;
    if (out.done) part.done = out.done + t0 * B;
    if (out.perf) part.perf = out.perf + t0 * B;
    part.trace = out.trace + t0 * B;
    int32_t rc = launch_update(s, spec_dev, st, actions + t0 * B, part, B, n,
                               t0 == 0 ? reset_first : 0, use_table, plane, stream);
    if (rc != CAMPX_OK) return rc;
    rc = launch_renders(s, spec_dev, part, B, n, plane, stream);
    if (rc != CAMPX_OK) return rc;
  }
  return CAMPX_OK;
}

int32_t launch(const CampxSpec* spec_host, const CampxSpec* spec_dev, CampxState st,
               const int8_t* actions, CampxOutputs out, int64_t B, int32_t T, int32_t reset_first,
               int32_t emit_first, void* stream, bool interpreter_only = false) {
  if (!spec_host || !spec_dev || !st.pos || !st.done || !out.obs || B <= 0 || T < 0)
    return CAMPX_EINVAL;
  if (T > 0 && !actions) return CAMPX_EINVAL;
  if (out.perf && spec_host->perf_dyn < 0) return CAMPX_EINVAL;
  if (reinterpret_cast<uintptr_t>(out.obs) & 15) return CAMPX_EINVAL;
  if (B > (int64_t)0x7fffffff * 16) return CAMPX_EINVAL;
  const int32_t v = campx_spec_validate(spec_host);
  if (v != CAMPX_OK) return v;
  if (lds_bytes(*spec_host, out.board != nullptr, kWave) + 8 * 1024 > kLdsPerWorkgroup) return CAMPX_ESPEC;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool use_table =
      spec_host->table_valid && spec_host->n_dyn == 1 && !interpreter_only && !knob_no_table();
  if (out.obs_format < CAMPX_OBS_INT8 || out.obs_format > CAMPX_OBS_BF16) return CAMPX_EINVAL;
  if (!emit_first && !interpreter_only && split_ok(*spec_host, out, B, T))
    return launch_split(*spec_host, spec_dev, st, actions, out, B, T, reset_first, use_table, s);
  // (16-bit observations: the render kernel above, or the one-frame kernels below)
  if (use_table && T == 1 && !emit_first && spec_host->render_valid && !knob_no_step())
    return launch_step_table(*spec_host, spec_dev, st, actions, out, B, reset_first, s);
  if (T == 1 && !emit_first && spec_host->n_dyn == 2 && st.pair_table && spec_host->render_valid &&
      !interpreter_only && !knob_no_table() && !knob_no_step())
    return launch_step_pair(*spec_host, spec_dev, st, actions, out, B, reset_first, s);
  if (T == 1 && !emit_first && spec_host->n_dyn >= 3 && st.pair_table && spec_host->render_valid &&
      !interpreter_only && !knob_no_table() && !knob_no_step())
    return launch_step_tuple(*spec_host, spec_dev, st, actions, out, B, reset_first, s);
  if (out.obs_format != CAMPX_OBS_INT8) return CAMPX_EINVAL;
  if (use_table)
    return launch_table(*spec_host, spec_dev, st, actions, out, B, T, reset_first, emit_first, s);
  switch (spec_host->n_dyn) {
    case 1:
      return launch_k<1>(*spec_host, spec_dev, st, actions, out, B, T, reset_first, emit_first, s);
    case 2:
      return launch_k<2>(*spec_host, spec_dev, st, actions, out, B, T, reset_first, emit_first, s);
    case 3:
      return launch_k<3>(*spec_host, spec_dev, st, actions, out, B, T, reset_first, emit_first, s);
    case 4:
      return launch_k<4>(*spec_host, spec_dev, st, actions, out, B, T, reset_first, emit_first, s);
    default:
      return CAMPX_ESPEC;
  }
}

// The update pass of every action (engine.py:200-204: things in update-schedule order,
// rewards summed as r + total, plot.py:208-211) and the things' paint parameters, packed
// for shape_rollout_kernel.  `s` has passed campx_shape_spec_validate.
ShapeParams make_shape_params(const CampxShapeSpec& s) {
  ShapeParams sp;
  memset(&sp, 0, sizeof(sp));
  sp.rows = s.rows;
  sp.cols = s.cols;
  sp.n_layers = s.n_layers;
  sp.n_things = s.n_things;
  sp.first_drape = s.first_drape;
  const CampxShapeThing& last = s.things[s.n_things - 1];
  sp.n_list = last.cell_begin + last.n_cells;
  for (int k = 0; k < s.n_things; ++k) {
    const CampxShapeThing& th = s.things[k];
    sp.thing[k] = (th.n_cells ? (uint32_t)th.cell_begin : 0u) | ((uint32_t)th.n_cells << 11) |
                  ((uint32_t)th.layer << 23) | ((th.visible ? 1u : 0u) << 28);
  }
  for (int a = 0; a < CAMPX_N_ACTIONS; ++a) {
    ShapeAction& e = sp.act[a];
    bool first = true;
    for (int u = 0; u < s.n_things; ++u) {
      const int k = s.update_order[u];
      const CampxShapeThing& th = s.things[k];
      if ((th.terminate_mask >> a) & 1) e.flags |= 1u;  // plot.py:183-184
      e.drow[k >> 2] |= (uint32_t)(uint8_t)th.drow[a] << (8 * (k & 3));
      e.dcol[k >> 2] |= (uint32_t)(uint8_t)th.dcol[a] << (8 * (k & 3));
      if ((th.has_reward_mask >> a) & 1) {
        e.reward = first ? th.reward[a] : th.reward[a] + e.reward;
        first = false;
      }
    }
    if (first)
      e.reward = __builtin_nanf("");  // nobody called add_reward: None
    else
      e.flags |= 2u;
  }
  memcpy(sp.layer_char, s.layer_char, CAMPX_MAX_LAYERS);
  return sp;
}

}  // namespace

extern "C" {

int32_t campx_spec_size(void) { return (int32_t)sizeof(CampxSpec); }

int32_t campx_spec_validate(const CampxSpec* s) {
  if (!s) return CAMPX_EINVAL;
  if (s->magic != CAMPX_SPEC_MAGIC || s->version != CAMPX_SPEC_VERSION) return CAMPX_ESPEC;
  if (s->rows < 1 || s->cols < 1 || s->rows > 127 || s->cols > 127) return CAMPX_ESPEC;
  const int HW = s->rows * s->cols;
  if (HW > CAMPX_MAX_CELLS) return CAMPX_ESPEC;
  if (s->n_layers < 1 || s->n_layers > CAMPX_MAX_LAYERS) return CAMPX_ESPEC;
  if (s->n_dyn < 1 || s->n_dyn > CAMPX_MAX_DYN) return CAMPX_ESPEC;
  if (s->n_static < 0 || s->n_static > CAMPX_MAX_STATIC) return CAMPX_ESPEC;
  if (s->n_rules < 0 || s->n_rules > CAMPX_MAX_RULES) return CAMPX_ESPEC;
  for (int d = 0; d < s->n_dyn; ++d) {
    if (s->dyn_layer[d] < 0 || s->dyn_layer[d] >= s->n_layers) return CAMPX_ESPEC;
    if (s->dyn_z[d] < 1 || s->dyn_z[d] > 255) return CAMPX_ESPEC;
    if (s->dyn_row0[d] < 0 || s->dyn_row0[d] >= s->rows) return CAMPX_ESPEC;
    if (s->dyn_col0[d] < 0 || s->dyn_col0[d] >= s->cols) return CAMPX_ESPEC;
  }
  for (int i = 0; i < HW; ++i) {
    if (s->static_top_layer[i] >= s->n_layers) return CAMPX_ESPEC;
    if (s->n_static < 16 && (s->static_cover[i] >> s->n_static)) return CAMPX_ESPEC;
  }
  for (int i = 0; i < s->n_layers * HW; ++i)
    if (s->obs_template[i] != 0 && s->obs_template[i] != 1) return CAMPX_ESPEC;
  for (int i = 0; i < s->n_rules; ++i) {
    const CampxRule& r = s->rules[i];
    if (r.dyn < 0 || r.dyn >= s->n_dyn) return CAMPX_ESPEC;
    switch (r.op) {
      case CAMPX_OP_AGENT:
        break;
      case CAMPX_OP_DIR_HOVER:
        if (r.aux < 0 || r.aux >= s->n_layers) return CAMPX_ESPEC;
        break;
      case CAMPX_OP_BOX:
        if (r.aux < 0 || r.aux >= s->n_dyn) return CAMPX_ESPEC;
        break;
      case CAMPX_OP_GOAL:
        if (r.aux < 0 || r.aux >= s->n_static) return CAMPX_ESPEC;
        break;
      default:
        return CAMPX_ESPEC;
    }
  }
  if (s->n_rules > 0 && !s->rules[s->n_rules - 1].end_group) return CAMPX_ESPEC;
  if (s->perf_dyn < -1 || s->perf_dyn >= s->n_dyn) return CAMPX_ESPEC;
  if (s->perf_dyn >= 0) {
    if (s->perf_n < 2 || s->perf_n > 255) return CAMPX_ESPEC;
    for (int i = 0; i < HW; ++i)
      if (s->cell_class[i] > s->perf_n) return CAMPX_ESPEC;
  }
  return CAMPX_OK;
}

int32_t campx_spec_compile(CampxSpec* spec, void* stream) {
  const int32_t v = campx_spec_validate(spec);
  if (v != CAMPX_OK) return v;
  spec->table_valid = 0;
  {
    const int HW = spec->rows * spec->cols, LHW = spec->n_layers * HW;
    const int pitch_obs = ((LHW + 15) & ~15) + 16, pitch_board = ((HW + 15) & ~15) + 16;
    for (int r = 0; r < 16; ++r) {
      for (int j = 0; j < pitch_obs; ++j)
        spec->rot_obs[r * pitch_obs + j] = spec->obs_template[(j + r) % LHW];
      for (int j = 0; j < pitch_board; ++j)
        spec->rot_board[r * pitch_board + j] =
            (int8_t)spec->layer_char[spec->static_top_layer[(j + r) % HW]];
    }
    spec->render_valid = 1;
  }
  if (spec->n_dyn != 1) return CAMPX_OK;
  const int W = spec->cols, HW = spec->rows * spec->cols;
  const int n = HW * CAMPX_N_ACTIONS;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // One scratch allocation: spec | trace | reward | pos | done | actions | done_out
  const size_t off_obs = (sizeof(CampxSpec) + 255) & ~(size_t)255;
  const size_t off_reward = (off_obs + (size_t)n + 255) & ~(size_t)255;   // (off_obs: the trace)
  const size_t off_pos = off_reward + sizeof(float) * n;
  const size_t off_done = off_pos + 2 * (size_t)n;
  const size_t off_act = off_done + n;
  const size_t off_dout = off_act + n;
  const size_t off_perf = off_dout + n;
  const size_t total = off_perf + n;
  char* dev = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&dev), total);
  if (e != hipSuccess) return hip_failed(e);
  // host images of pos / actions: pseudo-environment i = (cell i/5, action i%5)
  int8_t* host = static_cast<int8_t*>(malloc(8 * (size_t)n + sizeof(float) * n));
  if (!host) {
    (void)hipFree(dev);
    return CAMPX_ENOMEM;
  }
  int8_t* h_pos = host;
  int8_t* h_act = host + 2 * n;
  uint8_t* h_done = reinterpret_cast<uint8_t*>(host + 3 * n);
  int8_t* h_perf = host + 4 * n;
  float* h_reward = reinterpret_cast<float*>(host + 8 * n);
  for (int i = 0; i < n; ++i) {
    const int cell = i / CAMPX_N_ACTIONS;
    h_pos[i] = (int8_t)(cell / W);
    h_pos[n + i] = (int8_t)(cell % W);
    h_act[i] = (int8_t)(i % CAMPX_N_ACTIONS);
  }
  int32_t rc = CAMPX_OK;
#define CAMPX_TRY(call)           \
  do {                            \
    e = (call);                   \
    if (e != hipSuccess) {        \
      rc = hip_failed(e);         \
      goto done;                  \
    }                             \
  } while (0)
  CAMPX_TRY(hipMemcpyAsync(dev, spec, sizeof(CampxSpec), hipMemcpyHostToDevice, s));
  CAMPX_TRY(hipMemcpyAsync(dev + off_pos, h_pos, 2 * (size_t)n, hipMemcpyHostToDevice, s));
  CAMPX_TRY(hipMemcpyAsync(dev + off_act, h_act, (size_t)n, hipMemcpyHostToDevice, s));
  CAMPX_TRY(hipMemsetAsync(dev + off_done, 0, (size_t)n, s));
  {
    CampxState st = {reinterpret_cast<int8_t*>(dev + off_pos),
                     reinterpret_cast<uint8_t*>(dev + off_done), nullptr, nullptr};
    CampxOutputs out;
    memset(&out, 0, sizeof(out));
    out.reward = reinterpret_cast<float*>(dev + off_reward);
    out.done = reinterpret_cast<uint8_t*>(dev + off_dout);
    out.perf = spec->perf_dyn >= 0 ? reinterpret_cast<int8_t*>(dev + off_perf) : nullptr;
    out.trace = reinterpret_cast<uint8_t*>(dev + off_obs);   // [1, 1, n], not read back
    // the interpreter in trace mode (no observation image): one frame of every (cell, action)
    launch_trace_k<1>(*spec, reinterpret_cast<const CampxSpec*>(dev), st,
                      reinterpret_cast<const int8_t*>(dev + off_act), out, n, 1, 0, (int64_t)n, s);
    CAMPX_TRY(hipGetLastError());
  }
  CAMPX_TRY(hipMemcpyAsync(h_pos, dev + off_pos, 2 * (size_t)n, hipMemcpyDeviceToHost, s));
  CAMPX_TRY(hipMemcpyAsync(h_done, dev + off_dout, (size_t)n, hipMemcpyDeviceToHost, s));
  CAMPX_TRY(hipMemcpyAsync(h_perf, dev + off_perf, (size_t)n, hipMemcpyDeviceToHost, s));
  CAMPX_TRY(hipMemcpyAsync(h_reward, dev + off_reward, sizeof(float) * n, hipMemcpyDeviceToHost, s));
  CAMPX_TRY(hipStreamSynchronize(s));
#undef CAMPX_TRY
  for (int i = 0; i < n; ++i) {
    CampxTransition& tr = spec->table[i];
    tr.reward = h_reward[i];
    tr.next_cell = (uint8_t)((int)h_pos[i] * W + (int)h_pos[n + i]);
    tr.done = h_done[i];
    tr.perf = spec->perf_dyn >= 0 ? h_perf[i] : (int8_t)0;
    tr.paint = (uint8_t)(spec->static_top_layer[tr.next_cell] |
                         (spec->static_top_z[tr.next_cell] > spec->dyn_z[0] ? 0x80u : 0u));
  }
  spec->table_valid = 1;
done:
  free(host);
  (void)hipFree(dev);
  return rc;
}

int64_t campx_pair_table_bytes(const CampxSpec* spec) {
  if (!spec || campx_spec_validate(spec) != CAMPX_OK || spec->n_dyn < 2) return 0;
  const int64_t HW = (int64_t)spec->rows * spec->cols;
  if (HW > 128) return 0;  // cells are 7-bit fields
  int64_t n = CAMPX_N_ACTIONS;
  for (int d = 0; d < spec->n_dyn; ++d) n *= HW;
  if (spec->n_dyn == 2) {
    const int64_t bytes = 256 * (int64_t)sizeof(float) + n * (int64_t)sizeof(uint32_t);
    return bytes <= (1 << 20) ? bytes : 0;
  }
  // three / four movers: 64-bit entries, read from global memory
  const int64_t bytes = 256 * (int64_t)sizeof(float) + n * (int64_t)sizeof(uint64_t);
  return bytes <= kTupleTableMaxBytes ? bytes : 0;
}

int32_t campx_pair_table_build(const CampxSpec* spec, const CampxSpec* spec_dev, void* table_dev,
                               void* stream) {
  const int64_t bytes = campx_pair_table_bytes(spec);
  if (bytes == 0 || !spec_dev || !table_dev) return CAMPX_EINVAL;
  const int K = spec->n_dyn, W = spec->cols, HW = spec->rows * spec->cols;
  size_t n = CAMPX_N_ACTIONS;
  for (int d = 0; d < K; ++d) n *= (size_t)HW;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // device scratch: reward[n] | trace[K][n] | pos[2K][n] | done[n] | actions[n] | done_out[n] | perf[n]
  // (the interpreter in trace mode writes no observations)
  const size_t off_trace = sizeof(float) * n;
  const size_t off_pos = off_trace + (size_t)K * n;
  const size_t off_done = off_pos + 2 * (size_t)K * n;
  const size_t off_act = off_done + n;
  const size_t off_dout = off_act + n;
  const size_t off_perf = off_dout + n;
  const size_t total = off_perf + n;
  char* dev = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&dev), total);
  if (e != hipSuccess) return hip_failed(e);
  // host scratch, widest arrays first so that nothing needs an alignment pad:
  // table (256 floats + entries) | reward[n] | pos[2K][n] | act[n] | done[n] | perf[n] | trace[K][n]
  const size_t table_bytes = ((size_t)bytes + 7) & ~(size_t)7;
  const size_t host_bytes = table_bytes + n * 4 + n * (size_t)(2 * K + 1 + 1 + 1 + K);
  char* host = static_cast<char*>(malloc(host_bytes));
  if (!host) {
    (void)hipFree(dev);
    return CAMPX_ENOMEM;
  }
  float* h_table = reinterpret_cast<float*>(host);
  uint32_t* h_entries32 = reinterpret_cast<uint32_t*>(h_table + 256);
  uint64_t* h_entries64 = reinterpret_cast<uint64_t*>(h_table + 256);
  float* h_reward = reinterpret_cast<float*>(host + table_bytes);
  int8_t* h_pos = reinterpret_cast<int8_t*>(h_reward + n);
  int8_t* h_act = h_pos + 2 * (size_t)K * n;
  uint8_t* h_done = reinterpret_cast<uint8_t*>(h_act + n);
  int8_t* h_perf = reinterpret_cast<int8_t*>(h_done + n);
  uint8_t* h_trace = reinterpret_cast<uint8_t*>(h_perf + n);
  for (size_t i = 0; i < n; ++i) {  // index = ((cell_0 * HW + cell_1) * HW + ...) * 5 + action
    size_t rest = i / CAMPX_N_ACTIONS;
    h_act[i] = (int8_t)(i % CAMPX_N_ACTIONS);
    for (int d = K - 1; d >= 0; --d) {
      const int cell = (int)(rest % (size_t)HW);
      rest /= (size_t)HW;
      h_pos[(size_t)(2 * d) * n + i] = (int8_t)(cell / W);
      h_pos[(size_t)(2 * d + 1) * n + i] = (int8_t)(cell % W);
    }
  }
  int32_t rc = CAMPX_OK;
  int n_rewards = 0;
  uint32_t reward_bits[256];
#define CAMPX_TRY(call)           \
  do {                            \
    e = (call);                   \
    if (e != hipSuccess) {        \
      rc = hip_failed(e);         \
      goto done;                  \
    }                             \
  } while (0)
  CAMPX_TRY(hipMemcpyAsync(dev + off_pos, h_pos, 2 * (size_t)K * n, hipMemcpyHostToDevice, s));
  CAMPX_TRY(hipMemcpyAsync(dev + off_act, h_act, n, hipMemcpyHostToDevice, s));
  CAMPX_TRY(hipMemsetAsync(dev + off_done, 0, n, s));
  CAMPX_TRY(hipMemsetAsync(dev + off_perf, 0, n, s));
  {
    CampxState st = {reinterpret_cast<int8_t*>(dev + off_pos),
                     reinterpret_cast<uint8_t*>(dev + off_done), nullptr, nullptr};
    CampxOutputs out;
    memset(&out, 0, sizeof(out));
    out.reward = reinterpret_cast<float*>(dev);
    out.done = reinterpret_cast<uint8_t*>(dev + off_dout);
    out.perf = spec->perf_dyn >= 0 ? reinterpret_cast<int8_t*>(dev + off_perf) : nullptr;
    out.trace = reinterpret_cast<uint8_t*>(dev + off_trace);
    // the interpreter in trace mode: positions, visibility, reward, done, perf
    const int8_t* acts = reinterpret_cast<const int8_t*>(dev + off_act);
    switch (K) {
      case 2: launch_trace_k<2>(*spec, spec_dev, st, acts, out, (int64_t)n, 1, 0, (int64_t)n, s); break;
      case 3: launch_trace_k<3>(*spec, spec_dev, st, acts, out, (int64_t)n, 1, 0, (int64_t)n, s); break;
      default: launch_trace_k<4>(*spec, spec_dev, st, acts, out, (int64_t)n, 1, 0, (int64_t)n, s); break;
    }
    CAMPX_TRY(hipGetLastError());
  }
  CAMPX_TRY(hipMemcpyAsync(h_trace, dev + off_trace, (size_t)K * n, hipMemcpyDeviceToHost, s));
  CAMPX_TRY(hipMemcpyAsync(h_reward, dev, sizeof(float) * n, hipMemcpyDeviceToHost, s));
  CAMPX_TRY(hipMemcpyAsync(h_done, dev + off_dout, n, hipMemcpyDeviceToHost, s));
  CAMPX_TRY(hipMemcpyAsync(h_perf, dev + off_perf, n, hipMemcpyDeviceToHost, s));
  CAMPX_TRY(hipStreamSynchronize(s));
  for (int i = 0; i < 256; ++i) h_table[i] = 0.0f;
  for (size_t i = 0; i < n; ++i) {
    uint32_t bits;
    memcpy(&bits, &h_reward[i], 4);
    int idx = -1;
    for (int k = 0; k < n_rewards; ++k) {
      if (reward_bits[k] == bits) {
        idx = k;
        break;
      }
    }
    if (idx < 0) {
      if (n_rewards == 256) {
        rc = CAMPX_ESPEC;
        goto done;
      }
      idx = n_rewards++;
      reward_bits[idx] = bits;
      h_table[idx] = h_reward[i];
    }
    const uint32_t perf = (uint32_t)((spec->perf_dyn >= 0 ? h_perf[i] : 0) + 1);
    const uint32_t over = (uint32_t)(h_done[i] & 1);
    if (K == 2) {
      const uint32_t ta = h_trace[i], tb = h_trace[n + i];
      h_entries32[i] = (ta & 0x7fu) | ((tb & 0x7fu) << 7) | ((ta >> 7) << 14) | ((tb >> 7) << 15) |
                       (over << 16) | (perf << 17) | ((uint32_t)idx << 19);
    } else {
      uint32_t lo = 0;
      for (int d = 0; d < K; ++d) {
        const uint32_t tr = h_trace[(size_t)d * n + i];
        lo |= ((tr & 0x7fu) << (7 * d)) | ((tr >> 7) << (28 + d));
      }
      h_entries64[i] = (uint64_t)lo | ((uint64_t)(over | (perf << 1) | ((uint32_t)idx << 3)) << 32);
    }
  }
  CAMPX_TRY(hipMemcpyAsync(table_dev, h_table, (size_t)bytes, hipMemcpyHostToDevice, s));
  CAMPX_TRY(hipStreamSynchronize(s));
#undef CAMPX_TRY
done:
  free(host);
  (void)hipFree(dev);
  return rc;
}

int32_t campx_reset_launch(const CampxSpec* spec_host, const CampxSpec* spec_dev, CampxState state,
                           CampxOutputs out, int64_t B, void* stream) {
  return launch(spec_host, spec_dev, state, nullptr, out, B, 0, 1, 1, stream);
}

int32_t campx_rollout_launch(const CampxSpec* spec_host, const CampxSpec* spec_dev,
                             CampxState state, const int8_t* actions, CampxOutputs out, int64_t B,
                             int32_t T, int32_t reset_first, void* stream) {
  return launch(spec_host, spec_dev, state, actions, out, B, T, reset_first, 0, stream);
}

int32_t campx_update_launch(const CampxSpec* spec_host, const CampxSpec* spec_dev, CampxState st,
                            const int8_t* actions, CampxOutputs out, int64_t B, int32_t T,
                            int32_t reset_first, void* stream) {
  if (!spec_host || !spec_dev || !st.pos || !st.done || !actions || !out.trace || B <= 0 || T <= 0)
    return CAMPX_EINVAL;
  if (out.perf && spec_host->perf_dyn < 0) return CAMPX_EINVAL;
  const int32_t v = campx_spec_validate(spec_host);
  if (v != CAMPX_OK) return v;
  if (!spec_host->render_valid) return CAMPX_ESPEC;
  const bool use_table = spec_host->table_valid && spec_host->n_dyn == 1 && !knob_no_table();
  return launch_update(*spec_host, spec_dev, st, actions, out, B, T, reset_first, use_table,
                       (int64_t)T * B, static_cast<hipStream_t>(stream));
}

int32_t campx_render_launch(const CampxSpec* spec_host, const CampxSpec* spec_dev, CampxOutputs out,
                            int64_t B, int32_t T, void* stream) {
  if (!spec_host || !spec_dev || !out.trace || !out.obs || B <= 0 || T <= 0 || T > 65535)
    return CAMPX_EINVAL;
  if (reinterpret_cast<uintptr_t>(out.obs) & 15) return CAMPX_EINVAL;
  if (out.obs_format < CAMPX_OBS_INT8 || out.obs_format > CAMPX_OBS_BF16) return CAMPX_EINVAL;
  const int32_t v = campx_spec_validate(spec_host);
  if (v != CAMPX_OK) return v;
  CampxOutputs probe = out;   // the conditions of the two-kernel path, frames back to back
  if (!split_ok(*spec_host, probe, B, T)) return CAMPX_EINVAL;
  return launch_renders(*spec_host, spec_dev, out, B, T, (int64_t)T * B, static_cast<hipStream_t>(stream));
}

int32_t campx_shape_spec_size(void) { return (int32_t)sizeof(CampxShapeSpec); }

int32_t campx_shape_spec_validate(const CampxShapeSpec* s) {
  if (!s) return CAMPX_EINVAL;
  if (s->magic != CAMPX_SHAPE_SPEC_MAGIC || s->version != CAMPX_SHAPE_SPEC_VERSION) return CAMPX_ESPEC;
  if (s->rows < 1 || s->cols < 1 || s->rows > 127 || s->cols > 127) return CAMPX_ESPEC;
  const int HW = s->rows * s->cols;
  if (HW > CAMPX_SHAPE_MAX_CELLS) return CAMPX_ESPEC;
  if (s->n_layers < 1 || s->n_layers > CAMPX_MAX_LAYERS) return CAMPX_ESPEC;
  if (s->n_things < 1 || s->n_things > CAMPX_SHAPE_MAX_THINGS) return CAMPX_ESPEC;
  if (s->first_drape < 0 || s->first_drape >= s->n_things) return CAMPX_ESPEC;
  uint32_t seen = 0;
  for (int u = 0; u < s->n_things; ++u) {
    if (s->update_order[u] < 0 || s->update_order[u] >= s->n_things) return CAMPX_ESPEC;
    seen |= 1u << s->update_order[u];
  }
  if (seen != (1u << s->n_things) - 1u) return CAMPX_ESPEC;
  for (int k = 0; k < s->n_things; ++k) {
    const CampxShapeThing& t = s->things[k];
    if (t.layer < 0 || t.layer >= s->n_layers) return CAMPX_ESPEC;
    if ((k < s->first_drape) != (t.is_sprite != 0) && k < s->first_drape) return CAMPX_ESPEC;
    if (k == s->first_drape && t.is_sprite) return CAMPX_ESPEC;
    if (t.n_cells < 0 || t.cell_begin < 0 || t.cell_begin + t.n_cells > CAMPX_SHAPE_MAX_LIST)
      return CAMPX_ESPEC;
    for (int i = 0; i < t.n_cells; ++i) {
      const uint16_t c = s->cells[t.cell_begin + i];
      if ((c >> 8) >= s->rows || (c & 0xff) >= s->cols) return CAMPX_ESPEC;
    }
    for (int a = 0; a < CAMPX_N_ACTIONS; ++a)
      if (t.drow[a] < 0 || t.drow[a] >= s->rows || t.dcol[a] < 0 || t.dcol[a] >= s->cols)
        return CAMPX_ESPEC;
    if ((t.has_reward_mask | t.terminate_mask) >> CAMPX_N_ACTIONS) return CAMPX_ESPEC;
  }
  for (int i = 0; i < HW; ++i)
    if (s->backdrop[i] >= s->n_layers) return CAMPX_ESPEC;
  return CAMPX_OK;
}

int32_t campx_shape_rollout_launch(const CampxShapeSpec* spec_host, const CampxShapeSpec* spec_dev,
                                   CampxState st, int8_t* backdrop_state, const int8_t* actions,
                                   CampxOutputs out, int64_t B, int32_t T, int32_t reset_first,
                                   int32_t emit_first, void* stream) {
  if (!spec_host || !spec_dev || !st.pos || !st.done || !out.obs || B <= 0 || T < 0)
    return CAMPX_EINVAL;
  if (T > 0 && !actions) return CAMPX_EINVAL;
  if (out.obs_format != CAMPX_OBS_INT8 || out.perf || out.trace) return CAMPX_EINVAL;
  if ((reinterpret_cast<uintptr_t>(out.obs) | reinterpret_cast<uintptr_t>(out.board) |
       reinterpret_cast<uintptr_t>(backdrop_state)) & 3)
    return CAMPX_EINVAL;
  const int32_t v = campx_shape_spec_validate(spec_host);
  if (v != CAMPX_OK) return v;
  bool trails = false;
  for (int k = 0; k < spec_host->first_drape; ++k) trails = trails || spec_host->things[k].visible;
  if (trails && !backdrop_state) return CAMPX_EINVAL;
  const dim3 grid((unsigned)((B + kShapeWaves - 1) / kShapeWaves)), block(kShapeWaves * kWave);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const ShapeParams sp = make_shape_params(*spec_host);
  if (out.board)
    hipLaunchKernelGGL(shape_rollout_kernel<true>, grid, block, 0, s, sp, spec_dev, st,
                       backdrop_state, actions, out, B, T, reset_first, emit_first);
  else
    hipLaunchKernelGGL(shape_rollout_kernel<false>, grid, block, 0, s, sp, spec_dev, st,
                       backdrop_state, actions, out, B, T, reset_first, emit_first);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? CAMPX_OK : hip_failed(e);
}

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

const char* campx_strerror(int32_t code) {
  switch (code) {
    case CAMPX_OK:
      return "ok";
    case CAMPX_EINVAL:
      return "invalid argument (NULL, misaligned or out of range)";
    case CAMPX_ESPEC:
      return "GameSpec failed validation";
    case CAMPX_ELAUNCH:
      return "HIP launch failed (see campx_last_hip_error)";
    case CAMPX_ENODEV:
      return "no usable HIP device";
    case CAMPX_ENOMEM:
      return "out of host memory";
    default:
      return "unknown campx error";
  }
}

int32_t campx_last_hip_error(void) { return g_last_hip_error; }

int32_t campx_device_arch(int32_t ordinal, char* buf, int32_t buf_len) {
  if (!buf || buf_len < 2) return CAMPX_EINVAL;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || ordinal < 0 || ordinal >= n) return CAMPX_ENODEV;
  hipDeviceProp_t prop;
  const hipError_t e = hipGetDeviceProperties(&prop, ordinal);
  if (e != hipSuccess) return hip_failed(e);
  strncpy(buf, prop.gcnArchName, (size_t)buf_len - 1);
  buf[buf_len - 1] = '\0';
  return CAMPX_OK;
}

}  // extern "C"
