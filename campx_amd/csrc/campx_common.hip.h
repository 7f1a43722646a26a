// campx_common.hip.h - what the kernels' translation units share: constants, the streaming
// store flavours, the rule interpreter's state types, LDS image fill / stream-out helpers,
// parameter blocks that more than one kernel takes, and the declarations of the host-side
// launchers (one per k_*.hip file) that campx_api.hip dispatches to.
//
// Everything device-side in here is __forceinline__; nothing in here defines a kernel.
#ifndef CAMPX_COMMON_HIP_H_
#define CAMPX_COMMON_HIP_H_

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "campx_hip.h"


namespace campx_impl {

constexpr int kWave = 64;
// Actions are staged through LDS kChunk frames at a time, so that the frame loop
// itself issues no global loads: a load in the loop would make every frame wait
// (vmcnt is in-order) for the previous frames' observation stores to drain.
constexpr int kChunk = 64;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// The streaming 16-byte store of the observation writers: system-scope write-through +
// non-temporal, i.e. the line is not kept anywhere on its way to HBM.  (The other cache policies
// were builds of round 1 - ms per 100-frame boat-race launch, fused / split path: plain 0.254 /
// 0.336, nt 0.233 / 0.220, sc1 0.259 / 0.247, sc0 sc1 0.261 / 0.250, sc0 sc1 nt 0.224 / 0.196 - and
// are gone.)
__device__ __forceinline__ void store16_streaming(u32x4* p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// The same store with the address as (wave-uniform 64-bit base in SGPRs) + (32-bit byte
// offset per lane): no 64-bit vector address arithmetic.  `base` must be provably uniform
// (kernel arguments, blockIdx).
__device__ __forceinline__ void store16_streaming_at(const void* base, uint32_t off, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1 nt\n\ts_nop 1" ::"v"(off), "v"(v), "s"(base)
               : "memory");
}

// The part of the GameSpec the interpreter reads every frame.  Passed BY VALUE
// so that it lives in the kernarg segment (scalar loads, scalar branches).
struct RuleBlock {
  int32_t rows, cols, n_layers, n_dyn, n_rules, any_reward;
  int32_t perf_dyn, perf_n;
  int32_t perf_mode, perf_mask, perf_scale, perf_offset;
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

// What the state tables say about a frame besides where things are: a 3-bit hidden-
// performance code (value = scale * code + offset, CampxSpec.perf_*) and a 4-bit discount
// code (0 = the default: 0.0 when the episode ended on the frame, else 1.0; otherwise an
// index into CampxSpec.discount_list).  By value in the kernel arguments; the kernels copy
// the discount list to LDS before indexing it with a lane's code.
struct FrameCodec {
  int32_t perf_scale, perf_offset;
  int32_t has_dcodes;            // some entry of the game's table carries a discount code
  float discounts[16];
};

inline FrameCodec make_codec(const CampxSpec& s) {
  FrameCodec c;
  memset(&c, 0, sizeof(c));
  c.perf_scale = s.perf_scale;
  c.perf_offset = s.perf_offset;
  for (int i = 1; i < 16; ++i) {
    c.discounts[i] = s.discount_list[i];
    if (s.discount_list[i] != 0.0f) c.has_dcodes = 1;
  }
  c.has_dcodes = c.has_dcodes || s.table_only;   // (a listed discount may be 0.0 itself)
  c.discounts[0] = 1.0f;
  return c;
}

__device__ __forceinline__ uint32_t perf_byte(const FrameCodec& fc, uint32_t code) {
  return (uint32_t)((int)code * fc.perf_scale + fc.perf_offset) & 0xffu;
}

// bits of a discount as a float: `list` is the LDS copy of FrameCodec.discounts
__device__ __forceinline__ uint32_t discount_bits(const float* list, uint32_t dcode, uint32_t done) {
  const uint32_t plain = done ? 0u : 0x3f800000u;
  return dcode ? __float_as_uint(list[dcode]) : plain;
}

// the codes out of the upper half of an update_table_kernel LDS entry (y >> 16), of a pair
// entry and of the high word of a tuple entry
__device__ __forceinline__ uint32_t perf_code_y16(uint32_t y) { return ((y >> 9) & 3u) | ((y >> 13) & 4u); }
__device__ __forceinline__ uint32_t dcode_y16(uint32_t y) { return (y >> 11) & 15u; }
__device__ __forceinline__ uint32_t perf_code_pair(uint32_t e) { return ((e >> 17) & 3u) | ((e >> 29) & 4u); }
__device__ __forceinline__ uint32_t dcode_pair(uint32_t e) { return (e >> 27) & 15u; }
__device__ __forceinline__ uint32_t perf_code_tuple(uint32_t hi) { return ((hi >> 1) & 3u) | ((hi >> 9) & 4u); }
__device__ __forceinline__ uint32_t dcode_tuple(uint32_t hi) { return (hi >> 12) & 15u; }

// code of a perf VALUE (host side, when tables are packed); -1 if it is not one of the eight
inline int perf_code_of(const CampxSpec& s, int value) {
  if (s.perf_dyn < 0 || s.perf_scale == 0) return value == 0 ? 0 : -1;
  const int d = value - s.perf_offset;
  if (d % s.perf_scale) return -1;
  const int c = d / s.perf_scale;
  return (c >= 0 && c <= 7) ? c : -1;
}

// Row pitch of the per-frame scalar streams and the trace (CampxOutputs.scalar_pitch), and
// the extent of a row that may be written: with a pitch that leaves room for it the batch's
// last 16-element group is stored whole (into the pad) instead of element by element.
__host__ __device__ __forceinline__ int64_t row_pitch(const CampxOutputs& out, int64_t B) {
  return out.scalar_pitch ? out.scalar_pitch : B;
}
__host__ __device__ __forceinline__ int64_t row_extent(const CampxOutputs& out, int64_t B) {
  const int64_t up = (B + 15) & ~(int64_t)15;
  return (out.scalar_pitch >= up) ? up : B;
}

// What a frame's reward adds to the running return: a frame nobody rewarded reports None
// (NaN, campx/plot.py:208-211) and adds nothing.
__device__ __forceinline__ float real_reward(float r) { return r == r ? r : 0.0f; }

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


// Parameters of the one-mover kernels (rollout_table_kernel, step_*_kernel, update_table_kernel).
struct MoverParams {
  int32_t rows, cols, n_layers, dyn_layer, dyn_z, row0, col0;
};

// Parameters of the three- and four-mover kernels (update_tuple_kernel, step_tuple_kernel).
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

inline TupleParams make_tuple_params(const CampxSpec& s) {
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

// ---------------------------------------------------------------------------- host side

// campx_api.hip
extern thread_local int32_t g_last_hip_error;
int32_t hip_failed(hipError_t e);
// The library's settings (campx_api.hip holds the table: name, default, range, what each selects;
// campx_config_set / _get / _string in include/campx_hip.h).  Read at every use - a plain load.
enum Knob : int {
  K_TRACE_CHUNK_MB, K_TRACE_WHOLE_MB, K_SHAPE_CHUNK_KF, K_SHAPE_SPLIT, K_BIG_WGS, K_FLOW,
  K_FLOW_MAX_NAPS, K_FLOW_DEBUG_DELAY, K_WIDE_LDS_MAX, K_WIDE_STEP, K_COUNT
};
int64_t knob(Knob k);
constexpr uint32_t kFlowMaxNaps = 1u << 20;   // looks at stale entries before a render wave gives up (seconds)
constexpr size_t kWideLdsMax = 144 * 1024;    // wide tier: state tables up to this size are staged in LDS

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

// A launcher whose kernel was refused its dynamic LDS returns CAMPX_EINVAL (the game's rows are
// too large for this device) with the HIP error in campx_last_hip_error().
int32_t lds_refused(hipError_t e);
#define CAMPX_ALLOW_LDS(kernel, bytes)                              \
  do {                                                              \
    const hipError_t lds_e_ = allow_lds(kernel, bytes);             \
    if (lds_e_ != hipSuccess) return lds_refused(lds_e_);           \
  } while (0)

// k_interp.hip: the rule interpreter (fused, or in trace mode as the update pass / table builder)
size_t lds_bytes(const CampxSpec& s, bool board, int envs);
int32_t launch_interp(const CampxSpec& s, const CampxSpec* spec_dev, CampxState st,
                      const int8_t* actions, CampxOutputs out, int64_t B, int32_t T,
                      int32_t reset_first, int32_t emit_first, hipStream_t stream);
void launch_trace(const CampxSpec& s, const CampxSpec* spec_dev, CampxState st,
                  const int8_t* actions, CampxOutputs out, int64_t B, int32_t T,
                  int32_t reset_first, int64_t trace_plane, hipStream_t stream);

// k_rollout_table.hip: one-mover games, fused
int32_t launch_table(const CampxSpec& s, const CampxSpec* spec_dev, CampxState st,
                     const int8_t* actions, CampxOutputs out, int64_t B, int32_t T,
                     int32_t reset_first, int32_t emit_first, hipStream_t stream);

// k_step.hip: Engine.play(), one frame
int32_t launch_step_table(const CampxSpec& s, const CampxSpec* spec_dev, CampxState st,
                          const int8_t* actions, CampxOutputs out, int64_t B, int32_t reset_first,
                          hipStream_t stream);
int32_t launch_step_pair(const CampxSpec& s, const CampxSpec* spec_dev, CampxState st,
                         const int8_t* actions, CampxOutputs out, int64_t B, int32_t reset_first,
                         hipStream_t stream);
int32_t launch_step_tuple(const CampxSpec& s, const CampxSpec* spec_dev, CampxState st,
                          const int8_t* actions, CampxOutputs out, int64_t B, int32_t reset_first,
                          hipStream_t stream);

// k_update.hip: the update pass of the two-kernel path
int32_t launch_update(const CampxSpec& s, const CampxSpec* spec_dev, CampxState st,
                      const int8_t* actions, CampxOutputs out, int64_t B, int32_t T,
                      int32_t reset_first, bool use_table, int64_t trace_plane,
                      hipStream_t stream);

// k_update.hip: the update pass of one rollout and the render pass of the one before it in one launch
bool pipe_ok(const CampxSpec& s, const CampxOutputs& out, const CampxOutputs& prev, int64_t B,
             int32_t T, bool use_table);
int32_t launch_pipe(const CampxSpec& s, const CampxSpec* spec_dev, CampxState st,
                    const int8_t* actions, CampxOutputs out, CampxOutputs prev, int64_t B,
                    int32_t T, int32_t reset_first, hipStream_t stream);

// k_update.hip: update pass and render of ONE rollout in one launch (the render role waits for
// the update role's published progress)
bool flow_ok(const CampxSpec& s, const CampxOutputs& out, int64_t B, int32_t T, bool use_table,
             hipStream_t stream, bool ask_stream = true);
int64_t flow_scratch_bytes(int64_t B, int32_t T);
int32_t launch_flow(const CampxSpec& s, const CampxSpec* spec_dev, CampxState st,
                    const int8_t* actions, CampxOutputs out, int64_t B, int32_t T,
                    int32_t reset_first, hipStream_t stream);

// k_render.hip: the observation stream of the two-kernel path
// Where a render launch finds a game's tables (k_render.hip).
// (render_kernel<kVar>, wide_step kernels)
// kVar: bytes [0, n) of a 16-byte scenery chunk come from one environment's row, bytes [n, 16)
// from the next environment's, whose scenery is another variant (1 <= n <= 15).
__device__ __forceinline__ u32x4 variant_chunk(const int8_t* here, int64_t stride, int v) {
  return *reinterpret_cast<const u32x4*>(here + (int64_t)v * stride);
}
__device__ __forceinline__ uint32_t merge_word(uint32_t a, uint32_t b, int left) {
  // `left`: bytes of this dword that are still the first row's (<= 0: none, >= 4: all)
  const uint32_t mask = left >= 4 ? 0xffffffffu : (left <= 0 ? 0u : ((1u << (8 * left)) - 1u));
  return (a & mask) | (b & ~mask);
}
// (component by component: arrays indexed in a loop went to scratch memory - 80 bytes a lane,
// and the kernel to 1.3 TB/s)
__device__ __forceinline__ u32x4 merge_rows(u32x4 a, u32x4 b, int n) {
  return u32x4{merge_word(a.x, b.x, n), merge_word(a.y, b.y, n - 4), merge_word(a.z, b.z, n - 8),
               merge_word(a.w, b.w, n - 12)};
}

struct RenderSource {
  int32_t rows, cols, n_layers, n_dyn;
  int32_t dyn_layer[CAMPX_WIDE_MAX_DYN];
  uint8_t layer_char[CAMPX_MAX_LAYERS];
  const int8_t* rot_obs;      // device: 16 x (round_up(R, 16) + 16) bytes, see CampxSpec.rot_obs
  const int8_t* rot_board;
  const uint8_t* top_layer;   // device: scenery layer per cell (one-byte trace only)
  bool wide;                  // 16-bit trace entries (k_wide.hip)
  // a scenery that changes with the state (CampxWideSpec.n_variants > 1): `rot_obs` / `rot_board`
  // hold one set of rotations per variant, these many bytes apart, and plane `n_dyn` of the trace
  // names each (frame, environment)'s variant
  int32_t n_variants;
  int64_t rot_obs_stride, rot_board_stride;
  // pieces of the scenery that come and go (CampxWideSpec.n_pieces > 0): plane `n_dyn` of the trace
  // holds each (frame, environment)'s 16-bit mask of the pieces that show; device tables of
  // CAMPX_WIDE_MAX_PIECES words: layered rows  (offset of the byte a piece sets) | (offset of the
  // scenery byte it clears) << 16;  flat board  cell | character << 16
  int32_t n_pieces;
  const uint32_t* pieces_obs;
  const uint32_t* pieces_board;
};
int32_t launch_render_from(const RenderSource& src, const void* trace, int8_t* dst, int64_t B,
                           int32_t T, int64_t plane_rows, int64_t pitch, bool is_board, int fmt,
                           hipStream_t stream);
int32_t launch_render(const CampxSpec& s, const CampxSpec* spec_dev, const uint8_t* trace,
                      int8_t* dst, int64_t B, int32_t T, int64_t plane_rows, int64_t pitch,
                      bool is_board, int fmt, hipStream_t stream);

}  // namespace campx_impl

#endif  // CAMPX_COMMON_HIP_H_
