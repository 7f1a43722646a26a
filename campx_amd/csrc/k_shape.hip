// k_shape.hip - the shape tier (Hello World): shape_rollout_kernel and its C entry points.

#include "campx_common.hip.h"

namespace campx_impl {

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
  uint32_t one_cell[CAMPX_SHAPE_MAX_THINGS];  // things of ONE cell (every sprite): its art cell,
                                           // row << 8 | col - painted from the scalar unit
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

// kFmt: CAMPX_OBS_INT8, or f16 / bf16 observations (0.0 / 1.0) for a policy network - the
// eight cells a lane expands per layer plane then go out as one 16-byte store of eight halves.
template <bool kBoard, int kFmt = 0>
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
        if (n == 1) {
          // a one-cell thing (every sprite): its cell comes off the scalar unit and one lane
          // writes one byte - no cell-list read, no per-lane wrap arithmetic (a pass of the
          // loop below costs ~15 vector instructions whatever the thing's size)
          const uint32_t packed = sp.one_cell[z];
          int r = (int)(packed >> 8) + dr, c = (int)(packed & 0xffu) + dc;
          r = r >= H ? r - H : r;
          c = c >= W ? c - W : c;
          if (lane == 0) target[r * W + c] = layer;
          continue;
        }
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
        // one layer's eight cells (bytes 0 / 1 in e0, e1) to memory, in the observation format
        auto put = [&](uint32_t e0, uint32_t e1) {
          if (kFmt == 0) {
            *reinterpret_cast<uint2*>(plane + at) = make_uint2(e0, e1);
          } else {
            constexpr uint32_t kOne = (kFmt == 1) ? 0x3C00u : 0x3F80u;
            u32x4 v;
            v.x = ((e0 & 0xffu) | ((e0 << 8) & 0x00ff0000u)) * kOne;
            v.y = (((e0 >> 16) & 0xffu) | ((e0 >> 8) & 0x00ff0000u)) * kOne;
            v.z = ((e1 & 0xffu) | ((e1 << 8) & 0x00ff0000u)) * kOne;
            v.w = (((e1 >> 16) & 0xffu) | ((e1 >> 8) & 0x00ff0000u)) * kOne;
            *reinterpret_cast<u32x4*>(plane + 2 * (int64_t)at) = v;   // `plane` counts bytes
          }
        };
        constexpr int kElem = kFmt ? 2 : 1;
        if (L <= 8) {
          // v_perm_b32 as a byte-wise table lookup: a board byte b (a layer index 0..7)
          // selects byte b of the 64-bit constant 1 << 8 l, which is (b == l) - one
          // instruction per four cells per layer, against four for the arithmetic below
          uint32_t hi = 0u, lo = 1u;     // 1 << 8 l as {hi, lo}
          for (int l = 0; l < L; ++l) {
            const uint32_t e0 = __builtin_amdgcn_perm(hi, lo, b0);
            const uint32_t e1 = __builtin_amdgcn_perm(hi, lo, b1);
            put(e0, e1);
            plane += kElem * HW;
            hi = (l == 3) ? 1u : hi << 8;
            lo = lo << 8;                // (0 after the fourth layer)
          }
        } else {
          uint32_t lc = 0u;
          for (int l = 0; l < L; ++l) {
            // bytes < 0x80: 0x80 - (b ^ l) has bit 7 set iff they are equal
            const uint32_t e0 = ((0x80808080u - (b0 ^ lc)) & 0x80808080u) >> 7;
            const uint32_t e1 = ((0x80808080u - (b1 ^ lc)) & 0x80808080u) >> 7;
            put(e0, e1);
            plane += kElem * HW;
            lc += 0x01010101u;
          }
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
        for (int l = 0; l < L; ++l) {
          if (kFmt == 0)
            obs_dst[(int64_t)l * HW + i] = (int8_t)(b == l);
          else
            reinterpret_cast<uint16_t*>(obs_dst)[(int64_t)l * HW + i] =
                (uint16_t)(b == l ? ((kFmt == 1) ? 0x3C00u : 0x3F80u) : 0u);
        }
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
    // (an action nobody rewards - Hello World's quit, an id outside 0..4 - reports None =
    // NaN for the frame and leaves the running return alone)
    if (valid && (sp.act[a_raw].flags & 2u)) ret += reward;
    const int64_t slot = showtime ? 0 : t;
    paint_and_emit(out.obs + (slot * out.obs_t_stride + env * LHW) * (kFmt ? 2 : 1),
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
          const int64_t at = (int64_t)(t0 + lane) * row_pitch(out, B) + env;
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
    if (th.n_cells == 1) sp.one_cell[k] = s.cells[th.cell_begin];
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


// ---------------------------------------------------------------------------
// Two-kernel path for shape games WITHOUT trails (no visible sprite is painted before the
// first drape: campx/rendering.py:128,150,178 - the backdrop is then the same in every frame
// of every environment).  A frame's observation is a function of the things' offsets alone,
// so - as for the one-cell games - the update pass can run ahead and the observation stream
// can be written frame-major by one-shot blocks with memory-aligned stores, the store pattern
// that reaches the chip's write ceiling (the serial kernel's "every wave streams its own row
// per frame" stays at 4.1-4.4 TB/s whatever its instruction count: NOTES.md 3.7).
//
//   shape_update_kernel  one lane per environment, T frames: offsets += delta[action]
//                        (the same SWAR words as the serial kernel), the frame's offsets
//                        as four dwords [4][T][B] - the shape trace -, reward / discount /
//                        done, state back.
//   shape_render_kernel  block (x, t) writes the 16 KiB window x of frame t (aligned in
//                        MEMORY to 1 KiB).  Its four waves take the environments whose rows
//                        overlap the window in turn: board = backdrop + things at their
//                        offsets, in LDS; expand into the window's LDS image (8 cells per
//                        lane, every layer: v_perm_b32 lookups), clipped to the window; then
//                        all waves stream the image out, one aligned KiB per wave-instruction.
//                        A row that straddles two windows is built by both blocks (one row
//                        in six at Hello World's size).
// A/B knobs: bytes of a frame per block and waves per block.  A row that straddles two
// windows is built by both blocks, so small windows repeat work (16 KiB at Hello World's
// 3 276-byte rows: 7 boards per 5 rows of output, and the whole kernel ran at 3.7 TB/s -
// instruction-bound); 64 KiB: 22 per 20.
#ifndef CAMPX_SHAPE_SPAN_KB
#define CAMPX_SHAPE_SPAN_KB 64
#endif
#ifndef CAMPX_SHAPE_RENDER_WAVES
#define CAMPX_SHAPE_RENDER_WAVES 8
#endif
constexpr int kRenderSpan = CAMPX_SHAPE_SPAN_KB * 1024;   // bytes of a frame per block
constexpr int kRenderBlockWaves = CAMPX_SHAPE_RENDER_WAVES;

__global__ __launch_bounds__(256) void shape_update_kernel(
    ShapeParams sp, CampxState st, const int8_t* __restrict__ actions, CampxOutputs out,
    uint32_t* __restrict__ trace, int64_t B, int32_t T, int32_t reset_first) {
  __shared__ ShapeAction act[CAMPX_N_ACTIONS];   // (indexed by a lane's action: LDS, not kernarg)
  if (threadIdx.x < CAMPX_N_ACTIONS) act[threadIdx.x] = sp.act[threadIdx.x];
  __syncthreads();
  const int64_t env = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= B) return;
  const int H = sp.rows, W = sp.cols, N = sp.n_things;
  uint32_t orow[2] = {0u, 0u}, ocol[2] = {0u, 0u};
  int over = 0, bad = 0;
  float ret = 0.0f;
  if (!reset_first) {
#pragma unroll
    for (int k = 0; k < CAMPX_SHAPE_MAX_THINGS; ++k)
      if (k < N) {
        orow[k >> 2] |= (uint32_t)(uint8_t)st.pos[(int64_t)(2 * k) * B + env] << (8 * (k & 3));
        ocol[k >> 2] |= (uint32_t)(uint8_t)st.pos[(int64_t)(2 * k + 1) * B + env] << (8 * (k & 3));
      }
    over = st.done[env];
    if (st.ret) ret = st.ret[env];
  }
  const int64_t plane = (int64_t)T * B;
  constexpr int kAhead = 16;                    // actions in flight per lane
  for (int t0 = 0; t0 < T; t0 += kAhead) {
    int a16[kAhead];
#pragma unroll
    for (int j = 0; j < kAhead; ++j) {
      const int t = t0 + j < T ? t0 + j : T - 1;
      a16[j] = actions[(int64_t)t * B + env];
    }
#pragma unroll
    for (int j = 0; j < kAhead; ++j) {
      const int t = t0 + j;
      if (t < T) {
        const int a = a16[j];
        const bool valid = (unsigned)a < (unsigned)CAMPX_N_ACTIONS;
        bad += valid ? 0 : 1;
        if (over) {   // a fresh make_game() + its_showtime()
          orow[0] = orow[1] = ocol[0] = ocol[1] = 0u;
          over = 0;
          ret = 0.0f;
        }
        float reward = __builtin_nanf("");
        if (valid) {
          const ShapeAction e = act[a];
          orow[0] = swar_add_wrap(orow[0], e.drow[0], (uint32_t)H);
          ocol[0] = swar_add_wrap(ocol[0], e.dcol[0], (uint32_t)W);
          orow[1] = swar_add_wrap(orow[1], e.drow[1], (uint32_t)H);
          ocol[1] = swar_add_wrap(ocol[1], e.dcol[1], (uint32_t)W);
          reward = e.reward;
          if (e.flags & 2u) ret += reward;
          if (e.flags & 1u) over = 1;   // plot.py:183-184 (discount 0 on that frame)
        }
        const int64_t at = (int64_t)t * B + env;
        trace[at] = orow[0];
        trace[plane + at] = orow[1];
        trace[2 * plane + at] = ocol[0];
        trace[3 * plane + at] = ocol[1];
        if (out.reward) out.reward[at] = reward;
        if (out.discount) out.discount[at] = over ? 0.0f : 1.0f;
        if (out.done) out.done[at] = (uint8_t)over;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < CAMPX_SHAPE_MAX_THINGS; ++k)
    if (k < N) {
      st.pos[(int64_t)(2 * k) * B + env] = (int8_t)((orow[k >> 2] >> (8 * (k & 3))) & 0xffu);
      st.pos[(int64_t)(2 * k + 1) * B + env] = (int8_t)((ocol[k >> 2] >> (8 * (k & 3))) & 0xffu);
    }
  st.done[env] = (uint8_t)over;
  if (st.ret) st.ret[env] = ret;
  report_bad_actions(out, bad);
}

struct ShapeRenderParams {
  uint32_t R;                 // bytes of an environment's row of the stream (L*H*W, or H*W: flat board)
  uint32_t m, sh1, sh2;       // exact n / R for 32-bit n (Granlund-Montgomery)
  uint32_t slab_bytes;        // B * R, a multiple of 16
  uint32_t shift_base, shift_slab;   // (address of frame 0) and slab_bytes modulo 1 KiB
};

template <bool kBoard>
__global__ __launch_bounds__(kRenderBlockWaves * kWave) void shape_render_kernel(
    ShapeParams sp, ShapeRenderParams rp, const CampxShapeSpec* __restrict__ spec,
    const uint32_t* __restrict__ trace, int8_t* __restrict__ dst, int64_t B, int64_t plane) {
  __shared__ __attribute__((aligned(16))) int8_t image[kRenderSpan];
  __shared__ __attribute__((aligned(16))) uint8_t lds_board[kRenderBlockWaves][CAMPX_SHAPE_MAX_CELLS];
  __shared__ __attribute__((aligned(16))) uint8_t lds_backdrop[CAMPX_SHAPE_MAX_CELLS];
  __shared__ uint16_t lds_cells[CAMPX_SHAPE_MAX_LIST];
  __shared__ uint32_t lds_char[CAMPX_MAX_LAYERS / 4];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int H = sp.rows, W = sp.cols, HW = H * W, L = sp.n_layers, N = sp.n_things;
  // windows are aligned in memory: they start `shift` bytes before the frame (offsets
  // modulo 2^32: the head window's bytes before the frame fail the one bounds test)
  const uint32_t shift = (rp.shift_base + blockIdx.y * rp.shift_slab) & 1023u;
  if ((uint64_t)blockIdx.x * kRenderSpan >= (uint64_t)rp.slab_bytes + shift) return;
  const uint32_t w0 = blockIdx.x * (uint32_t)kRenderSpan - shift;          // first byte of the window
  const uint32_t wlo = blockIdx.x == 0 ? 0u : w0;                           // ... inside the frame
  const uint32_t whi_end = (w0 + kRenderSpan < rp.slab_bytes) ? w0 + kRenderSpan : rp.slab_bytes;
  const uint32_t h0 = __umulhi(rp.m, wlo), h1 = __umulhi(rp.m, whi_end - 1u);
  const uint32_t first_row = (((wlo - h0) >> rp.sh1) + h0) >> rp.sh2;
  const uint32_t last_row = (((whi_end - 1u - h1) >> rp.sh1) + h1) >> rp.sh2;

  for (int i = threadIdx.x; i < sp.n_list; i += kRenderBlockWaves * kWave) lds_cells[i] = spec->cells[i];
  for (int i = threadIdx.x; 4 * i < HW; i += kRenderBlockWaves * kWave)
    reinterpret_cast<uint32_t*>(lds_backdrop)[i] = reinterpret_cast<const uint32_t*>(spec->backdrop)[i];
  if (kBoard && threadIdx.x < CAMPX_MAX_LAYERS / 4) lds_char[threadIdx.x] = sp.layer_char[threadIdx.x];
  __syncthreads();

  uint8_t* board = lds_board[wave];
  const uint32_t* frame_trace = trace + (int64_t)blockIdx.y * B;
  for (uint32_t row = first_row + (uint32_t)wave; row <= last_row; row += kRenderBlockWaves) {
    // the things' offsets in environment `row` at this frame (the same address in every lane)
    const uint32_t or0 = __builtin_amdgcn_readfirstlane(frame_trace[row]);
    const uint32_t or1 = __builtin_amdgcn_readfirstlane(frame_trace[plane + row]);
    const uint32_t oc0 = __builtin_amdgcn_readfirstlane(frame_trace[2 * plane + row]);
    const uint32_t oc1 = __builtin_amdgcn_readfirstlane(frame_trace[3 * plane + row]);
    // board = backdrop, then every thing back to front (one wave: LDS operations complete
    // in order, no barrier)
    for (int base = 0; base * 16 < HW; base += kWave) {
      const int i = base + lane;
      if (i * 16 < HW) reinterpret_cast<u32x4*>(board)[i] = reinterpret_cast<const u32x4*>(lds_backdrop)[i];
    }
    for (int z = 0; z < N; ++z) {
      const uint32_t th = sp.thing[z];
      if (!((th >> 28) & 1u)) continue;
      const int begin = (int)(th & 0x7ffu), n = (int)((th >> 11) & 0xfffu);
      const uint8_t layer = (uint8_t)((th >> 23) & 0x1fu);
      const int sh = 8 * (z & 3);
      const int dr = (int)(((z < 4 ? or0 : or1) >> sh) & 0xffu);
      const int dc = (int)(((z < 4 ? oc0 : oc1) >> sh) & 0xffu);
      if (n == 1) {
        const uint32_t packed = sp.one_cell[z];
        int r = (int)(packed >> 8) + dr, c = (int)(packed & 0xffu) + dc;
        r = r >= H ? r - H : r;
        c = c >= W ? c - W : c;
        if (lane == 0) board[r * W + c] = layer;
        continue;
      }
      for (int base = 0; base < n; base += kWave) {
        const int i = base + lane;
        if (i < n) {
          const uint32_t packed = lds_cells[begin + i];
          int r = (int)(packed >> 8) + dr, c = (int)(packed & 0xffu) + dc;
          r = r >= H ? r - H : r;
          c = c >= W ? c - W : c;
          board[r * W + c] = layer;
        }
      }
    }
    // expand into the window's image: 8 cells per lane (the launcher admits H * W % 4 == 0,
    // H * W >= 8 only).  A row wholly inside the window - most - writes 8 bytes per layer
    // unconditionally; the window's first and last row are clipped, two guarded 4-byte
    // writes per layer.
    const uint32_t row0 = row * rp.R - w0;          // the row's first byte, relative to the window
    const bool inside = row0 <= (uint32_t)kRenderSpan - rp.R;   // (uniform; a wrapped row0 is huge)
    if (inside && !kBoard && L <= 8) {
      for (int qbase = 0; 8 * qbase < HW; qbase += kWave) {
        const int q = qbase + lane;
        if (8 * q >= HW) continue;
        const uint32_t at = (uint32_t)(8 * q + 8 <= HW ? 8 * q : HW - 8);
        const uint32_t b0 = *reinterpret_cast<const uint32_t*>(board + at);
        const uint32_t b1 = *reinterpret_cast<const uint32_t*>(board + at + 4);
        int8_t* to = image + row0 + at;             // (4-byte aligned: two dword writes)
        uint32_t hi = 0u, lo = 1u;
        for (int l = 0; l < L; ++l) {
          reinterpret_cast<uint32_t*>(to)[0] = __builtin_amdgcn_perm(hi, lo, b0);
          reinterpret_cast<uint32_t*>(to)[1] = __builtin_amdgcn_perm(hi, lo, b1);
          to += HW;
          hi = (l == 3) ? 1u : hi << 8;
          lo = lo << 8;
        }
      }
      continue;
    }
    for (int qbase = 0; 8 * qbase < HW; qbase += kWave) {
      const int q = qbase + lane;
      if (8 * q >= HW) continue;
      const uint32_t at = (uint32_t)(8 * q + 8 <= HW ? 8 * q : HW - 8);
      const uint32_t b0 = *reinterpret_cast<const uint32_t*>(board + at);
      const uint32_t b1 = *reinterpret_cast<const uint32_t*>(board + at + 4);
      if (kBoard) {
        const uint8_t* ch = reinterpret_cast<const uint8_t*>(lds_char);
        auto chars = [&](uint32_t b4) {
          return (uint32_t)ch[b4 & 0xffu] | ((uint32_t)ch[(b4 >> 8) & 0xffu] << 8) |
                 ((uint32_t)ch[(b4 >> 16) & 0xffu] << 16) | ((uint32_t)ch[b4 >> 24] << 24);
        };
        const uint32_t io = row0 + at;
        if (io < (uint32_t)kRenderSpan) *reinterpret_cast<uint32_t*>(image + io) = chars(b0);
        if (io + 4u < (uint32_t)kRenderSpan) *reinterpret_cast<uint32_t*>(image + io + 4u) = chars(b1);
      } else if (L <= 8) {
        uint32_t hi = 0u, lo = 1u, io = row0 + at;    // 1 << 8 l as {hi, lo}: see shape_rollout_kernel
        for (int l = 0; l < L; ++l) {
          if (io < (uint32_t)kRenderSpan)
            *reinterpret_cast<uint32_t*>(image + io) = __builtin_amdgcn_perm(hi, lo, b0);
          if (io + 4u < (uint32_t)kRenderSpan)
            *reinterpret_cast<uint32_t*>(image + io + 4u) = __builtin_amdgcn_perm(hi, lo, b1);
          io += (uint32_t)HW;
          hi = (l == 3) ? 1u : hi << 8;
          lo = lo << 8;
        }
      } else {
        uint32_t lc = 0u, io = row0 + at;
        for (int l = 0; l < L; ++l) {
          if (io < (uint32_t)kRenderSpan)
            *reinterpret_cast<uint32_t*>(image + io) = ((0x80808080u - (b0 ^ lc)) & 0x80808080u) >> 7;
          if (io + 4u < (uint32_t)kRenderSpan)
            *reinterpret_cast<uint32_t*>(image + io + 4u) = ((0x80808080u - (b1 ^ lc)) & 0x80808080u) >> 7;
          io += (uint32_t)HW;
          lc += 0x01010101u;
        }
      }
    }
  }
  __syncthreads();
  // ---- out: every store instruction of a wave is one aligned, contiguous KiB
  const int8_t* frame = dst + (int64_t)blockIdx.y * rp.slab_bytes;   // uniform
#pragma unroll
  for (int j = 0; j < kRenderSpan / (kRenderBlockWaves * kWave * 16); ++j) {
    const uint32_t o = ((uint32_t)j * kRenderBlockWaves * kWave + threadIdx.x) * 16u;
    const uint32_t off = w0 + o;
    if (off < rp.slab_bytes)        // (chunks before the frame wrap to huge offsets; the slab is whole chunks)
      store16_streaming_at(frame, off, *reinterpret_cast<const u32x4*>(image + o));
  }
}

// Can this launch take the two-kernel path?  No trails, every frame kept back to back, rows
// of whole dwords and frames of whole 16-byte chunks below 4 GiB, boards of at least 8 cells.
bool shape_split_ok(const CampxShapeSpec& s, const CampxOutputs& out, int64_t B, int32_t T,
                    int32_t emit_first) {
  if (!out.trace || emit_first || T < 1 || T > 65535) return false;
  for (int k = 0; k < s.first_drape; ++k)
    if (s.things[k].visible) return false;
  const int64_t HW = (int64_t)s.rows * s.cols, R = HW * s.n_layers;
  if (HW % 4 || HW < 8 || (B * R) % 16 || B * R >= (1ll << 32) - 65536) return false;
  if (out.board && ((B * HW) % 16 || out.board_t_stride != B * HW)) return false;
  if (out.scalar_pitch && out.scalar_pitch != B) return false;
  return out.obs_t_stride == B * R;
}

template <bool kBoard>
void launch_shape_render(const ShapeParams& sp, const CampxShapeSpec& s, const CampxShapeSpec* spec_dev,
                         const uint32_t* trace, int8_t* dst, int64_t B, int32_t T, hipStream_t stream) {
  ShapeRenderParams rp;
  memset(&rp, 0, sizeof(rp));
  rp.R = (uint32_t)(s.rows * s.cols * (kBoard ? 1 : s.n_layers));
  uint32_t l = 0;
  while ((1ull << l) < rp.R) ++l;
  rp.m = (uint32_t)(((1ull << 32) * ((1ull << l) - rp.R)) / rp.R + 1);
  rp.sh1 = l < 1 ? l : 1;
  rp.sh2 = l > 0 ? l - 1 : 0;
  rp.slab_bytes = (uint32_t)(B * rp.R);
  rp.shift_base = (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 1023u);
  rp.shift_slab = rp.slab_bytes & 1023u;
  const uint64_t reach = (uint64_t)rp.slab_bytes + ((rp.shift_base | rp.shift_slab) ? 1023u : 0u);
  const dim3 grid((unsigned)((reach + kRenderSpan - 1) / kRenderSpan), (unsigned)T);
  hipLaunchKernelGGL((shape_render_kernel<kBoard>), grid, dim3(kRenderBlockWaves * kWave), 0, stream,
                     sp, rp, spec_dev, trace, dst, B, (int64_t)T * B);
}

}  // namespace campx_impl

using namespace campx_impl;

extern "C" {

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
  if (out.obs_format < CAMPX_OBS_INT8 || out.obs_format > CAMPX_OBS_BF16 || out.perf) return CAMPX_EINVAL;
  if (out.trace && (reinterpret_cast<uintptr_t>(out.trace) & 3)) return CAMPX_EINVAL;
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
  // (the 16-bit formats: the serial kernel only; its 16-byte stores of eight halves are 8-byte
  // aligned when rows*cols is 4 modulo 8 - legal on this stack, tools/probes/unaligned_probe.hip)
  if (out.obs_format == CAMPX_OBS_INT8 && shape_split_ok(*spec_host, out, B, T, emit_first)) {
    // games without trails, every frame kept: update pass -> shape trace -> frame-major render
    uint32_t* trace = reinterpret_cast<uint32_t*>(out.trace);
    hipLaunchKernelGGL(shape_update_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, sp, st,
                       actions, out, trace, B, T, reset_first);
    // (a third variant - one wave per environment-frame storing its row straight from
    // registers, no window image - measured 4.08 TB/s at B = 32 768, between this path and
    // the serial kernel; removed again)
    launch_shape_render<false>(sp, *spec_host, spec_dev, trace, out.obs, B, T, s);
    if (out.board) launch_shape_render<true>(sp, *spec_host, spec_dev, trace, out.board, B, T, s);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? CAMPX_OK : hip_failed(e);
  }
#define CAMPX_SHAPE_LAUNCH(BOARD, FMT)                                                         \
  hipLaunchKernelGGL((shape_rollout_kernel<BOARD, FMT>), grid, block, 0, s, sp, spec_dev, st,  \
                     backdrop_state, actions, out, B, T, reset_first, emit_first)
  if (out.obs_format == CAMPX_OBS_F16) {
    if (out.board) CAMPX_SHAPE_LAUNCH(true, 1); else CAMPX_SHAPE_LAUNCH(false, 1);
  } else if (out.obs_format == CAMPX_OBS_BF16) {
    if (out.board) CAMPX_SHAPE_LAUNCH(true, 2); else CAMPX_SHAPE_LAUNCH(false, 2);
  } else {
    if (out.board) CAMPX_SHAPE_LAUNCH(true, 0); else CAMPX_SHAPE_LAUNCH(false, 0);
  }
#undef CAMPX_SHAPE_LAUNCH
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? CAMPX_OK : hip_failed(e);
}

}  // extern "C"
