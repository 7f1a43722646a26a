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
// Frame-major path (round 5): update pass + a render pass of one-shot waves with memory-aligned
// 2 KiB windows - render_kernel's store pattern, the one that reaches the chip's write ceiling
// (the serial kernel's "every wave streams its own row per frame" stays at 4.2-5.3 TB/s).  Round 3's
// attempt painted whole boards in LDS per row and lost (NOTES.md 3.7); this one never paints.
//
// An environment-frame of the observation is L planes of H rows of W bytes 0 / 1: L * H "slots"
// of W bits.  A slot is computed ARITHMETICALLY from 64-bit row words:
//     thing z (multi-cell): rowbits_z[dc][(r - dr) mod H] - the thing's mask, every column
//                           rotation precomputed (the tables blob: W * H words per thing);
//     thing z (one cell):   (r == its row) ? 1 << its column : 0;
//     front to back: visible_z = word_z & ~covered, covered |= word_z;
//     backdrop of layer l:  (static_l[r] & ~trails) | trail words of layer l, & ~covered.
// A wave computes the <= 2 KiB / W + 2 slots its window overlaps (one slot per lane, into LDS),
// then every lane takes the 16 bits of its 16-byte chunk from two neighbouring slots - output
// byte k of the frame is bit (k mod W) of slot k / W, whatever plane or environment it falls
// in - expands them to bytes and stores: aligned, contiguous KiB per wave-instruction.
//
// Trails (sprites painted before the first drape write into the backdrop for good,
// campx/rendering.py:128,150): the update pass keeps, per environment, one W-bit word per trail
// sprite and board row ("painted by this sprite last") and writes them out every kShapeKey-th
// frame (a KEYFRAME, 8 * S * H bytes per environment: 1.6 % of the observation stream for Hello
// World); a render wave starts from the keyframe at or before its frame and applies the at most
// kShapeKey - 1 frames of sprite positions since, which the offset trace holds anyway.  A frame
// that began with a rebuild (the episode had ended) carries a flag in the trace: trails restart.
#ifndef CAMPX_SHAPE_KEY
#define CAMPX_SHAPE_KEY 4
#endif
constexpr int kShapeKey = CAMPX_SHAPE_KEY;
constexpr uint32_t kShapeTablesMagic = 0x54485343u;   // 'CSHT'
#ifndef CAMPX_SHAPE_SPLIT_WIN
#define CAMPX_SHAPE_SPLIT_WIN 8
#endif
// One wave per block, kSplitWin KiB of a frame per wave (see shape_render_split_kernel).
constexpr int kSplitWaves = 1, kSplitWin = CAMPX_SHAPE_SPLIT_WIN;
constexpr uint32_t kSplitSpan = 1024u * kSplitWin;

struct ShapeTablesHeader {       // the device blob campx_shape_tables_build() fills
  uint32_t magic;
  int32_t rows, cols, n_layers, n_things, first_drape;
  uint32_t static_off;           // uint64 [L][H]: the art's backdrop, one word per layer and row
  uint32_t rowbits_off[CAMPX_SHAPE_MAX_THINGS];   // uint64 [W][H] per multi-cell thing; 0: none
  uint32_t bytes;
};

struct FastDiv {                 // exact n / d for 32-bit n (Granlund & Montgomery 1994, fig. 4.1)
  uint32_t m, sh1, sh2, d;
  __device__ __forceinline__ uint32_t div(uint32_t n) const {
    const uint32_t hi = __umulhi(m, n);
    return (((n - hi) >> sh1) + hi) >> sh2;
  }
};
static FastDiv make_div(uint32_t d) {
  FastDiv f;
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;
  f.m = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  f.sh1 = l < 1 ? l : 1;
  f.sh2 = l > 0 ? l - 1 : 0;
  f.d = d;
  return f;
}

struct ShapeSplitParams {
  int32_t rows, cols, n_layers, n_things, first_drape, n_trail;
  uint32_t R, slab_bytes, shift_base, shift_slab, n_slots;   // n_slots = B * L * H
  FastDiv by_w, by_lh, by_h;
  // per thing (z-order): layer | visible << 8 | one_cell << 9 | art row << 16 | art col << 24
  uint32_t thing[CAMPX_SHAPE_MAX_THINGS];
  uint32_t trail_z[CAMPX_SHAPE_MAX_THINGS];    // the visible sprites before the first drape
  // the visible things at / in front of the first drape, FRONT TO BACK, for the render pass:
  // front[k] = thing word | z << 12 (bits 12-14); its row words (one-cell things: static_rows,
  // never used).  n_front of them.
  int32_t n_front;
  uint32_t front[CAMPX_SHAPE_MAX_THINGS];
  const uint64_t* front_rows[CAMPX_SHAPE_MAX_THINGS];
  FastDiv by_sh;                               // (update pass: word index / (S * H))
  const uint64_t* static_rows;                 // device: [L][H]
  const uint64_t* rowbits[CAMPX_SHAPE_MAX_THINGS];   // device: [W][H], or null
  const uint32_t* trace;                       // [4][T][B] offsets (bit 7 of orow0: rebuilt)
  const uint64_t* keys;                        // [ceil(T / key)][B][S][H]
  int64_t B, plane;                            // plane = T * B
  uint32_t max_pairs;                          // (environment, row) pairs a render window can touch
};

__device__ __forceinline__ uint32_t wrap_add(uint32_t a, uint32_t d, uint32_t n) {
  const uint32_t t = a + d;
  return t >= n ? t - n : t;
}

// The update pass: one lane per environment, 64 environments per workgroup (their trail words in
// LDS: [S * H][64] uint64, 8 * S * H * 64 bytes of dynamic shared memory per wave).
// A workgroup is blockDim.x / 64 REPLICAS of that wave (round 5, late): every replica walks the
// whole chain of its 64 environments - the offsets, the trail words in an LDS copy of its own -
// and writes only ITS share of what the pass produces: replica r the frames and the keyframe of
// the key intervals k with k % replicas == r; replica 0 the state at the end.  What a frame costs
// a wave (Hello World, B = 4 096, T = 100, one replica: 78 us) is mostly what it WRITES - the
// keyframes 28 us, the trace and scalar rows 13 us, the trail painting 11 us, everything else
// 26 us - and a second wave that repeats the cheap part to take half of the expensive one is the
// whole hand-over: no ring, no barrier, no flag (profiles/r05_shape_rocprofv3.txt).
// KS: the number of trail sprites when it is 0, 1 or 2 (loops unrolled, per-sprite constants in
// registers), -1: whatever pp.n_trail says.
constexpr int kShapeMaxReplicas = 4;
template <int KS>
__global__ __launch_bounds__(kWave * kShapeMaxReplicas) void shape_update_split_kernel(
    ShapeParams sp, ShapeSplitParams pp, const CampxShapeSpec* __restrict__ spec, CampxState st,
    uint64_t* __restrict__ state_words, const int8_t* __restrict__ actions, CampxOutputs out,
    uint32_t* __restrict__ trace, uint64_t* __restrict__ keys, int64_t B, int32_t T, int32_t reset_first) {
  // this wave's 64 environments' trail words, laid out as the keyframes are: [64][S][H]
  extern __shared__ __attribute__((aligned(16))) uint64_t all_trails[];
  __shared__ ShapeAction act[CAMPX_N_ACTIONS];   // (indexed by a lane's action: LDS, not kernarg)
  const int lane = threadIdx.x & (kWave - 1);
  const int replica = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), replicas = (int)(blockDim.x >> 6);
  if (threadIdx.x < CAMPX_N_ACTIONS) act[threadIdx.x] = sp.act[threadIdx.x];
  const int64_t env0 = (int64_t)blockIdx.x * kWave;
  const int64_t env = env0 + lane;
  const bool live = env < B;
  const int H = sp.rows, W = sp.cols, N = sp.n_things, S = KS >= 0 ? KS : pp.n_trail, SH = S * H;
  uint64_t* const trail = all_trails + (size_t)replica * SH * kWave;      // this replica's copy
  const int64_t n_words = (B - env0 < kWave ? B - env0 : (int64_t)kWave) * SH;
  // (the carried trail words: shape_words_from_backdrop_kernel made them from the backdrop state)
  for (int i = lane; i < SH * kWave; i += kWave)
    trail[i] = (!reset_first && i < n_words) ? state_words[env0 * SH + i] : 0ull;
  // (the carried state is loaded BEFORE the barrier: replica 0 writes the same words at the end of
  // a chunk, and with a chunk of one or two frames nothing else would order another replica's
  // loads before those stores - round 5 advice)
  uint32_t orow[2] = {0u, 0u}, ocol[2] = {0u, 0u};
  int over = 0, bad = 0;
  float ret = 0.0f;
  if (live && !reset_first) {
#pragma unroll
    for (int k = 0; k < CAMPX_SHAPE_MAX_THINGS; ++k)
      if (k < N) {
        orow[k >> 2] |= (uint32_t)(uint8_t)st.pos[(int64_t)(2 * k) * B + env] << (8 * (k & 3));
        ocol[k >> 2] |= (uint32_t)(uint8_t)st.pos[(int64_t)(2 * k + 1) * B + env] << (8 * (k & 3));
      }
    over = st.done[env];
    if (st.ret) ret = st.ret[env];
  }
  __syncthreads();
  uint32_t trail_pos[4] = {0u, 0u, 0u, 0u};   // sprite s: row | col << 8 in half s & 1 of word s >> 1
  // (the sprites' constants, fetched once: thing word and the byte of the offset words it owns)
  uint32_t tz[KS > 0 ? KS : 1], tth[KS > 0 ? KS : 1];
  if (KS > 0) {
#pragma unroll
    for (int s = 0; s < (KS > 0 ? KS : 0); ++s) {
      tz[s] = pp.trail_z[s];
      tth[s] = pp.thing[tz[s]];
    }
  }
  uint64_t* const my_trail = trail + lane * SH;
  auto paint_trails = [&]() {        // every trail sprite, back to front: mine, nobody else's
    trail_pos[0] = trail_pos[1] = trail_pos[2] = trail_pos[3] = 0u;
#pragma unroll
    for (int s = 0; s < (KS >= 0 ? KS : CAMPX_SHAPE_MAX_THINGS); ++s) {
      if (KS < 0 && s >= S) break;
      const uint32_t z = KS > 0 ? tz[s] : pp.trail_z[s], th = KS > 0 ? tth[s] : pp.thing[z];
      const int sh = 8 * (z & 3);
      const uint32_t r = wrap_add(th >> 16 & 0xffu, ((z < 4 ? orow[0] : orow[1]) >> sh) & 0xffu, (uint32_t)H);
      const uint32_t c = wrap_add(th >> 24, ((z < 4 ? ocol[0] : ocol[1]) >> sh) & 0xffu, (uint32_t)W);
      const uint64_t bit = 1ull << c;
      // (LDS atomics without a return value: the wave never waits for a word to come back - as
      // read-modify-write these four dependent round trips were most of a frame's 1.2 us)
#pragma unroll
      for (int q = 0; q < (KS >= 0 ? KS : CAMPX_SHAPE_MAX_THINGS); ++q) {
        if (KS < 0 && q >= S) break;
        uint64_t* w = my_trail + q * H + (int)r;
        if (q == s) __hip_atomic_fetch_or(w, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_and(w, ~bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      const uint32_t packed = (r | (c << 8)) << (16 * (s & 1));
      trail_pos[0] |= (s >> 1) == 0 ? packed : 0u;
      trail_pos[1] |= (s >> 1) == 1 ? packed : 0u;
      trail_pos[2] |= (s >> 1) == 2 ? packed : 0u;
      trail_pos[3] |= (s >> 1) == 3 ? packed : 0u;
    }
  };
  const int64_t plane = (int64_t)T * B;
  constexpr int kAhead = 16;                    // actions in flight per lane
  for (int t0 = 0; t0 < T; t0 += kAhead) {
    int a16[kAhead];
#pragma unroll
    for (int j = 0; j < kAhead; ++j) {
      const int t = t0 + j < T ? t0 + j : T - 1;
      a16[j] = live ? (int)actions[(int64_t)t * B + env] : 4;
    }
#pragma unroll 4
    for (int j = 0; j < kAhead; ++j) {
      const int t = t0 + j;
      if (t >= T) break;        // (uniform)
      const int a = a16[j];
      const bool valid = (unsigned)a < (unsigned)CAMPX_N_ACTIONS;
      bad += (valid || !live) ? 0 : 1;
      uint32_t rebuilt = 0u;
      if (over) {   // a fresh make_game() + its_showtime()
        orow[0] = orow[1] = ocol[0] = ocol[1] = 0u;
        over = 0;
        ret = 0.0f;
        rebuilt = 0x80u;
        for (int i = 0; i < SH; ++i) my_trail[i] = 0ull;
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
      if (S > 0) paint_trails();
      // (this replica's share: the frames and the keyframe of every `replicas`-th key interval)
      const bool mine = (t / kShapeKey) % replicas == replica;
      if (live && mine) {
        const int64_t at = (int64_t)t * B + env;
        trace[at] = orow[0] | rebuilt;
        trace[plane + at] = orow[1];
        trace[2 * plane + at] = ocol[0];
        trace[3 * plane + at] = ocol[1];
        // (where the trail sprites stand after this frame, decoded: the render pass replays them)
        if (S > 0) trace[4 * plane + at] = trail_pos[0];
        if (S > 2) trace[5 * plane + at] = trail_pos[1];
        if (S > 4) trace[6 * plane + at] = trail_pos[2];
        if (S > 6) trace[7 * plane + at] = trail_pos[3];
        if (out.reward) out.reward[at] = reward;
        if (out.discount) out.discount[at] = over ? 0.0f : 1.0f;
        if (out.done) out.done[at] = (uint8_t)over;
      }
      if (S > 0 && t % kShapeKey == 0 && mine) {
        // keyframe: this wave's 64 environments x S * H words, contiguous in [key][B][S][H]
        uint64_t* to = keys + ((int64_t)(t / kShapeKey) * B + env0) * SH;
        for (int i0 = 0; i0 < n_words; i0 += 8 * kWave) {       // (eight LDS reads in flight)
          uint64_t v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int i = i0 + j * kWave + lane;
            v[j] = trail[i < n_words ? i : 0];
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int i = i0 + j * kWave + lane;
            if (i < n_words) to[i] = v[j];
          }
        }
      }
    }
  }
  if (replica != 0) return;         // (every replica ends in the same state: one writes it down)
  if (live) {
#pragma unroll
    for (int k = 0; k < CAMPX_SHAPE_MAX_THINGS; ++k)
      if (k < N) {
        st.pos[(int64_t)(2 * k) * B + env] = (int8_t)((orow[k >> 2] >> (8 * (k & 3))) & 0xffu);
        st.pos[(int64_t)(2 * k + 1) * B + env] = (int8_t)((ocol[k >> 2] >> (8 * (k & 3))) & 0xffu);
      }
    st.done[env] = (uint8_t)over;
    if (st.ret) st.ret[env] = ret;
  }
  if (S > 0)        // the trail words after the last frame: shape_backdrop_from_words_kernel reads them
    for (int i = lane; i < n_words; i += kWave) state_words[env0 * SH + i] = trail[i];
  report_bad_actions(out, bad);
}

// The carried state of the two shape kernels is the per-environment backdrop, a layer per cell
// (campx_shape_rollout_launch's `backdrop_state`); the frame-major path's update pass works on
// trail WORDS.  Two small, fully parallel kernels convert at the ends of a launch (inside the
// update pass, one lane walking its environment's 468 cells, they cost 85 us per launch).
// backdrop -> words: a cell that differs from the art's backdrop was painted by the trail
// sprite of that layer; one thread per (environment, sprite, row).
__global__ __launch_bounds__(256) void shape_words_from_backdrop_kernel(
    ShapeSplitParams pp, const CampxShapeSpec* __restrict__ spec, const int8_t* __restrict__ backdrop_state,
    uint64_t* __restrict__ words, int64_t B) {
  const int H = pp.rows, W = pp.cols, S = pp.n_trail;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * S * H) return;
  const int r = (int)(i % H), s = (int)((i / H) % S);
  const int64_t env = i / ((int64_t)S * H);
  const uint32_t layer = pp.thing[pp.trail_z[s]] & 0xffu;
  const uint8_t* mine = reinterpret_cast<const uint8_t*>(backdrop_state) + env * (int64_t)(H * W) + r * W;
  const uint8_t* art = spec->backdrop + r * W;
  uint64_t w = 0ull;
  for (int c = 0; c < W; ++c) w |= (mine[c] != art[c] && mine[c] == layer) ? 1ull << c : 0ull;
  words[i] = w;
}

// words -> backdrop: one thread per (environment, row).  A wave's 64 rows are 64 * W contiguous,
// 16-byte aligned bytes of the state: each lane paints its row into LDS, then the wave streams
// the block out in whole 16-byte chunks (round 6: the lanes used to store their 36 bytes one by one,
// 25 us per launch at B = 32 768 - 1.5 % of a Hello World rollout - for 15 MB).
__global__ __launch_bounds__(256) void shape_backdrop_from_words_kernel(
    ShapeSplitParams pp, const CampxShapeSpec* __restrict__ spec, const uint64_t* __restrict__ words,
    int8_t* __restrict__ backdrop_state, int64_t B) {
  __shared__ __attribute__((aligned(16))) int8_t stage_all[4][kWave * 128];     // (W <= 127)
  const int H = pp.rows, W = pp.cols, S = pp.n_trail;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int8_t* stage = stage_all[wave];
  const int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * kWave;          // the wave's first (env * H + r)
  const int64_t n_rows = B * H;
  const int64_t i = i0 + lane;
  if (i < n_rows) {
    const int r = (int)(i % H);
    const int64_t env = i / H;
    uint64_t tw[CAMPX_SHAPE_MAX_THINGS];
    uint32_t layer[CAMPX_SHAPE_MAX_THINGS];
#pragma unroll
    for (int s = 0; s < CAMPX_SHAPE_MAX_THINGS; ++s) {
      tw[s] = s < S ? words[(env * S + s) * H + r] : 0ull;
      layer[s] = s < S ? pp.thing[pp.trail_z[s]] & 0xffu : 0u;
    }
    const uint8_t* art = spec->backdrop + r * W;
    int8_t* mine = stage + lane * W;
    for (int c = 0; c < W; ++c) {
      uint32_t v = art[c];
#pragma unroll
      for (int s = 0; s < CAMPX_SHAPE_MAX_THINGS; ++s) v = ((tw[s] >> c) & 1ull) ? layer[s] : v;
      mine[c] = (int8_t)v;
    }
  }
  __syncthreads();
  if (i0 >= n_rows) return;
  const int64_t valid = n_rows - i0 < kWave ? n_rows - i0 : (int64_t)kWave;
  const int n_bytes = (int)valid * W;
  int8_t* out = backdrop_state + i0 * W;                                // 64 * W * k: 16-byte aligned
  for (int at = lane * 16; at < n_bytes; at += kWave * 16) {
    if (at + 16 <= n_bytes) {
      *reinterpret_cast<u32x4*>(out + at) = *reinterpret_cast<const u32x4*>(stage + at);
    } else {
      for (int b = at; b < n_bytes; ++b) out[b] = stage[b];             // the state's last bytes
    }
  }
}

// 16 bits -> 16 bytes 0 / 1 (bit i -> byte i)
__device__ __forceinline__ u32x4 bits_to_bytes(uint32_t b) {
  auto four = [](uint32_t n) { return __umul24(n & 0xfu, 0x00204081u) & 0x01010101u; };
  return u32x4{four(b), four(b >> 4), four(b >> 8), four(b >> 12)};
}

// NF: visible things at / in front of the first drape (1..8); NS: trail sprites, rounded up to
// 0 / 2 / 8 (surplus entries repeat the last sprite: painting twice changes nothing).
//
// A block = ONE one-shot wave = kSplitWin consecutive KiB of one frame, aligned in memory.
//   stage A  one lane per (environment, board row) the window touches - a few dozen: 8 KiB of
//            Hello World are 2.5 environments of 13 rows, one pass - : everything that does not
//            depend on the layer (the things' row words, what covers what, the trail words
//            brought up to this frame) and from it the finished row of EVERY layer, into LDS:
//            done once per (environment, row), not once per (environment, layer, row);
//   stage B  one lane per slot of the window: the rows in the order the stream has them;
//   stage C  one lane per 16-byte chunk: 16 bits of two neighbouring slots -> 16 bytes -> one
//            aligned store; a KiB per instruction, kSplitWin instructions.
// Every memory trip of stage A is issued as ONE batch: first the offsets of its frame (and of
// the frames since the keyframe), then - their addresses depend on those - every thing's row
// word and the keyframe's trail words.  (History, profiles/r05_shape_rocprofv3.txt: a loop over
// the things paid one trip per thing: 3.3 TB/s; batched, every slot-lane recomputing its row's
// things, 2 KiB per wave: VALU-bound at 4.4 TB/s, 3.0 with trails; stage A by one wave of a
// four-wave block behind a barrier: fewer instructions and slower, 3.1 / 2.7 - three waves idle
// for two memory trips.)
// LDS (dynamic): rows [L][pairs] uint64, then slots [kSplitSpan / W + 8].
template <int NF, int NS>
__global__ __launch_bounds__(kWave) void shape_render_split_kernel(
    ShapeSplitParams pp, int8_t* __restrict__ dst) {
  extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
  const uint32_t lane = threadIdx.x;
  const uint32_t t = blockIdx.y;
  uint32_t bx = blockIdx.x;
  bx = (bx & 7u) * (gridDim.x >> 3) + (bx >> 3);     // gridDim.x is a multiple of 8: one XCD, one eighth
  const uint32_t shift = (pp.shift_base + t * pp.shift_slab) & (kSplitSpan - 1u);
  const uint32_t H = (uint32_t)pp.rows, W = (uint32_t)pp.cols, L = (uint32_t)pp.n_layers;
  const uint32_t LH = L * H;
  if ((uint64_t)bx * kSplitSpan >= (uint64_t)pp.slab_bytes + shift) return;
  const uint32_t woff0 = bx * kSplitSpan - shift;
  const uint32_t wlo = bx * kSplitSpan < shift ? 0u : woff0;
  const uint32_t wend = (woff0 + kSplitSpan - 1u < pp.slab_bytes) ? woff0 + kSplitSpan - 1u : pp.slab_bytes - 1u;
  const uint32_t s_first = pp.by_w.div(wlo);
  uint32_t s_last = pp.by_w.div(wend) + 1u;                    // (a chunk's tail reaches into the next slot)
  s_last = s_last < pp.n_slots ? s_last : pp.n_slots - 1u;
  const uint32_t env_first = pp.by_lh.div(s_first);
  const uint32_t env_last = pp.by_lh.div(s_last);
  const uint32_t n_pairs = (env_last - env_first + 1u) * H;
  const uint32_t P = pp.max_pairs;                   // row pitch of `rows`
  uint64_t* rows = lds64;                            // [L][P]
  uint64_t* slots = lds64 + L * P;
  const uint32_t* frame_trace = pp.trace + (int64_t)t * pp.B;
  const uint32_t key_frame = t - t % (uint32_t)kShapeKey;
  const bool upper = pp.n_things > 4;            // (uniform) offsets of things 4..7 are in use

  // ---- stage A
  for (uint32_t p0 = 0; p0 < n_pairs; p0 += kWave) {
    uint32_t p = p0 + lane;
    const bool live = p < n_pairs;
    p = live ? p : n_pairs - 1u;
    const uint32_t ei = pp.by_h.div(p);
    const uint32_t r = p - ei * H;
    const uint32_t env = env_first + ei;
    // trip 1: the offsets of this frame, and of the frames since the keyframe
    uint32_t off_r[2], off_c[2];
    off_r[0] = frame_trace[env];
    off_c[0] = frame_trace[2 * pp.plane + env];
    off_r[1] = upper ? frame_trace[pp.plane + env] : 0u;
    off_c[1] = upper ? frame_trace[3 * pp.plane + env] : 0u;
    constexpr int kPosWords = (NS + 1) / 2 > 0 ? (NS + 1) / 2 : 1;
    uint32_t ev_flag[kShapeKey - 1], ev_pos[kShapeKey - 1][kPosWords];
    uint64_t tw[NS > 0 ? NS : 1];
    const int S = pp.n_trail;
    if (NS > 0) {
#pragma unroll
      for (int i = 0; i < kShapeKey - 1; ++i) {
        // frame key_frame + 1 + i, clamped to t: a frame applied twice changes nothing
        const uint32_t f = key_frame + 1u + (uint32_t)i <= t ? key_frame + 1u + (uint32_t)i : t;
        const uint32_t* ft = pp.trace + (int64_t)f * pp.B;
        ev_flag[i] = ft[env];                       // (bit 7: the frame began with a rebuild)
#pragma unroll
        for (int j = 0; j < kPosWords; ++j)
          ev_pos[i][j] = 2 * j < S ? ft[(4 + j) * pp.plane + env] : 0u;
      }
      // (the keyframe's trail words: their address does not depend on the offsets - same trip)
      const uint64_t* key = pp.keys + ((int64_t)(key_frame / (uint32_t)kShapeKey) * pp.B + env) * (S * (int)H);
#pragma unroll
      for (int s = 0; s < NS; ++s) tw[s] = key[(s < S ? s : S - 1) * (int)H + (int)r];
    }
    // trip 2: every thing's row word
    uint64_t wd[NF];
    uint32_t one_r[NF], one_c[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) {
      const uint32_t th = pp.front[k];
      const uint32_t z = (th >> 12) & 7u;
      const uint32_t sh = 8u * (z & 3u);
      const uint32_t dr = ((off_r[z >> 2] & ~0x80u) >> sh) & 0xffu, dc = (off_c[z >> 2] >> sh) & 0xffu;
      const uint32_t src = r >= dr ? r - dr : r + H - dr;
      // (one-cell things load a word they never use: no branch between the loads)
      const uint32_t idx = ((th >> 9) & 1u) ? 0u : dc * H + src;
      wd[k] = pp.front_rows[k][idx];
      one_r[k] = wrap_add(th >> 16 & 0xffu, dr, H);
      one_c[k] = wrap_add(th >> 24, dc, W);
    }
    // the trail words, brought from the keyframe up to this frame (under trip 2)
    uint64_t any = 0ull;
    if (NS > 0) {
#pragma unroll
      for (int i = 0; i < kShapeKey - 1; ++i) {
        if (ev_flag[i] & 0x80u) {             // the frame began with a rebuild: trails restart
#pragma unroll
          for (int s = 0; s < NS; ++s) tw[s] = 0ull;
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const int ss = s < S ? s : S - 1;
          const uint32_t pos = ev_pos[i][s >> 1] >> (16 * (s & 1));
          // (a surplus entry s >= S repeats sprite S - 1: same words, same position)
          const uint32_t pos_ss = s < S ? pos : ev_pos[i][(S - 1) >> 1] >> (16 * ((S - 1) & 1));
          const uint32_t rr = pos_ss & 0xffu, cc = (pos_ss >> 8) & 0xffu;
          const uint64_t bit = rr == r ? 1ull << cc : 0ull;
#pragma unroll
          for (int q = 0; q < NS; ++q) {
            const int qq = q < S ? q : S - 1;
            tw[q] = qq == ss ? (tw[q] | bit) : (tw[q] & ~bit);
          }
        }
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) any |= tw[s];
    }
    // the things, front to back: what each shows of this row, what is covered
    uint64_t covered = 0ull, vis[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) {
      const uint32_t th = pp.front[k];
      const uint64_t w = ((th >> 9) & 1u) ? (one_r[k] == r ? 1ull << one_c[k] : 0ull) : wd[k];
      vis[k] = w & ~covered;
      covered |= w;
    }
    // the finished row of every layer (the layer index is uniform: scalar selects)
    const uint64_t open = ~(covered | any);          // where the art's backdrop shows
    for (uint32_t l = 0; l < L; ++l) {
      uint64_t row = pp.static_rows[l * H + r] & open;
#pragma unroll
      for (int k = 0; k < NF; ++k) row |= ((pp.front[k] & 0xffu) == l) ? vis[k] : 0ull;
      if (NS > 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const int ss = s < S ? s : S - 1;
          row |= ((pp.thing[pp.trail_z[ss]] & 0xffu) == l) ? (tw[s] & ~covered) : 0ull;
        }
      }
      if (live) rows[l * P + p] = row;
    }
  }
  // ---- stage B: the window's slots in stream order (one wave: LDS operations complete in order)
  for (uint32_t base = s_first; base <= s_last; base += kWave) {
    uint32_t g = base + lane;
    g = g <= s_last ? g : s_last;                 // (surplus lanes repeat the last slot)
    const uint32_t env = pp.by_lh.div(g);
    const uint32_t rem = g - env * LH;
    const uint32_t l = pp.by_h.div(rem);
    const uint32_t r = rem - l * H;
    slots[base - s_first + lane] = rows[l * P + (env - env_first) * H + r];
  }
  // ---- stage C: every lane's 16 bytes are 16 bits of two neighbouring slots
  int8_t* frame = dst + (int64_t)t * pp.slab_bytes;
#pragma unroll
  for (int j = 0; j < kSplitWin; ++j) {
    const uint32_t off = woff0 + (uint32_t)j * 1024u + lane * 16u;
    const bool inside = off < pp.slab_bytes;          // (bytes before the frame wrap to huge offsets)
    const uint32_t o = inside ? off : wlo;
    const uint32_t s = pp.by_w.div(o);
    const uint32_t c = o - s * W;
    const uint32_t idx = s - s_first;
    const uint64_t lo = slots[idx], hi = slots[idx + 1u];
    const uint32_t bits = (uint32_t)(lo >> c) | (uint32_t)((hi << 1) << (W - 1u - c));
    if (inside) store16_streaming_at(frame, off, bits_to_bytes(bits));
  }
}

static inline uint32_t align8(uint32_t x) { return (x + 7u) & ~7u; }

// Which games take the frame-major path: rows of 16 to 64 cells (a 16-byte chunk then spans at
// most two slots, a slot is one 64-bit word), every trail sprite's words fitting the update
// pass's LDS.
static uint32_t shape_max_pairs(const CampxShapeSpec& s) {
  const int64_t R = (int64_t)s.rows * s.cols * s.n_layers;
  const int64_t envs = ((int64_t)kSplitSpan + (int64_t)s.cols) / R + 2;
  return (uint32_t)(envs * s.rows);
}
// dynamic LDS of a render wave: rows [L][pairs], slots [span / W + 8], 8 bytes each
static size_t shape_render_lds(const CampxShapeSpec& s) {
  return 8u * ((size_t)s.n_layers * shape_max_pairs(s) + (size_t)kSplitSpan / (size_t)s.cols + 72u);
}
static bool shape_tables_ok(const CampxShapeSpec& s) {
  if (s.cols < 16 || s.cols > 64) return false;
  int n_trail = 0;
  for (int k = 0; k < s.first_drape; ++k) n_trail += s.things[k].visible ? 1 : 0;
  if (n_trail * s.rows > 120) return false;          // 8 * S * H * 64 bytes of LDS per update wave
  if (shape_render_lds(s) > 24 * 1024) return false;  // (a render wave's rows and slots)
  for (int k = 0; k < s.first_drape; ++k)
    if (s.things[k].visible && s.things[k].n_cells != 1) return false;   // (sprites are one cell)
  return true;
}

int64_t shape_tables_bytes(const CampxShapeSpec& s) {
  if (!shape_tables_ok(s)) return 0;
  int64_t bytes = align8((uint32_t)sizeof(ShapeTablesHeader)) + 8ll * s.n_layers * s.rows;
  for (int k = s.first_drape; k < s.n_things; ++k)
    if (s.things[k].visible && s.things[k].n_cells > 1) bytes += 8ll * s.cols * s.rows;
  return bytes;
}

int32_t shape_tables_build(const CampxShapeSpec& s, void* host, int64_t bytes) {
  if (bytes < shape_tables_bytes(s) || !shape_tables_ok(s)) return CAMPX_EINVAL;
  memset(host, 0, (size_t)bytes);
  ShapeTablesHeader* h = static_cast<ShapeTablesHeader*>(host);
  char* blob = static_cast<char*>(host);
  const int H = s.rows, W = s.cols;
  h->magic = kShapeTablesMagic;
  h->rows = H;
  h->cols = W;
  h->n_layers = s.n_layers;
  h->n_things = s.n_things;
  h->first_drape = s.first_drape;
  uint32_t at = align8((uint32_t)sizeof(ShapeTablesHeader));
  h->static_off = at;
  uint64_t* st = reinterpret_cast<uint64_t*>(blob + at);
  for (int r = 0; r < H; ++r)
    for (int c = 0; c < W; ++c) st[s.backdrop[r * W + c] * H + r] |= 1ull << c;
  at += 8u * (uint32_t)(s.n_layers * H);
  for (int k = s.first_drape; k < s.n_things; ++k) {
    const CampxShapeThing& th = s.things[k];
    if (!th.visible || th.n_cells <= 1) continue;
    h->rowbits_off[k] = at;
    uint64_t* rb = reinterpret_cast<uint64_t*>(blob + at);     // [dc][row]: the mask's row, rotated by dc
    for (int i = 0; i < th.n_cells; ++i) {
      const int r = s.cells[th.cell_begin + i] >> 8, c = s.cells[th.cell_begin + i] & 0xff;
      for (int dc = 0; dc < W; ++dc) rb[dc * H + r] |= 1ull << ((c + dc) % W);
    }
    at += 8u * (uint32_t)(W * H);
  }
  h->bytes = at;
  return CAMPX_OK;
}

static int shape_n_trail(const CampxShapeSpec& s) {
  int n = 0;
  for (int k = 0; k < s.first_drape; ++k) n += s.things[k].visible ? 1 : 0;
  return n;
}

// bytes of the offset trace: uint32 [4 + ceil(S / 2)][T][B] - the things' offsets, then where the
// trail sprites stand (row | col << 8, two sprites per word)
static int64_t shape_offsets_bytes(const CampxShapeSpec& s, int64_t B, int32_t T) {
  return ((4ll + (shape_n_trail(s) + 1) / 2) * 4 * T * B + 7) & ~7ll;
}

// Frames per chunk of a frame-major launch of T frames of B environments (launch_shape_split).
static int32_t shape_chunk_frames(int64_t B, int32_t T) {
  int64_t chunk = knob(K_SHAPE_CHUNK_KF) * 1000 / B;
  chunk = chunk / kShapeKey * kShapeKey;
  chunk = chunk < kShapeKey ? kShapeKey : chunk;
  return (int32_t)(chunk < T ? chunk : T);
}

// Scratch of the frame-major path (CampxOutputs.trace), sized for ONE chunk of frames (every
// chunk of a launch reuses it): the offset trace, then
// (8-byte aligned) the keyframes uint64 [ceil(T / key)][B][S][H], then the trail words at the
// ends of the launch uint64 [B][S][H].
int64_t shape_scratch_bytes(const CampxShapeSpec& s, int64_t B, int32_t T) {
  if (!shape_tables_ok(s) || B <= 0 || T <= 0) return 0;
  const int32_t Tc = shape_chunk_frames(B, T);
  const int64_t offsets = shape_offsets_bytes(s, B, Tc);
  const int64_t keys = 8ll * ((Tc + kShapeKey - 1) / kShapeKey + 1) * B * shape_n_trail(s) * s.rows;
  return offsets + keys;
}

// Can this launch take the frame-major path?  Every frame kept back to back, int8, no flat
// board, frames of whole 16-byte chunks below 4 GiB, the tables and the scratch given.
bool shape_split_ok(const CampxShapeSpec& s, const void* tables, const CampxOutputs& out, int64_t B,
                    int32_t T, int32_t emit_first) {
  if (!knob(K_SHAPE_SPLIT) || !tables || !out.trace || out.board || emit_first || T < 1 || T > 65535) return false;
  if (!shape_tables_ok(s) || out.obs_format != CAMPX_OBS_INT8) return false;
  const int64_t R = (int64_t)s.rows * s.cols * s.n_layers;
  if ((B * R) % 16 || B * R >= (1ll << 32) - 65536 || (int64_t)T * B >= (1ll << 31)) return false;
  if (reinterpret_cast<uintptr_t>(out.obs) & 15) return false;
  if (out.scalar_pitch && out.scalar_pitch != B) return false;
  return out.obs_t_stride == B * R;
}

int32_t launch_shape_split(const ShapeParams& sp, const CampxShapeSpec& s, const CampxShapeSpec* spec_dev,
                           const void* tables_dev, CampxState st, int8_t* backdrop_state,
                           const int8_t* actions, CampxOutputs out, int64_t B, int32_t T,
                           int32_t reset_first, hipStream_t stream) {
  ShapeSplitParams pp;
  memset(&pp, 0, sizeof(pp));
  const int H = s.rows, W = s.cols, L = s.n_layers;
  pp.rows = H;
  pp.cols = W;
  pp.n_layers = L;
  pp.n_things = s.n_things;
  pp.first_drape = s.first_drape;
  pp.R = (uint32_t)(L * H * W);
  pp.slab_bytes = (uint32_t)(B * pp.R);
  pp.shift_base = (uint32_t)(reinterpret_cast<uintptr_t>(out.obs) & (kSplitSpan - 1u));
  pp.shift_slab = pp.slab_bytes & (kSplitSpan - 1u);
  pp.n_slots = (uint32_t)(B * L * H);
  pp.by_w = make_div((uint32_t)W);
  pp.by_lh = make_div((uint32_t)(L * H));
  pp.by_h = make_div((uint32_t)H);
  // the tables blob: the host builds the same layout (shape_tables_build), so the offsets are known
  const char* blob = static_cast<const char*>(tables_dev);
  uint32_t at = align8((uint32_t)sizeof(ShapeTablesHeader));
  pp.static_rows = reinterpret_cast<const uint64_t*>(blob + at);
  at += 8u * (uint32_t)(L * H);
  for (int k = 0; k < s.n_things; ++k) {
    const CampxShapeThing& th = s.things[k];
    uint32_t art = 0u;
    if (th.n_cells >= 1) art = ((uint32_t)(s.cells[th.cell_begin] >> 8) << 16) | ((uint32_t)(s.cells[th.cell_begin] & 0xff) << 24);
    pp.thing[k] = (uint32_t)th.layer | ((th.visible && th.n_cells > 0 ? 1u : 0u) << 8) |
                  ((th.n_cells == 1 ? 1u : 0u) << 9) | art;
    if (k >= s.first_drape && th.visible && th.n_cells > 1) {
      pp.rowbits[k] = reinterpret_cast<const uint64_t*>(blob + at);
      at += 8u * (uint32_t)(W * H);
    }
    if (k < s.first_drape && th.visible) pp.trail_z[pp.n_trail++] = (uint32_t)k;
  }
  for (int k = s.n_things - 1; k >= s.first_drape; --k) {
    if (!((pp.thing[k] >> 8) & 1u)) continue;
    pp.front[pp.n_front] = pp.thing[k] | ((uint32_t)k << 12);
    pp.front_rows[pp.n_front] = pp.rowbits[k] ? pp.rowbits[k] : pp.static_rows;
    ++pp.n_front;
  }
  pp.by_sh = make_div((uint32_t)(pp.n_trail * H > 0 ? pp.n_trail * H : 1));
  pp.B = B;
  pp.max_pairs = shape_max_pairs(s);
  // replicas of an update wave (see the kernel): 4, fewer where four copies of the trail words do
  // not fit the 64 KiB a workgroup may ask for
  constexpr int want_replicas = kShapeMaxReplicas;
  const size_t lds_one = (size_t)8 * pp.n_trail * H * kWave;
  // (Hello World, T = 100, of peak, 1 / 2 / 4 replicas: B = 4 096 0.507 / 0.533 / 0.545, 16 384 0.745 /
  // 0.757 / 0.765, 32 768 0.801 / 0.802 / 0.808, 65 536 0.830 / 0.833 / 0.79-0.81: past 512 workgroups
  // four waves each cost more than they take off the chain - two there)
  int replicas = (B + kWave - 1) / kWave > 512 && want_replicas > 2 ? 2 : want_replicas;
  while (replicas > 1 && lds_one * replicas > 60 * 1024) --replicas;
  const size_t lds = lds_one * replicas;
  const size_t render_lds = shape_render_lds(s);
  // the trail words at the ends of the launch (and of every chunk) sit behind a full chunk's keyframes
  const int32_t chunk = shape_chunk_frames(B, T);
  uint64_t* state_words = reinterpret_cast<uint64_t*>(out.trace + shape_offsets_bytes(s, B, chunk)) +
                          (int64_t)((chunk + kShapeKey - 1) / kShapeKey) * B * pp.n_trail * H;
  const bool carried = pp.n_trail > 0 && backdrop_state != nullptr;
  if (carried && !reset_first) {
    const int64_t n = B * pp.n_trail * H;
    hipLaunchKernelGGL(shape_words_from_backdrop_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       pp, spec_dev, backdrop_state, state_words, B);
  }
  // Long launches run as CHUNKS of frames, update pass and render alternating, like the one-cell
  // tier's (campx_api.hip launch_split): a render wave gathers two dozen small pieces of the offset
  // trace and the keyframes, which come from the 256 MB memory-side cache while a chunk's scratch
  // fits it and from HBM, piece by piece, when it does not - Hello World, render kernel alone:
  // B x T = 1.6 M environment-frames 5.8 TB/s, 3.3 M 5.4, 6.6 M 3.9 (the trail-free art, which
  // reads a third of that, holds 6.7).  A chunk is at most CAMPX_SHAPE_CHUNK_KF thousand
  // environment-frames (2 000; bench.py --game hello_world, of peak, whole / 700 / 1 000 / 1 400 /
  // 2 000 / 2 800: B = 32 768 0.70 / 0.77 / 0.79 / 0.76 / 0.81 / 0.80, B = 65 536 0.47 / 0.76 / 0.79 /
  // 0.83 / 0.83 / 0.83), a multiple of the key interval.
  const uint64_t reach = (uint64_t)pp.slab_bytes + ((pp.shift_base | pp.shift_slab) ? kSplitSpan - 1u : 0u);
  const uint64_t block_span = (uint64_t)kSplitSpan * kSplitWaves;
  const unsigned grid_x = (unsigned)((((reach + block_span - 1) / block_span) + 7u) & ~(uint64_t)7);
  for (int64_t t0 = 0; t0 < T; t0 += chunk) {
    const int32_t n = (int32_t)(T - t0 < chunk ? T - t0 : (int64_t)chunk);
    CampxOutputs part = out;
    part.obs = out.obs + t0 * out.obs_t_stride;
    if (out.reward) part.reward = out.reward + t0 * B;
    if (out.discount) part.discount = out.discount + t0 * B;
    if (out.done) part.done = out.done + t0 * B;
    uint32_t* trace = reinterpret_cast<uint32_t*>(out.trace);
    uint64_t* keys = reinterpret_cast<uint64_t*>(out.trace + shape_offsets_bytes(s, B, n));
    pp.trace = trace;
    pp.keys = keys;
    pp.plane = (int64_t)n * B;
    // (windows are aligned in memory: the alignment of THIS chunk's first frame)
    pp.shift_base = (uint32_t)(reinterpret_cast<uintptr_t>(part.obs) & (kSplitSpan - 1u));
    const int32_t fresh = t0 == 0 ? reset_first : 0;
#define CAMPX_SHAPE_UPDATE(KS)                                                                              \
  hipLaunchKernelGGL(shape_update_split_kernel<KS>, dim3((unsigned)((B + kWave - 1) / kWave)), dim3(kWave * replicas), lds, \
                     stream, sp, pp, spec_dev, st, state_words, actions + t0 * B, part, trace, keys, B, n, fresh)
    if (pp.n_trail == 0) CAMPX_SHAPE_UPDATE(0);
    else if (pp.n_trail == 1) CAMPX_SHAPE_UPDATE(1);
    else if (pp.n_trail == 2) CAMPX_SHAPE_UPDATE(2);
    else CAMPX_SHAPE_UPDATE(-1);
#undef CAMPX_SHAPE_UPDATE
    const dim3 grid(grid_x, (unsigned)n);
#define CAMPX_SHAPE_RENDER(NF, NS) \
  hipLaunchKernelGGL((shape_render_split_kernel<NF, NS>), grid, dim3(kWave), render_lds, stream, pp, part.obs)
#define CAMPX_SHAPE_RENDER_NS(NF)                     \
  do {                                                \
    if (pp.n_trail == 0) CAMPX_SHAPE_RENDER(NF, 0);   \
    else if (pp.n_trail <= 2) CAMPX_SHAPE_RENDER(NF, 2); \
    else CAMPX_SHAPE_RENDER(NF, 8);                   \
  } while (0)
    switch (pp.n_front) {
      case 1: CAMPX_SHAPE_RENDER_NS(1); break;
      case 2: CAMPX_SHAPE_RENDER_NS(2); break;
      case 3: CAMPX_SHAPE_RENDER_NS(3); break;
      case 4: CAMPX_SHAPE_RENDER_NS(4); break;
      case 5: CAMPX_SHAPE_RENDER_NS(5); break;
      case 6: CAMPX_SHAPE_RENDER_NS(6); break;
      case 7: CAMPX_SHAPE_RENDER_NS(7); break;
      default: CAMPX_SHAPE_RENDER_NS(8); break;
    }
#undef CAMPX_SHAPE_RENDER_NS
#undef CAMPX_SHAPE_RENDER
  }
  if (carried) {
    const int64_t n = B * H;
    hipLaunchKernelGGL(shape_backdrop_from_words_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       pp, spec_dev, state_words, backdrop_state, B);     // (a wave: 64 rows, staged in LDS)
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? CAMPX_OK : hip_failed(e);
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

int64_t campx_shape_tables_bytes(const CampxShapeSpec* spec_host) {
  if (!spec_host || campx_shape_spec_validate(spec_host) != CAMPX_OK) return 0;
  return shape_tables_bytes(*spec_host);
}

int32_t campx_shape_tables_build(const CampxShapeSpec* spec_host, void* tables_host, int64_t bytes) {
  if (!spec_host || !tables_host) return CAMPX_EINVAL;
  const int32_t v = campx_shape_spec_validate(spec_host);
  if (v != CAMPX_OK) return v;
  return shape_tables_build(*spec_host, tables_host, bytes);
}

int64_t campx_shape_scratch_bytes(const CampxShapeSpec* spec_host, int64_t B, int32_t T) {
  if (!spec_host || campx_shape_spec_validate(spec_host) != CAMPX_OK) return 0;
  return shape_scratch_bytes(*spec_host, B, T);
}

int32_t campx_shape_rollout_launch(const CampxShapeSpec* spec_host, const CampxShapeSpec* spec_dev,
                                   const void* tables_dev, CampxState st, int8_t* backdrop_state,
                                   const int8_t* actions, CampxOutputs out, int64_t B, int32_t T,
                                   int32_t reset_first, int32_t emit_first, void* stream) {
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
  if (shape_split_ok(*spec_host, tables_dev, out, B, T, emit_first)) {
    if (reinterpret_cast<uintptr_t>(out.trace) & 7) return CAMPX_EINVAL;
    // every frame kept, int8: update pass -> offset trace + trail keyframes -> frame-major render
    return launch_shape_split(sp, *spec_host, spec_dev, tables_dev, st, backdrop_state, actions, out, B, T,
                              reset_first, s);
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
