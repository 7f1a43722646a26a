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
    // (an action nobody rewards - Hello World's quit, an id outside 0..4 - reports None =
    // NaN for the frame and leaves the running return alone)
    if (valid && (sp.act[a_raw].flags & 2u)) ret += reward;
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

}  // extern "C"
