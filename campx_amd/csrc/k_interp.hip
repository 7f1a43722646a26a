// k_interp.hip - rollout_kernel: the rule interpreter, fused with the render (one lane = one
// environment, one wave = one workgroup) or in trace mode (the update pass of the two-kernel
// path for games without a table, and the builder of every table).
#include "campx_common.hip.h"

namespace campx_impl {

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
    ret += real_reward(reward);
    if (out.perf && rb.perf_dyn >= 0 && live) {
      int code;
      if (rb.perf_mode == 0) {   // progress round the cycle of cell classes
        const int perf_to = sel<K>(pos.cell, rb.perf_dyn);
        code = class_progress(cell_class[perf_from], cell_class[perf_to], rb.perf_n) + 1;
      } else {                   // penalty classes of where the watched things stand now
        code = 0;
#pragma unroll
        for (int k = 0; k < K; ++k)
          code += ((rb.perf_mask >> k) & 1) ? (int)cell_class[sel<K>(pos.cell, k)] : 0;
      }
      out.perf[(int64_t)t * row_pitch(out, B) + env] =
          (int8_t)(code * rb.perf_scale + rb.perf_offset);
    }

    if (kTrace) {
      // ---- split path: record where things are (and whether they show); the
      // render kernel turns that into observations.
      if (live) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int cell = sel<K>(pos.cell, k);
          const uint32_t vis = shown_layer<K>(rb, tab, W, cell, pos) == rb.dyn_layer[k];
          out.trace[(int64_t)k * trace_plane + (int64_t)t * row_pitch(out, B) + env] = pack_trace(cell, vis);
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
      const int64_t at = (int64_t)t * row_pitch(out, B) + env;
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

size_t lds_bytes(const CampxSpec& s, bool board, int envs) {
  const int HW = s.rows * s.cols, LHW = s.n_layers * HW;
  size_t n = (size_t)((envs * LHW + 15) & ~15);
  if (board) n += (size_t)((envs * HW + 15) & ~15);
  n += (size_t)((LHW + 15) & ~15);
  n += CAMPX_MAX_CELLS * 2 + CAMPX_MAX_CELLS * sizeof(uint16_t) + CAMPX_MAX_LAYERS;
  n += (size_t)kChunk * kWave + CAMPX_MAX_CELLS;
  return (n + 15) & ~(size_t)15;
}

namespace {

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
  rb.perf_mode = s.perf_mode;
  rb.perf_mask = s.perf_mask;
  rb.perf_scale = s.perf_scale;
  rb.perf_offset = s.perf_offset;
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
  const bool nt = out.obs_t_stride != 0;
#define CAMPX_LAUNCH_E(BOARD, NT, ENVS)                                                  \
  CAMPX_ALLOW_LDS((rollout_kernel<K, BOARD, NT, ENVS, false>), shmem);                       \
  hipLaunchKernelGGL((rollout_kernel<K, BOARD, NT, ENVS, false>), grid, block, shmem, stream, \
                     rb, spec_dev, st, actions, out, B, T, reset_first, emit_first, 0, \
                     (int64_t)T * row_pitch(out, B))
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

}  // namespace

int32_t launch_interp(const CampxSpec& s, const CampxSpec* spec_dev, CampxState st,
                      const int8_t* actions, CampxOutputs out, int64_t B, int32_t T,
                      int32_t reset_first, int32_t emit_first, hipStream_t stream) {
  switch (s.n_dyn) {
    case 1: return launch_k<1>(s, spec_dev, st, actions, out, B, T, reset_first, emit_first, stream);
    case 2: return launch_k<2>(s, spec_dev, st, actions, out, B, T, reset_first, emit_first, stream);
    case 3: return launch_k<3>(s, spec_dev, st, actions, out, B, T, reset_first, emit_first, stream);
    case 4: return launch_k<4>(s, spec_dev, st, actions, out, B, T, reset_first, emit_first, stream);
    default: return CAMPX_ESPEC;
  }
}

void launch_trace(const CampxSpec& s, const CampxSpec* spec_dev, CampxState st,
                  const int8_t* actions, CampxOutputs out, int64_t B, int32_t T,
                  int32_t reset_first, int64_t trace_plane, hipStream_t stream) {
  switch (s.n_dyn) {
    case 1: launch_trace_k<1>(s, spec_dev, st, actions, out, B, T, reset_first, trace_plane, stream); break;
    case 2: launch_trace_k<2>(s, spec_dev, st, actions, out, B, T, reset_first, trace_plane, stream); break;
    case 3: launch_trace_k<3>(s, spec_dev, st, actions, out, B, T, reset_first, trace_plane, stream); break;
    default: launch_trace_k<4>(s, spec_dev, st, actions, out, B, T, reset_first, trace_plane, stream); break;
  }
}

}  // namespace campx_impl
