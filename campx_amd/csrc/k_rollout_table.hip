// k_rollout_table.hip - rollout_table_kernel: one-mover games in the single fused kernel,
// the update pass a lookup in the (cell, action) table.
#include "campx_common.hip.h"

namespace campx_impl {

// ---------------------------------------------------------------------------
// One-mover games (K == 1) after campx_spec_compile(): the update pass of a frame
// is one lookup in the (cell, action) transition table, which was produced by the
// interpreter kernel above.  Rendering and streaming are unchanged.
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
                              ((uint32_t)(uint8_t)tr.perf << 16));   // (done: bit 0 + discount code)
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
    ret += real_reward(reward);

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
      const int64_t at = (int64_t)t * row_pitch(out, B) + env;
      if (out.reward) out.reward[at] = reward;
      if (out.discount) {
        const uint32_t dcode = (tr.y >> 12) & 15u;
        out.discount[at] = dcode ? spec->discount_list[dcode] : (over ? 0.0f : 1.0f);
      }
      if (out.done) out.done[at] = (uint8_t)over;
      if (out.perf) out.perf[at] = (int8_t)(tr.y >> 16);
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

size_t table_lds_bytes(const CampxSpec& s, bool board, int envs) {
  const int HW = s.rows * s.cols, LHW = s.n_layers * HW;
  size_t n = (size_t)((envs * LHW + 15) & ~15);
  if (board) n += (size_t)((envs * HW + 15) & ~15);
  n += sizeof(uint2) * CAMPX_MAX_CELLS * CAMPX_N_ACTIONS;
  n += CAMPX_MAX_CELLS * sizeof(uint16_t) + CAMPX_MAX_CELLS;
  n += (size_t)kChunk * kWave + (size_t)LHW;
  return (n + 15) & ~(size_t)15;
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
  const bool nt = out.obs_t_stride != 0;
  const size_t shmem = table_lds_bytes(s, board, envs);
  const dim3 grid((unsigned)((B + envs - 1) / envs)), block(kWave);
  const MoverParams mp = {s.rows, s.cols, s.n_layers, s.dyn_layer[0], s.dyn_z[0],
                          s.dyn_row0[0], s.dyn_col0[0]};
#define CAMPX_LAUNCH_E(BOARD, NT, ENVS)                                                      \
  CAMPX_ALLOW_LDS((rollout_table_kernel<BOARD, NT, ENVS>), shmem);                              \
  hipLaunchKernelGGL((rollout_table_kernel<BOARD, NT, ENVS>), grid, block, shmem, stream, mp, \
                     spec_dev, st, actions, out, B, T, reset_first, emit_first, 0)
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

}  // namespace campx_impl
