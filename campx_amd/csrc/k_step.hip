// k_step.hip - Engine.play(): one frame per launch.
#include "campx_common.hip.h"

namespace campx_impl {

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
    CAMPX_ALLOW_LDS((step_table_kernel<true>), shmem);
    hipLaunchKernelGGL(step_table_kernel<true>, grid, block, shmem, stream, mp, spec_dev, st,
                       actions, out, B, reset_first);
  } else {
    CAMPX_ALLOW_LDS((step_table_kernel<false>), shmem);
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
    CAMPX_ALLOW_LDS((step_pair_kernel<true>), shmem);
    hipLaunchKernelGGL(step_pair_kernel<true>, grid, block, shmem, stream, mp, s.dyn_layer[1],
                       s.dyn_row0[1], s.dyn_col0[1], spec_dev, st, actions, out, B, reset_first);
  } else {
    CAMPX_ALLOW_LDS((step_pair_kernel<false>), shmem);
    hipLaunchKernelGGL(step_pair_kernel<false>, grid, block, shmem, stream, mp, s.dyn_layer[1],
                       s.dyn_row0[1], s.dyn_col0[1], spec_dev, st, actions, out, B, reset_first);
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? CAMPX_OK : hip_failed(e);
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
    CAMPX_ALLOW_LDS((step_tuple_kernel<KK, BOARD>), shmem);                                     \
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

}  // namespace campx_impl
