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
  ret += real_reward(tr.reward);
  if (!(tr.paint & 0x80u)) {   // the mover shows at its cell
    int8_t* my_obs = obs_img + lane * LHW;
    my_obs[(int)(tr.paint & 0x7fu) * HW + cell] = 0;
    my_obs[mp.dyn_layer * HW + cell] = 1;
    if (kBoard) board_img[lane * HW + cell] = (int8_t)spec->layer_char[mp.dyn_layer];
  }
  if (live) {
    const uint8_t ended = tr.done & 1u;
    if (out.reward) out.reward[env] = tr.reward;
    if (out.discount)
      out.discount[env] = (tr.done >> 4) ? spec->discount_list[tr.done >> 4] : (ended ? 0.0f : 1.0f);
    if (out.done) out.done[env] = ended;
    if (out.perf) out.perf[env] = tr.perf;
    st.pos[env] = (int8_t)(cell / W);
    st.pos[B + env] = (int8_t)(cell % W);
    st.done[env] = ended;
    if (st.ret) st.ret[env] = ret;
  }
  // one wave: LDS operations complete in order, no barrier needed
  step_stream_obs(obs_img, out, env0 * LHW, n_live * LHW, lane);
  if (kBoard) stream_out<false>(board_img, out.board + env0 * HW, n_live * HW, lane);
  report_bad_actions(out, bad);
}

// ---------------------------------------------------------------------------
// The default one-frame kernel of one-mover games: row-group-major.  A wave owns n
// consecutive environments (n * R a multiple of 16 bytes, about 3 KiB: 16 environments of
// the boat race = 2 800 B) instead of 64: four times as many waves as step_table_kernel,
// each with a quarter of the image to fill and stream, so the chip sees one short burst
// from every SIMD at once instead of a long serial stream from a quarter of them.
// Everything the wave needs comes in ONE round trip - state and action of its n
// environments (one per lane), the transition table (only the KiB in use: one for the boat
// race), the scenery chunks of its span - then the lookup is an LDS read, the patch two LDS
// bytes, the output up to four 16-byte stores per lane.  State stays in place: an
// environment's row is read and written by one lane only (memory-aligned windows, as the
// render kernel has them, would share rows between waves, which in-place state cannot allow).
// Addresses are (uniform 64-bit base) + (32-bit lane offset) and the divisions by game
// constants are multiplications by reciprocals from the launcher: a wave of the one-shot
// kernel is a few hundred instructions, and with sixteen waves per CU that instruction
// stream, not HBM, is most of the kernel's time above the launch floor.
struct RowsParams {
  int32_t cols, cells, R, dyn_layer, cell0, mover_char;
  int32_t n;            // environments per wave
  int32_t n_tab;        // KiB of the transition table in use (whole chunks per lane)
  int32_t n_obs;        // KiB chunks of a wave's observation span, rounded up
  uint32_t inv_w;       // cell / cols == (cell * inv_w) >> 16 for cell < 128
  uint32_t inv_r;       // x / R == (x * inv_r) >> 24 for x < 1024
  uint32_t inv_hw;      // x / cells == (x * inv_hw) >> 24 for x < 1024
  uint32_t step_r;      // 1024 % R
};

constexpr int kRowsMaxChunks = 4;    // a wave's observation span is at most 4 KiB
constexpr int kRowsTableLoads = 5;   // the whole table: 5 KiB = 5 chunks per lane
#ifndef CAMPX_STEP_WAVES
#define CAMPX_STEP_WAVES 1           // waves per workgroup (independent: no barrier); 1, 2, 4 measure alike
#endif
constexpr int kStepWaves = CAMPX_STEP_WAVES;

// The wave's span of the frame out of its LDS image: 16 bytes per lane per store (a partial
// group - the batch's last - element by element).  Shared by the row-group-major kernels.
template <int kMaxChunks, int kBoardChunks, bool kBoard, int kFmt, bool kNT>
__device__ __forceinline__ void rows_stream_out(const int8_t* obs_img, const int8_t* board_img,
                                                const CampxOutputs& out, int64_t env0, uint32_t n,
                                                uint32_t n_live, uint32_t R, uint32_t HW,
                                                uint32_t n_obs, uint32_t lane) {
  const uint32_t span = n * R;
  constexpr uint32_t kElem = kFmt ? 2u : 1u;
  int8_t* obs_dst = out.obs + env0 * (int64_t)(R * kElem);
  int8_t* board_dst = kBoard ? out.board + env0 * (int64_t)HW : nullptr;
  if (n_live == n) {           // (every wave but, perhaps, the batch's last)
    if (kFmt == 0) {
#pragma unroll
      for (int j = 0; j < kMaxChunks; ++j) {
        const uint32_t o = ((uint32_t)j * kWave + lane) * 16u;
        if ((uint32_t)j < n_obs && o < span) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(obs_img + o);
          if (kNT) store16_streaming_at(obs_dst, o, v);
          else *reinterpret_cast<u32x4*>(obs_dst + o) = v;
        }
      }
    } else {
      // 16-bit observations (0.0 / 1.0 in f16 or bf16): 8 image bytes -> 16 output bytes
      constexpr uint32_t kOne = (kFmt == 1) ? 0x3C00u : 0x3F80u;
#pragma unroll
      for (int j = 0; j < 2 * kMaxChunks; ++j) {
        const uint32_t o = ((uint32_t)j * kWave + lane) * 8u;
        if ((uint32_t)j < 2u * n_obs && o < span) {
          const uint2 b = *reinterpret_cast<const uint2*>(obs_img + o);
          u32x4 v;
          v.x = ((b.x & 0xffu) | ((b.x << 8) & 0x00ff0000u)) * kOne;
          v.y = (((b.x >> 16) & 0xffu) | ((b.x >> 8) & 0x00ff0000u)) * kOne;
          v.z = ((b.y & 0xffu) | ((b.y << 8) & 0x00ff0000u)) * kOne;
          v.w = (((b.y >> 16) & 0xffu) | ((b.y >> 8) & 0x00ff0000u)) * kOne;
          if (kNT) store16_streaming_at(obs_dst, 2u * o, v);
          else *reinterpret_cast<u32x4*>(obs_dst + 2u * o) = v;
        }
      }
    }
    if (kBoard) {
#pragma unroll
      for (int j = 0; j < kBoardChunks; ++j) {
        const uint32_t o = ((uint32_t)j * kWave + lane) * 16u;
        if (o < n * HW)
          *reinterpret_cast<u32x4*>(board_dst + o) = *reinterpret_cast<const u32x4*>(board_img + o);
      }
    }
  } else {                     // a partial group: element by element
    const uint32_t live = n_live * R;
    if (kFmt == 0) {
#pragma unroll 1
      for (uint32_t i = lane; i < live; i += kWave) obs_dst[i] = obs_img[i];
    } else {
      const uint16_t one = (kFmt == 1) ? 0x3C00u : 0x3F80u;
#pragma unroll 1
      for (uint32_t i = lane; i < live; i += kWave)
        reinterpret_cast<uint16_t*>(obs_dst)[i] = obs_img[i] ? one : (uint16_t)0;
    }
    if (kBoard) {
#pragma unroll 1
      for (uint32_t i = lane; i < n_live * HW; i += kWave) board_dst[i] = board_img[i];
    }
  }
}

template <bool kBoard, int kFmt, bool kNT>
__global__ __launch_bounds__(kStepWaves * kWave) void step_rows_kernel(
    RowsParams rp, const CampxSpec* __restrict__ spec, CampxState st,
    const int8_t* __restrict__ actions, CampxOutputs out, int64_t B, int32_t reset_first) {
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t W = (uint32_t)rp.cols, HW = (uint32_t)rp.cells, R = (uint32_t)rp.R;
  const uint32_t n = (uint32_t)rp.n;
  // (block b runs on XCD b % 8: each XCD sweeps its own contiguous eighth of the frame)
  const uint32_t blk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int64_t env0 = (int64_t)(blk * (uint32_t)kStepWaves + wave) * n;
  if (env0 >= B) return;   // (uniform per wave; waves share nothing)
  const uint32_t n_live = (B - env0 < (int64_t)n) ? (uint32_t)(B - env0) : n;
  const uint32_t n_obs = (uint32_t)rp.n_obs, n_tab = (uint32_t)rp.n_tab;
  int8_t* obs_img = lds + wave * ((n_obs + n_tab + (kBoard ? 1u : 0u)) << 10);
  int8_t* board_img = obs_img + (n_obs << 10);
  const uint2* lds_table = reinterpret_cast<const uint2*>(board_img + (kBoard ? 1024 : 0));

  // ---- every load of the kernel, back to back (no branch between them: a conditional
  // load makes hipcc wait for it at the join - one round trip per load; chunks past the
  // table's / the span's end are clamped to valid addresses - every offset of the
  // cyclically continued scenery row is one - and dropped when they land)
  const bool mine = lane < n_live;
  const uint32_t ln = mine ? lane : n_live - 1u;    // surplus lanes reload the last environment
  int8_t* pos_r = st.pos + env0;
  int8_t* pos_c = st.pos + B + env0;
  uint8_t* done_base = st.done + env0;
  // (no running return asked for: any valid address, the value is not used)
  const float* ret_src = st.ret ? st.ret + env0 : reinterpret_cast<const float*>(spec) - ln;
  int a = (actions + env0)[ln];
  int r = pos_r[ln], c = pos_c[ln], over = done_base[ln];
  float ret = ret_src[ln];
  u32x4 v_table[kRowsTableLoads];
  const uint32_t last_vec = (n_tab << 6) - 1u;
#pragma unroll
  for (int j = 0; j < kRowsTableLoads; ++j) {
    uint32_t idx = (uint32_t)j * kWave + lane;
    idx = idx < last_vec ? idx : last_vec;
    v_table[j] = reinterpret_cast<const u32x4*>(spec->table)[idx];
  }
  u32x4 v_obs[kRowsMaxChunks], v_board;
  {
    const uint32_t pitch = ((R + 15u) & ~15u) + 16u;
    const uint32_t x = lane * 16u;
    uint32_t k = x - ((x * rp.inv_r) >> 24) * R;   // x % R
#pragma unroll
    for (int j = 0; j < kRowsMaxChunks; ++j) {
      v_obs[j] = *reinterpret_cast<const u32x4*>(spec->rot_obs + ((k & 15u) * pitch + (k & ~15u)));
      k += rp.step_r;
      k = k >= R ? k - R : k;
    }
    if (kBoard) {   // n * HW <= 1 KiB (launcher)
      const uint32_t bpitch = ((HW + 15u) & ~15u) + 16u;
      const uint32_t kb = x - ((x * rp.inv_hw) >> 24) * HW;
      v_board = *reinterpret_cast<const u32x4*>(spec->rot_board + ((kb & 15u) * bpitch + (kb & ~15u)));
    }
  }

  // ---- the one wait of the kernel: everything above has been issued, all of it lands here
  // (the empty asm statements keep hipcc from sinking a load into the branch that uses it)
#pragma unroll
  for (int j = 0; j < kRowsTableLoads; ++j) asm volatile("" : "+v"(v_table[j]));
#pragma unroll
  for (int j = 0; j < kRowsMaxChunks; ++j) asm volatile("" : "+v"(v_obs[j]));
  const int bad = (mine && (unsigned)a > 4u) ? 1 : 0;
  a = ((unsigned)a > 4u) ? 4 : a;
  // a finished episode (or reset_first) is rebuilt from the art before its action
  over = reset_first ? 1 : over;
  uint32_t cell = over ? (uint32_t)rp.cell0 : (uint32_t)r * W + (uint32_t)c;
  ret = over ? 0.0f : ret;
  // ---- land (LDS operations of a wave complete in order: no barrier)
  u32x4* tab_dst = reinterpret_cast<u32x4*>(const_cast<uint2*>(lds_table)) + lane;
#pragma unroll
  for (int j = 0; j < kRowsTableLoads; ++j)
    if ((uint32_t)j < n_tab) tab_dst[j * kWave] = v_table[j];          // uniform condition
  u32x4* img_dst = reinterpret_cast<u32x4*>(obs_img) + lane;
#pragma unroll
  for (int j = 0; j < kRowsMaxChunks; ++j)
    if ((uint32_t)j < n_obs) img_dst[j * kWave] = v_obs[j];             // uniform condition
  if (kBoard) reinterpret_cast<u32x4*>(board_img)[lane] = v_board;

  const uint2 tr = lds_table[cell * CAMPX_N_ACTIONS + (uint32_t)a];     // CampxTransition
  const float reward = __uint_as_float(tr.x);
  cell = tr.y & 0xffu;
  const uint32_t done = (tr.y >> 8) & 1u, dcode = (tr.y >> 12) & 15u, paint = tr.y >> 24;
  if (mine) {
    if (!(paint & 0x80u)) {   // the mover shows at its cell
      int8_t* my_obs = obs_img + lane * R;
      my_obs[(paint & 0x7fu) * HW + cell] = 0;
      my_obs[(uint32_t)rp.dyn_layer * HW + cell] = 1;
      if (kBoard) board_img[lane * HW + cell] = (int8_t)rp.mover_char;
    }
    if (out.reward) (out.reward + env0)[lane] = reward;
    if (out.discount)
      (out.discount + env0)[lane] = dcode ? spec->discount_list[dcode] : (done ? 0.0f : 1.0f);
    if (out.done) (out.done + env0)[lane] = (uint8_t)done;
    if (out.perf) (out.perf + env0)[lane] = (int8_t)(tr.y >> 16);
    const uint32_t row = (cell * rp.inv_w) >> 16;
    pos_r[lane] = (int8_t)row;
    pos_c[lane] = (int8_t)(cell - row * W);
    done_base[lane] = (uint8_t)done;
    if (st.ret) (st.ret + env0)[lane] = ret + real_reward(reward);
  }
  rows_stream_out<kRowsMaxChunks, 1, kBoard, kFmt, kNT>(obs_img, board_img, out, env0, n, n_live, R, HW, n_obs, lane);
  report_bad_actions(out, bad);
}

// ---------------------------------------------------------------------------
// Row-group-major for games with two to four movers (and, as an A/B, one): the shape of
// step_rows_kernel - a wave owns n consecutive environments with n * R about 3 KiB, every
// independent load issued before the first wait - around the ONE dependent trip a K-mover
// game cannot avoid: its (cell, ..., cell, action) table is megabytes, so the entry is a
// gather from L2, issued as soon as the state has landed (the state loads go first, so that
// waiting for them - vmcnt counts in order - does not wait for the scenery), and the scenery
// chunks, the reward list (1 KiB) and the scenery's top layers (128 B) land in LDS under it.
// Round 2's step_pair_kernel / step_tuple_kernel gave a wave 64 environments (18-21 KiB of
// image: eight waves per CU, four launches' worth of serial stream per wave).
struct DepParams {
  int32_t dyn_layer[CAMPX_MAX_DYN], cell0[CAMPX_MAX_DYN], mover_char[CAMPX_MAX_DYN];
  int32_t perf_scale, perf_offset;
  uint32_t step_hw;     // 1024 % cells
};

constexpr int kDepMaxChunks = 8;     // a wave's observation span is at most 8 KiB (kChunks: 4, 6 or 8 loads per lane) ...
constexpr int kDepBoardChunks = 2;   // ... and its flat boards 2 KiB

template <int K, int kChunks, bool kBoard, int kFmt>
__global__ __launch_bounds__(kWave) void step_rows_dep_kernel(
    RowsParams rp, DepParams dp, const CampxSpec* __restrict__ spec, CampxState st,
    const int8_t* __restrict__ actions, CampxOutputs out, int64_t B, int32_t reset_first) {
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];
  const uint32_t lane = threadIdx.x;
  const uint32_t W = (uint32_t)rp.cols, HW = (uint32_t)rp.cells, R = (uint32_t)rp.R;
  const uint32_t n = (uint32_t)rp.n;
  // (block b runs on XCD b % 8: each XCD sweeps its own contiguous eighth of the frame)
  const uint32_t blk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int64_t env0 = (int64_t)blk * n;
  if (env0 >= B) return;
  const uint32_t n_live = (B - env0 < (int64_t)n) ? (uint32_t)(B - env0) : n;
  const uint32_t n_obs = (uint32_t)rp.n_obs;
  int8_t* obs_img = lds;
  int8_t* board_img = obs_img + (n_obs << 10);
  float* lds_rewards = reinterpret_cast<float*>(board_img + (kBoard ? kDepBoardChunks * 1024 : 0));   // K > 1
  uint8_t* lds_top = reinterpret_cast<uint8_t*>(lds_rewards + 256);                // K > 1

  // ---- the state first (what the table index needs) ...
  const bool mine = lane < n_live;
  const uint32_t ln = mine ? lane : n_live - 1u;    // surplus lanes reload the last environment
  uint8_t* done_base = st.done + env0;
  const float* ret_src = st.ret ? st.ret + env0 : reinterpret_cast<const float*>(spec) - ln;
  int a = (actions + env0)[ln];
  int over = done_base[ln];
  uint32_t cells[K];
  int rr[K], cc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    rr[k] = (st.pos + (int64_t)(2 * k) * B + env0)[ln];
    cc[k] = (st.pos + (int64_t)(2 * k + 1) * B + env0)[ln];
  }
  float ret = ret_src[ln];
  // ---- ... then everything that does not depend on it
  u32x4 v_obs[kChunks], v_board[kDepBoardChunks], rw4, top4;
  {
    const uint32_t pitch = ((R + 15u) & ~15u) + 16u;
    const uint32_t x = lane * 16u;
    uint32_t k = x - ((x * rp.inv_r) >> 24) * R;   // x % R
#pragma unroll
    for (int j = 0; j < kChunks; ++j) {
      v_obs[j] = *reinterpret_cast<const u32x4*>(spec->rot_obs + ((k & 15u) * pitch + (k & ~15u)));
      k += rp.step_r;
      k = k >= R ? k - R : k;
    }
    if (kBoard) {   // n * HW <= 2 KiB (launcher)
      const uint32_t bpitch = ((HW + 15u) & ~15u) + 16u;
      uint32_t kb = x - ((x * rp.inv_hw) >> 24) * HW;
#pragma unroll
      for (int j = 0; j < kDepBoardChunks; ++j) {
        v_board[j] = *reinterpret_cast<const u32x4*>(spec->rot_board + ((kb & 15u) * bpitch + (kb & ~15u)));
        kb += dp.step_hw;
        kb = kb >= HW ? kb - HW : kb;
      }
    }
    if (K > 1) {
      rw4 = static_cast<const u32x4*>(st.pair_table)[lane];
      top4 = reinterpret_cast<const u32x4*>(spec->static_top_layer)[lane & 7u];
    }
  }
  // ---- the table entry: the one dependent trip.  (The empty asm statements pin the state
  // loads where they were written - hipcc otherwise sinks some of them into the branch that
  // uses them, behind a full wait; needing their values HERE costs a wait for them alone,
  // vmcnt counts in order and they were issued first.)
  asm volatile("" : "+v"(a), "+v"(over));
#pragma unroll
  for (int k = 0; k < K; ++k) asm volatile("" : "+v"(rr[k]), "+v"(cc[k]));
  const int bad = (mine && (unsigned)a > 4u) ? 1 : 0;
  a = ((unsigned)a > 4u) ? 4 : a;
  over = reset_first ? 1 : over;       // a finished episode is rebuilt from the art first
  ret = over ? 0.0f : ret;
#pragma unroll
  for (int k = 0; k < K; ++k)
    cells[k] = over ? (uint32_t)dp.cell0[k] : (uint32_t)rr[k] * W + (uint32_t)cc[k];
  uint32_t idx = cells[0];
#pragma unroll
  for (int k = 1; k < K; ++k) idx = idx * HW + cells[k];
  idx = idx * CAMPX_N_ACTIONS + (uint32_t)a;
  uint32_t e_lo, e_hi = 0;     // K = 1: CampxTransition {reward bits, cell | done | perf | paint}
  if (K == 1) {
    const uint2 tr = reinterpret_cast<const uint2*>(spec->table)[idx];
    e_lo = tr.y;
    e_hi = tr.x;
  } else if (K == 2) {
    e_lo = (reinterpret_cast<const uint32_t*>(static_cast<const float*>(st.pair_table) + 256))[idx];
  } else {
    const uint2 e = (reinterpret_cast<const uint2*>(static_cast<const float*>(st.pair_table) + 256))[idx];
    e_lo = e.x;
    e_hi = e.y;
  }
  // ---- land the scenery under it (LDS operations of a wave complete in order: no barrier)
#pragma unroll
  for (int j = 0; j < kChunks; ++j) asm volatile("" : "+v"(v_obs[j]));
  u32x4* img_dst = reinterpret_cast<u32x4*>(obs_img) + lane;
#pragma unroll
  for (int j = 0; j < kChunks; ++j)
    if ((uint32_t)j < n_obs) img_dst[j * kWave] = v_obs[j];             // uniform condition
  if (kBoard) {
#pragma unroll
    for (int j = 0; j < kDepBoardChunks; ++j)
      reinterpret_cast<u32x4*>(board_img)[j * kWave + lane] = v_board[j];
  }
  if (K > 1) {
    reinterpret_cast<u32x4*>(lds_rewards)[lane] = rw4;
    if (lane < 8u) reinterpret_cast<u32x4*>(lds_top)[lane] = top4;
  }
  // ---- decode, patch this lane's environment, scalars and state
  float reward;
  uint32_t done, dcode, perf_byte_v, shows = 0;
  if (K == 1) {
    reward = __uint_as_float(e_hi);
    cells[0] = e_lo & 0xffu;
    done = (e_lo >> 8) & 1u;
    dcode = (e_lo >> 12) & 15u;
    perf_byte_v = (e_lo >> 16) & 0xffu;
    shows = ((e_lo >> 31) & 1u) ^ 1u;
  } else if (K == 2) {
    cells[0] = e_lo & 0x7fu;
    cells[1] = (e_lo >> 7) & 0x7fu;
    shows = (e_lo >> 14) & 3u;
    done = (e_lo >> 16) & 1u;
    dcode = dcode_pair(e_lo);
    perf_byte_v = (uint32_t)((int)perf_code_pair(e_lo) * dp.perf_scale + dp.perf_offset);
    reward = lds_rewards[(e_lo >> 19) & 0xffu];
  } else {
#pragma unroll
    for (int k = 0; k < K; ++k) cells[k] = (e_lo >> (7 * k)) & 0x7fu;
    shows = (e_lo >> 28) & 15u;
    done = e_hi & 1u;
    dcode = dcode_tuple(e_hi);
    perf_byte_v = (uint32_t)((int)perf_code_tuple(e_hi) * dp.perf_scale + dp.perf_offset);
    reward = lds_rewards[(e_hi >> 3) & 0xffu];
  }
  if (mine) {
    int8_t* my_obs = obs_img + lane * R;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if ((shows >> k) & 1u) {   // it is the character its cell shows
        const uint32_t c = cells[k];
        const uint32_t under = (K == 1) ? ((e_lo >> 24) & 0x7fu) : (uint32_t)lds_top[c];
        my_obs[under * HW + c] = 0;
        my_obs[(uint32_t)dp.dyn_layer[k] * HW + c] = 1;
        if (kBoard) board_img[lane * HW + c] = (int8_t)dp.mover_char[k];
      }
    }
    if (out.reward) (out.reward + env0)[lane] = reward;
    if (out.discount)
      (out.discount + env0)[lane] = dcode ? spec->discount_list[dcode] : (done ? 0.0f : 1.0f);
    if (out.done) (out.done + env0)[lane] = (uint8_t)done;
    if (out.perf) (out.perf + env0)[lane] = (int8_t)perf_byte_v;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const uint32_t row = (cells[k] * rp.inv_w) >> 16;
      (st.pos + (int64_t)(2 * k) * B + env0)[lane] = (int8_t)row;
      (st.pos + (int64_t)(2 * k + 1) * B + env0)[lane] = (int8_t)(cells[k] - row * W);
    }
    done_base[lane] = (uint8_t)done;
    if (st.ret) (st.ret + env0)[lane] = ret + real_reward(reward);
  }
  rows_stream_out<kChunks, kDepBoardChunks, kBoard, kFmt, false>(obs_img, board_img, out, env0, n, n_live, R, HW, n_obs, lane);
  report_bad_actions(out, bad);
}

// The same for two-mover games with a pair table (campx_pair_table_build): one more
// dependent round trip for the table entry (the reward list and the scenery layers the movers
// cover come along in the first).
template <bool kBoard>
__global__ __launch_bounds__(kWave) void step_pair_kernel(
    MoverParams mp, int32_t layer1, int32_t row1, int32_t col1,
    const CampxSpec* __restrict__ spec, CampxState st, const int8_t* __restrict__ actions,
    CampxOutputs out, int64_t B, int32_t reset_first) {
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];
  __shared__ __attribute__((aligned(16))) float lds_rewards[256];
  __shared__ __attribute__((aligned(16))) uint8_t lds_top[CAMPX_MAX_CELLS];
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
  // The reward list (256 floats) and the scenery's top layer per cell (128 bytes) ride in
  // this first round trip - four floats and two cells per lane, parked in LDS - so that what
  // the table entry points at is an LDS read, not a third trip to memory.
  const u32x4 rw4 = reinterpret_cast<const u32x4*>(g_rewards)[lane];
  const uint16_t top2 = reinterpret_cast<const uint16_t*>(spec->static_top_layer)[lane];

  const int bad = (live && (unsigned)a > 4u) ? 1 : 0;
  a = ((unsigned)a > 4u) ? 4 : a;
  if (over) {  // rebuilt from the art before its next action
    c0 = init0;
    c1 = init1;
    ret = 0.0f;
  }
  const uint32_t e = g_entries[pair_index(c0, c1, HW) + (uint32_t)a];
  reinterpret_cast<u32x4*>(lds_rewards)[lane] = rw4;
  reinterpret_cast<uint16_t*>(lds_top)[lane] = top2;
  fill_land<kStepObsLoads>(v_obs, obs_img, kWave * LHW, 0, lane);
  fill_rest<kStepObsLoads>(obs_img, kWave * LHW, LHW, spec->rot_obs, k_obs, lane);
  if (kBoard) {
    fill_land<kStepBoardLoads>(v_board, board_img, kWave * HW, 0, lane);
    fill_rest<kStepBoardLoads>(board_img, kWave * HW, HW, spec->rot_board, k_board, lane);
  }
  c0 = e & 0x7fu;
  c1 = (e >> 7) & 0x7fu;
  const float reward = lds_rewards[(e >> 19) & 0xffu];
  const int done = (int)((e >> 16) & 1u);
  ret += real_reward(reward);
  int8_t* my_obs = obs_img + lane * LHW;
  if ((e >> 14) & 1u) {
    my_obs[(int)lds_top[c0] * HW + (int)c0] = 0;
    my_obs[mp.dyn_layer * HW + (int)c0] = 1;
    if (kBoard) board_img[lane * HW + (int)c0] = (int8_t)spec->layer_char[mp.dyn_layer];
  }
  if ((e >> 15) & 1u) {
    my_obs[(int)lds_top[c1] * HW + (int)c1] = 0;
    my_obs[layer1 * HW + (int)c1] = 1;
    if (kBoard) board_img[lane * HW + (int)c1] = (int8_t)spec->layer_char[layer1];
  }
  if (live) {
    if (out.reward) out.reward[env] = reward;
    if (out.discount)
      out.discount[env] = dcode_pair(e) ? spec->discount_list[dcode_pair(e)] : (done ? 0.0f : 1.0f);
    if (out.done) out.done[env] = (uint8_t)done;
    if (out.perf)
      out.perf[env] = (int8_t)((int)perf_code_pair(e) * spec->perf_scale + spec->perf_offset);
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
  __shared__ __attribute__((aligned(16))) float lds_rewards[256];
  __shared__ __attribute__((aligned(16))) uint8_t lds_top[CAMPX_MAX_CELLS];
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
  // (reward list and scenery top layers in the first round trip, as in step_pair_kernel)
  const u32x4 rw4 = reinterpret_cast<const u32x4*>(g_rewards)[lane];
  const uint16_t top2 = reinterpret_cast<const uint16_t*>(spec->static_top_layer)[lane];

  const int bad = (live && (unsigned)a > 4u) ? 1 : 0;
  a = ((unsigned)a > 4u) ? 4 : a;
  if (over) {  // rebuilt from the art before its next action
    cells = init;
    ret = 0.0f;
  }
  const uint64_t e = g_entries[tuple_index<K>(cells, (uint32_t)HW) + (uint32_t)a];
  reinterpret_cast<u32x4*>(lds_rewards)[lane] = rw4;
  reinterpret_cast<uint16_t*>(lds_top)[lane] = top2;
  fill_land<kStepObsLoads>(v_obs, obs_img, kWave * LHW, 0, lane);
  fill_rest<kStepObsLoads>(obs_img, kWave * LHW, LHW, spec->rot_obs, k_obs, lane);
  if (kBoard) {
    fill_land<kStepBoardLoads>(v_board, board_img, kWave * HW, 0, lane);
    fill_rest<kStepBoardLoads>(board_img, kWave * HW, HW, spec->rot_board, k_board, lane);
  }
  const uint32_t lo = (uint32_t)e, hi = (uint32_t)(e >> 32);
  const float reward = lds_rewards[(hi >> 3) & 0xffu];
  const int done = (int)(hi & 1u);
  ret += real_reward(reward);
  int8_t* my_obs = obs_img + lane * LHW;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int c = (int)((lo >> (7 * k)) & 0x7fu);
    if ((lo >> (28 + k)) & 1u) {   // it is the character its cell shows
      my_obs[(int)lds_top[c] * HW + c] = 0;
      my_obs[tp.dyn_layer[k] * HW + c] = 1;
      if (kBoard) board_img[lane * HW + c] = (int8_t)spec->layer_char[tp.dyn_layer[k]];
    }
  }
  if (live) {
    if (out.reward) out.reward[env] = reward;
    if (out.discount)
      out.discount[env] = dcode_tuple(hi) ? spec->discount_list[dcode_tuple(hi)] : (done ? 0.0f : 1.0f);
    if (out.done) out.done[env] = (uint8_t)done;
    if (out.perf)
      out.perf[env] = (int8_t)((int)perf_code_tuple(hi) * spec->perf_scale + spec->perf_offset);
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

// Environments per wave of step_rows_kernel for this game, or 0 when its rows do not fit the
// kernel's shape: n * R and (with a board) n * HW must be multiples of 16 bytes - 8 for the
// 16-bit formats -, n <= 64, the observation span at most 4 KiB and the board span 1 KiB.
int rows_per_wave(const CampxSpec& s, bool board, int fmt, int max_chunks = kRowsMaxChunks,
                  int board_chunks = 1, int target = 0) {
  const int HW = s.rows * s.cols, R = s.n_layers * HW;
  auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
  const int unit = fmt ? 8 : 16;
  int n0 = unit / gcd(R, unit);
  if (board) {
    const int nb = 16 / gcd(HW, 16);
    n0 = n0 / gcd(n0, nb) * nb;   // lcm
  }
  if (n0 > kWave || n0 * R > max_chunks * 1024 || (board && n0 * HW > board_chunks * 1024) ||
      R < 16 || HW < 4)
    return 0;
  // about `target` bytes of observation per wave, in whole multiples of n0
  if (target <= 0) target = 3072;
  if (target > max_chunks * 1024) target = max_chunks * 1024;
  int n = n0 * (target / (n0 * R) > 0 ? target / (n0 * R) : 1);
  while (n > n0 && (n > kWave || (board && n * HW > board_chunks * 1024))) n -= n0;
  return n;
}

int32_t launch_step_rows(const CampxSpec& s, const CampxSpec* spec_dev, CampxState st,
                         const int8_t* actions, CampxOutputs out, int64_t B, int32_t reset_first,
                         int n, hipStream_t stream) {
  const int HW = s.rows * s.cols, R = s.n_layers * HW;
  const bool board = out.board != nullptr;
  RowsParams rp;
  rp.cols = s.cols;
  rp.cells = HW;
  rp.R = R;
  rp.dyn_layer = s.dyn_layer[0];
  rp.cell0 = s.dyn_row0[0] * s.cols + s.dyn_col0[0];
  rp.mover_char = s.layer_char[s.dyn_layer[0]];
  rp.n = n;
  rp.n_tab = (HW * CAMPX_N_ACTIONS * (int)sizeof(CampxTransition) + 1023) / 1024;
  rp.n_obs = (n * R + 1023) / 1024;
  // exact for the ranges the kernel uses them on (checked by tests/test_launch_math.py's
  // restatement: cell < 128 with cols <= 127; x < 1024 with x * d < 2^24)
  rp.inv_w = (65536u + (uint32_t)s.cols - 1u) / (uint32_t)s.cols;
  rp.inv_r = ((1u << 24) + (uint32_t)R - 1u) / (uint32_t)R;
  rp.inv_hw = ((1u << 24) + (uint32_t)HW - 1u) / (uint32_t)HW;
  rp.step_r = 1024u % (uint32_t)R;
  size_t shmem = (size_t)kStepWaves * 1024 * (size_t)(rp.n_obs + rp.n_tab + (board ? 1 : 0));
  const int64_t waves = (B + n - 1) / n;
  // rounded up to a multiple of 8 for the XCD remap; surplus waves exit at once
  const dim3 grid((unsigned)((((waves + kStepWaves - 1) / kStepWaves) + 7) & ~(int64_t)7)),
      block(kStepWaves * kWave);
#define CAMPX_ROWS3(BOARD, FMT, NT)                                                                \
  do {                                                                                             \
    CAMPX_ALLOW_LDS((step_rows_kernel<BOARD, FMT, NT>), shmem);                                    \
    hipLaunchKernelGGL((step_rows_kernel<BOARD, FMT, NT>), grid, block, shmem, stream, rp,         \
                       spec_dev, st, actions, out, B, reset_first);                                \
  } while (0)
#define CAMPX_ROWS(BOARD, FMT)                                           \
  do {                                                                   \
    /* (streaming stores measured no faster for one frame: profiles/r03_play_rocprofv3.txt) */ \
    CAMPX_ROWS3(BOARD, FMT, false);                                      \
  } while (0)
  if (board) {
    if (out.obs_format == CAMPX_OBS_F16) CAMPX_ROWS(true, 1);
    else if (out.obs_format == CAMPX_OBS_BF16) CAMPX_ROWS(true, 2);
    else CAMPX_ROWS(true, 0);
  } else {
    if (out.obs_format == CAMPX_OBS_F16) CAMPX_ROWS(false, 1);
    else if (out.obs_format == CAMPX_OBS_BF16) CAMPX_ROWS(false, 2);
    else CAMPX_ROWS(false, 0);
  }
#undef CAMPX_ROWS
#undef CAMPX_ROWS3
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? CAMPX_OK : hip_failed(e);
}

// Environments per wave of step_rows_dep_kernel: spans of up to 8 KiB (measured, B = 65 536,
// kernel us at spans of 3 / 4 / 8 KiB on one box: wall world 11.5 / 10.5 / 10.6, sokoban with
// three boxes 10.0 / 9.3 / 8.2, with two 8.3 / 7.7 / 7.7, boat race 6.5 / 6.4 / 6.4;
// profiles/r04_play_rocprofv3.txt).
int dep_rows_per_wave(const CampxSpec& s, bool board, int fmt) {
  return rows_per_wave(s, board, fmt, kDepMaxChunks, kDepBoardChunks, 8192);
}

int32_t launch_step_rows_dep(const CampxSpec& s, const CampxSpec* spec_dev, CampxState st,
                             const int8_t* actions, CampxOutputs out, int64_t B,
                             int32_t reset_first, int n, hipStream_t stream) {
  const int HW = s.rows * s.cols, R = s.n_layers * HW, K = s.n_dyn;
  const bool board = out.board != nullptr;
  RowsParams rp;
  rp.cols = s.cols;
  rp.cells = HW;
  rp.R = R;
  rp.dyn_layer = s.dyn_layer[0];
  rp.cell0 = s.dyn_row0[0] * s.cols + s.dyn_col0[0];
  rp.mover_char = s.layer_char[s.dyn_layer[0]];
  rp.n = n;
  rp.n_tab = 0;
  rp.n_obs = (n * R + 1023) / 1024;
  rp.inv_w = (65536u + (uint32_t)s.cols - 1u) / (uint32_t)s.cols;
  rp.inv_r = ((1u << 24) + (uint32_t)R - 1u) / (uint32_t)R;
  rp.inv_hw = ((1u << 24) + (uint32_t)HW - 1u) / (uint32_t)HW;
  rp.step_r = 1024u % (uint32_t)R;
  DepParams dp;
  memset(&dp, 0, sizeof(dp));
  for (int d = 0; d < K; ++d) {
    dp.dyn_layer[d] = s.dyn_layer[d];
    dp.cell0[d] = s.dyn_row0[d] * s.cols + s.dyn_col0[d];
    dp.mover_char[d] = s.layer_char[s.dyn_layer[d]];
  }
  dp.perf_scale = s.perf_scale;
  dp.perf_offset = s.perf_offset;
  dp.step_hw = 1024u % (uint32_t)HW;
  size_t shmem = 1024 * (size_t)(rp.n_obs + (board ? kDepBoardChunks : 0)) + (K > 1 ? 1024 + 128 : 0);
  const int64_t waves = (B + n - 1) / n;
  const dim3 grid((unsigned)((waves + 7) & ~(int64_t)7)), block(kWave);
#define CAMPX_DEP4(KK, CH, BOARD, FMT)                                                             \
  do {                                                                                             \
    CAMPX_ALLOW_LDS((step_rows_dep_kernel<KK, CH, BOARD, FMT>), shmem);                            \
    hipLaunchKernelGGL((step_rows_dep_kernel<KK, CH, BOARD, FMT>), grid, block, shmem, stream, rp, \
                       dp, spec_dev, st, actions, out, B, reset_first);                            \
  } while (0)
#define CAMPX_DEP3(KK, BOARD, FMT)                                                     \
  do {                                                                                 \
    if (rp.n_obs <= 4) CAMPX_DEP4(KK, 4, BOARD, FMT);                                  \
    else if (rp.n_obs <= 6) CAMPX_DEP4(KK, 6, BOARD, FMT);                             \
    else CAMPX_DEP4(KK, 8, BOARD, FMT);                                                \
  } while (0)
#define CAMPX_DEP2(KK, BOARD)                                            \
  do {                                                                   \
    if (out.obs_format == CAMPX_OBS_F16) CAMPX_DEP3(KK, BOARD, 1);       \
    else if (out.obs_format == CAMPX_OBS_BF16) CAMPX_DEP3(KK, BOARD, 2); \
    else CAMPX_DEP3(KK, BOARD, 0);                                       \
  } while (0)
#define CAMPX_DEP(KK)                                     \
  do {                                                    \
    if (board) CAMPX_DEP2(KK, true); else CAMPX_DEP2(KK, false); \
  } while (0)
  if (K == 1) CAMPX_DEP(1);
  else if (K == 2) CAMPX_DEP(2);
  else if (K == 3) CAMPX_DEP(3);
  else CAMPX_DEP(4);
#undef CAMPX_DEP
#undef CAMPX_DEP2
#undef CAMPX_DEP3
#undef CAMPX_DEP4
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? CAMPX_OK : hip_failed(e);
}

int32_t launch_step_table(const CampxSpec& s, const CampxSpec* spec_dev, CampxState st,
                          const int8_t* actions, CampxOutputs out, int64_t B, int32_t reset_first,
                          hipStream_t stream) {
  const int HW = s.rows * s.cols, LHW = s.n_layers * HW;
  const bool board = out.board != nullptr;
  const int nd = dep_rows_per_wave(s, board, out.obs_format);
  if (nd > 0) return launch_step_rows_dep(s, spec_dev, st, actions, out, B, reset_first, nd, stream);
  // (rows the dependent-trip kernel's shape does not take: round 3's kernel, the table in LDS)
  const int n = rows_per_wave(s, board, out.obs_format);
  if (n > 0) return launch_step_rows(s, spec_dev, st, actions, out, B, reset_first, n, stream);
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
  const int n = dep_rows_per_wave(s, board, out.obs_format);
  if (n > 0) return launch_step_rows_dep(s, spec_dev, st, actions, out, B, reset_first, n, stream);
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
  const int n = dep_rows_per_wave(s, board, out.obs_format);
  if (n > 0) return launch_step_rows_dep(s, spec_dev, st, actions, out, B, reset_first, n, stream);
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
