// k_wide.hip - the wide tier: games run from their STATE table ((state, action) -> state,
// one row per reachable state), which has no board-size limit: boards above 128 cells (up to
// 1 024), up to eight things that show, hidden values behind them.
//
// The update pass is a walk through that table, which the HOST filled
// (campx_amd/tabulate.py); the observation stream is k_render.hip's render kernel reading a
// 16-bit trace.  A frame row is L*rows*cols bytes - 1 536 for a 16x16 board with six
// characters - against 2 bytes of trace per thing, 4 of reward and 1 of action, so the update
// kernel here is the plain one (a lane per environment, the table in LDS, the frame's
// scalars stored as they come): it moves 1-2 % of the launch's bytes and takes 1.4 % of
// its time (20 us of 1 430 at 16x16, B = 65 536).  What bounds the tier is the render kernel's
// write stream (7.2 TB/s).

#include "campx_common.hip.h"

#include <type_traits>

namespace campx_impl {

constexpr int kWideThreads = 256;
constexpr int kWideAhead = 8;     // frames whose actions are fetched before their chain runs

struct WideParams {
  int32_t n_states, n_dyn;
  int32_t has_dcodes;          // some entry of the table carries a discount code
  float discounts[16];
  int64_t plane;               // entries from one thing's plane of the trace to the next's (the
                               // whole rollout's frames x row pitch, also when a launch runs a chunk of them)
};

// Table-blob entry of (state, action): x = reward; y = [0:23] the state after the frame,
// [24] done, [25:28] discount code.  (The state the NEXT frame starts from is state 0 when
// the frame ended the episode: the rebuild is one select on the chain.)
__host__ __device__ __forceinline__ uint32_t wide_pack(uint32_t next, uint32_t done, uint32_t dcode) {
  return next | (done << 24) | (dcode << 25);
}

// kLds: the state table (entries, per-state trace entries, perf bytes) sits in LDS; else it
// is read through L1 / L2 (games with thousands of states).
template <bool kLds, bool kPerf>
__global__ __launch_bounds__(kWideThreads) void wide_update_kernel(
    WideParams wp, const uint2* __restrict__ g_entries, const u32x4* __restrict__ g_cells,
    const int8_t* __restrict__ g_perf, int32_t* __restrict__ state, CampxState st,
    const int8_t* __restrict__ actions, CampxOutputs out, int64_t B, int32_t T,
    int32_t reset_first) {
  extern __shared__ __attribute__((aligned(16))) uint2 lds_tables[];
  __shared__ float discounts[16];
  const int S = wp.n_states, n_entries = S * CAMPX_N_ACTIONS, K = wp.n_dyn;
  const uint2* entries = g_entries;
  const u32x4* cells = g_cells;
  const int8_t* perf_tab = g_perf;
  if (kLds) {
    uint2* l_entries = lds_tables;
    u32x4* l_cells = reinterpret_cast<u32x4*>(l_entries + n_entries + (n_entries & 1));   // 16-byte aligned
    int8_t* l_perf = reinterpret_cast<int8_t*>(l_cells + S);
    for (int i = threadIdx.x; i < n_entries; i += kWideThreads) l_entries[i] = g_entries[i];
    for (int i = threadIdx.x; i < S; i += kWideThreads) l_cells[i] = g_cells[i];
    if (kPerf)
      for (int i = threadIdx.x; i < n_entries; i += kWideThreads) l_perf[i] = g_perf[i];
    entries = l_entries;
    cells = l_cells;
    perf_tab = l_perf;
  }
  if (threadIdx.x < 16) discounts[threadIdx.x] = wp.discounts[threadIdx.x];
  __syncthreads();

  const int64_t env = (int64_t)blockIdx.x * kWideThreads + threadIdx.x;
  if (env >= B) return;
  uint32_t now = 0;
  int over = 0;
  float ret = 0.0f;
  if (!reset_first) {
    now = (uint32_t)state[env];
    now = now < (uint32_t)S ? now : 0u;      // (a state index from outside: start over)
    over = st.done[env];
    if (st.ret) ret = st.ret[env];
  }
  uint32_t from = over ? 0u : now;
  uint16_t* trace = reinterpret_cast<uint16_t*>(out.trace);
  const int64_t P = row_pitch(out, B), plane = wp.plane;
  int bad = 0;
  int64_t at = env;                                // element (frame, env) of the [T, P] streams
  // The actions of a chunk of frames are loaded ONE CHUNK AHEAD, before the previous chunk's
  // stores are issued: vector memory operations complete in order, so a load issued after a
  // chunk's forty stores would wait for all of them (that was 28 us per 100 frames at
  // B = 4 096; this way the wait only covers what was issued before the loads).
  auto fetch = [&](int t0, uint32_t (&dst)[kWideAhead]) {
#pragma unroll
    for (int j = 0; j < kWideAhead; ++j) {
      const int t = t0 + j < T ? t0 + j : T - 1;   // (clamped: a load that is ignored)
      dst[j] = (uint8_t)actions[(int64_t)t * B + env];
    }
  };
  uint32_t a_next[kWideAhead];
  fetch(0, a_next);
  // One chunk of frames.  `kPlain`: a whole chunk of a game without discount codes and with
  // the usual output streams - no test inside, so the frames' table reads overlap (with
  // the tests every frame made three dependent LDS round trips and ten scalar branches:
  // 275 ns per frame, 28 us per 100 frames at B = 4 096).
  // (`plain_tag`: 0 = the general chunk; k = 1 .. 8 = a plain chunk of a game with k things)
  auto chunk = [&](auto plain_tag, int t0, const uint32_t (&a)[kWideAhead]) {
    constexpr int kThings = decltype(plain_tag)::value;
    constexpr bool kPlain = kThings > 0;
#pragma unroll
    for (int j = 0; j < kWideAhead; ++j) {
      if (kPlain || t0 + j < T) {
        bad += a[j] > 4u;
        const uint32_t idx = from * CAMPX_N_ACTIONS + (a[j] > 4u ? 4u : a[j]);
        const uint2 e = entries[idx];
        now = e.y & 0xffffffu;
        const uint32_t done = (e.y >> 24) & 1u, dcode = (e.y >> 25) & 15u;
        from = done ? 0u : now;                      // the chain: state -> entry -> state
        const u32x4 c = cells[now];                  // where things show in the state reached
        trace[at] = (uint16_t)c.x;
        if (kPlain) {
          const uint32_t w[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
          for (int d = 1; d < kThings; ++d)
            trace[at + d * plane] = (uint16_t)(w[d >> 1] >> (16 * (d & 1)));
        }
        if (!kPlain && K > 1) {
          uint16_t* tk = trace + at + plane;
          tk[0] = (uint16_t)(c.x >> 16);
          if (K > 2) tk[plane] = (uint16_t)c.y;
          if (K > 3) tk[2 * plane] = (uint16_t)(c.y >> 16);
          if (K > 4) tk[3 * plane] = (uint16_t)c.z;
          if (K > 5) tk[4 * plane] = (uint16_t)(c.z >> 16);
          if (K > 6) tk[5 * plane] = (uint16_t)c.w;
          if (K > 7) tk[6 * plane] = (uint16_t)(c.w >> 16);
        }
        if (kPlain) {
          out.reward[at] = __uint_as_float(e.x);
          out.discount[at] = done ? 0.0f : 1.0f;
          out.done[at] = (uint8_t)done;
        } else {
          if (out.reward) out.reward[at] = __uint_as_float(e.x);
          if (out.discount) out.discount[at] = __uint_as_float(discount_bits(discounts, dcode, done));
          if (out.done) out.done[at] = (uint8_t)done;
        }
        if (kPerf && out.perf) out.perf[at] = perf_tab[idx];
        ret = (over ? 0.0f : ret) + real_reward(__uint_as_float(e.x));
        over = (int)done;
        at += P;
      }
    }
  };
  const bool plain = !wp.has_dcodes && out.reward && out.discount && out.done &&
                     (!kPerf || out.perf);
  for (int t0 = 0; t0 < T; t0 += kWideAhead) {
    uint32_t a[kWideAhead];
#pragma unroll
    for (int j = 0; j < kWideAhead; ++j) a[j] = a_next[j];
    if (t0 + kWideAhead < T) fetch(t0 + kWideAhead, a_next);
    if (plain && t0 + kWideAhead <= T) {
      switch (K) {      // (one uniform branch per chunk of eight frames)
        case 1: chunk(std::integral_constant<int, 1>{}, t0, a); break;
        case 2: chunk(std::integral_constant<int, 2>{}, t0, a); break;
        case 3: chunk(std::integral_constant<int, 3>{}, t0, a); break;
        case 4: chunk(std::integral_constant<int, 4>{}, t0, a); break;
        case 5: chunk(std::integral_constant<int, 5>{}, t0, a); break;
        case 6: chunk(std::integral_constant<int, 6>{}, t0, a); break;
        case 7: chunk(std::integral_constant<int, 7>{}, t0, a); break;
        default: chunk(std::integral_constant<int, 8>{}, t0, a); break;
      }
    } else {
      chunk(std::integral_constant<int, 0>{}, t0, a);
    }
  }
  state[env] = (int32_t)now;
  st.done[env] = (uint8_t)over;
  if (st.ret) st.ret[env] = ret;
  report_bad_actions(out, bad);
}

// ---------------------------------------------------------------------------
// Engine.play() in ONE launch (T == 1, rows that are whole 16-byte chunks).
// A wave owns n consecutive environments - state, scalars and the n rows of the frame - so
// nothing it reads is written by another wave (the update + render pair needs the trace in
// between for exactly that reason).  Lanes 0 .. n-1 walk the table for one environment each
// and write its scalars; every lane then builds 16-byte chunks of the wave's span of the
// frame in registers: the scenery's chunk (rotation 0 of the render tables: rows start on
// 16-byte boundaries) with, per thing that shows, one byte set and one cleared.  No LDS image,
// no second kernel: 13-17 -> 6.5-7 us per call at small batches, where the three launches of the
// two-kernel path were what a call cost.
struct WideStepParams {
  int32_t n_states, n_dyn, cells, R, n_env;     // n_env: environments per wave
  uint32_t inv_r;                                // ceil(2^32 / R): x / R for x < 2^16 * ...
  int32_t dyn_off[CAMPX_WIDE_MAX_DYN];           // byte offset of thing d's layer inside a row
  int32_t dyn_char[CAMPX_WIDE_MAX_DYN];
  float discounts[16];
  // a scenery in variants (slot n_dyn of a state's entries names the variant): one set of rotations
  // per variant, these many bytes apart; planes of the trace (the things', and the variant's / the mask's)
  int32_t n_variants, n_planes;
  int64_t rot_obs_stride, rot_board_stride;
};

// slot K (1 .. 7) of a state's eight 16-bit entries
__device__ __forceinline__ uint32_t wide_slot(const u32x4& c, int K) {
  const uint32_t w[4] = {c.x, c.y, c.z, c.w};
  uint32_t v = 0u;
#pragma unroll
  for (int d = 1; d < CAMPX_WIDE_MAX_DYN; ++d) v = d == K ? (w[d >> 1] >> (16 * (d & 1))) & 0xffffu : v;
  return v;
}

constexpr int kStepEnvMax = 16;    // environments per wave at most

// One byte of a 16-byte chunk (chunk starts at byte `base` of the row; `at` is the byte's
// offset in the row): set to `val` when it falls inside.
__device__ __forceinline__ void poke(u32x4& v, int base, int at, uint32_t val) {
  const uint32_t rel = (uint32_t)(at - base);
  if (rel < 16u) {
    const uint32_t sh = (rel & 3u) * 8u, m = ~(0xffu << sh), b = val << sh;
    const uint32_t w = rel >> 2;
    v.x = w == 0u ? (v.x & m) | b : v.x;
    v.y = w == 1u ? (v.y & m) | b : v.y;
    v.z = w == 2u ? (v.z & m) | b : v.z;
    v.w = w == 3u ? (v.w & m) | b : v.w;
  }
}

// kFmt: CAMPX_OBS_INT8, or f16 / bf16 observations (a chunk's sixteen cells leave as two
// 16-byte stores of eight halves each); the flat board is always int8.
template <bool kBoard, bool kPerf, int kFmt>
__global__ __launch_bounds__(kWideThreads) void wide_step_kernel(
    WideStepParams sp, const uint2* __restrict__ entries, const u32x4* __restrict__ cells,
    const int8_t* __restrict__ perf_tab, const int8_t* __restrict__ rot_obs,
    const int8_t* __restrict__ rot_board, int32_t* __restrict__ state, CampxState st,
    const int8_t* __restrict__ actions, CampxOutputs out, int64_t B) {
  __shared__ u32x4 shown[kWideThreads / kWave][kStepEnvMax];   // per wave: its environments' trace entries
  // (readfirstlane: the wave index is uniform, and saying so keeps the span's base address
  // in scalar registers for the stores)
  const int lane = threadIdx.x & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = sp.n_env, K = sp.n_dyn;
  const int64_t env0 = ((int64_t)blockIdx.x * (kWideThreads / kWave) + wave) * n;
  if (env0 >= B) return;                          // (wave-uniform; no workgroup barrier below)
  const int n_here = (B - env0 < n) ? (int)(B - env0) : n;
  // ---- update pass: lane i < n_here owns environment env0 + i
  int bad = 0;
  if (lane < n_here) {
    const int64_t env = env0 + lane;
    uint32_t now = (uint32_t)state[env];
    now = now < (uint32_t)sp.n_states ? now : 0u;
    const int over = st.done[env];
    const uint32_t a = (uint8_t)actions[env];
    bad = a > 4u;
    const uint32_t idx = (over ? 0u : now) * CAMPX_N_ACTIONS + (a > 4u ? 4u : a);
    const uint2 e = entries[idx];
    now = e.y & 0xffffffu;
    const uint32_t done = (e.y >> 24) & 1u, dcode = (e.y >> 25) & 15u;
    const u32x4 c = cells[now];
    shown[wave][lane] = c;
    state[env] = (int32_t)now;
    st.done[env] = (uint8_t)done;
    if (st.ret) st.ret[env] = (over ? 0.0f : st.ret[env]) + real_reward(__uint_as_float(e.x));
    if (out.reward) out.reward[env] = __uint_as_float(e.x);
    if (out.discount)
      out.discount[env] = __uint_as_float(dcode ? __float_as_uint(sp.discounts[dcode])
                                                : (done ? 0u : 0x3f800000u));
    if (out.done) out.done[env] = (uint8_t)done;
    if (kPerf && out.perf) out.perf[env] = perf_tab[idx];
    uint16_t* trace = reinterpret_cast<uint16_t*>(out.trace);
    const int64_t P = row_pitch(out, B);
    const uint32_t w[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
    for (int d = 0; d < CAMPX_WIDE_MAX_DYN; ++d)
      if (d < sp.n_planes) trace[(int64_t)d * P + env] = (uint16_t)(w[d >> 1] >> (16 * (d & 1)));
  }
  report_bad_actions(out, bad);
  // (the wave's own LDS writes are visible to its own later reads: one wave, in order)

  // ---- render: the wave's span of the frame, 16 bytes per lane per round
  auto rows = [&](int8_t* dst, const int8_t* rot, int R, bool board, int64_t stride) {
    const int span = n_here * R;                  // bytes, a multiple of 16
    for (int byte = lane * 16; byte < span; byte += kWave * 16) {
      const int el = (int)(((uint64_t)(uint32_t)byte * sp.inv_r) >> 32);   // byte / R (R = sp.R)
      const int e_local = board ? byte / R : el;
      const int off = byte - e_local * R;
      const u32x4 c = shown[wave][e_local];
      // (a scenery in variants: the environment's own set of rotations; rows are whole chunks, so a
      // chunk never runs into the next environment's)
      const int64_t from = sp.n_variants > 1 ? (int64_t)wide_slot(c, K) * stride : 0;
      u32x4 v = *reinterpret_cast<const u32x4*>(rot + from + off);
      const uint32_t w[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
      for (int d = 0; d < CAMPX_WIDE_MAX_DYN; ++d) {
        if (d < K) {
          const uint32_t t = (w[d >> 1] >> (16 * (d & 1))) & 0xffffu;
          if (t >> 15) {
            const int cell = (int)(t & 0x3ffu);
            if (board) {
              poke(v, off, cell, (uint32_t)(uint8_t)sp.dyn_char[d]);
            } else {
              poke(v, off, (int)((t >> 10) & 15u) * sp.cells + cell, 0u);
              poke(v, off, sp.dyn_off[d] + cell, 1u);
            }
          }
        }
      }
      if (kFmt == 0 || board) {
        store16_streaming(reinterpret_cast<u32x4*>(dst + byte), v);
      } else {
        constexpr uint32_t kOne = (kFmt == 1) ? 0x3C00u : 0x3F80u;
        const uint32_t b[4] = {v.x, v.y, v.z, v.w};
        u32x4 h[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          h[i].x = ((b[2 * i] & 0xffu) | ((b[2 * i] << 8) & 0x00ff0000u)) * kOne;
          h[i].y = (((b[2 * i] >> 16) & 0xffu) | ((b[2 * i] >> 8) & 0x00ff0000u)) * kOne;
          h[i].z = ((b[2 * i + 1] & 0xffu) | ((b[2 * i + 1] << 8) & 0x00ff0000u)) * kOne;
          h[i].w = (((b[2 * i + 1] >> 16) & 0xffu) | ((b[2 * i + 1] >> 8) & 0x00ff0000u)) * kOne;
        }
        store16_streaming(reinterpret_cast<u32x4*>(dst + 2 * (int64_t)byte), h[0]);
        store16_streaming(reinterpret_cast<u32x4*>(dst + 2 * (int64_t)byte + 16), h[1]);
      }
    }
  };
  rows(out.obs + env0 * sp.R * (kFmt ? 2 : 1), rot_obs, sp.R, false, sp.rot_obs_stride);
  if (kBoard) rows(out.board + env0 * sp.cells, rot_board, sp.cells, true, sp.rot_board_stride);
}

// Engine.play() in one launch for the games wide_step_kernel cannot take: rows that are not whole
// 16-byte chunks (a 6x10 coin field: 300 bytes) and a scenery of pieces (round 6).  The same
// ownership - a wave, n consecutive environments, n chosen so that its span of the frame STARTS
// on a 16-byte boundary - but the span is built in LDS the way render_kernel builds its windows:
// the scenery's chunks from the rotations (which continue a row with its own start - the next
// environment's row, the scenery being the same), then every environment's lane writes the bytes
// its things and pieces set and clear, then aligned 16-byte stores; the last bytes of a span that
// is not whole chunks (the frame's end) leave one by one.  (A first form that poked the patches
// into the chunks' registers - two environments per chunk, sixteen unrolled pieces - ran 24-29 us per
// call for the coin field at B = 65 536 where the update + render pair takes 13.)
constexpr int kStepLdsSpan = 4096;   // bytes of a wave's span at most

struct WideStepLdsParams {
  int32_t n_pieces, n_planes;                    // pieces of the scenery; planes of the trace
  int32_t pitch_obs, pitch_board;                // bytes from one rotation of the scenery row to the next
  uint32_t piece_obs[CAMPX_WIDE_MAX_PIECES];     // byte a piece sets | byte of the scenery it clears << 16
  uint32_t piece_board[CAMPX_WIDE_MAX_PIECES];   // cell | character << 16
};

template <bool kBoard, bool kPerf, int kFmt>
__global__ __launch_bounds__(kWideThreads) void wide_step_lds_kernel(
    WideStepParams sp, WideStepLdsParams lp, const uint2* __restrict__ entries,
    const u32x4* __restrict__ cells, const int8_t* __restrict__ perf_tab,
    const int8_t* __restrict__ rot_obs, const int8_t* __restrict__ rot_board,
    int32_t* __restrict__ state, CampxState st, const int8_t* __restrict__ actions, CampxOutputs out,
    int64_t B) {
  __shared__ __attribute__((aligned(16))) int8_t win_all[kWideThreads / kWave][kStepLdsSpan + 16];
  __shared__ uint16_t variant_all[kWideThreads / kWave][kStepEnvMax + 1];   // (a scenery in variants)
  const int lane = threadIdx.x & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = sp.n_env, K = sp.n_dyn;
  const int64_t env0 = ((int64_t)blockIdx.x * (kWideThreads / kWave) + wave) * n;
  if (env0 >= B) return;                          // (wave-uniform; no workgroup barrier below)
  const int n_here = (B - env0 < n) ? (int)(B - env0) : n;
  // ---- update pass: lane i < n_here owns environment env0 + i (as in wide_step_kernel)
  int bad = 0;
  u32x4 mine = {0u, 0u, 0u, 0u};                  // where this lane's environment's things show
  if (lane < n_here) {
    const int64_t env = env0 + lane;
    uint32_t now = (uint32_t)state[env];
    now = now < (uint32_t)sp.n_states ? now : 0u;
    const int over = st.done[env];
    const uint32_t a = (uint8_t)actions[env];
    bad = a > 4u;
    const uint32_t idx = (over ? 0u : now) * CAMPX_N_ACTIONS + (a > 4u ? 4u : a);
    const uint2 e = entries[idx];
    now = e.y & 0xffffffu;
    const uint32_t done = (e.y >> 24) & 1u, dcode = (e.y >> 25) & 15u;
    mine = cells[now];
    state[env] = (int32_t)now;
    st.done[env] = (uint8_t)done;
    if (st.ret) st.ret[env] = (over ? 0.0f : st.ret[env]) + real_reward(__uint_as_float(e.x));
    if (out.reward) out.reward[env] = __uint_as_float(e.x);
    if (out.discount)
      out.discount[env] = __uint_as_float(dcode ? __float_as_uint(sp.discounts[dcode])
                                                : (done ? 0u : 0x3f800000u));
    if (out.done) out.done[env] = (uint8_t)done;
    if (kPerf && out.perf) out.perf[env] = perf_tab[idx];
    uint16_t* trace = reinterpret_cast<uint16_t*>(out.trace);
    const int64_t P = row_pitch(out, B);
    const uint32_t w[4] = {mine.x, mine.y, mine.z, mine.w};
#pragma unroll
    for (int d = 0; d < CAMPX_WIDE_MAX_DYN; ++d)
      if (d < lp.n_planes) trace[(int64_t)d * P + env] = (uint16_t)(w[d >> 1] >> (16 * (d & 1)));
    if (sp.n_variants > 1) variant_all[wave][lane] = (uint16_t)wide_slot(mine, K);
  }
  report_bad_actions(out, bad);

  // ---- render: the wave's span of the frame, built in LDS
  int8_t* const win = win_all[wave];
  auto rows = [&](int8_t* dst, const int8_t* rot, int R, bool board, int rot_pitch, int64_t stride) {
    const int span = n_here * R;                  // bytes; starts on a 16-byte boundary of `dst`
    for (int byte = lane * 16; byte < span; byte += kWave * 16) {
      const int e = board ? byte / R : (int)(((uint64_t)(uint32_t)byte * sp.inv_r) >> 32);
      const int off = byte - e * R;
      const int8_t* here = rot + (off & 15) * rot_pitch + (off & ~15);
      u32x4 v;
      if (sp.n_variants > 1) {
        // the environment's own variant; a chunk that runs over the end of its row takes the rest from
        // the NEXT environment's (as render_kernel<kVar> does: the same offset in that variant's rotations)
        const int v0 = variant_all[wave][e], v1 = variant_all[wave][e + 1 < n_here ? e + 1 : e];
        v = variant_chunk(here, stride, v0);
        const int left = R - off;
        if (left < 16 && v1 != v0) v = merge_rows(v, variant_chunk(here, stride, v1), left);
      } else {
        v = *reinterpret_cast<const u32x4*>(here);
      }
      *reinterpret_cast<u32x4*>(win + byte) = v;
    }
    // (the wave's own LDS operations complete in order: the chunks above are in place)
    if (lane < n_here) {
      int8_t* const row = win + lane * R;
      const uint32_t w[4] = {mine.x, mine.y, mine.z, mine.w};
#pragma unroll
      for (int d = 0; d < CAMPX_WIDE_MAX_DYN; ++d) {
        if (d < K) {
          const uint32_t t = (w[d >> 1] >> (16 * (d & 1))) & 0xffffu;
          if (t >> 15) {
            const int cell = (int)(t & 0x3ffu);
            if (board) {
              row[cell] = (int8_t)sp.dyn_char[d];
            } else {
              row[(int)((t >> 10) & 15u) * sp.cells + cell] = 0;
              row[sp.dyn_off[d] + cell] = 1;
            }
          }
        }
      }
      if (lp.n_pieces > 0) {
        const uint32_t mask = wide_slot(mine, K);   // slot n_dyn of the entries: the pieces that show
#pragma unroll
        for (int p = 0; p < CAMPX_WIDE_MAX_PIECES; ++p) {
          if (p < lp.n_pieces && ((mask >> p) & 1u)) {
            if (board) {
              row[lp.piece_board[p] & 0xffffu] = (int8_t)(lp.piece_board[p] >> 16);
            } else {
              row[lp.piece_obs[p] >> 16] = 0;
              row[lp.piece_obs[p] & 0xffffu] = 1;
            }
          }
        }
      }
    }
    for (int byte = lane * 16; byte < span; byte += kWave * 16) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(win + byte);
      const int left = span - byte;               // (only the frame's last span can end short)
      if (kFmt == 0 || board) {
        if (left >= 16) {
          store16_streaming(reinterpret_cast<u32x4*>(dst + byte), v);
        } else {
          for (int i = 0; i < left; ++i) dst[byte + i] = win[byte + i];
        }
      } else {
        constexpr uint32_t kOne = (kFmt == 1) ? 0x3C00u : 0x3F80u;
        const uint32_t b[4] = {v.x, v.y, v.z, v.w};
        u32x4 h[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          h[i].x = ((b[2 * i] & 0xffu) | ((b[2 * i] << 8) & 0x00ff0000u)) * kOne;
          h[i].y = (((b[2 * i] >> 16) & 0xffu) | ((b[2 * i] >> 8) & 0x00ff0000u)) * kOne;
          h[i].z = ((b[2 * i + 1] & 0xffu) | ((b[2 * i + 1] << 8) & 0x00ff0000u)) * kOne;
          h[i].w = (((b[2 * i + 1] >> 16) & 0xffu) | ((b[2 * i + 1] >> 8) & 0x00ff0000u)) * kOne;
        }
        if (left >= 16) {
          store16_streaming(reinterpret_cast<u32x4*>(dst + 2 * (int64_t)byte), h[0]);
          store16_streaming(reinterpret_cast<u32x4*>(dst + 2 * (int64_t)byte + 16), h[1]);
        } else {
          uint16_t* to = reinterpret_cast<uint16_t*>(dst) + byte;
          for (int i = 0; i < left; ++i) to[i] = (uint16_t)((uint32_t)(uint8_t)win[byte + i] * kOne);
        }
      }
    }
  };
  rows(out.obs + env0 * sp.R * (kFmt ? 2 : 1), rot_obs, sp.R, false, lp.pitch_obs, sp.rot_obs_stride);
  if (kBoard) rows(out.board + env0 * sp.cells, rot_board, sp.cells, true, lp.pitch_board, sp.rot_board_stride);
}

// its_showtime(): state 0 and the trace rows of the first observation.
__global__ void wide_reset_kernel(const u32x4* __restrict__ cells, int32_t K, int32_t* __restrict__ state,
                                  CampxState st, uint16_t* __restrict__ trace, int64_t P, int64_t B) {
  const int64_t env = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= B) return;
  const u32x4 c = cells[0];
  state[env] = 0;
  st.done[env] = 0;
  if (st.ret) st.ret[env] = 0.0f;
  trace[env] = (uint16_t)c.x;
  if (K > 1) trace[P + env] = (uint16_t)(c.x >> 16);
  if (K > 2) trace[2 * P + env] = (uint16_t)c.y;
  if (K > 3) trace[3 * P + env] = (uint16_t)(c.y >> 16);
  if (K > 4) trace[4 * P + env] = (uint16_t)c.z;
  if (K > 5) trace[5 * P + env] = (uint16_t)(c.z >> 16);
  if (K > 6) trace[6 * P + env] = (uint16_t)c.w;
  if (K > 7) trace[7 * P + env] = (uint16_t)(c.w >> 16);
}

// ---- the table blob: [entries uint2 x 5S, padded to 16][cells 8 x uint16 x S][perf int8 x 5S, padded to 16]
// [rot_obs][rot_board][pieces: 16 words for the layered rows, 16 for the flat board]
struct WideLayout {
  int64_t n_entries, cells_off, perf_off, rot_obs_off, rot_board_off, pieces_off, total;
  int pitch_obs, pitch_board;
  int n_variants;          // sets of rotations (1: the scenery never changes)
  int n_planes;            // planes of the trace: the things, plus the variant's when there are several,
                           // or the mask of the pieces that show
};

static int wide_variants(const CampxWideSpec& s) { return s.n_variants > 1 ? s.n_variants : 1; }

WideLayout wide_layout(const CampxWideSpec& s) {
  WideLayout w;
  const int64_t HW = (int64_t)s.rows * s.cols, R = HW * s.n_layers, S = s.n_states;
  w.n_entries = S * CAMPX_N_ACTIONS;
  w.cells_off = (w.n_entries * (int64_t)sizeof(uint2) + 15) & ~(int64_t)15;
  w.perf_off = w.cells_off + S * (int64_t)sizeof(u32x4);
  w.rot_obs_off = (w.perf_off + w.n_entries + 15) & ~(int64_t)15;
  w.pitch_obs = (int)(((R + 15) & ~(int64_t)15) + 16);
  w.n_variants = wide_variants(s);
  w.n_planes = s.n_dyn + ((w.n_variants > 1 || s.n_pieces > 0) ? 1 : 0);
  w.rot_board_off = w.rot_obs_off + 16ll * w.pitch_obs * w.n_variants;
  w.pitch_board = (int)(((HW + 15) & ~(int64_t)15) + 16);
  w.pieces_off = w.rot_board_off + 16ll * w.pitch_board * w.n_variants;
  w.total = w.pieces_off + (s.n_pieces > 0 ? 2 * CAMPX_WIDE_MAX_PIECES * (int64_t)sizeof(uint32_t) : 0);
  return w;
}

RenderSource wide_render_source(const CampxWideSpec& s, const void* tables_dev) {
  const WideLayout w = wide_layout(s);
  RenderSource src;
  memset(&src, 0, sizeof(src));
  src.rows = s.rows;
  src.cols = s.cols;
  src.n_layers = s.n_layers;
  src.n_dyn = s.n_dyn;
  for (int d = 0; d < s.n_dyn; ++d) src.dyn_layer[d] = s.dyn_layer[d];
  memcpy(src.layer_char, s.layer_char, sizeof(src.layer_char));
  const char* blob = static_cast<const char*>(tables_dev);
  src.rot_obs = reinterpret_cast<const int8_t*>(blob + w.rot_obs_off);
  src.rot_board = reinterpret_cast<const int8_t*>(blob + w.rot_board_off);
  src.top_layer = nullptr;
  src.wide = true;
  src.n_variants = w.n_variants;
  src.rot_obs_stride = 16ll * w.pitch_obs;
  src.rot_board_stride = 16ll * w.pitch_board;
  src.n_pieces = s.n_pieces;
  if (s.n_pieces > 0) {
    src.pieces_obs = reinterpret_cast<const uint32_t*>(blob + w.pieces_off);
    src.pieces_board = src.pieces_obs + CAMPX_WIDE_MAX_PIECES;
  }
  return src;
}

// Frames back to back, or - strides 0 - only the last one.
bool wide_last_only(const CampxOutputs& out) {
  return out.obs_t_stride == 0 && (!out.board || out.board_t_stride == 0) &&
         out.obs_format == CAMPX_OBS_INT8;
}

// The plain fields of a spec (what a launch relies on).
int32_t wide_validate_plain(const CampxWideSpec* s) {
  if (!s) return CAMPX_EINVAL;
  if (s->magic != CAMPX_SPEC_MAGIC || s->version != CAMPX_SPEC_VERSION) return CAMPX_ESPEC;
  if (s->rows < 1 || s->cols < 1 || s->rows > 127 || s->cols > 127) return CAMPX_ESPEC;
  const int HW = s->rows * s->cols;
  if (HW < 16 || HW > CAMPX_WIDE_MAX_CELLS) return CAMPX_ESPEC;
  if (s->n_layers < 1 || s->n_layers > CAMPX_MAX_LAYERS) return CAMPX_ESPEC;
  if (s->n_dyn < 1 || s->n_dyn > CAMPX_WIDE_MAX_DYN) return CAMPX_ESPEC;
  if (s->n_states < 1 || s->n_states > CAMPX_WIDE_MAX_STATES) return CAMPX_ESPEC;
  for (int d = 0; d < s->n_dyn; ++d)
    if (s->dyn_layer[d] < 0 || s->dyn_layer[d] >= s->n_layers) return CAMPX_ESPEC;
  if ((s->has_perf | s->any_reward | s->any_dcode) & ~1) return CAMPX_ESPEC;
  for (int i = 0; i < HW; ++i)
    if (s->static_top_layer[i] >= s->n_layers) return CAMPX_ESPEC;
  // a scenery of several variants takes one plane of the trace (and one slot of a state's entries)
  if (s->n_variants < 0 || s->n_variants > CAMPX_WIDE_MAX_VARIANTS) return CAMPX_ESPEC;
  if (s->n_variants > 1 && s->n_dyn > CAMPX_WIDE_MAX_DYN - 1) return CAMPX_ESPEC;
  // ... and so does the mask of the pieces that show; one or the other
  if (s->n_pieces < 0 || s->n_pieces > CAMPX_WIDE_MAX_PIECES) return CAMPX_ESPEC;
  if (s->n_pieces > 0 && (s->n_variants > 1 || s->n_dyn > CAMPX_WIDE_MAX_DYN - 1)) return CAMPX_ESPEC;
  for (int p = 0; p < s->n_pieces; ++p) {
    if (s->piece_cell[p] >= HW || s->piece_layer[p] >= s->n_layers) return CAMPX_ESPEC;
    // (a piece that paints the scenery's own layer there would set and clear the same byte)
    if (s->piece_layer[p] == s->static_top_layer[s->piece_cell[p]]) return CAMPX_ESPEC;
  }
  return CAMPX_OK;
}

int32_t wide_check(const CampxWideSpec* s, const void* tables, const CampxState& st,
                   const CampxOutputs& out, int64_t B, int32_t T) {
  if (!s || !tables || !st.pos || !st.done || !out.obs || !out.trace || B <= 0 || T < 0)
    return CAMPX_EINVAL;
  if ((reinterpret_cast<uintptr_t>(out.obs) & 15) || (reinterpret_cast<uintptr_t>(out.trace) & 1) ||
      (reinterpret_cast<uintptr_t>(st.pos) & 3))
    return CAMPX_EINVAL;
  if (out.scalar_pitch && out.scalar_pitch < B) return CAMPX_EINVAL;
  if (out.obs_format < CAMPX_OBS_INT8 || out.obs_format > CAMPX_OBS_BF16) return CAMPX_EINVAL;
  const int32_t v = wide_validate_plain(s);
  if (v != CAMPX_OK) return v;
  if (out.perf && !s->has_perf) return CAMPX_EINVAL;
  const int64_t HW = (int64_t)s->rows * s->cols, LHW = HW * s->n_layers;
  if (B * LHW >= (1ll << 32) - 65536) return CAMPX_EINVAL;
  const bool every = out.obs_t_stride == B * LHW && (!out.board || out.board_t_stride == B * HW);
  if (T > 0 && !every && !wide_last_only(out)) return CAMPX_EINVAL;
  return CAMPX_OK;
}

// `plane`: distance between two things' planes of the trace, in entries.
int32_t wide_renders(const CampxWideSpec& s, const void* tables_dev, const uint16_t* trace,
                     CampxOutputs out, int64_t B, int32_t T, int64_t plane, hipStream_t stream) {
  const RenderSource src = wide_render_source(s, tables_dev);
  const int64_t pitch = row_pitch(out, B);
  const int64_t elem = out.obs_format == CAMPX_OBS_INT8 ? 1 : 2;
  if (wide_last_only(out)) {
    trace += (int64_t)(T - 1) * pitch;
    T = 1;
  }
  // (a render launch has one grid row per frame: at most 65 535 of them)
  for (int64_t t0 = 0; t0 < T; t0 += 65520) {
    const int32_t n = (int32_t)(T - t0 < 65520 ? T - t0 : 65520);
    int32_t rc = launch_render_from(src, trace + t0 * pitch, out.obs + t0 * out.obs_t_stride * elem,
                                    B, n, plane, pitch, false, out.obs_format, stream);
    if (rc != CAMPX_OK) return rc;
    if (out.board) {
      rc = launch_render_from(src, trace + t0 * pitch, out.board + t0 * out.board_t_stride, B, n,
                              plane, pitch, true, 0, stream);
      if (rc != CAMPX_OK) return rc;
    }
  }
  return CAMPX_OK;
}


// ---------------------------------------------------------------------------
// One frame of a RULE game's update pass for N given states under each of the five actions:
// the building block of the device-side state enumeration (include/campx_hip.h
// campx_wide_enumerate_launch).  The rules are rollout_kernel's (k_interp.hip: same CampxRule
// records, same order of reads and writes, `shown` = where things stood at the latest repaint,
// campx/engine.py:195-208) with cells of up to ten bits and the scenery tables in global memory:
// one thread per (state, action), nothing shared, nothing kept.
struct EnumParams {
  int32_t rows, cols, n_rules, any_reward;
  int32_t perf_dyn, perf_n, perf_mode, perf_mask, perf_scale, perf_offset;
  int32_t dyn_layer[CAMPX_MAX_DYN], dyn_z[CAMPX_MAX_DYN];
  CampxRule rules[CAMPX_MAX_RULES];
};

template <int K>
__global__ __launch_bounds__(256) void wide_enumerate_kernel(
    EnumParams ep, const uint8_t* __restrict__ top_layer, const uint8_t* __restrict__ top_z,
    const uint16_t* __restrict__ cover, const uint8_t* __restrict__ cell_class,
    const uint16_t* __restrict__ cells_in, int64_t N, uint16_t* __restrict__ next_cells,
    float* __restrict__ reward_out, uint8_t* __restrict__ done_out, uint8_t* __restrict__ shows_out,
    int8_t* __restrict__ perf_out) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * CAMPX_N_ACTIONS) return;
  const int64_t s = idx / CAMPX_N_ACTIONS;
  const int a = (int)(idx - s * CAMPX_N_ACTIONS);
  const int H = ep.rows, W = ep.cols;
  int pr[K], pc[K], sr[K], sc[K];   // now / at the latest repaint (indexed through select chains)
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int cell = cells_in[s * K + k];
    pr[k] = sr[k] = cell / W;
    pc[k] = sc[k] = cell - (cell / W) * W;
  }
  auto get = [&](const int (&v)[K], int d) {
    int x = v[0];
#pragma unroll
    for (int k = 1; k < K; ++k) x = (d == k) ? v[k] : x;
    return x;
  };
  auto put2 = [&](int d, int r, int c) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      pr[k] = (d == k) ? r : pr[k];
      pc[k] = (d == k) ? c : pc[k];
    }
  };
  // layer shown at `cell` when the things stand at (r, c): engine.py:306-324
  auto shown_at = [&](int cell, const int (&r)[K], const int (&c)[K]) {
    int layer = top_layer[cell], z = top_z[cell];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const bool here = (r[k] * W + c[k] == cell) && (ep.dyn_z[k] > z);
      layer = here ? ep.dyn_layer[k] : layer;
      z = here ? ep.dyn_z[k] : z;
    }
    return layer;
  };
  const int perf_from = ep.perf_dyn >= 0 ? get(pr, ep.perf_dyn) * W + get(pc, ep.perf_dyn) : 0;
  float reward = 0.0f;
  bool first = true;
  int over = 0;
  auto add_reward = [&](float r) {  // plot.py:208-211: r + total
    reward = first ? r : r + reward;
    first = false;
  };
  for (int i = 0; i < ep.n_rules; ++i) {
    const CampxRule& R = ep.rules[i];
    const int d = R.dyn;
    switch (R.op) {
      case CAMPX_OP_AGENT: {
        int r2, c2;
        moved(a, H, W, get(pr, d), get(pc, d), r2, c2);
        const int target = shown_at(r2 * W + c2, sr, sc);
        const bool blocked = (R.block_layers >> target) & 1u;
        r2 = blocked ? get(sr, d) : r2;
        c2 = blocked ? get(sc, d) : c2;
        put2(d, r2, c2);
        if (R.has_reward) {
          float r = R.base;
          if (R.reward_layers) r += (float)((R.reward_layers >> shown_at(r2 * W + c2, sr, sc)) & 1u);
          add_reward(r);
        }
        break;
      }
      case CAMPX_OP_DIR_HOVER: {
        const int under = shown_at(get(pr, d) * W + get(pc, d), sr, sc);
        add_reward(R.base + ((under == R.aux) ? 1.0f : 0.0f) * R.bonus[a]);
        break;
      }
      case CAMPX_OP_BOX: {
        int ar, ac, br, bc;
        moved(a, H, W, get(sr, R.aux), get(sc, R.aux), ar, ac);
        const int box_r = get(pr, d), box_c = get(pc, d);
        moved(a, H, W, box_r, box_c, br, bc);
        const int beyond = shown_at(br * W + bc, sr, sc);
        const bool go = (ar == box_r) && (ac == box_c) && !((R.block_layers >> beyond) & 1u);
        put2(d, go ? br : box_r, go ? bc : box_c);
        break;
      }
      case CAMPX_OP_GOAL: {
        const int arrived = (cover[get(pr, d) * W + get(pc, d)] >> R.aux) & 1;
        add_reward(R.base + (float)arrived * R.bonus[0]);
        if (arrived) over = 1;  // plot.py:183-184
        break;
      }
      default:
        break;
    }
    if (R.end_group) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        sr[k] = pr[k];
        sc[k] = pc[k];
      }
    }
  }
  if (!ep.any_reward) reward = __builtin_nanf("");
  uint32_t shows = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int cell = pr[k] * W + pc[k];
    next_cells[idx * K + k] = (uint16_t)cell;
    shows |= (uint32_t)(shown_at(cell, pr, pc) == ep.dyn_layer[k]) << k;
  }
  reward_out[idx] = reward;
  done_out[idx] = (uint8_t)over;
  shows_out[idx] = (uint8_t)shows;
  if (perf_out) {
    int code = 0;
    if (ep.perf_dyn >= 0) {
      if (ep.perf_mode == 0) {
        const int to = get(pr, ep.perf_dyn) * W + get(pc, ep.perf_dyn);
        code = class_progress(cell_class[perf_from], cell_class[to], ep.perf_n) + 1;
      } else {
#pragma unroll
        for (int k = 0; k < K; ++k)
          code += ((ep.perf_mask >> k) & 1) ? (int)cell_class[pr[k] * W + pc[k]] : 0;
      }
    }
    perf_out[idx] = (int8_t)(code * ep.perf_scale + ep.perf_offset);
  }
}

int32_t launch_wide_enumerate(const CampxWideRules& r, const uint16_t* cells_in, int64_t N,
                              uint16_t* next_cells, float* reward, uint8_t* done, uint8_t* shows,
                              int8_t* perf, hipStream_t stream) {
  EnumParams ep;
  memset(&ep, 0, sizeof(ep));
  ep.rows = r.rows;
  ep.cols = r.cols;
  ep.n_rules = r.n_rules;
  ep.any_reward = r.any_reward;
  ep.perf_dyn = r.perf_dyn;
  ep.perf_n = r.perf_n;
  ep.perf_mode = r.perf_mode;
  ep.perf_mask = r.perf_mask;
  ep.perf_scale = r.perf_scale;
  ep.perf_offset = r.perf_offset;
  memcpy(ep.dyn_layer, r.dyn_layer, sizeof(ep.dyn_layer));
  memcpy(ep.dyn_z, r.dyn_z, sizeof(ep.dyn_z));
  memcpy(ep.rules, r.rules, sizeof(ep.rules));
  const int64_t threads = N * CAMPX_N_ACTIONS;
  const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
#define CAMPX_ENUM(KK)                                                                          \
  hipLaunchKernelGGL((wide_enumerate_kernel<KK>), grid, block, 0, stream, ep, r.top_layer,      \
                     r.top_z, r.cover, r.cell_class, cells_in, N, next_cells, reward, done, shows, perf)
  switch (r.n_dyn) {
    case 1: CAMPX_ENUM(1); break;
    case 2: CAMPX_ENUM(2); break;
    case 3: CAMPX_ENUM(3); break;
    default: CAMPX_ENUM(4); break;
  }
#undef CAMPX_ENUM
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? CAMPX_OK : hip_failed(e);
}

}  // namespace campx_impl

using namespace campx_impl;

extern "C" {

int32_t campx_wide_spec_size(void) { return (int32_t)sizeof(CampxWideSpec); }

int32_t campx_wide_spec_validate(const CampxWideSpec* s) {
  const int32_t v = wide_validate_plain(s);
  if (v != CAMPX_OK) return v;
  const int HW = s->rows * s->cols, S = s->n_states, K = s->n_dyn;
  if (s->state_cells)
    for (int64_t i = 0; i < (int64_t)S * K; ++i)
      if ((s->state_cells[i] & 0x3ffu) >= (uint32_t)HW || (s->state_cells[i] & 0x7c00u)) return CAMPX_ESPEC;
  if (s->next_state)
    for (int64_t i = 0; i < (int64_t)S * CAMPX_N_ACTIONS; ++i)
      if (s->next_state[i] < 0 || s->next_state[i] >= S) return CAMPX_ESPEC;
  if (s->done)
    for (int64_t i = 0; i < (int64_t)S * CAMPX_N_ACTIONS; ++i)
      if ((s->done[i] & 0x0eu) || ((s->done[i] >> 4) && !s->any_dcode)) return CAMPX_ESPEC;
  if (s->n_variants > 1) {
    if (s->variant_top_layer) {
      for (int64_t i = 0; i < (int64_t)s->n_variants * HW; ++i)
        if (s->variant_top_layer[i] >= s->n_layers) return CAMPX_ESPEC;
      for (int i = 0; i < HW; ++i)     // (variant 0 is the scenery of the plain fields)
        if (s->variant_top_layer[i] != s->static_top_layer[i]) return CAMPX_ESPEC;
    }
    if (s->state_variant)
      for (int64_t i = 0; i < S; ++i)
        if (s->state_variant[i] >= s->n_variants) return CAMPX_ESPEC;
  }
  if (s->n_pieces > 0 && s->state_pieces)
    for (int64_t i = 0; i < S; ++i)
      if (s->state_pieces[i] >> s->n_pieces) return CAMPX_ESPEC;
  return CAMPX_OK;
}

int64_t campx_wide_tables_bytes(const CampxWideSpec* s) {
  if (wide_validate_plain(s) != CAMPX_OK) return 0;
  return wide_layout(*s).total;
}

int32_t campx_wide_tables_build(const CampxWideSpec* s, void* tables_dev, void* stream) {
  if (!s || !tables_dev) return CAMPX_EINVAL;
  if (!s->state_cells || !s->next_state || !s->reward || !s->done) return CAMPX_EINVAL;
  if (s->has_perf && !s->perf) return CAMPX_EINVAL;
  if (s->n_variants > 1 && (!s->variant_top_layer || !s->state_variant)) return CAMPX_EINVAL;
  if (s->n_pieces > 0 && !s->state_pieces) return CAMPX_EINVAL;
  const int32_t v = campx_wide_spec_validate(s);
  if (v != CAMPX_OK) return v;
  const WideLayout w = wide_layout(*s);
  const int HW = s->rows * s->cols, R = HW * s->n_layers, S = s->n_states, K = s->n_dyn;
  char* blob = static_cast<char*>(calloc(1, (size_t)w.total));
  if (!blob) return CAMPX_ENOMEM;
  uint2* entries = reinterpret_cast<uint2*>(blob);
  u32x4* cells = reinterpret_cast<u32x4*>(blob + w.cells_off);
  int8_t* perf = reinterpret_cast<int8_t*>(blob + w.perf_off);
  for (int64_t i = 0; i < (int64_t)S * CAMPX_N_ACTIONS; ++i) {
    uint32_t bits;
    memcpy(&bits, &s->reward[i], 4);
    entries[i].x = bits;
    entries[i].y = wide_pack((uint32_t)s->next_state[i], s->done[i] & 1u, (uint32_t)(s->done[i] >> 4));
    perf[i] = s->perf ? s->perf[i] : 0;
  }
  const int V = w.n_variants;
  for (int st = 0; st < S; ++st) {
    uint32_t e[CAMPX_WIDE_MAX_DYN] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    // (what a thing covers is the scenery of the STATE's variant)
    const int variant = V > 1 ? s->state_variant[st] : 0;
    const uint8_t* top = V > 1 ? s->variant_top_layer + (int64_t)variant * HW : s->static_top_layer;
    for (int d = 0; d < K; ++d) {
      const uint32_t c = s->state_cells[(int64_t)st * K + d];
      const uint32_t cell = c & 0x3ffu;
      e[d] = cell | ((uint32_t)top[cell] << 10) | ((c >> 15) ? 0u : 0x8000u);
    }
    if (V > 1) e[K] = (uint32_t)variant;       // plane K of the trace: never painted (bit 15 clear)
    if (s->n_pieces > 0) e[K] = s->state_pieces[st];   // (or the pieces that show: all sixteen bits)
    cells[st].x = e[0] | (e[1] << 16);
    cells[st].y = e[2] | (e[3] << 16);
    cells[st].z = e[4] | (e[5] << 16);
    cells[st].w = e[6] | (e[7] << 16);
  }
  // the scenery's row (layers by equality, campx/rendering.py:204-215) and its rotations - per variant
  int8_t* row = static_cast<int8_t*>(calloc(1, (size_t)R + HW));
  if (!row) {
    free(blob);
    return CAMPX_ENOMEM;
  }
  int8_t* brow = row + R;
  for (int variant = 0; variant < V; ++variant) {
    const uint8_t* top = V > 1 ? s->variant_top_layer + (int64_t)variant * HW : s->static_top_layer;
    memset(row, 0, (size_t)R + HW);
    for (int i = 0; i < HW; ++i) {
      row[(int)top[i] * HW + i] = 1;
      brow[i] = (int8_t)s->layer_char[top[i]];
    }
    int8_t* rot_obs = reinterpret_cast<int8_t*>(blob + w.rot_obs_off) + 16ll * w.pitch_obs * variant;
    int8_t* rot_board = reinterpret_cast<int8_t*>(blob + w.rot_board_off) + 16ll * w.pitch_board * variant;
    for (int r = 0; r < 16; ++r) {
      for (int j = 0; j < w.pitch_obs; ++j) rot_obs[(int64_t)r * w.pitch_obs + j] = row[(j + r) % R];
      for (int j = 0; j < w.pitch_board; ++j) rot_board[(int64_t)r * w.pitch_board + j] = brow[(j + r) % HW];
    }
  }
  free(row);
  if (s->n_pieces > 0) {
    uint32_t* pieces_obs = reinterpret_cast<uint32_t*>(blob + w.pieces_off);
    uint32_t* pieces_board = pieces_obs + CAMPX_WIDE_MAX_PIECES;
    for (int p = 0; p < s->n_pieces; ++p) {
      const uint32_t cell = s->piece_cell[p];
      const uint32_t sets = (uint32_t)s->piece_layer[p] * (uint32_t)HW + cell;          // < 16 * 1024
      const uint32_t clears = (uint32_t)s->static_top_layer[cell] * (uint32_t)HW + cell;
      pieces_obs[p] = sets | (clears << 16);
      pieces_board[p] = cell | ((uint32_t)s->layer_char[s->piece_layer[p]] << 16);
    }
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipError_t e = hipMemcpyAsync(tables_dev, blob, (size_t)w.total, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  free(blob);
  return e == hipSuccess ? CAMPX_OK : hip_failed(e);
}

int32_t campx_wide_reset_launch(const CampxWideSpec* s, const void* tables_dev, CampxState st,
                                CampxOutputs out, int64_t B, void* stream) {
  int32_t rc = wide_check(s, tables_dev, st, out, B, 0);
  if (rc != CAMPX_OK) return rc;
  if (out.obs_format != CAMPX_OBS_INT8) return CAMPX_EINVAL;
  hipStream_t hs = static_cast<hipStream_t>(stream);
  const WideLayout w = wide_layout(*s);
  const u32x4* cells = reinterpret_cast<const u32x4*>(static_cast<const char*>(tables_dev) + w.cells_off);
  uint16_t* trace = reinterpret_cast<uint16_t*>(out.trace);
  const int64_t P = row_pitch(out, B);
  hipLaunchKernelGGL(wide_reset_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, hs, cells,
                     w.n_planes, reinterpret_cast<int32_t*>(st.pos), st, trace, P, B);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_failed(e);
  CampxOutputs one = out;     // one frame, written to slot 0 of each buffer
  one.obs_t_stride = 0;
  one.board_t_stride = 0;
  return wide_renders(*s, tables_dev, trace, one, B, 1, P, hs);
}

int32_t campx_wide_rollout_launch(const CampxWideSpec* s, const void* tables_dev, CampxState st,
                                  const int8_t* actions, CampxOutputs out, int64_t B, int32_t T,
                                  int32_t reset_first, void* stream) {
  int32_t rc = wide_check(s, tables_dev, st, out, B, T);
  if (rc != CAMPX_OK) return rc;
  if (T == 0) return CAMPX_OK;
  if (!actions) return CAMPX_EINVAL;
  hipStream_t hs = static_cast<hipStream_t>(stream);
  const WideLayout w = wide_layout(*s);
  {
    // Engine.play(): one launch when the rows are whole 16-byte chunks (see wide_step_kernel)
    const int HW = s->rows * s->cols, R = HW * s->n_layers;
    if (T == 1 && !reset_first && (R & 15) == 0 && s->n_pieces == 0 && knob(K_WIDE_STEP) &&
        (!out.board || (HW & 15) == 0) &&
        (int64_t)kStepEnvMax * R < (1ll << 24)) {
      WideStepParams sp;
      memset(&sp, 0, sizeof(sp));
      sp.n_states = s->n_states;
      sp.n_dyn = s->n_dyn;
      sp.cells = HW;
      sp.R = R;
      // environments per wave: a span of about 2 KiB, whole KiB when the row allows
      int n_env = 1;
      if (R < 2048) {
        n_env = 2048 / R;
        for (int c = 1; c <= kStepEnvMax; ++c)
          if (((int64_t)c * R) % 1024 == 0) {
            n_env = c;
            break;
          }
        n_env = n_env > kStepEnvMax ? kStepEnvMax : (n_env < 1 ? 1 : n_env);
      }
      sp.n_env = n_env;
      sp.inv_r = (uint32_t)(((1ull << 32) + R - 1) / R);
      sp.n_variants = w.n_variants;
      sp.n_planes = w.n_planes;
      sp.rot_obs_stride = 16ll * w.pitch_obs;
      sp.rot_board_stride = 16ll * w.pitch_board;
      for (int d = 0; d < s->n_dyn; ++d) {
        sp.dyn_off[d] = s->dyn_layer[d] * HW;
        sp.dyn_char[d] = s->layer_char[s->dyn_layer[d]];
      }
      sp.discounts[0] = 1.0f;
      for (int i = 1; i < 16; ++i) sp.discounts[i] = s->discount_list[i];
      const char* blob = static_cast<const char*>(tables_dev);
      const uint2* entries = reinterpret_cast<const uint2*>(blob);
      const u32x4* cells = reinterpret_cast<const u32x4*>(blob + w.cells_off);
      const int8_t* perf = reinterpret_cast<const int8_t*>(blob + w.perf_off);
      const int8_t* rot_obs = reinterpret_cast<const int8_t*>(blob + w.rot_obs_off);
      const int8_t* rot_board = reinterpret_cast<const int8_t*>(blob + w.rot_board_off);
      const int64_t n_waves = (B + n_env - 1) / n_env;
      const dim3 grid((unsigned)((n_waves + kWideThreads / kWave - 1) / (kWideThreads / kWave)));
      int32_t* state = reinterpret_cast<int32_t*>(st.pos);
#define CAMPX_WIDE_STEP2(BOARD, PERF, FMT)                                                       \
  hipLaunchKernelGGL((wide_step_kernel<BOARD, PERF, FMT>), grid, dim3(kWideThreads), 0, hs, sp,  \
                     entries, cells, perf, rot_obs, rot_board, state, st, actions, out, B)
#define CAMPX_WIDE_STEP(BOARD, PERF)                                      \
  do {                                                                    \
    if (out.obs_format == CAMPX_OBS_F16) CAMPX_WIDE_STEP2(BOARD, PERF, 1);      \
    else if (out.obs_format == CAMPX_OBS_BF16) CAMPX_WIDE_STEP2(BOARD, PERF, 2); \
    else CAMPX_WIDE_STEP2(BOARD, PERF, 0);                                \
  } while (0)
      if (out.board && out.perf) CAMPX_WIDE_STEP(true, true);
      else if (out.board) CAMPX_WIDE_STEP(true, false);
      else if (out.perf) CAMPX_WIDE_STEP(false, true);
      else CAMPX_WIDE_STEP(false, false);
#undef CAMPX_WIDE_STEP
#undef CAMPX_WIDE_STEP2
      const hipError_t e = hipGetLastError();
      return e == hipSuccess ? CAMPX_OK : hip_failed(e);
    }
    // ... and the rest - rows that are not whole chunks, a scenery of pieces - when a wave's span can
    // start on a 16-byte boundary and fits its LDS window (wide_step_lds_kernel), for frames up to
    // 24 MB: us per play(), one kernel / the pair (tools/bench_wide_play.py, round 6): the coin
    // field (300-byte rows, three pieces) B = 1 000 6.6 / 12.9, 4 096 6.7 / 13.0, 16 384 7.3 / 13.1,
    // 65 536 (19.7 MB) 12.1 / 13.3, 131 072 (39 MB) 20.4 / 15.8, 262 144 36.3 / 22.7; seven coins on
    // 4x9 B = 65 536 7.5 / 13.0, 262 144 (47 MB) 18.4 / 20.4
    if (T == 1 && !reset_first && knob(K_WIDE_STEP) && B * R <= (24ll << 20) &&
        (!out.board || (reinterpret_cast<uintptr_t>(out.board) & 15) == 0)) {
      // environments per aligned span: of the layered rows and - when the flat board is asked for - of
      // the board's (rows * cols divides R)
      const int unit = out.board ? HW : R;
      int step = 1;
      while (step < 16 && (unit * step) % 16 != 0) step <<= 1;
      int n_env = (kStepLdsSpan / R) / step * step;                // as many as the window holds
      n_env = n_env < step ? step : (n_env > kStepEnvMax ? kStepEnvMax : n_env);
      if ((int64_t)n_env * R <= kStepLdsSpan) {
        WideStepParams sp;
        memset(&sp, 0, sizeof(sp));
        sp.n_states = s->n_states;
        sp.n_dyn = s->n_dyn;
        sp.cells = HW;
        sp.R = R;
        sp.n_env = n_env;
        sp.inv_r = (uint32_t)(((1ull << 32) + R - 1) / R);
        sp.n_variants = w.n_variants;
        sp.n_planes = w.n_planes;
        sp.rot_obs_stride = 16ll * w.pitch_obs;
        sp.rot_board_stride = 16ll * w.pitch_board;
        for (int d = 0; d < s->n_dyn; ++d) {
          sp.dyn_off[d] = s->dyn_layer[d] * HW;
          sp.dyn_char[d] = s->layer_char[s->dyn_layer[d]];
        }
        sp.discounts[0] = 1.0f;
        for (int i = 1; i < 16; ++i) sp.discounts[i] = s->discount_list[i];
        WideStepLdsParams lp;
        memset(&lp, 0, sizeof(lp));
        lp.n_pieces = s->n_pieces;
        lp.n_planes = w.n_planes;
        lp.pitch_obs = w.pitch_obs;
        lp.pitch_board = w.pitch_board;
        for (int p = 0; p < s->n_pieces; ++p) {
          const uint32_t cell = s->piece_cell[p];
          lp.piece_obs[p] = ((uint32_t)s->piece_layer[p] * (uint32_t)HW + cell) |
                            (((uint32_t)s->static_top_layer[cell] * (uint32_t)HW + cell) << 16);
          lp.piece_board[p] = cell | ((uint32_t)s->layer_char[s->piece_layer[p]] << 16);
        }
        const char* blob = static_cast<const char*>(tables_dev);
        const uint2* entries = reinterpret_cast<const uint2*>(blob);
        const u32x4* cells = reinterpret_cast<const u32x4*>(blob + w.cells_off);
        const int8_t* perf = reinterpret_cast<const int8_t*>(blob + w.perf_off);
        const int8_t* rot_obs = reinterpret_cast<const int8_t*>(blob + w.rot_obs_off);
        const int8_t* rot_board = reinterpret_cast<const int8_t*>(blob + w.rot_board_off);
        const int64_t n_waves = (B + n_env - 1) / n_env;
        const dim3 grid((unsigned)((n_waves + kWideThreads / kWave - 1) / (kWideThreads / kWave)));
        int32_t* state = reinterpret_cast<int32_t*>(st.pos);
#define CAMPX_WIDE_STEP2(BOARD, PERF, FMT)                                                           \
  hipLaunchKernelGGL((wide_step_lds_kernel<BOARD, PERF, FMT>), grid, dim3(kWideThreads), 0, hs, sp,  \
                     lp, entries, cells, perf, rot_obs, rot_board, state, st, actions, out, B)
#define CAMPX_WIDE_STEP(BOARD, PERF)                                      \
  do {                                                                    \
    if (out.obs_format == CAMPX_OBS_F16) CAMPX_WIDE_STEP2(BOARD, PERF, 1);      \
    else if (out.obs_format == CAMPX_OBS_BF16) CAMPX_WIDE_STEP2(BOARD, PERF, 2); \
    else CAMPX_WIDE_STEP2(BOARD, PERF, 0);                                \
  } while (0)
        if (out.board && out.perf) CAMPX_WIDE_STEP(true, true);
        else if (out.board) CAMPX_WIDE_STEP(true, false);
        else if (out.perf) CAMPX_WIDE_STEP(false, true);
        else CAMPX_WIDE_STEP(false, false);
#undef CAMPX_WIDE_STEP
#undef CAMPX_WIDE_STEP2
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? CAMPX_OK : hip_failed(e);
      }
    }
  }
  WideParams wp;
  memset(&wp, 0, sizeof(wp));
  wp.n_states = s->n_states;
  wp.n_dyn = w.n_planes;           // (planes the update pass writes: the variant's / the pieces' is one of them)
  wp.discounts[0] = 1.0f;
  for (int i = 1; i < 16; ++i) wp.discounts[i] = s->discount_list[i];
  wp.has_dcodes = s->any_dcode;
  const int64_t pitch = row_pitch(out, B);
  wp.plane = (int64_t)T * pitch;
  const char* blob = static_cast<const char*>(tables_dev);
  const uint2* entries = reinterpret_cast<const uint2*>(blob);
  const u32x4* cells = reinterpret_cast<const u32x4*>(blob + w.cells_off);
  const int8_t* perf = reinterpret_cast<const int8_t*>(blob + w.perf_off);
  int32_t* state = reinterpret_cast<int32_t*>(st.pos);
  const size_t want = (size_t)(w.cells_off) + (size_t)s->n_states * sizeof(u32x4) +
                      (out.perf ? (size_t)w.n_entries : 0);
  // (setting wide_lds_max = bytes: tests run small games through the global-memory path that
  // games with thousands of states take)
  const size_t lds_max = (size_t)knob(K_WIDE_LDS_MAX);
  const bool in_lds = want <= lds_max;
  const size_t lds = in_lds ? want : 0;
  const dim3 grid((unsigned)((B + kWideThreads - 1) / kWideThreads));
  // The render kernel runs at the write ceiling only while the trace it reads stays cached
  // (launch_split() in campx_api.hip has the one-cell tier's numbers).  Here an entry is two
  // bytes and a game has up to nine planes.  The coin field of examples/coins_batched.py, B = 262 144,
  // T = 100, TB/s of observations in one piece / in chunks whose trace - all planes - is at most
  // 16 / 32 / 64 MB (tools/bench_pieces.py, profiles/r06_variants.txt): the walker's plane and the
  // pieces' mask, 105 MB: 6.20 / 6.49 / 6.63 / 6.71; the same with a scenery in seven variants:
  // 5.01 / 5.57 / 5.65 / 5.69; four planes of things, 210 MB: 4.09 / 6.36 / 6.37 / 6.53; one plane of
  // 52 MB in one piece loses nothing (6.64 against the one-cell tier's 6.75).  64 MB chunks are the
  // best there and the worst elsewhere (fraction of peak with chunks of 64 / 32 / 16 MB,
  // profiles/r06_sweep.txt: the coin field at B = 524 288 0.674 / 0.862 / 0.859, at B = 65 536 with
  // T = 1 000 0.797 / 0.847 / 0.821, maze 16x16 at B = 524 288 0.774 / 0.884 / 0.876; at B = 262 144
  // 0.871 / 0.860 / 0.842).  So: a launch whose trace is more than 4 x trace_chunk_mb (64 MB) runs
  // as chunks of frames of at most 2 x trace_chunk_mb (32 MB), update pass and render alternating.
  const int64_t per_frame = pitch * (int64_t)sizeof(uint16_t) * w.n_planes;
  int64_t chunk = (2 * (knob(K_TRACE_CHUNK_MB) << 20)) / per_frame;
  chunk = chunk < 16 ? 16 : chunk & ~(int64_t)15;
  const bool whole = per_frame * T <= 4 * (knob(K_TRACE_CHUNK_MB) << 20) || T <= chunk || wide_last_only(out);
  const int64_t elem = out.obs_format == CAMPX_OBS_INT8 ? 1 : 2;
  const uint16_t* trace0 = reinterpret_cast<const uint16_t*>(out.trace);
  for (int64_t t0 = 0; t0 < T; t0 += whole ? T : chunk) {
    const int32_t n = whole ? T : (int32_t)(T - t0 < chunk ? T - t0 : chunk);
    CampxOutputs part = out;
    if (!whole) {
      part.obs = out.obs + t0 * out.obs_t_stride * elem;
      if (out.board) part.board = out.board + t0 * out.board_t_stride;
      if (out.reward) part.reward = out.reward + t0 * pitch;
      if (out.discount) part.discount = out.discount + t0 * pitch;
      if (out.done) part.done = out.done + t0 * pitch;
      if (out.perf) part.perf = out.perf + t0 * pitch;
      part.trace = reinterpret_cast<uint8_t*>(const_cast<uint16_t*>(trace0 + t0 * pitch));
    }
    const int8_t* acts = actions + t0 * B;
    const int32_t first = t0 == 0 ? reset_first : 0;
#define CAMPX_WIDE_LAUNCH(LDS, PERF)                                                              \
  do {                                                                                            \
    CAMPX_ALLOW_LDS((wide_update_kernel<LDS, PERF>), lds);                                        \
    hipLaunchKernelGGL((wide_update_kernel<LDS, PERF>), grid, dim3(kWideThreads), lds, hs, wp,    \
                       entries, cells, perf, state, st, acts, part, B, n, first);                 \
  } while (0)
    if (in_lds && out.perf) CAMPX_WIDE_LAUNCH(true, true);
    else if (in_lds) CAMPX_WIDE_LAUNCH(true, false);
    else if (out.perf) CAMPX_WIDE_LAUNCH(false, true);
    else CAMPX_WIDE_LAUNCH(false, false);
#undef CAMPX_WIDE_LAUNCH
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_failed(e);
    const int32_t rc = wide_renders(*s, tables_dev, trace0 + t0 * pitch, part, B, n, wp.plane, hs);
    if (rc != CAMPX_OK) return rc;
  }
  return CAMPX_OK;
}

int32_t campx_wide_rules_size(void) { return (int32_t)sizeof(CampxWideRules); }

int32_t campx_wide_enumerate_launch(const CampxWideRules* r, const uint16_t* cells_in, int64_t N,
                                    uint16_t* next_cells, float* reward, uint8_t* done,
                                    uint8_t* shows, int8_t* perf, void* stream) {
  if (!r || !cells_in || !next_cells || !reward || !done || !shows || N < 0) return CAMPX_EINVAL;
  if (r->magic != CAMPX_SPEC_MAGIC || r->version != CAMPX_SPEC_VERSION) return CAMPX_ESPEC;
  if (r->rows < 1 || r->cols < 1 || r->rows > 127 || r->cols > 127 ||
      r->rows * r->cols > CAMPX_WIDE_MAX_CELLS)
    return CAMPX_ESPEC;
  if (r->n_dyn < 1 || r->n_dyn > CAMPX_MAX_DYN || r->n_rules < 0 || r->n_rules > CAMPX_MAX_RULES)
    return CAMPX_ESPEC;
  if (!r->top_layer || !r->top_z || !r->cover || !r->cell_class) return CAMPX_EINVAL;
  for (int i = 0; i < r->n_rules; ++i) {
    const CampxRule& R = r->rules[i];
    if (R.dyn < 0 || R.dyn >= r->n_dyn) return CAMPX_ESPEC;
    if (R.op == CAMPX_OP_BOX && (R.aux < 0 || R.aux >= r->n_dyn)) return CAMPX_ESPEC;
  }
  if (N == 0) return CAMPX_OK;
  if (N > (int64_t)0x7fffffff * 256 / CAMPX_N_ACTIONS) return CAMPX_EINVAL;
  return launch_wide_enumerate(*r, cells_in, N, next_cells, reward, done, shows, perf,
                               static_cast<hipStream_t>(stream));
}

}  // extern "C"
