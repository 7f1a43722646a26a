// k_wide.hip - the wide tier: one-mover games on boards above 128 cells (up to 1 024).
//
// The update pass is a walk through the game's (cell, action) table, which the HOST filled
// (campx_amd/tabulate.py); the observation stream is k_render.hip's render kernel reading a
// 16-bit trace.  A frame row is L*rows*cols bytes - 1 280 for a 16x16 board with five
// characters - against 2 bytes of trace, 4 of reward and 1 of action, so the update kernel
// here is the plain one (a lane per environment, the table in LDS, the frame's scalars
// stored as they come): it moves 1-2 % of the launch's bytes and takes a few per cent of
// its time.  What bounds the tier is the render kernel's write stream.

#include "campx_common.hip.h"

namespace campx_impl {

constexpr int kWideThreads = 256;
constexpr int kWideAhead = 8;     // frames whose actions are fetched before their chain runs

struct WideParams {
  int32_t cols, cells, init_cell;
  float discounts[16];
};

// LDS / table-blob entry: x = reward; y = [0:9] the cell the NEXT frame starts from (the
// art's cell when this frame ended the episode: the rebuild is folded into the chain),
// [10:19] the cell after this frame, [20] whether the mover shows there, [21] done,
// [22:25] discount code, [26:29] the scenery layer it covers there.
__host__ __device__ __forceinline__ uint32_t wide_pack(uint32_t from, uint32_t next, uint32_t vis,
                                                       uint32_t done, uint32_t dcode, uint32_t cover) {
  return from | (next << 10) | (vis << 20) | (done << 21) | (dcode << 22) | (cover << 26);
}

template <bool kPerf>
__global__ __launch_bounds__(kWideThreads) void wide_update_kernel(
    WideParams wp, const uint2* __restrict__ entries, const int8_t* __restrict__ perf_tab,
    CampxState st, const int8_t* __restrict__ actions, CampxOutputs out, int64_t B, int32_t T,
    int32_t reset_first) {
  extern __shared__ __attribute__((aligned(16))) uint2 table[];   // cells * 5 (+ perf bytes)
  __shared__ float discounts[16];
  const int n_entries = wp.cells * CAMPX_N_ACTIONS;
  int8_t* perf_lds = reinterpret_cast<int8_t*>(table + n_entries);
  for (int i = threadIdx.x; i < n_entries; i += kWideThreads) table[i] = entries[i];
  if (kPerf)
    for (int i = threadIdx.x; i < n_entries; i += kWideThreads) perf_lds[i] = perf_tab[i];
  if (threadIdx.x < 16) discounts[threadIdx.x] = wp.discounts[threadIdx.x];
  __syncthreads();

  const int64_t env = (int64_t)blockIdx.x * kWideThreads + threadIdx.x;
  if (env >= B) return;
  const int W = wp.cols;
  int cell = wp.init_cell, over = 0;
  float ret = 0.0f;
  if (!reset_first) {
    cell = (int)st.pos[env] * W + (int)st.pos[B + env];
    over = st.done[env];
    if (st.ret) ret = st.ret[env];
  }
  uint32_t from = (uint32_t)(over ? wp.init_cell : cell);
  uint16_t* trace = reinterpret_cast<uint16_t*>(out.trace);
  const int64_t P = row_pitch(out, B);
  int bad = 0;
  for (int t0 = 0; t0 < T; t0 += kWideAhead) {
    uint32_t a[kWideAhead];
#pragma unroll
    for (int j = 0; j < kWideAhead; ++j) {
      const int t = t0 + j < T ? t0 + j : T - 1;   // (clamped: a load that is ignored)
      a[j] = (uint8_t)actions[(int64_t)t * B + env];
    }
#pragma unroll
    for (int j = 0; j < kWideAhead; ++j) {
      if (t0 + j < T) {
        bad += a[j] > 4u;
        const uint32_t idx = from * CAMPX_N_ACTIONS + (a[j] > 4u ? 4u : a[j]);
        const uint2 e = table[idx];
        from = e.y & 0x3ffu;                       // the chain: cell -> entry -> cell
        const uint32_t done = (e.y >> 21) & 1u, dcode = (e.y >> 22) & 15u;
        const int64_t at = (int64_t)(t0 + j) * P + env;
        trace[at] = (uint16_t)(((e.y >> 10) & 0x3ffu) | (((e.y >> 26) & 15u) << 10) |
                               (((e.y >> 20) & 1u) << 15));
        if (out.reward) out.reward[at] = __uint_as_float(e.x);
        if (out.discount)
          out.discount[at] = __uint_as_float(discount_bits(discounts, dcode, done));
        if (out.done) out.done[at] = (uint8_t)done;
        if (kPerf && out.perf) out.perf[at] = perf_lds[idx];
        ret = (over ? 0.0f : ret) + real_reward(__uint_as_float(e.x));
        over = (int)done;
        cell = (int)((e.y >> 10) & 0x3ffu);
      }
    }
  }
  st.pos[env] = (int8_t)(cell / W);
  st.pos[B + env] = (int8_t)(cell % W);
  st.done[env] = (uint8_t)over;
  if (st.ret) st.ret[env] = ret;
  report_bad_actions(out, bad);
}

// its_showtime(): state from the art and the trace row of the first observation.
__global__ void wide_reset_kernel(int32_t row0, int32_t col0, uint32_t entry, CampxState st,
                                  uint16_t* __restrict__ trace, int64_t B) {
  const int64_t env = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= B) return;
  st.pos[env] = (int8_t)row0;
  st.pos[B + env] = (int8_t)col0;
  st.done[env] = 0;
  if (st.ret) st.ret[env] = 0.0f;
  trace[env] = (uint16_t)entry;
}

// ---- the table blob: [entries uint2 x n][perf int8 x n, padded to 16][rot_obs][rot_board]
struct WideLayout {
  int64_t n_entries, perf_off, rot_obs_off, rot_board_off, total;
  int pitch_obs, pitch_board;
};

WideLayout wide_layout(const CampxWideSpec& s) {
  WideLayout w;
  const int64_t HW = (int64_t)s.rows * s.cols, R = HW * s.n_layers;
  w.n_entries = HW * CAMPX_N_ACTIONS;
  w.perf_off = w.n_entries * (int64_t)sizeof(uint2);
  w.rot_obs_off = (w.perf_off + w.n_entries + 15) & ~(int64_t)15;
  w.pitch_obs = (int)(((R + 15) & ~(int64_t)15) + 16);
  w.rot_board_off = w.rot_obs_off + 16ll * w.pitch_obs;
  w.pitch_board = (int)(((HW + 15) & ~(int64_t)15) + 16);
  w.total = w.rot_board_off + 16ll * w.pitch_board;
  return w;
}

RenderSource wide_render_source(const CampxWideSpec& s, const void* tables_dev) {
  const WideLayout w = wide_layout(s);
  RenderSource src;
  memset(&src, 0, sizeof(src));
  src.rows = s.rows;
  src.cols = s.cols;
  src.n_layers = s.n_layers;
  src.n_dyn = 1;
  src.dyn_layer[0] = s.dyn_layer;
  memcpy(src.layer_char, s.layer_char, sizeof(src.layer_char));
  const char* blob = static_cast<const char*>(tables_dev);
  src.rot_obs = reinterpret_cast<const int8_t*>(blob + w.rot_obs_off);
  src.rot_board = reinterpret_cast<const int8_t*>(blob + w.rot_board_off);
  src.top_layer = nullptr;
  src.wide = true;
  return src;
}

// Frames back to back, or - strides 0 - only the last one.
bool wide_last_only(const CampxOutputs& out) {
  return out.obs_t_stride == 0 && (!out.board || out.board_t_stride == 0) &&
         out.obs_format == CAMPX_OBS_INT8;
}

int32_t wide_check(const CampxWideSpec* s, const void* tables, const CampxState& st,
                   const CampxOutputs& out, int64_t B, int32_t T) {
  if (!s || !tables || !st.pos || !st.done || !out.obs || !out.trace || B <= 0 || T < 0)
    return CAMPX_EINVAL;
  if ((reinterpret_cast<uintptr_t>(out.obs) & 15) || (reinterpret_cast<uintptr_t>(out.trace) & 1))
    return CAMPX_EINVAL;
  if (out.scalar_pitch && out.scalar_pitch < B) return CAMPX_EINVAL;
  if (out.obs_format < CAMPX_OBS_INT8 || out.obs_format > CAMPX_OBS_BF16) return CAMPX_EINVAL;
  const int32_t v = campx_wide_spec_validate(s);
  if (v != CAMPX_OK) return v;
  if (out.perf && !s->has_perf) return CAMPX_EINVAL;
  const int64_t HW = (int64_t)s->rows * s->cols, LHW = HW * s->n_layers;
  if (B * LHW >= (1ll << 32) - 65536) return CAMPX_EINVAL;
  const bool every = out.obs_t_stride == B * LHW && (!out.board || out.board_t_stride == B * HW);
  if (T > 0 && !every && !wide_last_only(out)) return CAMPX_EINVAL;
  return CAMPX_OK;
}

int32_t wide_renders(const CampxWideSpec& s, const void* tables_dev, const uint16_t* trace,
                     CampxOutputs out, int64_t B, int32_t T, hipStream_t stream) {
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
                                    B, n, 0, pitch, false, out.obs_format, stream);
    if (rc != CAMPX_OK) return rc;
    if (out.board) {
      rc = launch_render_from(src, trace + t0 * pitch, out.board + t0 * out.board_t_stride, B, n, 0,
                              pitch, true, 0, stream);
      if (rc != CAMPX_OK) return rc;
    }
  }
  return CAMPX_OK;
}

}  // namespace campx_impl

using namespace campx_impl;

extern "C" {

int32_t campx_wide_spec_size(void) { return (int32_t)sizeof(CampxWideSpec); }

int32_t campx_wide_spec_validate(const CampxWideSpec* s) {
  if (!s) return CAMPX_EINVAL;
  if (s->magic != CAMPX_SPEC_MAGIC || s->version != CAMPX_SPEC_VERSION) return CAMPX_ESPEC;
  if (s->rows < 1 || s->cols < 1 || s->rows > 127 || s->cols > 127) return CAMPX_ESPEC;
  const int HW = s->rows * s->cols;
  if (HW < 16 || HW > CAMPX_WIDE_MAX_CELLS) return CAMPX_ESPEC;
  if (s->n_layers < 1 || s->n_layers > CAMPX_MAX_LAYERS) return CAMPX_ESPEC;
  if (s->dyn_layer < 0 || s->dyn_layer >= s->n_layers) return CAMPX_ESPEC;
  if (s->init_cell < 0 || s->init_cell >= HW) return CAMPX_ESPEC;
  if ((s->init_hidden | s->has_perf | s->any_reward) & ~1) return CAMPX_ESPEC;
  for (int i = 0; i < HW; ++i)
    if (s->static_top_layer[i] >= s->n_layers) return CAMPX_ESPEC;
  for (int i = 0; i < HW * CAMPX_N_ACTIONS; ++i) {
    const CampxWideTransition& tr = s->table[i];
    if ((tr.next_cell & 0x3ffu) >= (uint32_t)HW || (tr.next_cell & 0x7c00u) || (tr.done & 0x0eu))
      return CAMPX_ESPEC;
  }
  return CAMPX_OK;
}

int64_t campx_wide_tables_bytes(const CampxWideSpec* s) {
  if (campx_wide_spec_validate(s) != CAMPX_OK) return 0;
  return wide_layout(*s).total;
}

int32_t campx_wide_tables_build(const CampxWideSpec* s, void* tables_dev, void* stream) {
  if (!s || !tables_dev) return CAMPX_EINVAL;
  const int32_t v = campx_wide_spec_validate(s);
  if (v != CAMPX_OK) return v;
  const WideLayout w = wide_layout(*s);
  const int HW = s->rows * s->cols, R = HW * s->n_layers;
  char* blob = static_cast<char*>(calloc(1, (size_t)w.total));
  if (!blob) return CAMPX_ENOMEM;
  uint2* entries = reinterpret_cast<uint2*>(blob);
  int8_t* perf = reinterpret_cast<int8_t*>(blob + w.perf_off);
  for (int i = 0; i < HW * CAMPX_N_ACTIONS; ++i) {
    const CampxWideTransition& tr = s->table[i];
    const uint32_t next = tr.next_cell & 0x3ffu, hidden = tr.next_cell >> 15;
    const uint32_t done = tr.done & 1u, dcode = tr.done >> 4;
    uint32_t bits;
    memcpy(&bits, &tr.reward, 4);
    entries[i].x = bits;
    entries[i].y = wide_pack(done ? (uint32_t)s->init_cell : next, next, hidden ? 0u : 1u, done, dcode,
                             s->static_top_layer[next]);
    perf[i] = tr.perf;
  }
  // the scenery's row (layers by equality, campx/rendering.py:204-215) and its rotations
  int8_t* row = static_cast<int8_t*>(calloc(1, (size_t)R + HW));
  if (!row) {
    free(blob);
    return CAMPX_ENOMEM;
  }
  int8_t* brow = row + R;
  for (int i = 0; i < HW; ++i) {
    row[(int)s->static_top_layer[i] * HW + i] = 1;
    brow[i] = (int8_t)s->layer_char[s->static_top_layer[i]];
  }
  int8_t* rot_obs = reinterpret_cast<int8_t*>(blob + w.rot_obs_off);
  int8_t* rot_board = reinterpret_cast<int8_t*>(blob + w.rot_board_off);
  for (int r = 0; r < 16; ++r) {
    for (int j = 0; j < w.pitch_obs; ++j) rot_obs[(int64_t)r * w.pitch_obs + j] = row[(j + r) % R];
    for (int j = 0; j < w.pitch_board; ++j) rot_board[(int64_t)r * w.pitch_board + j] = brow[(j + r) % HW];
  }
  free(row);
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
  const uint32_t entry = (uint32_t)s->init_cell | ((uint32_t)s->static_top_layer[s->init_cell] << 10) |
                         (s->init_hidden ? 0u : 0x8000u);
  uint16_t* trace = reinterpret_cast<uint16_t*>(out.trace);
  hipLaunchKernelGGL(wide_reset_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, hs,
                     s->init_cell / s->cols, s->init_cell % s->cols, entry, st, trace, B);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_failed(e);
  CampxOutputs one = out;     // one frame, written to slot 0 of each buffer
  one.obs_t_stride = 0;
  one.board_t_stride = 0;
  return wide_renders(*s, tables_dev, trace, one, B, 1, hs);
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
  WideParams wp;
  memset(&wp, 0, sizeof(wp));
  wp.cols = s->cols;
  wp.cells = s->rows * s->cols;
  wp.init_cell = s->init_cell;
  wp.discounts[0] = 1.0f;
  for (int i = 1; i < 16; ++i) wp.discounts[i] = s->discount_list[i];
  const char* blob = static_cast<const char*>(tables_dev);
  const uint2* entries = reinterpret_cast<const uint2*>(blob);
  const int8_t* perf = reinterpret_cast<const int8_t*>(blob + w.perf_off);
  const size_t lds = (size_t)w.n_entries * sizeof(uint2) + (out.perf ? (size_t)w.n_entries : 0);
  const dim3 grid((unsigned)((B + kWideThreads - 1) / kWideThreads));
  if (out.perf) {
    CAMPX_ALLOW_LDS(wide_update_kernel<true>, lds);
    hipLaunchKernelGGL(wide_update_kernel<true>, grid, dim3(kWideThreads), lds, hs, wp, entries, perf,
                       st, actions, out, B, T, reset_first);
  } else {
    CAMPX_ALLOW_LDS(wide_update_kernel<false>, lds);
    hipLaunchKernelGGL(wide_update_kernel<false>, grid, dim3(kWideThreads), lds, hs, wp, entries, perf,
                       st, actions, out, B, T, reset_first);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_failed(e);
  return wide_renders(*s, tables_dev, reinterpret_cast<const uint16_t*>(out.trace), out, B, T, hs);
}

}  // extern "C"
