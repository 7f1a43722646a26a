// campx_api.hip - batched CampX grid-world engine for MI355X (gfx950, CDNA4): the C ABI
// declared in include/campx_hip.h, the dispatch from an entry point to the kernels'
// launchers (k_*.hip, one translation unit per kernel family, declared in
// campx_common.hip.h), and the set-up-time table builders.
//
// What one launch computes, per environment and per frame, is the reference's
// Engine.play() (campx/engine.py:114-166): every entity's update() in schedule
// order with one repaint per update group (engine.py:195-208), the Plot's reward /
// discount / game-over bookkeeping (campx/plot.py:161-211, engine.py:285-292) and
// the occluded layered-board render (campx/rendering.py:104-219).
//
// Kernels (NOTES.md section 3 has the numbers):
//   rollout_kernel        rule interpreter + render, fused.  One lane = one
//                         environment, one wave = one workgroup = 64 environments;
//                         rules arrive in the kernarg segment (scalar loads/branches),
//                         scenery tables and the wave's 64 x L*H*W-byte output image
//                         live in LDS; a frame patches a few bytes of the image and
//                         streams it out, 1 KiB per wave-instruction.
//   rollout_table_kernel  same, for games with one moving thing: the update pass is a
//                         lookup in a (cell, action) table that campx_spec_compile()
//                         fills by running rollout_kernel over every pair.
//   update_*_kernel       the update pass alone (producer, consumer and loader waves)
//                         from the game's state table: (cell, action) in LDS for one
//                         mover, (cell, cell, action) in LDS for two, (cell, ..., action)
//                         in global memory for three and four,
//   render_kernel         and the observation stream alone: one-shot blocks, every
//                         wave one aligned KiB store - the store pattern that reaches
//                         the chip's HBM write ceiling.  The default for rollouts.
//   step_*_kernel         Engine.play(): one frame, one-shot, one wave per 64 environments.
//   shape_rollout_kernel  Hello-World-style games (rigidly translated multi-cell things):
//                         one wave per environment, scalar update pass.
// No MFMA anywhere: the path has no contraction; every kernel is bound by the HBM
// write stream of observations.
//
// Compiled with -ffp-contract=off: rewards are sums of a few float terms and must
// round exactly like the reference's float32 tensor arithmetic.

#include "campx_common.hip.h"

namespace campx_impl {

thread_local int32_t g_last_hip_error = 0;

// ----------------------------------------------------------------------- configuration
// Everything about the library's behaviour that is not an argument of a call, in ONE table: set
// through campx_config_set() (tests, an embedding application) or, for a whole process, through
// the environment variable CAMPX_CONFIG="name=value,name=value" - the only getenv of this
// library, read once, before the first value is asked for.  campx_config_string() writes the
// effective values.  Until round 5 there were some thirty-five environment variables, read once
// each wherever they were used; those whose A/B measurement is settled are gone, with the code
// paths they selected (which profile settled each: NOTES.md R6.2).
struct KnobRow {
  const char* name;
  int64_t value, lo, hi;
  const char* what;
};
static KnobRow g_knobs[K_COUNT] = {
    // largest trace one update + render pair of a rollout works on; a longer rollout runs in chunks
    // of whole frames (the render's trace reads then hit the memory-side cache: profiles/r02_sweep.txt).
    // 0: a chunk per frame (tests/test_chunked_rollouts.py runs the chunked form at test sizes)
    {"trace_chunk_mb", 16, 0, 1 << 20, "MiB of trace per plane and update + render pair of a chunked rollout (all planes: x 2)"},
    {"trace_whole_mb", 28, 0, 1 << 20, "largest trace (MiB) a rollout may have and still run as one pair"},
    // frame-major shape tier: environment-frames per chunk, in thousands (profiles/r05_shape_rocprofv3.txt)
    {"shape_chunk_kf", 2000, 1, 1 << 30, "thousand environment-frames per chunk of a frame-major shape rollout"},
    // 0: always the one-wave-per-environment shape kernel (the second implementation the parity tests hold
    // the frame-major kernels against)
    {"shape_split", 1, 0, 1, "frame-major shape kernels where the launch allows them"},
    // number of 8-wave "big" update workgroups from which launch_update prefers them; -1: one per
    // compute unit of the device (1: at every batch size - tests/test_update_workgroups.py)
    {"big_wgs", -1, -1, 1 << 30, "big update workgroups from this many up (-1: one per CU)"},
    // 0: a rollout is always two launches (update pass, render), never the tagged-trace single launch
    // (the parity twin of tests/test_flow.py)
    {"flow", 1, 0, 1, "one-launch rollouts (tagged trace) for table games up to 8 192 environments"},
    // how long a render wave of a one-launch rollout waits for its trace entries before it raises
    // CAMPX_ERR_FLOW_TIMEOUT, and a debugging delay (s_sleep units) in front of the update role's
    // stores: tests/test_flow.py provokes the timeout with (1, 3000)
    {"flow_max_naps", kFlowMaxNaps, 1, 1ll << 32, "naps before a one-launch rollout's render wave gives up"},
    {"flow_debug_delay", 0, 0, 1 << 24, "debug: delay in front of the update role's tagged stores"},
    // wide tier: largest state table (bytes) staged in LDS; 0 sends every game through L2 / HBM gathers
    {"wide_lds_max", (int64_t)kWideLdsMax, 0, (int64_t)kWideLdsMax, "largest state table (bytes) kept in LDS by wide_update_kernel"},
    // 0: Engine.play() of a state-table game is always the update + render pair, never wide_step_kernel
    // (the parity twin of tests/test_wide_parity.py)
    {"wide_step", 1, 0, 1, "one-kernel Engine.play() for state-table games"},
};

static void knobs_from_environment() {
  const char* v = getenv("CAMPX_CONFIG");
  if (!v) return;
  std::string text(v);
  size_t at = 0;
  while (at < text.size()) {
    size_t end = text.find(',', at);
    if (end == std::string::npos) end = text.size();
    const std::string item = text.substr(at, end - at);
    const size_t eq = item.find('=');
    if (eq != std::string::npos) {
      const std::string name = item.substr(0, eq);
      bool known = false;
      for (KnobRow& row : g_knobs)
        if (name == row.name) {
          const long long x = atoll(item.c_str() + eq + 1);
          known = true;
          if (x >= row.lo && x <= row.hi) row.value = x;
          else fprintf(stderr, "campx: CAMPX_CONFIG: %s=%lld is outside %lld..%lld, ignored\n", row.name, x,
                       (long long)row.lo, (long long)row.hi);
        }
      if (!known) fprintf(stderr, "campx: CAMPX_CONFIG: no setting is called '%s', ignored\n", name.c_str());
    }
    at = end + 1;
  }
}

static std::once_flag g_knobs_read;

int64_t knob(Knob k) {
  std::call_once(g_knobs_read, knobs_from_environment);
  return __atomic_load_n(&g_knobs[k].value, __ATOMIC_RELAXED);
}

int32_t hip_failed(hipError_t e) {
  g_last_hip_error = (int32_t)e;
  return CAMPX_ELAUNCH;
}

int32_t lds_refused(hipError_t e) {
  g_last_hip_error = (int32_t)e;
  return CAMPX_EINVAL;
}

// Can this call take the two-kernel path?  Frames must be stored back to back and be
// whole 16-byte chunks, and a chunk may span at most two rows.
// Strides of 0: every frame overwrites the first slot, so only the last survives - the
// two-kernel path then renders just that one from the last row of the trace.
bool last_frame_only(const CampxOutputs& out) {
  return out.obs_t_stride == 0 && (!out.board || out.board_t_stride == 0) &&
         out.obs_format == CAMPX_OBS_INT8;
}

bool split_ok(const CampxSpec& s, const CampxOutputs& out, int64_t B, int32_t T) {
  const int64_t HW = (int64_t)s.rows * s.cols, LHW = HW * s.n_layers;
  if (!out.trace || !s.render_valid || T <= 0) return false;
  if (LHW < 16 || B * LHW >= (1ll << 32) - 65536) return false;
  if (out.board && HW < 16) return false;
  // every frame kept, back to back - or only the last one (strides 0)
  const bool every = out.obs_t_stride == B * LHW && (!out.board || out.board_t_stride == B * HW);
  return every || last_frame_only(out);
}

int32_t launch_renders(const CampxSpec& s, const CampxSpec* spec_dev, CampxOutputs out, int64_t B,
                       int32_t T, int64_t plane_rows, hipStream_t stream) {
  const uint8_t* first = out.trace;
  const int64_t pitch = row_pitch(out, B);
  if (last_frame_only(out)) {
    first += (int64_t)(T - 1) * pitch;
    T = 1;
  }
  int32_t rc = launch_render(s, spec_dev, first, out.obs, B, T, plane_rows, pitch, false,
                             out.obs_format, stream);
  if (rc != CAMPX_OK) return rc;
  if (out.board)
    rc = launch_render(s, spec_dev, first, out.board, B, T, plane_rows, pitch, true, 0, stream);
  return rc;
}

int32_t launch_split(const CampxSpec& s, const CampxSpec* spec_dev, CampxState st,
                     const int8_t* actions, CampxOutputs out, int64_t B, int32_t T,
                     int32_t reset_first, bool use_table, hipStream_t stream) {
  const int64_t pitch = row_pitch(out, B);
  const int64_t plane = (int64_t)T * pitch;
  // The render kernel runs at the write ceiling only while the trace it reads stays cached
  // (boat race, B = 65 536: 6.96 TB/s with a 26 MB trace at T = 400, 5.35 TB/s with 65 MB at
  // T = 1 000; the same at B = 524 288, T = 100): run long launches as chunks of frames,
  // update pass and render alternating, each chunk's trace plane at most 16 MB (setting trace_chunk_mb).
  // (us per launch, render kernels only, no chunks / 28 / 16 / 8 MB: T = 1 000: 2 265 / 1 820 /
  // 1 641 / 1 644; B = 524 288: 1 739 / 1 504 / 1 314 / 1 316 - gpurun_out/t16.  A 26 MB trace
  // in one piece is still at full speed, so launches up to 28 MB (trace_whole_mb) are not cut.)
  // (per moving thing's plane of the trace: sokoban with three boxes, four planes of 13 MB,
  // renders at full speed in one piece, and 4 % slower cut in four.  But not four planes of 26 MB:
  // round 6 found the state-table tier's launches losing up to a third of their rate once ALL the
  // planes together pass 64 MB (k_wide.hip), and here too - B = 262 144, T = 100: sokoban with two /
  // three boxes in one piece 0.763 / 0.683 of peak, in 16 MB chunks 0.859 / 0.842 - so a launch also
  // goes in one piece only while its planes together stay within 4 x trace_chunk_mb.  And a chunk's
  // planes together within 2 x trace_chunk_mb: B = 524 288 with chunks of 16 / 8 MB per plane:
  // two boxes 0.689 / 0.822, three 0.629 / 0.802 - one box, two planes, 0.817 / 0.788;
  // profiles/r06_sweep.txt.)
  const int64_t per_frame = B;
  const int64_t planes = s.n_dyn > 2 ? s.n_dyn : 2;
  int64_t chunk = (2 * (knob(K_TRACE_CHUNK_MB) << 20)) / (per_frame * planes);
  chunk = chunk < 16 ? 16 : chunk & ~(int64_t)15;
  chunk = chunk > 65520 ? 65520 : chunk;   // a render launch has one grid row per frame
  const bool whole = (per_frame * T <= (knob(K_TRACE_WHOLE_MB) << 20) &&
                      per_frame * T * s.n_dyn <= 4 * (knob(K_TRACE_CHUNK_MB) << 20) && T <= 65535) || T <= chunk;
  // (two to four movers: their pair / tuple table is the caller's, CampxState.pair_table)
  const bool multi_table = s.n_dyn >= 2 && st.pair_table;
  if (whole && !last_frame_only(out) && flow_ok(s, out, B, T, use_table || multi_table, stream)) {
    // table games of small batches: one launch, the render role following the update role as
    // its entries arrive (k_update.hip, launch_flow)
    int32_t rc = launch_flow(s, spec_dev, st, actions, out, B, T, reset_first, stream);
    if (rc != CAMPX_OK || !out.board) return rc;
    return launch_render(s, spec_dev, out.trace, out.board, B, T, plane, pitch, true, 0, stream);
  }
  if (last_frame_only(out) || whole) {
    const int32_t rc = launch_update(s, spec_dev, st, actions, out, B, T, reset_first, use_table,
                                     plane, stream);
    if (rc != CAMPX_OK) return rc;
    return launch_renders(s, spec_dev, out, B, T, plane, stream);
  }
  const int64_t elem = out.obs_format == CAMPX_OBS_INT8 ? 1 : 2;
  for (int64_t t0 = 0; t0 < T; t0 += chunk) {
    const int32_t n = (int32_t)(T - t0 < chunk ? T - t0 : chunk);
    CampxOutputs part = out;
    part.obs = out.obs + t0 * out.obs_t_stride * elem;
    if (out.board) part.board = out.board + t0 * out.board_t_stride;
    if (out.reward) part.reward = out.reward + t0 * pitch;
    if (out.discount) part.discount = out.discount + t0 * pitch;
    if (out.done) part.done = out.done + t0 * pitch;
    if (out.perf) part.perf = out.perf + t0 * pitch;
    part.trace = out.trace + t0 * pitch;
    int32_t rc = launch_update(s, spec_dev, st, actions + t0 * B, part, B, n,
                               t0 == 0 ? reset_first : 0, use_table, plane, stream);
    if (rc != CAMPX_OK) return rc;
    rc = launch_renders(s, spec_dev, part, B, n, plane, stream);
    if (rc != CAMPX_OK) return rc;
  }
  return CAMPX_OK;
}

int32_t launch(const CampxSpec* spec_host, const CampxSpec* spec_dev, CampxState st,
               const int8_t* actions, CampxOutputs out, int64_t B, int32_t T, int32_t reset_first,
               int32_t emit_first, void* stream) {
  if (!spec_host || !spec_dev || !st.pos || !st.done || !out.obs || B <= 0 || T < 0)
    return CAMPX_EINVAL;
  if (T > 0 && !actions) return CAMPX_EINVAL;
  if (out.perf && spec_host->perf_dyn < 0) return CAMPX_EINVAL;
  if (reinterpret_cast<uintptr_t>(out.obs) & 15) return CAMPX_EINVAL;
  if (B > (int64_t)0x7fffffff * 16) return CAMPX_EINVAL;
  if (out.scalar_pitch && out.scalar_pitch < B) return CAMPX_EINVAL;
  const int32_t v = campx_spec_validate(spec_host);
  if (v != CAMPX_OK) return v;
  if (lds_bytes(*spec_host, out.board != nullptr, kWave) + 8 * 1024 > kLdsPerWorkgroup) return CAMPX_ESPEC;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // a host-tabulated game (table_only) has no rules: its tables are not optional
  const bool only = spec_host->table_only != 0;
  const bool use_table = spec_host->table_valid && spec_host->n_dyn == 1;
  if (only && T > 0 && spec_host->n_dyn >= 2 && !st.pair_table) return CAMPX_ESPEC;
  if (out.obs_format < CAMPX_OBS_INT8 || out.obs_format > CAMPX_OBS_BF16) return CAMPX_EINVAL;
  if (!emit_first && split_ok(*spec_host, out, B, T))
    return launch_split(*spec_host, spec_dev, st, actions, out, B, T, reset_first, use_table, s);
  // (16-bit observations: the render kernel above, or the one-frame kernels below)
  if (use_table && T == 1 && !emit_first && spec_host->render_valid)
    return launch_step_table(*spec_host, spec_dev, st, actions, out, B, reset_first, s);
  if (T == 1 && !emit_first && spec_host->n_dyn == 2 && st.pair_table && spec_host->render_valid)
    return launch_step_pair(*spec_host, spec_dev, st, actions, out, B, reset_first, s);
  if (T == 1 && !emit_first && spec_host->n_dyn >= 3 && st.pair_table && spec_host->render_valid)
    return launch_step_tuple(*spec_host, spec_dev, st, actions, out, B, reset_first, s);
  if (out.obs_format != CAMPX_OBS_INT8) return CAMPX_EINVAL;
  if (use_table)
    return launch_table(*spec_host, spec_dev, st, actions, out, B, T, reset_first, emit_first, s);
  // (T == 0: the its_showtime() observation - positions from the art, no rule is run)
  if (only && T > 0) return CAMPX_ESPEC;
  return launch_interp(*spec_host, spec_dev, st, actions, out, B, T, reset_first, emit_first, s);
}

// Pack one frame's outcome of every (cell, ..., cell, action) tuple into the pair / tuple
// table format (include/campx_hip.h, campx_pair_table_build): `h_table` = 256 floats (the
// reward list) followed by the entries.  CAMPX_ESPEC for more than 256 distinct rewards.
int32_t pack_state_table(const CampxSpec& spec, size_t n, const uint8_t* h_trace,
                         const float* h_reward, const uint8_t* h_done, const int8_t* h_perf,
                         float* h_table) {
  const int K = spec.n_dyn;
  uint32_t* h_entries32 = reinterpret_cast<uint32_t*>(h_table + 256);
  uint64_t* h_entries64 = reinterpret_cast<uint64_t*>(h_table + 256);
  int n_rewards = 0;
  uint32_t reward_bits[256];
  for (int i = 0; i < 256; ++i) h_table[i] = 0.0f;
  for (size_t i = 0; i < n; ++i) {
    uint32_t bits;
    memcpy(&bits, &h_reward[i], 4);
    int idx = -1;
    for (int k = 0; k < n_rewards; ++k) {
      if (reward_bits[k] == bits) {
        idx = k;
        break;
      }
    }
    if (idx < 0) {
      if (n_rewards == 256) return CAMPX_ESPEC;
      idx = n_rewards++;
      reward_bits[idx] = bits;
      h_table[idx] = h_reward[i];
    }
    const int pc = perf_code_of(spec, spec.perf_dyn >= 0 && h_perf ? h_perf[i] : 0);
    if (pc < 0) return CAMPX_ESPEC;   // a value the spec's scale / offset cannot give
    const uint32_t perf = (uint32_t)pc;
    const uint32_t over = (uint32_t)(h_done[i] & 1), dcode = (uint32_t)(h_done[i] >> 4);
    if (K == 2) {
      const uint32_t ta = h_trace[i], tb = h_trace[n + i];
      h_entries32[i] = (ta & 0x7fu) | ((tb & 0x7fu) << 7) | ((ta >> 7) << 14) | ((tb >> 7) << 15) |
                       (over << 16) | ((perf & 3u) << 17) | ((uint32_t)idx << 19) | (dcode << 27) |
                       ((perf >> 2) << 31);
    } else {
      uint32_t lo = 0;
      for (int d = 0; d < K; ++d) {
        const uint32_t tr = h_trace[(size_t)d * n + i];
        lo |= ((tr & 0x7fu) << (7 * d)) | ((tr >> 7) << (28 + d));
      }
      const uint32_t hi = over | ((perf & 3u) << 1) | ((uint32_t)idx << 3) | ((perf >> 2) << 11) |
                          (dcode << 12);
      h_entries64[i] = (uint64_t)lo | ((uint64_t)hi << 32);
    }
  }
  return CAMPX_OK;
}

}  // namespace campx_impl

using namespace campx_impl;

extern "C" {

int32_t campx_spec_size(void) { return (int32_t)sizeof(CampxSpec); }

int64_t campx_flow_scratch_bytes(int64_t B, int32_t T) {
  return B > 0 && T > 0 ? flow_scratch_bytes(B, T) : 0;
}

int32_t campx_spec_validate(const CampxSpec* s) {
  if (!s) return CAMPX_EINVAL;
  if (s->magic != CAMPX_SPEC_MAGIC || s->version != CAMPX_SPEC_VERSION) return CAMPX_ESPEC;
  if (s->rows < 1 || s->cols < 1 || s->rows > 127 || s->cols > 127) return CAMPX_ESPEC;
  const int HW = s->rows * s->cols;
  if (HW > CAMPX_MAX_CELLS) return CAMPX_ESPEC;
  if (s->n_layers < 1 || s->n_layers > CAMPX_MAX_LAYERS) return CAMPX_ESPEC;
  if (s->n_dyn < 1 || s->n_dyn > CAMPX_MAX_DYN) return CAMPX_ESPEC;
  if (s->n_static < 0 || s->n_static > CAMPX_MAX_STATIC) return CAMPX_ESPEC;
  if (s->n_rules < 0 || s->n_rules > CAMPX_MAX_RULES) return CAMPX_ESPEC;
  for (int d = 0; d < s->n_dyn; ++d) {
    if (s->dyn_layer[d] < 0 || s->dyn_layer[d] >= s->n_layers) return CAMPX_ESPEC;
    // (z rank 0 = behind the backdrop: a tracked value that is never painted - the z-order
    // mode of a host-tabulated game that re-orders its things, campx_amd/tabulate.py)
    if (s->dyn_z[d] < (s->table_only == 1 && d > 0 ? 0 : 1) || s->dyn_z[d] > 255) return CAMPX_ESPEC;
    if (s->dyn_row0[d] < 0 || s->dyn_row0[d] >= s->rows) return CAMPX_ESPEC;
    if (s->dyn_col0[d] < 0 || s->dyn_col0[d] >= s->cols) return CAMPX_ESPEC;
  }
  for (int i = 0; i < HW; ++i) {
    if (s->static_top_layer[i] >= s->n_layers) return CAMPX_ESPEC;
    if (s->n_static < 16 && (s->static_cover[i] >> s->n_static)) return CAMPX_ESPEC;
  }
  for (int i = 0; i < s->n_layers * HW; ++i)
    if (s->obs_template[i] != 0 && s->obs_template[i] != 1) return CAMPX_ESPEC;
  for (int i = 0; i < s->n_rules; ++i) {
    const CampxRule& r = s->rules[i];
    if (r.dyn < 0 || r.dyn >= s->n_dyn) return CAMPX_ESPEC;
    switch (r.op) {
      case CAMPX_OP_AGENT:
        break;
      case CAMPX_OP_DIR_HOVER:
        if (r.aux < 0 || r.aux >= s->n_layers) return CAMPX_ESPEC;
        break;
      case CAMPX_OP_BOX:
        if (r.aux < 0 || r.aux >= s->n_dyn) return CAMPX_ESPEC;
        break;
      case CAMPX_OP_GOAL:
        if (r.aux < 0 || r.aux >= s->n_static) return CAMPX_ESPEC;
        break;
      default:
        return CAMPX_ESPEC;
    }
  }
  if (s->n_rules > 0 && !s->rules[s->n_rules - 1].end_group) return CAMPX_ESPEC;
  if (s->table_only != 0 && s->table_only != 1) return CAMPX_ESPEC;
  if (s->table_only) {
    if (s->n_rules != 0) return CAMPX_ESPEC;
    // (one mover: the host-filled transition table is the game)
    if (s->n_dyn == 1 && !s->table_valid) return CAMPX_ESPEC;
  }
  // Whoever says the table is filled - campx_spec_compile() for a rule game, the host tabulator, or
  // a C caller with a table of its own - has its entries looked at: the table kernels index the
  // board with `next_cell` and the scenery's layers with `paint` (until round 5 only table_only
  // games were checked: a hand-filled table on a rule game went through to the kernels).
  if (s->table_valid && s->n_dyn == 1)
    for (int i = 0; i < HW * CAMPX_N_ACTIONS; ++i)
      if (s->table[i].next_cell >= HW || (s->table[i].done & 0x0eu) ||
          (s->table[i].paint & 0x7fu) >= (uint32_t)s->n_layers)
        return CAMPX_ESPEC;
  if (s->perf_dyn < -1 || s->perf_dyn >= s->n_dyn) return CAMPX_ESPEC;
  if (s->perf_dyn >= 0) {
    if (s->perf_scale == 0 || s->perf_scale < -16 || s->perf_scale > 16) return CAMPX_ESPEC;
    if (s->perf_offset < -16 || s->perf_offset > 16) return CAMPX_ESPEC;
    if (s->perf_mode == 0) {          // progress round a cycle of cell classes
      if (s->perf_n < 2 || s->perf_n > 255) return CAMPX_ESPEC;
      for (int i = 0; i < HW; ++i)
        if (s->cell_class[i] > s->perf_n) return CAMPX_ESPEC;
    } else if (s->perf_mode == 1) {   // penalty classes of where the things of perf_mask stand
      if (s->perf_mask <= 0 || (s->perf_mask >> s->n_dyn)) return CAMPX_ESPEC;
      int most = 0, things = 0;
      for (int i = 0; i < HW; ++i) most = s->cell_class[i] > most ? s->cell_class[i] : most;
      for (int d = 0; d < s->n_dyn; ++d) things += (s->perf_mask >> d) & 1;
      if (most * things > 7) return CAMPX_ESPEC;   // the tables carry the sum in 3 bits
    } else {
      return CAMPX_ESPEC;
    }
  }
  for (int i = 1; i < 16; ++i)
    if (!(s->discount_list[i] >= 0.0f && s->discount_list[i] <= 1.0f)) return CAMPX_ESPEC;
  return CAMPX_OK;
}

int32_t campx_spec_compile(CampxSpec* spec, void* stream) {
  const int32_t v = campx_spec_validate(spec);
  if (v != CAMPX_OK) return v;
  if (!spec->table_only) spec->table_valid = 0;   // (a host-tabulated game: its table IS the game)
  {
    const int HW = spec->rows * spec->cols, LHW = spec->n_layers * HW;
    const int pitch_obs = ((LHW + 15) & ~15) + 16, pitch_board = ((HW + 15) & ~15) + 16;
    for (int r = 0; r < 16; ++r) {
      for (int j = 0; j < pitch_obs; ++j)
        spec->rot_obs[r * pitch_obs + j] = spec->obs_template[(j + r) % LHW];
      for (int j = 0; j < pitch_board; ++j)
        spec->rot_board[r * pitch_board + j] =
            (int8_t)spec->layer_char[spec->static_top_layer[(j + r) % HW]];
    }
    spec->render_valid = 1;
  }
  if (spec->n_dyn != 1 || spec->table_only) return CAMPX_OK;
  const int W = spec->cols, HW = spec->rows * spec->cols;
  const int n = HW * CAMPX_N_ACTIONS;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // One scratch allocation: spec | trace | reward | pos | done | actions | done_out
  const size_t off_obs = (sizeof(CampxSpec) + 255) & ~(size_t)255;
  const size_t off_reward = (off_obs + (size_t)n + 255) & ~(size_t)255;   // (off_obs: the trace)
  const size_t off_pos = off_reward + sizeof(float) * n;
  const size_t off_done = off_pos + 2 * (size_t)n;
  const size_t off_act = off_done + n;
  const size_t off_dout = off_act + n;
  const size_t off_perf = off_dout + n;
  const size_t total = off_perf + n;
  char* dev = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&dev), total);
  if (e != hipSuccess) return hip_failed(e);
  // host images of pos / actions: pseudo-environment i = (cell i/5, action i%5)
  int8_t* host = static_cast<int8_t*>(malloc(8 * (size_t)n + sizeof(float) * n));
  if (!host) {
    (void)hipFree(dev);
    return CAMPX_ENOMEM;
  }
  int8_t* h_pos = host;
  int8_t* h_act = host + 2 * n;
  uint8_t* h_done = reinterpret_cast<uint8_t*>(host + 3 * n);
  int8_t* h_perf = host + 4 * n;
  float* h_reward = reinterpret_cast<float*>(host + 8 * n);
  for (int i = 0; i < n; ++i) {
    const int cell = i / CAMPX_N_ACTIONS;
    h_pos[i] = (int8_t)(cell / W);
    h_pos[n + i] = (int8_t)(cell % W);
    h_act[i] = (int8_t)(i % CAMPX_N_ACTIONS);
  }
  int32_t rc = CAMPX_OK;
#define CAMPX_TRY(call)           \
  do {                            \
    e = (call);                   \
    if (e != hipSuccess) {        \
      rc = hip_failed(e);         \
      goto done;                  \
    }                             \
  } while (0)
  CAMPX_TRY(hipMemcpyAsync(dev, spec, sizeof(CampxSpec), hipMemcpyHostToDevice, s));
  CAMPX_TRY(hipMemcpyAsync(dev + off_pos, h_pos, 2 * (size_t)n, hipMemcpyHostToDevice, s));
  CAMPX_TRY(hipMemcpyAsync(dev + off_act, h_act, (size_t)n, hipMemcpyHostToDevice, s));
  CAMPX_TRY(hipMemsetAsync(dev + off_done, 0, (size_t)n, s));
  {
    CampxState st = {reinterpret_cast<int8_t*>(dev + off_pos),
                     reinterpret_cast<uint8_t*>(dev + off_done), nullptr, nullptr};
    CampxOutputs out;
    memset(&out, 0, sizeof(out));
    out.reward = reinterpret_cast<float*>(dev + off_reward);
    out.done = reinterpret_cast<uint8_t*>(dev + off_dout);
    out.perf = spec->perf_dyn >= 0 ? reinterpret_cast<int8_t*>(dev + off_perf) : nullptr;
    out.trace = reinterpret_cast<uint8_t*>(dev + off_obs);   // [1, 1, n], not read back
    // the interpreter in trace mode (no observation image): one frame of every (cell, action)
    launch_trace(*spec, reinterpret_cast<const CampxSpec*>(dev), st,
                      reinterpret_cast<const int8_t*>(dev + off_act), out, n, 1, 0, (int64_t)n, s);
    CAMPX_TRY(hipGetLastError());
  }
  CAMPX_TRY(hipMemcpyAsync(h_pos, dev + off_pos, 2 * (size_t)n, hipMemcpyDeviceToHost, s));
  CAMPX_TRY(hipMemcpyAsync(h_done, dev + off_dout, (size_t)n, hipMemcpyDeviceToHost, s));
  CAMPX_TRY(hipMemcpyAsync(h_perf, dev + off_perf, (size_t)n, hipMemcpyDeviceToHost, s));
  CAMPX_TRY(hipMemcpyAsync(h_reward, dev + off_reward, sizeof(float) * n, hipMemcpyDeviceToHost, s));
  CAMPX_TRY(hipStreamSynchronize(s));
#undef CAMPX_TRY
  for (int i = 0; i < n; ++i) {
    CampxTransition& tr = spec->table[i];
    tr.reward = h_reward[i];
    tr.next_cell = (uint8_t)((int)h_pos[i] * W + (int)h_pos[n + i]);
    tr.done = h_done[i];
    tr.perf = spec->perf_dyn >= 0 ? h_perf[i] : (int8_t)0;
    tr.paint = (uint8_t)(spec->static_top_layer[tr.next_cell] |
                         (spec->static_top_z[tr.next_cell] > spec->dyn_z[0] ? 0x80u : 0u));
  }
  spec->table_valid = 1;
done:
  free(host);
  (void)hipFree(dev);
  return rc;
}

int64_t campx_pair_table_bytes(const CampxSpec* spec) {
  if (!spec || campx_spec_validate(spec) != CAMPX_OK || spec->n_dyn < 2) return 0;
  const int64_t HW = (int64_t)spec->rows * spec->cols;
  if (HW > 128) return 0;  // cells are 7-bit fields
  int64_t n = CAMPX_N_ACTIONS;
  for (int d = 0; d < spec->n_dyn; ++d) n *= HW;
  if (spec->n_dyn == 2) {
    const int64_t bytes = 256 * (int64_t)sizeof(float) + n * (int64_t)sizeof(uint32_t);
    return bytes <= (1 << 20) ? bytes : 0;
  }
  // three / four movers: 64-bit entries, read from global memory
  const int64_t bytes = 256 * (int64_t)sizeof(float) + n * (int64_t)sizeof(uint64_t);
  return bytes <= kTupleTableMaxBytes ? bytes : 0;
}

int32_t campx_pair_table_build(const CampxSpec* spec, const CampxSpec* spec_dev, void* table_dev,
                               void* stream) {
  const int64_t bytes = campx_pair_table_bytes(spec);
  if (bytes == 0 || !spec_dev || !table_dev) return CAMPX_EINVAL;
  if (spec->table_only) return CAMPX_ESPEC;   // no rules to run: campx_pair_table_pack()
  const int K = spec->n_dyn, W = spec->cols, HW = spec->rows * spec->cols;
  size_t n = CAMPX_N_ACTIONS;
  for (int d = 0; d < K; ++d) n *= (size_t)HW;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // device scratch: reward[n] | trace[K][n] | pos[2K][n] | done[n] | actions[n] | done_out[n] | perf[n]
  // (the interpreter in trace mode writes no observations)
  const size_t off_trace = sizeof(float) * n;
  const size_t off_pos = off_trace + (size_t)K * n;
  const size_t off_done = off_pos + 2 * (size_t)K * n;
  const size_t off_act = off_done + n;
  const size_t off_dout = off_act + n;
  const size_t off_perf = off_dout + n;
  const size_t total = off_perf + n;
  char* dev = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&dev), total);
  if (e != hipSuccess) return hip_failed(e);
  // host scratch, widest arrays first so that nothing needs an alignment pad:
  // table (256 floats + entries) | reward[n] | pos[2K][n] | act[n] | done[n] | perf[n] | trace[K][n]
  const size_t table_bytes = ((size_t)bytes + 7) & ~(size_t)7;
  const size_t host_bytes = table_bytes + n * 4 + n * (size_t)(2 * K + 1 + 1 + 1 + K);
  char* host = static_cast<char*>(malloc(host_bytes));
  if (!host) {
    (void)hipFree(dev);
    return CAMPX_ENOMEM;
  }
  float* h_table = reinterpret_cast<float*>(host);
  float* h_reward = reinterpret_cast<float*>(host + table_bytes);
  int8_t* h_pos = reinterpret_cast<int8_t*>(h_reward + n);
  int8_t* h_act = h_pos + 2 * (size_t)K * n;
  uint8_t* h_done = reinterpret_cast<uint8_t*>(h_act + n);
  int8_t* h_perf = reinterpret_cast<int8_t*>(h_done + n);
  uint8_t* h_trace = reinterpret_cast<uint8_t*>(h_perf + n);
  for (size_t i = 0; i < n; ++i) {  // index = ((cell_0 * HW + cell_1) * HW + ...) * 5 + action
    size_t rest = i / CAMPX_N_ACTIONS;
    h_act[i] = (int8_t)(i % CAMPX_N_ACTIONS);
    for (int d = K - 1; d >= 0; --d) {
      const int cell = (int)(rest % (size_t)HW);
      rest /= (size_t)HW;
      h_pos[(size_t)(2 * d) * n + i] = (int8_t)(cell / W);
      h_pos[(size_t)(2 * d + 1) * n + i] = (int8_t)(cell % W);
    }
  }
  int32_t rc = CAMPX_OK;
#define CAMPX_TRY(call)           \
  do {                            \
    e = (call);                   \
    if (e != hipSuccess) {        \
      rc = hip_failed(e);         \
      goto done;                  \
    }                             \
  } while (0)
  CAMPX_TRY(hipMemcpyAsync(dev + off_pos, h_pos, 2 * (size_t)K * n, hipMemcpyHostToDevice, s));
  CAMPX_TRY(hipMemcpyAsync(dev + off_act, h_act, n, hipMemcpyHostToDevice, s));
  CAMPX_TRY(hipMemsetAsync(dev + off_done, 0, n, s));
  CAMPX_TRY(hipMemsetAsync(dev + off_perf, 0, n, s));
  {
    CampxState st = {reinterpret_cast<int8_t*>(dev + off_pos),
                     reinterpret_cast<uint8_t*>(dev + off_done), nullptr, nullptr};
    CampxOutputs out;
    memset(&out, 0, sizeof(out));
    out.reward = reinterpret_cast<float*>(dev);
    out.done = reinterpret_cast<uint8_t*>(dev + off_dout);
    out.perf = spec->perf_dyn >= 0 ? reinterpret_cast<int8_t*>(dev + off_perf) : nullptr;
    out.trace = reinterpret_cast<uint8_t*>(dev + off_trace);
    // the interpreter in trace mode: positions, visibility, reward, done, perf
    const int8_t* acts = reinterpret_cast<const int8_t*>(dev + off_act);
    launch_trace(*spec, spec_dev, st, acts, out, (int64_t)n, 1, 0, (int64_t)n, s);
    CAMPX_TRY(hipGetLastError());
  }
  CAMPX_TRY(hipMemcpyAsync(h_trace, dev + off_trace, (size_t)K * n, hipMemcpyDeviceToHost, s));
  CAMPX_TRY(hipMemcpyAsync(h_reward, dev, sizeof(float) * n, hipMemcpyDeviceToHost, s));
  CAMPX_TRY(hipMemcpyAsync(h_done, dev + off_dout, n, hipMemcpyDeviceToHost, s));
  CAMPX_TRY(hipMemcpyAsync(h_perf, dev + off_perf, n, hipMemcpyDeviceToHost, s));
  CAMPX_TRY(hipStreamSynchronize(s));
  rc = pack_state_table(*spec, n, h_trace, h_reward, h_done, h_perf, h_table);
  if (rc != CAMPX_OK) goto done;
  CAMPX_TRY(hipMemcpyAsync(table_dev, h_table, (size_t)bytes, hipMemcpyHostToDevice, s));
  CAMPX_TRY(hipStreamSynchronize(s));
#undef CAMPX_TRY
done:
  free(host);
  (void)hipFree(dev);
  return rc;
}

int32_t campx_pair_table_pack(const CampxSpec* spec, const uint8_t* trace, const float* reward,
                              const uint8_t* done, const int8_t* perf, void* table_dev,
                              void* stream) {
  const int64_t bytes = campx_pair_table_bytes(spec);
  if (bytes == 0 || !trace || !reward || !done || !table_dev) return CAMPX_EINVAL;
  const int K = spec->n_dyn, HW = spec->rows * spec->cols;
  size_t n = CAMPX_N_ACTIONS;
  for (int d = 0; d < K; ++d) n *= (size_t)HW;
  for (size_t i = 0; i < (size_t)K * n; ++i)
    if ((trace[i] & 0x7fu) >= (uint32_t)HW) return CAMPX_EINVAL;   // a cell outside the board
  float* h_table = static_cast<float*>(malloc(((size_t)bytes + 7) & ~(size_t)7));
  if (!h_table) return CAMPX_ENOMEM;
  int32_t rc = pack_state_table(*spec, n, trace, reward, done, perf, h_table);
  if (rc == CAMPX_OK) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemcpyAsync(table_dev, h_table, (size_t)bytes, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) rc = hip_failed(e);
  }
  free(h_table);
  return rc;
}

int32_t campx_reset_launch(const CampxSpec* spec_host, const CampxSpec* spec_dev, CampxState state,
                           CampxOutputs out, int64_t B, void* stream) {
  return launch(spec_host, spec_dev, state, nullptr, out, B, 0, 1, 1, stream);
}

int32_t campx_rollout_launch(const CampxSpec* spec_host, const CampxSpec* spec_dev,
                             CampxState state, const int8_t* actions, CampxOutputs out, int64_t B,
                             int32_t T, int32_t reset_first, void* stream) {
  return launch(spec_host, spec_dev, state, actions, out, B, T, reset_first, 0, stream);
}

int32_t campx_update_launch(const CampxSpec* spec_host, const CampxSpec* spec_dev, CampxState st,
                            const int8_t* actions, CampxOutputs out, int64_t B, int32_t T,
                            int32_t reset_first, void* stream) {
  if (!spec_host || !spec_dev || !st.pos || !st.done || !actions || !out.trace || B <= 0 || T <= 0)
    return CAMPX_EINVAL;
  if (out.perf && spec_host->perf_dyn < 0) return CAMPX_EINVAL;
  const int32_t v = campx_spec_validate(spec_host);
  if (v != CAMPX_OK) return v;
  if (!spec_host->render_valid) return CAMPX_ESPEC;
  const bool use_table = spec_host->table_valid && spec_host->n_dyn == 1;
  if (out.scalar_pitch && out.scalar_pitch < B) return CAMPX_EINVAL;
  return launch_update(*spec_host, spec_dev, st, actions, out, B, T, reset_first, use_table,
                       (int64_t)T * row_pitch(out, B), static_cast<hipStream_t>(stream));
}

int32_t campx_render_launch(const CampxSpec* spec_host, const CampxSpec* spec_dev, CampxOutputs out,
                            int64_t B, int32_t T, void* stream) {
  if (!spec_host || !spec_dev || !out.trace || !out.obs || B <= 0 || T <= 0 || T > 65535)
    return CAMPX_EINVAL;
  if (reinterpret_cast<uintptr_t>(out.obs) & 15) return CAMPX_EINVAL;
  if (out.obs_format < CAMPX_OBS_INT8 || out.obs_format > CAMPX_OBS_BF16) return CAMPX_EINVAL;
  const int32_t v = campx_spec_validate(spec_host);
  if (v != CAMPX_OK) return v;
  CampxOutputs probe = out;   // the conditions of the two-kernel path, frames back to back
  if (!split_ok(*spec_host, probe, B, T)) return CAMPX_EINVAL;
  if (out.scalar_pitch && out.scalar_pitch < B) return CAMPX_EINVAL;
  return launch_renders(*spec_host, spec_dev, out, B, T, (int64_t)T * row_pitch(out, B),
                        static_cast<hipStream_t>(stream));
}

int32_t campx_update_render_launch(const CampxSpec* spec_host, const CampxSpec* spec_dev,
                                   CampxState st, const int8_t* actions, CampxOutputs out,
                                   CampxOutputs prev, int64_t B, int32_t T, int32_t reset_first,
                                   void* stream) {
  if (!prev.trace)
    return campx_update_launch(spec_host, spec_dev, st, actions, out, B, T, reset_first, stream);
  // (every check of the two calls this one stands for, before anything is launched)
  if (!spec_host || !spec_dev || !st.pos || !st.done || !actions || !out.trace || B <= 0 || T <= 0 ||
      T > 65535 || !prev.obs || prev.trace == out.trace)
    return CAMPX_EINVAL;
  if (out.perf && spec_host->perf_dyn < 0) return CAMPX_EINVAL;
  if (reinterpret_cast<uintptr_t>(prev.obs) & 15) return CAMPX_EINVAL;
  if (prev.obs_format < CAMPX_OBS_INT8 || prev.obs_format > CAMPX_OBS_BF16) return CAMPX_EINVAL;
  const int32_t v = campx_spec_validate(spec_host);
  if (v != CAMPX_OK) return v;
  if (!spec_host->render_valid) return CAMPX_ESPEC;
  if ((out.scalar_pitch && out.scalar_pitch < B) || (prev.scalar_pitch && prev.scalar_pitch < B))
    return CAMPX_EINVAL;
  CampxOutputs probe = prev;
  if (!split_ok(*spec_host, probe, B, T)) return CAMPX_EINVAL;
  const bool use_table = spec_host->table_valid && spec_host->n_dyn == 1;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // (games of two to four movers share the launch too, over the caller's pair / tuple table)
  const bool multi_table = spec_host->n_dyn >= 2 && st.pair_table;
  if (pipe_ok(*spec_host, out, prev, B, T, use_table || multi_table))
    return launch_pipe(*spec_host, spec_dev, st, actions, out, prev, B, T, reset_first, s);
  const int32_t rc = launch_update(*spec_host, spec_dev, st, actions, out, B, T, reset_first, use_table,
                                   (int64_t)T * row_pitch(out, B), s);
  if (rc != CAMPX_OK) return rc;
  return launch_renders(*spec_host, spec_dev, prev, B, T, (int64_t)T * row_pitch(prev, B), s);
}

int32_t campx_update_render_shared(const CampxSpec* spec_host, int64_t B, int32_t T) {
  if (!spec_host || B <= 0 || T <= 0 || campx_spec_validate(spec_host) != CAMPX_OK) return 0;
  const bool use_table = spec_host->table_valid && spec_host->n_dyn == 1;
  CampxOutputs prev{};       // int8 observations of every frame, back to back, 16-byte aligned
  prev.obs = reinterpret_cast<int8_t*>(uintptr_t{4096});
  prev.trace = reinterpret_cast<uint8_t*>(uintptr_t{4096});
  prev.obs_t_stride = B * spec_host->n_layers * spec_host->rows * spec_host->cols;
  prev.obs_format = CAMPX_OBS_INT8;
  // (two to four movers: with their pair / tuple table in CampxState.pair_table, which this
  // query cannot see - a caller without one gets two launches whatever this says)
  const bool multi_table = spec_host->n_dyn >= 2;
  return pipe_ok(*spec_host, prev, prev, B, T, use_table || multi_table) ? 1 : 0;
}

int32_t campx_flow_shared(const CampxSpec* spec_host, int64_t B, int32_t T, int64_t scalar_pitch) {
  if (!spec_host || B <= 0 || T <= 0 || campx_spec_validate(spec_host) != CAMPX_OK) return 0;
  const bool use_table = spec_host->table_valid && spec_host->n_dyn == 1;
  // a call as the header describes it: every output this path looks at present and aligned
  static CampxFlowState some_state;        // (never read or written: flow_ok only asks whether there is one)
  static int32_t some_flag;
  CampxOutputs out{};
  out.obs = reinterpret_cast<int8_t*>(uintptr_t{4096});
  out.trace = reinterpret_cast<uint8_t*>(uintptr_t{4096});
  out.obs_t_stride = B * spec_host->n_layers * spec_host->rows * spec_host->cols;
  out.obs_format = CAMPX_OBS_INT8;
  out.scalar_pitch = scalar_pitch;
  out.overlap_ctl = reinterpret_cast<uint32_t*>(uintptr_t{4096});
  out.overlap_ctl_bytes = flow_scratch_bytes(B, T);
  out.flow_state = &some_state;
  out.error_flag = &some_flag;
  // (two to four movers: with their pair / tuple table in CampxState.pair_table, which this query
  // cannot see - a caller without one gets two launches whatever this says)
  const bool multi_table = spec_host->n_dyn >= 2;
  return flow_ok(*spec_host, out, B, T, use_table || multi_table, nullptr, /*ask_stream=*/false) ? 1 : 0;
}

int32_t campx_config_set(const char* name, int64_t value) {
  if (!name) return CAMPX_EINVAL;
  (void)knob(K_TRACE_CHUNK_MB);           // (the environment first: an explicit call overrides it)
  for (KnobRow& row : g_knobs)
    if (!strcmp(name, row.name)) {
      if (value < row.lo || value > row.hi) return CAMPX_EINVAL;
      __atomic_store_n(&row.value, value, __ATOMIC_RELAXED);
      return CAMPX_OK;
    }
  return CAMPX_EINVAL;
}

int32_t campx_config_get(const char* name, int64_t* value) {
  if (!name || !value) return CAMPX_EINVAL;
  for (int k = 0; k < K_COUNT; ++k)
    if (!strcmp(name, g_knobs[k].name)) {
      *value = knob((Knob)k);
      return CAMPX_OK;
    }
  return CAMPX_EINVAL;
}

int32_t campx_config_string(char* buf, int32_t buf_len) {
  std::string text;
  for (int k = 0; k < K_COUNT; ++k) {
    if (k) text += ' ';
    text += g_knobs[k].name;
    text += '=';
    text += std::to_string((long long)knob((Knob)k));
  }
  if (buf && buf_len > 0) {
    strncpy(buf, text.c_str(), (size_t)buf_len - 1);
    buf[buf_len - 1] = '\0';
  }
  return (int32_t)text.size() + 1;
}

const char* campx_strerror(int32_t code) {
  switch (code) {
    case CAMPX_OK:
      return "ok";
    case CAMPX_EINVAL:
      return "invalid argument (NULL, misaligned or out of range)";
    case CAMPX_ESPEC:
      return "GameSpec failed validation";
    case CAMPX_ELAUNCH:
      return "HIP launch failed (see campx_last_hip_error)";
    case CAMPX_ENODEV:
      return "no usable HIP device";
    case CAMPX_ENOMEM:
      return "out of host memory";
    default:
      return "unknown campx error";
  }
}

int32_t campx_last_hip_error(void) { return g_last_hip_error; }

int32_t campx_device_arch(int32_t ordinal, char* buf, int32_t buf_len) {
  if (!buf || buf_len < 2) return CAMPX_EINVAL;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || ordinal < 0 || ordinal >= n) return CAMPX_ENODEV;
  hipDeviceProp_t prop;
  const hipError_t e = hipGetDeviceProperties(&prop, ordinal);
  if (e != hipSuccess) return hip_failed(e);
  strncpy(buf, prop.gcnArchName, (size_t)buf_len - 1);
  buf[buf_len - 1] = '\0';
  return CAMPX_OK;
}

}  // extern "C"
