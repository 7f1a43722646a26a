// campx_torch.cpp - the torch custom-op face of libcampx_hip.so.
//
// `Engine.play()` / `Engine.rollout()` / `Engine.its_showtime()` of a batched engine
// lower to these ops (campx_amd/fused.py); they are what replaces the reference's
// per-frame Python path  Engine.play -> _update_and_render -> <entity>.update ->
// _render -> renderer.render  (campx/engine.py:114-324, campx/rendering.py:104-219).
//
//   campx::reset    its_showtime(): state from the art + the first observation
//   campx::step     one Engine.play() frame for B environments
//   campx::rollout  T consecutive frames in one launch
//   campx::update / campx::render   the two kernels of a rollout as separate ops
//   campx::update_render            update of one rollout + render of the one before it
//   campx::shape_rollout            the shape tier (Hello World): reset / step / rollout
//   campx::wide_rollout             the wide tier (boards above 128 cells): reset / step / rollout
//   campx::onehot_to_ids / campx::check_actions   action-format helpers
//
// Contract: every tensor is caller-owned and contiguous; outputs are written in
// place (declared mutable in the schema, so functionalization and torch.compile see
// the writes); the work is enqueued on torch's CURRENT HIP stream of the tensors'
// device and nothing synchronises.  Registered for the CUDA dispatch key (= HIP on
// ROCm) and Meta (shape-free no-op, which is also the fake-tensor implementation).
// There is deliberately no CPU kernel: calling these with CPU state raises.
// An ADInplaceOrView kernel (one boxed function, registered for every op) bumps the version
// counter of every tensor an op writes, so that autograd refuses a backward pass through a
// tensor that a later play() / rollout() has overwritten (the engine's frame buffers are
// reused) instead of silently differentiating stale data.
// (ROCm builds of torch present HIP devices as device type "cuda"; the guard and stream
// types below are torch's own names for that arrangement.)
//
// The ops only unpack tensors into the C ABI of include/campx_hip.h; all kernels
// live in csrc/k_*.hip.

#include <ATen/core/Tensor.h>
#include <ATen/core/dispatch/Dispatcher.h>
#include <ATen/core/stack.h>
#include <c10/core/impl/LocalDispatchKeySet.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <hip/hip_runtime.h>
#include <torch/library.h>

#include <map>
#include <mutex>
#include <optional>
#include <unordered_map>
#include <vector>

#include "campx_hip.h"

namespace {

using at::Tensor;
using OptTensor = std::optional<Tensor>;

void check_ok(int32_t rc, const char* what) {
  TORCH_CHECK(rc == CAMPX_OK, what, " failed: ", campx_strerror(rc), " (code ", rc, ", hipError ",
              campx_last_hip_error(), ")");
}

const CampxSpec* host_spec(const Tensor& spec_host) {
  TORCH_CHECK(spec_host.device().is_cpu() && spec_host.scalar_type() == at::kByte &&
                  spec_host.is_contiguous() && spec_host.numel() == (int64_t)sizeof(CampxSpec),
              "campx: spec_host must be a contiguous CPU uint8 tensor of ", sizeof(CampxSpec),
              " bytes (the CampxSpec blob)");
  return reinterpret_cast<const CampxSpec*>(spec_host.data_ptr());
}

void want(const Tensor& t, const char* name, at::ScalarType dtype, const c10::Device& dev,
          c10::IntArrayRef shape) {
  TORCH_CHECK(t.device() == dev, "campx: ", name, " must be on ", dev, ", it is on ", t.device());
  TORCH_CHECK(t.scalar_type() == dtype, "campx: ", name, " must be ", dtype, ", it is ",
              t.scalar_type());
  TORCH_CHECK(t.is_contiguous(), "campx: ", name, " must be contiguous");
  TORCH_CHECK(t.sizes() == shape, "campx: ", name, " must have shape ", shape, ", it has ",
              t.sizes());
}

// The per-frame scalar streams [T, B] and the trace [K, T, B] may be PADDED: contiguous
// within a row, rows `pitch` >= B elements apart, the same pitch for every stream of a call
// (the planes of the trace then T * pitch apart).  rollout_buffers() pads to a multiple of
// 16 when the batch size is not one, so that every row starts aligned
// (CampxOutputs.scalar_pitch).  `pitch` is 0 until the first stream has been seen.
void want_rows(const Tensor& t, const char* name, at::ScalarType dtype, const c10::Device& dev,
               int64_t T, int64_t B, int64_t& pitch) {
  TORCH_CHECK(t.device() == dev, "campx: ", name, " must be on ", dev, ", it is on ", t.device());
  TORCH_CHECK(t.scalar_type() == dtype, "campx: ", name, " must be ", dtype, ", it is ",
              t.scalar_type());
  TORCH_CHECK(t.dim() == 2 && t.size(0) == T && t.size(1) == B, "campx: ", name,
              " must have shape [", T, ", ", B, "], it has ", t.sizes());
  TORCH_CHECK(B == 1 || t.stride(1) == 1, "campx: ", name, " must be contiguous within a row");
  if (T == 1) {
    // One row: its own pitch is never used, but the call's pitch (taken from the planes of the
    // trace, want_trace() runs first) decides whether the update kernels store the row's last
    // 16-element group whole (row_extent()): the row must then reach that far.
    const int64_t up = (B + 15) / 16 * 16;
    if (pitch >= up) {
      const int64_t room = static_cast<int64_t>(t.storage().nbytes() / t.element_size()) -
                           t.storage_offset();
      TORCH_CHECK(room >= up, "campx: ", name, " is a [1, ", B, "] row with room for ", room,
                  " elements, but the call's trace has a padded row pitch (", pitch,
                  "): the row must reach ", up, " elements (take every buffer from rollout_buffers())");
    }
    return;
  }
  const int64_t p = t.stride(0);
  TORCH_CHECK(p >= B && (pitch == 0 || p == pitch), "campx: ", name, " has row pitch ", p,
              "; every per-frame stream of a call must have the same pitch >= B");
  pitch = p;
}

// (call it before want_rows: with one frame only the planes of the trace tell the pitch)
void want_trace(const Tensor& t, const c10::Device& dev, int64_t K, int64_t T, int64_t B,
                int64_t& pitch) {
  TORCH_CHECK(t.device() == dev && t.scalar_type() == at::kByte && t.dim() == 3 &&
                  t.size(0) == K && t.size(1) == T && t.size(2) == B,
              "campx: trace must be uint8 [", K, ", ", T, ", ", B, "] on ", dev);
  TORCH_CHECK(B == 1 || t.stride(2) == 1, "campx: trace must be contiguous within a row");
  int64_t p = pitch;
  if (T > 1) p = t.stride(1);
  else if (K > 1) p = t.stride(0);
  if (p == 0) return;            // one frame, one plane: nothing to tell
  TORCH_CHECK(p >= B && (pitch == 0 || p == pitch) && (K == 1 || t.stride(0) == T * p),
              "campx: trace must have the row pitch of the other per-frame streams and planes "
              "T * pitch apart");
  pitch = p;
}

template <typename T>
T* opt_ptr(const OptTensor& t) {
  return t.has_value() ? reinterpret_cast<T*>(t->data_ptr()) : nullptr;
}

// Device-visible address of the bad-action flag: device memory, or pinned host memory
// (so that the host can poll it without a stream synchronisation).
int32_t* flag_ptr(const OptTensor& t, const c10::Device& dev) {
  if (!t.has_value()) return nullptr;
  TORCH_CHECK(t->scalar_type() == at::kInt && t->numel() >= 1 && t->is_contiguous(),
              "campx: bad_flag must be a contiguous int32 tensor");
  if (t->device() == dev) return reinterpret_cast<int32_t*>(t->data_ptr());
  TORCH_CHECK(t->device().is_cpu() && t->is_pinned(),
              "campx: bad_flag must live on the state's device or in pinned host memory");
  void* mapped = nullptr;
  const hipError_t e = hipHostGetDevicePointer(&mapped, t->data_ptr(), 0);
  TORCH_CHECK(e == hipSuccess, "campx: hipHostGetDevicePointer(bad_flag) failed: ",
              hipGetErrorString(e));
  return static_cast<int32_t*>(mapped);
}

struct Game {
  const CampxSpec* spec_host;
  const CampxSpec* spec_dev;
  CampxState state;
  c10::Device dev;
  int64_t B, K, L, H, W;
};

Game unpack_game(const Tensor& spec_host, const Tensor& spec_dev, const Tensor& pos,
                 const Tensor& done, const OptTensor& ret, const OptTensor& pair_table) {
  const CampxSpec* hs = host_spec(spec_host);
  TORCH_CHECK(pos.device().is_cuda(), "campx: the fused tier runs on a HIP device only; state is on ",
              pos.device(), " (there is no CPU implementation of these ops)");
  const c10::Device dev = pos.device();
  TORCH_CHECK(pos.dim() == 2, "campx: pos must be [2*K, B]");
  const int64_t K = hs->n_dyn, B = pos.size(1);
  want(pos, "pos", at::kChar, dev, {2 * K, B});
  want(done, "done", at::kByte, dev, {B});
  if (ret.has_value()) want(*ret, "ret", at::kFloat, dev, {B});
  TORCH_CHECK(spec_dev.device() == dev && spec_dev.scalar_type() == at::kByte &&
                  spec_dev.is_contiguous() && spec_dev.numel() == (int64_t)sizeof(CampxSpec),
              "campx: spec_dev must be the CampxSpec blob as a uint8 tensor on ", dev);
  if (pair_table.has_value())
    TORCH_CHECK(pair_table->device() == dev && pair_table->is_contiguous() &&
                    pair_table->nbytes() == (size_t)campx_pair_table_bytes(hs),
                "campx: pair_table has the wrong size or device");
  Game g{hs,
         reinterpret_cast<const CampxSpec*>(spec_dev.data_ptr()),
         CampxState{reinterpret_cast<int8_t*>(pos.data_ptr()),
                    reinterpret_cast<uint8_t*>(done.data_ptr()), opt_ptr<float>(ret),
                    pair_table.has_value() ? pair_table->data_ptr() : nullptr},
         dev,
         B,
         K,
         hs->n_layers,
         hs->rows,
         hs->cols};
  return g;
}

int32_t obs_format_of(const Tensor& obs) {
  switch (obs.scalar_type()) {
    case at::kChar: return CAMPX_OBS_INT8;
    case at::kHalf: return CAMPX_OBS_F16;
    case at::kBFloat16: return CAMPX_OBS_BF16;
    default: TORCH_CHECK(false, "campx: obs must be int8, float16 or bfloat16, it is ", obs.scalar_type());
  }
  return 0;
}

void reset(const Tensor& spec_host, const Tensor& spec_dev, Tensor& pos, Tensor& done,
           const OptTensor& ret, const OptTensor& pair_table, Tensor& obs, const OptTensor& board) {
  const Game g = unpack_game(spec_host, spec_dev, pos, done, ret, pair_table);
  want(obs, "obs", at::kChar, g.dev, {g.B, g.L, g.H, g.W});
  if (board.has_value()) want(*board, "board", at::kChar, g.dev, {g.B, g.H, g.W});
  CampxOutputs out{};
  out.obs = reinterpret_cast<int8_t*>(obs.data_ptr());
  out.board = opt_ptr<int8_t>(board);
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(g.dev);
  check_ok(campx_reset_launch(g.spec_host, g.spec_dev, g.state, out, g.B,
                              c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()),
           "campx_reset_launch");
}

void rollout(const Tensor& spec_host, const Tensor& spec_dev, Tensor& pos, Tensor& done,
             const OptTensor& ret, const OptTensor& pair_table, const Tensor& actions, Tensor& obs,
             const OptTensor& board, const OptTensor& reward, const OptTensor& discount,
             const OptTensor& step_done, const OptTensor& perf, const OptTensor& trace,
             const OptTensor& bad_count, const OptTensor& bad_flag, bool reset_first,
             const OptTensor& scratch, const OptTensor& scratch_state, const OptTensor& error_flag) {
  const Game g = unpack_game(spec_host, spec_dev, pos, done, ret, pair_table);
  TORCH_CHECK(actions.dim() == 2, "campx::rollout: actions must be int8 [T, B]");
  const int64_t T = actions.size(0);
  TORCH_CHECK(T <= 0x7fffffff, "campx::rollout: too many frames");
  want(actions, "actions", at::kChar, g.dev, {T, g.B});
  CampxOutputs out{};
  out.obs_format = obs_format_of(obs);
  const bool keep = obs.dim() == 5;  // every frame kept, else one frame buffer overwritten
  if (keep)
    want(obs, "obs", obs.scalar_type(), g.dev, {T, g.B, g.L, g.H, g.W});
  else
    want(obs, "obs", obs.scalar_type(), g.dev, {g.B, g.L, g.H, g.W});
  out.obs = reinterpret_cast<int8_t*>(obs.data_ptr());
  out.obs_t_stride = keep ? g.B * g.L * g.H * g.W : 0;
  if (board.has_value()) {
    if (board->dim() == 4) {
      want(*board, "board", at::kChar, g.dev, {T, g.B, g.H, g.W});
      out.board_t_stride = g.B * g.H * g.W;
    } else {
      want(*board, "board", at::kChar, g.dev, {g.B, g.H, g.W});
    }
    out.board = opt_ptr<int8_t>(board);
  }
  int64_t pitch = 0;
  if (trace.has_value()) want_trace(*trace, g.dev, g.K, T, g.B, pitch);
  if (reward.has_value()) want_rows(*reward, "reward", at::kFloat, g.dev, T, g.B, pitch);
  if (discount.has_value()) want_rows(*discount, "discount", at::kFloat, g.dev, T, g.B, pitch);
  if (step_done.has_value()) want_rows(*step_done, "step_done", at::kByte, g.dev, T, g.B, pitch);
  if (perf.has_value()) want_rows(*perf, "perf", at::kChar, g.dev, T, g.B, pitch);
  if (bad_count.has_value()) want(*bad_count, "bad_count", at::kInt, g.dev, {1});
  out.scalar_pitch = pitch;
  out.reward = opt_ptr<float>(reward);
  out.discount = opt_ptr<float>(discount);
  out.done = opt_ptr<uint8_t>(step_done);
  out.perf = opt_ptr<int8_t>(perf);
  out.trace = opt_ptr<uint8_t>(trace);
  out.bad_count = opt_ptr<int32_t>(bad_count);
  out.bad_flag = flag_ptr(bad_flag, g.dev);
  if (scratch.has_value()) {   // CampxOutputs.overlap_ctl: zeroed once by its owner
    // (how many bytes each user of it needs is the library's check: too small a block only
    // means the launch takes another path)
    TORCH_CHECK(scratch->device() == g.dev && scratch->scalar_type() == at::kInt &&
                    scratch->is_contiguous() && scratch->numel() >= 4,
                "campx::rollout: scratch must be a contiguous int32 tensor on ", g.dev);
    out.overlap_ctl = reinterpret_cast<uint32_t*>(scratch->data_ptr());
    out.overlap_ctl_bytes = scratch->numel() * 4;
    // the block's CampxFlowState lives with its owner, in host memory: six int64 (zeroed when
    // the block was allocated); without it, or without an error word, the library runs two launches
    if (scratch_state.has_value()) {
      TORCH_CHECK(scratch_state->device().is_cpu() && scratch_state->scalar_type() == at::kLong &&
                      scratch_state->is_contiguous() &&
                      scratch_state->numel() * 8 == (int64_t)sizeof(CampxFlowState),
                  "campx::rollout: scratch_state must be a contiguous CPU int64 tensor of ",
                  sizeof(CampxFlowState) / 8, " elements");
      out.flow_state = reinterpret_cast<CampxFlowState*>(scratch_state->data_ptr());
    }
  }
  out.error_flag = flag_ptr(error_flag, g.dev);
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(g.dev);
  check_ok(campx_rollout_launch(g.spec_host, g.spec_dev, g.state, reinterpret_cast<const int8_t*>(actions.data_ptr()),
                                out, g.B, (int32_t)T, reset_first ? 1 : 0,
                                c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()),
           "campx_rollout_launch");
}

// The two halves of the two-kernel rollout path as ops of their own, so that a caller
// can issue them on different streams (fused.py rollout(pipelined=True)).
void update(const Tensor& spec_host, const Tensor& spec_dev, Tensor& pos, Tensor& done,
            const OptTensor& ret, const OptTensor& pair_table, const Tensor& actions,
            const OptTensor& reward, const OptTensor& discount, const OptTensor& step_done,
            const OptTensor& perf, Tensor& trace, const OptTensor& bad_count,
            const OptTensor& bad_flag, bool reset_first) {
  const Game g = unpack_game(spec_host, spec_dev, pos, done, ret, pair_table);
  TORCH_CHECK(actions.dim() == 2, "campx::update: actions must be int8 [T, B]");
  const int64_t T = actions.size(0);
  TORCH_CHECK(T >= 1 && T <= 0x7fffffff, "campx::update: bad frame count");
  want(actions, "actions", at::kChar, g.dev, {T, g.B});
  int64_t pitch = 0;
  want_trace(trace, g.dev, g.K, T, g.B, pitch);
  if (reward.has_value()) want_rows(*reward, "reward", at::kFloat, g.dev, T, g.B, pitch);
  if (discount.has_value()) want_rows(*discount, "discount", at::kFloat, g.dev, T, g.B, pitch);
  if (step_done.has_value()) want_rows(*step_done, "step_done", at::kByte, g.dev, T, g.B, pitch);
  if (perf.has_value()) want_rows(*perf, "perf", at::kChar, g.dev, T, g.B, pitch);
  if (bad_count.has_value()) want(*bad_count, "bad_count", at::kInt, g.dev, {1});
  CampxOutputs out{};
  out.scalar_pitch = pitch;
  out.reward = opt_ptr<float>(reward);
  out.discount = opt_ptr<float>(discount);
  out.done = opt_ptr<uint8_t>(step_done);
  out.perf = opt_ptr<int8_t>(perf);
  out.trace = reinterpret_cast<uint8_t*>(trace.data_ptr());
  out.bad_count = opt_ptr<int32_t>(bad_count);
  out.bad_flag = flag_ptr(bad_flag, g.dev);
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(g.dev);
  check_ok(campx_update_launch(g.spec_host, g.spec_dev, g.state,
                               reinterpret_cast<const int8_t*>(actions.data_ptr()), out, g.B,
                               (int32_t)T, reset_first ? 1 : 0,
                               c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()),
           "campx_update_launch");
}

void render(const Tensor& spec_host, const Tensor& spec_dev, const Tensor& trace, Tensor& obs,
            const OptTensor& board) {
  const CampxSpec* hs = host_spec(spec_host);
  TORCH_CHECK(trace.device().is_cuda() && trace.dim() == 3, "campx::render: trace must be a HIP uint8 [K, T, B] tensor");
  const c10::Device dev = trace.device();
  const int64_t K = hs->n_dyn, T = trace.size(1), B = trace.size(2);
  int64_t pitch = 0;
  want_trace(trace, dev, K, T, B, pitch);
  TORCH_CHECK(spec_dev.device() == dev && spec_dev.scalar_type() == at::kByte &&
                  spec_dev.is_contiguous() && spec_dev.numel() == (int64_t)sizeof(CampxSpec),
              "campx: spec_dev must be the CampxSpec blob as a uint8 tensor on ", dev);
  CampxOutputs out{};
  out.scalar_pitch = pitch;
  out.obs_format = obs_format_of(obs);
  want(obs, "obs", obs.scalar_type(), dev, {T, B, hs->n_layers, hs->rows, hs->cols});
  out.obs = reinterpret_cast<int8_t*>(obs.data_ptr());
  out.obs_t_stride = B * hs->n_layers * hs->rows * hs->cols;
  if (board.has_value()) {
    want(*board, "board", at::kChar, dev, {T, B, hs->rows, hs->cols});
    out.board = opt_ptr<int8_t>(board);
    out.board_t_stride = B * hs->rows * hs->cols;
  }
  out.trace = reinterpret_cast<uint8_t*>(trace.data_ptr());
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
  check_ok(campx_render_launch(hs, reinterpret_cast<const CampxSpec*>(spec_dev.data_ptr()), out, B,
                               (int32_t)T,
                               c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()),
           "campx_render_launch");
}

// A rollout over TWO streams (fused.py rollout(pipelined=True), in C++ since round 5): the update
// pass on a HIGH-priority side stream, the render on the caller's stream behind it - so that the
// update pass of the NEXT call runs under this call's render.  What makes it pay is the host: as
// two op dispatches and five stream / event calls from Python it cost 36-46 us per call, more
// than a middle-sized rollout takes (sokoban B = 16 384: 62 us in order); here it is one dispatch.
// The side stream and the events live in this binding layer, per device (the C library below it
// holds no state); `resync`: work has been issued on the caller's stream since the last
// pipelined call that the update pass must come after (state set up by other calls).
struct PipeStreams {
  // held for the whole body of rollout_pipelined: the op releases the GIL, so two actor threads
  // on one device would otherwise interleave their event records / waits and rehash `readers`
  // under each other (round 5 advice).  One pipelined rollout per device at a time is issued.
  std::mutex busy;
  hipStream_t side = nullptr;
  hipEvent_t updated = nullptr, synced = nullptr;
  std::unordered_map<const void*, hipEvent_t> readers;   // trace buffer -> its last render
};

void hip_ok(hipError_t e, const char* what) {
  TORCH_CHECK(e == hipSuccess, "campx: ", what, " failed: ", hipGetErrorString(e));
}

PipeStreams& pipe_streams(int device) {
  static std::mutex lock;
  static std::map<int, PipeStreams> all;
  std::lock_guard<std::mutex> hold(lock);
  PipeStreams& p = all[device];
  if (!p.side) {
    int least = 0, greatest = 0;
    hip_ok(hipDeviceGetStreamPriorityRange(&least, &greatest), "hipDeviceGetStreamPriorityRange");
    hip_ok(hipStreamCreateWithPriority(&p.side, hipStreamNonBlocking, greatest), "hipStreamCreateWithPriority");
    hip_ok(hipEventCreateWithFlags(&p.updated, hipEventDisableTiming), "hipEventCreateWithFlags");
    hip_ok(hipEventCreateWithFlags(&p.synced, hipEventDisableTiming), "hipEventCreateWithFlags");
  }
  return p;
}

void rollout_pipelined(const Tensor& spec_host, const Tensor& spec_dev, Tensor& pos, Tensor& done,
                       const OptTensor& ret, const OptTensor& pair_table, const Tensor& actions,
                       Tensor& obs, const OptTensor& board, const OptTensor& reward,
                       const OptTensor& discount, const OptTensor& step_done, const OptTensor& perf,
                       Tensor& trace, const OptTensor& bad_count, const OptTensor& bad_flag,
                       bool reset_first, bool resync) {
  const Game g = unpack_game(spec_host, spec_dev, pos, done, ret, pair_table);
  TORCH_CHECK(actions.dim() == 2, "campx::rollout_pipelined: actions must be int8 [T, B]");
  const int64_t T = actions.size(0);
  TORCH_CHECK(T >= 1 && T <= 65535, "campx::rollout_pipelined: 1 to 65535 frames");
  want(actions, "actions", at::kChar, g.dev, {T, g.B});
  int64_t pitch = 0;
  want_trace(trace, g.dev, g.K, T, g.B, pitch);
  if (reward.has_value()) want_rows(*reward, "reward", at::kFloat, g.dev, T, g.B, pitch);
  if (discount.has_value()) want_rows(*discount, "discount", at::kFloat, g.dev, T, g.B, pitch);
  if (step_done.has_value()) want_rows(*step_done, "step_done", at::kByte, g.dev, T, g.B, pitch);
  if (perf.has_value()) want_rows(*perf, "perf", at::kChar, g.dev, T, g.B, pitch);
  if (bad_count.has_value()) want(*bad_count, "bad_count", at::kInt, g.dev, {1});
  CampxOutputs upd{};
  upd.scalar_pitch = pitch;
  upd.reward = opt_ptr<float>(reward);
  upd.discount = opt_ptr<float>(discount);
  upd.done = opt_ptr<uint8_t>(step_done);
  upd.perf = opt_ptr<int8_t>(perf);
  upd.trace = reinterpret_cast<uint8_t*>(trace.data_ptr());
  upd.bad_count = opt_ptr<int32_t>(bad_count);
  upd.bad_flag = flag_ptr(bad_flag, g.dev);
  CampxOutputs ren{};
  ren.scalar_pitch = pitch;
  ren.obs_format = obs_format_of(obs);
  want(obs, "obs", obs.scalar_type(), g.dev, {T, g.B, g.L, g.H, g.W});
  ren.obs = reinterpret_cast<int8_t*>(obs.data_ptr());
  ren.obs_t_stride = g.B * g.L * g.H * g.W;
  if (board.has_value()) {
    want(*board, "board", at::kChar, g.dev, {T, g.B, g.H, g.W});
    ren.board = opt_ptr<int8_t>(board);
    ren.board_t_stride = g.B * g.H * g.W;
  }
  ren.trace = upd.trace;
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(g.dev);
  hipStream_t main = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
  PipeStreams& p = pipe_streams(g.dev.index());
  std::lock_guard<std::mutex> one_at_a_time(p.busy);
  if (resync) {       // everything issued on the caller's stream so far, renders included
    hip_ok(hipEventRecord(p.synced, main), "hipEventRecord");
    hip_ok(hipStreamWaitEvent(p.side, p.synced, 0), "hipStreamWaitEvent");
    // (events of trace buffers last rendered before this point are covered by it)
  }
  // the side stream runs ahead of the caller's without bound; what it may not do is overwrite a
  // trace buffer whose last render is still running
  if (p.readers.size() >= 64 && !p.readers.count(upd.trace)) {
    // (a caller that allocates new buffers for every rollout: the events of buffers long gone are
    // let go - behind one wait for everything the caller's stream holds, which covers them all)
    hip_ok(hipEventRecord(p.synced, main), "hipEventRecord");
    hip_ok(hipStreamWaitEvent(p.side, p.synced, 0), "hipStreamWaitEvent");
    for (auto& kept : p.readers) (void)hipEventDestroy(kept.second);
    p.readers.clear();
  }
  hipEvent_t reader = p.readers[upd.trace];      // (by value: the map may rehash)
  if (reader && !resync) hip_ok(hipStreamWaitEvent(p.side, reader, 0), "hipStreamWaitEvent");
  check_ok(campx_update_launch(g.spec_host, g.spec_dev, g.state,
                               reinterpret_cast<const int8_t*>(actions.data_ptr()), upd, g.B, (int32_t)T,
                               reset_first ? 1 : 0, p.side),
           "campx_update_launch");
  hip_ok(hipEventRecord(p.updated, p.side), "hipEventRecord");
  hip_ok(hipStreamWaitEvent(main, p.updated, 0), "hipStreamWaitEvent");
  check_ok(campx_render_launch(g.spec_host, g.spec_dev, ren, g.B, (int32_t)T, main), "campx_render_launch");
  if (!reader) {
    hip_ok(hipEventCreateWithFlags(&reader, hipEventDisableTiming), "hipEventCreateWithFlags");
    p.readers[upd.trace] = reader;
  }
  hip_ok(hipEventRecord(reader, main), "hipEventRecord");
}

// The update pass of one rollout and the render pass of the one before it as ONE call
// (campx_update_render_launch: a single launch where the game and the shapes allow it).
void update_render(const Tensor& spec_host, const Tensor& spec_dev, Tensor& pos, Tensor& done,
                   const OptTensor& ret, const OptTensor& pair_table, const Tensor& actions,
                   const OptTensor& reward, const OptTensor& discount, const OptTensor& step_done,
                   const OptTensor& perf, Tensor& trace, const OptTensor& bad_count,
                   const OptTensor& bad_flag, bool reset_first, const Tensor& prev_trace,
                   Tensor& prev_obs) {
  const Game g = unpack_game(spec_host, spec_dev, pos, done, ret, pair_table);
  TORCH_CHECK(actions.dim() == 2, "campx::update_render: actions must be int8 [T, B]");
  const int64_t T = actions.size(0);
  TORCH_CHECK(T >= 1 && T <= 65535, "campx::update_render: 1 to 65535 frames");
  want(actions, "actions", at::kChar, g.dev, {T, g.B});
  int64_t pitch = 0, prev_pitch = 0;
  want_trace(trace, g.dev, g.K, T, g.B, pitch);
  want_trace(prev_trace, g.dev, g.K, T, g.B, prev_pitch);
  TORCH_CHECK(prev_trace.data_ptr() != trace.data_ptr(),
              "campx::update_render: the two rollouts need a trace buffer each");
  if (reward.has_value()) want_rows(*reward, "reward", at::kFloat, g.dev, T, g.B, pitch);
  if (discount.has_value()) want_rows(*discount, "discount", at::kFloat, g.dev, T, g.B, pitch);
  if (step_done.has_value()) want_rows(*step_done, "step_done", at::kByte, g.dev, T, g.B, pitch);
  if (perf.has_value()) want_rows(*perf, "perf", at::kChar, g.dev, T, g.B, pitch);
  if (bad_count.has_value()) want(*bad_count, "bad_count", at::kInt, g.dev, {1});
  CampxOutputs out{};
  out.scalar_pitch = pitch;
  out.reward = opt_ptr<float>(reward);
  out.discount = opt_ptr<float>(discount);
  out.done = opt_ptr<uint8_t>(step_done);
  out.perf = opt_ptr<int8_t>(perf);
  out.trace = reinterpret_cast<uint8_t*>(trace.data_ptr());
  out.bad_count = opt_ptr<int32_t>(bad_count);
  out.bad_flag = flag_ptr(bad_flag, g.dev);
  CampxOutputs prev{};
  prev.scalar_pitch = prev_pitch;
  prev.obs_format = obs_format_of(prev_obs);
  want(prev_obs, "prev_obs", prev_obs.scalar_type(), g.dev, {T, g.B, g.L, g.H, g.W});
  prev.obs = reinterpret_cast<int8_t*>(prev_obs.data_ptr());
  prev.obs_t_stride = g.B * g.L * g.H * g.W;
  prev.trace = reinterpret_cast<uint8_t*>(prev_trace.data_ptr());
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(g.dev);
  check_ok(campx_update_render_launch(g.spec_host, g.spec_dev, g.state,
                                      reinterpret_cast<const int8_t*>(actions.data_ptr()), out, prev,
                                      g.B, (int32_t)T, reset_first ? 1 : 0,
                                      c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()),
           "campx_update_render_launch");
}

// One Engine.play() frame: actions [B], per-frame outputs [B].
void step(const Tensor& spec_host, const Tensor& spec_dev, Tensor& pos, Tensor& done,
          const OptTensor& ret, const OptTensor& pair_table, const Tensor& actions, Tensor& obs,
          const OptTensor& board, const OptTensor& reward, const OptTensor& discount,
          const OptTensor& step_done, const OptTensor& perf, const OptTensor& bad_count,
          const OptTensor& bad_flag) {
  const Game g = unpack_game(spec_host, spec_dev, pos, done, ret, pair_table);
  want(actions, "actions", at::kChar, g.dev, {g.B});
  const int32_t format = obs_format_of(obs);   // int8, or f16 / bf16 for a policy network
  want(obs, "obs", obs.scalar_type(), g.dev, {g.B, g.L, g.H, g.W});
  if (board.has_value()) want(*board, "board", at::kChar, g.dev, {g.B, g.H, g.W});
  if (reward.has_value()) want(*reward, "reward", at::kFloat, g.dev, {g.B});
  if (discount.has_value()) want(*discount, "discount", at::kFloat, g.dev, {g.B});
  if (step_done.has_value()) want(*step_done, "step_done", at::kByte, g.dev, {g.B});
  if (perf.has_value()) want(*perf, "perf", at::kChar, g.dev, {g.B});
  if (bad_count.has_value()) want(*bad_count, "bad_count", at::kInt, g.dev, {1});
  CampxOutputs out{};
  out.obs = reinterpret_cast<int8_t*>(obs.data_ptr());
  out.obs_format = format;
  out.board = opt_ptr<int8_t>(board);
  out.reward = opt_ptr<float>(reward);
  out.discount = opt_ptr<float>(discount);
  out.done = opt_ptr<uint8_t>(step_done);
  out.perf = opt_ptr<int8_t>(perf);
  out.bad_count = opt_ptr<int32_t>(bad_count);
  out.bad_flag = flag_ptr(bad_flag, g.dev);
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(g.dev);
  check_ok(campx_rollout_launch(g.spec_host, g.spec_dev, g.state,
                                reinterpret_cast<const int8_t*>(actions.data_ptr()), out, g.B, 1, 0,
                                c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()),
           "campx_rollout_launch");
}

// Shape tier (Hello World): reset (actions = None, emit_first), one frame (actions [B],
// obs [B, L, H, W]) or T frames (actions [T, B], obs [T, B, L, H, W]) through one op.
void shape_rollout(const Tensor& spec_host, const Tensor& spec_dev, Tensor& pos, Tensor& done,
                   const OptTensor& ret, const OptTensor& backdrop_state, const OptTensor& actions,
                   Tensor& obs, const OptTensor& board, const OptTensor& reward,
                   const OptTensor& discount, const OptTensor& step_done,
                   const OptTensor& bad_count, const OptTensor& bad_flag, bool reset_first,
                   bool emit_first, const OptTensor& trace, const OptTensor& tables) {
  TORCH_CHECK(spec_host.device().is_cpu() && spec_host.scalar_type() == at::kByte &&
                  spec_host.is_contiguous() && spec_host.numel() == (int64_t)sizeof(CampxShapeSpec),
              "campx: spec_host must be the CampxShapeSpec blob as a CPU uint8 tensor");
  const CampxShapeSpec* hs = reinterpret_cast<const CampxShapeSpec*>(spec_host.data_ptr());
  TORCH_CHECK(pos.device().is_cuda() && pos.dim() == 2,
              "campx::shape_rollout: state must be on a HIP device (no CPU implementation)");
  const c10::Device dev = pos.device();
  const int64_t B = pos.size(1), N = hs->n_things, L = hs->n_layers, H = hs->rows, W = hs->cols;
  want(pos, "pos", at::kChar, dev, {2 * N, B});
  want(done, "done", at::kByte, dev, {B});
  if (ret.has_value()) want(*ret, "ret", at::kFloat, dev, {B});
  if (backdrop_state.has_value()) want(*backdrop_state, "backdrop_state", at::kChar, dev, {B, H * W});
  TORCH_CHECK(spec_dev.device() == dev && spec_dev.scalar_type() == at::kByte &&
                  spec_dev.is_contiguous() && spec_dev.numel() == (int64_t)sizeof(CampxShapeSpec),
              "campx: spec_dev must be the CampxShapeSpec blob as a uint8 tensor on ", dev);
  int64_t T = 0;
  bool frames = false;  // outputs carry a leading frame axis
  if (actions.has_value()) {
    frames = actions->dim() == 2;
    T = frames ? actions->size(0) : 1;
    if (frames) want(*actions, "actions", at::kChar, dev, {T, B});
    else want(*actions, "actions", at::kChar, dev, {B});
  }
  CampxOutputs out{};
  auto shape = [&](std::initializer_list<int64_t> tail) {
    std::vector<int64_t> v;
    if (frames) v.push_back(T);
    v.push_back(B);
    v.insert(v.end(), tail);
    return v;
  };
  const bool keep = frames && obs.dim() == 5;
  out.obs_format = obs_format_of(obs);       // int8, or f16 / bf16 for a policy network
  if (keep || !frames) want(obs, "obs", obs.scalar_type(), dev, shape({L, H, W}));
  else want(obs, "obs", obs.scalar_type(), dev, {B, L, H, W});
  out.obs = reinterpret_cast<int8_t*>(obs.data_ptr());
  out.obs_t_stride = keep ? B * L * H * W : 0;
  if (board.has_value()) {
    const bool bkeep = frames && board->dim() == 4;
    if (bkeep) want(*board, "board", at::kChar, dev, {T, B, H, W});
    else want(*board, "board", at::kChar, dev, {B, H, W});
    out.board = opt_ptr<int8_t>(board);
    out.board_t_stride = bkeep ? B * H * W : 0;
  }
  if (reward.has_value()) want(*reward, "reward", at::kFloat, dev, shape({}));
  if (discount.has_value()) want(*discount, "discount", at::kFloat, dev, shape({}));
  if (step_done.has_value()) want(*step_done, "step_done", at::kByte, dev, shape({}));
  if (bad_count.has_value()) want(*bad_count, "bad_count", at::kInt, dev, {1});
  out.reward = opt_ptr<float>(reward);
  out.discount = opt_ptr<float>(discount);
  out.done = opt_ptr<uint8_t>(step_done);
  out.bad_count = opt_ptr<int32_t>(bad_count);
  out.bad_flag = flag_ptr(bad_flag, dev);
  const void* tables_dev = nullptr;
  if (trace.has_value() && tables.has_value() && frames) {
    // the frame-major path: its scratch (campx_shape_scratch_bytes) and the game's row tables
    const int64_t need = campx_shape_scratch_bytes(hs, B, (int32_t)T);
    TORCH_CHECK(trace->device() == dev && trace->scalar_type() == at::kLong && trace->is_contiguous() &&
                    need > 0 && trace->numel() * 8 >= need,
                "campx::shape_rollout: trace must be a contiguous int64 tensor of at least ", need,
                " bytes on ", dev);
    TORCH_CHECK(tables->device() == dev && tables->scalar_type() == at::kLong && tables->is_contiguous() &&
                    tables->numel() * 8 >= campx_shape_tables_bytes(hs),
                "campx::shape_rollout: tables must hold campx_shape_tables_build()'s blob on ", dev);
    out.trace = reinterpret_cast<uint8_t*>(trace->data_ptr());
    tables_dev = tables->data_ptr();
  }
  CampxState state{reinterpret_cast<int8_t*>(pos.data_ptr()), reinterpret_cast<uint8_t*>(done.data_ptr()),
                   opt_ptr<float>(ret), nullptr};
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
  check_ok(campx_shape_rollout_launch(
               hs, reinterpret_cast<const CampxShapeSpec*>(spec_dev.data_ptr()), tables_dev, state,
               opt_ptr<int8_t>(backdrop_state), opt_ptr<int8_t>(actions), out, B, (int32_t)T,
               reset_first ? 1 : 0, emit_first ? 1 : 0,
               c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()),
           "campx_shape_rollout_launch");
}

// Wide tier (state-table games: boards above 128 cells, include/campx_hip.h): reset
// (actions = None), one frame (actions [B], outputs [B...]) or T frames (actions [T, B])
// through one op.  `tables`: the device blob campx_wide_tables_build() filled.  `state`:
// int32 [B], the environments' state indices.  `trace`: int16 [K, B] / [K, T, B] (rows may be
// padded like the other per-frame streams).  With T frames, `obs` is [T, B, L, H, W] (every
// frame) or [B, L, H, W] (the last one).
void wide_rollout(const Tensor& spec_host, const Tensor& tables, Tensor& state, Tensor& done,
                  const OptTensor& ret, const OptTensor& actions, Tensor& obs, const OptTensor& board,
                  const OptTensor& reward, const OptTensor& discount, const OptTensor& step_done,
                  const OptTensor& perf, Tensor& trace, const OptTensor& bad_count,
                  const OptTensor& bad_flag, bool reset_first) {
  TORCH_CHECK(spec_host.device().is_cpu() && spec_host.scalar_type() == at::kByte &&
                  spec_host.is_contiguous() && spec_host.numel() == (int64_t)sizeof(CampxWideSpec),
              "campx: spec_host must be the CampxWideSpec blob as a CPU uint8 tensor");
  const CampxWideSpec* hs = reinterpret_cast<const CampxWideSpec*>(spec_host.data_ptr());
  TORCH_CHECK(state.device().is_cuda() && state.dim() == 1,
              "campx::wide_rollout: state must be on a HIP device (no CPU implementation)");
  const c10::Device dev = state.device();
  // (planes of the trace: the things, plus the scenery's variant when it has several - or the mask
  // of its pieces that show)
  const int64_t B = state.size(0), K = hs->n_dyn + ((hs->n_variants > 1 || hs->n_pieces > 0) ? 1 : 0), L = hs->n_layers, H = hs->rows,
                W = hs->cols;
  want(state, "state", at::kInt, dev, {B});
  want(done, "done", at::kByte, dev, {B});
  if (ret.has_value()) want(*ret, "ret", at::kFloat, dev, {B});
  TORCH_CHECK(tables.device() == dev && tables.scalar_type() == at::kByte && tables.is_contiguous() &&
                  tables.numel() == campx_wide_tables_bytes(hs),
              "campx: tables must be the campx_wide_tables_build() blob as a uint8 tensor on ", dev);
  int64_t T = 0;
  bool frames = false;
  if (actions.has_value()) {
    frames = actions->dim() == 2;
    T = frames ? actions->size(0) : 1;
    if (frames) want(*actions, "actions", at::kChar, dev, {T, B});
    else want(*actions, "actions", at::kChar, dev, {B});
    TORCH_CHECK(T >= 1 && T <= 0x7fffffff, "campx::wide_rollout: bad frame count");
  }
  CampxOutputs out{};
  int64_t pitch = 0;
  // the trace first: with one frame only its planes tell the row pitch
  TORCH_CHECK(trace.device() == dev && trace.scalar_type() == at::kShort,
              "campx: trace must be an int16 tensor on ", dev);
  if (frames) {
    TORCH_CHECK(trace.dim() == 3 && trace.size(0) == K && trace.size(1) == T && trace.size(2) == B &&
                    (B == 1 || trace.stride(2) == 1),
                "campx: trace must be int16 [", K, ", ", T, ", ", B, "], contiguous within a row");
    if (T > 1) pitch = trace.stride(1);
    else if (K > 1) pitch = trace.stride(0);
    TORCH_CHECK(pitch == 0 || (pitch >= B && (K == 1 || trace.stride(0) == T * pitch)),
                "campx: trace rows must be >= B apart and its planes T * pitch apart");
  } else {
    TORCH_CHECK(trace.dim() == 2 && trace.size(0) == K && trace.size(1) == B &&
                    (B == 1 || trace.stride(1) == 1),
                "campx: trace must be int16 [", K, ", ", B, "], contiguous within a row");
    if (K > 1) pitch = trace.stride(0);
    TORCH_CHECK(pitch == 0 || pitch >= B, "campx: trace rows must be >= B apart");
  }
  auto stream_of = [&](const OptTensor& t, const char* name, at::ScalarType dtype) {
    if (!t.has_value()) return;
    if (frames) want_rows(*t, name, dtype, dev, T, B, pitch);
    else want(*t, name, dtype, dev, {B});
  };
  stream_of(reward, "reward", at::kFloat);
  stream_of(discount, "discount", at::kFloat);
  stream_of(step_done, "step_done", at::kByte);
  stream_of(perf, "perf", at::kChar);
  out.scalar_pitch = pitch;
  out.obs_format = obs_format_of(obs);
  const bool keep = frames && obs.dim() == 5;
  if (keep) want(obs, "obs", obs.scalar_type(), dev, {T, B, L, H, W});
  else want(obs, "obs", obs.scalar_type(), dev, {B, L, H, W});
  out.obs = reinterpret_cast<int8_t*>(obs.data_ptr());
  out.obs_t_stride = (keep || !frames) ? B * L * H * W : 0;
  if (board.has_value()) {
    const bool bkeep = frames && board->dim() == 4;
    TORCH_CHECK(bkeep == keep, "campx::wide_rollout: obs and board must both keep every frame or both the last");
    if (bkeep) want(*board, "board", at::kChar, dev, {T, B, H, W});
    else want(*board, "board", at::kChar, dev, {B, H, W});
    out.board = opt_ptr<int8_t>(board);
    out.board_t_stride = (bkeep || !frames) ? B * H * W : 0;
  }
  if (bad_count.has_value()) want(*bad_count, "bad_count", at::kInt, dev, {1});
  out.reward = opt_ptr<float>(reward);
  out.discount = opt_ptr<float>(discount);
  out.done = opt_ptr<uint8_t>(step_done);
  out.perf = opt_ptr<int8_t>(perf);
  out.trace = reinterpret_cast<uint8_t*>(trace.data_ptr());
  out.bad_count = opt_ptr<int32_t>(bad_count);
  out.bad_flag = flag_ptr(bad_flag, dev);
  CampxState st{reinterpret_cast<int8_t*>(state.data_ptr()), reinterpret_cast<uint8_t*>(done.data_ptr()),
                opt_ptr<float>(ret), nullptr};
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
  void* stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
  if (!actions.has_value())
    check_ok(campx_wide_reset_launch(hs, tables.data_ptr(), st, out, B, stream),
             "campx_wide_reset_launch");
  else
    check_ok(campx_wide_rollout_launch(hs, tables.data_ptr(), st,
                                       reinterpret_cast<const int8_t*>(actions->data_ptr()), out, B,
                                       (int32_t)T, reset_first ? 1 : 0, stream),
             "campx_wide_rollout_launch");
}

void onehot_to_ids(const Tensor& onehot, Tensor& ids, Tensor& bad_count) {
  TORCH_CHECK(onehot.device().is_cuda(), "campx::onehot_to_ids: HIP tensors only");
  const c10::Device dev = onehot.device();
  const int64_t n = ids.numel();
  TORCH_CHECK(onehot.scalar_type() == at::kFloat && onehot.is_contiguous() &&
                  onehot.numel() == n * CAMPX_N_ACTIONS,
              "campx::onehot_to_ids: onehot must be contiguous float32 [..., 5]");
  TORCH_CHECK(ids.device() == dev && ids.scalar_type() == at::kChar && ids.is_contiguous(),
              "campx::onehot_to_ids: ids must be contiguous int8 on ", dev);
  want(bad_count, "bad_count", at::kInt, dev, {1});
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
  check_ok(campx_onehot_to_ids_launch(reinterpret_cast<const float*>(onehot.data_ptr()),
                                      reinterpret_cast<int8_t*>(ids.data_ptr()), n,
                                      reinterpret_cast<int32_t*>(bad_count.data_ptr()),
                                      c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()),
           "campx_onehot_to_ids_launch");
}

void check_actions(const Tensor& actions, Tensor& bad_count) {
  TORCH_CHECK(actions.device().is_cuda(), "campx::check_actions: HIP tensors only");
  const c10::Device dev = actions.device();
  TORCH_CHECK(actions.scalar_type() == at::kChar && actions.is_contiguous(),
              "campx::check_actions: actions must be contiguous int8");
  want(bad_count, "bad_count", at::kInt, dev, {1});
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
  check_ok(campx_check_actions_launch(reinterpret_cast<const int8_t*>(actions.data_ptr()),
                                      actions.numel(),
                                      reinterpret_cast<int32_t*>(bad_count.data_ptr()),
                                      c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()),
           "campx_check_actions_launch");
}

// Meta / fake-tensor implementations: the ops return nothing and write in place, so
// there is nothing to infer.
void reset_meta(const Tensor&, const Tensor&, Tensor&, Tensor&, const OptTensor&, const OptTensor&,
                Tensor&, const OptTensor&) {}
void rollout_meta(const Tensor&, const Tensor&, Tensor&, Tensor&, const OptTensor&, const OptTensor&,
                  const Tensor&, Tensor&, const OptTensor&, const OptTensor&, const OptTensor&,
                  const OptTensor&, const OptTensor&, const OptTensor&, const OptTensor&,
                  const OptTensor&, bool, const OptTensor&, const OptTensor&, const OptTensor&) {}
void rollout_pipelined_meta(const Tensor&, const Tensor&, Tensor&, Tensor&, const OptTensor&, const OptTensor&,
                            const Tensor&, Tensor&, const OptTensor&, const OptTensor&, const OptTensor&,
                            const OptTensor&, const OptTensor&, Tensor&, const OptTensor&, const OptTensor&,
                            bool, bool) {}
void step_meta(const Tensor&, const Tensor&, Tensor&, Tensor&, const OptTensor&, const OptTensor&,
               const Tensor&, Tensor&, const OptTensor&, const OptTensor&, const OptTensor&,
               const OptTensor&, const OptTensor&, const OptTensor&, const OptTensor&) {}
void update_meta(const Tensor&, const Tensor&, Tensor&, Tensor&, const OptTensor&, const OptTensor&,
                 const Tensor&, const OptTensor&, const OptTensor&, const OptTensor&,
                 const OptTensor&, Tensor&, const OptTensor&, const OptTensor&, bool) {}
void render_meta(const Tensor&, const Tensor&, const Tensor&, Tensor&, const OptTensor&) {}
void update_render_meta(const Tensor&, const Tensor&, Tensor&, Tensor&, const OptTensor&,
                        const OptTensor&, const Tensor&, const OptTensor&, const OptTensor&,
                        const OptTensor&, const OptTensor&, Tensor&, const OptTensor&,
                        const OptTensor&, bool, const Tensor&, Tensor&) {}
void shape_rollout_meta(const Tensor&, const Tensor&, Tensor&, Tensor&, const OptTensor&,
                        const OptTensor&, const OptTensor&, Tensor&, const OptTensor&,
                        const OptTensor&, const OptTensor&, const OptTensor&, const OptTensor&,
                        const OptTensor&, bool, bool, const OptTensor&, const OptTensor&) {}
void wide_rollout_meta(const Tensor&, const Tensor&, Tensor&, Tensor&, const OptTensor&,
                       const OptTensor&, Tensor&, const OptTensor&, const OptTensor&,
                       const OptTensor&, const OptTensor&, const OptTensor&, Tensor&,
                       const OptTensor&, const OptTensor&, bool) {}
void onehot_to_ids_meta(const Tensor&, Tensor&, Tensor&) {}
void check_actions_meta(const Tensor&, Tensor&) {}

// ADInplaceOrView: run the op, then mark every argument the schema declares written
// (`Tensor(a!)`) as modified in place.
void run_then_bump_versions(const c10::OperatorHandle& op, c10::DispatchKeySet keys,
                            torch::jit::Stack* stack) {
  const auto& arguments = op.schema().arguments();
  const size_t n = arguments.size();
  at::Tensor written[16];
  size_t n_written = 0;
  for (size_t i = 0; i < n; ++i) {
    const c10::AliasInfo* alias = arguments[i].alias_info();
    if (!alias || !alias->isWrite()) continue;
    const c10::IValue& v = torch::jit::peek(*stack, i, n);
    if (v.isTensor() && v.toTensor().defined() && n_written < 16) written[n_written++] = v.toTensor();
  }
  {
    c10::impl::ExcludeDispatchKeyGuard below(c10::autograd_dispatch_keyset_with_ADInplaceOrView);
    op.redispatchBoxed(keys & c10::after_ADInplaceOrView_keyset, stack);
  }
  for (size_t i = 0; i < n_written; ++i)
    if (!written[i].is_inference()) written[i].unsafeGetTensorImpl()->bump_version();
}

}  // namespace

TORCH_LIBRARY(campx, m) {
  m.def(
      "reset(Tensor spec_host, Tensor spec_dev, Tensor(a!) pos, Tensor(b!) done, Tensor(c!)? ret, "
      "Tensor? pair_table, Tensor(d!) obs, Tensor(e!)? board) -> ()");
  m.def(
      "step(Tensor spec_host, Tensor spec_dev, Tensor(a!) pos, Tensor(b!) done, Tensor(c!)? ret, "
      "Tensor? pair_table, Tensor actions, Tensor(d!) obs, Tensor(e!)? board, Tensor(f!)? reward, "
      "Tensor(g!)? discount, Tensor(h!)? step_done, Tensor(i!)? perf, Tensor(j!)? bad_count, "
      "Tensor(k!)? bad_flag) -> ()");
  m.def(
      "rollout(Tensor spec_host, Tensor spec_dev, Tensor(a!) pos, Tensor(b!) done, Tensor(c!)? ret, "
      "Tensor? pair_table, Tensor actions, Tensor(d!) obs, Tensor(e!)? board, Tensor(f!)? reward, "
      "Tensor(g!)? discount, Tensor(h!)? step_done, Tensor(i!)? perf, Tensor(j!)? trace, "
      "Tensor(k!)? bad_count, Tensor(l!)? bad_flag, bool reset_first, Tensor(m!)? scratch=None, "
      "Tensor(n!)? scratch_state=None, Tensor(o!)? error_flag=None) -> ()");
  m.def(
      "update(Tensor spec_host, Tensor spec_dev, Tensor(a!) pos, Tensor(b!) done, Tensor(c!)? ret, "
      "Tensor? pair_table, Tensor actions, Tensor(d!)? reward, Tensor(e!)? discount, "
      "Tensor(f!)? step_done, Tensor(g!)? perf, Tensor(h!) trace, Tensor(i!)? bad_count, "
      "Tensor(j!)? bad_flag, bool reset_first) -> ()");
  m.def(
      "render(Tensor spec_host, Tensor spec_dev, Tensor trace, Tensor(a!) obs, Tensor(b!)? board) "
      "-> ()");
  m.def(
      "rollout_pipelined(Tensor spec_host, Tensor spec_dev, Tensor(a!) pos, Tensor(b!) done, "
      "Tensor(c!)? ret, Tensor? pair_table, Tensor actions, Tensor(d!) obs, Tensor(e!)? board, "
      "Tensor(f!)? reward, Tensor(g!)? discount, Tensor(h!)? step_done, Tensor(i!)? perf, "
      "Tensor(j!) trace, Tensor(k!)? bad_count, Tensor(l!)? bad_flag, bool reset_first, bool resync) -> ()");
  m.def(
      "update_render(Tensor spec_host, Tensor spec_dev, Tensor(a!) pos, Tensor(b!) done, "
      "Tensor(c!)? ret, Tensor? pair_table, Tensor actions, Tensor(d!)? reward, Tensor(e!)? discount, "
      "Tensor(f!)? step_done, Tensor(g!)? perf, Tensor(h!) trace, Tensor(i!)? bad_count, "
      "Tensor(j!)? bad_flag, bool reset_first, Tensor prev_trace, Tensor(k!) prev_obs) -> ()");
  m.def(
      "shape_rollout(Tensor spec_host, Tensor spec_dev, Tensor(a!) pos, Tensor(b!) done, "
      "Tensor(c!)? ret, Tensor(d!)? backdrop_state, Tensor? actions, Tensor(e!) obs, "
      "Tensor(f!)? board, Tensor(g!)? reward, Tensor(h!)? discount, Tensor(i!)? step_done, "
      "Tensor(j!)? bad_count, Tensor(k!)? bad_flag, bool reset_first, bool emit_first, "
      "Tensor(l!)? trace=None, Tensor? tables=None) -> ()");
  m.def(
      "wide_rollout(Tensor spec_host, Tensor tables, Tensor(a!) state, Tensor(b!) done, "
      "Tensor(c!)? ret, Tensor? actions, Tensor(d!) obs, Tensor(e!)? board, Tensor(f!)? reward, "
      "Tensor(g!)? discount, Tensor(h!)? step_done, Tensor(i!)? perf, Tensor(j!) trace, "
      "Tensor(k!)? bad_count, Tensor(l!)? bad_flag, bool reset_first) -> ()");
  m.def("onehot_to_ids(Tensor onehot, Tensor(a!) ids, Tensor(b!) bad_count) -> ()");
  m.def("check_actions(Tensor actions, Tensor(a!) bad_count) -> ()");
}

TORCH_LIBRARY_IMPL(campx, CUDA, m) {
  m.impl("reset", &reset);
  m.impl("step", &step);
  m.impl("rollout", &rollout);
  m.impl("update", &update);
  m.impl("render", &render);
  m.impl("rollout_pipelined", &rollout_pipelined);
  m.impl("update_render", &update_render);
  m.impl("shape_rollout", &shape_rollout);
  m.impl("wide_rollout", &wide_rollout);
  m.impl("onehot_to_ids", &onehot_to_ids);
  m.impl("check_actions", &check_actions);
}

TORCH_LIBRARY_IMPL(campx, ADInplaceOrView, m) {
  for (const char* name : {"reset", "step", "rollout", "update", "render", "rollout_pipelined", "update_render", "shape_rollout",
                           "wide_rollout", "onehot_to_ids", "check_actions"})
    m.impl(name, torch::CppFunction::makeFromBoxedFunction<&run_then_bump_versions>());
}

TORCH_LIBRARY_IMPL(campx, Meta, m) {
  m.impl("reset", &reset_meta);
  m.impl("step", &step_meta);
  m.impl("rollout", &rollout_meta);
  m.impl("update", &update_meta);
  m.impl("render", &render_meta);
  m.impl("rollout_pipelined", &rollout_pipelined_meta);
  m.impl("update_render", &update_render_meta);
  m.impl("shape_rollout", &shape_rollout_meta);
  m.impl("wide_rollout", &wide_rollout_meta);
  m.impl("onehot_to_ids", &onehot_to_ids_meta);
  m.impl("check_actions", &check_actions_meta);
}
