// k_render.hip - render_kernel: expands the trace into the observation stream.
#include "campx_common.hip.h"

namespace campx_impl {

// ---------------------------------------------------------------------------
// Split path, second half: expand the trace into the observation stream.
// One-shot blocks, ONE aligned 16-byte store per thread, block (x, t) writing bytes
// [x*4096, (x+1)*4096) of frame t: the dispatcher walks the output linearly.  That
// is the store pattern that reaches the HBM write ceiling on this chip
// (tools/probes: 6.9 TB/s, against 5.4 TB/s for long-lived waves that each stream a
// private tile, whatever the tile size).
//
// A frame is B rows of R bytes (R = L*H*W for the layered board, H*W for the flat
// board).  A row is the scenery's row with at most two bytes changed per moving
// thing, given by its trace entry.  spec->rot_* hold 16 byte-rotations of the
// cyclically continued scenery row, so any 16-byte window of back-to-back rows is
// ONE aligned 16-byte load (L1-resident).
struct RenderParams {
  uint32_t R;                 // row bytes
  uint32_t m, sh1, sh2;       // exact n / R for 32-bit n (Granlund-Montgomery)
  uint32_t slab_bytes;        // B * R
  uint32_t shift_base, shift_slab;  // (address of frame 0) and slab_bytes modulo the window span
  int32_t n_dyn, is_board, cells;
  int64_t B;
  int64_t pitch;              // rows of the trace from one frame to the next (B unless padded)
  int32_t dyn_char[CAMPX_WIDE_MAX_DYN];
  int32_t dyn_off[CAMPX_WIDE_MAX_DYN];   // byte offset of moving thing d's layer inside a row
  const int8_t* rot;          // device: the 16 rotations of the scenery row (layered or flat board)
  const uint8_t* top_layer;   // device: scenery layer per cell (one-byte trace only)
  int64_t rot_stride;         // kVar: bytes from one variant's rotations to the next's
  int32_t n_variants;         // kVar: how many there are
  uint32_t inv_p;             // run-time thing count (five to eight things): ceil(2^32 / patches per row)
  const uint32_t* pieces;     // kMask: device, per piece (byte it sets | byte it clears << 16), or (cell | character << 16)
  int32_t piece_shift;        // kMask: log2 of the pieces per row, rounded up to a power of two
};

// The trace entry of one moving thing in one frame.  One byte in the one-cell tier (cell |
// visible << 7; the scenery layer the thing covers is looked up per cell).  Boards above 128
// cells (the wide tier, k_wide.hip) write 16 bits: cell | covered layer << 10 | visible << 15,
// so the render needs no per-cell table at all.
template <bool kWide>
struct TraceFormat {
  using Entry = uint8_t;
  static __device__ __forceinline__ int cell(uint32_t e) { return (int)(e & 0x7fu); }
  static __device__ __forceinline__ bool visible(uint32_t e) { return (e >> 7) != 0; }
};
template <>
struct TraceFormat<true> {
  using Entry = uint16_t;
  static __device__ __forceinline__ int cell(uint32_t e) { return (int)(e & 0x3ffu); }
  static __device__ __forceinline__ bool visible(uint32_t e) { return (e >> 15) != 0; }
  static __device__ __forceinline__ int covered(uint32_t e) { return (int)((e >> 10) & 0xfu); }
};

// Block (x, t) writes bytes [x*4096*kWin, (x+1)*4096*kWin) of frame t; each of its four
// waves owns kWin aligned KiB of it (every store instruction of a wave is one aligned,
// contiguous KiB: tools/probes show -12..-21 % for anything less aligned).
//
// A wave first lays the scenery's bytes for its window into LDS (one aligned 16-byte
// load from the rotated scenery table per lane), then the few lanes that hold a
// patch - (row overlapping the window) x (moving thing) x (set | clear) - write their
// single byte into it, then every lane reads its 16 bytes back and stores them.
// A patch comes from the thing's trace byte (cell, visible): the 1 it paints is at
// dyn_off + cell, the scenery's 1 it hides at scen_off[cell], a per-wave LDS copy of
// spec->static_top_layer[cell] * cells + cell.  Nothing is shared between waves, so
// there is no workgroup barrier.
// A/B knobs of the render kernel's shape: waves per block, KiB windows per wave, and
// whether block indices are remapped so that each XCD (block b runs on XCD b % 8) sweeps
// its own contiguous eighth of a frame instead of every eighth block of it (neighbouring
// windows then share trace lines inside ONE L2).  Measured with rocprofv3, avg of 63
// launches (gpurun_out/r2c), boat race / wall world / sokoban render in us:
//   4 waves, 2 KiB, no remap   176.9 / 2006.7 / 361.8
//   4 waves, 2 KiB, remap      173.5 / 1877.2 / 362.3
//   2 waves, 2 KiB, remap      172.7 / 1873.7 / 363.6   <- default
//   4 waves, 4 KiB, remap      190.3 / 2099.3 / 362.3
//   2 waves, 4 KiB, remap      177.7 / 1984.6 / 356.0
// Again with settled clocks (250 launches after 50 warm-up ones, gpurun_out/t10, remap on):
//   2 waves x 2 KiB 164.8 / 1842 / 338.1 (default)   4 x 2: 165.6 / 1873 / 340.0
//   1 x 2: 166.1 / 1893 / 345.3   2 x 4: 180.2 / 1978 / 360.9   1 x 4: 179.4 / 2008 / 371.4
//   4 x 1: 187.7 / 1994 / 388.7
#ifndef CAMPX_RENDER_WAVES
#define CAMPX_RENDER_WAVES 2
#endif
#ifndef CAMPX_RENDER_WIN
#define CAMPX_RENDER_WIN 2
#endif
#ifndef CAMPX_RENDER_XCD
#define CAMPX_RENDER_XCD 1
#endif
constexpr int kRenderWaves = CAMPX_RENDER_WAVES;

// kOdd: B * R is not a multiple of 16, so frames do not start on a 16-byte boundary and a
// lane's (memory-aligned) 16-byte chunk can straddle two frames.  A chunk belongs to the
// frame it STARTS in and is written whole, with the first bytes of the next frame's first
// row (the trace plane is [T * B] rows: row B of this frame is row 0 of the next); only at
// the two ends of a launch is a chunk written byte by byte - the part after the first
// frame's start, the part before the last frame's end.
// kVar (wide tier, round 6): the scenery is a function of the state - a Backdrop that repaints
// itself over many cells.  Plane `n_dyn` of the trace holds, per (frame, environment), the
// index of the scenery variant; a chunk's scenery bytes come from that variant's rotations, and a
// chunk that runs over the end of an environment's row takes the rest from the NEXT environment's
// variant (the rotations continue a row cyclically with its own start, which is what the next
// row starts with when both show the same variant - and what merge_rows() replaces when not).
// kMask (wide tier, round 6): pieces of the scenery that come and go - the cells of a drape whose
// curtain loses them one by one, the cells a Backdrop repaints.  Plane `n_dyn` of the trace holds,
// per (frame, environment), the 16-bit mask of the pieces that show; the wave patches them onto
// the plain scenery like the things (a byte set in the piece's layer, the scenery's byte under it
// cleared) from ONE trace entry per row however many pieces there are - eight things tracked
// one by one load sixteen slots a row (3.7 TB/s on a 4x9 board against 6.1 this way), a scenery
// in variants waits for the entry before it can fetch its row (5.2): here nothing waits for
// anything but the trace.
template <int K, bool kBoard, bool kNT, int kWin, int kFmt, bool kOdd = false, bool kWide = false,
          int kScen = 0>
__global__ __launch_bounds__(kRenderWaves * kWave) void render_kernel(
    RenderParams rp, const typename TraceFormat<kWide>::Entry* __restrict__ trace,
    int8_t* __restrict__ dst, int64_t n_rows) {
  using Fmt = TraceFormat<kWide>;
  constexpr bool kVar = kScen == 1, kMask = kScen == 2;
  __shared__ __attribute__((aligned(16))) int8_t lds[kRenderWaves * kWin * 1024];
  __shared__ uint16_t scen_off_all[kRenderWaves][kWide ? 2 : CAMPX_MAX_CELLS];
  // kVar: the variant of every row the wave's windows overlap (+ the one after): rows of at least
  // 16 bytes, so at most kWin * 64 + 2 of them
  __shared__ uint16_t row_variant_all[kRenderWaves][kVar ? kWin * 64 + 4 : 2];
  // readfirstlane: the wave index is uniform, and saying so keeps everything derived
  // from it (window offsets, the divisions, base addresses) on the scalar unit
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // this wave's kWin consecutive KiB windows of the frame
  uint32_t bx = blockIdx.x;
#if CAMPX_RENDER_XCD
  bx = (bx & 7u) * (gridDim.x >> 3) + (bx >> 3);     // gridDim.x is a multiple of 8
#endif
  // Windows are aligned in MEMORY, not in the frame: when a frame does not start on a
  // window boundary (B * R not a multiple of the span) they start `shift` bytes before it,
  // so every store of every frame is still one aligned KiB (499 984 environments: 4.2 TB/s
  // with frame-aligned windows).  Offsets are modulo 2^32: the head window's start is
  // "negative", its lanes before the frame fail the one `off < slab_bytes` test below.
  const uint32_t span = 1024u * kWin;
  const uint32_t shift = (rp.shift_base + blockIdx.y * rp.shift_slab) & (span - 1u);
  const uint32_t widx = bx * (uint32_t)kRenderWaves + (uint32_t)wave;
  if ((uint64_t)widx * span >= (uint64_t)rp.slab_bytes + shift) return;
  const uint32_t woff0 = widx * span - shift;
  const uint32_t wlo = widx * span < shift ? 0u : woff0;   // first byte inside the frame
  int8_t* win0 = lds + wave * (kWin * 1024);
  uint16_t* scen_off = scen_off_all[wave];
  const int R = (int)rp.R;
  const int pitch = ((R + 15) & ~15) + 16;
  const int8_t* rot = rp.rot;
  // patches per row; the K = 8 instantiation (wide tier, five to eight things) reads the
  // count from its arguments: its surplus planes of the trace do not exist
  constexpr bool kRuntimeK = K > CAMPX_MAX_DYN;
  const int P = kRuntimeK ? (kBoard ? rp.n_dyn : 2 * rp.n_dyn) : (kBoard ? K : 2 * K);
  // slot -> (row, patch): a division by P.  By a count only known at run time it is a few dozen
  // instructions three times over (the run-time-K instantiation measured 8 % below the templated
  // ones): one multiply by ceil(2^32 / P), exact while slot * P < 2^32
  auto row_of_slot = [&](int sidx) -> int {
    if constexpr (kRuntimeK) return (int)__umulhi((uint32_t)sidx, rp.inv_p);
    else return sidx / P;
  };
  const typename Fmt::Entry* frame_trace = trace + (int64_t)blockIdx.y * rp.pitch;
  // (kOdd: a chunk that straddles two frames asks for "row B" = row 0 of the next frame)
  auto trace_row = [&](uint32_t row) -> int64_t {
    return (int64_t)row < rp.B ? (int64_t)row : (int64_t)row - rp.B + rp.pitch;
  };

  // ---- patches: (row overlapping the windows) x (moving thing) x (set | clear).
  // Their trace bytes come from HBM / L2: issue those loads first.
  const uint32_t whi = __umulhi(rp.m, wlo);
  const uint32_t first_row = (((wlo - whi) >> rp.sh1) + whi) >> rp.sh2;
  const bool last_frame = blockIdx.y == gridDim.y - 1u;
  constexpr uint32_t kOut = kFmt ? 8u : 16u;   // image bytes of a lane's 16-byte store
  const uint32_t frame_end = rp.slab_bytes + ((kOdd && !last_frame) ? kOut - 1u : 0u);   // exclusive
  const uint32_t wend = (woff0 + span - 1u < frame_end) ? woff0 + span - 1u : frame_end - 1u;
  const uint32_t ehi = __umulhi(rp.m, wend);
  const uint32_t last_row = (((wend - ehi) >> rp.sh1) + ehi) >> rp.sh2;
  const int slots = (int)(last_row - first_row + 1u) * P;
  // (slots <= 128 covers the common case; five to eight things on rows of a few hundred bytes make
  // two hundred: seven coins and a walker on a 4x9 board rendered at 2.5 TB/s through the tail loop)
  constexpr int kMaxIter = kRuntimeK ? 4 : 2;
  uint32_t ent[kMaxIter];
#pragma unroll
  for (int it = 0; it < kMaxIter; ++it) {
    const int sidx = lane + it * kWave;
    const int r = row_of_slot(sidx), p = sidx - r * P;
    const int d = kBoard ? p : (p >> 1);
    uint32_t row = first_row + (uint32_t)r;
    row = row <= last_row ? row : last_row;            // clamp: slot unused, entry ignored
    ent[it] = (it == 0 || slots > it * kWave) ? frame_trace[(int64_t)d * n_rows + trace_row(row)] : 0u;
  }

  // kMask: slot = (row, piece), sixteen (or fewer: a power of two) to a row, so a lane's piece is
  // the same in every round of slots; its table entry and the rows' masks travel with the things'
  // entries (the mask of one row is one address for all its slots: the loads coalesce)
  constexpr int kMaskIter = 4;     // (nine pieces on 140-byte rows: sixteen slots x fifteen rows)
  uint32_t piece = 0, shown[kMaskIter] = {0u, 0u, 0u, 0u};
  int mask_slots = 0;
  if constexpr (kMask) {
    const typename Fmt::Entry* masks = frame_trace + (int64_t)rp.n_dyn * n_rows;
    const int sh = rp.piece_shift;
    mask_slots = (int)(last_row - first_row + 1u) << sh;
    piece = rp.pieces[lane & ((1 << sh) - 1)];
#pragma unroll
    for (int it = 0; it < kMaskIter; ++it) {
      uint32_t row = first_row + (uint32_t)((lane + it * kWave) >> sh);
      row = row <= last_row ? row : last_row;
      shown[it] = (it == 0 || mask_slots > it * kWave) ? (uint32_t)masks[trace_row(row)] : 0u;
    }
  }

  if constexpr (kVar) {
    // (one load per ROW instead of two per chunk and lane: the 2-byte gathers were what made a
    // first form of this path 28 % slower than the plain kernel - it stores as fast as the
    // texture path delivers, and every extra wave-wide load is felt)
    uint16_t* row_variant = row_variant_all[wave];
    const typename Fmt::Entry* vars = frame_trace + (int64_t)rp.n_dyn * n_rows;
    const uint32_t top = (uint32_t)rp.B - 1u, limit = last_frame ? top : 2u * top + 1u;
    for (uint32_t i = (uint32_t)lane; i <= last_row - first_row + 1u; i += kWave) {
      uint32_t r = first_row + i;
      r = r < limit ? r : limit;       // rows past this frame's last: the next frame's first, or none to store
      row_variant[i] = (uint16_t)Fmt::cell(vars[trace_row(r)]);
    }
  }

  // ---- scenery: issue all loads, then park them in LDS
  u32x4 scen[kWin];
#pragma unroll
  for (int j = 0; j < kWin; ++j) {
    const uint32_t off = woff0 + j * 1024u + (uint32_t)lane * 16u;
    const uint32_t hi = __umulhi(rp.m, off);
    const uint32_t row = (((off - hi) >> rp.sh1) + hi) >> rp.sh2;  // off / R
    int k = (int)(off - row * rp.R);                                 // off % R
    // a chunk that starts 1..15 bytes before the frame: with frames that are not whole chunks, and
    // (round 5) with 16-bit observations of frames that are whole 8-element stores but not whole
    // 16-byte chunks of the IMAGE (B * R = 8 mod 16: boat race at B = 1 000) - every other frame
    // then starts 8 image bytes into a scenery chunk, and `off % R` of the wrapped offset put
    // another part of the row in the frame's first 8 elements
    if ((kOdd || kFmt != 0) && off >= 0xfffffff0u) k = R - (int)(0u - off);
    if constexpr (kVar) {
      // which environment's row the chunk starts in, and the next one's: rows past this frame's
      // last are the next frame's first (the trace plane is [T * B] rows), except in the
      // launch's last frame, whose chunks past the end are never stored; a chunk that starts
      // BEFORE the frame (wrapped offset) only ever has its bytes of row 0 stored by this block
      const bool before = off >= 0xfffffff0u;
      // (index into the wave's staged variants: rows from first_row on, clamped to what was staged)
      const uint32_t n_staged = last_row - first_row + 2u;
      uint32_t i0 = before ? 0u : row - first_row, i1 = before ? 0u : row + 1u - first_row;
      i0 = i0 < n_staged ? i0 : n_staged - 1u;
      i1 = i1 < n_staged ? i1 : n_staged - 1u;
      const int at = (k & 15) * pitch + (k & ~15);
      // What was measured on the way (tools/bench_variants.py: a 6x8 board whose whole floor turns,
      // two pictures, B = 262 144 / 65 536; the same game with a plain Backdrop through these kernels:
      // 6.6 / 6.5 TB/s; profiles/r06_variants.txt): the variant entry loaded per chunk and the
      // scenery chunk behind it, 4.9 / 5.0; the chunk fetched from the first FOUR variants beside
      // the entry, 3.9 / 4.3 (four times the loads of a kernel that stores as fast as the texture
      // path delivers); from the first TWO, 4.7 / 4.8; the entries staged once per row in LDS
      // (above), the same; the thing count a template argument instead of the run-time-K
      // instantiation, 5.0 / 5.4; and with all that ONE load again, of the variant the staged entry
      // names, 5.4 / 5.7 - what ships: the second trip costs less than a second candidate's bytes.
      // (All of these with a trace of two 52 MB planes read back from HBM: since the state-table
      // tier cuts such launches into chunks whose trace stays cached - campx_wide_rollout_launch -
      // the shipped form reads 6.42 against the plain kernel's 6.65.)
      // (Plain functions of values throughout: a form with closures over the vectors put 80 bytes
      // a lane into scratch memory - 1.3 TB/s.)
      const int8_t* here = rot + at;
      const uint32_t v0 = row_variant_all[wave][i0], v1 = row_variant_all[wave][i1];
      u32x4 mine = variant_chunk(here, rp.rot_stride, (int)v0);
      const int left = R - k;                          // bytes of the chunk inside row r0
      if (left < 16 && v1 != v0) mine = merge_rows(mine, variant_chunk(here, rp.rot_stride, (int)v1), left);
      scen[j] = mine;
    } else {
      scen[j] = *reinterpret_cast<const u32x4*>(rot + (k & 15) * pitch + (k & ~15));
    }
  }
  // the scenery layer of two cells per lane (kBoard needs none of it)
  uint32_t top2 = 0;
  if (!kBoard && !kWide) top2 = *reinterpret_cast<const uint16_t*>(rp.top_layer + 2 * lane);

#pragma unroll
  for (int j = 0; j < kWin; ++j)
    *reinterpret_cast<u32x4*>(win0 + j * 1024 + lane * 16) = scen[j];
  if (!kBoard && !kWide) {
    const uint32_t c = 2u * (uint32_t)lane;
    const uint32_t lo = (top2 & 0xffu) * (uint32_t)rp.cells + c;
    const uint32_t hi2 = (top2 >> 8) * (uint32_t)rp.cells + c + 1u;
    *reinterpret_cast<uint32_t*>(scen_off + c) = lo | (hi2 << 16);
  }

  // rp.dyn_off / dyn_char of a lane-varying thing: a chain of selects over the K kernel
  // arguments (indexing the array made hipcc fetch it with a vector load from the kernarg
  // segment and wait for it inside the patch branch: one more memory trip per wave)
  // (readfirstlane makes each argument an opaque scalar: from three things up the
  // optimiser otherwise turns the select chain back into the indexed load)
  auto of_thing = [&](const int32_t (&arr)[CAMPX_WIDE_MAX_DYN], int d) {
    int v = __builtin_amdgcn_readfirstlane(arr[0]);
#pragma unroll
    for (int k = 1; k < K; ++k) v = (d == k) ? __builtin_amdgcn_readfirstlane(arr[k]) : v;
    return v;
  };
  auto apply = [&](int sidx, uint32_t e) {
    const int r = row_of_slot(sidx), p = sidx - r * P;
    const int d = kBoard ? p : (p >> 1);
    const int cell = Fmt::cell(e);
    int byte;   // offset inside the row
    int8_t val;
    if (kBoard) {
      byte = cell;
      val = (int8_t)of_thing(rp.dyn_char, d);
    } else {
      int under;   // where the scenery's 1 that the thing hides sits
      if constexpr (kWide) under = Fmt::covered(e) * rp.cells + cell;
      else under = (int)scen_off[cell];
      byte = (p & 1) ? of_thing(rp.dyn_off, d) + cell : under;
      val = (int8_t)(p & 1);
    }
    // offsets inside a frame fit 32 bits (split_ok); a patch left of the window wraps to a
    // huge unsigned value and fails the one comparison
    const uint32_t at = (first_row + (uint32_t)r) * (uint32_t)R + (uint32_t)byte - woff0;
    if (sidx < slots && Fmt::visible(e) && at < span) win0[at] = val;
  };
#pragma unroll
  for (int it = 0; it < kMaxIter; ++it) apply(lane + it * kWave, ent[it]);
  for (int sidx = lane + kMaxIter * kWave; sidx < slots; sidx += kWave) {   // tiny rows only
    const int r = row_of_slot(sidx), p = sidx - r * P;
    const int d = kBoard ? p : (p >> 1);
    apply(sidx, frame_trace[(int64_t)d * n_rows + trace_row(first_row + (uint32_t)r)]);
  }
  if constexpr (kMask) {
    // (a piece that shows and a thing that shows never share a cell: the order does not matter)
    const int sh = rp.piece_shift;
    const uint32_t bit = 1u << (lane & ((1 << sh) - 1));
    auto lay = [&](int slot, uint32_t mask) {
      if (slot < mask_slots && (mask & bit)) {
        const uint32_t row0 = (first_row + (uint32_t)(slot >> sh)) * (uint32_t)R - woff0;
        const uint32_t a = row0 + (piece & 0xffffu), b = row0 + (piece >> 16);
        if (kBoard) {
          if (a < span) win0[a] = (int8_t)(piece >> 16);
        } else {
          if (a < span) win0[a] = 1;
          if (b < span) win0[b] = 0;
        }
      }
    };
#pragma unroll
    for (int it = 0; it < kMaskIter; ++it) lay(lane + it * kWave, shown[it]);
    const typename Fmt::Entry* masks = frame_trace + (int64_t)rp.n_dyn * n_rows;
    for (int slot = lane + kMaskIter * kWave; slot < mask_slots; slot += kWave)
      lay(slot, (uint32_t)masks[trace_row(first_row + (uint32_t)(slot >> sh))]);
  }

  // ---- out: aligned, contiguous KiB stores
  if (kFmt == 0) {
#pragma unroll
    for (int j = 0; j < kWin; ++j) {
      const uint32_t off = woff0 + j * 1024u + (uint32_t)lane * 16u;
      const int8_t* frame = dst + (int64_t)blockIdx.y * rp.slab_bytes;   // uniform
      if (off < rp.slab_bytes) {                         // the chunk starts inside the frame
        const u32x4 v = *reinterpret_cast<const u32x4*>(win0 + j * 1024 + lane * 16);
        if (kOdd && last_frame && off + 16u > rp.slab_bytes) {   // the launch's last bytes
          const uint32_t w[4] = {v.x, v.y, v.z, v.w};
          for (uint32_t i = 0; off + i < rp.slab_bytes; ++i)
            const_cast<int8_t*>(frame)[off + i] = (int8_t)(w[i >> 2] >> ((i & 3u) * 8u));
        } else if (kNT) {
          store16_streaming_at(frame, off, v);
        } else {
          *reinterpret_cast<u32x4*>(const_cast<int8_t*>(frame) + off) = v;
        }
      } else if (kOdd && blockIdx.y == 0 && off >= 0xfffffff0u) {   // the launch's first bytes
        const u32x4 v = *reinterpret_cast<const u32x4*>(win0 + j * 1024 + lane * 16);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        for (uint32_t i = 0u - off; i < 16u; ++i)
          const_cast<int8_t*>(frame)[(int32_t)(off + i)] = (int8_t)(w[i >> 2] >> ((i & 3u) * 8u));
      }
    }
  } else {
    // 16-bit observations (f16 / bf16 0.0 and 1.0) for a policy network: a lane turns
    // 8 bytes of the image into 8 halves; lanes stay contiguous, so each store
    // instruction is again one aligned KiB (of 512 elements).
    constexpr uint32_t kOne = (kFmt == 1) ? 0x3C00u : 0x3F80u;
#pragma unroll
    for (int h = 0; h < 2 * kWin; ++h) {
      const uint32_t elem = woff0 + (uint32_t)h * 512u + (uint32_t)lane * 8u;  // in the frame
      uint16_t* frame16 = reinterpret_cast<uint16_t*>(dst) + (int64_t)blockIdx.y * rp.slab_bytes;
      const bool inside = elem < rp.slab_bytes;
      const bool head = kOdd && blockIdx.y == 0 && elem >= 0xfffffff8u;   // the launch's first elements
      if (inside || head) {
        const uint2 b = *reinterpret_cast<const uint2*>(win0 + h * 512 + lane * 8);
        u32x4 v;
        v.x = ((b.x & 0xffu) | ((b.x << 8) & 0x00ff0000u)) * kOne;
        v.y = (((b.x >> 16) & 0xffu) | ((b.x >> 8) & 0x00ff0000u)) * kOne;
        v.z = ((b.y & 0xffu) | ((b.y << 8) & 0x00ff0000u)) * kOne;
        v.w = (((b.y >> 16) & 0xffu) | ((b.y >> 8) & 0x00ff0000u)) * kOne;
        if (head || (kOdd && last_frame && elem + 8u > rp.slab_bytes)) {   // element by element
          const uint32_t w[4] = {v.x, v.y, v.z, v.w};
          for (uint32_t i = head ? 0u - elem : 0u; i < 8u && (head || elem + i < rp.slab_bytes); ++i)
            frame16[(int32_t)(elem + i)] = (uint16_t)(w[i >> 1] >> ((i & 1u) * 16u));
        } else {
          u32x4* o = reinterpret_cast<u32x4*>(frame16 + elem);
          if (kNT)
            store16_streaming(o, v);
          else
            *o = v;
        }
      }
    }
  }
}

// `trace` points at the first frame to render, `T` frames from there; `plane_rows` is the
// distance (in rows = environments) between two moving things' planes of the trace, i.e.
// B times the number of frames the trace holds.  `src`: the game's render tables (the
// one-cell tier's live in its CampxSpec blob, the wide tier's in its table blob).
int32_t launch_render_from(const RenderSource& src, const void* trace, int8_t* dst, int64_t B,
                           int32_t T, int64_t plane_rows, int64_t pitch, bool is_board, int fmt,
                           hipStream_t stream) {
  const int HW = src.rows * src.cols;
  RenderParams rp;
  memset(&rp, 0, sizeof(rp));
  rp.R = (uint32_t)(is_board ? HW : src.n_layers * HW);
  // exact unsigned 32-bit division by R (Granlund & Montgomery 1994, fig. 4.1)
  uint32_t l = 0;
  while ((1ull << l) < rp.R) ++l;
  rp.m = (uint32_t)(((1ull << 32) * ((1ull << l) - rp.R)) / rp.R + 1);
  rp.sh1 = l < 1 ? l : 1;
  rp.sh2 = l > 0 ? l - 1 : 0;
  rp.slab_bytes = (uint32_t)(B * rp.R);
  rp.n_dyn = src.n_dyn;
  rp.is_board = is_board ? 1 : 0;
  rp.B = B;
  rp.pitch = pitch;
  rp.cells = HW;
  rp.rot = is_board ? src.rot_board : src.rot_obs;
  rp.rot_stride = is_board ? src.rot_board_stride : src.rot_obs_stride;
  rp.n_variants = src.n_variants > 1 ? src.n_variants : 1;
  {
    const uint32_t patches = (uint32_t)(is_board ? src.n_dyn : 2 * src.n_dyn);
    rp.inv_p = (uint32_t)(((1ull << 32) + patches - 1) / (patches ? patches : 1));
  }
  rp.top_layer = src.top_layer;
  if (src.n_pieces > 0) {
    rp.pieces = is_board ? src.pieces_board : src.pieces_obs;
    while ((1 << rp.piece_shift) < src.n_pieces) ++rp.piece_shift;
  }
  for (int d = 0; d < src.n_dyn; ++d) {
    rp.dyn_char[d] = src.layer_char[src.dyn_layer[d]];
    rp.dyn_off[d] = src.dyn_layer[d] * HW;
  }
  // KiB of the int8 image per wave: what a wave WRITES is what counts (2 KiB: 164.8 us, 4 KiB:
  // 180.2 us for the boat race), so the 16-bit formats take half the window
  constexpr int kWin = CAMPX_RENDER_WIN, kWin16 = kWin > 1 ? kWin / 2 : 1;
  const bool sixteen = !is_board && fmt != 0;
  const uint32_t wspan = 1024u * (uint32_t)(sixteen ? kWin16 : kWin);   // one wave's windows
  const uint32_t span = wspan * kRenderWaves;                            // one block's
  // windows aligned in memory; for the 16-bit formats in units of image bytes = elements
  rp.shift_base = (uint32_t)((reinterpret_cast<uintptr_t>(dst) >> (sixteen ? 1 : 0)) & (wspan - 1u));
  rp.shift_slab = rp.slab_bytes & (wspan - 1u);
  const uint64_t reach = (uint64_t)rp.slab_bytes + ((rp.shift_base | rp.shift_slab) ? wspan - 1u : 0u);
  // rounded up to a multiple of 8 for the XCD remap; surplus blocks exit at once
  const dim3 grid((unsigned)((((reach + span - 1u) / span) + 7u) & ~(uint64_t)7), (unsigned)T);
  const int64_t n_rows = plane_rows;
  const bool odd = (rp.slab_bytes & (sixteen ? 7u : 15u)) != 0;   // frames are not whole chunks
#define CAMPX_RENDER5(KK, BOARD, NT, FMT, ODD, WIDE)                                          \
  hipLaunchKernelGGL((render_kernel<KK, BOARD, NT, (FMT) ? kWin16 : kWin, FMT, ODD, WIDE>),   \
                     grid, dim3(kRenderWaves * kWave), 0, stream, rp,                         \
                     static_cast<const typename TraceFormat<WIDE>::Entry*>(trace), dst, n_rows)
#define CAMPX_RENDER4(KK, BOARD, NT, FMT, ODD) CAMPX_RENDER5(KK, BOARD, NT, FMT, ODD, false)
#define CAMPX_RENDER3(KK, BOARD, NT)                                      \
  do {                                                                    \
    if (!BOARD && fmt == 1 && odd) CAMPX_RENDER4(KK, false, NT, 1, true);   \
    else if (!BOARD && fmt == 1) CAMPX_RENDER4(KK, false, NT, 1, false);    \
    else if (!BOARD && fmt == 2 && odd) CAMPX_RENDER4(KK, false, NT, 2, true); \
    else if (!BOARD && fmt == 2) CAMPX_RENDER4(KK, false, NT, 2, false);    \
    else if (odd) CAMPX_RENDER4(KK, BOARD, NT, 0, true);                    \
    else CAMPX_RENDER4(KK, BOARD, NT, 0, false);                            \
  } while (0)
#define CAMPX_RENDER2(KK, BOARD)                                                    \
  do {                                                                              \
    /* (streaming stores always: plain ones measured 0.336 against 0.196 ms, campx_common.hip.h) */ \
    CAMPX_RENDER3(KK, BOARD, true);                                                 \
  } while (0)
#define CAMPX_RENDER1(KK)                                                   \
  do {                                                                      \
    if (is_board) CAMPX_RENDER2(KK, true); else CAMPX_RENDER2(KK, false);   \
  } while (0)
  if (src.wide && (src.n_variants > 1 || src.n_pieces > 0)) {
    // a scenery of several variants, or of pieces that come and go (the run-time-K instantiation
    // measured 28 % below the plain kernel for a one-thing game - its patch loop divides by a count
    // it only knows at run time - so here too the count is a template argument up to four things)
#define CAMPX_RENDER_VAR5(KK, BOARD, FMT, ODD, SCEN)                                                \
  hipLaunchKernelGGL((render_kernel<KK, BOARD, true, (FMT) ? kWin16 : kWin, FMT, ODD, true, SCEN>), \
                     grid, dim3(kRenderWaves * kWave), 0, stream, rp,                               \
                     static_cast<const uint16_t*>(trace), dst, n_rows)
#define CAMPX_RENDER_VAR4(KK, BOARD, FMT, SCEN)                                               \
  do {                                                                                        \
    if (odd) CAMPX_RENDER_VAR5(KK, BOARD, FMT, true, SCEN);                                   \
    else CAMPX_RENDER_VAR5(KK, BOARD, FMT, false, SCEN);                                      \
  } while (0)
#define CAMPX_RENDER_VAR3(KK, SCEN)                                                           \
  do {                                                                                        \
    if (is_board) CAMPX_RENDER_VAR4(KK, true, 0, SCEN);                                       \
    else if (fmt == 1) CAMPX_RENDER_VAR4(KK, false, 1, SCEN);                                 \
    else if (fmt == 2) CAMPX_RENDER_VAR4(KK, false, 2, SCEN);                                 \
    else CAMPX_RENDER_VAR4(KK, false, 0, SCEN);                                               \
  } while (0)
#define CAMPX_RENDER_VAR(KK)                                                                  \
  do {                                                                                        \
    if (src.n_pieces > 0) CAMPX_RENDER_VAR3(KK, 2); else CAMPX_RENDER_VAR3(KK, 1);            \
  } while (0)
    switch (src.n_dyn) {
      case 1: CAMPX_RENDER_VAR(1); break;
      case 2: CAMPX_RENDER_VAR(2); break;
      case 3: CAMPX_RENDER_VAR(3); break;
      case 4: CAMPX_RENDER_VAR(4); break;
      default: CAMPX_RENDER_VAR(8); break;
    }
#undef CAMPX_RENDER_VAR
#undef CAMPX_RENDER_VAR3
#undef CAMPX_RENDER_VAR4
#undef CAMPX_RENDER_VAR5
  } else if (src.wide) {
#define CAMPX_RENDER_WIDE(KK)                                             \
  do {                                                                    \
    if (is_board) {                                                       \
      if (odd) CAMPX_RENDER5(KK, true, true, 0, true, true);              \
      else CAMPX_RENDER5(KK, true, true, 0, false, true);                 \
    } else if (fmt == 1) {                                                \
      if (odd) CAMPX_RENDER5(KK, false, true, 1, true, true);             \
      else CAMPX_RENDER5(KK, false, true, 1, false, true);                \
    } else if (fmt == 2) {                                                \
      if (odd) CAMPX_RENDER5(KK, false, true, 2, true, true);             \
      else CAMPX_RENDER5(KK, false, true, 2, false, true);                \
    } else {                                                              \
      if (odd) CAMPX_RENDER5(KK, false, true, 0, true, true);             \
      else CAMPX_RENDER5(KK, false, true, 0, false, true);                \
    }                                                                     \
  } while (0)
    switch (src.n_dyn) {
      case 1: CAMPX_RENDER_WIDE(1); break;
      case 2: CAMPX_RENDER_WIDE(2); break;
      case 3: CAMPX_RENDER_WIDE(3); break;
      case 4: CAMPX_RENDER_WIDE(4); break;
      default: CAMPX_RENDER_WIDE(8); break;    // five to eight things: the count at run time
    }
#undef CAMPX_RENDER_WIDE
  } else {
    switch (src.n_dyn) {
      case 1: CAMPX_RENDER1(1); break;
      case 2: CAMPX_RENDER1(2); break;
      case 3: CAMPX_RENDER1(3); break;
      default: CAMPX_RENDER1(4); break;
    }
  }
#undef CAMPX_RENDER1
#undef CAMPX_RENDER2
#undef CAMPX_RENDER3
#undef CAMPX_RENDER4
#undef CAMPX_RENDER5
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? CAMPX_OK : hip_failed(e);
}

int32_t launch_render(const CampxSpec& s, const CampxSpec* spec_dev, const uint8_t* trace,
                      int8_t* dst, int64_t B, int32_t T, int64_t plane_rows, int64_t pitch,
                      bool is_board, int fmt, hipStream_t stream) {
  RenderSource src;
  memset(&src, 0, sizeof(src));
  src.rows = s.rows;
  src.cols = s.cols;
  src.n_layers = s.n_layers;
  src.n_dyn = s.n_dyn;
  for (int d = 0; d < s.n_dyn; ++d) src.dyn_layer[d] = s.dyn_layer[d];
  memcpy(src.layer_char, s.layer_char, sizeof(src.layer_char));
  const char* blob = reinterpret_cast<const char*>(spec_dev);
  src.rot_obs = reinterpret_cast<const int8_t*>(blob + offsetof(CampxSpec, rot_obs));
  src.rot_board = reinterpret_cast<const int8_t*>(blob + offsetof(CampxSpec, rot_board));
  src.top_layer = reinterpret_cast<const uint8_t*>(blob + offsetof(CampxSpec, static_top_layer));
  src.wide = false;
  return launch_render_from(src, trace, dst, B, T, plane_rows, pitch, is_board, fmt, stream);
}

}  // namespace campx_impl
