// k_update.hip - the update pass of the two-kernel path from the game's state table:
// producer / consumer / loader waves (update_table_kernel, update_pair_kernel,
// update_tuple_kernel) and launch_update, which picks one (or the interpreter in trace mode).
#include "campx_common.hip.h"

#include <mutex>

#include <stdio.h>
#include <type_traits>

namespace campx_impl {

// ---------------------------------------------------------------------------
// Split path, first half: the update pass alone.  Out: the trace (one byte per moving
// thing per frame per environment: cell + "is the character its cell shows") and the
// per-frame scalars (reward, discount, done, perf).
//
// The only loop-carried dependency of a frame is state -> table[state, action] ->
// state: one LDS read.  A workgroup owns kEnvs = 64 * kProd consecutive environments
// and has three kinds of waves:
//   kProd producer waves   run that dependent chain, one environment per lane,
//                          a group of (16) frames at a time, into a double-buffered LDS ring;
//   kCons consumer waves   turn the previous group into the output streams while the
//                          producers run the next one: 16 bytes per lane per store
//                          (4 environments of a float stream, 16 of a byte stream),
//                          write-through, so that nothing is left dirty in L2 for the
//                          end-of-kernel flush and the render kernel behind it;
//   kProd / 2 loader waves bring the actions in, 16 bytes per lane per load, one
//                          64-frame chunk ahead.  They issue no stores, so waiting for
//                          their loads never waits for a store (vmcnt is in order).
// One s_barrier per group, in each role's own loop.  The accesses are 16 bytes whatever the
// batch size: dword-aligned for the float streams, byte-aligned for the byte streams and the
// actions when B is not a multiple of 16 (legal on this stack; tools/probes/unaligned_probe.hip),
// and only the batch's last, partial group of 16 environments goes byte by byte.

// Cache policy of the update kernels' output stores (A/B builds): 0 plain, 1 sc0 sc1
// (write-through), 2 sc0 sc1 nt.
#ifndef CAMPX_UPD_FLAVOR
#define CAMPX_UPD_FLAVOR 1
#endif
// (CAMPX_UPD_CLOBBER=0: the store asm without its "memory" clobber, so that the compiler may
// move a consumer's next LDS reads above it - an A/B build)
#ifndef CAMPX_UPD_CLOBBER
#define CAMPX_UPD_CLOBBER 1
#endif
#if CAMPX_UPD_CLOBBER
#define CAMPX_UPD_CLOBBERS : "memory"
#else
#define CAMPX_UPD_CLOBBERS
#endif
__device__ __forceinline__ void store16_update(void* p, u32x4 v) {
#if defined(CAMPX_UPD_DEBUG) && CAMPX_UPD_DEBUG == 1
  if (v.x == 0x12345678u && v.y == 0x9abcdef0u) *reinterpret_cast<u32x4*>(p) = v;   // (never)
  return;
#endif
#if CAMPX_UPD_FLAVOR == 0
  *reinterpret_cast<u32x4*>(p) = v;
#elif CAMPX_UPD_FLAVOR == 1
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) CAMPX_UPD_CLOBBERS);
#else
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) CAMPX_UPD_CLOBBERS);
#endif
}

// The tagged copy of the trace (one-launch rollouts) is read by render waves of the SAME launch,
// on other XCDs, as it is written: always write-through, whatever CAMPX_UPD_FLAVOR an A/B build
// gives the other streams (a plain store would sit in the writer's L2 until the kernel ends).
__device__ __forceinline__ void store16_tagged(void* p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// Sixteen trace bytes (four packed dwords) as sixteen entries byte | tag << 8: two such stores.
__device__ __forceinline__ void store_tagged_row(uint16_t* at, const uint32_t (&tr)[4], uint32_t tag) {
  const uint32_t tt = (tag << 8) | (tag << 24);
  uint32_t e[8];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    e[2 * k] = (tr[k] & 0xffu) | ((tr[k] & 0xff00u) << 8) | tt;
    e[2 * k + 1] = ((tr[k] >> 16) & 0xffu) | ((tr[k] >> 8) & 0xff0000u) | tt;
  }
  store16_tagged(at, u32x4{e[0], e[1], e[2], e[3]});
  store16_tagged(at + 8, u32x4{e[4], e[5], e[6], e[7]});
}
__device__ __forceinline__ void store_tagged_one(uint16_t* at, uint32_t byte, uint32_t tag) {
  __hip_atomic_store(at, (uint16_t)((byte & 0xffu) | (tag << 8)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Number of bytes >= 5 (as unsigned) among the 16 of v: the action ids outside 0..4.
__device__ __forceinline__ int count_bad16(u32x4 v) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  int n = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t hi = (((w[i] & 0x7f7f7f7fu) + 0x7b7b7b7bu) | w[i]) & 0x80808080u;
    n += __builtin_popcount(hi);
  }
  return n;
}

// Bytes > 4 (as unsigned) of v replaced by 4: an id outside 0..4 acts as "stay".
__device__ __forceinline__ uint32_t clamp_ids(uint32_t w) {
  const uint32_t hi = (((w & 0x7f7f7f7fu) + 0x7b7b7b7bu) | w) & 0x80808080u;  // bad bytes
  const uint32_t m = (hi >> 7) * 0xffu;
  return (w & ~m) | (0x04040404u & m);
}

// The staged actions of a chunk are packed two frames to a byte - frame 2p in the low
// nibble of row p, frame 2p + 1 in the high one, already clamped to 0..4 - which halves
// their LDS footprint (what decides how many update workgroups fit on a CU).
//
// The loader waves' state: one chunk (kChunk frames x kEnvs environments) of actions in
// flight in registers between issue() and land().
template <int kEnvs, int kLoaders>
struct ActionLoader {
  static constexpr int kVecPerRow = kEnvs / 16;
  static constexpr int kLanes = kWave * kLoaders;          // loader lanes of the workgroup
  static constexpr int kPairs = (kChunk / 2) * kVecPerRow / kLanes;   // row pairs per lane
  static_assert(kPairs >= 1 && (kChunk / 2) * kVecPerRow % kLanes == 0, "loader shape");
  u32x4 pend[2 * kPairs];

  // 16-byte loads (byte-aligned unless B % 16 == 0: unaligned 16-byte accesses are legal on
  // this stack, tools/probes/unaligned_probe.hip), no branch between them; rows past T and
  // groups of 16 environments that are not wholly below B are clamped to valid addresses
  // inside the buffer's T * B bytes (a C-ABI caller may have allocated exactly that) and
  // redone or neutralised in land().
  // `lane` counts over all loader waves: 0 .. kLanes-1
  __device__ __forceinline__ void issue(const int8_t* __restrict__ actions, int64_t B, int32_t T,
                                        int t0, int64_t env0, int lane) {
    const int64_t last = (int64_t)T * B - 16;
    if (last < 0) {   // uniform: a buffer shorter than one load; land() reads it byte by byte
#pragma unroll
      for (int i = 0; i < 2 * kPairs; ++i) pend[i] = u32x4{0x04040404u, 0x04040404u, 0x04040404u, 0x04040404u};
      return;
    }
#pragma unroll
    for (int i = 0; i < kPairs; ++i) {
      const int v = lane + i * kLanes;
      const int rp = v / kVecPerRow, q = v % kVecPerRow;
      int64_t e = env0 + 16 * q;
      e = e + 16 <= B ? e : 0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int row = t0 + 2 * rp + h;
        row = row < T ? row : T - 1;
        int64_t at = (int64_t)row * B + e;
        at = at <= last ? at : last;   // (only ever changes a clamped, ignored load)
        pend[2 * i + h] = *reinterpret_cast<const u32x4*>(actions + at);
      }
    }
  }

  // v shifted down by `bytes` (0..15) bytes, "stay" ids shifted in at the top
  static __device__ __forceinline__ u32x4 shift_down(u32x4 v, uint32_t bytes) {
    const uint32_t w[5] = {v.x, v.y, v.z, v.w, 0x04040404u};
    const uint32_t dw = bytes >> 2, sub = bytes & 3u;
    uint32_t x[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      uint32_t t = 0x04040404u;
#pragma unroll
      for (int k = 0; k < 4; ++k) t = (dw == (uint32_t)k && i + k < 5) ? w[i + k < 5 ? i + k : 4] : t;
      x[i] = t;
    }
    u32x4 r;
    r.x = __builtin_amdgcn_alignbyte(x[1], x[0], sub);
    r.y = __builtin_amdgcn_alignbyte(x[2], x[1], sub);
    r.z = __builtin_amdgcn_alignbyte(x[3], x[2], sub);
    r.w = __builtin_amdgcn_alignbyte(x[4], x[3], sub);
    return r;
  }

  // bytes n .. 15 of v replaced by "stay"
  static __device__ __forceinline__ u32x4 keep_low(u32x4 v, int n) {
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int have = n - 4 * i;                        // bytes of this dword that stay
      const uint32_t m = have >= 4 ? 0xffffffffu : (have <= 0 ? 0u : (1u << (8 * have)) - 1u);
      w[i] = (w[i] & m) | (0x04040404u & ~m);
    }
    return u32x4{w[0], w[1], w[2], w[3]};
  }

  __device__ __forceinline__ int land(int8_t* staged, const int8_t* __restrict__ actions, int64_t B,
                                      int32_t T, int t0, int64_t env0, int lane) {
    int bad = 0;
    const int64_t last = (int64_t)T * B - 16;
#pragma unroll
    for (int i = 0; i < kPairs; ++i) {
      const int v = lane + i * kLanes;
      const int rp = v / kVecPerRow, q = v % kVecPerRow;
      const int64_t e = env0 + 16 * q;
      const bool here = e + 16 <= B;
      const bool real0 = here && (t0 + 2 * rp < T), real1 = here && (t0 + 2 * rp + 1 < T);
      u32x4 lo = pend[2 * i], hi = pend[2 * i + 1];
      if (!here && e < B) {
        // The batch's last, PARTIAL group of 16 environments (one lane per row pair of the
        // last workgroup).  Loaded like any other group - its upper bytes are then the first
        // environments of the next row, masked to "stay" - except where that would pass the
        // end of the buffer (the last row): there the load moves back to the buffer's last
        // 16 bytes and the wanted bytes are shifted down.  (Until round 3 this went byte by
        // byte: one lane's chain of thirty dependent loads per chunk, 20 us on top of every
        // launch whose batch size is not a multiple of 16.)
        const int n_env = (int)(B - e);
        u32x4 x[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int row = t0 + 2 * rp + h;
          x[h] = u32x4{0x04040404u, 0x04040404u, 0x04040404u, 0x04040404u};
          if (row < T && last >= 0) {
            const int64_t at = (int64_t)row * B + e;
            x[h] = *reinterpret_cast<const u32x4*>(actions + (at <= last ? at : last));
            if (at > last) x[h] = shift_down(x[h], (uint32_t)(at - last));
            x[h] = keep_low(x[h], n_env);
          } else if (row < T) {
            // (uniform) the whole buffer is shorter than one load: byte by byte, static
            // indices only - a dynamically indexed private array lives in scratch
            uint32_t w[4] = {0x04040404u, 0x04040404u, 0x04040404u, 0x04040404u};
#pragma unroll
            for (int k = 0; k < 16; ++k) {
              if (k < n_env) {
                const uint32_t a = (uint8_t)actions[(int64_t)row * B + e + k];
                w[k >> 2] = (w[k >> 2] & ~(0xffu << (8 * (k & 3)))) | (a << (8 * (k & 3)));
              }
            }
            x[h] = u32x4{w[0], w[1], w[2], w[3]};
          }
        }
        const uint32_t l[4] = {x[0].x, x[0].y, x[0].z, x[0].w}, h4[4] = {x[1].x, x[1].y, x[1].z, x[1].w};
        uint32_t out[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k] = clamp_ids(l[k]) | (clamp_ids(h4[k]) << 4);
        const u32x4 packed = {out[0], out[1], out[2], out[3]};
        *reinterpret_cast<u32x4*>(staged + rp * kEnvs + 16 * q) = packed;
        bad += count_bad16(x[0]) + count_bad16(x[1]);   // (neutralised bytes are 4: never counted)
        continue;
      }
      const uint32_t l[4] = {lo.x, lo.y, lo.z, lo.w}, h[4] = {hi.x, hi.y, hi.z, hi.w};
      uint32_t out[4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        out[k] = (real0 ? clamp_ids(l[k]) : 0x04040404u) | ((real1 ? clamp_ids(h[k]) : 0x04040404u) << 4);
      const u32x4 packed = {out[0], out[1], out[2], out[3]};
      *reinterpret_cast<u32x4*>(staged + rp * kEnvs + 16 * q) = packed;
      bad += (real0 ? count_bad16(lo) : 0) + (real1 ? count_bad16(hi) : 0);
    }
    return bad;
  }
};

// The action of frame j of a group that starts at frame t0 of its chunk,
// for environment `le` of the workgroup.
template <int kEnvs>
__device__ __forceinline__ uint32_t staged_action(const int8_t* chunk, int t0_in_chunk, int j,
                                                   int le) {
  const uint32_t b = (uint8_t)chunk[((t0_in_chunk + j) >> 1) * kEnvs + le];
  return (b >> (4 * (j & 1))) & 0xfu;   // t0_in_chunk is a multiple of the group size (even)
}

// Four bytes (the low byte of each argument) as one dword.
__device__ __forceinline__ uint32_t pack4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  return (a & 0xffu) | ((b & 0xffu) << 8) | ((c & 0xffu) << 16) | (d << 24);
}

// A/B: give each XCD a contiguous eighth of the batch (see tile_of_block).
#ifndef CAMPX_UPD_XCD
#define CAMPX_UPD_XCD 0
#endif

// loader waves of an update workgroup: one per 128 environments (a 256-environment
// chunk held by one wave is 64 VGPRs of loads in flight: with two the pair kernel stops
// spilling).  CAMPX_UPD_LOADERS overrides for A/B builds.
#ifdef CAMPX_UPD_LOADERS
constexpr int update_loaders(int) { return CAMPX_UPD_LOADERS; }
#else
constexpr int update_loaders(int prod) { return prod >= 4 ? prod / 2 : 1; }
#endif

// A/B knobs: roll the consumers' float-stream loop (fewer registers, measured +0.8 us on
// the boat race), and a register cap in waves per SIMD (capping the pair kernel at 96
// VGPRs for two workgroups per CU measured 47 us against 42 us uncapped at 110).
#ifndef CAMPX_UPD_ROLL_A
#define CAMPX_UPD_ROLL_A 0
#endif
#ifndef CAMPX_UPD_MINWAVES
#define CAMPX_UPD_MINWAVES 1
#endif
constexpr int update_min_waves(int, int) { return CAMPX_UPD_MINWAVES; }

// Where this kernel's time goes (round 3, kernel trace at B = 4 096, T = 16 .. 512:
// 8.2 / 10.0 / 11.9 / 17.0 / 28.9 / 52.2 us): ~6 us fixed (launch, the table and the first
// chunk of actions, the drain) + 92 ns per frame.  The per-frame part is NOT the producers'
// dependent chain (state -> entry -> state, one LDS lookup per frame), as rounds 1-2 assumed:
//   * CAMPX_UPD_COMPOSE == 2 halves the chain - a second table, built per workgroup from the
//     first, maps (cell, action, action) straight to the cell TWO frames on, the per-frame
//     entries are looked up off the chain, the three lookups of a step issued together - and
//     changes nothing (T = 256: 29.1 against 29.0 us; B = 65 536: 16.9 against 16.6);
//   * neither does removing the per-frame `j < n` test and its scalar branch (whole groups run
//     branch-free now);
//   * -DCAMPX_UPD_DEBUG=2 (consumers idle) reads 21.3 us at T = 256, =3 (producers idle) 27.3:
//     the CONSUMER waves bound the frame (83 ns: LDS read -> unpack -> store, five dependent
//     rounds per group), the producers come second (60 ns);
//   * -DCAMPX_UPD_DEBUG=1 (no global stores at all) saves 0.8 us: it is not the memory system.
// Hence 8 consumer waves instead of 4 for the 256-environment workgroups (B = 4 096: 14.9 ->
// 12.9 us; T = 256: 29.0 -> 23.5); at B = 65 536 the kernel reads 16.3-16.7 us either way.
// (The composed chain was an A/B build through round 3; round 4 removed it with its knob.)

// LDS of an update workgroup of the one-mover kernel (a struct, so that pipe_table_kernel - this
// body as one role of a launch - can lay its other role's windows over the same bytes).
template <int kProd, int kG>
struct UpdateTableLds {
  // entry: x = reward; y = [0:15] byte offset of the table row the NEXT frame starts
  // from (the art's cell when this frame ended the episode: the rebuild is folded into
  // the chain), [16:22] the cell after this frame, [23] whether the mover shows there,
  // [24] done, [25:26] + [31] hidden-performance code, [27:30] discount code.  The ring keeps
  // x and the upper half of y.
  uint2 table[CAMPX_MAX_CELLS * CAMPX_N_ACTIONS];
  float discounts[16];
  __attribute__((aligned(16))) int8_t staged[2][(kChunk / 2) * kProd * kWave];
  __attribute__((aligned(16))) float ring_r[2][kG][kProd * kWave];
  __attribute__((aligned(16))) uint16_t ring_y[2][kG][kProd * kWave];
};

// The body of update_table_kernel.  `wg`: which kEnvs environments this workgroup owns.
// `tagged` (one-launch rollouts, else null): a second copy of the trace as 16-bit entries,
// byte | tag << 8 - every entry says by itself which launch wrote it, so a render wave of the
// same launch needs no flag and this role no drain (MI355X_MICROARCH.md: data-tagged granules).
template <int kProd, int kCons, int kG, bool kTagged = false>
__device__ __forceinline__ void update_table_body(
    UpdateTableLds<kProd, kG>& L, uint32_t wg, const MoverParams& mp,
    const CampxSpec* __restrict__ spec, const CampxState& st,
    const int8_t* __restrict__ actions, const CampxOutputs& out, int64_t B, int32_t T,
    int32_t reset_first, const FrameCodec& fc, uint16_t* tagged = nullptr, uint32_t tag = 0) {
  constexpr int kLoad = update_loaders(kProd);
  constexpr int E = kProd * kWave, CL = kCons * kWave, kThreads = (kProd + kCons + kLoad) * kWave;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const bool producer = wave < kProd, loader = wave >= kProd + kCons;
  const int llane = (int)threadIdx.x - (kProd + kCons) * kWave;  // loaders: 0 .. 64*kLoad-1
  const int W = mp.cols, HW = mp.rows * mp.cols;
  const int64_t env0 = (int64_t)wg * E;

  const int cell0 = mp.row0 * W + mp.col0;
  constexpr int kRowBytes = CAMPX_N_ACTIONS * (int)sizeof(uint2);
  for (int i = threadIdx.x; i < HW * CAMPX_N_ACTIONS; i += kThreads) {
    const CampxTransition tr = spec->table[i];
    const uint32_t ended = tr.done & 1u, dcode = tr.done >> 4;
    const uint32_t from = ended ? (uint32_t)cell0 : (uint32_t)tr.next_cell;
    const uint32_t vis = (tr.paint & 0x80u) ? 0u : 1u;  // scenery in front hides the mover
    // (the table holds perf VALUES; a game without hidden performance has scale 0)
    const uint32_t pc = fc.perf_scale ? (uint32_t)(((int)tr.perf - fc.perf_offset) / fc.perf_scale) & 7u : 0u;
    L.table[i] = make_uint2(__float_as_uint(tr.reward),
                          (from * kRowBytes) | ((uint32_t)tr.next_cell << 16) | (vis << 23) |
                              (ended << 24) | ((pc & 3u) << 25) | (dcode << 27) | ((pc >> 2) << 31));
  }
  if (threadIdx.x < 16) L.discounts[threadIdx.x] = fc.discounts[threadIdx.x];
  ActionLoader<E, kLoad> ld;
  int bad = 0;
  if (loader && T > 0) {
    ld.issue(actions, B, T, 0, env0, llane);
    bad += ld.land(L.staged[0], actions, B, T, 0, env0, llane);
  }

  const int le = wave * kWave + lane;  // producers: this lane's environment in the workgroup
  const int64_t env = env0 + le;
  const bool live = producer && env < B;
  int cell = cell0, over = 0;
  float ret = 0.0f;
  if (live && !reset_first) {
    cell = (int)st.pos[env] * W + (int)st.pos[B + env];
    over = st.done[env];
    if (st.ret) ret = st.ret[env];
  }
  uint32_t row_off = (uint32_t)(over ? cell0 : cell) * kRowBytes;  // the chain's state
  uint32_t last_y = ((uint32_t)cell << 16) | ((uint32_t)over << 24);  // upper half of the latest entry
  const char* table_bytes = reinterpret_cast<const char*>(L.table);
  const int clane = (int)threadIdx.x - kProd * kWave;  // consumers: 0 .. CL-1
  constexpr int kGroupsPerChunk = kChunk / kG;
  __syncthreads();

  const int n_groups = (T + kG - 1) / kG;
  // Each kind of wave runs its own loop (one s_barrier per group in each, so the counts
  // agree): registers are then allocated per role, and the loads a loader keeps in flight
  // across iterations do not take registers from the other two.
  if (producer) {
    for (int g = 0; g <= n_groups; ++g) {
        if (g < n_groups) {
          const int t0 = g * kG;
          const int8_t* chunk = L.staged[(t0 / kChunk) & 1];
          const int n = (T - t0 < kG) ? T - t0 : kG;
          uint32_t col_off[kG];  // action * sizeof(entry), off the dependent chain
  #pragma unroll
          for (int j = 0; j < kG; ++j)
            col_off[j] = staged_action<E>(chunk, t0 & (kChunk - 1), j, le) * (uint32_t)sizeof(uint2);
          // A frame's bookkeeping besides the chain: into the ring, and the running return
          // (which restarts after an episode end: bit 24 of the PREVIOUS frame's entry).  Where
          // the mover stands and whether the episode is over are read off the last entry
          // after the loop.  `kFull`: a whole group - no per-frame test, hence no branch
          // between frames, and the compiler overlaps one frame's bookkeeping with the next
          // frame's lookup (with the test, every frame ended in a scalar branch and waited
          // for its own LDS read: the kernel spent ~190 cycles per frame issuing, not waiting
          // on the chain; round 3 measured the chain itself by halving it: no change).
          auto frames = [&](auto full_tag) {
            constexpr bool kFull = decltype(full_tag)::value;
            auto book = [&](int j, uint2 e) {
              if (kFull || j < n) {
                L.ring_r[g & 1][j][le] = __uint_as_float(e.x);
                L.ring_y[g & 1][j][le] = (uint16_t)(e.y >> 16);
                ret = (((last_y >> 24) & 1u) ? 0.0f : ret) + real_reward(__uint_as_float(e.x));
                last_y = e.y;
              }
            };
  #pragma unroll
            for (int j = 0; j < kG; ++j) {
              if (kFull || j < n) {
                // the dependent chain: row offset -> entry -> row offset
                const uint2 e = *reinterpret_cast<const uint2*>(table_bytes + row_off + col_off[j]);
                row_off = e.y & 0xffffu;
                book(j, e);
              }
            }
          };
#if defined(CAMPX_UPD_DEBUG) && CAMPX_UPD_DEBUG == 3
          if (T < 0) frames(std::true_type{});
#else
          if (n == kG) frames(std::true_type{});
          else frames(std::false_type{});
#endif
        }
      
      __syncthreads();
    }
  } else if (!loader) {
    // rows of the output streams are P elements apart; Bv: how much of a row may be written
    // (a padded pitch lets the batch's last group be stored whole, into the pad)
    const int64_t P = row_pitch(out, B), Bv = row_extent(out, B);
    for (int g = 0; g <= n_groups; ++g) {
#if defined(CAMPX_UPD_DEBUG) && CAMPX_UPD_DEBUG == 2
        if (g > 0 && T < 0) {
#else
        if (g > 0) {
#endif
          const int gp = g - 1, t0 = gp * kG, rb = gp & 1;
          const int n = (T - t0 < kG) ? T - t0 : kG;
          // ---- float streams: item = (frame j, 4 environments)
          constexpr int QA = E / 4, kItA = (kG * QA + CL - 1) / CL;
  #if CAMPX_UPD_ROLL_A
#pragma unroll 1
#else
#pragma unroll
#endif
          for (int it = 0; it < kItA; ++it) {
            const int item = clane + it * CL;
            const int j = item / QA, q = item % QA;
            const int64_t e0 = env0 + 4 * q;
            if (j < n && e0 < B) {
              const u32x4 r4 = *reinterpret_cast<const u32x4*>(&L.ring_r[rb][j][4 * q]);
              const uint2 y4 = *reinterpret_cast<const uint2*>(&L.ring_y[rb][j][4 * q]);
              const uint32_t y[4] = {y4.x & 0xffffu, y4.x >> 16, y4.y & 0xffffu, y4.y >> 16};
              uint32_t dc[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                dc[i] = ((y[i] >> 8) & 1u) ? 0u : 0x3f800000u;
                if (fc.has_dcodes) dc[i] = discount_bits(L.discounts, dcode_y16(y[i]), (y[i] >> 8) & 1u);
              }
              const int64_t at = (int64_t)(t0 + j) * P + e0;
              if (e0 + 4 <= Bv) {   // (dword-aligned; 16-byte aligned when the row pitch is a multiple of 4)
                if (out.reward) store16_update(out.reward + at, r4);
                if (out.discount) {
                  const u32x4 d4 = {dc[0], dc[1], dc[2], dc[3]};
                  store16_update(out.discount + at, d4);
                }
              } else {
                const uint32_t rw[4] = {r4.x, r4.y, r4.z, r4.w};
                for (int i = 0; i < 4 && e0 + i < B; ++i) {
                  if (out.reward) out.reward[at + i] = __uint_as_float(rw[i]);
                  if (out.discount) out.discount[at + i] = __uint_as_float(dc[i]);
                }
              }
            }
          }
          // ---- byte streams: item = (frame j, 16 environments)
          constexpr int QB = E / 16, kItB = (kG * QB + CL - 1) / CL;
  #pragma unroll
          for (int it = 0; it < kItB; ++it) {
            const int item = clane + it * CL;
            const int j = item / QB, q = item % QB;
            const int64_t e0 = env0 + 16 * q;
            if (item < kG * QB && j < n && e0 < B) {
              const u32x4 ya = *reinterpret_cast<const u32x4*>(&L.ring_y[rb][j][16 * q]);
              const u32x4 yb = *reinterpret_cast<const u32x4*>(&L.ring_y[rb][j][16 * q + 8]);
              const uint32_t w[8] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w};
              uint32_t tr[4], dn[4], pf[4];
  #pragma unroll
              for (int k = 0; k < 4; ++k) {
                const uint32_t lo = w[2 * k], hi = w[2 * k + 1];  // two environments each
                tr[k] = pack4(lo, lo >> 16, hi, (hi >> 16) & 0xffu);
                dn[k] = pack4((lo >> 8) & 1u, (lo >> 24) & 1u, (hi >> 8) & 1u, (hi >> 24) & 1u);
                pf[k] = pack4(perf_byte(fc, perf_code_y16(lo & 0xffffu)), perf_byte(fc, perf_code_y16(lo >> 16)),
                              perf_byte(fc, perf_code_y16(hi & 0xffffu)), perf_byte(fc, perf_code_y16(hi >> 16)));
              }
              const int64_t at = (int64_t)(t0 + j) * P + e0;
              if (e0 + 16 <= Bv) {  // (aligned when the row pitch is a multiple of 16; else legal, slower)
                const u32x4 t4 = {tr[0], tr[1], tr[2], tr[3]};
                store16_update(out.trace + at, t4);
                if (kTagged) store_tagged_row(tagged + at, tr, tag);
                if (out.done) {
                  const u32x4 d4 = {dn[0], dn[1], dn[2], dn[3]};
                  store16_update(out.done + at, d4);
                }
                if (out.perf) {
                  const u32x4 p4 = {pf[0], pf[1], pf[2], pf[3]};
                  store16_update(out.perf + at, p4);
                }
              } else {
                for (int i = 0; i < 16 && e0 + i < B; ++i) {
                  const int sh = (i & 3) * 8;
                  out.trace[at + i] = (uint8_t)(tr[i >> 2] >> sh);
                  if (kTagged) store_tagged_one(tagged + at + i, tr[i >> 2] >> sh, tag);
                  if (out.done) out.done[at + i] = (uint8_t)(dn[i >> 2] >> sh);
                  if (out.perf) out.perf[at + i] = (int8_t)(pf[i >> 2] >> sh);
                }
              }
            }
          }
        }
      __syncthreads();
    }
  } else {
    for (int g = 0; g <= n_groups; ++g) {
        // loader: while the producers are in chunk c, fetch chunk c + 1
        const int c = g / kGroupsPerChunk, phase = g % kGroupsPerChunk;
        const int t_next = (c + 1) * kChunk;
        if (t_next < T) {
          if (phase == 0) ld.issue(actions, B, T, t_next, env0, llane);
          if (phase == kGroupsPerChunk - 1)
            bad += ld.land(L.staged[(c + 1) & 1], actions, B, T, t_next, env0, llane);
        }
      
      __syncthreads();
    }
  }

  if (live) {
    cell = (int)((last_y >> 16) & 0x7fu);
    over = (int)((last_y >> 24) & 1u);
    st.pos[env] = (int8_t)(cell / W);
    st.pos[B + env] = (int8_t)(cell % W);
    st.done[env] = (uint8_t)over;
    if (st.ret) st.ret[env] = ret;
  }
  report_bad_actions(out, bad);
}


template <int kProd, int kCons, int kG>   // kG: frames per group (a ring slot)
__global__ __launch_bounds__((kProd + kCons + update_loaders(kProd)) * kWave,
                             update_min_waves(kProd, kCons)) void update_table_kernel(
    MoverParams mp, const CampxSpec* __restrict__ spec, CampxState st,
    const int8_t* __restrict__ actions, CampxOutputs out, int64_t B, int32_t T,
    int32_t reset_first, FrameCodec fc) {
  __shared__ UpdateTableLds<kProd, kG> L;
  update_table_body<kProd, kCons, kG>(L, tile_of_block(blockIdx.x, gridDim.x, CAMPX_UPD_XCD), mp, spec, st,
                                      actions, out, B, T, reset_first, fc);
}

// ---------------------------------------------------------------------------
// Two-mover games: the same layout over the (cell, cell, action) pair table
// (campx_pair_table_build).  The dependent chain goes through the 32-bit entries
// themselves, which carry everything a frame outputs and go to the ring as they are; they
// sit in LDS (dynamic, kLdsEntries) when the table fits, else they are read through L1/L2.
// (Chaining through a separate 16-bit next-index table in LDS with the entries fetched off
// the chain from global memory was built and measured slower - 68.9 against 51.4 us - and
// removed.)
struct PairParams {
  int32_t rows, cols, n_layers;
  int32_t dyn_layer[2], row0[2], col0[2];
  int32_t lds_table;  // entries fit in LDS
};

#ifndef CAMPX_PAIR_LDS_ENTRIES
#define CAMPX_PAIR_LDS_ENTRIES 8192
#endif
constexpr int kPairLdsEntries = CAMPX_PAIR_LDS_ENTRIES;  // 32 KiB of LDS for the entries at most

#ifndef CAMPX_PAIR_GROUP
#define CAMPX_PAIR_GROUP 16   // frames per ring slot group (A/B builds)
#endif

// LDS of an update workgroup of the two-mover kernel without the entries themselves (which are
// dynamic shared memory when they are staged at all): a struct, so that pipe_multi_kernel - this
// body as one role of a launch - can lay its other role's windows over the same bytes.
template <int kProd, int kG>
struct UpdatePairLds {
  float reward_list[256];
  float discounts[16];
  __attribute__((aligned(16))) int8_t staged[2][(kChunk / 2) * kProd * kWave];
  __attribute__((aligned(16))) uint32_t ring[2][kG][kProd * kWave];
};

// The body of update_pair_kernel.  `wg`: which kProd * 64 environments this workgroup owns.
// kTagged (one-launch rollouts): a tagged copy of each mover's plane of the trace beside it, planes
// `trace_plane` entries apart like the trace's own (update_table_body's `tagged`, `tag`).
template <bool kLdsEntries, int kProd, int kCons, bool kTagged = false>
__device__ __forceinline__ void update_pair_body(
    UpdatePairLds<kProd, CAMPX_PAIR_GROUP>& L, uint32_t* lds_entries, uint32_t wg, const PairParams& pp,
    const CampxState& st, const int8_t* __restrict__ actions, const CampxOutputs& out, int64_t B, int32_t T,
    int32_t reset_first, int64_t trace_plane, const FrameCodec& fc, uint16_t* tagged = nullptr,
    uint32_t tag = 0) {
  constexpr int kLoad = update_loaders(kProd), kG = CAMPX_PAIR_GROUP;
  constexpr int E = kProd * kWave, CL = kCons * kWave, kThreads = (kProd + kCons + kLoad) * kWave;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const bool producer = wave < kProd, loader = wave >= kProd + kCons;
  const int llane = (int)threadIdx.x - (kProd + kCons) * kWave;  // loaders: 0 .. 64*kLoad-1
  const int W = pp.cols, HW = pp.rows * pp.cols;
  const int64_t env0 = (int64_t)wg * E;

  const float* g_rewards = static_cast<const float*>(st.pair_table);
  const uint32_t* g_entries = reinterpret_cast<const uint32_t*>(g_rewards + 256);
  const int n_entries = HW * HW * CAMPX_N_ACTIONS;
  if (kLdsEntries)
    for (int i = threadIdx.x; i < n_entries; i += kThreads) lds_entries[i] = g_entries[i];
  for (int i = threadIdx.x; i < 256; i += kThreads) L.reward_list[i] = g_rewards[i];
  if (threadIdx.x < 16) L.discounts[threadIdx.x] = fc.discounts[threadIdx.x];
  ActionLoader<E, kLoad> ld;
  int bad = 0;
  if (loader && T > 0) {
    ld.issue(actions, B, T, 0, env0, llane);
    bad += ld.land(L.staged[0], actions, B, T, 0, env0, llane);
  }

  const int le = wave * kWave + lane;
  const int64_t env = env0 + le;
  const bool live = producer && env < B;
  const uint32_t init0 = (uint32_t)(pp.row0[0] * W + pp.col0[0]);
  const uint32_t init1 = (uint32_t)(pp.row0[1] * W + pp.col0[1]);
  uint32_t c0 = init0, c1 = init1;
  int over = 0;
  float ret = 0.0f;
  if (live && !reset_first) {
    c0 = (uint32_t)((int)st.pos[env] * W + (int)st.pos[B + env]);
    c1 = (uint32_t)((int)st.pos[2 * B + env] * W + (int)st.pos[3 * B + env]);
    over = st.done[env];
    if (st.ret) ret = st.ret[env];
  }
  const int clane = (int)threadIdx.x - kProd * kWave;
  constexpr int kGroupsPerChunk = kChunk / kG;
  __syncthreads();

  const int n_groups = (T + kG - 1) / kG;
  // Each kind of wave runs its own loop (one s_barrier per group in each, so the counts
  // agree): registers are then allocated per role, and the loads a loader keeps in flight
  // across iterations do not take registers from the other two.
  if (producer) {
    for (int g = 0; g <= n_groups; ++g) {
        if (g < n_groups) {
          const int t0 = g * kG;
          const int8_t* chunk = L.staged[(t0 / kChunk) & 1];
          const int n = (T - t0 < kG) ? T - t0 : kG;
          uint32_t act[kG];
  #pragma unroll
          for (int j = 0; j < kG; ++j) act[j] = staged_action<E>(chunk, t0 & (kChunk - 1), j, le);
          // (whole groups run without the per-frame test, like update_table_kernel's: no scalar
          // branch between frames, so a frame's reward lookup overlaps the next frame's chain read)
          auto frames = [&](auto full_tag) {
            constexpr bool kFull = decltype(full_tag)::value;
  #pragma unroll
            for (int j = 0; j < kG; ++j) {
              if (kFull || j < n) {
                c0 = over ? init0 : c0;   // rebuilt from the art before its next action
                c1 = over ? init1 : c1;
                const uint32_t idx = pair_index(c0, c1, HW) + act[j];
                const uint32_t e = kLdsEntries ? lds_entries[idx] : g_entries[idx];   // the chain
                c0 = e & 0x7fu;
                c1 = (e >> 7) & 0x7fu;
                L.ring[g & 1][j][le] = e;
                ret = (over ? 0.0f : ret) + real_reward(L.reward_list[(e >> 19) & 0xffu]);
                over = (int)((e >> 16) & 1u);
              }
            }
          };
          if (n == kG) frames(std::true_type{});
          else frames(std::false_type{});
        }
      
      __syncthreads();
    }
  } else if (!loader) {
    // rows of the output streams are P elements apart; Bv: how much of a row may be written
    // (a padded pitch lets the batch's last group be stored whole, into the pad)
    const int64_t P = row_pitch(out, B), Bv = row_extent(out, B);
    for (int g = 0; g <= n_groups; ++g) {
        if (g > 0) {
          const int gp = g - 1, t0 = gp * kG, rb = gp & 1;
          const int n = (T - t0 < kG) ? T - t0 : kG;
          const int64_t plane = trace_plane;  // from one moving thing's trace to the next's
          constexpr int QA = E / 4, kItA = (kG * QA + CL - 1) / CL;
#ifndef CAMPX_PAIR_UNROLL_A
#define CAMPX_PAIR_UNROLL_A 1
#endif
  #pragma unroll CAMPX_PAIR_UNROLL_A   // (fully unrolled, the iterations' lookups pile up in registers and spill)
          for (int it = 0; it < kItA; ++it) {
            const int item = clane + it * CL;
            const int j = item / QA, q = item % QA;
            const int64_t e0 = env0 + 4 * q;
            if (j < n && e0 < B) {
              const u32x4 e4 = *reinterpret_cast<const u32x4*>(&L.ring[rb][j][4 * q]);
              const uint32_t e[4] = {e4.x, e4.y, e4.z, e4.w};
              uint32_t rw[4], dc[4];
  #pragma unroll
              for (int i = 0; i < 4; ++i) {
                rw[i] = __float_as_uint(L.reward_list[(e[i] >> 19) & 0xffu]);
                dc[i] = ((e[i] >> 16) & 1u) ? 0u : 0x3f800000u;
                if (fc.has_dcodes) dc[i] = discount_bits(L.discounts, dcode_pair(e[i]), (e[i] >> 16) & 1u);
              }
              const int64_t at = (int64_t)(t0 + j) * P + e0;
              if (e0 + 4 <= Bv) {   // (dword-aligned; 16-byte aligned when the row pitch is a multiple of 4)
                if (out.reward) {
                  const u32x4 r4 = {rw[0], rw[1], rw[2], rw[3]};
                  store16_update(out.reward + at, r4);
                }
                if (out.discount) {
                  const u32x4 d4 = {dc[0], dc[1], dc[2], dc[3]};
                  store16_update(out.discount + at, d4);
                }
              } else {
                for (int i = 0; i < 4 && e0 + i < B; ++i) {
                  if (out.reward) out.reward[at + i] = __uint_as_float(rw[i]);
                  if (out.discount) out.discount[at + i] = __uint_as_float(dc[i]);
                }
              }
            }
          }
          constexpr int QB = E / 16, kItB = (kG * QB + CL - 1) / CL;
  #pragma unroll
          for (int it = 0; it < kItB; ++it) {
            const int item = clane + it * CL;
            const int j = item / QB, q = item % QB;
            const int64_t e0 = env0 + 16 * q;
            if (item < kG * QB && j < n && e0 < B) {
              uint32_t ta[4], tb[4], dn[4], pf[4];
  #pragma unroll
              for (int k = 0; k < 4; ++k) {
                const u32x4 e4 = *reinterpret_cast<const u32x4*>(&L.ring[rb][j][16 * q + 4 * k]);
                const uint32_t e[4] = {e4.x, e4.y, e4.z, e4.w};
                uint32_t a[4], b[4], d[4], p[4];
  #pragma unroll
                for (int i = 0; i < 4; ++i) {
                  a[i] = (e[i] & 0x7fu) | (((e[i] >> 14) & 1u) << 7);
                  b[i] = ((e[i] >> 7) & 0x7fu) | (((e[i] >> 15) & 1u) << 7);
                  d[i] = (e[i] >> 16) & 1u;
                  p[i] = perf_byte(fc, perf_code_pair(e[i]));
                }
                ta[k] = pack4(a[0], a[1], a[2], a[3]);
                tb[k] = pack4(b[0], b[1], b[2], b[3]);
                dn[k] = pack4(d[0], d[1], d[2], d[3]);
                pf[k] = pack4(p[0], p[1], p[2], p[3]);
              }
              const int64_t at = (int64_t)(t0 + j) * P + e0;
              if (e0 + 16 <= Bv) {  // (aligned when the row pitch is a multiple of 16; else legal, slower)
                const u32x4 a4 = {ta[0], ta[1], ta[2], ta[3]}, b4 = {tb[0], tb[1], tb[2], tb[3]};
                store16_update(out.trace + at, a4);
                store16_update(out.trace + plane + at, b4);
                if (kTagged) {
                  store_tagged_row(tagged + at, ta, tag);
                  store_tagged_row(tagged + plane + at, tb, tag);
                }
                if (out.done) {
                  const u32x4 d4 = {dn[0], dn[1], dn[2], dn[3]};
                  store16_update(out.done + at, d4);
                }
                if (out.perf) {
                  const u32x4 p4 = {pf[0], pf[1], pf[2], pf[3]};
                  store16_update(out.perf + at, p4);
                }
              } else {
                for (int i = 0; i < 16 && e0 + i < B; ++i) {
                  const int sh = (i & 3) * 8;
                  out.trace[at + i] = (uint8_t)(ta[i >> 2] >> sh);
                  out.trace[plane + at + i] = (uint8_t)(tb[i >> 2] >> sh);
                  if (kTagged) {
                    store_tagged_one(tagged + at + i, ta[i >> 2] >> sh, tag);
                    store_tagged_one(tagged + plane + at + i, tb[i >> 2] >> sh, tag);
                  }
                  if (out.done) out.done[at + i] = (uint8_t)(dn[i >> 2] >> sh);
                  if (out.perf) out.perf[at + i] = (int8_t)(pf[i >> 2] >> sh);
                }
              }
            }
          }
        }
      
      __syncthreads();
    }
  } else {
    for (int g = 0; g <= n_groups; ++g) {
        const int c = g / kGroupsPerChunk, phase = g % kGroupsPerChunk;
        const int t_next = (c + 1) * kChunk;
        if (t_next < T) {
          if (phase == 0) ld.issue(actions, B, T, t_next, env0, llane);
          if (phase == kGroupsPerChunk - 1)
            bad += ld.land(L.staged[(c + 1) & 1], actions, B, T, t_next, env0, llane);
        }
      
      __syncthreads();
    }
  }

  if (live) {
    st.pos[env] = (int8_t)(c0 / (uint32_t)W);
    st.pos[B + env] = (int8_t)(c0 % (uint32_t)W);
    st.pos[2 * B + env] = (int8_t)(c1 / (uint32_t)W);
    st.pos[3 * B + env] = (int8_t)(c1 % (uint32_t)W);
    st.done[env] = (uint8_t)over;
    if (st.ret) st.ret[env] = ret;
  }
  report_bad_actions(out, bad);
}

template <bool kLdsEntries, int kProd, int kCons>
__global__ __launch_bounds__((kProd + kCons + update_loaders(kProd)) * kWave,
                             update_min_waves(kProd, kCons)) void update_pair_kernel(
    PairParams pp, const CampxSpec* __restrict__ spec, CampxState st,
    const int8_t* __restrict__ actions, CampxOutputs out, int64_t B, int32_t T,
    int32_t reset_first, int64_t trace_plane, FrameCodec fc) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_entries[];  // kLdsEntries: n_entries
  __shared__ UpdatePairLds<kProd, CAMPX_PAIR_GROUP> L;
  update_pair_body<kLdsEntries, kProd, kCons>(L, lds_entries, tile_of_block(blockIdx.x, gridDim.x, CAMPX_UPD_XCD),
                                              pp, st, actions, out, B, T, reset_first, trace_plane, fc);
}

// ---------------------------------------------------------------------------
// Three- and four-mover games: the same producer / consumer / loader layout over a
// direct-indexed (cell, cell, cell[, cell], action) table in GLOBAL memory
// (campx_tuple_table_build: 4.4 MB for three movers on a 6x8 board, 212 MB for four).
// 64-bit entries: bits 0-27 the things' cells after the frame (7 bits each), 28-31
// whether each is the character its cell shows, 32 done, 33-34 perf + 1, 35-42 index into
// the reward list.  The dependent chain is one global load per frame, so the kernel wants
// every environment in flight at once: 8-frame groups keep the ring small enough for two
// 256-environment workgroups per CU.
// Four waves per SIMD = two workgroups per CU, so that 131 072 environments are all in
// flight at once (four movers: 131 -> 128 VGPRs; sokoban level 2 87 -> 65 us per launch).
#ifndef CAMPX_TUPLE_MINWAVES
#define CAMPX_TUPLE_MINWAVES 4
#endif

template <int kProd>
struct UpdateTupleLds {
  float reward_list[256];
  float discounts[16];
  __attribute__((aligned(16))) int8_t staged[2][(kChunk / 2) * kProd * kWave];
  __attribute__((aligned(16))) uint64_t ring[2][kTupleGroup][kProd * kWave];
};

// The body of update_tuple_kernel.  `wg`: which kProd * 64 environments this workgroup owns.
template <int K, int kProd, int kCons, bool kTagged = false>
__device__ __forceinline__ void update_tuple_body(
    UpdateTupleLds<kProd>& L, const uint32_t wg, const TupleParams tp, const CampxState st,
    const int8_t* __restrict__ actions, const CampxOutputs out, const int64_t B, const int32_t T,
    const int32_t reset_first, const int64_t trace_plane, const FrameCodec fc, uint16_t* const tagged = nullptr,
    const uint32_t tag = 0) {
  constexpr int kLoad = update_loaders(kProd), kG = kTupleGroup;
  constexpr int E = kProd * kWave, CL = kCons * kWave, kThreads = (kProd + kCons + kLoad) * kWave;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const bool producer = wave < kProd, loader = wave >= kProd + kCons;
  const int llane = (int)threadIdx.x - (kProd + kCons) * kWave;
  const int W = tp.cols;
  const uint32_t HW = (uint32_t)(tp.rows * tp.cols);
  const int64_t env0 = (int64_t)wg * E;
  const float* g_rewards = static_cast<const float*>(st.pair_table);
  const uint64_t* g_entries = reinterpret_cast<const uint64_t*>(g_rewards + 256);
  for (int i = threadIdx.x; i < 256; i += kThreads) L.reward_list[i] = g_rewards[i];
  if (threadIdx.x < 16) L.discounts[threadIdx.x] = fc.discounts[threadIdx.x];
  ActionLoader<E, kLoad> ld;
  int bad = 0;
  if (loader && T > 0) {
    ld.issue(actions, B, T, 0, env0, llane);
    bad += ld.land(L.staged[0], actions, B, T, 0, env0, llane);
  }
  const int le = wave * kWave + lane;
  const int64_t env = env0 + le;
  const bool live = producer && env < B;
  uint32_t init = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) init |= (uint32_t)(tp.row0[k] * W + tp.col0[k]) << (7 * k);
  uint32_t cells = init;
  int over = 0;
  float ret = 0.0f;
  if (live && !reset_first) {
    cells = 0;
    // (rolled: with the 2K addresses of an unrolled loop live at once the four-mover kernel,
    // capped at 128 VGPRs for two workgroups per CU, spilled 36 bytes here)
#pragma unroll 1
    for (int k = 0; k < K; ++k)
      cells |= (uint32_t)((int)st.pos[(int64_t)(2 * k) * B + env] * W +
                          (int)st.pos[(int64_t)(2 * k + 1) * B + env]) << (7 * k);
    over = st.done[env];
    if (st.ret) ret = st.ret[env];
  }
  const int clane = (int)threadIdx.x - kProd * kWave;
  constexpr int kGroupsPerChunk = kChunk / kG;
  __syncthreads();

  const int n_groups = (T + kG - 1) / kG;
  if (producer) {
    for (int g = 0; g <= n_groups; ++g) {
      if (g < n_groups) {
        const int t0 = g * kG;
        const int8_t* chunk = L.staged[(t0 / kChunk) & 1];
        const int n = (T - t0 < kG) ? T - t0 : kG;
        uint32_t act[kG];
#pragma unroll
        for (int j = 0; j < kG; ++j) act[j] = staged_action<E>(chunk, t0 & (kChunk - 1), j, le);
        auto frames = [&](auto full_tag) {      // (whole groups: no per-frame test, see above)
          constexpr bool kFull = decltype(full_tag)::value;
#pragma unroll
          for (int j = 0; j < kG; ++j) {
            if (kFull || j < n) {
              cells = over ? init : cells;  // rebuilt from the art before its next action
              const uint64_t e = g_entries[tuple_index<K>(cells, HW) + act[j]];  // the chain
              cells = (uint32_t)e & 0x0fffffffu;
              L.ring[g & 1][j][le] = e;
              ret = (over ? 0.0f : ret) + real_reward(L.reward_list[(uint32_t)(e >> 35) & 0xffu]);
              over = (int)((e >> 32) & 1u);
            }
          }
        };
        if (n == kG) frames(std::true_type{});
        else frames(std::false_type{});
      }
      __syncthreads();
    }
  } else if (!loader) {
    // rows of the output streams are P elements apart; Bv: how much of a row may be written
    // (a padded pitch lets the batch's last group be stored whole, into the pad)
    const int64_t P = row_pitch(out, B), Bv = row_extent(out, B);
    for (int g = 0; g <= n_groups; ++g) {
      if (g > 0) {
        const int gp = g - 1, t0 = gp * kG, rb = gp & 1;
        const int n = (T - t0 < kG) ? T - t0 : kG;
        const int64_t plane = trace_plane;  // from one moving thing's trace to the next's
        // ---- float streams: item = (frame j, 4 environments)
        constexpr int QA = E / 4, kItA = (kG * QA + CL - 1) / CL;
#pragma unroll 1
        for (int it = 0; it < kItA; ++it) {
          const int item = clane + it * CL;
          const int j = item / QA, q = item % QA;
          const int64_t e0 = env0 + 4 * q;
          if (item < kG * QA && j < n && e0 < B) {
            uint32_t rw[4], dc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const uint32_t hi = (uint32_t)(L.ring[rb][j][4 * q + i] >> 32);
              rw[i] = __float_as_uint(L.reward_list[(hi >> 3) & 0xffu]);
              dc[i] = (hi & 1u) ? 0u : 0x3f800000u;
              if (fc.has_dcodes) dc[i] = discount_bits(L.discounts, dcode_tuple(hi), hi & 1u);
            }
            const int64_t at = (int64_t)(t0 + j) * P + e0;
            if (e0 + 4 <= Bv) {   // (dword-aligned; 16-byte aligned when the row pitch is a multiple of 4)
              if (out.reward) {
                const u32x4 r4 = {rw[0], rw[1], rw[2], rw[3]};
                store16_update(out.reward + at, r4);
              }
              if (out.discount) {
                const u32x4 d4 = {dc[0], dc[1], dc[2], dc[3]};
                store16_update(out.discount + at, d4);
              }
            } else {
              for (int i = 0; i < 4 && e0 + i < B; ++i) {
                if (out.reward) out.reward[at + i] = __uint_as_float(rw[i]);
                if (out.discount) out.discount[at + i] = __uint_as_float(dc[i]);
              }
            }
          }
        }
        // ---- byte streams: item = (frame j, 16 environments)
        constexpr int QB = E / 16, kItB = (kG * QB + CL - 1) / CL;
#pragma unroll 1
        for (int it = 0; it < kItB; ++it) {
          const int item = clane + it * CL;
          const int j = item / QB, q = item % QB;
          const int64_t e0 = env0 + 16 * q;
          if (item < kG * QB && j < n && e0 < B) {
            uint32_t tr[K][4], dn[4], pf[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              uint32_t lo[4], hi[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const uint64_t e = L.ring[rb][j][16 * q + 4 * w + i];
                lo[i] = (uint32_t)e;
                hi[i] = (uint32_t)(e >> 32);
              }
#pragma unroll
              for (int k = 0; k < K; ++k) {
                uint32_t b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                  b[i] = ((lo[i] >> (7 * k)) & 0x7fu) | (((lo[i] >> (28 + k)) & 1u) << 7);
                tr[k][w] = pack4(b[0], b[1], b[2], b[3]);
              }
              dn[w] = pack4(hi[0] & 1u, hi[1] & 1u, hi[2] & 1u, hi[3] & 1u);
              pf[w] = pack4(perf_byte(fc, perf_code_tuple(hi[0])), perf_byte(fc, perf_code_tuple(hi[1])),
                            perf_byte(fc, perf_code_tuple(hi[2])), perf_byte(fc, perf_code_tuple(hi[3])));
            }
            const int64_t at = (int64_t)(t0 + j) * P + e0;
            if (e0 + 16 <= Bv) {  // (aligned when the row pitch is a multiple of 16; else legal, slower)
#pragma unroll
              for (int k = 0; k < K; ++k) {
                const u32x4 t4 = {tr[k][0], tr[k][1], tr[k][2], tr[k][3]};
                store16_update(out.trace + k * plane + at, t4);
              }
              if (out.done) {
                const u32x4 d4 = {dn[0], dn[1], dn[2], dn[3]};
                store16_update(out.done + at, d4);
              }
              if (out.perf) {
                const u32x4 p4 = {pf[0], pf[1], pf[2], pf[3]};
                store16_update(out.perf + at, p4);
              }
              if (kTagged) {
                // the tagged copy in a pass of its own, one mover at a time from the ring again (rolled:
                // beside the K planes above, the entries' registers spilled - 11 / 28 VGPRs for three /
                // four movers under the 128 of two workgroups per CU)
#pragma unroll 1
                for (int k = 0; k < K; ++k) {
                  uint32_t tk[4];
#pragma unroll
                  for (int w = 0; w < 4; ++w) {
                    uint32_t b[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                      const uint32_t lo = (uint32_t)L.ring[rb][j][16 * q + 4 * w + i];
                      b[i] = ((lo >> (7 * k)) & 0x7fu) | (((lo >> (28 + k)) & 1u) << 7);
                    }
                    tk[w] = pack4(b[0], b[1], b[2], b[3]);
                  }
                  store_tagged_row(tagged + k * plane + at, tk, tag);
                }
              }
            } else {
              for (int i = 0; i < 16 && e0 + i < B; ++i) {
                const int sh = (i & 3) * 8;
#pragma unroll
                for (int k = 0; k < K; ++k) out.trace[k * plane + at + i] = (uint8_t)(tr[k][i >> 2] >> sh);
                if (kTagged) {
#pragma unroll
                  for (int k = 0; k < K; ++k) store_tagged_one(tagged + k * plane + at + i, tr[k][i >> 2] >> sh, tag);
                }
                if (out.done) out.done[at + i] = (uint8_t)(dn[i >> 2] >> sh);
                if (out.perf) out.perf[at + i] = (int8_t)(pf[i >> 2] >> sh);
              }
            }
          }
        }
      }
      __syncthreads();
    }
  } else {
    for (int g = 0; g <= n_groups; ++g) {
      const int c = g / kGroupsPerChunk, phase = g % kGroupsPerChunk;
      const int t_next = (c + 1) * kChunk;
      if (t_next < T) {
        if (phase == 0) ld.issue(actions, B, T, t_next, env0, llane);
        if (phase == kGroupsPerChunk - 1)
          bad += ld.land(L.staged[(c + 1) & 1], actions, B, T, t_next, env0, llane);
      }
      __syncthreads();
    }
  }

  if (live) {
    // (the environment's index is derived again rather than kept in two VGPRs across the frame
    // loop: the empty asm hides `le` from common-subexpression elimination; the four-mover
    // kernel, capped at 128 VGPRs, otherwise spills it)
    int le_again = le;
    asm volatile("" : "+v"(le_again));
    const int64_t env_again = env0 + le_again;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const uint32_t c = (cells >> (7 * k)) & 0x7fu;
      st.pos[(int64_t)(2 * k) * B + env_again] = (int8_t)(c / (uint32_t)W);
      st.pos[(int64_t)(2 * k + 1) * B + env_again] = (int8_t)(c % (uint32_t)W);
    }
    st.done[env_again] = (uint8_t)over;
    if (st.ret) st.ret[env_again] = ret;
  }
  report_bad_actions(out, bad);
}

template <int K, int kProd, int kCons>
__global__ __launch_bounds__((kProd + kCons + update_loaders(kProd)) * kWave,
                             CAMPX_TUPLE_MINWAVES) void update_tuple_kernel(
    TupleParams tp, CampxState st, const int8_t* __restrict__ actions, CampxOutputs out, int64_t B,
    int32_t T, int32_t reset_first, int64_t trace_plane, FrameCodec fc) {
  __shared__ UpdateTupleLds<kProd> L;
  update_tuple_body<K, kProd, kCons>(L, blockIdx.x, tp, st, actions, out, B, T, reset_first, trace_plane, fc);
}

constexpr int kBigEnvs = 8 * kWave;   // environments of a "big" update workgroup

// Number of big workgroups from which launch_update prefers them: one per CU of the chip
// (setting big_wgs overrides; a huge value turns them off).
int64_t knob_big_workgroups() {
  const int64_t forced = knob(K_BIG_WGS);
  if (forced >= 0) return forced;
  // the current device's CU count, asked once per device (a process may drive several)
  static std::mutex lock;
  static std::map<int, int64_t> cus_of;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  std::lock_guard<std::mutex> guard(lock);
  const auto it = cus_of.find(dev);
  if (it != cus_of.end()) return it->second;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  cus_of[dev] = cus;
  return (int64_t)cus;
}

PairParams make_pair_params(const CampxSpec& s) {
  PairParams pp;
  memset(&pp, 0, sizeof(pp));
  pp.rows = s.rows;
  pp.cols = s.cols;
  pp.n_layers = s.n_layers;
  for (int d = 0; d < 2; ++d) {
    pp.dyn_layer[d] = s.dyn_layer[d];
    pp.row0[d] = s.dyn_row0[d];
    pp.col0[d] = s.dyn_col0[d];
  }
  return pp;
}

// Shape of the update kernels' workgroups: producer and consumer waves (A/B builds can
// override).  Measured, whole rollout launch at the BASELINE sizes (gpurun_out/r2a):
// one-mover table kernel, boat race: (1,1) 0.2006, (1,3) 0.2112, (2,2) 0.2034,
// (2,4) 0.2005, (4,4) 0.1946 ms - the 256-environment workgroup writes 1 KiB / 256 B row
// pieces instead of 512 / 128 B.  With the final kernels (gpurun_out/r2y, kernel time in
// us): table kernel, boat race / wall world: (4,4) 16.4 / 73, (2,2) 17.1 / 66.5,
// (4,2) 21.3 / 63; pair kernel, sokoban: (4,4) 44-50, (4,2) 42, (2,2) 62, (2,1) 57.
#ifndef CAMPX_UPD_PROD
#define CAMPX_UPD_PROD 4
#endif
// (8 consumer waves for the 256-environment workgroups: the consumers, not the producers'
// chain, bound the kernel per frame - see the note above update_table_kernel: B = 4 096
// 14.9 -> 12.9 us, T = 256 29.0 -> 23.5, B = 65 536 unchanged)
#ifndef CAMPX_UPD_CONS
#define CAMPX_UPD_CONS 8
#endif
#ifndef CAMPX_UPD_GROUP
// frames per group of the 256-environment one-mover workgroups; kernel us per 100 frames at
// B = 4 096 / 65 536 (gpurun_out/t20-t21): 4: 18.1 / 21.4, 8: 14.8 / 17.8, 16: 12.4 / 15.6,
// 32: 12.7 / 16.8 - a group costs ~0.3 us of hand-over, a longer one more fill and drain
#define CAMPX_UPD_GROUP 16
#endif
#ifndef CAMPX_PAIR_BIG
#define CAMPX_PAIR_BIG 1     // 512-environment workgroups (8 producer, 4 consumer waves) for large batches
#endif
#ifndef CAMPX_PAIR_PROD
#define CAMPX_PAIR_PROD 4
#endif
// (6 consumer waves for the 256-environment pair workgroups, 2 until round 3: their consumers
// bound the kernel - sokoban B = 16 384: 31.3 -> 18.7 us per 100 frames (8 waves: 18.0), B = 4 096
// T = 256: 68.2 -> 35.8.  The three- / four-mover kernel keeps 2: 4 or 6 read 34-36 instead of
// 38-40 us at B = 4 096 but 78-93 instead of 60-71 us at B = 131 072, where the extra waves
// cost resident workgroups.)
#ifndef CAMPX_PAIR_CONS
#define CAMPX_PAIR_CONS 6
#endif
#ifndef CAMPX_TUPLE_PROD
#define CAMPX_TUPLE_PROD 4
#endif
#ifndef CAMPX_TUPLE_CONS
#define CAMPX_TUPLE_CONS 2
#endif

// `trace_plane`: rows (environments) from one moving thing's plane of the trace to the
// next's - B times the frames the whole trace holds, which is more than T when the caller
// runs a launch in chunks.
int32_t launch_update(const CampxSpec& s, const CampxSpec* spec_dev, CampxState st,
                     const int8_t* actions, CampxOutputs out, int64_t B, int32_t T,
                     int32_t reset_first, bool use_table, int64_t trace_plane,
                     hipStream_t stream) {
  // 512-environment workgroups (twice the row piece per store) once there are enough
  // environments to give every CU one; 256-environment workgroups below that
  const bool big = B >= (int64_t)kBigEnvs * knob_big_workgroups();
  const FrameCodec fc = make_codec(s);
  if (use_table) {
    const MoverParams mp = {s.rows, s.cols, s.n_layers, s.dyn_layer[0], s.dyn_z[0],
                            s.dyn_row0[0], s.dyn_col0[0]};
    if (big) {
      constexpr int kProd = 8, kCons = 4;
      const dim3 grid((unsigned)((B + kBigEnvs - 1) / kBigEnvs)),
          block((kProd + kCons + update_loaders(kProd)) * kWave);
      hipLaunchKernelGGL((update_table_kernel<kProd, kCons, 8>), grid, block, 0, stream, mp,
                         spec_dev, st, actions, out, B, T, reset_first, fc);
    } else {
      constexpr int kProd = CAMPX_UPD_PROD, kCons = CAMPX_UPD_CONS, kEnvs = kProd * kWave;
      const dim3 grid((unsigned)((B + kEnvs - 1) / kEnvs)),
          block((kProd + kCons + update_loaders(kProd)) * kWave);
      hipLaunchKernelGGL((update_table_kernel<kProd, kCons, CAMPX_UPD_GROUP>), grid, block, 0, stream, mp,
                         spec_dev, st, actions, out, B, T, reset_first, fc);
    }
  } else if (s.n_dyn == 2 && st.pair_table) {
    const PairParams pp = make_pair_params(s);
    const int n_entries = s.rows * s.cols * s.rows * s.cols * CAMPX_N_ACTIONS;
    // the entries in LDS when they fit (CAMPX_PAIR_MODE=0: read them through L1/L2 anyway)
    const bool in_lds = n_entries <= kPairLdsEntries;
    const size_t shmem = in_lds ? (((size_t)n_entries * sizeof(uint32_t) + 15) & ~(size_t)15) : 0;
#define CAMPX_PAIR_LAUNCH(PROD, CONS)                                                          \
  do {                                                                                         \
    const dim3 grid((unsigned)((B + (PROD) * kWave - 1) / ((PROD) * kWave))),                  \
        block(((PROD) + (CONS) + update_loaders(PROD)) * kWave);                               \
    if (in_lds)                                                                                \
      hipLaunchKernelGGL((update_pair_kernel<true, PROD, CONS>), grid, block, shmem, stream,   \
                         pp, spec_dev, st, actions, out, B, T, reset_first, trace_plane, fc);  \
    else                                                                                       \
      hipLaunchKernelGGL((update_pair_kernel<false, PROD, CONS>), grid, block, 0, stream, pp,  \
                         spec_dev, st, actions, out, B, T, reset_first, trace_plane, fc);      \
  } while (0)
    // The 256-environment pair workgroup (12 waves of ~110 VGPRs) does not share a CU with a
    // second one, so past one workgroup per CU a launch runs in two rounds (B = 99 999: 44 us
    // against 21 at 65 536); the 512-environment workgroups take over from there (one round,
    // ~38 us), not only from 512 environments per CU.
    const bool big_pair = B > (int64_t)(kBigEnvs / 2) * knob_big_workgroups();
    if ((big || big_pair) && CAMPX_PAIR_BIG)
      CAMPX_PAIR_LAUNCH(8, 4);
    else
      CAMPX_PAIR_LAUNCH(CAMPX_PAIR_PROD, CAMPX_PAIR_CONS);
#undef CAMPX_PAIR_LAUNCH
  } else if (s.n_dyn >= 3 && st.pair_table) {
    constexpr int kProd = CAMPX_TUPLE_PROD, kCons = CAMPX_TUPLE_CONS, kEnvs = kProd * kWave;
    const dim3 grid((unsigned)((B + kEnvs - 1) / kEnvs)),
        block((kProd + kCons + update_loaders(kProd)) * kWave);
    const TupleParams tp = make_tuple_params(s);
    if (s.n_dyn == 3)
      hipLaunchKernelGGL((update_tuple_kernel<3, kProd, kCons>), grid, block, 0, stream, tp, st,
                         actions, out, B, T, reset_first, trace_plane, fc);
    else
      hipLaunchKernelGGL((update_tuple_kernel<4, kProd, kCons>), grid, block, 0, stream, tp, st,
                         actions, out, B, T, reset_first, trace_plane, fc);
  } else {
    if (s.table_only) return CAMPX_ESPEC;   // a host-tabulated game without its table
    launch_trace(s, spec_dev, st, actions, out, B, T, reset_first, trace_plane, stream);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_failed(e);
  return CAMPX_OK;
}


// What a render workgroup of the shared launches below needs (pipe_table_kernel).
struct PipeRender {
  uint32_t R, m, sh1, sh2, slab_bytes, shift_base, shift_slab;
  int32_t cells, dyn_off;
  int64_t pitch;               // rows of the trace from one frame to the next
  const int8_t* rot;
  const uint8_t* top_layer;
  const uint8_t* trace;
  int8_t* dst;
  int32_t T;
  uint32_t per_frame;          // render workgroups per frame (a multiple of 8)
  uint32_t U;                  // update workgroups
  uint16_t* tagged;            // one-launch rollouts: the trace's tagged copy [T, pitch], and
  uint32_t tag;                // this launch's tag (1..255)
  int32_t dyn_offs[CAMPX_MAX_DYN];   // games of two to four movers: each one's layer offset in a row,
  int64_t plane;               // and the rows from one mover's plane of the trace to the next's
  uint32_t max_naps;           // looks at stale entries before a render wave gives up - loudly
  uint32_t debug_delay;        // tests only: s_sleep rounds in front of the update role
  int32_t* error_flag;         // CampxOutputs.error_flag
};

// (Round 4's first attempt at one rollout in one launch - overlap_table_kernel: persistent 14-wave
// workgroups, tickets, per-workgroup progress words polled by the render role - measured slower
// than two launches at every batch size, profiles/r04_overlap_ab.txt, and was removed once the
// tagged-trace form below had replaced it; NOTES.md R4.4, R4.12.)

// ---------------------------------------------------------------------------
// Rollouts pipelined ACROSS calls (round 4): one launch holds the update pass of rollout i + 1
// AND the render pass of rollout i.  The two have nothing to do with each other - the render
// role reads the trace the PREVIOUS launch left complete in memory, the update role writes
// another trace buffer - so nothing polls and nothing
// persists: workgroups [0, U) are update workgroups (dispatched first: they are the latency
// chain), every later one renders one 2 KiB window per wave of one frame and leaves, like a
// render_kernel block.  What a caller pays for it: the observations of a rollout are complete
// only after the NEXT call (or FusedGame.flush()), and the next rollout's actions must be known
// when this one's observations are asked for - open-loop action streams (bench.py --deferred).
// Scope: one-mover table games, int8 observations of every frame,
// frames of whole 16-byte chunks; anything else runs the two passes one after the other.
#ifndef CAMPX_PIPE_WIN
#define CAMPX_PIPE_WIN 2
#endif
constexpr int kPipeWin = CAMPX_PIPE_WIN;                   // KiB per render wave
constexpr uint32_t kPipeSpan = 1024u * kPipeWin;
#ifndef CAMPX_PIPE_PROD
#define CAMPX_PIPE_PROD 1
#endif
#ifndef CAMPX_PIPE_CONS
#define CAMPX_PIPE_CONS 2
#endif
// Small workgroups - one producer wave (64 environments), two consumer waves, one loader: what
// the render role needs is many short waves per CU that come and go one by one (as 14-wave
// workgroups of the update kernel's own shape it streamed at two thirds of render_kernel's rate)
#ifndef CAMPX_FLOW_NAP          // s_sleep units (64 clocks) between two looks at stale entries:
#define CAMPX_FLOW_NAP 8        // the first four times,
#endif
#ifndef CAMPX_FLOW_NAP_LONG
#define CAMPX_FLOW_NAP_LONG 32  // and from then on
#endif
#ifndef CAMPX_PIPE_GROUP
#define CAMPX_PIPE_GROUP 16
#endif
// A render wave that gives up says so: CAMPX_ERR_FLOW_TIMEOUT in the caller's error word (system
// scope: the word may be host memory), once per wave; the frames it then writes are wrong.
__device__ __forceinline__ void flow_gave_up(int32_t* error_flag) {
  if (error_flag && (threadIdx.x & 63u) == 0u)
    __hip_atomic_fetch_or(error_flag, CAMPX_ERR_FLOW_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
constexpr int kPipeProd = CAMPX_PIPE_PROD, kPipeCons = CAMPX_PIPE_CONS, kPipeGroup = CAMPX_PIPE_GROUP;
constexpr int kPipeWaves = kPipeProd + kPipeCons + update_loaders(kPipeProd);
constexpr int kPipeEnvs = kPipeProd * kWave;

// kFlow: the render role is THIS rollout's (launch_flow below): the update role stores a tagged
// copy of the trace, a render wave waits for its entries to carry this launch's tag.
template <bool kFlow>
__global__ __launch_bounds__(kPipeWaves * kWave) void pipe_table_kernel(
    MoverParams mp, const CampxSpec* __restrict__ spec, CampxState st,
    const int8_t* __restrict__ actions, CampxOutputs out, int64_t B, int32_t T,
    int32_t reset_first, FrameCodec fc, PipeRender rr) {
  __shared__ UpdateTableLds<kPipeProd, kPipeGroup> L;
  static_assert(sizeof(L) >= kPipeWaves * (kPipeSpan + 2 * CAMPX_MAX_CELLS), "render windows fit the update LDS");
  // Which role: the first U workgroups update.  (Spreading them among the render ones - every
  // 24th / 48th / 96th workgroup - was slower at every size tried: an update wave that shares
  // its SIMD with streaming waves walks its chain more slowly, lives longer, and more of them
  // pile up; profiles/r04_deferred_ab.txt section 5.)
  if (blockIdx.x < rr.U) {
    if (kFlow && rr.debug_delay)     // (tests: hold the update role back so that render waves time out)
      for (uint32_t i = 0; i < rr.debug_delay; ++i) __builtin_amdgcn_s_sleep(127);
    update_table_body<kPipeProd, kPipeCons, kPipeGroup, kFlow>(L, blockIdx.x, mp, spec, st, actions, out, B, T,
                                                         reset_first, fc, rr.tagged, rr.tag);
    return;
  }
  const uint32_t item = blockIdx.x - rr.U;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t t = item / rr.per_frame;
  uint32_t wx = item - t * rr.per_frame;
  wx = (wx & 7u) * (rr.per_frame >> 3) + (wx >> 3);    // per_frame is a multiple of 8: one XCD, one eighth
  const uint32_t shift = (rr.shift_base + t * rr.shift_slab) & (kPipeSpan - 1u);
  const uint32_t widx = wx * (uint32_t)kPipeWaves + wave;
  if ((uint64_t)widx * kPipeSpan >= (uint64_t)rr.slab_bytes + shift) return;
  int8_t* win0 = reinterpret_cast<int8_t*>(&L) + wave * (kPipeSpan + 2 * CAMPX_MAX_CELLS);
  uint16_t* scen_off = reinterpret_cast<uint16_t*>(win0 + kPipeSpan);
  const uint32_t R = rr.R;
  const uint32_t rot_pitch = ((R + 15u) & ~15u) + 16u;
  const uint32_t woff0 = widx * kPipeSpan - shift;
  const uint32_t wlo = widx * kPipeSpan < shift ? 0u : woff0;
  const uint32_t whi = __umulhi(rr.m, wlo);
  const uint32_t first_row = (((wlo - whi) >> rr.sh1) + whi) >> rr.sh2;
  const uint32_t wend = (woff0 + kPipeSpan - 1u < rr.slab_bytes) ? woff0 + kPipeSpan - 1u : rr.slab_bytes - 1u;
  const uint32_t ehi = __umulhi(rr.m, wend);
  const uint32_t last_row = (((wend - ehi) >> rr.sh1) + ehi) >> rr.sh2;
  const uint32_t slots = (last_row - first_row + 1u) * 2u;     // (row) x (set | clear)
  const uint8_t* frame_trace = rr.trace + (int64_t)t * rr.pitch;
  // ---- loads first: two trace bytes, the window's scenery chunks, two cells' scenery layer
  uint32_t ent[2];
  // kFlow: THIS rollout's trace, from its tagged copy: an entry is valid when it carries this
  // launch's tag (agent-scope loads of the aligned dword an entry sits in).  The first look is
  // issued here, in front of the scenery loads; whether it found this launch's tags is asked
  // only after the scenery is staged.
  const uint16_t* frame_tagged = kFlow ? rr.tagged + (int64_t)t * rr.pitch : nullptr;
  auto look = [&](uint32_t row) {     // (the raw dword: shifting it here would wait for it here)
    return __hip_atomic_load(reinterpret_cast<const uint32_t*>(frame_tagged + (row & ~1u)),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  uint32_t row0 = first_row + (lane >> 1), row1 = first_row + ((lane + kWave) >> 1);
  row0 = row0 <= last_row ? row0 : last_row;
  row1 = row1 <= last_row ? row1 : last_row;
  if (kFlow) {
    ent[0] = look(row0);
    ent[1] = look(row1);
  } else {
    ent[0] = frame_trace[row0];
    ent[1] = frame_trace[row1];
  }
  u32x4 scen[kPipeWin];
#pragma unroll
  for (int j = 0; j < kPipeWin; ++j) {
    const uint32_t off = woff0 + (uint32_t)j * 1024u + lane * 16u;
    const uint32_t hi = __umulhi(rr.m, off);
    const uint32_t row = (((off - hi) >> rr.sh1) + hi) >> rr.sh2;
    const uint32_t k = off - row * R;
    scen[j] = *reinterpret_cast<const u32x4*>(rr.rot + (k & 15u) * rot_pitch + (k & ~15u));
  }
  const uint32_t top2 = *reinterpret_cast<const uint16_t*>(rr.top_layer + 2u * lane);
#pragma unroll
  for (int j = 0; j < kPipeWin; ++j)
    *reinterpret_cast<u32x4*>(win0 + j * 1024 + lane * 16u) = scen[j];
  {
    const uint32_t c = 2u * lane;
    const uint32_t lo = (top2 & 0xffu) * (uint32_t)rr.cells + c;
    const uint32_t hi2 = (top2 >> 8) * (uint32_t)rr.cells + c + 1u;
    *reinterpret_cast<uint32_t*>(scen_off + c) = lo | (hi2 << 16);
  }
  if (kFlow) {
#ifndef CAMPX_FLOW_NOPOLL       // (=1: a TIMING experiment - never wait; results are wrong)
    // a wave whose rows are not there yet sleeps and looks again - at its own entries, so nobody
    // polls one line
    uint32_t naps = 0;
    const uint32_t sh0 = (row0 & 1u) * 16u, sh1 = (row1 & 1u) * 16u;
    while (__any(((ent[0] >> sh0) & 0xff00u) != (rr.tag << 8) || ((ent[1] >> sh1) & 0xff00u) != (rr.tag << 8))) {
      if (naps < 4u) __builtin_amdgcn_s_sleep(CAMPX_FLOW_NAP);
      else __builtin_amdgcn_s_sleep(CAMPX_FLOW_NAP_LONG);
      // (seconds of waiting: the entries will never carry this launch's tag - two launches
      // sharing one scratch block at the same time, which the header forbids.  A launch that
      // never ends helps nobody, so the wave goes on with what it has - and raises the
      // caller's error word: these frames are wrong and the host is told)
      if (++naps > rr.max_naps) {
        flow_gave_up(rr.error_flag);
        break;
      }
      ent[0] = look(row0);
      ent[1] = look(row1);
    }
    ent[0] = (ent[0] >> sh0) & 0xffu;
    ent[1] = (ent[1] >> sh1) & 0xffu;
#else
    ent[0] = (ent[0] >> ((row0 & 1u) * 16u)) & 0xffu;
    ent[1] = (ent[1] >> ((row1 & 1u) * 16u)) & 0xffu;
#endif
  }
  // ---- patches: the mover's 1, and the scenery's 1 it hides
  auto apply = [&](uint32_t sidx, uint32_t e) {
    const uint32_t r = sidx >> 1, p = sidx & 1u;
    const uint32_t cell = e & 0x7fu;
    const uint32_t byte = p ? (uint32_t)rr.dyn_off + cell : (uint32_t)scen_off[cell];
    const uint32_t at = (first_row + r) * R + byte - woff0;
    if (sidx < slots && (e >> 7) && at < kPipeSpan) win0[at] = (int8_t)p;
  };
  apply(lane, ent[0]);
  apply(lane + kWave, ent[1]);
  for (uint32_t sidx = lane + 2u * kWave; sidx < slots; sidx += kWave)     // tiny rows only
    if (kFlow) {   // (each lane waits for its own entry: rows of under 32 bytes, boards of a few cells)
      const uint16_t* at = rr.tagged + (int64_t)t * rr.pitch + first_row + (sidx >> 1);
      uint32_t e = __hip_atomic_load(at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (uint32_t naps = 0; (e >> 8) != rr.tag && naps < rr.max_naps; ++naps) {
        __builtin_amdgcn_s_sleep(CAMPX_FLOW_NAP_LONG);
        e = __hip_atomic_load(at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if ((e >> 8) != rr.tag && rr.error_flag)     // (per lane here: tiny rows only)
        __hip_atomic_fetch_or(rr.error_flag, CAMPX_ERR_FLOW_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      apply(sidx, e & 0xffu);
    } else {
      apply(sidx, (uint32_t)frame_trace[first_row + (sidx >> 1)]);
    }
  // ---- out: aligned, contiguous KiB stores
  int8_t* frame = rr.dst + (int64_t)t * rr.slab_bytes;
#pragma unroll
  for (int j = 0; j < kPipeWin; ++j) {
    const uint32_t off = woff0 + (uint32_t)j * 1024u + lane * 16u;
    if (off < rr.slab_bytes)
      __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(win0 + j * 1024 + lane * 16u),
                                  reinterpret_cast<u32x4*>(frame + off));
  }
}

// ---------------------------------------------------------------------------
// The shared launch for games of two to four movers (round 5): the same shape - update workgroups
// of 64 environments first (one producer wave walking the pair / tuple table, two consumers, one
// loader), every later workgroup four one-shot render waves of one 2 KiB window each - with the
// render role patching up to 2K bytes per row from K planes of the PREVIOUS rollout's trace
// (render_kernel's patch rule: a mover that shows sets its own layer's byte and clears the byte
// of the scenery it stands in front of).  Deferred rollouts only: the two roles share nothing.
// kLdsEntries (two movers): the pair table's entries staged in dynamic LDS - the chain is then an
// LDS read per frame, but EVERY workgroup of the launch, the render ones too, is charged for the
// bytes; without, the chain goes through L1 / L2 and the launch keeps 5 workgroups per CU.
template <int K>
using PipeMultiLds = std::conditional_t<K == 2, UpdatePairLds<kPipeProd, CAMPX_PAIR_GROUP>, UpdateTupleLds<kPipeProd>>;

#ifndef CAMPX_PIPE_MULTI_MINWAVES
#define CAMPX_PIPE_MULTI_MINWAVES 4     // (128 VGPRs: the four-mover update role would take 138 and a wave per SIMD)
#endif
// kFlow: the render role is THIS rollout's, as in pipe_table_kernel<true>: the update role stores a
// tagged copy of every mover's plane, a render wave waits for the entries of its rows.
template <int K, bool kLdsEntries, bool kFlow = false>
__global__ __launch_bounds__(kPipeWaves * kWave, CAMPX_PIPE_MULTI_MINWAVES) void pipe_multi_kernel(
    PairParams pp, TupleParams tp, CampxState st, const int8_t* __restrict__ actions, CampxOutputs out,
    int64_t B, int32_t T, int32_t reset_first, int64_t trace_plane, FrameCodec fc, PipeRender rr) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_entries[];
  __shared__ PipeMultiLds<K> L;
  static_assert(sizeof(L) >= kPipeWaves * (kPipeSpan + 2 * CAMPX_MAX_CELLS), "render windows fit the update LDS");
  static_assert(K >= 2 && K <= CAMPX_MAX_DYN, "two to four movers");
  if (blockIdx.x < rr.U) {
    if (kFlow && rr.debug_delay)     // (tests: hold the update role back so that render waves time out)
      for (uint32_t i = 0; i < rr.debug_delay; ++i) __builtin_amdgcn_s_sleep(127);
    if constexpr (K == 2)
      update_pair_body<kLdsEntries, kPipeProd, kPipeCons, kFlow>(L, lds_entries, blockIdx.x, pp, st, actions, out,
                                                                 B, T, reset_first, trace_plane, fc, rr.tagged, rr.tag);
    else
      update_tuple_body<K, kPipeProd, kPipeCons, kFlow>(L, blockIdx.x, tp, st, actions, out, B, T, reset_first,
                                                        trace_plane, fc, rr.tagged, rr.tag);
    return;
  }
  constexpr uint32_t P = 2u * K;                       // patches per row: (mover) x (set | clear)
  const uint32_t item = blockIdx.x - rr.U;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t t = item / rr.per_frame;
  uint32_t wx = item - t * rr.per_frame;
  wx = (wx & 7u) * (rr.per_frame >> 3) + (wx >> 3);    // per_frame is a multiple of 8: one XCD, one eighth
  const uint32_t shift = (rr.shift_base + t * rr.shift_slab) & (kPipeSpan - 1u);
  const uint32_t widx = wx * (uint32_t)kPipeWaves + wave;
  if ((uint64_t)widx * kPipeSpan >= (uint64_t)rr.slab_bytes + shift) return;
  int8_t* win0 = reinterpret_cast<int8_t*>(&L) + wave * (kPipeSpan + 2 * CAMPX_MAX_CELLS);
  uint16_t* scen_off = reinterpret_cast<uint16_t*>(win0 + kPipeSpan);
  const uint32_t R = rr.R;
  const uint32_t rot_pitch = ((R + 15u) & ~15u) + 16u;
  const uint32_t woff0 = widx * kPipeSpan - shift;
  const uint32_t wlo = widx * kPipeSpan < shift ? 0u : woff0;
  const uint32_t whi = __umulhi(rr.m, wlo);
  const uint32_t first_row = (((wlo - whi) >> rr.sh1) + whi) >> rr.sh2;
  const uint32_t wend = (woff0 + kPipeSpan - 1u < rr.slab_bytes) ? woff0 + kPipeSpan - 1u : rr.slab_bytes - 1u;
  const uint32_t ehi = __umulhi(rr.m, wend);
  const uint32_t last_row = (((wend - ehi) >> rr.sh1) + ehi) >> rr.sh2;
  const uint32_t slots = (last_row - first_row + 1u) * P;
  const uint8_t* frame_trace = rr.trace + (int64_t)t * rr.pitch;
  // ---- loads first: two trace bytes per lane, the window's scenery chunks, two cells' scenery layer
  // kFlow: from the tagged copy - the raw aligned dword a slot's entry sits in (agent-scope loads;
  // shifting it here would wait for it here), valid when it carries this launch's tag
  const uint16_t* frame_tagged = kFlow ? rr.tagged + (int64_t)t * rr.pitch : nullptr;
  auto entry_of = [&](uint32_t sidx) {
    const uint32_t r = sidx / P, d = (sidx - r * P) >> 1;
    uint32_t row = first_row + r;
    row = row <= last_row ? row : last_row;              // clamp: slot unused, entry ignored
    if (kFlow)
      return __hip_atomic_load(reinterpret_cast<const uint32_t*>(frame_tagged + (int64_t)d * rr.plane + (row & ~1u)),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (uint32_t)frame_trace[(int64_t)d * rr.plane + row];
  };
  auto half_of = [&](uint32_t sidx) {                    // which half of that dword
    uint32_t row = first_row + sidx / P;
    row = row <= last_row ? row : last_row;
    return (row & 1u) * 16u;
  };
  uint32_t ent[2];
  const bool two = slots > kWave;
  ent[0] = entry_of(lane);
  ent[1] = two ? entry_of(lane + kWave) : 0u;
  u32x4 scen[kPipeWin];
#pragma unroll
  for (int j = 0; j < kPipeWin; ++j) {
    const uint32_t off = woff0 + (uint32_t)j * 1024u + lane * 16u;
    const uint32_t hi = __umulhi(rr.m, off);
    const uint32_t row = (((off - hi) >> rr.sh1) + hi) >> rr.sh2;
    const uint32_t k = off - row * R;
    scen[j] = *reinterpret_cast<const u32x4*>(rr.rot + (k & 15u) * rot_pitch + (k & ~15u));
  }
  const uint32_t top2 = *reinterpret_cast<const uint16_t*>(rr.top_layer + 2u * lane);
#pragma unroll
  for (int j = 0; j < kPipeWin; ++j)
    *reinterpret_cast<u32x4*>(win0 + j * 1024 + lane * 16u) = scen[j];
  {
    const uint32_t c = 2u * lane;
    const uint32_t lo = (top2 & 0xffu) * (uint32_t)rr.cells + c;
    const uint32_t hi2 = (top2 >> 8) * (uint32_t)rr.cells + c + 1u;
    *reinterpret_cast<uint32_t*>(scen_off + c) = lo | (hi2 << 16);
  }
  if (kFlow) {
    // a wave whose rows are not there yet sleeps and looks again - at its own entries
    uint32_t naps = 0;
    const uint32_t sh0 = half_of(lane), sh1 = half_of(lane + kWave);
    const uint32_t want = rr.tag << 8;
    while (__any(((ent[0] >> sh0) & 0xff00u) != want || (two && ((ent[1] >> sh1) & 0xff00u) != want))) {
      if (naps < 4u) __builtin_amdgcn_s_sleep(CAMPX_FLOW_NAP);
      else __builtin_amdgcn_s_sleep(CAMPX_FLOW_NAP_LONG);
      if (++naps > rr.max_naps) {      // (see pipe_table_kernel: loudly, and on with what it has)
        flow_gave_up(rr.error_flag);
        break;
      }
      ent[0] = entry_of(lane);
      if (two) ent[1] = entry_of(lane + kWave);
    }
    ent[0] = (ent[0] >> sh0) & 0xffu;
    ent[1] = (ent[1] >> sh1) & 0xffu;
  }
  // ---- patches (a select chain over the kernel arguments, as in render_kernel)
  auto off_of = [&](uint32_t d) {
    int v = __builtin_amdgcn_readfirstlane(rr.dyn_offs[0]);
#pragma unroll
    for (int k = 1; k < K; ++k) v = (d == (uint32_t)k) ? __builtin_amdgcn_readfirstlane(rr.dyn_offs[k]) : v;
    return (uint32_t)v;
  };
  auto apply = [&](uint32_t sidx, uint32_t e) {
    const uint32_t r = sidx / P, p = sidx - r * P;
    const uint32_t cell = e & 0x7fu;
    const uint32_t byte = (p & 1u) ? off_of(p >> 1) + cell : (uint32_t)scen_off[cell];
    const uint32_t at = (first_row + r) * R + byte - woff0;
    if (sidx < slots && (e >> 7) && at < kPipeSpan) win0[at] = (int8_t)(p & 1u);
  };
  apply(lane, ent[0]);
  apply(lane + kWave, ent[1]);
  for (uint32_t sidx = lane + 2u * kWave; sidx < slots; sidx += kWave)     // short rows only
    if (kFlow) {   // (each lane waits for its own entry)
      const uint32_t sh = half_of(sidx);
      uint32_t e = entry_of(sidx) >> sh;
      for (uint32_t naps = 0; ((e >> 8) & 0xffu) != rr.tag && naps < rr.max_naps; ++naps) {
        __builtin_amdgcn_s_sleep(CAMPX_FLOW_NAP_LONG);
        e = entry_of(sidx) >> sh;
      }
      if (((e >> 8) & 0xffu) != rr.tag && rr.error_flag)
        __hip_atomic_fetch_or(rr.error_flag, CAMPX_ERR_FLOW_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      apply(sidx, e & 0xffu);
    } else {
      apply(sidx, entry_of(sidx));
    }
  // ---- out: aligned, contiguous KiB stores
  int8_t* frame = rr.dst + (int64_t)t * rr.slab_bytes;
#pragma unroll
  for (int j = 0; j < kPipeWin; ++j) {
    const uint32_t off = woff0 + (uint32_t)j * 1024u + lane * 16u;
    if (off < rr.slab_bytes)
      __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(win0 + j * 1024 + lane * 16u),
                                  reinterpret_cast<u32x4*>(frame + off));
  }
}

// Whether launch_pipe() can put these two passes in one launch (`prev`: the rollout to render).
bool pipe_ok(const CampxSpec& s, const CampxOutputs& out, const CampxOutputs& prev, int64_t B,
             int32_t T, bool use_table) {
  const int64_t HW = (int64_t)s.rows * s.cols, R = HW * s.n_layers;
  // (`use_table`: the game's table is there - the one-mover table in the spec, or, for two to
  // four movers, the caller's pair / tuple table)
  if (!use_table || s.n_dyn < 1 || s.n_dyn > CAMPX_MAX_DYN || !prev.trace || !prev.obs || prev.board)
    return false;
  if (prev.obs_format != CAMPX_OBS_INT8 || prev.obs_t_stride != B * R) return false;
  if ((B * R) % 16 != 0 || B * R >= (int64_t)1 << 31 || T > 65535) return false;
  if (reinterpret_cast<uintptr_t>(prev.obs) & 15) return false;
  // Where one launch beats two (boat race / wall world, T = 100, of HBM peak, two launches ->
  // one; profiles/r04_deferred_ab.txt): B = 4 096 0.34 -> 0.53, 16 384 0.62 -> 0.75 / 0.77 ->
  // 0.80, 32 768 0.74 -> 0.78, 65 536 0.82 -> 0.79-0.88 / 0.87 -> 0.83, 200 000 0.83 -> 0.57.  The
  // render role streams ~4 % below render_kernel's rate (84 VGPRs: 20 waves per CU), which
  // hiding ~15 us of update pass repays only while the observations of a rollout are under
  // ~2 GB; and the update workgroups, dispatched first, must leave room for render ones in the
  // first wave of resident workgroups (5 per CU, 1 280): 1 024 of them (65 536 environments)
  // is where the shared launch stops winning consistently - single launches then swing
  // between 165 and 200 us - so it is taken up to 512 (32 768 environments).
  constexpr int64_t max_b = 512 * kPipeEnvs, max_bytes = 2000000000ll;
  if (B > max_b || B * R * T > max_bytes) return false;
  if (s.n_dyn >= 2) {
    // Two to four movers (pipe_multi_kernel; sokoban levels 0 / 1 / 2, T = 100, of HBM peak, two
    // launches in order -> the shared launch, profiles/r05_multimover_deferred_ab.txt): B = 4 096
    // 0.30 -> 0.49 / 0.25 -> 0.42 / 0.29 -> 0.40, 8 192 0.45 -> 0.60 / 0.41 -> 0.56 / 0.43 -> 0.57,
    // 16 384 0.60 -> 0.67 / 0.58 -> 0.61 / 0.60 -> 0.61, 32 768 0.73 -> 0.73 / 0.72 -> 0.65 / 0.72 ->
    // 0.63: the update role's registers (99 / 122 / 128 VGPRs) leave the render role 16 waves per
    // CU, which hiding the update pass repays only while that pass is a large part of the rollout.
    // From 16 384 environments up the update pass of three- / four-mover games on a high-priority
    // side stream (campx::rollout_pipelined, one op) does better - 0.70 / 0.75, 32 768: 0.85 / 0.84 -
    // so they take the shared launch up to 8 192, the two-mover game up to 16 384.
    const int64_t multi_max = s.n_dyn == 2 ? 16384 : 8192;
    if (B > multi_max) return false;
  }
  (void)out;
  return true;
}

// The tagged copy of the trace of one-launch rollouts lives in the caller's scratch block
// (CampxOutputs.overlap_ctl): 16 bytes of header, then per mover [T, pitch] 16-bit entries.
int64_t flow_scratch_bytes(int64_t B, int32_t T) {
  const int64_t pitch = (B + 15) / 16 * 16;       // (the widest row pitch a caller may use)
  return 16 + 2 * (int64_t)CAMPX_MAX_DYN * T * pitch;     // (and the most movers a game may have)
}

// A launch's tag: 1..255, counting up per scratch block - in the CALLER's CampxFlowState (the
// library keeps nothing between calls).  Every launch rewrites every entry of its T frames, so at
// its start they all carry the previous launch's tag - unless the last launch on this block had
// another T, B or row pitch (entries of the pad columns or of frames past T would keep an old tag,
// which the 255-tag wrap would make current again), or the state is fresh: then the whole block is
// zeroed first (stream-ordered; tag 0 is never used).
static uint32_t next_flow_tag(CampxFlowState& l, void* block, int64_t block_bytes, int64_t B, int32_t T,
                              int64_t pitch, int n_dyn, hipStream_t stream) {
  if (l.tag < 1 || l.tag > 255 || l.B != B || l.T != T || l.pitch != pitch || l.n_dyn != n_dyn ||
      l.block != (int64_t)(intptr_t)block) {
    (void)hipMemsetAsync(block, 0, (size_t)block_bytes, stream);
    l = CampxFlowState{0, B, T, pitch, n_dyn, (int64_t)(intptr_t)block};
  }
  l.tag = l.tag % 255 + 1;
  return (uint32_t)l.tag;
}

static uint32_t flow_max_naps() { return (uint32_t)knob(K_FLOW_MAX_NAPS); }
static uint32_t flow_debug_delay() { return (uint32_t)knob(K_FLOW_DEBUG_DELAY); }

static int32_t launch_pipe_or_flow(bool flow, const CampxSpec& s, const CampxSpec* spec_dev, CampxState st,
                                   const int8_t* actions, CampxOutputs out, CampxOutputs prev, int64_t B,
                                   int32_t T, int32_t reset_first, hipStream_t stream) {
  const int HW = s.rows * s.cols;
  const MoverParams mp = {s.rows, s.cols, s.n_layers, s.dyn_layer[0], s.dyn_z[0],
                          s.dyn_row0[0], s.dyn_col0[0]};
  const FrameCodec fc = make_codec(s);
  PipeRender rr;
  memset(&rr, 0, sizeof(rr));
  rr.R = (uint32_t)(s.n_layers * HW);
  uint32_t l = 0;
  while ((1ull << l) < rr.R) ++l;
  rr.m = (uint32_t)(((1ull << 32) * ((1ull << l) - rr.R)) / rr.R + 1);   // as launch_render_from
  rr.sh1 = l < 1 ? l : 1;
  rr.sh2 = l > 0 ? l - 1 : 0;
  rr.slab_bytes = (uint32_t)(B * rr.R);
  rr.shift_base = (uint32_t)(reinterpret_cast<uintptr_t>(prev.obs) & (kPipeSpan - 1u));
  rr.shift_slab = rr.slab_bytes & (kPipeSpan - 1u);
  rr.cells = HW;
  rr.dyn_off = s.dyn_layer[0] * HW;
  rr.pitch = row_pitch(prev, B);
  const char* blob = reinterpret_cast<const char*>(spec_dev);
  rr.rot = reinterpret_cast<const int8_t*>(blob + offsetof(CampxSpec, rot_obs));
  rr.top_layer = reinterpret_cast<const uint8_t*>(blob + offsetof(CampxSpec, static_top_layer));
  rr.trace = prev.trace;
  rr.dst = prev.obs;
  rr.T = T;
  for (int d = 0; d < s.n_dyn && d < CAMPX_MAX_DYN; ++d) rr.dyn_offs[d] = s.dyn_layer[d] * HW;
  rr.plane = (int64_t)T * rr.pitch;
  const uint64_t reach = (uint64_t)rr.slab_bytes + ((rr.shift_base | rr.shift_slab) ? kPipeSpan - 1u : 0u);
  const uint64_t block_span = (uint64_t)kPipeSpan * kPipeWaves;
  rr.per_frame = (uint32_t)(((reach + block_span - 1) / block_span + 7) & ~7ull);
  rr.U = (uint32_t)((B + kPipeEnvs - 1) / kPipeEnvs);
  const uint64_t n_blocks = (uint64_t)rr.U + (uint64_t)T * rr.per_frame;
  if (n_blocks > 0x7fffffffull) return CAMPX_EINVAL;
  const dim3 grid((unsigned)n_blocks), block(kPipeWaves * kWave);
  // (84 VGPRs: 20 waves per CU.  Squeezed into 80 or 72 registers - 24 / 28 waves, the update
  // body spilling - the launch was slower from 32 768 environments up and at 4 096, level at
  // 16 384: profiles/r04_deferred_ab.txt)
  if (s.n_dyn >= 2) {
    // two to four movers (deferred rollouts only): pipe_multi_kernel over the caller's pair / tuple table
    if (!st.pair_table) return CAMPX_EINVAL;
    if (flow) {
      rr.tagged = reinterpret_cast<uint16_t*>(out.overlap_ctl + 4);
      rr.tag = next_flow_tag(*out.flow_state, out.overlap_ctl, out.overlap_ctl_bytes, B, T, rr.pitch, s.n_dyn, stream);
      rr.max_naps = flow_max_naps();
      rr.debug_delay = flow_debug_delay();
      rr.error_flag = out.error_flag;
    }
    const PairParams pp = make_pair_params(s);
    const TupleParams tp = make_tuple_params(s);
    const int64_t plane = (int64_t)T * row_pitch(out, B);
    const int n_entries = HW * HW * CAMPX_N_ACTIONS;
    // (the pair table's entries staged in LDS when they fit - charged to every workgroup of the
    // launch, the render ones too, and still the faster form: sokoban B = 4 096 / 8 192 / 16 384
    // 48 / 50 / 66 us through L1 / L2, 37 / 41 / 60 from LDS)
    const bool in_lds = s.n_dyn == 2 && n_entries <= kPairLdsEntries;
    const size_t shmem = in_lds ? (((size_t)n_entries * sizeof(uint32_t) + 15) & ~(size_t)15) : 0;
#define CAMPX_PIPE_MULTI(KK, LDS)                                                                        \
  do {                                                                                                   \
    if (flow)                                                                                            \
      hipLaunchKernelGGL((pipe_multi_kernel<KK, LDS, true>), grid, block, shmem, stream, pp, tp, st,     \
                         actions, out, B, T, reset_first, plane, fc, rr);                                \
    else                                                                                                 \
      hipLaunchKernelGGL((pipe_multi_kernel<KK, LDS, false>), grid, block, shmem, stream, pp, tp, st,    \
                         actions, out, B, T, reset_first, plane, fc, rr);                                \
  } while (0)
    if (s.n_dyn == 2 && in_lds) CAMPX_PIPE_MULTI(2, true);
    else if (s.n_dyn == 2) CAMPX_PIPE_MULTI(2, false);
    else if (s.n_dyn == 3) CAMPX_PIPE_MULTI(3, false);
    else CAMPX_PIPE_MULTI(4, false);
#undef CAMPX_PIPE_MULTI
  } else if (flow) {
    rr.tagged = reinterpret_cast<uint16_t*>(out.overlap_ctl + 4);
    rr.tag = next_flow_tag(*out.flow_state, out.overlap_ctl, out.overlap_ctl_bytes, B, T, rr.pitch, s.n_dyn, stream);
    rr.max_naps = flow_max_naps();
    rr.debug_delay = flow_debug_delay();
    rr.error_flag = out.error_flag;
    hipLaunchKernelGGL(pipe_table_kernel<true>, grid, block, 0, stream, mp, spec_dev, st, actions, out, B, T,
                       reset_first, fc, rr);
  } else {
    hipLaunchKernelGGL(pipe_table_kernel<false>, grid, block, 0, stream, mp, spec_dev, st, actions, out, B, T,
                       reset_first, fc, rr);
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? CAMPX_OK : hip_failed(e);
}

int32_t launch_pipe(const CampxSpec& s, const CampxSpec* spec_dev, CampxState st,
                    const int8_t* actions, CampxOutputs out, CampxOutputs prev, int64_t B,
                    int32_t T, int32_t reset_first, hipStream_t stream) {
  return launch_pipe_or_flow(false, s, spec_dev, st, actions, out, prev, B, T, reset_first, stream);
}

// ---------------------------------------------------------------------------
// ONE rollout in one launch, the render following the update pass as it goes: the same launch
// shape with the render role reading THIS rollout's trace - from a tagged copy, 16-bit entries
// byte | tag << 8 that the update role's consumers store beside the trace (write-through, no drain,
// no flag).  A render wave loads the entries of its rows with agent-scope loads and, while any of
// them carries another launch's tag, sleeps and looks again.  Every workgroup of the grid is
// dispatched in order, so the update ones (lowest indices, at most 1 024 <= the 1 280 resident
// ones) all run before any render workgroup can wait for them: no tickets, no deadlock.
// What the two earlier forms cost (tools/gpu_flow_variants.sh, B = 4 096 / 16 384, two launches
// 27 / 60 us, waits compiled out 19 / 47): progress WORDS side by side, polled by every waiting
// wave: 38 / 71 (64 update workgroups' words in four lines of one L2 channel - everything else
// through that channel, the update role's stores among it, queued behind the polls); the words
// 256 bytes apart: 24 / 57 (a flag is one more dependent trip in front of each short wave's
// loads, and a hop's price sits in the consumer CU's own memory queue: MI355X_MICROARCH.md,
// handoff-flag against handoff-1to1).  Tagged entries are that guide's data-tagged granules.
// Measured (boat race, T = 100, us per rollout, two launches / one; profiles/r04_flow_ab.txt):
// B = 1 024 20.7 / 16.9, 4 096 27.9 / 23.8, 8 192 38.2 / 35.9, 16 384 60.2 / 61.8, 32 768 101 / 113,
// 65 536 184 / 234 - the tagged copy is read from the fabric (write-through stores drop their
// lines from L2) by every render wave, which costs more than hiding the update pass saves once
// the render is the longer part: on for B <= 8 192; setting flow=0: never.
// Not while the stream is being captured into a graph: a replay would reuse the launch's tag.
bool flow_ok(const CampxSpec& s, const CampxOutputs& out, int64_t B, int32_t T, bool use_table,
             hipStream_t stream, bool ask_stream) {
  constexpr int64_t max_b = 8192;
  if (!knob(K_FLOW) || B > max_b || !out.overlap_ctl || !out.trace || !out.obs) return false;
  // (no caller-owned tag state, or nowhere to report a render wave that gave up: two launches)
  if (!out.flow_state || !out.error_flag) return false;
  if (s.n_dyn < 1 || out.overlap_ctl_bytes < 16 + 2 * (int64_t)s.n_dyn * T * row_pitch(out, B) ||
      (reinterpret_cast<uintptr_t>(out.overlap_ctl) & 15))
    return false;
  if (row_pitch(out, B) % 2 != 0) return false;     // (entries are read as aligned dwords)
  // Two to four movers (pipe_multi_kernel<K, ., true>; sokoban levels 0 / 1 / 2, T = 100, us per
  // rollout, two launches / one; profiles/r05_multimover_flow_ab.txt): B = 1 024 28 / 23, 50 / 46,
  // 52 / 68; 4 096 35 / 30, 62 / 53, 65 / 71; 8 192 46 / 44, 77 / 71, 86 / 79.  The four-mover game's
  // update role (64 environments a workgroup, its chain a load from the 212 MB tuple table per
  // frame, 128 VGPRs with 12 spilled) is the longer part of a small launch and slower beside
  // render waves than alone: it takes the one launch from 4 097 environments up only.
  if (s.n_dyn >= 4 && B < 4097) return false;
  CampxOutputs self = out;
  self.board = nullptr;          // (rendered by the ordinary kernel afterwards)
  if (!pipe_ok(s, out, self, B, T, use_table)) return false;
  if (!ask_stream) return true;           // (campx_flow_shared: "a stream that is not being captured")
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &capturing) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return capturing == hipStreamCaptureStatusNone;
}

int32_t launch_flow(const CampxSpec& s, const CampxSpec* spec_dev, CampxState st,
                    const int8_t* actions, CampxOutputs out, int64_t B, int32_t T,
                    int32_t reset_first, hipStream_t stream) {
  return launch_pipe_or_flow(true, s, spec_dev, st, actions, out, out, B, T, reset_first, stream);
}


}  // namespace campx_impl
