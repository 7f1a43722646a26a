/*
 * campx_hip.h - C ABI of libcampx_hip.so, the MI355X (gfx950) batched grid-world
 * step/render engine.
 *
 * The reference (OpenMined/CampX) is pure Python and has no FFI; its hot path is
 * the per-frame call chain
 *
 *     Engine.play                      campx/engine.py:114-166
 *       Engine._update_and_render      campx/engine.py:168-208
 *         <entity>.update(...)         examples/boat_race.py:35-59, 69-91
 *         Engine._render               campx/engine.py:295-324
 *           BaseObservationRenderer.*  campx/rendering.py:104-219
 *       Engine._apply_and_clear_plot   campx/engine.py:211-293
 *
 * which produces, for ONE environment, `(Observation(board, layers,
 * layered_board), reward, discount)`.  The entry points below are what a binding
 * for that path binds instead: the same function for B environments at once,
 * for T consecutive frames per call, on device buffers the caller owns.
 * INTEGRATION.md shows the ctypes stub a CampX maintainer would add.
 *
 * Conventions
 *   - plain C, no exceptions: every function returns CAMPX_OK (0) or a negative
 *     CAMPX_E* code; campx_strerror() names it.
 *   - the library owns no memory between calls: all buffers are caller-allocated DEVICE
 *     memory unless a parameter says "host".  The only process-wide state: the settings
 *     (campx_config_set / _get / _string below: one table, no environment variable but
 *     CAMPX_CONFIG) and - read-only after first use - per device, its CU count and the
 *     dynamic-LDS limit already granted to a kernel.
 *   - launches are asynchronous on the hipStream_t passed as `void* stream`
 *     (NULL = the default stream); nothing in here synchronises.
 *   - re-entrant; calls on distinct state buffers may be issued concurrently.
 */
#ifndef CAMPX_HIP_H_
#define CAMPX_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAMPX_SPEC_MAGIC 0x58504d43u /* 'CMPX' */
#define CAMPX_SPEC_VERSION 1u

#define CAMPX_MAX_CELLS 128  /* rows * cols */
#define CAMPX_MAX_LAYERS 16  /* characters of a game = channels of layered_board */
#define CAMPX_MAX_DYN 4      /* one-cell things that move (agents, boxes) */
#define CAMPX_MAX_STATIC 16  /* drapes whose mask never changes */
#define CAMPX_MAX_RULES 16
#define CAMPX_N_ACTIONS 5    /* left, right, up, down, stay (examples/boat_race.py:26) */

enum {
  CAMPX_OK = 0,
  CAMPX_EINVAL = -1,   /* NULL / misaligned / out-of-range argument */
  CAMPX_ESPEC = -2,    /* GameSpec fails validation */
  CAMPX_ELAUNCH = -3,  /* HIP refused the launch (see campx_last_hip_error) */
  CAMPX_ENODEV = -4,   /* no gfx950 device */
  CAMPX_ENOMEM = -5    /* host scratch allocation failed (set-up calls only) */
};

enum { CAMPX_OBS_INT8 = 0, CAMPX_OBS_F16 = 1, CAMPX_OBS_BF16 = 2 };

/* Rule opcodes: the update() bodies of the reference's example entities. */
enum {
  /* Move dynamic thing `dyn` one cell by the action, cyclically; if the cell it
   * would enter SHOWED (at the latest repaint) a character in `block_layers`,
   * it returns to the cell it was painted at.  Optional reward: `base`
   * (+1 when the entered cell showed a character in `reward_layers`).
   * examples/boat_race.py:35-59; Demo 1/2/3 cell 3. */
  CAMPX_OP_AGENT = 1,
  /* reward = base + [the cell dynamic thing `dyn` is in now showed layer `aux`
   * at the latest repaint] * bonus[action].  examples/boat_race.py:69-91. */
  CAMPX_OP_DIR_HOVER = 2,
  /* Box `dyn` moves one cell by the action if agent `aux`'s painted position,
   * moved by the action, is the box's cell and the cell beyond showed no
   * character of `block_layers`.  campx_amd/rules.py BoxDrape. */
  CAMPX_OP_BOX = 3,
  /* reward = base + [dynamic thing `dyn` stands on static drape `aux`] *
   * bonus[0]; if it does, the episode terminates with discount 0.
   * campx_amd/rules.py GoalDrape. */
  CAMPX_OP_GOAL = 4
};

typedef struct CampxRule {
  int32_t op;
  int32_t end_group;      /* 1 = a repaint follows this rule (campx/engine.py:208) */
  int32_t dyn;
  int32_t aux;
  uint32_t block_layers;  /* bit l = layer l */
  uint32_t reward_layers;
  int32_t has_reward;     /* rule calls Plot.add_reward (campx/plot.py:186) */
  float base;
  float bonus[CAMPX_N_ACTIONS];
  int32_t reserved[3];
} CampxRule;              /* 64 bytes */

/* One entry of the (cell, action) transition table of a one-mover game. */
typedef struct CampxTransition {
  float reward;       /* summed reward of the frame (NaN = None) */
  uint8_t next_cell;  /* row * cols + col after the frame */
  uint8_t done;       /* bit 0: the episode terminated on the frame; bits 4-7: discount code c:
                         the frame reports spec.discount_list[c], or - c == 0 - the default,
                         0.0 when it terminated and 1.0 otherwise (campx/plot.py:161-184,
                         232-257) */
  int8_t perf;        /* hidden performance of the frame (perf_scale * code + perf_offset) */
  uint8_t paint;      /* how the mover paints at next_cell: bits 0-6 the scenery layer it
                         covers there, bit 7 set when the scenery hides it instead */
} CampxTransition;    /* 8 bytes */

/*
 * GameSpec: the immutable description of one game, produced once per game by the
 * host (campx_amd/gamespec.py from the ascii_art_to_game() arguments,
 * campx/ascii_art.py:60-309) and uploaded by the caller to device memory.
 * Plain data, no pointers, same bytes on host and device.
 */
typedef struct CampxSpec {
  uint32_t magic, version;
  int32_t rows, cols;
  int32_t n_layers;                     /* L; layer l shows character layer_char[l], ascending */
  int32_t n_dyn;                        /* K */
  int32_t n_static;
  int32_t n_rules;
  int32_t any_reward;                   /* 0: nobody ever calls add_reward -> reward is NaN (None) */
  int32_t table_valid;                  /* 1: `table` below is filled (campx_spec_compile) */
  int32_t render_valid;                 /* 1: `rot_obs` / `rot_board` below are filled */
  int32_t perf_dyn;                     /* moving thing whose hidden performance is scored, or -1 */
  int32_t perf_n;                       /* length n of the cycle of cell classes (>= 2) */
  int32_t table_only;                   /* 1: `rules` is empty and the update pass exists only as
                                           tables the HOST filled by running the game's own Python
                                           update() bodies over every reachable state
                                           (campx_amd/tabulate.py: `table` for one mover,
                                           campx_pair_table_pack() for two to four).  Calls that
                                           would interpret the rules return CAMPX_ESPEC. */
  int32_t reserved0[2];
  uint8_t layer_char[CAMPX_MAX_LAYERS];
  int32_t dyn_layer[CAMPX_MAX_DYN];     /* layer painted by dynamic thing d */
  int32_t dyn_z[CAMPX_MAX_DYN];         /* its z rank, 1 = rearmost thing (0 = backdrop).  A
                                           host-tabulated game (table_only) may give its LAST
                                           things rank 0: values the tables are indexed by that
                                           are never painted - the z-order in force of a game
                                           that calls Plot.change_z_order (campx/plot.py:121-159),
                                           whose effect on the screen is in the table entries'
                                           "is the character its cell shows" bits */
  int32_t dyn_row0[CAMPX_MAX_DYN];      /* position in the art */
  int32_t dyn_col0[CAMPX_MAX_DYN];
  CampxRule rules[CAMPX_MAX_RULES];     /* update-schedule order */
  /* Per cell, considering the backdrop and the static drapes only: */
  uint8_t static_top_layer[CAMPX_MAX_CELLS]; /* layer of the front-most one */
  uint8_t static_top_z[CAMPX_MAX_CELLS];     /* its z rank (0 = backdrop) */
  uint16_t static_cover[CAMPX_MAX_CELLS];    /* bit s = static drape s covers the cell */
  /* layered board of the static scenery alone, [L][rows*cols] 0/1 */
  int8_t obs_template[CAMPX_MAX_LAYERS * CAMPX_MAX_CELLS];
  /* Games with ONE moving thing: the whole update pass of a frame is a function of
   * (cell the thing is in, action).  campx_spec_compile() tabulates it by running
   * the rule interpreter kernel once over every (cell, action) pair; the frame
   * loop then does one lookup instead of interpreting the rules.
   * Index: cell * CAMPX_N_ACTIONS + action. */
  CampxTransition table[CAMPX_MAX_CELLS * CAMPX_N_ACTIONS];
  /* For the render kernel of the two-kernel path, filled by campx_spec_compile():
   * 16 byte-rotations of the scenery's row (the layered board, resp. the flat board,
   * of the static scenery), each continued cyclically, so that any 16 consecutive
   * bytes of back-to-back rows are one aligned 16-byte load:
   *   rot[r * pitch + j] = row[(j + r) mod R],  pitch = round_up(R, 16) + 16,
   * R = L*rows*cols (rot_obs) or rows*cols (rot_board).  16-byte aligned in the blob. */
  int8_t rot_obs[16 * (CAMPX_MAX_LAYERS * CAMPX_MAX_CELLS + 16)];
  int8_t rot_board[16 * (CAMPX_MAX_CELLS + 16)];
  /* Hidden performance (examples/boat_race.py:117-151): class 1..perf_n of each cell,
   * 0 = none.  A frame scores +1 when thing perf_dyn goes from class i to class i+1
   * (cyclically), -1 for the reverse, else 0. */
  uint8_t cell_class[CAMPX_MAX_CELLS];
  /* Two kinds of hidden performance, both a small code per frame that the state tables
   * carry in 3 bits; the value written to CampxOutputs.perf is perf_scale * code + perf_offset.
   *   perf_mode 0  progress round a cycle of cell classes (above): code = progress + 1,
   *                scale 1, offset -1.
   *   perf_mode 1  a penalty for where things stand: code = sum over the moving things in
   *                perf_mask (bit d = thing d) of cell_class[its cell], at most 7 - the
   *                side-effects penalty of sokoban (-5 for a box next to a
   *                wall, -10 for a box in a corner: classes 1 and 2, perf_scale -5).
   * perf_dyn >= 0 says that the game has a hidden performance at all (mode 1: any thing
   * of perf_mask). */
  int32_t perf_mode, perf_mask, perf_scale, perf_offset;
  /* Discounts other than the default (Plot.change_default_discount, terminate_episode(d):
   * campx/plot.py:161-184, 232-257): code -> value, codes 1..15; code 0 is the default.
   * Only host-tabulated games (table_only) produce them. */
  float discount_list[16];
} CampxSpec;

/* Dynamic state of B environments, struct-of-arrays, DEVICE pointers. */
typedef struct CampxState {
  int8_t* pos;     /* [2*K, B]: plane 2d = row of dynamic thing d, 2d+1 = its column */
  uint8_t* done;   /* [B] game-over latch (campx/engine.py:285); a latched environment is
                      rebuilt from the art before its next action is applied */
  float* ret;      /* [B] return accumulated since the last rebuild, or NULL */
  const void* pair_table; /* optional, games with two to four moving things: the device table
                      campx_pair_table_build() filled; lets the two-kernel path look the
                      update pass up instead of interpreting the rules.  NULL = interpret. */
} CampxState;

/* Per-frame outputs, DEVICE pointers; any of them except `obs` may be NULL.
 * Frame t of a call is written at  base + t * <t_stride>  (in elements); a stride
 * of 0 makes every frame overwrite the first slot, so only the last survives. */
/* Bits of *CampxOutputs.error_flag. */
#define CAMPX_ERR_FLOW_TIMEOUT 1

/* What the library remembers about a scratch block of one-launch rollouts - in the CALLER's
 * memory (CampxOutputs.flow_state), so that the library holds no state of its own. */
typedef struct CampxFlowState {
  int64_t tag;    /* 1..255, of the block's last launch; 0: none yet (the block is cleared first) */
  int64_t B, T;   /* what the block's entries were written for */
  int64_t pitch;  /* the row pitch they were written with */
  int64_t n_dyn;  /* how many movers' planes they were written for (a game with MORE movers on the
                     same block would find stale tags in the extra planes) */
  int64_t block;  /* the address of the block this state describes (a state handed over with
                     another block starts afresh) */
} CampxFlowState;

typedef struct CampxOutputs {
  int8_t* obs;        /* layered_board [*, B, L, rows, cols] 0/1 (campx/rendering.py:213-215);
                         16-byte aligned */
  int64_t obs_t_stride;
  int8_t* board;      /* flat board [*, B, rows, cols] of character codes (rendering.py:217) */
  int64_t board_t_stride;
  float* reward;      /* [T, B]; NaN where the reference returns None */
  float* discount;    /* [T, B] 1.0, or 0.0 on the frame the episode ended (plot.py:179-184) */
  uint8_t* done;      /* [T, B] game-over flag after the frame */
  int8_t* perf;       /* [T, B] hidden performance of the frame (-1, 0, +1), or NULL; needs
                         spec.perf_dyn >= 0 */
  uint8_t* trace;     /* optional [K, T, B]: for moving thing d at frame t in environment e,
                           bits 0-6  the cell (row*cols + col) it is in after the frame
                           bit  7    1 when it is the character that cell shows; when 0 it
                                     is hidden and changes nothing in the observation
                         A compact trajectory in its own right (1 byte per thing per frame
                         against L*rows*cols of observation); and when it is given, frames are
                         stored back to back (obs_t_stride == B*L*rows*cols; any batch size),
                         the library runs the update pass and the render as
                         two kernels, which streams the observations to HBM faster (DESIGN.md
                         "Kernels"); with strides of 0 (only the last frame survives) it
                         renders just that frame from the last row of the trace.  Written
                         only on that path. */
  int32_t obs_format; /* element type of `obs` (obs_t_stride counts elements):
                         CAMPX_OBS_INT8 0/1 bytes (the default, 0);
                         CAMPX_OBS_F16 / CAMPX_OBS_BF16: 0.0 / 1.0 in that format, the tensor
                         the reference's driver builds with `layered_board.view(-1).float()`
                         (examples/reinforce.py:123,149) handed over without a conversion
                         pass.  16-bit formats are produced by the render kernel (rollouts:
                         they need `trace` and back-to-back frames) and by the one-frame
                         kernels of games with their tables (T == 1);
                         anything else returns CAMPX_EINVAL. */
  int32_t* bad_count; /* optional device int32: += number of action ids outside 0..4 this call
                         consumed (the reference asserts sum(act) == 1 per step,
                         examples/boat_race.py:48; here the check rides in the kernel that
                         reads the actions anyway and the caller looks when it likes).  Never
                         reset by the library. */
  int32_t* bad_flag;  /* optional int32, device memory or host memory mapped into the device
                         (hipHostMalloc): set to 1, by a plain system-scope store, when the
                         call consumed any id outside 0..4.  Lets a host poll for bad
                         actions without a stream synchronisation. */
  int64_t scalar_pitch; /* elements from one frame's row to the next in reward, discount, done,
                         perf and trace (the planes of `trace` are then T * scalar_pitch
                         apart); 0 = B, rows back to back.  With a batch size that is not a
                         multiple of 16 a row of a [T, B] array starts at any byte, and the
                         update kernels' 16-byte stores are misaligned for most frames (legal,
                         but 4.3 instead of 7 TB/s: 47 instead of 16 us per 100 frames at
                         B = 65 535).  A caller can pad the arrays instead - [T, pitch] with
                         pitch = B rounded up to a multiple of 16, the first B elements of a
                         row used: every row then starts aligned and the kernels write whole
                         16-byte groups (the pad holds unspecified values).  `actions` and the
                         observation / board frames are never padded. */
  uint32_t* overlap_ctl; /* optional device scratch of campx_flow_scratch_bytes(B, T) bytes, 16-byte
                         aligned; NULL: two launches per rollout.  With it - AND `flow_state` AND
                         `error_flag` below - a rollout of a table game (one mover; two to four
                         with CampxState.pair_table) at a batch of at most 8 192 environments
                         (int8 observations of every frame, whole 16-byte chunks per frame) runs
                         as ONE launch whose update pass and render overlap: update workgroups
                         first, render workgroups behind them reading a tagged 16-bit copy of
                         each mover's trace kept in this block as the update role writes it
                         (csrc/k_update.hip, pipe_table_kernel<true>: 15 / 22 / 34 us against
                         20 / 27 / 38 at B = 1 024 / 4 096 / 8 192; pipe_multi_kernel<K, ., true>,
                         sokoban with one box: 23 / 30 / 44 against 28 / 35 / 46; the four-mover
                         level from 4 097 environments up).  Not while `stream` is being
                         captured into a graph.  Setting flow = 0 (campx_config_set): never.
                         Two launches that may run at the same time must not share a block.
                         campx_flow_shared() says whether a call will take this path. */
  int64_t overlap_ctl_bytes;
  struct CampxFlowState* flow_state; /* HOST memory, caller-owned, one per `overlap_ctl` block,
                         zero-initialised by the caller when the block is allocated and from then
                         on read and written only by the library, inside the launch call: the tag
                         of the block's last launch and the (B, T, row pitch, bytes) its entries
                         were written for (a launch with any of them changed first clears the
                         block, stream-ordered).  The library itself keeps NO state between
                         calls.  NULL: two launches per rollout. */
  int32_t* error_flag; /* int32, device memory or host memory mapped into the device: bits OR-ed in
                         (system scope) when a launch could not do what it was asked to, although
                         the call returned CAMPX_OK: CAMPX_ERR_FLOW_TIMEOUT - a render wave of a
                         one-launch rollout waited for trace entries of its own launch until it
                         gave up (seconds; two launches sharing one block at the same time is
                         the known way to get there) and wrote frames from stale entries: the
                         observations of that launch are WRONG.  Never cleared by the library.
                         Required for the one-launch path (NULL: two launches per rollout), so
                         that this failure cannot pass unseen. */
} CampxOutputs;

/* Size of CampxOutputs.overlap_ctl for rollouts of T frames of B environments (of any game: a
 * 16-byte header and, for each of up to CAMPX_MAX_DYN movers, T rows of B rounded up to 16 entries). */
int64_t campx_flow_scratch_bytes(int64_t B, int32_t T);

/* 1 when campx_rollout_launch() of T frames of B environments of this game, with a scratch
 * block, flow state and error flag supplied, int8 observations of every frame at a 16-byte
 * aligned address, rows of the per-frame streams `scalar_pitch` apart (0: B) and a stream that
 * is not being captured, runs as ONE launch (pipe_table_kernel<true>, pipe_multi_kernel<K, ., true>
 * - the latter given the game's pair / tuple table, which this query cannot see); 0 when it runs as an
 * update launch and a render launch.  The library's own bounds and environment knobs, for
 * callers that allocate the block only where it is used and for whoever reports which kernel
 * ran (bench.py). */
int32_t campx_flow_shared(const CampxSpec* spec_host, int64_t B, int32_t T, int64_t scalar_pitch);

/* sizeof(CampxSpec), for bindings that allocate the blob themselves. */
int32_t campx_spec_size(void);

/* Check a HOST GameSpec: magic/version, bounds, rule operands. */
int32_t campx_spec_validate(const CampxSpec* spec_host);

/*
 * Optional, once per game: derive the acceleration tables of a GameSpec.
 *   - rot_obs / rot_board (render_valid): host arithmetic on the scenery tables.
 *   - table (table_valid), only for n_dyn == 1: the rule interpreter kernel is run
 *     over all (cell, action) pairs on the current device.
 * Set-up time only: allocates and frees its own scratch device memory and
 * synchronises `stream`.  Upload the spec to the device AFTER this call.
 */
int32_t campx_spec_compile(CampxSpec* spec_host, void* stream);

/*
 * Games with TWO TO FOUR moving things: the update pass of a frame is a function of
 * (cell of thing 0, ..., cell of thing K-1, action).  campx_pair_table_bytes() is the
 * size of its table for this game (0 when n_dyn < 2, or the table would exceed 1 MiB for
 * two movers / 512 MiB for three and four); campx_pair_table_build() fills
 * caller-allocated DEVICE memory of that size by running the rule interpreter kernel over
 * every (cell, ..., cell, action) tuple, the same way campx_spec_compile() does for
 * one-mover games (set-up time only: scratch allocation + stream synchronisation inside;
 * about 19 bytes of host and of device scratch per tuple).  Returns CAMPX_ESPEC when a
 * frame of this game can pay more than 256 distinct rewards (the table indexes a reward
 * list).  Layout: 256 floats (reward list), then n = (rows*cols)^K * 5 entries, index
 * ((cell0 * rows*cols + cell1) * rows*cols + ...) * 5 + action.
 * Two movers, uint32 entries:
 *   bits 0-6 cell of thing 0 after the frame, 7-13 cell of thing 1, 14/15 whether
 *   thing 0 / 1 is the character its cell shows, 16 done, 17-18 and 31 the hidden-performance
 *   code (3 bits), 19-26 index into the reward list, 27-30 discount code.
 * Three and four movers, uint64 entries:
 *   bits 7d..7d+6 cell of thing d after the frame, 28+d whether it is the character its
 *   cell shows, 32 done, 33-34 and 43 the hidden-performance code, 35-42 index into the
 *   reward list, 44-47 discount code.
 */
int64_t campx_pair_table_bytes(const CampxSpec* spec_host);
int32_t campx_pair_table_build(const CampxSpec* spec_host, const CampxSpec* spec_dev,
                               void* table_dev, void* stream);

/*
 * The same table from HOST arrays instead of the rule interpreter: for games whose update()
 * bodies are arbitrary Python (the reference's own examples/boat_race.py:28-91 classes, a
 * user's Drapes), tabulated on the host by running them (campx_amd/tabulate.py).  With
 * n = (rows*cols)^K * 5 entries indexed as above:
 *   trace   [K][n]  thing d after the frame: cell | (it is the character its cell shows) << 7
 *   reward  [n]     summed reward of the frame, NaN = None (campx/plot.py:208-211)
 *   done    [n]     bit 0: the episode terminated on the frame; bits 4-7: discount code
 *                   (index into spec_host->discount_list; 0 = the default)
 *   perf    [n]     hidden performance (a value perf_scale * code + perf_offset), or NULL
 * Packs them into `table_dev` (campx_pair_table_bytes() bytes of device memory) and
 * synchronises `stream`.  CAMPX_ESPEC for more than 256 distinct rewards or a perf value
 * that is not one of the eight the spec's scale and offset give.
 */
int32_t campx_pair_table_pack(const CampxSpec* spec_host, const uint8_t* trace,
                              const float* reward, const uint8_t* done, const int8_t* perf,
                              void* table_dev, void* stream);

/*
 * Put B environments into the state its_showtime() leaves them in
 * (campx/engine.py:487-544: positions from the art, game-over clear, return 0)
 * and, if out->obs / out->board are non-NULL, write that first observation to
 * slot 0 of each.
 */
int32_t campx_reset_launch(const CampxSpec* spec_host, const CampxSpec* spec_dev, CampxState state,
                           CampxOutputs out, int64_t B, void* stream);

/*
 * Advance B environments by T frames: T back-to-back Engine.play() calls for
 * each environment, with `actions[t*B + e]` (ids 0..4) the action of
 * environment e at frame t.  T = 1 is Engine.play().
 *
 * reset_first != 0 rebuilds every environment from the art before frame 0 (a
 * fresh make_game() per episode, examples/reinforce.py:122).
 * Action ids outside 0..4 are treated as 4 (stay) on the device; out.bad_count /
 * out.bad_flag (or campx_check_actions_launch()) detect them.  `actions` needs no padding:
 * nothing is read outside its T * B bytes.
 */
int32_t campx_rollout_launch(const CampxSpec* spec_host, const CampxSpec* spec_dev,
                             CampxState state, const int8_t* actions, CampxOutputs out, int64_t B,
                             int32_t T, int32_t reset_first, void* stream);

/*
 * The two halves of campx_rollout_launch()'s two-kernel path, for callers that want to
 * overlap them across calls (the update pass of launch i+1 is a short, latency-bound
 * kernel that fits under the observation stream of launch i when issued on another
 * stream; campx_amd/fused.py `rollout(pipelined=True)`):
 *
 *   campx_update_launch   the update pass alone: reads state and actions, writes state,
 *                         out.trace (required) and the per-frame scalars out.reward /
 *                         discount / done / perf / bad_*; ignores out.obs / out.board.
 *   campx_render_launch   expands out.trace into out.obs (and out.board): frames back to
 *                         back (obs_t_stride == B*L*rows*cols) or only the last one
 *                         (strides 0), T <= 65535 (one grid row per frame), else
 *                         CAMPX_EINVAL.  Reads nothing but the trace and the spec.
 * update then render on one stream == campx_rollout_launch(), which in addition runs
 * launches whose trace planes (T*B bytes) exceed 28 MB as chunks of frames.
 */
int32_t campx_update_launch(const CampxSpec* spec_host, const CampxSpec* spec_dev, CampxState state,
                            const int8_t* actions, CampxOutputs out, int64_t B, int32_t T,
                            int32_t reset_first, void* stream);
int32_t campx_render_launch(const CampxSpec* spec_host, const CampxSpec* spec_dev, CampxOutputs out,
                            int64_t B, int32_t T, void* stream);

/*
 * Rollouts pipelined across calls: the update pass of one rollout (as campx_update_launch:
 * `actions`, `out`) together with the render pass of the rollout BEFORE it (as
 * campx_render_launch: `prev.trace` -> `prev.obs` / `prev.board`, same B and T), for callers
 * whose next actions do not depend on the observations they are waiting for (open-loop action
 * streams; the reference has no such call - it is Engine.play(), campx/engine.py:145-222,
 * T times for rollout i + 1 interleaved with the _render() calls, engine.py:286-324, of
 * rollout i).  For table games of ONE TO FOUR movers (one: the table in the spec; two to four:
 * the pair / tuple table in `state.pair_table`, round 5) with int8 observations of whole 16-byte
 * chunks per frame and no flat board, up to 32 768 environments and 2 GB of observations per
 * rollout, the two passes share ONE launch (update workgroups first, the others render);
 * otherwise they are issued one after the other on `stream`.  prev.trace == NULL: the update
 * pass alone.
 */
int32_t campx_update_render_launch(const CampxSpec* spec_host, const CampxSpec* spec_dev,
                                   CampxState state, const int8_t* actions, CampxOutputs out,
                                   CampxOutputs prev, int64_t B, int32_t T, int32_t reset_first,
                                   void* stream);
/* 1 if campx_update_render_launch() puts the two passes of such rollouts (int8 observations of
 * every frame, no flat board) in one launch, 0 if it would issue them one after the other - in
 * which case a caller does better not to defer at all: campx_rollout_launch() renders a trace
 * that is still cached, a deferred render reads one that a whole launch has since evicted
 * (boat race, B = 200 000: 0.67 of peak against 0.83).  For a game of two to four movers the
 * answer assumes its table in CampxState.pair_table (without one: one after the other). */
int32_t campx_update_render_shared(const CampxSpec* spec_host, int64_t B, int32_t T);

/*
 * ---- Shape tier --------------------------------------------------------------------
 * Games made only of rigidly translated things that interact with nothing: the
 * reference's Hello World (examples/Hello World Example.ipynb cell 3: RollingDrape
 * `np.roll`s its whole mask, SlidingSprite moves diagonally, action 4 quits).  A thing
 * is a set of art cells plus a cyclic (row, col) offset that the action changes; things
 * may cover many cells and boards go up to CAMPX_SHAPE_MAX_CELLS cells, so these games
 * have their own spec and kernel (one wavefront per environment).
 *
 * Rendering follows campx/engine.py:295-324 and campx/rendering.py:104-219 including
 * the renderer's aliasing: paint_all_of() makes the canvas share the backdrop's storage
 * (rendering.py:128) until the first paint_drape rebinds it (rendering.py:178), so
 * sprites painted BEFORE the first drape in z-order are written into the backdrop for
 * good and leave trails.  The backdrop is therefore per-environment state here.
 */
#define CAMPX_SHAPE_SPEC_MAGIC 0x50485343u /* 'CSHP' */
#define CAMPX_SHAPE_SPEC_VERSION 1u
#define CAMPX_SHAPE_MAX_CELLS 1024 /* rows * cols; rows, cols <= 127 */
#define CAMPX_SHAPE_MAX_THINGS 8
#define CAMPX_SHAPE_MAX_LIST 2048  /* cells covered by all things together */

typedef struct CampxShapeThing {
  int32_t layer;            /* layer (character) it paints */
  int32_t is_sprite;        /* painted by paint_sprite (rendering.py:150), else paint_drape */
  int32_t visible;          /* Sprite.visible (campx/things.py:380); drapes: 1 */
  int32_t n_cells;          /* its shape: cells[cell_begin .. cell_begin + n_cells) */
  int32_t cell_begin;
  int32_t has_reward_mask;  /* bit a: action a makes it call Plot.add_reward(reward[a]) */
  int32_t terminate_mask;   /* bit a: action a makes it call Plot.terminate_episode() */
  int32_t reserved;
  int8_t drow[8], dcol[8];  /* offset change per action, already modulo rows / cols (>= 0) */
  float reward[8];
} CampxShapeThing;          /* 80 bytes */

typedef struct CampxShapeSpec {
  uint32_t magic, version;
  int32_t rows, cols;
  int32_t n_layers;
  int32_t n_things;                          /* stored in z-order, back to front */
  int32_t first_drape;                       /* index of the first thing that is not a sprite */
  int32_t any_reward;
  uint8_t layer_char[CAMPX_MAX_LAYERS];
  int32_t update_order[CAMPX_SHAPE_MAX_THINGS]; /* thing indices in update-schedule order
                                                (rewards are summed in that order) */
  CampxShapeThing things[CAMPX_SHAPE_MAX_THINGS];
  uint8_t backdrop[CAMPX_SHAPE_MAX_CELLS];   /* layer per cell of the Backdrop as its_showtime()
                                                leaves it (trail sprites' art cells painted) */
  uint16_t cells[CAMPX_SHAPE_MAX_LIST];      /* art cell of each shape cell: row << 8 | col */
} CampxShapeSpec;

int32_t campx_shape_spec_size(void);
int32_t campx_shape_spec_validate(const CampxShapeSpec* spec_host);

/*
 * Advance B environments of a shape game by T frames (T = 0 with emit_first: the
 * its_showtime() observation).  State: state.pos [2*n_things, B] int8 = cyclic (row, col)
 * offset of each thing from its art position, state.done, state.ret as for
 * campx_rollout_launch, and `backdrop_state` [B, rows*cols] int8 layer per cell (the
 * per-environment Backdrop; may be NULL when no visible sprite precedes the first
 * drape).  Outputs: out.obs (any out.obs_format), out.board, out.reward, out.discount, out.done,
 * out.bad_count / bad_flag; frames at base + t * stride as for campx_rollout_launch.
 * Action ids outside 0..4 move nothing, end nothing and are counted as bad.
 * The frame-major path (round 5, csrc/k_shape.hip): with `tables_dev` (campx_shape_tables_build)
 * and out.trace (8-byte aligned scratch of campx_shape_scratch_bytes(spec, B, T) bytes: the
 * things' offsets per frame and, for games with trails, the per-environment trail words every
 * fourth frame), a call that keeps every frame (int8, back to back, frames of whole 16-byte
 * chunks, no flat board, no emit_first) runs as two kernels - the update pass, then a render pass
 * of one-shot waves with memory-aligned 2 KiB windows that computes every W-cell row of the
 * observation arithmetically from 64-bit row words - which streams the observations at the
 * one-cell tier's rate.  Rows of 16 to 64 cells only; other games and calls ignore both (NULL is
 * fine) and run the one-wave-per-environment kernel.  Setting shape_split = 0: never.
 */
int64_t campx_shape_tables_bytes(const CampxShapeSpec* spec_host);   /* 0: not a game for that path */
/* Fills `tables_host` (HOST memory, `bytes` >= campx_shape_tables_bytes); the caller copies it
 * to the device.  Pure host code. */
int32_t campx_shape_tables_build(const CampxShapeSpec* spec_host, void* tables_host, int64_t bytes);
int64_t campx_shape_scratch_bytes(const CampxShapeSpec* spec_host, int64_t B, int32_t T);
int32_t campx_shape_rollout_launch(const CampxShapeSpec* spec_host, const CampxShapeSpec* spec_dev,
                                   const void* tables_dev, CampxState state, int8_t* backdrop_state,
                                   const int8_t* actions, CampxOutputs out, int64_t B, int32_t T,
                                   int32_t reset_first, int32_t emit_first, void* stream);


/*
 * Wide tier: games whose update pass is a STATE table - one row per state the game can
 * reach, (state, action) -> state - instead of a table indexed by the things' cells.  That
 * form has no board-size limit and no (rows*cols)^K blow-up, so it takes what the one-cell
 * tier cannot: boards ABOVE 128 cells (PyColab-sized mazes: 16x16, 20x20, 32x32 ...;
 * campx/engine.py:31 sets no limit), up to eight things that show, any number of hidden
 * values behind them (the z-order in force after Plot.change_z_order, keys that were
 * picked up, doors that opened).  The caller enumerates the states (campx_amd/tabulate.py runs
 * the game's own update() classes breadth-first from its_showtime()); state 0 is the
 * its_showtime() state.  The observation stream is the one-cell tier's render kernel, fed
 * by a 16-bit trace:
 *     entry = cell | (scenery layer the thing covers there) << 10 | shows << 15.
 * A frame of B environments is B rows of L*rows*cols bytes (1 536 B for 16x16 with six
 * characters, 6 KiB for 32x32), so the path is even more purely a write stream than the
 * one-cell tier's.
 */
#define CAMPX_WIDE_MAX_CELLS 1024        /* rows * cols; rows, cols <= 127 */
#define CAMPX_WIDE_MAX_STATES (1 << 24)
#define CAMPX_WIDE_MAX_DYN 8             /* things that move / come and go */
#define CAMPX_WIDE_MAX_VARIANTS 256      /* pictures of a scenery that changes */
#define CAMPX_WIDE_MAX_PIECES 16         /* cells of the scenery that come and go one by one */

typedef struct CampxWideSpec {
  uint32_t magic, version;
  int32_t rows, cols;
  int32_t n_layers;
  int32_t n_dyn;                   /* K: things that move / come and go, 1 .. CAMPX_WIDE_MAX_DYN */
  int32_t n_states;                /* S: reachable states, 1 .. CAMPX_WIDE_MAX_STATES */
  int32_t any_reward;
  int32_t has_perf;                /* `perf` below means something */
  int32_t any_dcode;               /* 1: some entry of `done` carries a discount code (0: the
                                      update kernel need not look the discount up) */
  int32_t n_variants;              /* V: pictures of the SCENERY the game shows (0 or 1: one; up to
                                      CAMPX_WIDE_MAX_VARIANTS).  A Backdrop.update() that repaints
                                      the backdrop (campx/things.py:103-148) - day and night over a
                                      whole floor - or a Drape of several cells that come and go
                                      (coins taken one by one) makes the scenery a function of
                                      the state: the render kernel then lays, per environment and
                                      frame, the pre-rotated row of the state's variant, which the
                                      update pass hands it as one more (never painted) plane of the
                                      trace.  With V > 1: n_dyn <= CAMPX_WIDE_MAX_DYN - 1, the trace
                                      holds n_dyn + 1 planes, and `variant_top_layer` /
                                      `state_variant` below are given. */
  int32_t n_pieces;                /* P: PIECES of the scenery, 0 .. CAMPX_WIDE_MAX_PIECES - fixed
                                      cells that show a character of their own in some states and
                                      the plain scenery in the others: the cells of a Drape whose
                                      curtain loses them one by one (coins that are taken, ice that
                                      breaks; campx/things.py:161-262 sets no one-cell limit), the
                                      cells a Backdrop.update() repaints (lamps).  Which of them
                                      show is a 16-bit MASK per state (`state_pieces`), handed to
                                      the render kernel as one more (never painted) plane of the
                                      trace; it patches them onto the scenery's row like the
                                      things, from ONE trace entry per environment however many
                                      there are.  With P > 0: n_dyn <= CAMPX_WIDE_MAX_DYN - 1,
                                      n_variants <= 1, the trace holds n_dyn + 1 planes. */
  uint8_t layer_char[CAMPX_MAX_LAYERS];
  int32_t dyn_layer[CAMPX_WIDE_MAX_DYN];   /* layer thing d paints */
  float discount_list[16];                 /* as CampxSpec.discount_list */
  uint8_t static_top_layer[CAMPX_WIDE_MAX_CELLS];  /* front-most scenery layer per cell */
  uint16_t piece_cell[CAMPX_WIDE_MAX_PIECES];      /* the cell of piece p (row * cols + col) */
  uint8_t piece_layer[CAMPX_WIDE_MAX_PIECES];      /* the layer it paints when it shows (it hides
                                                      static_top_layer there, like a thing) */
  /* HOST arrays, read by campx_wide_spec_validate(full) / campx_wide_tables_build() only -
   * the launch calls never touch them, they may be gone by then: */
  const uint16_t* state_cells;     /* [S][K]: bits 0-9 the cell thing d is on in state s; bit 15
                                      set when it does NOT show there (hidden by something in
                                      front, or not on the board at all) */
  const int32_t* next_state;       /* [S][5]: the state after (state, action) */
  const float* reward;             /* [S][5]: summed reward of the frame, NaN = None */
  const uint8_t* done;             /* [S][5]: as CampxTransition.done (bit 0 terminated, bits
                                      4-7 discount code) */
  const int8_t* perf;              /* [S][5] hidden performance (the value), or NULL */
  const uint8_t* variant_top_layer; /* [V][rows*cols]: the front-most scenery layer per cell in
                                      variant v (row 0 = static_top_layer); NULL when V <= 1 */
  const uint16_t* state_variant;   /* [S]: the variant the scenery shows in state s; NULL when V <= 1 */
  const uint16_t* state_pieces;    /* [S]: bit p set = piece p SHOWS in state s (not taken, and nothing
                                      in front of it); NULL when P = 0 */
} CampxWideSpec;

int32_t campx_wide_spec_size(void);
/* Checks the plain fields; the host arrays too when they are non-NULL. */
int32_t campx_wide_spec_validate(const CampxWideSpec* spec_host);

/*
 * Once per game: campx_wide_tables_bytes() of DEVICE memory, filled by
 * campx_wide_tables_build() from the host spec and its arrays (host arithmetic + one copy;
 * synchronises `stream`): the state table in the update kernel's packed form and the 16
 * byte-rotations of the scenery's row for the render kernel (CampxSpec.rot_obs / rot_board
 * explain them).
 */
int64_t campx_wide_tables_bytes(const CampxWideSpec* spec_host);
int32_t campx_wide_tables_build(const CampxWideSpec* spec_host, void* tables_dev, void* stream);

/*
 * State enumeration on the device for RULE games (campx_amd.rules classes, lowered to
 * CampxRule as for CampxSpec) on boards the one-cell tier cannot take (more than 128 cells): the
 * wide tier runs any state table, and for games of arbitrary Python classes the host fills it by
 * running them on the generic tier (campx_amd/tabulate.py: a frame of Python per (state,
 * action)); a multi-mover rule game on a PyColab-sized board (campx/engine.py:31 sets no size
 * limit) has millions of reachable states.  campx_wide_enumerate_launch() applies one frame of
 * the rules - the update pass of campx/engine.py:168-208 as rollout_kernel interprets it - to
 * N given states under each of the five actions; the caller (campx_amd/enumerate_states.py)
 * drives a breadth-first enumeration with it and fills CampxWideSpec's arrays from the last pass.
 *   cells_in    DEVICE uint16 [N][K]   the cell of every moving thing (row * cols + col)
 *   next_cells  DEVICE uint16 [N][5][K]
 *   reward      DEVICE float  [N][5]   NaN = None        done   DEVICE uint8 [N][5]
 *   shows       DEVICE uint8  [N][5]   bit d: thing d is the character its cell shows AFTER the frame
 *   perf        DEVICE int8   [N][5]   hidden performance of the frame, or NULL
 * The tables of CampxWideRules are DEVICE arrays of rows*cols entries (CampxSpec.static_* and
 * cell_class explain them).  Nothing is allocated; no host synchronisation.
 */
typedef struct CampxWideRules {
  uint32_t magic, version;         /* CAMPX_SPEC_MAGIC, CAMPX_SPEC_VERSION */
  int32_t rows, cols, n_layers;
  int32_t n_dyn;                   /* K: 1 .. CAMPX_MAX_DYN */
  int32_t n_rules, any_reward;
  int32_t perf_dyn, perf_n, perf_mode, perf_mask, perf_scale, perf_offset;
  int32_t dyn_layer[CAMPX_MAX_DYN];
  int32_t dyn_z[CAMPX_MAX_DYN];
  CampxRule rules[CAMPX_MAX_RULES];
  const uint8_t* top_layer;        /* DEVICE [rows*cols] front-most scenery layer per cell */
  const uint8_t* top_z;            /* DEVICE its z rank (0 = backdrop) */
  const uint16_t* cover;           /* DEVICE bit s = static drape s covers the cell */
  const uint8_t* cell_class;       /* DEVICE hidden-performance class per cell (all 0 without one) */
} CampxWideRules;

int32_t campx_wide_rules_size(void);
int32_t campx_wide_enumerate_launch(const CampxWideRules* rules_host, const uint16_t* cells_in,
                                    int64_t N, uint16_t* next_cells, float* reward, uint8_t* done,
                                    uint8_t* shows, int8_t* perf, void* stream);

/*
 * campx_reset_launch / campx_rollout_launch for a wide game.  The dynamic state of an
 * environment is its STATE INDEX: state.pos points at int32 [B] (4-byte aligned; the
 * positions, if wanted, are in the trace), state.pair_table is ignored.  out.trace is REQUIRED
 * and holds uint16 entries, [K, T, pitch] (reset: [K, pitch]) with pitch = out.scalar_pitch or
 * B, 2-byte aligned; frames are kept back to back (obs_t_stride == B*L*rows*cols,
 * board_t_stride == B*rows*cols) or, with strides of 0, only the last one.  B * L*rows*cols
 * must stay below 2^32 - 2^16.  16-bit observation formats as for campx_rollout_launch.
 * T = 1 is Engine.play().
 */
int32_t campx_wide_reset_launch(const CampxWideSpec* spec_host, const void* tables_dev,
                                CampxState state, CampxOutputs out, int64_t B, void* stream);
int32_t campx_wide_rollout_launch(const CampxWideSpec* spec_host, const void* tables_dev,
                                  CampxState state, const int8_t* actions, CampxOutputs out,
                                  int64_t B, int32_t T, int32_t reset_first, void* stream);

/* *bad_count (device int32, caller-zeroed) += number of ids outside 0..4 in
 * actions[0..n). */
int32_t campx_check_actions_launch(const int8_t* actions, int64_t n, int32_t* bad_count,
                                   void* stream);

/* Convert one-hot float actions [n, 5] (the reference's action format,
 * examples/boat_race.py:154-184) to ids [n]; rows that are not exactly one-hot
 * are counted in *bad_count (device int32, caller-zeroed; the reference asserts
 * sum(act) == 1, boat_race.py:48) and become id 5, which the step / rollout kernels
 * treat as "stay" and report through bad_count / bad_flag like any other bad id. */
int32_t campx_onehot_to_ids_launch(const float* onehot, int8_t* ids, int64_t n,
                                   int32_t* bad_count, void* stream);

/*
 * Settings: everything about the library's behaviour that is not an argument of a call.  ONE table
 * (campx_amd/csrc/campx_api.hip: name, default, range, what each selects; INTEGRATION.md lists
 * them).  A value is set for the process by campx_config_set() - thread-safe, takes effect with
 * the next call - or, before the library's first call, by the environment variable
 * CAMPX_CONFIG="name=value,name=value" (the library's only getenv).  campx_config_string() writes
 * "name=value name=value ..." of the EFFECTIVE values into `buf` (host, NUL-terminated, truncated to
 * buf_len) and returns the length needed including the NUL; a bench or an embedding application
 * records it beside its numbers.  campx_config_set / _get return CAMPX_EINVAL for a name that is
 * not in the table or a value outside its range.  No reference counterpart.
 */
int32_t campx_config_set(const char* name, int64_t value);
int32_t campx_config_get(const char* name, int64_t* value);
int32_t campx_config_string(char* buf, int32_t buf_len);

/* Measurement aid, not on the path (SURVEY section 8d "Bound": the measured achievable HBM write
 * bandwidth of the box, quoted next to the vendor peak): one launch that fills `n_bytes` (a
 * multiple of 16, `dst` 16-byte aligned, DEVICE memory) with the 32-bit `value` through the render
 * kernel's own store form - 16 bytes per lane, `sc0 sc1 nt`, one 2 KiB window per wave,
 * XCD-contiguous block order - and nothing else.  bench.py times it over the bytes a rollout
 * launch writes: `roofline.measured_write_ceiling_gbs`.  No reference counterpart. */
int32_t campx_write_probe_launch(void* dst, int64_t n_bytes, uint32_t value, void* stream);

const char* campx_strerror(int32_t code);
/* hipError_t of the most recent failed HIP call on this thread (0 if none). */
int32_t campx_last_hip_error(void);
/* "gfx950" etc. of device `ordinal`, written to buf (host); CAMPX_ENODEV if none. */
int32_t campx_device_arch(int32_t ordinal, char* buf, int32_t buf_len);

#ifdef __cplusplus
}
#endif
#endif /* CAMPX_HIP_H_ */
