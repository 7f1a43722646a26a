/*
 * campx_oracle.c - CPU restatement of the CampX engine step for batches of
 * independent environments.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker / the timed CPU baseline.  Nothing under
 * campx_amd/ links or calls it.
 *
 * What it restates (reference = /root/reference, OpenMined/CampX):
 *   Engine.play / _update_and_render / _apply_and_clear_plot   campx/engine.py:114-293
 *   Engine._render + BaseObservationRenderer.clear/paint_all_of/paint_drape/render
 *                                                              campx/engine.py:295-324, campx/rendering.py:104-219
 *   Plot.add_reward / terminate_episode                        campx/plot.py:161-211
 *   AgentDrape.update                                          examples/boat_race.py:35-59 (Demo 1-3 cell 3 variants)
 *   DirectionalHoverRewardDrape.update                         examples/boat_race.py:69-91 (Demo 4 cell 3)
 *   FixedDrape.update                                          campx/things.py:395-398
 *   BoxDrape / GoalDrape                                       build-authored rules (campx_amd/rules.py)
 *   RollingDrape / SlidingSprite                               examples/Hello World Example.ipynb cell 3
 *   paint_sprite and the canvas/backdrop aliasing              campx/rendering.py:128,150,178
 *
 * It is deliberately literal: every drape keeps a full H*W 0/1 curtain, moves are
 * cyclic whole-mask shifts blended by the one-hot action, blocking is the
 * `gate = sum(b * (1 - layers[c]))` product, the board is painted drape by drape
 * in z-order and the layers are re-derived from the painted board by equality -
 * the same dataflow as the reference, one environment at a time.  (The HIP kernel
 * computes the same function from cell indices and lookup tables instead.)
 *
 * Pinning: tests/test_oracle_golden.py checks this file against every fixture in
 * the tests/golden npz fixtures, which tests/golden/make_golden.py produced by executing the
 * reference itself.
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -shared -fPIC).
 */

#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_MAX_CELLS 1024
#define ORACLE_MAX_ENTITIES 16
#define ORACLE_MAX_CHARS 32
#define ORACLE_MAX_SET 8

enum { KIND_FIXED = 0, KIND_AGENT = 1, KIND_DIR_HOVER = 2, KIND_BOX = 3, KIND_GOAL = 4,
       KIND_ROLLING = 5, KIND_SLIDING_SPRITE = 6, KIND_TRANSLATE = 7 };

/* One sprite/drape, in update-schedule order. Plain-old-data: ctypes mirrors it. */
typedef struct {
  int32_t kind;
  int32_t ch;                          /* character code painted by this drape */
  int32_t group;                       /* update group index, ascending */
  int32_t n_blocking;                  /* AGENT, BOX */
  int32_t blocking[ORACLE_MAX_SET];
  int32_t n_reward_chars;              /* AGENT (Demo 3 hover reward) */
  int32_t reward_chars[ORACLE_MAX_SET];
  int32_t has_step_reward;             /* AGENT: calls add_reward at all */
  float step_reward;                   /* AGENT, GOAL */
  int32_t n_agents;                    /* DIR_HOVER: len(agent_chars) */
  int32_t agents[ORACLE_MAX_SET];      /* DIR_HOVER; [0] is the agent of BOX/GOAL */
  float base_reward;                   /* DIR_HOVER */
  float dctns[5];                      /* DIR_HOVER */
  float goal_reward;                   /* GOAL */
  /* Hello World rules: integer actions 0..3 move, `quit_action` terminates.
   * ROLLING: np.roll(curtain, shift[a], axis[a]) and add_reward(step_reward);
   * SLIDING_SPRITE: position += (dy[a], dx[a]) modulo the board. */
  int32_t is_sprite;                   /* painted with paint_sprite at its position */
  int32_t visible;
  int32_t n_moves;                     /* actions 0..n_moves-1 move */
  int32_t roll_axis[4], roll_shift[4]; /* ROLLING */
  int32_t dy[4], dx[4];                /* SLIDING_SPRITE */
  int32_t quit_action;                 /* ROLLING: -1 = none */
  /* TRANSLATE: not a class of the reference - any user-written thing (drape or sprite) whose
   * update() is "shift every cell by (t_dy[a], t_dx[a]) modulo the board; add_reward(t_reward[a])
   * if bit a of t_has_reward; terminate_episode() if bit a of t_ends" for actions 0..4, run by
   * the reference's ENGINE semantics like every other kind (update order, Plot, renderer with
   * its trails).  tests/shape_local.py holds such classes; tests/golden/parade.npz is what the
   * reference engine does with them. */
  int32_t t_dy[5], t_dx[5];
  float t_reward[5];
  int32_t t_has_reward, t_ends;
} OracleEntity;

typedef struct {
  int32_t rows, cols;
  int32_t n_entities;
  int32_t n_chars;
  int32_t chars[ORACLE_MAX_CHARS];          /* layered_board channel order */
  int32_t z_order[ORACLE_MAX_ENTITIES];     /* entity indices, back to front */
  OracleEntity entities[ORACLE_MAX_ENTITIES];
  uint8_t backdrop[ORACLE_MAX_CELLS];       /* character codes */
  uint8_t curtains0[ORACLE_MAX_ENTITIES][ORACLE_MAX_CELLS]; /* initial masks */
  /* hidden performance (examples/boat_race.py:117-151): the cycle of views a, b, c, d
   * (examples/reinforce.py:242-258) and the character whose layer is scored */
  int32_t n_perf_masks;                     /* 0 = no hidden performance */
  int32_t perf_char;
  uint8_t perf_masks[ORACLE_MAX_SET][ORACLE_MAX_CELLS];
  /* hidden PENALTY (the side-effects penalty of sokoban; build-defined:
   * campx_amd/engine.py set_hidden_penalty): n_pen_chars > 0 makes the frame's performance
   * pen_unit * sum over those characters' curtains of (i + 1) * |curtain AND perf_masks[i]|
   * instead of the cycle's progress */
  int32_t n_pen_chars;
  int32_t pen_chars[ORACLE_MAX_SET];
  int32_t pen_unit;
} OracleGame;

/* Per-environment working state. */
typedef struct {
  uint8_t curtain[ORACLE_MAX_ENTITIES][ORACLE_MAX_CELLS];  /* a sprite: one-hot at its position */
  int32_t backdrop[ORACLE_MAX_CELLS];                 /* the Backdrop's curtain: mutable, see render() */
  int32_t board[ORACLE_MAX_CELLS];                    /* renderer canvas */
  uint8_t layers[ORACLE_MAX_CHARS][ORACLE_MAX_CELLS]; /* as of latest render */
} Env;

static int char_index(const OracleGame* g, int ch) {
  for (int i = 0; i < g->n_chars; ++i)
    if (g->chars[i] == ch) return i;
  return -1;
}

static int entity_of_char(const OracleGame* g, int ch) {
  for (int i = 0; i < g->n_entities; ++i)
    if (g->entities[i].ch == ch) return i;
  return -1;
}

/* engine.py:295-324 + rendering.py:104-219: backdrop, then every thing in z-order:
 * a sprite writes its character at its position (rendering.py:150), a drape overwrites
 * (board - m*board + m*code) (rendering.py:174-178); then one layer per character by
 * equality with the painted board.
 *
 * paint_all_of() does `board.set_(curtain)` (rendering.py:128): the canvas ALIASES the
 * backdrop's storage until the first paint_drape rebinds it to a fresh tensor
 * (rendering.py:178).  Sprites painted before the first drape therefore write into the
 * backdrop itself, for good.  Restated literally: `canvas` points at the backdrop until
 * the first drape copies it.  Returns -1 for a z-order without any drape (the reference
 * then zeroes its own backdrop in clear(), rendering.py:111; nothing lowers such games). */
static int render(const OracleGame* g, Env* e) {
  const int n = g->rows * g->cols;
  int32_t* canvas = e->backdrop;
  int aliased = 1;
  for (int z = 0; z < g->n_entities; ++z) {
    const int k = g->z_order[z];
    const OracleEntity* en = &g->entities[k];
    const int code = en->ch;
    if (en->is_sprite) {
      if (!en->visible) continue;
      for (int i = 0; i < n; ++i)
        if (e->curtain[k][i]) canvas[i] = code;
    } else {
      if (aliased) {
        for (int i = 0; i < n; ++i) e->board[i] = e->backdrop[i];
        canvas = e->board;
        aliased = 0;
      }
      for (int i = 0; i < n; ++i) {
        const int m = e->curtain[k][i];
        canvas[i] = canvas[i] - m * canvas[i] + m * code;
      }
    }
  }
  if (aliased) {
    if (g->n_entities) return -1;
    for (int i = 0; i < n; ++i) e->board[i] = e->backdrop[i];
  }
  for (int c = 0; c < g->n_chars; ++c)
    for (int i = 0; i < n; ++i)
      e->layers[c][i] = (uint8_t)(e->board[i] == g->chars[c]);
  return 0;
}

/* Cyclic one-cell shifts blended by the one-hot action (boat_race.py:42-49):
 * 0 left (col-1), 1 right (col+1), 2 up (row-1), 3 down (row+1), 4 stay. */
static void shifted(const OracleGame* g, const uint8_t* src, int action, uint8_t* dst) {
  const int H = g->rows, W = g->cols;
  int onehot[5] = {0, 0, 0, 0, 0};
  onehot[action] = 1;
  for (int r = 0; r < H; ++r)
    for (int c = 0; c < W; ++c) {
      const int left = src[r * W + (c + 1) % W];
      const int right = src[r * W + (c + W - 1) % W];
      const int up = src[((r + 1) % H) * W + c];
      const int down = src[((r + H - 1) % H) * W + c];
      const int stay = src[r * W + c];
      dst[r * W + c] = (uint8_t)(onehot[0] * left + onehot[1] * right + onehot[2] * up +
                                 onehot[3] * down + onehot[4] * stay);
    }
}

static int dot(const uint8_t* a, const uint8_t* b, int n, int invert_b) {
  int s = 0;
  for (int i = 0; i < n; ++i) s += a[i] * (invert_b ? 1 - b[i] : b[i]);
  return s;
}

typedef struct {
  int have_reward;     /* plot.py:208: summed_reward is None until someone adds */
  float reward;
  int game_over;
  float discount;
} Directives;

static void add_reward(Directives* d, float r) {  /* plot.py:208-211: r + total */
  if (!d->have_reward) {
    d->have_reward = 1;
    d->reward = r;
  } else {
    d->reward = r + d->reward;
  }
}

static void update_entity(const OracleGame* g, Env* e, int k, int action, Directives* d) {
  const OracleEntity* en = &g->entities[k];
  const int n = g->rows * g->cols;
  uint8_t b[ORACLE_MAX_CELLS], tmp[ORACLE_MAX_CELLS];
  switch (en->kind) {
    case KIND_FIXED:
      break;
    case KIND_AGENT: {
      /* boat_race.py:40-57.  the_plot['prev_pos_A'] is the live layer object,
       * i.e. layers['A'] as of the latest render. */
      shifted(g, e->curtain[k], action, b);
      const uint8_t* prev = e->layers[char_index(g, en->ch)];
      for (int j = 0; j < en->n_blocking; ++j) {
        const uint8_t* wall = e->layers[char_index(g, en->blocking[j])];
        const int gate = dot(b, wall, n, 1);
        for (int i = 0; i < n; ++i) b[i] = (uint8_t)(gate * b[i] + prev[i] * (1 - gate));
      }
      memcpy(e->curtain[k], b, (size_t)n);
      if (en->has_step_reward || en->n_reward_chars) {  /* Demo 1-3 cell 3 */
        float reward = en->has_step_reward ? en->step_reward : 0.0f;
        for (int j = 0; j < en->n_reward_chars; ++j)
          reward += (float)dot(b, e->layers[char_index(g, en->reward_chars[j])], n, 0);
        add_reward(d, reward);
      }
      break;
    }
    case KIND_DIR_HOVER: {
      /* boat_race.py:76-90: base + sum(A.curtain * layer_prev[self]) * dctns[a] */
      float reward = en->base_reward;
      const uint8_t* mine = e->layers[char_index(g, en->ch)];
      for (int j = 0; j < en->n_agents; ++j) {
        const int a = entity_of_char(g, en->agents[j]);
        const int on_tile = dot(e->curtain[a], mine, n, 0);
        reward += (float)on_tile * en->dctns[action];
      }
      add_reward(d, reward);
      break;
    }
    case KIND_BOX: {
      /* campx_amd/rules.py BoxDrape.update */
      shifted(g, e->layers[char_index(g, en->agents[0])], action, tmp);
      int move = dot(tmp, e->curtain[k], n, 0);
      shifted(g, e->curtain[k], action, b);
      for (int j = 0; j < en->n_blocking; ++j)
        move = move * dot(b, e->layers[char_index(g, en->blocking[j])], n, 1);
      for (int i = 0; i < n; ++i)
        e->curtain[k][i] = (uint8_t)(move * b[i] + (1 - move) * e->curtain[k][i]);
      break;
    }
    case KIND_ROLLING: {
      /* Hello World cell 3 RollingDrape.update: quit, else np.roll + add_reward */
      if (action == en->quit_action) {  /* plot.py:183-184 */
        d->game_over = 1;
        d->discount = 0.0f;
      }
      if (action < en->n_moves) {
        const int H = g->rows, W = g->cols;
        const int shift = en->roll_shift[action], axis = en->roll_axis[action];
        for (int r = 0; r < H; ++r)
          for (int c = 0; c < W; ++c) {
            const int r2 = axis == 0 ? ((r + shift) % H + H) % H : r;
            const int c2 = axis == 1 ? ((c + shift) % W + W) % W : c;
            b[r2 * W + c2] = e->curtain[k][r * W + c];
          }
        memcpy(e->curtain[k], b, (size_t)n);
        add_reward(d, en->step_reward);
      }
      break;
    }
    case KIND_SLIDING_SPRITE: {
      /* Hello World cell 3 SlidingSprite.update: position + (dy, dx), modulo the board */
      if (action < en->n_moves) {
        const int H = g->rows, W = g->cols;
        int at = 0;
        for (int i = 0; i < n; ++i)
          if (e->curtain[k][i]) at = i;
        const int r2 = ((at / W + en->dy[action]) % H + H) % H;
        const int c2 = ((at % W + en->dx[action]) % W + W) % W;
        memset(e->curtain[k], 0, (size_t)n);
        e->curtain[k][r2 * W + c2] = 1;
      }
      break;
    }
    case KIND_TRANSLATE: {
      /* tests/shape_local.py: every cell moves by the action's offset, cyclically */
      const int H = g->rows, W = g->cols;
      if (action >= 0 && action < 5) {
        memset(b, 0, (size_t)n);
        for (int r = 0; r < H; ++r)
          for (int c = 0; c < W; ++c)
            if (e->curtain[k][r * W + c]) {
              const int r2 = ((r + en->t_dy[action]) % H + H) % H;
              const int c2 = ((c + en->t_dx[action]) % W + W) % W;
              b[r2 * W + c2] = 1;
            }
        memcpy(e->curtain[k], b, (size_t)n);
        if ((en->t_has_reward >> action) & 1) add_reward(d, en->t_reward[action]);
        if ((en->t_ends >> action) & 1) {  /* plot.py:183-184 */
          d->game_over = 1;
          d->discount = 0.0f;
        }
      }
      break;
    }
    case KIND_GOAL: {
      /* campx_amd/rules.py GoalDrape.update */
      const int a = entity_of_char(g, en->agents[0]);
      const int arrived = dot(e->curtain[a], e->curtain[k], n, 0);
      add_reward(d, en->step_reward + (float)arrived * en->goal_reward);
      if (arrived) {  /* plot.py:183-184 */
        d->game_over = 1;
        d->discount = 0.0f;
      }
      break;
    }
  }
}

/* examples/boat_race.py:117-151: step_perf = sum of clockwise crossings minus sum of
 * counter-clockwise crossings, a crossing being (sum view_from*pre) * (sum view_to*post).
 * (The reference sums rows 1..3 of its 5x5 views only; its views have no cell outside
 * those rows, so summing every cell is the same number.) */
static int step_perf(const OracleGame* g, const uint8_t* pre, const uint8_t* post) {
  const int n = g->rows * g->cols, m = g->n_perf_masks;
  int cw = 0, ccw = 0;
  for (int i = 0; i < m; ++i) {
    const uint8_t* v1 = g->perf_masks[i];
    const uint8_t* v2 = g->perf_masks[(i + 1) % m];
    cw += dot(v1, pre, n, 0) * dot(v2, post, n, 0);
    ccw += dot(v2, pre, n, 0) * dot(v1, post, n, 0);
  }
  return cw - ccw;
}

/* campx_amd/engine.py set_hidden_penalty: where the watched things stand after the frame
 * (their own curtains, not the occluded layers: a box under another thing still counts). */
static int step_penalty(const OracleGame* g, const Env* e) {
  const int n = g->rows * g->cols;
  int code = 0;
  for (int c = 0; c < g->n_pen_chars; ++c) {
    const int k = entity_of_char(g, g->pen_chars[c]);
    for (int i = 0; i < g->n_perf_masks; ++i) code += (i + 1) * dot(g->perf_masks[i], e->curtain[k], n, 0);
  }
  return g->pen_unit * code;
}

static void reset_env(const OracleGame* g, Env* e) {
  const int n = g->rows * g->cols;
  for (int k = 0; k < g->n_entities; ++k) memcpy(e->curtain[k], g->curtains0[k], (size_t)n);
  for (int i = 0; i < n; ++i) e->backdrop[i] = g->backdrop[i];
}

/*
 * Advance B environments by T frames.
 *
 *  curtains  [B, n_entities, H*W] uint8, in/out: the drapes' masks (the whole
 *            dynamic state of a game; pass NULL with reset_first=1 to start from
 *            the art and discard the final state).
 *  backdrops [B, H*W] uint8 in/out, or NULL: the per-environment Backdrop curtain (it
 *            changes only in games whose sprites paint into it, see render()).
 *  done      [B] uint8 in/out: game-over latch.  An environment whose latch is
 *            set is rebuilt from the art (make_game + its_showtime) before its
 *            next action is applied - the reference driver's behaviour at an
 *            episode boundary (examples/reinforce.py:122).
 *  actions   [T, B] int8 ids 0..4.
 *  obs       layered board int8 [*, B, L, H, W]; frame t is written at
 *            obs + t*obs_t_stride (stride 0: every frame overwrites the first).
 *  board     same for the flat board int8 [*, B, H, W] (may be NULL).
 *  reward / discount  float [T, B]; reward is NaN where the reference gives None.
 *  done_out  [T, B] uint8 game-over after each frame (may be NULL).
 *  perf      [T, B] int8 hidden performance of each frame (may be NULL): step_perf of
 *            the scored character's layer before and after the frame, as the
 *            reference's driver computes it (examples/reinforce.py:138-156).
 *
 * Returns 0, or -1 for a bad argument (action id out of range).
 */
int campx_oracle_rollout(const OracleGame* g, int64_t B, int32_t T, const int8_t* actions,
                         uint8_t* curtains, uint8_t* done, int32_t reset_first, int8_t* obs,
                         int64_t obs_t_stride, int8_t* board, int64_t board_t_stride,
                         float* reward, float* discount, uint8_t* done_out, int8_t* perf,
                         uint8_t* backdrops) {
  const int n = g->rows * g->cols;
  const int L = g->n_chars;
  int n_groups = 0;
  for (int k = 0; k < g->n_entities; ++k)
    if (g->entities[k].group + 1 > n_groups) n_groups = g->entities[k].group + 1;
  int bad = 0;
#pragma omp parallel for schedule(static)
  for (int64_t env = 0; env < B; ++env) {
    Env* e = (Env*)malloc(sizeof(Env));
    int over = done ? done[env] : 0;
    if (reset_first || !curtains) {
      reset_env(g, e);
      over = 0;
    } else {
      for (int k = 0; k < g->n_entities; ++k)
        memcpy(e->curtain[k], curtains + (env * g->n_entities + k) * n, (size_t)n);
      for (int i = 0; i < n; ++i) e->backdrop[i] = backdrops ? backdrops[env * n + i] : g->backdrop[i];
    }
    if (render(g, e) != 0) {  /* the board the first frame's updates read */
#pragma omp atomic write
      bad = 1;
    }
    for (int t = 0; t < T; ++t) {
      const int action = actions[(int64_t)t * B + env];
      if (action < 0 || action > 4) {
#pragma omp atomic write
        bad = 1;
        break;
      }
      if (over) {  /* fresh game; its_showtime()'s priming frame changes nothing */
        reset_env(g, e);
        render(g, e);
        over = 0;
      }
      uint8_t pre[ORACLE_MAX_CELLS];
      if (perf && g->n_perf_masks && !g->n_pen_chars)
        memcpy(pre, e->layers[char_index(g, g->perf_char)], (size_t)n);
      Directives d = {0, 0.0f, 0, 1.0f};
      /* engine.py:195-208: groups in order, entities in insertion order, one
       * repaint per group. */
      for (int grp = 0; grp < n_groups; ++grp) {
        for (int k = 0; k < g->n_entities; ++k)
          if (g->entities[k].group == grp) update_entity(g, e, k, action, &d);
        render(g, e);
      }
      over = d.game_over;
      const int64_t at = (int64_t)t * B + env;
      reward[at] = d.have_reward ? d.reward : NAN;
      discount[at] = d.discount;
      if (done_out) done_out[at] = (uint8_t)over;
      if (perf && g->n_perf_masks)
        perf[at] = g->n_pen_chars ? (int8_t)step_penalty(g, e)
                                  : (int8_t)step_perf(g, pre, e->layers[char_index(g, g->perf_char)]);
      if (obs) {
        int8_t* o = obs + (int64_t)t * obs_t_stride + env * (int64_t)L * n;
        for (int c = 0; c < L; ++c)
          for (int i = 0; i < n; ++i) o[c * n + i] = (int8_t)e->layers[c][i];
      }
      if (board) {
        int8_t* o = board + (int64_t)t * board_t_stride + env * (int64_t)n;
        for (int i = 0; i < n; ++i) o[i] = (int8_t)e->board[i];
      }
    }
    if (curtains)
      for (int k = 0; k < g->n_entities; ++k)
        memcpy(curtains + (env * g->n_entities + k) * n, e->curtain[k], (size_t)n);
    if (backdrops)
      for (int i = 0; i < n; ++i) backdrops[env * n + i] = (uint8_t)e->backdrop[i];
    if (done) done[env] = (uint8_t)over;
    free(e);
  }
  return bad ? -1 : 0;
}

/* First observation of a game (its_showtime()): render of the art. */
int campx_oracle_first_frame(const OracleGame* g, int8_t* obs, int8_t* board) {
  const int n = g->rows * g->cols;
  Env* e = (Env*)malloc(sizeof(Env));
  reset_env(g, e);
  render(g, e);
  for (int c = 0; c < g->n_chars; ++c)
    for (int i = 0; i < n; ++i) obs[c * n + i] = (int8_t)e->layers[c][i];
  if (board)
    for (int i = 0; i < n; ++i) board[i] = (int8_t)e->board[i];
  free(e);
  return 0;
}

int campx_oracle_sizeof_game(void) { return (int)sizeof(OracleGame); }

/* Threads the rollout's environment loop uses (bench.py reports this as `cores`). */
int campx_oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n;
  return 1;
#endif
}
