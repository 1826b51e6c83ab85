/*
 * azref.c -- CPU ORACLE for the AlphaZero.jl self-play hot path.
 *
 * THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load it.  The product (libazhip.so and the
 * azhip Python package) never links, imports or calls anything in this directory.
 *
 * PARITY UNPINNED: the reference cannot run in the build container (no Julia) and
 * its own tests hold no numeric result for this path.  The
 * restatement is pinned only against (i) SURVEY.md Appendix D's RNG-free vectors,
 * (ii) the PLSchedule known-answer in src/schedule.jl:82-87, (iii) the 6000 legal
 * Connect-Four positions in games/connect-four/benchmark/Test_L*_R* AND the exact
 * solver scores recorded beside them (azr_c4_solve: a negamax over this file's rules
 * reproduces all 1000 end-game scores of Test_L3_R1), and (iv) an independent
 * NumPy/PyTorch restatement (oracle/pyref.py).  The MCTS and network numerics remain
 * unpinned to the reference itself.
 *
 * Every function cites the reference lines it restates (paths relative to
 * /root/reference).  Arithmetic follows the reference's types: priors Float32,
 * W Float64, N Int64, UCT in Float64 evaluated left to right without contraction.
 * Randomness and transcendental functions follow the RNG / numerics contract of SURVEY.md §8c
 * (Julia's streams cannot be reproduced), implemented HERE in oracle/ref_numerics.h, separately
 * from the product's include/az_numerics.h (round 4: the two no longer share code).
 *
 * Build: see oracle/Makefile  (gcc -O3 -ffp-contract=off -mavx2 -mfma).
 */
#include <immintrin.h>
#include <math.h>
#include <omp.h>
#include <sys/mman.h>
#include <time.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ref_numerics.h"   /* the oracle's OWN implementation of the numerics + RNG contract (not the product's header) */

#define AZR_C4 0
#define AZR_TTT 1
#define AZR_MANCALA 2

#define AZR_AMAX 9      /* max #actions over the three games */
#define AZR_CELLS 42    /* max #cells in a state */

#define WHITE 1
#define BLACK 2

/* ------------------------------------------------------------------ states */
/* A state is the reference's `(board=..., curplayer=...)` named tuple, stored as
 * raw cells + player.  Connect-Four: cells[col + 7*row] in {0 EMPTY,1 WHITE,2 BLACK}
 * (games/connect-four/game.jl:11-22).  Tic-tac-toe: cells[pos] in {0 nothing,
 * 1 WHITE(true), 2 BLACK(false)} (games/tictactoe/game.jl:7-14).  Mancala:
 * cells[0..1] = stores, cells[2 + (player-1) + 2*(num-1)] = houses[player,num]
 * (games/mancala/game.jl:20-31). */
typedef struct {
  uint8_t cells[AZR_CELLS];
  uint8_t curplayer;
  uint8_t pad;
} azr_state;

typedef struct {
  int game;
  azr_state s;
  uint8_t finished;
  uint8_t winner;
  uint8_t amask[AZR_AMAX];
} azr_env;

static int azr_num_actions_(int game) { return game == AZR_C4 ? 7 : game == AZR_TTT ? 9 : 6; }
int azr_num_actions(int game) { return azr_num_actions_(game); }

static uint8_t other(uint8_t p) { return (uint8_t)(3 - p); }

/* ============================ Connect Four ============================== */
#define C4_COLS 7
#define C4_ROWS 6
#define C4(b, col, row) ((b)[(col) + C4_COLS * (row)]) /* 0-based col,row */

/* games/connect-four/game.jl:87-93 */
static int c4_first_free(const uint8_t* b, int col) {
  int row = 0;
  while (row < C4_ROWS && C4(b, col, row) != 0) row++;
  return row; /* == C4_ROWS when the column is full */
}
/* games/connect-four/game.jl:95-99 */
static void c4_update_actions_mask(azr_env* g) {
  for (int col = 0; col < C4_COLS; ++col) g->amask[col] = c4_first_free(g->s.cells, col) < C4_ROWS;
}
static int c4_valid(int col, int row) { return col >= 0 && col < C4_COLS && row >= 0 && row < C4_ROWS; }
/* games/connect-four/game.jl:105-114 */
static int c4_num_connected_dir(const uint8_t* b, uint8_t player, int col, int row, int dc, int dr) {
  int n = 0;
  col += dc; row += dr;
  while (c4_valid(col, row) && C4(b, col, row) == player) { n++; col += dc; row += dr; }
  return n;
}
/* games/connect-four/game.jl:116-127 */
static int c4_winning_pattern_at(const uint8_t* b, uint8_t player, int col, int row) {
  static const int axes[4][2] = {{1, 1}, {1, -1}, {1, 0}, {0, 1}};
  for (int a = 0; a < 4; ++a) {
    int n = 1 + c4_num_connected_dir(b, player, col, row, axes[a][0], axes[a][1]) +
            c4_num_connected_dir(b, player, col, row, -axes[a][0], -axes[a][1]);
    if (n >= 4) return 1;
  }
  return 0;
}
static int any_mask(const azr_env* g) {
  int n = azr_num_actions_(g->game);
  for (int i = 0; i < n; ++i) if (g->amask[i]) return 1;
  return 0;
}
/* games/connect-four/game.jl:39-48 */
static void c4_init(azr_env* g) {
  memset(g, 0, sizeof *g);
  g->game = AZR_C4;
  g->s.curplayer = WHITE;
  for (int c = 0; c < C4_COLS; ++c) g->amask[c] = 1;
}
/* games/connect-four/game.jl:50-68 (only the TOP stone of each column is tested) */
static void c4_set_state(azr_env* g, const azr_state* st) {
  g->s = *st;
  c4_update_actions_mask(g);
  if (!any_mask(g)) g->finished = 1;
  for (int col = 0; col < C4_COLS; ++col) {
    int top = c4_first_free(g->s.cells, col);
    if (top == 0) continue;
    int row = top - 1;
    uint8_t c = C4(st->cells, col, row);
    if (c != 0 && c4_winning_pattern_at(st->cells, c, col, row)) {
      g->winner = c; g->finished = 1; break;
    }
  }
}
/* games/connect-four/game.jl:130-146 */
static void c4_play(azr_env* g, int col) {
  int row = c4_first_free(g->s.cells, col);
  C4(g->s.cells, col, row) = g->s.curplayer;
  c4_update_actions_mask(g);
  if (c4_winning_pattern_at(g->s.cells, g->s.curplayer, col, row)) {
    g->winner = g->s.curplayer; g->finished = 1;
  } else {
    g->finished = !any_mask(g);
  }
  g->s.curplayer = other(g->s.curplayer);
}
/* games/connect-four/game.jl:160-168 */
static double c4_white_reward(const azr_env* g) {
  if (g->finished) {
    if (g->winner == WHITE) return 1.;
    if (g->winner == BLACK) return -1.;
    return 0.;
  }
  return 0.;
}
/* games/connect-four/game.jl:226-241: Float32[board[col,row]==c for col,row,c in (EMPTY,WHITE,BLACK)]
 * after swapping colours when black is to move; column-major 7x6x3. */
static void c4_vectorize(const azr_state* st, float* out) {
  for (int ch = 0; ch < 3; ++ch)
    for (int row = 0; row < C4_ROWS; ++row)
      for (int col = 0; col < C4_COLS; ++col) {
        uint8_t c = C4(st->cells, col, row);
        if (st->curplayer != WHITE && c != 0) c = other(c);
        out[col + C4_COLS * (row + C4_ROWS * ch)] = (c == ch) ? 1.0f : 0.0f;
      }
}

/* ============================= Tic-tac-toe ============================== */
/* games/tictactoe/game.jl:42-58; pos = (y-1)*3 + x */
static const int TTT_AL[8][3] = {
    {0, 3, 6}, {1, 4, 7}, {2, 5, 8},   /* [(i,j) for j] for i : x fixed           */
    {0, 1, 2}, {3, 4, 5}, {6, 7, 8},   /* [(i,j) for i] for j : y fixed           */
    {0, 4, 8}, {2, 4, 6}};
static int ttt_has_won(const azr_env* g, uint8_t player) {
  for (int a = 0; a < 8; ++a)
    if (g->s.cells[TTT_AL[a][0]] == player && g->s.cells[TTT_AL[a][1]] == player &&
        g->s.cells[TTT_AL[a][2]] == player) return 1;
  return 0;
}
static void ttt_refresh(azr_env* g) { /* actions_mask = map(isnothing, board), :69 */
  for (int i = 0; i < 9; ++i) g->amask[i] = g->s.cells[i] == 0;
  /* terminal_white_reward, :75-82 */
  g->finished = 1;
  if (ttt_has_won(g, WHITE)) g->winner = WHITE;
  else if (ttt_has_won(g, BLACK)) g->winner = BLACK;
  else if (!any_mask(g)) g->winner = 0;
  else g->finished = 0;
}
static void ttt_init(azr_env* g) {
  memset(g, 0, sizeof *g);
  g->game = AZR_TTT;
  g->s.curplayer = WHITE;
  ttt_refresh(g);
}
static void ttt_set_state(azr_env* g, const azr_state* st) { g->s = *st; g->winner = 0; ttt_refresh(g); }
/* games/tictactoe/game.jl:89-92 (no terminal guard) */
static void ttt_play(azr_env* g, int pos) {
  g->s.cells[pos] = g->s.curplayer;
  g->s.curplayer = other(g->s.curplayer);
  g->winner = 0;
  ttt_refresh(g);
}
static double ttt_white_reward(const azr_env* g) {
  if (!g->finished) return 0.;
  return g->winner == WHITE ? 1. : g->winner == BLACK ? -1. : 0.;
}
/* games/tictactoe/game.jl:126-143: 3x3x3, x fastest, channels (nothing, WHITE, BLACK) */
static void ttt_vectorize(const azr_state* st, float* out) {
  for (int ch = 0; ch < 3; ++ch)
    for (int pos = 0; pos < 9; ++pos) {
      uint8_t c = st->cells[pos];
      if (st->curplayer != WHITE && c != 0) c = other(c);
      out[pos + 9 * ch] = (c == ch) ? 1.0f : 0.0f;
    }
}

/* =============================== Mancala ================================ */
#define MH 6
#define M_STORE(b, p) ((b)[(p) - 1])
#define M_HOUSE(b, p, n) ((b)[2 + ((p) - 1) + 2 * ((n) - 1)]) /* 1-based player, num */
typedef struct { int is_store; int player; int num; } mpos;
/* games/mancala/game.jl:80-97 */
static mpos m_next_pos(mpos pos, int player) {
  mpos r = {0, 0, 0};
  if (pos.is_store) { r.is_store = 0; r.player = 3 - player; r.num = MH; return r; }
  if (pos.num > 1) { r.player = pos.player; r.num = pos.num - 1; return r; }
  if (pos.player == player) { r.is_store = 1; r.player = player; return r; }
  r.player = player; r.num = MH; return r;
}
static uint8_t m_read(const uint8_t* b, mpos p) { return p.is_store ? M_STORE(b, p.player) : M_HOUSE(b, p.player, p.num); }
static void m_write(uint8_t* b, mpos p, uint8_t v) { if (p.is_store) M_STORE(b, p.player) = v; else M_HOUSE(b, p.player, p.num) = v; }
static int m_sum_houses(const uint8_t* b, int player) {
  int s = 0;
  for (int n = 1; n <= MH; ++n) s += M_HOUSE(b, player, n);
  return s;
}
/* games/mancala/game.jl:137-142 */
static void m_capture_leftovers(uint8_t* b, int player) {
  M_STORE(b, player) = (uint8_t)(M_STORE(b, player) + m_sum_houses(b, player));
  for (int p = 1; p <= 2; ++p) for (int n = 1; n <= MH; ++n) M_HOUSE(b, p, n) = 0;
}
static void m_refresh_mask(azr_env* g) { /* :121-123 */
  for (int n = 1; n <= MH; ++n) g->amask[n - 1] = M_HOUSE(g->s.cells, g->s.curplayer, n) > 0;
}
static void m_init(azr_env* g) {
  memset(g, 0, sizeof *g);
  g->game = AZR_MANCALA;
  g->s.curplayer = WHITE;
  for (int p = 1; p <= 2; ++p) for (int n = 1; n <= MH; ++n) M_HOUSE(g->s.cells, p, n) = 3;
  m_refresh_mask(g);
}
/* games/mancala/game.jl:54-60 */
static void m_set_state(azr_env* g, const azr_state* st) {
  g->s = *st;
  if (m_sum_houses(g->s.cells, g->s.curplayer) == 0 || m_sum_houses(g->s.cells, 3 - g->s.curplayer) == 0)
    g->finished = 1;
  m_refresh_mask(g);
}
/* games/mancala/game.jl:144-177 */
static void m_play(azr_env* g, int a /*0-based*/) {
  uint8_t* b = g->s.cells;
  int cur = g->s.curplayer;
  mpos pos = {0, cur, a + 1};
  int nseeds = m_read(b, pos);
  m_write(b, pos, 0);
  for (int i = 0; i < nseeds; ++i) {
    pos = m_next_pos(pos, cur);
    m_write(b, pos, (uint8_t)(m_read(b, pos) + 1));
  }
  if (m_sum_houses(b, cur) == 0) {
    m_capture_leftovers(b, 3 - cur);
    g->finished = 1;
  } else if (!pos.is_store) {
    if (m_read(b, pos) == 1 && cur == pos.player) {
      /* capture_last_and_opposite, :125-133 */
      mpos opp = {0, 3 - pos.player, MH - pos.num + 1};
      M_STORE(b, pos.player) = (uint8_t)(M_STORE(b, pos.player) + m_read(b, opp) + 1);
      m_write(b, pos, 0);
      m_write(b, opp, 0);
      if (m_sum_houses(b, 3 - cur) == 0) { m_capture_leftovers(b, cur); g->finished = 1; m_refresh_mask(g); return; }
      if (m_sum_houses(b, cur) == 0) { m_capture_leftovers(b, 3 - cur); g->finished = 1; m_refresh_mask(g); return; }
    }
    g->s.curplayer = (uint8_t)(3 - cur);
  }
  m_refresh_mask(g);
}
/* games/mancala/game.jl:187-206 */
static double m_white_reward(const azr_env* g) {
  if (!g->finished) return 0.;
  int nw = M_STORE(g->s.cells, 1), nb = M_STORE(g->s.cells, 2);
  return nw > nb ? 1. : nw < nb ? -1. : 0.;
}
/* games/mancala/game.jl:224-257.  BUG-COMPATIBLE: flip_colors ignores its argument and
 * returns the INITIAL board (stores 0, houses 3), so when black is to move the network
 * sees the initial position.  14x1x5, positions: white houses 6..1, white store, black
 * houses 6..1, black store; channels nstones, whouse, wstore, bhouse, bstore. */
static void m_vectorize(const azr_state* st, float* out) {
  uint8_t init[AZR_CELLS];
  memset(init, 0, sizeof init);
  for (int p = 1; p <= 2; ++p) for (int n = 1; n <= MH; ++n) M_HOUSE(init, p, n) = 3;
  const uint8_t* b = st->curplayer == WHITE ? st->cells : init;
  for (int i = 0; i < 14; ++i) {
    int player = i < 7 ? 1 : 2, j = i % 7;
    int is_store = (j == 6);
    float nst = is_store ? (float)M_STORE(b, player) : (float)M_HOUSE(b, player, MH - j);
    out[i + 14 * 0] = nst;
    out[i + 14 * 1] = (!is_store && player == 1) ? 1.f : 0.f;
    out[i + 14 * 2] = (is_store && player == 1) ? 1.f : 0.f;
    out[i + 14 * 3] = (!is_store && player == 2) ? 1.f : 0.f;
    out[i + 14 * 4] = (is_store && player == 2) ? 1.f : 0.f;
  }
}

/* ========================= GameInterface dispatch ======================= */
/* src/game.jl:34-336 -- the subset the path uses (SURVEY.md §8b seam 4). */
/* Test infrastructure only: exact value of a Connect-Four position, in the convention of the second column of the
 * reference's games/connect-four/benchmark/Test_L*_R* files (scripts/pons_benchmark.jl:49-98 replays them; the scores are
 * John Tromp / Pascal Pons solver results): 0 = draw, +k = the player to move wins with his k-th stone counted from
 * his last one (22 - stones he has played when he connects four), -k = the opponent does.  Plain negamax with
 * alpha-beta over THIS file's rules -- c4_play, its win test, its actions mask, its termination -- so that agreeing with
 * the recorded scores pins those rules (legal moves, four-in-a-row in all directions, draw on a full board) to data the
 * reference ships.  `moves`: 0-based columns from the empty board.  Returns the score, or 99 if the position is illegal /
 * already finished, or 98 if more than node_limit nodes were visited. */
static int c4_negamax(const azr_env* g, int nstones, int alpha, int beta, long long* nodes, long long limit) {
  if (++*nodes > limit) return 98;
  static const int order[C4_COLS] = {3, 2, 4, 1, 5, 0, 6};
  for (int k = 0; k < C4_COLS; ++k) {                 /* a winning move now: the mover connects with stone nstones/2 + 1 */
    int col = order[k];
    if (!g->amask[col]) continue;
    azr_env c = *g;
    c4_play(&c, col);
    if (c.finished && c.winner != 0) return (C4_COLS * C4_ROWS + 1 - nstones) / 2;
  }
  if (nstones + 1 == C4_COLS * C4_ROWS) return 0;     /* the last cell, no win: draw */
  int max = (C4_COLS * C4_ROWS - 1 - nstones) / 2;    /* cannot win before the move after next */
  if (beta > max) { beta = max; if (alpha >= beta) return beta; }
  for (int k = 0; k < C4_COLS; ++k) {
    int col = order[k];
    if (!g->amask[col]) continue;
    azr_env c = *g;
    c4_play(&c, col);
    if (c.finished) { if (0 > alpha) alpha = 0; if (alpha >= beta) return alpha; continue; }   /* full board (a win was caught above) */
    int sc = c4_negamax(&c, nstones + 1, -beta, -alpha, nodes, limit);
    if (sc == 98) return 98;
    sc = -sc;
    if (sc >= beta) return sc;
    if (sc > alpha) alpha = sc;
  }
  return alpha;
}
int azr_c4_solve(const int* moves, int nmoves, long long node_limit, long long* nodes_out) {
  azr_env g;
  c4_init(&g);
  for (int i = 0; i < nmoves; ++i) {
    if (g.finished || moves[i] < 0 || moves[i] >= C4_COLS || !g.amask[moves[i]]) return 99;
    c4_play(&g, moves[i]);
  }
  if (g.finished) return 99;
  long long nodes = 0;
  int sc = c4_negamax(&g, nmoves, -(C4_COLS * C4_ROWS) / 2, (C4_COLS * C4_ROWS) / 2, &nodes, node_limit);
  if (nodes_out) *nodes_out = nodes;
  return sc;
}

void azr_init(azr_env* g, int game) {
  if (game == AZR_C4) c4_init(g); else if (game == AZR_TTT) ttt_init(g); else m_init(g);
}
void azr_init_state(azr_env* g, int game, const azr_state* st) { /* GI.init(gspec, state) */
  azr_init(g, game);
  if (game == AZR_C4) c4_set_state(g, st); else if (game == AZR_TTT) ttt_set_state(g, st); else m_set_state(g, st);
}
void azr_play(azr_env* g, int a) {
  if (g->game == AZR_C4) c4_play(g, a); else if (g->game == AZR_TTT) ttt_play(g, a); else m_play(g, a);
}
double azr_white_reward(const azr_env* g) {
  return g->game == AZR_C4 ? c4_white_reward(g) : g->game == AZR_TTT ? ttt_white_reward(g) : m_white_reward(g);
}
int azr_terminated(const azr_env* g) { return g->finished; }
int azr_white_playing(const azr_env* g) { return g->s.curplayer == WHITE; }
void azr_actions_mask(const azr_env* g, uint8_t* mask) { memcpy(mask, g->amask, (size_t)azr_num_actions_(g->game)); }
void azr_current_state(const azr_env* g, azr_state* st) { *st = g->s; }
void azr_state_dims(int game, int* w, int* h, int* c) {
  if (game == AZR_C4) { *w = 7; *h = 6; *c = 3; } else if (game == AZR_TTT) { *w = 3; *h = 3; *c = 3; } else { *w = 14; *h = 1; *c = 5; }
}
void azr_vectorize_state(int game, const azr_state* st, float* out) {
  if (game == AZR_C4) c4_vectorize(st, out); else if (game == AZR_TTT) ttt_vectorize(st, out); else m_vectorize(st, out);
}
/* available_actions (src/game.jl:318-321): indices of set mask bits, in action order */
static int available(const azr_env* g, int* acts) {
  int n = 0, A = azr_num_actions_(g->game);
  for (int a = 0; a < A; ++a) if (g->amask[a]) acts[n++] = a;
  return n;
}

/* Packed 16-byte state key shared with the C ABI (include/azhip.h "state keys"). */
void azr_pack_key(int game, const azr_state* st, uint64_t key[2]) {
  uint64_t a = 0, b = 0;
  if (game == AZR_C4) {
    for (int col = 0; col < 7; ++col) for (int row = 0; row < 6; ++row) {
      uint8_t c = C4(st->cells, col, row);
      if (c == WHITE) a |= 1ULL << (col * 7 + row);
      if (c == BLACK) b |= 1ULL << (col * 7 + row);
    }
  } else if (game == AZR_TTT) {
    for (int p = 0; p < 9; ++p) {
      if (st->cells[p] == WHITE) a |= 1ULL << p;
      if (st->cells[p] == BLACK) b |= 1ULL << p;
    }
  } else {
    for (int n = 1; n <= MH; ++n) {
      a |= (uint64_t)M_HOUSE(st->cells, 1, n) << (8 * (n - 1));
      b |= (uint64_t)M_HOUSE(st->cells, 2, n) << (8 * (n - 1));
    }
    a |= (uint64_t)M_STORE(st->cells, 1) << 48;
    b |= (uint64_t)M_STORE(st->cells, 2) << 48;
  }
  if (st->curplayer == BLACK) a |= 1ULL << 63;
  key[0] = a; key[1] = b;
}
void azr_unpack_key(int game, const uint64_t key[2], azr_state* st) {
  memset(st, 0, sizeof *st);
  uint64_t a = key[0], b = key[1];
  st->curplayer = (a >> 63) ? BLACK : WHITE;
  if (game == AZR_C4) {
    for (int col = 0; col < 7; ++col) for (int row = 0; row < 6; ++row) {
      int bit = col * 7 + row;
      C4(st->cells, col, row) = ((a >> bit) & 1) ? WHITE : ((b >> bit) & 1) ? BLACK : 0;
    }
  } else if (game == AZR_TTT) {
    for (int p = 0; p < 9; ++p) st->cells[p] = ((a >> p) & 1) ? WHITE : ((b >> p) & 1) ? BLACK : 0;
  } else {
    for (int n = 1; n <= MH; ++n) {
      M_HOUSE(st->cells, 1, n) = (uint8_t)(a >> (8 * (n - 1)));
      M_HOUSE(st->cells, 2, n) = (uint8_t)(b >> (8 * (n - 1)));
    }
    M_STORE(st->cells, 1) = (uint8_t)(a >> 48);
    M_STORE(st->cells, 2) = (uint8_t)(b >> 48);
  }
}

/* ================================ Network ================================ */
/* src/networks/architectures/resnet.jl:53-92 in test mode, fp32.
 * Parameter blob = fp32 arrays in Flux layout, in this order (include/azhip.h):
 *   stem   conv W(kw,kh,Cin,F) b(F)  bn g,b,mu,var (F each)
 *   block  x num_blocks: conv1 W(3,3,F,F) b bn(4F)  conv2 W b bn(4F)
 *   phead  conv W(1,1,F,npf) b bn(4npf)  dense W(A, P*npf) b(A)
 *   vhead  conv W(1,1,F,nvf) b bn(4nvf)  dense1 W(F, P*nvf) b(F)  dense2 W(1,F) b(1)
 * Conv = true convolution (flipped kernel), pad 1 (NNlib default -- from memory, not in
 * tree); BN eps = 1f-5 (Flux default -- from memory).
 * SUMMATION ORDER (the fp32 contract, include/azhip.h "fp32 contract"): every output is
 * one fp32 fma chain starting from +0:
 *   3x3 conv, Cin even : taps t=0..8 ((dy,dx) = (t/3-1, t%3-1)), inside a tap channels in
 *                        the order c = j, Cin/2 + j  for j = 0..Cin/2-1
 *   stem (Cin = planes): k = t*Cin + c, K = 9 Cin, in the order k = j, K2 + j for j = 0..K2-1,
 *                        K2 = ceil(K/2) (k = K is a zero pad when K is odd)
 *   1x1 conv           : c = j, Cin/2 + j
 *   dense              : k = p*nf + f ascending (p = x + W*y board position, f = filter)
 * then y = fma(acc, scale, shift) with scale = g / sqrtf(var + eps),
 * shift = fma(b - mu, scale, beta); dense: y = acc + bias. */
typedef struct {
  int W, H, C, A;            /* board dims, planes, actions */
  int nblocks, F, npf, nvf;
  const float* blob;
  float* pk;                 /* net_pack(blob), built lazily */
} azr_net;

static size_t net_nparams(const azr_net* n) {
  size_t P = (size_t)n->W * n->H, F = n->F;
  size_t s = 9 * (size_t)n->C * F + F + 4 * F;
  s += (size_t)n->nblocks * 2 * (9 * F * F + F + 4 * F);
  s += F * n->npf + n->npf + 4 * (size_t)n->npf + (size_t)n->A * P * n->npf + n->A;
  s += F * n->nvf + n->nvf + 4 * (size_t)n->nvf + F * P * n->nvf + F + F + 1;
  return s;
}
size_t azr_net_num_params(int W, int H, int C, int A, int nblocks, int F, int npf, int nvf) {
  azr_net n = {W, H, C, A, nblocks, F, npf, nvf, 0, 0};
  return net_nparams(&n);
}

static void bn_fold(const float* bias, const float* bn, int n, float* scale, float* shift) {
  const float *g = bn, *be = bn + n, *mu = bn + 2 * n, *var = bn + 3 * n;
  for (int i = 0; i < n; ++i) {
    scale[i] = g[i] / sqrtf(var[i] + 1e-5f);
    shift[i] = __builtin_fmaf(bias[i] - mu[i], scale[i], be[i]);
  }
}

/* Flux conv weight W[i + k*(j + k*(ci + Cin*co))] -> [tap][ci][co] (co fastest) so the inner loop runs
 * over output channels; tap t = (dy+1)*3 + (dx+1) reads W[i = 1-dx, j = 1-dy] (true convolution). */
static void pack_conv_w(const float* Wt, int ksz, int Cin, int Cout, float* dst) {
  int ntap = ksz * ksz;
  for (int t = 0; t < ntap; ++t) {
    int dy = ksz == 3 ? t / 3 - 1 : 0, dx = ksz == 3 ? t % 3 - 1 : 0;
    int wi = ksz == 3 ? 1 - dx : 0, wj = ksz == 3 ? 1 - dy : 0;
    for (int ci = 0; ci < Cin; ++ci) for (int co = 0; co < Cout; ++co)
      dst[((size_t)t * Cin + ci) * Cout + co] = Wt[(size_t)wi + (size_t)ksz * (wj + (size_t)ksz * (ci + (size_t)Cin * co))];
  }
}

/* A position's output channel: o[co] = the chain a = fma(vk[k], wk[k][co], a) over k = 0 .. K-1 IN THE ORDER GIVEN, from a = 0 --
 * the sequence of single-rounding fmas of the scalar loop this replaces (for every k: o[co] = fmaf(v, w[co], o[co])).  Channels and
 * positions are independent of each other, so eight channels per AVX register, 16 per pass and FOUR positions at once change no
 * bit: they keep the accumulators in registers instead of storing and reloading o[] for every k, and a weight row fetched once
 * (from L2: a layer's weights are 147 KB or more) serves four positions (2.2 -> 0.64 ms per 5x64 board and thread).  Channels beyond
 * a multiple of 16 take the scalar chain. */
static void fma_chain4(const float* v0, const float* v1, const float* v2, const float* v3, const float* const* wk, int K, int Cout,
                       float* o0, float* o1, float* o2, float* o3) {
  int cb = 0;
  for (; cb + 16 <= Cout; cb += 16) {
    __m256 a00 = _mm256_setzero_ps(), a01 = a00, a10 = a00, a11 = a00, a20 = a00, a21 = a00, a30 = a00, a31 = a00;
    for (int k = 0; k < K; ++k) {
      const float* w = wk[k] + cb;
      const __m256 w0 = _mm256_loadu_ps(w), w1 = _mm256_loadu_ps(w + 8);
      __m256 v = _mm256_broadcast_ss(v0 + k); a00 = _mm256_fmadd_ps(v, w0, a00); a01 = _mm256_fmadd_ps(v, w1, a01);
      v = _mm256_broadcast_ss(v1 + k);        a10 = _mm256_fmadd_ps(v, w0, a10); a11 = _mm256_fmadd_ps(v, w1, a11);
      v = _mm256_broadcast_ss(v2 + k);        a20 = _mm256_fmadd_ps(v, w0, a20); a21 = _mm256_fmadd_ps(v, w1, a21);
      v = _mm256_broadcast_ss(v3 + k);        a30 = _mm256_fmadd_ps(v, w0, a30); a31 = _mm256_fmadd_ps(v, w1, a31);
    }
    _mm256_storeu_ps(o0 + cb, a00); _mm256_storeu_ps(o0 + cb + 8, a01); _mm256_storeu_ps(o1 + cb, a10); _mm256_storeu_ps(o1 + cb + 8, a11);
    _mm256_storeu_ps(o2 + cb, a20); _mm256_storeu_ps(o2 + cb + 8, a21); _mm256_storeu_ps(o3 + cb, a30); _mm256_storeu_ps(o3 + cb + 8, a31);
  }
  if (cb < Cout) {
    /* the remaining channels, one position at a time */
    const float* vv[4] = {v0, v1, v2, v3}; float* oo[4] = {o0, o1, o2, o3};
    for (int q = 0; q < 4; ++q)
      for (int c = cb; c < Cout; ++c) {
        float acc = 0.0f;
        for (int k = 0; k < K; ++k) acc = __builtin_fmaf(vv[q][k], wk[k][c], acc);
        oo[q][c] = acc;
      }
  }
}

/* in: [P][Cin] (position-major, channel fastest); out: [P][Cout]; Wp packed by pack_conv_w. */
static void conv_bn(const azr_net* n, const float* in, int Cin, int Cout, int ksz, const float* Wp,
                    const float* bias, const float* bn, const float* res, int relu, int paired, float* out) {
  int W = n->W, H = n->H;
  const int ntap = ksz * ksz, K = ntap * Cin;
  float* scale = malloc(sizeof(float) * (2 * (size_t)Cout + 4 * (size_t)K));
  float* shift = scale + Cout;
  float* vk = shift + Cout;                              /* four positions' K input values each, in summation order ... */
  const float** wk = malloc(sizeof(const float*) * (size_t)K);   /* ... and the weight row [Cout] each of them multiplies */
  bn_fold(bias, bn, Cout, scale, shift);
  int half = Cin / 2;
  /* the summation order of a position's K terms -- the same for every position: (tap, input channel) of term k */
  int* tk = malloc(sizeof(int) * 2 * (size_t)K);
  int* ck = tk + K;
  int nk = 0;
  if (paired == 2) {                        /* stem: pairing over the whole K = 9 Cin */
    int K2 = (K + 1) / 2;
    for (int s2 = 0; s2 < 2 * K2; ++s2) {
      int k = (s2 & 1) ? K2 + (s2 >> 1) : (s2 >> 1);
      if (k >= K) continue;
      tk[nk] = k / Cin; ck[nk++] = k % Cin;
    }
  } else {
    for (int t = 0; t < ntap; ++t) for (int kk = 0; kk < Cin; ++kk) {
      tk[nk] = t; ck[nk++] = paired ? ((kk & 1) ? half + (kk >> 1) : (kk >> 1)) : kk;
    }
  }
  for (int k = 0; k < nk; ++k) wk[k] = Wp + ((size_t)tk[k] * Cin + ck[k]) * Cout;
  const int NP = W * H;
  float* dummy = malloc(sizeof(float) * (size_t)Cout);           /* output of the positions that pad the last group of four */
  for (int p0 = 0; p0 < NP; p0 += 4) {
    float* og[4];
    for (int q = 0; q < 4; ++q) {
      const int pz = p0 + q < NP ? p0 + q : p0;                    /* (a padding position repeats the group's first one) */
      const int x = pz % W, y = pz / W;
      og[q] = p0 + q < NP ? out + (size_t)pz * Cout : dummy;
      const float* tap_in[9];               /* the input row [Cin] a tap reads at this position, NULL outside the board (zeros) */
      for (int t = 0; t < ntap; ++t) {
        int dy = ksz == 3 ? t / 3 - 1 : 0, dx = ksz == 3 ? t % 3 - 1 : 0;
        int yy = y + dy, xx = x + dx;
        tap_in[t] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? in + (size_t)(xx + W * yy) * Cin : 0;
      }
      float* vq = vk + (size_t)q * K;
      for (int k = 0; k < nk; ++k) vq[k] = tap_in[tk[k]] ? tap_in[tk[k]][ck[k]] : 0.0f;
    }
    fma_chain4(vk, vk + K, vk + 2 * (size_t)K, vk + 3 * (size_t)K, wk, nk, Cout, og[0], og[1], og[2], og[3]);
    for (int q = 0; q < 4 && p0 + q < NP; ++q) {
      float* o = og[q];
      const size_t pz = (size_t)(p0 + q);
      for (int co = 0; co < Cout; ++co) {
        float v = __builtin_fmaf(o[co], scale[co], shift[co]);
        if (res) v = v + res[pz * Cout + co];
        if (relu) v = v > 0.0f ? v : 0.0f;
        o[co] = v;
      }
    }
  }
  free(dummy);
  free(scale); free(tk); free((void*)wk);
}

/* copy of the blob with every conv kernel re-ordered by pack_conv_w and the value head's first dense
 * matrix transposed to [k][o]; everything else unchanged (same offsets). */
static float* net_pack(const azr_net* n) {
  size_t np = net_nparams(n);
  float* pk = malloc(sizeof(float) * np);
  memcpy(pk, n->blob, sizeof(float) * np);
  int P = n->W * n->H, F = n->F, C = n->C;
  size_t off = 0;
  pack_conv_w(n->blob + off, 3, C, F, pk + off); off += 9 * (size_t)C * F + 5 * (size_t)F;
  for (int l = 0; l < 2 * n->nblocks; ++l) { pack_conv_w(n->blob + off, 3, F, F, pk + off); off += 9 * (size_t)F * F + 5 * (size_t)F; }
  pack_conv_w(n->blob + off, 1, F, n->npf, pk + off); off += (size_t)F * n->npf + 5 * (size_t)n->npf;
  off += (size_t)n->A * P * n->npf + n->A;
  pack_conv_w(n->blob + off, 1, F, n->nvf, pk + off); off += (size_t)F * n->nvf + 5 * (size_t)n->nvf;
  return pk;
}

/* Network.forward (src/networks/flux.jl:127-132) on ONE sample given as Flux-layout planes
 * x[W][H][C] (column-major, W fastest).  Outputs the raw softmax p[A] and tanh value. */
static void net_forward_one(const azr_net* n, const float* xin, float* p, float* v) {
  int P = n->W * n->H, F = n->F, C = n->C, A = n->A;
  const float* w = n->pk;
  float* x0 = malloc(sizeof(float) * (size_t)P * (C + 3 * F + n->npf + n->nvf + F));
  float* a = x0 + (size_t)P * C;
  float* t = a + (size_t)P * F;
  float* b2 = t + (size_t)P * F;
  float* hp = b2 + (size_t)P * F;
  float* hv = hp + (size_t)P * n->npf;
  float* vh = hv + (size_t)P * n->nvf;
  for (int pz = 0; pz < P; ++pz) for (int c = 0; c < C; ++c) x0[(size_t)pz * C + c] = xin[pz + (size_t)P * c];
  /* stem, resnet.jl:75-77 */
  conv_bn(n, x0, C, F, 3, w, w + 9 * C * F, w + 9 * C * F + F, 0, 1, 2, a);
  w += 9 * (size_t)C * F + 5 * (size_t)F;
  /* tower, resnet.jl:53-63,78 */
  for (int blk = 0; blk < n->nblocks; ++blk) {
    size_t cw = 9 * (size_t)F * F;
    conv_bn(n, a, F, F, 3, w, w + cw, w + cw + F, 0, 1, 1, t);
    w += cw + 5 * (size_t)F;
    conv_bn(n, t, F, F, 3, w, w + cw, w + cw + F, a, 1, 1, b2);
    w += cw + 5 * (size_t)F;
    memcpy(a, b2, sizeof(float) * (size_t)P * F);
  }
  /* policy head, resnet.jl:79-84 */
  conv_bn(n, a, F, n->npf, 1, w, w + (size_t)F * n->npf, w + (size_t)F * n->npf + n->npf, 0, 1, 1, hp);
  w += (size_t)F * n->npf + 5 * (size_t)n->npf;
  {
    float logits[128];            /* the network restatement takes any geometry up to 128 actions (9x9 planes: 82) */
    if (A > 128) { fprintf(stderr, "azref: too many actions\n"); abort(); }
    int K = P * n->npf;
    for (int o = 0; o < A; ++o) {
      float acc = 0.0f;
      for (int pz = 0; pz < P; ++pz) for (int f = 0; f < n->npf; ++f)
        acc = __builtin_fmaf(hp[(size_t)pz * n->npf + f], w[o + (size_t)A * (pz + (size_t)P * f)], acc);
      logits[o] = acc + w[(size_t)A * K + o];
    }
    w += (size_t)A * K + A;
    /* softmax (NNlib, from memory): exp(x - max) / sum, accumulated in action order */
    float m = logits[0];
    for (int o = 1; o < A; ++o) m = logits[o] > m ? logits[o] : m;
    float s = 0.0f;
    for (int o = 0; o < A; ++o) { p[o] = rn_expf(logits[o] - m); s += p[o]; }
    for (int o = 0; o < A; ++o) p[o] = p[o] / s;
  }
  /* value head, resnet.jl:85-90 */
  conv_bn(n, a, F, n->nvf, 1, w, w + (size_t)F * n->nvf, w + (size_t)F * n->nvf + n->nvf, 0, 1, 1, hv);
  w += (size_t)F * n->nvf + 5 * (size_t)n->nvf;
  {
    int K = P * n->nvf;
    for (int o = 0; o < F; ++o) vh[o] = 0.0f;
    for (int pz = 0; pz < P; ++pz) for (int f = 0; f < n->nvf; ++f) {
      float hvv = hv[(size_t)pz * n->nvf + f];
      const float* wr = w + (size_t)F * (pz + (size_t)P * f);
      for (int o = 0; o < F; ++o) vh[o] = __builtin_fmaf(hvv, wr[o], vh[o]);
    }
    for (int o = 0; o < F; ++o) {
      float acc = vh[o] + w[(size_t)F * K + o];
      vh[o] = acc > 0.0f ? acc : 0.0f;
    }
    w += (size_t)F * K + F;
    float acc = 0.0f;
    for (int k = 0; k < F; ++k) acc = __builtin_fmaf(vh[k], w[k], acc);
    acc = acc + w[F];
    *v = rn_tanhf(acc);
  }
  free(x0);
}

/* Network.forward_normalized (src/networks/network.jl:264-271), one column */
static void forward_normalized_one(const azr_net* n, const float* x, const float* amask, float* p, float* v, float* pinv) {
  net_forward_one(n, x, p, v);
  float sp = 0.0f;
  for (int a = 0; a < n->A; ++a) { p[a] = p[a] * amask[a]; sp += p[a]; }
  for (int a = 0; a < n->A; ++a) p[a] = p[a] / (sp + 1.1920929e-7f); /* eps(Float32) */
  *pinv = 1.0f - sp;
}
/* batch form: X is W*H*C*N (WHCN), Amask is A*N; outputs P (A*N), V (N), Pinv (N) */
void azr_net_forward_normalized(int W, int H, int C, int A, int nblocks, int F, int npf, int nvf,
                                const float* blob, const float* X, const float* Amask, int N,
                                float* P, float* V, float* Pinv) {
  azr_net n = {W, H, C, A, nblocks, F, npf, nvf, blob, 0};
  n.pk = net_pack(&n);
  size_t xs = (size_t)W * H * C;
  for (int i = 0; i < N; ++i)
    forward_normalized_one(&n, X + xs * i, Amask + (size_t)A * i, P + (size_t)A * i, V + i, Pinv + i);
  free(n.pk);
}

/* ================================= MCTS ================================= */
/* src/mcts.jl:78-89 */
typedef struct {
  azr_state key;
  uint32_t hash32;              /* low word of the key's hash (re-bucketing without re-hashing 44 bytes) */
  int n;                        /* #available actions */
  float P[AZR_AMAX];            /* ActionStats.P :: Float32 */
  double Wt[AZR_AMAX];          /* ActionStats.W :: Float64 */
  int64_t N[AZR_AMAX];          /* ActionStats.N :: Int     */
  float Vest;                   /* StateInfo.Vest :: Float32 */
} azr_node;

#define AZR_ORACLE_UNIFORM 0   /* MCTS.RandomOracle, src/mcts.jl:62-72 */
#define AZR_ORACLE_HASH 1      /* synthetic: priors/value derived from the packed key */
#define AZR_ORACLE_NET 2       /* Network.evaluate, src/networks/network.jl:287-298 */
#define AZR_ORACLE_ROLLOUT 3   /* MCTS.RolloutOracle, src/mcts.jl:35-60 (gamma = 1, src/benchmark.jl:141-143) */
#define AZR_ORACLE_EXTERNAL 4  /* REPLAY MODE (SURVEY.md §7 hard part 3): oracle(state) answers come from a table the CALLER fills --
                                * e.g. with the device network's P / V -- so the reference's tree (this file) runs on another
                                * evaluator's numbers.  A state the table does not hold yet suspends the simulation (no side effect:
                                * run_simulation! touches the tree only after the oracle has answered, src/mcts.jl:205-222). */

/* The caller-filled evaluation table of replay mode: state key -> (P by FULL action index, V).  An evaluation is a pure function of
 * the state (Network.evaluate_batch evaluates every query on its own, src/networks/network.jl:308-315; test-mode BatchNorm), so one
 * table serves every worker.  Read by the workers in parallel, written only between their rounds. */
typedef struct { uint64_t k0, k1; float P[AZR_AMAX]; float V; int32_t st; int32_t pad; } azr_eval_ent;   /* st: 0 empty, 1 asked for, 2 answered */
typedef struct { azr_eval_ent* tab; size_t cap, count; int64_t asked, answered, wipes; } azr_evals;
azr_evals* azr_evals_new(int log2cap) {
  azr_evals* t = calloc(1, sizeof *t);
  t->cap = (size_t)1 << log2cap;
  const size_t bytes = t->cap * sizeof(azr_eval_ent);
  /* gigabytes touched at random: ask for transparent huge pages where the kernel gives them (fewer faults, fewer TLB misses) */
  void* p = mmap(0, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (p == MAP_FAILED) { fprintf(stderr, "azref: cannot map %zu bytes for the evaluation table\n", bytes); abort(); }
#ifdef MADV_HUGEPAGE
  (void)madvise(p, bytes, MADV_HUGEPAGE);
#endif
  t->tab = p;
  return t;
}
void azr_evals_free(azr_evals* t) { if (t) { munmap(t->tab, t->cap * sizeof(azr_eval_ent)); free(t); } }
void azr_evals_counters(const azr_evals* t, int64_t* out) { out[0] = t->asked; out[1] = t->answered; out[2] = t->wipes; out[3] = (int64_t)t->count; }
static size_t evals_home(const azr_evals* t, const uint64_t* key) { return (size_t)rn_mix64(key[0] ^ rn_mix64(key[1] + 0x9e3779b97f4a7c15ULL)) & (t->cap - 1); }
static azr_eval_ent* evals_find(const azr_evals* t, const uint64_t* key, int insert) {
  size_t j = evals_home(t, key);
  while (t->tab[j].st) {
    if (t->tab[j].k0 == key[0] && t->tab[j].k1 == key[1]) return &t->tab[j];
    j = (j + 1) & (t->cap - 1);
  }
  if (!insert) return 0;
  t->tab[j].k0 = key[0]; t->tab[j].k1 = key[1];
  return &t->tab[j];
}

typedef struct {
  int game;
  /* Dict{State,StateInfo} (src/mcts.jl:126): open addressing over node INDICES (bucket = 1 + index into `pool`, 0 = empty), exact
   * key compare; the nodes themselves sit in creation order in `pool`, so what a tree holds in memory is its nodes, not a
   * half-empty table of them (a BASELINE phase replays 4096 - 8192 trees side by side) */
  uint32_t* tab;
  azr_node* pool;
  size_t cap, count, pool_cap;
  int oracle_kind;
  azr_net net;
  /* src/mcts.jl:129-137 */
  double gamma, cpuct, noise_eps, noise_alpha, prior_temperature;
  int64_t total_simulations, total_nodes_traversed;
  int64_t oracle_calls;
  /* RNG context of the running simulation (rollout oracle): seed, game id, move, simulation index */
  uint64_t rng_seed; uint32_t rng_game, rng_move, rng_sim;
  /* replay mode (AZR_ORACLE_EXTERNAL): the caller's table, and the state the running simulation is suspended on */
  const azr_evals* ext; int ext_miss; uint64_t ext_key[2];
} azr_mcts;

/* (where a state sits in the table is nobody's business but the table's: eight bytes at a time -- the byte-wise FNV this replaces
 * was a tenth of a replayed simulation, two lookups per tree level) */
static uint64_t state_hash(const azr_state* s) {
  uint64_t w[(sizeof(azr_state) + 7) / 8] = {0};
  memcpy(w, s, sizeof(azr_state));
  uint64_t h = 0xcbf29ce484222325ULL;
  for (size_t i = 0; i < sizeof w / sizeof w[0]; ++i) { h = (h ^ w[i]) * 0x9e3779b97f4a7c15ULL; h ^= h >> 32; }
  return rn_mix64(h);
}
static azr_node* tree_find(azr_mcts* e, const azr_state* s, int insert) {
  if (insert && (e->count + 1) * 2 > e->cap) {
    size_t ncap = e->cap ? e->cap * 2 : 1024;
    uint32_t* nt = calloc(ncap, sizeof(uint32_t));
    for (size_t i = 0; i < e->count; ++i) {
      size_t j = (size_t)e->pool[i].hash32 & (ncap - 1);
      while (nt[j]) j = (j + 1) & (ncap - 1);
      nt[j] = (uint32_t)i + 1;
    }
    free(e->tab); e->tab = nt; e->cap = ncap;
  }
  if (!e->cap) return 0;
  const uint32_t h32 = (uint32_t)state_hash(s);
  size_t j = (size_t)h32 & (e->cap - 1);
  while (e->tab[j]) {
    azr_node* nd = &e->pool[e->tab[j] - 1];
    if (nd->hash32 == h32 && memcmp(&nd->key, s, sizeof(azr_state)) == 0) return nd;
    j = (j + 1) & (e->cap - 1);
  }
  if (!insert) return 0;
  if (e->count == e->pool_cap) {                                    /* (pointers into the pool do not survive an insertion: callers re-find) */
    e->pool_cap = e->pool_cap ? e->pool_cap * 2 : 512;
    e->pool = realloc(e->pool, e->pool_cap * sizeof(azr_node));
  }
  azr_node* nd = &e->pool[e->count];
  memset(nd, 0, sizeof *nd);
  nd->key = *s; nd->hash32 = h32;
  e->tab[j] = (uint32_t)(++e->count);
  return nd;
}

azr_mcts* azr_mcts_new(int game, int oracle_kind, double gamma, double cpuct, double noise_eps,
                       double noise_alpha, double prior_temperature) {
  azr_mcts* e = calloc(1, sizeof *e);
  e->game = game; e->oracle_kind = oracle_kind;
  e->gamma = gamma; e->cpuct = cpuct; e->noise_eps = noise_eps; e->noise_alpha = noise_alpha;
  e->prior_temperature = prior_temperature;
  return e;
}
void azr_mcts_set_net(azr_mcts* e, int nblocks, int F, int npf, int nvf, const float* blob) {
  int W, H, C; azr_state_dims(e->game, &W, &H, &C);
  azr_net n = {W, H, C, azr_num_actions_(e->game), nblocks, F, npf, nvf, blob, 0};
  free(e->net.pk);
  n.pk = net_pack(&n);
  e->net = n;
}
/* MCTS.reset! (src/mcts.jl:278-281): empties the tree, keeps the counters */
void azr_mcts_reset(azr_mcts* e) { if (e->tab) memset(e->tab, 0, e->cap * sizeof(uint32_t)); e->count = 0; }
void azr_mcts_free(azr_mcts* e) { free(e->tab); free(e->pool); free(e->net.pk); free(e); }
int64_t azr_mcts_num_nodes(const azr_mcts* e) { return (int64_t)e->count; }
int64_t azr_mcts_total_simulations(const azr_mcts* e) { return e->total_simulations; }
int64_t azr_mcts_total_nodes_traversed(const azr_mcts* e) { return e->total_nodes_traversed; }
int64_t azr_mcts_oracle_calls(const azr_mcts* e) { return e->oracle_calls; }

/* Synthetic oracle: exact small-integer arithmetic so CPU and GPU agree bit for bit.
 * raw_a = 1 + 16 low bits of mix(key, a) for each AVAILABLE action; P = raw / sum(raw)
 * (fp32, sum in action order); V = (h16 - 32768) / 65536. */
void azr_hash_oracle(int game, const azr_state* st, const uint8_t* mask, float* Pfull, float* V) {
  uint64_t key[2]; azr_pack_key(game, st, key);
  uint64_t h = rn_hash_key(key[0], key[1]);
  int A = azr_num_actions_(game);
  float s = 0.0f;
  for (int a = 0; a < A; ++a) {
    float raw = mask[a] ? (float)(1 + (int)(rn_mix64(h + (uint64_t)(a + 1)) & 0xffff)) : 0.0f;
    Pfull[a] = raw; s += raw;
  }
  for (int a = 0; a < A; ++a) Pfull[a] = Pfull[a] / s;
  *V = (float)((int)(rn_mix64(h + 99) & 0xffff) - 32768) / 65536.0f;
}

/* rollout! (src/mcts.jl:41-50); rand(available_actions) = action floor(u * n) of the RNG contract */
static double rollout(azr_env* g, double gamma, rn_stream* r) {
  int acts[AZR_AMAX]; int n = available(g, acts);
  int k = (int)(rn_u64(r) * (double)n);
  if (k >= n) k = n - 1;
  azr_play(g, acts[k]);
  double wr = azr_white_reward(g);
  if (azr_terminated(g)) return wr;
  return wr + gamma * rollout(g, gamma, r);
}

/* oracle(state) -> (P over available actions, V); src/mcts.jl:6-17 */
static void call_oracle(azr_mcts* e, const azr_state* st, float* P, float* V) {
  azr_env g; azr_init_state(&g, e->game, st);       /* GI.init(gspec, state) */
  int acts[AZR_AMAX]; int n = available(&g, acts);
  e->oracle_calls++;
  if (e->oracle_kind == AZR_ORACLE_UNIFORM) {
    for (int i = 0; i < n; ++i) P[i] = (float)(1.0 / (double)n);   /* ones(n) ./ n -> Float32(p) */
    *V = 0.0f;
  } else if (e->oracle_kind == AZR_ORACLE_ROLLOUT) {
    /* (r::RolloutOracle)(state), src/mcts.jl:52-60 */
    int wp = azr_white_playing(&g);
    for (int i = 0; i < n; ++i) P[i] = (float)(1.0 / (double)n);
    rn_stream r = rn_open(e->rng_seed, e->rng_game, e->rng_move, RN_ROLLOUT);
    r.draw = e->rng_sim * 1024u;
    double wr = rollout(&g, 1.0, &r);
    *V = (float)(wp ? wr : -wr);
  } else if (e->oracle_kind == AZR_ORACLE_HASH) {
    float pf[AZR_AMAX];
    azr_hash_oracle(e->game, st, g.amask, pf, V);
    for (int i = 0; i < n; ++i) P[i] = pf[acts[i]];
  } else if (e->oracle_kind == AZR_ORACLE_EXTERNAL) {
    /* the answer is the caller's: P over all actions, compressed to the available ones like evaluate_batch's P[A[:,i],i] (network.jl:314) */
    uint64_t key[2]; azr_pack_key(e->game, st, key);
    const azr_eval_ent* ent = e->ext ? evals_find(e->ext, key, 0) : 0;
    if (!ent || ent->st != 2) {                      /* not answered yet: suspend (the caller evaluates, then the simulation is run again) */
      e->ext_miss = 1; e->ext_key[0] = key[0]; e->ext_key[1] = key[1];
      e->oracle_calls--;
      for (int i = 0; i < n; ++i) P[i] = 0.f;
      *V = 0.f;
      return;
    }
    for (int i = 0; i < n; ++i) P[i] = ent->P[acts[i]];
    *V = ent->V;
  } else {
    /* Network.evaluate (src/networks/network.jl:287-298) */
    int A = e->net.A;
    float x[AZR_CELLS * 5], am[AZR_AMAX], pf[AZR_AMAX], pinv;
    azr_vectorize_state(e->game, st, x);
    for (int a = 0; a < A; ++a) am[a] = g.amask[a] ? 1.0f : 0.0f;
    forward_normalized_one(&e->net, x, am, pf, V, &pinv);
    for (int i = 0; i < n; ++i) P[i] = pf[acts[i]];
  }
}

/* Util.apply_temperature on a Float64 vector (src/util.jl:98-110) */
static void apply_temperature(const double* pi, int n, double tau, double* res) {
  if (tau == 1.0) { for (int i = 0; i < n; ++i) res[i] = pi[i]; return; }
  if (tau == 0.0) {
    int am = 0;
    for (int i = 1; i < n; ++i) if (pi[i] > pi[am]) am = i;       /* first maximum */
    for (int i = 0; i < n; ++i) res[i] = 0.0;
    res[am] = 1.0;
    return;
  }
  double inv = 1.0 / tau, s = 0.0;
  for (int i = 0; i < n; ++i) { res[i] = rn_pow(pi[i], inv); s += res[i]; }
  for (int i = 0; i < n; ++i) res[i] = res[i] / s;
}

/* init_state_info + state_info (src/mcts.jl:157-174) */
static azr_node* state_info(azr_mcts* e, const azr_state* st, int* new_node) {
  azr_node* nd = tree_find(e, st, 0);
  if (nd) { *new_node = 0; return nd; }
  float P[AZR_AMAX], V;
  call_oracle(e, st, P, &V);
  if (e->ext_miss) { *new_node = 0; return 0; }     /* replay mode: suspended before anything was stored */
  nd = tree_find(e, st, 1);
  azr_env g; azr_init_state(&g, e->game, st);
  int acts[AZR_AMAX]; int n = available(&g, acts);
  nd->n = n;
  double pd[AZR_AMAX], pt[AZR_AMAX];
  for (int i = 0; i < n; ++i) pd[i] = (double)P[i];
  if (e->prior_temperature == 1.0) for (int i = 0; i < n; ++i) pt[i] = pd[i];
  else apply_temperature(pd, n, e->prior_temperature, pt);
  for (int i = 0; i < n; ++i) { nd->P[i] = (float)pt[i]; nd->Wt[i] = 0.0; nd->N[i] = 0; }
  nd->Vest = V;
  *new_node = 1;
  return nd;
}

/* uct_scores + argmax (src/mcts.jl:180-188,211): Float64, left to right, first maximum */
static int select_action(const azr_node* nd, double cpuct, double eps, const double* eta) {
  int64_t ntot = 0;
  for (int i = 0; i < nd->n; ++i) ntot += nd->N[i];
  double sqrtNtot = sqrt((double)ntot);
  int best = 0; double bests = 0.0;
  for (int i = 0; i < nd->n; ++i) {
    int64_t N = nd->N[i];
    double Q = nd->Wt[i] / (double)(N > 1 ? N : 1);
    double P = (eps == 0.0) ? (double)nd->P[i] : (1.0 - eps) * (double)nd->P[i] + eps * eta[i];
    double sc = Q + cpuct * P * sqrtNtot / (double)(N + 1);
    if (i == 0 || sc > bests) { best = i; bests = sc; }
  }
  return best;
}

/* run_simulation! (src/mcts.jl:199-226), recursive as in the reference */
static double run_simulation(azr_mcts* e, azr_env* game, const double* eta, int root) {
  if (azr_terminated(game)) return 0.;
  azr_state st; azr_current_state(game, &st);
  int acts[AZR_AMAX]; available(game, acts);
  int new_node;
  azr_node* info = state_info(e, &st, &new_node);
  if (!info) return 0.;                             /* replay mode: suspended (e->ext_miss) */
  if (new_node) return (double)info->Vest;
  const size_t info_idx = (size_t)(info - e->pool);   /* (the pool may move while the recursion inserts; a node's index does not change) */
  double eps = root ? e->noise_eps : 0.;
  int aid = select_action(info, e->cpuct, eps, eta);
  int action = acts[aid];
  int wp = azr_white_playing(game);
  azr_play(game, action);
  double wr = azr_white_reward(game);
  double r = wp ? wr : -wr;
  int pswitch = wp != azr_white_playing(game);
  double qnext = run_simulation(e, game, eta, 0);
  if (e->ext_miss) return 0.;                       /* replay mode: unwind without update_state_info! */
  qnext = pswitch ? -qnext : qnext;
  double q = r + e->gamma * qnext;
  /* update_state_info! (:190-194) */
  info = &e->pool[info_idx];
  info->Wt[aid] += q; info->N[aid] += 1;
  e->total_nodes_traversed += 1;
  return q;
}

/* explore! (src/mcts.jl:239-245) in two parts so that replay mode can suspend it between simulations.
 * explore_begin: the noise of this call -- caller-provided (length = #available actions) or drawn from the RNG contract keyed by
 * (seed, game id, move) -- and the RNG context.  explore_run: simulations `from` .. nsims-1; returns the index of the first simulation
 * that did NOT complete (nsims when all did; less only in replay mode, with e->ext_key the state to evaluate). */
static void explore_begin(azr_mcts* e, const azr_env* game, const double* eta_in, uint64_t seed, uint32_t game_id, uint32_t move, double* eta) {
  int acts[AZR_AMAX]; int n = available(game, acts);
  if (eta_in) memcpy(eta, eta_in, sizeof(double) * (size_t)n);
  else { rn_stream r = rn_open(seed, game_id, move, RN_NOISE); rn_dirichlet(&r, n, e->noise_alpha, eta); }
  e->rng_seed = seed; e->rng_game = game_id; e->rng_move = move;
}
static int explore_run(azr_mcts* e, const azr_env* game, int from, int nsims, const double* eta) {
  for (int i = from; i < nsims; ++i) {
    e->rng_sim = (uint32_t)i;
    e->ext_miss = 0;
    azr_env clone = *game;                       /* GI.clone */
    run_simulation(e, &clone, eta, 1);
    if (e->ext_miss) return i;                   /* nothing was counted or stored: simulation i runs again once the state is answered */
    e->total_simulations += 1;
  }
  return nsims;
}
void azr_mcts_explore(azr_mcts* e, const azr_env* game, int nsims, const double* eta_in, uint64_t seed,
                      uint32_t game_id, uint32_t move) {
  double eta[AZR_AMAX];
  explore_begin(e, game, eta_in, seed, game_id, move, eta);
  int done = explore_run(e, game, 0, nsims, eta);
  if (done != nsims) { fprintf(stderr, "azref: azr_mcts_explore cannot suspend (replay mode goes through azr_sim_step)\n"); abort(); }
}
/* Convenience for tests: explore from a state, returning root statistics by rank. */
int azr_mcts_root_stats(azr_mcts* e, const azr_state* st, int64_t* N, double* W, float* P, float* Vest) {
  azr_node* nd = tree_find(e, st, 0);
  if (!nd) return -1;
  for (int i = 0; i < nd->n; ++i) { N[i] = nd->N[i]; W[i] = nd->Wt[i]; P[i] = nd->P[i]; }
  *Vest = nd->Vest;
  return nd->n;
}
/* policy (src/mcts.jl:255-271) */
int azr_mcts_policy(azr_mcts* e, const azr_env* game, int* actions, double* pi) {
  azr_state st; azr_current_state(game, &st);
  azr_node* nd = tree_find(e, &st, 0);
  if (!nd) return -1;
  int n = available(game, actions);
  int64_t ntot = 0;
  for (int i = 0; i < n; ++i) ntot += nd->N[i];
  double s = 0.0;
  for (int i = 0; i < n; ++i) { pi[i] = (double)nd->N[i] / (double)ntot; s += pi[i]; }
  for (int i = 0; i < n; ++i) pi[i] = pi[i] / s;
  return n;
}

/* ============================ schedules / sampling ======================= */
/* PLSchedule getindex (src/schedule.jl:64-80), Float64 schedule */
double azr_plschedule(const int* xs, const double* ys, int len, int i) {
  int ptidx = -1;
  for (int k = 0; k < len; ++k) if (xs[k] <= i) ptidx = k;
  if (ptidx < 0) return ys[0];
  if (ptidx == len - 1) return ys[len - 1];
  double x0 = xs[ptidx], y0 = ys[ptidx], x1 = xs[ptidx + 1], y1 = ys[ptidx + 1];
  return y0 + (y1 - y0) / (x1 - x0) * ((double)i - x0);
}
/* Integer PLSchedule (ceil), the known-answer of src/schedule.jl:82-87 */
int azr_plschedule_int(const int* xs, const int* ys, int len, int i) {
  int ptidx = -1;
  for (int k = 0; k < len; ++k) if (xs[k] <= i) ptidx = k;
  if (ptidx < 0) return ys[0];
  if (ptidx == len - 1) return ys[len - 1];
  double x0 = xs[ptidx], y0 = ys[ptidx], x1 = xs[ptidx + 1], y1 = ys[ptidx + 1];
  return (int)ceil(y0 + (y1 - y0) / (x1 - x0) * ((double)i - x0));
}
/* fix_probvec + rand_categorical (src/util.jl:68-90) with the uniform given */
int azr_rand_categorical(const double* pi, int n, float u) {
  float p[AZR_AMAX];
  float s = 0.0f;
  for (int i = 0; i < n; ++i) { p[i] = (float)pi[i]; s += p[i]; }
  /* !(s ≈ 1): isapprox with rtol = sqrt(eps(Float32)) */
  float tol = 0.00034526698f * (fabsf(s) > 1.0f ? fabsf(s) : 1.0f);
  if (!(fabsf(s - 1.0f) <= tol)) {
    if (s == 0.0f) for (int i = 0; i < n; ++i) p[i] = 1.0f / (float)n;
    else for (int i = 0; i < n; ++i) p[i] = p[i] / s;
  }
  return rn_categorical(p, n, u);
}
void azr_apply_temperature(const double* pi, int n, double tau, double* res) { apply_temperature(pi, n, tau, res); }

/* ============================ play_game / simulate ======================= */
typedef struct {
  /* MctsParams (src/params.jl:49-57) */
  double gamma, cpuct, noise_eps, noise_alpha, prior_temperature;
  int num_iters_per_turn;
  int temp_len; int temp_xs[8]; double temp_ys[8];   /* PLSchedule; len 1 == ConstSchedule */
  /* SimParams (src/params.jl:92-101) */
  int num_games, num_workers, reset_every;
  uint64_t seed;
  int game, oracle_kind;
  int nblocks, F, npf, nvf; const float* blob;
  int first_game_id;   /* azr_simulate: global id of the first game (a shard of simulate_distributed, simulations.jl:268-278) */
  double flip_probability;   /* azr_simulate: play_game's keyword (play.jl:297, simulations.jl:227); azr_arena takes its own argument */
} azr_sim_params;

/* One trace record per move: state BEFORE the move (packed key), visit counts by action
 * (full width, 0 for unavailable), the action played, white reward after the move. */
typedef struct {
  uint64_t key[2];
  int32_t N[AZR_AMAX + 1];
  int32_t action;
  float reward;
} azr_move_rec;
typedef struct {
  int32_t game_id, slot, num_moves, first_move;   /* index into the move record array */
  int64_t nodes;                                   /* length(env.tree) at game end */
  int64_t total_simulations, total_nodes_traversed; /* cumulative per worker (mcts.jl:136-137) */
  uint64_t final_key[2];
} azr_game_rec;

typedef struct {
  azr_mcts* mcts;
  azr_env game;
  int game_id, nmoves, first_move, worker_sim_id, active;
  /* the turn in progress (stepping form): 0 = not started, 1 = thinking (explore! at simulation sim_i), 2 = move played */
  int turn, sim_i, pre_n, pre_acts[AZR_AMAX];
  double eta[AZR_AMAX];
} azr_slot;

/* simulate (src/simulations.jl:207-244) with num_workers lock-step workers: every round
 * each active worker plays ONE move of its game (think -> sample -> play!, play.jl:298-315);
 * game ids are handed out in increasing order, ties between workers finishing in the same
 * round resolved by worker index (the reference's assignment is a race, util.jl:181-188).
 *
 * Stepping form.  azr_sim_new / azr_sim_step / azr_sim_free hold that loop's state in an object so that, in replay mode
 * (AZR_ORACLE_EXTERNAL), a round can stop where workers wait for evaluations: azr_sim_step runs every worker until it has played
 * its move or is suspended on a state the evaluation table does not hold, and returns the (distinct) states to evaluate;
 * the caller answers them (azr_sim_feed) and steps again.  0 = the phase is over.  With the built-in oracles nothing ever
 * suspends and azr_simulate is one azr_sim_step call: the loop below IS simulate for every oracle kind. */
int azr_num_symmetries(int game);
static void apply_symmetry(azr_env* g, int k);
typedef struct {
  azr_sim_params p;
  int G, maxlen;
  azr_slot* slots;
  azr_move_rec* stage;          /* move records of one game must be contiguous: per-slot staging area first */
  int64_t nm;
  int next_game, finished;
  azr_game_rec* games; azr_move_rec* moves; int64_t moves_cap;   /* the caller's */
  azr_evals* ext; int own_ext;
  int64_t rounds, steps;
  uint8_t* waiting;             /* [G] the worker is suspended on an evaluation */
  int32_t* next_of;             /* azr_sim_set_assignment: index of the next game of the same worker, -1 = none; NULL = ids in finishing order */
  double tm[4];                 /* azr_sim_run, seconds on the calling thread: its share of the workers' turns, waiting for the other threads, serial part, inside the evaluator */
} azr_sim;

azr_sim* azr_sim_new(const azr_sim_params* p, azr_game_rec* games, azr_move_rec* moves, int64_t moves_cap, azr_evals* evals) {
  azr_sim* h = calloc(1, sizeof *h);
  h->p = *p; h->games = games; h->moves = moves; h->moves_cap = moves_cap;
  int G = h->G = p->num_workers < p->num_games ? p->num_workers : p->num_games;
  if (p->flip_probability != 0. && azr_num_symmetries(p->game) == 0) { fprintf(stderr, "azref: no symmetries were declared for this game\n"); abort(); }
  if (p->oracle_kind == AZR_ORACLE_EXTERNAL) {
    h->ext = evals ? evals : azr_evals_new(22); h->own_ext = !evals;
    if (h->ext->cap < (size_t)8 * (size_t)(G > 0 ? G : 1)) { fprintf(stderr, "azref: evaluation table too small for %d workers\n", G); abort(); }
  }
  h->slots = calloc((size_t)(G > 0 ? G : 1), sizeof(azr_slot));
  for (int s = 0; s < G; ++s) {
    azr_slot* sl = &h->slots[s];
    sl->mcts = azr_mcts_new(p->game, p->oracle_kind, p->gamma, p->cpuct, p->noise_eps, p->noise_alpha, p->prior_temperature);
    if (p->oracle_kind == AZR_ORACLE_NET) azr_mcts_set_net(sl->mcts, p->nblocks, p->F, p->npf, p->nvf, p->blob);
    sl->mcts->ext = h->ext;
    sl->game_id = p->first_game_id + h->next_game++; sl->active = 1;
    azr_init(&sl->game, p->game);
    sl->first_move = -1;
  }
  h->waiting = calloc((size_t)(G > 0 ? G : 1), 1);
  h->maxlen = 512;
  h->stage = calloc((size_t)(G > 0 ? G : 1) * h->maxlen, sizeof(azr_move_rec));
  return h;
}
void azr_sim_free(azr_sim* h) {
  for (int s = 0; s < h->G; ++s) azr_mcts_free(h->slots[s].mcts);
  if (h->own_ext) azr_evals_free(h->ext);
  free(h->slots); free(h->stage); free(h->waiting); free(h->next_of); free(h);
}
/* Which worker plays which game.  The reference hands the next id to whichever worker asks first, under a lock (util.jl:181-188): the
 * assignment is a race, and with reset_every != 1 (or the cumulative per-worker counters of self_play_measurements) the results depend
 * on it.  By default this file resolves the race in lock step (ids in finishing order, ties by worker index).  A free-running device
 * phase resolves it its own way and reports how (az_game_rec.slot): worker_of[i] = worker of game first_game_id + i replays THAT
 * outcome -- every worker plays its games in increasing id order, as any outcome of the reference's race has it.  Must be called before
 * the first step; the first min(num_workers, num_games) games belong to workers 0, 1, 2 ... (that is how every phase starts).
 * Returns 0, or -1 if the assignment is not one the reference could produce. */
int azr_sim_set_assignment(azr_sim* h, const int32_t* worker_of) {
  const int n = h->p.num_games, G = h->G;
  if (h->rounds || h->steps) return -1;
  for (int i = 0; i < n; ++i) if (worker_of[i] < 0 || worker_of[i] >= G) return -1;
  for (int s = 0; s < G; ++s) if (worker_of[s] != s) return -1;
  free(h->next_of);
  h->next_of = malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  int32_t* last = malloc(sizeof(int32_t) * (size_t)(G > 0 ? G : 1));
  for (int s = 0; s < G; ++s) last[s] = -1;
  for (int i = 0; i < n; ++i) {
    h->next_of[i] = -1;
    const int s = worker_of[i];
    if (last[s] >= 0) h->next_of[last[s]] = i;
    last[s] = i;
  }
  free(last);
  return 0;
}
int64_t azr_sim_num_moves(const azr_sim* h) { return h->nm; }
/* out: move rounds completed, azr_sim_step calls, oracle calls of the live workers (answered evaluations the trees consumed) */
void azr_sim_counters(const azr_sim* h, int64_t* out) {
  int64_t oc = 0;
  for (int s = 0; s < h->G; ++s) oc += h->slots[s].mcts->oracle_calls;
  out[0] = h->rounds; out[1] = h->steps; out[2] = oc;
}

/* One worker's turn (the body of play_game's loop, play.jl:298-315), resumable.  Returns 1 when the worker is suspended on an
 * evaluation (replay mode), 0 when its move has been played. */
static int slot_turn(azr_sim* h, int s) {
  const azr_sim_params* p = &h->p;
  azr_slot* sl = &h->slots[s];
  azr_move_rec* mr = &h->stage[(size_t)s * h->maxlen + sl->nmoves];
  if (sl->turn == 0) {
    memset(mr, 0, sizeof *mr);
    azr_pack_key(p->game, &sl->game.s, mr->key);                    /* trace.states[i]: pushed BEFORE this turn's flip (play.jl:299, 313) */
    sl->pre_n = available(&sl->game, sl->pre_acts);                 /* available actions of that state, in order */
    if (p->flip_probability != 0.) {                                /* play.jl:305-307 */
      int nsym = azr_num_symmetries(p->game);
      rn_stream r = rn_open(p->seed, (uint32_t)sl->game_id, (uint32_t)sl->nmoves, RN_FLIP);
      if (rn_u64(&r) < p->flip_probability) {
        int k = (int)(rn_u64(&r) * (double)nsym);
        if (k >= nsym) k = nsym - 1;
        apply_symmetry(&sl->game, k);
        mr->N[AZR_AMAX] = k + 1;
      }
    }
    /* think (play.jl:196-206) */
    explore_begin(sl->mcts, &sl->game, 0, p->seed, (uint32_t)sl->game_id, (uint32_t)sl->nmoves, sl->eta);
    sl->sim_i = 0; sl->turn = 1;
  }
  if (sl->turn == 1) {
    sl->sim_i = explore_run(sl->mcts, &sl->game, sl->sim_i, p->num_iters_per_turn, sl->eta);
    if (sl->sim_i < p->num_iters_per_turn) return 1;
    int acts[AZR_AMAX]; double pi[AZR_AMAX], pis[AZR_AMAX];
    int n = azr_mcts_policy(sl->mcts, &sl->game, acts, pi);
    /* the trace holds pi_target over the available actions of the state the player SAW; convert_sample and apply_symmetry
     * spread it over the actions mask of trace.states[i], the state before the flip (learning.jl:31-33, memory.jl:115-118):
     * entry i goes to that state's i-th available action.  Without a flip the two lists are the same. */
    { azr_node* nd = tree_find(sl->mcts, &sl->game.s, 0);
      if (n != sl->pre_n) { fprintf(stderr, "azref: a symmetry changed the number of available actions\n"); abort(); }
      for (int i = 0; i < n; ++i) mr->N[sl->pre_acts[i]] = (int32_t)nd->N[i]; }
    /* temperature index = #moves already played (play.jl:309) */
    double tau = azr_plschedule(p->temp_xs, p->temp_ys, p->temp_len, sl->nmoves);
    apply_temperature(pi, n, tau, pis);
    rn_stream r = rn_open(p->seed, (uint32_t)sl->game_id, (uint32_t)sl->nmoves, RN_MOVE);
    int a = acts[azr_rand_categorical(pis, n, rn_u32(&r))];
    azr_play(&sl->game, a);
    mr->action = a; mr->reward = (float)azr_white_reward(&sl->game);
    sl->nmoves++;
    if (sl->nmoves >= h->maxlen) { fprintf(stderr, "azref: game too long\n"); abort(); }
    sl->turn = 2;
  }
  return 0;
}

/* The three parts of a step.  sim_work: workers [s0, s1) run until each has played its move or is suspended (parallel part: the
 * workers are independent -- one tree, one game, one RNG stream each -- so a round's turns may run on separate host threads without
 * changing any result).  sim_collect: the distinct states the suspended workers wait for, each reported once (serial).
 * sim_round_end: end-of-round bookkeeping in worker order (serial). */
static void sim_work(azr_sim* h, int s0, int s1) {
  for (int s = s0; s < s1; ++s) {
    h->waiting[s] = 0;
    if (h->slots[s].active && h->slots[s].turn != 2) h->waiting[s] = (uint8_t)slot_turn(h, s);
  }
}
static int64_t sim_collect(azr_sim* h, uint64_t* keys_out, int64_t keys_cap) {
  const int G = h->G;
  int64_t nk = 0, nwait = 0;
  for (int s = 0; s < G; ++s) nwait += h->waiting[s];
  if (!nwait) return 0;
  azr_evals* t = h->ext;
  if ((t->count + (size_t)nwait) * 10 > t->cap * 6) {              /* full: forget everything (an evaluation can always be asked for again) */
    memset(t->tab, 0, t->cap * sizeof(azr_eval_ent)); t->count = 0; t->wipes++;
  }
  for (int s = 0; s < G; ++s) if (h->waiting[s]) __builtin_prefetch(&t->tab[evals_home(t, h->slots[s].mcts->ext_key)], 1);
  const int32_t stamp = (int32_t)(h->steps & 0x7fffffff);
  for (int s = 0; s < G; ++s) if (h->waiting[s]) {
    azr_eval_ent* ent = evals_find(t, h->slots[s].mcts->ext_key, 1);
    if (ent->st == 2) continue;                                     /* (answered meanwhile: cannot happen within a step, harmless) */
    if (ent->st == 0) { ent->st = 1; t->count++; }
    else if (ent->pad == stamp) continue;                           /* already reported in this step */
    ent->pad = stamp;                                               /* asked for in an earlier step and never answered: report again */
    if (nk >= keys_cap) { fprintf(stderr, "azref: replay needs room for one key per worker\n"); abort(); }
    keys_out[2 * nk] = ent->k0; keys_out[2 * nk + 1] = ent->k1; nk++;
    t->asked++;
  }
  return nk;
}
static void sim_round_end(azr_sim* h) {
  const azr_sim_params* p = &h->p;
  h->rounds++;
  for (int s = 0; s < h->G; ++s) {
    azr_slot* sl = &h->slots[s];
    if (!sl->active) continue;
    sl->turn = 0;
    if (!azr_terminated(&sl->game)) continue;
    azr_game_rec* gr = &h->games[sl->game_id - p->first_game_id];
    gr->game_id = sl->game_id; gr->slot = s; gr->num_moves = sl->nmoves; gr->first_move = (int32_t)h->nm;
    if (h->nm + sl->nmoves > h->moves_cap) { fprintf(stderr, "azref: move buffer too small\n"); abort(); }
    memcpy(h->moves + h->nm, h->stage + (size_t)s * h->maxlen, sizeof(azr_move_rec) * (size_t)sl->nmoves);
    h->nm += sl->nmoves;
    /* measure (training.jl:269-273) happens BEFORE the periodic reset */
    gr->nodes = azr_mcts_num_nodes(sl->mcts);
    gr->total_simulations = sl->mcts->total_simulations;
    gr->total_nodes_traversed = sl->mcts->total_nodes_traversed;
    azr_pack_key(p->game, &sl->game.s, gr->final_key);
    sl->worker_sim_id++;
    if (p->reset_every > 0 && sl->worker_sim_id % p->reset_every == 0) azr_mcts_reset(sl->mcts);
    h->finished++;
    if (h->next_of) {                                               /* a given outcome of the race (azr_sim_set_assignment) */
      const int nx = h->next_of[sl->game_id - p->first_game_id];
      if (nx >= 0) { sl->game_id = p->first_game_id + nx; sl->nmoves = 0; h->next_game++; azr_init(&sl->game, p->game); }
      else sl->active = 0;
    } else if (h->next_game < p->num_games) {
      sl->game_id = p->first_game_id + h->next_game++; sl->nmoves = 0;
      azr_init(&sl->game, p->game);
    } else sl->active = 0;
  }
}

/* Runs the phase until workers wait for evaluations or it is over.  keys_out (2 x keys_cap words; replay mode only): the distinct
 * states to evaluate, each reported once.  Returns their number, 0 when every game has been played. */
int64_t azr_sim_step(azr_sim* h, uint64_t* keys_out, int64_t keys_cap) {
  const int G = h->G;
  h->steps++;
  while (h->finished < h->p.num_games) {
    const int chunk = h->ext ? 8 : 1;              /* replay mode: a turn slice is one simulation or so; otherwise a whole move */
#pragma omp parallel for schedule(dynamic, chunk)
    for (int s = 0; s < G; ++s) sim_work(h, s, s + 1);
    const int64_t nk = sim_collect(h, keys_out, keys_cap);
    if (nk) return nk;
    sim_round_end(h);
  }
  return 0;
}
/* The caller's answers: P [n][A] by FULL action index (0 on unavailable actions, as Network.evaluate_batch's masked
 * forward_normalized gives them, network.jl:264-271), V [n]. */
void azr_sim_feed(azr_sim* h, const uint64_t* keys, const float* P, const float* V, int64_t n) {
  int A = azr_num_actions_(h->p.game);
  for (int64_t i = 0; i < n; ++i) __builtin_prefetch(&h->ext->tab[evals_home(h->ext, keys + 2 * i)], 1);
  for (int64_t i = 0; i < n; ++i) {
    azr_eval_ent* ent = evals_find(h->ext, keys + 2 * i, 0);
    if (!ent) { fprintf(stderr, "azref: azr_sim_feed: a state nobody asked for\n"); abort(); }
    for (int a = 0; a < A; ++a) ent->P[a] = P[(size_t)i * A + a];
    ent->V = V[i];
    if (ent->st != 2) h->ext->answered++;
    ent->st = 2;
  }
}
/* The whole replay in ONE call: azr_sim_step's loop with the evaluator called from inside (`fn`: the address of any function with
 * Network.evaluate_batch's meaning -- keys in, P [n][A] and V [n] out, 0 = ok; the tests pass the device library's
 * az_net_evaluate_keys with its engine as `user`, or a ctypes callback).  Same three parts as azr_sim_step; what differs is only how
 * the host threads are kept: one team for the whole call, waiting for the serial part (and the evaluator) in a spin barrier, because
 * a fork / join per step costs more than the step's work on a large host (12 k - 36 k steps per phase).  fn runs on the calling
 * thread.  Returns 0, or fn's status if it fails. */
typedef int (*azr_eval_fn)(void* user, const uint64_t* keys, int32_t n, float* P, float* V);
static inline void cpu_relax(void) { __builtin_ia32_pause(); }
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
void azr_sim_times(const azr_sim* h, double* out) { for (int i = 0; i < 4; ++i) out[i] = h->tm[i]; }
int azr_sim_run(azr_sim* h, azr_eval_fn fn, void* user, int nthreads, int64_t* evaluated) {
  const int G = h->G, A = azr_num_actions_(h->p.game);
  const int CH = 4, nchunks = (G + CH - 1) / CH;
  uint64_t* keys = malloc(sizeof(uint64_t) * 2 * (size_t)(G > 0 ? G : 1));
  float* P = malloc(sizeof(float) * (size_t)A * (size_t)(G > 0 ? G : 1));
  float* V = malloc(sizeof(float) * (size_t)(G > 0 ? G : 1));
  int next_chunk = 0, arrived = 0, gen = 0, done = h->finished >= h->p.num_games, status = 0;
  int64_t nev = 0;
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
  {
    const int me = omp_get_thread_num(), nt = omp_get_num_threads();
    int mygen = 0;
    while (!__atomic_load_n(&done, __ATOMIC_ACQUIRE)) {
      int c;
      const double t0 = me == 0 ? now_s() : 0.;
      while ((c = __atomic_fetch_add(&next_chunk, 1, __ATOMIC_RELAXED)) < nchunks) sim_work(h, c * CH, (c + 1) * CH < G ? (c + 1) * CH : G);
      if (me != 0) {
        __atomic_fetch_add(&arrived, 1, __ATOMIC_RELEASE);
        while (__atomic_load_n(&gen, __ATOMIC_ACQUIRE) == mygen) cpu_relax();
      } else {
        const double t1 = now_s();
        while (__atomic_load_n(&arrived, __ATOMIC_ACQUIRE) != nt - 1) cpu_relax();
        const double t2 = now_s();
        double te = 0.;
        /* serial part, on the calling thread */
        h->steps++;
        const int64_t nk = sim_collect(h, keys, G);
        if (nk) {
          const double t3 = now_s();
          const int rc = fn(user, keys, (int32_t)nk, P, V);
          te = now_s() - t3;
          if (rc != 0) { status = rc; __atomic_store_n(&done, 1, __ATOMIC_RELEASE); }
          else { azr_sim_feed(h, keys, P, V, nk); nev += nk; }
        } else {
          sim_round_end(h);
          if (h->finished >= h->p.num_games) __atomic_store_n(&done, 1, __ATOMIC_RELEASE);
        }
        h->tm[0] += t1 - t0; h->tm[1] += t2 - t1; h->tm[2] += now_s() - t2 - te; h->tm[3] += te;
        __atomic_store_n(&arrived, 0, __ATOMIC_RELAXED);
        __atomic_store_n(&next_chunk, 0, __ATOMIC_RELAXED);
        __atomic_store_n(&gen, mygen + 1, __ATOMIC_RELEASE);
      }
      mygen++;
    }
  }
  free(keys); free(P); free(V);
  if (evaluated) *evaluated = nev;
  return status;
}

/* worker_of: NULL, or the assignment to replay (azr_sim_set_assignment); -1 if it is not a valid one */
int64_t azr_simulate_assigned(const azr_sim_params* p, azr_game_rec* games, azr_move_rec* moves, int64_t moves_cap, const int32_t* worker_of) {
  if (p->oracle_kind == AZR_ORACLE_EXTERNAL) { fprintf(stderr, "azref: replay mode goes through azr_sim_step\n"); abort(); }
  azr_sim* h = azr_sim_new(p, games, moves, moves_cap, 0);
  if (worker_of && azr_sim_set_assignment(h, worker_of) != 0) { azr_sim_free(h); return -1; }
  int64_t r = azr_sim_step(h, 0, 0);
  if (r != 0) abort();
  int64_t nm = h->nm;
  azr_sim_free(h);
  return nm;
}
int64_t azr_simulate(const azr_sim_params* p, azr_game_rec* games, azr_move_rec* moves, int64_t moves_cap) {
  if (p->oracle_kind == AZR_ORACLE_EXTERNAL) { fprintf(stderr, "azref: replay mode goes through azr_sim_step\n"); abort(); }
  azr_sim* h = azr_sim_new(p, games, moves, moves_cap, 0);
  int64_t r = azr_sim_step(h, 0, 0);
  if (r != 0) abort();
  int64_t nm = h->nm;
  azr_sim_free(h);
  return nm;
}

/* Batch forms of two built-in oracles by state key, for feeding replay mode from this file (CPU tests: replay mode fed with the
 * oracle's own answers must reproduce the direct run).  P [n][A] full width, V [n]. */
void azr_hash_oracle_keys(int game, const uint64_t* keys, int64_t n, float* P, float* V) {
  int A = azr_num_actions_(game);
  for (int64_t i = 0; i < n; ++i) {
    azr_state st; azr_unpack_key(game, keys + 2 * i, &st);
    azr_env g; azr_init_state(&g, game, &st);
    azr_hash_oracle(game, &st, g.amask, P + (size_t)i * A, V + i);
  }
}
/* the hash oracle with azr_sim_run's evaluator signature (user = the game id): the C-function-pointer form of the CPU tests */
int azr_eval_hash(void* user, const uint64_t* keys, int32_t n, float* P, float* V) {
  azr_hash_oracle_keys((int)(intptr_t)user, keys, n, P, V);
  return 0;
}
void azr_net_evaluate_keys(int game, int nblocks, int F, int npf, int nvf, const float* blob, const uint64_t* keys, int64_t n, float* P, float* V) {
  int W, H, C; azr_state_dims(game, &W, &H, &C);
  int A = azr_num_actions_(game);
  azr_net net = {W, H, C, A, nblocks, F, npf, nvf, blob, 0};
  net.pk = net_pack(&net);
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t i = 0; i < n; ++i) {
    azr_state st; azr_unpack_key(game, keys + 2 * i, &st);
    azr_env g; azr_init_state(&g, game, &st);
    float x[AZR_CELLS * 5], am[AZR_AMAX], pinv;
    azr_vectorize_state(game, &st, x);
    for (int a = 0; a < A; ++a) am[a] = g.amask[a] ? 1.0f : 0.0f;
    forward_normalized_one(&net, x, am, P + (size_t)i * A, V + i, &pinv);
  }
  free(net.pk);
}

/* ================================ arena ================================== */
/* GI.symmetries: connect-four has ONE (column mirror, games/connect-four/game.jl:243-257),
 * tic-tac-toe the 7 non-trivial dihedral maps in the order rot, rot2, rot3, flip, flip.rot,
 * flip.rot2, flip.rot3 (games/tictactoe/game.jl:149-168), mancala declares none. */
int azr_num_symmetries(int game) { return game == AZR_C4 ? 1 : game == AZR_TTT ? 7 : 0; }
static void ttt_rot(int* x, int* y) { int nx = *y, ny = 3 - *x + 1; *x = nx; *y = ny; }   /* rot((x,y)) = (y, N-x+1) */
static void ttt_flip(int* x, int* y) { (void)x; *y = 3 - *y + 1; }                        /* flip((x,y)) = (x, N-y+1) */
void azr_symmetry(int game, const azr_state* st, int k, azr_state* out) {
  *out = *st;
  if (game == AZR_C4) {
    /* flipped_board: board[col,row] for col in reverse(1:NUM_COLS) */
    for (int col = 0; col < C4_COLS; ++col) for (int row = 0; row < C4_ROWS; ++row)
      C4(out->cells, col, row) = C4(st->cells, C4_COLS - 1 - col, row);
  } else if (game == AZR_TTT) {
    /* Board(s.board[sym]) with sym[p] = pos_of_xy(f(xy_of_pos(p))), 1-based positions */
    for (int p = 1; p <= 9; ++p) {
      int x = (p - 1) % 3 + 1, y = (p - 1) / 3 + 1;
      int nrot = k < 3 ? k + 1 : k - 3;            /* (f . g)(xy) = f(g(xy)): the rotations act first */
      for (int r = 0; r < nrot; ++r) ttt_rot(&x, &y);
      if (k >= 3) ttt_flip(&x, &y);
      out->cells[p - 1] = st->cells[(y - 1) * 3 + (x - 1)];
    }
  }
}
/* apply_random_symmetry! (src/game.jl:329-336), index chosen by the caller */
static void apply_symmetry(azr_env* g, int k) {
  azr_state st;
  azr_symmetry(g->game, &g->s, k, &st);
  azr_init_state(g, g->game, &st);
}
static int cmp_key(const void* a, const void* b) { return memcmp(a, b, 16); }

/* pit_networks (src/training.jl:130-144): simulate (src/simulations.jl:207-244) over
 * TwoPlayers(MctsPlayer(contender), MctsPlayer(baseline)) (src/play.jl:248-282), lock-step
 * workers as in azr_simulate.  pc holds the contender's MctsParams / network and the SimParams
 * (num_games, num_workers, reset_every, seed for the flips); pb the baseline's MctsParams /
 * network.  Writes traces (key = state BEFORE the turn's flip, N / action in the flipped frame,
 * N[AZR_AMAX] = 1 + symmetry index), rewards from the contender's side and the redundancy
 * (rewards_and_redundancy, simulations.jl:296-311).  Returns the number of move records. */
/* worker_of: NULL (ids in finishing order, ties by worker index), or the outcome of the reference's id race to replay -- worker_of[i] =
 * worker of game first_game_id + i, as a free-running device arena reports it (azr_sim_set_assignment's rules; -1 if it breaks them) */
int64_t azr_arena_assigned(const azr_sim_params* pc, const azr_sim_params* pb, int alternate_colors, double flip_probability,
                           int first_game_id, azr_game_rec* games, azr_move_rec* moves, int64_t moves_cap, double* rewards,
                           double* redundancy, const int32_t* worker_of) {
  int G = pc->num_workers < pc->num_games ? pc->num_workers : pc->num_games;
  int32_t* next_of = 0;
  if (worker_of) {
    const int n = pc->num_games;
    for (int i = 0; i < n; ++i) if (worker_of[i] < 0 || worker_of[i] >= G) return -1;
    for (int s = 0; s < G; ++s) if (worker_of[s] != s) return -1;
    next_of = malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    int32_t* last = malloc(sizeof(int32_t) * (size_t)(G > 0 ? G : 1));
    for (int s = 0; s < G; ++s) last[s] = -1;
    for (int i = 0; i < n; ++i) { next_of[i] = -1; if (last[worker_of[i]] >= 0) next_of[last[worker_of[i]]] = i; last[worker_of[i]] = i; }
    free(last);
  }
  const azr_sim_params* pp[2] = {pc, pb};
  azr_slot* slots = calloc((size_t)G, sizeof(azr_slot));
  azr_mcts** trees[2];
  for (int k = 0; k < 2; ++k) {
    trees[k] = calloc((size_t)G, sizeof(azr_mcts*));
    for (int s = 0; s < G; ++s) {
      const azr_sim_params* p = pp[k];
      trees[k][s] = azr_mcts_new(pc->game, p->oracle_kind, p->gamma, p->cpuct, p->noise_eps, p->noise_alpha, p->prior_temperature);
      if (p->oracle_kind == AZR_ORACLE_NET) azr_mcts_set_net(trees[k][s], p->nblocks, p->F, p->npf, p->nvf, p->blob);
    }
  }
  int64_t nm = 0;
  int next_game = 0, finished = 0, maxlen = 512;
  for (int s = 0; s < G; ++s) { slots[s].game_id = first_game_id + next_game++; slots[s].active = 1; azr_init(&slots[s].game, pc->game); }
  azr_move_rec* stage = calloc((size_t)G * maxlen, sizeof(azr_move_rec));
  int nsym = azr_num_symmetries(pc->game);
  if (flip_probability != 0. && nsym == 0) { fprintf(stderr, "azref: no symmetries were declared for this game\n"); abort(); }
  while (finished < pc->num_games) {
    for (int s = 0; s < G; ++s) {
      azr_slot* sl = &slots[s];
      if (!sl->active) continue;
      azr_move_rec* mr = &stage[(size_t)s * maxlen + sl->nmoves];
      memset(mr, 0, sizeof *mr);
      azr_pack_key(pc->game, &sl->game.s, mr->key);
      if (flip_probability != 0.) {                                   /* play.jl:305-307 */
        rn_stream r = rn_open(pc->seed, (uint32_t)sl->game_id, (uint32_t)sl->nmoves, RN_FLIP);
        if (rn_u64(&r) < flip_probability) {
          int k = (int)(rn_u64(&r) * (double)nsym);
          if (k >= nsym) k = nsym - 1;
          apply_symmetry(&sl->game, k);
          mr->N[AZR_AMAX] = k + 1;
        }
      }
      int colors_flipped = alternate_colors && ((sl->game_id - first_game_id + 1) % 2 == 1);   /* simulations.jl:221-223 */
      int who = (azr_white_playing(&sl->game) != colors_flipped) ? 0 : 1;      /* think(::TwoPlayers), play.jl:255-261 */
      const azr_sim_params* p = pp[who];
      azr_mcts* m = trees[who][s];
      int acts[AZR_AMAX]; double pi[AZR_AMAX], pis[AZR_AMAX];
      int n;
      if (p->num_iters_per_turn > 0) {                                /* MctsPlayer.think, play.jl:196-206 */
        azr_mcts_explore(m, &sl->game, p->num_iters_per_turn, 0, p->seed, (uint32_t)sl->game_id, (uint32_t)sl->nmoves);
        n = azr_mcts_policy(m, &sl->game, acts, pi);
        azr_node* nd = tree_find(m, &sl->game.s, 0);
        for (int i = 0; i < n; ++i) mr->N[acts[i]] = (int32_t)nd->N[i];
      } else {                                                        /* NetworkPlayer.think, play.jl:230-235 */
        float P[AZR_AMAX], V;
        n = available(&sl->game, acts);
        call_oracle(m, &sl->game.s, P, &V);
        for (int i = 0; i < n; ++i) { pi[i] = (double)P[i]; memcpy(&mr->N[acts[i]], &P[i], 4); }
        mr->N[AZR_AMAX] |= 0x100;
      }
      double tau = azr_plschedule(p->temp_xs, p->temp_ys, p->temp_len, sl->nmoves);   /* player_temperature, play.jl:276-282; PlayerWithTemperature */
      apply_temperature(pi, n, tau, pis);
      rn_stream r = rn_open(p->seed, (uint32_t)sl->game_id, (uint32_t)sl->nmoves, RN_MOVE);
      int a = acts[azr_rand_categorical(pis, n, rn_u32(&r))];
      azr_play(&sl->game, a);
      mr->action = a; mr->reward = (float)azr_white_reward(&sl->game);
      sl->nmoves++;
      if (sl->nmoves >= maxlen) { fprintf(stderr, "azref: game too long\n"); abort(); }
    }
    for (int s = 0; s < G; ++s) {
      azr_slot* sl = &slots[s];
      if (!sl->active || !azr_terminated(&sl->game)) continue;
      int gi = sl->game_id - first_game_id;
      azr_game_rec* gr = &games[gi];
      memset(gr, 0, sizeof *gr);
      gr->game_id = sl->game_id; gr->slot = s; gr->num_moves = sl->nmoves; gr->first_move = (int32_t)nm;
      if (nm + sl->nmoves > moves_cap) { fprintf(stderr, "azref: move buffer too small\n"); abort(); }
      memcpy(moves + nm, stage + (size_t)s * maxlen, sizeof(azr_move_rec) * (size_t)sl->nmoves);
      azr_pack_key(pc->game, &sl->game.s, gr->final_key);
      /* total_reward (src/trace.jl:45-47), sign by colors_flipped (simulations.jl:304-307) */
      double wr = 0., gp = 1.;
      for (int i = 0; i < sl->nmoves; ++i) { wr += gp * (double)moves[nm + i].reward; gp *= pc->gamma; }
      int colors_flipped = alternate_colors && ((sl->game_id - first_game_id + 1) % 2 == 1);
      if (rewards) rewards[gi] = colors_flipped ? -wr : wr;
      nm += sl->nmoves;
      sl->worker_sim_id++;
      if (pc->reset_every > 0 && sl->worker_sim_id % pc->reset_every == 0) { azr_mcts_reset(trees[0][s]); azr_mcts_reset(trees[1][s]); }
      finished++;
      if (next_of) {
        const int nx = next_of[gi];
        if (nx >= 0) { sl->game_id = first_game_id + nx; sl->nmoves = 0; next_game++; azr_init(&sl->game, pc->game); }
        else sl->active = 0;
      } else if (next_game < pc->num_games) { sl->game_id = first_game_id + next_game++; sl->nmoves = 0; azr_init(&sl->game, pc->game); }
      else sl->active = 0;
    }
  }
  free(next_of);
  if (redundancy) {                                                   /* compute_redundancy, simulations.jl:296-299 */
    int64_t ns = nm + pc->num_games, j = 0;
    uint64_t* keys = malloc((size_t)ns * 16);
    for (int64_t i = 0; i < nm; ++i) { keys[2 * j] = moves[i].key[0]; keys[2 * j + 1] = moves[i].key[1]; ++j; }
    for (int g = 0; g < pc->num_games; ++g) { keys[2 * j] = games[g].final_key[0]; keys[2 * j + 1] = games[g].final_key[1]; ++j; }
    qsort(keys, (size_t)ns, 16, cmp_key);
    int64_t uniq = ns > 0;
    for (int64_t i = 1; i < ns; ++i) if (memcmp(keys + 2 * i, keys + 2 * (i - 1), 16)) uniq++;
    *redundancy = 1. - (double)uniq / (double)ns;
    free(keys);
  }
  for (int k = 0; k < 2; ++k) { for (int s = 0; s < G; ++s) azr_mcts_free(trees[k][s]); free(trees[k]); }
  free(slots); free(stage);
  return nm;
}

int64_t azr_arena(const azr_sim_params* pc, const azr_sim_params* pb, int alternate_colors, double flip_probability,
                  int first_game_id, azr_game_rec* games, azr_move_rec* moves, int64_t moves_cap, double* rewards,
                  double* redundancy) {
  return azr_arena_assigned(pc, pb, alternate_colors, flip_probability, first_game_id, games, moves, moves_cap, rewards, redundancy, 0);
}

/* push_trace! (src/memory.jl:74-87): discounted side-relative z and t for each position */
void azr_push_trace(int game, const azr_move_rec* moves, int n, double gamma, double* z, double* t) {
  double wr = 0.;
  for (int i = n - 1; i >= 0; --i) {
    wr = gamma * wr + (double)moves[i].reward;
    int wp = !(moves[i].key[0] >> 63);
    (void)game;
    z[i] = wp ? wr : -wr;
    t[i] = (double)(n - i);
  }
}

/* ===================== replay memory + learning status ==================== */
/* TrainingSample (src/memory.jl:20-26); pi by FULL action index (0 where unavailable) */
typedef struct {
  uint64_t key[2];
  double pi[AZR_AMAX];
  double z, t;
  int64_t n;
} azr_sample;

/* push_trace! (src/memory.jl:74-87): one sample per position, pushed from the LAST position to the first;
 * pi = MCTS.policy of the recorded visit counts (src/mcts.jl:255-271).  Returns the number of samples. */
int azr_samples_from_trace(int game, const azr_move_rec* moves, int n, double gamma, azr_sample* out) {
  double wr = 0.;
  int A = azr_num_actions_(game);
  for (int i = n - 1; i >= 0; --i) {
    wr = gamma * wr + (double)moves[i].reward;
    azr_state st; azr_unpack_key(game, moves[i].key, &st);
    azr_env g; azr_init_state(&g, game, &st);
    azr_sample* e = &out[n - 1 - i];
    memset(e, 0, sizeof *e);
    e->key[0] = moves[i].key[0]; e->key[1] = moves[i].key[1];
    double ntot = 0.;
    for (int a = 0; a < A; ++a) if (g.amask[a]) ntot += (double)moves[i].N[a];
    double s = 0.;
    for (int a = 0; a < A; ++a) if (g.amask[a]) { e->pi[a] = (double)moves[i].N[a] / ntot; s += e->pi[a]; }
    for (int a = 0; a < A; ++a) if (g.amask[a]) e->pi[a] = e->pi[a] / s;
    e->z = azr_white_playing(&g) ? wr : -wr;
    e->t = (double)(n - i);
    e->n = 1;
  }
  return n;
}

/* action permutation of symmetry k: pi'[j] = pi[perm[j]] (memory.jl:122-132 with the (state, aperm) pairs of
 * GI.symmetries: connect-four sigma = 7..1, tic-tac-toe the board permutation itself) */
static void symmetry_perm(int game, int k, int* perm) {
  if (game == AZR_C4) { for (int j = 0; j < 7; ++j) perm[j] = 6 - j; return; }
  for (int p = 1; p <= 9; ++p) {
    int x = (p - 1) % 3 + 1, y = (p - 1) / 3 + 1;
    int nrot = k < 3 ? k + 1 : k - 3;
    for (int r = 0; r < nrot; ++r) ttt_rot(&x, &y);
    if (k >= 3) ttt_flip(&x, &y);
    perm[p - 1] = (y - 1) * 3 + (x - 1);
  }
}
/* augment_with_symmetries (memory.jl:134-138): [samples ; (apply_symmetry(s, sym) for s in samples for sym in symmetries)] */
int64_t azr_augment_with_symmetries(int game, const azr_sample* in, int64_t n, azr_sample* out) {
  int nsym = azr_num_symmetries(game), A = azr_num_actions_(game);
  memcpy(out, in, sizeof(azr_sample) * (size_t)n);
  int64_t m = n;
  for (int64_t i = 0; i < n; ++i) for (int k = 0; k < nsym; ++k) {
    azr_state st, sy; azr_unpack_key(game, in[i].key, &st);
    azr_symmetry(game, &st, k, &sy);
    azr_sample* e = &out[m++];
    *e = in[i];
    azr_pack_key(game, &sy, e->key);
    int perm[AZR_AMAX]; symmetry_perm(game, k, perm);
    for (int j = 0; j < A; ++j) e->pi[j] = in[i].pi[perm[j]];
  }
  return m;
}

/* merge_by_state + merge_samples (memory.jl:89-114): samples of one state are averaged in buffer order
 * (mean over a generator = sequential sum / count), n summed.  The reference returns them in Dict order
 * (unspecified); here: ascending (key[0], key[1]). */
typedef struct { uint64_t k0, k1; int64_t idx; } merge_ent;
static int cmp_merge(const void* a, const void* b) {
  const merge_ent *x = a, *y = b;
  if (x->k0 != y->k0) return x->k0 < y->k0 ? -1 : 1;
  if (x->k1 != y->k1) return x->k1 < y->k1 ? -1 : 1;
  return x->idx < y->idx ? -1 : x->idx > y->idx;
}
int64_t azr_merge_by_state(int game, const azr_sample* in, int64_t n, azr_sample* out) {
  int A = azr_num_actions_(game);
  merge_ent* ents = malloc(sizeof(merge_ent) * (size_t)(n > 0 ? n : 1));
  for (int64_t i = 0; i < n; ++i) { ents[i].k0 = in[i].key[0]; ents[i].k1 = in[i].key[1]; ents[i].idx = i; }
  qsort(ents, (size_t)n, sizeof(merge_ent), cmp_merge);
  int64_t m = 0;
  for (int64_t i = 0; i < n;) {
    int64_t j = i;
    azr_sample acc = in[ents[i].idx];
    int64_t cnt = 1;
    for (j = i + 1; j < n && ents[j].k0 == ents[i].k0 && ents[j].k1 == ents[i].k1; ++j) {
      const azr_sample* e = &in[ents[j].idx];
      for (int a = 0; a < A; ++a) acc.pi[a] += e->pi[a];
      acc.z += e->z; acc.t += e->t; acc.n += e->n;
      cnt++;
    }
    for (int a = 0; a < A; ++a) acc.pi[a] = acc.pi[a] / (double)cnt;
    acc.z = acc.z / (double)cnt; acc.t = acc.t / (double)cnt;
    out[m++] = acc;
    i = j;
  }
  free(ents);
  return m;
}

/* convert_samples (src/learning.jl:17-51): policy 0 CONSTANT_WEIGHT, 1 LOG_WEIGHT, 2 LINEAR_WEIGHT (params.jl:177) */
void azr_convert_samples(int game, int policy, const azr_sample* es, int64_t n, float* W, float* X, float* Am, float* P, float* V) {
  int w, h, c; azr_state_dims(game, &w, &h, &c);
  int A = azr_num_actions_(game);
  size_t xs = (size_t)w * h * c;
  for (int64_t i = 0; i < n; ++i) {
    const azr_sample* e = &es[i];
    W[i] = policy == 0 ? 1.0f : policy == 1 ? (float)(rn_log2((double)e->n) + 1.0) : (float)e->n;
    azr_state st; azr_unpack_key(game, e->key, &st);
    azr_env g; azr_init_state(&g, game, &st);
    azr_vectorize_state(game, &st, X + xs * (size_t)i);
    for (int a = 0; a < A; ++a) { Am[(size_t)i * A + a] = g.amask[a] ? 1.0f : 0.0f; P[(size_t)i * A + a] = (float)e->pi[a]; }
    V[i] = (float)e->z;
  }
}

/* Sum of squares of the TRAINABLE parameters (Flux.params: everything but the BatchNorm running statistics),
 * the L2 term of `losses` (learning.jl:68-84 regularises ALL of Network.params) */
static double net_reg_sum(const azr_net* n) {
  size_t P = (size_t)n->W * n->H, F = n->F;
  const float* b = n->blob;
  double s = 0.;
  #define TAKE(cnt) do { for (size_t _i = 0; _i < (size_t)(cnt); ++_i) s += (double)b[_i] * (double)b[_i]; b += (cnt); } while (0)
  #define SKIP(cnt) do { b += (cnt); } while (0)
  TAKE(9 * (size_t)n->C * F); TAKE(F); TAKE(2 * F); SKIP(2 * F);
  for (int l = 0; l < 2 * n->nblocks; ++l) { TAKE(9 * F * F); TAKE(F); TAKE(2 * F); SKIP(2 * F); }
  TAKE(F * n->npf); TAKE(n->npf); TAKE(2 * (size_t)n->npf); SKIP(2 * (size_t)n->npf); TAKE((size_t)n->A * P * n->npf); TAKE(n->A);
  TAKE(F * n->nvf); TAKE(n->nvf); TAKE(2 * (size_t)n->nvf); SKIP(2 * (size_t)n->nvf); TAKE(F * P * n->nvf); TAKE(F); TAKE(F); TAKE(1);
  #undef TAKE
  #undef SKIP
  return s;
}

/* Report.LearningStatus (src/report.jl) of a converted data set: `losses` (learning.jl:67-90) per batch of
 * loss_computation_batch_size samples (partial last batch kept), Hpnet, then mean_learning_status weighted by
 * the batches' total weights (learning.jl:148-181).  Float32 per-sample terms, Float64 accumulators. */
typedef struct { float L, Lp, Lv, Lreg, Linv, Hp, Hpnet, Wmean; } azr_learning_status_t;
void azr_learning_status(int W_, int H_, int C_, int A, int nblocks, int F, int npf, int nvf, const float* blob,
                         const float* W, const float* X, const float* Am, const float* P, const float* V, int64_t n,
                         double l2, double cinv, double renorm, int64_t batch, azr_learning_status_t* out) {
  azr_net net = {W_, H_, C_, A, nblocks, F, npf, nvf, blob, 0};
  net.pk = net_pack(&net);
  size_t xs = (size_t)W_ * H_ * C_;
  const float epsf = 1.1920929e-07f;
  double sw = 0., shp = 0.;
  for (int64_t i = 0; i < n; ++i) {
    sw += (double)W[i];
    for (int a = 0; a < A; ++a) { float p = P[i * A + a]; shp += (double)(p * rn_logf(p + epsf) * W[i]); }
  }
  float Wmean = (float)(sw / (double)n);                    /* mean(W), learning.jl:110 */
  float Hp = (float)(-shp / sw);                            /* entropy_wmean(P, W), :111 */
  float Lreg = l2 == 0. ? 0.f : (float)((double)(float)l2 * net_reg_sum(&net));
  if (batch > n) batch = n;
  double aL = 0., aLp = 0., aLv = 0., aLreg = 0., aLinv = 0., aHn = 0., aw = 0.;
  for (int64_t b0 = 0; b0 < n; b0 += batch) {
    int64_t m = n - b0 < batch ? n - b0 : batch;
    double bw = 0., kl = 0., mse = 0., inv = 0., hn = 0.;
    for (int64_t i = b0; i < b0 + m; ++i) {
      float ph[AZR_AMAX], vh, pinv;
      forward_normalized_one(&net, X + xs * (size_t)i, Am + (size_t)i * A, ph, &vh, &pinv);
      float w = W[i];
      bw += (double)w;
      for (int a = 0; a < A; ++a) {
        kl += (double)(P[i * A + a] * rn_logf(ph[a] + epsf) * w);
        hn += (double)(ph[a] * rn_logf(ph[a] + epsf) * w);
      }
      float d = vh / (float)renorm - V[i] / (float)renorm;
      mse += (double)(d * d * w);
      inv += (double)(pinv * w);
    }
    float Lp = (float)(-kl / bw) - Hp;
    float Lv = (float)(mse / bw);
    float Linv = cinv == 0. ? 0.f : (float)cinv * (float)(inv / bw);
    float L = ((float)(bw / (double)m) / Wmean) * (Lp + Lv + Lreg + Linv);
    float Hn = (float)(-hn / bw);
    aL += (double)L * bw; aLp += (double)Lp * bw; aLv += (double)Lv * bw; aLreg += (double)Lreg * bw;
    aLinv += (double)Linv * bw; aHn += (double)Hn * bw; aw += bw;
  }
  out->L = (float)(aL / aw); out->Lp = (float)(aLp / aw); out->Lv = (float)(aLv / aw); out->Lreg = (float)(aLreg / aw);
  out->Linv = (float)(aLinv / aw); out->Hp = Hp; out->Hpnet = (float)(aHn / aw); out->Wmean = Wmean;
  free(net.pk);
}
size_t azr_sizeof_sample(void) { return sizeof(azr_sample); }

/* ================================ misc =================================== */
int azr_numerics_selftest(void) { return rn_selftest(); }
float azr_expf(float x) { return rn_expf(x); }
float azr_tanhf(float x) { return rn_tanhf(x); }
double azr_log(double x) { return rn_log(x); }
double azr_exp(double x) { return rn_exp(x); }
double azr_pow(double x, double y) { return rn_pow(x, y); }
void azr_philox(const uint32_t* ctr, const uint32_t* key, uint32_t* out) { rn_philox(ctr, key, out); }
void azr_dirichlet(uint64_t seed, uint32_t game, uint32_t move, int n, double alpha, double* eta) {
  rn_stream r = rn_open(seed, game, move, RN_NOISE); rn_dirichlet(&r, n, alpha, eta);
}
float azr_move_uniform(uint64_t seed, uint32_t game, uint32_t move) {
  rn_stream r = rn_open(seed, game, move, RN_MOVE); return rn_u32(&r);
}
double azr_log2(double x) { return rn_log2(x); }
float azr_logf(float x) { return rn_logf(x); }
uint64_t azr_hash_key(uint64_t a, uint64_t b) { return rn_hash_key(a, b); }
int azr_categorical(const float* p, int n, float u) { return rn_categorical(p, n, u); }
/* n f64 uniforms followed by n f32 uniforms of one stream (draws 0 .. 2n-1) */
void azr_stream_uniforms(uint64_t seed, uint32_t game, uint32_t move, uint32_t purpose, int n, double* u64, float* u32) {
  rn_stream s = rn_open(seed, game, move, purpose);
  for (int i = 0; i < n; ++i) u64[i] = rn_u64(&s);
  for (int i = 0; i < n; ++i) u32[i] = rn_u32(&s);
}
size_t azr_sizeof_state(void) { return sizeof(azr_state); }
size_t azr_sizeof_env(void) { return sizeof(azr_env); }
size_t azr_sizeof_move_rec(void) { return sizeof(azr_move_rec); }
size_t azr_sizeof_game_rec(void) { return sizeof(azr_game_rec); }
size_t azr_sizeof_sim_params(void) { return sizeof(azr_sim_params); }
