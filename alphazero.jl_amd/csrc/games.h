// games.h -- device twins of the reference's game plugins (GameInterface, src/game.jl:34-336).
//
// The reference keeps a mutable GameEnv per game (games/*/game.jl); the search only needs
// value semantics, so a twin is a 20-byte register-resident struct: the 16-byte packed state
// key of include/azhip.h plus the env-only fields (`finished`, `winner`) that are NOT part of
// the state tuple.  All functions are branch-light integer code usable from host and device.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define AZ_GHD __host__ __device__ inline
#else
#define AZ_GHD inline
#endif

struct GEnv {
  uint64_t a, b;   // packed key (bit 63 of a: BLACK to move)
  uint32_t fin;    // bit 0 finished, bits 1..2 winner (0 none, 1 WHITE, 2 BLACK)
};

static constexpr uint64_t AZ_BLACK_BIT = 1ULL << 63;

AZ_GHD int az_popc64(uint64_t x) { return __builtin_popcountll(x); }

// ------------------------------------------------------------------ Connect Four
// games/connect-four/game.jl.  Bitboards with 7 bits per column (bit col*7+row, row 0 = bottom,
// bit 6 of each column always clear so shifts never wrap between columns).
struct ConnectFour {
  static constexpr int ID = 0, A = 7, APAD = 8, W = 7, H = 6, C = 3, P = 42;
  static constexpr int MAX_PLIES = 42;          // a game lasts at most 42 moves
  // row-5 (top) bit of every column: sum over c of 1 << (7c + 5)
  static constexpr uint64_t TOP = (1ULL << 5) | (1ULL << 12) | (1ULL << 19) | (1ULL << 26) |
                                  (1ULL << 33) | (1ULL << 40) | (1ULL << 47);

  AZ_GHD static GEnv init() { return GEnv{0, 0, 0}; }   // game.jl:39-48
  AZ_GHD static bool has4(uint64_t p) {                 // == winning_pattern_at over all stones, game.jl:105-127
    uint64_t m = p & (p >> 7);
    if (m & (m >> 14)) return true;   // axis (1,0)
    m = p & (p >> 6);
    if (m & (m >> 12)) return true;   // axis (1,-1)
    m = p & (p >> 8);
    if (m & (m >> 16)) return true;   // axis (1,1)
    m = p & (p >> 1);
    return (m & (m >> 2)) != 0;       // axis (0,1)
  }
  AZ_GHD static uint32_t mask(const GEnv& g) {          // update_actions_mask!, game.jl:95-99
    uint64_t occ = (g.a | g.b) & ~AZ_BLACK_BIT;
    uint32_t m = 0;
#pragma unroll
    for (int c = 0; c < 7; ++c) m |= (uint32_t)(((occ >> (c * 7 + 5)) & 1) ^ 1) << c;
    return m;
  }
  // set_state!, game.jl:50-68.  The reference tests only the top stone of each column; for every
  // state reachable by play! (the game stops at the first alignment) that equals "any alignment".
  AZ_GHD static GEnv from_key(uint64_t a, uint64_t b) {
    GEnv g{a, b, 0};
    uint64_t w = a & ~AZ_BLACK_BIT;
    if ((((w | b) & TOP) == TOP)) g.fin = 1;
    if (has4(w)) g.fin = 1 | (1u << 1);
    else if (has4(b)) g.fin = 1 | (2u << 1);
    return g;
  }
  AZ_GHD static bool white_playing(const GEnv& g) { return !(g.a & AZ_BLACK_BIT); }
  AZ_GHD static void play(GEnv& g, int col) {           // play!, game.jl:140-146
    uint64_t w = g.a & ~AZ_BLACK_BIT;
    uint64_t occ = w | g.b;
    int row = az_popc64((occ >> (col * 7)) & 0x3f);      // first_free, game.jl:87-93
    uint64_t bit = 1ULL << (col * 7 + row);
    bool wp = white_playing(g);
    uint64_t mine = (wp ? w : g.b) | bit;
    occ |= bit;
    uint32_t fin = 0;
    if (has4(mine)) fin = 1 | ((wp ? 1u : 2u) << 1);     // update_status!, game.jl:130-138
    else if ((occ & TOP) == TOP) fin = 1;
    if (wp) g.a = mine | AZ_BLACK_BIT; else { g.a = w; g.b = mine; }
    g.fin = fin;
  }
  AZ_GHD static float white_reward(const GEnv& g) {     // game.jl:160-168
    uint32_t wn = g.fin >> 1;
    return (g.fin & 1) ? (wn == 1 ? 1.f : wn == 2 ? -1.f : 0.f) : 0.f;
  }
  // GI.symmetries, game.jl:243-257: ONE symmetry, the column mirror (sigma = 7..1)
  static constexpr int NSYM = 1;
  AZ_GHD static uint64_t mirror(uint64_t x) {
    uint64_t out = 0;
#pragma unroll
    for (int c = 0; c < 7; ++c) out |= ((x >> (7 * c)) & 0x7f) << (7 * (6 - c));
    return out;
  }
  AZ_GHD static GEnv sym(const GEnv& g, int) {
    return GEnv{mirror(g.a & ~AZ_BLACK_BIT) | (g.a & AZ_BLACK_BIT), mirror(g.b), g.fin};
  }
  AZ_GHD static int sym_action(int, int j) { return 6 - j; }   // aperm: pi'[j] = pi[sigma[j]], memory.jl:122-126
  // vectorize_state, game.jl:226-241: plane c in (EMPTY, player to move, opponent)
  AZ_GHD static float plane(const GEnv& g, int p /* x + 7*y */, int c) {
    int x = p % 7, y = p / 7;
    uint64_t w = g.a & ~AZ_BLACK_BIT;
    bool wp = white_playing(g);
    uint64_t me = wp ? w : g.b, op = wp ? g.b : w;
    int bit = x * 7 + y;
    uint32_t mb = (uint32_t)(me >> bit) & 1, ob = (uint32_t)(op >> bit) & 1;
    uint32_t v = c == 0 ? (1u ^ mb ^ ob) : c == 1 ? mb : ob;
    return (float)v;
  }
};

// ------------------------------------------------------------------ Tic-tac-toe
// games/tictactoe/game.jl.  a = WHITE marks (bits 0..8), b = BLACK marks.
struct TicTacToe {
  static constexpr int ID = 1, A = 9, APAD = 16, W = 3, H = 3, C = 3, P = 9;
  static constexpr int MAX_PLIES = 9;
  AZ_GHD static bool won(uint32_t m) {                  // has_won, game.jl:52-58
    return ((m & 0x007) == 0x007) | ((m & 0x038) == 0x038) | ((m & 0x1C0) == 0x1C0) |
           ((m & 0x049) == 0x049) | ((m & 0x092) == 0x092) | ((m & 0x124) == 0x124) |
           ((m & 0x111) == 0x111) | ((m & 0x054) == 0x054);
  }
  AZ_GHD static uint32_t status(uint64_t a, uint64_t b) {   // terminal_white_reward, game.jl:75-82
    uint32_t w = (uint32_t)a & 0x1ff, k = (uint32_t)b & 0x1ff;
    if (won(w)) return 1 | (1u << 1);
    if (won(k)) return 1 | (2u << 1);
    if ((w | k) == 0x1ff) return 1;
    return 0;
  }
  AZ_GHD static GEnv init() { return GEnv{0, 0, 0}; }
  AZ_GHD static GEnv from_key(uint64_t a, uint64_t b) { return GEnv{a, b, status(a, b)}; }
  AZ_GHD static uint32_t mask(const GEnv& g) { return ~((uint32_t)g.a | (uint32_t)g.b) & 0x1ff; }  // game.jl:69
  AZ_GHD static bool white_playing(const GEnv& g) { return !(g.a & AZ_BLACK_BIT); }
  AZ_GHD static void play(GEnv& g, int pos) {           // play!, game.jl:89-92
    if (white_playing(g)) g.a = (g.a | (1ULL << pos)) | AZ_BLACK_BIT;
    else { g.b |= 1ULL << pos; g.a &= ~AZ_BLACK_BIT; }
    g.fin = status(g.a, g.b);
  }
  AZ_GHD static float white_reward(const GEnv& g) {
    uint32_t wn = g.fin >> 1;
    return (g.fin & 1) ? (wn == 1 ? 1.f : wn == 2 ? -1.f : 0.f) : 0.f;
  }
  // GI.symmetries, game.jl:149-168: the 7 non-trivial dihedral maps in the reference's order
  // (rot, rot2, rot3, flip, flip.rot, flip.rot2, flip.rot3); new board[p] = board[src(k, p)], where
  // src = pos_of_xy . f . xy_of_pos with rot(x,y) = (y, 2-x), flip(x,y) = (x, 2-y) (0-based).
  static constexpr int NSYM = 7;
  AZ_GHD static int sym_src(int k, int p) {
    int x = p % 3, y = p / 3;
    const int nrot = k < 3 ? k + 1 : k - 3;
    for (int r = 0; r < nrot; ++r) { int t = x; x = y; y = 2 - t; }
    if (k >= 3) y = 2 - y;
    return y * 3 + x;
  }
  AZ_GHD static GEnv sym(const GEnv& g, int k) {
    uint64_t na = g.a & AZ_BLACK_BIT, nb = 0;
    for (int p = 0; p < 9; ++p) {
      const int q = sym_src(k, p);
      na |= ((g.a >> q) & 1) << p;
      nb |= ((g.b >> q) & 1) << p;
    }
    return GEnv{na, nb, g.fin};
  }
  AZ_GHD static int sym_action(int k, int j) { return sym_src(k, j); }   // the action permutation is the board permutation
  AZ_GHD static float plane(const GEnv& g, int p, int c) {  // vectorize_state, game.jl:126-143
    bool wp = white_playing(g);
    uint32_t w = (uint32_t)g.a & 0x1ff, k = (uint32_t)g.b & 0x1ff;
    uint32_t me = wp ? w : k, op = wp ? k : w;
    uint32_t mb = (me >> p) & 1, ob = (op >> p) & 1;
    uint32_t v = c == 0 ? (1u ^ mb ^ ob) : c == 1 ? mb : ob;
    return (float)v;
  }
};

// ------------------------------------------------------------------ Mancala
// games/mancala/game.jl.  a: bytes 0..5 WHITE houses 1..6, byte 6 WHITE store; b likewise BLACK.
struct Mancala {
  static constexpr int ID = 2, A = 6, APAD = 8, W = 14, H = 1, C = 5, P = 14;
  static constexpr int MAX_PLIES = 256;        // capacity, not a rule: overflow is reported
  static constexpr uint64_t HOUSES = 0x0000FFFFFFFFFFFFULL;
  AZ_GHD static uint32_t sum_houses(uint64_t w) {       // sum_houses, game.jl:135
    return (uint32_t)(((w & HOUSES) * 0x0101010101010101ULL) >> 56);
  }
  AZ_GHD static uint32_t byte_at(uint64_t w, int i) { return (uint32_t)(w >> (8 * i)) & 0xff; }
  AZ_GHD static GEnv init() { return GEnv{0x0000030303030303ULL, 0x0000030303030303ULL, 0}; }
  AZ_GHD static bool white_playing(const GEnv& g) { return !(g.a & AZ_BLACK_BIT); }
  AZ_GHD static GEnv from_key(uint64_t a, uint64_t b) { // set_state!, game.jl:54-60
    GEnv g{a, b, 0};
    if (sum_houses(a) == 0 || sum_houses(b) == 0) g.fin = 1;
    return g;
  }
  AZ_GHD static uint32_t mask(const GEnv& g) {          // actions_mask, game.jl:121-123
    uint64_t w = white_playing(g) ? g.a : g.b;
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) m |= (uint32_t)(byte_at(w, i) != 0) << i;
    return m;
  }
  // play!, game.jl:144-177.  Sowing ring of the mover (next_pos, game.jl:80-97): own houses
  // num-1 … 1, own store, opponent houses 6 … 1, then own houses 6 … (skips the other store).
  AZ_GHD static void play(GEnv& g, int a0) {
    bool wp = white_playing(g);
    uint64_t mine = (wp ? g.a : g.b) & ~AZ_BLACK_BIT, theirs = (wp ? g.b : g.a) & ~AZ_BLACK_BIT;
    int idx = 5 - a0;                                   // ring index of house num = a0+1 is 6-num
    uint32_t nseeds = byte_at(mine, a0);
    mine &= ~(0xffULL << (8 * a0));
    for (uint32_t i = 0; i < nseeds; ++i) {
      idx = idx == 12 ? 0 : idx + 1;
      if (idx <= 5) mine += 1ULL << (8 * (5 - idx));
      else if (idx == 6) mine += 1ULL << 48;
      else theirs += 1ULL << (8 * (12 - idx));
    }
    bool finished = false, sw = false;
    if (sum_houses(mine) == 0) {                        // mover has no seeds left
      theirs = ((uint64_t)(byte_at(theirs, 6) + sum_houses(theirs))) << 48;   // capture_leftovers(other)
      mine &= 0xffULL << 48;
      finished = true;
    } else if (idx != 6) {
      if (idx <= 5 && byte_at(mine, 5 - idx) == 1) {    // last seed in an empty own house
        int hb = 5 - idx;                               // byte of house num = hb+1; opposite num' = 6-num+1 -> byte 5-hb
        uint32_t cap = byte_at(theirs, 5 - hb) + 1;
        mine = (mine & ~(0xffULL << (8 * hb))) + ((uint64_t)cap << 48);
        theirs &= ~(0xffULL << (8 * (5 - hb)));
        if (sum_houses(theirs) == 0) {                  // capture_leftovers(mover)
          mine = ((uint64_t)(byte_at(mine, 6) + sum_houses(mine))) << 48;
          theirs &= 0xffULL << 48;
          finished = true;
        } else if (sum_houses(mine) == 0) {             // capture_leftovers(other)
          theirs = ((uint64_t)(byte_at(theirs, 6) + sum_houses(theirs))) << 48;
          mine &= 0xffULL << 48;
          finished = true;
        }
      }
      if (!finished) sw = true;                         // early `return`s keep the mover
    }
    bool next_white = sw ? !wp : wp;
    uint64_t na = wp ? mine : theirs, nb = wp ? theirs : mine;
    g.a = na | (next_white ? 0 : AZ_BLACK_BIT);
    g.b = nb;
    g.fin = finished ? 1u : 0u;
  }
  AZ_GHD static float white_reward(const GEnv& g) {     // game.jl:187-206
    if (!(g.fin & 1)) return 0.f;
    uint32_t nw = byte_at(g.a, 6), nb = byte_at(g.b, 6);
    return nw > nb ? 1.f : nw < nb ? -1.f : 0.f;
  }
  static constexpr int NSYM = 0;                        // no GI.symmetries method: apply_random_symmetry! asserts
  AZ_GHD static GEnv sym(const GEnv& g, int) { return g; }
  AZ_GHD static int sym_action(int, int j) { return j; }
  // vectorize_state, game.jl:224-257, BUG-COMPATIBLE: when BLACK is to move flip_colors returns
  // the INITIAL board.  Positions: WHITE houses 6..1, WHITE store, BLACK houses 6..1, BLACK store.
  AZ_GHD static float plane(const GEnv& g, int p, int c) {
    bool wp = white_playing(g);
    int player = p < 7 ? 0 : 1, j = p < 7 ? p : p - 7;
    bool store = (j == 6);
    if (c == 0) {
      if (!wp) return store ? 0.f : 3.f;
      uint64_t w = player == 0 ? g.a : g.b;
      return (float)byte_at(w, store ? 6 : 5 - j);
    }
    if (c == 1) return (!store && player == 0) ? 1.f : 0.f;
    if (c == 2) return (store && player == 0) ? 1.f : 0.f;
    if (c == 3) return (!store && player == 1) ? 1.f : 0.f;
    return (store && player == 1) ? 1.f : 0.f;
  }
};


// ------------------------------------------------------------------ 9x9 boards with 4 planes and 82 actions (network only)
// BASELINE configs[4]: OpenSpiel 9x9 Go through src/openspiel.jl.  Foreign rules cannot run on the device: such games keep
// their rules and their tree on the host and use the network through az_net_forward (planes in, policy / value out --
// Network.forward_normalized, src/networks/network.jl:264-271).  This type only carries the tensor geometry the network
// kernels are instantiated for: GI.state_dim = (9, 9, 4) (OpenSpiel's go observation tensor: black, white, empty, to-play /
// komi plane), GI.num_actions = 82 (81 points + pass).  There is no device twin: plane / mask are never called (the
// FROM_PLANES = false kernels are not instantiated for it) and every search entry point rejects the game id.
struct Go9Planes {
  static constexpr int ID = 3, A = 82, APAD = 88, W = 9, H = 9, C = 4, P = 81;
  static constexpr int MAX_PLIES = 1;
  static constexpr int NSYM = 0;
  AZ_GHD static float plane(const GEnv&, int, int) { return 0.f; }
  AZ_GHD static uint32_t mask(const GEnv&) { return 0; }
};
