// tree.h -- search-tree kernels: the device form of MCTS.Env (src/mcts.jl) and of the move
// step of play_game (src/play.jl:298-315).
//
// Layout in HBM (one engine = G game slots, every array is slot-major):
//   nodes  [G][cap][NODE_BYTES]   one record per stored state (the reference's StateInfo +
//                                 Vector{ActionStats}, src/mcts.jl:78-87), full action width:
//                                   per action 16 B { W f64 | N i32 | P f32 } | child links (18 bit per action)
//                                 (128 B for 7 actions: one cache line per visit, ONE 16-byte load per lane, and the
//                                 read-modify-write of a backup touches one 32-byte sector instead of two)
//                                 Big pools live in a VIRTUAL range whose 2 MB chunks are mapped on demand (node_at)
//   keys   [G][cap][4] u64        side record of every node: state key (only hash probes read it) | Vest f32 | unused
//   ht     [G][H] u64             the Dict{State,StateInfo} of src/mcts.jl:126 as an open-addressed
//                                 table: epoch(16) | tag(16) | node index+1 (32); an entry is live
//                                 only if its epoch equals the slot's epoch, so MCTS.reset! is O(1)
//   path   [G][max_depth] u64     explicit stack replacing the recursion of run_simulation!
//
// Child links.  tree[state] is a pure function of the state for as long as the tree is not reset, so the node a
// (node, action) edge leads to is memoised in the parent's record the first time the Dict lookup answers it (at
// expansion, or when a probe finds a transposition).  A descent then costs ONE dependent 128-byte load per ply; the
// hash table is probed only on edges never taken before.  Results cannot change: a link is the answer the probe
// would give (tests compare every visit count with the oracle's Dict walk).
//
// Thread mapping: APAD lanes per slot (8 for 7/6 actions, 16 for 9), lane a owns action a.  A
// 64-wide wavefront therefore advances 8 (or 4) slots; N/P/W rows are read as one coalesced
// 32/32/64-byte segment per slot, sums and argmax are wavefront shuffles inside the lane group.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/az_numerics.h"
#include "../../include/azhip.h"
#include "games.h"

enum { LEAF_NONE = 0, LEAF_NEW = 1, LEAF_TERMINAL = 2 };
enum { DERR_NODE_POOL = 1, DERR_HASH_FULL = 2, DERR_DEPTH = 3, DERR_MOVES = 4, DERR_NO_ROOT = 5 };

struct DParams {
  double gamma, cpuct, eps, alpha, prior_temp;
  int nsims, temp_len;
  int temp_xs[AZ_SCHED_MAX];
  double temp_ys[AZ_SCHED_MAX];
  uint64_t seed;
  int oracle, reset_every;
  double flip_p;          // play_game's flip_probability (play.jl:305-307) for the self-play loop on the device; the arena flips on the host
  int retire;             // 1 (self-play): a slot whose node pool or move record overflows is RETIRED (finished = 2, the game is
                          // reported as aborted, the phase goes on); 0 (explore! / arena hooks): a device error
};

// Per-slot search state (round 4): what used to be thirteen slot-major arrays -- thirteen 32-byte sectors read and up to eight
// written per slot and wave, of which 4 to 24 bytes each were used -- is one 64-byte record: two sectors.
//   root_a / root_b / root_fin   the live game (GI state of the root, play.jl:299-313)
//   tot_sims / tot_trav          MCTS.Env.total_simulations / total_nodes_traversed (mcts.jl:128-130)
//   leaf_kd                      pending leaf: kind (LEAF_*) | depth << 2 | (free-running) parity of the evaluation batch it is in << 30
//   eidx                         slot -> index in the evaluation batch
//   node_count                   length(env.tree)
//   epoch                        live epoch of the slot's hash table (MCTS.reset! is a new epoch)
//   leaf_ins                     table position the pending leaf will take
//   root_idx                     node index of the current root, -1 = not looked up yet this move
//   active                       bit 0 (SR_ACTIVE): the slot searches; bits 8.. (round 6, free-running phases only): simulations of the
//                                current explore! that are complete -- at num_iters_per_turn the slot waits for k_move_fr
enum { SR_ACTIVE = 1, SR_MS_SHIFT = 8 };
struct alignas(64) SlotRec {
  unsigned long long root_a, root_b;
  long long tot_sims, tot_trav;
  uint32_t root_fin;
  int leaf_kd;
  int eidx, node_count;
  uint32_t epoch, leaf_ins;
  int root_idx, active;
  __host__ __device__ inline GEnv root() const { GEnv g; g.a = root_a; g.b = root_b; g.fin = root_fin; return g; }
  __host__ __device__ inline void set_root(const GEnv& g) { root_a = g.a; root_b = g.b; root_fin = g.fin; }
};
static_assert(sizeof(SlotRec) == 64, "slot record");

// Evaluation cache (round 5): oracle answers by state, shared by all slots of the engine.  The reference evaluates every query on its
// own (src/simulations.jl:23-38, src/networks/network.jl:308-315); a test-mode evaluation is a pure function of the state and every
// tower form gives the same bits (tests/test_net.py), so an answer that was computed once -- for another slot, in an earlier wave --
// IS the answer: the records of a phase do not change (tests/test_eval_cache_gpu.py), the network evaluates each state once while
// the entry lives.  Direct-mapped, 64-byte entries, RAW oracle output (prior_temperature is applied by the slot that stores the node).
//   meta = seq << 2 | state:  0 empty;  EC_PENDING claimed by the slot that sent the state to the network (launch seq of the claim);
//          EC_FILLING its answer is being written (exclusive: taken by CAS);  EC_READY answered.  seq makes every value of meta
//          unique per claim.
//   cs   = checksum over key, P, V and the READY value of meta: a reader takes ONE look at the entry (all of it in one round of
//          loads) and believes it only if the state is READY, the key is its own and the checksum holds, so a look that straddles
//          an eviction + refill -- a mixture of two claims' words -- is refused (2^-32 per such look; they need a concurrent
//          writer of the same entry inside the ~1 us of the look to exist at all).  Writers own the entry (FILLING) while they write,
//          and wait for their stores (s_waitcnt) before the READY that publishes them.
// All accesses are agent-scope relaxed atomics (cache-bypassing loads, write-through stores): workgroups on other XCDs fill and
// evict concurrently in the same launch -- the idiom of k_tower16s' exchange.
struct alignas(64) ECEnt { unsigned long long ka, kb; float P[AZ_MAX_ACTIONS]; float V; uint32_t cs; uint32_t meta; };
static_assert(sizeof(ECEnt) == 64, "evaluation cache entry");
enum { EC_EMPTY = 0, EC_PENDING = 1, EC_READY = 2, EC_FILLING = 3, EC_STALE_AFTER = 64 };

struct DView {
  int G, cap_nodes, ht_size, max_depth, max_moves;
  ECEnt* ec;              // evaluation cache [ec_mask + 1] or NULL (off)
  uint32_t ec_mask, ec_seq;   // ec_seq: number of this k_tree launch (any slot group of the engine), > 0
  uint32_t ec_floor;          // entries whose claim number is <= ec_floor are EMPTY: az_net_set_params empties the table by raising the floor (a 1 GB memset per weight update before)
  int* ec_claim;          // [G][2] cache entry the slot's pending leaf claimed (its answer goes there when the leaf is expanded; -1 = none) and the meta value of the claim
  int* needy_host;        // host-mapped word: the slots this group's previous wave launch left to the background search (free-running; wave_net_f's choice between the paired tower's two register budgets), or NULL
  int* nleaf_host;        // host-mapped word: the network batch of this group's previous wave (pick_tower's launch-size estimate), or NULL
  float* Phit; float* Vhit; // [G][APAD], [G]: the answer of a leaf the cache answered, by SLOT (written by the slot's phase B, read by its next phase A; SlotRec::eidx = -1 says so)
  uint32_t tag_mask;      // 0xffff; tests narrow it (AZHIP_HT_TAG_BITS) so that unequal states share tags and every probe chain reaches the exact key compare
  uint32_t epoch0;        // first live epoch of a slot's table: 1; tests start near the 16-bit wrap (AZHIP_HT_EPOCH0)
  SlotRec* sr;            // [G] the search state of a slot that k_tree reads and writes every wave, ONE 64-byte record (round 4)
  uint32_t* game_id;
  uint32_t* move_idx;
  int* worker_sim_id;
  double* eta;            // [G][APAD] by full action index
  unsigned long long* ht;
  char* nodes;
  // node idx of slot s lives at nodes + ((idx >> node_sh) * node_row + s * node_stride) + (idx & node_mask) * NODE_BYTES.
  // Plain pool: node_sh = 31 (one "chunk" = the slot's whole region, node_stride = cap * NODE_BYTES).  Mapped-on-demand pool:
  // chunks of 2^node_sh nodes = 2 MB, chunk row r of all slots contiguous (node_row = G * 2 MB), so that the host can back
  // row r of a run of slots with one mapping; slot_cap[s] = nodes of slot s that are backed by memory right now.
  int node_sh; uint32_t node_mask; size_t node_row, node_stride;
  const int* slot_cap;    // [G] or NULL (plain pool: cap_nodes for every slot)
  unsigned long long* keys; // side record of node idx of slot s: state key (2 words), Vest (f32, StateInfo.Vest, src/mcts.jl:86) in word 2, at
                            // keys + (idx >> node_sh) * key_row + s * key_stride + (idx & node_mask) * 4 (in u64 words; side_at).  Plain pool:
                            // [G][cap][4].  Mapped-on-demand pool (round 5): the pieces follow the node chunks -- chunk row r of slot s has its
                            // 2^node_sh side records at piece (r * G + s); the host backs the 2 MB granule a piece lies in together with the
                            // node chunk, so the side records cost a quarter of the nodes that exist instead of a quarter of the worst case
  size_t key_row, key_stride;
  const int* perm;          // [G] or NULL: lane group i of k_tree serves slot perm[i] (AZHIP_TREE_SORT: slots ordered by the depth of their
                            // last explore!, so that the 8 slots of a wavefront finish together; results are by slot, whatever the order)
  int bk_mode;              // backup of phase A: 0 = read-modify-write by the ply's lane; 1 / 2 = no-return atomics at agent scope (AZHIP_TREE_ATOMIC,
                            // 1: phase B waits for them and drops the CU's L1 copy, 2: only drops the copy -- an experiment, see k_tree)
  unsigned long long* path;
  GEnv* leaf_env;         // [G] state of the slot's pending leaf (read by the network kernels through eval_slots)
  int* eval_slots;        // [2][eval_stride] evaluation batch -> slot, double-buffered by wave parity like n_eval (a free-running phase's
                          // background search adds to the NEXT wave's batch while the network reads this wave's)
  int eval_stride;
  const int* bg_stop; int bg_seq;   // background launch: leave when *bg_stop >= bg_seq -- the host's stream sets the word when the wave's tower has run (NULL: run to run_k)
  int* bg_list; int* bg_cnt; // free-running: [2][eval_stride] / [2] the slots a wave's launch left without a question for the network (by wave parity):
                          // the background launch that follows serves exactly these -- a handful of wavefronts instead of one per 8 slots
  int* n_eval;            // [2] leaves of the wave, double-buffered by wave parity (k_tree zeroes the other one)
  float* Pout;            // [n_eval][APAD] masked-normalised priors, full width
  float* Vout;            // [n_eval]
  az_move_rec* trace;     // [G][max_moves]
  az_game_rec* grec;      // [G]
  int* finished;          // [G] set by k_move when the slot's game ended this round
  int* err;               // device error word (first error wins)
  int* xerr;              // the slot group's own word for "an exchange of this group's split tower gave up" (DERR_EXCHANGE): set by the
                          // group's tower, read by the group's NEXT k_tree -- same stream order, so a launch sees one value throughout
  int* skipped;           // [2] launches of k_tree that did nothing because a split tower's exchange had given up: with / without select
  long long* stat;        // [workgroups of k_tree][4]: simulations, nodes traversed, leaf evals, spare -- accumulated per workgroup
  unsigned long long* dbg; // optional [16] cycle stamps of k_tree's first wavefront (az_debug_tree_stamps): start, phase A done,
                          // root loaded, descent done, leaf stored, block atomics done, end
  // (round 6) free-running phases (az_selfplay_*): a slot does not wait for the other slots' simulations.  One k_tree launch carries
  // a slot through up to run_k simulations -- every one that ends on a terminal state or on a state the evaluation cache answers is
  // completed on the spot -- until it needs the network (a cache miss), has done num_iters_per_turn of them (k_move_fr then plays its
  // move, whatever the other slots are doing) or reaches run_k.  0 = lock step: one simulation per slot and launch, moves in rounds.
  int run_k;
  int fr;                 // 1: a free-running PHASE (two batches in flight, background launches, k_move_fr); run_k > 0 with fr = 0: an explore!
                          // of the hooks / the arena that runs ahead -- every slot up to nsims simulations, the host moves (busy_host tells it when)
  int* busy_host;         // host-mapped: slots that had not finished their explore! after the launch before the previous one (run_k > 0, fr = 0), or NULL
  int low_prio;           // 1: a background launch -- it must not take issue slots from the network launch it runs under (no s_setprio)
  int slot0;              // engine-wide index of this view's first slot (az_game_rec.slot)
  int* fr_active;         // free-running: the view's count of searching slots (FRState::active[group]), or NULL
};

// Device words of a free-running phase (k_move_fr and k_tree write, the host reads at its synchronisation points).
static constexpr int AZ_MAX_GROUPS = 4;
struct FRState {
  unsigned long long resv;          // finished games << 40 | move records written to the phase buffer: reserved together by one CAS
  long long moves;                  // moves played (az_selfplay_stats.moves)
  int next_game;                    // index (from first_game_id) of the next game to hand out (Util.mapreduce's `next`, util.jl:169-200)
  int active[AZ_MAX_GROUPS];        // slots of each slot group that search
};
struct FRArgs {
  FRState* st;
  az_game_rec* done; long long* done_off;   // [done_cap] records of the finished games in finishing order, offset of each one's first move record
  az_move_rec* recs; long long recs_cap;    // the phase buffer (bounded phase) or the staging area the host drains (unbounded)
  int done_cap, total_games;                // total_games < 0: refill forever
  uint32_t first_game_id;
  int group;
  int* host_words;                          // host-mapped [2]: finished games, searching slots -- as of the previous wave; the host looks without synchronising
};

// Node record: N i32[A] | P f32[A] | W f64[A] | LO u16[A] | HI, padded to a multiple of 32 B.  With A = 7 that is
// exactly 128 B = ONE cache line per visited node (Connect-Four; 128 B for Mancala's A = 6, 192 B for A = 9).
// Child link of action a = LO[a] | ((HI >> 2a) & 3) << 16 = 1 + node index of the state the action leads to, 0 =
// not known yet (18 bits: pools above LINK_MAX nodes per slot simply never write links and always probe).
// Vest (only read by the explorer hook) and the state keys live in side arrays; the availability mask is recomputed
// from the state in registers.
template <class Gm> struct NodeL {
  static constexpr int A = Gm::A;
  struct Stat { double W; int N; float P; };        // ActionStats (src/mcts.jl:78-82), 16 bytes: lane a loads nd + 16 a as one dwordx4
  static constexpr int OFF_LO = 16 * A;
  static constexpr int HI_BYTES = A <= 8 ? 2 : 4;
  static constexpr int OFF_HI = (OFF_LO + 2 * A + HI_BYTES - 1) / HI_BYTES * HI_BYTES;
  static constexpr int BYTES = (OFF_HI + HI_BYTES + 31) / 32 * 32;
  using hi_t = typename std::conditional<A <= 8, uint16_t, uint32_t>::type;
  static __host__ __device__ inline Stat* stat(char* nd, int a) { return (Stat*)(nd + 16 * a); }
  static __host__ __device__ inline const Stat* stat(const char* nd, int a) { return (const Stat*)(nd + 16 * a); }
};
static_assert(sizeof(NodeL<ConnectFour>::Stat) == 16, "ActionStats record");
// address of node idx of slot `slot` (DView::node_*); v.nodes of a slot-group view is already offset to the group's first slot
template <class Gm> __device__ __forceinline__ char* node_at(const DView& v, int slot, int idx) {
  return v.nodes + (size_t)((uint32_t)idx >> v.node_sh) * v.node_row + (size_t)slot * v.node_stride + (size_t)((uint32_t)idx & v.node_mask) * NodeL<Gm>::BYTES;
}
__device__ __forceinline__ unsigned long long* side_at(const DView& v, int slot, int idx) {
  return v.keys + (size_t)((uint32_t)idx >> v.node_sh) * v.key_row + (size_t)slot * v.key_stride + (size_t)((uint32_t)idx & v.node_mask) * 4;
}
static constexpr int LINK_MAX = (1 << 18) - 2;

__device__ inline void dev_fail(const DView& v, int code) { atomicCAS(v.err, 0, code); }

// ---- lane-group helpers (L = 8 or 16 lanes, group-uniform control flow) -----------------
template <int L> __device__ inline unsigned group_ballot(bool pred) {
  unsigned long long b = __ballot(pred);
  int base = (threadIdx.x & 63) & ~(L - 1);
  return (unsigned)((b >> base) & ((1u << L) - 1));
}
// Reductions inside a slot's lane group (L = 8 or 16 lanes, aligned, inside one 16-lane DPP row) by DPP butterflies:
// partner of step 0 = lane ^ 1 (quad_perm [1,0,3,2]), step 1 = lane ^ 2 ([2,3,0,1]), step 2 = 7 - lane within each half
// row (row_half_mirror), step 3 (L = 16) = 15 - lane (row_mirror).  Every step pairs lanes that hold different partial
// results, so after log2 L steps every lane holds the group's result.  (__shfl_xor compiles to ds_bpermute_b32: an LDS
// round trip of ~120 cycles per step in the middle of the descent's dependent chain -- ten of them per ply.)
template <int STEP> __device__ __forceinline__ int dpp_partner(int x) {
  constexpr int ctrl = STEP == 0 ? 0xB1 : STEP == 1 ? 0x4E : STEP == 2 ? 0x141 : 0x140;
  return __builtin_amdgcn_update_dpp(x, x, ctrl, 0xF, 0xF, false);
}
template <int STEP> __device__ __forceinline__ double dpp_partner(double x) {
  const int lo = dpp_partner<STEP>(__double2loint(x)), hi = dpp_partner<STEP>(__double2hiint(x));
  return __hiloint2double(hi, lo);
}
template <int L> __device__ __forceinline__ int group_sum(int x) {
  static_assert(L == 8 || L == 16, "lane groups of 8 or 16");
  x += dpp_partner<0>(x);
  x += dpp_partner<1>(x);
  x += dpp_partner<2>(x);
  if constexpr (L == 16) x += dpp_partner<3>(x);
  return x;
}
// first maximum: highest score, ties to the lowest lane (argmax, src/mcts.jl:211); `carry` travels with the winner.
// Maximum first, position second.  The round-2 butterfly exchanged (score, index, carry) and decided with two Float64
// compares per round: ~45 instructions, most of them waiting on the previous one -- 876 cycles per ply with one or two
// wavefronts on a SIMD (tools/tree_stamps.py; an all-pairs form with 14 Float64 compares was slower still: 1094).  Here the
// three rounds only carry v_max_f64, ONE compare marks the lanes that hold the maximum, a ballot picks the lowest of them in
// every group (ties, +0 = -0 included, go to the lowest lane as before), and the winner's link is OR-reduced over the group.
template <int CTRL> __device__ __forceinline__ double dpp_ctrl(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
template <int L> __device__ __forceinline__ int group_argmax(double s, int lane, int* carry) {
  static_assert(L == 8 || L == 16, "lane groups of 8 or 16");
  constexpr int Q1 = 0xB1, Q2 = 0x4E, HM = 0x141, RM = 0x140;      // quad_perm [1,0,3,2], [2,3,0,1]; row_half_mirror; row_mirror
  double m = __builtin_fmax(s, dpp_ctrl<Q1>(s));
  m = __builtin_fmax(m, dpp_ctrl<Q2>(m));
  m = __builtin_fmax(m, dpp_ctrl<HM>(m));
  if constexpr (L == 16) m = __builtin_fmax(m, dpp_ctrl<RM>(m));
  const unsigned long long eq = __ballot(s == m);
  const int base = (threadIdx.x & 63) & ~(L - 1);
  const int idx = __ffs((unsigned)((eq >> base) & ((1u << L) - 1u))) - 1;
  int c = (lane == idx) ? *carry : 0;                               // exactly one lane of the group contributes
  c |= __builtin_amdgcn_update_dpp(0, c, Q1, 0xF, 0xF, false);
  c |= __builtin_amdgcn_update_dpp(0, c, Q2, 0xF, 0xF, false);
  c |= __builtin_amdgcn_update_dpp(0, c, HM, 0xF, 0xF, false);
  if constexpr (L == 16) c |= __builtin_amdgcn_update_dpp(0, c, RM, 0xF, 0xF, false);
  *carry = c;
  return idx;
}
template <int L> __device__ __forceinline__ int group_argmax(double s, int lane) { int c = 0; return group_argmax<L>(s, lane, &c); }

// ---- evaluation cache helpers ----------------------------------------------------------------
template <int L> __device__ __forceinline__ uint32_t group_xor(uint32_t x) {
  x ^= (uint32_t)dpp_partner<0>((int)x);
  x ^= (uint32_t)dpp_partner<1>((int)x);
  x ^= (uint32_t)dpp_partner<2>((int)x);
  if constexpr (L == 16) x ^= (uint32_t)dpp_partner<3>((int)x);
  return x;
}
// value of the group's lane 0 in every lane (DPP adds of one non-zero term: no LDS round trip)
template <int L> __device__ __forceinline__ int group_bcast0(int x, int lane) { return group_sum<L>(lane == 0 ? x : 0); }
__device__ __forceinline__ uint32_t ec_ld32(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ec_ld64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ec_ldf(const float* p) { return __uint_as_float(__hip_atomic_load((const uint32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ void ec_st32(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ec_st64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ec_stf(float* p, float v) { __hip_atomic_store((uint32_t*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool ec_cas(uint32_t* p, uint32_t expect, uint32_t want) {
  return __hip_atomic_compare_exchange_strong(p, &expect, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ec_term(float p, int lane) {    // a lane's share of the checksum: its prior's bits, rotated by its action
  const uint32_t b = __float_as_uint(p), r = (uint32_t)(5 * lane + 1) & 31u;
  return (b << r) | (b >> ((32u - r) & 31u));
}
__device__ __forceinline__ uint32_t ec_checksum(uint32_t px, unsigned long long ka, unsigned long long kb, float V, uint32_t ready_meta) {
  unsigned long long h = az_hash_key(ka, kb);
  h = az_mix64(h ^ (((unsigned long long)px << 32) | (unsigned long long)__float_as_uint(V)));
  h = az_mix64(h + (unsigned long long)ready_meta);
  return (uint32_t)(h >> 32) ^ (uint32_t)h;
}
__device__ __forceinline__ uint32_t ec_index(unsigned long long ka, unsigned long long kb, uint32_t mask) {
  return (uint32_t)(az_hash_key(ka, kb) >> 17) & mask;              // other bits than the slot table's position (low) -- and than its tag where the mask allows
}

// ---- Dict lookup: haskey(env.tree, state) (src/mcts.jl:165-174) ---------------------------
// Returns the node index or -1; *ins receives the table position a new entry would take.
template <class Gm>
__device__ inline int ht_lookup(const DView& v, int slot, int lane, unsigned long long ka,
                                unsigned long long kb, uint32_t epoch, uint32_t* ins) {
  constexpr int L = Gm::APAD;
  const unsigned long long hk = az_hash_key(ka, kb);
  const uint32_t H1 = (uint32_t)v.ht_size - 1;
  const uint32_t h0 = (uint32_t)hk & H1;
  const uint32_t tag = (uint32_t)(hk >> 40) & v.tag_mask;
  const unsigned long long* tab = v.ht + (size_t)slot * v.ht_size;
  const int iters = v.ht_size / L;
  for (int i = 0; i < iters; ++i) {
    uint32_t pos = (h0 + (uint32_t)(i * L + lane)) & H1;
    unsigned long long e = tab[pos];
    uint32_t idx1 = (uint32_t)e;
    bool live = ((uint32_t)(e >> 48) == epoch) && idx1 != 0;
    bool match = false;
    if (live && ((uint32_t)(e >> 32) & 0xffff) == tag) {
      const unsigned long long* k = side_at(v, slot, (int)(idx1 - 1));
      match = (k[0] == ka) && (k[1] == kb);
    }
    unsigned mb = group_ballot<L>(match), db = group_ballot<L>(!live);
    int fm = mb ? __ffs(mb) - 1 : 99, fd = db ? __ffs(db) - 1 : 99;
    if (fm < fd) {
      int base = (threadIdx.x & 63) & ~(L - 1);
      return (int)__shfl((int)idx1, base + fm) - 1;
    }
    if (fd != 99) { *ins = (h0 + (uint32_t)(i * L + fd)) & H1; return -1; }
  }
  dev_fail(v, DERR_HASH_FULL);
  *ins = 0;
  return -1;
}

// ---- dirichlet_noise (src/mcts.jl:228-232): one draw per explore!, by rank among available --
template <class Gm>
__device__ inline void arm_noise(const DView& v, const DParams& p, int slot, const GEnv& env) {
  constexpr int L = Gm::APAD;
  uint32_t m = Gm::mask(env);
  int n = __popc(m);
  double eta[AZ_MAX_ACTIONS];
  az_rng r = az_rng_make(p.seed, v.game_id[slot], v.move_idx[slot], AZ_RNG_NOISE);
  az_dirichlet(&r, n, p.alpha, eta);
  int k = 0;
  for (int a = 0; a < L; ++a) {
    double e = 0.0;
    if (a < Gm::A && ((m >> a) & 1)) e = eta[k++];
    v.eta[(size_t)slot * L + a] = e;
  }
}

// write the child link of (node, act): lane-0 work of a slot's lane group
template <class Gm> __device__ inline void set_link(char* nd, int act, uint32_t link) {
  using NL = NodeL<Gm>;
  ((uint16_t*)(nd + NL::OFF_LO))[act] = (uint16_t)(link & 0xffff);
  typename NL::hi_t* hp = (typename NL::hi_t*)(nd + NL::OFF_HI);
  *hp = (typename NL::hi_t)((*hp & ~((typename NL::hi_t)3 << (2 * act))) | ((typename NL::hi_t)(link >> 16) << (2 * act)));
}

// =========================================================================================
// k_tree: ONE kernel per search wave and slot group.
//   phase A (do_backup): the second half of a run_simulation! -- init_state_info for the new leaf with the oracle's answer
//           (mcts.jl:157-161, util.jl:98-110), update_state_info! along the path (mcts.jl:190-194, 218-223)
//   phase B (do_select): the descent of the next run_simulation! (mcts.jl:199-217) until an unseen or a terminal state, and the
//           compaction of the new leaves into the evaluation batch (Batchifier.launch_server, batchifier.jl:47-81; the batch order
//           is arbitrary -- test-mode evaluations are independent, Appendix A.13)
// The network runs between two launches; the last simulation of an explore! is completed by a launch with do_select = 0.
// Lock step (DView::run_k = 0): one A and one B per slot and launch -- a wave is one simulation of every slot.
// Free-running (run_k > 0, round 6): A, B, A, B ... inside the launch for as long as the slot's simulations end on a terminal state or on
// a state the evaluation cache answers, i.e. until the slot really has a question for the network (or has finished its explore!, or
// has selected run_k times).  A worker of the reference runs its simulations one after the other and independently of the other
// workers (mcts.jl:239-245, simulations.jl:216-243) and the oracle is a pure function of the state, so the slot's sequence of
// simulations -- and with it every record -- is the same; what changes is that a wave's network batch holds one board of EVERY
// searching slot instead of the 44 % of them whose simulation happened to need one.
// APAD lanes per slot (lane = action).  Dependent loads: phase A path -> statistics of the path's nodes (all at once,
// one lane per ply); phase B one node record per ply (child links), a table probe only on edges never taken before.
// =========================================================================================
struct KArgs { DView v; DParams p; };   // the head of k_tree's kernarg segment
template <class Gm>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(160))) k_tree(DView v_arg, DParams p_arg, int do_backup, int do_select, int par) {   // (amdgpu_num_vgpr counts half of the unified register file on gfx90a and later: a budget of 320, without it the loop spills.)  The kernel takes 149 (allocated: 152): a wavefront fits beside the two of a 176-register tower form on a SIMD (k_tower16, k_tower16x2c: 2 x 176 + 152 of 512), not beside k_tower16x2's two of 198 -- net_impl.h wave_net_f picks the form by that
  if (!v_arg.low_prio) __builtin_amdgcn_s_setprio(3);   // short latency-bound kernel on the wave's critical path: win issue arbitration against co-resident tower waves
  const DView& v = v_arg;
  constexpr int L = Gm::APAD;
  using NL = NodeL<Gm>;
  __shared__ int s_new[4], s_hit[4], s_evl[4], s_sims[4], s_trav[4], s_need[4], s_base, s_nbase;   // (workgroups of 1024 threads -- a quarter of the returning atomics -- gain 10 % at 64 k slots and lose 30 % at 256 k and 1 M)
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int lgrp = tid / L, lane = tid % L;
  const int wl = threadIdx.x & 63, w = threadIdx.x >> 6, gbase = wl & ~(L - 1);
  // which slot this lane group advances: its own index, (AZHIP_TREE_SORT) the slot the last move step's ordering put here, or -- the
  // background launch of a free-running phase -- the next entry of the list the wave's launch left
  int nlive = v.G;
  const int* map = v.perm;
  if (v.fr && !do_backup && v.bg_list) { nlive = v.bg_cnt[par ^ 1]; map = v.bg_list + (size_t)(par ^ 1) * v.eval_stride; }
  const bool live = lgrp < nlive;
  const int slot_in = live ? (map ? map[lgrp] : lgrp) : 0;
  const bool links_ok = v.cap_nodes <= LINK_MAX;
  unsigned long long* dbg = (v.dbg && blockIdx.x == 0 && threadIdx.x == 0) ? v.dbg : nullptr;
  if (dbg) dbg[0] = __builtin_readcyclecounter();
  // The oracle's answers of this group's pending wave are not to be trusted once an exchange of k_tower16s has given up (the host
  // has not noticed yet: launches are queued ahead): do nothing, be counted; the host re-evaluates the pending leaves without
  // the split and launches the skipped waves again (azhip.hip recover_split).  Uniform over the grid: the word belongs to this
  // slot group, only the group's own tower (earlier on the same stream order) sets it and only the host clears it, between launches.
  if (__hip_atomic_load(v.xerr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)DERR_EXCHANGE) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && (do_backup || !v.fr)) atomicAdd(v.skipped + (do_select ? 0 : 1), 1);   // (a background launch is not a wave)
    return;
  }

  // Everything whose address does not depend on another load is requested up front, in one round trip: the pending leaf's
  // kind / depth / batch index, the node count, the first L path entries (one per lane) and, for phase B, the root state and
  // the epoch.  (Inside the branches below they were three dependent rounds: kind -> depth, eidx -> path, Pout.)
  const SlotRec s0 = v.sr[slot_in];                                 // 64 bytes, the same for all lanes of the slot
  unsigned long long st = (v.path + (size_t)slot_in * v.max_depth)[lane < v.max_depth ? lane : 0];   // path entry `lane` of the simulation in hand (phase B keeps it current)
  GEnv lenv = v.leaf_env[slot_in];                                  // state of its leaf
  int kind = s0.leaf_kd & 3, depth = (s0.leaf_kd >> 2) & 0x0fffffff, e0 = s0.eidx, nc = s0.node_count;
  GEnv root0 = s0.root();
  uint32_t epoch0 = s0.epoch, ins = s0.leaf_ins;
  int ridx = s0.root_idx, active0 = s0.active;
  int claim0 = -1; uint32_t cmeta0 = 0;                             // the pending leaf's cache entry (its answer is stored there), if it holds one
  if (v.ec) { const int2 c2 = ((const int2*)v.ec_claim)[slot_in]; claim0 = c2.x; cmeta0 = (uint32_t)c2.y; }
  // (The compiler sinks a load into the branch that uses it, which turned this one round trip into several dependent ones; the
  // values are therefore pinned into registers right here -- one wait for all of them.)
#define AZ_PIN(x) asm volatile("" : "+v"(x))
  AZ_PIN(kind); AZ_PIN(depth); AZ_PIN(e0); AZ_PIN(nc); AZ_PIN(st); AZ_PIN(root0.a); AZ_PIN(root0.b); AZ_PIN(root0.fin);
  AZ_PIN(epoch0); AZ_PIN(ins); AZ_PIN(claim0); AZ_PIN(cmeta0); AZ_PIN(lenv.a); AZ_PIN(lenv.b); AZ_PIN(lenv.fin); AZ_PIN(ridx); AZ_PIN(active0);
  // Free-running: which of the slots are this launch's business.  A wave's launch (do_backup) completes the simulations whose leaves
  // were in the PREVIOUS wave's batch -- the network has answered them -- and selects on; the background launch that follows it
  // (do_backup = 0, under the wave's network launch) carries on with the slots that have no question pending, and what they find joins the
  // NEXT wave's batch (`par`): a slot waiting for the network, and in the next wave's launch a slot that is already in the batch being
  // gathered, is left alone.
  const bool mine = live && !(v.fr && kind != LEAF_NONE && (!do_backup || ((s0.leaf_kd >> 30) & 1) == par));
  if (!(do_backup && mine)) kind = LEAF_NONE;
  const int nc_in = nc, ridx_in = ridx;
  const bool inrec = lane < Gm::A;
  // the oracle's answer to the pending leaf: from the network's batch, or (eidx = -1, lock step only) from the evaluation cache -- the
  // same bits either way
  float Pans = 0.f, Vans = 0.f;
  if (kind == LEAF_NEW) {
    Pans = e0 >= 0 ? v.Pout[(size_t)e0 * L + lane] : v.Phit[(size_t)slot_in * L + lane];
    Vans = e0 >= 0 ? v.Vout[e0] : v.Vhit[slot_in];
  }
  const bool searching = mine && (active0 & SR_ACTIVE);
  int msims = active0 >> SR_MS_SHIFT;                               // free-running: simulations of this explore! that are complete
  int done_sims = 0, done_trav = 0;                                 // completed in this launch (mcts.jl:242, :222)
  int n_sel = 0, n_trav = 0, n_evl = 0, n_hit = 0;                  // started in this launch: simulations, their depths, oracle calls, of those answered by the cache
  bool retired = false, ishit = false;
  int claim = -1; uint32_t cmeta = 0;

  for (int it = 0;; ++it) {
    // Every iteration reads the kernel's arguments again (scalar loads from the kernarg segment through a pointer the compiler cannot
    // see through) and recomputes its addresses from a laundered slot index: with the arguments and the per-lane addresses of BOTH phases
    // loop-invariant the compiler kept all of them in registers across the loop -- 128 VGPRs + 52 bytes of scratch and 102 spilled
    // SGPRs against 73 / 0 / 4 for the straight-line kernel of round 5.
    KArgs ka;
    {
      auto kp = __builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(kp));
      __builtin_memcpy(&ka, (const void*)kp, sizeof(KArgs));
    }
    const DView& v = ka.v;
    const DParams& p = ka.p;
    int slot = slot_in; asm volatile("" : "+v"(slot));
    unsigned long long* const path = v.path + (size_t)slot * v.max_depth;
    // ---------------------------------------------------------------- phase A: expand + backup of the simulation in hand
    if (kind != LEAF_NONE) {
      double q = 0.0;                                               // terminal: return 0.
      bool ok = true;
      if (kind == LEAF_NEW) {
        const int idx = nc;
        if (it == 0 && v.ec && claim0 >= 0) {
          // this leaf went to the network and holds a cache entry: the answer goes in, for whoever reaches the state next.  The
          // entry is still this claim's iff meta is unchanged (unique per claim) -- taken exclusively, written, published.
          ECEnt* const ent = v.ec + claim0;
          const float Praw = inrec ? Pans : 0.f;
          const uint32_t fill = (cmeta0 & ~3u) | EC_FILLING, ready = (cmeta0 & ~3u) | EC_READY;
          int got = 0;
          if (lane == 0) got = ec_cas(&ent->meta, cmeta0, fill) ? 1 : 0;
          got = group_bcast0<L>(got, lane);
          const uint32_t px = group_xor<L>(inrec ? ec_term(Praw, lane) : 0u);
          if (got) {
            if (inrec) ec_stf(&ent->P[lane], Praw);
            if (lane == 0) { ec_stf(&ent->V, Vans); ec_st32(&ent->cs, ec_checksum(px, lenv.a, lenv.b, Vans, ready)); }
            __builtin_amdgcn_s_waitcnt(0);                           // every lane's stores have been performed ...
            if (lane == 0) ec_st32(&ent->meta, ready);              // ... before the value that lets readers in
          }
        }
        if (idx >= (v.slot_cap ? v.slot_cap[slot] : v.cap_nodes)) {
          // the slot's pool is exhausted (the reference has no such limit, src/mcts.jl:124-151): self-play retires the slot --
          // its game is reported as aborted and the phase goes on; the hooks report a capacity error
          ok = false;
          if (!p.retire) dev_fail(v, DERR_NODE_POOL);
          else {
            retired = true;
            if (lane == 0) { v.finished[slot] = 2; if (v.fr_active) atomicSub(v.fr_active, 1); }
          }
        } else {
          const uint32_t m = Gm::mask(lenv);
          float Pf = Pans;
          if (p.prior_temp != 1.0) {                                // Util.apply_temperature, util.jl:98-110
            const bool av = (m >> lane) & 1;
            double res;
            if (p.prior_temp == 0.0) {
              int am = group_argmax<L>(av ? (double)Pf : -__builtin_inf(), lane);
              res = (lane == am) ? 1.0 : 0.0;
            } else {
              double pw = av ? az_pow((double)Pf, 1.0 / p.prior_temp) : 0.0;
              double sm = 0.0;
              for (int a = 0; a < Gm::A; ++a) { double t = __shfl(pw, gbase + a); if ((m >> a) & 1) sm += t; }
              res = pw / sm;
            }
            Pf = av ? (float)res : 0.f;
          }
          if (depth == 0) ridx = idx;
          char* nd = node_at<Gm>(v, slot, idx);
          // the edge that reached the new node: entry depth - 1 of the path, in lane depth - 1's register when depth <= L
          const unsigned long long st_par = (depth >= 1 && depth <= L) ? __shfl(st, gbase + depth - 1) : (depth > L ? path[depth - 1] : 0ULL);
          if (lane < Gm::A) {
            *NL::stat(nd, lane) = typename NL::Stat{0.0, 0, Pf};
            ((uint16_t*)(nd + NL::OFF_LO))[lane] = 0;
          }
          if (lane == 0) {
            *(typename NL::hi_t*)(nd + NL::OFF_HI) = 0;
            unsigned long long* kk = side_at(v, slot, idx);
            kk[0] = lenv.a; kk[1] = lenv.b; kk[2] = (unsigned long long)__float_as_uint(Vans);
            const unsigned long long hk = az_hash_key(lenv.a, lenv.b);
            const unsigned long long tag = (hk >> 40) & v.tag_mask;
            v.ht[(size_t)slot * v.ht_size + ins] =
                ((unsigned long long)epoch0 << 48) | (tag << 32) | (unsigned long long)(idx + 1);
            if (depth != 0 && links_ok)                             // memoise tree[state] on the edge that reached it
              set_link<Gm>(node_at<Gm>(v, slot, (int)(uint32_t)st_par), (int)((st_par >> 32) & 0xff), (uint32_t)(idx + 1));
          }
          nc = idx + 1;
          q = (double)Vans;                                         // return info.Vest
        }
      }
      if (ok) {
        // unwinding (mcts.jl:218-223): q_k = r_k + gamma * (pswitch_k ? -q_{k+1} : q_{k+1}); the chain is register
        // arithmetic on shuffled path entries, the read-modify-writes of a chunk of L plies go out together
        for (int c = (depth - 1) / L; c >= 0 && depth > 0; --c) {
          const int k0 = c * L, kn = (depth - k0) < L ? (depth - k0) : L;
          const unsigned long long sc = lane < kn ? (c == 0 ? st : path[k0 + lane]) : 0ULL;
          double qmine = 0.0;
          for (int k = kn - 1; k >= 0; --k) {
            const unsigned long long sk = __shfl(sc, gbase + k);
            const bool psw = (sk >> 40) & 1;
            const double r = (double)((int)((sk >> 41) & 3) - 1);
            q = psw ? -q : q;
            q = r + p.gamma * q;                                    // mcts.jl:219-220
            if (lane == k) qmine = q;
          }
          if (lane < kn) {
            typename NL::Stat* sa = NL::stat(node_at<Gm>(v, slot, (int)(uint32_t)sc), (int)((sc >> 32) & 0xff));
            if (v.bk_mode) {
              // (round 5 experiment, VERDICT r4 #6 ii) the update leaves the dependent chain: one (node, action) record gets exactly
              // one update per slot and simulation, so the sum is the same IEEE add wherever it is performed -- here by L2's atomic unit,
              // nothing comes back.  Phase B must not read the record from this CU's L1 afterwards (see the fence below).
              unsafeAtomicAdd(&sa->W, qmine);
              __hip_atomic_fetch_add(&sa->N, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
              sa->W += qmine;                                       // update_state_info!, mcts.jl:190-194
              sa->N += 1;
            }
          }
        }
        done_trav += depth;                                         // mcts.jl:222
        done_sims += 1;                                             // mcts.jl:242
        msims += v.run_k ? 1 : 0;
      }
      kind = LEAF_NONE;                                             // the simulation in hand is complete
    }
    if (!do_select) break;
    if (dbg && it == 0) dbg[1] = __builtin_readcyclecounter();
    if (v.bk_mode == 0) __threadfence_block();   // phase A's stores (other lanes of the group) are ordered before phase B's loads
    else if (v.bk_mode == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the atomics have been performed at L2; this CU's L1 copies of the path's lines (set_link read them) are dropped
    else { __threadfence_block(); asm volatile("buffer_inv sc1" ::: "memory"); }  // experiment: stores ordered, L1 dropped, the atomics NOT waited for (same address = same L2 channel, in order)
    if (!searching || retired) break;
    if (v.run_k) { if (msims >= p.nsims || it >= v.run_k) break; }  // explore! complete (k_move_fr is due) / enough for one launch
    else if (it >= 1) break;
    if (v.bg_stop && it > 0 && __hip_atomic_load(v.bg_stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - v.bg_seq >= 0) break;   // background: the network launch it ran under is over

    // ---------------------------------------------------------------- phase B: select
    {
      GEnv env = root0;
      int idx = ridx;
      bool probe = idx < 0;
      int pidx = 0, pact = 0;
      depth = 0; ins = 0; ishit = false;
      // what a lane holds of a node: its action's statistics (one 16-byte load), its link's low half, the shared high bits
      typename NL::Stat sa = {0.0, 0, 0.0f};
      uint32_t lo = 0, hi = 0;
      bool have = false;                                            // sa / lo / hi already hold node idx (requested a ply ago)
      for (;;) {
        if (env.fin & 1) { kind = LEAF_TERMINAL; break; }           // mcts.jl:200-201
        if (probe) {                                                // haskey(env.tree, state), mcts.jl:165-174
          idx = ht_lookup<Gm>(v, slot, lane, env.a, env.b, epoch0, &ins);
          if (idx < 0) { kind = LEAF_NEW; break; }                  // mcts.jl:205-207
          if (depth == 0) ridx = idx;
          else if (lane == 0 && links_ok) set_link<Gm>(node_at<Gm>(v, slot, pidx), pact, (uint32_t)(idx + 1));
          have = false;
        }
        if (depth >= v.max_depth) { dev_fail(v, DERR_DEPTH); kind = LEAF_NONE; break; }
        if (!have) {
          const char* nd = node_at<Gm>(v, slot, idx);
          if (inrec) { sa = *NL::stat(nd, lane); lo = ((const uint16_t*)(nd + NL::OFF_LO))[lane]; }
          hi = *(const typename NL::hi_t*)(nd + NL::OFF_HI);
        }
        const uint32_t amask = Gm::mask(env);
        const int N = sa.N;
        const float Pf = sa.P;
        const double W = sa.W;
        const int link = (int)(lo | (((hi >> (2 * lane)) & 3u) << 16));
        if (dbg && it == 0 && depth == 0) dbg[2] = __builtin_readcyclecounter() + (unsigned long long)(N & 0);   // after the root record has arrived
        // uct_scores (mcts.jl:180-188): Float64, evaluated left to right
        const int Ntot = group_sum<L>(N);
        const double sqrtNtot = __builtin_sqrt((double)Ntot);
        const double Q = W / (double)(N > 1 ? N : 1);
        double Pd = (double)Pf;
        if (depth == 0 && p.eps != 0.0) Pd = (1.0 - p.eps) * Pd + p.eps * v.eta[(size_t)slot * L + lane];
        double sc = Q + p.cpuct * Pd * sqrtNtot / (double)(N + 1);
        if (!((amask >> lane) & 1)) sc = -__builtin_inf();
        int nxt = link;                                             // the winner's child link travels with the argmax
        if (dbg && it == 0 && depth == 0) dbg[7] = __builtin_readcyclecounter() + (unsigned long long)(__double2loint(sc) & 0);   // scores of the root ready
        const int act = group_argmax<L>(sc, lane, &nxt);
        if (dbg && it == 0 && depth == 0) dbg[8] = __builtin_readcyclecounter() + (unsigned long long)(act & 0);   // argmax done
        // The child's record is requested NOW, before play! / the terminal test / the path entry: its address needs only the link,
        // and the ~840 cycles of a dependent load cover that work (round 2 issued it a ply later, after all of it).  A link is
        // only ever written for a stored state, so the address is valid even if the child turns out to be unreachable because
        // the game ended (then the data is dropped).
        have = nxt != 0;
        if (have) {
          const char* nd = node_at<Gm>(v, slot, nxt - 1);
          if (inrec) { sa = *NL::stat(nd, lane); lo = ((const uint16_t*)(nd + NL::OFF_LO))[lane]; }
          hi = *(const typename NL::hi_t*)(nd + NL::OFF_HI);
        }
        const bool wp = Gm::white_playing(env);
        Gm::play(env, act);                                         // mcts.jl:213-217
        const float wr = Gm::white_reward(env);
        const int r = (int)(wp ? wr : -wr);
        const bool psw = wp != Gm::white_playing(env);
        const unsigned long long ent = (unsigned long long)(uint32_t)idx | ((unsigned long long)act << 32) |
                                       ((unsigned long long)(psw ? 1 : 0) << 40) | ((unsigned long long)(r + 1) << 41);
        if (lane == 0) path[depth] = ent;                           // the stack of run_simulation!'s recursion: in memory for the next launch ...
        if (lane == depth) st = ent;                                // ... and entry `lane` in a register for this one
        pidx = idx; pact = act;
        if (dbg && it == 0 && depth == 0) dbg[9] = __builtin_readcyclecounter() + (env.a & 0);   // play! / reward / path entry of the first ply done
        depth++;
        probe = nxt == 0;
        idx = nxt - 1;
      }
      if (dbg && it == 0) dbg[3] = __builtin_readcyclecounter();
      lenv = env;
      if (kind != LEAF_NONE) { n_sel += 1; n_trav += depth; }
      if (kind == LEAF_NEW) {
        n_evl += 1;
        if (v.ec) {
          // oracle(state): has the engine evaluated this state before (for another slot, in an earlier wave)?  One look at the entry.
          const uint32_t ci = ec_index(env.a, env.b, v.ec_mask);
          ECEnt* const ent = v.ec + ci;
          const uint32_t m1 = ec_ld32(&ent->meta);
          const unsigned long long ka = ec_ld64(&ent->ka), kb = ec_ld64(&ent->kb);
          const float pl = inrec ? ec_ldf(&ent->P[lane]) : 0.f;
          const float vv = ec_ldf(&ent->V);
          const uint32_t cs = ec_ld32(&ent->cs);
          const uint32_t px = group_xor<L>(inrec ? ec_term(pl, lane) : 0u);
          const bool okc = (m1 & 3u) == EC_READY && (m1 >> 2) > v.ec_floor && ka == env.a && kb == env.b && cs == ec_checksum(px, ka, kb, vv, m1);
          ishit = group_bcast0<L>(okc ? 1 : 0, lane) != 0;          // lane 0's verdict (its checksum covers every lane's prior)
          if (ishit) { Pans = inrec ? pl : 0.f; Vans = vv; n_hit += 1; }   // no network for this leaf
          else {
            // the state goes to the network; its answer goes into the cache if the entry can be taken: empty, answered (the older
            // answer makes room), or a claim so old that its slot has gone away (a phase that ended, a retired slot).  Launches of
            // other slot groups run side by side and may carry a HIGHER number than this one: a claim "from the future" is fresh
            // (signed age).
            claim = -1; cmeta = 0;
            if (lane == 0) {
              const int32_t age = (int32_t)(v.ec_seq - (m1 >> 2));
              if (m1 == 0u || (m1 & 3u) == EC_READY || age > (int32_t)EC_STALE_AFTER || (m1 >> 2) <= v.ec_floor) {
                const uint32_t mine = (v.ec_seq << 2) | EC_PENDING;
                if (ec_cas(&ent->meta, m1, mine)) { ec_st64(&ent->ka, env.a); ec_st64(&ent->kb, env.b); claim = (int)ci; cmeta = mine; }
              }
            }
          }
        }
      }
    }
    if (kind == LEAF_NONE) break;                                   // a device error was raised
    if (kind == LEAF_NEW && !ishit) break;                          // the slot has a question for the network
    if (!v.run_k) break;                                            // lock step: the answer the slot already holds waits for the next launch
    __threadfence_block();                                          // the path entries lane 0 stored are read by the other lanes' phase A
  }

  // ------------------------------------------------------------------ the slot's state for the next launch
  const int slot = slot_in;
  SlotRec* const sr = v.sr + slot;
  if (mine && lane == 0) {
    sr->leaf_kd = kind == LEAF_NONE ? (int)LEAF_NONE : (kind | (depth << 2) | (par << 30));
    if (do_select && kind != LEAF_NONE) { v.leaf_env[slot] = lenv; sr->leaf_ins = ins; }
    if (nc != nc_in) sr->node_count = nc;
    if (ridx != ridx_in) sr->root_idx = ridx;
    if (done_sims) { sr->tot_trav += done_trav; sr->tot_sims += done_sims; }   // mcts.jl:222, :242
    const int aw = retired ? 0 : ((active0 & ((1 << SR_MS_SHIFT) - 1)) | (msims << SR_MS_SHIFT));
    if (aw != active0) sr->active = aw;
  }
  if (!do_select) return;
  if (ishit && kind == LEAF_NEW) {
    // (lock step) the cache's answer waits where the slot's next phase A looks when eidx = -1
    v.Phit[(size_t)slot * L + lane] = Pans;
    if (lane == 0) { v.Vhit[slot] = Vans; sr->eidx = -1; }
  }
  if (dbg) dbg[4] = __builtin_readcyclecounter();

  // ------------------------------------------------------------------ evaluation batch + statistics of the wave
  const bool head = mine && lane == 0;
  const bool isnew = head && kind == LEAF_NEW && !ishit;            // goes to the network
  const unsigned long long bal = __ballot(isnew);
  // (free-running, a wave's launch) still searching, nothing to ask: the background launch carries on with this slot
  // (an explore! that runs ahead, fr = 0: the same counter holds the slots that are not done yet -- a question pending, or simulations to go)
  const bool needy = head && v.run_k && v.bg_list && (v.fr ? (do_backup && searching && !retired && kind == LEAF_NONE && msims < p_arg.nsims)
                                                           : (kind != LEAF_NONE || (searching && !retired && msims < p_arg.nsims)));
  const unsigned long long baln = __ballot(needy);
  int sims = head ? n_sel : 0, trav = head ? n_trav : 0, evl = head ? n_evl : 0, hits = head ? n_hit : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sims += __shfl_down(sims, o); trav += __shfl_down(trav, o); evl += __shfl_down(evl, o); hits += __shfl_down(hits, o); }
  if (wl == 0) { s_need[w] = __popcll(baln); s_new[w] = __popcll(bal); s_hit[w] = hits; s_evl[w] = evl; s_sims[w] = sims; s_trav[w] = trav; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = blockDim.x >> 6;
    int tot = 0, toth = 0, te = 0, ts = 0, tt = 0;
    for (int i = 0; i < nw; ++i) { tot += s_new[i]; toth += s_hit[i]; te += s_evl[i]; ts += s_sims[i]; tt += s_trav[i]; }
    s_base = tot ? atomicAdd(v.n_eval + par, tot) : 0;
    { int tn = 0; for (int i = 0; i < nw; ++i) tn += s_need[i]; s_nbase = tn ? atomicAdd(v.bg_cnt + par, tn) : 0; }
    if (ts) {                                                       // statistics: simulations (mcts.jl:242), traversed nodes (:222), oracle calls
      // per-workgroup accumulators, summed by the host when somebody asks: three same-address atomics per workgroup and wave
      // were what bounded the kernel at 1 M slots (32 768 workgroups)
      long long* sp = v.stat + (size_t)blockIdx.x * 4;
      sp[0] += ts; sp[1] += tt; sp[2] += te; sp[3] += toth;
    }
    if (blockIdx.x == 0 && (do_backup || !v.fr)) {
      // the previous wave's network batch goes to the host (a mapped word it looks at without synchronising: the size of the
      // launches to come decides which tower form serves them), then its counter becomes the next wave's
      if (v.nleaf_host && do_select) __hip_atomic_store(v.nleaf_host, v.n_eval[par ^ 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      v.n_eval[par ^ 1] = 0;
      if (v.bg_cnt) {
        if (v.busy_host && v.run_k && !v.fr && do_backup) __hip_atomic_store(v.busy_host, v.bg_cnt[par ^ 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (v.needy_host && v.fr && do_backup) __hip_atomic_store(v.needy_host, v.bg_cnt[par ^ 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        v.bg_cnt[par ^ 1] = 0;
      }
    }
  }
  __syncthreads();
  if (dbg) dbg[5] = __builtin_readcyclecounter();
  if (isnew) {
    int base = s_base;
    for (int i = 0; i < w; ++i) base += s_new[i];
    const int e = base + __popcll(bal & ((1ULL << wl) - 1ULL));
    sr->eidx = e;
    (v.eval_slots + (size_t)par * v.eval_stride)[e] = slot;
  }
  if (needy && v.fr) {
    int base = s_nbase;
    for (int i = 0; i < w; ++i) base += s_need[i];
    (v.bg_list + (size_t)par * v.eval_stride)[base + __popcll(baln & ((1ULL << wl) - 1ULL))] = slot;
  }
  if (v.ec && head && kind == LEAF_NEW) ((int2*)v.ec_claim)[slot] = make_int2(ishit ? -1 : claim, ishit ? 0 : (int)cmeta);
  if (dbg) dbg[6] = __builtin_readcyclecounter();
}

// rollout! (src/mcts.jl:41-50) from a leaf, value from the leaf's player (mcts.jl:52-60): uniform random
// available actions from the RNG contract's rollout stream of simulation `sim` of (game, move)
template <class Gm>
__host__ __device__ inline float rollout_value(GEnv g, uint64_t seed, uint32_t game_id, uint32_t move, uint32_t sim) {
  const bool wp = Gm::white_playing(g);
  az_rng r = az_rng_make(seed, game_id, move, AZ_RNG_ROLLOUT);
  r.ctr[3] = sim * 1024u;
  double wr = 0.0;
  // rewards are 0 before the end in all three games, so wr + 1.0 * rollout!(...) unwinds to the final reward
  // (gamma = 1: every step of the recursion is 0.0 + 1.0 * x, exact)
  for (int ply = 0; ply < 1024 && !(g.fin & 1); ++ply) {
    const uint32_t m = Gm::mask(g);
    int acts[AZ_MAX_ACTIONS], n = 0;
    for (int a = 0; a < Gm::A; ++a) if ((m >> a) & 1) acts[n++] = a;
    int k = (int)(az_rng_f64(&r) * (double)n);
    if (k >= n) k = n - 1;
    Gm::play(g, acts[k]);
    wr = (double)Gm::white_reward(g);
  }
  return (float)(wp ? wr : -wr);
}

// NN-free oracles: MCTS.RandomOracle (src/mcts.jl:62-72), MCTS.RolloutOracle (:35-60) and the synthetic hash oracle
template <class Gm>
__global__ void __launch_bounds__(256) k_synth_oracle(DView v, DParams p, uint32_t sim_idx, int par) {
  __builtin_amdgcn_s_setprio(3);   // short latency-bound kernel: win issue arbitration against co-resident tower waves
  constexpr int L = Gm::APAD;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= v.n_eval[par]) return;
  const int* const eslots = v.eval_slots + (size_t)par * v.eval_stride;
  const GEnv env = v.leaf_env[eslots[e]];
  const uint32_t m = Gm::mask(env);
  float P[L];
  float V = 0.f;
  if (p.oracle == AZ_ORACLE_UNIFORM || p.oracle == AZ_ORACLE_ROLLOUT) {
    const float u = (float)(1.0 / (double)__popc(m));             // Float32(ones(n) ./ n)
    for (int a = 0; a < L; ++a) P[a] = ((m >> a) & 1) ? u : 0.f;
    if (p.oracle == AZ_ORACLE_ROLLOUT) {
      const int slot = eslots[e];
      V = rollout_value<Gm>(env, p.seed, v.game_id[slot], v.move_idx[slot], sim_idx);
    }
  } else {
    const unsigned long long h = az_hash_key(env.a, env.b);
    float s = 0.f;
    for (int a = 0; a < L; ++a) {
      float raw = (a < Gm::A && ((m >> a) & 1)) ? (float)(1 + (int)(az_mix64(h + (unsigned long long)(a + 1)) & 0xffff)) : 0.f;
      P[a] = raw;
      if (a < Gm::A) s += raw;
    }
    for (int a = 0; a < L; ++a) P[a] = P[a] / s;
    V = (float)((int)(az_mix64(h + 99) & 0xffff) - 32768) / 65536.0f;
  }
  for (int a = 0; a < L; ++a) v.Pout[(size_t)e * L + a] = P[a];
  v.Vout[e] = V;
}

// =========================================================================================
// move: MCTS.policy (mcts.jl:255-271) + the body of play_game's loop (play.jl:308-313) +
// end-of-game bookkeeping of simulate (simulations.jl:231-240)
// =========================================================================================
__host__ __device__ inline double pl_schedule(const DParams& p, int i) {   // schedule.jl:64-80
  int pt = -1;
  for (int k = 0; k < p.temp_len; ++k) if (p.temp_xs[k] <= i) pt = k;
  if (pt < 0) return p.temp_ys[0];
  if (pt == p.temp_len - 1) return p.temp_ys[pt];
  double x0 = p.temp_xs[pt], y0 = p.temp_ys[pt], x1 = p.temp_xs[pt + 1], y1 = p.temp_ys[pt + 1];
  return y0 + (y1 - y0) / (x1 - x0) * ((double)i - x0);
}

// tree[state] of one slot (serial probe, one lane): node record or nullptr
template <class Gm>
__device__ inline const char* find_node(const DView& v, int slot, unsigned long long ka, unsigned long long kb,
                                        uint32_t* idx_out = nullptr) {
  const uint32_t epoch = v.sr[slot].epoch;
  const unsigned long long hk = az_hash_key(ka, kb);
  const uint32_t H1 = (uint32_t)v.ht_size - 1, tag = (uint32_t)(hk >> 40) & v.tag_mask;
  const unsigned long long* tab = v.ht + (size_t)slot * v.ht_size;
  for (uint32_t i = 0; i <= H1; ++i) {
    unsigned long long e = tab[((uint32_t)hk + i) & H1];
    uint32_t idx1 = (uint32_t)e;
    if (!((uint32_t)(e >> 48) == epoch && idx1 != 0)) break;
    if (((uint32_t)(e >> 32) & 0xffff) == tag) {
      const unsigned long long* k = side_at(v, slot, (int)(idx1 - 1));
      if (k[0] == ka && k[1] == kb) { if (idx_out) *idx_out = idx1 - 1; return node_at<Gm>(v, slot, (int)(idx1 - 1)); }
    }
  }
  return nullptr;
}

// MCTS.policy (mcts.jl:255-271) -> apply_temperature (play.jl:309-310) -> fix_probvec + rand_categorical
// (util.jl:68-90) for one root: Nn = visit counts by full action index, m = availability mask, mv = number of
// moves already played (temperature index, play.jl:309).  Shared by k_move and the arena's host loop, so the
// device and the host draw the same action from the same counts.
// apply_temperature + sampling of a policy pi over the n available actions acts[]
__host__ __device__ inline int sample_policy(const DParams& p, const int* acts, const double* pi, int n, uint32_t mv, uint32_t game_id) {
  double pis[AZ_MAX_ACTIONS];
  const double tau = pl_schedule(p, (int)mv);
  if (tau == 1.0) for (int i = 0; i < n; ++i) pis[i] = pi[i];
  else if (tau == 0.0) {
    int am = 0;
    for (int i = 1; i < n; ++i) if (pi[i] > pi[am]) am = i;
    for (int i = 0; i < n; ++i) pis[i] = 0.0;
    pis[am] = 1.0;
  } else {
    const double inv = 1.0 / tau;
    double t = 0.0;
    for (int i = 0; i < n; ++i) { pis[i] = az_pow(pi[i], inv); t += pis[i]; }
    for (int i = 0; i < n; ++i) pis[i] = pis[i] / t;
  }
  float pf[AZ_MAX_ACTIONS];
  float fs = 0.f;
  for (int i = 0; i < n; ++i) { pf[i] = (float)pis[i]; fs += pf[i]; }
  const float tol = 0.00034526698f * (__builtin_fabsf(fs) > 1.0f ? __builtin_fabsf(fs) : 1.0f);
  if (!(__builtin_fabsf(fs - 1.0f) <= tol)) {
    if (fs == 0.0f) for (int i = 0; i < n; ++i) pf[i] = 1.0f / (float)n;
    else for (int i = 0; i < n; ++i) pf[i] = pf[i] / fs;
  }
  az_rng r = az_rng_make(p.seed, game_id, mv, AZ_RNG_MOVE);
  return acts[az_categorical_f32(pf, n, az_rng_f32(&r))];
}
template <class Gm>
__host__ __device__ inline int select_action(const DParams& p, const int* Nn, uint32_t m, uint32_t mv, uint32_t game_id) {
  int acts[AZ_MAX_ACTIONS];
  double pi[AZ_MAX_ACTIONS];
  int n = 0;
  long long ntot = 0;
  for (int a = 0; a < Gm::A; ++a) if ((m >> a) & 1) { acts[n++] = a; ntot += Nn[a]; }
  double s = 0.0;
  for (int i = 0; i < n; ++i) { pi[i] = (double)Nn[acts[i]] / (double)ntot; s += pi[i]; }
  for (int i = 0; i < n; ++i) pi[i] = pi[i] / s;
  return sample_policy(p, acts, pi, n, mv, game_id);
}
// NetworkPlayer (play.jl:226-235) under PlayerWithTemperature: the policy is the network's P over the available actions
template <class Gm>
__host__ __device__ inline int select_action_net(const DParams& p, const float* P, uint32_t m, uint32_t mv, uint32_t game_id) {
  int acts[AZ_MAX_ACTIONS];
  double pi[AZ_MAX_ACTIONS];
  int n = 0;
  for (int a = 0; a < Gm::A; ++a) if ((m >> a) & 1) { pi[n] = (double)P[a]; acts[n++] = a; }
  return sample_policy(p, acts, pi, n, mv, game_id);
}

// play_game's per-turn flip (play.jl:305-307 -> apply_random_symmetry!, game.jl:329-336) for the device's self-play loop: turn
// `mv` of the slot's game starts from `env`.  The turn's trace record gets the state BEFORE the flip (trace.states[i] is pushed
// before the next turn's flip, play.jl:313) and 1 + the symmetry's index in N[AZ_MAX_ACTIONS]; `env` becomes the image the
// player thinks about and plays on.  Same draws as the arena's host loop (az_arena_run) and the oracle.
template <class Gm>
__device__ inline void turn_flip(const DView& v, const DParams& p, int slot, GEnv& env, uint32_t game_id, uint32_t mv) {
  if ((int)mv >= v.max_moves) return;                             // the next k_move retires the slot before it reads the record
  az_move_rec* rec = v.trace + (size_t)slot * v.max_moves + mv;
  rec->key[0] = env.a; rec->key[1] = env.b;
  int k1 = 0;
  if (Gm::NSYM > 0) {
    az_rng r = az_rng_make(p.seed, game_id, mv, AZ_RNG_FLIP);
    if (az_rng_f64(&r) < p.flip_p) {
      int k = (int)(az_rng_f64(&r) * (double)Gm::NSYM);
      if (k >= Gm::NSYM) k = Gm::NSYM - 1;
      env = Gm::sym(env, k);
      k1 = k + 1;
    }
  }
  rec->N[AZ_MAX_ACTIONS] = k1;
}

// GI.init(gspec) on a slot (play.jl:299) + MCTS.reset! when asked: what k_start_games does for a listed slot and what a free-running
// slot does for itself when its game has ended
template <class Gm>
__device__ inline void start_game_on_slot(const DView& v, const DParams& p, int slot, GEnv env, bool have_root, uint32_t game_id, int reset_tree) {
  SlotRec* const sr = v.sr + slot;
  sr->set_root(env);
  sr->root_idx = -1;
  sr->leaf_kd = LEAF_NONE;
  v.game_id[slot] = game_id;
  v.move_idx[slot] = 0;
  sr->active = SR_ACTIVE;
  v.finished[slot] = 0;
  uint32_t ep = sr->epoch;
  if (reset_tree) { ep += 1; sr->node_count = 0; }
  if (ep == 0 || ep >= 0xffff) {                                  // epoch space exhausted: really clear
    unsigned long long* tab = v.ht + (size_t)slot * v.ht_size;
    for (int k = 0; k < v.ht_size; ++k) tab[k] = 0;
    ep = 1; sr->node_count = 0;
  }
  sr->epoch = ep;
  if (!have_root && p.flip_p != 0.0) { turn_flip<Gm>(v, p, slot, env, game_id, 0); sr->set_root(env); }   // self-play: the first turn's flip
}

// The move step of one slot (the body of play_game's loop after think, play.jl:308-313, and simulate's end-of-game bookkeeping,
// simulations.jl:231-240).  FR = false: every slot of the engine, in rounds, the host collects finished games and refills.
// FR = true (free-running phase): the slot whose explore! is complete plays its move as soon as the next k_move_fr comes by; when
// its game ends it writes the game out itself -- game record and move records into the phase buffer, space for both reserved by
// one CAS -- and takes the next game id from the phase's counter: Util.mapreduce's lock (util.jl:181-188) as an atomic.
template <class Gm, bool FR>
__device__ inline void move_slot(const DView& v, const DParams& p, int slot, const FRArgs& fa) {
  using NL = NodeL<Gm>;
  SlotRec* const sr = v.sr + slot;
  const int aw = sr->active;
  if (!(aw & SR_ACTIVE)) return;
  if (FR && ((aw >> SR_MS_SHIFT) < p.nsims || (sr->leaf_kd & 3) != LEAF_NONE)) return;   // still thinking
  GEnv env = sr->root();
  const uint32_t epoch = sr->epoch;
  const char* nd = find_node<Gm>(v, slot, env.a, env.b);          // tree[state] must exist after explore!
  if (!nd) { dev_fail(v, DERR_NO_ROOT); return; }
  const uint32_t m = Gm::mask(env);
  int Nn[AZ_MAX_ACTIONS];
  for (int a = 0; a < Gm::A; ++a) Nn[a] = NL::stat(nd, a)->N;
  const uint32_t mv = v.move_idx[slot];
  const uint32_t gid = v.game_id[slot];
  if ((int)mv >= v.max_moves) {                                   // a game longer than the trace can hold: retired like a full node pool
    if (!p.retire) dev_fail(v, DERR_MOVES);
    else { v.finished[slot] = 2; sr->active = 0; if (FR && v.fr_active) atomicSub(v.fr_active, 1); }
    return;
  }
  const int act = select_action<Gm>(p, Nn, m, mv, gid);
  GEnv nxt = env;
  Gm::play(nxt, act);
  unsigned long long off = 0; int di = 0;
  if (FR && (nxt.fin & 1)) {
    // room for the game's records (a bounded phase always has it; an unbounded one -- bench, polling -- waits here, the slot's
    // move still unplayed, until the host has drained the staging area)
    const unsigned long long nm = (unsigned long long)mv + 1;
    unsigned long long cur = __hip_atomic_load(&fa.st->resv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
      di = (int)(cur >> 40); off = cur & ((1ULL << 40) - 1);
      if (di >= fa.done_cap || (long long)(off + nm) > fa.recs_cap) return;
      if (__hip_atomic_compare_exchange_strong(&fa.st->resv, &cur, cur + (1ULL << 40) + nm, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
  }
  az_move_rec* rec = v.trace + (size_t)slot * v.max_moves + mv;
  if (p.flip_p == 0.0) {
    rec->key[0] = env.a; rec->key[1] = env.b;
    for (int a = 0; a < AZ_MAX_ACTIONS + 1; ++a) rec->N[a] = (a < Gm::A && ((m >> a) & 1)) ? Nn[a] : 0;
  } else {
    // turn_flip has written the turn's un-flipped state and the symmetry.  The reference's trace keeps pi_target as a vector
    // over the AVAILABLE actions of the state the player saw; convert_sample and apply_symmetry spread it over the actions mask of
    // trace.states[i], the un-flipped state (learning.jl:31-33, memory.jl:115-118): the i-th available action of the image lands on the i-th available action of the
    // state.  The record stores the counts that way, so that everything downstream of the trace reads it as before.
    GEnv pre; pre.a = rec->key[0]; pre.b = rec->key[1]; pre.fin = env.fin;
    const uint32_t mp = rec->N[AZ_MAX_ACTIONS] ? Gm::mask(pre) : m;
    int cnt[AZ_MAX_ACTIONS], n = 0;
    for (int a = 0; a < Gm::A; ++a) if ((m >> a) & 1) cnt[n++] = Nn[a];
    n = 0;
    for (int a = 0; a < AZ_MAX_ACTIONS; ++a) rec->N[a] = (a < Gm::A && ((mp >> a) & 1)) ? cnt[n++] : 0;
  }
  env = nxt;
  rec->action = act;
  rec->reward = Gm::white_reward(env);
  sr->set_root(env);
  sr->root_idx = -1;
  v.move_idx[slot] = mv + 1;
  if (FR) { sr->active = SR_ACTIVE; atomicAdd((unsigned long long*)&fa.st->moves, 1ULL); }   // the next explore! starts at simulation 0
  if (env.fin & 1) {
    az_game_rec gr;
    gr.game_id = (int32_t)gid;
    gr.slot = v.slot0 + slot;
    gr.num_moves = (int32_t)(mv + 1);
    gr.first_move = 0;
    gr.nodes = sr->node_count;                                   // measured BEFORE the periodic reset
    gr.total_simulations = sr->tot_sims;
    gr.total_nodes_traversed = sr->tot_trav;
    gr.final_key[0] = env.a; gr.final_key[1] = env.b;
    const int ws = v.worker_sim_id[slot] + 1;
    v.worker_sim_id[slot] = ws;
    const bool reset = p.reset_every > 0 && ws % p.reset_every == 0;   // reset_player!, simulations.jl:235-237
    if (!FR) {
      v.grec[slot] = gr;
      v.finished[slot] = 1;
      sr->active = 0;
      if (reset) { sr->epoch = epoch + 1; sr->node_count = 0; }    // wrap handled by k_start_games
    } else {
      // the trace: the slot's move records, packed behind the games that finished before it
      const uint4* src = (const uint4*)(v.trace + (size_t)slot * v.max_moves);
      uint4* dst = (uint4*)(fa.recs + off);
      for (int k = 0; k < (int)(mv + 1) * (int)(sizeof(az_move_rec) / 16); ++k) dst[k] = src[k];
      gr.first_move = (int32_t)(off > 0x7fffffffULL ? 0x7fffffffULL : off);
      fa.done[di] = gr;
      fa.done_off[di] = (long long)off;
      // the next game id, or no more games: the slot goes idle (util.jl:181-188)
      const int k = atomicAdd(&fa.st->next_game, 1);
      const bool more = fa.total_games < 0 ? (long long)fa.first_game_id + k < (long long)AZ_REPLACEMENT_GAME_BIT : k < fa.total_games;
      if (more) {
        start_game_on_slot<Gm>(v, p, slot, Gm::init(), false, fa.first_game_id + (uint32_t)k, reset ? 1 : 0);
        arm_noise<Gm>(v, p, slot, sr->root());
      } else {
        sr->active = 0;
        if (reset) { sr->epoch = epoch + 1; sr->node_count = 0; }
        if (v.fr_active) atomicSub(v.fr_active, 1);
      }
    }
  } else {
    if (p.flip_p != 0.0) { turn_flip<Gm>(v, p, slot, env, gid, mv + 1); sr->set_root(env); }
    arm_noise<Gm>(v, p, slot, env);                               // next explore! draws its eta
  }
}

template <class Gm>
__global__ void __launch_bounds__(256) k_move(DView v, DParams p) {
  __builtin_amdgcn_s_setprio(3);   // short latency-bound kernel: win issue arbitration against co-resident tower waves
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;       // one lane per slot: n <= 9, serial
  if (slot >= v.G) return;
  move_slot<Gm, false>(v, p, slot, FRArgs{});
}
// free-running phase: once per wave and slot group, behind the group's k_tree; almost every lane leaves at its first test
template <class Gm>
__global__ void __launch_bounds__(256) k_move_fr(DView v, DParams p, FRArgs fa) {
  if (!v.low_prio) __builtin_amdgcn_s_setprio(3);
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x == 0 && fa.host_words && fa.group == 0) {
    // progress for the host, one wave late (it looks at the mapped words without synchronising: is the phase over?)
    const unsigned long long r = __hip_atomic_load(&fa.st->resv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int act = 0;
    for (int g = 0; g < AZ_MAX_GROUPS; ++g) act += __hip_atomic_load(&fa.st->active[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(fa.host_words, (int)(r >> 40), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(fa.host_words + 1, act, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (slot >= v.G) return;
  move_slot<Gm, true>(v, p, slot, fa);
}

// start games on a list of slots (GI.init(gspec), play.jl:299); MCTS.reset! when asked
template <class Gm>
__global__ void __launch_bounds__(256) k_start_games(DView v, DParams p, const int* slots, const uint32_t* game_ids,
                                                     const GEnv* roots, int n, int reset_tree) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  start_game_on_slot<Gm>(v, p, slots[i], roots ? roots[i] : Gm::init(), roots != nullptr, game_ids ? game_ids[i] : 0, reset_tree);
}
// eta for explore!: given by the caller (full action index) or drawn from the RNG contract
template <class Gm>
__global__ void __launch_bounds__(256) k_arm_noise(DView v, DParams p, const int* slots, const uint32_t* moves,
                                                   const double* eta_in, int n) {
  constexpr int L = Gm::APAD;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int slot = slots[i];
  if (moves) v.move_idx[slot] = moves[i];
  if (eta_in) { for (int a = 0; a < L; ++a) v.eta[(size_t)slot * L + a] = a < AZ_MAX_ACTIONS ? eta_in[(size_t)i * AZ_MAX_ACTIONS + a] : 0.0; }
  else arm_noise<Gm>(v, p, slot, v.sr[slot].root());
}

// root visit counts of a list of slots (MCTS.policy's input, mcts.jl:255-271): out[i] = {found, N[0..AZ_MAX_ACTIONS)}
template <class Gm>
__global__ void __launch_bounds__(256) k_root_visits(DView v, const int* slots, const GEnv* roots, int n, int* out) {
  using NL = NodeL<Gm>;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int* o = out + (size_t)i * (AZ_MAX_ACTIONS + 1);
  const char* nd = find_node<Gm>(v, slots[i], roots[i].a, roots[i].b);
  o[0] = nd != nullptr;
  const uint32_t m = Gm::mask(roots[i]);
  for (int a = 0; a < AZ_MAX_ACTIONS; ++a) o[1 + a] = (nd && a < Gm::A && ((m >> a) & 1)) ? NL::stat(nd, a)->N : 0;
}
// MCTS.reset! (mcts.jl:278-281) on a list of slots: a new epoch empties the slot's table in O(1)
static __global__ void __launch_bounds__(256) k_reset_slots(DView v, const int* slots, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int slot = slots[i];
  uint32_t ep = v.sr[slot].epoch + 1;
  if (ep >= 0xffff) {
    unsigned long long* tab = v.ht + (size_t)slot * v.ht_size;
    for (int k = 0; k < v.ht_size; ++k) tab[k] = 0;
    ep = 1;
  }
  v.sr[slot].epoch = ep;
  v.sr[slot].node_count = 0;
  v.sr[slot].root_idx = -1;
}

// host-side maintenance of the slot records (what were hipMemsets of single arrays): any combination of
enum { SR_CLEAR_ACTIVE = 1, SR_CLEAR_LEAF = 2, SR_CLEAR_TOTALS = 4, SR_RESET_TREE = 8, SR_ZERO = 16 };
static __global__ void __launch_bounds__(256) k_slot_records(DView v, int what) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= v.G) return;
  SlotRec* const sr = v.sr + slot;
  if (what & SR_ZERO) { SlotRec z = {}; z.epoch = v.epoch0; z.root_idx = -1; *sr = z; return; }
  if (what & SR_CLEAR_ACTIVE) sr->active = 0;
  if (what & SR_CLEAR_LEAF) sr->leaf_kd = LEAF_NONE;
  if (what & SR_CLEAR_TOTALS) { sr->tot_sims = 0; sr->tot_trav = 0; }
  if (what & SR_RESET_TREE) { sr->node_count = 0; sr->epoch = v.epoch0; sr->root_idx = -1; }   // the caller has zeroed the hash tables
}
// stream-ordered store of one word (the stop signal of a free-running phase's background search)
static __global__ void k_set_word(int* p, int val) { __hip_atomic_store(p, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// node counts of all slots, dense (the host maps pool chunks ahead of the slots, azhip.hip vm_grow)
static __global__ void __launch_bounds__(256) k_node_counts(DView v, int* out) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot < v.G) out[slot] = v.sr[slot].node_count;
}

// AZHIP_TREE_SORT (round 5 experiment, VERDICT r4 #6 i): the order in which k_tree's lane groups take the slots of a slot group.
// A wavefront advances 8 slots and runs until its DEEPEST slot is done; with more wavefronts than the chip holds at once (64 k slots
// and up) a launch costs the sum over wavefronts of their deepest slot, so slots of similar depth belong together.  Key = mean depth
// of the slot's last explore! in half plies ((tot_trav - previous) * 2 / (tot_sims - previous), 0 ... 62; idle slots last): a counting
// sort by one workgroup at the move step, order inside a bin arbitrary -- results are by slot whatever the order.
static __global__ void __launch_bounds__(1024) k_depth_order(DView v, long long* prev /* [G][2] */, int* perm /* [G] */, int* kbuf /* [G] */) {
  __shared__ int bins[64], base[64];
  const int t = threadIdx.x;
  if (t < 64) bins[t] = 0;
  __syncthreads();
  for (int s = t; s < v.G; s += blockDim.x) {
    const SlotRec* r = v.sr + s;
    const long long tt = r->tot_trav, ts = r->tot_sims;
    const long long dt = tt - prev[2 * s], ds = ts - prev[2 * s + 1];
    prev[2 * s] = tt; prev[2 * s + 1] = ts;
    int k = 63;
    if (r->active) { const long long h = ds > 0 ? 2 * dt / ds : 0; k = h < 0 ? 0 : h > 62 ? 62 : (int)h; }
    kbuf[s] = k;
    atomicAdd(&bins[k], 1);
  }
  __syncthreads();
  if (t == 0) { int acc = 0; for (int k = 0; k < 64; ++k) { base[k] = acc; acc += bins[k]; } }
  __syncthreads();
  for (int s = t; s < v.G; s += blockDim.x) perm[atomicAdd(&base[kbuf[s]], 1)] = s;
}

// gather the move records of finished games into one contiguous staging area
static __global__ void k_gather_traces(DView v, const int* slots, const int* offsets, int n, az_move_rec* out) {
  const int i = blockIdx.x;
  if (i >= n) return;
  const int slot = slots[i];
  const int nm = v.grec[slot].num_moves;
  const uint4* src = (const uint4*)(v.trace + (size_t)slot * v.max_moves);
  uint4* dst = (uint4*)(out + offsets[i]);
  for (int k = threadIdx.x; k < nm * 4; k += blockDim.x) dst[k] = src[k];
}

// ---- game plugin kernels (GI.vectorize_state / actions_mask / play!) ------------------------
template <class Gm>
__global__ void k_game_encode(const unsigned long long* keys, int n, float* X, float* A) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const GEnv env = Gm::from_key(keys[2 * i], keys[2 * i + 1]);
  for (int c = 0; c < Gm::C; ++c)
    for (int q = 0; q < Gm::P; ++q) X[(size_t)i * Gm::C * Gm::P + c * Gm::P + q] = Gm::plane(env, q, c);
  const uint32_t m = Gm::mask(env);
  for (int a = 0; a < Gm::A; ++a) A[(size_t)i * Gm::A + a] = (float)((m >> a) & 1);
}
template <class Gm>
__global__ void k_game_play(const unsigned long long* keys, const int* actions, int n,
                            unsigned long long* next, signed char* term, float* reward) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  GEnv env = Gm::from_key(keys[2 * i], keys[2 * i + 1]);
  if (!(env.fin & 1) && actions[i] >= 0) Gm::play(env, actions[i]);
  next[2 * i] = env.a; next[2 * i + 1] = env.b;
  term[i] = (signed char)(env.fin & 1);
  reward[i] = Gm::white_reward(env);
}
