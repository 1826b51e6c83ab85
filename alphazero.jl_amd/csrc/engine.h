// engine.h -- what the translation units of libazhip.so share: the engine record, error / launch macros, the
// profiling helpers and the entry points of net.hip (network launches) used by azhip.hip, memory.hip and train.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <map>
#include <mutex>
#include <vector>
#include <unordered_set>

#include "../../include/az_numerics.h"
#include "../../include/azhip.h"
#include "games.h"
#include "resnet.h"
#include "resnet16.h"
#include "resnet16b.h"
#include "tree.h"

// ------------------------------------------------------------------------------- errors
int fail(int code, const char* fmt, ...);   // sets az_last_error() (thread-local), returns code; defined in azhip.hip
#define HIPCHK(x)                                                                                   \
  do {                                                                                              \
    hipError_t _e = (x);                                                                            \
    if (_e != hipSuccess) return fail(AZ_ERR_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)
#define AZCHK(x)            \
  do {                      \
    int _s = (x);           \
    if (_s != AZ_OK) return _s; \
  } while (0)


#define DISPATCH_GAME(gid, ...)                                   \
  switch (gid) {                                                  \
    case AZ_GAME_CONNECT_FOUR: { using Gm = ConnectFour; __VA_ARGS__; break; } \
    case AZ_GAME_TICTACTOE: { using Gm = TicTacToe; __VA_ARGS__; break; }      \
    case AZ_GAME_MANCALA: { using Gm = Mancala; __VA_ARGS__; break; }          \
    default: return fail(AZ_ERR_BAD_ARG, "unknown game id %d", (int)(gid));    \
  }

struct GameInfo { int A, APAD, W, H, C, P, max_plies, node_bytes; };
inline bool game_info(int gid, GameInfo* gi) {
  switch (gid) {
    case AZ_GAME_CONNECT_FOUR: *gi = {ConnectFour::A, ConnectFour::APAD, ConnectFour::W, ConnectFour::H, ConnectFour::C, ConnectFour::P, ConnectFour::MAX_PLIES, NodeL<ConnectFour>::BYTES}; return true;
    case AZ_GAME_TICTACTOE: *gi = {TicTacToe::A, TicTacToe::APAD, TicTacToe::W, TicTacToe::H, TicTacToe::C, TicTacToe::P, TicTacToe::MAX_PLIES, NodeL<TicTacToe>::BYTES}; return true;
    case AZ_GAME_MANCALA: *gi = {Mancala::A, Mancala::APAD, Mancala::W, Mancala::H, Mancala::C, Mancala::P, Mancala::MAX_PLIES, NodeL<Mancala>::BYTES}; return true;
    case AZ_GAME_GO9_PLANES: *gi = {Go9Planes::A, Go9Planes::APAD, Go9Planes::W, Go9Planes::H, Go9Planes::C, Go9Planes::P, Go9Planes::MAX_PLIES, 64}; return true;   // network-only geometry: no tree
  }
  return false;
}

// ------------------------------------------------------------------------------- engine
// AZ_MAX_GROUPS: tree.h
struct ProfRec { hipEvent_t a = nullptr, b = nullptr; int cls = 0; };
struct az_engine {
  az_engine_cfg cfg;
  GameInfo gi;
  int device;
  hipStream_t stream;
  DView v;                       // global view over all G slots (start / move / hooks)
  DParams p;
  // slot groups: num_workers / batch_size interleaved half-batches, each with its own stream, so the
  // tree kernels of one group run under the network of another (the device form of the reference's
  // num_workers = 2 x batch_size, games/connect-four/params.jl:18-19)
  int ngroups;
  DView gv[AZ_MAX_GROUPS];
  hipStream_t gs[AZ_MAX_GROUPS];      // tree-kernel stream of the group (high priority when ngroups > 1)
  hipStream_t gt[AZ_MAX_GROUPS];      // network stream of the group
  hipEvent_t ev_tree[AZ_MAX_GROUPS], ev_net[AZ_MAX_GROUPS];
  float* g_hfeat[AZ_MAX_GROUPS];
  // k_tower16s (split tower, 128 filters): publish areas per feature buffer (slot g = group g, AZ_MAX_GROUPS = d_hfeat), launch epoch
  unsigned long long* xch[AZ_MAX_GROUPS + 1]; unsigned long long xch_epoch;
  // (r4) a split tower whose exchange gives up degrades instead of failing the phase (azhip.hip recover_split)
  int* d_xerr; int* d_skipped;   // [AZ_MAX_GROUPS + 1] exchange words, [..][2] counters of idle k_tree launches (DView::xerr / skipped)
  int* h_needy; int* d_needy;    // [AZ_MAX_GROUPS] host-mapped: slots each group's previous wave launch left to the background search (tree.h DView::needy_host)
  int* h_nleaf; int* d_nleaf;    // [AZ_MAX_GROUPS] host-mapped: network batch of each group's previous wave (k_tree writes, pick_tower's estimate reads)
  int* h_xflag; int* d_xflag;    // host-mapped word the kernel sets when an exchange gives up; looked at before every launch
  bool split_off;                // the split is disabled for this engine after the first time
  long long xch_fail_at, xch_launches;   // AZHIP_XCH_FAIL_AT = n: the n-th split launch loses a partner (fault injection for the tests)
  int split_registered;          // streams this engine contributes to the device's count of split-tower users
  std::vector<void*> allocs;
  size_t alloc_bytes;              // device bytes behind `allocs` (az_engine_device_bytes)
  // node pool mapped on demand (azhip.hip "node pool"): a virtual range of rows x G chunks of 2 MB; chunk (row r, slot s) is
  // backed when slot s is about to need it.  vm_rows = 0: plain hipMalloc pool.
  char* vm_base; size_t vm_bytes, vm_chunk; int vm_rows, vm_chunk_nodes;
  std::vector<hipMemGenericAllocationHandle_t> vm_handles; std::vector<char*> vm_at;
  std::vector<int> h_slot_cap, h_node_count; int* d_slot_cap; int* d_node_count; size_t vm_budget, vm_mapped;
  // (r5) the side records of a mapped-on-demand pool follow the node chunks: a second virtual range of rows x G pieces of
  // vm_chunk_nodes x 32 B, backed by 2 MB granules (vmk_granule[i] = granule i is mapped); vmk_base = NULL: dense [G][cap][4] array
  char* vmk_base; size_t vmk_bytes, vmk_piece; std::vector<uint8_t> vmk_granule;
  // (r5 experiments) AZHIP_TREE_SORT: slot order of k_tree by the depth of the last explore! (d_perm [G] order | [G] keys,
  // d_perm_prev [G][2] totals at the last move step); AZHIP_TREE_ATOMIC: DView::bk_mode
  int tree_sort; int* d_perm; long long* d_perm_prev;
  // evaluation cache (tree.h ECEnt): one table per engine = per network; az_net_set_params empties it
  ECEnt* d_ec; uint32_t ec_mask; uint32_t ec_seq; int* d_ec_claim; float* d_Phit; float* d_Vhit;
  size_t stat_words; std::vector<long long> h_stat;   // per-workgroup statistics accumulators of k_tree (DView::stat), summed on request
  std::vector<int32_t> aborted_ids;   // games retired because their slot ran out of nodes / move records (az_selfplay_aborted)
  std::vector<void*> net_allocs;   // device copies of the packed network (replaced by az_net_set_params)
  // network
  bool net_loaded;
  std::vector<float> blob;
  NetDev net;
  Net16bDev net16b;              // bf16 fragments (cfg.net_bf16)
  Net16Dev net16;                // k_tower16 fragments (64 filters)
  uint16_t* d_geo[7];            // [6]: the 19-tile paired form (k_tower16x2<.., 10, 9>); [0..5]: Geo16 tables of the 11-tile, 3-tile and 21-tile tower kernels (resnet16.h), [3]: 22 tiles (k_tower16b, 8 boards), [4]: 6 tiles (k_conv16_layer of the trainer at small batches), [5]: the exact-fit variant (NTM<Game> tiles), if the game has one
  int nts;                       // row tiles of the game's latency tower variant (NTS<Game>, resnet16.h)
  int ntm;                       // row tiles of the game's exact-fit variant (NTM<Game>), 0 = none
  long long tower_hist[4];       // network launches served by: the split tower, the 3-row-tile form, the packed 11 / 21-tile forms, others (AZHIP_TRACE_ARENA)
  char last_tower[96];           // name of the tower kernel that served the most recent network launch (az_net_last_kernel)
  int heads_pick;                // AZHIP_HEADS=16|32 forces k_heads16 / k_heads_mfma; 0 = by launch size
  int tower_pick;                // AZHIP_TOWER=16|32|3|21|22|7 forces a tower kernel (3 = k_tower16 with 3 row tiles, 21 = k_tower16x2, 7 = the exact-fit variant NTM); 0 = choose per launch
  int num_cu;
  int nn_cap;
  float* d_hfeat; float* d_X; float* d_A; float* d_P; float* d_V; float* d_Pinv;
  GEnv* d_tmp_env; int* d_iota; int* d_ntmp;
  // pinned staging of the key-based network seam (az_net_evaluate_keys): states + count in, P / V out -- pageable copies were most of a call
  GEnv* h_env; int* h_n; float* h_pv;
  // staging
  int* d_slots; uint32_t* d_gids; GEnv* d_roots; uint32_t* d_moves; double* d_eta; int* d_offsets;
  az_move_rec* d_stage; unsigned long long* d_keys; int* d_actions; unsigned long long* d_next; signed char* d_term; float* d_reward;
  char* d_nodebuf;
  int* d_visits;
  int io_cap;
  // self-play state
  bool running;
  int total_games, next_game, first_game_id, games_done, wave_in_move, active_slots;
  // hipGraphs of wave pairs (one slot group only): key = upper bound of the leaves of a launch (decides the tower kernel and
  // its grid); rebuilt after az_net_set_params (new device pointers)
  std::map<int, hipGraphExec_t> wave_graphs; int use_graphs;
  bool pending[AZ_MAX_GROUPS];       // the group's last wave awaits its expand + backup (flush_pending)
  int wave_par[AZ_MAX_GROUPS];       // parity of the group's last wave: which of its two leaf counters is current
  int group_active[AZ_MAX_GROUPS];   // active slots per slot group (host count; bounds the leaves of a network launch)
  // (r6) free-running phases (az_engine_cfg.lock_step = 0; tree.h k_tree / k_move_fr): the device hands out game ids and writes
  // finished games into the phase buffer itself; the host looks every fr_round_waves waves (fr_round, azhip.hip)
  bool fr_on;                        // the phase in progress is free-running
  int fr_k, fr_kbg, fr_round_waves;  // simulations a slot may select per wave launch / per background launch (DView::run_k); waves between two looks of the host
  hipStream_t fr_s[4]; hipEvent_t fr_ev[4];   // one slot group: the tree / network streams and events of its free-running phases (gs[0] / gt[0] / ev_* point at them meanwhile)
  FRState* d_fr; az_game_rec* d_done; long long* d_done_off; int done_cap;
  int* h_busy; int* d_busy;      // [AZ_MAX_GROUPS] host-mapped: slots of each group still inside an explore! that runs ahead (DView::busy_host)
  int explore_k;                 // simulations per slot and launch inside MCTS.explore! of the hooks and the arena (AZHIP_EXPLORE_K; 0 / 1 = lock step)
  int* d_bg_stop; int bg_seq; bool bg_signal;   // the background search's stop word: set to bg_seq on the wave's stream once its tower has run (bg_signal: this wave has one)
  int* h_fr_words; int* d_fr_words;  // host-mapped: finished games / searching slots as of the previous wave (FRArgs::host_words)
  int fr_prev_done, fr_since_round, fr_given_up; long long fr_prev_recs;
  std::vector<az_game_rec> h_done; std::vector<long long> h_done_off;
  std::vector<az_game_rec> q_games;
  std::vector<az_move_rec> q_moves;
  // device-resident records of the current phase (az_selfplay_run / begin with num_games > 0): the move records of every
  // finished game stay in HBM (d_phase, game after game in finishing order)
  az_move_rec* d_phase; int64_t phase_cap, phase_n; bool host_moves;
  std::vector<az_game_rec> ph_games;
  std::vector<int64_t> ph_off;       // offset of each ph_games entry's first move record in d_phase
  az_selfplay_stats stats;
  std::chrono::steady_clock::time_point t_begin;
  // profiling
  bool prof_on;
  int prof_mask;                 // 0 = every kernel class, else bit per az_kernel_class
  std::vector<ProfRec> prof_pool;
  size_t prof_used;
  az_prof prof;
  double next_exec;              // executed-product fraction of the NEXT profiled launch (set by the tower launches, consumed by prof_begin)
  std::vector<int> h_finished;
  std::vector<az_game_rec> h_grec;
};

template <class T> inline int dalloc(az_engine* e, T** p, size_t n, bool zero = true) {
  void* q = nullptr;
  size_t bytes = std::max<size_t>(n * sizeof(T), 16);
  hipError_t r = hipMalloc(&q, bytes);
  if (r != hipSuccess) return fail(AZ_ERR_HIP, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(r));
  e->allocs.push_back(q);
  e->alloc_bytes += bytes;
  if (zero) HIPCHK(hipMemsetAsync(q, 0, bytes, e->stream));
  *p = (T*)q;
  return AZ_OK;
}

inline int sync_groups(az_engine* e) {
  for (int g = 0; g < e->ngroups; ++g) if (e->gs[g] != e->stream) {
    HIPCHK(hipStreamSynchronize(e->gt[g]));
    HIPCHK(hipStreamSynchronize(e->gs[g]));
  }
  if (e->fr_on && e->fr_s[1]) for (int i = 1; i < 4; ++i) HIPCHK(hipStreamSynchronize(e->fr_s[i]));   // one slot group, free-running: the side stream (move step, background search)
  return AZ_OK;
}
inline int sync_all(az_engine* e) {
  AZCHK(sync_groups(e));
  HIPCHK(hipStreamSynchronize(e->stream));
  return AZ_OK;
}
inline int prof_flush(az_engine* e) {
  if (!e->prof_used) return AZ_OK;
  AZCHK(sync_all(e));
  for (size_t i = 0; i < e->prof_used; ++i) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e->prof_pool[i].a, e->prof_pool[i].b));
    e->prof.ms[e->prof_pool[i].cls] += ms;
  }
  e->prof_used = 0;
  return AZ_OK;
}
inline int prof_begin(az_engine* e, hipStream_t st, int cls, int64_t units) {
  if (!e->prof_on || (e->prof_mask && !((e->prof_mask >> cls) & 1))) return AZ_OK;
  if (e->prof_used == e->prof_pool.size()) AZCHK(prof_flush(e));
  ProfRec& r = e->prof_pool[e->prof_used];
  r.cls = cls;
  e->prof.launches[cls] += 1;
  e->prof.units[cls] += units;
  e->prof.exec_units[cls] += (double)units * e->next_exec;          // the tower launches set next_exec (tower_exec_frac); everything else counts in full
  e->next_exec = 1.0;
  HIPCHK(hipEventRecord(r.a, st));
  return AZ_OK;
}
inline int prof_end(az_engine* e, hipStream_t st, int cls) {
  if (!e->prof_on || (e->prof_mask && !((e->prof_mask >> cls) & 1))) return AZ_OK;
  HIPCHK(hipEventRecord(e->prof_pool[e->prof_used].b, st));
  e->prof_used++;
  return AZ_OK;
}
#define LAUNCH_ON(e, st, cls, units, kern, grid, block, shmem, ...)         \
  do {                                                                      \
    AZCHK(prof_begin(e, st, cls, units));                                   \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), shmem, st, __VA_ARGS__); \
    AZCHK(prof_end(e, st, cls));                                            \
  } while (0)
#define LAUNCH(e, cls, units, kern, grid, block, shmem, ...) LAUNCH_ON(e, (e)->stream, cls, units, kern, grid, block, shmem, __VA_ARGS__)

#define ENGINE(e)                                              \
  if (!(e)) return fail(AZ_ERR_BAD_ARG, "engine is NULL");      \
  HIPCHK(hipSetDevice((e)->device))

// azhip.hip: waits for the engine's streams and turns a device-side error code (tree.h DERR_*, resnet16.h DERR_EXCHANGE) into a status
int check_device_error(az_engine* e);

// azhip.hip: how many streams of this process may launch split towers on `device` at the same time (all engines with 128 filters)
int split_streams_on_device(int device);

// ---- net.hip: every instantiation of the tower / heads kernels lives there ----------------------------------
int net_set_kernel_attrs(az_engine* e);   // + uploads the row permutation tables of the tower kernels (e->d_geo)
// tower + heads on n_max boards (device count in n_ptr when given): from_planes ? X / Amask : envs[eslots[i]]
int net_launch(az_engine* e, hipStream_t st, bool from_planes, float* hfeat, const GEnv* envs, const int* eslots, const int* n_ptr,
               int n_max, const float* X, const float* Amask, float* Pout, float* Vout, float* Pinv, int pstride);
// the network part of one search wave of slot group g (at most nmax leaves)
int net_wave(az_engine* e, int g, bool split, int nmax);

// ---- replay memory records shared by memory.hip and train.hip -------------------------------------------------
struct az_memory {
  int game, device;
  GameInfo gi;
  hipStream_t stream;
  az_sample* d_buf;
  int64_t cap, total, cur_batch;     // total = samples pushed since the last empty!; sample of sequence q sits at q % cap
};
struct az_dataset {
  int game, device;
  GameInfo gi;
  hipStream_t stream;
  int64_t n, sum_n;
  double Wtot;
  float Wmean, Hp;
  az_sample* d_samples;
  GEnv* d_envs;
  float *d_W, *d_X, *d_A, *d_P, *d_V;
  std::vector<void*> allocs;
};
// push_trace! (memory.jl:74-87) of ng games whose move records are ALREADY on the device: game g = d_moves[first[g] .. +cnt[g])
int memory_push_device(az_memory* m, const az_move_rec* d_moves, const std::vector<long long>& first, const std::vector<int>& cnt, double gamma);
template <class T> inline int mem_alloc(std::vector<void*>* keep, T** p, size_t n) {
  void* q = nullptr;
  size_t bytes = std::max<size_t>(n * sizeof(T), 16);
  hipError_t r = hipMalloc(&q, bytes);
  if (r != hipSuccess) return fail(AZ_ERR_HIP, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(r));
  if (keep) keep->push_back(q);
  *p = (T*)q;
  return AZ_OK;
}
