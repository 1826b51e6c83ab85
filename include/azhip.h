/*
 * azhip.h -- C ABI of the MI355X-native self-play engine (libazhip.so).
 *
 * This is the drop-in boundary for ONE path of jonathan-laurent/AlphaZero.jl: the batched
 * MCTS self-play loop plus the ResNet oracle forward.  Each entry point names the
 * reference interface it replaces (paths relative to the reference repository) -- see
 * INTEGRATION.md for the Julia `ccall` glue a maintainer would add.
 *
 * Conventions
 *   - every function returns an int status: AZ_OK (0) or a negative az_status; no C++
 *     exception crosses the boundary; az_last_error() gives a thread-local message that
 *     stays valid until the next call on that thread;
 *   - the caller allocates and frees every host buffer and passes capacities; the engine
 *     owns all device memory and frees it in az_engine_destroy();
 *   - one engine per GPU, not thread-safe, calls block until their results are ready;
 *   - callbacks are invoked synchronously on the calling thread only.
 *
 * State keys.  A game state (the reference's `(board=…, curplayer=…)` named tuple) crosses
 * the ABI as two 64-bit words `key[0], key[1]`; bit 63 of key[0] is set when BLACK is to move.
 *   Connect-Four  key[0] bits col*7+row = WHITE stones, key[1] same for BLACK
 *                 (col 0..6, row 0..5, row 0 = bottom; games/connect-four/game.jl:19,87-93)
 *   Tic-tac-toe   key[0] bits pos = WHITE marks, key[1] bits pos = BLACK marks
 *                 (pos = (y-1)*3 + x - 1; games/tictactoe/game.jl:39)
 *   Mancala       key[0] bytes 0..5 = WHITE houses 1..6, byte 6 = WHITE store;
 *                 key[1] the same for BLACK (games/mancala/game.jl:20-31)
 *
 * fp32 contract of the network (what "the same result" means, bit for bit).  Every output
 * of a convolution or dense layer is ONE fp32 fused-multiply-add chain starting from +0:
 *   3x3 conv (Cin even): taps t = 0..8 with (dy,dx) = (t/3-1, t%3-1); inside a tap the
 *                        channels in the order c = j, Cin/2 + j for j = 0..Cin/2-1 (the K
 *                        order of v_mfma_f32_32x32x2_f32: lanes 0-31 then lanes 32-63)
 *   stem conv          : k = t*Cin + c, K = 9*Cin; order k = j, K2 + j for j = 0..K2-1 with
 *                        K2 = ceil(K/2) (same MFMA, K padded to even with a zero)
 *   1x1 conv           : c = j, Cin/2 + j
 *   dense              : k = p*nf + f ascending (p = x + W*y, f = head filter)
 * followed by y = fma(acc, scale, shift), scale = gamma / sqrtf(var + 1e-5f),
 * shift = fma(bias - mean, scale, beta) (test-mode BatchNorm folded), `+ residual`, ReLU;
 * dense layers add their bias after the chain.  softmax/tanh use az_expf/az_tanhf of
 * az_numerics.h.  The reference (Flux/NNlib/cuDNN) defines no summation order; any order
 * is within the 1e-5 tolerance BASELINE.json states, this one is also reproducible.
 *
 * Environment (diagnostics and tests; the product path sets none of them): AZHIP_TOWER / AZHIP_HEADS force a tower / heads kernel
 * (read at az_engine_create), AZHIP_GRAPH=1 replays wave pairs as hipGraphs, AZHIP_VMM=0|1 forces the plain / mapped-on-demand
 * node pool and AZHIP_POOL_GB bounds the physical memory of the latter, AZHIP_RCCL_LIB=<path> substitutes the library az_comm_*
 * loads (tests/rccl_stub: several ranks on one GPU), AZHIP_TRAIN_ONE_STREAM=1 keeps the trainer's weight gradients on the
 * step's own stream and AZHIP_TRAIN_WG_LATE=1 starts them after the data gradient of their layer instead of beside it (same values
 * either way, read at az_trainer_create).
 *
 * RNG contract: include/az_numerics.h (philox4x32-10 keyed by seed, counter = game id,
 * move index, purpose, draw index).
 */
#ifndef AZHIP_H
#define AZHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AZ_ABI_VERSION 4   /* 4 (round 6): az_engine_cfg.lock_step, az_selfplay_stats.slot_launches; 3 (round 5): az_gather_stats.replaced_games, az_selfplay_stats.evals_reused; 2 (round 4): az_selfplay_stats.aborted_games, az_gather_stats' report fields, az_prof.exec_units, az_comm_version */

typedef enum {
  AZ_OK = 0,
  AZ_ERR_BAD_ARG = -1,   /* invalid argument / unsupported configuration */
  AZ_ERR_CAPACITY = -2,  /* node pool, hash table, path, trace or caller buffer too small */
  AZ_ERR_HIP = -3,       /* a HIP runtime call failed */
  AZ_ERR_STATE = -4,     /* call made in the wrong engine state */
  AZ_ERR_COMM = -5       /* an RCCL call failed / librccl.so could not be loaded */
} az_status;

typedef enum {
  AZ_GAME_CONNECT_FOUR = 0, AZ_GAME_TICTACTOE = 1, AZ_GAME_MANCALA = 2,
  /* Network-only tensor geometry, no device twin: GI.state_dim = (9, 9, 4), 82 actions -- OpenSpiel 9x9 Go through
   * src/openspiel.jl (BASELINE configs[4]).  Rules and tree stay on the host; the engine serves az_net_set_params /
   * az_net_forward (Network.forward_normalized) for it and rejects every search, game and key-based entry point. */
  AZ_GAME_GO9_PLANES = 3
} az_game_id;

/* Which oracle the search consults (src/mcts.jl:6-17). */
typedef enum {
  AZ_ORACLE_UNIFORM = 0, /* MCTS.RandomOracle (src/mcts.jl:62-72): uniform prior, V = 0 */
  AZ_ORACLE_HASH = 1,    /* synthetic, exact: priors/value derived from the state key (tests) */
  AZ_ORACLE_RESNET = 2,  /* the two-headed ResNet (src/networks/architectures/resnet.jl) */
  AZ_ORACLE_ROLLOUT = 3  /* MCTS.RolloutOracle (src/mcts.jl:35-60), gamma = 1 as Benchmark.MctsRollouts builds it
                            (src/benchmark.jl:141-143): uniform prior, V = outcome of one random playout */
} az_oracle_kind;

#define AZ_MAX_ACTIONS 9
#define AZ_SCHED_MAX 8

/* MctsParams (src/params.jl:49-57) + SimParams (src/params.jl:92-101) + ResNetHP
 * (src/networks/architectures/resnet.jl:30-37).  Fill with az_engine_cfg_init() first. */
typedef struct {
  int32_t struct_size;        /* sizeof(az_engine_cfg), set by az_engine_cfg_init */
  int32_t device;             /* HIP device ordinal */
  int32_t game;               /* az_game_id */
  int32_t oracle;             /* az_oracle_kind */
  /* MctsParams */
  double gamma;               /* reward discount */
  double cpuct;
  double dirichlet_noise_eps;
  double dirichlet_noise_alpha;
  double prior_temperature;
  int32_t num_iters_per_turn; /* >= 2; 0 = NetworkPlayer, no search (az_arena_run only, ResNet oracle) */
  int32_t temperature_len;    /* PLSchedule breakpoints; 1 == ConstSchedule */
  int32_t temperature_xs[AZ_SCHED_MAX];
  double temperature_ys[AZ_SCHED_MAX];
  /* SimParams */
  int32_t num_workers;        /* number of device game slots searched in lock-step */
  int32_t batch_size;         /* must be <= num_workers (src/params.jl:361-384); the device path
                                 evaluates every pending leaf of a wave in one pass */
  int32_t reset_every;        /* reset a slot's tree every n games; 0 = never (`nothing`) */
  int32_t fill_batches;       /* accepted, no effect: test-mode BN is per sample (Appendix A.13) */
  double flip_probability;    /* play.jl:305-307; honoured by az_selfplay_* (on the device) and az_arena_run */
  uint64_t seed;
  /* capacities; 0 = derive from the game and num_iters_per_turn */
  int32_t max_nodes_per_slot;
  int32_t max_moves_per_game;
  /* ResNetHP (kernel 3x3) */
  int32_t num_blocks;
  int32_t num_filters;
  int32_t num_policy_head_filters;
  int32_t num_value_head_filters;
  /* 0: fp32 tower, the fp32 contract above (default).  1: the residual tower runs in bfloat16 (weights and activations
   * rounded to bf16, fp32 accumulation, csrc/resnet16b.h) -- BASELINE configs[4] "ResNet 10x128 bf16"; outputs agree with
   * the fp32 network to bf16 accuracy, not bit for bit, so searches are not comparable move for move with the oracle. */
  int32_t net_bf16;
  /* How az_selfplay_* schedules the workers (round 6).
   * 0 (default): FREE-RUNNING.  A worker of the reference runs its simulations and its games at its own pace (simulations.jl:216-243,
   *   util.jl:169-200: the next game id is taken under a lock by whichever worker finishes first).  So does a slot: within one launch
   *   of the tree kernel it completes every simulation that ends on a terminal state or on a state the engine has evaluated before,
   *   and stops only where it needs the network; it plays its move when ITS explore! is complete and takes the next game id from an
   *   atomic counter when ITS game ends.  Every network launch then holds one board of every searching slot.  A game's records depend
   *   on its id and on the games that shared its tree; with reset_every = 1 on the id alone, whatever the schedule.  With
   *   reset_every != 1 the slot -> game assignment (az_game_rec.slot; a slot plays its games in increasing id order) is one outcome
   *   of the reference's own race and can differ from run to run; the parity tests hand the assignment the device reports to the oracle.
   * 1: LOCK STEP, rounds 1-5: one simulation per slot and wave, all slots move together every num_iters_per_turn waves, finished
   *   slots take the next ids in slot order -- a fixed assignment, reproducible for every reset_every.
   * The hooks (az_mcts_explore), the arena, the rollout oracle and hipGraph replay always run in lock step.  AZHIP_FREE_RUN=0|1
   * overrides the field (tests, A/B runs). */
  int32_t lock_step;
} az_engine_cfg;

typedef struct az_engine az_engine;

const char* az_last_error(void);
int az_abi_version(void);
/* sizeof() of a struct of this header as the LIBRARY was compiled (which = az_struct_id), -1 for an unknown id: a host mirror
 * (ctypes, Julia isbits structs) asserts its own sizes against these, so that a stale mirror fails at load time instead of
 * being written out of bounds. */
typedef enum {
  AZ_STRUCT_ENGINE_CFG = 0, AZ_STRUCT_MOVE_REC = 1, AZ_STRUCT_GAME_REC = 2, AZ_STRUCT_TRACE_BUF = 3, AZ_STRUCT_SELFPLAY_STATS = 4,
  AZ_STRUCT_SAMPLE = 5, AZ_STRUCT_DATASET_INFO = 6, AZ_STRUCT_LEARNING_STATUS = 7, AZ_STRUCT_TRAIN_CFG = 8, AZ_STRUCT_GATHER_STATS = 9,
  AZ_STRUCT_PROF = 10
} az_struct_id;
int az_abi_struct_size(int32_t which);

/* Defaults = games/connect-four/params.jl:5-30 with the 64-filter trunk. */
int az_engine_cfg_init(az_engine_cfg* cfg);
int az_engine_create(const az_engine_cfg* cfg, az_engine** out);
int az_engine_destroy(az_engine* e);
/* Device memory the engine holds (node pools, tables, network buffers, the phase buffer): a host that keeps engines alive
 * between phases (the reference rebuilds its MCTS.Env per phase, src/simulations.jl:216-218) budgets with it. */
int az_engine_device_bytes(az_engine* e, int64_t* bytes);
/* Frees the device-resident records of the last phase (up to 64 B x num_games x max_moves) once they have been pushed /
 * gathered; the next az_selfplay_run / az_selfplay_begin allocates again. */
int az_engine_release_phase(az_engine* e);

/* ---- game plugin, device twins (GameInterface, src/game.jl:34-336) -------------------- */
int az_game_num_actions(int game, int32_t* num_actions);
int az_game_state_dim(int game, int32_t* w, int32_t* h, int32_t* c); /* GI.state_dim */
int az_game_init_key(int game, uint64_t key[2]);                   /* current_state(init(gspec)) */
/* GI.vectorize_state + GI.actions_mask(GI.init(gspec, s)) for n states, computed on the GPU.
 * X: n*C*H*W floats (per state the Julia W x H x C array in memory order), A: n*num_actions. */
int az_game_encode(az_engine* e, const uint64_t* keys, int32_t n, float* X, float* A);
/* GI.init(gspec, s); GI.play!(g, a): next state, game_terminated, white_reward, on the GPU.
 * actions are 0-based; action -1 = no move (status of GI.init(gspec, s) only).  States that are
 * already terminal are returned unchanged. */
int az_game_play(az_engine* e, const uint64_t* keys, const int32_t* actions, int32_t n,
                 uint64_t* next_keys, int8_t* terminated, float* white_reward);

/* ---- network plugin (Network interface, src/networks/network.jl:30-206) ---------------- */
/* Number of fp32 values in the parameter blob.  Layout, Flux array memory order:
 *   stem   conv W(3,3,C,F) b(F)  bn gamma,beta,mean,var (F each)
 *   block  x num_blocks: conv1 W(3,3,F,F) b bn(4F)  conv2 W(3,3,F,F) b bn(4F)
 *   phead  conv W(1,1,F,npf) b bn(4npf)  dense W(A, P*npf) b(A)
 *   vhead  conv W(1,1,F,nvf) b bn(4nvf)  dense W(F, P*nvf) b(F)  dense W(1,F) b(1)   */
int az_net_num_params(const az_engine* e, int64_t* n);
int az_net_set_params(az_engine* e, const float* blob, int64_t n); /* Network.copy(nn; on_gpu=true, test_mode=true) */
int az_net_get_params(const az_engine* e, float* blob, int64_t n);
/* Network.forward_normalized (network.jl:264-271) on host arrays: X W*H*C*N, A nA*N ->
 * P nA*N, V N, Pinv N. */
int az_net_forward(az_engine* e, const float* X, const float* A, int32_t N, float* P, float* V, float* Pinv);
/* Network.evaluate_batch (network.jl:308-315) from state keys: encode + forward on the GPU.
 * P is full width (0 on unavailable actions). */
int az_net_evaluate_keys(az_engine* e, const uint64_t* keys, int32_t N, float* P, float* V);

/* ---- MCTS hooks (src/mcts.jl:239-281; parity + explorer UI, src/ui/explorer.jl:70-86) -- */
int az_mcts_reset(az_engine* e);  /* MCTS.reset! on every slot (counters kept) */
/* MCTS.explore!(env, GI.init(gspec, root), nsims) on slots 0..nslots-1, one root per slot.
 * eta: nslots*AZ_MAX_ACTIONS Dirichlet noise by FULL action index, or NULL to draw it from the
 * RNG contract with (seed, game_ids[i], moves[i]). */
int az_mcts_explore(az_engine* e, const uint64_t* root_keys, int32_t nslots, int32_t nsims,
                    const double* eta, const uint32_t* game_ids, const uint32_t* moves);
/* tree[state] of one slot: per FULL action index (unavailable: N = 0, P = 0); returns
 * AZ_ERR_BAD_ARG if the state is not in the tree. mask = availability bitmask. */
int az_mcts_node_stats(az_engine* e, int32_t slot, const uint64_t key[2], int32_t* N, double* W,
                       float* P, float* Vest, uint32_t* mask);
int az_mcts_counters(az_engine* e, int32_t slot, int64_t* total_simulations,
                     int64_t* total_nodes_traversed, int64_t* num_nodes);

/* ---- self-play (simulate, src/simulations.jl:207-244; play_game, src/play.jl:298-315) --- */
/* One record per move: state BEFORE the move, visit counts by full action index (policy =
 * N / sum N, src/mcts.jl:255-271), the action played (0-based), white reward after it.
 * Self-play with flip_probability > 0 (play.jl:305-307): key is trace.states[i], the state BEFORE the turn's random
 * symmetry; N[AZ_MAX_ACTIONS] = 1 + the symmetry's index in GI.symmetries (0 = the turn was not flipped); `action` is
 * the action played on the IMAGE; N[0..A) hold the image's visit counts placed on the AVAILABLE actions of `key` by
 * rank -- what the reference's convert_sample does with the trace's policy vector (learning.jl:31-33, memory.jl:115-118),
 * so az_memory_push* read a flipped trace like any other. */
typedef struct {
  uint64_t key[2];
  int32_t N[AZ_MAX_ACTIONS + 1];
  int32_t action;
  float reward;
} az_move_rec;
/* One record per game (Trace + self_play_measurements, src/training.jl:269-273). */
typedef struct {
  int32_t game_id;
  int32_t slot;
  int32_t num_moves;
  int32_t first_move;               /* index of the game's first az_move_rec */
  int64_t nodes;                    /* length(env.tree) at the end of the game */
  int64_t total_simulations;        /* cumulative per slot (src/mcts.jl:136-137) */
  int64_t total_nodes_traversed;
  uint64_t final_key[2];
} az_game_rec;
typedef struct {
  az_game_rec* games; int64_t games_cap; int64_t num_games;
  az_move_rec* moves; int64_t moves_cap; int64_t num_moves;
} az_trace_buf;
typedef struct {
  int64_t simulations;              /* MCTS simulations run (src/mcts.jl:242) */
  int64_t nodes_traversed;          /* src/mcts.jl:222 */
  int64_t leaf_evals;               /* oracle calls */
  int64_t moves;                    /* samples */
  int64_t games;
  int64_t waves;                    /* launches of the wave sequence (tree kernel, network) per slot group */
  double seconds;
  int64_t aborted_games;            /* games whose slot ran out of tree nodes (max_nodes_per_slot / device memory) or of move
                                     * records (max_moves_per_game): the slot is retired, the game dropped and counted here
                                     * (ids: az_selfplay_aborted), the phase goes on.  A bounded phase still returns num_games
                                     * games: the slot plays ONE replacement game with id = aborted id | AZ_REPLACEMENT_GAME_BIT
                                     * (its own RNG streams); if that overflows as well the game is given up (both ids reported,
                                     * the phase returns one game fewer).  The reference's Dict has no such limit
                                     * (src/mcts.jl:124-151); with the default pool sizes this stays 0.  Hosts should warn when
                                     * it is not (azhip/training.py, julia/AlphaZeroHIP.jl do). */
  int64_t tower_fallbacks;          /* times the split tower of small launches (k_tower16s: two workgroups per board exchanging halves
                                     * of every layer) gave up waiting for a partner that was not co-resident and the engine fell back
                                     * to the unsplit kernel for good: results are unaffected, small launches get slower.  0 unless
                                     * something else (a trainer, another process) holds the device's CUs. */
  int64_t evals_reused;             /* of leaf_evals: oracle answers taken from the engine's evaluation cache -- the same state already
                                     * evaluated for another slot in this wave, or in an earlier wave since az_net_set_params -- instead
                                     * of a network evaluation.  The reference evaluates every query on its own
                                     * (src/simulations.jl:23-38, src/networks/network.jl:308-315); a test-mode evaluation is a pure
                                     * function of the state and every tower form gives the same bits, so the answers -- and with them
                                     * every record of the phase -- are unchanged (tests/test_eval_cache_gpu.py).  leaf_evals -
                                     * evals_reused = boards the network evaluated.  0 when the cache is off (AZHIP_EVAL_CACHE=0). */
  int64_t slot_launches;            /* sum over the waves of the slots that searched in them: simulations / slot_launches = simulations a slot
                                     * completes per launch of the tree kernel (1 in lock step, ~2 free-running with the evaluation cache) */
} az_selfplay_stats;
#define AZ_REPLACEMENT_GAME_BIT 0x40000000   /* game ids handed to az_selfplay_* must stay below it */
typedef void (*az_progress_cb)(void* user);   /* game_simulated(), once per finished game */

/* simulate(simulator, gspec, SimParams(num_games=…)): plays num_games games with global ids
 * first_game_id .. first_game_id+num_games-1 (so a sharded run is independent of the shard
 * count) and writes the traces sorted by game id. */
int az_selfplay_run(az_engine* e, int32_t num_games, int32_t first_game_id, az_trace_buf* out,
                    az_progress_cb cb, void* user, az_selfplay_stats* stats);
/* Stepping form of the same loop (bench, polling): begin, step `nwaves` search waves, collect finished games, end.
 * num_games < 0 = refill slots forever.  Lock step (az_engine_cfg.lock_step): one simulation per active slot per wave, the move
 * step runs after every num_iters_per_turn waves.  Free-running: a wave carries every slot to its next network evaluation; az_selfplay_step
 * returns with the device idle and every game that ended collectable. */
int az_selfplay_begin(az_engine* e, int32_t num_games, int32_t first_game_id);
int az_selfplay_step(az_engine* e, int32_t nwaves);
int az_selfplay_collect(az_engine* e, az_trace_buf* out);   /* finished, not yet collected */
int az_selfplay_get_stats(az_engine* e, az_selfplay_stats* stats);
int az_selfplay_active(az_engine* e, int32_t* active_slots);
/* ids of the games the current / last phase aborted (see az_selfplay_stats.aborted_games); cap = 0 only counts. */
int az_selfplay_aborted(az_engine* e, int32_t* game_ids, int32_t cap, int32_t* n);
int az_selfplay_end(az_engine* e);

/* ---- arena (pit_networks, src/training.jl:130-144; TwoPlayers, src/play.jl:248-282) ------ */
/* simulate() over TwoPlayers(MctsPlayer(contender), MctsPlayer(baseline)): each engine is one
 * player (its own network, MctsParams and per-worker trees); workers = min(contender's
 * num_workers, num_games); reset_every, flip_probability, gamma and the flip RNG seed are the
 * contender's.  alternate_colors != 0 swaps the colours of the games with odd 1-based sim_id
 * (simulations.jl:221-223): sim_id = game id - first_game_id + 1, the index within this call.  rewards[i] (may be NULL) = total reward of
 * game first_game_id+i from the CONTENDER's side (rewards_and_redundancy, simulations.jl:302-311),
 * *redundancy = 1 - #unique states / #states over all traces.  `out` (may be NULL) receives the
 * traces sorted by game id: az_move_rec.key is trace.states[i] (the state BEFORE the turn's random
 * symmetry), N / action refer to the state the player saw (after it), N[AZ_MAX_ACTIONS] = 1 + index
 * of the symmetry applied that turn (0: none); az_game_rec.nodes / total_* are 0.
 * Benchmark.Duel players (src/benchmark.jl:124-192): Full = ResNet oracle, MctsRollouts = AZ_ORACLE_ROLLOUT,
 * NetworkOnly(tau) = an engine with num_iters_per_turn = 0 (NetworkPlayer, src/play.jl:226-235, under
 * PlayerWithTemperature(ConstSchedule(tau))): its moves carry the policy's Float32 bits in N[0..A) and
 * bit 8 of N[AZ_MAX_ACTIONS]. */
int az_arena_run(az_engine* contender, az_engine* baseline, int32_t num_games, int32_t first_game_id,
                 int32_t alternate_colors, az_trace_buf* out, double* rewards, double* redundancy,
                 az_progress_cb cb, void* user);

/* push_trace! (src/memory.jl:74-87): z (discounted, side relative) and t per move record. */
int az_push_trace(const az_move_rec* moves, int32_t n, double gamma, double* z, double* t);

/* Device-only phase: az_selfplay_run with out->moves == NULL keeps the move records in HBM (the engine's phase buffer) and
 * returns the game records only (first_move = -1); az_memory_push_engine / az_comm_gather_push consume them there. */

/* ---- replay memory (src/memory.jl) and learning status (src/learning.jl) on the device ---- */
typedef struct az_memory az_memory;      /* MemoryBuffer (src/memory.jl:34-45): circular buffer of samples in HBM */
typedef struct az_dataset az_dataset;    /* Trainer's converted data (src/learning.jl:98-121), device resident */
/* TrainingSample (src/memory.jl:20-26); pi by FULL action index (0 where the action is unavailable). 112 bytes. */
typedef struct {
  uint64_t key[2];
  double pi[AZ_MAX_ACTIONS];
  double z, t;
  int64_t n;
} az_sample;
typedef enum { AZ_WEIGHT_CONSTANT = 0, AZ_WEIGHT_LOG = 1, AZ_WEIGHT_LINEAR = 2 } az_weighing_policy;   /* params.jl:177 */
int az_memory_create(int32_t game, int32_t device, int64_t capacity, az_memory** out);
int az_memory_destroy(az_memory* m);
/* push_trace!(mem, trace, gamma) (src/memory.jl:74-87) for every game of `traces` in buffer order: one sample
 * per move record, pi = MCTS.policy of the recorded visit counts, z / t as az_push_trace. */
int az_memory_push(az_memory* m, const az_trace_buf* traces, double gamma);
/* self_play_step!'s push loop (src/training.jl:284-299) without the host hop: push_trace!(mem, trace, gamma) for every game
 * of the engine's last bounded self-play phase (az_selfplay_run, or az_selfplay_begin with num_games > 0), in game-id
 * order, reading the move records from the engine's device-resident phase buffer. */
int az_memory_push_engine(az_memory* m, az_engine* e, double gamma);
/* push!(mem.buf, sample) for samples that live on the host (a reference-side MemoryBuffer, a game-stage subset of
 * memory_report, src/learning.jl:192-216); does not advance cur_batch_size. */
int az_memory_push_samples(az_memory* m, const az_sample* samples, int64_t n);
int az_memory_length(az_memory* m, int64_t* length, int64_t* cur_batch_size);   /* length, cur_batch_size (:53-59) */
int az_memory_new_batch(az_memory* m);                                            /* new_batch! (:55) */
int az_memory_empty(az_memory* m);                                                /* empty! (:57-60) */
/* The data of a Trainer: which = 0 get_experience / 1 last_batch (:47-51); use_symmetries =
 * augment_with_symmetries (:116-138, applied by learning_step!, src/training.jl:199-202) ; use_position_averaging =
 * merge_by_state (:98-114; samples of a state are averaged in buffer order, output sorted by key);
 * convert_samples (src/learning.jl:17-51) with the weighing policy.  Everything stays on the device. */
int az_dataset_create(az_memory* m, int32_t which, int32_t use_symmetries, int32_t use_position_averaging,
                      int32_t weighing_policy, az_dataset** out);
int az_dataset_destroy(az_dataset* d);
typedef struct {
  int64_t num_samples;   /* length(samples) after symmetries / merging (= num_boards when merged) */
  int64_t sum_n;         /* sum(e.n), Report.Samples.num_samples (src/learning.jl:186) */
  double Wtot;           /* sum(W) */
  float Wmean, Hp;       /* mean(W), entropy_wmean(P, W) (src/learning.jl:110-111) */
} az_dataset_info;
int az_dataset_get_info(az_dataset* d, az_dataset_info* out);
/* samples / tensors [first, first+count) to host buffers (any pointer may be NULL): W [n], X [n][C][H][W]
 * (= Julia WHCN memory), A [n][nA], P [n][nA], V [n] */
int az_dataset_read(az_dataset* d, int64_t first, int64_t count, az_sample* samples, float* W, float* X, float* A,
                    float* P, float* V);
/* learning_status(tr) (src/learning.jl:158-181): `losses` (:67-90) per batch of loss_computation_batch_size samples
 * (partial last batch kept) with the engine's network in test mode, batches weighted by their total weight. */
typedef struct { float L, Lp, Lv, Lreg, Linv, Hp, Hpnet; } az_learning_status_t;   /* Report.LearningStatus */
int az_learning_status(az_engine* e, az_dataset* d, double l2_regularization, double nonvalidity_penalty,
                       double rewards_renormalization, int64_t loss_computation_batch_size, az_learning_status_t* out);

/* ---- the optimiser step (src/learning.jl:123-141, src/networks/flux.jl:68-95) --------------- */
typedef struct az_trainer az_trainer;   /* Trainer (src/learning.jl:98-121): network in train mode + optimiser state */
typedef enum { AZ_OPT_ADAM = 0, AZ_OPT_CYCLIC_NESTEROV = 1 } az_optimiser;   /* src/networks/network.jl:163-190 */
typedef struct {
  int32_t struct_size;        /* = sizeof(az_train_cfg), set by az_train_cfg_init */
  int32_t optimiser;          /* az_optimiser */
  float lr;                   /* Adam(lr) */
  float lr_base, lr_high, lr_low, momentum_low, momentum_high;   /* CyclicNesterov */
  double l2_regularization, nonvalidity_penalty, rewards_renormalization;   /* LearningParams, src/params.jl:235-248 */
  int32_t batch_size;         /* min(batch_size, #samples) is used (src/learning.jl:113) */
  float batch_norm_momentum;  /* ResNetHP.batch_norm_momentum (resnet.jl:30-37) */
  uint64_t seed;              /* batch shuffling (AZ_RNG_SHUFFLE) */
} az_train_cfg;
int az_train_cfg_init(az_train_cfg* cfg);
/* Trainer(gspec, network, samples, params): the engine supplies the architecture and the initial parameters, the
 * data set (az_dataset_create) the converted samples, Wmean and Hp.  The engine's own network is not modified:
 * fetch the result with az_trainer_get_params and install it with az_net_set_params.  The engine and the data set
 * must outlive the trainer. */
int az_trainer_create(az_engine* e, az_dataset* d, const az_train_cfg* cfg, az_trainer** out);
int az_trainer_destroy(az_trainer* t);
/* batch_updates!(tr, n): n optimiser steps (forward in train mode = BatchNorm with batch statistics, `losses`,
 * backward, Adam / Nesterov with the L2 term, running statistics) on successive shuffled batches (DataLoader
 * partial = false, cycled); losses[i] = L of step i before its update (Network.train! callback).  Like the reference
 * (Flux.setup inside train!), every call starts from a fresh optimiser state; the batch stream continues. */
int az_trainer_batch_updates(az_trainer* t, int32_t n, float* losses);
int az_trainer_get_params(az_trainer* t, float* blob, int64_t n);   /* get_trained_network (src/learning.jl:127-129) */
/* Parity hook: loss and data gradient (Flux parameter order, running-statistics entries 0, L2 term excluded) of the batch
 * made of the given batch_size sample indices; no update, running statistics untouched.
 * parts (may be NULL) = Lp, Lv, Lreg, Linv, mean(W)/Wmean. */
int az_trainer_gradients(az_trainer* t, const int32_t* sample_idx, float* loss, float* parts, float* grad, int64_t n);

/* ---- multi-GPU exchange (simulate_distributed, src/simulations.jl:252-290; one process per GPU) ---------------------- */
/* RCCL over xGMI.  Self-play needs no communication (games are sharded by global game id: az_selfplay_run's
 * first_game_id); afterwards every rank's records are all-gathered device to device and pushed into the replay memory,
 * and new network parameters are broadcast before the next phase.  librccl.so is loaded on first use. */
#define AZ_COMM_ID_BYTES 128
typedef struct az_comm az_comm;
typedef struct {
  int64_t games, moves;             /* over all ranks */
  int64_t bytes;                    /* received per rank by the all-gathers */
  double gather_ms, total_ms;       /* pack + all-gathers + game-record read-back; plus the push into the memory */
  /* Report.SelfPlay's inputs over ALL ranks' games (src/training.jl:293-296): mean over games of average_exploration_depth,
   * the largest tree (nodes; x memory_footprint_per_node = mcts_memory_footprint), and the raw totals */
  int64_t ranks, total_simulations, total_nodes_traversed, max_nodes;
  double mean_game_depth;
  int64_t replaced_games;           /* gathered games whose id carries AZ_REPLACEMENT_GAME_BIT: games some rank aborted and played again.
                                     * Every rank sees the same number, so a host can apply its abort policy AFTER the collective and
                                     * fail on all ranks together (asked games - `games` = games given up for good) */
} az_gather_stats;
/* ncclGetUniqueId on ONE rank; the 128 bytes go to the other ranks by the host's own means (Distributed, MPI, a file). */
int az_comm_unique_id(uint8_t id[AZ_COMM_ID_BYTES]);
/* Which collective library az_comm_* bound: ncclGetVersion's code (e.g. 22105 = 2.21.5; 0 if the library does not say)
 * and the path of the shared object (librccl.so next to the HIP runtime, or AZHIP_RCCL_LIB).  No reference counterpart:
 * evidence for bench.py's `gather` object. */
int az_comm_version(int32_t* version, char* path, int32_t cap);
/* ncclCommInitRank: collective over all `world` ranks, each on its own device. */
int az_comm_init(int32_t device, int32_t rank, int32_t world, const uint8_t id[AZ_COMM_ID_BYTES], az_comm** out);
int az_comm_destroy(az_comm* c);
/* Collective.  A rank that cannot take part (no phase held, wrong device, allocation failure) still enters the first
 * all-gather with its status, and EVERY rank then returns an error together (the failing rank its own, the others
 * AZ_ERR_COMM) -- no rank is left waiting; a collective that fails half way aborts the communicator (later calls
 * return AZ_ERR_COMM).  All-gathers the ranks' device-resident phase records (the `fetch` + `vcat` of simulations.jl:280-289) and,
 * where `m` is not NULL, runs push_trace! (src/memory.jl:74-87) for ALL games in global game-id order into m: every rank
 * that passes a memory ends up with the same samples a single-GPU run of all the games would have pushed. */
int az_comm_gather_push(az_comm* c, az_engine* e, az_memory* m, double gamma, az_gather_stats* stats);
/* Collective (same failure agreement as az_comm_gather_push).  ncclBroadcast of `root`'s parameter blob, then az_net_set_params on every rank (the network shipped to the
 * workers before a phase, src/training.jl:278-282). */
int az_comm_broadcast_params(az_comm* c, az_engine* e, int32_t root);

/* ---- profiling (bench.py roofline): HIP-event time per kernel class ---------------------- */
#define AZ_PROF_NUM 8
typedef enum {
  AZ_K_SELECT = 0, AZ_K_COMPACT = 1, AZ_K_TOWER = 2, AZ_K_HEADS = 3,
  AZ_K_EXPAND = 4, AZ_K_MOVE = 5, AZ_K_SYNTH = 6, AZ_K_START = 7
} az_kernel_class;
typedef struct {
  int64_t launches[AZ_PROF_NUM];
  double ms[AZ_PROF_NUM];
  int64_t units[AZ_PROF_NUM];       /* boards (tower/heads) or slots processed (upper bound known to the host at launch) */
  double exec_units[AZ_PROF_NUM];   /* tower: sum over launches of units x the fraction of the 3x3 convolutions' (row tile, tap)
                                       products the kernel that ran really executes (the others fall off the board for a whole
                                       tile and are skipped); other classes: = units.  exec_units / units = the average executed
                                       fraction of the timed launches */
} az_prof;
int az_prof_enable(az_engine* e, int32_t on);   /* 1: wrap every launch in a HIP event pair; 1 | (mask << 1):
                                                   only the kernel classes whose bit is set in mask; 0: off */
int az_prof_get(az_engine* e, az_prof* out);    /* synchronises, accumulates, returns totals */
int az_prof_reset(az_engine* e);
int az_device_info(az_engine* e, char* name, int32_t name_cap, int32_t* num_cu, int64_t* hbm_bytes);
/* Name of the tower kernel the engine chose for its most recent network launch ("" before the first one): the
 * kernel is picked per launch size (pick_tower, csrc/net.hip), bench.py reports the one that actually ran. */
int az_net_last_kernel(const az_engine* e, char* name, int32_t cap);

#ifdef __cplusplus
}
#endif
#endif /* AZHIP_H */
