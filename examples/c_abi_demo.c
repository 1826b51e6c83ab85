/* c_abi_demo.c -- the boundary used from plain C (no Python, no torch): what a Julia `ccall`, a cgo or a JNI
 * binding does.  Plays a small Tic-tac-toe self-play phase with MCTS.RandomOracle on GPU 0 and prints the traces.
 *
 *   gcc -std=c99 -Iinclude examples/c_abi_demo.c -Lalphazero.jl_amd/csrc -lazhip -Wl,-rpath,$PWD/alphazero.jl_amd/csrc -o c_abi_demo
 *
 * Exit status 0 on success, 2 when no GPU is visible (the library reports AZ_ERR_HIP, it never crashes). */
#include <stdio.h>
#include <stdlib.h>

#include "azhip.h"

static int played = 0;
static void game_simulated(void* user) { (void)user; played++; }

int main(void) {
  az_engine_cfg cfg;
  az_engine* e = NULL;
  if (az_engine_cfg_init(&cfg) != AZ_OK) return 1;
  cfg.game = AZ_GAME_TICTACTOE;
  cfg.oracle = AZ_ORACLE_UNIFORM;
  cfg.num_workers = 4; cfg.batch_size = 4; cfg.num_iters_per_turn = 64; cfg.cpuct = 1.0;
  cfg.dirichlet_noise_eps = 0.25; cfg.reset_every = 1; cfg.seed = 7;
  int st = az_engine_create(&cfg, &e);
  if (st != AZ_OK) {
    fprintf(stderr, "az_engine_create: status %d: %s\n", st, az_last_error());
    return st == AZ_ERR_HIP ? 2 : 1;
  }
  enum { NG = 6 };
  az_game_rec games[NG];
  az_move_rec moves[NG * 9];
  az_trace_buf tb = {games, NG, 0, moves, NG * 9, 0};
  az_selfplay_stats stats;
  st = az_selfplay_run(e, NG, 0, &tb, game_simulated, NULL, &stats);
  if (st != AZ_OK) { fprintf(stderr, "az_selfplay_run: %s\n", az_last_error()); az_engine_destroy(e); return 1; }
  printf("games %lld (callback %d) moves %lld simulations %lld leaf evals %lld\n", (long long)tb.num_games, played,
         (long long)tb.num_moves, (long long)stats.simulations, (long long)stats.leaf_evals);
  for (int g = 0; g < (int)tb.num_games; ++g) {
    printf("game %d:", games[g].game_id);
    for (int k = 0; k < games[g].num_moves; ++k) printf(" %d", moves[games[g].first_move + k].action + 1);
    printf("  white reward %+.0f  nodes %lld\n", moves[games[g].first_move + games[g].num_moves - 1].reward, (long long)games[g].nodes);
  }
  double z[9], t[9];
  az_push_trace(moves + games[0].first_move, games[0].num_moves, 1.0, z, t);
  printf("game 0 targets z:");
  for (int k = 0; k < games[0].num_moves; ++k) printf(" %+.0f", z[k]);
  printf("\n");
  az_engine_destroy(e);
  return (tb.num_games == NG && played == NG) ? 0 : 1;
}
