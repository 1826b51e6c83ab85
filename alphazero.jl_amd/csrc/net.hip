// net.hip -- dispatch of the network launches on the engine's game; the kernels are instantiated per game in
// net_c4.hip / net_ttt.hip / net_mancala.hip (net_impl.h).  azhip.hip / memory.hip call net_launch / net_wave (engine.h).
#include "engine.h"

#define AZ_NET_DECL(sfx)                                                                                                     \
  int net_set_kernel_attrs_##sfx(az_engine* e);                                                                                       \
  int net_launch_##sfx(az_engine* e, hipStream_t st, bool from_planes, float* hfeat, const GEnv* envs, const int* eslots,    \
                       const int* n_ptr, int n_max, const float* X, const float* Amask, float* Pout, float* Vout, float* Pinv, \
                       int pstride);                                                                                         \
  int net_wave_##sfx(az_engine* e, int g, bool split, int nmax);                                                             \
  int net_geometry_##sfx(int which, uint16_t* out, int64_t cap, int* rows, int* products);
AZ_NET_DECL(c4)
AZ_NET_DECL(ttt)
AZ_NET_DECL(mancala)
AZ_NET_DECL(go9)

int net_set_kernel_attrs(az_engine* e) {
  switch (e->cfg.game) {
    case AZ_GAME_CONNECT_FOUR: return net_set_kernel_attrs_c4(e);
    case AZ_GAME_TICTACTOE: return net_set_kernel_attrs_ttt(e);
    case AZ_GAME_MANCALA: return net_set_kernel_attrs_mancala(e);
    case AZ_GAME_GO9_PLANES: return net_set_kernel_attrs_go9(e);
  }
  return fail(AZ_ERR_BAD_ARG, "unknown game id %d", e->cfg.game);
}
int net_launch(az_engine* e, hipStream_t st, bool from_planes, float* hfeat, const GEnv* envs, const int* eslots, const int* n_ptr,
               int n_max, const float* X, const float* Amask, float* Pout, float* Vout, float* Pinv, int pstride) {
  switch (e->cfg.game) {
    case AZ_GAME_CONNECT_FOUR: return net_launch_c4(e, st, from_planes, hfeat, envs, eslots, n_ptr, n_max, X, Amask, Pout, Vout, Pinv, pstride);
    case AZ_GAME_TICTACTOE: return net_launch_ttt(e, st, from_planes, hfeat, envs, eslots, n_ptr, n_max, X, Amask, Pout, Vout, Pinv, pstride);
    case AZ_GAME_MANCALA: return net_launch_mancala(e, st, from_planes, hfeat, envs, eslots, n_ptr, n_max, X, Amask, Pout, Vout, Pinv, pstride);
    case AZ_GAME_GO9_PLANES: return net_launch_go9(e, st, from_planes, hfeat, envs, eslots, n_ptr, n_max, X, Amask, Pout, Vout, Pinv, pstride);
  }
  return fail(AZ_ERR_BAD_ARG, "unknown game id %d", e->cfg.game);
}
int net_wave(az_engine* e, int g, bool split, int nmax) {
  switch (e->cfg.game) {
    case AZ_GAME_CONNECT_FOUR: return net_wave_c4(e, g, split, nmax);
    case AZ_GAME_TICTACTOE: return net_wave_ttt(e, g, split, nmax);
    case AZ_GAME_MANCALA: return net_wave_mancala(e, g, split, nmax);
    case AZ_GAME_GO9_PLANES: return net_wave_go9(e, g, split, nmax);
  }
  return fail(AZ_ERR_BAD_ARG, "unknown game id %d", e->cfg.game);
}

// debug aid (not part of the ABI in azhip.h): the row permutation of a tower kernel (which = 0: 11 tiles, 1: 3 tiles,
// 2: 21 tiles, 3: the 22 tiles of the bf16 8-board form): out = pos [rows] then nbr [9][rows] (Geo16, resnet16.h), *products = (tile, tap) products per convolution
extern "C" int az_debug_tower_geometry(int32_t game, int32_t which, uint16_t* out, int64_t cap, int32_t* rows, int32_t* products) {
  if (!out || !rows || !products || which < 0 || which > 3) return fail(AZ_ERR_BAD_ARG, "bad argument");
  switch (game) {
    case AZ_GAME_CONNECT_FOUR: return net_geometry_c4(which, out, cap, rows, products);
    case AZ_GAME_TICTACTOE: return net_geometry_ttt(which, out, cap, rows, products);
    case AZ_GAME_MANCALA: return net_geometry_mancala(which, out, cap, rows, products);
    case AZ_GAME_GO9_PLANES: return net_geometry_go9(which, out, cap, rows, products);
  }
  return fail(AZ_ERR_BAD_ARG, "unknown game id %d", game);
}
