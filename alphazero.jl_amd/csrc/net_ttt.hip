#include "net_impl.h"
AZ_NET_GAME_TU(TicTacToe, ttt)
