#include "net_impl.h"
AZ_NET_GAME_TU(Mancala, mancala)
