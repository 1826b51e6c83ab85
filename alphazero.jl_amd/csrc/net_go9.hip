#include "net_impl.h"
AZ_NET_GAME_TU(Go9Planes, go9)
