#include "net_impl.h"
AZ_NET_GAME_TU(ConnectFour, c4)


// debug aid (not part of the ABI in azhip.h): s_memtime stamps of one k_tower16 launch (4 boards per workgroup, 8 words each)
// on n boards: start, after the stem, after the residual tower, after the head convolution (features written)
extern "C" int az_debug_tower_timeline(az_engine* e, int32_t n, unsigned long long* out, int64_t cap) {
  ENGINE(e);
  if (!e->net_loaded || n < 1 || n > e->nn_cap) return fail(AZ_ERR_BAD_ARG, "bad n / no net");
  if (e->cfg.game != AZ_GAME_CONNECT_FOUR || e->cfg.num_filters != 64) return fail(AZ_ERR_BAD_ARG, "connect-four with 64 filters only");
  using T = T16<ConnectFour, 64>;
  const int nb = (n + T::TB - 1) / T::TB;
  if (cap < (int64_t)nb * 8) return fail(AZ_ERR_CAPACITY, "need %d words", nb * 8);
  unsigned long long* d = nullptr;
  AZCHK(dalloc(e, &d, (size_t)nb * 8));
  std::vector<GEnv> envs(n, ConnectFour::init());
  HIPCHK(hipMemcpyAsync(e->d_tmp_env, envs.data(), sizeof(GEnv) * n, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(e->d_ntmp, &n, sizeof(int), hipMemcpyHostToDevice, e->stream));
  Net16Dev nd = e->net16;
  for (int rep = 0; rep < 2; ++rep) {
    nd.dbg = rep ? d : nullptr;
    hipLaunchKernelGGL((k_tower16<ConnectFour, 64, false>), dim3(nb), dim3(T::THREADS), T::BYTES, e->stream, nd, e->d_tmp_env, e->d_iota, e->d_ntmp, n, (const float*)nullptr, e->d_hfeat);
  }
  HIPCHK(hipMemcpyAsync(out, d, sizeof(unsigned long long) * nb * 8, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  e->allocs.pop_back();
  (void)hipFree(d);
  return AZ_OK;
}
