#include "net_impl.h"
AZ_NET_GAME_TU(ConnectFour, c4)


// debug aid (not part of the ABI in azhip.h): s_memtime stamps of one k_tower16 launch on n boards, 8 words per workgroup
// (Net16Dev::dbg); nt = 11 (4 boards per workgroup) or 3 (one board per workgroup, the small-launch variant)
template <int F, int NT> static int tower_timeline(az_engine* e, int32_t n, unsigned long long* out, int64_t cap) {
  using T = T16<ConnectFour, F, NT>;
  const int nb = (n + T::TB - 1) / T::TB;
  if (cap < (int64_t)nb * 8) return fail(AZ_ERR_CAPACITY, "need %d words", nb * 8);
  unsigned long long* d = nullptr;
  AZCHK(dalloc(e, &d, (size_t)nb * 8));
  HIPCHK(hipMemsetAsync(d, 0, sizeof(unsigned long long) * nb * 8, e->stream));
  std::vector<GEnv> envs(n, ConnectFour::init());
  HIPCHK(hipMemcpyAsync(e->d_tmp_env, envs.data(), sizeof(GEnv) * n, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(e->d_ntmp, &n, sizeof(int), hipMemcpyHostToDevice, e->stream));
  Net16Dev nd = e->net16;
  for (int rep = 0; rep < 2; ++rep) {
    nd.dbg = rep ? d : nullptr;
    hipLaunchKernelGGL((k_tower16<ConnectFour, F, false, NT>), dim3(nb), dim3(T::THREADS), T::BYTES, e->stream, nd, e->d_tmp_env, e->d_iota, e->d_ntmp, n, (const float*)nullptr, e->d_hfeat);
  }
  HIPCHK(hipMemcpyAsync(out, d, sizeof(unsigned long long) * nb * 8, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  e->allocs.pop_back();
  (void)hipFree(d);
  return AZ_OK;
}
static int tower_timeline_split(az_engine* e, int32_t n, unsigned long long* out, int64_t cap) {
  using T = T16S<ConnectFour, 128>;
  const int nb = 2 * ((n + T::TB - 1) / T::TB);
  if (nb > e->num_cu) return fail(AZ_ERR_BAD_ARG, "split tower: at most %d boards", e->num_cu / 2 * T::TB);
  if (cap < (int64_t)nb * 8) return fail(AZ_ERR_CAPACITY, "need %d words", nb * 8);
  unsigned long long* d = nullptr;
  AZCHK(dalloc(e, &d, (size_t)nb * 8));
  std::vector<GEnv> envs(n, ConnectFour::init());
  HIPCHK(hipMemcpyAsync(e->d_tmp_env, envs.data(), sizeof(GEnv) * n, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(e->d_ntmp, &n, sizeof(int), hipMemcpyHostToDevice, e->stream));
  unsigned long long* xa; unsigned long long ep;
  Net16Dev nd = e->net16;
  for (int rep = 0; rep < 2; ++rep) {
    nd.dbg = rep ? d : nullptr;
    AZCHK(xch_slot<ConnectFour>(e, e->d_hfeat, &xa, &ep));
    hipLaunchKernelGGL((k_tower16s<ConnectFour, 128, false>), dim3(nb), dim3(T::THREADS), T::BYTES, e->stream, nd, e->d_tmp_env, e->d_iota, e->d_ntmp, n, (const float*)nullptr, e->d_hfeat,
                       xa, ep, e->v.err, e->d_xflag);
  }
  HIPCHK(hipMemcpyAsync(out, d, sizeof(unsigned long long) * nb * 8, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  e->allocs.pop_back();
  (void)hipFree(d);
  return AZ_OK;
}
template <int F> static int tower_timeline_bf16(az_engine* e, int32_t n, unsigned long long* out, int64_t cap) {
  using T = T16B<ConnectFour, F, 11>;
  const int nb = (n + T::TB - 1) / T::TB;
  if (cap < (int64_t)nb * 8) return fail(AZ_ERR_CAPACITY, "need %d words", nb * 8);
  unsigned long long* d = nullptr;
  AZCHK(dalloc(e, &d, (size_t)nb * 8));
  std::vector<GEnv> envs(n, ConnectFour::init());
  HIPCHK(hipMemcpyAsync(e->d_tmp_env, envs.data(), sizeof(GEnv) * n, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(e->d_ntmp, &n, sizeof(int), hipMemcpyHostToDevice, e->stream));
  Net16bDev nd = e->net16b;
  for (int rep = 0; rep < 2; ++rep) {
    nd.dbg = rep ? d : nullptr;
    hipLaunchKernelGGL((k_tower16b<ConnectFour, F, false, 11>), dim3(nb), dim3(T::THREADS), T::BYTES, e->stream, nd, e->d_tmp_env, e->d_iota, e->d_ntmp, n, (const float*)nullptr, e->d_hfeat);
  }
  HIPCHK(hipMemcpyAsync(out, d, sizeof(unsigned long long) * nb * 8, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  e->allocs.pop_back();
  (void)hipFree(d);
  return AZ_OK;
}
// debug aid: cycle stamps of one split k_heads16 launch on n boards (after a tower launch that fills the head features):
// workgroup 2 t = value tiles of board tile t, 2 t + 1 = its policy tiles; per wavefront 4 words: start, chain done,
// workgroup barrier passed, end
template <int F> static int heads_timeline(az_engine* e, int32_t n, unsigned long long* out, int64_t cap) {
  using H = H16<ConnectFour, F, true>;
  const int nb = 2 * ((n + 15) / 16);
  if (cap < (int64_t)nb * H::WAVES * 4) return fail(AZ_ERR_CAPACITY, "need %d words", nb * H::WAVES * 4);
  unsigned long long* d = nullptr;
  AZCHK(dalloc(e, &d, (size_t)nb * H::WAVES * 4));
  std::vector<GEnv> envs(n, ConnectFour::init());
  HIPCHK(hipMemcpyAsync(e->d_tmp_env, envs.data(), sizeof(GEnv) * n, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(e->d_ntmp, &n, sizeof(int), hipMemcpyHostToDevice, e->stream));
  AZCHK(net_launch(e, e->stream, false, e->d_hfeat, e->d_tmp_env, e->d_iota, e->d_ntmp, n, nullptr, nullptr, e->d_P, e->d_V, nullptr, ConnectFour::A));
  NetDev nd = e->net;
  nd.dbg = d;
  hipLaunchKernelGGL((k_heads16<ConnectFour, F, true>), dim3(nb), dim3(H::THREADS), H::BYTES, e->stream, nd, e->d_tmp_env, e->d_iota, e->d_ntmp, n, (const float*)nullptr, e->d_hfeat, e->d_P, e->d_V, (float*)nullptr, ConnectFour::A);
  HIPCHK(hipMemcpyAsync(out, d, sizeof(unsigned long long) * nb * H::WAVES * 4, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  e->allocs.pop_back();
  (void)hipFree(d);
  return AZ_OK;
}
// debug aid: one k_tower16s launch in which workgroup 1 leaves at once, so that workgroup 0 never gets its partner's half:
// the bounded poll must give up (DERR_EXCHANGE -> AZ_ERR_HIP from this call) instead of hanging the GPU
extern "C" int az_debug_exchange_timeout(az_engine* e) {
  ENGINE(e);
  if (!e->net_loaded || e->cfg.game != AZ_GAME_CONNECT_FOUR || e->cfg.num_filters != 128 || e->cfg.net_bf16) return fail(AZ_ERR_BAD_ARG, "connect-four, 128 filters, fp32");
  using T = T16S<ConnectFour, 128>;
  const int n = 2;
  std::vector<GEnv> envs(n, ConnectFour::init());
  HIPCHK(hipMemcpyAsync(e->d_tmp_env, envs.data(), sizeof(GEnv) * n, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(e->d_ntmp, &n, sizeof(int), hipMemcpyHostToDevice, e->stream));
  unsigned long long* xa; unsigned long long ep;
  AZCHK(xch_slot<ConnectFour>(e, e->d_hfeat, &xa, &ep));
  hipLaunchKernelGGL((k_tower16s<ConnectFour, 128, false>), dim3(2 * n), dim3(T::THREADS), T::BYTES, e->stream, e->net16, e->d_tmp_env, e->d_iota, e->d_ntmp, n, (const float*)nullptr, e->d_hfeat,
                     xa, ep | (1ull << 63), e->v.err, e->d_xflag);
  return check_device_error(e);
}
// debug aid: out == NULL -> from now on the first wavefront of every k_tree launch of group 0 leaves 8 cycle stamps
// (DView::dbg); out != NULL -> copy the stamps of the most recent launch
extern "C" int az_debug_tree_stamps(az_engine* e, unsigned long long* out) {
  ENGINE(e);
  if (!out) {
    unsigned long long* d = nullptr;
    AZCHK(dalloc(e, &d, 16));
    AZCHK(sync_all(e));
    e->v.dbg = d; e->gv[0].dbg = d;
    return AZ_OK;
  }
  if (!e->gv[0].dbg) return fail(AZ_ERR_STATE, "stamps are not enabled");
  AZCHK(sync_all(e));
  HIPCHK(hipMemcpy(out, e->gv[0].dbg, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return AZ_OK;
}
extern "C" int az_debug_heads_timeline(az_engine* e, int32_t n, unsigned long long* out, int64_t cap) {
  ENGINE(e);
  if (!e->net_loaded || n < 1 || n > e->nn_cap || e->cfg.game != AZ_GAME_CONNECT_FOUR || !e->net.hd16_ok) return fail(AZ_ERR_BAD_ARG, "connect-four with 32 head filters");
  return e->cfg.num_filters == 128 ? heads_timeline<128>(e, n, out, cap) : heads_timeline<64>(e, n, out, cap);
}
extern "C" int az_debug_tower_timeline(az_engine* e, int32_t n, int32_t nt, unsigned long long* out, int64_t cap) {
  ENGINE(e);
  if (!e->net_loaded || n < 1 || n > e->nn_cap) return fail(AZ_ERR_BAD_ARG, "bad n / no net");
  if (e->cfg.game == AZ_GAME_CONNECT_FOUR && e->cfg.net_bf16 && nt == 11) return e->cfg.num_filters == 128 ? tower_timeline_bf16<128>(e, n, out, cap) : tower_timeline_bf16<64>(e, n, out, cap);
  if (e->cfg.game != AZ_GAME_CONNECT_FOUR || e->cfg.net_bf16 || (nt != 11 && nt != 3 && nt != 2)) return fail(AZ_ERR_BAD_ARG, "connect-four; fp32: nt = 11, 3 or 2 (= split tower), bf16: nt = 11");
  if (nt == 2) return e->cfg.num_filters == 128 ? tower_timeline_split(e, n, out, cap) : fail(AZ_ERR_BAD_ARG, "split tower: 128 filters");
  if (e->cfg.num_filters == 64) return nt == 11 ? tower_timeline<64, 11>(e, n, out, cap) : tower_timeline<64, 3>(e, n, out, cap);
  return nt == 11 ? tower_timeline<128, 11>(e, n, out, cap) : tower_timeline<128, 3>(e, n, out, cap);
}
