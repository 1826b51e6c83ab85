// net_impl.h -- launch logic of the network kernels (resnet.h, resnet16.h), templated on the game: which tower kernel
// serves a launch (pick_tower) and the launches themselves.  Instantiated once per game in net_c4.hip / net_ttt.hip /
// net_mancala.hip (three translation units, so that the big fully unrolled tower kernels compile in parallel);
// net.hip dispatches on the engine's game.
#pragma once
#include "engine.h"

template <class Gm, int F, bool PLANES_ONLY = false> static int set_kernel_attrs_f() {
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tower16<Gm, F, false>), hipFuncAttributeMaxDynamicSharedMemorySize, T16<Gm, F>::BYTES));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tower16<Gm, F, true>), hipFuncAttributeMaxDynamicSharedMemorySize, T16<Gm, F>::BYTES));
  if constexpr (NTM<Gm> > 0) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tower16<Gm, F, false, NTM<Gm>>), hipFuncAttributeMaxDynamicSharedMemorySize, T16<Gm, F, NTM<Gm>>::BYTES));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tower16<Gm, F, true, NTM<Gm>>), hipFuncAttributeMaxDynamicSharedMemorySize, T16<Gm, F, NTM<Gm>>::BYTES));
  }
  if constexpr (F == 64) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tower16x2<Gm, F, false>), hipFuncAttributeMaxDynamicSharedMemorySize, T16P<Gm, F>::BYTES));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tower16x2<Gm, F, true>), hipFuncAttributeMaxDynamicSharedMemorySize, T16P<Gm, F>::BYTES));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tower16x2<Gm, F, false, 10, 9>), hipFuncAttributeMaxDynamicSharedMemorySize, T16P<Gm, F, 10, 9>::BYTES));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tower16x2<Gm, F, true, 10, 9>), hipFuncAttributeMaxDynamicSharedMemorySize, T16P<Gm, F, 10, 9>::BYTES));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tower16x2m<Gm, F, false>), hipFuncAttributeMaxDynamicSharedMemorySize, T16P<Gm, F>::BYTES));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tower16x2c<Gm, F, false>), hipFuncAttributeMaxDynamicSharedMemorySize, T16P<Gm, F>::BYTES));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tower16x2c<Gm, F, true>), hipFuncAttributeMaxDynamicSharedMemorySize, T16P<Gm, F>::BYTES));
  }
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tower16b<Gm, F, false, 11>), hipFuncAttributeMaxDynamicSharedMemorySize, T16B<Gm, F, 11>::BYTES));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tower16b<Gm, F, true, 11>), hipFuncAttributeMaxDynamicSharedMemorySize, T16B<Gm, F, 11>::BYTES));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tower16b<Gm, F, false, NTS<Gm>>), hipFuncAttributeMaxDynamicSharedMemorySize, T16B<Gm, F, NTS<Gm>>::BYTES));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tower16b<Gm, F, true, NTS<Gm>>), hipFuncAttributeMaxDynamicSharedMemorySize, T16B<Gm, F, NTS<Gm>>::BYTES));
  if constexpr (F == 128) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tower16b<Gm, F, false, 22>), hipFuncAttributeMaxDynamicSharedMemorySize, T16B<Gm, F, 22>::BYTES));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tower16b<Gm, F, true, 22>), hipFuncAttributeMaxDynamicSharedMemorySize, T16B<Gm, F, 22>::BYTES));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tower16s<Gm, F, false>), hipFuncAttributeMaxDynamicSharedMemorySize, T16S<Gm, F>::BYTES));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tower16s<Gm, F, true>), hipFuncAttributeMaxDynamicSharedMemorySize, T16S<Gm, F>::BYTES));
  }
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_heads16<Gm, F, true>), hipFuncAttributeMaxDynamicSharedMemorySize, H16<Gm, F, true>::BYTES));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_heads16<Gm, F, false>), hipFuncAttributeMaxDynamicSharedMemorySize, H16<Gm, F, false>::BYTES));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tower<Gm, F, false>), hipFuncAttributeMaxDynamicSharedMemorySize, TowerLds<F>::BYTES));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tower<Gm, F, true>), hipFuncAttributeMaxDynamicSharedMemorySize, TowerLds<F>::BYTES));
  return AZ_OK;
}
// (Re-arranging the cells inside Geo16's class constraint so that the tap-shifted rows of every ds_read_b128 pass are
// distinct mod 8 -- a host-side descent took the excess lanes of the Connect-Four layout from 186 to 68 over 170 passes --
// changed no kernel time: 844 vs 845 us per 4096 boards, bf16 10x128 4.88 vs 4.83 M sims/s.  The A reads are not what a
// tower waits for; the tables stay Geo16's own.)
// the row permutation tables of the three k_tower16 geometries (Geo16, resnet16.h) go to the device once per engine
template <class G> static int upload_geo(az_engine* e, int which) {
  std::vector<uint16_t> h((size_t)10 * G::RPAD);
  for (int i = 0; i < G::RPAD; ++i) h[i] = G::tab.pos[i];
  for (int i = 0; i < 9 * G::RPAD; ++i) h[G::RPAD + i] = G::tab.nbr[i];
  uint16_t* d = nullptr;
  AZCHK(dalloc(e, &d, h.size(), false));
  HIPCHK(hipMemcpyAsync(d, h.data(), h.size() * sizeof(uint16_t), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  e->d_geo[which] = d;
  return AZ_OK;
}
template <class G> static int geometry_out(uint16_t* out, int64_t cap, int* rows, int* products) {
  if (cap < (int64_t)10 * G::RPAD) return fail(AZ_ERR_CAPACITY, "need %d words", 10 * G::RPAD);
  for (int i = 0; i < G::RPAD; ++i) out[i] = G::tab.pos[i];
  for (int i = 0; i < 9 * G::RPAD; ++i) out[G::RPAD + i] = G::tab.nbr[i];
  *rows = G::RPAD; *products = G::tab.cost;
  return AZ_OK;
}
template <class Gm> static int set_kernel_attrs(az_engine* e) {
  e->nts = NTS<Gm>;
  AZCHK((set_kernel_attrs_f<Gm, 64>()));
  AZCHK((set_kernel_attrs_f<Gm, 128>()));
  AZCHK((upload_geo<typename T16<Gm, 64, 11>::Geo>(e, 0)));
  AZCHK((upload_geo<typename T16<Gm, 64, NTS<Gm>>::Geo>(e, 1)));
  AZCHK((upload_geo<typename T16P<Gm, 64>::Geo>(e, 2)));
  AZCHK((upload_geo<typename T16P<Gm, 64, 10, 9>::Geo>(e, 6)));
  AZCHK((upload_geo<typename T16B<Gm, 128, 22>::Geo>(e, 3)));
  AZCHK((upload_geo<typename T16<Gm, 64, 6>::Geo>(e, 4)));
  e->d_geo[5] = nullptr;
  if constexpr (NTM<Gm> > 0) AZCHK((upload_geo<typename T16<Gm, 64, NTM<Gm>>::Geo>(e, 5)));
  e->ntm = NTM<Gm>;
  return AZ_OK;
}

static void note_tower(az_engine* e, int tw, int F) {
  static const char* const gn[] = {"ConnectFour", "TicTacToe", "Mancala", "Go9Planes"};
  const char* g = gn[e->cfg.game];
  e->tower_hist[tw == 2 ? 0 : tw == 3 ? 1 : (tw == 16 || tw == 21 || tw == 19 || tw == 20) ? 2 : 3]++;
  if (e->cfg.net_bf16) { snprintf(e->last_tower, sizeof e->last_tower, "k_tower16b<%s,%d,NT=%d>", g, F, tw == 3 ? e->nts : tw == 22 ? 22 : 11); return; }
  if (tw == 2) snprintf(e->last_tower, sizeof e->last_tower, "k_tower16s<%s,%d>", g, F);
  else if (tw == 21) snprintf(e->last_tower, sizeof e->last_tower, "k_tower16x2<%s,%d>", g, F);
  else if (tw == 19) snprintf(e->last_tower, sizeof e->last_tower, "k_tower16x2<%s,%d,NT=19>", g, F);
  else if (tw == 20) snprintf(e->last_tower, sizeof e->last_tower, "k_tower16x2c<%s,%d>", g, F);
  else if (tw == 3) snprintf(e->last_tower, sizeof e->last_tower, "k_tower16<%s,%d,NT=%d>", g, F, e->nts);
  else if (tw == 7) snprintf(e->last_tower, sizeof e->last_tower, "k_tower16<%s,%d,NT=%d>", g, F, e->ntm);
  else if (tw == 16) snprintf(e->last_tower, sizeof e->last_tower, "k_tower16<%s,%d,NT=11>", g, F);
  else snprintf(e->last_tower, sizeof e->last_tower, "k_tower<%s,%d>", g, F);
}

// launches tower + heads on `n` boards (device count in n_ptr when n < 0)
// Which tower kernel serves a launch of up to n boards.  A workgroup's layer chain is sequential and the MFMA
// pipe of a CU is shared by its resident workgroups, so the launch costs (workgroups per CU, rounded up) x (rows
// per workgroup) x (fraction of the tile-tap products it executes): k_tower16 packs 176 rows (4 Connect-Four boards),
// k_tower 128 (3 boards, every tap), k_tower16 with 3 row tiles 48 (1 board; +10 %: a third of the weight reuse, more
// barriers per row).  Measured round 2 (tools/run_config.py): Mancala 8192 slots 13.5 M sims/s on k_tower, 21.5 M on
// k_tower16 (31 of 99 products); 5x128 two groups 1.10 vs 1.20 M.  Returns 16, 21, 32 or 3.
// n = boards the launch is expected to hold (what the forms are priced with), nb >= n = the most it can hold (what the split
// tower's co-residency needs; the grid is sized for nb in any case and workgroups beyond the device's count leave at once).
// INVARIANT: the choice depends on a launch size the host reads from a device-written word WITHOUT synchronising
// (wave_net_f: h_nleaf), so which form serves a wave differs from run to run.  Results do not, because every fp32 tower form
// computes the same bits and every bf16 form the same bits (tests/test_net.py, tests/test_net_bf16_gpu.py force each form and compare).
// A new form must pass those cross-form equality tests before it may be returned from here.
template <class Gm, int F> static int pick_tower(const az_engine* e, int n, int nb = -1) {
  if (nb < n) nb = n;
  // split tower (k_tower16s, 128 filters): both workgroups of every pair must be resident at once -> all slot groups'
  // launches together at most num_cu workgroups (beyond that the towers queue for CUs and the split only costs: 256
  // slots in two groups 0.62 vs 0.71 M sims/s); its kernel arguments change per launch (epoch): not under hipGraph replay
  // (r4) the count runs over ALL engines of the process on this device that may launch split towers side by side (an arena has
  // two); an engine whose exchange has given up once (another process, a trainer: work this count cannot see) stays unsplit
  const bool can_split = F == 128 && !e->cfg.net_bf16 && !e->use_graphs && e->cfg.num_blocks <= 127 && !e->split_off &&
                         2 * std::max(std::max(1, e->ngroups), split_streams_on_device(e->device)) * ((nb + T16<Gm, F, NTS<Gm>>::TB - 1) / T16<Gm, F, NTS<Gm>>::TB) <= (e->num_cu > 0 ? e->num_cu : 256);
  if (e->tower_pick == 2 && can_split) return 2;
  if (!e->cfg.net_bf16 && (e->tower_pick == 16 || e->tower_pick == 32 || e->tower_pick == 3 || ((e->tower_pick == 21 || e->tower_pick == 19 || e->tower_pick == 20) && F == 64) || (e->tower_pick == 7 && NTM<Gm> > 0))) return e->tower_pick;
  if (e->cfg.net_bf16) {                                           // k_tower16b: 22 (128 filters), 11 or 3 row tiles
    if (e->tower_pick == 3 || e->tower_pick == 16) return e->tower_pick;
    if (e->tower_pick == 22 && F == 128) return 22;
    const long cu2 = e->num_cu > 0 ? e->num_cu : 256;
    const long a16 = (n + T16B<Gm, F>::TB - 1) / T16B<Gm, F>::TB, a3 = (n + T16B<Gm, F, NTS<Gm>>::TB - 1) / T16B<Gm, F, NTS<Gm>>::TB;
    const long per = F == 64 ? 2 : 1;                               // workgroups per CU
    // rounds x rows per workgroup x fraction of the tile-tap products executed (Geo16)
    constexpr double g16 = T16B<Gm, F>::Geo::tab.cost / (9.0 * 11), g3 = T16B<Gm, F, NTS<Gm>>::Geo::tab.cost / (9.0 * NTS<Gm>);
    const double d16 = (double)((a16 + per * cu2 - 1) / (per * cu2)) * T16B<Gm, F>::RPAD * g16;
    const double d3 = 1.1 * (double)((a3 + per * cu2 - 1) / (per * cu2)) * T16B<Gm, F, NTS<Gm>>::RPAD * g3;
    if constexpr (F == 128) {
      // 22 row tiles = twice the boards per workgroup: each weight fragment serves twice the rows (10x128, 4096 boards: 578
      // vs 684 us); only where it does not leave CUs idle that the 11-tile form would use
      constexpr double g22 = T16B<Gm, F, 22>::Geo::tab.cost / (9.0 * 22);
      const long a22 = (n + T16B<Gm, F, 22>::TB - 1) / T16B<Gm, F, 22>::TB;
      const double d22 = 0.9 * (double)((a22 + cu2 - 1) / cu2) * T16B<Gm, F, 22>::RPAD * g22;
      if (d22 <= d16 && d22 <= d3) return 22;
    }
    return d3 <= d16 ? 3 : 16;
  }
  const long cu = e->num_cu > 0 ? e->num_cu : 256;
  const long b16 = (n + T16<Gm, F>::TB - 1) / T16<Gm, F>::TB, b32 = (n + TOWER_ROWS / Gm::P - 1) / (TOWER_ROWS / Gm::P);
  const long b3 = (n + T16<Gm, F, NTS<Gm>>::TB - 1) / T16<Gm, F, NTS<Gm>>::TB;
  // the k_tower16 family executes only the (tile, tap) products that touch the board (Geo16): cost x executed fraction
  constexpr double f16 = T16<Gm, F>::Geo::tab.cost / (9.0 * T16<Gm, F>::NTILE), f3 = T16<Gm, F, NTS<Gm>>::Geo::tab.cost / (9.0 * NTS<Gm>),
                   f21 = T16P<Gm, 64>::Geo::tab.cost / (9.0 * T16P<Gm, 64>::NTW);
  const double c16 = (double)((b16 + cu - 1) / cu) * T16<Gm, F>::RPAD * f16;
  const double c32 = (double)((b32 + cu - 1) / cu) * TOWER_ROWS;     // k_tower (32x32x2) computes every tap
  const double c3 = 1.1 * (double)((b3 + cu - 1) / cu) * T16<Gm, F, NTS<Gm>>::RPAD * f3;
  if (c3 <= c16 && c3 <= c32) return can_split && e->tower_pick != 3 ? 2 : 3;
  // (r4) the exact-fit variant (NTM tiles: whole boards, no padding rows): where its workgroups fill the CUs evenly it beats
  // the 11-tile form by the padding rows and by the last, partly filled round of workgroups
  if constexpr (NTM<Gm> > 0) {
    using TM = T16<Gm, F, NTM<Gm>>;
    constexpr double fm = TM::Geo::tab.cost / (9.0 * NTM<Gm>);
    const long bm = (n + TM::TB - 1) / TM::TB;
    // +3 %: fewer rows per weight fragment; the half-size form at 128 filters +20 %: half the rows per weight fragment there (the
    // vector-memory path that streams the weights is what a 128-filter layer waits for: DESIGN.md 4d, the trainer's 6-tile layer)
    const double cm = (Gm::P == 42 && F == 128 ? 1.2 : 1.03) * (double)((bm + cu - 1) / cu) * TM::RPAD * fm;
    if (cm < c16 && cm < c32 && (F != 64 || cm < (double)(((n + T16P<Gm, 64>::TB - 1) / T16P<Gm, 64>::TB + cu - 1) / cu) * T16P<Gm, 64>::RPAD * f21)) return 7;
  }
  // paired k_tower16x2: 336 rows = 8 Connect-Four boards per workgroup, no padding rows.  Its 92 KB of LDS allow one
  // workgroup per CU, so with several slot groups the groups' towers cannot interleave on a CU (Connect-Four, two groups:
  // 3.80 vs 4.11 M sims/s in round 1): +8 % on its cost there
  if (F == 64) {
    const long b21 = (n + T16P<Gm, 64>::TB - 1) / T16P<Gm, 64>::TB;
    const double c21 = (double)((b21 + cu - 1) / cu) * T16P<Gm, 64>::RPAD * f21 * (e->ngroups > 1 ? 1.08 : 1.0);
    if (c21 < c16 && c21 < c32) return 21;
  }
  return c16 <= c32 ? 16 : 32;
}
// Fraction of a 3x3 convolution's (row tile, tap) products the tower kernel `tw` executes (Geo16 skips the products whose tap
// falls off the board for a whole tile); k_tower (32x32x2) computes every tap.  az_prof.exec_units weighs a launch's boards
// with it, so that a roofline can count the work DONE instead of the dense convolution's.
template <class Gm, int F> static double tower_exec_frac(const az_engine* e, int tw) {
  if (e->cfg.net_bf16) {
    if constexpr (F == 128) { if (tw == 22) return T16B<Gm, F, 22>::Geo::tab.cost / (9.0 * 22); }
    if (tw == 3) return T16B<Gm, F, NTS<Gm>>::Geo::tab.cost / (9.0 * NTS<Gm>);
    return T16B<Gm, F>::Geo::tab.cost / (9.0 * 11);
  }
  if (tw == 2 || tw == 3) return T16<Gm, F, NTS<Gm>>::Geo::tab.cost / (9.0 * NTS<Gm>);
  if constexpr (NTM<Gm> > 0) { if (tw == 7) return T16<Gm, F, NTM<Gm>>::Geo::tab.cost / (9.0 * NTM<Gm>); }
  if (tw == 16) return T16<Gm, F>::Geo::tab.cost / (9.0 * T16<Gm, F>::NTILE);
  if (tw == 21 || tw == 20) return T16P<Gm, 64>::Geo::tab.cost / (9.0 * T16P<Gm, 64>::NTW);
  if (tw == 19) return T16P<Gm, 64, 10, 9>::Geo::tab.cost / (9.0 * T16P<Gm, 64, 10, 9>::NTW);
  return 1.0;
}
// publish areas of k_tower16s for the launches that write `hfeat` (one feature buffer = one stream at a time)
template <class Gm> static int xch_slot(az_engine* e, const float* hfeat, unsigned long long** xch, unsigned long long* epoch) {
  using T = T16S<Gm, 128>;
  const size_t words = (size_t)(e->num_cu > 0 ? e->num_cu : 256) * T::XCH_WORDS;
  int slot = AZ_MAX_GROUPS;
  for (int g = 0; g < e->ngroups; ++g) if (hfeat == e->g_hfeat[g]) slot = g;
  if (!e->xch[slot]) {
    // zeroed: tag 0 is never expected.  The clear runs on e->stream while the launch that follows is on the group's own
    // (non-blocking) stream: it must have finished before that kernel polls the area -- recycled device memory can hold an
    // earlier engine's words, and every engine's epochs start at 1, so stale tags WOULD match (seen once in a full test
    // run: a two-group engine created after other engines had been freed evaluated garbage)
    AZCHK(dalloc(e, &e->xch[slot], words));
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  // a tag carries 24 bits of the launch epoch: before they repeat, every area is cleared (an area that a long run of
  // smaller launches has not touched could otherwise still hold words with the tag of exactly 2^24 launches ago)
  if (((e->xch_epoch + 1) & 0xFFFFFFull) == 0) {
    AZCHK(sync_all(e));
    for (int k = 0; k <= AZ_MAX_GROUPS; ++k) if (e->xch[k]) HIPCHK(hipMemset(e->xch[k], 0, words * sizeof(unsigned long long)));
    ++e->xch_epoch;                                                  // skip the epoch whose tags would start at 0
  }
  *xch = e->xch[slot];
  *epoch = ++e->xch_epoch;
  if (e->xch_fail_at > 0 && ++e->xch_launches == e->xch_fail_at) *epoch |= 1ull << 63;   // tests: this launch loses a partner (AZHIP_XCH_FAIL_AT)
  return AZ_OK;
}
// Dense heads of n_max boards: 16-board tiles (k_heads16: a quarter of the chain latency, 23 instead of 43 us per launch
// at 64 filters whatever the size) unless several slot groups run big launches side by side -- there the heads hide
// under the other group's tower anyway and twice as many workgroups asking for a CU cost 1 % (0.867 vs 0.857 ms per
// step, two groups of 2048): 32-board tiles (k_heads_mfma).  The vector-ALU kernel when the head filters do not fit
// the MFMA fragments.
static constexpr int HEADS16_MAX_BOARDS = 1024;
template <class Gm, int F>
static int launch_heads(az_engine* e, hipStream_t st, const float* hfeat, const GEnv* envs, const int* eslots, const int* n_ptr, int n_max,
                        const float* Amask, float* Pout, float* Vout, float* Pinv, int pstride) {
  const bool small = e->heads_pick == 16 || (e->heads_pick != 32 && (n_max <= HEADS16_MAX_BOARDS || e->ngroups == 1));
  const int tiles = (n_max + 15) / 16;
  if (e->net.hd16_ok && small && 2 * tiles <= (e->num_cu > 0 ? e->num_cu : 256))    // value and policy tiles in two workgroups, features staged in LDS
    LAUNCH_ON(e, st, AZ_K_HEADS, n_max, (k_heads16<Gm, F, true>), 2 * tiles, (H16<Gm, F, true>::THREADS), (H16<Gm, F, true>::BYTES), e->net, envs, eslots, n_ptr, n_max, Amask, hfeat, Pout, Vout, Pinv, pstride);
  else if (e->net.hd16_ok && small)
    LAUNCH_ON(e, st, AZ_K_HEADS, n_max, (k_heads16<Gm, F, false>), tiles, (H16<Gm, F, false>::THREADS), (H16<Gm, F, false>::BYTES), e->net, envs, eslots, n_ptr, n_max, Amask, hfeat, Pout, Vout, Pinv, pstride);
  else if (e->net.hd_ok)
    LAUNCH_ON(e, st, AZ_K_HEADS, n_max, (k_heads_mfma<Gm, F>), (n_max + 31) / 32, 64 * (F / 32 + HEADS_NPT<Gm>), 0, e->net, envs, eslots, n_ptr, n_max, Amask, hfeat, Pout, Vout, Pinv, pstride);
  else
    LAUNCH_ON(e, st, AZ_K_HEADS, n_max, (k_heads<Gm, F>), (n_max + 3) / 4, 4 * (F + HEADS_LP<Gm>), 0, e->net, envs, eslots, n_ptr, n_max, Amask, hfeat, Pout, Vout, Pinv, pstride);
  return AZ_OK;
}
template <class Gm, int F, bool FROM_PLANES>
static int launch_net_f(az_engine* e, hipStream_t st, float* hfeat, const GEnv* envs, const int* eslots, const int* n_ptr, int n_max, const float* X,
                        const float* Amask, float* Pout, float* Vout, float* Pinv, int pstride) {
  constexpr int TB = TOWER_ROWS / Gm::P;
  const int gt = (n_max + TB - 1) / TB;
  if (gt == 0) return AZ_OK;
  constexpr int TB16 = T16<Gm, F>::TB, THR16 = T16<Gm, F>::THREADS, LDS16 = T16<Gm, F>::BYTES;
  constexpr int TB3 = T16<Gm, F, NTS<Gm>>::TB, LDS3 = T16<Gm, F, NTS<Gm>>::BYTES;
  constexpr int TB21 = T16P<Gm, 64>::TB, THR21 = T16P<Gm, 64>::THREADS, LDS21 = T16P<Gm, 64>::BYTES;
  const int tw = pick_tower<Gm, F>(e, n_max);
  note_tower(e, tw, F);
  e->next_exec = tower_exec_frac<Gm, F>(e, tw);
  if (e->cfg.net_bf16) {
    constexpr int TBb = T16B<Gm, F>::TB, THRb = T16B<Gm, F>::THREADS, LDSb = T16B<Gm, F>::BYTES, TBb3 = T16B<Gm, F, NTS<Gm>>::TB, LDSb3 = T16B<Gm, F, NTS<Gm>>::BYTES;
    if (tw == 22) {
      constexpr int TB22 = T16B<Gm, F, 22>::TB, LDS22 = T16B<Gm, F, 22>::BYTES;
      if constexpr (F == 128) LAUNCH_ON(e, st, AZ_K_TOWER, n_max, (k_tower16b<Gm, F, FROM_PLANES, 22>), (n_max + TB22 - 1) / TB22, THRb, LDS22, e->net16b, envs, eslots, n_ptr, n_max, X, hfeat);
    } else if (tw == 3) LAUNCH_ON(e, st, AZ_K_TOWER, n_max, (k_tower16b<Gm, F, FROM_PLANES, NTS<Gm>>), (n_max + TBb3 - 1) / TBb3, THRb, LDSb3, e->net16b, envs, eslots, n_ptr, n_max, X, hfeat);
    else LAUNCH_ON(e, st, AZ_K_TOWER, n_max, (k_tower16b<Gm, F, FROM_PLANES, 11>), (n_max + TBb - 1) / TBb, THRb, LDSb, e->net16b, envs, eslots, n_ptr, n_max, X, hfeat);
  } else if (tw == 2) {
    if constexpr (F == 128) {
      unsigned long long* xa; unsigned long long ep;
      AZCHK(xch_slot<Gm>(e, hfeat, &xa, &ep));
      LAUNCH_ON(e, st, AZ_K_TOWER, n_max, (k_tower16s<Gm, F, FROM_PLANES>), 2 * ((n_max + TB3 - 1) / TB3), (T16S<Gm, F>::THREADS), LDS3, e->net16, envs, eslots, n_ptr, n_max, X, hfeat,
                xa, ep, e->v.err, e->d_xflag);
    }
  } else if (tw == 21) {
    if constexpr (F == 64)
      LAUNCH_ON(e, st, AZ_K_TOWER, n_max, (k_tower16x2<Gm, F, FROM_PLANES>), (n_max + TB21 - 1) / TB21, THR21, LDS21, e->net16, envs, eslots, n_ptr, n_max, X, hfeat, 0);
  } else if (tw == 20) {
    if constexpr (F == 64)
      LAUNCH_ON(e, st, AZ_K_TOWER, n_max, (k_tower16x2c<Gm, F, FROM_PLANES>), (n_max + TB21 - 1) / TB21, THR21, LDS21, e->net16, envs, eslots, n_ptr, n_max, X, hfeat, 0);
  } else if (tw == 19) {
    if constexpr (F == 64) {
      using T19 = T16P<Gm, 64, 10, 9>;
      LAUNCH_ON(e, st, AZ_K_TOWER, n_max, (k_tower16x2<Gm, F, FROM_PLANES, 10, 9>), (n_max + T19::TB - 1) / T19::TB, (T19::THREADS), (T19::BYTES), e->net16, envs, eslots, n_ptr, n_max, X, hfeat, 0);
    }
  } else if (tw == 7) {
    if constexpr (NTM<Gm> > 0) {
      using TM = T16<Gm, F, NTM<Gm>>;
      LAUNCH_ON(e, st, AZ_K_TOWER, n_max, (k_tower16<Gm, F, FROM_PLANES, NTM<Gm>>), (n_max + TM::TB - 1) / TM::TB, THR16, (TM::BYTES), e->net16, envs, eslots, n_ptr, n_max, X, hfeat);
    }
  } else if (tw == 3)
    LAUNCH_ON(e, st, AZ_K_TOWER, n_max, (k_tower16<Gm, F, FROM_PLANES, NTS<Gm>>), (n_max + TB3 - 1) / TB3, THR16, LDS3, e->net16, envs, eslots, n_ptr, n_max, X, hfeat);
  else if (tw == 16)
    LAUNCH_ON(e, st, AZ_K_TOWER, n_max, (k_tower16<Gm, F, FROM_PLANES>), (n_max + TB16 - 1) / TB16, THR16, LDS16, e->net16, envs, eslots, n_ptr, n_max, X, hfeat);
  else
    LAUNCH_ON(e, st, AZ_K_TOWER, n_max, (k_tower<Gm, F, FROM_PLANES>), gt, TowerCfg<F>::THREADS, TowerLds<F>::BYTES, e->net, envs, eslots, n_ptr, n_max, X, hfeat);
  return launch_heads<Gm, F>(e, st, hfeat, envs, eslots, n_ptr, n_max, Amask, Pout, Vout, Pinv, pstride);
}
// launches tower + heads on `n` boards (device count in n_ptr when given)
template <class Gm, bool FROM_PLANES>
static int launch_net(az_engine* e, hipStream_t st, float* hfeat, const GEnv* envs, const int* eslots, const int* n_ptr, int n_max, const float* X,
                      const float* Amask, float* Pout, float* Vout, float* Pinv, int pstride) {
  if (e->cfg.num_filters == 128) return launch_net_f<Gm, 128, FROM_PLANES>(e, st, hfeat, envs, eslots, n_ptr, n_max, X, Amask, Pout, Vout, Pinv, pstride);
  return launch_net_f<Gm, 64, FROM_PLANES>(e, st, hfeat, envs, eslots, n_ptr, n_max, X, Amask, Pout, Vout, Pinv, pstride);
}


template <class Gm, int F> static int wave_net_f(az_engine* e, int g, bool split, int nmax) {
  constexpr int L = Gm::APAD, TB = TOWER_ROWS / Gm::P;
  const DView& v = e->gv[g];
  hipStream_t st = e->gs[g], sn = e->gt[g];
  const int G = v.G;
  if (split) { HIPCHK(hipEventRecord(e->ev_tree[g], st)); HIPCHK(hipStreamWaitEvent(sn, e->ev_tree[g], 0)); }
  constexpr int TB16 = T16<Gm, F>::TB, THR16 = T16<Gm, F>::THREADS, LDS16 = T16<Gm, F>::BYTES;
  constexpr int TB3 = T16<Gm, F, NTS<Gm>>::TB, LDS3 = T16<Gm, F, NTS<Gm>>::BYTES;
  constexpr int TB21 = T16P<Gm, 64>::TB, THR21 = T16P<Gm, 64>::THREADS, LDS21 = T16P<Gm, 64>::BYTES;
  // N = upper bound of this wave's leaves: the group's active slots (a draining phase or a partial explore! launches
  // -- and picks its tower kernel -- for what is left, not for the group's capacity)
  const int N = std::max(1, std::min(G, nmax));
  const int* nev = v.n_eval + e->wave_par[g];                      // the leaf counter of this wave (k_tree)
  const int* const eslots = v.eval_slots + (size_t)e->wave_par[g] * v.eval_stride;   // ... and its batch -> slot list
  // how many boards the launch will really hold: with the evaluation cache answering part of every wave, what the device
  // reported for the group's last completed wave (+ 1/8: a launch priced too small costs more than one priced too large)
  int npick = N;
  if (v.nleaf_host) { const int seen = ((volatile int*)e->h_nleaf)[g]; if (seen >= 0) npick = std::max(1, std::min(N, seen + seen / 8 + 8)); }
  int tw = pick_tower<Gm, F>(e, npick, N);
  if (tw == 21 && e->fr_on && e->ngroups == 1 && e->tower_pick == 0 && e->fr_kbg > 0 && v.nleaf_host && v.needy_host) {
    // The paired tower beside the side streams' kernels (resnet16.h k_tower16x2c): a CU that holds a workgroup of the background search
    // (256 threads = 32 Connect-Four slots of the needy list) takes no workgroup of the 198-register form for as long as that search
    // runs.  Measured: in the steady state (~8 such workgroups on average, more in some waves) the launch gains a round of workgroups
    // beyond ~478 of its 512; the margin of 14 is that measurement (the move step is not part of it: within 96 registers it changes
    // nothing).  Where the counts the device reported for the previous wave say that this adds a round, the 176-register form serves
    // the launch.
    const int cu = e->num_cu > 0 ? e->num_cu : 256;
    const int seen = ((volatile int*)e->h_nleaf)[g], needy = ((volatile int*)e->h_needy)[g];
    if (seen > 0) {
      const int wgs = (seen + TB21 - 1) / TB21, rounds = (wgs + cu - 1) / cu, bg = (std::max(needy, 0) * L + 255) / 256;
      if (wgs > rounds * (cu - bg) - 14) tw = 20;
    }
  }
  note_tower(e, tw, F);
  e->next_exec = tower_exec_frac<Gm, F>(e, tw);
  if (e->cfg.net_bf16) {
    constexpr int TBb = T16B<Gm, F>::TB, THRb = T16B<Gm, F>::THREADS, LDSb = T16B<Gm, F>::BYTES, TBb3 = T16B<Gm, F, NTS<Gm>>::TB, LDSb3 = T16B<Gm, F, NTS<Gm>>::BYTES;
    if (tw == 22) {
      constexpr int TB22 = T16B<Gm, F, 22>::TB, LDS22 = T16B<Gm, F, 22>::BYTES;
      if constexpr (F == 128) LAUNCH_ON(e, sn, AZ_K_TOWER, N, (k_tower16b<Gm, F, false, 22>), (N + TB22 - 1) / TB22, THRb, LDS22, e->net16b, v.leaf_env, eslots, nev, N, (const float*)nullptr, e->g_hfeat[g]);
    } else if (tw == 3) LAUNCH_ON(e, sn, AZ_K_TOWER, N, (k_tower16b<Gm, F, false, NTS<Gm>>), (N + TBb3 - 1) / TBb3, THRb, LDSb3, e->net16b, v.leaf_env, eslots, nev, N, (const float*)nullptr, e->g_hfeat[g]);
    else LAUNCH_ON(e, sn, AZ_K_TOWER, N, (k_tower16b<Gm, F, false, 11>), (N + TBb - 1) / TBb, THRb, LDSb, e->net16b, v.leaf_env, eslots, nev, N, (const float*)nullptr, e->g_hfeat[g]);
  } else if (tw == 2) {
    if constexpr (F == 128) {
      unsigned long long* xa; unsigned long long ep;
      AZCHK(xch_slot<Gm>(e, e->g_hfeat[g], &xa, &ep));
      LAUNCH_ON(e, sn, AZ_K_TOWER, N, (k_tower16s<Gm, F, false>), 2 * ((N + TB3 - 1) / TB3), (T16S<Gm, F>::THREADS), LDS3, e->net16, v.leaf_env, eslots, nev, N, (const float*)nullptr, e->g_hfeat[g],
                xa, ep, v.xerr, e->d_xflag);
    }
  } else if (tw == 20) {
    if constexpr (F == 64)
      LAUNCH_ON(e, sn, AZ_K_TOWER, N, (k_tower16x2c<Gm, F, false>), (N + TB21 - 1) / TB21, THR21, LDS21, e->net16, v.leaf_env, eslots, nev, N, (const float*)nullptr, e->g_hfeat[g], 0);
  } else if (tw == 21 || tw == 19) {
    if constexpr (F == 64) {
      // (r6, measured, OFF by default) Two rounds of workgroups, the second one lighter.  A batch of b boards with 8 cu < b <= 8 cu + 7 cu
      // (cu = 256: 2048 < b <= 3840 -- a free-running wave's batch at 4096 slots is ~3800) costs two rounds of the 8-board form whatever b
      // is; served as one launch of cu 8-board workgroups and one of up to cu 7-board workgroups for the rest, the second round should be
      // 7/8 as long.  It is not: a launch of ONE round lasts as long as its slowest workgroup (490 us against the 399 us a round
      // averages inside a two-round launch, where a CU that is done early simply takes the next workgroup; the 7-board launch 677 us):
      // 1.10 ms per wave instead of 0.80 (profiles/r6/README.md).  The 19-tile form stays (bit-exact like every form, tests/test_net.py
      // forces it with AZHIP_TOWER=19; 355 us per round of 7-board workgroups); AZHIP_TOWER_ROUNDS=1 switches the two-launch scheme on.
      // (r6) ... and as ONE launch of both forms (k_tower16x2m, AZHIP_TOWER_MIXED=1): 1.09 ms per wave as well.  So it is not the lost
      // balance between two launches: a workgroup's time hardly shrinks with its boards (399 us for 8, 355 for 7 when a whole launch is
      // 7-board workgroups) because every workgroup streams the same 1.5 MB of weights through its CU's vector-memory path, and a second
      // round of 252 seven-board workgroups beside a first round's stragglers is no cheaper than 221 eight-board ones.  Boards per
      // workgroup want to go UP (k_tower16b's 22 tiles), which fp32 activations in 160 KB of LDS do not allow.  Both schemes stay off.
      static const bool mixed_on = getenv("AZHIP_TOWER_MIXED") && atoi(getenv("AZHIP_TOWER_MIXED")) != 0;
      using T19 = T16P<Gm, 64, 10, 9>;
      static const bool rounds_on = getenv("AZHIP_TOWER_ROUNDS") && atoi(getenv("AZHIP_TOWER_ROUNDS")) != 0;
      const int cu = e->num_cu > 0 ? e->num_cu : 256, first = cu * TB21;
      const int seen = v.nleaf_host ? ((volatile int*)e->h_nleaf)[g] : -1;
      if (tw == 19) {
        LAUNCH_ON(e, sn, AZ_K_TOWER, N, (k_tower16x2<Gm, F, false, 10, 9>), (N + T19::TB - 1) / T19::TB, (T19::THREADS), (T19::BYTES), e->net16, v.leaf_env, eslots, nev, N, (const float*)nullptr, e->g_hfeat[g], 0);
      } else if (mixed_on && !rounds_on && e->tower_pick == 0 && N > first && seen > first && seen + 16 <= first + cu * T19::TB) {
        // the launch holds boards of both forms: price the executed fraction by their shares of the expected batch
        e->next_exec = (first * tower_exec_frac<Gm, F>(e, 21) + (seen - first) * tower_exec_frac<Gm, F>(e, 19)) / (double)seen;
        LAUNCH_ON(e, sn, AZ_K_TOWER, N, (k_tower16x2m<Gm, F, false>), cu + (N - first + T19::TB - 1) / T19::TB, THR21, LDS21, e->net16, v.leaf_env, eslots, nev, N, (const float*)nullptr, e->g_hfeat[g], cu);
        snprintf(e->last_tower, sizeof e->last_tower, "k_tower16x2m<%s,%d>", Gm::ID == 0 ? "ConnectFour" : Gm::ID == 1 ? "TicTacToe" : "Mancala", F);
      } else if (rounds_on && e->tower_pick == 0 && N > first && seen > first && seen + 16 <= first + cu * T19::TB) {
        LAUNCH_ON(e, sn, AZ_K_TOWER, first, (k_tower16x2<Gm, F, false>), cu, THR21, LDS21, e->net16, v.leaf_env, eslots, nev, N, (const float*)nullptr, e->g_hfeat[g], 0);
        e->next_exec = tower_exec_frac<Gm, F>(e, 19);
        LAUNCH_ON(e, sn, AZ_K_TOWER, N - first, (k_tower16x2<Gm, F, false, 10, 9>), (N - first + T19::TB - 1) / T19::TB, (T19::THREADS), (T19::BYTES), e->net16, v.leaf_env, eslots, nev, N, (const float*)nullptr, e->g_hfeat[g], first);
      } else {
        LAUNCH_ON(e, sn, AZ_K_TOWER, N, (k_tower16x2<Gm, F, false>), (N + TB21 - 1) / TB21, THR21, LDS21, e->net16, v.leaf_env, eslots, nev, N, (const float*)nullptr, e->g_hfeat[g], 0);
      }
    }
  } else if (tw == 7) {
    if constexpr (NTM<Gm> > 0) {
      using TM = T16<Gm, F, NTM<Gm>>;
      LAUNCH_ON(e, sn, AZ_K_TOWER, N, (k_tower16<Gm, F, false, NTM<Gm>>), (N + TM::TB - 1) / TM::TB, THR16, (TM::BYTES), e->net16, v.leaf_env, eslots, nev, N, (const float*)nullptr, e->g_hfeat[g]);
    }
  } else if (tw == 3)
    LAUNCH_ON(e, sn, AZ_K_TOWER, N, (k_tower16<Gm, F, false, NTS<Gm>>), (N + TB3 - 1) / TB3, THR16, LDS3, e->net16, v.leaf_env, eslots, nev, N, (const float*)nullptr, e->g_hfeat[g]);
  else if (tw == 16)
    LAUNCH_ON(e, sn, AZ_K_TOWER, N, (k_tower16<Gm, F, false>), (N + TB16 - 1) / TB16, THR16, LDS16, e->net16, v.leaf_env, eslots, nev, N, (const float*)nullptr, e->g_hfeat[g]);
  else
    LAUNCH_ON(e, sn, AZ_K_TOWER, N, (k_tower<Gm, F, false>), (N + TB - 1) / TB, TowerCfg<F>::THREADS, TowerLds<F>::BYTES, e->net, v.leaf_env, eslots, nev, N, (const float*)nullptr, e->g_hfeat[g]);
  // Free-running phase: the heads follow the tower on the network stream and the tree stream is NOT made to wait here -- the caller
  // (wave_group, azhip.hip) first queues the group's move step and its background search behind k_tree, then the wait for ev_net: both
  // run under this tower.
  const bool fr = e->fr_on && split;
  if (e->bg_signal) {
    // the tower has run: the background search of this wave winds up while the heads run.  The word is set from a stream of its own behind
    // an event (a one-thread launch IN the wave's stream sat 7.6 us between tower and heads)
    HIPCHK(hipEventRecord(e->fr_ev[3], sn));
    HIPCHK(hipStreamWaitEvent(e->fr_s[3], e->fr_ev[3], 0));
    hipLaunchKernelGGL(k_set_word, dim3(1), dim3(1), 0, e->fr_s[3], e->d_bg_stop, e->bg_seq);
  }
  if (split && !fr) { HIPCHK(hipEventRecord(e->ev_net[g], sn)); HIPCHK(hipStreamWaitEvent(st, e->ev_net[g], 0)); }
  AZCHK((launch_heads<Gm, F>(e, fr ? sn : st, e->g_hfeat[g], v.leaf_env, eslots, nev, N, (const float*)nullptr, v.Pout, v.Vout, (float*)nullptr, L)));
  if (fr) HIPCHK(hipEventRecord(e->ev_net[g], sn));
  return AZ_OK;
}
template <class Gm> static int wave_net(az_engine* e, int g, bool split, int nmax) {
  return e->cfg.num_filters == 128 ? wave_net_f<Gm, 128>(e, g, split, nmax) : wave_net_f<Gm, 64>(e, g, split, nmax);
}

// the three entry points of one game's translation unit
#define AZ_NET_GAME_TU(Gm, sfx)                                                                                              \
  int net_set_kernel_attrs_##sfx(az_engine* e) { return set_kernel_attrs<Gm>(e); }                                                      \
  int net_launch_##sfx(az_engine* e, hipStream_t st, bool from_planes, float* hfeat, const GEnv* envs, const int* eslots,    \
                       const int* n_ptr, int n_max, const float* X, const float* Amask, float* Pout, float* Vout, float* Pinv, \
                       int pstride) {                                                                                        \
    return from_planes ? launch_net<Gm, true>(e, st, hfeat, envs, eslots, n_ptr, n_max, X, Amask, Pout, Vout, Pinv, pstride)  \
                       : launch_net<Gm, false>(e, st, hfeat, envs, eslots, n_ptr, n_max, X, Amask, Pout, Vout, Pinv, pstride); \
  }                                                                                                                          \
  int net_wave_##sfx(az_engine* e, int g, bool split, int nmax) { return wave_net<Gm>(e, g, split, nmax); }                  \
  int net_geometry_##sfx(int which, uint16_t* out, int64_t cap, int* rows, int* products) {                                  \
    return which == 0 ? geometry_out<typename T16<Gm, 64, 11>::Geo>(out, cap, rows, products)                                \
         : which == 1 ? geometry_out<typename T16<Gm, 64, NTS<Gm>>::Geo>(out, cap, rows, products)                                 \
         : which == 2 ? geometry_out<typename T16P<Gm, 64>::Geo>(out, cap, rows, products)                                   \
                      : geometry_out<typename T16B<Gm, 128, 22>::Geo>(out, cap, rows, products);                             \
  }
