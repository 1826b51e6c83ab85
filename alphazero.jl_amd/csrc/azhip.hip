// azhip.hip -- host side of libazhip.so: engine lifetime, the lock-step self-play driver and the
// C ABI of include/azhip.h.  Device code lives in tree.h (search) and resnet.h (network).
//
// Reference seams: simulate (src/simulations.jl:207-244) -> az_selfplay_*;
// MCTS.explore!/policy/reset! (src/mcts.jl:239-281) -> az_mcts_*; Network.forward_normalized /
// evaluate_batch (src/networks/network.jl:264-315) -> az_net_*; GameInterface -> az_game_*.
#include "engine.h"

// ------------------------------------------------------------------------------- errors
static thread_local std::string g_err;
int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
extern "C" const char* az_last_error(void) { return g_err.c_str(); }
extern "C" int az_abi_version(void) { return AZ_ABI_VERSION; }
extern "C" int az_abi_struct_size(int32_t which) {
  switch (which) {
    case AZ_STRUCT_ENGINE_CFG: return (int)sizeof(az_engine_cfg);
    case AZ_STRUCT_MOVE_REC: return (int)sizeof(az_move_rec);
    case AZ_STRUCT_GAME_REC: return (int)sizeof(az_game_rec);
    case AZ_STRUCT_TRACE_BUF: return (int)sizeof(az_trace_buf);
    case AZ_STRUCT_SELFPLAY_STATS: return (int)sizeof(az_selfplay_stats);
    case AZ_STRUCT_SAMPLE: return (int)sizeof(az_sample);
    case AZ_STRUCT_DATASET_INFO: return (int)sizeof(az_dataset_info);
    case AZ_STRUCT_LEARNING_STATUS: return (int)sizeof(az_learning_status_t);
    case AZ_STRUCT_TRAIN_CFG: return (int)sizeof(az_train_cfg);
    case AZ_STRUCT_GATHER_STATS: return (int)sizeof(az_gather_stats);
    case AZ_STRUCT_PROF: return (int)sizeof(az_prof);
  }
  return -1;
}

// Streams of this process that may launch split towers (k_tower16s) on a device at the same time: the sum of the slot groups of
// all engines with 128-filter fp32 networks that have a search IN PROGRESS (between az_selfplay_begin / explore_begin and their
// end: an arena drives two engines side by side; an idle engine -- a host keeps them cached between phases -- holds no CU and is
// not counted: counting it cost the arena of a training iteration 12.7 s instead of 5.6).  pick_tower keeps the split only
// while all of them together fit the chip, so that every pair of workgroups is co-resident.
static std::mutex g_split_mu;
static std::map<int, int> g_split_streams;
int split_streams_on_device(int device) {
  std::lock_guard<std::mutex> lk(g_split_mu);
  auto it = g_split_streams.find(device);
  return it == g_split_streams.end() ? 0 : it->second;
}
static void split_register(az_engine* e, int n) {
  if (!(e->cfg.oracle == AZ_ORACLE_RESNET && e->cfg.num_filters == 128 && !e->cfg.net_bf16)) n = 0;   // never launches split towers
  std::lock_guard<std::mutex> lk(g_split_mu);
  g_split_streams[e->device] += n - e->split_registered;
  e->split_registered = n;
}

int check_device_error(az_engine* e) {
  int code = 0;
  AZCHK(sync_groups(e));
  HIPCHK(hipMemcpyAsync(&code, e->v.err, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipGetLastError());
  if (code == 0) return AZ_OK;
  int zero = 0;
  HIPCHK(hipMemcpy(e->v.err, &zero, sizeof(int), hipMemcpyHostToDevice));
  switch (code) {
    case DERR_NODE_POOL: return fail(AZ_ERR_CAPACITY, "node pool exhausted (max_nodes_per_slot = %d)", e->v.cap_nodes);
    case DERR_HASH_FULL: return fail(AZ_ERR_CAPACITY, "hash table full (%d entries per slot)", e->v.ht_size);
    case DERR_DEPTH: return fail(AZ_ERR_CAPACITY, "simulation path deeper than %d plies", e->v.max_depth);
    case DERR_MOVES: return fail(AZ_ERR_CAPACITY, "game longer than max_moves_per_game = %d", e->v.max_moves);
    case DERR_NO_ROOT: return fail(AZ_ERR_STATE, "MCTS.explore! must be called before MCTS.policy");
    case DERR_EXCHANGE: return fail(AZ_ERR_HIP, "k_tower16s: a workgroup waited for its partner's half of a layer for too long");
  }
  return fail(AZ_ERR_HIP, "unknown device error %d", code);
}

extern "C" int az_engine_cfg_init(az_engine_cfg* c) {
  if (!c) return fail(AZ_ERR_BAD_ARG, "cfg is NULL");
  memset(c, 0, sizeof *c);
  c->struct_size = (int32_t)sizeof *c;
  c->game = AZ_GAME_CONNECT_FOUR;
  c->oracle = AZ_ORACLE_RESNET;
  // games/connect-four/params.jl:5-30 (self-play), 64-filter trunk (scripts/profile/self_play.jl:28-30)
  c->gamma = 1.0; c->cpuct = 2.0; c->dirichlet_noise_eps = 0.25; c->dirichlet_noise_alpha = 1.0;
  c->prior_temperature = 1.0;
  c->num_iters_per_turn = 600;
  c->temperature_len = 3;
  c->temperature_xs[0] = 0; c->temperature_xs[1] = 20; c->temperature_xs[2] = 30;
  c->temperature_ys[0] = 1.0; c->temperature_ys[1] = 1.0; c->temperature_ys[2] = 0.3;
  c->num_workers = 128; c->batch_size = 64; c->reset_every = 2; c->fill_batches = 1;
  c->flip_probability = 0.0; c->seed = 1;
  c->num_blocks = 5; c->num_filters = 64; c->num_policy_head_filters = 32; c->num_value_head_filters = 32;
  return AZ_OK;
}

static size_t net_nparams(const GameInfo& gi, const az_engine_cfg& c) {
  size_t P = gi.P, F = c.num_filters, npf = c.num_policy_head_filters, nvf = c.num_value_head_filters;
  size_t s = 9 * (size_t)gi.C * F + 5 * F;
  s += (size_t)c.num_blocks * 2 * (9 * F * F + 5 * F);
  s += F * npf + 5 * npf + (size_t)gi.A * P * npf + gi.A;
  s += F * nvf + 5 * nvf + F * P * nvf + F + F + 1;
  return s;
}

static void drop_wave_graphs(az_engine* e);
extern "C" int az_engine_destroy(az_engine* e) {
  if (!e) return AZ_OK;
  (void)hipSetDevice(e->device);
  if (e->ngroups == 1 && e->fr_s[0] && e->gs[0] == e->fr_s[0]) { for (int i = 0; i < 4; ++i) (void)hipStreamSynchronize(e->fr_s[i]); e->gs[0] = e->gt[0] = e->stream; }
  for (int g = 0; g < e->ngroups; ++g) if (e->gs[g] && e->gs[g] != e->stream) {
    (void)hipStreamSynchronize(e->gt[g]); (void)hipStreamSynchronize(e->gs[g]);
    (void)hipStreamDestroy(e->gt[g]); (void)hipStreamDestroy(e->gs[g]);
    (void)hipEventDestroy(e->ev_tree[g]); (void)hipEventDestroy(e->ev_net[g]);
  }
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  split_register(e, 0);
  if (e->h_xflag) (void)hipHostFree(e->h_xflag);
  if (e->h_nleaf) (void)hipHostFree(e->h_nleaf);
  if (e->h_needy) (void)hipHostFree(e->h_needy);
  if (e->h_fr_words) (void)hipHostFree(e->h_fr_words);
  if (e->h_busy) (void)hipHostFree(e->h_busy);
  for (int i = 0; i < 4; ++i) { if (e->fr_s[i]) { (void)hipStreamSynchronize(e->fr_s[i]); (void)hipStreamDestroy(e->fr_s[i]); } if (e->fr_ev[i]) (void)hipEventDestroy(e->fr_ev[i]); }
  if (e->d_done) (void)hipFree(e->d_done);
  if (e->d_done_off) (void)hipFree(e->d_done_off);
  if (e->h_env) (void)hipHostFree(e->h_env);
  if (e->h_n) (void)hipHostFree(e->h_n);
  if (e->h_pv) (void)hipHostFree(e->h_pv);
  drop_wave_graphs(e);
  for (auto& r : e->prof_pool) { if (r.a) (void)hipEventDestroy(r.a); if (r.b) (void)hipEventDestroy(r.b); }
  for (void* q : e->net_allocs) (void)hipFree(q);
  for (void* q : e->allocs) (void)hipFree(q);
  if (e->d_phase) (void)hipFree(e->d_phase);
  for (size_t i = 0; i < e->vm_handles.size(); ++i) { (void)hipMemUnmap(e->vm_at[i], e->vm_chunk); (void)hipMemRelease(e->vm_handles[i]); }
  if (e->vm_base) (void)hipMemAddressFree(e->vm_base, e->vm_bytes);
  if (e->vmk_base) (void)hipMemAddressFree(e->vmk_base, e->vmk_bytes);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
  return AZ_OK;
}

__global__ void k_fill_u32(uint32_t* p, uint32_t val, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = val;
}
__global__ void k_iota(int* p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}

// ------------------------------------------------------------------------------- node pool
// The reference's tree is a Dict that grows as it is used (src/mcts.jl:124-151).  Here a slot's nodes are an array indexed by
// creation order; its worst case (one new node per simulation, every ply of the longest game) is what `cap` says, and for
// Mancala at BASELINE configs[3] (8192 slots x 800 simulations x 128 plies) that is 107 GB of which a phase touches a
// fraction.  Pools above VM_THRESHOLD therefore live in a VIRTUAL address range (hipMemAddressReserve) laid out
// [chunk row][slot][2 MB]: the kernels' address arithmetic is a shift and a mask (node_at, tree.h), and the host backs chunk
// (row, slot) with physical memory (hipMemCreate + hipMemMap, ~30 us each) at the move step before slot s can reach it --
// a slot adds at most one node per wave, so one look at the node counts every num_iters_per_turn waves is enough.
// tools/probes/vmm_probe.hip: the runtime hands out physical memory in 2 MB units whatever size is asked for; handles above a
// few MB crashed it, so every mapping is one 2 MB handle.  If the device runs out of memory (or AZHIP_POOL_GB is reached)
// a slot simply stops growing and is retired when it fills up (DParams::retire).
static constexpr size_t VM_CHUNK = (size_t)2 << 20;
static constexpr size_t VM_THRESHOLD = (size_t)24 << 30;
static int vm_map_at(az_engine* e, char* at) {
  if (e->vm_mapped + VM_CHUNK > e->vm_budget) return 1;             // budget reached: not an error, the slot stops growing
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = e->device;
  hipMemGenericAllocationHandle_t h;
  if (hipMemCreate(&h, VM_CHUNK, &prop, 0) != hipSuccess) { (void)hipGetLastError(); return 1; }
  hipMemAccessDesc acc = {};
  acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  if (hipMemMap(at, VM_CHUNK, 0, h, 0) != hipSuccess || hipMemSetAccess(at, VM_CHUNK, &acc, 1) != hipSuccess) {
    (void)hipGetLastError(); (void)hipMemRelease(h);
    return fail(AZ_ERR_HIP, "hipMemMap of a node-pool chunk failed");
  }
  e->vm_handles.push_back(h); e->vm_at.push_back(at);
  e->vm_mapped += VM_CHUNK;
  return AZ_OK;
}
// chunk (row, slot) of the node pool -- and, first, the 2 MB granule its side records lie in (round 5: the side records follow the
// node chunks; one granule holds the pieces of VM_CHUNK / vmk_piece neighbouring (row, slot) pairs and is mapped when the first of
// them is needed).  1 = out of budget / memory: the slot keeps what it has.
static int vm_map(az_engine* e, int row, int slot) {
  if (e->vmk_base) {
    const size_t gi = ((size_t)row * e->v.G + slot) * e->vmk_piece / VM_CHUNK;
    if (!e->vmk_granule[gi]) {
      const int st = vm_map_at(e, e->vmk_base + gi * VM_CHUNK);
      if (st != AZ_OK) return st;
      e->vmk_granule[gi] = 1;
    }
  }
  return vm_map_at(e, e->vm_base + ((size_t)row * e->v.G + slot) * VM_CHUNK);
}
// backs the chunks the slots will need before the host looks again (`ahead` nodes from now); uploads the new capacities
static int vm_grow(az_engine* e, int ahead) {
  if (!e->vm_rows) return AZ_OK;
  const int G = e->v.G;
  hipLaunchKernelGGL(k_node_counts, dim3((G + 255) / 256), dim3(256), 0, e->stream, e->v, e->d_node_count);
  HIPCHK(hipMemcpyAsync(e->h_node_count.data(), e->d_node_count, sizeof(int) * G, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  bool changed = false;
  for (int s = 0; s < G; ++s) {
    const long long want = std::min<long long>((long long)e->h_node_count[s] + ahead, e->v.cap_nodes);
    while (e->h_slot_cap[s] < want) {
      const int row = e->h_slot_cap[s] / e->vm_chunk_nodes;
      if (row >= e->vm_rows) break;
      const int st = vm_map(e, row, s);
      if (st == 1) break;                                             // out of memory / budget: the slot keeps what it has
      AZCHK(st);
      e->h_slot_cap[s] = std::min<long long>((long long)(row + 1) * e->vm_chunk_nodes, e->v.cap_nodes);
      changed = true;
    }
  }
  if (changed) {
    HIPCHK(hipMemcpyAsync(e->d_slot_cap, e->h_slot_cap.data(), sizeof(int) * G, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  return AZ_OK;
}

extern "C" int az_engine_create(const az_engine_cfg* c, az_engine** out) {
  if (!c || !out) return fail(AZ_ERR_BAD_ARG, "NULL argument");
  *out = nullptr;
  if (c->struct_size != (int32_t)sizeof(az_engine_cfg)) return fail(AZ_ERR_BAD_ARG, "az_engine_cfg size mismatch (%d vs %zu): call az_engine_cfg_init", c->struct_size, sizeof(az_engine_cfg));
  GameInfo gi;
  if (!game_info(c->game, &gi)) return fail(AZ_ERR_BAD_ARG, "unknown game id %d", c->game);
  if (c->oracle < AZ_ORACLE_UNIFORM || c->oracle > AZ_ORACLE_ROLLOUT) return fail(AZ_ERR_BAD_ARG, "unknown oracle kind %d", c->oracle);
  if (c->game == AZ_GAME_GO9_PLANES && c->oracle != AZ_ORACLE_RESNET) return fail(AZ_ERR_BAD_ARG, "the 9x9x4 plane geometry has no device twin: it serves az_net_forward only (ResNet oracle)");
  if (c->num_workers < 1) return fail(AZ_ERR_BAD_ARG, "num_workers must be >= 1");
  if (c->batch_size > c->num_workers) return fail(AZ_ERR_BAD_ARG, "batch_size (%d) must be <= num_workers (%d) (src/params.jl:361-384)", c->batch_size, c->num_workers);
  // 0 = no search: the engine is a NetworkPlayer (play.jl:226-235) for az_arena_run
  if (!(c->num_iters_per_turn >= 2 || (c->num_iters_per_turn == 0 && c->oracle == AZ_ORACLE_RESNET)))
    return fail(AZ_ERR_BAD_ARG, "num_iters_per_turn must be >= 2 (with 1 the policy is 0/0, src/mcts.jl:267), or 0 with the ResNet oracle (NetworkPlayer, arena only)");
  if (!(c->flip_probability >= 0.0 && c->flip_probability <= 1.0)) return fail(AZ_ERR_BAD_ARG, "flip_probability must be in [0, 1]");
  if (c->temperature_len < 1 || c->temperature_len > AZ_SCHED_MAX) return fail(AZ_ERR_BAD_ARG, "temperature schedule needs 1..%d breakpoints", AZ_SCHED_MAX);
  if (c->reset_every < 0) return fail(AZ_ERR_BAD_ARG, "reset_every must be >= 0");
  if (c->lock_step != 0 && c->lock_step != 1) return fail(AZ_ERR_BAD_ARG, "lock_step must be 0 (free-running self-play) or 1");
  if (!(c->prior_temperature >= 0.0)) return fail(AZ_ERR_BAD_ARG, "prior_temperature must be >= 0");
  if (c->oracle == AZ_ORACLE_RESNET) {
    if (c->num_filters != 64 && c->num_filters != 128) return fail(AZ_ERR_BAD_ARG, "num_filters = %d: this build instantiates the 64- and 128-filter towers", c->num_filters);
    int hf = c->num_policy_head_filters + c->num_value_head_filters;
    if (c->num_policy_head_filters < 1 || c->num_value_head_filters < 1 || hf > c->num_filters) return fail(AZ_ERR_BAD_ARG, "head filters %d/%d unsupported (policy + value must be <= num_filters)", c->num_policy_head_filters, c->num_value_head_filters);
    if (c->num_blocks < 0 || c->num_blocks > 64) return fail(AZ_ERR_BAD_ARG, "num_blocks out of range");
    if (c->net_bf16 != 0 && c->net_bf16 != 1) return fail(AZ_ERR_BAD_ARG, "net_bf16 must be 0 or 1");
  }
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (c->device < 0 || c->device >= ndev) return fail(AZ_ERR_BAD_ARG, "device %d not available (%d visible)", c->device, ndev);
  HIPCHK(hipSetDevice(c->device));
  az_engine* e = new (std::nothrow) az_engine();
  if (!e) return fail(AZ_ERR_HIP, "out of host memory");
  e->cfg = *c; e->gi = gi; e->device = c->device; e->stream = nullptr; e->ngroups = 0; e->alloc_bytes = 0;
  e->vm_base = nullptr; e->vm_bytes = 0; e->vm_chunk = 0; e->vm_rows = 0; e->vm_chunk_nodes = 0; e->d_slot_cap = nullptr; e->vm_budget = 0; e->vm_mapped = 0;
  e->vmk_base = nullptr; e->vmk_bytes = 0; e->vmk_piece = 0; e->tree_sort = 0; e->d_perm = nullptr; e->d_perm_prev = nullptr;
  e->next_exec = 1.0;
  e->h_env = nullptr; e->h_n = nullptr; e->h_pv = nullptr; e->h_nleaf = nullptr; e->d_nleaf = nullptr; e->h_needy = nullptr; e->d_needy = nullptr;
  e->d_ec = nullptr; e->ec_mask = 0; e->ec_seq = 0; e->d_ec_claim = nullptr; e->d_Phit = nullptr; e->d_Vhit = nullptr;
  e->fr_on = false; e->fr_k = 0; e->fr_kbg = 0; e->fr_round_waves = 0; for (int i = 0; i < 4; ++i) { e->fr_s[i] = nullptr; e->fr_ev[i] = nullptr; } e->d_fr = nullptr; e->d_done = nullptr; e->d_done_off = nullptr; e->done_cap = 0;
  e->h_fr_words = nullptr; e->d_fr_words = nullptr; e->h_busy = nullptr; e->d_busy = nullptr; e->explore_k = 0; e->fr_prev_done = 0; e->fr_since_round = 0; e->fr_given_up = 0; e->fr_prev_recs = 0;
  e->h_xflag = nullptr; e->d_xflag = nullptr; e->split_off = false; e->split_registered = 0; e->xch_launches = 0;
  { const char* fa = getenv("AZHIP_XCH_FAIL_AT"); e->xch_fail_at = fa ? atoll(fa) : 0; }
  e->net_loaded = false; e->running = false; e->prof_on = false; { const char* ug = getenv("AZHIP_GRAPH"); e->use_graphs = ug ? atoi(ug) : 0; }
  e->d_phase = nullptr; e->phase_cap = 0; e->phase_n = 0; e->host_moves = true; e->last_tower[0] = 0; memset(e->tower_hist, 0, sizeof e->tower_hist); e->prof_mask = 0; e->prof_used = 0;
  memset(&e->prof, 0, sizeof e->prof);
  memset(&e->stats, 0, sizeof e->stats);
  int st = [&]() -> int {
    // The engine's own stream gets a hardware queue of its own (a stream made with a CU mask -- here: every CU -- is not folded
    // into the runtime's pool of GPU_MAX_HW_QUEUES shared queues).  Two one-group engines searching side by side (the arena's two
    // players) otherwise overlap or take turns depending on which pooled queue their streams happened to land on (DESIGN.md 6b:
    // 5.6 s against 8.2 s for the same evaluation).  AZHIP_POOLED_QUEUE=1 keeps the runtime's assignment.
    {
      int ncu = 0;
      HIPCHK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device));
      std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
      for (int i = 0; i < ncu; ++i) mask[(size_t)i / 32] |= 1u << (i % 32);
      if (getenv("AZHIP_POOLED_QUEUE") || ncu <= 0 || hipExtStreamCreateWithCUMask(&e->stream, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
        (void)hipGetLastError();
        e->stream = nullptr;
        HIPCHK(hipStreamCreate(&e->stream));
      }
    }
    const int G = c->num_workers;
    DView& v = e->v;
    memset(&v, 0, sizeof v);
    v.G = G;
    v.max_moves = c->max_moves_per_game > 0 ? c->max_moves_per_game : gi.max_plies;
    v.max_depth = gi.max_plies + 1;
    long long cap = c->max_nodes_per_slot;
    if (cap <= 0) {
      // Mancala has no fixed length (MAX_PLIES = 256 bounds the trace); self-play games of the shipped parameters last
      // 30-70 plies, the pool holds 128 plies of new nodes per game between resets (288 GB of HBM pay for the margin);
      // a longer game reports AZ_ERR_CAPACITY and max_nodes_per_slot raises the bound
      long long plies = c->game == AZ_GAME_MANCALA ? 128 : gi.max_plies;
      cap = (long long)c->num_iters_per_turn * plies * std::max(1, c->reset_every);
      if (c->reset_every == 0) cap *= 4;
      cap = std::max<long long>(cap, 1024);
    }
    if (cap > (1LL << 30)) return fail(AZ_ERR_BAD_ARG, "max_nodes_per_slot too large");
    v.cap_nodes = (int)cap;
    int hs = 1024;
    while ((long long)hs * 2 < cap * 3) hs <<= 1;
    v.ht_size = hs;
    // test knobs for events that are rare at the shipped sizes: AZHIP_HT_TAG_BITS narrows the 16-bit tag of a table entry (unequal
    // states then share tags and every probe chain is decided by the exact key compare), AZHIP_HT_EPOCH0 starts the 16-bit table
    // epoch near its wrap (a reset past 0xfffe really clears the table)
    { const char* tb = getenv("AZHIP_HT_TAG_BITS"); const int b = tb ? atoi(tb) : 16; v.tag_mask = b >= 16 ? 0xffffu : b <= 0 ? 0u : ((1u << b) - 1u); }
    { const char* e0 = getenv("AZHIP_HT_EPOCH0"); const long x = e0 ? atol(e0) : 1; v.epoch0 = (uint32_t)(x >= 1 && x < 0xffff ? x : 1); }
    AZCHK(dalloc(e, &v.sr, G)); AZCHK(dalloc(e, &v.game_id, G));
    AZCHK(dalloc(e, &v.move_idx, G)); AZCHK(dalloc(e, &e->d_node_count, G));
    AZCHK(dalloc(e, &v.worker_sim_id, G));
    AZCHK(dalloc(e, &v.eta, (size_t)G * gi.APAD));
    AZCHK(dalloc(e, &v.ht, (size_t)G * hs));
    {
      const size_t pool_bytes = (size_t)G * cap * gi.node_bytes;
      const char* fv = getenv("AZHIP_VMM");                          // 1 / 0 force the mapped-on-demand pool on / off (tests)
      const size_t chunk_nodes = VM_CHUNK / gi.node_bytes;
      const bool pow2 = (chunk_nodes & (chunk_nodes - 1)) == 0 && chunk_nodes * gi.node_bytes == VM_CHUNK;
      const bool want = fv ? atoi(fv) != 0 : pool_bytes > VM_THRESHOLD;
      void* base = nullptr;
      const int rows = (int)((cap + chunk_nodes - 1) / chunk_nodes);
      const size_t vbytes = (size_t)rows * G * VM_CHUNK;
      // a runtime without virtual memory management (or without room in the address space) gets the plain pool
      const bool vm_ok = want && pow2 && (size_t)cap > chunk_nodes && gi.node_bytes > 0 &&
                         hipMemAddressReserve(&base, vbytes, VM_CHUNK, nullptr, 0) == hipSuccess;
      if (want && !vm_ok) (void)hipGetLastError();
      if (vm_ok) {
        e->vm_chunk = VM_CHUNK; e->vm_chunk_nodes = (int)chunk_nodes;
        e->vm_rows = rows;
        e->vm_bytes = vbytes;
        e->vm_base = (char*)base;
        e->vm_budget = ~(size_t)0;                                   // set once every fixed allocation of the engine is made (below)
        v.nodes = e->vm_base;
        int sh = 0; while (((size_t)1 << sh) < chunk_nodes) ++sh;
        v.node_sh = sh; v.node_mask = (uint32_t)chunk_nodes - 1; v.node_row = (size_t)G * VM_CHUNK; v.node_stride = VM_CHUNK;
        e->h_slot_cap.assign(G, 0); e->h_node_count.assign(G, 0);
        AZCHK(dalloc(e, &e->d_slot_cap, G));
        v.slot_cap = e->d_slot_cap;
      } else {
        AZCHK(dalloc(e, &v.nodes, pool_bytes, false));
        v.node_sh = 31; v.node_mask = 0x7fffffffu; v.node_row = 0; v.node_stride = (size_t)cap * gi.node_bytes; v.slot_cap = nullptr;
      }
    }
    AZCHK(dalloc(e, &v.path, (size_t)G * v.max_depth));
    AZCHK(dalloc(e, &v.leaf_env, G));
    AZCHK(dalloc(e, &v.eval_slots, (size_t)2 * G)); v.eval_stride = G;
    AZCHK(dalloc(e, &v.bg_list, (size_t)2 * G)); AZCHK(dalloc(e, &v.bg_cnt, 2 * AZ_MAX_GROUPS));
    AZCHK(dalloc(e, &v.n_eval, 2 * AZ_MAX_GROUPS));
    {
      // side records (state key + Vest, 32 B per node).  Plain pool: dense [G][cap][4].  Mapped-on-demand pool: a second virtual
      // range whose pieces follow the node chunks (DView::keys), so that they cost 32 B per node that EXISTS -- dense they were
      // 26.8 GB of BASELINE configs[3]'s 57 GB engine.  AZHIP_VMM_KEYS=0 keeps the dense array beside a mapped pool.
      const char* fk = getenv("AZHIP_VMM_KEYS");
      const size_t piece = (size_t)e->vm_chunk_nodes * 32;
      void* kbase = nullptr;
      bool mapped_keys = e->vm_rows > 0 && !(fk && atoi(fk) == 0) && piece > 0 && VM_CHUNK % piece == 0;
      size_t kbytes = 0;
      if (mapped_keys) {
        kbytes = ((size_t)e->vm_rows * G * piece + VM_CHUNK - 1) / VM_CHUNK * VM_CHUNK;
        if (hipMemAddressReserve(&kbase, kbytes, VM_CHUNK, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); mapped_keys = false; }
      }
      if (mapped_keys) {
        e->vmk_base = (char*)kbase; e->vmk_bytes = kbytes; e->vmk_piece = piece;
        e->vmk_granule.assign(kbytes / VM_CHUNK, 0);
        v.keys = (unsigned long long*)kbase;
        v.key_row = (size_t)G * piece / 8; v.key_stride = piece / 8;
      } else {
        AZCHK(dalloc(e, &v.keys, (size_t)G * cap * 4, false));
        v.key_row = 0; v.key_stride = (size_t)cap * 4;
        if (e->vm_rows) v.key_row = (size_t)e->vm_chunk_nodes * 4;   // dense array under a mapped pool: idx = (row, offset) again, rows of a slot are contiguous
      }
    }
    hipLaunchKernelGGL(k_slot_records, dim3((G + 255) / 256), dim3(256), 0, e->stream, v, (int)SR_ZERO);   // epoch 1, no root, nothing else
    AZCHK(dalloc(e, &v.Pout, (size_t)std::max(G, 1) * gi.APAD)); AZCHK(dalloc(e, &v.Vout, G));
    {
      // Evaluation cache (tree.h): on for the ResNet oracle -- what an evaluation costs there makes a repeated one worth a look-up.
      // AZHIP_EVAL_CACHE=0 switches it off, =1 forces it on for the exact synthetic oracles as well (tests; never for the rollout
      // oracle, whose answer depends on the simulation's RNG stream, nor under hipGraph replay: the launch number is a kernel argument).
      // AZHIP_EVAL_CACHE_LOG2 = log2 of the number of 64-byte entries (default 24: 1 GB; a BASELINE phase evaluates 2-6 x 10^7 states).
      const char* ce = getenv("AZHIP_EVAL_CACHE");
      const bool on = ce ? atoi(ce) != 0 && c->oracle != AZ_ORACLE_ROLLOUT : c->oracle == AZ_ORACLE_RESNET;
      if (on && !e->use_graphs) {
        // sized from the workload: 8 x slots x simulations per move entries, 2^16 (4 MB) ... 2^24 (1 GB: BASELINE configs[1];
        // a phase there evaluates 2-6 x 10^7 states) -- an 8-slot test engine and the 128-worker arena players no longer hold 1 GB each
        const char* cl = getenv("AZHIP_EVAL_CACHE_LOG2");
        int lg = 16;
        while (lg < 24 && ((long long)1 << lg) < 8LL * G * std::max(1, c->num_iters_per_turn)) ++lg;
        if (cl) lg = atoi(cl);
        lg = std::min(std::max(lg, 4), 28);
        AZCHK(dalloc(e, &e->d_ec, (size_t)1 << lg));                // zeroed: every entry EC_EMPTY
        e->ec_mask = ((uint32_t)1 << lg) - 1u;
        AZCHK(dalloc(e, &e->d_ec_claim, (size_t)2 * G, false));
        HIPCHK(hipMemsetAsync(e->d_ec_claim, 0xff, sizeof(int) * 2 * (size_t)G, e->stream));
        AZCHK(dalloc(e, &e->d_Phit, (size_t)G * gi.APAD)); AZCHK(dalloc(e, &e->d_Vhit, G));
        v.ec = e->d_ec; v.ec_mask = e->ec_mask; v.ec_seq = 0; v.ec_claim = e->d_ec_claim; v.Phit = e->d_Phit; v.Vhit = e->d_Vhit;
      }
    }
    AZCHK(dalloc(e, &v.trace, (size_t)G * v.max_moves)); AZCHK(dalloc(e, &v.grec, G));
    AZCHK(dalloc(e, &v.finished, G)); AZCHK(dalloc(e, &v.err, 1));
    // split tower fallback: one exchange word per slot group (+ one nobody sets for the whole-engine view), the counters of the
    // k_tree launches that stood still meanwhile, and a host-mapped word the kernel raises so that the host notices without a sync
    AZCHK(dalloc(e, &e->d_xerr, AZ_MAX_GROUPS + 1)); AZCHK(dalloc(e, &e->d_skipped, 2 * (AZ_MAX_GROUPS + 1)));
    v.xerr = e->d_xerr + AZ_MAX_GROUPS; v.skipped = e->d_skipped + 2 * AZ_MAX_GROUPS;
    HIPCHK(hipHostMalloc((void**)&e->h_xflag, sizeof(int), hipHostMallocMapped));
    *e->h_xflag = 0;
    HIPCHK(hipHostGetDevicePointer((void**)&e->d_xflag, e->h_xflag, 0));
    HIPCHK(hipHostMalloc((void**)&e->h_nleaf, sizeof(int) * AZ_MAX_GROUPS, hipHostMallocMapped));
    for (int g = 0; g < AZ_MAX_GROUPS; ++g) e->h_nleaf[g] = -1;      // nothing reported yet: launches are priced at their upper bound
    HIPCHK(hipHostGetDevicePointer((void**)&e->d_nleaf, e->h_nleaf, 0));
    HIPCHK(hipHostMalloc((void**)&e->h_needy, sizeof(int) * AZ_MAX_GROUPS, hipHostMallocMapped));
    for (int g = 0; g < AZ_MAX_GROUPS; ++g) e->h_needy[g] = 0;
    HIPCHK(hipHostGetDevicePointer((void**)&e->d_needy, e->h_needy, 0));
    // free-running phases: the device's own bookkeeping (FRState) and two host-mapped progress words
    AZCHK(dalloc(e, &e->d_fr, 1)); AZCHK(dalloc(e, &e->d_bg_stop, 1)); e->bg_seq = 0; e->bg_signal = false;
    HIPCHK(hipHostMalloc((void**)&e->h_busy, sizeof(int) * AZ_MAX_GROUPS, hipHostMallocMapped));
    for (int g = 0; g < AZ_MAX_GROUPS; ++g) e->h_busy[g] = -1;
    HIPCHK(hipHostGetDevicePointer((void**)&e->d_busy, e->h_busy, 0));
    { const char* ek = getenv("AZHIP_EXPLORE_K"); e->explore_k = ek ? atoi(ek) : 8; }
    HIPCHK(hipHostMalloc((void**)&e->h_fr_words, sizeof(int) * 2, hipHostMallocMapped));
    e->h_fr_words[0] = e->h_fr_words[1] = 0;
    HIPCHK(hipHostGetDevicePointer((void**)&e->d_fr_words, e->h_fr_words, 0));

    // staging
    e->io_cap = std::max(G, 4096);
    AZCHK(dalloc(e, &e->d_slots, e->io_cap)); AZCHK(dalloc(e, &e->d_gids, e->io_cap)); AZCHK(dalloc(e, &e->d_roots, e->io_cap));
    AZCHK(dalloc(e, &e->d_moves, e->io_cap)); AZCHK(dalloc(e, &e->d_eta, (size_t)e->io_cap * AZ_MAX_ACTIONS));
    AZCHK(dalloc(e, &e->d_offsets, e->io_cap)); AZCHK(dalloc(e, &e->d_stage, (size_t)G * v.max_moves));
    AZCHK(dalloc(e, &e->d_keys, (size_t)2 * e->io_cap)); AZCHK(dalloc(e, &e->d_actions, e->io_cap));
    AZCHK(dalloc(e, &e->d_next, (size_t)2 * e->io_cap)); AZCHK(dalloc(e, &e->d_term, e->io_cap)); AZCHK(dalloc(e, &e->d_reward, e->io_cap));
    AZCHK(dalloc(e, &e->d_nodebuf, 512));
    AZCHK(dalloc(e, &e->d_visits, (size_t)e->io_cap * (AZ_MAX_ACTIONS + 1)));
    // network buffers
    e->nn_cap = e->io_cap;
    AZCHK(dalloc(e, &e->d_hfeat, (size_t)e->nn_cap * gi.P * std::max(64, c->num_filters), false));
    AZCHK(dalloc(e, &e->d_X, (size_t)e->nn_cap * gi.C * gi.P)); AZCHK(dalloc(e, &e->d_A, (size_t)e->nn_cap * gi.A));
    AZCHK(dalloc(e, &e->d_P, (size_t)e->nn_cap * std::max(16, gi.APAD))); AZCHK(dalloc(e, &e->d_V, e->nn_cap)); AZCHK(dalloc(e, &e->d_Pinv, e->nn_cap));
    AZCHK(dalloc(e, &e->d_tmp_env, e->nn_cap)); AZCHK(dalloc(e, &e->d_iota, e->nn_cap)); AZCHK(dalloc(e, &e->d_ntmp, 1));
    HIPCHK(hipHostMalloc((void**)&e->h_env, sizeof(GEnv) * (size_t)e->nn_cap, hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void**)&e->h_n, sizeof(int), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void**)&e->h_pv, sizeof(float) * (size_t)e->nn_cap * (gi.A + 1), hipHostMallocDefault));
    hipLaunchKernelGGL(k_iota, dim3((e->nn_cap + 255) / 256), dim3(256), 0, e->stream, e->d_iota, e->nn_cap);
    memset(&e->net, 0, sizeof e->net);
    memset(&e->net16, 0, sizeof e->net16);
    { const char* tw = getenv("AZHIP_TOWER"); e->tower_pick = tw ? atoi(tw) : 0; }
    { const char* hd = getenv("AZHIP_HEADS"); e->heads_pick = hd ? atoi(hd) : 0; }
    { const char* ep = getenv("AZHIP_XCH_EPOCH0"); e->xch_epoch = ep ? strtoull(ep, nullptr, 0) : 0; }   // tests: start k_tower16s' launch epoch near its 24-bit wrap
    { hipDeviceProp_t pr; HIPCHK(hipGetDeviceProperties(&pr, c->device)); e->num_cu = pr.multiProcessorCount; }
    AZCHK(net_set_kernel_attrs(e));
    // round-5 experiments on k_tree (off by default, measured in profiles/r5/README.md): AZHIP_TREE_ATOMIC = 1 | 2
    // backs up with no-return atomics (DView::bk_mode), AZHIP_TREE_SORT = 1 orders the slots of a launch by depth at every move step
    { const char* ta = getenv("AZHIP_TREE_ATOMIC"); const int m = ta ? atoi(ta) : 0; v.bk_mode = m == 1 || m == 2 ? m : 0; }
    { const char* ts = getenv("AZHIP_TREE_SORT"); e->tree_sort = ts && atoi(ts) != 0 ? 1 : 0; }
    if (e->tree_sort) { AZCHK(dalloc(e, &e->d_perm, (size_t)2 * G)); AZCHK(dalloc(e, &e->d_perm_prev, (size_t)2 * G)); }
    // slot groups
    int ng = c->batch_size > 0 ? G / c->batch_size : 1;
    if (ng < 1) ng = 1;
    if (ng > AZ_MAX_GROUPS) ng = AZ_MAX_GROUPS;
    while (ng > 1 && G % ng != 0) --ng;
    const int Gh = G / ng;
    // statistics accumulators: one record per k_tree workgroup of every slot group (and of the whole-engine view, used by the hooks)
    const int blk_group = (Gh * gi.APAD + 255) / 256, blk_all = (G * gi.APAD + 255) / 256;
    e->stat_words = (size_t)4 * ((size_t)blk_group * ng + blk_all);
    AZCHK(dalloc(e, &v.stat, e->stat_words));
    long long* stat_base = v.stat;
    v.stat = stat_base + (size_t)4 * blk_group * ng;                 // the whole-engine view's records come after the groups'
    for (int g = 0; g < ng; ++g) {
      DView gv = v;
      gv.stat = stat_base + (size_t)4 * blk_group * g;
      const size_t o = (size_t)g * Gh;
      gv.G = Gh; gv.slot0 = (int)o;
      gv.sr += o; gv.game_id += o; gv.move_idx += o;
      gv.worker_sim_id += o; gv.eta += o * gi.APAD;
      gv.ht += o * hs; gv.nodes += o * v.node_stride; gv.path += o * v.max_depth;
      if (gv.slot_cap) gv.slot_cap += o;
      gv.leaf_env += o; gv.eval_slots += o; gv.bg_list += o; gv.bg_cnt += 2 * g;
      if (gv.ec) { gv.ec_claim += 2 * o; gv.Phit += o * gi.APAD; gv.Vhit += o; }
      gv.xerr = e->d_xerr + g; gv.skipped = e->d_skipped + 2 * g;
      if (e->tree_sort) {                                            // group-relative order, identity until the first move step
        gv.perm = e->d_perm + o;
        hipLaunchKernelGGL(k_iota, dim3((Gh + 255) / 256), dim3(256), 0, e->stream, e->d_perm + o, Gh);
      }
      gv.needy_host = e->d_needy + g;
      gv.nleaf_host = e->d_ec ? e->d_nleaf + g : nullptr;            // only the evaluation cache makes a wave's network batch differ from its active slots
      gv.n_eval += 2 * g; gv.keys += o * v.key_stride; gv.Pout += o * gi.APAD; gv.Vout += o; gv.trace += o * v.max_moves; gv.grec += o; gv.finished += o;
      e->gv[g] = gv;
      if (ng == 1) {
        e->gs[g] = e->gt[g] = e->stream;
        // a free-running phase runs the group's tree kernels under its own network launch: two streams for that (az_selfplay_begin / _end)
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHK(hipStreamCreateWithPriority(&e->fr_s[0], hipStreamNonBlocking, hi));
        HIPCHK(hipStreamCreateWithPriority(&e->fr_s[1], hipStreamNonBlocking, hi));
        HIPCHK(hipStreamCreateWithPriority(&e->fr_s[2], hipStreamNonBlocking, hi));
        HIPCHK(hipStreamCreateWithPriority(&e->fr_s[3], hipStreamNonBlocking, hi));
        for (int i = 0; i < 4; ++i) HIPCHK(hipEventCreateWithFlags(&e->fr_ev[i], hipEventDisableTiming));
      } else {
        // the short latency-bound tree kernels must not queue behind the other group's tower workgroups:
        // they get the high-priority queue, the tower the normal one
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHK(hipStreamCreateWithPriority(&e->gs[g], hipStreamNonBlocking, hi));
        HIPCHK(hipStreamCreateWithPriority(&e->gt[g], hipStreamNonBlocking, lo));
        HIPCHK(hipEventCreateWithFlags(&e->ev_tree[g], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&e->ev_net[g], hipEventDisableTiming));
      }
      AZCHK(dalloc(e, &e->g_hfeat[g], (size_t)Gh * gi.P * std::max(64, c->num_filters), false));
      e->ngroups = g + 1;
    }
    // search parameters
    DParams& p = e->p;
    memset(&p, 0, sizeof p);
    p.gamma = c->gamma; p.cpuct = c->cpuct; p.eps = c->dirichlet_noise_eps; p.alpha = c->dirichlet_noise_alpha;
    p.prior_temp = c->prior_temperature; p.nsims = c->num_iters_per_turn; p.temp_len = c->temperature_len;
    for (int i = 0; i < AZ_SCHED_MAX; ++i) { p.temp_xs[i] = c->temperature_xs[i]; p.temp_ys[i] = c->temperature_ys[i]; }
    p.seed = c->seed; p.oracle = c->oracle; p.reset_every = c->reset_every; p.retire = 0;
    p.flip_p = c->flip_probability;                                  // read by the self-play kernels only (k_move, k_start_games without roots)
    if (e->vm_rows) {
      // Budget of the growing pool: what the device has left AFTER this engine's fixed allocations, minus what is
      // still to come -- the phase buffer of a bounded phase (num_workers x 4 games of move records is the usual order), the
      // replay memory / trainer of the same process -- as a margin of 1/8 of the device, at least 4 GB.  AZHIP_POOL_GB overrides.
      size_t fr = 0, tot = 0;
      HIPCHK(hipMemGetInfo(&fr, &tot));
      const size_t margin = std::max<size_t>(tot / 8, (size_t)4 << 30);
      const char* gb = getenv("AZHIP_POOL_GB");
      e->vm_budget = gb ? (size_t)(atof(gb) * (double)((size_t)1 << 30)) : (fr > margin ? fr - margin : fr / 2);
      AZCHK(vm_grow(e, 1));                                          // the first chunk of every slot
    }
    e->h_finished.resize(G); e->h_grec.resize(G);
    e->prof_pool.resize(2048);
    for (auto& r : e->prof_pool) { HIPCHK(hipEventCreate(&r.a)); HIPCHK(hipEventCreate(&r.b)); }
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipGetLastError());
    return AZ_OK;
  }();
  if (st != AZ_OK) { std::string keep = g_err; az_engine_destroy(e); g_err = keep; return st; }
  *out = e;
  return AZ_OK;
}


extern "C" int az_engine_device_bytes(az_engine* e, int64_t* bytes) {
  if (!e || !bytes) return fail(AZ_ERR_BAD_ARG, "NULL");
  *bytes = (int64_t)e->alloc_bytes + (int64_t)sizeof(az_move_rec) * e->phase_cap + (int64_t)e->vm_mapped;
  return AZ_OK;
}
extern "C" int az_engine_release_phase(az_engine* e) {
  ENGINE(e);
  if (e->running) return fail(AZ_ERR_STATE, "self-play in progress");
  AZCHK(sync_all(e));
  if (e->d_phase) HIPCHK(hipFree(e->d_phase));
  e->d_phase = nullptr; e->phase_cap = 0; e->phase_n = 0; e->ph_games.clear(); e->ph_off.clear();
  return AZ_OK;
}

extern "C" int az_device_info(az_engine* e, char* name, int32_t name_cap, int32_t* num_cu, int64_t* hbm_bytes) {
  ENGINE(e);
  hipDeviceProp_t pr;
  HIPCHK(hipGetDeviceProperties(&pr, e->device));
  if (name && name_cap > 0) { snprintf(name, (size_t)name_cap, "%s (%s)", pr.name, pr.gcnArchName); }
  if (num_cu) *num_cu = pr.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)pr.totalGlobalMem;
  return AZ_OK;
}

// ------------------------------------------------------------------------------- game plugin
extern "C" int az_game_num_actions(int game, int32_t* n) {
  GameInfo gi;
  if (!n || !game_info(game, &gi)) return fail(AZ_ERR_BAD_ARG, "bad game id or NULL");
  *n = gi.A;
  return AZ_OK;
}
extern "C" int az_game_state_dim(int game, int32_t* w, int32_t* h, int32_t* c) {
  GameInfo gi;
  if (!w || !h || !c || !game_info(game, &gi)) return fail(AZ_ERR_BAD_ARG, "bad game id or NULL");
  *w = gi.W; *h = gi.H; *c = gi.C;
  return AZ_OK;
}
extern "C" int az_game_init_key(int game, uint64_t key[2]) {
  if (!key) return fail(AZ_ERR_BAD_ARG, "NULL");
  DISPATCH_GAME(game, { GEnv g = Gm::init(); key[0] = g.a; key[1] = g.b; });
  return AZ_OK;
}
extern "C" int az_game_encode(az_engine* e, const uint64_t* keys, int32_t n, float* X, float* A) {
  ENGINE(e);
  if (n < 0 || (n > 0 && (!keys || !X || !A))) return fail(AZ_ERR_BAD_ARG, "NULL buffer");
  const GameInfo& gi = e->gi;
  for (int off = 0; off < n; off += e->io_cap) {
    int m = std::min(e->io_cap, n - off);
    HIPCHK(hipMemcpyAsync(e->d_keys, keys + 2 * (size_t)off, sizeof(uint64_t) * 2 * m, hipMemcpyHostToDevice, e->stream));
    DISPATCH_GAME(e->cfg.game, hipLaunchKernelGGL((k_game_encode<Gm>), dim3((m + 255) / 256), dim3(256), 0, e->stream, e->d_keys, m, e->d_X, e->d_A));
    HIPCHK(hipMemcpyAsync(X + (size_t)off * gi.C * gi.P, e->d_X, sizeof(float) * (size_t)m * gi.C * gi.P, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(A + (size_t)off * gi.A, e->d_A, sizeof(float) * (size_t)m * gi.A, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  HIPCHK(hipGetLastError());
  return AZ_OK;
}
extern "C" int az_game_play(az_engine* e, const uint64_t* keys, const int32_t* actions, int32_t n,
                            uint64_t* next_keys, int8_t* terminated, float* white_reward) {
  ENGINE(e);
  if (n < 0 || (n > 0 && (!keys || !actions || !next_keys || !terminated || !white_reward))) return fail(AZ_ERR_BAD_ARG, "NULL buffer");
  for (int i = 0; i < n; ++i) if (actions[i] < -1 || actions[i] >= e->gi.A) return fail(AZ_ERR_BAD_ARG, "action %d out of range at %d", actions[i], i);
  for (int off = 0; off < n; off += e->io_cap) {
    int m = std::min(e->io_cap, n - off);
    HIPCHK(hipMemcpyAsync(e->d_keys, keys + 2 * (size_t)off, sizeof(uint64_t) * 2 * m, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->d_actions, actions + off, sizeof(int) * m, hipMemcpyHostToDevice, e->stream));
    DISPATCH_GAME(e->cfg.game, hipLaunchKernelGGL((k_game_play<Gm>), dim3((m + 255) / 256), dim3(256), 0, e->stream, e->d_keys, e->d_actions, m, e->d_next, e->d_term, e->d_reward));
    HIPCHK(hipMemcpyAsync(next_keys + 2 * (size_t)off, e->d_next, sizeof(uint64_t) * 2 * m, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(terminated + off, e->d_term, (size_t)m, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(white_reward + off, e->d_reward, sizeof(float) * m, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  HIPCHK(hipGetLastError());
  return AZ_OK;
}

// ------------------------------------------------------------------------------- network
extern "C" int az_net_num_params(const az_engine* e, int64_t* n) {
  if (!e || !n) return fail(AZ_ERR_BAD_ARG, "NULL");
  *n = (int64_t)net_nparams(e->gi, e->cfg);
  return AZ_OK;
}

// scale = gamma / sqrtf(var + eps), shift = fma(bias - mean, scale, beta)  (fp32 contract)
static void bn_fold(const float* bias, const float* bn, int n, float* scale, float* shift) {
  const float *g = bn, *be = bn + n, *mu = bn + 2 * n, *var = bn + 3 * n;
  for (int i = 0; i < n; ++i) {
    scale[i] = g[i] / sqrtf(var[i] + 1e-5f);
    shift[i] = az_fmaf(bias[i] - mu[i], scale[i], be[i]);
  }
}
// Flux conv weight W[i + k*(j + k*(ci + Cin*co))] -> MFMA B-fragment order.  Tap t = (dy+1)*3 +
// (dx+1) reads W[i = 1-dx, j = 1-dy] (true convolution, flipped kernel).
static void pack_conv(const float* Wt, int ksz, int Cin, int Cout, int CoutPad, float* dst /* [ntap][CoutPad/32][Cin/8][64][4] */) {
  const int ntap = ksz * ksz, NT = CoutPad / 32, JQ = Cin / 8, half = Cin / 2;
  for (int t = 0; t < ntap; ++t) {
    int dy = ksz == 3 ? t / 3 - 1 : 0, dx = ksz == 3 ? t % 3 - 1 : 0;
    int wi = ksz == 3 ? 1 - dx : 0, wj = ksz == 3 ? 1 - dy : 0;
    for (int n = 0; n < NT; ++n) for (int jq = 0; jq < JQ; ++jq) for (int l = 0; l < 64; ++l) for (int q = 0; q < 4; ++q) {
      int ci = (l >> 5) * half + jq * 4 + q, co = n * 32 + (l & 31);
      float val = co < Cout ? Wt[(size_t)wi + (size_t)ksz * (wj + (size_t)ksz * (ci + (size_t)Cin * co))] : 0.0f;
      dst[((((size_t)t * NT + n) * JQ + jq) * 64 + l) * 4 + q] = val;
    }
  }
}

static int ec_empty(az_engine* e, bool wipe = false);
extern "C" int az_net_set_params(az_engine* e, const float* blob, int64_t n) {
  ENGINE(e);
  if (e->cfg.oracle != AZ_ORACLE_RESNET) return fail(AZ_ERR_STATE, "engine was created without the ResNet oracle");
  if (e->running) return fail(AZ_ERR_STATE, "self-play in progress");
  const GameInfo& gi = e->gi;
  const az_engine_cfg& c = e->cfg;
  if (!blob || n != (int64_t)net_nparams(gi, c)) return fail(AZ_ERR_BAD_ARG, "parameter blob has %lld values, expected %zu", (long long)n, net_nparams(gi, c));
  const int F = c.num_filters, npf = c.num_policy_head_filters, nvf = c.num_value_head_filters, P = gi.P, C = gi.C, A = gi.A;
  const int HF = F, L = gi.APAD, nb = c.num_blocks;   // head features padded to the trunk width
  e->blob.assign(blob, blob + n);
  const float* w = blob;
  // stem fragments: k = t*C + ci (tap t = (dy+1)*3 + (dx+1) reads W[1-dx, 1-dy]), K padded to 2*K2,
  // lane l supplies k = (l>>5)*K2 + j for MFMA j
  const int K2 = (9 * C + 1) / 2;
  std::vector<float> stem_w((size_t)(F / 32) * K2 * 64, 0.0f), stem_ss(2 * F);
  for (int nt = 0; nt < F / 32; ++nt) for (int j = 0; j < K2; ++j) for (int l = 0; l < 64; ++l) {
    int k = (l >> 5) * K2 + j, co = nt * 32 + (l & 31);
    if (k >= 9 * C) continue;
    int t = k / C, ci = k % C, dy = t / 3 - 1, dx = t % 3 - 1, wi = 1 - dx, wj = 1 - dy;
    stem_w[((size_t)nt * K2 + j) * 64 + l] = w[wi + 3 * (wj + 3 * (ci + (size_t)C * co))];
  }
  bn_fold(w + 9 * C * F, w + 9 * C * F + F, F, stem_ss.data(), stem_ss.data() + F);
  w += (size_t)9 * C * F + 5 * F;
  const size_t layer_f = (size_t)9 * (F / 32) * (F / 8) * 64 * 4;
  std::vector<float> conv_w(layer_f * 2 * nb + 4), conv_ss((size_t)2 * nb * 2 * F + 4);
  for (int l = 0; l < 2 * nb; ++l) {
    pack_conv(w, 3, F, F, F, conv_w.data() + layer_f * l);
    bn_fold(w + (size_t)9 * F * F, w + (size_t)9 * F * F + F, F, conv_ss.data() + (size_t)l * 2 * F, conv_ss.data() + (size_t)l * 2 * F + F);
    w += (size_t)9 * F * F + 5 * F;
  }
  // k_tower16 fragments (16x16x4 MFMA): lane l of step s = 4 sq + q supplies channel c = (g&1)*32 + 2s + (g>>1),
  // g = l >> 4, for output column ct*16 + (l & 15)
  std::vector<float> c16_w(4), s16_w(4);
  const int CT = F / 16;                     // column tiles = waves = float4 of B per tap and lane
  {
    c16_w.assign((size_t)2 * nb * 9 * CT * CT * 64 * 4 + 4, 0.0f);
    const float* wc = blob + (size_t)9 * C * F + 5 * F;
    for (int l = 0; l < 2 * nb; ++l) {
      for (int t = 0; t < 9; ++t) {
        int dy = t / 3 - 1, dx = t % 3 - 1, wi = 1 - dx, wj = 1 - dy;
        for (int ct = 0; ct < CT; ++ct) for (int sq = 0; sq < CT; ++sq) for (int ln = 0; ln < 64; ++ln) for (int q = 0; q < 4; ++q) {
          int s = 4 * sq + q, g = ln >> 4, ci = (g & 1) * (F / 2) + 2 * s + (g >> 1), co = ct * 16 + (ln & 15);
          c16_w[(((((size_t)l * 9 + t) * CT + ct) * CT + sq) * 64 + ln) * 4 + q] = wc[wi + 3 * (wj + 3 * (ci + (size_t)F * co))];
        }
      }
      wc += (size_t)9 * F * F + 5 * F;
    }
    const int KK = 9 * C, K2s = (KK + 1) / 2, NS = (2 * K2s + 3) / 4;
    s16_w.assign((size_t)CT * NS * 64, 0.0f);
    for (int ct = 0; ct < CT; ++ct) for (int s = 0; s < NS; ++s) for (int ln = 0; ln < 64; ++ln) {
      int p = 4 * s + (ln >> 4), k = (p & 1) * K2s + (p >> 1), co = ct * 16 + (ln & 15);
      if (k >= KK || p >= 2 * K2s) continue;
      int t = k / C, ci = k % C, dy = t / 3 - 1, dx = t % 3 - 1, wi = 1 - dx, wj = 1 - dy;
      s16_w[((size_t)ct * NS + s) * 64 + ln] = blob[wi + 3 * (wj + 3 * (ci + (size_t)C * co))];
    }
  }
  // k_tower16b fragments (bf16, 16x16x32 MFMA): lane l of k step ks supplies input channels ks*32 + (l >> 4)*8 + e, e = 0..7,
  // for output column ct*16 + (l & 15); values rounded to nearest even
  auto to_bf16 = [](float f) -> uint16_t {
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  };
  std::vector<uint16_t> c16b_w(8), h16b_w(8);
  const int KSB = F / 32;
  if (c.net_bf16) {
    c16b_w.assign((size_t)2 * nb * 9 * CT * KSB * 64 * 8 + 8, 0);
    const float* wc = blob + (size_t)9 * C * F + 5 * F;
    for (int l = 0; l < 2 * nb; ++l) {
      for (int t = 0; t < 9; ++t) {
        int dy = t / 3 - 1, dx = t % 3 - 1, wi = 1 - dx, wj = 1 - dy;
        for (int ct = 0; ct < CT; ++ct) for (int ks = 0; ks < KSB; ++ks) for (int ln = 0; ln < 64; ++ln) for (int el = 0; el < 8; ++el) {
          int ci = ks * 32 + (ln >> 4) * 8 + el, co = ct * 16 + (ln & 15);
          c16b_w[(((((size_t)l * 9 + t) * CT + ct) * KSB + ks) * 64 + ln) * 8 + el] = to_bf16(wc[wi + 3 * (wj + 3 * (ci + (size_t)F * co))]);
        }
      }
      wc += (size_t)9 * F * F + 5 * F;
    }
  }
  // heads: concatenate the two 1x1 convolutions along the output channel
  std::vector<float> hw((size_t)F * HF, 0.0f), hb(HF, 0.0f), hbn((size_t)4 * HF, 0.0f), head_w((size_t)(HF / 32) * (F / 8) * 64 * 4), head_ss(2 * HF);
  for (int i = 0; i < HF; ++i) { hbn[i] = 0.0f; hbn[3 * HF + i] = 1.0f; }   // padded channels: gamma 0, var 1
  const float* pw = w; const float* pb = pw + (size_t)F * npf; const float* pbn = pb + npf;
  const float* pdw = pbn + 4 * npf; const float* pdb = pdw + (size_t)A * P * npf;
  const float* vw = pdb + A; const float* vb = vw + (size_t)F * nvf; const float* vbn = vb + nvf;
  const float* vdw = vbn + 4 * nvf; const float* vdb = vdw + (size_t)F * P * nvf;
  const float* v2w = vdb + F; const float* v2b = v2w + F;
  for (int co = 0; co < npf; ++co) {
    for (int ci = 0; ci < F; ++ci) hw[ci + (size_t)F * co] = pw[ci + (size_t)F * co];
    hb[co] = pb[co];
    for (int k = 0; k < 4; ++k) hbn[(size_t)k * HF + co] = pbn[(size_t)k * npf + co];
  }
  for (int co = 0; co < nvf; ++co) {
    for (int ci = 0; ci < F; ++ci) hw[ci + (size_t)F * (npf + co)] = vw[ci + (size_t)F * co];
    hb[npf + co] = vb[co];
    for (int k = 0; k < 4; ++k) hbn[(size_t)k * HF + npf + co] = vbn[(size_t)k * nvf + co];
  }
  pack_conv(hw.data(), 1, F, HF, HF, head_w.data());
  std::vector<float> h16_w((size_t)CT * CT * 64 * 4, 0.0f);
  for (int ct = 0; ct < CT; ++ct) for (int sq = 0; sq < CT; ++sq) for (int ln = 0; ln < 64; ++ln) for (int q = 0; q < 4; ++q) {
    int s = 4 * sq + q, g = ln >> 4, ci = (g & 1) * (F / 2) + 2 * s + (g >> 1), co = ct * 16 + (ln & 15);
    h16_w[((((size_t)ct) * CT + sq) * 64 + ln) * 4 + q] = hw[ci + (size_t)F * co];
  }
  if (c.net_bf16) {
    h16b_w.assign((size_t)CT * KSB * 64 * 8, 0);
    for (int ct = 0; ct < CT; ++ct) for (int ks = 0; ks < KSB; ++ks) for (int ln = 0; ln < 64; ++ln) for (int el = 0; el < 8; ++el) {
      int ci = ks * 32 + (ln >> 4) * 8 + el, co = ct * 16 + (ln & 15);
      h16b_w[((((size_t)ct) * KSB + ks) * 64 + ln) * 8 + el] = to_bf16(hw[ci + (size_t)F * co]);
    }
  }
  bn_fold(hb.data(), hbn.data(), HF, head_ss.data(), head_ss.data() + HF);
  // dense layers, k-major with k = p*nf + f; Flux Dense W[out + nout*(p + P*f)]
  std::vector<float> pol_w((size_t)P * npf * L, 0.0f), pol_b(L, 0.0f), val_w((size_t)P * nvf * F), val_b(F), val2_w(F);
  for (int q = 0; q < P; ++q) for (int f = 0; f < npf; ++f) for (int a = 0; a < A; ++a)
    pol_w[(size_t)(q * npf + f) * L + a] = pdw[a + (size_t)A * (q + (size_t)P * f)];
  for (int a = 0; a < A; ++a) pol_b[a] = pdb[a];
  for (int q = 0; q < P; ++q) for (int f = 0; f < nvf; ++f) for (int o = 0; o < F; ++o)
    val_w[(size_t)(q * nvf + f) * F + o] = vdw[o + (size_t)F * (q + (size_t)P * f)];
  for (int o = 0; o < F; ++o) { val_b[o] = vdb[o]; val2_w[o] = v2w[o]; }
  // MFMA dense-head fragments: per MFMA pair i (k = 4i..4i+3) and lane: (W[4i+h][o], W[4i+2+h][o])
  const bool hd_ok = (npf % 4 == 0) && (nvf % 4 == 0);
  const int NVT = F / 32;
  std::vector<float> hd_w(4);
  if (hd_ok) {
    const size_t vsteps = (size_t)P * nvf / 4, psteps = (size_t)P * npf / 4;
    const int NPT = (A + 31) / 32;                                  // 32-column policy tiles (1 for the device games, 3 for 82 actions)
    hd_w.assign((NVT * vsteps + NPT * psteps) * 64 * 2, 0.0f);
    for (int t = 0; t < NVT; ++t) for (size_t i = 0; i < vsteps; ++i) for (int l = 0; l < 64; ++l) {
      int hh = l >> 5, o = t * 32 + (l & 31);
      float* d = &hd_w[((t * vsteps + i) * 64 + l) * 2];
      d[0] = val_w[(4 * i + hh) * F + o];
      d[1] = val_w[(4 * i + 2 + hh) * F + o];
    }
    for (int pt = 0; pt < NPT; ++pt) for (size_t i = 0; i < psteps; ++i) for (int l = 0; l < 64; ++l) {
      int hh = l >> 5, o = pt * 32 + (l & 31);
      float* d = &hd_w[((NVT * vsteps + pt * psteps + i) * 64 + l) * 2];
      d[0] = o < A ? pol_w[(4 * i + hh) * L + o] : 0.0f;
      d[1] = o < A ? pol_w[(4 * i + 2 + hh) * L + o] : 0.0f;
    }
  }
  // k_heads16 fragments (16-board tiles, 32 head filters): per 16-k block j and lane one float4, element s = W[16j + 4s + g][16 tile + (lane & 15)]
  const bool hd16_ok = npf == 32 && nvf == 32;
  std::vector<float> hd16_w(4);
  if (hd16_ok) {
    const int NV16 = F / 16, NP16 = (A + 15) / 16;
    const size_t NB = (size_t)2 * P;
    hd16_w.assign((size_t)(NV16 + NP16) * NB * 64 * 4, 0.0f);
    for (int t = 0; t < NV16 + NP16; ++t) for (size_t j = 0; j < NB; ++j) for (int l = 0; l < 64; ++l) for (int s4 = 0; s4 < 4; ++s4) {
      const size_t k = 16 * j + 4 * s4 + (l >> 4);
      const int o = (t < NV16 ? t : t - NV16) * 16 + (l & 15);
      hd16_w[((t * NB + j) * 64 + l) * 4 + s4] = t < NV16 ? val_w[k * F + o] : (o < A ? pol_w[k * L + o] : 0.0f);
    }
  }
  AZCHK(sync_all(e));
  drop_wave_graphs(e);                                             // they hold the old parameter pointers
  for (void* q : e->net_allocs) (void)hipFree(q);
  e->net_allocs.clear();
  e->net_loaded = false;
  auto up = [&](const std::vector<float>& h, const float** d) -> int {
    float* q = nullptr;
    AZCHK(dalloc(e, &q, h.size(), false));
    e->allocs.pop_back();
    e->net_allocs.push_back(q);
    HIPCHK(hipMemcpyAsync(q, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice, e->stream));
    *d = q;
    return AZ_OK;
  };
  NetDev nd;
  memset(&nd, 0, sizeof nd);
  nd.nblocks = nb; nd.F = F; nd.npf = npf; nd.nvf = nvf; nd.HF = HF;
  const float* tmp = nullptr;
  AZCHK(up(stem_w, &nd.stem_w)); AZCHK(up(stem_ss, &nd.stem_ss));
  AZCHK(up(conv_w, &tmp)); nd.conv_w = (const float4*)tmp;
  AZCHK(up(conv_ss, &nd.conv_ss));
  AZCHK(up(head_w, &tmp)); nd.head_w = (const float4*)tmp;
  AZCHK(up(head_ss, &nd.head_ss));
  AZCHK(up(pol_w, &nd.pol_w)); AZCHK(up(pol_b, &nd.pol_b)); AZCHK(up(val_w, &nd.val_w)); AZCHK(up(val_b, &nd.val_b)); AZCHK(up(val2_w, &nd.val2_w));
  nd.val2_b = *v2b;
  AZCHK(up(hd_w, &tmp)); nd.hd_w = (const float2*)tmp;
  nd.hd_ok = hd_ok ? 1 : 0;
  AZCHK(up(hd16_w, &tmp)); nd.hd16_w = (const float4*)tmp;
  nd.hd16_ok = hd16_ok ? 1 : 0;
  Net16Dev n16;
  memset(&n16, 0, sizeof n16);
  {
    n16.nblocks = nb;
    AZCHK(up(s16_w, &n16.stem_w)); n16.stem_ss = nd.stem_ss;
    AZCHK(up(c16_w, &tmp)); n16.conv_w = (const float4*)tmp; n16.conv_ss = nd.conv_ss;
    AZCHK(up(h16_w, &tmp)); n16.head_w = (const float4*)tmp; n16.head_ss = nd.head_ss;
    for (int k = 0; k < 3; ++k) n16.geo[k] = e->d_geo[k];
    n16.geo[3] = e->d_geo[5];
    n16.geo[4] = e->d_geo[6];
  }
  e->net16 = n16;
  {
    Net16bDev nbd;
    memset(&nbd, 0, sizeof nbd);
    nbd.nblocks = nb; nbd.stem_w = n16.stem_w; nbd.stem_ss = n16.stem_ss; nbd.conv_ss = n16.conv_ss; nbd.head_ss = n16.head_ss;
    for (int k = 0; k < 3; ++k) nbd.geo[k] = e->d_geo[k];
    nbd.geo[2] = e->d_geo[3];                                        // bf16 tower: [2] = the 22-tile geometry (8 Connect-Four boards per workgroup)
    if (c.net_bf16) {
      auto up16 = [&](const std::vector<uint16_t>& h, const bf16x8v** d) -> int {
        uint16_t* q = nullptr;
        AZCHK(dalloc(e, &q, h.size(), false));
        e->allocs.pop_back();
        e->net_allocs.push_back(q);
        HIPCHK(hipMemcpyAsync(q, h.data(), h.size() * sizeof(uint16_t), hipMemcpyHostToDevice, e->stream));
        *d = (const bf16x8v*)q;
        return AZ_OK;
      };
      AZCHK(up16(c16b_w, &nbd.conv_w)); AZCHK(up16(h16b_w, &nbd.head_w));
    }
    e->net16b = nbd;
  }
  HIPCHK(hipStreamSynchronize(e->stream));
  e->net = nd;
  e->net_loaded = true;
  AZCHK(ec_empty(e));                                                // the answers of the old network are gone with it
  return AZ_OK;
}
extern "C" int az_net_last_kernel(const az_engine* e, char* name, int32_t cap) {
  if (!e || !name || cap < 1) return fail(AZ_ERR_BAD_ARG, "NULL");
  snprintf(name, (size_t)cap, "%s", e->last_tower);
  return AZ_OK;
}
extern "C" int az_net_get_params(const az_engine* e, float* blob, int64_t n) {
  if (!e || !blob) return fail(AZ_ERR_BAD_ARG, "NULL");
  if (!e->net_loaded) return fail(AZ_ERR_STATE, "no parameters loaded");
  if (n != (int64_t)e->blob.size()) return fail(AZ_ERR_BAD_ARG, "blob size mismatch");
  memcpy(blob, e->blob.data(), sizeof(float) * (size_t)n);
  return AZ_OK;
}

// The network seam (one launch at a time on the engine's stream): did the split tower of the launch just enqueued give up?  If
// so the split is switched off for this engine, the error is cleared and the caller launches again (k_tower16<NT = 3>: the same
// bits).  Costs a stream synchronisation, only while this engine has ever used the split tower.
static bool split_gave_up(az_engine* e) {
  if (!e->xch_epoch || e->split_off) return false;
  if (hipStreamSynchronize(e->stream) != hipSuccess) return false;
  if (!*(volatile int*)e->h_xflag) return false;
  int code = 0;
  if (hipMemcpy(&code, e->v.err, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess || code != DERR_EXCHANGE) return false;
  code = 0;
  (void)hipMemcpy(e->v.err, &code, sizeof(int), hipMemcpyHostToDevice);
  *e->h_xflag = 0;
  e->split_off = true;
  e->stats.tower_fallbacks++;
  return true;
}

extern "C" int az_net_forward(az_engine* e, const float* X, const float* A, int32_t N, float* P, float* V, float* Pinv) {
  ENGINE(e);
  if (!e->net_loaded) return fail(AZ_ERR_STATE, "az_net_set_params has not been called");
  if (N < 0 || (N > 0 && (!X || !A || !P || !V))) return fail(AZ_ERR_BAD_ARG, "NULL buffer");
  const GameInfo& gi = e->gi;
  const size_t xs = (size_t)gi.C * gi.P;
  for (int off = 0; off < N; off += e->nn_cap) {
    int m = std::min(e->nn_cap, N - off);
    HIPCHK(hipMemcpyAsync(e->d_X, X + xs * off, sizeof(float) * xs * m, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->d_A, A + (size_t)gi.A * off, sizeof(float) * gi.A * m, hipMemcpyHostToDevice, e->stream));
    AZCHK(net_launch(e, e->stream, true, e->d_hfeat, nullptr, nullptr, nullptr, m, e->d_X, e->d_A, e->d_P, e->d_V, e->d_Pinv, gi.A));
    if (split_gave_up(e)) AZCHK(net_launch(e, e->stream, true, e->d_hfeat, nullptr, nullptr, nullptr, m, e->d_X, e->d_A, e->d_P, e->d_V, e->d_Pinv, gi.A));
    HIPCHK(hipMemcpyAsync(P + (size_t)gi.A * off, e->d_P, sizeof(float) * gi.A * m, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(V + off, e->d_V, sizeof(float) * m, hipMemcpyDeviceToHost, e->stream));
    if (Pinv) HIPCHK(hipMemcpyAsync(Pinv + off, e->d_Pinv, sizeof(float) * m, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  HIPCHK(hipGetLastError());
  if (e->xch_epoch) AZCHK(check_device_error(e));                  // k_tower16s: a bounded wait can give up (DERR_EXCHANGE)
  return AZ_OK;
}

// Network.evaluate_batch on device-twin states: P [n][A] (masked, normalised), V [n]
static int evaluate_envs(az_engine* e, const std::vector<GEnv>& envs, std::vector<float>& P, std::vector<float>& V) {
  const GameInfo& gi = e->gi;
  const int N = (int)envs.size();
  P.assign((size_t)N * gi.A, 0.f); V.assign(N, 0.f);
  for (int off = 0; off < N; off += e->nn_cap) {
    int m = std::min(e->nn_cap, N - off);
    HIPCHK(hipMemcpyAsync(e->d_tmp_env, envs.data() + off, sizeof(GEnv) * m, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->d_ntmp, &m, sizeof(int), hipMemcpyHostToDevice, e->stream));
    AZCHK(net_launch(e, e->stream, false, e->d_hfeat, e->d_tmp_env, e->d_iota, e->d_ntmp, m, nullptr, nullptr, e->d_P, e->d_V, nullptr, gi.A));
    if (split_gave_up(e)) AZCHK(net_launch(e, e->stream, false, e->d_hfeat, e->d_tmp_env, e->d_iota, e->d_ntmp, m, nullptr, nullptr, e->d_P, e->d_V, nullptr, gi.A));
    HIPCHK(hipMemcpyAsync(P.data() + (size_t)gi.A * off, e->d_P, sizeof(float) * gi.A * m, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(V.data() + off, e->d_V, sizeof(float) * m, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  HIPCHK(hipGetLastError());
  if (e->xch_epoch) AZCHK(check_device_error(e));                  // k_tower16s: a bounded wait can give up (DERR_EXCHANGE)
  return AZ_OK;
}

extern "C" int az_net_evaluate_keys(az_engine* e, const uint64_t* keys, int32_t N, float* P, float* V) {
  ENGINE(e);
  if (!e->net_loaded) return fail(AZ_ERR_STATE, "az_net_set_params has not been called");
  if (N < 0 || (N > 0 && (!keys || !P || !V))) return fail(AZ_ERR_BAD_ARG, "NULL buffer");
  const GameInfo& gi = e->gi;
  for (int off = 0; off < N; off += e->nn_cap) {
    const int m = std::min(e->nn_cap, N - off);
    GEnv* envs = e->h_env;                                          // pinned: the copies below are real asynchronous DMA
    DISPATCH_GAME(e->cfg.game, { for (int i = 0; i < m; ++i) envs[i] = Gm::from_key(keys[2 * (size_t)(off + i)], keys[2 * (size_t)(off + i) + 1]); });
    *e->h_n = m;
    float* hP = e->h_pv; float* hV = e->h_pv + (size_t)gi.A * e->nn_cap;
    HIPCHK(hipMemcpyAsync(e->d_tmp_env, envs, sizeof(GEnv) * m, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->d_ntmp, e->h_n, sizeof(int), hipMemcpyHostToDevice, e->stream));
    AZCHK(net_launch(e, e->stream, false, e->d_hfeat, e->d_tmp_env, e->d_iota, e->d_ntmp, m, nullptr, nullptr, e->d_P, e->d_V, nullptr, gi.A));
    if (split_gave_up(e)) AZCHK(net_launch(e, e->stream, false, e->d_hfeat, e->d_tmp_env, e->d_iota, e->d_ntmp, m, nullptr, nullptr, e->d_P, e->d_V, nullptr, gi.A));
    HIPCHK(hipMemcpyAsync(hP, e->d_P, sizeof(float) * gi.A * m, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(hV, e->d_V, sizeof(float) * m, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    memcpy(P + (size_t)gi.A * off, hP, sizeof(float) * gi.A * m);
    memcpy(V + off, hV, sizeof(float) * m);
  }
  HIPCHK(hipGetLastError());
  if (e->xch_epoch) AZCHK(check_device_error(e));                  // k_tower16s: a bounded wait can give up (DERR_EXCHANGE)
  return AZ_OK;
}

// ------------------------------------------------------------------------------- search waves
// One wave = one run_simulation! for every active slot of every group: k_tree (the previous wave's expand + backup,
// this wave's select + leaf gathering) -> oracle.  Nothing is read back by the host.  The simulation a wave starts is
// completed by the group's next k_tree launch: the next wave's, or flush_pending's.
// Evaluation cache: every k_tree launch of the engine gets a number (claims carry it, tree.h).  30 bits of it are kept in an entry:
// long before they wrap the table is emptied and the count starts again.
// Empties the table.  wipe = false (new parameters): every claim made so far falls below the floor (DView::ec_floor) -- nothing is
// written but the slots' claim words; wipe = true (the claim numbers are about to wrap): the entries are really cleared.
static int ec_empty(az_engine* e, bool wipe) {
  if (!e->d_ec) return AZ_OK;
  AZCHK(sync_all(e));
  if (wipe) HIPCHK(hipMemsetAsync(e->d_ec, 0, sizeof(ECEnt) * ((size_t)e->ec_mask + 1), e->stream));
  HIPCHK(hipMemsetAsync(e->d_ec_claim, 0xff, sizeof(int) * 2 * (size_t)e->v.G, e->stream));   // no slot holds an entry any more
  HIPCHK(hipStreamSynchronize(e->stream));
  if (wipe) e->ec_seq = 0;
  e->v.ec_floor = e->ec_seq;
  for (int g = 0; g < e->ngroups; ++g) e->gv[g].ec_floor = e->ec_seq;
  return AZ_OK;
}
static int ec_next_launch(az_engine* e, DView* v) {
  if (!e->d_ec) return AZ_OK;
  if (e->ec_seq >= (1u << 29)) AZCHK(ec_empty(e, true));
  v->ec_seq = ++e->ec_seq;
  return AZ_OK;
}
// sim_idx: index of this simulation within the current explore! (keys the rollout oracle's RNG stream)
template <class Gm> static int wave_group(az_engine* e, int g, uint32_t sim_idx) {
  constexpr int L = Gm::APAD;
  AZCHK(ec_next_launch(e, &e->gv[g]));
  const DView& v = e->gv[g];
  hipStream_t st = e->gs[g], sn = e->gt[g];
  const bool split = st != sn;
  const int G = v.G;
  const int gb = (G * L + 255) / 256;
  const int par = (e->wave_par[g] ^= 1);
  e->stats.slot_launches += e->group_active[g];                     // (the host's count: as of its last look in a free-running phase)
  LAUNCH_ON(e, st, AZ_K_SELECT, G, (k_tree<Gm>), gb, 256, 0, v, e->p, (e->pending[g] || e->fr_on) ? 1 : 0, 1, par);
  e->pending[g] = true;
  // One slot group, free-running: the wave's launch, the tower and the heads follow each other on ONE stream (no event between them: a
  // dependency across streams costs ~10 us, three of them per wave were 3 % of it); the move step and the background search go to a second
  // stream, behind an event recorded here, and the next wave's launch waits for them.
  hipStream_t s2 = st;
  e->bg_signal = false;
  const bool side = e->fr_on && !split && e->cfg.oracle == AZ_ORACLE_RESNET && e->fr_s[1] && st == e->fr_s[0];
  // (the stop word of the background search: without it a background launch of fr_kbg simulations outlasts a SHORT tower and the next wave
  // waits for it -- Mancala, 8192 slots: 19.8 instead of 39.7 M sims/s; AZHIP_BG_STOP=0 switches it off)
  static const bool bg_stop_on = !(getenv("AZHIP_BG_STOP") && atoi(getenv("AZHIP_BG_STOP")) == 0);
  if (side) { s2 = e->fr_s[1]; HIPCHK(hipEventRecord(e->fr_ev[0], st)); ++e->bg_seq; e->bg_signal = bg_stop_on && e->fr_kbg > 0; }
  if (e->cfg.oracle == AZ_ORACLE_RESNET) {
    AZCHK(net_wave(e, g, split, e->group_active[g]));
  } else {
    LAUNCH_ON(e, st, AZ_K_SYNTH, G, (k_synth_oracle<Gm>), (G + 255) / 256, 256, 0, v, e->p, sim_idx, par);
  }
  if (e->fr_on) {
    // Under the wave's network launch (the tree stream has not been made to wait for it yet): the slots whose explore! is complete play
    // their move (and take their next game if it was the last one), then every slot that has no question pending -- it moved just now,
    // or the wave's launch left it without one -- searches on in the background; what it finds joins the NEXT wave's batch.
    FRArgs fa;
    fa.st = e->d_fr; fa.done = e->d_done; fa.done_off = e->d_done_off;
    fa.recs = e->d_phase ? e->d_phase : e->d_stage; fa.recs_cap = e->d_phase ? (long long)e->phase_cap : (long long)e->v.G * e->v.max_moves;
    fa.done_cap = e->done_cap; fa.total_games = e->total_games; fa.first_game_id = (uint32_t)e->first_game_id; fa.group = g; fa.host_words = e->d_fr_words;
    static const int bg_prio = getenv("AZHIP_BG_PRIO") ? atoi(getenv("AZHIP_BG_PRIO")) : 0;   // A/B aid: 3 = the background launches keep the wave launches' priority
    // (side streams: the move step and the background search touch different slots -- explore! complete / still searching -- and run
    // side by side: one serial move under a busy tower takes 0.4 ms, which the background search would otherwise wait out)
    hipStream_t s3 = side ? e->fr_s[2] : s2;
    if (side) { HIPCHK(hipStreamWaitEvent(s2, e->fr_ev[0], 0)); HIPCHK(hipStreamWaitEvent(s3, e->fr_ev[0], 0)); }
    { DView mv = v; mv.low_prio = bg_prio ? 0 : 1; LAUNCH_ON(e, s3, AZ_K_MOVE, G, (k_move_fr<Gm>), (G + 255) / 256, 256, 0, mv, e->p, fa); }
    if (e->fr_kbg > 0) {
      AZCHK(ec_next_launch(e, &e->gv[g]));
      DView bv = e->gv[g];
      bv.run_k = e->fr_kbg; bv.low_prio = bg_prio ? 0 : 1;
      if (side && e->bg_signal) { bv.bg_stop = e->d_bg_stop; bv.bg_seq = e->bg_seq; }   // ... until the wave's tower has run (net_impl.h sets the word)
      LAUNCH_ON(e, s2, AZ_K_EXPAND, G, (k_tree<Gm>), gb, 256, 0, bv, e->p, 0, 1, par ^ 1);
    }
    if (side) {
      HIPCHK(hipEventRecord(e->fr_ev[1], s2)); HIPCHK(hipStreamWaitEvent(st, e->fr_ev[1], 0));
      HIPCHK(hipEventRecord(e->fr_ev[2], s3)); HIPCHK(hipStreamWaitEvent(st, e->fr_ev[2], 0));
    }   // the next wave's launch finds the slots' state settled
    else if (split && e->cfg.oracle == AZ_ORACLE_RESNET) HIPCHK(hipStreamWaitEvent(st, e->ev_net[g], 0));   // the next wave's launch needs this wave's answers
  }
  return AZ_OK;
}
// A split tower (k_tower16s) whose exchange gave up -- the partner workgroup was not co-resident: another engine, a trainer or
// another process holds the CUs -- must not cost the phase.  The kernel raises a host-mapped word; every k_tree of
// that slot group launched since has stood still and been counted (tree.h), every split tower has returned at once.  Here: wait
// for the device, switch the split off for this engine, evaluate the pending leaves of every group again without it (all tower
// forms give the same bits, so this is idempotent) and launch the waves that stood still once more.  Returns with the engine in
// the state the caller believes it is in.
template <class Gm> static int recover_split(az_engine* e) {
  AZCHK(sync_all(e));
  *e->h_xflag = 0;
  int xerr[AZ_MAX_GROUPS + 1], skipped[2 * (AZ_MAX_GROUPS + 1)];
  HIPCHK(hipMemcpy(xerr, e->d_xerr, sizeof xerr, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(skipped, e->d_skipped, sizeof skipped, hipMemcpyDeviceToHost));
  HIPCHK(hipMemsetAsync(e->d_xerr, 0, sizeof xerr, e->stream));
  HIPCHK(hipMemsetAsync(e->d_skipped, 0, sizeof skipped, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));                           // before anything is launched on the groups' own streams
  bool any = false;
  for (int g = 0; g < e->ngroups; ++g) any = any || xerr[g] == DERR_EXCHANGE;
  if (!any) return AZ_OK;                                            // raised by the network seam's launch: split_gave_up handles that one
  e->split_off = true;
  e->stats.tower_fallbacks++;
  for (int g = 0; g < e->ngroups; ++g) {
    e->wave_par[g] ^= (skipped[2 * g] & 1);                          // back to the leaf counter of the last wave that really ran
    if (skipped[2 * g + 1]) e->pending[g] = true;                    // its expand + backup did not happen
  }
  if (e->cfg.oracle == AZ_ORACLE_RESNET)
    for (int g = 0; g < e->ngroups; ++g)
      if (e->pending[g] && e->group_active[g] > 0) {
        AZCHK(net_wave(e, g, e->gs[g] != e->gt[g], e->group_active[g]));
        if (e->fr_on && e->gs[g] != e->gt[g]) HIPCHK(hipStreamWaitEvent(e->gs[g], e->ev_net[g], 0));   // (a free-running wave_group waits at its end; this launch has none)
      }
  for (int g = 0; g < e->ngroups; ++g)
    for (int i = 0; i < skipped[2 * g]; ++i) AZCHK(wave_group<Gm>(e, g, 0));
  return AZ_OK;
}
template <class Gm> static int wave(az_engine* e, int ngroups_active, uint32_t sim_idx) {
  if (*(volatile int*)e->h_xflag) AZCHK(recover_split<Gm>(e));
  for (int g = 0; g < ngroups_active; ++g) {
    if (e->group_active[g] == 0) continue;                         // nothing left to search in this group
    AZCHK(wave_group<Gm>(e, g, sim_idx));
  }
  e->stats.waves++;
  return AZ_OK;
}
// completes the simulations the last wave started (expand + backup only); every group, asynchronous
template <class Gm> static int flush_pending(az_engine* e) {
  constexpr int L = Gm::APAD;
  for (int g = 0; g < e->ngroups; ++g) {
    if (!e->pending[g]) continue;
    AZCHK(ec_next_launch(e, &e->gv[g]));
    const DView& v = e->gv[g];
    LAUNCH_ON(e, e->gs[g], AZ_K_EXPAND, v.G, (k_tree<Gm>), (v.G * L + 255) / 256, 256, 0, v, e->p, 1, 0, e->wave_par[g]);
    e->pending[g] = false;
  }
  return AZ_OK;
}

// Two consecutive waves of the single slot group as ONE hipGraph launch (k_tree, tower, heads, k_tree, tower, heads):
// with 32 ... 256 slots a wave is a handful of 10 ... 300-us kernels and the gaps between dependent launches weigh as
// much as the tree kernel itself.  Valid when the group is in steady state (a simulation pending, leaf-counter parity 0),
// profiling is off and the oracle does not depend on the simulation index.
// OPT-IN (AZHIP_GRAPH=1), measured on MI355X / ROCm 7.2: graph launches are 2-8 % SLOWER than the plain stream launches
// (128 slots 5x128: 0.346 vs 0.355 M sims/s; 128 slots 5x64: 0.94 vs 1.00 M; 32 Tic-tac-toe slots: 0.33 vs 0.36 M; 4096
// slots: 4.65 vs 4.71 M) -- the runtime replays a captured stream as the same AQL packets with the same barriers.  What
// removed the launch-bound regime was fusing the four tree kernels into k_tree (32 slots: 0.15 -> 0.36 M sims/s).
static void drop_wave_graphs(az_engine* e) {
  for (auto& kv : e->wave_graphs) (void)hipGraphExecDestroy(kv.second);
  e->wave_graphs.clear();
}
template <class Gm> static int wave_pair_graph(az_engine* e, hipGraphExec_t* out) {
  const int key = e->group_active[0];
  auto it = e->wave_graphs.find(key);
  if (it != e->wave_graphs.end()) { *out = it->second; return AZ_OK; }
  hipGraph_t g = nullptr;
  HIPCHK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
  const int64_t waves0 = e->stats.waves;
  int st = wave<Gm>(e, 1, 0);
  if (st == AZ_OK) st = wave<Gm>(e, 1, 1);
  e->stats.waves = waves0;                                          // nothing ran: the capture only recorded the launches
  const hipError_t ce = hipStreamEndCapture(e->stream, &g);
  if (st != AZ_OK) { if (g) (void)hipGraphDestroy(g); return st; }
  if (ce != hipSuccess) return fail(AZ_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(ce));
  hipGraphExec_t ex = nullptr;
  const hipError_t ie = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (ie != hipSuccess) return fail(AZ_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ie));
  e->wave_graphs[key] = ex;
  *out = ex;
  return AZ_OK;
}
// runs up to `n` waves of the current explore! (all slot groups); returns how many it ran
template <class Gm> static int run_waves(az_engine* e, int nga, int n, uint32_t sim0) {
  int done = 0;
  const bool graphable = e->use_graphs && e->ngroups == 1 && nga == 1 && !e->prof_on && e->cfg.oracle != AZ_ORACLE_ROLLOUT && e->group_active[0] > 0;
  while (done < n) {
    if (graphable && n - done >= 2 && e->pending[0] && e->wave_par[0] == 0) {
      hipGraphExec_t ex = nullptr;
      AZCHK(wave_pair_graph<Gm>(e, &ex));
      HIPCHK(hipGraphLaunch(ex, e->stream));
      e->stats.waves += 2;
      done += 2;
    } else {
      AZCHK(wave<Gm>(e, nga, sim0 + (uint32_t)done));
      done += 1;
    }
  }
  return AZ_OK;
}

// no simulation in flight: pending leaves dropped, both leaf counters of every group zero
static int reset_wave_state(az_engine* e) {
  AZCHK(sync_groups(e));
  hipLaunchKernelGGL(k_slot_records, dim3((e->v.G + 255) / 256), dim3(256), 0, e->stream, e->v, (int)SR_CLEAR_LEAF);
  HIPCHK(hipMemsetAsync(e->v.n_eval, 0, sizeof(int) * 2 * AZ_MAX_GROUPS, e->stream));
  HIPCHK(hipMemsetAsync(e->v.bg_cnt, 0, sizeof(int) * 2 * AZ_MAX_GROUPS, e->stream));
  for (int g = 0; g < AZ_MAX_GROUPS; ++g) { e->pending[g] = false; e->wave_par[g] = 0; }
  return AZ_OK;
}

template <class Gm>
static int start_games(az_engine* e, const std::vector<int>& slots, const std::vector<uint32_t>& gids,
                       const std::vector<GEnv>* roots, int reset_tree, int arm) {
  const int n = (int)slots.size();
  if (!n) return AZ_OK;
  HIPCHK(hipMemcpyAsync(e->d_slots, slots.data(), sizeof(int) * n, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(e->d_gids, gids.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, e->stream));
  if (roots) HIPCHK(hipMemcpyAsync(e->d_roots, roots->data(), sizeof(GEnv) * n, hipMemcpyHostToDevice, e->stream));
  LAUNCH(e, AZ_K_START, n, (k_start_games<Gm>), (n + 255) / 256, 256, 0, e->v, e->p, e->d_slots, e->d_gids, roots ? e->d_roots : nullptr, n, reset_tree);
  if (arm) LAUNCH(e, AZ_K_START, n, (k_arm_noise<Gm>), (n + 255) / 256, 256, 0, e->v, e->p, e->d_slots, (const uint32_t*)nullptr, (const double*)nullptr, n);
  HIPCHK(hipStreamSynchronize(e->stream));   // host vectors go out of scope after return
  return AZ_OK;
}

extern "C" int az_mcts_reset(az_engine* e) {
  ENGINE(e);
  if (e->running) return fail(AZ_ERR_STATE, "self-play in progress");
  const int G = e->v.G;
  HIPCHK(hipMemsetAsync(e->v.ht, 0, sizeof(unsigned long long) * (size_t)G * e->v.ht_size, e->stream));
  hipLaunchKernelGGL(k_slot_records, dim3((G + 255) / 256), dim3(256), 0, e->stream, e->v, (int)SR_RESET_TREE);
  AZCHK(reset_wave_state(e));
  HIPCHK(hipStreamSynchronize(e->stream));
  return AZ_OK;
}

// MCTS.explore! (mcts.jl:239-245) for a list of slots, in three steps so that two engines can be driven
// concurrently (arena): begin (roots + noise), nsims lock-step waves (asynchronous launches), end (sync + checks).
// The other slots keep their trees.
template <class Gm>
static int explore_begin(az_engine* e, const std::vector<int>& slots, const std::vector<GEnv>& roots,
                         const std::vector<uint32_t>& gids, const std::vector<uint32_t>& mv, const double* eta, int* nga) {
  const int n = (int)slots.size();
  *nga = 0;
  if (!n) return AZ_OK;
  split_register(e, e->ngroups);                                     // a search is in progress on this engine (explore_end takes it back)
  int maxslot = 0;
  for (int s : slots) maxslot = std::max(maxslot, s);
  AZCHK(reset_wave_state(e));
  hipLaunchKernelGGL(k_slot_records, dim3((e->v.G + 255) / 256), dim3(256), 0, e->stream, e->v, (int)SR_CLEAR_ACTIVE);
  AZCHK(start_games<Gm>(e, slots, gids, &roots, 0, 0));
  HIPCHK(hipMemcpyAsync(e->d_moves, mv.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, e->stream));
  if (eta) HIPCHK(hipMemcpyAsync(e->d_eta, eta, sizeof(double) * (size_t)n * AZ_MAX_ACTIONS, hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL((k_arm_noise<Gm>), dim3((n + 255) / 256), dim3(256), 0, e->stream, e->v, e->p, e->d_slots, e->d_moves, eta ? e->d_eta : nullptr, n);
  HIPCHK(hipStreamSynchronize(e->stream));
  *nga = std::min(e->ngroups, maxslot / e->gv[0].G + 1);
  for (int g = 0; g < AZ_MAX_GROUPS; ++g) e->group_active[g] = 0;
  for (int sl : slots) e->group_active[sl / e->gv[0].G]++;
  // (round 6) the explore! runs ahead: inside one launch a slot completes every simulation that ends on a terminal state or on a
  // state the evaluation cache answers (tree.h k_tree, run_k > 0 with fr = 0), every slot exactly nsims of them; the host launches
  // waves until the device reports nobody busy (explore_waves).  Not for the rollout oracle (keyed by the wave's index) nor under hipGraph replay.
  const bool ahead = e->explore_k > 1 && e->cfg.oracle != AZ_ORACLE_ROLLOUT && !e->use_graphs && !e->tree_sort;
  for (int g = 0; g < e->ngroups; ++g) { e->gv[g].run_k = ahead ? e->explore_k : 0; e->gv[g].fr = 0; e->gv[g].busy_host = ahead ? e->d_busy + g : nullptr; e->h_busy[g] = -1; }
  return AZ_OK;
}
// the waves of an explore! of `nsims` simulations per slot on `nga` slot groups: nsims lock-step waves, or -- running ahead -- as many
// as it takes until every slot has done its nsims (at most nsims; the device's count of busy slots is looked at every 16 waves)
template <class Gm> static int explore_waves(az_engine* e, int nga, int nsims) {
  if (!e->gv[0].run_k) return run_waves<Gm>(e, nga, nsims, 0);
  for (int i = 0; i < nsims + 1; ++i) {
    AZCHK(wave<Gm>(e, nga, (uint32_t)i));
    if ((i & 15) == 15) {
      AZCHK(sync_groups(e)); HIPCHK(hipStreamSynchronize(e->stream));
      bool busy = false;
      for (int g = 0; g < nga; ++g) busy = busy || (e->group_active[g] > 0 && ((volatile int*)e->h_busy)[g] != 0);
      if (!busy) break;
    }
  }
  return AZ_OK;
}
template <class Gm> static int explore_end(az_engine* e, int nga) {
  struct Unreg { az_engine* e; ~Unreg() { split_register(e, 0); for (int g = 0; g < e->ngroups; ++g) { e->gv[g].run_k = 0; e->gv[g].busy_host = nullptr; } } } unreg{e};
  if (!nga) return AZ_OK;
  AZCHK(flush_pending<Gm>(e));                                     // the last simulation's expand + backup
  AZCHK(sync_groups(e));
  if (e->xch_epoch && !e->split_off) {                              // split towers have run: did one give up? (the word is only valid once the device is idle)
    AZCHK(sync_all(e));
    if (*(volatile int*)e->h_xflag) { AZCHK(recover_split<Gm>(e)); AZCHK(flush_pending<Gm>(e)); AZCHK(sync_all(e)); }
  }
  hipLaunchKernelGGL(k_slot_records, dim3((e->v.G + 255) / 256), dim3(256), 0, e->stream, e->v, (int)SR_CLEAR_ACTIVE);
  return check_device_error(e);
}
template <class Gm>
static int explore_slots(az_engine* e, const std::vector<int>& slots, const std::vector<GEnv>& roots,
                         const std::vector<uint32_t>& gids, const std::vector<uint32_t>& mv, const double* eta, int nsims) {
  int nga = 0;
  AZCHK(explore_begin<Gm>(e, slots, roots, gids, mv, eta, &nga));
  AZCHK(vm_grow(e, nsims + 2));                                    // mapped-on-demand pool: room for this explore! (a no-op for plain pools)
  struct Sims { az_engine* e; int keep; ~Sims() { e->p.nsims = keep; } } sims{e, e->p.nsims};   // the kernel counts a slot's simulations against DParams::nsims
  e->p.nsims = nsims;
  if (nga) AZCHK(explore_waves<Gm>(e, nga, nsims));
  return explore_end<Gm>(e, nga);
}

extern "C" int az_mcts_explore(az_engine* e, const uint64_t* root_keys, int32_t nslots, int32_t nsims,
                               const double* eta, const uint32_t* game_ids, const uint32_t* moves) {
  ENGINE(e);
  if (e->running) return fail(AZ_ERR_STATE, "self-play in progress");
  if (!root_keys || nslots < 1 || nslots > e->v.G) return fail(AZ_ERR_BAD_ARG, "nslots must be in 1..num_workers");
  if (nsims < 1) return fail(AZ_ERR_BAD_ARG, "nsims must be >= 1");
  if (e->cfg.oracle == AZ_ORACLE_RESNET && !e->net_loaded) return fail(AZ_ERR_STATE, "az_net_set_params has not been called");
  std::vector<int> slots(nslots);
  std::vector<uint32_t> gids(nslots), mv(nslots);
  std::vector<GEnv> roots(nslots);
  for (int i = 0; i < nslots; ++i) { slots[i] = i; gids[i] = game_ids ? game_ids[i] : 0; mv[i] = moves ? moves[i] : 0; }
  DISPATCH_GAME(e->cfg.game, {
    for (int i = 0; i < nslots; ++i) {
      roots[i] = Gm::from_key(root_keys[2 * i], root_keys[2 * i + 1]);
      if (roots[i].fin & 1) return fail(AZ_ERR_BAD_ARG, "root state %d is terminal", i);
    }
  });
  DISPATCH_GAME(e->cfg.game, AZCHK(explore_slots<Gm>(e, slots, roots, gids, mv, eta, nsims)));
  return AZ_OK;
}

template <class Gm>
__global__ void k_node_stats(DView v, int slot, unsigned long long ka, unsigned long long kb, char* out) {
  using NL = NodeL<Gm>;
  const uint32_t epoch = v.sr[slot].epoch;
  const unsigned long long hk = az_hash_key(ka, kb);
  const uint32_t H1 = (uint32_t)v.ht_size - 1, tag = (uint32_t)(hk >> 40) & v.tag_mask;
  const unsigned long long* tab = v.ht + (size_t)slot * v.ht_size;
  int* found = (int*)out;
  *found = 0;
  for (uint32_t i = 0; i <= H1; ++i) {
    unsigned long long e = tab[((uint32_t)hk + i) & H1];
    uint32_t idx1 = (uint32_t)e;
    if (!((uint32_t)(e >> 48) == epoch && idx1 != 0)) return;
    if (((uint32_t)(e >> 32) & 0xffff) == tag) {
      const char* nd = node_at<Gm>(v, slot, (int)(idx1 - 1));
      const unsigned long long* k = side_at(v, slot, (int)(idx1 - 1));
      if (k[0] == ka && k[1] == kb) {
        *found = 1;
        *(float*)(out + 4) = __uint_as_float((uint32_t)k[2]);
        for (int b = 0; b < NL::BYTES; ++b) out[16 + b] = nd[b];
        return;
      }
    }
  }
}

extern "C" int az_mcts_node_stats(az_engine* e, int32_t slot, const uint64_t key[2], int32_t* N, double* W, float* P,
                                  float* Vest, uint32_t* mask) {
  ENGINE(e);
  if (slot < 0 || slot >= e->v.G || !key) return fail(AZ_ERR_BAD_ARG, "bad slot or NULL key");
  AZCHK(sync_groups(e));
  DISPATCH_GAME(e->cfg.game, hipLaunchKernelGGL((k_node_stats<Gm>), dim3(1), dim3(1), 0, e->stream, e->v, (int)slot, (unsigned long long)key[0], (unsigned long long)key[1], e->d_nodebuf));
  char buf[512];
  HIPCHK(hipMemcpyAsync(buf, e->d_nodebuf, 512, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  int found; memcpy(&found, buf, 4);
  if (!found) return fail(AZ_ERR_BAD_ARG, "state not in the tree of slot %d", slot);
  const char* nd = buf + 16;
  const int A = e->gi.A;                                            // NodeL::Stat (tree.h): per action { W f64 | N i32 | P f32 }
  for (int a = 0; a < A; ++a) {
    if (W) memcpy(&W[a], nd + 16 * a, 8);
    if (N) memcpy(&N[a], nd + 16 * a + 8, 4);
    if (P) memcpy(&P[a], nd + 16 * a + 12, 4);
  }
  if (Vest) memcpy(Vest, buf + 4, 4);
  if (mask) DISPATCH_GAME(e->cfg.game, *mask = Gm::mask(Gm::from_key(key[0], key[1])));
  return AZ_OK;
}

extern "C" int az_mcts_counters(az_engine* e, int32_t slot, int64_t* ts, int64_t* tt, int64_t* nn) {
  ENGINE(e);
  if (slot < 0 || slot >= e->v.G) return fail(AZ_ERR_BAD_ARG, "bad slot");
  AZCHK(sync_groups(e));
  SlotRec r;
  HIPCHK(hipMemcpyAsync(&r, e->v.sr + slot, sizeof r, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  const long long a = r.tot_sims, b = r.tot_trav; const int c = r.node_count;
  if (ts) *ts = a;
  if (tt) *tt = b;
  if (nn) *nn = c;
  return AZ_OK;
}

// ------------------------------------------------------------------------------- self-play
extern "C" int az_selfplay_begin(az_engine* e, int32_t num_games, int32_t first_game_id) {
  ENGINE(e);
  if (e->running) return fail(AZ_ERR_STATE, "self-play already in progress");
  if (num_games == 0) return fail(AZ_ERR_BAD_ARG, "num_games must be != 0");
  if (e->p.nsims < 2) return fail(AZ_ERR_BAD_ARG, "num_iters_per_turn = 0 (NetworkPlayer) is for az_arena_run only");
  // ids with bit 30 set are replacement games (an id that already had it would be given up at its first overflow, and its RNG
  // streams could collide with a replacement's): ADVICE r4
  if (first_game_id < 0 || (long long)first_game_id + std::max<long long>(num_games, e->v.G) >= (long long)AZ_REPLACEMENT_GAME_BIT)
    return fail(AZ_ERR_BAD_ARG, "game ids must stay below AZ_REPLACEMENT_GAME_BIT (first_game_id %d, num_games %d)", (int)first_game_id, (int)num_games);
  if (e->cfg.flip_probability != 0.0) {
    bool nosym = false;
    DISPATCH_GAME(e->cfg.game, nosym = Gm::NSYM == 0);
    if (nosym) return fail(AZ_ERR_BAD_ARG, "flip_probability != 0 but no symmetries were declared for this game (game.jl:332)");
  }
  if (e->cfg.oracle == AZ_ORACLE_RESNET && !e->net_loaded) return fail(AZ_ERR_STATE, "az_net_set_params has not been called");
  const int G = e->v.G;
  // a fresh player per worker (simulations.jl:217-218): empty trees, zero counters
  AZCHK(reset_wave_state(e));
  hipLaunchKernelGGL(k_slot_records, dim3((G + 255) / 256), dim3(256), 0, e->stream, e->v, (int)(SR_CLEAR_ACTIVE | SR_CLEAR_TOTALS));
  HIPCHK(hipMemsetAsync(e->v.finished, 0, sizeof(int) * G, e->stream));
  HIPCHK(hipMemsetAsync(e->v.worker_sim_id, 0, sizeof(int) * G, e->stream));

  HIPCHK(hipMemsetAsync(e->gv[0].stat, 0, sizeof(long long) * e->stat_words, e->stream));   // gv[0].stat = the base of the accumulator array
  e->total_games = num_games; e->first_game_id = first_game_id; e->next_game = 0; e->games_done = 0; e->wave_in_move = 0;
  e->q_games.clear(); e->q_moves.clear();
  e->ph_games.clear(); e->ph_off.clear(); e->phase_n = 0;
  {
    // the phase's move records also stay on the device (az_memory_push_engine, az_comm_gather_push): a bounded phase only
    const int64_t want = num_games > 0 ? (int64_t)num_games * e->v.max_moves : 0;
    if (want > e->phase_cap || want == 0) { if (e->d_phase) (void)hipFree(e->d_phase); e->d_phase = nullptr; e->phase_cap = 0; }
    if (want > 0 && !e->d_phase) {
      if (want * (int64_t)sizeof(az_move_rec) > (16LL << 30)) return fail(AZ_ERR_CAPACITY, "num_games x max_moves_per_game move records exceed the 16 GB phase buffer: split the phase");
      HIPCHK(hipMalloc((void**)&e->d_phase, sizeof(az_move_rec) * (size_t)want));
      e->phase_cap = want;
    }
  }
  memset(&e->stats, 0, sizeof e->stats);
  e->aborted_ids.clear();
  const int n0 = num_games < 0 ? G : std::min(G, (int)num_games);
  std::vector<int> slots(n0);
  std::vector<uint32_t> gids(n0);
  for (int i = 0; i < n0; ++i) { slots[i] = i; gids[i] = (uint32_t)(first_game_id + e->next_game++); }
  DISPATCH_GAME(e->cfg.game, AZCHK(start_games<Gm>(e, slots, gids, nullptr, 1, 1)));
  AZCHK(vm_grow(e, e->p.nsims + 2));                                 // every slot can take its first explore!
  e->active_slots = n0;
  for (int g = 0; g < AZ_MAX_GROUPS; ++g) e->group_active[g] = 0;
  for (int i = 0; i < n0; ++i) e->group_active[i / e->gv[0].G]++;
  {
    // free-running or lock step (az_engine_cfg.lock_step; AZHIP_FREE_RUN=0|1 overrides).  Always lock step: the rollout oracle (its
    // answer is keyed by the simulation's index in the wave sequence), hipGraph replay (fixed launch sequences), the depth-ordering experiment
    const char* fr = getenv("AZHIP_FREE_RUN");
    const bool want = fr ? atoi(fr) != 0 : e->cfg.lock_step == 0;
    const bool on = want && e->cfg.oracle != AZ_ORACLE_ROLLOUT && !e->use_graphs && !e->tree_sort;
    const char* rk = getenv("AZHIP_RUN_K"); const char* rw = getenv("AZHIP_FR_ROUND");
    const char* rb = getenv("AZHIP_RUN_KBG");
    // wave launch: up to 3 simulations per slot (it is on the wave's critical path: 35 / 50 / 62 / 75 us for 1 / 2 / 3 / 4 at 4096 slots);
    // background launch: one slot group -- up to 32 more, or until the tower has run (the stop word); several groups -- a few (it runs on
    // the group's tree stream, ahead of the next wave's launch).  profiles/r6/README.md has the sweeps.
    e->fr_k = std::max(1, rk ? atoi(rk) : 3);
    e->fr_kbg = std::max(0, rb ? atoi(rb) : (e->ngroups == 1 ? 32 : 8));
    e->fr_round_waves = std::max(1, rw ? atoi(rw) : 128);
    if (on) {
      const int want_cap = num_games > 0 ? num_games : G;
      if (want_cap > e->done_cap) {
        if (e->d_done) (void)hipFree(e->d_done);
        if (e->d_done_off) (void)hipFree(e->d_done_off);
        e->d_done = nullptr; e->d_done_off = nullptr; e->done_cap = 0;
        HIPCHK(hipMalloc((void**)&e->d_done, sizeof(az_game_rec) * (size_t)want_cap));
        HIPCHK(hipMalloc((void**)&e->d_done_off, sizeof(long long) * (size_t)want_cap));
        e->done_cap = want_cap;
      }
      if (num_games < 0) e->done_cap = std::min(e->done_cap, G);     // an unbounded phase's staging area holds G games' records
      FRState fs;
      memset(&fs, 0, sizeof fs);
      fs.next_game = e->next_game;
      for (int g = 0; g < AZ_MAX_GROUPS; ++g) fs.active[g] = e->group_active[g];
      HIPCHK(hipMemcpyAsync(e->d_fr, &fs, sizeof fs, hipMemcpyHostToDevice, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
      e->h_fr_words[0] = 0; e->h_fr_words[1] = n0;
      e->fr_prev_done = 0; e->fr_prev_recs = 0; e->fr_since_round = 0; e->fr_given_up = 0;
    }
    for (int g = 0; g < e->ngroups; ++g) { e->gv[g].run_k = on ? e->fr_k : 0; e->gv[g].fr = on ? 1 : 0; e->gv[g].fr_active = on ? &e->d_fr->active[g] : nullptr; }
    if (on && e->ngroups == 1 && e->cfg.oracle == AZ_ORACLE_RESNET) {   // tree kernels under the group's own network launch: two streams (ev_* with them)
      HIPCHK(hipStreamSynchronize(e->stream));
      e->gs[0] = e->gt[0] = e->fr_s[0];                                // (the second stream, fr_s[1], carries the move step and the background search: wave_group)
    }
    e->fr_on = on;
  }
  split_register(e, e->ngroups);
  e->p.retire = 1;                                                  // an overflowing slot is retired, the phase goes on; set only once
                                                                    // nothing can fail any more (the hooks must never see it: ADVICE r3)
  e->running = true;
  e->t_begin = std::chrono::steady_clock::now();
  return AZ_OK;
}

// another game id to hand out?  An unbounded phase (bench, polling) stops refilling before its ids reach the replacement bit.
static inline bool more_games(const az_engine* e) {
  return e->total_games < 0 ? (long long)e->first_game_id + e->next_game < (long long)AZ_REPLACEMENT_GAME_BIT : e->next_game < e->total_games;
}
// the move step (play.jl:308-313) for every slot, then collection of finished games and refill
template <class Gm> static int move_round(az_engine* e) {
  const int G = e->v.G;
  AZCHK(flush_pending<Gm>(e));                                     // explore! is over: the last simulation's expand + backup
  AZCHK(sync_groups(e));
  if (e->xch_epoch && !e->split_off) {                              // split towers have run: did one give up? (the word is only valid once the device is idle)
    AZCHK(sync_all(e));
    if (*(volatile int*)e->h_xflag) { AZCHK(recover_split<Gm>(e)); AZCHK(flush_pending<Gm>(e)); AZCHK(sync_all(e)); }
  }
  LAUNCH(e, AZ_K_MOVE, G, (k_move<Gm>), (G + 255) / 256, 256, 0, e->v, e->p);
  if (e->tree_sort)                                                // the explore! that just ended says how deep each slot searches now
    for (int g = 0; g < e->ngroups; ++g) {
      const size_t o = (size_t)g * e->gv[0].G;
      hipLaunchKernelGGL(k_depth_order, dim3(1), dim3(1024), 0, e->stream, e->gv[g], e->d_perm_prev + 2 * o, e->d_perm + o, e->d_perm + G + o);
    }
  HIPCHK(hipMemcpyAsync(e->h_finished.data(), e->v.finished, sizeof(int) * G, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(e->h_grec.data(), e->v.grec, sizeof(az_game_rec) * G, hipMemcpyDeviceToHost, e->stream));
  AZCHK(check_device_error(e));   // synchronises
  std::vector<int> fslots, offs, aslots;
  int total = 0;
  for (int s = 0; s < G; ++s) {
    if (e->h_finished[s] == 1) { fslots.push_back(s); offs.push_back(total); total += e->h_grec[s].num_moves; }
    else if (e->h_finished[s] == 2) aslots.push_back(s);             // retired: node pool or move record full
  }
  e->stats.moves += e->active_slots - (int)aslots.size();
  std::vector<int> rslots, aslots_refill;
  std::vector<uint32_t> rgids, agids;
  if (!aslots.empty()) {
    // the retired slots' games are reported as aborted (their ids via az_selfplay_aborted), count as done, and the slots
    // take the next games with an empty tree
    std::vector<uint32_t> gid(G);
    HIPCHK(hipMemcpyAsync(gid.data(), e->v.game_id, sizeof(uint32_t) * G, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int sl : aslots) {
      const int32_t id = (int32_t)gid[sl];
      e->aborted_ids.push_back(id);
      e->stats.aborted_games++;
      e->active_slots--; e->group_active[sl / e->gv[0].G]--;
      if (!(id & AZ_REPLACEMENT_GAME_BIT)) {
        // the phase still owes the caller this game: ONE replacement with its own RNG streams (id | bit 30) takes the slot, so
        // that num_games games come back and every game_simulated() callback a host waits for arrives
        aslots_refill.push_back(sl); agids.push_back((uint32_t)(id | AZ_REPLACEMENT_GAME_BIT));
        e->active_slots++; e->group_active[sl / e->gv[0].G]++;
        continue;
      }
      e->games_done++;                                               // the replacement overflowed too: given up, counted, reported
      if (more_games(e)) {
        aslots_refill.push_back(sl); agids.push_back((uint32_t)(e->first_game_id + e->next_game++));
        e->active_slots++; e->group_active[sl / e->gv[0].G]++;
      }
    }
  }
  if (fslots.empty() && aslots.empty()) return vm_grow(e, e->p.nsims + 2);
  const int nf = (int)fslots.size();
  if (nf) {
    HIPCHK(hipMemcpyAsync(e->d_slots, fslots.data(), sizeof(int) * nf, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->d_offsets, offs.data(), sizeof(int) * nf, hipMemcpyHostToDevice, e->stream));
  }
  // the finished games' records: packed behind the phase's earlier games in HBM (or into the staging area when the
  // phase is unbounded), and from there to the host only if the caller wants host traces
  if (e->d_phase && e->phase_n + total > e->phase_cap) return fail(AZ_ERR_CAPACITY, "phase buffer overflow");
  az_move_rec* dst = e->d_phase ? e->d_phase + e->phase_n : e->d_stage;
  if (nf) hipLaunchKernelGGL(k_gather_traces, dim3(nf), dim3(64), 0, e->stream, e->v, e->d_slots, e->d_offsets, nf, dst);
  const size_t m0 = e->q_moves.size();
  if ((e->host_moves || !e->d_phase) && total) {
    e->q_moves.resize(m0 + total);
    HIPCHK(hipMemcpyAsync(e->q_moves.data() + m0, dst, sizeof(az_move_rec) * total, hipMemcpyDeviceToHost, e->stream));
  }
  HIPCHK(hipMemsetAsync(e->v.finished, 0, sizeof(int) * G, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  for (int i = 0; i < nf; ++i) {
    az_game_rec g = e->h_grec[fslots[i]];
    g.first_move = (int32_t)(m0 + offs[i]);
    if (e->d_phase) { e->ph_games.push_back(g); e->ph_off.push_back(e->phase_n + offs[i]); }
    e->q_games.push_back(g);
    e->games_done++;
    e->stats.games++;
    e->active_slots--;
    e->group_active[fslots[i] / e->gv[0].G]--;
    if (more_games(e)) {      // next id, in slot order (util.jl:181-188)
      rslots.push_back(fslots[i]);
      rgids.push_back((uint32_t)(e->first_game_id + e->next_game++));
      e->active_slots++;
      e->group_active[fslots[i] / e->gv[0].G]++;
    }
  }
  if (e->d_phase) e->phase_n += total;
  AZCHK(start_games<Gm>(e, rslots, rgids, nullptr, 0, 1));
  AZCHK(start_games<Gm>(e, aslots_refill, agids, nullptr, 1, 1));    // a retired slot starts over with an empty tree
  return vm_grow(e, e->p.nsims + 2);                                 // chunks for the next explore! of every slot
}

// Free-running phase: what the host still does, every fr_round_waves waves and at the end of every az_selfplay_step -- wait for the
// device, fetch the games that ended since its last look (game records and, if the caller wants host traces, their move records),
// deal with retired slots (replacement games), back the node-pool chunks the slots will reach before the next look.
template <class Gm> static int fr_round(az_engine* e) {
  const int G = e->v.G;
  e->fr_since_round = 0;
  AZCHK(sync_groups(e));
  if (e->xch_epoch && !e->split_off) {                              // split towers have run: did one give up? (the word is only valid once the device is idle)
    AZCHK(sync_all(e));
    if (*(volatile int*)e->h_xflag) { AZCHK(recover_split<Gm>(e)); AZCHK(sync_all(e)); }
  }
  FRState fs;
  HIPCHK(hipMemcpyAsync(&fs, e->d_fr, sizeof fs, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(e->h_finished.data(), e->v.finished, sizeof(int) * G, hipMemcpyDeviceToHost, e->stream));
  AZCHK(check_device_error(e));   // synchronises
  bool dirty = false;
  const int done = (int)(fs.resv >> 40);
  const long long nrec = (long long)(fs.resv & ((1ULL << 40) - 1));
  const int nd = done - e->fr_prev_done;
  if (nd > 0) {
    e->h_done.resize(nd); e->h_done_off.resize(nd);
    HIPCHK(hipMemcpyAsync(e->h_done.data(), e->d_done + e->fr_prev_done, sizeof(az_game_rec) * nd, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(e->h_done_off.data(), e->d_done_off + e->fr_prev_done, sizeof(long long) * nd, hipMemcpyDeviceToHost, e->stream));
    const size_t m0 = e->q_moves.size();
    const long long nm = nrec - e->fr_prev_recs;
    const bool to_host = (e->host_moves || !e->d_phase) && nm > 0;
    if (to_host) {
      e->q_moves.resize(m0 + (size_t)nm);
      const az_move_rec* src = (e->d_phase ? e->d_phase : e->d_stage) + e->fr_prev_recs;
      HIPCHK(hipMemcpyAsync(e->q_moves.data() + m0, src, sizeof(az_move_rec) * (size_t)nm, hipMemcpyDeviceToHost, e->stream));
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int i = 0; i < nd; ++i) {
      az_game_rec g = e->h_done[i];
      g.first_move = (int32_t)((long long)m0 + e->h_done_off[i] - e->fr_prev_recs);
      if (e->d_phase) { e->ph_games.push_back(g); e->ph_off.push_back(e->h_done_off[i]); }
      e->q_games.push_back(g);
      e->games_done++;
      e->stats.games++;
    }
    if (e->d_phase) e->phase_n = nrec;
  }
  e->fr_prev_done = done; e->fr_prev_recs = nrec;
  e->stats.moves = fs.moves;
  if (e->total_games >= 0) fs.next_game = std::min(fs.next_game, e->total_games);   // slots that asked in vain pushed the counter past the end
  e->next_game = fs.next_game;
  // retired slots (node pool or move records full; the kernels have taken them out of the counts): reported as aborted; the slot
  // plays ONE replacement game, or the next game, with an empty tree -- as in lock step (move_round)
  std::vector<int> aslots, aslots_refill;
  std::vector<uint32_t> agids;
  for (int sl = 0; sl < G; ++sl) if (e->h_finished[sl] == 2) aslots.push_back(sl);
  if (!aslots.empty()) {
    std::vector<uint32_t> gid(G);
    HIPCHK(hipMemcpyAsync(gid.data(), e->v.game_id, sizeof(uint32_t) * G, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int sl : aslots) {
      const int32_t id = (int32_t)gid[sl];
      e->aborted_ids.push_back(id);
      e->stats.aborted_games++;
      if (!(id & AZ_REPLACEMENT_GAME_BIT)) {
        aslots_refill.push_back(sl); agids.push_back((uint32_t)(id | AZ_REPLACEMENT_GAME_BIT));
        fs.active[sl / e->gv[0].G]++;
        continue;
      }
      e->games_done++; e->fr_given_up++;                             // the replacement overflowed too: given up, counted, reported
      if (more_games(e)) {
        aslots_refill.push_back(sl); agids.push_back((uint32_t)(e->first_game_id + e->next_game++));
        fs.active[sl / e->gv[0].G]++;
      }
    }
    fs.next_game = e->next_game;
    HIPCHK(hipMemsetAsync(e->v.finished, 0, sizeof(int) * G, e->stream));
    AZCHK(start_games<Gm>(e, aslots_refill, agids, nullptr, 1, 1));  // a retired slot starts over with an empty tree
    dirty = true;
  }
  if (!e->d_phase && (done > 0 || nrec > 0)) {                      // unbounded phase: the staging area has been drained
    fs.resv = 0; e->fr_prev_done = 0; e->fr_prev_recs = 0; dirty = true;
  }
  if (dirty) {
    HIPCHK(hipMemcpyAsync(e->d_fr, &fs, sizeof fs, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  e->active_slots = 0;
  for (int g = 0; g < AZ_MAX_GROUPS; ++g) { e->group_active[g] = g < e->ngroups ? std::max(0, fs.active[g]) : 0; e->active_slots += e->group_active[g]; }
  e->h_fr_words[0] = e->fr_prev_done; e->h_fr_words[1] = e->active_slots;
  return vm_grow(e, e->fr_round_waves * e->fr_k + 2);               // a slot adds at most run_k nodes per wave
}
template <class Gm> static int fr_step(az_engine* e, int nwaves) {
  for (int w = 0; w < nwaves; ++w) {
    if (e->active_slots == 0) break;
    // the device's own progress words, one wave late and read without synchronising: every game played, or nobody left searching?
    const int seen_done = ((volatile int*)e->h_fr_words)[0], seen_active = ((volatile int*)e->h_fr_words)[1];
    if (seen_active == 0 || (e->total_games > 0 && e->d_phase && seen_done + e->fr_given_up >= e->total_games)) break;
    AZCHK(wave<Gm>(e, e->ngroups, 0));
    if (++e->fr_since_round >= e->fr_round_waves) AZCHK(fr_round<Gm>(e));
  }
  return fr_round<Gm>(e);                                           // the call returns with the device idle and every finished game collectable
}

extern "C" int az_selfplay_step(az_engine* e, int32_t nwaves) {
  ENGINE(e);
  if (!e->running) return fail(AZ_ERR_STATE, "az_selfplay_begin has not been called");
  if (e->fr_on) { DISPATCH_GAME(e->cfg.game, AZCHK(fr_step<Gm>(e, nwaves))); return AZ_OK; }
  for (int w = 0; w < nwaves;) {
    if (e->active_slots == 0) break;
    const int chunk = std::min(nwaves - w, e->p.nsims - e->wave_in_move);      // up to the next move step
    DISPATCH_GAME(e->cfg.game, AZCHK(run_waves<Gm>(e, e->ngroups, chunk, (uint32_t)e->wave_in_move)));
    w += chunk;
    if ((e->wave_in_move += chunk) == e->p.nsims) {
      e->wave_in_move = 0;
      DISPATCH_GAME(e->cfg.game, AZCHK(move_round<Gm>(e)));
    }
  }
  return AZ_OK;
}

extern "C" int az_selfplay_active(az_engine* e, int32_t* n) {
  ENGINE(e);
  if (!n) return fail(AZ_ERR_BAD_ARG, "NULL");
  *n = e->running ? e->active_slots : 0;
  return AZ_OK;
}

extern "C" int az_selfplay_get_stats(az_engine* e, az_selfplay_stats* s) {
  ENGINE(e);
  if (!s) return fail(AZ_ERR_BAD_ARG, "NULL");
  long long st[4] = {0, 0, 0, 0};
  AZCHK(sync_groups(e));
  e->h_stat.resize(e->stat_words);
  HIPCHK(hipMemcpyAsync(e->h_stat.data(), e->gv[0].stat, sizeof(long long) * e->stat_words, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  for (size_t i = 0; i < e->stat_words; ++i) st[i & 3] += e->h_stat[i];
  e->stats.simulations = st[0]; e->stats.nodes_traversed = st[1]; e->stats.leaf_evals = st[2]; e->stats.evals_reused = st[3];
  e->stats.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - e->t_begin).count();
  *s = e->stats;
  return AZ_OK;
}

extern "C" int az_selfplay_collect(az_engine* e, az_trace_buf* out) {
  ENGINE(e);
  if (!out) return fail(AZ_ERR_BAD_ARG, "NULL");
  const int64_t ng = (int64_t)e->q_games.size(), nm = (int64_t)e->q_moves.size();
  if (!e->host_moves && e->d_phase) {                              // device-only phase: game records only, first_move = -1
    if (ng > out->games_cap || (ng && !out->games)) return fail(AZ_ERR_CAPACITY, "trace buffer too small: need %lld games", (long long)ng);
    std::vector<int> ord(ng);
    for (int i = 0; i < ng; ++i) ord[i] = i;
    std::sort(ord.begin(), ord.end(), [&](int a, int b) { return e->q_games[a].game_id < e->q_games[b].game_id; });
    for (int64_t i = 0; i < ng; ++i) { out->games[i] = e->q_games[ord[i]]; out->games[i].first_move = -1; }
    out->num_games = ng; out->num_moves = 0;
    e->q_games.clear();
    return AZ_OK;
  }
  if (ng > out->games_cap || nm > out->moves_cap || (ng && !out->games) || (nm && !out->moves))
    return fail(AZ_ERR_CAPACITY, "trace buffer too small: need %lld games / %lld moves", (long long)ng, (long long)nm);
  // sorted by game id, move records re-packed in that order
  std::vector<int> ord(ng);
  for (int i = 0; i < ng; ++i) ord[i] = i;
  std::sort(ord.begin(), ord.end(), [&](int a, int b) { return e->q_games[a].game_id < e->q_games[b].game_id; });
  int64_t m = 0;
  for (int64_t i = 0; i < ng; ++i) {
    az_game_rec g = e->q_games[ord[i]];
    memcpy(out->moves + m, e->q_moves.data() + g.first_move, sizeof(az_move_rec) * (size_t)g.num_moves);
    g.first_move = (int32_t)m;
    m += g.num_moves;
    out->games[i] = g;
  }
  out->num_games = ng; out->num_moves = nm;
  e->q_games.clear(); e->q_moves.clear();
  return AZ_OK;
}

extern "C" int az_selfplay_aborted(az_engine* e, int32_t* game_ids, int32_t cap, int32_t* n) {
  ENGINE(e);
  if (!n || (cap > 0 && !game_ids)) return fail(AZ_ERR_BAD_ARG, "NULL");
  *n = (int32_t)e->aborted_ids.size();
  if (*n > cap) return cap == 0 ? AZ_OK : fail(AZ_ERR_CAPACITY, "%d aborted games, room for %d", *n, cap);
  for (int i = 0; i < *n; ++i) game_ids[i] = e->aborted_ids[i];
  return AZ_OK;
}

extern "C" int az_selfplay_end(az_engine* e) {
  ENGINE(e);
  e->p.retire = 0;
  split_register(e, 0);
  if (!e->running) return AZ_OK;
  AZCHK(reset_wave_state(e));                                      // an unfinished simulation (stepping form stopped mid-move) is dropped
  hipLaunchKernelGGL(k_slot_records, dim3((e->v.G + 255) / 256), dim3(256), 0, e->stream, e->v, (int)SR_CLEAR_ACTIVE);
  HIPCHK(hipStreamSynchronize(e->stream));
  e->running = false;
  e->active_slots = 0;
  e->fr_on = false;
  for (int g = 0; g < e->ngroups; ++g) { e->gv[g].run_k = 0; e->gv[g].fr = 0; e->gv[g].fr_active = nullptr; }
  if (e->ngroups == 1 && e->gs[0] != e->stream) {                    // back to the engine's one stream
    HIPCHK(hipStreamSynchronize(e->gs[0])); for (int i = 1; i < 4; ++i) HIPCHK(hipStreamSynchronize(e->fr_s[i]));
    e->gs[0] = e->gt[0] = e->stream;
  }
  return AZ_OK;
}

extern "C" int az_selfplay_run(az_engine* e, int32_t num_games, int32_t first_game_id, az_trace_buf* out,
                               az_progress_cb cb, void* user, az_selfplay_stats* stats) {
  ENGINE(e);
  if (num_games < 1 || !out) return fail(AZ_ERR_BAD_ARG, "num_games must be >= 1 and out non-NULL");
  if (out->games_cap < num_games) return fail(AZ_ERR_CAPACITY, "games_cap %lld < num_games %d", (long long)out->games_cap, num_games);
  // out->moves == NULL: device-only phase -- the move records stay in HBM for az_memory_push_engine / az_comm_gather_push,
  // only the game records (56 B per game) come back
  e->host_moves = out->moves != nullptr;
  const int bst = az_selfplay_begin(e, num_games, first_game_id);
  if (bst != AZ_OK) { e->host_moves = true; return bst; }
  int reported = 0;
  int st = AZ_OK;
  while (e->games_done < num_games) {
    st = az_selfplay_step(e, e->fr_on ? e->fr_round_waves : e->p.nsims);
    if (st != AZ_OK) break;
    if (cb) for (; reported < e->games_done; ++reported) cb(user);   // game_simulated()
  }
  if (st == AZ_OK && stats) st = az_selfplay_get_stats(e, stats);
  std::string keep = g_err;
  // on an error (a capacity overflow in one slot, say) the games that did finish are still handed over: `out` then
  // holds fewer than num_games games and the status tells why
  const int cst = az_selfplay_collect(e, out);
  if (st == AZ_OK) { st = cst; keep = g_err; }
  az_selfplay_end(e);
  e->host_moves = true;
  if (st != AZ_OK) g_err = keep;
  return st;
}

// =========================================================================================
// arena: pit_networks (training.jl:130-144) = simulate (simulations.jl:207-244) over
// TwoPlayers(MctsPlayer(contender), MctsPlayer(baseline)) (play.jl:248-282).  The two players
// keep separate trees (one engine each); every ply the slots are split by the player to move,
// each engine explores its share (MCTS.explore! on a slot list), and the host applies
// play_game's loop body (flip, temperature, sample, play!, play.jl:305-313) with the same
// select_action / Gm::play code the device self-play uses.
// =========================================================================================
namespace {
struct ArenaSlot {
  GEnv env;
  uint32_t gid = 0;
  int nmoves = 0, worker_sim_id = 0;
  bool active = false;
  std::vector<az_move_rec> moves;
};
struct KeyHash { size_t operator()(const std::pair<uint64_t, uint64_t>& k) const { return (size_t)az_hash_key(k.first, k.second); } };
}

template <class Gm>
static int root_visits(az_engine* e, const std::vector<int>& slots, const std::vector<GEnv>& roots, std::vector<int>& out) {
  const int n = (int)slots.size();
  out.assign((size_t)n * (AZ_MAX_ACTIONS + 1), 0);
  if (!n) return AZ_OK;
  HIPCHK(hipMemcpyAsync(e->d_slots, slots.data(), sizeof(int) * n, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(e->d_roots, roots.data(), sizeof(GEnv) * n, hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL((k_root_visits<Gm>), dim3((n + 255) / 256), dim3(256), 0, e->stream, e->v, e->d_slots, e->d_roots, n, e->d_visits);
  HIPCHK(hipMemcpyAsync(out.data(), e->d_visits, sizeof(int) * out.size(), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return AZ_OK;
}
static int reset_slots(az_engine* e, const std::vector<int>& slots) {
  const int n = (int)slots.size();
  if (!n) return AZ_OK;
  HIPCHK(hipMemcpyAsync(e->d_slots, slots.data(), sizeof(int) * n, hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(k_reset_slots, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->v, e->d_slots, n);
  HIPCHK(hipStreamSynchronize(e->stream));
  return AZ_OK;
}

template <class Gm>
static int arena_run(az_engine* ec, az_engine* eb, int num_games, int first_game_id, bool alternate, az_trace_buf* out,
                     double* rewards, double* redundancy, az_progress_cb cb, void* user) {
  const int G = std::min(ec->v.G, num_games);
  const double flip_p = ec->cfg.flip_probability;
  if (flip_p != 0.0 && Gm::NSYM == 0) return fail(AZ_ERR_BAD_ARG, "flip_probability != 0 but no symmetries were declared for this game (game.jl:332)");
  AZCHK(az_mcts_reset(ec));                                        // a fresh player per worker (simulations.jl:217-218)
  AZCHK(az_mcts_reset(eb));
  std::vector<ArenaSlot> sl(G);
  int next_game = 0, finished = 0;
  for (int s = 0; s < G; ++s) { sl[s].env = Gm::init(); sl[s].gid = (uint32_t)(first_game_id + next_game++); sl[s].active = true; }
  if (out) { out->num_games = 0; out->num_moves = 0; }
  std::unordered_set<std::pair<uint64_t, uint64_t>, KeyHash> uniq;
  int64_t nstates = 0;
  std::vector<int> slots[2], visits[2], fin;
  std::vector<float> netP[2], netV[2];
  std::vector<GEnv> roots[2];
  std::vector<uint32_t> gids[2], mvs[2];
  az_engine* eng[2] = {ec, eb};
  while (finished < num_games) {
    for (int k = 0; k < 2; ++k) { slots[k].clear(); roots[k].clear(); gids[k].clear(); mvs[k].clear(); }
    for (int s = 0; s < G; ++s) {
      ArenaSlot& a = sl[s];
      if (!a.active) continue;
      if (a.nmoves >= ec->v.max_moves) return fail(AZ_ERR_CAPACITY, "game %u exceeds max_moves_per_game = %d", a.gid, ec->v.max_moves);
      az_move_rec rec;
      memset(&rec, 0, sizeof rec);
      rec.key[0] = a.env.a; rec.key[1] = a.env.b;                  // trace.states[i]: the state BEFORE the flip
      if (flip_p != 0.0) {                                         // play.jl:305-307
        az_rng r = az_rng_make(ec->p.seed, a.gid, (uint32_t)a.nmoves, AZ_RNG_FLIP);
        if (az_rng_f64(&r) < flip_p) {
          int k = (int)(az_rng_f64(&r) * (double)Gm::NSYM);
          if (k >= Gm::NSYM) k = Gm::NSYM - 1;
          a.env = Gm::sym(a.env, k);
          rec.N[AZ_MAX_ACTIONS] = k + 1;
        }
      }
      a.moves.push_back(rec);
      // simulations.jl:221-223: sim_id is 1-based, colors are flipped for odd sim_id
      const bool colors_flipped = alternate && (((int)a.gid - first_game_id + 1) % 2 == 1);
      const int who = (Gm::white_playing(a.env) != colors_flipped) ? 0 : 1;   // 0 contender, 1 baseline
      slots[who].push_back(s); roots[who].push_back(a.env); gids[who].push_back(a.gid); mvs[who].push_back((uint32_t)a.nmoves);
    }
    // think (play.jl:196-206): the two engines' waves are enqueued alternately on their own streams, so the
    // contender's and the baseline's searches overlap on the GPU (each has only part of the workers)
    int nga[2] = {0, 0};
    for (int k = 0; k < 2; ++k) if (eng[k]->p.nsims > 0) { HIPCHK(hipSetDevice(eng[k]->device)); AZCHK(explore_begin<Gm>(eng[k], slots[k], roots[k], gids[k], mvs[k], nullptr, &nga[k])); }
    // (round 6: the explores run ahead -- explore_begin -- and a player is done when its device reports no busy slot: looked at every 16 waves)
    bool done[2] = {nga[0] == 0, nga[1] == 0};
    for (int i = 0; i < std::max(ec->p.nsims, eb->p.nsims) + 1 && !(done[0] && done[1]); ++i) {
      for (int k = 0; k < 2; ++k) if (!done[k]) {
        const bool ahead = eng[k]->gv[0].run_k > 0;
        if (i >= eng[k]->p.nsims + (ahead ? 1 : 0)) { done[k] = true; continue; }
        HIPCHK(hipSetDevice(eng[k]->device)); AZCHK(wave<Gm>(eng[k], nga[k], (uint32_t)i));
      }
      if ((i & 15) == 15)
        for (int k = 0; k < 2; ++k) if (!done[k] && eng[k]->gv[0].run_k > 0) {
          HIPCHK(hipSetDevice(eng[k]->device)); AZCHK(sync_groups(eng[k])); HIPCHK(hipStreamSynchronize(eng[k]->stream));
          bool busy = false;
          for (int g = 0; g < nga[k]; ++g) busy = busy || (eng[k]->group_active[g] > 0 && ((volatile int*)eng[k]->h_busy)[g] != 0);
          done[k] = !busy;
        }
    }
    for (int k = 0; k < 2; ++k) {
      HIPCHK(hipSetDevice(eng[k]->device));
      if (eng[k]->p.nsims > 0) {
        AZCHK(explore_end<Gm>(eng[k], nga[k]));
        AZCHK(root_visits<Gm>(eng[k], slots[k], roots[k], visits[k]));
      } else {
        AZCHK(evaluate_envs(eng[k], roots[k], netP[k], netV[k]));   // NetworkPlayer.think (play.jl:230-235)
      }
    }
    for (int k = 0; k < 2; ++k) for (size_t i = 0; i < slots[k].size(); ++i) {
      ArenaSlot& a = sl[slots[k][i]];
      az_move_rec& rec = a.moves.back();
      const uint32_t m = Gm::mask(a.env);
      int act;
      if (eng[k]->p.nsims > 0) {
        const int* vis = &visits[k][i * (AZ_MAX_ACTIONS + 1)];
        if (!vis[0]) return fail(AZ_ERR_STATE, "root of slot %d missing after explore", slots[k][i]);
        for (int x = 0; x < AZ_MAX_ACTIONS; ++x) rec.N[x] = vis[1 + x];
        act = select_action<Gm>(eng[k]->p, vis + 1, m, (uint32_t)a.nmoves, a.gid);
      } else {
        const float* P = &netP[k][i * (size_t)Gm::A];
        for (int x = 0; x < Gm::A; ++x) memcpy(&rec.N[x], &P[x], 4);      // the policy's Float32 bits
        rec.N[AZ_MAX_ACTIONS] |= 0x100;
        act = select_action_net<Gm>(eng[k]->p, P, m, (uint32_t)a.nmoves, a.gid);
      }
      Gm::play(a.env, act);
      rec.action = act;
      rec.reward = Gm::white_reward(a.env);
      a.nmoves++;
    }
    // end-of-round bookkeeping in worker order (simulations.jl:231-240)
    fin.clear();
    for (int s = 0; s < G; ++s) {
      ArenaSlot& a = sl[s];
      if (!a.active || !(a.env.fin & 1)) continue;
      const int gi = (int)a.gid - first_game_id;
      const bool colors_flipped = alternate && ((gi + 1) % 2 == 1);
      double wr = 0.0, gp = 1.0;                                   // total_reward (trace.jl:45-47)
      for (const az_move_rec& r : a.moves) { wr += gp * (double)r.reward; gp *= ec->p.gamma; }
      if (rewards) rewards[gi] = colors_flipped ? -wr : wr;        // rewards_and_redundancy, simulations.jl:304-307
      for (const az_move_rec& r : a.moves) uniq.insert({r.key[0], r.key[1]});
      uniq.insert({a.env.a, a.env.b});
      nstates += (int64_t)a.moves.size() + 1;
      if (out) {
        if (gi >= out->games_cap || out->num_moves + (int64_t)a.moves.size() > out->moves_cap) return fail(AZ_ERR_CAPACITY, "trace buffer too small");
        az_game_rec& g = out->games[gi];
        memset(&g, 0, sizeof g);
        g.game_id = (int32_t)a.gid; g.slot = s; g.num_moves = a.nmoves; g.first_move = (int32_t)out->num_moves;
        g.final_key[0] = a.env.a; g.final_key[1] = a.env.b;
        memcpy(out->moves + out->num_moves, a.moves.data(), sizeof(az_move_rec) * a.moves.size());
        out->num_moves += (int64_t)a.moves.size();
        out->num_games++;
      }
      a.worker_sim_id++;
      if (ec->p.reset_every > 0 && a.worker_sim_id % ec->p.reset_every == 0) fin.push_back(s);   // reset_player!
      finished++;
      if (cb) cb(user);
      a.moves.clear(); a.nmoves = 0;
      if (next_game < num_games) { a.gid = (uint32_t)(first_game_id + next_game++); a.env = Gm::init(); }
      else a.active = false;
    }
    for (int k = 0; k < 2; ++k) { HIPCHK(hipSetDevice(eng[k]->device)); AZCHK(reset_slots(eng[k], fin)); }
  }
  if (redundancy) *redundancy = nstates ? 1.0 - (double)uniq.size() / (double)nstates : 0.0;   // simulations.jl:296-299
  return AZ_OK;
}

extern "C" int az_arena_run(az_engine* contender, az_engine* baseline, int32_t num_games, int32_t first_game_id,
                            int32_t alternate_colors, az_trace_buf* out, double* rewards, double* redundancy,
                            az_progress_cb cb, void* user) {
  ENGINE(contender);
  ENGINE(baseline);
  if (contender == baseline) return fail(AZ_ERR_BAD_ARG, "contender and baseline must be two engines (two trees per worker, play.jl:248-251)");
  if (contender->running || baseline->running) return fail(AZ_ERR_STATE, "self-play in progress");
  if (contender->cfg.game != baseline->cfg.game) return fail(AZ_ERR_BAD_ARG, "the two engines play different games");
  if (num_games < 1) return fail(AZ_ERR_BAD_ARG, "num_games must be >= 1");
  if (baseline->v.G < std::min(contender->v.G, (int)num_games)) return fail(AZ_ERR_BAD_ARG, "baseline engine has fewer workers (%d) than the arena needs (%d)", baseline->v.G, std::min(contender->v.G, (int)num_games));
  for (az_engine* e : {contender, baseline})
    if (e->cfg.oracle == AZ_ORACLE_RESNET && !e->net_loaded) return fail(AZ_ERR_STATE, "az_net_set_params has not been called");
  if (out && (!out->games || !out->moves)) return fail(AZ_ERR_BAD_ARG, "NULL trace buffers");
  const bool trace = getenv("AZHIP_TRACE_ARENA") != nullptr;      // which tower forms served the evaluation, and who else counted as a split-tower user
  long long h0[2][4];
  const int others = split_streams_on_device(contender->device);
  for (int k = 0; k < 2; ++k) memcpy(h0[k], (k ? baseline : contender)->tower_hist, sizeof h0[k]);
  DISPATCH_GAME(contender->cfg.game, AZCHK(arena_run<Gm>(contender, baseline, num_games, first_game_id, alternate_colors != 0, out, rewards, redundancy, cb, user)));
  if (trace)
    for (int k = 0; k < 2; ++k) {
      const long long* h = (k ? baseline : contender)->tower_hist;
      fprintf(stderr, "azhip arena: %s launches split %lld, 3-tile %lld, packed %lld, other %lld; split-tower streams registered before the call: %d, split_off %d\n",
              k ? "baseline" : "contender", h[0] - h0[k][0], h[1] - h0[k][1], h[2] - h0[k][2], h[3] - h0[k][3], others, (int)(k ? baseline : contender)->split_off);
    }
  return AZ_OK;
}

extern "C" int az_push_trace(const az_move_rec* moves, int32_t n, double gamma, double* z, double* t) {
  if (n < 0 || (n > 0 && (!moves || !z || !t))) return fail(AZ_ERR_BAD_ARG, "NULL buffer");
  double wr = 0.;                                                // memory.jl:74-87
  for (int i = n - 1; i >= 0; --i) {
    wr = gamma * wr + (double)moves[i].reward;
    const bool wp = !(moves[i].key[0] >> 63);
    z[i] = wp ? wr : -wr;
    t[i] = (double)(n - i);
  }
  return AZ_OK;
}


// debug aid (not part of the ABI in azhip.h): the numerics contract evaluated on the device, so that tests can
// compare gfx950 against the host bit for bit (f64 sqrt / div, az_log / az_exp / az_pow, az_expf / az_tanhf)
__global__ void k_debug_math(int op, const double* x, const double* y, int n, double* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double r = 0.0;
  switch (op) {
    case 0: r = __builtin_sqrt(x[i]); break;
    case 1: r = x[i] / y[i]; break;
    case 2: r = az_log(x[i]); break;
    case 3: r = az_exp(x[i]); break;
    case 4: r = az_pow(x[i], y[i]); break;
    case 5: r = (double)az_expf((float)x[i]); break;
    case 6: r = (double)az_tanhf((float)x[i]); break;
    case 7: r = (double)((float)x[i] / (float)y[i]); break;
    case 8: { az_rng g = az_rng_make((uint64_t)x[i], (uint32_t)y[i], 3, AZ_RNG_NOISE); double eta[7]; az_dirichlet(&g, 7, 0.3, eta); r = eta[3]; break; }
  }
  out[i] = r;
}
extern "C" int az_debug_math(az_engine* e, int32_t op, const double* x, const double* y, int32_t n, double* out) {
  ENGINE(e);
  if (n < 0 || (n > 0 && (!x || !y || !out))) return fail(AZ_ERR_BAD_ARG, "NULL buffer");
  if (n == 0) return AZ_OK;
  double *dx = nullptr, *dy = nullptr, *dout = nullptr;
  HIPCHK(hipMalloc(&dx, sizeof(double) * n)); HIPCHK(hipMalloc(&dy, sizeof(double) * n)); HIPCHK(hipMalloc(&dout, sizeof(double) * n));
  HIPCHK(hipMemcpyAsync(dx, x, sizeof(double) * n, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(dy, y, sizeof(double) * n, hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(k_debug_math, dim3((n + 255) / 256), dim3(256), 0, e->stream, (int)op, dx, dy, (int)n, dout);
  HIPCHK(hipMemcpyAsync(out, dout, sizeof(double) * n, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dout);
  return AZ_OK;
}

// ------------------------------------------------------------------------------- profiling
extern "C" int az_prof_enable(az_engine* e, int32_t on) {
  ENGINE(e);
  AZCHK(prof_flush(e));
  e->prof_on = on != 0;
  e->prof_mask = on > 1 ? (on >> 1) : 0;      // on = 1 | (class mask << 1): time only the selected classes
  return AZ_OK;
}
extern "C" int az_prof_reset(az_engine* e) {
  ENGINE(e);
  AZCHK(prof_flush(e));
  memset(&e->prof, 0, sizeof e->prof);
  return AZ_OK;
}
extern "C" int az_prof_get(az_engine* e, az_prof* out) {
  ENGINE(e);
  if (!out) return fail(AZ_ERR_BAD_ARG, "NULL");
  AZCHK(prof_flush(e));
  *out = e->prof;
  return AZ_OK;
}
