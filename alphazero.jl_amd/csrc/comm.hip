// comm.hip -- the multi-GPU exchange of the path, behind the C ABI: RCCL over xGMI, one process per GPU.
//
//   simulate_distributed (src/simulations.jl:252-290): the games are split over the workers, every worker simulates its share
//     with NO communication (az_selfplay_run with global game ids), then the results are fetched and concatenated;
//   self_play_step! (src/training.jl:284-299) pushes every trace into the replay memory (push_trace!, src/memory.jl:74-87);
//   the new network goes out to the workers before the next phase (the closure serialised by Distributed, training.jl:278-282).
//
// Here: az_comm_gather_push = ONE all-gather of the ranks' device-resident move records (64 B per position) + one of
// their game records (56 B per game), straight from the engines' phase buffers into every rank's device buffer, and
// push_trace! of all games in global game-id order into the rank's az_memory -- no record visits the host, only the
// game records come back (they carry the counts).  az_comm_broadcast_params = ncclBroadcast of the parameter blob.
// RCCL is dlopen'ed by path next to the HIP runtime this library uses (a process that also imports PyTorch holds a second,
// bundled RCCL: binding by soname could hand our communicator to that copy's HIP runtime), on first use only: single-GPU
// users never load it.  The unique id travels between the processes by whatever the host has (Julia's Distributed, MPI,
// a file; bench.py uses torch.distributed's gloo store).
//
// Transport seam: AZHIP_RCCL_LIB=<path> makes rc::load() open that library instead of RCCL.  It exists so that the W > 1
// branches below (padding, per-rank offsets, the re-ordering by global game id) can be EXECUTED on a box with one GPU, where
// RCCL itself refuses two ranks per device: tests/rccl_stub/ builds a test-only library with the same six entry points that
// moves the bytes through host shared memory.  The product never sets the variable.
//
// Failure agreement: every local reason to refuse (no phase, wrong device, allocation failure, the root without parameters)
// is turned into a status word that travels with the first all-gather of a call, so all ranks leave together instead of one
// rank returning while the others wait inside ncclAllGather for ever; a failing collective aborts the communicator.
#include "engine.h"

#include <dlfcn.h>

// The part of RCCL's C ABI this file binds (rccl.h, NCCL 2.x ABI: stable enum values), declared here so that building
// libazhip.so does not need the RCCL development headers -- the library is dlopen'ed on first use only.
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclInt64 = 4, ncclFloat32 = 7 } ncclDataType_t;
}

namespace rc {
struct Lib {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;      // optional
  std::string path;                                // what was loaded (az_comm_version)
};
static Lib g;
static int load() {
  if (g.so) return AZ_OK;
  void* so = nullptr;
  Dl_info info;
  const char* forced = getenv("AZHIP_RCCL_LIB");
  if (forced && *forced) {
    // a transport seam for the tests (several ranks on one GPU: tests/rccl_stub), not a plug-in point: only a library that calls
    // itself librccl* is accepted, and it must export the NCCL entry points below like the real one
    const char* base = strrchr(forced, '/');
    base = base ? base + 1 : forced;
    if (strncmp(base, "librccl", 7) != 0) return fail(AZ_ERR_COMM, "AZHIP_RCCL_LIB=%s: not a librccl* library", forced);
    so = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
    if (!so) return fail(AZ_ERR_COMM, "cannot load AZHIP_RCCL_LIB=%s (%s)", forced, dlerror());
  }
  if (!so && dladdr(reinterpret_cast<void*>(&hipGetLastError), &info) && info.dli_fname) {
    std::string p(info.dli_fname);
    const size_t k = p.rfind('/');
    if (k != std::string::npos) so = dlopen((p.substr(0, k) + "/librccl.so").c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!so && k != std::string::npos) so = dlopen((p.substr(0, k) + "/librccl.so.1").c_str(), RTLD_NOW | RTLD_LOCAL);
  }
  if (!so) so = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!so) so = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!so) return fail(AZ_ERR_COMM, "cannot load librccl.so (%s)", dlerror());
  g.GetUniqueId = (decltype(g.GetUniqueId))dlsym(so, "ncclGetUniqueId");
  g.CommInitRank = (decltype(g.CommInitRank))dlsym(so, "ncclCommInitRank");
  g.CommDestroy = (decltype(g.CommDestroy))dlsym(so, "ncclCommDestroy");
  g.CommAbort = (decltype(g.CommAbort))dlsym(so, "ncclCommAbort");
  g.AllGather = (decltype(g.AllGather))dlsym(so, "ncclAllGather");
  g.Broadcast = (decltype(g.Broadcast))dlsym(so, "ncclBroadcast");
  g.GetErrorString = (decltype(g.GetErrorString))dlsym(so, "ncclGetErrorString");
  g.GetVersion = (decltype(g.GetVersion))dlsym(so, "ncclGetVersion");
  if (Dl_info li; g.GetUniqueId && dladdr(reinterpret_cast<void*>(g.GetUniqueId), &li) && li.dli_fname) g.path = li.dli_fname;
  if (!g.GetUniqueId || !g.CommInitRank || !g.CommDestroy || !g.AllGather || !g.Broadcast || !g.GetErrorString)
    return fail(AZ_ERR_COMM, "librccl.so lacks the expected symbols");
  g.so = so;
  return AZ_OK;
}
}  // namespace rc

#define RCCLCHK(x)                                                                                              \
  do {                                                                                                          \
    ncclResult_t _r = (x);                                                                                      \
    if (_r != ncclSuccess) return fail(AZ_ERR_COMM, "%s failed: %s (%s:%d)", #x, rc::g.GetErrorString(_r), __FILE__, __LINE__); \
  } while (0)

struct az_comm {
  ncclComm_t comm;
  int rank, world, device;
  hipStream_t stream;
  // status / count words of the failure agreement, allocated ONCE at az_comm_init: a collective call must not be able to fail
  // locally (an allocation) before it has entered the first all-gather, or the other ranks would wait in it for ever
  long long *d_word_send, *d_word_all;             // [3], [3 * world]
};
static_assert(sizeof(ncclUniqueId) == AZ_COMM_ID_BYTES, "ncclUniqueId size");

extern "C" int az_comm_unique_id(uint8_t* id) {
  if (!id) return fail(AZ_ERR_BAD_ARG, "NULL");
  AZCHK(rc::load());
  ncclUniqueId u;
  RCCLCHK(rc::g.GetUniqueId(&u));
  memcpy(id, &u, sizeof u);
  return AZ_OK;
}

extern "C" int az_comm_version(int32_t* version, char* path, int32_t cap) {
  AZCHK(rc::load());
  int v = 0;
  if (rc::g.GetVersion && rc::g.GetVersion(&v) != ncclSuccess) v = 0;
  if (version) *version = v;
  if (path && cap > 0) { strncpy(path, rc::g.path.c_str(), (size_t)cap - 1); path[cap - 1] = 0; }
  return AZ_OK;
}

extern "C" int az_comm_destroy(az_comm* c) {
  if (!c) return AZ_OK;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)rc::g.CommDestroy(c->comm);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->d_word_send) (void)hipFree(c->d_word_send);
  if (c->d_word_all) (void)hipFree(c->d_word_all);
  delete c;
  return AZ_OK;
}

extern "C" int az_comm_init(int32_t device, int32_t rank, int32_t world, const uint8_t* id, az_comm** out) {
  if (!id || !out) return fail(AZ_ERR_BAD_ARG, "NULL argument");
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return fail(AZ_ERR_BAD_ARG, "rank %d of %d", rank, world);
  AZCHK(rc::load());
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(AZ_ERR_BAD_ARG, "device %d not available (%d visible)", device, ndev);
  HIPCHK(hipSetDevice(device));
  az_comm* c = new (std::nothrow) az_comm();
  if (!c) return fail(AZ_ERR_HIP, "out of host memory");
  c->comm = nullptr; c->rank = rank; c->world = world; c->device = device; c->stream = nullptr;
  c->d_word_send = c->d_word_all = nullptr;
  int st = [&]() -> int {
    HIPCHK(hipStreamCreate(&c->stream));
    HIPCHK(hipMalloc((void**)&c->d_word_send, 3 * sizeof(long long)));
    HIPCHK(hipMalloc((void**)&c->d_word_all, (size_t)3 * world * sizeof(long long)));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    RCCLCHK(rc::g.CommInitRank(&c->comm, world, u, rank));
    return AZ_OK;
  }();
  if (st != AZ_OK) { az_comm_destroy(c); return st; }
  *out = c;
  return AZ_OK;
}

// packs the phase's move records in game-id order: game i of the order -> out[dst[i] ..), from d_phase[src[i] ..)
static __global__ void k_pack_moves(const az_move_rec* __restrict__ phase, const long long* __restrict__ src, const long long* __restrict__ dst,
                                    const int* __restrict__ cnt, int ngames, az_move_rec* __restrict__ out) {
  const int g = blockIdx.x;
  if (g >= ngames) return;
  const uint4* s = (const uint4*)(phase + src[g]);
  uint4* d = (uint4*)(out + dst[g]);
  for (int k = threadIdx.x; k < cnt[g] * 4; k += blockDim.x) d[k] = s[k];
}

// One int64 per rank: AZ_OK or the first local failure.  Every rank learns whether ALL ranks can go on (max |status|).
static int agree(az_comm* c, int local_status, long long* d_send, long long* d_all, int* worst_rank) {
  const int W = c->world;
  long long v = local_status;
  HIPCHK(hipMemcpyAsync(d_send, &v, sizeof v, hipMemcpyHostToDevice, c->stream));
  RCCLCHK(rc::g.AllGather(d_send, d_all, 1, ncclInt64, c->comm, c->stream));
  std::vector<long long> all((size_t)W);
  HIPCHK(hipMemcpyAsync(all.data(), d_all, sizeof(long long) * W, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  *worst_rank = -1;
  for (int r = 0; r < W; ++r) if (all[r] != AZ_OK && *worst_rank < 0) *worst_rank = r;
  return AZ_OK;
}
// a collective failed half way: the other ranks may be inside it -- tear the communicator down so that they fail too
static void abort_comm(az_comm* c) {
  if (c->comm && rc::g.CommAbort) { (void)rc::g.CommAbort(c->comm); c->comm = nullptr; }
}
#define COMM_ALIVE(c) if (!(c)->comm) return fail(AZ_ERR_COMM, "the communicator was aborted by an earlier failed collective")

extern "C" int az_comm_gather_push(az_comm* c, az_engine* e, az_memory* m, double gamma, az_gather_stats* stats) {
  if (!c) return fail(AZ_ERR_BAD_ARG, "NULL communicator");
  COMM_ALIVE(c);
  if (hipSetDevice(c->device) != hipSuccess) { abort_comm(c); return fail(AZ_ERR_HIP, "hipSetDevice(%d) failed; communicator aborted", c->device); }
  // local validation first; its verdict travels with the counts, so that every rank returns together (never one rank
  // leaving while the others sit in ncclAllGather)
  int local = AZ_OK;
  if (!e) local = fail(AZ_ERR_BAD_ARG, "NULL engine");
  else if (e->device != c->device || (m && m->device != c->device)) local = fail(AZ_ERR_BAD_ARG, "engine / memory / communicator are on different devices");
  else if (m && m->game != e->cfg.game) local = fail(AZ_ERR_BAD_ARG, "memory and engine differ in game");
  else if (e->running) local = fail(AZ_ERR_STATE, "self-play in progress");
  else if (!e->d_phase) local = fail(AZ_ERR_STATE, "the engine holds no device-resident phase (az_selfplay_run first)");
  else local = sync_all(e);
  std::string local_msg = local != AZ_OK ? az_last_error() : "";
  const auto t0 = std::chrono::steady_clock::now();
  const int W = c->world;
  const size_t ng = local == AZ_OK ? e->ph_games.size() : 0;
  std::vector<int> ord(ng);
  for (size_t i = 0; i < ng; ++i) ord[i] = (int)i;
  std::sort(ord.begin(), ord.end(), [&](int a, int b) { return e->ph_games[a].game_id < e->ph_games[b].game_id; });
  std::vector<long long> src(ng), dst(ng);
  std::vector<int> cnt(ng);
  std::vector<az_game_rec> gsend(ng);
  long long nm = 0;
  for (size_t i = 0; i < ng; ++i) {
    const az_game_rec& r = e->ph_games[ord[i]];
    src[i] = e->ph_off[ord[i]]; dst[i] = nm; cnt[i] = r.num_moves;
    gsend[i] = r; gsend[i].first_move = (int32_t)nm;
    nm += r.num_moves;
  }
  std::vector<void*> tmp;
  auto cleanup = [&]() { for (void* p : tmp) (void)hipFree(p); };
  bool in_collective = false;
  int st = [&]() -> int {
    // 1. counts and status of every rank
    long long hc[3] = {(long long)ng, nm, (long long)local};
    long long *d_cnt_send = c->d_word_send, *d_cnt_all = c->d_word_all;   // allocated at az_comm_init: nothing can fail before the all-gather
    in_collective = true;
    HIPCHK(hipMemcpyAsync(d_cnt_send, hc, sizeof hc, hipMemcpyHostToDevice, c->stream));
    RCCLCHK(rc::g.AllGather(d_cnt_send, d_cnt_all, 3, ncclInt64, c->comm, c->stream));
    std::vector<long long> all((size_t)3 * W);
    HIPCHK(hipMemcpyAsync(all.data(), d_cnt_all, sizeof(long long) * 3 * W, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    in_collective = false;
    long long maxg = 1, maxm = 1, totg = 0, totm = 0;
    if (local != AZ_OK) return fail(local, "%s", local_msg.c_str());   // a failing rank reports its own reason
    for (int r = 0; r < W; ++r) {
      if (all[3 * r + 2] != AZ_OK) {
        return fail(AZ_ERR_COMM, "rank %d cannot take part in the gather (status %lld); nothing was exchanged", r, all[3 * r + 2]);
      }
      maxg = std::max(maxg, all[3 * r]); maxm = std::max(maxm, all[3 * r + 1]); totg += all[3 * r]; totm += all[3 * r + 1];
    }
    // 2. this rank's records packed in game-id order, then the two all-gathers (equal, padded segments per rank)
    az_move_rec *d_msend = nullptr, *d_mall = nullptr; az_game_rec *d_gsend = nullptr, *d_gall = nullptr; long long *d_src = nullptr, *d_dst = nullptr; int* d_n = nullptr;
    int ast = [&]() -> int {
      AZCHK(mem_alloc(&tmp, &d_msend, (size_t)maxm)); AZCHK(mem_alloc(&tmp, &d_mall, (size_t)maxm * W));
      AZCHK(mem_alloc(&tmp, &d_gsend, (size_t)maxg)); AZCHK(mem_alloc(&tmp, &d_gall, (size_t)maxg * W));
      AZCHK(mem_alloc(&tmp, &d_src, std::max<size_t>(ng, 1))); AZCHK(mem_alloc(&tmp, &d_dst, std::max<size_t>(ng, 1))); AZCHK(mem_alloc(&tmp, &d_n, std::max<size_t>(ng, 1)));
      return AZ_OK;
    }();
    if (W > 1) {                       // the buffers are sized by the largest rank: a rank that cannot allocate them says so first
      std::string amsg = ast != AZ_OK ? az_last_error() : "";
      int bad = -1;
      in_collective = true;
      AZCHK(agree(c, ast, d_cnt_send, d_cnt_all, &bad));
      in_collective = false;
      if (bad >= 0) return bad == c->rank || ast != AZ_OK ? fail(ast != AZ_OK ? ast : AZ_ERR_COMM, "%s", ast != AZ_OK ? amsg.c_str() : "another rank failed")
                                                            : fail(AZ_ERR_COMM, "rank %d could not allocate the gather buffers; nothing was exchanged", bad);
    } else AZCHK(ast);
    in_collective = true;
    if (ng) {
      HIPCHK(hipMemcpyAsync(d_src, src.data(), sizeof(long long) * ng, hipMemcpyHostToDevice, c->stream));
      HIPCHK(hipMemcpyAsync(d_dst, dst.data(), sizeof(long long) * ng, hipMemcpyHostToDevice, c->stream));
      HIPCHK(hipMemcpyAsync(d_n, cnt.data(), sizeof(int) * ng, hipMemcpyHostToDevice, c->stream));
      HIPCHK(hipMemcpyAsync(d_gsend, gsend.data(), sizeof(az_game_rec) * ng, hipMemcpyHostToDevice, c->stream));
      hipLaunchKernelGGL(k_pack_moves, dim3((unsigned)ng), dim3(64), 0, c->stream, e->d_phase, d_src, d_dst, d_n, (int)ng, d_msend);
    }
    RCCLCHK(rc::g.AllGather(d_msend, d_mall, (size_t)maxm * sizeof(az_move_rec), ncclInt8, c->comm, c->stream));
    RCCLCHK(rc::g.AllGather(d_gsend, d_gall, (size_t)maxg * sizeof(az_game_rec), ncclInt8, c->comm, c->stream));
    // 3. the game records (56 B per game) tell where every game's records sit in the gathered buffer
    std::vector<az_game_rec> gall((size_t)maxg * W);
    HIPCHK(hipMemcpyAsync(gall.data(), d_gall, sizeof(az_game_rec) * gall.size(), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipGetLastError());
    in_collective = false;
    const auto t1 = std::chrono::steady_clock::now();
    struct Ref { int32_t id; long long first; int n; };
    std::vector<Ref> refs;
    refs.reserve((size_t)totg);
    for (int r = 0; r < W; ++r)
      for (long long i = 0; i < all[3 * r]; ++i) {
        const az_game_rec& gr = gall[(size_t)r * maxg + i];
        refs.push_back({gr.game_id, (long long)r * maxm + gr.first_move, gr.num_moves});   // rank r's segment starts at r * maxm
      }
    std::sort(refs.begin(), refs.end(), [](const Ref& a, const Ref& b) { return a.id < b.id; });
    if (m) {
      std::vector<long long> first(refs.size());
      std::vector<int> n(refs.size());
      for (size_t i = 0; i < refs.size(); ++i) { first[i] = refs[i].first; n[i] = refs[i].n; }
      AZCHK(memory_push_device(m, d_mall, first, n, gamma));
    }
    if (stats) {
      stats->games = totg; stats->moves = totm;
      stats->bytes = (int64_t)W * (maxm * (long long)sizeof(az_move_rec) + maxg * (long long)sizeof(az_game_rec) + 24);
      stats->gather_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
      stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      // Report.SelfPlay wants mean depth and the largest tree over ALL games (training.jl:293-296), not the rank's own
      long long sims = 0, trav = 0, nodes = 0, repl = 0;
      for (int r = 0; r < W; ++r)
        for (long long i = 0; i < all[3 * r]; ++i) {
          const az_game_rec& gr = gall[(size_t)r * maxg + i];
          repl += (gr.game_id & AZ_REPLACEMENT_GAME_BIT) != 0;
          sims += gr.total_simulations; trav += gr.total_nodes_traversed; nodes = std::max<long long>(nodes, gr.nodes);
        }
      double dsum = 0.0;                 // mean over games of each game's average depth, in global game-id order
      {
        std::vector<std::pair<int32_t, double>> dep;
        dep.reserve((size_t)totg);
        for (int r = 0; r < W; ++r)
          for (long long i = 0; i < all[3 * r]; ++i) {
            const az_game_rec& gr = gall[(size_t)r * maxg + i];
            dep.push_back({gr.game_id, gr.total_simulations ? (double)gr.total_nodes_traversed / (double)gr.total_simulations : 0.0});
          }
        std::sort(dep.begin(), dep.end(), [](const std::pair<int32_t, double>& a, const std::pair<int32_t, double>& b) { return a.first < b.first; });
        for (const auto& d : dep) dsum += d.second;
      }
      stats->ranks = W;
      stats->total_simulations = sims; stats->total_nodes_traversed = trav; stats->max_nodes = nodes;
      stats->mean_game_depth = totg ? dsum / (double)totg : 0.0;
      stats->replaced_games = repl;
    }
    return AZ_OK;
  }();
  if (st != AZ_OK && in_collective) abort_comm(c);
  cleanup();
  return st;
}

extern "C" int az_comm_broadcast_params(az_comm* c, az_engine* e, int32_t root) {
  if (!c) return fail(AZ_ERR_BAD_ARG, "NULL communicator");
  COMM_ALIVE(c);
  if (root < 0 || root >= c->world) return fail(AZ_ERR_BAD_ARG, "root %d of %d", root, c->world);   // the same argument on every rank
  if (hipSetDevice(c->device) != hipSuccess) { abort_comm(c); return fail(AZ_ERR_HIP, "hipSetDevice(%d) failed; communicator aborted", c->device); }
  int local = AZ_OK;
  int64_t n = 0;
  if (!e) local = fail(AZ_ERR_BAD_ARG, "NULL engine");
  else if (e->device != c->device) local = fail(AZ_ERR_BAD_ARG, "engine and communicator are on different devices");
  else if (e->cfg.oracle != AZ_ORACLE_RESNET) local = fail(AZ_ERR_STATE, "engine was created without the ResNet oracle");
  else if (c->rank == root && !e->net_loaded) local = fail(AZ_ERR_STATE, "the root has no parameters loaded");
  else local = az_net_num_params(e, &n);
  std::string local_msg = local != AZ_OK ? az_last_error() : "";
  std::vector<void*> tmp;
  std::vector<float> h;
  bool in_collective = false;
  int st = [&]() -> int {
    long long *d_s = c->d_word_send, *d_all = c->d_word_all;
    float* d = nullptr;
    if (local == AZ_OK) { local = mem_alloc(&tmp, &d, (size_t)n); if (local != AZ_OK) local_msg = az_last_error(); }
    int bad = -1;
    in_collective = true;
    AZCHK(agree(c, local, d_s, d_all, &bad));
    in_collective = false;
    if (local != AZ_OK) return fail(local, "%s", local_msg.c_str());
    if (bad >= 0) return fail(AZ_ERR_COMM, "rank %d cannot take part in the broadcast%s; nothing was sent", bad, bad == root ? " (the root has no parameters?)" : "");
    h.resize((size_t)n);
    in_collective = true;
    if (c->rank == root) HIPCHK(hipMemcpyAsync(d, e->blob.data(), sizeof(float) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    RCCLCHK(rc::g.Broadcast(d, d, (size_t)n, ncclFloat32, root, c->comm, c->stream));
    HIPCHK(hipMemcpyAsync(h.data(), d, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    in_collective = false;
    return AZ_OK;
  }();
  if (st != AZ_OK && in_collective) abort_comm(c);
  for (void* p : tmp) (void)hipFree(p);
  AZCHK(st);
  // Network.copy(nn; on_gpu = true, test_mode = true) on every rank: the kernels want their packed fragments
  return az_net_set_params(e, h.data(), n);
}
