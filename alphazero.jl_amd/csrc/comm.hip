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
// a file; bench.py uses torch.distributed's store).
#include "engine.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace rc {
struct Lib {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static Lib g;
static int load() {
  if (g.so) return AZ_OK;
  void* so = nullptr;
  Dl_info info;
  if (dladdr(reinterpret_cast<void*>(&hipGetLastError), &info) && info.dli_fname) {
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
  g.AllGather = (decltype(g.AllGather))dlsym(so, "ncclAllGather");
  g.Broadcast = (decltype(g.Broadcast))dlsym(so, "ncclBroadcast");
  g.GetErrorString = (decltype(g.GetErrorString))dlsym(so, "ncclGetErrorString");
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

extern "C" int az_comm_destroy(az_comm* c) {
  if (!c) return AZ_OK;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)rc::g.CommDestroy(c->comm);
  if (c->stream) (void)hipStreamDestroy(c->stream);
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
  int st = [&]() -> int {
    HIPCHK(hipStreamCreate(&c->stream));
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

extern "C" int az_comm_gather_push(az_comm* c, az_engine* e, az_memory* m, double gamma, az_gather_stats* stats) {
  if (!c || !e) return fail(AZ_ERR_BAD_ARG, "NULL argument");
  if (e->device != c->device || (m && m->device != c->device)) return fail(AZ_ERR_BAD_ARG, "engine / memory / communicator are on different devices");
  if (m && m->game != e->cfg.game) return fail(AZ_ERR_BAD_ARG, "memory and engine differ in game");
  if (e->running) return fail(AZ_ERR_STATE, "self-play in progress");
  if (!e->d_phase) return fail(AZ_ERR_STATE, "the engine holds no device-resident phase (az_selfplay_run first)");
  HIPCHK(hipSetDevice(c->device));
  AZCHK(sync_all(e));
  const auto t0 = std::chrono::steady_clock::now();
  const int W = c->world;
  const size_t ng = e->ph_games.size();
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
  int st = [&]() -> int {
    // 1. counts of every rank
    long long hc[2] = {(long long)ng, nm};
    long long *d_cnt_send, *d_cnt_all;
    AZCHK(mem_alloc(&tmp, &d_cnt_send, 2)); AZCHK(mem_alloc(&tmp, &d_cnt_all, (size_t)2 * W));
    HIPCHK(hipMemcpyAsync(d_cnt_send, hc, sizeof hc, hipMemcpyHostToDevice, c->stream));
    RCCLCHK(rc::g.AllGather(d_cnt_send, d_cnt_all, 2, ncclInt64, c->comm, c->stream));
    std::vector<long long> all((size_t)2 * W);
    HIPCHK(hipMemcpyAsync(all.data(), d_cnt_all, sizeof(long long) * 2 * W, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    long long maxg = 1, maxm = 1, totg = 0, totm = 0;
    for (int r = 0; r < W; ++r) { maxg = std::max(maxg, all[2 * r]); maxm = std::max(maxm, all[2 * r + 1]); totg += all[2 * r]; totm += all[2 * r + 1]; }
    // 2. this rank's records packed in game-id order, then the two all-gathers (equal, padded segments per rank)
    az_move_rec *d_msend, *d_mall; az_game_rec *d_gsend, *d_gall; long long *d_src, *d_dst; int* d_n;
    AZCHK(mem_alloc(&tmp, &d_msend, (size_t)maxm)); AZCHK(mem_alloc(&tmp, &d_mall, (size_t)maxm * W));
    AZCHK(mem_alloc(&tmp, &d_gsend, (size_t)maxg)); AZCHK(mem_alloc(&tmp, &d_gall, (size_t)maxg * W));
    AZCHK(mem_alloc(&tmp, &d_src, std::max<size_t>(ng, 1))); AZCHK(mem_alloc(&tmp, &d_dst, std::max<size_t>(ng, 1))); AZCHK(mem_alloc(&tmp, &d_n, std::max<size_t>(ng, 1)));
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
    const auto t1 = std::chrono::steady_clock::now();
    struct Ref { int32_t id; long long first; int n; };
    std::vector<Ref> refs;
    refs.reserve((size_t)totg);
    for (int r = 0; r < W; ++r)
      for (long long i = 0; i < all[2 * r]; ++i) {
        const az_game_rec& gr = gall[(size_t)r * maxg + i];
        refs.push_back({gr.game_id, (long long)r * maxm + gr.first_move, gr.num_moves});
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
      stats->bytes = (int64_t)W * (maxm * (long long)sizeof(az_move_rec) + maxg * (long long)sizeof(az_game_rec) + 16);
      stats->gather_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
      stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return AZ_OK;
  }();
  cleanup();
  return st;
}

extern "C" int az_comm_broadcast_params(az_comm* c, az_engine* e, int32_t root) {
  if (!c || !e) return fail(AZ_ERR_BAD_ARG, "NULL argument");
  if (root < 0 || root >= c->world) return fail(AZ_ERR_BAD_ARG, "root %d of %d", root, c->world);
  if (e->cfg.oracle != AZ_ORACLE_RESNET) return fail(AZ_ERR_STATE, "engine was created without the ResNet oracle");
  if (c->rank == root && !e->net_loaded) return fail(AZ_ERR_STATE, "the root has no parameters loaded");
  HIPCHK(hipSetDevice(c->device));
  int64_t n = 0;
  AZCHK(az_net_num_params(e, &n));
  float* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, sizeof(float) * (size_t)n));
  std::vector<float> h((size_t)n);
  int st = [&]() -> int {
    if (c->rank == root) HIPCHK(hipMemcpyAsync(d, e->blob.data(), sizeof(float) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    RCCLCHK(rc::g.Broadcast(d, d, (size_t)n, ncclFloat32, root, c->comm, c->stream));
    HIPCHK(hipMemcpyAsync(h.data(), d, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return AZ_OK;
  }();
  (void)hipFree(d);
  AZCHK(st);
  // Network.copy(nn; on_gpu = true, test_mode = true) on every rank: the kernels want their packed fragments
  return az_net_set_params(e, h.data(), n);
}
