// memory.hip -- the replay memory and the learning-status evaluation on the device.
//
//   MemoryBuffer / push_trace! / get_experience / last_batch            src/memory.jl:20-87
//   augment_with_symmetries, merge_by_state                              src/memory.jl:89-138
//   convert_samples (W, X, A, P, V), Trainer's Wmean / Hp                src/learning.jl:17-51,98-121
//   losses, learning_status, mean_learning_status                        src/learning.jl:59-90,148-181
//
// Samples never leave HBM: the move records of a self-play phase go up once (64 B per position), the ring
// buffer, the symmetric images, the by-state merge (two stable radix sorts + one segment walk, so the averages
// are accumulated in buffer order like the reference's), the Float32 tensors and the loss terms are all device
// arrays.  The network forward of the loss is the same fused tower + heads as self-play (test mode).
// The loss's network forward goes through net_launch (engine.h).
#include "engine.h"
#include "prims.h"

// push_trace! (memory.jl:74-87): one thread per game walks its records from the last position to the first
template <class Gm>
__global__ void __launch_bounds__(256) k_mem_push(const az_move_rec* __restrict__ moves, const long long* __restrict__ first,
                                                  const int* __restrict__ nmoves, const long long* __restrict__ seq0, int ngames,
                                                  double gamma, az_sample* __restrict__ buf, long long cap, long long min_seq) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ngames) return;
  const int n = nmoves[g];
  const az_move_rec* mv = moves + first[g];
  double wr = 0.0;
  for (int i = n - 1; i >= 0; --i) {
    wr = gamma * wr + (double)mv[i].reward;
    const long long seq = seq0[g] + (n - 1 - i);
    if (seq < min_seq) continue;                                   // would be overwritten within this very call
    const GEnv env = Gm::from_key(mv[i].key[0], mv[i].key[1]);
    const uint32_t m = Gm::mask(env);
    az_sample e;
    e.key[0] = mv[i].key[0]; e.key[1] = mv[i].key[1];
    double ntot = 0.0, s = 0.0;
    for (int a = 0; a < Gm::A; ++a) if ((m >> a) & 1) ntot += (double)mv[i].N[a];
    for (int a = 0; a < AZ_MAX_ACTIONS; ++a) {
      const bool ok = a < Gm::A && ((m >> a) & 1);
      e.pi[a] = ok ? (double)mv[i].N[a] / ntot : 0.0;
      if (ok) s += e.pi[a];
    }
    for (int a = 0; a < Gm::A; ++a) if ((m >> a) & 1) e.pi[a] = e.pi[a] / s;
    e.z = Gm::white_playing(env) ? wr : -wr;
    e.t = (double)(n - i);
    e.n = 1;
    buf[seq % cap] = e;
  }
}
__global__ void k_mem_gather(const az_sample* __restrict__ buf, long long cap, long long seq0, long long n, az_sample* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = buf[(seq0 + i) % cap];
}
// augment_with_symmetries (memory.jl:116-138): image k of sample i lands at n + i*NSYM + k
template <class Gm>
__global__ void k_mem_augment(const az_sample* __restrict__ in, long long n, az_sample* __restrict__ out) {
  constexpr int NS = Gm::NSYM > 0 ? Gm::NSYM : 1;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * NS) return;
  const long long i = t / NS;
  const int k = (int)(t % NS);
  az_sample e = in[i];
  const GEnv s = Gm::sym(GEnv{e.key[0], e.key[1], 0}, k);
  az_sample o = e;
  o.key[0] = s.a; o.key[1] = s.b;
  for (int j = 0; j < Gm::A; ++j) o.pi[j] = e.pi[Gm::sym_action(k, j)];
  out[n + t] = o;
}
__global__ void k_mem_keys(const az_sample* __restrict__ s, long long n, unsigned long long* k0, unsigned long long* k1, unsigned int* idx) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  k0[i] = s[i].key[0]; k1[i] = s[i].key[1]; idx[i] = (unsigned int)i;
}
__global__ void k_mem_gather_u64(const unsigned long long* __restrict__ src, const unsigned int* __restrict__ idx, long long n, unsigned long long* dst) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}
__global__ void k_mem_heads(const az_sample* __restrict__ s, const unsigned int* __restrict__ order, long long n, int* head) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool h = true;
  if (i > 0) {
    const az_sample &a = s[order[i]], &b = s[order[i - 1]];
    h = a.key[0] != b.key[0] || a.key[1] != b.key[1];
  }
  head[i] = h ? 1 : 0;
}
static constexpr long long MERGE_LONG = 256;
// merge_samples (memory.jl:89-96): 16 lanes per segment head, lane f owns the f-th 8-byte word of the sample
// (key, key, pi[0..8], z, t, n) and accumulates it over the segment in buffer order -- every field's sum is the
// reference's sequential one, the loads of a record coalesce over the lanes and pipeline over the (unrolled) loop.
__global__ void __launch_bounds__(256) k_mem_merge(const az_sample* __restrict__ s, const unsigned int* __restrict__ order,
                                                   const int* __restrict__ head, const int* __restrict__ segid, long long n,
                                                   az_sample* __restrict__ out, int* __restrict__ long_count,
                                                   long long* __restrict__ long_list) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long i = gid >> 4;
  const int f = (int)(gid & 15);
  if (i >= n || !head[i] || f >= 14) return;
  const int sg = segid[i];
  long long lo = i + 1, hi = n;                                    // first position of the next segment
  while (lo < hi) { const long long mid = (lo + hi) >> 1; if (segid[mid] > sg) hi = mid; else lo = mid + 1; }
  const long long end = lo, cnt = end - i;
  if (cnt >= MERGE_LONG) {                                         // k_mem_merge_long takes the big segments
    if (f == 0) { const int k = atomicAdd(long_count, 1); long_list[2 * k] = i; long_list[2 * k + 1] = end; }
    return;
  }
  const unsigned long long* base = (const unsigned long long*)s;
  unsigned long long* o = (unsigned long long*)(out + (sg - 1));
  const unsigned long long w0 = base[(size_t)order[i] * 14 + f];
  if (f < 2) { o[f] = w0; return; }
  if (f == 13) {
    long long acc = (long long)w0;
#pragma unroll 8
    for (long long j = i + 1; j < end; ++j) acc += (long long)base[(size_t)order[j] * 14 + 13];
    o[13] = (unsigned long long)acc;
    return;
  }
  double acc = az_u2d(w0);
#pragma unroll 8
  for (long long j = i + 1; j < end; ++j) acc += az_u2d(base[(size_t)order[j] * 14 + f]);
  o[f] = az_d2u(acc / (double)cnt);
}
// Segments of MERGE_LONG or more samples (the opening positions: one sample per game): a workgroup of 14
// wavefronts per segment, wavefront f owns word f.  The 64 lanes load 64 consecutive samples at once (all the
// memory parallelism the serial walk lacks); the additions stay strictly in buffer order -- the reference's mean over a
// generator is a sequential fold (memory.jl:89-96), and a Float64 sum in another order is another number.
// (r4) The 64 loaded words go through LDS and every lane adds them in order from broadcast reads (independent, pipelined; the
// dependent chain is one v_add_f64 per sample).  Round 3 fetched lane k's word with two v_readlane per sample inside a
// predicated, fully unrolled loop: 1.5 ms for the 32 768-sample segment of the opening position (16 384 games x 2 images),
// a third of the whole data-set build of bench.py's memory block.
__global__ void __launch_bounds__(14 * 64) k_mem_merge_long(const az_sample* __restrict__ s, const unsigned int* __restrict__ order,
                                                            const int* __restrict__ segid, const int* __restrict__ long_count,
                                                            const long long* __restrict__ long_list, az_sample* __restrict__ out) {
  if ((int)blockIdx.x >= *long_count) return;
  __shared__ unsigned long long sh[14][64];
  const long long i = long_list[2 * blockIdx.x], end = long_list[2 * blockIdx.x + 1], cnt = end - i;
  const int f = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned long long* base = (const unsigned long long*)s;
  unsigned long long* o = (unsigned long long*)(out + (segid[i] - 1));
  double acc = 0.0;
  long long iacc = 0;
  unsigned long long wn = i + lane < end ? base[(size_t)order[i + lane] * 14 + f] : 0ULL;
  for (long long j0 = i; j0 < end; j0 += 64) {                       // uniform over the workgroup: every wavefront walks the same segment
    sh[f][lane] = wn;
    const long long jn = j0 + 64 + lane;                             // the next 64 samples travel while these are added
    wn = jn < end ? base[(size_t)order[jn] * 14 + f] : 0ULL;
    __syncthreads();
    const int m = (int)((end - j0) < 64 ? (end - j0) : 64);
    if (f >= 2 && f < 13) {
      int k = 0;
      if (j0 == i) { acc = az_u2d(sh[f][0]); k = 1; }                // the sum STARTS with the first sample (keeps -0.0)
      if (m == 64) {
#pragma unroll
        for (int q = 0; q < 64; ++q) if (q >= k) acc += az_u2d(sh[f][q]);
      } else for (; k < m; ++k) acc += az_u2d(sh[f][k]);
    } else if (f == 13) {
      for (int k = 0; k < m; ++k) iacc += (long long)sh[f][k];
    }
    __syncthreads();
  }
  if (lane == 0) {
    if (f < 2) o[f] = base[(size_t)order[i] * 14 + f];
    else o[f] = f == 13 ? (unsigned long long)iacc : az_d2u(acc / (double)cnt);
  }
}
// convert_samples (learning.jl:17-51) + per-sample entropy term of Hp (learning.jl:65,111)
template <class Gm>
__global__ void k_mem_convert(const az_sample* __restrict__ s, long long n, int policy, GEnv* envs, float* W, float* A, float* P, float* V,
                              double* t_w, double* t_hp, double* t_n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const az_sample e = s[i];
  const GEnv env = Gm::from_key(e.key[0], e.key[1]);
  envs[i] = env;
  const float w = policy == AZ_WEIGHT_CONSTANT ? 1.0f : policy == AZ_WEIGHT_LOG ? (float)(az_log2((double)e.n) + 1.0) : (float)e.n;
  W[i] = w;
  const uint32_t m = Gm::mask(env);
  double hp = 0.0;
  for (int a = 0; a < Gm::A; ++a) {
    const float p = (float)e.pi[a];
    A[i * Gm::A + a] = (float)((m >> a) & 1);
    P[i * Gm::A + a] = p;
    hp += (double)(p * az_logf(p + 1.1920929e-07f) * w);
  }
  V[i] = (float)e.z;
  t_w[i] = (double)w; t_hp[i] = hp; t_n[i] = (double)e.n;
}
template <class Gm>
__global__ void k_mem_planes(const GEnv* __restrict__ envs, long long n, float* __restrict__ X) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  constexpr int E = Gm::C * Gm::P;
  if (t >= n * E) return;
  const long long i = t / E;
  const int r = (int)(t % E);
  X[t] = Gm::plane(envs[i], r % Gm::P, r / Gm::P);
}
// per-sample terms of `losses` and of Hpnet (learning.jl:59-90,158-168), Float32 like the reference's arrays
__global__ void k_loss_terms(const float* __restrict__ W, const float* __restrict__ P, const float* __restrict__ V, const float* __restrict__ Ph,
                             const float* __restrict__ Vh, const float* __restrict__ Pinv, long long n, int A, float renorm,
                             double* t_kl, double* t_hn, double* t_mse, double* t_inv) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float w = W[i];
  double kl = 0.0, hn = 0.0;
  for (int a = 0; a < A; ++a) {
    const float ph = Ph[i * A + a];
    const float lg = az_logf(ph + 1.1920929e-07f);
    kl += (double)(P[i * A + a] * lg * w);
    hn += (double)(ph * lg * w);
  }
  const float d = Vh[i] / renorm - V[i] / renorm;
  t_kl[i] = kl; t_hn[i] = hn; t_mse[i] = (double)(d * d * w); t_inv[i] = (double)(Pinv[i] * w);
}

__global__ void k_f32_to_f64(const float* __restrict__ in, long long n, double* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (double)in[i];
}

// Five sums per loss batch in one launch: block b reduces samples [b*batch, min(n, (b+1)*batch)) of the five term
// arrays with a fixed-shape tree (thread-strided partial sums, then shared-memory halving), so the result does not
// depend on scheduling.  out[b][0..5) = sum w, kl, hn, mse, inv.
__global__ void __launch_bounds__(256) k_batch_sums(const double* __restrict__ tw, const double* __restrict__ tkl, const double* __restrict__ thn,
                                                    const double* __restrict__ tmse, const double* __restrict__ tinv, long long n, long long batch,
                                                    double* __restrict__ out) {
  __shared__ double sh[5][256];
  const long long b0 = (long long)blockIdx.x * batch;
  const long long m = (n - b0) < batch ? (n - b0) : batch;
  const double* src[5] = {tw, tkl, thn, tmse, tinv};
  double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (long long i = threadIdx.x; i < m; i += 256)
#pragma unroll
    for (int k = 0; k < 5; ++k) acc[k] += src[k][b0 + i];
#pragma unroll
  for (int k = 0; k < 5; ++k) sh[k][threadIdx.x] = acc[k];
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w)
#pragma unroll
      for (int k = 0; k < 5; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x < 5) out[(size_t)blockIdx.x * 5 + threadIdx.x] = sh[threadIdx.x][0];
}

struct DevReducer {                     // deterministic sum of doubles (prims.h: fixed tiles, fixed tree)
  double* tmp = nullptr;
  int init(long long nmax, hipStream_t) {
    HIPCHK(hipMalloc((void**)&tmp, sizeof(double) * prims::sum_tmp_doubles(nmax)));
    return AZ_OK;
  }
  int sum(const double* d_in, long long n, hipStream_t st, double* out) {
    double* d_res = nullptr;
    HIPCHK(prims::sum_doubles(d_in, n, tmp, &d_res, st));
    HIPCHK(hipMemcpyAsync(out, d_res, sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return AZ_OK;
  }
  ~DevReducer() { if (tmp) (void)hipFree(tmp); }
};

#define MEMORY(m) if (!(m)) return fail(AZ_ERR_BAD_ARG, "memory is NULL"); HIPCHK(hipSetDevice((m)->device))

extern "C" int az_memory_create(int32_t game, int32_t device, int64_t capacity, az_memory** out) {
  if (!out) return fail(AZ_ERR_BAD_ARG, "NULL argument");
  *out = nullptr;
  GameInfo gi;
  if (!game_info(game, &gi)) return fail(AZ_ERR_BAD_ARG, "unknown game id %d", game);
  if (game == AZ_GAME_GO9_PLANES) return fail(AZ_ERR_BAD_ARG, "game id %d is a network-only tensor geometry (no device twin): the device replay memory needs the game's state keys", game);
  if (capacity < 1 || capacity > (1LL << 31) - 1) return fail(AZ_ERR_BAD_ARG, "capacity must be in 1..2^31-1");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(AZ_ERR_BAD_ARG, "device %d not available (%d visible)", device, ndev);
  HIPCHK(hipSetDevice(device));
  az_memory* m = new (std::nothrow) az_memory();
  if (!m) return fail(AZ_ERR_HIP, "out of host memory");
  m->game = game; m->device = device; m->gi = gi; m->cap = capacity; m->total = 0; m->cur_batch = 0; m->d_buf = nullptr; m->stream = nullptr;
  int st = [&]() -> int {
    HIPCHK(hipStreamCreate(&m->stream));
    AZCHK(mem_alloc<az_sample>(nullptr, &m->d_buf, (size_t)capacity));
    return AZ_OK;
  }();
  if (st != AZ_OK) { if (m->stream) (void)hipStreamDestroy(m->stream); delete m; return st; }
  *out = m;
  return AZ_OK;
}
extern "C" int az_memory_destroy(az_memory* m) {
  if (!m) return AZ_OK;
  (void)hipSetDevice(m->device);
  if (m->d_buf) (void)hipFree(m->d_buf);
  if (m->stream) (void)hipStreamDestroy(m->stream);
  delete m;
  return AZ_OK;
}
int memory_push_device(az_memory* m, const az_move_rec* d_moves, const std::vector<long long>& first, const std::vector<int>& cnt, double gamma) {
  const int ng = (int)first.size();
  if (!ng) return AZ_OK;
  std::vector<long long> seq0(ng);
  long long tot = 0;
  for (int g = 0; g < ng; ++g) { seq0[g] = m->total + tot; tot += cnt[g]; }
  long long *d_first, *d_seq; int* d_cnt;
  std::vector<void*> tmp;
  auto cleanup = [&]() { for (void* p : tmp) (void)hipFree(p); };
  int st = [&]() -> int {
    AZCHK(mem_alloc(&tmp, &d_first, ng)); AZCHK(mem_alloc(&tmp, &d_cnt, ng)); AZCHK(mem_alloc(&tmp, &d_seq, ng));
    HIPCHK(hipMemcpyAsync(d_first, first.data(), sizeof(long long) * ng, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(d_cnt, cnt.data(), sizeof(int) * ng, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(d_seq, seq0.data(), sizeof(long long) * ng, hipMemcpyHostToDevice, m->stream));
    const long long min_seq = std::max<long long>(0, m->total + tot - m->cap);
    DISPATCH_GAME(m->game, hipLaunchKernelGGL((k_mem_push<Gm>), dim3((ng + 255) / 256), dim3(256), 0, m->stream, d_moves, d_first, d_cnt, d_seq, ng, gamma, m->d_buf, (long long)m->cap, min_seq));
    HIPCHK(hipStreamSynchronize(m->stream));
    HIPCHK(hipGetLastError());
    return AZ_OK;
  }();
  cleanup();
  AZCHK(st);
  m->total += tot;
  m->cur_batch += tot;                                             // mem.cur_batch_size += n, memory.jl:86
  return AZ_OK;
}
extern "C" int az_memory_push(az_memory* m, const az_trace_buf* tr, double gamma) {
  MEMORY(m);
  if (!tr || tr->num_games < 0 || (tr->num_games > 0 && (!tr->games || !tr->moves))) return fail(AZ_ERR_BAD_ARG, "NULL trace buffers");
  const int ng = (int)tr->num_games;
  if (!ng) return AZ_OK;
  std::vector<long long> first(ng);
  std::vector<int> cnt(ng);
  for (int g = 0; g < ng; ++g) {
    const az_game_rec& r = tr->games[g];
    if (r.num_moves < 0 || r.first_move < 0 || (int64_t)r.first_move + r.num_moves > tr->num_moves) return fail(AZ_ERR_BAD_ARG, "game record %d points outside the move records", g);
    first[g] = r.first_move; cnt[g] = r.num_moves;
  }
  az_move_rec* d_moves = nullptr;
  HIPCHK(hipMalloc((void**)&d_moves, sizeof(az_move_rec) * (size_t)std::max<int64_t>(tr->num_moves, 1)));
  int st = [&]() -> int {
    HIPCHK(hipMemcpyAsync(d_moves, tr->moves, sizeof(az_move_rec) * (size_t)tr->num_moves, hipMemcpyHostToDevice, m->stream));
    return memory_push_device(m, d_moves, first, cnt, gamma);
  }();
  (void)hipFree(d_moves);
  return st;
}
// push_trace! for every game of the engine's last self-play phase, in game-id order, straight from the engine's
// device-resident move records (src/training.jl:284-299 -> src/memory.jl:74-87 without the 64 B per position ever
// visiting the host)
extern "C" int az_memory_push_engine(az_memory* m, az_engine* e, double gamma) {
  MEMORY(m);
  if (!e) return fail(AZ_ERR_BAD_ARG, "engine is NULL");
  if (e->running) return fail(AZ_ERR_STATE, "self-play in progress");
  if (e->cfg.game != m->game || e->device != m->device) return fail(AZ_ERR_BAD_ARG, "memory and engine differ in game or device");
  if (!e->d_phase) return fail(AZ_ERR_STATE, "the engine holds no device-resident phase (az_selfplay_run / az_selfplay_begin with num_games > 0 first)");
  const size_t ng = e->ph_games.size();
  std::vector<int> ord(ng);
  for (size_t i = 0; i < ng; ++i) ord[i] = (int)i;
  std::sort(ord.begin(), ord.end(), [&](int a, int b) { return e->ph_games[a].game_id < e->ph_games[b].game_id; });
  std::vector<long long> first(ng);
  std::vector<int> cnt(ng);
  for (size_t i = 0; i < ng; ++i) { first[i] = e->ph_off[ord[i]]; cnt[i] = e->ph_games[ord[i]].num_moves; }
  AZCHK(sync_all(e));
  return memory_push_device(m, e->d_phase, first, cnt, gamma);
}
// push!(mem.buf, sample) for host-resident TrainingSamples (a reference-side MemoryBuffer, or subsets such as the game
// stages of memory_report, src/learning.jl:192-216); cur_batch_size is not advanced (push_trace! does that)
extern "C" int az_memory_push_samples(az_memory* m, const az_sample* samples, int64_t n) {
  MEMORY(m);
  if (n < 0 || (n > 0 && !samples)) return fail(AZ_ERR_BAD_ARG, "NULL samples");
  // keep only what survives in the ring, oldest first, then at most two contiguous copies
  const int64_t skip = std::max<int64_t>(0, n - m->cap);
  int64_t left = n - skip;
  const az_sample* src = samples + skip;
  int64_t seq = m->total + skip;
  while (left > 0) {
    const int64_t pos = seq % m->cap, run = std::min<int64_t>(left, m->cap - pos);
    HIPCHK(hipMemcpyAsync(m->d_buf + pos, src, sizeof(az_sample) * (size_t)run, hipMemcpyHostToDevice, m->stream));
    src += run; seq += run; left -= run;
  }
  HIPCHK(hipStreamSynchronize(m->stream));
  m->total += n;
  return AZ_OK;
}

extern "C" int az_memory_length(az_memory* m, int64_t* length, int64_t* cur_batch_size) {
  MEMORY(m);
  const int64_t len = std::min<int64_t>(m->total, m->cap);
  if (length) *length = len;
  if (cur_batch_size) *cur_batch_size = std::min<int64_t>(m->cur_batch, len);     // memory.jl:53
  return AZ_OK;
}
extern "C" int az_memory_new_batch(az_memory* m) { MEMORY(m); m->cur_batch = 0; return AZ_OK; }
extern "C" int az_memory_empty(az_memory* m) { MEMORY(m); m->total = 0; m->cur_batch = 0; return AZ_OK; }

template <class Gm>
static int dataset_build(az_memory* m, az_dataset* d, int which, bool use_sym, bool merge, int policy) {
  hipStream_t st = m->stream;
  const int64_t len = std::min<int64_t>(m->total, m->cap);
  const int64_t n0 = which == 1 ? std::min<int64_t>(m->cur_batch, len) : len;
  const long long seq0 = m->total - n0;                            // the newest n0 samples, oldest first
  const int nsym = use_sym ? Gm::NSYM : 0;
  const int64_t n1 = n0 * (1 + nsym);
  if (n1 > (1LL << 31) - 1) return fail(AZ_ERR_CAPACITY, "data set too large");
  std::vector<void*> tmp;
  auto cleanup = [&]() { for (void* p : tmp) (void)hipFree(p); };
  int rc = [&]() -> int {
    az_sample* s1;
    AZCHK(mem_alloc(&tmp, &s1, (size_t)n1));
    if (n0) hipLaunchKernelGGL(k_mem_gather, dim3((unsigned)((n0 + 255) / 256)), dim3(256), 0, st, m->d_buf, (long long)m->cap, seq0, (long long)n0, s1);
    if (nsym && n0) hipLaunchKernelGGL((k_mem_augment<Gm>), dim3((unsigned)((n0 * nsym + 255) / 256)), dim3(256), 0, st, s1, (long long)n0, s1);
    int64_t n2 = n1;
    if (merge && n1 > 0) {
      unsigned long long *k0, *k1, *ks, *ks2; unsigned int *i0, *i1, *i2; int *head, *seg;
      AZCHK(mem_alloc(&tmp, &k0, (size_t)n1)); AZCHK(mem_alloc(&tmp, &k1, (size_t)n1)); AZCHK(mem_alloc(&tmp, &ks, (size_t)n1)); AZCHK(mem_alloc(&tmp, &ks2, (size_t)n1));
      AZCHK(mem_alloc(&tmp, &i0, (size_t)n1)); AZCHK(mem_alloc(&tmp, &i1, (size_t)n1)); AZCHK(mem_alloc(&tmp, &i2, (size_t)n1));
      AZCHK(mem_alloc(&tmp, &head, (size_t)n1)); AZCHK(mem_alloc(&tmp, &seg, (size_t)n1));
      const unsigned gb = (unsigned)((n1 + 255) / 256);
      hipLaunchKernelGGL(k_mem_keys, dim3(gb), dim3(256), 0, st, s1, (long long)n1, k0, k1, i0);
      // LSD: stable sort by key[1], then by key[0] -> ascending (key[0], key[1], buffer index)
      int* stmp;
      AZCHK(mem_alloc(&tmp, &stmp, std::max(prims::sort_tmp_ints(n1), prims::scan_tmp_ints(n1))));
      HIPCHK(prims::sort_pairs(k1, ks, i0, i1, n1, stmp, st));        // (k1, i0) are the sort's ping-pong partner: not used again
      hipLaunchKernelGGL(k_mem_gather_u64, dim3(gb), dim3(256), 0, st, k0, i1, (long long)n1, ks2);
      HIPCHK(prims::sort_pairs(ks2, ks, i1, i2, n1, stmp, st));
      hipLaunchKernelGGL(k_mem_heads, dim3(gb), dim3(256), 0, st, s1, i2, (long long)n1, head);
      HIPCHK(prims::scan_ints(head, seg, n1, true, stmp, st));
      int nseg = 0;
      HIPCHK(hipMemcpyAsync(&nseg, seg + (n1 - 1), sizeof(int), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      n2 = nseg;
      AZCHK(mem_alloc(&d->allocs, &d->d_samples, (size_t)n2));
      static_assert(sizeof(az_sample) == 14 * 8, "k_mem_merge walks az_sample as 14 words");
      int* long_count; long long* long_list;
      const long long max_long = n1 / MERGE_LONG + 1;
      AZCHK(mem_alloc(&tmp, &long_count, 1)); AZCHK(mem_alloc(&tmp, &long_list, (size_t)(2 * max_long)));
      HIPCHK(hipMemsetAsync(long_count, 0, sizeof(int), st));
      hipLaunchKernelGGL(k_mem_merge, dim3((unsigned)((n1 * 16 + 255) / 256)), dim3(256), 0, st, s1, i2, head, seg, (long long)n1, d->d_samples, long_count, long_list);
      hipLaunchKernelGGL(k_mem_merge_long, dim3((unsigned)max_long), dim3(14 * 64), 0, st, s1, i2, seg, long_count, long_list, d->d_samples);
    } else {
      AZCHK(mem_alloc(&d->allocs, &d->d_samples, (size_t)n2));
      if (n2) HIPCHK(hipMemcpyAsync(d->d_samples, s1, sizeof(az_sample) * (size_t)n2, hipMemcpyDeviceToDevice, st));
    }
    d->n = n2;
    AZCHK(mem_alloc(&d->allocs, &d->d_envs, (size_t)n2)); AZCHK(mem_alloc(&d->allocs, &d->d_W, (size_t)n2)); AZCHK(mem_alloc(&d->allocs, &d->d_V, (size_t)n2));
    AZCHK(mem_alloc(&d->allocs, &d->d_A, (size_t)n2 * Gm::A)); AZCHK(mem_alloc(&d->allocs, &d->d_P, (size_t)n2 * Gm::A));
    AZCHK(mem_alloc(&d->allocs, &d->d_X, (size_t)n2 * Gm::C * Gm::P));
    d->sum_n = 0; d->Wtot = 0.0; d->Wmean = 0.f; d->Hp = 0.f;
    if (n2) {
      double *tw, *thp, *tn;
      AZCHK(mem_alloc(&tmp, &tw, (size_t)n2)); AZCHK(mem_alloc(&tmp, &thp, (size_t)n2)); AZCHK(mem_alloc(&tmp, &tn, (size_t)n2));
      hipLaunchKernelGGL((k_mem_convert<Gm>), dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, d->d_samples, (long long)n2, policy, d->d_envs, d->d_W, d->d_A, d->d_P, d->d_V, tw, thp, tn);
      const long long ne = (long long)n2 * Gm::C * Gm::P;
      hipLaunchKernelGGL((k_mem_planes<Gm>), dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, d->d_envs, (long long)n2, d->d_X);
      DevReducer red;
      AZCHK(red.init(n2, st));
      double sw, shp, sn;
      AZCHK(red.sum(tw, n2, st, &sw)); AZCHK(red.sum(thp, n2, st, &shp)); AZCHK(red.sum(tn, n2, st, &sn));
      d->Wtot = sw; d->sum_n = (int64_t)sn;
      d->Wmean = (float)(sw / (double)n2);                         // mean(W), learning.jl:110
      d->Hp = (float)(-shp / sw);                                  // entropy_wmean(P, W), learning.jl:111
    }
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    return AZ_OK;
  }();
  cleanup();
  return rc;
}

extern "C" int az_dataset_destroy(az_dataset* d) {
  if (!d) return AZ_OK;
  (void)hipSetDevice(d->device);
  for (void* p : d->allocs) (void)hipFree(p);
  delete d;
  return AZ_OK;
}
extern "C" int az_dataset_create(az_memory* m, int32_t which, int32_t use_symmetries, int32_t use_position_averaging,
                                 int32_t weighing_policy, az_dataset** out) {
  MEMORY(m);
  if (!out) return fail(AZ_ERR_BAD_ARG, "NULL argument");
  *out = nullptr;
  if (which != 0 && which != 1) return fail(AZ_ERR_BAD_ARG, "which must be 0 (get_experience) or 1 (last_batch)");
  if (weighing_policy < AZ_WEIGHT_CONSTANT || weighing_policy > AZ_WEIGHT_LINEAR) return fail(AZ_ERR_BAD_ARG, "unknown samples_weighing_policy %d", weighing_policy);
  az_dataset* d = new (std::nothrow) az_dataset();
  if (!d) return fail(AZ_ERR_HIP, "out of host memory");
  d->game = m->game; d->device = m->device; d->gi = m->gi; d->stream = m->stream; d->n = 0;
  int st = AZ_OK;
  switch (m->game) {
    case AZ_GAME_CONNECT_FOUR: st = dataset_build<ConnectFour>(m, d, which, use_symmetries != 0, use_position_averaging != 0, weighing_policy); break;
    case AZ_GAME_TICTACTOE: st = dataset_build<TicTacToe>(m, d, which, use_symmetries != 0, use_position_averaging != 0, weighing_policy); break;
    case AZ_GAME_MANCALA: st = dataset_build<Mancala>(m, d, which, use_symmetries != 0, use_position_averaging != 0, weighing_policy); break;
    default: st = fail(AZ_ERR_BAD_ARG, "game id %d has no device twin", m->game); break;
  }
  if (st != AZ_OK) { az_dataset_destroy(d); return st; }
  *out = d;
  return AZ_OK;
}
extern "C" int az_dataset_get_info(az_dataset* d, az_dataset_info* out) {
  if (!d || !out) return fail(AZ_ERR_BAD_ARG, "NULL argument");
  out->num_samples = d->n; out->sum_n = d->sum_n; out->Wtot = d->Wtot; out->Wmean = d->Wmean; out->Hp = d->Hp;
  return AZ_OK;
}
extern "C" int az_dataset_read(az_dataset* d, int64_t first, int64_t count, az_sample* samples, float* W, float* X, float* A,
                               float* P, float* V) {
  if (!d) return fail(AZ_ERR_BAD_ARG, "dataset is NULL");
  HIPCHK(hipSetDevice(d->device));
  if (first < 0 || count < 0 || first + count > d->n) return fail(AZ_ERR_BAD_ARG, "range [%lld, %lld) outside the %lld samples", (long long)first, (long long)(first + count), (long long)d->n);
  if (!count) return AZ_OK;
  const GameInfo& gi = d->gi;
  const size_t xs = (size_t)gi.C * gi.P;
  if (samples) HIPCHK(hipMemcpyAsync(samples, d->d_samples + first, sizeof(az_sample) * (size_t)count, hipMemcpyDeviceToHost, d->stream));
  if (W) HIPCHK(hipMemcpyAsync(W, d->d_W + first, sizeof(float) * (size_t)count, hipMemcpyDeviceToHost, d->stream));
  if (V) HIPCHK(hipMemcpyAsync(V, d->d_V + first, sizeof(float) * (size_t)count, hipMemcpyDeviceToHost, d->stream));
  if (A) HIPCHK(hipMemcpyAsync(A, d->d_A + (size_t)first * gi.A, sizeof(float) * (size_t)count * gi.A, hipMemcpyDeviceToHost, d->stream));
  if (P) HIPCHK(hipMemcpyAsync(P, d->d_P + (size_t)first * gi.A, sizeof(float) * (size_t)count * gi.A, hipMemcpyDeviceToHost, d->stream));
  if (X) HIPCHK(hipMemcpyAsync(X, d->d_X + (size_t)first * xs, sizeof(float) * (size_t)count * xs, hipMemcpyDeviceToHost, d->stream));
  HIPCHK(hipStreamSynchronize(d->stream));
  return AZ_OK;
}

// Sum of squares of the trainable parameters (Flux.params: all but the BatchNorm running statistics), host side
static double reg_sum_trainable(const az_engine* e) {
  const GameInfo& gi = e->gi;
  const size_t P = gi.P, F = e->cfg.num_filters, npf = e->cfg.num_policy_head_filters, nvf = e->cfg.num_value_head_filters;
  const float* b = e->blob.data();
  double s = 0.0;
  auto take = [&](size_t c) { for (size_t i = 0; i < c; ++i) s += (double)b[i] * (double)b[i]; b += c; };
  auto skip = [&](size_t c) { b += c; };
  take(9 * (size_t)gi.C * F); take(F); take(2 * F); skip(2 * F);
  for (int l = 0; l < 2 * e->cfg.num_blocks; ++l) { take(9 * F * F); take(F); take(2 * F); skip(2 * F); }
  take(F * npf); take(npf); take(2 * npf); skip(2 * npf); take((size_t)gi.A * P * npf); take(gi.A);
  take(F * nvf); take(nvf); take(2 * nvf); skip(2 * nvf); take(F * P * nvf); take(F); take(F); take(1);
  return s;
}

template <class Gm>
static int learning_status_run(az_engine* e, az_dataset* d, double l2, double cinv, double renorm, int64_t batch, az_learning_status_t* out) {
  const int64_t n = d->n;
  hipStream_t st = e->stream;
  std::vector<void*> tmp;
  auto cleanup = [&]() { for (void* p : tmp) (void)hipFree(p); };
  int rc = [&]() -> int {
    float *Ph, *Vh, *Pinv; double *tkl, *thn, *tmse, *tinv, *tw;
    AZCHK(mem_alloc(&tmp, &Ph, (size_t)n * Gm::A)); AZCHK(mem_alloc(&tmp, &Vh, (size_t)n)); AZCHK(mem_alloc(&tmp, &Pinv, (size_t)n));
    AZCHK(mem_alloc(&tmp, &tkl, (size_t)n)); AZCHK(mem_alloc(&tmp, &thn, (size_t)n)); AZCHK(mem_alloc(&tmp, &tmse, (size_t)n)); AZCHK(mem_alloc(&tmp, &tinv, (size_t)n));
    AZCHK(mem_alloc(&tmp, &tw, (size_t)n));
    HIPCHK(hipStreamSynchronize(d->stream));
    // Network.forward_normalized (network.jl:264-271) in test mode on the device-resident states, nn_cap at a time
    const int64_t nchunks = (n + e->nn_cap - 1) / e->nn_cap;
    std::vector<int> counts((size_t)nchunks);
    for (int64_t c = 0; c < nchunks; ++c) counts[c] = (int)std::min<int64_t>(e->nn_cap, n - c * e->nn_cap);
    int* d_counts;
    AZCHK(mem_alloc(&tmp, &d_counts, (size_t)nchunks));
    HIPCHK(hipMemcpyAsync(d_counts, counts.data(), sizeof(int) * (size_t)nchunks, hipMemcpyHostToDevice, st));
    for (int64_t c = 0; c < nchunks; ++c) {
      const int64_t off = c * e->nn_cap;
      AZCHK(net_launch(e, st, false, e->d_hfeat, d->d_envs + off, e->d_iota, d_counts + c, counts[c], nullptr, nullptr, Ph + (size_t)off * Gm::A, Vh + off, Pinv + off, Gm::A));
    }
    HIPCHK(hipStreamSynchronize(st));                              // `counts` must outlive the copy
    hipLaunchKernelGGL(k_loss_terms, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d->d_W, d->d_P, d->d_V, Ph, Vh, Pinv, (long long)n, Gm::A, (float)renorm, tkl, thn, tmse, tinv);
    hipLaunchKernelGGL(k_f32_to_f64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d->d_W, (long long)n, tw);
    if (batch > n) batch = n;
    const int64_t nb = (n + batch - 1) / batch;                    // DataLoader(partial = true), learning.jl:171-176
    double* d_sums;
    AZCHK(mem_alloc(&tmp, &d_sums, (size_t)nb * 5));
    hipLaunchKernelGGL(k_batch_sums, dim3((unsigned)nb), dim3(256), 0, st, tw, tkl, thn, tmse, tinv, (long long)n, (long long)batch, d_sums);
    std::vector<double> sums((size_t)nb * 5);
    HIPCHK(hipMemcpyAsync(sums.data(), d_sums, sizeof(double) * sums.size(), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const float Lreg = l2 == 0.0 ? 0.f : (float)((double)(float)l2 * reg_sum_trainable(e));
    double aL = 0., aLp = 0., aLv = 0., aLreg = 0., aLinv = 0., aHn = 0., aw = 0.;
    for (int64_t b = 0; b < nb; ++b) {
      const int64_t m = std::min<int64_t>(batch, n - b * batch);
      const double bw = sums[b * 5], kl = sums[b * 5 + 1], hn = sums[b * 5 + 2], mse = sums[b * 5 + 3], inv = sums[b * 5 + 4];
      const float Lp = (float)(-kl / bw) - d->Hp;
      const float Lv = (float)(mse / bw);
      const float Linv = cinv == 0.0 ? 0.f : (float)cinv * (float)(inv / bw);
      const float L = ((float)(bw / (double)m) / d->Wmean) * (Lp + Lv + Lreg + Linv);
      const float Hn = (float)(-hn / bw);
      aL += (double)L * bw; aLp += (double)Lp * bw; aLv += (double)Lv * bw; aLreg += (double)Lreg * bw;
      aLinv += (double)Linv * bw; aHn += (double)Hn * bw; aw += bw;
    }
    out->L = (float)(aL / aw); out->Lp = (float)(aLp / aw); out->Lv = (float)(aLv / aw); out->Lreg = (float)(aLreg / aw);
    out->Linv = (float)(aLinv / aw); out->Hp = d->Hp; out->Hpnet = (float)(aHn / aw);
    HIPCHK(hipGetLastError());
    return AZ_OK;
  }();
  cleanup();
  return rc;
}

extern "C" int az_learning_status(az_engine* e, az_dataset* d, double l2_regularization, double nonvalidity_penalty,
                                  double rewards_renormalization, int64_t loss_computation_batch_size, az_learning_status_t* out) {
  ENGINE(e);
  if (!d || !out) return fail(AZ_ERR_BAD_ARG, "NULL argument");
  if (e->running) return fail(AZ_ERR_STATE, "self-play in progress");
  if (e->cfg.oracle != AZ_ORACLE_RESNET || !e->net_loaded) return fail(AZ_ERR_STATE, "az_net_set_params has not been called");
  if (d->game != e->cfg.game || d->device != e->device) return fail(AZ_ERR_BAD_ARG, "data set and engine differ in game or device");
  if (d->n < 1) return fail(AZ_ERR_BAD_ARG, "empty data set");
  if (loss_computation_batch_size < 1) return fail(AZ_ERR_BAD_ARG, "loss_computation_batch_size must be >= 1");
  if (!(rewards_renormalization > 0.0)) return fail(AZ_ERR_BAD_ARG, "rewards_renormalization must be > 0");
  AZCHK(sync_all(e));
  DISPATCH_GAME(e->cfg.game, AZCHK(learning_status_run<Gm>(e, d, l2_regularization, nonvalidity_penalty, rewards_renormalization, loss_computation_batch_size, out)));
  return AZ_OK;
}
