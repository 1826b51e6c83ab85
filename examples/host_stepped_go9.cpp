// host_stepped_go9.cpp -- BASELINE configs[4] END TO END in the only form it can take: the game's rules and the search tree
// stay on the host, the ResNet runs on the MI355X through the public C ABI (az_net_set_params / az_net_forward = the
// Network plugin seam, src/networks/network.jl:264-271).  That is how the reference plays any game that is not written in
// Julia: src/openspiel.jl:7-173 wraps OpenSpiel's C++ rules in a GameInterface, the stock MCTS (src/mcts.jl) walks host
// trees and the inference server (src/simulations.jl:23-56) batches the leaves of `num_workers` games for the network.
//
// What is measured: simulations per second of W lock-step workers (one tree each, src/mcts.jl:157-226 semantics: Float64
// UCT, first maximum, Dirichlet noise at the root, trees kept between the moves of a game), with the share of wall time the
// host spends in rules + tree code against the share it waits for az_net_forward (PCIe both ways included).
//
// What is NOT claimed: OpenSpiel is not in /root/reference, so its Go rules cannot be used or parity-checked.  The rules
// below are a plain stand-in with the same tensor geometry -- 9 x 9 board, 82 actions (81 points + pass), 4 planes (stones
// of the player to move, opponent's stones, empty points, a constant plane that tells who moves), positional superko reduced
// to simple ko, area scoring with komi 7.5 -- so that the host does the work a Go host does.  Host code only: plain C++17,
// std::thread, nothing but include/azhip.h from the product.
//
//   g++ -O2 -std=c++17 -Iinclude examples/host_stepped_go9.cpp -Lalphazero.jl_amd/csrc -lazhip -lpthread ...
//       -Wl,-rpath,$PWD/alphazero.jl_amd/csrc -o examples/host_stepped_go9      (or: make -C examples)
//   examples/host_stepped_go9 [--workers 512] [--sims 1600] [--seconds 8] [--threads 0=all usable] [--arena-gb 0=auto] [--fp32] [--blocks 10] [--filters 128]
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <random>
#include <thread>
#include <sched.h>
#include <sys/mman.h>
#include <unordered_map>
#include <vector>

#include "azhip.h"

namespace {
constexpr int N = 9, P = 81, A = 82, PASS = 81, C = 4;

struct Zob { uint64_t st[2][P], ko[P + 1], side; } ZB;
void init_zobrist() {
  std::mt19937_64 r(12345);
  for (auto& row : ZB.st) for (auto& v : row) v = r();
  for (auto& v : ZB.ko) v = r();
  ZB.side = r();
}

struct Go {
  int8_t b[P];          // 0 empty, 1 black, 2 white
  int8_t to_play;       // 1 black (moves first), 2 white
  int8_t ko;            // point forbidden by simple ko, -1 none
  int8_t passes;
  int16_t nmoves;
  uint64_t hash;
  void reset() { memset(b, 0, sizeof b); to_play = 1; ko = -1; passes = 0; nmoves = 0; hash = ZB.ko[P]; }
  bool over() const { return passes >= 2 || nmoves >= 2 * P; }
  static int nbrs(int p, int* out) {
    int n = 0, x = p % N, y = p / N;
    if (x > 0) out[n++] = p - 1;
    if (x < N - 1) out[n++] = p + 1;
    if (y > 0) out[n++] = p - N;
    if (y < N - 1) out[n++] = p + N;
    return n;
  }
  // stones of the group at p into `grp`; returns its number of liberties (capped: we only need 0 / 1 / more)
  int group(int p, int* grp, int* ng, uint8_t* mark, int stamp_unused = 0) const {
    (void)stamp_unused;
    const int col = b[p];
    int libs = 0, head = 0;
    bool seen_lib[P] = {false};
    *ng = 0;
    grp[(*ng)++] = p; mark[p] = 1;
    while (head < *ng) {
      int q = grp[head++], nb[4], k = nbrs(q, nb);
      for (int i = 0; i < k; ++i) {
        int r = nb[i];
        if (b[r] == 0) { if (!seen_lib[r]) { seen_lib[r] = true; ++libs; } }
        else if (b[r] == col && !mark[r]) { mark[r] = 1; grp[(*ng)++] = r; }
      }
    }
    for (int i = 0; i < *ng; ++i) mark[grp[i]] = 0;
    return libs;
  }
  bool legal(int a) const {
    if (a == PASS) return true;
    if (b[a] != 0 || a == ko) return false;
    int nb[4], k = nbrs(a, nb), grp[P], ng;
    uint8_t mark[P] = {0};
    const int me = to_play, opp = 3 - me;
    for (int i = 0; i < k; ++i) if (b[nb[i]] == 0) return true;                 // a liberty of its own
    Go t = *this;
    t.b[a] = (int8_t)me;
    for (int i = 0; i < k; ++i) if (t.b[nb[i]] == opp && t.group(nb[i], grp, &ng, mark) == 0) return true;   // captures
    return t.group(a, grp, &ng, mark) > 0;                                      // joins a group that keeps a liberty
  }
  void play(int a) {
    hash ^= ZB.ko[ko < 0 ? P : ko];
    ko = -1;
    if (a == PASS) { ++passes; }
    else {
      passes = 0;
      const int me = to_play, opp = 3 - me;
      b[a] = (int8_t)me; hash ^= ZB.st[me - 1][a];
      int nb[4], k = nbrs(a, nb), grp[P], ng, captured = 0, cap_at = -1;
      uint8_t mark[P] = {0};
      for (int i = 0; i < k; ++i)
        if (b[nb[i]] == opp && group(nb[i], grp, &ng, mark) == 0) {
          for (int j = 0; j < ng; ++j) { b[grp[j]] = 0; hash ^= ZB.st[opp - 1][grp[j]]; }
          captured += ng; cap_at = grp[0];
        }
      if (captured == 1) {                                                      // simple ko: a lone stone that took a lone stone
        int libs = group(a, grp, &ng, mark);
        if (ng == 1 && libs == 1) ko = (int8_t)cap_at;
      }
    }
    hash ^= ZB.ko[ko < 0 ? P : ko];
    to_play = (int8_t)(3 - to_play); hash ^= ZB.side;
    ++nmoves;
  }
  // area score from black's point of view minus komi: > 0 black wins
  double score() const {
    double s = -7.5;
    bool seen[P] = {false};
    for (int p = 0; p < P; ++p) {
      if (b[p] == 1) s += 1; else if (b[p] == 2) s -= 1;
      else if (!seen[p]) {
        int stack[P], n = 0, cnt = 0; bool tb = false, tw = false;
        stack[n++] = p; seen[p] = true;
        while (n) {
          int q = stack[--n], nb[4], k = nbrs(q, nb);
          ++cnt;
          for (int i = 0; i < k; ++i) {
            int r = nb[i];
            if (b[r] == 1) tb = true; else if (b[r] == 2) tw = true;
            else if (!seen[r]) { seen[r] = true; stack[n++] = r; }
          }
        }
        if (tb && !tw) s += cnt; else if (tw && !tb) s -= cnt;
      }
    }
    return s;
  }
  double white_reward() const { return !over() ? 0.0 : (score() > 0 ? 1.0 : -1.0); }   // "white" of the reference = first player = black stones
  void planes(float* X, float* mask) const {                                    // X: C x 9 x 9 (Julia W x H x C memory order)
    const int me = to_play, opp = 3 - me;
    for (int p = 0; p < P; ++p) {
      X[0 * P + p] = b[p] == me; X[1 * P + p] = b[p] == opp; X[2 * P + p] = b[p] == 0; X[3 * P + p] = me == 2 ? 1.f : 0.f;
    }
    for (int a = 0; a < A; ++a) mask[a] = legal(a) ? 1.f : 0.f;
  }
  uint64_t key() const { return hash; }
};

// StateInfo (mcts.jl:78-87), statistics by rank among the available actions: one record of 8 + 17 n bytes in the worker's
// slab (W, then P, N, the action ids).  Four std::vectors per node were four mallocs per simulation, and with several threads
// the page faults of a heap growing by half a gigabyte per second went through one lock: 10 s of system time in a 5 s run.
struct Node {
  int n; float Vest;
  double* W() { return (double*)(this + 1); }
  float* Pr() { return (float*)(W() + n); }
  int* Nv() { return (int*)(Pr() + n); }
  uint8_t* acts() { return (uint8_t*)(Nv() + n); }
  static size_t bytes(int n) { return (sizeof(Node) + (size_t)n * 17 + 7) & ~(size_t)7; }
};
static_assert(sizeof(Node) == 8, "header");
struct PathEntry { Node* nd; int i; double r; bool pswitch; };
// memory for the trees: slabs of 2 MB from one region that is mapped and touched before the clock starts (transparent huge
// pages where the kernel grants them); a worker keeps its slabs across games
struct SlabPool {
  static constexpr size_t SLAB = 2u << 20;
  char* base = nullptr; size_t nslabs = 0; std::atomic<size_t> next{0};
  void init(size_t bytes) {
    nslabs = std::max<size_t>(1, bytes / SLAB);
    base = (char*)aligned_alloc(SLAB, nslabs * SLAB);
    if (!base) { nslabs = 0; return; }
    madvise(base, nslabs * SLAB, MADV_HUGEPAGE);
  }
  char* get() {
    const size_t i = next.fetch_add(1);
    if (i < nslabs) return base + i * SLAB;
    return (char*)aligned_alloc(4096, SLAB);                       // past the region: the heap (page faults and all)
  }
} SLABS;
// the worker's Dict{State, StateInfo}: open addressing on the Zobrist key, grows by doubling
struct Table {
  std::vector<std::pair<uint64_t, Node*>> t;
  size_t used = 0;
  Table() : t(4096, {0, nullptr}) {}
  Node* find(uint64_t k) const {
    for (size_t i = (size_t)(k * 0x9E3779B97F4A7C15ull) & (t.size() - 1);; i = (i + 1) & (t.size() - 1)) {
      if (!t[i].second) return nullptr;
      if (t[i].first == k) return t[i].second;
    }
  }
  void put(uint64_t k, Node* nd) {
    if (2 * (used + 1) > t.size()) {
      std::vector<std::pair<uint64_t, Node*>> old(2 * t.size(), {0, nullptr});
      old.swap(t);
      used = 0;
      for (auto& e : old) if (e.second) put(e.first, e.second);
    }
    size_t i = (size_t)(k * 0x9E3779B97F4A7C15ull) & (t.size() - 1);
    while (t[i].second) i = (i + 1) & (t.size() - 1);
    t[i] = {k, nd}; ++used;
  }
  void clear() { std::fill(t.begin(), t.end(), std::pair<uint64_t, Node*>{0, nullptr}); used = 0; }
  size_t size() const { return used; }
};

struct Worker {
  Go root, leaf;
  Table tree;
  std::vector<char*> slabs; size_t slab_i = 0, slab_off = 0;       // bump allocation over the worker's slabs
  Node* new_node(int n) {
    const size_t b = Node::bytes(n);
    if (slabs.empty() || slab_off + b > SlabPool::SLAB) {
      if (!slabs.empty() && slab_off + b > SlabPool::SLAB) ++slab_i;
      if (slab_i >= slabs.size()) slabs.push_back(SLABS.get());
      slab_off = 0;
    }
    Node* nd = (Node*)(slabs[slab_i] + slab_off);
    slab_off += b;
    nd->n = n;
    return nd;
  }
  void clear_tree() { tree.clear(); slab_i = 0; slab_off = 0; }
  std::vector<PathEntry> path;
  std::vector<double> eta;
  std::mt19937_64 rng;
  int sims_in_move = 0, leaf_kind = 0;      // 0 none, 1 new, 2 terminal
  int batch_index = -1;
  long long sims = 0, traversed = 0, games = 0, moves = 0;
  void new_noise(double alpha) {
    int n = 0;
    for (int a = 0; a < A; ++a) n += root.legal(a);
    eta.resize(n);
    std::gamma_distribution<double> g(alpha, 1.0);
    double s = 0;
    for (auto& e : eta) { e = g(rng); s += e; }
    for (auto& e : eta) e /= s;
  }
};

struct Opt { bool dry = false; int workers = 512, sims = 1600, threads = 0, blocks = 10, filters = 128, bf16 = 1; double seconds = 8.0, cpuct = 2.0, eps = 0.25, alpha = 0.03, arena_gb = 0.0; };

// select (mcts.jl:199-217) until an unseen or a terminal state
void descend(Worker& w, const Opt& o) {
  Go g = w.root;
  w.path.clear();
  bool root = true;
  for (;;) {
    if (g.over()) { w.leaf_kind = 2; w.leaf = g; return; }
    Node* nd = w.tree.find(g.key());
    if (!nd) { w.leaf_kind = 1; w.leaf = g; return; }
    const int n = nd->n;
    const double* Wv = nd->W(); const float* Pr = nd->Pr(); const int* Nv = nd->Nv();
    long long ntot = 0;
    for (int i = 0; i < n; ++i) ntot += Nv[i];
    const double sq = std::sqrt((double)ntot);
    int best = 0; double bs = -1e300;
    for (int i = 0; i < n; ++i) {
      const double Q = Wv[i] / (double)std::max(Nv[i], 1);
      double Pd = (double)Pr[i];
      if (root && o.eps != 0.0 && (size_t)i < w.eta.size()) Pd = (1.0 - o.eps) * Pd + o.eps * w.eta[i];
      const double sc = Q + o.cpuct * Pd * sq / (double)(Nv[i] + 1);
      if (sc > bs) { bs = sc; best = i; }
    }
    const bool wp = g.to_play == 1;
    g.play(nd->acts()[best]);
    const double wr = g.white_reward();
    w.path.push_back({nd, best, wp ? wr : -wr, wp != (g.to_play == 1)});
    root = false;
  }
}
// init_state_info + the unwinding of run_simulation! (mcts.jl:157-161, 218-223)
void expand_backup(Worker& w, const float* Pb, const float* Vb) {
  double q = 0.0;
  if (w.leaf_kind == 1) {
    const float* Pl = Pb + (size_t)w.batch_index * A;
    uint8_t legal[A]; int n = 0;
    for (int a = 0; a < A; ++a) if (w.leaf.legal(a)) legal[n++] = (uint8_t)a;
    Node* nd = w.new_node(n);
    for (int i = 0; i < n; ++i) { nd->W()[i] = 0.0; nd->Pr()[i] = Pl[legal[i]]; nd->Nv()[i] = 0; nd->acts()[i] = legal[i]; }
    nd->Vest = Vb[w.batch_index];
    q = (double)nd->Vest;
    w.tree.put(w.leaf.key(), nd);
  }
  for (int k = (int)w.path.size() - 1; k >= 0; --k) {
    const PathEntry& e = w.path[k];
    q = e.r + (e.pswitch ? -q : q);
    e.nd->W()[e.i] += q; e.nd->Nv()[e.i] += 1;
  }
  w.traversed += (long long)w.path.size();
  w.sims += 1;
}
// policy + move (mcts.jl:255-271, play.jl:308-313): temperature 1 for the first 20 moves, then the most visited
void make_move(Worker& w, const Opt& o) {
  Node* nd = w.tree.find(w.root.key());
  int a = PASS;
  if (nd) {
    const int n = nd->n; const int* Nv = nd->Nv();
    long long tot = 0;
    for (int i = 0; i < n; ++i) tot += Nv[i];
    if (tot > 0) {
      if (w.root.nmoves < 20) {
        std::uniform_real_distribution<double> u(0.0, 1.0);
        double x = u(w.rng) * (double)tot, c = 0;
        int i = 0;
        for (; i + 1 < n; ++i) { c += Nv[i]; if (c > x) break; }
        a = nd->acts()[i];
      } else a = nd->acts()[std::max_element(Nv, Nv + n) - Nv];
    }
  }
  w.root.play(a);
  w.moves++;
  if (w.root.over()) { w.root.reset(); w.clear_tree(); w.games++; }     // reset_every = 1
  w.new_noise(o.alpha);
  w.sims_in_move = 0;
}

// Persistent worker threads.  A wave has two short parallel phases (select + encode, expand + backup + move), each a few
// hundred microseconds of work: the threads spin on a generation counter for a while before they sleep, and the last one
// out signals through an atomic, so a phase costs microseconds of synchronisation, not a mutex hand-over per thread
// (the first version woke hardware_concurrency() = 256 threads through one mutex four times per wave: 6 ms per phase).
class Pool {
 public:
  explicit Pool(int n) : n_(n) { for (int t = 1; t < n; ++t) th_.emplace_back([this] { loop(); }); }
  ~Pool() { stop_.store(true); { std::lock_guard<std::mutex> l(m_); gen_.fetch_add(1); } cv_.notify_all(); for (auto& t : th_) t.join(); }
  template <class F> void run(int n, F f) {
    if (n_ <= 1) { for (int i = 0; i < n; ++i) f(i); return; }
    std::function<void(int)> fn = f;
    fn_ = &fn; total_ = n; next_.store(0); busy_.store(n_ - 1);
    { std::lock_guard<std::mutex> l(m_); gen_.fetch_add(1, std::memory_order_release); }
    cv_.notify_all();
    work();
    for (int spins = 0; busy_.load(std::memory_order_acquire) != 0; ++spins) if (spins > 2000) std::this_thread::yield();
    fn_ = nullptr;
  }
 private:
  void work() { for (int i; (i = next_.fetch_add(4)) < total_;) for (int j = i; j < std::min(total_, i + 4); ++j) (*fn_)(j); }
  void loop() {
    unsigned long long seen = 0;
    for (;;) {
      int spins = 0;
      while (gen_.load(std::memory_order_acquire) == seen) {
        if (++spins < 20000) { __builtin_ia32_pause(); continue; }
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return gen_.load() != seen; });
      }
      seen = gen_.load();
      if (stop_.load()) return;
      work();
      busy_.fetch_sub(1, std::memory_order_release);
    }
  }
  int n_, total_ = 0;
  std::atomic<int> busy_{0};
  std::atomic<bool> stop_{false};
  std::atomic<unsigned long long> gen_{0};
  std::atomic<int> next_{0};
  std::function<void(int)>* fn_ = nullptr;
  std::mutex m_;
  std::condition_variable cv_;
  std::vector<std::thread> th_;
};
// one helper thread that sits in the blocking network call while the pool works on the other half of the workers
class Helper {
 public:
  Helper() : th_([this] { loop(); }) {}
  ~Helper() { { std::lock_guard<std::mutex> l(m_); stop_ = true; ++posted_; } cv_.notify_one(); th_.join(); }
  template <class F> void start(F f) { { std::lock_guard<std::mutex> l(m_); fn_ = f; ++posted_; } cv_.notify_one(); }
  void wait() { for (int spins = 0; done_.load(std::memory_order_acquire) != posted_; ++spins) if (spins > 2000) std::this_thread::yield(); }
 private:
  void loop() {
    unsigned long long seen = 0;
    for (;;) {
      { std::unique_lock<std::mutex> l(m_); cv_.wait(l, [&] { return posted_ != seen; }); seen = posted_; if (stop_) return; }
      fn_();
      done_.store(seen, std::memory_order_release);
    }
  }
  std::function<void()> fn_;
  unsigned long long posted_ = 0;
  std::atomic<unsigned long long> done_{0};
  bool stop_ = false;
  std::mutex m_;
  std::condition_variable cv_;
  std::thread th_;
};
// CPUs this process may run on (a container's affinity mask, not the machine's core count), at most 64 threads
int usable_threads() {
  cpu_set_t set;
  int n = 0;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
  if (n <= 0) n = (int)std::max(1u, std::thread::hardware_concurrency());
  return std::min(n, 64);
}
double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

int main(int argc, char** argv) {
  Opt o;
  for (int i = 1; i < argc; ++i) {
    auto is = [&](const char* s) { return !strcmp(argv[i], s) && i + 1 < argc; };
    if (is("--workers")) o.workers = atoi(argv[++i]); else if (is("--sims")) o.sims = atoi(argv[++i]);
    else if (is("--seconds")) o.seconds = atof(argv[++i]); else if (is("--threads")) o.threads = atoi(argv[++i]);
    else if (is("--arena-gb")) o.arena_gb = atof(argv[++i]);
    else if (is("--blocks")) o.blocks = atoi(argv[++i]); else if (is("--filters")) o.filters = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--fp32")) o.bf16 = 0;
    else if (!strcmp(argv[i], "--dry")) o.dry = true;            // no GPU: MCTS.RandomOracle in place of the network (host ceiling, CPU tests)
    else { fprintf(stderr, "unknown option %s\n", argv[i]); return 1; }
  }
  // a phase is workers x ~10 us of work: beyond ~64 workers per thread the synchronisation costs more than the threads bring
  // (1024 workers on the 256-CPU MI355X host: 16 threads 0.98 M sims/s, 32: 0.86 M, 64: 0.37 M)
  if (o.threads <= 0) o.threads = std::max(1, std::min({usable_threads(), 16, o.workers / 64}));
  init_zobrist();
  az_engine_cfg cfg;
  az_engine* e = nullptr;
  if (az_engine_cfg_init(&cfg) != AZ_OK) return 1;
  cfg.game = AZ_GAME_GO9_PLANES; cfg.oracle = AZ_ORACLE_RESNET; cfg.num_workers = 1; cfg.batch_size = 1; cfg.num_iters_per_turn = 2;
  cfg.num_blocks = o.blocks; cfg.num_filters = o.filters; cfg.num_policy_head_filters = 32; cfg.num_value_head_filters = 32; cfg.net_bf16 = o.bf16;
  if (o.workers < 2) o.workers = 2;
  int st = o.dry ? AZ_OK : az_engine_create(&cfg, &e);
  if (st != AZ_OK) { fprintf(stderr, "az_engine_create: status %d: %s\n", st, az_last_error()); return st == AZ_ERR_HIP ? 2 : 1; }
  int64_t np = 0;
  if (!o.dry) az_net_num_params(e, &np);
  if (!o.dry) {
    // synthetic weights in the blob layout of include/azhip.h: uniform(-s, s) convolutions / dense layers, batch norm near identity
    std::vector<float> blob((size_t)np);
    std::mt19937 r(2026);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    const int F = o.filters;
    size_t k = 0;
    auto fill = [&](size_t n, float s) { for (size_t i = 0; i < n; ++i) blob[k++] = s * u(r); };
    auto zeros = [&](size_t n) { for (size_t i = 0; i < n; ++i) blob[k++] = 0.f; };
    auto bn = [&](size_t n) { for (size_t i = 0; i < n; ++i) blob[k++] = 1.f + 0.1f * u(r); fill(n, 0.1f); fill(n, 0.1f); for (size_t i = 0; i < n; ++i) blob[k++] = 1.f + 0.1f * u(r); };
    auto conv = [&](int ks, int cin, int cout) { fill((size_t)ks * ks * cin * cout, std::sqrt(6.f / (float)(ks * ks * (cin + cout)))); zeros(cout); bn(cout); };
    conv(3, C, F);
    for (int b = 0; b < 2 * o.blocks; ++b) conv(3, F, F);
    conv(1, F, 32); fill((size_t)A * P * 32, std::sqrt(6.f / (float)(A + P * 32))); zeros(A);
    conv(1, F, 32); fill((size_t)F * P * 32, std::sqrt(6.f / (float)(F + P * 32))); zeros(F); fill(F, std::sqrt(6.f / (float)(F + 1))); zeros(1);
    if ((int64_t)k != np) { fprintf(stderr, "blob layout mismatch: %zu vs %lld\n", k, (long long)np); return 1; }
    if ((st = az_net_set_params(e, blob.data(), np)) != AZ_OK) { fprintf(stderr, "az_net_set_params: %s\n", az_last_error()); return 1; }
  }
  const int W = o.workers;
  Pool pool(o.threads);
  if (o.arena_gb <= 0.0) {
    // 2 M simulations/s x 1.4 KB per node = 2.8 GB/s of tree: enough for the run (trees are only reset when a game ends), at most a
    // quarter of what the host has available
    o.arena_gb = 4.0 + 3.5 * (o.seconds + std::min(2.0, 0.25 * o.seconds));
    double avail_gb = 16.0;
    if (FILE* f = fopen("/proc/meminfo", "r")) {
      char line[256];
      while (fgets(line, sizeof line, f)) { long long kb; if (sscanf(line, "MemAvailable: %lld kB", &kb) == 1) avail_gb = (double)kb / 1048576.0; }
      fclose(f);
    }
    o.arena_gb = std::min(o.arena_gb, 0.25 * avail_gb);
  }
  SLABS.init((size_t)(o.arena_gb * 1073741824.0));
  pool.run((int)SLABS.nslabs, [&](int i) { char* q = SLABS.base + (size_t)i * SlabPool::SLAB; for (size_t k = 0; k < SlabPool::SLAB; k += 4096) q[k] = 0; });
  std::vector<Worker> ws((size_t)W);
  for (int i = 0; i < W; ++i) { ws[i].rng.seed(1000 + i); ws[i].root.reset(); ws[i].new_noise(o.alpha); }
  // The workers form two halves that take turns: while az_net_forward evaluates the leaves of one half (a helper thread sits in
  // the blocking call), the host threads expand / back up / select the other half -- the seam's two sides overlap instead of
  // adding up (1024 workers in one batch: host 51 % + network 49 % of the wall time).
  const int Wh = W / 2;
  std::vector<float> X[2], M[2], Pb[2], Vb[2], Pinv[2];
  for (int h = 0; h < 2; ++h) { X[h].resize((size_t)(W - Wh) * C * P); M[h].resize((size_t)(W - Wh) * A); Pb[h].resize((size_t)(W - Wh) * A); Vb[h].resize(W - Wh); Pinv[h].resize(W - Wh); }
  int nb[2] = {0, 0};
  const int lo[2] = {0, Wh}, cnt[2] = {Wh, W - Wh};
  double t_tree = 0, t_net = 0, t_move = 0, t_call = 0;
  long long launches = 0, boards = 0;
  std::atomic<int> fwd_status{AZ_OK};
  // select (and the new leaves encode themselves into the half's batch: its order is arbitrary, evaluations are independent)
  auto select = [&](int h) {
    std::atomic<int> nba{0};
    pool.run(cnt[h], [&](int i) {
      Worker& w = ws[lo[h] + i];
      descend(w, o);
      w.batch_index = w.leaf_kind == 1 ? nba.fetch_add(1) : -1;
      if (w.batch_index >= 0) w.leaf.planes(&X[h][(size_t)w.batch_index * C * P], &M[h][(size_t)w.batch_index * A]);
    });
    nb[h] = nba.load();
  };
  // expand + backup, and the move when the search of this ply is complete
  auto finish = [&](int h) {
    pool.run(cnt[h], [&](int i) { Worker& w = ws[lo[h] + i]; expand_backup(w, Pb[h].data(), Vb[h].data()); if (++w.sims_in_move >= o.sims) make_move(w, o); });
  };
  double call_s[2] = {0, 0};
  auto forward = [&](int h) {
    const double c0 = now();
    if (nb[h] && o.dry) {
      for (int i = 0; i < nb[h]; ++i) {
        float n = 0;
        for (int a = 0; a < A; ++a) n += M[h][(size_t)i * A + a];
        for (int a = 0; a < A; ++a) Pb[h][(size_t)i * A + a] = M[h][(size_t)i * A + a] / n;
        Vb[h][i] = 0.f;
      }
    } else if (nb[h]) {
      const int r = az_net_forward(e, X[h].data(), M[h].data(), nb[h], Pb[h].data(), Vb[h].data(), Pinv[h].data());
      if (r != AZ_OK) { fprintf(stderr, "az_net_forward: %s\n", az_last_error()); fwd_status.store(r); }
    }
    call_s[h] = now() - c0;
  };
  Helper helper;
  const double t_start = now();
  double t_meas0 = 0;
  long long sims0 = 0, trav0 = 0;
  bool warm = true, pending1 = false;
  select(0);
  for (;;) {
    const double t0 = now();
    double host = 0, wait = 0;
    for (int h = 0; h < 2; ++h) {
      // the network on half h, the host on the other half
      helper.start([&, h] { forward(h); });
      const double a0 = now();
      if (h == 0) { if (pending1) finish(1); if (cnt[1]) select(1); pending1 = cnt[1] > 0; }
      else { finish(0); select(0); }
      const double a1 = now();
      helper.wait();
      const double a2 = now();
      host += a1 - a0; wait += a2 - a1;
      if (nb[h]) { ++launches; boards += nb[h]; t_call += call_s[h]; }
      if (fwd_status.load() != AZ_OK) return 1;
    }
    const double t4 = now();
    (void)t0;
    if (warm) {
      if (t4 - t_start > std::min(2.0, 0.25 * o.seconds)) {
        warm = false; t_meas0 = t4; t_tree = t_net = t_move = t_call = 0; launches = boards = 0;
        sims0 = trav0 = 0;
        for (auto& w : ws) { sims0 += w.sims; trav0 += w.traversed; }
      }
      continue;
    }
    t_tree += host; t_net += wait;
    if (t4 - t_meas0 >= o.seconds) break;
  }
  const double wall = now() - t_meas0;
  long long sims = -sims0, trav = -trav0, games = 0, moves = 0;
  size_t nodes = 0;
  for (auto& w : ws) { sims += w.sims; trav += w.traversed; games += w.games; moves += w.moves; nodes = std::max(nodes, w.tree.size()); }
  char kernel[96] = "";
  if (!o.dry) az_net_last_kernel(e, kernel, sizeof kernel); else snprintf(kernel, sizeof kernel, "none (--dry: uniform oracle on the host)");
  printf("{\"workload\": \"9x9 Go-shaped host game (stand-in rules, 82 actions, 9x9x4 planes), host tree in C++ on %d threads, %d workers, %d sims/move, ResNet %dx%d %s through az_net_forward\", "
         "\"value\": %.1f, \"unit\": \"sims/s\", \"seconds\": %.2f, \"host_tree_share\": %.3f, \"network_wait_share\": %.3f, \"network_busy_share\": %.3f, "
         "\"boards_per_launch\": %.1f, \"launches\": %lld, \"network_boards_per_s_while_in_call\": %.1f, \"avg_exploration_depth\": %.2f, "
         "\"games_finished\": %lld, \"moves\": %lld, \"largest_tree_nodes\": %zu, \"kernel\": \"%s\", \"threads\": %d, "
         "\"rules\": \"host stand-in, NOT OpenSpiel's (absent from the reference tree): no parity claim\"}\n",
         o.threads, W, o.sims, o.blocks, o.filters, o.bf16 ? "bf16" : "fp32", sims / wall, wall, t_tree / wall, t_net / wall, t_call / wall,
         launches ? (double)boards / launches : 0.0, launches, t_call > 0 ? boards / t_call : 0.0, sims ? (double)trav / sims : 0.0,
         games, moves, nodes, kernel, o.threads);
  if (e) az_engine_destroy(e);
  return 0;
}
