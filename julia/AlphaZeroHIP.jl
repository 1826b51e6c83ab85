# AlphaZeroHIP.jl -- the Julia side of the drop-in: `ccall` bindings to libazhip.so (include/azhip.h).
#
# WRITTEN BLIND: the build container has no Julia (SURVEY.md §0), so this file has never been run.
# Every call it makes is mirrored one-to-one by the Python binding (alphazero.jl_amd/azhip/_lib.py,
# engine.py, simulations.py), which IS exercised by tests/ on a real MI355X.  Keep the two in sync.
#
# What it provides (SURVEY.md §8b):
#   seam 3  HipResNet <: AbstractNetwork      the Network plugin backed by the HIP tower/heads kernels
#   seam 2  Network.forward_normalized / evaluate_batch on a HipResNet (stock Julia MCTS on top of it)
#   seam 1  AlphaZero.simulate(simulator, gspec::DeviceGameSpec, p) -> whole self-play phase on the GPU
#           AlphaZero.simulate_distributed(simulator, gspec::DeviceGameSpec, p) -> one worker process per GPU, global game ids
#           device_self_play_step!(gspec, nn, params, mem, comm) -> self_play_step! with records and samples resident in HBM,
#           az_comm_gather_push (RCCL) between the ranks
#   seam 5  DeviceMctsEnv: explore! / state_info / counters / reset! on a device tree (Explorer's MCTS statistics)
# `Scripts.train` is unchanged: pick `HipResNet` as the experiment's network type.
module AlphaZeroHIP

using AlphaZero
using AlphaZero: GI, Network, MctsPlayer, MctsParams, SimParams, Simulator, Trace, ConstSchedule, PLSchedule
import AlphaZero.Examples: ConnectFour, Tictactoe, Mancala

const LIB = get(ENV, "AZHIP_LIB", "libazhip.so")
const MAX_ACTIONS = 9

# ---- az_engine_cfg / az_move_rec / az_game_rec / az_trace_buf / az_selfplay_stats (isbits mirrors) ----
struct EngineCfg
  struct_size::Int32; device::Int32; game::Int32; oracle::Int32
  gamma::Float64; cpuct::Float64; dirichlet_noise_eps::Float64; dirichlet_noise_alpha::Float64
  prior_temperature::Float64
  num_iters_per_turn::Int32; temperature_len::Int32
  temperature_xs::NTuple{8,Int32}; temperature_ys::NTuple{8,Float64}
  num_workers::Int32; batch_size::Int32; reset_every::Int32; fill_batches::Int32
  flip_probability::Float64; seed::UInt64
  max_nodes_per_slot::Int32; max_moves_per_game::Int32
  num_blocks::Int32; num_filters::Int32; num_policy_head_filters::Int32; num_value_head_filters::Int32
  net_bf16::Int32
  lock_step::Int32   # 0 = free-running workers (the reference's own schedule: util.jl:181-188), 1 = rounds in lock step
end
struct MoveRec
  key::NTuple{2,UInt64}; N::NTuple{10,Int32}; action::Int32; reward::Float32
end
struct GameRec
  game_id::Int32; slot::Int32; num_moves::Int32; first_move::Int32
  nodes::Int64; total_simulations::Int64; total_nodes_traversed::Int64; final_key::NTuple{2,UInt64}
end
mutable struct TraceBuf
  games::Ptr{GameRec}; games_cap::Int64; num_games::Int64
  moves::Ptr{MoveRec}; moves_cap::Int64; num_moves::Int64
end
mutable struct SelfplayStats
  simulations::Int64; nodes_traversed::Int64; leaf_evals::Int64; moves::Int64; games::Int64; waves::Int64
  seconds::Float64
  aborted_games::Int64
  tower_fallbacks::Int64
  evals_reused::Int64
  slot_launches::Int64
  SelfplayStats() = new(0, 0, 0, 0, 0, 0, 0.0, 0, 0, 0, 0)
end
@assert sizeof(EngineCfg) == 224 && sizeof(MoveRec) == 64 && sizeof(GameRec) == 56 && sizeof(SelfplayStats) == 88
const ABI_VERSION = 4   # include/azhip.h AZ_ABI_VERSION the structs above are written against
"Called once before the first engine is created: a library built from another header must not be written into these structs."
function check_abi()
  v = ccall((:az_abi_version, LIB), Cint, ())
  v == ABI_VERSION || error("libazhip.so has ABI version $v, this glue is written against $ABI_VERSION")
  for (which, T) in ((0, EngineCfg), (1, MoveRec), (2, GameRec), (3, TraceBuf), (4, SelfplayStats))
    n = ccall((:az_abi_struct_size, LIB), Cint, (Int32,), Int32(which))
    n == sizeof(T) || error("$T: the library's struct is $n bytes, the glue's $(sizeof(T))")
  end
end

last_error() = unsafe_string(ccall((:az_last_error, LIB), Cstring, ()))
check(status::Integer) = status == 0 ? nothing : error("azhip status $status: $(last_error())")

# ---- device game twins: packed 16-byte keys (include/azhip.h "State keys") -------------------------
const DeviceGameSpec = Union{ConnectFour.GameSpec, Tictactoe.GameSpec, Mancala.GameSpec}
game_id(::ConnectFour.GameSpec) = Int32(0)
game_id(::Tictactoe.GameSpec) = Int32(1)
game_id(::Mancala.GameSpec) = Int32(2)
"Games without a device twin (e.g. OpenSpiel games through src/openspiel.jl) can still use the HIP network when their
tensor geometry is one the library instantiates: AZ_GAME_GO9_PLANES = 9 x 9 x 4 planes, 82 actions (OpenSpiel 9x9 Go).
Only `HipResNet` / `Network.forward_normalized` work for them (seam 2); `simulate` falls back to the stock Julia loop."
function game_id(gspec::AlphaZero.AbstractGameSpec)
  GI.state_dim(gspec) == (9, 9, 4) && GI.num_actions(gspec) == 82 && return Int32(3)
  error("no device twin and no matching plane geometry for $(typeof(gspec))")
end
const BLACK_BIT = UInt64(1) << 63

function encode_state(::ConnectFour.GameSpec, s)
  a = UInt64(0); b = UInt64(0)
  for col in 1:7, row in 1:6
    c = s.board[col, row]
    bit = UInt64(1) << ((col - 1) * 7 + (row - 1))
    c == ConnectFour.WHITE && (a |= bit)
    c == ConnectFour.BLACK && (b |= bit)
  end
  s.curplayer == ConnectFour.BLACK && (a |= BLACK_BIT)
  return (a, b)
end
function decode_state(::ConnectFour.GameSpec, k)
  a, b = k
  cells = [((a >> ((col-1)*7 + row-1)) & 1 == 1) ? ConnectFour.WHITE :
           ((b >> ((col-1)*7 + row-1)) & 1 == 1) ? ConnectFour.BLACK : ConnectFour.EMPTY
           for col in 1:7, row in 1:6]
  return (board=ConnectFour.Board(cells), curplayer=(a & BLACK_BIT != 0) ? ConnectFour.BLACK : ConnectFour.WHITE)
end
function encode_state(::Tictactoe.GameSpec, s)
  a = UInt64(0); b = UInt64(0)
  for pos in 1:9
    c = s.board[pos]
    c === true && (a |= UInt64(1) << (pos - 1))
    c === false && (b |= UInt64(1) << (pos - 1))
  end
  s.curplayer == Tictactoe.BLACK && (a |= BLACK_BIT)
  return (a, b)
end
function decode_state(::Tictactoe.GameSpec, k)
  a, b = k
  cells = Tictactoe.Cell[((a >> (p-1)) & 1 == 1) ? true : ((b >> (p-1)) & 1 == 1) ? false : nothing for p in 1:9]
  return (board=Tictactoe.Board(cells), curplayer=(a & BLACK_BIT == 0))
end
function encode_state(::Mancala.GameSpec, s)
  a = UInt64(0); b = UInt64(0)
  for n in 1:6
    a |= UInt64(s.board.houses[1, n]) << (8 * (n - 1))
    b |= UInt64(s.board.houses[2, n]) << (8 * (n - 1))
  end
  a |= UInt64(s.board.stores[1]) << 48
  b |= UInt64(s.board.stores[2]) << 48
  s.curplayer == Mancala.BLACK && (a |= BLACK_BIT)
  return (a, b)
end
function decode_state(::Mancala.GameSpec, k)
  a, b = k
  houses = [UInt8(((p == 1 ? a : b) >> (8 * (n - 1))) & 0xff) for p in 1:2, n in 1:6]
  stores = [UInt8((a >> 48) & 0xff), UInt8((b >> 48) & 0xff)]
  board = Mancala.Board(Mancala.SVector{2,UInt8}(stores), Mancala.SMatrix{2,6,UInt8,12}(houses))
  return (board=board, curplayer=(a & BLACK_BIT != 0) ? Mancala.BLACK : Mancala.WHITE)
end

# ---- engine handle ------------------------------------------------------------------------------------
mutable struct Engine
  h::Ptr{Cvoid}
  function Engine(cfg::EngineCfg)
    check_abi()
    out = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:az_engine_create, LIB), Cint, (Ref{EngineCfg}, Ref{Ptr{Cvoid}}), cfg, out))
    e = new(out[])
    finalizer(x -> (x.h != C_NULL && ccall((:az_engine_destroy, LIB), Cint, (Ptr{Cvoid},), x.h); x.h = C_NULL), e)
    return e
  end
end

schedule_points(s::ConstSchedule) = (Int32[0], Float64[s.value])
schedule_points(s::PLSchedule) = (Int32.(s.xs), Float64.(s.ys))
pad8(v, T) = ntuple(i -> i <= length(v) ? T(v[i]) : zero(T), 8)

"MctsParams + SimParams + ResNetHP -> az_engine_cfg (SURVEY.md §8b config mapping)"
function make_cfg(gspec, mcts::MctsParams, sim::SimParams, hp; oracle=2, seed=1, device=0, arena=false, bf16=false, lock_step=false)
  xs, ys = schedule_points(mcts.temperature)
  EngineCfg(Int32(sizeof(EngineCfg)), device, game_id(gspec), oracle,
    mcts.gamma, mcts.cpuct, mcts.dirichlet_noise_ϵ, mcts.dirichlet_noise_α, mcts.prior_temperature,
    mcts.num_iters_per_turn, length(xs), pad8(xs, Int32), pad8(ys, Float64),
    sim.num_workers, sim.batch_size, isnothing(sim.reset_every) ? 0 : sim.reset_every, sim.fill_batches ? 1 : 0,
    sim.flip_probability, UInt64(seed), 0, 0,
    hp.num_blocks, hp.num_filters, hp.num_policy_head_filters, hp.num_value_head_filters, bf16 ? 1 : 0, lock_step ? 1 : 0)
end

# ---- seam 3: the network plugin -------------------------------------------------------------------------
"""
    HipResNet(gspec, hyper::ResNetHP)  /  HipResNet(nn::ResNet)

Network plugin whose forward pass is the HIP tower (fp32 MFMA).  Parameters live in one Float32 blob in
Flux array order (see az_net_num_params in include/azhip.h); `HipResNet(nn::ResNet)` imports a trained Flux
network.  Training is delegated: keep a Flux `ResNet` as `curnn` and convert after each learning step.
"""
mutable struct HipResNet <: AlphaZero.AbstractNetwork
  gspec
  hyper::AlphaZero.ResNetHP
  blob::Vector{Float32}
  engine::Union{Nothing, Engine}
  on_gpu::Bool
end
Network.HyperParams(::Type{HipResNet}) = AlphaZero.ResNetHP
Network.hyperparams(nn::HipResNet) = nn.hyper
Network.game_spec(nn::HipResNet) = nn.gspec
Network.on_gpu(nn::HipResNet) = nn.on_gpu
Network.to_gpu(nn::HipResNet) = (nn.on_gpu = true; nn)
Network.to_cpu(nn::HipResNet) = (nn.on_gpu = false; nn)
Network.set_test_mode!(::HipResNet, mode) = nothing          # inference only: BatchNorm is always test mode
Network.convert_input(::HipResNet, x) = x
Network.convert_output(::HipResNet, x) = x
Network.params(nn::HipResNet) = (nn.blob,)
Network.regularized_params(nn::HipResNet) = ()
Network.gc(nn::HipResNet) = (nn.engine = nothing; GC.gc(true))
Base.copy(nn::HipResNet) = HipResNet(nn.gspec, nn.hyper, copy(nn.blob), nothing, nn.on_gpu)
Network.train!(cb, ::HipResNet, opt, loss, data, n) =
  error("HipResNet is inference-only: train a Flux ResNet and convert it with HipResNet(nn)")

flat(x) = vec(Float32.(Array(x)))
bn_blob(bn) = vcat(flat(bn.γ), flat(bn.β), flat(bn.μ), flat(bn.σ²))
conv_blob(c) = vcat(flat(c.weight), flat(c.bias))
"Flatten a Flux ResNet (resnet.jl:65-92) into the blob order of az_net_set_params."
function HipResNet(nn::AlphaZero.ResNet)
  parts = Vector{Float32}[]
  push!(parts, conv_blob(nn.common[1]), bn_blob(nn.common[2]))
  for i in 1:nn.hyper.num_blocks
    inner = nn.common[2 + i][1].layers          # Chain(SkipConnection(Chain(conv, bn, conv, bn), +), relu)
    push!(parts, conv_blob(inner[1]), bn_blob(inner[2]), conv_blob(inner[3]), bn_blob(inner[4]))
  end
  ph, vh = nn.phead, nn.vhead
  push!(parts, conv_blob(ph[1]), bn_blob(ph[2]), flat(ph[4].weight), flat(ph[4].bias))
  push!(parts, conv_blob(vh[1]), bn_blob(vh[2]), flat(vh[4].weight), flat(vh[4].bias), flat(vh[5].weight), flat(vh[5].bias))
  return HipResNet(nn.gspec, nn.hyper, reduce(vcat, parts), nothing, true)
end

"""
    flux_resnet(nn::HipResNet) -> AlphaZero.ResNet

The inverse of `HipResNet(::ResNet)`: a Flux ResNet (resnet.jl:65-92) whose arrays are filled from the blob, e.g. after
`device_batch_updates!`, so that sessions / checkpoints (`src/ui/session.jl:92-118`) keep serialising a stock network.
"""
function flux_resnet(nn::HipResNet)
  out = AlphaZero.ResNet(nn.gspec, nn.hyper)
  off = Ref(0)
  take!(a) = (n = length(a); copyto!(a, reshape(view(nn.blob, off[]+1 : off[]+n), size(a))); off[] += n)
  conv!(c) = (take!(c.weight); take!(c.bias))
  bn!(b) = (take!(b.γ); take!(b.β); take!(b.μ); take!(b.σ²))
  conv!(out.common[1]); bn!(out.common[2])
  for i in 1:nn.hyper.num_blocks
    inner = out.common[2 + i][1].layers
    conv!(inner[1]); bn!(inner[2]); conv!(inner[3]); bn!(inner[4])
  end
  ph, vh = out.phead, out.vhead
  conv!(ph[1]); bn!(ph[2]); take!(ph[4].weight); take!(ph[4].bias)
  conv!(vh[1]); bn!(vh[2]); take!(vh[4].weight); take!(vh[4].bias); take!(vh[5].weight); take!(vh[5].bias)
  @assert off[] == length(nn.blob)
  return out
end

"Raw checkpoint of the parameter blob: magic, game id, ResNetHP fields, count, Float32 little-endian values (same file as azhip.network.save_params)"
function save_params(path, nn::HipResNet)
  open(path, "w") do io
    write(io, b"AZHIPNET"); write(io, Int32(game_id(nn.gspec)), Int32(nn.hyper.num_blocks), Int32(nn.hyper.num_filters),
      Int32(nn.hyper.num_policy_head_filters), Int32(nn.hyper.num_value_head_filters), Int64(length(nn.blob)))
    write(io, nn.blob)
  end
end

function engine!(nn::HipResNet)
  if isnothing(nn.engine)
    mcts = MctsParams(num_iters_per_turn=2, dirichlet_noise_ϵ=0., dirichlet_noise_α=1.)
    sim = SimParams(num_games=1, num_workers=1, batch_size=1)
    nn.engine = Engine(make_cfg(nn.gspec, mcts, sim, nn.hyper))
    check(ccall((:az_net_set_params, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64), nn.engine.h, nn.blob, length(nn.blob)))
  end
  return nn.engine
end

# seam 2: Network.forward_normalized (network.jl:264-271); X is W x H x C x N, A is nA x N (Float32)
function Network.forward_normalized(nn::HipResNet, X, A)
  N = size(X)[end]; nA = size(A, 1)
  P = Matrix{Float32}(undef, nA, N); V = Matrix{Float32}(undef, 1, N); Pinv = Matrix{Float32}(undef, 1, N)
  check(ccall((:az_net_forward, LIB), Cint,
    (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Int32, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
    engine!(nn).h, Array{Float32}(X), Array{Float32}(A), N, P, V, Pinv))
  return (P, V, Pinv)
end
Network.forward(nn::HipResNet, X) =
  Network.forward_normalized(nn, X, ones(Float32, GI.num_actions(nn.gspec), size(X)[end]))[1:2]

# Network.evaluate_batch (network.jl:308-315) with the state encode fused on the device
function Network.evaluate_batch(nn::HipResNet, batch)
  gspec = nn.gspec; N = length(batch); nA = GI.num_actions(gspec)
  keys = Vector{NTuple{2,UInt64}}([encode_state(gspec, s) for s in batch])
  P = Matrix{Float32}(undef, nA, N); V = Vector{Float32}(undef, N)
  check(ccall((:az_net_evaluate_keys, LIB), Cint, (Ptr{Cvoid}, Ptr{NTuple{2,UInt64}}, Int32, Ptr{Float32}, Ptr{Float32}),
    engine!(nn).h, keys, N, P, V))
  return [(P[GI.actions_mask(GI.init(gspec, batch[i])), i], V[i]) for i in eachindex(batch)]
end

# ---- seam 1: the whole self-play phase ---------------------------------------------------------------------
policy_from_visits(N, mask) = (π = [N[a] for a in eachindex(mask) if mask[a]] ./ sum(N[a] for a in eachindex(mask) if mask[a]); π ./ sum(π))

struct WorkerView; mcts; end                 # what `measure(trace, colors_flipped, player)` reads
struct WorkerStats; nodes::Int64; sims::Int64; trav::Int64; nbytes::Int; end
AlphaZero.MCTS.approximate_memory_footprint(w::WorkerStats) = w.nbytes * w.nodes
AlphaZero.MCTS.average_exploration_depth(w::WorkerStats) = w.sims == 0 ? 0 : w.trav / w.sims

const progress_cb = Ref{Any}(nothing)
c_progress(::Ptr{Cvoid})::Cvoid = (isnothing(progress_cb[]) || progress_cb[](); nothing)

"simulate(::Simulator, gspec, ::SimParams; game_simulated) on the GPU when the player is MctsPlayer + HipResNet"
function AlphaZero.simulate(simulator::Simulator, gspec::DeviceGameSpec, p::SimParams; game_simulated,
                            first_game_id=0, seed=1)
  oracle = simulator.make_oracles()
  player = simulator.make_player(oracle)
  if !(oracle isa HipResNet && player isa MctsPlayer && isnothing(player.timeout))
    return invoke(AlphaZero.simulate, Tuple{Simulator, AlphaZero.AbstractGameSpec, SimParams}, simulator, gspec, p;
                  game_simulated=game_simulated)
  end
  m = player.mcts
  mp = MctsParams(gamma=m.gamma, cpuct=m.cpuct, num_iters_per_turn=player.niters, temperature=player.τ,
                  dirichlet_noise_ϵ=m.noise_ϵ, dirichlet_noise_α=m.noise_α, prior_temperature=m.prior_temperature)
  e = Engine(make_cfg(gspec, mp, p, oracle.hyper; seed=seed, device=parse(Int, get(ENV, "AZHIP_DEVICE", "0"))))
  check(ccall((:az_net_set_params, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64), e.h, oracle.blob, length(oracle.blob)))
  maxlen = gspec isa ConnectFour.GameSpec ? 42 : gspec isa Tictactoe.GameSpec ? 9 : 256
  games = Vector{GameRec}(undef, p.num_games); moves = Vector{MoveRec}(undef, p.num_games * maxlen)
  stats = SelfplayStats()
  progress_cb[] = game_simulated
  GC.@preserve games moves begin
    tb = TraceBuf(pointer(games), length(games), 0, pointer(moves), length(moves), 0)
    check(ccall((:az_selfplay_run, LIB), Cint,
      (Ptr{Cvoid}, Int32, Int32, Ref{TraceBuf}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{SelfplayStats}),
      e.h, p.num_games, first_game_id, tb, @cfunction(c_progress, Cvoid, (Ptr{Cvoid},)), C_NULL, stats))
    resize!(games, tb.num_games)
  end
  # the Python host's policy (azhip/training.py): warn, and refuse when games are missing or more than 5 % were aborted
  if stats.aborted_games > 0
    msg = "azhip: $(stats.aborted_games) of $(p.num_games) games were aborted (tree node pool / move record full), $(tb.num_games) came back: raise max_nodes_per_slot / max_moves_per_game"
    (tb.num_games < p.num_games || 20 * stats.aborted_games > max(p.num_games, 1)) ? error(msg) : @warn msg
  end
  nA = GI.num_actions(gspec)
  hb = nA <= 8 ? 2 : 4                                            # width of the child-link high bits (NodeL, csrc/tree.h)
  nbytes = cld(cld(cld(8nA, 8) * 8 + 8nA + 2nA, hb) * hb + hb, 32) * 32 + 32 + 12   # node record + side record (key, Vest) + hash-table share
  return map(games) do g
    recs = moves[g.first_move + 1 : g.first_move + g.num_moves]
    trace = Trace(decode_state(gspec, recs[1].key))
    for (k, r) in enumerate(recs)
      s = decode_state(gspec, r.key)
      mask = GI.actions_mask(GI.init(gspec, s))
      next = k < length(recs) ? decode_state(gspec, recs[k+1].key) : decode_state(gspec, g.final_key)
      push!(trace, policy_from_visits(r.N, mask), Float64(r.reward), next)
    end
    simulator.measure(trace, false, WorkerView(WorkerStats(g.nodes, g.total_simulations, g.total_nodes_traversed, nbytes)))
  end
end

# the replay memory on the device (az_memory_*): the handle only -- push / data set / learning status are in AlphaZeroHIPExtras.jl
mutable struct DeviceMemory
  h::Ptr{Cvoid}
  gspec
  function DeviceMemory(gspec, size; device=0)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:az_memory_create, LIB), Cint, (Int32, Int32, Int64, Ref{Ptr{Cvoid}}), game_id(gspec), device, size, out))
    m = new(out[], gspec)
    finalizer(x -> (x.h != C_NULL && ccall((:az_memory_destroy, LIB), Cint, (Ptr{Cvoid},), x.h); x.h = C_NULL), m)
    return m
  end
end

# ---- multi-GPU: simulate_distributed (src/simulations.jl:252-290) and the device-resident self-play step -------------
import Distributed

struct GatherStats
  games::Int64; moves::Int64; bytes::Int64; gather_ms::Float64; total_ms::Float64
  ranks::Int64; total_simulations::Int64; total_nodes_traversed::Int64; max_nodes::Int64; mean_game_depth::Float64
  replaced_games::Int64
end
const COMM_ID_BYTES = 128

"ncclGetUniqueId on the calling process (rank 0); ship the bytes to the other ranks with Distributed"
function comm_unique_id()
  id = Vector{UInt8}(undef, COMM_ID_BYTES)
  check(ccall((:az_comm_unique_id, LIB), Cint, (Ptr{UInt8},), id))
  return id
end

"One rank's RCCL communicator (az_comm_init is collective over all `world` ranks, one GPU each)"
mutable struct Comm
  h::Ptr{Cvoid}; rank::Int; world::Int; device::Int
  function Comm(device, rank, world, id::Vector{UInt8})
    @assert length(id) == COMM_ID_BYTES
    out = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:az_comm_init, LIB), Cint, (Int32, Int32, Int32, Ptr{UInt8}, Ref{Ptr{Cvoid}}), device, rank, world, id, out))
    c = new(out[], rank, world, device)
    finalizer(x -> (x.h != C_NULL && ccall((:az_comm_destroy, LIB), Cint, (Ptr{Cvoid},), x.h); x.h = C_NULL), c)
    return c
  end
end

"Collective: every rank's device-resident phase records -> (optionally) this rank's DeviceMemory, global game-id order"
function gather_push!(c::Comm, e::Engine, m::Union{Nothing, DeviceMemory}, gamma)
  st = Ref(GatherStats(0, 0, 0, 0.0, 0.0, 0, 0, 0, 0, 0.0, 0))
  check(ccall((:az_comm_gather_push, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Ref{GatherStats}),
    c.h, e.h, isnothing(m) ? C_NULL : m.h, gamma, st))
  return st[]
end
"Collective: the root's parameters to every rank's engine (the network shipped to the workers, training.jl:278-282)"
broadcast_params!(c::Comm, e::Engine, root=0) =
  check(ccall((:az_comm_broadcast_params, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int32), c.h, e.h, root))

"push_trace! for every game of the engine's last device-only phase, on the device (az_memory_push_engine)"
push_engine!(m::DeviceMemory, e::Engine, gamma) =
  check(ccall((:az_memory_push_engine, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Float64), m.h, e.h, gamma))
release_phase!(e::Engine) = check(ccall((:az_engine_release_phase, LIB), Cint, (Ptr{Cvoid},), e.h))
function device_bytes(e::Engine)
  n = Ref{Int64}(0)
  check(ccall((:az_engine_device_bytes, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}), e.h, n))
  return n[]
end

"az_selfplay_run with the move records kept in HBM: returns the game records (first_move = -1) and the phase statistics"
function selfplay_device_only!(e::Engine, num_games, first_game_id; game_simulated=nothing)
  games = Vector{GameRec}(undef, num_games)
  stats = SelfplayStats()
  progress_cb[] = game_simulated
  GC.@preserve games begin
    tb = TraceBuf(pointer(games), length(games), 0, Ptr{MoveRec}(C_NULL), 0, 0)
    check(ccall((:az_selfplay_run, LIB), Cint,
      (Ptr{Cvoid}, Int32, Int32, Ref{TraceBuf}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{SelfplayStats}),
      e.h, num_games, first_game_id, tb, @cfunction(c_progress, Cvoid, (Ptr{Cvoid},)), C_NULL, stats))
    resize!(games, tb.num_games)
  end
  return games, stats
end

"simulate_distributed's split (simulations.jl:268-278): divrem, the remainder to the first worker; global ids contiguous in rank order"
function shard_games(num_games, world, rank)
  num_each, rem = divrem(num_games, world)
  @assert num_each >= 1
  counts = [r == 0 ? num_each + rem : num_each for r in 0:world-1]
  return sum(counts[1:rank]), counts[rank + 1]
end

"""
    simulate_distributed(simulator, gspec::DeviceGameSpec, p; game_simulated)

More specific method of the reference's function (src/simulations.jl:252-290) for games with a device twin: one Julia worker
process per GPU (worker i drives device i-1), the games split by `divrem` exactly as the reference does, every worker runs
`simulate` on its GPU with GLOBAL game ids (so the traces do not depend on the number of workers), results fetched and
concatenated in worker order = game-id order: host traces like the reference's.  With `comm` (and `memory`) the call is ONE
RANK of a job with one process per GPU: device-only phase + `az_comm_gather_push`, no record leaves the GPUs;
`device_self_play_step!` below wraps that form into `self_play_step!`'s report.
"""
function AlphaZero.simulate_distributed(simulator::Simulator, gspec::DeviceGameSpec, p::SimParams; game_simulated, seed=1,
                                        comm::Union{Nothing, Comm}=nothing, memory::Union{Nothing, DeviceMemory}=nothing, gamma=1.0)
  if !isnothing(comm)
    # One rank of a multi-GPU job (one process per GPU, `comm` over RCCL): the shard of this rank is simulated DEVICE-ONLY
    # with global game ids, then ONE collective all-gathers the records of all ranks and pushes every game, in game-id order,
    # into this rank's device memory (`az_comm_gather_push`).  Returns the GatherStats (games, samples, depth, footprint).
    oracle = simulator.make_oracles()
    player = simulator.make_player(oracle)
    @assert oracle isa HipResNet && player isa MctsPlayer "the device-only form needs MctsPlayer + HipResNet"
    first, count = shard_games(p.num_games, comm.world, comm.rank)
    m = player.mcts
    mp = MctsParams(gamma=m.gamma, cpuct=m.cpuct, num_iters_per_turn=player.niters, temperature=player.τ,
                    dirichlet_noise_ϵ=m.noise_ϵ, dirichlet_noise_α=m.noise_α, prior_temperature=m.prior_temperature)
    e = Engine(make_cfg(gspec, mp, SimParams(p; num_games=count), oracle.hyper; seed=seed, device=comm.device))
    check(ccall((:az_net_set_params, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64), e.h, oracle.blob, length(oracle.blob)))
    selfplay_device_only!(e, count, first; game_simulated=game_simulated)
    gs = gather_push!(comm, e, memory, gamma)
    release_phase!(e)
    return gs
  end
  workers = Distributed.workers()
  length(workers) == 1 && return AlphaZero.simulate(simulator, gspec, p; game_simulated=game_simulated, seed=seed)
  chan = Distributed.RemoteChannel(() -> Channel{Nothing}(1))
  @async for i in 1:p.num_games
    take!(chan); game_simulated()
  end
  remote_game_simulated() = put!(chan, nothing)
  tasks = map(enumerate(workers)) do (i, w)
    first, count = shard_games(p.num_games, length(workers), i - 1)
    Distributed.@spawnat w begin
      ENV["AZHIP_DEVICE"] = string(i - 1)
      AlphaZero.simulate(simulator, gspec, SimParams(p; num_games=count);
                         game_simulated=remote_game_simulated, first_game_id=first, seed=seed)
    end
  end
  return reduce(vcat, fetch.(tasks))
end
# ---- everything beyond the hot path's three seams (arena, replay memory, learning status, optimiser step, device-resident
# self-play step, explorer view) and the UNBOUND list live in an optional second file
isfile(joinpath(@__DIR__, "AlphaZeroHIPExtras.jl")) && include(joinpath(@__DIR__, "AlphaZeroHIPExtras.jl"))

end # module
