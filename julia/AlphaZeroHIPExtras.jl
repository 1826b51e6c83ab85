# AlphaZeroHIPExtras.jl -- included by AlphaZeroHIP.jl (inside the module) when present: the SURVEY 8(f) rows and helpers around the
# three seams of the hot path (pit_networks, device replay memory / learning status / optimiser step, the device-resident
# self-play step, the explorer's view of a device tree) and the list of header entry points this glue deliberately leaves unbound.
# Written blind like the core file (no Julia in the build image); tests/test_julia_glue_static.py parses both files.

# ---- arena: pit_networks (src/training.jl:130-144) --------------------------------------------------------
"""
    pit_networks(gspec::DeviceGameSpec, contender::HipResNet, baseline::HipResNet, params::ArenaParams, handler)

More specific method of the reference's own function: one engine per network, az_arena_run plays
`params.sim.num_games` games of TwoPlayers(MctsPlayer(contender), MctsPlayer(baseline)) honouring
`flip_probability` and `alternate_colors`, and returns `(rewards, redundancy)` like rewards_and_redundancy.
"""
function AlphaZero.pit_networks(gspec::DeviceGameSpec, contender::HipResNet, baseline::HipResNet, params, handler;
                                seed=1)
  engines = map((contender, baseline)) do nn
    e = Engine(make_cfg(gspec, params.mcts, params.sim, nn.hyper; seed=seed, arena=true))
    check(ccall((:az_net_set_params, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64), e.h, nn.blob, length(nn.blob)))
    e
  end
  n = params.sim.num_games
  rewards = Vector{Float64}(undef, n)
  redundancy = Ref{Float64}(0.0)
  progress_cb[] = () -> AlphaZero.Handlers.checkpoint_game_played(handler)
  check(ccall((:az_arena_run, LIB), Cint,
    (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Cvoid}, Ptr{Float64}, Ref{Float64}, Ptr{Cvoid}, Ptr{Cvoid}),
    engines[1].h, engines[2].h, n, 0, params.sim.alternate_colors ? 1 : 0, C_NULL, rewards, redundancy,
    @cfunction(c_progress, Cvoid, (Ptr{Cvoid},)), C_NULL))
  return rewards, redundancy[]
end

# ---- replay memory + learning status on the device (src/memory.jl, src/learning.jl:59-90,148-190) ------------
"az_sample: TrainingSample with π by full action index (112 bytes)"
struct AzSample
  key::NTuple{2, UInt64}
  pi::NTuple{9, Float64}
  z::Float64
  t::Float64
  n::Int64
end
struct DatasetInfo; num_samples::Int64; sum_n::Int64; Wtot::Float64; Wmean::Float32; Hp::Float32; end
struct LearningStatusRec; L::Float32; Lp::Float32; Lv::Float32; Lreg::Float32; Linv::Float32; Hp::Float32; Hpnet::Float32; end

"MemoryBuffer whose samples live in HBM; `push_records!` takes the packed records az_selfplay_run returned"
function push_records!(m::DeviceMemory, games::Vector{GameRec}, moves::Vector{MoveRec}, gamma)
  GC.@preserve games moves begin
    tb = TraceBuf(pointer(games), length(games), length(games), pointer(moves), length(moves), length(moves))
    check(ccall((:az_memory_push, LIB), Cint, (Ptr{Cvoid}, Ref{TraceBuf}, Float64), m.h, tb, gamma))
  end
end

"""
    device_learning_status(nn::HipResNet, m::DeviceMemory, lp::LearningParams; use_symmetries, last_batch=false)

`learning_status(Trainer(gspec, nn, experience, lp))` (src/learning.jl:98-121,158-181) without any sample leaving the
GPU: augment_with_symmetries, merge_by_state, convert_samples and the losses run on the device data set.
"""
function device_learning_status(nn::HipResNet, m::DeviceMemory, lp; use_symmetries::Bool, last_batch::Bool=false)
  ds = Ref{Ptr{Cvoid}}(C_NULL)
  check(ccall((:az_dataset_create, LIB), Cint, (Ptr{Cvoid}, Int32, Int32, Int32, Int32, Ref{Ptr{Cvoid}}),
    m.h, last_batch ? 1 : 0, use_symmetries ? 1 : 0, lp.use_position_averaging ? 1 : 0, Int32(lp.samples_weighing_policy), ds))
  try
    out = Ref(LearningStatusRec(0, 0, 0, 0, 0, 0, 0))
    check(ccall((:az_learning_status, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Float64, Int64, Ref{LearningStatusRec}),
      engine!(nn).h, ds[], lp.l2_regularization, lp.nonvalidity_penalty, lp.rewards_renormalization, lp.loss_computation_batch_size, out))
    r = out[]
    return AlphaZero.Report.LearningStatus(AlphaZero.Report.Loss(r.L, r.Lp, r.Lv, r.Lreg, r.Linv), r.Hp, r.Hpnet)
  finally
    ccall((:az_dataset_destroy, LIB), Cint, (Ptr{Cvoid},), ds[])
  end
end

# ---- the optimiser step: batch_updates! (src/learning.jl:131-141) on the device --------------------------------
struct TrainCfg
  struct_size::Int32; optimiser::Int32; lr::Float32
  lr_base::Float32; lr_high::Float32; lr_low::Float32; momentum_low::Float32; momentum_high::Float32
  l2_regularization::Float64; nonvalidity_penalty::Float64; rewards_renormalization::Float64
  batch_size::Int32; batch_norm_momentum::Float32; seed::UInt64
end
train_cfg(lp, hp; seed=1) = begin
  o = lp.optimiser
  adam = o isa AlphaZero.Adam
  TrainCfg(Int32(sizeof(TrainCfg)), adam ? 0 : 1, adam ? o.lr : 0f0,
    adam ? 0f0 : o.lr_base, adam ? 0f0 : o.lr_high, adam ? 0f0 : o.lr_low, adam ? 0f0 : o.momentum_low, adam ? 0f0 : o.momentum_high,
    lp.l2_regularization, lp.nonvalidity_penalty, lp.rewards_renormalization, lp.batch_size, hp.batch_norm_momentum, UInt64(seed))
end

"""
    device_batch_updates!(nn::HipResNet, m::DeviceMemory, lp::LearningParams, n; use_symmetries) -> losses

`batch_updates!(Trainer(gspec, nn, experience, lp), n)` on the device; `nn.blob` receives the trained parameters
(get_trained_network).  A long-lived trainer handle would be kept across checkpoints in a real integration.
"""
function device_batch_updates!(nn::HipResNet, m::DeviceMemory, lp, n; use_symmetries::Bool, seed=1)
  ds = Ref{Ptr{Cvoid}}(C_NULL); tr = Ref{Ptr{Cvoid}}(C_NULL)
  check(ccall((:az_dataset_create, LIB), Cint, (Ptr{Cvoid}, Int32, Int32, Int32, Int32, Ref{Ptr{Cvoid}}),
    m.h, 0, use_symmetries ? 1 : 0, lp.use_position_averaging ? 1 : 0, Int32(lp.samples_weighing_policy), ds))
  losses = Vector{Float32}(undef, n)
  try
    check(ccall((:az_trainer_create, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{TrainCfg}, Ref{Ptr{Cvoid}}),
      engine!(nn).h, ds[], train_cfg(lp, nn.hyper; seed=seed), tr))
    check(ccall((:az_trainer_batch_updates, LIB), Cint, (Ptr{Cvoid}, Int32, Ptr{Float32}), tr[], n, losses))
    check(ccall((:az_trainer_get_params, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64), tr[], nn.blob, length(nn.blob)))
    check(ccall((:az_net_set_params, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64), engine!(nn).h, nn.blob, length(nn.blob)))
  finally
    tr[] != C_NULL && ccall((:az_trainer_destroy, LIB), Cint, (Ptr{Cvoid},), tr[])
    ccall((:az_dataset_destroy, LIB), Cint, (Ptr{Cvoid},), ds[])
  end
  return losses
end


"""
    device_self_play_step!(gspec, bestnn::HipResNet, params::SelfPlayParams, mem::DeviceMemory, comm; seed) -> Report.SelfPlay

`self_play_step!` (src/training.jl:275-300) on one rank of a multi-GPU job, nothing leaving HBM: the rank's shard of the games
is simulated device-only (global game ids), `az_comm_gather_push` all-gathers the records of all ranks over RCCL and runs
`push_trace!` for ALL games in game-id order into this rank's device memory -- every rank ends with the samples a single-GPU
run would have pushed.  `comm === nothing`: single GPU (az_memory_push_engine).  Call it on every rank (e.g. with
`Distributed.@spawnat` on one worker per GPU after `Comm(device, rank, world, id)` there).
"""
function device_self_play_step!(gspec::DeviceGameSpec, bestnn::HipResNet, params, mem::DeviceMemory, comm::Union{Nothing, Comm};
                                seed=1, game_simulated=nothing)
  world, rank, device = isnothing(comm) ? (1, 0, 0) : (comm.world, comm.rank, comm.device)
  first, count = shard_games(params.sim.num_games, world, rank)
  e = Engine(make_cfg(gspec, params.mcts, SimParams(params.sim; num_games=count), bestnn.hyper; seed=seed, device=device))
  check(ccall((:az_net_set_params, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64), e.h, bestnn.blob, length(bestnn.blob)))
  t0 = time()
  games, stats = selfplay_device_only!(e, count, first; game_simulated=game_simulated)
  elapsed = time() - t0
  check(ccall((:az_memory_new_batch, LIB), Cint, (Ptr{Cvoid},), mem.h))
  nA = GI.num_actions(gspec)
  hb = nA <= 8 ? 2 : 4
  node_bytes = cld(cld(cld(8nA, 8) * 8 + 8nA + 2nA, hb) * hb + hb, 32) * 32 + 32 + 12
  if isnothing(comm)
    push_engine!(mem, e, params.mcts.gamma)
    nsamples = stats.moves
    edepth = isempty(games) ? 0.0 : sum(g.total_nodes_traversed / max(g.total_simulations, 1) for g in games) / length(games)
    footprint = node_bytes * maximum(g.nodes for g in games; init=0)
  else
    gs = gather_push!(comm, e, mem, params.mcts.gamma)
    elapsed += gs.gather_ms / 1000            # simulate_distributed's `fetch` is inside the reference's @timed region
    nsamples, edepth, footprint = gs.moves, gs.mean_game_depth, node_bytes * gs.max_nodes
  end
  release_phase!(e)
  len = Ref{Int64}(0); cur = Ref{Int64}(0)
  check(ccall((:az_memory_length, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}), mem.h, len, cur))
  ds = Ref{Ptr{Cvoid}}(C_NULL)                 # memory_num_distinct_boards = length(merge_by_state(get_experience(mem)))
  check(ccall((:az_dataset_create, LIB), Cint, (Ptr{Cvoid}, Int32, Int32, Int32, Int32, Ref{Ptr{Cvoid}}), mem.h, 0, 0, 1, 0, ds))
  info = Ref(DatasetInfo(0, 0, 0.0, 0f0, 0f0))
  check(ccall((:az_dataset_get_info, LIB), Cint, (Ptr{Cvoid}, Ref{DatasetInfo}), ds[], info))
  ccall((:az_dataset_destroy, LIB), Cint, (Ptr{Cvoid},), ds[])
  return AlphaZero.Report.SelfPlay(nsamples / elapsed, edepth, footprint, len[], info[].num_samples)
end

# ---- seam 5: the explorer's view of a device tree (src/ui/explorer.jl:70-86; MCTS.explore! / reset!, mcts.jl:239-281) --------
"""
    DeviceMctsEnv(gspec, nn::HipResNet, params::MctsParams)

`MCTS.Env` whose tree lives in slot 0 of a one-worker engine: `explore!`, `policy`-style statistics of a state, counters and
`reset!` -- what `Explorer.state_statistics` reads of `player.mcts` (N, W, P per action, Ntot, Vest).
"""
mutable struct DeviceMctsEnv
  gspec
  engine::Engine
  params::MctsParams
end
function DeviceMctsEnv(gspec::DeviceGameSpec, nn::HipResNet, params::MctsParams; seed=1)
  e = Engine(make_cfg(gspec, params, SimParams(num_games=1, num_workers=1, batch_size=1), nn.hyper; seed=seed))
  check(ccall((:az_net_set_params, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64), e.h, nn.blob, length(nn.blob)))
  return DeviceMctsEnv(gspec, e, params)
end
"MCTS.explore!(env, game, nsims) (mcts.jl:239-245); the Dirichlet noise is drawn from the library's RNG contract for (game_id, move)"
function explore!(env::DeviceMctsEnv, game, nsims=env.params.num_iters_per_turn; game_id=0, move=0)
  key = [encode_state(env.gspec, GI.current_state(game))]
  check(ccall((:az_mcts_explore, LIB), Cint,
    (Ptr{Cvoid}, Ptr{NTuple{2,UInt64}}, Int32, Int32, Ptr{Float64}, Ptr{UInt32}, Ptr{UInt32}),
    env.engine.h, key, 1, nsims, C_NULL, UInt32[game_id], UInt32[move]))
end
"tree[state]: (N, W, P) over the available actions in GI.actions order, Vest -- or nothing when the state is not in the tree"
function state_info(env::DeviceMctsEnv, state)
  key = [encode_state(env.gspec, state)]
  N = Vector{Int32}(undef, MAX_ACTIONS); W = Vector{Float64}(undef, MAX_ACTIONS); P = Vector{Float32}(undef, MAX_ACTIONS)
  vest = Ref{Float32}(0); mask = Ref{UInt32}(0)
  st = ccall((:az_mcts_node_stats, LIB), Cint,
    (Ptr{Cvoid}, Int32, Ptr{NTuple{2,UInt64}}, Ptr{Int32}, Ptr{Float64}, Ptr{Float32}, Ref{Float32}, Ref{UInt32}),
    env.engine.h, 0, key, N, W, P, vest, mask)
  st == 0 || return nothing
  avail = [a for a in 1:GI.num_actions(env.gspec) if (mask[] >> (a - 1)) & 1 == 1]
  return (N=N[avail], W=W[avail], P=P[avail], Vest=vest[])
end
"(total_simulations, total_nodes_traversed, length(tree)): MCTS.average_exploration_depth / memory footprint inputs (mcts.jl:286-321)"
function counters(env::DeviceMctsEnv)
  a = Ref{Int64}(0); b = Ref{Int64}(0); c = Ref{Int64}(0)
  check(ccall((:az_mcts_counters, LIB), Cint, (Ptr{Cvoid}, Int32, Ref{Int64}, Ref{Int64}, Ref{Int64}), env.engine.h, 0, a, b, c))
  return (total_simulations=a[], total_nodes_traversed=b[], num_nodes=c[])
end
"MCTS.reset!(env) (mcts.jl:278-281)"
reset!(env::DeviceMctsEnv) = check(ccall((:az_mcts_reset, LIB), Cint, (Ptr{Cvoid},), env.engine.h))

# ---- what this file deliberately does not bind, and why (tests/test_julia_glue_static.py checks the list against the header:
# an entry point of include/azhip.h that is neither `ccall`ed above nor named here fails the test) ---------------------------
const UNBOUND = Dict(
  :az_comm_version => "evidence for bench.py's gather object (which collective library was bound); nothing in the training loop needs it",
  :az_engine_cfg_init => "EngineCfg is filled field by field from MctsParams / SimParams (make_cfg)",
  :az_game_num_actions => "GI.num_actions(gspec) answers it on the Julia side",
  :az_game_state_dim => "GI.state_dim(gspec) answers it on the Julia side",
  :az_game_init_key => "encode_state(gspec, GI.current_state(GI.init(gspec))) answers it",
  :az_game_encode => "GI.vectorize_state is the reference's own; the device encode is fused into az_net_evaluate_keys",
  :az_game_play => "GI.play! is the reference's own; the twins are checked against it by tests/, not used from Julia",
  :az_net_num_params => "length(nn.blob) is known from the Flux network",
  :az_net_get_params => "nn.blob is the master copy; az_trainer_get_params returns trained parameters",
  :az_selfplay_begin => "stepping form (bench / polling): az_selfplay_run is what simulate needs",
  :az_selfplay_step => "stepping form, see az_selfplay_begin",
  :az_selfplay_collect => "stepping form, see az_selfplay_begin",
  :az_selfplay_get_stats => "az_selfplay_run returns the statistics",
  :az_selfplay_active => "stepping form, see az_selfplay_begin",
  :az_selfplay_end => "stepping form, see az_selfplay_begin",
  :az_selfplay_aborted => "ids of aborted games: simulate checks SelfplayStats.aborted_games, warns, and raises when games are missing or more than 5 % were aborted (the Python host's policy)",
  :az_push_trace => "host-side helper for foreign hosts; Julia has the reference's push_trace!",
  :az_memory_push_samples => "host TrainingSamples -> device memory: only needed when mixing host and device memories",
  :az_memory_empty => "empty!(mem): not used by the training loop (src/training.jl)",
  :az_dataset_read => "debug / test read-back of the converted samples",
  :az_train_cfg_init => "TrainCfg is filled from LearningParams (train_cfg)",
  :az_trainer_gradients => "test hook (gradients of one batch against autograd)",
  :az_prof_enable => "bench.py's HIP-event profiling", :az_prof_get => "bench.py's HIP-event profiling",
  :az_prof_reset => "bench.py's HIP-event profiling", :az_device_info => "bench.py's report", :az_net_last_kernel => "bench.py's report",
)

