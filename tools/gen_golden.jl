# gen_golden.jl -- golden vectors produced by the REFERENCE ITSELF (jonathan-laurent/AlphaZero.jl), for pinning the oracle
# and the HIP engine (VERDICT r1 item 1b).
#
#   julia --project=/path/to/AlphaZero.jl tools/gen_golden.jl [outdir = tests/golden]
#
# WRITTEN BLIND: the build container has no Julia, so this script has never been run (its loader,
# tests/test_reference_golden.py, is exercised against the same schema written by oracle/pyref.py).  It calls the
# reference's own code for everything that is being pinned and injects only DATA:
#   * MCTS.explore! / run_simulation! / uct_scores / policy          src/mcts.jl:157-271       (called, not restated)
#   * play_game, player_temperature, apply_temperature               src/play.jl:298-315, src/util.jl:98-110
#   * fix_probvec + Distributions.jl's Categorical sampler           src/util.jl:68-90         (the real `rand(rng, Categorical(π))`)
#   * Network.evaluate_batch / forward_normalized on a Flux ResNet   src/networks/network.jl:264-315, resnet.jl:65-92
# Injected: the Dirichlet noise η of every explore! (MCTS.dirichlet_noise is replaced by a queue: η is data; Julia's Gamma
# sampler and RNG stream are not reproducible elsewhere), the uniform u of every categorical draw (a shim RNG hands the
# queued Float32 to the real sampler), and the oracles, which are exact functions of the state: MCTS.RandomOracle and the
# "hash oracle" of include/azhip.h (AZ_ORACLE_HASH: integer hashing + Float32 divisions, bit-reproducible in Julia).
#
# Output (schema shared with tests/test_reference_golden.py):
#   ref_mcts.json   explore! from a given root: η (by rank among available actions), N, W, P, Vest, π, counters
#   ref_play.json   whole play_game traces with per-move η and u: states (packed keys), π targets, actions, rewards; two cases
#                   with flip_probability > 0 ("flips": which symmetry each turn took)
#   ref_net.json    ResNet outputs (P over available actions, V) for a list of states + ref_net_blob.f32 (parameters in
#                   the blob order of az_net_set_params, flattened by julia/AlphaZeroHIP.jl's HipResNet(::ResNet))
using AlphaZero
using AlphaZero: GI, MCTS, Util, Network, MctsParams, MctsPlayer, PLSchedule, ConstSchedule
using Distributions: Categorical, Dirichlet
using Random
import JSON3
import Flux                      # a dependency of AlphaZero.jl (src/networks/flux.jl)

include(joinpath(@__DIR__, "..", "julia", "AlphaZeroHIP.jl"))
using .AlphaZeroHIP: encode_state, HipResNet

const AZVER = try string(pkgversion(AlphaZero)) catch; "(version unknown)" end
outdir = length(ARGS) >= 1 ? ARGS[1] : joinpath(@__DIR__, "..", "tests", "golden")
mkpath(outdir)

# ---- injection points ---------------------------------------------------------------------------------------------------
const ETA_QUEUE = Vector{Vector{Float64}}()
@eval MCTS dirichlet_noise(game, α) = popfirst!(Main.ETA_QUEUE)      # η by rank among available actions (mcts.jl:228-232)

"RNG whose only supported draws are scalar uniforms taken from a queue; the sampler that consumes them is the reference's"
struct ShimRNG <: Random.AbstractRNG
  queue::Vector{Float32}
end
Random.rand(r::ShimRNG, ::Random.SamplerTrivial{Random.CloseOpen01{Float32}}) = popfirst!(r.queue)
Random.rand(r::ShimRNG, ::Random.SamplerTrivial{Random.CloseOpen01{Float64}}) = Float64(popfirst!(r.queue))
const SHIM = ShimRNG(Float32[])
@eval Util function rand_categorical(π)                                 # util.jl:87-90 with the RNG made explicit
  π = fix_probvec(π)
  return rand(Main.SHIM, Categorical(π))
end

# ---- the RNG contract's move uniform (include/az_numerics.h), so that u is what the engine itself would draw -------------
function philox4x32_10(ctr::NTuple{4,UInt32}, key::NTuple{2,UInt32})
  c0, c1, c2, c3 = ctr; k0, k1 = key
  for _ in 1:10
    p0 = UInt64(0xD2511F53) * c0; p1 = UInt64(0xCD9E8D57) * c2
    c0, c1, c2, c3 = (UInt32(p1 >> 32) ⊻ c1 ⊻ k0, UInt32(p1 & 0xffffffff), UInt32(p0 >> 32) ⊻ c3 ⊻ k1, UInt32(p0 & 0xffffffff))
    k0 += 0x9E3779B9; k1 += 0xBB67AE85
  end
  return (c0, c1, c2, c3)
end
move_uniform(seed::UInt64, game::Integer, move::Integer) =
  Float32(philox4x32_10((UInt32(game), UInt32(move), UInt32(2), UInt32(0)), (UInt32(seed & 0xffffffff), UInt32(seed >> 32)))[3] >> 8) * Float32(2.0^-24)

# ---- oracles -----------------------------------------------------------------------------------------------------------------
function mix64(x::UInt64)
  x ⊻= x >> 30; x *= 0xbf58476d1ce4e5b9
  x ⊻= x >> 27; x *= 0x94d049bb133111eb
  x ⊻= x >> 31
  return x
end
hash_key(a::UInt64, b::UInt64) = mix64(a ⊻ mix64(b + 0x9e3779b97f4a7c15))
"AZ_ORACLE_HASH (csrc/tree.h k_synth_oracle, oracle/azref.c): priors and value derived from the packed state key"
struct HashOracle{G}; gspec::G; end
function (o::HashOracle)(state)
  g = GI.init(o.gspec, state)
  mask = GI.actions_mask(g)
  a, b = encode_state(o.gspec, state)
  h = hash_key(a, b)
  raw = Float32[Float32(1 + Int(mix64(h + UInt64(i)) & 0xffff)) for i in eachindex(mask) if mask[i]]   # action index i is 1-based here = a + 1
  s = 0f0
  for r in raw; s += r; end
  P = raw ./ s
  V = Float32(Int(mix64(h + UInt64(99)) & 0xffff) - 32768) / 65536f0
  return P, V
end

keystr(k) = [string(k[1]), string(k[2])]                               # u64 as decimal strings (JSON numbers are doubles)
game_ids = Dict("connect-four" => 0, "tictactoe" => 1, "mancala" => 2)

function make_oracle(name, gspec)
  name == "uniform" && return MCTS.RandomOracle(gspec)
  name == "hash" && return HashOracle(gspec)
  error("unknown oracle $name")
end

# ---- ref_mcts.json ----------------------------------------------------------------------------------------------------------------
mcts_cases = []
rng = MersenneTwister(2026)
for (gname, oname, nsims, cpuct, gamma, eps, ptemp, nprefix) in [
    ("connect-four", "hash", 400, 2.0, 1.0, 0.25, 1.0, 0), ("connect-four", "hash", 600, 2.0, 1.0, 0.25, 1.0, 5),
    ("connect-four", "uniform", 400, 2.0, 1.0, 0.0, 1.0, 0), ("connect-four", "hash", 200, 1.3, 0.95, 0.25, 0.5, 3),
    ("tictactoe", "hash", 64, 1.0, 1.0, 0.25, 1.0, 0), ("tictactoe", "hash", 200, 1.0, 0.9, 0.5, 2.0, 2),
    ("mancala", "hash", 800, 2.0, 1.0, 0.25, 1.0, 0), ("mancala", "hash", 300, 2.0, 0.97, 0.25, 1.0, 6),
    ("mancala", "uniform", 200, 1.0, 1.0, 0.25, 1.0, 4)]
  gspec = AlphaZero.Examples.games[gname]
  game = GI.init(gspec)
  prefix = Int[]
  for _ in 1:nprefix
    GI.game_terminated(game) && break
    a = rand(rng, GI.available_actions(game))
    g2 = GI.clone(game); GI.play!(g2, a)
    GI.game_terminated(g2) && break
    GI.play!(game, a); push!(prefix, findfirst(==(a), GI.actions(gspec)) - 1)
  end
  n = length(GI.available_actions(game))
  η = rand(rng, Dirichlet(n, 1.0))
  env = MCTS.Env(gspec, make_oracle(oname, gspec); gamma=gamma, cpuct=cpuct, noise_ϵ=eps, noise_α=1.0, prior_temperature=ptemp)
  push!(ETA_QUEUE, copy(η))
  MCTS.explore!(env, game, nsims)
  actions, π = MCTS.policy(env, game)
  info = env.tree[GI.current_state(game)]
  push!(mcts_cases, Dict(
    "game" => game_ids[gname], "oracle" => oname, "nsims" => nsims, "cpuct" => cpuct, "gamma" => gamma, "eps" => eps,
    "prior_temperature" => ptemp, "prefix" => prefix, "root_key" => keystr(encode_state(gspec, GI.current_state(game))),
    "eta" => η, "actions" => [findfirst(==(a), GI.actions(gspec)) - 1 for a in actions],
    "N" => [Int(s.N) for s in info.stats], "W" => [Float64(s.W) for s in info.stats], "P" => [Float64(s.P) for s in info.stats],
    "Vest" => Float64(info.Vest), "pi" => π, "total_simulations" => env.total_simulations,
    "total_nodes_traversed" => env.total_nodes_traversed, "num_nodes" => length(env.tree)))
end
open(joinpath(outdir, "ref_mcts.json"), "w") do io
  JSON3.write(io, Dict("generator" => "AlphaZero.jl $(AZVER) via tools/gen_golden.jl", "cases" => mcts_cases))
end

# ---- ref_play.json ----------------------------------------------------------------------------------------------------------------
play_cases = []
for (gname, oname, nsims, cpuct, xs, ys, seed, gid, flip_p) in [
    ("connect-four", "hash", 100, 2.0, [0, 20, 30], [1.0, 1.0, 0.3], UInt64(1), 0, 0.0),
    ("connect-four", "hash", 60, 2.0, [0, 4, 8], [1.0, 0.5, 0.0], UInt64(7), 12345, 0.0),
    ("tictactoe", "hash", 64, 1.0, [0], [1.0], UInt64(1), 3, 0.0),
    ("mancala", "hash", 80, 2.0, [0, 20, 30], [1.0, 1.0, 0.3], UInt64(5), 42, 0.0),
    # play_game's flip_probability (play.jl:305-307): the reference's own apply_random_symmetry! picks the image, which one it
    # was is recorded as data ("flips": 0 = the turn was not flipped, else 1 + index in GI.symmetries)
    ("connect-four", "hash", 60, 2.0, [0, 10], [1.0, 0.5], UInt64(3), 77, 0.5),
    ("tictactoe", "hash", 64, 1.0, [0], [1.0], UInt64(2), 5, 0.6)]
  gspec = AlphaZero.Examples.games[gname]
  τ = length(xs) == 1 ? ConstSchedule(ys[1]) : PLSchedule(xs, ys)
  params = MctsParams(num_iters_per_turn=nsims, cpuct=cpuct, temperature=τ, dirichlet_noise_ϵ=0.25, dirichlet_noise_α=1.0)
  player = MctsPlayer(gspec, make_oracle(oname, gspec), params)
  # η and u for at most 300 moves are queued up front; what the game consumed is what gets recorded
  etas = Vector{Vector{Float64}}(); us = Float32[]
  empty!(ETA_QUEUE); empty!(SHIM.queue)
  # a move's η depends on the number of available actions, which is only known when the move is reached: play_game is
  # therefore driven move by move through `think` (the body of play.jl:298-315 is reproduced by calling its own pieces)
  game = GI.init(gspec)
  trace = AlphaZero.Trace(GI.current_state(game))
  Ns = Vector{Vector{Int}}(); acts = Int[]; flips = Int[]
  while !GI.game_terminated(game)
    flip = 0
    if !iszero(flip_p) && rand(rng) < flip_p                         # play.jl:305-307
      before = GI.current_state(game)
      GI.apply_random_symmetry!(game)
      flip = findfirst(sym -> sym[1] == GI.current_state(game), GI.symmetries(gspec, before))
    end
    push!(flips, flip)
    n = length(GI.available_actions(game))
    η = rand(rng, Dirichlet(n, 1.0)); push!(etas, η); push!(ETA_QUEUE, copy(η))
    u = move_uniform(seed, gid, length(trace)); push!(us, u); push!(SHIM.queue, u)
    actions, π_target = AlphaZero.think(player, game)
    τm = AlphaZero.player_temperature(player, game, length(trace))
    π_sample = Util.apply_temperature(π_target, τm)
    a = actions[Util.rand_categorical(π_sample)]
    push!(Ns, [Int(s.N) for s in player.mcts.tree[GI.current_state(game)].stats])
    push!(acts, findfirst(==(a), GI.actions(gspec)) - 1)
    GI.play!(game, a)
    push!(trace, π_target, GI.white_reward(game), GI.current_state(game))
  end
  push!(play_cases, Dict(
    "game" => game_ids[gname], "oracle" => oname, "nsims" => nsims, "cpuct" => cpuct, "temp_xs" => xs, "temp_ys" => ys,
    "seed" => string(seed), "game_id" => gid, "etas" => etas, "us" => [Float64(u) for u in us],
    "states" => [keystr(encode_state(gspec, s)) for s in trace.states], "policies" => trace.policies,
    "rewards" => trace.rewards, "actions" => acts, "N" => Ns, "flips" => flips,
    "total_simulations" => player.mcts.total_simulations, "total_nodes_traversed" => player.mcts.total_nodes_traversed,
    "num_nodes" => length(player.mcts.tree)))
end
open(joinpath(outdir, "ref_play.json"), "w") do io
  JSON3.write(io, Dict("generator" => "AlphaZero.jl $(AZVER) via tools/gen_golden.jl", "cases" => play_cases))
end

# ---- ref_net.json -----------------------------------------------------------------------------------------------------------------
net_cases = []
for (gname, nblocks, nf, npf, nvf) in [("connect-four", 2, 64, 32, 32), ("tictactoe", 1, 64, 32, 32), ("mancala", 1, 64, 32, 32)]
  gspec = AlphaZero.Examples.games[gname]
  hp = AlphaZero.ResNetHP(num_blocks=nblocks, num_filters=nf, conv_kernel_size=(3, 3), num_policy_head_filters=npf, num_value_head_filters=nvf)
  nn = AlphaZero.ResNet(gspec, hp)
  # exercise BatchNorm: non-trivial running statistics and affine parameters (Flux initialises them to 0 / 1)
  for chain in (nn.common, nn.phead, nn.vhead), bn in Flux.modules(chain)
    bn isa Flux.BatchNorm || continue
    bn.γ .= 1f0 .+ 0.1f0 .* (2f0 .* rand(rng, Float32, size(bn.γ)) .- 1f0); bn.β .= 0.1f0 .* (2f0 .* rand(rng, Float32, size(bn.β)) .- 1f0)
    bn.μ .= 0.1f0 .* (2f0 .* rand(rng, Float32, size(bn.μ)) .- 1f0); bn.σ² .= 1f0 .+ 0.1f0 .* (2f0 .* rand(rng, Float32, size(bn.σ²)) .- 1f0)
  end
  nn = Network.copy(nn, on_gpu=false, test_mode=true)
  states = []
  for _ in 1:24
    game = GI.init(gspec)
    for _ in 1:rand(rng, 0:20)
      GI.game_terminated(game) && break
      g2 = GI.clone(game); GI.play!(g2, rand(rng, GI.available_actions(g2)))
      GI.game_terminated(g2) && break
      game = g2
    end
    push!(states, GI.current_state(game))
  end
  out = Network.evaluate_batch(nn, states)
  blobfile = "ref_net_blob_$(game_ids[gname]).f32"
  write(joinpath(outdir, blobfile), HipResNet(nn).blob)
  push!(net_cases, Dict(
    "game" => game_ids[gname], "num_blocks" => nblocks, "num_filters" => nf, "num_policy_head_filters" => npf,
    "num_value_head_filters" => nvf, "blob_file" => blobfile, "states" => [keystr(encode_state(gspec, s)) for s in states],
    "P" => [Float64.(p) for (p, v) in out], "V" => [Float64(v) for (p, v) in out]))
end
open(joinpath(outdir, "ref_net.json"), "w") do io
  JSON3.write(io, Dict("generator" => "AlphaZero.jl $(AZVER) + Flux via tools/gen_golden.jl", "cases" => net_cases))
end
println("wrote ref_mcts.json ($(length(mcts_cases)) cases), ref_play.json ($(length(play_cases))), ref_net.json ($(length(net_cases))) to $outdir")
