"""Time series of a bounded self-play phase (the iteration's: 5000 games, 4096 workers in two groups, ResNet 5x128, 600 sims/move):
every `--waves` waves the wall clock, the finished games, the slots still playing, simulations and network evaluations -- where a phase
with fewer games than two per worker loses its rate."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alphazero.jl_amd"))
import azhip
from azhip.network import ResNetHP, random_params

ap = argparse.ArgumentParser()
ap.add_argument("--games", type=int, default=5000); ap.add_argument("--workers", type=int, default=4096); ap.add_argument("--groups", type=int, default=2)
ap.add_argument("--filters", type=int, default=128); ap.add_argument("--sims", type=int, default=600); ap.add_argument("--waves", type=int, default=512)
ap.add_argument("--reset-every", type=int, default=2)
ap.add_argument("--prof", action="store_true", help="time every kernel class with HIP events (az_prof): avg us per launch and launches per window")
ap.add_argument("--max-seconds", type=float, default=1e9)
a = ap.parse_args()
hp = ResNetHP(num_blocks=5, num_filters=a.filters, num_policy_head_filters=32, num_value_head_filters=32)
blob = random_params(azhip.GAME_CONNECT_FOUR, hp, seed=1)
with azhip.Engine(game=azhip.GAME_CONNECT_FOUR, oracle=azhip.ORACLE_RESNET, num_workers=a.workers, batch_size=a.workers // a.groups, num_iters_per_turn=a.sims,
                  gamma=1.0, cpuct=2.0, dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0, prior_temperature=1.0, temperature=((0, 20, 30), (1.0, 1.0, 0.3)),
                  reset_every=a.reset_every, seed=1, num_blocks=5, num_filters=a.filters, num_policy_head_filters=32, num_value_head_filters=32) as e:
    e.net_set_params(blob)
    e.selfplay_begin(a.games, 0)
    if a.prof:
        e.prof_enable(True)
    t0 = time.perf_counter(); last = (0.0, 0, 0)
    rows = []
    while True:
        e.selfplay_step(a.waves)
        st = e.selfplay_stats(); act = e.selfplay_active(); t = time.perf_counter() - t0
        rows.append(dict(t=round(t, 3), active=act, games=int(st.games), sims=int(st.simulations), evals=int(st.leaf_evals), reused=int(st.evals_reused),
                         msims_per_s=round((st.simulations - last[1]) / max(1e-9, t - last[0]) / 1e6, 3),
                         boards_per_wave=round((st.leaf_evals - st.evals_reused - last[2]) / a.waves, 1), ms_per_wave=round((t - last[0]) / a.waves * 1e3, 4)))
        rows[-1]["tower_kernel"] = e.net_last_kernel()
        if a.prof:
            pr = e.prof_get(); e.prof_reset()
            rows[-1]["kernels_us_x_launches"] = {k: [round(1e3 * v["ms"] / max(1, v["launches"]), 1), int(v["launches"])] for k, v in pr.items() if v["launches"]}
        last = (t, st.simulations, st.leaf_evals - st.evals_reused)
        if act == 0 or t > a.max_seconds:
            break
    e.selfplay_end()
for r in rows:
    print(json.dumps(r))
print(json.dumps(dict(total_seconds=rows[-1]["t"], sims=rows[-1]["sims"], msims_per_s=round(rows[-1]["sims"] / rows[-1]["t"] / 1e6, 3), args=vars(a))))
