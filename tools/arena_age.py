"""Does an arena evaluation get slower with the age of the process (engines / streams created and destroyed before it)?
bench.py's whole default run measured its last arena at 8.2 s against 5.6 s for the same games early in a process."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alphazero.jl_amd"))
import azhip  # noqa: E402
from azhip.network import ResNetHP, random_params  # noqa: E402

gspec = azhip.ConnectFourSpec()
hp = azhip.ResNetHP(num_blocks=5, num_filters=128, num_policy_head_filters=32, num_value_head_filters=32)
c, b = azhip.ResNet(gspec, hp, seed=1), azhip.ResNet(gspec, hp, seed=2)
mp = azhip.MctsParams(num_iters_per_turn=600, cpuct=2.0, dirichlet_noise_ϵ=0.05, dirichlet_noise_α=1.0, temperature=azhip.ConstSchedule(0.2))
params = azhip.ArenaParams(mcts=mp, sim=azhip.SimParams(num_games=128, num_workers=128, batch_size=128, use_gpu=True, reset_every=2,
                                                        flip_probability=0.5, alternate_colors=True), update_threshold=0.05)


def arena(tag):
    t0 = time.perf_counter()
    ev = azhip.compare_networks(gspec, c, b, params)
    print("%-40s %.2f s  avgr %.3f" % (tag, time.perf_counter() - t0, ev.avgr), flush=True)


arena("fresh process")
arena("again (cached engines)")
what = sys.argv[1] if len(sys.argv) > 1 else "engines"
if what == "engines":
    blob = random_params(0, ResNetHP(num_blocks=5, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32), seed=1)
    for i in range(8):                                               # two-group engines: 5 streams each, a few GB of pools
        with azhip.Engine(game=0, oracle=azhip.ORACLE_RESNET, num_workers=4096, batch_size=2048, num_iters_per_turn=400, num_blocks=5, num_filters=64,
                          num_policy_head_filters=32, num_value_head_filters=32) as e:
            e.net_set_params(blob)
            e.selfplay_begin(-1, 0); e.selfplay_step(50); e.selfplay_end()
    arena("after 8 two-group engines came and went")
elif what == "memory":
    import ctypes as C
    from azhip._lib import lib
    mems = [azhip.MemoryBuffer(gspec, 40_000_000) for _ in range(6)]  # 6 x 4.5 GB held
    arena("with 27 GB of replay memories held")
    [m.close() for m in mems]
    arena("after releasing them")
elif what == "alive":
    # engines that stay ALIVE (the bench's engine cache): their streams keep their share of the hardware queues
    blob = random_params(0, ResNetHP(num_blocks=5, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32), seed=1)
    held = []
    for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
        e = azhip.Engine(game=0, oracle=azhip.ORACLE_RESNET, num_workers=1024 + 64 * i, batch_size=512 + 32 * i, num_iters_per_turn=100, num_blocks=5,
                         num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
        e.net_set_params(blob)
        e.selfplay_begin(-1, 0); e.selfplay_step(20); e.selfplay_end()
        held.append(e)
        if i in (0, 1, 3, 7, 11, 15):
            arena("%d two-group engines alive (cached arena engines)" % (i + 1))
    azhip.engine.clear_engine_cache()
    arena("%d alive, NEW arena engines" % len(held))
    [e.close() for e in held]
    arena("all closed, same arena engines")
elif what == "load":
    # the bench's order: a long self-play phase of the big network first, the arena right behind it.  Is it the chip's state
    # (clocks under a power / temperature limit) rather than anything in the process?
    import subprocess

    def smi(tag):
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True)
        keep = [ln.strip() for ln in r.stdout.splitlines() if any(k in ln for k in ("sclk", "Power", "junction", "edge", "mclk"))]
        print("  [%s] %s" % (tag, " | ".join(keep)), flush=True)
    smi("idle")
    blob = random_params(0, hp, seed=3)
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 45.0
    with azhip.Engine(game=0, oracle=azhip.ORACLE_RESNET, num_workers=4096, batch_size=2048, num_iters_per_turn=600, num_blocks=5, num_filters=128,
                      num_policy_head_filters=32, num_value_head_filters=32) as e:
        e.net_set_params(blob)
        e.selfplay_begin(-1, 0)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < secs:
            e.selfplay_step(600)
        smi("under load, %.0f s in" % (time.perf_counter() - t0))
        e.selfplay_end()
    arena("right after %.0f s of 5x128 self-play" % secs)
    smi("after that arena")
    arena("once more")
    time.sleep(30)
    smi("after 30 s idle")
    arena("after 30 s idle")
azhip.engine.clear_engine_cache()
arena("fresh engines, old process")
