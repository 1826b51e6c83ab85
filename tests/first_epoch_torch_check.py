"""Is the first epoch's behaviour -- learning_status rises, the value loss sits at ~1.9 -- the device trainer's or the recipe's?

NOT a test (pytest does not collect it; it runs for about an hour on 8 cores): an INDEPENDENT fp64 torch restatement of the ResNet in
train mode (tests/test_train_gpu.py::TorchNet), torch.optim.Adam(2e-3) on mini-batches of 1024, Flux's running statistics (momentum
0.1, unbiased variance), the loss of learning.jl:67-90 at the shipped parameters (L2 1e-4, nonvalidity penalty 1, LOG_WEIGHT) -- on
self-play data the CPU oracle makes (256 Connect-Four games, uniform-oracle MCTS, symmetries, merge); after every step the whole data
set's TEST-mode status by the oracle's learning_status.  No GPU, no device code.

    python tests/first_epoch_torch_check.py <blocks> <filters> <steps>        # profiles/r5/first_epoch_torch_fp64_5x128.txt: 5 128 36

Result (5 x 128, the shipped network): the train-mode loss jumps from 2.35 to 5.3 with Adam's second step and settles at ~2.55; the
TEST-mode value loss goes from 0.89 to 1.94 within three steps and stays there for the rest of the run (the tanh value head saturated:
a constant +-1 against z = +-1), the whole-data status from 1.54 to 2.56 -- what the device shows in its first iteration (status
1.85 -> 2.66, Lv 0.85 -> 1.90, profiles/r5/README.md) and leaves again in its second.  The recipe does it, not the kernels."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "alphazero.jl_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import azref as R
from azhip.network import ResNetHP, random_params, split_params, param_layout
import importlib.util
spec = importlib.util.spec_from_file_location("ttg", os.path.join(ROOT, "tests", "test_train_gpu.py")); ttg = importlib.util.module_from_spec(spec); spec.loader.exec_module(ttg)

torch.set_num_threads(24)
game = R.C4
hp = ResNetHP(num_blocks=int(sys.argv[1]) if len(sys.argv) > 1 else 5, num_filters=int(sys.argv[2]) if len(sys.argv) > 2 else 64, num_policy_head_filters=32, num_value_head_filters=32)
blob = random_params(game, hp, seed=1)
# self-play data from a uniform-oracle MCTS (what a random network's first iteration looks like: noisy z, flat-ish pi)
t = time.time()
g, m, nm = R.simulate(game, R.ORACLE_UNIFORM, 256, 64, 100, cpuct=2.0, noise_eps=0.25, noise_alpha=1.0, temp_xs=(0, 20, 30), temp_ys=(1.0, 1.0, 0.3), reset_every=2, seed=3)
samples = []
for i in range(256):
    samples += list(R.samples_from_trace(game, m, g[i].first_move, g[i].num_moves))
samples = R.merge_by_state(game, R.augment_with_symmetries(game, samples))
W, X, A, P, V = R.convert_samples(game, 1, samples)   # LOG_WEIGHT?
print("samples", len(W), "t", time.time() - t)
st = R.learning_status(game, (hp.num_blocks, hp.num_filters, 32, 32), blob, (W, X, A, P, V))
print("test-mode status at init: L %.4f Lp %.4f Lv %.4f Lreg %.4f Linv %.4f Hp %.4f" % (st.L, st.Lp, st.Lv, st.Lreg, st.Linv, st.Hp))
net = ttg.TorchNet(game, hp, blob)
n = min(len(W), 1024)
idx = np.arange(n)
batch = (W[idx], X[idx], A[idx], P[idx], V[idx])
with torch.no_grad():
    L, (Lp, Lv, Lreg, Linv, sc) = net.losses(batch, float(W.mean()), float(st.Hp), 1e-4, 1.0, 1.0)
print("train-mode (batch statistics) at init: L %.4f Lp %.4f Lv %.4f Lreg %.4f Linv %.4f" % (L, Lp, Lv, Lreg, Linv))

# ---- Adam(2e-3) steps in torch (fp64), Flux running statistics (momentum 0.1, unbiased variance), test-mode loss by the C oracle
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
params = [t for k, t in net.p.items() if t.requires_grad]
opt = torch.optim.Adam(params, lr=2e-3, betas=(0.9, 0.999), eps=1e-8)
rng = np.random.default_rng(0)
Wm, Hp0 = float(W.mean()), float(st.Hp)
for s in range(steps):
    idx = rng.choice(len(W), size=min(1024, len(W)), replace=False)
    batch = (W[idx], X[idx], A[idx], P[idx], V[idx])
    opt.zero_grad()
    L, (Lp, Lv, Lreg, Linv, sc) = net.losses(batch, Wm, Hp0, 1e-4, 1.0, 1.0)
    L.backward()
    opt.step()
    with torch.no_grad():
        for pre, (mu, var, cnt) in net.batch_stats.items():
            net.p[pre + ".mean"].mul_(0.9).add_(0.1 * mu)
            net.p[pre + ".var"].mul_(0.9).add_(0.1 * var * cnt / (cnt - 1))
    if True:
        b = net.blob().astype(np.float32)
        t_ = R.learning_status(game, (hp.num_blocks, hp.num_filters, 32, 32), b, (W, X, A, P, V))
        print("step %3d train-mode batch L %.4f (Lp %.4f Lv %.4f) | test-mode whole data L %.4f Lp %.4f Lv %.4f Linv %.4f" % (s + 1, L.item(), Lp.item(), Lv.item(), t_.L, t_.Lp, t_.Lv, t_.Linv))
