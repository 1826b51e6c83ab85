"""azhip -- host-side mirror of AlphaZero.jl's self-play interface over the MI355X engine.

The names follow the reference (MctsParams, SimParams, ResNetHP, ResNet, MctsPlayer, Simulator,
simulate, simulate_distributed, Trace, push_trace ...); every call ends in the C ABI of
include/azhip.h (libazhip.so, hand-written HIP for gfx950).  There is no CPU fallback.
"""
from . import _lib
from ._lib import (AzError, GAME_CONNECT_FOUR, GAME_GO9_PLANES, GAME_MANCALA, GAME_TICTACTOE, ORACLE_HASH, ORACLE_RESNET,
                   ORACLE_ROLLOUT, ORACLE_UNIFORM)
from .engine import Engine, cached_engine, clear_engine_cache, default_cfg
from .comm import Comm
from .params import ArenaParams, ConstSchedule, MctsParams, PLSchedule, SimParams
from .game import ConnectFourSpec, GameEnv, GameSpec, MancalaSpec, TicTacToeSpec
from .network import ResNet, ResNetHP
from . import mcts as MCTS
from .play import MctsPlayer, NetworkPlayer, PlayerWithTemperature, TwoPlayers, flipped_colors, play_game
from .trace import Trace
from .memory import Dataset, MemoryBuffer, TrainingSample, push_trace
from .simulations import Simulator, record_trace, self_play_measurements, simulate, simulate_distributed
from .training import SelfPlayParams, SelfPlayReport, broadcast_params, self_play_step
from .arena import Evaluation, compare_networks, pit_networks, pit_players
from . import benchmark as Benchmark
from .learning import (CONSTANT_WEIGHT, LINEAR_WEIGHT, LOG_WEIGHT, Adam, CyclicNesterov, LearningParams, LearningStatus, Loss,
                       Samples, Trainer, memory_report)
