"""pyref.py -- an INDEPENDENT pure-Python restatement of the search path, used only to cross-check the C
oracle (tests/test_pyref.py).  Test infrastructure, like everything under oracle/.

It restates, directly from the reference sources and without looking at azref.c's data structures:
  games/connect-four/game.jl, games/tictactoe/game.jl, games/mancala/game.jl  (rules, reward, mask)
  src/mcts.jl:157-271   (state_info, uct_scores, run_simulation!, explore!, policy)
with Python floats as Float64 and numpy.float32 for the priors.  The only shared piece is the integer
hash of the synthetic oracle (az_mix64 / az_hash_key of include/az_numerics.h), restated here too.
"""
import math

import numpy as np

M64 = (1 << 64) - 1


# ---------------------------------------------------------------- games (immutable tuple states)
class ConnectFour:
    A = 7

    @staticmethod
    def init():
        return (tuple([0] * 42), 1, False, 0)          # board[col + 7*row], curplayer, finished, winner

    @staticmethod
    def mask(g):
        b = g[0]
        return [b[c + 7 * 5] == 0 for c in range(7)]

    @staticmethod
    def _connected(b, p, col, row, dc, dr):
        n, c, r = 0, col + dc, row + dr
        while 0 <= c < 7 and 0 <= r < 6 and b[c + 7 * r] == p:
            n, c, r = n + 1, c + dc, r + dr
        return n

    @classmethod
    def play(cls, g, col):
        b, cur, _, _ = g
        row = 0
        while b[col + 7 * row] != 0:
            row += 1
        b = list(b)
        b[col + 7 * row] = cur
        win = any(1 + cls._connected(b, cur, col, row, dc, dr) + cls._connected(b, cur, col, row, -dc, -dr) >= 4
                  for dc, dr in ((1, 1), (1, -1), (1, 0), (0, 1)))
        fin = win or all(b[c + 35] != 0 for c in range(7))
        return (tuple(b), 3 - cur, fin, cur if win else 0)

    @staticmethod
    def key(g):
        a = b = 0
        for col in range(7):
            for row in range(6):
                v = g[0][col + 7 * row]
                if v == 1:
                    a |= 1 << (col * 7 + row)
                if v == 2:
                    b |= 1 << (col * 7 + row)
        if g[1] == 2:
            a |= 1 << 63
        return a, b

    @staticmethod
    def reward(g):
        return (1.0 if g[3] == 1 else -1.0 if g[3] == 2 else 0.0) if g[2] else 0.0

    @staticmethod
    def symmetries(g):
        """GI.symmetries, games/connect-four/game.jl:247-257: board[col, row] for col in reverse(1:NUM_COLS)"""
        b = g[0]
        return [(tuple(b[(6 - c) + 7 * r] for r in range(6) for c in range(7)), g[1], g[2], g[3])]


class TicTacToe:
    A = 9
    AL = [(0, 3, 6), (1, 4, 7), (2, 5, 8), (0, 1, 2), (3, 4, 5), (6, 7, 8), (0, 4, 8), (2, 4, 6)]

    @classmethod
    def _status(cls, b):
        for p in (1, 2):
            if any(all(b[i] == p for i in al) for al in cls.AL):
                return True, p
        return (all(x != 0 for x in b), 0)

    @classmethod
    def init(cls):
        return (tuple([0] * 9), 1, False, 0)

    @staticmethod
    def mask(g):
        return [x == 0 for x in g[0]]

    @classmethod
    def play(cls, g, pos):
        b = list(g[0])
        b[pos] = g[1]
        fin, w = cls._status(b)
        return (tuple(b), 3 - g[1], fin, w)

    @staticmethod
    def key(g):
        a = sum(1 << i for i, x in enumerate(g[0]) if x == 1)
        b = sum(1 << i for i, x in enumerate(g[0]) if x == 2)
        return (a | (1 << 63) if g[1] == 2 else a), b

    @staticmethod
    def reward(g):
        return (1.0 if g[3] == 1 else -1.0 if g[3] == 2 else 0.0) if g[2] else 0.0

    @staticmethod
    def symmetries(g):
        """GI.symmetries, games/tictactoe/game.jl:149-168: the seven non-trivial dihedral maps; image board = board[sym]
        with sym[p] = pos_of_xy(f(xy_of_pos(p))), positions 1-based, pos_of_xy((x, y)) = (y-1)*3 + x (game.jl:39-41)."""
        N = 3
        rot = lambda xy: (xy[1], N - xy[0] + 1)
        flip = lambda xy: (xy[0], N - xy[1] + 1)
        comp = lambda f, h: (lambda xy: f(h(xy)))
        rot2 = comp(rot, rot)
        rot3 = comp(rot2, rot)
        xy_of = lambda p: ((p - 1) % N + 1, (p - 1) // N + 1)
        pos_of = lambda xy: (xy[1] - 1) * N + (xy[0] - 1) + 1
        out = []
        for f in (rot, rot2, rot3, flip, comp(flip, rot), comp(flip, rot2), comp(flip, rot3)):
            sym = [pos_of(f(xy_of(p))) for p in range(1, 10)]
            out.append((tuple(g[0][q - 1] for q in sym), g[1], g[2], g[3]))
        return out


class Mancala:
    A = 6

    @staticmethod
    def init():
        return ((3,) * 6, (3,) * 6, 0, 0, 1, False)        # houses W, houses B, store W, store B, cur, finished

    @staticmethod
    def mask(g):
        return [h > 0 for h in (g[0] if g[4] == 1 else g[1])]

    @staticmethod
    def play(g, a):
        hw, hb, sw, sb, cur, _ = g
        houses = {1: list(hw), 2: list(hb)}
        stores = {1: sw, 2: sb}
        other = 3 - cur
        pos = ("h", cur, a + 1)
        n = houses[cur][a]
        houses[cur][a] = 0

        def nxt(p):
            if p[0] == "s":
                return ("h", other, 6)
            if p[2] > 1:
                return ("h", p[1], p[2] - 1)
            return ("s", cur) if p[1] == cur else ("h", cur, 6)
        for _ in range(n):
            pos = nxt(pos)
            if pos[0] == "s":
                stores[pos[1]] += 1
            else:
                houses[pos[1]][pos[2] - 1] += 1

        def leftovers(p):
            stores[p] += sum(houses[p])
            houses[1] = [0] * 6
            houses[2] = [0] * 6
        fin, newcur = False, cur
        if sum(houses[cur]) == 0:
            leftovers(other)
            fin = True
        elif pos[0] == "h":
            early = False
            if houses[pos[1]][pos[2] - 1] == 1 and pos[1] == cur:
                opp = 6 - pos[2]                      # opposite house index (0-based) = 6 - num
                stores[cur] += houses[other][opp] + 1
                houses[cur][pos[2] - 1] = 0
                houses[other][opp] = 0
                if sum(houses[other]) == 0:
                    leftovers(cur)
                    fin = early = True
                elif sum(houses[cur]) == 0:
                    leftovers(other)
                    fin = early = True
            if not early:
                newcur = other
        return (tuple(houses[1]), tuple(houses[2]), stores[1], stores[2], newcur, fin)

    @staticmethod
    def key(g):
        a = sum(h << (8 * i) for i, h in enumerate(g[0])) | (g[2] << 48)
        b = sum(h << (8 * i) for i, h in enumerate(g[1])) | (g[3] << 48)
        return (a | (1 << 63) if g[4] == 2 else a), b

    @staticmethod
    def reward(g):
        if not g[5]:
            return 0.0
        return 1.0 if g[2] > g[3] else -1.0 if g[2] < g[3] else 0.0


def finished(G, g):
    return g[5] if G is Mancala else g[2]


def white_playing(G, g):
    return (g[4] if G is Mancala else g[1]) == 1


def state_of(G, g):
    """the key of the Dict: (board, curplayer) -- env-only fields (finished, winner) excluded"""
    return (g[0], g[1], g[2], g[3], g[4]) if G is Mancala else (g[0], g[1])


# ---------------------------------------------------------------- synthetic oracle (integer hash)
def mix64(x):
    x &= M64
    x ^= x >> 30
    x = (x * 0xbf58476d1ce4e5b9) & M64
    x ^= x >> 27
    x = (x * 0x94d049bb133111eb) & M64
    x ^= x >> 31
    return x


def hash_oracle(G, g):
    a, b = G.key(g)
    h = mix64(a ^ mix64((b + 0x9e3779b97f4a7c15) & M64))
    m = G.mask(g)
    raw = [np.float32(1 + (mix64((h + i + 1) & M64) & 0xffff)) if m[i] else np.float32(0) for i in range(G.A)]
    s = np.float32(0)
    for r in raw:
        s = np.float32(s + r)
    P = [np.float32(r / s) for r, ok in zip(raw, m) if ok]
    V = np.float32(np.float32((mix64((h + 99) & M64) & 0xffff) - 32768) / np.float32(65536.0))
    return P, V


def uniform_oracle(G, g):
    n = sum(G.mask(g))
    return [np.float32(1.0 / n)] * n, np.float32(0.0)


# ---------------------------------------------------------------- MCTS (src/mcts.jl)
class Mcts:
    def __init__(self, G, oracle, gamma=1.0, cpuct=1.0, eps=0.0):
        self.G, self.oracle, self.gamma, self.cpuct, self.eps = G, oracle, gamma, cpuct, eps
        self.tree = {}
        self.total_simulations = 0
        self.total_nodes_traversed = 0

    def run_simulation(self, g, eta, root=True):
        G = self.G
        if finished(G, g):
            return 0.0
        st = state_of(G, g)
        acts = [a for a, ok in enumerate(G.mask(g)) if ok]
        if st not in self.tree:
            P, V = self.oracle(G, g)
            self.tree[st] = [[p, 0.0, 0] for p in P] + [V]
            return float(V)
        info = self.tree[st]
        stats = info[:-1]
        eps = self.eps if root else 0.0
        sqrtN = math.sqrt(sum(s[2] for s in stats))
        best, bests = 0, None
        for i, (P, W, N) in enumerate(stats):
            Q = W / max(N, 1)
            Pp = float(P) if eps == 0 else (1 - eps) * float(P) + eps * eta[i]
            sc = Q + self.cpuct * Pp * sqrtN / (N + 1)
            if bests is None or sc > bests:
                best, bests = i, sc
        wp = white_playing(G, g)
        g2 = G.play(g, acts[best])
        wr = G.reward(g2)
        r = wr if wp else -wr
        pswitch = wp != white_playing(G, g2)
        qn = self.run_simulation(g2, eta, False)
        qn = -qn if pswitch else qn
        q = r + self.gamma * qn
        stats[best][1] += q
        stats[best][2] += 1
        self.total_nodes_traversed += 1
        return q

    def explore(self, g, nsims, eta):
        for _ in range(nsims):
            self.total_simulations += 1
            self.run_simulation(g, eta)

    def root_stats(self, g):
        info = self.tree[state_of(self.G, g)]
        return [s[2] for s in info[:-1]], [s[1] for s in info[:-1]], [s[0] for s in info[:-1]], info[-1]


GAMES = {0: ConnectFour, 1: TicTacToe, 2: Mancala}
