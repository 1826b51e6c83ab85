"""Static checks of julia/AlphaZeroHIP.jl against include/azhip.h (the glue was written blind -- no Julia
in the image -- and nothing tied its `ccall` tuples and isbits structs to the header).  Pure text parsing, no Julia:
  * every `ccall((:az_x, LIB), Cint, (types...), args...)` names a function the header declares, has the header's
    arity, the same number of argument expressions as types, and each Julia type is ABI-compatible with the C type;
  * every Julia struct that mirrors a C record has the record's size and field offsets (C layout rules), checked
    against the ctypes mirrors in azhip/_lib.py, which the GPU tests exercise against the real library."""
import ctypes as C
import os
import re

from azhip import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = open(os.path.join(ROOT, "julia", "AlphaZeroHIP.jl")).read() + "\n" + open(os.path.join(ROOT, "julia", "AlphaZeroHIPExtras.jl")).read()   # core (the three seams of the hot path) + optional extras
HDR = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "azhip.h")).read(), flags=re.S)


def split_top(s):
    """split on commas that are not nested in (), {} or []"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def header_protos():
    protos = {}
    for m in re.finditer(r"\b(int|const char\*)\s+(az_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", HDR, flags=re.S):
        args = [] if m.group(3).strip() in ("void", "") else split_top(" ".join(m.group(3).split()))
        protos[m.group(2)] = [re.sub(r"\s*\b[a-zA-Z_][a-zA-Z0-9_]*(\[\w*\])?$", lambda mm: "*" if mm.group(1) else "", a).strip()
                              if not a.endswith("*") else a for a in args]
    return protos


HANDLES = ("az_engine", "az_memory", "az_dataset", "az_trainer", "az_comm")
JL_OK = {   # C parameter type -> Julia ccall types that pass it correctly
    "int32_t": {"Int32", "Cint"}, "int": {"Int32", "Cint"}, "int64_t": {"Int64"}, "double": {"Float64"}, "float": {"Float32"},
    "az_progress_cb": {"Ptr{Cvoid}"}, "void*": {"Ptr{Cvoid}"},
    "const float*": {"Ptr{Float32}"}, "float*": {"Ptr{Float32}", "Ref{Float32}"},
    "double*": {"Ptr{Float64}", "Ref{Float64}"}, "const double*": {"Ptr{Float64}"},
    "const uint64_t*": {"Ptr{NTuple{2,UInt64}}", "Ptr{UInt64}"}, "uint64_t*": {"Ptr{NTuple{2,UInt64}}", "Ptr{UInt64}"},
    "const int32_t*": {"Ptr{Int32}"}, "int32_t*": {"Ptr{Int32}", "Ref{Int32}"}, "int64_t*": {"Ptr{Int64}", "Ref{Int64}"},
    "int8_t*": {"Ptr{Int8}"}, "uint32_t*": {"Ptr{UInt32}", "Ref{UInt32}"}, "const uint32_t*": {"Ptr{UInt32}"},
    "char*": {"Ptr{UInt8}", "Cstring"}, "uint8_t*": {"Ptr{UInt8}"}, "const uint8_t*": {"Ptr{UInt8}"},
    "az_gather_stats*": {"Ref{GatherStats}"},
    "const az_engine_cfg*": {"Ref{EngineCfg}"}, "az_engine_cfg*": {"Ref{EngineCfg}"},
    "az_trace_buf*": {"Ref{TraceBuf}", "Ptr{Cvoid}"}, "const az_trace_buf*": {"Ref{TraceBuf}", "Ptr{Cvoid}"},
    "az_selfplay_stats*": {"Ref{SelfplayStats}"}, "const az_move_rec*": {"Ptr{MoveRec}"},
    "const az_sample*": {"Ptr{AzSample}"}, "az_sample*": {"Ptr{AzSample}"},
    "az_dataset_info*": {"Ref{DatasetInfo}"}, "az_learning_status_t*": {"Ref{LearningStatusRec}"},
    "const az_train_cfg*": {"Ref{TrainCfg}"}, "az_train_cfg*": {"Ref{TrainCfg}"}, "az_prof*": {"Ref{Prof}"},
}
for h in HANDLES:
    JL_OK[h + "*"] = {"Ptr{Cvoid}"}
    JL_OK["const " + h + "*"] = {"Ptr{Cvoid}"}
    JL_OK[h + "**"] = {"Ref{Ptr{Cvoid}}"}


def julia_ccalls():
    calls = []
    for m in re.finditer(r"ccall\(\(:(az_[a-z0-9_]+), LIB\),\s*([A-Za-z]+),\s*\(", JL):
        i, depth = m.end(), 1
        while depth:                                   # the type tuple
            depth += {"(": 1, ")": -1}.get(JL[i], 0)
            i += 1
        types = split_top(JL[m.end():i - 1])
        j, depth = i, 1                                # the rest of the ccall's argument list
        while depth:
            depth += {"(": 1, ")": -1}.get(JL[j], 0)
            j += 1
        rest = JL[i:j - 1].lstrip()
        args = split_top(rest[1:]) if rest.startswith(",") else []
        calls.append((m.group(1), m.group(2), types, args, JL.count("\n", 0, m.start()) + 1))
    return calls


def test_every_ccall_matches_the_header():
    protos = header_protos()
    calls = julia_ccalls()
    assert len(calls) >= 20
    for name, ret, types, args, line in calls:
        assert name in protos, "line %d: %s is not declared in include/azhip.h" % (line, name)
        want = protos[name]
        assert ret == ("Cstring" if name == "az_last_error" else "Cint"), (line, name, ret)
        assert len(types) == len(want), "line %d: %s takes %d arguments, the ccall passes %d types" % (line, name, len(want), len(types))
        assert len(args) == len(types), "line %d: %s: %d argument expressions for %d types" % (line, name, len(args), len(types))
        for k, (jt, ct) in enumerate(zip(types, want)):
            ct = " ".join(ct.split())
            assert ct in JL_OK, "line %d: %s arg %d: no rule for C type %r" % (line, name, k, ct)
            assert jt.replace(" ", "") in {x.replace(" ", "") for x in JL_OK[ct]}, \
                "line %d: %s arg %d: Julia %s does not match C %s" % (line, name, k, jt, ct)


SIZES = {"Int8": 1, "UInt8": 1, "Int32": 4, "UInt32": 4, "Float32": 4, "Int64": 8, "UInt64": 8, "Float64": 8, "Cint": 4}


def jl_type_layout(t):
    """(size, alignment) of a Julia isbits field type under the C layout rules"""
    t = t.replace(" ", "")
    if t.startswith("Ptr{"):
        return 8, 8
    m = re.fullmatch(r"NTuple\{(\d+),(\w+)\}", t)
    if m:
        s = SIZES[m.group(2)]
        return int(m.group(1)) * s, s
    return SIZES[t], SIZES[t]


def julia_struct(name):
    m = re.search(r"(?:mutable\s+)?struct\s+%s\b(.*?)\bend\b" % name, JL, flags=re.S)
    assert m, name
    body = re.sub(r"#.*", "", m.group(1))
    fields = []
    for part in re.split(r"[;\n]", body):
        mm = re.fullmatch(r"\s*(\w+)::(.+?)\s*", part)
        if mm:
            fields.append((mm.group(1), mm.group(2)))
    return fields


def c_offsets(fields):
    off, maxal, out = 0, 1, []
    for name, t in fields:
        size, al = jl_type_layout(t)
        off = (off + al - 1) // al * al
        out.append((name, off, size))
        off += size
        maxal = max(maxal, al)
    return out, (off + maxal - 1) // maxal * maxal


def test_isbits_structs_have_the_c_record_layout():
    pairs = [("EngineCfg", L.EngineCfg), ("MoveRec", L.MoveRec), ("GameRec", L.GameRec), ("TraceBuf", L.TraceBuf),
             ("SelfplayStats", L.SelfplayStats), ("AzSample", L.Sample), ("DatasetInfo", L.DatasetInfo),
             ("LearningStatusRec", L.LearningStatusRec), ("TrainCfg", L.TrainCfg), ("GatherStats", L.GatherStats)]
    for jname, ct in pairs:
        offs, size = c_offsets(julia_struct(jname))
        assert size == C.sizeof(ct), (jname, size, C.sizeof(ct))
        cf = [(n, getattr(ct, n).offset, getattr(ct, n).size) for n, _ in ct._fields_]
        assert len(offs) == len(cf), (jname, len(offs), len(cf))
        for (jn, jo, js), (cn, co, cs) in zip(offs, cf):
            assert (jo, js) == (co, cs), "%s.%s at %d (+%d) but the C record has %s at %d (+%d)" % (jname, jn, jo, js, cn, co, cs)
    m = re.search(r"@assert sizeof\(EngineCfg\) == (\d+) && sizeof\(MoveRec\) == (\d+) && sizeof\(GameRec\) == (\d+)", JL)
    assert m and [int(x) for x in m.groups()] == [C.sizeof(L.EngineCfg), C.sizeof(L.MoveRec), C.sizeof(L.GameRec)]


def test_node_footprint_formula_matches_the_device_record():
    """the glue reports approximate_memory_footprint from the device node size: NodeL (csrc/tree.h) + side record (key, Vest) + table share"""
    m = re.search(r"nbytes = (.+?)\s+#", JL)
    assert m
    expr = m.group(1).replace("cld", "_cld").replace("8nA", "8*nA").replace("2nA", "2*nA").replace("?", " and ").replace(":", " or ")
    for nA, node in ((7, 128), (6, 128), (9, 192)):
        hb = 2 if nA <= 8 else 4
        assert eval(expr, {"_cld": lambda a, b: -(-a // b), "nA": nA, "hb": hb}) == node + 32 + 12, nA


def test_every_entry_point_of_the_header_is_bound_or_listed_as_unbound():
    """the glue bound 15 of 57 entry points and nothing said which were missing on purpose.  Every function
    include/azhip.h declares must either be `ccall`ed by julia/AlphaZeroHIP.jl or be named, with a reason, in its
    `const UNBOUND = Dict(...)`; the list may not name functions that are bound or that do not exist."""
    protos = set(header_protos())
    bound = {c[0] for c in julia_ccalls()}
    m = re.search(r"const UNBOUND = Dict\((.*?)\n\)", JL, flags=re.S)
    assert m, "UNBOUND list not found"
    unbound = dict(re.findall(r":(az_[a-z0-9_]+)\s*=>\s*\"([^\"]+)\"", m.group(1)))
    assert all(len(r) > 10 for r in unbound.values())
    assert not (set(unbound) & bound), "listed as unbound but ccall'ed: %s" % sorted(set(unbound) & bound)
    assert not (set(unbound) - protos), "UNBOUND names functions the header does not declare: %s" % sorted(set(unbound) - protos)
    missing = protos - bound - set(unbound)
    assert not missing, "entry points of include/azhip.h neither bound nor listed in UNBOUND: %s" % sorted(missing)
    # the round-3 additions are really bound: the RCCL exchange, the device-only phase, the explorer seam
    for f in ("az_comm_unique_id", "az_comm_init", "az_comm_destroy", "az_comm_gather_push", "az_comm_broadcast_params",
              "az_memory_push_engine", "az_engine_release_phase", "az_mcts_explore", "az_mcts_node_stats", "az_mcts_counters", "az_mcts_reset"):
        assert f in bound, f
    assert "function AlphaZero.simulate_distributed(simulator::Simulator, gspec::DeviceGameSpec" in JL
    start = JL.index("function AlphaZero.simulate_distributed(")
    body = JL[start:JL.index("function device_self_play_step!", start)]
    # the rank form: divrem shard, device-only phase, ONE collective into the rank's device memory
    assert "shard_games(p.num_games, comm.world, comm.rank)" in body and "selfplay_device_only!(e, count, first" in body and "gather_push!(comm, e, memory, gamma)" in body


def test_shard_games_formula_is_the_python_mirror_s():
    """the glue's shard_games (divrem, remainder to rank 0) evaluated by text: same split as azhip.simulations.shard_games"""
    from azhip.simulations import shard_games
    assert "num_each, rem = divrem(num_games, world)" in JL and "counts = [r == 0 ? num_each + rem : num_each for r in 0:world-1]" in JL
    assert "return sum(counts[1:rank]), counts[rank + 1]" in JL
    for n, w in ((10, 4), (32768, 8), (7, 7)):
        each, rem = divmod(n, w)
        counts = [each + rem if r == 0 else each for r in range(w)]
        for r in range(w):
            assert shard_games(n, w, r) == (sum(counts[:r]), counts[r])


def test_julia_sources_are_structurally_balanced():
    """the glue and the golden-vector script have never been executed (no Julia in the image): at least every block is closed
    and every bracket paired (tools/julia_balance.py), and the checker does notice a missing `end` / bracket"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "julia_balance.py")
    for f in ("julia/AlphaZeroHIP.jl", "julia/AlphaZeroHIPExtras.jl", "tools/gen_golden.jl"):
        r = subprocess.run([sys.executable, tool, os.path.join(root, f)], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.startswith("balanced"), (f, r.stdout)
    src = open(os.path.join(root, "julia", "AlphaZeroHIP.jl")).read()
    import tempfile
    for broken in (src.replace("\nend\n", "\n\n", 1), src.replace("ccall((", "ccall(", 1)):
        with tempfile.NamedTemporaryFile("w", suffix=".jl", delete=False) as tf:
            tf.write(broken)
        r = subprocess.run([sys.executable, tool, tf.name], capture_output=True, text=True)
        os.unlink(tf.name)
        assert r.returncode == 1, r.stdout
