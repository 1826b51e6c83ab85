"""The oracle under oracle/ is the checker, never the product: only tests/, __graft_entry__.smoke() / build() and bench.py's
cpu_baseline leg may import, link or execute anything of it.  A static walk over the shipped sources, plus the loader's
behaviour when the HIP library is missing (it fails, it does not fall back)."""
import ast
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MARKS = re.compile(r"azref|libazref|ref_numerics|pyref|oracle/|[\"']oracle[\"']")       # bench.py / __graft_entry__.py: also the directory by name
NAMES = re.compile(r"azref|libazref|ref_numerics|pyref")                                 # product sources ("oracle" alone is MCTS vocabulary: mcts.jl:6-17)


def _files(top, exts):
    for d, _, fs in os.walk(os.path.join(ROOT, top)):
        if "__pycache__" in d:
            continue
        for f in fs:
            if f.endswith(exts):
                yield os.path.join(d, f)


def test_product_sources_never_name_the_oracle():
    bad = []
    for top, exts in (("alphazero.jl_amd", (".py", ".hip", ".h", ".cpp", ".c")), ("include", (".h",)), ("julia", (".jl",)),
                      ("examples", (".py", ".c", ".cpp", ".h"))):
        for p in _files(top, exts):
            for n, line in enumerate(open(p, errors="replace"), 1):
                code = line.split("#", 1)[0] if p.endswith((".py", ".jl")) else line.split("//", 1)[0]
                if NAMES.search(code):
                    bad.append("%s:%d: %s" % (os.path.relpath(p, ROOT), n, line.strip()))
    assert not bad, "\n".join(bad)


def _functions_touching_the_oracle(path):
    tree = ast.parse(open(path).read())
    hits = set()
    for fn in [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef))]:      # top level; nested ones belong to their parent
        src = ast.get_source_segment(open(path).read(), fn)
        body = "\n".join(l.split("#", 1)[0] for l in src.splitlines())
        body = re.sub(r'"""(?:.|\n)*?"""', "", body, count=1)              # the docstring may talk about it
        if MARKS.search(body):
            hits.add(fn.name)
    top = "\n".join(ast.get_source_segment(open(path).read(), n) or "" for n in tree.body
                    if not isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef, ast.Expr)))
    return hits, bool(MARKS.search("\n".join(l.split("#", 1)[0] for l in top.splitlines())))


def test_bench_touches_the_oracle_only_in_its_cpu_baseline_leg():
    hits, at_top = _functions_touching_the_oracle(os.path.join(ROOT, "bench.py"))
    assert hits == {"cpu_baseline"} and not at_top, hits


def test_graft_entry_touches_the_oracle_only_in_build_and_smoke():
    hits, at_top = _functions_touching_the_oracle(os.path.join(ROOT, "__graft_entry__.py"))
    assert hits <= {"build", "smoke", "_build_oracle", "_oracle"} and {"smoke"} <= hits and not at_top, hits


def test_a_missing_hip_library_is_an_error_not_a_fallback(tmp_path):
    """AZHIP_LIB pointing nowhere: importing the package's loader raises; nothing else gets loaded in its place"""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from azhip import _lib\n"
            "try:\n    _lib.lib()\nexcept Exception as e:\n    print('RAISED', type(e).__name__); sys.exit(0)\n"
            "print('LOADED'); sys.exit(1)\n") % os.path.join(ROOT, "alphazero.jl_amd")
    env = dict(os.environ, AZHIP_LIB=str(tmp_path / "no_such_libazhip.so"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "RAISED" in r.stdout, r.stdout + r.stderr
    assert "azref" not in r.stdout + r.stderr
