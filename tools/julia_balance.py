"""Structural check of a Julia source without Julia: strings / comments stripped, every block opener (module, function,
struct, if, for, while, let, begin, do, try, macro, quote, abstract / primitive type) outside brackets is closed by an `end`,
brackets pair up.  Not a parser -- it catches the slips a file that was never executed would otherwise carry to its first user.
usage: tools/julia_balance.py file.jl   (prints `balanced: N lines` or one line per problem; exit status 1 on problems)"""
import re, sys
src = open(sys.argv[1]).read()
# strip: triple-quoted strings, strings (with interpolation kept simple), chars, comments (#= =#, #)
out = []
i = 0; n = len(src)
def skip_string(i, quote):
    j = i + len(quote)
    while j < n:
        if src[j] == '\\': j += 2; continue
        if src.startswith(quote, j): return j + len(quote)
        j += 1
    return n
while i < n:
    c = src[i]
    if src.startswith('#=', i):
        j = src.find('=#', i + 2); i = n if j < 0 else j + 2; continue
    if c == '#':
        j = src.find('\n', i); i = n if j < 0 else j; continue
    if src.startswith('"""', i): i = skip_string(i, '"""'); out.append(' "" '); continue
    if c == '"': i = skip_string(i, '"'); out.append(' "" '); continue
    if c == "'" and i + 2 < n and (src[i+2] == "'" or (src[i+1] == '\\' and "'" in src[i+2:i+6])):
        j = src.find("'", i + 2 if src[i+1] != '\\' else i + 3); i = j + 1; out.append(" 'c' "); continue
    out.append(c); i += 1
code = ''.join(out)
tokens = re.findall(r"[A-Za-z_@!][A-Za-z_0-9!]*|[()\[\]{}]|\n|:|\S", code)
depth = 0; stack = []; line = 1; errors = []
openers = {"module", "baremodule", "function", "struct", "if", "for", "while", "let", "begin", "do", "try", "macro", "quote"}
prev = None
for k, t in enumerate(tokens):
    if t == '\n': line += 1; prev = t; continue
    if t in '([{': stack.append((t, line)); depth += 1
    elif t in ')]}':
        if not stack or stack[-1][0] in openers | {"type"}: errors.append("line %d: unmatched %s" % (line, t))
        else:
            o, _ = stack.pop(); depth -= 1
            if "([{".index(o) != ")]}".index(t): errors.append("line %d: %s closes %s" % (line, t, o))
    elif depth_br := sum(1 for s, _ in stack if s in '([{'):
        pass                                                   # keywords inside brackets: comprehensions, a[end], ternaries
    elif t in openers and prev != ':':                         # :if etc. are symbols
        if t == "struct" and stack and stack[-1][0] == "mutable": stack.pop()
        stack.append((t, line))
    elif t == "mutable": stack.append((t, line))
    elif t == "type" and prev in ("abstract", "primitive"): stack.append((t, line))
    elif t == "end" and prev != ':':
        if not stack or stack[-1][0] in '([{': errors.append("line %d: unmatched end" % line)
        else: stack.pop()
    prev = t
for s, l in stack: errors.append("line %d: %s never closed" % (l, s))
print("\n".join(errors) if errors else "balanced: %d lines" % line)
sys.exit(1 if errors else 0)
