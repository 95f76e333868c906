"""Resolve the #if / #elif / #else / #endif blocks whose condition mentions only
macros given on the command line (NAME=VALUE), leave every other directive
alone.  Used to turn settled compile-time knobs into plain code.
   python tools/unifdef_lite.py file NAME=VALUE [NAME=VALUE ...]"""
import re
import sys

path = sys.argv[1]
known = dict(a.split("=") for a in sys.argv[2:])


def evaluate(cond):
    names = set(re.findall(r"[A-Za-z_]\w*", cond)) - {"defined"}
    if not names or not names <= set(known):
        return None
    expr = cond
    for n in names:
        expr = re.sub(r"\b%s\b" % n, known[n], expr)
    expr = expr.replace("&&", " and ").replace("||", " or ")
    expr = re.sub(r"!(?!=)", " not ", expr)
    return bool(eval(expr))


out = []
stack = []  # entries: dict(resolved, taken_before, active)
for line in open(path).read().split("\n"):
    st = line.strip()
    m = re.match(r"#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)", st)
    emit = all(f["active"] for f in stack if f["resolved"])
    if m:
        kind, rest = m.group(1), m.group(2).strip()
        if kind in ("if", "ifdef", "ifndef"):
            val = None
            if kind == "if":
                val = evaluate(rest)
            elif rest in known:
                val = kind == "ifdef"
            if kind == "ifndef" and rest in known:
                val = False
            stack.append({"resolved": val is not None, "active": bool(val),
                          "taken": bool(val)})
            if val is None and emit:
                out.append(line)
            continue
        f = stack[-1]
        if kind == "elif":
            if f["resolved"]:
                val = evaluate(rest)
                assert val is not None, line
                f["active"] = (not f["taken"]) and val
                f["taken"] = f["taken"] or val
            elif all(g["active"] for g in stack[:-1] if g["resolved"]):
                out.append(line)
            continue
        if kind == "else":
            if f["resolved"]:
                f["active"] = not f["taken"]
            elif all(g["active"] for g in stack[:-1] if g["resolved"]):
                out.append(line)
            continue
        if kind == "endif":
            stack.pop()
            if not f["resolved"] and all(g["active"] for g in stack if g["resolved"]):
                out.append(line)
            continue
    if emit:
        out.append(line)
open(path, "w").write("\n".join(out))
