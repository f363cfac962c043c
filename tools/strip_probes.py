#!/usr/bin/env python3
"""Mini unifdef: removes preprocessor branches guarded by the named macros from the product sources, taking every named macro as
UNDEFINED. Handles #ifdef X / #ifndef X / #if defined(X) / #elif defined(Y) / #else / #endif with nesting; a chain is only rewritten when
its FIRST condition names a listed macro (later #elif conditions that name unlisted macros are kept as the head of the chain).
usage: tools/strip_probes.py MACRO[,MACRO...] file..."""
import re, sys

def cond_of(line):
    s = line.strip()
    m = re.match(r"#\s*ifdef\s+(\w+)", s)
    if m: return ("ifdef", m.group(1))
    m = re.match(r"#\s*ifndef\s+(\w+)", s)
    if m: return ("ifndef", m.group(1))
    m = re.match(r"#\s*if\s+defined\s*\(\s*(\w+)\s*\)\s*(//.*)?$", s)
    if m: return ("ifdef", m.group(1))
    m = re.match(r"#\s*elif\s+defined\s*\(\s*(\w+)\s*\)\s*(//.*)?$", s)
    if m: return ("elifdef", m.group(1))
    if re.match(r"#\s*if\b", s): return ("if", None)
    if re.match(r"#\s*elif\b", s): return ("elif", None)
    if re.match(r"#\s*else\b", s): return ("else", None)
    if re.match(r"#\s*endif\b", s): return ("endif", None)
    return None

def strip(lines, dead):
    out = []; stack = []  # frames: dict(ours, state) state in {"keep","drop","done"}; for foreign frames ours=False
    def emitting(): return all(f["state"] == "keep" for f in stack if f["ours"])
    for ln in lines:
        c = cond_of(ln)
        if c is None:
            if emitting(): out.append(ln)
            continue
        kind, name = c
        if kind in ("ifdef", "ifndef", "if"):
            if name in dead and kind != "if":
                stack.append({"ours": True, "state": "drop" if kind == "ifdef" else "keep", "reopened": False})
            else:
                if emitting(): out.append(ln)
                stack.append({"ours": False, "state": "keep"})
        elif kind in ("elifdef", "elif", "else"):
            f = stack[-1]
            if not f["ours"]:
                if emitting(): out.append(ln)
            elif f.get("reopened"):      # the chain continues as a foreign one
                parent_emit = all(g["state"] == "keep" for g in stack[:-1] if g["ours"])
                if parent_emit: out.append(ln)
            elif f["state"] == "keep":   # an #ifndef DEAD branch was taken: the rest of the chain is dropped
                f["state"] = "done"
            elif f["state"] == "drop":
                if kind == "else": f["state"] = "keep"
                elif kind == "elifdef" and name in dead: pass
                else:                      # #elif on a foreign condition: becomes the head of a foreign chain
                    parent_emit = all(g["state"] == "keep" for g in stack[:-1] if g["ours"])
                    head = re.sub(r"#(\s*)elif", r"#\1if", ln, count=1)
                    if parent_emit: out.append(head)
                    f["reopened"] = True; f["state"] = "keep"
        elif kind == "endif":
            f = stack.pop()
            if not f["ours"] or f.get("reopened"):
                if emitting(): out.append(ln)
    assert not stack
    return out

dead = set(sys.argv[1].split(","))
for path in sys.argv[2:]:
    src = open(path).read().splitlines(keepends=True)
    res = strip(src, dead)
    if res != src:
        open(path, "w").writelines(res); print(path, len(src), "->", len(res), "lines")
