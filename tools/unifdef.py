#!/usr/bin/env python3
"""Partial preprocessor used for the round-6 prune: resolves the #if / #ifdef / #ifndef / #elif / #else / #endif blocks whose condition depends
only on the macros given on the command line, substitutes those macros' values in the remaining text and leaves everything else untouched.

    python tools/unifdef.py file.hip -DNAME=VALUE ... -UNAME ...      (rewrites the file in place; --check prints a diff summary only)

-DNAME=V: NAME is defined with integer value V (its `#ifndef NAME / #define NAME V / #endif` default block disappears, uses become V).
-UNAME:   NAME is never defined (`#ifdef NAME` blocks disappear)."""
import re
import sys


def evaluate(expr, defs, undefs):
    """-> int value, or None when the expression mentions an unknown identifier."""
    def repl_defined(m):
        n = m.group(1) or m.group(2)
        if n in defs:
            return "1"
        if n in undefs:
            return "0"
        return "__UNKNOWN__"
    e = re.sub(r"defined\s*\(\s*(\w+)\s*\)|defined\s+(\w+)", repl_defined, expr)
    e = re.sub(r"\b([A-Za-z_]\w*)\b", lambda m: str(defs[m.group(1)]) if m.group(1) in defs else ("0" if m.group(1) in undefs else "__UNKNOWN__"), e)
    if "__UNKNOWN__" in e:
        return None
    e = e.replace("&&", " and ").replace("||", " or ").replace("!", " not ").replace(" not =", "!=")
    return int(bool(eval(e, {"__builtins__": {}})))     # noqa: S307  (our own sources)


def process(text, defs, undefs):
    out = []
    stack = []      # frames: dict(kind="resolved"|"pass", taken=bool, done=bool)
    lines = text.split("\n")
    i = 0
    while i < len(lines):
        line = lines[i]
        full = line
        while full.rstrip().endswith("\\") and i + 1 < len(lines) and re.match(r"\s*#", line):
            i += 1
            full += "\n" + lines[i]
        m = re.match(r"\s*#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)", line, re.S)
        live = all(f["taken"] for f in stack if f["kind"] == "resolved")
        if m:
            d, rest = m.group(1), re.sub(r"/\*.*?\*/|//.*", "", m.group(2)).strip()
            if d in ("if", "ifdef", "ifndef"):
                if not live:
                    stack.append(dict(kind="dead"))
                else:
                    v = evaluate(rest if d == "if" else (f"defined({rest})" if d == "ifdef" else f"!defined({rest})"), defs, undefs)
                    if v is None:
                        stack.append(dict(kind="pass", taken=True))
                        out.append(full)
                    else:
                        stack.append(dict(kind="resolved", taken=bool(v), done=bool(v)))
            elif d in ("elif", "else"):
                f = stack[-1]
                if f["kind"] == "pass":
                    out.append(full)
                elif f["kind"] == "resolved":
                    if f["done"]:
                        f["taken"] = False
                    elif d == "else":
                        f["taken"] = f["done"] = True
                    else:
                        v = evaluate(rest, defs, undefs)
                        if v is None:
                            raise SystemExit(f"unresolvable #elif after a resolved #if: {line}")
                        f["taken"] = f["done"] = bool(v)
            else:
                f = stack.pop()
                if f["kind"] == "pass":
                    out.append(full)
        elif live and not any(f["kind"] == "dead" for f in stack):
            if defs and not re.match(r"\s*#\s*define\s+(" + "|".join(map(re.escape, defs)) + r")\b", full):
                full = re.sub(r"\b(" + "|".join(map(re.escape, defs)) + r")\b", lambda mm: str(defs[mm.group(1)]), full)
                out.append(full)
            elif not defs:
                out.append(full)
        i += 1
    assert not stack, "unbalanced conditionals"
    return "\n".join(out)


def main():
    files = [a for a in sys.argv[1:] if not a.startswith("-")]
    defs, undefs = {}, set()
    for a in sys.argv[1:]:
        if a.startswith("-D"):
            n, _, v = a[2:].partition("=")
            defs[n] = int(v or 1)
        elif a.startswith("-U"):
            undefs.add(a[2:])
    for f in files:
        src = open(f).read()
        dst = process(src, defs, undefs)
        if "--check" in sys.argv:
            print(f, len(src.split("\n")), "->", len(dst.split("\n")), "lines")
        else:
            open(f, "w").write(dst)


if __name__ == "__main__":
    main()
