#!/usr/bin/env python3
"""A small unifdef: resolve preprocessor conditionals on a given set of symbols (treated as defined / undefined), leave every other conditional
alone.  Used once in round 4 to take the laboratory switches out of csrc/qr_mpc_kernel.hip; kept for the next time.
   tools/unifdef.py FILE -U SYM ... -D SYM ...   (writes FILE in place; `#if 0` / `#if 1` are resolved too)"""
import re
import sys


def main():
    path = sys.argv[1]
    undef, define = set(), set()
    mode = None
    for a in sys.argv[2:]:
        if a in ("-U", "-D"):
            mode = a
        elif mode == "-U":
            undef.add(a)
        else:
            define.add(a)
    known = undef | define

    def evaluate(expr):
        """-> True / False when the expression only involves known symbols (or literals 0 / 1), else None."""
        e = expr.strip()
        if e in ("0", "1"):
            return e == "1"
        syms = set(re.findall(r"defined\s*\(\s*(\w+)\s*\)|defined\s+(\w+)", e))
        names = {a or b for a, b in syms}
        if not names or not names <= known:
            return None
        py = re.sub(r"defined\s*\(\s*(\w+)\s*\)", lambda m: str(m.group(1) in define), e)
        py = py.replace("&&", " and ").replace("||", " or ").replace("!", " not ")
        if re.search(r"[^\w\s()]", py.replace("True", "").replace("False", "")):
            return None
        return bool(eval(py))

    out = []
    stack = []          # entries: dict(kind='keep'|'resolved', emitting=bool, taken=bool, parent_emit=bool)
    emitting = True
    for line in open(path).read().split("\n"):
        m = re.match(r"^\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)$", line)
        if not m:
            if emitting:
                out.append(line)
            continue
        kw, rest = m.group(1), m.group(2)
        rest_nc = re.sub(r"//.*$", "", rest).strip()
        if kw in ("ifdef", "ifndef", "if"):
            if kw == "ifdef":
                val = (rest_nc in define) if rest_nc in known else None
            elif kw == "ifndef":
                val = (rest_nc not in define) if rest_nc in known else None
            else:
                val = evaluate(rest_nc)
            if val is None:
                stack.append(dict(kind="keep", parent_emit=emitting))
                if emitting:
                    out.append(line)
            else:
                stack.append(dict(kind="resolved", parent_emit=emitting, taken=val))
                emitting = emitting and val
        elif kw == "elif":
            top = stack[-1]
            if top["kind"] == "keep":
                if emitting:
                    out.append(line)
            else:
                raise SystemExit("unifdef: #elif on a resolved conditional is not handled: " + line)
        elif kw == "else":
            top = stack[-1]
            if top["kind"] == "keep":
                if emitting:
                    out.append(line)
            else:
                emitting = top["parent_emit"] and not top["taken"]
        else:
            top = stack.pop()
            if top["kind"] == "keep":
                if emitting:
                    out.append(line)
            else:
                emitting = top["parent_emit"]
    assert not stack
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main()
