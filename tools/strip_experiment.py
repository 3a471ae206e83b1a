"""Remove the -DSRGANFD_EXPERIMENT branches from a source file (keeps the product side of every #ifdef / #ifndef SRGANFD_EXPERIMENT block).
    python tools/strip_experiment.py <file> ...   (rewrites in place)"""
import re
import sys


def strip(text: str) -> str:
    out, stack = [], []          # stack entries: None = unrelated conditional, else [kind, in_else]; kind = "ifdef" / "ifndef"
    keep = lambda: all(e is None or ((e[0] == "ifndef") != e[1]) for e in stack)
    for line in text.split("\n"):
        s = line.strip()
        if re.match(r"#\s*ifdef\s+SRGANFD_EXPERIMENT\b", s):
            stack.append(["ifdef", False]); continue
        if re.match(r"#\s*ifndef\s+SRGANFD_EXPERIMENT\b", s):
            stack.append(["ifndef", False]); continue
        if re.match(r"#\s*if", s):
            if keep(): out.append(line)
            stack.append(None); continue
        if re.match(r"#\s*else\b", s) and stack and stack[-1] is not None:
            stack[-1][1] = True; continue
        if re.match(r"#\s*endif\b", s):
            e = stack.pop()
            if e is None and keep(): out.append(line)
            continue
        if keep(): out.append(line)
    assert not stack
    return "\n".join(out)


for f in sys.argv[1:]:
    t = open(f).read()
    open(f, "w").write(strip(t))
