"""Regenerate the appendix of INTEGRATION.md from include/nmrf_hip.h: one row per exported entry point -- what it is (first line of its
comment) and the reference lines that comment cites.  python tools/gen_abi_index.py [--check]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- abi-index:begin -->", "<!-- abi-index:end -->"
CITE = re.compile(r"(?:[\w./]+/)?[\w]+\.(?:py|cuh|cu|h|cpp):\d+(?:-\d+)?(?:\s*,\s*\d+(?:-\d+)?)*")


def entries():
    src = open(os.path.join(ROOT, "include", "nmrf_hip.h")).read()
    out = []
    for m in re.finditer(r"^(?:int|const char \*)\s*(nmrf_[a-z0-9_]+)\s*\(", src, re.M):
        name = m.group(1)
        # the comment block that ends closest before the prototype (entry points sharing one comment all cite it)
        head = src[:m.start()]
        ce = head.rfind("*/")
        cs = head.rfind("/*", 0, ce)
        comment = head[cs + 2:ce] if cs >= 0 else ""
        lines = [re.sub(r"^\s*\*\s?", "", l).strip() for l in comment.strip().splitlines()]
        first = next((l for l in lines if l), "")
        whole = re.sub(r"\s+", " ", " ".join(lines))
        own = re.search(re.escape(name) + r"(?: builds| is)?[:] ?(.*)", whole)      # a shared comment block: the sentence that names this one
        if own:
            first = own.group(1)
        if name == "nmrf_strerror":
            first = "Text of an NMRF_E* status code (every entry point returns 0 or one of them; nothing is thrown across the boundary)."
        first = re.sub(r"\s+", " ", re.sub(r"^-+\s*|\s*-+$", "", first))
        if len(first) > 150:
            first = first[:147].rsplit(" ", 1)[0] + " ..."
        cites = []
        for c in CITE.findall(" ".join(lines)):
            c = re.sub(r"\s+", "", c)
            if c not in cites and not c.startswith(("nmrf_hip", "split_mfma", "common")):
                cites.append(c)
        out.append((name, first.replace("|", "\\|"), ", ".join("`%s`" % c for c in cites[:4]) or "–"))
    return out


def table():
    rows = ["| entry point | what (first line of its comment in `include/nmrf_hip.h`) | reference lines cited there |", "|---|---|---|"]
    rows += ["| `%s` | %s | %s |" % e for e in entries()]
    return "\n".join(rows)


if __name__ == "__main__":
    path = os.path.join(ROOT, "INTEGRATION.md")
    doc = open(path).read()
    a, b = doc.index(BEGIN) + len(BEGIN), doc.index(END)
    new = doc[:a] + "\n" + table() + "\n" + doc[b:]
    if "--check" in sys.argv:
        sys.exit(0 if new == doc else "INTEGRATION.md appendix is stale: python tools/gen_abi_index.py")
    open(path, "w").write(new)
    print("%d entry points" % len(entries()))
