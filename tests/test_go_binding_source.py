"""go/ipx cannot be compiled here (no Go toolchain in the image), so at least its contact surface with the C ABI is checked as text:
every C.ipx_* function, C.IPX_* constant and C.ipx_* type the Go files name must be declared in include/ipx.h, and a call must pass
as many arguments as the prototype has parameters."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header():
    return open(os.path.join(ROOT, "include", "ipx.h")).read()


def _go_sources():
    d = os.path.join(ROOT, "go", "ipx")
    return {f: open(os.path.join(d, f)).read() for f in sorted(os.listdir(d)) if f.endswith(".go")}


def _prototypes(h):
    """name -> number of parameters, for every function include/ipx.h declares"""
    text = re.sub(r"/\*.*?\*/", " ", h, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(ipx_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        params = m.group(2).strip()
        out[m.group(1)] = 0 if params in ("", "void") else params.count(",") + 1
    return out


def _call_args(src, start):
    """number of top-level arguments of the call whose '(' is at src[start]"""
    depth, n, i, seen = 0, 0, start, False
    while i < len(src):
        ch = src[i]
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
            if depth == 0:
                return n + (1 if seen else 0)
        elif ch == "," and depth == 1:
            n += 1
        elif depth >= 1 and not ch.isspace():
            seen = True
        i += 1
    raise AssertionError("unbalanced call")


def test_every_c_name_the_go_files_use_exists_in_the_header():
    h = _header()
    protos = _prototypes(h)
    declared = set(re.findall(r"\b(ipx_\w+|IPX_\w+)\b", re.sub(r"/\*.*?\*/", " ", h, flags=re.S)))
    missing = []
    for f, src in _go_sources().items():
        for name in set(re.findall(r"\bC\.((?:ipx|IPX)_\w+)", src)):
            if name not in declared:
                missing.append("%s: C.%s" % (f, name))
    assert not missing, "names that include/ipx.h does not declare: " + ", ".join(sorted(missing))
    assert len(protos) > 60                      # the parser found the header's functions


def test_calls_pass_the_prototypes_number_of_arguments():
    protos = _prototypes(_header())
    bad = []
    for f, src in _go_sources().items():
        for m in re.finditer(r"\bC\.(ipx_\w+)\(", src):
            name = m.group(1)
            if name not in protos:
                continue                              # a type conversion such as C.ipx_ticket(x)
            got = _call_args(src, m.end() - 1)
            if got != protos[name]:
                bad.append("%s: C.%s called with %d arguments, the prototype has %d" % (f, name, got, protos[name]))
    assert not bad, "; ".join(bad)
