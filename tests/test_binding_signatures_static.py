"""The three statements of the drop-in boundary agree argument by argument (CPU only, nothing is called):

  include/gsi_hip.h            the C prototypes the library is compiled against,
  geostatinversion.jl_amd/_lib the ctypes table every test and bench.py call through,
  julia/*.jl                   the `ccall` tuples of the Julia shim -- which has never been executed (no `julia` here or on
                               the GPU box), so a wrong arity or an Int64 where the library reads a 32-bit int
                               would only show at a maintainer's first call.

Each parameter is reduced to a class {i32, i64, u64, f64, ptr, fnptr}; the three lists must be equal per symbol, and a
`ccall` must pass exactly as many values as its tuple has types.  A crude block balance of the Julia files (openers
against `end`) stands in for the parser that is not here.
"""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JULIA_FILES = [os.path.join(ROOT, "julia", f) for f in ("GeostatInversionHIP.jl", "runtests.jl", "make_reference_fixtures.jl")]


# ---------------------------------------------------------------- include/gsi_hip.h
def _strip_c_comments(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return re.sub(r"//[^\n]*", " ", src)


def _c_class(decl):
    d = decl.strip()
    if d == "void" or d == "":
        return None
    if "*" in d:
        return "ptr"
    if re.search(r"\bgsi_randn_fn\b", d):
        return "fnptr"
    if re.search(r"\buint64_t\b", d):
        return "u64"
    if re.search(r"\bint64_t\b", d):
        return "i64"
    if re.search(r"\bint32_t\b|\bint\b", d):
        return "i32"
    if re.search(r"\bdouble\b", d):
        return "f64"
    raise AssertionError(f"unclassified C parameter: {decl!r}")


def header_prototypes():
    src = _strip_c_comments(open(os.path.join(ROOT, "include", "gsi_hip.h")).read())
    protos = {}
    for m in re.finditer(r"\b(int|const\s+char\s*\*)\s+(gsi_[A-Za-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret = "i32" if m.group(1) == "int" else "ptr"
        params = [p for p in (_c_class(x) for x in m.group(3).split(",")) if p is not None]
        protos[m.group(2)] = (ret, params)
    return protos


# ---------------------------------------------------------------- ctypes table
def _ctypes_class(t):
    if t in (C.c_int, C.c_int32):
        return "i32"
    if t is C.c_int64:
        return "i64"
    if t is C.c_uint64:
        return "u64"
    if t is C.c_double:
        return "f64"
    if t in (C.c_void_p, C.c_char_p) or isinstance(t, type) and issubclass(t, C._Pointer):
        return "ptr"
    if isinstance(t, type) and issubclass(t, C._CFuncPtr):
        return "fnptr"
    raise AssertionError(f"unclassified ctypes parameter: {t!r}")


# ---------------------------------------------------------------- julia ccall tuples
_JL_CLASS = {"Cint": "i32", "Int32": "i32", "Int64": "i64", "Clonglong": "i64", "UInt64": "u64", "Culonglong": "u64",
             "Cdouble": "f64", "Float64": "f64", "Cstring": "ptr"}


def _jl_class(t):
    t = t.strip()
    if t in _JL_CLASS:
        return _JL_CLASS[t]
    if t.startswith("Ptr{") or t.startswith("Ref{"):
        return "ptr"
    raise AssertionError(f"unclassified Julia ccall type: {t!r}")


def _strip_jl(src):
    """Drop comments and string contents (strings may hold parentheses and the word `end`)."""
    out, i, n = [], 0, len(src)
    while i < n:
        if src.startswith('"""', i):
            j = src.find('"""', i + 3)
            j = n if j < 0 else j + 3
            out.append('""' + "\n" * src.count("\n", i, j))
            i = j
        elif src[i] == '"':
            j = i + 1
            while j < n and src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            out.append('""' + "\n" * src.count("\n", i, j))
            i = j + 1
        elif src.startswith("#=", i):
            j = src.find("=#", i)
            i = n if j < 0 else j + 2
        elif src[i] == "#":
            j = src.find("\n", i)
            i = n if j < 0 else j
        elif src[i] == "'" and i + 2 < n and (src[i + 2] == "'" or (src[i + 1] == "\\" and src[i + 3] == "'")) \
                and not (i > 0 and (src[i - 1].isalnum() or src[i - 1] in ")]}_'")):
            i += 3 if src[i + 2] == "'" else 4      # a character literal, not an adjoint
            out.append("' '")
        else:
            out.append(src[i])
            i += 1
    return "".join(out)


def _split_top(s):
    """Split at commas that are not inside (), [] or {}."""
    parts, depth, cur = [], 0, []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append("".join(cur))
            cur = []
        else:
            cur.append(ch)
    parts.append("".join(cur))
    return [p.strip() for p in parts]


def julia_ccalls(path):
    src = _strip_jl(open(path).read())
    calls = []
    for m in re.finditer(r"\bccall\s*\(", src):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        args = _split_top(src[m.end():i - 1])
        line = src.count("\n", 0, m.start()) + 1
        sym = re.match(r"\(\s*:(\w+)\s*,\s*\w+\s*\)", args[0])
        assert sym, f"{path}:{line}: ccall target {args[0]!r}"
        tup = args[2]
        assert tup.startswith("(") and tup.endswith(")"), f"{path}:{line}: argument tuple {tup!r}"
        types = [t for t in _split_top(tup[1:-1]) if t]
        calls.append({"where": f"{os.path.basename(path)}:{line}", "name": sym.group(1), "ret": args[1],
                      "types": types, "nvalues": len(args) - 3})
    return calls


# ---------------------------------------------------------------- tests
def test_header_parses_every_symbol():
    from test_cabi_symbols import header_symbols
    assert sorted(header_prototypes()) == header_symbols()


def test_ctypes_table_matches_header(gsi):
    protos = header_prototypes()
    bad = []
    for name, (res, args) in gsi._lib.SIGNATURES.items():
        ret = "ptr" if res is C.c_char_p else _ctypes_class(res)
        mine = (ret, [_ctypes_class(a) for a in args])
        if mine != protos[name]:
            bad.append(f"{name}: ctypes {mine} != header {protos[name]}")
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("path", JULIA_FILES, ids=[os.path.basename(p) for p in JULIA_FILES])
def test_julia_ccalls_match_header(path):
    protos = header_prototypes()
    bad = []
    for c in julia_ccalls(path):
        if c["name"] not in protos:
            bad.append(f"{c['where']}: {c['name']} is not in include/gsi_hip.h")
            continue
        ret, params = protos[c["name"]]
        jl = [_jl_class(t) for t in c["types"]]
        # a Julia callback goes over as a Ptr{Cvoid} made by @cfunction
        want = ["ptr" if p == "fnptr" else p for p in params]
        if jl != want:
            bad.append(f"{c['where']}: {c['name']} ccall types {jl} != header {want}")
        if c["nvalues"] != len(c["types"]):
            bad.append(f"{c['where']}: {c['name']} passes {c['nvalues']} values for {len(c['types'])} types")
        if _jl_class(c["ret"]) != ret:
            bad.append(f"{c['where']}: {c['name']} return {c['ret']} != header {ret}")
    assert not bad, "\n".join(bad)


def test_julia_shim_binds_the_hot_path_entry_points():
    """What getxis / randsvd / rangefinder / the LowRankCovMatrix products reach is bound in the shim."""
    bound = {c["name"] for c in julia_ccalls(JULIA_FILES[0])}
    need = {"gsi_ctx_create", "gsi_ctx_destroy", "gsi_op_dense", "gsi_op_lowrank", "gsi_op_destroy", "gsi_op_mul",
            "gsi_rangefinder", "gsi_randsvd", "gsi_randsvd_dense_host", "gsi_rangefinder_dense_host", "gsi_last_error",
            "gsi_eig_nystrom", "gsi_rangefinder_adaptive"}
    assert need <= bound, sorted(need - bound)


_OPENERS = re.compile(r"(?<![\w.:])(function|if|for|while|let|begin|try|do|struct|module|macro|quote|baremodule)\b")


@pytest.mark.parametrize("path", JULIA_FILES, ids=[os.path.basename(p) for p in JULIA_FILES])
def test_julia_blocks_balance(path):
    """Every block opener has its `end`, brackets close, and `end` inside [...] (an index) is not counted."""
    src = _strip_jl(open(path).read())
    depth_sq = 0
    opens = ends = 0
    stack = []
    for tok in re.finditer(r"[\[\](){}]|(?<![\w.:!])(?:mutable\s+struct|function|if|for|while|let|begin|try|struct|module|"
                           r"macro|quote|baremodule|end)\b|\bdo\b", src):
        t = tok.group(0)
        line = src.count("\n", 0, tok.start()) + 1
        if t in "([{":
            stack.append((t, line))
            depth_sq += t == "["
        elif t in ")]}":
            assert stack, f"{path}:{line}: unmatched {t}"
            o, ol = stack.pop()
            assert {"(": ")", "[": "]", "{": "}"}[o] == t, f"{path}:{line}: {t} closes {o} of line {ol}"
            depth_sq -= t == "]"
        elif t == "end":
            if depth_sq == 0:
                ends += 1
        elif t in ("for", "if") and stack:
            pass                                   # a generator / comprehension / ternary-free filter inside brackets
        else:
            opens += 1
    assert not stack, f"{path}: unclosed {stack[-1][0]} of line {stack[-1][1]}"
    assert opens == ends, f"{path}: {opens} block openers, {ends} `end`s"


def test_default_context_is_created_once():
    """`something(default_ctx[], (default_ctx[] = Context(0)))` built a new GPU context at EVERY `ctx()` call (a function's
    arguments are evaluated before it runs): operators and matrices of one session ended up on different contexts."""
    src = _strip_jl(open(JULIA_FILES[0]).read())
    assert not re.search(r"something\s*\([^)]*Context\s*\(", src)
    body = re.search(r"function ctx\(\)(.*?)\nend", src, flags=re.S)
    assert body and "=== nothing" in body.group(1) and body.group(1).count("Context(0)") == 1
