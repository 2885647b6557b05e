"""Static rules of the product sources (no GPU needed).

* No null-stream hipMemcpy / hipMemset in the library: its streams are non-blocking, so a null-stream copy is not ordered
  against work queued on them (the tape-list race of round 1, commit 6ebfa3e).  Every copy goes through the context's
  stream (h2d / hipMemcpyAsync / hipMemsetAsync).
* The product never touches the oracle (test infrastructure) and has no environment-variable debug hooks.
* The experimental stage kernels stay out of the default build.
"""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mpas-ocean.jl_amd", "csrc")


def _code(path):
    """File contents without // comments (the rule above is quoted in one)."""
    return "\n".join(line.split("//", 1)[0] for line in open(path).read().split("\n"))


def product_sources():
    return [p for p in glob.glob(os.path.join(CSRC, "*")) if os.path.isfile(p)]


def test_no_null_stream_copies_in_the_library():
    bad = []
    for p in product_sources():
        for m in re.finditer(r"\bhipMem(cpy|set|cpyDtoH|cpyHtoD|cpyDtoD|cpyPeer)\s*\(", _code(p)):
            bad.append((os.path.basename(p), m.group(0)))
    assert not bad, bad


def test_no_debug_environment_hooks_and_no_oracle_in_the_product():
    for p in product_sources():
        src = _code(p)
        assert "getenv" not in src, p
        assert "oracle" not in src.lower(), p
    for p in glob.glob(os.path.join(ROOT, "mpas-ocean.jl_amd", "moka_hip", "*.py")):
        src = open(p).read()
        assert not re.search(r"^\s*(import|from)\s+oracle\b", src, re.M), p


def test_only_the_product_kernels_are_built():
    """The execution shapes that were measured and lost in rounds 1-3 (csrc/experiments) are gone; their record is
    profiles/r0*_variants.txt.  The library offers the default stage kernel (11), its two fallbacks (4, 3) and auto (0)."""
    assert not os.path.exists(os.path.join(CSRC, "experiments"))
    from moka_hip import lib as L
    avail = [v for v in range(15) if L.lib().moka_kernel_variant_available(v)]
    assert avail == [0, 3, 4, 11]


def test_every_symbol_the_julia_shim_calls_is_declared_in_the_header():
    """julia/*.jl is never executed in this pipeline (no Julia in the image): at least every `ccall((:moka_..., lib)` names an
    entry point include/moka_hip.h declares (and the library exports: tests/test_plan_host.py checks header == exports)."""
    import glob
    import re
    header = open(os.path.join(ROOT, "include", "moka_hip.h")).read()
    declared = set(re.findall(r"\b(moka_[a-z0-9_]+)\s*\(", header))
    files = glob.glob(os.path.join(ROOT, "mpas-ocean.jl_amd", "julia", "*.jl"))
    assert files
    called = set()
    for f in files:
        called |= set(re.findall(r"ccall\(\(:(moka_[a-z0-9_]+)\s*,", open(f).read()))
    assert len(called) >= 30 and not (called - declared), sorted(called - declared)
    # ... with as many argument types in the ccall's tuple as the prototype has parameters
    plain = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    nargs = {}
    for m in re.finditer(r"\b(moka_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", plain, flags=re.S):
        a = m.group(2).strip()
        nargs[m.group(1)] = 0 if a in ("", "void") else len(a.split(","))
    bad = []
    for f in files:
        t = open(f).read()
        for m in re.finditer(r"ccall\(\(:(moka_[a-z0-9_]+)\s*,\s*lib\)\s*,\s*[A-Za-z{}\.]+\s*,\s*\(", t):
            i, depth = m.end(), 1
            j = i
            while depth:
                depth += {"(": 1, ")": -1}.get(t[j], 0)
                j += 1
            parts, d, cur = [], 0, ""
            for c in t[i:j - 1]:
                d += {"(": 1, "{": 1, "[": 1, ")": -1, "}": -1, "]": -1}.get(c, 0)
                if c == "," and d == 0:
                    parts.append(cur)
                    cur = ""
                else:
                    cur += c
            parts.append(cur)
            n = len([q for q in parts if q.strip()])
            if nargs.get(m.group(1)) != n:
                bad.append((os.path.basename(f), m.group(1), n, nargs.get(m.group(1))))
    assert not bad, bad

