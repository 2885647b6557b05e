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
    for p in product_sources() + glob.glob(os.path.join(CSRC, "experiments", "*")):
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


def test_experiments_are_not_in_the_default_build():
    mk = open(os.path.join(ROOT, "mpas-ocean.jl_amd", "Makefile")).read()
    default_objs = mk.split("ifeq ($(VARIANTS),1)")[0]
    assert "stage_variants" not in default_objs
    assert os.path.exists(os.path.join(CSRC, "experiments", "stage_variants.hip"))
    from moka_hip import lib as L
    avail = [v for v in range(12) if L.lib().moka_kernel_variant_available(v)]
    assert set(avail) >= {0, 3, 4, 11}


def test_experimental_kernels_still_compile():
    """csrc/experiments/ (the measured-and-lost execution shapes, kept as a record) includes kernels_common.hpp and StageArgs,
    which keep changing: `make VARIANTS=1` into a build directory of its own must still compile them (hipcc cross-compiles
    gfx950 without a GPU), and the resulting library must offer the variants the default build refuses."""
    import ctypes
    import subprocess
    pkg = os.path.join(ROOT, "mpas-ocean.jl_amd")
    r = subprocess.run(["make", "-C", pkg, "--no-print-directory", "-j4", "VARIANTS=1", "BUILD=build_variants",
                        "SONAME=libmoka_hip_variants.so"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lib = ctypes.CDLL(os.path.join(pkg, "libmoka_hip_variants.so"))
    assert all(lib.moka_kernel_variant_available(v) for v in range(0, 15))
    from moka_hip import lib as L
    assert not L.lib().moka_kernel_variant_available(12)          # the product library stays without them
