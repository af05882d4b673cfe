"""The C-ABI library loads and exports every symbol include/*.h declares (no compute calls:
runs without a GPU), error paths that need no device behave, and the product package never reaches
for the oracle."""
import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    syms = set()
    for header, prefix in (("mbgc_swsem.h", "swsem_"), ("mbgc_fasta.h", "mbgc_fasta_"), ("mbgc_copmem.h", "mbgc_copmem_")):
        src = open(os.path.join(ROOT, "include", header)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        syms |= set(re.findall(r"\b(%s[A-Za-z0-9_]+)\s*\(" % prefix, src))
    return sorted(syms)


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    lib = ctypes.CDLL(os.path.join(ROOT, "mbgc_amd", "libmbgc_hip.so"))
    syms = declared_symbols()
    assert len(syms) >= 35
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    from mbgc_amd import binding
    from mbgc_amd import copmem, fasta
    assert set(binding.EXPORTS) <= set(syms) and set(fasta.EXPORTS) <= set(syms) and set(copmem.EXPORTS) <= set(syms)


def test_exchange_library_exports_every_declared_symbol():
    """include/mbgc_exchange.h (the C++ host's exchange between the GPUs of a node) against libmbgc_xchg.so"""
    import __graft_entry__ as g
    g.build()
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mbgc_exchange.h")).read(), flags=re.S)
    syms = sorted(set(re.findall(r"\b(mbgc_xchg_[A-Za-z0-9_]+)\s*\(", src)))
    assert len(syms) >= 14, syms
    lib = ctypes.CDLL(os.path.join(ROOT, "mbgc_amd", "libmbgc_xchg.so"))
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    lib.mbgc_xchg_hostmem_min_bytes.restype = ctypes.c_uint64
    assert lib.mbgc_xchg_hostmem_min_bytes(2) >= 64
    x = ctypes.c_void_p()
    assert lib.mbgc_xchg_create_hostmem(ctypes.byref(x), None, 0, 0, 2, 0) != 0          # no mapping: refused before any device call


def test_no_device_means_loud_failure_not_fallback():
    from mbgc_amd import binding
    L = binding.lib()
    if L.swsem_device_count() > 0:
        return      # on the GPU box this is covered by the gpu tests
    try:
        binding.SlidingWindowSparseEMMatcher(1 << 20)
    except binding.SwsemError as e:
        assert "no HIP device" in str(e) or "HIP" in str(e)
    else:
        raise AssertionError("matcher creation must fail without a GPU")


def test_parameter_validation_needs_no_device():
    from mbgc_amd import binding
    L = binding.lib()
    h = ctypes.c_void_p()
    assert L.swsem_create(ctypes.byref(h), 1 << 20, 32, 0, 1, 16, 0) == -1       # k1 must be positive (an odd one is the identity-encoding variant)
    assert b"sampling step" in L.swsem_last_error()
    assert L.swsem_create(ctypes.byref(h), 1 << 20, 32, 16, 2, 16, 0) == -1      # k2 != 1
    assert L.swsem_create(ctypes.byref(h), 1 << 20, 8, 16, 1, 16, 0) == -1       # L too short
    p = binding.emit_params(0)
    assert p.frugal64bitLenEncoding == 0 and p.unmatchedFractionFactorTweakForDissimilarContigs == 32
    p = binding.emit_params(2)
    assert p.allowedTargetsOutrunForDissimilarContigs == 0


def test_product_never_imports_the_oracle():
    for base, _, files in os.walk(os.path.join(ROOT, "mbgc_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "_orc" not in text and "liboracle" not in text, (base, f)
