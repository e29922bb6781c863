"""The C-ABI library loads without a GPU and exports every symbol include/seevcn_hip.h declares."""
import re

import seevcn_amd._lib as L


def _declared_symbols():
    src = open(L.HEADER_PATH).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sv_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(hip_lib):
    names = _declared_symbols()
    assert len(names) >= 5
    for n in names:
        assert hasattr(hip_lib, n), f"{n} declared in seevcn_hip.h but not exported"


def test_binding_table_matches_header(hip_lib):
    assert sorted(L.SIGNATURES) == _declared_symbols()


def _declared_prototypes():
    """name -> (return type, [parameter types]) parsed from the header's C declarations"""
    src = open(L.HEADER_PATH).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for ret, name, params in re.findall(r"\b(int64_t|size_t|int|const char\*)\s+(sv_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        plist = [] if params.strip() in ("", "void") else [" ".join(p.split()) for p in params.split(",")]
        protos[name] = (ret, plist)
    return protos


def test_binding_signatures_match_header_prototypes(hip_lib):
    """ctypes argtypes / restype of every binding against the header: parameter count, pointer vs int vs int64 vs float vs double."""
    import ctypes
    protos = _declared_prototypes()
    assert sorted(protos) == _declared_symbols()

    def kind(ctype_decl):
        t = ctype_decl.rsplit(" ", 1)[0] if not ctype_decl.endswith("*") else ctype_decl
        if "*" in ctype_decl:
            return ctypes.c_void_p
        return {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "size_t": ctypes.c_size_t, "float": ctypes.c_float, "double": ctypes.c_double}[t.replace("const ", "")]

    for name, (ret, params) in protos.items():
        restype, argtypes = L.SIGNATURES[name]
        want_ret = {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "size_t": ctypes.c_size_t, "const char*": ctypes.c_char_p}[ret]
        assert restype == want_ret, (name, restype, ret)
        assert len(argtypes) == len(params), (name, len(argtypes), params)
        for a, p in zip(argtypes, params):
            assert a == kind(p), (name, p, a)


def test_abi_version_and_sizes(hip_lib):
    assert hip_lib.sv_abi_version() == 1
    # 1024 cells = one chunk: 32 words * 8 B + 2 ints
    assert hip_lib.sv_index_persistent_bytes(1024) == 32 * 8 + 8
    assert hip_lib.sv_index_persistent_bytes(1025) == 2 * (32 * 8 + 8)
    assert hip_lib.sv_voxelize_dynamic_scratch_bytes(1000, 1 << 20, 1000) >= 1000 * 12


def test_ops_refuse_cpu_tensors(hip_lib):
    import pytest
    import torch
    from seevcn_amd.pcdet.ops import voxel_ops
    with pytest.raises(L.SeevcnHipError):
        voxel_ops.voxelize_dynamic(torch.zeros(4, 4), [0, 0, 0, 1, 1, 1], [1, 1, 1], [1, 1, 1], 1)
