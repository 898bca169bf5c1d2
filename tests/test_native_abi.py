"""CPU-side checks of the C-ABI library: it loads, and exports every symbol include/oasr.h declares
(no compute calls -- there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    import __graft_entry__ as g
    from olmoasr_amd import _native
    if not os.path.isfile(_native.LIB_PATH):
        g.build()
    return _native


def test_library_exports_every_declared_symbol(native):
    lib = native.lib()
    hdr = open(os.path.join(ROOT, "include", "oasr.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(oasr_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 25
    for n in sorted(names):
        assert hasattr(lib, n), f"{n} declared in include/oasr.h but not exported"
    assert set(native.EXPORTS) <= names | {"oasr_last_error"}
    assert lib.oasr_version() == 100


def test_param_table_matches_reference_state_dict(native):
    """The engine's parameter table must carry exactly the reference's state_dict names/shapes."""
    from oracle import model_oracle as mo
    lib = native.lib()
    for variant in ("tiny", "base"):
        dims = mo.VARIANTS[variant]
        cd = native.Dims(*[getattr(dims, f[0]) for f in native.Dims._fields_])
        ctx = lib.oasr_create(ctypes.byref(cd))
        assert ctx
        sd = mo.init_state_dict(dims, seed=0)
        n = lib.oasr_param_count(ctx)
        got = {}
        end = 0
        for i in range(n):
            name = ctypes.create_string_buffer(128)
            off, numel, ndim = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
            shape = (ctypes.c_int64 * 4)()
            assert lib.oasr_param_info(ctx, i, name, 128, ctypes.byref(off), ctypes.byref(numel), ctypes.byref(ndim), shape) == 0
            assert off.value == end and off.value % 8 == 0
            end += numel.value
            got[name.value.decode()] = tuple(shape[j] for j in range(ndim.value))
        assert end == lib.oasr_param_numel(ctx) and end % 8 == 0
        want = {k: tuple(v.shape) for k, v in sd.items() if k != "encoder.positional_embedding"}
        assert got == want
        # gradient segments tile the arena exactly once
        segs = []
        for i in range(lib.oasr_segment_count(ctx)):
            o, m = ctypes.c_int64(), ctypes.c_int64()
            assert lib.oasr_segment_info(ctx, i, ctypes.byref(o), ctypes.byref(m)) == 0
            segs.append((o.value, m.value))
        pos = 0
        for o, m in sorted(segs):
            assert o == pos
            pos += m
        assert pos == end
        assert lib.oasr_workspace_bytes(ctx, 2, 448, 1) > lib.oasr_workspace_bytes(ctx, 2, 448, 0) > 0
        lib.oasr_destroy(ctx)


def test_null_context_is_rejected_loudly(native):
    lib = native.lib()
    assert lib.oasr_create(None) is None
    assert b"null dims" in lib.oasr_last_error()
    with pytest.raises(native.NativeError):
        native.require_gpu(__import__("torch").zeros(1), "x")
