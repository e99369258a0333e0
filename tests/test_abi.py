"""CPU-side checks of the C ABI: the library loads, exports every symbol include/fh_hip.h declares, and the
ctypes structures match the header's layout.  No compute calls (no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "fh_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|int64_t)\s+(fh_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from free_hunch_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/fh_hip.h but not exported"
    assert sorted(_lib.exported_symbols()) == names, "ctypes prototypes out of sync with the header"
    assert lib.fh_version() >= 100


def test_struct_layout_matches_header():
    from free_hunch_amd import _lib
    assert ctypes.sizeof(_lib.FhProblem) == 8 * 4 + 8 + 8 + 8 * 8 + 16 + 3 * 8 + 4 * 8 and _lib.FhProblem.fold_fwd_w.offset == 152
    assert _lib.FhProblem.d.offset == 32 and _lib.FhProblem.sigma_y2.offset == 40
    assert _lib.FhProblem.tap_dy.offset == 48 and _lib.FhProblem.M.offset == 104
    # struct fh_cov_state: int64 d, 6 x int32, 3 x (4 pointers), 7 pointers
    assert ctypes.sizeof(_lib.FhCovState) == 8 + 6 * 4 + 12 * 8 + 7 * 8 and _lib.FhCovState.D.offset == 32
    assert ctypes.sizeof(_lib.FhCgInfo) == 24


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from free_hunch_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _lib.load()
    except _lib.FhError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("loading a missing library must raise")


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "free-hunch_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py") and fn != "smoke.py":
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("# oracle", ""), f"{fn} mentions the oracle"
