"""The C-ABI library builds for gfx950, loads without a GPU and exports every symbol include/magpo.h declares."""
import ctypes
import os

import numpy as np
import pytest

from magpo_amd import _lib


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    return _lib.lib()


def test_header_parses_and_every_symbol_is_exported(built):
    protos = _lib.parse_header()
    assert len(protos) >= 40
    dll = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in protos if not hasattr(dll, n)]
    assert not missing, f"declared in include/magpo.h but not exported: {missing}"
    for name, (ret, params) in protos.items():
        assert ret in ("int", "long", "const char*")
        for t, _ in params:
            assert "*" in t or t.replace("const", "").strip() in ("int", "long", "float", "uint32_t", "magpo_stream_t"), (name, t)


def test_no_setters_and_no_environment_reads(built):
    """no global mutable state behind the C ABI.  Every tuning knob is a per-call argument (include/magpo.h), so the
    library exports no setter, does not import getenv, and the exported symbols are exactly the header's (+ nothing hidden)."""
    import subprocess
    protos = _lib.parse_header()
    assert not [n for n in protos if "_set_" in n or n.startswith("magpo_set")], "setter declared in include/magpo.h"
    dyn = subprocess.run(["nm", "-D", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout.splitlines()
    exported = {ln.split()[-1] for ln in dyn if " T " in ln and ln.split()[-1].startswith("magpo_")}
    assert not [n for n in exported if "_set_" in n], exported
    assert exported == set(protos), (exported - set(protos), set(protos) - exported)
    undefined = {ln.split()[-1].split("@")[0] for ln in dyn if " U " in ln}
    assert "getenv" not in undefined and "secure_getenv" not in undefined and "setenv" not in undefined
    # per-call knobs are validated, not remembered
    with pytest.raises(ValueError):
        built.call("magpo_retention_chunk_fwd", None, 64, None, 64, None, 64, None, 64, None, None, None, None, None, 1, 8, 4, 1, 0.5, 64, None, 48, None)


def test_no_torch_types_and_plain_c_header():
    text = open(_lib.HEADER).read()
    assert "torch" not in text and "at::" not in text and 'extern "C"' in text


def test_host_entry_points_without_gpu(built):
    from oracle import prng
    assert built.call("magpo_abi_version") == 3
    # chunks of at most 32 tokens by default (csrc/retention32.hpp), 64 on request or for teams of more than 32 agents
    assert built.call("magpo_retention_num_chunks", 128, 4, 0) == 16 and built.call("magpo_retention_num_chunks", 128, 3, 32) == 13
    assert built.call("magpo_retention_num_chunks", 128, 40, 0) == 128
    # the chunk size is a per-call argument: nothing carries over between calls
    assert built.call("magpo_retention_num_chunks", 128, 4, 64) == 8 and built.call("magpo_retention_num_chunks", 128, 3, 64) == 7
    assert built.call("magpo_retention_num_chunks", 128, 4, 0) == 16
    key = prng.prng_key(99)
    out = np.zeros((5, 2), np.uint32)
    built.raw("magpo_key_split_host")(key.ctypes.data, 5, out.ctypes.data)
    assert np.array_equal(out, prng.split(key, 5))
    bits = np.zeros(9, np.uint32)
    built.raw("magpo_random_bits_host")(key.ctypes.data, 9, bits.ctypes.data)
    assert np.array_equal(bits, prng.random_bits(key, 9))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        _lib._Lib()


def test_product_does_not_import_the_oracle():
    import pathlib
    root = pathlib.Path(_lib.__file__).parent
    for f in root.rglob("*.py"):
        src = f.read_text()
        assert "import oracle" not in src and "from oracle" not in src, f
