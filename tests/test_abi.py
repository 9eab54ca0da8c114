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


def test_no_torch_types_and_plain_c_header():
    text = open(_lib.HEADER).read()
    assert "torch" not in text and "at::" not in text and 'extern "C"' in text


def test_host_entry_points_without_gpu(built):
    from oracle import prng
    assert built.call("magpo_abi_version") == 1
    # chunks of at most 32 tokens by default (csrc/retention32.hpp), 64 on request or for teams of more than 32 agents
    assert built.call("magpo_retention_num_chunks", 128, 4) == 16 and built.call("magpo_retention_num_chunks", 128, 3) == 13
    assert built.call("magpo_retention_num_chunks", 128, 40) == 128
    prev = built.call("magpo_retention_set_chunk_tokens", 64)
    assert prev == 32
    assert built.call("magpo_retention_num_chunks", 128, 4) == 8 and built.call("magpo_retention_num_chunks", 128, 3) == 7
    assert built.call("magpo_retention_set_chunk_tokens", prev) == 64
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
