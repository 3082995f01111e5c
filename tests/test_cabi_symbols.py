"""The C-ABI library loads on a box without a GPU and exports every symbol include/lightglue_amd.h
declares (no compute calls here)."""
import ctypes
import re
from pathlib import Path

import pytest

from lightglue_amd import _cabi

ROOT = Path(__file__).resolve().parent.parent
HEADER = (ROOT / "include" / "lightglue_amd.h").read_text()


def declared_functions():
    # prototypes look like:  <ret> lg_xxx(...);
    return sorted(set(re.findall(r"\b(lg_[a-z_]+)\s*\(", HEADER)))


def test_library_is_built():
    assert _cabi.library_path().exists(), "run __graft_entry__.build() first"


def test_exports_every_declared_symbol():
    lib = _cabi.load()
    names = declared_functions()
    assert set(names) == set(_cabi.EXPORTED_SYMBOLS), (names, _cabi.EXPORTED_SYMBOLS)
    for n in names:
        assert getattr(lib, n) is not None


def test_struct_layouts_match_header():
    # field order of the ctypes mirrors == order of the declarations in the header
    def fields_of(struct_name):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct_name, struct_name), HEADER, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = decl.split(None, 1)[1] if not decl.startswith("const") else decl.split(None, 2)[2]
            out += [n.strip().lstrip("*") for n in names.split(",")]
        return out
    assert fields_of("lg_config") == [f[0] for f in _cabi.LgConfig._fields_]
    assert fields_of("lg_forward_io") == [f[0] for f in _cabi.LgForwardIO._fields_]
    assert fields_of("lg_unpack_io") == [f[0] for f in _cabi.LgUnpackIO._fields_]
    # the wire width of the Python side == LG_WIRE_WIDTH of the header
    macro = re.search(r"#define LG_WIRE_WIDTH\(n0, n1\) (.*?)\s+/\*", HEADER).group(1)
    assert eval(macro.replace("LL", ""), {"n0": 300, "n1": 77}) == _cabi.wire_width(300, 77)


def test_envelope_is_refused_without_touching_a_gpu():
    """LG_MAX_KEYPOINTS / LG_MAX_ROWS / LG_MAX_SIM_ELEMS (include/lightglue_amd.h): a call outside the envelope returns LG_ERR_INVALID with a message before any
    allocation or launch (4.3 M rows must be an error, not silently dropped stores)."""
    lib = _cabi.load()
    limits = {k: int(re.search(r"#define %s (\d+)" % k, HEADER).group(1)) for k in ("LG_MAX_KEYPOINTS", "LG_MAX_ROWS", "LG_MAX_SIM_ELEMS")}
    assert limits == {"LG_MAX_KEYPOINTS": 8192, "LG_MAX_ROWS": 2 ** 21, "LG_MAX_SIM_ELEMS": 2 ** 31 - 1}
    h = ctypes.c_void_p()
    ok = _cabi.LgConfig(256, 256, 9, 4, 0, 0.95, 0.99, 0.1, -1, 4, -1)
    assert lib.lg_engine_create(ctypes.byref(ok), ctypes.byref(h)) == _cabi.LG_OK
    for (B, n0, n1, word) in ((1, 8193, 100, b"LG_MAX_KEYPOINTS"), (1050, 2048, 2048, b"LG_MAX_ROWS"), (129, 4096, 4096, b"LG_MAX_SIM_ELEMS"), (513, 2048, 2048, b"LG_MAX_ROWS")):
        assert lib.lg_engine_reserve(h, B, n0, n1) == _cabi.LG_ERR_INVALID, (B, n0, n1)
        assert word in lib.lg_last_error(), lib.lg_last_error()
    lib.lg_engine_destroy(h)
    # the receiving side of the wire: bad strides / counts are refused, an empty gather is a no-op
    io = _cabi.LgUnpackIO()
    io.rows, io.n0, io.n1, io.pairs_out = 0, 16, 16, 0
    assert lib.lg_unpack_wire(ctypes.byref(io), None) == _cabi.LG_OK
    io.rows, io.pairs_out, io.wire, io.wire_stride = 2, 2, 1 << 20, _cabi.wire_width(16, 16) - 1
    assert lib.lg_unpack_wire(ctypes.byref(io), None) == _cabi.LG_ERR_INVALID


def test_argument_validation_without_gpu():
    lib = _cabi.load()
    assert lib.lg_version().decode().startswith("lightglue_amd")
    h = ctypes.c_void_p()
    bad = _cabi.LgConfig(256, 128, 9, 4, 0, 0.95, 0.99, 0.1, -1, 4, -1)  # descriptor_dim != 256
    assert lib.lg_engine_create(ctypes.byref(bad), ctypes.byref(h)) == _cabi.LG_ERR_INVALID
    assert b"descriptor_dim" in lib.lg_last_error()
    with pytest.raises(AssertionError):
        _cabi.check(_cabi.LG_ERR_INVALID)
    ok = _cabi.LgConfig(256, 256, 9, 4, 0, 0.95, 0.99, 0.1, -1, 4, -1)
    assert lib.lg_engine_create(ctypes.byref(ok), ctypes.byref(h)) == _cabi.LG_OK
    # forward before weights -> state error, no GPU touched
    io = _cabi.LgForwardIO()
    io.batch, io.n0, io.n1 = 1, 4, 4
    assert lib.lg_engine_forward(h, ctypes.byref(io), None) == _cabi.LG_ERR_STATE
    # options: unknown keys (and the switches of removed experiments) are refused, with a message
    assert lib.lg_engine_set_option(h, b"fused_next", 0) == _cabi.LG_OK
    assert lib.lg_engine_set_option(h, b"no_such_option", 1) == _cabi.LG_ERR_INVALID and b"no_such_option" in lib.lg_last_error()
    assert lib.lg_engine_set_option(h, b"tail_variant", 1) == _cabi.LG_ERR_INVALID
    lib.lg_engine_destroy(h)
    # the split-f16 conv stack addresses one image with 32-bit element offsets: larger images are refused before anything is touched
    fake = ctypes.c_void_p(16)
    arr = (ctypes.c_void_p * 24)(*[16] * 24)
    assert lib.lg_sp_encode_split(fake, 1, 8192, 4096, arr, fake, 1 << 40, fake, fake, None) == _cabi.LG_ERR_INVALID and b"2^25" in lib.lg_last_error()
    assert lib.lg_sp_pack_conv_weight_split(fake, 65, 64, 3, fake, None) == _cabi.LG_ERR_INVALID       # 3 x 3 layers: cout % 64
    assert lib.lg_sp_pack_conv_weight_split(fake, 64, 48, 1, fake, None) == _cabi.LG_ERR_INVALID       # cin % 32
    # precision enum: every named mode is accepted; the removed split-bf16 value (3) and anything past the last one are not
    for name, val in _cabi.LG_PREC.items():
        cfg = _cabi.LgConfig(256, 256, 9, 4, 0, 0.95, 0.99, 0.1, -1, val, -1)
        assert lib.lg_engine_create(ctypes.byref(cfg), ctypes.byref(h)) == _cabi.LG_OK, name
        lib.lg_engine_destroy(h)
    for val in (3, max(_cabi.LG_PREC.values()) + 1):
        cfg = _cabi.LgConfig(256, 256, 9, 4, 0, 0.95, 0.99, 0.1, -1, val, -1)
        assert lib.lg_engine_create(ctypes.byref(cfg), ctypes.byref(h)) == _cabi.LG_ERR_INVALID and b"precision" in lib.lg_last_error()
    # attention precision: the linear precision itself, or one f16 plane together with f16x3 (the fast opt-in); nothing else
    P = _cabi.LG_PREC
    for prec, attn, want in ((P["f16x3"], P["f16x3"], _cabi.LG_OK), (P["f16x3"], P["fp16"], _cabi.LG_OK), (P["f16x3"], P["bf16"], _cabi.LG_ERR_INVALID),
                             (P["fp32"], P["fp16"], _cabi.LG_ERR_INVALID), (P["bf16"], P["bf16"], _cabi.LG_OK)):
        cfg = _cabi.LgConfig(256, 256, 9, 4, 0, 0.95, 0.99, 0.1, -1, prec, attn)
        assert lib.lg_engine_create(ctypes.byref(cfg), ctypes.byref(h)) == want, (prec, attn)
        if want == _cabi.LG_OK:
            lib.lg_engine_destroy(h)
