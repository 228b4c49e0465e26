"""The C-ABI library loads on a GPU-less host and exports every symbol include/sva.h declares
(no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "sva.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sva_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported():
    from streamvoiceanon_amd import engine as E

    lib = ctypes.CDLL(E.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sva.h but not exported by libsva_hip.so"
    assert sorted(E.EXPORTED_SYMBOLS) == names


def test_defaults_match_reference_yaml():
    from streamvoiceanon_amd import engine as E

    lib = E.load_library()
    cfg = E.SvaConfig()
    assert lib.sva_config_default(ctypes.byref(cfg)) == 0
    assert list(cfg.enc_dims) == [128, 256, 384, 512] and list(cfg.enc_depths) == [3, 3, 9, 3]
    assert (cfg.ar_dim, cfg.ar_layers, cfg.ar_fast_layers, cfg.ar_vocab, cfg.codebook_size) == (768, 12, 4, 8192, 1000)
    p = E.SvaStreamParams()
    assert lib.sva_stream_params_default(ctypes.byref(p)) == 0
    assert (p.encode_window_frames, p.decode_window_frames, p.delay, p.max_seq_frames, p.buffer_frames) == (128, 64, 2, 768, 32)
    assert abs(p.temperature - 0.7) < 1e-7 and abs(p.top_p - 0.7) < 1e-7


def test_no_cpu_fallback_without_gpu():
    import pytest
    import torch
    from streamvoiceanon_amd import engine as E

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        E.Engine({})


def _struct_fields(name):
    """(type, field, array length) of a struct in include/sva.h, in declaration order."""
    src = open(os.path.join(ROOT, "include", "sva.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), src, flags=re.S).group(1)
    out = []
    for ty, field, arr in re.findall(r"\b(int|float)\s+([a-z_0-9]+)(?:\[(\d+)\])?\s*;", body):
        out.append((ty, field, int(arr) if arr else 0))
    return out


def test_ctypes_structs_mirror_the_header():
    """The ctypes Structures of the binding have the header's fields, types and order: a mismatch would corrupt memory
    silently (the library writes sizeof(struct) bytes through the pointer it is given)."""
    from streamvoiceanon_amd import engine as E

    for cname, py in (("sva_config", E.SvaConfig), ("sva_stream_params", E.SvaStreamParams)):
        want = _struct_fields(cname)
        got = []
        for field, ctype in py._fields_:
            if hasattr(ctype, "_length_"):
                got.append(("int" if ctype._type_ is ctypes.c_int else "float", field, ctype._length_))
            else:
                got.append(("int" if ctype is ctypes.c_int else "float", field, 0))
        assert got == want, cname
        assert ctypes.sizeof(py) == sum(4 * max(n, 1) for _, _, n in want)
