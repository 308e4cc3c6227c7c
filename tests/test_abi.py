"""C-ABI contract checks that need no GPU: the library loads, exports every symbol include/fyc.h
declares, and the ctypes mirrors in followyourclick_amd/_lib.py have the C struct layout."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fyc.h")

STRUCTS = {
    "fyc_gemm_args": "GemmArgs", "fyc_attn_args": "AttnArgs", "fyc_tattn_args": "TAttnArgs",
    "fyc_gn_stats_args": "GnStatsArgs", "fyc_gn_apply_args": "GnApplyArgs", "fyc_layernorm_args": "LayerNormArgs",
    "fyc_softmax_args": "SoftmaxArgs", "fyc_concat_args": "ConcatArgs", "fyc_silu_args": "SiluArgs",
    "fyc_cast_args": "CastArgs", "fyc_cast_to_args": "CastArgs", "fyc_unet_input_args": "UnetInputArgs",
    "fyc_cfg_ddim_args": "CfgDdimArgs", "fyc_nchw_in_args": "NchwInArgs", "fyc_nhwc_out_args": "NhwcOutArgs",
    "fyc_embed_args": "EmbedArgs", "fyc_patchify_args": "PatchifyArgs", "fyc_row_stats_args": "RowStatsArgs",
    "fyc_pack_conv3x3_args": "PackConv3x3Args", "fyc_pack_geglu_args": "PackGegluArgs",
    "fyc_temporal_block_args": "TemporalBlockArgs", "fyc_ff_block_args": "FFBlockArgs",
    "fyc_panel_linear_args": "PanelLinearArgs",
}


@pytest.fixture(scope="module")
def lib():
    from followyourclick_amd import _build, _lib
    if not os.path.exists(_lib.LIB_PATH):
        _build.build(verbose=False)
    return _lib.load()


def _header_version():
    return int(re.search(r"#define FYC_VERSION (\d+)", open(HEADER).read()).group(1))


def test_binding_refuses_a_library_of_another_major_version(monkeypatch):
    """argument structs grow between majors: a binding must not call into a library whose major differs"""
    from followyourclick_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "FYC_VERSION", _lib.FYC_VERSION + 100)
    with pytest.raises(_lib.FycError, match="ABI version"):
        _lib.load()
    monkeypatch.setattr(_lib, "_lib", None)


def declared_functions():
    src = open(HEADER).read()
    return sorted(set(re.findall(r"\b(fyc_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/fyc.h but not exported by libfyc_hip.so"
    from followyourclick_amd import _lib
    assert lib.fyc_version() == _lib.FYC_VERSION == _header_version()


def test_binding_tables_cover_header():
    from followyourclick_amd import _lib
    assert sorted(list(_lib.OPS) + _lib.MISC) == declared_functions()


def test_struct_layouts_match_c(tmp_path):
    from followyourclick_amd import _lib
    prog = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){"]
    for c_name, py_name in STRUCTS.items():
        cls = getattr(_lib, py_name)
        prog.append(f'printf("{c_name} %zu\\n", sizeof({c_name}));')
        for fname, _ in cls._fields_:
            prog.append(f'printf("{c_name}.{fname} %zu\\n", offsetof({c_name}, {fname}));')
    prog.append("return 0;}")
    src = tmp_path / "abi.c"
    src.write_text("\n".join(prog))
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c11", "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    got = dict(line.split() for line in out.strip().splitlines())
    for c_name, py_name in STRUCTS.items():
        cls = getattr(_lib, py_name)
        assert int(got[c_name]) == ctypes.sizeof(cls), c_name
        for fname, _ in cls._fields_:
            assert int(got[f"{c_name}.{fname}"]) == getattr(cls, fname).offset, f"{c_name}.{fname}"


def test_argument_validation_without_gpu(lib):
    """error paths return a code + message instead of crashing (no kernel is launched)"""
    from followyourclick_amd import _lib
    g = _lib.GemmArgs()
    rc = lib.fyc_gemm(ctypes.byref(g), None)
    assert rc != 0 and b"fyc" in lib.fyc_last_error()
    assert lib.fyc_init(None) != 0
    assert lib.fyc_set_tuning(99, 1) != 0


def test_source_digest_identifies_the_kernel_sources():
    """bench.py matches PMC traffic profiles to the running library through this digest (and the binary's sha256)"""
    import json
    import os
    import re
    from followyourclick_amd._build import source_digest
    d = source_digest()
    assert re.fullmatch(r"[0-9a-f]{64}", d) and d == source_digest()
    prof = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    stamped = [json.load(open(os.path.join(prof, f))).get("source_sha256") for f in sorted(os.listdir(prof)) if f.endswith("_hbm_traffic.json")]
    assert any(stamped), "no traffic profile carries a source digest"
