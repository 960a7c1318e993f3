"""CPU: the C-ABI library loads, exports every symbol include/aid_hip.h declares, and the ctypes mirrors of the
parameter-table structs have exactly the C layout (checked with a gcc-compiled sizeof/offsetof probe).  No GPU calls."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "aid_hip.h")


@pytest.fixture(scope="module")
def nat():
    from aid_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _native


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(aid_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(nat):
    lib = C.CDLL(nat.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.aid_version.restype = C.c_int
    hdr = open(os.path.join(ROOT, "include", "aid_hip.h")).read()
    want = int(re.search(r"#define AID_ABI_VERSION (\d+)", hdr).group(1))
    from aid_amd import _native
    assert lib.aid_version() == want == _native.ABI_VERSION           # no GPU needed
    lib.aid_last_error.restype = C.c_char_p
    assert isinstance(lib.aid_last_error(), bytes)


def test_ctypes_structs_match_c_layout(nat, tmp_path):
    structs = ["AidDiTBlock", "AidScoreNet", "AidLinear", "AidLN", "AidPolicyNet", "AidValueNet", "AidDynamicsNet", "AidRewardNet",
               "AidDecoderNet", "AidEpistemicNet", "AidDiffusionParams", "AidElboNoise", "AidElboCfg", "AidEfeNoise", "AidEfeCfg",
               "AidEncoderNet", "AidFeatureDecoder", "AidSNConv", "AidDecoderBlock", "AidConvDecoder"]
    probes = {"AidScoreNet": ["blk", "adaf_w", "op2_w"], "AidPolicyNet": ["trunk", "lstd2"], "AidValueNet": ["net", "out"],
              "AidEpistemicNet": ["jp_ln", "mn6"], "AidElboCfg": ["seed", "defer_time_importance"], "AidEfeCfg": ["epistemic_weight", "seed"],
              "AidDecoderNet": ["ln", "out"], "AidLinear": ["out"], "AidElboNoise": ["d_obs_out"], "AidEncoderNet": ["att_conv_b", "out5"],
              "AidFeatureDecoder": ["l4"], "AidConvDecoder": ["blk", "out6_b"], "AidDecoderBlock": ["res"]}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){"]
    for s in structs:
        lines.append(f'printf("{s} %zu\\n", sizeof({s}));')
        for f in probes.get(s, []):
            lines.append(f'printf("{s}.{f} %zu\\n", offsetof({s}, {f}));')
    lines.append("return 0;}")
    src = tmp_path / "probe.c"; src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-std=c11", str(src), "-o", str(exe)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines())
    for s in structs:
        py = getattr(nat, s)
        assert C.sizeof(py) == int(out[s]), (s, C.sizeof(py), out[s])
        for f in probes.get(s, []):
            pf = "out" if f == "out" else f
            assert getattr(py, pf).offset == int(out[f"{s}.{f}"]), (s, f)


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under the product package may reference it."""
    pkg = os.path.join(ROOT, "active-inference-diffusion_amd")
    bad = []
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M):
                    bad.append(os.path.join(d, f))
    assert not bad, bad


def test_missing_library_fails_loudly(nat, monkeypatch):
    monkeypatch.setattr(nat, "_lib", None)
    monkeypatch.setattr(nat, "LIB_PATH", "/nonexistent/libaid_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        nat.lib()
