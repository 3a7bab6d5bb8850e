"""CPU: the C-ABI library loads, exports exactly what include/hpf.h declares,
and refuses to run without a GPU (no CPU fallback anywhere on the product path)."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import pytest
import torch

from hgaprec_amd import capi

ROOT = Path(__file__).resolve().parent.parent


def header_functions():
    text = (ROOT / "include" / "hpf.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hpf_[a-z_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert header_functions() == sorted(capi.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = capi.load_library()
    for name in header_functions():
        assert getattr(lib, name) is not None
    hdr = (Path(capi.__file__).resolve().parent.parent / "include" / "hpf.h").read_text()
    assert lib.hpf_abi_version() == int(re.search(r"#define HPF_ABI_VERSION (\d+)", hdr).group(1)) == 8
    assert lib.hpf_strerror(0) == b"ok"
    assert b"device" in lib.hpf_strerror(-2)


def test_config_struct_layout_matches_header():
    # 12 x 4-byte fields, one pointer, two doubles, novb + reserved (ABI v4)
    assert C.sizeof(capi.HpfConfig) == 12 * 4 + 8 + 16 + 8
    assert C.sizeof(capi.HpfTiming) == 36            # 8 floats + the iteration counter
    assert C.sizeof(capi.HpfWorkInfo) == 8 + 24 * 4 + 2 * 8      # ABI v5: + tile rows and the heavy bars of both sides; v7: + start_sums_pending; v8: + tile_chunk_user / _item, reserved0


def test_no_oracle_on_the_product_path():
    """the shipped package and native sources never reference oracle/"""
    for p in list((ROOT / "hgaprec_amd").rglob("*.py")) + list((ROOT / "hgaprec_amd" / "csrc").rglob("*.*")):
        if p.suffix in (".py", ".cpp", ".hpp", ".hip", ".h"):
            t = p.read_text()
            assert "liborc" not in t and "from oracle" not in t and "import oracle" not in t, p


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a box WITHOUT a GPU")
def test_create_fails_loudly_without_gpu():
    with pytest.raises(capi.HpfError):
        capi.Hpf(10, 10, 4)


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a box WITHOUT a GPU")
def test_page_locked_memory_is_refused_without_gpu():
    """hpf_host_alloc (ABI v6) says HPF_ERR_NO_DEVICE without a HIP device -- callers fall back to malloc -- and
    freeing nothing is fine"""
    import ctypes as C
    lib = capi.load_library()
    p = C.c_void_p(1)
    assert lib.hpf_host_alloc(C.byref(p), 1 << 20) == -2 and not p.value
    assert lib.hpf_host_alloc(None, 16) == -1
    assert lib.hpf_host_free(None) == 0
    with pytest.raises(capi.HpfError):
        capi.pinned_empty((4, 4))


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a box WITHOUT a GPU")
def test_cli_fails_loudly_without_gpu(tmp_path):
    d = tmp_path / "data"
    d.mkdir()
    for f in ("train.tsv", "validation.tsv", "test.tsv"):
        (d / f).write_text("1\t1\t3\n2\t1\t4\n1\t2\t5\n")
    r = subprocess.run([str(ROOT / "hgaprec_amd" / "hgaprec"), "-dir", str(d), "-n", "5", "-m", "5",
                        "-k", "2", "-hier"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode != 0
    assert "no CPU fallback" in r.stderr


def test_cli_usage_and_unknown_option(tmp_path):
    exe = str(ROOT / "hgaprec_amd" / "hgaprec")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("gaprec -dir")
    r = subprocess.run([exe, "-dir", "x", "-bogus"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode != 0 and "error: unknown option -bogus" in r.stdout
    r = subprocess.run([exe, "-dir", "x", "-nmf"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 2 and "outside the MI355X hot-path build" in r.stderr


def test_cli_takes_novb_on_one_gpu_and_across_ranks(tmp_path):
    """-novb only changes the reference in vb_bias() (-bias without -hier,
    hgaprec.cc:1276-1297): both rate updates there use the previous iteration's
    expectations.  Round 3 built that order on one GPU (hpf_config.novb), round 4 across
    ranks too (the start state's sum_u E[theta] is reduced once, hpf_start_sums; parity in
    tests/test_gpu_parity.py, test_gpu_cli.py and test_gpu_multi.py).  With -hier (or
    without -bias) the reference never reads the flag."""
    exe = str(ROOT / "hgaprec_amd" / "hgaprec")
    for extra in ([], ["-hier"]):
        r = subprocess.run([exe, "-dir", str(tmp_path / "missing"), "-n", "5", "-m", "5", "-k", "2", "-bias", "-novb"] + extra,
                           cwd=tmp_path, capture_output=True, text=True)
        assert "outside the MI355X hot-path build" not in r.stderr      # accepted (fails later: no GPU / no data)
    r = subprocess.run([exe, "-dir", str(tmp_path / "missing"), "-n", "5", "-m", "5", "-k", "2", "-bias", "-novb",
                        "-single-allreduce", "-comm", "host"], cwd=tmp_path, capture_output=True, text=True)
    assert "unknown option" not in r.stdout and "outside the MI355X hot-path build" not in r.stderr
