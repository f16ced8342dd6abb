"""CPU: the C ABI loads, exports every symbol include/emspec.h declares, and fails loudly
without a GPU (there is no CPU fallback in the product)."""
import ctypes as C
import os
import re
import shutil
import subprocess
import sys

import pytest
import torch

import emspec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAS_GPU = torch.cuda.is_available()


def declared_functions(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(emspec_[a-z_0-9]+)\s*\(", src)))


def exported_functions(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted({ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("emspec_")})


def test_header_symbols_exported():
    """The product library exports exactly what include/emspec.h declares: no diagnostic entry points."""
    lib = emspec.load()
    names = declared_functions("emspec.h")
    assert len(names) >= 15
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/emspec.h but not exported by libemspec.so"
    assert set(emspec.SYMBOLS) <= set(names)
    assert exported_functions(emspec.LIB_PATH) == names


def test_diag_library_adds_only_the_debug_header():
    dbg = declared_functions("emspec_debug.h")
    assert "emspec_debug_row_lookup" in dbg and "emspec_debug_phase_cycles" in dbg and "emspec_debug_recip" in dbg and "emspec_debug_recip64" in dbg
    assert exported_functions(emspec.DIAG_LIB_PATH) == sorted(set(declared_functions("emspec.h")) | set(dbg))
    out = subprocess.run(["strings", "-n", "6", emspec.LIB_PATH], capture_output=True, text=True).stdout
    for needle in ("EMSPEC_FUSED_VARIANT", "EMSPEC_NO_FUSED", "EMSPEC_SEGLEN", "EMSPEC_NO_WALK", "fused4096_r8t", "fused4096_r8_kernel", "occupy_kernel"):
        assert needle not in out, f"{needle} is diagnostic and must not be in libemspec.so"


def test_library_has_gfx950_code_object():
    out = subprocess.run(["strings", "-n", "6", emspec.LIB_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in out
    assert "fused4096" in out and "frames_kernel" in out


def test_pure_functions():
    assert emspec.num_columns(1 << 22, 4096, 256) == 16369
    assert emspec.num_columns(4095, 4096, 256) == 0
    assert emspec.latency_columns(4096, 256, True) == 8
    assert emspec.latency_columns(16384, 512, True) == 16
    assert emspec.latency_columns(4096, 256, False) == 0
    cfg = emspec.default_config()
    assert (cfg.rows, cfg.abi_version) == (1024, 2)
    assert cfg.sample_rate == 48000.0 and cfg.fmin_hz == 20.0


@pytest.mark.skipif(HAS_GPU, reason="checks the no-GPU failure mode")
def test_create_fails_loudly_without_gpu():
    with pytest.raises(emspec.EmspecError) as ei:
        emspec.Engine()
    assert ei.value.code == emspec.ERR_NO_DEVICE
    assert "no CPU path" in str(ei.value)


def test_create_rejects_bad_config():
    lib = emspec.load()
    h = C.c_void_p()
    cfg = emspec.default_config(rows=1000 + 1)
    assert lib.emspec_create(C.byref(cfg), C.byref(h)) == emspec.ERR_INVALID_ARG
    cfg = emspec.default_config(abi_version=99)
    assert lib.emspec_create(C.byref(cfg), C.byref(h)) == emspec.ERR_INVALID_ARG
    assert b"abi_version" in lib.emspec_last_error(None)


def test_build_info_carries_the_sources_sha():
    """emspec_build_info: the sha1 of the kernel sources the LOADED library was compiled from (tools/sources_sha.py, compiled
    in by the Makefile) - what bench.py compares with the tree and the profile before it quotes a counter."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import sources_sha
    finally:
        sys.path.pop(0)
    for diag in (False, True):
        info = emspec.build_info(diag)
        assert info.startswith("emspec abi=2 sources=") and info.endswith(" arch=gfx950"), info
        assert f"sources={sources_sha.sources_sha()}" in info, (info, "the in-tree .so is older than the sources: rebuild")
    assert emspec.load().emspec_mode(None) == -1


def test_version_1_clients_are_rejected():
    """ABI 2 reads emspec_config.mode (reserved and ignored in version 1): a client built against the old header must
    fail emspec_create instead of getting an engine whose mode field was never initialised."""
    lib = emspec.load()
    h = C.c_void_p()
    cfg = emspec.default_config(abi_version=1)
    assert lib.emspec_create(C.byref(cfg), C.byref(h)) == emspec.ERR_INVALID_ARG


def test_product_does_not_link_the_oracle():
    """The oracle is test infrastructure: libemspec.so and the addon must not depend on it."""
    out = subprocess.run(["ldd", emspec.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out
    for root, _, files in os.walk(os.path.join(ROOT, "em-spec_amd")):
        for f in files:
            if f.endswith((".cpp", ".hip", ".h", ".inc", ".c", ".js", ".py")):
                for line in open(os.path.join(root, f), errors="ignore"):
                    code = line.split("//")[0]
                    bad = ("#include" in code and "oracle" in code) or "libemspec_oracle" in code \
                        or re.match(r"\s*(import|from)\s+(oracle|ref_numpy)\b", code)
                    assert not bad, (f, line)


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_node_addon_loads_and_reports_errors():
    js = os.path.join(ROOT, "em-spec_amd", "js")
    if not os.path.exists(os.path.join(js, "emspec.node")):
        pytest.skip("addon not built")
    code = ("const m=require('./index.js');"
            "if(m.numColumns(4194304,4096,256)!==16369) process.exit(2);"
            "if(m.latencyColumns(4096,256,true)!==8) process.exit(3);"
            "try{m.createEngine({rows:1001});process.exit(4);}catch(e){if(e.code!=='EMSPEC_ERR_INVALID_ARG')process.exit(5);}"
            + ("" if HAS_GPU else
               "try{m.createEngine({});process.exit(6);}catch(e){if(e.code!=='EMSPEC_ERR_NO_DEVICE')process.exit(7);}"))
    r = subprocess.run(["node", "-e", code], cwd=js, capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stderr)


def test_flagship_kernels_do_not_spill():
    """The fused kernels run 16 waves per CU on exactly 128 VGPRs each; spilled registers are reloaded serially
    behind vmcnt and once cost 2x.  Parse the code-object metadata of a fresh device-only compile;
    up to two spilled dwords are tolerated (today: one, in the once-per-workgroup prologue of the hop-256 build)."""
    import re, shutil, subprocess
    if not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    csrc = os.path.join(ROOT, "em-spec_amd", "csrc")
    subprocess.check_call(["make", "-s", "-C", csrc, "asm"])
    text = open(os.path.join(csrc, "kernels.s")).read()
    seen = 0
    for m in re.finditer(r"\.name:\s+(\S+)\n", text):
        name = m.group(1)
        if not any(k in name for k in ("fused4096_pp_kernelILi256ELb0", "fused4096_pp_kernelILi512ELb0", "fused8192_kernel",
                                       "fused_small_pp_kernel", "fused16384_kernel")):
            continue
        blk = text[m.start() - 400:m.start() + 1600]
        meta = dict(re.findall(r"\.(vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):\s+(\d+)", blk))
        assert int(meta["vgpr_spill_count"]) <= 2 and int(meta["private_segment_fixed_size"]) <= 16, (name, meta)
        assert int(meta["vgpr_count"]) <= 128, (name, meta)
        seen += 1
    assert seen >= 14, seen
    # the EXACT mode's flagships: the one-kernel N = 4096 path in its branch-free form, and the N = 16384 frames
    # kernel of the records path - 1024-thread workgroups, so 128 VGPRs is the ceiling and a spill is paid by sixteen lock-step waves
    xseen = 0
    for m in re.finditer(r"\.name:\s+(\S+)\n", text):
        name = m.group(1)
        # (exact_fused4096_lr_kernel<STAMP = false, FASTX, PAF = false, ABL = 0, S>: both per-bin cores, all three sizes)
        if not any(k in name for k in ("exact_fused4096_lr_kernelILb0ELb1ELb0ELi0E", "exact_fused4096_lr_kernelILb0ELb0ELb0ELi0E",
                                       "exact_frames16384_kernelILb1ELb0", "exact_frames16384_kernelILb0ELb0")):
            continue
        blk = text[m.start() - 400:m.start() + 1600]
        meta = dict(re.findall(r"\.(vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):\s+(\d+)", blk))
        assert int(meta["vgpr_spill_count"]) == 0 and int(meta["private_segment_fixed_size"]) == 0, (name, meta)
        assert int(meta["vgpr_count"]) <= 128, (name, meta)
        xseen += 1
    assert xseen >= 8, xseen


def test_display_laws_are_shared_by_every_binding():
    """emspec_warped_edges_hz / emspec_make_colormap: one implementation behind the Python and the JS helpers."""
    import json
    import sys
    import numpy as np
    e = emspec.warped_edges_hz(1024, 20.0, 24000.0, 1.0, 1.0)
    r = np.arange(1025) / 1024.0
    assert np.allclose(e, 20.0 * (24000.0 / 20.0) ** r, rtol=3e-7)            # (1, 1) is the plain log axis
    w = emspec.warped_edges_hz(512, 30.0, 20000.0, 2.0, 1.5)
    assert w[0] == np.float32(30.0) and np.all(np.diff(w) > 0)
    assert abs(w[-1] - 30.0 * (20000.0 / 30.0) ** (1 / 1.5)) < 1e-2
    assert w[256] < 30.0 * (20000.0 / 30.0) ** (0.5 / 1.5)                      # low-end boost: half the rows end lower
    with pytest.raises(emspec.EmspecError):
        emspec.warped_edges_hz(16, 100.0, 50.0)
    lut = emspec.make_colormap(0.5)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    assert np.array_equal(lut, O.default_lut())                                  # brightness 0.5 = the engine's default palette
    assert emspec.make_colormap(1.0)[128].tolist() == [255, 255, 200, 255]       # twice as bright: saturates half way
    node = shutil.which("node") or shutil.which("nodejs")
    if node and os.path.exists(os.path.join(ROOT, "em-spec_amd", "js", "emspec.node")):
        js = ("const em=require('%s');process.stdout.write(JSON.stringify({w:Array.from(em.warpedEdges(512,30,20000,2,1.5)),"
              "c:Array.from(em.makeColormap(0.7))}))" % os.path.join(ROOT, "em-spec_amd", "js", "index.js"))
        out = json.loads(subprocess.run([node, "-e", js], capture_output=True, text=True, check=True, timeout=60).stdout)
        assert np.array_equal(np.array(out["w"], np.float32), w)
        assert np.array_equal(np.array(out["c"], np.uint8).reshape(256, 4), emspec.make_colormap(0.7))


def test_wire_unpack_host_matches_the_numpy_wire_reference():
    """emspec_wire_unpack_host (plain C in the product library: no device, no engine) expands the images that the numpy
    restatement of the wire format (oracle/wire_ref.py) packs - random sparse columns, dense columns, empty columns, a row
    count that is not a multiple of 32 - and rejects damaged images instead of reading or writing out of range."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import wire_ref as W
    rng = np.random.default_rng(77)
    for columns, rows, density in ((1, 64, 0.5), (37, 1024, 0.06), (300, 1024, 0.0), (12, 100, 1.0), (129, 256, 0.3)):
        idx = (rng.integers(1, 256, size=(columns, rows)) * (rng.random((columns, rows)) < density)).astype(np.uint8)
        img = W.pack(idx)
        got = emspec.wire_unpack_host(img, columns, rows)
        assert np.array_equal(got, idx), (columns, rows, density)
        assert np.array_equal(emspec.wire_unpack_host(np.concatenate([img, np.zeros(40, np.uint8)]), columns, rows), idx)   # slack behind the image
        for bad in (img[:31], img[:len(img) // 2] if density > 0 else img[:32 + columns * 2]):
            with pytest.raises(emspec.EmspecError):
                emspec.wire_unpack_host(bad, columns, rows)
        with pytest.raises(emspec.EmspecError):
            emspec.wire_unpack_host(img, columns + 1, rows)
        wrong = img.copy()
        wrong[0] ^= 1                                        # magic
        with pytest.raises(emspec.EmspecError):
            emspec.wire_unpack_host(wrong, columns, rows)
    # a mask that claims more cells than the payload holds is caught while expanding
    idx = np.zeros((2, 64), np.uint8)
    idx[0, 3] = 9
    img = W.pack(idx).copy()
    img[32 + 2 * 4 + 4] = 0xFF                               # column 0's second mask word: eight more cells than the payload has
    with pytest.raises(emspec.EmspecError):
        emspec.wire_unpack_host(img, 2, 64)
