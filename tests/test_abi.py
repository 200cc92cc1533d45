"""CPU: the C-ABI library builds/loads and exports every symbol include/creste_hip.h declares
(no compute calls -- there is no GPU here), and the product path refuses to run without it."""
import ctypes
import os
import re

import pytest
import torch

from creste_public_amd import _lib

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "creste_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(creste_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    if not os.path.exists(_lib.LIB_PATH):
        from creste_public_amd.build import build
        build(verbose=False)
    names = declared_symbols()
    assert len(names) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in creste_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in _lib.SIGNATURES"
    assert sorted(_lib.SIGNATURES) == names
    handle = _lib.load()
    assert handle.creste_abi_version() == _lib.ABI_VERSION
    # pure-host queries work without a GPU
    assert handle.creste_conv_packed_weight_bytes(496, 496, 3, 3, 0) == 512 * 9 * 496 * 4
    assert handle.creste_bev_splat_workspace_bytes(1, 100, 256, 256) > 0
    assert handle.creste_se_partial_count(4096, 96) == 26      # 10 pixel slices x 16 rows


def test_conv_engine_routing_is_a_host_query():
    """Which conv shapes each operand mode is built for (host-only query, creste_hip.h) and the host-side routing on
    top of it: every dense conv of the reference path runs on the f16x3 engine except the 4 -> 32 strided stem."""
    from creste_public_amd import ops
    F32, BF16X6, F16X3 = ops.PREC_F32, ops.PREC_BF16X6, ops.PREC_F16X3
    for k, s in ((1, 1), (3, 1), (5, 1), (7, 1), (1, 2), (3, 2), (7, 2)):
        assert ops.conv_supported(F16X3, k, s), (k, s)
        assert ops.conv_supported(F32, k, s)
    assert not ops.conv_supported(F16X3, 5, 2) and not ops.conv_supported(F16X3, 9, 1)
    # bf16x6: every row-kernel shape (the 7x7/2 stem's halo patch fits the LDS in three pieces as two row-parity passes)
    assert ops.conv_supported(BF16X6, 3, 1) and ops.conv_supported(BF16X6, 3, 2) and ops.conv_supported(BF16X6, 7, 1)
    assert ops.conv_supported(BF16X6, 5, 1) and ops.conv_supported(BF16X6, 1, 2) and ops.conv_supported(BF16X6, 7, 2)
    assert not ops.conv_supported(BF16X6, 5, 2)
    assert ops.conv_precision(F16X3, 7, 2, 96) == F16X3          # BEV stem
    assert ops.conv_precision(F16X3, 3, 2, 4) == F32             # encoder stem: one mostly-empty channel chunk
    assert ops.conv_precision(F16X3, 5, 1, 40) == F16X3          # reward FCN
    assert ops.conv_precision(BF16X6, 7, 2, 96) == BF16X6        # BEV stem (round 5; exact-fp32 engine before)
    lib = _lib.load()
    assert lib.creste_conv_packed_weight_bytes(64, 96, 7, 7, F16X3) == 4 * 6 * 49 * 2 * 2 * 64 * 16   # 4 padded units


def test_argument_errors_are_reported_not_thrown():
    handle = _lib.load()
    rc = handle.creste_conv2d_nhwc(None, None)
    assert rc == -1 and b"null descriptor" in handle.creste_last_error()


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setenv("CRESTE_HIP_LIB", "/nonexistent/libcreste_hip.so")
    with pytest.raises(_lib.HipLibraryError):
        _lib.load()
    monkeypatch.delenv("CRESTE_HIP_LIB")
    monkeypatch.setattr(_lib, "_lib", None)
    _lib.load()


def test_cpu_tensors_are_rejected():
    from creste_public_amd import ops
    with pytest.raises(_lib.HipLibraryError):
        ops.nchw_to_nhwc(torch.zeros(1, 4, 8, 8))


def test_plan_dispatch_table_is_current_and_loader_rejects_garbage(tmp_path):
    """csrc/plan_dispatch.inc (the marshalling thunks the Python-free runtime replays a plan through) is generated from
    SIGNATURES: the committed file must be what the generator produces; creste_hip_model_load refuses a file that is not
    a plan, and a plan naming an unknown entry point -- host-side parsing only, no GPU needed."""
    import importlib.util
    import struct
    spec = importlib.util.spec_from_file_location("gen", os.path.join(ROOT, "scripts", "gen_plan_dispatch.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    committed = open(os.path.join(ROOT, "creste_public_amd", "csrc", "plan_dispatch.inc")).read()
    assert gen.generate() == committed, "run scripts/gen_plan_dispatch.py"
    lib = _lib.load()
    h = ctypes.c_void_p()
    bad = tmp_path / "bad.plan"
    bad.write_bytes(b"not a plan at all")
    assert lib.creste_hip_model_load(str(bad).encode(), 0, ctypes.byref(h)) != 0
    assert b"not a creste plan" in lib.creste_last_error()
    assert lib.creste_hip_model_load(b"/nonexistent.plan", 0, ctypes.byref(h)) != 0
    assert lib.creste_hip_model_num_outputs(None) == -1


def test_hot_kernels_stay_out_of_scratch():
    """register spills are silent and expensive (a per-round gate array once pushed the common 1x1 conv kernel into 132
    bytes of scratch per lane: +1 ms per batch-16 step): the compiler's resource remarks of the last build are kept next
    to the objects (creste_public_amd/build.py) and every kernel must be spill-free unless it is on this short list."""
    from creste_public_amd import build
    build.build(verbose=False)
    usage = build.resource_usage()
    assert len(usage) > 200, "kernel-resource-usage remarks missing: rebuild with `python -m creste_public_amd.build --force`"
    allowed = {                                    # mangled-name fragment -> bytes per lane tolerated
        "conv_patch3_kernelILi2ELi2ELb0EE": 48,       # bf16x3 twin
        "conv_patch_kernelILi1ELi2ELi2ELb1ELb0E": 16,
        "conv_patch_kernelILi1ELi2ELi2ELb1ELb1E": 160,  # gated flat 1x1 at 128-channel tiles: not reached by the network
        "conv_patch_row_kernelILi7E": 48,
        "wgrad_rows_kernelILi5ELi25E": 40,            # 100 accumulators at 3 waves/SIMD: the row-prefetch registers spill
                                                      # around the matrix loop (3 scratch ops per row, none inside it)
        # the rendezvous form of the persistent MDP solver (CRESTE_VI_SYNC=1) is built for 6 waves per SIMD (80 registers) so
        # that two 9-wave workgroups fit a CU under any wave placement (csrc/value_iteration.hip); what spills sits at the
        # chunk head / in the redo path, the eight sweeps between the barriers are scratch-free (checked in the ISA)
        "vi_persist_kernelILi1E": 12,
    }
    bad = []
    for name, u in usage.items():
        lim = max([v for k, v in allowed.items() if k in name] or [0])
        if u.get("scratch", 0) > lim:
            bad.append((name, u["scratch"], lim))
    assert not bad, bad


def test_plan_loader_rejects_foreign_abi_and_descriptor_layout(tmp_path):
    """creste_hip_model_load checks the plan header BEFORE touching the device: a plan exported under another C-ABI
    version or with another creste_conv_desc layout must not load (its recorded arguments would be misread)."""
    import struct
    handle = _lib.load()
    desc = ctypes.sizeof(_lib.ConvDesc)

    def try_load(version, desc_size, abi):
        p = tmp_path / f"h_{version}_{desc_size}_{abi}.plan"
        p.write_bytes(b"CRESTEPLAN\0\0" + struct.pack("<III", version, desc_size, abi) + struct.pack("<I", 0) + struct.pack("<I", 0))
        h = ctypes.c_void_p()
        rc = handle.creste_hip_model_load(str(p).encode(), 0, ctypes.byref(h))
        return rc, handle.creste_last_error().decode()

    rc, msg = try_load(2, desc, _lib.ABI_VERSION)        # round 4's format: calls without a stream, no stream-order edges
    assert rc != 0 and "format version" in msg
    rc, msg = try_load(3, desc, _lib.ABI_VERSION - 1)
    assert rc != 0 and "C-ABI version" in msg
    rc, msg = try_load(3, desc - 8, _lib.ABI_VERSION)
    assert rc != 0 and "descriptor" in msg
    # the per-argument kind signatures the loader validates against are generated with the thunks
    inc = open(os.path.join(ROOT, "creste_public_amd", "csrc", "plan_dispatch.inc")).read()
    assert '{"creste_conv2d_nhwc", 1, "D", thunk_creste_conv2d_nhwc}' in inc
    assert re.search(r'\{"creste_fill_u32", 3, "pil", ', inc)


def test_no_packed_fp32_valu_in_the_device_code(tmp_path):
    """The library is built without v_pk_{fma,mul,add}_f32 (creste_public_amd/build.py: NO_PK): on gfx950 their results were
    observed to be corrupted while another kernel's MFMA waves share the CU (two streams: pipelined inference, the IRL
    prefetch).  Disassemble every gfx950 code object of the shipped library and look."""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    so = shutil.copy(_lib.LIB_PATH if hasattr(_lib, "LIB_PATH") else
                     os.path.join(os.path.dirname(_lib.__file__), "lib", "libcreste_hip.so"), tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", str(so)], check=True, cwd=tmp_path, stdout=subprocess.DEVNULL)
    cos = sorted(p for p in os.listdir(tmp_path) if "gfx950" in p)
    assert len(cos) >= 10, cos
    pat = re.compile(r"\bv_pk_\w+")            # (today no packed VALU op of any type is left; conversions are v_cvt_pk_*)
    hits = 0
    for co in cos:
        txt = subprocess.run([objdump, "-d", str(tmp_path / co)], check=True, capture_output=True, text=True).stdout
        assert "v_mfma" in txt or "s_endpgm" in txt
        hits += len(pat.findall(txt))
    assert hits == 0, f"{hits} packed-fp32 VALU instructions in the device code"
