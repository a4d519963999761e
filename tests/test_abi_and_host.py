"""CPU: the C-ABI library loads and exports every symbol include/dvq.h declares (no compute
calls without a GPU), argument validation that needs no device, host-side logic."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "dvq.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dvq_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    from dynamicvectorquantization_amd import _lib
    names = _declared()
    assert len(names) >= 10
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "libdvq.so does not export %s" % n
    assert set(names) == set(_lib.EXPORTS)
    assert _lib.lib.dvq_version() >= 100


def test_dynamic_symbol_table_is_exactly_the_abi():
    """libdvq.so is linked with -fvisibility=hidden and a version script (csrc/libdvq.map): `nm -D` shows the
    entry points include/dvq.h declares and NOTHING else -- no mangled launcher, no kernel stub"""
    import subprocess
    from dynamicvectorquantization_amd import _lib
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    syms = sorted(ln.split()[-1] for ln in out.splitlines() if ln.strip())
    assert not [s for s in syms if s.startswith("_Z") or "__device_stub__" in s or "__hip" in s], syms
    extra = set(syms) - set(_declared())
    if os.path.basename(_lib.LIB_PATH) != "libdvq.so":          # DVQ_LIBRARY=<tuning build>: plus its A/B switches
        extra = {s for s in extra if not s.startswith("dvq_tuning_")}
    assert not extra, extra
    assert set(_declared()) <= set(syms)


def test_header_is_plain_c_and_links_from_a_c_program(tmp_path):
    """the boundary is a C ABI, not a C++ one: include/dvq.h compiles as C99 with -pedantic and no warning, a C program that takes
    the address of EVERY declared entry point links against libdvq.so with gcc alone (no hipcc, no C++ runtime on its side), and its
    size queries answer without a GPU -- what a cgo / JNI / ctypes binding of the reference side would rely on"""
    import shutil
    import subprocess
    from dynamicvectorquantization_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    names = _declared()
    src = tmp_path / "abi_from_c.c"
    src.write_text("#include \"dvq.h\"\n#include <stdio.h>\n"
                   "typedef void (*entry_t)(void);\n"
                   "static const entry_t entry[] = {" + ", ".join("(entry_t)%s" % n for n in names) + "};\n"
                   "int main(void) {\n"
                   "    size_t i, n = sizeof entry / sizeof entry[0];\n"
                   "    for (i = 0; i < n; ++i) if (!entry[i]) return 2;\n"
                   "    printf(\"%d %zu %zu\\n\", dvq_version(), n, dvq_codebook_prep_bytes(1024, 256));\n"
                   "    return 0;\n}\n")
    exe = tmp_path / "abi_from_c"
    libdir = os.path.dirname(_lib.LIB_PATH)
    cc = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src),
                         "-L", libdir, "-l:" + os.path.basename(_lib.LIB_PATH), "-Wl,-rpath," + libdir, "-o", str(exe)],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-2000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr[-1000:]
    ver, n, prep = run.stdout.split()
    assert int(ver) == _lib.lib.dvq_version() and int(n) == len(names) and int(prep) == _lib.lib.dvq_codebook_prep_bytes(1024, 256)


def test_documents_name_only_entry_points_that_exist():
    """every `dvq_*` identifier README / INTEGRATION / DESIGN / profiles/README spell out in full is declared in include/dvq.h
    (patterns with braces or wildcards and prefixes ending in `_` are skipped); the tuning build's two extra exports, the
    snippet's own Python function and two file stems are the only other names"""
    decl = set(_declared())
    other = {"dvq_tuning_set", "dvq_tuning_buffers", "dvq_forward", "dvq_filter", "dvq_oracle", "dvq_common", "dvq_abi"}
    for doc in ("README.md", "INTEGRATION.md", "DESIGN.md", os.path.join("profiles", "README.md")):
        text = open(os.path.join(ROOT, doc)).read()
        names = set(re.findall(r"(?<![A-Za-z0-9_*{}])(dvq_[a-z0-9_]+)(?![A-Za-z0-9_*{}])", text))
        bad = sorted(n for n in names if n not in decl and n not in other and not n.endswith("_"))
        assert not bad, (doc, bad)


def test_documents_point_at_files_that_exist():
    """every back-ticked path under tools/, profiles/, tests/, oracle/, include/, csrc/ that the documents cite in full exists
    (`oracle/_ref` is the task's name for a compiled reference, which a Python reference does not have)"""
    for doc in ("README.md", "INTEGRATION.md", "DESIGN.md", os.path.join("profiles", "README.md")):
        text = open(os.path.join(ROOT, doc)).read()
        paths = set(re.findall(r"`((?:tools|profiles|tests|oracle|include|dynamicvectorquantization_amd|csrc|archive)/[A-Za-z0-9_./\-]+)`", text))
        missing = [p for p in sorted(paths) if p != "oracle/_ref" and not any(
            os.path.exists(os.path.join(ROOT, c)) for c in (p, os.path.join("dynamicvectorquantization_amd", p), os.path.join("profiles", p)))]
        assert not missing, (doc, missing)


def test_size_queries_and_validation_without_gpu():
    from dynamicvectorquantization_amd import _lib
    L = _lib.lib
    assert L.dvq_codebook_prep_bytes(1024, 256) >= 1024 * 256 * 4
    assert L.dvq_codebook_prep_bytes(0, 256) == 0
    assert L.dvq_vq_assign_workspace_bytes(256, 256, 1024, 1024, 0) >= (256 * 1024 // 128) * 8
    assert L.dvq_vq_assign_workspace_bytes(0, 256, 1024, 1024, 0) == 0
    # null pointers / unsupported shapes are rejected before anything touches a device
    assert L.dvq_vq_assign_nchw_f32(0, 0, 0, 0, 1, 256, 1, 1, 0.25, 0, 0, 0, 0, 0, 0, 0) == -1
    assert b"null" in L.dvq_last_error_string()
    assert L.dvq_vq_assign_nchw_f32(1, 1, 1, 0, 1, 100, 1, 1, 0.25, 0, 1, 0, 0, 0, 0, 0) == -2   # D=100
    assert L.dvq_vq_assign_nchw_f32(1, 1, 1, 0, 1, 256, 1, 1, 0.25, 0, 1, 0, 0, 0, 7, 0) == -1   # mode
    assert L.dvq_codebook_prepare_f32(1, 1024, 256, 256, 16, 0) == -3                          # too small
    assert L.dvq_route_select_dual_f32(0, 0, 0, 0, 1, 1, 1, 2, 0, 0, 0, 0) == -1
    assert L.dvq_embed_gather_f32(1, 4, 6, 1, 1, 1, 0) == -2                                    # D % 4
    # newer entry points: same discipline
    assert L.dvq_route_select_dual_entropy_f32(0, 1.0, 0, 0, 1, 1, 1, 2, 0, 0, 0, 0, 0) == -1
    assert L.dvq_ema_accumulate_nchw_f32(0, 0, 1, 64, 1, 8, 0, 0, 0) == -1
    assert L.dvq_entropy_map_f32(1, 1, 250, 256, 16, 1, 0) == -2                               # H % 16
    assert L.dvq_router_gate_workspace_bytes(2, 64, 256, 16, 16, 32, 512) >= 512 * 512 * 4 + 64 * 512 * 256 * 4
    assert L.dvq_router_gate_workspace_bytes(4, 64, 256, 16, 16, 32, 512) == 0                 # 2 or 3 branches
    assert L.dvq_router_gate_prep_bytes(2, 256, 512) >= 512 * 512 * 4
    assert L.dvq_router_gate_prepare_f32(1, 2, 256, 512, 256, 16, 0) == -3                     # buffer too small
    assert L.dvq_router_gate_prepare_f32(0, 2, 256, 512, 256, 1 << 30, 0) == -1                # null weight
    args = [2, 1, 0, 1, 1, 256, 16, 16, 32, 1e-6, 1, 1, 0, 0, 1, 1, 1, 1, 1, 1, 512, 1, 0, 1, 0, 0, 0]
    assert L.dvq_router_gate_f32(*args) == -3 and b"workspace" in L.dvq_last_error_string()
    args[5] = 100                                                                               # C % 8
    assert L.dvq_router_gate_f32(*args) == -1 or L.dvq_router_gate_f32(*args) == -2
    args[5], args[0] = 256, 5
    assert L.dvq_router_gate_f32(*args) == -1                                                  # branches
    assert L.dvq_permute_dual_count_i64(0, 1, 16, 16, 0, 0, 0) == -1
    # round-3 entry points: the conv fused into the assign, the backward kernels
    q = 256                                                                                     # a "pointer" that passes the alignment checks
    assert L.dvq_vq_assign_qconv_f32(0, q, 1, 1, 0, 1, 256, 1, 1, 0.25, 0, 1, 0, q, 0, q, 1 << 30, 1, 0) == -1      # null x
    assert L.dvq_vq_assign_qconv_f32(1, q, 1, 1, 0, 1, 128, 1, 1, 0.25, 0, 1, 0, q, 0, q, 1 << 30, 1, 0) == -2      # D = 128
    assert b"256" in L.dvq_last_error_string()
    assert L.dvq_vq_assign_qconv_f32(1, q, 1, 1, 0, 1, 256, 1, 1, 0.25, 0, 1, 0, q, 0, q, 1 << 30, 0, 0) == -1      # exact mode
    assert L.dvq_vq_assign_qconv_f32(1, q, 1, 1, 0, 1, 256, 1, 1, 0.25, 0, 1, 0, 0, 1, q, 1 << 30, 1, 0) == -1      # h_all without an h_buf (h_buf alone is optional since 0.6.0)
    assert L.dvq_vq_assign_qconv_f32(1, q, 1, 1, 0, 1, 256, 1024, 1024, 0.25, 0, 1, 0, q, 0, q, 16, 1, 0) == -3    # workspace
    assert L.dvq_vq_assign_routed_qconv_dual_f32(1, 0, 0.0, 1, 1, q, 1, 1, 1, 128, 4, 4, 64, 0.25, 0, 1, 0, 1, 1, 0, q, 0,
                                                 q, 1 << 30, 1, 0) == -2                                              # D = 128
    assert L.dvq_vq_assign_routed_qconv_triple_f32(1, 0, 1, 0, 1, q, 1, 1, 1, 256, 4, 4, 64, 0.25, 0, 1, 0, 1, 1, q, 0,
                                                   q, 1 << 30, 1, 0) == -1                                            # null h_median
    assert L.dvq_vq_backward_nchw_f32(0, 1, 1, 0, 1, 1, 1.0, 1, 256, 1, 8, 1, 0) == -1                              # null z
    assert L.dvq_vq_backward_nchw_f32(1, 16, 1, 0, 0, 0, 1.0, 1, 256, 1, 8, 1, 0) == -1                             # nothing to propagate
    assert L.dvq_vq_backward_nchw_f32(1, 16, 1, 0, 1, 1, 1.0, 1, 100, 1, 8, 1, 0) == -2                             # D % 16
    assert L.dvq_vq_backward_codebook_nchw_f32(1, 1, 1, 0, 0, 1.0, 1, 256, 1, 8, 1, 0) == -1                        # null g_loss
    assert L.dvq_vq_backward_codebook_nchw_f32(1, 1, 1, 0, 1, 1.0, 1, 256, 1, 10000, 1, 0) == -2                    # K > 8192


def test_cpu_tensors_fail_loudly():
    """the product path has no CPU fallback: CPU tensors raise instead of silently computing"""
    from dynamicvectorquantization_amd import _lib
    from dynamicvectorquantization_amd.quantize import VectorQuantize2, VectorQuantizer2
    from dynamicvectorquantization_amd.router import entropy_gate, route_select_dual
    vq = VectorQuantize2(64, 256).eval()
    with pytest.raises(_lib.DvqError):
        vq(torch.zeros(1, 256, 4, 4))
    with pytest.raises(_lib.DvqError):
        VectorQuantizer2(64, 256, 0.25)(torch.zeros(1, 256, 4, 4))
    with pytest.raises(_lib.DvqError):
        entropy_gate(torch.zeros(1, 16, 16), 1.0)
    with pytest.raises(_lib.DvqError):
        route_select_dual(torch.zeros(1, 2, 2, 2), torch.zeros(1, 4, 2, 2), torch.zeros(1, 4, 4, 4))


def test_state_dict_keys_match_reference():
    """checkpoint compatibility: same keys and shapes as the reference modules"""
    from dynamicvectorquantization_amd.quantize import VectorQuantize2, VectorQuantizer2
    from dynamicvectorquantization_amd.router import DualGrainFeatureRouter, TripleGrainFeatureRouter
    sd = VectorQuantize2(1024, 256).state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == {
        "codebook.weight": (1025, 256), "codebook.cluster_size_ema": (1024,), "codebook.embed_ema": (1024, 256)}
    vq = VectorQuantize2(1024, 256)
    assert float(vq.codebook.weight.abs().max()) <= 1.0 / 1024 and not vq.codebook.weight.requires_grad
    sd = VectorQuantizer2(1024, 256, beta=0.25).state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == {"embedding.weight": (1024, 256)}
    keys = list(DualGrainFeatureRouter(256, "group-32", "2layer-fc-SiLu").state_dict().keys())
    assert keys == ["gate.0.weight", "gate.0.bias", "gate.2.weight", "gate.2.bias",
                    "feature_norm_fine.weight", "feature_norm_fine.bias",
                    "feature_norm_coarse.weight", "feature_norm_coarse.bias"]
    keys = list(TripleGrainFeatureRouter(256, "group-32", "2layer-fc-SiLu").state_dict().keys())
    assert keys[:4] == ["gate.0.weight", "gate.0.bias", "gate.2.weight", "gate.2.bias"] and len(keys) == 10
    g = np.load(os.path.join(ROOT, "tests", "golden", "feature_router_dual.npz"))
    assert list(g["keys"]) == list(DualGrainFeatureRouter(256, "group-32", "2layer-fc-SiLu").state_dict().keys())


def test_entropy_router_threshold_lookup(golden_dir):
    from dynamicvectorquantization_amd.router import DualGrainFixedEntropyRouter
    js = os.path.join(golden_dir, "entropy_thresholds_imagenet_train_patch-16.json")
    assert DualGrainFixedEntropyRouter(js, 0.5).fine_grain_threshold == 1.6777750253677368
    assert DualGrainFixedEntropyRouter(json_path=js, fine_grain_ratito=0.3).fine_grain_threshold > 1.7
    with pytest.raises(KeyError):
        DualGrainFixedEntropyRouter(js, 1.5)


def test_synth_is_deterministic_and_offsettable():
    from dynamicvectorquantization_amd import synth
    a = synth.normal(5, (4, 7))
    assert np.array_equal(a, synth.normal(5, (4, 7))) and abs(float(a.mean())) < 1.0
    E = synth.codebook_trained(64, 256)
    full = synth.z_tokens(E, 4, 8, 8, 123)
    part = synth.z_tokens(E, 2, 8, 8, 123, image_offset=2)
    assert np.array_equal(full[2:], part)          # rank shards regenerate their slice of the global batch
    big = synth.normal(9, (1 << 16,))
    assert 0.98 < float(big.std()) < 1.02


def test_shard_slices_cover_batch():
    from dynamicvectorquantization_amd.encode import shard_slice
    for B, G in ((1024, 8), (10, 4), (3, 8), (256, 1)):
        sl = [shard_slice(B, r, G) for r in range(G)]
        assert sl[0][0] == 0 and sl[-1][1] == B
        assert all(sl[i][1] == sl[i + 1][0] for i in range(G - 1))
        assert max(e - s for s, e in sl) - min(e - s for s, e in sl) <= 1


def test_conv_prologue_isa_has_no_load_hazards():
    """the conv prologue of pass 1 (csrc/vq_assign_filter.hip, CONV form) issues its x loads as inline asm three k-steps ahead and
    waits with counted s_waitcnt: on the generated gfx950 ISA nothing may touch a load's destination registers before its covering
    wait, and the sixteen counted waits must be the ones the source placed (tools/isa_hazard_check.py; hipcc cross-compiles here)"""
    import os
    import shutil
    import subprocess
    import sys
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "isa_hazard_check.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("0 hazards, counted waits as placed") == 4, r.stdout      # dense + routed, one-workgroup and split forms


def test_product_library_reads_no_environment():
    """include/dvq.h promises that libdvq.so reads no environment variable (kernel choices are functions of the arguments and
    compile-time constants): the product library does not even import getenv; the A/B switches live in libdvq_tuning.so"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "dynamicvectorquantization_amd", "csrc", "libdvq.so")
    out = subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True, text=True).stdout
    assert out and "getenv" not in out
