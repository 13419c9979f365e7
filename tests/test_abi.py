"""The C ABI: headers compile as C and C++, the Python mirrors have the C
layout, and the built libraries export every function the headers declare.
No compute call is made, so this runs without a GPU."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

from raytracer_amd import _ffi as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INCLUDE = os.path.join(ROOT, "include")


def declared_functions(header):
    text = open(os.path.join(INCLUDE, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rth?_[a-z0-9_]+)\s*\(", text)))


def test_headers_compile_as_c_and_cpp(tmp_path):
    src = tmp_path / "abi.c"
    src.write_text('#include "rt_mi355.h"\n#include "rt_host.h"\n'
                   "#include <stddef.h>\n"
                   "_Static_assert(sizeof(RtNode) == 32, \"RtNode\");\n"
                   "_Static_assert(sizeof(RtSphere) == 20, \"RtSphere\");\n"
                   "_Static_assert(sizeof(RtPlane) == 28, \"RtPlane\");\n"
                   "_Static_assert(sizeof(RtTriangle) == 52, \"RtTriangle\");\n"
                   "_Static_assert(sizeof(RtMaterial) == 40 && offsetof(RtMaterial, tex_offset) == 32, \"RtMaterial\");\n"
                   "_Static_assert(sizeof(RtHitRecord) == 12, \"RtHitRecord\");\n"
                   f"_Static_assert(sizeof(RtSceneDesc) == {C.sizeof(F.RtSceneDesc)}, \"RtSceneDesc\");\n"
                   f"_Static_assert(sizeof(RtRenderParams) == {C.sizeof(F.RtRenderParams)}, \"RtRenderParams\");\n"
                   f"_Static_assert(offsetof(RtRenderParams, rng_seed) == {F.RtRenderParams.rng_seed.offset}, \"seed\");\n"
                   f"_Static_assert(offsetof(RtRenderParams, tile_x) == {F.RtRenderParams.tile_x.offset}, \"tile\");\n"
                   f"_Static_assert(sizeof(RtStats) == {C.sizeof(F.RtStats)}, \"RtStats\");\n"
                   f"_Static_assert(offsetof(RtStats, trace_ms) == {F.RtStats.trace_ms.offset}, \"trace_ms\");\n"
                   f"_Static_assert(RT_ABI_VERSION == {F.RT_ABI_VERSION}, \"version\");\n"
                   "int main(void) { return 0; }\n")
    subprocess.run(["gcc", "-std=c11", "-pedantic", "-Wall", "-Werror", "-I", INCLUDE, "-fsyntax-only", str(src)], check=True)
    cpp = tmp_path / "abi.cpp"
    cpp.write_text('#include "rt_mi355.h"\n#include "rt_host.h"\nint main() { return RT_OK; }\n')
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I", INCLUDE, "-fsyntax-only", str(cpp)], check=True)


def test_device_library_exports_every_declared_function():
    names = declared_functions("rt_mi355.h")
    assert "rt_render" in names and "rt_upload_scene" in names and len(names) >= 13
    lib = F.device_lib()  # loads, binds every symbol, checks the ABI version; no GPU call
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(F._DEVICE_SYMBOLS) == names
    assert lib.rt_abi_version() == F.RT_ABI_VERSION


def test_host_library_exports_every_declared_function():
    names = [n for n in declared_functions("rt_host.h") if n.startswith("rth_")]
    lib = F.host_lib()
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(F._HOST_SYMBOLS) == names


def test_null_handles_are_errors_not_crashes():
    lib = F.device_lib()
    assert lib.rt_set_option(None, 1, 1) == -1
    assert lib.rt_get_stats(None, None) == -1
    assert lib.rt_upload_scene(None, None) == -1
    assert lib.rt_last_error(None) == b"null context"
    lib.rt_destroy(None)


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    monkeypatch.setattr(F, "LIB_DIR", str(tmp_path))
    monkeypatch.setattr(F, "_device", None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        F.device_lib()


def test_product_package_never_touches_the_oracle():
    # only tests/, bench.py's cpu_baseline and smoke() may use oracle/
    pkg = os.path.join(ROOT, "raytracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower(), os.path.join(dirpath, f)


def test_every_declared_entry_point_is_bound_in_python_and_in_the_rust_binding():
    import re

    header = open(os.path.join(ROOT, "include", "rt_mi355.h")).read()
    declared = set(re.findall(r"^(?:int|void\*?|uint32_t|const char\*|RtContext\*)\s+(rt_\w+)\(", header, flags=re.M))
    assert {"rt_create", "rt_render", "rt_trace_rays_noalloc", "rt_get_stats", "rt_group_render", "rt_group_member"} <= declared
    assert declared == set(F._DEVICE_SYMBOLS)
    rust = open(os.path.join(ROOT, "bindings", "rt_mi355.rs")).read()
    bound = set(re.findall(r"pub fn (rt_\w+)\(", rust))
    assert declared <= bound, sorted(declared - bound)


def test_counter_record_is_quoted_only_for_the_kernels_it_was_taken_from(monkeypatch):
    """The hardware-counter record bench.py quotes is stamped with a hash of the kernels'
    code (comments and blank space aside); a record of other kernels is not quoted."""
    import argparse

    import bench

    assert bench._code_only('a = "x // y"; // c\n/* z */ b = \'"\';  int\tq;') == 'a="x // y";b=\'"\';intq;'
    args = argparse.Namespace(workload="room", width=1920, height=1080, bounces=60, detail=4, fast=False)
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: "not the recorded one")
    assert bench.recorded_counters(args) is None


def test_every_exported_definition_is_exception_guarded():
    """include/rt_mi355.h promises that nothing throws across the boundary: every definition
    of an exported function in the device library's sources must be a function-try-block
    closed by one of the guard macros of device/abi_guard.h."""
    device = os.path.join(ROOT, "raytracer_amd", "csrc", "device")
    exported = set(declared_functions("rt_mi355.h"))
    defined = {}
    for name in sorted(os.listdir(device)):
        if not name.endswith(".hip"):
            continue
        text = open(os.path.join(device, name)).read()
        # a definition: return type and name at the start of a line, the signature up to its opening brace
        for m in re.finditer(r"^(?:extern \"C\" )?(?:int|void\*?|const char\*|uint32_t|RtContext\*) (rt_[a-z0-9_]+)\(([^;{]*?)\)\s*(try\s*)?\{",
                             text, flags=re.M | re.S):
            if m.group(1) in exported:
                defined[m.group(1)] = (name, bool(m.group(3)))
        assert text.count(") try {") == len(re.findall(r"\} RT_ABI_END_(?:CTX|INT|VOID|VALUE)\b", text)), name
    assert set(defined) == exported, sorted(exported - set(defined))
    unguarded = sorted(n for n, (_, guarded) in defined.items() if not guarded)
    assert not unguarded, f"exported without an exception guard: {unguarded}"
