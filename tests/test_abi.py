"""The C-ABI library loads on a CPU-only box and exports every symbol include/rten_hip.h declares.
No compute calls here (no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "rten_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rten_hip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from rten_amd import lib
    so = lib.load()
    syms = header_symbols()
    assert len(syms) >= 40
    for s in syms:
        assert hasattr(so, s), f"librten_hip.so does not export {s}"
    # and the python binding declares a prototype for each of them
    assert sorted(lib.PROTOTYPES) == syms


def test_abi_version_and_variants():
    from rten_amd import lib
    so = lib.load()
    assert so.rten_hip_abi_version() == 2
    assert so.rten_hip_num_gemm_variants() == 24


def test_struct_layouts_match_header():
    # sizes computed by the C compiler for the header's structs must match the ctypes mirrors
    import subprocess
    import tempfile
    from rten_amd import lib
    src = r'''
    #include <stdio.h>
    #include "rten_hip.h"
    int main(void) {
        printf("%zu %zu %zu %zu %zu\n", sizeof(rten_hip_gemm_desc), sizeof(rten_hip_gemm_int8_desc),
               sizeof(rten_hip_conv2d_desc), sizeof(rten_hip_conv2d_int8_desc), sizeof(rten_hip_pool2d_desc));
        return 0;
    }'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        sizes = [int(v) for v in subprocess.check_output([os.path.join(d, "t")]).split()]
    assert sizes == [ctypes.sizeof(lib.GemmDesc), ctypes.sizeof(lib.GemmInt8Desc), ctypes.sizeof(lib.Conv2dDesc),
                     ctypes.sizeof(lib.Conv2dInt8Desc), ctypes.sizeof(lib.Pool2dDesc)]


def test_output_size_through_abi_matches_reference_cases():
    import json
    from rten_amd import lib, ops
    so = lib.load()

    class _Ctx:  # calc_output_size_and_padding is host-side shape logic: needs no device
        lib = so

    G = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_literals.json")))
    for c in G["output_size"]["cases"]:
        args = dict(in_size=c.get("in_size", [5, 5]), kernel=c.get("kernel", [3, 3]), strides=c.get("strides", [1, 1]),
                    padding=c.get("padding", [0, 0, 0, 0]), dilations=c.get("dilations", [1, 1]), ceil_mode=c.get("ceil", False))
        if "error" in c:
            with pytest.raises(ops.OpError) as e:
                ops.calc_output_size_and_padding(_Ctx, **args)
            assert e.value == ops.InvalidValue(c["error"])
        else:
            oh, ow, pads = ops.calc_output_size_and_padding(_Ctx, **args)
            assert [oh, ow, pads] == c["expected"]
    with pytest.raises(ops.OpError) as e:  # pooling.rs:1165-1169
        ops.calc_output_size_and_padding(_Ctx, (5, 5), (3, 3), (1, 1), [0, 0])
    assert e.value == ops.InvalidValue("Expected 4 padding values")


def test_no_silent_cpu_fallback():
    """Without a GPU the backend must refuse to run (BackendUnavailable), never compute on the CPU."""
    import torch
    from rten_amd import lib
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the -m gpu tests")
    with pytest.raises(lib.BackendUnavailable):
        lib.Context(0)


def test_product_library_is_not_a_trace_or_ablation_build():
    """The phase stamps (-DRTEN_TRACE: device printf per launch) and ablation switches are for tools/debug and tools/probes; the library that
    travels with the tree must be the plain build."""
    from rten_amd import lib
    blob = open(lib.SO_PATH, "rb").read()
    assert b"[i8 trace]" not in blob and b"[trace] wave" not in blob


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under rten_amd/ (or bench.py's product path) may import it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rten_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".sh")):
                text = open(os.path.join(dirpath, f)).read()
                assert "from oracle" not in text and "import oracle" not in text and "rten_oracle" not in text, f


# ---- the Rust -sys crate (bindings/rten-hip-sys) against the header ------------------------------------------------
RS_PATH = os.path.join(ROOT, "bindings", "rten-hip-sys", "src", "lib.rs")
_RS_SIZES = {"i32": (4, 4), "u32": (4, 4), "f32": (4, 4), "i64": (8, 8), "u64": (8, 8), "u8": (1, 1), "i8": (1, 1), "usize": (8, 8)}


def _rs_structs():
    """{name: [(field, type)]} of every #[repr(C)] struct with fields in the generated Rust file, in file order."""
    text = open(RS_PATH).read()
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^\]]*\)\]\s*)?pub struct (\w+) \{(.*?)\n\}", text, flags=re.S):
        fields = re.findall(r"pub (?:r#)?(\w+): ([^,]+),", m.group(2))
        if fields:
            out[m.group(1)] = fields
    return out


def _rs_layout(ty, structs):
    """(size, align) of a Rust FFI type under repr(C) rules."""
    ty = ty.strip()
    arr = re.match(r"\[(.+); (\d+)\]$", ty)
    if arr:
        s, a = _rs_layout(arr.group(1), structs)
        return s * int(arr.group(2)), a
    if ty.startswith("*"):
        return 8, 8
    if ty in _RS_SIZES:
        return _RS_SIZES[ty]
    offs, size, align = _rs_struct_layout(structs[ty], structs)
    return size, align


def _rs_struct_layout(fields, structs):
    off, align, offs = 0, 1, []
    for _, ty in fields:
        s, a = _rs_layout(ty, structs)
        off = (off + a - 1) // a * a
        offs.append(off)
        off += s
        align = max(align, a)
    return offs, (off + align - 1) // align * align, align


def test_rust_sys_crate_is_generated_from_the_header():
    import subprocess
    import sys
    assert subprocess.call([sys.executable, os.path.join(ROOT, "tools", "gen_rust_bindings.py"), "--check"]) == 0, \
        "bindings/rten-hip-sys/src/lib.rs is stale: run python tools/gen_rust_bindings.py"
    text = open(RS_PATH).read()
    assert "one line per remaining" not in text and "..." not in text  # complete: no elided externs
    fns = re.findall(r"pub fn (rten_hip_[a-z0-9_]+)\(", text)
    assert sorted(fns) == header_symbols() and len(fns) == len(set(fns))
    # every `#define RTEN_HIP_*` integer constant is mirrored with the same value
    hdr = open(os.path.join(ROOT, "include", "rten_hip.h")).read()
    for name, val in re.findall(r"^#define\s+(RTEN_HIP_[A-Z0-9_]+)\s+([0-9]+)u?\b", hdr, flags=re.M):
        assert re.search(rf"pub const {name}: [iu]32 = {val};", text), name


def test_rust_repr_c_field_order_and_offsets_match_the_c_compiler():
    import subprocess
    import tempfile
    structs = _rs_structs()
    assert set(structs) == {"rten_hip_gemm_desc", "rten_hip_gemm_int8_desc", "rten_hip_conv2d_desc", "rten_hip_conv2d_int8_desc",
                            "rten_hip_pool2d_desc", "rten_hip_sdpa_desc"}
    lines = []
    for name, fields in structs.items():
        for f, _ in fields:
            lines.append(f'printf("{name} {f} %zu\\n", offsetof({name}, {f}));')
        lines.append(f'printf("{name} __size %zu\\n", sizeof({name}));')
    src = "#include <stdio.h>\n#include <stddef.h>\n#include \"rten_hip.h\"\nint main(void) {\n" + "\n".join(lines) + "\nreturn 0; }\n"
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        c_layout = {}
        for ln in subprocess.check_output([os.path.join(d, "t")]).decode().splitlines():
            s, f, off = ln.split()
            c_layout.setdefault(s, {})[f] = int(off)
    for name, fields in structs.items():
        offs, size, _ = _rs_struct_layout(fields, structs)
        c = c_layout[name]
        assert size == c["__size"], name
        # the C struct has exactly these fields (a field missing on the Rust side would shift a later offset or the size)
        assert [c[f] for f, _ in fields] == offs, (name, [c[f] for f, _ in fields], offs)
        assert offs == sorted(offs)
