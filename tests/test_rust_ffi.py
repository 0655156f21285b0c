"""The Rust side of the drop-in (rust/zbra-hip-sys, rust/patches/zbra-core.patch) cannot be compiled
here (no Rust toolchain), so its declarations are checked against include/zbra_hip.h by parsing
both: every function with the same name, arity and argument / return types, every struct with the
same fields in the same order and of the same sizes, every constant with the same value.  The patch
must apply cleanly to the reference's sources where those are present."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "zbra_hip.h")
LIB_RS = os.path.join(ROOT, "rust", "zbra-hip-sys", "src", "lib.rs")
PATCH = os.path.join(ROOT, "rust", "patches", "zbra-core.patch")

# C type -> Rust type, after whitespace normalisation
TYPES = {
    "int": "c_int", "int *": "*mut c_int", "size_t": "usize", "size_t *": "*mut usize", "int64_t": "i64", "uint32_t": "u32",
    "uint64_t": "u64", "const char *": "*const c_char", "void *": "*mut c_void", "const void *": "*const c_void",
    "void *const *": "*const *mut c_void", "const uint8_t *": "*const u8", "uint8_t *": "*mut u8", "const int64_t *": "*const i64",
    "int64_t *": "*mut i64", "const uint64_t *": "*const u64", "uint64_t *": "*mut u64", "const uint32_t *": "*const u32",
    "uint32_t *": "*mut u32", "int32_t": "i32",
}
for _s in ("zbra_dev_result", "zbra_array_desc", "zbra_encode_desc", "zbra_enum_desc", "zbra_file_array", "zbra_file_info", "zbra_file_job"):
    TYPES[f"{_s} *"] = f"*mut {_s}"
    TYPES[f"const {_s} *"] = f"*const {_s}"
SIZES = {"u32": 4, "u64": 8, "i64": 8, "*const u8": 8, "*mut u8": 8, "*mut c_void": 8, "*const c_void": 8, "*const u32": 8, "*mut u32": 8,
         "*mut u64": 8, "i32": 4, "*const zbra_file_array": 8, "*const *mut c_void": 8, "*mut c_int": 8}


def _c_text():
    return re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)


def _norm_c_type(t):
    t = re.sub(r"\s+", " ", t.strip())
    t = re.sub(r"\s*\*\s*", " *", t)          # "uint8_t* x" / "uint8_t *x" -> "uint8_t *"
    t = t.replace("* *", "**").replace(" *const *", " *const *")
    return t.strip()


def _c_functions():
    out = {}
    for ret, name, args in re.findall(r"([\w\s\*]+?)\b(zbra_hip_\w+)\s*\(([^)]*)\)\s*;", _c_text()):
        arg_types = []
        if args.strip() != "void":
            for a in args.split(","):
                a = a.strip()
                m = re.match(r"(.*?)(\w+)$", a)          # the last identifier is the parameter name
                arg_types.append(_norm_c_type(m.group(1)))
        out[name] = (_norm_c_type(ret), arg_types)
    return out


def _rs_text():
    return re.sub(r"//.*", "", open(LIB_RS).read())


def _rs_functions():
    block = re.search(r'extern "C" \{(.*?)\n    \}', _rs_text(), flags=re.S).group(1)
    out = {}
    for name, args, ret in re.findall(r"pub fn (zbra_hip_\w+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", block):
        arg_types = [re.sub(r"\s+", " ", a.split(":", 1)[1].strip()) for a in args.split(",") if a.strip()]
        out[name] = (re.sub(r"\s+", " ", (ret or "()").strip()), arg_types)
    return out


def test_every_function_of_the_header_is_declared_with_the_same_signature():
    c, rs = _c_functions(), _rs_functions()
    assert len(c) >= 29 and sorted(c) == sorted(rs)
    for name, (ret, args) in c.items():
        want = (TYPES[ret], [TYPES[a] for a in args])
        assert rs[name] == want, (name, rs[name], want)


def _c_structs():
    out = {}
    for body, name in re.findall(r"typedef struct \w+ \{(.*?)\} (\w+);", _c_text(), flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            first, *more = [x.strip() for x in decl.split(",")]   # "uint64_t a, b" declares two fields
            m = re.match(r"(.*[\s\*])(\w+)$", first)
            t = _norm_c_type(m.group(1))
            for f in [m.group(2)] + more:
                fields.append((f, TYPES[t]))
        out[name] = fields
    return out


def _rs_structs():
    out = {}
    for name, body in re.findall(r"pub struct (zbra_\w+) \{(.*?)\n    \}", _rs_text(), flags=re.S):
        out[name] = [(f, re.sub(r"\s+", " ", t.strip())) for f, t in re.findall(r"pub (\w+):\s*([^,\n]+),", body)]
    return out


def test_structs_have_the_same_fields_in_the_same_order():
    c, rs = _c_structs(), _rs_structs()
    assert sorted(c) == ["zbra_array_desc", "zbra_dev_result", "zbra_encode_desc", "zbra_enum_desc", "zbra_file_array", "zbra_file_info", "zbra_file_job"]
    assert sorted(rs) == sorted(c)
    for name in c:
        assert rs[name] == c[name], name
    # #[repr(C)] on every one of them, and the sizes the C compiler gives (no padding surprises:
    # every 8-byte field sits at a multiple of 8)
    text = open(LIB_RS).read()
    for name in c:
        assert re.search(r"#\[repr\(C\)\]\s*(#\[derive\([^\]]*\)\]\s*)?pub struct " + name + r"\b", text), name
    sizes = {n: sum(SIZES[t] for _, t in f) for n, f in rs.items()}
    src = '#include <stdio.h>\n#include "zbra_hip.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(zbra_array_desc), sizeof(zbra_dev_result), sizeof(zbra_encode_desc), sizeof(zbra_enum_desc), sizeof(zbra_file_array), sizeof(zbra_file_info), sizeof(zbra_file_job)); return 0;}\n'
    exe = "/tmp/zbra_sizes_check"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-x", "c", "-", "-o", exe], input=src, text=True, check=True)
    got = [int(x) for x in subprocess.check_output([exe], text=True).split()]
    assert got == [sizes[n] for n in sorted(c)], (got, sizes)


def test_constants_match():
    c = _c_text()
    rs = _rs_text()
    enums = dict(re.findall(r"\b(ZBRA_(?:OK|ERR_\w+|ELEM_\w+|ROLE_\w+))\s*=\s*(\d+)", c))
    assert len(enums) >= 19
    for name, value in enums.items():
        m = re.search(r"pub const " + name + r": \w+ = (\d+);", rs)
        assert m and m.group(1) == value, name
    assert re.search(r"#define ZBRA_HIP_NOALLOC_MAX_ARRAYS 4096", c) and re.search(r"ZBRA_HIP_NOALLOC_MAX_ARRAYS: usize = 4096;", rs)
    assert re.search(r"#define ZBRA_HIP_NOALLOC_MAX_TILES \(1u << 20\)", c) and re.search(r"ZBRA_HIP_NOALLOC_MAX_TILES: usize = 1 << 20;", rs)


def test_error_messages_are_the_reference_strings():
    rs = open(LIB_RS).read()
    assert 'Error::ShortHeader => f.write_str("Invalid compressed data length")' in rs       # compression.rs:302
    assert 'Error::ShortPacked => f.write_str("Insufficient data for packed array")' in rs   # compression.rs:314
    from zbra_amd._lib import strerror
    assert strerror(1) == "Invalid compressed data length" and strerror(2) == "Insufficient data for packed array"


def test_patch_applies_to_the_reference():
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "zbra-core", "src")):
        pytest.skip("the reference's sources are not on this machine")
    if shutil.which("patch") is None:
        pytest.skip("no patch(1)")
    tmp = "/tmp/zbra_patch_check"
    shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(os.path.join(tmp, "zbra-core", "src"))
    for rel in ("zbra-core/Cargo.toml", "zbra-core/src/compression.rs", "zbra-core/src/binary.rs"):
        shutil.copy(os.path.join(ref, rel), os.path.join(tmp, rel))
    p = subprocess.run(["patch", "-p1", "--no-backup-if-mismatch", "-i", PATCH], cwd=tmp, capture_output=True, text=True)
    assert p.returncode == 0 and "FAILED" not in p.stdout and "fuzz" not in p.stdout, p.stdout + p.stderr
    patched = open(os.path.join(tmp, "zbra-core", "src", "binary.rs")).read()
    for needle in ("zbra_hip_sys::encode_frames", "zbra_hip_decode_batch", "fn register_for_read", "fn read_f64_array_compressed"):
        assert needle in patched
    comp = open(os.path.join(tmp, "zbra-core", "src", "compression.rs")).read()
    assert "zbra_hip_sys::compress_int_array(values)" in comp and "zbra_hip_sys::decompress_int_array(data, count)" in comp
    assert "zbra-hip-sys" in open(os.path.join(tmp, "zbra-core", "Cargo.toml")).read()
    shutil.rmtree(tmp, ignore_errors=True)
