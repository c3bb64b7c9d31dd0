"""CPU: the Rust binding that ships as files (bindings/rust/otters-hip-sys) cannot be compiled here (no rustc in the image),
so it is held to the C header mechanically instead: every `#[repr(C)]` struct of src/lib.rs is laid out by the C rules
(field order, sizes, alignment) and compared with what a C11 compiler reports for include/otters_hip.h
(tests/c/abi_layout), every `extern "C"` item is compared with the header's prototype (name, arity, each argument's and
the return type), and every constant with the header's enum / #define.  A header change that the binding does not follow
fails here."""
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "bindings", "rust", "otters-hip-sys")
PATCH = os.path.join(ROOT, "bindings", "rust", "patch")


def _strip_rust_comments(s: str) -> str:
    s = re.sub(r"/\*.*?\*/", "", s, flags=re.S)
    return re.sub(r"//[^\n]*", "", s)


def _strip_c_comments(s: str) -> str:
    return re.sub(r"/\*.*?\*/", "", s, flags=re.S)


def rust_source():
    return _strip_rust_comments(open(os.path.join(CRATE, "src", "lib.rs")).read())


def c_header():
    return _strip_c_comments(open(os.path.join(ROOT, "include", "otters_hip.h")).read())


# size, alignment of the Rust types the binding may use in a #[repr(C)] struct (LP64)
RUST_LAYOUT = {"u8": (1, 1), "u32": (4, 4), "i32": (4, 4), "f32": (4, 4), "u64": (8, 8), "i64": (8, 8), "f64": (8, 8), "c_int": (4, 4)}


def rust_structs(src):
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[[^\]]*\]\s*)*pub struct (\w+)\s*\{(.*?)\}", src, flags=re.S):
        fields = []
        for f in m.group(2).split(","):
            f = f.strip()
            if not f:
                continue
            fm = re.match(r"(?:pub\s+)?(\w+)\s*:\s*(.+)$", f, flags=re.S)
            assert fm, f
            fields.append((fm.group(1), " ".join(fm.group(2).split())))
        out[m.group(1)] = fields
    return out


def c_layout_of(fields):
    """repr(C): fields in order, each aligned to its own alignment, the struct padded to its largest alignment."""
    off, align_max, offs = 0, 1, {}
    for name, ty in fields:
        size, align = (8, 8) if ty.startswith("*") else RUST_LAYOUT[ty]
        off = (off + align - 1) // align * align
        offs[name] = off
        off += size
        align_max = max(align_max, align)
    return offs, (off + align_max - 1) // align_max * align_max


def abi_layout():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "c"), "-s"])
    out = subprocess.run([os.path.join(ROOT, "tests", "c", "abi_layout")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    return json.loads(out.stdout)


def test_crate_files_exist_and_name_the_library():
    for rel in ("Cargo.toml", "build.rs", "src/lib.rs"):
        assert os.path.exists(os.path.join(CRATE, rel)), rel
    cargo = open(os.path.join(CRATE, "Cargo.toml")).read()
    assert 'name = "otters-hip-sys"' in cargo and 'links = "otters_hip"' in cargo
    build = open(os.path.join(CRATE, "build.rs")).read()
    assert "rustc-link-lib=dylib=otters_hip" in build and "OTTERS_HIP_LIB_DIR" in build
    for rel in ("README.md", "Cargo.toml.patch", "vec_hip.rs", "meta_hip.rs"):
        assert os.path.exists(os.path.join(PATCH, rel)), rel


def test_repr_c_structs_match_the_c_compilers_layout():
    lay = abi_layout()
    structs = rust_structs(rust_source())
    for cname in ("ott_hit", "ott_query_desc", "ott_stats", "ott_leaf"):
        assert cname in structs, f"{cname}: no #[repr(C)] struct of that name in lib.rs"
        offs, size = c_layout_of(structs[cname])
        want = {k.split(".", 1)[1]: v for k, v in lay["offsetof"].items() if k.startswith(cname + ".")}
        assert [n for n, _ in structs[cname]] == sorted(want, key=lambda n: want[n]), (cname, structs[cname])  # same fields, same ORDER
        assert offs == want, (cname, offs, want)
        assert size == lay["sizeof"][cname], (cname, size)
    # the const assertion block of lib.rs restates the same sizes
    src = rust_source()
    for cname in ("ott_hit", "ott_query_desc", "ott_stats", "ott_leaf"):
        m = re.search(r"size_of::<%s>\(\)\s*==\s*(\d+)" % cname, src)
        assert m and int(m.group(1)) == lay["sizeof"][cname], cname
    # opaque handles carry no layout
    for cname in ("ott_store", "ott_comm"):
        assert structs[cname] == [("_opaque", "[u8; 0]")]


C2RUST = {
    "void": None, "int": "c_int", "int*": "*mut c_int", "const int*": "*const c_int", "uint32_t": "u32", "uint64_t": "u64", "int64_t": "i64", "float": "f32",
    "const char*": "*const c_char", "const float*": "*const f32", "float*": "*mut f32", "const void*": "*const c_void", "void*": "*mut c_void",
    "const uint64_t*": "*const u64", "uint64_t*": "*mut u64", "uint32_t*": "*mut u32",
    "ott_store*": "*mut ott_store", "const ott_store*": "*const ott_store", "ott_store**": "*mut *mut ott_store",
    "ott_comm*": "*mut ott_comm", "const ott_comm*": "*const ott_comm", "ott_comm**": "*mut *mut ott_comm",
    "const ott_query_desc*": "*const ott_query_desc", "ott_hit*": "*mut ott_hit", "ott_stats*": "*mut ott_stats",
    "const ott_leaf*": "*const ott_leaf", "ott_allgather_fn": "ott_allgather_fn",
}


def c_prototypes():
    protos = {}
    for ret, name, args in re.findall(r"^\s*((?:const\s+)?[a-z_0-9]+\s*\**)\s*(ott_[a-z0-9_]+)\s*\(([^;{]*)\)\s*;", c_header(), flags=re.M):
        ret = re.sub(r"\s*\*", "*", " ".join(ret.split()))
        alist = []
        args = " ".join(args.split())
        if args != "void":
            for a in args.split(","):
                a = re.sub(r"/\*.*?\*/", "", a).strip()
                m = re.match(r"(.*?)(\w+)$", a)  # type, then the parameter name
                ty = re.sub(r"\s*\*", "*", " ".join(m.group(1).split()))
                alist.append(ty)
        protos[name] = (ret, alist)
    return protos


def rust_externs():
    src = rust_source()
    block = re.search(r'extern "C"\s*\{(.*?)\n\}', src, flags=re.S).group(1)
    out = {}
    for name, args, ret in re.findall(r"pub fn (\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        alist = []
        for a in args.split(","):
            a = a.strip()
            if a:
                alist.append(" ".join(a.split(":", 1)[1].split()))
        out[name] = (" ".join(ret.split()) if ret else None, alist)
    return out


def test_every_header_function_is_bound_with_the_same_signature():
    c, r = c_prototypes(), rust_externs()
    assert len(c) >= 40 and sorted(c) == sorted(r), (sorted(set(c) ^ set(r)))
    for name, (ret, args) in c.items():
        rret, rargs = r[name]
        assert C2RUST[ret] == rret, (name, ret, rret)
        assert len(args) == len(rargs), (name, args, rargs)
        for ca, ra in zip(args, rargs):
            assert C2RUST[ca] == ra, (name, ca, ra)
    src = rust_source()
    # the callback type: int (*)(void* user, const void* send, void* recv, uint64_t bytes)
    m = re.search(r"pub type ott_allgather_fn\s*=\s*Option<unsafe extern \"C\" fn\((.*?)\)\s*->\s*c_int>", src, flags=re.S)
    assert m
    assert [" ".join(a.split(":")[1].split()) for a in m.group(1).split(",")] == ["*mut c_void", "*const c_void", "*mut c_void", "u64"]


def test_constants_match_the_headers_enums_and_defines():
    hdr = c_header()
    want = {}
    for body in re.findall(r"typedef enum\s*\{(.*?)\}", hdr, flags=re.S):
        for name, val in re.findall(r"(OTT_[A-Z0-9_]+)\s*=\s*(-?\d+)", body):
            want[name] = int(val)
    for name, val in re.findall(r"#define\s+(OTT_[A-Z0-9_]+)\s+(\d+)", hdr):
        want[name] = int(val)
    src = rust_source()
    got = {n: int(v) for n, v in re.findall(r"pub const (OTT_[A-Z0-9_]+)\s*:\s*\w+\s*=\s*(-?\d+)\s*;", src)}
    assert len(want) >= 35 and got == want, {k: (want.get(k), got.get(k)) for k in set(want) ^ set(got) | {k for k in want if got.get(k) != want[k]}}


def test_patch_files_call_only_bound_functions():
    """vec_hip.rs / meta_hip.rs: every `sys::ott_*` they call exists in the sys crate with that arity, every `sys::OTT_*`
    constant and `sys::ott_*` type they name is declared there, and the struct literals name exactly the struct's fields."""
    r = rust_externs()
    src_sys = rust_source()
    structs = rust_structs(src_sys)
    consts = set(re.findall(r"pub const (OTT_[A-Z0-9_]+)", src_sys))
    helpers = set(re.findall(r"pub fn (\w+)\s*\(", src_sys)) - set(r)
    for fn in ("vec_hip.rs", "meta_hip.rs"):
        src = _strip_rust_comments(open(os.path.join(PATCH, fn)).read())
        for name in set(re.findall(r"sys::(OTT_[A-Z0-9_]+)", src)):
            assert name in consts, (fn, name)
        for name in set(re.findall(r"sys::(ott_[a-z0-9_]+|last_error|check)\b", src)):
            assert name in r or name in structs or name in helpers, (fn, name)
        # call arity: sys::ott_xxx( ... ) with balanced parentheses
        for m in re.finditer(r"sys::(ott_[a-z0-9_]+)\s*\(", src):
            name = m.group(1)
            if name not in r:
                continue
            depth, i, args, cur = 1, m.end(), [], ""
            while depth:
                ch = src[i]
                if ch in "([{":
                    depth += 1
                elif ch in ")]}":
                    depth -= 1
                    if depth == 0:
                        break
                if ch == "," and depth == 1:
                    args.append(cur)
                    cur = ""
                else:
                    cur += ch
                i += 1
            if cur.strip():
                args.append(cur)
            assert len(args) == len(r[name][1]), (fn, name, len(args), len(r[name][1]))
        # struct literals (fields split at depth-0 commas: initialisers contain calls and closures)
        for m in re.finditer(r"sys::(ott_query_desc|ott_hit|ott_leaf)\s*\{", src):
            depth, i, parts, cur = 1, m.end(), [], ""
            while depth:
                ch = src[i]
                if ch in "([{":
                    depth += 1
                elif ch in ")]}":
                    depth -= 1
                    if depth == 0:
                        break
                if ch == "," and depth == 1:
                    parts.append(cur)
                    cur = ""
                else:
                    cur += ch
                i += 1
            parts.append(cur)
            names = [f.split(":")[0].strip() for f in parts if f.strip()]
            assert sorted(names) == sorted(n for n, _ in structs[m.group(1)]), (fn, m.group(1), names)
