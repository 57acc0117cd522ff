"""The Rust binding under rust/ cannot be compiled in this image (no rustc): at least hold its `extern "C"`
blocks to the header they bind.  Every function it declares must exist in include/tfhe_hip.h with the same number of
parameters and the same shape per parameter (pointer / const / integer width / double) and the same return type --
the drift a Rust compiler would NOT catch either (an FFI declaration is taken on faith)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RUST_TO_C = {
    "c_int": "int", "i32": "int", "u32": "uint32_t", "u8": "uint8_t", "f64": "double", "usize": "size_t",
    "c_char": "char", "c_void": "void", "u64": "uint64_t",
}


def _rust_sources():
    out = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rust")):
        out += [os.path.join(dirpath, f) for f in files if f.endswith(".rs")]
    return sorted(out)


def _rust_decls():
    text = "\n".join(open(f).read() for f in _rust_sources())
    decls = {}
    for block in re.findall(r'extern "C" \{(.*?)\n\}', text, flags=re.S):
        block = re.sub(r"//[^\n]*", "", block)
        for m in re.finditer(r"fn\s+(\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
            name, args, ret = m.group(1), m.group(2), (m.group(3) or "()").strip()
            params = [a.split(":", 1)[1].strip() for a in args.split(",") if ":" in a]
            decls[name] = (params, ret)
    return decls


def _c_decls():
    text = open(os.path.join(ROOT, "include", "tfhe_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(tfhe_hip_\w+)\s*\(([^;{]*?)\)\s*;", text):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), m.group(3)
        params = [] if args.strip() in ("", "void") else [" ".join(a.split()) for a in args.split(",")]
        decls[name] = (params, ret)
    return decls


def _shape_rust(t):
    """(pointer depth, pointee const at the outermost level, base type in C spelling)"""
    depth, const = 0, False
    t = t.strip()
    while t.startswith("*"):
        m = re.match(r"\*(const|mut)\s+(.*)", t)
        if depth == 0:
            const = m.group(1) == "const"
        depth += 1
        t = m.group(2).strip()
    base = RUST_TO_C.get(t, "struct" if t and t[0].isupper() else t)
    return depth, const, base


def _shape_c(t, has_name=True):
    t = t.strip()
    arr = re.search(r"\[[^\]]*\]\s*$", t)  # `const uint8_t key[32]` is `const uint8_t *key`
    if arr:
        t = t[: arr.start()].strip()
    depth = t.count("*") + (1 if arr else 0)
    const = bool(re.search(r"\bconst\b", t.split("*")[0])) if depth else False
    words = re.sub(r"\bconst\b|\*", " ", t).split()
    if has_name and len(words) > 1:
        words = words[:-1]  # the parameter's name
    base = " ".join(words)
    if base.startswith("tfhe_hip_") or base.startswith("struct"):
        base = "struct"
    base = {"unsigned int": "uint32_t", "unsigned char": "uint8_t", "unsigned long": "size_t"}.get(base, base)
    return depth, const, base


def test_integration_md_shows_the_files_it_names():
    """INTEGRATION.md includes the binding by reference: every rust/ path it names exists, and the FFI excerpt it prints
    is the head of rust/src/bootstrap/hip.rs word for word."""
    import re as _re

    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for path in set(_re.findall(r"\((rust/[^)]+)\)", md)):
        assert os.path.exists(os.path.join(ROOT, path)), path
    hip = open(os.path.join(ROOT, "rust", "src", "bootstrap", "hip.rs")).read()
    m = _re.search(r"Its head:\n\n```rust\n(.*?)// \.\.\. \(constants", md, flags=_re.S)
    assert m and m.group(1) in hip


def test_rust_ffi_block_matches_the_header():
    rust, c = _rust_decls(), _c_decls()
    assert len(_rust_sources()) >= 3
    assert len(rust) >= 20, sorted(rust)
    assert "tfhe_hip_pool_batch_gate_dev" in rust and "tfhe_hip_pool_synchronize" in rust
    problems = []
    for name, (rparams, rret) in sorted(rust.items()):
        if name not in c:
            problems.append(f"{name}: not declared in include/tfhe_hip.h")
            continue
        cparams, cret = c[name]
        if len(rparams) != len(cparams):
            problems.append(f"{name}: {len(rparams)} parameters in the binding, {len(cparams)} in the header")
            continue
        for i, (rp, cp) in enumerate(zip(rparams, cparams)):
            if _shape_rust(rp) != _shape_c(cp):
                problems.append(f"{name} parameter {i}: `{rp}` vs `{cp}` ({_shape_rust(rp)} vs {_shape_c(cp)})")
        rshape = (0, False, "void") if rret == "()" else _shape_rust(rret)
        if rshape != _shape_c(cret, has_name=False):
            problems.append(f"{name} returns `{rret}` vs `{cret}`")
    assert not problems, "\n".join(problems)


def test_rust_ffi_symbols_are_exported():
    so = os.path.join(ROOT, "rs-tfhe_amd", "libtfhe_hip.so")
    if not os.path.exists(so):
        import pytest

        pytest.skip("library not built")
    import subprocess

    syms = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in syms.splitlines() if line.strip()}
    missing = [n for n in _rust_decls() if n not in exported]
    assert not missing, missing
