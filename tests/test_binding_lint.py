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


# ---- the Rust-side API: signatures, paths and the crate patch (no rustc here: what a reviewer would check by eye) --------
def _gates_hip_fns():
    import importlib.util

    spec = importlib.util.spec_from_file_location("mrs", os.path.join(ROOT, "tests", "golden", "make_reference_signatures.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    text = open(os.path.join(ROOT, "rust", "src", "gates_hip.rs")).read()
    fns = {}
    for m in re.finditer(r"^pub fn (\w+)\s*\(", text, flags=re.M):
        fns[m.group(1)] = mod.free_fn_signature(text, m.group(1))
    return fns, mod


def test_gates_hip_keeps_the_reference_signatures():
    """Every `pub fn X_hip` in rust/src/gates_hip.rs that has a namesake in the reference (src/gates.rs:352-547 batch_*,
    :293-312 mux / mux_naive, src/trgsw.rs:289 batch_blind_rotate) takes the namesake's parameter types in its order and
    returns its type -- no extra engine argument (the round-4 verdict's finding) -- and all of them are there.  The table
    tests/golden/reference_signatures.json was read from the reference by tests/golden/make_reference_signatures.py;
    where the reference tree is present the table itself is re-derived and compared."""
    import json

    fns, mod = _gates_hip_fns()
    table = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_signatures.json")))
    assert set(table) <= set(fns), sorted(set(table) - set(fns))
    for name, want in table.items():
        got = fns[name]
        assert got is not None, name
        assert got["params"] == want["params"], (name, got["params"], want["params"])
        assert got["ret"] == want["ret"], (name, got["ret"], want["ret"])
    if os.path.isdir("/root/reference/src"):
        fresh = mod.collect()
        assert fresh == table, "tests/golden/reference_signatures.json is stale: re-run make_reference_signatures.py"
    text = open(os.path.join(ROOT, "rust", "src", "gates_hip.rs")).read()
    assert not re.search(r"//[^\n]*\bsame\b", text), "a comment standing in for code"
    assert "HipEngine" not in re.sub(r"//[^\n]*", "", text), "gates_hip.rs must not ask its caller for an engine"


def _rust_uses(path):
    """`use crate::a::b::{C, d::E};` / `use rs_tfhe::{a, b};` -> `a::b::C` style item paths (one level of braces, `self`
    dropped), plus the `crate::x::y` paths written inline in expressions."""
    text = re.sub(r"//[^\n]*", "", open(path).read())
    out = set()
    for m in re.finditer(r"\buse\s+(?:crate|rs_tfhe)::([^;]+);", text):
        body = re.sub(r"\s+", "", m.group(1))
        g = re.match(r"(?:(.*?)::)?\{(.*)\}$", body)
        if g:
            prefix = g.group(1) + "::" if g.group(1) else ""
            for item in g.group(2).split(","):
                if item and item != "self":
                    out.add(prefix + item)
            if g.group(1):
                out.add(g.group(1))
        else:
            out.add(body)
    for m in re.finditer(r"\b(?:crate|rs_tfhe)::((?:\w+::)+\w+)", text):
        out.add(m.group(1))
    return out


def test_rust_paths_resolve_in_the_patched_crate():
    """Every `crate::...` / `rs_tfhe::...` path the files under rust/ name resolves in the crate AS PATCHED: the module
    exists (in the reference tree or among the files rust/apply.sh copies) and declares the item `pub`.  Needs the
    reference tree (build container only)."""
    import pytest

    ref = "/root/reference/src"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present")
    ours = {"bootstrap/hip.rs": "rust/src/bootstrap/hip.rs", "gates_hip.rs": "rust/src/gates_hip.rs",
            "proxy_reenc_hip.rs": "rust/src/proxy_reenc_hip.rs"}
    patch_blocks = open(os.path.join(ROOT, "rust", "patches", "rs-tfhe-hip.patch")).read().split("diff -ruN")

    def added_by_patch(rel):
        lines = []
        for blk in patch_blocks:
            if f"b/src/{rel}" in blk.split("\n")[0]:
                lines += [ln[1:] for ln in blk.split("\n") if ln.startswith("+") and not ln.startswith("+++")]
        return "\n".join(lines)

    def module_text(parts):
        """Source of module a::b: file a/b.rs or a/b/mod.rs (patched lines included), or an inline `pub mod b {` of a."""
        rel = "/".join(parts)
        for cand in (rel + ".rs", rel + "/mod.rs"):
            if cand in ours:
                return open(os.path.join(ROOT, ours[cand])).read()
            if os.path.exists(os.path.join(ref, cand)):
                return open(os.path.join(ref, cand)).read() + "\n" + added_by_patch(cand)
        if len(parts) > 1:  # params::tlwe_lv0 is `pub mod tlwe_lv0 {` inside params.rs (re-exported by `pub use`)
            parent = module_text(parts[:-1])
            if parent and re.search(r"pub mod " + parts[-1] + r"\b", parent):
                return parent
        return None

    problems = []
    for f in _rust_sources() + [os.path.join(ROOT, "rust", "tests", "hip_gates.rs")]:
        for path in sorted(_rust_uses(f)):
            parts = path.split("::")
            k = len(parts)  # the longest prefix that names a module; what follows it is an item of that module
            while k > 0 and module_text(parts[:k]) is None:
                k -= 1
            if k == len(parts):
                continue  # the path names a module
            if k == 0:
                problems.append(f"{os.path.relpath(f, ROOT)}: no module for `{path}`")
                continue
            text, item = module_text(parts[:k]), parts[k]  # (parts after the item: associated functions / constants)
            if not re.search(r"pub\s+(?:unsafe\s+)?(?:fn|struct|enum|trait|type|const|static|mod)\s+" + item + r"\b", text) \
                    and not re.search(r"pub use [^;]*\b" + item + r"\b", text):
                problems.append(f"{os.path.relpath(f, ROOT)}: `{path}`: no `pub` item `{item}` in module `{'::'.join(parts[:k])}`")
    assert not problems, "\n".join(problems)


def test_crate_patch_applies_to_the_reference(tmp_path):
    """rust/apply.sh on a scratch copy of the reference's touched files: the patch applies without fuzz or rejects, the
    `hip` feature, both modules, the accessor and the cfg routes are in place.  Needs the reference tree."""
    import shutil
    import subprocess

    import pytest

    if not os.path.isdir("/root/reference/src") or not shutil.which("patch"):
        pytest.skip("reference tree or patch(1) not present")
    crate = tmp_path / "rs-tfhe"
    for rel in ("Cargo.toml", "build.rs", "src/lib.rs", "src/gates.rs", "src/trgsw.rs", "src/bootstrap/mod.rs"):
        (crate / rel).parent.mkdir(parents=True, exist_ok=True)
        shutil.copy(os.path.join("/root/reference", rel), crate / rel)
    p = subprocess.run(["sh", os.path.join(ROOT, "rust", "apply.sh"), str(crate)], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "fuzz" not in p.stdout and "FAILED" not in p.stdout and not list(crate.rglob("*.rej")), p.stdout
    assert 'hip = []' in (crate / "Cargo.toml").read_text() and "CARGO_FEATURE_HIP" in (crate / "build.rs").read_text()
    assert "pub mod gates_hip;" in (crate / "src/lib.rs").read_text()
    assert "pub mod hip;" in (crate / "src/bootstrap/mod.rs").read_text() and "hip::HipBootstrap::new()" in (crate / "src/bootstrap/mod.rs").read_text()
    gates = (crate / "src/gates.rs").read_text()
    for name in ("nand", "and", "or", "xor", "nor", "xnor"):
        assert f"return crate::gates_hip::batch_{name}_hip(inputs, cloud_key);" in gates
    trgsw = (crate / "src/trgsw.rs").read_text()
    assert "pub fn rows(&self)" in trgsw and "crate::gates_hip::batch_blind_rotate_hip(srcs, cloud_key)" in trgsw
    for rel in ("src/bootstrap/hip.rs", "src/gates_hip.rs", "src/proxy_reenc_hip.rs", "tests/hip_gates.rs"):
        assert (crate / rel).exists(), rel


RUST_STD_METHODS = {  # methods of std types the binding calls (slices, iterators, Option/Result, Mutex, OnceLock, CStr, str, pointers)
    "all", "as_mut_ptr", "as_ptr", "as_ref", "borrow", "borrow_mut", "chunks_exact", "clone", "collect", "copy_from_slice", "drain",
    "enumerate", "expect", "extend_from_slice", "fill_bytes", "filter", "find", "for_each", "get_or_init", "is_null", "is_power_of_two",
    "iter", "iter_mut", "len", "lock", "map", "map_or", "min_by_key", "parse", "pop", "position", "push", "remove", "split", "to_str",
    "to_string_lossy", "trailing_zeros", "trim", "unwrap", "unwrap_or", "with", "wrapping_mul", "zip", "to_bits", "get", "is_empty",
    "par_iter", "take", "elapsed", "as_secs_f64",  # (rayon's ParallelIterator, Iterator::take, Instant / Duration)
}


def test_rust_methods_and_fields_exist():
    """A typo guard in place of the compiler: every `.method(` the files under rust/ call is a `fn` of the crate (the
    reference tree or the binding itself) or a known std method, and every `.field` they read is a field of some struct of
    the crate or the binding.  Needs the reference tree."""
    import glob

    import pytest

    if not os.path.isdir("/root/reference/src"):
        pytest.skip("reference tree not present")
    ours = _rust_sources() + [os.path.join(ROOT, "rust", "tests", "hip_gates.rs")]
    crate = glob.glob("/root/reference/src/**/*.rs", recursive=True)
    fns, fields = set(), set()
    for f in ours + crate:
        text = open(f).read()
        fns |= set(re.findall(r"\bfn\s+(\w+)", text))
        for body in re.findall(r"\bstruct\s+\w+[^{;]*\{(.*?)\n\}", text, flags=re.S):
            fields |= set(re.findall(r"(?:pub\s+)?(\w+)\s*:", body))
    problems = []
    for f in ours:
        text = re.sub(r"//[^\n]*", "", open(f).read())
        text = re.sub(r'"(?:[^"\\]|\\.)*"', '""', text)  # string literals out of the way
        for name in sorted(set(re.findall(r"\.(\w+)\s*\(", text)) - fns - RUST_STD_METHODS):
            problems.append(f"{os.path.relpath(f, ROOT)}: method `.{name}()` is neither a fn of the crate nor a known std method")
        for name in sorted(set(re.findall(r"\.([a-z_]\w*)\b(?!\s*[(:!])", text)) - fields - fns - {"unsafe", "gen", "await"}):
            if not name[0].isdigit():
                problems.append(f"{os.path.relpath(f, ROOT)}: field `.{name}` is not a field of any struct of the crate or the binding")
    assert not problems, "\n".join(problems)


def test_rust_repr_c_structs_match_the_header():
    """Every `#[repr(C)] struct` with fields under rust/ against the C struct of the same fields in include/tfhe_hip.h: same
    number of fields, same names in the same order, same widths.  (A by-value struct with a swapped or missing field is the
    other thing rustc takes on faith.)  Opaque handles (`_private: [u8; 0]`) are skipped."""
    c_text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "tfhe_hip.h")).read(), flags=re.S)
    c_structs = {}
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\} (\w+);", c_text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            ctype, names = decl.rsplit(" ", 1)[0], decl.rsplit(" ", 1)[1]
            # `double a, b;` declares several
            parts = [p.strip() for p in decl[len(decl.split(" ")[0]) + 1:].split(",")] if "," in decl else [names]
            base = decl.split(" ")[0] if "," in decl else ctype
            for nme in parts:
                fields.append((nme, {"int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "double": "f64", "int": "c_int",
                                     "size_t": "usize"}.get(base, base)))
        c_structs[m.group(3)] = fields
    rust_text = "\n".join(open(f).read() for f in _rust_sources())
    checked = 0
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:pub\s+)?struct (\w+)\s*\{(.*?)\}", rust_text, flags=re.S):
        name, body = m.group(1), m.group(2)
        fields = [(f.split(":")[0].strip().replace("pub ", ""), f.split(":")[1].strip()) for f in body.split(",") if ":" in f]
        if [f for f in fields if f[0] == "_private"]:
            continue
        # TfheHipParams -> tfhe_hip_params
        cname = re.sub(r"(?<!^)(?=[A-Z])", "_", name).lower()
        assert cname in c_structs, f"{name}: no struct {cname} in include/tfhe_hip.h"
        want = c_structs[cname]
        assert [f[0] for f in fields] == [w[0] for w in want], (name, fields, want)
        for (fn, ft), (_, wt) in zip(fields, want):
            assert ft == wt or {ft, wt} <= {"i32", "c_int"}, (name, fn, ft, wt)
        checked += 1
    assert checked >= 1


def test_rust_sources_are_lexically_well_formed():
    """No compiler here, so at least: in every .rs file under rust/ the brackets ( [ { balance and nest properly outside of
    comments, string literals and character literals (lifetimes like `'a` are not character literals), and every
    statement-level `let` / `fn` line that opens a block closes it -- the class of typo that would stop `cargo build` at
    its first line of output."""
    pairs = {")": "(", "]": "[", "}": "{"}
    for path in _rust_sources() + [os.path.join(ROOT, "rust", "tests", "hip_gates.rs")]:
        text = open(path).read()
        stack, i, n, line = [], 0, len(text), 1
        while i < n:
            c = text[i]
            if c == "\n":
                line += 1
            if text.startswith("//", i):
                i = text.find("\n", i)
                i = n if i < 0 else i
                continue
            if text.startswith("/*", i):
                j = text.find("*/", i + 2)
                assert j >= 0, f"{path}:{line}: unterminated block comment"
                line += text.count("\n", i, j)
                i = j + 2
                continue
            if c == '"' or (c == "r" and text.startswith('r#"', i)) or (c == "b" and text.startswith('b"', i)):
                if c == "r":
                    j = text.find('"#', i + 3)
                    assert j >= 0, f"{path}:{line}: unterminated raw string"
                    line += text.count("\n", i, j)
                    i = j + 2
                    continue
                j = i + (2 if c == "b" else 1)
                while j < n and text[j] != '"':
                    j += 2 if text[j] == "\\" else 1
                assert j < n, f"{path}:{line}: unterminated string literal"
                line += text.count("\n", i, j)
                i = j + 1
                continue
            if c == "'":
                m = re.match(r"'(\\.|[^\\'])'", text[i:i + 4])
                if m:  # a character literal
                    i += m.end()
                    continue
                i += 1  # a lifetime
                continue
            if c in "([{":
                stack.append((c, line))
            elif c in ")]}":
                assert stack and stack[-1][0] == pairs[c], f"{os.path.relpath(path, ROOT)}:{line}: unmatched `{c}`" + (f" (open `{stack[-1][0]}` from line {stack[-1][1]})" if stack else "")
                stack.pop()
            i += 1
        assert not stack, f"{os.path.relpath(path, ROOT)}: unclosed `{stack[-1][0]}` from line {stack[-1][1]}"
