"""Writes tests/golden/reference_signatures.json: the parameter and return TYPES (no bodies, no names) of the reference
functions that rust/src/gates_hip.rs mirrors, read from /root/reference (run in the build container; the GPU box has
no reference tree).  tests/test_binding_lint.py holds every `pub fn *_hip` to this table and, where the reference tree
is present, the table to the tree."""
import json
import os
import re
import sys

REF = "/root/reference/src"
WANTED = {  # rust/src/gates_hip.rs name -> (reference file, function)
    "batch_nand_hip": ("gates.rs", "batch_nand"), "batch_and_hip": ("gates.rs", "batch_and"),
    "batch_or_hip": ("gates.rs", "batch_or"), "batch_xor_hip": ("gates.rs", "batch_xor"),
    "batch_nor_hip": ("gates.rs", "batch_nor"), "batch_xnor_hip": ("gates.rs", "batch_xnor"),
    "batch_blind_rotate_hip": ("trgsw.rs", "batch_blind_rotate"),
    "mux_hip": ("gates.rs", "mux"), "mux_naive_hip": ("gates.rs", "mux_naive"),
}


def norm_type(t: str) -> str:
    """`key::CloudKey` / `crate::key::CloudKey` / `CloudKey` -> `CloudKey`; `tlwe::TLWELv0` == `Ciphertext` (utils.rs:7)."""
    t = re.sub(r"\s+", "", t)
    t = re.sub(r"(?:\w+::)+", "", t)
    return t.replace("TLWELv0", "Ciphertext")


def free_fn_signature(text: str, name: str):
    """Parameter types and return type of the FREE function `pub fn name(` (column 0: not a method of `impl Gates`)."""
    m = re.search(r"^pub fn " + name + r"\s*\((.*?)\)\s*->\s*([^{]+?)\s*\{", text, flags=re.S | re.M)
    if not m:
        return None
    params = [norm_type(a.split(":", 1)[1]) for a in re.split(r",(?![^<(\[]*[>)\]])", m.group(1)) if ":" in a]
    return {"params": params, "ret": norm_type(m.group(2))}


def collect(ref=REF):
    out = {}
    for hip_name, (fname, fn) in WANTED.items():
        sig = free_fn_signature(open(os.path.join(ref, fname)).read(), fn)
        assert sig, (fname, fn)
        out[hip_name] = dict(sig, reference=f"src/{fname}::{fn}")
    return out


if __name__ == "__main__":
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_signatures.json")
    json.dump(collect(), open(dst, "w"), indent=1, sort_keys=True)
    print("wrote", dst, file=sys.stderr)
