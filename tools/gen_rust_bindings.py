#!/usr/bin/env python3
"""Generates rust/boojum_hip_sys.rs — the `extern "C"` declarations of EVERY entry point, struct, enum and constant of
include/boojum_hip.h — so that the Rust side of the boundary (rust/prove_hip.rs) never drifts from the header.
There is no Rust toolchain in the build image: the output is shipped as source, and tests/test_rust_bindings.py checks it
against the header (same symbols, same arity, same struct fields) and against the symbols the shared library exports.

    python tools/gen_rust_bindings.py            # rewrites rust/boojum_hip_sys.rs
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "boojum_hip.h")
OUT = os.path.join(ROOT, "rust", "boojum_hip_sys.rs")

SCALARS = {
    "int": "c_int", "unsigned": "c_uint", "unsigned int": "c_uint", "size_t": "usize", "uint64_t": "u64", "uint32_t": "u32",
    "float": "f32", "double": "f64", "unsigned char": "u8", "char": "c_char", "void": "c_void",
}


FN_TYPEDEFS = {}     # name -> Rust `Option<unsafe extern "C" fn(..)>` of every `typedef ret (*name)(params);` in the header


def strip_comments(src):
    return re.sub(r"/\*.*?\*/", " ", src, flags=re.S)


def rust_type(ctype, enums, structs):
    """C type (no declarator name) -> Rust type."""
    t = " ".join(ctype.replace("*", " * ").split())
    depth = t.count("*")
    base = t.replace("*", " ").split()
    # `const T *const *` : constness of the pointee decides *const / *mut at each level (outermost first)
    toks = t.split()
    consts = []          # per pointer level, from innermost to outermost: is the thing pointed to const?
    cur_const = False
    name_parts = []
    for tok in toks:
        if tok == "const":
            cur_const = True
        elif tok == "*":
            consts.append(cur_const)
            cur_const = False
        else:
            name_parts.append(tok)
    name = " ".join(name_parts)
    if name in FN_TYPEDEFS and not consts:          # typedef'd function pointer, passed by value
        return FN_TYPEDEFS[name]
    if name in SCALARS:
        r = SCALARS[name]
    elif name in enums:
        r = "c_int"
    elif name in structs:
        r = name
    else:
        raise ValueError("unknown C type %r" % ctype)
    for is_const in consts:
        r = ("*const " if is_const else "*mut ") + r
    assert depth == len(consts) and base
    return r


def split_params(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch == "(":
            depth += 1
        if ch == ")":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def parse_param(p, enums, structs, decay=False):
    """'const uint64_t *d_in' -> (name, rust type); function pointers handled by the caller.  `decay`: the declaration is a
    FUNCTION PARAMETER, where C adjusts an array type `T name[N]` to the pointer `T *name` (C11 6.7.6.3p7) — a by-value
    `[T; N]` on the Rust side would push N elements where the callee expects one address; struct fields keep their arrays."""
    m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)(\[(\d*)\])?$", p.strip())
    ctype, name, brackets, arr = m.groups()
    if brackets and decay:
        rt = rust_type(ctype.rstrip() + " *", enums, structs)
    else:
        rt = rust_type(ctype, enums, structs)
        if brackets:
            assert arr, "an array field needs its extent: %r" % p
            rt = "[%s; %s]" % (rt, arr)
    if name in ("type", "in", "ref", "fn", "match", "mod", "box", "loop", "move", "use"):
        name += "_"
    return name, rt


def parse_fn_pointer(decl, enums, structs):
    m = re.match(r"^(.*?)\(\s*\*\s*([A-Za-z_][A-Za-z0-9_]*)\s*\)\s*\((.*)\)$", decl.strip(), flags=re.S)
    ret, name, params = m.groups()
    ps = [parse_param(p, enums, structs, decay=True) for p in split_params(params)]
    r = rust_type(ret, enums, structs)
    sig = "unsafe extern \"C\" fn(%s)%s" % (", ".join("%s: %s" % p for p in ps), "" if r == "c_void" else " -> " + r)
    return name, "Option<%s>" % sig


def parse_header(path=HEADER):
    src = strip_comments(open(path).read())
    defines = [(m.group(1), m.group(2)) for m in re.finditer(r"^#define\s+(BJ_[A-Z0-9_]+)\s+(\d+|0[xX][0-9a-fA-F]+)[uU]?\s*$", src, flags=re.M)]
    enums, enum_consts = {}, []
    for m in re.finditer(r"(typedef\s+)?enum\s*([A-Za-z_0-9]*)\s*\{(.*?)\}\s*([A-Za-z_0-9]*)\s*;", src, flags=re.S):
        name = m.group(4) or m.group(2)
        items = []
        nxt = 0
        for it in m.group(3).split(","):
            it = it.strip()
            if not it:
                continue
            if "=" in it:
                k, v = (x.strip() for x in it.split("="))
                nxt = int(v, 0)
            else:
                k = it
            items.append((k, nxt))
            nxt += 1
        if name:
            enums[name] = items
        enum_consts += items
    opaque = re.findall(r"typedef\s+struct\s+([A-Za-z_0-9]+)\s+\1\s*;", src)
    struct_names = set(opaque) | set(re.findall(r"typedef\s+struct\s+([A-Za-z_0-9]+)\s*\{", src))
    structs = []
    for m in re.finditer(r"typedef\s+struct\s+([A-Za-z_0-9]+)\s*\{(.*?)\}\s*\1\s*;", src, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            if "(*" in decl.replace(" ", "") or re.search(r"\(\s*\*", decl):
                fields.append(parse_fn_pointer(decl, enums, struct_names))
                continue
            # `unsigned rank, world` style lists
            first = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*(\[\d+\])?)((\s*,\s*[A-Za-z_][A-Za-z0-9_]*)*)$", decl)
            ctype = first.group(1)
            names = [first.group(2)] + [x.strip() for x in first.group(4).split(",") if x.strip()]
            for nm in names:
                fields.append(parse_param(ctype + " " + nm, enums, struct_names))
        structs.append((m.group(1), fields))
    fn_types = []
    for m in re.finditer(r"typedef\s+([A-Za-z_][A-Za-z0-9_ \*]*?)\(\s*\*\s*(bj_[A-Za-z0-9_]+)\s*\)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        nm, rt = parse_fn_pointer("%s (*%s)(%s)" % (m.group(1), m.group(2), " ".join(m.group(3).split())), enums, struct_names)
        FN_TYPEDEFS[nm] = nm
        fn_types.append((nm, rt))
    src = re.sub(r"typedef\s+[A-Za-z_][A-Za-z0-9_ \*]*?\(\s*\*\s*bj_[A-Za-z0-9_]+\s*\)\s*\([^;]*?\)\s*;", " ", src, flags=re.S)
    body = re.sub(r"typedef\s+struct\s+[A-Za-z_0-9]+\s*\{.*?\}\s*[A-Za-z_0-9]+\s*;", " ", src, flags=re.S)
    body = re.sub(r"(typedef\s+)?enum\s*[A-Za-z_0-9]*\s*\{.*?\}\s*[A-Za-z_0-9]*\s*;", " ", body, flags=re.S)
    body = re.sub(r"^#.*$", " ", body, flags=re.M)
    body = body.replace('extern "C" {', " ").replace("}", " ")
    funcs = []
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(bj_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", body, flags=re.S):
        ret, name, params = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        if ret.startswith("typedef"):
            continue
        ps = [] if params in ("void", "") else [parse_param(p, enums, struct_names, decay=True) for p in split_params(params)]
        funcs.append((name, rust_type(ret, enums, struct_names), ps))
    return dict(defines=defines, enums=enums, enum_consts=enum_consts, opaque=opaque, structs=structs, funcs=funcs, fn_types=fn_types)


def generate():
    h = parse_header()
    o = ["// GENERATED by tools/gen_rust_bindings.py from include/boojum_hip.h — do not edit.",
         "// Raw FFI surface of libboojum_hip.so: %d functions, %d structs.  The safe wrapper and the `prove_hip` entry point that"
         % (len(h["funcs"]), len(h["structs"])),
         "// marshals the reference's WitnessSet / SetupBaseStorage / VerificationKey are in prove_hip.rs.",
         "#![allow(non_camel_case_types, non_upper_case_globals, dead_code)]",
         "use std::os::raw::{c_char, c_int, c_uint, c_void};", ""]
    for k, v in h["defines"]:
        o.append("pub const %s: u32 = %s;" % (k, v))
    o.append("")
    for name, items in h["enums"].items():
        o.append("pub type %s = c_int;" % name)
    for k, v in h["enum_consts"]:
        o.append("pub const %s: c_int = %d;" % (k, v))
    o.append("")
    for name, sig in h["fn_types"]:
        o.append("pub type %s = %s;" % (name, sig))
    for name in h["opaque"]:
        o += ["#[repr(C)]", "pub struct %s {" % name, "    _private: [u8; 0],", "}"]
    o.append("")
    for name, fields in h["structs"]:
        o += ["#[repr(C)]", "#[derive(Clone, Copy)]", "pub struct %s {" % name]
        o += ["    pub %s: %s," % f for f in fields]
        o += ["}", ""]
    o.append('#[link(name = "boojum_hip")]')
    o.append('extern "C" {')
    for name, ret, ps in h["funcs"]:
        sig = ", ".join("%s: %s" % p for p in ps)
        o.append("    pub fn %s(%s)%s;" % (name, sig, "" if ret == "c_void" else " -> " + ret))
    o += ["}", ""]
    return "\n".join(o)


def main():
    txt = generate()
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not os.path.exists(OUT) or open(OUT).read() != txt:
        with open(OUT, "w") as f:
            f.write(txt)
    return OUT


if __name__ == "__main__":
    print(main())
