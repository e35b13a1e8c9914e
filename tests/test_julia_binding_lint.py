"""CPU: static checks of the Julia binding (no Julia binary exists here, so the file can never be executed):
  * include/vcmi.h is parsed into canonical prototypes and compared with the ctypes table (_lib.SIGNATURES);
  * every `ccall((:sym, libvcmi), Ret, (Args...), values...)` of VoiceConversionMI.jl is checked against the header:
    symbol, return type, arity, each argument type, and the number of values passed;
  * every `var.field` access on a variable annotated with one of the module's struct types must name a declared field
    (this is the check that catches `tgmm.handle` when the field is `h`);
  * the export list covers the hot-path names of the reference's export list (src/VoiceConversion.jl:12-38)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT

JL = os.path.join(ROOT, "voiceconversion.jl_amd", "julia", "VoiceConversionMI.jl")
HDR = os.path.join(ROOT, "include", "vcmi.h")


# ------------------------------------------------------------------------------------------ header parsing
def c_class(t):
    """Canonical class of a C parameter / return type."""
    t = re.sub(r"\s+", " ", t.strip())
    t = re.sub(r"\b[A-Za-z_][A-Za-z0-9_]*$", "", t).strip() if not t.endswith("*") and " " in t else t   # drop the name
    t = t.replace("const ", "").replace(" const", "").replace(" ", "")
    table = {"double*": "f64*", "int64_t*": "i64*", "double**": "f64**", "int64_t**": "i64**", "int": "i32", "int64_t": "i64",
             "double": "f64", "void*": "void*", "int*": "i32*", "char*": "char*", "void": "void", "size_t": "u64"}
    if t in table:
        return table[t]
    m = re.fullmatch(r"(vcmi_[a-z_]+)(\*+)", t)
    if m:
        return "void*" if len(m.group(2)) == 1 else "void**"
    raise AssertionError(f"unparsed C type {t!r}")


def header_prototypes():
    text = open(HDR).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(vcmi_[A-Za-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        if "typedef" in ret:
            continue
        args = [a for a in (x.strip() for x in args.split(",")) if a and a != "void"]
        protos[name] = (c_class(ret + " x" if not ret.strip().endswith("*") else ret), [c_class(a) for a in args])
    return protos


def ctypes_class(t):
    if t is None:
        return "void"
    table = {C.c_int: "i32", C.c_int64: "i64", C.c_double: "f64", C.c_void_p: "void*", C.c_char_p: "char*", C.c_size_t: "u64",
             C.POINTER(C.c_double): "f64*", C.POINTER(C.c_int64): "i64*", C.POINTER(C.c_int): "i32*",
             C.POINTER(C.POINTER(C.c_double)): "f64**", C.POINTER(C.POINTER(C.c_int64)): "i64**", C.POINTER(C.c_void_p): "void**"}
    return table[t]


def compatible(a, b):
    """Two canonical classes may be bound to each other: equal, or a typed data pointer against an untyped one
    (device pointers travel as void* / integers)."""
    if a == b:
        return True
    single = {"f64*", "i64*", "i32*", "void*"}
    return a in single and b in single and "void*" in (a, b)


def test_ctypes_table_matches_the_header():
    import __graft_entry__ as ge
    ge.build()
    from voiceconversion_jl_amd import _lib
    protos = header_prototypes()
    assert set(protos) == set(_lib.SIGNATURES)
    for name, (ret, args) in protos.items():
        cres, cargs = _lib.SIGNATURES[name]
        assert compatible(ret, ctypes_class(cres)), (name, ret, cres)
        assert len(args) == len(cargs), (name, args, cargs)
        for k, (a, c) in enumerate(zip(args, cargs)):
            assert compatible(a, ctypes_class(c)), (name, k, a, c)


# ------------------------------------------------------------------------------------------- Julia parsing
def split_top(s):
    """Split on commas at bracket depth 0."""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def balanced(text, start):
    """text[start] == '(' -> index one past its matching ')'."""
    depth = 0
    for i in range(start, len(text)):
        if text[i] == "(":
            depth += 1
        elif text[i] == ")":
            depth -= 1
            if depth == 0:
                return i + 1
    raise AssertionError("unbalanced parentheses")


JL_TYPES = {"Cint": "i32", "Int64": "i64", "Cdouble": "f64", "Float64": "f64", "Cstring": "char*", "Ptr{Cvoid}": "void*",
            "Ptr{Float64}": "f64*", "Ref{Float64}": "f64*", "Ptr{Int64}": "i64*", "Ref{Int64}": "i64*", "Ptr{Cint}": "i32*",
            "Ref{Cint}": "i32*", "Ptr{Ptr{Float64}}": "f64**", "Ptr{Ptr{Int64}}": "i64**", "Ref{Ptr{Cvoid}}": "void**",
            "Cvoid": "void", "Csize_t": "u64"}


def julia_ccalls(text):
    calls = []
    for m in re.finditer(r"ccall\(", text):
        end = balanced(text, m.end() - 1)
        parts = split_top(text[m.end():end - 1])
        sym = re.fullmatch(r"\(:([A-Za-z0-9_]+),\s*libvcmi\)", parts[0])
        assert sym, f"ccall target not of the form (:sym, libvcmi): {parts[0]}"
        argt = parts[2].strip()
        assert argt.startswith("(") and argt.endswith(")"), parts[2]
        types = split_top(argt[1:-1])
        calls.append({"sym": sym.group(1), "ret": parts[1].strip(), "types": types, "values": parts[3:],
                      "line": text.count("\n", 0, m.start()) + 1})
    return calls


def lint_ccalls(text, protos):
    errors = []
    for c in julia_ccalls(text):
        where = f"line {c['line']} {c['sym']}"
        if c["sym"] not in protos:
            errors.append(f"{where}: not declared in vcmi.h")
            continue
        ret, args = protos[c["sym"]]
        if c["ret"] not in JL_TYPES or not compatible(JL_TYPES[c["ret"]], ret):
            errors.append(f"{where}: return type {c['ret']} vs C {ret}")
        if len(c["types"]) != len(args):
            errors.append(f"{where}: {len(c['types'])} argument types, C has {len(args)}")
            continue
        if len(c["values"]) != len(args):
            errors.append(f"{where}: {len(c['values'])} values passed for {len(args)} parameters")
        for k, (jt, ct) in enumerate(zip(c["types"], args)):
            if jt not in JL_TYPES:
                errors.append(f"{where}: unknown Julia type {jt}")
            elif not compatible(JL_TYPES[jt], ct):
                errors.append(f"{where}: argument {k + 1} is {jt}, C expects {ct}")
    return errors


def julia_structs(text):
    structs = {}
    for m in re.finditer(r"^(?:mutable )?struct (\w+)[^\n]*\n(.*?)^end", text, flags=re.S | re.M):
        body = m.group(2)
        fields = []
        for line in body.splitlines():
            if re.match(r"\s+function\b", line):
                break                                      # inner constructor: fields are above it
            f = re.match(r"\s+([^\s:#]+)::", line)
            if f:
                fields.append(f.group(1))
        structs[m.group(1)] = fields
    return structs


IDENT = r"[^\W\d][\w⁻¹²ˣʸᵛ]*"


def lint_fields(text, structs):
    """Every `var.field` where var is annotated `var::Struct` in the enclosing definition (or is the object a finalizer
    lambda receives) must be a declared field."""
    errors = []
    text = re.sub(r"#[^\n]*", lambda m: " " * len(m.group(0)), text)      # comments cite files like src/gmm.jl
    # definitions: `function name(args) ... end` blocks and one-line `name(args) = ...` methods
    blocks = []
    for m in re.finditer(r"^( *)function [^\n]*\n(?:.*?\n)*?\1end", text, flags=re.M):
        blocks.append((m.start(), m.group(0)))
    for m in re.finditer(r"^[\w\.!]+\([^\n]*\) = [^\n]*(?:\n {4,}[^\n]*)*", text, flags=re.M):
        blocks.append((m.start(), m.group(0)))
    for start, blk in blocks:
        line0 = text.count("\n", 0, start) + 1
        sig_end = balanced(blk, blk.index("("))
        sig = blk[:sig_end]
        types = {v: t for v, t in re.findall(rf"({IDENT})::(\w+)", sig) if t in structs}
        # objects built in the body: `x = new(...)` inside struct S, `x = S(...)`
        owner = None
        for sname in structs:
            sm = re.search(rf"^(?:mutable )?struct {sname}\b.*?^end", text, flags=re.S | re.M)
            if sm and sm.start() <= start < sm.end():
                owner = sname
        for v, t in re.findall(rf"({IDENT}) = (\w+)\(", blk):
            if t == "new" and owner:
                types.setdefault(v, owner)
            elif t in structs:
                types.setdefault(v, t)
        # finalizer(x -> ... , obj): x has obj's type
        for fm in re.finditer(rf"finalizer\(({IDENT}) -> (.*?), ({IDENT})\)\s*$", blk, flags=re.M):
            lam, body, obj = fm.groups()
            if obj in types:
                for fld in re.findall(rf"\b{lam}\.({IDENT})", body):
                    if fld not in structs[types[obj]]:
                        errors.append(f"line ~{line0}: finalizer accesses {types[obj]}.{fld}, fields are {structs[types[obj]]}")
        for v, t in types.items():
            for am in re.finditer(rf"(?<![\w\.]){re.escape(v)}\.({IDENT})", blk):
                fld = am.group(1)
                if fld not in structs[t]:
                    ln = line0 + blk.count("\n", 0, am.start())
                    errors.append(f"line {ln}: {v}::{t} has no field {fld} (fields: {structs[t]})")
    return errors


def test_every_ccall_matches_the_header():
    text = open(JL).read()
    protos = header_prototypes()
    calls = julia_ccalls(text)
    assert len(calls) >= 35
    errors = lint_ccalls(text, protos)
    assert not errors, "\n".join(errors)
    # the binding reaches every host-pointer entry point of the hot path
    used = {c["sym"] for c in calls}
    for sym in ("vcmi_gmmmap_create", "vcmi_gmmmap_convert", "vcmi_vc_frames", "vcmi_gmmmap_posterior", "vcmi_gmmmap_predict",
                "vcmi_dtw_fit", "vcmi_dtw_fit_batch", "vcmi_align", "vcmi_align_batch", "vcmi_estep_diag", "vcmi_estep_full",
                "vcmi_traj_create", "vcmi_traj_convert", "vcmi_traj_convert_batch", "vcmi_vc_traj", "vcmi_trajgv_create",
                "vcmi_trajgv_convert", "vcmi_trajgv_convert_batch", "vcmi_push_delta", "vcmi_variance_scaling", "vcmi_diffgmm",
                "vcmi_align_mcep", "vcmi_set_devices"):
        assert sym in used, f"the Julia module never calls {sym}"


def test_field_accesses_name_declared_fields():
    text = open(JL).read()
    structs = julia_structs(text)
    for s in ("GMMMapParam", "GMM", "GMMMap", "TrajectoryGMMMap", "TrajectoryGVGMMMap", "DTW", "VarianceScaling", "GMMEM"):
        assert s in structs and structs[s], s
    # the fields the reference's own code reads (src/trajectory_gmmmap.jl:20-22,82; src/diffgmm.jl:10-15; src/dtw.jl:12-16)
    assert {"params", "px"} <= set(structs["GMMMap"])
    assert {"weights", "μˣ", "μʸ", "Σˣˣ", "Σˣʸ", "Σʸˣ", "Σʸʸ", "ΣʸˣΣˣˣ⁻¹"} == set(structs["GMMMapParam"])
    assert {"tgmm", "μᵛ", "Σᵛᵛ"} <= set(structs["TrajectoryGVGMMMap"]) and "gmmmap" in structs["TrajectoryGMMMap"]
    assert ["fstep", "bstep", "template", "costtable", "backpointer"] == structs["DTW"]
    errors = lint_fields(text, structs)
    assert not errors, "\n".join(errors)


def test_lint_catches_the_round_1_bugs():
    """The checks above must fail on the defects round 1 shipped: a wrong field name and a drifted ccall signature."""
    text = open(JL).read()
    structs = julia_structs(text)
    bad = text.replace("tgmm.h, μᵛ, Σᵛᵛ, h))", "tgmm.handle, μᵛ, Σᵛᵛ, h))")
    assert bad != text and any("no field handle" in e for e in lint_fields(bad, structs))
    protos = header_prototypes()
    bad2 = text.replace("(Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}),\n                t.h, X, size(X, 2), Y))",
                        "(Ptr{Cvoid}, Ptr{Float64}, Cint, Ptr{Float64}),\n                t.h, X, size(X, 2), Y))")
    assert bad2 != text and any("argument 3" in e for e in lint_ccalls(bad2, protos))
    bad3 = text.replace("g.h, fm, size(fm, 2), out))", "g.h, fm, out))")
    assert bad3 != text and any("values passed" in e for e in lint_ccalls(bad3, protos))


def test_reference_signatures_are_kept():
    text = open(JL).read()
    for needle in ("predict_proba(gmm::GMM, X::Matrix{Float64})", "predict_proba(gmm::GMM, x::Vector{Float64})",
                   "predict(gmm::GMM, X::Matrix{Float64})", "predict(gmm::GMM, x::Vector{Float64})",
                   "function diffgmm(p::GMMMapParam)", "function vc(tgv::TrajectoryGVGMMMap, fm::Matrix{Float64})",
                   "function vc(t::TrajectoryGMMMap, fm::Matrix{Float64})", "function vc(g::GMMMap, fm::Matrix{Float64})",
                   "GMMMap(weights::Vector{Float64}, μ::Matrix{Float64}, Σ::Array{Float64,3}; swap::Bool=false)",
                   "TrajectoryGMMMap(g::GMMMap, T::Int)", "DTW(; fstep=0, bstep=1)"):
        assert needle in text, needle
    exports = re.search(r"^export (.*?)\n\n", text, flags=re.S | re.M).group(1)
    names = {n.strip() for n in exports.replace("\n", " ").split(",")}
    # hot-path part of the reference's export list, src/VoiceConversion.jl:12-38 (Dataset types are file I/O: out of scope)
    for n in ("FrameByFrameConverter", "TrajectoryConverter", "GMMMapParam", "GMMMap", "TrajectoryGMMMap", "TrajectoryGVGMMMap",
              "fvconvert", "vc", "ncomponents", "dim", "VarianceScaling", "fvpostf!", "fvpostf", "align", "align_mcep",
              "push_delta"):
        assert n in names, n


# Julia 0.5-only syntax that no longer parses / dispatches on Julia >= 1.0 (the module targets >= 1.0: `mutable struct`,
# `undef`, `GC.@preserve`).  One such token anywhere makes `include("VoiceConversionMI.jl")` fail as a whole.
JULIA_05_TOKENS = [
    (r"^\s*immutable\b", "`immutable` (write `struct`)"),
    (r"^\s*type\s+\w", "`type` (write `mutable struct`)"),
    (r"^\s*abstract\s+(?!type\b)\w", "`abstract X` (write `abstract type X end`)"),
    (r"^\s*typealias\b", "`typealias`"),
    (r"\bArray\(\s*[A-Z]\w*\s*,", "`Array(T, dims...)` (write `Array{T}(undef, dims...)`)"),
    (r"\bVector\(\s*[A-Z]\w*\s*,", "`Vector(T, n)`"),
    (r"\bMatrix\(\s*[A-Z]\w*\s*,", "`Matrix(T, m, n)`"),
    (r"\bVoid\b", "`Void` (write `Cvoid` / `Nothing`)"),
    (r"\b(sumabs2|indmax|indmin|repmat|blkdiag|speye|findin)\(", "a Base function removed in Julia 1.0"),
    (r"\bfind\(", "`find(` (write `findall`)"),
    (r"\bfunction\s+\w+\{", "`f{T}(...)` (write `f(...) where T`)"),
    (r"\bfinalizer\(\s*[^\W\d]\w*\s*,\s*\w+\s*->", "`finalizer(obj, f)` (Julia >= 1.0 takes the function first)"),
]


def julia_05_errors(text):
    code = re.sub(r"#[^\n]*", "", text)                   # comments may describe the old syntax
    code = re.sub(r'"(?:[^"\\\n]|\\.)*"', '""', code)     # and so may strings
    errors = []
    for pat, what in JULIA_05_TOKENS:
        for m in re.finditer(pat, code, flags=re.M):
            errors.append(f"line {code.count(chr(10), 0, m.start()) + 1}: Julia 0.5-only {what}")
    return errors


def test_no_julia_05_only_syntax():
    text = open(JL).read()
    assert not julia_05_errors(text), "\n".join(julia_05_errors(text))
    # the lint must catch what round 2 shipped (`immutable GVDataset`, `Array(Float64, Dout, n)`)
    assert any("immutable" in e for e in julia_05_errors(text.replace("\nstruct GVDataset", "\nimmutable GVDataset")))
    assert any("Array(T" in e for e in julia_05_errors(text.replace("Matrix{Float64}(undef, Dout, n)", "Array(Float64, Dout, n)")))


def test_julia_block_structure_balances():
    """Every block opener of the module has its `end` (a cheap stand-in for the parser that is not in the image)."""
    code = re.sub(r"#[^\n]*", "", open(JL).read())
    code = re.sub(r'"(?:[^"\\\n]|\\.)*"', '""', code)
    code = re.sub(r"\[[^\[\]\n]*\bend\b[^\[\]\n]*\]", "[]", code)          # a[2:end]
    # openers start a line (comprehension `for`s and `x = cond ? a : b` never do); `do` blocks end one
    opens = len(re.findall(r"^[ \t]*(?:module|function|struct|mutable[ \t]+struct|abstract[ \t]+type|if|for|while|let|try|begin|macro)\b", code, flags=re.M))
    opens += len(re.findall(r"\bdo\b[^\n]*$", code, flags=re.M)) + len(re.findall(r"\S[ \t]+begin[ \t]*$", code, flags=re.M))
    ends = len(re.findall(r"(?<![\w.:])end\b", code))
    assert opens == ends, (opens, ends)
