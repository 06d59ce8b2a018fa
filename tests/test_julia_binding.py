"""Static check of julia/ROCMeshField.jl against include/lsm.h.

The Julia glue cannot be executed here (no Julia runtime), so its C side is checked mechanically instead:
  * every `struct Lsm*` mirror: field names, order and types equal the header's `typedef struct`;
  * every `ccall((:lsm_…, libhiplsm), Ret, (Args…), …)`: the symbol is declared in the header, the return type matches,
    and the argument tuple has the declared number of arguments, each of the declared kind (pointer / int / int64 /
    double), and as many values follow the tuple as it has entries;
  * constants mirrored by hand (LSM_GHOST, LSM_COMM_ID_BYTES, the LSM_BC_NONE kind) equal the header's.
The checker itself is checked: swapping two struct fields, or dropping a ccall argument, must make it fail."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lsm.h")
JULIA = os.path.join(ROOT, "julia", "ROCMeshField.jl")


# ----------------------------------------------------------------------------- the header
def _strip_c_comments(s):
    return re.sub(r"/\*.*?\*/", " ", s, flags=re.S)


def _c_kind(t):
    """pointer / int / int64 / double / void / struct:<name> of a C type (with declarator suffix)."""
    t = t.strip()
    if "*" in t or "[" in t or t.startswith("LsmStageHook"):
        return "pointer"
    base = re.sub(r"\b(const|struct)\b", "", t).split()[0]
    return {"int": "int", "int32_t": "int", "int64_t": "int64", "double": "double", "void": "void"}.get(base, "struct:" + base)


def header_structs(text):
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*\1\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            fm = re.match(r"(.*?)(\w+)\s*(\[\s*(\w+)\s*\])?$", decl, flags=re.S)
            ctype, name, _, dim = fm.group(1).strip(), fm.group(2), fm.group(3), fm.group(4)
            base = "ptr" if "*" in ctype else re.sub(r"\b(const|struct)\b", "", ctype).split()[0]
            fields.append((name, base, dim))
        out[m.group(1)] = fields
    return out


def header_functions(text):
    out = {}
    for m in re.finditer(r"(?m)^(const\s+char\s*\*|int|void)\s+(lsm_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret = "pointer" if "*" in m.group(1) else m.group(1)
        args = m.group(3).strip()
        kinds = [] if args in ("", "void") else [_c_kind(a) for a in _split_top(args)]
        out[m.group(2)] = (ret, kinds)
    return out


def header_constants(text):
    c = {k: int(v) for k, v in re.findall(r"#define\s+(LSM_\w+)\s+(\d+)", text)}
    for body in re.findall(r"enum\s*\{(.*?)\}", text, flags=re.S):
        for k, v in re.findall(r"(LSM_\w+)\s*=\s*(-?\d+)", body):
            c[k] = int(v)
    return c


def _split_top(s):
    """split at commas outside (), {} and []."""
    parts, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return [p.strip() for p in parts]


# ----------------------------------------------------------------------------- the Julia file
def _strip_jl_comments(s):
    return re.sub(r"(?m)#.*$", "", s)


def julia_structs(text):
    out = {}
    for m in re.finditer(r"(?m)^struct\s+(Lsm\w+)\s*\n(.*?)^end", text, flags=re.S):
        fields = []
        for line in m.group(2).splitlines():
            line = line.strip()
            if line:
                name, jt = line.split("::")
                fields.append((name.strip(), jt.strip()))
        out[m.group(1)] = fields
    return out


_JL_SCALAR = {"Int32": "int32_t", "Int64": "int64_t", "Float64": "double", "Cint": "int32_t"}


def _jl_field(jt):
    """(base, dim) of a Julia field type in the header's vocabulary."""
    m = re.match(r"NTuple\{\s*(\d+)\s*,\s*(.+)\}$", jt)
    dim = None
    if m:
        dim, jt = m.group(1), m.group(2).strip()
    base = "ptr" if jt.startswith("Ptr{") else _JL_SCALAR.get(jt, jt)
    return base, dim


def _jl_kind(t):
    t = t.strip()
    if t.startswith(("Ptr{", "Ref{")) or t == "Cstring":
        return "pointer"
    return {"Cint": "int", "Int32": "int", "Int64": "int64", "Float64": "double", "Cdouble": "double", "Cvoid": "void"}.get(t, "?" + t)


def _balanced(text, i):
    """index just past the parenthesis group opening at text[i] == '('."""
    depth = 0
    for j in range(i, len(text)):
        if text[j] == "(":
            depth += 1
        elif text[j] == ")":
            depth -= 1
            if depth == 0:
                return j + 1
    raise ValueError("unbalanced parentheses")


def julia_ccalls(text):
    """[(symbol, return type, [arg types], number of values passed)]"""
    out = []
    for m in re.finditer(r"ccall\(", text):
        end = _balanced(text, m.end() - 1)
        parts = _split_top(text[m.end():end - 1])
        sym = re.match(r"\(\s*:(\w+)\s*,\s*libhiplsm\s*\)$", parts[0])
        assert sym, f"ccall with an unexpected target: {parts[0]}"
        tup = parts[2].strip()
        assert tup.startswith("(") and tup.endswith(")"), tup
        args = [a for a in _split_top(tup[1:-1]) if a]
        out.append((sym.group(1), parts[1].strip(), args, len(parts) - 3))
    return out


# ----------------------------------------------------------------------------- the comparison
def compare(header_text, julia_text):
    """list of mismatch messages (empty = the binding matches the header)"""
    h, j = _strip_c_comments(header_text), _strip_jl_comments(julia_text)
    errs = []
    hs, js = header_structs(h), julia_structs(j)
    for name, jf in js.items():
        if name not in hs:
            errs.append(f"struct {name} is not in the header")
            continue
        hf = hs[name]
        if len(hf) != len(jf):
            errs.append(f"struct {name}: {len(jf)} fields, the header has {len(hf)}")
            continue
        for (hn, hb, hd), (jn, jt) in zip(hf, jf):
            jb, jd = _jl_field(jt)
            consts = header_constants(h)
            hdim = None if hd is None else str(consts.get(hd, hd))
            if (hn, hb, hdim) != (jn, jb, jd):
                errs.append(f"struct {name}: field {jn}::{jt} where the header has {hb} {hn}" + (f"[{hd}]" if hd else ""))
    for need in ("LsmGrid", "LsmBc", "LsmSlab", "LsmLayout", "LsmCoeff", "LsmTerm", "LsmBand"):
        if need not in js:
            errs.append(f"struct {need} is not mirrored")
    hf = header_functions(h)
    calls = julia_ccalls(j)
    for sym, ret, args, nvals in calls:
        if sym not in hf:
            errs.append(f"ccall :{sym}: not declared in the header")
            continue
        hret, hkinds = hf[sym]
        if _jl_kind(ret) != hret:
            errs.append(f"ccall :{sym}: returns {ret}, the header says {hret}")
        kinds = [_jl_kind(a) for a in args]
        if kinds != hkinds:
            errs.append(f"ccall :{sym}: argument kinds {kinds}, the header says {hkinds}")
        if nvals != len(args):
            errs.append(f"ccall :{sym}: {nvals} values for {len(args)} declared arguments")
    consts = header_constants(h)
    for name in ("LSM_GHOST", "LSM_COMM_ID_BYTES"):
        m = re.search(rf"const\s+{name}\s*=\s*(\d+)", j)
        if not m or int(m.group(1)) != consts[name]:
            errs.append(f"constant {name} differs from the header")
    m = re.search(r"const\s+LSM_BC_NONE\s*=\s*LsmBc\((\d+)", j)
    if not m or int(m.group(1)) != consts["LSM_BC_NONE"]:
        errs.append("LSM_BC_NONE differs from the header")
    return errs, {c[0] for c in calls}


def test_julia_binding_matches_the_header():
    errs, called = compare(open(HEADER).read(), open(JULIA).read())
    assert not errs, "\n".join(errs)
    # the binding reaches the whole hot path and the multi-GPU entry points
    for sym in ("lsm_create", "lsm_destroy", "lsm_set_stream", "lsm_layout", "lsm_upload", "lsm_download", "lsm_compute_cfl",
                "lsm_advance_fe", "lsm_advance_rk2", "lsm_advance_rk3", "lsm_eikonal_sign", "lsm_extrema", "lsm_fill_ghosts",
                "lsm_comm_unique_id", "lsm_comm_attach_rccl", "lsm_comm_attach_local", "lsm_halo_start", "lsm_halo_wait",
                "lsm_halo_exchange", "lsm_allreduce_dt", "lsm_comm_detach", "lsm_comm_info", "lsm_comm_set_overlap",
                "lsm_band_update", "lsm_stage_band", "lsm_compute_cfl_band", "lsm_reinitialize", "lsm_volume", "lsm_perimeter",
                # a NarrowBandMeshField steps through the same _advance! dispatch (src/timestepping.jl:128-202), slab-decomposed or not
                "lsm_advance_band_fe", "lsm_advance_band_rk2", "lsm_advance_band_rk3", "lsm_band_status", "lsm_band_halo", "lsm_band_retile",
                "lsm_band_overlap_config", "lsm_band_overlap_mask", "lsm_band_overlap_values", "lsm_comm_abort"):
        assert sym in called, f"{sym} is not bound"


def test_the_band_field_has_every_method_the_integrator_dispatches_on():
    """integrate! on a ROCNarrowBandMeshField must not end in a MethodError: _alloc_buffers and _advance! for the three
    integrators, compute_cfl, update_band!, copy and copy! are defined on it (src/timestepping.jl:101-202, src/levelsetequation.jl:67-76)."""
    j = _strip_jl_comments(open(JULIA).read())
    for pat in (r"LSM\._alloc_buffers\(::LSM\.ForwardEuler, ϕ::ROCNarrowBandMeshField\)",
                r"LSM\._alloc_buffers\(::Union\{LSM\.RK2, LSM\.RK3\}, ϕ::ROCNarrowBandMeshField\)",
                r"function LSM\._advance!\(::LSM\.ForwardEuler, ϕ::ROCNarrowBandMeshField",
                r"function LSM\._advance!\(::LSM\.RK2, ϕ::ROCNarrowBandMeshField",
                r"function LSM\._advance!\(::LSM\.RK3, ϕ::ROCNarrowBandMeshField",
                r"function LSM\.compute_cfl\(terms, ϕ::ROCNarrowBandMeshField",
                r"function LSM\.update_band!\(ϕ::ROCNarrowBandMeshField",
                r"function Base\.copy\(ϕ::ROCNarrowBandMeshField",
                r"function Base\.copy!\(dst::ROCNarrowBandMeshField"):
        assert re.search(pat, j), pat


def test_header_parser_sees_every_declaration():
    from lsm_amd import _lib
    h = _strip_c_comments(open(HEADER).read())
    assert set(header_functions(h)) == set(_lib.EXPORTS)
    hs = header_structs(h)
    assert [f[0] for f in hs["LsmCoeff"]] == ["kind", "time_kind", "time_param", "value", "field", "sep"]
    assert hs["LsmGrid"][2] == ("n", "int64_t", "LSM_MAX_DIM")


@pytest.mark.parametrize("mutate,expect", [
    (lambda s: s.replace("    origin::Int64\n    total::Int64\n", "    total::Int64\n    origin::Int64\n"), "struct LsmLayout"),
    (lambda s: s.replace("    time_param::Float64\n", "    time_param::Float32\n"), "struct LsmCoeff"),
    (lambda s: s.replace("(Ptr{Cvoid}, Ptr{LsmTerm}, Cint, Ptr{Cvoid}, Float64, Ref{Float64}),\n        ϕ.h.ptr, ts, length(ts), pointer(ϕ.buf), t, dt)",
                         "(Ptr{Cvoid}, Ptr{LsmTerm}, Cint, Ptr{Cvoid}, Ref{Float64}),\n        ϕ.h.ptr, ts, length(ts), pointer(ϕ.buf), dt)"), "lsm_compute_cfl"),
    (lambda s: s.replace("(:lsm_halo_wait, libhiplsm), Cint, (Ptr{Cvoid},), ϕ.h.ptr)", "(:lsm_halo_wait, libhiplsm), Cint, (Ptr{Cvoid},), ϕ.h.ptr, 0)"), "lsm_halo_wait"),
    (lambda s: s.replace("(Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint), ϕ.h.ptr, id, ϕ.h.rank, ϕ.h.world)", "(Ptr{Cvoid}, Ptr{Cvoid}, Cint, Float64), ϕ.h.ptr, id, ϕ.h.rank, ϕ.h.world)"),
     "lsm_comm_attach_rccl"),
    (lambda s: s.replace("(:lsm_sync, libhiplsm)", "(:lsm_synchronise, libhiplsm)"), "lsm_synchronise"),
    (lambda s: s.replace("const LSM_GHOST = 3", "const LSM_GHOST = 2"), "LSM_GHOST"),
])
def test_the_checker_catches_a_broken_binding(mutate, expect):
    src = open(JULIA).read()
    broken = mutate(src)
    assert broken != src, "the mutation did not apply: update this test with the binding"
    errs, _ = compare(open(HEADER).read(), broken)
    assert any(expect in e for e in errs), errs
