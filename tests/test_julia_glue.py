"""The Julia side of the C ABI ships as files (julia/ext/*.jl, julia/clarabel_l1_seam.patch, julia/parity_dump.jl); there is no `julia` in the build
image, so what can be checked without executing them is checked here: every `ccall((:sym, libhipkkt), Ret, (ArgTypes...), ...)`
names a function declared in include/hipkkt.h, with the declared number of parameters, a matching return type and matching parameter
type classes (Ptr{Float64} <-> double*, Int64 <-> int64_t, Ptr{Cvoid} <-> hipkkt_handle, ...); the `HipKKTOpts` struct mirrors
`hipkkt_opts` field for field; INTEGRATION.md refers to the files instead of carrying copies of them; and the core patch of seam L1
is a REAL patch: it applies to the reference checkout (`patch --dry-run`, when /root/reference is there), never mentions the extension
module (the core must not import what imports it), and every kktsolver_* function it calls is defined by the patch itself (generic
default) and extended by the extension."""
import shutil
import subprocess
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL_DIR = os.path.join(ROOT, "julia", "ext")
PATCH = os.path.join(ROOT, "julia", "clarabel_l1_seam.patch")
JL_FILES = [os.path.join(JL_DIR, f) for f in ("ClarabelHipKKTExt.jl", "hipkkt_lib.jl", "directldl_hip.jl", "kktsolver_hip.jl")]


def _strip_c_comments(s):
    return re.sub(r"/\*.*?\*/", " ", s, flags=re.S)


def c_prototypes():
    src = _strip_c_comments(open(os.path.join(ROOT, "include", "hipkkt.h")).read())
    src = re.sub(r"^\s*#[^\n]*$", "", src, flags=re.M)      # preprocessor lines
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(hipkkt_\w+)\s*\(([^()]*)\)\s*;", src):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()
        plist = [] if params in ("", "void") else [p.strip() for p in params.split(",")]
        protos[name] = (ret, plist)
    return protos


def c_class(p):
    """type class of a C parameter / return declaration"""
    p = re.sub(r"\bconst\b", "", p).strip()
    p = re.sub(r"/\*.*", "", p)
    stars = p.count("*")
    base = re.sub(r"[\*\s]+\w*$", "", p) if stars == 0 and " " in p else p
    toks = p.replace("*", " * ").split()
    base = toks[0]
    if base == "hipkkt_handle":
        return "handle*" if stars else "handle"
    if base in ("double", "int64_t", "int32_t", "hipkkt_opts", "char", "void"):
        return base + "*" * stars
    return base + "*" * stars


JL_CLASS = {
    "Float64": "double", "Int64": "int64_t", "Int32": "int32_t", "Cvoid": "void", "Cstring": "char*",
    "Ptr{Float64}": "double*", "Ref{Float64}": "double*", "Ptr{Int64}": "int64_t*", "Ref{Int64}": "int64_t*",
    "Ptr{Int32}": "int32_t*", "Ref{Int32}": "int32_t*", "Ptr{Cvoid}": "handle", "Ref{Ptr{Cvoid}}": "handle*",
    "Ref{HipKKTOpts}": "hipkkt_opts*",
}


def split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "{(":
            depth += 1
        elif ch in "})":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def jl_ccalls(path):
    src = open(path).read()
    src = re.sub(r"#[^\n]*", "", src)           # comments
    calls = []
    for m in re.finditer(r"ccall\(\(:(\w+),\s*libhipkkt\)\s*,\s*([\w{}]+)\s*,\s*\(", src):
        i = m.end()
        depth, j = 1, i
        while depth:
            depth += {"(": 1, ")": -1}.get(src[j], 0)
            j += 1
        types = split_top(src[i:j - 1])
        # the actual arguments: up to the closing parenthesis of the ccall
        k, depth = j, 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[k], 0)
            k += 1
        args = split_top(src[j:k - 1].lstrip().lstrip(","))
        calls.append((m.group(1), m.group(2), types, args, os.path.basename(path)))
    return calls


def test_files_exist_and_the_module_includes_them():
    for f in JL_FILES + [PATCH, os.path.join(ROOT, "julia", "parity_dump.jl"),
                         os.path.join(ROOT, "julia", "make_config_json.py"), os.path.join(ROOT, "julia", "compare_parity.py")]:
        assert os.path.isfile(f), f
    mod = open(JL_FILES[0]).read()
    for inc in ("hipkkt_lib.jl", "directldl_hip.jl", "kktsolver_hip.jl"):
        assert f'include("./{inc}")' in mod


@pytest.mark.parametrize("path", JL_FILES[1:], ids=os.path.basename)
def test_every_ccall_matches_the_header(path):
    protos = c_prototypes()
    calls = jl_ccalls(path)
    assert calls, f"no ccall found in {path}"
    for name, ret, types, args, fn in calls:
        assert name in protos, f"{fn}: ccall of {name}, which include/hipkkt.h does not declare"
        cret, cparams = protos[name]
        assert len(types) == len(cparams), f"{fn}: {name} takes {len(cparams)} parameters, the ccall lists {len(types)} types"
        assert len(args) == len(types), f"{fn}: {name}: {len(types)} argument types but {len(args)} arguments"
        assert JL_CLASS.get(ret) == c_class(cret), f"{fn}: {name} returns {cret!r}, the ccall says {ret}"
        for k, (jt, cp) in enumerate(zip(types, cparams)):
            assert jt in JL_CLASS, f"{fn}: {name}: unknown Julia argument type {jt}"
            assert JL_CLASS[jt] == c_class(cp), f"{fn}: {name} parameter {k + 1} is {cp!r}, the ccall says {jt}"


def test_the_plugins_cover_the_contract_and_the_widened_rows():
    """the symbols each seam must reach (SURVEY section 8b + 8f)"""
    used = {c[0] for f in JL_FILES[1:] for c in jl_ccalls(f)}
    l0 = {"hipkkt_create", "hipkkt_update_values", "hipkkt_scale_values", "hipkkt_refactor", "hipkkt_ldl_solve", "hipkkt_info",
          "hipkkt_is_available", "hipkkt_abi_version", "hipkkt_destroy", "hipkkt_default_opts", "hipkkt_last_error"}
    l1 = {"hipkkt_create_from_parts", "hipkkt_set_hs", "hipkkt_set_soc_batch", "hipkkt_set_genpow", "hipkkt_setrhs", "hipkkt_solve",
          "hipkkt_update_P", "hipkkt_update_A", "hipkkt_get_dims"}
    widened = {"hipkkt_set_cone_types", "hipkkt_update_scaling", "hipkkt_set_hs_psd", "hipkkt_solve_multi", "hipkkt_set_qb",
               "hipkkt_kkt_solve_reduced", "hipkkt_residuals", "hipkkt_trim_cache"}
    assert l0 <= used and l1 <= used and widened <= used, sorted((l0 | l1 | widened) - used)
    l1src = open(JL_FILES[3]).read()
    assert "hip_device()" in l1src and "CLARABEL_HIP_DEVICE" in open(JL_FILES[1]).read()      # both seams read the device from the environment


def test_opts_struct_mirrors_the_header():
    src = _strip_c_comments(open(os.path.join(ROOT, "include", "hipkkt.h")).read())
    body = re.search(r"typedef\s+struct\s+hipkkt_opts\s*\{(.*?)\}\s*hipkkt_opts\s*;", src, flags=re.S).group(1)
    cfields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            toks = decl.replace("*", " * ").split()
            cfields.append((c_class(" ".join(toks[:-1])), toks[-1]))
    jl = open(JL_FILES[1]).read()
    jbody = re.search(r"struct HipKKTOpts\n(.*?)\nend", jl, flags=re.S).group(1)
    jfields = [(n.strip(), t.strip()) for n, t in re.findall(r"(\w+)::([\w{}]+)", jbody)]
    assert [n for n, _ in jfields] == [n for _, n in cfields]
    jmap = {"Int32": "int32_t", "Float64": "double", "Ptr{Int64}": "int64_t*"}
    assert [jmap[t] for _, t in jfields] == [c for c, _ in cfields]


def test_integration_md_points_at_the_files():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for f in ("julia/ext/directldl_hip.jl", "julia/ext/kktsolver_hip.jl", "julia/clarabel_l1_seam.patch", "julia/parity_dump.jl"):
        assert f in text, f
    assert "mutable struct HipDirectLDLSolver" not in text and "mutable struct HipKKTSolver" not in text      # no inlined copies


# ---- seam L1: the core patch ------------------------------------------------------------------

def _patch_added_lines():
    return [ln[1:] for ln in open(PATCH).read().splitlines() if ln.startswith("+") and not ln.startswith("+++")]


def test_core_patch_never_names_the_extension():
    """the patch hard-coded ClarabelHipKKTExt.HipKKTSolver inside module Clarabel -- the core would have to import the
    extension that imports it.  Registration is by Val dispatch, extended FROM the extension."""
    text = open(PATCH).read()
    for tok in ("ClarabelHipKKTExt.", "HipKKTSolver", "hipkkt"):
        assert tok not in text, tok
    # the TOKEN :hip appears in exactly one place of the core: the priority list of `:auto` (directldl_auto.jl:24), where an engine
    # that is not loaded simply reports unavailable (directldl_defaults.jl:22-27) -- review of round 5, "missing" item 3
    hip_lines = [ln for ln in text.splitlines() if ":hip" in ln]
    assert hip_lines == ["+    priority = [:hip,:panua,:mkl,:ma57,:qdldl]"], hip_lines
    ext = open(os.path.join(JL_DIR, "kktsolver_hip.jl")).read()
    assert "Clarabel.kktsolver_constructor(::Val{:hip}) = HipKKTSolver" in ext
    assert "isdefined(Clarabel, :kktsolver_constructor)" in ext            # an unpatched core still loads the extension (seam L0 only)
    l0 = open(os.path.join(JL_DIR, "directldl_hip.jl")).read()
    assert "ldlsolver_constructor(::Val{" in l0 and ":hip_ldl" in l0


def test_core_patch_defines_what_it_calls_and_the_extension_extends_it():
    added = "\n".join(_patch_added_lines())
    called = set(re.findall(r"\b(kktsolver_\w+!?)\(", added))
    hooks = {"kktsolver_constructor", "kktsolver_defers_constant_rhs", "kktsolver_has_reduced_solve", "kktsolver_kkt_solve_reduced!",
             "kktsolver_set_qb!"}
    assert hooks <= called, hooks - called
    ext = open(os.path.join(JL_DIR, "kktsolver_hip.jl")).read()
    for f in hooks:
        # a generic definition in the patch (default) ...
        assert re.search(r"^\+\s*(function\s+)?" + re.escape(f) + r"\(", open(PATCH).read(), flags=re.M), f
        # ... and a method for HipKKTSolver in the extension
        assert "Clarabel." + f + "(" in ext, f
    # the new struct field is a real hunk: declared, initialised by new(...), set and cleared
    assert "const_pending::Bool" in added and "work_conic,false)" in added
    assert "kktsystem.const_pending = true" in added and "kktsystem.const_pending = false" in added
    # update_q! / update_b! reach the solver (the resident q, b went stale)
    assert added.count("kktsolver_set_qb!(s.kktsystem.kktsolver,s.data.q,s.data.b)") == 2
    # the extension package stanza
    assert 'ClarabelHipKKTExt = "AMDGPU"' in added


@pytest.mark.skipif(not (os.path.isdir("/root/reference/src") and shutil.which("patch")), reason="needs the reference checkout and patch(1)")
def test_core_patch_applies_to_the_reference():
    r = subprocess.run(["patch", "-p1", "--dry-run", "-d", "/root/reference", "-i", PATCH], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    for f in ("src/kktsolvers/kktsolver_defaults.jl", "src/kktsystem.jl", "src/data_updating.jl", "src/kktsolvers/direct-ldl/directldl_auto.jl", "Project.toml"):
        assert f in r.stdout, r.stdout


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="needs the reference checkout")
def test_reference_test_kit_uses_only_names_that_exist():
    """julia/run_reference_tests_hip.jl (the reference's own acceptance tests with the plugin's tokens, for the day a Julia box exists)
    cannot be executed here; what can be checked: every Clarabel name it uses is defined by the reference or by the patch, every
    extension name by julia/ext, every test file it reads exists and contains the text it substitutes."""
    kit = open(os.path.join(ROOT, "julia", "run_reference_tests_hip.jl")).read()
    ref_src = ""
    for dp, _, files in os.walk("/root/reference/src"):
        for f in files:
            if f.endswith(".jl"):
                ref_src += open(os.path.join(dp, f)).read() + "\n"
    patched = ref_src + "\n".join(_patch_added_lines())
    for name in sorted(set(re.findall(r"\bClarabel\.([A-Za-z_]\w*!?)", kit))):
        assert re.search(r"\b" + re.escape(name) + r"\b", patched), f"Clarabel.{name} is not defined by the (patched) reference"
    ext = "".join(open(os.path.join(JL_DIR, f)).read() for f in os.listdir(JL_DIR) if f.endswith(".jl"))
    for name in sorted(set(re.findall(r"\bClarabelHipKKTExt\.([A-Za-z_]\w*!?)", kit)) - {"jl"}):     # ("ClarabelHipKKTExt.jl" is the file)
        assert re.search(r"(function\s+|struct\s+|^)" + re.escape(name) + r"\b", ext, flags=re.M), f"ClarabelHipKKTExt.{name}"
    tests_dir = "/root/reference/test/OptTests"
    for f in re.findall(r'"((?:basic_\w+|linear_solvers)\.jl)"', kit):
        txt = open(os.path.join(tests_dir, f)).read()
        if f == "linear_solvers.jl":
            assert re.search(r"SolverTypes\s*=\s*\[[^\]]*\]", txt)        # the list the kit replaces (linear_solvers.jl:11)
            assert "direct_solve_method = SolverType" in txt
        else:
            assert "Clarabel.Solver(" in txt and "UnitTestFloats" in txt   # the constructor calls the kit redirects; the float list it pins
    assert "get_auto_ldl_solver() === :hip" in kit


def test_extension_checks_the_abi_version():
    lib = open(os.path.join(JL_DIR, "hipkkt_lib.jl")).read()
    hdr = open(os.path.join(ROOT, "include", "hipkkt.h")).read()
    v = re.search(r"#define\s+HIPKKT_ABI_VERSION\s+(\d+)", hdr).group(1)
    assert f"const HIPKKT_ABI_VERSION = Int32({v})" in lib and "hipkkt_abi_version" in lib
    from clarabel_jl_amd import hipkkt
    assert hipkkt.ABI_VERSION == int(v)
