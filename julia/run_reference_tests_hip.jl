# The reference's OWN acceptance tests with the HIP plugin as the linear-system path -- ready for the day a `julia` toolchain sits
# next to an MI355X (there is none in this repository's build image: NOT executed here; tests/test_julia_glue.py checks that every
# name this script uses exists in the patched reference / the extension).
#
#     julia julia/run_reference_tests_hip.jl /path/to/Clarabel.jl [results.json]
#
# What it does, in order:
#   1. applies julia/clarabel_l1_seam.patch to the checkout if it is not applied yet (`patch -p1`; the KKT-solver registry
#      `kktsolver_constructor` of seam L1, the N2 hook, `:hip` in the `:auto` priority list) and activates that checkout's project;
#   2. loads Clarabel and the extension (julia/ext/ClarabelHipKKTExt.jl), checks the ABI version and that a device is visible;
#   3. runs test/OptTests/linear_solvers.jl -- the reference's acceptance test for an LDL plugin (QP, SOCP, SDP through every listed
#      `direct_solve_method`, linear_solvers.jl:17-67) -- with its solver list replaced by [:hip, :hip_ldl, :auto] (:hip = seam L1 on
#      the patched core, :hip_ldl = seam L0, :auto must now resolve to :hip);
#   4. runs test/OptTests/basic_{unconstrained,eq_constrained,lp,qp,socp,sdp}.jl (run_solver_tests.jl:11-18) once per method with
#      every `Clarabel.Solver(P,c,A,b,cones)` of those files constructed with `direct_solve_method = method` (Float64 only: the
#      plugin is a Float64 engine, the BigFloat half of UnitTestFloats stays with :qdldl);
#   5. writes one JSON record per (file, method): pass / fail / error counts -- the file julia/compare_parity.py reads next to
#      the dumps of julia/parity_dump.jl.
# The reference's test files are read from the checkout at run time and evaluated with `include_string`; nothing of them is
# copied into this repository.
using Test, LinearAlgebra, SparseArrays, Random

const REF = length(ARGS) >= 1 ? abspath(ARGS[1]) : error("usage: julia run_reference_tests_hip.jl /path/to/Clarabel.jl [results.json]")
const OUT = length(ARGS) >= 2 ? ARGS[2] : joinpath(@__DIR__, "results", "reference_tests_hip.json")
const PATCH = joinpath(@__DIR__, "clarabel_l1_seam.patch")

# ---- 1. the core patch of seam L1
if !occursin("kktsolver_constructor", read(joinpath(REF, "src", "kktsolvers", "kktsolver_defaults.jl"), String))
    run(pipeline(`patch -p1 -d $REF`; stdin = PATCH))
end
import Pkg
Pkg.activate(REF)
Pkg.instantiate()

# ---- 2. Clarabel + the extension
using Clarabel
include(joinpath(@__DIR__, "ext", "ClarabelHipKKTExt.jl"))
ClarabelHipKKTExt.hip_check_abi()
ClarabelHipKKTExt.hip_is_available() || error("no HIP device visible to libclarabel_hipkkt")
@assert Clarabel.ldlsolver_is_available(:hip) && Clarabel.ldlsolver_is_available(:hip_ldl)
@assert Clarabel.get_auto_ldl_solver() === :hip                       # the patched priority list
@assert Clarabel.kktsolver_constructor(:hip) === ClarabelHipKKTExt.HipKKTSolver
@assert Clarabel.kktsolver_constructor(:qdldl) === Clarabel.DirectLDLKKTSolver

const METHODS = [:hip, :hip_ldl, :auto]
const METHOD = Ref(:hip)
# what `Clarabel.Solver(P,c,A,b,cones)` becomes inside the reference's test files
hip_solver(P, c, A, b, cones) = Clarabel.Solver(P, c, A, b, cones, Clarabel.Settings{eltype(c)}(direct_solve_method = METHOD[]))
function hip_solver(P, c, A, b, cones, settings)
    settings.direct_solve_method = METHOD[]
    return Clarabel.Solver(P, c, A, b, cones, settings)
end

UnitTestFloats = [Float64]                                             # (the files honour an existing definition)
FloatT = Float64
tol = FloatT(1e-3)

results = Dict{String,Any}[]
function run_file(file::String, method::Symbol, transform)
    METHOD[] = method
    src = transform(read(joinpath(REF, "test", "OptTests", file), String))
    ts = @testset "$(file) [$(method)]" begin
        include_string(Main, src, file)
    end
    c = Test.get_test_counts(ts)
    push!(results, Dict("file" => file, "method" => String(method), "passes" => c.passes + c.cumulative_passes,
                        "fails" => c.fails + c.cumulative_fails, "errors" => c.errors + c.cumulative_errors,
                        "broken" => c.broken + c.cumulative_broken))
    return ts
end

# ---- 3. the LDL-plugin acceptance test with the plugin's tokens
run_file("linear_solvers.jl", :hip, s -> replace(s, r"SolverTypes\s*=\s*\[[^\]]*\]" => "SolverTypes = $(METHODS)"))

# ---- 4. the basic problem classes through every `Clarabel.Solver(...)` call of the files
for method in METHODS, file in ("basic_unconstrained.jl", "basic_eq_constrained.jl", "basic_lp.jl", "basic_qp.jl", "basic_socp.jl", "basic_sdp.jl")
    run_file(file, method, s -> replace(s, "Clarabel.Solver(" => "hip_solver("))
end

# ---- 5. the record
mkpath(dirname(OUT))
open(OUT, "w") do io
    println(io, "[")
    for (i, r) in enumerate(results)
        println(io, "  {\"file\": \"$(r["file"])\", \"method\": \"$(r["method"])\", \"passes\": $(r["passes"]), \"fails\": $(r["fails"]), ",
                "\"errors\": $(r["errors"]), \"broken\": $(r["broken"])}", i < length(results) ? "," : "")
    end
    println(io, "]")
end
nbad = sum(r["fails"] + r["errors"] for r in results)
println(nbad == 0 ? "all reference tests pass with the HIP plugin" : "$(nbad) failing checks: see $(OUT)")
exit(nbad == 0 ? 0 : 1)
