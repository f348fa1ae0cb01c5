# HIP-vs-REAL-Clarabel parity in one command, for the day a `julia` toolchain sits next to an MI355X:
#
#     python julia/make_config_json.py julia/problems            # the five BASELINE.json configs as Clarabel JSON (N3 wire format)
#     julia --project=/path/to/Clarabel.jl julia/parity_dump.jl julia/problems julia/results
#     python julia/compare_parity.py julia/results                # gate: status equal, iterations +-1, objective / residuals 1e-10
#
# Every problem file is solved twice through the reference's own front end -- direct_solve_method = :qdldl (the reference path) and
# :hip (libclarabel_hipkkt.so behind ClarabelHipKKTExt) -- and x, obj_val, obj_val_dual, r_prim, r_dual, iterations, status and the
# solve time are written to <results>/<name>.<method>.json.  Problems are read with the reference's load_from_file
# (src/json.jl:61-85) and, to pin the wire format both ways, written back with save_to_file (src/json.jl:25-58) next to the results.
# Not executed in this repository's build image (no julia there); tests/test_julia_glue.py checks its ccall-free use of the plugin.
using Clarabel, JSON, SparseArrays, LinearAlgebra
include(joinpath(@__DIR__, "ext", "ClarabelHipKKTExt.jl"))

function solve_and_dump(file::String, method::Symbol, outdir::String)
    solver = Clarabel.load_from_file(file)
    solver.settings.direct_solve_method = method
    solver.settings.verbose = false
    # settings changed after load: build a fresh solver on the same file so that the KKT solver is constructed with them
    fresh = Clarabel.load_from_file(file, solver.settings)
    t0 = time()
    sol = Clarabel.solve!(fresh)
    elapsed = time() - t0
    name = splitext(basename(file))[1]
    out = Dict("name" => name, "method" => String(method), "status" => string(sol.status), "iterations" => sol.iterations,
               "obj_val" => sol.obj_val, "obj_val_dual" => sol.obj_val_dual, "r_prim" => sol.r_prim, "r_dual" => sol.r_dual,
               "solve_time" => sol.solve_time, "wall_time" => elapsed, "x" => sol.x, "z" => sol.z, "s" => sol.s,
               "linear_solver" => string(fresh.info.linsolver.name), "nnzL" => fresh.info.linsolver.nnzL)
    open(joinpath(outdir, "$(name).$(method).json"), "w") do io
        JSON.print(io, out)
    end
    method === :qdldl && Clarabel.save_to_file(fresh, joinpath(outdir, "$(name).roundtrip.json"))
    return out
end

function main(indir::String, outdir::String)
    mkpath(outdir)
    files = sort(filter(f -> endswith(f, ".json"), readdir(indir; join = true)))
    isempty(files) && error("no problem files in $indir (run julia/make_config_json.py first)")
    for f in files
        for method in (:qdldl, :hip)
            if method === :hip && !ClarabelHipKKTExt.hip_is_available()
                @warn "no HIP device visible: skipping :hip for $f"
                continue
            end
            r = solve_and_dump(f, method, outdir)
            println(rpad(r["name"], 12), rpad(r["method"], 8), rpad(r["status"], 10), "it ", r["iterations"], "  obj ", r["obj_val"],
                    "  r_prim ", r["r_prim"], "  r_dual ", r["r_dual"], "  ", round(r["solve_time"]; digits = 3), " s")
        end
    end
end

main(length(ARGS) >= 1 ? ARGS[1] : joinpath(@__DIR__, "problems"), length(ARGS) >= 2 ? ARGS[2] : joinpath(@__DIR__, "results"))
