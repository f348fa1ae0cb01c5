# ClarabelHipKKTExt -- the Julia side of libclarabel_hipkkt.so (include/hipkkt.h): the :hip linear-system path of Clarabel.jl
# on AMD MI355X.  Modelled on the reference's optional engines (ext/PardisoExt.jl + ext/directldl_pardiso.jl,
# ext/HSLExt.jl + ext/directldl_hsl.jl): a module that `include`s the plugin files, nothing else.
#
#   hipkkt_lib.jl      library location, `hipkkt_opts` mirror, status / error helpers, device selection
#   directldl_hip.jl   seam L0: HipDirectLDLSolver <: AbstractDirectLDLSolver          (zero edits to Clarabel.jl)
#   kktsolver_hip.jl   seam L1: HipKKTSolver <: AbstractKKTSolver + the widened rows N1 / N2 / N4 of SURVEY.md section 8(f)
#   kktsystem_hip.patch  the edits to src/kktsystem.jl that seam L1 needs (constructor choice; deferred constant-rhs solve;
#                        reduced-system algebra on the device)
#
# Usage (L0, no patch):   using Clarabel; include("julia/ClarabelHipKKTExt/ClarabelHipKKTExt.jl")
#                         settings = Clarabel.Settings(direct_solve_method = :hip)
# There is no `julia` in the build image of this repository: these files have not been executed here.  What IS checked
# (tests/test_julia_glue.py) is that every `ccall` names an exported symbol of include/hipkkt.h with the right number of
# arguments; the call sequences are the ones the Python twin (clarabel.jl_amd/kktsolver.py, hipkkt.py) runs in the GPU tests.
module ClarabelHipKKTExt

include("./hipkkt_lib.jl")
include("./directldl_hip.jl")
include("./kktsolver_hip.jl")

end
