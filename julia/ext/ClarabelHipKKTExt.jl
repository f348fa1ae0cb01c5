# ClarabelHipKKTExt -- the Julia side of libclarabel_hipkkt.so (include/hipkkt.h): the :hip linear-system path of Clarabel.jl on
# AMD MI355X, laid out as a package extension like the reference's optional engines (ext/PardisoExt.jl + ext/directldl_pardiso.jl,
# ext/HSLExt.jl + ext/directldl_hsl.jl): copy this directory's files into Clarabel.jl's ext/ and add the [weakdeps] / [extensions]
# lines of julia/clarabel_l1_seam.patch to Project.toml; `using Clarabel, AMDGPU` then loads it (AMDGPU.jl is only the trigger: the
# extension itself needs nothing but `ccall`).  Without touching Project.toml: `include("ext/ClarabelHipKKTExt.jl")` after
# `using Clarabel`.
#
#   hipkkt_lib.jl      library location, ABI version check, `hipkkt_opts` mirror, status / error helpers, device selection
#   directldl_hip.jl   seam L0: HipDirectLDLSolver <: AbstractDirectLDLSolver, registered as :hip (and :hip_ldl) by Val dispatch
#                      exactly like :mkl / :panua / :ma57 -- ZERO edits to Clarabel.jl
#   kktsolver_hip.jl   seam L1: HipKKTSolver <: AbstractKKTSolver + the widened rows N1 / N2 / N4 of SURVEY.md section 8(f),
#                      registered as :hip through `kktsolver_constructor(::Val{:hip})` -- the registry julia/clarabel_l1_seam.patch
#                      adds to src/kktsolvers/kktsolver_defaults.jl (same Val-dispatch pattern, extended from here; the core never
#                      names this module).  On an unpatched core that registry does not exist and :hip means seam L0.
#
# Selection:  Clarabel.Settings(direct_solve_method = :hip)      L1 on a patched core, L0 otherwise
#             Clarabel.Settings(direct_solve_method = :hip_ldl)  always L0
# There is no `julia` in the build image of this repository: these files have not been executed here.  What IS checked
# (tests/test_julia_glue.py): every `ccall` names an exported symbol of include/hipkkt.h with the right argument types; the patch
# applies to the reference checkout (`patch --dry-run`) and calls nothing the core or this extension does not define; the core
# patch never mentions this module.  The call sequences are the ones the Python twin (clarabel.jl_amd/kktsolver.py, hipkkt.py)
# runs in the GPU tests.
module ClarabelHipKKTExt

include("./hipkkt_lib.jl")
include("./directldl_hip.jl")
include("./kktsolver_hip.jl")

end
