# Seam L0 -- HipDirectLDLSolver <: AbstractDirectLDLSolver  (contract: src/kktsolvers/direct-ldl/directldl_defaults.jl:1-72;
# pattern: ext/directldl_pardiso.jl:25-148, ext/directldl_hsl.jl:8-110).  ZERO edits to Clarabel.jl: KKT assembly, the static
# regulariser's bookkeeping and iterative refinement stay in DirectLDLKKTSolver (CPU); the value scatter into the resident image,
# the numeric LDL^T and the triangular solves run on the GPU.
import Clarabel: AbstractDirectLDLSolver
import Clarabel: ldlsolver_constructor, ldlsolver_matrix_shape, ldlsolver_is_available
import Clarabel: linear_solver_info, update_values!, scale_values!, refactor!, solve!

mutable struct HipDirectLDLSolver{T} <: AbstractDirectLDLSolver{T}
    handle::Ptr{Cvoid}
    nnzA::DefaultInt
    nvars::DefaultInt

    function HipDirectLDLSolver{T}(KKT::SparseMatrixCSC{T,DefaultInt}, Dsigns, settings) where {T}
        T === Float64 || error("direct_solve_method = :hip supports Float64 only; use :qdldl for $T")
        ldlsolver_is_available(:hip) || error("no HIP device visible to libclarabel_hipkkt")
        h = Ref{Ptr{Cvoid}}(C_NULL)
        opts = Ref(hip_default_opts(settings))
        # symbolic analysis only -- same contract as directldl_qdldl.jl:18-25 (logical = true)
        rc = ccall((:hipkkt_create, libhipkkt), Int32,
                   (Int32, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Int64}, Ref{HipKKTOpts}, Ref{Ptr{Cvoid}}),
                   hip_device(), size(KKT, 1), KKT.colptr, KKT.rowval, KKT.nzval, Vector{Int64}(Dsigns), opts, h)
        rc == 0 || error("hipkkt_create failed ($rc): " * hip_last_error())
        s = new(h[], nnz(KKT), size(KKT, 1))
        finalizer(hip_destroy!, s)            # MOI.empty! calls finalize(solver): MOI_wrapper.jl:133
        return s
    end
end

# registration by Val dispatch (directldl_defaults.jl:12-30, pattern: ext/directldl_pardiso.jl:142-148).  :hip_ldl always means this
# seam; :hip means it on a core without the KKT-solver registry of julia/clarabel_l1_seam.patch (kktsolver_hip.jl otherwise)
for tok in (:hip, :hip_ldl)
    @eval ldlsolver_constructor(::Val{$(QuoteNode(tok))}) = HipDirectLDLSolver
    @eval ldlsolver_matrix_shape(::Val{$(QuoteNode(tok))}) = :triu
    @eval ldlsolver_is_available(::Val{$(QuoteNode(tok))}) = hip_is_available()
end

linear_solver_info(s::HipDirectLDLSolver{T}) where {T} = hip_linear_solver_info(s.handle)

# index may be Vector{Int} or MVector{2,Int} / MVector{3,Int} (directldl_datamaps.jl:11,86)
function update_values!(s::HipDirectLDLSolver{T}, index::AbstractVector{DefaultInt}, values::Vector{T}) where {T}
    idx = index isa Vector{Int64} ? index : collect(Int64, index)
    ccall((:hipkkt_update_values, libhipkkt), Int32, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Float64}, Int64),
          s.handle, idx, values, length(idx))
    return nothing
end

function scale_values!(s::HipDirectLDLSolver{T}, index::AbstractVector{DefaultInt}, scale::T) where {T}
    idx = index isa Vector{Int64} ? index : collect(Int64, index)
    ccall((:hipkkt_scale_values, libhipkkt), Int32, (Ptr{Cvoid}, Ptr{Int64}, Int64, Float64),
          s.handle, idx, length(idx), scale)
    return nothing
end

# DirectLDLKKTSolver has already written the +-eps-shifted diagonal through update_values! (kktsolver_directldl.jl:266-273), so
# the library's own static shift is switched off here.
function refactor!(s::HipDirectLDLSolver{T}, K::SparseMatrixCSC{T}) where {T}
    rc = ccall((:hipkkt_refactor, libhipkkt), Int32,
               (Ptr{Cvoid}, Int32, Float64, Float64, Ptr{Float64}, Ptr{Int64}),
               s.handle, 0, 0.0, 0.0, C_NULL, C_NULL)
    return rc == 0                    # > 0 numerical failure, < 0 device error: both map to `false`
end

function solve!(s::HipDirectLDLSolver{T}, K::SparseMatrixCSC{T}, x::Vector{T}, b::Vector{T}) where {T}
    ccall((:hipkkt_ldl_solve, libhipkkt), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), s.handle, x, b)
    return nothing
end
