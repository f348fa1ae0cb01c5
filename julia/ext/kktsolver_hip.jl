# Seam L1 -- HipKKTSolver <: AbstractKKTSolver  (contract: src/kktsolvers/kktsolver_defaults.jl:2-47; method-for-method twin of
# src/kktsolvers/kktsolver_directldl.jl; Python twin: clarabel.jl_amd/kktsolver.py).  Everything below the method boundary runs
# on the GPU: KKT assembly, Hs / sparse-cone value updates, static regulariser, numeric LDL^T, solves AND iterative refinement.
# What this seam needs from Clarabel.jl is julia/clarabel_l1_seam.patch: a KKT-solver registry next to the LDL one
# (kktsolver_constructor(::Val{S}), used at src/kktsystem.jl:33 instead of the hard-wired DirectLDLKKTSolver) and three optional
# capabilities with defaults in kktsolver_defaults.jl -- all extended from THIS file; the core never refers to the extension.
#
# Beyond the six contract methods, the rows SURVEY.md section 8(f) widens:
#   N1  kktsolver_update_scaled!      update_scaling! + get_Hs! of Zero / Nonnegative / SecondOrder / PSD cones formed from (s, z)
#   N2  kktsolver_solve_multi!        two right-hand sides on one factorisation, concurrently
#       kktsolver_kkt_solve_reduced!  kkt_solve! between the cone algebra and mul_Hs!: dtau dots, quad_form, dx, dz on the device
#   N4  kktsolver_residuals!          residuals_update! from the resident P, A, q, b
import Clarabel: AbstractKKTSolver, CompositeCone, SecondOrderCone, GenPowerCone, ZeroCone, NonnegativeCone, PSDTriangleCone
import Clarabel: get_Hs!, Hs_is_diagonal, is_sparse_expandable, numel
import Clarabel: kktsolver_update!, kktsolver_setrhs!, kktsolver_solve!
import Clarabel: kktsolver_update_P!, kktsolver_update_A!, kktsolver_linear_solver_info

mutable struct HipKKTSolver{T} <: AbstractKKTSolver{T}
    handle::Ptr{Cvoid}
    m::Int; n::Int
    settings
    Hsblocks::Vector{T}             # ref: _allocate_kkt_Hsblocks, directldl_kkt_assembly.jl:3
    soc_u::Vector{T}; soc_v::Vector{T}; soc_eta2::Vector{T}
    w::Vector{T}; λ::Vector{T}; η::Vector{T}      # N1: the device's scaling, in cone order (what mul_Hs! etc. read)
    psd_cones::Vector{Any}
    diagonal_regularizer::T
    has_qb::Bool

    function HipKKTSolver{T}(P::SparseMatrixCSC{T}, A::SparseMatrixCSC{T}, cones, m, n, settings) where {T}
        T === Float64 || error("direct_solve_method = :hip supports Float64 only; use :qdldl for $T")
        hip_is_available() || error("no HIP device visible to libclarabel_hipkkt")
        nc      = length(cones)
        cnumel  = Int64[numel(c) for c in cones]
        cdense  = Int32[Hs_is_diagonal(c) ? 0 : 1 for c in cones]
        # HIPKKT_SPARSE_SOC = 1, HIPKKT_SPARSE_GENPOW = 2 (directldl_datamaps.jl:8-22, 81-99)
        ckind   = Int32[!is_sparse_expandable(c) ? 0 : (c isa SecondOrderCone ? 1 : 2) for c in cones]
        cdim1   = Int64[ckind[i] == 2 ? length(cones[i].α) : 0 for i in 1:nc]
        h = Ref{Ptr{Cvoid}}(C_NULL)
        opts = Ref(hip_default_opts(settings))
        rc = ccall((:hipkkt_create_from_parts, libhipkkt), Int32,
                   (Int32, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64},
                    Int64, Ptr{Int64}, Ptr{Int32}, Ptr{Int32}, Ptr{Int64}, Ref{HipKKTOpts}, Ref{Ptr{Cvoid}}),
                   hip_device(), n, m, P.colptr, P.rowval, P.nzval, A.colptr, A.rowval, A.nzval,
                   nc, cnumel, cdense, ckind, cdim1, opts, h)
        rc == 0 || error("hipkkt_create_from_parts failed ($rc): " * hip_last_error())
        dims = zeros(Int64, 16)
        ccall((:hipkkt_get_dims, libhipkkt), Int32, (Ptr{Cvoid}, Ptr{Int64}), h[], dims)
        nsocrows = sum(cnumel[i] for i in 1:nc if ckind[i] == 1; init = 0)
        nsoc_all = count(c -> c isa SecondOrderCone, cones)
        s = new(h[], m, n, settings, zeros(T, dims[6]), zeros(T, nsocrows), zeros(T, nsocrows),
                zeros(T, count(==(1), ckind)), zeros(T, m), zeros(T, m), zeros(T, nsoc_all),
                Any[c for c in cones if c isa PSDTriangleCone], zero(T), false)
        finalizer(hip_destroy!, s)
        # N1: cone kinds for hipkkt_update_scaling (0 Zero, 1 Nonnegative, 2 SecondOrder, 3 PSDTriangle, -1 = stays with set_hs / set_genpow)
        kinds = Int32[c isa ZeroCone ? 0 : c isa NonnegativeCone ? 1 : c isa SecondOrderCone ? 2 : c isa PSDTriangleCone ? 3 : -1 for c in cones]
        ccall((:hipkkt_set_cone_types, libhipkkt), Int32, (Ptr{Cvoid}, Int64, Ptr{Int32}), s.handle, length(kinds), kinds)
        return s
    end
end

function _hip_refactor!(ks::HipKKTSolver{T}) where {T}          # ref: _kktsolver_regularize_and_refactor!, kktsolver_directldl.jl:247-294
    st = ks.settings; eps = Ref{Float64}(0.0)
    rc = ccall((:hipkkt_refactor, libhipkkt), Int32, (Ptr{Cvoid}, Int32, Float64, Float64, Ref{Float64}, Ptr{Int64}),
               ks.handle, st.static_regularization_enable, st.static_regularization_constant,
               st.static_regularization_proportional, eps, C_NULL)
    ks.diagonal_regularizer = eps[]
    return rc == 0
end

# ref: kktsolver_update!, kktsolver_directldl.jl:197-245
function kktsolver_update!(ks::HipKKTSolver{T}, cones::CompositeCone{T}) where {T}
    get_Hs!(cones, ks.Hsblocks)                                  # :223, cone algebra stays in Julia
    ccall((:hipkkt_set_hs, libhipkkt), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64),
          ks.handle, ks.Hsblocks, length(ks.Hsblocks))           # :225-228 (negation on the device)
    off = 0; k = 0; sparse_idx = 0                               # :235-241
    for cone in cones
        is_sparse_expandable(cone) || continue
        if cone isa SecondOrderCone
            d = numel(cone); k += 1
            ks.soc_u[off+1:off+d] .= cone.sparse_data.u
            ks.soc_v[off+1:off+d] .= cone.sparse_data.v
            ks.soc_eta2[k] = cone.η^2
            off += d
        elseif cone isa GenPowerCone                             # directldl_datamaps.jl:146-167
            dat = cone.data
            ccall((:hipkkt_set_genpow, libhipkkt), Int32, (Ptr{Cvoid}, Int64, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                  ks.handle, sparse_idx, sqrt(dat.μ), dat.p, dat.q, dat.r)
        end
        sparse_idx += 1
    end
    k > 0 && ccall((:hipkkt_set_soc_batch, libhipkkt), Int32,
                   (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64),
                   ks.handle, k, ks.soc_eta2, ks.soc_u, ks.soc_v, off)
    return _hip_refactor!(ks)
end

# N1 (PSD part alone): the packed triu of W (x)_s W formed on the device from W = R R' of every PSD cone (row-major, concatenated),
# hs_off0 = 0-based offset of the cone's block inside Hsblocks -- replaces the skron! loop of get_Hs!, coneops_psdtrianglecone.jl:153-161
function kktsolver_set_hs_psd!(ks::HipKKTSolver{T}, hs_off0::Vector{Int64}, dims::Vector{Int64}, W_all::Vector{T}) where {T}
    rc = ccall((:hipkkt_set_hs_psd, libhipkkt), Int32, (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}),
               ks.handle, length(dims), hs_off0, dims, W_all)
    return rc == 0
end

# N1: the plugin forms update_scaling! + get_Hs! (coneops_nncone.jl:77-101, coneops_socone.jl:75-192,
# coneops_psdtrianglecone.jl:145-161) and the Hs / sparse-cone part of _kktsolver_update_inner! from the iterate (s, z); the PSD cones'
# Cholesky / SVD stay in Julia, only their n x n factor R (column-major = Julia's layout, concatenated) goes down.
# w, λ (length m, cone order) and η (one per second-order cone) come back for mul_Hs! / affine_ds! / combined_ds_shift!.
function kktsolver_update_scaled!(ks::HipKKTSolver{T}, cones, s::Vector{T}, z::Vector{T}) where {T}
    R = isempty(ks.psd_cones) ? C_NULL : reduce(vcat, (vec(K.data.R) for K in ks.psd_cones))
    ok = Ref{Int32}(0)
    rc = ccall((:hipkkt_update_scaling, libhipkkt), Int32,
               (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Int32}),
               ks.handle, s, z, R, ks.w, ks.λ, ks.η, ok)
    (rc == 0 && ok[] == 1) || return false           # ok == 0 <=> update_scaling! would return false (SOC s or z not interior)
    return _hip_refactor!(ks)
end

# ref: kktsolver_setrhs!, kktsolver_directldl.jl:313-327
kktsolver_setrhs!(ks::HipKKTSolver{T}, rhsx::AbstractVector{T}, rhsz::AbstractVector{T}) where {T} =
    (ccall((:hipkkt_setrhs, libhipkkt), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), ks.handle, rhsx, rhsz); nothing)

# ref: kktsolver_solve!, kktsolver_directldl.jl:346-371; lhsx / lhsz may be `nothing` (:330-343) -> C NULL
function kktsolver_solve!(ks::HipKKTSolver{T}, lhsx, lhsz) where {T}
    st = ks.settings
    rc = ccall((:hipkkt_solve, libhipkkt), Int32,
               (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int32, Float64, Float64, Int64, Float64, Ptr{Int64}),
               ks.handle, isnothing(lhsx) ? C_NULL : lhsx, isnothing(lhsz) ? C_NULL : lhsz,
               st.iterative_refinement_enable, st.iterative_refinement_reltol, st.iterative_refinement_abstol,
               st.iterative_refinement_max_iter, st.iterative_refinement_stop_ratio, C_NULL)
    return rc == 0
end

# N2: nrhs right-hand sides on ONE factorisation, two at a time on concurrent device contexts.  rhsx / lhsx are nrhs x n stored
# row-major for the C side, i.e. Julia matrices of size (n, nrhs); likewise rhsz / lhsz (m, nrhs).
function kktsolver_solve_multi!(ks::HipKKTSolver{T}, rhsx::Matrix{T}, rhsz::Matrix{T}, lhsx::Matrix{T}, lhsz::Matrix{T}) where {T}
    st = ks.settings
    rc = ccall((:hipkkt_solve_multi, libhipkkt), Int32,
               (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32, Float64, Float64, Int64, Float64, Ptr{Int64}),
               ks.handle, size(rhsx, 2), rhsx, rhsz, lhsx, lhsz, st.iterative_refinement_enable, st.iterative_refinement_reltol,
               st.iterative_refinement_abstol, st.iterative_refinement_max_iter, st.iterative_refinement_stop_ratio, C_NULL)
    return rc == 0
end

# N2 / N4: q and b resident in the plugin (once after set-up, again from update_q! / update_b!: the hooks of
# julia/clarabel_l1_seam.patch in data_updating.jl:109-150)
function hip_set_qb!(ks::HipKKTSolver{T}, q::AbstractVector{T}, b::AbstractVector{T}) where {T}
    rc = ccall((:hipkkt_set_qb, libhipkkt), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), ks.handle, q, b)
    ks.has_qb = rc == 0
    return ks.has_qb
end

# N2, second half: kkt_solve! (kktsystem.jl:135-215) between the caller's cone algebra (the vector c of Hs dz + ds = -c) and mul_Hs!.
# workx = rhs.x, workz = c - rhs.z, x = variables.x;  const_pending: the constant-rhs solve that kkt_update! left pending runs in the
# same call.  Writes lhs.x, lhs.z and returns (is_success, dtau).  One PCIe round trip, one host synchronisation.
function hip_kkt_solve_reduced!(ks::HipKKTSolver{T}, workx::AbstractVector{T}, workz::AbstractVector{T}, x::AbstractVector{T}, τ::T, κ::T, rhsτ::T, rhsκ::T,
                                const_pending::Bool, lhsx::AbstractVector{T}, lhsz::AbstractVector{T}) where {T}
    st = ks.settings
    scal_in = T[τ, κ, rhsτ, rhsκ]
    scal_out = zeros(T, 10)
    rc = ccall((:hipkkt_kkt_solve_reduced, libhipkkt), Int32,
               (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                Int32, Float64, Float64, Int64, Float64, Ptr{Int64}),
               ks.handle, workx, workz, x, scal_in, const_pending ? 1 : 0, lhsx, lhsz, scal_out,
               st.iterative_refinement_enable, st.iterative_refinement_reltol, st.iterative_refinement_abstol,
               st.iterative_refinement_max_iter, st.iterative_refinement_stop_ratio, C_NULL)
    return (rc == 0, scal_out[1])
end

# N4: residuals_update!(residuals, variables, data), residuals.jl:1-37, from the resident P, A, q, b.  NOT wired into the IPM loop on the
# Julia side: residuals_update! is called from solver.jl:227, which stays untouched; a caller that owns its loop may use it directly.
function kktsolver_residuals!(ks::HipKKTSolver{T}, residuals, variables) where {T}
    scal5 = zeros(T, 5)
    rc = ccall((:hipkkt_residuals, libhipkkt), Int32,
               (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Float64,
                Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
               ks.handle, variables.x, variables.z, variables.s, variables.τ, variables.κ,
               residuals.rx, residuals.rz, residuals.rx_inf, residuals.rz_inf, residuals.Px, scal5)
    rc == 0 || error("hipkkt_residuals failed ($rc): " * hip_last_error(ks.handle))
    residuals.dot_qx, residuals.dot_bz, residuals.dot_sz, residuals.dot_xPx, residuals.rτ = scal5
    return nothing
end

# ref: kktsolver_update_P!/A!, kktsolver_directldl.jl:374-386
kktsolver_update_P!(ks::HipKKTSolver{T}, P::SparseMatrixCSC{T}) where {T} =
    ccall((:hipkkt_update_P, libhipkkt), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64), ks.handle, P.nzval, nnz(P))
kktsolver_update_A!(ks::HipKKTSolver{T}, A::SparseMatrixCSC{T}) where {T} =
    ccall((:hipkkt_update_A, libhipkkt), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64), ks.handle, A.nzval, nnz(A))

kktsolver_linear_solver_info(ks::HipKKTSolver{T}) where {T} = hip_linear_solver_info(ks.handle)


# ---- registration (seam L1): only on a core that carries julia/clarabel_l1_seam.patch.  Same Val-dispatch pattern as
#      ldlsolver_constructor (directldl_defaults.jl:12-30, ext/directldl_pardiso.jl:142-148), extended from the extension.
if isdefined(Clarabel, :kktsolver_constructor)
    Clarabel.kktsolver_constructor(::Val{:hip}) = HipKKTSolver
    Clarabel.kktsolver_defers_constant_rhs(::HipKKTSolver{T}) where {T} = true
    Clarabel.kktsolver_has_reduced_solve(::HipKKTSolver{T}) where {T} = true
    Clarabel.kktsolver_set_qb!(ks::HipKKTSolver{T}, q::AbstractVector{T}, b::AbstractVector{T}) where {T} = (hip_set_qb!(ks, q, b); nothing)
    Clarabel.kktsolver_kkt_solve_reduced!(ks::HipKKTSolver{T}, workx::AbstractVector{T}, workz::AbstractVector{T}, x::AbstractVector{T},
                                          τ::T, κ::T, rhsτ::T, rhsκ::T, const_pending::Bool,
                                          lhsx::AbstractVector{T}, lhsz::AbstractVector{T}) where {T} =
        hip_kkt_solve_reduced!(ks, workx, workz, x, τ, κ, rhsτ, rhsκ, const_pending, lhsx, lhsz)
end
