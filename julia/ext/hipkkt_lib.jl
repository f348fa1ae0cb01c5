# Shared pieces of the two plugin seams: where the library is, how options and errors cross the C ABI.
using SparseArrays, Clarabel
import Clarabel: DefaultInt, LinearSolverInfo

# export CLARABEL_HIPKKT_LIB=/path/to/clarabel.jl_amd/libclarabel_hipkkt.so   (built by clarabel.jl_amd/csrc/build.sh)
const libhipkkt = get(ENV, "CLARABEL_HIPKKT_LIB", "libclarabel_hipkkt.so")

# the GPU a solver lives on: one process (or Julia thread) per GPU drives its own solvers (SURVEY.md section 8e)
hip_device() = Int32(parse(Int, get(ENV, "CLARABEL_HIP_DEVICE", "0")))

# mirrors `struct hipkkt_opts` (include/hipkkt.h)
struct HipKKTOpts
    index_base::Int32; supernode_max_width::Int32; relax_supernodes::Int32
    update_policy::Int32; update_batch::Int32; front_min_panels::Int32
    dynamic_reg_eps::Float64; dynamic_reg_delta::Float64; amd_dense_scale::Float64
    user_perm::Ptr{Int64}
end

function hip_default_opts(settings)
    r = Ref{HipKKTOpts}()
    ccall((:hipkkt_default_opts, libhipkkt), Cvoid, (Ref{HipKKTOpts},), r)
    o = r[]
    # index_base = 1: Julia's 1-based colptr / rowval / index vectors go through as they are
    HipKKTOpts(1, o.supernode_max_width, o.relax_supernodes, o.update_policy, o.update_batch, 0,
               settings.dynamic_regularization_eps, settings.dynamic_regularization_delta,
               1.5, C_NULL)                       # 1.5 = amd_dense_scale of directldl_qdldl.jl:24
end

hip_last_error(handle::Ptr{Cvoid} = C_NULL) =
    unsafe_string(ccall((:hipkkt_last_error, libhipkkt), Cstring, (Ptr{Cvoid},), handle))

# include/hipkkt.h HIPKKT_ABI_VERSION this file was written against: signatures may change between versions, never within one
const HIPKKT_ABI_VERSION = Int32(4)
function hip_check_abi()
    v = ccall((:hipkkt_abi_version, libhipkkt), Int32, ())
    v == HIPKKT_ABI_VERSION || error("libclarabel_hipkkt implements ABI version $v, this extension was written against $HIPKKT_ABI_VERSION")
    return true
end

hip_is_available() = hip_check_abi() && ccall((:hipkkt_is_available, libhipkkt), Int32, ()) > 0

# gives the library's cache of device memory blocks back to the driver (e.g. before another package needs the HBM)
hip_trim_cache() = ccall((:hipkkt_trim_cache, libhipkkt), Int32, (Int32,), hip_device())

function hip_destroy!(x)
    x.handle == C_NULL || ccall((:hipkkt_destroy, libhipkkt), Cvoid, (Ptr{Cvoid},), x.handle)
    x.handle = C_NULL
    return nothing
end

function hip_linear_solver_info(handle::Ptr{Cvoid})
    nnzA = Ref{Int64}(0); nnzL = Ref{Int64}(0)
    ccall((:hipkkt_info, libhipkkt), Int32, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}), handle, nnzA, nnzL)
    LinearSolverInfo(:hip, 1, true, nnzA[], nnzL[])
end
