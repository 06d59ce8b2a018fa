# ROCMeshField.jl — the reference-side binding a maintainer would add to LevelSetMethods.jl
# (e.g. as ext/ROCMExt.jl with AMDGPU as a weak dependency).  Host code stays Julia; AMDGPU.jl only
# owns the device arrays; every kernel is reached through `ccall` into libhiplsm.so (include/lsm.h).
#
# NEVER EXECUTED IN THIS REPOSITORY: there is no Julia runtime in the build container (SURVEY.md, "Container
# facts").  What is checked mechanically is its C side: tests/test_julia_binding.py parses this file and compares every
# struct (field order and types) and every `ccall` (symbol, return type, argument count and kinds) with include/lsm.h.
# The Python host layer (levelsetmethods.jl_amd/api.py) drives the SAME C ABI call for call and is what the GPU tests
# exercise; this file mirrors it one to one.
#
# Extension point used: the AbstractMeshField interface (src/meshfield.jl:11-33) and the methods
# the integrator reaches a field through (src/timestepping.jl:126-202, src/levelsetterms.jl:22-38).

module ROCMExt

using AMDGPU
using LevelSetMethods
import LevelSetMethods as LSM
using StaticArrays

const libhiplsm = get(ENV, "LSM_AMD_LIB", "libhiplsm.so")
const LSM_GHOST = 3
const LSM_COMM_ID_BYTES = 128

# ---- POD mirrors of include/lsm.h ------------------------------------------------------------
struct LsmGrid
    ndim::Int32
    _pad::Int32
    n::NTuple{3, Int64}
    lc::NTuple{3, Float64}
    hc::NTuple{3, Float64}
end
struct LsmBc
    kind::Int32
    degree::Int32
end
struct LsmSlab
    lo::Int64
    n::Int64
end
struct LsmLayout
    n::NTuple{3, Int64}
    g::NTuple{3, Int64}
    stride::NTuple{3, Int64}
    origin::Int64
    total::Int64
end
struct LsmCoeff
    kind::Int32
    time_kind::Int32
    time_param::Float64
    value::NTuple{4, Float64}
    field::NTuple{3, Ptr{Cvoid}}
    sep::NTuple{3, Ptr{Float64}}
end
struct LsmTerm
    kind::Int32
    scheme::Int32
    coeff::LsmCoeff
    s0::Ptr{Cvoid}
end

_pad3(t, fill) = ntuple(i -> i <= length(t) ? t[i] : fill, 3)
_pad4(t) = ntuple(i -> i <= length(t) ? t[i] : 0.0, 4)

lsmgrid(g::LSM.CartesianGrid{N}) where {N} =
    LsmGrid(N, 0, _pad3(Int64.(g.n), Int64(1)), _pad3(Float64.(g.lc), 0.0), _pad3(Float64.(g.hc), 1.0))

lsmbc(::LSM.PeriodicBC) = LsmBc(0, 0)
lsmbc(::LSM.ExtrapolationBC{P}) where {P} = LsmBc(1, P)
lsmbc(::LSM.SymmetryBC) = LsmBc(2, 0)
const LSM_BC_NONE = LsmBc(3, 0)          # slab interface: ghosts come from the halo exchange

function _check(h, code, what)
    code == 0 && return
    msg = unsafe_string(ccall((:lsm_last_error, libhiplsm), Cstring, (Ptr{Cvoid},), h))
    error("$what failed ($code): $msg")
end

# ---- the handle: one per (mesh, bc, slab, storage type), shared by a field and its copies ----------------
mutable struct Handle
    ptr::Ptr{Cvoid}
    layout::LsmLayout
    strict::Bool
    lo::Int          # first plane (0-based) of the slab along the last dimension; 0 on a whole grid
    rank::Int
    world::Int
end

_dtype(::Type{Float64}) = Cint(0)       # LSM_DTYPE_F64
_dtype(::Type{Float32}) = Cint(1)       # LSM_DTYPE_F32: a storage format, every computation is Float64 (include/lsm.h)

# slab = nothing (whole grid) or (lo, n, rank, world): planes lo+1:lo+n of the last dimension (SURVEY.md §8e)
function Handle(mesh::LSM.CartesianGrid{N}, bcs, ::Type{S}; strict = false, slab = nothing) where {N, S}
    g = Ref(lsmgrid(mesh))
    bc = [d <= N ? lsmbc(bcs[d][s]) : LsmBc(1, 0) for s in 1:2, d in 1:3]   # C order bc[dim][side]
    lo, rank, world = 0, 0, 1
    slabref = C_NULL
    sl = Ref(LsmSlab(0, mesh.n[N]))
    if slab !== nothing
        lo, n, rank, world = slab
        periodic = bcs[N][1] isa LSM.PeriodicBC
        (rank > 0 || (periodic && world > 1)) && (bc[1, N] = LSM_BC_NONE)          # faces towards neighbouring ranks;
        (rank < world - 1 || (periodic && world > 1)) && (bc[2, N] = LSM_BC_NONE)  # both on a periodic ring
        sl[] = LsmSlab(lo, n)
        slabref = Base.unsafe_convert(Ptr{LsmSlab}, sl)
    end
    h = Ref{Ptr{Cvoid}}(C_NULL)
    code = GC.@preserve sl ccall((:lsm_create, libhiplsm), Cint,
        (Ref{LsmGrid}, Ptr{LsmBc}, Ptr{LsmSlab}, Cint, Cint, Cint, Ref{Ptr{Cvoid}}),
        g, bc, slabref, _dtype(S), strict ? 1 : 0, AMDGPU.device_id(AMDGPU.device()) - 1, h)
    _check(C_NULL, code, "lsm_create")
    # kernels run on the stream AMDGPU.jl uses for its own copies
    _check(h[], ccall((:lsm_set_stream, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), h[], AMDGPU.stream().stream), "lsm_set_stream")
    lay = Ref{LsmLayout}()
    _check(h[], ccall((:lsm_layout, libhiplsm), Cint, (Ptr{Cvoid}, Ref{LsmLayout}), h[], lay), "lsm_layout")
    hd = Handle(h[], lay[], strict, lo, rank, world)
    finalizer(x -> ccall((:lsm_destroy, libhiplsm), Cvoid, (Ptr{Cvoid},), x.ptr), hd)   # lsm_destroy detaches a communicator
    return hd
end

# ---- the device field --------------------------------------------------------------------------
"""
    ROCMeshField{N,T,B,S} <: AbstractMeshField{N,T,S}

Dense level-set field resident in HBM in libhiplsm's padded layout (ghost layers materialised).
`S` is the storage type: `Float64`, or `Float32` (LSM_DTYPE_F32 handles: values widen exactly on load,
every computation is Float64, results are rounded once on store).
"""
mutable struct ROCMeshField{N, T, B, S} <: LSM.AbstractMeshField{N, T, S}
    buf::ROCVector{S}
    mesh::LSM.CartesianGrid{N, T}
    bcs::B
    h::Handle
    samples::Dict{UInt, Vector{ROCVector{Float64}}}   # closures sampled on the host into FIELD coefficients (see _coeff)
end

_local(v::AbstractArray{<:Any, N}, slab) where {N} = slab === nothing ? v : collect(selectdim(v, N, (slab[1] + 1):(slab[1] + slab[2])))

function ROCMeshField(ϕ::LSM.MeshField{N, T}; strict = false, slab = nothing) where {N, T}
    LSM._check_bc(ϕ)
    S = eltype(values(ϕ))
    bcs = LSM.boundary_conditions(ϕ)
    h = Handle(LSM.mesh(ϕ), bcs, S; strict, slab)
    buf = AMDGPU.zeros(S, h.layout.total)
    f = ROCMeshField{N, T, typeof(bcs), S}(buf, LSM.mesh(ϕ), bcs, h, Dict{UInt, Vector{ROCVector{Float64}}}())
    host = _local(values(ϕ), slab)
    _check(h.ptr, ccall((:lsm_upload, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), h.ptr, pointer(buf), host), "lsm_upload")
    return check_range(f)
end

# values(ϕ): host copy of the (local) interior on demand (show, hooks, tests)
function Base.values(ϕ::ROCMeshField{N, T, B, S}) where {N, T, B, S}
    out = Array{S, N}(undef, ntuple(d -> Int(ϕ.h.layout.n[d]), N))
    _check(ϕ.h.ptr, ccall((:lsm_download, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), ϕ.h.ptr, pointer(ϕ.buf), out), "lsm_download")
    return out
end
LSM.MeshField(ϕ::ROCMeshField) = LSM.MeshField(values(ϕ), LSM.mesh(ϕ), ϕ.bcs)

# a copy shares the handle (same mesh, bc, slab, storage type, mode and communicator): only the values are new
Base.copy(ϕ::ROCMeshField{N, T, B, S}) where {N, T, B, S} = ROCMeshField{N, T, B, S}(copy(ϕ.buf), ϕ.mesh, ϕ.bcs, ϕ.h, ϕ.samples)
Base.copy!(dst::ROCMeshField, src::ROCMeshField) = (copyto!(dst.buf, src.buf); dst)
LSM.update_band!(::ROCMeshField) = nothing
# scalar indexing: slow path for tests and hooks
Base.getindex(ϕ::ROCMeshField, I::CartesianIndex) = LSM.MeshField(ϕ)[I]

# ---- multi-GPU: one process per GPU, one slab per process (include/lsm.h, "multi-GPU") -----------------------
# rank 0:  id = comm_unique_id();  hand `id` to every rank (MPI.Bcast!, a file, ...)
# each:    ϕ = ROCMeshField(ϕ_host; slab = (lo, n, rank, world));  attach_rccl!(ϕ, id)
# integrate!(eq, tf) then runs unchanged: _advance! -> lsm_advance_* exchanges the ghost planes after every stage
# (overlapped with the interior update), compute_cfl all-reduces Δt.
function comm_unique_id()
    id = Vector{UInt8}(undef, LSM_COMM_ID_BYTES)
    _check(C_NULL, ccall((:lsm_comm_unique_id, libhiplsm), Cint, (Ptr{Cvoid},), id), "lsm_comm_unique_id")
    return id
end
function attach_rccl!(ϕ::ROCMeshField, id::Vector{UInt8})
    _check(ϕ.h.ptr, ccall((:lsm_comm_attach_rccl, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint), ϕ.h.ptr, id, ϕ.h.rank, ϕ.h.world), "lsm_comm_attach_rccl")
    halo!(ϕ)
    return ϕ
end
# every rank as a field of THIS process (one Julia task per rank, or one task driving all ranks stage by stage)
function attach_local!(ϕs::Vector{<:ROCMeshField})
    hs = Ptr{Cvoid}[ϕ.h.ptr for ϕ in ϕs]
    _check(hs[1], ccall((:lsm_comm_attach_local, libhiplsm), Cint, (Ptr{Ptr{Cvoid}}, Cint), hs, length(hs)), "lsm_comm_attach_local")
    return ϕs
end
# ghosts of a field written from outside a step: boundary conditions, then the neighbours' planes
function halo!(ϕ::ROCMeshField)
    _check(ϕ.h.ptr, ccall((:lsm_fill_ghosts, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}), ϕ.h.ptr, pointer(ϕ.buf), 7, C_NULL), "lsm_fill_ghosts")
    ϕ.h.world > 1 && _check(ϕ.h.ptr, ccall((:lsm_halo_exchange, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ϕ.h.ptr, pointer(ϕ.buf)), "lsm_halo_exchange")
    return ϕ
end
# the two halves, for a driver that overlaps the exchange itself (lsm_stage_planes on the interface planes, start, interior, wait)
halo_start!(ϕ::ROCMeshField) = _check(ϕ.h.ptr, ccall((:lsm_halo_start, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ϕ.h.ptr, pointer(ϕ.buf)), "lsm_halo_start")
halo_wait!(ϕ::ROCMeshField) = _check(ϕ.h.ptr, ccall((:lsm_halo_wait, libhiplsm), Cint, (Ptr{Cvoid},), ϕ.h.ptr), "lsm_halo_wait")
set_overlap!(ϕ::ROCMeshField, on::Bool) = _check(ϕ.h.ptr, ccall((:lsm_comm_set_overlap, libhiplsm), Cint, (Ptr{Cvoid}, Cint), ϕ.h.ptr, on), "lsm_comm_set_overlap")
detach!(ϕ::ROCMeshField) = _check(ϕ.h.ptr, ccall((:lsm_comm_detach, libhiplsm), Cint, (Ptr{Cvoid},), ϕ.h.ptr), "lsm_comm_detach")
function comm_info(ϕ::ROCMeshField)
    r, w, t = Ref{Cint}(), Ref{Cint}(), Ref{Cint}()
    _check(ϕ.h.ptr, ccall((:lsm_comm_info, libhiplsm), Cint, (Ptr{Cvoid}, Ref{Cint}, Ref{Cint}, Ref{Cint}), ϕ.h.ptr, r, w, t), "lsm_comm_info")
    return (rank = Int(r[]), world = Int(w[]), transport = Int(t[]))
end

# ---- terms -> LsmTerm ---------------------------------------------------------------------------
# Julia closures cannot run on the device (src/levelsetterms.jl:42-43 evaluates `f(getnode(ϕ, I), t)` per node):
# constants, catalogued analytic fields and device fields are passed through; a closure is sampled on the host into a
# FIELD coefficient — at construction of the LsmTerm and again, by the stage hook, before every stage at its stage time.
struct RigidRotation; ω::Float64; c::NTuple{2, Float64}; end     # u = ω·(-(x₂-c₂), x₁-c₁, 0)

const _NOFIELD = (C_NULL, C_NULL, C_NULL)
const _NOSEP = (Ptr{Float64}(C_NULL), Ptr{Float64}(C_NULL), Ptr{Float64}(C_NULL))
_coeff(v::Number, ϕ, t, K) = LsmCoeff(0, 0, 1.0, (Float64(v), 0.0, 0.0, 0.0), _NOFIELD, _NOSEP)
_coeff(v::Union{SVector, Tuple{Vararg{Number}}}, ϕ, t, K) = LsmCoeff(0, 0, 1.0, _pad4(Float64.(Tuple(v))), _NOFIELD, _NOSEP)
_coeff(r::RigidRotation, ϕ, t, K) = LsmCoeff(1, 0, 1.0, (r.ω, r.c[1], r.c[2], 0.0), _NOFIELD, _NOSEP)
# device fields: the library reads coefficient fields as Float64 side arrays in ϕ's padded layout
_f64(f::ROCMeshField{N, T, B, Float64}, ϕ) where {N, T, B} = f.buf
_f64(f::ROCMeshField, ϕ) = get!(() -> [ROCVector{Float64}(undef, length(f.buf))], ϕ.samples, objectid(f))[1] .= f.buf   # a Float32 field widens exactly
_coeff(f::ROCMeshField, ϕ, t, K) = LsmCoeff(3, 0, 1.0, (0.0, 0.0, 0.0, 0.0), (Ptr{Cvoid}(pointer(_f64(f, ϕ))), C_NULL, C_NULL), _NOSEP)
_coeff(fs::Tuple{Vararg{ROCMeshField}}, ϕ, t, K) =
    LsmCoeff(3, 0, 1.0, (0.0, 0.0, 0.0, 0.0), _pad3(map(f -> Ptr{Cvoid}(pointer(_f64(f, ϕ))), fs), C_NULL), _NOSEP)
# a closure f(x, t) -> Number | SVector: K components sampled at the local nodes, uploaded into arrays that live as long as ϕ's handle
function _coeff(f::Function, ϕ, t, K)      # ϕ::ROCField (dense or band: both carry mesh, h, samples)
    N = length(ϕ.mesh.n)
    arrs = get!(() -> [AMDGPU.zeros(Float64, ϕ.h.layout.total) for _ in 1:K], ϕ.samples, objectid(f))
    nloc = ntuple(d -> Int(ϕ.h.layout.n[d]), N)
    off = ntuple(d -> d == N ? ϕ.h.lo : 0, N)                   # a slab's nodes sit at global indices lo+1 : lo+n
    host = [Array{Float64, N}(undef, nloc) for _ in 1:K]
    for I in CartesianIndices(nloc)
        v = f(LSM.getnode(LSM.mesh(ϕ), I + CartesianIndex(off)), t)
        for k in 1:K
            host[k][I] = v[k]          # a Number indexes as v[1]
        end
    end
    for k in 1:K
        _check(ϕ.h.ptr, ccall((:lsm_upload_f64, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), ϕ.h.ptr, pointer(arrs[k]), host[k]), "lsm_upload_f64")
    end
    return LsmCoeff(3, 0, 1.0, (0.0, 0.0, 0.0, 0.0), _pad3(map(a -> Ptr{Cvoid}(pointer(a)), Tuple(arrs)), C_NULL), _NOSEP)
end
_needs_sampling(c) = c isa Function

_term(t::LSM.AdvectionTerm, ϕ, tt) = LsmTerm(0, LSM.scheme(t) isa LSM.WENO5 ? 1 : 0, _coeff(LSM.velocity(t), ϕ, tt, length(ϕ.mesh.n)), C_NULL)
_term(t::LSM.NormalMotionTerm, ϕ, tt) = LsmTerm(1, 0, _coeff(LSM.speed(t), ϕ, tt, 1), C_NULL)
_term(t::LSM.CurvatureTerm, ϕ, tt) = LsmTerm(2, 0, _coeff(LSM.coefficient(t), ϕ, tt, 1), C_NULL)
_term(t::LSM.EikonalReinitializationTerm{Nothing}, ϕ, tt) = LsmTerm(3, 0, _coeff(0.0, ϕ, tt, 1), C_NULL)
_term(t::LSM.EikonalReinitializationTerm{<:ROCVector{Float64}}, ϕ, tt) = LsmTerm(3, 0, _coeff(0.0, ϕ, tt, 1), Ptr{Cvoid}(pointer(t.S₀)))
_coefficient(t::LSM.AdvectionTerm) = LSM.velocity(t)
_coefficient(t::LSM.NormalMotionTerm) = LSM.speed(t)
_coefficient(t::LSM.CurvatureTerm) = LSM.coefficient(t)
_coefficient(::LSM.EikonalReinitializationTerm) = nothing

# EikonalReinitializationTerm(ϕ₀::ROCMeshField): S₀ = ϕ₀/√(ϕ₀²+Δx²) on the device (src/levelsetterms.jl:217-221), a Float64 side array
function LSM.EikonalReinitializationTerm(ϕ₀::ROCMeshField)
    S₀ = ROCVector{Float64}(undef, length(ϕ₀.buf))
    _check(ϕ₀.h.ptr, ccall((:lsm_eikonal_sign, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        ϕ₀.h.ptr, pointer(ϕ₀.buf), pointer(S₀), C_NULL), "lsm_eikonal_sign")
    return LSM.EikonalReinitializationTerm{typeof(S₀)}(S₀)
end

# ---- the methods the integrator dispatches on -----------------------------------------------------
# compute_cfl (src/levelsetterms.jl:22-28): the library returns the raw local minimum; across slabs it is min-reduced
# with NaN winning; Julia throws.
function LSM.compute_cfl(terms, ϕ::ROCMeshField, t)
    ts = [_term(term, ϕ, t) for term in terms]
    dt = Ref{Float64}(0.0)
    _check(ϕ.h.ptr, ccall((:lsm_compute_cfl, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{LsmTerm}, Cint, Ptr{Cvoid}, Float64, Ref{Float64}),
        ϕ.h.ptr, ts, length(ts), pointer(ϕ.buf), t, dt), "lsm_compute_cfl")
    ϕ.h.world > 1 && _check(ϕ.h.ptr, ccall((:lsm_allreduce_dt, libhiplsm), Cint, (Ptr{Cvoid}, Ref{Float64}), ϕ.h.ptr, dt), "lsm_allreduce_dt")
    Δt = dt[]
    Δt > 0 || throw(ArgumentError("invalid time-step based on CFL condition: Δt = $Δt (check for NaN/Inf in velocity or speed)"))
    return Δt
end

# stage hook: lets update_func(field, stage_field, stage_time) run — and closure coefficients be re-sampled at the stage
# time — between stage launches (src/timestepping.jl:131,149,158,174,185,196).  NULL when every term has the default no-op
# update_func (src/levelsetterms.jl:63,141,146: `(_...) -> nothing`, one anonymous function per constructor) and no closure.
const _NOOPS = (typeof(LSM.AdvectionTerm(0).update_func), typeof(LSM.NormalMotionTerm(0).update_func))
_is_noop(t) = !hasproperty(t, :update_func) || typeof(t.update_func) in _NOOPS
function _hook(terms, fields)
    all(_is_noop, terms) && !any(t -> _needs_sampling(_coefficient(t)), terms) && return C_NULL
    cb = (user, stage, ptr, tstage) -> begin
        ψ = fields[stage + 1]
        for term in terms
            LSM.update_term!(term, ψ, tstage)
            _needs_sampling(_coefficient(term)) && _term(term, ψ, tstage)     # refresh the sampled arrays in place
        end
        Cint(0)
    end
    return @cfunction($cb, Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Float64))
end

LSM._alloc_buffers(::LSM.ForwardEuler, ϕ::ROCMeshField) = (copy(ϕ),)
LSM._alloc_buffers(::Union{LSM.RK2, LSM.RK3}, ϕ::ROCMeshField) = (copy(ϕ), copy(ϕ))

# On a slab handle with a communicator attached the same three calls run the slab's step: interface planes first, ghost
# planes exchanged over RCCL while the interior is updated (include/lsm.h).
function LSM._advance!(::LSM.ForwardEuler, ϕ::ROCMeshField, (dst,), terms, tc, Δt)
    ts = [_term(term, ϕ, tc) for term in terms]
    hook = _hook(terms, (ϕ,))
    GC.@preserve hook _check(ϕ.h.ptr, ccall((:lsm_advance_fe, libhiplsm), Cint,
        (Ptr{Cvoid}, Ptr{LsmTerm}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Ptr{Cvoid}, Ptr{Cvoid}),
        ϕ.h.ptr, ts, length(ts), pointer(ϕ.buf), pointer(dst.buf), tc, Δt, hook, C_NULL), "lsm_advance_fe")
    return ϕ
end
function LSM._advance!(::LSM.RK2, ϕ::ROCMeshField, (pred, corr), terms, tc, Δt)
    ts = [_term(term, ϕ, tc) for term in terms]
    hook = _hook(terms, (ϕ, pred))
    GC.@preserve hook _check(ϕ.h.ptr, ccall((:lsm_advance_rk2, libhiplsm), Cint,
        (Ptr{Cvoid}, Ptr{LsmTerm}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Ptr{Cvoid}, Ptr{Cvoid}),
        ϕ.h.ptr, ts, length(ts), pointer(ϕ.buf), pointer(pred.buf), pointer(corr.buf), tc, Δt, hook, C_NULL), "lsm_advance_rk2")
    return ϕ
end
function LSM._advance!(::LSM.RK3, ϕ::ROCMeshField, (buf1, buf2), terms, tc, Δt)
    ts = [_term(term, ϕ, tc) for term in terms]
    hook = _hook(terms, (ϕ, buf1, buf2))
    GC.@preserve hook _check(ϕ.h.ptr, ccall((:lsm_advance_rk3, libhiplsm), Cint,
        (Ptr{Cvoid}, Ptr{LsmTerm}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Ptr{Cvoid}, Ptr{Cvoid}),
        ϕ.h.ptr, ts, length(ts), pointer(ϕ.buf), pointer(buf1.buf), pointer(buf2.buf), tc, Δt, hook, C_NULL), "lsm_advance_rk3")
    return ϕ
end

# domain of the FAST arithmetic mode (include/lsm.h, LSM_FAST_MAX_ABS): called when a field is handed to an equation
function check_range(ϕ::ROCMeshField)
    ok, m = Ref{Cint}(), Ref{Float64}()
    _check(ϕ.h.ptr, ccall((:lsm_check_range, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Cint}, Ref{Float64}), ϕ.h.ptr, pointer(ϕ.buf), ok, m), "lsm_check_range")
    ok[] != 0 || throw(ArgumentError("max|ϕ| = $(m[]) is outside the domain of the FAST arithmetic mode; rescale the field or use ROCMeshField(ϕ; strict = true)"))
    return ϕ
end

# show needs extrema (src/meshfield.jl:300-303)
function Base.extrema(ϕ::ROCMeshField)
    lo, hi = Ref{Float64}(), Ref{Float64}()
    _check(ϕ.h.ptr, ccall((:lsm_extrema, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}, Ref{Float64}), ϕ.h.ptr, pointer(ϕ.buf), lo, hi), "lsm_extrema")
    return lo[], hi[]
end

# ---- next rows (SURVEY.md §8f) ------------------------------------------------------------------

# volume / perimeter (src/levelsetops.jl:139-149,171-183) of the local slab (sum over the ranks for a decomposed grid)
function LSM.volume(ϕ::ROCMeshField)
    out = Ref{Float64}()
    _check(ϕ.h.ptr, ccall((:lsm_volume, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}), ϕ.h.ptr, pointer(ϕ.buf), out), "lsm_volume")
    return out[]
end
function LSM.perimeter(ϕ::ROCMeshField)
    out = Ref{Float64}()
    _check(ϕ.h.ptr, ccall((:lsm_perimeter, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}), ϕ.h.ptr, pointer(ϕ.buf), out), "lsm_perimeter")
    return out[]
end

# volume / perimeter of a band field (src/levelsetops.jl:34-116,150-166); `mask` is the band's byte mask, ϕ prepared
# (lsm_band_prepare) for the perimeter's centred gradient
function band_volume(ϕ::ROCMeshField, mask::ROCVector{UInt8})
    out = Ref{Float64}()
    _check(ϕ.h.ptr, ccall((:lsm_band_volume, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}), ϕ.h.ptr, pointer(ϕ.buf), pointer(mask), out), "lsm_band_volume")
    return out[]
end
function band_perimeter(ϕ::ROCMeshField, mask::ROCVector{UInt8})
    out = Ref{Float64}()
    _check(ϕ.h.ptr, ccall((:lsm_band_perimeter, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}), ϕ.h.ptr, pointer(ϕ.buf), pointer(mask), out), "lsm_band_perimeter")
    return out[]
end

# extend_along_normals!(F, ϕ; ...) (src/velocityextension.jl:20-116); frozen = nothing -> band rule
function LSM.extend_along_normals!(F::ROCMeshField, ϕ::ROCMeshField; nb_iters = 50, cfl = 0.45, frozen = nothing,
                                   interface_band = 1.5, min_norm = 1.0e-14)
    N = length(ϕ.mesh.n)
    work = [similar(ϕ.buf) for _ in 1:(N + 1)]
    w = [i <= length(work) ? pointer(work[i]) : C_NULL for i in 1:4]
    _check(ϕ.h.ptr, ccall((:lsm_extend_along_normals, libhiplsm), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Float64, Float64, Float64),
        ϕ.h.ptr, pointer(F.buf), pointer(ϕ.buf), frozen === nothing ? C_NULL : pointer(frozen), w[1], w[2], w[3], w[4],
        nb_iters, cfl, interface_band, min_norm), "lsm_extend_along_normals")
    _check(ϕ.h.ptr, ccall((:lsm_sync, libhiplsm), Cint, (Ptr{Cvoid},), ϕ.h.ptr), "lsm_sync")   # work buffers die here
    return F
end

# curvature / gradient / normal of ϕ at every node (src/levelsetops.jl:197-226) into Float64 device arrays in ϕ's padded
# layout; `band`: only |ϕ[I]| <= band is evaluated (else `fill`) and `frozen` marks those nodes with 1.0 — the seed loop of
# the reference's speed update functions (test/test-velocityextension.jl:118-131) without a host pass.
function curvature_field!(out::ROCVector{Float64}, ϕ::ROCMeshField; scale = 1.0, band = -1.0, fill = 0.0, frozen = nothing)
    _check(ϕ.h.ptr, ccall((:lsm_geometry, libhiplsm), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Float64, Float64, Float64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        ϕ.h.ptr, 0, pointer(ϕ.buf), scale, band, fill, pointer(out), C_NULL, C_NULL, frozen === nothing ? C_NULL : pointer(frozen), C_NULL),
        "lsm_geometry")
    return out
end
function _vector_field!(what::Integer, outs::NTuple{N, ROCVector{Float64}}, ϕ::ROCMeshField; scale = 1.0) where {N}
    o = ntuple(d -> d <= N ? pointer(outs[d]) : C_NULL, 3)
    _check(ϕ.h.ptr, ccall((:lsm_geometry, libhiplsm), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Float64, Float64, Float64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        ϕ.h.ptr, what, pointer(ϕ.buf), scale, -1.0, 0.0, o[1], o[2], o[3], C_NULL, C_NULL), "lsm_geometry")
    return outs
end
# band fields: the same over the active nodes of a prepared band (`mask` = the band's byte mask)
function band_curvature_field!(out::ROCVector{Float64}, ϕ::ROCMeshField, mask::ROCVector{UInt8}; scale = 1.0, band = -1.0, fill = 0.0, frozen = nothing)
    _check(ϕ.h.ptr, ccall((:lsm_band_geometry, libhiplsm), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Float64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        ϕ.h.ptr, 0, pointer(ϕ.buf), pointer(mask), scale, band, fill, pointer(out), C_NULL, C_NULL, frozen === nothing ? C_NULL : pointer(frozen), C_NULL),
        "lsm_band_geometry")
    return out
end
gradient_field!(outs, ϕ::ROCMeshField; kw...) = _vector_field!(1, outs, ϕ; kw...)
normal_field!(outs, ϕ::ROCMeshField; kw...) = _vector_field!(2, outs, ϕ; kw...)

# InterpolatedField(ϕ, order) evaluated at points (src/interpolation.jl:117-151,228-260): `pts` is an ndim x npts device matrix
# (column = point); returns values, and fills `grad` (ndim x npts) / `hess` (ndim x ndim x npts) when given
function interpolate!(val::ROCVector{Float64}, ϕ::ROCMeshField, order::Integer, pts::ROCMatrix{Float64}; grad = nothing, hess = nothing)
    _check(ϕ.h.ptr, ccall((:lsm_interpolate, libhiplsm), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        ϕ.h.ptr, pointer(ϕ.buf), order, size(pts, 2), pointer(pts), pointer(val), grad === nothing ? C_NULL : pointer(grad),
        hess === nothing ? C_NULL : pointer(hess), C_NULL), "lsm_interpolate")
    return val
end

# NewtonSDF(ϕ; ...) (src/sdf.jl:57-127) as a device object: build once, query points, read the samples back
mutable struct ROCNewtonSDF
    ptr::Ptr{Cvoid}
    h::Handle
    nsamples::Int64
end
function ROCNewtonSDF(ϕ::ROCMeshField; order = 3, upsample = 2, maxiters = 10, xtol = nothing, ftol = nothing, mask = nothing)
    xt, ft = something(xtol, sqrt(eps(Float64))), something(ftol, sqrt(eps(Float64)))
    out, ns = Ref{Ptr{Cvoid}}(), Ref{Int64}()
    _check(ϕ.h.ptr, ccall((:lsm_sdf_create, libhiplsm), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Float64, Float64, Ref{Ptr{Cvoid}}, Ref{Int64}),
        ϕ.h.ptr, pointer(ϕ.buf), mask === nothing ? C_NULL : pointer(mask), order, upsample, maxiters, xt, ft, out, ns), "lsm_sdf_create")
    sdf = ROCNewtonSDF(out[], ϕ.h, ns[])
    finalizer(s -> ccall((:lsm_sdf_destroy, libhiplsm), Cvoid, (Ptr{Cvoid},), s.ptr), sdf)
    return sdf
end
# signed distances at the columns of `pts` (ndim x npts device matrix); `cp` optionally receives the closest points
function (sdf::ROCNewtonSDF)(dist::ROCVector{Float64}, pts::ROCMatrix{Float64}; cp = nothing)
    nfail = Ref{Int64}()
    _check(sdf.h.ptr, ccall((:lsm_sdf_eval, libhiplsm), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Int64}),
        sdf.ptr, size(pts, 2), pointer(pts), pointer(dist), cp === nothing ? C_NULL : pointer(cp), nfail), "lsm_sdf_eval")
    return dist
end
function sample_points(sdf::ROCNewtonSDF, ndim::Integer)
    out = ROCMatrix{Float64}(undef, ndim, sdf.nsamples)
    _check(sdf.h.ptr, ccall((:lsm_sdf_samples, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), sdf.ptr, pointer(out)), "lsm_sdf_samples")
    return out
end

# reinitialize!(ϕ; ...) (src/reinitializer.jl:12-42)
function LSM.reinitialize!(ϕ::ROCMeshField; order = 3, upsample = 2, maxiters = 20, xtol = nothing, ftol = nothing)
    xt, ft = something(xtol, sqrt(eps(Float64))), something(ftol, sqrt(eps(Float64)))
    _check(ϕ.h.ptr, ccall((:lsm_fill_ghosts, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}), ϕ.h.ptr, pointer(ϕ.buf), 7, C_NULL), "lsm_fill_ghosts")
    work = similar(ϕ.buf)
    nc, nfail, nfar = Ref{Int64}(), Ref{Int64}(), Ref{Int64}()
    _check(ϕ.h.ptr, ccall((:lsm_reinitialize, libhiplsm), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Float64, Float64, Ref{Int64}, Ref{Int64}, Ref{Int64}),
        ϕ.h.ptr, pointer(ϕ.buf), C_NULL, pointer(work), order, upsample, maxiters, xt, ft, nc, nfail, nfar), "lsm_reinitialize")
    nfail[] > 0 && @warn "reinitialize!: closest-point solver did not converge for $(nfail[]) points"
    return ϕ
end

# ---- NarrowBandMeshField on the device (src/meshfield.jl:314-588): dense padded values + byte masks --------------------
# The reference steps a band field with the SAME _advance! / compute_cfl / update_band! sequence as a dense one
# (src/timestepping.jl:101-122,126-202; src/meshfield.jl:555-588); the methods below are that sequence for a device band.
const BAND_MC = 16            # planes per tile along the last dimension (8, 12, 16, 24, 32 measured at 768³: 16 is the fastest)
const BAND_OVERLAP = 10       # planes of each neighbouring rank a slab of a band holds (include/lsm.h, lsm_band_overlap_config)
mutable struct ROCNarrowBandMeshField{N, T, B, S} <: LSM.AbstractMeshField{N, T, S}
    buf::ROCVector{S}
    mesh::LSM.CartesianGrid{N, T}
    bcs::B
    h::Handle
    samples::Dict{UInt, Vector{ROCVector{Float64}}}
    nlayers::Int
    mask::ROCVector{UInt8}
    halo::ROCVector{UInt8}
    tiles::ROCVector{UInt8}
    scratch::NTuple{2, ROCVector{UInt8}}
    hlist::ROCVector{Int64}       # (halo node -> nearest band node) entries, 2 Int64 each
    hcount::ROCVector{UInt32}
    own::Tuple{Int, Int}          # (first local plane, count) of the planes this rank owns; the rest are overlap planes
end
const ROCField = Union{ROCMeshField, ROCNarrowBandMeshField}

function _tile_count(h::Handle)
    n = Ref{Int64}()
    _check(h.ptr, ccall((:lsm_band_tile_count, libhiplsm), Cint, (Ptr{Cvoid}, Cint, Ref{Int64}), h.ptr, BAND_MC, n), "lsm_band_tile_count")
    return Int(n[])
end
function _empty_band(buf::ROCVector{S}, mesh::LSM.CartesianGrid{N, T}, bcs, h, nlayers, own) where {N, T, S}
    total = h.layout.total
    bytes() = AMDGPU.zeros(UInt8, total)
    return ROCNarrowBandMeshField{N, T, typeof(bcs), S}(buf, mesh, bcs, h, Dict{UInt, Vector{ROCVector{Float64}}}(), nlayers, bytes(), bytes(),
        AMDGPU.zeros(UInt8, _tile_count(h)), (bytes(), bytes()), ROCVector{Int64}(undef, 2 * (1 << 16)), AMDGPU.zeros(UInt32, 1), own)
end

# NarrowBandMeshField(ϕ; nlayers) (src/meshfield.jl:411-440) on the device.  slab = (lo, n, rank, world): the OWNED planes;
# the handle's slab is extended by BAND_OVERLAP planes of each neighbour (attach a communicator, then band_overlap!).
function ROCNarrowBandMeshField(nb::LSM.NarrowBandMeshField{N, T}; strict = false, slab = nothing) where {N, T}
    bcs = LSM.boundary_conditions(nb)
    any(bc -> bc isa LSM.PeriodicBC, Iterators.flatten(bcs)) && throw(ArgumentError("PeriodicBC is not supported on a NarrowBandMeshField"))
    dense = LSM.MeshField(nb)                      # every node: the band is re-derived on the device (from_dense)
    S = eltype(values(dense))
    ext, own = slab, (0, LSM.mesh(nb).n[N])
    if slab !== nothing
        lo, n, rank, world = slab
        wlo, whi = rank > 0 ? BAND_OVERLAP : 0, rank < world - 1 ? BAND_OVERLAP : 0
        ext, own = (lo - wlo, n + wlo + whi, rank, world), (wlo, n)
    end
    h = Handle(LSM.mesh(nb), bcs, S; strict, slab = ext)
    ϕ = _empty_band(AMDGPU.zeros(S, h.layout.total), LSM.mesh(nb), bcs, h, nb.nlayers, own)
    _check(h.ptr, ccall((:lsm_upload, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), h.ptr, pointer(ϕ.buf), _local(values(dense), ext)), "lsm_upload")
    slab === nothing && LSM.update_band!(ϕ; from_dense = true)     # a slab builds its band in band_overlap!, once attached
    return ϕ
end
# after attach_rccl! / attach_local! on the band fields of all ranks: declare the overlap, build the band, take the overlap
# planes from their owners
function band_overlap!(ϕ::ROCNarrowBandMeshField)
    _check(ϕ.h.ptr, ccall((:lsm_band_overlap_config, libhiplsm), Cint, (Ptr{Cvoid}, Int64), ϕ.h.ptr, BAND_OVERLAP), "lsm_band_overlap_config")
    return LSM.update_band!(ϕ; from_dense = true)
end
attach_rccl!(ϕ::ROCNarrowBandMeshField, id::Vector{UInt8}) =
    (_check(ϕ.h.ptr, ccall((:lsm_comm_attach_rccl, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint), ϕ.h.ptr, id, ϕ.h.rank, ϕ.h.world), "lsm_comm_attach_rccl"); band_overlap!(ϕ))
# a rank whose hook threw: the other ranks' exchanges return LSM_ERR_COMM instead of waiting for it
comm_abort!(ϕ::ROCField) = _check(ϕ.h.ptr, ccall((:lsm_comm_abort, libhiplsm), Cint, (Ptr{Cvoid},), ϕ.h.ptr), "lsm_comm_abort")

# a copy shares the handle; the band set travels with it (the Dict copy of src/meshfield.jl:420-423)
function Base.copy(ϕ::ROCNarrowBandMeshField{N, T, B, S}) where {N, T, B, S}
    return ROCNarrowBandMeshField{N, T, B, S}(copy(ϕ.buf), ϕ.mesh, ϕ.bcs, ϕ.h, ϕ.samples, ϕ.nlayers, copy(ϕ.mask), copy(ϕ.halo), copy(ϕ.tiles),
        (similar(ϕ.mask), similar(ϕ.mask)), copy(ϕ.hlist), copy(ϕ.hcount), ϕ.own)
end
function Base.copy!(dst::ROCNarrowBandMeshField, src::ROCNarrowBandMeshField)      # src/meshfield.jl:282-292: values AND keys
    copyto!(dst.buf, src.buf); copyto!(dst.mask, src.mask); copyto!(dst.halo, src.halo); copyto!(dst.tiles, src.tiles)
    dst.hlist = copy(src.hlist); copyto!(dst.hcount, src.hcount)
    # the handle's compact tile lists, the halo list's length as the host knows it and a prefetched Δt describe the band these
    # buffers held BEFORE: rebuild them from the new contents (api.py's copy_ does the same; include/lsm.h, lsm_band_invalidate)
    _check(dst.h.ptr, ccall((:lsm_band_retile, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint), dst.h.ptr, pointer(dst.mask), pointer(dst.tiles), BAND_MC), "lsm_band_retile")
    _band_status(dst)
    return dst
end
function Base.values(ϕ::ROCNarrowBandMeshField{N, T, B, S}) where {N, T, B, S}
    out = Array{S, N}(undef, ntuple(d -> Int(ϕ.h.layout.n[d]), N))
    _check(ϕ.h.ptr, ccall((:lsm_download, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), ϕ.h.ptr, pointer(ϕ.buf), out), "lsm_download")
    return out
end

struct LsmBand
    mask::Ptr{Cvoid}
    tiles::Ptr{Cvoid}
    mc::Int32
    _pad::Int32
    halo_list::Ptr{Cvoid}
    halo_cap::Int64
    halo_count::Ptr{Cvoid}
end
_band(ϕ::ROCNarrowBandMeshField) = Ref(LsmBand(pointer(ϕ.mask), pointer(ϕ.tiles), BAND_MC, 0, pointer(ϕ.hlist), length(ϕ.hlist) ÷ 2, pointer(ϕ.hcount)))

function _band_status(ϕ::ROCNarrowBandMeshField)
    want, missed = Ref{Int64}(), Ref{Cint}()
    _check(ϕ.h.ptr, ccall((:lsm_band_status, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Int64}, Ref{Cint}),
        ϕ.h.ptr, pointer(ϕ.hcount), want, missed), "lsm_band_status")
    missed[] != 0 && throw(ArgumentError("index is more than $(LSM._BAND_SEARCH_RADIUS) nodes from the band"))   # src/meshfield.jl:499-500
    return Int(want[])
end
_band_halo(ϕ::ROCNarrowBandMeshField) = _check(ϕ.h.ptr, ccall((:lsm_band_halo, libhiplsm), Cint,
    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Int64, Ptr{Cvoid}),
    ϕ.h.ptr, pointer(ϕ.buf), pointer(ϕ.mask), pointer(ϕ.halo), pointer(ϕ.tiles), BAND_MC, pointer(ϕ.hlist), length(ϕ.hlist) ÷ 2, pointer(ϕ.hcount)), "lsm_band_halo")

# update_band!(ϕ) (src/timestepping.jl:115, src/meshfield.jl:555-588)
function LSM.update_band!(ϕ::ROCNarrowBandMeshField; from_dense = false)
    _check(ϕ.h.ptr, ccall((:lsm_band_update, libhiplsm), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Int64, Ptr{Cvoid}),
        ϕ.h.ptr, pointer(ϕ.buf), pointer(ϕ.mask), from_dense, ϕ.nlayers, pointer(ϕ.scratch[1]), pointer(ϕ.scratch[2]),
        pointer(ϕ.halo), pointer(ϕ.tiles), BAND_MC, pointer(ϕ.hlist), length(ϕ.hlist) ÷ 2, pointer(ϕ.hcount)), "lsm_band_update")
    want = _band_status(ϕ)
    if ϕ.h.world > 1
        # slab: band set and new values are right only away from the cut faces — the overlap planes come from their owners
        # (mask as whole planes, then only the band nodes' values), and tiles / lists / halo are re-derived from the full mask
        _check(ϕ.h.ptr, ccall((:lsm_band_overlap_mask, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ϕ.h.ptr, pointer(ϕ.mask)), "lsm_band_overlap_mask")
        _check(ϕ.h.ptr, ccall((:lsm_band_overlap_values, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ϕ.h.ptr, pointer(ϕ.buf)), "lsm_band_overlap_values")
        _check(ϕ.h.ptr, ccall((:lsm_band_retile, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint), ϕ.h.ptr, pointer(ϕ.mask), pointer(ϕ.tiles), BAND_MC), "lsm_band_retile")
        _band_status(ϕ)
        _band_halo(ϕ)
        want = _band_status(ϕ)
    end
    while want > length(ϕ.hlist) ÷ 2               # list too short: grow it, derive the halo again
        ϕ.hlist = ROCVector{Int64}(undef, 4 * want)
        _band_halo(ϕ)
        want = _band_status(ϕ)
    end
    return ϕ
end

function LSM.compute_cfl(terms, ϕ::ROCNarrowBandMeshField, t)
    ts = [_term(term, ϕ, t) for term in terms]
    dt = Ref{Float64}(0.0)
    _check(ϕ.h.ptr, ccall((:lsm_compute_cfl_band, libhiplsm), Cint,
        (Ptr{Cvoid}, Ptr{LsmTerm}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Float64, Ref{Float64}),
        ϕ.h.ptr, ts, length(ts), pointer(ϕ.buf), pointer(ϕ.mask), pointer(ϕ.tiles), BAND_MC, t, dt), "lsm_compute_cfl_band")
    ϕ.h.world > 1 && _check(ϕ.h.ptr, ccall((:lsm_allreduce_dt, libhiplsm), Cint, (Ptr{Cvoid}, Ref{Float64}), ϕ.h.ptr, dt), "lsm_allreduce_dt")
    Δt = dt[]
    Δt > 0 || throw(ArgumentError("invalid time-step based on CFL condition: Δt = $Δt (check for NaN/Inf in velocity or speed)"))
    return Δt
end

# buffers: value arrays only — their off-band entries are scratch, the band set is ϕ's (src/timestepping.jl:126,141,168 copy ϕ)
_value_buffer(ϕ::ROCNarrowBandMeshField{N, T, B, S}) where {N, T, B, S} = ROCMeshField{N, T, B, S}(similar(ϕ.buf), ϕ.mesh, ϕ.bcs, ϕ.h, ϕ.samples)
LSM._alloc_buffers(::LSM.ForwardEuler, ϕ::ROCNarrowBandMeshField) = (_value_buffer(ϕ),)
LSM._alloc_buffers(::Union{LSM.RK2, LSM.RK3}, ϕ::ROCNarrowBandMeshField) = (_value_buffer(ϕ), _value_buffer(ϕ))

# _advance! on a band (src/timestepping.jl:128-202 over active_nodeindices): lsm_advance_band_* prepares every stage input
# (band halo by affine extrapolation, src/meshfield.jl:481-511, then the boundary ghosts), runs the band-restricted stages and,
# on a slab, refreshes the overlap planes of every stage result
function LSM._advance!(::LSM.ForwardEuler, ϕ::ROCNarrowBandMeshField, (dst,), terms, tc, Δt)
    ts = [_term(term, ϕ, tc) for term in terms]
    hook, band = _hook(terms, (ϕ,)), _band(ϕ)
    GC.@preserve hook _check(ϕ.h.ptr, ccall((:lsm_advance_band_fe, libhiplsm), Cint,
        (Ptr{Cvoid}, Ptr{LsmTerm}, Cint, Ref{LsmBand}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Ptr{Cvoid}, Ptr{Cvoid}),
        ϕ.h.ptr, ts, length(ts), band, pointer(ϕ.buf), pointer(dst.buf), tc, Δt, hook, C_NULL), "lsm_advance_band_fe")
    return ϕ
end
function LSM._advance!(::LSM.RK2, ϕ::ROCNarrowBandMeshField, (pred, corr), terms, tc, Δt)
    ts = [_term(term, ϕ, tc) for term in terms]
    hook, band = _hook(terms, (ϕ, pred)), _band(ϕ)
    GC.@preserve hook _check(ϕ.h.ptr, ccall((:lsm_advance_band_rk2, libhiplsm), Cint,
        (Ptr{Cvoid}, Ptr{LsmTerm}, Cint, Ref{LsmBand}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Ptr{Cvoid}, Ptr{Cvoid}),
        ϕ.h.ptr, ts, length(ts), band, pointer(ϕ.buf), pointer(pred.buf), pointer(corr.buf), tc, Δt, hook, C_NULL), "lsm_advance_band_rk2")
    return ϕ
end
function LSM._advance!(::LSM.RK3, ϕ::ROCNarrowBandMeshField, (buf1, buf2), terms, tc, Δt)
    ts = [_term(term, ϕ, tc) for term in terms]
    hook, band = _hook(terms, (ϕ, buf1, buf2)), _band(ϕ)
    GC.@preserve hook _check(ϕ.h.ptr, ccall((:lsm_advance_band_rk3, libhiplsm), Cint,
        (Ptr{Cvoid}, Ptr{LsmTerm}, Cint, Ref{LsmBand}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Ptr{Cvoid}, Ptr{Cvoid}),
        ϕ.h.ptr, ts, length(ts), band, pointer(ϕ.buf), pointer(buf1.buf), pointer(buf2.buf), tc, Δt, hook, C_NULL), "lsm_advance_band_rk3")
    return ϕ
end

# one stage by hand (a driver of its own): the stage input made readable by stencils, then the band-restricted stage
function _band_stage!(ϕ::ROCNarrowBandMeshField, ts, psi, phin, out, out2, mode, cdt, cdt2, t)
    _check(ϕ.h.ptr, ccall((:lsm_band_prepare, libhiplsm), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Cint),
        ϕ.h.ptr, psi, pointer(ϕ.mask), pointer(ϕ.hlist), length(ϕ.hlist) ÷ 2, pointer(ϕ.hcount), pointer(ϕ.tiles), BAND_MC), "lsm_band_prepare")
    _check(ϕ.h.ptr, ccall((:lsm_stage_band, libhiplsm), Cint,
        (Ptr{Cvoid}, Ptr{LsmTerm}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Float64, Float64, Float64, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}),
        ϕ.h.ptr, ts, length(ts), psi, phin, out, out2, mode, cdt, cdt2, t, pointer(ϕ.mask), pointer(ϕ.tiles), BAND_MC, C_NULL), "lsm_stage_band")
end

# usage (drop-in):
#   ϕ  = MeshField(x -> norm(x) - 0.5, grid; bc = NeumannBC())
#   eq = LevelSetEquation(; terms = (AdvectionTerm(RigidRotation(1.0, (0.0, 0.0))),), ic = ROCMeshField(ϕ), integrator = RK3())
#   integrate!(eq, 1.0)              # _integrate! (src/timestepping.jl:101-122) runs unchanged on the host
#   ϕ_final = MeshField(current_state(eq))

end # module
