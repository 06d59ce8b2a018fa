# ROCMeshField.jl — the reference-side binding a maintainer would add to LevelSetMethods.jl
# (e.g. as ext/ROCMExt.jl with AMDGPU as a weak dependency).  Host code stays Julia; AMDGPU.jl only
# owns the device arrays; every kernel is reached through `ccall` into libhiplsm.so (include/lsm.h).
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: there is no Julia runtime in the build container
# (SURVEY.md, "Container facts").  The Python host layer (levelsetmethods.jl_amd/api.py) drives the
# SAME C ABI call for call and is what the GPU tests exercise; this file mirrors it one to one.
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

lsmgrid(g::LSM.CartesianGrid{N}) where {N} =
    LsmGrid(N, 0, _pad3(Int64.(g.n), Int64(1)), _pad3(Float64.(g.lc), 0.0), _pad3(Float64.(g.hc), 1.0))

lsmbc(::LSM.PeriodicBC) = LsmBc(0, 0)
lsmbc(::LSM.ExtrapolationBC{P}) where {P} = LsmBc(1, P)
lsmbc(::LSM.SymmetryBC) = LsmBc(2, 0)

function _check(h, code, what)
    code == 0 && return
    msg = unsafe_string(ccall((:lsm_last_error, libhiplsm), Cstring, (Ptr{Cvoid},), h))
    error("$what failed ($code): $msg")
end

# ---- the device field --------------------------------------------------------------------------
"""
    ROCMeshField{N,T,B} <: AbstractMeshField{N,T,Float64}

Dense level-set field resident in HBM in libhiplsm's padded layout (ghost layers materialised).
"""
mutable struct ROCMeshField{N, T, B} <: LSM.AbstractMeshField{N, T, Float64}
    buf::ROCVector{Float64}
    mesh::LSM.CartesianGrid{N, T}
    bcs::B
    handle::Ptr{Cvoid}
    layout::LsmLayout
end

function _create(mesh::LSM.CartesianGrid{N}, bcs; strict = false) where {N}
    g = Ref(lsmgrid(mesh))
    bc = [d <= N ? lsmbc(bcs[d][s]) : LsmBc(1, 0) for s in 1:2, d in 1:3]   # C order bc[dim][side]
    h = Ref{Ptr{Cvoid}}(C_NULL)
    code = ccall((:lsm_create, libhiplsm), Cint,
        (Ref{LsmGrid}, Ptr{LsmBc}, Ptr{Cvoid}, Cint, Cint, Cint, Ref{Ptr{Cvoid}}),
        g, bc, C_NULL, 0, strict ? 1 : 0, AMDGPU.device_id(AMDGPU.device()) - 1, h)
    _check(C_NULL, code, "lsm_create")
    # kernels run on the stream AMDGPU.jl uses for its own copies
    _check(h[], ccall((:lsm_set_stream, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), h[], AMDGPU.stream().stream), "lsm_set_stream")
    lay = Ref{LsmLayout}()
    _check(h[], ccall((:lsm_layout, libhiplsm), Cint, (Ptr{Cvoid}, Ref{LsmLayout}), h[], lay), "lsm_layout")
    return h[], lay[]
end

function ROCMeshField(ϕ::LSM.MeshField{N, T}; strict = false) where {N, T}
    LSM._check_bc(ϕ)
    h, lay = _create(LSM.mesh(ϕ), LSM.boundary_conditions(ϕ); strict)
    buf = AMDGPU.zeros(Float64, lay.total)
    f = ROCMeshField{N, T, typeof(LSM.boundary_conditions(ϕ))}(buf, LSM.mesh(ϕ), LSM.boundary_conditions(ϕ), h, lay)
    finalizer(x -> ccall((:lsm_destroy, libhiplsm), Cvoid, (Ptr{Cvoid},), x.handle), f)
    _check(h, ccall((:lsm_upload, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}), h, pointer(buf), values(ϕ)), "lsm_upload")
    return f
end

# values(ϕ): host copy on demand (show, hooks, tests)
function Base.values(ϕ::ROCMeshField{N}) where {N}
    out = Array{Float64, N}(undef, size(LSM.mesh(ϕ)))
    _check(ϕ.handle, ccall((:lsm_download, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}), ϕ.handle, pointer(ϕ.buf), out), "lsm_download")
    return out
end
LSM.MeshField(ϕ::ROCMeshField) = LSM.MeshField(values(ϕ), LSM.mesh(ϕ), ϕ.bcs)

function Base.copy(ϕ::ROCMeshField{N, T, B}) where {N, T, B}
    h, lay = _create(ϕ.mesh, ϕ.bcs)
    f = ROCMeshField{N, T, B}(copy(ϕ.buf), ϕ.mesh, ϕ.bcs, h, lay)
    finalizer(x -> ccall((:lsm_destroy, libhiplsm), Cvoid, (Ptr{Cvoid},), x.handle), f)
    return f
end
Base.copy!(dst::ROCMeshField, src::ROCMeshField) = (copyto!(dst.buf, src.buf); dst)
LSM.update_band!(::ROCMeshField) = nothing
# scalar indexing: slow path for tests and hooks
Base.getindex(ϕ::ROCMeshField, I::CartesianIndex) = LSM.MeshField(ϕ)[I]

# ---- terms -> LsmTerm ---------------------------------------------------------------------------
# Julia closures cannot run on the device: constants, catalogued analytic fields and device
# MeshFields are passed through; anything else is sampled on the host into a FIELD before each stage.
struct RigidRotation; ω::Float64; c::NTuple{2, Float64}; end     # u = ω·(-(x₂-c₂), x₁-c₁, 0)

_coeff(v::Number) = LsmCoeff(0, 0, 1.0, (Float64(v), 0.0, 0.0, 0.0), (C_NULL, C_NULL, C_NULL), (C_NULL, C_NULL, C_NULL))
_coeff(v::Union{SVector, Tuple}) = LsmCoeff(0, 0, 1.0, _pad4(Float64.(Tuple(v))), (C_NULL, C_NULL, C_NULL), (C_NULL, C_NULL, C_NULL))
_coeff(r::RigidRotation) = LsmCoeff(1, 0, 1.0, (r.ω, r.c[1], r.c[2], 0.0), (C_NULL, C_NULL, C_NULL), (C_NULL, C_NULL, C_NULL))
_coeff(fs::NTuple{K, ROCMeshField}) where {K} =
    LsmCoeff(3, 0, 1.0, (0.0, 0.0, 0.0, 0.0), _pad3(map(f -> Ptr{Cvoid}(pointer(f.buf)), fs), C_NULL), (C_NULL, C_NULL, C_NULL))
_pad4(t) = ntuple(i -> i <= length(t) ? t[i] : 0.0, 4)

_term(t::LSM.AdvectionTerm) = LsmTerm(0, LSM.scheme(t) isa LSM.WENO5 ? 1 : 0, _coeff(LSM.velocity(t)), C_NULL)
_term(t::LSM.NormalMotionTerm) = LsmTerm(1, 0, _coeff(LSM.speed(t)), C_NULL)
_term(t::LSM.CurvatureTerm) = LsmTerm(2, 0, _coeff(LSM.coefficient(t)), C_NULL)
_term(t::LSM.EikonalReinitializationTerm{Nothing}) = LsmTerm(3, 0, _coeff(0.0), C_NULL)
_term(t::LSM.EikonalReinitializationTerm{<:ROCMeshField}) = LsmTerm(3, 0, _coeff(0.0), Ptr{Cvoid}(pointer(t.S₀.buf)))

# EikonalReinitializationTerm(ϕ₀::ROCMeshField): S₀ = ϕ₀/√(ϕ₀²+Δx²) on the device (src/levelsetterms.jl:217-221)
function LSM.EikonalReinitializationTerm(ϕ₀::ROCMeshField)
    S₀ = copy(ϕ₀)
    _check(ϕ₀.handle, ccall((:lsm_eikonal_sign, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        ϕ₀.handle, pointer(ϕ₀.buf), pointer(S₀.buf), C_NULL), "lsm_eikonal_sign")
    return LSM.EikonalReinitializationTerm{typeof(S₀)}(S₀)
end

# ---- the three methods the integrator dispatches on ---------------------------------------------
# compute_cfl (src/levelsetterms.jl:22-28): the library returns the raw minimum, Julia throws.
function LSM.compute_cfl(terms, ϕ::ROCMeshField, t)
    ts = [_term(term) for term in terms]
    dt = Ref{Float64}(0.0)
    _check(ϕ.handle, ccall((:lsm_compute_cfl, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{LsmTerm}, Cint, Ptr{Cvoid}, Float64, Ref{Float64}),
        ϕ.handle, ts, length(ts), pointer(ϕ.buf), t, dt), "lsm_compute_cfl")
    Δt = dt[]
    Δt > 0 || throw(ArgumentError("invalid time-step based on CFL condition: Δt = $Δt (check for NaN/Inf in velocity or speed)"))
    return Δt
end

# stage hook: lets update_func(field, stage_field, stage_time) run between stage launches
# (src/timestepping.jl:131,149,158,174,185,196); NULL when every term has the default no-op hook.
function _hook(terms, fields)
    all(t -> !hasproperty(t, :update_func) || t.update_func === LSM.AdvectionTerm(0).update_func, terms) && return C_NULL
    cb = (user, stage, ptr, tstage) -> begin
        for term in terms; LSM.update_term!(term, fields[stage + 1], tstage); end
        Cint(0)
    end
    return @cfunction($cb, Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Float64))
end

LSM._alloc_buffers(::LSM.ForwardEuler, ϕ::ROCMeshField) = (copy(ϕ),)
LSM._alloc_buffers(::Union{LSM.RK2, LSM.RK3}, ϕ::ROCMeshField) = (copy(ϕ), copy(ϕ))

function LSM._advance!(::LSM.ForwardEuler, ϕ::ROCMeshField, (dst,), terms, tc, Δt)
    ts = [_term(term) for term in terms]
    _check(ϕ.handle, ccall((:lsm_advance_fe, libhiplsm), Cint,
        (Ptr{Cvoid}, Ptr{LsmTerm}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Ptr{Cvoid}, Ptr{Cvoid}),
        ϕ.handle, ts, length(ts), pointer(ϕ.buf), pointer(dst.buf), tc, Δt, _hook(terms, (ϕ,)), C_NULL), "lsm_advance_fe")
    return ϕ
end
function LSM._advance!(::LSM.RK2, ϕ::ROCMeshField, (pred, corr), terms, tc, Δt)
    ts = [_term(term) for term in terms]
    _check(ϕ.handle, ccall((:lsm_advance_rk2, libhiplsm), Cint,
        (Ptr{Cvoid}, Ptr{LsmTerm}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Ptr{Cvoid}, Ptr{Cvoid}),
        ϕ.handle, ts, length(ts), pointer(ϕ.buf), pointer(pred.buf), pointer(corr.buf), tc, Δt, _hook(terms, (ϕ, pred)), C_NULL), "lsm_advance_rk2")
    return ϕ
end
function LSM._advance!(::LSM.RK3, ϕ::ROCMeshField, (buf1, buf2), terms, tc, Δt)
    ts = [_term(term) for term in terms]
    _check(ϕ.handle, ccall((:lsm_advance_rk3, libhiplsm), Cint,
        (Ptr{Cvoid}, Ptr{LsmTerm}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Ptr{Cvoid}, Ptr{Cvoid}),
        ϕ.handle, ts, length(ts), pointer(ϕ.buf), pointer(buf1.buf), pointer(buf2.buf), tc, Δt, _hook(terms, (ϕ, buf1, buf2)), C_NULL), "lsm_advance_rk3")
    return ϕ
end

# show needs extrema (src/meshfield.jl:300-303)
function Base.extrema(ϕ::ROCMeshField)
    lo, hi = Ref{Float64}(), Ref{Float64}()
    _check(ϕ.handle, ccall((:lsm_extrema, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}, Ref{Float64}), ϕ.handle, pointer(ϕ.buf), lo, hi), "lsm_extrema")
    return lo[], hi[]
end

# ---- next rows (SURVEY.md §8f) ------------------------------------------------------------------

# volume / perimeter (src/levelsetops.jl:139-149,171-183)
function LSM.volume(ϕ::ROCMeshField)
    out = Ref{Float64}()
    _check(ϕ.handle, ccall((:lsm_volume, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}), ϕ.handle, pointer(ϕ.buf), out), "lsm_volume")
    return out[]
end
function LSM.perimeter(ϕ::ROCMeshField)
    out = Ref{Float64}()
    _check(ϕ.handle, ccall((:lsm_perimeter, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}), ϕ.handle, pointer(ϕ.buf), out), "lsm_perimeter")
    return out[]
end

# volume / perimeter of a band field (src/levelsetops.jl:34-116,150-166); `mask` is the band's byte mask, ϕ prepared
# (lsm_band_prepare) for the perimeter's centred gradient
function band_volume(ϕ::ROCMeshField, mask::ROCVector{UInt8})
    out = Ref{Float64}()
    _check(ϕ.handle, ccall((:lsm_band_volume, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}), ϕ.handle, pointer(ϕ.buf), pointer(mask), out), "lsm_band_volume")
    return out[]
end
function band_perimeter(ϕ::ROCMeshField, mask::ROCVector{UInt8})
    out = Ref{Float64}()
    _check(ϕ.handle, ccall((:lsm_band_perimeter, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}), ϕ.handle, pointer(ϕ.buf), pointer(mask), out), "lsm_band_perimeter")
    return out[]
end

# extend_along_normals!(F, ϕ; ...) (src/velocityextension.jl:20-116); frozen = nothing -> band rule
function LSM.extend_along_normals!(F::ROCMeshField, ϕ::ROCMeshField; nb_iters = 50, cfl = 0.45, frozen = nothing,
                                   interface_band = 1.5, min_norm = 1.0e-14)
    N = length(ϕ.mesh.n)
    work = [similar(ϕ.buf) for _ in 1:(N + 1)]
    w = [i <= length(work) ? pointer(work[i]) : C_NULL for i in 1:4]
    _check(ϕ.handle, ccall((:lsm_extend_along_normals, libhiplsm), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Float64, Float64, Float64),
        ϕ.handle, pointer(F.buf), pointer(ϕ.buf), frozen === nothing ? C_NULL : pointer(frozen.buf), w[1], w[2], w[3], w[4],
        nb_iters, cfl, interface_band, min_norm), "lsm_extend_along_normals")
    _check(ϕ.handle, ccall((:lsm_sync, libhiplsm), Cint, (Ptr{Cvoid},), ϕ.handle), "lsm_sync")   # work buffers die here
    return F
end

# curvature / gradient / normal of ϕ at every node (src/levelsetops.jl:197-226) into Float64 device arrays in ϕ's padded
# layout; `band`: only |ϕ[I]| <= band is evaluated (else `fill`) and `frozen` marks those nodes with 1.0 — the seed loop of
# the reference's speed update functions (test/test-velocityextension.jl:118-131) without a host pass.
function curvature_field!(out::ROCVector{Float64}, ϕ::ROCMeshField; scale = 1.0, band = -1.0, fill = 0.0, frozen = nothing)
    _check(ϕ.handle, ccall((:lsm_geometry, libhiplsm), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Float64, Float64, Float64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        ϕ.handle, 0, pointer(ϕ.buf), scale, band, fill, pointer(out), C_NULL, C_NULL, frozen === nothing ? C_NULL : pointer(frozen), C_NULL),
        "lsm_geometry")
    return out
end
function _vector_field!(what::Integer, outs::NTuple{N, ROCVector{Float64}}, ϕ::ROCMeshField; scale = 1.0) where {N}
    o = ntuple(d -> d <= N ? pointer(outs[d]) : C_NULL, 3)
    _check(ϕ.handle, ccall((:lsm_geometry, libhiplsm), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Float64, Float64, Float64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        ϕ.handle, what, pointer(ϕ.buf), scale, -1.0, 0.0, o[1], o[2], o[3], C_NULL, C_NULL), "lsm_geometry")
    return outs
end
# band fields: the same over the active nodes of a prepared band (`mask` = the band's byte mask)
function band_curvature_field!(out::ROCVector{Float64}, ϕ::ROCMeshField, mask::ROCVector{UInt8}; scale = 1.0, band = -1.0, fill = 0.0, frozen = nothing)
    _check(ϕ.handle, ccall((:lsm_band_geometry, libhiplsm), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Float64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        ϕ.handle, 0, pointer(ϕ.buf), pointer(mask), scale, band, fill, pointer(out), C_NULL, C_NULL, frozen === nothing ? C_NULL : pointer(frozen), C_NULL),
        "lsm_band_geometry")
    return out
end
gradient_field!(outs, ϕ::ROCMeshField; kw...) = _vector_field!(1, outs, ϕ; kw...)
normal_field!(outs, ϕ::ROCMeshField; kw...) = _vector_field!(2, outs, ϕ; kw...)

# InterpolatedField(ϕ, order) evaluated at points (src/interpolation.jl:117-151,228-260): `pts` is an ndim x npts device matrix
# (column = point); returns values, and fills `grad` (ndim x npts) / `hess` (ndim x ndim x npts) when given
function interpolate!(val::ROCVector{Float64}, ϕ::ROCMeshField, order::Integer, pts::ROCMatrix{Float64}; grad = nothing, hess = nothing)
    _check(ϕ.handle, ccall((:lsm_interpolate, libhiplsm), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        ϕ.handle, pointer(ϕ.buf), order, size(pts, 2), pointer(pts), pointer(val), grad === nothing ? C_NULL : pointer(grad),
        hess === nothing ? C_NULL : pointer(hess), C_NULL), "lsm_interpolate")
    return val
end

# NewtonSDF(ϕ; ...) (src/sdf.jl:57-127) as a device object: build once, query points, read the samples back
mutable struct ROCNewtonSDF
    ptr::Ptr{Cvoid}
    handle::Ptr{Cvoid}
    nsamples::Int64
end
function ROCNewtonSDF(ϕ::ROCMeshField; order = 3, upsample = 2, maxiters = 10, xtol = nothing, ftol = nothing, mask = nothing)
    xt, ft = something(xtol, sqrt(eps(Float64))), something(ftol, sqrt(eps(Float64)))
    out, ns = Ref{Ptr{Cvoid}}(), Ref{Int64}()
    _check(ϕ.handle, ccall((:lsm_sdf_create, libhiplsm), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Float64, Float64, Ref{Ptr{Cvoid}}, Ref{Int64}),
        ϕ.handle, pointer(ϕ.buf), mask === nothing ? C_NULL : pointer(mask), order, upsample, maxiters, xt, ft, out, ns), "lsm_sdf_create")
    sdf = ROCNewtonSDF(out[], ϕ.handle, ns[])
    finalizer(s -> ccall((:lsm_sdf_destroy, libhiplsm), Cvoid, (Ptr{Cvoid},), s.ptr), sdf)
    return sdf
end
# signed distances at the columns of `pts` (ndim x npts device matrix); `cp` optionally receives the closest points
function (sdf::ROCNewtonSDF)(dist::ROCVector{Float64}, pts::ROCMatrix{Float64}; cp = nothing)
    nfail = Ref{Int64}()
    _check(sdf.handle, ccall((:lsm_sdf_eval, libhiplsm), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Int64}),
        sdf.ptr, size(pts, 2), pointer(pts), pointer(dist), cp === nothing ? C_NULL : pointer(cp), nfail), "lsm_sdf_eval")
    return dist
end

# reinitialize!(ϕ; ...) (src/reinitializer.jl:12-42)
function LSM.reinitialize!(ϕ::ROCMeshField; order = 3, upsample = 2, maxiters = 20, xtol = nothing, ftol = nothing)
    xt, ft = something(xtol, sqrt(eps(Float64))), something(ftol, sqrt(eps(Float64)))
    _check(ϕ.handle, ccall((:lsm_fill_ghosts, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}), ϕ.handle, pointer(ϕ.buf), 7, C_NULL), "lsm_fill_ghosts")
    work = similar(ϕ.buf)
    nc, nfail, nfar = Ref{Int64}(), Ref{Int64}(), Ref{Int64}()
    _check(ϕ.handle, ccall((:lsm_reinitialize, libhiplsm), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Float64, Float64, Ref{Int64}, Ref{Int64}, Ref{Int64}),
        ϕ.handle, pointer(ϕ.buf), C_NULL, pointer(work), order, upsample, maxiters, xt, ft, nc, nfail, nfar), "lsm_reinitialize")
    nfail[] > 0 && @warn "reinitialize!: closest-point solver did not converge for $(nfail[]) points"
    return ϕ
end

# NarrowBandMeshField on the device (src/meshfield.jl:314-588): dense padded values + byte masks.
const BAND_MC = 8
mutable struct ROCNarrowBandMeshField{N,T,B} <: LSM.AbstractMeshField{N,T,Float64}
    buf::ROCVector{Float64}; mesh::CartesianGrid{N,T}; bcs::B; handle::Ptr{Cvoid}; layout::LsmLayout
    nlayers::Int
    mask::ROCVector{UInt8}; halo::ROCVector{UInt8}; tiles::ROCVector{UInt8}
    scratch::NTuple{2,ROCVector{UInt8}}
    hlist::ROCVector{Int64}; hcount::ROCVector{UInt32}     # (halo node -> nearest band node) entries, 2 Int64 each
end

# update_band!(ϕ) (src/timestepping.jl:115)
function LSM.update_band!(ϕ::ROCNarrowBandMeshField; from_dense = false)
    while true
        _check(ϕ.handle, ccall((:lsm_band_update, libhiplsm), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Int64, Ptr{Cvoid}),
            ϕ.handle, pointer(ϕ.buf), pointer(ϕ.mask), from_dense, ϕ.nlayers, pointer(ϕ.scratch[1]), pointer(ϕ.scratch[2]),
            pointer(ϕ.halo), pointer(ϕ.tiles), BAND_MC, pointer(ϕ.hlist), length(ϕ.hlist) ÷ 2, pointer(ϕ.hcount)), "lsm_band_update")
        want, missed = Ref{Int64}(), Ref{Cint}()
        _check(ϕ.handle, ccall((:lsm_band_status, libhiplsm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Int64}, Ref{Cint}),
            ϕ.handle, pointer(ϕ.hcount), want, missed), "lsm_band_status")
        missed[] != 0 && throw(ArgumentError("index is more than $(LSM._BAND_SEARCH_RADIUS) nodes from the band"))   # src/meshfield.jl:499-500
        want[] <= length(ϕ.hlist) ÷ 2 && return ϕ
        ϕ.hlist = ROCVector{Int64}(undef, 4 * want[])       # list too short: grow it, derive the halo again
        from_dense = false
    end
end

# one stage input made readable by stencils, then the band-restricted stage (what _advance! loops over)
function _band_stage!(ϕ::ROCNarrowBandMeshField, ts, psi, phin, out, out2, mode, cdt, cdt2, t)
    _check(ϕ.handle, ccall((:lsm_band_prepare, libhiplsm), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Cint),
        ϕ.handle, psi, pointer(ϕ.mask), pointer(ϕ.hlist), length(ϕ.hlist) ÷ 2, pointer(ϕ.hcount), pointer(ϕ.tiles), BAND_MC), "lsm_band_prepare")
    _check(ϕ.handle, ccall((:lsm_stage_band, libhiplsm), Cint,
        (Ptr{Cvoid}, Ptr{LsmTerm}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Float64, Float64, Float64, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}),
        ϕ.handle, ts, length(ts), psi, phin, out, out2, mode, cdt, cdt2, t, pointer(ϕ.mask), pointer(ϕ.tiles), BAND_MC, C_NULL), "lsm_stage_band")
end

function LSM.compute_cfl(terms, ϕ::ROCNarrowBandMeshField, t)
    ts = [_term(term) for term in terms]
    dt = Ref{Float64}(0.0)
    _check(ϕ.handle, ccall((:lsm_compute_cfl_band, libhiplsm), Cint,
        (Ptr{Cvoid}, Ptr{LsmTerm}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Float64, Ref{Float64}),
        ϕ.handle, ts, length(ts), pointer(ϕ.buf), pointer(ϕ.mask), pointer(ϕ.tiles), BAND_MC, t, dt), "lsm_compute_cfl_band")
    Δt = dt[]
    Δt > 0 || throw(ArgumentError("invalid time-step based on CFL condition: Δt = $Δt (check for NaN/Inf in velocity or speed)"))
    return Δt
end

# usage (drop-in):
#   ϕ  = MeshField(x -> norm(x) - 0.5, grid; bc = NeumannBC())
#   eq = LevelSetEquation(; terms = (AdvectionTerm(RigidRotation(1.0, (0.0, 0.0))),), ic = ROCMeshField(ϕ), integrator = RK3())
#   integrate!(eq, 1.0)              # _integrate! (src/timestepping.jl:101-122) runs unchanged on the host
#   ϕ_final = MeshField(current_state(eq))

end # module
