# make_reference_goldens.jl — REFERENCE-generated fixtures for the hot path.
#
# Run by a maintainer who has Julia (>= 1.10) and the real package; it cannot run in the build container of this
# repository (no Julia there), which is why tests/golden/*.npz are oracle-generated and the oracle is "parity
# unpinned" (DESIGN.md §4).  Files written by this script pin it:
#
#     julia --project=/path/to/LevelSetMethods.jl tools/make_reference_goldens.jl [tests/golden/reference]
#
# (the environment needs LevelSetMethods v0.2.0 and StaticArrays).  For each of the three golden workloads of
# tests/golden/make_golden.py it writes a directory of .npy arrays (NumPy format 1.0, written by hand below: no
# NPZ.jl dependency) that tests/test_reference_goldens.py picks up and checks the oracle and the HIP path against:
#
#   headline_3d/  n, phi0, tables, dts, phi3        vortex-deformation WENO5 advection + EikonalReinitializationTerm(),
#                                                   RK3, NeumannBC, 20x18x16, 3 steps of 0.5·compute_cfl   (BASELINE config 4)
#   zalesak_2d/   phi0, dts_adv, adv4, dts_re, reinit2   slotted disk, rigid rotation WENO5 + RK3, NeumannBC, 48², 4 steps; then
#                                                   2 RK3 steps of EikonalReinitializationTerm(ϕ) (frozen sign) (config 2)
#   mcf_3d/       phi0, dts, phi3                   sphere, NormalMotionTerm(0.1) + CurvatureTerm(-0.1), RK3,
#                                                   ExtrapolationBC(2), 18³, 3 steps                       (config 3)
#
# Inputs are WRITTEN, not assumed: phi0 (and the per-axis velocity factors of the vortex field) as this Julia session
# computed them, so that the other side starts from bit-identical data whatever libm it has.  Each step is the body of
# `_integrate!` (src/timestepping.jl:104-116) with Δt = 0.5·compute_cfl — `update_term!`s, `compute_cfl`, `_advance!` —
# called directly, so that no `tf - tc` rounding enters Δt.
using LevelSetMethods, StaticArrays, LinearAlgebra
const LSM = LevelSetMethods

outdir = length(ARGS) >= 1 ? ARGS[1] : joinpath(@__DIR__, "..", "tests", "golden", "reference")

# ---- NumPy .npy (format 1.0), column-major data flagged fortran_order
function write_npy(path::AbstractString, A::AbstractArray{T}) where {T <: Union{Float64, Int64}}
    descr = T === Float64 ? "<f8" : "<i8"
    shape = ndims(A) == 1 ? "($(length(A)),)" : "(" * join(size(A), ", ") * ")"
    hdr = "{'descr': '$descr', 'fortran_order': True, 'shape': $shape, }"
    pad = 64 - mod(10 + length(hdr) + 1, 64)
    hdr = hdr * " "^(pad == 64 ? 0 : pad) * "\n"
    open(path, "w") do io
        write(io, UInt8[0x93], "NUMPY", UInt8[0x01, 0x00], htol(UInt16(length(hdr))), hdr)
        write(io, htol.(vec(collect(A))))
    end
    return path
end

function dump(case::AbstractString, arrays::Pair...)
    dir = joinpath(outdir, case)
    mkpath(dir)
    for (name, A) in arrays
        write_npy(joinpath(dir, "$name.npy"), A)
    end
    println(case, ": ", join(first.(arrays), ", "))
end

# k steps of Δt = 0.5·compute_cfl; returns the Δt's (the state of `eq` is advanced in place)
function steps!(eq, k)
    ϕ, terms, integ = LSM.current_state(eq), eq.terms, LSM.time_integrator(eq)
    buffers = LSM._alloc_buffers(integ, ϕ)
    t, dts = LSM.current_time(eq), Float64[]
    for _ in 1:k
        for term in terms
            LSM.update_term!(term, ϕ, t)
        end
        Δt = 0.5 * LSM.compute_cfl(terms, ϕ, t)
        LSM._advance!(integ, ϕ, buffers, terms, t, Δt)
        t += Δt
        push!(dts, Δt)
    end
    return dts
end

vals(ϕ) = copy(values(ϕ))

# ---- headline (BASELINE config 4 in miniature)
let n = (20, 18, 16)
    grid = CartesianGrid((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), n)
    ϕ = MeshField(x -> sqrt((x[1] - 0.35)^2 + (x[2] - 0.35)^2 + (x[3] - 0.35)^2) - 0.15, grid)
    𝐮 = (x, t) -> SVector(
        2 * sin(π * x[1])^2 * sin(2π * x[2]) * sin(2π * x[3]) * cos(π * t / 3),
        -sin(2π * x[1]) * sin(π * x[2])^2 * sin(2π * x[3]) * cos(π * t / 3),
        -sin(2π * x[1]) * sin(2π * x[2]) * sin(π * x[3])^2 * cos(π * t / 3),
    )
    # the per-axis factors exactly as the closure evaluates them: component c, axis d -> tables[c][d]
    ax(d) = [LSM.getnode(grid, CartesianIndex(ntuple(j -> j == d ? i : 1, 3)))[d] for i in 1:n[d]]
    s2(a) = sin(π * a)^2
    s(a) = sin(2π * a)
    x, y, z = ax(1), ax(2), ax(3)
    tables = vcat(2 .* s2.(x), s.(y), s.(z), -s.(x), s2.(y), s.(z), -s.(x), s.(y), s2.(z))
    eq = LevelSetEquation(; terms = (AdvectionTerm(𝐮, WENO5()), EikonalReinitializationTerm()), ic = ϕ, bc = NeumannBC(), integrator = RK3())
    ϕ0 = vals(LSM.current_state(eq))
    dts = steps!(eq, 3)
    dump("headline_3d", "n" => Int64[n...], "phi0" => ϕ0, "tables" => tables, "dts" => dts, "phi3" => vals(LSM.current_state(eq)))
end

# ---- Zalesak (config 2 in miniature; docs/src/example-zalesak.md:21-26)
let n = (48, 48)
    grid = CartesianGrid((-1.5, -1.5), (1.5, 1.5), n)
    disk = MeshField(x -> hypot(x[1] + 0.75, x[2]) - 0.5, grid)
    rec = MeshField(x -> max(abs(x[1] + 0.75) - 0.1, abs(x[2] + 0.25) - 0.5), grid)
    ϕ = MeshField(max.(values(disk), .-values(rec)), grid)
    eq = LevelSetEquation(; terms = (AdvectionTerm((x, t) -> SVector(-x[2], x[1]), WENO5()),), ic = ϕ, bc = NeumannBC(), integrator = RK3())
    ϕ0 = vals(LSM.current_state(eq))
    dts_adv = steps!(eq, 4)
    adv4 = vals(LSM.current_state(eq))
    # periodic PDE reinitialisation: a second equation with the sign frozen from the current state
    cur = MeshField(copy(adv4), grid)
    re = LevelSetEquation(; terms = (EikonalReinitializationTerm(cur),), ic = cur, bc = NeumannBC(), integrator = RK3())
    dts_re = steps!(re, 2)
    dump("zalesak_2d", "phi0" => ϕ0, "dts_adv" => dts_adv, "adv4" => adv4, "dts_re" => dts_re, "reinit2" => vals(LSM.current_state(re)))
end

# ---- mean-curvature flow (config 3 in miniature)
let n = (18, 18, 18)
    grid = CartesianGrid((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0), n)
    ϕ = MeshField(x -> sqrt(x[1] * x[1] + x[2] * x[2] + x[3] * x[3]) - 0.5, grid)
    eq = LevelSetEquation(; terms = (NormalMotionTerm((x, t) -> 0.1), CurvatureTerm((x, t) -> -0.1)), ic = ϕ, bc = ExtrapolationBC(2), integrator = RK3())
    ϕ0 = vals(LSM.current_state(eq))
    dts = steps!(eq, 3)
    dump("mcf_3d", "phi0" => ϕ0, "dts" => dts, "phi3" => vals(LSM.current_state(eq)))
end

println("written to ", abspath(outdir), " — commit the directories; tests/test_reference_goldens.py uses them when present")
