# TortoiseHIP.jl — thin `ccall` shim over libtortoise_hip.so (include/tortoise_hip.h).
#
# Drop-in for the ONE hot call of the reference: `TrajectoryOptimization.solve!(sat, solver)`
# (src/TortoiseSat.jl:199) and the loop body `solve(solver,U)` of src/monte_carlo.jl:196, for a batch of slews.
# The host keeps igrf / kep_ECI / magnetic_simulation / eigen_axis_slew / Bryson weights in Julia exactly as the
# reference scripts do and hands plain column-major arrays across the ABI (Julia's native layout — no copies).
#
# NOTE: no Julia toolchain exists in the authoring image; this file is syntax-reviewed only (DESIGN.md §1).
module TortoiseHIP

const LIB = joinpath(@__DIR__, "..", "tortoisesat.jl_amd", "csrc", "libtortoise_hip.so")

# struct tsat_options — field order and types exactly as include/tortoise_hip.h
Base.@kwdef mutable struct Options
    n_knots::Int32 = 0
    n_tab::Int32 = 0
    integrator::Int32 = 3            # rk3, src/TortoiseSat.jl:146
    precision::Int32 = 64
    max_outer::Int32 = 20            # opts_al.iterations, src/TortoiseSat.jl:196
    max_inner::Int32 = 50            # opts_al.opts_uncon.iterations, src/TortoiseSat.jl:195
    max_linesearch::Int32 = 20
    dj_counter_limit::Int32 = 10     # solver.opts.dJ_counter_limit, src/monte_carlo.jl:191
    cost_tol::Float64 = 1e-4
    grad_tol::Float64 = 1e-5
    constraint_tol::Float64 = 1e-3
    penalty_init::Float64 = 1.0
    penalty_scale::Float64 = 10.0
    penalty_max::Float64 = 1e8
    dual_max::Float64 = 1e8
    reg_init::Float64 = 0.0
    reg_scale::Float64 = 1.6
    reg_min::Float64 = 1e-8
    reg_max::Float64 = 1e8
    reg_fp::Float64 = 10.0
    ls_lower::Float64 = 1e-8
    ls_upper::Float64 = 10.0
    max_state::Float64 = 1e8
    u_scale::Float64 = 1e-2          # src/DerivFunction.jl:37
    terminal_mask::Int32 = 0x7f
    error_state::Int32 = 0
end

# struct tsat_stats (64 bytes)
struct Stats
    status::Int32; outer_iters::Int32; inner_iters::Int32; ls_trials::Int32
    n_backward::Int32; n_forward::Int32; bp_restarts::Int32; fp_fails::Int32
    cost::Float64; cost_al::Float64; c_max::Float64; grad::Float64
end

mutable struct HIPSolver          # plays the role of AugmentedLagrangianSolver (src/TortoiseSat.jl:197)
    handle::Ptr{Cvoid}
    opts::Options
    function HIPSolver(opts::Options = Options(); device::Integer = 0)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:tsat_create, LIB), Cint, (Ptr{Ptr{Cvoid}}, Cint), h, device)
        rc == 0 || error("tsat_create failed ($rc): a gfx950 GPU is required; there is no CPU fallback")
        s = new(h[], opts)
        finalizer(x -> ccall((:tsat_destroy, LIB), Cint, (Ptr{Cvoid},), x.handle), s)
        return s
    end
end

"""
BatchProblem: T independent slews, arrays in the reference's own shapes.
  x0, xf :: 7×T   (ω; q scalar-first — the 8th time state of src/TortoiseSat.jl:124 is dropped)
  B_ECI  :: 3×n_tab×n_btab   (transpose of the reference's 2N×3 table, src/TortoiseSat.jl:89)
  btab_idx :: T (0-based table index)     tau0, dtau, dt :: T
  J :: 3×3×T    Q, Qf :: 7×T (diagonals, src/TortoiseSat.jl:157-167)    R :: 3×T
  u_min, u_max :: 3×T (BoundConstraint, :178)    U0 :: 3×(N-1)×T (initial_controls!, :191)
"""
Base.@kwdef mutable struct BatchProblem
    N::Int
    x0::Matrix{Float64}; xf::Matrix{Float64}
    B_ECI::Array{Float64,3}; btab_idx::Vector{Int32}
    tau0::Vector{Float64}; dtau::Vector{Float64}; dt::Vector{Float64}
    J::Array{Float64,3}
    Q::Matrix{Float64}; Qf::Matrix{Float64}; R::Matrix{Float64}
    u_min::Matrix{Float64}; u_max::Matrix{Float64}
    U0::Array{Float64,3}
    X::Array{Float64,3} = zeros(0, 0, 0)     # 7×N×T after solve!
    U::Array{Float64,3} = zeros(0, 0, 0)     # 3×(N-1)×T
    K::Array{Float64,4} = zeros(0, 0, 0, 0)  # 3×7×(N-1)×T
    stats::Vector{Stats} = Stats[]
end

"solve!(prob, solver) — mutates prob.X, prob.U, prob.K, prob.stats like the reference's solve! mutates sat.X/sat.U"
function solve!(p::BatchProblem, s::HIPSolver)
    T = size(p.x0, 2); N = p.N
    o = s.opts; o.n_knots = N; o.n_tab = size(p.B_ECI, 2)
    p.X = zeros(7, N, T); p.U = zeros(3, N - 1, T); p.K = zeros(3, 7, N - 1, T)
    p.stats = Vector{Stats}(undef, T)
    rc = ccall((:tsat_solve_batch, LIB), Cint,
        (Ptr{Cvoid}, Ref{Options}, Int64, Int64,
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Stats}),
        s.handle, o, T, size(p.B_ECI, 3),
        p.x0, p.xf, p.B_ECI, p.btab_idx, p.tau0, p.dtau, p.dt, p.J,
        p.Q, p.Qf, p.R, p.u_min, p.u_max, p.U0,
        p.X, p.U, p.K, p.stats)
    rc == 0 || error("tsat_solve_batch failed ($rc): " *
                     unsafe_string(ccall((:tsat_last_error, LIB), Cstring, (Ptr{Cvoid},), s.handle)))
    return p
end

end # module
