# TortoiseHIP.jl — thin `ccall` shim over libtortoise_hip.so (include/tortoise_hip.h).
#
# Drop-in for the hot call of the reference: `TrajectoryOptimization.solve!(sat, solver)`
# (src/TortoiseSat.jl:199) and the loop body `solve(solver,U)` of src/monte_carlo.jl:196, for a batch of slews —
# plus the batched forms of the calls on either side of it: `magnetic_simulation` (src/magnetic_toolbox.jl:33-106),
# `magnetic_gramian` + `condition_based_time` (:1-31), `attitude_simulation` with the slew-time statistic
# (src/attitude_controller.jl:1-48, src/monte_carlo.jl:242-262) and the result files (src/monte_carlo.jl:334-343).
# eigen_axis_slew / Bryson weights stay in Julia exactly as the reference scripts have them; plain column-major arrays
# cross the ABI (Julia's native layout — no copies).
#
# NOTE: no Julia toolchain exists in the authoring image, so this file has never been executed. What IS checked, in the CPU test
# tier (tests/test_julia_shim.py): every `ccall` below names a function of include/tortoise_hip.h with the header's arity and
# layout-compatible argument / return types, every header function is bound here, and the structs mirror the C structs field
# for field (names, order, types, sizes).
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

"library / ABI version: major*100 + minor"
version() = ccall((:tsat_version, LIB), Cint, ())

"Options() as tsat_default_options fills them (the defaults above are the same numbers)"
function default_options()
    o = Options()
    ccall((:tsat_default_options, LIB), Cvoid, (Ref{Options},), o)
    return o
end

"set_kernel_variant!(s, v) — 0 automatic by batch size; 1 wide, 2 dense, 3 packed, 4 packed8, 5 packed8w, 6 packed16w, 7 packed4w (results do not depend on it)"
set_kernel_variant!(s::HIPSolver, v::Integer) =
    check(s, ccall((:tsat_set_kernel_variant, LIB), Cint, (Ptr{Cvoid}, Int32), s.handle, v), "tsat_set_kernel_variant")

"set_endgame!(s, n) — packed builds: park the last n live trajectories for a one-per-wavefront launch (-1 automatic, 0 never)"
set_endgame!(s::HIPSolver, n::Integer) =
    check(s, ccall((:tsat_set_endgame, LIB), Cint, (Ptr{Cvoid}, Int32), s.handle, n), "tsat_set_endgame")

"selected_build(s, o) — (build, endgame_at) the next run launches on the reserved batch: 1 wide, 2 dense, 3 packed, 4 packed8, 5 packed8w, 6 packed16w, 7 packed4w"
function selected_build(s::HIPSolver, o::Options)
    b = Ref{Int32}(0); e = Ref{Int32}(0)
    check(s, ccall((:tsat_selected_build, LIB), Cint, (Ptr{Cvoid}, Ref{Options}, Ref{Int32}, Ref{Int32}), s.handle, o, b, e), "tsat_selected_build")
    return Int(b[]), Int(e[])
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
    n_knots::Vector{Int32} = Int32[]         # optional per-slew horizons length(t_total[i]) (src/monte_carlo.jl:145); empty = all N
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
    if isempty(p.n_knots)
        rc = ccall((:tsat_solve_batch, LIB), Cint,
            (Ptr{Cvoid}, Ref{Options}, Int64, Int64,
             Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
             Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
             Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Stats}),
            s.handle, o, T, size(p.B_ECI, 3),
            p.x0, p.xf, p.B_ECI, p.btab_idx, p.tau0, p.dtau, p.dt, p.J,
            p.Q, p.Qf, p.R, p.u_min, p.u_max, p.U0,
            p.X, p.U, p.K, p.stats)
        check(s, rc, "tsat_solve_batch")
    else   # variable horizons: the resident-batch calls, with tsat_batch_knots between upload and run
        check(s, ccall((:tsat_batch_reserve, LIB), Cint, (Ptr{Cvoid}, Int64, Int32, Int32, Int64, Int32),
                       s.handle, T, N, o.n_tab, size(p.B_ECI, 3), o.max_linesearch), "tsat_batch_reserve")
        check(s, ccall((:tsat_batch_upload, LIB), Cint,
            (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
             Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
            s.handle, p.x0, p.xf, p.B_ECI, p.btab_idx, p.tau0, p.dtau, p.dt, p.J, p.Q, p.Qf, p.R, p.u_min, p.u_max, p.U0),
            "tsat_batch_upload")
        check(s, ccall((:tsat_batch_knots, LIB), Cint, (Ptr{Cvoid}, Ptr{Int32}), s.handle, p.n_knots), "tsat_batch_knots")
        check(s, ccall((:tsat_batch_run, LIB), Cint, (Ptr{Cvoid}, Ref{Options}, Ptr{Cfloat}), s.handle, o, C_NULL), "tsat_batch_run")
        check(s, ccall((:tsat_batch_download, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Stats}),
                       s.handle, p.X, p.U, p.K, p.stats), "tsat_batch_download")
    end
    return p
end

"bytes of HBM reserved for the resident batch / held by the grow-only workspaces of the stages around the solve"
batch_bytes(s::HIPSolver) = ccall((:tsat_batch_bytes, LIB), Int64, (Ptr{Cvoid},), s.handle)
workspace_bytes(s::HIPSolver) = ccall((:tsat_workspace_bytes, LIB), Int64, (Ptr{Cvoid},), s.handle)
"workspace_trim!(s; everything = false) — free the staging buffers of downloads / gathers (everything: all workspaces)"
workspace_trim!(s::HIPSolver; everything::Bool = false) =
    check(s, ccall((:tsat_workspace_trim, LIB), Cint, (Ptr{Cvoid}, Int32), s.handle, everything ? 1 : 0), "tsat_workspace_trim")

"""
trace!(s, rows) before solve!, then `trace(s, rows, T)` after it: per-iteration rows
[outer, inner, J_prev, J_new, alpha_index (-1 = none), rho, dV1, dV2] :: 8×rows×T (debugging parity).
"""
trace!(s::HIPSolver, rows::Integer) = check(s, ccall((:tsat_batch_trace, LIB), Cint, (Ptr{Cvoid}, Int32), s.handle, rows), "tsat_batch_trace")
function trace(s::HIPSolver, rows::Integer, T::Integer)
    tr = zeros(8, rows, T)
    check(s, ccall((:tsat_batch_trace_download, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), s.handle, tr), "tsat_batch_trace_download")
    return tr
end

"""
export_device!(s; X, U, K, stats) — unpack the resident results into DEVICE buffers the caller owns on the same GPU
(raw pointers, e.g. from AMDGPU.jl's `pointer(roc_array)`; C_NULL = not wanted): the hand-off to a collective of the host's own.
"""
export_device!(s::HIPSolver; X::Ptr{Cvoid} = C_NULL, U::Ptr{Cvoid} = C_NULL, K::Ptr{Cvoid} = C_NULL, stats::Ptr{Cvoid} = C_NULL) =
    check(s, ccall((:tsat_batch_export_device, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                   s.handle, X, U, K, stats), "tsat_batch_export_device")

check(s::HIPSolver, rc, what) = rc == 0 ? nothing :
    error("$what failed ($rc): " * unsafe_string(ccall((:tsat_last_error, LIB), Cstring, (Ptr{Cvoid},), s.handle)))

# ---------------------------------------------------------------------------------------------------------------
# the callers on either side of the solve
# ---------------------------------------------------------------------------------------------------------------
# struct tsat_btable_options
Base.@kwdef mutable struct BtableOptions
    n_half::Int32 = 5000             # N, src/TortoiseSat.jl:61
    reserved::Int32 = 0
    mjd::Float64 = 58155.0           # MJD_0, src/TortoiseSat.jl:44
    gm::Float64 = 3.986004418e5      # km^3/s^2, src/input_parameters.jl:26
    r_igrf_km::Float64 = 6771.0      # alt + R_E, src/magnetic_toolbox.jl:44,81
    date::Float64 = 2019.0           # src/magnetic_toolbox.jl:81
end

function default_btable_options()
    o = BtableOptions()
    ccall((:tsat_btable_default_options, LIB), Cvoid, (Ref{BtableOptions},), o)
    return o
end
"counter bumped by every tsat_btable_batch on the handle: identifies the field tables left resident on the device"
btable_generation(s::HIPSolver) = ccall((:tsat_btable_generation, LIB), Int64, (Ptr{Cvoid},), s.handle)

"""
bryson_eigen_axis_batch(n_knots, t0, dt, theta_f, axis, J; alpha, beta) — the loop body's script arithmetic for T slews between
the same two attitudes that differ only in their horizon (src/eigen_axis_slew.jl:1-38 + src/monte_carlo.jl:161-176), batched
on the host by the library. J :: 3×3 (symmetric: row- and column-major coincide). Returns Q 7×T, Qf 7×T, R 3×T diagonals.
"""
function bryson_eigen_axis_batch(n_knots::Vector{Int32}, t0::Float64, dt::Float64, theta_f::Float64, axis::Vector{Float64},
                                 J::Matrix{Float64}; alpha::Float64 = 0.1, beta::Float64 = 1.0e3)
    T = length(n_knots); Q = zeros(7, T); Qf = zeros(7, T); R = zeros(3, T)
    rc = ccall((:tsat_bryson_eigen_axis_batch, LIB), Cint,
        (Int64, Ptr{Int32}, Float64, Float64, Float64, Ptr{Float64}, Ptr{Float64}, Float64, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        T, n_knots, t0, dt, theta_f, axis, Matrix(transpose(J)), alpha, beta, Q, Qf, R)
    rc == 0 || error("tsat_bryson_eigen_axis_batch failed ($rc): " * (rc == -2 ? "a guess has no positive torque sample" : "bad arguments"))
    return Q, Qf, R
end

"""
magnetic_simulation(s, A, t0, tf, N) — batched `magnetic_simulation(A[i,:], t0, tf[i], N, mag_field, GM, MJD_0)[1]`
(src/magnetic_toolbox.jl:33-106). A :: 6×T Keplerian elements (the rows of the reference's `A`), t0, tf :: T.
Returns B_ECI :: 3×2N×T (Tesla; last row zero as in the reference) and pos :: 3×(2N+1)×T (km).
"""
function magnetic_simulation(s::HIPSolver, A::Matrix{Float64}, t0::Vector{Float64}, tf::Vector{Float64}, N::Integer;
                             opts::BtableOptions = BtableOptions())
    T = size(A, 2); opts.n_half = N
    B = zeros(3, 2N, T); pos = zeros(3, 2N + 1, T)
    check(s, ccall((:tsat_btable_batch, LIB), Cint,
        (Ptr{Cvoid}, Ref{BtableOptions}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        s.handle, opts, T, A, t0, tf, B, pos), "tsat_btable_batch")
    return B, pos
end

"""
condition_based_time(s, B, dt, cutoff) — batched `condition_based_time(magnetic_gramian(B_N, dt), cutoff)`
(src/magnetic_toolbox.jl:1-31; call sites src/TortoiseSat.jl:73-82, src/monte_carlo.jl:137-140).
B :: 3×n_rows×T. Returns the 1-based row index per orbit (0 = never below the cutoff) and the condition number there.
"""
function condition_based_time(s::HIPSolver, B::Array{Float64,3}, dt::Vector{Float64}, cutoff::Vector{Float64})
    T = size(B, 3); idx = zeros(Int32, T); cnd = zeros(T)
    check(s, ccall((:tsat_horizon_batch, LIB), Cint,
        (Ptr{Cvoid}, Int64, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}),
        s.handle, T, size(B, 2), B, dt, cutoff, idx, cnd), "tsat_horizon_batch")
    return idx, cnd
end

# struct tsat_tvlqr_options / tsat_tvlqr_stats
Base.@kwdef mutable struct TvlqrOptions
    n_knots::Int32 = 0
    n_tab::Int32 = 0
    linearize_dt_sq::Int32 = 1       # `dt = S[end]^2`, src/attitude_controller.jl:137
    min_steps::Int32 = 10            # src/monte_carlo.jl:251
    u_scale::Float64 = 1e-2          # src/gain_simulator.jl:42
    w_tol::Float64 = 0.05            # slew_limits, src/monte_carlo.jl:70-71
    angle_tol::Float64 = 0.08727
    noise_mode::Int32 = 0            # 1: the kernel draws the plant noise itself (Philox4x32-10 keyed by noise_seed)
    rate_as_written::Int32 = 0       # 1: `norm(sim_states[i][1:3,i])` as src/monte_carlo.jl:247 has it (rate of sample i = trial number)
    noise_seed::UInt64 = 0
    sigma_gyro::Float64 = (0.38 * pi / 180)^2    # src/simulator.jl:5
    sigma_att::Float64 = (pi / 180)^2            # src/simulator.jl:10
    field_amp::Float64 = 1e-10                   # src/simulator.jl:22
end
struct TvlqrStats
    slew_index::Int32; failed::Int32; slew_time::Float64; final_w_norm::Float64; final_angle::Float64
end

function default_tvlqr_options()
    o = TvlqrOptions()
    ccall((:tsat_tvlqr_default_options, LIB), Cvoid, (Ref{TvlqrOptions},), o)
    return o
end

"""
attitude_simulation(s, p, x0_lqr, Q_lqr, Qf_lqr, R_lqr; noise) — batched
`attitude_simulation(f!, f_gains!, :rk4, X, U, dt, x0_lqr, t0, tf, Q_lqr, R_lqr, Qf_lqr)` (src/attitude_controller.jl:1-48)
around the solved `p.X`, `p.U`, plus the slew-time / failure statistic of src/monte_carlo.jl:242-262.
x0_lqr :: 7×T; Q_lqr, Qf_lqr :: 6×T and R_lqr :: 3×T diagonals; noise :: 9×4×(N-1)×T draws of `simulator`
(gyro noise, attitude-noise rotation vector, field noise per RK4 stage; src/simulator.jl:5,10,22) or `nothing`.
Returns X_sim 7×N×T, U_sim 3×(N-1)×T, K 3×6×(N-1)×T, stats.
"""
function attitude_simulation(s::HIPSolver, p::BatchProblem, x0_lqr::Matrix{Float64}, Q_lqr::Matrix{Float64},
                             Qf_lqr::Matrix{Float64}, R_lqr::Matrix{Float64};
                             noise::Union{Nothing,Array{Float64,4}} = nothing, noise_id::Vector{Int64} = Int64[],
                             opts::TvlqrOptions = TvlqrOptions())
    T = size(p.x0, 2); N = p.N
    opts.n_knots = N; opts.n_tab = size(p.B_ECI, 2)
    Xs = zeros(7, N, T); Us = zeros(3, N - 1, T); K = zeros(3, 6, N - 1, T); st = Vector{TvlqrStats}(undef, T)
    check(s, ccall((:tsat_tvlqr_batch, LIB), Cint,
        (Ptr{Cvoid}, Ref{TvlqrOptions}, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32},
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{TvlqrStats}, Ptr{Int32}, Ptr{Int64}),
        s.handle, opts, T, size(p.B_ECI, 3), p.X, p.U, p.xf, p.B_ECI, p.btab_idx, p.tau0, p.dtau, p.dt, p.J,
        Q_lqr, Qf_lqr, R_lqr, x0_lqr, noise === nothing ? C_NULL : noise, Xs, Us, K, st,
        isempty(p.n_knots) ? C_NULL : p.n_knots, isempty(noise_id) ? C_NULL : noise_id), "tsat_tvlqr_batch")
    return Xs, Us, K, st
end

"attitude_simulation_resident(s, p, ...) — as above for the batch still resident after solve!(p, s): nothing is re-uploaded"
function attitude_simulation_resident(s::HIPSolver, p::BatchProblem, x0_lqr::Matrix{Float64}, Q_lqr::Matrix{Float64},
                                      Qf_lqr::Matrix{Float64}, R_lqr::Matrix{Float64};
                                      noise::Union{Nothing,Array{Float64,4}} = nothing, noise_id::Vector{Int64} = Int64[],
                                      opts::TvlqrOptions = TvlqrOptions())
    T = size(p.x0, 2); N = p.N
    Xs = zeros(7, N, T); Us = zeros(3, N - 1, T); K = zeros(3, 6, N - 1, T); st = Vector{TvlqrStats}(undef, T)
    check(s, ccall((:tsat_tvlqr_resident, LIB), Cint,
        (Ptr{Cvoid}, Ref{TvlqrOptions}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{TvlqrStats}, Ptr{Int64}),
        s.handle, opts, Q_lqr, Qf_lqr, R_lqr, x0_lqr, noise === nothing ? C_NULL : noise, Xs, Us, K, st,
        isempty(noise_id) ? C_NULL : noise_id), "tsat_tvlqr_resident")
    return Xs, Us, K, st
end

"""
receding_horizon!(s, p, n_steps; plant_integrator = 4) — `tsat_mpc_run` on the batch `p` (uploaded here): re-solve the
horizon every control step with the budget of `s.opts`, apply U[:,1] to the noise-free plant, shift the plan.
No reference equivalent (BASELINE.json configs[4]). Returns X_hist 7×(n_steps+1)×T, U_hist 3×n_steps×T.
"""
function receding_horizon!(s::HIPSolver, p::BatchProblem, n_steps::Integer; plant_integrator::Integer = 4)
    T = size(p.x0, 2); N = p.N
    o = s.opts; o.n_knots = N; o.n_tab = size(p.B_ECI, 2)
    check(s, ccall((:tsat_batch_reserve, LIB), Cint, (Ptr{Cvoid}, Int64, Int32, Int32, Int64, Int32),
                   s.handle, T, N, o.n_tab, size(p.B_ECI, 3), o.max_linesearch), "tsat_batch_reserve")
    check(s, ccall((:tsat_batch_upload, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        s.handle, p.x0, p.xf, p.B_ECI, p.btab_idx, p.tau0, p.dtau, p.dt, p.J, p.Q, p.Qf, p.R, p.u_min, p.u_max, p.U0),
        "tsat_batch_upload")
    isempty(p.n_knots) || check(s, ccall((:tsat_batch_knots, LIB), Cint, (Ptr{Cvoid}, Ptr{Int32}), s.handle, p.n_knots), "tsat_batch_knots")
    Xh = zeros(7, n_steps + 1, T); Uh = zeros(3, n_steps, T)
    check(s, ccall((:tsat_mpc_run, LIB), Cint, (Ptr{Cvoid}, Ref{Options}, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Stats}, Ptr{Cfloat}),
                   s.handle, o, n_steps, plant_integrator, Xh, Uh, C_NULL, C_NULL), "tsat_mpc_run")
    return Xh, Uh
end

"mpc_tally(s, T) — executed counts of the last receding_horizon! summed over its control steps: 4×T [backward, forward, dual updates, inner iterations]"
function mpc_tally(s::HIPSolver, T::Integer)
    t = zeros(Int64, 4, T)
    check(s, ccall((:tsat_mpc_tally, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}), s.handle, t), "tsat_mpc_tally")
    return t
end

# ---- sweep exchange across GPUs (include/tortoise_hip.h: tsat_comm_*, tsat_sweep_allgather) ----------------------------------
# One process per GPU (e.g. Distributed.jl workers or MPI ranks), each with its own HIPSolver and an equal shard of the sweep.
# Rank 0 makes the 128-byte communicator id and sends it to the others by whatever the host uses; after solve! every rank calls
# sweep_allgather and receives everybody's X (7 x N x world*T), U (3 x (N-1) x world*T) and stats in rank order — the result
# lists the reference's serial loop appends to (src/monte_carlo.jl:52-66, 199-235), assembled by one RCCL all-gather over xGMI.
const COMM_ID_BYTES = 128
"local probe, no communication: can this process load RCCL behind the C ABI? Agree on it across ranks BEFORE anybody calls comm_init"
comm_available() = ccall((:tsat_comm_available, LIB), Cint, ()) == 0
function comm_unique_id()
    id = zeros(UInt8, COMM_ID_BYTES)
    rc = ccall((:tsat_comm_unique_id, LIB), Cint, (Ptr{UInt8},), id)
    rc == 0 || error("tsat_comm_unique_id failed ($rc): is librccl.so.1 loadable?")
    return id
end
comm_init(s::HIPSolver, id::Vector{UInt8}, rank::Integer, world::Integer) =
    check(s, ccall((:tsat_comm_init, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Int32, Int32), s.handle, id, rank, world), "tsat_comm_init")
function sweep_allgather(s::HIPSolver, p::BatchProblem, world::Integer)
    T, N = size(p.x0, 2), p.N
    X = zeros(7, N, world * T); U = zeros(3, N - 1, world * T); st = Vector{Stats}(undef, world * T)
    check(s, ccall((:tsat_sweep_allgather, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Stats}, Int32),
                   s.handle, X, U, st, 0), "tsat_sweep_allgather")
    return X, U, st
end
comm_destroy(s::HIPSolver) = check(s, ccall((:tsat_comm_destroy, LIB), Cint, (Ptr{Cvoid},), s.handle), "tsat_comm_destroy")

"""
write_results(dir, n, A, sim_states, sim_control_inputs, B_ECI_total, t_total) — the files of src/monte_carlo.jl:334-343
(`{n}_A.h5` "A"; per trial `{n}_states_{i}.h5` "states", `{n}_control_{i}.h5` "control", `{n}_B_N_{i}.h5` "B_ECI",
`{n}_t_total_{i}.h5` "t_total"). Needs HDF5.jl, as the reference does.
"""
function write_results(dir, n, A, sim_states, sim_control_inputs, B_ECI_total, t_total)
    HDF5 = Base.require(Base.PkgId(Base.UUID("f67ccb44-e63f-5c2f-98bd-6dc0ccc4ba2f"), "HDF5"))
    HDF5.h5write(joinpath(dir, "$(n)_A.h5"), "A", A)
    for i in eachindex(sim_states)
        HDF5.h5write(joinpath(dir, "$(n)_states_$(i).h5"), "one_state", sim_states[i])      # the script writes both names (:338-339)
        HDF5.h5write(joinpath(dir, "$(n)_states_$(i).h5"), "states", sim_states[i])
        HDF5.h5write(joinpath(dir, "$(n)_control_$(i).h5"), "control", sim_control_inputs[i])
        HDF5.h5write(joinpath(dir, "$(n)_B_N_$(i).h5"), "B_ECI", B_ECI_total[i])
        HDF5.h5write(joinpath(dir, "$(n)_t_total_$(i).h5"), "t_total", collect(t_total[i]))
    end
end

# Entry points of include/tortoise_hip.h deliberately left without a binding here (read by tests/test_julia_shim.py): none.
const UNBOUND = Symbol[]

end # module
