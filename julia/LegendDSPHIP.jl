# LegendDSPHIP.jl — Julia host side of libldsp_hip.so (include/ldsp.h, ABI version 4): what a LegendDSP.jl maintainer
# adds as a package extension so that `dsp_icpc` / `dsp_sipm` and the filter functors / extractors they are built from run
# on an MI355X for batches that live in AMDGPU.ROCArray memory.
#
# STATUS: NOT EXECUTED.  Neither the build container nor the GPU box has a Julia toolchain (no `julia`, no depot, no
# network).  The file is complete and mechanical: every struct of include/ldsp.h with the same field order and types,
# one `ccall` per entry point, the lowering of DSPConfig / PropDict to the parameter blocks (field by field as
# legenddsp.jl_amd/config.py does and tests), `dsp_icpc` / `dsp_sipm` with their output Tables, and the functor overloads
# of the interface list at reference src/LegendDSP.jl:29-31.  tests/test_julia_binding.py parses the struct definitions
# below and checks field order, sizes and offsets against the C compiler's layout of include/ldsp.h and against
# `ldsp_abi_sizeof`, and that every entry point of the header has a `ccall` here with the right argument count.
#
# Reference call sites are cited as file:line relative to the LegendDSP.jl checkout.

module LegendDSPHIP

using LegendDSP, RadiationDetectorDSP, RadiationDetectorSignals, ArraysOfArrays, TypedTables, Unitful, IntervalSets, PropDicts
using AMDGPU: ROCArray, ROCVector
import AMDGPU
import RadiationDetectorDSP: fltinstance, rdfilt!, flt_output_length, flt_input_length, flt_output_smpltype, flt_input_smpltype,
    flt_output_time_axis, smplinfo, SamplingInfo, AbstractRadSigFilterInstance, LinearFiltering, NonlinearFiltering

const libldsp = get(ENV, "LDSP_HIP_LIB", joinpath(@__DIR__, "..", "legenddsp.jl_amd", "csrc", "libldsp_hip.so"))
const LDSP_ABI_VERSION = 4
const LDSP_MAX_TRIG = 64
const LDSP_ICPC_NCOLS = 48

# ------------------------------------------------------------------------------------------------------------------
# structs: field order / types exactly as include/ldsp.h (isbits, C layout)

struct LdspTrap
    navg::Int32
    ngap::Int32
    navg2::Int32
end

struct LdspCuspZac
    sigma::Float64
    flat::Int32
    length::Int32
    tau::Float64
    beta::Float64
end

struct LdspDni
    npts::Int32
    degree::Int32
end

struct LdspIcpcParams
    L::Int32
    _pad0::Int32
    t_first::Float64
    dt::Float64
    unit_per_us::Float64
    sat_low::Float64
    sat_high::Float64
    bl_from::Int32
    bl_until::Int32
    tail_from::Int32
    tail_until::Int32
    pz_c::Float64
    t0_trap::LdspTrap
    t0_mintot::Int32
    t0_threshold::Float64
    t0inv_trap::LdspTrap
    tx_mintot::Int32
    int_est::LdspDni
    qdrift_d1::Float64
    qdrift_d2::Float64
    lq_d1::Float64
    lq_d2::Float64
    trap_fixed::NTuple{3,LdspTrap}
    trap_opt::LdspTrap
    trap_pickoff::Float64
    sig_est::LdspDni
    cusp::LdspCuspZac
    zac::LdspCuspZac
    cusp_pickoff::Float64
    zac_pickoff::Float64
    sg_npts::NTuple{3,Int32}
    sg_degree::Int32
    cur_left::Float64
    cur_right::Float64
    intrace_nsigma::Float64
    intrace_mintot::Int32
    _pad1::Int32
    bl_left::Float64
    bl_right::Float64
end

struct LdspIcpcOut
    cols::NTuple{48,Ptr{Cvoid}}
    stride::Int64
end

struct LdspIcpcOpts
    ext_baseline::Ptr{Float32}
    ext_baseline_scale::Float64
    main_only::Int32
    in_u16::Int32
end

struct LdspSipmParams
    L::Int32
    _pad0::Int32
    t_first::Float64
    dt::Float64
    unit_per_us::Float64
    trunc_from::Int32
    trunc_until::Int32
    sg_npts::Int32
    sg_degree::Int32
    sg_mintot::Int32
    sg_maxtot::Int32
    sg_min_thr::Float64
    sg_max_thr::Float64
    sg_nsigma::Float64
    sg_min_dc_thr::Float64
    sg_max_dc_thr::Float64
    sg_nsigma_dc::Float64
    pz_c::Float64
    trap::LdspTrap
    trap_mintot::Int32
    trap_maxtot::Int32
    _pad1::Int32
    trap_min_thr::Float64
    trap_max_thr::Float64
    trap_nsigma::Float64
    trap_min_dc_thr::Float64
    trap_max_dc_thr::Float64
    trap_nsigma_dc::Float64
end

struct LdspTrigOut
    count::Ptr{Int32}
    x::Ptr{Float64}        # positions: Float64 like the reference's time axis (src/dsp_sipm.jl:87-88)
    x_high::Ptr{Float64}
    x_tot::Ptr{Float64}
    max::Ptr{Float32}
    cap::Int32
    _pad::Int32
end

struct LdspSipmOut
    scalars::NTuple{20,Ptr{Float32}}
    trig::LdspTrigOut
    trig_DC::LdspTrigOut
    trig_trap::LdspTrigOut
    trig_DC_trap::LdspTrigOut
end

struct LdspTrapGridParams
    L::Int32
    _pad0::Int32
    t_first::Float64
    dt::Float64
    bl_from::Int32
    bl_until::Int32
    pz_c::Float64
    sig_est::LdspDni
    pick_mode::Int32
    tx_mintot::Int32
    pick_time::Float64
end

const ICPC_COLS = (:blmean, :blsigma, :blslope, :bloffset, :tailmean, :tailsigma, :tailslope, :tailoffset,
    :t0, :t10, :t50, :t80, :t90, :t99, :t50_current, :drift_time, :tail_τ, :tail_mean, :tail_sigma, :e_max, :e_min,
    :e_10410, :e_535, :e_313, :e_10410_inv, :e_313_inv, :t0_inv, :e_trap, :e_cusp, :e_zac, :e_trap_max, :e_cusp_max, :e_zac_max,
    :t_trap_max, :t_cusp_max, :t_zac_max, :qdrift, :lq, :a_sg, :a_60, :a_100, :a_raw, :inTrace_intersect, :inTrace_n,
    :n_sat_low, :n_sat_high, :n_sat_low_cons, :n_sat_high_cons)
const ICPC_INT_COLS = (:inTrace_n, :n_sat_low, :n_sat_high, :n_sat_low_cons, :n_sat_high_cons)
const SIPM_SCALAR_COLS = (:t_max, :t_min, :t_max_lar, :t_min_lar, :e_max, :e_min, :e_max_lar, :e_min_lar,
    :blmean, :blsigma, :blslope, :bloffset, :wfmean, :wfsigma, :wfslope, :wfoffset,
    :threshold, :threshold_DC, :threshold_trap, :threshold_DC_trap)

# ------------------------------------------------------------------------------------------------------------------
# context, errors, helpers

struct LdspError <: Exception
    code::Int
    msg::String
end
last_error() = unsafe_string(ccall((:ldsp_last_error_string, libldsp), Cstring, ()))
function check(rc::Integer)
    rc == 0 && return nothing
    # LDSP_ERR_WINDOW (-2) is the reference's @assert on windows outside the trace (src/tailstats.jl:23-25)
    rc == -2 && throw(AssertionError(last_error()))
    throw(LdspError(rc, last_error()))
end

mutable struct LdspCtx
    h::Ptr{Cvoid}
end
function LdspCtx(dev::Integer = 0)
    ccall((:ldsp_abi_version, libldsp), Cint, ()) == LDSP_ABI_VERSION || error("libldsp_hip.so: ABI version mismatch")
    for (which, T) in enumerate((LdspIcpcParams, LdspIcpcOut, LdspSipmParams, LdspSipmOut, LdspTrigOut, LdspIcpcOpts))
        ccall((:ldsp_abi_sizeof, libldsp), Int64, (Cint,), which - 1) == sizeof(T) || error("struct size mismatch: $T")
    end
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:ldsp_ctx_create, libldsp), Cint, (Cint, Ptr{Ptr{Cvoid}}), dev, r))
    c = LdspCtx(r[])
    finalizer(c -> ccall((:ldsp_ctx_destroy, libldsp), Cint, (Ptr{Cvoid},), c.h), c)
    c
end
const _default_ctx = Dict{Int,LdspCtx}()
default_ctx(dev::Integer = AMDGPU.device_id(AMDGPU.device()) - 1) = get!(() -> LdspCtx(dev), _default_ctx, Int(dev))

set_stream!(c::LdspCtx, s::Ptr{Cvoid}) = check(ccall((:ldsp_ctx_set_stream, libldsp), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), c.h, s))
use_own_stream!(c::LdspCtx) = check(ccall((:ldsp_ctx_use_own_stream, libldsp), Cint, (Ptr{Cvoid},), c.h))
synchronize!(c::LdspCtx) = check(ccall((:ldsp_ctx_synchronize, libldsp), Cint, (Ptr{Cvoid},), c.h))
set_option!(c::LdspCtx, key::AbstractString, v::Integer) =
    check(ccall((:ldsp_ctx_set_option, libldsp), Cint, (Ptr{Cvoid}, Cstring, Int64), c.h, key, v))
enable_timing!(c::LdspCtx, on::Bool = true) = check(ccall((:ldsp_ctx_enable_timing, libldsp), Cint, (Ptr{Cvoid}, Cint), c.h, on))
function last_kernel_ms(c::LdspCtx)
    r = Ref{Cfloat}(0)
    check(ccall((:ldsp_ctx_last_kernel_ms, libldsp), Cint, (Ptr{Cvoid}, Ptr{Cfloat}), c.h, r))
    r[]
end
check_params(p::LdspIcpcParams) = check(ccall((:ldsp_icpc_check_params, libldsp), Cint, (Ref{LdspIcpcParams},), p))
last_kernel_name(c::LdspCtx) = unsafe_string(ccall((:ldsp_ctx_last_kernel_name, libldsp), Cstring, (Ptr{Cvoid},), c.h))
function last_stage_ms(c::LdspCtx, stage::Integer)
    r = Ref{Cfloat}(0)
    check(ccall((:ldsp_ctx_last_stage_ms, libldsp), Cint, (Ptr{Cvoid}, Cint, Ptr{Cfloat}), c.h, stage, r))
    r[]
end

# GPU batch of waveforms: the flat data of an ArrayOfSimilarVectors{Float32} in device memory, L x n column-major,
# i.e. row-major [n][L] as the C ABI wants it
const GPUWaveforms = ArrayOfRDWaveforms{<:Any,<:Any,<:Any,<:Any,<:VectorOfSimilarVectors{Float32,<:ROCArray}}
_flat(wvfs::GPUWaveforms) = flatview(wvfs.signal)
_axis(wvfs) = first(wvfs.time)                               # all traces share one time axis (src/dsp_icpc.jl:88,90)
_ns(t) = Float64(ustrip(u"ns", t))
_t0dt(wvfs) = (_ns(first(_axis(wvfs))), _ns(step(_axis(wvfs))))
nsmp(t, Δt) = round(Int32, ustrip(NoUnits, t / Δt))          # Julia's round (half to even) on the exact quotient
widx(t, t_first, Δt) = round(Int32, ustrip(NoUnits, (t - t_first) / Δt))   # 0-based sample index of time t
devptr(a::ROCArray{T}) where {T} = Ptr{T}(UInt(pointer(a)))
_rewrap(wvfs, y::ROCArray{Float32,2}, axis) = ArrayOfRDWaveforms((fill(axis, size(y, 2)), nestedview(y)))

# ------------------------------------------------------------------------------------------------------------------
# lowering of DSPConfig / PropDict to the parameter blocks  (legenddsp.jl_amd/config.py: lower_icpc, lower_sipm)

trap_samples(avg, gap, Δt, avg2 = avg) = LdspTrap(nsmp(avg, Δt), nsmp(gap, Δt), nsmp(avg2, Δt))
function sg_npoints(len, Δt)          # assumption A2 (DESIGN.md): round(len/Δt) made odd
    n = max(nsmp(len, Δt), Int32(1))
    iseven(n) ? n + Int32(1) : n
end
cuspzac_lowered(rt, ft, τ, len, beta, Δt) =
    LdspCuspZac(Float64(ustrip(NoUnits, rt / Δt)), nsmp(ft, Δt), nsmp(len, Δt), Float64(ustrip(NoUnits, τ / Δt)), Float64(beta))

"""Lower (DSPConfig, τ, pars_filter) and the sampling of the traces to `ldsp_icpc_params` — the parameter unpacking of
reference src/dsp_icpc.jl:64-99."""
function lower_icpc(config::DSPConfig, τ, pars_filter::PropDict, L::Integer, t_first, Δt)
    kw = config.kwargs_pars
    trap_rt, trap_ft = LegendDSP.get_fltpars(pars_filter, :trap, config)
    cusp_rt, cusp_ft = LegendDSP.get_fltpars(pars_filter, :cusp, config)
    zac_rt, zac_ft = LegendDSP.get_fltpars(pars_filter, :zac, config)
    sg_wl = LegendDSP.get_fltpars(pars_filter, :sg, config)
    bit_depth = kw.fc_bit_depth
    τ_off = 10000000.0u"µs"                                                   # src/dsp_icpc.jl:98
    t0p = kw.t0_flt_pars
    blw, tlw, cw = config.bl_window, config.tail_window, config.current_window
    LdspIcpcParams(
        Int32(L), Int32(0), _ns(t_first), _ns(Δt), 1000.0,
        0.0, Float64(2^bit_depth - bit_depth),                                   # :94
        widx(leftendpoint(blw), t_first, Δt), widx(rightendpoint(blw), t_first, Δt),
        widx(leftendpoint(tlw), t_first, Δt), widx(rightendpoint(tlw), t_first, Δt),
        Float64(ustrip(NoUnits, Δt / τ)),                                        # InvCRFilter(τ)          :119
        trap_samples(t0p[1], t0p[2], Δt, t0p[3]), max(Int32(1), nsmp(kw.t0_mintot, Δt)), Float64(config.t0_threshold),
        trap_samples(40u"ns", 100u"ns", Δt, 2000u"ns"),                          # default flt_pars         :207
        max(Int32(1), nsmp(kw.tx_mintot, Δt)),
        LdspDni(nsmp(kw.int_interpolation_length, Δt), Int32(kw.int_interpolation_order)),
        _ns(first(config.qdrift_int_length)), _ns(last(config.qdrift_int_length)),
        _ns(first(config.lq_int_length)), _ns(last(config.lq_int_length)),
        (trap_samples(10u"µs", 4u"µs", Δt), trap_samples(5u"µs", 3u"µs", Δt), trap_samples(3u"µs", 1u"µs", Δt)),   # :147-154
        trap_samples(trap_rt, trap_ft, Δt), _ns(trap_rt + trap_ft / 2),          # :160-163
        LdspDni(nsmp(kw.sig_interpolation_length, Δt), Int32(kw.sig_interpolation_order)),
        cuspzac_lowered(cusp_rt, cusp_ft, τ_off, config.flt_length_cusp, ustrip(NoUnits, config.flt_length_cusp / Δt), Δt),
        cuspzac_lowered(zac_rt, zac_ft, τ_off, config.flt_length_zac, ustrip(NoUnits, config.flt_length_zac / Δt), Δt),
        _ns(config.flt_length_cusp / 2), _ns(config.flt_length_zac / 2),         # :170,177
        (sg_npoints(sg_wl, Δt), sg_npoints(60u"ns", Δt), sg_npoints(100u"ns", Δt)), Int32(config.sg_flt_degree),   # :181-185
        _ns(leftendpoint(cw)), _ns(rightendpoint(cw)),
        Float64(config.inTraceCut_std_threshold), max(Int32(1), nsmp(kw.intrace_mintot, Δt)), Int32(0),
        _ns(leftendpoint(blw)), _ns(rightendpoint(blw)))
end

"""Lower the PropDict config of `dsp_sipm` (reference src/dsp_sipm.jl:49-78)."""
function lower_sipm(config::PropDict, pars_optimization::PropDict, L::Integer, t_first, Δt)
    sg, tr = config.filters.sg, config.filters.trap
    a, b = first(config.t0_hpge_window), last(config.t0_hpge_window)
    from = max(Int32(0), ceil(Int32, ustrip(NoUnits, (a - t_first) / Δt)))      # TruncateFilter(a..b): samples inside the closed interval
    until = min(Int32(L - 1), floor(Int32, ustrip(NoUnits, (b - t_first) / Δt)))
    LdspSipmParams(
        Int32(L), Int32(0), _ns(t_first), _ns(Δt), 1000.0, from, until,
        sg_npoints(pars_optimization.sg.wl, Δt), Int32(config.sg_flt_degree),
        max(Int32(1), nsmp(sg.min_tot_intersect, Δt)), max(Int32(1), nsmp(sg.max_tot_intersect, Δt)),
        Float64(sg.min_threshold), Float64(sg.max_threshold), Float64(sg.n_σ_threshold),
        Float64(sg.min_dc_threshold), Float64(sg.max_dc_threshold), Float64(sg.n_σ_dc_threshold),
        Float64(ustrip(NoUnits, Δt / tr.pz_tau)), trap_samples(tr.rt, tr.ft, Δt),
        max(Int32(1), nsmp(tr.min_tot_intersect, Δt)), max(Int32(1), nsmp(tr.max_tot_intersect, Δt)), Int32(0),
        Float64(tr.min_threshold), Float64(tr.max_threshold), Float64(tr.n_σ_threshold),
        Float64(tr.min_dc_threshold), Float64(tr.max_dc_threshold), Float64(tr.n_σ_dc_threshold))
end

# ------------------------------------------------------------------------------------------------------------------
# fused routines

"""`dsp_icpc(data, config, τ, pars_filter)` on a GPU batch — reference src/dsp_icpc.jl:62-230, same 53 columns."""
function LegendDSP.dsp_icpc(data::Table, config::DSPConfig, τ::Quantity, pars_filter::PropDict, wvfs::GPUWaveforms = data.waveform;
                            ctx::LdspCtx = default_ctx(), opts::Union{Nothing,LdspIcpcOpts} = nothing)
    x = _flat(wvfs)
    L, n = size(x)
    p = lower_icpc(config, τ, pars_filter, L, first(_axis(wvfs)), step(_axis(wvfs)))
    tab = ROCArray{Float32}(undef, LDSP_ICPC_NCOLS, n)                                   # [n][48] row-major
    out = LdspIcpcOut(ntuple(i -> Ptr{Cvoid}(UInt(pointer(tab)) + 4 * (i - 1)), 48), Int64(LDSP_ICPC_NCOLS))
    if opts === nothing
        check(ccall((:ldsp_icpc_run, libldsp), Cint,
                    (Ptr{Cvoid}, Ptr{Float32}, Int64, Ref{LdspIcpcParams}, Ref{LdspIcpcOut}),
                    ctx.h, devptr(x), n, Ref(p), Ref(out)))
    else    # e.g. LdspIcpcOpts(C_NULL, 1.0, 0, 1) with the UInt16 ADC counts reinterpreted: the kernel converts as it loads
        check(ccall((:ldsp_icpc_run_opts, libldsp), Cint,
                    (Ptr{Cvoid}, Ptr{Float32}, Int64, Ref{LdspIcpcParams}, Ref{LdspIcpcOpts}, Ref{LdspIcpcOut}),
                    ctx.h, devptr(x), n, Ref(p), Ref(opts), Ref(out)))
    end
    synchronize!(ctx)
    h = Array(tab)                                                                       # 192 B per trace over PCIe
    col(i) = ICPC_COLS[i] in ICPC_INT_COLS ? Int.(reinterpret(Int32, h[i, :])) : h[i, :]
    us(v) = v .* u"µs"
    nt = (; (ICPC_COLS[i] => col(i) for i in 1:48)...)
    TypedTables.Table(
        blmean = nt.blmean, blsigma = nt.blsigma, blslope = nt.blslope ./ u"ns", bloffset = nt.bloffset,
        tailmean = nt.tailmean, tailsigma = nt.tailsigma, tailslope = nt.tailslope ./ u"ns", tailoffset = nt.tailoffset,
        qc_label = fill(-1, n),                                                                         # :108
        t0 = us(nt.t0), t10 = us(nt.t10), t50 = us(nt.t50), t80 = us(nt.t80), t90 = us(nt.t90), t99 = us(nt.t99),
        t50_current = us(nt.t50_current), drift_time = nt.drift_time .* u"ns",
        tail_τ = nt.tail_τ .* u"ns", tail_mean = nt.tail_mean, tail_sigma = nt.tail_sigma,
        e_max = nt.e_max, e_min = nt.e_min,
        e_10410 = nt.e_10410, e_535 = nt.e_535, e_313 = nt.e_313, e_10410_inv = nt.e_10410_inv, e_313_inv = nt.e_313_inv,
        t0_inv = us(nt.t0_inv),
        e_trap = nt.e_trap, e_cusp = nt.e_cusp, e_zac = nt.e_zac,
        e_trap_max = nt.e_trap_max, e_cusp_max = nt.e_cusp_max, e_zac_max = nt.e_zac_max,
        t_trap_max = nt.t_trap_max .* u"ns", t_cusp_max = nt.t_cusp_max .* u"ns", t_zac_max = nt.t_zac_max .* u"ns",
        qdrift = nt.qdrift, lq = nt.lq,
        a_sg = nt.a_sg, a_60 = nt.a_60, a_100 = nt.a_100, a_raw = nt.a_raw,
        blfc = data.baseline, timestamp = data.timestamp, eventID_fadc = data.eventnumber, e_fc = data.daqenergy,   # :226
        inTrace_intersect = nt.inTrace_intersect .* u"ns", inTrace_n = nt.inTrace_n,
        n_sat_low = nt.n_sat_low, n_sat_high = nt.n_sat_high, n_sat_low_cons = nt.n_sat_low_cons, n_sat_high_cons = nt.n_sat_high_cons)
end

"BASELINE config 2 sub-chain: signalstats(bl).mean -> shift -> InvCR -> Trap(10 µs, 4 µs) -> maximum  (src/dsp_icpc.jl:102-105,119-120,147-148)"
function icpc_pz_trap(wvfs::GPUWaveforms, config::DSPConfig, τ, pars_filter::PropDict; ctx::LdspCtx = default_ctx())
    x = _flat(wvfs)
    L, n = size(x)
    p = lower_icpc(config, τ, pars_filter, L, first(_axis(wvfs)), step(_axis(wvfs)))
    blmean, e10410 = ROCVector{Float32}(undef, n), ROCVector{Float32}(undef, n)
    check(ccall((:ldsp_icpc_pz_trap_run, libldsp), Cint,
                (Ptr{Cvoid}, Ptr{Float32}, Int64, Ref{LdspIcpcParams}, Ptr{Float32}, Ptr{Float32}),
                ctx.h, devptr(x), n, Ref(p), devptr(blmean), devptr(e10410)))
    (blmean = blmean, e_10410 = e10410)
end
"The same on UInt16 ADC counts ([L, n] device array; converted by the kernel as it loads them: ldsp_icpc_pz_trap_run_u16)"
function icpc_pz_trap(x::ROCArray{UInt16,2}, t_first, dt, config::DSPConfig, τ, pars_filter::PropDict; ctx::LdspCtx = default_ctx())
    L, n = size(x)
    p = lower_icpc(config, τ, pars_filter, L, t_first, dt)
    blmean, e10410 = ROCVector{Float32}(undef, n), ROCVector{Float32}(undef, n)
    check(ccall((:ldsp_icpc_pz_trap_run_u16, libldsp), Cint,
                (Ptr{Cvoid}, Ptr{UInt16}, Int64, Ref{LdspIcpcParams}, Ptr{Float32}, Ptr{Float32}),
                ctx.h, devptr(x), n, Ref(p), devptr(blmean), devptr(e10410)))
    (blmean = blmean, e_10410 = e10410)
end

# one trigger group of dsp_sipm: count + four slabs of `cap` entries per trace
struct TrigBuffers
    count::ROCVector{Int32}
    x::ROCArray{Float64,2}
    x_high::ROCArray{Float64,2}
    x_tot::ROCArray{Float64,2}
    max::ROCArray{Float32,2}
end
TrigBuffers(n::Integer, cap::Integer) = TrigBuffers(ROCVector{Int32}(undef, n), (ROCArray{Float64}(undef, cap, n) for _ in 1:3)...,
                                                    ROCArray{Float32}(undef, cap, n))
_trig_out(t::TrigBuffers) = LdspTrigOut(devptr(t.count), devptr(t.x), devptr(t.x_high), devptr(t.x_tot), devptr(t.max), Int32(size(t.x, 1)), Int32(0))

# UInt16 ADC counts go through ldsp_sipm_run_u16 (converted by the kernel as it loads them)
function _sipm_run(x::ROCArray{UInt16,2}, p::LdspSipmParams, ctx::LdspCtx, cap::Integer)
    n = size(x, 2)
    sc = ROCArray{Float32}(undef, n, 20)
    groups = ntuple(_ -> TrigBuffers(n, cap), 4)
    out = LdspSipmOut(ntuple(i -> Ptr{Float32}(UInt(pointer(sc)) + 4 * n * (i - 1)), 20), _trig_out.(groups)...)
    check(ccall((:ldsp_sipm_run_u16, libldsp), Cint,
                (Ptr{Cvoid}, Ptr{UInt16}, Int64, Ref{LdspSipmParams}, Ref{LdspSipmOut}),
                ctx.h, devptr(x), n, Ref(p), Ref(out)))
    synchronize!(ctx)
    sc, groups
end

function _sipm_run(x::ROCArray{Float32,2}, p::LdspSipmParams, ctx::LdspCtx, cap::Integer)
    n = size(x, 2)
    sc = ROCArray{Float32}(undef, n, 20)
    groups = ntuple(_ -> TrigBuffers(n, cap), 4)
    out = LdspSipmOut(ntuple(i -> Ptr{Float32}(UInt(pointer(sc)) + 4 * n * (i - 1)), 20), _trig_out.(groups)...)
    check(ccall((:ldsp_sipm_run, libldsp), Cint,
                (Ptr{Cvoid}, Ptr{Float32}, Int64, Ref{LdspSipmParams}, Ref{LdspSipmOut}),
                ctx.h, devptr(x), n, Ref(p), Ref(out)))
    synchronize!(ctx)
    sc, groups
end

# VectorOfVectors of one field: every trigger of every trace (the reference pushes every crossing, src/intersect_maximum.jl:49-56)
function _ragged(field::Matrix{<:AbstractFloat}, count::Vector{Int32}, unit)
    cap = size(field, 1)
    VectorOfVectors([field[1:min(Int(c), cap), i] .* unit for (i, c) in enumerate(count)])
end

"""`dsp_sipm(data, config, pars_optimization)` on a GPU batch — reference src/dsp_sipm.jl:47-158: 24 scalar columns
(4 passthrough) and 12 `VectorOfVectors` columns.  Traces with more triggers than the default slab run again with
slabs sized from their counts (two-pass count-then-fill), nothing is truncated."""
function LegendDSP.dsp_sipm(data::Table, config::PropDict, pars_optimization::PropDict, wvfs::GPUWaveforms = data.waveform;
                            ctx::LdspCtx = default_ctx())
    x = _flat(wvfs)
    L, n = size(x)
    p = lower_sipm(config, pars_optimization, L, first(_axis(wvfs)), step(_axis(wvfs)))
    sc, groups = _sipm_run(x, p, ctx, LDSP_MAX_TRIG)
    h = Array(sc)
    s = (; (SIPM_SCALAR_COLS[i] => h[:, i] for i in 1:20)...)
    counts = [Array(g.count) for g in groups]
    fields = [(Array(g.x), Array(g.x_high), Array(g.x_tot), Array(g.max)) for g in groups]
    mx = maximum(maximum(c; init = Int32(0)) for c in counts)
    if mx > LDSP_MAX_TRIG                                  # second pass for the overflowing traces only
        rows = findall(i -> any(c[i] > LDSP_MAX_TRIG for c in counts), 1:n)
        _, g2 = _sipm_run(x[:, rows], p, ctx, nextpow(2, Int(mx)))
        for (k, g) in enumerate(g2)
            f2 = (Array(g.x), Array(g.x_high), Array(g.x_tot), Array(g.max))
            grown = ntuple(j -> vcat(fields[k][j], fill(eltype(f2[j])(NaN), size(f2[j], 1) - size(fields[k][j], 1), n)), 4)
            for (jj, i) in enumerate(rows), j in 1:4
                grown[j][:, i] .= f2[j][:, jj]
            end
            fields[k] = grown
        end
    end
    vv(k, j, unit) = _ragged(fields[k][j], counts[k], unit)
    ns = u"ns"
    TypedTables.Table(
        blfc = data.baseline, timestamp = data.timestamp, eventID_fadc = data.eventnumber, e_fc = data.daqenergy,
        t_max = s.t_max .* u"µs", t_min = s.t_min .* u"µs", t_max_lar = s.t_max_lar .* u"µs", t_min_lar = s.t_min_lar .* u"µs",
        e_max = s.e_max, e_min = s.e_min, e_max_lar = s.e_max_lar, e_min_lar = s.e_min_lar,
        blmean = s.blmean, blsigma = s.blsigma, blslope = s.blslope ./ ns, bloffset = s.bloffset,
        wfmean = s.wfmean, wfsigma = s.wfsigma, wfslope = s.wfslope ./ ns, wfoffset = s.wfoffset,
        threshold = s.threshold, threshold_DC = s.threshold_DC,
        trig_pos = vv(1, 1, ns), trig_max = vv(1, 4, 1), trig_pos_DC = vv(2, 1, ns), trig_max_DC = vv(2, 4, 1),     # :149-152
        threshold_trap = s.threshold_trap, threshold_DC_trap = s.threshold_DC_trap,
        trig_pos_trap = vv(3, 1, ns), trig_pos_high_trap = vv(3, 2, ns), trig_pos_tot_trap = vv(3, 3, ns), trig_max_trap = vv(3, 4, 1),
        trig_pos_DC_trap = vv(4, 1, ns), trig_pos_high_DC_trap = vv(4, 2, ns), trig_pos_tot_DC_trap = vv(4, 3, ns), trig_max_DC_trap = vv(4, 4, 1))
end

# ------------------------------------------------------------------------------------------------------------------
# filter-optimisation grid scans (SURVEY 8(f) row 1)

function _grid_params(wvfs, config::DSPConfig, τ, pick_mode::Integer, pick_time)
    t_first, Δt = first(_axis(wvfs)), step(_axis(wvfs))
    kw = config.kwargs_pars
    blw = config.bl_window
    LdspTrapGridParams(Int32(size(_flat(wvfs), 1)), Int32(0), _ns(t_first), _ns(Δt),
        widx(leftendpoint(blw), t_first, Δt), widx(rightendpoint(blw), t_first, Δt), Float64(ustrip(NoUnits, Δt / τ)),
        LdspDni(nsmp(kw.sig_interpolation_length, Δt), Int32(kw.sig_interpolation_order)),
        Int32(pick_mode), max(Int32(1), nsmp(kw.tx_mintot, Δt)), _ns(pick_time))
end

"`dsp_trap_rt_optimization` / `dsp_trap_ft_optimization` (src/dsp_filter_optimization.jl:102-133, 241-274): [grid, N] energies in one launch"
function trap_grid(wvfs::GPUWaveforms, config::DSPConfig, τ, traps::Vector{LdspTrap}, offsets::Union{Nothing,Vector{Float64}}, pick_time;
                   ctx::LdspCtx = default_ctx())
    x = _flat(wvfs)
    n, G = size(x, 2), length(traps)
    p = _grid_params(wvfs, config, τ, offsets === nothing ? 0 : 1, pick_time)
    out = ROCArray{Float32}(undef, n, G)
    check(ccall((:ldsp_trap_grid_run, libldsp), Cint,
                (Ptr{Cvoid}, Ptr{Float32}, Int64, Ref{LdspTrapGridParams}, Int32, Ptr{LdspTrap}, Ptr{Float64}, Ptr{Float32}),
                ctx.h, devptr(x), n, Ref(p), G, traps, offsets === nothing ? C_NULL : offsets, devptr(out)))
    out
end

"CUSP / ZAC scans (src/dsp_filter_optimization.jl:145-229, 286-374): taps [Lf, G] on the host, one FIR per grid point"
function fir_grid(wvfs::GPUWaveforms, config::DSPConfig, τ, taps::Matrix{Float64}, offsets::Union{Nothing,Vector{Float64}}, pick_time;
                  ctx::LdspCtx = default_ctx())
    x = _flat(wvfs)
    n = size(x, 2)
    Lf, G = size(taps)
    p = _grid_params(wvfs, config, τ, offsets === nothing ? 0 : 1, pick_time)
    out = ROCArray{Float32}(undef, n, G)
    check(ccall((:ldsp_fir_grid_run, libldsp), Cint,
                (Ptr{Cvoid}, Ptr{Float32}, Int64, Ref{LdspTrapGridParams}, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float32}),
                ctx.h, devptr(x), n, Ref(p), G, Lf, taps, offsets === nothing ? C_NULL : offsets, devptr(out)))
    out
end

"`dsp_sg_optimization` (src/dsp_filter_optimization.jl:393-441): A per window length, E, t50, baseline mean and slope"
function sg_grid(wvfs::GPUWaveforms, config::DSPConfig, τ, trap::LdspTrap, trap_offset, npts::Vector{Int32}, degree::Integer,
                 from::Vector{Int32}, until::Vector{Int32}; ctx::LdspCtx = default_ctx())
    x = _flat(wvfs)
    n, W = size(x, 2), length(npts)
    p = _grid_params(wvfs, config, τ, 1, 0.0u"ns")
    amax = ROCArray{Float32}(undef, n, W)
    energy, t50, blm, bls = (ROCVector{Float32}(undef, n) for _ in 1:4)
    check(ccall((:ldsp_sg_grid_run, libldsp), Cint,
                (Ptr{Cvoid}, Ptr{Float32}, Int64, Ref{LdspTrapGridParams}, Ref{LdspTrap}, Float64, Float64, Int32, Ptr{Int32}, Int32,
                 Ptr{Int32}, Ptr{Int32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
                ctx.h, devptr(x), n, Ref(p), Ref(trap), _ns(trap_offset), 1000.0, W, npts, degree, from, until,
                devptr(amax), devptr(energy), devptr(t50), devptr(blm), devptr(bls)))
    (amax = amax, energy = energy, t50 = t50, blmean = blm, blslope = bls)
end

# ------------------------------------------------------------------------------------------------------------------
# filter functors on a GPU batch: flt(wvfs) = what RadiationDetectorDSP's broadcast flt.(wvfs) returns
# (fltinstance / rdfilt! / flt_output_length / flt_output_time_axis protocol, reference src/LegendDSP.jl:29-31;
# in-repo instances src/derivative.jl:37-55, src/haar_filter.jl:17-39, src/moving_window_multi.jl:70-129)

# valid-mode filters stamp an output sample with the time of the LAST input sample under the kernel (assumption A1)
_trailing_axis(axis, flen::Integer) = axis[flen:end]

function (flt::InvCRFilter)(wvfs::GPUWaveforms; ctx::LdspCtx = default_ctx())
    x = _flat(wvfs); L, n = size(x); y = similar(x)
    check(ccall((:ldsp_rdfilt_invcr, libldsp), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64, Int32, Float64, Ptr{Float32}),
                ctx.h, devptr(x), n, L, ustrip(NoUnits, step(_axis(wvfs)) / flt.cr), devptr(y)))
    _rewrap(wvfs, y, _axis(wvfs))
end

function (flt::IntegratorFilter)(wvfs::GPUWaveforms; ctx::LdspCtx = default_ctx())
    x = _flat(wvfs); L, n = size(x); y = similar(x)
    check(ccall((:ldsp_rdfilt_integrator, libldsp), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64, Int32, Float64, Ptr{Float32}),
                ctx.h, devptr(x), n, L, Float64(flt.gain), devptr(y)))
    _rewrap(wvfs, y, _axis(wvfs))
end

function (flt::TrapezoidalChargeFilter)(wvfs::GPUWaveforms; ctx::LdspCtx = default_ctx())
    x = _flat(wvfs); L, n = size(x)
    t = trap_samples(flt.avgtime, flt.gaptime, step(_axis(wvfs)), flt.avgtime2)
    flen = t.navg + t.ngap + t.navg2
    y = ROCArray{Float32}(undef, L - flen + 1, n)
    check(ccall((:ldsp_rdfilt_trap, libldsp), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64, Int32, LdspTrap, Ptr{Float32}),
                ctx.h, devptr(x), n, L, t, devptr(y)))
    _rewrap(wvfs, y, _trailing_axis(_axis(wvfs), flen))
end

function _fir(wvfs::GPUWaveforms, h::Vector{Float64}, ctx::LdspCtx)
    x = _flat(wvfs); L, n = size(x)
    y = ROCArray{Float32}(undef, L - length(h) + 1, n)
    check(ccall((:ldsp_rdfilt_fir, libldsp), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64, Int32, Ptr{Float64}, Int32, Ptr{Float32}),
                ctx.h, devptr(x), n, L, h, length(h), devptr(y)))
    _rewrap(wvfs, y, _trailing_axis(_axis(wvfs), length(h)))
end
function cusp_coeffs(p::LdspCuspZac)
    h = Vector{Float64}(undef, p.length)
    check(ccall((:ldsp_cusp_coeffs, libldsp), Cint, (Ref{LdspCuspZac}, Ptr{Float64}), Ref(p), h)); h
end
function zac_coeffs(p::LdspCuspZac)
    h = Vector{Float64}(undef, p.length)
    check(ccall((:ldsp_zac_coeffs, libldsp), Cint, (Ref{LdspCuspZac}, Ptr{Float64}), Ref(p), h)); h
end
function sg_coeffs(npts::Integer, degree::Integer, derivative::Integer)
    h = Vector{Float64}(undef, npts)
    check(ccall((:ldsp_sg_coeffs, libldsp), Cint, (Int32, Int32, Int32, Ptr{Float64}), npts, degree, derivative, h)); h
end
(flt::CUSPChargeFilter)(wvfs::GPUWaveforms; ctx::LdspCtx = default_ctx()) =
    _fir(wvfs, cusp_coeffs(cuspzac_lowered(flt.sigma, flt.toplen, flt.tau, flt.length, flt.beta, step(_axis(wvfs)))), ctx)
(flt::ZACChargeFilter)(wvfs::GPUWaveforms; ctx::LdspCtx = default_ctx()) =
    _fir(wvfs, zac_coeffs(cuspzac_lowered(flt.sigma, flt.toplen, flt.tau, flt.length, flt.beta, step(_axis(wvfs)))), ctx)
(flt::SavitzkyGolayFilter)(wvfs::GPUWaveforms; ctx::LdspCtx = default_ctx()) =
    _fir(wvfs, sg_coeffs(sg_npoints(flt.length, step(_axis(wvfs))), flt.degree, flt.derivative), ctx)

function (flt::LegendDSP.DerivativeFilter)(wvfs::GPUWaveforms; ctx::LdspCtx = default_ctx())       # src/derivative.jl:26-55
    x = _flat(wvfs); L, n = size(x); y = similar(x)
    check(ccall((:ldsp_rdfilt_derivative, libldsp), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64, Int32, Float64, Ptr{Float32}),
                ctx.h, devptr(x), n, L, Float64(ustrip(flt.gain)), devptr(y)))
    _rewrap(wvfs, y, _axis(wvfs))
end

function (flt::LegendDSP.HaarAveragingFilter)(wvfs::GPUWaveforms; ctx::LdspCtx = default_ctx())   # src/haar_filter.jl:3-39
    x = _flat(wvfs); L, n = size(x)
    ds = Int32(flt.down_sampling_rate)
    y = ROCArray{Float32}(undef, cld(L, ds), n)
    check(ccall((:ldsp_rdfilt_haar, libldsp), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64, Int32, Int32, Ptr{Float32}),
                ctx.h, devptr(x), n, L, ds, devptr(y)))
    _rewrap(wvfs, y, _axis(wvfs)[1:ds:end])
end

function (flt::LegendDSP.MovingWindowFilter)(wvfs::GPUWaveforms; ctx::LdspCtx = default_ctx())    # src/moving_window_multi.jl:99-116
    x = _flat(wvfs); L, n = size(x); y = similar(x)
    check(ccall((:ldsp_rdfilt_moving_window, libldsp), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64, Int32, Int32, Ptr{Float32}),
                ctx.h, devptr(x), n, L, nsmp(flt.length, step(_axis(wvfs))), devptr(y)))
    _rewrap(wvfs, y, _axis(wvfs))
end
function (flt::LegendDSP.MovingWindowMultiFilter)(wvfs::GPUWaveforms; ctx::LdspCtx = default_ctx())   # :118-129
    x = _flat(wvfs); L, n = size(x); y = similar(x)
    check(ccall((:ldsp_rdfilt_moving_window_multi, libldsp), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64, Int32, Int32, Ptr{Float32}),
                ctx.h, devptr(x), n, L, nsmp(flt.length, step(_axis(wvfs))), devptr(y)))
    _rewrap(wvfs, y, _axis(wvfs))
end

# shift_waveform / multiply_waveform / reverse_waveform / TruncateFilter: one entry point
function _affine(wvfs::GPUWaveforms, from::Integer, until::Integer, scale, shift, per_trace, reverse::Bool, axis, ctx::LdspCtx)
    x = _flat(wvfs); L, n = size(x)
    y = ROCArray{Float32}(undef, until - from + 1, n)
    check(ccall((:ldsp_rdfilt_affine, libldsp), Cint,
                (Ptr{Cvoid}, Ptr{Float32}, Int64, Int32, Int32, Int32, Float64, Float64, Ptr{Float32}, Int32, Ptr{Float32}),
                ctx.h, devptr(x), n, L, from, until, Float64(scale), Float64(shift), per_trace === nothing ? C_NULL : devptr(per_trace),
                reverse, devptr(y)))
    _rewrap(wvfs, y, axis)
end
RadiationDetectorDSP.shift_waveform(wvfs::GPUWaveforms, c::Real; ctx = default_ctx()) =
    _affine(wvfs, 0, size(_flat(wvfs), 1) - 1, 1.0, c, nothing, false, _axis(wvfs), ctx)                 # src/dsp_icpc.jl:105
RadiationDetectorDSP.shift_waveform(wvfs::GPUWaveforms, c::ROCVector{Float32}; ctx = default_ctx()) =
    _affine(wvfs, 0, size(_flat(wvfs), 1) - 1, 1.0, 0.0, c, false, _axis(wvfs), ctx)
RadiationDetectorDSP.multiply_waveform(wvfs::GPUWaveforms, c::Real; ctx = default_ctx()) =
    _affine(wvfs, 0, size(_flat(wvfs), 1) - 1, c, 0.0, nothing, false, _axis(wvfs), ctx)                 # :199
RadiationDetectorDSP.reverse_waveform(wvfs::GPUWaveforms; ctx = default_ctx()) =
    _affine(wvfs, 0, size(_flat(wvfs), 1) - 1, 1.0, 0.0, nothing, true, _axis(wvfs), ctx)                # src/dsp_routines.jl:79
function (flt::TruncateFilter)(wvfs::GPUWaveforms; ctx::LdspCtx = default_ctx())                        # src/dsp_sipm.jl:94
    ax = _axis(wvfs)
    from = max(0, ceil(Int, ustrip(NoUnits, (leftendpoint(flt.interval) - first(ax)) / step(ax))))
    until = min(length(ax) - 1, floor(Int, ustrip(NoUnits, (rightendpoint(flt.interval) - first(ax)) / step(ax))))
    _affine(wvfs, from, until, 1.0, 0.0, nothing, false, ax[from+1:until+1], ctx)
end

# QC classifier front end (src/dsp_ml_routines.jl:9-70): Haar x levels, normalisation -> feature matrix for f_evaluate_qc
function qc_features(wvfs::GPUWaveforms, levels::Integer, bl_from::Integer = -1, bl_until::Integer = -1; ctx::LdspCtx = default_ctx())
    x = _flat(wvfs); L, n = size(x)
    Lout = ccall((:ldsp_qc_features_len, libldsp), Int32, (Int32, Int32), L, levels)
    feat, norm = ROCArray{Float32}(undef, Lout, n), ROCVector{Float32}(undef, n)
    check(ccall((:ldsp_qc_features, libldsp), Cint,
                (Ptr{Cvoid}, Ptr{Float32}, Int64, Int32, Int32, Int32, Int32, Ptr{Float32}, Ptr{Float32}),
                ctx.h, devptr(x), n, L, levels, bl_from, bl_until, devptr(feat), devptr(norm)))
    feat, norm
end

# ------------------------------------------------------------------------------------------------------------------
# extractors on a GPU batch: struct-of-arrays NamedTuples, as broadcasting the reference's callables yields

function _window(wvfs, start, stop)
    ax = _axis(wvfs)
    a, b = widx(start, first(ax), step(ax)), widx(stop, first(ax), step(ax))
    0 <= a <= b <= length(ax) - 1 || throw(AssertionError("window outside the trace"))           # src/tailstats.jl:23-25
    a, b
end
_outs(n, k) = ntuple(_ -> ROCVector{Float32}(undef, n), k)

function RadiationDetectorDSP.signalstats(wvfs::GPUWaveforms, start, stop; ctx::LdspCtx = default_ctx())
    x = _flat(wvfs); L, n = size(x); a, b = _window(wvfs, start, stop); t0, dt = _t0dt(wvfs)
    o = _outs(n, 4)
    check(ccall((:ldsp_signalstats, libldsp), Cint,
                (Ptr{Cvoid}, Ptr{Float32}, Int64, Int32, Int32, Int32, Float64, Float64, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
                ctx.h, devptr(x), n, L, a, b, t0, dt, devptr.(o)...))
    (mean = o[1], sigma = o[2], slope = o[3], offset = o[4])
end
function LegendDSP.tailstats(wvfs::GPUWaveforms, start, stop; ctx::LdspCtx = default_ctx())         # src/tailstats.jl:13-72
    x = _flat(wvfs); L, n = size(x); a, b = _window(wvfs, start, stop); t0, dt = _t0dt(wvfs)
    o = _outs(n, 3)
    check(ccall((:ldsp_tailstats, libldsp), Cint,
                (Ptr{Cvoid}, Ptr{Float32}, Int64, Int32, Int32, Int32, Float64, Float64, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
                ctx.h, devptr(x), n, L, a, b, t0, dt, devptr.(o)...))
    (mean = o[1], sigma = o[2], τ = o[3])
end
function LegendDSP.extremestats(wvfs::GPUWaveforms, start = first(_axis(wvfs)), stop = last(_axis(wvfs)); ctx::LdspCtx = default_ctx())   # src/extremestats.jl:14-40
    x = _flat(wvfs); L, n = size(x); a, b = _window(wvfs, start, stop); t0, dt = _t0dt(wvfs)
    o = _outs(n, 4)
    check(ccall((:ldsp_extremestats, libldsp), Cint,
                (Ptr{Cvoid}, Ptr{Float32}, Int64, Int32, Int32, Int32, Float64, Float64, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
                ctx.h, devptr(x), n, L, a, b, t0, dt, devptr.(o)...))
    (min = o[1], max = o[2], tmin = o[3], tmax = o[4])
end
function LegendDSP.thresholdstats(wvfs::GPUWaveforms, lo = -Inf, hi = Inf; ctx::LdspCtx = default_ctx())        # src/thresholdstats.jl:14-41
    x = _flat(wvfs); L, n = size(x); o = ROCVector{Float32}(undef, n)
    check(ccall((:ldsp_thresholdstats, libldsp), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64, Int32, Float64, Float64, Ptr{Float32}),
                ctx.h, devptr(x), n, L, lo, hi, devptr(o)))
    o
end
function LegendDSP.thresholdstats_mad(wvfs::GPUWaveforms, lo = -Inf, hi = Inf; ctx::LdspCtx = default_ctx())    # :56-71
    x = _flat(wvfs); L, n = size(x); o = ROCVector{Float32}(undef, n)
    check(ccall((:ldsp_thresholdstats_mad, libldsp), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64, Int32, Float64, Float64, Ptr{Float32}),
                ctx.h, devptr(x), n, L, lo, hi, devptr(o)))
    o
end
function LegendDSP.saturation(wvfs::GPUWaveforms, low, high; ctx::LdspCtx = default_ctx())                      # src/saturation.jl:12-65
    x = _flat(wvfs); L, n = size(x)
    o = ntuple(_ -> ROCVector{Int32}(undef, n), 4)
    check(ccall((:ldsp_saturation, libldsp), Cint,
                (Ptr{Cvoid}, Ptr{Float32}, Int64, Int32, Int32, Int32, Float64, Float64, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}),
                ctx.h, devptr(x), n, L, 0, L - 1, low, high, devptr.(o)...))
    (low = o[1], high = o[2], max_cons_low = o[3], max_cons_high = o[4])
end
function LegendDSP.get_wvf_maximum(wvfs::GPUWaveforms, start, stop; ctx::LdspCtx = default_ctx())               # src/interpolation.jl:21-46
    x = _flat(wvfs); L, n = size(x); a, b = _window(wvfs, start, stop); o = ROCVector{Float32}(undef, n)
    check(ccall((:ldsp_get_wvf_maximum, libldsp), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64, Int32, Int32, Int32, Ptr{Float32}),
                ctx.h, devptr(x), n, L, a, b, devptr(o)))
    o
end

function (f::Intersect)(wvfs::GPUWaveforms, thr::ROCVector{Float32}; ctx::LdspCtx = default_ctx())              # src/dsp_routines.jl:18,35,74
    x = _flat(wvfs); L, n = size(x); t0, dt = _t0dt(wvfs)
    xo, mult = ROCVector{Float32}(undef, n), ROCVector{Int32}(undef, n)
    check(ccall((:ldsp_intersect, libldsp), Cint,
                (Ptr{Cvoid}, Ptr{Float32}, Int64, Int32, Float64, Float64, Ptr{Float32}, Int32, Ptr{Float32}, Ptr{Int32}),
                ctx.h, devptr(x), n, L, t0, dt, devptr(thr), max(Int32(1), nsmp(f.mintot, step(_axis(wvfs)))), devptr(xo), devptr(mult)))
    (x = xo, multiplicity = mult)
end

function (f::LegendDSP.IntersectMaximum)(wvfs::GPUWaveforms, thr::ROCVector{Float32}; ctx::LdspCtx = default_ctx(), cap::Integer = LDSP_MAX_TRIG)   # src/intersect_maximum.jl:18-119
    x = _flat(wvfs); L, n = size(x); t0, dt = _t0dt(wvfs); Δt = step(_axis(wvfs))
    t = TrigBuffers(n, cap)
    check(ccall((:ldsp_intersect_maximum, libldsp), Cint,
                (Ptr{Cvoid}, Ptr{Float32}, Int64, Int32, Float64, Float64, Ptr{Float32}, Int32, Int32, Ref{LdspTrigOut}),
                ctx.h, devptr(x), n, L, t0, dt, devptr(thr), max(Int32(1), nsmp(f.mintot, Δt)), max(Int32(1), nsmp(f.maxtot, Δt)), Ref(_trig_out(t))))
    synchronize!(ctx)
    count = Array(t.count)
    mx = maximum(count; init = Int32(0))
    mx > cap && return f(wvfs, thr; ctx = ctx, cap = nextpow(2, Int(mx)))      # every crossing is returned: run again with room for all
    ns = u"ns"
    (x = _ragged(Array(t.x), count, ns), x_high = _ragged(Array(t.x_high), count, ns), x_tot = _ragged(Array(t.x_tot), count, ns),
     max = _ragged(Array(t.max), count, 1), multiplicity = count)
end

function (f::LegendDSP.MultiIntersect)(wvfs::GPUWaveforms; ctx::LdspCtx = default_ctx())                         # src/multi_intersect.jl:26-104
    x = _flat(wvfs); L, n = size(x); t0, dt = _t0dt(wvfs)
    r = collect(Float64, f.threshold_ratios); K = length(r)
    xo, status = ROCArray{Float32}(undef, K, n), ROCVector{Int32}(undef, n)
    check(ccall((:ldsp_multi_intersect, libldsp), Cint,
                (Ptr{Cvoid}, Ptr{Float32}, Int64, Int32, Float64, Float64, Ptr{Float64}, Int32, Int32, Int32, Int32, Int32, Ptr{Float32}, Ptr{Int32}),
                ctx.h, devptr(x), n, L, t0, dt, r, K, max(Int32(1), nsmp(f.mintot, step(_axis(wvfs)))), f.n, f.d, f.sampling_rate, devptr(xo), devptr(status)))
    synchronize!(ctx)
    any(!iszero, Array(status)) && throw(AssertionError("cannot interpolate intersect on left boundary"))   # :77-78
    xo
end

function (f::SignalEstimator{<:PolynomialDNI})(wvfs::GPUWaveforms, t::ROCVector{Float32}; ctx::LdspCtx = default_ctx())   # src/dsp_icpc.jl:157-177
    x = _flat(wvfs); L, n = size(x); t0, dt = _t0dt(wvfs)
    est = LdspDni(nsmp(f.method.length, step(_axis(wvfs))), Int32(f.method.degree))
    o = ROCVector{Float32}(undef, n)
    check(ccall((:ldsp_signal_estimator, libldsp), Cint,
                (Ptr{Cvoid}, Ptr{Float32}, Int64, Int32, Float64, Float64, Ptr{Float32}, LdspDni, Ptr{Float32}),
                ctx.h, devptr(x), n, L, t0, dt, devptr(t), est, devptr(o)))
    o
end

end # module
