# WaterLilyHIPExt — Julia-side binding of libwlhip.so (include/wlhip.h) for WaterLily.jl.
#
# STATUS: written against the reference's sources, NOT executed — no Julia runtime exists in the build
# container or on the GPU box (SURVEY.md §0).  Everything with logic lives in the library; this file is
# mechanical dispatch glue: one array type + one method per hot-path function, `ccall`ing the C ABI.
# It imitates the pattern of ext/WaterLilyAMDGPUExt.jl (array type selects the backend) but does not use
# AMDGPU.jl / KernelAbstractions.
#
# Use:   using WaterLily, WaterLilyHIP
#        sim = Simulation((512,512,512),(0,0,0),512; U=1, ν=512/1600, T=Float32, mem=HipArray)
#        sim_step!(sim; remeasure=false)
module WaterLilyHIPExt

using WaterLily
import WaterLily: BC!, perBC!, exitBC!, conv_diff!, BDIM!, scale_u!, CFL, L₂, mom_step!, mom_project!,
                  set_diag!, update!, mult!, residual!, increment!, Jacobi!, GaussSeidelRB!, pcg!, restrict!, prolongate!,
                  restrictL!, solver!, Vcycle!, L₁, L∞, quick, vanLeer, cds, Flow, Poisson, MultiLevelPoisson, AbstractPoisson

const libwlhip = get(ENV, "WLHIP_LIB", "libwlhip.so")

# ---- error convention: 0 ok, >0 hipError_t, <0 WL_E* -------------------------------------------------
function chk(rc::Cint)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:wl_last_error_string, libwlhip), Cstring, ()))
    rc == -3 && throw(AssertionError("MultiLevelPoisson requires size=a2ⁿ, where n>2"))   # src/MultiLevelPoisson.jl:73-74
    error("libwlhip error $rc: $msg")
end
__init__() = chk(ccall((:wl_init, libwlhip), Cint, (Cint,), 0))   # like ext/WaterLilyAMDGPUExt.jl:11

# ---- the array type: dense column-major Float32 in HBM, byte-identical to the Julia Array layout --------
mutable struct HipArray{T,N} <: AbstractArray{T,N}
    ptr::Ptr{T}
    dims::NTuple{N,Int}
    function HipArray{T,N}(::UndefInitializer, dims::NTuple{N,Int}) where {T,N}
        T === Float32 || error("the HIP path computes in Float32")
        p = Ref{Ptr{Cvoid}}()
        chk(ccall((:wl_malloc, libwlhip), Cint, (Ref{Ptr{Cvoid}}, Csize_t), p, prod(dims) * sizeof(T)))
        a = new{T,N}(Ptr{T}(p[]), dims)
        finalizer(x -> ccall((:wl_free, libwlhip), Cint, (Ptr{Cvoid},), x.ptr), a)   # Julia owns lifetimes (SURVEY §8b)
    end
end
# `zeros(T,Ng) |> mem` / `Array{T}(undef,…) |> mem`   (src/Flow.jl:139,143-144): mem(::Array) does the H2D copy
function HipArray(a::Array{T,N}) where {T,N}
    d = HipArray{T,N}(undef, size(a))
    chk(ccall((:wl_h2d, libwlhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}), d.ptr, a, sizeof(a), C_NULL)); d
end
Base.size(a::HipArray) = a.dims
Base.similar(a::HipArray{T}, dims::Dims{N}) where {T,N} = HipArray{T,N}(undef, dims)
Base.Array(a::HipArray{T,N}) where {T,N} = (h = Array{T,N}(undef, a.dims);
    chk(ccall((:wl_d2h, libwlhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}), h, a.ptr, sizeof(h), C_NULL)); h)
Base.copyto!(d::HipArray, s::HipArray) = (chk(ccall((:wl_d2d, libwlhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}), d.ptr, s.ptr, length(s) * 4, C_NULL)); d)
Base.copy(a::HipArray) = copyto!(similar(a, size(a)), a)
Base.fill!(a::HipArray, v) = (chk(ccall((:wl_fill, libwlhip), Cint, (Ptr{Cfloat}, Cfloat, Csize_t, Ptr{Cvoid}), a.ptr, v, length(a), C_NULL)); a)
# scalar getindex/setindex! (tests use GPUArrays.@allowscalar): one-element transfers
Base.getindex(a::HipArray{T}, i::Int) where T = (r = Ref{T}(); chk(ccall((:wl_d2h, libwlhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}), r, a.ptr + (i - 1) * sizeof(T), sizeof(T), C_NULL)); r[])
Base.setindex!(a::HipArray{T}, v, i::Int) where T = (r = Ref{T}(v); chk(ccall((:wl_h2d, libwlhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}), a.ptr + (i - 1) * sizeof(T), r, sizeof(T), C_NULL)); v)
Base.IndexStyle(::Type{<:HipArray}) = IndexLinear()
# reductions and broadcasts the path uses (src/Poisson.jl:95,189-191; src/Flow.jl:39,157,225,230,236)
Base.sum(a::HipArray) = (r = Ref{Cdouble}(); chk(ccall((:wl_sum, libwlhip), Cint, (Ptr{Cfloat}, Csize_t, Ref{Cdouble}, Ptr{Cvoid}), a.ptr, length(a), r, C_NULL)); Float32(r[]))
Base.maximum(a::HipArray) = (r = Ref{Cfloat}(); chk(ccall((:wl_max, libwlhip), Cint, (Ptr{Cfloat}, Csize_t, Ref{Cfloat}, Ptr{Cvoid}), a.ptr, length(a), r, C_NULL)); r[])
scale!(a::HipArray, s) = chk(ccall((:wl_scale, libwlhip), Cint, (Ptr{Cfloat}, Cfloat, Csize_t, Ptr{Cvoid}), a.ptr, s, length(a), C_NULL))
unscale!(a::HipArray, s) = chk(ccall((:wl_div_scalar, libwlhip), Cint, (Ptr{Cfloat}, Cfloat, Csize_t, Ptr{Cvoid}), a.ptr, s, length(a), C_NULL))

# ---- wl_grid of a single-domain array ---------------------------------------------------------------------
struct WlGrid; D::Int32; nx::Int32; ny::Int32; nz::Int32; k0::Int32; k1::Int32; gk::Int32; gnz::Int32; end
grid(dims::NTuple{2}) = WlGrid(2, dims[1], dims[2], 1, 0, 1, 0, 1)
grid(dims::NTuple{3}) = WlGrid(3, dims[1], dims[2], dims[3], 1, dims[3] - 1, 0, dims[3])
sgrid(a::HipArray) = Ref(grid(size(a)))
vgrid(a::HipArray) = Ref(grid(Base.front(size(a))))
pmask(perdir) = UInt32(sum((1 << (j - 1) for j in perdir); init=0))
scheme(λ) = λ === quick ? 0 : λ === vanLeer ? 1 : λ === cds ? 2 : error("λ must be quick, vanLeer or cds on the HIP path")

const HA = HipArray{Float32}
const HFlow = Flow{D,Float32,<:HA} where D
const HPois = Poisson{Float32,<:HA}
const HML = MultiLevelPoisson{Float32,<:HA}

# ---- core.jl ------------------------------------------------------------------------------------------------
# BC!(a,U::tuple,saveexit,perdir,t)  src/core.jl:200  (Function-valued uBC falls back to host staging — SURVEY §8b)
BC!(a::HA, U::Union{Tuple,AbstractVector}, saveexit=false, perdir=(), t=0) =
    chk(ccall((:wl_bc_vec, libwlhip), Cint, (Ptr{Cfloat}, Ref{WlGrid}, Ref{NTuple{3,Cfloat}}, Cint, Cuint, Ptr{Cvoid}),
              a.ptr, vgrid(a), ntuple(i -> i <= length(U) ? Cfloat(U[i]) : 0f0, 3), saveexit, pmask(perdir), C_NULL))
# BC!(a,uBC::Function,…) src/core.jl:201-219: the closure is evaluated on the host over the boundary shell (two layers per side, all
# the kernel reads) and handed over as a table; the device applies the reference's sequential edge/corner semantics.
function BC!(a::HA, uBC::Function, saveexit=false, perdir=(), t=0)
    N, n = WaterLily.size_u(a); T = eltype(a)
    tab = zeros(T, size(a))
    for j in 1:n, s in (1, 2, N[j]-1, N[j]), i in 1:n
        j in perdir && continue
        for I in WaterLily.slice(N, s, j); tab[I, i] = uBC(i, loc(i, I, T), t); end
    end
    Ub = HipArray(tab)
    chk(ccall((:wl_bc_vec_fn, libwlhip), Cint, (Ptr{Cfloat}, Ptr{Cfloat}, Ref{WlGrid}, Cint, Cuint, Ptr{Cvoid}), a.ptr, Ub.ptr, vgrid(a), saveexit, pmask(perdir), C_NULL))
end
# accelerate!(r,t,g,U) src/Flow.jl:69-73 for closures: tabulate g(i,x,t)+∂ₜU(i,x,t) on the host, add on the device
function WaterLily.accelerate!(r::HA, t, f::Function)
    T = eltype(r); tab = zeros(T, size(r))
    for Ii in CartesianIndices(tab); tab[Ii] = f(last(Ii), loc(Ii, T), t); end
    G = HipArray(tab)
    chk(ccall((:wl_accelerate_field, libwlhip), Cint, (Ptr{Cfloat}, Ptr{Cfloat}, Ref{WlGrid}, Ptr{Cvoid}), r.ptr, G.ptr, vgrid(r), C_NULL))
end
perBC!(a::HA, perdir::Tuple) = isempty(perdir) ? nothing :
    chk(ccall((:wl_bc_per_scalar, libwlhip), Cint, (Ptr{Cfloat}, Ref{WlGrid}, Cuint, Ptr{Cvoid}), a.ptr, sgrid(a), pmask(perdir), C_NULL))
exitBC!(u::HA, u⁰::HA, Δt) = chk(ccall((:wl_exit_bc, libwlhip), Cint, (Ptr{Cfloat}, Ptr{Cfloat}, Ref{WlGrid}, Cfloat, Ptr{Cvoid}), u.ptr, u⁰.ptr, vgrid(u), Δt, C_NULL))
L₂(a::HA) = (r = Ref{Cdouble}(); chk(ccall((:wl_L2_inside, libwlhip), Cint, (Ptr{Cfloat}, Ref{WlGrid}, Ref{Cdouble}, Ptr{Cvoid}), a.ptr, sgrid(a), r, C_NULL)); r[])   # ext/WaterLilyAMDGPUExt.jl:18

# ---- Flow.jl -------------------------------------------------------------------------------------------------
conv_diff!(r::HA, u::HA, Φ::HA, λ::F; ν=0.1, perdir=()) where {F} =
    chk(ccall((:wl_conv_diff, libwlhip), Cint, (Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ref{WlGrid}, Cfloat, Cuint, Cint, Ptr{Cvoid}),
              r.ptr, u.ptr, Φ.ptr, vgrid(u), ν, pmask(perdir), scheme(λ), C_NULL))
BDIM!(a::HFlow) = chk(ccall((:wl_bdim, libwlhip), Cint, (Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ref{WlGrid}, Cfloat, Cfloat, Cfloat, Ptr{Cvoid}),
                          a.u.ptr, a.u⁰.ptr, a.f.ptr, a.V.ptr, a.μ₀.ptr, a.μ₁.ptr, vgrid(a.u), a.Δt[end], 1f0, 1f0, C_NULL))
scale_u!(a::HFlow, s) = chk(ccall((:wl_scale_u, libwlhip), Cint, (Ptr{Cfloat}, Ref{WlGrid}, Cfloat, Ptr{Cvoid}), a.u.ptr, vgrid(a.u), s, C_NULL))
function CFL(a::HFlow; Δt_max=10)
    r = Ref{Cfloat}()
    chk(ccall((:wl_cfl, libwlhip), Cint, (Ptr{Cfloat}, Ptr{Cfloat}, Ref{WlGrid}, Cfloat, Cfloat, Ref{Cfloat}, Ptr{Cvoid}), a.u.ptr, a.σ.ptr, sgrid(a.σ), a.ν, Δt_max, r, C_NULL)); r[]
end
function mom_project!(a::HFlow, b::AbstractPoisson, w, t)        # src/Flow.jl:223-232 on device arrays
    dt = Float32(w) * a.Δt[end]
    chk(ccall((:wl_div, libwlhip), Cint, (Ptr{Cfloat}, Ptr{Cfloat}, Ref{WlGrid}, Ptr{Cvoid}), b.z.ptr, a.u.ptr, sgrid(b.z), C_NULL)); scale!(b.x, dt)
    solver!(b)
    chk(ccall((:wl_project, libwlhip), Cint, (Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ref{WlGrid}, Ptr{Cvoid}), a.u.ptr, b.L.ptr, b.x.ptr, sgrid(b.x), C_NULL))
    unscale!(b.x, dt); BC!(a.u, a.uBC, a.exitBC, a.perdir, t)
end
# mom_step!, mom_predict!, mom_correct! of the reference run UNCHANGED on top of these methods (u⁰ .= u → copyto!).

# ---- Poisson.jl / MultiLevelPoisson.jl --------------------------------------------------------------------------
set_diag!(D::HA, iD::HA, L::HA) = chk(ccall((:wl_set_diag, libwlhip), Cint, (Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ref{WlGrid}, Ptr{Cvoid}), D.ptr, iD.ptr, L.ptr, sgrid(D), C_NULL))
mult!(p::HPois, x::HA) = (perBC!(x, p.perdir); chk(ccall((:wl_mult, libwlhip), Cint, (Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ref{WlGrid}, Ptr{Cvoid}), p.z.ptr, p.L.ptr, p.D.ptr, x.ptr, sgrid(x), C_NULL)); p.z)
residual!(p::HPois) = (perBC!(p.x, p.perdir); chk(ccall((:wl_residual, libwlhip), Cint, (Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ref{WlGrid}, Ptr{Cvoid}, Ptr{Cvoid}),
                                                      p.r.ptr, p.x.ptr, p.z.ptr, p.L.ptr, p.D.ptr, p.iD.ptr, sgrid(p.x), C_NULL, C_NULL)))
increment!(p::HPois; ω=1) = (perBC!(p.ϵ, p.perdir); chk(ccall((:wl_increment, libwlhip), Cint, (Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ref{WlGrid}, Cfloat, Ptr{Cvoid}),
                                                             p.r.ptr, p.x.ptr, p.ϵ.ptr, p.L.ptr, p.D.ptr, sgrid(p.x), ω, C_NULL)))
Jacobi!(p::HPois; it=1, ω=1) = chk(ccall((:wl_jacobi, libwlhip), Cint, (Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ref{WlGrid}, Cint, Cfloat, Cuint, Ptr{Cvoid}),
                                         p.ϵ.ptr, p.r.ptr, p.x.ptr, p.L.ptr, p.D.ptr, p.iD.ptr, sgrid(p.x), it, ω, pmask(p.perdir), C_NULL))
GaussSeidelRB!(p::HPois; it=4, ω=1) = chk(ccall((:wl_gsrb, libwlhip), Cint, (Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ref{WlGrid}, Cint, Cfloat, Cuint, Ptr{Cvoid}),
                                                p.ϵ.ptr, p.r.ptr, p.x.ptr, p.L.ptr, p.D.ptr, p.iD.ptr, sgrid(p.x), it, ω, pmask(p.perdir), C_NULL))
pcg!(p::HPois; it=6, kwargs...) = chk(ccall((:wl_pcg, libwlhip), Cint, (Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ref{WlGrid}, Cint, Cuint, Ptr{Cvoid}),
                                           p.ϵ.ptr, p.r.ptr, p.x.ptr, p.z.ptr, p.L.ptr, p.D.ptr, p.iD.ptr, sgrid(p.x), it, pmask(p.perdir), C_NULL))   # src/Poisson.jl:166
function solver!(p::HPois; tol=2e-3, itmx=1e3)                                                                                            # src/Poisson.jl:212
    n = Ref{Cint}()
    chk(ccall((:wl_poisson_solve, libwlhip), Cint, (Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ref{WlGrid}, Cdouble, Cint, Cuint, Ref{Cint}, Ptr{Cdouble}, Ptr{Cfloat}, Ptr{Cvoid}),
              p.ϵ.ptr, p.r.ptr, p.x.ptr, p.z.ptr, p.L.ptr, p.D.ptr, p.iD.ptr, sgrid(p.x), tol, Int32(min(itmx, typemax(Int32))), pmask(p.perdir), n, C_NULL, C_NULL, C_NULL))
    push!(p.n, n[])
end
function norms(p::HPois)
    l1 = Ref{Cdouble}(); li = Ref{Cfloat}()
    chk(ccall((:wl_norms, libwlhip), Cint, (Ptr{Cfloat}, Ref{WlGrid}, Ref{Cdouble}, Ref{Cfloat}, Ptr{Cvoid}, Ptr{Cvoid}), p.r.ptr, sgrid(p.r), l1, li, C_NULL, C_NULL)); (Float32(l1[]), li[])
end
L₁(p::HPois) = norms(p)[1]
L∞(p::HPois) = norms(p)[2]
restrict!(a::HA, b::HA, c) = chk(ccall((:wl_restrict, libwlhip), Cint, (Ptr{Cfloat}, Ref{WlGrid}, Ptr{Cfloat}, Ref{WlGrid}, Ptr{Cvoid}), a.ptr, sgrid(a), b.ptr, sgrid(b), C_NULL))
prolongate!(a::HA, b::HA, c) = chk(ccall((:wl_prolongate, libwlhip), Cint, (Ptr{Cfloat}, Ref{WlGrid}, Ptr{Cfloat}, Ref{WlGrid}, Ptr{Cvoid}), a.ptr, sgrid(a), b.ptr, sgrid(b), C_NULL))
restrictL!(a::HA, b::HA, c; perdir=()) = chk(ccall((:wl_restrictL, libwlhip), Cint, (Ptr{Cfloat}, Ref{WlGrid}, Ptr{Cfloat}, Ref{WlGrid}, Cuint, Ptr{Cvoid}), a.ptr, vgrid(a), b.ptr, vgrid(b), pmask(perdir), C_NULL))
# With the methods above the reference's own Vcycle!/solver!/update!/mom_step! drive the device unchanged.
#
# Fast path (what bench.py times): hand the whole solve / step to the library's composites, which fuse launches,
# keep the convergence scalars on device and need one host read per V-cycle.  `pois_ctor` is the official hook
# (src/WaterLily.jl:69-74,96-97):   Simulation(...; mem=HipArray, pois_ctor = flow -> HipMultiLevel(flow))
mutable struct HipMultiLevel <: AbstractPoisson{Float32,HA,HA}
    x::HA; L::HA; z::HA; n::Vector{Int16}; perdir::NTuple; handle::Ptr{Cvoid}
end
function HipMultiLevel(flow; perdir=flow.perdir, maxlevels=10)
    h = Ref{Ptr{Cvoid}}()
    chk(ccall((:wl_mg_create, libwlhip), Cint, (Ref{Ptr{Cvoid}}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ref{WlGrid}, Cuint, Cint), h, flow.p.ptr, flow.μ₀.ptr, flow.σ.ptr, sgrid(flow.p), pmask(perdir), maxlevels))
    m = HipMultiLevel(flow.p, flow.μ₀, flow.σ, Int16[], perdir, h[])
    finalizer(x -> ccall((:wl_mg_destroy, libwlhip), Cint, (Ptr{Cvoid},), x.handle), m)
end
update!(m::HipMultiLevel) = chk(ccall((:wl_mg_update, libwlhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), m.handle, C_NULL))
function solver!(m::HipMultiLevel; tol=2e-3, itmx=32)
    n = Ref{Cint}(); r1 = Ref{Cdouble}(); ri = Ref{Cfloat}()
    chk(ccall((:wl_mg_solve, libwlhip), Cint, (Ptr{Cvoid}, Cdouble, Cint, Ref{Cint}, Ref{Cdouble}, Ref{Cfloat}, Ptr{Cvoid}), m.handle, tol, itmx, n, r1, ri, C_NULL))
    push!(m.n, n[])
end

# temporal averages (src/Metrics.jl:236-252) on device arrays
function WaterLily.update!(m::WaterLily.MeanFlow{Float32,<:HA}, flow::WaterLily.AbstractFlow)
    dt = WaterLily.time(flow) - m.t[end]
    ε = length(m.t) == 1 ? 1f0 : dt / (dt + WaterLily.time(m) + eps(Float32))
    chk(ccall((:wl_meanflow_update, libwlhip), Cint, (Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ref{WlGrid}, Cfloat, Ptr{Cvoid}),
              m.P.ptr, m.U.ptr, m.uu_stats ? m.UU.ptr : C_NULL, flow.p.ptr, flow.u.ptr, sgrid(flow.p), ε, C_NULL))
    push!(m.t, m.t[end] + dt)
end

# ---- closed-form bodies (SURVEY row f1) --------------------------------------------------------------------------
# An AutoBody holds arbitrary Julia closures, which cannot be shipped to a HIP kernel through a C ABI.  The shapes whose
# sdf/gradient are known in closed form get their own AbstractBody type; everything else keeps the reference's host path
# (measure! on Array fields, then copyto! the HipArrays).
struct WlBody; kind::Int32; c::NTuple{3,Cfloat}; R::Cfloat; m::NTuple{3,Cfloat}; V::NTuple{3,Cfloat}; end      # include/wlhip.h wl_body
pad3(v) = ntuple(i -> i <= length(v) ? Cfloat(v[i]) : 0f0, 3)
"""
    HipBody(:sphere, c, R; V) | HipBody(:cylinder, c, R, axis; V) | HipBody(:plane, point, normal; V)

sdf = |m∘(x−c)|−R (an axis with m=0 is dropped) or m·(x−c).  `c(t)`/`V(t)` may be functions of time: the translating map
`x − ∫V dt` of the reference's `AutoBody(sdf, map)` (src/AutoBody.jl:36-37).
"""
struct HipBody{C,VV} <: WaterLily.AbstractBody
    kind::Int32; c::C; R::Float32; m::NTuple{3,Cfloat}; V::VV
end
HipBody(s::Symbol, c, a...; V=(0, 0, 0)) =
    s === :sphere   ? HipBody(Int32(1), c, Float32(a[1]), (1f0, 1f0, 1f0), V) :
    s === :cylinder ? HipBody(Int32(1), c, Float32(a[1]), ntuple(i -> i == a[2] ? 0f0 : 1f0, 3), V) :
    s === :plane    ? HipBody(Int32(2), c, 0f0, pad3(a[1]), V) : error("HipBody: :sphere, :cylinder or :plane")
at(v::Function, t) = v(t); at(v, t) = v
wlbody(b::HipBody, D, t) = Ref(WlBody(b.kind, pad3(at(b.c, t)), b.R, ntuple(i -> i <= D ? b.m[i] : 0f0, 3), pad3(at(b.V, t))))
# measure!(a::Flow,body;t,ϵ)  src/Body.jl:28-51
function WaterLily.measure!(a::HFlow{D}, body::HipBody; t=zero(Float32), ϵ=1) where D
    chk(ccall((:wl_measure_body, libwlhip), Cint, (Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ref{WlGrid}, Ref{WlBody}, Cfloat, Cint, Cuint, Ptr{Cvoid}),
              a.σ.ptr, a.μ₀.ptr, a.μ₁.ptr, a.V.ptr, sgrid(a.σ), wlbody(body, D, t), ϵ, a.exitBC, pmask(a.perdir), C_NULL))
end
# pressure_force / viscous_force (src/Metrics.jl:116-133,140-154): Float64 sums on device, flow.f is not used as scratch
function WaterLily.pressure_force(p::HA, df, body::HipBody, t=0; T=Float64)          # src/Metrics.jl:128
    out = zeros(Cdouble, 3); D = ndims(p)
    chk(ccall((:wl_pressure_force_body, libwlhip), Cint, (Ptr{Cfloat}, Ref{WlGrid}, Ref{WlBody}, Ptr{Cdouble}, Ptr{Cvoid}), p.ptr, sgrid(p), wlbody(body, D, t), out, C_NULL))
    T.(out[1:D])
end
function WaterLily.viscous_force(u::HA, ν, df, body::HipBody, t=0; T=Float64)        # src/Metrics.jl:149
    out = zeros(Cdouble, 3); D = ndims(u) - 1
    chk(ccall((:wl_viscous_force_body, libwlhip), Cint, (Ptr{Cfloat}, Ref{WlGrid}, Cfloat, Ref{WlBody}, Ptr{Cdouble}, Ptr{Cvoid}), u.ptr, vgrid(u), ν, wlbody(body, D, t), out, C_NULL))
    T.(out[1:D])
end

function WaterLily.pressure_moment(x₀, p::HA, df, body::HipBody, t=0)                                       # src/Metrics.jl:169
    out = zeros(Cdouble, 3); D = ndims(p)
    chk(ccall((:wl_pressure_moment_body, libwlhip), Cint, (Ref{NTuple{3,Cfloat}}, Ptr{Cfloat}, Ref{WlGrid}, Ref{WlBody}, Ptr{Cdouble}, Ptr{Cvoid}), Ref(pad3(x₀)), p.ptr, sgrid(p), wlbody(body, D, t), out, C_NULL))
    out[1:D]
end
function WaterLily.viscous_moment(x₀, u::HA, ν, df, body::HipBody, t=0)                                     # src/Metrics.jl:183
    out = zeros(Cdouble, 3); D = ndims(u) - 1
    chk(ccall((:wl_viscous_moment_body, libwlhip), Cint, (Ref{NTuple{3,Cfloat}}, Ptr{Cfloat}, Ref{WlGrid}, Cfloat, Ref{WlBody}, Ptr{Cdouble}, Ptr{Cvoid}), Ref(pad3(x₀)), u.ptr, vgrid(u), ν, wlbody(body, D, t), out, C_NULL))
    out[1:D]
end

export HipArray, HipMultiLevel, HipBody
end # module
