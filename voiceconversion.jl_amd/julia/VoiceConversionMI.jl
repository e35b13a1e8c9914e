# VoiceConversionMI.jl -- drop-in Julia host side for the MI355X hot path of VoiceConversion.jl.
#
# Thin `ccall` wrapper over libvcmi.so (include/vcmi.h).  It keeps the reference's exported names, signatures and
# the struct fields its own code reads (reference src/VoiceConversion.jl:12-38, src/dtw.jl:7):
#   GMMMapParam, GMMMap (fields params, px), fvconvert, vc, dim, ncomponents, predict_proba(g.px, X),
#   predict(g.px, X), TrajectoryGMMMap, TrajectoryGVGMMMap, DTW, fit!, update!, set_template!, backward, align,
#   align_mcep, push_delta, VarianceScaling, fvpostf, fvpostf!, diffgmm(params)
# and adds the batch methods the reference lacks (fvconvert(g, X::Matrix), fit!(d, templates, sequences),
# estep_diag, estep_full, GMMEM) plus set_devices (single host process driving several GPUs).
#
# NOTE: no Julia binary exists in the build image or on the GPU box, so this file has never been executed.  What can
# be checked without Julia is checked by tests/test_julia_binding_lint.py: every `ccall` (symbol, return type, arity
# and each argument type) against include/vcmi.h, and every `obj.field` access against the struct declarations.
# The identical ABI is exercised from Python ctypes (voiceconversion.jl_amd/_lib.py) by the parity tests.
# Written for Julia >= 1.0; for the reference's Julia 0.5 replace `mutable struct` by `type`, `struct` by
# `immutable`, `finalizer(f, obj)` by `finalizer(obj, f)`, `Cvoid` by `Void`, `undef` constructors by `Array(T, dims)`.
module VoiceConversionMI

import LinearAlgebra

export DeviceMatrix, FrameByFrameConverter, TrajectoryConverter, GMMMapParam, GMMMap, TrajectoryGMMMap, TrajectoryGVGMMMap,
       fvconvert, vc, ncomponents, dim,
       VarianceScaling, fvpostf!, fvpostf,
       align, align_mcep, push_delta, GVDataset,
       DTW, fit!, update!, set_template!, backward,
       predict_proba, predict_proba!, predict, predict!, diffgmm,
       estep_diag, estep_full, estep_set_path, estep_get_path, ESTEP_AUTO, ESTEP_HARD, ESTEP_SOFT, GMMEM, estep!, mstep!, params, set_devices, device_count, set_prune!, convert_plan, pin!, unpin!, ispinned

const libvcmi = get(ENV, "LIBVCMI", "libvcmi")

# vcmi_status -> the exception the reference would have thrown
function check(status::Cint)
    status == 0 && return
    msg = unsafe_string(ccall((:vcmi_last_error, libvcmi), Cstring, ()))
    status == 1 && throw(DimensionMismatch(msg))                       # src/gmmmap.jl:102
    status == 2 && throw(LinearAlgebra.PosDefException(0))             # MvNormal in src/gmm.jl:17
    error("libvcmi status $status: $msg")
end

function device_count()
    n = Ref{Cint}(0)
    check(ccall((:vcmi_device_count, libvcmi), Cint, (Ref{Cint},), n))
    Int(n[])
end

# One host process, several GPUs: after set_devices(0:7) the host-pointer entry points (fvconvert / vc on matrices,
# fit! / align on vectors of pairs, trajectory batches, estep_*) shard their frames / pairs / utterances over these
# devices; only the E-step exchanges data (one RCCL all-reduce of the statistics).  set_devices(Int[]) restores the
# single-device behaviour.
function set_devices(devices)
    d = Cint[Cint(x) for x in devices]
    check(ccall((:vcmi_set_devices, libvcmi), Cint, (Ptr{Cint}, Cint), d, length(d)))
end

# Page-lock an array this process keeps across calls (vcmi_host_register): fvconvert / vc / predict_proba on it then DMA
# straight from / into it -- no staging copy (include/vcmi.h).  pin!(X) returns X and unpins it when X is finalized;
# unpin!(X) does it now.  Results are identical with and without.
# The table keeps the REGISTERED address and length (not the arrays: it must not keep them alive) behind a lock -- finalizers
# run at arbitrary points, possibly on another thread.  A pinned array must not be resized (resize! / append! move its data:
# the old range would stay page-locked; unpin! it first).
const PINNED = Dict{UInt,Tuple{Ptr{Cvoid},Csize_t}}()     # objectid(X) -> (registered pointer, bytes)
const PINNED_LOCK = ReentrantLock()
function pin!(X::Array{Float64})
    lock(PINNED_LOCK) do
        haskey(PINNED, objectid(X)) && return X
        p = Ptr{Cvoid}(pointer(X))
        check(ccall((:vcmi_host_register, libvcmi), Cint, (Ptr{Cvoid}, Csize_t), p, sizeof(X)))
        PINNED[objectid(X)] = (p, Csize_t(sizeof(X)))
        finalizer(unpin_quietly, X)
        X
    end
end
# unregisters the pointer that WAS registered (pointer(X) may have moved); returns the library's status
function unpin_status(X::Array{Float64})
    lock(PINNED_LOCK) do
        e = pop!(PINNED, objectid(X), nothing)
        e === nothing && return Cint(0)
        ccall((:vcmi_host_unregister, libvcmi), Cint, (Ptr{Cvoid},), e[1])
    end
end
unpin_quietly(X::Array{Float64}) = (unpin_status(X); nothing)        # finalizer: never throws
function unpin!(X::Array{Float64})
    check(unpin_status(X))
    X
end
function ispinned(X::Array{Float64})
    f = Ref{Cint}(0)
    check(ccall((:vcmi_host_is_registered, libvcmi), Cint, (Ptr{Cvoid}, Csize_t, Ptr{Cint}), X, sizeof(X), f))
    f[] != 0
end

abstract type AbstractConverter end                                    # src/common.jl:2-4
abstract type FrameByFrameConverter <: AbstractConverter end
abstract type TrajectoryConverter <: AbstractConverter end

# ------------------------------------------------------------------------------------------ GMMMapParam
struct GMMMapParam                                                     # src/gmmmap.jl:10-39
    weights::Vector{Float64}
    μˣ::Matrix{Float64}
    μʸ::Matrix{Float64}
    Σˣˣ::Array{Float64,3}
    Σˣʸ::Array{Float64,3}
    Σʸˣ::Array{Float64,3}
    Σʸʸ::Array{Float64,3}
    ΣʸˣΣˣˣ⁻¹::Array{Float64,3}
end

function GMMMapParam(weights::Vector{Float64}, μˣ::Matrix{Float64}, μʸ::Matrix{Float64}, Σˣˣ::Array{Float64,3},
                     Σˣʸ::Array{Float64,3}, Σʸˣ::Array{Float64,3}, Σʸʸ::Array{Float64,3})
    M = length(weights)
    D = size(μˣ, 1)
    A = Array{Float64,3}(undef, D, D, M)
    for m = 1:M                                                        # src/gmmmap.jl:33-36 (one-time set-up)
        A[:, :, m] = Σʸˣ[:, :, m] * inv(Σˣˣ[:, :, m])
    end
    GMMMapParam(weights, μˣ, μʸ, Σˣˣ, Σˣʸ, Σʸˣ, Σʸʸ, A)
end

function split_joint_gmm(μ::Matrix{Float64}, Σ::Array{Float64,3})     # src/gmmmap.jl:41-52
    D = size(μ, 1) >> 1
    μ[1:D, :], μ[D+1:end, :], Σ[1:D, 1:D, :], Σ[1:D, D+1:end, :], Σ[D+1:end, 1:D, :], Σ[D+1:end, D+1:end, :]
end

# joint (μ, Σ) image of a parameter set: what vcmi_gmmmap_create takes
function joint_gmm(p::GMMMapParam)
    μ = vcat(p.μˣ, p.μʸ)
    D, M = size(p.μˣ)
    Σ = Array{Float64,3}(undef, 2D, 2D, M)
    Σ[1:D, 1:D, :] = p.Σˣˣ
    Σ[1:D, D+1:end, :] = p.Σˣʸ
    Σ[D+1:end, 1:D, :] = p.Σʸˣ
    Σ[D+1:end, D+1:end, :] = p.Σʸʸ
    μ, Σ
end

# ------------------------------------------------------------------------------------------ GMM (g.px)
# In the reference g.px is a Distributions.MixtureModel (src/gmm.jl:5-20); here it is a view onto the converter's
# device-resident whitening blocks.  `owner` keeps the handle alive.
mutable struct GMM
    h::Ptr{Cvoid}
    D::Int
    M::Int
    owner::Any
end

# ---------------------------------------------------------------------------------------------- GMMMap
mutable struct GMMMap <: FrameByFrameConverter                        # src/gmmmap.jl:57-60
    params::GMMMapParam
    px::GMM
    h::Ptr{Cvoid}
end

function gmmmap_handle(weights::Vector{Float64}, μ::Matrix{Float64}, Σ::Array{Float64,3}, swap::Bool)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:vcmi_gmmmap_create, libvcmi), Cint,
                (Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint, Cint, Cint, Ref{Ptr{Cvoid}}),
                weights, μ, Σ, size(μ, 1), length(weights), swap ? 1 : 0, h))
    h[]
end

function GMMMap(weights::Vector{Float64}, μ::Matrix{Float64}, Σ::Array{Float64,3}; swap::Bool=false)   # src/gmmmap.jl:62-90
    size(Σ) == (size(μ, 1), size(μ, 1), length(weights)) || throw(DimensionMismatch("weights, μ, Σ are inconsistent"))
    μˣ, μʸ, Σˣˣ, Σˣʸ, Σʸˣ, Σʸʸ = split_joint_gmm(μ, Σ)
    if swap                                                            # src/gmmmap.jl:74-78
        μˣ, μʸ = μʸ, μˣ
        Σˣˣ, Σʸʸ = Σʸʸ, Σˣˣ
        Σˣʸ, Σʸˣ = Σʸˣ, Σˣʸ
    end
    p = GMMMapParam(weights, μˣ, μʸ, Σˣˣ, Σˣʸ, Σʸˣ, Σʸʸ)
    h = gmmmap_handle(weights, μ, Σ, swap)
    g = GMMMap(p, GMM(h, size(μˣ, 1), length(weights), nothing), h)
    g.px.owner = g
    finalizer(x -> ccall((:vcmi_gmmmap_destroy, libvcmi), Cint, (Ptr{Cvoid},), x.h), g)
    g
end

# converter over an explicit parameter set, e.g. GMMMap(diffgmm(g.params))
function GMMMap(p::GMMMapParam)
    μ, Σ = joint_gmm(p)
    h = gmmmap_handle(p.weights, μ, Σ, false)
    g = GMMMap(p, GMM(h, size(p.μˣ, 1), length(p.weights), nothing), h)
    g.px.owner = g
    finalizer(x -> ccall((:vcmi_gmmmap_destroy, libvcmi), Cint, (Ptr{Cvoid},), x.h), g)
    g
end

Base.length(g::GMMMap) = 1                                             # src/gmmmap.jl:93
dim(g::GMMMap) = Int(ccall((:vcmi_gmmmap_dim, libvcmi), Cint, (Ptr{Cvoid},), g.h))
ncomponents(g::GMMMap) = Int(ccall((:vcmi_gmmmap_ncomponents, libvcmi), Cint, (Ptr{Cvoid},), g.h))
Base.size(g::GMMMap) = (dim(g), length(g))
# not in the reference: posterior pruning threshold of fvconvert in nats (default 46.0; Inf = evaluate every mixture)
set_prune!(g::GMMMap, nats::Real) = check(ccall((:vcmi_gmmmap_set_prune, libvcmi), Cint, (Ptr{Cvoid}, Cdouble), g.h, Float64(nats)))
# not in the reference: which loop the library runs for this model (0 dense, 1 "broad", 2 "peaked", -1 no tile kernel) and the
# two model properties the choice rests on (measurement; include/vcmi.h: vcmi_gmmmap_convert_plan)
function convert_plan(g::GMMMap)
    issued = Ref{Int64}(0); shape = Ref{Cint}(0); active = Ref{Float64}(0.0); undecided = Ref{Float64}(0.0)
    check(ccall((:vcmi_gmmmap_convert_plan, libvcmi), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Cint}, Ptr{Float64}, Ptr{Float64}),
                g.h, issued, shape, active, undecided))
    # the C argument order, which is also what Python's GMMMap.convert_plan returns; named, so that no caller indexes by position
    (issued = issued[], shape = Int(shape[]), active = active[], undecided = undecided[])
end

# fvconvert(g, x) -- src/gmmmap.jl:101-118
function fvconvert(g::GMMMap, x::Vector{Float64})
    length(x) == dim(g) || throw(DimensionMismatch("Inconsistent dimentions."))
    y = Vector{Float64}(undef, dim(g))
    check(ccall((:vcmi_gmmmap_convert, libvcmi), Cint,
                (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64), g.h, x, dim(g), 1, y, dim(g)))
    y
end

# batch method: every column of X in one launch
function fvconvert(g::GMMMap, X::Matrix{Float64})
    size(X, 1) == dim(g) || throw(DimensionMismatch("Inconsistent dimentions."))
    Y = similar(X)
    check(ccall((:vcmi_gmmmap_convert, libvcmi), Cint,
                (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64),
                g.h, X, size(X, 1), size(X, 2), Y, size(Y, 1)))
    Y
end

# vc(c::FrameByFrameConverter, fm) -- src/common.jl:7-26 (row 1 = power, kept)
function vc(g::GMMMap, fm::Matrix{Float64})
    size(fm, 1) == dim(g) + 1 || throw(DimensionMismatch("Inconsistent dimentions."))
    out = similar(fm)
    check(ccall((:vcmi_vc_frames, libvcmi), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}),
                g.h, fm, size(fm, 2), out))
    out
end

# predict_proba / predict -- src/gmm.jl:24-58, called as predict_proba(g.px, x) (src/gmmmap.jl:114) and
# predict(g.px, X) (src/trajectory_gmmmap.jl:82)
function predict_proba!(r::Matrix{Float64}, gmm::GMM, X::Matrix{Float64})          # src/gmm.jl:32-37
    size(X, 1) == gmm.D || throw(DimensionMismatch("Inconsistent dimentions."))
    size(r) == (gmm.M, size(X, 2)) || throw(DimensionMismatch("posterior matrix must be (M,T)"))
    check(ccall((:vcmi_gmmmap_posterior, libvcmi), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Float64}),
                gmm.h, X, size(X, 1), size(X, 2), r))
    r
end
predict_proba(gmm::GMM, X::Matrix{Float64}) = predict_proba!(Matrix{Float64}(undef, gmm.M, size(X, 2)), gmm, X)   # :39-41
predict_proba(gmm::GMM, x::Vector{Float64}) = vec(predict_proba(gmm, reshape(x, length(x), 1)))                    # :24-30

function predict!(r::Vector{Int}, gmm::GMM, X::Matrix{Float64})                    # src/gmm.jl:49-54
    size(X, 1) == gmm.D || throw(DimensionMismatch("Inconsistent dimentions."))
    length(r) == size(X, 2) || throw(DimensionMismatch("label vector must have one entry per frame"))
    check(ccall((:vcmi_gmmmap_predict, libvcmi), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Int64}),
                gmm.h, X, size(X, 1), size(X, 2), r))
    r
end
predict(gmm::GMM, X::Matrix{Float64}) = predict!(Vector{Int}(undef, size(X, 2)), gmm, X)                           # :56-58
predict(gmm::GMM, x::Vector{Float64}) = predict(gmm, reshape(x, length(x), 1))[1]                                  # :44-47

# diffgmm(params) -- src/diffgmm.jl:9-25 (the reference's `(Σˣʸ - Σˣˣ)'` is applied per mixture)
function diffgmm(p::GMMMapParam)
    μ, Σ = joint_gmm(p)
    μd = similar(μ); Σd = similar(Σ)
    check(ccall((:vcmi_diffgmm, libvcmi), Cint, (Ptr{Float64}, Ptr{Float64}, Cint, Cint, Ptr{Float64}, Ptr{Float64}),
                μ, Σ, size(μ, 1), size(μ, 2), μd, Σd))
    μˣ, μʸ, Σˣˣ, Σˣʸ, Σʸˣ, Σʸʸ = split_joint_gmm(μd, Σd)
    GMMMapParam(p.weights, μˣ, μʸ, Σˣˣ, Σˣʸ, Σʸˣ, Σʸʸ)
end

# ---------------------------------------------------------------------------------- TrajectoryGMMMap
mutable struct TrajectoryGMMMap <: TrajectoryConverter                 # src/trajectory_gmmmap.jl:3
    gmmmap::GMMMap
    h::Ptr{Cvoid}
    function TrajectoryGMMMap(g::GMMMap, T::Int)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:vcmi_traj_create, libvcmi), Cint, (Ptr{Cvoid}, Int64, Ref{Ptr{Cvoid}}), g.h, T, h))
        t = new(g, h[])
        finalizer(x -> ccall((:vcmi_traj_destroy, libvcmi), Cint, (Ptr{Cvoid},), x.h), t)
        t
    end
end
# length(t) follows the reference: fvconvert rebuilds W for a new T (src/trajectory_gmmmap.jl:70-72), so the length
# is that of the last converted (sub-)sequence; the library tracks it.
Base.length(t::TrajectoryGMMMap) = Int(ccall((:vcmi_traj_length, libvcmi), Int64, (Ptr{Cvoid},), t.h))
dim(t::TrajectoryGMMMap) = dim(t.gmmmap)
ncomponents(t::TrajectoryGMMMap) = ncomponents(t.gmmmap)
Base.size(t::TrajectoryGMMMap) = (dim(t), length(t))

# fvconvert(tgmm, X) -- src/trajectory_gmmmap.jl:65-110
function fvconvert(t::TrajectoryGMMMap, X::Matrix{Float64})
    size(X, 1) == dim(t) || throw(DimensionMismatch("Inconsistent dimentions."))
    Y = Matrix{Float64}(undef, size(X, 1) >> 1, size(X, 2))
    check(ccall((:vcmi_traj_convert, libvcmi), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}),
                t.h, X, size(X, 2), Y))
    Y
end

# batch method: independent utterances in one launch
function fvconvert(t::TrajectoryGMMMap, Xs::Vector{Matrix{Float64}})
    n = length(Xs)
    T = Int64[size(x, 2) for x in Xs]
    Ys = [Matrix{Float64}(undef, dim(t) >> 1, k) for k in T]
    GC.@preserve Xs Ys begin
        check(ccall((:vcmi_traj_convert_batch, libvcmi), Cint,
                    (Ptr{Cvoid}, Int64, Ptr{Ptr{Float64}}, Ptr{Int64}, Ptr{Ptr{Float64}}),
                    t.h, n, pointer.(Xs), T, pointer.(Ys)))
    end
    Ys
end

# vc(c::TrajectoryConverter, fm) -- src/common.jl:31-63: all chunks of length(c) frames in one launch
function vc(t::TrajectoryGMMMap, fm::Matrix{Float64})
    size(fm, 1) == dim(t) + 1 || throw(DimensionMismatch("Inconsistent dimentions."))
    out = Matrix{Float64}(undef, (size(fm, 1) - 1) >> 1 + 1, size(fm, 2))
    check(ccall((:vcmi_vc_traj, libvcmi), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}),
                t.h, fm, size(fm, 2), out))
    out
end

function push_delta(src::Matrix{Float64})                              # src/datasets.jl:6-13
    out = Matrix{Float64}(undef, 2size(src, 1), size(src, 2))
    check(ccall((:vcmi_push_delta, libvcmi), Cint, (Ptr{Float64}, Cint, Int64, Ptr{Float64}),
                src, size(src, 1), size(src, 2), out))
    out
end

# ------------------------------------------------------------------------------------------------ DTW
mutable struct DTW                                                     # src/dtw.jl:11-17
    fstep::Int
    bstep::Int
    template::Matrix{Float64}
    costtable::Matrix{Float64}
    backpointer::Matrix{Int}
end
DTW(; fstep=0, bstep=1) = DTW(fstep, bstep, zeros(1, 1), zeros(1, 1), zeros(Int, 1, 1))   # src/dtw.jl:19-21

# fit!(d, template, sequence) -- src/dtw.jl:93-128 (+ backward :133-145); fills d.costtable / d.backpointer
function fit!(d::DTW, template::Matrix{Float64}, sequence::Matrix{Float64})
    S, T = size(template, 2), size(sequence, 2)
    size(template, 1) == size(sequence, 1) || throw(DimensionMismatch("feature dimension"))
    d.template = template
    d.costtable = Matrix{Float64}(undef, S, T + 1)
    d.backpointer = Matrix{Int}(undef, S, T + 1)
    path = Vector{Int}(undef, T)
    check(ccall((:vcmi_dtw_fit, libvcmi), Cint,
                (Ptr{Float64}, Int64, Ptr{Float64}, Int64, Cint, Cint, Cint, Ptr{Int64}, Ptr{Float64}, Ptr{Int64}),
                template, S, sequence, T, size(template, 1), d.fstep, d.bstep, path, d.costtable, d.backpointer))
    path
end
fit!(d::DTW, sequence::Matrix{Float64}) = fit!(d, d.template, sequence)               # src/dtw.jl:130

# batch method: n pairs, one workgroup each, path-only
function fit!(d::DTW, templates::Vector{Matrix{Float64}}, sequences::Vector{Matrix{Float64}})
    n = length(templates)
    S = Int64[size(t, 2) for t in templates]; T = Int64[size(s, 2) for s in sequences]
    paths = [Vector{Int}(undef, t) for t in T]
    GC.@preserve templates sequences paths begin
        check(ccall((:vcmi_dtw_fit_batch, libvcmi), Cint,
                    (Int64, Ptr{Ptr{Float64}}, Ptr{Int64}, Ptr{Ptr{Float64}}, Ptr{Int64}, Cint, Cint, Cint, Ptr{Ptr{Int64}}),
                    n, pointer.(templates), S, pointer.(sequences), T, size(templates[1], 1), d.fstep, d.bstep,
                    pointer.(paths)))
    end
    paths
end

function set_template!(d::DTW, template::Matrix{Float64})              # src/dtw.jl:38-42,53-56
    d.template = template
    S = size(template, 2)
    d.costtable = reshape(collect(1.0:S), S, 1)
    d.backpointer = reshape(collect(1:S), S, 1)
end

# update!(d, v): one on-line column -- src/dtw.jl:61-90; stays on the host (one S-cell column is not GPU work)
function update!(d::DTW, v::AbstractVector)
    S, T = size(d.costtable)
    last = d.costtable[:, T]
    cur = zeros(S); curbp = zeros(Int, S)
    for i = 1:S
        obs = 0.0
        for k = 1:length(v)
            obs += (v[k] - d.template[k, i])^2
        end
        minindex = i
        mincost = last[i] + obs + 1.0
        for j = i-d.bstep:i+d.fstep
            (j < 1 || j > S) && continue
            c = last[j] + obs + (i == j + 1 ? 0.0 : (i == j ? 1.0 : 2.0))
            if c < mincost
                mincost, minindex = c, j
            end
        end
        cur[i] = mincost; curbp[i] = minindex
    end
    d.costtable = [d.costtable cur]
    d.backpointer = [d.backpointer curbp]
end

function backward(d::DTW)                                              # src/dtw.jl:133-145
    T = size(d.costtable, 2) - 1
    minpath = zeros(Int, T)
    minpath[end] = argmin(d.costtable[:, T+1])
    for i = reverse(2:T)
        minpath[i-1] = d.backpointer[minpath[i], i+1]
    end
    minpath
end

# align(src, tgt) -- src/align.jl:8-35
function align(src::Matrix{Float64}, tgt::Matrix{Float64})
    size(src, 1) == size(tgt, 1) || throw(DimensionMismatch("order of feature vector must be equal"))
    newtgt = similar(src)
    check(ccall((:vcmi_align, libvcmi), Cint,
                (Ptr{Float64}, Int64, Ptr{Float64}, Int64, Cint, Ptr{Float64}, Ptr{Int64}),
                src, size(src, 2), tgt, size(tgt, 2), size(src, 1), newtgt, C_NULL))
    src, newtgt
end

# batch method: n pairs in one launch
function align(srcs::Vector{Matrix{Float64}}, tgts::Vector{Matrix{Float64}})
    n = length(srcs)
    S = Int64[size(s, 2) for s in srcs]; T = Int64[size(t, 2) for t in tgts]
    newtgts = [similar(s) for s in srcs]
    GC.@preserve srcs tgts newtgts begin
        check(ccall((:vcmi_align_batch, libvcmi), Cint,
                    (Int64, Ptr{Ptr{Float64}}, Ptr{Int64}, Ptr{Ptr{Float64}}, Ptr{Int64}, Cint, Ptr{Ptr{Float64}}),
                    n, pointer.(srcs), S, pointer.(tgts), T, size(srcs[1], 1), pointer.(newtgts)))
    end
    srcs, newtgts
end

# ---------------------------------------------------------------------------------------------- E-step
# sufficient statistics of a diagonal GMM on joint features X (Dj,N); replaces the E-step inside
# `gmm[:fit](dataset.X')` of bin/train_gmm.jl:103
function estep_diag(X::Matrix{Float64}, w::Vector{Float64}, μ::Matrix{Float64}, σ²::Matrix{Float64})
    Dj, M = size(μ)
    S0 = Vector{Float64}(undef, M); S1 = similar(μ); S2 = similar(μ); ll = Ref{Float64}(0.0)
    check(ccall((:vcmi_estep_diag, libvcmi), Cint,
                (Ptr{Float64}, Int64, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                 Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}),
                X, size(X, 2), Dj, M, w, μ, σ², S0, S1, S2, ll))
    S0, S1, S2, ll[]
end

# full-covariance statistics: Σ is (Dj,Dj,M) as in the model file; S2 is (Dj,Dj,M) = Σₙ γₙₘ xₙxₙᵀ.  This is the
# E-step of the sklearn.mixture.GMM(covariance_type="full") that bin/train_gmm.jl:84-89 constructs.
# Which of the diagonal E-step's two paths the calling thread takes from 65536 frames on (include/vcmi.h): ESTEP_AUTO decides per
# call, on the device, from a sample of the call's own frames; ESTEP_HARD / ESTEP_SOFT pin one (e.g. on every rank of a training run)
const ESTEP_AUTO, ESTEP_HARD, ESTEP_SOFT = Cint(0), Cint(1), Cint(2)
estep_set_path(path::Integer) = check(ccall((:vcmi_estep_set_path, libvcmi), Cint, (Cint,), path))
function estep_get_path()
    p = Ref{Cint}(0)
    check(ccall((:vcmi_estep_get_path, libvcmi), Cint, (Ptr{Cint},), p))
    p[]
end

function estep_full(X::Matrix{Float64}, w::Vector{Float64}, μ::Matrix{Float64}, Σ::Array{Float64,3})
    Dj, M = size(μ)
    size(Σ) == (Dj, Dj, M) || throw(DimensionMismatch("Σ must be (Dj,Dj,M)"))
    S0 = Vector{Float64}(undef, M); S1 = similar(μ); S2 = similar(Σ); ll = Ref{Float64}(0.0)
    check(ccall((:vcmi_estep_full, libvcmi), Cint,
                (Ptr{Float64}, Int64, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                 Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}),
                X, size(X, 2), Dj, M, w, μ, Σ, S0, S1, S2, ll))
    S0, S1, S2, ll[]
end

# --------------------------------------------------------------------------- GV trajectory converter, post filter
# TrajectoryGVGMMMap(tgmm, μᵛ, Σᵛᵛ) and fvconvert(tgv, X; epochs, α) -- src/trajectory_gmmmap.jl:114-189
mutable struct TrajectoryGVGMMMap <: TrajectoryConverter
    tgmm::TrajectoryGMMMap
    μᵛ::Vector{Float64}
    Σᵛᵛ::Matrix{Float64}
    h::Ptr{Cvoid}
    function TrajectoryGVGMMMap(tgmm::TrajectoryGMMMap, μᵛ::Vector{Float64}, Σᵛᵛ::Matrix{Float64})
        @assert sum(μᵛ .< 0) == 0                                       # src/trajectory_gmmmap.jl:124
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:vcmi_trajgv_create, libvcmi), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ref{Ptr{Cvoid}}),
                    tgmm.h, μᵛ, Σᵛᵛ, h))
        tgv = new(tgmm, μᵛ, Σᵛᵛ, h[])
        finalizer(x -> ccall((:vcmi_trajgv_destroy, libvcmi), Cint, (Ptr{Cvoid},), x.h), tgv)
        tgv
    end
end
Base.length(t::TrajectoryGVGMMMap) = length(t.tgmm)
dim(t::TrajectoryGVGMMMap) = dim(t.tgmm)
ncomponents(t::TrajectoryGVGMMMap) = ncomponents(t.tgmm)
Base.size(t::TrajectoryGVGMMMap) = size(t.tgmm)

function fvconvert(tgv::TrajectoryGVGMMMap, X::Matrix{Float64}; epochs::Int=100, α::Float64=1.0e-5, verbose::Bool=false)
    size(X, 1) == dim(tgv) || throw(DimensionMismatch("Inconsistent dimentions."))
    Y = Matrix{Float64}(undef, dim(tgv) >> 1, size(X, 2))
    check(ccall((:vcmi_trajgv_convert, libvcmi), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Cint, Cdouble, Ptr{Float64}),
                tgv.h, X, size(X, 2), epochs, α, Y))
    Y
end

# vc(c::TrajectoryConverter, fm) for the GV converter -- src/common.jl:31-63: the chunks (default epochs / α, as the
# reference's loop calls fvconvert(c, phrase)) go to the device as one batch
function vc(tgv::TrajectoryGVGMMMap, fm::Matrix{Float64})
    size(fm, 1) == dim(tgv) + 1 || throw(DimensionMismatch("Inconsistent dimentions."))
    T = size(fm, 2)
    limit = length(tgv)
    spans = [(b, min(b + limit - 1, T)) for b in 1:limit:T]
    Xs = [fm[2:end, b:e] for (b, e) in spans]
    Ts = Int64[size(x, 2) for x in Xs]
    Ys = [Matrix{Float64}(undef, dim(tgv) >> 1, k) for k in Ts]
    GC.@preserve Xs Ys begin
        check(ccall((:vcmi_trajgv_convert_batch, libvcmi), Cint,
                    (Ptr{Cvoid}, Int64, Ptr{Ptr{Float64}}, Ptr{Int64}, Cint, Cdouble, Ptr{Ptr{Float64}}),
                    tgv.h, length(Xs), pointer.(Xs), Ts, 100, 1.0e-5, pointer.(Ys)))
    end
    out = Matrix{Float64}(undef, dim(tgv) >> 1 + 1, T)
    for (k, (b, e)) in enumerate(spans)
        out[2:end, b:e] = Ys[k]
    end
    out[1, :] = fm[1, :]
    out
end

# fvpostf(vs::VarianceScaling, src) -- src/gv.jl:6-21
struct VarianceScaling
    σ²::Vector{Float64}
end
function fvpostf(vs::VarianceScaling, src::Matrix{Float64})
    out = similar(src)
    check(ccall((:vcmi_variance_scaling, libvcmi), Cint, (Ptr{Float64}, Cint, Int64, Ptr{Float64}, Ptr{Float64}),
                src, size(src, 1), size(src, 2), vs.σ², out))
    out
end
function fvpostf!(vs::VarianceScaling, src::Matrix{Float64})          # in place: the library allows out == src
    check(ccall((:vcmi_variance_scaling, libvcmi), Cint, (Ptr{Float64}, Cint, Int64, Ptr{Float64}, Ptr{Float64}),
                src, size(src, 1), size(src, 2), vs.σ², src))
    src
end

# ---- the steps either side of vc without leaving HBM (SURVEY 8(f) rank 4) -------------------------------------------
# vc(c, fm, vs): out[2:end, :] = fvpostf(vs, vc(c, fm)[2:end, :]) -- src/common.jl:7-63 followed by src/gv.jl:10-15, the
# converted matrix filtered on the device before it is downloaded (one upload, one download)
function vc(g::GMMMap, fm::Matrix{Float64}, vs::VarianceScaling)
    size(fm, 1) == dim(g) + 1 || throw(DimensionMismatch("Inconsistent dimentions."))
    length(vs.σ²) == dim(g) || throw(DimensionMismatch("σ² must have one entry per converted feature row"))
    out = similar(fm)
    check(ccall((:vcmi_vc_frames_postf, libvcmi), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}),
                g.h, fm, size(fm, 2), vs.σ², out))
    out
end
function vc(t::TrajectoryGMMMap, fm::Matrix{Float64}, vs::VarianceScaling)
    size(fm, 1) == dim(t) + 1 || throw(DimensionMismatch("Inconsistent dimentions."))
    length(vs.σ²) == dim(t) >> 1 || throw(DimensionMismatch("σ² must have one entry per converted feature row"))
    out = Matrix{Float64}(undef, (size(fm, 1) - 1) >> 1 + 1, size(fm, 2))
    check(ccall((:vcmi_vc_traj_postf, libvcmi), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}),
                t.h, fm, size(fm, 2), vs.σ², out))
    out
end

# A Float64 matrix that lives in HBM: device address, shape and leading dimension (what a GPU array package, or the
# library's own `_dev` entry points, hand around).  push_delta and fvpostf! on it run where the data is, on `stream`.
struct DeviceMatrix
    ptr::Ptr{Float64}
    rows::Int
    cols::Int
    ld::Int
end
function push_delta(src::DeviceMatrix, out::DeviceMatrix; stream::Ptr{Cvoid}=C_NULL)      # src/datasets.jl:6-13
    (out.rows == 2src.rows && out.cols == src.cols) || throw(DimensionMismatch("out must be (2D,T)"))
    check(ccall((:vcmi_push_delta_dev, libvcmi), Cint, (Ptr{Float64}, Int64, Cint, Int64, Ptr{Float64}, Int64, Ptr{Cvoid}),
                src.ptr, src.ld, src.rows, src.cols, out.ptr, out.ld, stream))
    out
end
function fvpostf!(vs::VarianceScaling, src::DeviceMatrix; stream::Ptr{Cvoid}=C_NULL)      # src/gv.jl:10-15, in place
    length(vs.σ²) == src.rows || throw(DimensionMismatch("σ² must have one entry per feature row"))
    check(ccall((:vcmi_variance_scaling_dev, libvcmi), Cint,
                (Ptr{Float64}, Int64, Cint, Int64, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Cvoid}),
                src.ptr, src.ld, src.rows, src.cols, vs.σ², src.ptr, src.ld, stream))
    src
end

# align_mcep(src, tgt, α, fftlen; threshold, remove_silence) -- src/align.jl:38-55 (mc2e included)
function align_mcep(src::Matrix{Float64}, tgt::Matrix{Float64}, α::AbstractFloat, fftlen::Integer;
                    threshold::Float64=-14.0, remove_silence::Bool=true)
    size(src, 1) == size(tgt, 1) ||
        throw(DimensionMismatch("order of feature vector between source and target must be equal"))
    D, S = size(src)
    so = similar(src); to = similar(src); k = Ref{Int64}(0)
    check(ccall((:vcmi_align_mcep, libvcmi), Cint,
                (Ptr{Float64}, Int64, Ptr{Float64}, Int64, Cint, Cdouble, Cint, Cdouble, Cint, Ptr{Float64}, Ptr{Float64}, Ref{Int64}),
                src, S, tgt, size(tgt, 2), D, α, fftlen, threshold, remove_silence ? 1 : 0, so, to, k))
    so[:, 1:k[]], to[:, 1:k[]]
end

# GVDataset(path; ignore0th, add_delta, nmax) -- src/datasets.jl:134-183, from in-memory feature matrices (loading the
# `.jld` files stays with the caller): X = the (Dout, n) matrix of per-utterance variances var(tgt, 2)
struct GVDataset
    X::Matrix{Float64}
    function GVDataset(fms::Vector{Matrix{Float64}}; ignore0th::Bool=true, add_delta::Bool=false, nmax::Int=100)
        fms = fms[1:min(length(fms), nmax)]
        n = length(fms)
        n == 0 && return new(zeros(0, 0))
        D = size(fms[1], 1)
        all(f -> size(f, 1) == D, fms) || throw(DimensionMismatch("all feature matrices must share the feature dimension"))
        T = Int64[size(f, 2) for f in fms]
        Dout = (D - (ignore0th ? 1 : 0)) * (add_delta ? 2 : 1)
        out = Matrix{Float64}(undef, Dout, n); k = Ref{Int64}(0)
        check(ccall((:vcmi_gv_dataset, libvcmi), Cint,
                    (Int64, Ptr{Ptr{Float64}}, Ptr{Int64}, Cint, Cint, Cint, Ptr{Float64}, Ref{Int64}),
                    n, Ptr{Float64}[pointer(f) for f in fms], T, D, ignore0th ? 1 : 0, add_delta ? 1 : 0, out, k))
        X = out[:, 1:k[]]
        @assert all(isfinite.(X))
        new(X)
    end
end

# ------------------------------------------------------------------------------------- device-resident EM
# The loop behind `gmm[:fit](dataset.X')` (bin/train_gmm.jl:84-103) with parameters, statistics and whitening blocks
# kept in HBM.  `dX` / `dstats` are device pointers (e.g. from AMDGPU.jl); on several GPUs the caller all-reduces
# `dstats` between estep! and mstep!.
mutable struct GMMEM
    h::Ptr{Cvoid}
    Dj::Int
    M::Int
    function GMMEM(w::Vector{Float64}, μ::Matrix{Float64}, Σ::Array{Float64,3}; min_covar::Float64=1.0e-7)
        Dj, M = size(μ)
        size(Σ) == (Dj, Dj, M) || throw(DimensionMismatch("Σ must be (Dj,Dj,M)"))
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:vcmi_gmm_em_create, libvcmi), Cint,
                    (Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cdouble, Ref{Ptr{Cvoid}}),
                    Dj, M, w, μ, Σ, min_covar, h))
        em = new(h[], Dj, M)
        finalizer(e -> ccall((:vcmi_gmm_em_destroy, libvcmi), Cint, (Ptr{Cvoid},), e.h), em)
        em
    end
end

estep!(em::GMMEM, dX::Ptr{Float64}, N::Integer, dstats::Ptr{Float64}; stream::Ptr{Cvoid}=C_NULL) =
    check(ccall((:vcmi_gmm_em_estep_dev, libvcmi), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Cvoid}),
                em.h, dX, N, dstats, stream))

function mstep!(em::GMMEM, dstats::Ptr{Float64}; stream::Ptr{Cvoid}=C_NULL)
    ll = Ref{Float64}(0.0)
    check(ccall((:vcmi_gmm_em_mstep, libvcmi), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Cvoid}, Ref{Float64}),
                em.h, dstats, stream, ll))
    ll[]
end

function params(em::GMMEM)
    w = Vector{Float64}(undef, em.M); μ = Matrix{Float64}(undef, em.Dj, em.M); Σ = Array{Float64,3}(undef, em.Dj, em.Dj, em.M)
    check(ccall((:vcmi_gmm_em_get, libvcmi), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), em.h, w, μ, Σ))
    w, μ, Σ
end

end # module
