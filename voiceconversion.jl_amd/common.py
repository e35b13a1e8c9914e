"""Type hierarchy and the `vc` batch drivers -- reference src/common.jl:1-63."""


class AbstractConverter:            # src/common.jl:2
    pass


class FrameByFrameConverter(AbstractConverter):   # src/common.jl:3
    pass


class TrajectoryConverter(AbstractConverter):     # src/common.jl:4
    pass


def vc(c, fm, postfilter=None):
    """vc(c, fm): row 1 of `fm` is the power coefficient and is passed through; the remaining rows are
    converted -- frame by frame for a FrameByFrameConverter (src/common.jl:7-26; here one kernel launch
    over all T frames), in chunks of length(c) frames for a TrajectoryConverter (src/common.jl:31-63).
    postfilter: a VarianceScaling applied to the converted rows 2..end before the result leaves the device
    (out[2:end,:] = fvpostf(postfilter, vc(c, fm)[2:end,:]), src/gv.jl:10-15): one upload, one download."""
    if postfilter is None:
        return c._vc(fm)
    return c._vc(fm, postfilter)


def fvconvert(c, x, **kw):
    """fvconvert(c, x): src/gmmmap.jl:101-118 (vector or, as a batch extension, (D,T) matrix) and
    src/trajectory_gmmmap.jl:65-110 ((2D,T) matrix)."""
    return c._fvconvert(x, **kw)


def dim(c):
    return c._dim()


def ncomponents(c):
    return c._ncomponents()


def size(c):
    return (dim(c), len(c))
