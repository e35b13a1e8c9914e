"""GPU parity: DTW fit!/backward and align through the C-ABI -- BIT-EXACT against the reference's KATs
(test/dtw.jl:7-31), the committed golden tables and the C oracle."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vc():
    import voiceconversion_jl_amd as m
    assert m.device_count() >= 1
    return m


def test_reference_kat_3dim(vc):
    """test/dtw.jl:7-19"""
    v1 = np.array([[1., 2, 3], [1, 2, 4], [1, 8, 5], [10, 3, 6]]).T
    v2 = np.array([[1., 2, 3], [1, 2, 4], [1, 2, 5], [1, 8, 5], [10, 3, 6]]).T
    d = vc.DTW(bstep=1, fstep=0)
    indices = vc.fit_(d, v1, v2)
    assert indices.tolist() == [1, 2, 2, 3, 4]
    assert d.costtable.shape == (4, 6) and d.backpointer.shape == (4, 6)
    assert vc.backward(d).tolist() == [1, 2, 2, 3, 4]


def test_reference_kat_1dim(vc):
    """test/dtw.jl:21-31"""
    v1 = np.array([[0., 1, 2, 3, 4, 5]])
    v2 = np.array([[0., 0, 1, 2, 3, 4, 4, 5]])
    d = vc.DTW(bstep=1, fstep=0)
    assert vc.fit_(d, v1, v2).tolist() == [1, 1, 2, 3, 4, 5, 5, 6]


def test_golden_tables_bit_exact(vc):
    z = load_golden("dtw_cases.npz")
    names = [f"kat{k}" for k in range(2)] + [f"r{k}" for k in range(int(z["n_random"]))]
    for nm in names:
        fs, bs = (int(x) for x in z[f"{nm}_steps"])
        d = vc.DTW(fstep=fs, bstep=bs)
        path = vc.fit_(d, z[f"{nm}_tmpl"].T, z[f"{nm}_seq"].T)
        assert np.array_equal(path, z[f"{nm}_path"]), nm
        assert np.array_equal(d.costtable, z[f"{nm}_cost"].T), nm          # bit-exact Float64
        assert np.array_equal(d.backpointer, z[f"{nm}_bp"].T), nm
        assert np.array_equal(vc.fit_(d, z[f"{nm}_tmpl"].T, z[f"{nm}_seq"].T, tables=False), z[f"{nm}_path"]), nm
        if nm.startswith("r"):
            src, newtgt, apath = vc.align(z[f"{nm}_tmpl"].T, z[f"{nm}_seq"].T, return_path=True)
            assert np.array_equal(newtgt, z[f"{nm}_align_newtgt"].T), nm
            assert np.array_equal(apath, z[f"{nm}_align_path"]), nm


def _warped_pair(rng, S, T, D):
    t = rng.standard_normal((S, D))
    idx = np.clip(np.sort(rng.integers(0, S, T)), 0, S - 1)
    return t, t[idx] + 0.2 * rng.standard_normal((T, D))


@pytest.mark.parametrize("S,T,D,fs,bs", [(500, 500, 40, 0, 2), (450, 550, 40, 0, 1), (1024, 300, 13, 0, 2), (130, 700, 40, 2, 3),
                                         (1, 7, 3, 0, 1), (9, 1, 2, 0, 2), (1500, 200, 5, 0, 2), (300, 100, 150, 0, 2)])
def test_vs_oracle_bit_exact(vc, S, T, D, fs, bs):
    """Config-4-sized pairs and the edge shapes: S=1, T=1, S>1024 and D>128 (generic kernel), wide step
    windows (byte codes), LDS-resident and HBM-resident step codes."""
    from oracle import c_oracle as co
    rng = np.random.default_rng(S * 7 + T)
    t, s = _warped_pair(rng, S, T, D)
    p_ref, c_ref, b_ref = co.dtw_fit(t, s, fs, bs)
    d = vc.DTW(fstep=fs, bstep=bs)
    path = vc.fit_(d, t.T, s.T)
    assert np.array_equal(path, p_ref)
    assert np.array_equal(d.costtable, c_ref.T)
    assert np.array_equal(d.backpointer, b_ref.T)
    assert np.array_equal(vc.fit_(d, t.T, s.T, tables=False), p_ref)


def test_batch_ragged_and_align(vc):
    from oracle import c_oracle as co
    rng = np.random.default_rng(99)
    pairs = [_warped_pair(rng, int(rng.integers(40, 300)), int(rng.integers(40, 300)), 40) for _ in range(37)]
    d = vc.DTW(fstep=0, bstep=2)
    paths = vc.fit_batch(d, [t.T for t, _ in pairs], [s.T for _, s in pairs])
    outs = vc.align_batch([t.T for t, _ in pairs], [s.T for _, s in pairs])
    for (t, s), p, (src, newtgt) in zip(pairs, paths, outs):
        assert np.array_equal(p, co.dtw_fit(t, s, 0, 2, tables=False))
        nt_ref, _ = co.align(t, s)
        assert np.array_equal(newtgt, nt_ref.T)
        assert src.shape == newtgt.shape


@pytest.mark.parametrize("D", [40, 13, 41, 46])
def test_device_resident_batch(vc, D):
    """vcmi_dtw_fit_batch_dev: features already in HBM, ragged pairs spanning several 256-frame row blocks and
    128-frame column blocks of the observation kernel; bit-exact paths."""
    import torch
    from oracle import c_oracle as co
    from voiceconversion_jl_amd.dtw import fit_batch_dev
    rng = np.random.default_rng(D)
    shapes = [(500, 500), (513, 129), (64, 550), (129, 128), (550, 450), (1, 7), (300, 1)]
    pairs = [_warped_pair(rng, S, T, D) for S, T in shapes]
    chunks, toff, soff, off = [], [], [], 0
    for t, s in pairs:
        toff.append(off); chunks.append(t.ravel()); off += t.size
        soff.append(off); chunks.append(s.ravel()); off += s.size
    feats = torch.from_numpy(np.concatenate(chunks)).cuda()
    d = vc.DTW(fstep=0, bstep=2)
    paths, poff = fit_batch_dev(d, feats, toff, [s for s, _ in shapes], soff, [t for _, t in shapes], D)
    got = paths.cpu().numpy()
    for i, (t, s) in enumerate(pairs):
        ref = co.dtw_fit(t, s, 0, 2, tables=False)
        assert np.array_equal(got[poff[i]:poff[i] + len(ref)], ref), f"pair {i} {shapes[i]}"


def test_empty_sequence_and_errors(vc):
    d = vc.DTW()
    p = vc.fit_(d, np.ones((3, 4)), np.zeros((3, 0)))
    assert p.shape == (0,) and d.costtable.shape == (4, 1) and d.costtable[:, 0].tolist() == [1, 2, 3, 4]
    assert vc.fit_batch(d, [], []) == []
    with pytest.raises(vc.DimensionMismatch):
        vc.align(np.zeros((3, 5)), np.zeros((4, 5)))           # src/align.jl:11-13


def test_online_update_matches_fit(vc):
    """update! (src/dtw.jl:61-90) column by column reproduces fit!'s tables."""
    rng = np.random.default_rng(5)
    t, s = _warped_pair(rng, 12, 9, 4)
    d1 = vc.DTW(bstep=2)
    vc.fit_(d1, t.T, s.T)
    d2 = vc.DTW(bstep=2)
    vc.set_template_(d2, t.T)
    for k in range(s.shape[0]):
        vc.update_(d2, s[k])
    assert np.array_equal(d1.costtable, d2.costtable) and np.array_equal(d1.backpointer, d2.backpointer)
    assert np.array_equal(vc.backward(d2), vc.backward(d1))


@pytest.mark.parametrize("bs", [1, 2])
def test_fused_kernel_strips_ties_and_two_kernel_path(vc, bs):
    """The fused forward kernel (observation + recurrence in one hand-scheduled loop, two rows per lane, waves coupled
    through sequence-tagged LDS rings, long templates split into row strips that hand their top two rows over through
    HBM) against the oracle AND against the observation + recurrence kernels it replaced (still the path of tables,
    D > 48 and wide windows).  Shapes: strip boundaries (512 / 513 / 549 / 1025 / 1537 rows), odd and even S, every
    wave count, T not a multiple of 16, padded D, coarse features (exact cost ties exercise the strict '<' rule)."""
    from oracle import c_oracle as co
    from voiceconversion_jl_amd import _lib
    rng = np.random.default_rng(1234 + bs)
    shapes = [(512, 160, 40), (513, 97, 40), (549, 130, 40), (550, 33, 24), (1025, 70, 40), (1537, 40, 8), (127, 500, 40),
              (128, 17, 40), (129, 16, 40), (255, 15, 13), (2, 300, 40), (3, 1, 40), (383, 64, 32), (64, 65, 1),
              # 40 < D <= 48: the three-chunk loops (two columns per iteration: odd and even T, T = 1); D = 41 reads the
              # unpadded 41-double rows directly, 42..48 run padded in the DMAX = 48 kernel
              (500, 500, 41), (549, 131, 41), (1025, 70, 41), (130, 1, 41), (2, 2, 41), (255, 16, 41), (64, 33, 41),
              (513, 97, 48), (300, 64, 48), (127, 47, 45), (1100, 30, 42), (3, 9, 47), (129, 2, 44)]
    pairs = []
    for k, (S, T, D) in enumerate(shapes):
        t, s = _warped_pair(rng, S, T, D)
        if k % 2:
            t, s = np.round(t * 2) / 2, np.round(s * 2) / 2
        pairs.append((t, s))
    d = vc.DTW(fstep=0, bstep=bs)
    for Dsel in sorted({D for _, _, D in shapes}):
        sub = [(t, s) for (t, s), (_, _, D) in zip(pairs, shapes) if D == Dsel]
        fused = vc.fit_batch(d, [t.T for t, _ in sub], [s.T for _, s in sub])
        _lib.debug_force(_lib.DBG_DTW_TWO_KERNELS)
        try:
            two = vc.fit_batch(d, [t.T for t, _ in sub], [s.T for _, s in sub])
        finally:
            _lib.debug_force(0)
        for (t, s), a, b in zip(sub, fused, two):
            ref = co.dtw_fit(t, s, 0, bs, tables=False)
            assert np.array_equal(a, ref), (t.shape, s.shape)
            assert np.array_equal(b, ref), (t.shape, s.shape)


def test_fused_kernel_more_workgroups_than_slots(vc):
    """700 small pairs + 40 two-strip pairs in one launch: several rounds of workgroups per CU, upper strips waiting on
    flags set by workgroups earlier in the grid; align output included."""
    from oracle import c_oracle as co
    rng = np.random.default_rng(77)
    pairs = [_warped_pair(rng, int(rng.integers(20, 140)), int(rng.integers(20, 90)), 16) for _ in range(700)]
    pairs += [_warped_pair(rng, int(rng.integers(513, 700)), int(rng.integers(20, 60)), 16) for _ in range(40)]
    outs = vc.align_batch([t.T for t, _ in pairs], [s.T for _, s in pairs])
    d = vc.DTW(fstep=0, bstep=2)
    paths = vc.fit_batch(d, [t.T for t, _ in pairs], [s.T for _, s in pairs])
    for i in list(range(0, 740, 23)) + list(range(700, 740)):
        t, s = pairs[i]
        assert np.array_equal(paths[i], co.dtw_fit(t, s, 0, 2, tables=False)), i
        assert np.array_equal(outs[i][1], co.align(t, s)[0].T), i


@pytest.mark.parametrize("S,T", [(60, 12000), (300, 9800)])
def test_long_sequence_falls_back_to_two_kernels(vc, S, T):
    """A sequence of more than ~9.7k frames (50 s at a 5 ms shift) exceeds the LDS budget of the fused forward kernel (it
    keeps 16 bytes per column of strip-boundary costs in LDS); path-only calls must then take the observation + recurrence
    kernels instead of failing, with the same bit-exact path; align included."""
    from oracle import c_oracle as co
    rng = np.random.default_rng(S + T)
    t, s = _warped_pair(rng, S, T, 8)
    d = vc.DTW(fstep=0, bstep=2)
    ref = co.dtw_fit(t, s, 0, 2, tables=False)
    assert np.array_equal(vc.fit_(d, t.T, s.T, tables=False), ref)
    assert np.array_equal(vc.fit_batch(d, [t.T, t[:40].T], [s.T, s[:100].T])[0], ref)
    src, newtgt = vc.align(t.T, s.T)
    assert np.array_equal(newtgt, co.align(t, s)[0].T)


@pytest.mark.parametrize("S,T,fs,bs,tables", [(500, 40_000, 0, 2, True), (320, 41_000, 0, 3, False), (200, 45_000, 1, 1, False)])
def test_long_sequence_short_template_beyond_the_lds_path(vc, S, T, fs, bs, tables):
    """a SEQUENCE of more than ~34k frames with a template the two-kernel fast path would take (S <= 1024, D <= 96)
    needs 4 T bytes of path in LDS on top of the cost columns -- beyond 160 KB the launch failed (VCMI_ERR_HIP) whenever the
    fused kernel did not apply: tables requested, bstep not in {1, 2}, fstep != 0.  These pairs now go to the generic kernel on
    its HBM scratch.  Bit-exact path and tables (src/dtw.jl:93-145: no length limit)."""
    from oracle import c_oracle as co
    rng = np.random.default_rng(S + T)
    t, s = _warped_pair(rng, S, T, 5)
    d = vc.DTW(fstep=fs, bstep=bs)
    if tables:
        pref, cref, bref = co.dtw_fit(t, s, fs, bs)
        assert np.array_equal(vc.fit_(d, t.T, s.T), pref)
        assert np.array_equal(d.costtable, cref.T) and np.array_equal(d.backpointer, bref.T)
    else:
        assert np.array_equal(vc.fit_(d, t.T, s.T, tables=False), co.dtw_fit(t, s, fs, bs, tables=False))


@pytest.mark.parametrize("S,T,bs", [(7000, 600, 2), (6100, 6400, 1)])
def test_long_template_runs_on_an_hbm_scratch(vc, S, T, bs):
    """A TEMPLATE of more than ~5800 frames (half a minute at a 5 ms shift) no longer fits two cost columns, the path and the
    align() lists in LDS: the call used to fail ("exceeds the supported length"; the reference has no limit,
    src/dtw.jl:11-59).  The generic kernel now keeps those arrays in a per-pair HBM scratch: the same bit-exact path and
    newtgt, in a batch with a short pair too."""
    from oracle import c_oracle as co
    rng = np.random.default_rng(S + T)
    t, s = _warped_pair(rng, S, T, 6)
    d = vc.DTW(fstep=0, bstep=bs)
    ref = co.dtw_fit(t, s, 0, bs, tables=False)
    assert np.array_equal(vc.fit_(d, t.T, s.T, tables=False), ref)
    both = vc.fit_batch(d, [t.T, t[:40].T], [s.T, s[:100].T])
    assert np.array_equal(both[0], ref) and np.array_equal(both[1], co.dtw_fit(t[:40], s[:100], 0, bs, tables=False))
    if bs == 2:
        src, newtgt = vc.align(t.T, s.T)
        assert np.array_equal(newtgt, co.align(t, s)[0].T)


@pytest.mark.parametrize("D", [16, 41])
def test_column_segments_and_persistent_workgroups(vc, D):
    """More jobs than the device has slots and sequences long enough: every strip's columns are cut into segments (a segment
    starts from the last cost column of its predecessor; upper strips take each segment's boundary costs from the strip
    below) and persistent workgroups draw the jobs from a ticket counter.  Bit-exact against the oracle, against
    whole-length jobs and against one workgroup per job in grid order; templates of one, two, three and four strips
    (packed one-wave and wide bottom strips), T not a multiple of 16, exact ties."""
    from oracle import c_oracle as co
    from voiceconversion_jl_amd import _lib
    rng = np.random.default_rng(4242 + D)
    shapes = [(int(rng.integers(100, 500)), int(rng.integers(130, 330))) for _ in range(560)]
    shapes += [(int(rng.integers(513, 640)), int(rng.integers(130, 300))) for _ in range(30)]      # packed bottom + upper
    shapes += [(int(rng.integers(700, 1000)), int(rng.integers(130, 260))) for _ in range(10)]     # wide bottom + upper
    shapes += [(1100, 143), (1537, 131)]                                                           # three / four strips
    pairs = []
    for k, (S, T) in enumerate(shapes):
        t, s = _warped_pair(rng, S, T, D)
        if k % 3 == 0:
            t, s = np.round(t * 2) / 2, np.round(s * 2) / 2
        pairs.append((t, s))
    d = vc.DTW(fstep=0, bstep=2)
    tl, sl = [t.T for t, _ in pairs], [s.T for _, s in pairs]
    paths = vc.fit_batch(d, tl, sl)
    # (DBG_DTW_WHOLE_FIRST: round 4's schedule experiment -- a full round's worth of single-strip pairs as whole-length jobs,
    # only the rest cut; DBG_DTW_TWO_SEGMENTS: at most two segments per strip)
    for flag in (_lib.DBG_DTW_NO_SEGMENTS, _lib.DBG_DTW_GRID_ORDER, _lib.DBG_DTW_NO_SEGMENTS | _lib.DBG_DTW_GRID_ORDER,
                 _lib.DBG_DTW_WHOLE_FIRST, _lib.DBG_DTW_TWO_SEGMENTS):
        _lib.debug_force(flag)
        try:
            other = vc.fit_batch(d, tl, sl)
        finally:
            _lib.debug_force(0)
        assert all(np.array_equal(a, b) for a, b in zip(paths, other)), flag
    for i in list(range(0, len(pairs), 29)) + list(range(560, len(pairs))):
        t, s = pairs[i]
        assert np.array_equal(paths[i], co.dtw_fit(t, s, 0, 2, tables=False)), (i, shapes[i])
    outs = vc.align_batch(tl[550:], sl[550:])
    for (t, s), (src, newtgt) in zip(pairs[550:], outs):
        assert np.array_equal(newtgt, co.align(t, s)[0].T)


@pytest.mark.gpu
def test_segmented_jobs_at_benchmark_size_repeatedly(vc):
    """The jobs of the fused kernel hand boundary pairs and last cost columns to each other through agent-scope (sc1) stores
    and loads, ordered by hand, instead of acquire / release fences (csrc/dtw.hip, dtw_wait_flag): a race there would be a
    rare event, so the benchmark batch (1000 pairs of ~500 x 500 frames: 5465 jobs drawn by 512 persistent workgroups on all
    eight XCDs) runs several times, segmented and whole-length, and every path must come out the same each time; a sample
    is checked against the oracle."""
    from oracle import c_oracle as co
    from voiceconversion_jl_amd import _lib
    rng = np.random.default_rng(77)
    pairs = []
    for _ in range(1000):
        S, T = int(rng.integers(450, 551)), int(rng.integers(450, 551))
        pairs.append(_warped_pair(rng, S, T, 40))
    d = vc.DTW(fstep=0, bstep=2)
    tl, sl = [t.T for t, _ in pairs], [s.T for _, s in pairs]
    _lib.debug_force(_lib.DBG_DTW_NO_SEGMENTS | _lib.DBG_DTW_GRID_ORDER)
    try:
        whole = vc.fit_batch(d, tl, sl)
    finally:
        _lib.debug_force(0)
    for rep in range(6):
        _lib.debug_force(_lib.DBG_DTW_WHOLE_FIRST if rep >= 4 else 0)       # the last two: whole-length jobs first (round 4 experiment)
        try:
            seg = vc.fit_batch(d, tl, sl)
        finally:
            _lib.debug_force(0)
        bad = [i for i, (a, b) in enumerate(zip(whole, seg)) if not np.array_equal(a, b)]
        assert not bad, (rep, bad[:10])
    for i in range(0, 1000, 97):
        t, s = pairs[i]
        assert np.array_equal(whole[i], co.dtw_fit(t, s, 0, 2, tables=False)), i
