"""wf_array keeps its data in HBM: which copy is authoritative, and what crosses PCIe (pythtb.py:2644-2672
for wf[i,j]; the reference's `_wfs` attribute is its live storage).  Counted with tbk_ctx_transfer_stats."""
import numpy as np
import pytest

import helpers as hp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tb():
    import pythtb_amd
    return pythtb_amd


def stats(tb, reset=False):
    return tb._lib.default_context().transfer_stats(reset)


def test_reading_points_between_berry_calls_moves_no_array(tb):
    """examples/kane_mele.py-style use: solve, look at wf[i,j], Berry phases, look again, flux."""
    m = hp.kane_mele(tb.tb_model, "odd")
    w = tb.wf_array(m, [1025, 513])            # 1025*513*4*4*16 B = 134 MB: larger than the whole-mirror bound
    w.solve_on_grid([-0.5, -0.5])
    stats(tb, reset=True)
    a = np.array(w[3, 4])
    wl = w.berry_phase([0, 1], 0, contin=False, berry_evals=True)
    b = w[3, 4]
    assert np.array_equal(a, b) and a.shape == (4, 2, 2)
    assert b.flags.writeable                   # writable like the reference's view (test_getitem_is_writable_like_the_reference)
    fl = w.berry_flux([0, 1])
    c = w[-1, -1]
    assert np.max(np.abs(c - w[0, 0] * np.exp(-2j * np.pi * (m._orb[:, 0] + m._orb[:, 1]))[None, :, None])) < 1e-14
    st = stats(tb)
    assert st["h2d_calls"] == 0 and st["h2d_bytes"] == 0
    assert st["d2h_bytes"] <= 5 * 4 * 4 * 16   # five single points
    assert wl.shape == (513, 2) and np.isfinite(fl)
    # a small array is mirrored whole on the first read, and still never re-uploaded
    s = tb.wf_array(hp.haldane(tb.tb_model, 0.2), [31, 31])
    s.solve_on_grid([0.0, 0.0])
    stats(tb, reset=True)
    p1 = s.berry_phase([0], 1)
    x = s[5, 6]
    p2 = s.berry_phase([0], 1)
    st = stats(tb)
    assert st["h2d_calls"] == 0 and st["d2h_calls"] == 1
    assert np.array_equal(p1, p2) and x.shape == (2, 2)


def test_setitem_on_resident_array_uploads_one_point(tb):
    from oracle import tb_oracle as orc
    m = hp.haldane(tb.tb_model, 0.3)
    mesh, start = [40, 33], [0.1, 0.2]
    w = tb.wf_array(m, mesh)
    w.solve_on_grid(start)
    host = np.array(w.to_host())
    stats(tb, reset=True)
    v = host[7, 9] * np.exp(0.3j)             # a gauge change at one point leaves plaquette sums unchanged ...
    w[7, 9] = v
    rot = np.array([[np.cos(0.4), np.sin(0.4)], [-np.sin(0.4), np.cos(0.4)]]) @ host[11, 3]
    w[11, 3] = rot                             # ... a band rotation does not
    st = stats(tb)
    assert st["h2d_calls"] == 2 and st["h2d_bytes"] == 2 * 2 * 2 * 16
    host[7, 9] = v
    host[11, 3] = rot
    ref = orc.berry_flux(host, 2, [0], individual_phases=True, vectorised=True)
    got = w.berry_flux([0], individual_phases=True)
    assert np.max(np.abs((got - ref + np.pi) % (2 * np.pi) - np.pi)) < 1e-10
    assert np.array_equal(w[11, 3], rot) and np.array_equal(w.to_host(), host)
    assert stats(tb)["h2d_calls"] == 2


@pytest.mark.parametrize("mesh", [[40, 33], [1025, 513]])
def test_getitem_is_writable_like_the_reference(tb, mesh):
    """pythtb.py:2662-2666 returns `self._wfs[key]`, a live view: `wf[i,j][0] *= z` changes the array.  Here the point is
    handed out with a snapshot and written back before the next device use -- for the mirrored small array and for the
    134 MB one whose points are fetched one at a time (VERDICT r3 #9a)."""
    from oracle import tb_oracle as orc
    m = hp.haldane(tb.tb_model, 0.3)
    start = [0.1, 0.2]
    w = tb.wf_array(m, mesh)
    w.solve_on_grid(start)
    twin = tb.wf_array(m, mesh)                 # (deterministic kernels: the same bits; `w` itself keeps no mirror yet)
    twin.solve_on_grid(start)
    ref = np.array(twin.to_host())              # what the reference's `_wfs` would hold
    del twin
    z = np.exp(0.7j)
    stats(tb, reset=True)
    w[7, 9][0] *= z                             # the temporary dies at once: the write must not be lost
    ref[7, 9][0] *= z
    held = w[11, 3]                             # a view the script keeps ...
    row = w[12, 4][1]                           # ... and a sub-view of a point whose array is not kept
    assert np.array_equal(w[7, 9], ref[7, 9])
    rot = np.array([[np.cos(0.4), np.sin(0.4)], [-np.sin(0.4), np.cos(0.4)]])
    held[...] = rot @ held                      # band rotation at one point: changes plaquette sums
    ref[11, 3] = rot @ ref[11, 3]
    row *= -1.0
    ref[12, 4][1] *= -1.0
    sub = [slice(0, 20), slice(0, 20)]
    got = w.berry_flux([0], individual_phases=True)[tuple(sub)]
    exp = orc.berry_flux(ref[:21, :21], 2, [0], individual_phases=True, vectorised=True)
    assert np.max(np.abs((got - exp + np.pi) % (2 * np.pi) - np.pi)) < 1e-10
    st = stats(tb)
    assert st["h2d_calls"] == 1 and st["h2d_bytes"] == 3 * 2 * 2 * 16      # three points in ONE batched upload, never the array
    assert np.array_equal(w[11, 3], ref[11, 3]) and np.array_equal(w[12, 4], ref[12, 4])
    assert np.array_equal(w.to_host()[:21, :21], ref[:21, :21])
    # a later write through the same held view is seen too, and a device-side write shows up in it
    held[0] *= 1j
    ref[11, 3][0] *= 1j
    assert np.array_equal(w[11, 3], ref[11, 3])
    w.solve_on_grid(start)
    fresh = tb.wf_array(m, mesh)
    fresh.solve_on_grid(start)
    assert np.array_equal(held, fresh[11, 3]) and np.array_equal(row, fresh[12, 4][1])
    # (the write to `held` was still pending when solve_on_grid overwrote the whole array: nothing had to be uploaded for it)
    st = stats(tb)
    assert st["h2d_calls"] <= 2 and st["h2d_bytes"] <= 4 * 2 * 2 * 16
    held[1] *= 2.0                              # and the refreshed view is still live
    ref2 = np.array(fresh[11, 3])
    ref2[1] *= 2.0
    w.berry_flux([0])
    assert np.array_equal(w.to_host()[11, 3], ref2)


def test_read_loop_over_more_points_than_the_pool_follows_stays_resident(tb):
    """`for i, j: x = wf[i, j]` past the pool's cap used to download the whole array and leave it exported, so
    every later berry_* call re-uploaded it.  Chunks nobody holds an array of are recycled instead: after reading 3 000 points
    of a 1024^2 array (cap lowered to 1 024 points) two berry_flux calls move no array in either direction, and a write made on
    the way still counts."""
    m = hp.haldane(tb.tb_model, 0.3)
    w = tb.wf_array(m, [1025, 1025])                 # 67 MB: past _SMALL_MIRROR_BYTES, points are fetched into the pool
    w.solve_on_grid([0.0, 0.0])
    w._PT_TRACK_MAX = 1024
    f0 = w.berry_flux([0])
    stats(tb, reset=True)
    acc = 0.0
    for i in range(3):
        for j in range(1000):
            x = w[i, j]
            acc += abs(x[0, 0])
    w[2, 500][0] *= -1.0                             # (a band's sign: gauge only, the flux is unchanged; it must still be uploaded)
    assert not w._host_exported and len(w._pt_copies) <= 1024
    assert abs(w.berry_flux([0]) - f0) < 1e-9 and abs(w.berry_flux([0]) - f0) < 1e-9
    st = stats(tb)
    assert st["h2d_bytes"] == 2 * 2 * 16 and st["h2d_calls"] == 1          # the one changed point
    assert st["d2h_bytes"] < 3100 * 64 + 4096                              # the points read (+ the small results), never 67 MB


def test_exported_mirror_stays_live(tb):
    """A script may keep the array it got from `_wfs` and write to it at any time: every later
    device call must see those writes, and device-side writes must show up in the held array."""
    from oracle import tb_oracle as orc
    m = hp.haldane(tb.tb_model, 0.3)
    mesh, start = [24, 19], [0.0, 0.0]
    w = tb.wf_array(m, mesh)
    w.solve_on_grid(start)
    held = w._wfs                               # exported: host is authoritative from here on
    f0 = w.berry_flux([0], individual_phases=True)
    held[5, 5] = held[5, 5][::-1].copy()        # swap the bands at one point, through the held array
    f1 = w.berry_flux([0], individual_phases=True)
    ref = orc.berry_flux(np.array(held), 2, [0], individual_phases=True, vectorised=True)
    assert np.max(np.abs((f1 - ref + np.pi) % (2 * np.pi) - np.pi)) < 1e-10
    assert np.max(np.abs(f1 - f0)) > 1e-3
    w.solve_on_grid(start)                      # device-side write: the held array is refreshed in place
    assert held is w._wfs
    fresh = tb.wf_array(m, mesh)
    fresh.solve_on_grid(start)
    assert np.array_equal(held, fresh.to_host())
    w.impose_loop(0)
    assert np.array_equal(held[-1], held[0])
    w.release_host()                            # promise: no more writes through `held`
    stats(tb, reset=True)
    w.berry_flux([0])
    w.berry_phase([0], 1)
    assert stats(tb)["h2d_calls"] == 0


def test_position_hwf_mesh_reads_the_resident_array(tb):
    """cubic slab HWF loop (examples/cubic_slab_hwf.py) without the array leaving the device."""
    m3 = hp.quiet(tb.tb_model, 3, 3, np.identity(3), [[0, 0, 0]])
    for R in ([1, 0, 0], [0, 1, 0], [0, 0, 1]):
        m3.set_hop(-1.0, 0, 0, R)
    slab = m3.cut_piece(9, 2, glue_edgs=False)
    w = tb.wf_array(slab, [120, 90])
    w.solve_on_grid([0.0, 0.0])
    stats(tb, reset=True)
    hwfc = w.position_hwf_mesh(range(4), 2)
    one = w.position_hwf([7, 8], range(4), 2)
    ex = w.position_expectation([7, 8], [0, 2], 2)
    st = stats(tb)
    assert st["h2d_calls"] == 0 and st["d2h_calls"] == 0
    assert hwfc.shape == (120, 90, 4) and np.array_equal(hwfc[7, 8], one)
    host = w.to_host()
    ref = slab.position_hwf(host[7, 8][:4], 2)
    assert np.max(np.abs(ref - one)) < 1e-12
    assert np.max(np.abs(ex - slab.position_expectation(host[7, 8][[0, 2]], 2))) < 1e-13
    hw2, vec2 = w.position_hwf_mesh(range(4), 2, hwf_evec=True, basis="orbital")
    r2, v2 = slab.position_hwf(host.reshape(-1, 9, 9)[:, :4], 2, hwf_evec=True, basis="orbital")
    assert np.max(np.abs(hw2.reshape(-1, 4) - r2)) < 1e-12
    assert np.max(np.abs(np.abs(vec2.reshape(-1, 4, 9)) - np.abs(v2))) < 1e-9


def test_multi_gpu_drivers_single_rank_rccl_and_every_ranks_share(tb):
    """pythtb_amd/multi.py on the one GPU of the box: (i) world = 1 through the RCCL all-gather-v (send/recv to self),
    (ii) every rank's share of a world of 3 and of 8 computed in turn with a recording communicator -- the
    assembled arrays equal the unsharded calls bit for bit (configs[3]: strings along axis 0 cut along axis 1,
    uneven counts; configs[4]: axis-0 slabs of a 3-D mesh)."""
    import ctypes as C
    from pythtb_amd import _lib, multi
    lib, ctx = _lib.lib, _lib.default_context()
    km = hp.kane_mele(tb.tb_model, "odd")
    mesh, start = [65, 19], [-0.5, -0.5]
    full = tb.wf_array(km, mesh)
    full.solve_on_grid(start)
    ref = full.berry_phase([0, 1], 0, contin=False, berry_evals=True)
    m16 = hp.cubic16(tb.tb_model)
    mesh3, start3 = [10, 9, 33], [0.0, 0.1, 0.2]
    full3 = tb.wf_array(m16, mesh3)
    g3 = full3.solve_on_grid(start3)
    ref3 = full3.berry_phase(range(8), 2, contin=False)

    uid = (C.c_ubyte * 128)()
    _lib.check(lib.tbk_comm_unique_id(uid))
    rccl = multi.RcclComm(ctx, bytes(uid), 1, 0)
    try:
        got, _ = multi.wilson_loops_sharded(tb.wf_array, km, mesh, start, [0, 1], rccl, 0, 1)
        assert np.array_equal(got, ref)
        got3, gaps3 = multi.mesh_phases_sharded(tb.wf_array, m16, mesh3, start3, list(range(8)), rccl, 0, 1)
        assert np.array_equal(got3, ref3) and np.array_equal(gaps3, g3)
        # uneven blocks through the all-gather-v itself (one rank: its own block at displacement 0)
        x = np.arange(7, dtype=float)
        assert np.array_equal(rccl.allgatherv(x, [7]), x)
        assert np.array_equal(rccl.allgather(x), x.reshape(1, 7))
        # the eigenvalue gather of a sharded solve_all: band-major rows, device-resident from the k
        # chunk's upload to the gathered array's download; and with the k list generated on the device
        hal = hp.haldane(tb.tb_model, 0.2)
        k = np.random.default_rng(8).random((1000, 2))
        assert np.array_equal(multi.solve_all_sharded(hal, k, rccl, 0, 1), hal.solve_all(k))
        assert np.array_equal(rccl.allgatherv_rows(np.arange(12.0).reshape(3, 4), [4]), np.arange(12.0).reshape(3, 4))
        ev16 = multi.solve_all_mesh_sharded(m16, [6, 5, 7], rccl, 0, 1)
        assert np.array_equal(ev16, m16.solve_all(m16.k_uniform_mesh([6, 5, 7])))
        ends = multi.solve_all_mesh_sharded(m16, [6, 5, 7], rccl, 0, 1, download=False)
        assert np.array_equal(ends, ev16[:, [0, -1]])
        # the ROOTED gather (tbk_comm_gatherv_rows_f64) and row counts beyond one ncclGroup (32 rows each)
        assert np.array_equal(rccl.gatherv_rows(np.arange(12.0).reshape(3, 4), [4], root=0), np.arange(12.0).reshape(3, 4))
        wide = np.random.default_rng(9).random((130, 5))
        assert np.array_equal(rccl.allgatherv_rows(wide, [5]), wide) and np.array_equal(rccl.gatherv_rows(wide, [5]), wide)
        st = {}
        assert np.array_equal(multi.solve_all_mesh_sharded(m16, [6, 5, 7], rccl, 0, 1, root=0, stats=st), ev16)
        assert st["recv_bytes"] == ev16.nbytes and st["sent_bytes"] == ev16.nbytes and st["solve_ms"] > 0 and st["gather_ms"] > 0
        assert np.array_equal(multi.solve_all_mesh_sharded(m16, [6, 5, 7], rccl, 0, 1, download=False, root=0), ev16[:, [0, -1]])
        assert np.array_equal(multi.solve_all_sharded(hal, k, rccl, 0, 1, root=0), hal.solve_all(k))
        # a solve that fails is reported by the CHECKED entry point the drivers now use: a NaN model makes the
        # QL iteration hit its limit; the error is raised here and the context's status is clean for the next call
        bad = hp.cubic16(tb.tb_model)
        bad.set_onsite(float("nan"), 3, mode="reset")
        with pytest.raises(Exception, match="did not converge"):
            multi.solve_all_mesh_sharded(bad, [4, 4, 4], rccl, 0, 1, root=0)
        assert np.array_equal(multi.solve_all_mesh_sharded(m16, [6, 5, 7], rccl, 0, 1), ev16)
    finally:
        rccl.close()

    class Recorder(object):
        """Stands in for the collective when the ranks run one after another on one GPU."""
        def __init__(self, world):
            self.world, self.parts, self.rank = world, {}, 0
        def allgatherv(self, mine, counts):
            key = len(self.parts.setdefault(self.rank, []))
            self.parts[self.rank].append(np.array(mine, dtype=float).reshape(-1))
            assert self.parts[self.rank][key].size == counts[self.rank]
            return np.zeros(int(sum(counts)))                 # placeholder: assembled below from all ranks

    for world in (3, 8):
        rec = Recorder(world)
        for r in range(world):
            rec.rank = r
            multi.wilson_loops_sharded(tb.wf_array, km, mesh, start, [0, 1], rec, r, world)
        got_w = np.concatenate([rec.parts[r][0] for r in range(world)]).reshape(19, 2)
        # (bit for bit except the string on the periodic-image column, which the last rank solves as a column of its own while
        # the unsharded array takes it from column 0's lanes: equal to rounding -- tests/test_gpu_parity.py::test_sharded_windows...)
        assert np.array_equal(got_w[:-1], ref[:-1]) and np.max(np.abs(got_w[-1] - ref[-1])) < 1e-13
        rec = Recorder(world)
        for r in range(world):
            rec.rank = r
            multi.mesh_phases_sharded(tb.wf_array, m16, mesh3, start3, list(range(8)), rec, r, world)
        # ONE collective per rank: [its planes' phases | its min gaps]
        assert all(len(rec.parts[r]) == 1 for r in range(world))
        ng = len(g3)
        assert np.array_equal(np.concatenate([rec.parts[r][0][:-ng] for r in range(world)]).reshape(10, 9), ref3)
        assert np.array_equal(np.min([rec.parts[r][0][-ng:] for r in range(world)], axis=0), g3)

    # every rank's chunk of a k list in turn (worlds 3 and 8, 1000 k-points: uneven), chunks of a device-generated mesh too
    class RowRecorder(object):
        def __init__(self):
            self.parts = []
        def allgatherv_rows(self, mine, counts):
            self.parts.append(np.array(mine))
            return np.zeros((mine.shape[0], int(sum(counts))))
    hal = hp.haldane(tb.tb_model, 0.2)
    k = np.random.default_rng(8).random((1000, 2))
    ref_ev = hal.solve_all(k)
    for world in (3, 8):
        rec = RowRecorder()
        for r in range(world):
            multi.solve_all_sharded(hal, k, rec, r, world)
        assert np.array_equal(np.concatenate(rec.parts, axis=1), ref_ev)
    # (the ranged device generator against the host list)
    mesh32 = np.array([6, 5, 7], dtype=np.int32)
    kh = m16.k_uniform_mesh([6, 5, 7])
    kd = C.c_void_p()
    _lib.check(lib.tbk_dev_alloc(ctx.handle, 8 * 3 * 100, C.byref(kd)))
    got = np.zeros((100, 3))
    _lib.check(lib.tbk_k_uniform_mesh_range_dev(ctx.handle, 3, _lib.iptr(mesh32), 57, 100, kd))
    _lib.check(lib.tbk_dev_download(ctx.handle, got.ctypes.data_as(C.c_void_p), kd, got.nbytes))
    _lib.check(lib.tbk_dev_free(ctx.handle, kd))
    assert np.array_equal(got, kh[57:157])


def test_singular_link_matrix_is_reported_not_returned(tb):
    """Wilson-loop eigenphases need the polar factor of every link overlap matrix; a rank-deficient one
    (orthogonal occupied subspaces at neighbouring points) has none.  The workgroup-level pipeline (3+ bands) must
    raise instead of multiplying a non-unitary factor into the loop."""
    m = hp.quiet(tb.tb_model, 1, 1, [[1.0]], [[0.0], [0.2], [0.5], [0.7]])
    m.set_hop(-1.0, 0, 1, [0])
    w = tb.wf_array(m, [4])
    e = np.identity(4, dtype=complex)
    w[0] = e                                   # occupied {e0, e1, e2}
    w[1] = e[[1, 2, 3, 0]]                     # occupied {e1, e2, e3}: overlap with the previous point has rank 2
    w[2] = e
    w[3] = e
    with pytest.raises(tb._lib.TbkError, match="singular"):
        w.berry_phase([0, 1, 2], berry_evals=True, contin=False)
    w[1] = e                                   # repaired: a trivial loop, all eigenphases zero; the sticky flag was cleared
    got = w.berry_phase([0, 1, 2], berry_evals=True, contin=False)
    assert np.max(np.abs(got)) < 1e-12


def test_choose_states_copies_band_planes_on_the_device(tb):
    """pythtb.py:2568-2608 on a resident array: the subset is made of device-to-device plane copies (no PCIe), equals
    np.take on the host copy, and Berry calls on it equal the occ-list form on the parent."""
    m = hp.kane_mele(tb.tb_model, "odd")
    w = tb.wf_array(m, [301, 97])
    w.solve_on_grid([-0.5, -0.5])
    stats(tb, reset=True)
    sub = w.choose_states([1, 0])
    low = w.choose_states([0, 1])
    up = w.choose_states([3, -2])                      # negative indices as np.take allows
    assert stats(tb)["h2d_calls"] == 0 and stats(tb)["d2h_calls"] == 0
    assert sub._nsta_arr == 2 and sub._mesh_arr.tolist() == [301, 97]
    host = w.to_host()
    assert np.array_equal(sub.to_host(), np.take(host, [1, 0], axis=2))
    assert np.array_equal(up.to_host(), np.take(host, [3, -2], axis=2))
    assert np.array_equal(low.berry_phase([0, 1], 0, contin=False, berry_evals=True),
                          w.berry_phase([0, 1], 0, contin=False, berry_evals=True))
    assert abs(low.berry_flux("All") - w.berry_flux([0, 1])) < 1e-12
    e = w.empty_like(nsta_arr=3)
    assert e._wfs.shape == (301, 97, 3, 2, 2) and e._nsta_arr == 3
    with pytest.raises(IndexError):
        w.choose_states([4])


@pytest.mark.gpu
def test_parked_model_blobs_survive_the_first_small_solve():
    """A fresh context: models are uploaded and freed (their small table blobs are parked in the context), THEN the first small
    solve allocates the mapped host buffer, then a parked blob is reused by the next upload.  (The buffer's first allocation once
    freed the parked blobs without emptying the pool: the next upload copied into freed memory.)  C ABI directly, because the
    Python classes always use the default context."""
    import ctypes as C
    import pythtb_amd as tb
    from pythtb_amd import _lib
    lib = _lib.lib
    ctx = _lib.Context()
    m = hp.haldane(tb.tb_model, 0.2)
    ref = m.solve_all(np.array([[0.1, 0.2], [0.3, -0.4]]))          # default context

    def upload():
        orb, onsite, hi, hj, hR, amp = m._flat_tables()
        h = C.c_void_p()
        _lib.check(lib.tbk_model_upload(ctx.handle, m._dim_k, m._norb, m._nspin, _lib.dptr(orb), _lib.dptr(onsite.view(float)), len(hi),
                                        _lib.iptr(hi), _lib.iptr(hj), _lib.iptr(hR.reshape(-1)), _lib.dptr(amp.view(float)), C.byref(h)))
        return h
    for _ in range(3):                                               # three blobs parked, no solve yet
        _lib.check(lib.tbk_model_free(upload()))
    k = np.array([[0.1, 0.2], [0.3, -0.4]])
    for _ in range(4):
        h = upload()                                                 # reuses a parked blob
        ev = np.zeros((2, 2))
        _lib.check(lib.tbk_solve_list(h, _lib.dptr(k), 2, _lib.dptr(ev), None))   # the first one allocates the mapped buffer
        assert np.array_equal(ev, ref)
        _lib.check(lib.tbk_model_free(h))


def test_copies_and_pickles_after_the_per_call_buffers_exist(tb):
    """solve_on_grid / berry_flux keep their small argument buffers and ctypes pointers on the array (round 4): a deepcopy, a
    pickle round trip and choose_states after such calls must still work, and give an array that answers like the original
    -- with TBK_POLL_DONE=0 (hipStreamSynchronize instead of the polled completion word) the same bits."""
    import copy
    import pickle
    from pythtb_amd import _lib
    m = hp.haldane(tb.tb_model)
    w = tb.wf_array(m, [31, 29])
    gaps = w.solve_on_grid([-0.5, -0.5])
    flux = w.berry_flux([0])
    for other in (copy.deepcopy(w), pickle.loads(pickle.dumps(w)), w.choose_states([0, 1])):
        assert other.berry_flux([0]) == flux
        assert np.array_equal(other.solve_on_grid([-0.5, -0.5]), gaps)
        assert other.berry_flux([0]) == flux
    with _lib.knob("TBK_POLL_DONE", 0):
        w2 = tb.wf_array(m, [31, 29])
        assert np.array_equal(w2.solve_on_grid([-0.5, -0.5]), gaps) and w2.berry_flux([0]) == flux
    for _ in range(50):       # many completions in a row: each wait returns THIS call's result
        s = [-0.5 + 0.01 * _, -0.5]
        g = w.solve_on_grid(s)
        f = w.berry_flux([0])
        with _lib.knob("TBK_POLL_DONE", 0):
            assert np.array_equal(w2.solve_on_grid(s), g) and w2.berry_flux([0]) == f


def test_small_calls_polled_completion_equals_stream_synchronisation(tb):
    """Small calls (band-structure paths, solve_one, _gen_ham, Berry phases of small arrays) wait on a completion word that the
    call's last kernel stores in mapped host memory (tbk_done_wait) instead of hipStreamSynchronize: the same bits either way,
    call after call, for every model size that takes the small-list kernel and one that does not."""
    from pythtb_amd import _lib
    rng = np.random.default_rng(4)
    models = [hp.haldane(tb.tb_model), hp.kane_mele(tb.tb_model), hp.random_model(tb.tb_model, 3, 2, 1, seed=3, nhop=9, rmax=1),
              hp.random_model(tb.tb_model, 6, 2, 1, seed=6, nhop=20, rmax=1)]
    for m in models:
        k = rng.random((121, m._dim_k))
        ref = {}
        for poll in (0, 1, 1, 0):
            with _lib.knob("TBK_POLL_DONE", poll):
                got = (m.solve_all(k), m.solve_all(k, eig_vectors=True), m.solve_one(k[7]), m._gen_ham(k[5]),
                       m.solve_one(k[9], eig_vectors=True))
                w = tb.wf_array(m, [13, 11])
                gaps = w.solve_on_grid([0.1, -0.2])
                occ = list(range(max(1, m._nsta // 2)))
                ph = (w.berry_phase(occ, 1, contin=False), w.berry_phase(occ, 0, contin=False, berry_evals=len(occ) > 1),
                      w.berry_flux(occ), w.berry_flux(occ, individual_phases=True))
            flat = [np.asarray(x) for g in got for x in (g if isinstance(g, tuple) else (g,))] + [np.asarray(gaps)] + [np.asarray(p) for p in ph]
            if not ref:
                ref = flat
            else:
                assert len(flat) == len(ref) and all(np.array_equal(a, b) for a, b in zip(flat, ref))


def test_berry_flux_sharded_slabs_equal_the_unsharded_flux(tb):
    """configs[2]'s partitioning (pythtb.py:2475-2497 / :3135-3150) through the library driver
    multi.berry_flux_sharded: a 1-rank RCCL communicator end to end, and worlds 3 and 8 with the ranks run one after another
    on this GPU -- every slab recomputes its halo row (the same bits: plaquette phases equal the unsharded array's bit for
    bit), partial sums add up to the unsharded flux, min gaps min-reduce, one collective per rank."""
    import ctypes as C
    from pythtb_amd import _lib, multi
    lib, ctx = _lib.lib, _lib.default_context()
    hal = hp.haldane(tb.tb_model, 0.0)
    mesh, start = [203, 131], [-0.5, -0.5]
    full = tb.wf_array(hal, mesh)
    g = full.solve_on_grid(start)
    ref_plaq = full.berry_flux([0], individual_phases=True)
    ref_t = full.berry_flux([0], dirs=[1, 0], individual_phases=True)
    ref_tot = full.berry_flux([0])
    assert abs(ref_tot / (2 * np.pi) + 1.0) < 1e-10
    uid = (C.c_ubyte * 128)()
    _lib.check(lib.tbk_comm_unique_id(uid))
    rccl = multi.RcclComm(ctx, bytes(uid), 1, 0)
    try:
        tot, gaps = multi.berry_flux_sharded(tb.wf_array, hal, mesh, start, [0], rccl, 0, 1)
        assert tot == ref_tot and np.array_equal(gaps, g)
        plq, gaps = multi.berry_flux_sharded(tb.wf_array, hal, mesh, start, [0], rccl, 0, 1, individual_phases=True)
        assert np.array_equal(plq, ref_plaq) and np.array_equal(gaps, g)
    finally:
        rccl.close()

    class Recorder(object):
        def __init__(self):
            self.parts = []
        def allgatherv(self, mine, counts):
            self.parts.append(np.array(mine, dtype=float).reshape(-1))
            assert self.parts[-1].size == counts[len(self.parts) - 1]
            return np.zeros(int(sum(counts)))
    for world in (3, 8):
        plans = multi.plan_slabs(mesh[0], world)
        rec = Recorder()
        for r in range(world):
            multi.berry_flux_sharded(tb.wf_array, hal, mesh, start, [0], rec, r, world)
        assert len(rec.parts) == world
        tot, gaps = multi.combine_flux_blocks(np.array(rec.parts), 1)
        assert abs(tot - ref_tot) < 1e-11 and np.array_equal(gaps, g)
        for dirs, ref in (([0, 1], ref_plaq), ([1, 0], ref_t)):
            rec = Recorder()
            for r in range(world):
                multi.berry_flux_sharded(tb.wf_array, hal, mesh, start, [0], rec, r, world, dirs=dirs, individual_phases=True)
            rows = np.concatenate([rec.parts[r][:-1].reshape(plans[r][1] - 1, mesh[1] - 1) for r in range(world)])
            assert np.array_equal(rows if dirs == [0, 1] else rows.T, ref)
