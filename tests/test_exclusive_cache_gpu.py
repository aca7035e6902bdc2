"""GPU: the EXCLUSIVE-CELL CACHE of the camera-form forward (include/mmt_hip.h `exclusive_cache`, csrc/lift_splat_tile.hip):
a persistent device-side memory, per calibration, of the BEV cells that a single run of the forward reaches -- such runs
are stored instead of added atomically.  Whatever the cache has or has not learnt, the map must be the one the call gives
without it (lss_fpn.py:441-464: the sum over the points of a cell), and the cells (pos_memo) must not change at all."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

VC, VS, VN = [-51.2 + 0.4, -51.2 + 0.4, -1.0], [0.8, 0.8, 8.0], [128, 128, 1]
HDR, META = 64 + 8 * (8 + 8 * 16), 4          # lift_splat_tile.hip: header words (64 + the mailbox of 8 samples), words of slot metadata


def _frustum(final_dim, ds, d_bound):
    from tests.test_oracle_golden import _frustum_torch
    return _frustum_torch(final_dim, ds, d_bound)


def _rig(B, N, W, H, seed, pitch_deg=0.0, jitter=0.05):
    from mm_training_amd import synthetic
    s2e, K = synthetic.camera_rig(B, N, W, H, jitter=jitter, seed=seed)
    c_, s_ = math.cos(math.radians(pitch_deg)), math.sin(math.radians(pitch_deg))
    rx = torch.tensor([[1, 0, 0, 0], [0, c_, -s_, 0], [0, s_, c_, 0], [0, 0, 0, 1]], dtype=torch.float32)
    return s2e.matmul(rx).matmul(torch.inverse(K)).contiguous()


def _cache(N, slots):
    from mm_training_amd.ops.bev_geometry import new_exclusive_cache
    return new_exclusive_cache(N, VN, "cuda", slots)


def _modes(cache, B):
    torch.cuda.synchronize()
    h = cache[:64].tolist()
    return h[24:24 + B], h[8:8 + B]                # what the last call did with sample b: (stage it found, slot)


def _states(cache, N, slots, slot):
    cells = VN[0] * VN[1]
    off = HDR + slots * (META + N * 16) + slot * cells
    return cache[off:off + cells].cpu().numpy()


def _forward(combine, fr, C, excl, **kw):
    from tests.test_camera_form_gpu import _run_forward_cam
    out, pos, _, _ = _run_forward_cam(combine, fr, VC, VS, VN, C=C, excl=excl, **kw)
    return out, pos


def _same(out, ref):
    assert not np.isnan(out).any()
    assert np.abs(out - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), np.abs(out - ref).max()
    # the same cells are touched (per cell, not per element: a channel sum may cancel to exactly 0 in one order of additions only)
    assert np.array_equal((out != 0).any(-1), (ref != 0).any(-1))


@pytest.mark.parametrize("C,bf16,pitch", [(80, False, 0.0), (64, False, 0.0), (80, True, 0.0), (80, False, 3.0)])
def test_cache_learns_in_two_calls_and_never_changes_the_map(mmt_lib, C, bf16, pitch):
    """Claim, MARK, VERIFY, then USE: the header shows what each call did; every call gives the map and the cells of the call
    without a cache; the cells the cache declares single-run do receive their points from one column of one camera.  pitch 3
    degrees: columns that straddle cell borders (bins walked row by row)."""
    from mm_training_amd import _lib
    B, N = 2, 3
    fr = _frustum((128, 352), 16, (2.0, 58.0, 0.5))               # D = 112, fH = 8, fW = 22
    D, fH, fW, _ = fr.shape
    cb = _rig(B, N, 352, 128, seed=3, pitch_deg=pitch)
    ref, ref_pos = _forward(cb, fr, C, None, bf16=bf16)
    assert not _lib.lib().mmt_lss_last_kernel_family(0) & _lib.LSS_FAMILY_EXCLUSIVE
    slots = 64
    cache = _cache(N, slots)
    want = [[0, 0], [1, 1], [2, 2], [3, 3], [3, 3]]
    for call in range(5):
        out, pos = _forward(cb, fr, C, cache, bf16=bf16)
        assert _lib.lib().mmt_lss_last_kernel_family(0) & _lib.LSS_FAMILY_EXCLUSIVE
        modes, slot_of = _modes(cache, B)
        assert modes == want[call], (call, modes)
        assert slot_of[0] != slot_of[1]                  # (direct-mapped by a hash of the matrices: these two do not collide)
        _same(out, ref)
        assert np.array_equal(pos, ref_pos)
    # what was learnt: state > 0 <=> one run; a single run means a single (camera, column) source
    P = N * D * fH * fW
    pm = ref_pos.reshape(B, N, fH, fW, D, 3)                       # pixel-major point order
    for b in range(B):
        st = _states(cache, N, slots, slot_of[b])
        cam_col = (np.arange(N)[:, None, None, None] * fW + np.arange(fW)[None, None, :, None]) + np.zeros((N, fH, fW, D), np.int64)
        kept = pm[b, ..., 0] >= 0
        cell = (pm[b, ..., 1].astype(np.int64) * VN[0] + pm[b, ..., 2])[kept]
        src = cam_col[kept]
        lo = np.full(VN[0] * VN[1], 1 << 40, np.int64)
        hi = np.full(VN[0] * VN[1], -1, np.int64)
        np.minimum.at(lo, cell, src)
        np.maximum.at(hi, cell, src)
        hit = hi >= 0
        single = st > 0
        assert single[hit].sum() > 0.2 * hit.sum(), (single[hit].sum(), hit.sum())        # a good share of the cells is single-run
        assert np.all(lo[hit & single] == hi[hit & single])
        assert np.all(st[hit] != 0)                                                       # every cell that is hit was marked
        assert np.all((st[hit] > 0) | (st[hit] == -1))
    assert P == ref_pos.shape[1]


def test_samples_that_share_their_matrices_and_batches_in_another_order(mmt_lib):
    """Two samples of one rig in a batch learn through the first of them; a later batch with the samples swapped finds both
    calibrations by their matrices (the run ids do not depend on a sample's place in the batch)."""
    B, N, C = 2, 3, 80
    fr = _frustum((128, 352), 16, (2.0, 58.0, 0.5))
    a, b2 = _rig(1, N, 352, 128, seed=11), _rig(1, N, 352, 128, seed=12)
    aa = torch.cat([a, a], 0)
    ref_aa, _ = _forward(aa, fr, C, None)
    cache = _cache(N, 64)
    for want in ([0, 0], [1, 1], [2, 2], [3, 3], [3, 3]):
        out, _ = _forward(aa, fr, C, cache)
        modes, slot_of = _modes(cache, B)
        assert modes == want and slot_of[0] == slot_of[1], (modes, slot_of)
        _same(out, ref_aa)
    ab, ba = torch.cat([a, b2], 0), torch.cat([b2, a], 0)
    ref_ab, _ = _forward(ab, fr, C, None)
    ref_ba, _ = _forward(ba, fr, C, None)
    for batch, ref, want in ((ab, ref_ab, [3, 0]), (ba, ref_ba, [1, 3]), (ab, ref_ab, [3, 2]), (ba, ref_ba, [3, 3]), (ab, ref_ab, [3, 3])):
        out, _ = _forward(batch, fr, C, cache)
        modes, slot_of = _modes(cache, B)
        assert modes == want and slot_of[0] != slot_of[1], (modes, slot_of)
        _same(out, ref)


def test_calibrations_that_share_a_slot_and_a_change_of_shape_or_axes(mmt_lib):
    """The table is direct-mapped: calibrations that fall into one slot (here: a table of one) take it from each other and
    never get past the claim -- and stay correct; one of them alone does learn.  Another depth axis or another grid under the
    same cache start it over instead of using states learnt for other runs."""
    N, C = 3, 80
    fr = _frustum((128, 352), 16, (2.0, 58.0, 0.5))
    rigs = [_rig(1, N, 352, 128, seed=s) for s in (21, 22, 23)]
    refs = [_forward(r, fr, C, None)[0] for r in rigs]
    cache = _cache(N, 1)
    for rnd in range(3):
        for r, ref in zip(rigs, refs):
            out, _ = _forward(r, fr, C, cache)
            assert _modes(cache, 1) == ([0], [0])
            _same(out, ref)
    for want in ([0], [1], [2], [3], [3]):
        out, _ = _forward(rigs[1], fr, C, cache)
        assert _modes(cache, 1)[0] == want
        _same(out, refs[1])
    # another depth axis under the same cache, then back
    two = torch.cat(rigs[:2], 0)
    cache = _cache(N, 64)
    fr2 = _frustum((128, 352), 16, (2.5, 58.5, 0.5))
    ref2, ref2b = _forward(two, fr, C, None)[0], _forward(two, fr2, C, None)[0]
    for frustum, ref in ((fr, ref2), (fr2, ref2b), (fr, ref2)):
        for want in ([0, 0], [1, 1], [2, 2], [3, 3]):
            out, _ = _forward(two, frustum, C, cache)
            assert _modes(cache, 2)[0] == want, (_modes(cache, 2), want)
            _same(out, ref)
    # another batch size (the launch shape may change with it: then the table starts over; the map is right either way)
    out, _ = _forward(rigs[0], fr, C, cache)
    assert _modes(cache, 1)[0] in ([0], [3])
    _same(out, refs[0])


def test_two_way_sets_and_the_call_counters(mmt_lib):
    """Round 4: slots 2s and 2s + 1 form a set.  Two calibrations that meet in one set (a table of two slots = one set) both
    learn and are both used -- the direct-mapped table made them evict each other on every visit; a third one in the same set
    takes a way from them in turn and everything stays correct.  Header words 32..34 count hits / learning calls / misses."""
    N, C = 3, 80
    fr = _frustum((128, 352), 16, (2.0, 58.0, 0.5))
    rigs = [_rig(1, N, 352, 128, seed=s) for s in (41, 42, 43)]
    refs = [_forward(r, fr, C, None)[0] for r in rigs]
    cache = _cache(N, 2)
    seen = {0: [], 1: []}
    for rnd in range(5):
        for i in (0, 1):
            out, _ = _forward(rigs[i], fr, C, cache)
            seen[i].append(_modes(cache, 1)[0][0])
            _same(out, refs[i])
    assert seen[0] == [0, 1, 2, 3, 3] and seen[1] == [0, 1, 2, 3, 3], seen
    hit, learning, miss = cache[32:35].tolist()
    assert (hit, learning, miss) == (4, 4, 2), (hit, learning, miss)
    # a third calibration in the same set: correct throughout, and the two that keep coming back are found again after their
    # way was taken
    for rnd in range(4):
        for i in (0, 1, 2):
            out, _ = _forward(rigs[i], fr, C, cache)
            _same(out, refs[i])
    for want in (None, None, None, 3, 3):
        out, _ = _forward(rigs[2], fr, C, cache)
        _same(out, refs[2])
        if want is not None:
            assert _modes(cache, 1)[0] == [want]
    h2, l2, m2 = cache[32:35].tolist()
    assert h2 + l2 + m2 == 10 + 12 + 5 and m2 > miss


def test_cache_under_graph_replay_and_through_the_module(mmt_lib):
    """A captured forward (zero-fill + select + walk) learns and uses the cache across replays; LSSFPN owns one cache per
    (device, cameras, stream) and reports it in the kernel family."""
    from mm_training_amd import _lib
    from mm_training_amd.ops.bev_geometry import frustum_axes, last_kernel_family, lift_splat_camera
    B, N, C = 2, 3, 80
    fr = _frustum((128, 352), 16, (2.0, 58.0, 0.5))
    D, fH, fW, _ = fr.shape
    cb = _rig(B, N, 352, 128, seed=31).cuda()
    axes = [t.cuda() for t in frustum_axes(fr)]
    g = torch.Generator().manual_seed(0)
    depth = torch.rand(B * N, D, fH, fW, generator=g).softmax(1).cuda()
    ctx = torch.randn(B * N, C, fH, fW, generator=g).cuda()
    ref = lift_splat_camera(cb, axes, depth, ctx, VN, VC, VS).clone()
    cache = _cache(N, 64)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        static_out = lift_splat_camera(cb, axes, depth, ctx, VN, VC, VS, exclusive_cache=cache)      # warm-up outside the capture: the claim
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            static_out = lift_splat_camera(cb, axes, depth, ctx, VN, VC, VS, exclusive_cache=cache)
    torch.cuda.current_stream().wait_stream(s)
    assert "exclusive" in last_kernel_family(detail=True) and "register" in last_kernel_family(detail=True)
    seen = []
    for _ in range(5):
        graph.replay()
        seen.append(_modes(cache, B)[0])
        assert float((static_out - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    assert seen == [[1, 1], [2, 2], [3, 3], [3, 3], [3, 3]], seen
    # new inputs under the same graph
    depth.copy_(torch.rand(B * N, D, fH, fW, generator=g).softmax(1))
    ref2 = lift_splat_camera(cb, axes, depth, ctx, VN, VC, VS).clone()
    graph.replay()
    torch.cuda.synchronize()
    assert float((static_out - ref2).abs().max()) <= 2e-5 * max(1.0, float(ref2.abs().max()))
    assert _lib.lib().mmt_lss_last_kernel_family(0) & _lib.LSS_FAMILY_REGISTER


def test_cache_limits_more_samples_or_cameras_than_it_tracks(mmt_lib):
    """The first 8 samples of a call take part, the rest run without the cache; a rig of more than 8 cameras, a column of
    more than 16 rows or a long ray (the LDS-record walk) ignore it altogether -- the map is right in every case."""
    from mm_training_amd import _lib
    C = 80
    fr = _frustum((64, 176), 16, (2.0, 58.0, 2.0))                 # D = 28, fH = 4, fW = 11
    big = _rig(10, 2, 176, 64, seed=41)                            # 10 samples
    ref, _ = _forward(big, fr, C, None)
    cache = _cache(2, 256)
    for want in (0, 1, 2, 3, 3):
        out, _ = _forward(big, fr, C, cache)
        assert _modes(cache, 8)[0] == [want] * 8
        _same(out, ref)
    wide = _rig(1, 9, 176, 64, seed=42)                            # 9 cameras
    ref, _ = _forward(wide, fr, C, None)
    cache9 = _cache(9, 4)
    for _ in range(4):
        out, _ = _forward(wide, fr, C, cache9)
        assert not _lib.lib().mmt_lss_last_kernel_family(0) & _lib.LSS_FAMILY_EXCLUSIVE
        _same(out, ref)
    assert int(cache9.abs().sum()) == 0                            # never touched
    tall = _frustum((272, 176), 16, (2.0, 58.0, 2.0))              # fH = 17
    rig2 = _rig(2, 2, 176, 272, seed=43)
    ref, _ = _forward(rig2, tall, C, None)
    for _ in range(4):
        out, _ = _forward(rig2, tall, C, cache)
        assert not _lib.lib().mmt_lss_last_kernel_family(0) & (_lib.LSS_FAMILY_EXCLUSIVE | _lib.LSS_FAMILY_REGISTER)
        _same(out, ref)


def test_training_lssfpn_with_and_without_the_cache(mmt_lib, monkeypatch):
    """Through the module (camera form, 64 channels): three rigs in rotation under SGD, so that calibrations are claimed,
    learnt and used while the weights move -- the loss curve is the one of the run without a cache to the noise of atomic
    summation order, and the cache does get used."""
    from mm_training_amd import synthetic
    from mm_training_amd.dp import make_config
    from mm_training_amd.layers.backbones import LSSFPN
    from mm_training_amd.ops.bev_geometry import last_kernel_family
    cfg = make_config("tiny")
    bc = dict(cfg["backbone_conf"], output_channels=64)
    H, W = cfg["final_dim"]
    B, N = 2, cfg["num_cams"]
    rigs = []
    for seed in (0, 1, 2):
        s2e, K = synthetic.camera_rig(B, N, W, H, jitter=0.05, seed=seed)
        rigs.append(dict(sensor2ego_mats=s2e.view(B, 1, N, 4, 4).cuda(), intrin_mats=K.view(B, 1, N, 4, 4).cuda(),
                         bda_mat=torch.eye(4).repeat(B, 1, 1).cuda()))
    imgs = torch.rand(B, 1, N, 3, H, W, generator=torch.Generator().manual_seed(1)).cuda()
    curves, fams = {}, {}
    for slots in ("64", "0"):
        monkeypatch.setenv("MMT_LSS_EXCL_SLOTS", slots)
        torch.manual_seed(0)
        m = LSSFPN(**bc).cuda().train()
        m.plan_form = False                    # (the ray-walk forward is what takes the exclusive-cell cache; the default is the plan form)
        assert m.exclusive_slots == int(slots)
        opt = torch.optim.SGD(m.parameters(), lr=1e-2)
        losses = []
        for step in range(15):
            opt.zero_grad(set_to_none=True)
            loss = m(imgs, rigs[step % 3]).square().mean()
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        curves[slots], fams[slots] = losses, last_kernel_family(detail=True)
        if slots == "64":
            (cache,) = m._excl_caches.values()
            torch.cuda.synchronize()
            assert cache[24:24 + B].tolist() == [3] * B          # the fifth round of each rig: its calibrations are in use
    assert "exclusive" in fams["64"] and "exclusive" not in fams["0"]
    for a, b in zip(curves["64"], curves["0"]):
        assert a == a and abs(a - b) <= 1e-3 * max(1e-6, abs(b)), (curves["64"], curves["0"])
