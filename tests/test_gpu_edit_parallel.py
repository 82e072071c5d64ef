"""SURVEY §8f-2, the parallel form of octree_update.comp: batches whose paths do not collide are applied one lane per
invocation with prefix-sum cell allocation and must equal the ordered walk (= the oracle = llvmpipe) bit for bit; anything
doubtful must fall back to the ordered walk on the device."""
import numpy as np
import pytest

import oracle_py
from octree_util import distinct_deltas, edit_setup as setup
from tdt4230_project_raytracing_amd import host, rt

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg,n", [(2, 1000), (3, 4096), (1, 40)])
def test_non_colliding_batch_runs_in_parallel_and_equals_the_ordered_walk(oracle, cfg, n):
    scene = host.Scene.config(cfg)
    used = scene.counts["cells"]
    depth = scene.max_depth
    scene.blobs[0] = np.concatenate([scene.blobs[0], np.zeros(16 * (n * depth + 8), np.uint32)])   # a free pool to allocate from
    rng = np.random.default_rng(7 + cfg)
    delta = distinct_deltas(rng, n, depth, scene.blobs[0])
    assert len(delta) >= min(n, 12)
    n = len(delta)
    want_cells, want_counter = oracle_py.oracle_octree_update(oracle, scene, delta, used, (n, 1, 1))
    r, upd, counter = setup(scene, used, delta)
    try:
        upd.dispatch_compute(n, 1, 1)
        assert r.ctx.last_edit_path() == 2, "a collision-free batch must take the parallel path"
        assert int(counter.read(np.uint32)[0]) == want_counter
        assert np.array_equal(r.vbos[0].read(np.uint32), want_cells)
        # and the ordered walk on a fresh copy gives the same bytes (mode 1 forces it)
        r2, upd2, counter2 = setup(scene, used, delta)
        r2.ctx.edit_mode(1)
        upd2.dispatch_compute(n, 1, 1)
        assert r2.ctx.last_edit_path() == 1
        assert np.array_equal(r2.vbos[0].read(np.uint32), want_cells) and int(counter2.read(np.uint32)[0]) == want_counter
        r2.close()
        # the trace sees the edited tree (LDS-table image rebuilt)
        scene.blobs[0] = want_cells
        assert (r.render().view(np.uint32) == oracle.render(scene, r.camera, threads=4).view(np.uint32)).all()
    finally:
        r.close()


@pytest.mark.parametrize("case", ["same_cell", "same_delta_twice", "dirty_pool", "out_of_room", "piled_up"])
def test_doubtful_batches_fall_back_to_the_ordered_walk(oracle, case):
    scene = host.Scene.config(2)
    used = scene.counts["cells"]
    depth = scene.max_depth
    rng = np.random.default_rng(5)
    n = 64
    delta = distinct_deltas(rng, n, depth, scene.blobs[0] if case in ("dirty_pool", "out_of_room") else None)
    assert len(delta) == n
    dispatch = (n, 1, 1)
    pool = 16 * (n * depth + 8)
    if case == "same_cell":
        delta[17, :3] = delta[3, :3]                     # two invocations end on one node
    elif case == "same_delta_twice":
        dispatch = (8, 8, 1)                             # delta_index = x + y: most deltas are applied several times
    elif case == "out_of_room":
        pool = 16 * 3                                    # the walk leaves the buffer
    elif case == "piled_up":
        delta[:, :3] = delta[0, :3] + rng.uniform(-1e-3, 1e-3, size=(n, 3)).astype(np.float32)
    scene.blobs[0] = np.concatenate([scene.blobs[0], np.zeros(pool, np.uint32)])
    if case == "dirty_pool":
        scene.blobs[0][16 * used + 1::2] = 1             # the free pool is not EMPTY: every node claims to be a PARENT (of cell 0)
    want_cells, want_counter = oracle_py.oracle_octree_update(oracle, scene, delta, used, dispatch)
    r, upd, counter = setup(scene, used, delta)
    try:
        upd.dispatch_compute(*dispatch)
        path = r.ctx.last_edit_path()
        assert np.array_equal(r.vbos[0].read(np.uint32), want_cells)
        assert int(counter.read(np.uint32)[0]) == want_counter
        assert path == 1, f"{case}: expected the ordered walk"
    finally:
        r.close()


def test_single_invocation_goes_straight_to_the_ordered_walk():
    scene = host.Scene.demo()
    delta = np.zeros((1, 8), np.float32)
    delta[0, :5] = [0.3, 0.6, 0.2, 2.0, 5.0]
    r, upd, counter = setup(scene, 19, delta)
    try:
        upd.dispatch_compute(0, 1, 0)                    # main.rs:568 -> octree.rs:179
        assert r.ctx.last_edit_path() == 1
    finally:
        r.close()
