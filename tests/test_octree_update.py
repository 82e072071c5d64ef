"""Next row §8f-2: the voxel edit kernel (assets/shaders/octree_update.comp) — oracle and HIP kernel
against what the reference's own shader did on llvmpipe (tests/golden/edit_*.npz: the changed node
dwords, the atomic counter, and a render of the edited tree)."""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_py
from tdt4230_project_raytracing_amd import host, rt

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "edit_*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    scene = host.Scene.config(meta["scene"][1])
    pad = meta["pad"]
    if pad > 0:
        scene.blobs[0] = np.concatenate([scene.blobs[0], np.zeros(16 * pad, np.uint32)])
    elif pad < 0:
        scene.blobs[0] = scene.blobs[0][:pad].copy()
    assert hashlib.sha256(scene.blobs[0].tobytes()).hexdigest() == meta["scene_sha256_before"]
    expect = scene.blobs[0].copy()
    expect[z["changed_index"]] = z["changed_value"]
    return z, meta, scene, expect


def test_fixtures_present():
    assert len(CASES) == 4


@pytest.mark.parametrize("name", CASES)
def test_oracle_edit_equals_reference(oracle, name):
    z, meta, scene, expect = load(name)
    cells, cnt = oracle_py.oracle_octree_update(oracle, scene, z["delta"], meta["counter_before"], meta["dispatch"])
    assert cnt == meta["counter_after"]
    assert np.array_equal(cells, expect)
    scene.blobs[0] = cells
    img = oracle.render(scene, host.camera_reference_pose(128, 96, 2, 6), threads=4)
    assert (img.view(np.uint32) == z["image"].view(np.uint32)).all()        # the trace on the edited tree


def test_oracle_edit_vs_reference_shader_live(oracle, glref):
    rng = np.random.default_rng(11)
    scene = host.Scene.config(1)
    scene.blobs[0] = np.concatenate([scene.blobs[0], np.zeros(16 * 32, np.uint32)])
    counter = 70
    for _ in range(6):                                                        # edits pile up on one tree
        delta = np.zeros((1, 8), np.float32)
        delta[0, :3] = rng.random(3)
        delta[0, 3], delta[0, 4] = float(rng.integers(0, 3)), float(rng.integers(0, 4))
        ref = oracle_py.glref_octree_update(scene, delta, counter, (0, 1, 0))
        got = oracle_py.oracle_octree_update(oracle, scene, delta, counter, (0, 1, 0))
        assert got[1] == ref[1] and np.array_equal(got[0], ref[0])
        scene.blobs[0], counter = ref


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_edit_equals_reference(oracle, name):
    """Through the C ABI exactly as octree.rs does it: init_global_buffers' bindings, update_vbo, then a frame."""
    z, meta, scene, expect = load(name)
    cam = host.camera_reference_pose(128, 96, 2, 6)
    r = rt.Renderer(scene, cam)                                               # raytracer program + scene bindings
    try:
        before = r.render()
        assert (before.view(np.uint32) == oracle.render(scene, cam, threads=4).view(np.uint32)).all()
        upd = rt.ComputeShader(r.ctx, rt.PROGRAM_OCTREE_UPDATE)
        assert upd.group_size == [1, 1, 1]
        counter = rt.VertexBufferObject(r.ctx, np.array([meta["counter_before"]], np.int32))       # octree.rs:105-117
        r.ctx.bind_buffer_base(rt.ATOMIC_COUNTER_BUFFER, 0, counter)
        delta_vbo = rt.VertexBufferObject(r.ctx, np.zeros(1000, np.float32))                        # octree.rs:124-144
        r.ctx.bind_buffer_base(rt.SHADER_STORAGE_BUFFER, 5, delta_vbo)
        delta_vbo.sub_data(0, z["delta"])
        upd.dispatch_compute(*meta["dispatch"])
        assert int(counter.read(np.uint32)[0]) == meta["counter_after"]
        assert np.array_equal(r.vbos[0].read(np.uint32), expect)
        after = r.render()                                                    # LDS-table image must have been rebuilt
        assert (after.view(np.uint32) == z["image"].view(np.uint32)).all()
        if z["changed_index"].size and name != "edit_demo_out_of_room":
            assert (after != before).any()
    finally:
        r.close()


@pytest.mark.gpu
def test_update_vbo_dispatch_shape(oracle):
    """Octree::update_vbo(delta, 5, ..) (main.rs:568): 20 bytes of BufferSubData and a (0, 1, 0) dispatch = one invocation."""
    scene = host.Scene.demo()
    cam = host.camera_reference_pose(64, 64, 1, 2)
    r = rt.Renderer(scene, cam)
    try:
        upd = rt.ComputeShader(r.ctx, rt.PROGRAM_OCTREE_UPDATE)
        counter = rt.VertexBufferObject(r.ctx, np.array([19], np.int32))
        r.ctx.bind_buffer_base(rt.ATOMIC_COUNTER_BUFFER, 0, counter)
        delta_vbo = rt.VertexBufferObject(r.ctx, np.zeros(1000, np.float32))
        r.ctx.bind_buffer_base(rt.SHADER_STORAGE_BUFFER, 5, delta_vbo)
        delta = np.zeros(500, np.float32)
        delta[:5] = [0.3, 0.6, 0.2, 2.0, 5.0]
        rt.update_vbo(r.ctx, delta_vbo, delta, 5, upd)
        d8 = np.zeros((1, 8), np.float32)
        d8[0, :5] = delta[:5]
        cells, cnt = oracle_py.oracle_octree_update(oracle, scene, d8, 19, (0, 1, 0))
        assert int(counter.read(np.uint32)[0]) == cnt and np.array_equal(r.vbos[0].read(np.uint32), cells)
        with pytest.raises(rt.TdtError):
            upd.program.set_i32("camera.image_width", 3)                      # the edit program has no uniforms
    finally:
        r.close()
