"""A context over several devices (mirt_context_create_multi) through the C ABI.  The GPU box has one GPU, so the
device list names it two, three or four times: the shares of the frame then run side by side on it, through the same
code that drives separate GPUs (one host thread, one scene copy and one workspace per entry, peer copies to the
first entry, sums added there).  The image must not depend on the split: with two samples per pixel bit for bit
(every channel is the sum of at most two terms), else to f32 summation order."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import raytracing_amd as R

pytestmark = pytest.mark.gpu
F = np.float32


class MultiScene:
    """An example scene lowered by the host mirror and uploaded through a context over `devices`."""

    def __init__(self, name, devices, row_bands=0):
        self.L = R.mirt()
        self.X = R.examples()
        self.lowered = self.X.rt_lowered_open(name.encode(), 1, str(R.ASSET_DIR).encode())
        assert self.lowered, self.X.rt_example_last_error()
        self.ctx = C.c_void_p()
        ids = (C.c_int * len(devices))(*devices)
        rc = self.L.mirt_context_create_multi(ids, len(devices), row_bands, C.byref(self.ctx))
        assert rc == 0, self.L.mirt_last_error(None)
        self.scene = C.c_void_p()
        rc = self.X.rt_lowered_upload(self.lowered, self.ctx, C.byref(self.scene))
        assert rc == 0, self.L.mirt_last_error(self.ctx)

    def render(self, cam, bg, w, h, spp, depth, seed=5, first_sample=0, first_row=0, n_rows=0, flags=0):
        p = R.RenderParams(w, h, spp, first_sample, depth, (C.c_float * 3)(*bg), seed, 0, flags, first_row, n_rows)
        rgb = np.zeros((h, w, 3), np.uint8)
        sums = np.zeros((h, w, 3), F)
        st = R.Stats()
        rc = self.L.mirt_render(self.ctx, self.scene, C.byref(cam), C.byref(p), rgb.ctypes.data, sums.ctypes.data,
                                C.byref(st))
        assert rc == 0, self.L.mirt_last_error(self.ctx)
        return rgb, sums, st.as_dict()

    def close(self):
        self.L.mirt_scene_destroy(self.ctx, self.scene)
        self.L.mirt_context_destroy(self.ctx)
        self.X.rt_lowered_close(self.lowered)


@pytest.fixture(scope="module")
def single():
    g = R.ExampleScene("final_scene")
    yield g
    g.close()


@pytest.mark.parametrize("devices,row_bands", [([0, 0], 0), ([0, 0, 0], 1), ([0, 0, 0, 0], 2), ([0, 0], 2)])
def test_shared_render_equals_single_device_render(single, devices, row_bands):
    w, h = 256, 144
    cam, bg = single.camera(w / h)
    m = MultiScene("final_scene", devices, row_bands)
    assert m.L.mirt_context_device_count(m.ctx) == len(devices)
    try:
        # 2 spp: bit for bit, whatever the split (sample ranges 1+1, or whole samples on row bands)
        ref = single.render(w, h, 2, aspect=w / h, seed=5)
        rgb, sums, st = m.render(cam, bg, w, h, 2, single.depth)
        assert st["samples"] == w * h * 2
        assert np.array_equal(sums.view(np.uint32), ref[1].view(np.uint32))
        assert np.array_equal(rgb, ref[0])
        # 13 spp (does not divide evenly), starting at sample 3: f32 summation order only
        ref = single.render(w, h, 13, aspect=w / h, seed=5, first_sample=3)
        rgb, sums, st = m.render(cam, bg, w, h, 13, single.depth, first_sample=3)
        ok = np.isfinite(ref[1])
        assert np.array_equal(ok, np.isfinite(sums))
        np.testing.assert_allclose(sums[ok], ref[1][ok], rtol=3e-6, atol=1e-6)
        assert st["samples"] == w * h * 13 and st["segments"] == 0     # (work counters only on request)
        assert np.mean(rgb != ref[0]) < 1e-4                              # an 8-bit step at a rounding boundary
    finally:
        m.close()


def test_more_devices_than_samples_and_row_window(single):
    w, h = 128, 72
    cam, bg = single.camera(w / h)
    m = MultiScene("final_scene", [0, 0, 0, 0], 1)
    try:
        ref = single.render(w, h, 2, aspect=w / h, seed=5)
        rgb, sums, _ = m.render(cam, bg, w, h, 2, single.depth)          # two of the four shares are empty
        assert np.array_equal(sums.view(np.uint32), ref[1].view(np.uint32))
        # a window of rows through the shared path: rows outside stay zero, rows inside equal the full frame's
        rgb, sums, st = m.render(cam, bg, w, h, 2, single.depth, first_row=16, n_rows=24)
        assert st["samples"] == w * 24 * 2
        assert not sums[:16].any() and not sums[40:].any()
        assert np.array_equal(sums[16:40].view(np.uint32), ref[1][16:40].view(np.uint32))
    finally:
        m.close()


def test_row_window_on_one_device_against_the_oracle(single):
    """first_row / n_rows of mirt_render_params on an ordinary context: the strip is the same strip of the oracle's
    frame (renderer.rs:57-71 renders strips of rows; same camera, same streams)."""
    w, h = 96, 54
    o = O.Scene("final_scene")
    _, osums, _ = o.render(w, h, 4, aspect=w / h, seed=9)
    cam, bg = single.camera(w / h)
    L = R.mirt()
    ctx, scene = single.L.rt_example_context(single.h), single.L.rt_example_scene(single.h)
    p = R.RenderParams(w, h, 4, 0, single.depth, (C.c_float * 3)(*bg), 9, 0, 0, 30, 19)   # 19 rows: not a multiple of 8
    sums = np.zeros((h, w, 3), F)
    rc = L.mirt_render(ctx, scene, C.byref(cam), C.byref(p), None, sums.ctypes.data, None)
    assert rc == 0, L.mirt_last_error(ctx)
    assert not sums[:30].any() and not sums[49:].any()
    ok = np.isfinite(osums[30:49]).all(axis=2)
    close = np.isclose(sums[30:49], osums[30:49], rtol=1e-4, atol=1e-5).all(axis=2)
    assert np.mean(close[ok]) > 0.99
    # out-of-range windows are rejected
    p = R.RenderParams(w, h, 4, 0, single.depth, (C.c_float * 3)(*bg), 9, 0, 0, 50, 19)
    assert L.mirt_render(ctx, scene, C.byref(cam), C.byref(p), None, sums.ctypes.data, None) == 1


def test_create_multi_rejects_bad_lists():
    L = R.mirt()
    ctx = C.c_void_p()
    assert L.mirt_context_create_multi((C.c_int * 2)(0, 0), 2, 3, C.byref(ctx)) == 1       # 3 bands over 2 devices
    assert b"divide" in L.mirt_last_error(None)
    assert L.mirt_context_create_multi((C.c_int * 2)(0, 99), 2, 0, C.byref(ctx)) == 1      # no such device
    assert L.mirt_context_create_multi(None, 0, 0, C.byref(ctx)) == 1
