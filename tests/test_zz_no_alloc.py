"""mirt_render_accumulate_static (the entry point that allocates nothing) against mirt_render_accumulate, through
the C ABI: same scene, camera, seed and samples.  With two samples per pixel every channel of the image is the sum
of at most two terms added to zero, so the order of the device's atomic adds cannot change a bit and the images
must be identical; at 16 samples per pixel the order may differ and the tolerance is a few f32 roundings."""
import ctypes as C

import numpy as np
import pytest
import torch

import raytracing_amd as R

pytestmark = pytest.mark.gpu


def _render(entry, g, width, height, spp, pool=0, flags=0):
    L = R.mirt()
    cam, bg = g.camera(width / height)
    p = R.RenderParams(width, height, spp, 0, g.depth, (C.c_float * 3)(*bg), 11, pool, flags)
    sums = torch.zeros((height, width, 3), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    st = R.Stats()
    ctx, scene = g.L.rt_example_context(g.h), g.L.rt_example_scene(g.h)
    rc = entry(ctx, scene, C.byref(cam), C.byref(p), C.c_void_p(sums.data_ptr()), None, C.byref(st))
    msg = L.mirt_last_error(ctx).decode() if rc else ""
    return rc, msg, sums.cpu().numpy(), st.as_dict()


@pytest.mark.parametrize("scene", ["final_scene", "synthetic_bvh:40"])   # scene served from LDS / from global memory
def test_static_workspace_render_equals_allocating_render(scene):
    L = R.mirt()
    g = R.ExampleScene(scene)
    # two calls in a row on the shared workspace, the second one smaller
    for width, height in ((320, 180), (96, 54)):
        rc, msg, ref, st_ref = _render(L.mirt_render_accumulate, g, width, height, 2)
        assert rc == 0, msg
        rc, msg, got, st = _render(L.mirt_render_accumulate_static, g, width, height, 2)
        assert rc == 0, msg
        assert st["path_pool"] <= L.mirt_static_workspace_paths()
        assert st["samples"] == st_ref["samples"] == width * height * 2
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), "2-spp images differ bitwise"
        assert np.isfinite(ref).all() and ref.max() > 0
    rc, msg, ref, _ = _render(L.mirt_render_accumulate, g, 96, 54, 16)
    assert rc == 0, msg
    rc, msg, got, _ = _render(L.mirt_render_accumulate_static, g, 96, 54, 16)
    assert rc == 0, msg
    np.testing.assert_allclose(got, ref, rtol=2e-6, atol=1e-6)   # 16 adds per pixel in another order
    g.close()


def test_static_workspace_rejects_what_it_cannot_hold():
    L = R.mirt()
    g = R.ExampleScene("cornel_box")
    limit = L.mirt_static_workspace_paths()
    rc, msg, _, _ = _render(L.mirt_render_accumulate_static, g, 64, 64, 1, pool=limit + 1)
    assert rc == 1 and "fixed workspace" in msg                     # MIRT_ERR_INVALID_ARGUMENT
    rc, msg, _, _ = _render(L.mirt_render_accumulate_static, g, 64, 64, 1, flags=R.MIRT_RENDER_TIME_KERNELS)
    assert rc == 1 and "allocation" in msg
    rc, msg, img, st = _render(L.mirt_render_accumulate_static, g, 64, 64, 1, pool=limit)
    assert rc == 0 and st["path_pool"] <= limit, msg
    g.close()


def test_static_workspace_call_leaves_device_memory_alone():
    """No hipMalloc / hipFree inside the call: the device's free memory reads the same before and after (the
    allocating entry point, called first on a fresh pool size, does move it)."""
    L = R.mirt()
    g = R.ExampleScene("final_scene")
    rc, msg, _, _ = _render(L.mirt_render_accumulate_static, g, 64, 36, 1)      # first use on this device done
    assert rc == 0, msg
    cam, bg = g.camera(16 / 9)
    p = R.RenderParams(320, 180, 4, 0, g.depth, (C.c_float * 3)(*bg), 11, 0, 0)
    sums = torch.zeros((180, 320, 3), dtype=torch.float32, device="cuda:0")
    st = R.Stats()
    ctx, scene = g.L.rt_example_context(g.h), g.L.rt_example_scene(g.h)
    torch.cuda.synchronize()
    free_before, _ = torch.cuda.mem_get_info(0)
    rc = L.mirt_render_accumulate_static(ctx, scene, C.byref(cam), C.byref(p), C.c_void_p(sums.data_ptr()), None, C.byref(st))
    free_after, _ = torch.cuda.mem_get_info(0)
    assert rc == 0, L.mirt_last_error(ctx).decode()
    assert free_after == free_before, (free_before, free_after)
    assert float(sums.sum()) > 0
    g.close()


def test_static_workspace_turns_an_overlapping_call_away():
    """Two calls at once on one GPU would share the workspace: the second is rejected, not raced."""
    import threading
    import time
    L = R.mirt()
    a, b = R.ExampleScene("final_scene"), R.ExampleScene("cornel_box")     # two contexts, one GPU
    result = {}

    def long_call():
        # (the short calls of the loop below may hold the workspace at the moment this one arrives: it is then the
        # one turned away, and asks again)
        for _ in range(1000):
            result["a"] = _render(L.mirt_render_accumulate_static, a, 1920, 1080, 16)
            if not (result["a"][0] == 1 and "in use" in result["a"][1]):
                break
            time.sleep(0.001)

    # (the second caller's buffers are made beforehand: nothing in its loop waits for the device)
    cam, bg = b.camera(1.0)
    p = R.RenderParams(32, 32, 1, 0, b.depth, (C.c_float * 3)(*bg), 11, 0, 0)
    sums = torch.zeros((32, 32, 3), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    ctx, scene = b.L.rt_example_context(b.h), b.L.rt_example_scene(b.h)
    t = threading.Thread(target=long_call)
    t.start()
    rejected, accepted = 0, 0
    while t.is_alive():
        rc = L.mirt_render_accumulate_static(ctx, scene, C.byref(cam), C.byref(p), C.c_void_p(sums.data_ptr()), None, None)
        if rc == 1 and "in use" in L.mirt_last_error(ctx).decode():
            rejected += 1
        else:
            assert rc == 0, L.mirt_last_error(ctx).decode()
            accepted += 1
            time.sleep(0.005)      # (a gap for the long call to get in)
    t.join()
    rc, msg, img, _ = result["a"]
    assert rc == 0, msg
    assert np.isfinite(img[np.isfinite(img)]).all() and np.nanmax(img) > 0
    assert rejected > 0, (rejected, accepted)
    rc, msg, _, _ = _render(L.mirt_render_accumulate_static, b, 32, 32, 1)    # and free again afterwards
    assert rc == 0, msg
    a.close()
    b.close()
