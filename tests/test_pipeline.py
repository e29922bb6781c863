"""bench.py's input-side prefetch (SceneStep.front on a side stream, one batch ahead) against the in-line step: same losses, same weights."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_hip_prefetched_steps_equal_inline_steps(cuda, hip_lib):
    sys.path.insert(0, ROOT)
    import bench
    saved = bench.SCENES_PER_GPU, bench.OBJECTS_PER_GPU
    bench.SCENES_PER_GPU, bench.OBJECTS_PER_GPU = 4, 16
    try:
        points, objects, scene, *_ = bench.make_inputs(0, cuda)
        inputs = (points, objects, scene)
        results = []
        for prefetch in (False, True):
            model = bench.build_model(cuda).train()
            params = [p for p in model.parameters() if p.requires_grad]
            opt = torch.optim.SGD(params, lr=1e-3, momentum=0.9, fused=True)
            pre = bench.Prefetch(model, inputs) if prefetch else None
            losses = []
            for _ in range(4):
                loss = bench.run_step_prefetched(model, opt, params, pre, 1) if prefetch else bench.run_step(model, opt, params, inputs, 1)
                losses.append(float(loss))
            torch.cuda.synchronize()
            results.append((losses, torch.cat([p.detach().reshape(-1) for p in params]).cpu()))
    finally:
        bench.SCENES_PER_GPU, bench.OBJECTS_PER_GPU = saved
    (l0, w0), (l1, w1) = results
    # same kernels on the same data in the same order per stream; only atomics inside a kernel (weight-gradient partial sums, scatter adds)
    # may reorder, exactly as between two in-line runs
    assert l0 == pytest.approx(l1, rel=1e-5)
    torch.testing.assert_close(w1, w0, rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
def test_hip_front_then_compute_equals_forward(cuda, hip_lib):
    sys.path.insert(0, ROOT)
    import bench
    saved = bench.SCENES_PER_GPU, bench.OBJECTS_PER_GPU
    bench.SCENES_PER_GPU, bench.OBJECTS_PER_GPU = 2, 8
    try:
        points, objects, scene, *_ = bench.make_inputs(0, cuda)
        model = bench.build_model(cuda).eval()
        with torch.no_grad():
            a = model(points, objects, scene, 2)["spatial_features"]
            bd = model.front(points, objects, scene, 2)
            assert "spconv_indice_dict" in bd and len(bd["spconv_indice_dict"]) >= 5
            b = model.compute(bd)["spatial_features"]
        # float atomics of the voxel mean accumulate in a run-dependent order: equal up to that, as two in-line forwards are
        assert a.shape == b.shape
        torch.testing.assert_close(b, a, rtol=1e-4, atol=1e-5)
    finally:
        bench.SCENES_PER_GPU, bench.OBJECTS_PER_GPU = saved
