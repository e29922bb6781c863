"""CenterHead (CenterPoint dense head): heat-map / regression target assignment (one HIP launch), losses and top-K decode + NMS
against the goldens produced by the reference's own CenterHead on CPU (tests/golden/make_center_golden.py)."""
import os

import numpy as np
import pytest
import torch

from center_inputs import GRID, RANGE, VOXEL, make_inputs
from oracle import heads as oh
from seeding import seeded_state_dict
from seevcn_amd.pcdet import model_cfgs as C

TCFG = C.CENTER_HEAD['TARGET_ASSIGNER_CONFIG']
KW = dict(num_max_objs=TCFG['NUM_MAX_OBJS'], gaussian_overlap=TCFG['GAUSSIAN_OVERLAP'], min_radius=TCFG['MIN_RADIUS'])


def _head():
    from seevcn_amd.pcdet.models import dense_heads
    head = dense_heads.__all__['CenterHead'](model_cfg=C.CENTER_HEAD, input_channels=24, num_class=10, class_names=C.NUSC_CLASS_NAMES,
                                             grid_size=np.array(GRID), point_cloud_range=np.array(RANGE, np.float32), voxel_size=VOXEL,
                                             predict_boxes_when_training=False)
    head.load_state_dict(seeded_state_dict(head, seed=8))
    return head


def test_oracle_center_targets_match_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "center_head.npz"))
    t = oh.center_assign_targets(make_inputs()['gt_boxes'], C.NUSC_CLASS_NAMES, C.CENTER_HEAD['CLASS_NAMES_EACH_HEAD'], (32, 32), 8, RANGE, VOXEL, **KW)
    for h in range(6):
        assert np.array_equal(t['masks'][h], g[f'masks_{h}']) and np.array_equal(t['inds'][h][:, :40], g[f'inds_{h}'])
        np.testing.assert_allclose(t['heatmaps'][h], g[f'heatmap_{h}'], rtol=0, atol=1e-6)
        np.testing.assert_allclose(t['target_boxes'][h][:, :40], g[f'target_boxes_{h}'], rtol=1e-5, atol=1e-5)
    assert sum(int(g[f'masks_{h}'].sum()) for h in range(6)) == 22 + 27 - 1          # every box but the degenerate one lands in a head


def test_center_head_state_dict_keys():
    sd = _head().state_dict()
    assert 'shared_conv.0.bias' in sd and 'heads_list.5.hm.1.bias' in sd and 'heads_list.0.center.0.0.weight' in sd
    assert sd['heads_list.1.hm.1.weight'].shape == (2, 64, 3, 3) and sd['heads_list.0.vel.1.weight'].shape == (2, 64, 3, 3)


@pytest.mark.gpu
def test_hip_center_targets_bitexact_vs_oracle_and_golden(golden_dir, cuda, hip_lib):
    g = np.load(os.path.join(golden_dir, "center_head.npz"))
    head = _head().to(cuda)
    gt = make_inputs()['gt_boxes']
    td = head.assign_targets(torch.from_numpy(gt).to(cuda), feature_map_size=(32, 32))
    t = oh.center_assign_targets(gt, C.NUSC_CLASS_NAMES, C.CENTER_HEAD['CLASS_NAMES_EACH_HEAD'], (32, 32), 8, RANGE, VOXEL, **KW)
    for h in range(6):
        assert np.array_equal(td['masks'][h].cpu().numpy(), g[f'masks_{h}'])
        assert np.array_equal(td['inds'][h].cpu().numpy(), t['inds'][h])
        assert np.array_equal(td['inds'][h].cpu().numpy()[:, :40], g[f'inds_{h}'])
        np.testing.assert_allclose(td['heatmaps'][h].cpu().numpy(), g[f'heatmap_{h}'], rtol=0, atol=1e-6)
        np.testing.assert_allclose(td['target_boxes'][h].cpu().numpy(), t['target_boxes'][h], rtol=1e-5, atol=1e-5)
    # ragged: a scene with no boxes and a larger map
    rng = np.random.default_rng(5)
    gt2 = np.zeros((3, 600, 10), np.float32)
    for b in (0, 2):
        n = 600 if b == 0 else 17                                                        # 600 > NUM_MAX_OBJS for one head when all cars
        gt2[b, :n, 0:2] = rng.uniform(-12.8, 12.8, (n, 2))
        gt2[b, :n, 3:6] = rng.uniform(0.3, 5, (n, 3))
        gt2[b, :n, 6] = rng.uniform(-3, 3, n)
        gt2[b, :n, 9] = 1 if b == 0 else rng.integers(1, 11, n)
    td = head.assign_targets(torch.from_numpy(gt2).to(cuda), feature_map_size=(32, 32))
    t = oh.center_assign_targets(gt2, C.NUSC_CLASS_NAMES, C.CENTER_HEAD['CLASS_NAMES_EACH_HEAD'], (32, 32), 8, RANGE, VOXEL, **KW)
    for h in range(6):
        assert np.array_equal(td['masks'][h].cpu().numpy(), t['masks'][h]) and np.array_equal(td['inds'][h].cpu().numpy(), t['inds'][h])
        np.testing.assert_allclose(td['heatmaps'][h].cpu().numpy(), t['heatmaps'][h], rtol=0, atol=1e-6)
        np.testing.assert_allclose(td['target_boxes'][h].cpu().numpy(), t['target_boxes'][h], rtol=1e-5, atol=1e-5)
    assert int(td['masks'][0][0].sum()) == 500 and int(td['masks'][0][1].sum()) == 0


@pytest.mark.gpu
def test_hip_center_head_losses_and_decode_match_reference_golden(golden_dir, cuda, hip_lib):
    g = np.load(os.path.join(golden_dir, "center_head.npz"))
    inp = make_inputs()
    head = _head().to(cuda).train()
    head({'spatial_features_2d': torch.from_numpy(inp['feat']).to(cuda), 'gt_boxes': torch.from_numpy(inp['gt_boxes']).to(cuda), 'batch_size': 2})
    loss, tb = head.get_loss()
    for k, v in tb.items():
        assert abs(v - float(g[k])) <= 1e-3 * abs(float(g[k])), (k, v, float(g[k]))   # 1e-3 relative (north_star float tolerance)
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in head.parameters())
    head.eval()
    with torch.no_grad():
        dd = head({'spatial_features_2d': torch.from_numpy(inp['feat']).to(cuda), 'batch_size': 2})
    # GPU vs CPU convolutions differ in the last bits, so near-tied scores may swap rank / flip an NMS decision: compare as sets,
    # every golden detection must have a same-label twin within the float tolerance (>= 99 % matched, same count +-1 %).
    for k in range(2):
        fb = dd['final_box_dicts'][k]
        pb, ps, pl = fb['pred_boxes'].cpu().numpy(), fb['pred_scores'].cpu().numpy(), fb['pred_labels'].cpu().numpy()
        gb, gs, gl = g[f'pred_boxes_{k}'], g[f'pred_scores_{k}'], g[f'pred_labels_{k}']
        assert abs(len(pb) - len(gb)) <= max(1, len(gb) // 100)
        hit = 0
        for i in range(len(gb)):
            d = np.abs(pb - gb[i]).max(1) + np.abs(ps - gs[i]) + (pl != gl[i]) * 1e3
            hit += d.min() < 2e-3
        assert hit >= 0.99 * len(gb), (hit, len(gb))


@pytest.mark.gpu
def test_hip_center_head_merged_branches_equal_the_separate_ones(cuda, hip_lib):
    """merged_branches (one 64 -> 36 x 64 convolution, one BatchNorm, one block-diagonal last convolution) against the 36 branch modules run one by
    one on the same weights: every prediction map, every parameter gradient and the BatchNorm running statistics -- the same sums in another
    order, held to 1e-4 of each tensor's scale; eval mode too; a head with a hook stands the merged form down."""
    import copy
    from seevcn_amd.pcdet.models.dense_heads import center_head as ch
    inp = make_inputs()
    feat = torch.from_numpy(inp['feat']).to(cuda)
    a = _head().to(cuda).train()
    with torch.no_grad():
        for m in a.modules():                                   # non-trivial BatchNorm parameters and statistics
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5), m.bias.uniform_(-0.3, 0.3), m.running_mean.uniform_(-0.2, 0.2), m.running_var.uniform_(0.5, 2.0)
    b = copy.deepcopy(a)

    def run(head, merged, train):
        head.train(train)
        saved, ch.MERGE_BRANCHES = ch.MERGE_BRANCHES, merged
        try:
            x = head.shared_conv(feat)
            preds = ch.merged_branches(head.heads_list, x) if merged else [h(x) for h in head.heads_list]
        finally:
            ch.MERGE_BRANCHES = saved
        assert preds is not None
        return preds

    pa, pb = run(a, True, True), run(b, False, True)
    gen = torch.Generator().manual_seed(3)
    loss_a = loss_b = 0.0
    for da, db in zip(pa, pb):
        assert list(da) == list(db)
        for k in da:
            assert da[k].shape == db[k].shape
            scale = float(db[k].abs().max())
            assert float((da[k] - db[k]).abs().max()) <= 1e-4 * scale, (k, float((da[k] - db[k]).abs().max()), scale)
            w = torch.randn(da[k].shape, generator=gen).to(cuda)
            loss_a = loss_a + (da[k] * w).sum()
            loss_b = loss_b + (db[k] * w).sum()
    loss_a.backward(), loss_b.backward()
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        if q.grad is None:
            assert p.grad is None, n
            continue
        # a conv bias in front of a batch-statistics BatchNorm has gradient 0: what is left of it is rounding noise on either side (absolute floor)
        scale = float(q.grad.abs().max())
        assert float((p.grad - q.grad).abs().max()) <= 2e-4 * scale + 2e-4, (n, float((p.grad - q.grad).abs().max()), scale)
    for (n, u), (_, v) in zip(a.named_buffers(), b.named_buffers()):
        assert torch.allclose(u.float(), v.float(), rtol=1e-5, atol=1e-6), n
    assert int(a.heads_list[3].dim[0][1].num_batches_tracked) == 1
    with torch.no_grad():
        for da, db in zip(run(a, True, False), run(b, False, False)):
            for k in da:
                assert float((da[k] - db[k]).abs().max()) <= 1e-4 * float(db[k].abs().max()), k
    handle = a.heads_list[2].rot[1].register_forward_hook(lambda *args: None)
    assert ch.merged_branches(a.heads_list, a.shared_conv(feat)) is None
    handle.remove()
