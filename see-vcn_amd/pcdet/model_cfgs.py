"""Model-section dictionaries of the reference YAMLs this path is measured on (values copied as data from
detector3d/tools/cfgs/kitti_models/second.yaml:7-86); used by tests, the golden generators and bench.py."""

SECOND_BACKBONE_2D = dict(NAME='BaseBEVBackbone', LAYER_NUMS=[5, 5], LAYER_STRIDES=[1, 2], NUM_FILTERS=[128, 256],
                          UPSAMPLE_STRIDES=[1, 2], NUM_UPSAMPLE_FILTERS=[256, 256])


def _anchor(name, size, bottom, matched, unmatched):
    return dict(class_name=name, anchor_sizes=[size], anchor_rotations=[0, 1.57], anchor_bottom_heights=[bottom], align_center=False,
                feature_map_stride=8, matched_threshold=matched, unmatched_threshold=unmatched)


SECOND_DENSE_HEAD = dict(
    NAME='AnchorHeadSingle', CLASS_AGNOSTIC=False, USE_DIRECTION_CLASSIFIER=True, DIR_OFFSET=0.78539, DIR_LIMIT_OFFSET=0.0, NUM_DIR_BINS=2,
    ANCHOR_GENERATOR_CONFIG=[_anchor('Car', [3.9, 1.6, 1.56], -1.78, 0.6, 0.45),
                             _anchor('Pedestrian', [0.8, 0.6, 1.73], -0.6, 0.5, 0.35),
                             _anchor('Cyclist', [1.76, 0.6, 1.73], -0.6, 0.5, 0.35)],
    TARGET_ASSIGNER_CONFIG=dict(NAME='AxisAlignedTargetAssigner', POS_FRACTION=-1.0, SAMPLE_SIZE=512, NORM_BY_NUM_EXAMPLES=False,
                                MATCH_HEIGHT=False, BOX_CODER='ResidualCoder'),
    LOSS_CONFIG=dict(LOSS_WEIGHTS=dict(cls_weight=1.0, loc_weight=2.0, dir_weight=0.2, code_weights=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0])))

CLASS_NAMES = ['Car', 'Pedestrian', 'Cyclist']
KITTI_RANGE = [0, -40, -3, 70.4, 40, 1]
