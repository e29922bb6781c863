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

SECOND_POST_PROCESSING = dict(RECALL_THRESH_LIST=[0.3, 0.5, 0.7], SCORE_THRESH=0.1, OUTPUT_RAW_SCORE=False, EVAL_METRIC='kitti',
                              NMS_CONFIG=dict(MULTI_CLASSES_NMS=False, NMS_TYPE='nms_gpu', NMS_THRESH=0.01, NMS_PRE_MAXSIZE=4096,
                                              NMS_POST_MAXSIZE=500))


def second_model_cfg(dynamic_vfe=True):
    """MODEL section of kitti_models/second.yaml; dynamic_vfe swaps MeanVFE (hard voxels from the dataloader) for DynMeanVFE."""
    return dict(NAME='SECONDNet', VFE=dict(NAME='DynMeanVFE' if dynamic_vfe else 'MeanVFE'), BACKBONE_3D=dict(NAME='VoxelBackBone8x'),
                MAP_TO_BEV=dict(NAME='HeightCompression', NUM_BEV_FEATURES=256), BACKBONE_2D=SECOND_BACKBONE_2D,
                DENSE_HEAD=SECOND_DENSE_HEAD, POST_PROCESSING=SECOND_POST_PROCESSING)


class SyntheticDatasetInfo:
    """The attributes Detector3DTemplate.build_networks reads from a dataset (detector3d_template.py:36-44)."""

    def __init__(self, class_names=CLASS_NAMES, point_cloud_range=KITTI_RANGE, voxel_size=(0.05, 0.05, 0.1), num_point_features=3):
        import numpy as np
        from types import SimpleNamespace
        self.class_names = list(class_names)
        self.point_cloud_range = np.array(point_cloud_range, np.float32)
        self.voxel_size = list(voxel_size)
        self.grid_size = np.round((self.point_cloud_range[3:6] - self.point_cloud_range[0:3]) / np.array(voxel_size)).astype(np.int64)
        self.point_feature_encoder = SimpleNamespace(num_point_features=num_point_features)
        self.depth_downsample_factor = None

# detector3d/tools/cfgs/kitti_models/pointpillar.yaml:5-60 (values as data)
PP_RANGE = [0, -39.68, -3, 69.12, 39.68, 1]
PP_VOXEL = dict(VOXEL_SIZE=[0.16, 0.16, 4], MAX_POINTS_PER_VOXEL=32, MAX_NUMBER_OF_VOXELS={'train': 16000, 'test': 40000})
PP_VFE = dict(NAME='PillarVFE', WITH_DISTANCE=False, USE_ABSLOTE_XYZ=True, USE_NORM=True, NUM_FILTERS=[64])
PP_MAP_TO_BEV = dict(NAME='PointPillarScatter', NUM_BEV_FEATURES=64)
PP_BACKBONE_2D = dict(NAME='BaseBEVBackbone', LAYER_NUMS=[3, 5, 5], LAYER_STRIDES=[2, 2, 2], NUM_FILTERS=[64, 128, 256],
                      UPSAMPLE_STRIDES=[1, 2, 4], NUM_UPSAMPLE_FILTERS=[128, 128, 128])


PP_DENSE_HEAD = dict(SECOND_DENSE_HEAD, ANCHOR_GENERATOR_CONFIG=[dict(a, feature_map_stride=2) for a in SECOND_DENSE_HEAD['ANCHOR_GENERATOR_CONFIG']])


def pointpillar_model_cfg():
    """MODEL section of kitti_models/pointpillar.yaml:48-140 (values as data)."""
    return dict(NAME='PointPillar', VFE=PP_VFE, MAP_TO_BEV=PP_MAP_TO_BEV, BACKBONE_2D=PP_BACKBONE_2D, DENSE_HEAD=PP_DENSE_HEAD,
                POST_PROCESSING=SECOND_POST_PROCESSING)


def _sa(mlps, radii, nsample, factor=None):
    d = dict(MLPS=[list(m) for m in mlps], POOL_RADIUS=list(radii), NSAMPLE=list(nsample))
    if factor is not None:
        d['DOWNSAMPLE_FACTOR'] = factor
    return d


def pvrcnn_cfg(num_keypoints=2048, features_source=('bev', 'x_conv1', 'x_conv2', 'x_conv3', 'x_conv4', 'raw_points'), roi_per_image=128,
               nms_post_train=512, nms_pre_train=9000, dp_ratio=0.3):
    """PFE / POINT_HEAD / ROI_HEAD sections of detector3d/tools/cfgs/kitti_models/pv_rcnn.yaml:113-219 (values as data)."""
    pfe = dict(NAME='VoxelSetAbstraction', POINT_SOURCE='raw_points', NUM_KEYPOINTS=num_keypoints, NUM_OUTPUT_FEATURES=128, SAMPLE_METHOD='FPS',
               FEATURES_SOURCE=list(features_source),
               SA_LAYER=dict(raw_points=_sa([[16, 16], [16, 16]], [0.4, 0.8], [16, 16]),
                             x_conv1=_sa([[16, 16], [16, 16]], [0.4, 0.8], [16, 16], 1),
                             x_conv2=_sa([[32, 32], [32, 32]], [0.8, 1.2], [16, 32], 2),
                             x_conv3=_sa([[64, 64], [64, 64]], [1.2, 2.4], [16, 32], 4),
                             x_conv4=_sa([[64, 64], [64, 64]], [2.4, 4.8], [16, 32], 8)))
    point_head = dict(NAME='PointHeadSimple', CLS_FC=[256, 256], CLASS_AGNOSTIC=True, USE_POINT_FEATURES_BEFORE_FUSION=True,
                      TARGET_CONFIG=dict(GT_EXTRA_WIDTH=[0.2, 0.2, 0.2]), LOSS_CONFIG=dict(LOSS_REG='smooth-l1', LOSS_WEIGHTS=dict(point_cls_weight=1.0)))
    roi_head = dict(
        NAME='PVRCNNHead', CLASS_AGNOSTIC=True, SHARED_FC=[256, 256], CLS_FC=[256, 256], REG_FC=[256, 256], DP_RATIO=dp_ratio,
        NMS_CONFIG=dict(TRAIN=dict(NMS_TYPE='nms_gpu', MULTI_CLASSES_NMS=False, NMS_PRE_MAXSIZE=nms_pre_train, NMS_POST_MAXSIZE=nms_post_train, NMS_THRESH=0.8),
                        TEST=dict(NMS_TYPE='nms_gpu', MULTI_CLASSES_NMS=False, NMS_PRE_MAXSIZE=1024, NMS_POST_MAXSIZE=100, NMS_THRESH=0.7)),
        ROI_GRID_POOL=dict(GRID_SIZE=6, MLPS=[[64, 64], [64, 64]], POOL_RADIUS=[0.8, 1.6], NSAMPLE=[16, 16], POOL_METHOD='max_pool'),
        TARGET_CONFIG=dict(BOX_CODER='ResidualCoder', ROI_PER_IMAGE=roi_per_image, FG_RATIO=0.5, SAMPLE_ROI_BY_EACH_CLASS=True, CLS_SCORE_TYPE='roi_iou',
                           CLS_FG_THRESH=0.75, CLS_BG_THRESH=0.25, CLS_BG_THRESH_LO=0.1, HARD_BG_RATIO=0.8, REG_FG_THRESH=0.55),
        LOSS_CONFIG=dict(CLS_LOSS='BinaryCrossEntropy', REG_LOSS='smooth-l1', CORNER_LOSS_REGULARIZATION=True,
                         LOSS_WEIGHTS=dict(rcnn_cls_weight=1.0, rcnn_reg_weight=1.0, rcnn_corner_weight=1.0, code_weights=[1.0] * 7)))
    return pfe, point_head, roi_head


def pvrcnn_model_cfg(**kw):
    pfe, point_head, roi_head = pvrcnn_cfg(**kw)
    pp = dict(SECOND_POST_PROCESSING, NMS_CONFIG=dict(SECOND_POST_PROCESSING['NMS_CONFIG'], NMS_THRESH=0.1))
    return dict(NAME='PVRCNN', VFE=dict(NAME='DynMeanVFE'), BACKBONE_3D=dict(NAME='VoxelBackBone8x'),
                MAP_TO_BEV=dict(NAME='HeightCompression', NUM_BEV_FEATURES=256), BACKBONE_2D=SECOND_BACKBONE_2D, DENSE_HEAD=SECOND_DENSE_HEAD,
                PFE=pfe, POINT_HEAD=point_head, ROI_HEAD=roi_head, POST_PROCESSING=pp)

# detector3d/tools/cfgs/nuscenes_models/cbgs_voxel0075_res3d_centerpoint.yaml:1-140 (values as data)
NUSC_CLASS_NAMES = ['car', 'truck', 'construction_vehicle', 'bus', 'trailer', 'barrier', 'motorcycle', 'bicycle', 'pedestrian', 'traffic_cone']
CENTER_HEAD = dict(
    NAME='CenterHead', CLASS_AGNOSTIC=False,
    CLASS_NAMES_EACH_HEAD=[['car'], ['truck', 'construction_vehicle'], ['bus', 'trailer'], ['barrier'], ['motorcycle', 'bicycle'],
                           ['pedestrian', 'traffic_cone']],
    SHARED_CONV_CHANNEL=64, USE_BIAS_BEFORE_NORM=True, NUM_HM_CONV=2,
    SEPARATE_HEAD_CFG=dict(HEAD_ORDER=['center', 'center_z', 'dim', 'rot', 'vel'],
                           HEAD_DICT=dict(center=dict(out_channels=2, num_conv=2), center_z=dict(out_channels=1, num_conv=2),
                                          dim=dict(out_channels=3, num_conv=2), rot=dict(out_channels=2, num_conv=2), vel=dict(out_channels=2, num_conv=2))),
    TARGET_ASSIGNER_CONFIG=dict(FEATURE_MAP_STRIDE=8, NUM_MAX_OBJS=500, GAUSSIAN_OVERLAP=0.1, MIN_RADIUS=2),
    LOSS_CONFIG=dict(LOSS_WEIGHTS=dict(cls_weight=1.0, loc_weight=0.25, code_weights=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2, 1.0, 1.0])),
    POST_PROCESSING=dict(SCORE_THRESH=0.1, POST_CENTER_LIMIT_RANGE=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], MAX_OBJ_PER_SAMPLE=500,
                         NMS_CONFIG=dict(NMS_TYPE='nms_gpu', NMS_THRESH=0.2, NMS_PRE_MAXSIZE=1000, NMS_POST_MAXSIZE=83)))


# detector3d/tools/cfgs/kitti_models/second_iou.yaml:88-140 (values as data)
def second_iou_roi_head(in_channel=512, shared_fc=(256, 256), iou_fc=(256, 256), roi_per_image=128, train_post=512, test_post=100):
    return dict(
        NAME='SECONDHead', CLASS_AGNOSTIC=True, SHARED_FC=list(shared_fc), IOU_FC=list(iou_fc), DP_RATIO=0.3,
        NMS_CONFIG=dict(TRAIN=dict(NMS_TYPE='nms_gpu', MULTI_CLASSES_NMS=False, NMS_PRE_MAXSIZE=9000, NMS_POST_MAXSIZE=train_post, NMS_THRESH=0.8),
                        TEST=dict(NMS_TYPE='nms_gpu', MULTI_CLASSES_NMS=False, NMS_PRE_MAXSIZE=1024, NMS_POST_MAXSIZE=test_post, NMS_THRESH=0.7)),
        ROI_GRID_POOL=dict(GRID_SIZE=7, IN_CHANNEL=in_channel, DOWNSAMPLE_RATIO=8),
        TARGET_CONFIG=dict(BOX_CODER='ResidualCoder', ROI_PER_IMAGE=roi_per_image, FG_RATIO=0.5, SAMPLE_ROI_BY_EACH_CLASS=True,
                           CLS_SCORE_TYPE='roi_iou', CLS_FG_THRESH=0.75, CLS_BG_THRESH=0.25, CLS_BG_THRESH_LO=0.1, HARD_BG_RATIO=0.8,
                           REG_FG_THRESH=0.55),
        LOSS_CONFIG=dict(IOU_LOSS='BinaryCrossEntropy', LOSS_WEIGHTS=dict(rcnn_iou_weight=1.0, code_weights=[1.0] * 7)))


SECOND_IOU_POST = dict(RECALL_THRESH_LIST=[0.3, 0.5, 0.7], SCORE_THRESH=0.1, OUTPUT_RAW_SCORE=False, EVAL_METRIC='kitti',
                       NMS_CONFIG=dict(MULTI_CLASSES_NMS=False, NMS_TYPE='nms_gpu', NMS_THRESH=0.01, NMS_PRE_MAXSIZE=4096, NMS_POST_MAXSIZE=500,
                                       SCORE_TYPE='iou'))


def centerpoint_model_cfg():
    """MODEL section of nuscenes_models/cbgs_voxel0075_res3d_centerpoint.yaml:64-140 (values as data)."""
    return dict(NAME='CenterPoint', VFE=dict(NAME='MeanVFE'), BACKBONE_3D=dict(NAME='VoxelResBackBone8x'),
                MAP_TO_BEV=dict(NAME='HeightCompression', NUM_BEV_FEATURES=256), BACKBONE_2D=SECOND_BACKBONE_2D, DENSE_HEAD=CENTER_HEAD,
                POST_PROCESSING=dict(RECALL_THRESH_LIST=[0.3, 0.5, 0.7], EVAL_METRIC='kitti'))


# detector3d/tools/cfgs/source-nuscenes/pvrcnn.yaml:60-215 (values as data): the SEE-VCN PV-RCNN -- one class, 4096 keypoints, no x_conv1/x_conv2 sources
DA_RANGE = [-75.2, -75.2, -2.0, 75.2, 75.2, 4.0]
DA_VOXEL = [0.1, 0.1, 0.15]


def see_pvrcnn_model_cfg(dynamic_vfe=True, **kw):
    kw.setdefault('num_keypoints', 4096)
    kw.setdefault('features_source', ('bev', 'x_conv3', 'x_conv4', 'raw_points'))
    pfe, point_head, roi_head = pvrcnn_cfg(**kw)
    roi_head['NMS_CONFIG']['TEST'] = dict(NMS_TYPE='nms_gpu', MULTI_CLASSES_NMS=False, NMS_PRE_MAXSIZE=4096, NMS_POST_MAXSIZE=300, NMS_THRESH=0.85)
    head = dict(SECOND_DENSE_HEAD, ANCHOR_GENERATOR_CONFIG=[dict(_anchor('car', [4.2, 2.0, 1.6], 0, 0.55, 0.4))])
    pp = dict(SECOND_POST_PROCESSING, NMS_CONFIG=dict(SECOND_POST_PROCESSING['NMS_CONFIG'], NMS_THRESH=0.7, NMS_POST_MAXSIZE=500))
    return dict(NAME='PVRCNN', VFE=dict(NAME='DynMeanVFE' if dynamic_vfe else 'MeanVFE'), BACKBONE_3D=dict(NAME='VoxelBackBone8x'),
                MAP_TO_BEV=dict(NAME='HeightCompression', NUM_BEV_FEATURES=256), BACKBONE_2D=SECOND_BACKBONE_2D, DENSE_HEAD=head,
                PFE=pfe, POINT_HEAD=point_head, ROI_HEAD=roi_head, POST_PROCESSING=pp)
