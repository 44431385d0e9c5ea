"""Config dicts of the hot path, in the reference's own schema.

``pts_bbox_head`` is the drop-in contract: the three TransCAR configs
(projects/configs/detr3d/detr3d_res101_gridmask.py:51-102 and the _cbgs /
vovnet variants) carry an identical ``pts_bbox_head`` block, so
``build_head(cfg)`` accepts that block unchanged.  Only the FPN level shapes
differ between the ResNet-101 and VoVNet configs.
"""
import copy

point_cloud_range = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
voxel_size = [0.2, 0.2, 8]

pts_bbox_head = dict(
    type='Detr3DHead',
    num_query=900,
    num_classes=10,
    in_channels=256,
    sync_cls_avg_factor=True,
    with_box_refine=True,
    as_two_stage=False,
    transformer=dict(
        type='Detr3DTransformer',
        decoder=dict(
            type='Detr3DTransformerDecoder',
            num_layers=6,
            return_intermediate=True,
            transformerlayers=dict(
                type='DetrTransformerDecoderLayer',
                attn_cfgs=[
                    dict(type='MultiheadAttention', embed_dims=256,
                         num_heads=8, dropout=0.1),
                    dict(type='Detr3DCrossAtten', pc_range=point_cloud_range,
                         num_points=1, embed_dims=256),
                ],
                feedforward_channels=512,
                ffn_dropout=0.1,
                operation_order=('self_attn', 'norm', 'cross_attn', 'norm',
                                 'ffn', 'norm')))),
    bbox_coder=dict(
        type='NMSFreeCoder',
        post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
        pc_range=point_cloud_range,
        max_num=300,
        voxel_size=voxel_size,
        num_classes=10),
    positional_encoding=dict(type='SinePositionalEncoding', num_feats=128,
                             normalize=True, offset=-0.5),
    loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25,
                  loss_weight=2.0),
    loss_bbox=dict(type='L1Loss', loss_weight=0.25),
    loss_iou=dict(type='GIoULoss', loss_weight=0.0))

train_cfg_pts = dict(
    grid_size=[512, 512, 1],
    voxel_size=voxel_size,
    point_cloud_range=point_cloud_range,
    out_size_factor=4,
    assigner=dict(
        type='HungarianAssigner3D',
        cls_cost=dict(type='FocalLossCost', weight=2.0),
        reg_cost=dict(type='BBox3DL1Cost', weight=0.25),
        iou_cost=dict(type='IoUCost', weight=0.0),
        pc_range=point_cloud_range))

#: (H, W) of the four FPN levels the head receives, per backbone config
#: (image padded to 928x1600: transform_3d.py:36; strides from CFG:43-50 and
#: CFG_VOV:39-47).
LEVEL_SHAPES = {
    'res101': [(116, 200), (58, 100), (29, 50), (15, 25)],
    'vovnet': [(232, 400), (116, 200), (58, 100), (29, 50)],
    # tiny maps used by the parity tests (seconds on CPU)
    'tiny': [(8, 12), (4, 6), (2, 3), (1, 2)],
}
IMG_SHAPE = (928, 1600, 3)


def head_cfg(num_query=900):
    cfg = copy.deepcopy(pts_bbox_head)
    cfg['num_query'] = num_query
    return cfg
