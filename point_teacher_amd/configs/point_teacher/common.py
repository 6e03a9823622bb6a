"""Builder for the four AI-TOD-v2 Point-Teacher settings (point noise m = 0/30/60/100 %).
Produces the same config dictionary as the reference's
HBB_TOD/configs/point_teacher/aitodv2_point_teacher_{0,30,60,100}%.py (checked key by key in
tests/test_config.py when the reference tree is present)."""


def _proposal(base_ratios, min_scale, shake_ratio=None, gen_num_neg=0):
    return dict(gen_mode='refine', gen_proposal_mode='fix_gen', cut_mode=None, shake_ratio=shake_ratio,
                base_ratios=base_ratios, min_scale=min_scale, pos_iou_thr=0.3, neg_iou_thr=0.3,
                gen_num_neg=gen_num_neg)


def _topk(num_pre, cls_w):
    return dict(type='TopkAssigner', num_pre=num_pre, topk=num_pre,
                cls_cost=dict(type='FocalLossCost', weight=cls_w),
                reg_cost=dict(type='PointCost', mode='L1', weight=1.0))


SETTINGS = {
    # percent: (num_training, lamda, _point_, coarse ratios, extensive stage-0 (ratios, shake), stage-1 ratios)
    0: (100, 1.0, 0.0, [1.0], ([1.0, 1.2, 1.3, 0.8, 0.7], None), [1.0, 1.2, 1.3, 0.8, 0.7]),
    30: (75, 0.5, 0.3, [1.0, 1.3, 0.8], ([1.0, 1.3, 0.7], [0.1]), [1.0, 1.2, 1.3, 0.8, 0.7]),
    60: (75, 0.5, 0.6, [1.0, 1.2, 0.8], ([1.0, 1.2, 0.8], [0.1]), [1.0, 1.2, 1.3, 0.8, 0.7]),
    100: (75, 0.5, 1.0, [1.0, 1.3, 0.8], ([1.0, 1.3, 0.7], [0.1]), [1.0, 1.2, 1.3, 0.8, 0.7]),
}


def make(percent):
    n_train, lamda, point, coarse, (ext0, shake0), ext1 = SETTINGS[percent]
    cfg = dict(num_classes=8, burn_in_step=4000, ema_alpha=0.999, num_stages=1, mil_stack_conv=0, top_k=1,
               mil_neg_samples=200, num_training_burninstep1=n_train, num_training_burninstep2=n_train,
               lamda=lamda, _point_=point, beta=0.25, alpha=[0.01, 0.25],
               shape_list=[[20, 20, 0.5, 0.5], [10, 20, 0.5, 0.5], [30, 80, 0.5, 0.5], [20, 50, 0.5, 0.5],
                           [30, 120, 0.5, 0.5], [30, 40, 0.5, 0.5]])
    head = dict(
        type='TS_P2BFCOSHead', norm_cfg=None, num_classes=cfg['num_classes'], in_channels=256, stacked_convs=4,
        mil_stack_conv=cfg['mil_stack_conv'], feat_channels=256, strides=[8], norm_on_bbox=True,
        centerness_on_reg=True, dcn_on_last_conv=False, center_sampling=True, conv_bias=True, beta=cfg['beta'],
        top_k=cfg['top_k'], num_stages=cfg['num_stages'],
        bbox_roi_extractor=dict(type='SingleRoIExtractor', roi_layer=dict(type='RoIAlign', output_size=7),
                                out_channels=256, featmap_strides=[8]),
        loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
        loss_bbox_burn1=dict(type='DIoULoss', loss_weight=1.0),
        loss_bbox_burn2=dict(type='DN_DIoULoss', loss_weight=1.0, hyper=0.1),
        loss_bbox_denosing=dict(type='DN_DIoULoss', loss_weight=1.0, hyper=0.2),
        loss_centerness=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0))
    cfg['detector'] = dict(
        type='Student_FCOS',
        backbone=dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                      norm_cfg=dict(type='BN', requires_grad=False), norm_eval=True, style='caffe',
                      init_cfg=dict(type='Pretrained', checkpoint='open-mmlab://detectron/resnet50_caffe')),
        neck=dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256, start_level=1,
                  add_extra_convs='on_output', num_outs=5, relu_before_extra_convs=True),
        neck_agg=dict(type='PSAGG', num_aggregation=5, in_channels=256, out_channels=256),
        bbox_head=head)
    fuse = dict(type='FUSETopkAssigner', num_pre=5, topk=3, cls_cost=dict(type='FocalLossCost', weight=1.0),
                reg_cost=dict(type='PointCost', mode='L1', weight=1.0),
                location_cost=dict(type='InsiderCost', weight=1.0))
    cfg['model'] = dict(
        type='TS_P2B_FCOS', _model_=cfg['detector'], ema_alpha=0.999, num_stages=cfg['num_stages'],
        burn_in_step=cfg['burn_in_step'], filter_score=0.0, lamda=lamda, _point_=point, alpha=cfg['alpha'],
        shape_list=cfg['shape_list'], num_training_burninstep1=n_train, num_training_burninstep2=n_train,
        train_cfg=dict(
            assigner=_topk(1, 1.0), pseudo_assigner=_topk(3, 0.0), syn_assigner=_topk(3, 0.0), fuse_assigner=fuse,
            fine_proposal_cfg=[_proposal(coarse, 0, gen_num_neg=200),
                               _proposal([1.0], 4, gen_num_neg=cfg['mil_neg_samples'])],
            fine_proposal_extensive_cfg=[_proposal(ext0, 4, shake0), _proposal(ext1, 16, [0.1])]),
        test_cfg=dict(nms_pre=3000, min_bbox_size=0, score_thr=0.05, nms=dict(type='nms', iou_threshold=0.5),
                      max_per_img=3000))
    norm = dict(mean=[0.0, 0.0, 0.0], std=[1.0, 1.0, 1.0], to_rgb=False)
    cfg['img_norm_cfg'] = norm
    cfg['train_pipeline'] = [
        dict(type='LoadImageFromFile'), dict(type='LoadAnnotations', with_bbox=True),
        dict(type='Resize', img_scale=(800, 800), keep_ratio=True), dict(type='RandomFlip', flip_ratio=0.0),
        dict(type='Normalize', **norm), dict(type='Pad', size_divisor=32), dict(type='DefaultFormatBundle'),
        dict(type='Collect', keys=['img', 'gt_bboxes', 'gt_labels'])]
    cfg['test_pipeline'] = [
        dict(type='LoadImageFromFile'),
        dict(type='MultiScaleFlipAug', img_scale=(800, 800), flip=False,
             transforms=[dict(type='Resize', keep_ratio=True), dict(type='RandomFlip'),
                         dict(type='Normalize', **norm), dict(type='Pad', size_divisor=32),
                         dict(type='ImageToTensor', keys=['img']), dict(type='Collect', keys=['img'])])]
    cfg['data'] = dict(samples_per_gpu=2, workers_per_gpu=2, train=dict(pipeline=cfg['train_pipeline']),
                       val=dict(pipeline=cfg['test_pipeline']), test=dict(pipeline=cfg['test_pipeline']))
    cfg['optimizer'] = dict(lr=0.01 / 2, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.))
    cfg['optimizer_config'] = dict(_delete_=True, grad_clip=dict(max_norm=35, norm_type=2))
    cfg['lr_config'] = dict(policy='step', warmup='constant', warmup_iters=10000, warmup_ratio=1.0 / 3,
                            step=[8, 11])
    cfg['runner'] = dict(type='EpochBasedRunner', max_epochs=12)
    return cfg
