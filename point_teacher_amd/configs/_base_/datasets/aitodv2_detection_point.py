# AI-TOD-v2 point-supervised dataset settings (values of the reference's
# configs/_base_/datasets/aitodv2_detection_point.py; the paths are the author's).
dataset_type = 'AITODDataset'
image_root = '/home/xuchang/dataset/AI-TODv2/'
data_root = '/home/xuchang/dataset/AI-TODv2/annotations/'
img_norm_cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
train_pipeline = [
    dict(type='LoadImageFromFile'),
    dict(type='LoadAnnotations', with_bbox=True),
    dict(type='Resize', img_scale=(800, 800), keep_ratio=True),
    dict(type='RandomFlip', flip_ratio=0.0),
    dict(type='Pad', size_divisor=32),
    dict(type='DefaultFormatBundle'),
    dict(type='Collect', keys=['img', 'gt_bboxes', 'gt_labels']),
]
test_pipeline = [
    dict(type='LoadImageFromFile'),
    dict(type='MultiScaleFlipAug', img_scale=(800, 800), flip=False,
         transforms=[dict(type='Resize', keep_ratio=True), dict(type='RandomFlip'),
                     dict(type='Pad', size_divisor=32), dict(type='ImageToTensor', keys=['img']),
                     dict(type='Collect', keys=['img'])])
]
data = dict(
    samples_per_gpu=2, workers_per_gpu=2,
    train=dict(type=dataset_type, ann_file=data_root + 'aitodv2_train.json', img_prefix=image_root + 'train/',
               pipeline=train_pipeline),
    val=dict(type=dataset_type, ann_file=data_root + 'aitodv2_val.json', img_prefix=image_root + 'val/',
             pipeline=test_pipeline),
    test=dict(type=dataset_type, ann_file=data_root + 'aitodv2_val.json', img_prefix=image_root + 'val/',
              pipeline=test_pipeline))
evaluation = dict(interval=12, metric='bbox')
