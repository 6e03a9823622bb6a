# SODA-A (DOTA-format annotations, 1200x1200 crops) - OBB_TOD/configs/_base_/datasets/sodaarewrite.py
dataset_type = 'SODAADOTADataset'
data_root = '/data/zhr/SODA/SODA-A/'
img_norm_cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)


def _tail(keys):
    return [dict(type='Normalize', **img_norm_cfg), dict(type='Pad', size_divisor=32),
            dict(type='DefaultFormatBundle'), dict(type='Collect', keys=keys)]


train_pipeline = [dict(type='LoadImageFromFile'), dict(type='LoadAnnotations', with_bbox=True),
                  dict(type='RResize', img_scale=(1200, 1200)), dict(type='RRandomFlip', flip_ratio=0.0)] \
    + _tail(['img', 'gt_bboxes', 'gt_labels'])
test_pipeline = [dict(type='LoadImageFromFile'),
                 dict(type='MultiScaleFlipAug', img_scale=(1200, 1200), flip=False,
                      transforms=[dict(type='RResize')] + _tail(['img']))]


def _split(name, pipeline):
    return dict(type=dataset_type, ann_file=data_root + f'divData/{name}/Annotations_filter/',
                img_prefix=data_root + f'divData/{name}/Images_filter/', pipeline=pipeline)


data = dict(samples_per_gpu=2, workers_per_gpu=2, train=_split('train', train_pipeline),
            val=_split('val', test_pipeline), test=_split('val', test_pipeline))
