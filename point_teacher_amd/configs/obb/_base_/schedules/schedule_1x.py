# 12-epoch SGD schedule of the oriented-box tree (OBB_TOD/configs/_base_/schedules/schedule_1x.py).
evaluation = dict(interval=1, metric='mAP')
optimizer = dict(type='SGD', lr=0.0025, momentum=0.9, weight_decay=0.0001)
optimizer_config = dict(grad_clip=dict(max_norm=35, norm_type=2))
lr_config = dict(policy='step', warmup='linear', warmup_iters=500, warmup_ratio=1.0 / 3, step=[8, 11])
runner = dict(type='EpochBasedRunner', max_epochs=12)
