# Runtime defaults of the oriented-box tree (same values as OBB_TOD/configs/_base_/default_runtime.py).
checkpoint_config = dict(interval=1)
log_config = dict(interval=50, hooks=[dict(type='TextLoggerHook')])
dist_params = dict(backend='nccl')      # 'nccl' is RCCL on ROCm
log_level = 'INFO'
load_from = None
resume_from = None
workflow = [('train', 1)]
opencv_num_threads = 0
mp_start_method = 'fork'
