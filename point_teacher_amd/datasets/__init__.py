"""Data side of the Point-Teacher path: the reference's dataset / pipeline / sampler / dataloader
registry surface, with the pixel work fused into one HIP launch per image (`pt_image_prep`)."""
from .coco_api import COCO                                                           # noqa: F401
from .datasets import (DATASETS, ROTATED_DATASETS, AITODDataset, CocoDataset, CustomDataset, SODAADataset,  # noqa: F401
                       SODAADOTADataset, build_dataset, merge_patch_detections, min_area_rect, poly2obb_np)
from .loader import (DeviceLoader, DistributedGroupSampler, DistributedSampler, EpochBatches, GroupSampler,  # noqa: F401
                     build_dataloader, collate_to_device)
from .pipelines import (PIPELINES, ROTATED_PIPELINES, Collect, Compose, DataContainer, DefaultFormatBundle, DeviceImageCache,  # noqa: F401
                        ImageToTensor, LazyImage, LoadAnnotations, LoadImageFromFile, MultiScaleFlipAug, Normalize, Pad,
                        RandomFlip, Resize, RRandomFlip, RResize, decode_image, rescale_size)
