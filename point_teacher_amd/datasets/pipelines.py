"""Train/test pipelines of the reference, re-designed for MI355X.

Same transform names, constructor keywords, `results` keys and meta keys as
/root/reference/HBB_TOD/mmdet/datasets/pipelines/{loading,transforms,formating,test_time_aug,compose}.py and
/root/reference/OBB_TOD/mmrotate/datasets/pipelines/transforms.py (RResize, RRandomFlip), so the configs'
`train_pipeline` / `test_pipeline` lists build unchanged.  What differs is WHERE the pixel work runs:

* the reference touches every pixel four times on CPU dataloader workers (cv2.resize, cv2.flip, subtract/multiply,
  copyMakeBorder) and a fifth time in collate's zero padding, then ships float32 over PCIe;
* here `results['img']` is a `LazyImage` - the decoded uint8 array plus the operations still owed.  Resize / flip /
  normalize / pad only RECORD their parameters (and do the box and meta arithmetic on the host, as the reference
  does); the batch collate (`loader.collate_to_device`) uploads the uint8 bytes (4x fewer than float32) and renders
  every sample straight into its slice of the channels-last batch tensor with ONE `pt_image_prep` launch.

The pixel arithmetic is the reference's (OpenCV 8-bit fixed-point bilinear, imnormalize roundings): see
csrc/image_prep.hip.  Transform orders the fused kernel cannot express (e.g. Pad before Normalize) raise loudly.
"""
import collections
import io
import os.path as osp
import warnings

import numpy as np
import torch

from ..registry import Registry, build_from_cfg

PIPELINES = Registry('pipeline')
ROTATED_PIPELINES = PIPELINES            # mmrotate/datasets/builder.py: ROTATED_PIPELINES = PIPELINES

_FLIP_BITS = {None: 0, 'horizontal': 1, 'vertical': 2, 'diagonal': 3}


class DataContainer:
    """mmcv.parallel.DataContainer: a value plus how collate treats it."""

    def __init__(self, data, stack=False, padding_value=0, cpu_only=False, pad_dims=2):
        self.data, self.stack, self.padding_value, self.cpu_only, self.pad_dims = data, stack, padding_value, cpu_only, pad_dims

    def __repr__(self):
        return f'DataContainer({self.data!r})'


DC = DataContainer


class LazyImage:
    """A decoded uint8 HxWx3 (BGR) image and the pixel operations the pipeline has asked for so far.
    `shape` / `dtype` answer what the reference's ndarray would answer at the same point of the pipeline."""
    _ORDER = ('resize', 'flip', 'normalize', 'pad')

    def __init__(self, array, cache_key=None):
        """`array`: the decoded image as a host ndarray, or - when it comes out of the loader's HBM cache - as a CUDA uint8
        tensor that needs no upload.  `cache_key`: where the collate may keep the uploaded bytes for the next epoch."""
        if isinstance(array, torch.Tensor):
            if not (array.is_cuda and array.dtype == torch.uint8 and array.dim() == 3 and array.shape[2] == 3 and array.is_contiguous()):
                raise TypeError('LazyImage: a cached image must be a contiguous CUDA uint8 HxWx3 tensor')
            self.src = array
        else:
            if array.dtype != np.uint8 or array.ndim != 3 or array.shape[2] != 3:
                raise TypeError(f'LazyImage needs a uint8 HxWx3 array, got {array.dtype} {array.shape}')
            self.src = np.ascontiguousarray(array)
        self.cache_key = cache_key
        self.rs = None              # (h, w) after Resize
        self.flip = 0               # bit 0 horizontal, bit 1 vertical
        self.norm = None            # (mean float32[3], std float32[3], to_rgb)
        self.pad = None             # (h, w, pad_val)
        self._stage = -1

    def _advance(self, op):
        k = self._ORDER.index(op)
        if k < self._stage or (k == self._stage and op != 'flip'):
            raise NotImplementedError(
                f'{op} after {self._ORDER[self._stage]}: the fused GPU image preparation (pt_image_prep) renders '
                'Resize -> RandomFlip -> Normalize -> Pad in that order, as every Point-Teacher config does')
        self._stage = k

    @property
    def shape(self):
        if self.pad is not None:
            return (self.pad[0], self.pad[1], 3)
        if self.rs is not None:
            return (self.rs[0], self.rs[1], 3)
        return tuple(self.src.shape)

    @property
    def dtype(self):
        return np.dtype(np.float32) if self.norm is not None else np.dtype(np.uint8)

    def resize(self, size_wh):
        self._advance('resize')
        self.rs = (int(size_wh[1]), int(size_wh[0]))

    def flip_(self, direction):
        self._advance('flip')
        self.flip ^= _FLIP_BITS[direction]

    def normalize(self, mean, std, to_rgb):
        self._advance('normalize')
        self.norm = (np.asarray(mean, np.float32), np.asarray(std, np.float32), bool(to_rgb))

    def pad_to(self, h, w, pad_val):
        self._advance('pad')
        if isinstance(pad_val, dict):
            pad_val = pad_val.get('img', 0)
        self.pad = (int(h), int(w), float(pad_val))

    def copy(self):
        out = LazyImage.__new__(LazyImage)
        out.__dict__.update(self.__dict__)
        return out

    def render(self, dst, device_src=None, stream=None):
        """One pt_image_prep launch: write this image into `dst`, a float32 CUDA view [3, H, W] of any strides
        (H, W >= the padded shape; the margin is zero-filled as mmcv's collate does)."""
        from .. import hip
        if not (isinstance(dst, torch.Tensor) and dst.is_cuda and dst.dtype == torch.float32 and dst.dim() == 3 and dst.shape[0] == 3):
            raise RuntimeError('LazyImage.render: dst must be a float32 CUDA view [3, H, W] (no CPU path)')
        if device_src is None:
            device_src = self.src if isinstance(self.src, torch.Tensor) else torch.from_numpy(self.src).to(dst.device, non_blocking=True)
        sh, sw = self.src.shape[:2]
        rh, rw = self.rs if self.rs is not None else (sh, sw)
        ph, pw, pv = self.pad if self.pad is not None else (rh, rw, 0.0)
        mean = stdinv = None
        to_rgb = 0
        if self.norm is not None:
            m, s, to_rgb = self.norm
            mean = hip.host_floats(m.tolist())
            stdinv = hip.host_doubles((1 / np.float64(s)).tolist())      # mmcv imnormalize_: 1 / np.float64(std)
        args = [device_src, sh, sw, sw * 3, 3, rh, rw, self.flip, mean, stdinv, int(to_rgb), ph, pw, pv,
                dst.shape[1], dst.shape[2], dst.data_ptr(), dst.stride(0), dst.stride(1), dst.stride(2)]
        if stream is not None:
            args.append(stream)
        hip.call('pt_image_prep', *args)
        return dst


def decode_image(content, flag='color'):
    """mmcv.imfrombytes(content, flag='color') -> uint8 HxWx3 BGR.  The reference's default backend is cv2 (not
    installed); this is mmcv's own 'pillow' backend (mmcv/image/io.py `_pillow2array`), lossless formats (AI-TOD is
    PNG) decode to identical bytes."""
    if flag != 'color':
        raise NotImplementedError(f"imfrombytes flag '{flag}': the Point-Teacher pipelines decode with 'color'")
    try:
        from PIL import Image, ImageOps
    except ImportError as e:        # pragma: no cover
        raise RuntimeError('no image decoder: Pillow is required by LoadImageFromFile') from e
    with Image.open(io.BytesIO(content)) as im:
        im = ImageOps.exif_transpose(im)                # cv2.imread honours the EXIF orientation
        if im.mode != 'RGB':
            im = im.convert('RGB')
        rgb = np.asarray(im)
    return np.ascontiguousarray(rgb[:, :, ::-1])


class DeviceImageCache:
    """Decoded tiles kept in HBM between epochs.  AI-TOD-v2's 11 214 training tiles are 21.5 GB as uint8, SODA-A's ~31 k
    patches ~60 GB: either fits beside the model in the 288 GB of one MI355X, so from the second epoch on an image costs no
    file read, no PNG/JPEG decode and no PCIe transfer - only its `pt_image_prep` launch.  The tensors are the very uploads
    the first epoch made (nothing is copied to fill the cache); `max_bytes` bounds it, later images simply stay uncached."""

    def __init__(self, max_bytes=64 << 30):
        import threading
        self.max_bytes, self.bytes, self.hits, self.misses = int(max_bytes), 0, 0, 0
        self._store, self._lock = {}, threading.Lock()

    def get(self, key):
        t = self._store.get(key)
        with self._lock:
            if t is None:
                self.misses += 1
            else:
                self.hits += 1
        return t

    def put(self, key, tensor):
        n = tensor.numel()
        with self._lock:
            if key in self._store or self.bytes + n > self.max_bytes:
                return False
            self._store[key] = tensor
            self.bytes += n
        return True

    def __len__(self):
        return len(self._store)


class Compose:
    """pipelines/compose.py:8-51"""

    def __init__(self, transforms):
        assert isinstance(transforms, collections.abc.Sequence)
        self.transforms = []
        for t in transforms:
            if isinstance(t, dict):
                self.transforms.append(build_from_cfg(t, PIPELINES))
            elif callable(t):
                self.transforms.append(t)
            else:
                raise TypeError('transform must be callable or a dict')

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
            if data is None:
                return None
        return data

    def __repr__(self):
        return self.__class__.__name__ + '(' + ''.join(f'\n    {t}' for t in self.transforms) + '\n)'


@PIPELINES.register_module()
class LoadImageFromFile:
    """loading.py:12-77"""

    def __init__(self, to_float32=False, color_type='color', file_client_args=dict(backend='disk')):
        if to_float32:
            raise NotImplementedError('to_float32=True resizes in floating point (a different cv2 code path); '
                                      'no Point-Teacher config uses it')
        if file_client_args.get('backend', 'disk') != 'disk':
            raise NotImplementedError("only the 'disk' file client backend")
        self.to_float32, self.color_type, self.file_client_args = to_float32, color_type, dict(file_client_args)
        self.cache = None               # a DeviceImageCache, installed by DeviceLoader(cache_bytes=...)

    def __call__(self, results):
        if results['img_prefix'] is not None:
            filename = osp.join(results['img_prefix'], results['img_info']['filename'])
        else:
            filename = results['img_info']['filename']
        cached = self.cache.get(filename) if self.cache is not None else None
        if cached is not None:
            img = LazyImage(cached)
        else:
            with open(filename, 'rb') as f:
                img = LazyImage(decode_image(f.read(), self.color_type), cache_key=filename if self.cache is not None else None)
        results['filename'] = filename
        results['ori_filename'] = results['img_info']['filename']
        results['img'] = img
        results['img_shape'] = img.shape
        results['ori_shape'] = img.shape
        results['img_fields'] = ['img']
        return results

    def __repr__(self):
        return (f"{self.__class__.__name__}(to_float32={self.to_float32}, color_type='{self.color_type}', "
                f'file_client_args={self.file_client_args})')


@PIPELINES.register_module()
class LoadAnnotations:
    """loading.py:196-384, the box / label part (masks and segmentation maps are not on the Point-Teacher path)."""

    def __init__(self, with_bbox=True, with_label=True, with_mask=False, with_seg=False, poly2mask=True,
                 file_client_args=dict(backend='disk')):
        if with_mask or with_seg:
            raise NotImplementedError('with_mask / with_seg: Point-Teacher trains from boxes (points) only')
        self.with_bbox, self.with_label = with_bbox, with_label

    def __call__(self, results):
        ann = results['ann_info']
        if self.with_bbox:
            results['gt_bboxes'] = ann['bboxes'].copy()
            ignore = ann.get('bboxes_ignore', None)
            if ignore is not None:
                results['gt_bboxes_ignore'] = ignore.copy()
                results['bbox_fields'].append('gt_bboxes_ignore')
            results['bbox_fields'].append('gt_bboxes')
        if self.with_label:
            results['gt_labels'] = ann['labels'].copy()
        return results

    def __repr__(self):
        return f'{self.__class__.__name__}(with_bbox={self.with_bbox}, with_label={self.with_label})'


def rescale_size(old_size, scale, return_scale=False):
    """mmcv/image/geometric.py rescale_size: old (w, h); scale = factor or (long edge, short edge) bound."""
    w, h = old_size
    if isinstance(scale, (float, int)):
        if scale <= 0:
            raise ValueError(f'Invalid scale {scale}, must be positive.')
        scale_factor = scale
    elif isinstance(scale, tuple):
        max_long_edge, max_short_edge = max(scale), min(scale)
        scale_factor = min(max_long_edge / max(h, w), max_short_edge / min(h, w))
    else:
        raise TypeError(f'Scale must be a number or tuple of int, but got {type(scale)}')
    new_size = (int(w * float(scale_factor) + 0.5), int(h * float(scale_factor) + 0.5))
    return (new_size, scale_factor) if return_scale else new_size


@PIPELINES.register_module()
class Resize:
    """transforms.py:26-316"""

    def __init__(self, img_scale=None, multiscale_mode='range', ratio_range=None, keep_ratio=True, bbox_clip_border=True,
                 backend='cv2', override=False):
        if img_scale is None:
            self.img_scale = None
        else:
            self.img_scale = img_scale if isinstance(img_scale, list) else [img_scale]
            assert all(isinstance(s, tuple) for s in self.img_scale)
        if ratio_range is not None:
            assert len(self.img_scale) == 1
        else:
            assert multiscale_mode in ['value', 'range']
        if backend != 'cv2':
            raise NotImplementedError("Resize backend: only 'cv2' arithmetic (bilinear, 8-bit fixed point) is implemented")
        self.backend, self.multiscale_mode, self.ratio_range = backend, multiscale_mode, ratio_range
        self.keep_ratio, self.override, self.bbox_clip_border = keep_ratio, override, bbox_clip_border

    def _pick_scale(self):
        """-> (scale, index or None).  One place for the three sampling rules of transforms.py:92-186; the order and kind of the
        numpy draws is the reference's (tests/golden/pipeline_flow.npz pins it):
          ratio_range            one uniform draw r in [lo, hi): the single img_scale times r, truncated per edge
          one img_scale          that scale, no draw
          'range' (two scales)   an integer long edge, then an integer short edge, each uniform between the two scales' edges
          'value'                one integer draw: an entry of the list."""
        scales = self.img_scale
        if self.ratio_range is not None:
            lo, hi = self.ratio_range
            assert lo <= hi and len(scales[0]) == 2
            r = np.random.random_sample() * (hi - lo) + lo
            return tuple(int(edge * r) for edge in scales[0]), None
        if len(scales) == 1:
            return scales[0], 0
        if self.multiscale_mode == 'range':
            assert len(scales) == 2
            edges = [sorted((max(s) for s in scales)), sorted((min(s) for s in scales))]       # [long lo..hi], [short lo..hi]
            long_edge, short_edge = (np.random.randint(lo, hi + 1) for lo, hi in edges)
            return (long_edge, short_edge), None
        i = np.random.randint(len(scales))
        return scales[i], i

    def _resize_img(self, results):
        keep = self.keep_ratio
        for key in results.get('img_fields', ['img']):
            img = results[key]
            h, w = img.shape[:2]
            new_w, new_h = rescale_size((w, h), results['scale']) if keep else results['scale']
            img.resize((new_w, new_h))
            fx, fy = new_w / w, new_h / h
            results.update(scale_factor=np.array([fx, fy, fx, fy], dtype=np.float32), img_shape=img.shape, pad_shape=img.shape, keep_ratio=keep)

    def _resize_bboxes(self, results):
        """boxes x (fx, fy, fx, fy), then (bbox_clip_border) every x into [0, width] and every y into [0, height] with one
        minimum / maximum against a per-column limit row."""
        fields = results.get('bbox_fields', [])
        if not fields:
            return
        hh, ww = results['img_shape'][:2]
        for key in fields:
            boxes = results[key] * results['scale_factor']
            if self.bbox_clip_border:
                upper = np.resize(np.asarray([ww, hh], dtype=boxes.dtype), boxes.shape[-1])
                boxes = np.minimum(np.maximum(boxes, 0), upper)
            results[key] = boxes

    def __call__(self, results):
        have_scale, have_factor = 'scale' in results, 'scale_factor' in results
        if have_scale and not self.override:
            assert not have_factor, 'scale and scale_factor cannot be both set.'
        elif have_scale:                                   # override: forget what an earlier Resize chose and draw again
            del results['scale']
            results.pop('scale_factor', None)
            have_scale = have_factor = False
        if not have_scale:
            if have_factor:                                # a caller-given float factor on the current image size, (w, h) order
                f = results['scale_factor']
                assert isinstance(f, float)
                results['scale'] = tuple(int(side * f) for side in results['img'].shape[1::-1])
            else:
                results['scale'], results['scale_idx'] = self._pick_scale()
        self._resize_img(results)
        self._resize_bboxes(results)
        return results

    def __repr__(self):
        return (f'{self.__class__.__name__}(img_scale={self.img_scale}, multiscale_mode={self.multiscale_mode}, '
                f'ratio_range={self.ratio_range}, keep_ratio={self.keep_ratio}, bbox_clip_border={self.bbox_clip_border})')


@PIPELINES.register_module()
class RResize(Resize):
    """mmrotate transforms.py:16-46: always keep_ratio; (cx, cy) scale per axis, (w, h) by sqrt(w_scale * h_scale)."""

    def __init__(self, img_scale=None, multiscale_mode='range', ratio_range=None):
        super().__init__(img_scale=img_scale, multiscale_mode=multiscale_mode, ratio_range=ratio_range, keep_ratio=True)

    def _resize_bboxes(self, results):
        """Every (cx, cy, w, h, a) row times (w_scale, h_scale, g, g, 1), g = sqrt(w_scale * h_scale): ONE broadcast multiply per
        field in the boxes' own dtype (the same float32 products, column by column, as mmrotate's transforms.py:36-46 forms with
        three in-place column updates; pinned by tests/golden/pipeline_flow.npz)."""
        w_scale, h_scale = results['scale_factor'][:2]
        g = np.sqrt(w_scale * h_scale)
        for key in results.get('bbox_fields', []):
            boxes = results[key]
            factors = np.asarray([w_scale, h_scale, g, g, 1], dtype=boxes.dtype)
            results[key] = (boxes.reshape(-1, 5) * factors).reshape(boxes.shape)


@PIPELINES.register_module()
class RandomFlip:
    """transforms.py:319-472"""

    def __init__(self, flip_ratio=None, direction='horizontal'):
        if isinstance(flip_ratio, list):
            assert all(isinstance(r, float) for r in flip_ratio) and 0 <= sum(flip_ratio) <= 1
        elif isinstance(flip_ratio, float):
            assert 0 <= flip_ratio <= 1
        elif flip_ratio is not None:
            raise ValueError('flip_ratios must be None, float, or list of float')
        self.flip_ratio = flip_ratio
        valid = ['horizontal', 'vertical', 'diagonal']
        if isinstance(direction, str):
            assert direction in valid
        elif isinstance(direction, list):
            assert set(direction).issubset(set(valid))
        else:
            raise ValueError('direction must be either str or list of str')
        self.direction = direction
        if isinstance(flip_ratio, list):
            assert len(self.flip_ratio) == len(self.direction)

    # direction -> (source column of every output column, sign, which image side is added): x' = W - x swaps the x columns,
    # y' = H - y the y columns; 'diagonal' does both.  One gather + one fused multiply-add instead of a branch per direction.
    _FLIP = {'horizontal': ((2, 1, 0, 3), (-1, 1, -1, 1), (1, 0, 1, 0)),
             'vertical': ((0, 3, 2, 1), (1, -1, 1, -1), (0, 1, 0, 1)),
             'diagonal': ((2, 3, 0, 1), (-1, -1, -1, -1), (1, 1, 1, 1))}

    def bbox_flip(self, bboxes, img_shape, direction):
        """[..., 4k] (x1, y1, x2, y2) groups mirrored inside an image of `img_shape` (h, w, ...)."""
        if direction not in self._FLIP:
            raise ValueError(f"Invalid flipping direction '{direction}'")
        n = bboxes.shape[-1]
        assert n % 4 == 0
        src, sign, side = (np.asarray(t) for t in self._FLIP[direction])
        cols = (np.arange(n) // 4 * 4)[:, None].reshape(-1, 4)[:, :1] + src                  # per group of four: its source columns
        h, w = img_shape[0], img_shape[1]
        extent = np.where(np.tile(side, n // 4) == 1, np.tile([w, h, w, h], n // 4), 0).astype(bboxes.dtype)
        return extent + np.tile(sign, n // 4).astype(bboxes.dtype) * bboxes[..., cols.reshape(-1)]

    def _draw_direction(self):
        """None (no flip) or a direction, drawn with ONE np.random.choice over [directions..., None] (transforms.py:415-433): a
        list of ratios pairs with the list of directions, a single ratio is shared equally between them."""
        dirs = list(self.direction) if isinstance(self.direction, list) else [self.direction]
        if isinstance(self.flip_ratio, list):
            probs = self.flip_ratio + [1 - sum(self.flip_ratio)]
        else:
            probs = [self.flip_ratio / len(dirs)] * len(dirs) + [1 - self.flip_ratio]
        return np.random.choice(dirs + [None], p=probs)

    def __call__(self, results):
        if 'flip' not in results:
            drawn = self._draw_direction()
            results['flip'] = drawn is not None
            results.setdefault('flip_direction', drawn)
        if results['flip']:
            d = results['flip_direction']
            for key in results.get('img_fields', ['img']):
                results[key].flip_(d)
            for key in results.get('bbox_fields', []):
                results[key] = self.bbox_flip(results[key], results['img_shape'], d)
        return results

    def __repr__(self):
        return self.__class__.__name__ + f'(flip_ratio={self.flip_ratio})'


def norm_angle(angle, angle_range):
    """mmrotate/core/bbox/transforms.py norm_angle"""
    if angle_range == 'oc':
        return angle
    if angle_range == 'le135':
        return (angle + np.pi / 4) % np.pi - np.pi / 4
    if angle_range == 'le90':
        return (angle + np.pi / 2) % np.pi - np.pi / 2
    raise NotImplementedError(angle_range)


@PIPELINES.register_module()
class RRandomFlip(RandomFlip):
    """mmrotate transforms.py:50-95.  `version` defaults to 'oc' and the SODA-A config does not pass it
    (OBB_TOD/configs/_base_/datasets/sodaa.py:10), so its le90 boxes take the 'oc' rule: kept as is."""

    def __init__(self, flip_ratio=None, direction='horizontal', version='oc'):
        self.version = version
        super().__init__(flip_ratio, direction)

    def bbox_flip(self, bboxes, img_shape, direction):
        """mmrotate transforms.py:75-96 as masks over the box table: which centre coordinates are mirrored is a property of the
        direction, what happens to (w, h, a) a property of the angle version - 'oc': a -> pi/2 - a with w and h swapped, except
        rows whose angle is exactly pi/2; otherwise a -> norm_angle(pi - a).  A diagonal flip leaves (w, h, a) alone."""
        if bboxes.shape[-1] % 5:
            raise AssertionError(bboxes.shape)
        mirror = {'horizontal': (True, False), 'vertical': (False, True), 'diagonal': (True, True)}.get(direction)
        if mirror is None:
            raise ValueError(f'Invalid flipping direction "{direction}"')
        rows = bboxes.reshape(-1, 5)
        out = rows.copy()
        for col, (on, extent) in enumerate(zip(mirror, (img_shape[1], img_shape[0]))):
            if on:
                out[:, col] = extent - rows[:, col] - 1
        if direction != 'diagonal':
            if self.version == 'oc':
                turn = rows[:, 4] != np.pi / 2
                out[:, 4] = np.where(turn, np.pi / 2 - rows[:, 4], rows[:, 4])
                out[:, 2] = np.where(turn, rows[:, 3], rows[:, 2])
                out[:, 3] = np.where(turn, rows[:, 2], rows[:, 3])
            else:
                out[:, 4] = norm_angle(np.pi - rows[:, 4], self.version)
        return out.reshape(bboxes.shape)


@PIPELINES.register_module()
class Normalize:
    """transforms.py:637-675"""

    def __init__(self, mean, std, to_rgb=True):
        self.mean, self.std, self.to_rgb = np.array(mean, dtype=np.float32), np.array(std, dtype=np.float32), to_rgb

    def __call__(self, results):
        for key in results.get('img_fields', ['img']):
            results[key].normalize(self.mean, self.std, self.to_rgb)
        results['img_norm_cfg'] = dict(mean=self.mean, std=self.std, to_rgb=self.to_rgb)
        return results

    def __repr__(self):
        return f'{self.__class__.__name__}(mean={self.mean}, std={self.std}, to_rgb={self.to_rgb})'


@PIPELINES.register_module()
class Pad:
    """transforms.py:566-634"""

    def __init__(self, size=None, size_divisor=None, pad_val=0):
        self.size, self.size_divisor, self.pad_val = size, size_divisor, pad_val
        assert size is not None or size_divisor is not None
        assert size is None or size_divisor is None

    def __call__(self, results):
        for key in results.get('img_fields', ['img']):
            img = results[key]
            h, w = img.shape[:2]
            if self.size is not None:
                ph, pw = self.size
                assert ph >= h and pw >= w
            else:
                ph = int(np.ceil(h / self.size_divisor)) * self.size_divisor
                pw = int(np.ceil(w / self.size_divisor)) * self.size_divisor
            img.pad_to(ph, pw, self.pad_val)
            results['pad_shape'] = img.shape
        results['pad_fixed_size'] = self.size
        results['pad_size_divisor'] = self.size_divisor
        return results

    def __repr__(self):
        return f'{self.__class__.__name__}(size={self.size}, size_divisor={self.size_divisor}, pad_val={self.pad_val})'


def to_tensor(data):
    """formating.py:11-33"""
    if isinstance(data, torch.Tensor):
        return data
    if isinstance(data, np.ndarray):
        return torch.from_numpy(data)
    if isinstance(data, collections.abc.Sequence) and not isinstance(data, str):
        return torch.tensor(data)
    if isinstance(data, int):
        return torch.LongTensor([data])
    if isinstance(data, float):
        return torch.FloatTensor([data])
    raise TypeError(f'type {type(data)} cannot be converted to tensor.')


@PIPELINES.register_module()
class ImageToTensor:
    """formating.py:66-99: the image stays lazy - it becomes a CHW tensor in the batch collate."""

    def __init__(self, keys):
        self.keys = keys

    def __call__(self, results):
        for key in self.keys:
            if not isinstance(results[key], LazyImage):
                raise TypeError(f'ImageToTensor: results[{key!r}] is not a pipeline image')
        return results

    def __repr__(self):
        return self.__class__.__name__ + f'(keys={self.keys})'


@PIPELINES.register_module()
class DefaultFormatBundle:
    """formating.py:174-247"""

    def __call__(self, results):
        if 'img' in results:
            img = results['img']
            results.setdefault('pad_shape', img.shape)
            results.setdefault('scale_factor', 1.0)
            results.setdefault('img_norm_cfg', dict(mean=np.zeros(3, dtype=np.float32), std=np.ones(3, dtype=np.float32),
                                                    to_rgb=False))
            results['img'] = DC(img, stack=True)            # transposed + uploaded + rendered by the collate
        for key in ['proposals', 'gt_bboxes', 'gt_bboxes_ignore', 'gt_labels']:
            if key in results:
                results[key] = DC(to_tensor(results[key]))
        return results

    def __repr__(self):
        return self.__class__.__name__


@PIPELINES.register_module()
class Collect:
    """formating.py:251-326"""

    def __init__(self, keys, meta_keys=('filename', 'ori_filename', 'ori_shape', 'img_shape', 'pad_shape', 'scale_factor',
                                        'flip', 'flip_direction', 'img_norm_cfg')):
        self.keys, self.meta_keys = keys, meta_keys

    def __call__(self, results):
        data = {'img_metas': DC({key: results[key] for key in self.meta_keys}, cpu_only=True)}
        for key in self.keys:
            data[key] = results[key]
        return data

    def __repr__(self):
        return self.__class__.__name__ + f'(keys={self.keys}, meta_keys={self.meta_keys})'


@PIPELINES.register_module()
class MultiScaleFlipAug:
    """test_time_aug.py:10-121"""

    def __init__(self, transforms, img_scale=None, scale_factor=None, flip=False, flip_direction='horizontal'):
        self.transforms = Compose(transforms)
        assert (img_scale is None) ^ (scale_factor is None), 'Must have but only one variable can be setted'
        if img_scale is not None:
            self.img_scale = img_scale if isinstance(img_scale, list) else [img_scale]
            self.scale_key = 'scale'
            assert all(isinstance(s, tuple) for s in self.img_scale)
        else:
            self.img_scale = scale_factor if isinstance(scale_factor, list) else [scale_factor]
            self.scale_key = 'scale_factor'
        self.flip = flip
        self.flip_direction = flip_direction if isinstance(flip_direction, list) else [flip_direction]
        assert all(isinstance(d, str) for d in self.flip_direction)
        if not self.flip and self.flip_direction != ['horizontal']:
            warnings.warn('flip_direction has no effect when flip is set to False')
        if self.flip and not any(t['type'] in ('RandomFlip', 'RRandomFlip') for t in transforms):
            warnings.warn('flip has no effect when RandomFlip is not in transforms')

    def __call__(self, results):
        aug_data = []
        flip_args = [(False, None)]
        if self.flip:
            flip_args += [(True, d) for d in self.flip_direction]
        for scale in self.img_scale:
            for flip, direction in flip_args:
                _results = results.copy()
                for key in _results.get('img_fields', ['img']):
                    _results[key] = _results[key].copy()            # each augmentation owes its own operations
                _results[self.scale_key] = scale
                _results['flip'] = flip
                _results['flip_direction'] = direction
                aug_data.append(self.transforms(_results))
        out = {key: [] for key in aug_data[0]}
        for data in aug_data:
            for key, val in data.items():
                out[key].append(val)
        return out

    def __repr__(self):
        return (f'{self.__class__.__name__}(transforms={self.transforms}, img_scale={self.img_scale}, flip={self.flip}, '
                f'flip_direction={self.flip_direction})')
