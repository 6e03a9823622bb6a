"""The slice of the COCO annotation index the dataset readers use.

The reference goes through `aitodpycocotools.coco.COCO` (HBB_TOD/mmdet/datasets/api_wrappers/coco_api.py:5-47), an
un-vendored fork of pycocotools that is not installed here; this is the published pycocotools `COCO` index
(createIndex, getAnnIds, getCatIds, getImgIds, loadAnns/Cats/Imgs) with the wrapper's snake-case aliases."""
import json
from collections import defaultdict


def _as_list(x):
    return x if isinstance(x, (list, tuple, set)) else [x]


class COCO:
    def __init__(self, annotation_file=None):
        self.dataset, self.anns, self.cats, self.imgs = {}, {}, {}, {}
        self.imgToAnns, self.catToImgs = defaultdict(list), defaultdict(list)
        if annotation_file is not None:
            with open(annotation_file, 'r') as f:
                dataset = json.load(f)
            assert isinstance(dataset, dict), f'annotation file format {type(dataset)} not supported'
            self.dataset = dataset
            self.createIndex()
        self.img_ann_map, self.cat_img_map = self.imgToAnns, self.catToImgs

    def createIndex(self):
        anns, cats, imgs = {}, {}, {}
        imgToAnns, catToImgs = defaultdict(list), defaultdict(list)
        for ann in self.dataset.get('annotations', []):
            imgToAnns[ann['image_id']].append(ann)
            anns[ann['id']] = ann
        for img in self.dataset.get('images', []):
            imgs[img['id']] = img
        for cat in self.dataset.get('categories', []):
            cats[cat['id']] = cat
        if 'categories' in self.dataset:
            for ann in self.dataset.get('annotations', []):
                catToImgs[ann['category_id']].append(ann['image_id'])
        self.anns, self.imgToAnns, self.catToImgs, self.imgs, self.cats = anns, imgToAnns, catToImgs, imgs, cats

    def getAnnIds(self, imgIds=[], catIds=[], areaRng=[], iscrowd=None):
        imgIds, catIds = _as_list(imgIds), _as_list(catIds)
        if len(imgIds) == len(catIds) == len(areaRng) == 0:
            anns = self.dataset['annotations']
        else:
            if len(imgIds) > 0:
                anns = [a for i in imgIds if i in self.imgToAnns for a in self.imgToAnns[i]]
            else:
                anns = self.dataset['annotations']
            if len(catIds) > 0:
                anns = [a for a in anns if a['category_id'] in catIds]
            if len(areaRng) > 0:
                anns = [a for a in anns if areaRng[0] < a['area'] < areaRng[1]]
        if iscrowd is not None:
            return [a['id'] for a in anns if a['iscrowd'] == iscrowd]
        return [a['id'] for a in anns]

    def getCatIds(self, catNms=[], supNms=[], catIds=[]):
        catNms, supNms, catIds = _as_list(catNms), _as_list(supNms), _as_list(catIds)
        cats = self.dataset['categories']
        if len(catNms) > 0:
            cats = [c for c in cats if c['name'] in catNms]
        if len(supNms) > 0:
            cats = [c for c in cats if c['supercategory'] in supNms]
        if len(catIds) > 0:
            cats = [c for c in cats if c['id'] in catIds]
        return [c['id'] for c in cats]

    def getImgIds(self, imgIds=[], catIds=[]):
        imgIds, catIds = _as_list(imgIds), _as_list(catIds)
        if len(imgIds) == len(catIds) == 0:
            ids = self.imgs.keys()
        else:
            ids = set(imgIds)
            for i, c in enumerate(catIds):
                if i == 0 and len(ids) == 0:
                    ids = set(self.catToImgs[c])
                else:
                    ids &= set(self.catToImgs[c])
        return list(ids)

    def loadAnns(self, ids=[]):
        return [self.anns[i] for i in ids] if isinstance(ids, (list, tuple)) else [self.anns[ids]]

    def loadCats(self, ids=[]):
        return [self.cats[i] for i in ids] if isinstance(ids, (list, tuple)) else [self.cats[ids]]

    def loadImgs(self, ids=[]):
        return [self.imgs[i] for i in ids] if isinstance(ids, (list, tuple)) else [self.imgs[ids]]

    # mmdet's wrapper also answers to snake-case method AND keyword names (api_wrappers/coco_api.py:24-41): one adapter that
    # renames the keywords, no second body per method
    def _snake(method, **rename):
        def call(self, *args, **kw):
            return method(self, *args, **{rename.get(k, k): v for k, v in kw.items()})
        call.__name__ = method.__name__
        return call

    get_ann_ids = _snake(getAnnIds, img_ids='imgIds', cat_ids='catIds', area_rng='areaRng')
    get_cat_ids = _snake(getCatIds, cat_names='catNms', sup_names='supNms', cat_ids='catIds')
    get_img_ids = _snake(getImgIds, img_ids='imgIds', cat_ids='catIds')
    load_anns, load_cats, load_imgs = _snake(loadAnns), _snake(loadCats), _snake(loadImgs)
    del _snake
