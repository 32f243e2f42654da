"""Static dataset metadata needed by `MGNet.from_config` (mg_net.py:146-192).  The reference registers it while
scanning the datasets on disk (mgnet/data/cityscapes_scene_seg.py:17-47, kitti_eigen_scene_seg.py); only the constant
part is reproduced: 20 train ids for Cityscapes (0 = "ego vehicle" ... thing ids 12-19), 19 for KITTI (no ego vehicle),
label_divisor 1000, ignore_label 255."""
from types import SimpleNamespace

_CS_STUFF = ["ego vehicle", "road", "sidewalk", "building", "wall", "fence", "pole", "traffic light", "traffic sign",
             "vegetation", "terrain", "sky"]
_THINGS = ["person", "rider", "car", "truck", "bus", "train", "motorcycle", "bicycle"]
# dataset (label-file) ids of the categories, cityscapes_scene_seg.py:18-41 -- what tools/generate_pseudo_labels.py maps train ids back to
_DATASET_ID = {"ego vehicle": 1, "road": 7, "sidewalk": 8, "building": 11, "wall": 12, "fence": 13, "pole": 17, "traffic light": 19,
               "traffic sign": 20, "vegetation": 21, "terrain": 22, "sky": 23, "person": 24, "rider": 25, "car": 26, "truck": 27, "bus": 28,
               "train": 31, "motorcycle": 32, "bicycle": 33}


def _make(names_stuff):
    cats = [{"name": n, "trainId": i, "isthing": 0, "id": _DATASET_ID[n]} for i, n in enumerate(names_stuff)]
    cats += [{"name": n, "trainId": len(names_stuff) + i, "isthing": 1, "id": _DATASET_ID[n]} for i, n in enumerate(_THINGS)]
    # cityscapes_scene_seg.py:228-235: things and stuff in SEPARATE maps (keys there are the raw dataset ids; only the
    # contiguous train ids -- the values -- are used by the model: mg_net.py:150-180)
    things = {c["trainId"]: c["trainId"] for c in cats if c["isthing"]}
    stuff = {c["trainId"]: c["trainId"] for c in cats if not c["isthing"]}
    return SimpleNamespace(categories=cats, thing_dataset_id_to_contiguous_id=things,
                           stuff_dataset_id_to_contiguous_id=stuff, label_divisor=1000, ignore_label=255,
                           stuff_classes=[c["name"] for c in cats], thing_classes=list(_THINGS))


class _Catalog:
    def __init__(self):
        self._d = {}

    def get(self, name):
        if name not in self._d:
            self._d[name] = _make(_CS_STUFF[1:] if name.startswith("kitti") else _CS_STUFF)
            self._d[name].name = name
        return self._d[name]


MetadataCatalog = _Catalog()
