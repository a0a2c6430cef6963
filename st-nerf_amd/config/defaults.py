"""yacs-free stand-in for the reference's ``config/defaults.py`` (:17-153): the same key tree for everything the
render path reads, as attribute namespaces, with yml override.  ``CfgNode`` offers the part of yacs's surface the
reference uses (``merge_from_file`` / ``freeze`` / ``defrost`` / ``clone``), so it can also stand in for
``yacs.config.CfgNode`` where yacs is not installed (``stnerf_amd.dropin``)."""
import copy

import yaml


class CfgNode(dict):
    """Attribute-access dict with yacs's merge_from_file / freeze / clone surface."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        if self.get("_frozen", False) and k != "_frozen":
            raise AttributeError(f"Attempted to set {k} on a frozen CfgNode")
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    def freeze(self):
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze()
        dict.__setitem__(self, "_frozen", True)

    def defrost(self):
        dict.__setitem__(self, "_frozen", False)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.defrost()

    def merge_from_dict(self, d):
        for k, v in d.items():
            if isinstance(v, dict):
                if k not in self or not isinstance(self[k], CfgNode):
                    dict.__setitem__(self, k, CfgNode())
                self[k].merge_from_dict(v)
            else:
                old = self.get(k)
                if isinstance(old, float) and isinstance(v, (str, int)) and not isinstance(v, bool):
                    v = float(v)  # yacs coerces to the default's type; PyYAML reads "1e10" as a string
                dict.__setitem__(self, k, v)

    def merge_from_file(self, path):
        with open(path) as f:
            self.merge_from_dict(yaml.safe_load(f) or {})


def _defaults():
    c = CfgNode()
    c.merge_from_dict(dict(
        MODEL=dict(DEVICE="cuda", COARSE_RAY_SAMPLING=64, FINE_RAY_SAMPLING=80, SAMPLE_METHOD="NEAR_FAR",
                   BOARDER_WEIGHT=1e10, SAME_SPACENET=False, TKERNEL_INC_RAW=True, POSE_REFINEMENT=True,
                   USE_DIR=True, REMOVE_OUTLIERS=False, TRAIN_BY_POINTCLOUD=False, USE_DEFORM_VIEW=False,
                   USE_DEFORM_TIME=False, BKGD_USE_DEFORM_TIME=False, BKGD_USE_SPACE_TIME=False,
                   USE_SPACE_TIME=False, DEEP_RGB=True),
        INPUT=dict(SIZE_TRAIN=[400, 250], SIZE_TEST=[400, 250], SIZE_LAYER=[400, 250]),
        DATASETS=dict(TRAIN="", TMP_RAYS="rays_tmp", FIXED_NEAR=-1.0, FIXED_FAR=-1.0, SCALE=1.0, FRAME_OFFSET=0,
                      FRAME_NUM=0, LAYER_NUM=0, CAMERA_NUM=0),
        TEST=dict(IMS_PER_BATCH=8, WEIGHT=""),
        OUTPUT_DIR=""))
    return c


_C = _defaults()
cfg = _C


def get_cfg_defaults():
    return _C.clone()
