"""Configuration with the reference's keys and defaults (nmrf/config/default.py:20-175).

A small self-contained CfgNode (attribute dict with clone / freeze / merge_from_list /
merge_from_file incl. the `__BASE__` inheritance of nmrf/config/config.py:52-115) so the drop-in
needs neither yacs nor omegaconf.  Only the keys the inference path reads matter; the training
keys are carried so that the reference's yaml files merge without error.
"""
import ast
import copy
import os

import yaml

BASE_KEY = "__BASE__"


class CfgNode(dict):
    _FROZEN = "__frozen__"

    def __init__(self, init=None):
        super().__init__()
        self.__dict__[CfgNode._FROZEN] = False
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.__dict__[CfgNode._FROZEN]:
            raise AttributeError("Attempted to set %s on a frozen CfgNode" % name)
        self[name] = value

    def __deepcopy__(self, memo):
        new = CfgNode()
        for k, v in self.items():
            dict.__setitem__(new, k, copy.deepcopy(v, memo))
        return new

    def clone(self):
        return copy.deepcopy(self)

    def _freeze(self, flag):
        self.__dict__[CfgNode._FROZEN] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._freeze(flag)

    def freeze(self):
        self._freeze(True)

    def defrost(self):
        self._freeze(False)

    def is_frozen(self):
        return self.__dict__[CfgNode._FROZEN]

    def merge_from_other_cfg(self, other):
        for k, v in other.items():
            if k not in self:
                raise KeyError("Non-existent config key: %s" % k)
            if isinstance(v, dict) and isinstance(self[k], CfgNode):
                self[k].merge_from_other_cfg(v)
            else:
                self[k] = v

    @staticmethod
    def load_yaml_with_base(filename):
        with open(filename, "r") as f:
            cfg = yaml.safe_load(f) or {}

        def merge(a, b):
            for k, v in a.items():
                if isinstance(v, dict) and isinstance(b.get(k), dict):
                    merge(v, b[k])
                else:
                    b[k] = v

        if BASE_KEY in cfg:
            bases = cfg.pop(BASE_KEY)
            merged = {}
            for base in (bases if isinstance(bases, list) else [bases]):
                base = os.path.expanduser(base)
                if not os.path.isabs(base):
                    base = os.path.join(os.path.dirname(filename), base)
                merge(CfgNode.load_yaml_with_base(base), merged)
            merge(cfg, merged)
            return merged
        return cfg

    def merge_from_file(self, filename, allow_unsafe=True):
        assert os.path.isfile(filename), "Config file '%s' does not exist!" % filename
        self.merge_from_other_cfg(CfgNode(self.load_yaml_with_base(filename)))

    def merge_from_list(self, opts):
        opts = list(opts)
        assert len(opts) % 2 == 0, "override list must be KEY VALUE pairs"
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node[p]
            if parts[-1] not in node:
                raise KeyError("Non-existent config key: %s" % key)
            if isinstance(val, str):
                try:
                    val = ast.literal_eval(val)
                except (ValueError, SyntaxError):
                    pass
            old = node[parts[-1]]
            if old is not None and val is not None and not isinstance(old, (list, tuple)) \
                    and not isinstance(val, type(old)):
                val = type(old)(val)
            node[parts[-1]] = val


def _defaults():
    c = CfgNode()
    c.VERSION = 2
    c.BACKBONE = CfgNode(dict(MODEL_TYPE="resnet", NORM_FN="instance", OUT_CHANNELS=256, WEIGHT_URL="",
                              DROP_PATH=0.0, COMPAT=True))
    c.DPN = CfgNode(dict(MAX_DISP=320, COST_GROUP=4, NUM_PROPOSALS=4, CONTEXT_DIM=64))
    c.NMP = CfgNode(dict(PROP_EMBED_DIM=128, INFER_EMBED_DIM=128, MLP_RATIO=4, SPLIT_SIZE=1, WINDOW_SIZE=6,
                         REFINE_WINDOW_SIZE=4, PROP_N_HEADS=4, INFER_N_HEADS=4, NUM_PROP_LAYERS=5,
                         NUM_INFER_LAYERS=5, NUM_REFINE_LAYERS=5, RETURN_INTERMEDIATE=True, ATTN_DROP=0.0,
                         PROJ_DROP=0.0, DROP_PATH=0.0, DROPOUT=0.0, NORMALIZE_BEFORE=True, WITH_REFINEMENT=True))
    c.DATASETS = CfgNode(dict(TRAIN=["sceneflow"], TEST=["things"], IMG_GAMMA=None, SATURATION_RANGE=[0, 1.4],
                              DO_FLIP=False, SPATIAL_SCALE=[-0.2, 0.4], YJITTER=False, CROP_SIZE=[384, 768],
                              DIVIS_BY=8))
    c.DATALOADER = CfgNode(dict(NUM_WORKERS=4))
    c.SOLVER = CfgNode(dict(MAX_ITER=300000, BASE_LR=0.0005, BASE_LR_END=0.0, BACKBONE_LR_DECAY=0.1,
                            WEIGHT_DECAY=0.00001, WEIGHT_DECAY_NORM=0.00001, BACKBONE_WEIGHT_DECAY=0.00001,
                            CHECKPOINT_PERIOD=100000, LATEST_CHECKPOINT_PERIOD=1000, IMS_PER_BATCH=8, GRAD_CLIP=1.0,
                            LOSS_WEIGHTS=[1.0, 1.0, 1.0, 1.4, 1.4, 1.4, 1.4, 1.6, 2.0, 2.0], RESUME=None,
                            STRICT_RESUME=True, NO_RESUME_OPTIMIZER=False, AUX_LOSS=True, MAX_DISP=192,
                            LOSS_TYPE="L1"))
    c.TEST = CfgNode(dict(EVAL_PERIOD=20000, EVAL_THRESH=[["1.0", "3.0"]], EVAL_MAX_DISP=[192],
                          EVAL_ONLY_VALID=[True], EVAL_PROP=[True]))
    c.SEED = 326
    c.CUDNN_BENCHMARK = True
    c.GLOBAL = CfgNode(dict(HACK=1.0))
    return c


def get_cfg():
    """A fresh copy of the default config (nmrf/config/config.py:183-192)."""
    return _defaults()
