"""Model presets as data.

The models and engines read ``cfg.MODEL_*`` / ``cfg.TEST_*`` attributes whose NAMES are the reference's config contract
(configs/models/default.py, default_deaot.py and the per-model files); this module keeps them in two tables -- the shared
defaults and one small override dict per model of the zoo -- and builds the ``configs.models.<name>.ModelConfig`` classes
the reference's callers import from them.
"""

_COMMON = dict(
    # architecture
    MODEL_VOS='aot', MODEL_ENGINE='aotengine', MODEL_NAME='AOTDefault',
    MODEL_ENCODER='mobilenetv2', MODEL_ENCODER_PRETRAIN='./pretrain_models/mobilenet_v2-b0353104.pth', MODEL_ENCODER_DIM=[24, 32, 96, 1280],   # 4x, 8x, 16x, 16x
    MODEL_ENCODER_EMBEDDING_DIM=256, MODEL_LSTT_NUM=1, MODEL_SELF_HEADS=8, MODEL_ATT_HEADS=8,
    MODEL_DECODER_INTERMEDIATE_LSTT=True, MODEL_MAX_OBJ_NUM=10, MODEL_ALIGN_CORNERS=True,
    MODEL_FREEZE_BN=True, MODEL_FREEZE_BACKBONE=False, MODEL_EPSILON=1e-5, MODEL_USE_PREV_PROB=False,
    # memory schedule
    TEST_LONG_TERM_MEM_GAP=9999, TEST_SHORT_TERM_MEM_SKIP=1, TRAIN_LONG_TERM_MEM_GAP=9999, TRAIN_AUG_TYPE='v1',
    # read by the model constructors (reference configs/default.py:79-86); training-only dropouts, identity at inference
    TRAIN_ENCODER_FREEZE_AT=2, TRAIN_LSTT_EMB_DROPOUT=0., TRAIN_LSTT_ID_DROPOUT=0., TRAIN_LSTT_DROPPATH=0.1,
    TRAIN_LSTT_DROPPATH_SCALING=False, TRAIN_LSTT_DROPPATH_LST=False, TRAIN_LSTT_LT_DROPOUT=0., TRAIN_LSTT_ST_DROPOUT=0.,
)
_DEAOT = dict(MODEL_VOS='deaot', MODEL_ENGINE='deaotengine', MODEL_NAME='DeAOTDefault', MODEL_DECODER_INTERMEDIATE_LSTT=False,
              MODEL_SELF_HEADS=1, MODEL_ATT_HEADS=1, TRAIN_AUG_TYPE='v2')
_LARGE = dict(MODEL_LSTT_NUM=3, TRAIN_LONG_TERM_MEM_GAP=2, TEST_LONG_TERM_MEM_GAP=5)      # the "L" memory schedule
_R50 = dict(MODEL_ENCODER='resnet50', MODEL_ENCODER_DIM=[256, 512, 1024, 1024],
            MODEL_ENCODER_PRETRAIN='./pretrain_models/resnet50-0676ba61.pth')
_R101 = dict(MODEL_ENCODER='resnet101', MODEL_ENCODER_DIM=[256, 512, 1024, 1024],
             MODEL_ENCODER_PRETRAIN='./pretrain_models/resnet101-63fe2227.pth')
_SWINB = dict(MODEL_ENCODER='swin_base', MODEL_ENCODER_DIM=[128, 256, 512, 512], MODEL_ALIGN_CORNERS=False,
              MODEL_ENCODER_PRETRAIN='./pretrain_models/swin_base_patch4_window7_224_22k.pth')

# name -> (display name, DeAOT?, override dicts applied in order)
ZOO = {
    'aott': ('AOTT', False, ()),
    'aots': ('AOTS', False, (dict(MODEL_LSTT_NUM=2),)),
    'aotb': ('AOTB', False, (dict(MODEL_LSTT_NUM=3),)),
    'aotl': ('AOTL', False, (_LARGE,)),
    'r50_aotl': ('R50_AOTL', False, (_LARGE, _R50)),
    'r101_aotl': ('R101_AOTL', False, (_LARGE, _R101)),
    'swinb_aotl': ('SwinB_AOTL', False, (_LARGE, _SWINB)),
    'deaott': ('DeAOTT', True, ()),
    'deaots': ('DeAOTS', True, (dict(MODEL_LSTT_NUM=2),)),
    'deaotb': ('DeAOTB', True, (dict(MODEL_LSTT_NUM=3),)),
    'deaotl': ('DeAOTL', True, (_LARGE,)),
    'r50_deaotl': ('R50_DeAOTL', True, (_LARGE, _R50)),
    'swinb_deaotl': ('SwinB_DeAOTL', True, (_LARGE, _SWINB)),
}


def _fill(obj, *tables):
    for t in tables:
        for k, v in t.items():
            setattr(obj, k, list(v) if isinstance(v, list) else v)


class DefaultModelConfig:
    def __init__(self):
        _fill(self, _COMMON)


class DefaultDeAOTModelConfig(DefaultModelConfig):
    def __init__(self):
        super().__init__()
        _fill(self, _DEAOT)


def preset(name):
    """The ``ModelConfig`` class of zoo entry ``name`` (what ``configs/models/<name>.py`` exports)."""
    display, deaot, overrides = ZOO[name]
    base = DefaultDeAOTModelConfig if deaot else DefaultModelConfig

    def __init__(self):
        base.__init__(self)
        _fill(self, *overrides)
        self.MODEL_NAME = display
    return type('ModelConfig', (base,), {'__init__': __init__, '__doc__': '%s preset' % display})
