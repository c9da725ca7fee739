"""Model presets.  Attribute names follow the reference (configs/models/default.py:1-27,
default_deaot.py:4-17) because models and engines read ``cfg.MODEL_*`` / ``cfg.TEST_*``."""


class DefaultModelConfig():
    def __init__(self):
        self.MODEL_NAME = 'AOTDefault'
        self.MODEL_VOS = 'aot'
        self.MODEL_ENGINE = 'aotengine'
        self.MODEL_ALIGN_CORNERS = True
        self.MODEL_ENCODER = 'mobilenetv2'
        self.MODEL_ENCODER_PRETRAIN = ''
        self.MODEL_ENCODER_DIM = [24, 32, 96, 1280]  # 4x, 8x, 16x, 16x
        self.MODEL_ENCODER_EMBEDDING_DIM = 256
        self.MODEL_DECODER_INTERMEDIATE_LSTT = True
        self.MODEL_FREEZE_BN = True
        self.MODEL_FREEZE_BACKBONE = False
        self.MODEL_MAX_OBJ_NUM = 10
        self.MODEL_SELF_HEADS = 8
        self.MODEL_ATT_HEADS = 8
        self.MODEL_LSTT_NUM = 1
        self.MODEL_EPSILON = 1e-5
        self.MODEL_USE_PREV_PROB = False
        self.TRAIN_LONG_TERM_MEM_GAP = 9999
        self.TRAIN_AUG_TYPE = 'v1'
        self.TEST_LONG_TERM_MEM_GAP = 9999
        self.TEST_SHORT_TERM_MEM_SKIP = 1
        # engine-level attributes the model constructor reads (reference configs/default.py:79-86);
        # all of them are training-only dropouts, identity on the inference path
        self.TRAIN_ENCODER_FREEZE_AT = 2
        self.TRAIN_LSTT_EMB_DROPOUT = 0.
        self.TRAIN_LSTT_ID_DROPOUT = 0.
        self.TRAIN_LSTT_DROPPATH = 0.1
        self.TRAIN_LSTT_DROPPATH_SCALING = False
        self.TRAIN_LSTT_DROPPATH_LST = False
        self.TRAIN_LSTT_LT_DROPOUT = 0.
        self.TRAIN_LSTT_ST_DROPOUT = 0.


class DefaultDeAOTModelConfig(DefaultModelConfig):
    def __init__(self):
        super().__init__()
        self.MODEL_NAME = 'DeAOTDefault'
        self.MODEL_VOS = 'deaot'
        self.MODEL_ENGINE = 'deaotengine'
        self.MODEL_DECODER_INTERMEDIATE_LSTT = False
        self.MODEL_SELF_HEADS = 1
        self.MODEL_ATT_HEADS = 1
        self.TRAIN_AUG_TYPE = 'v2'
