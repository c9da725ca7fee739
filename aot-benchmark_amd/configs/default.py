"""Engine config for the inference path.  Mirrors ``DefaultEngineConfig`` of the
reference (configs/default.py:5-107) for the attributes the hot path and its callers
read; unlike the reference it never creates directories (``init_dir`` :109-138 is
training/IO plumbing and out of scope)."""
import importlib

# what the training-side slice reads (losses, schedule, optimiser, EMA, forward): the reference's names and defaults
# (configs/default.py:37-78)
_TRAIN = dict(
    TRAIN_TOTAL_STEPS=100000, TRAIN_START_STEP=0, TRAIN_WEIGHT_DECAY=0.07, TRAIN_WEIGHT_DECAY_EXCLUSIVE={},
    TRAIN_WEIGHT_DECAY_EXEMPTION=['absolute_pos_embed', 'relative_position_bias_table', 'relative_emb_v', 'conv_out'],
    TRAIN_LR=2e-4, TRAIN_LR_POWER=0.9, TRAIN_LR_ENCODER_RATIO=0.1, TRAIN_LR_WARM_UP_RATIO=0.05, TRAIN_LR_COSINE_DECAY=False,
    TRAIN_LR_RESTART=1, TRAIN_LR_UPDATE_STEP=1, TRAIN_AUX_LOSS_WEIGHT=1.0, TRAIN_AUX_LOSS_RATIO=1.0, TRAIN_OPT='adamw',
    TRAIN_BATCH_SIZE=16, TRAIN_TOP_K_PERCENT_PIXELS=0.15, TRAIN_SEQ_TRAINING_FREEZE_PARAMS=['patch_wise_id_bank'],
    TRAIN_SEQ_TRAINING_START_RATIO=0.5, TRAIN_HARD_MINING_RATIO=0.5, TRAIN_EMA_RATIO=0.1, TRAIN_CLIP_GRAD_NORM=5.,
    TRAIN_ENABLE_PREV_FRAME=False,
)


class DefaultEngineConfig():
    def __init__(self, exp_name='default', model='aott'):
        model_cfg = importlib.import_module('configs.models.' + model).ModelConfig()
        self.__dict__.update(model_cfg.__dict__)
        self.EXP_NAME = exp_name + '_' + self.MODEL_NAME
        self.TEST_GPU_ID = 0
        self.TEST_GPU_NUM = 1
        self.TEST_CKPT_PATH = None
        self.TEST_FLIP = False
        self.TEST_MULTISCALE = [1]
        self.TEST_MAX_SHORT_EDGE = None
        self.TEST_MAX_LONG_EDGE = 800 * 1.3
        self.DIST_BACKEND = 'nccl'  # RCCL on ROCm
        for k, v in _TRAIN.items():
            setattr(self, k, type(v)(v) if isinstance(v, (list, dict)) else v)
        self.TRAIN_LR_MIN = 2e-5 if 'mobilenetv2' in self.MODEL_ENCODER else 1e-5


EngineConfig = DefaultEngineConfig
