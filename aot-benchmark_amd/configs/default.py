"""Engine configs as data.  Same attribute names and values as ``DefaultEngineConfig`` of the reference
(configs/default.py:5-107) and its stage files (configs/{pre,pre_dav,pre_ytb,pre_ytb_dav,ytb}.py) -- callers pick a stage with
``importlib.import_module('configs.' + stage).EngineConfig(exp_name, model)`` (tools/eval.py, tools/demo.py) -- kept as one
table of defaults and one small override table per stage.  Unlike the reference, constructing a config touches no directory:
the DIR_* attributes are plain strings, and ``init_dir(create=True)`` makes them on request (reference :109-138 always does)."""
import copy
import importlib
import os

_DEFAULTS = dict(
    STAGE_NAME='YTB', DATASETS=['youtubevos'],
    # sampling / augmentation of the training clips
    DATA_WORKERS=8, DATA_RANDOMFLIP=0.5, DATA_MAX_CROP_STEPS=10, DATA_SHORT_EDGE_LEN=480, DATA_MIN_SCALE_FACTOR=0.7,
    DATA_MAX_SCALE_FACTOR=1.3, DATA_RANDOM_REVERSE_SEQ=True, DATA_SEQ_LEN=5, DATA_DAVIS_REPEAT=5, DATA_RANDOM_GAP_DAVIS=12,
    DATA_RANDOM_GAP_YTB=3, DATA_DYNAMIC_MERGE_PROB=0.3,
    PRETRAIN=True, PRETRAIN_FULL=False, PRETRAIN_MODEL='./data_wd/pretrain_model/mobilenet_v2.pth',
    # optimisation
    TRAIN_TOTAL_STEPS=100000, TRAIN_START_STEP=0, TRAIN_WEIGHT_DECAY=0.07, TRAIN_WEIGHT_DECAY_EXCLUSIVE={},
    TRAIN_WEIGHT_DECAY_EXEMPTION=['absolute_pos_embed', 'relative_position_bias_table', 'relative_emb_v', 'conv_out'],
    TRAIN_LR=2e-4, TRAIN_LR_POWER=0.9, TRAIN_LR_ENCODER_RATIO=0.1, TRAIN_LR_WARM_UP_RATIO=0.05, TRAIN_LR_COSINE_DECAY=False,
    TRAIN_LR_RESTART=1, TRAIN_LR_UPDATE_STEP=1, TRAIN_AUX_LOSS_WEIGHT=1.0, TRAIN_AUX_LOSS_RATIO=1.0, TRAIN_OPT='adamw',
    TRAIN_SGD_MOMENTUM=0.9, TRAIN_GPUS=4, TRAIN_BATCH_SIZE=16, TRAIN_TBLOG=False, TRAIN_TBLOG_STEP=50, TRAIN_LOG_STEP=20,
    TRAIN_IMG_LOG=True, TRAIN_TOP_K_PERCENT_PIXELS=0.15, TRAIN_SEQ_TRAINING_FREEZE_PARAMS=['patch_wise_id_bank'],
    TRAIN_SEQ_TRAINING_START_RATIO=0.5, TRAIN_HARD_MINING_RATIO=0.5, TRAIN_EMA_RATIO=0.1, TRAIN_CLIP_GRAD_NORM=5.,
    TRAIN_SAVE_STEP=5000, TRAIN_MAX_KEEP_CKPT=8, TRAIN_RESUME=False, TRAIN_RESUME_CKPT=None, TRAIN_RESUME_STEP=0,
    TRAIN_AUTO_RESUME=True, TRAIN_DATASET_FULL_RESOLUTION=False, TRAIN_ENABLE_PREV_FRAME=False,
    # read by the model constructors
    TRAIN_ENCODER_FREEZE_AT=2, TRAIN_LSTT_EMB_DROPOUT=0., TRAIN_LSTT_ID_DROPOUT=0., TRAIN_LSTT_DROPPATH=0.1,
    TRAIN_LSTT_DROPPATH_SCALING=False, TRAIN_LSTT_DROPPATH_LST=False, TRAIN_LSTT_LT_DROPOUT=0., TRAIN_LSTT_ST_DROPOUT=0.,
    # evaluation
    TEST_GPU_ID=0, TEST_GPU_NUM=1, TEST_FRAME_LOG=False, TEST_DATASET='youtubevos', TEST_DATASET_FULL_RESOLUTION=False,
    TEST_DATASET_SPLIT='val', TEST_CKPT_PATH=None, TEST_CKPT_STEP=None, TEST_FLIP=False, TEST_MULTISCALE=[1],
    TEST_MAX_SHORT_EDGE=None, TEST_MAX_LONG_EDGE=800 * 1.3, TEST_WORKERS=4,
    # processes: one per GPU, backend "nccl" is RCCL on ROCm
    DIST_ENABLE=True, DIST_BACKEND='nccl', DIST_URL='tcp://127.0.0.1:13241', DIST_START_GPU=0,
)

# the stage a run resumes from: full checkpoint of that stage's EMA weights
_FROM_PRE = dict(PRETRAIN_FULL=True, _PRETRAIN_STAGE='PRE', _PRETRAIN_CKPT='save_step_100000.pth')
STAGES = {
    'default': None,                                     # DefaultEngineConfig itself: no directories
    'pre': dict(STAGE_NAME='PRE', DATASETS=['static'], DATA_DYNAMIC_MERGE_PROB=1.0, TRAIN_LR=4e-4, TRAIN_LR_MIN=2e-5,
                TRAIN_WEIGHT_DECAY=0.03, TRAIN_SEQ_TRAINING_START_RATIO=1.0, TRAIN_AUX_LOSS_RATIO=0.1),
    'pre_dav': dict(_FROM_PRE, STAGE_NAME='PRE_DAV', DATASETS=['davis2017'], TRAIN_TOTAL_STEPS=50000),
    'pre_ytb': dict(_FROM_PRE, STAGE_NAME='PRE_YTB'),
    'pre_ytb_dav': dict(_FROM_PRE, STAGE_NAME='PRE_YTB_DAV', DATASETS=['youtubevos', 'davis2017']),
    'ytb': dict(STAGE_NAME='YTB'),
}


class DefaultEngineConfig:
    def __init__(self, exp_name='default', model='aott'):
        model_cfg = importlib.import_module('configs.models.' + model).ModelConfig()
        self.__dict__.update(model_cfg.__dict__)
        self.EXP_NAME = exp_name + '_' + self.MODEL_NAME
        for k, v in _DEFAULTS.items():
            setattr(self, k, copy.deepcopy(v))
        self.DATA_RANDOMCROP = (465, 465) if self.MODEL_ALIGN_CORNERS else (464, 464)
        self.TRAIN_LR_MIN = 2e-5 if 'mobilenetv2' in self.MODEL_ENCODER else 1e-5

    def init_dir(self, create=False):
        """Fills the DIR_* attributes (reference :109-138); directories are only made when asked to."""
        self.DIR_DATA = '../VOS02/datasets'
        for name, sub in (('DIR_DAVIS', 'DAVIS'), ('DIR_YTB', 'YTB'), ('DIR_STATIC', 'Static')):
            setattr(self, name, os.path.join(self.DIR_DATA, sub))
        self.DIR_ROOT = './'
        self.DIR_RESULT = os.path.join(self.DIR_ROOT, 'result', self.EXP_NAME, self.STAGE_NAME)
        for name, sub in (('DIR_CKPT', ('ckpt',)), ('DIR_EMA_CKPT', ('ema_ckpt',)), ('DIR_LOG', ('log',)),
                          ('DIR_TB_LOG', ('log', 'tensorboard'))):
            setattr(self, name, os.path.join(self.DIR_RESULT, *sub))
        self.DIR_IMG_LOG = './img_logs'
        self.DIR_EVALUATION = './results'
        if create:
            for d in (self.DIR_RESULT, self.DIR_CKPT, self.DIR_EMA_CKPT, self.DIR_LOG, self.DIR_EVALUATION, self.DIR_IMG_LOG,
                      self.DIR_TB_LOG):
                os.makedirs(d, exist_ok=True)


def stage(name):
    """The ``EngineConfig`` class of training / evaluation stage ``name`` (what ``configs/<name>.py`` exports)."""
    table = STAGES[name]

    def __init__(self, exp_name='default', model='AOTT'):
        DefaultEngineConfig.__init__(self, exp_name, model)
        head = {k: v for k, v in table.items() if not k.startswith('_')}
        self.STAGE_NAME = head.pop('STAGE_NAME')
        self.init_dir()
        for k, v in head.items():
            setattr(self, k, copy.deepcopy(v))
        if '_PRETRAIN_STAGE' in table:
            self.PRETRAIN_MODEL = os.path.join(self.DIR_ROOT, 'result', self.EXP_NAME, table['_PRETRAIN_STAGE'], 'ema_ckpt',
                                               table['_PRETRAIN_CKPT'])
    return type('EngineConfig', (DefaultEngineConfig,), {'__init__': __init__, '__doc__': 'stage %s' % name})


EngineConfig = DefaultEngineConfig
