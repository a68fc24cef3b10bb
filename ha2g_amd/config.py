"""Hyper-parameters of the hierarchy train step, restated as data.

Values are those of the reference's config/hierarchy.yml:19-47 (TED-Gesture, 9 bones / 27-d) and
config_expressive/hierarchy.yml:19-48 (TED-Expressive, 42 bones / 126-d), plus the parse_args.py
defaults that the step reads (dropout_prob :39, discriminator_lr_weight :60, freeze_wordembed :33).
`mean_dir_vec` keeps configargparse's shape quirk -- a list of 1-element lists, i.e. (P,1) -- because
train_hierarchy.py:245 does `.squeeze(1)` on it.
"""
from types import SimpleNamespace

MEAN_DIR_VEC_GESTURE = [
    0.0154009, -0.9690125, -0.0884354, -0.0022264, -0.8655276, 0.4342174, -0.0035145, -0.8755367, -0.4121039,
    -0.9236511, 0.3061306, -0.0012415, -0.5155854, 0.8129665, 0.0871897, 0.2348464, 0.1846561, 0.8091402, 0.9271948,
    0.2960011, -0.013189, 0.5233978, 0.8092403, 0.0725451, -0.2037076, 0.1924306, 0.8196916
]

MEAN_DIR_VEC_EXPRESSIVE = [
    -0.0737964, -0.9968923, -0.1082858, 0.9111595, 0.2399522, -0.102547, -0.8936886, 0.3131501, -0.1039348,
    0.2093927, 0.958293, 0.0824881, -0.1689021, -0.0353824, -0.7588258, -0.2794763, -0.2495191, -0.614666,
    -0.3877234, 0.005006, -0.5301695, -0.5098616, 0.2257808, 0.0053111, -0.2393621, -0.1022204, -0.6583039,
    -0.4992898, 0.1228059, -0.3292085, -0.4753748, 0.2132857, 0.1742853, -0.2062069, 0.2305175, -0.5897119,
    -0.5452555, 0.1303197, -0.2181693, -0.5221036, 0.1211322, 0.1337591, -0.2164441, 0.0743345, -0.6464546,
    -0.5284583, 0.0457585, -0.319634, -0.5074904, 0.1537192, 0.1365934, -0.4354402, -0.3836682, -0.3850554,
    -0.4927187, -0.2417618, -0.3054556, -0.3556116, -0.281753, -0.5164358, -0.3064435, 0.9284261, -0.067134,
    0.2764367, 0.006997, -0.7365526, 0.2421269, -0.225798, -0.6387642, 0.3788997, 0.0283412, -0.5451686, 0.5753376,
    0.1935219, 0.0632555, 0.2122412, -0.0624179, -0.6755542, 0.5212831, 0.1043523, -0.345288, 0.5443628, 0.128029,
    0.2073687, 0.2197118, 0.2821399, -0.580695, 0.573988, 0.0786667, -0.2133071, 0.5532452, -0.0006157, 0.1598754,
    0.2093099, 0.124119, -0.6504359, 0.5465003, 0.0114155, -0.3203954, 0.5512083, 0.0489287, 0.1676814, 0.4190787,
    -0.4018607, -0.3912126, 0.4841548, -0.2668508, -0.3557675, 0.3416916, -0.2419564, -0.5509825, 0.0485515,
    -0.6343101, -0.6817347, -0.4705639, -0.6380668, 0.4641643, 0.4540192, -0.6486361, 0.4604001, -0.3256226,
    0.1883097, 0.8057457, 0.3257385, 0.1292366, 0.815372
]

_COMMON = dict(
    model='hierarchy', wordembed_dim=300, freeze_wordembed=False, n_layers=4, hidden_size=300,
    z_type='speaker', input_context='both', dropout_prob=0.3, learning_rate=5e-4,
    discriminator_lr_weight=0.2, loss_gan_weight=5.0, loss_warmup=10, loss_kld_weight=0.1,
    loss_reg_weight=0.05, loss_contrastive_pos_weight=0.2, loss_contrastive_neg_weight=0.005,
    loss_physical_weight=0.01, n_poses=34, n_pre_poses=4, motion_resampling_framerate=15,
)


def hierarchy_args(expressive=False, **overrides):
    """Namespace equivalent to parse_args() on config[_expressive]/hierarchy.yml."""
    d = dict(_COMMON)
    if expressive:
        d.update(pose_dim=126, loss_regression_weight=250.0, batch_size=96,
                 mean_dir_vec=[[v] for v in MEAN_DIR_VEC_EXPRESSIVE])
    else:
        d.update(pose_dim=27, loss_regression_weight=70.0, batch_size=256,
                 mean_dir_vec=[[v] for v in MEAN_DIR_VEC_GESTURE])
    d.update(overrides)
    return SimpleNamespace(**d)


# Parity cases shared by tests/golden/gen_golden.py (reference side) and tests/ (this side).
# dropout is 0 in every parity case; n_words / n_spk are small because embeddings are
# pure gathers -- their size does not change the arithmetic.
CASES = {
    # reduced widths: fast unit-level pins of every module and of the whole step
    'small': dict(B=3, hidden_size=32, n_layers=2, n_words=40, n_spk=6, seed=11),
    # config 1 of BASELINE.json: hierarchy.yml shapes, B=4
    'cfg1': dict(B=4, hidden_size=300, n_layers=4, n_words=60, n_spk=8, seed=12),
    # TED-Expressive twin (6 levels, P=126), step only
    'expr_small': dict(B=3, hidden_size=32, n_layers=2, n_words=40, n_spk=6, seed=13, expressive=True),
}


def make_args(case):
    return hierarchy_args(expressive=bool(case.get('expressive')), hidden_size=case['hidden_size'],
                          n_layers=case['n_layers'], dropout_prob=0.0)

# Physical-angle prior statistics (bone-pair mean/variance of angle/pi), data restated from
# train_eval/train_hierarchy.py:9-16.  Pairs are (3,4),(4,5),(6,7),(7,8).
PHYS_GESTURE = (
    (0.22037504613399506, 0.4590071439743042, 0.22463147342205048, 0.45562979578971863),
    (0.0018439559498801827, 0.013570506125688553, 0.0017794054001569748, 0.013684595935046673),
)
PHYS_GESTURE_PAIRS = ((3, 4), (4, 5), (6, 7), (7, 8))
