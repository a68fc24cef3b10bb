"""Validation pass of the hierarchy model -- MI355X-native mirror of scripts/train.py:326-500 `evaluate_testset` (TED-Gesture, model ==
'hierarchy'): eval-mode audio encoder + coarse-to-fine generators per test batch, L1 loss, FGD evaluator push, joint MAE and
acceleration difference; returns the reference's dict {'loss', 'joint_mae'[, 'frechet', 'feat_dist', 'diversity', 'bc']}.

TED-Expressive (scripts/train_expressive.py:394-626, six generators + MotionAE evaluator) is the same loop; note that the reference's
expressive loop imports the TED-Gesture `convert_dir_vec_to_pose` (train_expressive.py:23), so its joint MAE / acceleration walk the
10-joint chain over the first nine of the 42 direction vectors -- reproduced here as is (tests/golden/evalset.npz pins both).

The forward (encoders, generators, pre_seq pack, L1) and the evaluator run on the device; only the per-batch joint-space metrics follow
the reference to the host (numpy, `convert_dir_vec_to_pose`).  The beat-consistency branch is disabled in the reference
(`beat_consistency_score = False`, :343) and therefore reports bc = 0 here too.
"""
import random
import time

import numpy as np
import torch

from . import ops
from .config import EXPRESSIVE_SPEC, GESTURE_SPEC

# adjacency and bone length, scripts/utils/data_utils.py:14-15
dir_vec_pairs = [(0, 1, 0.26), (1, 2, 0.18), (2, 3, 0.14), (1, 4, 0.22), (4, 5, 0.36), (5, 6, 0.33), (1, 7, 0.22), (7, 8, 0.36), (8, 9, 0.33)]


def convert_dir_vec_to_pose(vec):
    """Unit direction vectors (…, 27) or (…, 9, 3) -> joint positions (…, 10, 3) by walking the bone chain (data_utils.py:77-98)."""
    vec = np.array(vec)
    if vec.shape[-1] != 3:
        vec = vec.reshape(vec.shape[:-1] + (-1, 3))
    joint_pos = np.zeros(vec.shape[:-2] + (10, 3))
    for j, pair in enumerate(dir_vec_pairs):
        joint_pos[..., pair[1], :] = joint_pos[..., pair[0], :] + pair[2] * vec[..., j, :]
    return joint_pos


class _Meter:
    def __init__(self):
        self.sum, self.count, self.avg = 0, 0, 0

    def update(self, val, n=1):
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def evaluate_testset(test_data_loader, gens, audio_encoder, embed_space_evaluator, args, device=None, vid_source=None):
    """test_data_loader yields the loader's 8-tuples (in_text, text_lengths, in_text_padded, pose_seq, vec_seq, in_audio, in_spec, aux_info),
    scripts/data_loader/lmdb_data_loader.py:45-55.  vid_source(speaker_model, batch_size) -> list of speaker ids (default: the
    reference's random.choice over the speaker vocabulary, :363-365)."""
    spec = GESTURE_SPEC if len(gens) == 3 else EXPRESSIVE_SPEC
    device = device or next(gens[0].parameters()).device
    for m in list(gens) + [audio_encoder]:
        m.train(False)
    if embed_space_evaluator:
        embed_space_evaluator.reset()
    losses, joint_mae, accel = _Meter(), _Meter(), _Meter()
    start = time.time()
    dims = spec['pose_dims']
    cols = [torch.tensor(c, dtype=torch.long, device=device) for c in spec['level_cols']]
    tables = [ops.scatter_tables(P, dims[k - 1] if k else 0, spec['scatter'][k], device) for k, P in enumerate(dims)]
    mean_dir = np.array(args.mean_dir_vec).squeeze()
    with torch.no_grad():
        for data in test_data_loader:
            _, _, in_text_padded, _, target_vec, in_audio, in_spec, _ = data
            batch_size = target_vec.size(0)
            in_text_padded = in_text_padded.to(device)
            in_spec = in_spec.float().to(device)
            target = target_vec.to(device).float()
            g0 = gens[0].module if hasattr(gens[0], 'module') else gens[0]
            speaker_model = getattr(g0, 'z_obj', None)
            if speaker_model is not None and hasattr(speaker_model, 'word2index'):
                vids = vid_source(speaker_model, batch_size) if vid_source else \
                    [random.choice(list(speaker_model.word2index.values())) for _ in range(batch_size)]
                vid_indices = torch.LongTensor(vids).to(device)
            else:
                vid_indices = None
            _, _, _, _, blend = audio_encoder(in_spec, vid_indices)
            prev = None
            for k, g in enumerate(gens):                  # train.py:378-415: per-level targets, pre_seq, coarse-to-fine scatter
                tk = target if len(cols[k]) == dims[-1] else target.index_select(2, cols[k])
                prev, *_ = g(ops.pre_seq(tk, prev, tables[k], args.n_pre_poses), in_text_padded, blend[k], vid_indices)
            out_dir_vec = prev
            loss = ops.eltwise(ops.OP_AXPBY, out_dir_vec.contiguous(), target.contiguous(), alpha=1.0, beta=-1.0).abs().mean()   # F.l1_loss
            losses.update(loss.item(), batch_size)
            if embed_space_evaluator:
                embed_space_evaluator.push_samples(in_text_padded, in_audio, out_dir_vec, target)
            out_np = out_dir_vec.cpu().numpy()
            out_np = out_np + mean_dir
            out_joint_poses = convert_dir_vec_to_pose(out_np)
            tgt_np = target_vec.cpu().numpy()
            tgt_np = tgt_np + mean_dir
            target_poses = convert_dir_vec_to_pose(tgt_np)
            if out_joint_poses.shape[1] == args.n_poses:
                diff = out_joint_poses[:, args.n_pre_poses:] - target_poses[:, args.n_pre_poses:]
            else:
                diff = out_joint_poses - target_poses[:, args.n_pre_poses:]
            joint_mae.update(np.mean(np.absolute(diff)), batch_size)
            accel.update(np.mean(np.abs(np.diff(target_poses, n=2, axis=1) - np.diff(out_joint_poses, n=2, axis=1))), batch_size)
    for m in list(gens) + [audio_encoder]:
        m.train(True)
    ret_dict = {'loss': losses.avg, 'joint_mae': joint_mae.avg}
    if embed_space_evaluator and embed_space_evaluator.get_no_of_samples() > 0:
        frechet_dist, feat_dist = embed_space_evaluator.get_scores()
        ret_dict['frechet'] = frechet_dist
        ret_dict['feat_dist'] = feat_dist
        ret_dict['diversity'] = embed_space_evaluator.get_diversity_scores()
        ret_dict['bc'] = 0
    ret_dict['_accel'] = accel.avg                          # logged, not returned, by the reference
    ret_dict['_elapsed'] = time.time() - start
    return ret_dict
