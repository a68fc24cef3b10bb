"""Hierarchical inference over a long clip: sliding 34-frame windows with a 4-frame overlap, each window decoded coarse to fine
(audio encoder -> g1 -> g2 -> g3 in eval mode), the first 4 frames of a window constrained to the last 4 of the previous one, and
the overlap cross-faded -- the MI355X-native mirror of scripts/synthesize_hierarchy.py:36-215 `generate_gestures_hierarchy`
(SURVEY 8 f1).  Windows are sequentially dependent (window i's seed poses are window i-1's output), so the loop stays on the host;
everything inside it -- the mel front-end, the encoders/generators, the pre_seq pack, the cross-fade -- runs in HIP kernels and the
sequence never leaves the device until the end.
"""
import math

import numpy as np
import torch

from . import ops
from ._lib import check, lib
from .config import EXPRESSIVE_SPEC, GESTURE_SPEC


def calc_spectrogram_length_from_motion_length(n_frames, fps):
    """scripts/utils/data_utils.py:41-43"""
    return int(round((n_frames / fps * 16000 - 1024) / 512 + 1))


def plan_windows(clip_length, n_poses, n_pre_poses, fps):
    """(unit_time, stride_time, num_subdivision) -- synthesize_hierarchy.py:54-59"""
    unit_time = n_poses / fps
    stride_time = (n_poses - n_pre_poses) / fps
    if clip_length < unit_time:
        return unit_time, stride_time, 1
    return unit_time, stride_time, math.ceil((clip_length - unit_time) / stride_time) + 1


def words_in_time_range(word_list, start_time, end_time):
    """scripts/data_loader/data_preprocessor.py:174-187: words (text, start, end) overlapping [start_time, end_time)"""
    words = []
    for word in word_list:
        if word[1] >= end_time:
            break
        if word[2] <= start_time:
            continue
        words.append(word)
    return words


def frame_tokens(lang_model, words, start_time, end_time, n_frames):
    """Word ids at their onset frames, 0 (padding) elsewhere -- synthesize_hierarchy.py:103-117."""
    ext = np.zeros(n_frames)
    frame_duration = (end_time - start_time) / n_frames
    for word in words_in_time_range(words, start_time, end_time):
        idx = max(0, int(np.floor((word[1] - start_time) / frame_duration)))
        ext[idx] = lang_model.get_word_index(word[0])
    return torch.LongTensor(ext).unsqueeze(0)


def window_blend(win, out_all, index, n_pre):
    """Cross-fade window `index` ([T, P] on the device) into the running sequence out_all [frames, P] (in place)."""
    T, P = win.shape
    assert out_all.shape[1] == P and out_all.shape[0] >= index * (T - n_pre) + T
    check(lib.ha2g_window_blend_f32(win.contiguous().data_ptr(), out_all.data_ptr(), index, T, n_pre, P,
                                    torch.cuda.current_stream().cuda_stream))
    return out_all


def generate_gestures_hierarchy(args, gens, audio_encoder, lang_model, audio, words, targets=None, audio_sr=16000, vid=None,
                                fade_out=False, spectrogram=None, device=None):
    """-> numpy [frames, P] direction vectors (mean-subtracted), like the reference function.
    gens = (g1, g2, g3) (or the six expressive generators); audio = 1-D array of samples (only its length and, when `spectrogram`
    is None, the on-GPU log-mel of it are used); words = [(text, start_s, end_s)]; targets = optional per-level seed tensors
    [1, n_poses, P_k] (zeros when None, as the reference's callers pass)."""
    spec = GESTURE_SPEC if len(gens) == 3 else EXPRESSIVE_SPEC
    device = device or next(gens[0].parameters()).device
    n_frames, n_pre = args.n_poses, args.n_pre_poses
    fps = args.motion_resampling_framerate
    clip_length = len(audio) / audio_sr
    if spectrogram is None:
        from .audio_frontend import extract_melspectrogram
        spectrogram = extract_melspectrogram(audio, audio_sr)
    spectrogram = torch.as_tensor(spectrogram).to(device=device, dtype=torch.float32)
    unit_time, stride_time, num_subdivision = plan_windows(clip_length, n_frames, n_pre, fps)
    spec_len = calc_spectrogram_length_from_motion_length(n_frames, fps)
    audio_sample_length = int(unit_time * audio_sr)
    end_padding_duration = 0
    if args.z_type == 'speaker':
        if not vid:
            import random
            vid = random.randrange(gens[0].z_obj.n_words)
        vid_t = torch.LongTensor([vid]).to(device)
    else:
        vid_t = None
    dims = spec['pose_dims']
    cols = [torch.tensor(c, dtype=torch.long, device=device) for c in spec['level_cols']]
    tables = [ops.scatter_tables(P, dims[k - 1] if k else 0, spec['scatter'][k], device) for k, P in enumerate(dims)]
    if targets is None:
        targets = [torch.zeros(1, n_frames, P, device=device) for P in dims]
    targets = [t.to(device=device, dtype=torch.float32).clone() for t in targets]
    out_all = torch.zeros((num_subdivision - 1) * (n_frames - n_pre) + n_frames, dims[-1], device=device)
    out_dir_vec = None
    for m in list(gens) + [audio_encoder]:
        m.eval()
    with torch.no_grad():
        for i in range(num_subdivision):
            start_time = i * stride_time
            end_time = start_time + unit_time
            # the reference scales by spectrogram.shape[0] (the 128 mel bins, not the frame count): kept as is (:84-87)
            a0 = math.floor(start_time / clip_length * spectrogram.shape[0])
            in_spec = spectrogram[:, a0:a0 + spec_len].unsqueeze(0).contiguous()
            s0 = math.floor(start_time / clip_length * len(audio))
            if len(audio) - s0 < audio_sample_length and i == num_subdivision - 1:
                end_padding_duration = audio_sample_length - (len(audio) - s0)
            in_text_padded = frame_tokens(lang_model, words, start_time, end_time, n_frames).to(device)
            if i > 0:        # seed poses = the previous window's last n_pre frames, per level (:121-126)
                for k, c in enumerate(cols):
                    targets[k][:, 0:n_pre] = out_dir_vec[:, -n_pre:] if len(c) == dims[-1] else out_dir_vec[:, -n_pre:].index_select(2, c)
            _, _, _, _, blend = audio_encoder(in_spec, vid_t)
            prev = None
            for k, g in enumerate(gens):
                pre = ops.pre_seq(targets[k], prev, tables[k], n_pre)
                prev, *_ = g(pre, in_text_padded, blend[k], vid_t)
            out_dir_vec = prev
            window_blend(out_dir_vec[0], out_all, i, n_pre)
    out = out_all.cpu().numpy()
    if fade_out:             # host-side post-processing, as in the reference (:193-213)
        n_smooth = n_pre
        start_frame = len(out) - int(end_padding_duration / audio_sr * fps)
        end_frame = start_frame + n_smooth * 2
        if len(out) < end_frame:
            out = np.pad(out, [(0, end_frame - len(out)), (0, 0)], mode='constant')
        out[end_frame - n_smooth:] = np.zeros((dims[-1]))
        y = out[start_frame:end_frame]
        x = np.array(range(0, y.shape[0]))
        w = np.ones(len(y))
        w[0] = 5
        w[-1] = 5
        coeffs = np.polyfit(x, y, 2, w=w)
        out[start_frame:end_frame] = np.transpose(np.asarray([np.poly1d(coeffs[:, k])(x) for k in range(y.shape[1])]))
    return out
