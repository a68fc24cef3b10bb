"""tests/golden/cfg1_tail.npz (200 one-ulp-perturbed float32 runs of the reference, tests/golden/gen_tail_study.py): the facts the round-3
decision on the audio tower's forward arithmetic rests on (tests/test_gpu_tail.py judges the HIP variants against the same data)."""
import numpy as np


def test_the_frozen_forward_tensor_has_a_measured_heavy_tail(golden):
    """The fact the decision rests on: the reference's own deviation on layer4.0.downsample.0.weight spans three orders of magnitude over 200
    one-ulp-perturbed runs, and its maximum is far above the 25-run maximum the round-2 tolerance was built from."""
    truth, tail = golden('cfg1'), golden('cfg1_tail')
    key = 'step0/grad/audio.feat_extractor.layer4.0.downsample.0.weight/sample'
    d = np.sort(tail[key + '@dev'])
    assert d[-1] > 3 * float(truth[key + '@noise'])            # max of 200 vs max of 25
    assert np.median(d) < 1e-2 * d[-1]                         # heavy tail: the typical run is 100x closer than the worst
    assert d[-1] > 7.8e-4                                      # covers what the 3-piece forward split measured in round 2
