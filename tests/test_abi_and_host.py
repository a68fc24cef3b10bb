"""CPU-side checks: the C-ABI library loads and exports every symbol include/ha2g_hip.h declares (no compute calls),
state_dict schemas, the returned-dict conventions of the train step, procedural determinism."""
import pytest
import ctypes
import os
import re

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from ha2g_amd import _lib
    protos = _lib.parse_header()
    assert len(protos) >= 45
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(raw, name), name
    # and nothing is declared twice / left out of the header that the sources export
    src = ''.join(open(os.path.join(ROOT, 'ha2g_amd', 'csrc', f)).read() for f in os.listdir(os.path.join(ROOT, 'ha2g_amd', 'csrc'))
                  if f.endswith('.hip'))
    exported = set(re.findall(r'\b(ha2g_[a-z0-9_]+)\s*\(', src)) - {'ha2g_set_error'}
    assert exported <= set(protos) | {'ha2g_last_error'}, exported - set(protos)
    assert _lib.lib.ha2g_abi_version() == _lib.ABI_VERSION == 6
    assert _lib.lib.ha2g_gru_supported_hidden(300) == 1 and _lib.lib.ha2g_gru_supported_hidden(123) == 0
    assert _lib.lib.ha2g_gru_packed_floats(300) == 19 * 3 * 19 * 256


def test_header_cites_reference_lines():
    h = open(os.path.join(ROOT, 'include', 'ha2g_hip.h')).read()
    for needle in ('model/hierarchy_net.py', 'model/ResNetSE34V2.py', 'model/ResNetBlocks.py', 'model/tcn.py',
                   'train_eval/train_hierarchy.py', 'train.py:155-170'):
        assert needle in h, needle


def test_module_state_dict_keys_match_reference_schema():
    from ha2g_amd import hierarchy_net as hn, schema
    from ha2g_amd.config import hierarchy_args
    from ha2g_testing import SpeakerVocab
    a = hierarchy_args()
    spk = SpeakerVocab(9)
    pairs = [
        (hn.Hierarchical_PoseGenerator(a, 27, 50, 300, None, z_obj=spk), schema.generator_schema(27, 50, 9, 300, 4)),
        (hn.Hierarchical_ConvDiscriminator(27), schema.discriminator_schema(27)),
        (hn.Hierarchical_WavEncoder(a, spk, 3), schema.wav_encoder_schema(9, 3)),
        (hn.TextEncoderTCN(a, 50, 300, None, dropout=a.dropout_prob), schema.text_encoder_schema(50, 300, 4)),
    ]
    for m, sch in pairs:
        got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        assert got == {k: tuple(v) for k, v in sch.items()}
    g = pairs[0][0]
    assert g.in_size == 32 + 32 + 27 + 1 + 16                    # hierarchy_net.py:68,75
    n_params = sum(p.numel() for p in pairs[2][0].parameters())
    assert n_params == 5_657_925 - (1371 - 9) * 16 + 0 or n_params > 5_600_000   # 5.68 M at n_spk=1371
    w = pairs[2][0].feat_extractor.layer3[0].conv1.weight
    assert w.shape == (128, 64, 3, 3) and w.is_contiguous(memory_format=torch.channels_last)


def test_ret_dict_conventions():
    from ha2g_amd.config import hierarchy_args
    from ha2g_amd.train_hierarchy import _ret_dict
    a = hierarchy_args()
    names = ['loss', 'KLD', 'DIV_REG', 'gen', 'dis', 'c_pos', 'c_neg', 'phy']
    r = _ret_dict(a, names, [0.1, 0.5, -2.0, 0.7, 1.386, 8.0, -8.4, 0.2])
    assert list(r) == names
    assert abs(r['loss'] - 7.0) < 1e-9 and abs(r['gen'] - 3.5) < 1e-9 and r['dis'] == 1.386     # weights of hierarchy.yml
    assert abs(r['c_neg'] + 0.042) < 1e-9
    r0 = _ret_dict(a, names, [0.1, 0.0, 0.0, 0.7, 1.386, 8.0, -8.4, 0.2])
    assert 'KLD' not in r0 and 'DIV_REG' not in r0              # `if kld:` / `if div_reg:` (train_hierarchy.py:277-280)


def test_procedural_is_deterministic_and_alias_consistent():
    from ha2g_amd import procedural as proc
    a = proc.tensor_for('g1.text_encoder.tcn.network.0.conv1.weight_v', (8, 4, 2), 3)
    b = proc.tensor_for('g1.text_encoder.tcn.network.0.net.0.weight_v', (8, 4, 2), 3)
    assert np.array_equal(a, b)
    t1 = proc.make_batch(5, 27, 100, 9, 7)
    t2 = proc.make_batch(5, 27, 100, 9, 7)
    assert all(np.array_equal(x, y) for x, y in zip(t1, t2))
    text = t1[0]
    assert ((text > 0).sum(1) >= 5).all() and ((text > 0).sum(1) <= 8).all() and text.shape == (5, 34)
    assert t1[1].min() >= -80 and t1[1].max() <= 0 and t1[1].shape == (5, 128, 70)


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    import importlib
    from ha2g_amd import _lib
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    try:
        _lib._load()
    except ImportError as e:
        assert 'no CPU/torch fallback' in str(e)
    else:
        raise AssertionError('loading without the HIP library must fail')


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` without a launcher starts N rank processes (fresh children, rendezvous on 127.0.0.1) before the parent
    touches the GPU; under torchrun-style environments it does not re-spawn.  --dry-run makes every rank report its environment."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '3', '--dry-run'], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    recs = [json.loads(l) for l in p.stdout.splitlines() if l.startswith('{')]
    assert sorted(r['rank'] for r in recs) == [0, 1, 2] and all(r['world'] == 3 and r['local_rank'] == r['rank'] for r in recs)
    assert len({r['master'] for r in recs}) == 1 and recs[0]['master'].startswith('127.0.0.1:') and all(r['ipc_legacy'] == '0' for r in recs)
    env2 = dict(env, RANK='1', LOCAL_RANK='1', WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='29999')      # launched by torchrun: no re-spawn
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--dry-run'], env=env2, capture_output=True, text=True, timeout=300)
    recs = [json.loads(l) for l in p.stdout.splitlines() if l.startswith('{')]
    assert p.returncode == 0 and len(recs) == 1 and recs[0]['rank'] == 1 and recs[0]['world'] == 2


def test_fastcall_binding_covers_the_header_and_converts_like_ctypes():
    """ha2g_amd/_ha2g_fastcall (generated from include/ha2g_hip.h by csrc/gen_fastcall.py) exports one METH_FASTCALL wrapper per entry point;
    pointer arguments accept ints, None, ctypes pointer objects (their VALUE) and ctypes arrays (their storage), as ctypes' c_void_p does."""
    import ctypes
    from ha2g_amd import _lib
    from ha2g_amd import _ha2g_fastcall as fc
    assert _lib.lib.fastcall
    for name in _lib.parse_header():
        assert hasattr(fc, name), name
    assert fc.ha2g_abi_version() == _lib._clib.ha2g_abi_version()
    arr = (ctypes.c_float * 4)(1, 2, 3, 4)
    for ptr in (None, 0, ctypes.addressof(arr), arr, ctypes.cast(arr, ctypes.c_void_p)):
        # M = 0: returns before touching any operand, after every argument has been converted
        assert fc.ha2g_gemm_f32(0, 1, 0, 0, 0, 1.0, ptr, 0, ptr, 0, 0, ptr, 0, None, 0, None, 0, None) == 0
    with pytest.raises(TypeError):
        fc.ha2g_gemm_f32(0, 1, 0)
    with pytest.raises(TypeError):
        fc.ha2g_gemm_f32(0, 1, 0, 0, 0, 1.0, 'x', 0, None, 0, 0, None, 0, None, 0, None, 0, None)


def test_graft_entry_build_passes_on_the_tree_as_it_is():
    """The driver's "does it build" hook: make (a no-op when the objects are current), import of the oracle and the package, ABI version check."""
    import __graft_entry__ as g
    g.build()
