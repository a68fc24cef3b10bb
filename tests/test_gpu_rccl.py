"""RCCL smoke on the GPU box: a single-rank "nccl" process group (the backend the multi-GPU bench uses) must broadcast and
all-reduce the flat parameter / gradient buffers of a trainer and leave a train step working.  Runs in a child process so
the process group never leaks into the pytest process.  (The world_size-2 arithmetic is covered on CPU by test_ddp_gloo.)"""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
from bench import Vocab
from ha2g_amd import procedural as proc
from ha2g_amd.config import hierarchy_args
from ha2g_amd.train import HierarchyTrainer
args = hierarchy_args(hidden_size=32, n_layers=2)
tr = HierarchyTrainer(args, Vocab(50), Vocab(7), 27, dev)
tr.broadcast_parameters(0)
text, spec, target, vid = (torch.from_numpy(x).to(dev) for x in proc.make_batch(3, 27, 50, 7, 5))
r0 = tr.train_iter(11, text, spec, target, vid)
for o in tr.gen_opts + [tr.audio_opt, tr.text_opt, tr.dis_opt]:
    before = o.flat_g.clone()
    dist.all_reduce(o.flat_g)                      # world of one: must be the identity, through RCCL
    assert torch.equal(before, o.flat_g)
r1 = tr.train_iter(11, text, spec, target, vid)
torch.cuda.synchronize()
assert all(v == v for v in r1.values()), r1
dist.destroy_process_group()
print('RCCL_OK')
'''


def test_single_rank_rccl_group():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, '-c', CHILD % dict(root=ROOT)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and 'RCCL_OK' in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])
