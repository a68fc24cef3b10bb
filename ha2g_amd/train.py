"""Driver-side surface the train step sits behind: init_model() and the optimizer set-up of the reference's
scripts/train.py:50-88,114-170, plus a minimal batch loop (:254-316) for synthetic / in-memory data.
Dataset building, evaluation (FGD), checkpoint I/O and video synthesis are out of scope."""
import torch

from .hierarchy_net import (Hierarchical_ConvDiscriminator, Hierarchical_PoseGenerator, Hierarchical_WavEncoder,
                            TextEncoderTCN)
from . import ops
from .optim import FusedAdam
from .train_hierarchy import train_iter_hierarchy, train_iter_hierarchy_expressive


def init_model(args, lang_model, speaker_model, pose_dim, _device=None, pose_level=3):
    """reference scripts/train.py:50-88 (only `args.model == 'hierarchy'` is on the hot path); train_expressive.py:95-133
    is the same with pose_level=6 for the audio encoder."""
    generator = discriminator = audio_encoder = text_encoder = loss_fn = None
    if args.model == 'hierarchy':
        generator = Hierarchical_PoseGenerator(args, n_words=lang_model.n_words, word_embed_size=args.wordembed_dim,
                                               word_embeddings=lang_model.word_embedding_weights, z_obj=speaker_model,
                                               pose_dim=pose_dim)
        discriminator = Hierarchical_ConvDiscriminator(pose_dim, n_frames=int(getattr(args, 'n_poses', 34)))
        audio_encoder = Hierarchical_WavEncoder(args, z_obj=speaker_model, pose_level=pose_level, nOut=32)
        text_encoder = TextEncoderTCN(args, lang_model.n_words, args.wordembed_dim,
                                      pre_trained_embedding=lang_model.word_embedding_weights, dropout=args.dropout_prob)
    else:
        raise NotImplementedError('ha2g_amd implements the `hierarchy` model family only (got %r)' % args.model)
    return generator, discriminator, audio_encoder, text_encoder, loss_fn


class HierarchyTrainer:
    """Everything train_epochs() builds before its batch loop (reference scripts/train.py:114-170)."""

    def __init__(self, args, lang_model, speaker_model, pose_dim, device, pose_dims=(15, 21, 27), sparse_embeddings=None):
        """sparse_embeddings: update the word-embedding tables row-wise from compact gradients (bit-identical to the dense update,
        see ha2g_amd.optim.FusedAdam); call sync_sparse() before reading those tables outside the step (checkpoints).  None (default) = row-wise
        under data parallelism -- the four n_words x 300 tables are 45 % of the gradient bytes and almost all zeros, only the touched rows travel
        (ddp.exchange_sparse_) -- dense on a single GPU, where the compaction launches cost more than the Adam traffic they save."""
        from . import ddp
        self.args, self.device = args, device
        if sparse_embeddings is None:
            sparse_embeddings = ddp.active()
        self.sparse_embeddings = bool(sparse_embeddings)
        self.expressive = len(pose_dims) == 6        # scripts/train_expressive.py:160-168: six generators 24/30/36/66/96/126
        _, self.discriminator, self.audio_encoder, self.text_encoder, _ = init_model(args, lang_model, speaker_model, pose_dim, device,
                                                                                     pose_level=len(pose_dims))
        self.gens = [init_model(args, lang_model, speaker_model, pd, device)[0] for pd in pose_dims]
        for m in self.modules():
            m.to(device)
        self._flatten_bn_buffers()
        self.make_optimizers()

    def modules(self):
        return list(self.gens) + [self.discriminator, self.audio_encoder, self.text_encoder]

    def make_optimizers(self):
        a = self.args
        lr = float(a.learning_rate)
        sp = (lambda enc: [enc.embedding.weight]) if self.sparse_embeddings else (lambda enc: [])
        self.gen_opts = [FusedAdam(g.parameters(), lr=lr, betas=(0.5, 0.999), sparse=sp(g.text_encoder)) for g in self.gens]
        self.audio_opt = FusedAdam(self.audio_encoder.parameters(), lr=lr, betas=(0.5, 0.999))
        self.text_opt = FusedAdam(self.text_encoder.parameters(), lr=lr, betas=(0.5, 0.999), sparse=sp(self.text_encoder))
        self.dis_opt = FusedAdam(self.discriminator.parameters(), lr=lr * a.discriminator_lr_weight, betas=(0.5, 0.999))

    def sync(self):
        """Waits for the device and raises if a cluster-GRU hand-off of an earlier step timed out.  train_iter returns as soon as the
        step's losses have reached the host (they exist before the backward); the backward / optimizer kernels may still be running."""
        from . import ops
        from .train_hierarchy import drain_cluster_errors
        torch.cuda.synchronize(self.device)
        try:
            drain_cluster_errors(block=True)
        except ops.Ha2gClusterError:
            # with the retry on, a flag found HERE (the device is drained: the word is final and, under data parallelism, equal on every rank
            # -- sync() is called by all ranks at the same point) is handled like one found at the next step: the flagged updates were skipped on
            # the device, switch to gru.hip and go on.  The BatchNorm statistics of the flagged forward stay (no snapshot is guaranteed to predate it)
            if not self.retry_on_cluster_error:
                raise
            self._recover_from_cluster_error(restore_buffers=False)

    def sync_sparse(self):
        for o in self.gen_opts + [self.text_opt]:
            o.sync_sparse()

    def state_dict(self):
        """The reference's checkpoint payload (scripts/train.py:202-243: gen_dict_1..L, dis_dict, audio_dict, text_dict) with every table current:
        row-wise embedding tables are updated lazily, so the rows a step did not touch are brought up to date first (sync_sparse) -- reading
        module.state_dict() directly under `sparse_embeddings` returns stale rows."""
        self.sync()
        self.sync_sparse()
        out = {'gen_dict_%d' % (i + 1): g.state_dict() for i, g in enumerate(self.gens)}
        out.update(dis_dict=self.discriminator.state_dict(), audio_dict=self.audio_encoder.state_dict(), text_dict=self.text_encoder.state_dict())
        return out

    def broadcast_parameters(self, src=0):
        """DDP start-up: every rank adopts rank `src`'s parameters and buffers."""
        from . import ddp
        for o in self.gen_opts + [self.audio_opt, self.text_opt, self.dis_opt]:
            ddp.broadcast_one_(o.flat_p, src)
        for m in self.modules():
            for p in m.parameters():
                if not p.requires_grad or getattr(p, '_ha2g_sparse', None) is not None:      # frozen / row-wise tables live outside the flat buffers
                    ddp.broadcast_one_(p.data, src)
            for b in m.buffers():
                ddp.broadcast_one_(b, src)

    def sync_bn_stats(self, mode='mean'):
        """Call before a checkpoint / validation pass under data parallelism: the BatchNorm running statistics of the audio encoder and the
        discriminator are rank-local during training (ha2g_amd.ddp.sync_bn_stats_); afterwards every rank would save the same state_dict."""
        from . import ddp
        return ddp.sync_bn_stats_(self.modules(), mode)

    def capture_step(self, epoch, in_text_padded, in_spec, target, vid_indices, warmup=2):
        """Capture one whole train step on these (static) input tensors into a hipGraph.  Returns (graph, names, packed):
        `graph.replay()` runs the step (fresh dropout masks / noise per replay, device-side Adam counters), `packed` then holds
        the logged scalars in `names` order.  Update the input tensors in place to feed new batches."""
        s = torch.cuda.Stream(self.device)
        s.wait_stream(ops.cur_stream(self.device))
        with torch.cuda.stream(s):                   # allocations / workspaces of the capture stream are created outside capture
            for _ in range(warmup):
                self.train_iter(epoch, in_text_padded, in_spec, target, vid_indices, return_tensors=True)
        ops.cur_stream(self.device).wait_stream(s)
        torch.cuda.synchronize(self.device)
        from .train_hierarchy import drain_cluster_errors
        drain_cluster_errors(block=True)             # error words of earlier eager steps are looked at (and raised) BEFORE the capture starts
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):      # same stream as the warm-up: its workspace exists already, outside the capture
            names, packed = self.train_iter(epoch, in_text_padded, in_spec, target, vid_indices, return_tensors=True)
        return graph, names, packed

    # ---- a flagged cluster-GRU step is harmless and recoverable ----
    # The error word of gru_cluster.hip guards every optimizer kernel ON THE DEVICE (FusedAdam.step -> ha2g_adam_guarded_f32): a step whose
    # recurrences timed out leaves parameters, moments and step counters bit-identical.  The host learns about it from the packed loss
    # read-back (forward launches: same call) or from the end-of-step copy of the word (BPTT launches: start of the next call).  Recovery:
    # drain the device, restore the BatchNorm buffers the flagged forward touched, clear the word, switch this process to the single-workgroup
    # recurrences of gru.hip (fresh launches, same process) and run the batch at hand.  cluster_retries counts the recoveries.
    retry_on_cluster_error = True
    cluster_retries = 0

    def _flatten_bn_buffers(self):
        """The BatchNorm running statistics of all modules become views of ONE flat fp32 tensor (and the num_batches_tracked counters of one
        int64 tensor), like the parameters are views of the optimizers' flat buffers: the pre-step snapshot that the retry needs is then two
        copies instead of 123 (41 BatchNorm layers x 3 buffers: 0.5 ms of tiny launches on the main queue per step).  Same values, same
        state_dict; the kernels update the views in place."""
        fl, il = [], []
        for m in self.modules():
            for sub in m.modules():
                for k in ('running_mean', 'running_var'):
                    b = sub._buffers.get(k)
                    if b is not None and b.dtype == torch.float32:
                        fl.append((sub, k, b))
                b = sub._buffers.get('num_batches_tracked')
                if b is not None and b.dtype == torch.int64 and b.dim() == 0:
                    il.append((sub, b))
        self._bn_flat = self._bn_cnt = None
        if fl:
            flat = torch.cat([b.reshape(-1) for _, _, b in fl])
            o = 0
            for sub, k, b in fl:
                sub._buffers[k] = flat[o:o + b.numel()].view(b.shape)
                o += b.numel()
            self._bn_flat = flat
        if il:
            cnt = torch.stack([b for _, b in il])
            for i, (sub, _) in enumerate(il):
                sub._buffers['num_batches_tracked'] = cnt[i]
            self._bn_cnt = cnt
        self._bn_snap = None
        self._alias_probe = {}

    def _bn_buffers(self):
        return [b for m in self.modules() for k, b in m.named_buffers() if k.endswith(('running_mean', 'running_var', 'num_batches_tracked'))]

    def _snapshot_buffers(self, slot=0):
        """slot: the data-parallel path keeps the snapshots of the last three steps (a flagged step is noticed two calls later)"""
        if self._bn_snap is None:
            self._bn_snap = {}
        if slot not in self._bn_snap:
            self._bn_snap[slot] = tuple(torch.empty_like(t) if t is not None else None for t in (self._bn_flat, self._bn_cnt))
        for src, name in ((self._bn_flat, 'running statistics'), (self._bn_cnt, 'num_batches_tracked')):
            if src is not None:
                # the module buffers must still be the views of the flat tensor installed by _flatten_bn_buffers: a later module.to(dtype) or an
                # assign-style load_state_dict re-binds them, and the snapshot / restore would then silently do nothing.  The probe buffer is looked
                # up once per flat tensor (walking every module twice per step was host time); the check itself stays per step.
                probe = self._alias_probe.get(name)
                if probe is None or probe[2] is not src:          # (owning module, attribute): the module's CURRENT buffer is fetched every step
                    probe = self._alias_probe[name] = next((m, k, src) for mod in self.modules() for m in mod.modules()
                                                           for k, b in m.named_buffers(recurse=False)
                                                           if k in ('running_mean', 'running_var', 'num_batches_tracked') and b.dtype == src.dtype)
                first = getattr(probe[0], probe[1])
                assert first.data_ptr() == src.data_ptr(), 'HierarchyTrainer: the BatchNorm %s no longer alias the flat snapshot buffer ' \
                    '(module.to(dtype) / load_state_dict(assign=True) after construction?): call _flatten_bn_buffers() again' % name
        for dst, src in zip(self._bn_snap[slot], (self._bn_flat, self._bn_cnt)):
            if src is not None:
                dst.copy_(src)

    def _restore_buffers(self, slot=0):
        for dst, src in zip((self._bn_flat, self._bn_cnt), self._bn_snap[slot]):
            if dst is not None:
                dst.copy_(src)

    def _recover_from_cluster_error(self, restore_buffers, slot=0):
        from . import ops
        from .train_hierarchy import _err_free, _err_watch
        torch.cuda.synchronize(self.device)
        _err_free.extend(w for _, w, _ in _err_watch)   # copies of the word taken while it was set: dropped, their pinned words go back to the pool
        _err_watch.clear()
        if restore_buffers and self._bn_snap and slot in self._bn_snap:
            self._restore_buffers(slot)
        err = ops.gru_cluster_error_tensor(self.device)
        if err is not None:
            err.zero_()
        ops.USE_GRU_CLUSTER = False
        self.cluster_retries += 1
        if self.cluster_retries > self.max_cluster_retries:
            raise ops.Ha2gClusterError('ha2g_amd: %d cluster-GRU recoveries in this process (max_cluster_retries = %d): the fallback recurrences '
                                       'do not use the cluster protocol, so something else sets the error word' % (self.cluster_retries, self.max_cluster_retries))

    max_cluster_retries = 3      # a process switches to gru.hip at its FIRST recovery: more than a couple means the word is being set by something else

    def _step(self, epoch, in_text_padded, in_spec, target, vid_indices, **kw):
        fn = train_iter_hierarchy_expressive if self.expressive else train_iter_hierarchy
        return fn(self.args, epoch, in_text_padded, in_spec, target, vid_indices, *self.gens, self.discriminator, self.audio_encoder,
                  self.text_encoder, *self.gen_opts, self.dis_opt, self.audio_opt, self.text_opt, **kw)

    def train_iter(self, epoch, in_text_padded, in_spec, target, vid_indices, **kw):
        from . import ddp, ops
        capturing = self.device.type == 'cuda' and torch.cuda.is_current_stream_capturing()
        if not self.retry_on_cluster_error or capturing or kw.get('return_tensors'):
            return self._step(epoch, in_text_padded, in_spec, target, vid_indices, **kw)
        if ddp.active():
            return self._train_iter_ddp(epoch, in_text_padded, in_spec, target, vid_indices, **kw)
        from .train_hierarchy import drain_cluster_errors
        try:
            drain_cluster_errors()                      # the PREVIOUS step's BPTT launches flagged: its update was skipped on the device
        except ops.Ha2gClusterError:
            self._recover_from_cluster_error(restore_buffers=False)      # that step's forward was valid: its BatchNorm statistics stay
            return self._step(epoch, in_text_padded, in_spec, target, vid_indices, **kw)
        self._snapshot_buffers()
        try:
            return self._step(epoch, in_text_padded, in_spec, target, vid_indices, **kw)
        except ops.Ha2gClusterError:
            self._recover_from_cluster_error(restore_buffers=True)
            return self._step(epoch, in_text_padded, in_spec, target, vid_indices, **kw)

    _ddp_call = 0

    def _train_iter_ddp(self, epoch, in_text_padded, in_spec, target, vid_indices, **kw):
        """Data parallel: the recovery is COLLECTIVE.  The step issues all-reduces / all-gathers, so a rank may neither raise nor
        re-run a step on its own.  Every rank therefore decides from the same information at the same call: the end-of-step error words --
        MAX-reduced over the ranks inside each step before the optimizer kernels read them -- examined with a fixed lag of one call
        (train_hierarchy.drain_cluster_errors_lagged).  A step flagged on ANY rank was a no-op for the optimizer state on EVERY rank, and so
        was the step after it (the word is sticky until cleared here); at the call that notices it all ranks restore the BatchNorm statistics
        to what they were before the first flagged step, clear the word, switch to the gru.hip recurrences and run the batch at hand.  The
        two flagged batches are dropped (their loss dicts may hold garbage on the rank that timed out).  The rank-local forward-time word is not
        part of the loss read-back in this mode (train_hierarchy._train_iter), so nothing raises on one rank only."""
        from .train_hierarchy import drain_cluster_errors_lagged
        k = self._ddp_call
        self._ddp_call = k + 1
        if drain_cluster_errors_lagged(raise_=False):
            # the copies looked at belong to the calls <= k - 2; the sticky word makes every call from the first flagged one onwards a no-op, and the
            # oldest snapshot still held (taken before call k - 2) predates the first of them that this drain can have seen for the first time
            self._recover_from_cluster_error(restore_buffers=True, slot=(k - 2) % 3 if k >= 2 else 0)
        self._snapshot_buffers(slot=k % 3)
        return self._step(epoch, in_text_padded, in_spec, target, vid_indices, **kw)
