"""train_iter_hierarchy -- MI355X-native drop-in for scripts/train_eval/train_hierarchy.py:71-293.

Same signature, same side effects (parameters, Adam state, BatchNorm running statistics, .grad) and the same
returned dict of python floats (keys loss / KLD / DIV_REG / gen / dis / c_pos / c_neg / phy, with the reference's
truthiness quirk that drops KLD / DIV_REG when exactly 0).  Differences are purely in scheduling:

  * the three generator chains of a step (D-phase chain, main chain, random-speaker chain; reference lines
    100-117, 153-170, 194-211) share weights and inputs, so they run as ONE pass over a 3B (2B during warm-up)
    batch -- the persistent GRU kernels are latency-bound, so the extra rows are almost free -- and only the
    main third receives a non-zero gradient;
  * the <= 8 logged scalars come back in a single device->host copy instead of ~9 .item() syncs.
"""
import math
import os
import time

import torch

from . import ops
from .config import EXPRESSIVE_SPEC, GESTURE_SPEC

FUSE_CHAINS = True            # False = literal three-pass ordering of the reference (used as a cross-check in tests)
randperm_source = None        # tests inject a fixed permutation: callable(n, device) -> LongTensor

_const_cache = {}


def _consts(spec, args, device):
    key = (spec['name'], id(args), str(device))
    c = _const_cache.get(key)
    if c is None:
        c = dict(mean_dir=torch.tensor(args.mean_dir_vec, dtype=torch.float32).squeeze(1).to(device),
                 pairs=torch.tensor(spec['phys_pairs'], dtype=torch.int32, device=device),
                 avg=torch.tensor(spec['phys_avg'], dtype=torch.float32, device=device),
                 var=torch.tensor(spec['phys_var'], dtype=torch.float32, device=device),
                 cols=[torch.tensor(c, dtype=torch.long, device=device) for c in spec['level_cols']],
                 # per-level scatter tables of the pre_seq pack kernel (csrc/pack.hip)
                 tables=[ops.scatter_tables(P, spec['pose_dims'][k - 1] if k else 0, spec['scatter'][k], device)
                         for k, P in enumerate(spec['pose_dims'])])
        _const_cache[key] = c
    return c


def _chain(spec, args, gens, targets, in_text, blend, vids, tables):
    """coarse-to-fine decode g1 -> ... -> gL (reference train_hierarchy.py:153-170 / train_hierarchy_expressive.py:272-316).
    Level k's pre_seq = (target frames, constraint bit) for the first n_pre frames; its later frames are seeded with level
    k-1's output through spec['scatter'][k] -- one pack kernel per level (differentiable w.r.t. the coarser output) instead
    of the reference's zeros + slice writes.  All inputs may carry a multiple of B rows."""
    n = args.n_pre_poses
    outs, last = [], None
    text_feats = _grouped_text_features(gens, in_text)
    for k, g in enumerate(gens):
        pre = ops.pre_seq(targets[k], outs[-1] if k else None, tables[k], n)
        o, z, mu, logvar = g(pre, in_text, blend[k], vids, **({} if text_feats is None else {'text_feat_seq': text_feats[k]}))
        outs.append(o)
        last = (z, mu, logvar)
    return outs, last[0], last[1], last[2]


FUSE_TEXT = True     # the generators' text encoders as grouped launches (hierarchy_net.grouped_text_encoders)


def _grouped_text_features(gens, in_text):
    """Per-generator text features from ONE lockstep pass over all generators' text encoders, or None when they cannot be grouped.
    Honours the fused-chain row split (hierarchy_net._row_split): rows that receive no gradient are evaluated under no_grad."""
    from .hierarchy_net import grouped_conv_weights, grouped_text_encoders, text_encoders_groupable
    mods = [g.module if hasattr(g, 'module') else g for g in gens]
    if not FUSE_TEXT or not in_text.is_cuda or any(m.input_context == 'none' for m in mods):
        return None
    encs = [m.text_encoder for m in mods]
    if not text_encoders_groupable(encs):
        return None
    gs = mods[0].gru.grad_slice
    if gs is None or not torch.is_grad_enabled():
        return grouped_text_encoders(encs, in_text)
    s0, cnt = gs
    B = in_text.shape[0]
    parts = []
    wn = grouped_conv_weights(encs)                      # once per step, shared by the row blocks
    for a, b, grad in ((0, s0, False), (s0, s0 + cnt, True), (s0 + cnt, B, False)):
        if b <= a:
            continue
        if grad:
            parts.append(grouped_text_encoders(encs, in_text[a:b], wn))
        else:
            with torch.no_grad():
                parts.append(grouped_text_encoders(encs, in_text[a:b], wn))
    return torch.cat(parts, dim=1) if len(parts) > 1 else parts[0]


_pinned_bufs = {}
host_clock = None        # bench.py: {'busy': s, 'steps': n} -- host time spent inside train_iter excluding the final wait for the loss read-back
phase_clock = None       # tools/phase_spans.py: a list -> (label, GPU event on the compute stream, host perf_counter) at the step's phase boundaries
comm_clock = None        # bench.py (N > 1): list of (event behind the last backward kernel, event in front of the first optimizer kernel) per step --
                         # the time between them is the EXPOSED gradient exchange: the audio tower's bucket (last by design) + whatever is left of the others


def _phase(label):
    if phase_clock is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        phase_clock.append((label, ev, time.perf_counter()))


_err_watch = []          # (event, pinned int32 word, device): end-of-step copies of the cluster error word not yet looked at
_err_free = []


def _pinned(dev, slot, n):
    key = (dev.index, slot)
    b = _pinned_bufs.get(key)
    if b is None or b.numel() < n:
        b = _pinned_bufs[key] = torch.empty(max(n, 32), dtype=torch.float32, pin_memory=True)
    return b[:n]


def _watch_cluster_errors(dev):
    """Asynchronous copy of the cluster-GRU error word at the end of a step (it then covers the step's BPTT launches)."""
    err = ops.gru_cluster_error_tensor(dev)
    if err is None:
        return
    word = _err_free.pop() if _err_free else torch.empty(1, dtype=torch.int32, pin_memory=True)
    word.copy_(err[:1], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    _err_watch.append((ev, word, dev))


def drain_cluster_errors(block=False, keep_last=0, raise_=True):
    """Looks at the end-of-step error words whose copies have completed (block=True: waits for all of them) and raises
    Ha2gClusterError when one is set (raise_=False: returns True instead).  Called at the start of every train step and by
    HierarchyTrainer.sync().  keep_last=n leaves the n most recent copies alone: see drain_cluster_errors_lagged."""
    bad = False
    while len(_err_watch) > keep_last and (block or _err_watch[0][0].query()):
        ev, word, dev = _err_watch.pop(0)
        ev.synchronize()
        bad = bad or int(word[0]) != 0
        _err_free.append(word)
    if bad and raise_:
        raise ops.Ha2gClusterError(_CLUSTER_MSG)
    return bad


def drain_cluster_errors_lagged(raise_=True):
    """The data-parallel form.  Ranks issue collectives inside the step, so they must notice a flagged step AT THE SAME CALL -- a non-blocking
    `event.query()` (drain_cluster_errors) may fire on one rank a call earlier than on its peer, and a rank that retries while the others move
    on leaves their all-reduces unmatched.  Rule: at the start of step k + 1 look at the end-of-step words of the steps <= k - 1, blocking.  Those
    copies ARE complete by then -- train_iter(k) returned only after step k's loss read-back (recorded behind the whole of step k - 1 on the
    compute stream) had fired -- so the wait costs nothing and the host's run-ahead into step k's backward is untouched; step k's own word
    is left for the next call.  The words themselves are rank-symmetric: every step MAX-reduces the word over the ranks (_sync_guard) before its
    end-of-step copy.  Same words, same call => every rank takes the same decision."""
    return drain_cluster_errors(block=True, keep_last=1, raise_=raise_)


_CLUSTER_MSG = ('ha2g_amd: a GRU cluster hand-off timed out (gru_cluster.hip): that step\'s GRU outputs and gradients are invalid -- '
                'the device cannot co-schedule the cluster\'s workgroups')


def _sync_guard(dev):
    """The optimizer kernels are predicated on the device's cluster-GRU error word (FusedAdam.step); under data parallelism the word is
    MAX-reduced over the ranks first, so that the replicas skip (or apply) a step together."""
    from . import ddp
    if ddp.active():
        ddp.sync_flag_(ops.gru_cluster_error_tensor(dev))


def _allreduce(optimizers):
    from . import ddp
    ddp.average_module_grads_(optimizers)


def train_iter_hierarchy(args, epoch, in_text_padded, in_spec, target, vid_indices,
                         g1, g2, g3, discriminator, audio_encoder, text_encoder,
                         gen_optimizer_1, gen_optimizer_2, gen_optimizer_3, dis_optimizer,
                         audio_optimizer, text_optimizer, return_tensors=False):
    """TED-Gesture (3 levels, 27-d pose): drop-in for scripts/train_eval/train_hierarchy.py:71-74."""
    return _train_iter(GESTURE_SPEC, args, epoch, in_text_padded, in_spec, target, vid_indices, (g1, g2, g3), discriminator,
                       audio_encoder, text_encoder, (gen_optimizer_1, gen_optimizer_2, gen_optimizer_3), dis_optimizer,
                       audio_optimizer, text_optimizer, return_tensors)


def train_iter_hierarchy_expressive(args, epoch, in_text_padded, in_spec, target, vid_indices,
                                    g1, g2, g3, g4, g5, g6, discriminator, audio_encoder, text_encoder,
                                    gen_optimizer_1, gen_optimizer_2, gen_optimizer_3,
                                    gen_optimizer_4, gen_optimizer_5, gen_optimizer_6, dis_optimizer,
                                    audio_optimizer, text_optimizer, return_tensors=False):
    """TED-Expressive (6 levels, 126-d pose): drop-in for scripts/train_eval/train_hierarchy_expressive.py:124-128."""
    return _train_iter(EXPRESSIVE_SPEC, args, epoch, in_text_padded, in_spec, target, vid_indices, (g1, g2, g3, g4, g5, g6),
                       discriminator, audio_encoder, text_encoder,
                       (gen_optimizer_1, gen_optimizer_2, gen_optimizer_3, gen_optimizer_4, gen_optimizer_5, gen_optimizer_6),
                       dis_optimizer, audio_optimizer, text_optimizer, return_tensors)


def _train_iter(spec, args, epoch, in_text_padded, in_spec, target, vid_indices, gens, discriminator, audio_encoder, text_encoder,
                gen_optimizers, dis_optimizer, audio_optimizer, text_optimizer, return_tensors=False):
    # Inside the step the backward functions may leave their in-place weight gradients running on the side stream (ops.SideStream.defer): every
    # backward() below is followed by a flush() before the gradient exchange / the optimizers touch the buffers.
    prev, ops.side.allow_defer = ops.side.allow_defer, True
    try:
        return _train_iter_impl(spec, args, epoch, in_text_padded, in_spec, target, vid_indices, gens, discriminator, audio_encoder, text_encoder,
                                gen_optimizers, dis_optimizer, audio_optimizer, text_optimizer, return_tensors)
    finally:
        ops.side.allow_defer = prev
        ops.gru_prep_clear()                                 # (also on an exception: a later step must not find this step's weight images)
        if target.is_cuda:
            ops.side.flush(target.device)


def _train_iter_impl(spec, args, epoch, in_text_padded, in_spec, target, vid_indices, gens, discriminator, audio_encoder, text_encoder,
                     gen_optimizers, dis_optimizer, audio_optimizer, text_optimizer, return_tensors=False):
    warm_up_epochs = args.loss_warmup
    dev = target.device
    B = target.shape[0]
    t_host0 = time.perf_counter() if host_clock is not None else 0.0
    if not (dev.type == 'cuda' and torch.cuda.is_current_stream_capturing()):
        # time-outs of earlier steps whose read-back has arrived (never inside a capture); data parallel: the rank-symmetric lagged form
        from . import ddp as _ddp
        (drain_cluster_errors_lagged if _ddp.active() else drain_cluster_errors)()
    ops.rng.begin_step()
    L = len(gens)
    consts = _consts(spec, args, dev)

    prefetched = False
    if (ops.GRU_PREFETCH and FUSE_CHAINS and args.z_type == 'speaker' and dev.type == 'cuda' and ops.side.enabled and torch.is_grad_enabled()
            and all(hasattr(g, 'gru') and hasattr(g.gru, '_names') for g in gens)):
        # the generators' GRU weight images (W_hh packs, W_ih piece planes and their transposes for the backward) do not depend on the step's data: they are
        # prepared on the side stream NOW, beside the audio tower's forward, instead of on the main queue between the recurrences (9 + 6 launches, ~0.15 ms)
        gan_ = epoch > warm_up_epochs and args.loss_gan_weight > 0.0
        div_ = (args.z_type == 'speaker' or args.z_type == 'random') and args.loss_reg_weight > 0.0
        rows_ = ((1 if gan_ else 0) + 1 + (1 if div_ else 0)) * B * target.shape[1]
        main_ = ops.cur_stream(dev)
        with ops.side.section(dev):
            for g in gens:
                ops.prefetch_gru([getattr(g.gru, n) for n in g.gru._names], g.gru.hidden_size, target.shape[1], rows_, main_)
        prefetched = True
    _phase('start')
    weight, feat_low, feat_mid, feat_high, linear_blend_feat = audio_encoder(in_spec, vid_indices)
    _phase('audio tower forward')
    text_feat = text_encoder(in_text_padded)
    _phase('text encoder forward')
    # Cut the autograd graph at the encoders' outputs: the backward runs in two stages (generators + losses, then the
    # encoders), so that under data parallelism the generators' gradient all-reduce is in flight while the audio tower --
    # the longest part of the backward -- is still back-propagating (BASELINE north_star: "all-reduce overlapped with
    # backward").  Same arithmetic: the encoders receive exactly the gradients accumulated on the cut tensors.
    enc_outs = [weight, feat_low, feat_mid, feat_high, text_feat] + list(linear_blend_feat)
    enc_cut = [t.detach().requires_grad_(True) if (torch.is_tensor(t) and t.requires_grad) else t for t in enc_outs]
    weight, feat_low, feat_mid, feat_high, text_feat = enc_cut[:5]
    linear_blend_feat = enc_cut[5:]

    # per-level targets = column subsets of the full pose (train_hierarchy.py:86-88 / expressive :140-145)
    targets = [target if len(c) == target.shape[2] else target.index_select(2, c) for c in consts['cols']]

    gan = epoch > warm_up_epochs and args.loss_gan_weight > 0.0
    use_div = (args.z_type == 'speaker' or args.z_type == 'random') and args.loss_reg_weight > 0.0
    rand_vids = None
    if use_div and args.z_type == 'speaker':
        rp = randperm_source(B, dev) if randperm_source is not None else torch.randperm(B, device=dev)
        rand_vids = vid_indices[rp]

    if prefetched:
        ops.side.join(dev)                                   # the prepared weight images (a few microseconds of side-queue work, long done)
    fused = None
    if FUSE_CHAINS and args.z_type == 'speaker':
        # row blocks: [D-phase chain (GAN phase only) | main chain | random-speaker chain]; eps is drawn block by block
        # in that order, which is the order the reference's three passes consume it.
        blocks = (['dis'] if gan else []) + ['main'] + (['rand'] if use_div else [])
        # physical row order: the gradient-carrying main block LAST, so that the rows evaluated under no_grad (row-wise
        # sub-networks, see Hierarchical_PoseGenerator._row_split) form ONE contiguous range = one fuller launch per op
        phys = [b for b in blocks if b != 'main'] + ['main']
        k = len(phys)
        rep = lambda t: t.repeat(*([k] + [1] * (t.dim() - 1))) if k > 1 else t
        vids_all = torch.cat([rand_vids if b == 'rand' else vid_indices for b in phys]) if k > 1 else vid_indices
        blend_all = []
        for lvl in range(L):
            f = linear_blend_feat[lvl]
            blend_all.append(torch.cat([f if b == 'main' else f.detach() for b in phys]) if k > 1 else f)
        main_at = phys.index('main')
        for g in gens:                                   # only the main block's rows carry gradient through the GRUs
            g.gru.grad_slice = (main_at * B, B) if k > 1 else None
            g.eps_block_order = [blocks.index(b) for b in phys]       # logical (reference pass) index of each physical block
        try:
            outs_all, z_all, mu_all, lv_all = _chain(spec, args, gens, [rep(t) for t in targets], rep(in_text_padded), blend_all, vids_all, consts['tables'])
        finally:
            for g in gens:
                g.gru.grad_slice = None
                g.eps_block_order = None
        fused = {}
        for i, b in enumerate(phys):
            sl = slice(i * B, (i + 1) * B)
            fused[b] = ([o[sl] for o in outs_all], z_all[sl], mu_all[sl], lv_all[sl])

    _phase('generator chains forward (fused rows)')
    ###########################################################################################
    # train D   (reference :93-131)
    dis_error = None
    if gan:
        dis_optimizer.zero_grad()
        if fused is not None:
            out_dir_vec_d = fused['dis'][0][-1]
        else:
            outs_d, *_ = _chain(spec, args, gens, targets, in_text_padded, linear_blend_feat, vid_indices, consts['tables'])
            out_dir_vec_d = outs_d[-1]
        if FUSE_CHAINS and hasattr(discriminator, 'forward_pair'):
            dis_real, dis_fake = discriminator.forward_pair(target, out_dir_vec_d.detach(), in_text_padded)
        else:
            dis_real = discriminator(target, in_text_padded)
            dis_fake = discriminator(out_dir_vec_d.detach(), in_text_padded)
        dis_error = ops.dis_loss(dis_real, dis_fake)                      # ns-gan
        dis_error.backward()
        ops.side.flush(dev)                              # deferred side-stream weight gradients (ops.SideStream.defer) before D's optimizer
        _allreduce([dis_optimizer])
        _sync_guard(dev)                                 # data parallel: a time-out on ANY rank makes this a no-op step on EVERY rank
        dis_optimizer.step()

    _phase('D phase (forward, backward, Adam)')
    ###########################################################################################
    # train G   (reference :135-274)
    for o in tuple(gen_optimizers) + (audio_optimizer, text_optimizer):
        o.zero_grad()

    N = text_feat.shape[0] * text_feat.shape[1]
    if args.loss_contrastive_pos_weight > 0.0:
        text_high_contrastive = ops.contrastive(text_feat.reshape(N, -1), feat_high.reshape(N, -1), spec['contrastive_expressive'])
    if args.loss_contrastive_neg_weight > 0.0:                # reference: text_low_contrastive = -contrastive(text, low); the sign rides in the weight
        text_low_pos = ops.contrastive(text_feat.reshape(N, -1), feat_low.reshape(N, -1), spec['contrastive_expressive'])

    if fused is not None:
        outs_main, z_context, z_mu, z_logvar = fused['main']
    else:
        outs_main, z_context, z_mu, z_logvar = _chain(spec, args, gens, targets, in_text_padded, linear_blend_feat, vid_indices, consts['tables'])
    out_dir_vec = outs_main[-1]

    beta = 0.1
    hubers = [ops.huber(o_k, t_k, beta) for o_k, t_k in zip(outs_main, targets)]
    dis_output = discriminator(out_dir_vec, in_text_padded)              # always executed, as in the reference (:179)
    gen_error = ops.gen_loss(dis_output)
    kld = div_reg = None

    # total loss = one weighted sum over the scalar terms (reference :226-262; same terms, same weights)
    terms, weights = list(hubers), [args.loss_regression_weight] * len(hubers)
    if use_div:
        if fused is not None:
            outs_rand, z_context_rand, _, _ = fused['rand']
        else:
            with torch.no_grad():
                outs_rand, z_context_rand, _, _ = _chain(spec, args, gens, targets, in_text_padded,
                                                         [f.detach() for f in linear_blend_feat], rand_vids, consts['tables'])
        out_dir_vec_rand_vid = outs_rand[-1]
        div_reg = ops.div_reg(out_dir_vec, out_dir_vec_rand_vid.detach(), z_context.detach(), z_context_rand.detach(), 0.05)
        if args.z_type == 'speaker':
            kld = ops.kld(z_mu, z_logvar)
            terms.append(kld); weights.append(args.loss_kld_weight)
        terms.append(div_reg); weights.append(args.loss_reg_weight)
    if epoch > warm_up_epochs:
        terms.append(gen_error); weights.append(args.loss_gan_weight)
    if args.loss_contrastive_pos_weight > 0.0:
        terms.append(text_high_contrastive); weights.append(args.loss_contrastive_pos_weight)
    if args.loss_contrastive_neg_weight > 0.0:
        terms.append(text_low_pos); weights.append(-args.loss_contrastive_neg_weight)
    if args.loss_physical_weight > 0.0:
        physical_loss = ops.phys_angle(out_dir_vec, consts['mean_dir'], consts['pairs'], consts['avg'], consts['var'], spec['palm'])
        terms.append(physical_loss); weights.append(args.loss_physical_weight)
    loss = ops.weighted_sum(terms, weights)

    # ---- one packed device->host transfer of the logged scalars, issued as soon as they exist ----
    # 'loss' = the sum of the per-level Huber terms (added on the host in fp32, left to right like the reference's
    # `huber_loss = h1 + h2 + ...`); 'c_neg' is logged with the reference's sign (it holds -contrastive(text, low)).
    # Every logged scalar is a forward quantity, so the copy is enqueued BEFORE the backward: train_iter returns when the copy's event
    # has fired instead of draining the device (the reference's `.item()` calls wait for backward + optimizer too: 1.2 ms of idle
    # matrix cores per step, the host re-starts every step with an empty launch queue).  The values are the same.
    names, vals = [], []
    for i, h in enumerate(hubers):
        names.append('loss' if i == 0 else '+loss'); vals.append(h.detach())
    if kld is not None:
        names.append('KLD'); vals.append(kld.detach())
    if div_reg is not None:
        names.append('DIV_REG'); vals.append(div_reg.detach())
    if gan:
        names += ['gen', 'dis']; vals += [gen_error.detach(), dis_error.detach()]
    if args.loss_contrastive_pos_weight > 0.0:
        names.append('c_pos'); vals.append(text_high_contrastive.detach())
    if args.loss_contrastive_neg_weight > 0.0:
        names.append('-c_neg'); vals.append(text_low_pos.detach())
    if args.loss_physical_weight > 0.0:
        names.append('phy'); vals.append(physical_loss.detach())
    from . import ddp
    err = ops.gru_cluster_error_tensor(dev)
    if err is not None and not (ddp.active() and not return_tensors):
        # cluster-GRU hand-off time-out flag (forward launches so far) rides along.  NOT under data parallelism: at this point the word is
        # rank-local, and a rank that raised here would leave its peers' collectives of this very step unmatched -- there the flag surfaces
        # through the end-of-step word, which every rank reads after the MAX-reduce (drain_cluster_errors_lagged)
        names.append('_cluster_err'); vals.append(err[0].to(torch.float32))
    packed = torch.stack(vals)
    readback = None
    if not return_tensors:
        host = _pinned(dev, 'loss', packed.numel())
        host.copy_(packed, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        readback = (host, ev)

    _phase('losses, D(fake), loss assembly')
    loss.backward()                                      # stage 1: losses, discriminator, generators (down to the cut)
    ops.side.flush(dev)                                  # ... their deferred side-stream weight gradients before the exchange / stage 2 / the optimizers
    _phase('backward stage 1 (losses, D, generators incl. BPTT)')
    works = []
    if ddp.active():
        for o in gen_optimizers:
            for tb in getattr(o, 'sparse_tables', ()):
                ddp.exchange_sparse_(tb)
        works = [ddp.average_(o.flat_g, async_op=True) if hasattr(o, 'flat_g') else ddp.average_module_grads_([o])
                 for o in gen_optimizers]
    # stage 2: the encoders, overlapping the collectives above.  The stand-alone text encoder goes FIRST (0.5 ms of kernels) so that its bucket
    # -- 30 MB with a dense embedding table -- is in flight under the audio tower's ~20 ms backward instead of after it.
    pairs = [(i, o, c.grad) for i, (o, c) in enumerate(zip(enc_outs, enc_cut)) if c is not o and c.grad is not None]
    text_pairs = [(o, g) for i, o, g in pairs if i == 4]
    audio_pairs = [(o, g) for i, o, g in pairs if i != 4]
    if text_pairs:
        torch.autograd.backward([p[0] for p in text_pairs], [p[1] for p in text_pairs])
        ops.side.flush(dev)
    _phase('text encoder backward')
    if ddp.active():
        for tb in getattr(text_optimizer, 'sparse_tables', ()):
            ddp.exchange_sparse_(tb)
        works.append(ddp.average_(text_optimizer.flat_g, async_op=True) if hasattr(text_optimizer, 'flat_g')
                     else ddp.average_module_grads_([text_optimizer]))
    if audio_pairs:
        torch.autograd.backward([p[0] for p in audio_pairs], [p[1] for p in audio_pairs])
        ops.side.flush(dev)
    _phase('audio tower backward')
    ev_bwd_done = None
    if comm_clock is not None and dev.type == 'cuda':
        ev_bwd_done = torch.cuda.Event(enable_timing=True)
        ev_bwd_done.record()
    _allreduce((audio_optimizer,))
    for w in works:
        if w is not None:
            w.wait()
    g_opts = tuple(gen_optimizers) + (audio_optimizer, text_optimizer)
    _sync_guard(dev)
    if ev_bwd_done is not None:
        ev_opt = torch.cuda.Event(enable_timing=True)
        ev_opt.record()
        comm_clock.append((ev_bwd_done, ev_opt))
    for o in g_opts:
        o.step()
    ops.gru_prep_clear()
    ops.rng.end_step()
    _phase('optimizers')

    if return_tensors:                                   # graph-captured steps read the packed buffer after replay
        err = ops.gru_cluster_error_tensor(dev)
        if err is not None:                              # the error word as it stands at the END of the step (BPTT launches included)
            if names[-1] == '_cluster_err':
                packed = packed[:-1]
            else:
                names.append('_cluster_err')
            packed = torch.cat([packed, err[:1].to(torch.float32)])
        return names, packed
    _watch_cluster_errors(dev)                            # BPTT hand-off time-outs: checked without blocking (see drain_cluster_errors)
    if host_clock is not None:
        host_clock['busy'] += time.perf_counter() - t_host0
        host_clock['steps'] += 1
    readback[1].synchronize()                            # the losses left the device before the backward started: no device-wide sync here
    return _ret_dict(args, names, readback[0].tolist())


def _ret_dict(args, names, vals):
    w = {'loss': args.loss_regression_weight, 'KLD': args.loss_kld_weight, 'DIV_REG': args.loss_reg_weight,
         'gen': args.loss_gan_weight, 'dis': 1.0, 'c_pos': args.loss_contrastive_pos_weight,
         'c_neg': args.loss_contrastive_neg_weight, 'phy': args.loss_physical_weight}
    ret = {}
    import numpy as np
    merged = []
    for n, v in zip(names, vals):                        # '+x' adds to the previous entry in fp32, '-x' negates
        if n.startswith('+'):
            merged[-1][1] = float(np.float32(merged[-1][1]) + np.float32(v))
        elif n.startswith('-'):
            merged.append([n[1:], -v])
        else:
            merged.append([n, v])
    for n, v in merged:
        if n == '_cluster_err':
            if v:
                raise ops.Ha2gClusterError(_CLUSTER_MSG)
            continue
        if n in ('KLD', 'DIV_REG') and not v:            # reference: `if kld:` / `if div_reg:` (train_hierarchy.py:277-280)
            continue
        ret[n] = w[n] * v
    return ret
