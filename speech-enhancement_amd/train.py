"""The GAN train / validation step of core/function.py (train_gan :182-343, validate_gan :346-451,
compute_self_correcting_loss_weights :705-760, batch_stft :664-683) on the HIP path.

One process per GPU.  Data parallelism (main_gan.py:133-188) = utterance sharding + one RCCL all-reduce of the
flat generator gradient buffer and one of the flat discriminator gradient buffer per step (torch.distributed
backend "nccl" == RCCL over xGMI) + the SyncBatchNorm statistic exchanges inside the Conformer conv modules.
PESQ labels are third-party CPU arithmetic (PyPI `pesq`, not available offline): they come from a pluggable
provider (`set_pesq_provider`); parity / bench runs supply them as inputs.
"""
import os
import time

import torch
import torch.distributed as dist

from . import frontend as FE
from . import layers as LY
from . import losses as LS
from . import ops as O
from .utils import AverageMeter, adjust_learning_rate

_PESQ_PROVIDER = None


def set_pesq_provider(fn):
    """fn(clean_list, noisy_list) -> FloatTensor [B] of (pesq - 1) / 3.5 on the current device."""
    global _PESQ_PROVIDER
    _PESQ_PROVIDER = fn


class PesqSideChannel:
    """Runs the PESQ provider off the critical path.  The reference calls batch_pesq between the
    generator and the discriminator step (core/function.py:283-287): a device-to-host copy that waits for the whole
    generator backward, then ~100 ms of CPU work with the GPU idle.  Here the enhanced audio is copied to pinned host
    memory on a side stream as soon as the iSTFT has produced it, a worker thread waits for that copy only and calls
    the provider while the GPU runs the generator backward, its optimizer step and the discriminator forwards; the
    main thread blocks on the result only where the loss needs it (L_E).  Same labels, same arithmetic order."""

    def __init__(self):
        from concurrent.futures import ThreadPoolExecutor
        self.pool = ThreadPoolExecutor(max_workers=1)
        self.stream = None
        self.host = {}
        self.calls = 0
        self.last_use = [None, None]          # the future that last read buffer set 0 / 1

    def _pinned(self, key, like):
        buf = self.host.get(key)
        if buf is None or buf.shape != like.shape:
            buf = torch.empty(like.shape, dtype=like.dtype, pin_memory=True)
            self.host[key] = buf
        return buf

    def _claim_set(self):
        """Pinned staging buffers are DOUBLE-BUFFERED: with stale_labels=True step t submits before step t - 1's future
        has been read, and the provider (~100 ms) is slower than the generator forward (~30 ms) -- one buffer set would let
        the new D2H copy overwrite the audio the worker is still scoring.  Set k is reused only after the future that last
        read it has finished (a no-op in the default schedule, where every future is consumed within its step)."""
        k = self.calls & 1
        self.calls += 1
        prev = self.last_use[k]
        if prev is not None and not prev.done():
            try:
                prev.result()
            except Exception:                 # the consumer of that future reports the provider error
                pass
        return k

    def submit(self, clean_dev, est_dev, extra=None):
        """clean_dev / est_dev: [B, L] device tensors that are complete on the current stream.  Returns a future of
        (q_est, q_extra...) CPU tensors; `extra` = optional dict name -> [B, L] device tensor scored against clean."""
        if self.stream is None:
            self.stream = torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        names = ['clean', 'est'] + sorted(extra or {})
        tensors = [clean_dev, est_dev] + [extra[k] for k in sorted(extra or {})]
        bset = self._claim_set()
        with torch.cuda.stream(self.stream):
            host = []
            for n, t in zip(names, tensors):
                t = t.detach()
                t.record_stream(self.stream)
                h = self._pinned((bset, n), t)
                h.copy_(t, non_blocking=True)
                host.append(h)
            ev = torch.cuda.Event()
            ev.record(self.stream)

        dev = est_dev.device

        def work():
            # the current device is thread-local and a new thread starts on device 0: pin the worker to the caller's
            # device so a provider that honours the "current device" contract never opens a context on GPU 0 from
            # rank k > 0; results are handed back as CPU tensors and uploaded by the main thread on its own stream
            with torch.cuda.device(dev):
                ev.synchronize()
                clean_list = list(host[0].numpy())
                return [torch.as_tensor(pesq_labels(clean_list, list(h.numpy()), device='cpu')).float().cpu()
                        for h in host[1:]]
        fut = self.pool.submit(work)
        self.last_use[bset] = fut
        return fut


_SIDE = None


def pesq_side_channel():
    global _SIDE
    if _SIDE is None:
        _SIDE = PesqSideChannel()
    return _SIDE


def pesq_labels(clean_list, noisy_list, device=None):
    """models/discriminator.py:25-32 (`batch_pesq`): (pesq - 1) / 3.5 per clip, -1 on a PESQ exception.  `device`:
    where the label tensor should live (the reference hard-codes 'cuda'; default = the current CUDA device)."""
    if _PESQ_PROVIDER is not None:
        return _PESQ_PROVIDER(clean_list, noisy_list)
    try:
        from pesq import pesq          # noqa: F401  (third-party; models/discriminator.py:17-32)
    except ImportError as e:
        raise RuntimeError('no PESQ provider: install `pesq` or call train.set_pesq_provider(fn)') from e
    import numpy as np
    from joblib import Parallel, delayed

    def one(c, n):
        try:
            return pesq(16000, c, n, 'wb')
        except Exception:
            return -1
    s = np.array(Parallel(n_jobs=-1)(delayed(one)(c, n) for c, n in zip(clean_list, noisy_list)))
    q = torch.FloatTensor((s - 1) / 3.5)
    return q.to(device if device is not None else torch.device('cuda', torch.cuda.current_device()))


class DataParallelHooks(LY.DPHooks):
    """RCCL hooks: SyncBatchNorm statistic all-reduce + flat gradient all-reduce (average).

    Backend "nccl" (== RCCL over xGMI) reduces the device buffers in place.  Backend "gloo" is the test transport
    (several ranks sharing ONE GPU, which RCCL refuses): device tensors are staged through the host around the
    collective, so exactly the same step logic runs in both cases."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.stage_host = dist.get_backend(group) == 'gloo'
        # bench.py --gpus N: when a list, every exchange appends (kind, start event, end event) recorded on the stream the
        # exchange is issued on ("comm" object of the bench line); None = off, nothing is recorded
        self.comm_events = None

    def _mark(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def allreduce(self, t, kind='stats'):
        """small blocking-in-stream all-reduce (SUM): SyncBatchNorm statistics (256 / 384 doubles), the scp gradient triple"""
        e0 = self._mark() if (self.comm_events is not None and kind is not None and t.is_cuda) else None
        if self.stage_host and t.is_cuda:
            h = t.cpu()
            dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, group=self.group)
        if e0 is not None:
            self.comm_events.append((kind, e0, self._mark()))
        return t

    def average_grads(self, optimizer, kind='grads_d'):
        self.finish_average(self.start_average(optimizer, kind))

    def start_average(self, optimizer, kind='grads_g'):
        """launch the all-reduce of the flat gradient buffers asynchronously (RCCL runs on its own stream)"""
        e0 = self._mark() if self.comm_events is not None else None
        if self.stage_host:
            works = [(self.allreduce(g, kind=None), None) for g in optimizer.flat_grads()]
        else:
            works = [(g, dist.all_reduce(g, group=self.group, async_op=True)) for g in optimizer.flat_grads()]
        return works, kind, e0

    def finish_average(self, handle):
        works, kind, e0 = handle
        # `<kind>_exposed`: from the moment the consumer stream has nothing else left to do (everything issued before this
        # point) until the reduced buffer is usable = the part of the collective that was NOT hidden behind other work
        e_pre = self._mark() if e0 is not None else None
        for g, w in works:
            if w is not None:
                w.wait()
            g.mul_(1.0 / self.world)
        if e0 is not None:
            e1 = self._mark()
            self.comm_events.append((kind, e0, e1))
            self.comm_events.append((kind + '_exposed', e_pre, e1))


def attach_data_parallel(model, discriminator, group=None):
    """main_gan.py:154-171 counterpart: broadcast rank-0 parameters/buffers once, install the hooks."""
    hooks = DataParallelHooks(group)
    for m in (model, discriminator):
        for t in list(m.parameters()) + list(m.buffers()):
            if hooks.stage_host and t.is_cuda:
                h = t.data.cpu()
                dist.broadcast(h, 0, group=group)
                t.data.copy_(h)
            else:
                dist.broadcast(t.data, 0, group=group)
    model.dp = hooks
    return hooks


def batch_stft(batch, args, config):
    """core/function.py:664-683 on the kernel path.  Returns the reference's 8-tuple (complex [B,F,T] specs)."""
    clean, noisy = batch['audio'], batch['noisy']
    if getattr(args, 'gpu', None) is not None:
        clean, noisy = clean.cuda(args.gpu, non_blocking=True), noisy.cuda(args.gpu, non_blocking=True)
    c = O.clip_scale(noisy.contiguous())
    npl, npad = FE.stft_planes(noisy, config.N_FFT, config.HOP_SAMPLES, 'pow', scale=c)
    cpl, cpad = FE.stft_planes(clean, config.N_FFT, config.HOP_SAMPLES, 'pow', scale=c)
    h = config.N_FFT // 2
    clean_n, noisy_n = cpad[:, h:h + clean.shape[1]], npad[:, h:h + noisy.shape[1]]
    clean_spec, noisy_spec = FE.planes_to_spec(cpl), FE.planes_to_spec(npl)
    one_labels = torch.ones(len(clean), device=clean.device)
    window = torch.hamming_window(config.N_FFT, device=clean.device)
    return clean_n, noisy_n, clean_spec, noisy_spec, clean_spec.real.unsqueeze(1), clean_spec.imag.unsqueeze(1), \
        one_labels, window


def _mse(a, b):
    return ((a - b) ** 2).mean()


def self_correcting_weights(CE, CN, EN, EE, NN):
    """core/function.py:736-748 (EE, NN include the +1e-14)."""
    if CE > 0:
        wE = 1.0
        wN = 1.0 if (CN + wE * EN) > 0 else -(CN) / NN - (EN) / NN
    else:
        wE = -(CE) / EE
        wN = 1.0 if (CN + wE * EN) > 0 else -(CN) / NN + (CE * EN) / (EE * NN)
    return 1.0, wE, wN


_D_OVERLAP = os.environ.get('SE_NO_D_OVERLAP') != '1'
_SIDE_STREAMS = {}


def _side_stream(device):
    """one side stream per device for the discriminator step (see _gan_step)"""
    key = torch.device(device).index
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)     # (a high-priority stream measured no different: 99.1 vs 99.1 ms)
    return _SIDE_STREAMS[key]


class LabelCache:
    """the reference recomputes Q_y_y = PESQ(clean, clean) and Q_x_y = PESQ(clean, noisy)
    every step of the scp / sc recipes (core/function.py:292-301) although neither depends on the generator -- two of
    the three PESQ batches per step are constants of the crop.  Labels are cached per utterance under a caller-supplied
    key (`batch['keys']`: one hashable per clip that identifies file AND crop, e.g. (index, crop_start)); only the
    missing clips of a batch go to the provider.  Without keys nothing is cached (the crop is unknown)."""

    def __init__(self, max_items=1 << 20):
        self.q = {}
        self.max_items = max_items
        self.hits = self.misses = 0

    def lookup(self, keys, kind):
        """-> (values list with None for misses, indices of the misses)"""
        vals = [self.q.get((kind, k)) for k in keys]
        miss = [i for i, v in enumerate(vals) if v is None]
        self.hits += len(keys) - len(miss)
        self.misses += len(miss)
        return vals, miss

    def store(self, keys, kind, values):
        if len(self.q) + len(keys) > self.max_items:
            self.q.clear()
        for k, v in zip(keys, values):
            self.q[(kind, k)] = float(v)


_LABEL_CACHE = LabelCache()


def label_cache():
    return _LABEL_CACHE


def generator_pass(model, discriminator, clean, noisy, arch, loss_weights, n_fft=400, hop=100, comp_type='pow',
                   gan_on=True, grad=True, after_istft=None):
    """The shared front half of train_gan and validate_gan (core/function.py:216-272 == :362-413): normalise, 2 STFT,
    generator, iSTFT, the cmgan or consistency-preserving (scp / cp) spectral + time losses, and the GAN term unless the
    `--gen-first` gate is closed.  grad=False (validation): nothing is recorded for backward.  `after_istft(r)` is
    called as soon as the enhanced audio exists (PESQ side channel).  Returns a dict of tensors."""
    B, Ls = clean.shape
    r = {}
    c = O.clip_scale(noisy.contiguous())
    noisy_pl, noisy_pad = FE.stft_planes(noisy, n_fft, hop, 'pow', scale=c)
    clean_pl, clean_pad = FE.stft_planes(clean, n_fft, hop, 'pow', scale=c)
    h = n_fft // 2
    clean_n, noisy_n = clean_pad[:, h:h + Ls], noisy_pad[:, h:h + Ls]
    est = model.forward_planes(noisy_pl)
    est_audio = FE.istft_planes(est, n_fft, hop, 'pow')
    r.update(noisy_pl=noisy_pl, clean_pl=clean_pl, clean_n=clean_n, noisy_n=noisy_n, est=est, est_audio=est_audio)
    if after_istft is not None:
        after_istft(r)
    if arch in ('scp', 'cp'):
        # enhanced-audio pipeline: re-STFT with args.comp_type; clean* pipeline: iSTFT -> STFT of the clean spectrum
        est_p = FE.stft_planes_grad(est_audio, n_fft, hop, comp_type) if grad else \
            FE.stft_planes(est_audio, n_fft, hop, comp_type)[0]
        with torch.no_grad():
            clean_audio_p = FE.istft_planes(clean_pl, n_fft, hop, 'pow')
            clean_p, _ = FE.stft_planes(clean_audio_p, n_fft, hop, comp_type)
        loss_mag, loss_ri = LS.spec_losses(est_p, clean_p)
        time_loss = LS.l1_time_loss(est_audio, clean_audio_p)
    else:
        loss_mag, loss_ri = LS.spec_losses(est, clean_pl)
        time_loss = LS.l1_time_loss(est_audio, clean_n)
    w = loss_weights
    ones = torch.ones(B, device=clean.device)
    if gan_on:
        gen_gan = _mse(discriminator.forward_planes(clean_pl, est, detach_params=True).flatten(), ones)
        loss = w[0] * loss_ri + w[1] * loss_mag + w[2] * time_loss + w[3] * gen_gan
    else:
        gen_gan = torch.zeros((), device=clean.device)
        loss = w[0] * loss_ri + w[1] * loss_mag + w[2] * time_loss
    r.update(loss_mag=loss_mag, loss_ri=loss_ri, time_loss=time_loss, gan=gen_gan, loss=loss, ones=ones)
    return r


def clip_grad_norm(optimizer, params, max_norm):
    """torch.nn.utils.clip_grad_norm_ (core/function.py:275-276, 311-312) on the flat gradient buffers when the
    optimizer has them (one fused sum-of-squares + one scale launch per buffer, no host sync)."""
    if hasattr(optimizer, 'clip_grad_norm'):
        return optimizer.clip_grad_norm(max_norm)
    return torch.nn.utils.clip_grad_norm_(params, max_norm)


class _StaleState:
    """what the one-step-stale discriminator update of step t needs from step t - 1"""
    __slots__ = ('clean_pl', 'noisy_pl', 'est_d', 'pending', 'ones', 'keys')


def _discriminator_update(discriminator, optimizer_disc, est_d, clean_pl, noisy_pl, ones, arch, labels, pending, keys, hooks,
                          max_norm, out):
    """the discriminator half of the train_gan loop body (core/function.py:286-315) on the current stream: forwards on
    (clean, est.detach()), (clean, clean) (and (clean, noisy) for scp / sc), metric losses (self-correcting weights for scp / sc),
    backward, clipping, optimizer step; fills L_C / L_E / loss_d (and L_N / w_E / w_N) of `out`."""
    d_gx = discriminator.forward_planes(clean_pl, est_d)
    d_yy = discriminator.forward_planes(clean_pl, clean_pl)
    d_xy = discriminator.forward_planes(clean_pl, noisy_pl) if arch in ('scp', 'sc') else None
    if labels is None:                     # all discriminator forwards are queued: only now wait for the CPU side
        got = dict(zip(pending['names'], pending['future'].result()))
        if keys is not None:
            for kind in ('clean', 'noisy'):
                if kind in got:
                    _LABEL_CACHE.store(keys, kind, got[kind].tolist())
        got.update(pending['cached'])
        labels = {k: v.to(clean_pl.device, non_blocking=True) for k, v in got.items()}
    q_est = labels['est']
    L_E = _mse(d_gx.flatten(), q_est)
    if arch in ('scp', 'sc'):
        q_clean = labels['clean']
        L_C = _mse(d_yy.flatten(), q_clean)
        q_noisy = labels['noisy']
        L_N = _mse(d_xy.flatten(), q_noisy)
        loss_d, wE, wN = _self_correcting_backward(discriminator, optimizer_disc, L_C, L_E, L_N, hooks)
        out.update(L_N=L_N.detach(), w_E=wE, w_N=wN)
    else:
        L_C = _mse(d_yy.flatten(), ones)
        loss_d = L_C + L_E
        loss_d.backward()
        if hooks is not None:
            hooks.average_grads(optimizer_disc)
    if max_norm != 0.0:
        clip_grad_norm(optimizer_disc, discriminator.parameters(), max_norm)
    optimizer_disc.step()
    out.update(L_C=L_C.detach(), L_E=L_E.detach(), loss_d=loss_d.detach())


def gan_step(model, discriminator, optimizer, optimizer_disc, clean, noisy, *args, **kwargs):
    """`_gan_step` inside the step's zero-filled scratch arena (ops.ZeroArena: one fill launch instead of ~250)."""
    O.ARENA.begin(clean.device)
    try:
        return _gan_step(model, discriminator, optimizer, optimizer_disc, clean, noisy, *args, **kwargs)
    finally:
        O.ARENA.end()


def _gan_step(model, discriminator, optimizer, optimizer_disc, clean, noisy, arch, loss_weights, n_fft=400, hop=100,
              comp_type='pow', max_norm=0.0, gan_on=True, labels=None, hooks=None, keys=None, stale_labels=False):
    """One iteration of the train_gan loop body (core/function.py:216-315).  `labels`: dict with 'est'
    (and 'clean', 'noisy' for scp/sc) giving the PESQ targets directly; if None the PESQ provider is called on the
    audio like the reference does (asynchronously, see PesqSideChannel); `keys`: optional per-clip crop keys for the
    crop-constant label cache (LabelCache).  `stale_labels=True` (opt-in, NOT the reference's
    schedule): the discriminator update of this call uses the batch, the enhanced spectrum and the PESQ labels of the
    PREVIOUS call (kept on `discriminator._stale`), so the label latency is hidden behind a whole step instead of the
    generator backward; the first call then makes no discriminator update.
    Returns a dict of python-float-convertible loss tensors (no host sync)."""
    B, Ls = clean.shape
    out = {}
    optimizer.zero_grad()
    pending = {}

    def submit_labels(r):
        # PESQ labels: start the host side now, collect it at the discriminator loss
        if labels is not None or not gan_on:
            return
        est_audio = r['est_audio']
        length = est_audio.size(-1)
        extra, cached = None, {}
        if arch in ('scp', 'sc'):
            extra = {'clean_self': r['clean_n'][:, :length], 'noisy': r['noisy_n'][:, :length]}
            if keys is not None:          # crop-constant labels: only when the whole batch hits (one provider call less)
                for name, kind in (('clean_self', 'clean'), ('noisy', 'noisy')):
                    vals, miss = _LABEL_CACHE.lookup(keys, kind)
                    if not miss:
                        cached[kind] = torch.tensor(vals, dtype=torch.float32)
                        del extra[name]
        pending['future'] = pesq_side_channel().submit(r['clean_n'][:, :length], est_audio, extra or None)
        pending['names'] = ['est'] + [{'clean_self': 'clean', 'noisy': 'noisy'}[k] for k in sorted(extra or {})]
        pending['cached'] = cached

    r = generator_pass(model, discriminator, clean, noisy, arch, loss_weights, n_fft, hop, comp_type, gan_on, True,
                       submit_labels)
    est, clean_pl, noisy_pl, ones, loss = r['est'], r['clean_pl'], r['noisy_pl'], r['ones'], r['loss']
    # The discriminator step (2-3 forwards, backward, optimizer: ~150 launches of latency-bound kernels on a few workgroups each,
    # 6-7 ms when run alone) reads est.detach() and the discriminator's own parameters only.  It is issued on a SIDE STREAM and runs
    # concurrently with the generator backward (55 ms of large kernels), whose gaps and tails it fills.  For that the backward is
    # taken in two stages: loss -> est first (loss kernels, iSTFT / re-STFT backward and the input gradient THROUGH the
    # discriminator, which must read the discriminator's PReLU / InstanceNorm parameters before the side stream updates them),
    # an event, then est -> generator parameters.  Same dataflow, same results; SE_NO_D_OVERLAP=1 restores the serial order.
    overlap = _D_OVERLAP and gan_on and not stale_labels and est.is_cuda and est.requires_grad
    main = side = ev = None
    if overlap:
        main, side = torch.cuda.current_stream(est.device), _side_stream(est.device)
        dest, = torch.autograd.grad(loss, est)
        ev = torch.cuda.Event()
        ev.record(main)
        est.backward(dest)          # enqueued BEFORE the discriminator step: the host is never blocked (PESQ wait) ahead of it
        del dest
    else:
        loss.backward()
    # Data parallel: the generator-gradient all-reduce (7.3 MB over xGMI) is launched asynchronously and only waited
    # for after the discriminator step has been issued -- the discriminator step reads est.detach() and no generator
    # parameter, so deferring optimizer.step() past it changes no result and hides the collective behind the three
    # discriminator forwards + backward (on a single GPU the order below is exactly the reference's).
    g_works = hooks.start_average(optimizer) if hooks is not None else None

    def finish_generator_step():
        if g_works is not None:
            hooks.finish_average(g_works)
        if max_norm != 0.0:
            clip_grad_norm(optimizer, model.parameters(), max_norm)
        optimizer.step()

    if hooks is None:
        finish_generator_step()
    out.update(loss_ri=r['loss_ri'].detach(), loss_mag=r['loss_mag'].detach(), time_loss=r['time_loss'].detach(),
               gan=r['gan'].detach(), loss_g=loss.detach())

    if overlap:
        side.wait_event(ev)
        try:
            with torch.cuda.stream(side):
                optimizer_disc.zero_grad()      # on the side stream too: ordered before the backward that accumulates into it
                _discriminator_update(discriminator, optimizer_disc, est.detach(), clean_pl, noisy_pl, ones, arch, labels, pending,
                                      keys, hooks, max_norm, out)
        finally:
            # ALWAYS joined (also when the PESQ provider or a kernel launch raised): the caller's `finally` recycles the step's
            # scratch arena on the main stream, which must not overtake work the side stream has already been given
            main.wait_stream(side)  # the next generator forward reads the updated discriminator; scratch is recycled per step
        if hooks is not None:
            finish_generator_step()
        return out
    optimizer_disc.zero_grad()
    if not gan_on:
        if hooks is not None:
            finish_generator_step()
        out['loss_d'] = torch.zeros((), device=clean.device)
        return out
    est_d = est.detach()
    if stale_labels and labels is None:
        cur = _StaleState()
        cur.clean_pl, cur.noisy_pl, cur.est_d, cur.pending, cur.ones = clean_pl, noisy_pl, est_d, dict(pending), ones
        cur.keys = keys
        prev = getattr(discriminator, '_stale', None)
        discriminator._stale = cur
        if prev is None or prev.ones.shape != ones.shape:
            if hooks is not None:
                finish_generator_step()
            out['loss_d'] = torch.zeros((), device=clean.device)
            return out
        clean_pl, noisy_pl, est_d, pending, ones = prev.clean_pl, prev.noisy_pl, prev.est_d, prev.pending, prev.ones
        keys = prev.keys
    _discriminator_update(discriminator, optimizer_disc, est_d, clean_pl, noisy_pl, ones, arch, labels, pending, keys, hooks,
                          max_norm, out)
    if hooks is not None:
        finish_generator_step()
    return out


def _self_correcting_backward(discriminator, optimizer_disc, L_C, L_E, L_N, hooks):
    """compute_self_correcting_loss_weights + the caller's backward (core/function.py:705-760, 310): three
    gradient vectors, six dot products, piecewise weights; the optimizer then sees 2x the combined gradient, as in
    the reference (param.grad is written, then backward() on the weighted loss adds the same quantity again).
    Multi-GPU semantics (undefined in the reference): the three gradients are averaged over
    ranks BEFORE the dot products, so every rank computes identical weights == the single-process result at the
    global batch."""
    params = [p for p in discriminator.parameters()]
    flats = []
    for L_ in (L_C, L_E, L_N):
        gs = torch.autograd.grad(L_, params, retain_graph=True, allow_unused=True)
        flats.append(torch.cat([(g if g is not None else torch.zeros_like(p)).reshape(-1) for g, p in zip(gs, params)]))
    if hooks is not None:
        cat = torch.cat(flats)             # one 2.18 MB all-reduce instead of three
        hooks.allreduce(cat, kind='scp_grad_triple')
        cat.mul_(1.0 / hooks.world)
        flats = list(cat.split(flats[0].numel()))
    C, E, N = flats
    dots = torch.zeros(5, device=C.device, dtype=torch.float64)
    for i, (a, b) in enumerate(((E, E), (N, N), (C, E), (C, N), (E, N))):
        O.dot(a, b, dots[i:i + 1])
    EE, NN, CE, CN, EN = (float(v) for v in dots.cpu())
    wC, wE, wN = self_correcting_weights(CE, CN, EN, EE + 1e-14, NN + 1e-14)
    comb = O.axpbypcz(C, E, N, 2.0 * wC, 2.0 * wE, 2.0 * wN)
    o = 0
    for p in params:
        k = p.numel()
        p.grad.copy_(comb[o:o + k].view_as(p))
        o += k
    return (wC * L_C + wE * L_E + wN * L_N), wE, wN


def _to_gpu(batch, args):
    clean, noisy = batch['audio'], batch['noisy']
    if getattr(args, 'gpu', None) is not None:
        clean, noisy = clean.cuda(args.gpu, non_blocking=True), noisy.cuda(args.gpu, non_blocking=True)
    return clean, noisy


def train_gan(train_loader, model, discriminator, criterion, optimizer, optimizer_disc, logger, epoch, args, config):
    """core/function.py:182-343: same signature and return values (avg generator / discriminator loss)."""
    if getattr(args, 'debug', False):
        torch.autograd.set_detect_anomaly(True)
    batch_time, data_time, gen_losses, disc_losses = AverageMeter(), AverageMeter(), AverageMeter(), AverageMeter()
    model.train()
    discriminator.train()
    hooks = model.dp if getattr(model.dp, 'world', 1) > 1 else None
    start = end = time.time()
    iters = len(train_loader)
    for idx, batch in enumerate(train_loader):
        data_time.update(time.time() - end)
        adjust_learning_rate([optimizer, optimizer_disc], epoch + idx / iters, config)
        clean, noisy = _to_gpu(batch, args)
        gan_on = epoch >= int(args.epochs * 0.3) or not args.gen_first
        out = gan_step(model, discriminator, optimizer, optimizer_disc, clean, noisy, args.arch, config.LOSS_WEIGHTS,
                       config.N_FFT, config.HOP_SAMPLES, args.comp_type, args.max_norm, gan_on,
                       labels=batch.get('labels'), hooks=hooks, keys=batch.get('keys'))
        torch.cuda.synchronize()
        gen_losses.update(out['loss_g'].item(), clean.size(0))
        disc_losses.update(out['loss_d'].item(), clean.size(0))
        batch_time.update(time.time() - end)
        end = time.time()
        if idx % args.print_freq == 0 and logger is not None:
            logger.info(f'Train: [{epoch}/{args.epochs}][{idx}/{iters}]\t'
                        f'lr {optimizer.param_groups[0]["lr"]:.6f}\ttime {batch_time.val:.4f} ({batch_time.avg:.4f})\t'
                        f'generator loss {gen_losses.val:.4f} ({gen_losses.avg:.4f})\t'
                        f'discriminator loss {disc_losses.val:.4f} ({disc_losses.avg:.4f})\t'
                        f'mem {torch.cuda.max_memory_allocated() / 2 ** 20:.0f}MB')
    if logger is not None:
        logger.info(f'EPOCH {epoch} training takes {time.time() - start:.0f}s')
    return gen_losses.avg, disc_losses.avg


@torch.no_grad()
def validate_gan(valid_loader, model, discriminator, criterion, logger, epoch, args, config):
    """core/function.py:346-451: the generator loss follows the training recipe -- consistency-preserving losses for
    scp / cp (:373-397), GAN term gated by --gen-first before 0.3 * epochs (:401-413); the discriminator loss is
    always L_C(ones) + L_E ("unable to compute validation self-correcting loss", :427-428).  No backward."""
    model.eval()
    discriminator.eval()
    gen_losses, disc_losses = AverageMeter(), AverageMeter()
    arch = getattr(args, 'arch', 'cmgan')
    gan_on = epoch >= int(getattr(args, 'epochs', 0) * 0.3) or not getattr(args, 'gen_first', False)
    for batch in valid_loader:
        clean, noisy = _to_gpu(batch, args)
        B = clean.shape[0]
        r = generator_pass(model, discriminator, clean, noisy, arch, config.LOSS_WEIGHTS, config.N_FFT,
                           config.HOP_SAMPLES, getattr(args, 'comp_type', 'pow'), gan_on, grad=False)
        est_audio, clean_n, clean_pl = r['est_audio'], r['clean_n'], r['clean_pl']
        labels = batch.get('labels')
        d_gx = discriminator.forward_planes(clean_pl, r['est'])
        if labels is not None:
            q = labels['est']
        else:
            length = est_audio.size(-1)
            q = pesq_labels(list(clean_n[:, :length].cpu().numpy()), list(est_audio.cpu().numpy()),
                            device=clean.device).to(clean.device)
        d_yy = discriminator.forward_planes(clean_pl, clean_pl)
        loss_d = _mse(d_yy.flatten(), r['ones']) + _mse(d_gx.flatten(), q)
        gen_losses.update(r['loss'].item(), B)
        disc_losses.update(loss_d.item(), B)
    return gen_losses.avg, disc_losses.avg
