#!/usr/bin/env python3
"""Counterpart of the reference trainer (``model/traintest_MegaCRN.py``) around the HIP hot path.

Same flags and defaults (``:158-188``), same semantics:
  * data: ``{train,val,test}.npz`` with ``x, y`` of shape (S, T, N, 2) (``generate_training_data.py:97-103``),
    ``StandardScaler`` fitted on ``x_train[..., 0]`` and applied to channel 0 of x AND y (``:274-277``);
  * loader: shuffle ONCE at construction, pad the last batch by repeating the last sample
    (``model/utils.py:6-43``);
  * step: ``FlatTrainer.train_step`` = forward, 3-term loss on inverse-transformed tensors (``:118-125``),
    backward, ``clip_grad_norm_(max_grad_norm)``, Adam(lr, eps) (``:104,:129-130``); ``batches_seen`` drives the
    curriculum (``:117,:127``);
  * epoch: ``MultiStepLR(steps, lr_decay_ratio)`` stepped per epoch (``:105,:132``), val + test evaluation every
    epoch (``:133,:139``), best-val checkpoint of ``state_dict`` (``:141-144``), early stop after ``patience`` bad
    epochs (``:146-150``), reload best and test (``:153-155``);
  * evaluate: masked MAE / MAPE / RMSE overall and at horizons 3 / 6 / 12 as means of per-batch values
    (``:50-99``).
Multi-GPU: launch with ``python -m torch.distributed.run --nproc-per-node N -m megacrn_amd.train ...``; every rank
takes its (equal) block of each batch, one all-reduce per step; rank 0's validation figures drive checkpointing and
early stopping on every rank.  Inputs travel through a pinned, double-buffered copy stream (``Prefetcher``), the
evaluation figures are accumulated on the device by one launch per batch (``DeviceMetrics``).  ``--synthetic`` generates METR-LA-shaped windows when the
datasets are absent (they are not shipped with the reference either).
"""
from __future__ import annotations

import argparse
import logging
import os
import time

import numpy as np
import torch


# ---------------------------------------------------------------------------------------------
# data side (numpy, host) -- model/utils.py:6-54
# ---------------------------------------------------------------------------------------------
class StandardScaler:
    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def transform(self, data):
        return (data - self.mean) / self.std

    def inverse_transform(self, data):
        return (data * self.std) + self.mean


class DataLoader:
    """Shuffles once (if asked) and pads with the last sample so every batch is full."""

    def __init__(self, xs, ys, batch_size, pad_with_last_sample=True, shuffle=False):
        self.batch_size = batch_size
        if pad_with_last_sample:
            n_pad = (batch_size - (len(xs) % batch_size)) % batch_size
            xs = np.concatenate([xs, np.repeat(xs[-1:], n_pad, axis=0)], axis=0)
            ys = np.concatenate([ys, np.repeat(ys[-1:], n_pad, axis=0)], axis=0)
        self.size = len(xs)
        self.num_batch = self.size // batch_size
        if shuffle:
            perm = np.random.permutation(self.size)
            xs, ys = xs[perm], ys[perm]
        self.xs, self.ys = xs, ys

    def get_iterator(self):
        for i in range(self.num_batch):
            lo = self.batch_size * i
            hi = min(self.size, lo + self.batch_size)
            yield self.xs[lo:hi], self.ys[lo:hi]


def synthetic_windows(num_samples, T, N, seed):
    """METR-LA-like windows: speed channel with a daily profile + noise + 8 % missing (exact zeros),
    time-of-day channel as in generate_training_data.py:31-34."""
    rng = np.random.default_rng(seed)
    t0 = rng.integers(0, 288 * 7, size=num_samples)
    steps = t0[:, None] + np.arange(2 * T)[None, :]
    tod = (steps % 288) / 288.0
    base = 55 + 10 * np.sin(2 * np.pi * tod)[:, :, None] + 5 * rng.standard_normal((1, 1, N))
    speed = base + 3 * rng.standard_normal((num_samples, 2 * T, N))
    speed[rng.random(speed.shape) < 0.08] = 0.0
    data = np.stack([speed, np.broadcast_to(tod[:, :, None], speed.shape)], axis=-1)
    return data[:, :T].copy(), data[:, T:].copy()


# ---------------------------------------------------------------------------------------------
def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--dataset', type=str, choices=['METRLA', 'PEMSBAY'], default='METRLA')
    p.add_argument('--data_dir', type=str, default=None, help='directory with train/val/test.npz (default ../<dataset>)')
    p.add_argument('--synthetic', type=int, default=0, help='>0: generate this many synthetic training windows')
    p.add_argument('--trainval_ratio', type=float, default=0.8, help='accepted for command-line parity; like the reference '
                   '(:160, never read after parsing) it does not change the pre-split train/val/test.npz files')
    p.add_argument('--val_ratio', type=float, default=0.125, help='accepted for command-line parity (:161, unused there too)')
    p.add_argument('--num_nodes', type=int, default=207)
    p.add_argument('--seq_len', type=int, default=12)
    p.add_argument('--horizon', type=int, default=12)
    p.add_argument('--input_dim', type=int, default=1)
    p.add_argument('--output_dim', type=int, default=1)
    p.add_argument('--max_diffusion_step', type=int, default=3, help='max diffusion step or Cheb K')
    p.add_argument('--num_rnn_layers', type=int, default=1)
    p.add_argument('--rnn_units', type=int, default=64)
    p.add_argument('--mem_num', type=int, default=20)
    p.add_argument('--mem_dim', type=int, default=64)
    p.add_argument('--loss', type=str, default='mask_mae_loss')
    p.add_argument('--lamb', type=float, default=0.01)
    p.add_argument('--lamb1', type=float, default=0.01)
    p.add_argument('--epochs', type=int, default=200)
    p.add_argument('--patience', type=int, default=20)
    p.add_argument('--batch_size', type=int, default=64)
    p.add_argument('--lr', type=float, default=0.01)
    p.add_argument('--steps', type=eval, default=[50, 100])
    p.add_argument('--lr_decay_ratio', type=float, default=0.1)
    p.add_argument('--epsilon', type=float, default=1e-3)
    p.add_argument('--max_grad_norm', type=int, default=5)
    p.add_argument('--use_curriculum_learning', type=eval, choices=[True, False], default='True')
    p.add_argument('--cl_decay_steps', type=int, default=2000)
    p.add_argument('--test_every_n_epochs', type=int, default=5)
    p.add_argument('--gpu', type=int, default=0)
    p.add_argument('--seed', type=int, default=None, help='the reference leaves seeding commented out (:264-266)')
    p.add_argument('--save_dir', type=str, default='../save')
    p.add_argument('--max_batches', type=int, default=0, help='>0: stop each epoch after this many batches (smoke runs)')
    return p


def load_data(args):
    if args.synthetic > 0:
        n = args.synthetic
        sets = {}
        for cat, cnt, seed in (('train', n, 0), ('val', max(n // 8, args.batch_size), 1), ('test', max(n // 4, args.batch_size), 2)):
            sets['x_' + cat], sets['y_' + cat] = synthetic_windows(cnt, args.seq_len, args.num_nodes, seed)
        return sets
    d = args.data_dir or f'../{args.dataset}'
    sets = {}
    for cat in ('train', 'val', 'test'):
        z = np.load(os.path.join(d, cat + '.npz'))
        sets['x_' + cat], sets['y_' + cat] = z['x'], z['y']
    return sets


def split_x_y(x, y, args, lo=None, hi=None):
    """Host half of prepare_x_y (:33-48): channel split, float32; optional [lo,hi) = this rank's shard."""
    if lo is not None:
        x, y = x[lo:hi], y[lo:hi]
    x0 = np.ascontiguousarray(x[..., :args.input_dim], dtype=np.float32)
    y0 = np.ascontiguousarray(y[..., :args.output_dim], dtype=np.float32)
    y1 = np.ascontiguousarray(y[..., args.output_dim:], dtype=np.float32)
    return x0, y0, y1


def prepare_x_y(x, y, args, device, lo=None, hi=None):
    """(:33-48) channel split + float32 + synchronous H2D -> x, y, y_cov."""
    return tuple(torch.from_numpy(a).to(device) for a in split_x_y(x, y, args, lo, hi))


class Prefetcher:
    """Pinned, double-buffered host-to-device input pipeline (the H2D boundary of :116, once per batch).

    While the GPU works on batch i, the host thread splits batch i+1 into pinned staging buffers and a
    dedicated copy stream moves it into the other device slot; the compute stream only waits on the copy's
    event.  A slot is refilled only after the compute stream has passed the event recorded when its previous
    occupant was consumed (`release`), so nothing is overwritten while kernels still read it.  Order and contents
    are exactly those of iterating the loader and calling prepare_x_y on every batch."""

    def __init__(self, iterator, args, device, shard=None, depth=2):
        self.it, self.args, self.device, self.shard, self.depth = iterator, args, device, shard, depth
        self.copy = torch.cuda.Stream(device=device)
        self.slots = [None] * depth          # per slot: (pinned triple, device triple)
        self.ready = [torch.cuda.Event() for _ in range(depth)]
        self.free = [None] * depth           # event after which the slot's device buffers may be overwritten
        self.queue = []
        self.n = 0
        self._fill()

    def _fill(self):
        while len(self.queue) < self.depth:
            try:
                x, y = next(self.it)
            except StopIteration:
                return
            lo, hi = self.shard(len(x)) if self.shard else (None, None)
            host = split_x_y(x, y, self.args, lo, hi)
            k = self.n % self.depth
            self.n += 1
            if self.slots[k] is None or any(p.shape != h.shape for p, h in zip(self.slots[k][0], host)):
                pin = tuple(torch.empty(h.shape, dtype=torch.float32).pin_memory() for h in host)
                dev = tuple(torch.empty(h.shape, dtype=torch.float32, device=self.device) for h in host)
                # the blocks come from the caching allocator on the CONSUMER stream but are first written on the copy
                # stream: order the copy stream behind whatever last used those blocks, and tell the allocator about
                # the second stream so that a dropped Prefetcher cannot hand a block out while an H2D is in flight
                self.copy.wait_stream(torch.cuda.current_stream(self.device))
                for d in dev:
                    d.record_stream(self.copy)
                self.slots[k] = (pin, dev)
            pin, dev = self.slots[k]
            if self.free[k] is not None:
                self.ready[k].synchronize()                    # the earlier copy OUT of this pinned staging is done
            with torch.cuda.stream(self.copy):
                if self.free[k] is not None:
                    self.copy.wait_event(self.free[k])         # consumer kernels of the old occupant are done
                for p, h, d in zip(pin, host, dev):
                    p.copy_(torch.from_numpy(h))
                    d.copy_(p, non_blocking=True)
                self.ready[k].record(self.copy)
            self.queue.append(k)

    def __iter__(self):
        return self

    def __next__(self):
        if not self.queue:
            raise StopIteration
        k = self.queue.pop(0)
        torch.cuda.current_stream(self.device).wait_event(self.ready[k])
        self._last = k
        return self.slots[k][1]

    def release(self):
        """Call after the kernels that read the batch returned by the last __next__ have been enqueued."""
        k = self._last
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.free[k] = ev
        self._fill()

    def close(self):
        """Leaving the loop early (--max_batches): wait for the copies still in flight before the slots are dropped."""
        self.copy.synchronize()
        self.queue.clear()

    def __del__(self):
        try:
            self.copy.synchronize()
        except Exception:
            pass


class DeviceMetrics:
    """Accumulates the evaluation figures of :63-93 on the device: one HIP launch per batch
    (`mcrn_eval_metrics`), no `.item()`; `result()` is the single synchronisation of an evaluation."""

    def __init__(self, args, scaler, device):
        from ._lib import lib, check
        import ctypes as C
        self.lib, self.check, self.C = lib, check, C
        self.hz = [h for h in (3, 6, 12) if h <= args.horizon]
        self.harr = (C.c_int * 3)(*(self.hz + [0] * (3 - len(self.hz))))
        self.mean, self.std, self.lamb, self.lamb1 = float(scaler.mean), float(scaler.std), args.lamb, args.lamb1
        self.scratch = torch.zeros(64 + 18 * 1024, device=device)
        self.acc = torch.zeros(17, device=device)

    def reset(self):
        self.acc.zero_()

    def add(self, output, labels, query, pos, neg):
        B, T, N, od = output.shape
        self.check(self.lib.mcrn_eval_metrics(B, T, N, od, query.shape[-1], output.data_ptr(), labels.data_ptr(),
                                              query.data_ptr(), pos.data_ptr(), neg.data_ptr(), self.mean, self.std,
                                              self.lamb, self.lamb1, 1.0, self.harr, len(self.hz),
                                              self.scratch.data_ptr(), self.acc.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream), "mcrn_eval_metrics")

    def result(self):
        a = self.acc.cpu().numpy().astype(np.float64)        # the one sync
        n = max(a[13], 1.0)
        res = {'loss': a[0] / n, 'mae': a[1] / n, 'mape': a[2] / n, 'rmse': float(np.sqrt(a[3] / n))}
        for i, h in enumerate(self.hz):
            res[f'mae_{h}'], res[f'mape_{h}'] = a[4 + 3 * i] / n, a[5 + 3 * i] / n
            res[f'rmse_{h}'] = float(np.sqrt(a[6 + 3 * i] / n))
        return {k: float(v) for k, v in res.items()}


def evaluate(model, loader, scaler, args, device, log=None, mode='val', metrics=None):
    """(:50-99) eval-mode forward over the loader; every figure is accumulated on the device."""
    metrics = metrics or DeviceMetrics(args, scaler, device)
    metrics.reset()
    with torch.no_grad():
        model.eval()
        pf = Prefetcher(loader.get_iterator(), args, device)
        for x, y, ycov in pf:
            output, h_att, query, pos, neg = model(x, ycov)
            metrics.add(output, y, query, pos, neg)
            pf.release()
    res = metrics.result()
    if log and mode == 'test':
        log.info('Horizon overall: mae: {:.4f}, mape: {:.4f}, rmse: {:.4f}'.format(res['mae'], res['mape'], res['rmse']))
        for h in metrics.hz:
            log.info('Horizon {}mins: mae: {:.4f}, mape: {:.4f}, rmse: {:.4f}'.format(
                5 * h, res[f'mae_{h}'], res[f'mape_{h}'], res[f'rmse_{h}']))
    return res


def main(argv=None):
    args = build_parser().parse_args(argv)
    import megacrn_amd
    import torch.distributed as dist
    from megacrn_amd import dp
    from megacrn_amd.trainer import FlatTrainer

    rank, local_rank, world = dp.init_from_env()
    if args.dataset == 'PEMSBAY' and args.synthetic == 0:
        args.num_nodes = 325
    if args.batch_size % world:
        # equal shards: the step averages per-rank gradients with weight 1/world (multi-GPU semantics)
        raise ValueError(f'--batch_size {args.batch_size} must be divisible by the number of ranks ({world})')
    device = torch.device('cuda', local_rank if world > 1 else args.gpu)
    torch.cuda.set_device(device)
    log = logging.getLogger('megacrn_amd.train')
    if not log.handlers:
        logging.basicConfig(level=logging.INFO if rank == 0 else logging.WARNING, format='%(message)s')
    if args.seed is not None:
        np.random.seed(args.seed)
        torch.manual_seed(args.seed)
    elif world > 1:
        dp.seed_curriculum(1234)          # ranks must agree on shuffling and curriculum draws

    data = load_data(args)
    scaler = StandardScaler(mean=data['x_train'][..., 0].mean(), std=data['x_train'][..., 0].std())
    for cat in ('train', 'val', 'test'):
        data['x_' + cat][..., 0] = scaler.transform(data['x_' + cat][..., 0])
        data['y_' + cat][..., 0] = scaler.transform(data['y_' + cat][..., 0])
    loaders = {'train': DataLoader(data['x_train'], data['y_train'], args.batch_size, shuffle=True),
               'val': DataLoader(data['x_val'], data['y_val'], args.batch_size, shuffle=False),
               'test': DataLoader(data['x_test'], data['y_test'], args.batch_size, shuffle=False)}

    def get_model():
        return megacrn_amd.MegaCRN(num_nodes=args.num_nodes, input_dim=args.input_dim, output_dim=args.output_dim,
                                   horizon=args.horizon, rnn_units=args.rnn_units, num_layers=args.num_rnn_layers,
                                   mem_num=args.mem_num, mem_dim=args.mem_dim, cheb_k=args.max_diffusion_step,
                                   cl_decay_steps=args.cl_decay_steps,
                                   use_curriculum_learning=args.use_curriculum_learning).to(device)

    model = get_model()
    tr = FlatTrainer(model, lr=args.lr, eps=args.epsilon, max_grad_norm=args.max_grad_norm, lamb=args.lamb,
                     lamb1=args.lamb1, scaler_mean=float(scaler.mean), scaler_std=float(scaler.std))
    metrics = DeviceMetrics(args, scaler, device)
    # one run directory for the whole job: rank 0 names it, everybody else is told
    stamp = [time.strftime("%Y%m%d%H%M%S")]
    if world > 1:
        dist.broadcast_object_list(stamp, src=0)
    path = os.path.join(args.save_dir, f'{args.dataset}_MegaCRN_{stamp[0]}')
    if rank == 0:
        os.makedirs(path, exist_ok=True)
    modelpt_path = os.path.join(path, 'MegaCRN.pt')
    milestones = sorted(args.steps)
    min_val, wait, history = float('inf'), 0, []
    shard = (lambda n: dp.shard_bounds(n, rank, world)) if world > 1 else None
    for epoch in range(args.epochs):
        t0 = time.time()
        model.train()
        losses = []
        pf = Prefetcher(loaders['train'].get_iterator(), args, device, shard)
        for bi, (xb, yb, ycov) in enumerate(pf):
            losses.append(tr.train_step(xb, ycov, yb))
            pf.release()
            if args.max_batches and bi + 1 >= args.max_batches:
                pf.close()
                break
        tl = torch.stack(losses).mean()                               # one sync per epoch, not per step
        if world > 1:                                                 # the logged loss covers every rank's shard
            dist.all_reduce(tl)
            tl /= world
        train_loss = float(tl.item())
        tr.lr = args.lr * args.lr_decay_ratio ** sum(1 for m in milestones if epoch + 1 >= m)   # MultiStepLR (:132)
        # Every rank evaluates the full val / test sets (replicated weights), but ranks may autotune different GEMM
        # tiles and differ in the last bits: rank 0's figures decide checkpoint and early stop for everybody, so
        # all ranks leave the loop in the same epoch (no rank is left alone inside an all-reduce).
        val = evaluate(model, loaders['val'], scaler, args, device, metrics=metrics)
        test = evaluate(model, loaders['test'], scaler, args, device, log, 'test', metrics=metrics)
        if world > 1:
            dec = torch.tensor([val['loss'], test['mae']], device=device, dtype=torch.float64)
            dist.broadcast(dec, src=0)
            val['loss'], test['mae'] = float(dec[0].item()), float(dec[1].item())
        log.info('Epoch [{}/{}] ({}) train_loss: {:.4f}, val_loss: {:.4f}, lr: {:.6f}, {:.1f}s'.format(
            epoch + 1, args.epochs, tr.batches_seen, train_loss, val['loss'], tr.lr, time.time() - t0))
        history.append((train_loss, val['loss'], test['mae']))
        if val['loss'] < min_val:
            wait, min_val = 0, val['loss']
            if rank == 0:
                torch.save(model.state_dict(), modelpt_path)
        else:
            wait += 1
            if wait == args.patience:
                log.info('Early stopping at epoch: %d' % epoch)
                break
    log.info('=' * 35 + 'Best model performance' + '=' * 35)
    if world > 1:
        dist.barrier()                     # rank 0's last checkpoint write is complete before anyone loads it
    if os.path.exists(modelpt_path):
        best = get_model()
        best.load_state_dict(torch.load(modelpt_path))
        evaluate(best, loaders['test'], scaler, args, device, log, 'test', metrics=metrics)
    return history


if __name__ == '__main__':
    main()
