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
takes its block of each batch, one all-reduce per step.  ``--synthetic`` generates METR-LA-shaped windows when the
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
# device-side metrics -- model/utils.py:126-160
# ---------------------------------------------------------------------------------------------
def _masked(y_pred, y_true, fn):
    mask = (y_true != 0).float()
    mask = mask / mask.mean()
    loss = fn(y_pred, y_true) * mask
    loss = torch.where(torch.isnan(loss), torch.zeros_like(loss), loss)
    return loss.mean()


def masked_mae_loss(p, t):
    return _masked(p, t, lambda a, b: torch.abs(a - b))


def masked_mape_loss(p, t):
    return _masked(p, t, lambda a, b: torch.abs((b - a) / b))


def masked_mse_loss(p, t):
    return _masked(p, t, lambda a, b: (b - a) ** 2)


# ---------------------------------------------------------------------------------------------
def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--dataset', type=str, choices=['METRLA', 'PEMSBAY'], default='METRLA')
    p.add_argument('--data_dir', type=str, default=None, help='directory with train/val/test.npz (default ../<dataset>)')
    p.add_argument('--synthetic', type=int, default=0, help='>0: generate this many synthetic training windows')
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


def prepare_x_y(x, y, args, device, lo=None, hi=None):
    """(:33-48) channel split + float32 + H2D; optional [lo,hi) = this rank's shard of the batch."""
    if lo is not None:
        x, y = x[lo:hi], y[lo:hi]
    x0 = torch.from_numpy(np.ascontiguousarray(x[..., :args.input_dim])).float()
    y0 = torch.from_numpy(np.ascontiguousarray(y[..., :args.output_dim])).float()
    y1 = torch.from_numpy(np.ascontiguousarray(y[..., args.output_dim:])).float()
    return x0.to(device), y0.to(device), y1.to(device)


def evaluate(model, loader, scaler, args, device, log=None, mode='val'):
    horizons = [h for h in (3, 6, 12) if h <= args.horizon]
    with torch.no_grad():
        model.eval()
        losses, maes, mapes, mses = [], [], [], []
        per_h = {h: ([], [], []) for h in horizons}
        triplet, mse = torch.nn.TripletMarginLoss(margin=1.0), torch.nn.MSELoss()
        for x, y in loader.get_iterator():
            x, y, ycov = prepare_x_y(x, y, args, device)
            output, h_att, query, pos, neg = model(x, ycov)
            y_pred, y_true = scaler.inverse_transform(output), scaler.inverse_transform(y)
            loss = masked_mae_loss(y_pred, y_true) + args.lamb * triplet(query, pos, neg) + args.lamb1 * mse(query, pos)
            losses.append(loss.item())
            maes.append(masked_mae_loss(y_pred, y_true).item())
            mapes.append(masked_mape_loss(y_pred, y_true).item())
            mses.append(masked_mse_loss(y_pred, y_true).item())
            for h in horizons:
                p, t = y_pred[:, h - 1:h], y_true[:, h - 1:h]
                per_h[h][0].append(masked_mae_loss(p, t).item())
                per_h[h][1].append(masked_mape_loss(p, t).item())
                per_h[h][2].append(masked_mse_loss(p, t).item())
    res = {'loss': float(np.mean(losses)), 'mae': float(np.mean(maes)), 'mape': float(np.mean(mapes)),
           'rmse': float(np.sqrt(np.mean(mses)))}
    for h in horizons:
        res[f'mae_{h}'], res[f'mape_{h}'] = float(np.mean(per_h[h][0])), float(np.mean(per_h[h][1]))
        res[f'rmse_{h}'] = float(np.sqrt(np.mean(per_h[h][2])))
    if log and mode == 'test':
        log.info('Horizon overall: mae: {:.4f}, mape: {:.4f}, rmse: {:.4f}'.format(res['mae'], res['mape'], res['rmse']))
        for h in horizons:
            log.info('Horizon {}mins: mae: {:.4f}, mape: {:.4f}, rmse: {:.4f}'.format(
                5 * h, res[f'mae_{h}'], res[f'mape_{h}'], res[f'rmse_{h}']))
    return res


def main(argv=None):
    args = build_parser().parse_args(argv)
    import megacrn_amd
    from megacrn_amd import dp
    from megacrn_amd.trainer import FlatTrainer

    rank, local_rank, world = dp.init_from_env()
    if args.dataset == 'PEMSBAY' and args.synthetic == 0:
        args.num_nodes = 325
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
    path = os.path.join(args.save_dir, f'{args.dataset}_MegaCRN_{time.strftime("%Y%m%d%H%M%S")}')
    if rank == 0:
        os.makedirs(path, exist_ok=True)
    modelpt_path = os.path.join(path, 'MegaCRN.pt')
    milestones = sorted(args.steps)
    min_val, wait, history = float('inf'), 0, []
    for epoch in range(args.epochs):
        t0 = time.time()
        model.train()
        losses = []
        for bi, (x, y) in enumerate(loaders['train'].get_iterator()):
            lo, hi = dp.shard_bounds(len(x), rank, world)
            xb, yb, ycov = prepare_x_y(x, y, args, device, lo, hi)
            losses.append(tr.train_step(xb, ycov, yb))
            if args.max_batches and bi + 1 >= args.max_batches:
                break
        train_loss = float(torch.stack(losses).mean().item())        # one sync per epoch, not per step
        tr.lr = args.lr * args.lr_decay_ratio ** sum(1 for m in milestones if epoch + 1 >= m)   # MultiStepLR (:132)
        val = evaluate(model, loaders['val'], scaler, args, device)
        log.info('Epoch [{}/{}] ({}) train_loss: {:.4f}, val_loss: {:.4f}, lr: {:.6f}, {:.1f}s'.format(
            epoch + 1, args.epochs, tr.batches_seen, train_loss, val['loss'], tr.lr, time.time() - t0))
        test = evaluate(model, loaders['test'], scaler, args, device, log, 'test')
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
    if os.path.exists(modelpt_path):
        best = get_model()
        best.load_state_dict(torch.load(modelpt_path))
        evaluate(best, loaders['test'], scaler, args, device, log, 'test')
    return history


if __name__ == '__main__':
    main()
