#!/usr/bin/env python3
"""Counterpart of the reference's tools/train.py (same CLI, YAML schema and work-dir layout) on the
MI355X hot path.

    python tools/train.py <config.yml> [--resume_from weights.npz] [--synthetic N_CLASSES]

Triplet mode runs the fused step (one forward, on-GPU distance matrix + mining + hinge, backward,
optimizer); siamese mode trains SiameseNet with contrastive_loss.  Per-epoch schedule as the reference:
lr0 * decay^floor(epoch/step) (train.py:80-81), ReduceLROnPlateau(0.1, patience 4) (:82-83),
EarlyStopping(patience 10) (:84-86), best-only checkpoints weights/epoch_XXX.npz (:87-90).
Multi-GPU: launch with torchrun; classes are sharded per rank and gradients all-reduced over RCCL.
`--synthetic` swaps the file loader for an in-memory synthetic dataset (no dataset ships with the repo).
"""
import argparse
import os
import sys

BASE_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT_DIR = os.path.dirname(BASE_DIR)
sys.path.insert(0, ROOT_DIR)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from embedding_net.models import TripletNet, SiameseNet  # noqa: E402
from embedding_net.utils import parse_params  # noqa: E402
from embedding_net.losses_and_accuracies import contrastive_loss, triplet_loss, accuracy  # noqa: E402
from embeddingnet_amd.datagenerators import (ENDataLoader, SyntheticDataLoader, TripletsDataGenerator,  # noqa: E402
                                             SimpleTripletsDataGenerator, SiameseDataGenerator)
from embeddingnet_amd.parallel import (GradReducer, all_reduce_mean, average_buffers, broadcast_model,  # noqa: E402
                                       init_distributed, shard_classes)
from embeddingnet_amd.train_step import TripletTrainer  # noqa: E402


def parse_args():
    parser = argparse.ArgumentParser(description='Train an embedding network')
    parser.add_argument('config', help='model config file path')
    parser.add_argument('--resume_from', help='the checkpoint file to resume from')
    parser.add_argument('--synthetic', type=int, default=0, help='use N synthetic classes instead of DATALOADER')
    parser.add_argument('--max_epochs', type=int, default=None, help='cap TRAIN.n_epochs (smoke runs)')
    return parser.parse_args()


def create_save_folders(params):
    work_dir_path = os.path.join(params['work_dir'], params['project_name'])
    paths = {k: os.path.join(work_dir_path, v) for k, v in
             dict(weights='weights/', pretrained='pretraining_model/weights/', encodings='encodings/', plots='plots/',
                  tf_log='tf_log/', pretrained_log='pretraining_model/tf_log/').items()}
    for p in [work_dir_path] + list(paths.values()):
        os.makedirs(p, exist_ok=True)
    return paths


def _optimizer_state_path(weights_path):
    """<project>/weights/epoch_007.npz -> <project>/optimizer/epoch_007.npz (the weights folder holds weight files only)."""
    d, f = os.path.split(os.path.abspath(weights_path))
    return os.path.join(os.path.dirname(d), 'optimizer', f)


class Plateau:
    """The three monitor-driven Keras callbacks of the reference (train.py:82-90), each with its own state as in Keras:
      ReduceLROnPlateau(factor .1, patience 4): `min_delta` 1e-4 (Keras default, mode 'min': an epoch improves only if
        value < best - 1e-4), cooldown 0, min_lr 0;  after 4 epochs without that improvement the learning rate is
        multiplied by 0.1 and the wait restarts;
      EarlyStopping(patience 10, min_delta 0): stop after 10 epochs without value < best;
      ModelCheckpoint(save_best_only): save when value < best.
    update(value, lr) -> (save checkpoint?, stop?, lr after the epoch).
    NB the reference ALSO installs LearningRateScheduler(lambda epoch: lr0 * decay ** floor(epoch / step)) — a one-argument
    schedule, which Keras applies at every epoch BEGIN by setting the optimizer's lr outright.  It therefore overwrites
    whatever ReduceLROnPlateau set at the previous epoch end: in the reference the plateau reduction never reaches a
    training step (it only shows in the log).  main() reproduces that order; `persistent=True` is this port's opt-in
    (TRAIN.plateau_persistent) to let reductions accumulate as a multiplier on the schedule instead."""

    def __init__(self, factor=0.1, patience=4, min_delta=1e-4, stop_patience=10, persistent=False):
        self.factor, self.patience, self.min_delta, self.stop_patience = factor, patience, min_delta, stop_patience
        self.persistent, self.scale = persistent, 1.0
        self.rl_best, self.rl_wait = float('inf'), 0          # ReduceLROnPlateau
        self.es_best, self.es_wait = float('inf'), 0          # EarlyStopping
        self.best = float('inf')                              # ModelCheckpoint

    def update(self, value, lr=None):
        if value < self.rl_best - self.min_delta:
            self.rl_best, self.rl_wait = value, 0
        else:
            self.rl_wait += 1
            if self.rl_wait >= self.patience:
                self.rl_wait = 0
                self.scale *= self.factor
                if lr is not None:
                    lr = lr * self.factor
                    print(f'ReduceLROnPlateau reducing learning rate to {lr:g}.', flush=True)
        if value < self.es_best:
            self.es_best, self.es_wait = value, 0
        else:
            self.es_wait += 1
        save = value < self.best
        if save:
            self.best = value
        return save, self.es_wait >= self.stop_patience, lr


def apply_gpu_ids(gpu_ids):
    """GENERAL.gpu_ids (reference train.py:121-133: CUDA_VISIBLE_DEVICES + n_gpu): the listed devices become the
    visible ones; with more than one id and no launcher around us, re-run this script as one process per listed GPU
    under torch.distributed.run (the data-parallel world) and exit with its status.  Must run before the GPU is touched."""
    if not gpu_ids:
        return
    ids = [s.strip() for s in str(gpu_ids).split(',') if s.strip()]
    if 'WORLD_SIZE' in os.environ:                    # already under torchrun: the launcher decided the world
        if int(os.environ['WORLD_SIZE']) != len(ids):
            print(f"GENERAL.gpu_ids lists {len(ids)} devices but WORLD_SIZE={os.environ['WORLD_SIZE']}: using the launcher's world")
        return
    os.environ.setdefault('HIP_VISIBLE_DEVICES', ','.join(ids))
    print(f'Using gpu ids: {",".join(ids)}')
    if len(ids) > 1:
        # one process per listed GPU, started and WATCHED by embeddingnet_amd/launch.py: a rank that dies ends the world at
        # once with its exit code instead of leaving the others in a collective until the process-group timeout
        from embeddingnet_amd import launch
        sys.exit(launch.spawn(len(ids), [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))


def main():
    args = parse_args()
    cfg = parse_params(args.config)
    p_train, p_model, p_loader, p_gen = cfg['train'], cfg['model'], cfg['dataloader'], cfg['generator']
    apply_gpu_ids(cfg['general'].get('gpu_ids'))
    paths = create_save_folders(cfg['general'])
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:        # before the first GPU call: stay on the cores next to this rank's GPU
        from embeddingnet_amd import launch
        print(f"[rank {os.environ.get('RANK', '0')}] {launch.pin_to_gpu_numa(int(os.environ.get('LOCAL_RANK', '0')))}", flush=True)
    rank, world, local = init_distributed()
    dev = torch.device('cuda', local % max(torch.cuda.device_count(), 1))    # (ranks > GPUs only in the gloo debug mode)
    torch.cuda.set_device(dev)
    p_model['device'] = dev
    # every rank builds the same dataset and the same initial model; what differs per rank is what it samples
    import random
    random.seed(1234 + rank)
    np.random.seed(1234 + rank)

    if args.synthetic:
        data_loader = SyntheticDataLoader(args.synthetic, 24, p_model['input_shape'],
                                          validate=p_loader.get('validate', True), seed=0)
    else:
        data_loader = ENDataLoader(**{k: v for k, v in p_loader.items() if k != 'csv_file'})
    monitor = 'val_loss' if data_loader.validate else 'loss'
    gen_kw = {k: v for k, v in p_gen.items()}

    siamese = p_model['mode'] == 'siamese'
    if siamese:
        model = SiameseNet(cfg, training=True)
        train_gen = SiameseDataGenerator(class_files_paths=data_loader.train_data, class_names=data_loader.class_names,
                                         **gen_kw)
        val_gen = SiameseDataGenerator(class_files_paths=data_loader.val_data, class_names=data_loader.class_names,
                                       val_gen=True, **gen_kw) if data_loader.validate else None
        trainable = model.model
    else:
        model = TripletNet(cfg, training=True)
        if world > 1:                                 # whole classes per rank, mining stays local
            _, gen_kw['k_classes'] = shard_classes(p_gen['k_classes'], world, rank)
        train_gen = TripletsDataGenerator(embedding_model=model.base_model, class_files_paths=data_loader.train_data,
                                          class_names=data_loader.class_names, **gen_kw)
        val_gen = SimpleTripletsDataGenerator(data_loader.val_data, data_loader.class_names,
                                              **gen_kw) if data_loader.validate else None
        trainable = model.base_model
    if args.resume_from is not None:
        model.load_model(args.resume_from)            # the mining model IS base_model, so it resumes too
    broadcast_model(trainable)                        # identical start on every rank: parameters and BN buffers
    if 'softmax' in cfg:                              # reference train.py:164-170
        # every rank pre-trains on its own batches with the gradients all-reduced (no rank waits in a collective for the
        # length of a pre-training: the RCCL watchdog would abort it); rank 0 writes the pre-training checkpoints
        from embedding_net.backbones import pretrain_backbone_softmax
        pretrain_backbone_softmax(model.backbone_model, data_loader, cfg['softmax'], cfg['general'],
                                  max_epochs=args.max_epochs, distributed=world > 1)
        broadcast_model(trainable)                    # belt and braces: BN moving statistics are rank-local

    params = [p for p in trainable.parameters() if p.requires_grad]
    opt = p_train['optimizer'].build(params)
    lr0 = p_train['learning_rate']
    if args.resume_from is not None:                  # optimizer slots and step count, when the checkpoint has them (Keras'
        # load_model restores the optimizer with the weights; the epoch count and LR schedule restart, as in the reference)
        from embeddingnet_amd.backbones import keras_weights
        from embeddingnet_amd.optimizers import load_optimizer_state
        opt_path = _optimizer_state_path(args.resume_from)
        if os.path.exists(opt_path):
            extra = load_optimizer_state(opt_path, opt, {k: v for k, v in keras_weights(trainable).items()
                                                         if isinstance(v, torch.nn.Parameter)})
            print(f'resumed optimizer state from {opt_path} (iterations {opt.iterations}, saved after epoch {int(extra.get("epoch", 0))})')
    reducer = GradReducer(params) if world > 1 else None
    if siamese:
        # the Siamese step (reference train.py:108-119: fit of SiameseNet.model with contrastive_loss) as a trainer: its own step
        # context, one-launch optimizer, and a captured step where the host cannot keep up (EMBNET_GRAPH=auto)
        from embeddingnet_amd.train_step import SiameseTrainer
        trainer = SiameseTrainer(model.model, opt, contrastive_loss, seed=rank, reducer=reducer)
    else:
        trainer = TripletTrainer(
            model.base_model, opt, gen_kw['k_classes'], p_gen['k_samples'], margin=p_gen['margin'],
            negatives_selection_mode=p_gen['negatives_selection_mode'], seed=rank, reducer=reducer)
    plateau = Plateau(persistent=bool(p_train.get('plateau_persistent', False)))
    history = {'loss': [], 'val_loss': []}
    n_epochs = min(p_train['n_epochs'], args.max_epochs or p_train['n_epochs'])

    # triplet mode: batches are planned on this thread and decoded / uploaded ahead of the step (input_pipeline.Feeder: the
    # role of Keras' fit_generator enqueuer, reference train.py:172-177)
    feeder = None if siamese else train_gen.feeder(dev, log=(print if rank == 0 else None))
    import time
    try:        # (the feeder owns worker processes and a staging file in /dev/shm: released on ANY exit)
        for epoch in range(n_epochs):
            t_epoch, n_images = time.perf_counter(), 0
            # LearningRateScheduler.on_epoch_begin (reference train.py:80-81): sets the rate outright, which discards the
            # previous epoch's ReduceLROnPlateau reduction (see Plateau) unless TRAIN.plateau_persistent is set
            lr = lr0 * p_train['decay_factor'] ** np.floor(epoch / p_train['step_size'])
            if plateau.persistent:
                lr *= plateau.scale
            for g in opt.param_groups:
                g['lr'] = lr
            trainable.train()
            losses = []
            for _ in range(len(train_gen)):
                if siamese:
                    (x1, x2), y = train_gen[0]
                    losses.append(trainer.step(torch.from_numpy(x1).to(dev), torch.from_numpy(x2).to(dev),
                                               torch.from_numpy(np.asarray(y, dtype=np.float32)).to(dev).reshape(-1, 1)))
                else:
                    xb = feeder.next()
                    n_images += xb.shape[0]
                    losses.append(trainer.step(xb))
            epoch_loss = all_reduce_mean(float(torch.stack(losses).mean().item()))     # mean over ranks (logging + monitor)
            history['loss'].append(epoch_loss)
            msg = f'Epoch {epoch + 1}/{n_epochs} - lr {lr:.3g} - loss {epoch_loss:.4f}'
            if n_images:                                  # (the .item() above waited for the epoch's last step)
                msg += f' - {n_images * world / (time.perf_counter() - t_epoch):.0f} images/s'
            value = epoch_loss
            if val_gen is not None:
                trainable.eval()
                vals = []
                with torch.no_grad():
                    for _ in range(len(val_gen)):
                        xs, y = val_gen[0]
                        xs = [torch.from_numpy(a).to(dev) for a in xs]
                        if siamese:
                            out = model.model(xs)[0]
                            vals.append(contrastive_loss(torch.from_numpy(y).to(dev), out))
                        else:
                            vals.append(triplet_loss(p_gen['margin'])(None, model.model(xs)).mean())
                value = all_reduce_mean(float(torch.stack(vals).mean().item()))
                history['val_loss'].append(value)
                msg += f' - val_loss {value:.4f}'
            if rank == 0:
                print(msg, flush=True)
            improved, stop, lr_end = plateau.update(value, lr)   # `value` is the all-reduced mean: same decisions on every rank
            for g in opt.param_groups:                    # ReduceLROnPlateau.on_epoch_end (until the scheduler's next epoch begin)
                g['lr'] = lr_end
            if improved:
                average_buffers(trainable)                # BN moving statistics: mean over the ranks' local batches
            if improved and rank == 0:
                path = os.path.join(paths['weights'], f'epoch_{epoch + 1:03d}.npz')
                model.save_weights(path)
                from embeddingnet_amd.backbones import keras_weights
                from embeddingnet_amd.optimizers import save_optimizer_state
                os.makedirs(os.path.dirname(_optimizer_state_path(path)), exist_ok=True)
                save_optimizer_state(_optimizer_state_path(path), opt,
                                     {k: v for k, v in keras_weights(trainable).items() if isinstance(v, torch.nn.Parameter)},
                                     extra={'epoch': epoch + 1})
                print(f'{monitor} improved to {value:.5f}, saving model to {path}')
            if stop:
                print('EarlyStopping')
                break
    finally:
        if feeder is not None:
            feeder.close()                            # decode worker processes, staging file in /dev/shm
    if rank == 0:
        np.savez(os.path.join(paths['plots'], 'history.npz'), **{k: np.asarray(v) for k, v in history.items()})
    dump = os.environ.get('EMBNET_DUMP_FINAL_WEIGHTS')     # diagnostics: EVERY rank's final weights -> <prefix><rank>.npz
    if dump:                                               # (data-parallel ranks must end with identical trainable weights)
        model.save_weights(f'{dump}{rank}.npz')
    return history


if __name__ == '__main__':
    main()
