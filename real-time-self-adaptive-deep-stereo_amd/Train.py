"""Offline supervised training on MI355X -- the flags, loop and outputs of the reference script (Train.py:22-178): batches of
random crops from --trainingSet, multi-scale mean_l1 against the ground truth, Adam(lr, 0.9), a log line every 100 steps,
a checkpoint every 10000 steps (and at the end) as a TensorFlow V2 checkpoint under --output.
The per-step device work is madnet_hip.trainer.Trainer.step (one hipGraph replay of the HIP kernels).

Kept from the reference on purpose: --decayStep is accepted but has no effect (Train.py:94-95 builds the exponentially
decayed rate and then hands args.lr to AdamOptimizer).  Not carried over: TensorBoard summaries and the colourised image
dumps (Train.py:86-87,105-107,112) -- the loss / EPE / bad3 go to <output>/train_log.csv instead; --validationSet is
evaluated with the current weights every 1000 steps on one batch (the reference only wires it into the summaries).
Multi-GPU: launch with `python -m torch.distributed.run --nproc-per-node N Train.py ...`: one process per GPU, each on its own
shard of the shuffled stream, gradients all-reduced over RCCL every step."""
import argparse
import datetime
import os
import time

import numpy as np

import Nets
from Data_utils import data_reader, tf_checkpoint
from Stereo_Online_Adaptation import load_weights

PIXEL_TH = 3
MAX_DISP = 192


def save_checkpoint(stereo_net, output, step):
    """main_saver.save(sess, <output>/weights.ckpt, global_step) (Train.py:152-154): variables + Adam slots + beta powers."""
    eng = stereo_net.engine
    P = eng.params
    tensors = {name: P.tensor(name).detach().cpu().numpy() for name, _ in P.manifest}
    if getattr(P, 'v', None) is not None:
        tensors.update({name + '/Adam': P.tensor(name, 'm').detach().cpu().numpy() for name, _ in P.manifest})
        tensors.update({name + '/Adam_1': P.tensor(name, 'v').detach().cpu().numpy() for name, _ in P.manifest})
        st = eng.adam_state.detach().cpu().numpy()
        tensors['training_error/beta1_power'] = np.asarray(st[0], np.float32)
        tensors['training_error/beta2_power'] = np.asarray(st[1], np.float32)
    tensors['training_error/Variable'] = np.asarray(step, np.int32)
    path = os.path.join(output, 'weights.ckpt-%d' % step)
    tf_checkpoint.write_checkpoint(path, tensors)
    return path


def main(args):
    import torch
    from madnet_hip.trainer import Trainer
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    dev = 'cuda:%d' % local
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl')
    H, W = args.imageShape
    n_pred = 6 if args.modelName == 'MADNet' else 7               # predicted scales: MadNet.py:268-364 / DispNet.py:75-152
    if args.lossWeights is not None and len(args.lossWeights) < n_pred:
        raise SystemExit('--lossWeights needs %d values for %s (one per predicted scale, full resolution first); got %d'
                         % (n_pred, args.modelName, len(args.lossWeights)))
    data_set = data_reader.dataset(args.trainingSet, batch_size=args.batchSize, crop_shape=args.imageShape,
                                   num_epochs=args.numEpochs, augment=args.augment, is_training=True, shuffle=True, seed=rank,
                                   shard=(rank, world))
    validation_set = None
    if args.validationSet is not None:
        validation_set = data_reader.dataset(args.validationSet, batch_size=args.batchSize, crop_shape=args.imageShape,
                                             augment=False, is_training=False, shuffle=True, seed=1000 + rank, num_epochs=10 ** 6)
    B = args.batchSize
    net_args = {'left_img': torch.zeros(B, H, W, 3, device=dev), 'right_img': torch.zeros(B, H, W, 3, device=dev),
                'split_layers': [None], 'sequence': True, 'train_portion': 'BEGIN', 'bulkhead': False,
                'weights': load_weights(args.weights or 'xavier:0', args.modelName), 'precision': args.precision}
    stereo_net = Nets.get_stereo_net(args.modelName, net_args)
    print('Stereo Prediction Model:\n', stereo_net)
    trainer = Trainer(stereo_net, lr=args.lr, loss_weights=args.lossWeights, loss_type=args.lossType, max_disp=MAX_DISP,
                      data_parallel=world > 1)
    max_steps = data_set.get_max_steps()
    log = open(os.path.join(args.output, 'train_log.csv'), 'w') if rank == 0 else None
    if log:
        log.write('step,loss,EPE,bad3,val_EPE,val_bad3\n')
    val_iter = iter(validation_set) if validation_set is not None else None
    exec_time, step_eval = 0.0, 0
    start_time = time.time()
    try:
        frames = data_reader.device_prefetcher(data_set, dev, depth=3, consumer_stream=trainer.stream)
        for left, right, gt in frames:
            out = trainer.step(left, right, gt[..., 0])
            if step_eval % 100 == 0 and rank == 0:
                fbTime = (time.time() - start_time)
                exec_time += fbTime
                fbTime = fbTime / 100
                missing_time = (max_steps - step_eval) * fbTime
                print('Step:{:4d}\tLoss:{:.2f}\tf/b time:{:3f}\tMissing time:{}'.format(
                    step_eval, out['loss'], fbTime, datetime.timedelta(seconds=missing_time)))
                val = ('', '')
                if val_iter is not None and step_eval % 1000 == 0:
                    vl, vr, vg = next(val_iter)
                    val = validate(stereo_net, vl, vr, vg)
                log.write('%d,%.6f,%.4f,%.5f,%s,%s\n' % (step_eval, out['loss'], out['epe'], out['bad3'], val[0], val[1]))
                log.flush()
                start_time = time.time()
            if step_eval % 10000 == 0 and rank == 0:
                save_checkpoint(stereo_net, args.output, step_eval)
            step_eval = out['global_step']
    finally:
        if rank == 0:
            print('checkpoint:', save_checkpoint(stereo_net, args.output, step_eval))
            log.close()
        print('All Done, Bye Bye!')


def validate(stereo_net, left, right, gt):
    """EPE / bad3 of the current weights on one validation batch (Train.py:75-88), inference plan only."""
    import torch
    eng = stereo_net.engine
    if not hasattr(eng, '_val_plan'):
        eng._val_plan = eng.build_plan('NONE')
    eng.set_inputs(left, right, np.asarray(gt)[..., 0])
    eng._val_plan.run(stereo_net._lib, 0)
    torch.cuda.synchronize()
    m = eng.res_met.cpu().numpy()
    return '%.4f' % m[0], '%.5f' % m[1]


def build_parser():
    parser = argparse.ArgumentParser(description='Supervised training of a deep stereo network on the MI355X engine')
    parser.add_argument("--trainingSet", help='list file (left,right,gt per row) of the training set', required=True)
    parser.add_argument("--validationSet", help="list file of the validation set", default=None, type=str)
    parser.add_argument("-o", "--output", help="folder for the checkpoints and the training log", required=True)
    parser.add_argument("--weights", help="initial weights: a TF checkpoint prefix / folder (optional)")
    parser.add_argument("--modelName", help="stereo model", default="Dispnet", choices=Nets.STEREO_FACTORY.keys())
    parser.add_argument("--lr", help="learning rate of Adam", default=0.0001, type=float)
    parser.add_argument("--imageShape", help='height and width of the random crop', nargs='+', type=int, default=[320, 1216])
    parser.add_argument("--batchSize", help='samples per step (per GPU)', type=int, default=4)
    parser.add_argument("--numEpochs", help='passes over the training list', type=int, default=50)
    parser.add_argument("--augment", help="colour augmentation of the two views", action='store_true')
    parser.add_argument("--lossWeights", help="weight of the loss at each predicted scale, full resolution first", nargs='+', default=None, type=float)
    parser.add_argument('--lossType', help="supervised loss", choices=['mean_l1'], default="mean_l1", type=str)
    parser.add_argument("--decayStep", help="accepted for compatibility: the reference never applies its decayed rate", type=int, default=500000)
    parser.add_argument("--precision", help="MFMA arithmetic of the conv kernels: fp32 (default; the reference trains in fp32), mixed (forward within fp32 tolerance, bf16 gradients) or bf16 (opt-in throughput mode)", choices=['fp32', 'mixed', 'bf16'], default='fp32')
    return parser


if __name__ == '__main__':
    args = build_parser().parse_args()
    if not os.path.exists(args.output):
        os.makedirs(args.output)
    main(args)
