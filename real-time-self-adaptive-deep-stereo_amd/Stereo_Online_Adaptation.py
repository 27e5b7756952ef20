"""Online adaptation driver for MI355X -- same flags, loop semantics and output files (stats.csv,
series.csv, params.sh, config.json, disparities/*.png) as the reference script
(Stereo_Online_Adaptation.py:30-325); the per-frame loop body lives in madnet_hip.adapter.Adapter.step().

--weights accepts a TensorFlow V2 checkpoint prefix (`<prefix>.index` + `.data-*`, read without TensorFlow by
Data_utils/tf_checkpoint.py -- what the reference restores with weights_utils, :150-153), an .npz of
{TF variable name: HWIO array}, or the literal `xavier[:seed]` / `calibrated[:seed]` for synthetic weights."""
import argparse
import datetime
import json
import os
import shutil
import sys
import time

import numpy as np

import Nets
from Data_utils import data_reader
from Sampler import sampler_factory

MAX_DISP = 256
PIXEL_TH = 3


def load_weights(spec, model_name="MADNet", radius_d=2, stride=1, allow_missing=False):
    """allow_missing: a checkpoint that lacks some of the model's variables is an ERROR unless this is set (--allowMissingWeights);
    then -- like the reference, whose Saver restores the matching names and leaves the others at their initializer
    (Stereo_Online_Adaptation.py:150-153, weights_utils.py:4-37) -- the missing ones keep Xavier values."""
    from madnet_hip import engine as E, dispnet_engine as DE, synthetic
    shapes = dict(E.madnet_manifest(radius_d, stride) if model_name == "MADNet" else DE.dispnet_manifest())
    kind = spec.split(':')[0]
    if kind in ('xavier', 'calibrated'):
        seed = int(spec.split(':')[1]) if ':' in spec else (0 if kind == 'xavier' else 1)
        return synthetic.xavier_weights(shapes, seed) if kind == 'xavier' else synthetic.calibrated_weights(shapes, seed)
    if spec.endswith('.npz'):
        z = np.load(spec)
        w = {k: z[k] for k in z.files}
        assert len(w) > 0                      # Stereo_Online_Adaptation.py:151
        return w
    from Data_utils import tf_checkpoint
    if os.path.isdir(spec):
        spec = tf_checkpoint.latest_checkpoint(spec) or spec
    if tf_checkpoint.is_checkpoint(spec):
        # Stereo_Online_Adaptation.py:150-153: restore every checkpoint variable whose name matches a graph variable
        reader = tf_checkpoint.CheckpointReader(spec)
        have = reader.get_variable_to_shape_map()
        w = {k: reader.get_tensor(k).astype(np.float32) for k in shapes if k in have}
        assert len(w) > 0, "no variable of %s found in checkpoint %s" % (model_name, spec)
        missing = [k for k in shapes if k not in have]
        if missing and not allow_missing:
            raise Exception('checkpoint %s lacks %d of the %d variables of %s (first: %s); pass --allowMissingWeights to keep their '
                            'Xavier initial values like the reference does' % (spec, len(missing), len(shapes), model_name, missing[0]))
        if missing:
            print('WARNING: %d variables not in the checkpoint keep their synthetic initial value (first: %s)' % (len(missing), missing[0]))
            w0 = synthetic.xavier_weights(shapes, 0)
            for k in missing:
                w[k] = w0[k]
        print('Disparity Net Restored?: {}, number of restored variables: {}'.format(True, len(w) - len(missing)))
        return w
    raise Exception('Unsupported --weights %r: expected a TF checkpoint prefix, an .npz of TF-named variables, '
                    'xavier[:seed] or calibrated[:seed]' % spec)


def main(args):
    import torch
    from madnet_hip.adapter import Adapter
    with open(args.blockConfig) as json_data:
        train_config = json.load(json_data)
    data_set = data_reader.dataset(args.list, batch_size=1, crop_shape=args.imageShape, num_epochs=1,
                                   augment=False, is_training=False, shuffle=False, keep_uint8=True)   # 8-bit frames cross PCIe as bytes
    H, W = args.imageShape
    dev = 'cuda'
    left_img_batch = torch.zeros(1, H, W, 3, device=dev)
    right_img_batch = torch.zeros(1, H, W, 3, device=dev)
    net_args = {'left_img': left_img_batch, 'right_img': right_img_batch, 'split_layers': [None], 'sequence': True,
                'train_portion': 'BEGIN', 'bulkhead': True if args.mode == 'MAD' else False,
                'weights': load_weights(args.weights, args.modelName, allow_missing=getattr(args, 'allowMissingWeights', False))}
    stereo_net = Nets.get_stereo_net(args.modelName, net_args)
    print('Stereo Prediction Model:\n', stereo_net)
    predictions = stereo_net.get_disparities()
    adapter = Adapter(stereo_net, mode=args.mode, block_config=train_config, lr=args.lr, sample_mode=args.sampleMode,
                      num_blocks=args.numBlocks, fixed_id=args.fixedID, sample_frequency=args.sampleFrequency,
                      ssim_th=args.SSIMTh, reprojection_scale=args.reprojectionScale)
    print('Disparity Net Restored?: {}, number of restored variables: {}'.format(True, len(stereo_net.engine.params.manifest)))

    epe_accumulator, bad3_accumulator = [], []
    exec_time = 0
    step = 0
    max_steps = data_set.get_max_steps()
    start_time = time.time()
    t_begin = time.time()
    try:
        # decode + upload frame t+1 while frame t adapts (pinned ring + copy stream, Data_utils/data_reader.py)
        frames = data_reader.device_prefetcher(data_set, dev, depth=3, consumer_stream=adapter.stream, cast=False)
        for left, right, gt in frames:
            out = adapter.step(left, right, gt[..., 0])
            new_loss = out['loss']
            epe_accumulator.append(out['epe'])
            bad3_accumulator.append(out['bad3'])
            if step % 100 == 0:
                fbTime = (time.time() - start_time)
                exec_time += fbTime
                fbTime = fbTime / 100
                missing_time = (max_steps - step) * fbTime
                print('Step:{:4d}\tbad3:{:.2f}\tEPE:{:.2f}\tSSIM:{:.2f}\tf/b time:{:3f}\tMissing time:{}'.format(
                    step, out['bad3'], out['epe'], new_loss, fbTime, datetime.timedelta(seconds=missing_time)))
                start_time = time.time()
            if args.logDispStep != -1 and step % args.logDispStep == 0:
                from PIL import Image
                dispy = out['disparity'][0].detach().cpu().numpy()
                dispy_to_save = (np.clip(dispy, 0, MAX_DISP) * 256.0).astype(np.uint16)
                Image.fromarray(dispy_to_save).save(os.path.join(args.output, 'disparities/disparity_{}.png'.format(step)))
            step += 1
    finally:
        wall = time.time() - t_begin
        epe_array, bad3_array = epe_accumulator, bad3_accumulator
        epe_sum, bad3_sum = np.sum(epe_accumulator), np.sum(bad3_accumulator)
        nstep = max(step, 1)
        exec_time = max(exec_time, 1e-9)
        with open(os.path.join(args.output, 'stats.csv'), 'w+') as f_out:
            f_out.write('Metrics,cumulative,average\n')
            f_out.write('EPE,{},{}\n'.format(epe_sum, epe_sum / nstep))
            f_out.write('bad3,{},{}\n'.format(bad3_sum, bad3_sum / nstep))
            f_out.write('time,{},{}\n'.format(exec_time, exec_time / nstep))
            f_out.write('FPS,{}\n'.format(1 / (exec_time / nstep)))
            f_out.write('#resets,{}\n'.format(adapter.reset_counter))
            f_out.write('Blocks')
            # the reference truncates `predictions` (drops the full-resolution one) only in MAD mode (:88) -> 5 columns for
            # MADNet MAD, 6 for FULL / NONE (Stereo_Online_Adaptation.py:271-274)
            for n in range(len(predictions) - (1 if args.mode == 'MAD' else 0)):
                f_out.write(',{}'.format(n))
            f_out.write(',final\n')
            f_out.write('fetch_counter')
            for c in adapter.fetch_counter:
                f_out.write(',{}'.format(c))
            f_out.write('\n')
            for c in adapter.sample_distribution:
                f_out.write(',{}'.format(c))
            f_out.write('\n')
        # stats.csv stays byte-compatible with the reference's report (positional parsers); the true wall-clock rate
        # (the reference's FPS drops the tail steps from the time but not from the count) goes to its own file
        with open(os.path.join(args.output, 'wall_clock.csv'), 'w+') as f_out:
            f_out.write('steps,wall_seconds,wall_FPS\n{},{},{}\n'.format(nstep, wall, nstep / max(wall, 1e-9)))
        step_time = exec_time / nstep
        with open(os.path.join(args.output, 'series.csv'), 'w+') as f_out:
            f_out.write('Iteration,Time,EPE,bad3\n')
            for i, (e, b) in enumerate(zip(epe_array, bad3_array)):
                f_out.write('{},{},{},{}\n'.format(i, str(i * step_time), e, b))
        print('Result saved in {}'.format(args.output))
        print('All Done, Bye Bye!')


def build_parser():
    parser = argparse.ArgumentParser(description='Online adaptation of a deep stereo network on the MI355X engine')
    parser.add_argument("-l", "--list", help="CSV list of the frames to process (left,right,gt[,proxy] per row)", required=True)
    parser.add_argument("-o", "--output", help="folder that receives the reports (created if missing)", required=True)
    parser.add_argument("--weights", help="initial weights: TF checkpoint prefix, .npz of TF-named variables, xavier[:seed] or calibrated[:seed]", required=True)
    parser.add_argument("--modelName", help="which registered stereo network to build", default="Dispnet", choices=Nets.STEREO_FACTORY.keys())
    parser.add_argument("--numBlocks", help="how many network portions are trained per frame (MAD)", type=int, default=1)
    parser.add_argument("--lr", help="SGD-with-momentum learning rate", default=0.0001, type=float)
    parser.add_argument("--blockConfig", help="json file listing the layers of every trainable portion", required=True)
    parser.add_argument("--sampleMode", help="strategy that picks the portions to train", choices=sampler_factory.AVAILABLE_SAMPLER, default='SAMPLE')
    parser.add_argument("--fixedID", help="portion indices for --sampleMode FIXED", type=int, nargs='+', default=[0])
    parser.add_argument("--reprojectionScale", help="losses at 1/scale resolution (only 1 is supported here)", default=1, type=int)
    parser.add_argument("--summary", help="accepted for compatibility; no TensorBoard summaries are written", action='store_true')
    parser.add_argument("--imageShape", help="height width every frame is centre-cropped / zero-padded to", nargs='+', type=int, default=[320, 1216])
    parser.add_argument("--SSIMTh", help="restore the initial weights when the loss exceeds this value", type=float, default=0.5)
    parser.add_argument("--sampleFrequency", help="draw new portions every K frames", type=int, default=1)
    parser.add_argument("--mode", help="NONE = inference only, FULL = full back-propagation, MAD = modular adaptation", choices=['NONE', 'FULL', 'MAD'], default='MAD')
    parser.add_argument("--allowMissingWeights", help="variables absent from the checkpoint keep Xavier values (the reference's silent behaviour) instead of raising", action='store_true')
    parser.add_argument("--logDispStep", help="dump the disparity every K frames (-1: never)", default=-1, type=int)
    return parser


if __name__ == '__main__':
    args = build_parser().parse_args()
    if not os.path.exists(args.output):
        os.makedirs(args.output)
    if args.logDispStep != -1 and not os.path.exists(os.path.join(args.output, 'disparities')):
        os.makedirs(os.path.join(args.output, 'disparities'))
    shutil.copy(args.blockConfig, os.path.join(args.output, 'config.json'))
    with open(os.path.join(args.output, 'params.sh'), 'w+') as out:
        sys.argv[0] = os.path.join(os.getcwd(), sys.argv[0])
        out.write('#!/bin/bash\n')
        out.write('python3 ')
        out.write(' '.join(sys.argv))
        out.write('\n')
    main(args)
