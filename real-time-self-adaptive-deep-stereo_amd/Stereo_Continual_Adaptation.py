"""Continual online adaptation with proxy labels on MI355X -- same flags, loop semantics and output files (overall.csv,
series.csv, histogram.csv, params.sh, config.json, disparities/*.png, weights/model-<step>) as the reference script
(Stereo_Continual_Adaptation.py:30-345).  Differences of the loop to Stereo_Online_Adaptation: the loss is the
proxy-label mean_l1 (weight 0.01 full / 0.1 per MAD block, :75,112), the weights are only updated every --dilation
frames (:205), the reward update uses --decay / --uf (:218-221), the report is EPE + D1 (:241-249).
The per-frame device work is madnet_hip.adapter.Adapter.step(left, right, gt, proxy)."""
import argparse
import json
import os
import shutil
import sys
import time

import numpy as np

import Nets
from Data_utils import continual_data_reader, data_reader, tf_checkpoint
from Sampler import sampler_factory
from Stereo_Online_Adaptation import load_weights

MAX_DISP = 256
PIXEL_TH = 3


def d1_and_epe(disp, gt):
    """KITTI D1-all and EPE of one frame (Stereo_Continual_Adaptation.py:241-246), on the device tensors."""
    val = gt > 0
    diff = (gt[val] - disp[val]).abs()
    if diff.numel() == 0:
        return float('nan'), float('nan')
    outliers = (diff > 3) & ((diff / gt[val]) >= 0.05)
    return float(outliers.float().mean().item() * 100.), float(diff.mean().item())


def main(args):
    import torch
    from madnet_hip.adapter import Adapter
    with open(args.blockConfig) as json_data:
        train_config = json.load(json_data)
    data_set = continual_data_reader.dataset(args.list, batch_size=1, crop_shape=args.imageShape, num_epochs=1,
                                             augment=False, is_training=False, proxies=True, shuffle=False)
    H, W = args.imageShape
    dev = 'cuda'
    net_args = {'left_img': torch.zeros(1, H, W, 3, device=dev), 'right_img': torch.zeros(1, H, W, 3, device=dev),
                'split_layers': [None], 'sequence': True, 'train_portion': 'BEGIN',
                'bulkhead': True if args.mode == 'MAD' else False, 'weights': load_weights(args.weights, args.modelName)}
    stereo_net = Nets.get_stereo_net(args.modelName, net_args)
    print('Stereo Prediction Model:\n', stereo_net)
    adapter = Adapter(stereo_net, mode=args.mode, block_config=train_config, lr=args.lr, sample_mode=args.sampleMode,
                      num_blocks=args.numBlocks, fixed_id=args.fixedID, sample_frequency=args.sampleFrequency,
                      ssim_th=args.SSIMTh, reprojection_scale=args.reprojectionScale, loss='proxy',
                      dilation=args.dilation, decay=args.decay, uf=args.uf)
    avg_accumulator, d1_accumulator = [], []
    step = 0
    with open(os.path.join(args.output, 'histogram.csv'), 'w') as f_out:
        f_out.write('Histogram\n')
    try:
        frames = data_reader.device_prefetcher(data_set, dev, depth=3, consumer_stream=adapter.stream, cast=False)
        for left, right, gt, proxy, real_width in frames:
            out = adapter.step(left, right, gt[..., 0], proxy=proxy[..., 0])
            d1, epe = d1_and_epe(out['disparity'][0], gt[0, ..., 0])
            d1_accumulator.append(d1)
            avg_accumulator.append(epe)
            if step % 100 == 0:
                with open(os.path.join(args.output, 'histogram.csv'), 'a') as f_out:
                    f_out.write('%s\n' % adapter.fetch_counter)
                print('Step: %04d \tEPE:%.3f\tD1:%.3f\t' % (step, epe, d1))
            if args.logDispStep != -1 and step % args.logDispStep == 0:
                from PIL import Image
                dispy = out['disparity'][0].detach().cpu().numpy()
                dispy_to_save = np.clip(dispy.astype(np.uint16), 0, MAX_DISP)        # (integer disparities * 256, :279-280)
                Image.fromarray((dispy_to_save * 256).astype(np.uint16)).save(
                    os.path.join(args.output, 'disparities/disparity_{}.png'.format(step)))
            step += 1
    finally:
        with open(os.path.join(args.output, 'overall.csv'), 'w+') as f_out:
            print(adapter.fetch_counter)
            f_out.write('EPE\tD1\n')
            f_out.write('%.3f\t%.3f\n' % (np.nanmean(np.asarray(avg_accumulator)) if avg_accumulator else float('nan'),
                                         np.nanmean(np.asarray(d1_accumulator)) if d1_accumulator else float('nan')))
        with open(os.path.join(args.output, 'series.csv'), 'w+') as f_out:
            f_out.write('step\tEPE\tD1\n')
            for i, (a, b) in enumerate(zip(avg_accumulator, d1_accumulator)):
                f_out.write('%d & %.3f & %.3f\n' % (i, a, b))
        if args.saveWeights:
            # adaptation_saver.save(sess, output + '/weights/model', global_step=step) -> a TF V2 checkpoint
            P = stereo_net.engine.params
            tensors = {name: P.tensor(name).detach().cpu().numpy() for name, _ in P.manifest}
            tensors.update({name + '/Momentum': P.tensor(name, 'm').detach().cpu().numpy() for name, _ in P.manifest})
            tf_checkpoint.write_checkpoint(os.path.join(args.output, 'weights', 'model-%d' % step), tensors)
            print('Checkpoint saved in {}/weights'.format(args.output))
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
    parser.add_argument("--logDispStep", help="dump the disparity every K frames (-1: never)", default=-1, type=int)
    parser.add_argument("--eval", help="accepted for compatibility", choices=['DISP', 'DEPTH', 'SSIM'], default='DISP')
    parser.add_argument("--saveWeights", help="write the adapted weights as a TF checkpoint under <output>/weights", action='store_true')
    parser.add_argument("--dilation", help="update the weights only every K-th frame", type=int, default=1)
    parser.add_argument("--decay", help="multiplicative decay of the sampling logits", type=float, default=0.99)
    parser.add_argument("--uf", help="gain of the reward added to the logits of the last trained portions", type=float, default=0.01)
    return parser


if __name__ == '__main__':
    args = build_parser().parse_args()
    if not os.path.exists(args.output):
        os.makedirs(args.output)
    os.makedirs(os.path.join(args.output, 'weights'), exist_ok=True)
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
