"""Live demo driver behind the reference's CLI (Demo/Live_Adaptation_Demo.py:15-71): a grabber thread feeds a one-slot queue, the
RealTimeStereo thread adapts on every frame it takes from it.  Same flags; on top of them --frames (stop after N frames instead of
waiting for a key press: headless runs), --output / --logDispStep (disparity PNGs, the windows' replacement), --device and
--rewardAsOnline.  --cameraName picks a registered source: 'Synthetic' and 'ImageList' ship with the package (grabber.py), camera SDK
wrappers are registered by the user."""
import argparse
import inspect
import os
import queue
import sys

currentdir = os.path.dirname(os.path.abspath(inspect.getfile(inspect.currentframe())))
sys.path.insert(0, os.path.dirname(currentdir))
sys.path.insert(0, currentdir)

import numpy as np  # noqa: E402

import Nets  # noqa: E402
import grabber  # noqa: E402
import demo_model  # noqa: E402


def build_parser():
    parser = argparse.ArgumentParser(description='Real-time self-adaptive stereo, live loop on the MI355X engine')
    parser.add_argument("--modelName", help="which registered stereo network to run", default="MADNet", choices=Nets.STEREO_FACTORY.keys())
    parser.add_argument("--weights", help="initial weights (TF checkpoint prefix, .npz, xavier[:seed], calibrated[:seed]); none = random initialisation", default=None)
    parser.add_argument("--mode", help="NONE = inference only, FULL = full back-propagation, MAD = one sampled portion per frame", default='NONE', choices=['NONE', 'FULL', 'MAD'])   # (the reference's default, Demo/Live_Adaptation_Demo.py:19)
    parser.add_argument("--strictWeights", help="fail if the weight file lacks a model variable (default: like the reference, restore what matches)", action='store_true')
    parser.add_argument("--lr", help="Adam learning rate", default=0.0001, type=float)
    parser.add_argument("--blockConfig", help="json file listing the layers of every trainable portion", default=os.path.join(currentdir, '..', 'block_config', 'MadNet_full.json'))
    parser.add_argument("--imageShape", help="height width the camera frames are rescaled to, -1 to disable", nargs='+', type=int, default=[480, 640])
    parser.add_argument("--cropShape", help="height width the rescaled frames are centre-cropped / padded to, -1 to disable", nargs='+', type=int, default=[320, 512])
    parser.add_argument("--SSIMTh", help="restore the initial network when the loss exceeds this value", type=float, default=0.5)
    parser.add_argument("--cameraConfig", help="json configuration of the frame source", default=None)
    parser.add_argument("--cameraName", help="registered frame source", default="Synthetic", choices=grabber.get_available_camera())
    parser.add_argument("--frames", help="stop after this many frames (default: run until a key is pressed)", type=int, default=None)
    parser.add_argument("--framerate", help="target frames per second of the grabber (0 = as fast as the network takes them)", type=float, default=30)
    parser.add_argument("--output", help="folder for the disparity PNGs (16 bit, value*256)", default=None)
    parser.add_argument("--logDispStep", help="with --output: save the disparity every K frames", type=int, default=1)
    parser.add_argument("--device", help="torch device of the engine", default='cuda')
    parser.add_argument("--rewardAsOnline", help="block sampling rewards as in Stereo_Online_Adaptation.py instead of the demo's (see demo_model.py)", action='store_true')
    return parser


def main(args):
    assert args.cameraConfig is None or os.path.exists(args.cameraConfig)
    assert len(args.imageShape) in (1, 2) and len(args.cropShape) in (1, 2)
    on_frame = None
    if args.output is not None:
        os.makedirs(args.output, exist_ok=True)

        def on_frame(it, record, left, right, disp):
            if it % max(1, args.logDispStep) == 0:
                from PIL import Image
                d = (np.clip(disp[0].detach().cpu().numpy(), 0, 255) * 256.0).astype(np.uint16)
                Image.fromarray(d).save(os.path.join(args.output, 'disparity_{}.png'.format(it)))

    camera_frames = queue.Queue(1)                      # one slot: the network always works on the freshest frame
    dd = demo_model.RealTimeStereo(camera_frames, model_name=args.modelName, weight_path=args.weights, learning_rate=args.lr,
                                   block_config_path=args.blockConfig, image_shape=args.imageShape, crop_shape=args.cropShape,
                                   SSIMTh=args.SSIMTh, mode=args.mode, device=args.device, on_frame=on_frame, max_frames=args.frames,
                                   reward_as_online=args.rewardAsOnline, allow_missing=not args.strictWeights)
    gg = grabber.get_camera(args.cameraName, camera_frames, config=args.cameraConfig, framerate=args.framerate)
    print('Threads ready to start')
    gg.start()
    dd.start()
    if args.frames is None and sys.stdin.isatty():
        input('Press something to stop')
    else:
        dd.join()
    print('Requesting Stops')
    gg.stop()
    gg.join()
    print('Camera grabber stopped')
    dd.stop()
    dd.join()
    print('detector stopped')
    print('frames: {}  resets: {}  steady-state FPS: {:.1f}'.format(len(dd.history), sum(1 for h in dd.history if h[2]), dd.frames_per_second))
    if dd.error is not None:
        raise dd.error
    return dd


if __name__ == '__main__':
    main(build_parser().parse_args())
