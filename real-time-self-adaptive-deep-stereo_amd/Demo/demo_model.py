"""The adaptation thread of the live demo behind the reference's surface (Demo/demo_model.py:12-66 constructor, :223-287 stop/run):
frames come from the grabber's queue, every frame is ONE replay of a compiled step plan of the MI355X engine (forward, full-resolution
reprojection loss, the sampled block's -- or the whole network's -- backward, the Adam update; madnet_hip.adapter.Adapter with
optimizer='adam'), then the host-side reward / reset logic of Demo/demo_model.py:252-266.

What the reference builds per call of session.run -- placeholders, rescale to image_shape, centre crop / pad to crop_shape (:72-86) --
happens here on the device right after the frame's upload: Data_utils.preprocessing.rescale_image (the HIP resize kernel) and a centre
crop / zero pad.

Reference behaviour kept as it is (so a sequence adapts the same way):
  * `first` is never cleared in the reference's run() (:236,253-255), so both remembered losses are re-seeded on every frame, the expected
    loss equals the current one and the gain is always 0: the block-sampling logits stay at zero (uniform sampling).  Pass
    reward_as_online=True for the online script's rule (Stereo_Online_Adaptation.py:211-224) instead.
  * a reset (loss > SSIMTh) with a weight file restores the variables of the file and leaves the Adam slots alone; without one the
    reference re-runs the initialisers, slots and beta powers included (:195-208).  Here the no-file reset returns to the SAME initial
    draw (the reference draws new random values) and clears the slots.

No OpenCV / GUI dependency: the three imshow windows of :228-231,268-272 are replaced by the `on_frame` callback (and cv2 windows only
if display=True and cv2 imports).
"""
import json
import queue
import threading
import time

import numpy as np
import torch

import Nets
from Data_utils import preprocessing
from madnet_hip.adapter import Adapter


def _disabled(shape):
    return shape is None or shape[0] is None or int(shape[0]) < 0


def crop_or_pad(x, th, tw):
    """tf.image.resize_image_with_crop_or_pad on a [B,h,w,c] device tensor (Demo/demo_model.py:84-86): centre crop with offset
    (in - target)//2, centre zero pad with (target - in)//2 in front."""
    h, w = x.shape[1], x.shape[2]
    if h > th:
        o = (h - th) // 2
        x = x[:, o:o + th]
    if w > tw:
        o = (w - tw) // 2
        x = x[:, :, o:o + tw]
    h, w = x.shape[1], x.shape[2]
    if h < th or w < tw:
        out = x.new_zeros(x.shape[0], th, tw, x.shape[3])
        pt, pl = (th - h) // 2, (tw - w) // 2
        out[:, pt:pt + h, pl:pl + w] = x
        x = out
    return x


class RealTimeStereo(threading.Thread):
    """Real time self adaptive stereo: camera_buffer is the queue the grabber fills with [2,h,w,c] frames (None = end of stream)."""

    def __init__(self, camera_buffer, model_name='MADNet', weight_path=None, learning_rate=0.0001,
                 block_config_path='../block_config/MadNet_full.json', image_shape=[480, 640], crop_shape=[None, None],
                 SSIMTh=0.5, mode='MAD', device='cuda', on_frame=None, display=False, max_frames=None, reward_as_online=False,
                 precision=None, _lib=None, allow_missing=True):
        """allow_missing (default True, like the reference demo whose Saver restores the matching names and leaves the rest at their
        initializer, weights_utils.get_var_to_restore_list): a checkpoint that lacks some model variables still loads."""
        if mode not in ('NONE', 'FULL', 'MAD'):
            raise ValueError('mode must be NONE, FULL or MAD')
        self._camera_buffer = camera_buffer
        self._model_name = model_name
        self._weight_path = weight_path
        self._learning_rate = learning_rate
        self._block_config_path = block_config_path
        self._image_shape = None if _disabled(image_shape) else [int(image_shape[0]), int(image_shape[1])]
        self._crop_shape = None if _disabled(crop_shape) else [int(crop_shape[0]), int(crop_shape[1])]
        self._SSIMTh = SSIMTh
        self._mode = mode
        self._device = device
        self._on_frame = on_frame
        self._display = display
        self._max_frames = max_frames
        self._reward_as_online = reward_as_online
        self._precision = precision
        self._lib = _lib
        self._allow_missing = bool(allow_missing)
        self._stop_flag = False
        self._adapter = None
        self.history = []                              # (loss, trained blocks, reset?) per processed frame
        self.frames_per_second = 0.0
        self.error = None
        net_shape = self._crop_shape or self._image_shape
        self._ready = False
        if net_shape is not None:                      # otherwise the network is sized by the first frame the camera delivers
            self._ready = self._setup_graph(net_shape)
        threading.Thread.__init__(self, daemon=True)

    # ------------------------------------------------------------------------------------------------
    def _load_block_config(self):
        with open(self._block_config_path) as json_data:
            self._train_config = json.load(json_data)

    def _initial_weights(self):
        # Stereo_Online_Adaptation.load_weights understands TF checkpoints, .npz and the synthetic initialisers; no file = the
        # reference's global_variables_initializer (Demo/demo_model.py:192-206), here a seeded Xavier draw
        import Stereo_Online_Adaptation as SOA
        return SOA.load_weights(self._weight_path if self._weight_path is not None else 'xavier:0', self._model_name, allow_missing=self._allow_missing)

    def _setup_graph(self, net_shape):
        H, W = net_shape
        z = torch.zeros(1, H, W, 3, device=self._device)
        net_args = {'left_img': z, 'right_img': z, 'split_layers': [None], 'sequence': True, 'train_portion': 'BEGIN',
                    'bulkhead': True if self._mode == 'MAD' else False, 'weights': self._initial_weights()}
        if self._lib is not None:
            net_args['_lib'], net_args['_device'] = self._lib, self._device
        if self._precision is not None:
            net_args['precision'] = self._precision
        self._net = Nets.get_stereo_net(self._model_name, net_args)
        self._train_config = None
        if self._mode == 'MAD':
            self._load_block_config()
        # Demo/demo_model.py:144-154: PROBABILITY sampler with one block for MAD, a fixed single op otherwise
        self._adapter = Adapter(self._net, mode=self._mode, block_config=self._train_config, lr=self._learning_rate,
                                sample_mode='PROBABILITY', num_blocks=1, fixed_id=[0], ssim_th=self._SSIMTh, optimizer='adam',
                                reset_optimizer=self._weight_path is None, reward_every_step_first=not self._reward_as_online)
        self._net_shape = (H, W)
        print('Network Ready')
        return True

    def _prepare(self, frames):
        """[2,h,w,c] camera frames -> left, right [1,H,W,3] float32 device tensors of the network's shape."""
        x = torch.as_tensor(np.ascontiguousarray(frames)[..., :3]).to(self._device, non_blocking=True).to(torch.float32)
        if self._image_shape is not None:
            x = preprocessing.rescale_image(x, self._image_shape)
        if self._crop_shape is not None:
            x = crop_or_pad(x, *self._crop_shape)
        if not self._ready:
            self._ready = self._setup_graph((x.shape[1], x.shape[2]))
        if (x.shape[1], x.shape[2]) != self._net_shape:
            raise ValueError('frame of shape {} after rescale / crop, the network was built for {}'.format(tuple(x.shape[1:3]), self._net_shape))
        return x[:1], x[1:]

    def stop(self):
        """Stop the prediction and end the thread."""
        self._stop_flag = True

    def _show(self, left, right, disp):
        if self._on_frame is not None:
            self._on_frame(len(self.history) - 1, self.history[-1], left, right, disp)
        if self._display:
            import cv2
            cv2.imshow('left frame', left[0].cpu().numpy().astype(np.uint8))
            cv2.imshow('right frame', right[0].cpu().numpy().astype(np.uint8))
            cv2.imshow('disparity prediction', cv2.applyColorMap(disp[0].clamp(0, 255).cpu().numpy().astype(np.uint8), cv2.COLORMAP_JET))
            cv2.waitKey(1)

    def run(self):
        it = 0
        t0 = None
        try:
            while not self._stop_flag:
                try:
                    frames = self._camera_buffer.get(block=True, timeout=0.05)
                except queue.Empty:
                    continue
                if frames is None:                         # the grabber's end-of-stream mark
                    break
                left, right = self._prepare(frames)
                if self._device != 'cpu' and self._adapter.cuda:
                    self._adapter.stream.wait_stream(torch.cuda.current_stream())      # the step's plan runs on the adapter's stream
                out = self._adapter.step(left, right)
                if t0 is None:
                    t0 = time.time()                       # the first frame compiles / captures the plan: not part of the rate
                full_ssim = out['loss']
                print('Step {}: {}'.format(it, full_ssim))
                if out['reset']:
                    print('Resetting Network...')
                self.history.append((full_ssim, out['blocks'], out['reset']))
                self._show(left, right, out['disparity'])
                it += 1
                if self._max_frames is not None and it >= self._max_frames:
                    break
        except Exception as e:                             # a thread's exception would otherwise only be printed: keep it for the caller
            self.error = e
            raise
        finally:
            if t0 is not None and it > 1:
                self.frames_per_second = (it - 1) / max(time.time() - t0, 1e-9)
            if self._display:
                import cv2
                cv2.destroyAllWindows()

    @property
    def sample_distribution(self):
        return None if self._adapter is None else self._adapter.sample_distribution
