"""List reader of the continual-adaptation driver: rows `left,right,gt,proxy` separated by ',' or ';', '#' lines skipped
(Data_utils/continual_data_reader.py:55-78); frames are centre-cropped / zero-padded to crop_shape
(tf.image.resize_image_with_crop_or_pad, :150), gt and proxy cut to the left image's width first (:137,146); 16-bit
PNGs are value/256 (:128-133,139-144).  Yields (left, right, gt, proxy, real_width) with [1,H,W,C] float32 arrays."""
import re

import numpy as np

from .data_reader import _read_image, center_crop_or_pad


def read_list_file(path_file):
    with open(path_file, 'r') as f_in:
        lines = [x for x in f_in.readlines() if x.strip() and not x.strip()[0] == '#']
    left, right, gt, proxy = [], [], [], []
    for l in lines:
        to_load = re.split(',|;', l.strip())
        left.append(to_load[0]); right.append(to_load[1])
        if len(to_load) > 2:
            gt.append(to_load[2])
        if len(to_load) > 3:
            proxy.append(to_load[3])
    return left, right, gt, proxy


class dataset(object):
    def __init__(self, path_file, batch_size=1, crop_shape=(320, 1216), num_epochs=1, augment=False, is_training=False,
                 proxies=True, shuffle=False):
        if batch_size != 1 or augment or is_training or shuffle:
            raise NotImplementedError('continual adaptation reads frames in order, batch 1, no augmentation')
        l, r, g, p = read_list_file(path_file)
        if not (len(l) == len(r) == len(g) == len(p)):
            raise Exception('Expected rows left,right,gt,proxy')
        self._couples = list(zip(l, r, g, p))
        self._crop, self._epochs = tuple(crop_shape), num_epochs

    def __len__(self):
        return len(self._couples)

    def get_max_steps(self):
        return len(self) * self._epochs

    def __iter__(self):
        th, tw = self._crop
        for _ in range(self._epochs):
            for l, r, g, p in self._couples:
                left = _read_image(l); right = _read_image(r)
                real_width = left.shape[1]
                gt = _read_image(g, True)[:, :real_width]
                px = _read_image(p, True)[:, :real_width]
                yield tuple(center_crop_or_pad(a, th, tw)[None] for a in (left, right, gt, px)) + (np.float32(real_width),)
