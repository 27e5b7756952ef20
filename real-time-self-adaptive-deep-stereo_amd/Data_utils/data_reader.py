"""Input list reader for the online-adaptation driver and the offline trainer: same list format and centre crop/pad
semantics as the reference's tf.data pipeline (Data_utils/data_reader.py:55-197), as a plain Python
iterator (host IO is outside the hot path).

List file: one sample per row `left,right,gt` (Data_utils/data_reader.py:55-78).  Images: PNG/JPG
via Pillow; ground truth: 16-bit PNG (value/256, KITTI convention, :88-92), .pfm (:11-53) or .npy.
Every frame is centre-cropped / zero-padded to crop_shape like tf.image.resize_image_with_crop_or_pad
(:150) and yielded as float32 [1,H,W,C] holding the raw 0..255 values (:98)."""
import os
import re

import time
import numpy as np


def read_list_file(path_file):
    """Returns (left_files, right_files, gt_files); rows must have >= 3 comma separated fields."""
    with open(path_file, 'r') as f_in:
        rows = [x.strip().split(',') for x in f_in.readlines() if x.strip()]
    if any(len(r) < 3 for r in rows):
        raise Exception('Expected lines with at least 3 comma separated fields: left,right,gt')
    return [r[0] for r in rows], [r[1] for r in rows], [r[2] for r in rows]


def readPFM(file):
    """Portable float map reader (header: PF|Pf, 'w h', scale; rows bottom-to-top)."""
    with open(file, 'rb') as f:
        header = f.readline().rstrip().decode('ascii')
        if header not in ('PF', 'Pf'):
            raise Exception('Not a PFM file.')
        color = header == 'PF'
        m = re.match(r'^(\d+)\s(\d+)\s$', f.readline().decode('ascii'))
        if not m:
            raise Exception('Malformed PFM header.')
        width, height = int(m.group(1)), int(m.group(2))
        scale = float(f.readline().rstrip().decode('ascii'))
        data = np.fromfile(f, ('<' if scale < 0 else '>') + 'f')
    shape = (height, width, 3) if color else (height, width, 1)
    return np.flipud(np.reshape(data, shape)).astype(np.float32), abs(scale)


def _read_image(path, is_gt=False, keep_uint8=False):
    """keep_uint8: 8-bit images stay uint8 [H,W,3] (the float cast then happens on the GPU, device_prefetcher)."""
    ext = os.path.splitext(path)[1].lower()
    if ext == '.npy':
        a = np.load(path).astype(np.float32)
        return a if a.ndim == 3 else a[..., None]
    if ext == '.pfm':
        return readPFM(path)[0]
    from PIL import Image
    im = Image.open(path)
    a = np.asarray(im)
    if is_gt:
        a = a.astype(np.float32)
        if a.ndim == 3:
            a = a[..., 0]
        if np.asarray(im).dtype != np.uint8:
            a = a / 256.0                       # 16-bit KITTI disparity PNG
        return a[..., None]
    if not (keep_uint8 and a.dtype == np.uint8):
        a = a.astype(np.float32)
    if a.ndim == 2:
        a = np.stack([a, a, a], -1)
    return a[..., :3]


def center_crop_or_pad(img, th, tw):
    """tf.image.resize_image_with_crop_or_pad: centre crop (offset (in-target)//2) / zero pad."""
    h, w = img.shape[:2]
    if h > th:
        o = (h - th) // 2
        img = img[o:o + th]
    if w > tw:
        o = (w - tw) // 2
        img = img[:, o:o + tw]
    h, w = img.shape[:2]
    if h < th or w < tw:
        pt, pl = (th - h) // 2, (tw - w) // 2
        out = np.zeros((th, tw) + img.shape[2:], img.dtype)
        out[pt:pt + h, pl:pl + w] = img
        img = out
    return img


def random_crop(crop_shape, arrays, rng):
    """Aligned random crop of [H,W,C] arrays (preprocessing.random_crop, Data_utils/preprocessing.py:31-58): the start row /
    column are uniform in [0, H - crop_h - 1) / [0, W - crop_w - 1) -- the reference's own upper bounds, which never pick the
    last admissible offset -- and [0, 1) = 0 when the image is not larger than the crop (short images are NOT padded there:
    the slice is simply shorter and tf.set_shape fails; here that case raises)."""
    h, w = arrays[0].shape[:2]
    ch, cw = int(crop_shape[0]), int(crop_shape[1])
    if h < ch or w < cw:
        raise ValueError("random_crop: image %dx%d smaller than the crop %dx%d" % (h, w, ch, cw))
    max_row, max_col = h - ch - 1, w - cw - 1
    r0 = int(rng.integers(0, max_row if max_row > 0 else 1))
    c0 = int(rng.integers(0, max_col if max_col > 0 else 1))
    return [x[r0:r0 + ch, c0:c0 + cw] for x in arrays]


def _rgb_to_hsv(x):
    mx, mn = x.max(-1), x.min(-1)
    d = mx - mn
    s = np.where(mx > 0, d / np.where(mx > 0, mx, 1), 0)
    dd = np.where(d > 0, d, 1)
    r, g, b = x[..., 0], x[..., 1], x[..., 2]
    h = np.where(mx == r, (g - b) / dd, np.where(mx == g, 2.0 + (b - r) / dd, 4.0 + (r - g) / dd))
    h = np.where(d > 0, (h / 6.0) % 1.0, 0.0)
    return h, s, mx


def _hsv_to_rgb(h, s, v):
    k = (np.stack([h * 6.0 + 5.0, h * 6.0 + 3.0, h * 6.0 + 1.0], -1)) % 6.0
    return v[..., None] - (v * s)[..., None] * np.clip(np.minimum(k, 4.0 - k), 0.0, 1.0)


def augment(left_img, right_img, rng):
    """preprocessing.augment (Data_utils/preprocessing.py:63-89) on float32 [H,W,3] images holding 0..255: with probability 1/2
    each (applied when the uniform draw is <= 0.5, like the tf.where there) the SAME brightness delta in +-0.05, contrast
    factor in [0.8, 1.2] and hue rotation in [0.8, 1.2] turns (i.e. +-0.2 of the colour circle) go on both views; then
    clip to [0, 255].  (The gamma branch is commented out in the reference.)"""
    active = rng.uniform(0.0, 1.0, size=4)
    delta = rng.uniform(-0.05, 0.05)
    contrast = rng.uniform(0.8, 1.2)
    hue = rng.uniform(0.8, 1.2)
    out = []
    for img in (left_img, right_img):
        x = np.asarray(img, np.float32)
        if active[1] <= 0.5:
            x = x + np.float32(delta)                                   # tf.image.adjust_brightness on a float image
        if active[2] <= 0.5:
            m = x.mean(axis=(0, 1), keepdims=True)                       # tf.image.adjust_contrast: per-channel mean
            x = (x - m) * np.float32(contrast) + m
        if active[3] <= 0.5:
            h, s, v = _rgb_to_hsv(x)
            x = _hsv_to_rgb((h + hue) % 1.0, s, v).astype(np.float32)
        out.append(np.clip(x, 0.0, 255.0).astype(np.float32))
    return out[0], out[1]


class dataset(object):
    """Iterator with the reference's constructor surface (Data_utils/data_reader.py:104-197).  Online adaptation reads the list
    in order, batch 1, centre crop / pad.  is_training=True is Train.py's pipeline: repeat(num_epochs) -> shuffle buffer of
    50 * batch_size samples -> aligned random crop -> optional augmentation -> batches of batch_size (remainder dropped)."""

    def __init__(self, path_file, batch_size=1, crop_shape=(320, 1216), num_epochs=1, augment=False,
                 is_training=False, shuffle=False, seed=0, keep_uint8=False, shard=(0, 1)):
        """keep_uint8 (no augmentation): 8-bit frames are yielded as uint8 [B,H,W,3] instead of float32 -- device_prefetcher then
        moves 1 byte per value over PCIe and casts on the GPU (mh_u8_to_f32); values are identical."""
        self._u8 = bool(keep_uint8) and not augment
        # shard = (rank, world): data-parallel training reads every world-th sample of each epoch (one pass over the list per
        # epoch in total, not one per rank)
        self._rank, self._world = int(shard[0]), max(1, int(shard[1]))
        self._left, self._right, self._gt = read_list_file(path_file)
        self._crop = tuple(crop_shape)
        self._epochs = num_epochs
        self._batch, self._augment, self._training, self._shuffle = int(batch_size), augment, is_training, shuffle
        self._rng = np.random.default_rng(seed)

    def __len__(self):
        return len(self._left)

    def _per_rank(self):
        """samples of one epoch EVERY rank reads: the list is cut to a multiple of the world size, so that all ranks yield the same number of
        batches -- each training step issues collectives (madnet_hip/trainer.py), and a rank with one batch more would wait in an all-reduce
        its peers never enter"""
        return len(self._left) // self._world

    def get_max_steps(self):
        return (self._per_rank() * self._epochs) // self._batch

    def _samples(self):
        n = self._per_rank() * self._world
        order = [i for _ in range(self._epochs) for i in range(self._rank, n, self._world)]
        if not self._shuffle:
            for i in order:
                yield i
            return
        buf, cap = [], self._batch * 50                   # tf.data shuffle(buffer_size): draw uniformly from a sliding buffer
        for i in order:
            buf.append(i)
            if len(buf) > cap:
                yield buf.pop(int(self._rng.integers(0, len(buf))))
        while buf:
            yield buf.pop(int(self._rng.integers(0, len(buf))))

    def _load(self, i):
        th, tw = self._crop
        l, r, g = _read_image(self._left[i], keep_uint8=self._u8), _read_image(self._right[i], keep_uint8=self._u8), _read_image(self._gt[i], True)
        g = g[:, :l.shape[1]]                             # "crop gt to fit with image" (:146)
        if self._training:
            l, r, g = random_crop(self._crop, [l, r, g], self._rng)
        else:
            l, r, g = (center_crop_or_pad(x, th, tw) for x in (l, r, g))
        if self._augment:
            l, r = augment(l, r, self._rng)
        return l, r, g

    def __iter__(self):
        batch = []
        for i in self._samples():
            batch.append(self._load(i))
            if len(batch) == self._batch:
                yield tuple((lambda a: a if (self._u8 and a.dtype == np.uint8) else a.astype(np.float32))(np.stack([b[k] for b in batch]))
                            for k in range(3))
                batch = []


class device_prefetcher(object):
    """Decode-ahead + host-to-device overlap for the online loop (replaces tf.data's prefetch,
    Data_utils/data_reader.py:171-175): a reader thread decodes frames `depth` ahead into a ring of PINNED host
    buffers, the copies are issued on a private copy stream and each yielded triple carries a HIP event the
    consumer stream waits on -- frame t+1 is decoded and uploaded while frame t adapts.

        for left, right, gt in device_prefetcher(dataset(...), 'cuda', consumer_stream=adapter.stream):
            adapter.step(left, right, gt)

    The hand-over keeps the consumer's thread out of the reader's way (round 6, scripts/exp/prefetch_phases*.py: a reader woken by the consumer between two steps
    cost the loop 30 us per step in GIL ping-pong, a cross-stream event wait in front of the step's graph 10 more): the consumer never wakes the reader -- it appends
    the slot it is done with (+ an event on its stream) to a deque the reader POLLS when it runs out of slots -- never synchronises its stream, and waits on a
    frame's upload event only if the upload is not complete yet.
    """

    POLL = 2e-4            # seconds between two looks of a reader that is out of slots

    def __init__(self, data_set, device='cuda', depth=3, consumer_stream=None, lib=None, cast=True):
        """uint8 arrays from the data set are uploaded as uint8; cast=True: cast to float32 on the GPU (mh_u8_to_f32 on the copy stream; `lib` = the loaded
        library, default the product's) so that the consumer always sees float32 tensors; cast=False: yielded as uint8 device tensors (Adapter.step casts
        while it copies them into the engine's input buffers: one kernel and 5.6 MB of traffic less per image)."""
        self._lib = lib
        import collections
        import queue
        import threading
        import torch
        self._torch = torch
        self._ds, self._depth = data_set, max(2, depth)
        self._dev = torch.device(device)
        self._cuda = self._dev.type == 'cuda'
        self._cast = cast
        self._q = queue.Queue()                  # unbounded: the ring bounds what is in flight, and a put that never blocks is never woken by the consumer
        self._free = []                          # the reader's own list of free slots
        self._returned = collections.deque()     # (slot, event on the consumer's stream): appended by the consumer, popped by the reader
        self._ring = None
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._reader, daemon=True)
        self._copy_stream = torch.cuda.Stream(device=self._dev) if self._cuda else None
        self._consumer = consumer_stream         # stream the frames are consumed on (default: the current stream)

    def _slot(self, arrays):
        t = self._torch
        if self._ring is None:                   # allocate the ring on first use (shapes known now)
            self._ring = []
            u8 = [np.asarray(a).dtype == np.uint8 for a in arrays]
            if any(u8) and self._cast and self._lib is None:
                from madnet_hip import _ffi
                self._lib = _ffi.lib()
            for _ in range(self._depth + 1):
                host = [t.empty(np.shape(a), dtype=(t.uint8 if q else t.float32), pin_memory=self._cuda) for a, q in zip(arrays, u8)]
                stage = [t.empty(np.shape(a), dtype=t.uint8, device=self._dev) if q else None for a, q in zip(arrays, u8)]
                devb = [(s8 if (q and not self._cast) else t.empty(np.shape(a), dtype=t.float32, device=self._dev)) for a, q, s8 in zip(arrays, u8, stage)]
                self._ring.append((host, devb, t.cuda.Event() if self._cuda else None, stage, t.cuda.Event() if self._cuda else None))
            self._free = list(range(len(self._ring)))[::-1]
        while not self._free:
            try:
                i, done = self._returned.popleft()
            except IndexError:
                if self._stop.is_set():
                    return None
                time.sleep(self.POLL)
                continue
            if done is not None:
                done.synchronize()               # what the consumer enqueued on this slot's buffers has run (the GIL is released while waiting)
            self._free.append(i)
        return self._free.pop()

    def _reader(self):
        t = self._torch
        try:
            for arrays in self._ds:
                if self._stop.is_set():
                    return
                i = self._slot(arrays)
                if i is None:
                    return
                host, devb, ev, stage, _ = self._ring[i]
                for h, a in zip(host, arrays):
                    np.copyto(h.numpy(), np.asarray(a).reshape(tuple(h.shape)), casting='unsafe')     # straight into the pinned slot
                if self._cuda:
                    with t.cuda.stream(self._copy_stream):
                        for h, d, s8 in zip(host, devb, stage):
                            if s8 is None:
                                d.copy_(h, non_blocking=True)
                            else:
                                s8.copy_(h, non_blocking=True)
                                if self._cast:
                                    self._lib.u8_to_f32(s8.data_ptr(), d.data_ptr(), s8.numel(), self._copy_stream.cuda_stream)
                        ev.record(self._copy_stream)
                else:
                    for h, d, s8 in zip(host, devb, stage):
                        if s8 is None:
                            d.copy_(h)
                        else:
                            s8.copy_(h)
                            if self._cast and self._lib is not None:
                                self._lib.u8_to_f32(s8.data_ptr(), d.data_ptr(), s8.numel(), None)
                self._q.put(i)
            self._q.put(None)
        except Exception as e:                   # surface reader errors in the consumer
            self._q.put(e)

    def __iter__(self):
        self._thread.start()
        prev = None
        while True:
            i = self._q.get()
            if prev is not None:
                # the slot handed out last time: whatever the consumer enqueued on its buffers is in its stream by now -- an event behind it frees the slot for the reader
                done = None
                if self._cuda:
                    done = self._ring[prev][4]
                    done.record(self._consumer or self._torch.cuda.current_stream(self._dev))
                self._returned.append((prev, done))
            if i is None:
                return
            if isinstance(i, Exception):
                raise i
            host, devb, ev, _, _ = self._ring[i]
            if self._cuda and not ev.query():        # (uploads run a step ahead: usually complete -- no cross-stream edge in front of the step then)
                (self._consumer or self._torch.cuda.current_stream(self._dev)).wait_event(ev)
            prev = i
            yield tuple(devb)

    def close(self):
        self._stop.set()
