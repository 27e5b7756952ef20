"""Frame sources of the live demo behind the reference's grabber surface (Demo/grabber.py:13-29 factory, :36-92 ImageGrabber): a
thread that reads rectified left/right frames and puts them, stacked as one [2,h,w,c] array, on the shared queue the adaptation
thread (Demo/demo_model.RealTimeStereo) drains.

The camera SDK wrappers themselves (ZED Mini, Demo/grabber.py:100-150) need the vendor's Python module and the device; they are not
part of this package.  A camera is added the way the reference's commented example does it: subclass ImageGrabber, give it a
`_name`, implement the three hooks, decorate with @register_camera_to_factory().  Two file-less / hardware-less sources are
registered so the loop runs -- and is tested -- anywhere: 'Synthetic' (seeded textured scenes with a known disparity) and
'ImageList' (the CSV lists the adaptation scripts read).

Differences from the reference's thread, on purpose: put() polls with a timeout so stop() ends the thread even when nobody drains
the queue (the reference's blocking put can hang there), and a source that runs dry (a list without "loop") hands a None over as
the end-of-stream mark.
"""
import abc
import json
import queue
import threading
import time

import numpy as np

_GRABBER_FACTORY = {}


def get_camera(name, frame_queue, config=None, framerate=30):
    """Demo/grabber.py:13-20."""
    if name not in _GRABBER_FACTORY:
        raise Exception('Unrecognized camera type: {}'.format(name))
    return _GRABBER_FACTORY[name](frame_queue, config=config, framerate=framerate)


def get_available_camera():
    return _GRABBER_FACTORY.keys()


def register_camera_to_factory():
    def decorator(cls):
        _GRABBER_FACTORY[cls._name] = cls
        return cls
    return decorator


class ImageGrabber(threading.Thread, metaclass=abc.ABCMeta):
    """frame_queue: synchronized queue the frames go to; config: path of a json file (or a dict) with the source's parameters;
    framerate: target frames per second (the thread sleeps 1/framerate between frames; 0 = as fast as the consumer takes them)."""

    def __init__(self, frame_queue, config=None, framerate=30):
        threading.Thread.__init__(self, daemon=True)
        self._config = config
        self._connect_to_camera()
        self._buffer = frame_queue
        self._sleeptime = 1.0 / framerate if framerate else 0.0
        self._stop_acquire = False
        self.frames_delivered = 0

    def stop(self):
        self._stop_acquire = True

    def _put(self, item):
        while not self._stop_acquire:
            try:
                self._buffer.put(item, timeout=0.05)
                return True
            except queue.Full:
                continue
        return False

    def run(self):
        try:
            while not self._stop_acquire:
                frame = self._read_frame()
                if frame is None:                       # the source ran dry: tell the consumer, then leave
                    self._put(None)
                    break
                left, right = frame
                if self._put(np.stack([left, right], axis=0)):
                    self.frames_delivered += 1
                if self._sleeptime:
                    time.sleep(self._sleeptime)
        finally:
            self._disconnect_from_camera()

    @abc.abstractmethod
    def _read_frame(self):
        """-> (left, right) rectified [h,w,3] arrays, or None when there is nothing more to read."""

    @abc.abstractmethod
    def _connect_to_camera(self):
        pass

    @abc.abstractmethod
    def _disconnect_from_camera(self):
        pass


def _load_config(config):
    if config is None:
        return {}
    if isinstance(config, dict):
        return dict(config)
    with open(config) as f_in:
        return json.load(f_in)


@register_camera_to_factory()
class SyntheticStereo(ImageGrabber):
    """Seeded synthetic scenes (madnet_hip.synthetic.make_pair).  config: {"height": 480, "width": 640, "stream": 0,
    "frames": N or null (endless), "distinct": K scenes cycled}.  The ground truth of the last frame read is kept in `last_gt`."""
    _name = 'Synthetic'

    def _connect_to_camera(self):
        from madnet_hip import synthetic
        c = _load_config(self._config)
        self._h, self._w = int(c.get('height', 480)), int(c.get('width', 640))
        self._stream = int(c.get('stream', 0))
        self._limit = c.get('frames')
        self._distinct = max(1, int(c.get('distinct', 4)))
        self._make = synthetic.make_pair
        self._cache = {}
        self._n = 0
        self.last_gt = None

    def _read_frame(self):
        if self._limit is not None and self._n >= int(self._limit):
            return None
        k = self._n % self._distinct
        if k not in self._cache:
            l, r, gt = self._make(self._h, self._w, stream_id=self._stream + k)
            self._cache[k] = (l[0].astype(np.uint8), r[0].astype(np.uint8), gt[0])
        self._n += 1
        l, r, self.last_gt = self._cache[k]
        return l, r

    def _disconnect_from_camera(self):
        self._cache = {}


@register_camera_to_factory()
class ImageList(ImageGrabber):
    """Replays a recorded sequence: config {"list": CSV with one left,right[,...] row of image paths per frame -- the lists of
    the adaptation scripts work as they are, further columns are ignored --, "loop": false}."""
    _name = 'ImageList'

    def _connect_to_camera(self):
        from Data_utils import data_reader
        c = _load_config(self._config)
        if 'list' not in c:
            raise Exception("ImageList needs a config with a 'list' entry (CSV of left,right image paths)")
        with open(c['list']) as f_in:
            self._rows = [x.strip().split(',') for x in f_in if x.strip()]
        if any(len(r) < 2 for r in self._rows):
            raise Exception('Expected lines with at least 2 comma separated fields: left,right')
        if not self._rows:
            raise Exception('{} lists no frames'.format(c['list']))
        self._loop = bool(c.get('loop', False))
        self._read = data_reader._read_image
        self._n = 0

    def _read_frame(self):
        if self._n >= len(self._rows):
            if not self._loop:
                return None
            self._n = 0
        row = self._rows[self._n]
        self._n += 1
        return self._read(row[0], keep_uint8=True), self._read(row[1], keep_uint8=True)

    def _disconnect_from_camera(self):
        self._rows = []
