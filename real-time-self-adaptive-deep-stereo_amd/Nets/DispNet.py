"""DispNet-C entry of the factory (Nets/DispNet.py:9-152).  The class name, kwargs and validation
are kept; the MI355X engine for its graph (7x7/5x5 stride-2 convs, 1x1 redir, D=81 correlation,
4x4 transposed convs -- all of which the conv / corr kernels of libmadnet_hip.so already implement
and test) is the next §8 row and is not wired yet: constructing the net raises a clear error."""
from Nets import Stereo_net

MAX_DISP = 40


class DispNet(Stereo_net.StereoNet):
    _valid_args = [
        ("left_img", "tensor [B,H,W,3] for the left image batch"),
        ("right_img", "tensor [B,H,W,3] for the right image batch"),
        ("correlation", "flag to enable the use of the correlation layer"),
    ] + Stereo_net.StereoNet._valid_args
    _netName = "Dispnet"

    def __init__(self, **kwargs):
        super(DispNet, self).__init__(**kwargs)

    def _validate_args(self, args):
        args = super(DispNet, self)._validate_args(args)
        if ("left_img" not in args) or ("right_img" not in args):
            raise Exception('Missing input op for left and right images')
        if "correlation" not in args:
            print('WARNING: correlation layer flag not setted, using default True value')
            args['correlation'] = True
        return args

    def _preprocess_inputs(self, args):
        self._left_input_batch = args['left_img']
        self._right_input_batch = args['right_img']

    def _build_network(self, args):
        raise NotImplementedError("DispNet graph executor is not wired yet on MI355X (MADNet is); "
                                  "see DESIGN.md 'what comes next'")
