"""DispNet-C on MI355X: same class name, kwargs, layer keys and variable names as the reference graph
builder (Nets/DispNet.py:9-152), executed by madnet_hip.dispnet_engine.DispNetEngine (7x7/5x5 stride-2
convs, 1x1 redir, D=81 correlation, 4x4 transposed convs: all hand-written HIP behind include/madnet_hip.h).
Modes NONE / FULL; MAD is unusable for DispNet in the reference itself (5 config groups vs 6 predictions:
the assert at Stereo_Online_Adaptation.py:97 fails)."""
import torch

from Nets import Stereo_net
from madnet_hip import _ffi, dispnet_engine as DE, synthetic

MAX_DISP = DE.MAX_DISP


class DispNet(Stereo_net.StereoNet):
    _valid_args = [
        ("left_img", "tensor [B,H,W,3] for the left image batch"),
        ("right_img", "tensor [B,H,W,3] for the right image batch"),
        ("correlation", "flag to enable the use of the correlation layer"),
        ("weights", "(new) dict {TF variable name: array}; default Xavier like the reference initializer"),
    ] + Stereo_net.StereoNet._valid_args
    _netName = "Dispnet"

    def __init__(self, **kwargs):
        super(DispNet, self).__init__(**kwargs)

    def _validate_args(self, args):
        args = super(DispNet, self)._validate_args(args)
        if ("left_img" not in args) or ("right_img" not in args):
            raise Exception('Missing input op for left and right images')
        if "correlation" not in args:
            print('WARNING: Correlation unspecified, setting to True')
            args['correlation'] = True
        if not args['correlation']:
            # the reference's correlation=False branch reads an undefined attribute (DispNet.py:96, App. D.5)
            raise NotImplementedError("correlation=False is broken in the reference (AttributeError); not supported")
        return args

    def _preprocess_inputs(self, args):
        l, r = args['left_img'], args['right_img']
        if not (isinstance(l, torch.Tensor) and isinstance(r, torch.Tensor)) or l.dim() != 4 or l.shape[-1] != 3 or l.shape != r.shape:
            raise Exception('left_img / right_img must be torch tensors [B,H,W,3]')
        self._left_input_batch, self._right_input_batch = l, r
        self._restore_shape = (int(l.shape[1]), int(l.shape[2]))
        self._bulkhead = False

    def _build_network(self, args):
        l = self._left_input_batch
        B, H, W = int(l.shape[0]), int(l.shape[1]), int(l.shape[2])
        lib = args.get('_lib') or _ffi.lib()
        dev = args.get('_device') or l.device
        weights = args.get('weights')
        if weights is None:
            weights = synthetic.xavier_weights(dict(DE.dispnet_manifest()), seed=0)
        self.engine = eng = DE.DispNetEngine(lib, H, W, B=B, device=dev, weights=weights)
        self._lib = lib
        P = eng.params
        self._variables = {}

        def var_pair(scope):
            vs = [Stereo_net.Variable("model/%s/weights" % scope, P.tensor("model/%s/weights" % scope)),
                  Stereo_net.Variable("model/%s/bias" % scope, P.tensor("model/%s/bias" % scope))]
            for v in vs:
                self._variables[v.op_name] = v
            return vs

        def t(name):
            n = eng.nodes[name]
            return n.st.t[..., n.c0:n.c0 + n.C]

        # layer keys and the variables the reference registers for them (the reused
        # right-tower keys conv1b / conv2b map to [] like the reference's scope lookup)
        self._add_to_layers('conv1a', t('conv1a'), var_pair('conv1')); self._add_to_layers('conv1b', t('conv1b'), [])
        self._add_to_layers('conv2a', t('conv2a'), var_pair('conv2')); self._add_to_layers('conv2b', t('conv2b'), [])
        self._add_to_layers('conv_redir', t('conv_redir'), var_pair('conv_redir'))
        self._add_to_layers('corr', t('corr'), [])
        for k in ('conv3', 'conv3/1', 'conv4', 'conv4/1', 'conv5', 'conv5/1', 'conv6', 'conv6/1'):
            self._add_to_layers(k, t(k), var_pair(k))
        z = lambda: torch.zeros(B, H, W, device=dev)
        self._disp_bufs = []
        for name, _, _, _ in DE.UP_BLOCKS:
            for part in ('deconv', 'predict', 'up_predict', 'concat'):
                self._add_to_layers('%s/%s' % (name, part), t('%s/%s' % (name, part)), var_pair('%s/%s' % (name, part)))
            buf = z(); self._disp_bufs.append((name, buf)); self._disparities.append(buf[..., None])
        self._add_to_layers('prediction', t('prediction'), var_pair('prediction'))
        buf = z(); self._disp_bufs.append(('prediction', buf)); self._disparities.append(buf[..., None])
        self._layers['rescaled_prediction'] = eng.pred[..., None]
        self._disparities.append(self._layers['rescaled_prediction'])
        self._fwd_plan = None

    def run(self):
        """One forward pass on the current contents of left_img/right_img; refreshes all 7 disparities."""
        eng = self.engine
        eng.left.copy_(self._left_input_batch)
        eng.right.copy_(self._right_input_batch)
        if self._fwd_plan is None:
            from madnet_hip.plan import Recorder
            r = Recorder()
            eng.record_forward(r)
            for name, buf in self._disp_bufs:
                eng.record_make_disp(r, name, buf)
            self._fwd_plan = r.compile()
        stream = torch.cuda.current_stream().cuda_stream if eng.left.is_cuda else 0
        self._fwd_plan.run(self._lib, stream)
        return self._disparities

    def variable(self, name):
        return self._variables[name]
