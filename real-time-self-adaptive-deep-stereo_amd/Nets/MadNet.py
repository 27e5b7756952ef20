"""MADNet on MI355X: same class name, kwargs, defaults, layer keys and variable names as the
reference graph builder (Nets/MadNet.py:8-436), executed by madnet_hip.engine.MadNetEngine (hand
written HIP kernels behind the C-ABI of include/madnet_hip.h) instead of TF1 ops."""
import torch

from Nets import Stereo_net
from madnet_hip import _ffi, engine as E, synthetic


class MadNet(Stereo_net.StereoNet):
    _valid_args = [
        ("left_img", "tensor [B,H,W,3] for the left image batch"),
        ("right_img", "tensor [B,H,W,3] for the right image batch"),
        ("warping", "flag to enable warping"),
        ("context_net", "flag to enable context_net"),
        ("radius_d", "size f the patch using for correlation"),
        ("stride", "stride used for correlation"),
        ("bulkhead", "flag to stop gradient propagation among different resolution"),
        ("weights", "(new) dict {TF variable name: HWIO array}; default Xavier like the reference initializer"),
    ] + Stereo_net.StereoNet._valid_args
    _netName = "MADNet"

    def __init__(self, **kwargs):
        super(MadNet, self).__init__(**kwargs)

    def _validate_args(self, args):
        args = super(MadNet, self)._validate_args(args)
        if ('left_img' not in args) or ('right_img' not in args):
            raise Exception('Missing input op for left and right images')
        if 'warping' not in args:
            print('WARNING: warping flag not setted, setting default True value')
            args['warping'] = True
        if 'context_net' not in args:
            print('WARNING: context_net flag not setted, setting default True value')
            args['context_net'] = True
        if 'radius_d' not in args:
            print('WARNING: radius_d not setted, setting default value 2')
            args['radius_d'] = 2
        if 'stride' not in args:
            print('WARNING: stride not setted, setting default value 1')
            args['stride'] = 1
        if 'bulkhead' not in args:
            args['bulkhead'] = False
        if not args['context_net']:
            # the reference's context_net=False branch references a commented-out variable and
            # raises NameError (Nets/MadNet.py:350-360)
            raise NotImplementedError("context_net=False is broken in the reference (NameError); not supported")
        return args

    def _preprocess_inputs(self, args):
        l, r = args['left_img'], args['right_img']
        if not (isinstance(l, torch.Tensor) and isinstance(r, torch.Tensor)):
            raise Exception('left_img / right_img must be torch tensors [B,H,W,3]')
        if l.dim() != 4 or l.shape[-1] != 3 or l.shape != r.shape:
            raise Exception('left_img / right_img must both be [B,H,W,3]')
        self._left_input_batch, self._right_input_batch = l, r
        self._restore_shape = (int(l.shape[1]), int(l.shape[2]))
        self._bulkhead = bool(args['bulkhead'])

    def _build_network(self, args):
        l = self._left_input_batch
        B, H, W = int(l.shape[0]), int(l.shape[1]), int(l.shape[2])
        lib = args.get('_lib') or _ffi.lib()          # fails loudly without the HIP library / a GPU
        dev = args.get('_device') or l.device
        weights = args.get('weights')
        if weights is None:
            shapes = dict(E.madnet_manifest(args['radius_d'], args['stride']))
            weights = synthetic.xavier_weights(shapes, seed=0)
        self.engine = eng = E.MadNetEngine(lib, H, W, B=B, device=dev, radius_d=args['radius_d'],
                                           stride=args['stride'], warping=args['warping'], weights=weights,
                                           precision=args.get('precision', 'fp32'))   # extra kwarg: 'bf16' = MFMA throughput mode
        self._lib = lib
        P = eng.params

        def var_pair(base, wname='weights', bname='biases'):
            return [Stereo_net.Variable(base + '/' + wname, P.tensor(base + '/' + wname)),
                    Stereo_net.Variable(base + '/' + bname, P.tensor(base + '/' + bname))]

        self._variables = {}
        # pyramid towers: shared variables; the reused (right) tower's keys map to [] exactly like
        # the reference's scope-prefix lookup does
        for i in range(1, 13):
            vs = var_pair(E.pyr_name(i))
            for v in vs:
                self._variables[v.op_name] = v
            self._add_to_layers('left/conv%d' % i, eng.F[i][:B], vs)
        for i in range(1, 13):
            self._add_to_layers('right/conv%d' % i, eng.F[i][B:], [])
        for k in E.LEVELS:
            for j in range(1, 7):
                vs = var_pair(E.est_name(k, j))
                for v in vs:
                    self._variables[v.op_name] = v
                t = eng.V[k][..., None] if j == 6 else eng.E[k][j - 1]
                self._add_to_layers('fgc-volume-filtering-%d/disp%d' % (k, j), t, vs)
            if k != 2:
                self._disparities.append(eng.disp_k[k][..., None])
        for j in range(1, 8):
            vs = var_pair(E.ctx_name(j))
            for v in vs:
                self._variables[v.op_name] = v
            # context7's own output is fused into final_disp (accumulating epilogue)
            t = eng.final[..., None] if j == 7 else eng.Cx[j - 1]
            self._add_to_layers('context%d' % j, t, vs)
        self._add_to_layers('final_disp', eng.final[..., None], list(self._variables.values()))
        self._disparities.append(eng.disp_k[2][..., None])
        self._layers['rescaled_prediction'] = eng.pred[..., None]
        self._disparities.append(self._layers['rescaled_prediction'])
        self._fwd_plan = None

    # ------------------------------------------------------------------ execution (replaces sess.run)
    def run(self):
        """One forward pass on the current contents of left_img/right_img: refreshes every layer and
        all six disparities (the analogue of sess.run(net.get_disparities()))."""
        eng = self.engine
        eng.left.copy_(self._left_input_batch)
        eng.right.copy_(self._right_input_batch)
        if self._fwd_plan is None:
            from madnet_hip.plan import Recorder
            r = Recorder()
            eng.record_forward(r, make_disps=E.LEVELS)
            self._fwd_plan = r.compile()
        stream = torch.cuda.current_stream().cuda_stream if eng.left.is_cuda else 0
        self._fwd_plan.run(self._lib, stream)
        return self._disparities

    def variable(self, name):
        return self._variables[name]
