"""Operator library with the reference's signatures (Nets/sharedLayers.py:23-92): correlation,
conv2d, dilated_conv2d, conv2d_transpose -- eager torch tensors in/out (NHWC float32 on the GPU),
hand-written HIP kernels underneath, gradients registered through torch.autograd.Function the way
the reference registers ShiftCorrGrad with @tf.RegisterGradient (sharedLayers.py:15-17).

TF1 `variable_scope` / `get_variable` are replaced by a small explicit VariableStore (names keep
the TF convention '<scope>/<name>/<wName>').  `activation` accepts the reference's lambdas; a
`Leaky(alpha)` object (or the default) is fused into the conv epilogue, any other callable is
applied afterwards with torch."""
import numpy as np
import torch

from madnet_hip import _ffi, ops

MODE = 'HIP'          # the reference's default is 'TF' (pure-TF slices); here the native op IS the path


class Leaky(object):
    """tf.maximum(alpha*x, x); fusable into the conv epilogue."""

    def __init__(self, alpha):
        self.alpha = float(alpha)

    def __call__(self, x):
        return torch.where(x > 0, x, self.alpha * x)


class VariableStore(object):
    """name -> torch parameter; Xavier-uniform weights, zero biases (sharedLayers.py:4-5)."""

    def __init__(self, device='cuda', seed=0):
        self.vars = {}
        self.scope = []
        self.device = device
        self.rng = np.random.default_rng(seed)

    def get_variable(self, name, shape, reuse=False):
        full = '/'.join(self.scope + [name])
        if full not in self.vars:
            if reuse:
                raise ValueError('Variable %s does not exist' % full)
            if len(shape) == 1:
                v = np.zeros(shape, np.float32)
            else:
                kh, kw, a, b = shape
                lim = np.sqrt(6.0 / (kh * kw * a + kh * kw * b))
                v = self.rng.uniform(-lim, lim, size=shape).astype(np.float32)
            self.vars[full] = torch.from_numpy(v).to(self.device).requires_grad_(True)
        return self.vars[full]


_default_store = None


def default_store():
    global _default_store
    if _default_store is None:
        _default_store = VariableStore()
    return _default_store


def _lib():
    return _ffi.lib()


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0


def _alpha_of(activation):
    if isinstance(activation, Leaky):
        return activation.alpha, None
    if activation is None:
        return 1.0, None
    return 1.0, activation


class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, dil, alpha, transpose):
        x = x.contiguous(); w = w.contiguous(); b = b.contiguous()
        lib = _lib()
        B, H, W, _ = x.shape
        if transpose:
            cout = w.shape[2]
            out = torch.empty(B, H * stride, W * stride, cout, device=x.device)
            ops.conv2d_transpose_fwd(lib, ops.view(x), w, b, ops.view(out), stride=stride, alpha=alpha, stream=_stream(x))
        else:
            Ho, Wo, _, _ = ops.conv_geometry(H, W, w.shape[0], w.shape[1], stride, dil)
            out = torch.empty(B, Ho, Wo, w.shape[3], device=x.device)
            ops.conv2d_fwd(lib, ops.view(x), w, b, ops.view(out), stride=stride, dil=dil, alpha=alpha, stream=_stream(x))
        ctx.save_for_backward(x, w, out)
        ctx.cfg = (stride, dil, alpha, transpose)
        return out

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        stride, dil, alpha, transpose = ctx.cfg
        lib = _lib()
        s = _stream(x)
        dz = gy.contiguous().clone()
        if alpha != 1.0:
            ops.leaky_bwd(lib, ops.view(dz), ops.view(y), alpha, stream=s)
        dw = torch.zeros_like(w); db = torch.zeros(dz.shape[-1], device=x.device)
        dx = torch.empty_like(x)
        if transpose:
            # y = conv2d_transpose(x, w[kh,kw,Cout,Cin]) is the input-gradient of conv(w as HWIO[.,.,Cout,Cin]):
            # dx = that conv applied to dz ; dw = its filter gradient with (input=dz, output-grad=x)
            ops.conv2d_fwd(lib, ops.view(dz), w, torch.zeros(x.shape[-1], device=x.device), ops.view(dx), stride=stride, stream=s)
            ops.conv2d_wgrad(lib, ops.view(dz), ops.view(x), dw, None, stride=stride, stream=s)
            ops.bias_grad(lib, ops.view(dz), db, stream=s)             # BiasAddGrad (db was zeroed above)
        else:
            ops.conv2d_dgrad(lib, ops.view(dz), w, ops.view(dx), stride=stride, dil=dil, stream=s)
            ops.conv2d_wgrad(lib, ops.view(x), ops.view(dz), dw, db, stride=stride, dil=dil, stream=s)
        return dx, dw, db, None, None, None, None


class _CorrFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, max_disp, stride):
        x = x.contiguous(); y = y.contiguous()
        B, H, W, _ = x.shape
        D = 2 * max_disp // stride + 1
        out = torch.empty(B, H, W, D, device=x.device)
        ops.corr_fwd(_lib(), ops.view(x), ops.view(y), ops.view(out), max_disp, stride, stream=_stream(x))
        ctx.save_for_backward(x, y)
        ctx.cfg = (max_disp, stride)
        return out

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        max_disp, stride = ctx.cfg
        dx = torch.empty_like(x); dy = torch.empty_like(y)
        ops.corr_bwd(_lib(), ops.view(g.contiguous()), ops.view(x), ops.view(y), ops.view(dx), ops.view(dy),
                     max_disp, stride, stream=_stream(x))
        return dx, dy, None, None


def correlation(x, y, max_disp, name='corr', mode=MODE, stride=1):
    """corr[b,h,w,j] = mean_c x[b,h,w,c] * y[b,h,w+i_j,c], i_j = -max_disp + j*stride (zero outside)."""
    if mode not in ('TF', 'HIP', 'CUDA'):
        raise Exception("correlation mode must be 'TF', 'HIP' or 'CUDA' (reference sharedLayers.py:23-51), got %r" % (mode,))
    # 'TF' (the reference's default: pad + slice + reduce_mean in TensorFlow, sharedLayers.py:41-51) and 'CUDA' (its native ShiftCorr op, :30-40) name two
    # implementations of ONE formula; here both run the HIP op, which computes that formula (tests/test_ref_pin.py: three-way against the reference's own
    # kernel and the TF restatement).  No TensorFlow-style slicing path exists in the product -- the restatement of it is the parity oracle (oracle/tf_ops.py).
    return _CorrFn.apply(x, y, int(max_disp), int(stride))


def conv2d(x, kernel_shape, strides=1, activation=Leaky(0.1), padding='SAME', name='conv', reuse=False,
           wName='weights', bName='bias', batch_norm=False, training=False, store=None):
    if padding != 'SAME' or batch_norm:
        raise NotImplementedError("only padding='SAME', batch_norm=False are on the hot path")
    st = store or default_store()
    st.scope.append(name)
    try:
        W = st.get_variable(wName, kernel_shape, reuse)
        b = st.get_variable(bName, [kernel_shape[3]], reuse)
    finally:
        st.scope.pop()
    alpha, post = _alpha_of(activation)
    y = _ConvFn.apply(x, W, b, int(strides), 1, alpha, False)
    return post(y) if post is not None else y


def dilated_conv2d(x, kernel_shape, rate=1, activation=Leaky(0.1), padding='SAME', name='dilated_conv', reuse=False,
                   wName='weights', bName='biases', batch_norm=False, training=False, store=None):
    if padding != 'SAME' or batch_norm:
        raise NotImplementedError("only padding='SAME', batch_norm=False are on the hot path")
    st = store or default_store()
    st.scope.append(name)
    try:
        W = st.get_variable(wName, kernel_shape, reuse)
        b = st.get_variable(bName, [kernel_shape[3]], reuse)
    finally:
        st.scope.pop()
    alpha, post = _alpha_of(activation)
    y = _ConvFn.apply(x, W, b, 1, int(rate), alpha, False)
    return post(y) if post is not None else y


def conv2d_transpose(x, kernel_shape, strides=1, activation=Leaky(0.1), name='conv', reuse=False,
                     wName='weights', bName='bias', batch_norm=False, training=False, store=None):
    if batch_norm:
        raise NotImplementedError("batch_norm is never enabled by a reference caller")
    st = store or default_store()
    st.scope.append(name)
    try:
        W = st.get_variable(wName, kernel_shape, reuse)
        b = st.get_variable(bName, [kernel_shape[2]], reuse)
    finally:
        st.scope.pop()
    alpha, post = _alpha_of(activation)
    y = _ConvFn.apply(x, W, b, int(strides), 1, alpha, True)
    return post(y) if post is not None else y
