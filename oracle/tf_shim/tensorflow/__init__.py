"""TEST INFRASTRUCTURE (part of the CPU oracle, never imported by the product): an EAGER stand-in for the ~70 `tensorflow` 1.x symbols the
reference's own graph code touches, so that /root/reference/Nets/{Stereo_net,sharedLayers,MadNet,DispNet}.py, Losses/loss_factory.py and
Data_utils/preprocessing.py can be imported and EXECUTED as their authors wrote them (TensorFlow itself cannot be installed here).

What this pins: the graph WIRING -- concat order, the x20 / 2^k scales, relu-before/after-resize (MadNet.py:69 vs :362),
which u_k is stop_gradient-ed, the layer -> variable map of StereoNet._add_to_layers, the loss composition -- comes from the reference source,
not from a hand restatement.  What it cannot pin: the arithmetic INSIDE TensorFlow's library kernels (conv2d, resize_images, ...): every such
symbol delegates to the documented-semantics restatement in oracle/tf_ops.py, exactly what oracle/madnet.py calls too.

Only oracle/ref_graph.py puts this directory on sys.path (in a subprocess of its own: the module names Nets / Losses / Data_utils of the
reference collide with the product's API mirror).  Tensors are torch CPU tensors wrapped in `Tensor`; gradients come from torch autograd, with
TF's conventions where they differ (tf.maximum routes ties to its first argument)."""
import builtins as _builtins
import os as _os
import re as _re
import sys as _sys

import numpy as _np
import torch as _torch
import torch.nn.functional as _F

_here = _os.path.dirname(_os.path.abspath(__file__))
_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_here)))
if _root not in _sys.path:
    _sys.path.append(_root)
from oracle import tf_ops as _T      # noqa: E402  (the TF 1.12 kernel restatements)

__version__ = "1.12-shim"

# ---- dtypes --------------------------------------------------------------------------------------------------------
float32 = _torch.float32
float64 = _torch.float64
int32 = _torch.int32
int64 = _torch.int64
bool = _torch.bool          # noqa: A001  (tf.bool)
_DT = {"float32": float32, "float64": float64, "int32": int32, "int64": int64}


def _dtype(d):
    return _DT[d] if isinstance(d, str) else d


# ---- scopes: tf.variable_scope opens a variable scope (names of variables) AND a uniquified name scope (names of ops) --------------------
class _State(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.var_scope = []          # [(name, reuse)]
        self.name_scope = []         # uniquified op-name components
        self.used_names = {}         # parent name scope -> {child name: count}
        self.variables = {}          # full name (without ':0') -> Variable, creation order
        self.init_values = {}        # full name -> array-like given by the driver (the step's weights)
        self.collections = {}


_S = _State()


def reset_default_graph():
    _S.reset()


def _unique(parent, name):
    used = _S.used_names.setdefault(parent, {})
    n = used.get(name, 0)
    used[name] = n + 1
    return name if n == 0 else "%s_%d" % (name, n)


class _Scope(object):
    def __init__(self, name, reuse=None, variable=True):
        self.name, self.reuse, self.variable = name, reuse, variable

    def __enter__(self):
        parent = "/".join(_S.name_scope)
        _S.name_scope.append(_unique(parent, self.name))
        if self.variable:
            inherited = _S.var_scope[-1][1] if _S.var_scope else False
            _S.var_scope.append((self.name, builtins_bool(self.reuse) or inherited))
        return self

    def __exit__(self, *a):
        _S.name_scope.pop()
        if self.variable:
            _S.var_scope.pop()
        return False


builtins_bool = _builtins.bool


def variable_scope(name, reuse=None, **kw):
    return _Scope(name, reuse, True)


def name_scope(name, *a, **kw):
    return _Scope(name, None, False)


def _op_name(kind):
    return "/".join(_S.name_scope + [kind]) + ":0"


# ---- tensors ---------------------------------------------------------------------------------------------------------
class Dimension(object):
    def __init__(self, v):
        self.value = v

    def __int__(self):
        return int(self.value)

    __index__ = __int__

    def __eq__(self, o):
        return self.value == (o.value if isinstance(o, Dimension) else o)

    def __hash__(self):
        return hash(self.value)

    def __repr__(self):
        return "Dimension(%r)" % (self.value,)


class TensorShape(object):
    def __init__(self, dims):
        self._dims = [int(d) for d in dims]

    def as_list(self):
        return list(self._dims)

    def __getitem__(self, i):
        if isinstance(i, _builtins.slice):
            return TensorShape(self._dims[i])
        return Dimension(self._dims[i])

    def __len__(self):
        return len(self._dims)

    def __iter__(self):
        return iter(Dimension(d) for d in self._dims)

    def __repr__(self):
        return "TensorShape(%r)" % (self._dims,)

    __str__ = __repr__


def _raw(x, like=None):
    """torch tensor of anything the reference passes where TF takes a tensor"""
    if isinstance(x, Tensor):
        return x.t
    if isinstance(x, _torch.Tensor):
        return x
    if isinstance(x, _np.ndarray):
        # TF converts a float64 numpy array that meets float32 tensors in one op to float32
        t = _torch.from_numpy(_np.ascontiguousarray(x))
        if like is not None and t.is_floating_point() and like.is_floating_point():
            t = t.to(like.dtype)
        return t
    if isinstance(x, (list, tuple)):
        if like is not None:
            return _torch.tensor(x, dtype=like.dtype)
        return _torch.tensor(x)
    if like is not None:
        return _torch.tensor(x, dtype=like.dtype)
    return _torch.tensor(x)


class Tensor(object):
    """an eager value with the handful of tf.Tensor methods the reference calls"""
    __array_priority__ = 1000

    def __init__(self, t, kind="Op", name=None):
        self.t = t
        self.name = name if name is not None else _op_name(kind)

    # -- shape / dtype surface
    def get_shape(self):
        return TensorShape(self.t.shape)

    @property
    def shape(self):
        return TensorShape(self.t.shape)

    @property
    def dtype(self):
        return self.t.dtype

    def set_shape(self, shape):
        for have, want in zip(self.t.shape, shape):
            assert want is None or int(want) == have, ("set_shape mismatch", tuple(self.t.shape), shape)

    # -- arithmetic (python numbers take the tensor's dtype, like TF's constant conversion)
    def _bin(self, o, fn, kind, swap=False):
        b = _raw(o, like=self.t)
        a = self.t
        if b.dtype != a.dtype and b.is_floating_point() and a.is_floating_point():
            raise TypeError("dtype mismatch %s vs %s in %s (TF would refuse it too)" % (a.dtype, b.dtype, kind))
        return Tensor(fn(b, a) if swap else fn(a, b), kind)

    def __add__(self, o): return self._bin(o, _torch.add, "add")
    def __radd__(self, o): return self._bin(o, _torch.add, "add", True)
    def __sub__(self, o): return self._bin(o, _torch.sub, "sub")
    def __rsub__(self, o): return self._bin(o, _torch.sub, "sub", True)
    def __mul__(self, o): return self._bin(o, _torch.mul, "mul")
    def __rmul__(self, o): return self._bin(o, _torch.mul, "mul", True)
    def __truediv__(self, o): return self._bin(o, _torch.div, "truediv")
    def __rtruediv__(self, o): return self._bin(o, _torch.div, "truediv", True)
    def __floordiv__(self, o): return self._bin(o, lambda a, b: _torch.div(a, b, rounding_mode="floor"), "floordiv")
    def __mod__(self, o): return self._bin(o, _torch.remainder, "mod")
    def __neg__(self): return Tensor(-self.t, "Neg")

    def __pow__(self, o):
        return Tensor(self.t * self.t if o == 2 else _torch.pow(self.t, o), "pow")

    def __getitem__(self, idx):
        return Tensor(self.t[idx], "strided_slice")

    def __gt__(self, o): return self._bin(o, _torch.gt, "Greater")
    def __lt__(self, o): return self._bin(o, _torch.lt, "Less")

    def __repr__(self):
        return "<shim tf.Tensor %s shape=%s dtype=%s>" % (self.name, tuple(self.t.shape), self.t.dtype)


class Variable(Tensor):
    def __init__(self, t, name):
        Tensor.__init__(self, t, name=name + ":0")
        self.var_name = name


def _wrap(t, kind):
    return Tensor(t, kind)


def _is_py(x):
    return isinstance(x, (int, float, _builtins.bool, _np.integer, _np.floating))


# ---- variables / collections ------------------------------------------------------------------------------------------------
class GraphKeys(object):
    TRAINABLE_VARIABLES = "trainable_variables"
    WEIGHTS = "weights"


class _Initializer(object):
    def __init__(self, kind, value=0.0):
        self.kind, self.value = kind, value


def constant_initializer(value=0.0, **kw):
    return _Initializer("const", value)


class _Layers(object):
    @staticmethod
    def xavier_initializer(**kw):
        return _Initializer("xavier")


class _Contrib(object):
    layers = _Layers()


contrib = _Contrib()


def set_initial_values(values):
    """driver hook (oracle/ref_graph.py): {variable name: array} used instead of the initializers (the step's weights)"""
    _S.init_values = dict(values)


def get_variable(name, shape=None, initializer=None, dtype=float32, **kw):
    scope = "/".join(n for n, _ in _S.var_scope)
    full = (scope + "/" if scope else "") + name
    reuse = _S.var_scope[-1][1] if _S.var_scope else False
    if full in _S.variables:
        if not reuse:
            raise ValueError("Variable %s already exists, disallowed (no reuse=True)" % full)
        return _S.variables[full]
    if reuse:
        raise ValueError("Variable %s does not exist (reuse=True)" % full)
    shape = [int(s) for s in (shape if isinstance(shape, (list, tuple)) else [shape])]
    if full in _S.init_values:
        t = _torch.as_tensor(_np.asarray(_S.init_values[full]), dtype=dtype).reshape(shape).clone()
    elif initializer is not None and initializer.kind == "const":
        t = _torch.full(shape, float(initializer.value), dtype=dtype)
    else:   # xavier uniform: +-sqrt(6 / (fan_in + fan_out)), fans = receptive field x channels
        rf = int(_np.prod(shape[:-2])) if len(shape) > 2 else 1
        lim = float(_np.sqrt(6.0 / (rf * shape[-2] + rf * shape[-1]))) if len(shape) >= 2 else 0.0
        t = (_torch.rand(shape, dtype=dtype) * 2 - 1) * lim
    t.requires_grad_(True)
    v = Variable(t, full)
    _S.variables[full] = v
    _S.collections.setdefault(GraphKeys.TRAINABLE_VARIABLES, []).append(v)
    return v


def get_collection(key, scope=None):
    items = list(_S.collections.get(key, []))
    if scope is None:
        return items
    return [v for v in items if _re.match(scope, v.name)]        # tf.get_collection: re.match(scope, item.name)


def add_to_collection(key, value):
    _S.collections.setdefault(key, []).append(value)


def trainable_variables():
    return get_collection(GraphKeys.TRAINABLE_VARIABLES)


def placeholder(dtype, shape=None, name=None):
    raise NotImplementedError("tf.placeholder: the shim runs the graph eagerly (split layers are not used on the adaptation path)")


# ---- structural ops -------------------------------------------------------------------------------------------------------------
def shape(x):
    """dynamic shape: python ints (every size is known eagerly); slices of it stay lists"""
    return list(_raw(x).shape)


def cast(x, dtype, name=None):
    dtype = _dtype(dtype)
    if _is_py(x):
        return Tensor(_torch.tensor(x, dtype=dtype), "Cast")
    t = _raw(x)
    if not dtype.is_floating_point and t.is_floating_point():
        return Tensor(t.detach().to(dtype), "Cast")        # float -> int truncates; no gradient
    return Tensor(t.to(dtype), "Cast")


def to_int32(x):
    return cast(x, int32)


def constant(value, dtype=None, shape=None, name=None):
    t = _raw(value)
    if dtype is not None:
        t = t.to(_dtype(dtype))
    elif t.dtype == _torch.float64:
        t = t.to(float32)
    return Tensor(t, "Const")


def zeros(shape, dtype=float32, name=None):
    return Tensor(_torch.zeros([int(s) for s in shape], dtype=_dtype(dtype)), "zeros")


def ones(shape, dtype=float32, name=None):
    return Tensor(_torch.ones([int(s) for s in shape], dtype=_dtype(dtype)), "ones")


def zeros_like(x, dtype=None):
    t = _raw(x)
    return Tensor(_torch.zeros_like(t, dtype=_dtype(dtype) if dtype is not None else t.dtype), "zeros_like")


def ones_like(x, dtype=None):
    t = _raw(x)
    return Tensor(_torch.ones_like(t, dtype=_dtype(dtype) if dtype is not None else t.dtype), "ones_like")


def concat(values, axis, name=None):
    ref = next((v.t for v in values if isinstance(v, Tensor)), None)
    return Tensor(_torch.cat([_raw(v, like=ref) for v in values], dim=axis), "concat")


def stack(values, axis=0, name=None):
    if all(_is_py(v) for v in values):
        return [int(v) for v in values]                     # a shape built from python ints stays a shape
    ref = next((v.t for v in values if isinstance(v, Tensor)), None)
    return Tensor(_torch.stack([_raw(v, like=ref) for v in values], dim=axis), "stack")


def split(value, num_or_size_splits, axis=0):
    t = _raw(value)
    if isinstance(num_or_size_splits, int):
        return [Tensor(p, "split") for p in _torch.chunk(t, num_or_size_splits, dim=axis)]
    return [Tensor(p, "split") for p in _torch.split(t, list(num_or_size_splits), dim=axis)]


def reshape(x, shape, name=None):
    return Tensor(_raw(x).reshape([int(s) for s in shape]), "Reshape")


def expand_dims(x, axis):
    return Tensor(_raw(x).unsqueeze(axis), "ExpandDims")


def transpose(x, perm=None):
    t = _raw(x)
    return Tensor(t.permute(*perm) if perm is not None else t.t(), "transpose")


def tile(x, multiples):
    return Tensor(_raw(x).repeat(*[int(m) for m in multiples]), "Tile")


def slice(x, begin, size):      # noqa: A001
    t = _raw(x)
    idx = tuple(_builtins.slice(int(b), None if int(s) == -1 else int(b) + int(s)) for b, s in zip(begin, size))
    return Tensor(t[idx], "Slice")


def pad(x, paddings, mode="CONSTANT", name=None):
    t = _raw(x)
    p = [[int(a), int(b)] for a, b in paddings]
    assert t.dim() == 4 and p[0] == [0, 0] and p[3] == [0, 0], "shim tf.pad: NHWC, spatial padding only"
    (pt, pb), (pl, pr) = p[1], p[2]
    if pt == pb == pl == pr == 0:
        return Tensor(t, "Pad")
    n = t.permute(0, 3, 1, 2)
    n = _F.pad(n, (pl, pr, pt, pb), mode="reflect") if mode.upper() == "REFLECT" else _F.pad(n, (pl, pr, pt, pb))      # 
    return Tensor(n.permute(0, 2, 3, 1), "MirrorPad" if mode.upper() == "REFLECT" else "Pad")


def range(*a, **kw):      # noqa: A001
    dtype = _dtype(kw.get("dtype", None)) if kw.get("dtype", None) is not None else None
    vals = [float(_raw(v)) if not _is_py(v) else v for v in a]
    t = _torch.arange(*vals) if dtype is None else _torch.arange(*vals, dtype=dtype)
    if dtype is None and all(isinstance(v, (int, _np.integer)) for v in vals):
        t = t.to(int32)
    return Tensor(t, "range")


def gather(params, indices, axis=0):
    p, i = _raw(params), _raw(indices).to(_torch.int64)
    assert axis == 0
    return Tensor(p.index_select(0, i.reshape(-1)).reshape(tuple(i.shape) + tuple(p.shape[1:])), "GatherV2")


def gather_nd(params, indices):
    p, i = _raw(params), _raw(indices).to(_torch.int64)
    k = i.shape[-1]
    flat = i.reshape(-1, k)
    out = p[tuple(flat[:, d] for d in _builtins.range(k))]
    return Tensor(out.reshape(tuple(i.shape[:-1]) + tuple(p.shape[k:])), "GatherNd")


def stop_gradient(x):
    return Tensor(_raw(x).detach(), "StopGradient")


def add_n(xs):
    acc = _raw(xs[0])
    for v in xs[1:]:
        acc = acc + _raw(v)
    return Tensor(acc, "AddN")


def matmul(a, b):
    return Tensor(_raw(a) @ _raw(b), "MatMul")


# ---- element-wise --------------------------------------------------------------------------------------------------------------
def maximum(a, b):
    """tf.maximum; MaximumGrad sends the gradient to the FIRST argument where a >= b (ties included): for the reference's leaky
    tf.maximum(alpha * x, x) that is slope alpha at x == 0"""
    ta = _raw(a)
    tb = _raw(b, like=ta)
    return Tensor(_torch.where(ta >= tb, ta, tb), "Maximum")


def abs(x): return Tensor(_raw(x).abs(), "Abs")             # noqa: A001
def square(x): return Tensor(_raw(x) * _raw(x), "Square")
def sqrt(x): return Tensor(_raw(x).sqrt(), "Sqrt")
def exp(x): return Tensor(_raw(x).exp(), "Exp")
def sign(x): return Tensor(_raw(x).sign(), "Sign")
def sigmoid(x): return Tensor(_torch.sigmoid(_raw(x)), "Sigmoid")
def floor(x): return Tensor(_torch.floor(_raw(x)), "Floor")
def round(x): return Tensor(_torch.round(_raw(x)), "Round")     # noqa: A001


def floordiv(a, b):
    if _is_py(a) and _is_py(b):
        return a // b
    return Tensor(_torch.div(_raw(a), _raw(b), rounding_mode="floor"), "FloorDiv")


def clip_by_value(x, lo, hi):
    """gradient passes where lo <= x <= hi (torch.clamp's convention = TF's)"""
    lo = float(_raw(lo)) if not _is_py(lo) else lo
    hi = float(_raw(hi)) if not _is_py(hi) else hi
    return Tensor(_torch.clamp(_raw(x), lo, hi), "clip_by_value")


def _cmp(a, b, fn, kind):
    if _is_py(a) and _is_py(b):
        return fn(_torch.tensor(a), _torch.tensor(b)).item()
    ta = _raw(a) if not _is_py(a) else None
    tb = _raw(b, like=ta) if ta is not None else _raw(b)
    if ta is None:
        ta = _raw(a, like=tb)
    return Tensor(fn(ta, tb), kind)


def equal(a, b): return _cmp(a, b, _torch.eq, "Equal")
def greater(a, b): return _cmp(a, b, _torch.gt, "Greater")
def greater_equal(a, b): return _cmp(a, b, _torch.ge, "GreaterEqual")
def less_equal(a, b): return _cmp(a, b, _torch.le, "LessEqual")
def logical_or(a, b): return Tensor(_raw(a) | _raw(b), "LogicalOr")


def where(condition, x=None, y=None):
    if isinstance(condition, (_builtins.bool, _np.bool_)):
        return x if condition else y                        # pad_image(dynamic=True) selects between python ints
    c = _raw(condition)
    tx = _raw(x)
    return Tensor(_torch.where(c, tx, _raw(y, like=tx)), "Select")


def _axes(axis):
    return None if axis is None else (tuple(axis) if isinstance(axis, (list, tuple)) else axis)


def reduce_sum(x, axis=None, keepdims=False, **kw):
    if isinstance(x, (list, tuple)):
        x = stack(list(x)) if not all(_is_py(v) for v in x) else _torch.tensor(x)
    t = _raw(x)
    return Tensor(t.sum() if axis is None else t.sum(dim=_axes(axis), keepdim=keepdims), "Sum")


def reduce_mean(x, axis=None, keepdims=False, **kw):
    t = _raw(x)
    return Tensor(t.mean() if axis is None else t.mean(dim=_axes(axis), keepdim=keepdims), "Mean")


def reduce_max(x, axis=None, keepdims=False, **kw):
    t = _raw(x)
    return Tensor(t.max() if axis is None else t.amax(dim=_axes(axis), keepdim=keepdims), "Max")


def reduce_min(x, axis=None, keepdims=False, **kw):
    t = _raw(x)
    return Tensor(t.min() if axis is None else t.amin(dim=_axes(axis), keepdim=keepdims), "Min")


def Print(x, *a, **kw):
    return x


def cond(pred, true_fn, false_fn):
    return true_fn() if builtins_bool(_raw(pred)) else false_fn()


# ---- tf.nn: the TF 1.12 library kernels -> oracle/tf_ops.py -------------------------------------------------------------------------
class _NN(object):
    @staticmethod
    def conv2d(x, W, strides, padding, name=None):
        assert padding == "SAME" and strides[0] == 1 and strides[3] == 1 and strides[1] == strides[2]
        return Tensor(_T.conv2d(_raw(x), _raw(W), None, stride=int(strides[1]), dilation=1, alpha=1.0), "Conv2D")

    @staticmethod
    def atrous_conv2d(x, W, rate, padding, name=None):
        assert padding == "SAME"
        return Tensor(_T.conv2d(_raw(x), _raw(W), None, stride=1, dilation=int(rate), alpha=1.0), "convolution")

    @staticmethod
    def conv2d_transpose(x, W, output_shape, strides, padding="SAME", name=None):
        assert padding == "SAME" and strides[1] == strides[2]
        tx, tw = _raw(x), _raw(W)
        y = _T.conv2d_transpose(tx, tw, _torch.zeros(tw.shape[2], dtype=tx.dtype), stride=int(strides[1]), alpha=1.0)
        assert [int(s) for s in output_shape] == list(y.shape), (output_shape, tuple(y.shape))
        return Tensor(y, "conv2d_transpose")

    @staticmethod
    def bias_add(x, b, name=None):
        return Tensor(_raw(x) + _raw(b), "BiasAdd")

    @staticmethod
    def relu(x, name=None):
        return Tensor(_torch.relu(_raw(x)), "Relu")

    @staticmethod
    def avg_pool(x, ksize, strides, padding, name=None):
        assert list(ksize) == [1, 3, 3, 1] and list(strides) == [1, 1, 1, 1] and padding == "VALID"
        return Tensor(_T._avg_pool3_valid(_raw(x)), "AvgPool")

    @staticmethod
    def l2_normalize(x, axis=None, epsilon=1e-12, **kw):
        t = _raw(x)
        return Tensor(t * _torch.rsqrt(_torch.clamp((t * t).sum(dim=axis, keepdim=True), min=epsilon)), "l2_normalize")


nn = _NN()


# ---- tf.image -----------------------------------------------------------------------------------------------------------------------
class _ResizeMethod(object):
    BILINEAR = 0


class _Image(object):
    ResizeMethod = _ResizeMethod

    @staticmethod
    def resize_images(images, size, method=0, align_corners=False):
        assert method == 0 and not align_corners
        h, w = (int(_raw(s)) if not _is_py(s) else int(s) for s in size)
        return Tensor(_T.resize_bilinear(_raw(images), h, w), "ResizeBilinear")

    @staticmethod
    def resize_image_with_crop_or_pad(image, target_height, target_width):
        t = _raw(image)
        th, tw = int(target_height), int(target_width)
        assert th <= t.shape[1] and tw <= t.shape[2], "shim: crop branch only (the adaptation path never pads here)"
        return Tensor(_T.center_crop(t, th, tw), "crop_to_bounding_box")


image = _Image()


# ---- what the graph code only mentions -------------------------------------------------------------------------------------------------
class _Summary(object):
    @staticmethod
    def scalar(*a, **kw):
        return None


summary = _Summary()


def load_op_library(path):
    raise NotImplementedError("the CUDA correlation op is not loaded on this path (sharedLayers.MODE == 'TF')")


def RegisterGradient(name):
    return lambda fn: fn


class _LayersAPI(object):
    @staticmethod
    def batch_normalization(*a, **kw):
        raise NotImplementedError("batch_norm=False on the adaptation path")


layers = _LayersAPI()


def random_uniform(*a, **kw):
    raise NotImplementedError("augmentation is not on the adaptation path")
