"""StereoNet base class: kwargs validation, ordered layer registry, layer -> variables registry and
the public getters of the reference (Nets/Stereo_net.py:6-222), on top of the MI355X engine instead
of a TF1 graph.

Differences forced by dropping TF1 (documented in INTEGRATION.md):
  * `left_img` / `right_img` are torch tensors [B,H,W,3] (float32, 0..255) instead of TF ops; they act
    like the reference's input ops: `run()` reads their CURRENT contents;
  * layers / disparities are torch tensors backed by the engine's static HBM buffers, refreshed by
    every `run()` (the analogue of sess.run on the layer ops);
  * the split/placeholder mechanism (Stereo_net.py:81-97) is never enabled by any reference driver
    (split_layers=[None], sequence=True): API kept, no placeholders are ever created.
"""
from collections import OrderedDict


class Variable(object):
    """A trainable variable: TF-style `name` ('model/.../weights:0') + a view of the flat HBM buffer."""

    def __init__(self, name, tensor):
        self.name = name + ":0"
        self.op_name = name
        self.tensor = tensor
        self.shape = tuple(tensor.shape)

    def __repr__(self):
        return "<Variable %s shape=%s>" % (self.name, self.shape)


class StereoNet(object):
    """Registry of named layers (torch tensors backed by engine buffers), of the variables each layer owns, and of the
    disparity outputs; subclasses fill it in `_build_network`.  Public surface = Nets/Stereo_net.py:141-222."""
    _netName = "stereoNet"
    _valid_args = [
        ("split_layer", "name of the layer where the network will be splitted"),
        ("sequence", "flag to use network on a video sequence instead of on single images"),
        ("train_portion", "one among 'BEGIN' or 'END' specify which portion of the network will be trained"),
        ("is_training", "boolean to specify if the network is in train or inference mode"),
    ]
    # kwarg -> default when the caller omits it (the reference prints a warning and carries on, Stereo_net.py:141-157)
    _DEFAULTS = (("split_layers", [None]), ("sequence", False), ("is_training", False))

    @classmethod
    def getPossibleArsg(cls):          # (sic: the reference's spelling)
        return cls._valid_args

    def __init__(self, **kwargs):
        self._layers, self._layer_to_var = OrderedDict(), {}
        self._disparities, self._placeholders = [], []
        self._trainable = OrderedDict()
        bar = "=" * 50
        print("%s\nbuilding %s on the MI355X engine\n%s" % (bar, self._netName, bar))
        args = self._validate_args(kwargs)
        self._preprocess_inputs(args)
        self._build_network(args)
        print("%s ready: %d layers, %d disparity outputs\n%s" % (self._netName, len(self._layers), len(self._disparities), bar))

    # ---- argument handling ------------------------------------------------------------------------------------------
    def _validate_args(self, args):
        for key, default in self._DEFAULTS:
            if key not in args:
                print("WARNING: %s not given, using %r" % (key, default))
                args[key] = default
        if args["split_layers"] != [None]:
            raise NotImplementedError("split_layers / placeholders are not supported by the MI355X engine "
                                      "(no reference driver enables them)")
        portion = args.setdefault("train_portion", "BEGIN")
        if portion not in ("BEGIN", "END"):
            raise Exception("Invalid portion options {}".format(portion))
        # with no split point every layer lies "before the split": BEGIN = all trainable, END = none (Stereo_net.py:63-67)
        self._trainable_portion = (portion == "BEGIN")
        self._sequence = bool(args["sequence"])
        return args

    def _preprocess_inputs(self, args):
        raise NotImplementedError

    def _build_network(self, args):
        raise NotImplementedError

    # ---- registry ---------------------------------------------------------------------------------------------------------
    def _add_to_layers(self, name, op, variables=()):
        """`op`: tensor of the layer; `variables`: the trainable variables created in the layer's scope."""
        owned = list(variables)
        self._layers[name] = op
        self._layer_to_var[name] = owned
        if self._trainable_portion:
            self._trainable.update((v, True) for v in owned)

    def _get_layer_as_input(self, name):
        try:
            return self._layers[name]
        except KeyError:
            raise Exception("Trying to fetch an unknown layer!")

    def __getitem__(self, key):
        return self._layers[key]

    def __str__(self):
        is_pred = lambda t: any(t is d for d in self._disparities)
        return "".join("%s %s: %s\n" % ("Prediction Layer" if is_pred(t) else "Layer", k, tuple(t.shape))
                       for k, t in self._layers.items())

    __repr__ = __str__

    # ---- public getters ----------------------------------------------------------------------------------------------------
    def get_placeholders(self):
        return self._placeholders                  # always empty: no split point, no placeholders

    def get_placeholder(self, name):
        raise Exception("Unable to find placeholder for layer {}".format(name + "_placeholder"))

    def get_all_layers(self):
        return self._layers

    def get_layers_names(self):
        return self._layers.keys()

    def get_disparities(self):
        return self._disparities

    def get_trainable_variables(self):
        return list(self._trainable)

    def get_variables(self, layer_name):
        if layer_name in self._layers and layer_name not in self._layer_to_var:
            return []
        return self._layer_to_var[layer_name]
