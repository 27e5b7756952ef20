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
    _valid_args = [
        ("split_layer", "name of the layer where the network will be splitted"),
        ("sequence", "flag to use network on a video sequence instead of on single images"),
        ("train_portion", "one among 'BEGIN' or 'END' specify which portion of the network will be trained"),
        ("is_training", "boolean to specify if the network is in train or inference mode"),
    ]
    _netName = "stereoNet"

    @classmethod
    def getPossibleArsg(cls):
        return cls._valid_args

    def __init__(self, **kwargs):
        self._layers = OrderedDict()
        self._disparities = []
        self._placeholders = []
        self._placeholderable = []
        self._trainable_variables = OrderedDict()
        self._layer_to_var = {}
        self._after_split = False
        print('=' * 50)
        print('Starting Creation of {}'.format(self._netName))
        print('=' * 50)
        args = self._validate_args(kwargs)
        print('Args Validated, setting up graph')
        self._preprocess_inputs(args)
        print('Meta op to preprocess data created')
        self._build_network(args)
        print('Network ready')
        print('=' * 50)

    # ------------------------------------------------------------------ registry helpers
    def _get_placeholder_name(self, name):
        return name + '_placeholder'

    def _add_to_layers(self, name, op, variables=()):
        """Register `op` (a tensor) under `name` with the variables created in its scope."""
        self._layers[name] = op
        variables = list(variables)
        self._layer_to_var[name] = variables
        if not self._after_split:
            self._placeholderable.append(name)
        if self._after_split != self._train_beginning:
            for v in variables:
                self._trainable_variables[v] = True
        if name in self._split_layers_list:
            self._after_split = True

    def _get_layer_as_input(self, name):
        if self._get_placeholder_name(name) in self._layers:
            return self._layers[self._get_placeholder_name(name)]
        if name in self._layers:
            return self._layers[name]
        raise Exception('Trying to fetch an unknown layer!')

    def __str__(self):
        ss = ""
        for k, l in self._layers.items():
            kind = "Prediction Layer" if any(l is d for d in self._disparities) else "Layer"
            ss += "{} {}: {}\n".format(kind, k, str(tuple(l.shape)))
        return ss

    __repr__ = __str__

    def __getitem__(self, key):
        return self._layers[key]

    # ------------------------------------------------------------------ to be provided by subclasses
    def _preprocess_inputs(self, args):
        raise NotImplementedError

    def _build_network(self, args):
        raise NotImplementedError

    def _validate_args(self, args):
        portion_options = ['BEGIN', 'END']
        if 'split_layers' not in args:
            print('WARNING: no split points selected, the network will flow without interruption')
            args['split_layers'] = [None]
        if 'train_portion' not in args:
            print('WARNING: train_portion not specified, using default END')
            args['train_portion'] = 'END' if args['split_layers'] != [None] else 'BEGIN'
        elif args['train_portion'] not in portion_options:
            raise Exception('Invalid portion options {}'.format(args['train_portion']))
        if 'sequence' not in args:
            print('WARNING: sequence flag not setted, configuring the network for single image adaptation')
            args['sequence'] = False
        if 'is_training' not in args:
            print('WARNING: flag for trainign not setted, using default False')
            args['is_training'] = False
        if args['split_layers'] != [None]:
            raise NotImplementedError('split_layers / placeholders are not supported by the MI355X engine '
                                      '(no reference driver enables them)')
        self._split_layers_list = args['split_layers']
        self._train_beginning = (args['train_portion'] == 'BEGIN')
        self._sequence = args['sequence']
        self._isTraining = False
        return args

    # ------------------------------------------------------------------ public getters (Stereo_net.py:166-222)
    def get_placeholders(self):
        return self._placeholders

    def get_placeholder(self, name):
        placeholder_name = self._get_placeholder_name(name)
        if placeholder_name not in self._layers:
            raise Exception('Unable to find placeholder for layer {}'.format(placeholder_name))
        return self._layers[placeholder_name]

    def get_all_layers(self):
        return self._layers

    def get_layers_names(self):
        return self._layers.keys()

    def get_disparities(self):
        return self._disparities

    def get_trainable_variables(self):
        return list(self._trainable_variables.keys())

    def get_variables(self, layer_name):
        if layer_name in self._layers and layer_name not in self._layer_to_var:
            return []
        return self._layer_to_var[layer_name]
