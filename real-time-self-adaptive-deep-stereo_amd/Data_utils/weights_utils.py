"""Mirror of Data_utils/weights_utils.py:4-75 on top of tf_checkpoint.CheckpointReader (no TensorFlow).

The reference matches checkpoint keys against the graph's global variables by name (`prefix + key` with the
`ignore_list` substrings removed, `mask` = substrings of variables to skip) and restores through a Saver; here the
"graph variables" are the StereoNet's flat-parameter views (`net.get_all_variables()` / `Stereo_net.Variable`)."""
import os

import numpy as np
import torch

from . import tf_checkpoint


def _variables_of(net):
    if isinstance(net, dict):
        return net
    return dict(net._variables)               # op_name -> Stereo_net.Variable (name without the ':0' suffix)


def get_var_to_restore_list(ckpt_path, mask=[], prefix="", ignore_list=[], net=None):
    """-> {checkpoint key: variable}.  `net`: the StereoNet (stands in for tf.GraphKeys.GLOBAL_VARIABLES)."""
    variables_dict = {}
    for name, v in _variables_of(net).items():
        if any(m in name for m in mask):
            continue
        variables_dict[name] = v
    reader = tf_checkpoint.CheckpointReader(ckpt_path)
    var_to_restore = {}
    for key in reader.get_variable_to_shape_map():
        t_key = key
        for ig in ignore_list:
            t_key = t_key.replace(ig, '')
        if prefix + t_key in variables_dict:
            var_to_restore[key] = variables_dict[prefix + t_key]
    return var_to_restore


def restore(ckpt_path, var_to_restore):
    """tf.train.Saver(var_list=var_to_restore).restore(sess, ckpt_path): copy into the variables' storage."""
    reader = tf_checkpoint.CheckpointReader(ckpt_path)
    for key, var in var_to_restore.items():
        a = reader.get_tensor(key)
        t = var.tensor if hasattr(var, "tensor") else var
        if tuple(a.shape) != tuple(t.shape):
            raise ValueError("shape mismatch for %s: checkpoint %s vs variable %s" % (key, a.shape, tuple(t.shape)))
        t.copy_(torch.from_numpy(a.astype(np.float32)))
    return len(var_to_restore)


def check_for_weights_or_restore_them(logdir, net, initial_weights=None, prefix='', ignore_list=[]):
    ckpt = tf_checkpoint.latest_checkpoint(logdir)
    if ckpt:
        print('Found valid checkpoint file: {}'.format(ckpt))
        restore(ckpt, get_var_to_restore_list(ckpt, [], prefix="", net=net))
        step = int(ckpt.split('-')[-1]) if ckpt.split('-')[-1].isdigit() else 0
        return True, step
    elif initial_weights is not None:
        if os.path.isdir(initial_weights):
            initial_weights = tf_checkpoint.latest_checkpoint(initial_weights)
        var_to_restore = get_var_to_restore_list(initial_weights, [], prefix=prefix, ignore_list=ignore_list, net=net)
        print('Found {} variables to restore in {}'.format(len(var_to_restore), initial_weights))
        if len(var_to_restore) > 0:
            restore(initial_weights, var_to_restore)
            return True, 0
        return False, 0
    print('Unable to restore any weight')
    return False, 0
