"""Model registry of the package -- the same two public names as the reference's Nets/__init__.py:4-12:
`STEREO_FACTORY` (display name -> class) and `get_stereo_net(name, args)` (build a model from a kwargs dict)."""
from Nets import DispNet as _dispnet, MadNet as _madnet

# keyed by each class's own `_netName` ("Dispnet", "MADNet"), i.e. the strings the --modelName flag accepts
STEREO_FACTORY = {cls._netName: cls for cls in (_dispnet.DispNet, _madnet.MadNet)}


def get_stereo_net(name, args):
    """Instantiate the registered model `name` with the keyword arguments in the dict `args`."""
    try:
        model_cls = STEREO_FACTORY[name]
    except KeyError:
        raise Exception('Unrecognized network name: {}'.format(name))
    return model_cls(**args)
