"""Plugin loader: the surface of the reference's ``utils/model.py``.

Contract kept (reference utils/model.py:10-47): the backend is the package
whose NAME is the last component of ``--flownet_path`` and it is looked up on
``sys.path`` (not by file path); its ``net.Model`` is constructed as
``Model(device, **kw)`` where ``kw`` are those of ``options2model_kwargs`` that
the constructor can take; ``--sp`` optionally names a state dict (bare, or a
checkpoint with a ``'model'`` entry, utils/serializer.py:99); the model ends up
on ``device``.  ``--flownet_path dvs_of_training_framework_amd`` selects the
HIP build; any other package that honours the contract (DummyNet) loads too.
"""
import importlib
import importlib.util
import inspect
import logging
from pathlib import PurePath

import torch

from .options import options2model_kwargs

log = logging.getLogger(__name__)


def filter_kwargs(func, kwargs):
    """The subset of ``kwargs`` that ``func`` accepts by name; everything if
    it has a ``**`` catch-all.  Dropped names are reported once."""
    params = inspect.signature(func).parameters
    if any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params.values()):
        return kwargs
    taken = {name: value for name, value in kwargs.items() if name in params}
    dropped = sorted(set(kwargs) - set(taken))
    if dropped:
        log.warning('model constructor does not take %s: not passed', dropped)
    return taken


def load_backend(flownet_path):
    """``<flownet_path.name>.net`` as a module.  A missing package raises
    ModuleNotFoundError (from find_spec), a package without ``net`` fails the
    assertion -- the two failure modes of the reference's loader."""
    package = PurePath(flownet_path).name
    name = f'{package}.net'
    spec = importlib.util.find_spec(name)
    assert spec is not None, \
        f'backend module {name} (from --flownet_path {flownet_path}) not found'
    return importlib.import_module(name)


def _weights_from(path, device):
    blob = torch.load(path, map_location=device, weights_only=True)
    return blob.get('model', blob) if isinstance(blob, dict) else blob


def init_model(args, device):
    backend = load_backend(args.flownet_path)
    kwargs = filter_kwargs(backend.Model, options2model_kwargs(args))
    model = backend.Model(device, **kwargs)
    start_point = getattr(args, 'sp', None)
    if start_point is not None:
        model.load_state_dict(_weights_from(start_point, device))
    return model.to(device)
